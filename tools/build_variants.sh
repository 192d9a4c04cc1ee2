#!/bin/bash
# build tuning variants of the library: tools/build_variants.sh name "KDEFS" [FAST=1]
set -e
cd "$(dirname "$0")/../ray-tracer_amd/csrc"
name=$1; defs=$2; shift 2
mkdir -p ../lib/variants
make -s OBJ=_obj_$name OUT=../lib/variants/librt_$name.so KDEFS="$defs" "$@" 2>&1 | grep -v remark || true
grep -E "VGPRs:|ScratchSize" _obj_$name/resource_usage.txt | sed -n '13,14p;15,16p' | tr '\n' ' '; echo " <- $name (spheres-only, lens, no-count)"
