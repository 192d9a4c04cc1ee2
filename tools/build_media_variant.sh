#!/bin/bash
# A tuning variant of the kernel families with media / textures only (compilation 2 of rt_kernels.hip: the book-two cover's family):
# one hipcc run with the given -D's, linked with the in-tree objects of everything else.  Minutes instead of a whole library build.
#   tools/build_media_variant.sh <name> "<-D...>"   -> ray-tracer_amd/lib/variants/librt_<name>.so
set -e
cd "$(dirname "$0")/../ray-tracer_amd/csrc"
name=$1; defs=$2
mkdir -p ../lib/variants _obj_$name
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-parameter --offload-arch=gfx950 -mllvm -simplifycfg-sink-common=false \
  $defs -DRT_TU_PART=2 -DRT_PLAIN_DIV3 -Rpass-analysis=kernel-resource-usage -c rt_kernels.hip -o _obj_$name/rt_kernels_media.o 2> _obj_$name/resource_usage.txt
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../lib/variants/librt_$name.so _obj/rt_host.o _obj/rt_api.o _obj/rt_kernels.o _obj_$name/rt_kernels_media.o _obj/rt_kernels_nested.o _obj/rt_kernels_reclds.o
echo "built librt_$name.so ($defs)"
