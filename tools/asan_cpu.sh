#!/bin/bash
# CPU only: the host side of librt_mi355x.so (flattener, BVH builder, C ABI) built with AddressSanitizer + UBSan and
# the CPU test-suite run through it (GPU sanitizers are not available on the pool; the kernels' object is linked as is).
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
O=${1:-/tmp/rt_asan}
mkdir -p $O
cd $R/ray-tracer_amd/csrc
make -s
# the two hashes of csrc/Makefile (HASHED, KERNEL_HASHED): the Python side compares rt_version() with the tree
H=$(cat rt_kernels.hip rt_lane.h rt_libm.h rt_libm_tables.h rt_types.h rt_lds.h rt_host.cpp rt_host.h rt_api.cpp rt_scene_priv.h ../../include/rt_mi355x.h ../../include/rt_rng.h Makefile | sha256sum | cut -c1-16)
K=$(cat rt_kernels.hip rt_lane.h rt_libm.h rt_libm_tables.h rt_types.h rt_lds.h ../../include/rt_rng.h Makefile | sha256sum | cut -c1-16)
for f in rt_host rt_api; do
  g++ -O1 -g -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fsanitize=address,undefined -fno-omit-frame-pointer \
      -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ -DRT_SOURCE_HASH="\"$H\"" -DRT_KERNEL_HASH="\"$K\"" -c $f.cpp -o $O/$f.o
done
g++ -shared -fPIC -fsanitize=address,undefined -o $O/librt_asan.so $O/rt_host.o $O/rt_api.o _obj/rt_kernels.o _obj/rt_kernels_media.o _obj/rt_kernels_nested.o _obj/rt_kernels_reclds.o -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,/opt/rocm/lib
cd $R
LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
  RT_MI355X_LIB=$O/librt_asan.so python -m pytest tests -m "not gpu" -x -q --deselect tests/test_host_logic.py::test_the_library_exports_the_header_and_nothing_else  # (the sanitizer build is linked without the export map)
