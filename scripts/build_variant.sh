#!/bin/bash
# scripts/build_variant.sh name [-DMACRO ...]: variants/lib_<name>.so = the library with flacgpu.hip compiled with
# the given macros (ablation and tuning switches of subframe.hpp), for scripts/ab.sh.
set -e
name=$1; shift
cd "$(dirname "$0")/../flac_amd/csrc"
mkdir -p ../../variants
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -mllvm -pragma-unroll-threshold=131072 "$@" -c -o ../../variants/flacgpu_$name.o flacgpu.hip
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -pthread -shared -o ../../variants/lib_$name.so ../../variants/flacgpu_$name.o ../lib/obj/flacgpu_encode.o ../lib/obj/flacmeta.o -ldl
rm -f ../../variants/flacgpu_$name.o
echo built $name
