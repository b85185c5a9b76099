#!/bin/bash
# Kernel-variant experiments: rebuild ONE translation unit of libswarm_hip.so with extra flags and link it with
# the shipped objects into swarm_amd/lib/variants/libswarm_hip_<tag>.so (select it with SWARM_HIP_LIB=...).
#   tools/variant.sh <tag> <source.hip> [extra hipcc flags...]
# e.g. tools/variant.sh n8 kernels_generic.hip -DSWH_DISPATCH_ONLY=8      (instantiate nbod 8 only: a quick build)
# The shipped sources carry no experiment switches other than SWH_DISPATCH_ONLY (launch.hpp); a variant is an edit of
# the source under test, built here beside the shipped library.
set -e
tag=$1; src=$2; shift 2
root=$(cd "$(dirname "$0")/.." && pwd)
cs=$root/swarm_amd/csrc
out=$root/swarm_amd/lib/variants
mkdir -p "$out" "$cs/build/var_$tag"
obj=$cs/build/var_$tag/${src%.hip}.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I"$root/include" -Wall -Wno-unused-function "$@" \
    -Rpass-analysis=kernel-resource-usage -c "$cs/$src" -o "$obj" 2> "$cs/build/var_$tag/resource.txt" || { tail -30 "$cs/build/var_$tag/resource.txt"; exit 1; }
others=$(ls "$cs"/build/*.o | grep -v "/${src%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$out/libswarm_hip_$tag.so" $obj $others
echo "$out/libswarm_hip_$tag.so"
