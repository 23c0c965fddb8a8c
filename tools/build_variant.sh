#!/bin/bash
# Build a variant of the library with extra compiler flags for gemm.hip (kernel experiments):
#   tools/build_variant.sh <name> <flags...>   ->  tools/exp/<name>/libbtvae_hip.so   (use with BTVAE_HIP_LIB=...)
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/tools/exp/$name
mkdir -p $out
cs=$root/behavio_temporal_vae_amd/csrc
/opt/rocm/bin/hipcc -O3 -mllvm -pragma-unroll-threshold=1000000 -std=c++17 -fPIC --offload-arch=gfx950 -fvisibility=hidden -Wall -Wno-unused-function -Wno-unused-result "$@" -c $cs/gemm.hip -o $out/gemm.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $cs/core.o $out/gemm.o $cs/bn.o $cs/dist.o $cs/optim.o $cs/impute.o $cs/feed.o -o $out/libbtvae_hip.so
echo built $out/libbtvae_hip.so
