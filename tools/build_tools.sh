#!/bin/bash
# builds the standalone microbenchmarks next to their sources (binaries are git-ignored; they travel
# to the GPU box with the snapshot)
set -e
cd "$(dirname "$0")"
C=../interactive_vit_amd/csrc
hipcc -O3 -std=c++17 --offload-arch=gfx950 -DIVIT_GEMM_ABLATIONS gemm_bench.hip $C/kernels_gemm.hip -o gemm_bench.bin
[ -f attn_bench.hip ] && hipcc -O3 -std=c++17 --offload-arch=gfx950 attn_bench.hip $C/kernels_attn.hip -o attn_bench.bin
echo built
