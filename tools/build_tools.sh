#!/bin/bash
# builds the standalone microbenchmarks next to their sources (binaries are git-ignored; they travel
# to the GPU box with the snapshot).  `build_tools.sh gemm` builds only tools/gemm_bench.bin, `build_tools.sh attn` only tools/attn_bench.bin.
set -e
cd "$(dirname "$0")"
C=../interactive_vit_amd/csrc
F="-O3 -std=c++17 --offload-arch=gfx950 -DIVIT_GEMM_ABLATIONS"
if [ "$1" == "attn" ]; then
    hipcc $F -DATTN_BENCH_OWN_LDS_HELPER attn_bench.hip $C/kernels_attn.hip -o attn_bench.bin
    echo built; exit 0
fi
hipcc $F gemm_bench.hip $C/kernels_gemm.hip -o gemm_bench.bin &
if [ "$1" != "gemm" ]; then
    hipcc $F -DATTN_BENCH_OWN_LDS_HELPER attn_bench.hip $C/kernels_attn.hip -o attn_bench.bin &
    hipcc $F fused_bench.hip $C/kernels_gemm.hip $C/kernels_attn.hip -o fused_bench.bin &
    # the whole engine with every study variant compiled in (IVIT_LIB=tools/libivit_abl.so python bench.py ...: in-situ A/B through the env knobs of kernels_gemm.hip)
    hipcc $F -fPIC -shared $C/engine.hip $C/kernels_gemm.hip $C/kernels_attn.hip $C/kernels_misc.hip $C/kernels_mlp.hip -o libivit_abl.so -Wl,-rpath,/opt/rocm/lib &
    # fused MLP kernel against the two GEMM launches, bitwise + timing + per-workgroup stamps / ablations (links the product library: build it first)
    hipcc -O3 -std=c++17 --offload-arch=gfx950 mlp_fused_bench.hip -L../interactive_vit_amd -livit -Wl,-rpath,'$ORIGIN/../interactive_vit_amd' -o mlp_fused_bench.bin &
    for probe in mfma_peak mfma_pattern mfma_f8_probe permlane_probe dma_l1_probe simd_share_probe; do   # single-file hardware probes
        [ -f $probe.bin ] && [ $probe.bin -nt $probe.hip ] || hipcc -O3 -std=c++17 --offload-arch=gfx950 -Wno-unused-result $probe.hip -o $probe.bin
    done
fi
wait
echo built
