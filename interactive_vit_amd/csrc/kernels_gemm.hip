// Instantiations of the 16-bit / fp8 MFMA GEMM templates (gemm_kernel.h, gemm256s_kernel.h) and the per-shape tile choice.
#include "gemm256s_kernel.h"
#ifdef IVIT_GEMM_ABLATIONS   // microbenchmark-only builds (tools/gemm_bench): the variants that lost (csrc/study/)
#include "study/gemmp_kernel.h"
#include "study/gemmpe_kernel.h"
#include "study/gemm256p_kernel.h"
#include "study/gemm256ps_kernel.h"
#include "study/gemm160x256_kernel.h"
#include "study/gemm160x256w4_kernel.h"
#endif
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <map>
#include <mutex>
#include <string>
#include <set>
#include <utility>

namespace ivit {

static int device_cu_count();

using Tile128 = GemmTile<2, 2, 4, 4>;   // 128 x 128, 4 waves (64x64 each)
using Tile160 = GemmTile<2, 2, 5, 4>;   // 160 x 128, 4 waves (80x64 each)
using Tile256 = GemmTile<2, 4, 8, 4>;   // 256 x 256, 8 waves, plain double buffer (microbenchmark baseline only)

// ---- the product tiles for grids of many workgroups: ONE operand stage per workgroup, THREE workgroups per CU (gemm_kernel.h:
// gemm_body_sb).  Epilogue families: classic (run-time kind), _rs = residual add + 16-bit copy + row statistics for the next GEMM,
// _lf = LayerNorm applied in the epilogue.
#define IVIT_SB_KERNEL(NAME, TILE, FP8, EK, OP)                                               \
    __global__ __launch_bounds__(TILE::THREADS, 3) void NAME(GemmParams p) {                  \
        extern __shared__ __attribute__((aligned(16))) char smem[];                          \
        gemm_body_sb<TILE, FP8, EK, OP>(p, smem);                                             \
    }
// (_f32 = the classic f32-output epilogues - EPI_BIAS_F32 / EPI_BIAS_RESID_F32 / EPI_BIAS_ROWADD_F32 - as their own instantiations: gemm_kernel.h, KIND)
IVIT_SB_KERNEL(ivit_gemm_bf16_128x128x64_sb, Tile128, false, 0, OpBf16)
IVIT_SB_KERNEL(ivit_gemm_bf16_128x128x64_sb_f32, Tile128, false, 3, OpBf16)
IVIT_SB_KERNEL(ivit_gemm_bf16_160x128x64_sb_f32, Tile160, false, 3, OpBf16)
IVIT_SB_KERNEL(ivit_gemm_f16_128x128x64_sb_f32, Tile128, false, 3, OpF16)
IVIT_SB_KERNEL(ivit_gemm_f16_160x128x64_sb_f32, Tile160, false, 3, OpF16)
IVIT_SB_KERNEL(ivit_gemm_fp8_128x128x128_sb_f32, Tile128, true, 3, OpBf16)
IVIT_SB_KERNEL(ivit_gemm_fp8_160x128x128_sb_f32, Tile160, true, 3, OpBf16)
IVIT_SB_KERNEL(ivit_gemm_bf16_128x128x64_sb_rs, Tile128, false, 1, OpBf16)
IVIT_SB_KERNEL(ivit_gemm_bf16_128x128x64_sb_lf, Tile128, false, 2, OpBf16)
IVIT_SB_KERNEL(ivit_gemm_bf16_160x128x64_sb, Tile160, false, 0, OpBf16)
IVIT_SB_KERNEL(ivit_gemm_bf16_160x128x64_sb_rs, Tile160, false, 1, OpBf16)
IVIT_SB_KERNEL(ivit_gemm_bf16_160x128x64_sb_lf, Tile160, false, 2, OpBf16)
IVIT_SB_KERNEL(ivit_gemm_f16_128x128x64_sb, Tile128, false, 0, OpF16)
IVIT_SB_KERNEL(ivit_gemm_f16_128x128x64_sb_rs, Tile128, false, 1, OpF16)
IVIT_SB_KERNEL(ivit_gemm_f16_128x128x64_sb_lf, Tile128, false, 2, OpF16)
IVIT_SB_KERNEL(ivit_gemm_f16_160x128x64_sb, Tile160, false, 0, OpF16)
IVIT_SB_KERNEL(ivit_gemm_f16_160x128x64_sb_rs, Tile160, false, 1, OpF16)
IVIT_SB_KERNEL(ivit_gemm_f16_160x128x64_sb_lf, Tile160, false, 2, OpF16)
IVIT_SB_KERNEL(ivit_gemm_fp8_128x128x128_sb, Tile128, true, 0, OpBf16)   // e4m3 operands, K-tile of 128 elements
IVIT_SB_KERNEL(ivit_gemm_fp8_160x128x128_sb, Tile160, true, 0, OpBf16)
#undef IVIT_SB_KERNEL

// ---- two operand stages, two workgroups per CU (gemm_kernel.h: gemm_body): grids that fit ONE round of its 512 slots (N = D at ViT-B/16:
// out-projection, MLP down, patch embedding).  In a single round a third workgroup per CU has nothing to overlap with, and in a whole
// forward these f32-output GEMMs measure 1-4 % faster here than on the single-stage form (tools/libivit_abl.so A/B).
#define IVIT_2ST_KERNEL(NAME, TILE, EK, OP)                                                   \
    __global__ __launch_bounds__(TILE::THREADS, 2) void NAME(GemmParams p) {                  \
        extern __shared__ __attribute__((aligned(16))) char smem[];                          \
        gemm_body<TILE, false, EK, OP>(p, smem);                                              \
    }
IVIT_2ST_KERNEL(ivit_gemm_bf16_160x128x64, Tile160, 0, OpBf16)
IVIT_2ST_KERNEL(ivit_gemm_bf16_160x128x64_f32, Tile160, 3, OpBf16)
IVIT_2ST_KERNEL(ivit_gemm_f16_160x128x64_f32, Tile160, 3, OpF16)
IVIT_2ST_KERNEL(ivit_gemm_bf16_160x128x64_rs, Tile160, 1, OpBf16)
IVIT_2ST_KERNEL(ivit_gemm_bf16_160x128x64_lf, Tile160, 2, OpBf16)
IVIT_2ST_KERNEL(ivit_gemm_f16_160x128x64, Tile160, 0, OpF16)
IVIT_2ST_KERNEL(ivit_gemm_f16_160x128x64_rs, Tile160, 1, OpF16)
IVIT_2ST_KERNEL(ivit_gemm_f16_160x128x64_lf, Tile160, 2, OpF16)

// ---- 256 x 256 staggered tile, one workgroup per CU: grids with >= 3 rounds of such tiles (ViT-L / ViT-H batches)
#define IVIT_256S_KERNEL(NAME, FP8, EK, OP)                                                   \
    __global__ __launch_bounds__(Tile256P::THREADS, 2) void NAME(GemmParams p) {              \
        extern __shared__ __attribute__((aligned(16))) char smem[];                          \
        gemm256s_body<0, FP8, EK, OP>(p, smem);                                               \
    }
IVIT_256S_KERNEL(ivit_gemm_bf16_256x256x64_stag, false, 0, OpBf16)
IVIT_256S_KERNEL(ivit_gemm_bf16_256x256x64_stag_rs, false, 1, OpBf16)
IVIT_256S_KERNEL(ivit_gemm_bf16_256x256x64_stag_lf, false, 2, OpBf16)
IVIT_256S_KERNEL(ivit_gemm_f16_256x256x64_stag, false, 0, OpF16)
IVIT_256S_KERNEL(ivit_gemm_f16_256x256x64_stag_rs, false, 1, OpF16)
IVIT_256S_KERNEL(ivit_gemm_f16_256x256x64_stag_lf, false, 2, OpF16)
IVIT_256S_KERNEL(ivit_gemm_fp8_256x256x128_stag, true, 0, OpBf16)
IVIT_256S_KERNEL(ivit_gemm_bf16_256x256x64_stag_f32, false, 3, OpBf16)
IVIT_256S_KERNEL(ivit_gemm_f16_256x256x64_stag_f32, false, 3, OpF16)
IVIT_256S_KERNEL(ivit_gemm_fp8_256x256x128_stag_f32, true, 3, OpBf16)
#undef IVIT_256S_KERNEL

// small-M tile with a deep DMA ring (gemm_kernel.h: gemm_body_deep): 64 x 128, 8 waves of 16 x 64 (two per SIMD: one wave's DMA issue
// and LDS reads run beside its partner's MFMAs), four stages = 96 KiB LDS, one workgroup per CU.  Measured at M = 197 (one image):
// four waves of 32 x 64 took 0.5 us per K-tile with four stages AND with six - issue-bound, not latency-bound.
using Tile64D = GemmTileDeep<4, 2, 1, 4, 4>;
#define IVIT_DEEP_KERNEL(NAME, EK, OP)                                                       \
    __global__ __launch_bounds__(Tile64D::THREADS, 1) void NAME(GemmParams p) {              \
        extern __shared__ __attribute__((aligned(16))) char smem[];                          \
        gemm_body_deep<Tile64D, EK, OP>(p, smem);                                             \
    }
IVIT_DEEP_KERNEL(ivit_gemm_bf16_64x128x64_deep, 0, OpBf16)
IVIT_DEEP_KERNEL(ivit_gemm_bf16_64x128x64_deep_f32, 3, OpBf16)
IVIT_DEEP_KERNEL(ivit_gemm_f16_64x128x64_deep_f32, 3, OpF16)
IVIT_DEEP_KERNEL(ivit_gemm_bf16_64x128x64_deep_rs, 1, OpBf16)
IVIT_DEEP_KERNEL(ivit_gemm_bf16_64x128x64_deep_lf, 2, OpBf16)
IVIT_DEEP_KERNEL(ivit_gemm_f16_64x128x64_deep, 0, OpF16)
IVIT_DEEP_KERNEL(ivit_gemm_f16_64x128x64_deep_rs, 1, OpF16)
IVIT_DEEP_KERNEL(ivit_gemm_f16_64x128x64_deep_lf, 2, OpF16)
#undef IVIT_DEEP_KERNEL
// e4m3 operands on the same ring (round 5): the tail rows of the e4m3 256 x 256 grids (gemm_tail_rows)
#define IVIT_DEEP8_KERNEL(NAME, EK)                                                          \
    __global__ __launch_bounds__(Tile64D::THREADS, 1) void NAME(GemmParams p) {              \
        extern __shared__ __attribute__((aligned(16))) char smem[];                          \
        gemm_body_deep<Tile64D, EK, OpBf16, true>(p, smem);                                   \
    }
IVIT_DEEP8_KERNEL(ivit_gemm_fp8_64x128x128_deep_f32, 3)
#undef IVIT_DEEP8_KERNEL

#ifdef IVIT_GEMM_ABLATIONS   // microbenchmark-only builds (tools/gemm_bench): the variants that lost, and timing ablations
// the two-stage form of the 128 x 128 tile (the 160 x 128 one is a product kernel; it is the bit-identity reference of tools/gemm_bench)
IVIT_2ST_KERNEL(ivit_gemm_bf16_128x128x64, Tile128, -1, OpBf16)
IVIT_2ST_KERNEL(ivit_gemm_bf16_128x128x64_rs, Tile128, 1, OpBf16)
IVIT_2ST_KERNEL(ivit_gemm_bf16_128x128x64_lf, Tile128, 2, OpBf16)
// round 3 (late): single-stage tiles with MORE rows per workgroup, two workgroups per CU - fewer operand bytes per FLOP on the CU's vector-memory
// return path (45-52 B/clk/CU whatever the source: tools/dma_l1_probe): 256 x 128 on four waves of 128 x 64 (11.7 B/KFLOP against 14.1), 192 x 128 (13.0)
using Tile256x128 = GemmTile<2, 2, 8, 4>;
using Tile192 = GemmTile<2, 2, 6, 4>;
#define IVIT_SB2_KERNEL(NAME, TILE, EK)                                                       \
    __global__ __launch_bounds__(TILE::THREADS, 2) void NAME(GemmParams p) {                  \
        extern __shared__ __attribute__((aligned(16))) char smem[];                          \
        gemm_body_sb<TILE, false, EK, OpBf16>(p, smem);                                       \
    }
IVIT_SB2_KERNEL(ivit_gemm_bf16_256x128x64_sb, Tile256x128, -1)
IVIT_SB2_KERNEL(ivit_gemm_bf16_256x128x64_sb_rs, Tile256x128, 1)
IVIT_SB2_KERNEL(ivit_gemm_bf16_256x128x64_sb_lf, Tile256x128, 2)
IVIT_SB2_KERNEL(ivit_gemm_bf16_192x128x64_sb, Tile192, -1)
IVIT_SB2_KERNEL(ivit_gemm_bf16_192x128x64_sb_rs, Tile192, 1)
IVIT_SB2_KERNEL(ivit_gemm_bf16_192x128x64_sb_lf, Tile192, 2)
// persistent two-per-CU workgroups (study/gemmp_kernel.h; round 3): bit-identical, SLOWER (mlp1 + fold + GELU 92.8 us against 80.0; section
// stamps: profiles/r03a_persist_sections.txt) - IVIT_PERSIST_DUAL 1 = the finished tile's epilogue drained inside the next tile's loop
#ifndef IVIT_PERSIST_DUAL
#define IVIT_PERSIST_DUAL true
#endif
#define IVIT_PERSIST_KERNEL(NAME, TILE, OP, GELU)                                             \
    __global__ __launch_bounds__(TILE::THREADS, 2) void NAME(GemmParams p) {                  \
        extern __shared__ __attribute__((aligned(16))) char smem[];                          \
        gemm_body_persist_lf<TILE, OP, GELU, IVIT_PERSIST_DUAL>(p, smem);                     \
    }
IVIT_PERSIST_KERNEL(ivit_gemm_bf16_160x128x64_plf, Tile160, OpBf16, false)
IVIT_PERSIST_KERNEL(ivit_gemm_bf16_160x128x64_plf_gelu, Tile160, OpBf16, true)
IVIT_PERSIST_KERNEL(ivit_gemm_bf16_128x128x64_plf, Tile128, OpBf16, false)
IVIT_PERSIST_KERNEL(ivit_gemm_bf16_128x128x64_plf_gelu, Tile128, OpBf16, true)
#undef IVIT_PERSIST_KERNEL
// eight-wave forms of the two-per-CU tiles (four waves per SIMD, <= 128 registers): -2...-6 % only (the L2 -> LDS path is the limit, not latency hiding)
using Tile128W8A = GemmTile<2, 4, 4, 2>;   // wave tile 64 x 32
using Tile128W8B = GemmTile<4, 2, 2, 4>;   // wave tile 32 x 64
using Tile160W8 = GemmTile<2, 4, 5, 2>;    // wave tile 80 x 32
#define IVIT_W8_KERNEL(NAME, TILE, EK)                                                        \
    __global__ __launch_bounds__(TILE::THREADS, 4) void NAME(GemmParams p) {                  \
        extern __shared__ __attribute__((aligned(16))) char smem[];                          \
        gemm_body<TILE, false, EK>(p, smem);                                                  \
    }
IVIT_W8_KERNEL(ivit_gemm_bf16_128x128x64_w8a, Tile128W8A, -1)
IVIT_W8_KERNEL(ivit_gemm_bf16_128x128x64_w8a_lf, Tile128W8A, 2)
IVIT_W8_KERNEL(ivit_gemm_bf16_128x128x64_w8b, Tile128W8B, -1)
IVIT_W8_KERNEL(ivit_gemm_bf16_128x128x64_w8b_lf, Tile128W8B, 2)
IVIT_W8_KERNEL(ivit_gemm_bf16_128x128x64_w8b_rs, Tile128W8B, 1)
IVIT_W8_KERNEL(ivit_gemm_bf16_160x128x64_w8, Tile160W8, -1)
IVIT_W8_KERNEL(ivit_gemm_bf16_160x128x64_w8_lf, Tile160W8, 2)
#undef IVIT_W8_KERNEL
// persistent 256 x 128 tile with the previous tile's epilogue interleaved into the main loop (study/gemmpe_kernel.h; round 2, ties)
__global__ __launch_bounds__(TilePE::THREADS, 2) void ivit_gemm_bf16_256x128x64_pe(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    gemmpe_body<OpBf16, false>(p, smem);
}
__global__ __launch_bounds__(TilePE::THREADS, 2) void ivit_gemm_bf16_256x128x64_pe_gelu(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    gemmpe_body<OpBf16, true>(p, smem);
}
// (the 128 x 128 / 160 x 128 tiles on the deep ring of gemm_body_deep, one 4-wave workgroup per CU, were measured too: 87-138 us where the
// two-stage two-per-CU kernels take 53-77 us - one wave per SIMD cannot cover its own DMA issue)
__global__ __launch_bounds__(Tile160x256W4::THREADS, 1) void ivit_gemm_bf16_160x256x64_w4(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    gemm160x256w4_body(p, smem);
}
__global__ __launch_bounds__(Tile160x256::THREADS, 2) void ivit_gemm_bf16_160x256x64(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    gemm160x256_body(p, smem);
}
__global__ __launch_bounds__(Tile256::THREADS, 2) void ivit_gemm_bf16_256x256x64(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    gemm_body<Tile256, false, -1>(p, smem);
}
__global__ __launch_bounds__(Tile256P::THREADS, 2) void ivit_gemm_bf16_256x256x64_pipe(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    gemm256_body<0>(p, smem);
}
__global__ __launch_bounds__(Tile256P::THREADS, 2) void ivit_gemm_bf16_256x256x64_persist(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    gemm256ps_body(p, smem);
}
__global__ __launch_bounds__(Tile256P::THREADS, 2) void ivit_gemm_256pipe_nodma(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    gemm256_body<1>(p, smem);
}
__global__ __launch_bounds__(Tile256P::THREADS, 2) void ivit_gemm_256pipe_nomfma(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    gemm256_body<2>(p, smem);
}
__global__ __launch_bounds__(Tile256P::THREADS, 2) void ivit_gemm_256stag_nodma(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    gemm256s_body<1>(p, smem);
}
__global__ __launch_bounds__(Tile256P::THREADS, 2) void ivit_gemm_256stag_nomfma(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    gemm256s_body<2>(p, smem);
}
#endif

// epilogue family of a call: 0 classic 16-bit / fp8 outputs, 3 classic f32 outputs ("_f32"), 1 residual + statistics ("_rs"), 2 LayerNorm fold ("_lf")
static int gemm_family(int epi) {
    if (epi == EPI_BIAS_RESID_STATS || epi == EPI_BIAS_ROWADD_STATS) return 1;
    if (epi == EPI_LNFOLD_BF16 || epi == EPI_LNFOLD_GELU_BF16) return 2;
    if (epi == EPI_BIAS_F32 || epi == EPI_BIAS_RESID_F32 || epi == EPI_BIAS_ROWADD_F32) return 3;
    return 0;
}

const char* gemm_variant_name(int v) {
    switch (v) {
        case GEMM_TILE_128: return "ivit_gemm_bf16_128x128x64";
        case GEMM_TILE_160: return "ivit_gemm_bf16_160x128x64";
        case GEMM_TILE_256: return "ivit_gemm_bf16_256x256x64";
        case GEMM_TILE_256P: return "ivit_gemm_bf16_256x256x64_pipe";
        case GEMM_TILE_256S: return "ivit_gemm_bf16_256x256x64_stag";
        case GEMM_TILE_160X256: return "ivit_gemm_bf16_160x256x64";
        case GEMM_TILE_160X256W4: return "ivit_gemm_bf16_160x256x64_w4";
        case GEMM_TILE_256PS: return "ivit_gemm_bf16_256x256x64_persist";
        case GEMM_TILE_PE: return "ivit_gemm_bf16_256x128x64_pe";
        case GEMM_TILE_256X128SB: return "ivit_gemm_bf16_256x128x64_sb";
        case GEMM_TILE_192SB: return "ivit_gemm_bf16_192x128x64_sb";
        case GEMM_TILE_64D: return "ivit_gemm_bf16_64x128x64_deep";
        case GEMM_TILE_P160: return "ivit_gemm_bf16_160x128x64_plf";
        case GEMM_TILE_P128: return "ivit_gemm_bf16_128x128x64_plf";
        case GEMM_TILE_128W8A: return "ivit_gemm_bf16_128x128x64_w8a";
        case GEMM_TILE_128W8B: return "ivit_gemm_bf16_128x128x64_w8b";
        case GEMM_TILE_160W8: return "ivit_gemm_bf16_160x128x64_w8";
        case GEMM_TILE_128SB: return "ivit_gemm_bf16_128x128x64_sb";
        case GEMM_TILE_160SB: return "ivit_gemm_bf16_160x128x64_sb";
    }
    return "?";
}

hipError_t ensure_dynamic_lds(const void* kernel, int bytes) {
    static std::mutex mu;
    static std::set<std::pair<int, const void*>> configured;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lk(mu);
    if (configured.count({dev, kernel})) return hipSuccess;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) configured.insert({dev, kernel});
    return e;
}

// one workgroup per output tile; `lds` = the kernel's dynamic LDS (operand stages + BM * 8 bytes of row statistics for the _lf kernels)
template <class T, class K>
static hipError_t launch_grid(K kernel, const GemmParams& p, hipStream_t stream, int lds) {
    hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(kernel), lds);
    if (e != hipSuccess) return e;
    const int tiles = ceil_div(p.M, T::BM) * ceil_div(p.N, T::BN);
#ifdef IVIT_GEMM_ABLATIONS
    if (p.order == 2) {   // paired order: 512 blocks (some idle), see gemm_body
        hipLaunchKernelGGL(kernel, dim3(512), dim3(T::THREADS), lds, stream, p);
        return hipGetLastError();
    }
#endif
    hipLaunchKernelGGL(kernel, dim3(tiles), dim3(T::THREADS), lds, stream, p);
    return hipGetLastError();
}
template <class T, class K>
static hipError_t launch_tile(K kernel, const GemmParams& p, hipStream_t stream, int extra_lds = 0) { return launch_grid<T>(kernel, p, stream, T::LDS_BYTES + extra_lds); }
template <class T, class K>
static hipError_t launch_sb(K kernel, const GemmParams& p, hipStream_t stream, int extra_lds = 0) { return launch_grid<T>(kernel, p, stream, T::STAGE_BYTES + extra_lds); }

// Tile choice, from tools/gemm_bench on MI355X (ViT-B/16 shapes, M = 12608, us per launch with the fused epilogues; round 3,
// gpurun_out/sb_*.log: two-stage two-per-CU | single-stage three-per-CU | 256 x 256 staggered):
//                                   128x128  160x128 | 128 sb  160 sb | 256 stag
//   N=2304 K= 768 qkv  (fold)         60.0     60.1  |  53.8    53.7  |   52.4
//   N= 768 K= 768 proj (resid+stats)  38.0     30.8  |  35.5    30.1  |   42.1
//   N=3072 K= 768 mlp1 (fold + GELU)  83.8     80.8  |  77.0    72.5  |   85.5
//   N= 768 K=3072 mlp2 (resid+stats)  88.1     69.1  |  79.3    62.5  |   83.1
// The 160 x 128 tile carries 71 FLOP per operand byte against 64 (128 x 128) and quantises best at N = 768 / 3072; the 256 x 256
// tile (128 FLOP per byte, the K-loop headroom of DESIGN.md section 5) runs one workgroup per CU, so its epilogue is exposed and
// its grid quantises worse at this M: it wins once a grid has >= 3 rounds of its tiles - on the ViT-L / ViT-H shapes (M = 73856 /
// 65792 token rows) every shape by 10-20 % over the two-stage 160 x 128 (qkv 1125-1167 vs 929-972 TFLOP/s; hipBLASLt 1210-1260).
bool gemm_prefers_256(int M, int N, int K) {
    // ViT-B QKV-like (16-bit output, 1.8 rounds of 256 x 256 tiles): in a whole forward 51.2 us against 53.5 (160 x 128 single-stage) / 56.2 (two-stage)
    static const int qkv256 = [] { const char* v = getenv("IVIT_QKV_256"); return v ? atoi(v) : 1; }();   // IVIT_QKV_256=0: measurement knob
    if (qkv256 && N >= 2048 && N <= 2560 && K <= 1024 && M >= 4096) return true;
    const double rounds = (double)ceil_div(M, 256) * ceil_div(N, 256) / 256.0;
    return rounds >= 3.0;
}

// Small grids (the interactive path: one to a few 197-token images; the classifier head of any batch): when the 64 x 128 tiles of a
// shape fit one round of one workgroup per CU, every K-tile of a shallow ring is a DMA round trip - the deep-ring tile takes
// those (tools/gemm_bench at M = 197: qkv 10.8 -> 7.3 us, proj 13.2 -> 8.1, mlp1 12.1 -> 7.9, mlp2 33.9 -> 19.3; bit-identical).
int gemm_pick_variant(int M, int N, int K) {
    static const int sq160 = [] { const char* v = getenv("IVIT_SQUARE_160"); return v ? atoi(v) : 0; }();   // study knob: square GEMMs (the out-projection) on three 160 x 128 per CU
    if (sq160 && N == K && N % 4 == 0 && ceil_div(M, Tile160::BM) * ceil_div(N, Tile160::BN) > 512) return GEMM_TILE_160SB;
    if (N % 4) return GEMM_TILE_64D;   // ragged widths (a classifier of any size): the only tile whose edge epilogue guards single elements
    if (K >= 2 * GEMM_BK && ceil_div(M, Tile64D::BM) * ceil_div(N, Tile64D::BN) <= 256) return GEMM_TILE_64D;
    if (gemm_prefers_256(M, N, K)) return GEMM_TILE_256S;
    // 160 x 128 wherever the grid gives every CU work (fewer operand bytes per FLOP than 128 x 128, and all workgroups of a CU share its
    // L2 -> LDS path); the smaller tile only for grids below one workgroup per CU.  One round of the two-per-CU slots: the two-stage form;
    // more: three per CU on one stage.  IVIT_PICK_2STAGE = 0 / 1 forces the single-stage / two-stage form (measurement knob).
    static const int force = [] { const char* v = getenv("IVIT_PICK_2STAGE"); return v ? atoi(v) : -1; }();
    const int tiles160 = ceil_div(M, Tile160::BM) * ceil_div(N, Tile160::BN);
    if (tiles160 < 256) return GEMM_TILE_128SB;
    if (force >= 0) return force ? GEMM_TILE_160 : GEMM_TILE_160SB;
    return tiles160 <= 512 ? GEMM_TILE_160 : GEMM_TILE_160SB;
}

static int device_cu_count() {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    return cus;
}

#ifdef IVIT_GEMM_ABLATIONS
// persistent kernel: one workgroup per CU (or per tile when there are fewer tiles than CUs)
static hipError_t launch_persistent(const GemmParams& p, hipStream_t stream) {
    using T = Tile256P;
    hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(ivit_gemm_bf16_256x256x64_persist), T::LDS_BYTES);
    if (e != hipSuccess) return e;
    const int tiles = ceil_div(p.M, T::BM) * ceil_div(p.N, T::BN);
    const int grid = tiles < device_cu_count() ? tiles : device_cu_count();
    hipLaunchKernelGGL(ivit_gemm_bf16_256x256x64_persist, dim3(grid), dim3(T::THREADS), T::LDS_BYTES, stream, p);
    return hipGetLastError();
}

// ---- persistent LayerNorm-fold kernels (study/gemmp_kernel.h)
bool gemm_persist_supported(const GemmParams& p) {
    if (p.epi != EPI_LNFOLD_BF16 && p.epi != EPI_LNFOLD_GELU_BF16) return false;
    if (p.f16 || p.colscale || p.grp_in || !p.ln_s || !p.ln_part || p.ln_stats) return false;
    if (p.N % 128 || p.K % GEMM_BK || (p.ldo % 8)) return false;
    if (p.ln_dim <= 0 || (p.ln_dim % 128) || p.ln_dim > 128 * PERSIST_MAX_F4) return false;
    if (p.K / GEMM_BK < Tile160::FM + p.ln_dim / 128 + 1) return false;   // slices, then one pair of statistics slots per K-tile
    if ((unsigned long long)p.M * (unsigned)p.ldo * 2ull >= (1ull << 31)) return false;   // 32-bit buffer offsets of the epilogue stores
    return true;
}

template <class T>
static hipError_t launch_persist_lf(const GemmParams& p, hipStream_t stream) {
    if (!gemm_persist_supported(p)) return hipErrorInvalidValue;
    const bool gelu = p.epi == EPI_LNFOLD_GELU_BF16;
    constexpr bool T160 = std::is_same<T, Tile160>::value;
    void (*kernel)(GemmParams) = T160 ? (gelu ? ivit_gemm_bf16_160x128x64_plf_gelu : ivit_gemm_bf16_160x128x64_plf) : (gelu ? ivit_gemm_bf16_128x128x64_plf_gelu : ivit_gemm_bf16_128x128x64_plf);
    hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(kernel), PersistLds<T>::BYTES);
    if (e != hipSuccess) return e;
    const int tiles = ceil_div(p.M, T::BM) * (p.N / T::BN);
    const int grid = std::min(tiles, 2 * device_cu_count());
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(T::THREADS), PersistLds<T>::BYTES, stream, p);
    return hipGetLastError();
}

int gemm_persist_occupancy(int variant) {   // resident workgroups per CU the runtime reports for the bf16 GELU instantiation
    int n = -1;
    if (variant == GEMM_TILE_P160) (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, ivit_gemm_bf16_160x128x64_plf_gelu, Tile160::THREADS, PersistLds<Tile160>::BYTES);
    else (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, ivit_gemm_bf16_128x128x64_plf_gelu, Tile128::THREADS, PersistLds<Tile128>::BYTES);
    return n;
}

bool gemm_pe_supported(const GemmParams& p) {
    const bool fold = p.epi == EPI_LNFOLD_BF16 || p.epi == EPI_LNFOLD_GELU_BF16;
    if (!fold && p.epi != EPI_BIAS_BF16 && p.epi != EPI_BIAS_GELU_BF16) return false;
    if (p.f16) return false;   // bf16 instantiations only
    if (p.N % TilePE::BN || p.K % GEMM_BK || p.colscale || p.grp_in || !p.ln_s || (p.ldo % 8)) return false;
    const int nt = p.K / GEMM_BK;
    if (nt < TilePE::MIN_KT) return false;
    if (fold) {   // statistics arrive as per-slot pairs and the fold pipeline (an even number of 4-slot chunks) fits a tile
        if (!p.ln_part || p.ln_stats || p.ln_dim <= 0 || p.ln_dim > 64 * GEMM_LN_SLOTS) return false;
        const int nchunks = (((p.ln_dim + 63) >> 6) + 3) >> 2;
        if (std::max(2, (nchunks + 1) & ~1) > nt - 6) return false;
    }
    return true;
}

static hipError_t launch_pe(const GemmParams& p, hipStream_t stream) {
    if (!gemm_pe_supported(p)) return hipErrorInvalidValue;
    const bool gelu = p.epi == EPI_BIAS_GELU_BF16 || p.epi == EPI_LNFOLD_GELU_BF16;
    GemmParams q = p;
    if (p.epi == EPI_BIAS_BF16 || p.epi == EPI_BIAS_GELU_BF16) { q.ln_part = nullptr; q.ln_stats = nullptr; q.ln_dim = 0; }   // ln_s must be a zero vector
    auto kernel = gelu ? ivit_gemm_bf16_256x128x64_pe_gelu : ivit_gemm_bf16_256x128x64_pe;
    hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(kernel), TilePE::LDS_BYTES);
    if (e != hipSuccess) return e;
    const int tiles = ceil_div(p.M, TilePE::BM) * (p.N / TilePE::BN);
    const int grid = std::min(tiles, device_cu_count());
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(TilePE::THREADS), TilePE::LDS_BYTES, stream, q);
    return hipGetLastError();
}

// the variants only the microbenchmark can reach; hipErrorNotSupported = not one of them
static hipError_t launch_study_variant(const GemmParams& p, int variant, int family, hipStream_t stream) {
    if (p.f16) return hipErrorNotSupported;
    switch (variant) {
        case GEMM_TILE_PE: return launch_pe(p, stream);
        case GEMM_TILE_P160: return launch_persist_lf<Tile160>(p, stream);
        case GEMM_TILE_P128: return launch_persist_lf<Tile128>(p, stream);
        case GEMM_TILE_128:
            return family == 1 ? launch_tile<Tile128>(ivit_gemm_bf16_128x128x64_rs, p, stream) : family == 2 ? launch_tile<Tile128>(ivit_gemm_bf16_128x128x64_lf, p, stream, Tile128::BM * 8) : launch_tile<Tile128>(ivit_gemm_bf16_128x128x64, p, stream);
        case GEMM_TILE_256X128SB:
            return family == 1 ? launch_sb<Tile256x128>(ivit_gemm_bf16_256x128x64_sb_rs, p, stream) : family == 2 ? launch_sb<Tile256x128>(ivit_gemm_bf16_256x128x64_sb_lf, p, stream, Tile256x128::BM * 8) : launch_sb<Tile256x128>(ivit_gemm_bf16_256x128x64_sb, p, stream);
        case GEMM_TILE_192SB:
            return family == 1 ? launch_sb<Tile192>(ivit_gemm_bf16_192x128x64_sb_rs, p, stream) : family == 2 ? launch_sb<Tile192>(ivit_gemm_bf16_192x128x64_sb_lf, p, stream, Tile192::BM * 8) : launch_sb<Tile192>(ivit_gemm_bf16_192x128x64_sb, p, stream);
        case GEMM_TILE_128W8A:
            if (family == 1) return hipErrorInvalidValue;
            return family == 2 ? launch_tile<Tile128W8A>(ivit_gemm_bf16_128x128x64_w8a_lf, p, stream, Tile128W8A::BM * 8) : launch_tile<Tile128W8A>(ivit_gemm_bf16_128x128x64_w8a, p, stream);
        case GEMM_TILE_160W8:
            if (family == 1) return hipErrorInvalidValue;
            return family == 2 ? launch_tile<Tile160W8>(ivit_gemm_bf16_160x128x64_w8_lf, p, stream, Tile160W8::BM * 8) : launch_tile<Tile160W8>(ivit_gemm_bf16_160x128x64_w8, p, stream);
        case GEMM_TILE_128W8B:
            return family == 2 ? launch_tile<Tile128W8B>(ivit_gemm_bf16_128x128x64_w8b_lf, p, stream, Tile128W8B::BM * 8) : family == 1 ? launch_tile<Tile128W8B>(ivit_gemm_bf16_128x128x64_w8b_rs, p, stream) : launch_tile<Tile128W8B>(ivit_gemm_bf16_128x128x64_w8b, p, stream);
    }
    if (family) return hipErrorNotSupported;
    switch (variant) {
        case GEMM_TILE_256: return launch_tile<Tile256>(ivit_gemm_bf16_256x256x64, p, stream);
        case GEMM_TILE_256P:
            if (p.debug == 1) return launch_tile<Tile256P>(ivit_gemm_256pipe_nodma, p, stream);
            if (p.debug == 2) return launch_tile<Tile256P>(ivit_gemm_256pipe_nomfma, p, stream);
            return launch_tile<Tile256P>(ivit_gemm_bf16_256x256x64_pipe, p, stream);
        case GEMM_TILE_256PS:
            if (p.K < 2 * GEMM_BK) return launch_tile<Tile256P>(ivit_gemm_bf16_256x256x64_stag, p, stream);
            return launch_persistent(p, stream);
        case GEMM_TILE_160X256: return launch_tile<Tile160x256>(ivit_gemm_bf16_160x256x64, p, stream);
        case GEMM_TILE_160X256W4: return launch_tile<Tile160x256W4>(ivit_gemm_bf16_160x256x64_w4, p, stream);
        case GEMM_TILE_256S:
            if (p.debug == 1) return launch_tile<Tile256P>(ivit_gemm_256stag_nodma, p, stream);
            if (p.debug == 2) return launch_tile<Tile256P>(ivit_gemm_256stag_nomfma, p, stream);
            break;
    }
    return hipErrorNotSupported;
}
#else
bool gemm_pe_supported(const GemmParams&) { return false; }
bool gemm_persist_supported(const GemmParams&) { return false; }
#endif

// EPI_BIAS_RESID_STATS on the two-stage 160 x 128 tile, a grid of one to two workgroups per CU (out-projection / MLP down at ViT-B/16 B = 64): the
// workgroups a CU receives second fetch their residual rows before their K loop (gemm_kernel.h: RsPrefetch).  Study knob, OFF by default (round 4:
// microbenchmark -3 %, whole forward 0; it needs the whole-tile load group, -DIVIT_RSG_2STAGE=5, which loses 1.2 % to the one-row-ahead default).
static int rs_prefetch_from(const GemmParams& p, int variant) {
    static const int mode = [] { const char* v = getenv("IVIT_RS_PREFETCH"); return v ? atoi(v) : 0; }();
    if (!mode || variant != GEMM_TILE_160 || p.epi != EPI_BIAS_RESID_STATS || p.grp_in != 0) return 0;
    const int cus = device_cu_count(), tiles = ceil_div(p.M, Tile160::BM) * ceil_div(p.N, Tile160::BN);
    return (tiles > cus && tiles <= 2 * cus) ? cus : 0;
}

hipError_t launch_gemm_variant(const GemmParams& p_in, int variant, hipStream_t stream) {
    GemmParams p = p_in;
    if (p.rs_prefetch_from == 0) p.rs_prefetch_from = rs_prefetch_from(p, variant);
#ifdef IVIT_GEMM_ABLATIONS
    { static const int old_epi = [] { const char* v = getenv("IVIT_OLD_EPI"); return v ? atoi(v) : 0; }(); if (old_epi && !p.debug) p.debug = 7; }
#endif
    if (p.M <= 0 || p.N <= 0) return hipSuccess;
    if (p.K <= 0 || p.K % GEMM_BK != 0) return hipErrorInvalidValue;
    if ((p.lda % 8) || (p.ldw % 8)) return hipErrorInvalidValue;   // 16-B aligned rows for the DMA
    if ((p.ldo % 4) || (p.resid && (p.ldr % 4))) return hipErrorInvalidValue;
    if ((p.N % 4) && variant != GEMM_TILE_64D) return hipErrorInvalidValue;   // only the deep-ring tile guards single elements at the N edge
    const int family = gemm_family(p.epi);
    if (family == 1 || family == 2) {   // LayerNorm-fold epilogues: their own instantiations of the product tiles
        const bool remap = p.epi == EPI_BIAS_ROWADD_STATS;
        if ((p.grp_in != 0) != remap) return hipErrorInvalidValue;
        if (family == 1 && (!p.ln_part || !(remap ? p.rowadd : p.resid) || !p.xb || (p.ldxb % 4) || (remap && (p.ldra % 4)))) return hipErrorInvalidValue;
        if (family == 2 && ((!p.ln_stats && !p.ln_part) || !p.ln_s || p.ln_dim <= 0 || p.ln_dim > 64 * GEMM_LN_SLOTS)) return hipErrorInvalidValue;
    }
#ifdef IVIT_GEMM_ABLATIONS
    {
        const hipError_t e = launch_study_variant(p, variant, family == 3 ? 0 : family, stream);
        if (e != hipErrorNotSupported) return e;
    }
#endif
    // _lf kernels keep (mean, rstd) of their tile's rows in BM * 8 bytes of LDS behind the operand stages
#define IVIT_PICK4(T_, LAUNCH, K0, K1, K2, K3) (family == 0 ? LAUNCH<T_>(K0, p, stream, 0) : family == 1 ? LAUNCH<T_>(K1, p, stream, 0) : family == 2 ? LAUNCH<T_>(K2, p, stream, T_::BM * 8) : LAUNCH<T_>(K3, p, stream, 0))
    if (p.f16) {
        switch (variant) {
            case GEMM_TILE_128SB: return IVIT_PICK4(Tile128, launch_sb, ivit_gemm_f16_128x128x64_sb, ivit_gemm_f16_128x128x64_sb_rs, ivit_gemm_f16_128x128x64_sb_lf, ivit_gemm_f16_128x128x64_sb_f32);
            case GEMM_TILE_160SB: return IVIT_PICK4(Tile160, launch_sb, ivit_gemm_f16_160x128x64_sb, ivit_gemm_f16_160x128x64_sb_rs, ivit_gemm_f16_160x128x64_sb_lf, ivit_gemm_f16_160x128x64_sb_f32);
            case GEMM_TILE_160: return IVIT_PICK4(Tile160, launch_tile, ivit_gemm_f16_160x128x64, ivit_gemm_f16_160x128x64_rs, ivit_gemm_f16_160x128x64_lf, ivit_gemm_f16_160x128x64_f32);
            case GEMM_TILE_256S: return IVIT_PICK4(Tile256P, launch_tile, ivit_gemm_f16_256x256x64_stag, ivit_gemm_f16_256x256x64_stag_rs, ivit_gemm_f16_256x256x64_stag_lf, ivit_gemm_f16_256x256x64_stag_f32);
            case GEMM_TILE_64D: return IVIT_PICK4(Tile64D, launch_tile, ivit_gemm_f16_64x128x64_deep, ivit_gemm_f16_64x128x64_deep_rs, ivit_gemm_f16_64x128x64_deep_lf, ivit_gemm_f16_64x128x64_deep_f32);
        }
        return hipErrorInvalidValue;
    }
    switch (variant) {
        case GEMM_TILE_128SB: return IVIT_PICK4(Tile128, launch_sb, ivit_gemm_bf16_128x128x64_sb, ivit_gemm_bf16_128x128x64_sb_rs, ivit_gemm_bf16_128x128x64_sb_lf, ivit_gemm_bf16_128x128x64_sb_f32);
        case GEMM_TILE_160SB: return IVIT_PICK4(Tile160, launch_sb, ivit_gemm_bf16_160x128x64_sb, ivit_gemm_bf16_160x128x64_sb_rs, ivit_gemm_bf16_160x128x64_sb_lf, ivit_gemm_bf16_160x128x64_sb_f32);
        case GEMM_TILE_160: return IVIT_PICK4(Tile160, launch_tile, ivit_gemm_bf16_160x128x64, ivit_gemm_bf16_160x128x64_rs, ivit_gemm_bf16_160x128x64_lf, ivit_gemm_bf16_160x128x64_f32);
        case GEMM_TILE_256S: return IVIT_PICK4(Tile256P, launch_tile, ivit_gemm_bf16_256x256x64_stag, ivit_gemm_bf16_256x256x64_stag_rs, ivit_gemm_bf16_256x256x64_stag_lf, ivit_gemm_bf16_256x256x64_stag_f32);
        case GEMM_TILE_64D: return IVIT_PICK4(Tile64D, launch_tile, ivit_gemm_bf16_64x128x64_deep, ivit_gemm_bf16_64x128x64_deep_rs, ivit_gemm_bf16_64x128x64_deep_lf, ivit_gemm_bf16_64x128x64_deep_f32);
    }
#undef IVIT_PICK4
    return hipErrorInvalidValue;
}

// fp8 operands: the three-per-CU tiles or the 256 x 256 staggered tile, under the 16-bit rule (no deep-ring form: small fp8 grids take 160 x 128)
static int fp8_tile(const GemmParams& p) {
    // a grid that fits one round of 64 x 128 tiles, one workgroup per CU, AND is deep (K >= 2048 bytes): the deep-ring tile - the peeled tail rows of the 256 x 256
    // grids (gemm_tail_rows); the single-stage tiles pay a DMA round trip per K-tile when they are alone on a CU
    // (f32-output epilogues only: the one e4m3 GEMM of a ViT with K >= 2048 is MLP down)
    if (gemm_family(p.epi) == 3 && p.K >= 2048 && p.M <= 512 && ceil_div(p.M, Tile64D::BM) * ceil_div(p.N, Tile64D::BN) <= 256) return GEMM_TILE_64D;
    // the square out-projection (N = K = D, f32 residual epilogue) on three 160 x 128 workgroups per CU: 226.7 against 232.3 us at ViT-H/14 B = 256
    // (its read-modify-write burst is exposed at one workgroup per CU; round 4, in the forward: 3 865 -> 3 888 img/s)
    if (p.N == p.K && p.epi == EPI_BIAS_RESID_F32 && ceil_div(p.M, Tile160::BM) * ceil_div(p.N, Tile160::BN) >= 256) return GEMM_TILE_160SB;
    // MLP up (GELU + e4m3 quantisation in the epilogue) on three 160 x 128 workgroups per CU as well: with the MFMA time halved that epilogue is a quarter of a
    // 256 x 256 tile's life and nothing hides it at one workgroup per CU.  ViT-H/14 B = 256 in the forward, same box, alternating (round 5, IVIT_FP8_160 = bit mask:
    // 1 MLP up, 2 QKV, 4 MLP down):  0: 3 960 / 3 971 img/s (MLP up 597 us);  1: 4 118 / 4 095 (521 us; +3.6 %);  3: 4 100 / 4 085 (QKV 341 -> 373 us);
    // 5: 3 926 / 3 917 (MLP down 441 -> 527 us)
    static const int m160 = [] { const char* v = getenv("IVIT_FP8_160"); return v ? atoi(v) : 1; }();
    if (ceil_div(p.M, Tile160::BM) * ceil_div(p.N, Tile160::BN) >= 256 &&
        (((m160 & 1) && p.epi == EPI_BIAS_GELU_FP8) || ((m160 & 2) && p.epi == EPI_BIAS_BF16) || ((m160 & 4) && p.epi == EPI_BIAS_RESID_F32))) return GEMM_TILE_160SB;
    if (gemm_prefers_256(p.M, p.N, p.K)) return GEMM_TILE_256S;
    return ceil_div(p.M, Tile160::BM) * ceil_div(p.N, Tile160::BN) >= 256 ? GEMM_TILE_160SB : GEMM_TILE_128SB;
}

const char* gemm_fp8_kernel_name(const GemmParams& p) {
    const bool f32 = gemm_family(p.epi) == 3;
    switch (fp8_tile(p)) {
        case GEMM_TILE_256S: return f32 ? "ivit_gemm_fp8_256x256x128_stag_f32" : "ivit_gemm_fp8_256x256x128_stag";
        case GEMM_TILE_160SB: return f32 ? "ivit_gemm_fp8_160x128x128_sb_f32" : "ivit_gemm_fp8_160x128x128_sb";
        case GEMM_TILE_64D: return "ivit_gemm_fp8_64x128x128_deep_f32";
    }
    return f32 ? "ivit_gemm_fp8_128x128x128_sb_f32" : "ivit_gemm_fp8_128x128x128_sb";
}

const char* gemm_kernel_name(const GemmParams& p) {
    const int v = gemm_pick_variant(p.M, p.N, p.K);
    const int family = gemm_family(p.epi);
    static const char* base[2][5] = {{"ivit_gemm_bf16_64x128x64_deep", "ivit_gemm_bf16_128x128x64_sb", "ivit_gemm_bf16_160x128x64_sb", "ivit_gemm_bf16_256x256x64_stag", "ivit_gemm_bf16_160x128x64"},
                                      {"ivit_gemm_f16_64x128x64_deep", "ivit_gemm_f16_128x128x64_sb", "ivit_gemm_f16_160x128x64_sb", "ivit_gemm_f16_256x256x64_stag", "ivit_gemm_f16_160x128x64"}};
    static const char* suffix[4] = {"", "_rs", "_lf", "_f32"};
    static std::mutex mu;
    static std::map<std::pair<const char*, int>, std::string> names;   // interned: callers keep the pointer
    const char* b = base[p.f16 ? 1 : 0][v == GEMM_TILE_64D ? 0 : v == GEMM_TILE_128SB ? 1 : v == GEMM_TILE_160SB ? 2 : v == GEMM_TILE_160 ? 4 : 3];
    std::lock_guard<std::mutex> lk(mu);
    auto it = names.find({b, family});
    if (it == names.end()) it = names.emplace(std::make_pair(b, family), std::string(b) + suffix[family]).first;
    return it->second.c_str();
}

hipError_t launch_gemm_fp8(const GemmParams& p, hipStream_t stream) {
    if (p.M <= 0 || p.N <= 0) return hipSuccess;
    if (p.K <= 0 || p.K % 128 != 0 || !p.colscale) return hipErrorInvalidValue;
    if ((p.lda % 16) || (p.ldw % 16) || (p.ldo % 4) || (p.resid && (p.ldr % 4))) return hipErrorInvalidValue;
    const bool f32 = gemm_family(p.epi) == 3;
    switch (fp8_tile(p)) {
        case GEMM_TILE_256S: return f32 ? launch_tile<Tile256P>(ivit_gemm_fp8_256x256x128_stag_f32, p, stream) : launch_tile<Tile256P>(ivit_gemm_fp8_256x256x128_stag, p, stream);
        case GEMM_TILE_160SB: return f32 ? launch_sb<Tile160>(ivit_gemm_fp8_160x128x128_sb_f32, p, stream) : launch_sb<Tile160>(ivit_gemm_fp8_160x128x128_sb, p, stream);
        case GEMM_TILE_64D: return launch_tile<Tile64D>(ivit_gemm_fp8_64x128x128_deep_f32, p, stream);
    }
    return f32 ? launch_sb<Tile128>(ivit_gemm_fp8_128x128x128_sb_f32, p, stream) : launch_sb<Tile128>(ivit_gemm_fp8_128x128x128_sb, p, stream);
}

// (mean, rstd) of every row from its statistics pairs, once, for the consumers that would otherwise fold the same pairs at the start of every column tile
// (GemmParams::ln_stats): with 12 - 20 column tiles of 256 per row block (ViT-L / ViT-H) that redundant fold was 35 - 50 us per GEMM.  The same function, so the
// same bits.
__global__ __launch_bounds__(64) void ivit_ln_finalize(const float2* __restrict__ part, int rows, int dim, float eps, float2* __restrict__ stats) {
    const int m = blockIdx.x * 64 + threadIdx.x;   // (one wave per workgroup: 73 856 rows = 1 154 waves, four or five per CU, instead of two rounds of four-wave workgroups)
    if (m < rows) stats[m] = ln_row_stats_from_pairs(part, m, dim, eps);
}

hipError_t launch_ln_finalize(const float2* part, int rows, int dim, float eps, float2* stats, hipStream_t s) {
    if (rows <= 0) return hipSuccess;
    hipLaunchKernelGGL(ivit_ln_finalize, dim3(ceil_div(rows, 64)), dim3(64), 0, s, part, rows, dim, eps, stats);
    return hipGetLastError();
}

hipError_t launch_gemm(const GemmParams& p, hipStream_t stream) {
    return launch_gemm_variant(p, gemm_pick_variant(p.M, p.N, p.K), stream);
}

// Tail split (round 5).  A grid of one-per-CU 256 x 256 tiles runs in rounds of `cus` tiles; when the tile count is a few over a multiple of that, the last
// round is a handful of tiles on an otherwise idle chip - ViT-H/14 at B = 256 has 257 row tiles, so EVERY encoder GEMM ends on a round of 5 ... 20 tiles (out-
// projection / MLP down: 5.02 rounds - a sixth of the launch).  Returns how many rows to peel off the END of M so that what is left runs without that round; the
// caller launches the peeled rows (whole row tiles, so at most a few hundred rows) separately through the dispatcher, which gives them the small-grid tiles -
// same K order, bit-identical outputs.  0 = leave the launch alone (not the 256 x 256 tile, a row remap, fewer than three rounds, or a last round more than an
// eighth full).
// Measured in the ViT-H/14 B = 256 forward (same box, alternating; profiles/r05_tail_split.txt): MLP down (K = 5120, f32 residual epilogue) 824 -> 762 us, out-
// projection 322 -> 308, MLP up 837 -> 830, QKV 587 -> 591 (a last round of 15 tiles on an idle chip is cheaper than a launch of small tiles when K is short and
// the epilogue light) - so only the residual-epilogue GEMMs, deep K and MLP up are split; of the e4m3 GEMMs only MLP down (K = 5120), on the e4m3 deep-ring tile
// (463 -> 443 us; with the single-stage 128 x 128 tile, slow alone on a CU, none of the four gained).
int gemm_tail_rows(const GemmParams& p, bool fp8) {
    static const int wide = [] { const char* v = getenv("IVIT_GEMM_TAIL"); return v ? atoi(v) : 1; }();   // study knob: 2 = also last rounds up to 60 % full, every epilogue family
    static const int tail8 = [] { const char* v = getenv("IVIT_GEMM_TAIL_FP8"); return v ? atoi(v) : 1; }();
    if ((fp8 && (!tail8 || p.K < 2048)) || p.grp_in > 0 || p.M <= 0 || p.N <= 0) return 0;
    if (wide < 2 && gemm_family(p.epi) != 3 && gemm_family(p.epi) != 1 && p.K < 2048 && p.N < 4 * p.K) return 0;   // (N >= 4 K: MLP up - ViT-L/16-384 B = 128 662 -> 648 us, ViT-H 846 -> 839)
    const bool t256 = fp8 ? fp8_tile(p) == GEMM_TILE_256S : gemm_pick_variant(p.M, p.N, p.K) == GEMM_TILE_256S;
    if (!t256) return 0;
    static const int cus = device_cu_count();   // (one device per process: one process per GPU)
    const int tm = ceil_div(p.M, 256), tn = ceil_div(p.N, 256), tiles = tm * tn, rem = tiles % cus;
    if (tiles < 3 * cus || rem == 0 || (wide >= 2 ? rem * 10 > cus * 6 : rem * 8 > cus)) return 0;
    const int r = ceil_div(rem, tn);
    if (r >= tm) return 0;
    GemmParams q = p; q.M = (tm - r) * 256;
    if (!(fp8 ? fp8_tile(q) == GEMM_TILE_256S : gemm_pick_variant(q.M, q.N, q.K) == GEMM_TILE_256S)) return 0;
    return p.M - q.M;
}

// p advanced by m_off rows (every per-row pointer of the launch; fp8: A is a byte matrix)
GemmParams gemm_rows_from(const GemmParams& p, int m_off, bool fp8) {
    GemmParams t = p;
    t.M = p.M - m_off;
    t.A = reinterpret_cast<const bf16_t*>(reinterpret_cast<const char*>(p.A) + (size_t)m_off * p.lda * (fp8 ? 1 : 2));
    const int fam = gemm_family(p.epi);
    const size_t osz = fam == 3 || fam == 1 ? 4 : (p.epi == EPI_BIAS_GELU_FP8 ? 1 : 2);
    t.out = reinterpret_cast<char*>(p.out) + (size_t)m_off * p.ldo * osz;
    if (p.resid) t.resid = p.resid + (size_t)m_off * p.ldr;
    if (p.xb) t.xb = p.xb + (size_t)m_off * p.ldxb;
    if (p.ln_part) t.ln_part = p.ln_part + (size_t)m_off * GEMM_LN_SLOTS;
    if (p.ln_stats) t.ln_stats = p.ln_stats + m_off;
    return t;
}

}  // namespace ivit
