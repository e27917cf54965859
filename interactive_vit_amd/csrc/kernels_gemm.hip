// Instantiations of the bf16 MFMA GEMM template (gemm_kernel.h) and the per-shape tile choice.
#include "gemm256s_kernel.h"
#include "gemmpe_kernel.h"
#ifdef IVIT_GEMM_ABLATIONS   // microbenchmark-only builds (tools/gemm_bench): the variants that lost (csrc/study/)
#include "study/gemm256p_kernel.h"
#include "study/gemm256ps_kernel.h"
#include "study/gemm160x256_kernel.h"
#include "study/gemm160x256w4_kernel.h"
#endif
#include <algorithm>
#include <cmath>
#include <mutex>
#include <set>
#include <utility>

namespace ivit {

static int device_cu_count();

using Tile128 = GemmTile<2, 2, 4, 4>;   // 128 x 128, 4 waves (64x64 each), 64 KiB LDS, 2 blocks/CU
using Tile160 = GemmTile<2, 2, 5, 4>;   // 160 x 128, 4 waves (80x64 each), 72 KiB LDS, 2 blocks/CU
using Tile256 = GemmTile<2, 4, 8, 4>;   // 256 x 256, 8 waves, plain double buffer (microbenchmark baseline only)

__global__ __launch_bounds__(Tile128::THREADS, 2) void ivit_gemm_bf16_128x128x64(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    gemm_body<Tile128>(p, smem);
}
__global__ __launch_bounds__(Tile160::THREADS, 2) void ivit_gemm_bf16_160x128x64(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    gemm_body<Tile160>(p, smem);
}

// LayerNorm-fold kernel family: the same bodies with their own epilogues.
// _rs = residual add + bf16 copy + row statistics for the next GEMM; _lf = LayerNorm applied in the epilogue.
#define IVIT_LNFOLD_KERNEL(NAME, TILE, EK)                                                    \
    __global__ __launch_bounds__(TILE::THREADS, 2) void NAME(GemmParams p) {                  \
        extern __shared__ __attribute__((aligned(16))) char smem[];                          \
        gemm_body<TILE, false, EK>(p, smem);                                                  \
    }
IVIT_LNFOLD_KERNEL(ivit_gemm_bf16_128x128x64_rs, Tile128, 1)
IVIT_LNFOLD_KERNEL(ivit_gemm_bf16_128x128x64_lf, Tile128, 2)
IVIT_LNFOLD_KERNEL(ivit_gemm_bf16_160x128x64_rs, Tile160, 1)
IVIT_LNFOLD_KERNEL(ivit_gemm_bf16_160x128x64_lf, Tile160, 2)
#undef IVIT_LNFOLD_KERNEL
__global__ __launch_bounds__(Tile256P::THREADS, 2) void ivit_gemm_bf16_256x256x64_stag_rs(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    gemm256s_body<0, false, 1>(p, smem);
}
__global__ __launch_bounds__(Tile256P::THREADS, 2) void ivit_gemm_bf16_256x256x64_stag_lf(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    gemm256s_body<0, false, 2>(p, smem);
}

// persistent 256 x 128 tile with the previous tile's epilogue interleaved into the main loop (gemmpe_kernel.h)
__global__ __launch_bounds__(TilePE::THREADS, 2) void ivit_gemm_bf16_256x128x64_pe(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    gemmpe_body<OpBf16, false>(p, smem);
}
__global__ __launch_bounds__(TilePE::THREADS, 2) void ivit_gemm_bf16_256x128x64_pe_gelu(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    gemmpe_body<OpBf16, true>(p, smem);
}

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
IVIT_DEEP_KERNEL(ivit_gemm_bf16_64x128x64_deep_rs, 1, OpBf16)
IVIT_DEEP_KERNEL(ivit_gemm_bf16_64x128x64_deep_lf, 2, OpBf16)
IVIT_DEEP_KERNEL(ivit_gemm_f16_64x128x64_deep, 0, OpF16)
IVIT_DEEP_KERNEL(ivit_gemm_f16_64x128x64_deep_rs, 1, OpF16)
IVIT_DEEP_KERNEL(ivit_gemm_f16_64x128x64_deep_lf, 2, OpF16)
#undef IVIT_DEEP_KERNEL

// f16 operands (IVIT_PRECISION_F16): the same bodies on v_mfma_f32_16x16x32_f16 with f16 outputs (same rate as bf16, 11 bits)
#define IVIT_F16_KERNEL(NAME, THREADS_, BODY)                                                  \
    __global__ __launch_bounds__(THREADS_, 2) void NAME(GemmParams p) {                        \
        extern __shared__ __attribute__((aligned(16))) char smem[];                          \
        BODY;                                                                                  \
    }
IVIT_F16_KERNEL(ivit_gemm_f16_128x128x64, Tile128::THREADS, (gemm_body<Tile128, false, 0, OpF16>(p, smem)))
IVIT_F16_KERNEL(ivit_gemm_f16_128x128x64_rs, Tile128::THREADS, (gemm_body<Tile128, false, 1, OpF16>(p, smem)))
IVIT_F16_KERNEL(ivit_gemm_f16_128x128x64_lf, Tile128::THREADS, (gemm_body<Tile128, false, 2, OpF16>(p, smem)))
IVIT_F16_KERNEL(ivit_gemm_f16_160x128x64, Tile160::THREADS, (gemm_body<Tile160, false, 0, OpF16>(p, smem)))
IVIT_F16_KERNEL(ivit_gemm_f16_160x128x64_rs, Tile160::THREADS, (gemm_body<Tile160, false, 1, OpF16>(p, smem)))
IVIT_F16_KERNEL(ivit_gemm_f16_160x128x64_lf, Tile160::THREADS, (gemm_body<Tile160, false, 2, OpF16>(p, smem)))
IVIT_F16_KERNEL(ivit_gemm_f16_256x256x64_stag, Tile256P::THREADS, (gemm256s_body<0, false, 0, OpF16>(p, smem)))
IVIT_F16_KERNEL(ivit_gemm_f16_256x256x64_stag_rs, Tile256P::THREADS, (gemm256s_body<0, false, 1, OpF16>(p, smem)))
IVIT_F16_KERNEL(ivit_gemm_f16_256x256x64_stag_lf, Tile256P::THREADS, (gemm256s_body<0, false, 2, OpF16>(p, smem)))
#undef IVIT_F16_KERNEL

// fp8 (e4m3) operands: same tiles, K-tile of 128 elements, two fp8 MFMA steps per 16-B fragment
__global__ __launch_bounds__(Tile128::THREADS, 2) void ivit_gemm_fp8_128x128x128(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    gemm_body<Tile128, true>(p, smem);
}
__global__ __launch_bounds__(Tile160::THREADS, 2) void ivit_gemm_fp8_160x128x128(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    gemm_body<Tile160, true>(p, smem);
}
__global__ __launch_bounds__(Tile256P::THREADS, 2) void ivit_gemm_bf16_256x256x64_stag(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    gemm256s_body<0>(p, smem);
}
__global__ __launch_bounds__(Tile256P::THREADS, 2) void ivit_gemm_fp8_256x256x128_stag(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    gemm256s_body<0, true>(p, smem);
}
#ifdef IVIT_GEMM_ABLATIONS   // microbenchmark-only builds (tools/gemm_bench): the variants that lost, and timing ablations
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
    gemm_body<Tile256>(p, smem);
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
        case GEMM_TILE_64D: return "ivit_gemm_bf16_64x128x64_deep";
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

template <class T, class K>
static hipError_t launch_tile(K kernel, const GemmParams& p, hipStream_t stream, int extra_lds = 0) {
    hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(kernel), T::LDS_BYTES + extra_lds);
    if (e != hipSuccess) return e;
    const int tiles = ceil_div(p.M, T::BM) * ceil_div(p.N, T::BN);
#ifdef IVIT_GEMM_ABLATIONS
    if (p.order == 2) {   // paired order: 512 blocks (some idle), see gemm_body
        hipLaunchKernelGGL(kernel, dim3(512), dim3(T::THREADS), T::LDS_BYTES + extra_lds, stream, p);
        return hipGetLastError();
    }
#endif
    hipLaunchKernelGGL(kernel, dim3(tiles), dim3(T::THREADS), T::LDS_BYTES + extra_lds, stream, p);
    return hipGetLastError();
}

// Tile choice, from tools/gemm_bench on MI355X (ViT-B/16 shapes, M = 12608; TFLOP/s with the fused epilogues;
// profiles/r01f_gemm_microbench.txt, re-checked at the end of the round):
//                       128x128   160x128   256x256 staggered
//   N=2304 K= 768 (qkv)    809       784          823
//   N= 768 K= 768 (proj)   450       534          440        (+ bias + f32 residual)
//   N=3072 K= 768 (mlp1)   745       744          693        (+ bias + erf-GELU)
//   N= 768 K=3072 (mlp2)   717       898          785        (+ bias + f32 residual)
// The tiles differ in FLOP per operand byte (64 / 71 / 128) and in how a grid fills the 256 CUs: 160x128
// puts 474 tiles on the 512 two-per-CU slots at N = 768 where 128x128 needs two rounds; the 256x256 tile
// has the K-loop headroom but one workgroup per CU, so its epilogue is not hidden by
// another workgroup's main loop and its grid quantises worse at this M.  It wins on wide, bf16-output
// shapes with >= 2 rounds of tiles, and everywhere once a grid has many rounds:
// On the ViT-L / ViT-H shapes (M = 73856 / 65792 token rows, 4.5+ rounds of 256 x 256 tiles) the
// staggered 256 x 256 kernel wins every shape by 10-20 % (qkv 1125-1167 vs 929-972 TFLOP/s, mlp2
// 1122-1131 vs 905-953; hipBLASLt: 1210-1260), so the rule is "enough rounds to amortise the
// exposed epilogue and the ragged last round".
bool gemm_prefers_256(int M, int N, int K) {
    if (N >= 2048 && N <= 2560 && K <= 1024 && M >= 4096) return true;   // ViT-B QKV-like (1.8 rounds, bf16 out)
    const double rounds = (double)ceil_div(M, 256) * ceil_div(N, 256) / 256.0;
    return rounds >= 3.0;
}

// Small grids (the interactive path: one to a few 197-token images; the classifier head of any batch): when the 64 x 128 tiles of a
// shape fit one round of one workgroup per CU, every K-tile of the two-stage tiles is a DMA round trip - the deep-ring tile takes
// those (tools/gemm_bench at M = 197: qkv 10.8 -> 7.3 us, proj 13.2 -> 8.1, mlp1 12.1 -> 7.9, mlp2 33.9 -> 19.3; bit-identical).
int gemm_pick_variant(int M, int N, int K) {
    if (K >= 2 * GEMM_BK && ceil_div(M, Tile64D::BM) * ceil_div(N, Tile64D::BN) <= 256) return GEMM_TILE_64D;
    if (gemm_prefers_256(M, N, K)) return GEMM_TILE_256S;
    struct Cand { int v, bm, bn; double speed; };
    static const Cand cands[] = {
        {GEMM_TILE_160, Tile160::BM, Tile160::BN, 1.03},
        {GEMM_TILE_128, Tile128::BM, Tile128::BN, 1.00},
    };
    int best = GEMM_TILE_160;
    double best_t = 1e300;
    for (const Cand& c : cands) {
        const double tiles = (double)ceil_div(M, c.bm) * ceil_div(N, c.bn);
        const double rounds = std::ceil(tiles / 512.0);                 // 256 CUs x 2 resident workgroups
        const double t = rounds * c.bm * c.bn / c.speed;
        if (t < best_t) { best_t = t; best = c.v; }
    }
    return best;
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
#endif

static int device_cu_count() {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    return cus;
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

hipError_t launch_gemm_variant(const GemmParams& p, int variant, hipStream_t stream) {
    if (p.M <= 0 || p.N <= 0) return hipSuccess;
    if (p.K <= 0 || p.K % GEMM_BK != 0) return hipErrorInvalidValue;
    if ((p.lda % 8) || (p.ldw % 8)) return hipErrorInvalidValue;   // 16-B aligned rows for the DMA
    if ((p.ldo % 4) || (p.resid && (p.ldr % 4))) return hipErrorInvalidValue;
    if (variant == GEMM_TILE_PE) return launch_pe(p, stream);
    const int family = p.epi == EPI_BIAS_RESID_STATS ? 1 : (p.epi == EPI_LNFOLD_BF16 || p.epi == EPI_LNFOLD_GELU_BF16) ? 2 : 0;
    if (family) {   // LayerNorm-fold epilogues: their own instantiations of the three product tiles
        if (p.grp_in != 0) return hipErrorInvalidValue;
        if (family == 1 && (!p.ln_part || !p.resid || !p.xb || (p.ldxb % 4))) return hipErrorInvalidValue;
        if (family == 2 && ((!p.ln_stats && !p.ln_part) || !p.ln_s || p.ln_dim <= 0 || p.ln_dim > 64 * GEMM_LN_SLOTS)) return hipErrorInvalidValue;
    }
    if (p.f16) {   // f16 operands: the three product tiles, every epilogue family
        const int extra = family == 2 ? 8 : 0;   // _lf kernels keep (mean, rstd) of their tile's rows in BM * 8 bytes of LDS behind the operand stages
        switch (variant) {
            case GEMM_TILE_128: return launch_tile<Tile128>(family == 0 ? ivit_gemm_f16_128x128x64 : family == 1 ? ivit_gemm_f16_128x128x64_rs : ivit_gemm_f16_128x128x64_lf, p, stream, Tile128::BM * extra);
            case GEMM_TILE_160: return launch_tile<Tile160>(family == 0 ? ivit_gemm_f16_160x128x64 : family == 1 ? ivit_gemm_f16_160x128x64_rs : ivit_gemm_f16_160x128x64_lf, p, stream, Tile160::BM * extra);
            case GEMM_TILE_256S: return launch_tile<Tile256P>(family == 0 ? ivit_gemm_f16_256x256x64_stag : family == 1 ? ivit_gemm_f16_256x256x64_stag_rs : ivit_gemm_f16_256x256x64_stag_lf, p, stream, Tile256P::BM * extra);
            case GEMM_TILE_64D: return launch_tile<Tile64D>(family == 0 ? ivit_gemm_f16_64x128x64_deep : family == 1 ? ivit_gemm_f16_64x128x64_deep_rs : ivit_gemm_f16_64x128x64_deep_lf, p, stream, Tile64D::BM * extra);
            default: return hipErrorInvalidValue;
        }
    }
    if (family) {
        switch (variant) {   // _lf kernels keep (mean, rstd) of their tile's rows in BM * 8 bytes of LDS behind the operand stages
            case GEMM_TILE_128: return family == 1 ? launch_tile<Tile128>(ivit_gemm_bf16_128x128x64_rs, p, stream) : launch_tile<Tile128>(ivit_gemm_bf16_128x128x64_lf, p, stream, Tile128::BM * 8);
            case GEMM_TILE_160: return family == 1 ? launch_tile<Tile160>(ivit_gemm_bf16_160x128x64_rs, p, stream) : launch_tile<Tile160>(ivit_gemm_bf16_160x128x64_lf, p, stream, Tile160::BM * 8);
            case GEMM_TILE_256S: return family == 1 ? launch_tile<Tile256P>(ivit_gemm_bf16_256x256x64_stag_rs, p, stream) : launch_tile<Tile256P>(ivit_gemm_bf16_256x256x64_stag_lf, p, stream, Tile256P::BM * 8);
            case GEMM_TILE_64D: return family == 1 ? launch_tile<Tile64D>(ivit_gemm_bf16_64x128x64_deep_rs, p, stream) : launch_tile<Tile64D>(ivit_gemm_bf16_64x128x64_deep_lf, p, stream, Tile64D::BM * 8);
            default: return hipErrorInvalidValue;
        }
    }
    switch (variant) {
        case GEMM_TILE_64D: return launch_tile<Tile64D>(ivit_gemm_bf16_64x128x64_deep, p, stream);
        case GEMM_TILE_128: return launch_tile<Tile128>(ivit_gemm_bf16_128x128x64, p, stream);
        case GEMM_TILE_160: return launch_tile<Tile160>(ivit_gemm_bf16_160x128x64, p, stream);
#ifdef IVIT_GEMM_ABLATIONS
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
#endif
        case GEMM_TILE_256S:
#ifdef IVIT_GEMM_ABLATIONS
            if (p.debug == 1) return launch_tile<Tile256P>(ivit_gemm_256stag_nodma, p, stream);
            if (p.debug == 2) return launch_tile<Tile256P>(ivit_gemm_256stag_nomfma, p, stream);
#endif
            return launch_tile<Tile256P>(ivit_gemm_bf16_256x256x64_stag, p, stream);
    }
    return hipErrorInvalidValue;
}

static int fp8_tile(const GemmParams& p) {
    const double t160 = std::ceil((double)ceil_div(p.M, Tile160::BM) * ceil_div(p.N, Tile160::BN) / 512.0) * Tile160::BM / 1.03;
    const double t128 = std::ceil((double)ceil_div(p.M, Tile128::BM) * ceil_div(p.N, Tile128::BN) / 512.0) * Tile128::BM;
    if (gemm_prefers_256(p.M, p.N, p.K)) return GEMM_TILE_256S;
    return t160 <= t128 ? GEMM_TILE_160 : GEMM_TILE_128;
}

const char* gemm_fp8_kernel_name(const GemmParams& p) {
    switch (fp8_tile(p)) {
        case GEMM_TILE_256S: return "ivit_gemm_fp8_256x256x128_stag";
        case GEMM_TILE_160: return "ivit_gemm_fp8_160x128x128";
    }
    return "ivit_gemm_fp8_128x128x128";
}

const char* gemm_kernel_name(const GemmParams& p) {
    const int v = gemm_pick_variant(p.M, p.N, p.K);
    const int family = p.epi == EPI_BIAS_RESID_STATS ? 1 : (p.epi == EPI_LNFOLD_BF16 || p.epi == EPI_LNFOLD_GELU_BF16) ? 2 : 0;
    if (v == GEMM_TILE_64D) {
        static const char* deep[2][3] = {{"ivit_gemm_bf16_64x128x64_deep", "ivit_gemm_bf16_64x128x64_deep_rs", "ivit_gemm_bf16_64x128x64_deep_lf"},
                                         {"ivit_gemm_f16_64x128x64_deep", "ivit_gemm_f16_64x128x64_deep_rs", "ivit_gemm_f16_64x128x64_deep_lf"}};
        return deep[p.f16 ? 1 : 0][family];
    }
    static const char* names[3][3] = {
        {"ivit_gemm_bf16_128x128x64", "ivit_gemm_bf16_128x128x64_rs", "ivit_gemm_bf16_128x128x64_lf"},
        {"ivit_gemm_bf16_160x128x64", "ivit_gemm_bf16_160x128x64_rs", "ivit_gemm_bf16_160x128x64_lf"},
        {"ivit_gemm_bf16_256x256x64_stag", "ivit_gemm_bf16_256x256x64_stag_rs", "ivit_gemm_bf16_256x256x64_stag_lf"}};
    static const char* names16[3][3] = {
        {"ivit_gemm_f16_128x128x64", "ivit_gemm_f16_128x128x64_rs", "ivit_gemm_f16_128x128x64_lf"},
        {"ivit_gemm_f16_160x128x64", "ivit_gemm_f16_160x128x64_rs", "ivit_gemm_f16_160x128x64_lf"},
        {"ivit_gemm_f16_256x256x64_stag", "ivit_gemm_f16_256x256x64_stag_rs", "ivit_gemm_f16_256x256x64_stag_lf"}};
    return (p.f16 ? names16 : names)[v == GEMM_TILE_128 ? 0 : v == GEMM_TILE_160 ? 1 : 2][family];
}

hipError_t launch_gemm_fp8(const GemmParams& p, hipStream_t stream) {
    if (p.M <= 0 || p.N <= 0) return hipSuccess;
    if (p.K <= 0 || p.K % 128 != 0 || !p.colscale) return hipErrorInvalidValue;
    if ((p.lda % 16) || (p.ldw % 16) || (p.ldo % 4) || (p.resid && (p.ldr % 4))) return hipErrorInvalidValue;
    switch (fp8_tile(p)) {
        case GEMM_TILE_256S: return launch_tile<Tile256P>(ivit_gemm_fp8_256x256x128_stag, p, stream);
        case GEMM_TILE_160: return launch_tile<Tile160>(ivit_gemm_fp8_160x128x128, p, stream);
    }
    return launch_tile<Tile128>(ivit_gemm_fp8_128x128x128, p, stream);
}

hipError_t launch_gemm(const GemmParams& p, hipStream_t stream) {
    return launch_gemm_variant(p, gemm_pick_variant(p.M, p.N, p.K), stream);
}

}  // namespace ivit
