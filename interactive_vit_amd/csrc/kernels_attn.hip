// Multi-head self-attention core for ViT token counts (up to 608 keys), head dim 64 or 80:
//
//     out[b, q, h*64 + d] = sum_key softmax_key(scale * Q[q].K[key]) * V[key][d]
//
// One workgroup = one (image, head) at every supported length; K and V of the head are staged ONCE
// into LDS by LDS-DMA (global_load_lds_dwordx4: every chunk of the head in flight at once, no
// VGPRs, no ds_write; 197 keys: 28 KiB + 28 KiB, two workgroups per CU).  The 16-query blocks of the
// head are dealt round-robin to the (up to 8) waves.  All keys fit on chip, so the softmax is a single exact
// pass with every score of the wave's queries held in registers - no online rescaling at these
// lengths.  Both contractions run on v_mfma_f32_16x16x32_bf16 with the key index on the
// accumulator ROW:
//   S^T = K . Q^T    A operand = K rows (ds_read_b128, chunk XOR-swizzled by key & 7), B = Q fragments
//         -> lane (fr = lane&15, g = lane>>4) holds S[q = fr][key = 16 f + 4 g + j], j = 0..3
//   O^T = V^T . P^T  A operand = V^T fragments, B operand = P straight from those registers
//         -> lane holds O[q = fr][d = 16 dblk + 4 g + j]: an 8-byte bf16 store per fragment
// The k order inside an MFMA step may be any permutation as long as both operands agree, so the P
// registers of score fragments 2s and 2s+1 ARE the 8 k-slots of step s (slot 8g + j' = key
// 32 s + 16 (j'>>2) + 4 g + (j'&3)), with no lane movement.  V stays ROW-MAJOR in LDS ([key][64 d],
// 128-B rows) and is transposed on the fly by ds_read_b64_tr_b16: per 16-lane group the instruction
// reads a 4-key x 16-d block and hands lane i the 4 keys of column d0 + i - exactly the V^T fragment
// half (keys 32 s + 4 g .. +3, then +16).  V's 16-B chunks are XOR-swizzled by ((key >> 1) & 3) << 1,
// which makes those reads bank-conflict free.  Softmax statistics are fp32; row max / sum are wavefront shuffles across the 4 lane groups.
#include "gemm_kernel.h"
#include <algorithm>

namespace ivit {

// Head dims: 64 (ViT-Ti/B/L: 128-B LDS rows, XOR-swizzled chunks) and 80 (ViT-H/14: 160 B of data in
// 176-B rows - an odd number of 16-B chunks spreads rows over the banks without a swizzle; the
// third 32-deep MFMA step of Q.K^T covers d = 64..95 with 80..95 supplied as zero fragments).
template <int DH, int NKF, bool ODD = false>
struct AttLayout {
    static_assert(DH == 64 || DH == 80, "head dim 64 or 80");
    static constexpr int KEYS = NKF * 16;
    static constexpr int ROW = (DH == 64) ? 128 : 176;       // LDS row stride in bytes
    static constexpr int CHUNKS = DH / 8;                     // 16-B chunks of data per row
    static constexpr int KSTEPS = (DH + 31) / 32;             // MFMA k-steps of Q.K^T
    static constexpr int NDB = DH / 16;                       // 16-wide output column blocks
    // NKF is even (a P.V step is 32 keys).  ODD: the token count needs only NKF - 1 fragments (197 keys:
    // 13), so the LDS images hold (NKF - 1) * 16 rows and the last fragment is never read - its scores are
    // -inf, its probabilities 0.  At 197 keys that is the difference between two and THREE workgroups per
    // CU (53 KB against 57 KB each), i.e. 768 heads in one round of 768 slots instead of 1.5 rounds of 512.
    static constexpr int ROWS = (ODD ? NKF - 1 : NKF) * 16;
    static constexpr int K_BYTES = ROWS * ROW;
    static constexpr int LDS_BYTES = 2 * ROWS * ROW;
    static constexpr int CPR = ROW / 16;                      // 16-B chunk slots per LDS row (dh 80: 10 data + 1 pad)
    static constexpr int K_SWZ = 0, V_SWZ = 1;
    __device__ static int swz(int which, int key) { return DH == 64 ? (which == K_SWZ ? (key & 7) : (((key >> 1) & 3) << 1)) : 0; }
    __device__ static int k_off(int key, int ch) { return key * ROW + ((DH == 64 ? (ch ^ (key & 7)) : ch) << 4); }
    __device__ static int v_off(int key, int ch) { return key * ROW + ((DH == 64 ? (ch ^ (((key >> 1) & 3) << 1)) : ch) << 4); }
};

__device__ __forceinline__ bf16x4 lds_read_tr16(const char* p) {
    typedef __attribute__((ext_vector_type(4))) short s16x4;
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((IVIT_LDS s16x4*)p);
    return __builtin_bit_cast(bf16x4, v);
}

// PROBS = true: the inspection variant behind the `encoder.layers.<i>.attn` node - same staging, QK^T and
// softmax, but instead of P.V it writes the normalised probabilities as f32 [B, H, N, N].
// Stages one operand (K or V columns of the head) into its LDS image.  The image is addressed as a
// linear array of 16-B chunk slots; one DMA instruction fills 64 consecutive slots (LDS address =
// wave-uniform base + 16 * lane) from per-lane source addresses, which is where the row swizzle goes.
// Slots of keys >= N get zeros by ds_write from the lane that would have loaded them (0 * garbage must
// not be NaN in P.V); pad slots (dh 80) are never read and stay untouched.
template <class L>
__device__ __forceinline__ void att_stage(char* lds, const bf16_t* src0, int ld, int N, int which, int wave, int nwaves, int lane) {
    constexpr int SLOTS = L::ROWS * L::CPR;
    constexpr int INSTR = (SLOTS + 63) / 64;
    for (int i = wave; i < INSTR; i += nwaves) {
        const int slot = i * 64 + lane;
        const int key = slot / L::CPR, cl = slot % L::CPR;
        const bool data = slot < SLOTS && cl < L::CHUNKS;
        if (data && key < N) {
            const int ch = cl ^ L::swz(which, key);   // the data chunk that lives in slot cl of this row
            __builtin_amdgcn_global_load_lds((const IVIT_GLOBAL void*)(src0 + (size_t)key * ld + ch * 8),
                                             (IVIT_LDS void*)(lds + i * 1024), 16, 0, 0);
        } else if (data) {
            *reinterpret_cast<u32x4*>(lds + slot * 16) = u32x4{0u, 0u, 0u, 0u};
        }
    }
}

// OP: the 16-bit type of q|k|v, of the softmax numerators fed to P.V and of the output (OpBf16 / OpF16, common.h)
template <int DH, int NKF, bool ODD, bool PROBS, class OP>
__global__ __launch_bounds__(512) void ivit_attention_bf16(AttnParams p) {
    using L = AttLayout<DH, NKF, ODD>;
    constexpr int ATT_DH = DH;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* k_lds = smem;
    char* v_lds = smem + L::K_BYTES;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int fr = lane & 15, g = lane >> 4;
    const int b = blockIdx.z, h = blockIdx.y;
    const int N = p.tokens;
    const int D = p.heads * ATT_DH;
    const size_t row0 = (size_t)b * N;
    const bf16_t* qkv = p.qkv;
    const int ld = p.ldqkv;

    const int nwaves = blockDim.x >> 6;
    const int nblocks = (N + 15) >> 4;   // 16-query blocks; the launcher gives every wave at least one
    const int q0 = wave * 16;

    // ---- K and V by LDS-DMA (row-major, 16-B chunks, swizzle applied on the source address)
    att_stage<L>(k_lds, qkv + row0 * ld + h * ATT_DH + D, ld, N, L::K_SWZ, wave, nwaves, lane);
    att_stage<L>(v_lds, qkv + row0 * ld + h * ATT_DH + 2 * D, ld, N, L::V_SWZ, wave, nwaves, lane);

    // ---- Q fragments of the wave's first query block, behind the DMA queue so that their latency
    // overlaps the staging (B operand: lane holds Q[q + fr][kk*32 + 8g .. +7])
    const bf16x8 zero_frag = {0, 0, 0, 0, 0, 0, 0, 0};
    bf16x8 qf[L::KSTEPS];
    {
        const int qrow = min(q0 + fr, N - 1);
#pragma unroll
        for (int kk = 0; kk < L::KSTEPS; ++kk)
            qf[kk] = (kk * 32 + g * 8 + 8 <= ATT_DH)
                         ? *reinterpret_cast<const bf16x8*>(qkv + (row0 + qrow) * ld + h * ATT_DH + kk * 32 + g * 8) : zero_frag;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's DMA has landed; the barrier publishes everyone's
    __syncthreads();

    // transposed-read addressing: lane i = 4q + p of its 16-lane group supplies &V[key0 + q][d0 + 4p]
    const int tq = fr >> 2, tp = fr & 3;
    const float cexp = p.scale * 1.44269504088896340736f;  // exp(x*scale) = exp2(x*scale*log2 e)

    // The wave's 16-query blocks (wave, wave + nwaves, ...) run one after the other, so only one block's
    // scores (NKF x 4 registers) are live at a time: < 128 VGPRs at 197 keys, i.e. two workgroups per
    // CU and one's staging overlaps the other's math.  (K / V fragments are re-read per block: LDS has room.)
#pragma unroll 1
    for (int blk = wave; blk < nblocks; blk += nwaves) {
        const int qbase = blk * 16;
        asm volatile("" ::: "memory");   // keeps the (block-invariant) K / V fragment reads inside the loop: hoisted, they cost 100+ VGPRs
        if (blk != wave) {   // later blocks: plain load (the first block's fragments were prefetched above)
            const int qrow = min(qbase + fr, N - 1);
#pragma unroll
            for (int kk = 0; kk < L::KSTEPS; ++kk)
                qf[kk] = (kk * 32 + g * 8 + 8 <= ATT_DH)
                             ? *reinterpret_cast<const bf16x8*>(qkv + (row0 + qrow) * ld + h * ATT_DH + kk * 32 + g * 8) : zero_frag;
        }

        // ---- S^T = K Q^T
        f32x4 s[NKF];
#pragma unroll
        for (int f = 0; f < NKF; ++f) {
            if (ODD && f == NKF - 1) { s[f] = f32x4{0.f, 0.f, 0.f, 0.f}; continue; }   // no such rows in LDS: masked to -inf below
            const int key = f * 16 + fr;
            f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < L::KSTEPS; ++kk) {
                // every lane reads (EXEC stays full); chunks past the head dim read the next row's
                // bytes or the pad and are replaced by zeros
                const int ch = kk * 4 + g;
                bf16x8 kf = *reinterpret_cast<const bf16x8*>(k_lds + L::k_off(key, ch < L::CHUNKS ? ch : 0));
                if (ch >= L::CHUNKS) kf = zero_frag;
                a = OP::mfma(kf, qf[kk], a);
            }
            s[f] = a;
        }

        // ---- softmax over keys; only the last two fragments can hold padded keys (KEYS - N < 32)
#pragma unroll
        for (int f = NKF - 2; f < NKF; ++f)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (f * 16 + g * 4 + j >= N) s[f][j] = -INFINITY;
        float mx = -INFINITY;
#pragma unroll
        for (int f = 0; f < NKF; ++f) mx = fmaxf(mx, fmaxf(fmaxf(s[f][0], s[f][1]), fmaxf(s[f][2], s[f][3])));
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float mc = mx * cexp;
        float sum = 0.f;
#pragma unroll
        for (int f = 0; f < NKF; ++f)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float e = __builtin_amdgcn_exp2f(fmaf(s[f][j], cexp, -mc));
                s[f][j] = e;
                sum += e;
            }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        const float inv = 1.0f / sum;

        if (PROBS) {   // lane holds P[q = qbase + fr][key = 16 f + 4 g + j]: a float4 per fragment
            const int q = qbase + fr;
            if (q < N) {
                float* prow = p.probs + (((size_t)b * p.heads + h) * N + q) * N;
#pragma unroll
                for (int f = 0; f < NKF; ++f)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int key = f * 16 + g * 4 + j;
                        if (key < N) prow[key] = s[f][j] * inv;
                    }
            }
            continue;
        }

        // ---- O^T = V^T P^T
        f32x4 o[L::NDB];
#pragma unroll
        for (int d = 0; d < L::NDB; ++d) o[d] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int st = 0; st < NKF / 2; ++st) {
            const f32x4 p0 = s[2 * st], p1 = s[2 * st + 1];
            union { bf16x8 v; unsigned int u[4]; } pk;
            pk.u[0] = OP::pack2(p0[0], p0[1]);
            pk.u[1] = OP::pack2(p0[2], p0[3]);
            pk.u[2] = OP::pack2(p1[0], p1[1]);
            pk.u[3] = OP::pack2(p1[2], p1[3]);
            const int key_lo = 32 * st + 4 * g + tq;       // first half of the k-slots; second half = +16 keys
#pragma unroll
            for (int d = 0; d < L::NDB; ++d) {
                const int chunk = d * 2 + (tp >> 1);       // 16-B chunk of columns d*16 + 4*tp
                const char* lo = v_lds + L::v_off(key_lo, chunk) + (tp & 1) * 8;
                const char* hi = v_lds + L::v_off(key_lo + 16, chunk) + (tp & 1) * 8;
                union { bf16x8 v; bf16x4 h2[2]; } vf;
                vf.h2[0] = lds_read_tr16(lo);
                if (ODD && st == NKF / 2 - 1) vf.h2[1] = bf16x4{0, 0, 0, 0};   // the fragment that is not in LDS: its P is exactly 0
                else vf.h2[1] = lds_read_tr16(hi);
                o[d] = OP::mfma(vf.v, pk.v, o[d]);
            }
        }

        // ---- normalise and store: lane holds O[qbase + fr][16 d + 4 g .. +3]
        // (the guard diverges only here, after the last transposed read of this block, which needs
        // EXEC all ones; the next block's reads run with the full mask again)
        const int q = qbase + fr;
        if (q < N && p.out8) {   // fp8 data path: 4 consecutive d -> one dword of e4m3
            unsigned char* orow8 = p.out8 + (row0 + q) * p.ldo8 + h * ATT_DH + g * 4;
            const float sc = inv * p.scale8;
#pragma unroll
            for (int d = 0; d < L::NDB; ++d)
                *reinterpret_cast<unsigned int*>(orow8 + d * 16) = pack_fp8x4(o[d][0] * sc, o[d][1] * sc, o[d][2] * sc, o[d][3] * sc);
        } else if (q < N) {
            bf16_t* orow = p.out + (row0 + q) * p.ldo + h * ATT_DH + g * 4;
#pragma unroll
            for (int d = 0; d < L::NDB; ++d) {
                u32x2 pk2 = {OP::pack2(o[d][0] * inv, o[d][1] * inv), OP::pack2(o[d][2] * inv, o[d][3] * inv)};
                *reinterpret_cast<u32x2*>(orow + d * 16) = pk2;
            }
        }
    }
}

bool attention_supported(int tokens, int head_dim) {
    if (tokens < 1) return false;
    if (head_dim == 64) return tokens <= 38 * 16;
    if (head_dim == 80) return tokens <= 26 * 16;   // 176-B rows: 26 fragments = 143 KiB of LDS
    return false;
}

template <int DH, int NKF, bool ODD, bool PROBS, class OP>
static hipError_t launch_nkf_op(const AttnParams& p, hipStream_t stream) {
    using L = AttLayout<DH, NKF, ODD>;
    hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(ivit_attention_bf16<DH, NKF, ODD, PROBS, OP>), L::LDS_BYTES);
    if (e != hipSuccess) return e;
    const int blocks = ceil_div(p.tokens, 16);
    // every wave gets at least one 16-query block.  (Measured at 197 keys, three 5-wave workgroups per CU -
    // which the 53-KiB ODD image allows - against two of 7-8 waves: 0.333 vs 0.307 ms per 12 launches; the
    // launch moves 58 + 19 MB in 25 us and is bound by that, not by the 1.5-round grid.  Round 2, same question with up to 13
    // waves - one 16-query block each, one workgroup per CU: 5 / 7 / 8 / 10 / 13 waves -> 0.372 / 0.315 / 0.312 / 0.344 / 0.354 ms.)
    const int waves = std::min(8, blocks);
    dim3 grid(1, p.heads, p.batch);
    hipLaunchKernelGGL((ivit_attention_bf16<DH, NKF, ODD, PROBS, OP>), grid, dim3(waves * 64), L::LDS_BYTES, stream, p);
    return hipGetLastError();
}

template <int DH, int NKF, bool ODD, bool PROBS>
static hipError_t launch_nkf_impl(const AttnParams& p, hipStream_t stream) {
    return p.f16 ? launch_nkf_op<DH, NKF, ODD, PROBS, OpF16>(p, stream) : launch_nkf_op<DH, NKF, ODD, PROBS, OpBf16>(p, stream);
}

template <int DH, int NKF>
static hipError_t launch_nkf(const AttnParams& p, hipStream_t stream) {
    const bool odd = ceil_div(p.tokens, 16) == NKF - 1;   // the last of the NKF fragments holds no key at all
    if (odd) return p.probs ? launch_nkf_impl<DH, NKF, true, true>(p, stream) : launch_nkf_impl<DH, NKF, true, false>(p, stream);
    return p.probs ? launch_nkf_impl<DH, NKF, false, true>(p, stream) : launch_nkf_impl<DH, NKF, false, false>(p, stream);
}

hipError_t launch_attention(const AttnParams& p, hipStream_t stream) {
    if (!attention_supported(p.tokens, p.head_dim)) return hipErrorInvalidValue;
    if ((p.ldqkv % 8) || (!p.probs && !p.out8 && (p.ldo % 4)) || (p.out8 && (p.ldo8 % 4))) return hipErrorInvalidValue;
    const int nkf = round_up(ceil_div(p.tokens, 16), 2);
    if (p.head_dim == 80) {
        if (nkf <= 2) return launch_nkf<80, 2>(p, stream);
        if (nkf <= 8) return launch_nkf<80, 8>(p, stream);
        if (nkf <= 18) return launch_nkf<80, 18>(p, stream);   // ViT-H/14: 257 tokens
        return launch_nkf<80, 26>(p, stream);
    }
    if (nkf <= 2) return launch_nkf<64, 2>(p, stream);
    if (nkf <= 4) return launch_nkf<64, 4>(p, stream);
    if (nkf <= 8) return launch_nkf<64, 8>(p, stream);
    if (nkf <= 14) return launch_nkf<64, 14>(p, stream);   // 197 tokens (224^2 / 16)
    if (nkf <= 18) return launch_nkf<64, 18>(p, stream);   // 257 tokens (224^2 / 14)
    if (nkf <= 26) return launch_nkf<64, 26>(p, stream);
    return launch_nkf<64, 38>(p, stream);                  // 577 tokens (384^2 / 16)
}


// ------------------------------------------------------------------------------------------------------------------
// Fused QKV projection + attention for sequences of up to 224 tokens at head dim 64 (ViT-Ti/B at 224^2: 197 tokens).
//
// One workgroup per (image, head): phase 1 computes the head's q | k | v slice  C[224, 192] = x_img[224, D] . Wh[192, D]^T
// (LayerNorm folded: v = rstd (acc - mean s) + c, the expressions of gemm_epilogue_lnfold, so the values are bit-identical to
// what the QKV GEMM stores) straight into the LDS images that phase 2 - the attention loop of ivit_attention_bf16, unchanged
// but for Q coming from LDS - consumes.  The q|k|v tensor never exists in memory: -58 MB written and -58 MB read per layer and
// one launch less.
//   phase 1: 8 waves = 2 (M: 7 row fragments each) x 4 (N: 3 column fragments each); three-stage LDS-DMA ring of
//            (28 + 24) x 1 KiB pieces per K-step (7 per wave, 4 of the 56 slots repeat a piece so every wave counts the same),
//            counted s_waitcnt vmcnt(7), one raw barrier per K-step; 21 MFMAs and 10 fragment reads per wave and 32-deep half
//   phase 2: Q, K, V images (28 KiB each, rows = tokens, the swizzles of AttLayout<64, 14>) alias the dead ring; K / V rows of
//            tokens >= N are written as zeros (0 x garbage must not be NaN in P.V)
// Rows 197..223 of the tile are the next image's tokens or workspace padding: computed, never used.
struct FusedQkvAttnParams {
    const bf16_t* x; int ldx;            // [B*N, D] 16-bit operand copy of the residual stream
    const bf16_t* w; int ldw;            // [3D, D] LayerNorm-folded in_proj weight (q rows, k rows, v rows)
    const float* c; const float* s;      // [3D] fold vectors
    const float2* ln_part; const float2* ln_stats; float ln_eps; int ln_dim;
    bf16_t* out; int ldo;                // [B*N, D] attention output
    bf16_t* qkv_dbg; int ldq;            // nullptr, or [B*N, 3D]: also store the q|k|v tile (inspection taps)
    int batch, tokens, heads, dim, rows_total;
    float scale;
    unsigned long long* stamps; int debug;
};

#ifdef IVIT_FUSED_WAIT0
#define FUSED_WAIT_A "s_waitcnt lgkmcnt(0)"
#else
#define FUSED_WAIT_A "s_waitcnt lgkmcnt(10)"
#endif
#ifdef IVIT_GEMM_ABLATIONS
#define FUSED_STAMP(slot)                                                                                     \
    do {                                                                                                      \
        if (p.stamps && threadIdx.x == 0) {                                                                   \
            unsigned long long t_;                                                                            \
            asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                     \
            p.stamps[(size_t)blockIdx.x * 128 + (slot)] = t_;                                                 \
        }                                                                                                     \
    } while (0)
// shader-clock stamps of ONE half-step (hs == 12) per wave: slots 16 + wave * 8 + i of the block's 128-slot record
#define FUSED_LOOP_STAMP(i)                                                                                   \
    do {                                                                                                      \
        if (p.stamps && hs == 12 && lane == 0) {                                                              \
            unsigned long long t_;                                                                            \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                         \
            p.stamps[(size_t)blockIdx.x * 128 + 16 + wave * 8 + (i)] = t_;                                    \
        }                                                                                                     \
    } while (0)
#else
#define FUSED_STAMP(slot) do { } while (0)
#define FUSED_LOOP_STAMP(i) do { } while (0)
#endif

// inline-asm memory operations for code that runs next to in-flight LDS-DMA: for an ordinary load or LDS access there hipcc emits
// s_waitcnt vmcnt(0) (it cannot tell that the DMA targets other bytes), draining the operand ring.  The results are valid only after
// the caller's own counted s_waitcnt, which names them as "+v" operands.
__device__ __forceinline__ f32x4 asm_global_load16(const void* ptr) {
    f32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(ptr) : "memory");
    return v;
}
__device__ __forceinline__ void asm_lds_write8(unsigned addr, float a, float b) {
    typedef __attribute__((ext_vector_type(2))) float f32x2;
    const f32x2 v = {a, b};
    asm volatile("ds_write_b64 %0, %1" :: "v"(addr), "v"(v) : "memory");
}
__device__ __forceinline__ void asm_lds_write16(unsigned addr, f32x4 v) { asm volatile("ds_write_b128 %0, %1" :: "v"(addr), "v"(v) : "memory"); }

// fragment reads of one 32-deep half of a staged K-step (row-major 128-byte rows: 2 KiB between 16-row fragments), issued NOT waited
// for, in three groups that are dealt between the MFMAs of the previous half
__device__ __forceinline__ void fused_read_w3(bf16x8 (&wf)[3], unsigned w_addr) {
    asm volatile("ds_read_b128 %0, %3\n\tds_read_b128 %1, %3 offset:2048\n\tds_read_b128 %2, %3 offset:4096"
                 : "=&v"(wf[0]), "=&v"(wf[1]), "=&v"(wf[2]) : "v"(w_addr) : "memory");
}
__device__ __forceinline__ void fused_read_a4(bf16x8 (&af)[7], unsigned a_addr) {
    asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:2048\n\tds_read_b128 %2, %4 offset:4096\n\tds_read_b128 %3, %4 offset:6144"
                 : "=&v"(af[0]), "=&v"(af[1]), "=&v"(af[2]), "=&v"(af[3]) : "v"(a_addr) : "memory");
}
__device__ __forceinline__ void fused_read_a3(bf16x8 (&af)[7], unsigned a_addr) {
    asm volatile("ds_read_b128 %0, %3 offset:8192\n\tds_read_b128 %1, %3 offset:10240\n\tds_read_b128 %2, %3 offset:12288"
                 : "=&v"(af[4]), "=&v"(af[5]), "=&v"(af[6]) : "v"(a_addr) : "memory");
}
__device__ __forceinline__ void fused_wait_frags(bf16x8 (&wf)[3], bf16x8 (&af)[7]) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(wf[0]), "+v"(wf[1]), "+v"(wf[2]), "+v"(af[0]), "+v"(af[1]), "+v"(af[2]), "+v"(af[3]), "+v"(af[4]), "+v"(af[5]), "+v"(af[6])
                 :: "memory");
}

template <class OP>
__global__ __launch_bounds__(512, 2) void ivit_qkv_attention_fused(FusedQkvAttnParams p) {
    using L = AttLayout<64, 14, false>;
    // phase 1 ring: three 64-deep stages of 224 A rows + 192 W rows, 128-byte rows (the image read_frag reads: chunk ^ (row & 7))
    constexpr int A_ROWS = 224, W_ROWS = 192, A_BYTES = A_ROWS * 128, STAGE = (A_ROWS + W_ROWS) * 128, SLOTS = 3, RING = SLOTS * STAGE;
    constexpr int IMG = 224 * 128;   // one of the Q / K / V images
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    // block -> (image, head): the heads of an image on one XCD (they share x_img), images dealt round-robin over the XCDs
    const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3;
    const int b = (jx / p.heads) * 8 + xcd, h = jx % p.heads;
    if (b >= p.batch) return;
    const int N = p.tokens, D = p.dim;
    const size_t row0 = (size_t)b * N;
    const int wr = wave >> 2, wc = wave & 3;
    const unsigned smem_lds = (unsigned)(size_t)(IVIT_LDS char*)smem;
    FUSED_STAMP(0);
    float2* tile_stats = reinterpret_cast<float2*>(smem + RING);
    float* cs_lds = reinterpret_cast<float*>(smem + RING + 224 * 8);   // c[192] | s[192] of this head's q | k | v columns

    // ---- loads of the row statistics (thread r: token row min(r, 223)) and of the fold vectors, BEFORE any DMA: they return first
    const int nslots = (p.ln_dim + 63) >> 6;
    f32x4 raw[GEMM_LN_SLOTS / 2];
    {
        const size_t m = min(row0 + (size_t)min((int)threadIdx.x, 223), (size_t)p.rows_total - 1);
        if (p.ln_stats) {
            raw[0] = asm_global_load16(reinterpret_cast<const void*>(reinterpret_cast<size_t>(p.ln_stats + m) & ~(size_t)15));   // the aligned 16 bytes holding row m's (mean, rstd)
#pragma unroll
            for (int l = 1; l < GEMM_LN_SLOTS / 2; ++l) raw[l] = f32x4{0.f, 0.f, 0.f, 0.f};
        } else {
            const float4* pr = reinterpret_cast<const float4*>(p.ln_part + m * GEMM_LN_SLOTS);
#pragma unroll
            for (int l = 0; l < GEMM_LN_SLOTS / 2; ++l)
                if (2 * l < nslots) raw[l] = asm_global_load16(pr + l);      // uniform
                else raw[l] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    f32x4 csv;
    {
        const int v = threadIdx.x % 96, half = v / 48, q4 = (v % 48) * 4;   // q4: 0..188 within q | k | v
        csv = asm_global_load16((half ? p.s : p.c) + (q4 >> 6) * D + h * 64 + (q4 & 63));
    }

    // ---------------------------------------------------------------- phase 1: C = x_img . Wh^T
    // Two barriers per 64-deep K-step t, each followed at once by 21 MFMAs whose operands are already in registers:
    //   B_t: fragments (t, first half) in set X.  Under MFMA(X): read (t, second half) -> Y; stage pieces 3.. of K-step t+2 (slot of
    //        t-1, free since M_t-1).  Then wait: own pieces of t+1 landed (all of t+2 may be in flight), Y read.
    //   M_t: K-step t+1 has landed for every wave and slot t is free.  Under MFMA(Y): read (t+1, first half) -> X; stage pieces 0..2 of
    //        K-step t+3 into slot t.  Wait: X read.   Two K-steps (104 KB) are in flight throughout.
    // (Reads and staging issued ahead of the MFMAs left the matrix pipe idle ~40 % of a K-step: every wave leaves a barrier at the same
    // time.  Half-stage pieces - 16 rows x 64 B - would allow one barrier per half, but fetch half a cache line per row: slower DMA.)
    const size_t ldx_b = (size_t)p.ldx * 2, ldw_b = (size_t)p.ldw * 2;
    const int nt = D / GEMM_BK, last = nt - 1;
    const bool seven = wave < 4;   // 52 pieces of 8 rows x 128 B per K-step: waves 0-3 stage the 28 of A (7 each), waves 4-7 the 24 of W (6 each)
    const int r_in = lane >> 3, chunk = (lane & 7) ^ r_in;
    const char* src[7];
    int dst[7];
#pragma unroll
    for (int u = 0; u < 7; ++u) {
        if (seven) {
            const int piece = wave * 7 + u;
            const size_t row = min(row0 + (size_t)(piece * 8 + r_in), (size_t)p.rows_total - 1);
            src[u] = reinterpret_cast<const char*>(p.x) + row * ldx_b + chunk * 16;
            dst[u] = piece * 1024;
        } else {
            const int q = (wave - 4) * 6 + min(u, 5), slab = q >> 3, rr = (q & 7) * 8 + r_in;
            src[u] = reinterpret_cast<const char*>(p.w) + (size_t)(slab * D + h * 64 + rr) * ldw_b + chunk * 16;
            dst[u] = A_BYTES + q * 1024;
        }
    }
    // piece u of K-step kt into ring slot `slot`; pieces 0..2 are every wave's, 3..5 too, piece 6 only the A waves'
    auto stage_piece = [&](int kt, int slot, int u) {
        __builtin_amdgcn_global_load_lds((const IVIT_GLOBAL void*)(src[u] + min(kt, last) * 128), (IVIT_LDS void*)(smem + slot * STAGE + dst[u]), 16, 0, 0);
    };
    auto stage_all = [&](int kt, int slot) {
#pragma unroll
        for (int u = 0; u < 6; ++u) stage_piece(kt, slot, u);
        if (seven) stage_piece(kt, slot, 6);
    };
    stage_all(0, 0);
    stage_all(1, 1);
#pragma unroll
    for (int u = 0; u < 3; ++u) stage_piece(2, 2, u);
    FUSED_STAMP(1);

    // ---- statistics and fold vectors into LDS while the ring fills (their loads were issued first: both K-steps may stay in flight)
    if (seven) asm volatile("s_waitcnt vmcnt(17)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
    asm volatile("" : "+v"(raw[0]), "+v"(raw[1]), "+v"(raw[2]), "+v"(raw[3]), "+v"(raw[4]), "+v"(raw[5]), "+v"(raw[6]), "+v"(raw[7]), "+v"(csv));
    asm volatile("" : "+v"(raw[8]), "+v"(raw[9]), "+v"(raw[10]), "+v"(raw[11]), "+v"(raw[12]), "+v"(raw[13]), "+v"(raw[14]), "+v"(raw[15]));
    {
        float mean, rstd;
        if (p.ln_stats) {
            const size_t m = min(row0 + (size_t)min((int)threadIdx.x, 223), (size_t)p.rows_total - 1);
            const bool hi = (reinterpret_cast<size_t>(p.ln_stats + m) & 8) != 0;
            mean = hi ? raw[0][2] : raw[0][0];
            rstd = hi ? raw[0][3] : raw[0][1];
        } else {
            float m2 = 0.f;
            mean = 0.f;
#pragma unroll
            for (int q = 0; q < GEMM_LN_SLOTS; ++q)
                if (q < nslots) ln_chan_update(mean, m2, (q & 1) ? raw[q >> 1][2] : raw[q >> 1][0], (q & 1) ? raw[q >> 1][3] : raw[q >> 1][1], q, p.ln_dim);
            rstd = 1.0f / sqrtf(m2 / (float)p.ln_dim + p.ln_eps);
        }
        if (threadIdx.x < 224) asm_lds_write8(smem_lds + RING + threadIdx.x * 8, mean, rstd);
        if (threadIdx.x >= 224 && threadIdx.x < 224 + 96) {
            const int v = threadIdx.x % 96, half = v / 48, q4 = (v % 48) * 4;
            asm_lds_write16(smem_lds + RING + 224 * 8 + (half * 192 + q4) * 4, csv);
        }
    }

    f32x4 acc[7][3];
#pragma unroll
    for (int i = 0; i < 7; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // fragment read addresses within a ring slot (112 and 48 are multiples of 8: A and W rows share the swizzle term fr & 7); the
    // second 32-deep half of a K-step is the other 64 bytes of every row
    const unsigned swz0 = (unsigned)((fq ^ (fr & 7)) << 4), swz1 = (unsigned)(((4 + fq) ^ (fr & 7)) << 4);
    const unsigned a_rd0 = smem_lds + (wr * 112 + fr) * 128 + swz0, a_rd1 = smem_lds + (wr * 112 + fr) * 128 + swz1;
    const unsigned w_rd0 = smem_lds + A_BYTES + (wc * 48 + fr) * 128 + swz0, w_rd1 = smem_lds + A_BYTES + (wc * 48 + fr) * 128 + swz1;
    bf16x8 afX[7], wfX[3], afY[7], wfY[3];
    // K-step 0 has landed -> its first half into set X
    if (seven) asm volatile("s_waitcnt vmcnt(10) lgkmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(9) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    fused_read_w3(wfX, w_rd0);
    fused_read_a4(afX, a_rd0);
    fused_read_a3(afX, a_rd0);
    int slot = 0;   // = t % 3
    for (int t = 0; t < nt; ++t) {
        const int hs = 2 * t;   // (the stamps' half-step counter)
        const int slot1 = slot == 2 ? 0 : slot + 1, slot2 = slot == 0 ? 2 : slot - 1;   // (t + 1) % 3, (t + 2) % 3
        const unsigned cur = (unsigned)(slot * STAGE), nxt = (unsigned)(slot1 * STAGE);
        auto mfma_row = [&](bf16x8 (&cw)[3], bf16x8 (&ca)[7], int i) {
#pragma unroll
            for (int j = 0; j < 3; ++j) acc[i][j] = OP::mfma(cw[j], ca[i], acc[i][j]);
            __builtin_amdgcn_sched_barrier(0);
        };
        // ---- first half
        FUSED_LOOP_STAMP(0);
        fused_wait_frags(wfX, afX);
        FUSED_LOOP_STAMP(1);
        __builtin_amdgcn_s_barrier();                                   // B_t
        FUSED_LOOP_STAMP(2);
        __builtin_amdgcn_sched_barrier(0);
        mfma_row(wfX, afX, 0);
        fused_read_w3(wfY, w_rd1 + cur);
        __builtin_amdgcn_sched_barrier(0);
        mfma_row(wfX, afX, 1);
        fused_read_a4(afY, a_rd1 + cur);
        __builtin_amdgcn_sched_barrier(0);
        mfma_row(wfX, afX, 2);
        fused_read_a3(afY, a_rd1 + cur);
        __builtin_amdgcn_sched_barrier(0);
        mfma_row(wfX, afX, 3);
        stage_piece(t + 2, slot2, 3);
        __builtin_amdgcn_sched_barrier(0);
        mfma_row(wfX, afX, 4);
        stage_piece(t + 2, slot2, 4);
        __builtin_amdgcn_sched_barrier(0);
        mfma_row(wfX, afX, 5);
        stage_piece(t + 2, slot2, 5);
        if (seven) stage_piece(t + 2, slot2, 6);
        __builtin_amdgcn_sched_barrier(0);
        mfma_row(wfX, afX, 6);
        // ---- second half
        FUSED_LOOP_STAMP(3);
        if (seven) asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");   // own pieces of K-step t+1 landed (all of t+2 may be in flight)
        fused_wait_frags(wfY, afY);
        FUSED_LOOP_STAMP(4);
        __builtin_amdgcn_s_barrier();                                   // M_t
        FUSED_LOOP_STAMP(5);
        __builtin_amdgcn_sched_barrier(0);
        mfma_row(wfY, afY, 0);
        fused_read_w3(wfX, w_rd0 + nxt);
        __builtin_amdgcn_sched_barrier(0);
        mfma_row(wfY, afY, 1);
        fused_read_a4(afX, a_rd0 + nxt);
        __builtin_amdgcn_sched_barrier(0);
        mfma_row(wfY, afY, 2);
        fused_read_a3(afX, a_rd0 + nxt);
        __builtin_amdgcn_sched_barrier(0);
        mfma_row(wfY, afY, 3);
        stage_piece(t + 3, slot, 0);                                    // slot t % 3: its last reads (Y) were waited for before M_t
        __builtin_amdgcn_sched_barrier(0);
        mfma_row(wfY, afY, 4);
        stage_piece(t + 3, slot, 1);
        __builtin_amdgcn_sched_barrier(0);
        mfma_row(wfY, afY, 5);
        stage_piece(t + 3, slot, 2);
        __builtin_amdgcn_sched_barrier(0);
        mfma_row(wfY, afY, 6);
        FUSED_LOOP_STAMP(6);
        slot = slot1;
    }
    fused_wait_frags(wfX, afX);                         // the read past the last K-step (never used)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the surplus stagings past K still write the ring
    __syncthreads();
    FUSED_STAMP(2);

    // ---------------------------------------------------------------- epilogue: LayerNorm fold -> Q / K / V images in LDS
    char* q_lds = smem;
    char* k_lds = smem + IMG;
    char* v_lds = smem + 2 * IMG;
    {
        float4 c4[3], s4[3];
        int slab[3], dcol[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int nloc = wc * 48 + j * 16 + fq * 4;   // 0..191: q | k | v of this head
            slab[j] = nloc >> 6; dcol[j] = nloc & 63;
            c4[j] = *reinterpret_cast<const float4*>(cs_lds + nloc);
            s4[j] = *reinterpret_cast<const float4*>(cs_lds + 192 + nloc);
        }
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            const int m = wr * 112 + i * 16 + fr;        // token of this image
            const float2 st = tile_stats[m];
            const float mu = st.x, rs = st.y;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const float v0 = fmaf(rs, fmaf(-mu, s4[j].x, acc[i][j][0]), c4[j].x), v1 = fmaf(rs, fmaf(-mu, s4[j].y, acc[i][j][1]), c4[j].y);
                const float v2 = fmaf(rs, fmaf(-mu, s4[j].z, acc[i][j][2]), c4[j].z), v3 = fmaf(rs, fmaf(-mu, s4[j].w, acc[i][j][3]), c4[j].w);
                u32x2 pk = {OP::pack2(v0, v1), OP::pack2(v2, v3)};
                if (p.qkv_dbg && m < N)
                    *reinterpret_cast<u32x2*>(p.qkv_dbg + (row0 + m) * p.ldq + slab[j] * D + h * 64 + dcol[j]) = pk;
                if (slab[j] != 0 && m >= N) pk = u32x2{0u, 0u};     // padded keys: zeros in K and V
                const int ch = dcol[j] >> 3, sub = (dcol[j] & 7) * 2;
                char* img = slab[j] == 0 ? q_lds : slab[j] == 1 ? k_lds : v_lds;
                const int off = slab[j] == 2 ? L::v_off(m, ch) : L::k_off(m, ch);
                *reinterpret_cast<u32x2*>(img + off + sub) = pk;
            }
        }
    }
    __syncthreads();
    FUSED_STAMP(3);

    // ---------------------------------------------------------------- phase 2: attention (ivit_attention_bf16's loop, Q from LDS)
    constexpr int NKF = 14;
    const int g = fq;
    const int nwaves = 8;
    const int nblocks = (N + 15) >> 4;
    const int tq = fr >> 2, tp = fr & 3;
    const float cexp = p.scale * 1.44269504088896340736f;
#pragma unroll 1
    for (int blk = wave; blk < nblocks; blk += nwaves) {
        const int qbase = blk * 16;
        asm volatile("" ::: "memory");
        bf16x8 qf[2];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) qf[kk] = *reinterpret_cast<const bf16x8*>(q_lds + L::k_off(qbase + fr, kk * 4 + g));
        f32x4 s[NKF];
#pragma unroll
        for (int f = 0; f < NKF; ++f) {
            const int key = f * 16 + fr;
            f32x4 a = {0.f, 0.f, 0.f, 0.f};
            if (f * 16 < N) {            // uniform: short sequences skip the key fragments past their last token
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    const bf16x8 kf = *reinterpret_cast<const bf16x8*>(k_lds + L::k_off(key, kk * 4 + g));
                    a = OP::mfma(kf, qf[kk], a);
                }
            }
            if (f * 16 + 16 > N) {       // uniform: the fragment holds keys >= N
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (f * 16 + g * 4 + j >= N) a[j] = -INFINITY;
            }
            s[f] = a;
        }
        float mx = -INFINITY;
#pragma unroll
        for (int f = 0; f < NKF; ++f) mx = fmaxf(mx, fmaxf(fmaxf(s[f][0], s[f][1]), fmaxf(s[f][2], s[f][3])));
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float mc = mx * cexp;
        float sum = 0.f;
#pragma unroll
        for (int f = 0; f < NKF; ++f)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float e = __builtin_amdgcn_exp2f(fmaf(s[f][j], cexp, -mc));
                s[f][j] = e;
                sum += e;
            }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        const float inv = 1.0f / sum;
        f32x4 o[4];
#pragma unroll
        for (int d = 0; d < 4; ++d) o[d] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int st = 0; st < NKF / 2; ++st) {
            const f32x4 p0 = s[2 * st], p1 = s[2 * st + 1];
            union { bf16x8 v; unsigned int u[4]; } pk;
            pk.u[0] = OP::pack2(p0[0], p0[1]);
            pk.u[1] = OP::pack2(p0[2], p0[3]);
            pk.u[2] = OP::pack2(p1[0], p1[1]);
            pk.u[3] = OP::pack2(p1[2], p1[3]);
            const int key_lo = 32 * st + 4 * g + tq;
            if (32 * st >= N) continue;  // uniform: P is all zeros from here on
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const int chunk2 = d * 2 + (tp >> 1);
                const char* lo = v_lds + L::v_off(key_lo, chunk2) + (tp & 1) * 8;
                const char* hi = v_lds + L::v_off(key_lo + 16, chunk2) + (tp & 1) * 8;
                union { bf16x8 v; bf16x4 h2[2]; } vf;
                vf.h2[0] = lds_read_tr16(lo);
                vf.h2[1] = lds_read_tr16(hi);
                o[d] = OP::mfma(vf.v, pk.v, o[d]);
            }
        }
        const int q = qbase + fr;
        if (q < N) {
            bf16_t* orow = p.out + (row0 + q) * p.ldo + h * 64 + g * 4;
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                u32x2 pk2 = {OP::pack2(o[d][0] * inv, o[d][1] * inv), OP::pack2(o[d][2] * inv, o[d][3] * inv)};
                *reinterpret_cast<u32x2*>(orow + d * 16) = pk2;
            }
        }
    }
    FUSED_STAMP(4);
}

bool fused_qkv_attention_supported(int tokens, int head_dim, int dim) {
    return head_dim == 64 && tokens >= 1 && tokens <= 224 && dim % GEMM_BK == 0 && dim / GEMM_BK >= 2 && dim <= 64 * GEMM_LN_SLOTS;
}

hipError_t launch_fused_qkv_attention(const FusedQkvAttnArgs& a, hipStream_t stream) {
    if (!fused_qkv_attention_supported(a.tokens, a.head_dim, a.dim)) return hipErrorInvalidValue;
    if ((a.ldx % 8) || (a.ldw % 8) || (a.ldo % 4) || (!a.ln_part && !a.ln_stats) || !a.c || !a.s) return hipErrorInvalidValue;
    FusedQkvAttnParams p{};
    p.x = a.x; p.ldx = a.ldx; p.w = a.w; p.ldw = a.ldw; p.c = a.c; p.s = a.s;
    p.ln_part = a.ln_part; p.ln_stats = a.ln_stats; p.ln_eps = a.ln_eps; p.ln_dim = a.dim;
    p.out = a.out; p.ldo = a.ldo; p.qkv_dbg = a.qkv_dbg; p.ldq = a.ldq;
    p.batch = a.batch; p.tokens = a.tokens; p.heads = a.heads; p.dim = a.dim; p.rows_total = a.rows_total; p.scale = a.scale;
    p.stamps = a.stamps; p.debug = a.debug;
    constexpr int LDS = 3 * (224 + 192) * 128 + 224 * 8 + 2 * 192 * 4;   // operand ring + row statistics + fold vectors = 163072 B   // operand ring + row statistics + fold vectors = 163072 B
    const int grid = ceil_div(a.batch, 8) * 8 * a.heads;
    if (a.f16) {
        hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(ivit_qkv_attention_fused<OpF16>), LDS);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(ivit_qkv_attention_fused<OpF16>, dim3(grid), dim3(512), LDS, stream, p);
    } else {
        hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(ivit_qkv_attention_fused<OpBf16>), LDS);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(ivit_qkv_attention_fused<OpBf16>, dim3(grid), dim3(512), LDS, stream, p);
    }
    return hipGetLastError();
}

}  // namespace ivit
