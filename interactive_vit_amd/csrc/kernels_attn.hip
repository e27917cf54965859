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
#include <cstdlib>

#ifndef IVIT_ATT_WIDE_STORES
#define IVIT_ATT_WIDE_STORES 1
#endif

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

// One 16-query block of one (image, head): S^T = K Q^T from the LDS image of K, single exact softmax pass in registers, O^T = V^T P^T with
// V consumed row-major through ds_read_tr16_b64, normalise, store.  Shared by the one-head and the pipelined multi-head kernel.
// OP: the 16-bit type of q|k|v, of the softmax numerators fed to P.V and of the output (OpBf16 / OpF16, common.h)
// wait_v (wave-uniform; true for a wave's FIRST block in the one-head kernel): the V image may still be in flight - K was waited for alone
// so that Q.K^T and the softmax of the first block run under V's half of the staging burst; drain and publish it before the first P.V.
template <int DH, int NKF, bool ODD, bool PROBS, class OP>
__device__ __forceinline__ void att_block(const AttnParams& p, const char* k_lds, const char* v_lds, const bf16x8 (&qf)[AttLayout<DH, NKF, ODD>::KSTEPS],
                                          int qbase, int b, int h, size_t row0, int N, int fr, int g, float cexp, bool wait_v = false) {
    using L = AttLayout<DH, NKF, ODD>;
    constexpr int ATT_DH = DH;
    const bf16x8 zero_frag = {0, 0, 0, 0, 0, 0, 0, 0};
    const int tq = fr >> 2, tp = fr & 3;   // transposed-read addressing: lane i = 4q + p of its 16-lane group supplies &V[key0 + q][d0 + 4p]
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

    if (wait_v) {   // every wave of the workgroup passes here exactly once (its first block)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
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
        return;
    }

    // ---- O^T = V^T P^T
    f32x4 o[L::NDB];
#pragma unroll
    for (int d = 0; d < L::NDB; ++d) o[d] = f32x4{0.f, 0.f, 0.f, 0.f};
    // The transposed V reads of step st are at  vbase[d] + st * 32 rows  (+ 16 rows for the second half of the k-slots): the swizzle term
    // ((key >> 1) & 3 of key = 32 st + 16 half + 4 g + tq) does not depend on st or the half, but the compiler cannot know 4 g + tq < 16 and
    // kept one address register per (st, d, half) - 152 block-invariant registers at 577 keys, hoisted in front of the block loop, which
    // spilled 52 dwords when the kernel was last recompiled (ViT-L/16-384 attention 340 -> 645 us per launch; round 3).  Written as base +
    // constant the offsets are instruction immediates.
    int vbase[L::NDB];
#pragma unroll
    for (int d = 0; d < L::NDB; ++d) vbase[d] = L::v_off(4 * g + tq, d * 2 + (tp >> 1)) + (tp & 1) * 8;
#pragma unroll
    for (int st = 0; st < NKF / 2; ++st) {
        const f32x4 p0 = s[2 * st], p1 = s[2 * st + 1];
        union { bf16x8 v; unsigned int u[4]; } pk;
        pk.u[0] = OP::pack2(p0[0], p0[1]);
        pk.u[1] = OP::pack2(p0[2], p0[3]);
        pk.u[2] = OP::pack2(p1[0], p1[1]);
        pk.u[3] = OP::pack2(p1[2], p1[3]);
#pragma unroll
        for (int d = 0; d < L::NDB; ++d) {
            const char* lo = v_lds + vbase[d] + st * 32 * L::ROW;   // keys 32 st + 4 g + tq: first half of the k-slots, 16-B chunk of columns d*16 + 4*tp
            const char* hi = lo + 16 * L::ROW;                      // second half = + 16 keys
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
    if (p.out8) {   // fp8 data path: 4 consecutive d -> one dword of e4m3
        if (q < N) {
            unsigned char* orow8 = p.out8 + (row0 + q) * p.ldo8 + h * ATT_DH + g * 4;
            const float sc = inv * p.scale8;
#pragma unroll
            for (int d = 0; d < L::NDB; ++d)
                *reinterpret_cast<unsigned int*>(orow8 + d * 16) = pack_fp8x4(o[d][0] * sc, o[d][1] * sc, o[d][2] * sc, o[d][3] * sc);
        }
    } else if (IVIT_ATT_WIDE_STORES && NKF < 38 && !p.lo_off) {   // (the 577-key instantiation has no registers left for the swap: 12 bytes of scratch)
        // 16-byte stores (round 4): lane groups g / g ^ 1 trade the quads of a column-block PAIR (d, d + 1) through v_permlane16_swap, as the GEMM
        // epilogues do - even groups end up with 8 consecutive columns of block d, odd groups of block d + 1.  The swap needs every lane (EXEC all
        // ones), so it runs in front of the q < N guard; rows past N compute on clamped garbage and store nothing.
        bf16_t* orow = p.out + (row0 + q) * p.ldo + h * ATT_DH;
#pragma unroll
        for (int d = 0; d + 1 < L::NDB; d += 2) {
            const auto lo = __builtin_amdgcn_permlane16_swap(OP::pack2(o[d][0] * inv, o[d][1] * inv), OP::pack2(o[d + 1][0] * inv, o[d + 1][1] * inv), false, false);
            const auto hi = __builtin_amdgcn_permlane16_swap(OP::pack2(o[d][2] * inv, o[d][3] * inv), OP::pack2(o[d + 1][2] * inv, o[d + 1][3] * inv), false, false);
            u32x4 pk = {lo[0], hi[0], lo[1], hi[1]};
            if (q < N) *reinterpret_cast<u32x4*>(orow + (d + (g & 1)) * 16 + (g & ~1) * 4) = pk;
        }
        if (L::NDB & 1) {   // head dim 80: the fifth column block keeps its 8-byte stores
            constexpr int d = L::NDB - 1;
            u32x2 pk2 = {OP::pack2(o[d][0] * inv, o[d][1] * inv), OP::pack2(o[d][2] * inv, o[d][3] * inv)};
            if (q < N) *reinterpret_cast<u32x2*>(orow + d * 16 + g * 4) = pk2;
        }
    } else if (q < N) {
        bf16_t* orow = p.out + (row0 + q) * p.ldo + h * ATT_DH + g * 4;
#pragma unroll
        for (int d = 0; d < L::NDB; ++d) {
            u32x2 pk2 = {OP::pack2(o[d][0] * inv, o[d][1] * inv), OP::pack2(o[d][2] * inv, o[d][3] * inv)};
            *reinterpret_cast<u32x2*>(orow + d * 16) = pk2;
            if (p.lo_off) {   // low parts for the split-operand out-projection (wave-uniform)
                float lo[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) lo[r] = o[d][r] * inv - OP::to_f32(OP::from_f32(o[d][r] * inv));
                u32x2 pl = {OP::pack2(lo[0], lo[1]), OP::pack2(lo[2], lo[3])};
                *reinterpret_cast<u32x2*>(orow + p.lo_off + d * 16) = pl;
            }
        }
    }
}

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
#ifdef IVIT_GEMM_ABLATIONS   // head-major q|k|v layouts of the study build (tools/fused_bench); run-time strides cost the 577-key instantiation its last registers
    const size_t hs = p.head_stride ? (size_t)p.head_stride : (size_t)ATT_DH, ws = p.which_stride ? (size_t)p.which_stride : (size_t)D;
#else
    constexpr int hs = ATT_DH;
    const int ws = D;
#endif
    // Issue order (round 4): Q fragments of the wave's first block (B operand: lane holds Q[q + fr][kk*32 + 8g .. +7]) and the K image;
    // wait for both; THEN the V image, and the barrier that publishes K.  The first block's Q.K^T and softmax run while the V half of the
    // staging burst is still arriving (all workgroups of a launch stage at once: 58 MB at ViT-B/16 B = 64 before anyone computes);
    // att_block drains and publishes V before its first P.V (wait_v).  The Q registers are "used" by an empty asm right behind the wait:
    // hipcc waits for a pending register load at its first use with vmcnt(0) when it cannot count what was issued since (the staging
    // loops) - left to the first MFMA that would also have waited for V and for the next block's Q prefetch.
    const bf16x8 zero_frag = {0, 0, 0, 0, 0, 0, 0, 0};
    bf16x8 qf[L::KSTEPS];
    auto load_q = [&](int qbase, bf16x8 (&q)[L::KSTEPS]) {
        const int qrow = min(qbase + fr, N - 1);
#pragma unroll
        for (int kk = 0; kk < L::KSTEPS; ++kk)
            q[kk] = (kk * 32 + g * 8 + 8 <= ATT_DH)
                        ? *reinterpret_cast<const bf16x8*>(qkv + (row0 + qrow) * ld + h * hs + kk * 32 + g * 8) : zero_frag;
    };
    load_q(q0, qf);
    att_stage<L>(k_lds, qkv + row0 * ld + h * hs + ws, ld, N, L::K_SWZ, wave, nwaves, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's K pieces and Q fragments have landed
#pragma unroll
    for (int kk = 0; kk < L::KSTEPS; ++kk) asm volatile("" ::"v"(qf[kk]));
    att_stage<L>(v_lds, qkv + row0 * ld + h * hs + 2 * ws, ld, N, L::V_SWZ, wave, nwaves, lane);
    constexpr bool split_wait = !PROBS;
    if (!split_wait) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the zero rows this wave wrote
    __builtin_amdgcn_s_barrier();                          // publishes K (and V when it was waited for); a raw barrier: __syncthreads() would drain vmcnt
    const float cexp = p.scale * 1.44269504088896340736f;  // exp(x*scale) = exp2(x*scale*log2 e)

    // The wave's 16-query blocks (wave, wave + nwaves, ...) run one after the other, so only one block's
    // scores (NKF x 4 registers) are live at a time: < 128 VGPRs at 197 keys, i.e. two workgroups per
    // CU and one's staging overlaps the other's math.  (K / V fragments are re-read per block: LDS has room.)
    // (round 4: the NEXT block's Q fragments are loaded before the current block is computed - the plain load at the top of a block
    // exposed a full memory latency per block, 4-5 times per wave at 577 keys)
    bf16x8 qn[L::KSTEPS];
#pragma unroll 1
    for (int blk = wave; blk < nblocks; blk += nwaves) {
        const int qbase = blk * 16;
        asm volatile("" ::: "memory");   // keeps the (block-invariant) K / V fragment reads inside the loop: hoisted, they cost 100+ VGPRs
        if (blk != wave) {
#pragma unroll
            for (int kk = 0; kk < L::KSTEPS; ++kk) qf[kk] = qn[kk];
        }
        if (blk + nwaves < nblocks) load_q(qbase + nwaves * 16, qn);
        att_block<DH, NKF, ODD, PROBS, OP>(p, k_lds, v_lds, qf, qbase, b, h, row0, N, fr, g, cexp, split_wait && blk == wave);
    }
}

#ifdef IVIT_GEMM_ABLATIONS   // study kernel (round 3): correct, 8 % slower than the one-head kernel (25.5 against 23.6 us in a forward, tools/fused_bench
// with IVIT_ATTN_PIPE=1); kept out of libivit.so
// Pipelined form for short sequences (every wave owns exactly ONE 16-query block: tokens <= 16 x waves <= 256): a workgroup handles ITEMS
// consecutive heads of one image, K / V double buffered in LDS.  The one-head kernel puts 768 workgroups on the chip at once (ViT-B/16,
// B = 64): all of them stage 58 MB first, then all compute, then all store - three phases of a 24 us launch with the memory system idle
// in the middle one.  Here the staging of head i + 1 (and its Q fragments) is in flight under the math of head i.
template <int DH, int NKF, bool ODD, class OP>
__global__ __launch_bounds__(1024) void ivit_attention_pipe(AttnParams p, int items) {
    using L = AttLayout<DH, NKF, ODD>;
    constexpr int ATT_DH = DH;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int fr = lane & 15, g = lane >> 4;
    const int b = blockIdx.z, h0 = blockIdx.y * items;
    const int N = p.tokens;
    const int D = p.heads * ATT_DH;
    const size_t row0 = (size_t)b * N;
    const bf16_t* qkv = p.qkv;
    const int ld = p.ldqkv;
    const int nwaves = blockDim.x >> 6;
    const int qbase = wave * 16;                          // this wave's block (waves past the last block only help staging)
    const bool has_block = qbase < N;
    const float cexp = p.scale * 1.44269504088896340736f;
    const bf16x8 zero_frag = {0, 0, 0, 0, 0, 0, 0, 0};
    const int nh = min(items, p.heads - h0);
    const size_t hs = p.head_stride ? (size_t)p.head_stride : (size_t)ATT_DH, ws = p.which_stride ? (size_t)p.which_stride : (size_t)D;

    auto stage = [&](int h, int buf) {
        char* base = smem + buf * L::LDS_BYTES;
        att_stage<L>(base, qkv + row0 * ld + h * hs + ws, ld, N, L::K_SWZ, wave, nwaves, lane);
        att_stage<L>(base + L::K_BYTES, qkv + row0 * ld + h * hs + 2 * ws, ld, N, L::V_SWZ, wave, nwaves, lane);
    };
    auto load_q = [&](int h, bf16x8 (&q)[L::KSTEPS]) {
        const int qrow = min(qbase + fr, N - 1);
#pragma unroll
        for (int kk = 0; kk < L::KSTEPS; ++kk)
            q[kk] = (kk * 32 + g * 8 + 8 <= ATT_DH)
                        ? *reinterpret_cast<const bf16x8*>(qkv + (row0 + qrow) * ld + h * hs + kk * 32 + g * 8) : zero_frag;
    };

    bf16x8 qf[L::KSTEPS], qn[L::KSTEPS];
    stage(h0, 0);
    load_q(h0, qn);
    for (int it = 0; it < nh; ++it) {
        // head it: its K / V image and Q fragments have landed (every vector-memory operation of this wave is behind us); the barrier
        // publishes everyone's pieces and says that every wave is done with the OTHER buffer (head it - 1)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int kk = 0; kk < L::KSTEPS; ++kk) qf[kk] = qn[kk];
        if (it + 1 < nh) {
            // register loads FIRST, then the DMA: hipcc waits for everything in flight at the first use of a register load, and that
            // use is the copy above, one iteration later, behind the wait that this loop needs anyway
            load_q(h0 + it + 1, qn);
            stage(h0 + it + 1, (it + 1) & 1);
        }
        asm volatile("" ::: "memory");
        if (has_block) {
            const char* k_lds = smem + (it & 1) * L::LDS_BYTES;
            att_block<DH, NKF, ODD, false, OP>(p, k_lds, k_lds + L::K_BYTES, qf, qbase, b, h0 + it, row0, N, fr, g, cexp);
        }
    }
}

#endif


// ---------------------------------------------------------------- long sequences (round 4): 32 queries per wave, key tiles of 32, 32x32x16 MFMA
// At 577 keys the one-pass kernel above keeps 152 score registers per lane live (256 VGPRs, two waves per SIMD): LDS reads, softmax VALU and
// MFMA of a wave run back to back, 44 us per workgroup.  ivit_attention_q32 keeps the whole head resident in LDS too (K and V staged once by
// LDS-DMA), but a wave owns a block of 32 queries and walks the keys in tiles of 32 with v_mfma_f32_32x32x16 (half the LDS bytes and half
// the MFMA issue slots per FLOP of the 16x16x32 form), one score tile live at a time:
//   S^T tile = K_t . Q^T   A = K rows (ds_read_b128: lane (key = l & 31, hi = l >> 5) reads d = 16 kk + 8 hi .. +7), B = Q fragments in registers
//              -> lane (q = l & 31, hi) holds S[q][key = 32 t + (r & 3) + 8 (r >> 2) + 4 hi], r = 0..15
//   O^T += V_t^T . P^T     B = P straight from those registers (k-slot 8 hi + j of step s = register 8 s + j), A = V^T by ds_read_b64_tr_b16
//              -> lane holds O[q][d = 32 dblk + (r & 3) + 8 (r >> 2) + 4 hi]
// What was measured on the way (tools/attn_bench, tools/simd_share_probe; profiles/r04_attention_long.txt): a software-pipelined form with
// two waves per SIMD (QK^T of tile t + 1 issued under the softmax of tile t, hand-placed MFMA / VALU interleave) ran at the same 300 - 320 us
// per ViT-L launch as this one, which is the simpler code.  The bound is the SIMD's one instruction stream: per 32 x 32 score tile the softmax's
// 48 vector instructions cost ~240 cycles (v_exp_f32 8.2 each), an MFMA ~15 - 20 cycles of that stream besides its 32 in the pipe, an LDS read ~8.
struct AttLayout32 {   // head dim 64: 128-B rows of 8 chunks; K chunk slot = ch ^ ((key >> 1) & 7) (ds_read_b128 by 32 keys x one chunk: conflict free),
    static constexpr int ROW = 128, TILE = 32 * ROW;   // V chunk slot = ch ^ (((key >> 1) & 1) << 2) (tr reads of 4 keys x 32 d per half wave: conflict free)
    static constexpr int K_SWZ = 0, V_SWZ = 1;
    __device__ static int swz(int which, int key) { return which == K_SWZ ? ((key >> 1) & 7) : (((key >> 1) & 1) << 2); }
};

__device__ __forceinline__ void att_stage32(char* lds, const bf16_t* src0, int ld, int N, int rows, int which, int wave, int nwaves, int lane) {
    const int instr = rows >> 3;   // one DMA instruction = 64 chunk slots = 8 rows
    for (int i = wave; i < instr; i += nwaves) {
        const int slot = i * 64 + lane;
        const int key = slot >> 3, cl = slot & 7;
        if (key < N) {
            const int ch = cl ^ AttLayout32::swz(which, key);
            __builtin_amdgcn_global_load_lds((const IVIT_GLOBAL void*)(src0 + (size_t)key * ld + ch * 8), (IVIT_LDS void*)(lds + i * 1024), 16, 0, 0);
        } else {
            *reinterpret_cast<u32x4*>(lds + slot * 16) = u32x4{0u, 0u, 0u, 0u};   // 0 * garbage must not be NaN in P.V
        }
    }
}

__device__ __forceinline__ bf16x4 lds_read_tr16(IVIT_LDS const char* lds_addr) {
    typedef __attribute__((ext_vector_type(4))) short s16x4;
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((IVIT_LDS s16x4*)lds_addr);
    return __builtin_bit_cast(bf16x4, v);
}

// both lanes of a pair (l, l ^ 32) get the value of the low lane in `lo` and of the high lane in `hi`
__device__ __forceinline__ void pair_exchange(float x, float& lo, float& hi) {
    // (hipcc 7.2 folds max / add of the two results of one swap of the SAME value into an operation on result 0 alone; the empty asm on the
    // results keeps them apart)
    const auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned int, x), __builtin_bit_cast(unsigned int, x), false, false);
    unsigned int a = r[0], b = r[1];
    asm volatile("" : "+v"(a), "+v"(b));
    lo = __builtin_bit_cast(float, a);
    hi = __builtin_bit_cast(float, b);
}

// No software pipeline across key tiles: a wave needs < 128 registers and a workgroup runs up to 16 waves, four per SIMD (one wave alone
// issues a vector instruction every >= 5 cycles).
//  * Reference exponent: m = ceil(c x the row's maximum over the FIRST key tile a wave sees); numerators 2^(c s - m) of later tiles may
//    exceed 1 (a 16-bit float keeps its relative precision at any scale; sums are f32), so the tile loop has no maximum, no branch and no
//    rescale.  A numerator that left the 16-bit range shows up as a non-finite row sum; that block is then redone with the true maximum of
//    the rows concerned (never seen on real activations; the other rows repeat their first evaluation bit for bit).
//  * Balance: nt query blocks on W waves leave L = nt mod W blocks over (577 tokens: 19 blocks, 16 waves, 3 over).  One wave each for them
//    would run alone on its SIMD for a whole block time, at the rate of a lone wave.  Instead every leftover block is shared by W / L waves,
//    each taking a slice of the key tiles with its own reference exponent; the partial (O, sum, m) go through LDS - the K / V images are
//    dead by then - and the first wave of the group adds them up scaled by exact powers of two.
template <class OP, int MAXW>
__global__ __launch_bounds__(MAXW * 64) void ivit_attention_q32(AttnParams p, int nt) {
    using L = AttLayout32;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int rows = nt * 32;
    char* k_lds = smem;
    char* v_lds = smem + rows * L::ROW;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwaves = blockDim.x >> 6;
    const int ql = lane & 31, hi = lane >> 5;
    const int b = blockIdx.z, h = blockIdx.y;
    const int N = p.tokens;
    const int D = p.heads * 64;
    const size_t row0 = (size_t)b * N;
    const bf16_t* qkv = p.qkv;
    const int ld = p.ldqkv;

    att_stage32(k_lds, qkv + row0 * ld + h * 64 + D, ld, N, rows, L::K_SWZ, wave, nwaves, lane);
    att_stage32(v_lds, qkv + row0 * ld + h * 64 + 2 * D, ld, N, rows, L::V_SWZ, wave, nwaves, lane);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const float cexp = p.scale * 1.44269504088896340736f;

    // per-lane LDS addresses as address-space-3 pointers (tile offsets become instruction immediates), recomputed per slice (registers) and
    // opaque to the compiler, which otherwise keeps them relative to the dynamic-LDS base and re-adds that base - a literal 0 - at every read
    typedef IVIT_LDS const char* lds_ptr;
    lds_ptr kp[4], vp[2];
    auto set_tile = [&](int t) {
        lds_ptr lds0 = (lds_ptr)smem + t * L::TILE;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) kp[kk] = lds0 + ql * L::ROW + (((2 * kk + hi) ^ ((ql >> 1) & 7)) << 4);
        const int gg = lane >> 4, q4 = (lane & 15) >> 2, pp = lane & 3;   // transposed read: lane 4 q4 + pp of a 16-lane group supplies &V[key0 + q4][d0 + 4 pp]
#pragma unroll
        for (int dblk = 0; dblk < 2; ++dblk) {
            const int c = 4 * dblk + 2 * (gg & 1) + (pp >> 1);
            vp[dblk] = lds0 + rows * L::ROW + (4 * hi + q4) * L::ROW + ((c ^ ((q4 >> 1) << 2)) << 4) + (pp & 1) * 8;
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) asm volatile("" : "+v"(kp[kk]));
        asm volatile("" : "+v"(vp[0]), "+v"(vp[1]));
    };

    bf16x8 qf[4];
    auto load_q = [&](int qbase) {   // B operand of S^T: lane holds Q[qbase + ql][16 kk + 8 hi .. +7]; rows past N repeat the last one (never stored)
        const int qrow = min(qbase + ql, N - 1);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) qf[kk] = *reinterpret_cast<const bf16x8*>(qkv + (row0 + qrow) * ld + h * 64 + kk * 16 + hi * 8);
    };
    // S^T of the key tile at kp + off.  No masking: K rows past N are zeros in LDS, so such a key scores exactly 0, its numerator is the
    // same 2^-m for every one of them, and V's rows past N are zeros too - O is right as it is and the row sum is put right once per block.
    auto qk = [&](f32x16& a, int off) {
        bf16x8 kf[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) kf[kk] = *reinterpret_cast<IVIT_LDS const bf16x8*>(kp[kk] + off);
#pragma unroll
        for (int r = 0; r < 16; ++r) a[r] = 0.f;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) a = OP::mfma32(kf[kk], qf[kk], a);
    };
    auto row_max = [&](const f32x16& a) {   // over the 32 keys of a tile: this lane's 16 and its partner's
        float ta = fmaxf(fmaxf(a[0], a[1]), a[2]), tb = fmaxf(fmaxf(a[3], a[4]), a[5]);
#pragma unroll
        for (int r = 6; r < 14; r += 4) { ta = fmaxf(fmaxf(ta, a[r]), a[r + 1]); tb = fmaxf(fmaxf(tb, a[r + 2]), a[r + 3]); }
        float xa, xb;
        pair_exchange(fmaxf(fmaxf(ta, tb), fmaxf(a[14], a[15])), xa, xb);
        return fmaxf(xa, xb);
    };
    // Keys past N (the last key tile: zero K rows, so they would score exactly 0).  They are MASKED - their scores become -inf, their numerators exactly 0 - in
    // that one tile.  (Round 4 let them through and subtracted their npad x rn16(2^-m) from the row sum afterwards: for a row whose real logits are all far
    // below 0 the reference exponent m is very negative, those terms swamp the f32 sum and the subtraction cancels - 4e-4 at logits of -20, 41 % at -28 -
    // and 2^-m leaves the f16 range altogether; a softmax must not depend on where 0 lies.)  This lane's 16 keys of a tile: (r & 3) + 8 (r >> 2) + 4 hi.
    const int pad_lim = N - 32 * (nt - 1) - 4 * hi;
    const bool pad_any = (N & 31) != 0;
    auto mask_last = [&](f32x16& a) {
#pragma unroll
        for (int r = 0; r < 16; ++r) a[r] = ((r & 3) + 8 * (r >> 2) >= pad_lim) ? -INFINITY : a[r];
    };
    f32x16 o0, o1;
    float m_row, sum;   // sum: this lane's half of the row sum
    // key tiles [t0, t1) of the query block in qf -> o0, o1, sum, m_row
    auto tiles = [&](int t0, int t1) {
        f32x16 s;
        auto body = [&](int off, auto mask_tag) {   // the tile at vp + off (its scores are in s; mask_tag: it is the tile with the padded keys), then the scores of the tile after it
            if (decltype(mask_tag)::value) mask_last(s);
            union { bf16x8 v; bf16x4 h2[2]; } vf[2][2];
#pragma unroll
            for (int st = 0; st < 2; ++st)
#pragma unroll
                for (int dblk = 0; dblk < 2; ++dblk) {
                    vf[st][dblk].h2[0] = lds_read_tr16(vp[dblk] + off + (16 * st) * L::ROW);
                    vf[st][dblk].h2[1] = lds_read_tr16(vp[dblk] + off + (16 * st + 8) * L::ROW);
                }
            const float nm = -m_row;
            union { bf16x8 v; unsigned int u[4]; } pk[2];
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const float e0 = __builtin_amdgcn_exp2f(fmaf(s[r], cexp, nm));
                const float e1 = __builtin_amdgcn_exp2f(fmaf(s[r + 1], cexp, nm));
                const unsigned int pr = OP::pack2(e0, e1);
                sum = OP::add_pair(pr, sum);   // the row sum is the sum of the ROUNDED numerators: O / sum is a convex combination of V rows
                pk[r >> 3].u[(r & 7) >> 1] = pr;
            }
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                o0 = OP::mfma32(vf[st][0].v, pk[st].v, o0);
                o1 = OP::mfma32(vf[st][1].v, pk[st].v, o1);
            }
            qk(s, off + L::TILE);   // (behind the last tile of the slice: one tile further in LDS - at worst the head of V's image - and never used)
        };
#pragma unroll 1
        for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
            sum = 0.f;
            set_tile(t0);
            qk(s, 0);
            const bool tail = pad_any && t1 == nt;   // the slice ends on the tile with the padded keys: that tile runs the masking form of the body (wave-uniform)
            if (pass == 0) {
                if (tail && t1 - t0 == 1) mask_last(s);   // (idempotent: the body masks it again)
                m_row = ceilf(row_max(s) * cexp);
            }
            int rem = t1 - t0 - (tail ? 1 : 0);
            if (MAXW <= 12) {   // (two tiles a turn where the register budget allows: 3 waves per SIMD, 170 registers)
#pragma unroll 1
                for (; rem >= 2; rem -= 2) {
                    body(0, std::false_type{});
                    body(L::TILE, std::false_type{});
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) kp[kk] += 2 * L::TILE;
                    vp[0] += 2 * L::TILE; vp[1] += 2 * L::TILE;
                }
            }
#pragma unroll 1
            for (; rem >= 1; --rem) {
                body(0, std::false_type{});
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) kp[kk] += L::TILE;
                vp[0] += L::TILE; vp[1] += L::TILE;
            }
            if (tail) {
                body(0, std::true_type{});
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) kp[kk] += L::TILE;
                vp[0] += L::TILE; vp[1] += L::TILE;
            }
            if (pass == 1) break;
            float sa, sb;
            pair_exchange(sum, sa, sb);
            const bool bad = !(fabsf(sa + sb) < INFINITY);   // a numerator left the 16-bit range (or the input was not finite)
            if (__builtin_amdgcn_ballot_w64(bad) == 0) break;
            float mx = -INFINITY;   // the true maximum of the slice over its REAL keys, for the rows that need it
            set_tile(t0);
#pragma unroll 1
            for (int t = t0; t < t1; ++t) {
                qk(s, 0);
                if (tail && t == t1 - 1) mask_last(s);
                mx = fmaxf(mx, row_max(s));
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) kp[kk] += L::TILE;
            }
            if (bad) m_row = ceilf(mx * cexp);
        }
    };
    // O = (o0 | o1) * inv of the query block at qbase -> memory
    auto store_block = [&](int qbase, float inv) {
        const int q = qbase + ql;
        if (p.out8) {
            if (q < N) {
                unsigned char* orow8 = p.out8 + (row0 + q) * p.ldo8 + h * 64 + hi * 4;
                const float sc8 = inv * p.scale8;
#pragma unroll
                for (int dblk = 0; dblk < 2; ++dblk)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const f32x16& o = dblk ? o1 : o0;
                        *reinterpret_cast<unsigned int*>(orow8 + dblk * 32 + j * 8) = pack_fp8x4(o[4 * j] * sc8, o[4 * j + 1] * sc8, o[4 * j + 2] * sc8, o[4 * j + 3] * sc8);
                    }
            }
        } else if (p.lo_off) {   // high and low parts for the split-operand out-projection
            if (q < N) {
                bf16_t* orow = p.out + (row0 + q) * p.ldo + h * 64 + hi * 4;
#pragma unroll
                for (int dblk = 0; dblk < 2; ++dblk)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const f32x16& o = dblk ? o1 : o0;
                        float v[4], lo[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) { v[r] = o[4 * j + r] * inv; lo[r] = v[r] - OP::to_f32(OP::from_f32(v[r])); }
                        *reinterpret_cast<u32x2*>(orow + dblk * 32 + j * 8) = u32x2{OP::pack2(v[0], v[1]), OP::pack2(v[2], v[3])};
                        *reinterpret_cast<u32x2*>(orow + p.lo_off + dblk * 32 + j * 8) = u32x2{OP::pack2(lo[0], lo[1]), OP::pack2(lo[2], lo[3])};
                    }
            }
        } else {
            // 16-byte stores: the lanes of a pair hold d = 8 j + 0..3 (hi = 0) and 8 j + 4..7 (hi = 1); v_permlane32_swap on the groups j, j + 1
            // leaves the low lane with all 8 columns of group j and the high lane with those of group j + 1.  (EXEC all ones: in front of the guard)
            bf16_t* orow = p.out + (row0 + q) * p.ldo + h * 64 + hi * 8;
#pragma unroll
            for (int dblk = 0; dblk < 2; ++dblk)
#pragma unroll
                for (int j = 0; j < 4; j += 2) {
                    const f32x16& o = dblk ? o1 : o0;
                    const auto ra = __builtin_amdgcn_permlane32_swap(OP::pack2(o[4 * j] * inv, o[4 * j + 1] * inv), OP::pack2(o[4 * j + 4] * inv, o[4 * j + 5] * inv), false, false);
                    const auto rb = __builtin_amdgcn_permlane32_swap(OP::pack2(o[4 * j + 2] * inv, o[4 * j + 3] * inv), OP::pack2(o[4 * j + 6] * inv, o[4 * j + 7] * inv), false, false);
                    const u32x4 pkk = {ra[0], rb[0], ra[1], rb[1]};
                    if (q < N) *reinterpret_cast<u32x4*>(orow + dblk * 32 + j * 8) = pkk;
                }
        }
    };

    // ---- whole blocks: wave w takes blocks w, w + W, ... below full * W
    const int full = nt / nwaves, left = nt - full * nwaves;
#pragma unroll 1
    for (int k = 0; k < full; ++k) {
        const int qbase = (wave + k * nwaves) * 32;
        load_q(qbase);
        tiles(0, nt);
        float sa, sb;
        pair_exchange(sum, sa, sb);
        store_block(qbase, 1.0f / (sa + sb));
    }
    if (left == 0) return;   // (wave-uniform for the whole workgroup: nobody waits at the barriers below)

    // ---- leftover blocks: block full * W + j is shared by the waves [ceil(j W / L), ceil((j + 1) W / L)), a slice of the key tiles each
    const int j = wave * left / nwaves;
    const int w0 = (j * nwaves + left - 1) / left, w1 = ((j + 1) * nwaves + left - 1) / left;
    const int g = w1 - w0, pos = wave - w0;
    const int t0 = pos * nt / g, t1 = (pos + 1) * nt / g;   // (g <= W <= nt: no slice is empty)
    const int qbase = (full * nwaves + j) * 32;
    load_q(qbase);
    tiles(t0, t1);
    __builtin_amdgcn_s_barrier();   // every wave is done with the K / V images
    // partial of wave w: 34 rows of 64 floats at w * PART (lane-wise: the combining wave reads its own register layout back)
    constexpr int PART = 34 * 256;
    float* mine = reinterpret_cast<float*>(smem + wave * PART) + lane;
#pragma unroll
    for (int r = 0; r < 16; ++r) { mine[r * 64] = o0[r]; mine[(16 + r) * 64] = o1[r]; }
    mine[32 * 64] = sum;
    mine[33 * 64] = m_row;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (pos != 0) return;
    float mtop = m_row;
#pragma unroll 1
    for (int i = 1; i < g; ++i) mtop = fmaxf(mtop, (reinterpret_cast<const float*>(smem + (w0 + i) * PART) + lane)[33 * 64]);
    {
        const float f = __builtin_amdgcn_exp2f(m_row - mtop);   // integers: an exact power of two
#pragma unroll
        for (int r = 0; r < 16; ++r) { o0[r] *= f; o1[r] *= f; }
        sum *= f;
    }
#pragma unroll 1
    for (int i = 1; i < g; ++i) {
        const float* other = reinterpret_cast<const float*>(smem + (w0 + i) * PART) + lane;
        const float f = __builtin_amdgcn_exp2f(other[33 * 64] - mtop);
#pragma unroll
        for (int r = 0; r < 16; ++r) { o0[r] = fmaf(other[r * 64], f, o0[r]); o1[r] = fmaf(other[(16 + r) * 64], f, o1[r]); }
        sum = fmaf(other[32 * 64], f, sum);
    }
    float sa, sb;
    pair_exchange(sum, sa, sb);
    store_block(qbase, 1.0f / (sa + sb));
}

#ifdef IVIT_GEMM_ABLATIONS   // study (tools/attn_bench, IVIT_ATT32=4): two query blocks (64 queries) per wave sharing every K / V fragment read, 8 waves;
// whole units only (tokens a multiple of 64, at most 8 units) - a rate comparison against ivit_attention_q32, not a product kernel
template <class OP>
__global__ __launch_bounds__(512) void ivit_attention_q64_study(AttnParams p, int nt) {
    using L = AttLayout32;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int rows = nt * 32;
    char* k_lds = smem;
    char* v_lds = smem + rows * L::ROW;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwaves = blockDim.x >> 6;
    const int ql = lane & 31, hi = lane >> 5;
    const int b = blockIdx.z, h = blockIdx.y;
    const int N = p.tokens;
    const int D = p.heads * 64;
    const size_t row0 = (size_t)b * N;
    const bf16_t* qkv = p.qkv;
    const int ld = p.ldqkv;
    att_stage32(k_lds, qkv + row0 * ld + h * 64 + D, ld, N, rows, L::K_SWZ, wave, nwaves, lane);
    att_stage32(v_lds, qkv + row0 * ld + h * 64 + 2 * D, ld, N, rows, L::V_SWZ, wave, nwaves, lane);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const float cexp = p.scale * 1.44269504088896340736f;
    typedef IVIT_LDS const char* lds_ptr;
    lds_ptr kp[4], vp[2];
    {
        lds_ptr lds0 = (lds_ptr)smem;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) kp[kk] = lds0 + ql * L::ROW + (((2 * kk + hi) ^ ((ql >> 1) & 7)) << 4);
        const int gg = lane >> 4, q4 = (lane & 15) >> 2, pp = lane & 3;
#pragma unroll
        for (int dblk = 0; dblk < 2; ++dblk) {
            const int c = 4 * dblk + 2 * (gg & 1) + (pp >> 1);
            vp[dblk] = lds0 + rows * L::ROW + (4 * hi + q4) * L::ROW + ((c ^ ((q4 >> 1) << 2)) << 4) + (pp & 1) * 8;
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) asm volatile("" : "+v"(kp[kk]));
        asm volatile("" : "+v"(vp[0]), "+v"(vp[1]));
    }
    const int qbase = wave * 64;
    bf16x8 qf[2][4];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int qrow = min(qbase + 32 * u + ql, N - 1);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) qf[u][kk] = *reinterpret_cast<const bf16x8*>(qkv + (row0 + qrow) * ld + h * 64 + kk * 16 + hi * 8);
    }
    f32x16 o0[2], o1[2], s[2];
    float m_row[2], sum[2] = {0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) { o0[u][r] = 0.f; o1[u][r] = 0.f; }
    auto qk = [&](int off) {
        bf16x8 kf[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) kf[kk] = *reinterpret_cast<IVIT_LDS const bf16x8*>(kp[kk] + off);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[u][r] = 0.f;
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            s[0] = OP::mfma32(kf[kk], qf[0][kk], s[0]);
            s[1] = OP::mfma32(kf[kk], qf[1][kk], s[1]);
        }
    };
    qk(0);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        float ta = fmaxf(fmaxf(s[u][0], s[u][1]), s[u][2]);
#pragma unroll
        for (int r = 3; r < 15; r += 2) ta = fmaxf(fmaxf(ta, s[u][r]), s[u][r + 1]);
        float xa, xb;
        pair_exchange(fmaxf(ta, s[u][15]), xa, xb);
        m_row[u] = ceilf(fmaxf(xa, xb) * cexp);
    }
#pragma unroll 1
    for (int t = 0; t < nt; ++t) {
        union { bf16x8 v; bf16x4 h2[2]; } vf[2][2];
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
            for (int dblk = 0; dblk < 2; ++dblk) {
                vf[st][dblk].h2[0] = lds_read_tr16(vp[dblk] + (16 * st) * L::ROW);
                vf[st][dblk].h2[1] = lds_read_tr16(vp[dblk] + (16 * st + 8) * L::ROW);
            }
        union { bf16x8 v; unsigned int u[4]; } pk[2][2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const float nm = -m_row[u];
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const float e0 = __builtin_amdgcn_exp2f(fmaf(s[u][r], cexp, nm));
                const float e1 = __builtin_amdgcn_exp2f(fmaf(s[u][r + 1], cexp, nm));
                const unsigned int pr = OP::pack2(e0, e1);
                sum[u] = OP::add_pair(pr, sum[u]);
                pk[u][r >> 3].u[(r & 7) >> 1] = pr;
            }
        }
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                o0[u] = OP::mfma32(vf[st][0].v, pk[u][st].v, o0[u]);
                o1[u] = OP::mfma32(vf[st][1].v, pk[u][st].v, o1[u]);
            }
        qk(L::TILE);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) kp[kk] += L::TILE;
        vp[0] += L::TILE; vp[1] += L::TILE;
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        float sa, sb;
        pair_exchange(sum[u], sa, sb);
        const float inv = 1.0f / (sa + sb);
        const int q = qbase + 32 * u + ql;
        bf16_t* orow = p.out + (row0 + q) * p.ldo + h * 64 + hi * 8;
#pragma unroll
        for (int dblk = 0; dblk < 2; ++dblk)
#pragma unroll
            for (int j = 0; j < 4; j += 2) {
                const f32x16& o = dblk ? o1[u] : o0[u];
                const auto ra = __builtin_amdgcn_permlane32_swap(OP::pack2(o[4 * j] * inv, o[4 * j + 1] * inv), OP::pack2(o[4 * j + 4] * inv, o[4 * j + 5] * inv), false, false);
                const auto rb = __builtin_amdgcn_permlane32_swap(OP::pack2(o[4 * j + 2] * inv, o[4 * j + 3] * inv), OP::pack2(o[4 * j + 6] * inv, o[4 * j + 7] * inv), false, false);
                const u32x4 pkk = {ra[0], rb[0], ra[1], rb[1]};
                if (q < N) *reinterpret_cast<u32x4*>(orow + dblk * 32 + j * 8) = pkk;
            }
    }
}
#endif

template <class OP, int MAXW>
static hipError_t launch_q32_op(const AttnParams& p, hipStream_t stream) {
    const int nt = ceil_div(p.tokens, 32);
    const int lds = 2 * nt * AttLayout32::TILE;
    hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(ivit_attention_q32<OP, MAXW>), 2 * 19 * AttLayout32::TILE);
    if (e != hipSuccess) return e;
    int waves = std::min(MAXW, nt);
#ifdef IVIT_GEMM_ABLATIONS
    { static const int w = [] { const char* v = getenv("IVIT_ATT_WAVES"); return v ? atoi(v) : 0; }(); if (w > 0) waves = std::min(std::min(w, MAXW), nt); }   // study knob
#endif
    hipLaunchKernelGGL((ivit_attention_q32<OP, MAXW>), dim3(1, p.heads, p.batch), dim3(waves * 64), lds, stream, p, nt);
    return hipGetLastError();
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
#ifdef IVIT_GEMM_ABLATIONS
    // pipelined multi-head form: every wave owns one query block, two K / V images fit LDS, several heads per image
    if (!PROBS && blocks <= 16 && 2 * L::LDS_BYTES <= 160 * 1024 && p.heads >= 2) {
        static const int pipe = [] { const char* v = getenv("IVIT_ATTN_PIPE"); return v ? atoi(v) : 0; }();   // IVIT_ATTN_PIPE=1 selects it
        if (pipe) {
            const int items = p.heads % 3 == 0 ? 3 : (p.heads % 4 == 0 ? 4 : 2);
            auto kernel = ivit_attention_pipe<DH, NKF, ODD, OP>;
            e = ensure_dynamic_lds(reinterpret_cast<const void*>(kernel), 2 * L::LDS_BYTES);
            if (e != hipSuccess) return e;
            dim3 pgrid(1, ceil_div(p.heads, items), p.batch);
            hipLaunchKernelGGL(kernel, pgrid, dim3(blocks * 64), 2 * L::LDS_BYTES, stream, p, items);
            return hipGetLastError();
        }
    }
#endif
    int waves = std::min(8, blocks);
#ifdef IVIT_GEMM_ABLATIONS
    { static const int w = [] { const char* v = getenv("IVIT_ATT_WAVES"); return v ? atoi(v) : 0; }(); if (w > 0) waves = std::min(w, blocks); }   // study knob
#endif
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

#ifdef IVIT_GEMM_ABLATIONS
static int att32_force() { static const int force = [] { const char* v = getenv("IVIT_ATT32"); return v ? atoi(v) : -1; }(); return force; }   // study knob: 0 never, 1 whenever head dim 64, 2 the same with 12 waves, 4 the 64-query study kernel (tokens a multiple of 64, <= 512)
#endif
// Which kernel: the 32-query tiled form from 289 tokens on at head dim 64 (measured at B x H = 2048 heads, tools/attn_bench: 257 tokens 88.6 us
// against 90.5 for the one-pass form, 325 tokens 141 against 180, 417 tokens 212 against 268, 577 tokens 332 against 376; 197 tokens 51.6 against
// 47.0).  The inspection variant (probabilities written out) stays on the one-pass form.
static bool use_q32(const AttnParams& p) {
    bool q32 = p.head_dim == 64 && !p.probs && p.tokens > 288;
#ifdef IVIT_GEMM_ABLATIONS
    if (att32_force() >= 0) q32 = att32_force() && p.head_dim == 64 && !p.probs;
#endif
    return q32;
}
const char* attention_kernel_name(const AttnParams& p) { return use_q32(p) ? "ivit_attention_q32" : "ivit_attention_bf16"; }

hipError_t launch_attention(const AttnParams& p, hipStream_t stream) {
    if (!attention_supported(p.tokens, p.head_dim)) return hipErrorInvalidValue;
    if ((p.ldqkv % 8) || (!p.probs && !p.out8 && (p.ldo % 4)) || (p.out8 && (p.ldo8 % 4))) return hipErrorInvalidValue;
    // the 16-bit output goes out in 16-byte stores (the one-pass kernel below 38 key fragments, ivit_attention_q32's store_block): rows and base 16-byte aligned
    if (!p.probs && !p.out8 && !p.lo_off && ((p.ldo % 8) || (reinterpret_cast<uintptr_t>(p.out) & 15))) return hipErrorInvalidValue;
    const int nkf = round_up(ceil_div(p.tokens, 16), 2);
    if (use_q32(p)) {
#ifdef IVIT_GEMM_ABLATIONS
        if (att32_force() == 4 && p.tokens % 64 == 0 && p.tokens <= 512 && !p.f16 && !p.out8 && !p.lo_off) {
            const int nt = p.tokens / 32;
            hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(ivit_attention_q64_study<OpBf16>), 2 * 19 * AttLayout32::TILE);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL((ivit_attention_q64_study<OpBf16>), dim3(1, p.heads, p.batch), dim3((nt / 2) * 64), 2 * nt * AttLayout32::TILE, stream, p, nt);
            return hipGetLastError();
        }
        if (att32_force() == 2) return p.f16 ? launch_q32_op<OpF16, 12>(p, stream) : launch_q32_op<OpBf16, 12>(p, stream);
#endif
        return p.f16 ? launch_q32_op<OpF16, 16>(p, stream) : launch_q32_op<OpBf16, 16>(p, stream);
    }
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


#ifdef IVIT_GEMM_ABLATIONS   // microbenchmark builds only (tools/fused_bench): fused QKV projection + attention, measured in round 2, loses
#include "study/fused_qkv_attention.inc"
#endif

}  // namespace ivit
