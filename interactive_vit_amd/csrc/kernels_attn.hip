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


#ifdef IVIT_GEMM_ABLATIONS   // microbenchmark builds only (tools/fused_bench): fused QKV projection + attention, measured in round 2, loses
#include "study/fused_qkv_attention.inc"
#endif

}  // namespace ivit
