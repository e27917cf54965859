// 160 x 256 x 64 bf16 MFMA GEMM, FOUR waves (one per SIMD), software-pipelined inside each wave (gfx950).
//
// Why.  tools/blas_ceiling.py + rocprofv3 show what the vendor GEMM does on the ViT-B shapes
// (M = 12608): a 160 x 256 x 64 macro-tile, 4 waves of 80 x 128, local-read prefetch - and it beats the
// 160 x 128 / 256 x 256 kernels on the wide, short-K shapes (mlp1: 967 vs 834 TFLOP/s).  The 8-wave
// kernels keep the matrix pipe fed by pairing waves on a SIMD (one multiplies while its partner
// stages / reads); here ONE wave per SIMD issues its MFMAs back to back and slots every other
// instruction into their shadow:
//   * wave tile 80 x 128 = 5 x 8 fragments: 26 ds_read_b128 feed 80 MFMAs per 64-deep K-step
//     (0.33 reads per MFMA; the 8-wave 80 x 64 wave tile needs 0.45);
//   * BOTH 32-deep halves of a stage are held in registers, in two sets X / Y (208 VGPRs; the 160
//     accumulator registers live in AGPRs): while the 80 MFMAs of stage t run on X, the 26 reads of stage
//     t+1 fill Y, one read per 3 MFMAs - a stage is read a whole K-step before it is multiplied;
//   * so an LDS stage is busy only while it is in flight or being read: three 52-KiB stages, ONE
//     s_barrier per K-step at its top, and the slot released by that barrier (stage t's) takes the 13
//     LDS-DMA of stage t+3 at once.  TWO full stages (104 KiB) are always in flight.  That is what the
//     K loop's rate comes from: the L2->LDS path is latency-bound (bytes in flight / ~1 us from L2, ~2 us
//     from the Infinity Cache; MI355X_MICROARCH.md "gather into LDS" measures the same), not
//     bandwidth-bound, so what the stage ring must maximise is bytes in flight, not stages held.
// DMA goes through buffer descriptors (scalar K / piece offsets, out-of-range offsets past the end
// of K move no data but keep the vmcnt arithmetic constant), as in gemm160x256_kernel.h.
#pragma once
#include "../gemm256s_kernel.h"   // IVIT_STAMP

namespace ivit {

#ifdef IVIT_GEMM_ABLATIONS   // shader-clock stamps (s_memtime) next to the 100 MHz ones: slots 8.. of the block's record
#define IVIT_STAMP_CLK(slot)                                                                                  \
    do {                                                                                                      \
        if (p.stamps && threadIdx.x == 0) {                                                                   \
            unsigned long long t_;                                                                            \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                          \
            p.stamps[(size_t)blockIdx.x * 16 + 8 + (slot)] = t_;                                              \
        }                                                                                                     \
    } while (0)
#else
#define IVIT_STAMP_CLK(slot) do { } while (0)
#endif

struct Tile160x256W4 {
    static constexpr bool RAGGED_N = true;   // generic element-guarded edge epilogue
    static constexpr int WAVES_M = 2, WAVES_N = 2, FM = 5, FN = 8;
    static constexpr int WAVES = 4, THREADS = 256, BM = 160, BN = 256;
    static constexpr int A_BYTES = BM * 128, W_BYTES = BN * 128;
    static constexpr int STAGE_BYTES = A_BYTES + W_BYTES;       // 53248
    static constexpr int STAGES = 3;
    static constexpr int LDS_BYTES = STAGES * STAGE_BYTES;       // 159744 <= 163840
    static constexpr int A_DMA = 5, W_DMA = 8;                   // 1-KiB pieces per wave and stage
};

using w4_rsrc_t = decltype(__builtin_amdgcn_make_buffer_rsrc((void*)nullptr, (short)0, 0, 0));

struct W4Ctx {
    w4_rsrc_t a_rsrc, w_rsrc;
    int a_voff, w_voff;        // per-lane byte offsets of (tile row wave*8 + r_in, swizzled chunk)
    int a_group, w_group;      // byte stride between this wave's consecutive pieces (32 rows)
    int wave;
    char* smem;
    int a_rd[2], w_rd[2];      // per-lane LDS read offsets inside a stage for kk = 0, 1 (fragment 0)
    int ablate;                // microbenchmark only: 1 = no DMA traffic in the K loop, 2 = no fragment reads
};

constexpr int W4_OOB = 0x7f000000;

// DMA number i (0..12) of this wave for one stage: i < 5 -> A piece wave + 4 i, else W piece wave + 4 (i - 5)
template <int I>
__device__ __forceinline__ void w4_dma(const W4Ctx& c, int k_bytes, char* stage) {
    if (I < Tile160x256W4::A_DMA) {
        const int soff = k_bytes >= 0 ? k_bytes + I * c.a_group : W4_OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(c.a_rsrc, (IVIT_LDS void*)(stage + (c.wave + 4 * I) * 1024), 16, c.a_voff, soff, 0, 0);
    } else {
        constexpr int J = I - Tile160x256W4::A_DMA;
        const int soff = k_bytes >= 0 ? k_bytes + J * c.w_group : W4_OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(c.w_rsrc, (IVIT_LDS void*)(stage + Tile160x256W4::A_BYTES + (c.wave + 4 * J) * 1024),
                                                 16, c.w_voff, soff, 0, 0);
    }
}

template <int I>
__device__ __forceinline__ void w4_dma_all(const W4Ctx& c, int k_bytes, char* stage) {
    if constexpr (I < 13) {
        w4_dma<I>(c, k_bytes, stage);
        w4_dma_all<I + 1>(c, k_bytes, stage);
    }
}

// fragment number i (0..25) of a K-step: kk = i / 13; within a half: j < 5 -> A fragment j, else W fragment j - 5
template <int I>
__device__ __forceinline__ bf16x8 w4_read(const W4Ctx& c, const char* stage) {
    constexpr int KK = I / 13, J = I % 13;
    if (J < 5) return *reinterpret_cast<const bf16x8*>(stage + c.a_rd[KK] + J * 2048);
    return *reinterpret_cast<const bf16x8*>(stage + Tile160x256W4::A_BYTES + c.w_rd[KK] + (J - 5) * 2048);
}

template <int I>
__device__ __forceinline__ void w4_read_all(const W4Ctx& c, const char* stage, bf16x8 (&f)[26]) {
    if constexpr (I < 26) {
        f[I] = w4_read<I>(c, stage);
        w4_read_all<I + 1>(c, stage, f);
    }
}

// MFMA number j (0..79) of a K-step: half kk = j / 40, fragment pair (mf, nf) = ((j % 40) / 8, j % 8)
template <int J>
__device__ __forceinline__ void w4_mfma(f32x4 (&acc)[5][8], const bf16x8 (&f)[26]) {
    constexpr int KK = J / 40, MF = (J % 40) / 8, NF = J % 8;
    // accumulators pinned to AGPRs ("+a"): left to itself the register allocator splits the accumulators
    // between the two files and shuffles them with v_accvgpr moves inside the loop (208 per K-step)
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[MF][NF]) : "v"(f[KK * 13 + 5 + NF]), "v"(f[KK * 13 + MF]));
}

// One K-step: 80 MFMAs on `cur`, with group g (0..25) = { MFMA 3g, 3g+1, 3g+2 ; ds_read g of the NEXT
// stage -> nxt } and MFMAs 78, 79 at the end.
// The 13 LDS-DMA of the step are spread over it, one per six MFMAs: the CU's four waves share one
// vector-memory front end (64 B/clk: 16 cycles per 1-KiB DMA), and a wave that issues into a full
// queue stalls - with ONE wave per SIMD that is a stalled matrix pipe (13 DMA issued back to back by all
// four waves cost 949 cycles per K-step in in-kernel s_memtime stamps, against 1377 for the 80 MFMAs).
template <int G>
__device__ __forceinline__ void w4_step(const W4Ctx& c, f32x4 (&acc)[5][8], const bf16x8 (&cur)[26], bf16x8 (&nxt)[26], const char* rd_stage,
                                        int k_bytes, char* dma_stage) {
    if constexpr (G < 26) {
        w4_mfma<3 * G>(acc, cur);
        w4_mfma<3 * G + 1>(acc, cur);
        w4_mfma<3 * G + 2>(acc, cur);
        nxt[G] = w4_read<G>(c, rd_stage);
        if constexpr (G % 2 == 0) w4_dma<G / 2>(c, k_bytes, dma_stage);
        __builtin_amdgcn_sched_barrier(0);
        w4_step<G + 1>(c, acc, cur, nxt, rd_stage, k_bytes, dma_stage);
    } else {
        w4_mfma<78>(acc, cur);
        w4_mfma<79>(acc, cur);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// top of K-step t: this wave's part of stage t+1 has landed and its reads of stage t (issued during step
// t-1) are complete; the barrier publishes stage t+1 and releases stage t's slot, which takes the DMA of
// stage t+3 at once - TWO FULL stages (104 KiB) are in flight while the third is being read
#ifdef IVIT_GEMM_ABLATIONS
#define IVIT_FINE_CLK(slot) do { if (p.stamps && t == 4 && threadIdx.x == 0) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); p.stamps[(size_t)blockIdx.x * 16 + (slot)] = t_; } } while (0)
#else
#define IVIT_FINE_CLK(slot) do { } while (0)
#endif
__device__ __forceinline__ void w4_top(const GemmParams& p, const W4Ctx& c, int t, int nt, int slot) {
    IVIT_FINE_CLK(11);
    asm volatile("s_waitcnt vmcnt(13)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
    IVIT_FINE_CLK(12);
    __builtin_amdgcn_s_barrier();
    IVIT_FINE_CLK(13);
    IVIT_FINE_CLK(14);
}

__device__ __forceinline__ void gemm160x256w4_body(const GemmParams& p, char* smem) {
    using T = Tile160x256W4;
    IVIT_STAMP(0);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int fr = lane & 15, fq = lane >> 4;

    const int tiles_m = ceil_div(p.M, T::BM), tiles_n = ceil_div(p.N, T::BN);
    int tm, tn;
    tile_coords(xcd_tile(blockIdx.x, gridDim.x), tiles_m, tiles_n, tm, tn, p.debug >= 3 ? p.debug : GEMM_GROUP_N);
    const int m0 = tm * T::BM, n0 = tn * T::BN;

    W4Ctx c;
    c.smem = smem; c.wave = wave; c.ablate = p.debug;
    {
        // descriptors over the readable extent of each operand (engine.hip pads A to round_up(M,256)+256
        // rows and W to round_up(N,256) rows); tile rows past M / N read padding, never stored
        const unsigned a_bytes = (unsigned)((size_t)(round_up(p.M, 256) + 256) * p.lda * 2);
        const unsigned w_bytes = (unsigned)((size_t)round_up(p.N, 256) * p.ldw * 2);
        c.a_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.A), 0, a_bytes, 0x00020000);
        c.w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.W), 0, w_bytes, 0x00020000);
        const int r_in = lane >> 3;
        const int chunk = (lane & 7) ^ r_in;   // tile row & 7 == r_in (pieces are 8-row aligned)
        c.a_voff = ((m0 + wave * 8 + r_in) * p.lda) * 2 + chunk * 16;
        c.w_voff = ((n0 + wave * 8 + r_in) * p.ldw) * 2 + chunk * 16;
        c.a_group = 32 * p.lda * 2;
        c.w_group = 32 * p.ldw * 2;
    }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {   // row & 7 == fr & 7 for every fragment row (80 and 128 are multiples of 8)
        c.a_rd[kk] = (wr * 80 + fr) * 128 + (((kk * 4 + fq) ^ (fr & 7)) << 4);
        c.w_rd[kk] = (wc * 128 + fr) * 128 + (((kk * 4 + fq) ^ (fr & 7)) << 4);
    }

    f32x4 acc[5][8];
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nt = p.K / GEMM_BK;
    // ---- prologue: stages 0, 1, 2 in flight; X <- every fragment of stage 0
    w4_dma_all<0>(c, 0, smem);
    w4_dma_all<0>(c, nt > 1 ? 128 : -1, smem + T::STAGE_BYTES);
    w4_dma_all<0>(c, nt > 2 ? 256 : -1, smem + 2 * T::STAGE_BYTES);
    asm volatile("s_waitcnt vmcnt(26)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    bf16x8 fx[26], fy[26];
    w4_read_all<0>(c, smem, fx);
    IVIT_STAMP(1);
    IVIT_STAMP_CLK(1);

    // K-steps in pairs so that the two fragment sets swap roles without copies; slot = t % 3
    int slot = 0;
    for (int t = 0; t < nt; t += 2) {
        const int slot1 = (slot == 2) ? 0 : slot + 1, slot2 = (slot1 == 2) ? 0 : slot1 + 1;
        w4_top(p, c, t, nt, slot);
        // multiply stage t, read stage t+1, stage t+3 into stage t's slot
        w4_step<0>(c, acc, fx, fy, smem + slot1 * T::STAGE_BYTES, (t + 3 < nt && c.ablate != 1) ? (t + 3) * 128 : -1, smem + slot * T::STAGE_BYTES);
        IVIT_FINE_CLK(15);
        if (t + 1 < nt) {
            w4_top(p, c, t + 1, nt, slot1);
            w4_step<0>(c, acc, fy, fx, smem + slot2 * T::STAGE_BYTES, (t + 4 < nt && c.ablate != 1) ? (t + 4) * 128 : -1, smem + slot1 * T::STAGE_BYTES);
        }
        slot = slot2;
    }

    // the MFMAs are inline asm, invisible to the compiler's hazard recogniser: cover the longest
    // MFMA-result -> v_accvgpr_read distance by hand before the epilogue touches the accumulators
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    IVIT_STAMP(2);
    IVIT_STAMP_CLK(2);
    gemm_epilogue<T>(p, acc, m0 + wr * 80, n0 + wc * 128, fr, fq);
    IVIT_STAMP(3);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // out-of-range run-ahead stages still count
    IVIT_STAMP(4);
}

}  // namespace ivit
