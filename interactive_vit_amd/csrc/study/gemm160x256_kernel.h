// 160 x 256 x 64 bf16 MFMA GEMM with a three-stage LDS ring (gfx950).
//
// Why this shape.  Measured on MI355X: the GEMMs are bound by the bytes each
// CU can pull from L2 into LDS (~55-75 GB/s depending on how much is kept in flight), so what counts
// is FLOP per operand byte and never draining the DMA queue.  A 160 x 256 tile does 98 FLOP/B
// (160x128: 71) and - 12608 token rows being 78.8 x 160 - its grids are 237 / 711 / 948 tiles for
// N = 768 / 2304 / 3072, i.e. 92.6 % of 1, 3 and 4 rounds of the 256 CUs, where 256 x 256 tiles give
// 150 / 450 / 600 (59 % / 88 % / 78 %).  Three 52-KiB stages (156 KiB of the 160 KiB LDS, one
// 8-wave workgroup per CU) keep TWO stages = 104 KiB of LDS-DMA in flight behind a counted
// s_waitcnt vmcnt, with one raw s_barrier per 64-deep K-step.
//
// LDS-DMA goes through buffer descriptors (buffer_load_dwordx4 ... lds): the per-lane offset is
// loop-invariant, everything that moves (K-step, 64-row piece group) is a scalar offset - no VALU per
// DMA - and the descriptor's range check makes the loads that run ahead past the end of K (and the
// one surplus piece of waves 4-7) cost no memory traffic while still counting in vmcnt, so every
// wave issues exactly 7 DMA per stage and the wait count is the constant 7.
//
// 8 waves = 2 (M) x 4 (N); a wave owns 80 x 64 = 5 x 4 MFMA fragments (80 accumulator VGPRs); the
// fragment reads of the second 32-deep half of a K-step are issued before the MFMAs of the first.
#pragma once
#include "../gemm_kernel.h"

namespace ivit {

struct Tile160x256 {
    static constexpr bool RAGGED_N = true;   // generic element-guarded edge epilogue
    static constexpr int WAVES_M = 2, WAVES_N = 4, FM = 5, FN = 4;
    static constexpr int WAVES = 8, THREADS = 512, BM = 160, BN = 256;
    static constexpr int A_BYTES = BM * 128, W_BYTES = BN * 128;
    static constexpr int STAGE_BYTES = A_BYTES + W_BYTES;       // 53248
    static constexpr int STAGES = 3;
    static constexpr int SCRATCH_OFF = STAGES * STAGE_BYTES;     // 1 KiB landing zone of the out-of-range dummies
    static constexpr int LDS_BYTES = SCRATCH_OFF + 1024;         // 160768 <= 163840
    static constexpr int DMA_PER_WAVE = 7;                       // 3 A pieces (one may be a dummy) + 4 W pieces
};

using g160_rsrc_t = decltype(__builtin_amdgcn_make_buffer_rsrc((void*)nullptr, (short)0, 0, 0));

struct G160Ctx {
    g160_rsrc_t a_rsrc, w_rsrc;
    int a_voff, w_voff;        // per-lane byte offsets of (tile row wave*8 + r_in, swizzled chunk) inside A / W
    int a_group, w_group;      // scalar byte stride between 64-row piece groups (64 * ld * 2)
    int wave;
    char* smem;
};

// out-of-range scalar offset: raw buffer loads whose offset is >= num_records move no data
constexpr int G160_OOB = 0x7f000000;

// one stage = 7 buffer_load ... lds per wave; `k_bytes` < 0 requests a no-traffic (out-of-range) stage
__device__ __forceinline__ void g160_stage(const G160Ctx& c, int k_bytes, int slot) {
    char* stage = c.smem + slot * Tile160x256::STAGE_BYTES;
    const bool live = k_bytes >= 0;
#pragma unroll
    for (int i = 0; i < 3; ++i) {   // A: pieces i*8 + wave of 20
        const bool in_tile = (i < 2) || (c.wave < 4);
        const int soff = (live && in_tile) ? k_bytes + i * c.a_group : G160_OOB;
        char* dst = in_tile ? stage + (i * 8 + c.wave) * 1024 : c.smem + Tile160x256::SCRATCH_OFF;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(c.a_rsrc, (IVIT_LDS void*)dst, 16, c.a_voff, soff, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {   // W: pieces i*8 + wave of 32
        const int soff = live ? k_bytes + i * c.w_group : G160_OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(c.w_rsrc, (IVIT_LDS void*)(stage + Tile160x256::A_BYTES + (i * 8 + c.wave) * 1024),
                                                 16, c.w_voff, soff, 0, 0);
    }
}

__device__ __forceinline__ void gemm160x256_body(const GemmParams& p, char* smem) {
    using T = Tile160x256;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int fr = lane & 15, fq = lane >> 4;

    const int tiles_m = ceil_div(p.M, T::BM), tiles_n = ceil_div(p.N, T::BN);
    int tm, tn;
    tile_coords(xcd_tile(blockIdx.x, gridDim.x), tiles_m, tiles_n, tm, tn);
    const int m0 = tm * T::BM, n0 = tn * T::BN;

    G160Ctx c;
    c.smem = smem; c.wave = wave;
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
        c.a_group = 64 * p.lda * 2;
        c.w_group = 64 * p.ldw * 2;
    }
    // per-lane LDS read offsets (row & 7 == fr & 7 for every fragment row)
    int a_rd[2], w_rd[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        a_rd[kk] = (wr * 80 + fr) * 128 + (((kk * 4 + fq) ^ (fr & 7)) << 4);
        w_rd[kk] = T::A_BYTES + (wc * 64 + fr) * 128 + (((kk * 4 + fq) ^ (fr & 7)) << 4);
    }

    f32x4 acc[T::FM][T::FN];
#pragma unroll
    for (int i = 0; i < T::FM; ++i)
#pragma unroll
        for (int j = 0; j < T::FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nt = p.K / GEMM_BK;
    g160_stage(c, 0, 0);
    g160_stage(c, nt > 1 ? 128 : -1, 1);

    int slot = 0;   // t % 3
    for (int t = 0; t < nt; ++t) {
        // stage t has landed for this wave (its 7 DMA of stage t+1 may still be in flight); the barrier
        // publishes it and also orders every wave's reads of stage t-1 before that buffer is re-staged
        asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const int nslot = (slot == 0) ? 2 : slot - 1;          // (t + 2) % 3
        g160_stage(c, (t + 2 < nt) ? (t + 2) * 128 : -1, nslot);
        const char* st = smem + slot * T::STAGE_BYTES;

        bf16x8 af0[T::FM], wf0[T::FN], af1[T::FM], wf1[T::FN];
#pragma unroll
        for (int j = 0; j < T::FN; ++j) wf0[j] = *reinterpret_cast<const bf16x8*>(st + w_rd[0] + j * 2048);
#pragma unroll
        for (int i = 0; i < T::FM; ++i) af0[i] = *reinterpret_cast<const bf16x8*>(st + a_rd[0] + i * 2048);
#pragma unroll
        for (int j = 0; j < T::FN; ++j) wf1[j] = *reinterpret_cast<const bf16x8*>(st + w_rd[1] + j * 2048);
#pragma unroll
        for (int i = 0; i < T::FM; ++i) af1[i] = *reinterpret_cast<const bf16x8*>(st + a_rd[1] + i * 2048);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < T::FM; ++i)
#pragma unroll
            for (int j = 0; j < T::FN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf0[j], af0[i], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < T::FM; ++i)
#pragma unroll
            for (int j = 0; j < T::FN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf1[j], af1[i], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        slot = (slot == 2) ? 0 : slot + 1;
    }

    gemm_epilogue<T>(p, acc, m0 + wr * 80, n0 + wc * 64, fr, fq);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // out-of-range run-ahead stages still write LDS
}

}  // namespace ivit
