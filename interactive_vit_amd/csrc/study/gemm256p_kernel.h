// 256 x 256 x 64 bf16 GEMM, single-group pipelined K loop (the first 256 x 256 variant; superseded by the
// staggered two-group schedule of gemm256s_kernel.h, which it ties or loses to on every shape).  Built only
// into the microbenchmark (IVIT_GEMM_ABLATIONS, tools/gemm_bench) as the baseline of that comparison.
#pragma once
#include "../gemm256_kernel.h"

namespace ivit {

// One K-tile (4 phases) - deliberately BRANCH-FREE: a branch between the issue of a ds_read and its
// first use makes hipcc fall back to s_waitcnt lgkmcnt(0) (which would also wait for the reads just
// issued for the NEXT phase and serialise LDS latency with the MFMAs).  So:
//  * fragment reads are unconditional (past the last K-tile they fetch stale LDS that is never used);
//  * every phase stages its half-tile unconditionally: past the end of K the SOURCE K-tile is clamped
//    to the last one while the destination is still the (dead) buffer of the steady-state schedule,
//    so the DMA queue keeps its shape and every wait is the constant vmcnt(8).
// Register arrays: a0/a1 = A halves, bx = B0(t) on entry (read during the previous tile's P4),
// by = scratch for B1(t).  On exit a0 holds A0(t+1) and BY holds B0(t+1): the caller swaps the roles
// of bx/by every K-tile, so no copies and no third B array are needed (96 operand VGPRs in all).
template <int DBG>
__device__ __forceinline__ void g256_ktile(const G256Ctx& c, int t, int last_kt, f32x4 (&acc)[8][4], bf16x8 (&a0)[4][2],
                                           bf16x8 (&a1)[4][2], bf16x8 (&bx)[2][2], bf16x8 (&by)[2][2]) {
    const int s1 = min(t + 1, last_kt), s2 = min(t + 2, last_kt);   // clamped SOURCE K-tiles
    // ---- P1: MFMA A0.B0 | read B1(t) -> by | stage A1(t+1)
    IVIT_VMCNT(8);
    __builtin_amdgcn_s_barrier();
    if (DBG != 1) g256_stage<false>(c, s1, t + 1, 1);
    g256_read_b(c, t, 1, by);
    __builtin_amdgcn_sched_barrier(0);
    if (DBG != 2) g256_mma<0, 0>(acc, a0, bx);
    __builtin_amdgcn_sched_barrier(0);
    // ---- P2: MFMA A0.B1 | read A1(t) -> a1 | stage A0(t+2)
    IVIT_VMCNT(8);
    __builtin_amdgcn_s_barrier();
    if (DBG != 1) g256_stage<false>(c, s2, t + 2, 0);
    g256_read_a(c, t, 1, a1);
    __builtin_amdgcn_sched_barrier(0);
    if (DBG != 2) g256_mma<0, 1>(acc, a0, by);
    __builtin_amdgcn_sched_barrier(0);
    // ---- P3: MFMA A1.B1 | (nothing to read: B0 is still in bx) | stage B0(t+2)
    __builtin_amdgcn_s_barrier();
    if (DBG != 1) g256_stage<true>(c, s2, t + 2, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (DBG != 2) g256_mma<1, 1>(acc, a1, by);
    __builtin_amdgcn_sched_barrier(0);
    // ---- P4: MFMA A1.B0 | read A0(t+1) -> a0 (dead since P2), B0(t+1) -> by (dead since P3) | stage B1(t+2)
    IVIT_VMCNT(8);
    __builtin_amdgcn_s_barrier();
    if (DBG != 1) g256_stage<true>(c, s2, t + 2, 1);
    g256_read_b(c, t + 1, 0, by);
    g256_read_a(c, t + 1, 0, a0);
    __builtin_amdgcn_sched_barrier(0);
    if (DBG != 2) g256_mma<1, 0>(acc, a1, bx);
    __builtin_amdgcn_sched_barrier(0);
}

template <int DBG>
__device__ __forceinline__ void gemm256_body(const GemmParams& p, char* smem) {
    using T = Tile256P;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int fr = lane & 15, fq = lane >> 4;

    int tm, tn;
    tile_coords(xcd_tile(blockIdx.x, gridDim.x), ceil_div(p.M, T::BM), ceil_div(p.N, T::BN), tm, tn);
    const int m0 = tm * T::BM;
    const int n0 = tn * T::BN;

    G256Ctx c;
    c.smem = smem; c.wave = wave; c.lda = p.lda; c.ldw = p.ldw; c.a_wrap = p.a_wrap; c.a_shift = p.a_shift;
    {
        const int r_in = lane >> 3;
        const int chunk = (lane & 7) ^ r_in;   // half-tile row & 7 == r_in (pieces are 8-row aligned)
        c.a_src = p.A + (size_t)(m0 + wave * 8 + r_in) * p.lda + chunk * 8;
        c.w_src = p.W + (size_t)(n0 + (wave >> 2) * 64 + (wave & 3) * 8 + r_in) * p.ldw + chunk * 8;
    }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        // row inside the half-tile: A: wr*64 + mf*16 + fr ; B: wc*32 + nf*16 + fr ; (row & 7) == (fr & 7)
        c.a_rd[kk] = (wr * 64 + fr) * 128 + (((kk * 4 + fq) ^ (fr & 7)) << 4);
        c.b_rd[kk] = (wc * 32 + fr) * 128 + (((kk * 4 + fq) ^ (fr & 7)) << 4);
    }

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nt = p.K / GEMM_BK;
    const int last_kt = nt - 1;
    // ---- prologue: DMA queue in steady-state order: tile 0 = A0 B0 B1 A1, tile 1 = A0 B0 B1 (its A1
    // is staged by tile 0's P1, as in steady state); with a single K-tile the second group re-reads it
    const int k1 = min(1, last_kt);
    g256_stage<false>(c, 0, 0, 0);
    g256_stage<true>(c, 0, 0, 0);
    g256_stage<true>(c, 0, 0, 1);
    g256_stage<false>(c, 0, 0, 1);
    g256_stage<false>(c, k1, 1, 0);
    g256_stage<true>(c, k1, 1, 0);
    g256_stage<true>(c, k1, 1, 1);
    IVIT_VMCNT(10);   // A0(0), B0(0) landed; 5 half-tiles still in flight
    __builtin_amdgcn_s_barrier();
    bf16x8 a0[4][2], a1[4][2], bA[2][2], bB[2][2];
    g256_read_b(c, 0, 0, bA);
    g256_read_a(c, 0, 0, a0);

    // ---- main loop, two K-tiles per iteration (the B register arrays swap roles every tile)
    int t = 0;
    for (; t + 1 < nt; t += 2) {
        g256_ktile<DBG>(c, t, last_kt, acc, a0, a1, bA, bB);
        g256_ktile<DBG>(c, t + 1, last_kt, acc, a0, a1, bB, bA);
    }
    if (t < nt) g256_ktile<DBG>(c, t, last_kt, acc, a0, a1, bA, bB);   // odd K-tile count

    gemm_epilogue<T>(p, acc, m0 + wr * 128, n0 + wc * 64, fr, fq);
    IVIT_VMCNT(0);   // the clamped tail stagings may still be writing LDS
}

}  // namespace ivit
