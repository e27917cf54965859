// 256 x 256 x 64 bf16 MFMA GEMM tile with a deep global->LDS DMA pipeline (gfx950).
//
// Why a second kernel: the 128-class tiles of gemm_kernel.h need ~0.0156 operand bytes per FLOP
// from L2 - at 640 TFLOP/s that is ~20 TB/s, the practical L2->LDS ceiling - so they cannot go
// faster whatever their schedule.  A 256 x 256 tile halves the traffic; what it needs instead is
// latency tolerance at ONE workgroup per CU (128 KiB of LDS), i.e. DMA that stays in flight across
// barriers (counted s_waitcnt vmcnt, raw s_barrier) instead of a drain per K-step.
//
// Structure (8 waves = 2(M) x 4(N), each wave 128 x 64 = 8 x 4 MFMA fragments, 128 accumulator VGPRs):
//  * an operand K-tile (256 rows x 64 k, 32 KiB) is split in two HALF-TILES of 128 rows: A_h holds
//    rows {wr*128 + h*64 + i} (the h-th 64 rows of both wave rows), B_h holds W rows
//    {wc*64 + h*32 + j}.  LDS = 2 sets x {A0, A1, B0, B1} x 16 KiB = 128 KiB.
//  * a K-tile is multiplied in 4 PHASES of 16 MFMAs, one 64x32 quadrant of the wave's output each:
//        P1: A0.B0    P2: A0.B1    P3: A1.B1    P4: A1.B0        (B0 stays in registers P1..P4)
//    so every phase needs at most one NEW half-operand, which is read from LDS during the PREVIOUS
//    phase (software pipelining: ds_reads of phase p+1 are issued before the MFMAs of phase p and
//    return under them):   read in P4(t-1): A0,B0(t) | P1: B1 | P2: A1 | P3: -
//  * every phase stages ONE half-tile (2 global_load_lds_dwordx4 per wave) into the buffer whose
//    last ds_read was issued two phases earlier (all waves have consumed it before the barrier
//    that precedes the staging):   P1: A1(t+1)   P2: A0(t+2)   P3: B0(t+2)   P4: B1(t+2)
//    => 4 half-tiles (64 KiB) of DMA are in flight per CU at any time, 5-6 phases ahead of use.
//  * one s_barrier per phase, preceded by s_waitcnt vmcnt(8) (= all but the 4 newest half-tiles of
//    THIS wave's DMA have landed); after the barrier the half-tile that this phase's ds_reads
//    touch has landed for every wave.  P3 needs nothing new and does not wait.  The first and last
//    two K-tiles use the smaller counts that their shorter DMA queue implies (see gemm256_body).
// LDS rows are 128 B with the 16-B chunk XOR-swizzled by (row & 7) on the DMA source address and
// on the read address (conflict-free ds_read_b128), as in gemm_kernel.h.
#pragma once
#include "gemm_kernel.h"

namespace ivit {

struct Tile256P {   // shape constants shared with the generic epilogue
    static constexpr int WAVES_M = 2, WAVES_N = 4, FM = 8, FN = 4;
    static constexpr int WAVES = 8, THREADS = 512, BM = 256, BN = 256;
    static constexpr int HALF_BYTES = 128 * 128;          // 16 KiB
    static constexpr int SET_BYTES = 4 * HALF_BYTES;      // A0 A1 B0 B1
    static constexpr int LDS_BYTES = 2 * SET_BYTES;       // 128 KiB
    static constexpr int OFF_A0 = 0, OFF_A1 = HALF_BYTES, OFF_B0 = 2 * HALF_BYTES, OFF_B1 = 3 * HALF_BYTES;
};

#define IVIT_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")

struct G256Ctx {
    const bf16_t* a_src;   // this lane's DMA source for A half-tile h=0, piece i=0, k0=0
    const bf16_t* w_src;   // same for W
    int lda, ldw;
    char* smem;
    int wave;
    int a_rd[2];           // per-lane LDS read offsets inside an A half-tile for kk = 0, 1 (mf = 0)
    int b_rd[2];           // same inside a B half-tile (nf = 0)
};

// stage half-tile `h` of operand A (IS_W = false) or W (true) of K-tile `kt` into set `dst_kt & 1`
template <bool IS_W>
__device__ __forceinline__ void g256_stage(const G256Ctx& c, int kt, int dst_kt, int h) {
    char* dst = c.smem + (dst_kt & 1) * Tile256P::SET_BYTES + (IS_W ? Tile256P::OFF_B0 : Tile256P::OFF_A0) + h * Tile256P::HALF_BYTES;
    if (!IS_W) {
        // piece pc = wave + 8 i  ->  half-tile rows pc*8 .. +8  ->  tile row i*128 + h*64 + wave*8 + r_in
        const bf16_t* s = c.a_src + (size_t)(h * 64) * c.lda + kt * GEMM_BK;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_global_load_lds((const IVIT_GLOBAL void*)(s + (size_t)(i * 128) * c.lda),
                                             (IVIT_LDS void*)(dst + (c.wave + 8 * i) * 1024), 16, 0, 0);
    } else {
        // half-tile row = wave*8 + i*64 + r_in = wc'*32 + j with wc' = (wave>>2) + 2i, j = (wave&3)*8 + r_in
        // -> W tile row wc'*64 + h*32 + j ; the (wave, r_in) part is folded into w_src
        const bf16_t* s = c.w_src + (size_t)(h * 32) * c.ldw + kt * GEMM_BK;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_global_load_lds((const IVIT_GLOBAL void*)(s + (size_t)(i * 128) * c.ldw),
                                             (IVIT_LDS void*)(dst + (c.wave + 8 * i) * 1024), 16, 0, 0);
    }
}

// fragment reads of one half-operand (all kk): A half -> 4 x 2 fragments, B half -> 2 x 2
__device__ __forceinline__ void g256_read_a(const G256Ctx& c, int kt, int h, bf16x8 (&f)[4][2]) {
    const char* base = c.smem + (kt & 1) * Tile256P::SET_BYTES + Tile256P::OFF_A0 + h * Tile256P::HALF_BYTES;
#pragma unroll
    for (int mf = 0; mf < 4; ++mf)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
            f[mf][kk] = *reinterpret_cast<const bf16x8*>(base + c.a_rd[kk] + mf * 2048);
}
__device__ __forceinline__ void g256_read_b(const G256Ctx& c, int kt, int h, bf16x8 (&f)[2][2]) {
    const char* base = c.smem + (kt & 1) * Tile256P::SET_BYTES + Tile256P::OFF_B0 + h * Tile256P::HALF_BYTES;
#pragma unroll
    for (int nf = 0; nf < 2; ++nf)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
            f[nf][kk] = *reinterpret_cast<const bf16x8*>(base + c.b_rd[kk] + nf * 2048);
}

// 16 MFMAs: quadrant (ha, hb) of the wave's 8 x 4 accumulator fragments
// FP8: the fragments hold e4m3 bytes; the two 16-B k-chunks of a row are the operands of ONE 128-deep
// scaled-form MFMA (mfma_e4m3_16x16x128), so a quadrant is 8 MFMAs of twice the cycles instead of 16
template <int HA, int HB, bool FP8 = false>
__device__ __forceinline__ void g256_mma(f32x4 (&acc)[8][4], const bf16x8 (&a)[4][2], const bf16x8 (&b)[2][2]) {
    __builtin_amdgcn_s_setprio(1);
    if (FP8) {
#pragma unroll
        for (int mf = 0; mf < 4; ++mf)
#pragma unroll
            for (int nf = 0; nf < 2; ++nf)
                acc[HA * 4 + mf][HB * 2 + nf] = mfma_e4m3_16x16x128(b[nf], a[mf], acc[HA * 4 + mf][HB * 2 + nf]);
    } else {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int mf = 0; mf < 4; ++mf)
#pragma unroll
                for (int nf = 0; nf < 2; ++nf)
                    acc[HA * 4 + mf][HB * 2 + nf] =
                        __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[nf][kk], a[mf][kk], acc[HA * 4 + mf][HB * 2 + nf], 0, 0, 0);
    }
    __builtin_amdgcn_s_setprio(0);
}

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
    c.smem = smem; c.wave = wave; c.lda = p.lda; c.ldw = p.ldw;
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
