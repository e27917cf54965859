// 256 x 256 x 64 bf16 MFMA GEMM, staggered two-group schedule (gfx950).
//
// Same tile, half-tile LDS layout and DMA ring as gemm256_kernel.h, different phase structure.
// Measured there: with one barrier per phase all 8 waves issue their DMA + ds_reads together and then
// their MFMAs together, so on every SIMD the matrix pipe idles through ~300-400 cycles of issue
// overhead per 512 cycles of MFMA work (62 % in the ablation without DMA).  Here every phase is
// split in two barrier intervals,
//        SR_p : stage one half-tile (2 LDS-DMA per wave), ds_read the phase's new operand half
//        M_p  : 16 MFMAs (one 64 x 32 quadrant of the wave's 128 x 64 output)
// and waves 4-7 (which share their SIMDs with waves 0-3) run ONE INTERVAL BEHIND waves 0-3, so at
// any time one wave of a SIMD is in M while its partner is in SR: matrix beside memory.
//
//   K-tile phases (quadrants)  P1: A0.B0   P2: A0.B1   P3: A1.B1   P4: A1.B0   (B0 stays in registers)
//   ds_reads (in SR_p)         P1: A0,B0   P2: B1      P3: A1      P4: -
//   staging (in SR_p)          P1: B1(t+1) P2: A1(t+1) P3: A0(t+2) P4: B0(t+2)
//   s_waitcnt vmcnt(8) at the end of SR_4, SR_1, SR_2 (= the half-tile(s) read in the NEXT phase have
//   landed for this wave; 4 half-tiles = 64 KiB of DMA stay in flight); the barrier that closes the
//   interval publishes them to the other waves.
//
// Hazards (interval k = time between barriers k and k+1; group 0 runs SR_p in interval 2p, M_p in
// 2p+1; group 1 one later):
//   RAW  half-tile H read in phase r: group 0 reads in interval 2r, group 1 in 2r+1.  Every wave
//        waits for ITS pieces of H at the end of its SR_{r-1} (interval 2r-2 or 2r-1), i.e. before
//        barrier 2r, which precedes both reads.
//   WAR  H last read in phase p: group 1's reads are issued in interval 2p+1 and retired by its
//        lgkmcnt(0) at the start of M_p (interval 2p+2).  The earliest restage is in phase p+2
//        (group 0: interval 2p+4, group 1: 2p+5), both after barrier 2p+3 which group 1 passes after
//        that wait.  Schedule: A0 read P1 -> staged P3; B0 P1 -> P4; B1 P2 -> P1'; A1 P3 -> P2'.
// Past the end of K the staging goes on with a clamped source tile into the (dead) steady-state
// buffer, so the K-tile body is branch-free and every count constant.
#pragma once
#include "gemm256_kernel.h"

#ifndef IVIT_RSG_256
#define IVIT_RSG_256 2   // fragment rows per residual-load group of the 256 x 256 tile's f32-output epilogues (A/B builds: 1)
#endif

namespace ivit {

template <int DBG, bool FP8, class OP = OpBf16>
__device__ __forceinline__ void g256s_ktile(const G256Ctx& c, int t, int last_kt, f32x4 (&acc)[8][4],
                                            bf16x8 (&a)[4][2], bf16x8 (&b0)[2][2], bf16x8 (&b1)[2][2]) {
    const int s1 = min(t + 1, last_kt), s2 = min(t + 2, last_kt);   // clamped SOURCE K-tiles
    // ---- SR1: stage B1(t+1) | read A0, B0 (landed: waited at the end of the previous SR4)
    if (DBG != 1) g256_stage<true>(c, s1, t + 1, 1);
    g256_read_b(c, t, 0, b0);
    g256_read_a(c, t, 0, a);
    IVIT_VMCNT(8);                       // B1(t) landed (read in SR2)
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (DBG != 2) g256_mma<0, 0, FP8, OP>(acc, a, b0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    // ---- SR2: stage A1(t+1) | read B1
    if (DBG != 1) g256_stage<false>(c, s1, t + 1, 1);
    g256_read_b(c, t, 1, b1);
    IVIT_VMCNT(8);                       // A1(t) landed (read in SR3)
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (DBG != 2) g256_mma<0, 1, FP8, OP>(acc, a, b1);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    // ---- SR3: stage A0(t+2) | read A1 (into the registers A0 just vacated)
    if (DBG != 1) g256_stage<false>(c, s2, t + 2, 0);
    g256_read_a(c, t, 1, a);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (DBG != 2) g256_mma<1, 1, FP8, OP>(acc, a, b1);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    // ---- SR4: stage B0(t+2) | nothing to read (B0 is still in registers)
    if (DBG != 1) g256_stage<true>(c, s2, t + 2, 0);
    IVIT_VMCNT(8);                       // A0(t+1), B0(t+1) landed (read in the next SR1)
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (DBG != 2) g256_mma<1, 0, FP8, OP>(acc, a, b0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
}

#ifdef IVIT_GEMM_ABLATIONS
#define IVIT_STAMP(slot)                                                                                      \
    do {                                                                                                      \
        if (p.stamps && (threadIdx.x & 255) == 0) {                                                           \
            unsigned long long t_;                                                                            \
            asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                     \
            p.stamps[((size_t)blockIdx.x * 2 + (threadIdx.x >> 8)) * 8 + (slot)] = t_;                        \
        }                                                                                                     \
    } while (0)
#else
#define IVIT_STAMP(slot) do { } while (0)
#endif

// FP8: A and W are e4m3 BYTE matrices.  A K-tile is 128 bytes of every row either way, so the fp8
// operands are staged as bf16 matrices of half the row length (lda, ldw % 16 == 0, K % 128 == 0).
template <int DBG, bool FP8 = false, int EK = 0, class OP = OpBf16>
__device__ __forceinline__ void gemm256s_body(const GemmParams& p, char* smem) {
    using T = Tile256P;
    IVIT_STAMP(0);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int fr = lane & 15, fq = lane >> 4;

    int tm, tn;
    // column panels of at most six 256-wide tiles, balanced (9 column tiles: 5 + 4, 12: 6 + 6, 16: 6 + 5 + 5 as 6 / 6 / 4): the ~32 tiles an XCD runs at once
    // then cover ~6 x 5 row / column tiles - 4.3 MB of operands at K = 768 against 4.7 MB for the 8-wide panels of the smaller tiles' rule.  In the
    // forward, same box (round 5, IVIT_GEMM_GROUP_N): ViT-B/16 B = 64 QKV 53.1 -> 52.4 us, +0.5 % on the step; ViT-L/16-384 QKV 413 -> 409, MLP up 623 -> 618 us
    const int tiles_n = ceil_div(p.N, T::BN);
    const int group = p.debug >= 3 ? p.debug : ceil_div(tiles_n, ceil_div(tiles_n, 6));
    tile_coords(xcd_tile(blockIdx.x, gridDim.x), ceil_div(p.M, T::BM), tiles_n, tm, tn, group);
    const int m0 = tm * T::BM;
    const int n0 = tn * T::BN;

    G256Ctx c;
    c.smem = smem; c.wave = wave; c.lda = FP8 ? p.lda / 2 : p.lda; c.ldw = FP8 ? p.ldw / 2 : p.ldw; c.a_wrap = p.a_wrap; c.a_shift = p.a_shift;
    {
        const int r_in = lane >> 3;
        const int chunk = (lane & 7) ^ r_in;
        c.a_src = p.A + (size_t)(m0 + wave * 8 + r_in) * c.lda + chunk * 8;
        c.w_src = p.W + (size_t)(n0 + (wave >> 2) * 64 + (wave & 3) * 8 + r_in) * c.ldw + chunk * 8;
    }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        c.a_rd[kk] = (wr * 64 + fr) * 128 + (((kk * 4 + fq) ^ (fr & 7)) << 4);
        c.b_rd[kk] = (wc * 32 + fr) * 128 + (((kk * 4 + fq) ^ (fr & 7)) << 4);
    }

    f32x4 acc[8][4];
    gemm_acc_init<T, EK>(p, acc, n0 + wc * 64, fq);

    float2* tile_stats = reinterpret_cast<float2*>(smem + T::LDS_BYTES);   // EK == 2 kernels are launched with BM * 8 more bytes
#if !IVIT_LN_STATS_AFTER_DMA
    if (EK == 2) ln_tile_stats<T>(p, m0, tile_stats);
#endif

    const int nt = p.K / (FP8 ? 2 * GEMM_BK : GEMM_BK);
    const int last_kt = nt - 1;
    // ---- prologue: DMA queue in steady-state order: tile 0 = A0 B0 B1 A1, tile 1 = A0 B0
    const int k1 = min(1, last_kt);
    g256_stage<false>(c, 0, 0, 0);
    g256_stage<true>(c, 0, 0, 0);
    g256_stage<true>(c, 0, 0, 1);
    g256_stage<false>(c, 0, 0, 1);
    g256_stage<false>(c, k1, 1, 0);
    g256_stage<true>(c, k1, 1, 0);
#if IVIT_LN_STATS_AFTER_DMA
    if (EK == 2) ln_tile_stats<T>(p, m0, tile_stats);   // behind the prologue's DMA queue (the compiler's wait for these loads drains it once, here)
#endif
    IVIT_VMCNT(8);   // A0(0), B0(0) landed; 4 half-tiles in flight
    __builtin_amdgcn_s_barrier();
    IVIT_STAMP(1);
    // waves 4-7 start one barrier interval late; waves 0-3 make it up after the loop
    const bool late = wave >= 4;
    if (late) __builtin_amdgcn_s_barrier();

    bf16x8 a[4][2], b0[2][2], b1[2][2];
    for (int t = 0; t < nt; ++t) g256s_ktile<DBG, FP8, OP>(c, t, last_kt, acc, a, b0, b1);

    if (!late) __builtin_amdgcn_s_barrier();
    IVIT_STAMP(2);
    gemm_epilogue_family<T, EK, OP, IVIT_RSG_256>(p, acc, m0 + wr * 128, n0 + wc * 64, fr, fq, tile_stats + wr * 128);
    IVIT_STAMP(3);
    IVIT_VMCNT(0);   // the clamped tail stagings may still be writing LDS
    IVIT_STAMP(4);
}

}  // namespace ivit
