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
    static constexpr bool RAGGED_N = false;
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
    int a_wrap;            // GemmParams::a_wrap
    int a_shift;           // GemmParams::a_shift
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
        int ka = kt >> c.a_shift;
        if (c.a_wrap) { if (ka >= c.a_wrap) ka -= c.a_wrap; if (ka >= c.a_wrap) ka -= c.a_wrap; }
        const bf16_t* s = c.a_src + (size_t)(h * 64) * c.lda + ka * GEMM_BK;
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
template <int HA, int HB, bool FP8 = false, class OP = OpBf16>
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
                        OP::mfma(b[nf][kk], a[mf][kk], acc[HA * 4 + mf][HB * 2 + nf]);
    }
    __builtin_amdgcn_s_setprio(0);
}

}  // namespace ivit
