// Persistent 256 x 256 x 64 bf16 MFMA GEMM (gfx950): the staggered two-group schedule of
// gemm256s_kernel.h, one workgroup per CU walking a list of output tiles, with the LDS-DMA stream
// kept CONTINUOUS across output tiles.
//
// Why: the per-launch-tile version pays ~17 us of fixed cost per tile (workgroup launch, a cold
// DMA prologue, an epilogue nothing overlaps) against ~16 us of main loop at K = 768.  Here the K-tiles
// of all the workgroup's output tiles form one sequence g = 0, 1, 2, ...; the staging rules
// "SR1: B1(g+1), SR2: A1(g+1), SR3: A0(g+2), SR4: B0(g+2)" simply run on across the tile boundary,
// so the first K-tiles of output tile i+1 are loading while tile i finishes and while its epilogue
// stores drain.  The epilogue (bias / GELU / residual / remap, shared with the other GEMM kernels)
// runs between the last M4 of a tile and the first SR1 of the next; the two wave groups are brought
// level for it and re-staggered afterwards (one extra barrier each per tile).  s_waitcnt vmcnt(8) stays valid with epilogue stores
// in the queue (vmcnt counts stores too): it is then merely conservative.
// LDS buffer sets alternate with the running K-tile counter g (not the tile-local index), so an odd
// K-tile count per output tile is fine.  Requires K >= 128 (two K-tiles).
#pragma once
#include "../gemm256s_kernel.h"
#include "gemm256p_kernel.h"

namespace ivit {

struct G256PCtx {
    const bf16_t *a_cur, *w_cur;   // this lane's DMA source (half 0, piece 0, k = 0) in the current output tile
    const bf16_t *a_nxt, *w_nxt;   // same in the workgroup's next output tile (== current when there is none)
    int lda, ldw, nt;
    char* smem;
    int wave;
    int a_rd[2], b_rd[2];
};

// stage half-tile h of K-tile `kt` (tile-local index; >= nt means "the next output tile") into the LDS
// set of running K-tile counter `g`
template <bool IS_W>
__device__ __forceinline__ void g256p_stage(const G256PCtx& c, int kt, int g, int h) {
    const bool in_cur = kt < c.nt;
    const int k = in_cur ? kt : kt - c.nt;
    const bf16_t* base = IS_W ? (in_cur ? c.w_cur : c.w_nxt) : (in_cur ? c.a_cur : c.a_nxt);
    const int ld = IS_W ? c.ldw : c.lda;
    char* dst = c.smem + (g & 1) * Tile256P::SET_BYTES + (IS_W ? Tile256P::OFF_B0 : Tile256P::OFF_A0) + h * Tile256P::HALF_BYTES;
    const bf16_t* s = base + (size_t)(h * (IS_W ? 32 : 64)) * ld + k * GEMM_BK;
#pragma unroll
    for (int i = 0; i < 2; ++i)
        __builtin_amdgcn_global_load_lds((const IVIT_GLOBAL void*)(s + (size_t)(i * 128) * ld),
                                         (IVIT_LDS void*)(dst + (c.wave + 8 * i) * 1024), 16, 0, 0);
}

__device__ __forceinline__ void g256p_read_a(const G256PCtx& c, int g, int h, bf16x8 (&f)[4][2]) {
    const char* base = c.smem + (g & 1) * Tile256P::SET_BYTES + Tile256P::OFF_A0 + h * Tile256P::HALF_BYTES;
#pragma unroll
    for (int mf = 0; mf < 4; ++mf)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
            f[mf][kk] = *reinterpret_cast<const bf16x8*>(base + c.a_rd[kk] + mf * 2048);
}
__device__ __forceinline__ void g256p_read_b(const G256PCtx& c, int g, int h, bf16x8 (&f)[2][2]) {
    const char* base = c.smem + (g & 1) * Tile256P::SET_BYTES + Tile256P::OFF_B0 + h * Tile256P::HALF_BYTES;
#pragma unroll
    for (int nf = 0; nf < 2; ++nf)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
            f[nf][kk] = *reinterpret_cast<const bf16x8*>(base + c.b_rd[kk] + nf * 2048);
}

// one K-tile: t = tile-local K index (staging sources), g = running counter (LDS sets)
__device__ __forceinline__ void g256p_ktile(const G256PCtx& c, int t, int g, f32x4 (&acc)[8][4], bf16x8 (&a)[4][2],
                                            bf16x8 (&b0)[2][2], bf16x8 (&b1)[2][2]) {
    // ---- SR1: stage B1(g+1) | read A0, B0
    g256p_stage<true>(c, t + 1, g + 1, 1);
    g256p_read_b(c, g, 0, b0);
    g256p_read_a(c, g, 0, a);
    IVIT_VMCNT(8);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    g256_mma<0, 0>(acc, a, b0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    // ---- SR2: stage A1(g+1) | read B1
    g256p_stage<false>(c, t + 1, g + 1, 1);
    g256p_read_b(c, g, 1, b1);
    IVIT_VMCNT(8);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    g256_mma<0, 1>(acc, a, b1);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    // ---- SR3: stage A0(g+2) | read A1
    g256p_stage<false>(c, t + 2, g + 2, 0);
    g256p_read_a(c, g, 1, a);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    g256_mma<1, 1>(acc, a, b1);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    // ---- SR4: stage B0(g+2)
    g256p_stage<true>(c, t + 2, g + 2, 0);
    IVIT_VMCNT(8);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    g256_mma<1, 0>(acc, a, b0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
}

__device__ __forceinline__ void g256p_tile_src(const GemmParams& p, int tile, int tiles_m, int tiles_n, int wave, int lane,
                                               const bf16_t*& a_src, const bf16_t*& w_src, int& m0, int& n0) {
    int tm, tn;
    tile_coords(tile, tiles_m, tiles_n, tm, tn);
    m0 = tm * Tile256P::BM;
    n0 = tn * Tile256P::BN;
    const int r_in = lane >> 3;
    const int chunk = (lane & 7) ^ r_in;
    a_src = p.A + (size_t)(m0 + wave * 8 + r_in) * p.lda + chunk * 8;
    w_src = p.W + (size_t)(n0 + (wave >> 2) * 64 + (wave & 3) * 8 + r_in) * p.ldw + chunk * 8;
}

__device__ __forceinline__ void gemm256ps_body(const GemmParams& p, char* smem) {
    using T = Tile256P;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int fr = lane & 15, fq = lane >> 4;

    const int tiles_m = ceil_div(p.M, T::BM), tiles_n = ceil_div(p.N, T::BN);
    const int num_tiles = tiles_m * tiles_n;
    const int G = gridDim.x;
    // round r: workgroup b works on tile r*G + xcd_tile(b): the workgroups of one XCD take a
    // contiguous run of the (column-panel grouped) tile order
    int tile = xcd_tile(blockIdx.x, G);
    if (tile >= num_tiles) return;   // never: the host launches G <= num_tiles

    G256PCtx c;
    c.smem = smem; c.wave = wave; c.lda = p.lda; c.ldw = p.ldw; c.nt = p.K / GEMM_BK;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        c.a_rd[kk] = (wr * 64 + fr) * 128 + (((kk * 4 + fq) ^ (fr & 7)) << 4);
        c.b_rd[kk] = (wc * 32 + fr) * 128 + (((kk * 4 + fq) ^ (fr & 7)) << 4);
    }
    int m0, n0, m0n, n0n;
    g256p_tile_src(p, tile, tiles_m, tiles_n, wave, lane, c.a_cur, c.w_cur, m0, n0);
    c.a_nxt = c.a_cur; c.w_nxt = c.w_cur; m0n = m0; n0n = n0;   // placeholders until the loop sets them

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- prologue (first output tile only): K-tile 0 = A0 B0 B1 A1, K-tile 1 = A0 B0
    g256p_stage<false>(c, 0, 0, 0);
    g256p_stage<true>(c, 0, 0, 0);
    g256p_stage<true>(c, 0, 0, 1);
    g256p_stage<false>(c, 0, 0, 1);
    g256p_stage<false>(c, 1, 1, 0);
    g256p_stage<true>(c, 1, 1, 0);
    IVIT_VMCNT(8);
    __builtin_amdgcn_s_barrier();
    const bool late = wave >= 4;     // waves 4-7 run one barrier interval behind waves 0-3

    bf16x8 a[4][2], b0[2][2], b1[2][2];
    int g = 0;
    for (; tile < num_tiles; tile += G) {
        // (re-)establish the stagger: the late group's extra barrier pairs with the early group's
        // end-of-SR1 barrier; at the end of the tile the early group waits one interval so that both
        // groups run their epilogues CONCURRENTLY (left staggered, the barriers serialise them: the
        // late group cannot leave its last M4 before the early group has finished its epilogue)
        if (late) __builtin_amdgcn_s_barrier();
        const int next = tile + G;
        if (next < num_tiles) {
            g256p_tile_src(p, next, tiles_m, tiles_n, wave, lane, c.a_nxt, c.w_nxt, m0n, n0n);
        } else {   // no next tile: the run-ahead stagings re-read this tile's first K-tiles into dead buffers
            c.a_nxt = c.a_cur; c.w_nxt = c.w_cur;
        }
        for (int t = 0; t < c.nt; ++t, ++g) g256p_ktile(c, t, g, acc, a, b0, b1);
        if (!late) __builtin_amdgcn_s_barrier();

        gemm_epilogue<T>(p, acc, m0 + wr * 128, n0 + wc * 64, fr, fq);
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        c.a_cur = c.a_nxt; c.w_cur = c.w_nxt; m0 = m0n; n0 = n0n;
    }
    IVIT_VMCNT(0);   // the last run-ahead stagings may still be writing LDS
}

}  // namespace ivit
