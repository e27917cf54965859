// 256 x 128 x 64 MFMA GEMM, PERSISTENT workgroups, the epilogue of output tile i interleaved into the main loop of
// tile i+1 (gfx950).  For the wide 16-bit-output GEMMs of an encoder layer (QKV projection, MLP up + GELU, with or
// without the LayerNorm fold), whose grids have several rounds of tiles.
//
// Why another kernel (measured, tools/gemm_bench per-CU stamps, round 2): the 160 x 128 kernels already overlap one
// workgroup's epilogue with the co-resident workgroup's main loop (55-85 % of epilogue time), but their main loops sit
// on the L2 -> LDS path (14 operand bytes per KFLOP: ~1.45 PFLOP/s in-loop at the ~20 TB/s the chip delivers), and the
// 256 x 256 kernel - which has the operand intensity - runs one workgroup per CU and exposes 1.5 us of prologue and
// 5-13 us of epilogue per 16 us of main loop.  Here ONE workgroup per CU keeps the DMA ring running across output
// tiles (no per-tile prologue), holds TWO accumulator sets (64 + 64 registers per lane: the wave tile is 64 x 64), and
// while the MFMAs fill one set the VALU / store work of the previous tile's epilogue drains the other, one fragment
// row per K-step, in the same instruction stream.  The per-tile epilogue operands (bias / LayerNorm-fold vectors, row
// statistics) reach LDS by DMA or by loads that are waited for with the same counted s_waitcnt as the operand ring,
// so nothing in the steady state ever drains the vector-memory queue.
//
//   tile        256 (M) x 128 (N) x 64 (K-step), 8 waves = 4 (M) x 2 (N), wave tile 64 x 64 = 4 x 4 MFMA fragments
//   LDS         3-stage ring x (A 32 KiB + W 16 KiB) = 144 KiB; per-tile vectors c | s (1 KiB) and row statistics
//               (2 KiB), double buffered; Chan-update constants 256 B
//   K-step      s_waitcnt vmcnt(N) -> s_barrier -> issue the DMA of K-step +2 (6 x 1 KiB per wave) -> 16 ds_read_b128,
//               32 MFMA 16x16x32 (+ one epilogue slice of the previous tile in K-steps 0..3: 16 outputs per lane,
//               two 16-byte stores)
//   schedule    workgroup b (grid = min(tiles, CUs)) walks the tiles of its XCD's contiguous share of the XCD-aware
//               tile order (gemm_kernel.h: xcd_tile / tile_coords), 32 apart, so the workgroups of an XCD sweep
//               neighbouring tiles together exactly as the one-tile-per-workgroup kernels do.
//
// Results are bit-identical to the other tile shapes (same MFMA, same K order, same epilogue expressions, the same
// slot-ordered Chan fold of the row statistics): the batch-independence tests compare across them.
#pragma once
#include "../gemm_kernel.h"

namespace ivit {

#ifndef IVIT_PE_THIN
#define IVIT_PE_THIN 0   // 1: the first four K-steps of a tile in the thin form (16-MFMA intervals, 32 fragment registers)
#endif

struct TilePE {
    static constexpr bool RAGGED_N = true;   // generic element-guarded edge epilogue
    static constexpr int BM = 256, BN = 128, WAVES = 8, THREADS = 512;
    static constexpr int A_BYTES = BM * 128, W_BYTES = BN * 128;
    static constexpr int STAGE_BYTES = A_BYTES + W_BYTES;        // 48 KiB
    static constexpr int STAGES = 3;
    static constexpr int RING_BYTES = STAGES * STAGE_BYTES;      // 144 KiB
    static constexpr int VEC_BYTES = 2 * BN * 4;                 // c[128] | s[128]
    static constexpr int STAT_BYTES = BM * 8;                    // (mean, rstd)[256]
    static constexpr int TILEBUF_BYTES = VEC_BYTES + STAT_BYTES; // 3 KiB, x 2
    static constexpr int CHAN_BYTES = 2 * GEMM_LN_SLOTS * 4;     // 1 / (s + 1), 64 s / (s + 1)
    static constexpr int LDS_BYTES = RING_BYTES + 2 * TILEBUF_BYTES + CHAN_BYTES;
    static constexpr int DMA_PER_STAGE = 6;                      // LDS-DMA instructions per wave and K-step
    static constexpr int MIN_KT = 8;                             // K >= 512: the window + the statistics pipeline fit a tile
};

// s_waitcnt vmcnt(n) that also "defines" the registers an earlier inline-asm load filled: the compiler cannot move a use
// of them above the wait (it does not know the asm loads; the dependency through "+v" is what orders them)
#define IVIT_VMCNT_REGS(n, r0, r1) asm volatile("s_waitcnt vmcnt(" #n ")" : "+v"(r0), "+v"(r1) :: "memory")

__device__ __forceinline__ void pe_asm_load16(f32x4& dst, const void* addr) {
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(addr) : "memory");
}

struct PeCtx {
    const GemmParams* p;
    char* smem;
    int lane, wave, fr, fq;
    int a_rd[2], w_rd[2];          // per-lane fragment read offsets inside a stage (A part / W part), kk = 0, 1
    // DMA cursor: the K-step that the next staging call loads
    const char* a_src; const char* w_src;   // this lane's source addresses for piece 0 of the cursor's tile, K-step 0
    const char* na_src; const char* nw_src; // the same for the tile after the cursor's (set at the top of every tile)
    int d_kt;                      // K-step inside the cursor's tile
    int slot;                      // ring slot of the K-step being multiplied (the DMA goes two slots ahead)
    bool stamp_now;                // microbenchmark builds: this K-step is the one being time-stamped
    size_t lda_b, ldw_b;
    int nt, tiles_m, tiles_n;
    int first_tile, tile_step, n_my;   // this workgroup's tiles: first_tile + k * tile_step, k < n_my (indices into the XCD-aware linear order)
};

__device__ __forceinline__ void pe_tile_origin(const PeCtx& c, int k, int& m0, int& n0) {
    int tm, tn;
    tile_coords(c.first_tile + k * c.tile_step, c.tiles_m, c.tiles_n, tm, tn);
    m0 = tm * TilePE::BM; n0 = tn * TilePE::BN;
}

// this lane's DMA source addresses (A, W) for tile k of the walk, K-step 0, piece 0
__device__ __forceinline__ void pe_tile_sources(const PeCtx& c, int k, const char*& a, const char*& w) {
    int m0, n0;
    pe_tile_origin(c, k, m0, n0);
    const int r_in = c.lane >> 3, chunk = (c.lane & 7) ^ r_in;
    a = reinterpret_cast<const char*>(c.p->A) + (size_t)(m0 + c.wave * 8 + r_in) * c.lda_b + chunk * 16;
    w = reinterpret_cast<const char*>(c.p->W) + (size_t)(n0 + c.wave * 8 + r_in) * c.ldw_b + chunk * 16;
}

// stage the cursor's K-step into ring slot `slot` and advance the cursor; at the end of a tile it moves on to the sources
// prepared for the next one (after the last tile those are the last tile's own: the surplus stagings re-load valid memory
// into a slot nobody reads any more, so the K-step body is branch-free and every count constant)
__device__ __forceinline__ void pe_stage(PeCtx& c, int slot) {
#ifdef IVIT_GEMM_ABLATIONS
    if (c.p->debug == 1) return;   // timing ablation: no operand DMA in the loop
#endif
    char* dst = c.smem + slot * TilePE::STAGE_BYTES;
    const char* a = c.a_src + (size_t)c.d_kt * 128;
    const char* w = c.w_src + (size_t)c.d_kt * 128;
#pragma unroll
    for (int i = 0; i < 4; ++i)    // A: pieces wave, wave + 8, +16, +24 (8 rows each)
        __builtin_amdgcn_global_load_lds((const IVIT_GLOBAL void*)(a + (size_t)(i * 64) * c.lda_b), (IVIT_LDS void*)(dst + (c.wave + 8 * i) * 1024), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < 2; ++i)    // W: pieces wave, wave + 8
        __builtin_amdgcn_global_load_lds((const IVIT_GLOBAL void*)(w + (size_t)(i * 64) * c.ldw_b),
                                         (IVIT_LDS void*)(dst + TilePE::A_BYTES + (c.wave + 8 * i) * 1024), 16, 0, 0);
    const bool wrap = c.d_kt + 1 == c.nt;
    c.d_kt = wrap ? 0 : c.d_kt + 1;
    c.a_src = wrap ? c.na_src : c.a_src;
    c.w_src = wrap ? c.nw_src : c.w_src;
}

// fragment reads of one 32-deep half (kk) of a K-step: 4 W + 4 A fragments
__device__ __forceinline__ void pe_read(const PeCtx& c, int slot, int kk, bf16x8 (&af)[4], bf16x8 (&wf)[4]) {
    const char* a_tile = c.smem + slot * TilePE::STAGE_BYTES;
    const char* w_tile = a_tile + TilePE::A_BYTES;
#pragma unroll
    for (int j = 0; j < 4; ++j) wf[j] = *reinterpret_cast<const bf16x8*>(w_tile + c.w_rd[kk] + j * 2048);
#pragma unroll
    for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<const bf16x8*>(a_tile + c.a_rd[kk] + i * 2048);
}

template <class OP>
__device__ __forceinline__ void pe_mma16(f32x4 (&acc)[4][4], const bf16x8 (&af)[4], const bf16x8 (&wf)[4]) {

#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = OP::mfma(wf[j], af[i], acc[i][j]);
}

// Epilogue UNIT U (0..7) of the finished tile (origin pm0, pn0; its vectors / statistics in tile buffer `tb`): fragment row
// I = U / 2, fragment pair J = 2 (U % 2): v = rstd * (acc - mean * s) + c, optional GELU, 16-bit, ONE 16-byte store per lane
// (rows beyond M land in the padding rows of the output workspace).  Same expressions as gemm_epilogue_lnfold / _impl.
template <class OP, bool GELU, int U>
__device__ __forceinline__ void pe_epilogue_unit(const PeCtx& c, const f32x4 (&acc)[4][4], int pm0, int pn0, const char* tb) {
    constexpr int I = U >> 1, J = (U & 1) * 2;
    const int wr = c.wave >> 1, wc = c.wave & 1;
    const int row_local = wr * 64 + I * 16 + c.fr;
    // The tile buffer is read by inline asm: for an ordinary LDS load next to in-flight LDS-DMA hipcc emits s_waitcnt vmcnt(0)
    // (it cannot tell that the DMA targets other bytes), which would drain the operand ring in every K-step of the window.
    // One block = 5 reads + their wait, so the compiler's own lgkmcnt bookkeeping never sees them outstanding.
    typedef __attribute__((ext_vector_type(2))) float f32x2;
    f32x2 st;
    f32x4 c0, c1, s0, s1;
    {
        const unsigned tb_lds = (unsigned)(size_t)(IVIT_LDS const char*)tb;
        const unsigned a_st = tb_lds + TilePE::VEC_BYTES + row_local * 8;
        const unsigned a_c = tb_lds + (wc * 64 + J * 16 + c.fq * 4) * 4;
        asm volatile(
            "ds_read_b64 %0, %5\n\t"
            "ds_read_b128 %1, %6\n\t"
            "ds_read_b128 %2, %6 offset:64\n\t"
            "ds_read_b128 %3, %6 offset:512\n\t"
            "ds_read_b128 %4, %6 offset:576\n\t"
            "s_waitcnt lgkmcnt(0)"
            : "=&v"(st), "=&v"(c0), "=&v"(c1), "=&v"(s0), "=&v"(s1)
            : "v"(a_st), "v"(a_c)
            : "memory");
    }
    const float mu = st[0], rs = st[1];
    bf16_t* orow_p = reinterpret_cast<bf16_t*>(c.p->out) + (size_t)(pm0 + row_local) * c.p->ldo + pn0 + wc * 64;
    float a[4] = {fmaf(rs, fmaf(-mu, s0[0], acc[I][J][0]), c0[0]), fmaf(rs, fmaf(-mu, s0[1], acc[I][J][1]), c0[1]),
                  fmaf(rs, fmaf(-mu, s0[2], acc[I][J][2]), c0[2]), fmaf(rs, fmaf(-mu, s0[3], acc[I][J][3]), c0[3])};
    float b[4] = {fmaf(rs, fmaf(-mu, s1[0], acc[I][J + 1][0]), c1[0]), fmaf(rs, fmaf(-mu, s1[1], acc[I][J + 1][1]), c1[1]),
                  fmaf(rs, fmaf(-mu, s1[2], acc[I][J + 1][2]), c1[2]), fmaf(rs, fmaf(-mu, s1[3], acc[I][J + 1][3]), c1[3])};
    if (GELU) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { a[r] = gelu_erf(a[r]); b[r] = gelu_erf(b[r]); }
    }
    const auto lo = __builtin_amdgcn_permlane16_swap(OP::pack2(a[0], a[1]), OP::pack2(b[0], b[1]), false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap(OP::pack2(a[2], a[3]), OP::pack2(b[2], b[3]), false, false);
    const int n = (J + (c.fq & 1)) * 16 + (c.fq & ~1) * 4;
    u32x4 pk = {lo[0], hi[0], lo[1], hi[1]};
    *reinterpret_cast<u32x4*>(orow_p + n) = pk;
}

// DMA of the tile's epilogue vectors: c (= bias) and s, 128 floats each, into tile buffer `tb` - ONE 256-byte LDS-DMA per
// wave (waves 4-7 repeat what waves 0-3 load: every wave must issue the same number of vector-memory operations)
__device__ __forceinline__ void pe_stage_vectors(const PeCtx& c, int n0, char* tb) {
    const int part = c.wave & 3;
    const float* src = ((part < 2) ? c.p->bias : c.p->ln_s) + n0 + (part & 1) * 64 + c.lane;
    __builtin_amdgcn_global_load_lds((const IVIT_GLOBAL void*)src, (IVIT_LDS void*)(tb + part * 256), 4, 0, 0);
}

// Row statistics of a tile's rows, folded from the (sum, M2) pairs of its 64-column slots in slot order (gemm_kernel.h:
// ln_tile_stats - the same arithmetic, bit for bit), four slots per K-step: the loads of a chunk are issued two K-steps
// before it is folded and are retired by the K-loop's own counted waits.
struct PeStats {
    float mean, m2;
    const char* src;     // this thread's row of ln_part
};

__device__ __forceinline__ void pe_stats_issue(const PeCtx& c, const PeStats& s, int chunk, int nchunks, f32x4& r0, f32x4& r1) {
    // beyond the last chunk (and for GEMMs without a LayerNorm): the same number of loads, of a valid address
    const char* a = (chunk < nchunks) ? s.src + chunk * 32 : reinterpret_cast<const char*>(c.p->bias);
    pe_asm_load16(r0, a);
    pe_asm_load16(r1, (chunk < nchunks) ? a + 16 : a);
}

__device__ __forceinline__ void pe_stats_fold(const PeCtx& c, PeStats& s, int chunk, int nslots, const f32x4& r0, const f32x4& r1, const float* chan) {
    constexpr float inv64 = 1.0f / 64.0f;
    const float sm[4] = {r0[0], r0[2], r1[0], r1[2]}, mm[4] = {r0[1], r0[3], r1[1], r1[3]};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int slot = chunk * 4 + q;
        if (slot < nslots) {   // uniform
            const float d = fmaf(sm[q], inv64, -s.mean);
            s.mean = fmaf(d, chan[slot], s.mean);
            s.m2 = fmaf(d * d, chan[GEMM_LN_SLOTS + slot], s.m2 + mm[q]);
        }
    }
}

#ifdef IVIT_GEMM_ABLATIONS   // shader-clock stamps of ONE K-step per workgroup (tools/gemm_bench): slots wave * 16 + i of the block's 128-slot record
#define PE_STAMP(i)                                                                                           \
    do {                                                                                                      \
        if (c.stamp_now && c.lane == 0) {                                                                     \
            unsigned long long t_;                                                                            \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                         \
            c.p->stamps[(size_t)blockIdx.x * 128 + c.wave * 16 + (i)] = t_;                                   \
        }                                                                                                     \
    } while (0)
#else
#define PE_STAMP(i) do { } while (0)
#endif

__device__ __forceinline__ int pe_next_slot(int s) { return s == 2 ? 0 : s + 1; }
__device__ __forceinline__ int pe_write_slot(int s) { return s == 0 ? 2 : s - 1; }   // (s + 2) % 3

// A K-step runs in barrier intervals that alternate "SR" (LDS reads, DMA issue, counted wait) and "M" (MFMAs + the VALU
// work of the previous tile's epilogue), and waves 4-7, which share their SIMDs with waves 0-3, run ONE INTERVAL BEHIND
// them (they enter the loop through one extra barrier; waves 0-3 make it up at the end): at any time one wave of a SIMD is
// in an M interval while its partner is in an SR interval.  (Without the stagger all eight waves issued their DMA
// together and then their MFMAs together: 1.27 us per K-step against 0.51 us of MFMA work.)
//
//   thin K-step (the first four of a tile, which carry the epilogue units: 32 fragment registers, `prev` is live)
//      SR0: DMA of K-step t+2, ds_read kk = 0      M0: 16 MFMAs + unit U0
//      SR1: ds_read kk = 1, counted wait (t+1)     M1: 16 MFMAs + unit U1
//   fat K-step (the rest: `prev` is dead, so all 64 fragment registers of a K-step fit)
//      SR : DMA of K-step t+2, ds_read kk = 0, 1, statistics loads, counted wait (t+1)      M: 32 MFMAs (+ statistics fold)
// An M interval is one wave's MFMAs only, so the matrix pipe is busy (16 or 32) x 16 cycles per interval of that + ~120
// cycles of barrier / wait overhead: the fat form is what makes the loop worth having (stamps: tools/gemm_bench).
//
// Hazards (barrier k opens interval k; a group's SR intervals end with s_waitcnt lgkmcnt(0), so its LDS reads are retired
// before the barrier that closes the interval):
//   RAW  stage t+1 is first read in the first SR interval of K-step t+1.  Every wave waits for ITS pieces of it at the end of
//        its last SR interval of K-step t, which closes at least one barrier earlier for either group.
//   WAR  the DMA issued at the start of K-step t overwrites the slot of stage t-1, last read in the last SR interval of
//        K-step t-1 - for the other group that is the interval just before, retired before the barrier in between.
// The DMA of stage t+2 is issued at the START of K-step t and waited for at the END of K-step t+1's last SR interval:
// ~1.5 K-steps of lead, two stages in flight (issuing it next to the wait for stage t+1 left one stage in flight and the
// loop bound to one DMA round trip of ~2900 cycles per K-step).
//
// WAIT: vmcnt = every vector-memory operation this wave issued after the DMA of stage t+1 (K-step t-1's vector DMA, unit
// stores and statistics loads, this K-step's 6 DMA, vector DMA, first unit store and statistics loads) - minus, when a
// statistics chunk is folded in this K-step, everything up to and including its loads (they must be retired).
template <class OP, bool GELU, int U0, int U1, bool VEC, int WAIT>
__device__ __forceinline__ void pe_kstep_thin(PeCtx& c, f32x4 (&acc)[4][4], const f32x4 (&prev)[4][4], int pm0, int pn0, const char* tb_prev, int n0,
                                              char* tb_cur) {
    bf16x8 af[4], wf[4];
    // ---- SR0
    PE_STAMP(0);
    pe_stage(c, pe_write_slot(c.slot));
    if (VEC) pe_stage_vectors(c, n0, tb_cur);
    pe_read(c, c.slot, 0, af, wf);
    PE_STAMP(1);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    PE_STAMP(2);
    // ---- M0
    if (U0 >= 0) pe_epilogue_unit<OP, GELU, (U0 >= 0 ? U0 : 0)>(c, prev, pm0, pn0, tb_prev);
    pe_mma16<OP>(acc, af, wf);
    __builtin_amdgcn_sched_barrier(0);
    PE_STAMP(3);
    __builtin_amdgcn_s_barrier();
    PE_STAMP(4);
    // ---- SR1
    pe_read(c, c.slot, 1, af, wf);
    PE_STAMP(5);
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(WAIT) : "memory");
    PE_STAMP(6);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    PE_STAMP(7);
    // ---- M1
    if (U1 >= 0) pe_epilogue_unit<OP, GELU, (U1 >= 0 ? U1 : 0)>(c, prev, pm0, pn0, tb_prev);
    pe_mma16<OP>(acc, af, wf);
    __builtin_amdgcn_sched_barrier(0);
    PE_STAMP(8);
    __builtin_amdgcn_s_barrier();
    PE_STAMP(9);
    c.slot = pe_next_slot(c.slot);
}

// LOADS: the SR interval loads a statistics chunk into (l0, l1); FOLD: the wait retires the chunk in (f0, f1) - loaded in the
// previous K-step - and the M interval folds it.
template <class OP, int WAIT, bool LOADS, bool FOLD>
__device__ __forceinline__ void pe_kstep_fat(PeCtx& c, f32x4 (&acc)[4][4], PeStats& st, int load_chunk, int fold_chunk, int nchunks, int nslots,
                                             f32x4& l0, f32x4& l1, f32x4& f0, f32x4& f1, const float* chan) {
    bf16x8 af0[4], wf0[4], af1[4], wf1[4];
    // ---- SR
    PE_STAMP(0);
    pe_stage(c, pe_write_slot(c.slot));
    pe_read(c, c.slot, 0, af0, wf0);
    pe_read(c, c.slot, 1, af1, wf1);
    if (LOADS) pe_stats_issue(c, st, load_chunk, nchunks, l0, l1);
    PE_STAMP(1);
    if (FOLD) asm volatile("s_waitcnt vmcnt(%2) lgkmcnt(0)" : "+v"(f0), "+v"(f1) : "n"(WAIT) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(WAIT) : "memory");
    PE_STAMP(2);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    PE_STAMP(3);
    // ---- M
    if (FOLD) { if (fold_chunk < nchunks) pe_stats_fold(c, st, fold_chunk, nslots, f0, f1, chan); }
    pe_mma16<OP>(acc, af0, wf0);
    pe_mma16<OP>(acc, af1, wf1);
    __builtin_amdgcn_sched_barrier(0);
    PE_STAMP(4);
    __builtin_amdgcn_s_barrier();
    PE_STAMP(5);
    c.slot = pe_next_slot(c.slot);
}

// the fat K-step that also carries two epilogue units of the previous tile in its M interval (K-steps 0..3 of a tile)
template <class OP, bool GELU, int U0, int U1, bool VEC, int WAIT>
__device__ __forceinline__ void pe_kstep_fat_epi(PeCtx& c, f32x4 (&acc)[4][4], const f32x4 (&prev)[4][4], int pm0, int pn0, const char* tb_prev, int n0,
                                                 char* tb_cur) {
    bf16x8 af0[4], wf0[4], af1[4], wf1[4];
    pe_stage(c, pe_write_slot(c.slot));
    if (VEC) pe_stage_vectors(c, n0, tb_cur);
    pe_read(c, c.slot, 0, af0, wf0);
    pe_read(c, c.slot, 1, af1, wf1);
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(WAIT) : "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (U0 >= 0) pe_epilogue_unit<OP, GELU, (U0 >= 0 ? U0 : 0)>(c, prev, pm0, pn0, tb_prev);
    pe_mma16<OP>(acc, af0, wf0);
    if (U1 >= 0) pe_epilogue_unit<OP, GELU, (U1 >= 0 ? U1 : 0)>(c, prev, pm0, pn0, tb_prev);
    pe_mma16<OP>(acc, af1, wf1);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    c.slot = pe_next_slot(c.slot);
}

template <class OP, bool GELU>
__device__ __forceinline__ void gemmpe_body(const GemmParams& p, char* smem) {
    using T = TilePE;
    PeCtx c;
    c.p = &p; c.smem = smem;
    c.lane = threadIdx.x & 63;
    c.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    c.fr = c.lane & 15; c.fq = c.lane >> 4;
    const int wr = c.wave >> 1, wc = c.wave & 1;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        c.a_rd[kk] = (wr * 64 + c.fr) * 128 + (((kk * 4 + c.fq) ^ (c.fr & 7)) << 4);
        c.w_rd[kk] = (wc * 64 + c.fr) * 128 + (((kk * 4 + c.fq) ^ (c.fr & 7)) << 4);
    }
    c.lda_b = (size_t)p.lda * 2; c.ldw_b = (size_t)p.ldw * 2;
    c.nt = p.K / GEMM_BK;
    c.tiles_m = ceil_div(p.M, T::BM); c.tiles_n = p.N / T::BN;
    {   // this workgroup's walk: the tiles of its XCD's contiguous share, count_x apart (count_x workgroups share the XCD)
        const int tiles = c.tiles_m * c.tiles_n, G = gridDim.x, b = blockIdx.x;
        const int x = b & 7, l = b >> 3;
        const int count_x = (G - x + 7) >> 3;
        const int qd = tiles >> 3, rm = tiles & 7;
        const int start = (x < rm ? x * (qd + 1) : rm * (qd + 1) + (x - rm) * qd), len = qd + (x < rm ? 1 : 0);
        c.first_tile = start + l; c.tile_step = count_x;
        c.n_my = (l < len) ? (len - l + count_x - 1) / count_x : 0;
    }
    if (c.n_my == 0) return;

    char* tb0 = smem + T::RING_BYTES;                    // tile buffers 0 / 1, then the Chan constants
    float* chan = reinterpret_cast<float*>(smem + T::RING_BYTES + 2 * T::TILEBUF_BYTES);
    if (threadIdx.x < GEMM_LN_SLOTS) {                   // 1 / (s + 1) and 64 s / (s + 1) as ln_tile_stats' compile-time constants (IEEE divisions)
        const int s2 = threadIdx.x;
        chan[s2] = 1.0f / (float)(s2 + 1);
        chan[GEMM_LN_SLOTS + s2] = 64.0f * (float)s2 / (float)(s2 + 1);
    }
    __syncthreads();
    const bool ln = p.ln_part != nullptr;
    const int nslots = ln ? (p.ln_dim + 63) >> 6 : 0;
    const int nchunks = (nslots + 3) >> 2;
    const int nch = max(2, (nchunks + 1) & ~1);          // chunks walked: even and >= 2, so the register sets alternate with compile-time parity
    const int row_t = threadIdx.x & 255;                 // waves 4-7 repeat waves 0-3 (uniform vector-memory counts)

    f32x4 acc[4][4], prev[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) { acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; prev[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; }

    pe_tile_sources(c, 0, c.a_src, c.w_src);
    pe_tile_sources(c, c.n_my > 1 ? 1 : 0, c.na_src, c.nw_src);
    c.d_kt = 0;
    c.slot = 0;
    c.stamp_now = false;
#ifdef IVIT_GEMM_ABLATIONS
    if (p.stamps && threadIdx.x == 0)   // which XCD runs this block (is it blockIdx % 8 ?)
        p.stamps[(size_t)blockIdx.x * 128 + 15] = 1 + (__builtin_amdgcn_s_getreg((3 << 11) | 20) & 15);
#endif
    pe_stage(c, 0);
    pe_stage(c, 1);
    IVIT_VMCNT(6);                                       // K-step 0 landed for this wave
    __builtin_amdgcn_s_barrier();
    const bool late = c.wave >= 4;
    if (late) __builtin_amdgcn_s_barrier();              // waves 4-7 start one interval late

    // Statistics pipeline of a tile: chunk j (4 slots = two 16-byte loads) is loaded in the SR interval of K-step 4 + j and
    // folded in the M interval of K-step 5 + j.  Vector-memory operations besides the 6 operand DMA of a K-step: a statistics
    // chunk 2 loads, an epilogue unit 1 store, the vector DMA 1.
    int pm0 = 0, pn0 = 0;
    for (int k = 0; k < c.n_my; ++k) {
        int m0, n0;
        pe_tile_origin(c, k, m0, n0);
        pe_tile_sources(c, min(k + 1, c.n_my - 1), c.na_src, c.nw_src);   // where the DMA cursor goes when it runs off this tile (K-steps nt-2, nt-1)
        char* tb_cur = tb0 + (k & 1) * T::TILEBUF_BYTES;
        const char* tb_prev = tb0 + ((k & 1) ^ 1) * T::TILEBUF_BYTES;
        PeStats st;
        st.mean = 0.f; st.m2 = 0.f;
        st.src = reinterpret_cast<const char*>(p.ln_part) + (size_t)min(m0 + row_t, p.M - 1) * (GEMM_LN_SLOTS * 8);
        f32x4 ra0 = {0.f, 0.f, 0.f, 0.f}, ra1 = ra0, rb0 = ra0, rb1 = ra0;   // two chunks of statistics loads in flight (sets a / b)
#define PE_THIN c, acc, prev, pm0, pn0, tb_prev, n0, tb_cur
#define PE_FAT c, acc, st
        if (IVIT_PE_THIN) {
        if (k == 0) {   // first tile of the workgroup: nothing to drain
            pe_kstep_thin<OP, GELU, -1, -1, true, 7>(PE_THIN);     // 6 + the vector DMA
            pe_kstep_thin<OP, GELU, -1, -1, false, 7>(PE_THIN);    // vector DMA + 6
            pe_kstep_thin<OP, GELU, -1, -1, false, 6>(PE_THIN);
            pe_kstep_thin<OP, GELU, -1, -1, false, 6>(PE_THIN);
            pe_kstep_fat<OP, 8, true, false>(PE_FAT, 0, 0, nchunks, nslots, ra0, ra1, ra0, ra1, chan);                    // K-step 4: chunk 0 -> a; 6 + 2
        } else {        // steady state: the previous tile's epilogue, one unit per M interval
            pe_kstep_thin<OP, GELU, 0, 1, true, 8>(PE_THIN);       // 6 + vector DMA + store of M0
            pe_kstep_thin<OP, GELU, 2, 3, false, 10>(PE_THIN);     // vector DMA + 2 stores, 6 + store of M0
            pe_kstep_thin<OP, GELU, 4, 5, false, 9>(PE_THIN);      // 2 stores, 6 + 1
            pe_kstep_thin<OP, GELU, 6, 7, false, 9>(PE_THIN);
            pe_kstep_fat<OP, 10, true, false>(PE_FAT, 0, 0, nchunks, nslots, ra0, ra1, ra0, ra1, chan);                   // K-step 4: 2 stores, 6 + 2 loads
        }
        } else {
        // all K-steps fat; the first four carry two epilogue units each in their M interval (both stores after the DMA and the wait)
        if (k == 0) {
            pe_kstep_fat_epi<OP, GELU, -1, -1, true, 7>(PE_THIN);     // 6 + the vector DMA
            pe_kstep_fat_epi<OP, GELU, -1, -1, false, 7>(PE_THIN);    // vector DMA + 6
            pe_kstep_fat_epi<OP, GELU, -1, -1, false, 6>(PE_THIN);
            pe_kstep_fat_epi<OP, GELU, -1, -1, false, 6>(PE_THIN);
            pe_kstep_fat<OP, 8, true, false>(PE_FAT, 0, 0, nchunks, nslots, ra0, ra1, ra0, ra1, chan);                    // K-step 4: chunk 0 -> a; 6 + 2
        } else {
            pe_kstep_fat_epi<OP, GELU, 0, 1, true, 7>(PE_THIN);       // 6 + vector DMA
            pe_kstep_fat_epi<OP, GELU, 2, 3, false, 9>(PE_THIN);      // vector DMA + 2 stores + 6
            pe_kstep_fat_epi<OP, GELU, 4, 5, false, 8>(PE_THIN);      // 2 stores + 6
            pe_kstep_fat_epi<OP, GELU, 6, 7, false, 8>(PE_THIN);
            pe_kstep_fat<OP, 10, true, false>(PE_FAT, 0, 0, nchunks, nslots, ra0, ra1, ra0, ra1, chan);                   // K-step 4: 2 stores + 6 + 2 loads
        }
        }
        // K-steps 5 .. 3 + nch: load chunk j + 1, fold chunk j (the wait leaves this K-step's 6 DMA and 2 loads outstanding)
        int t = 5;
        for (int j = 0; j + 2 < nch; j += 2, t += 2) {
            pe_kstep_fat<OP, 8, true, true>(PE_FAT, j + 1, j, nchunks, nslots, rb0, rb1, ra0, ra1, chan);
            pe_kstep_fat<OP, 8, true, true>(PE_FAT, j + 2, j + 1, nchunks, nslots, ra0, ra1, rb0, rb1, chan);
        }
        pe_kstep_fat<OP, 8, true, true>(PE_FAT, nch - 1, nch - 2, nchunks, nslots, rb0, rb1, ra0, ra1, chan);              // K-step 3 + nch: the last load
        pe_kstep_fat<OP, 6, false, true>(PE_FAT, 0, nch - 1, nchunks, nslots, ra0, ra1, rb0, rb1, chan);                   // K-step 4 + nch: the last fold
        t += 2;
        for (; t < c.nt; ++t) {                           // the rest of the tile: operand ring only
#ifdef IVIT_GEMM_ABLATIONS
            c.stamp_now = p.stamps && k == 1 && t == c.nt - 2;
#endif
            pe_kstep_fat<OP, 6, false, false>(PE_FAT, 0, 0, nchunks, nslots, ra0, ra1, ra0, ra1, chan);
        }
#ifdef IVIT_GEMM_ABLATIONS
        c.stamp_now = false;
#endif
#undef PE_THIN
#undef PE_FAT
        // statistics of this tile's rows -> its tile buffer (read by the epilogue units during the NEXT tile's K-steps,
        // behind several barriers)
        {
            float2 fin = make_float2(0.f, 1.f);          // no LayerNorm: rstd (acc - 0 s) + c = acc + c exactly
            if (ln) fin = make_float2(st.mean, 1.0f / sqrtf(st.m2 / (float)p.ln_dim + p.ln_eps));
            // (inline asm for the same reason as the reads of pe_epilogue_unit: an ordinary LDS store next to in-flight LDS-DMA
            // is preceded by s_waitcnt vmcnt(0))
            typedef __attribute__((ext_vector_type(2))) float f32x2;
            const f32x2 fv = {fin.x, fin.y};
            const unsigned dst = (unsigned)(size_t)(IVIT_LDS char*)(tb_cur + T::VEC_BYTES + row_t * 8);
            asm volatile("ds_write_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" :: "v"(dst), "v"(fv) : "memory");
        }
        pm0 = m0; pn0 = n0;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) { prev[i][j] = acc[i][j]; acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    }
    if (!late) __builtin_amdgcn_s_barrier();
    // ---- the last tile's epilogue has no main loop to hide in
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the surplus stagings past the end still write LDS
    __syncthreads();
    {
        const char* tb_last = tb0 + ((c.n_my - 1) & 1) * T::TILEBUF_BYTES;
        pe_epilogue_unit<OP, GELU, 0>(c, prev, pm0, pn0, tb_last);
        pe_epilogue_unit<OP, GELU, 1>(c, prev, pm0, pn0, tb_last);
        pe_epilogue_unit<OP, GELU, 2>(c, prev, pm0, pn0, tb_last);
        pe_epilogue_unit<OP, GELU, 3>(c, prev, pm0, pn0, tb_last);
        pe_epilogue_unit<OP, GELU, 4>(c, prev, pm0, pn0, tb_last);
        pe_epilogue_unit<OP, GELU, 5>(c, prev, pm0, pn0, tb_last);
        pe_epilogue_unit<OP, GELU, 6>(c, prev, pm0, pn0, tb_last);
        pe_epilogue_unit<OP, GELU, 7>(c, prev, pm0, pn0, tb_last);
    }
}

}  // namespace ivit
