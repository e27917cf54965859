// Persistent LayerNorm-fold GEMM (EPI_LNFOLD_*), two workgroups per CU, with the epilogue of a finished output tile drained INSIDE
// the main loop of the workgroup's next tile.
//
// Why (measured with tools/gemm_bench stamps on the ViT-B/16 shapes): a 160 x 128 workgroup spends 11 us in its K
// loop at K = 768 and 1.5 us before it (launch, address set-up, the first DMA round trip, the statistics fold) plus 1.6 - 4.7 us after it
// (LayerNorm fold, erf-GELU, stores).  While one of the two workgroups of a CU is outside its loop the matrix pipes of that CU are fed by
// one wave per SIMD only.  Here
//   * workgroups are persistent (grid = 2 x CUs) and walk the XCD-aware tile order round by round: no per-tile launch, the first K-tile
//     of the next tile is staged under the last K-tile of the current one;
//   * a wave keeps TWO accumulator sets: `acc` of the tile it is multiplying and `prev` of the tile it finished; one 16-row fragment row of
//     `prev` is finished (fold, GELU, 16-byte stores) per K-tile, in the first FM K-tiles of the next tile - VALU and store issue beside
//     the partner wave's (and its own) MFMAs; both workgroups of a CU stay in their loops;
//   * the (mean, rstd) of a tile's rows and its c / s vectors reach LDS during the tile's own loop, after the slices: one 16-byte pair of
//     statistics slots per row and K-tile by LDS-DMA into a scratch area, folded (Chan) behind the next K-tile's wait; the vectors by
//     4-byte LDS-DMA straight into place; double buffered by tile parity.  No load in the loop has a register destination, and the
//     statistics / vector areas are read and written by inline asm: hipcc would otherwise put s_waitcnt vmcnt(0) in front of them
//     (it orders every LDS access it sees behind all LDS-DMA in flight; a register load behind in-flight DMA is waited for with vmcnt(0) too);
//   * stores are bounds-checked buffer stores (rows >= M are dropped by the hardware), so every wave issues the same number of
//     vector-memory instructions per K-tile and the K loop waits with a COUNTED s_waitcnt: the epilogue stores stay in flight across it.
// Same MFMA sequence per accumulator and the same epilogue arithmetic as gemm_body + gemm_epilogue_lnfold: bit-identical outputs
// (tools/gemm_bench compares bitwise; the batch-independence tests pin it end to end).
#pragma once
#include "../gemm_kernel.h"

namespace ivit {

template <class T>
struct PersistLds {
    static constexpr int STATS_OFF = T::LDS_BYTES;                   // 2 x BM (mean, rstd)
    static constexpr int VEC_OFF = STATS_OFF + 2 * T::BM * 8;        // 2 x { c[BN], s[BN] }
    static constexpr int RAW_OFF = VEC_OFF + 2 * 2 * T::BN * 4;      // one 16-byte pair of statistics slots per row, landed by LDS-DMA
    static constexpr int RAW_WAVES = (T::BM + 63) / 64;
    static constexpr int BYTES = RAW_OFF + RAW_WAVES * 1024;
    static_assert(T::BN <= 128 && T::BN % 64 == 0, "fold vectors: one 4-byte DMA per column, whole waves");
};

constexpr int PERSIST_MAX_F4 = 8;   // statistics pairs are folded two slots (one 16-byte load) per K-tile: LayerNorm width <= 1024

template <int N>
__device__ __forceinline__ void persist_vmcnt() {
    static_assert(N >= 0 && N <= 63, "counted wait");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// tile of workgroup w in round `it`: rounds are contiguous runs of G tiles of the linear order, XCD-aware inside a round
__device__ __forceinline__ int persist_tile(int it, int w, int G, int ntiles) {
    const int base = it * G;
    const int cnt = min(G, ntiles - base);
    if (cnt <= 0 || w >= cnt) return -1;
    return base + xcd_tile(w, cnt);
}

// LDS accesses of the statistics / vector areas by inline asm: hipcc orders every LDS access it can see behind ALL LDS-DMA in flight
// (s_waitcnt vmcnt(0) in front of it), which would drain the operand pipeline in the middle of a K-tile.  These areas are never a DMA
// target; the reads wait for themselves (lgkmcnt) inside the statement.
__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(size_t)(IVIT_LDS const char*)p; }
__device__ __forceinline__ void lds_read_4x16(float4& a, float4& b, float4& c, float4& d, unsigned pa, unsigned pb, unsigned pc, unsigned pd) {
    asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %5\n\tds_read_b128 %2, %6\n\tds_read_b128 %3, %7\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d) : "v"(pa), "v"(pb), "v"(pc), "v"(pd) : "memory");
}
__device__ __forceinline__ float2 lds_read_8(unsigned pa) {
    float2 r;
    asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(r) : "v"(pa) : "memory");
    return r;
}
__device__ __forceinline__ void lds_write_8(unsigned pa, float2 v) { asm volatile("ds_write_b64 %0, %1" ::"v"(pa), "v"(v) : "memory"); }
__device__ __forceinline__ void lds_write_4(unsigned pa, float v) { asm volatile("ds_write_b32 %0, %1" ::"v"(pa), "v"(v) : "memory"); }

// one fragment row (16 rows x the wave's FN * 16 columns) of a finished tile: rstd (acc - mean s) + c, optional GELU, 16-byte stores
template <class T, class OP, bool GELU, int I>
__device__ __forceinline__ void persist_lf_slice(const f32x4 (&acc)[T::FM][T::FN], __amdgpu_buffer_rsrc_t out_rsrc, int ldo, int m_base, int n_base,
                                                 int fr, int fq, const float2* stats_w, const float* c_w, const float* s_w) {
    static_assert(T::FN % 2 == 0, "fragment pairs");
    const float2 st = lds_read_8(lds_addr(stats_w + I * 16 + fr));
    const float mu = st.x, rs = st.y;
    const int m = m_base + I * 16 + fr;
#pragma unroll
    for (int j = 0; j < T::FN; j += 2) {
        float4 c0, c1, s0, s1;
        lds_read_4x16(c0, c1, s0, s1, lds_addr(c_w + j * 16 + fq * 4), lds_addr(c_w + (j + 1) * 16 + fq * 4), lds_addr(s_w + j * 16 + fq * 4), lds_addr(s_w + (j + 1) * 16 + fq * 4));
        float a[4] = {fmaf(rs, fmaf(-mu, s0.x, acc[I][j][0]), c0.x), fmaf(rs, fmaf(-mu, s0.y, acc[I][j][1]), c0.y),
                      fmaf(rs, fmaf(-mu, s0.z, acc[I][j][2]), c0.z), fmaf(rs, fmaf(-mu, s0.w, acc[I][j][3]), c0.w)};
        float b[4] = {fmaf(rs, fmaf(-mu, s1.x, acc[I][j + 1][0]), c1.x), fmaf(rs, fmaf(-mu, s1.y, acc[I][j + 1][1]), c1.y),
                      fmaf(rs, fmaf(-mu, s1.z, acc[I][j + 1][2]), c1.z), fmaf(rs, fmaf(-mu, s1.w, acc[I][j + 1][3]), c1.w)};
        if (GELU) {
#pragma unroll
            for (int r = 0; r < 4; ++r) { a[r] = gelu_erf(a[r]); b[r] = gelu_erf(b[r]); }
        }
        const auto lo = __builtin_amdgcn_permlane16_swap(OP::pack2(a[0], a[1]), OP::pack2(b[0], b[1]), false, false);
        const auto hi = __builtin_amdgcn_permlane16_swap(OP::pack2(a[2], a[3]), OP::pack2(b[2], b[3]), false, false);
        const int n = n_base + (j + (fq & 1)) * 16 + (fq & ~1) * 4;
        u32x4 pk = {lo[0], hi[0], lo[1], hi[1]};
        // rows >= M lie beyond the descriptor's byte count: the store is issued (the K loop counts it) and dropped by the bounds check
        __builtin_amdgcn_raw_buffer_store_b128(pk, out_rsrc, (int)(((unsigned)m * (unsigned)ldo + (unsigned)n) * 2u), 0, 0);
    }
}

// every fragment row in turn (run-time index, compile-time register names)
template <class T, class OP, bool GELU>
__device__ __forceinline__ void persist_lf_slice_any(int i, const f32x4 (&acc)[T::FM][T::FN], __amdgpu_buffer_rsrc_t out_rsrc, int ldo, int m_base,
                                                     int n_base, int fr, int fq, const float2* stats_w, const float* c_w, const float* s_w) {
    static_assert(T::FM <= 5, "slice dispatch");
    switch (i) {
        case 0: persist_lf_slice<T, OP, GELU, 0>(acc, out_rsrc, ldo, m_base, n_base, fr, fq, stats_w, c_w, s_w); break;
        case 1: persist_lf_slice<T, OP, GELU, (T::FM > 1 ? 1 : 0)>(acc, out_rsrc, ldo, m_base, n_base, fr, fq, stats_w, c_w, s_w); break;
        case 2: persist_lf_slice<T, OP, GELU, (T::FM > 2 ? 2 : 0)>(acc, out_rsrc, ldo, m_base, n_base, fr, fq, stats_w, c_w, s_w); break;
        case 3: persist_lf_slice<T, OP, GELU, (T::FM > 3 ? 3 : 0)>(acc, out_rsrc, ldo, m_base, n_base, fr, fq, stats_w, c_w, s_w); break;
        default: persist_lf_slice<T, OP, GELU, (T::FM > 4 ? 4 : 0)>(acc, out_rsrc, ldo, m_base, n_base, fr, fq, stats_w, c_w, s_w); break;
    }
}

// Chan's update of (mean, M2) with the two FULL 64-column slots 2F and 2F + 1 held in one 16-byte load: the arithmetic of ln_chan_update's
// full-slot branch with compile-time ratios (the LayerNorm width is a multiple of 64 here), so the statistics carry the same bits as
// ln_tile_stats gives them
template <int F>
__device__ __forceinline__ void persist_fold2(float& mean, float& m2, const float4& raw) {
    ln_chan_update(mean, m2, raw.x, raw.y, 2 * F, 64 * (2 * F + 2));
    ln_chan_update(mean, m2, raw.z, raw.w, 2 * F + 1, 64 * (2 * F + 2));
}

#ifdef IVIT_GEMM_ABLATIONS   // shader-clock time per section of the K loop, summed per wave 0 of every workgroup (tools/gemm_bench)
#define IVIT_PSEC(k)                                                                                     \
    do {                                                                                                  \
        if (p.stamps) {                                                                                   \
            unsigned long long t_;                                                                        \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                     \
            psec[k] += t_ - plast; plast = t_;                                                            \
        }                                                                                                 \
    } while (0)
#else
#define IVIT_PSEC(k) do { } while (0)
#endif

// DUAL = true: two accumulator sets, the finished tile's slices inside the next tile's loop (above).  DUAL = false: one set; the whole
// epilogue follows the tile's own loop at once, beside the already staged first K-tile of the next tile - what persistence alone buys
// (no per-tile launch, first DMA round trip and statistics fold hidden).
template <class T, class OP, bool GELU, bool DUAL = true>
__device__ __forceinline__ void gemm_body_persist_lf(const GemmParams& p, char* smem) {
    using L = PersistLds<T>;
#ifdef IVIT_GEMM_ABLATIONS
    unsigned long long psec[6] = {0, 0, 0, 0, 0, 0}, plast = 0;
    if (p.stamps) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(plast)::"memory");
#endif
    constexpr int STORES = T::FN / 2;                       // buffer stores per slice and wave
    constexpr int KT_LOAD0 = DUAL ? T::FM : 0;               // K-tile that loads the first pair of statistics slots of the tile being multiplied
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wave / T::WAVES_N, wc = wave % T::WAVES_N;
    const int fr = lane & 15, fq = lane >> 4;
    const int tiles_m = ceil_div(p.M, T::BM), tiles_n = p.N / T::BN, ntiles = tiles_m * tiles_n;
    const int G = gridDim.x, w = blockIdx.x;
    const int nt = p.K / GEMM_BK;
    const size_t lda_b = (size_t)p.lda * 2, ldw_b = (size_t)p.ldw * 2;
    const int nf4 = p.ln_dim >> 7;                           // 16-byte loads of statistics pairs per row (ln_dim % 128 == 0; nt >= KT_LOAD0 + nf4 + 1)

    float2* const stats_lds = reinterpret_cast<float2*>(smem + L::STATS_OFF);
    float* const vec_lds = reinterpret_cast<float*>(smem + L::VEC_OFF);
    const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (int)((unsigned)p.M * (unsigned)p.ldo * 2u), 0x00020000);

    auto coords = [&](int tile, int& m0, int& n0) {
        int tm, tn;
        tile_coords(tile, tiles_m, tiles_n, tm, tn);
        m0 = tm * T::BM; n0 = tn * T::BN;
    };
    auto stage = [&](int m0, int n0, int kt, int buf) {
        char* dst = smem + buf * T::STAGE_BYTES;
        stage_tile<T::A_PIECES, T::WAVES>(p.A, lda_b, m0, kt * 128, dst, wave, lane);
        stage_tile<T::W_PIECES, T::WAVES>(p.W, ldw_b, n0, kt * 128, dst + T::A_BYTES, wave, lane);
    };

    f32x4 acc[T::FM][T::FN], prev[T::FM][T::FN];
#pragma unroll
    for (int i = 0; i < T::FM; ++i)
#pragma unroll
        for (int j = 0; j < T::FN; ++j) prev[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    int pm0 = 0, pn0 = 0, ppar = 0;                         // the finished tile `prev` belongs to
    bool have_prev = false;
    int buf = 0;                                            // LDS stage holding the K-tile about to be multiplied

    int tile = persist_tile(0, w, G, ntiles);
    if (tile < 0) return;                                   // (uniform) more workgroups than tiles
    IVIT_BODY_STAMP(0);
    int m0, n0;
    coords(tile, m0, n0);
    stage(m0, n0, 0, 0);
    IVIT_BODY_STAMP(1);

    for (int it = 0; tile >= 0; ++it) {
        const int next = persist_tile(it + 1, w, G, ntiles);
        int nm0 = 0, nn0 = 0;
        if (next >= 0) coords(next, nm0, nn0);
        const int par = it & 1;
#pragma unroll
        for (int i = 0; i < T::FM; ++i)
#pragma unroll
            for (int j = 0; j < T::FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        // this thread's row of the statistics table / column of the fold vectors of the tile being multiplied
        const float4* pr = reinterpret_cast<const float4*>(p.ln_part + (size_t)min(m0 + min((int)threadIdx.x, T::BM - 1), p.M - 1) * GEMM_LN_SLOTS);
        const int ncol = n0 + min((int)threadIdx.x, T::BN - 1);
        float mean = 0.f, m2 = 0.f;

        for (int t = 0; t < nt; ++t) {
            // K-tile t has landed for this wave; vector-memory instructions younger than its staging - the two stores of the slice of the
            // previous K-tile - may stay in flight.  The barrier publishes everyone's pieces and retires the reads of the stage that is
            // overwritten next.
            IVIT_PSEC(5);
            if (DUAL && have_prev && t >= 1 && t <= T::FM) persist_vmcnt<STORES>();
            else if (!DUAL && it > 0 && t == 0) persist_vmcnt<T::FM * STORES>();   // the previous tile's epilogue stores, issued behind this K-tile's staging
            else persist_vmcnt<0>();
            IVIT_PSEC(0);
            __builtin_amdgcn_s_barrier();
            IVIT_PSEC(1);
            {   // statistics: the pair of slots loaded in the previous K-tile (landed: the wait above), folded BEFORE the next DMA is issued
                // (hipcc waits for everything in flight at the first use of an ordinary load's result)
                const int f = t - KT_LOAD0 - 1;
                if (f >= 0 && f < nf4 && wave < L::RAW_WAVES) {
                    float4 raw;
                    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(raw) : "v"(lds_addr(smem + L::RAW_OFF + threadIdx.x * 16)) : "memory");
                    switch (f) {
                        case 0: persist_fold2<0>(mean, m2, raw); break;
                        case 1: persist_fold2<1>(mean, m2, raw); break;
                        case 2: persist_fold2<2>(mean, m2, raw); break;
                        case 3: persist_fold2<3>(mean, m2, raw); break;
                        case 4: persist_fold2<4>(mean, m2, raw); break;
                        case 5: persist_fold2<5>(mean, m2, raw); break;
                        case 6: persist_fold2<6>(mean, m2, raw); break;
                        default: persist_fold2<7>(mean, m2, raw); break;
                    }
                    if (f == nf4 - 1) {
                        if ((int)threadIdx.x < T::BM) lds_write_8(lds_addr(stats_lds + par * T::BM + threadIdx.x), make_float2(mean, 1.0f / sqrtf(m2 / (float)p.ln_dim + p.ln_eps)));
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // written before this wave arrives at the next barrier
                    }
                }
            }
            if (t + 1 < nt) stage(m0, n0, t + 1, buf ^ 1);
            else if (next >= 0) stage(nm0, nn0, 0, buf ^ 1);
            asm volatile("" ::: "memory");   // the slice's stores and the statistics loads below stay YOUNGER than this staging (the counted waits rely on it)
            IVIT_PSEC(2);
            const char* a_tile = smem + buf * T::STAGE_BYTES;
            const char* w_tile = a_tile + T::A_BYTES;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                bf16x8 af[T::FM], wf[T::FN];
#pragma unroll
                for (int j = 0; j < T::FN; ++j) wf[j] = read_frag(w_tile, (wc * T::FN + j) * 16 + fr, kk * 4 + fq);
#pragma unroll
                for (int i = 0; i < T::FM; ++i) af[i] = read_frag(a_tile, (wr * T::FM + i) * 16 + fr, kk * 4 + fq);
#pragma unroll
                for (int i = 0; i < T::FM; ++i)
#pragma unroll
                    for (int j = 0; j < T::FN; ++j) acc[i][j] = OP::mfma(wf[j], af[i], acc[i][j]);
            }
            IVIT_PSEC(3);
            if (DUAL && have_prev && t < T::FM)
                persist_lf_slice_any<T, OP, GELU>(t, prev, out_rsrc, p.ldo, pm0 + wr * T::FM * 16, pn0 + wc * T::FN * 16, fr, fq,
                                                  stats_lds + ppar * T::BM + wr * T::FM * 16, vec_lds + ppar * 2 * T::BN + wc * T::FN * 16,
                                                  vec_lds + ppar * 2 * T::BN + T::BN + wc * T::FN * 16);
            IVIT_PSEC(4);
            {   // this tile's statistics pairs (one 16-byte pair of slots per row and K-tile) and, once, its fold vectors: LDS-DMA like the
                // operand tiles (no register destination, so hipcc adds no waits of its own), retired by the next K-tile's wait
                const int f = t - KT_LOAD0;
                if (f >= 0 && f < nf4) {
                    if (wave < L::RAW_WAVES)
                        __builtin_amdgcn_global_load_lds((const IVIT_GLOBAL void*)(pr + f), (IVIT_LDS void*)(smem + L::RAW_OFF + wave * 1024), 16, 0, 0);
                    if (f == 0 && wave < T::BN / 64) {
                        __builtin_amdgcn_global_load_lds((const IVIT_GLOBAL void*)(p.bias + ncol), (IVIT_LDS void*)(vec_lds + par * 2 * T::BN + wave * 64), 4, 0, 0);
                        __builtin_amdgcn_global_load_lds((const IVIT_GLOBAL void*)(p.ln_s + ncol), (IVIT_LDS void*)(vec_lds + par * 2 * T::BN + T::BN + wave * 64), 4, 0, 0);
                    }
                }
            }
            buf ^= 1;
        }
        if (DUAL) {
#pragma unroll
            for (int i = 0; i < T::FM; ++i)
#pragma unroll
                for (int j = 0; j < T::FN; ++j) prev[i][j] = acc[i][j];
            pm0 = m0; pn0 = n0; ppar = par; have_prev = true;
        } else {
            // the tile's statistics were written as late as its last K-tiles (by other waves too): one barrier, then the whole epilogue
            // while the next tile's first K-tile is in flight (its stores stay behind that staging: the next wait is vmcnt(0))
            __builtin_amdgcn_s_barrier();
#pragma unroll
            for (int i = 0; i < T::FM; ++i)
                persist_lf_slice_any<T, OP, GELU>(i, acc, out_rsrc, p.ldo, m0 + wr * T::FM * 16, n0 + wc * T::FN * 16, fr, fq,
                                                  stats_lds + par * T::BM + wr * T::FM * 16, vec_lds + par * 2 * T::BN + wc * T::FN * 16,
                                                  vec_lds + par * 2 * T::BN + T::BN + wc * T::FN * 16);
        }
        tile = next; m0 = nm0; n0 = nn0;
    }
    if (!DUAL) return;
    // the last tile's epilogue: its statistics were written (by other waves too) as late as its last K-tile
    __builtin_amdgcn_s_barrier();
    IVIT_BODY_STAMP(2);
#pragma unroll
    for (int i = 0; i < T::FM; ++i)
        persist_lf_slice_any<T, OP, GELU>(i, prev, out_rsrc, p.ldo, pm0 + wr * T::FM * 16, pn0 + wc * T::FN * 16, fr, fq,
                                          stats_lds + ppar * T::BM + wr * T::FM * 16, vec_lds + ppar * 2 * T::BN + wc * T::FN * 16,
                                          vec_lds + ppar * 2 * T::BN + T::BN + wc * T::FN * 16);
    IVIT_BODY_STAMP(3);
#ifdef IVIT_GEMM_ABLATIONS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    IVIT_BODY_STAMP(4);
    if (p.stamps && threadIdx.x == 0) {  // which CU ran this workgroup (per-CU timelines in tools/gemm_bench)
        p.stamps[(size_t)blockIdx.x * 16 + 5] = ((unsigned long long)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 15) << 32) | __builtin_amdgcn_s_getreg((31 << 11) | 4);
        for (int k = 0; k < 6; ++k) p.stamps[(size_t)blockIdx.x * 16 + 8 + k] = psec[k];
    }
#endif
}

}  // namespace ivit
