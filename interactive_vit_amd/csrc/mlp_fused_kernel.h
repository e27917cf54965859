// Fused MLP of one encoder layer for gfx950: LayerNorm (folded) -> up projection -> GELU -> down projection -> residual add
// (-> 16-bit copy + LayerNorm statistics pairs of the new rows), ONE launch, the hidden activations never leave the CU.
//
// Why: as two GEMM launches the hidden tensor u [M, Mlp] is written and read back (2 x 77.5 MB per layer at ViT-B/16 B = 64), each launch pays
// its own prologue / epilogue phases (35 % of a 160 x 128 tile's life at K = 768) and a kernel boundary.  Here a workgroup owns BM = 64 token
// rows for the whole MLP and walks the hidden dimension in chunks of 128:
//
//     phase 1   acc1[64 x 128]  = X[64 x D] . W1'[chunk, D]^T          (K = D: ND = D / 64 K-tiles, X resident in LDS)
//     epilogue  U[64 x 128]     = rn16(gelu(rstd (acc1 - mean s) + c))  -> LDS (16 KiB), read back as MFMA operand fragments (registers)
//     phase 2   acc2[64 x D]   += U . W2[:, chunk]^T                    (K = 128; acc2 = 96 VGPRs per lane, live for the whole kernel)
//
// and finishes with the residual epilogue of the MLP-down GEMM (gemm_epilogue_family<EK = 1 / 3>).  The MFMA sequence per accumulator is the
// one the two GEMM kernels issue (K-tiles ascending, kk = 0, 1 inside; hi before lo for split weights), the epilogue arithmetic is theirs:
// outputs are BIT-IDENTICAL to launch_gemm(EPI_LNFOLD_GELU_BF16) + launch_gemm(EPI_BIAS_RESID_STATS) (tools/mlp_fused_bench, tests).
//
// What bounds it: weights.  Every workgroup streams all of W1' and W2 (9.4 MB at ViT-B/16) through its CU's vector-memory return path
// (42-47 B/clk, profiles/r03b_operand_path_probe.txt): 393 KB per chunk against 6.1 k cycles of MFMA.  So the loop is a weight STREAM:
//   * 16-KiB slots (W1': 128 hidden rows x 64 k; W2: 64 output columns x 128 hidden) in a ring of FOUR LDS buffers filled by LDS-DMA
//     (global_load_lds_dwordx4, 2 pieces per wave and slot), issued three slots ahead behind a counted s_waitcnt vmcnt - 32-48 KiB in
//     flight per CU at all times, no drain in the loop;
//   * one s_barrier per slot ("step"); the fragments of slot u + 1 are read from LDS during the MFMAs of slot u (register double buffer), so
//     a step is [wait + barrier, issue DMA of slot u + 4 into the buffer slot u just vacated, ds_read slot u + 1, 8 MFMAs per wave];
//   * LDS = X (ND x 8 KiB) + 4 x 16 KiB = 160 KiB at D = 768: U has no home of its own - it is written into the buffer of the FIRST W2 slot
//     once that slot's fragments are in registers (one extra barrier per chunk), and that buffer's next DMA is issued one step late.
// Wave layout: 8 waves = 2 (rows) x 4; phase 1: wave (wr, wc) owns rows 32 wr.., hidden columns 32 wc.. of the chunk (2 x 2 fragments);
// phase 2: rows 32 wr.., output columns {64 (wc + 4 g) + 16 j} - i.e. WHOLE 64-column statistics slots wc, wc + 4, wc + 8, one 16-column
// fragment j of each per W2 slot (the W2 slot gathers those rows: an LDS-DMA source address is per lane) - so the residual / statistics
// epilogue of a wave is exactly the (FM = 2, FN = 4) epilogue of the GEMM kernels, slot by slot.
#pragma once
#include "gemm_kernel.h"
#include <utility>

namespace ivit {

using MlpTile = GemmTile<2, 4, 2, 4>;   // shape constants for the shared epilogues: 2 x 4 waves of (2 x 4) fragments = 64 rows x 256 columns per "pass"

template <int ND_, int SPLIT_>
struct MlpFusedGeom {
    static constexpr int ND = ND_, SPLIT = SPLIT_;
    static constexpr int BM = 64, HC = 128, THREADS = 512;
    static constexpr int P = ND * SPLIT;          // phase-1 steps (W1' slots) per chunk
    static constexpr int TS = 2 * P;              // steps per chunk
    static constexpr int NG = ND / 4;             // 64-column statistics slots per wave
    static constexpr int X_BYTES = ND * 8192, SLOT = 16384, LDS_BYTES = X_BYTES + 4 * SLOT;
    static constexpr int UC = P - 5;              // step that loads the chunk's fold vectors (consumed after step P - 1)
    static_assert(ND % 4 == 0 && P >= 8, "D must be a multiple of 256 and at least 512");
};

// compile-time loop: f(std::integral_constant<int, 0>) ... f(std::integral_constant<int, N - 1>) (a 48-step body is beyond what #pragma unroll accepts)
template <class F, int... I>
__device__ __forceinline__ void mlpf_static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void mlpf_static_for(F&& f) { mlpf_static_for_impl(f, std::make_integer_sequence<int, N>{}); }

#ifndef IVIT_MLPF_TOUCH
#define IVIT_MLPF_TOUCH 1   // 0 (A/B builds): no L2 warm-up loads
#endif
#ifndef IVIT_MLPF_STAGGER
#define IVIT_MLPF_STAGGER 1   // 0 (A/B builds): every wave runs the early-MFMA order
#endif
#ifndef IVIT_MLPF_ORDER
#define IVIT_MLPF_ORDER 1   // 0 (A/B builds): leave the instruction order inside a step to hipcc
#endif
#define IVIT_MLPF_WAIT(N) __builtin_amdgcn_s_waitcnt(((N) & 15) | (7 << 4) | (((N) >> 4) << 14))   /* vmcnt(N) lgkmcnt(0), gfx9 encoding: a builtin, so hipcc's own wait bookkeeping sees it */

#ifdef IVIT_MLPF_STAMPS   // tools/mlp_fused_bench only
#define IVIT_MLPF_STAMP(slot)                                                                                  \
    do {                                                                                                      \
        if (p.stamps && threadIdx.x == 0) {                                                                   \
            unsigned long long t_;                                                                            \
            asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                     \
            p.stamps[(size_t)blockIdx.x * 8 + (slot)] = t_;                                                   \
        }                                                                                                     \
    } while (0)
#else
#define IVIT_MLPF_STAMP(slot) do { } while (0)
#endif

// DBG (tools/mlp_fused_bench only; 0 in the product): timing ablations, a bit mask - 1 = no DMA inside the loop, 2 = no MFMA, 4 = no LDS fragment reads,
// 8 = no s_barrier inside the loop (results are wrong in every ablation; only the time is read)
template <int ND, int SPLIT, class OP, int DBG = 0>
__device__ __forceinline__ void mlp_fused_body(const MlpFusedParams& p, char* smem) {
    using G = MlpFusedGeom<ND, SPLIT>;
    using T = MlpTile;
    constexpr int P = G::P, TS = G::TS, NG = G::NG, SLOT = G::SLOT, UC = G::UC;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int fr = lane & 15, fq = lane >> 4;
    const int m0 = blockIdx.x * G::BM;
    char* ring = smem + G::X_BYTES;
    IVIT_MLPF_STAMP(0);

    // ---- LDS-DMA sources: 8 rows x 128 B per wave-instruction, the 16-B chunk XOR-swizzled with the row on the SOURCE side.  A source address is
    //      (wave-uniform base of the slot) + (32-bit lane offset): the uniform part is scalar arithmetic, the lane part one register per matrix
    const int r_in = lane >> 3, swz = (lane & 7) ^ r_in;
    const size_t ldw1_b = (size_t)p.ldw1 * 2, ldw2_b = (size_t)p.ldw2 * 2;
    const unsigned w1_off = (unsigned)((wave * 8 + r_in) * (int)ldw1_b + swz * 16);
    // W2 slot t gathers, for every wave column q, the 16 output columns 64 (q + 4 (t >> 2)) + 16 (t & 3) ..: piece `wave` = (q = wave >> 1, half = wave & 1)
    const unsigned w2_off = (unsigned)((64 * (wave >> 1) + 8 * (wave & 1) + r_in) * (int)ldw2_b + swz * 16);
    auto glds = [](const char* src, char* dst) {
        __builtin_amdgcn_global_load_lds((const IVIT_GLOBAL void*)src, (IVIT_LDS void*)dst, 16, 0, 0);
    };
    // slot s of chunk cc into ring buffer s & 3 (2 pieces per wave; s is a compile-time constant wherever this is called)
    auto dma_slot = [&](int cc, int s) {
        char* dst = ring + (s & 3) * SLOT + wave * 1024;
        if (s < P) {            // W1': hidden rows cc * 128 + [0, 128), K-tile s (split: hi / lo K-tiles alternate)
            const char* sb = reinterpret_cast<const char*>(p.W1) + (size_t)cc * 128 * ldw1_b + s * 128;
            glds(sb + w1_off, dst);
            glds(sb + 64 * ldw1_b + w1_off, dst + 8192);
        } else {                // W2: 64 gathered output rows x hidden [cc * 128, +128) as two 64-deep K-tile images (split: [hi | lo] of 64 hidden)
            const int v = s - P, t = v / SPLIT, sub = v % SPLIT;
            const char* sb = reinterpret_cast<const char*>(p.W2) + (size_t)(256 * (t >> 2) + 16 * (t & 3)) * ldw2_b + (size_t)(cc * SPLIT + sub) * 256;
            glds(sb + w2_off, dst);
            glds(sb + 128 + w2_off, dst + 8192);
        }
    };

    // ---- L2 warm-up one chunk ahead.  The workgroups that share an XCD (observed: blockIdx mod 8; speed only) ask for the same weight lines at almost
    //      the same time, so every one of them waits out the L2 miss of the first (Infinity Cache: ~0.6 us under this load - with 32-48 KiB in flight
    //      per CU that, not the CU's 45 B/clk return path, set the rate: profiles/r05_fused_mlp.txt).  Once per chunk every lane loads ONE dword of the
    //      next chunk's weight block (its 128-byte lines dealt over the workgroups of the XCD group, 2 x 256 per workgroup), fifteen steps before the
    //      first DMA that needs them: the DMA then hits L2.
    constexpr int W1_LPR = SPLIT * ND, W1_LINES = 128 * W1_LPR, W2_LPR = 2 * SPLIT, W2_LINES = ND * 64 * W2_LPR;   // lines per row / per chunk block
    static_assert(W1_LINES == W2_LINES, "the two halves of the workgroup warm equal shares");
    // waves 0-3 take W1' lines, waves 4-7 W2 lines (a wave-uniform choice: the base stays scalar, the lane part is one 32-bit offset)
    const int touch_line = ((int)(blockIdx.x >> 3) * 256 + (int)(threadIdx.x & 255)) % W1_LINES;
    const unsigned touch_off = wave < 4 ? (unsigned)((touch_line / W1_LPR) * (int)ldw1_b + (touch_line % W1_LPR) * 128)
                                        : (unsigned)((touch_line / W2_LPR) * (int)ldw2_b + (touch_line % W2_LPR) * 128);
    auto touch_ptr = [&](int cc) -> const unsigned* {
        const char* sb = wave < 4 ? reinterpret_cast<const char*>(p.W1) + (size_t)cc * 128 * ldw1_b : reinterpret_cast<const char*>(p.W2) + (size_t)cc * SPLIT * 256;
        return reinterpret_cast<const unsigned*>(sb + touch_off);
    };
    unsigned touch = 0;

    // ---- prologue: X rows (ND K-tile images of 64 rows x 128 B) and the first four slots
    {
        const char* x_lane = reinterpret_cast<const char*>(p.X) + (size_t)(m0 + wave * 8 + r_in) * ((size_t)p.ldx * 2) + swz * 16;
#pragma unroll
        for (int kt = 0; kt < ND; ++kt) glds(x_lane + kt * 128, smem + kt * 8192 + wave * 1024);
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) dma_slot(0, s);

    // (mean, rstd) of this lane's two accumulator rows, from the statistics pairs (the fold every _lf GEMM does); rows past M: a valid row's
    float mean_i[2], rstd_i[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const float2 st = ln_row_stats_from_pairs(p.ln_part_in, min(m0 + 32 * wr + 16 * i + fr, p.M - 1), ND * 64, p.ln_eps);
        mean_i[i] = st.x;
        rstd_i[i] = st.y;
    }

    // ---- LDS read offsets (row r, 16-B chunk q) -> r * 128 + ((q ^ (r & 7)) << 4); kk = 1 flips bit 2 of the chunk
    const int a0 = (fq ^ (fr & 7)) << 4;
    const int off_w1[2] = {(32 * wc + fr) * 128 + a0, (32 * wc + fr) * 128 + (a0 ^ 64)};
    const int off_x[2] = {(32 * wr + fr) * 128 + a0, (32 * wr + fr) * 128 + (a0 ^ 64)};
    const int off_w2[2] = {(16 * wc + fr) * 128 + a0, (16 * wc + fr) * 128 + (a0 ^ 64)};
    auto ld16 = [](const char* q) { return *reinterpret_cast<const bf16x8*>(q); };

    f32x4 acc2[NG][2][4];
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc2[g][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 acc1[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc1[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    bf16x8 wf[2][2][2] = {};    // W1' fragments [set][j][kk]: set = slot & 1
    bf16x8 xf[2][2][2] = {};    // X fragments [set][i][kk]: set = K-tile & 1 (split: kept across the hi / lo pair)
    bf16x8 w2f[2][4] = {};      // W2 fragments [set][k-step]
    bf16x8 uf[2][4] = {};       // U fragments [i][k-step of 32 hidden]
    float4 c4[2], s4[2];   // fold vectors of the chunk's hidden columns 32 wc + 16 j + 4 fq ..

    auto read_w1 = [&](int s, bf16x8 (&f)[2][2]) {
        const char* b = ring + (s & 3) * SLOT;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int j = 0; j < 2; ++j) f[j][kk] = ld16(b + off_w1[kk] + j * 2048);
    };
    auto read_x = [&](int kt, bf16x8 (&f)[2][2]) {
        const char* b = smem + kt * 8192;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 2; ++i) f[i][kk] = ld16(b + off_x[kk] + i * 2048);
    };
    auto read_w2 = [&](int s, bf16x8 (&f)[4]) {
        const char* b = ring + (s & 3) * SLOT;
#pragma unroll
        for (int q = 0; q < 4; ++q) f[q] = ld16(b + (q >> 1) * 8192 + off_w2[q & 1]);
    };

    // everything issued so far has landed (hipcc waits vmcnt(0) for the statistics loads above anyway); publish, then slot 0's fragments
    IVIT_MLPF_WAIT(0);
    __builtin_amdgcn_s_barrier();
    read_w1(0, wf[0]);
    read_x(0, xf[0]);
    IVIT_MLPF_STAMP(1);

    const int nchunks = p.Mlp / G::HC;
    // The chunk loop in two instruction orders (LATE = waves 4-7, which share their SIMDs with waves 0-3): the early half of the workgroup starts a
    // step with its MFMAs and issues its LDS reads / DMA between them, the late half issues its memory instructions first and its MFMAs after
    // them - on every SIMD one wave is in its matrix burst while its partner is in its memory burst, inside the same barrier interval.
    auto chunk_loop = [&](auto late_tag) {
    constexpr bool LATE = decltype(late_tag)::value;
    for (int c = 0; c < nchunks; ++c) {
        const int cn = c + 1 < nchunks ? c + 1 : 0;   // the last chunk's look-ahead re-loads chunk 0 into dead buffers: every count stays constant
        mlpf_static_for<TS>([&](auto u_tag) {
            constexpr int u = decltype(u_tag)::value;   // compile-time step index: every buffer, register set and wait count below is a constant
            // ---- slot u + 1 has landed for this wave: all but the N youngest vector-memory operations are done (2 per slot in flight behind it; the
            //      fold vectors add 4 for two steps; one slot fewer is in flight at step P + 1, see below); LDS reads of the previous step retired
            //      too, so after the barrier the buffer of slot u may be overwritten
            if (u == P + 1) IVIT_MLPF_WAIT(2);
            else if (u == UC + 1 || u == UC + 2) { if (IVIT_MLPF_TOUCH) IVIT_MLPF_WAIT(9); else IVIT_MLPF_WAIT(8); }
            else IVIT_MLPF_WAIT(4);
            if (!(DBG & 8)) __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            if (u == UC) {   // fold vectors of this chunk, ordinary loads issued BEFORE this step's DMA (they are older than slot u + 4: retired by the wait of step u + 3)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    c4[j] = *reinterpret_cast<const float4*>(p.c1 + c * 128 + 32 * wc + 16 * j + 4 * fq);
                    s4[j] = *reinterpret_cast<const float4*>(p.s1 + c * 128 + 32 * wc + 16 * j + 4 * fq);
                }
                if (IVIT_MLPF_TOUCH) touch = *touch_ptr(cn);   // the NEXT chunk's weights into this XCD's L2 (see touch_ptr)
            }
            if (u == UC + 3 && IVIT_MLPF_TOUCH) asm volatile("" ::"v"(touch));   // retired by this step's wait; the value is never used
            // ---- DMA: slot u + 4 into the buffer slot u vacated (its fragments were read during step u - 1).  The buffer of slot P holds U during
            //      step P, so its refill (slot P + 4) waits one step and goes out together with slot P + 5.
            if (u != P && !(DBG & 1)) {
                if (u == P + 1) { dma_slot(c, P + 4); dma_slot(c, P + 5); }
                else if (u + 4 < TS) dma_slot(c, u + 4);
                else dma_slot(cn, u + 4 - TS);
            }
            if (u == P) {   // U (written after the extra barrier below, published by this step's barrier) -> operand fragments
                const char* b = ring + (P & 3) * SLOT;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int q = 0; q < 4; ++q) uf[i][q] = ld16(b + (q >> 1) * 8192 + off_x[q & 1] + i * 2048);
            }
            // ---- fragments of slot u + 1 (they return under this step's MFMAs)
            if constexpr ((DBG & 4) != 0) {
            } else if constexpr (u + 1 < P) {
                read_w1(u + 1, wf[(u + 1) & 1]);
                if ((u + 1) % SPLIT == 0) read_x((u + 1) / SPLIT, xf[((u + 1) / SPLIT) & 1]);
            } else if constexpr (u + 1 < TS) {
                read_w2(u + 1, w2f[(u + 1 - P) & 1]);
            } else {
                read_w1(0, wf[0]);
                read_x(0, xf[0]);
            }
            // ---- MFMAs of slot u
            if constexpr ((DBG & 2) != 0) {   // keep the fragments alive (cdna_hip_programming.md rule 17)
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b)
#pragma unroll
                        for (int kk = 0; kk < 2; ++kk) { asm volatile("" ::"v"(wf[a][b][kk])); asm volatile("" ::"v"(xf[a][b][kk])); }
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int q = 0; q < 4; ++q) { asm volatile("" ::"v"(w2f[a][q])); asm volatile("" ::"v"(uf[a][q])); }
            } else if constexpr (u < P) {
                const int kt = u / SPLIT;
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) acc1[i][j] = OP::mfma(wf[u & 1][j][kk], xf[kt & 1][i][kk], acc1[i][j]);
            } else {
                const int v = u - P, t = v / SPLIT, sub = v % SPLIT;
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        // unsplit: the slot's four k-steps of 32 hidden; split: hi (kk 0, 1) then lo (kk 0, 1) of hidden K-tile 2 c + sub
                        const int ks = SPLIT == 1 ? q : 2 * sub + (q & 1);
                        acc2[t >> 2][i][t & 3] = OP::mfma(w2f[v & 1][q], uf[i][ks], acc2[t >> 2][i][t & 3]);
                    }
            }
            // ---- issue order inside the step (one scheduling region): the MFMAs start at once - their fragments are in registers - and the next slot's
            //      LDS reads (two per gap), then the DMA issues (one per gap), go out between them.  Left to itself hipcc puts the DMA issues first: every
            //      wave of the workgroup is in the same step, so the matrix pipes then idle through them after each barrier.
            if constexpr (DBG == 0 && u != P && IVIT_MLPF_ORDER) {
                constexpr int NR = (u + 1 < P) ? (((u + 1) % SPLIT == 0) ? 8 : 4) : (u + 1 < TS ? 4 : 8);
                constexpr int NV = (u == P + 1) ? 4 : 2;
                if constexpr (LATE) {   // DMA issues first (the partner wave's MFMAs run beside them), then reads and MFMAs interleaved
                    __builtin_amdgcn_sched_group_barrier(0x10, NV, 0);
#pragma unroll
                    for (int k = 0; k < NR / 2; ++k) { __builtin_amdgcn_sched_group_barrier(0x100, 2, 0); __builtin_amdgcn_sched_group_barrier(0x8, 1, 0); }
                    __builtin_amdgcn_sched_group_barrier(0x8, 8 - NR / 2, 0);
                } else {
                    __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
#pragma unroll
                    for (int k = 0; k < NR / 2; ++k) { __builtin_amdgcn_sched_group_barrier(0x100, 2, 0); __builtin_amdgcn_sched_group_barrier(0x8, 1, 0); }
#pragma unroll
                    for (int k = 0; k < NV; ++k) { __builtin_amdgcn_sched_group_barrier(0x10, 1, 0); if (1 + NR / 2 + k < 8) __builtin_amdgcn_sched_group_barrier(0x8, 1, 0); }
                    if constexpr (1 + NR / 2 + NV < 8) __builtin_amdgcn_sched_group_barrier(0x8, 8 - (1 + NR / 2 + NV), 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (u == P - 1) {
                // ---- phase-1 epilogue (gemm_epilogue_lnfold's arithmetic): u = rn16(gelu(rstd (acc - mean s) + c)), 8 consecutive hidden per lane after the
                //      fragment-pair lane swap, one 16-byte LDS store per fragment row
                u32x4 pk[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const float mu = mean_i[i], rs = rstd_i[i];
                    float a[4] = {fmaf(rs, fmaf(-mu, s4[0].x, acc1[i][0][0]), c4[0].x), fmaf(rs, fmaf(-mu, s4[0].y, acc1[i][0][1]), c4[0].y),
                                  fmaf(rs, fmaf(-mu, s4[0].z, acc1[i][0][2]), c4[0].z), fmaf(rs, fmaf(-mu, s4[0].w, acc1[i][0][3]), c4[0].w)};
                    float b[4] = {fmaf(rs, fmaf(-mu, s4[1].x, acc1[i][1][0]), c4[1].x), fmaf(rs, fmaf(-mu, s4[1].y, acc1[i][1][1]), c4[1].y),
                                  fmaf(rs, fmaf(-mu, s4[1].z, acc1[i][1][2]), c4[1].z), fmaf(rs, fmaf(-mu, s4[1].w, acc1[i][1][3]), c4[1].w)};
#pragma unroll
                    for (int r = 0; r < 4; ++r) { a[r] = gelu_erf(a[r]); b[r] = gelu_erf(b[r]); }
                    const auto lo = __builtin_amdgcn_permlane16_swap(OP::pack2(a[0], a[1]), OP::pack2(b[0], b[1]), false, false);
                    const auto hi = __builtin_amdgcn_permlane16_swap(OP::pack2(a[2], a[3]), OP::pack2(b[2], b[3]), false, false);
                    pk[i] = u32x4{lo[0], hi[0], lo[1], hi[1]};
                    acc1[i][0] = f32x4{0.f, 0.f, 0.f, 0.f};
                    acc1[i][1] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
                // every wave holds slot P's fragments (read above, retired by this wait): its buffer becomes U
                IVIT_MLPF_WAIT(6);   // lgkmcnt(0); the three slots in flight stay in flight
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                const int hcol = 32 * wc + (fq & 1) * 16 + (fq & ~1) * 4;   // first of this lane's 8 consecutive hidden columns
                char* ub = ring + (P & 3) * SLOT + (hcol >> 6) * 8192;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int row = 32 * wr + 16 * i + fr;
                    *reinterpret_cast<u32x4*>(ub + row * 128 + ((((hcol & 63) >> 3) ^ (row & 7)) << 4)) = pk[i];
                }
            }
        });
    }
    };
    if (IVIT_MLPF_STAGGER && wave >= 4) chunk_loop(std::true_type{});
    else chunk_loop(std::false_type{});
    IVIT_MLPF_STAMP(2);
    IVIT_MLPF_WAIT(0);   // the last chunk's look-ahead DMA still writes LDS

    // ---- residual epilogue of the MLP-down GEMM, one 64-column statistics slot at a time
    GemmParams gp{};
    gp.M = p.M; gp.N = p.D; gp.bias = p.b2; gp.resid = p.resid; gp.ldr = p.ldr; gp.out = p.out; gp.ldo = p.ldo;
    gp.xb = p.xb; gp.ldxb = p.ldxb; gp.ln_part = p.ln_part_out;
    gp.epi = p.stats_out ? EPI_BIAS_RESID_STATS : EPI_BIAS_RESID_F32;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const int n_base = 64 * (wc + 4 * g), m_base = m0 + 32 * wr;
        if (p.stats_out) gemm_epilogue_family<T, 1, OP, 1>(gp, acc2[g], m_base, n_base, fr, fq);
        else gemm_epilogue_family<T, 3, OP, 1>(gp, acc2[g], m_base, n_base, fr, fq);
    }
    IVIT_MLPF_STAMP(3);
}

}  // namespace ivit
