// Fused MLP of one encoder layer for gfx950: LayerNorm (folded) -> up projection -> GELU -> down projection -> residual add
// (-> 16-bit copy + LayerNorm statistics pairs of the new rows), ONE launch, the hidden activations never leave the CU.
//
// Why: as two GEMM launches the hidden tensor u [M, Mlp] is written and read back (2 x 77.5 MB per layer at ViT-B/16 B = 64), each launch pays
// its own prologue / epilogue phases (35 % of a 160 x 128 tile's life at K = 768) and a kernel boundary.  Here a workgroup owns BM = 64 token
// rows for the whole MLP and walks the hidden dimension in chunks of 128:
//
//     phase 1   acc1[64 x 128]  = X[64 x D] . W1'[chunk, D]^T          (K = D: ND = D / 64 K-tiles, X resident in LDS)
//     epilogue  U[64 x 128]     = rn16(gelu(rstd (acc1 - mean s) + c))  -> LDS (16 KiB), read back as MFMA operand fragments (registers)
//     phase 2   acc2[64 x D]   += U . W2[:, chunk]^T                    (K = 128; acc2 = D / 8 VGPRs per lane, live for the whole kernel)
//
// and finishes with the residual epilogue of the MLP-down GEMM.  The MFMA sequence per accumulator is the one the two GEMM kernels issue
// (K-tiles ascending, kk = 0, 1 inside; hi before lo for split weights), the epilogue arithmetic is theirs: outputs are BIT-IDENTICAL to
// launch_gemm(EPI_LNFOLD_GELU_BF16) + launch_gemm(EPI_BIAS_RESID_STATS) (tools/mlp_fused_bench, tests/test_gpu_parity.py).
//
// What bounds it: weights.  Every workgroup streams all of W1' and W2 (9.4 MB at ViT-B/16): 393 KB per chunk against 6.1 k cycles of MFMA - 1.86 GB per
// launch out of the XCDs' L2s at ViT-B/16 B = 64, ~120 us per workgroup + a 15 us epilogue burst.  (Measured and not explained by either simple model: half
// the row blocks alone take 125 us, and handing a sixth of every block's chunks to the CUs a 197-block grid leaves idle made nothing faster:
// profiles/r05_fused_mlp.txt.)  The first structure (weights through a
// ring of LDS buffers, one barrier per 16-KiB slot: profiles/r05_fused_mlp_v1.txt) lost to the two launches - the LDS that X leaves holds too
// little DMA in flight, and in barrier lockstep MFMA, LDS reads and DMA do not overlap.  This one shares NO weight fragment between waves:
//   * 1 x 8 wave layout: wave w owns ALL 64 rows and, in phase 1, hidden columns 16 w .. of the chunk (4 x 1 fragments), in phase 2 the output
//     columns of one whole 64-column statistics slot (w) plus half of a shared one (8 + w / 2) - D / 8 columns, (4 x ND / 2) fragments;
//   * so an operand stream that nobody shares does not go through LDS at all (cdna_hip_programming.md, "GEMV / M <= 16" row): both matrices
//     are re-laid ONCE (launch_mlp_pack_weights) into the order the kernel consumes them, 1-KiB blocks that ARE MFMA fragments (lane l holds
//     row l & 15, k-chunk l >> 4), and every wave pulls its own contiguous stream with plain 16-byte-per-lane loads, R blocks (8 KiB) ahead in
//     registers: 64 KiB in flight per CU, waits counted by hipcc itself (no LDS-DMA in the loop, so it can);
//   * LDS holds X (ND x 8 KiB, read 8 x redundantly: 4 fragment reads per K-step) and two U buffers; ONE s_barrier per chunk (U written ->
//     U read; the second buffer makes the write-after-read side free), so the waves of a CU drift apart and fill each other's stalls.
// The residual / statistics epilogue: a wave's whole slot is the (FM = 4, FN = 4) epilogue of the GEMM kernels; the two halves of a shared slot
// hand the even wave.s values to the odd one through LDS, which sums in the GEMM epilogue.s own order (one barrier), so the pairs come out bit-identical too.
#pragma once
#include "gemm_kernel.h"
#include <utility>

namespace ivit {

using MlpTile = GemmTile<1, 8, 4, 4>;   // shape constants for the shared epilogues: one wave = 4 x 4 fragments = 64 rows x 64 columns

#ifndef IVIT_MLPF_R
#define IVIT_MLPF_R 8       // weight blocks (KiB) in flight per wave at D = 768
#endif
#ifndef IVIT_MLPF_EPI_G
#define IVIT_MLPF_EPI_G 4   // fragment rows whose residual loads are in flight together in the final epilogue (4 = the whole slot)
#endif
template <int ND_, int S1_, int S2_>
struct MlpFusedGeom {
    static constexpr int ND = ND_, S1 = S1_, S2 = S2_;   // S1 / S2 = 2: W1' / W2 are hi / lo pairs of 16-bit values (K-tiles interleaved), else 1
    static constexpr int BM = 64, HC = 128, THREADS = 512;
    static constexpr int P = ND * S1;               // phase-1 steps per chunk (one 64-deep K-tile of W1' each; split: hi / lo alternate)
    static constexpr int NF = ND / 2;               // output column fragments per wave (D / 8 columns)
    static constexpr int NH = NF - 4;               // ... of which the last NH belong to the shared slot (0 at D = 512, 2 at D = 768)
    static constexpr int P2 = 2 * NF;               // phase-2 steps per chunk: two passes (hidden halves of 64) over the NF column fragments
    static constexpr int NB1 = 2 * P, NB2 = 2 * S2 * P2;  // 1-KiB fragment blocks per wave and chunk
    static constexpr int CB = NB1 + NB2;            // = 48 (96 split) at D = 768
    static constexpr int R = (ND == 12 ? IVIT_MLPF_R : 8);   // blocks in flight per wave (4 VGPRs each)
    static constexpr int X_BYTES = ND * 8192, U_BYTES = 16384, LDS_BYTES = X_BYTES + 2 * U_BYTES;
    static constexpr int UC = P - 3;                // phase-1 step that loads the chunk's fold vectors
    static_assert(ND == 8 || ND == 12, "D = 512 or 768: whole statistics slots per wave plus at most half of a shared one");
    static_assert(CB % R == 0, "the register ring wraps at the chunk boundary");
};

// where block b of wave w's stream for chunk c comes from (host + device: the pack kernel and its tests)
// phase 1 (b < NB1): K-tile s = b / 2 of W1' (split: K-tiles alternate hi / lo), 32-deep half kk = b % 2: rows c * 128 + 16 w + [0, 16)
// phase 2: hidden half h (64 of the chunk's 128; split: hidden K-tile 2 c + h), column fragment t, q: unsplit k-step q of 32 hidden; split hi (q < 2) / lo, kk = q & 1
template <int ND, int S1, int S2>
__host__ __device__ inline void mlpf_block_source(int c, int w, int b, int lane, bool* is_w1, int* row, int* col) {
    using G = MlpFusedGeom<ND, S1, S2>;
    const int r16 = lane & 15, kc = lane >> 4;
    if (b < G::NB1) {
        *is_w1 = true;
        *row = c * 128 + 16 * w + r16;
        *col = (b >> 1) * 64 + (b & 1) * 32 + 8 * kc;
    } else {
        const int b2 = b - G::NB1, h = b2 / (G::NF * 2 * S2), rem = b2 % (G::NF * 2 * S2), t = rem / (2 * S2), q = rem % (2 * S2);
        *is_w1 = false;
        *row = (t < 4 ? 64 * w + 16 * t : 64 * (8 + (w >> 1)) + 32 * (w & 1) + 16 * (t - 4)) + r16;
        *col = S2 == 1 ? c * 128 + 64 * h + 32 * q + 8 * kc : (2 * c + h) * 128 + (q >> 1) * 64 + (q & 1) * 32 + 8 * kc;
    }
}

// compile-time loop: f(std::integral_constant<int, 0>) ... f(std::integral_constant<int, N - 1>)
template <class F, int... I>
__device__ __forceinline__ void mlpf_static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void mlpf_static_for(F&& f) { mlpf_static_for_impl(f, std::make_integer_sequence<int, N>{}); }

#ifndef IVIT_MLPF_TOUCH
#define IVIT_MLPF_TOUCH 1   // 0 (A/B builds): no L2 warm-up loads (one dword per lane and chunk of the NEXT chunk's stream)
#endif

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

// DBG (tools/mlp_fused_bench only; 0 in the product): timing ablations, a bit mask - 1 = no weight loads inside the loop, 2 = no MFMA,
// 4 = no LDS fragment reads (results are wrong in every ablation; only the time is read)
template <int ND, int S1, int S2, class OP, int DBG = 0>
__device__ __forceinline__ void mlp_fused_body(const MlpFusedParams& p, char* smem) {
    using G = MlpFusedGeom<ND, S1, S2>;
    using T = MlpTile;
    constexpr int P = G::P, P2 = G::P2, NB1 = G::NB1, CB = G::CB, R = G::R, NH = G::NH, UC = G::UC;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    const int m0 = blockIdx.x * G::BM;
    char* ubuf = smem + G::X_BYTES;
    IVIT_MLPF_STAMP(0);

    // ---- the weight stream of this wave: block b of chunk cc at Wp + ((cc * 8 + wave) * CB + b) KiB, lane l its 16 bytes at l * 16
    const char* wp = reinterpret_cast<const char*>(p.Wp) + (size_t)wave * CB * 1024;
    const unsigned lane16 = (unsigned)lane * 16u;
    auto ldw = [&](int cc, int b) -> bf16x8 {   // b: compile-time constant wherever this is called
        const char* sb = wp + (size_t)cc * (8 * CB * 1024) + b * 1024;
        return *reinterpret_cast<const bf16x8*>(sb + lane16);
    };
    bf16x8 wq[R];   // the register ring: block b lives in wq[b % R]
#pragma unroll
    for (int b = 0; b < R; ++b) wq[b] = ldw(0, b);
    // this lane's line of a chunk's 8 CB KiB for the L2 warm-up loads (see the phase-1 epilogue)
    const char* touch_base = reinterpret_cast<const char*>(p.Wp) + (size_t)((((int)blockIdx.x >> 3) * 512 + (int)threadIdx.x) % (64 * CB)) * 128;
    unsigned touch = 0;
    if (IVIT_MLPF_TOUCH && p.Mlp > G::HC) touch = *reinterpret_cast<const unsigned*>(touch_base + (size_t)(8 * CB * 1024));   // chunk 1 (chunk 0 is being loaded)

    // ---- X rows -> LDS by LDS-DMA (ND K-tile images of 64 rows x 128 B, 16-B chunk XOR-swizzled with the row on the source side)
    {
        const int r_in = lane >> 3, swz = (lane & 7) ^ r_in;
        const char* x_lane = reinterpret_cast<const char*>(p.X) + (size_t)(m0 + wave * 8 + r_in) * ((size_t)p.ldx * 2) + swz * 16;
#pragma unroll
        for (int kt = 0; kt < ND; ++kt)
            __builtin_amdgcn_global_load_lds((const IVIT_GLOBAL void*)(x_lane + kt * 128), (IVIT_LDS void*)(smem + kt * 8192 + wave * 1024), 16, 0, 0);
    }
    // (mean, rstd) of the 64 rows from the statistics pairs (the fold every _lf GEMM does): one thread per row into the second U buffer, then every
    // lane takes the four rows of its accumulator fragments.  Rows past M: a valid row's statistics (their outputs are never stored).
    float mean_i[4], rstd_i[4];
    {
        float2* st = reinterpret_cast<float2*>(ubuf + G::U_BYTES);
        if (threadIdx.x < 64) st[threadIdx.x] = ln_row_stats_from_pairs(p.ln_part_in, min(m0 + (int)threadIdx.x, p.M - 1), ND * 64, p.ln_eps);
        // X has landed (every wave drains its own pieces), the statistics are written.  A builtin, not inline asm: hipcc's own wait bookkeeping must
        // see that the LDS-DMA is retired, or it drains the weight stream at every chunk for fear of it (cdna_hip_programming.md, 3 .s-level traps (b))
        __builtin_amdgcn_s_waitcnt(0x0070);   // vmcnt(0) lgkmcnt(0)
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int i = 0; i < 4; ++i) { const float2 v = st[16 * i + fr]; mean_i[i] = v.x; rstd_i[i] = v.y; }
    }

    // ---- LDS read offsets (row r, 16-B chunk q) -> r * 128 + ((q ^ (r & 7)) << 4); kk = 1 flips bit 2 of the chunk; fragment row i adds 2048
    const int a0 = (fq ^ (fr & 7)) << 4;
    const int off_x[2] = {fr * 128 + a0, fr * 128 + (a0 ^ 64)};
    auto ld16 = [](const char* q) { return *reinterpret_cast<const bf16x8*>(q); };

    constexpr int NHA = NH > 0 ? NH : 1;
    f32x4 accF[4][4];     // the wave's own statistics slot: rows 16 i + fr, columns 64 w + 16 j + 4 fq ..
    f32x4 accH[4][NHA];   // its half of the shared slot 8 + w / 2: columns 64 (8 + w / 2) + 32 (w & 1) + 16 j + 4 fq ..
    f32x4 acc1[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) accF[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < NHA; ++j) accH[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        acc1[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    bf16x8 xf[2][4][2] = {};   // X fragments [set = K-tile & 1][i][kk]: the next K-tile's reads return under this K-tile's MFMAs
    bf16x8 uf[4][2] = {};      // U fragments of the current hidden half [i][k-step of 32 hidden]
    float4 c4 = make_float4(0.f, 0.f, 0.f, 0.f), s4 = c4;   // fold vectors of the chunk's hidden columns 16 w + 4 fq ..
    // centred operand rows (kernels.h: GemmParams::ln_d): the chunk's up-product accumulators start from d1 instead of zero (gemm_acc_init); the next
    // chunk's four values are fetched during phase 2, where fewer registers are live
    float4 dn = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.d1) dn = *reinterpret_cast<const float4*>(p.d1 + 16 * wave + 4 * fq);
    auto read_x = [&](int kt, bf16x8 (&f)[4][2]) {
        if (DBG & 4) return;
        const char* b = smem + kt * 8192;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 4; ++i) f[i][kk] = ld16(b + off_x[kk] + i * 2048);
    };
    read_x(0, xf[0]);
    IVIT_MLPF_STAMP(1);

    const int nchunks = p.Mlp / G::HC;
    for (int c = 0; c < nchunks; ++c) {
        const int cn = c + 1 < nchunks ? c + 1 : 0;   // the last chunk's look-ahead re-loads chunk 0: valid memory, never used
        // ---------------- phase 1: acc1 = X . W1'[chunk]^T, one K-tile per step (split: hi, then lo against the same X fragments)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc1[i] = f32x4{dn.x, dn.y, dn.z, dn.w};   // (set here, not after its epilogue: dead, not live, through phase 2)
        mlpf_static_for<P>([&](auto s_tag) {
            constexpr int s = decltype(s_tag)::value, kt = s / S1;
            if constexpr (s + 1 < P && (s + 1) % S1 == 0) read_x(kt + 1, xf[(kt + 1) & 1]);   // the next K-tile's fragments return under this step's MFMAs
            if constexpr (s == UC) {
                c4 = *reinterpret_cast<const float4*>(p.c1 + c * 128 + 16 * wave + 4 * fq);
                s4 = *reinterpret_cast<const float4*>(p.s1 + c * 128 + 16 * wave + 4 * fq);
            }
            if constexpr (!(DBG & 2)) {
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc1[i] = OP::mfma(wq[(2 * s + kk) % R], xf[kt & 1][i][kk], acc1[i]);
            } else {
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    asm volatile("" ::"v"(wq[(2 * s + kk) % R]));
#pragma unroll
                    for (int i = 0; i < 4; ++i) asm volatile("" ::"v"(xf[kt & 1][i][kk]));
                }
            }
            if constexpr (!(DBG & 1)) {   // refill the two register slots just consumed: blocks R ahead in the stream.  (Issuing the refill of the PREVIOUS
                                          // step's slots first in the step instead, ahead of this step's wait, measured 6 % slower: profiles/r05_fused_mlp.txt)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) wq[(2 * s + kk) % R] = ldw(c, 2 * s + kk + R);   // < CB: phase 2 follows in the same chunk
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        // ---------------- phase-1 epilogue (gemm_epilogue_lnfold's arithmetic): u = rn16(gelu(rstd (acc - mean s) + c)), 4 consecutive hidden per lane
        {
            // L2 warm-up one chunk ahead.  In a forward every layer's 9.4 MB of weights are cold (twelve layers of them do not stay in L2), and the workgroups that
            // share an XCD (observed: blockIdx mod 8; speed only) ask for the same lines at almost the same time: all of them wait out the miss of the first.
            // One dword per lane of the NEXT chunk's block (its 128-byte lines dealt over the workgroups of the XCD group) starts those misses half a chunk
            // early.  Placed HERE because vector-memory results return in order: a load that misses to HBM holds up the wave's wait for every younger load,
            // and behind this point come ~0.6 us of GELU arithmetic and four steps on operands that are already in flight.
            if (IVIT_MLPF_TOUCH) touch = *reinterpret_cast<const unsigned*>(touch_base + (size_t)cn * (8 * CB * 1024));
            char* ub = ubuf + (c & 1) * G::U_BYTES + (wave >> 2) * 8192 + (fq & 1) * 8;
            const int ch = 2 * (wave & 3) + (fq >> 1);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float mu = mean_i[i], rs = rstd_i[i];
                float a[4] = {fmaf(rs, fmaf(-mu, s4.x, acc1[i][0]), c4.x), fmaf(rs, fmaf(-mu, s4.y, acc1[i][1]), c4.y),
                              fmaf(rs, fmaf(-mu, s4.z, acc1[i][2]), c4.z), fmaf(rs, fmaf(-mu, s4.w, acc1[i][3]), c4.w)};
#pragma unroll
                for (int r = 0; r < 4; ++r) a[r] = gelu_erf(a[r]);
                const int row = 16 * i + fr;
                u32x2 pk = {OP::pack2(a[0], a[1]), OP::pack2(a[2], a[3])};
                *reinterpret_cast<u32x2*>(ub + row * 128 + ((ch ^ (row & 7)) << 4)) = pk;
            }
        }
        // U written by every wave -> published.  (The other U buffer was last read two chunks ago, before every wave's previous barrier: no second barrier.)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // ---------------- phase 2: acc2[:, fragment t] += U[:, half h] . W2[fragment t rows, half h of the chunk]^T, h = 0, 1 (k ascending per accumulator)
        auto read_u = [&](int h) {
            if (DBG & 4) return;
            const char* b = ubuf + (c & 1) * G::U_BYTES + h * 8192;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) uf[i][kk] = ld16(b + off_x[kk] + i * 2048);
        };
        read_u(0);
        mlpf_static_for<P2>([&](auto v_tag) {
            constexpr int v = decltype(v_tag)::value, t = v % G::NF, NQ = 2 * S2, b0 = NB1 + NQ * v;
            if constexpr (v == G::NF) {
                read_u(1);
                if (p.d1) dn = *reinterpret_cast<const float4*>(p.d1 + cn * 128 + 16 * wave + 4 * fq);
            }
            if constexpr (v == P2 - 1) {
                read_x(0, xf[0]);   // the next chunk's first K-tile
                if (IVIT_MLPF_TOUCH) asm volatile("" ::"v"(touch));   // (the value is never used; the load is long complete)
            }
            if constexpr (!(DBG & 2)) {
#pragma unroll
                for (int q = 0; q < NQ; ++q)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        // unsplit: the two k-steps of 32 hidden; split: hi (kk 0, 1) then lo (kk 0, 1) of the hidden K-tile
                        if constexpr (t < 4) accF[i][t] = OP::mfma(wq[(b0 + q) % R], uf[i][q & 1], accF[i][t]);
                        else accH[i][t - 4] = OP::mfma(wq[(b0 + q) % R], uf[i][q & 1], accH[i][t - 4]);
                    }
            } else {
#pragma unroll
                for (int q = 0; q < NQ; ++q) asm volatile("" ::"v"(wq[(b0 + q) % R]));
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk) asm volatile("" ::"v"(uf[i][kk]));
            }
            if constexpr (!(DBG & 1)) {
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    if (b0 + q + R < CB) wq[(b0 + q) % R] = ldw(c, b0 + q + R);
                    else wq[(b0 + q) % R] = ldw(cn, b0 + q + R - CB);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        });
    }
    IVIT_MLPF_STAMP(2);
#pragma unroll
    for (int b = 0; b < R; ++b) asm volatile("" ::"v"(wq[b]));   // the look-ahead past the last chunk is waited for, not dropped mid-flight
    IVIT_MLPF_STAMP(4);

    // ---------------- residual epilogue of the MLP-down GEMM
    // (the row base and the lane's fragment coordinates re-enter through an opaque statement: hipcc otherwise computes the epilogue's row addresses ABOVE the
    // loop and keeps - or spills - them across it)
    int m0e = m0, fre = fr, fqe = fq;
    asm volatile("" : "+s"(m0e), "+v"(fre), "+v"(fqe));
    GemmParams gp{};
    gp.M = p.M; gp.N = p.D; gp.bias = p.b2; gp.resid = p.resid; gp.ldr = p.ldr; gp.out = p.out; gp.ldo = p.ldo;
    gp.xb = p.xb; gp.ldxb = p.ldxb; gp.ln_part = p.ln_part_out; gp.ln_centre = p.centre_out;
    gp.epi = p.stats_out ? EPI_BIAS_RESID_STATS : EPI_BIAS_RESID_F32;
    // residual rows of the shared half slot first: they are in flight under the whole-slot epilogue below (the epilogue is a burst of dependent
    // memory round trips on a CU with nothing else resident: everything that can be requested at once is)
    constexpr int NHL = NH > 0 ? 2 : 1;
    const int hslot = 8 + (wave >> 1), odd = wave & 1, n0 = 64 * hslot + 32 * odd;
    float4 xh[4][NHL], bias_h[NHL];
    bool row_ok[4];
    if constexpr (NH > 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0e + 16 * i + fre;
            row_ok[i] = m < p.M;
            const float* rs = p.resid + (size_t)(row_ok[i] ? m : p.M - 1) * p.ldr + n0 + 4 * fqe;
#pragma unroll
            for (int j = 0; j < 2; ++j) xh[i][j] = *reinterpret_cast<const float4*>(rs + 16 * j);
        }
    }
    // the wave's own slot: exactly the (FM = 4, FN = 4) epilogue of the GEMM kernels
    if (p.stats_out) gemm_epilogue_family<T, 1, OP, IVIT_MLPF_EPI_G>(gp, accF, m0e, 64 * wave, fre, fqe);
    else gemm_epilogue_family<T, 3, OP, IVIT_MLPF_EPI_G>(gp, accF, m0e, 64 * wave, fre, fqe);
    IVIT_MLPF_STAMP(5);
    if constexpr (NH > 0) {
        // the shared slot 8 + w / 2: the even wave holds its fragments j = 0, 1, the odd wave j = 2, 3.  Same element arithmetic; the slot's statistics below.
        static_assert(NH == 0 || NH == 2, "half of a four-fragment slot");
#pragma unroll
        for (int j = 0; j < 2; ++j) bias_h[j] = *reinterpret_cast<const float4*>(p.b2 + n0 + 16 * j + 4 * fqe);
        // rn16(x - centre) where the next LayerNorm's operand rows are centred (kernels.h: ln_centre); requested ahead of the rows' stores (loads and stores
        // return through one in-order counter)
        float4 ca = make_float4(0.f, 0.f, 0.f, 0.f), cb = ca;
        if (p.stats_out && p.centre_out) {
            ca = *reinterpret_cast<const float4*>(p.centre_out + n0 + 4 * fqe);
            cb = *reinterpret_cast<const float4*>(p.centre_out + n0 + 16 + 4 * fqe);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int mr = row_ok[i] ? m0e + 16 * i + fre : p.M - 1;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const float v[4] = {xh[i][j].x + (accH[i][j][0] + bias_h[j].x), xh[i][j].y + (accH[i][j][1] + bias_h[j].y),
                                    xh[i][j].z + (accH[i][j][2] + bias_h[j].z), xh[i][j].w + (accH[i][j][3] + bias_h[j].w)};
#pragma unroll
                for (int r = 0; r < 4; ++r) accH[i][j][r] = v[r];
                if (row_ok[i]) *reinterpret_cast<float4*>(p.out + (size_t)mr * p.ldo + n0 + 16 * j + 4 * fqe) = make_float4(v[0], v[1], v[2], v[3]);
            }
            if (p.stats_out) {   // 16-bit copy: 16-byte stores through the fragment-pair lane swap, as the GEMM epilogue (the swap itself in every lane)
                const auto lo = __builtin_amdgcn_permlane16_swap(OP::pack2(accH[i][0][0] - ca.x, accH[i][0][1] - ca.y), OP::pack2(accH[i][1][0] - cb.x, accH[i][1][1] - cb.y), false, false);
                const auto hi = __builtin_amdgcn_permlane16_swap(OP::pack2(accH[i][0][2] - ca.z, accH[i][0][3] - ca.w), OP::pack2(accH[i][1][2] - cb.z, accH[i][1][3] - cb.w), false, false);
                u32x4 pk = {lo[0], hi[0], lo[1], hi[1]};
                if (row_ok[i]) *reinterpret_cast<u32x4*>(p.xb + (size_t)mr * p.ldxb + n0 + (fqe & 1) * 16 + (fqe & ~1) * 4) = pk;
            }
        }
        IVIT_MLPF_STAMP(6);
        if (p.stats_out) {   // (uniform branch: every wave takes the same barrier)
            // The slot's row sum and M2 are the GEMM epilogue's sequential lane sums over the 16 values a lane would hold (j = 0..3, then r), i.e. the even wave's
            // 8 values then the odd wave's.  The even wave hands its 8 values per row to its partner through LDS - quads at [(i, j)][lane]: the partner's same lane
            // holds the same (row, column quad) - and the ODD wave evaluates both statistics alone, in that order: one barrier (round 5a chained sum -> mean -> M2
            // between the two waves through three: 4.9 us of the 15.6 us epilogue).  The scratch is X's K-tiles 1 .. 4: dead since the last chunk's phase 1 (a wave still
            // in that chunk's phase 2 reads only U and, for the look-ahead, X's K-tile 0).
            float4* xq = reinterpret_cast<float4*>(smem + 8192 + (wave >> 1) * 8192);
            if (!odd) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) xq[(i * 2 + j) * 64 + lane] = make_float4(accH[i][j][0], accH[i][j][1], accH[i][j][2], accH[i][j][3]);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (odd) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float4 e0 = xq[(i * 2 + 0) * 64 + lane], e1 = xq[(i * 2 + 1) * 64 + lane];
                    const float ev[8] = {e0.x, e0.y, e0.z, e0.w, e1.x, e1.y, e1.z, e1.w};
                    float s_ = 0.f;
#pragma unroll
                    for (int k = 0; k < 8; ++k) s_ += ev[k];
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int r = 0; r < 4; ++r) s_ += accH[i][j][r];
                    s_ += __shfl_xor(s_, 16, 64);
                    s_ += __shfl_xor(s_, 32, 64);
                    const float lm = s_ / 64.0f;
                    float q2 = 0.f;
#pragma unroll
                    for (int k = 0; k < 8; ++k) { const float d = ev[k] - lm; q2 = fmaf(d, d, q2); }
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int r = 0; r < 4; ++r) { const float d = accH[i][j][r] - lm; q2 = fmaf(d, d, q2); }
                    q2 += __shfl_xor(q2, 16, 64);
                    q2 += __shfl_xor(q2, 32, 64);
                    if (fqe == 0 && row_ok[i]) p.ln_part_out[(size_t)(m0e + 16 * i + fre) * GEMM_LN_SLOTS + hslot] = make_float2(s_, q2);
                }
            }
        }
    }
    IVIT_MLPF_STAMP(3);
}

}  // namespace ivit
