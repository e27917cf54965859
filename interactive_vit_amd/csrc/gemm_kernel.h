// bf16 MFMA GEMM kernel template for the dense projections of the ViT forward (patch embedding, QKV,
// attention out-projection, MLP up/down, classifier head):
//
//     C[M,N] = A[M,K] . W[N,K]^T   (+ fused epilogue: bias, exact GELU, residual add, row remap)
//
// Both operands are K-contiguous ("B^T" form), which is exactly the per-lane fragment shape of
// v_mfma_f32_16x16x32_bf16 (lane l holds 8 consecutive k of row l&15).  The product is issued
// with W as the MFMA "A" operand and the activations as "B", so an accumulator register quad holds
// FOUR CONSECUTIVE n of one output row: epilogue loads/stores are 8 B (bf16) / 16 B (f32) per lane.
//
// Tile = (WAVES_M * FM * 16) x (WAVES_N * FN * 16) x 64; each wave owns FM x FN 16x16 fragments.
// Operand tiles are staged global -> LDS with global_load_lds_dwordx4 (no VGPR round trip), double
// buffered: the loads of K-step t+1 are in flight while step t is multiplied.  The LDS image is
// lane-linear [row][128 B]; the 16-B chunk index is XOR-swizzled with (row & 7) on the SOURCE
// address and on the ds_read_b128 address, which makes every fragment read bank-conflict free
// (cdna_hip_programming.md T2 / rule 21).
#pragma once
#include "kernels.h"
#include <type_traits>

namespace ivit {

constexpr int GEMM_BK = 64;
#ifndef IVIT_RSG_2STAGE
#define IVIT_RSG_2STAGE 1   // fragment rows per residual-load group in the two-stage kernels' _rs / _f32 epilogues.  Inside the ViT-B/16 B = 64 forward
                            // (bench.py --steps 100, same box, alternating, medians): every row of the tile at once (T::FM) 21 531 img/s, round 3's row-by-row
                            // form 21 619, two rows ahead 21 695, ONE row ahead 21 791 - the microbenchmark ranks them the other way round
                            // (profiles/r04_epilogue_ab.txt): with the stream coming from the Infinity Cache, 40 MB of requests at once queue badly
#endif
#ifndef IVIT_LN_STATS_AFTER_DMA
#define IVIT_LN_STATS_AFTER_DMA 1   // _lf kernels (three-per-CU and 256 x 256 tiles): the tile's statistics pairs are loaded and folded BEHIND the first operand DMA
                                    // (round 4, in the forward: mlp1 76.8 -> 75.9 us, qkv 55.7 -> 55.1; profiles/r04_epilogue_ab.txt (f)); 0 = in front of it, as round 3
#endif
#ifndef IVIT_ASHIFT_REUSE
#define IVIT_ASHIFT_REUSE 1   // 0 (A/B builds): stage and read the activation K-tile for both K-tiles of a hi / lo weight pair, as round 3 did.
                              // Only the f16 instantiations carry the branch (weight pairs exist on the f16 data path only): on the bf16 kernels it
                              // measured -0.4 % of the headline step (profiles/r04_f16x_a_tile_reuse.txt)
#endif

template <int WAVES_M_, int WAVES_N_, int FM_, int FN_>
struct GemmTile {
    static constexpr int WAVES_M = WAVES_M_, WAVES_N = WAVES_N_, FM = FM_, FN = FN_;
    static constexpr int WAVES = WAVES_M * WAVES_N;
    static constexpr int THREADS = WAVES * 64;
    static constexpr int BM = WAVES_M * FM * 16;
    static constexpr int BN = WAVES_N * FN * 16;
    static constexpr int A_BYTES = BM * GEMM_BK * 2;
    static constexpr int W_BYTES = BN * GEMM_BK * 2;
    static constexpr int STAGE_BYTES = A_BYTES + W_BYTES;
    static constexpr int LDS_BYTES = 2 * STAGE_BYTES;
    static constexpr bool RAGGED_N = false;   // N % 4 == 0 required (edge tiles guard whole column quads); the deep-ring tile takes any N
    static constexpr int A_PIECES = BM / 8;   // 1-KiB DMA pieces (8 rows x 128 B) per A tile
    static constexpr int W_PIECES = BN / 8;
    static_assert(BM % 8 == 0 && BN % 8 == 0, "tile rows must be whole 8-row DMA pieces");
};

// Issue the global->LDS copies of one ROWS x 64 bf16 tile: each wave-instruction covers 8 rows x
// 128 B.  LDS slot (row r, chunk c) receives global chunk c ^ (r & 7).
// (byte addressing: the same code stages bf16 tiles of 64 k and fp8 tiles of 128 k - both 128-B rows)
template <int PIECES, int WAVES>
__device__ __forceinline__ void stage_tile(const void* __restrict__ g, size_t ld_bytes, int row0, int k0_bytes,
                                           char* lds_tile, int wave, int lane) {
    const int r_in = lane >> 3;            // row inside the 8-row piece
    const int chunk = (lane & 7) ^ r_in;   // (r_local & 7) == r_in because pieces are 8-row aligned
    const char* src = reinterpret_cast<const char*>(g) + (size_t)(row0 + r_in) * ld_bytes + k0_bytes + chunk * 16;
#pragma unroll
    for (int i = 0; i < (PIECES + WAVES - 1) / WAVES; ++i) {
        const int piece = i * WAVES + wave;
        if (PIECES % WAVES == 0 || piece < PIECES)
            __builtin_amdgcn_global_load_lds((const IVIT_GLOBAL void*)(src + (size_t)piece * 8 * ld_bytes),
                                             (IVIT_LDS void*)(lds_tile + piece * 1024), 16, 0, 0);
    }
}

// v_mfma_scale_f32_16x16x128_f8f6f4 with e4m3 operands and unit block scales (e8m0 0x7f = 2^0): the
// gfx950 fp8 form that runs at twice the bf16 rate (the non-scaled 16x16x32 fp8 MFMA runs at the bf16
// rate).  w2 / a2 = the two 16-B k-chunks a lane holds of its W / A row; first source = W, so the
// accumulator quad is C[m = lane & 15][n = (lane >> 4) * 4 + r] as for the bf16 form.  Semantics
// checked on hardware by tools/mfma_f8_probe.hip.
__device__ __forceinline__ f32x4 mfma_e4m3_16x16x128(const bf16x8 (&w2)[2], const bf16x8 (&a2)[2], f32x4 c) {
    typedef __attribute__((ext_vector_type(4))) int i32x4_t;
    typedef __attribute__((ext_vector_type(8))) int i32x8_t;
    const i32x8_t w = __builtin_shufflevector(__builtin_bit_cast(i32x4_t, w2[0]), __builtin_bit_cast(i32x4_t, w2[1]), 0, 1, 2, 3, 4, 5, 6, 7);
    const i32x8_t a = __builtin_shufflevector(__builtin_bit_cast(i32x4_t, a2[0]), __builtin_bit_cast(i32x4_t, a2[1]), 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(w, a, c, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
}

// K-tile of A that K-tile t of the product reads (GemmParams::a_wrap: at most three passes over A)
__device__ __forceinline__ int a_ktile(const GemmParams& p, int t) {
    t >>= p.a_shift;   // weight-only split: K-tiles 2t and 2t + 1 of the product share A's K-tile t
    if (p.a_wrap) {
        if (t >= p.a_wrap) t -= p.a_wrap;
        if (t >= p.a_wrap) t -= p.a_wrap;
    }
    return t;
}

__device__ __forceinline__ bf16x8 read_frag(const char* lds_tile, int r_local, int q) {
    const int off = r_local * 128 + ((q ^ (r_local & 7)) << 4);
    return *reinterpret_cast<const bf16x8*>(lds_tile + off);
}

// Epilogue.  acc[i][j][r] = C[m_base + i*16 + fr][n_base + j*16 + fq*4 + r].
// Interior tiles (every row < M, every column quad < N - the wave-uniform common case) take a
// branch-free path: the bias quads are loaded once per wave, and with no control flow between them
// the residual loads / stores of many fragments are in flight together.  (With a branch per
// fragment hipcc emitted load -> s_waitcnt vmcnt(0) -> store 32 times in a row, each paying a full
// memory latency: ~10 us per 256x256 tile.)  Edge tiles take the guarded path.
// KIND (round 4: the run-time choice among ALL classic kinds kept bias, column scales, addend rows and both store forms alive together and
// spilled 160-190 bytes per lane in the 256 x 256 and three-per-CU tiles): 0 = the 16-bit / fp8 output kinds (EPI_BIAS_BF16,
// EPI_BIAS_GELU_BF16, EPI_BIAS_GELU_FP8), 3 = the f32 output kinds (EPI_BIAS_F32, EPI_BIAS_RESID_F32, EPI_BIAS_ROWADD_F32) - separate
// kernel instantiations ("_f32"); within a kind the choice stays a wave-uniform run-time branch.
template <class T, bool INTERIOR, class OP = OpBf16, int KIND = -1>
__device__ __forceinline__ void gemm_epilogue_impl(const GemmParams& p, f32x4 (&acc)[T::FM][T::FN], int m_base,
                                                   int n_base, int fr, int fq) {
    const int epi = p.epi;
    constexpr bool K16 = KIND != 3, K32 = KIND != 0;   // KIND = -1 (microbenchmark-only study kernels): every classic kind in one kernel, as before
    float4 bias4[T::FN];
#pragma unroll
    for (int j = 0; j < T::FN; ++j) {
        const int n = n_base + j * 16 + fq * 4;
        if (INTERIOR || n + 3 < p.N) {
            bias4[j] = *reinterpret_cast<const float4*>(p.bias + n);
        } else {
            float b[4] = {0.f, 0.f, 0.f, 0.f};
            for (int r = 0; r < 4; ++r) if (n + r < p.N) b[r] = p.bias[n + r];
            bias4[j] = make_float4(b[0], b[1], b[2], b[3]);
        }
    }
    if (p.colscale) {   // fp8 operands: dequantise the accumulators (wave-uniform branch, once per tile)
#pragma unroll
        for (int j = 0; j < T::FN; ++j) {
            const int n = n_base + j * 16 + fq * 4;
            float c[4] = {0.f, 0.f, 0.f, 0.f};
            for (int r = 0; r < 4; ++r) if (INTERIOR || n + r < p.N) c[r] = p.colscale[n + r];
#pragma unroll
            for (int i = 0; i < T::FM; ++i) { acc[i][j][0] = __fmul_rn(acc[i][j][0], c[0]); acc[i][j][1] = __fmul_rn(acc[i][j][1], c[1]); acc[i][j][2] = __fmul_rn(acc[i][j][2], c[2]); acc[i][j][3] = __fmul_rn(acc[i][j][3], c[3]); }   // a rounded product in every instantiation (never contracted with the bias add: interior and edge tiles must agree bit for bit)
        }
    }
#pragma unroll
    for (int i = 0; i < T::FM; ++i) {
        const int m = m_base + i * 16 + fr;
        if (!INTERIOR && m >= p.M) continue;
        int orow = m;
        int arow = 0;
        if (p.grp_in > 0) {
            const int grp = m / p.grp_in, within = m - grp * p.grp_in;
            orow = grp * p.grp_out + p.grp_off + within;
            arow = p.grp_off + within;
        }
        float4 extra[T::FN];   // residual / row-add operand of this row, all column fragments at once
        if (K32 && INTERIOR && (epi == EPI_BIAS_RESID_F32 || epi == EPI_BIAS_ROWADD_F32)) {
            const float* src = (epi == EPI_BIAS_RESID_F32) ? p.resid + (size_t)orow * p.ldr : p.rowadd + (size_t)arow * p.ldra;
#pragma unroll
            for (int j = 0; j < T::FN; ++j) extra[j] = *reinterpret_cast<const float4*>(src + n_base + j * 16 + fq * 4);
        }
        if (K16 && INTERIOR && (T::FN % 2 == 0) && (epi == EPI_BIAS_BF16 || epi == EPI_BIAS_GELU_BF16)) {
            // bf16 output, interior tile: 16-byte stores.  A lane holds 4 consecutive n per fragment
            // (8 B of bf16); v_permlane16_swap trades quads between the lane rows fq and fq^1 of a
            // fragment PAIR (j, j+1): even-fq lanes end up with 8 consecutive n of fragment j, odd-fq
            // lanes with 8 consecutive n of fragment j+1.  Same bytes, half the store instructions -
            // the 8-byte form is store-ISSUE-bound (~7 B/clk/CU: 7.5 us per 256x256 tile, stamps).
            // Semantics pinned on hardware by tools/permlane_probe.hip.
            bf16_t* orow_p = reinterpret_cast<bf16_t*>(p.out) + (size_t)orow * p.ldo;
#pragma unroll
            for (int j = 0; j < T::FN; j += 2) {
                float a[4] = {acc[i][j][0] + bias4[j].x, acc[i][j][1] + bias4[j].y, acc[i][j][2] + bias4[j].z, acc[i][j][3] + bias4[j].w};
                float b[4] = {acc[i][j + 1][0] + bias4[j + 1].x, acc[i][j + 1][1] + bias4[j + 1].y,
                              acc[i][j + 1][2] + bias4[j + 1].z, acc[i][j + 1][3] + bias4[j + 1].w};
                if (epi == EPI_BIAS_GELU_BF16) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) { a[r] = gelu_erf(a[r]); b[r] = gelu_erf(b[r]); }
                }
                const auto lo = __builtin_amdgcn_permlane16_swap(OP::pack2(a[0], a[1]), OP::pack2(b[0], b[1]), false, false);
                const auto hi = __builtin_amdgcn_permlane16_swap(OP::pack2(a[2], a[3]), OP::pack2(b[2], b[3]), false, false);
                const int n = n_base + (j + (fq & 1)) * 16 + (fq & ~1) * 4;
                u32x4 pk = {lo[0], hi[0], lo[1], hi[1]};
                *reinterpret_cast<u32x4*>(orow_p + n) = pk;
            }
            continue;
        }
#pragma unroll
        for (int j = 0; j < T::FN; ++j) {
            const int n = n_base + j * 16 + fq * 4;
            if (!INTERIOR && n >= p.N) continue;
            float v[4] = {acc[i][j][0] + bias4[j].x, acc[i][j][1] + bias4[j].y, acc[i][j][2] + bias4[j].z, acc[i][j][3] + bias4[j].w};
            const bool full = INTERIOR || (n + 3 < p.N);
            if (K16 && (epi == EPI_BIAS_GELU_BF16 || epi == EPI_BIAS_GELU_FP8)) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = gelu_erf(v[r]);
            }
            if (K16 && epi == EPI_BIAS_GELU_FP8) {   // 4 consecutive n -> one dword of e4m3
                unsigned char* o = reinterpret_cast<unsigned char*>(p.out) + (size_t)orow * p.ldo + n;
                const unsigned int pk = pack_fp8x4(v[0] * p.out_scale, v[1] * p.out_scale, v[2] * p.out_scale, v[3] * p.out_scale);
                if (full) *reinterpret_cast<unsigned int*>(o) = pk;
                else for (int r = 0; r < 4; ++r) if (n + r < p.N) o[r] = (unsigned char)(pk >> (8 * r));
            } else if (K16) {
                bf16_t* o = reinterpret_cast<bf16_t*>(p.out) + (size_t)orow * p.ldo + n;
                if (full) {
                    u32x2 pk = {OP::pack2(v[0], v[1]), OP::pack2(v[2], v[3])};
                    *reinterpret_cast<u32x2*>(o) = pk;
                } else {
                    for (int r = 0; r < 4; ++r) if (n + r < p.N) o[r] = OP::from_f32(v[r]);
                }
            } else {
                float* o = reinterpret_cast<float*>(p.out) + (size_t)orow * p.ldo + n;
                if (INTERIOR) {
                    if (epi == EPI_BIAS_RESID_F32) {           // resid + (acc + bias)
                        v[0] = extra[j].x + v[0]; v[1] = extra[j].y + v[1]; v[2] = extra[j].z + v[2]; v[3] = extra[j].w + v[3];
                    } else if (epi == EPI_BIAS_ROWADD_F32) {   // (acc + bias) + rowadd
                        v[0] += extra[j].x; v[1] += extra[j].y; v[2] += extra[j].z; v[3] += extra[j].w;
                    }
                } else if (epi == EPI_BIAS_RESID_F32) {
                    const float* rs = p.resid + (size_t)orow * p.ldr + n;
                    if (full) {
                        const float4 x = *reinterpret_cast<const float4*>(rs);
                        v[0] = x.x + v[0]; v[1] = x.y + v[1]; v[2] = x.z + v[2]; v[3] = x.w + v[3];
                    } else {
                        for (int r = 0; r < 4; ++r) if (n + r < p.N) v[r] = rs[r] + v[r];
                    }
                } else if (epi == EPI_BIAS_ROWADD_F32) {
                    const float* ra = p.rowadd + (size_t)arow * p.ldra + n;
                    if (full) {
                        const float4 x = *reinterpret_cast<const float4*>(ra);
                        v[0] += x.x; v[1] += x.y; v[2] += x.z; v[3] += x.w;
                    } else {
                        for (int r = 0; r < 4; ++r) if (n + r < p.N) v[r] += ra[r];
                    }
                }
                if (full) {
                    *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
                } else {
                    for (int r = 0; r < 4; ++r) if (n + r < p.N) o[r] = v[r];
                }
            }
        }
    }
}

// Edge tiles of the classic kinds where N is a multiple of 4 (every tile but the deep-ring one, which serves ragged widths: T::RAGGED_N): a
// lane's column quad lies wholly inside or wholly outside the matrix, so the guards are one per fragment row and one per column quad
// instead of one per element.  The per-element form of gemm_epilogue_impl<..., false> unrolled over 8 x 4 x 4 elements was what spilled
// (148-312 bytes per lane in the 256 x 256 and 160 x 128 kernels, round 3's whitelist in tests/test_abi.py); this one keeps every
// GEMM kernel of the library free of scratch.  Same arithmetic, element for element.
template <class T, class OP, int KIND>
__device__ __forceinline__ void gemm_epilogue_edge_quads(const GemmParams& p, f32x4 (&acc)[T::FM][T::FN], int m_base, int n_base, int fr, int fq) {
    const int epi = p.epi;
    constexpr bool K16 = KIND != 3, K32 = KIND != 0;
    int orow[T::FM], arow[T::FM];
    if (p.grp_in > 0) {
#pragma unroll
        for (int i = 0; i < T::FM; ++i) {
            const int m = min(m_base + i * 16 + fr, p.M - 1);
            const int grp = m / p.grp_in, within = m - grp * p.grp_in;
            orow[i] = grp * p.grp_out + p.grp_off + within;
            arow[i] = p.grp_off + within;
        }
    } else {
#pragma unroll
        for (int i = 0; i < T::FM; ++i) { orow[i] = m_base + i * 16 + fr; arow[i] = 0; }
    }
#pragma unroll
    for (int j = 0; j < T::FN; ++j) {
        const int n = n_base + j * 16 + fq * 4;
        if (n >= p.N) continue;                                  // the whole quad is outside
        const float4 b = *reinterpret_cast<const float4*>(p.bias + n);
        float4 c = make_float4(1.f, 1.f, 1.f, 1.f);
        if (p.colscale) c = *reinterpret_cast<const float4*>(p.colscale + n);
#pragma unroll
        for (int i = 0; i < T::FM; ++i) {
            if (m_base + i * 16 + fr >= p.M) continue;
            float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
            if (p.colscale) { v[0] = __fmul_rn(v[0], c.x); v[1] = __fmul_rn(v[1], c.y); v[2] = __fmul_rn(v[2], c.z); v[3] = __fmul_rn(v[3], c.w); }
            v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
            if (K16 && (epi == EPI_BIAS_GELU_BF16 || epi == EPI_BIAS_GELU_FP8)) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = gelu_erf(v[r]);
            }
            if (K16 && epi == EPI_BIAS_GELU_FP8) {
                *reinterpret_cast<unsigned int*>(reinterpret_cast<unsigned char*>(p.out) + (size_t)orow[i] * p.ldo + n) =
                    pack_fp8x4(v[0] * p.out_scale, v[1] * p.out_scale, v[2] * p.out_scale, v[3] * p.out_scale);
            } else if (K16 && (epi == EPI_BIAS_BF16 || epi == EPI_BIAS_GELU_BF16)) {
                u32x2 pk = {OP::pack2(v[0], v[1]), OP::pack2(v[2], v[3])};
                *reinterpret_cast<u32x2*>(reinterpret_cast<bf16_t*>(p.out) + (size_t)orow[i] * p.ldo + n) = pk;
            } else if (K32) {
                if (epi == EPI_BIAS_RESID_F32) {           // resid + (acc + bias)
                    const float4 x = *reinterpret_cast<const float4*>(p.resid + (size_t)orow[i] * p.ldr + n);
                    v[0] = x.x + v[0]; v[1] = x.y + v[1]; v[2] = x.z + v[2]; v[3] = x.w + v[3];
                } else if (epi == EPI_BIAS_ROWADD_F32) {   // (acc + bias) + rowadd
                    const float4 x = *reinterpret_cast<const float4*>(p.rowadd + (size_t)arow[i] * p.ldra + n);
                    v[0] += x.x; v[1] += x.y; v[2] += x.z; v[3] += x.w;
                }
                *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.out) + (size_t)orow[i] * p.ldo + n) = make_float4(v[0], v[1], v[2], v[3]);
            }
        }
    }
}

// The f32-output kinds (EPI_BIAS_RESID_F32 / EPI_BIAS_ROWADD_F32 / EPI_BIAS_F32) on interior tiles (round 4).  In gemm_epilogue_impl a
// fragment row's residual loads sit behind the previous row's stores (`out` may alias `resid`, so hipcc keeps the order and waits with
// vmcnt(0) for both, FM times per tile - eight times on the 256 x 256 tile that runs every out-projection / MLP-down GEMM of ViT-L / ViT-H).
// Here the addend rows are loaded G fragment rows at a time into a ping-pong register set, group g + 1 before group g's stores (see
// gemm_epilogue_resid_stats_interior).  Same arithmetic in the same order as gemm_epilogue_impl: bit-identical.
template <class T, int G>
__device__ __forceinline__ void gemm_epilogue_f32_interior(const GemmParams& p, f32x4 (&acc)[T::FM][T::FN], int m_base, int n_base, int fr, int fq) {
    constexpr int NG = (T::FM + G - 1) / G;
    const int epi = p.epi;
    float4 bias4[T::FN];
#pragma unroll
    for (int j = 0; j < T::FN; ++j) bias4[j] = *reinterpret_cast<const float4*>(p.bias + n_base + j * 16 + fq * 4);
    if (p.colscale) {   // fp8 operands: dequantise the accumulators (wave-uniform branch, once per tile)
#pragma unroll
        for (int j = 0; j < T::FN; ++j) {
            const float4 c = *reinterpret_cast<const float4*>(p.colscale + n_base + j * 16 + fq * 4);
#pragma unroll
            for (int i = 0; i < T::FM; ++i) { acc[i][j][0] = __fmul_rn(acc[i][j][0], c.x); acc[i][j][1] = __fmul_rn(acc[i][j][1], c.y); acc[i][j][2] = __fmul_rn(acc[i][j][2], c.z); acc[i][j][3] = __fmul_rn(acc[i][j][3], c.w); }
        }
    }
    int orow[T::FM], arow[T::FM];
    if (p.grp_in > 0) {   // row remap m -> (m / grp_in) * grp_out + grp_off + m % grp_in (one uniform branch, not one per fragment row)
#pragma unroll
        for (int i = 0; i < T::FM; ++i) {
            const int m = m_base + i * 16 + fr;
            const int grp = m / p.grp_in, within = m - grp * p.grp_in;
            orow[i] = grp * p.grp_out + p.grp_off + within;
            arow[i] = p.grp_off + within;
        }
    } else {
#pragma unroll
        for (int i = 0; i < T::FM; ++i) { orow[i] = m_base + i * 16 + fr; arow[i] = 0; }
    }
    if (epi == EPI_BIAS_F32) {   // nothing to load
#pragma unroll
        for (int i = 0; i < T::FM; ++i) {
            float* o = reinterpret_cast<float*>(p.out) + (size_t)orow[i] * p.ldo + n_base + fq * 4;
#pragma unroll
            for (int j = 0; j < T::FN; ++j)
                *reinterpret_cast<float4*>(o + j * 16) = make_float4(acc[i][j][0] + bias4[j].x, acc[i][j][1] + bias4[j].y, acc[i][j][2] + bias4[j].z, acc[i][j][3] + bias4[j].w);
        }
        return;
    }
    const float* src[T::FM];
#pragma unroll
    for (int i = 0; i < T::FM; ++i)
        src[i] = (epi == EPI_BIAS_ROWADD_F32 ? p.rowadd + (size_t)arow[i] * p.ldra : p.resid + (size_t)orow[i] * p.ldr) + n_base + fq * 4;
    float4 x[2][G][T::FN];
#pragma unroll
    for (int ii = 0; ii < G; ++ii)
#pragma unroll
        for (int j = 0; j < T::FN; ++j)
            if (ii < T::FM) x[0][ii][j] = *reinterpret_cast<const float4*>(src[ii] + j * 16);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        if (g + 1 < NG) {
#pragma unroll
            for (int ii = 0; ii < G; ++ii)
#pragma unroll
                for (int j = 0; j < T::FN; ++j)
                    if ((g + 1) * G + ii < T::FM) x[(g + 1) & 1][ii][j] = *reinterpret_cast<const float4*>(src[(g + 1) * G + ii] + j * 16);
        }
#pragma unroll
        for (int ii = 0; ii < G; ++ii) {
            const int i = g * G + ii;
            if (i >= T::FM) continue;
            float* o = reinterpret_cast<float*>(p.out) + (size_t)orow[i] * p.ldo + n_base + fq * 4;
#pragma unroll
            for (int j = 0; j < T::FN; ++j) {
                const float4 xr = x[g & 1][ii][j];
                *reinterpret_cast<float4*>(o + j * 16) = make_float4(xr.x + (acc[i][j][0] + bias4[j].x), xr.y + (acc[i][j][1] + bias4[j].y),
                                                                     xr.z + (acc[i][j][2] + bias4[j].z), xr.w + (acc[i][j][3] + bias4[j].w));
            }
        }
    }
}

// G: fragment rows per residual-load group of the f32-output kinds (the register budget of the calling kernel)
template <class T, class OP = OpBf16, int G = 2, int KIND = -1>
__device__ __forceinline__ void gemm_epilogue(const GemmParams& p, f32x4 (&acc)[T::FM][T::FN], int m_base, int n_base,
                                              int fr, int fq) {
    // wave-uniform: the wave's whole FM*16 x FN*16 patch lies inside the matrix
    const bool interior = (m_base + T::FM * 16 <= p.M) && (n_base + T::FN * 16 <= p.N);
    const bool f32_out = p.epi == EPI_BIAS_RESID_F32 || p.epi == EPI_BIAS_ROWADD_F32 || p.epi == EPI_BIAS_F32;
    if (KIND != 0 && interior && (KIND == 3 || f32_out)) gemm_epilogue_f32_interior<T, G>(p, acc, m_base, n_base, fr, fq);
    else if (interior) gemm_epilogue_impl<T, true, OP, (KIND == 3 ? -1 : KIND)>(p, acc, m_base, n_base, fr, fq);
    else if constexpr (KIND >= 0 && !T::RAGGED_N) gemm_epilogue_edge_quads<T, OP, KIND>(p, acc, m_base, n_base, fr, fq);   // N % 4 == 0 (launch_gemm_variant)
    else gemm_epilogue_impl<T, false, OP, KIND>(p, acc, m_base, n_base, fr, fq);
}

// ---- LayerNorm-fold epilogues (their own kernel instantiations: the classic kernels stay as they are).
// EPI_BIAS_RESID_STATS: residual add as EPI_BIAS_RESID_F32; additionally the bf16 copy of the new rows and, per row,
// the (sum, M2 about the local mean) of this wave's 64 columns -> ln_part[row][n_base / 64].  EPI_BIAS_ROWADD_STATS: the same with the
// addend taken from a table row and the output row remapped (grp_in / grp_out / grp_off as EPI_BIAS_ROWADD_F32; x + y = y + x bit for bit).
// (Round 3: the residual rows loaded at the START of the kernel into 80 registers, so that the epilogue only adds and stores, was measured and
// removed: vmcnt counts in issue order, so the K loop's first wait also waits for those loads - 38.7 MB from every CU at once, 8 us -
// and the launch got slower: proj 33 -> 49 us in a forward.)
template <class T, bool INTERIOR, class OP = OpBf16>
__device__ __forceinline__ void gemm_epilogue_resid_stats(const GemmParams& p, f32x4 (&acc)[T::FM][T::FN], int m_base, int n_base, int fr, int fq) {
    static_assert(T::FN * 16 == 64, "one statistics slot per wave column");
    const int slot = n_base >> 6;
    const int ncols = max(0, min(64, p.N - n_base));
    float4 bias4[T::FN];
#pragma unroll
    for (int j = 0; j < T::FN; ++j) {
        const int n = n_base + j * 16 + fq * 4;
        if (INTERIOR || n + 3 < p.N) bias4[j] = *reinterpret_cast<const float4*>(p.bias + n);
        else { float b[4] = {0.f, 0.f, 0.f, 0.f}; for (int r = 0; r < 4; ++r) if (n + r < p.N) b[r] = p.bias[n + r]; bias4[j] = make_float4(b[0], b[1], b[2], b[3]); }
    }
#pragma unroll
    for (int i = 0; i < T::FM; ++i) {
        const int m = m_base + i * 16 + fr;
        const bool row_ok = INTERIOR || m < p.M;
        int mr = row_ok ? m : p.M - 1;
        const float* rs;
        if (p.grp_in > 0) {   // EPI_BIAS_ROWADD_STATS: GEMM row -> token row of its image, the addend is the table row (position embedding)
            const int grp = mr / p.grp_in, within = mr - grp * p.grp_in;
            mr = grp * p.grp_out + p.grp_off + within;
            rs = p.rowadd + (size_t)(p.grp_off + within) * p.ldra;
        } else {
            rs = p.resid + (size_t)mr * p.ldr;
        }
        float4 x[T::FN];
#pragma unroll
        for (int j = 0; j < T::FN; ++j) {
            const int n = n_base + j * 16 + fq * 4;
            if (INTERIOR || n + 3 < p.N) x[j] = *reinterpret_cast<const float4*>(rs + n);
            else { float t[4] = {0.f, 0.f, 0.f, 0.f}; for (int r = 0; r < 4; ++r) if (n + r < p.N) t[r] = rs[n + r]; x[j] = make_float4(t[0], t[1], t[2], t[3]); }
        }
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < T::FN; ++j) {
            const int n = n_base + j * 16 + fq * 4;
            float v[4] = {x[j].x + (acc[i][j][0] + bias4[j].x), x[j].y + (acc[i][j][1] + bias4[j].y),
                          x[j].z + (acc[i][j][2] + bias4[j].z), x[j].w + (acc[i][j][3] + bias4[j].w)};
#pragma unroll
            for (int r = 0; r < 4; ++r) { acc[i][j][r] = v[r]; if (INTERIOR || n + r < p.N) sum += v[r]; }
            float* o = reinterpret_cast<float*>(p.out) + (size_t)mr * p.ldo + n;
            bf16_t* ob = p.xb + (size_t)mr * p.ldxb + n;
            // the 16-bit copy is rn16(x - centre[n]) (GemmParams::ln_centre; the statistics and the f32 stream are those of x itself).  Re-read per fragment
            // row from L1 rather than held across the rows: 16 more live registers spilled in the edge workgroup of the fused MLP kernel
            float ct[4] = {0.f, 0.f, 0.f, 0.f};
            if (p.ln_centre) for (int r = 0; r < 4; ++r) if (INTERIOR || n + r < p.N) ct[r] = p.ln_centre[n + r];
            if (INTERIOR || (row_ok && n + 3 < p.N)) {
                *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
                u32x2 pk = {OP::pack2(v[0] - ct[0], v[1] - ct[1]), OP::pack2(v[2] - ct[2], v[3] - ct[3])};
                *reinterpret_cast<u32x2*>(ob) = pk;
            } else if (row_ok) {
                for (int r = 0; r < 4; ++r) if (n + r < p.N) { o[r] = v[r]; ob[r] = OP::from_f32(v[r] - ct[r]); }
            }
        }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        const float lmean = ncols > 0 ? sum / (float)ncols : 0.f;
        float m2 = 0.f;
#pragma unroll
        for (int j = 0; j < T::FN; ++j) {
            const int n = n_base + j * 16 + fq * 4;
#pragma unroll
            for (int r = 0; r < 4; ++r) if (INTERIOR || n + r < p.N) { const float d = acc[i][j][r] - lmean; m2 = fmaf(d, d, m2); }   // explicit fma: the interior and the edge instantiation must round alike
        }
        m2 += __shfl_xor(m2, 16, 64);
        m2 += __shfl_xor(m2, 32, 64);
        if (fq == 0 && row_ok && ncols > 0) p.ln_part[(size_t)mr * GEMM_LN_SLOTS + slot] = make_float2(sum, m2);
    }
}

// Interior tiles of the same epilogue (round 4).  The form above issues, per fragment row, four residual loads, waits, then nine
// stores - and because `out` may alias `resid` (the stream is updated in place) hipcc must keep every row's loads BEHIND the previous
// row's stores: s_waitcnt vmcnt(0) in front of each row waits for those stores to be acknowledged and then for the loads, FM times in
// a row (tools/gemm_bench stamps: 7.9 us of "epilogue issue" per 160 x 128 tile against a 14 us K loop; the ISA shows the 5 x
// [4 loads, vmcnt(0), 9 stores]).  Here the residual rows are loaded G fragment rows at a time into a ping-pong register set, group
// g + 1 BEFORE the stores of group g are issued (a thread only ever re-reads what it alone writes, so the order is free), and the
// wait in front of a group is a counted one that leaves the younger stores in flight.  G = FM where the register budget allows
// (two-stage 160 x 128: every load of the tile in flight at once); the 16-bit copy goes out in 16-byte stores through the
// fragment-pair lane swap of gemm_epilogue_impl.  Same arithmetic in the same order: bit-identical to the guarded form.
// Residual rows of a wave's patch loaded BEFORE the K loop (GemmParams::rs_prefetch_from: the workgroups a CU receives second).  A
// single-round grid of the two-per-CU kernel runs every K loop first and every epilogue after it, chip-wide: 97 MB of residual reads,
// stream stores and 16-bit copies in the last ~8 us of a 30 us launch (8-9 TB/s: the fabric, not latency - loading all rows of the
// tile at once at the start of the epilogue changed nothing), while the loops before it barely touch HBM.  With the second workgroup of
// every CU fetching its residual rows up front (its first K-tile wait also waits for them: vmcnt is in order - it starts ~4 us late,
// while the first workgroup has the CU's operand path to itself) the two phases of the two workgroups interleave.
#ifndef IVIT_RS_COPY_LAST
#define IVIT_RS_COPY_LAST 0   // study builds: 1 = the 16-bit copy always after the statistics (where the centred copy has to be)
#endif
#ifndef IVIT_RS_PREFETCH_ROWS
#define IVIT_RS_PREFETCH_ROWS 8   // fragment rows held across the K loop (16 registers each; capped at the tile's FM): all five of the 160 x 128 tile = 212 registers, no scratch
#endif
template <class T>
struct RsPrefetch {
    static constexpr int ROWS = T::FM < IVIT_RS_PREFETCH_ROWS ? T::FM : IVIT_RS_PREFETCH_ROWS;
    typedef float4 Rows[ROWS][T::FN];   // (a plain array passed by reference: a struct handed on by pointer stayed in scratch memory)
};
template <class T>
__device__ __forceinline__ void rs_prefetch_load(const GemmParams& p, int m_base, int n_base, int fr, int fq, typename RsPrefetch<T>::Rows& x) {
    const float* src = p.resid + (size_t)(m_base + fr) * p.ldr + n_base + fq * 4;
#pragma unroll
    for (int i = 0; i < RsPrefetch<T>::ROWS; ++i)
#pragma unroll
        for (int j = 0; j < T::FN; ++j) x[i][j] = *reinterpret_cast<const float4*>(src + (size_t)i * 16 * p.ldr + j * 16);
}

template <class T, class OP, int G, bool HOLD_CT = false>
__device__ __forceinline__ void gemm_epilogue_resid_stats_interior(const GemmParams& p, f32x4 (&acc)[T::FM][T::FN], int m_base, int n_base, int fr, int fq,
                                                                   const typename RsPrefetch<T>::Rows& pfx, bool pf_have) {
    static_assert(T::FN * 16 == 64 && T::FN % 2 == 0, "one statistics slot per wave column; fragment pairs for the 16-byte stores");
    constexpr int NG = (T::FM + G - 1) / G;
    const int slot = n_base >> 6;
    float4 bias4[T::FN];
#pragma unroll
    for (int j = 0; j < T::FN; ++j) bias4[j] = *reinterpret_cast<const float4*>(p.bias + n_base + j * 16 + fq * 4);
    const bool centred = p.ln_centre != nullptr;   // (wave-uniform) the 16-bit copy is rn16(x - centre[n]); statistics and stream are those of x itself
    int orow[T::FM];
    const float* src[T::FM];
    if (p.grp_in > 0) {   // EPI_BIAS_ROWADD_STATS: GEMM row -> token row of its image, the addend is the table row (position embedding)
#pragma unroll
        for (int i = 0; i < T::FM; ++i) {
            const int m = m_base + i * 16 + fr;
            const int grp = m / p.grp_in, within = m - grp * p.grp_in;
            orow[i] = grp * p.grp_out + p.grp_off + within;
            src[i] = p.rowadd + (size_t)(p.grp_off + within) * p.ldra + n_base + fq * 4;
        }
    } else {
#pragma unroll
        for (int i = 0; i < T::FM; ++i) {
            orow[i] = m_base + i * 16 + fr;
            src[i] = p.resid + (size_t)orow[i] * p.ldr + n_base + fq * 4;
        }
    }
    float4 x[2][G][T::FN];
    const bool pre = G >= T::FM && pf_have;   // the first residual rows of the patch are already in registers (wave-uniform)
    {   // group 0: the first ROWS fragment rows may be in registers already (copied unconditionally, then overwritten by the loads of a workgroup
        // that did not prefetch: a `pre ? registers : load` per element became ONE load through a pointer phi and kept the array in scratch)
        constexpr int PR = RsPrefetch<T>::ROWS;
#pragma unroll
        for (int ii = 0; ii < G; ++ii)
#pragma unroll
            for (int j = 0; j < T::FN; ++j)
                if (ii < T::FM && ii < PR && G >= T::FM) x[0][ii][j] = pfx[ii < PR ? ii : 0][j];
#pragma unroll
        for (int ii = 0; ii < G; ++ii)
#pragma unroll
            for (int j = 0; j < T::FN; ++j)
                if (ii < T::FM && !(pre && ii < PR)) x[0][ii][j] = *reinterpret_cast<const float4*>(src[ii] + j * 16);
    }
    // The centre vector of this wave's columns is (re)loaded per residual-load group, AHEAD of the next group's residual loads: vector-memory results return in
    // order, so a load issued behind them makes the copy of THIS group wait for the NEXT group's residual rows (the one-row-ahead pipeline gone: proj + 2.3 us per
    // launch).  Re-read per group from L1 rather than held across the tile: the 168-register tiles have no room for 16 more.  The three-per-CU 128 x 128 tile - FM = 4
    // with two rows of residual loads in flight - has no room for them even that long (36 bytes of scratch): it reads them behind the statistics
    // HOLD_CT (the two-per-CU tile: 256 registers a lane): loaded ONCE for the tile - every re-read is another 4 KiB per wave through the CU's vector-memory return
    // path, which is what an epilogue burst is bound by (five re-reads per 160 x 128 tile: proj + 2.3 us per launch)
    constexpr bool CT_EARLY = !(T::FM == 4 && G == 2);
    float4 ct[T::FN];
    if (centred && HOLD_CT) {
#pragma unroll
        for (int j = 0; j < T::FN; ++j) ct[j] = *reinterpret_cast<const float4*>(p.ln_centre + n_base + j * 16 + fq * 4);
    }
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        if (centred && CT_EARLY && !HOLD_CT) {
#pragma unroll
            for (int j = 0; j < T::FN; ++j) ct[j] = *reinterpret_cast<const float4*>(p.ln_centre + n_base + j * 16 + fq * 4);
        }
        if (g + 1 < NG) {
#pragma unroll
            for (int ii = 0; ii < G; ++ii)
#pragma unroll
                for (int j = 0; j < T::FN; ++j)
                    if ((g + 1) * G + ii < T::FM) x[(g + 1) & 1][ii][j] = *reinterpret_cast<const float4*>(src[(g + 1) * G + ii] + j * 16);
        }
#pragma unroll
        for (int ii = 0; ii < G; ++ii) {
            const int i = g * G + ii;
            if (i >= T::FM) continue;
            const int mr = orow[i];
            float* o = reinterpret_cast<float*>(p.out) + (size_t)mr * p.ldo + n_base + fq * 4;
            bf16_t* ob = p.xb + (size_t)mr * p.ldxb + n_base;
            float sum = 0.f;
#pragma unroll
            for (int j = 0; j < T::FN; ++j) {
                const float4 xr = x[g & 1][ii][j];
                const float v[4] = {xr.x + (acc[i][j][0] + bias4[j].x), xr.y + (acc[i][j][1] + bias4[j].y),
                                    xr.z + (acc[i][j][2] + bias4[j].z), xr.w + (acc[i][j][3] + bias4[j].w)};
#pragma unroll
                for (int r = 0; r < 4; ++r) { acc[i][j][r] = v[r]; sum += v[r]; }
                *reinterpret_cast<float4*>(o + j * 16) = make_float4(v[0], v[1], v[2], v[3]);
            }
            auto store_copy = [&]() {   // 16-bit copy: 16-byte stores (lane rows fq / fq ^ 1 trade quads of a fragment pair)
#pragma unroll
                for (int j = 0; j < T::FN; j += 2) {
                    const auto lo = __builtin_amdgcn_permlane16_swap(OP::pack2(acc[i][j][0], acc[i][j][1]), OP::pack2(acc[i][j + 1][0], acc[i][j + 1][1]), false, false);
                    const auto hi = __builtin_amdgcn_permlane16_swap(OP::pack2(acc[i][j][2], acc[i][j][3]), OP::pack2(acc[i][j + 1][2], acc[i][j + 1][3]), false, false);
                    u32x4 pk = {lo[0], hi[0], lo[1], hi[1]};
                    *reinterpret_cast<u32x4*>(ob + (j + (fq & 1)) * 16 + (fq & ~1) * 4) = pk;
                }
            };
            auto store_copy_centred = [&]() {   // the same on x - centre, the row's values left as they are (the statistics below are those of x)
#pragma unroll
                for (int j = 0; j < T::FN; j += 2) {
                    const auto lo = __builtin_amdgcn_permlane16_swap(OP::pack2(acc[i][j][0] - ct[j].x, acc[i][j][1] - ct[j].y), OP::pack2(acc[i][j + 1][0] - ct[j + 1].x, acc[i][j + 1][1] - ct[j + 1].y), false, false);
                    const auto hi = __builtin_amdgcn_permlane16_swap(OP::pack2(acc[i][j][2] - ct[j].z, acc[i][j][3] - ct[j].w), OP::pack2(acc[i][j + 1][2] - ct[j + 1].z, acc[i][j + 1][3] - ct[j + 1].w), false, false);
                    u32x4 pk = {lo[0], hi[0], lo[1], hi[1]};
                    *reinterpret_cast<u32x4*>(ob + (j + (fq & 1)) * 16 + (fq & ~1) * 4) = pk;
                }
            };
            // the copy goes out beside the f32 stores (its stores then overlap the statistics' shuffle chains: behind them, the centred form cost proj 2.4 us per
            // launch) - except on the tile that has no room for the centre vector that early, which subtracts in place after the statistics
            if (!centred && !IVIT_RS_COPY_LAST) store_copy();
            else if (centred && CT_EARLY && !IVIT_RS_COPY_LAST) store_copy_centred();
            sum += __shfl_xor(sum, 16, 64);
            sum += __shfl_xor(sum, 32, 64);
            const float lmean = sum / 64.0f;
            float m2 = 0.f;
#pragma unroll
            for (int j = 0; j < T::FN; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) { const float d = acc[i][j][r] - lmean; m2 = fmaf(d, d, m2); }
            m2 += __shfl_xor(m2, 16, 64);
            m2 += __shfl_xor(m2, 32, 64);
            if (fq == 0) p.ln_part[(size_t)mr * GEMM_LN_SLOTS + slot] = make_float2(sum, m2);
            // 16-bit copy, last: the row's values are dead after the statistics, so a centred copy (rn16(x - centre[n])) subtracts IN PLACE
            if (centred && (!CT_EARLY || IVIT_RS_COPY_LAST)) {
#pragma unroll
                for (int j = 0; j < T::FN; ++j) {
                    if (!CT_EARLY) ct[j] = *reinterpret_cast<const float4*>(p.ln_centre + n_base + j * 16 + fq * 4);
                    acc[i][j][0] -= ct[j].x; acc[i][j][1] -= ct[j].y; acc[i][j][2] -= ct[j].z; acc[i][j][3] -= ct[j].w;
                }
            }
            if ((centred && !CT_EARLY) || IVIT_RS_COPY_LAST) store_copy();
        }
    }
}

// Accumulators of a tile at the start of its K loop: zero - or, for an EPI_LNFOLD_* GEMM on centred operand rows (GemmParams::ln_d), d[n] in every row of
// column n, so that the epilogue's acc already holds rn16(x - centre) W'^T + centre W'^T.  (Adding d in the epilogue instead cost 16 more live registers
// there: 130-320 bytes of scratch in the 168-register and the 256 x 256 kernels.)  The loads are issued ahead of the first operand DMA: its wait covers them.
template <class T, int EK>
__device__ __forceinline__ void gemm_acc_init(const GemmParams& p, f32x4 (&acc)[T::FM][T::FN], int n_base, int fq) {
#pragma unroll
    for (int j = 0; j < T::FN; ++j) {
        f32x4 d = {0.f, 0.f, 0.f, 0.f};
        if (EK == 2 && p.ln_d) {   // (wave-uniform)
            const int n = n_base + j * 16 + fq * 4;
            if (n + 3 < p.N) { const float4 t = *reinterpret_cast<const float4*>(p.ln_d + n); d = f32x4{t.x, t.y, t.z, t.w}; }
            else for (int r = 0; r < 4; ++r) if (n + r < p.N) d[r] = p.ln_d[n + r];
        }
#pragma unroll
        for (int i = 0; i < T::FM; ++i) acc[i][j] = d;
    }
}

// EPI_LNFOLD_*: v = rstd[m] * (acc - mean[m] * s[n]) + c[n]  (c arrives as `bias`), optional GELU, bf16 out.
// The row statistics are read from LDS, where ln_tile_stats put them while the first operand tiles were in flight
// (folding the producers' pairs here, per wave, cost 6-22 us per GEMM; a separate finalize kernel 5 us per LayerNorm).
template <class T, bool INTERIOR, class OP = OpBf16>
__device__ __forceinline__ void gemm_epilogue_lnfold(const GemmParams& p, f32x4 (&acc)[T::FM][T::FN], int m_base, int n_base, int fr, int fq,
                                                     const float2* tile_stats) {
    // finished row statistics (mean, rstd) of the tile's rows, left in LDS by ln_tile_stats at the start of the kernel
    float mean_i[T::FM], rstd_i[T::FM];
#pragma unroll
    for (int i = 0; i < T::FM; ++i) {
        const float2 st = tile_stats[i * 16 + fr];
        mean_i[i] = st.x;
        rstd_i[i] = st.y;
    }
    float4 c4[T::FN], s4[T::FN];
#pragma unroll
    for (int j = 0; j < T::FN; ++j) {
        const int n = n_base + j * 16 + fq * 4;
        if (INTERIOR || n + 3 < p.N) { c4[j] = *reinterpret_cast<const float4*>(p.bias + n); s4[j] = *reinterpret_cast<const float4*>(p.ln_s + n); }
        else {
            float c[4] = {0.f, 0.f, 0.f, 0.f}, s[4] = {0.f, 0.f, 0.f, 0.f};
            for (int r = 0; r < 4; ++r) if (n + r < p.N) { c[r] = p.bias[n + r]; s[r] = p.ln_s[n + r]; }
            c4[j] = make_float4(c[0], c[1], c[2], c[3]); s4[j] = make_float4(s[0], s[1], s[2], s[3]);
        }
    }
    const bool gelu = p.epi == EPI_LNFOLD_GELU_BF16;
#pragma unroll
    for (int i = 0; i < T::FM; ++i) {
        const int m = m_base + i * 16 + fr;
        if (!INTERIOR && m >= p.M) continue;
        const float mu = mean_i[i], rs = rstd_i[i];
        bf16_t* orow_p = reinterpret_cast<bf16_t*>(p.out) + (size_t)m * p.ldo;
        if (INTERIOR && (T::FN % 2 == 0)) {   // 16-byte stores through the fragment-pair lane swap (see gemm_epilogue_impl)
#pragma unroll
            for (int j = 0; j < T::FN; j += 2) {
                float a[4] = {fmaf(rs, fmaf(-mu, s4[j].x, acc[i][j][0]), c4[j].x), fmaf(rs, fmaf(-mu, s4[j].y, acc[i][j][1]), c4[j].y),
                              fmaf(rs, fmaf(-mu, s4[j].z, acc[i][j][2]), c4[j].z), fmaf(rs, fmaf(-mu, s4[j].w, acc[i][j][3]), c4[j].w)};
                float b[4] = {fmaf(rs, fmaf(-mu, s4[j + 1].x, acc[i][j + 1][0]), c4[j + 1].x), fmaf(rs, fmaf(-mu, s4[j + 1].y, acc[i][j + 1][1]), c4[j + 1].y),
                              fmaf(rs, fmaf(-mu, s4[j + 1].z, acc[i][j + 1][2]), c4[j + 1].z), fmaf(rs, fmaf(-mu, s4[j + 1].w, acc[i][j + 1][3]), c4[j + 1].w)};
                if (gelu) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) { a[r] = gelu_erf(a[r]); b[r] = gelu_erf(b[r]); }
                }
                const auto lo = __builtin_amdgcn_permlane16_swap(OP::pack2(a[0], a[1]), OP::pack2(b[0], b[1]), false, false);
                const auto hi = __builtin_amdgcn_permlane16_swap(OP::pack2(a[2], a[3]), OP::pack2(b[2], b[3]), false, false);
                const int n = n_base + (j + (fq & 1)) * 16 + (fq & ~1) * 4;
                u32x4 pk = {lo[0], hi[0], lo[1], hi[1]};
                *reinterpret_cast<u32x4*>(orow_p + n) = pk;
            }
            continue;
        }
#pragma unroll
        for (int j = 0; j < T::FN; ++j) {
            const int n = n_base + j * 16 + fq * 4;
            if (n >= p.N) continue;
            float v[4] = {fmaf(rs, fmaf(-mu, s4[j].x, acc[i][j][0]), c4[j].x), fmaf(rs, fmaf(-mu, s4[j].y, acc[i][j][1]), c4[j].y),
                          fmaf(rs, fmaf(-mu, s4[j].z, acc[i][j][2]), c4[j].z), fmaf(rs, fmaf(-mu, s4[j].w, acc[i][j][3]), c4[j].w)};
            if (gelu) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = gelu_erf(v[r]);
            }
            bf16_t* o = orow_p + n;
            if (n + 3 < p.N) { u32x2 pk = {OP::pack2(v[0], v[1]), OP::pack2(v[2], v[3])}; *reinterpret_cast<u32x2*>(o) = pk; }
            else for (int r = 0; r < 4; ++r) if (n + r < p.N) o[r] = OP::from_f32(v[r]);
        }
    }
}

// (mean, rstd) of the BM rows of this workgroup's tile -> LDS, one thread per row, at the very start of the kernel
// (the loads overlap the first operand DMA).  Source: finished statistics (ln_stats, a kernel-level option the engine no longer uses) or the
// per-64-column (sum, M2) pairs a residual GEMM left (ln_part), folded in slot order with Chan's formula - exact
// two-pass statistics, the same bits whatever tile shape wrote the pairs and whichever column tile folds them.
// Chan's update of a row's running (mean, M2) with the (sum, M2) pair of 64-column slot s2 (callers unroll fully, so s2 and the
// ratios below are compile-time constants).  Shared by every consumer of the per-slot pairs: they all fold identically.
__device__ __forceinline__ void ln_chan_update(float& mean, float& m2, float sm, float mm, int s2, int ln_dim) {
    constexpr float inv64 = 1.0f / 64.0f;
    const int nk = min(64, ln_dim - s2 * 64);
    if (nk == 64) {
        const float d = fmaf(sm, inv64, -mean);
        mean = fmaf(d, 1.0f / (float)(s2 + 1), mean);
        m2 = fmaf(d * d, 64.0f * (float)s2 / (float)(s2 + 1), m2 + mm);
    } else {                                   // the last, ragged slot of a width that is not a multiple of 64
        const float nb = (float)nk, na = 64.0f * (float)s2, nn = na + nb, d = sm / nb - mean;
        mean = fmaf(d, nb / nn, mean);
        m2 = fmaf(d * d, na * nb / nn, m2 + mm);
    }
}

// (mean, rstd) of ONE row from the per-64-column (sum, M2) pairs a residual GEMM left for it, folded in slot order.  Every consumer of the
// pairs goes through this function (ln_tile_stats of the _lf GEMMs, the fused MLP kernel): the same bits whoever folds them.
__device__ __forceinline__ float2 ln_row_stats_from_pairs(const float2* ln_part, int m, int ln_dim, float ln_eps, const float4* first = nullptr) {
    const int nslots = (ln_dim + 63) >> 6;
    const float4* pr = reinterpret_cast<const float4*>(ln_part + (size_t)m * GEMM_LN_SLOTS);
    float mean = 0.f, m2 = 0.f;
    // Chan's update for slot k (0-based) when every earlier slot is a full 64 columns:
    //   d = mean_k - mean;  mean += d * n_k / (64 k + n_k);  M2 += M2_k + d^2 * 64 k n_k / (64 k + n_k)
    // fully unrolled, so for the full slots (n_k = 64) the three ratios are compile-time constants - the
    // loop with run-time divisions cost the MLP-up GEMM 7 us at the start of its tiles
#pragma unroll
    for (int s0 = 0; s0 < GEMM_LN_SLOTS; s0 += 12) {
        if (s0 >= nslots) break;                      // uniform
        float4 raw[6];
#pragma unroll
        for (int l = 0; l < 6; ++l) raw[l] = (s0 == 0 && first) ? first[l] : pr[min((s0 >> 1) + l, GEMM_LN_SLOTS / 2 - 1)];   // the first 12 slots may have been loaded ahead (ln_tile_stats_prefetch)
#pragma unroll
        for (int q = 0; q < 12; ++q) {
            const int s2 = s0 + q;                     // compile-time after unrolling
            if (s2 >= GEMM_LN_SLOTS || s2 >= nslots) continue;
            const float sm = (q & 1) ? raw[q >> 1].z : raw[q >> 1].x, mm = (q & 1) ? raw[q >> 1].w : raw[q >> 1].y;
            ln_chan_update(mean, m2, sm, mm, s2, ln_dim);
        }
    }
    return make_float2(mean, 1.0f / sqrtf(m2 / (float)ln_dim + ln_eps));
}

template <class T>
__device__ __forceinline__ void ln_tile_stats(const GemmParams& p, int m0, float2* tile_stats, const float4* first = nullptr) {
    for (int r = threadIdx.x; r < T::BM; r += T::THREADS) {
        const int m = min(m0 + r, p.M - 1);     // rows past M: a valid row's statistics (their outputs are never stored)
        tile_stats[r] = p.ln_stats ? p.ln_stats[m] : ln_row_stats_from_pairs(p.ln_part, m, p.ln_dim, p.ln_eps, first);
    }
}

// The first 12 slots of this thread's row, loaded at the very start of the kernel (one row per thread: BM <= THREADS).  The
// loads are in flight together with the first operand DMA and are retired by the K loop's first wait anyway; the fold itself
// (ln_tile_stats with `first`) then runs after that wait.  Folding before the loop cost every tile 1.6 us of exposed load
// latency (per-block stamps, tools/gemm_bench): 6-7 us per MLP-up launch.
template <class T>
__device__ __forceinline__ void ln_tile_stats_prefetch(const GemmParams& p, int m0, float4 (&first)[6]) {
    static_assert(T::BM <= T::THREADS, "one row per thread");
    const int m = min(m0 + min((int)threadIdx.x, T::BM - 1), p.M - 1);
    const float4* pr = reinterpret_cast<const float4*>(p.ln_part + (size_t)m * GEMM_LN_SLOTS);
#pragma unroll
    for (int l = 0; l < 6; ++l) first[l] = pr[l];
}

// EK = 0: the classic 16-bit / fp8 output epilogues (kind chosen at run time); 3: the classic f32 output epilogues; 1: EPI_BIAS_RESID_STATS; 2: EPI_LNFOLD_*.
// (One kernel per classic kind was tried too: no gain at the ViT-B shapes, 5-15 % slower at the ViT-H shapes.)
// RSG: fragment rows per residual-load group of the interior EPI_BIAS_RESID_STATS epilogue (register budget of the calling kernel)
template <class T, int EK, class OP = OpBf16, int RSG = T::FM, bool HOLD_CT = false>
__device__ __forceinline__ void gemm_epilogue_family(const GemmParams& p, f32x4 (&acc)[T::FM][T::FN], int m_base, int n_base, int fr, int fq,
                                                     const float2* tile_stats, const typename RsPrefetch<T>::Rows& pfx, bool pf_have) {
    const bool interior = (m_base + T::FM * 16 <= p.M) && (n_base + T::FN * 16 <= p.N);
    if constexpr (EK == 0 || EK == 3 || EK == -1) {
        gemm_epilogue<T, OP, (RSG > 2 ? 2 : RSG), EK>(p, acc, m_base, n_base, fr, fq);
    } else if constexpr (EK == 1) {
#ifdef IVIT_GEMM_ABLATIONS   // A/B in one binary (tools/gemm_bench argv[5] = 7, IVIT_OLD_EPI=1 with tools/libivit_abl.so): round 3's row-by-row form
        if (interior && p.debug == 7) { gemm_epilogue_resid_stats<T, true, OP>(p, acc, m_base, n_base, fr, fq); return; }
#endif
        if (interior) gemm_epilogue_resid_stats_interior<T, OP, RSG, HOLD_CT>(p, acc, m_base, n_base, fr, fq, pfx, pf_have);
        else gemm_epilogue_resid_stats<T, false, OP>(p, acc, m_base, n_base, fr, fq);
    } else {
        if (interior) gemm_epilogue_lnfold<T, true, OP>(p, acc, m_base, n_base, fr, fq, tile_stats);
        else gemm_epilogue_lnfold<T, false, OP>(p, acc, m_base, n_base, fr, fq, tile_stats);
    }
}

template <class T, int EK, class OP = OpBf16, int RSG = T::FM>
__device__ __forceinline__ void gemm_epilogue_family(const GemmParams& p, f32x4 (&acc)[T::FM][T::FN], int m_base, int n_base, int fr, int fq,
                                                     const float2* tile_stats = nullptr) {
    typename RsPrefetch<T>::Rows none;   // never read (pf_have = false)
    gemm_epilogue_family<T, EK, OP, RSG>(p, acc, m_base, n_base, fr, fq, tile_stats, none, false);
}

// XCD-aware, bijective block -> tile map: blocks that share an XCD (id % 8) get a contiguous run of
// tiles, n fastest, so neighbours re-use the same A rows out of that XCD's L2.
__device__ __forceinline__ int xcd_tile(int orig, int nwg) {
    const int xcd = orig & 7, qd = nwg >> 3, rm = nwg & 7;
    return (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (orig >> 3);
}

// Linear tile index -> (tm, tn): column panels GROUP_N tiles wide, row-major inside a panel.  The
// ~64 tiles an XCD runs concurrently (a contiguous run of the index, xcd_tile) then form an
// ~8 x 8 block: 8 A row-tiles + 8 W col-tiles (~3.5 MB at K = 768) fit its 4 MiB L2, where the plain
// n-fastest order swept all of W per row-tile and overflowed it (21 % L2 misses, 4.7x over-fetch).
constexpr int GEMM_GROUP_N = 8;
__device__ __forceinline__ void tile_coords(int tile, int tiles_m, int tiles_n, int& tm, int& tn, int group = GEMM_GROUP_N) {
    const int panel = tile / (group * tiles_m);
    const int within = tile - panel * group * tiles_m;
    const int width = min(group, tiles_n - panel * group);
    tm = within / width;
    tn = panel * group + (within - tm * width);
}

// Row bands per XCD: the linear tile order is band-major (8 bands of ~tiles_m / 8 row tiles), column panels inside a
// band, row-major inside a panel - so the contiguous run of tiles an XCD executes (xcd_tile) stays inside one band of
// A rows for the whole launch (A enters one L2) and sweeps the W panels.
__device__ __forceinline__ void tile_coords_banded(int tile, int tiles_m, int tiles_n, int& tm, int& tn, int group = GEMM_GROUP_N) {
    const int rpb = (tiles_m + 7) >> 3;
    const int band = tile / (rpb * tiles_n);
    const int within = tile - band * rpb * tiles_n;
    const int rows = min(rpb, tiles_m - band * rpb);
    tile_coords(within, rows, tiles_n, tm, tn, group);
    tm += band * rpb;
}

#ifdef IVIT_GEMM_ABLATIONS
#define IVIT_BODY_STAMP(slot)                                                                                 \
    do {                                                                                                      \
        if (p.stamps && threadIdx.x == 0) {                                                                   \
            unsigned long long t_;                                                                            \
            asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                     \
            p.stamps[(size_t)blockIdx.x * 16 + (slot)] = t_;                                                  \
        }                                                                                                     \
    } while (0)
#else
#define IVIT_BODY_STAMP(slot) do { } while (0)
#endif

template <class T, bool FP8 = false, int EK = 0, class OP = OpBf16>
__device__ __forceinline__ void gemm_body(const GemmParams& p, char* smem) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wave / T::WAVES_N, wc = wave % T::WAVES_N;
    IVIT_BODY_STAMP(0);

    int tm, tn;
#ifdef IVIT_GEMM_ABLATIONS   // tile orders that were measured and did not pay: row bands per XCD; pairs sharing a CU's L1
    if (p.order == 1) tile_coords_banded(xcd_tile(blockIdx.x, gridDim.x), ceil_div(p.M, T::BM), ceil_div(p.N, T::BN), tm, tn);
    else if (p.order == 2) {
        // Paired order for single-round grids of the two-workgroups-per-CU kernels (grid = 512): blocks b and b + 256 - the two
        // workgroups a CU receives when every CU gets one block before any gets its second - take the SAME row tile and
        // column tiles hn apart, so that the A lines one of them pulls through the CU's vector L1 could serve the other.
        const int tiles_m = ceil_div(p.M, T::BM), hn = ceil_div(p.N, T::BN) >> 1;
        const int first = blockIdx.x & 255, x = first & 7, j = first >> 3;
        tm = (j / hn) * 8 + x;
        tn = j % hn + ((blockIdx.x >> 8) ? hn : 0);
        if (tm >= tiles_m) return;
    } else
#endif
    tile_coords(xcd_tile(blockIdx.x, gridDim.x), ceil_div(p.M, T::BM), ceil_div(p.N, T::BN), tm, tn);
    const int m0 = tm * T::BM;
    const int n0 = tn * T::BN;

    f32x4 acc[T::FM][T::FN];
    gemm_acc_init<T, EK>(p, acc, n0 + wc * T::FN * 16, (threadIdx.x & 63) >> 4);

    // a K-tile is 128 BYTES of every row: 64 bf16 or 128 fp8
    constexpr int ESZ = FP8 ? 1 : 2;
    const size_t lda_b = (size_t)p.lda * ESZ, ldw_b = (size_t)p.ldw * ESZ;
    const int nt = p.K * ESZ / 128;
    stage_tile<T::A_PIECES, T::WAVES>(p.A, lda_b, m0, 0, smem, wave, lane);
    stage_tile<T::W_PIECES, T::WAVES>(p.W, ldw_b, n0, 0, smem + T::A_BYTES, wave, lane);
    float2* tile_stats = reinterpret_cast<float2*>(smem + T::LDS_BYTES);   // EK == 2 kernels are launched with BM * 8 more bytes
    float4 ln_first[6];
    const bool ln_deferred = EK == 2 && !p.ln_stats && nt >= 2;             // fold inside the K loop (behind its first wait); else here
    if (EK == 2) {
        if (ln_deferred) ln_tile_stats_prefetch<T>(p, m0, ln_first);
        else ln_tile_stats<T>(p, m0, tile_stats);                          // visible to every wave after the K loop's barriers
    }
    const int fr = lane & 15;   // fragment row (m for the A^T operand, n for the W operand)
    const int fq = lane >> 4;   // 16-B k-chunk inside a 32-deep MFMA step
    typename RsPrefetch<T>::Rows rs_x;
    bool rs_have = false;
    if constexpr (EK == 1) {
#pragma unroll
        for (int i = 0; i < RsPrefetch<T>::ROWS; ++i)
#pragma unroll
            for (int j = 0; j < T::FN; ++j) rs_x[i][j] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if constexpr (EK == 1 && (IVIT_RSG_2STAGE) >= T::FM) {   // see RsPrefetch (only with the whole-tile load group): the second workgroup of a CU brings its residual rows in before its K loop
        rs_have = p.rs_prefetch_from > 0 && (int)blockIdx.x >= p.rs_prefetch_from && p.grp_in == 0 && m0 + T::BM <= p.M && n0 + T::BN <= p.N;
        if (rs_have) rs_prefetch_load<T>(p, m0 + wr * T::FM * 16, n0 + wc * T::FN * 16, fr, fq, rs_x);
    }
    IVIT_BODY_STAMP(1);

    bf16x8 af2[T::FM][2];   // the activation fragments of the K-tile (16-bit operands): kept across a hi / lo pair of weight K-tiles (a_shift)
    // one K-tile; FOLD (first iteration of the LayerNorm-fold kernels only, a separate copy of the body so that the loop proper
    // carries neither the branch nor the prefetched registers): fold the prefetched statistics pairs right after the wait
    auto ktile = [&](int t, auto fold_tag) {
        constexpr bool FOLD = decltype(fold_tag)::value;
        // tile t has landed (every wave drains its own DMA, then the barrier publishes it); every
        // wave is also past its reads of the buffer that tile t+1 is about to overwrite
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        char* cur = smem + (t & 1) * T::STAGE_BYTES;
        // the prefetched statistics pairs have landed (the wait above); fold them BEFORE the next DMA is issued (a use of an
        // ordinary load behind in-flight LDS-DMA makes hipcc wait for all of it); published by the next iteration's barrier
        if (FOLD) ln_tile_stats<T>(p, m0, tile_stats, ln_first);
        // a_shift (weight-only split, hi / lo K-tiles alternate): K-tiles 2a and 2a + 1 multiply the same activation K-tile a - it is
        // staged for the even one only, and its fragments stay in registers (af2) for the odd one
        const bool a_new = !(IVIT_ASHIFT_REUSE && OP::F16 && p.a_shift && (t & 1));
        if (t + 1 < nt) {
            char* nxt = smem + ((t + 1) & 1) * T::STAGE_BYTES;
            if (!(IVIT_ASHIFT_REUSE && OP::F16 && p.a_shift && ((t + 1) & 1))) stage_tile<T::A_PIECES, T::WAVES>(p.A, lda_b, m0, a_ktile(p, t + 1) * 128, nxt, wave, lane);
            stage_tile<T::W_PIECES, T::WAVES>(p.W, ldw_b, n0, (t + 1) * 128, nxt + T::A_BYTES, wave, lane);
        }
        const char* a_tile = cur;
        const char* w_tile = cur + T::A_BYTES;
        if (FP8) {
            // one 128-deep MFMA per fragment pair and K-tile: a lane supplies the two 16-B k-chunks fq and
            // 4 + fq of its row for both operands (the instruction's k order inside the 128-block is the
            // same for both sources, so any consistent cut is a dot product over all 128 k)
            bf16x8 af[T::FM][2], wf[T::FN][2];
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
                for (int j = 0; j < T::FN; ++j) wf[j][kk] = read_frag(w_tile, (wc * T::FN + j) * 16 + fr, kk * 4 + fq);
#pragma unroll
                for (int i = 0; i < T::FM; ++i) af[i][kk] = read_frag(a_tile, (wr * T::FM + i) * 16 + fr, kk * 4 + fq);
            }
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < T::FM; ++i)
#pragma unroll
                for (int j = 0; j < T::FN; ++j) acc[i][j] = mfma_e4m3_16x16x128(wf[j], af[i], acc[i][j]);
            __builtin_amdgcn_s_setprio(0);
        } else {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                bf16x8 wf[T::FN];
#pragma unroll
                for (int j = 0; j < T::FN; ++j) wf[j] = read_frag(w_tile, (wc * T::FN + j) * 16 + fr, kk * 4 + fq);
                if (a_new) {
#pragma unroll
                    for (int i = 0; i < T::FM; ++i) af2[i][kk] = read_frag(a_tile, (wr * T::FM + i) * 16 + fr, kk * 4 + fq);
                }
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int i = 0; i < T::FM; ++i)
#pragma unroll
                    for (int j = 0; j < T::FN; ++j)
                        acc[i][j] = OP::mfma(wf[j], af2[i][kk], acc[i][j]);
                __builtin_amdgcn_s_setprio(0);
            }
        }
    };
    int t_first = 0;
    if (EK == 2) {
        if (ln_deferred) { ktile(0, std::true_type{}); t_first = 1; }
    }
    for (int t = t_first; t < nt; ++t) ktile(t, std::false_type{});
    IVIT_BODY_STAMP(2);
    gemm_epilogue_family<T, EK, OP, IVIT_RSG_2STAGE, true>(p, acc, m0 + wr * T::FM * 16, n0 + wc * T::FN * 16, fr, fq, tile_stats + wr * T::FM * 16, rs_x, rs_have);
    IVIT_BODY_STAMP(3);
#ifdef IVIT_GEMM_ABLATIONS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    IVIT_BODY_STAMP(4);
    if (p.stamps && threadIdx.x == 0)   // which CU ran this block (per-CU timelines in tools/gemm_bench)
        p.stamps[(size_t)blockIdx.x * 16 + 5] = ((unsigned long long)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 15) << 32) | __builtin_amdgcn_s_getreg((31 << 11) | 4);
#endif
}

// ---- single-stage form of gemm_body: ONE operand stage per workgroup (half the LDS), so that THREE workgroups fit a CU.
// The two-per-CU kernels saturate the CU's L2 -> LDS path (~78 GB/s per CU: 72 one-KiB DMA pieces per 1800 shader cycles, section
// stamps of tools/gemm_bench) only while BOTH workgroups are inside their K loops; a workgroup in its prologue or epilogue leaves the
// path to one, which alone keeps it 60 % busy.  With three resident workgroups two are in their loops most of the time (round 3,
// tools/gemm_bench at M = 12608: qkv 60.1 -> 53.7 us, mlp1 + fold + GELU 80.8 -> 72.5, proj 30.8 -> 30.1, mlp2 69.1 -> 62.5).  Per K-tile:
// wait + barrier (tile landed), ALL fragments of the K-tile to registers, barrier (everyone has read), restage the same buffer, MFMAs.
// The DMA round trip is exposed per workgroup and covered by the other two.  Same MFMA order and epilogues as gemm_body: bit-identical
// (tools/gemm_bench compares bitwise).  Four per CU (fragments read half by half, 128 registers) spills and loses 10-25 %.
template <class T, bool FP8 = false, int EK = 0, class OP = OpBf16>
__device__ __forceinline__ void gemm_body_sb(const GemmParams& p, char* smem) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wave / T::WAVES_N, wc = wave % T::WAVES_N;
    int tm, tn;
    tile_coords(xcd_tile(blockIdx.x, gridDim.x), ceil_div(p.M, T::BM), ceil_div(p.N, T::BN), tm, tn);
    const int m0 = tm * T::BM, n0 = tn * T::BN;
    f32x4 acc[T::FM][T::FN];
    gemm_acc_init<T, EK>(p, acc, n0 + wc * T::FN * 16, (threadIdx.x & 63) >> 4);
    // a K-tile is 128 BYTES of every row: 64 bf16 / f16 or 128 fp8
    constexpr int ESZ = FP8 ? 1 : 2;
    const size_t lda_b = (size_t)p.lda * ESZ, ldw_b = (size_t)p.ldw * ESZ;
    const int nt = p.K * ESZ / 128;
    float2* tile_stats = reinterpret_cast<float2*>(smem + T::STAGE_BYTES);   // EK == 2 kernels are launched with BM * 8 more bytes
#if IVIT_LN_STATS_AFTER_DMA
    // the first K-tile's DMA is issued BEFORE the statistics pairs are loaded and folded: their latency lies under the DMA round trip the loop's
    // first wait pays anyway (hipcc waits vmcnt(0) at the first use of the pairs - i.e. also for the DMA, which is wanted here)
    stage_tile<T::A_PIECES, T::WAVES>(p.A, lda_b, m0, 0, smem, wave, lane);
    stage_tile<T::W_PIECES, T::WAVES>(p.W, ldw_b, n0, 0, smem + T::A_BYTES, wave, lane);
    if (EK == 2) ln_tile_stats<T>(p, m0, tile_stats);
#else
    if (EK == 2) ln_tile_stats<T>(p, m0, tile_stats);                          // ordinary loads: before any DMA is in flight
    stage_tile<T::A_PIECES, T::WAVES>(p.A, lda_b, m0, 0, smem, wave, lane);
    stage_tile<T::W_PIECES, T::WAVES>(p.W, ldw_b, n0, 0, smem + T::A_BYTES, wave, lane);
#endif
    const int fr = lane & 15, fq = lane >> 4;
    bf16x8 af[T::FM][2];   // kept across a hi / lo pair of weight K-tiles (a_shift: the odd K-tile multiplies the activation K-tile already in registers)
    for (int t = 0; t < nt; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        bf16x8 wf[T::FN][2];
        const bool a_new = !(IVIT_ASHIFT_REUSE && OP::F16 && p.a_shift && (t & 1));
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
            for (int j = 0; j < T::FN; ++j) wf[j][kk] = read_frag(smem + T::A_BYTES, (wc * T::FN + j) * 16 + fr, kk * 4 + fq);
            if (a_new) {
#pragma unroll
                for (int i = 0; i < T::FM; ++i) af[i][kk] = read_frag(smem, (wr * T::FM + i) * 16 + fr, kk * 4 + fq);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();          // every wave holds its fragments: the stage may be overwritten
        if (t + 1 < nt) {
            if (!(IVIT_ASHIFT_REUSE && OP::F16 && p.a_shift && ((t + 1) & 1))) stage_tile<T::A_PIECES, T::WAVES>(p.A, lda_b, m0, a_ktile(p, t + 1) * 128, smem, wave, lane);
            stage_tile<T::W_PIECES, T::WAVES>(p.W, ldw_b, n0, (t + 1) * 128, smem + T::A_BYTES, wave, lane);
        }
        if (FP8) {   // one 128-deep scaled MFMA per fragment pair (see gemm_body)
#pragma unroll
            for (int i = 0; i < T::FM; ++i)
#pragma unroll
                for (int j = 0; j < T::FN; ++j) acc[i][j] = mfma_e4m3_16x16x128(wf[j], af[i], acc[i][j]);
        } else {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int i = 0; i < T::FM; ++i)
#pragma unroll
                    for (int j = 0; j < T::FN; ++j) acc[i][j] = OP::mfma(wf[j][kk], af[i][kk], acc[i][j]);
        }
    }
    gemm_epilogue_family<T, EK, OP, (T::FM > 4 ? 1 : 2)>(p, acc, m0 + wr * T::FM * 16, n0 + wc * T::FN * 16, fr, fq, tile_stats + wr * T::FM * 16);   // 168 registers: one fragment row ahead at FM = 5
}

// ---- small-M tile with a deep DMA ring (the interactive path: one to a few images, M = 197 ... ~1000 token rows).
// At these sizes a grid is a handful of workgroups and every K-tile of the two-stage gemm_body pays a full L2 -> LDS round
// trip (0.9 us: the MLP-down GEMM of ONE image, K = 3072, took 45 us on 12 workgroups).  Here STAGES - 1 K-tiles of DMA are in
// flight behind a counted s_waitcnt (every wave issues the same number of pieces per K-tile, surplus stagings past the end
// re-load the last K-tile into a dead slot so the count is a constant), one raw barrier per K-tile.  Same MFMA, same K
// order, same epilogues: bit-identical to the other tiles (the batch-independence tests compare across them).
template <int WAVES_M_, int WAVES_N_, int FM_, int FN_, int STAGES_>
struct GemmTileDeep : GemmTile<WAVES_M_, WAVES_N_, FM_, FN_> {
    using Base = GemmTile<WAVES_M_, WAVES_N_, FM_, FN_>;
    static constexpr int STAGES = STAGES_;
    static constexpr bool RAGGED_N = true;
    static constexpr int LDS_BYTES = STAGES * Base::STAGE_BYTES;
    static constexpr int PIECES_PER_WAVE = (Base::A_PIECES + Base::W_PIECES) / Base::WAVES;
    static_assert(Base::A_PIECES % Base::WAVES == 0 && Base::W_PIECES % Base::WAVES == 0, "every wave must issue the same number of DMA pieces");
};

template <class T, int EK = 0, class OP = OpBf16, bool FP8 = false>
__device__ __forceinline__ void gemm_body_deep(const GemmParams& p, char* smem) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wave / T::WAVES_N, wc = wave % T::WAVES_N;
    int tm, tn;
    tile_coords(xcd_tile(blockIdx.x, gridDim.x), ceil_div(p.M, T::BM), ceil_div(p.N, T::BN), tm, tn);
    const int m0 = tm * T::BM, n0 = tn * T::BN;

    f32x4 acc[T::FM][T::FN];
    gemm_acc_init<T, EK>(p, acc, n0 + wc * T::FN * 16, (threadIdx.x & 63) >> 4);

    // a K-tile is 128 BYTES of every row: 64 bf16 / f16 or 128 e4m3 (round 5: the tail rows of the e4m3 256 x 256 grids)
    constexpr int ESZ = FP8 ? 1 : 2;
    const size_t lda_b = (size_t)p.lda * ESZ, ldw_b = (size_t)p.ldw * ESZ;
    const int nt = p.K * ESZ / 128, last = nt - 1;
    float2* tile_stats = reinterpret_cast<float2*>(smem + T::LDS_BYTES);   // EK == 2 kernels are launched with BM * 8 more bytes
    if (EK == 2) {
        ln_tile_stats<T>(p, m0, tile_stats);   // ordinary loads: before any DMA is in flight
        __syncthreads();
    }
    auto stage = [&](int kt, int slot) {
        char* dst = smem + slot * T::STAGE_BYTES;
        stage_tile<T::A_PIECES, T::WAVES>(p.A, lda_b, m0, a_ktile(p, min(kt, last)) * 128, dst, wave, lane);
        stage_tile<T::W_PIECES, T::WAVES>(p.W, ldw_b, n0, min(kt, last) * 128, dst + T::A_BYTES, wave, lane);
    };
#pragma unroll
    for (int s = 0; s < T::STAGES - 1; ++s) stage(s, s);

    const int fr = lane & 15, fq = lane >> 4;
    int slot = 0;
    for (int t = 0; t < nt; ++t) {
        // K-tile t has landed for this wave (all but the STAGES - 2 newest stagings); the barrier publishes everyone's pieces and
        // separates the reads of K-tile t - 1 from the staging that now overwrites its slot
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"((T::STAGES - 2) * T::PIECES_PER_WAVE) : "memory");
        __builtin_amdgcn_s_barrier();
        stage(t + T::STAGES - 1, slot == 0 ? T::STAGES - 1 : slot - 1);
        const char* a_tile = smem + slot * T::STAGE_BYTES;
        const char* w_tile = a_tile + T::A_BYTES;
        if (FP8) {   // one 128-deep scaled MFMA per fragment pair and K-tile (see gemm_body)
            bf16x8 af[T::FM][2], wf[T::FN][2];
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
                for (int j = 0; j < T::FN; ++j) wf[j][kk] = read_frag(w_tile, (wc * T::FN + j) * 16 + fr, kk * 4 + fq);
#pragma unroll
                for (int i = 0; i < T::FM; ++i) af[i][kk] = read_frag(a_tile, (wr * T::FM + i) * 16 + fr, kk * 4 + fq);
            }
#pragma unroll
            for (int i = 0; i < T::FM; ++i)
#pragma unroll
                for (int j = 0; j < T::FN; ++j) acc[i][j] = mfma_e4m3_16x16x128(wf[j], af[i], acc[i][j]);
        } else
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
        slot = slot + 1 == T::STAGES ? 0 : slot + 1;
    }
    gemm_epilogue_family<T, EK, OP>(p, acc, m0 + wr * T::FM * 16, n0 + wc * T::FN * 16, fr, fq, tile_stats + wr * T::FM * 16);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the surplus stagings past the end still write LDS
}

}  // namespace ivit
