// Launchers of the gfx950 kernels (one translation unit each); called only by engine.hip.
#pragma once
#include "common.h"

namespace ivit {

// Raises a kernel's dynamic-LDS limit once per (device, kernel): the attribute belongs to the device that is current
// when it is set, and one process may hold engines on several devices (ivit_config.device).
hipError_t ensure_dynamic_lds(const void* kernel, int bytes);

// ---------------------------------------------------------------- GEMM (kernels_gemm.hip)
enum GemmEpilogue : int {
    EPI_BIAS_BF16 = 0,       // out(bf16)  = acc + bias
    EPI_BIAS_GELU_BF16 = 1,  // out(bf16)  = gelu_erf(acc + bias)
    EPI_BIAS_RESID_F32 = 2,  // out(f32)   = resid + (acc + bias)          (out may alias resid)
    EPI_BIAS_F32 = 3,        // out(f32)   = acc + bias
    EPI_BIAS_ROWADD_F32 = 4, // out(f32)[remap(m)] = (acc + bias) + rowadd[grp_off + m % grp_in]
    EPI_BIAS_GELU_FP8 = 5,   // out(fp8)  = sat_fp8(gelu_erf(acc + bias) * out_scale)
    // LayerNorm folded into the GEMM that consumes it (kernel family "lnfold", see GemmParams::ln_*):
    EPI_BIAS_RESID_STATS = 6,   // EPI_BIAS_RESID_F32, plus xb = bf16(out) and the row statistics of out for the NEXT GEMM
    EPI_LNFOLD_BF16 = 7,        // out(bf16) = rstd[m] * (acc - mean[m] * ln_s[n]) + bias[n]        (A = bf16(x), W = bf16(W . gamma))
    EPI_LNFOLD_GELU_BF16 = 8,   // out(bf16) = gelu_erf(the same)
    EPI_BIAS_ROWADD_STATS = 9,  // EPI_BIAS_ROWADD_F32 (row remap + table add), plus xb and the row statistics as EPI_BIAS_RESID_STATS: the patch
                                // embedding in front of a LayerNorm-folded first layer (no ivit_row_stats_pairs pass over the token stream)
};

struct GemmParams {
    const bf16_t* A; int lda;     // [M, K] bf16 row-major, rows readable up to round_up(M,256)+256
    const bf16_t* W; int ldw;     // [N, K] bf16 row-major, rows readable up to round_up(N,256)
    int M, N, K;                  // K % 64 == 0 (operands zero-padded); fp8 operands: K % 128 == 0
    int a_wrap;                   // 0, or the number of 64-deep K-tiles A really has: K-tile t of the product reads A's K-tile t mod a_wrap.
                                  // Split-operand GEMM on pairs of BOTH operands (IVIT_PRECISION_F16X out-projection, patch, head):
                                  // W = [W_hi | W_hi | W_lo] against A = [A_hi | A_lo] (wraps to A_hi) - hi/lo pairs of f16 values summed in the f32 accumulators
    int a_shift;                  // 0, or 1: product K-tiles 2t and 2t + 1 read A's K-tile t - the weight-only split (MLP up / down of F16X): W's K-tiles alternate
                                  // [hi t | lo t], so an activation K-tile is staged and its fragments are read ONCE for both passes (gemm_body / gemm_body_sb)
    int f16;                      // 0: A, W and 16-bit outputs are bf16; 1: IEEE f16 (IVIT_PRECISION_F16)
    const float* colscale;        // fp8 operands: acc *= colscale[n] (= activation scale x weight-row scale) before the bias
    float out_scale;              // EPI_BIAS_GELU_FP8: 1 / (scale of the fp8 output tensor)
    const float* bias;            // [N]
    int epi;
    void* out; int ldo;
    const float* resid; int ldr;  // EPI_BIAS_RESID_F32
    const float* rowadd; int ldra;  // EPI_BIAS_ROWADD_F32: [grp_out, N] table (position embedding)
    int grp_in, grp_out, grp_off;   // row remap m -> (m / grp_in) * grp_out + grp_off + m % grp_in
    // LayerNorm fold.  Row statistics travel between two GEMMs as per-row, per-64-column (sum, M2) pairs:
    // EPI_BIAS_RESID_STATS writes ln_part[row][column / 64] and xb; an EPI_LNFOLD_* kernel folds the pairs of its
    // tile's rows in slot order (exact two-pass statistics, independent of who wrote them and when) at its start.
    float2* ln_part;                // EPI_BIAS_RESID_STATS: [rows][GEMM_LN_SLOTS], written
    const float2* ln_stats;         // EPI_LNFOLD_*: [rows] finished (mean, rstd), or nullptr (what the engine passes): fold ln_part
    bf16_t* xb; int ldxb;           // EPI_BIAS_RESID_STATS: bf16 copy of the new residual rows
    const float* ln_s;              // EPI_LNFOLD_*: s[n] = sum_k W'[n][k]          (bias = c[n])
    float ln_eps; int ln_dim;       // EPI_LNFOLD_*: LayerNorm epsilon and width (= K of this GEMM)
    // Centred operand copy (round 5): the 16-bit copy of the rows is rn16(x - centre[n]) with a calibrated per-channel vector (ivit_ln_fold_calibrate), so that
    // channel-constant offsets and outlier channels - what makes |mean| / std of real checkpoints' rows large - are not rounded with the rows; the consumer adds
    // d[n] = sum_k centre[k] W'[n][k] back:  LN(x) W^T + b = rstd (rn16(x - centre) W'^T + d - mean s) + c.  Statistics stay those of x.  nullptr = not centred.
    const float* ln_centre;         // EPI_BIAS_RESID_STATS / EPI_BIAS_ROWADD_STATS: [N] subtracted before the 16-bit rounding of xb
    const float* ln_d;              // EPI_LNFOLD_*: [N] added to the accumulators
    int rs_prefetch_from;           // EPI_BIAS_RESID_STATS on the two-stage tile: workgroups with blockIdx.x >= this (> 0) load their residual rows BEFORE the
                                    // K loop (the workgroups a CU receives second, when every CU gets one before any gets two); 0 = off.  gemm_kernel.h: RsPrefetch
    int debug;                      // microbenchmark ablations only (0 in the product): 1 = no DMA in the K loop, 2 = no MFMA
    unsigned long long* stamps;     // microbenchmark builds only: per-block s_memrealtime stamps (nullptr in the product)
    int order;                      // microbenchmark builds only: 1 = row bands per XCD (tile_coords_banded), 2 = pairs sharing a CU (gemm_body)
};

enum GemmVariant : int { GEMM_TILE_128 = 0, GEMM_TILE_160 = 1, GEMM_TILE_256 = 2, GEMM_TILE_256P = 3, GEMM_TILE_256S = 4, GEMM_TILE_256PS = 5, GEMM_TILE_160X256 = 6, GEMM_TILE_160X256W4 = 7, GEMM_TILE_PE = 8, GEMM_TILE_64D = 9, GEMM_TILE_P160 = 10, GEMM_TILE_P128 = 11, GEMM_TILE_128W8A = 12, GEMM_TILE_128W8B = 13, GEMM_TILE_160W8 = 14, GEMM_TILE_128SB = 15, GEMM_TILE_160SB = 16, GEMM_TILE_256X128SB = 17, GEMM_TILE_192SB = 18, GEMM_VARIANTS = 19 };
// product variants: GEMM_TILE_128SB / 160SB (one operand stage, three workgroups per CU), GEMM_TILE_256S, GEMM_TILE_64D; the others exist only in
// microbenchmark builds (IVIT_GEMM_ABLATIONS: two-stage tiles, persistent kernels, eight-wave forms - csrc/study/)
// GEMM_TILE_P160 / P128 (study/gemmp_kernel.h: persistent two-per-CU workgroups, the finished tile's LayerNorm-fold epilogue drained inside
// the next tile's main loop) serve EPI_LNFOLD_* only; gemm_persist_supported says whether a call can take them (false in the product build)
bool gemm_persist_supported(const GemmParams& p);
// GEMM_TILE_PE (study/gemmpe_kernel.h: persistent 256 x 128, epilogue interleaved into the next tile's main loop) serves the 16-bit-output
// epilogues only; gemm_pe_supported says whether a call can take it (false in the product build)
bool gemm_pe_supported(const GemmParams& p);
// fp8 (e4m3) operands, f32 accumulate: A [M,K] and W [N,K] are BYTE matrices (lda/ldw in elements = bytes)
hipError_t launch_gemm_fp8(const GemmParams& p, hipStream_t stream);
// Operand allocations must be readable up to the tile edge: A rows up to round_up(M,256)+256,
// W rows up to round_up(N,256) (engine.hip pads every buffer accordingly).
hipError_t launch_gemm(const GemmParams& p, hipStream_t stream);                       // picks the tile
hipError_t launch_gemm_variant(const GemmParams& p, int variant, hipStream_t stream);
int gemm_pick_variant(int M, int N, int K);
int gemm_tail_rows(const GemmParams& p, bool fp8);                      // rows to peel off the end of M so that the 256 x 256 grid does not end on a nearly empty round (0: none)
GemmParams gemm_rows_from(const GemmParams& p, int m_off, bool fp8);   // the same launch for rows m_off .. M
bool gemm_prefers_256(int M, int N, int K);   // the one-workgroup-per-CU 256x256 tile is picked (many rounds of tiles)
const char* gemm_variant_name(int v);
// name of the kernel launch_gemm / launch_gemm_fp8 dispatches for these parameters (per-kernel profiling, dispatch tests)
const char* gemm_kernel_name(const GemmParams& p);
const char* gemm_fp8_kernel_name(const GemmParams& p);

// ---------------------------------------------------------------- fused MLP (kernels_mlp.hip, mlp_fused_kernel.h)
// One launch for LN2 (folded) -> MLP up -> GELU -> MLP down -> residual add (-> 16-bit copy + row statistics of the new rows): a workgroup owns 64
// token rows and streams both weight matrices once, straight from L2 into MFMA operand registers; the hidden activations never leave the CU.
// Bit-identical to launch_gemm(EPI_LNFOLD_GELU_BF16) followed by launch_gemm(EPI_BIAS_RESID_STATS / EPI_BIAS_RESID_F32) on the same operands.
struct MlpFusedParams {
    const bf16_t* X; int ldx;            // [M, D] 16-bit operand copy of the LayerNorm input rows (readable up to round_up(M, 64) rows)
    const float2* ln_part_in;            // [M][GEMM_LN_SLOTS] (sum, M2) pairs of those rows
    float ln_eps;
    const bf16_t* Wp;                    // both weight matrices in the kernel's fragment-native stream order (launch_mlp_pack_weights)
    const float* c1; const float* s1;    // [Mlp] fold vectors: c = W beta + b, s = row sums of W'
    const float* b2;                     // [D]
    const float* resid; int ldr;         // [M, D] f32 residual rows
    float* out; int ldo;                 // [M, D] f32 (may alias resid)
    bf16_t* xb; int ldxb;                // stats_out: 16-bit copy of the new rows (may alias X: a workgroup reads its rows before it writes them)
    float2* ln_part_out;                 // stats_out: their pairs (may alias ln_part_in)
    const float* d1;                     // [Mlp] or nullptr: X holds rn16(x - centre); d1[n] = sum_k centre[k] W1'[n][k] is added to the up product (GemmParams::ln_d)
    const float* centre_out;             // [D] or nullptr: stats_out writes xb = rn16(new rows - centre_out) (GemmParams::ln_centre)
    int M, D, Mlp;
    int f16;                             // 0: bf16 operands, 1: IEEE f16
    int split;                           // 0; 1: both weight matrices are hi / lo pairs; 2: only the up weight (IVIT_PRECISION_F16X's default split set)
    int stats_out;                       // 1: EPI_BIAS_RESID_STATS semantics, 0: EPI_BIAS_RESID_F32
    unsigned long long* stamps;          // microbenchmark builds only
};
bool mlp_fused_supported(int M, int D, int Mlp, int f16, int split);
size_t mlp_fused_packed_bytes(int D, int Mlp, int split);
// W1 [Mlp, D] (LayerNorm-folded up weight; split: hi / lo K-tiles interleaved, ldw1 >= 2 D), W2 [D, Mlp] (split: likewise, ldw2 >= 2 Mlp) -> out
hipError_t launch_mlp_pack_weights(const bf16_t* W1, int ldw1, const bf16_t* W2, int ldw2, int D, int Mlp, int split, bf16_t* out, hipStream_t stream);
hipError_t launch_mlp_fused(const MlpFusedParams& p, hipStream_t stream);
const char* mlp_fused_kernel_name(const MlpFusedParams& p);

// ---------------------------------------------------------------- attention (kernels_attn.hip)
struct AttnParams {
    const bf16_t* qkv; int ldqkv;  // [B*N, 3D] bf16: q | k | v, head h at columns h*dh
    // head-major form (both 0 = the row-major form above): element (which in q/k/v, head h, token row r, d) lives at
    // qkv + which * which_stride + h * head_stride + r * ldqkv + d, with ldqkv = dh: a head's K / V rows are contiguous in memory
    int64_t head_stride, which_stride;
    bf16_t* out; int ldo;          // [B*N, D] bf16
    int batch, tokens, heads, head_dim;
    float scale;                   // 1/sqrt(dh)
    int f16;                       // 0: q|k|v, P and the output are bf16; 1: IEEE f16
    int lo_off;                    // 0, or the element offset from an output value to its low part: out = rn16(O), out[+lo_off] = rn16(O - out)
    float* probs;                  // nullptr, or f32 [B, H, N, N]: write the attention probabilities instead of P.V
    unsigned char* out8; int ldo8; // nullptr, or e4m3 [B*N, D]: write sat_fp8(O * scale8) INSTEAD of the bf16 output
    float scale8;
};
hipError_t launch_attention(const AttnParams& p, hipStream_t stream);
const char* attention_kernel_name(const AttnParams& p);   // "ivit_attention_bf16" (one pass, <= 288 tokens or head dim 80) or "ivit_attention_q32"
bool attention_supported(int tokens, int head_dim);

#ifdef IVIT_GEMM_ABLATIONS   // study kernel (csrc/study/fused_qkv_attention.inc), microbenchmark builds only
// Fused QKV projection (LayerNorm folded) + attention, one workgroup per (image, head), for <= 224 tokens at head dim 64: the
// q|k|v tensor stays in LDS (ivit_qkv_attention_fused).  Same values as launch_gemm(EPI_LNFOLD_BF16) followed by
// launch_attention, bit for bit.
struct FusedQkvAttnArgs {
    const bf16_t* x; int ldx;            // [rows_total, D] 16-bit operand copy of the residual stream
    const bf16_t* w; int ldw;            // [3D, D] LayerNorm-folded in_proj weight
    const float* c; const float* s;      // [3D] fold vectors (bias + W beta; row sums of W')
    const float2* ln_part; const float2* ln_stats; float ln_eps;   // row statistics: per-slot pairs, or finished (mean, rstd)
    bf16_t* out; int ldo;                // [rows_total, D] attention output
    bf16_t* qkv_dbg; int ldq;            // nullptr, or [rows_total, 3D]: also store q|k|v (inspection taps)
    int batch, tokens, heads, head_dim, dim, rows_total, f16;
    float scale;
    unsigned long long* stamps;          // microbenchmark builds only: 8 s_memrealtime stamps per workgroup (nullptr in the product)
    int debug;                           // microbenchmark builds only: ablations (0 in the product)
};
bool fused_qkv_attention_supported(int tokens, int head_dim, int dim);
hipError_t launch_fused_qkv_attention(const FusedQkvAttnArgs& a, hipStream_t stream);
#endif

// ---------------------------------------------------------------- misc (kernels_misc.hip)
// (x - mean[c]) / std[c] on [B,3,S,S] f32
hipError_t launch_transform(const float* in, float* out, int batch, int image, hipStream_t s);
// [B,3,H,W] in [0,1] -> [B,3,image,image]: antialiased bilinear resize of the shorter side to `resize`, centre crop, normalise
hipError_t launch_preprocess(const float* in, int H, int W, float* out, int batch, int image, int resize, hipStream_t s);
// f32 image -> bf16 unfold matrix [B*Np, kpad] (columns >= 3p^2 zero); normalise fuses the transform
// split: rows are [hi | lo] of 2 * kpad columns (hi = rn16(x), lo = rn16(x - hi)) for the split-operand patch GEMM
hipError_t launch_unfold(const float* in, bf16_t* out, int batch, int image, int patch, int kpad,
                         int normalise, hipStream_t s, int f16 = 0, int split = 0);
// out[b,0,:] = cls + pos[0]; out[b,1+n,:] = in[b,n,:] + pos[1+n]   (in == nullptr: class rows only)
hipError_t launch_tokens(const float* in, const float* cls, const float* pos, float* out, int batch,
                         int patches, int dim, hipStream_t s);
// LayerNorm over the last dim of rows `row0 + i*row_stride` (i < rows) of a [*, dim] f32 matrix;
// writes bf16 (out_bf16, ld = ldo16) and/or f32 (out_f32, ld = ldo32) at row i.
// optional third output: e4m3 (out_fp8, ld = ldo8 bytes) = sat_fp8(y * scale8)
hipError_t launch_layernorm(const float* x, int ldx, int64_t row_stride, int rows, int dim, const float* gamma,
                            const float* beta, float eps, bf16_t* out_bf16, int ldo16, float* out_f32,
                            int ldo32, hipStream_t s, unsigned char* out_fp8 = nullptr, int ldo8 = 0, float scale8 = 1.0f, int f16 = 0,
                            int lo_off16 = 0);   // lo_off16 != 0: out_bf16[+lo_off16] = rn16(y - rn16(y)) as well
// fp8 support: tensor amax (atomicMax into *out, which the caller zeroes), per-row weight quantisation, vector scale
hipError_t launch_amax_bf16(const bf16_t* in, int ld, int rows, int cols, float* out, hipStream_t s);
hipError_t launch_quantize_weight_fp8(const bf16_t* w, int ld, int rows, int cols, unsigned char* w8, int ld8, float* rowscale,
                                      hipStream_t s);
hipError_t launch_scale_vec(const float* in, float a, float* out, int n, hipStream_t s);
// LayerNorm folded into the GEMM that follows it: weight preparation and row statistics
constexpr int GEMM_LN_SLOTS = 32;   // 64-column statistics slots per row (dim <= 2048)
hipError_t launch_fold_ln_weights(const bf16_t* w, int ld, int rows, int cols, const float* gamma, const float* beta, const float* bias,
                                  bf16_t* wf, float* s_out, float* c_out, hipStream_t s, int f16 = 0);
hipError_t launch_row_stats(const float* x, int ldx, int rows, int dim, bf16_t* xb, int ldxb, float2* part, hipStream_t s, int f16 = 0, int row_step = 1, const float* centre = nullptr);   // part: [rows][GEMM_LN_SLOTS] (sum, M2) pairs, as EPI_BIAS_RESID_STATS writes them
// atomicMax(*out, max over rows of |mean| / sqrt(var + eps)) of a [rows, dim] f32 matrix; the caller zeroes *out
hipError_t launch_row_mean_ratio(const float* x, int ldx, int rows, int dim, float eps, const float* centre, float* out, hipStream_t s);   // out[0]: plain copy, out[1]: centred copy
hipError_t launch_ln_finalize(const float2* part, int rows, int dim, float eps, float2* stats, hipStream_t s);   // pairs -> finished (mean, rstd) per row
hipError_t launch_col_means(const float* x, int ldx, int rows, int dim, float* out, hipStream_t s);
hipError_t launch_centre_dot(const bf16_t* w, int ld, int rows, int cols, int split, const float* centre, float* d, hipStream_t s, int f16);
// strided row gather: out[i,:] = in[i*row_stride, :dim]  (the `cls` node)
hipError_t launch_gather_rows(const float* in, int64_t row_stride, float* out, int rows, int dim, hipStream_t s);
// f32 [rows, cols] -> bf16 [rows, ldo] (columns >= cols zero)
hipError_t launch_f32_to_bf16(const float* in, int ldi, bf16_t* out, int ldo, int rows, int cols, hipStream_t s, int f16 = 0);
// hi/lo pairs of 16-bit values from an f32 matrix [rows, cols] (optionally times gamma[k]: the LayerNorm-folded weight from the f32
// original, one rounding): out rows are [hi | hi | lo] (both = 1: the other operand is split too; each part kpad columns, zero padded)
// or hi / lo interleaved per 64-column K-tile (both = 0: [hi t0 | lo t0 | hi t1 | ...], 2 * kpad columns; GemmParams::a_shift).  s_out / c_out (optional): s[n] = sum_k (hi + lo), c[n] = sum_k beta[k] w[n][k] + bias[n].
hipError_t launch_split_weight(const float* w, int ldw, int rows, int cols, const float* gamma, const float* beta, const float* bias,
                               bf16_t* out, int ld_out, int kpad, int both, float* s_out, float* c_out, hipStream_t s, int f16);
// hi/lo pair of a f32 [rows, cols] activation: out row = [hi | lo], each part ldhalf columns (zero padded)
hipError_t launch_f32_to_split16(const float* in, int ldi, bf16_t* out, int ldhalf, int rows, int cols, hipStream_t s, int f16);
// bf16 [rows, ldi] -> f32 [rows, cols]
hipError_t launch_bf16_to_f32(const bf16_t* in, int ldi, float* out, int rows, int cols, hipStream_t s, int f16 = 0);

}  // namespace ivit
