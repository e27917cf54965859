// HBM-bound glue kernels of the ViT forward: image normalisation, unfold (im2col) to bf16,
// class-token / position-embedding assembly, LayerNorm (wavefront-shuffle reductions), row gather
// and dtype conversion.  All are coalesced 16-B-per-lane streams; none is reshaped into a GEMM.
#include "kernels.h"

namespace ivit {

// the 16-bit operand type picked at run time (memory-bound kernels: a uniform select costs nothing)
__device__ __forceinline__ unsigned int pack16x2(int f16, float lo, float hi) { return f16 ? pack_f16x2(lo, hi) : pack_bf16x2(lo, hi); }
__device__ __forceinline__ float dec16(int f16, bf16_t b) { return f16 ? OpF16::to_f32(b) : bf2f(b); }
__device__ __forceinline__ bf16_t enc16(int f16, float x) { return f16 ? OpF16::from_f32(x) : f2bf(x); }

__constant__ float c_mean[3] = {0.485f, 0.456f, 0.406f};
__constant__ float c_std[3] = {0.229f, 0.224f, 0.225f};

constexpr int EW_THREADS = 256;
static inline int ew_grid(int64_t work_items) {
    int64_t blocks = (work_items + EW_THREADS - 1) / EW_THREADS;
    return (int)(blocks < 1 ? 1 : (blocks > 2048 * 4 ? 2048 * 4 : blocks));  // grid-stride beyond that
}

// ---------------------------------------------------------------------------- transform
__global__ void ivit_transform(const float* __restrict__ in, float* __restrict__ out, int64_t total, int plane) {
    // total = B*3*S*S, plane = S*S (multiple of 4 for every supported S)
    const int64_t n4 = total >> 2;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(((i << 2) / plane) % 3);
        const float m = c_mean[c], s = c_std[c];
        float4 v = reinterpret_cast<const float4*>(in)[i];
        v.x = (v.x - m) / s; v.y = (v.y - m) / s; v.z = (v.z - m) / s; v.w = (v.w - m) / s;
        reinterpret_cast<float4*>(out)[i] = v;
    }
    const int64_t tail = n4 << 2;   // plane % 4 != 0 only for exotic sizes
    for (int64_t i = tail + blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)((i / plane) % 3);
        out[i] = (in[i] - c_mean[c]) / c_std[c];
    }
}

hipError_t launch_transform(const float* in, float* out, int batch, int image, hipStream_t s) {
    const int64_t total = (int64_t)batch * 3 * image * image;
    hipLaunchKernelGGL(ivit_transform, dim3(ew_grid(total / 4 + 1)), dim3(EW_THREADS), 0, s, in, out, total, image * image);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------- preprocess
// The classification preset of the reference's model plugin (static/models/vgg16.py:40-42 applies
// `weights.transforms()`): resize the shorter side to R with the antialiased bilinear filter, centre-crop
// S x S, normalise.  Weights as in ATen's upsample_bilinear2d_aa (align_corners = false): scale = in / out,
// support = max(scale, 1), centre = scale (i + 0.5), taps [int(centre - support + 0.5), int(centre + support
// + 0.5)) clipped to the image, triangle weights (1 - |(j - centre + 0.5) / max(scale, 1)|)+, normalised.
// One thread per output pixel, both dimensions filtered at once (ATen filters W then H: same sum, other
// rounding order).
__device__ __forceinline__ void aa_taps(int out_i, float scale, int in_size, int& first, int& count, float& support, float& centre) {
    support = scale >= 1.0f ? scale : 1.0f;
    centre = scale * ((float)out_i + 0.5f);
    first = max((int)(centre - support + 0.5f), 0);
    count = min((int)(centre + support + 0.5f), in_size) - first;
}
__device__ __forceinline__ float aa_weight(int j, int first, float centre, float scale) {
    const float inv = scale >= 1.0f ? 1.0f / scale : 1.0f;
    const float x = ((float)(j + first) - centre + 0.5f) * inv;
    const float a = fabsf(x);
    return a < 1.0f ? 1.0f - a : 0.0f;
}

__global__ void ivit_preprocess(const float* __restrict__ in, int H, int W, float* __restrict__ out, int S, int RH, int RW,
                                int top, int left, int batch) {
    const float sh = (float)H / (float)RH, sw = (float)W / (float)RW;
    const int64_t total = (int64_t)batch * 3 * S * S;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % S), y = (int)((i / S) % S);
        const int c = (int)((i / ((int64_t)S * S)) % 3);
        const int64_t img = i / ((int64_t)3 * S * S);
        int y0, ny, x0, nx; float suph, ch, supw, cw;
        aa_taps(y + top, sh, H, y0, ny, suph, ch);
        aa_taps(x + left, sw, W, x0, nx, supw, cw);
        float wxs = 0.f, wys = 0.f;
        for (int j = 0; j < nx; ++j) wxs += aa_weight(j, x0, cw, sw);
        for (int j = 0; j < ny; ++j) wys += aa_weight(j, y0, ch, sh);
        const float* src = in + (img * 3 + c) * (int64_t)H * W;
        float acc = 0.f;
        for (int jy = 0; jy < ny; ++jy) {
            const float wy = aa_weight(jy, y0, ch, sh) / wys;
            const float* row = src + (int64_t)(y0 + jy) * W + x0;
            float r = 0.f;
            for (int jx = 0; jx < nx; ++jx) r += (aa_weight(jx, x0, cw, sw) / wxs) * row[jx];
            acc += wy * r;
        }
        out[i] = (acc - c_mean[c]) / c_std[c];
    }
}

hipError_t launch_preprocess(const float* in, int H, int W, float* out, int batch, int image, int resize, hipStream_t s) {
    if (H < 1 || W < 1 || resize < image) return hipErrorInvalidValue;
    // torchvision: shorter side -> resize, longer side -> int(resize * long / short); crop offsets int(round((r - S) / 2))
    int rh, rw;
    if (H <= W) { rh = resize; rw = (int)((int64_t)resize * W / H); } else { rw = resize; rh = (int)((int64_t)resize * H / W); }
    const int top = (int)lrintf((float)(rh - image) / 2.0f), left = (int)lrintf((float)(rw - image) / 2.0f);
    const int64_t total = (int64_t)batch * 3 * image * image;
    hipLaunchKernelGGL(ivit_preprocess, dim3(ew_grid(total)), dim3(EW_THREADS), 0, s, in, H, W, out, image, rh, rw, top, left, batch);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------- unfold (im2col)
// One thread produces 8 consecutive k of one patch row n: 16 B of bf16.  For p % 8 == 0 the 8
// sources are 8 consecutive pixels of one image row (two float4 loads); otherwise each element is
// located through unfold_offset().  Columns k >= 3p^2 (K padding up to a multiple of 64) are zero.
// split (f16 split-operand patch GEMM): an output row is [hi | lo], 2 * kpad columns, hi = rn16(x), lo = rn16(x - hi).
__global__ void ivit_unfold(const float* __restrict__ in, bf16_t* __restrict__ out, int batch, int image,
                            int patch, int kpad, int normalise, int f16, int split) {
    const int g = image / patch;
    const int np = g * g;
    const int kreal = 3 * patch * patch;
    const int kchunks = kpad >> 3;
    const int64_t total = (int64_t)batch * np * kchunks;
    const int64_t img_elems = (int64_t)3 * image * image;
    const int pp = patch * patch;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int kc = (int)(i % kchunks);
        const int64_t row = i / kchunks;            // b*Np + n
        const int n = (int)(row % np);
        const int b = (int)(row / np);
        const float* img = in + b * img_elems;
        const int k0 = kc * 8;
        float v[8];
        if ((patch & 7) == 0 && k0 + 8 <= kreal) {
            const float* src = img + unfold_offset(image, patch, n, k0);
            const float4 lo = *reinterpret_cast<const float4*>(src);
            const float4 hi = *reinterpret_cast<const float4*>(src + 4);
            v[0] = lo.x; v[1] = lo.y; v[2] = lo.z; v[3] = lo.w; v[4] = hi.x; v[5] = hi.y; v[6] = hi.z; v[7] = hi.w;
            if (normalise) {
                const int c = k0 / pp;
                const float m = c_mean[c], s = c_std[c];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (v[e] - m) / s;
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int k = k0 + e;
                float x = 0.f;
                if (k < kreal) {
                    x = img[unfold_offset(image, patch, n, k)];
                    if (normalise) { const int c = k / pp; x = (x - c_mean[c]) / c_std[c]; }
                }
                v[e] = x;
            }
        }
        u32x4 pk = {pack16x2(f16, v[0], v[1]), pack16x2(f16, v[2], v[3]), pack16x2(f16, v[4], v[5]), pack16x2(f16, v[6], v[7])};
        if (!split) {
            *reinterpret_cast<u32x4*>(out + row * kpad + k0) = pk;
        } else {
            float l[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) l[e] = v[e] - dec16(f16, enc16(f16, v[e]));
            u32x4 pl = {pack16x2(f16, l[0], l[1]), pack16x2(f16, l[2], l[3]), pack16x2(f16, l[4], l[5]), pack16x2(f16, l[6], l[7])};
            *reinterpret_cast<u32x4*>(out + row * 2 * kpad + k0) = pk;
            *reinterpret_cast<u32x4*>(out + row * 2 * kpad + kpad + k0) = pl;
        }
    }
}

hipError_t launch_unfold(const float* in, bf16_t* out, int batch, int image, int patch, int kpad, int normalise,
                         hipStream_t s, int f16, int split) {
    const int g = image / patch;
    const int64_t total = (int64_t)batch * g * g * (kpad / 8);
    hipLaunchKernelGGL(ivit_unfold, dim3(ew_grid(total)), dim3(EW_THREADS), 0, s, in, out, batch, image, patch, kpad, normalise, f16, split);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------- tokens
__global__ void ivit_tokens(const float* __restrict__ in, const float* __restrict__ cls, const float* __restrict__ pos,
                            float* __restrict__ out, int batch, int patches, int dim) {
    const int d4 = dim >> 2;
    const int N = patches + 1;
    const int rows_per_img = in ? N : 1;
    const int64_t total = (int64_t)batch * rows_per_img * d4;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % d4);
        const int64_t r = i / d4;
        const int t = (int)(r % rows_per_img);
        const int b = (int)(r / rows_per_img);
        const float4 pe = reinterpret_cast<const float4*>(pos + (size_t)t * dim)[c];
        float4 v;
        if (t == 0) v = reinterpret_cast<const float4*>(cls)[c];
        else v = reinterpret_cast<const float4*>(in + ((size_t)b * patches + (t - 1)) * dim)[c];
        v.x += pe.x; v.y += pe.y; v.z += pe.z; v.w += pe.w;
        reinterpret_cast<float4*>(out + ((size_t)b * N + t) * dim)[c] = v;
    }
}

hipError_t launch_tokens(const float* in, const float* cls, const float* pos, float* out, int batch, int patches,
                         int dim, hipStream_t s) {
    if (dim % 4) return hipErrorInvalidValue;
    const int64_t total = (int64_t)batch * (in ? patches + 1 : 1) * (dim / 4);
    hipLaunchKernelGGL(ivit_tokens, dim3(ew_grid(total)), dim3(EW_THREADS), 0, s, in, cls, pos, out, batch, patches, dim);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------- LayerNorm
// One wavefront per LN_RPW rows; every row (dim <= 64*4*VPL floats) stays in registers between the
// two statistics passes (mean, then centred variance - the same two-pass form as the oracle), so x
// is read from HBM once.  All LN_RPW rows' loads are issued before the first reduction (the
// one-row-per-wave version spent 70 % of its wave cycles waiting on its single load batch).
// Reductions are 64-lane xor-shuffles.
constexpr int LN_RPW = 2;   // rows per wave (ViT-B/16, B=64, 25 launches: 1 row 0.348 ms, 2 rows 0.328 ms, 4 rows 0.386 ms)
template <int VPL>  // float4 vectors per lane
__global__ __launch_bounds__(256) void ivit_layernorm(const float* __restrict__ x, int ldx, int64_t row_stride, int rows,
                                                      int dim, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, float eps,
                                                      bf16_t* __restrict__ o16, int ldo16, float* __restrict__ o32,
                                                      int ldo32, unsigned char* __restrict__ o8, int ldo8, float scale8, int f16, int lo_off16) {
    const int lane = threadIdx.x & 63;
    const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * LN_RPW;
    if (row0 >= rows) return;
    const int d4 = dim >> 2;
    float4 v[LN_RPW][VPL];
#pragma unroll
    for (int r = 0; r < LN_RPW; ++r) {
        const int row = min(row0 + r, rows - 1);          // clamp: the surplus rows are loaded, never stored
        const float* xr = x + (size_t)row * row_stride * ldx;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int c = i * 64 + lane;
            v[r][i] = (c < d4) ? reinterpret_cast<const float4*>(xr)[c] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    float4 gm[VPL], bt[VPL];
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int c = i * 64 + lane;
        gm[i] = (c < d4) ? reinterpret_cast<const float4*>(gamma)[c] : make_float4(0.f, 0.f, 0.f, 0.f);
        bt[i] = (c < d4) ? reinterpret_cast<const float4*>(beta)[c] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int r = 0; r < LN_RPW; ++r) {
        const int row = row0 + r;
        if (row >= rows) break;   // wave-uniform
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < VPL; ++i) sum += (v[r][i].x + v[r][i].y) + (v[r][i].z + v[r][i].w);
        const float mean = wave_sum(sum) / (float)dim;
        float sq = 0.f;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int c = i * 64 + lane;
            if (c < d4) {
                const float a = v[r][i].x - mean, b = v[r][i].y - mean, cc = v[r][i].z - mean, d = v[r][i].w - mean;
                sq += (a * a + b * b) + (cc * cc + d * d);
            }
        }
        const float rstd = 1.0f / sqrtf(wave_sum(sq) / (float)dim + eps);
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int c = i * 64 + lane;
            if (c < d4) {
                float4 y;
                y.x = (v[r][i].x - mean) * rstd * gm[i].x + bt[i].x;
                y.y = (v[r][i].y - mean) * rstd * gm[i].y + bt[i].y;
                y.z = (v[r][i].z - mean) * rstd * gm[i].z + bt[i].z;
                y.w = (v[r][i].w - mean) * rstd * gm[i].w + bt[i].w;
                if (o32) reinterpret_cast<float4*>(o32 + (size_t)row * ldo32)[c] = y;
                if (o16) {
                    u32x2 pk = {pack16x2(f16, y.x, y.y), pack16x2(f16, y.z, y.w)};
                    reinterpret_cast<u32x2*>(o16 + (size_t)row * ldo16)[c] = pk;
                    if (lo_off16) {   // low parts for a split-operand GEMM
                        u32x2 pl = {pack16x2(f16, y.x - dec16(f16, enc16(f16, y.x)), y.y - dec16(f16, enc16(f16, y.y))),
                                    pack16x2(f16, y.z - dec16(f16, enc16(f16, y.z)), y.w - dec16(f16, enc16(f16, y.w)))};
                        reinterpret_cast<u32x2*>(o16 + (size_t)row * ldo16 + lo_off16)[c] = pl;
                    }
                }
                if (o8)   // e4m3 with the tensor's calibrated scale (scale8 = 1 / scale)
                    reinterpret_cast<unsigned int*>(o8 + (size_t)row * ldo8)[c] = pack_fp8x4(y.x * scale8, y.y * scale8, y.z * scale8, y.w * scale8);
            }
        }
    }
}

hipError_t launch_layernorm(const float* x, int ldx, int64_t row_stride, int rows, int dim, const float* gamma,
                            const float* beta, float eps, bf16_t* o16, int ldo16, float* o32, int ldo32,
                            hipStream_t s, unsigned char* o8, int ldo8, float scale8, int f16, int lo_off16) {
    if (dim % 4 || dim > 64 * 4 * 8 || (lo_off16 % 4)) return hipErrorInvalidValue;
    if (rows <= 0) return hipSuccess;
    const dim3 grid(ceil_div(rows, 4 * LN_RPW)), block(256);
    const int vpl = ceil_div(dim / 4, 64);
#define IVIT_LN(V) hipLaunchKernelGGL(ivit_layernorm<V>, grid, block, 0, s, x, ldx, row_stride, rows, dim, gamma, beta, eps, o16, ldo16, o32, ldo32, o8, ldo8, scale8, f16, lo_off16)
    if (vpl <= 1) IVIT_LN(1);
    else if (vpl <= 2) IVIT_LN(2);
    else if (vpl <= 3) IVIT_LN(3);
    else if (vpl <= 4) IVIT_LN(4);
    else if (vpl <= 5) IVIT_LN(5);
    else IVIT_LN(8);
#undef IVIT_LN
    return hipGetLastError();
}

// ---------------------------------------------------------------------------- LayerNorm folded into the next GEMM
// LN(x) W^T + b  =  rstd (x W'^T - mu s) + c     with  W' = W . diag(gamma),  s_n = sum_k W'_nk,  c_n = sum_k beta_k W_nk + b_n.
// The GEMM multiplies bf16(x) by bf16(W') and its epilogue applies the row statistics (kernels.h: EPI_LNFOLD_*);
// this kernel prepares W', s and c from the engine's bf16 copy of W.  One wave per weight row.
__global__ __launch_bounds__(256) void ivit_fold_ln_weights(const bf16_t* __restrict__ w, int ld, int rows, int cols,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            const float* __restrict__ bias, bf16_t* __restrict__ wf, float* __restrict__ s_out,
                                                            float* __restrict__ c_out, int f16) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const bf16_t* wr = w + (size_t)row * ld;
    bf16_t* fr = wf + (size_t)row * ld;
    float s = 0.f, c = 0.f;
    for (int k = lane; k < ld; k += 64) {
        float folded = 0.f;
        if (k < cols) {
            const float wv = dec16(f16, wr[k]);
            const bf16_t q = enc16(f16, wv * gamma[k]);
            folded = dec16(f16, q);
            c = fmaf(beta[k], wv, c);
            fr[k] = q;
        } else {
            fr[k] = 0;   // K padding stays zero
        }
        s += folded;
    }
    s = wave_sum(s);
    c = wave_sum(c);
    if (lane == 0) { s_out[row] = s; c_out[row] = c + bias[row]; }
}

hipError_t launch_fold_ln_weights(const bf16_t* w, int ld, int rows, int cols, const float* gamma, const float* beta, const float* bias,
                                  bf16_t* wf, float* s_out, float* c_out, hipStream_t s, int f16) {
    hipLaunchKernelGGL(ivit_fold_ln_weights, dim3(ceil_div(rows, 4)), dim3(256), 0, s, w, ld, rows, cols, gamma, beta, bias, wf, s_out, c_out, f16);
    return hipGetLastError();
}

// Row statistics of the residual stream and its 16-bit copy where no GEMM produced the stream (the first layer of a call): the
// per-row, per-64-column (sum, M2) pairs and the copy that EPI_BIAS_RESID_STATS leaves behind (gemm_kernel.h:
// gemm_epilogue_resid_stats), computed in the SAME order - four threads per (row, slot), thread fq summing the column quads
// j * 16 + fq * 4 .. + 3 (j = 0..3) one after the other from 0, then (s0 + s1) + (s2 + s3); M2 about sum / ncols by explicit fma
// in that order - so a layer gives the same bits whether its input came out of the previous layer's MLP-down GEMM in the same call,
// out of an earlier call (chained nodes, which then skip this kernel) or from the caller.
// `centre` (nullptr = none): the 16-bit copy is rn16(x - centre[n]), as GemmParams::ln_centre; the pairs are those of x itself.
__global__ __launch_bounds__(256) void ivit_row_stats_pairs(const float* __restrict__ x, int ldx, int rows, int dim, bf16_t* __restrict__ xb, int ldxb,
                                                            float2* __restrict__ part, int f16, int row_step, const float* __restrict__ centre) {
    const int nslots = (dim + 63) >> 6, per_row = nslots * 4;
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int row = (int)(t / per_row);
    const int rem = (int)(t - (int64_t)row * per_row), slot = rem >> 2, fq = rem & 3;
    const bool live = row < rows;
    if (!live) row = rows - 1;                 // every lane stays for the shuffles
    row *= row_step;                           // (row_step = tokens: the class rows of a batch only)
    const int n_base = slot * 64, ncols = min(64, dim - n_base);
    const float* xr = x + (size_t)row * ldx;
    float v[4][4];
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n = n_base + j * 16 + fq * 4;
        if (n + 3 < dim) {
            const float4 q = *reinterpret_cast<const float4*>(xr + n);
            v[j][0] = q.x; v[j][1] = q.y; v[j][2] = q.z; v[j][3] = q.w;
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[j][r] = (n + r < dim) ? xr[n + r] : 0.f;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) if (n + r < dim) sum += v[j][r];
        if (live) {
            bf16_t* ob = xb + (size_t)row * ldxb + n;
            float ct[4] = {0.f, 0.f, 0.f, 0.f};
            if (centre) for (int r = 0; r < 4; ++r) if (n + r < dim) ct[r] = centre[n + r];
            if (n + 3 < dim) {
                u32x2 pk = {pack16x2(f16, v[j][0] - ct[0], v[j][1] - ct[1]), pack16x2(f16, v[j][2] - ct[2], v[j][3] - ct[3])};
                *reinterpret_cast<u32x2*>(ob) = pk;
            } else {
                for (int r = 0; r < 4; ++r) if (n + r < dim) ob[r] = enc16(f16, v[j][r] - ct[r]);
            }
        }
    }
    sum += __shfl_xor(sum, 1, 64);
    sum += __shfl_xor(sum, 2, 64);
    const float lmean = ncols > 0 ? sum / (float)ncols : 0.f;
    float m2 = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n = n_base + j * 16 + fq * 4;
#pragma unroll
        for (int r = 0; r < 4; ++r) if (n + r < dim) { const float d = v[j][r] - lmean; m2 = fmaf(d, d, m2); }
    }
    m2 += __shfl_xor(m2, 1, 64);
    m2 += __shfl_xor(m2, 2, 64);
    if (fq == 0 && live && ncols > 0) part[(size_t)row * GEMM_LN_SLOTS + slot] = make_float2(sum, m2);
}

hipError_t launch_row_stats(const float* x, int ldx, int rows, int dim, bf16_t* xb, int ldxb, float2* part, hipStream_t s, int f16, int row_step, const float* centre) {
    if (dim % 4 || dim > 64 * GEMM_LN_SLOTS || row_step < 1) return hipErrorInvalidValue;
    if (rows <= 0) return hipSuccess;
    const int64_t threads = (int64_t)rows * ((dim + 63) >> 6) * 4;
    hipLaunchKernelGGL(ivit_row_stats_pairs, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, x, ldx, rows, dim, xb, ldxb, part, f16, row_step, centre);
    return hipGetLastError();
}

// Calibration of the LayerNorm fold (ivit_ln_fold_calibrate).  The folded GEMM multiplies a 16-bit copy of the rows that is NOT centred per row: its
// operand-rounding noise is rms(copy) / std(x) times that of the unfolded form, which rounds (x - mean) rstd.  For the plain copy rn16(x) that factor is
// sqrt(1 + (mean / std)^2); for the centred copy rn16(x - centre) it is sqrt(meansq(x - centre) / var(x)).  The statistic is the factor as
// sqrt(factor^2 - 1) (so that the uncentred one reads |mean| / std), max over rows: out[0] of the plain copy, out[1] of the centred copy
// (centre = nullptr: out[1] = out[0]).  One wave per row.
__global__ __launch_bounds__(256) void ivit_row_mean_ratio(const float* __restrict__ x, int ldx, int rows, int dim, float eps, const float* __restrict__ centre,
                                                           unsigned int* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + (size_t)row * ldx;
    float sum = 0.f;
    for (int c = lane; c < dim; c += 64) sum += xr[c];
    const float mean = wave_sum(sum) / (float)dim;
    float sq = 0.f, zq = 0.f;
    for (int c = lane; c < dim; c += 64) {
        const float d = xr[c] - mean;
        sq = fmaf(d, d, sq);
        const float z = xr[c] - (centre ? centre[c] : 0.f);
        zq = fmaf(z, z, zq);
    }
    const float var = wave_sum(sq) / (float)dim + eps;
    const float ratio = fabsf(mean) / sqrtf(var);
    const float zr = sqrtf(fmaxf(wave_sum(zq) / (float)dim / var - 1.0f, 0.f));
    if (lane == 0) {   // non-negative floats order like their bit patterns (NaN above all of them: a non-finite statistic wins)
        atomicMax(out, __float_as_uint(ratio));
        atomicMax(out + 1, __float_as_uint(centre ? zr : ratio));
    }
}

hipError_t launch_row_mean_ratio(const float* x, int ldx, int rows, int dim, float eps, const float* centre, float* out, hipStream_t s) {
    if (rows <= 0) return hipSuccess;
    hipLaunchKernelGGL(ivit_row_mean_ratio, dim3(ceil_div(rows, 4)), dim3(256), 0, s, x, ldx, rows, dim, eps, centre, reinterpret_cast<unsigned int*>(out));
    return hipGetLastError();
}

// Column means of [rows, dim] f32 rows -> out[dim] (the centre vector of a LayerNorm input; calibration only).  Deterministic: a block owns 64 columns,
// its four waves walk the rows r = w, w + 4, ... in order with double accumulators, and the four partial sums are added in wave order.
__global__ __launch_bounds__(256) void ivit_col_means(const float* __restrict__ x, int ldx, int rows, int dim, float* __restrict__ out) {
    __shared__ double part[4][64];
    const int col = blockIdx.x * 64 + (threadIdx.x & 63), w = threadIdx.x >> 6;
    double acc = 0.0;
    if (col < dim) for (int r = w; r < rows; r += 4) acc += (double)x[(size_t)r * ldx + col];
    part[w][threadIdx.x & 63] = acc;
    __syncthreads();
    if (w == 0 && col < dim) out[col] = (float)((((part[0][col & 63] + part[1][col & 63]) + part[2][col & 63]) + part[3][col & 63]) / (double)rows);
}

hipError_t launch_col_means(const float* x, int ldx, int rows, int dim, float* out, hipStream_t s) {
    if (rows <= 0 || dim <= 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(ivit_col_means, dim3(ceil_div(dim, 64)), dim3(256), 0, s, x, ldx, rows, dim, out);
    return hipGetLastError();
}

// d[n] = sum_k centre[k] W'[n][k] over the 16-bit matrix the MFMA multiplies (GemmParams::ln_d): hi + lo where the rows are pairs (split = 1: hi / lo
// interleaved per 64-column K-tile, engine.hip: Matrix).  One wave per row, double accumulation; calibration only.
__global__ __launch_bounds__(256) void ivit_centre_dot(const bf16_t* __restrict__ w, int ld, int rows, int cols, int split, const float* __restrict__ centre,
                                                       float* __restrict__ d, int f16) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const bf16_t* wr = w + (size_t)row * ld;
    double acc = 0.0;
    for (int k = lane; k < cols; k += 64) {
        double wv;
        if (split == 1) { const int i = (k >> 6) * 128 + (k & 63); wv = (double)dec16(f16, wr[i]) + (double)dec16(f16, wr[i + 64]); }
        else wv = (double)dec16(f16, wr[k]);
        acc += (double)centre[k] * wv;
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if (lane == 0) d[row] = (float)acc;
}

hipError_t launch_centre_dot(const bf16_t* w, int ld, int rows, int cols, int split, const float* centre, float* d, hipStream_t s, int f16) {
    if (rows <= 0 || cols <= 0 || (split != 0 && split != 1)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(ivit_centre_dot, dim3(ceil_div(rows, 4)), dim3(256), 0, s, w, ld, rows, cols, split, centre, d, f16);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------- gather / convert
__global__ void ivit_gather_rows(const float* __restrict__ in, int64_t row_stride, float* __restrict__ out, int rows, int dim) {
    const int d4 = dim >> 2;
    const int64_t total = (int64_t)rows * d4;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % d4);
        const int64_t r = i / d4;
        reinterpret_cast<float4*>(out + r * dim)[c] = reinterpret_cast<const float4*>(in + r * row_stride * dim)[c];
    }
}

hipError_t launch_gather_rows(const float* in, int64_t row_stride, float* out, int rows, int dim, hipStream_t s) {
    if (dim % 4) return hipErrorInvalidValue;
    hipLaunchKernelGGL(ivit_gather_rows, dim3(ew_grid((int64_t)rows * dim / 4)), dim3(EW_THREADS), 0, s, in, row_stride, out, rows, dim);
    return hipGetLastError();
}

__global__ void ivit_f32_to_bf16(const float* __restrict__ in, int ldi, bf16_t* __restrict__ out, int ldo, int rows, int cols, int f16) {
    const int c4 = ldo >> 2;   // output is written over the full padded width
    const int64_t total = (int64_t)rows * c4;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % c4) * 4;
        const int64_t r = i / c4;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (c + e < cols) ? in[r * ldi + c + e] : 0.f;
        u32x2 pk = {pack16x2(f16, v[0], v[1]), pack16x2(f16, v[2], v[3])};
        *reinterpret_cast<u32x2*>(out + r * ldo + c) = pk;
    }
}

hipError_t launch_f32_to_bf16(const float* in, int ldi, bf16_t* out, int ldo, int rows, int cols, hipStream_t s, int f16) {
    if (ldo % 4) return hipErrorInvalidValue;
    hipLaunchKernelGGL(ivit_f32_to_bf16, dim3(ew_grid((int64_t)rows * ldo / 4)), dim3(EW_THREADS), 0, s, in, ldi, out, ldo, rows, cols, f16);
    return hipGetLastError();
}

// [hi | lo] pair of an f32 activation matrix (split-operand head GEMM on a caller's f32 class-token features)
__global__ void ivit_f32_to_split16(const float* __restrict__ in, int ldi, bf16_t* __restrict__ out, int ldhalf, int rows, int cols, int f16) {
    const int64_t total = (int64_t)rows * ldhalf;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % ldhalf);
        const int64_t r = i / ldhalf;
        const float v = c < cols ? in[r * ldi + c] : 0.f;
        const bf16_t h = enc16(f16, v);
        out[r * 2 * ldhalf + c] = h;
        out[r * 2 * ldhalf + ldhalf + c] = enc16(f16, v - dec16(f16, h));
    }
}
hipError_t launch_f32_to_split16(const float* in, int ldi, bf16_t* out, int ldhalf, int rows, int cols, hipStream_t s, int f16) {
    hipLaunchKernelGGL(ivit_f32_to_split16, dim3(ew_grid((int64_t)rows * ldhalf)), dim3(EW_THREADS), 0, s, in, ldi, out, ldhalf, rows, cols, f16);
    return hipGetLastError();
}

// hi/lo pairs of a weight matrix from its f32 original (launch_split_weight, kernels.h).  One wave per weight row.
__global__ __launch_bounds__(256) void ivit_split_weight(const float* __restrict__ w, int ldw, int rows, int cols, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, const float* __restrict__ bias, bf16_t* __restrict__ out,
                                                         int ld_out, int kpad, int both, float* __restrict__ s_out, float* __restrict__ c_out, int f16) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* wr = w + (size_t)row * ldw;
    bf16_t* o = out + (size_t)row * ld_out;
    float s = 0.f, c = 0.f;
    for (int k = lane; k < kpad; k += 64) {
        bf16_t h = 0, l = 0;
        if (k < cols) {
            const float wv = wr[k];
            const float v = gamma ? wv * gamma[k] : wv;
            h = enc16(f16, v);
            l = enc16(f16, v - dec16(f16, h));
            s += dec16(f16, h) + dec16(f16, l);
            if (beta) c = fmaf(beta[k], wv, c);
        }
        if (both) { o[k] = h; o[kpad + k] = h; o[2 * kpad + k] = l; }
        else {   // weight-only split: hi / lo INTERLEAVED per 64-column K-tile, [hi t0 | lo t0 | hi t1 | lo t1 | ...] - product K-tiles 2t and 2t + 1
                 // multiply the SAME activation K-tile t (GemmParams::a_shift), which the GEMM then stages and reads once for both (round 4)
            const int tile = k >> 6, within = k & 63;
            o[tile * 128 + within] = h;
            o[tile * 128 + 64 + within] = l;
        }
    }
    s = wave_sum(s);
    c = wave_sum(c);
    if (lane == 0) {
        if (s_out) s_out[row] = s;
        if (c_out) c_out[row] = c + (bias ? bias[row] : 0.f);
    }
}
hipError_t launch_split_weight(const float* w, int ldw, int rows, int cols, const float* gamma, const float* beta, const float* bias,
                               bf16_t* out, int ld_out, int kpad, int both, float* s_out, float* c_out, hipStream_t s, int f16) {
    if (ld_out < (both ? 3 : 2) * kpad) return hipErrorInvalidValue;
    hipLaunchKernelGGL(ivit_split_weight, dim3(ceil_div(rows, 4)), dim3(256), 0, s, w, ldw, rows, cols, gamma, beta, bias, out, ld_out, kpad, both, s_out, c_out, f16);
    return hipGetLastError();
}

__global__ void ivit_bf16_to_f32(const bf16_t* __restrict__ in, int ldi, float* __restrict__ out, int rows, int cols, int f16) {
    const int64_t total = (int64_t)rows * cols;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % cols);
        const int64_t r = i / cols;
        out[i] = dec16(f16, in[r * ldi + c]);
    }
}

hipError_t launch_bf16_to_f32(const bf16_t* in, int ldi, float* out, int rows, int cols, hipStream_t s, int f16) {
    hipLaunchKernelGGL(ivit_bf16_to_f32, dim3(ew_grid((int64_t)rows * cols)), dim3(EW_THREADS), 0, s, in, ldi, out, rows, cols, f16);
    return hipGetLastError();
}


// ---------------------------------------------------------------------------- fp8 support
// max |x| over a bf16 [rows, cols] matrix (ld elements) -> atomicMax on the f32 bit pattern (values >= 0)
__global__ void ivit_amax_bf16(const bf16_t* __restrict__ in, int ld, int rows, int cols, unsigned int* __restrict__ out) {
    const int c8 = cols >> 3;
    const int64_t total = (int64_t)rows * c8;
    float m = 0.f;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % c8);
        const int64_t r = i / c8;
        const u32x4 v = *reinterpret_cast<const u32x4*>(in + r * ld + c * 8);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            m = fmaxf(m, fabsf(bf2f((bf16_t)(v[e] & 0xffffu))));
            m = fmaxf(m, fabsf(bf2f((bf16_t)(v[e] >> 16))));
        }
    }
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) atomicMax(out, __float_as_uint(m));
}

hipError_t launch_amax_bf16(const bf16_t* in, int ld, int rows, int cols, float* out, hipStream_t s) {
    if (cols % 8) return hipErrorInvalidValue;
    hipLaunchKernelGGL(ivit_amax_bf16, dim3(ew_grid((int64_t)rows * cols / 8)), dim3(EW_THREADS), 0, s, in, ld, rows, cols,
                       reinterpret_cast<unsigned int*>(out));
    return hipGetLastError();
}

// bf16 weight matrix [rows, cols] (ld) -> e4m3 [rows, ld8] with one scale per output row:
// rowscale[n] = max_k |W[n,k]| / 448 (1 if the row is all zero); padding columns are written as 0.
__global__ __launch_bounds__(256) void ivit_quantize_weight_fp8(const bf16_t* __restrict__ w, int ld, int rows, int cols,
                                                                unsigned char* __restrict__ w8, int ld8, float* __restrict__ rowscale) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const bf16_t* wr = w + (size_t)row * ld;
    float m = 0.f;
    for (int c = lane; c < cols; c += 64) m = fmaxf(m, fabsf(bf2f(wr[c])));
    m = wave_max(m);
    const float scale = (m > 0.f) ? m / FP8_MAX : 1.0f;
    const float inv = 1.0f / scale;
    if (lane == 0) rowscale[row] = scale;
    for (int c4 = lane; c4 < (ld8 >> 2); c4 += 64) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) { const int c = c4 * 4 + e; v[e] = (c < cols) ? bf2f(wr[c]) * inv : 0.f; }
        reinterpret_cast<unsigned int*>(w8 + (size_t)row * ld8)[c4] = pack_fp8x4(v[0], v[1], v[2], v[3]);
    }
}

hipError_t launch_quantize_weight_fp8(const bf16_t* w, int ld, int rows, int cols, unsigned char* w8, int ld8, float* rowscale,
                                      hipStream_t s) {
    if (ld8 % 4) return hipErrorInvalidValue;
    hipLaunchKernelGGL(ivit_quantize_weight_fp8, dim3(ceil_div(rows, 4)), dim3(256), 0, s, w, ld, rows, cols, w8, ld8, rowscale);
    return hipGetLastError();
}

// out[n] = a * in[n]
__global__ void ivit_scale_vec(const float* __restrict__ in, float a, float* __restrict__ out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = a * in[i];
}
hipError_t launch_scale_vec(const float* in, float a, float* out, int n, hipStream_t s) {
    hipLaunchKernelGGL(ivit_scale_vec, dim3(ceil_div(n, 256)), dim3(256), 0, s, in, a, out, n);
    return hipGetLastError();
}

}  // namespace ivit
