/* ORACLE - test infrastructure, not product code.
 *
 * Plain-C restatement of the integer bookkeeping of the ViT patch embedding ("unfold-as-GEMM"),
 * the part of the hot path BASELINE.json requires to be BIT-EXACT.  The reference has no unfold of
 * its own (its only model plugin wraps torchvision modules, static/models/vgg16.py:10-14); the
 * contract is SURVEY.md Appendix B:  P[n,k] = x[c, gy*p+ky, gx*p+kx],  n = gy*G+gx,
 * k = c*p*p + ky*p + kx  - i.e. torch.nn.Conv2d(3, D, p, stride=p) seen as a matrix product, which
 * tests/test_oracle.py checks against torch itself.
 *
 * Written with nested loops over (gy, gx, c, ky, kx) - deliberately a different formulation from the
 * engine's div/mod form (interactive_vit_amd/csrc/common.h: unfold_offset) so that the two can
 * disagree.  Only tests/ load this (ctypes).
 */
#include <stdint.h>

/* idx[n*K + k] = flat offset into a [3,S,S] image; returns Np*K */
int64_t oracle_unfold_index(int32_t image, int32_t patch, int64_t* idx) {
    const int32_t g = image / patch;
    const int64_t K = 3LL * patch * patch;
    for (int32_t gy = 0; gy < g; ++gy)
        for (int32_t gx = 0; gx < g; ++gx) {
            int64_t* row = idx + ((int64_t)gy * g + gx) * K;
            int64_t k = 0;
            for (int32_t c = 0; c < 3; ++c)
                for (int32_t ky = 0; ky < patch; ++ky)
                    for (int32_t kx = 0; kx < patch; ++kx)
                        row[k++] = (int64_t)c * image * image + (int64_t)(gy * patch + ky) * image + (gx * patch + kx);
        }
    return (int64_t)g * g * K;
}

/* out[b, n, k] = in[b, idx[n,k]]  (pure gather, no arithmetic on the values) */
void oracle_unfold_f32(const float* in, float* out, int32_t batch, int32_t image, int32_t patch) {
    const int32_t g = image / patch;
    const int64_t K = 3LL * patch * patch, img = 3LL * image * image;
    for (int32_t b = 0; b < batch; ++b)
        for (int32_t gy = 0; gy < g; ++gy)
            for (int32_t gx = 0; gx < g; ++gx) {
                float* row = out + ((int64_t)b * g * g + (int64_t)gy * g + gx) * K;
                int64_t k = 0;
                for (int32_t c = 0; c < 3; ++c)
                    for (int32_t ky = 0; ky < patch; ++ky)
                        for (int32_t kx = 0; kx < patch; ++kx)
                            row[k++] = in[b * img + (int64_t)c * image * image + (int64_t)(gy * patch + ky) * image + (gx * patch + kx)];
            }
}

/* row of (image b, token t) in the engine's [B*N, D] activation matrices, and the patch GEMM's
 * output row remap m = b*Np + n  ->  b*N + 1 + n  (class token first) */
int64_t oracle_token_row(int64_t m, int32_t patches) { return (m / patches) * (patches + 1) + 1 + (m % patches); }
