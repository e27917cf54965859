// Shared device/host helpers for the gfx950 (MI355X, CDNA4) kernels.  wave = 64 lanes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ivit {

typedef uint16_t bf16_t;  // raw bf16 bits
typedef __attribute__((ext_vector_type(8))) short bf16x8;   // one MFMA A/B fragment (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;    // one 16x16 MFMA accumulator fragment
typedef __attribute__((ext_vector_type(16))) float f32x16;  // one 32x32 MFMA accumulator fragment
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

#define IVIT_LDS __attribute__((address_space(3)))
#define IVIT_GLOBAL __attribute__((address_space(1)))

constexpr int WAVE = 64;

__host__ __device__ inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
__host__ __device__ inline int round_up(int a, int b) { return ceil_div(a, b) * b; }

// f32 -> bf16, round to nearest even; a plain cast lowers to v_cvt_pk_bf16_f32 on gfx950 and keeps
// NaN a NaN (MI355X_MICROARCH.md, correctness boundaries).
__device__ __forceinline__ bf16_t f2bf(float x) {
    __bf16 b = (__bf16)x;
    return __builtin_bit_cast(bf16_t, b);
}
__device__ __forceinline__ float bf2f(bf16_t b) {
    return __builtin_bit_cast(float, ((unsigned int)b) << 16);
}
// two f32 -> packed bf16x2 in one dword (lo in bits 0..15)
__device__ __forceinline__ unsigned int pack_bf16x2(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
    typedef __attribute__((ext_vector_type(2))) float f2;
    f2 v = {lo, hi};
    bf2 r = __builtin_convertvector(v, bf2);
    return __builtin_bit_cast(unsigned int, r);
}

// two f32 -> packed IEEE f16x2 in one dword, round to nearest even (IVIT_PRECISION_F16 data path)
__device__ __forceinline__ unsigned int pack_f16x2(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) _Float16 h2;
    typedef __attribute__((ext_vector_type(2))) float f2;
    f2 v = {lo, hi};
    h2 r = __builtin_convertvector(v, h2);
    return __builtin_bit_cast(unsigned int, r);
}

// The 16-bit operand type of the data path (IVIT_PRECISION_BF16 / IVIT_PRECISION_F16): the MFMA it is multiplied by, the
// f32 -> 16-bit rounding (nearest even) and back.  gfx950 issues v_mfma_f32_16x16x32_f16 at the rate of the bf16 form, so
// the f16 data path costs the same time and carries 11 significant bits instead of 8 (1e-3 against a PLAIN f32 forward).
struct OpBf16 {
    static constexpr bool F16 = false;
    static __device__ __forceinline__ f32x4 mfma(const bf16x8& w, const bf16x8& a, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, a, c, 0, 0, 0); }
    static __device__ __forceinline__ f32x16 mfma32(const bf16x8& w, const bf16x8& a, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(w, a, c, 0, 0, 0); }
    static __device__ __forceinline__ unsigned int pack2(float lo, float hi) { return pack_bf16x2(lo, hi); }
    // acc + the two 16-bit values of a packed pair (v_dot2c_f32_bf16 against 1.0 | 1.0)
    static __device__ __forceinline__ float add_pair(unsigned int pk, float acc) {
        typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
        return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, pk), __builtin_bit_cast(bf2, 0x3f803f80u), acc, false);
    }
    static __device__ __forceinline__ bf16_t from_f32(float x) { return f2bf(x); }
    static __device__ __forceinline__ float to_f32(bf16_t b) { return bf2f(b); }
};
struct OpF16 {
    static constexpr bool F16 = true;
    typedef __attribute__((ext_vector_type(8))) _Float16 h8;
    static __device__ __forceinline__ f32x4 mfma(const bf16x8& w, const bf16x8& a, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, w), __builtin_bit_cast(h8, a), c, 0, 0, 0);
    }
    static __device__ __forceinline__ f32x16 mfma32(const bf16x8& w, const bf16x8& a, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, w), __builtin_bit_cast(h8, a), c, 0, 0, 0);
    }
    static __device__ __forceinline__ unsigned int pack2(float lo, float hi) { return pack_f16x2(lo, hi); }
    static __device__ __forceinline__ float add_pair(unsigned int pk, float acc) {
        typedef __attribute__((ext_vector_type(2))) _Float16 h2;
        return __builtin_amdgcn_fdot2(__builtin_bit_cast(h2, pk), __builtin_bit_cast(h2, 0x3c003c00u), acc, false);
    }
    static __device__ __forceinline__ bf16_t from_f32(float x) { return __builtin_bit_cast(bf16_t, (_Float16)x); }
    static __device__ __forceinline__ float to_f32(bf16_t b) { return (float)__builtin_bit_cast(_Float16, b); }
};

// fp8 (OCP e4m3fn on gfx950): two f32 -> two fp8 bytes, round to nearest even, SATURATING at +-448
// (clamped here in software: torch's cast gives NaN above 448 and the oracle clamps the same way).
constexpr float FP8_MAX = 448.0f;
__device__ __forceinline__ unsigned int pack_fp8x4(float a, float b, float c, float d) {
    a = fminf(fmaxf(a, -FP8_MAX), FP8_MAX); b = fminf(fmaxf(b, -FP8_MAX), FP8_MAX);
    c = fminf(fmaxf(c, -FP8_MAX), FP8_MAX); d = fminf(fmaxf(d, -FP8_MAX), FP8_MAX);
    int p = 0;
    p = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, p, false);   // bytes 0,1
    p = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, p, true);    // bytes 2,3
    return (unsigned int)p;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}

// erf-form GELU (torch.nn.GELU() default), 0.5 x (1 + erf(x / sqrt 2)), with erf by Abramowitz & Stegun 7.1.26
// (1 - (a1 t + ... + a5 t^5) exp(-z^2), t = 1 / (1 + p |z|); |error| <= 1.5e-7), written for the GEMM epilogues, where
// VALU issue slots are what the interleaved MFMA loop runs out of first:
//     gelu(x) = max(x, 0) - |x| P(t) exp(-x^2 / 2),    t = 1 / (1 + (p / sqrt 2) |x|),   P = (a1 t + ... + a5 t^5) / 2
// 13 VALU instructions (one v_rcp_f32, one v_exp_f32) instead of the 16 of "0.5 x (1 + copysign(erf(|z|), z))";
// <= 3.4e-7 absolute from the f64 GELU over |x| <= 9 (three orders below the bf16 rounding of the output it feeds).
// The branchy libm erff cost 22 % of the MLP-up GEMM when used in its epilogue.
__device__ __forceinline__ float gelu_erf(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.23164188826636045f, ax, 1.0f));
    const float w = x * 0.8493218002880191f;                       // exp(-x^2 / 2) = exp2(-(x sqrt(log2(e) / 2))^2)
    const float e = __builtin_amdgcn_exp2f(-(w * w));
    float poly = fmaf(0.5307027145f, t, -0.7265760135f);
    poly = fmaf(poly, t, 0.7107068705f);
    poly = fmaf(poly, t, -0.142248368f);
    poly = fmaf(poly, t, 0.127414796f);
    const float q = poly * (t * ax);
    return fmaf(-q, e, fmaxf(x, 0.0f));
}

// The unfold bookkeeping, shared by host (ivit_unfold_offset) and device (unfold kernel):
// patch n = gy*G + gx, column k = c*p*p + ky*p + kx  ->  flat offset in a [3,S,S] image.
__host__ __device__ inline int64_t unfold_offset(int image, int patch, int n, int k) {
    const int g = image / patch;
    const int gy = n / g, gx = n - gy * g;
    const int pp = patch * patch;
    const int c = k / pp, r = k - c * pp;
    const int ky = r / patch, kx = r - ky * patch;
    return (int64_t)c * image * image + (int64_t)(gy * patch + ky) * image + (gx * patch + kx);
}

}  // namespace ivit
