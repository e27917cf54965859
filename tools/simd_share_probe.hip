// Probe: what two waves on ONE SIMD share.  A 512-thread workgroup per CU (waves w and w + 4 sit on the same SIMD); waves 0-3 run role A,
// waves 4-7 role B, both start behind one barrier and each reports its own shader-clock cycles per loop iteration.
// Roles: 0 idle | 1 chain of dependent v_mfma_f32_32x32x16_bf16 (one accumulator) | 2 two accumulators alternating | 3 v_fma_f32 stream |
//        4 v_exp_f32 stream | 5 dependent chain with 8 v_fma_f32 between MFMAs | 6 dependent chain with 16 between | 7 four accumulators
//        8 dependent 16x16x32 chain | 9 independent 16x16x32 (4 accumulators)
// One loop iteration = 8 MFMAs (roles 1, 2, 5, 6, 7, 8, 9) or 64 VALU instructions (roles 3, 4).  Measurement aid only.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

#define FMA8(v) asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n" \
                             "v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n" \
                             : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]))
#define EXP8(v) asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n" \
                             "v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n" \
                             : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]))
#define M32(acc) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(fa), "v"(fb))
#define M16(acc) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(fa), "v"(fb))

template <int ROLE>
__device__ __forceinline__ void role_loop(int iters, float* sink) {
    bf16x8_t fa, fb;
    for (int i = 0; i < 8; ++i) { fa[i] = (__bf16)(0.01f * (threadIdx.x % 7)); fb[i] = (__bf16)(0.02f * (threadIdx.x % 5)); }
    bf16x8_t ra[4], rb[4];
    {
        unsigned h = (blockIdx.x * 1024u + threadIdx.x) * 2654435761u + 12345u;
        for (int s = 0; s < 4; ++s)
            for (int i = 0; i < 8; ++i) {
                h = h * 1664525u + 1013904223u; ra[s][i] = (__bf16)((float)(int)(h >> 8) * (1.0f / 8388608.0f) - 1.0f);
                h = h * 1664525u + 1013904223u; rb[s][i] = (__bf16)((float)(int)(h >> 8) * (1.0f / 8388608.0f) - 1.0f);
            }
    }
    f32x16 a0 = {}, a1 = {}, a2 = {}, a3 = {};
    f32x4 c0 = {}, c1 = {}, c2 = {}, c3 = {};
    float v[4] = {0.5f, 0.25f, 0.125f, 0.0625f};
    float e[4] = {0.1f, 0.2f, 0.3f, 0.4f}, tmx = 0.f;
    unsigned pkr = 0, pq[4] = {0, 0, 0, 0};
    typedef __attribute__((ext_vector_type(2))) float f2;
    f2 pd[4], ps[4];
    for (int i = 0; i < 4; ++i) { ps[i] = f2{0.5f + i, 0.25f * i}; pd[i] = f2{0.f, 0.f}; }
    typedef __attribute__((ext_vector_type(2))) unsigned u2; typedef __attribute__((ext_vector_type(4))) unsigned u4;
    u2 l2[8]; u4 l4[4];
    const unsigned laddr = (threadIdx.x & 63) * 16;
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
        if (ROLE == 1) { M32(a0); M32(a0); M32(a0); M32(a0); M32(a0); M32(a0); M32(a0); M32(a0); }
        if (ROLE == 2) { M32(a0); M32(a1); M32(a0); M32(a1); M32(a0); M32(a1); M32(a0); M32(a1); }
        if (ROLE == 7) { M32(a0); M32(a1); M32(a2); M32(a3); M32(a0); M32(a1); M32(a2); M32(a3); }
        if (ROLE == 3) { FMA8(v); FMA8(v); FMA8(v); FMA8(v); FMA8(v); FMA8(v); FMA8(v); FMA8(v); }
        if (ROLE == 4) { EXP8(v); EXP8(v); EXP8(v); EXP8(v); EXP8(v); EXP8(v); EXP8(v); EXP8(v); }
        if (ROLE == 5) { M32(a0); FMA8(v); M32(a0); FMA8(v); M32(a0); FMA8(v); M32(a0); FMA8(v); M32(a0); FMA8(v); M32(a0); FMA8(v); M32(a0); FMA8(v); M32(a0); FMA8(v); }
        if (ROLE == 6) { M32(a0); FMA8(v); FMA8(v); M32(a0); FMA8(v); FMA8(v); M32(a0); FMA8(v); FMA8(v); M32(a0); FMA8(v); FMA8(v);
                         M32(a0); FMA8(v); FMA8(v); M32(a0); FMA8(v); FMA8(v); M32(a0); FMA8(v); FMA8(v); M32(a0); FMA8(v); FMA8(v); }
        if (ROLE >= 20 && ROLE <= 26) {   // the attention step's mix: 16 x (fma, exp), 8 cvt_pk, 8 dot2c, 8 max3 [+ 8 MFMAs interleaved: 21+] [+ 12 LDS reads: 22+] [23: MFMAs in one burst]
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                if (ROLE == 21 || ROLE == 22) { if (g & 1) M32(a1); else M32(a0); }
                if (ROLE == 26) { if (g & 1) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(a1) : "v"(ra[g & 3]), "v"(rb[(g >> 1) & 3]));
                                  else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(a0) : "v"(ra[g & 3]), "v"(rb[(g >> 1) & 3])); }
                if (ROLE == 23 && g == 0) { M32(a0); M32(a1); M32(a0); M32(a1); M32(a0); M32(a1); M32(a0); M32(a1); }
                if ((ROLE == 22 || ROLE == 26) && g == 0) asm volatile("ds_read_b64_tr_b16 %0, %8\n ds_read_b64_tr_b16 %1, %8 offset:1024\n ds_read_b64_tr_b16 %2, %8 offset:2048\n ds_read_b64_tr_b16 %3, %8 offset:3072\n"
                                                       "ds_read_b64_tr_b16 %4, %8 offset:4096\n ds_read_b64_tr_b16 %5, %8 offset:5120\n ds_read_b64_tr_b16 %6, %8 offset:6144\n ds_read_b64_tr_b16 %7, %8 offset:7168\n"
                                                       : "=v"(l2[0]), "=v"(l2[1]), "=v"(l2[2]), "=v"(l2[3]), "=v"(l2[4]), "=v"(l2[5]), "=v"(l2[6]), "=v"(l2[7]) : "v"(laddr));
                if ((ROLE == 22 || ROLE == 26) && g == 4) asm volatile("ds_read_b128 %0, %4 offset:8192\n ds_read_b128 %1, %4 offset:9216\n ds_read_b128 %2, %4 offset:10240\n ds_read_b128 %3, %4 offset:11264\n"
                                                       : "=v"(l4[0]), "=v"(l4[1]), "=v"(l4[2]), "=v"(l4[3]) : "v"(laddr));
                asm volatile("v_fma_f32 %0, %4, %5, %6\n v_fma_f32 %1, %4, %5, %6\n v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n"
                             "v_cvt_pk_bf16_f32 %2, %7, %8\n v_dot2c_f32_bf16 %3, 0x3f803f80, %2\n v_max3_f32 %9, %9, %4, %5\n"
                             : "+v"(e[2 * (g & 1)]), "+v"(e[2 * (g & 1) + 1]), "=v"(pkr), "+v"(v[3]), "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(e[2 * ((g + 1) & 1)]), "+v"(e[2 * ((g + 1) & 1) + 1]), "+v"(tmx));
            }
            if (ROLE == 22 || ROLE == 26) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        if (ROLE == 30) { for (int g = 0; g < 16; ++g) asm volatile("v_cvt_pk_bf16_f32 %0, %4, %5\n v_cvt_pk_bf16_f32 %1, %5, %6\n v_cvt_pk_bf16_f32 %2, %6, %7\n v_cvt_pk_bf16_f32 %3, %7, %4\n"
                                                                   : "=v"(pq[0]), "=v"(pq[1]), "=v"(pq[2]), "=v"(pq[3]) : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3])); }
        if (ROLE == 31) { for (int g = 0; g < 16; ++g) asm volatile("v_dot2c_f32_bf16 %0, 0x3f803f80, %4\n v_dot2c_f32_bf16 %1, 0x3f803f80, %4\n v_dot2c_f32_bf16 %2, 0x3f803f80, %4\n v_dot2c_f32_bf16 %3, 0x3f803f80, %4\n"
                                                                   : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]) : "v"(pkr)); }
        if (ROLE == 32) { for (int g = 0; g < 16; ++g) asm volatile("v_max3_f32 %0, %0, %4, %5\n v_max3_f32 %1, %1, %5, %6\n v_max3_f32 %2, %2, %6, %7\n v_max3_f32 %3, %3, %7, %4\n"
                                                                   : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]) : "v"(e[0]), "v"(e[1]), "v"(e[2]), "v"(e[3])); }
        if (ROLE == 33) {   // v_fma_f32 with three DISTINCT source registers and a fourth destination (64 per iteration)
            for (int g = 0; g < 16; ++g) asm volatile("v_fma_f32 %0, %4, %5, %6\n v_fma_f32 %1, %5, %6, %7\n v_fma_f32 %2, %6, %7, %4\n v_fma_f32 %3, %7, %4, %5\n"
                                                       : "=v"(e[0]), "=v"(e[1]), "=v"(e[2]), "=v"(e[3]) : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]));
        }
        if (ROLE == 34) {   // 8 MFMAs over four accumulators AND rotating operand fragments (the kernel's register traffic: 24 source + 16 result registers each)
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %4, %5, %0\n v_mfma_f32_32x32x16_bf16 %1, %6, %7, %1\n v_mfma_f32_32x32x16_bf16 %2, %5, %6, %2\n v_mfma_f32_32x32x16_bf16 %3, %7, %4, %3\n"
                         "v_mfma_f32_32x32x16_bf16 %0, %6, %4, %0\n v_mfma_f32_32x32x16_bf16 %1, %7, %5, %1\n v_mfma_f32_32x32x16_bf16 %2, %4, %7, %2\n v_mfma_f32_32x32x16_bf16 %3, %5, %6, %3\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(ra[0]), "v"(rb[0]), "v"(ra[1]), "v"(rb[1]));
        }
#define MR(acc, A, B) "v_mfma_f32_32x32x16_bf16 %" #acc ", %" #A ", %" #B ", %" #acc "\n"
#define MS(acc, A, B) "v_mfma_f32_16x16x32_bf16 %" #acc ", %" #A ", %" #B ", %" #acc "\n"
        // operands: %0..%3 accumulators, %4 %6 = ra[0] ra[1], %5 %7 = rb[0] rb[1], %8 %9 = ra[2] rb[2]
        if (ROLE == 35) asm volatile(MR(0, 4, 5) MR(1, 6, 5) MR(2, 8, 5) MR(3, 4, 5) MR(0, 6, 5) MR(1, 8, 5) MR(2, 4, 5) MR(3, 6, 5)   // A rotates, B constant
                                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(ra[0]), "v"(rb[0]), "v"(ra[1]), "v"(rb[1]), "v"(ra[2]), "v"(rb[2]));
        if (ROLE == 36) asm volatile(MR(0, 4, 5) MR(1, 4, 7) MR(2, 4, 9) MR(3, 4, 5) MR(0, 4, 7) MR(1, 4, 9) MR(2, 4, 5) MR(3, 4, 7)   // A constant, B rotates
                                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(ra[0]), "v"(rb[0]), "v"(ra[1]), "v"(rb[1]), "v"(ra[2]), "v"(rb[2]));
        if (ROLE == 37) asm volatile(MS(0, 4, 5) MS(1, 6, 7) MS(2, 8, 9) MS(3, 4, 7) MS(0, 6, 9) MS(1, 8, 5) MS(2, 4, 9) MS(3, 6, 5)   // 16x16x32, both rotate
                                     : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(ra[0]), "v"(rb[0]), "v"(ra[1]), "v"(rb[1]), "v"(ra[2]), "v"(rb[2]));
        if (ROLE == 38) asm volatile(MR(0, 4, 5) MR(1, 4, 5) MR(2, 6, 7) MR(3, 6, 7) MR(0, 8, 9) MR(1, 8, 9) MR(2, 4, 7) MR(3, 4, 7)   // both rotate, every pair used twice in a row
                                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(ra[0]), "v"(rb[0]), "v"(ra[1]), "v"(rb[1]), "v"(ra[2]), "v"(rb[2]));
        if (ROLE == 39) asm volatile(MR(0, 4, 5) MR(1, 6, 5) MR(2, 8, 7) MR(3, 4, 7) MR(0, 6, 9) MR(1, 8, 9) MR(2, 4, 5) MR(3, 6, 5)   // A rotates, B changes every second MFMA (P.V: one P, two V^T)
                                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(ra[0]), "v"(rb[0]), "v"(ra[1]), "v"(rb[1]), "v"(ra[2]), "v"(rb[2]));
        if (ROLE == 40) asm volatile(MR(0, 4, 5) MR(0, 6, 7) MR(0, 8, 9) MR(0, 4, 7) MR(1, 6, 9) MR(1, 8, 5) MR(1, 4, 9) MR(1, 6, 5)   // both rotate, chains of four on one accumulator (QK^T)
                                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(ra[0]), "v"(rb[0]), "v"(ra[1]), "v"(rb[1]), "v"(ra[2]), "v"(rb[2]));
        if (ROLE == 43) {   // 64 v_pk_fma_f32 (two f32 lanes each), distinct sources
            for (int g = 0; g < 16; ++g) asm volatile("v_pk_fma_f32 %0, %4, %5, %6\n v_pk_fma_f32 %1, %5, %6, %7\n v_pk_fma_f32 %2, %6, %7, %4\n v_pk_fma_f32 %3, %7, %4, %5\n"
                                                       : "=v"(pd[0]), "=v"(pd[1]), "=v"(pd[2]), "=v"(pd[3]) : "v"(ps[0]), "v"(ps[1]), "v"(ps[2]), "v"(ps[3]));
        }
        if (ROLE == 44) {   // 64 v_pk_mul_f32
            for (int g = 0; g < 16; ++g) asm volatile("v_pk_mul_f32 %0, %4, %5\n v_pk_mul_f32 %1, %5, %6\n v_pk_mul_f32 %2, %6, %7\n v_pk_mul_f32 %3, %7, %4\n"
                                                       : "=v"(pd[0]), "=v"(pd[1]), "=v"(pd[2]), "=v"(pd[3]) : "v"(ps[0]), "v"(ps[1]), "v"(ps[2]), "v"(ps[3]));
        }
        if (ROLE == 41) {   // 512 v_fma_f32 in a straight line (4 KiB of code per iteration)
#pragma unroll
            for (int g = 0; g < 64; ++g) FMA8(v);
        }
        if (ROLE == 42) {   // 2048 v_fma_f32 in a straight line (16 KiB per iteration)
#pragma unroll
            for (int g = 0; g < 256; ++g) FMA8(v);
        }
        if (ROLE == 8) { M16(c0); M16(c0); M16(c0); M16(c0); M16(c0); M16(c0); M16(c0); M16(c0); }
        if (ROLE == 9) { M16(c0); M16(c1); M16(c2); M16(c3); M16(c0); M16(c1); M16(c2); M16(c3); }
    }
    if (ROLE == 26) { v[0] = (float)ra[0][0]; v[1] = (float)rb[1][3]; v[2] = (float)ra[2][5]; }
    if (ROLE == 22 || ROLE == 26) v[0] += __builtin_bit_cast(float, l2[0][0] ^ l2[7][1] ^ l4[0][0] ^ l4[3][3]);
    v[0] += __builtin_bit_cast(float, pq[0] ^ pq[1] ^ pq[2] ^ pq[3]) + pd[0][0] + pd[1][1] + pd[2][0] + pd[3][1];
    float s = e[0] + e[1] + e[2] + e[3] + tmx + __builtin_bit_cast(float, pkr) + a0[0] + a1[3] + a2[5] + a3[7] + c0[0] + c1[1] + c2[2] + c3[3] + v[0] + v[1] + v[2] + v[3];
    if (s == 123.456f) sink[0] = s;
}

template <int RA, int RB>
__global__ __launch_bounds__(512) void probe(int iters, float* sink, unsigned long long* clocks) {
    __shared__ char lds_pad[16384];
    if (iters < 0) lds_pad[threadIdx.x] = 1;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    __syncthreads();
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    if (wave < 4) { if (RA) role_loop<RA>(iters, sink); }
    else { if (RB) role_loop<RB>(iters, sink); }
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) clocks[wave] = t1 - t0;
}

template <int R0, int R1, int R2, int R3>
__global__ __launch_bounds__(1024) void probe4(int iters, float* sink, unsigned long long* clocks) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    __syncthreads();
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    const int grp = wave >> 2;
    if (grp == 0) { if (R0) role_loop<R0>(iters, sink); }
    else if (grp == 1) { if (R1) role_loop<R1>(iters, sink); }
    else if (grp == 2) { if (R2) role_loop<R2>(iters, sink); }
    else { if (R3) role_loop<R3>(iters, sink); }
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) clocks[wave] = t1 - t0;
}

static const char* names[35] = {"idle", "dep 32x32x16 chain", "2 accumulators", "v_fma stream", "v_exp stream", "dep chain + 8 fma / MFMA", "dep chain + 16 fma / MFMA",
                                "4 accumulators", "dep 16x16x32 chain", "4-acc 16x16x32", "", "", "", "", "", "", "", "", "", "",
                                "step VALU mix (56)", "step mix + 8 MFMA", "step mix + 8 MFMA + 12 LDS", "step mix, MFMAs in a burst", "", "", "mix + MFMA + LDS, random data", "", "", "", "", "", "", "v_fma, 3 distinct sources", "MFMA, 4 acc, rotating operands"};
template <int RA, int RB>
void run(float* sink, unsigned long long* dclk) {
    const int iters = 4000;
    hipLaunchKernelGGL((probe<RA, RB>), dim3(256), dim3(512), 0, 0, 50, sink, dclk); CK(hipDeviceSynchronize());
    CK(hipMemset(dclk, 0, 64));
    hipLaunchKernelGGL((probe<RA, RB>), dim3(256), dim3(512), 0, 0, iters, sink, dclk); CK(hipDeviceSynchronize());
    unsigned long long hc[8]; CK(hipMemcpy(hc, dclk, 64, hipMemcpyDeviceToHost));
    printf("A = %-26s | B = %-26s : A %7.1f cycles / iteration, B %7.1f\n", names[RA], names[RB], (double)hc[0] / iters, (double)hc[4] / iters);
}

template <int R0, int R1, int R2, int R3>
void run4(const char* label, float* sink, unsigned long long* dclk) {
    const int iters = 4000;
    hipLaunchKernelGGL((probe4<R0, R1, R2, R3>), dim3(256), dim3(1024), 0, 0, 50, sink, dclk); CK(hipDeviceSynchronize());
    CK(hipMemset(dclk, 0, 128));
    hipLaunchKernelGGL((probe4<R0, R1, R2, R3>), dim3(256), dim3(1024), 0, 0, iters, sink, dclk); CK(hipDeviceSynchronize());
    unsigned long long hc[16]; CK(hipMemcpy(hc, dclk, 128, hipMemcpyDeviceToHost));
    printf("%-64s: cycles / iteration of the four waves of SIMD 0: %7.1f %7.1f %7.1f %7.1f\n", label, (double)hc[0] / iters, (double)hc[4] / iters, (double)hc[8] / iters, (double)hc[12] / iters);
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    float* sink; unsigned long long* dclk;
    CK(hipMalloc(&sink, 4)); CK(hipMalloc(&dclk, 128));
    printf("iteration = 8 MFMAs [+ 64 / 128 v_fma] or 64 VALU instructions; cycles of s_memtime\n");
    run<1, 0>(sink, dclk); run<2, 0>(sink, dclk); run<7, 0>(sink, dclk); run<8, 0>(sink, dclk); run<9, 0>(sink, dclk); run<3, 0>(sink, dclk); run<4, 0>(sink, dclk);
    run<5, 0>(sink, dclk); run<6, 0>(sink, dclk);
    run<1, 1>(sink, dclk); run<7, 7>(sink, dclk); run<3, 3>(sink, dclk); run<1, 3>(sink, dclk); run<2, 3>(sink, dclk); run<7, 3>(sink, dclk); run<1, 4>(sink, dclk);
    run<7, 4>(sink, dclk); run<5, 5>(sink, dclk); run<6, 6>(sink, dclk); run<5, 3>(sink, dclk); run<8, 3>(sink, dclk); run<9, 3>(sink, dclk);
    run<20, 0>(sink, dclk); run<21, 0>(sink, dclk); run<22, 0>(sink, dclk); run<23, 0>(sink, dclk);
    run<26, 0>(sink, dclk); run<26, 26>(sink, dclk);
    run<20, 20>(sink, dclk); run<21, 21>(sink, dclk); run<22, 22>(sink, dclk); run<23, 23>(sink, dclk);
    printf("four waves per SIMD (64 instructions of the stream, or one step = 56 VALU [+ 8 MFMA + 12 LDS], per iteration)\n");
    run4<3, 0, 0, 0>("v_fma x1", sink, dclk); run4<3, 3, 0, 0>("v_fma x2", sink, dclk); run4<3, 3, 3, 3>("v_fma x4", sink, dclk);
    run4<4, 0, 0, 0>("v_exp x1", sink, dclk); run4<4, 4, 0, 0>("v_exp x2", sink, dclk); run4<4, 4, 4, 4>("v_exp x4", sink, dclk);
    run4<30, 0, 0, 0>("v_cvt_pk_bf16_f32 x1", sink, dclk); run4<30, 30, 0, 0>("v_cvt_pk_bf16_f32 x2", sink, dclk); run4<30, 30, 30, 30>("v_cvt_pk_bf16_f32 x4", sink, dclk);
    run4<31, 0, 0, 0>("v_dot2c_f32_bf16 x1", sink, dclk); run4<31, 31, 0, 0>("v_dot2c_f32_bf16 x2", sink, dclk); run4<31, 31, 31, 31>("v_dot2c_f32_bf16 x4", sink, dclk);
    run4<32, 0, 0, 0>("v_max3_f32 x1", sink, dclk); run4<32, 32, 0, 0>("v_max3_f32 x2", sink, dclk); run4<32, 32, 32, 32>("v_max3_f32 x4", sink, dclk);
    run4<1, 1, 1, 1>("dependent MFMA chain x4", sink, dclk);
    run4<1, 4, 0, 0>("MFMA chain + v_exp", sink, dclk); run4<1, 4, 4, 4>("MFMA chain + 3 x v_exp", sink, dclk); run4<1, 3, 3, 3>("MFMA chain + 3 x v_fma", sink, dclk);
    run4<20, 20, 20, 20>("step VALU mix x4", sink, dclk); run4<21, 21, 21, 21>("step mix + 8 MFMA x4", sink, dclk); run4<26, 26, 26, 26>("step mix + 8 MFMA + 12 LDS x4", sink, dclk);
    run4<23, 23, 23, 23>("step mix, MFMAs in a burst x4", sink, dclk);
    run4<33, 0, 0, 0>("v_fma (3 distinct sources) x1", sink, dclk); run4<33, 33, 33, 33>("v_fma (3 distinct sources) x4", sink, dclk);
    run4<34, 0, 0, 0>("MFMA 4 acc, rotating operands x1", sink, dclk); run4<34, 34, 0, 0>("MFMA rotating x2", sink, dclk);
    run4<34, 33, 33, 33>("MFMA rotating + 3 x v_fma distinct", sink, dclk); run4<1, 33, 33, 33>("MFMA chain + 3 x v_fma distinct", sink, dclk);
    run4<34, 4, 4, 4>("MFMA rotating + 3 x v_exp", sink, dclk); run4<34, 30, 30, 30>("MFMA rotating + 3 x v_cvt_pk", sink, dclk);
    run4<34, 34, 33, 33>("2 x MFMA rotating + 2 x v_fma distinct", sink, dclk);
    run4<43, 0, 0, 0>("v_pk_fma_f32 x1", sink, dclk); run4<43, 43, 43, 43>("v_pk_fma_f32 x4", sink, dclk); run4<1, 43, 43, 43>("MFMA chain + 3 x v_pk_fma_f32", sink, dclk);
    run4<44, 0, 0, 0>("v_pk_mul_f32 x1", sink, dclk); run4<44, 44, 44, 44>("v_pk_mul_f32 x4", sink, dclk);
    run4<41, 0, 0, 0>("512 v_fma straight line x1 (cycles per 64: divide by 8)", sink, dclk); run4<41, 41, 41, 41>("512 v_fma straight line x4 (divide by 8)", sink, dclk);
    run4<42, 0, 0, 0>("2048 v_fma straight line x1 (divide by 32)", sink, dclk); run4<42, 42, 42, 42>("2048 v_fma straight line x4 (divide by 32)", sink, dclk);
    run4<35, 0, 0, 0>("32x32x16: A rotates, B constant", sink, dclk); run4<36, 0, 0, 0>("32x32x16: A constant, B rotates", sink, dclk);
    run4<37, 0, 0, 0>("16x16x32: A and B rotate", sink, dclk); run4<38, 0, 0, 0>("32x32x16: A and B rotate, each pair twice in a row", sink, dclk);
    run4<39, 0, 0, 0>("32x32x16: A rotates, B every second", sink, dclk); run4<40, 0, 0, 0>("32x32x16: both rotate, chains of 4 on one accumulator", sink, dclk);
    run4<35, 35, 0, 0>("A rotates x2", sink, dclk); run4<37, 37, 0, 0>("16x16x32 both rotate x2", sink, dclk);
    return 0;
}
