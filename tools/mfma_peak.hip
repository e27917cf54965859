// What the matrix pipes sustain with nothing else in the way: every wave issues independent
// v_mfma_f32_16x16x32_bf16 back to back from registers (no LDS, no memory), W waves per SIMD, all 256 CUs,
// for long enough (~ms) that the clock settles under the power limit.  Reports TFLOP/s and the shader
// clock seen by s_memtime against the 100 MHz s_memrealtime.  Measurement aid.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int NACC, int RANDOM>
__global__ __launch_bounds__(512) void mfma_loop(int iters, float* sink, unsigned long long* clocks) {
    // RANDOM = 1: eight operand sets of pseudo-random bf16 in [-1, 1) used in turn (the toggling a real
    // GEMM produces); RANDOM = 0: one smooth operand pair (little switching activity)
    bf16x8_t av[8], bv[8];
    unsigned h = (blockIdx.x * 1024u + threadIdx.x) * 2654435761u + 12345u;
    for (int s = 0; s < 8; ++s)
        for (int i = 0; i < 8; ++i) {
            h = h * 1664525u + 1013904223u; const float ra = (float)(int)(h >> 8) * (1.0f / 8388608.0f) - 1.0f;
            h = h * 1664525u + 1013904223u; const float rb = (float)(int)(h >> 8) * (1.0f / 8388608.0f) - 1.0f;
            av[s][i] = RANDOM ? (__bf16)ra : (__bf16)(0.001f * (threadIdx.x + i));
            bv[s][i] = RANDOM ? (__bf16)rb : (__bf16)(0.002f * (threadIdx.x - i));
        }
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    unsigned long long c0, r0, c1, r1;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c0), "=s"(r0)::"memory");
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[i & 7], bv[(i >> 1) & 7], acc[i], 0, 0, 0);
    }
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c1), "=s"(r1)::"memory");
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 123.456f) sink[0] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clocks[0] = c1 - c0; clocks[1] = r1 - r0; }
}

int main() {
    float* sink; unsigned long long* dclk;
    CK(hipMalloc(&sink, 4)); CK(hipMalloc(&dclk, 16));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    constexpr int NACC = 16;
    // differential timing: (flops(N2) - flops(N1)) / (t(N2) - t(N1)) cancels launch + operand set-up time and
    // does not depend on how many of the blocks are resident at once
    for (int rnd = 0; rnd < 2; ++rnd)
    for (int waves_per_simd : {1, 2}) {
        const int threads = 256 * waves_per_simd, blocks = 256;
        auto kern = rnd ? mfma_loop<NACC, 1> : mfma_loop<NACC, 0>;
        const int n1 = 20000, n2 = 120000;
        float t[2]; double clk_mhz = 0;
        for (int k = 0; k < 2; ++k) {
            hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, 100, sink, dclk);   // warm
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, k ? n2 : n1, sink, dclk);
            CK(hipEventRecord(e1));
            CK(hipDeviceSynchronize());
            CK(hipEventElapsedTime(&t[k], e0, e1));
            unsigned long long hc[2]; CK(hipMemcpy(hc, dclk, 16, hipMemcpyDeviceToHost));
            clk_mhz = (double)hc[0] / ((double)hc[1] / 100.0);
        }
        const double flops = (double)blocks * (threads / 64) * (double)(n2 - n1) * NACC * 16.0 * 16 * 32 * 2;
        printf("%s operands, %d wave(s)/SIMD: %.3f -> %.3f ms for %d -> %d x %d MFMA 16x16x32 bf16 per wave:  %7.1f TFLOP/s sustained, shader clock %.0f MHz\n",
               rnd ? "random" : "smooth", waves_per_simd, t[0], t[1], n1, n2, NACC, flops / (t[1] - t[0]) / 1e9, clk_mhz);
    }
    return 0;
}
