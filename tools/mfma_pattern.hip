// Probe: issue rate of the 5 x 8 fragment MFMA pattern of the 160x256 four-wave GEMM (13 operand quads ->
// 40 accumulators) from registers only, accumulators in AGPRs (inline asm) or wherever the compiler puts
// them (builtin).  One wave per SIMD, 256 CUs.  Measurement aid only.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int MODE, int FM, int FN>
__global__ __launch_bounds__(256, 1) void pattern(int iters, float* sink, unsigned long long* clocks) {
    bf16x8_t f[FM + FN];
    unsigned h = (blockIdx.x * 1024u + threadIdx.x) * 2654435761u + 12345u;
    for (int s = 0; s < FM + FN; ++s)
        for (int i = 0; i < 8; ++i) { h = h * 1664525u + 1013904223u; f[s][i] = (__bf16)((float)(int)(h >> 8) * (1.0f / 8388608.0f) - 1.0f); }
    f32x4 acc[FM][FN];
    for (int i = 0; i < FM; ++i) for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    unsigned long long c0, c1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(c0)::"memory");
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                if (MODE == 0) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(f[FM + j]), "v"(f[i]));
                if (MODE == 1) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i][j]) : "v"(f[FM + j]), "v"(f[i]));
                if (MODE == 2) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[FM + j], f[i], acc[i][j], 0, 0, 0);
            }
    }
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(c1)::"memory");
    float s = 0.f;
    for (int i = 0; i < FM; ++i) for (int j = 0; j < FN; ++j) s += acc[i][j][0] + acc[i][j][3];
    if (s == 123.456f) sink[0] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) clocks[0] = c1 - c0;
}

template <int MODE, int FM, int FN>
void run(const char* label, float* sink, unsigned long long* dclk) {
    const int iters = 20000;
    hipLaunchKernelGGL((pattern<MODE, FM, FN>), dim3(256), dim3(256), 0, 0, 100, sink, dclk);
    CK(hipDeviceSynchronize());
    hipLaunchKernelGGL((pattern<MODE, FM, FN>), dim3(256), dim3(256), 0, 0, iters, sink, dclk);
    CK(hipDeviceSynchronize());
    unsigned long long hc; CK(hipMemcpy(&hc, dclk, 8, hipMemcpyDeviceToHost));
    printf("%-52s %5.1f shader cycles per MFMA\n", label, (double)hc / ((double)iters * FM * FN));
}

int main() {
    float* sink; unsigned long long* dclk;
    CK(hipMalloc(&sink, 4)); CK(hipMalloc(&dclk, 16));
    run<0, 5, 8>("5x8, accumulators in AGPRs (asm +a)", sink, dclk);
    run<1, 5, 4>("5x4, accumulators in VGPRs (asm +v)", sink, dclk);
    run<2, 5, 4>("5x4, builtin", sink, dclk);
    run<0, 5, 4>("5x4, accumulators in AGPRs (asm +a)", sink, dclk);
    run<2, 5, 8>("5x8, builtin", sink, dclk);
    run<0, 2, 2>("2x2, AGPR (each accumulator again after 4 MFMAs)", sink, dclk);
    run<0, 1, 8>("1x8, AGPR (A operand constant)", sink, dclk);
    return 0;
}
