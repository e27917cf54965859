// Do two workgroups of 80 KiB of LDS each share a CU on gfx950 (160 KiB per CU)?  A kernel that spins for a fixed time; 256 vs 512 workgroups.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(384) void spin(unsigned long long ticks, int* sink) {
    extern __shared__ char smem[];
    smem[threadIdx.x] = (char)threadIdx.x;
    const unsigned long long t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < ticks) { __builtin_amdgcn_s_sleep(8); }
    if (smem[(threadIdx.x + 1) % 384] == 123 && sink) *sink = 1;
}
int main() {
    int lds_sizes[] = {81920, 81408, 80896, 65536, 54272};
    for (int lds : lds_sizes) {
        hipFuncSetAttribute((const void*)spin, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        for (int wgs : {256, 512, 768}) {
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            hipLaunchKernelGGL(spin, dim3(wgs), dim3(384), lds, 0, 100000ull, nullptr);   // warm
            hipDeviceSynchronize();
            hipEventRecord(a);
            hipLaunchKernelGGL(spin, dim3(wgs), dim3(384), lds, 0, 10000000ull, nullptr);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            printf("lds %d B, %d workgroups of 384 threads: %.3f ms (%s)\n", lds, wgs, ms, hipGetErrorString(hipGetLastError()));
        }
    }
    return 0;
}
