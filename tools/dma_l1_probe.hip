// LDS-DMA source-locality probe (round 3): bytes per clock a CU's global -> LDS path delivers when the source of the
// global_load_lds_dwordx4 stream is (A) a per-workgroup region that fits the XCD's L2 but not the CU's L1, (B) a per-workgroup region small
// enough for the three resident workgroups to share the 32 KiB L1, (C) ONE region shared by every workgroup.  Question behind it: would
// co-resident GEMM workgroups that share their W column tile (L1 hits for two of three) lift the 40 B/clk/CU ceiling of the operand path?
//   hipcc -O3 --offload-arch=gfx950 tools/dma_l1_probe.hip -o tools/dma_l1_probe.bin && tools/dma_l1_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
#define GPTR __attribute__((address_space(1)))
#define LPTR __attribute__((address_space(3)))

__global__ __launch_bounds__(256) void probe(const char* __restrict__ src, size_t stride_wg, unsigned span, int iters, int pieces, unsigned long long* out) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const char* base = src + (size_t)blockIdx.x * stride_wg + lane * 16;
    unsigned off = 0;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        for (int p = wave; p < pieces; p += 4) {
            unsigned o = off + p * 1024;
            o %= span;   // (a piece may run up to 1 KiB past the region's end: the buffer has slack)
            __builtin_amdgcn_global_load_lds((const GPTR void*)(base + o), (LPTR void*)(lds + p * 1024), 16, 0, 0);
        }
        off += pieces * 1024;
        while (off >= span) off -= span;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}

// the same stream through registers: global_load_dwordx4 -> VGPRs -> ds_write_b128 (9 pieces per wave in flight)
__global__ __launch_bounds__(256) void probe_reg(const char* __restrict__ src, size_t stride_wg, unsigned span, int iters, int pieces, unsigned long long* out, int write_lds) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const char* base = src + (size_t)blockIdx.x * stride_wg + lane * 16;
    unsigned off = 0;
    u32x4 acc = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        u32x4 v[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            unsigned o = (off + (i * 4 + wave) * 1024) % span;
            v[i] = *reinterpret_cast<const u32x4*>(base + o);
        }
        off = (off + pieces * 1024) % span;
        if (write_lds) {
#pragma unroll
            for (int i = 0; i < 9; ++i) *reinterpret_cast<u32x4*>(lds + (i * 4 + wave) * 1024 + lane * 16) = v[i];
        } else {
#pragma unroll
            for (int i = 0; i < 9; ++i) acc ^= v[i];
        }
        __builtin_amdgcn_s_barrier();
    }
    if (acc[0] == 0x12345678u && acc[1] == 1u) out[blockIdx.x] = acc[2];   // keeps the loads alive
}

int main() {
    const int wgs = 768, iters = 200, pieces = 36;
    const size_t buf = (size_t)wgs * (1 << 20);
    char* d; CK(hipMalloc(&d, buf + (64 << 10))); CK(hipMemset(d, 1, buf + (64 << 10)));
    unsigned long long* dout; CK(hipMalloc(&dout, wgs * 8));
    CK(hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 52 * 1024));
    CK(hipFuncSetAttribute((const void*)probe_reg, hipFuncAttributeMaxDynamicSharedMemorySize, 52 * 1024));
    struct Mode { const char* name; size_t stride; unsigned span; };
    const Mode modes[] = {
        {"A: 36 KiB per workgroup, re-read (L2 hits; 108 KiB per CU > L1)", 1 << 20, 36 * 1024},
        {"A': 1 MiB per workgroup, cycled (MALL / HBM)", 1 << 20, 1 << 20},
        {"B: 9 KiB per workgroup, cycled (three workgroups fit L1)", 1 << 20, 9 * 1024},
        {"C: one 36 KiB region for every workgroup", 0, 36 * 1024},
        {"C': one 9 KiB region for every workgroup", 0, 9 * 1024},
    };
    for (int lds_kb : {50, 36}) {   // 50 KiB: three workgroups per CU; 36 KiB: four
        printf("dynamic LDS %d KiB per workgroup (%d per CU), %d pieces of 1 KiB per iteration, wait + barrier per iteration\n", lds_kb, 160 / lds_kb, pieces);
        for (const Mode& m : modes) {
            std::vector<double> bpc;
            for (int rep = 0; rep < 5; ++rep) {
                hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
                CK(hipEventRecord(e0, 0));
                hipLaunchKernelGGL(probe, dim3(wgs), dim3(256), lds_kb * 1024, 0, d, m.stride, m.span, iters, pieces, dout);
                CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                std::vector<unsigned long long> h(wgs); CK(hipMemcpy(h.data(), dout, wgs * 8, hipMemcpyDeviceToHost));
                double cyc = 0; for (auto v : h) cyc += (double)v; cyc /= wgs;       // s_memtime ticks (100 MHz constant clock)
                const double bytes_wg = (double)iters * pieces * 1024;
                // chip-level: all bytes / kernel time
                bpc.push_back((double)wgs * bytes_wg / (ms * 1e-3) / 1e9);
                (void)cyc;
            }
            std::sort(bpc.begin(), bpc.end());
            printf("   %-62s %8.1f GB/s chip = %6.1f GB/s per CU = %5.1f B/clk/CU at 2.4 GHz\n", m.name, bpc[2], bpc[2] / 256, bpc[2] / 256 / 2.4);
            for (int wl = 1; wl >= 0; --wl) {
                std::vector<double> r;
                for (int rep = 0; rep < 5; ++rep) {
                    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
                    CK(hipEventRecord(e0, 0));
                    hipLaunchKernelGGL(probe_reg, dim3(wgs), dim3(256), lds_kb * 1024, 0, d, m.stride, m.span, iters, pieces, dout, wl);
                    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                    r.push_back((double)wgs * iters * pieces * 1024 / (ms * 1e-3) / 1e9);
                }
                std::sort(r.begin(), r.end());
                printf("      through registers%s: %8.1f GB/s chip = %5.1f B/clk/CU\n", wl ? " + ds_write_b128" : " (no LDS write)       ", r[2], r[2] / 256 / 2.4);
            }
        }
    }
    return 0;
}
