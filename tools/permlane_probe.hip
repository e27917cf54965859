// prints what v_permlane16_swap does (rows of 16 lanes), to pin the semantics the bf16 epilogue relies on
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
  unsigned a = threadIdx.x, b = 1000 + threadIdx.x;
  auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  out[threadIdx.x * 2] = r[0]; out[threadIdx.x * 2 + 1] = r[1];
}
int main() {
  unsigned* d; hipMalloc(&d, 64 * 2 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  unsigned h[128]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; l += 8) printf("lane %2d: first' = %4u  second' = %4u\n", l, h[l * 2], h[l * 2 + 1]);
  return 0;
}
