// Hardware probe: semantics of v_mfma_scale_f32_16x16x128_f8f6f4 with e4m3 operands as libivit uses it
// (lane l supplies row l&15, the two 16-byte k-chunks (l>>4) and 4+(l>>4) of a 128-byte row; first source
// = W fragment, second = A fragment, so the accumulator quad of lane l is out[m = l&15][n = (l>>4)*4 + r]).
// Prints the max error against a host f64 reference for the scale encodings tried.  Measurement aid only.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

static float e4m3_to_float(unsigned char b) {
    const int s = b >> 7, e = (b >> 3) & 15, m = b & 7;
    float v = e == 0 ? ldexpf((float)m, -9) : ldexpf(1.0f + m / 8.0f, e - 7);
    return s ? -v : v;
}

template <int MODE>
__global__ void probe(const unsigned char* A, const unsigned char* W, float* out) {
    const int lane = threadIdx.x, fr = lane & 15, fq = lane >> 4;
    i32x8 a, w;
    const int* ar = reinterpret_cast<const int*>(A + fr * 128);
    const int* wr = reinterpret_cast<const int*>(W + fr * 128);
    for (int i = 0; i < 4; ++i) {
        a[i] = ar[fq * 4 + i]; a[4 + i] = ar[(4 + fq) * 4 + i];
        w[i] = wr[fq * 4 + i]; w[4 + i] = wr[(4 + fq) * 4 + i];
    }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (MODE == 0) acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(w, a, acc, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    if (MODE == 1) acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(w, a, acc, 0, 0, 0, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[fr * 16 + fq * 4 + r] = acc[r];   // out[m][n]
}

int main() {
    std::vector<unsigned char> hA(16 * 128), hW(16 * 128);
    srand(3);
    auto rnd = [] { unsigned char b; do { b = rand() & 0xff; } while ((b & 0x7f) == 0x7f); return b; };   // no NaN
    for (auto& b : hA) b = rnd();
    for (auto& b : hW) b = rnd();
    unsigned char *dA, *dW; float* dout;
    hipMalloc(&dA, hA.size()); hipMalloc(&dW, hW.size()); hipMalloc(&dout, 256 * 4);
    hipMemcpy(dA, hA.data(), hA.size(), hipMemcpyHostToDevice);
    hipMemcpy(dW, hW.data(), hW.size(), hipMemcpyHostToDevice);
    for (int mode = 0; mode < 2; ++mode) {
        if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(1), dim3(64), 0, 0, dA, dW, dout);
        else hipLaunchKernelGGL(probe<1>, dim3(1), dim3(64), 0, 0, dA, dW, dout);
        std::vector<float> ho(256);
        hipMemcpy(ho.data(), dout, 1024, hipMemcpyDeviceToHost);
        double maxerr = 0, maxref = 0;
        for (int m = 0; m < 16; ++m)
            for (int n = 0; n < 16; ++n) {
                double ref = 0;
                for (int k = 0; k < 128; ++k) ref += (double)e4m3_to_float(hA[m * 128 + k]) * e4m3_to_float(hW[n * 128 + k]);
                maxerr = fmax(maxerr, fabs(ref - ho[m * 16 + n])); maxref = fmax(maxref, fabs(ref));
            }
        printf("mode %d (%s): max err %.4e, max |ref| %.4e, out[0][0..3] = %g %g %g %g\n", mode, mode ? "scale 0/0" : "scale 0x7f", maxerr, maxref, ho[0], ho[1], ho[2], ho[3]);
    }
    return 0;
}
