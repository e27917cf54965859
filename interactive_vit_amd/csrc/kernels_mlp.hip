// Instantiations and launcher of the fused MLP kernel (mlp_fused_kernel.h): LN2 (folded) -> up -> GELU -> down -> residual in one launch.
#include "mlp_fused_kernel.h"
#include <cstdlib>

namespace ivit {

// one workgroup of 8 waves per CU (160 KiB of LDS at D = 768; up to 256 VGPRs per lane)
#define IVIT_MLPF_KERNEL(NAME, ND, SPLIT, OP)                                                 \
    __global__ __launch_bounds__(512, 2) void NAME(MlpFusedParams p) {                        \
        extern __shared__ __attribute__((aligned(16))) char smem[];                          \
        mlp_fused_body<ND, SPLIT, OP>(p, smem);                                               \
    }
IVIT_MLPF_KERNEL(ivit_mlp_fused_bf16_d768, 12, 1, OpBf16)
IVIT_MLPF_KERNEL(ivit_mlp_fused_f16_d768, 12, 1, OpF16)
IVIT_MLPF_KERNEL(ivit_mlp_fused_f16x_d768, 12, 2, OpF16)
IVIT_MLPF_KERNEL(ivit_mlp_fused_bf16_d512, 8, 1, OpBf16)
IVIT_MLPF_KERNEL(ivit_mlp_fused_f16_d512, 8, 1, OpF16)
IVIT_MLPF_KERNEL(ivit_mlp_fused_f16x_d512, 8, 2, OpF16)
#undef IVIT_MLPF_KERNEL

// D = 768 (ViT-B) and 512 (test models): X (D / 64 x 8 KiB) + the four-slot ring must fit the CU's 160 KiB of LDS, and a wave's share of the
// [64, D] f32 accumulator (D / 8 registers) its register file.  The hidden width walks in chunks of 128; weight pairs only on the f16 path.
bool mlp_fused_supported(int M, int D, int Mlp, int f16, int split) {
    if (M <= 0 || (D != 768 && D != 512) || Mlp <= 0 || (Mlp % 128)) return false;
    if (split && !f16) return false;
    return true;
}

const char* mlp_fused_kernel_name(const MlpFusedParams& p) {
    if (p.D == 768) return p.split ? "ivit_mlp_fused_f16x_d768" : p.f16 ? "ivit_mlp_fused_f16_d768" : "ivit_mlp_fused_bf16_d768";
    return p.split ? "ivit_mlp_fused_f16x_d512" : p.f16 ? "ivit_mlp_fused_f16_d512" : "ivit_mlp_fused_bf16_d512";
}

hipError_t launch_mlp_fused(const MlpFusedParams& p, hipStream_t stream) {
    if (!mlp_fused_supported(p.M, p.D, p.Mlp, p.f16, p.split)) return hipErrorInvalidValue;
    const int sp = p.split ? 2 : 1;
    if (!p.X || !p.ln_part_in || !p.W1 || !p.c1 || !p.s1 || !p.W2 || !p.b2 || !p.resid || !p.out) return hipErrorInvalidValue;
    if (p.stats_out && (!p.xb || !p.ln_part_out || (p.ldxb % 8))) return hipErrorInvalidValue;
    if ((p.ldx % 8) || (p.ldw1 % 8) || (p.ldw2 % 8) || (p.ldo % 4) || (p.ldr % 4)) return hipErrorInvalidValue;   // 16-byte rows for the DMA / the f32 quads
    if (p.ldx < p.D || p.ldw1 < sp * p.D || p.ldw2 < sp * p.Mlp) return hipErrorInvalidValue;
    if ((long long)p.ldw1 * 2 * 256 >= (1ll << 31) || (long long)p.ldw2 * 2 * 256 >= (1ll << 31)) return hipErrorInvalidValue;   // 32-bit lane offsets of the DMA sources
    void (*kernel)(MlpFusedParams) =
        p.D == 768 ? (p.split ? ivit_mlp_fused_f16x_d768 : p.f16 ? ivit_mlp_fused_f16_d768 : ivit_mlp_fused_bf16_d768)
                   : (p.split ? ivit_mlp_fused_f16x_d512 : p.f16 ? ivit_mlp_fused_f16_d512 : ivit_mlp_fused_bf16_d512);
    const int lds = (p.D / 64) * 8192 + 4 * 16384;
    hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(kernel), lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kernel, dim3(ceil_div(p.M, 64)), dim3(512), lds, stream, p);
    return hipGetLastError();
}

}  // namespace ivit
