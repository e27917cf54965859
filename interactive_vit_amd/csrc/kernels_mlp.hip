// Instantiations and launchers of the fused MLP kernel (mlp_fused_kernel.h): LN2 (folded) -> up -> GELU -> down -> residual in one launch,
// and the one-time re-layout of its two weight matrices into the fragment-native stream the kernel reads.
#include "mlp_fused_kernel.h"
#include <cstdlib>

namespace ivit {

// one workgroup of 8 waves per CU (X + two U buffers in LDS: 128 KiB at D = 768; up to 256 VGPRs per lane)
// (f16x: both weight matrices as hi / lo pairs; f16x1: only the up weight W1' - the down weight plain)
#define IVIT_MLPF_KERNEL(NAME, ND, S1, S2, OP, ...)                                           \
    __global__ __launch_bounds__(512, 2) void NAME(MlpFusedParams p) {                        \
        extern __shared__ __attribute__((aligned(16))) char smem[];                          \
        mlp_fused_body<ND, S1, S2, OP __VA_OPT__(,) __VA_ARGS__>(p, smem);                    \
    }
IVIT_MLPF_KERNEL(ivit_mlp_fused_bf16_d768, 12, 1, 1, OpBf16)
IVIT_MLPF_KERNEL(ivit_mlp_fused_f16_d768, 12, 1, 1, OpF16)
IVIT_MLPF_KERNEL(ivit_mlp_fused_f16x_d768, 12, 2, 2, OpF16)
IVIT_MLPF_KERNEL(ivit_mlp_fused_f16x1_d768, 12, 2, 1, OpF16)
IVIT_MLPF_KERNEL(ivit_mlp_fused_bf16_d512, 8, 1, 1, OpBf16)
IVIT_MLPF_KERNEL(ivit_mlp_fused_f16_d512, 8, 1, 1, OpF16)
IVIT_MLPF_KERNEL(ivit_mlp_fused_f16x_d512, 8, 2, 2, OpF16)
IVIT_MLPF_KERNEL(ivit_mlp_fused_f16x1_d512, 8, 2, 1, OpF16)
#undef IVIT_MLPF_KERNEL

// the stream: for chunk c (128 hidden), wave w, block b (mlpf_block_source), lane l: 8 consecutive k of one weight row - 16 bytes at
// ((c * 8 + w) * CB + b) * 1024 + l * 16.  One thread per 16-byte unit.
template <int ND, int S1, int S2>
__global__ __launch_bounds__(256) void ivit_mlp_pack_weights(const bf16_t* __restrict__ W1, int ldw1, const bf16_t* __restrict__ W2, int ldw2, int nchunks,
                                                             bf16_t* __restrict__ out) {
    using G = MlpFusedGeom<ND, S1, S2>;
    const long long unit = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)nchunks * 8 * G::CB * 64;
    if (unit >= total) return;
    const int lane = (int)(unit & 63);
    const long long blk = unit >> 6;
    const int b = (int)(blk % G::CB), w = (int)((blk / G::CB) & 7), c = (int)(blk / (8 * G::CB));
    bool is_w1; int row, col;
    mlpf_block_source<ND, S1, S2>(c, w, b, lane, &is_w1, &row, &col);
    const bf16_t* src = is_w1 ? W1 + (size_t)row * ldw1 + col : W2 + (size_t)row * ldw2 + col;
    *reinterpret_cast<u32x4*>(out + unit * 8) = *reinterpret_cast<const u32x4*>(src);
}

// D = 768 (ViT-B) and 512 (test models): whole 64-column statistics slots per wave (D >= 512) plus at most half of a shared one, X (D / 64 x
// 8 KiB) + two U buffers within the CU's 160 KiB of LDS, and a wave's share of the [64, D] f32 accumulator (D / 8 registers) within its
// register file.  The hidden width walks in chunks of 128; weight pairs only on the f16 path.
bool mlp_fused_supported(int M, int D, int Mlp, int f16, int split) {
    if (M <= 0 || (D != 768 && D != 512) || Mlp <= 0 || (Mlp % 128)) return false;
    if (split < 0 || split > 2 || (split && !f16)) return false;
    return true;
}

// split: 0 = plain 16-bit weights, 1 = both matrices as hi / lo pairs, 2 = only the up weight W1' as pairs (the down weight plain)
size_t mlp_fused_packed_bytes(int D, int Mlp, int split) {
    const size_t s1 = split ? 2 : 1, s2 = split == 1 ? 2 : 1;
    return (size_t)D * Mlp * 2 * (s1 + s2);   // both matrices, every element once
}

hipError_t launch_mlp_pack_weights(const bf16_t* W1, int ldw1, const bf16_t* W2, int ldw2, int D, int Mlp, int split, bf16_t* out, hipStream_t stream) {
    if (!mlp_fused_supported(1, D, Mlp, split ? 1 : 0, split) || !W1 || !W2 || !out) return hipErrorInvalidValue;
    const int s1 = split ? 2 : 1, s2 = split == 1 ? 2 : 1;
    if (ldw1 < s1 * D || ldw2 < s2 * Mlp || (ldw1 % 8) || (ldw2 % 8)) return hipErrorInvalidValue;
    const int nchunks = Mlp / 128;
    const long long units = (long long)mlp_fused_packed_bytes(D, Mlp, split) / 16;
    const dim3 grid((unsigned)((units + 255) / 256));
#define IVIT_PACK(ND, S1, S2) hipLaunchKernelGGL((ivit_mlp_pack_weights<ND, S1, S2>), grid, dim3(256), 0, stream, W1, ldw1, W2, ldw2, nchunks, out)
    if (D == 768) { if (split == 1) IVIT_PACK(12, 2, 2); else if (split == 2) IVIT_PACK(12, 2, 1); else IVIT_PACK(12, 1, 1); }
    else { if (split == 1) IVIT_PACK(8, 2, 2); else if (split == 2) IVIT_PACK(8, 2, 1); else IVIT_PACK(8, 1, 1); }
#undef IVIT_PACK
    return hipGetLastError();
}

const char* mlp_fused_kernel_name(const MlpFusedParams& p) {
    if (p.D == 768) return p.split == 1 ? "ivit_mlp_fused_f16x_d768" : p.split == 2 ? "ivit_mlp_fused_f16x1_d768" : p.f16 ? "ivit_mlp_fused_f16_d768" : "ivit_mlp_fused_bf16_d768";
    return p.split == 1 ? "ivit_mlp_fused_f16x_d512" : p.split == 2 ? "ivit_mlp_fused_f16x1_d512" : p.f16 ? "ivit_mlp_fused_f16_d512" : "ivit_mlp_fused_bf16_d512";
}

hipError_t launch_mlp_fused(const MlpFusedParams& p, hipStream_t stream) {
    if (!mlp_fused_supported(p.M, p.D, p.Mlp, p.f16, p.split)) return hipErrorInvalidValue;
    if (!p.X || !p.ln_part_in || !p.Wp || !p.c1 || !p.s1 || !p.b2 || !p.resid || !p.out) return hipErrorInvalidValue;
    if (p.stats_out && (!p.xb || !p.ln_part_out || (p.ldxb % 8))) return hipErrorInvalidValue;
    if ((p.ldx % 8) || (p.ldo % 4) || (p.ldr % 4) || p.ldx < p.D) return hipErrorInvalidValue;   // 16-byte rows for the DMA / the f32 quads
    void (*kernel)(MlpFusedParams) =
        p.D == 768 ? (p.split == 1 ? ivit_mlp_fused_f16x_d768 : p.split == 2 ? ivit_mlp_fused_f16x1_d768 : p.f16 ? ivit_mlp_fused_f16_d768 : ivit_mlp_fused_bf16_d768)
                   : (p.split == 1 ? ivit_mlp_fused_f16x_d512 : p.split == 2 ? ivit_mlp_fused_f16x1_d512 : p.f16 ? ivit_mlp_fused_f16_d512 : ivit_mlp_fused_bf16_d512);
    const int lds = (p.D / 64) * 8192 + 2 * 16384;
    hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(kernel), lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kernel, dim3(ceil_div(p.M, 64)), dim3(512), lds, stream, p);
    return hipGetLastError();
}

}  // namespace ivit
