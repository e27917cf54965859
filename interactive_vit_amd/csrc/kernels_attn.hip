// Multi-head self-attention core for ViT token counts (17 .. 608 keys), head dim 64:
//
//     out[b, q, h*64 + d] = sum_key softmax_key(scale * Q[q].K[key]) * V[key][d]
//
// One workgroup = 4 waves = 64 queries of one (image, head); each wave owns 16 queries.  All keys
// of the head fit in LDS (197 keys: 28 KiB K + 29 KiB V^T), so the softmax is a single exact pass
// with every score of the wave's 16 queries held in registers - no online rescaling is needed at
// these sequence lengths.
//
// Both contractions run on v_mfma_f32_16x16x32_bf16 with the key index on the accumulator ROW:
//   S^T = K . Q^T    (A operand = K rows from LDS, B operand = the wave's Q fragments)
//         -> lane (fr = lane&15, g = lane>>4) holds S[q = fr][key = 16 f + 4 g + j], j = 0..3
//   O^T = V^T . P^T  (A operand = V^T rows from LDS, B operand = P straight from those registers)
//         -> lane holds O[q = fr][d = 16 dblk + 4 g + j]: an 8-byte bf16 store per fragment
// The k order inside an MFMA step may be any permutation as long as both operands agree, so the
// P registers of fragments 2s and 2s+1 are used as the 8 k-slots of step s without lane movement;
// V^T is read with the matching key order (two 8-byte reads per fragment).
// Softmax statistics are fp32; the row reductions are wavefront shuffles across the 4 lane groups.
#include "kernels.h"

namespace ivit {

constexpr int ATT_DH = 64;
constexpr int ATT_THREADS = 256;
constexpr int ATT_QPB = 64;  // queries per block

template <int NKF>
struct AttLayout {
    static constexpr int KEYS = NKF * 16;
    static constexpr int K_BYTES = KEYS * ATT_DH * 2;
    static constexpr int VT_STRIDE_DW = NKF * 8 + 4;  // dwords per V^T row; == 4 (mod 8): conflict-free b64 reads
    static constexpr int VT_BYTES = ATT_DH * VT_STRIDE_DW * 4;
    static constexpr int LDS_BYTES = K_BYTES + VT_BYTES;
};

template <int NKF>
__global__ __launch_bounds__(ATT_THREADS) void ivit_attention_bf16(AttnParams p) {
    using L = AttLayout<NKF>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* k_lds = smem;
    char* vt_lds = smem + L::K_BYTES;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int fr = lane & 15, g = lane >> 4;
    const int b = blockIdx.z, h = blockIdx.y;
    const int N = p.tokens;
    const int D = p.heads * ATT_DH;
    const size_t row0 = (size_t)b * N;
    const bf16_t* qkv = p.qkv;
    const int ld = p.ldqkv;

    // ---- stage K (row-major, 16-B chunks XOR-swizzled by key&7) and V^T (key-contiguous rows)
    for (int c = threadIdx.x; c < L::KEYS * 8; c += ATT_THREADS) {
        const int key = c >> 3, ch = c & 7;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (key < N) v = *reinterpret_cast<const u32x4*>(qkv + (row0 + key) * ld + D + h * ATT_DH + ch * 8);
        *reinterpret_cast<u32x4*>(k_lds + key * 128 + ((ch ^ (key & 7)) << 4)) = v;
    }
    for (int c = threadIdx.x; c < L::KEYS * 8; c += ATT_THREADS) {
        const int key = c % L::KEYS, ch = c / L::KEYS;   // lanes run over keys: conflict-light 2-byte writes
        u32x4 v = {0u, 0u, 0u, 0u};                       // keys >= N must be finite (0 * NaN would poison PV)
        if (key < N) v = *reinterpret_cast<const u32x4*>(qkv + (row0 + key) * ld + 2 * D + h * ATT_DH + ch * 8);
        bf16_t* dst = reinterpret_cast<bf16_t*>(vt_lds) + key;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            dst[(size_t)(ch * 8 + 2 * i) * (L::VT_STRIDE_DW * 2)] = (bf16_t)(v[i] & 0xffffu);
            dst[(size_t)(ch * 8 + 2 * i + 1) * (L::VT_STRIDE_DW * 2)] = (bf16_t)(v[i] >> 16);
        }
    }
    __syncthreads();

    const int q0 = blockIdx.x * ATT_QPB + wave * 16;
    if (q0 >= N) return;  // whole wave idle (after the only barrier)

    // ---- Q fragments (B operand): lane holds Q[q0+fr][kk*32 + 8g .. +7]
    const int qrow = min(q0 + fr, N - 1);
    bf16x8 qf[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
        qf[kk] = *reinterpret_cast<const bf16x8*>(qkv + (row0 + qrow) * ld + h * ATT_DH + kk * 32 + g * 8);

    // ---- S^T = K Q^T
    f32x4 s[NKF];
#pragma unroll
    for (int f = 0; f < NKF; ++f) {
        const int key = f * 16 + fr;
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const bf16x8 kf = *reinterpret_cast<const bf16x8*>(k_lds + key * 128 + (((kk * 4 + g) ^ (key & 7)) << 4));
            a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[kk], a, 0, 0, 0);
        }
        s[f] = a;
    }

    // ---- softmax over keys (row q = fr lives in this lane's registers and the 3 other lane groups)
    float mx = -INFINITY;
#pragma unroll
    for (int f = 0; f < NKF; ++f)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int key = f * 16 + g * 4 + j;
            const float v = (key < N) ? s[f][j] : -INFINITY;
            s[f][j] = v;
            mx = fmaxf(mx, v);
        }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float c = p.scale * 1.44269504088896340736f;  // exp(x*scale) = exp2(x*scale*log2 e)
    const float mc = mx * c;
    float sum = 0.f;
#pragma unroll
    for (int f = 0; f < NKF; ++f)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float e = __builtin_amdgcn_exp2f(fmaf(s[f][j], c, -mc));
            s[f][j] = e;
            sum += e;
        }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.0f / sum;

    // ---- O^T = V^T P^T
    f32x4 o[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) o[d] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int st = 0; st < NKF / 2; ++st) {
        const f32x4 p0 = s[2 * st], p1 = s[2 * st + 1];
        union { bf16x8 v; unsigned int u[4]; } pf;
        pf.u[0] = pack_bf16x2(p0[0], p0[1]);
        pf.u[1] = pack_bf16x2(p0[2], p0[3]);
        pf.u[2] = pack_bf16x2(p1[0], p1[1]);
        pf.u[3] = pack_bf16x2(p1[2], p1[3]);
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const char* vrow = vt_lds + (size_t)(d * 16 + fr) * (L::VT_STRIDE_DW * 4) + (32 * st + 4 * g) * 2;
            union { bf16x8 v; u32x2 h2[2]; } vf;
            vf.h2[0] = *reinterpret_cast<const u32x2*>(vrow);
            vf.h2[1] = *reinterpret_cast<const u32x2*>(vrow + 32);
            o[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf.v, pf.v, o[d], 0, 0, 0);
        }
    }

    // ---- normalise and store: lane holds O[q0+fr][16 d + 4 g .. +3]
    if (q0 + fr < N) {
        bf16_t* orow = p.out + (row0 + q0 + fr) * p.ldo + h * ATT_DH + g * 4;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            u32x2 pk = {pack_bf16x2(o[d][0] * inv, o[d][1] * inv), pack_bf16x2(o[d][2] * inv, o[d][3] * inv)};
            *reinterpret_cast<u32x2*>(orow + d * 16) = pk;
        }
    }
}

bool attention_supported(int tokens, int head_dim) { return head_dim == ATT_DH && tokens >= 1 && tokens <= 38 * 16; }

template <int NKF>
static hipError_t launch_nkf(const AttnParams& p, hipStream_t stream) {
    using L = AttLayout<NKF>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(ivit_attention_bf16<NKF>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, L::LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    dim3 grid(ceil_div(p.tokens, ATT_QPB), p.heads, p.batch);
    hipLaunchKernelGGL(ivit_attention_bf16<NKF>, grid, dim3(ATT_THREADS), L::LDS_BYTES, stream, p);
    return hipGetLastError();
}

hipError_t launch_attention(const AttnParams& p, hipStream_t stream) {
    if (!attention_supported(p.tokens, p.head_dim)) return hipErrorInvalidValue;
    if ((p.ldqkv % 8) || (p.ldo % 4)) return hipErrorInvalidValue;
    const int nkf = round_up(ceil_div(p.tokens, 16), 2);
    if (nkf <= 2) return launch_nkf<2>(p, stream);
    if (nkf <= 4) return launch_nkf<4>(p, stream);
    if (nkf <= 8) return launch_nkf<8>(p, stream);
    if (nkf <= 14) return launch_nkf<14>(p, stream);   // 197 tokens (224^2 / 16)
    if (nkf <= 18) return launch_nkf<18>(p, stream);   // 257 tokens (224^2 / 14)
    if (nkf <= 26) return launch_nkf<26>(p, stream);
    return launch_nkf<38>(p, stream);                  // 577 tokens (384^2 / 16)
}

}  // namespace ivit
