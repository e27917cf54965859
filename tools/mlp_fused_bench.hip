// Fused MLP kernel (csrc/mlp_fused_kernel.h) against the two GEMM launches it replaces: BITWISE comparison of every output (f32 stream, 16-bit
// copy, statistics pairs) and interleaved timing on random data (cdna_hip_programming.md rules 24 / 25).  Links against libivit.so (the product
// kernels):   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/mlp_fused_bench.hip -Linteractive_vit_amd -livit -Wl,-rpath,'$ORIGIN/../interactive_vit_amd' -o tools/mlp_fused_bench.bin
// usage: mlp_fused_bench.bin [M = 12608] [mode: 0 bf16, 1 f16, 2 f16x] [rounds = 9] [D = 768] [Mlp = 4 D]
#define IVIT_MLPF_STAMPS
#include "../interactive_vit_amd/csrc/mlp_fused_kernel.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
using namespace ivit;

// study instantiations of the same body: timing ablations and per-workgroup section stamps (the product kernels live in libivit.so)
template <int DBG>
__global__ __launch_bounds__(512, 2) void mlpf_study_bf16(MlpFusedParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    mlp_fused_body<12, 1, 1, OpBf16, DBG>(p, smem);
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

int main(int argc, char** argv) {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const int M = argc > 1 ? atoi(argv[1]) : 64 * 197;
    const int mode = argc > 2 ? atoi(argv[2]) : 0;
    const int rounds = argc > 3 ? atoi(argv[3]) : 9;
    const int D = argc > 4 ? atoi(argv[4]) : 768;
    const int Mlp = argc > 5 ? atoi(argv[5]) : 4 * D;
    const int f16 = mode >= 1, split = mode == 2;
    if (!mlp_fused_supported(M, D, Mlp, f16, split)) { printf("shape not supported by the fused kernel\n"); return 1; }
    const int Mpad = round_up(M, 256) + 256;
    const int sp = split ? 2 : 1;
    std::mt19937 rng(7);
    std::normal_distribution<float> nd(0.f, 1.f);

    // residual stream x (f32), its 16-bit copy + statistics pairs (what the out-projection's epilogue leaves), weights, fold vectors
    std::vector<float> hx((size_t)Mpad * D, 0.f), hw1((size_t)Mlp * D), hw2((size_t)D * Mlp), hg(D), hbeta(D), hb1(Mlp), hb2(D);
    for (size_t i = 0; i < (size_t)M * D; ++i) hx[i] = nd(rng) + 0.1f;
    for (auto& v : hw1) v = 0.02f * nd(rng);
    for (auto& v : hw2) v = 0.02f * nd(rng);
    for (auto& v : hg) v = 1.0f + 0.1f * nd(rng);
    for (auto& v : hbeta) v = 0.1f * nd(rng);
    for (auto& v : hb1) v = 0.1f * nd(rng);
    for (auto& v : hb2) v = 0.1f * nd(rng);
    float *dx, *dw1f, *dw2f, *dg, *dbeta, *db1, *db2, *ds1, *dc1;
    CK(hipMalloc(&dx, hx.size() * 4)); CK(hipMalloc(&dw1f, hw1.size() * 4)); CK(hipMalloc(&dw2f, hw2.size() * 4));
    CK(hipMalloc(&dg, D * 4)); CK(hipMalloc(&dbeta, D * 4)); CK(hipMalloc(&db1, Mlp * 4)); CK(hipMalloc(&db2, D * 4)); CK(hipMalloc(&ds1, Mlp * 4)); CK(hipMalloc(&dc1, Mlp * 4));
    CK(hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dw1f, hw1.data(), hw1.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dw2f, hw2.data(), hw2.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dg, hg.data(), D * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dbeta, hbeta.data(), D * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(db1, hb1.data(), Mlp * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(db2, hb2.data(), D * 4, hipMemcpyHostToDevice));
    bf16_t *dxb, *dw1, *dw1u, *dw2, *du;
    float2* dpart;
    const int ldw1 = sp * D, ldw2 = sp * Mlp;
    CK(hipMalloc(&dxb, (size_t)Mpad * D * 2)); CK(hipMemset(dxb, 0, (size_t)Mpad * D * 2));
    CK(hipMalloc(&dpart, (size_t)Mpad * GEMM_LN_SLOTS * 8)); CK(hipMemset(dpart, 0, (size_t)Mpad * GEMM_LN_SLOTS * 8));
    CK(hipMalloc(&dw1, (size_t)round_up(Mlp, 256) * ldw1 * 2)); CK(hipMemset(dw1, 0, (size_t)round_up(Mlp, 256) * ldw1 * 2));
    CK(hipMalloc(&dw1u, (size_t)round_up(Mlp, 256) * D * 2)); CK(hipMemset(dw1u, 0, (size_t)round_up(Mlp, 256) * D * 2));
    CK(hipMalloc(&dw2, (size_t)round_up(D, 256) * ldw2 * 2)); CK(hipMemset(dw2, 0, (size_t)round_up(D, 256) * ldw2 * 2));
    CK(hipMalloc(&du, (size_t)Mpad * Mlp * 2)); CK(hipMemset(du, 0, (size_t)Mpad * Mlp * 2));
    CK(launch_row_stats(dx, D, M, D, dxb, D, dpart, 0, f16));
    if (split) {   // hi / lo pairs interleaved per K-tile; W1' = W1 . gamma from the f32 matrix
        CK(launch_split_weight(dw1f, D, Mlp, D, dg, dbeta, db1, dw1, ldw1, D, 0, ds1, dc1, 0, f16));
        CK(launch_split_weight(dw2f, Mlp, D, Mlp, nullptr, nullptr, nullptr, dw2, ldw2, Mlp, 0, nullptr, nullptr, 0, f16));
    } else {
        CK(launch_f32_to_bf16(dw1f, D, dw1u, D, Mlp, D, 0, f16));
        CK(launch_fold_ln_weights(dw1u, D, Mlp, D, dg, dbeta, db1, dw1, ds1, dc1, 0, f16));
        CK(launch_f32_to_bf16(dw2f, Mlp, dw2, Mlp, D, Mlp, 0, f16));
    }
    bf16_t* dwp;   // both matrices in the fused kernel's stream order
    CK(hipMalloc(&dwp, mlp_fused_packed_bytes(D, Mlp, split)));
    CK(launch_mlp_pack_weights(dw1, ldw1, dw2, ldw2, D, Mlp, split, dwp, 0));
    CK(hipDeviceSynchronize());

    // outputs: reference pair of launches vs the fused kernel
    float *dy_ref, *dy; bf16_t *dxb_ref, *dxb_f; float2 *dpart_ref, *dpart_f;
    CK(hipMalloc(&dy_ref, (size_t)Mpad * D * 4)); CK(hipMalloc(&dy, (size_t)Mpad * D * 4));
    CK(hipMalloc(&dxb_ref, (size_t)Mpad * D * 2)); CK(hipMalloc(&dxb_f, (size_t)Mpad * D * 2));
    CK(hipMalloc(&dpart_ref, (size_t)Mpad * GEMM_LN_SLOTS * 8)); CK(hipMalloc(&dpart_f, (size_t)Mpad * GEMM_LN_SLOTS * 8));

    auto run_ref = [&](int stats, hipStream_t st) {
        GemmParams a{};
        a.A = dxb; a.lda = D; a.W = dw1; a.ldw = ldw1; a.M = M; a.N = Mlp; a.K = ldw1; a.f16 = f16; a.a_shift = split;
        a.bias = dc1; a.epi = EPI_LNFOLD_GELU_BF16; a.out = du; a.ldo = Mlp; a.ln_part = dpart; a.ln_s = ds1; a.ln_eps = 1e-6f; a.ln_dim = D;
        CK(launch_gemm(a, st));
        GemmParams b{};
        b.A = du; b.lda = Mlp; b.W = dw2; b.ldw = ldw2; b.M = M; b.N = D; b.K = ldw2; b.f16 = f16; b.a_shift = split;
        b.bias = db2; b.epi = stats ? EPI_BIAS_RESID_STATS : EPI_BIAS_RESID_F32; b.out = dy_ref; b.ldo = D; b.resid = dx; b.ldr = D;
        b.ln_part = dpart_ref; b.xb = dxb_ref; b.ldxb = D; b.ln_eps = 1e-6f; b.ln_dim = D;
        CK(launch_gemm(b, st));
    };
    auto run_fused = [&](int stats, hipStream_t st) {
        MlpFusedParams p{};
        p.X = dxb; p.ldx = D; p.ln_part_in = dpart; p.ln_eps = 1e-6f; p.Wp = dwp; p.c1 = dc1; p.s1 = ds1; p.b2 = db2;
        p.resid = dx; p.ldr = D; p.out = dy; p.ldo = D; p.xb = dxb_f; p.ldxb = D; p.ln_part_out = dpart_f; p.M = M; p.D = D; p.Mlp = Mlp; p.f16 = f16; p.split = split; p.stats_out = stats;
        CK(launch_mlp_fused(p, st));
    };

    int bad = 0;
    for (int stats = 1; stats >= 0; --stats) {
        CK(hipMemset(dy_ref, 0xff, (size_t)Mpad * D * 4)); CK(hipMemset(dy, 0xff, (size_t)Mpad * D * 4));
        CK(hipMemset(dxb_ref, 0xff, (size_t)Mpad * D * 2)); CK(hipMemset(dxb_f, 0xff, (size_t)Mpad * D * 2));
        CK(hipMemset(dpart_ref, 0xff, (size_t)Mpad * GEMM_LN_SLOTS * 8)); CK(hipMemset(dpart_f, 0xff, (size_t)Mpad * GEMM_LN_SLOTS * 8));
        run_ref(stats, 0); run_fused(stats, 0);
        CK(hipDeviceSynchronize());
        std::vector<float> y0((size_t)Mpad * D), y1((size_t)Mpad * D);
        CK(hipMemcpy(y0.data(), dy_ref, y0.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(y1.data(), dy, y1.size() * 4, hipMemcpyDeviceToHost));
        size_t diff = 0, nan = 0; double maxd = 0;
        for (size_t i = 0; i < y0.size(); ++i) {
            if (memcmp(&y0[i], &y1[i], 4)) { ++diff; maxd = std::max(maxd, (double)std::fabs(y0[i] - y1[i])); }
            if (i < (size_t)M * D && y1[i] != y1[i]) ++nan;
        }
        printf("stats_out=%d  f32 stream: %zu of %zu words differ (max |d| %.3g), %zu NaN inside the matrix (rows past M must stay untouched: compared too)\n", stats, diff, y0.size(), maxd, nan);
        bad += diff != 0 || nan != 0;
        std::vector<bf16_t> b0((size_t)Mpad * D), b1((size_t)Mpad * D);
        CK(hipMemcpy(b0.data(), dxb_ref, b0.size() * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(b1.data(), dxb_f, b1.size() * 2, hipMemcpyDeviceToHost));
        diff = 0; for (size_t i = 0; i < b0.size(); ++i) diff += b0[i] != b1[i];
        printf("             16-bit copy: %zu of %zu differ\n", diff, b0.size()); bad += diff != 0;
        std::vector<float2> p0((size_t)Mpad * GEMM_LN_SLOTS), p1((size_t)Mpad * GEMM_LN_SLOTS);
        CK(hipMemcpy(p0.data(), dpart_ref, p0.size() * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(p1.data(), dpart_f, p1.size() * 8, hipMemcpyDeviceToHost));
        diff = 0; for (size_t i = 0; i < p0.size(); ++i) diff += memcmp(&p0[i], &p1[i], 8) != 0;
        printf("             statistics pairs: %zu of %zu differ\n", diff, p0.size()); bad += diff != 0;
    }

    // timing: interleaved rounds, 20 launches each
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> tr, tf;
    const int reps = 20;
    for (int r = 0; r < rounds + 1; ++r) {
        float ms;
        CK(hipEventRecord(e0, 0)); for (int i = 0; i < reps; ++i) run_ref(1, 0); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); if (r) tr.push_back(ms * 1000 / reps);
        CK(hipEventRecord(e0, 0)); for (int i = 0; i < reps; ++i) run_fused(1, 0); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); if (r) tf.push_back(ms * 1000 / reps);
    }
    std::sort(tr.begin(), tr.end()); std::sort(tf.begin(), tf.end());
    const double flops = 4.0 * M * (double)D * Mlp * sp;
    printf("M=%d D=%d Mlp=%d mode=%s\n", M, D, Mlp, mode == 0 ? "bf16" : mode == 1 ? "f16" : "f16x");
    printf("   two launches (mlp1_lf + mlp2_rs): median %8.2f us  min %8.2f   %7.1f TFLOP/s (MFMA work incl. hi/lo passes)\n", tr[tr.size() / 2], tr[0], flops / tr[tr.size() / 2] * 1e-6);
    printf("   %-33s median %8.2f us  min %8.2f   %7.1f TFLOP/s\n", "fused", tf[tf.size() / 2], tf[0], flops / tf[tf.size() / 2] * 1e-6);
    if (mode == 0 && D == 768) {   // ablations of the bf16 ViT-B kernel + section stamps
        unsigned long long* dst; const int nwg = ceil_div(M, 64);
        CK(hipMalloc(&dst, (size_t)nwg * 8 * 8)); CK(hipMemset(dst, 0, (size_t)nwg * 8 * 8));
        const int lds = 12 * 8192 + 2 * 16384;
        auto study = [&](auto kern, const char* name, bool stamps) {
            CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
            MlpFusedParams p{};
            p.X = dxb; p.ldx = D; p.ln_part_in = dpart; p.ln_eps = 1e-6f; p.Wp = dwp; p.c1 = dc1; p.s1 = ds1; p.b2 = db2;
            p.resid = dx; p.ldr = D; p.out = dy; p.ldo = D; p.xb = dxb_f; p.ldxb = D; p.ln_part_out = dpart_f; p.M = M; p.D = D; p.Mlp = Mlp; p.stats_out = 1;
            p.stamps = stamps ? dst : nullptr;
            std::vector<float> t;
            for (int r = 0; r < rounds + 1; ++r) {
                float ms;
                CK(hipEventRecord(e0, 0)); for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(nwg), dim3(512), lds, 0, p);
                CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); if (r) t.push_back(ms * 1000 / reps);
            }
            std::sort(t.begin(), t.end());
            printf("   %-33s median %8.2f us  min %8.2f\n", name, t[t.size() / 2], t[0]);
            if (stamps) {
                std::vector<unsigned long long> h((size_t)nwg * 8);
                CK(hipMemcpy(h.data(), dst, h.size() * 8, hipMemcpyDeviceToHost));
                unsigned long long t0 = ~0ull; for (int b = 0; b < nwg; ++b) t0 = std::min(t0, h[(size_t)b * 8]);
                double s[4] = {0, 0, 0, 0}, last = 0;
                for (int b = 0; b < nwg; ++b) {
                    s[0] += (double)(h[b * 8 + 0] - t0); s[1] += (double)(h[b * 8 + 1] - h[b * 8 + 0]); s[2] += (double)(h[b * 8 + 2] - h[b * 8 + 1]); s[3] += (double)(h[b * 8 + 3] - h[b * 8 + 2]);
                    last = std::max(last, (double)(h[b * 8 + 3] - t0));
                }
                // s_memrealtime ticks at 100 MHz
                printf("      per workgroup (mean over %d, us): start skew %.2f | prologue %.2f | chunk loop %.2f | epilogue %.2f | last workgroup ends at %.2f\n", nwg,
                       s[0] / nwg / 100, s[1] / nwg / 100, s[2] / nwg / 100, s[3] / nwg / 100, last / 100);
                double e4 = 0, e5 = 0, e6 = 0, e7 = 0;   // wave 0's epilogue: ring drained | own slot done | shared-slot values stored | statistics chain + end
                for (int b = 0; b < nwg; ++b) { e4 += (double)(h[b * 8 + 4] - h[b * 8 + 2]); e5 += (double)(h[b * 8 + 5] - h[b * 8 + 4]); e6 += (double)(h[b * 8 + 6] - h[b * 8 + 5]); e7 += (double)(h[b * 8 + 3] - h[b * 8 + 6]); }
                printf("         epilogue of wave 0: look-ahead drained %.2f | own slot %.2f | shared-slot values %.2f | statistics chain %.2f\n", e4 / nwg / 100, e5 / nwg / 100, e6 / nwg / 100, e7 / nwg / 100);
            }
        };
        study(mlpf_study_bf16<0>, "study build, as the product", true);
        study(mlpf_study_bf16<1>, "no weight loads inside the loop", false);
        study(mlpf_study_bf16<2>, "no MFMA", false);
        study(mlpf_study_bf16<4>, "no LDS fragment reads", false);
        study(mlpf_study_bf16<5>, "MFMA + epilogues only", false);
        study(mlpf_study_bf16<6>, "weight loads only", false);
        study(mlpf_study_bf16<3>, "LDS reads only", false);
        study(mlpf_study_bf16<7>, "skeleton (phase-1 epilogue + barrier per chunk)", true);
    }
    printf(bad ? "BITWISE MISMATCH\n" : "bitwise identical\n");
    return bad ? 2 : 0;
}
