// Microbenchmark of ivit_qkv_attention_fused against the QKV GEMM + attention kernels it replaces (ViT-B/16 layer shapes,
// random data): bitwise comparison of the attention output, interleaved timing, and per-phase s_memrealtime stamps.
// Build: tools/build_tools.sh.   Usage: fused_bench.bin [batch] [rounds]
#include "../interactive_vit_amd/csrc/kernels.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
using namespace ivit;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
static bf16_t h_f2bf(float f) { unsigned u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (bf16_t)(u >> 16); }

int main(int argc, char** argv) {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const int B = argc > 1 ? atoi(argv[1]) : 64, rounds = argc > 2 ? atoi(argv[2]) : 7;
    const int N = 197, D = 768, H = 12, M = B * N, Mp = round_up(M, 256) + 256;
    std::mt19937 rng(3);
    std::uniform_real_distribution<float> u(-1.f, 1.f);
    std::vector<bf16_t> hx((size_t)Mp * D), hw((size_t)3 * D * D);
    for (auto& v : hx) v = h_f2bf(u(rng));
    for (auto& v : hw) v = h_f2bf(u(rng) * 0.05f);
    std::vector<float> hc(3 * D), hs(3 * D);
    for (auto& v : hc) v = u(rng) * 0.1f;
    for (auto& v : hs) v = u(rng) * 0.1f;
    std::vector<float2> hpart((size_t)Mp * GEMM_LN_SLOTS);
    for (auto& v : hpart) v = make_float2(u(rng) * 4.f, 20.f + u(rng));
    bf16_t *x, *w, *qkv, *att0, *att1; float *c, *s; float2* part; unsigned long long* stamps;
    CK(hipMalloc(&x, hx.size() * 2)); CK(hipMalloc(&w, hw.size() * 2)); CK(hipMalloc(&qkv, (size_t)Mp * 3 * D * 2));
    CK(hipMalloc(&att0, (size_t)Mp * D * 2)); CK(hipMalloc(&att1, (size_t)Mp * D * 2));
    CK(hipMalloc(&c, 3 * D * 4)); CK(hipMalloc(&s, 3 * D * 4)); CK(hipMalloc(&part, hpart.size() * 8));
    const int grid = ceil_div(B, 8) * 8 * H;
    CK(hipMalloc(&stamps, (size_t)grid * 128 * 8));
    CK(hipMemcpy(x, hx.data(), hx.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(c, hc.data(), 3 * D * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(s, hs.data(), 3 * D * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(part, hpart.data(), hpart.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemset(att0, 0, (size_t)Mp * D * 2)); CK(hipMemset(att1, 0, (size_t)Mp * D * 2));

    GemmParams g{};
    g.A = x; g.lda = D; g.W = w; g.ldw = D; g.M = M; g.N = 3 * D; g.K = D; g.bias = c; g.epi = EPI_LNFOLD_BF16; g.out = qkv; g.ldo = 3 * D;
    g.ln_part = part; g.ln_s = s; g.ln_eps = 1e-6f; g.ln_dim = D;
    AttnParams ap{};
    ap.qkv = qkv; ap.ldqkv = 3 * D; ap.out = att0; ap.ldo = D; ap.batch = B; ap.tokens = N; ap.heads = H; ap.head_dim = 64; ap.scale = 0.125f;
    FusedQkvAttnArgs f{};
    f.x = x; f.ldx = D; f.w = w; f.ldw = D; f.c = c; f.s = s; f.ln_part = part; f.ln_eps = 1e-6f; f.out = att1; f.ldo = D;
    f.batch = B; f.tokens = N; f.heads = H; f.head_dim = 64; f.dim = D; f.rows_total = M; f.scale = 0.125f;
    const char* dbg = getenv("IVIT_FUSED_DEBUG");
    f.debug = dbg ? atoi(dbg) : 0;

    CK(launch_gemm(g, 0)); CK(launch_attention(ap, 0)); CK(launch_fused_qkv_attention(f, 0)); CK(hipDeviceSynchronize());
    std::vector<bf16_t> h0((size_t)M * D), h1((size_t)M * D);
    CK(hipMemcpy(h0.data(), att0, h0.size() * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(h1.data(), att1, h1.size() * 2, hipMemcpyDeviceToHost));
    size_t diff = 0;
    for (size_t i = 0; i < h0.size(); ++i) diff += h0[i] != h1[i];
    printf("fused vs (%s + attention): %zu of %zu output elements differ\n", gemm_kernel_name(g), diff, h0.size());

    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> tg, ta, tf;
    const int reps = 20;
    for (int r = 0; r < rounds; ++r) {
        float ms;
        CK(hipEventRecord(e0, 0)); for (int i = 0; i < reps; ++i) CK(launch_gemm(g, 0)); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1)); tg.push_back(ms * 1000.f / reps);
        CK(hipEventRecord(e0, 0)); for (int i = 0; i < reps; ++i) CK(launch_attention(ap, 0)); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1)); ta.push_back(ms * 1000.f / reps);
        CK(hipEventRecord(e0, 0)); for (int i = 0; i < reps; ++i) CK(launch_fused_qkv_attention(f, 0)); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1)); tf.push_back(ms * 1000.f / reps);
    }
    auto med = [](std::vector<float> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    printf("B=%d: qkv GEMM %.1f us + attention %.1f us = %.1f us;  fused %.1f us (debug %d)\n", B, med(tg), med(ta), med(tg) + med(ta), med(tf), f.debug);

    {   // attention alone: row-major q|k|v [M, 3D] against the head-major form (a head's K / V rows contiguous), warm (the same 58 MB
        // re-read) and cold (a 768 MB fill between launches evicts L2 and the Infinity Cache); timing only - head-major reads the
        // same buffer with other strides
        AttnParams hm = ap; hm.ldqkv = 64; hm.head_stride = (int64_t)Mp * 64; hm.which_stride = (int64_t)H * Mp * 64;
        void* junk; const size_t junk_bytes = (size_t)768 << 20; CK(hipMalloc(&junk, junk_bytes));
        for (int cold = 0; cold < 2; ++cold)
            for (int lay = 0; lay < 2; ++lay) {
                const AttnParams& q = lay ? hm : ap;
                std::vector<float> t;
                for (int r = 0; r < 15; ++r) {
                    if (cold) CK(hipMemsetAsync(junk, r, junk_bytes, 0));
                    float ms; CK(hipEventRecord(e0, 0)); CK(launch_attention(q, 0)); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
                    CK(hipEventElapsedTime(&ms, e0, e1)); t.push_back(ms * 1000.f);
                }
                printf("attention alone, %s q|k|v, %s caches: median %.1f us\n", lay ? "head-major" : "row-major ", cold ? "cold" : "warm", med(t));
            }
        CK(hipFree(junk));
    }
    CK(hipMemset(stamps, 0, (size_t)grid * 128 * 8));
    FusedQkvAttnArgs fs = f; fs.stamps = stamps;
    CK(launch_fused_qkv_attention(fs, 0)); CK(hipDeviceSynchronize());
    std::vector<unsigned long long> hst((size_t)grid * 128);
    CK(hipMemcpy(hst.data(), stamps, hst.size() * 8, hipMemcpyDeviceToHost));
    unsigned long long t0 = ~0ull, t1 = 0;
    double ph[4] = {0, 0, 0, 0}; int cnt = 0;
    for (int b = 0; b < grid; ++b) {
        const unsigned long long* q = &hst[(size_t)b * 128];
        if (!q[0] || !q[4]) continue;
        t0 = std::min(t0, q[0]); t1 = std::max(t1, q[4]);
        for (int k = 0; k < 4; ++k) ph[k] += (double)(q[k + 1] - q[k]) * 0.01;
        ++cnt;
    }
    printf("stamps: %d workgroups, span %.1f us; mean per workgroup: statistics %.2f us, K loop %.2f us, epilogue->LDS %.2f us, attention %.2f us (sum %.2f)\n",
           cnt, (double)(t1 - t0) * 0.01, ph[0] / cnt, ph[1] / cnt, ph[2] / cnt, ph[3] / cnt, (ph[0] + ph[1] + ph[2] + ph[3]) / cnt);
    // start-time histogram: how many rounds of workgroups
    std::vector<double> starts;
    for (int b = 0; b < grid; ++b) if (hst[(size_t)b * 128]) starts.push_back((double)(hst[(size_t)b * 128] - t0) * 0.01);
    std::sort(starts.begin(), starts.end());
    printf("start offsets (us): p10 %.1f p30 %.1f p50 %.1f p70 %.1f p90 %.1f max %.1f\n", starts[starts.size() / 10], starts[starts.size() * 3 / 10], starts[starts.size() / 2],
           starts[starts.size() * 7 / 10], starts[starts.size() * 9 / 10], starts.back());
    {   // one K-step (t = 6) per wave, shader clock
        const char* names[6] = {"wait X", "barrier B", "21 MFMA(X) + reads + DMA", "own-DMA + Y wait", "barrier M", "21 MFMA(Y) + reads + DMA"};
        for (int grp = 0; grp < 2; ++grp) {
            double seg[6] = {0}; int n = 0;
            for (int b = 0; b < grid; ++b)
                for (int w = grp * 4; w < grp * 4 + 4; ++w) {
                    const unsigned long long* t = &hst[(size_t)b * 128 + 16 + w * 8];
                    if (!t[0] || !t[6]) continue;
                    ++n;
                    for (int k = 0; k < 6; ++k) seg[k] += (double)(t[k + 1] - t[k]);
                }
            if (!n) continue;
            double tot = 0; for (int k = 0; k < 6; ++k) tot += seg[k] / n;
            printf("K-step stamps, waves %d-%d (%d waves), shader cycles: total %.0f |", grp * 4, grp * 4 + 3, n, tot);
            for (int k = 0; k < 6; ++k) printf(" %s %.0f |", names[k], seg[k] / n);
            printf("\n");
        }
    }
    return 0;
}
