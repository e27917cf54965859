// Standalone A/B microbenchmark of the GEMM tile variants on the ViT shapes (one process, interleaved
// rounds, random data - cdna_hip_programming.md rules 24/25).  Build: see tools/build_tools.sh.
#include "../interactive_vit_amd/csrc/kernels.h"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <map>
#include <vector>
using namespace ivit;
namespace ivit { int gemm_persist_occupancy(int variant); }

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void naive_rows(const bf16_t* A, int lda, const bf16_t* W, int ldw, const float* bias, float* out, int N, int K, const int* rows, int nrows) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    const int r = blockIdx.y;
    if (n >= N || r >= nrows) return;
    const int m = rows[r];
    float acc = 0.f;
    for (int k = 0; k < K; ++k) acc += bf2f(A[(size_t)m * lda + k]) * bf2f(W[(size_t)n * ldw + k]);
    out[(size_t)r * N + n] = acc + bias[n];
}

static bf16_t h_f2bf(float f) { unsigned u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (bf16_t)(u >> 16); }

int main(int argc, char** argv) {
    setvbuf(stdout, nullptr, _IONBF, 0);   // a faulting kernel must not take the log with it
    const int Mfull = argc > 1 ? atoi(argv[1]) : 64 * 197;
    const int only_shape = argc > 2 ? atoi(argv[2]) : -1;      // -1: all shapes
    const unsigned vmask = argc > 3 ? (unsigned)strtoul(argv[3], nullptr, 0) : 0xffffffffu;
    const int rounds = argc > 4 ? atoi(argv[4]) : 7;
    const int debug = argc > 5 ? atoi(argv[5]) : 0;
    struct Shape { const char* name; int M, N, K, epi; };
    std::vector<Shape> shapes = {
        {"qkv   ", Mfull, 2304, 768, EPI_BIAS_BF16}, {"proj  ", Mfull, 768, 768, EPI_BIAS_RESID_F32},
        {"mlp1  ", Mfull, 3072, 768, EPI_BIAS_GELU_BF16}, {"mlp2  ", Mfull, 768, 3072, EPI_BIAS_RESID_F32},
        {"patch ", 64 * 196, 768, 768, EPI_BIAS_F32}, {"mlp1ng", Mfull, 3072, 768, EPI_BIAS_BF16},
        {"projRS", Mfull, 768, 768, EPI_BIAS_RESID_STATS}, {"mlp2RS", Mfull, 768, 3072, EPI_BIAS_RESID_STATS},
        {"qkvLF ", Mfull, 2304, 768, EPI_LNFOLD_BF16}, {"mlp1LF", Mfull, 3072, 768, EPI_LNFOLD_GELU_BF16},
    };
    const char* set = getenv("IVIT_SHAPES");   // "vith": the ViT-H/14 layer shapes (dim 1280, mlp 5120)
    if (set && !strcmp(set, "vith"))
        shapes = {{"h_qkv ", Mfull, 3840, 1280, EPI_BIAS_BF16}, {"h_proj", Mfull, 1280, 1280, EPI_BIAS_RESID_F32},
                  {"h_mlp1", Mfull, 5120, 1280, EPI_BIAS_GELU_BF16}, {"h_mlp2", Mfull, 1280, 5120, EPI_BIAS_RESID_F32}};
    if (set && !strcmp(set, "vitl"))
        shapes = {{"l_qkv ", Mfull, 3072, 1024, EPI_BIAS_BF16}, {"l_proj", Mfull, 1024, 1024, EPI_BIAS_RESID_F32},
                  {"l_mlp1", Mfull, 4096, 1024, EPI_BIAS_GELU_BF16}, {"l_mlp2", Mfull, 1024, 4096, EPI_BIAS_RESID_F32}};
    int maxN = 0, maxK = 0;
    for (const Shape& s : shapes) { maxN = std::max(maxN, s.N); maxK = std::max(maxK, s.K); }
    int maxMs = Mfull;
    for (const Shape& s : shapes) maxMs = std::max(maxMs, s.M);   // (the patch shape has its own M)
    const int maxM = round_up(maxMs, 256) + 256;
    std::mt19937 rng(1);
    std::uniform_real_distribution<float> u(-1.f, 1.f);
    std::vector<bf16_t> hA((size_t)maxM * maxK), hW((size_t)maxN * maxK);
    for (auto& v : hA) v = h_f2bf(u(rng));
    for (auto& v : hW) v = h_f2bf(u(rng) * 0.05f);
    std::vector<float> hb(maxN);
    for (auto& v : hb) v = u(rng);
    bf16_t *dA, *dW; float *db, *dout, *dres, *dref; int* drows;
    CK(hipMalloc(&dA, hA.size() * 2)); CK(hipMalloc(&dW, hW.size() * 2)); CK(hipMalloc(&db, maxN * 4));
    CK(hipMalloc(&dout, (size_t)maxM * maxN * 4)); CK(hipMalloc(&dres, (size_t)maxM * maxN * 4));
    const int NR = 64;
    CK(hipMalloc(&dref, (size_t)NR * maxN * 4)); CK(hipMalloc(&drows, NR * 4));
    CK(hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dW, hW.data(), hW.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, hb.data(), maxN * 4, hipMemcpyHostToDevice));
    CK(hipMemset(dres, 0, (size_t)maxM * maxN * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int stamp_variant = argc > 6 ? atoi(argv[6]) : -1;   // variant whose blocks are time-stamped
    unsigned long long* dstamps = nullptr;
    const int max_blocks = 4096;
    CK(hipMalloc(&dstamps, (size_t)max_blocks * 16 * 8));

    float* dzeros; CK(hipMalloc(&dzeros, maxN * 4)); CK(hipMemset(dzeros, 0, maxN * 4));
    // LayerNorm-fold epilogues: statistics table and bf16 copy
    float2* dpart; bf16_t* dxb;
    CK(hipMalloc(&dpart, (size_t)maxM * GEMM_LN_SLOTS * 8)); CK(hipMalloc(&dxb, (size_t)maxM * maxN * 2));

    // ablation configurations "unused:order[,...]" (IVIT_CFGS); the first one is the baseline.  (The first field was a start
    // delay for the second workgroup of a CU: measured in round 2, no gain - co-resident workgroups de-synchronise by themselves.)
    struct Cfg { int stagger, order; };
    std::vector<Cfg> cfgs;
    {
        const char* cs = getenv("IVIT_CFGS");
        std::string str = cs ? cs : "0:0";
        size_t pos = 0;
        while (pos < str.size()) {
            size_t end = str.find(',', pos); if (end == std::string::npos) end = str.size();
            int a = 0, b = 0; sscanf(str.substr(pos, end - pos).c_str(), "%d:%d", &a, &b);
            cfgs.push_back({a, b}); pos = end + 1;
        }
    }

    // IVIT_PAIR="a,b": two shapes of the list as a kernel PAIR - back to back on one stream against side by side on two streams (round 4:
    // how much of a launch's epilogue / prologue phases another launch's K loops can fill; each shape on its product tile, own buffers)
    if (const char* pr = getenv("IVIT_PAIR")) {
        int ia = 0, ib = 0;
        if (sscanf(pr, "%d,%d", &ia, &ib) != 2 || ia < 0 || ib < 0 || ia >= (int)shapes.size() || ib >= (int)shapes.size()) { printf("IVIT_PAIR=a,b with shape indices\n"); return 1; }
        float *dout2, *dres2; bf16_t* dxb2; float2* dpart2;
        CK(hipMalloc(&dout2, (size_t)maxM * maxN * 4)); CK(hipMalloc(&dres2, (size_t)maxM * maxN * 4)); CK(hipMemset(dres2, 0, (size_t)maxM * maxN * 4));
        CK(hipMalloc(&dxb2, (size_t)maxM * maxN * 2)); CK(hipMalloc(&dpart2, (size_t)maxM * GEMM_LN_SLOTS * 8));
        auto mk = [&](const Shape& s, float* out, float* res, bf16_t* xb, float2* part) {
            GemmParams p{};
            p.A = dA; p.lda = s.K; p.W = dW; p.ldw = s.K; p.M = s.M; p.N = s.N; p.K = s.K; p.bias = db; p.epi = s.epi;
            p.out = out; p.ldo = s.N; p.resid = res; p.ldr = s.N; p.ln_part = part; p.xb = xb; p.ldxb = s.N;
            p.ln_s = (s.epi == EPI_LNFOLD_BF16 || s.epi == EPI_LNFOLD_GELU_BF16) ? db : dzeros; p.ln_eps = 1e-6f; p.ln_dim = s.K;
            return p;
        };
        const GemmParams pa = mk(shapes[ia], dout, dres, dxb, dpart), pb = mk(shapes[ib], dout2, dres2, dxb2, dpart2);
        {   // statistics pairs for the fold epilogues
            GemmParams r = pa; r.epi = EPI_BIAS_RESID_STATS; r.N = 768; r.K = 768; r.lda = r.ldw = 768; r.ldo = r.ldr = r.ldxb = 768; r.ln_s = nullptr;
            CK(launch_gemm_variant(r, GEMM_TILE_160, 0)); r.out = dout2; r.resid = dres2; r.xb = dxb2; r.ln_part = dpart2; CK(launch_gemm_variant(r, GEMM_TILE_160, 0)); CK(hipDeviceSynchronize());
        }
        hipStream_t s1, s2; CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
        hipEvent_t ea, eb, ej; CK(hipEventCreate(&ea)); CK(hipEventCreate(&eb)); CK(hipEventCreateWithFlags(&ej, hipEventDisableTiming));
        const int va = gemm_pick_variant(pa.M, pa.N, pa.K), vb = gemm_pick_variant(pb.M, pb.N, pb.K);
        std::vector<double> serial, side, alone_a, alone_b;
        const int iters = 10;
        for (int round = 0; round < rounds; ++round) {
            auto timed = [&](auto body) { CK(hipDeviceSynchronize()); CK(hipEventRecord(ea, s1)); body(); CK(hipEventRecord(eb, s1)); CK(hipEventSynchronize(eb)); float ms; CK(hipEventElapsedTime(&ms, ea, eb)); return (double)ms / iters * 1e3; };
            alone_a.push_back(timed([&] { for (int i = 0; i < iters; ++i) CK(launch_gemm_variant(pa, va, s1)); }));
            alone_b.push_back(timed([&] { for (int i = 0; i < iters; ++i) CK(launch_gemm_variant(pb, vb, s1)); }));
            serial.push_back(timed([&] { for (int i = 0; i < iters; ++i) { CK(launch_gemm_variant(pa, va, s1)); CK(launch_gemm_variant(pb, vb, s1)); } }));
            side.push_back(timed([&] {
                for (int i = 0; i < iters; ++i) {   // fork: b on s2 beside a on s1; join before the next pair
                    CK(hipEventRecord(ej, s1)); CK(hipStreamWaitEvent(s2, ej, 0));
                    CK(launch_gemm_variant(pa, va, s1)); CK(launch_gemm_variant(pb, vb, s2));
                    CK(hipEventRecord(ej, s2)); CK(hipStreamWaitEvent(s1, ej, 0));
                }
            }));
        }
        auto med = [](std::vector<double>& v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
        printf("pair %s (%s) + %s (%s): alone %.2f + %.2f = %.2f us; back to back %.2f us; side by side on two streams %.2f us\n", shapes[ia].name, gemm_variant_name(va),
               shapes[ib].name, gemm_variant_name(vb), med(alone_a), med(alone_b), med(alone_a) + med(alone_b), med(serial), med(side));
        return 0;
    }

    int shape_idx = -1;
    for (const Shape& s : shapes) {
        ++shape_idx;
        if (only_shape >= 0 && shape_idx != only_shape) continue;
        GemmParams p{};
        p.A = dA; p.lda = s.K; p.W = dW; p.ldw = s.K; p.M = s.M; p.N = s.N; p.K = s.K; p.bias = db; p.epi = s.epi;
        p.out = dout; p.ldo = s.N; p.resid = dres; p.ldr = s.N; p.debug = debug;
        p.ln_part = dpart; p.xb = dxb; p.ldxb = s.N;
        p.ln_s = (s.epi == EPI_LNFOLD_BF16 || s.epi == EPI_LNFOLD_GELU_BF16) ? db : dzeros; p.ln_eps = 1e-6f; p.ln_dim = s.K;   // EPI_LNFOLD_*: statistics folded from dpart
        // correctness on sampled rows (epilogue F32 so the values are comparable)
        std::vector<int> rows(NR);
        for (int i = 0; i < NR; ++i) rows[i] = (int)((long long)i * (s.M - 1) / (NR - 1));
        CK(hipMemcpy(drows, rows.data(), NR * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(naive_rows, dim3(ceil_div(s.N, 256), NR), dim3(256), 0, 0, dA, s.K, dW, s.K, db, dref, s.N, s.K, drows, NR);
        std::vector<float> href((size_t)NR * s.N), hout;
        CK(hipMemcpy(href.data(), dref, href.size() * 4, hipMemcpyDeviceToHost));
        printf("%s M=%d N=%d K=%d pick=%s\n", s.name, s.M, s.N, s.K, gemm_variant_name(gemm_pick_variant(s.M, s.N, s.K)));
        double best[GEMM_VARIANTS]; std::vector<double> times[GEMM_VARIANTS];
        for (int v = 0; v < GEMM_VARIANTS; ++v) {
            if (!((vmask >> v) & 1)) continue;
            if (v == GEMM_TILE_P160 || v == GEMM_TILE_P128) {   // persistent LayerNorm-fold kernels: the shape's own epilogue BITWISE against the 160x128 kernel
                if (!gemm_persist_supported(p)) { printf("   %-28s not applicable to this shape\n", gemm_variant_name(v)); continue; }
                std::vector<bf16_t> ref16((size_t)s.M * s.N), got16((size_t)s.M * s.N);
                {   // real statistics pairs: leave them with the residual epilogue of the 160x128 kernel on another buffer
                    GemmParams r = p; r.epi = EPI_BIAS_RESID_STATS; r.N = s.K; r.K = 768; r.out = dres; r.ldo = s.K; r.resid = dres; r.ldr = s.K; r.xb = dxb; r.ldxb = s.K; r.ln_part = dpart;
                    r.ln_s = nullptr; r.lda = r.K; r.ldw = r.K; r.debug = 0;
                    CK(launch_gemm_variant(r, GEMM_TILE_160, 0));
                    CK(hipDeviceSynchronize());
                }
                GemmParams a = p; a.debug = 0;
                CK(launch_gemm_variant(a, GEMM_TILE_160, 0));
                CK(hipDeviceSynchronize());
                CK(hipMemcpy(ref16.data(), dout, ref16.size() * 2, hipMemcpyDeviceToHost));
                size_t worst = 0;
                for (int rep = 0; rep < 3; ++rep) {   // three launches: a hand-off that only sometimes loses a race must not pass on one lucky run
                    CK(hipMemset(dout, 0xff, (size_t)s.M * s.N * 2));
                    CK(launch_gemm_variant(a, v, 0));
                    CK(hipDeviceSynchronize());
                    CK(hipMemcpy(got16.data(), dout, got16.size() * 2, hipMemcpyDeviceToHost));
                    size_t bad = 0, first = 0;
                    for (size_t i = 0; i < got16.size(); ++i) if (got16[i] != ref16[i]) { if (!bad) first = i; ++bad; }
                    if (bad) printf("   %-28s rep %d: %zu of %zu outputs differ; first at row %zu col %zu: %04x vs %04x\n", gemm_variant_name(v), rep, bad, got16.size(), first / s.N, first % s.N, got16[first], ref16[first]);
                    worst = std::max(worst, bad);
                }
                printf("   %-28s own epilogue vs 160x128: %s\n", gemm_variant_name(v), worst ? "MISMATCH" : "bit-identical (3 launches)");
                continue;
            }
            if (v == GEMM_TILE_PE) {   // 16-bit outputs only: bias epilogue against the naive rows, then the shape's own epilogue BITWISE against the 160x128 kernel
                GemmParams q = p; q.epi = EPI_BIAS_BF16; q.debug = 0; q.ln_s = dzeros; q.ldo = s.N;
                if (!gemm_pe_supported(q)) { printf("   %-28s not applicable to this shape\n", gemm_variant_name(v)); continue; }
                bf16_t* o16 = reinterpret_cast<bf16_t*>(dout);
                CK(hipMemset(dout, 0xff, (size_t)s.M * s.N * 2));
                CK(launch_gemm_variant(q, v, 0));
                CK(hipDeviceSynchronize());
                double maxerr = 0, maxref = 0;
                std::vector<bf16_t> h16((size_t)s.N);
                for (int i = 0; i < NR; ++i) {
                    CK(hipMemcpy(h16.data(), o16 + (size_t)rows[i] * s.N, s.N * 2, hipMemcpyDeviceToHost));
                    for (int n = 0; n < s.N; ++n) { unsigned u = (unsigned)h16[n] << 16; float f; memcpy(&f, &u, 4); maxerr = std::max(maxerr, (double)fabsf(f - href[(size_t)i * s.N + n])); maxref = std::max(maxref, (double)fabsf(href[(size_t)i * s.N + n])); }
                }
                printf("   %-28s check (bf16 out): max err %.3e (max ref %.3f) %s\n", gemm_variant_name(v), maxerr, maxref, maxerr <= 6e-3 * maxref ? "OK" : "MISMATCH");
                const bool own = s.epi == EPI_BIAS_BF16 || s.epi == EPI_BIAS_GELU_BF16 || s.epi == EPI_LNFOLD_BF16 || s.epi == EPI_LNFOLD_GELU_BF16;
                if (own && gemm_pe_supported(p)) {
                    std::vector<bf16_t> ref16((size_t)s.M * s.N), got16((size_t)s.M * s.N);
                    if (s.epi >= EPI_LNFOLD_BF16) {   // real statistics pairs: leave them with the residual epilogue of the 160x128 kernel on another buffer
                        GemmParams r = p; r.epi = EPI_BIAS_RESID_STATS; r.N = s.K; r.K = 768; r.out = dres; r.ldo = s.K; r.resid = dres; r.ldr = s.K; r.xb = dxb; r.ldxb = s.K; r.ln_part = dpart;
                        r.ln_s = nullptr; r.lda = r.K; r.ldw = r.K;
                        CK(launch_gemm_variant(r, GEMM_TILE_160, 0));
                        CK(hipDeviceSynchronize());
                    }
                    GemmParams a = p; a.debug = 0;
                    CK(launch_gemm_variant(a, GEMM_TILE_160, 0));
                    CK(hipDeviceSynchronize());
                    CK(hipMemcpy(ref16.data(), dout, ref16.size() * 2, hipMemcpyDeviceToHost));
                    CK(hipMemset(dout, 0xff, (size_t)s.M * s.N * 2));
                    CK(launch_gemm_variant(a, GEMM_TILE_PE, 0));
                    CK(hipDeviceSynchronize());
                    CK(hipMemcpy(got16.data(), dout, got16.size() * 2, hipMemcpyDeviceToHost));
                    size_t bad = 0, first = 0;
                    for (size_t i = 0; i < got16.size(); ++i) if (got16[i] != ref16[i]) { if (!bad) first = i; ++bad; }
                    printf("   %-28s own epilogue vs 160x128: %zu of %zu outputs differ%s", gemm_variant_name(v), bad, got16.size(), bad ? "" : " (bit-identical)\n");
                    if (bad) printf("; first at row %zu col %zu: %04x vs %04x\n", first / s.N, first % s.N, got16[first], ref16[first]);
                }
                continue;
            }
            GemmParams q = p; q.epi = EPI_BIAS_F32; q.debug = 0;
            CK(hipMemset(dout, 0xff, (size_t)s.M * s.N * 4));
            CK(launch_gemm_variant(q, v, 0));
            CK(hipDeviceSynchronize());
            double maxerr = 0, maxref = 0;
            hout.resize((size_t)s.N);
            for (int i = 0; i < NR; ++i) {
                CK(hipMemcpy(hout.data(), dout + (size_t)rows[i] * s.N, s.N * 4, hipMemcpyDeviceToHost));
                for (int n = 0; n < s.N; ++n) { maxerr = std::max(maxerr, (double)fabsf(hout[n] - href[(size_t)i * s.N + n])); maxref = std::max(maxref, (double)fabsf(href[(size_t)i * s.N + n])); }
            }
            printf("   %-28s check: max err %.3e (max ref %.3f) %s\n", gemm_variant_name(v), maxerr, maxref, maxerr <= 1e-3 * maxref ? "OK" : "MISMATCH");
        }
        if (s.epi == EPI_BIAS_RESID_STATS) {   // the statistics and the bf16 copy must not depend on the tile shape: bitwise
            std::vector<unsigned long long> base, cur((size_t)s.M * GEMM_LN_SLOTS);
            std::vector<unsigned short> xbase, xcur((size_t)s.M * s.N);
            for (int v : {GEMM_TILE_160, GEMM_TILE_128, GEMM_TILE_256S, GEMM_TILE_64D, GEMM_TILE_160SB, GEMM_TILE_128SB}) {
                CK(hipMemset(dpart, 0, (size_t)s.M * GEMM_LN_SLOTS * 8));
                CK(launch_gemm_variant(p, v, 0));
                CK(hipDeviceSynchronize());
                CK(hipMemcpy(cur.data(), dpart, cur.size() * 8, hipMemcpyDeviceToHost));
                CK(hipMemcpy(xcur.data(), dxb, xcur.size() * 2, hipMemcpyDeviceToHost));
                if (base.empty()) { base = cur; xbase = xcur; continue; }
                size_t bad = 0, xbad = 0, first = (size_t)-1;
                for (size_t i = 0; i < cur.size(); ++i) if (cur[i] != base[i]) { if (!bad) first = i; ++bad; }
                for (size_t i = 0; i < xcur.size(); ++i) if (xcur[i] != xbase[i]) ++xbad;
                printf("   %-28s row statistics vs 160x128: %zu of %zu pairs differ%s; bf16 copy: %zu differ\n", gemm_variant_name(v), bad, cur.size(),
                       bad ? "" : " (bit-identical)", xbad);
                if (bad) {
                    float a[2], b[2]; memcpy(a, &base[first], 8); memcpy(b, &cur[first], 8);
                    printf("      first: row %zu slot %zu: (%.9g, %.9g) vs (%.9g, %.9g)\n", first / GEMM_LN_SLOTS, first % GEMM_LN_SLOTS, a[0], a[1], b[0], b[1]);
                }
            }
        }
        for (int vb : {GEMM_TILE_64D, GEMM_TILE_160SB, GEMM_TILE_128SB}) if ((vmask >> vb) & 1) {   // product tiles: the shape's own epilogue BITWISE against the two-stage 160x128 kernel
            const bool out16 = s.epi == EPI_BIAS_BF16 || s.epi == EPI_BIAS_GELU_BF16 || s.epi == EPI_LNFOLD_BF16 || s.epi == EPI_LNFOLD_GELU_BF16;
            const size_t bytes = (size_t)s.M * s.N * (out16 ? 2 : 4);
            std::vector<unsigned char> ref(bytes), got(bytes);
            GemmParams a = p; a.debug = 0;
            if (a.epi == EPI_BIAS_RESID_F32 || a.epi == EPI_BIAS_RESID_STATS) { a.out = dout; a.resid = dres; }   // out of place: both runs read the same residual
            CK(launch_gemm_variant(a, GEMM_TILE_160, 0)); CK(hipDeviceSynchronize());
            CK(hipMemcpy(ref.data(), dout, bytes, hipMemcpyDeviceToHost));
            CK(hipMemset(dout, 0xff, bytes));
            if (a.epi >= EPI_LNFOLD_BF16) {   // real statistics pairs (left by a residual epilogue on another buffer)
                GemmParams r = p; r.epi = EPI_BIAS_RESID_STATS; r.N = s.K; r.K = 768; r.out = dres; r.ldo = s.K; r.resid = dres; r.ldr = s.K; r.xb = dxb; r.ldxb = s.K; r.ln_part = dpart;
                r.ln_s = nullptr; r.lda = r.K; r.ldw = r.K; r.debug = 0;
                CK(launch_gemm_variant(r, GEMM_TILE_160, 0)); CK(hipDeviceSynchronize());
                CK(launch_gemm_variant(a, GEMM_TILE_160, 0)); CK(hipDeviceSynchronize());
                CK(hipMemcpy(ref.data(), dout, bytes, hipMemcpyDeviceToHost));
                CK(hipMemset(dout, 0xff, bytes));
            }
            CK(launch_gemm_variant(a, vb, 0)); CK(hipDeviceSynchronize());
            CK(hipMemcpy(got.data(), dout, bytes, hipMemcpyDeviceToHost));
            size_t bad = 0; for (size_t i = 0; i < bytes; ++i) bad += got[i] != ref[i];
            printf("   %-28s own epilogue vs 160x128: %zu of %zu output bytes differ%s\n", gemm_variant_name(vb), bad, bytes, bad ? "" : " (bit-identical)");
        }
        std::vector<std::vector<double>> ctimes(cfgs.size() * GEMM_VARIANTS);
        for (int round = 0; round < rounds; ++round)
            for (size_t ci = 0; ci < cfgs.size(); ++ci)
            for (int v = 0; v < GEMM_VARIANTS; ++v) {
                if (!((vmask >> v) & 1)) continue;
                if (s.epi >= EPI_BIAS_RESID_STATS && v != GEMM_TILE_128 && v != GEMM_TILE_160 && v != GEMM_TILE_256S && v != GEMM_TILE_PE && v != GEMM_TILE_64D && v != GEMM_TILE_P160 && v != GEMM_TILE_P128 && v < GEMM_TILE_128W8A) continue;
                if (s.epi == EPI_BIAS_RESID_STATS && (v == GEMM_TILE_128W8A || v == GEMM_TILE_160W8)) continue;
                if (v == GEMM_TILE_PE && !gemm_pe_supported(p)) continue;
                if ((v == GEMM_TILE_P160 || v == GEMM_TILE_P128) && !gemm_persist_supported(p)) continue;
                if (ci > 0 && v != GEMM_TILE_128 && v != GEMM_TILE_160) continue;   // the ablation knobs live in gemm_body
                if (cfgs[ci].order == 2 && (v != GEMM_TILE_160 || ceil_div(s.M, 160) * ceil_div(s.N, 128) > 512 || (ceil_div(s.N, 128) & 1))) continue;
                GemmParams q = p; q.order = cfgs[ci].order;
                const int iters = rounds >= 7 ? 10 : 2;
                CK(launch_gemm_variant(q, v, 0));
                CK(hipEventRecord(e0, 0));
                for (int i = 0; i < iters; ++i) CK(launch_gemm_variant(q, v, 0));
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (ci == 0) times[v].push_back(ms / iters);
                ctimes[ci * GEMM_VARIANTS + v].push_back(ms / iters);
            }
        for (size_t ci = 1; ci < cfgs.size(); ++ci)
            for (int v = 0; v < GEMM_VARIANTS; ++v) {
                auto& tv = ctimes[ci * GEMM_VARIANTS + v];
                if (tv.empty()) continue;
                std::sort(tv.begin(), tv.end());
                const double fl = 2.0 * s.M * s.N * s.K, med = tv[tv.size() / 2];
                printf("   %-28s stagger %4d order %d: median %8.2f us  %7.1f TF/s   (min %8.2f us)\n", gemm_variant_name(v), cfgs[ci].stagger, cfgs[ci].order, med * 1e3, fl / med / 1e9, tv[0] * 1e3);
            }
        if (stamp_variant == GEMM_TILE_PE && gemm_pe_supported(p)) {   // shader-clock stamps of one plain K-step (tile 1, K-step nt-2) per workgroup
            CK(hipMemset(dstamps, 0, (size_t)max_blocks * 16 * 8));
            GemmParams q = p; q.stamps = dstamps;   // (p.debug = argv[5]: 1 = no DMA in the loop, 2 = no MFMA)
            for (int i = 0; i < 3; ++i) CK(launch_gemm_variant(p, GEMM_TILE_PE, 0));
            CK(hipDeviceSynchronize());
            CK(launch_gemm_variant(q, GEMM_TILE_PE, 0));
            CK(hipDeviceSynchronize());
            std::vector<unsigned long long> hs((size_t)256 * 128);
            CK(hipMemcpy(hs.data(), dstamps, hs.size() * 8, hipMemcpyDeviceToHost));
            {   // block -> XCD placement: the persistent walk assumes blocks b and b + 8 share an XCD
                int agree = 0, seen = 0, per_xcd[16] = {0};
                int first_xcd_of_class[8]; for (int i = 0; i < 8; ++i) first_xcd_of_class[i] = -1;
                for (int b = 0; b < 256; ++b) {
                    const int x = (int)hs[(size_t)b * 128 + 15] - 1;
                    if (x < 0) continue;
                    ++seen; ++per_xcd[x];
                    if (first_xcd_of_class[b & 7] < 0) first_xcd_of_class[b & 7] = x;
                    if (first_xcd_of_class[b & 7] == x) ++agree;
                }
                printf("   block placement: %d blocks, %d on the XCD of their (blockIdx %% 8) class; per XCD:", seen, agree);
                for (int i = 0; i < 8; ++i) printf(" %d", per_xcd[i]);
                printf("\n");
            }
            const char* names[9] = {"SR: DMA issue + 16 reads (+ loads)", "counted wait", "barrier 1", "M: 32 MFMA", "barrier 2", "-", "-", "-", "-"};   // a fat K-step (stamps 0..5)
            for (int grp = 0; grp < 2; ++grp) {
                double seg[9] = {0}; int cnt = 0;
                for (int b = 0; b < 256; ++b)
                    for (int w = grp * 4; w < grp * 4 + 4; ++w) {
                        const unsigned long long* t = &hs[(size_t)b * 128 + w * 16];
                        if (!t[0] || !t[5]) continue;
                        ++cnt;
                        for (int k = 0; k < 5; ++k) seg[k] += (double)(t[k + 1] - t[k]);
                    }
                if (!cnt) continue;
                double tot = 0; for (int k = 0; k < 5; ++k) tot += seg[k] / cnt;
                printf("   PE fat K-step stamps, waves %d-%d (%d waves), shader cycles: total %.0f |", grp * 4, grp * 4 + 3, cnt, tot);
                for (int k = 0; k < 5; ++k) printf(" %s %.0f |", names[k], seg[k] / cnt);
                printf("\n");
            }
        } else if (stamp_variant == GEMM_TILE_128 || stamp_variant == GEMM_TILE_160 || ((stamp_variant == GEMM_TILE_P160 || stamp_variant == GEMM_TILE_P128) && gemm_persist_supported(p))) {   // gemm_body: per-CU timelines (who overlaps whom); persistent kernels: slot 1 = first staging issued, 2 = all tiles multiplied
          if (stamp_variant >= GEMM_TILE_P160) printf("   occupancy query (workgroups per CU): %d\n", gemm_persist_occupancy(stamp_variant));
          for (size_t ci = 0; ci < cfgs.size(); ++ci) {
            CK(hipMemset(dstamps, 0, (size_t)max_blocks * 16 * 8));
            GemmParams q = p; q.stamps = dstamps; q.order = cfgs[ci].order;
            GemmParams qw = q; qw.stamps = nullptr;
            for (int i = 0; i < 3; ++i) CK(launch_gemm_variant(qw, stamp_variant, 0));
            CK(hipDeviceSynchronize());
            CK(launch_gemm_variant(q, stamp_variant, 0));
            CK(hipDeviceSynchronize());
            std::vector<unsigned long long> hs((size_t)max_blocks * 16);
            CK(hipMemcpy(hs.data(), dstamps, hs.size() * 8, hipMemcpyDeviceToHost));
            unsigned long long tmin = ~0ull, tmax = 0; int nb = 0;
            for (int b = 0; b < max_blocks; ++b) if (hs[(size_t)b * 16]) { tmin = std::min(tmin, hs[(size_t)b * 16]); tmax = std::max(tmax, hs[(size_t)b * 16 + 4]); nb = b + 1; }
            std::map<unsigned long long, std::vector<int>> by_cu;
            double seg[5] = {0, 0, 0, 0, 0}; int cnt = 0;
            for (int b = 0; b < nb; ++b) {
                const unsigned long long* t = &hs[(size_t)b * 16];
                if (!t[0]) continue;
                ++cnt;
                for (int k = 1; k < 5; ++k) seg[k] += (double)(t[k] - t[k - 1]) * 0.01;
                by_cu[t[5] & 0xf0000ff00ull].push_back(b);
            }
            // per CU: how much of every block's epilogue [t2, t4] lies inside another block's K loop [t1, t2] on the same CU
            double epi_total = 0, epi_covered = 0, k_total = 0, k_shared = 0;
            size_t max_res = 0;
            for (auto& kv : by_cu) {
                max_res = std::max(max_res, kv.second.size());
                for (int a : kv.second) {
                    const unsigned long long* ta = &hs[(size_t)a * 16];
                    epi_total += (double)(ta[4] - ta[2]);
                    k_total += (double)(ta[2] - ta[1]);
                    for (int b : kv.second) {
                        if (a == b) continue;
                        const unsigned long long* tb = &hs[(size_t)b * 16];
                        const long long lo = (long long)std::max(ta[2], tb[1]), hi = (long long)std::min(ta[4], tb[2]);
                        if (hi > lo) epi_covered += (double)(hi - lo);
                        const long long lo2 = (long long)std::max(ta[1], tb[1]), hi2 = (long long)std::min(ta[2], tb[2]);
                        if (hi2 > lo2) k_shared += (double)(hi2 - lo2);
                    }
                }
            }
            printf("   stamps %s stagger %d order %d: %d blocks on %zu CUs (max %zu per CU); kernel span %.2f us\n", gemm_variant_name(stamp_variant), cfgs[ci].stagger, cfgs[ci].order,
                   cnt, by_cu.size(), max_res, (double)(tmax - tmin) * 0.01);
            printf("      per block: start->K loop %.2f us | K loop %.2f us | epilogue issue %.2f us | store drain %.2f us\n", seg[1] / cnt, seg[2] / cnt, seg[3] / cnt, seg[4] / cnt);
            printf("      epilogue time inside another workgroup's K loop on the same CU: %.1f %%; K-loop time shared with another K loop: %.1f %%\n",
                   100.0 * epi_covered / epi_total, 100.0 * k_shared / k_total);
            if (stamp_variant >= GEMM_TILE_P160) {   // shader cycles of wave 0 per section of the K loop, mean over workgroups
                const char* sec[6] = {"counted wait", "barrier", "fold + DMA issue", "reads + MFMA", "slice", "statistics DMA + loop"};
                double sum[6] = {0}; int n = 0;
                for (int b = 0; b < nb; ++b) { const unsigned long long* t = &hs[(size_t)b * 16]; if (!t[0]) continue; ++n; for (int k = 0; k < 6; ++k) sum[k] += (double)t[8 + k]; }
                double tot = 0; for (int k = 0; k < 6; ++k) tot += sum[k] / n;
                printf("      wave 0 shader cycles per workgroup: total %.0f |", tot);
                for (int k = 0; k < 6; ++k) printf(" %s %.0f (%.0f %%) |", sec[k], sum[k] / n, 100.0 * sum[k] / n / tot);
                printf("\n");
            }
            // one CU's timeline
            const auto& one = by_cu.begin()->second;
            for (int b : one) {
                const unsigned long long* t = &hs[(size_t)b * 16];
                printf("      cu %llx block %4d slot %llu: start %7.2f  kloop %7.2f  epi %7.2f  end %7.2f\n", (unsigned long long)(t[5] & 0xf0000ff00ull), b, (t[5] >> 40) & 0xff,
                       (double)(t[0] - tmin) * 0.01, (double)(t[1] - tmin) * 0.01, (double)(t[2] - tmin) * 0.01, (double)(t[4] - tmin) * 0.01);
            }
          }
        } else if (stamp_variant >= 0) {
            CK(hipMemset(dstamps, 0, (size_t)max_blocks * 16 * 8));
            GemmParams q = p; q.stamps = dstamps;
            for (int i = 0; i < 3; ++i) CK(launch_gemm_variant(p, stamp_variant, 0));   // warm
            CK(hipDeviceSynchronize());
            CK(launch_gemm_variant(q, stamp_variant, 0));
            CK(hipDeviceSynchronize());
            std::vector<unsigned long long> hs((size_t)max_blocks * 16);
            CK(hipMemcpy(hs.data(), dstamps, hs.size() * 8, hipMemcpyDeviceToHost));
            unsigned long long tmin = ~0ull; int nb = 0;
            for (int b = 0; b < max_blocks; ++b) if (hs[(size_t)b * 16]) { tmin = std::min(tmin, hs[(size_t)b * 16]); nb = b + 1; }
            double seg[2][5] = {{0}}, startsum = 0, endmax = 0; int cnt = 0;
            for (int b = 0; b < nb; ++b) {
                if (!hs[(size_t)b * 16]) continue;
                ++cnt;
                for (int g = 0; g < 2; ++g) {
                    const unsigned long long* t = &hs[((size_t)b * 2 + g) * 8];
                    for (int k = 1; k < 5; ++k) seg[g][k] += (double)(t[k] - t[k - 1]) * 0.01;   // 100 MHz -> us
                }
                startsum += (double)(hs[(size_t)b * 16] - tmin) * 0.01;
                endmax = std::max(endmax, (double)(hs[(size_t)b * 16 + 4] - tmin) * 0.01);
            }
            printf("   stamps %s: %d blocks; mean start offset %.2f us, last end %.2f us\n", gemm_variant_name(stamp_variant), cnt, startsum / cnt, endmax);
            for (int g = 0; g < 2; ++g)
                printf("      group %d: prologue %.2f us | K loop %.2f us | epilogue issue %.2f us | drain %.2f us\n", g,
                       seg[g][1] / cnt, seg[g][2] / cnt, seg[g][3] / cnt, seg[g][4] / cnt);
            if (stamp_variant == GEMM_TILE_160X256W4) {   // this kernel also stamps the shader clock around its K loop
                double cyc = 0, us = 0;
                for (int b = 0; b < nb; ++b) {
                    if (!hs[(size_t)b * 16]) continue;
                    cyc += (double)(hs[(size_t)b * 16 + 10] - hs[(size_t)b * 16 + 9]);
                    us += (double)(hs[(size_t)b * 16 + 2] - hs[(size_t)b * 16 + 1]) * 0.01;
                }
                double fine[4] = {0, 0, 0, 0};
                for (int b = 0; b < nb; ++b) {
                    if (!hs[(size_t)b * 16]) continue;
                    for (int k = 0; k < 4; ++k) fine[k] += (double)(hs[(size_t)b * 16 + 12 + k] - hs[(size_t)b * 16 + 11 + k]);
                }
                printf("      K-step 4, wave 0 (shader cycles): waitcnt %.0f | barrier %.0f | 13 DMA issue %.0f | 80 MFMA + 26 ds_read %.0f\n",
                       fine[0] / cnt, fine[1] / cnt, fine[2] / cnt, fine[3] / cnt);
                printf("      K loop: %.0f shader cycles per block = %.0f per K-step; shader clock %.0f MHz\n", cyc / cnt, cyc / cnt / (s.K / 64), cyc / us);
            }
            // histogram of block start times (rounds)
            int late_blocks = 0;
            for (int b = 0; b < nb; ++b) if (hs[(size_t)b * 16] && (double)(hs[(size_t)b * 16] - tmin) * 0.01 > 5.0) ++late_blocks;
            printf("      blocks starting > 5 us after the first: %d\n", late_blocks);
        }
        for (int v = 0; v < GEMM_VARIANTS; ++v) {
            if (!((vmask >> v) & 1) || times[v].empty()) continue;
            std::sort(times[v].begin(), times[v].end());
            const double med = times[v][times[v].size() / 2], mn = times[v][0];
            best[v] = med;
            const double fl = 2.0 * s.M * s.N * s.K;
            printf("   %-28s median %8.2f us  %7.1f TF/s   (min %8.2f us %7.1f TF/s)\n", gemm_variant_name(v), med * 1e3, fl / med / 1e9, mn * 1e3, fl / mn / 1e9);
        }
    }
    return 0;
}
