// Lab harness (not product): k_gemm_slab against k_gemm_stream on the same operands -- time and column sums, for
// per-column partials and per-workgroup accumulators.  Build: make -C tools/lab slab_lab
#include "../../adalog_amd/csrc/gemm_score.hip"
#include <vector>
#include <algorithm>
#include <string.h>

static char g_err[512];
extern "C" void adalog_set_error(const char* where, hipError_t e) { snprintf(g_err, sizeof g_err, "%s: %s", where, hipGetErrorString(e)); }
extern "C" void adalog_set_error_msg(const char* msg) { snprintf(g_err, sizeof g_err, "%s", msg); }
extern "C" void adalog_note_kernel(const char*) {}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

struct Case { const char* name; int M, N, K, P; int rows; int dtype = 0; };

int main(int argc, char** argv) {
    if (argc > 1) setenv("ADALOG_GEMM_SLAB_MINM", argv[1], 1);   // e.g. 384: let the slab kernel take attn.proj's activation search
    const int T = 32 * 197;
    std::vector<Case> cases = {
        {"w-search qkv  i8 K=384", T, 1152 * 128, 384, 128, 0},
        {"w-search qkv fp8 K=384", T, 1152 * 128, 384, 128, 0, 3},
        {"a-search qkv  i8 K=384", 1152, T * 128, 384, 128, 1},
        {"a-search qkv fp8 K=384", 1152, T * 128, 384, 128, 1, 3},
        {"a-search fc1  i8 K=384", 1536, T * 128, 384, 128, 1},
        {"a-search proj i8 K=384", 384, T * 128, 384, 128, 1},
        {"w-search fc1  i8 K=384", T, 1536 * 128, 384, 128, 0},
        {"w-search proj i8 K=384", T, 384 * 128, 384, 128, 0},
        {"w-search tiny i8 K=192", T, 576 * 128, 192, 128, 0},
        {"w-search P=64  K=384", T, 1151 * 64, 384, 64, 0},
        {"a-search P=256 K=384", 1152, 1001 * 256, 384, 256, 1},
    };
    for (const Case& cs : cases) {
        const size_t abytes = (size_t)cs.M * cs.K, bbytes = (size_t)cs.N * cs.K;
        uint8_t *A, *B; float *ref, *sa, *sb, *rs, *rb, *bias;
        CK(hipMalloc(&A, abytes)); CK(hipMalloc(&B, bbytes));
        std::vector<uint8_t> h(std::max(abytes, bbytes));
        // integers -8..7, as int8 or as their e4m3 encodings
        static const uint8_t f8[16] = {0xD0, 0xCE, 0xCC, 0xCA, 0xC8, 0xC4, 0xC0, 0xB8, 0x00, 0x38, 0x40, 0x44, 0x48, 0x4A, 0x4C, 0x4E};
        auto enc = [&](unsigned v) { return cs.dtype == 3 ? f8[v & 15] : (uint8_t)((v & 15) - 8); };
        for (size_t i = 0; i < h.size(); ++i) h[i] = enc((unsigned)(i * 2654435761u >> 13));
        CK(hipMemcpy(A, h.data(), abytes, hipMemcpyHostToDevice));
        for (size_t i = 0; i < h.size(); ++i) h[i] = enc((unsigned)(i * 40503u >> 7));
        CK(hipMemcpy(B, h.data(), bbytes, hipMemcpyHostToDevice));
        const int n_eff = cs.N / cs.P;
        std::vector<float> hr((size_t)cs.M * n_eff);
        for (size_t i = 0; i < hr.size(); ++i) hr[i] = (float)((int)((i * 2246822519u >> 11) & 255) - 128) * 0.25f;
        CK(hipMalloc(&ref, hr.size() * 4)); CK(hipMemcpy(ref, hr.data(), hr.size() * 4, hipMemcpyHostToDevice));
        CK(hipMalloc(&sa, 4 * 256)); CK(hipMalloc(&sb, (size_t)cs.N * 4)); CK(hipMalloc(&rs, cs.M * 4)); CK(hipMalloc(&rb, cs.M * 4));
        CK(hipMalloc(&bias, (size_t)n_eff * 4));
        std::vector<float> f(std::max(cs.N, cs.M));
        for (size_t i = 0; i < f.size(); ++i) f[i] = 0.01f + 0.001f * (float)(i % 37);
        CK(hipMemcpy(sa, f.data(), 4 * 256, hipMemcpyHostToDevice)); CK(hipMemcpy(sb, f.data(), (size_t)cs.N * 4, hipMemcpyHostToDevice));
        for (size_t i = 0; i < f.size(); ++i) f[i] = 0.5f + 0.01f * (float)(i % 53);
        CK(hipMemcpy(rs, f.data(), cs.M * 4, hipMemcpyHostToDevice));
        for (size_t i = 0; i < f.size(); ++i) f[i] = 0.125f * (float)(i % 11);
        CK(hipMemcpy(rb, f.data(), cs.M * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(bias, f.data(), (size_t)n_eff * 4, hipMemcpyHostToDevice));
        const double flop = 2.0 * cs.M * (double)cs.N * cs.K;
        std::vector<double> sums[2][2];
        for (int red = 0; red < 2; ++red) {              // 0: per-column partials, 1: per-workgroup accumulators
            for (int slab = 0; slab < 2; ++slab) {
                g_slab_override = slab;
                int MT, Npad, mode;
                const int64_t pe = adalog_gemm_score_layout(cs.M, cs.N, 1, 1, 1, cs.P, red, cs.dtype, cs.K, cs.K, 1, &MT, &Npad, &mode);
                float* partial; CK(hipMalloc(&partial, pe * 4)); CK(hipMemset(partial, 0xff, pe * 4));
                auto run = [&]() {
                    int rc = adalog_gemm_score(cs.dtype, A, B, 0, 0, 0, 0, cs.M, cs.N, cs.K, 0, 1, 1, 1, ref, 1, 0, cs.M, cs.P,
                                               sa, cs.rows ? 1 : 0, 0, 1.0f, sb, cs.rows ? 0 : n_eff, 0, cs.rows ? 0 : 1,
                                               cs.rows ? nullptr : bias, 0, 0, cs.rows ? 0 : 1,
                                               cs.rows ? rs : nullptr, cs.rows ? rb : nullptr, partial, pe, nullptr, 0, 0, 0, 2, red, nullptr);
                    if (rc) { printf("gemm error: %s\n", g_err); exit(1); }
                };
                hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
                run(); run(); CK(hipDeviceSynchronize());
                CK(hipEventRecord(e0)); for (int i = 0; i < 5; ++i) run(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
                printf("%-24s %s %s: MT %3d mode %d  %.3f ms  %.1f TFLOP/s\n", cs.name, red ? "acc " : "cols", slab ? "slab  " : "stream", MT, mode, ms, flop / ms * 1e-9);
                if (slab && red) {                       // cycle stamps of the slab kernel: effective clock, share spent switching slabs
                    long long* tl; CK(hipMalloc(&tl, 4 * 8192 * sizeof(long long))); CK(hipMemset(tl, 0, 4 * 8192 * sizeof(long long)));
                    g_timeline = tl;
                    hipEvent_t e2, e3; CK(hipEventCreate(&e2)); CK(hipEventCreate(&e3));
                    CK(hipEventRecord(e2)); run(); CK(hipEventRecord(e3)); CK(hipEventSynchronize(e3));
                    float ms2; CK(hipEventElapsedTime(&ms2, e2, e3));
                    g_timeline = nullptr;
                    std::vector<long long> ht(4 * 8192);
                    CK(hipMemcpy(ht.data(), tl, ht.size() * sizeof(long long), hipMemcpyDeviceToHost));
                    double tot = 0, sw = 0, nsw = 0; int nw = 0;
                    for (int g = 0; g < 1024; ++g) if (ht[g * 8 + 1] > ht[g * 8]) { tot += (double)(ht[g * 8 + 1] - ht[g * 8]); sw += (double)ht[g * 8 + 2]; nsw += (double)ht[g * 8 + 3]; ++nw; }
                    if (nw) printf("   stamps: %d workgroups, %.0f cycles each (%.3f ms -> %.2f GHz if one runs the whole launch), %.1f slab switches, %.0f cycles each = %.1f %% of the kernel\n",
                                   nw, tot / nw, ms2, tot / nw / (ms2 * 1e6), nsw / nw, nsw ? sw / nsw : 0.0, 100.0 * sw / tot);
                    // the second switch of a few workgroups, per wave: arrival, after barrier 1, slab + scales landed, after barrier 2
                    for (int g = 100; g < 101; ++g) {
                        const long long* q = &ht[8192 + (size_t)g * 64];
                        if (!q[0]) continue;
                        long long t0 = q[0]; for (int w2 = 1; w2 < 8; ++w2) if (q[w2 * 8] && q[w2 * 8] < t0) t0 = q[w2 * 8];
                        printf("      wg %3d, second switch, per wave arrive/after barrier 1/stores issued/slab stored/scales loaded/all landed/after barrier 2 (cycles after the first arrival):", g);
                        for (int w2 = 0; w2 < 8; w2 += 2) printf("  w%d %lld/%lld/%lld/%lld/%lld/%lld/%lld", w2, q[w2 * 8] - t0, q[w2 * 8 + 1] - t0, q[w2 * 8 + 6] ? q[w2 * 8 + 6] - t0 : 0, q[w2 * 8 + 4] - t0, q[w2 * 8 + 5] - t0, q[w2 * 8 + 2] - t0, q[w2 * 8 + 3] - t0);
                        printf("\n");
                    }
                    CK(hipFree(tl));
                }
                std::vector<double>& out = sums[red][slab];
                if (mode == 2) {
                    std::vector<double> hd((size_t)pe / 2);
                    CK(hipMemcpy(hd.data(), partial, hd.size() * 8, hipMemcpyDeviceToHost));
                    out.assign(cs.P, 0.0);
                    for (int wg = 0; wg < MT; ++wg) for (int t = 0; t < 256; ++t) out[t % cs.P] += hd[(size_t)wg * 256 + t];
                } else {
                    std::vector<float> hp((size_t)pe);
                    CK(hipMemcpy(hp.data(), partial, hp.size() * 4, hipMemcpyDeviceToHost));
                    const size_t per = (size_t)Npad * cs.P;
                    out.assign((size_t)n_eff * cs.P, 0.0);
                    for (int mt = 0; mt < MT; ++mt) for (size_t c = 0; c < out.size(); ++c) out[c] += hp[mt * per + c];
                }
                CK(hipFree(partial));
            }
            double worst = 0.0; size_t bad = 0;
            const std::vector<double>& a = sums[red][0]; const std::vector<double>& b = sums[red][1];
            for (size_t c = 0; c < a.size(); ++c) {
                const double rel = fabs(a[c] - b[c]) / (fabs(a[c]) + 1e-30);
                if (rel > worst) worst = rel;
                if (!(rel <= 2e-5)) { if (bad < 4) printf("   mismatch %zu: %.6e vs %.6e\n", c, a[c], b[c]); ++bad; }
            }
            printf("   %s parity slab vs stream: worst rel %.2e, %zu of %zu off\n", red ? "acc " : "cols", worst, bad, a.size());
        }
        g_slab_override = -1;
        CK(hipFree(A)); CK(hipFree(B)); CK(hipFree(ref)); CK(hipFree(sa)); CK(hipFree(sb)); CK(hipFree(rs)); CK(hipFree(rb)); CK(hipFree(bias));
    }
    return 0;
}
