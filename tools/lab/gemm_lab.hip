// Lab harness (not product): runs the scoring GEMM at the deit_small search shapes with per-workgroup cycle stamps and
// prints where a tile's time goes.  Build: make -C tools/lab ; run on the GPU box: tools/lab/gemm_lab
#include "../../adalog_amd/csrc/gemm_score.hip"
#include <vector>
#include <algorithm>
#include <string.h>

static char g_err[512];
extern "C" void adalog_set_error(const char* where, hipError_t e) { snprintf(g_err, sizeof g_err, "%s: %s", where, hipGetErrorString(e)); }
extern "C" void adalog_set_error_msg(const char* msg) { snprintf(g_err, sizeof g_err, "%s", msg); }

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

struct Case { const char* name; int dtype; int M, N, K, P; int rows; int order; };

int main(int argc, char** argv) {
    // weight search: rows = tokens, cols = (out channel, candidate); activation search: rows = out channels (row scale),
    // cols = (token, candidate).  K in elements.
    const int T = 32 * 197;
    std::vector<Case> cases = {
        {"w-search qkv  i8  K=384", 0, T, 1152 * 128, 384, 128, 0, 2},
        {"a-search qkv  i8  K=384", 0, 1152, T * 128, 384, 128, 1, 2},
        {"w-search qkv  fp8 K=384", 3, T, 1152 * 128, 384, 128, 0, 2},
        {"a-search qkv  fp8 K=384", 3, 1152, T * 128, 384, 128, 1, 2},
        {"a-search fc2  bf16 K=1536", 1, 384, T * 128, 1536, 128, 1, 2},
        {"w-search fc2  bf16 K=1536", 1, T, 384 * 128, 1536, 128, 0, 2},
        {"a-search proj i8  K=384", 0, 384, T * 128, 384, 128, 1, 2},
        {"a-search fc2  i8  K=1536", 0, 384, T * 128, 1536, 128, 1, 2},
    };
    for (const Case& cs : cases) {
        const int esz = (cs.dtype == 0 || cs.dtype == 3) ? 1 : 2;
        const size_t abytes = (size_t)cs.M * cs.K * esz, bbytes = (size_t)cs.N * cs.K * esz;
        uint8_t *A, *B; float *ref, *sa, *sb, *rs, *rb, *partial; long long* tl;
        CK(hipMalloc(&A, abytes)); CK(hipMalloc(&B, bbytes));
        std::vector<uint8_t> h(std::max(abytes, bbytes));
        for (size_t i = 0; i < h.size(); ++i) h[i] = (cs.dtype == 0 || cs.dtype == 3) ? (uint8_t)((i * 2654435761u >> 13) & 7) : (uint8_t)((i & 1) ? 0x3f : ((i * 2654435761u >> 13) & 0x7f));
        CK(hipMemcpy(A, h.data(), abytes, hipMemcpyHostToDevice)); CK(hipMemcpy(B, h.data(), bbytes, hipMemcpyHostToDevice));
        const int n_eff = cs.N / cs.P;
        CK(hipMalloc(&ref, (size_t)cs.M * n_eff * 4)); CK(hipMemset(ref, 0, (size_t)cs.M * n_eff * 4));
        CK(hipMalloc(&sa, 4 * 128)); CK(hipMalloc(&sb, (size_t)cs.N * 4)); CK(hipMalloc(&rs, cs.M * 4)); CK(hipMalloc(&rb, cs.M * 4));
        std::vector<float> ones(std::max(cs.N, cs.M), 1.0f);
        CK(hipMemcpy(sa, ones.data(), 4 * 128, hipMemcpyHostToDevice)); CK(hipMemcpy(sb, ones.data(), (size_t)cs.N * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(rs, ones.data(), cs.M * 4, hipMemcpyHostToDevice)); CK(hipMemset(rb, 0, cs.M * 4));
        int MT, Npad;
        int mode;
        const int64_t pe = adalog_gemm_score_layout(cs.M, cs.N, 1, 1, 1, cs.P, 0, cs.dtype, cs.K, cs.K, 1, &MT, &Npad, &mode);
        CK(hipMalloc(&partial, pe * 4));
        const Layout L = layout_of(cs.M, cs.N, 1, 1, 1, cs.P, 0, true, (int64_t)cs.K * esz, (int64_t)cs.K * esz, true);
        const size_t nwg = (size_t)L.MT * L.NT;
        CK(hipMalloc(&tl, nwg * 8 * sizeof(long long))); CK(hipMemset(tl, 0, nwg * 8 * sizeof(long long)));
        auto run = [&]() {
            // weight search: sb[cand][n] (sb_c = n_eff, sb_n = 1); activation search: sa[cand], sb shared
            int rc = adalog_gemm_score(cs.dtype, A, B, 0, 0, 0, 0, cs.M, cs.N, cs.K, 0, 1, 1, 1, ref, 1, 0, cs.M, cs.P,
                                       sa, cs.rows ? 1 : 0, 0, 1.0f, sb, cs.rows ? 0 : n_eff, 0, cs.rows ? 0 : 1, nullptr, 0, 0, 0,
                                       cs.rows ? rs : nullptr, cs.rows ? rb : nullptr, partial, pe, nullptr, 0, 0, 0, cs.order, 0, nullptr);
            if (rc) { printf("gemm error: %s\n", g_err); exit(1); }
        };
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        g_timeline = nullptr;
        run(); run(); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0)); for (int i = 0; i < 5; ++i) run(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
        const double flop = 2.0 * cs.M * (double)cs.N * cs.K;
        printf("%-28s tiles %zu (MT %d NT %d rows/64 %d)  %.3f ms  %.1f TFLOP/s\n", cs.name, nwg, L.MT, L.NT, L.tm, ms, flop / ms * 1e-9);
        g_timeline = tl;
        hipEvent_t e2, e3; CK(hipEventCreate(&e2)); CK(hipEventCreate(&e3));
        CK(hipEventRecord(e2)); run(); CK(hipEventRecord(e3)); CK(hipEventSynchronize(e3));
        float ms2; CK(hipEventElapsedTime(&ms2, e2, e3));
        g_timeline = nullptr;
        std::vector<long long> ht(nwg * 8);
        CK(hipMemcpy(ht.data(), tl, nwg * 8 * sizeof(long long), hipMemcpyDeviceToHost));
        // counters are per XCD: only differences inside one workgroup mean anything.  Calibrate ticks/us from the
        // median workgroup span x workgroups per CU ~= kernel time (one workgroup per CU, back to back).
        std::vector<double> span(nwg);
        for (size_t wg = 0; wg < nwg; ++wg) span[wg] = (double)(ht[wg * 8 + 7] - ht[wg * 8]);
        std::sort(span.begin(), span.end());
        const double ticks_per_us = 1.0;
        printf("   median span %.0f ticks; if back-to-back: %.1f ticks/us\n", span[nwg / 2], span[nwg / 2] * (nwg / 256.0) / (ms2 * 1e3));
        const bool stream = !getenv("ADALOG_GEMM_STREAM") || atoi(getenv("ADALOG_GEMM_STREAM"));
        const char* names_s[7] = {"wait+barrier", "first step+stage", "rest of main loop", "epilogue", "barrier", "partial store", "-"};
        const char* names_g[7] = {"setup+issue", "first stage lands", "main loop", "sync", "epilogue", "sync", "partial store"};
        const char** names = stream ? names_s : names_g;
        printf("   stamped run %.3f ms; per-tile phases (median / p90, cycles):\n", ms2);
        double tot = 0;
        for (int ph = 0; ph < 7; ++ph) {
            std::vector<double> d(nwg);
            for (size_t wg = 0; wg < nwg; ++wg) d[wg] = (double)(ht[wg * 8 + ph + 1] - ht[wg * 8 + ph]) / ticks_per_us;
            std::sort(d.begin(), d.end());
            printf("     %-18s %7.0f / %7.0f\n", names[ph], d[nwg / 2], d[nwg * 9 / 10]);
            tot += d[nwg / 2];
        }
        std::vector<double> d(nwg);
        for (size_t wg = 0; wg < nwg; ++wg) d[wg] = (double)(ht[wg * 8 + 7] - ht[wg * 8]) / ticks_per_us;
        std::sort(d.begin(), d.end());
        printf("     %-18s %7.0f / %7.0f   (tiles/CU %.1f -> %.2f us per tile slot)\n", "tile total", d[nwg / 2], d[nwg * 9 / 10],
               nwg / 256.0, ms2 * 1e3 / (nwg / 256.0));
        CK(hipFree(A)); CK(hipFree(B)); CK(hipFree(ref)); CK(hipFree(sa)); CK(hipFree(sb)); CK(hipFree(rs)); CK(hipFree(rb));
        CK(hipFree(partial)); CK(hipFree(tl));
    }
    return 0;
}
