// Lab harness (not product): adalog_gemm_f32x3 on the twelve Linear products of a deit_small BRECQ iteration, timed with HIP
// events.  Built with -DBQ_LAB_NOSPLIT / NOMFMA / NODMA / NOEPI to time the parts of the main loop separately.
#include "../../adalog_amd/csrc/brecq_gemm.hip"
#include <vector>
#include <string.h>
static char g_err[512];
extern "C" void adalog_set_error(const char* where, hipError_t e) { snprintf(g_err, sizeof g_err, "%s: %s", where, hipGetErrorString(e)); }
extern "C" void adalog_set_error_msg(const char* msg) { snprintf(g_err, sizeof g_err, "%s", msg); }
extern "C" void adalog_note_kernel(const char*) {}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
struct Case { const char* name; int M, N, K, ta, tb, ea, eb; int planes = 0; };   // ea / eb: 0 general (three terms), 1 exact, 2 two terms
int main(int argc, char** argv) {
    const int T = 32 * 197;
    std::vector<Case> cases = {
        {"qkv fwd", T, 1152, 384, 0, 0, 0, 0}, {"qkv dx", T, 384, 1152, 0, 1, 0, 0}, {"qkv dw", 1152, 384, T, 1, 1, 0, 0},
        {"fc1 fwd", T, 1536, 384, 0, 0, 0, 0}, {"fc1 dx", T, 384, 1536, 0, 1, 0, 0}, {"fc1 dw", 1536, 384, T, 1, 1, 0, 0},
        {"fc2 fwd", T, 384, 1536, 0, 0, 0, 0}, {"fc2 dx", T, 1536, 384, 0, 1, 0, 0}, {"fc2 dw", 384, 1536, T, 1, 1, 0, 0},
        {"proj fwd", T, 384, 384, 0, 0, 0, 0},
        {"qkv fwdI", T, 1152, 384, 0, 0, 1, 0}, {"qkv dwI", 1152, 384, T, 1, 1, 0, 1},
        {"fc1 fwdI", T, 1536, 384, 0, 0, 1, 0}, {"fc1 dwI", 1536, 384, T, 1, 1, 0, 1},
        {"qkv fwdIP", T, 1152, 384, 0, 0, 1, 0, 1}, {"qkv dxP", T, 384, 1152, 0, 0, 0, 0, 1}, {"proj fwdIP", T, 384, 384, 0, 0, 1, 0, 1}, {"proj dxP", T, 384, 384, 0, 0, 0, 0, 1},
        {"fc1 fwdIP", T, 1536, 384, 0, 0, 1, 0, 1}, {"fc1 dxP", T, 384, 1536, 0, 0, 0, 0, 1}, {"fc2 fwdP", T, 384, 1536, 0, 0, 0, 0, 1}, {"fc2 dxP", T, 1536, 384, 0, 0, 0, 0, 1},
        // 22..: the forms a round-6 iteration issues (two-term general operands, integer activations; B of a forward K-major)
        {"qkv fwdI2", T, 1152, 384, 0, 1, 1, 2}, {"qkv dx2", T, 384, 1152, 0, 1, 2, 2}, {"qkv dwI2", 1152, 384, T, 1, 1, 2, 1},
        {"fc1 fwdI2", T, 1536, 384, 0, 1, 1, 2}, {"fc1 dx2", T, 384, 1536, 0, 1, 2, 2}, {"fc1 dwI2", 1536, 384, T, 1, 1, 2, 1},
        {"fc2 fwd2", T, 384, 1536, 0, 1, 2, 2}, {"fc2 dx2", T, 1536, 384, 0, 1, 2, 2}, {"fc2 dw2", 384, 1536, T, 1, 1, 2, 2},
        // 31..: vit_base
        {"vb qkv fwdI2", T, 2304, 768, 0, 1, 1, 2}, {"vb qkv dx2", T, 768, 2304, 0, 1, 2, 2}, {"vb qkv dwI2", 2304, 768, T, 1, 1, 2, 1},
        {"vb fc1 fwdI2", T, 3072, 768, 0, 1, 1, 2}, {"vb fc1 dx2", T, 768, 3072, 0, 1, 2, 2}, {"vb fc1 dwI2", 3072, 768, T, 1, 1, 2, 1},
        {"vb fc2 fwd2", T, 768, 3072, 0, 1, 2, 2}, {"vb fc2 dx2", T, 3072, 768, 0, 1, 2, 2}, {"vb fc2 dw2", 768, 3072, T, 1, 1, 2, 2},
    };
    float *A, *B, *C, *W;
    const size_t big = (size_t)T * 3072;
    CK(hipMalloc(&A, big * 4)); CK(hipMalloc(&B, big * 4)); CK(hipMalloc(&C, big * 4)); CK(hipMalloc(&W, (size_t)256 << 20));
    std::vector<float> h(big);
    for (size_t i = 0; i < big; ++i) h[i] = (float)((int)((i * 2654435761u >> 9) & 4095) - 2048) * (1.0f / 1024.0f) + 1e-4f * (float)(i % 97);
    CK(hipMemcpy(A, h.data(), big * 4, hipMemcpyHostToDevice));
    for (size_t i = 0; i < big; ++i) h[i] = (float)((int)((i * 40503u >> 5) & 4095) - 2048) * (1.0f / 2048.0f) + 1e-5f * (float)(i % 89);
    CK(hipMemcpy(B, h.data(), big * 4, hipMemcpyHostToDevice));
    // pre-split planes of B (three bf16 planes per row, Kt = K rounded to 32): any bit pattern will do for timing
    uint16_t* Bp; CK(hipMalloc(&Bp, big * 6 + (1 << 20))); CK(hipMemset(Bp, 0x3c, big * 6 + (1 << 20)));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    double total = 0;
    const int only = argc > 1 ? atoi(argv[1]) : -1, iters = argc > 2 ? atoi(argv[2]) : 30, from = argc > 3 ? atoi(argv[3]) : 0;
    int ci = -1;
    for (const Case& c : cases) {
        if ((++ci != only && only >= 0) || ci < from) continue;
        const int64_t lda = c.ta ? c.M : c.K, ldb = c.tb ? c.N : c.K;
        const bool planes = c.planes != 0;
        const int64_t Kt = (c.K + 31) / 32 * 32;
        auto run = [&]() {
            if (planes) return adalog_gemm_f32x3_planes(A, lda, Bp, Kt, C, c.N, c.M, c.N, c.K, 1, 0, 0, nullptr, 1.0f, nullptr, 1, c.ea, W, nullptr);
            return adalog_gemm_f32x3(A, lda, c.ta, B, ldb, c.tb, C, c.N, c.M, c.N, c.K, 1, 0, 0, 0, nullptr, 1.0f, nullptr, 1, c.ea, c.eb, W, nullptr); };
        if (argc > 4 && !strcmp(argv[4], "sweep")) {      // every (tile shape, K split) the planner could pick: us per call, reduce included
            double best = 1e30; int bs = -1, bp = -1;
            unsetenv("ADALOG_BQ_SHAPE"); unsetenv("ADALOG_BQ_SPLIT");
            auto time_it = [&]() {
                for (int i = 0; i < 3; ++i) if (run() != 0) return -1.0;
                CK(hipDeviceSynchronize());
                CK(hipEventRecord(e0));
                for (int i = 0; i < iters; ++i) run();
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                return (double)ms * 1e3 / iters;
            };
            const double planned = time_it();
            printf("%-12s %5dx%5dx%5d planner %6.1f |", c.name, c.M, c.N, c.K, planned);
            for (int sh : {2, 3, 4, 0}) {
                char b1[8]; snprintf(b1, sizeof b1, "%d", sh); setenv("ADALOG_BQ_SHAPE", b1, 1);
                printf(" s%d:", sh);
                for (int sp : {1, 2, 3, 4, 6, 8, 12}) {
                    char b2[8]; snprintf(b2, sizeof b2, "%d", sp); setenv("ADALOG_BQ_SPLIT", b2, 1);
                    if ((double)sp * c.M * c.N * 4.0 > 250e6 || planes) { printf("     -"); continue; }     // (the workspace holds 256 MB)
                    const double us = time_it();
                    printf(" %5.1f", us);
                    if (us > 0 && us < best) { best = us; bs = sh; bp = sp; }
                }
            }
            unsetenv("ADALOG_BQ_SHAPE"); unsetenv("ADALOG_BQ_SPLIT");
            printf(" | best %.1f (shape %d, split %d)\n", best, bs, bp);
            total += best;
            continue;
        }
        for (int i = 0; i < 3; ++i) if (run() != 0) { printf("%s: %s\n", c.name, g_err); return 1; }
        CK(hipDeviceSynchronize());
        const int it = iters;
        CK(hipEventRecord(e0));
        for (int i = 0; i < it; ++i) run();
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / it;
        const int prod = planes ? (c.ea ? 3 : 6) : bq_products(c.ea, c.eb, c.ta, c.tb);
        const BqPlan pl = bq_plan(c.M, c.N, c.K, 1, 1, prod, planes);
        printf("%-12s %5dx%5dx%5d  %-20s S=%2d wgs=%3d  %7.1f us  %7.1f bf16 TF/s\n", c.name, c.M, c.N, c.K, BQ_SHAPES[pl.shape].name, pl.S, pl.wgs,
               us, 2.0 * prod * c.M * c.N * c.K / us / 1e6);
        total += us;
    }
    printf("total %.1f us\n", total);
    return 0;
}
