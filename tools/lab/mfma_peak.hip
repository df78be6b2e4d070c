// Attainable MFMA rate on this part: back-to-back independent v_mfma_f32_32x32x16_bf16, nothing else.
//   hipcc --offload-arch=gfx950 -O3 tools/lab/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v16f __attribute__((ext_vector_type(16)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    v16f acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    v8bf a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(float)(threadIdx.x + j); b[j] = (__bf16)(float)(j); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
void run(int wpb_blocks, const char* tag) {
    float* out; hipMalloc(&out, 4 << 20);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        k<NACC><<<wpb_blocks, 256>>>(out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double flops = (double)wpb_blocks * 4 * iters * NACC * 32768.0;
        printf("%s blocks %d: %.3f ms  %.1f TFLOP/s\n", tag, wpb_blocks, ms, flops / ms / 1e9);
    }
    hipFree(out);
}
int main() {
    run<4>(256, "4 acc, 1 wave/SIMD");
    run<4>(512, "4 acc, 2 waves/SIMD");
    run<8>(256, "8 acc, 1 wave/SIMD");
    run<4>(1024, "4 acc, 4 waves/SIMD");
    return 0;
}
