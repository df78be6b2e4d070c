// Probe: v_mfma_scale_f32_32x32x64_f8f6f4 with fp8 (e4m3) operands built by v_cvt_pk_fp8_f32 from small integers:
// is the product exact, and which K elements does a lane hold?  (lab only)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

__global__ void k(const float* A, const float* B, float* D) {   // A[32][64], B[32][64] row-major floats (integers)
    const int lane = threadIdx.x, row = lane & 31, kh = lane >> 5;
    v8i a, b;
    for (int w = 0; w < 8; ++w) {
        const int k0 = kh * 32 + w * 4;
        int pa = 0, pb = 0;
        pa = __builtin_amdgcn_cvt_pk_fp8_f32(A[row * 64 + k0], A[row * 64 + k0 + 1], pa, false);
        pa = __builtin_amdgcn_cvt_pk_fp8_f32(A[row * 64 + k0 + 2], A[row * 64 + k0 + 3], pa, true);
        pb = __builtin_amdgcn_cvt_pk_fp8_f32(B[row * 64 + k0], B[row * 64 + k0 + 1], pb, false);
        pb = __builtin_amdgcn_cvt_pk_fp8_f32(B[row * 64 + k0 + 2], B[row * 64 + k0 + 3], pb, true);
        a[w] = pa; b[w] = pb;
    }
    v16f c;
    for (int r = 0; r < 16; ++r) c[r] = 0.0f;
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    // 32x32 accumulator layout as for the other 32x32 MFMAs: column = lane & 31, rows 8*(r>>2) + 4*(lane>>5) + (r&3)
    for (int r = 0; r < 16; ++r) D[(8 * (r >> 2) + 4 * kh + (r & 3)) * 32 + row] = c[r];
}

int main() {
    std::vector<float> A(32 * 64), B(32 * 64), D(32 * 32);
    unsigned s = 1;
    for (auto& v : A) { s = s * 1664525u + 1013904223u; v = (float)((int)((s >> 16) % 31) - 15); }
    for (auto& v : B) { s = s * 1664525u + 1013904223u; v = (float)((int)((s >> 16) % 31) - 15); }
    float *dA, *dB, *dD;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dD, D.size() * 4);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int m = 0; m < 32; ++m) for (int n = 0; n < 32; ++n) {
        float ref = 0; for (int kk = 0; kk < 64; ++kk) ref += A[m * 64 + kk] * B[n * 64 + kk];
        if (ref != D[m * 32 + n]) { if (bad < 5) printf("mismatch m %d n %d ref %g got %g\n", m, n, ref, D[m * 32 + n]); ++bad; }
    }
    printf("fp8 32x32x64 probe: %d mismatches of 1024\n", bad);
    return 0;
}
