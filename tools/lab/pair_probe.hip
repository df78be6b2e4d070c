// Lab probe (not product): do one wave's VALU instructions run in the shadow of its SIMD partner's MFMAs?
// A 512-thread workgroup per CU: waves 0-3 (one per SIMD) issue only MFMAs, waves 4-7 (their SIMD partners) only VALU.
// Timed three ways -- MFMA waves alone, VALU waves alone, both -- for fp8 (64-cycle) and bf16 (32-cycle) MFMAs and a few
// VALU kinds.  both ~= max(alone) means full overlap, both ~= sum means none.
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(x) x x x x x x x x
#define REP16(x) REP8(x) REP8(x)

// MK: 0 = v_mfma_scale_f32_32x32x64_f8f6f4 (8 per iteration), 1 = v_mfma_f32_32x32x16_bf16 (16 per iteration)
// VK: 0 = v_pk_fma_f32, 1 = v_fma_f32, 2 = v_mov_b32, 3 = v_pk_mul_f32 reading the "accumulator" range v[64:127]
template <int MK, int VK, int NV8>
__global__ __launch_bounds__(512) void k_pair(long long* out, int iters, int mode) {
    const int w = threadIdx.x >> 6;
    const int role = (w >> 2) & 1;
    long long t0 = clock64();
    if (role == 0) {
        if (mode & 1) {
            for (int it = 0; it < iters; ++it) {
                if (MK == 0) {
                    asm volatile(
                        "v_mfma_scale_f32_32x32x64_f8f6f4 a[0:15], v[0:7], v[8:15], a[0:15], v16, v16 op_sel_hi:[0,0,0]\n\t"
                        "v_mfma_scale_f32_32x32x64_f8f6f4 a[16:31], v[0:7], v[8:15], a[16:31], v16, v16 op_sel_hi:[0,0,0]\n\t"
                        "v_mfma_scale_f32_32x32x64_f8f6f4 a[32:47], v[0:7], v[8:15], a[32:47], v16, v16 op_sel_hi:[0,0,0]\n\t"
                        "v_mfma_scale_f32_32x32x64_f8f6f4 a[48:63], v[0:7], v[8:15], a[48:63], v16, v16 op_sel_hi:[0,0,0]\n\t"
                        "v_mfma_scale_f32_32x32x64_f8f6f4 a[64:79], v[0:7], v[8:15], a[64:79], v16, v16 op_sel_hi:[0,0,0]\n\t"
                        "v_mfma_scale_f32_32x32x64_f8f6f4 a[80:95], v[0:7], v[8:15], a[80:95], v16, v16 op_sel_hi:[0,0,0]\n\t"
                        "v_mfma_scale_f32_32x32x64_f8f6f4 a[96:111], v[0:7], v[8:15], a[96:111], v16, v16 op_sel_hi:[0,0,0]\n\t"
                        "v_mfma_scale_f32_32x32x64_f8f6f4 a[112:127], v[0:7], v[8:15], a[112:127], v16, v16 op_sel_hi:[0,0,0]\n\t"
                        ::: "memory", "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127");
                } else {
                    asm volatile(REP8(
                        "v_mfma_f32_32x32x16_bf16 a[0:15], v[0:3], v[8:11], a[0:15]\n\t"
                        "v_mfma_f32_32x32x16_bf16 a[16:31], v[0:3], v[8:11], a[16:31]\n\t") ::: "memory", "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31");
                }
            }
        }
    } else {
        if (mode & 2) {
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int r = 0; r < NV8; ++r) {
                    if (VK == 0)
                        asm volatile("v_pk_fma_f32 v[40:41], v[56:57], v[58:59], v[40:41]\n\tv_pk_fma_f32 v[42:43], v[56:57], v[58:59], v[42:43]\n\t"
                                     "v_pk_fma_f32 v[44:45], v[56:57], v[58:59], v[44:45]\n\tv_pk_fma_f32 v[46:47], v[56:57], v[58:59], v[46:47]\n\t"
                                     "v_pk_fma_f32 v[48:49], v[56:57], v[58:59], v[48:49]\n\tv_pk_fma_f32 v[50:51], v[56:57], v[58:59], v[50:51]\n\t"
                                     "v_pk_fma_f32 v[52:53], v[56:57], v[58:59], v[52:53]\n\tv_pk_fma_f32 v[54:55], v[56:57], v[58:59], v[54:55]\n\t" ::: "memory", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59");
                    else if (VK == 1)
                        asm volatile("v_fma_f32 v40, v56, v58, v40\n\tv_fma_f32 v41, v56, v58, v41\n\tv_fma_f32 v42, v56, v58, v42\n\tv_fma_f32 v43, v56, v58, v43\n\t"
                                     "v_fma_f32 v44, v56, v58, v44\n\tv_fma_f32 v45, v56, v58, v45\n\tv_fma_f32 v46, v56, v58, v46\n\tv_fma_f32 v47, v56, v58, v47\n\t" ::: "memory", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59");
                    else if (VK == 2)
                        asm volatile("v_mov_b32 v40, v56\n\tv_mov_b32 v41, v56\n\tv_mov_b32 v42, v56\n\tv_mov_b32 v43, v56\n\t"
                                     "v_mov_b32 v44, v56\n\tv_mov_b32 v45, v56\n\tv_mov_b32 v46, v56\n\tv_mov_b32 v47, v56\n\t" ::: "memory", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59");
                    else
                        asm volatile("v_pk_mul_f32 v[40:41], v[64:65], v[58:59]\n\tv_pk_mul_f32 v[42:43], v[70:71], v[58:59]\n\t"
                                     "v_pk_mul_f32 v[44:45], v[76:77], v[58:59]\n\tv_pk_mul_f32 v[46:47], v[82:83], v[58:59]\n\t"
                                     "v_pk_mul_f32 v[48:49], v[88:89], v[58:59]\n\tv_pk_mul_f32 v[50:51], v[94:95], v[58:59]\n\t"
                                     "v_pk_mul_f32 v[52:53], v[100:101], v[58:59]\n\tv_pk_mul_f32 v[54:55], v[106:107], v[58:59]\n\t" ::: "memory", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59");
                }
            }
        }
    }
    long long t1 = clock64();
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) out[w] = t1 - t0;
}

template <class K> void run(const char* name, K kern, int nv, int mfma_per_iter, int mfma_cyc) {
    long long* out; hipMalloc(&out, 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    double ms[4];
    long long clk[4][8];
    for (int mode = 1; mode <= 3; ++mode) {
        double best = 1e9;
        for (int r = 0; r < 3; ++r) {
            hipEventRecord(e0); hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, out, iters, mode); hipEventRecord(e1); hipEventSynchronize(e1);
            float t; hipEventElapsedTime(&t, e0, e1); if (t < best) best = t;
        }
        ms[mode] = best;
        hipMemcpy(clk[mode], out, 64, hipMemcpyDeviceToHost);
    }
    printf("%-28s nv/iter=%3d : mfma alone %.3f ms (%.1f clk/MFMA)  valu alone %.3f ms (%.2f clk/VALU)  both %.3f ms  [max %.3f sum %.3f]"
           "  in 'both': mfma wave %.2fx its alone time, valu wave %.2fx\n",
           name, nv, ms[1], (double)clk[1][0] / iters / mfma_per_iter, ms[2], (double)clk[2][4] / iters / nv, ms[3],
           ms[1] > ms[2] ? ms[1] : ms[2], ms[1] + ms[2], (double)clk[3][0] / clk[1][0], (double)clk[3][4] / clk[2][4]);
    (void)mfma_cyc;
    hipFree(out);
}

int main() {
#define RUN(MK, VK, NV8, nm) run(nm, k_pair<MK, VK, NV8>, NV8 * 8, MK == 0 ? 8 : 16, MK == 0 ? 64 : 32)
    RUN(0, 0, 16, "fp8x64 | pk_fma");
    RUN(0, 0, 8, "fp8x64 | pk_fma");
    RUN(0, 0, 4, "fp8x64 | pk_fma");
    RUN(0, 1, 16, "fp8x64 | fma");
    RUN(0, 1, 8, "fp8x64 | fma");
    RUN(0, 2, 16, "fp8x64 | mov");
    RUN(0, 3, 16, "fp8x64 | pk_mul(v64..)");
    RUN(1, 0, 16, "bf16x16 | pk_fma");
    RUN(1, 0, 8, "bf16x16 | pk_fma");
    RUN(1, 1, 16, "bf16x16 | fma");
    RUN(1, 2, 16, "bf16x16 | mov");
    return 0;
}
