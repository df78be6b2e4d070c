#!/usr/bin/env python3
"""Generates tools/lab/overlap_probe.hip: does a wave's own VALU / LDS work run in the shadow of its MFMAs?
One wave per SIMD; loop body = 4 x { v_mfma_f32_32x32x16_bf16 ; N filler instructions } for several N and filler kinds."""
import os
kinds = {
    "fma": lambda i: f"v_fma_f32 v{40 + i % 8}, v{50 + i % 4}, v{54 + i % 4}, v{40 + i % 8}",
    "lshladd": lambda i: f"v_lshl_add_u32 v{40 + i % 8}, v{50 + i % 4}, 9, v{54 + i % 4}",
    "salu": lambda i: f"s_add_i32 s{40 + i % 8}, s{48 + i % 4}, 1",
    "ldsb32": lambda i: f"ds_read_b32 v{40 + i % 8}, v60",
}
cases = []
for acc in ("a", "v"):
    for kind in kinds:
        for n in (0, 4, 6, 7, 8, 12):
            if acc == "v" or kind == "salu":
                continue
            cases.append((acc, kind, n))
src = ['#include <hip/hip_runtime.h>', '#include <cstdio>', '#include <cstring>',
       'extern __shared__ unsigned char smem[];']
for acc, kind, n in cases:
    accreg = (lambda t: f"a[{16 * t}:{16 * t + 15}]") if acc == "a" else (lambda t: f"v[{128 + 16 * t}:{128 + 16 * t + 15}]")
    body = []
    for t in range(4):
        body.append(f"v_mfma_f32_32x32x16_bf16 {accreg(t)}, v[0:3], v[4:7], {accreg(t)}")
        for i in range(n):
            body.append(kinds[kind](t * n + i))
        if kind == "ldsb32" and n:
            body.append("s_waitcnt lgkmcnt(8)")
    text = "\\n\\t".join(body)
    clob = ", ".join([f'"v{i}"' for i in list(range(0, 8)) + list(range(40, 64)) + list(range(128, 192))] +
                     [f'"a{i}"' for i in range(64)] + [f'"s{i}"' for i in range(40, 52)] + ['"memory"'])
    src.append(f'''__global__ __launch_bounds__(256, 2) void k_{acc}_{kind}_{n}(float* out, int iters) {{
    asm volatile("v_mov_b32 v60, 0\\n\\t" ::: "v60");
    for (int it = 0; it < iters; ++it) asm volatile("{text}\\n\\ts_waitcnt lgkmcnt(0)" ::: {clob});
    out[threadIdx.x] = (float)iters;
}}''')
src.append('''template <class K> double run(K kern, int iters, int grid = 256) {
    float* out; hipMalloc(&out, 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    double best = 1e9;
    for (int r = 0; r < 3; ++r) {
        hipEventRecord(e0); hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 1024, 0, out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    hipFree(out); return best;
}
int main() {
    const int iters = 20000;''')
for acc, kind, n in cases:
    src.append(f'    {{ double ms = run(k_{acc}_{kind}_{n}, iters), m2 = run(k_{acc}_{kind}_{n}, iters, 512); printf("acc={acc} filler={kind:8s} n=%2d : 1 wave/SIMD %.3f ms (%.1f cyc/MFMA)   2 waves/SIMD %.3f ms (%.1f cyc/MFMA of the SIMD)\\n", {n}, ms, ms * 1e-3 * 2.4e9 / (iters * 4.0), m2, m2 * 1e-3 * 2.4e9 / (iters * 8.0)); }}')
src.append("    return 0;\n}")
open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "overlap_probe.hip"), "w").write("\n".join(src) + "\n")
