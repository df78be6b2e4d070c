// K5/K6 -- exact order statistics by MSB-first radix select (4 passes x 8 bits over order-preserving uint32 keys),
// replacing the full sorts behind torch.quantile / sort in
//   quant_layers/linear.py:436-441,462-468 (weight / activation percentile candidates), matmul.py:225-226,
//   conv.py:275-280, and linear.py:763-798 (positive_percentile: rank ceil(count*q)-1 among the values > 0).
// Each pass streams the segment once (4 B/element, coalesced) and histograms into LDS, then one global integer
// atomicAdd per bin (integer atomics: order-independent, deterministic result).  Up to 8 ranks per segment are
// resolved together.  The final kernel applies torch.quantile's linear interpolation
//   pos = q*(n-1);  v = lerp(sorted[floor(pos)], sorted[ceil(pos)], pos - floor(pos))        
// and, for per-tensor activations above 2**24 elements, the reference's mean over chunk quantiles.
#include "common.h"

namespace {

constexpr int MAXR = 8;

__device__ __forceinline__ uint32_t f2key(float f) {
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key2f(uint32_t k) {
    uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return __uint_as_float(u);
}

struct SelState {              // per (segment, rank)
    uint32_t prefix;
    int64_t remaining;         // rank still to skip inside the current bucket; < 0 => empty selection
};

__global__ __launch_bounds__(256) void k_sel_init(SelState* st, const int64_t* ranks, int S, int R) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= S * R) return;
    st[i].prefix = 0;
    st[i].remaining = ranks ? ranks[i % R] : 0;
}

// Local segment `seg` (rows of x) accumulates into state / histogram slot  first + (seg / inner) * outer + seg % inner:
// identity for a single process; with image-sharded ranks each rank adds its rows' counts into the GLOBAL segment they
// belong to and the histograms are summed across ranks before the pick (adalog_select_* entry points).
__global__ __launch_bounds__(256) void k_sel_hist(const float* __restrict__ x, int64_t n, int R, int pass,
                                                  const SelState* __restrict__ st, unsigned* __restrict__ hist,
                                                  int positive_only, int first, int inner, int outer) {
    __shared__ unsigned h[MAXR][256];
    __shared__ uint32_t pre[MAXR];
    const int seg = blockIdx.y;
    const int slot = first + (seg / inner) * outer + seg % inner;
    for (int i = threadIdx.x; i < R * 256; i += blockDim.x) (&h[0][0])[i] = 0;
    if ((int)threadIdx.x < R) pre[threadIdx.x] = st[slot * R + threadIdx.x].prefix;
    __syncthreads();
    const int shift = 24 - 8 * pass;
    const float* xs = x + (int64_t)seg * n;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = xs[i];
        if (positive_only && !(v > 0.0f)) continue;
        const uint32_t key = f2key(v);
        const unsigned bin = (key >> shift) & 255u;
        if (pass == 0) {                                   // the histogram of the top byte is the same for every rank: count once
            atomicAdd(&h[0][bin], 1u);
        } else {
            const uint32_t hi = key >> (shift + 8);
            for (int r = 0; r < R; ++r)
                if (hi == pre[r]) atomicAdd(&h[r][bin], 1u);
        }
    }
    __syncthreads();
    unsigned* gh = hist + ((int64_t)slot * R) * 256;
    for (int i = threadIdx.x; i < R * 256; i += blockDim.x) {
        const unsigned c = pass == 0 ? h[0][i & 255] : (&h[0][0])[i];
        if (c) atomicAdd(gh + i, c);
    }
}

// one wavefront per (segment, rank): each lane takes four bins, a 64-lane scan locates the bin holding the rank, the
// state descends one byte, and the histogram is cleared for the next pass (a single thread walking 256 bins took 30 us)
__global__ __launch_bounds__(256) void k_sel_pick(SelState* st, unsigned* hist, int S, int R, int pass,
                                                  const float* __restrict__ qfrac, int positive_only) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= S * R) return;
    uint4* h4 = reinterpret_cast<uint4*>(hist + (int64_t)i * 256);
    const uint4 c = h4[lane];
    h4[lane] = make_uint4(0u, 0u, 0u, 0u);
    SelState s = st[i];
    const int64_t own = (int64_t)c.x + c.y + c.z + c.w;
    int64_t incl = own;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int64_t up = __shfl_up(incl, d);
        if (lane >= d) incl += up;
    }
    const int64_t total = __shfl(incl, 63);
    if (pass == 0 && positive_only) {
        // ranks = clamp(ceil(count * q) - 1, 0)      (linear.py:785-790; counts.float() * q in fp32)
        const float cq = ceilf((float)total * qfrac[i % R]);
        const int64_t rk = (int64_t)cq - 1;
        // counts.float() rounds to even above 2^24 positives: the rank can land ONE PAST the last positive value, where the
        // reference's sorted tensor holds NaN and the result is masked to 0 (linear.py:794-797) -- same here (-1 -> 0)
        s.remaining = (total == 0 || rk >= total) ? -1 : (rk < 0 ? 0 : rk);
    }
    if (s.remaining >= 0) {
        const unsigned long long hit = __ballot(incl > s.remaining);
        int b = 255;                          // no bin reaches the rank: only with NaNs in the data; keep it defined
        int64_t cum = total;
        if (hit) {
            const int fl = __ffsll(hit) - 1;
            const int64_t excl = __shfl(incl - own, fl);
            const unsigned cx = __shfl(c.x, fl), cy = __shfl(c.y, fl), cz = __shfl(c.z, fl);
            cum = excl; b = 4 * fl;
            if (cum + cx <= s.remaining) { cum += cx; ++b;
                if (cum + cy <= s.remaining) { cum += cy; ++b;
                    if (cum + cz <= s.remaining) { cum += cz; ++b; } } }
        }
        s.prefix = (s.prefix << 8) | (unsigned)b;
        s.remaining -= cum;
    }
    if (lane == 0) st[i] = s;
}

// quantile mode: out[j][col] = mean over the mbs chunk rows of lerp(v[2j], v[2j+1], w[j]);  R = 2*nq
__global__ __launch_bounds__(256) void k_sel_quantile_out(const SelState* __restrict__ st, int R, int nq,
                                                          const float* __restrict__ w, int cols, int mbs,
                                                          float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cols * nq) return;
    const int col = i % cols, j = i / cols;
    float acc = 0.0f;
    for (int m = 0; m < mbs; ++m) {
        const int seg = col * mbs + m;
        const float a = key2f(st[seg * R + 2 * j].prefix), b = key2f(st[seg * R + 2 * j + 1].prefix);
        const float wt = w[j];
        // ATen lerp (vectorised CPU form): weight < 0.5 ? fma(w, b-a, a) : fma(w-1, b-a, b)
        const float d = b - a;
        const float v = (fabsf(wt) < 0.5f) ? fmaf(wt, d, a) : fmaf(wt - 1.0f, d, b);
        acc += v;
    }
    out[(int64_t)j * cols + col] = mbs == 1 ? acc : acc / (float)mbs;
}

// order-statistic mode (positive_percentile): out[r][seg] = value, or 0 when nothing was selected
__global__ __launch_bounds__(256) void k_sel_value_out(const SelState* __restrict__ st, int S, int R,
                                                       float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= S * R) return;
    const int seg = i / R, r = i % R;
    out[(int64_t)r * S + seg] = st[i].remaining < 0 ? 0.0f : key2f(st[i].prefix);
}

int run_select(const float* x, int64_t S, int64_t n, int R, const int64_t* d_ranks, const float* d_qfrac,
               int positive_only, SelState* st, unsigned* hist, hipStream_t stream) {
    hipError_t e = hipMemsetAsync(hist, 0, sizeof(unsigned) * S * R * 256, stream);
    if (e != hipSuccess) { adalog_set_error("select/memset", e); return (int)e; }
    const int nsr = (int)(S * R);
    hipLaunchKernelGGL(k_sel_init, dim3(cdiv(nsr, 256)), dim3(256), 0, stream, st, d_ranks, (int)S, R);
    int64_t bps = (n + 256 * 16 - 1) / (256 * 16);
    if (bps > 512) bps = 512;
    if (bps < 1) bps = 1;
    while (bps * S > 65535LL * 16 && bps > 1) bps /= 2;
    for (int pass = 0; pass < 4; ++pass) {
        hipLaunchKernelGGL(k_sel_hist, dim3((unsigned)bps, (unsigned)S), dim3(256), 0, stream, x, n, R, pass, st, hist,
                           positive_only, 0, 1, 1);
        hipLaunchKernelGGL(k_sel_pick, dim3(cdiv(nsr, 4)), dim3(256), 0, stream, st, hist, (int)S, R, pass, d_qfrac,
                           positive_only);
    }
    return 0;
}

}  // namespace

extern "C" int64_t adalog_select_workspace_bytes(int64_t S, int R) {
    return (int64_t)S * R * (256 * sizeof(unsigned) + sizeof(SelState)) + 256;
}

// x: contiguous [S][n].  ranks_lo_hi: device int64 [2*nq] = {floor(pos_0), ceil(pos_0), floor(pos_1), ...};
// weights: device fp32 [nq] = pos_j - floor(pos_j).  out: [nq][S/mbs].
extern "C" int adalog_quantile_rows(const float* x, int64_t S, int64_t n, int nq, const int64_t* ranks_lo_hi,
                                    const float* weights, int mbs, float* out, void* workspace, int64_t workspace_bytes,
                                    void* stream) {
    ADALOG_ARG_CHECK(x && ranks_lo_hi && weights && out && workspace, "quantile_rows: null pointer");
    ADALOG_ARG_CHECK(S >= 1 && S <= 65535 && n >= 1 && nq >= 1 && 2 * nq <= MAXR && mbs >= 1 && S % mbs == 0,
                     "quantile_rows: bad sizes");
    const int R = 2 * nq;
    ADALOG_ARG_CHECK(workspace_bytes >= adalog_select_workspace_bytes(S, R), "quantile_rows: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    unsigned* hist = (unsigned*)workspace;
    SelState* state = (SelState*)((char*)workspace + ((sizeof(unsigned) * S * R * 256 + 15) / 16) * 16);
    int rc = run_select(x, S, n, R, ranks_lo_hi, nullptr, 0, state, hist, st);
    if (rc) return rc;
    const int cols = (int)(S / mbs);
    hipLaunchKernelGGL(k_sel_quantile_out, dim3(cdiv((int64_t)cols * nq, 256)), dim3(256), 0, st, state, R, nq, weights, cols,
                       mbs, out);
    ADALOG_LAUNCH_CHECK("adalog_quantile_rows");
    return 0;
}

// out[r][S] = value of rank ceil(count*q_r)-1 among the positive entries of each row (0 if there are none)
extern "C" int adalog_positive_percentile_rows(const float* x, int64_t S, int64_t n, int nq, const float* qfrac, float* out,
                                               void* workspace, int64_t workspace_bytes, void* stream) {
    ADALOG_ARG_CHECK(x && qfrac && out && workspace, "positive_percentile_rows: null pointer");
    ADALOG_ARG_CHECK(S >= 1 && S <= 65535 && n >= 1 && nq >= 1 && nq <= MAXR, "positive_percentile_rows: bad sizes");
    ADALOG_ARG_CHECK(workspace_bytes >= adalog_select_workspace_bytes(S, nq), "positive_percentile_rows: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    unsigned* hist = (unsigned*)workspace;
    SelState* state = (SelState*)((char*)workspace + ((sizeof(unsigned) * S * nq * 256 + 15) / 16) * 16);
    int rc = run_select(x, S, n, nq, nullptr, qfrac, 1, state, hist, st);
    if (rc) return rc;
    hipLaunchKernelGGL(k_sel_value_out, dim3(cdiv(S * nq, 256)), dim3(256), 0, st, state, (int)S, nq, out);
    ADALOG_LAUNCH_CHECK("adalog_positive_percentile_rows");
    return 0;
}

// ---------------------------------------------------------------------------------------------- sharded form (multi-GPU)
// The same four radix passes, split so that the caller can sum the histograms of all ranks between a pass's counting and
// its pick (torch.distributed all_reduce on the first S*R*256 uint32 of the workspace; integer sums: every rank then
// descends identically).  S, R describe the GLOBAL segments; a rank counts only the rows it holds.
static inline unsigned* ws_hist(void* ws) { return (unsigned*)ws; }
static inline SelState* ws_state(void* ws, int64_t S, int R) {
    return (SelState*)((char*)ws + ((sizeof(unsigned) * S * R * 256 + 15) / 16) * 16);
}

extern "C" int adalog_select_init(void* workspace, int64_t workspace_bytes, int64_t S, int R, const int64_t* ranks, void* stream) {
    ADALOG_ARG_CHECK(workspace && S >= 1 && S <= 65535 && R >= 1 && R <= MAXR, "select_init: bad arguments");
    ADALOG_ARG_CHECK(workspace_bytes >= adalog_select_workspace_bytes(S, R), "select_init: workspace too small");
    hipError_t e = hipMemsetAsync(workspace, 0, sizeof(unsigned) * S * R * 256, (hipStream_t)stream);
    if (e != hipSuccess) { adalog_set_error("select_init/memset", e); return (int)e; }
    hipLaunchKernelGGL(k_sel_init, dim3(cdiv(S * R, 256)), dim3(256), 0, (hipStream_t)stream, ws_state(workspace, S, R), ranks,
                       (int)S, R);
    ADALOG_LAUNCH_CHECK("adalog_select_init");
    return 0;
}

// x: this rank's rows, contiguous [S_local][n_local]; row s counts into global segment first + (s / inner) * outer + s % inner
extern "C" int adalog_select_hist(const float* x, int64_t S_local, int64_t n_local, int first, int inner, int outer,
                                  int64_t S, int R, int pass, int positive_only, void* workspace, void* stream) {
    ADALOG_ARG_CHECK(x && workspace && S_local >= 1 && S_local <= 65535 && n_local >= 1 && inner >= 1 && pass >= 0 && pass < 4,
                     "select_hist: bad arguments");
    ADALOG_ARG_CHECK(first >= 0 && first + ((S_local - 1) / inner) * outer + (S_local - 1) % inner < S, "select_hist: slot out of range");
    int64_t bps = (n_local + 256 * 16 - 1) / (256 * 16);
    if (bps > 512) bps = 512;
    if (bps < 1) bps = 1;
    while (bps * S_local > 65535LL * 16 && bps > 1) bps /= 2;
    hipLaunchKernelGGL(k_sel_hist, dim3((unsigned)bps, (unsigned)S_local), dim3(256), 0, (hipStream_t)stream, x, n_local, R, pass,
                       ws_state(workspace, S, R), ws_hist(workspace), positive_only, first, inner, outer);
    ADALOG_LAUNCH_CHECK("adalog_select_hist");
    return 0;
}

extern "C" int adalog_select_pick(void* workspace, int64_t S, int R, int pass, const float* qfrac, int positive_only, void* stream) {
    ADALOG_ARG_CHECK(workspace && (!positive_only || qfrac), "select_pick: bad arguments");
    hipLaunchKernelGGL(k_sel_pick, dim3(cdiv(S * R, 4)), dim3(256), 0, (hipStream_t)stream, ws_state(workspace, S, R),
                       ws_hist(workspace), (int)S, R, pass, qfrac, positive_only);
    ADALOG_LAUNCH_CHECK("adalog_select_pick");
    return 0;
}

// quantile: out [nq][S / mbs] (R = 2 * nq ranks per segment);  value: out [R][S] (0 where nothing was selected)
extern "C" int adalog_select_quantile_out(void* workspace, int64_t S, int nq, const float* weights, int mbs, float* out, void* stream) {
    ADALOG_ARG_CHECK(workspace && weights && out && mbs >= 1 && S % mbs == 0, "select_quantile_out: bad arguments");
    const int cols = (int)(S / mbs);
    hipLaunchKernelGGL(k_sel_quantile_out, dim3(cdiv((int64_t)cols * nq, 256)), dim3(256), 0, (hipStream_t)stream,
                       ws_state(workspace, S, 2 * nq), 2 * nq, nq, weights, cols, mbs, out);
    ADALOG_LAUNCH_CHECK("adalog_select_quantile_out");
    return 0;
}

extern "C" int adalog_select_value_out(void* workspace, int64_t S, int R, float* out, void* stream) {
    ADALOG_ARG_CHECK(workspace && out, "select_value_out: bad arguments");
    hipLaunchKernelGGL(k_sel_value_out, dim3(cdiv(S * R, 256)), dim3(256), 0, (hipStream_t)stream, ws_state(workspace, S, R),
                       (int)S, R, out);
    ADALOG_LAUNCH_CHECK("adalog_select_value_out");
    return 0;
}
