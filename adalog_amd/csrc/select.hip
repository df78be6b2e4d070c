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
#include "fpcs_tail.h"
#include <stdlib.h>

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

// The descent of k_sel_pick as a device function of ONE wavefront over a 256-bin histogram held in four values per lane (bins
// 4 lane .. 4 lane + 3), wherever it was read from (LDS: the one-block-per-segment kernel; global memory at agent scope: the last
// block of a segment in k_sel_hist_pick).
__device__ __forceinline__ SelState pick_wave(uint4 c, SelState s, int pass, int positive_only, float qf, int lane) {
    const int64_t own = (int64_t)c.x + c.y + c.z + c.w;
    int64_t incl = own;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int64_t up = __shfl_up(incl, d);
        if (lane >= d) incl += up;
    }
    const int64_t total = __shfl(incl, 63);
    if (pass == 0 && positive_only) {                      // (see k_sel_pick: linear.py:785-790 and the > 2^24 positives quirk)
        const float cq = ceilf((float)total * qf);
        const int64_t rk = (int64_t)cq - 1;
        s.remaining = (total == 0 || rk >= total) ? -1 : (rk < 0 ? 0 : rk);
    }
    if (s.remaining >= 0) {
        const unsigned long long hit = __ballot(incl > s.remaining);
        int b = 255;
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
    return s;
}

// ATen lerp (vectorised CPU form): weight < 0.5 ? fma(w, b-a, a) : fma(w-1, b-a, b)
__device__ __forceinline__ float aten_lerp(float a, float b, float wt) {
    const float d = b - a;
    return (fabsf(wt) < 0.5f) ? fmaf(wt, d, a) : fmaf(wt - 1.0f, d, b);
}

// ONE launch for a whole select (round 6): a workgroup per segment walks its row four times (the row is L2-resident after the first
// pass), histograms into LDS, its waves descend the R states, and it writes the segment's outputs -- 11 launches (memset, init, 4 x
// (count, pick), output) before.  For many short segments (weight rows: n = K) and for per-channel activations (S = channels, n =
// tokens: S blocks fill the chip).  mode 0: quantiles out[j][seg] = lerp(v[2 j], v[2 j + 1], w[j]) (mbs == 1); mode 1: order
// statistics out[r][seg] (0 where nothing was selected).
__global__ __launch_bounds__(256) void k_sel_one_block(const float* __restrict__ x, int64_t n, int S, int R, const int64_t* __restrict__ ranks,
                                                       const float* __restrict__ qfrac, int positive_only, int mode, int nq,
                                                       const float* __restrict__ w, float* __restrict__ out) {
    __shared__ unsigned h[MAXR][256];
    __shared__ SelState sst[MAXR];
    const int seg = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid < R) { sst[tid].prefix = 0; sst[tid].remaining = ranks ? ranks[tid] : 0; }
    const float* xs = x + (int64_t)seg * n;
    for (int pass = 0; pass < 4; ++pass) {
        for (int i = tid; i < R * 256; i += 256) (&h[0][0])[i] = 0;
        __syncthreads();
        const int shift = 24 - 8 * pass;
        uint32_t pre[MAXR];
#pragma unroll
        for (int r = 0; r < MAXR; ++r) pre[r] = r < R ? sst[r].prefix : 0xffffffffu;
        for (int64_t i = tid; i < n; i += 256) {
            const float v = xs[i];
            if (positive_only && !(v > 0.0f)) continue;
            const uint32_t key = f2key(v);
            const unsigned bin = (key >> shift) & 255u;
            if (pass == 0) {
                atomicAdd(&h[0][bin], 1u);
            } else {
                const uint32_t hi = key >> (shift + 8);
#pragma unroll
                for (int r = 0; r < MAXR; ++r)
                    if (r < R && hi == pre[r]) atomicAdd(&h[r][bin], 1u);
            }
        }
        __syncthreads();
        for (int r = wv; r < R; r += 4) {
            const unsigned* hp = pass == 0 ? h[0] : h[r];
            const uint4 c = make_uint4(hp[4 * lane], hp[4 * lane + 1], hp[4 * lane + 2], hp[4 * lane + 3]);
            const SelState s2 = pick_wave(c, sst[r], pass, positive_only, qfrac ? qfrac[r] : 0.0f, lane);
            if (lane == 0) sst[r] = s2;
        }
        __syncthreads();
    }
    if (mode == 0) {
        if (tid < nq) out[(int64_t)tid * S + seg] = aten_lerp(key2f(sst[2 * tid].prefix), key2f(sst[2 * tid + 1].prefix), w[tid]);
    } else {
        if (tid < R) out[(int64_t)tid * S + seg] = sst[tid].remaining < 0 ? 0.0f : key2f(sst[tid].prefix);
    }
}

// hist + state initialisation in one launch (a memset and k_sel_init before)
__global__ __launch_bounds__(256) void k_sel_clear(unsigned* __restrict__ hist, int64_t words, SelState* st, const int64_t* ranks, int nsr, int R) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < words) hist[i] = 0u;
    if (i < nsr) { st[i].prefix = 0; st[i].remaining = ranks ? ranks[i % R] : 0; }
}

// k_sel_hist with the pick folded in (single process): the last of a segment's blocks -- a ticket per segment -- descends the
// segment's R states from the finished global histogram and clears it for the next pass.
__global__ __launch_bounds__(256) void k_sel_hist_pick(const float* __restrict__ x, int64_t n, int R, int pass, SelState* st,
                                                       unsigned* __restrict__ hist, int positive_only,
                                                       const float* __restrict__ qfrac, unsigned int* tickets) {
    __shared__ unsigned h[MAXR][256];
    __shared__ uint32_t pre[MAXR];
    __shared__ int is_last;
    const int seg = blockIdx.y;
    for (int i = threadIdx.x; i < R * 256; i += blockDim.x) (&h[0][0])[i] = 0;
    if ((int)threadIdx.x < R) pre[threadIdx.x] = st[seg * R + threadIdx.x].prefix;
    __syncthreads();
    const int shift = 24 - 8 * pass;
    const float* xs = x + (int64_t)seg * n;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = xs[i];
        if (positive_only && !(v > 0.0f)) continue;
        const uint32_t key = f2key(v);
        const unsigned bin = (key >> shift) & 255u;
        if (pass == 0) {
            atomicAdd(&h[0][bin], 1u);
        } else {
            const uint32_t hi = key >> (shift + 8);
            for (int r = 0; r < R; ++r)
                if (hi == pre[r]) atomicAdd(&h[r][bin], 1u);
        }
    }
    __syncthreads();
    unsigned* gh = hist + ((int64_t)seg * R) * 256;
    for (int i = threadIdx.x; i < R * 256; i += blockDim.x) {
        const unsigned c = pass == 0 ? h[0][i & 255] : (&h[0][0])[i];
        if (c) __hip_atomic_fetch_add(gh + i, c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // (agent-scope atomics are performed at the coherence point: once they have returned -- vmcnt -- the counts are visible to the
    // segment's last block; a __threadfence() here writes back the XCD's whole L2 and cost 35 us per pass)
    fpcs::publish_wait();
    __syncthreads();
    if (threadIdx.x == 0) is_last = fpcs::ticket_last(tickets + seg, gridDim.x) ? 1 : 0;
    __syncthreads();
    if (!is_last) return;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int r = wv; r < R; r += 4) {
        unsigned* hp = gh + r * 256 + 4 * lane;
        uint4 c;
        c.x = __hip_atomic_load(hp + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        c.y = __hip_atomic_load(hp + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        c.z = __hip_atomic_load(hp + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        c.w = __hip_atomic_load(hp + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(hp + 0, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(hp + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(hp + 2, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(hp + 3, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int i = seg * R + r;
        const SelState s2 = pick_wave(c, st[i], pass, positive_only, qfrac ? qfrac[r] : 0.0f, lane);
        if (lane == 0) st[i] = s2;
    }
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

// A select that one launch finishes: a workgroup per segment (k_sel_one_block).  Taken when the segments alone fill the chip or the rows
// are short; mbs == 1 only (the chunk mean runs over several segments).
static bool one_block_ok(int64_t S, int64_t n, int mbs) {
    static const int off = getenv("ADALOG_SEL_ONE_BLOCK") ? atoi(getenv("ADALOG_SEL_ONE_BLOCK")) : 1;
    return off != 0 && mbs == 1 && (n <= 16384 || (S >= 256 && n <= 65536));
}

int run_select(const float* x, int64_t S, int64_t n, int R, const int64_t* d_ranks, const float* d_qfrac,
               int positive_only, SelState* st, unsigned* hist, hipStream_t stream) {
    const int nsr = (int)(S * R);
    const int64_t words = (int64_t)nsr * 256;
    int64_t bps = (n + 256 * 16 - 1) / (256 * 16);
    if (bps > 512) bps = 512;
    if (bps < 1) bps = 1;
    while (bps * S > 65535LL * 16 && bps > 1) bps /= 2;
    unsigned int* tickets = adalog_ticket_pool_on((int)S, stream);     // S <= 65535 (checked by the callers)
    static const int fused = getenv("ADALOG_SEL_FUSED") ? atoi(getenv("ADALOG_SEL_FUSED")) : 1;
    if (!tickets || !fused) {                              // the separate launches (also the sharded form's building blocks)
        hipError_t e = hipMemsetAsync(hist, 0, sizeof(unsigned) * words, stream);
        if (e != hipSuccess) { adalog_set_error("select/memset", e); return (int)e; }
        hipLaunchKernelGGL(k_sel_init, dim3(cdiv(nsr, 256)), dim3(256), 0, stream, st, d_ranks, (int)S, R);
        for (int pass = 0; pass < 4; ++pass) {
            hipLaunchKernelGGL(k_sel_hist, dim3((unsigned)bps, (unsigned)S), dim3(256), 0, stream, x, n, R, pass, st, hist,
                               positive_only, 0, 1, 1);
            hipLaunchKernelGGL(k_sel_pick, dim3(cdiv(nsr, 4)), dim3(256), 0, stream, st, hist, (int)S, R, pass, d_qfrac,
                               positive_only);
        }
        return 0;
    }
    hipLaunchKernelGGL(k_sel_clear, dim3(cdiv(words, 256)), dim3(256), 0, stream, hist, words, st, d_ranks, nsr, R);
    for (int pass = 0; pass < 4; ++pass)
        hipLaunchKernelGGL(k_sel_hist_pick, dim3((unsigned)bps, (unsigned)S), dim3(256), 0, stream, x, n, R, pass, st, hist,
                           positive_only, d_qfrac, tickets);
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
    if (one_block_ok(S, n, mbs)) {
        hipLaunchKernelGGL(k_sel_one_block, dim3((unsigned)S), dim3(256), 0, st, x, n, (int)S, R, ranks_lo_hi, nullptr, 0, 0, nq, weights, out);
        ADALOG_LAUNCH_CHECK("adalog_quantile_rows");
        return 0;
    }
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
    if (one_block_ok(S, n, 1)) {
        hipLaunchKernelGGL(k_sel_one_block, dim3((unsigned)S), dim3(256), 0, st, x, n, (int)S, nq, nullptr, qfrac, 1, 1, nq, nullptr, out);
        ADALOG_LAUNCH_CHECK("adalog_positive_percentile_rows");
        return 0;
    }
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
