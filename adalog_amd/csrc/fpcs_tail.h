// The tail of an FPCS step -- rank the P scores of a column, then write the next 16 x 8 candidate grid around the k survivors or
// commit the winner (reference quant_layers/linear.py:483-523, matmul.py:243-262, conv.py:292-311) -- as DEVICE code plus its
// arguments as a C struct, shared by every kernel that ends a step: k_topk_next (stand-alone), the finish kernels of the token forms
// (gemm_finish.inc: k_finish_tpo_topk / k_finish_wgacc_topk).  The struct carries what round 6 changed for ALL of them: the grid spacing is read from `delta_in` and
// written to `delta_out` (the memoised spacing is never cloned), and the commit step writes the winner straight into the quantiser's
// parameter storage (no copy_ launches afterwards).
// What was measured and NOT kept (same-box A/Bs, profiles/r06_notes.md): running the tail inside the kernels that produce final scores
// -- k_gram_score (a ticket per output row), k_ga_finish and k_score_sorted (tickets; or one 1 024-thread workgroup per score column
// ranking in LDS).  On this multi-XCD part an agent-scope ticket or score read is a ~2 us round trip, the last arrivals serialise
// the columns' tails, and a workgroup per column cannot issue its 2 176 divergent bisection loads as fast as the spread-out form:
// every variant lost to the separate 11 us k_topk_next launch.  Their `_tail` entry points run the two launches in one call.
// score_publish / ticket_last below serve k_sel_hist_pick (select.hip).
// Same deterministic order everywhere: score descending, candidate index ascending, NaN first (torch.topk with ties made
// deterministic).
#pragma once
#include "common.h"

// the C-ABI form of the arguments (include/adalog_hip.h declares the same struct)
extern "C" {
typedef struct adalog_fpcs_tail {
    int32_t k, new_cnt, has_clamp;
    float clamp_min;
    const float* scale; const float* zp; const float* third;    // the grid that was scored: [P][cols] (zp / third may be null)
    const float* lin;                                           // linspace(0, 1, new_cnt) (new_cnt > 0)
    const float* delta_in; float* delta_out;                    // [cols]: spacing read / spacing / (new_cnt - 0.5) written; may alias
    float* out_scale; float* out_zp; float* out_third;          // [k * new_cnt][cols], or the committed winner [cols] (new_cnt == 0)
} adalog_fpcs_tail;
}

namespace fpcs {

typedef adalog_fpcs_tail Tail;

// host-side check shared by the entry points that take a tail
static inline const char* tail_problem(const Tail* t, int P) {
    if (!t) return nullptr;
    if (!(t->scale && t->out_scale && t->k >= 1 && t->k <= P && P <= 256)) return "fpcs tail: needs scale, out_scale, 1 <= k <= P <= 256";
    if (t->new_cnt < 0 || (t->new_cnt > 0 && !(t->lin && t->delta_in && t->delta_out))) return "fpcs tail: the expansion needs lin and delta";
    if (t->new_cnt == 0 && t->k != 1) return "fpcs tail: the commit form takes k = 1";
    if ((t->zp == nullptr) != (t->out_zp == nullptr) || (t->third == nullptr) != (t->out_third == nullptr))
        return "fpcs tail: in / out parameter planes must match";
    if (t->new_cnt > 0 && (int64_t)t->k * t->new_cnt > 4096) return "fpcs tail: grid too large";
    return nullptr;
}

// rank of candidate me_i among the C scores s[0..C)
__device__ __forceinline__ int rank_of(const float* s, int C, int me_i) {
    const float me = s[me_i];
    const bool me_nan = me != me;
    int rank = 0;
    for (int j = 0; j < C; ++j) {
        const float o = s[j];
        const bool o_nan = o != o;
        bool before;
        if (o_nan || me_nan) before = (o_nan && !me_nan) || (o_nan && me_nan && j < me_i);
        else before = (o > me) || (o == me && j < me_i);
        rank += before ? 1 : 0;
    }
    return rank;
}

// the k survivors top[0..k) of column `col` -> committed winner (new_cnt == 0) or the next survivor-major grid; thread gt of GT
__device__ __forceinline__ void emit(const int* top, int cols, int col, int gt, int GT, float d, const Tail& t) {
#pragma clang fp contract(off)          // the grid values are parameters: scale + (lin - 0.5) * delta as separate IEEE operations
    if (t.new_cnt == 0) {
        if (gt == 0) {
            const int w = top[0];
            t.out_scale[col] = t.scale[(int64_t)w * cols + col];
            if (t.zp) t.out_zp[col] = t.zp[(int64_t)w * cols + col];
            if (t.third) t.out_third[col] = t.third[(int64_t)w * cols + col];
        }
        return;
    }
    const int total = t.k * t.new_cnt;
    for (int o = gt; o < total; o += GT) {
        const int j = o / t.new_cnt, i = o - j * t.new_cnt;
        const int w = top[j];
        float v = t.scale[(int64_t)w * cols + col] + (t.lin[i] - 0.5f) * d;           // linear.py:492-495
        if (t.has_clamp) v = fmaxf(v, t.clamp_min);                                  // linear.py:516
        const int64_t oo = (int64_t)o * cols + col;
        t.out_scale[oo] = v;
        if (t.zp) t.out_zp[oo] = t.zp[(int64_t)w * cols + col];
        if (t.third) t.out_third[oo] = t.third[(int64_t)w * cols + col];
    }
}

// Column `col` of scores [P][cols] by the GT threads of a group (gt = 0 .. GT - 1): GT = 256 with WAVE = false (a whole workgroup,
// __syncthreads) or GT = 64 with WAVE = true (one wavefront: its lanes run in lockstep, the LDS traffic only needs to have landed).
// `s` (>= P floats) and `top` (>= k ints) are LDS scratch of the group.  The scores are read at agent scope: other workgroups wrote
// them and announced it through the column's ticket (score_publish / ticket_last below).
template <int GT, bool WAVE, bool COHERENT = true>
__device__ __forceinline__ void column(const float* scores, int P, int cols, int col, int gt, const Tail& t, float* s, int* top) {
    auto sync = [&]() {
        if constexpr (WAVE) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }
        else __syncthreads();
    };
    for (int p = gt; p < P; p += GT) {
        // COHERENT: other workgroups of THIS launch wrote the scores (agent-scope loads); a stand-alone launch reads them plainly
        if constexpr (COHERENT) s[p] = __hip_atomic_load(scores + (int64_t)p * cols + col, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else s[p] = scores[(int64_t)p * cols + col];
    }
    const float d = t.new_cnt > 0 ? t.delta_in[col] : 0.0f;
    sync();
    for (int p = gt; p < P; p += GT) {
        const int r = rank_of(s, P, p);
        if (r < t.k) top[r] = p;
    }
    sync();
    emit(top, cols, col, gt, GT, d, t);
    if (t.new_cnt > 0) {
        sync();                                                                      // every thread holds its copy of d (delta may alias)
        if (gt == 0) t.delta_out[col] = d / ((float)t.new_cnt - 0.5f);               // linear.py:493
    }
}

// a producer's thread publishes one final score so that the column's last arrival can read it: agent-scope store, and the store has
// reached the coherence point before the ticket is drawn (a fence alone waits on lgkmcnt only)
__device__ __forceinline__ void score_publish(float* dst, float v) {
    __hip_atomic_store(dst, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void publish_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// one arrival at `counter` out of `expected`; true for the last one, which also puts the counter back to zero for the next launch
__device__ __forceinline__ bool ticket_last(unsigned int* counter, unsigned expected) {
    const unsigned tk = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tk != expected - 1) return false;
    __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return true;
}

}  // namespace fpcs

// zeroed device words for per-column tickets (brecq.hip): n <= 64 from the small ring, n <= 65536 from the wide pool of the stream
extern "C" unsigned int* adalog_ticket_pool_on(int n, void* stream);
