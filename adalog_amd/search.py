"""Device-resident Fast Progressive Combining Search (FPCS) engine.

Host-side driver for the searches of reference quant_layers/linear.py:483-523, matmul.py:243-262, conv.py:292-311
(coarse 128-candidate grid -> top-16 -> (16 survivors x 8 neighbours -> top-16) x 4 -> top-1, spacing /= 7.5 each step).
It only sequences kernels: candidate grids, scores, top-k and the committed winner all stay on the GPU, and nothing
in a search synchronises the host.  Candidate tensors are fp32 [P, cols] (candidate-major).

Multi-GPU: calibration images are sharded across ranks; every scoring call's score tensor is the sum over images of
per-image terms (linear.py:345,384,423; matmul.py:163,201,351; conv.py:255), so ranks all-reduce(SUM) the [P, cols]
scores (RCCL over xGMI) and then run the identical deterministic top-k -- no broadcast needed.
"""
from __future__ import annotations

from typing import Callable, Optional

import torch

from . import backend, parallel

_LIN_CACHE = {}


def linspace01(n: int, device) -> torch.Tensor:
    """torch.linspace(0, 1, n) evaluated on the host exactly as the reference does, cached on the device."""
    key = (n, str(device))
    if key not in _LIN_CACHE:
        _LIN_CACHE[key] = torch.linspace(0, 1, steps=n).to(device)
    return _LIN_CACHE[key]


def const_tensor(values, device, dtype=torch.float32) -> torch.Tensor:
    key = (tuple(values), str(device), dtype)
    if key not in _LIN_CACHE:
        _LIN_CACHE[key] = torch.tensor(values, dtype=dtype).to(device)
    return _LIN_CACHE[key]


# ------------------------------------------------------------------------------------------------ converged rounds
# The alternating rounds of a module's search (weights | activations, A | B: linear.py:538-541, matmul.py:275-277) re-run each
# output-MSE search with the OTHER operand's quantiser as it stands.  A search is a pure function of that quantiser (the
# captures, the weights and the percentile grid do not change inside a module's search), so when the other quantiser comes
# back bit-identical to what this search saw last time -- the rounds have converged, ~20 % of the Linear searches of a
# deit_small calibration -- its result is the one already committed and the search is not repeated.  One host read
# (torch.equal) per check.  ADALOG_SKIP_CONVERGED=0 re-runs everything, as the reference does.
SKIP_CONVERGED = __import__("os").environ.get("ADALOG_SKIP_CONVERGED", "1") != "0"
COLLECT_ROUND_STATS = False        # True: count unchanged inputs also when nothing is skipped (lab / tests)
ROUND_STATS = {"checked": 0, "unchanged": 0}
_ROUND_PENDING = []                # device flags of checks made while nothing is skipped (read by round_stats(), not by the search)


def round_stats():
    """{"checked", "unchanged"} so far; resolves the comparisons that were left on the device (one host read)"""
    if _ROUND_PENDING:
        ROUND_STATS["unchanged"] += int(torch.stack(_ROUND_PENDING).sum().item())
        _ROUND_PENDING.clear()
    return dict(ROUND_STATS)


def reset_round_stats():
    _ROUND_PENDING.clear()
    ROUND_STATS.update(checked=0, unchanged=0)


def quantizer_state(*quantizers):
    """the bits of the tensors that define calibrated quantisers (uniform: scale, zero point; AdaLog: scale, base q), as ONE
    flat int32 tensor: a state comparison is then one kernel and one host read"""
    parts = []
    for q in quantizers:
        # every tensor a search reads from the other operand's quantiser: scale, zero point, AdaLog base and the shift
        for t in (getattr(q, "scale", None), getattr(q, "zero_point", None), getattr(q, "q", None), getattr(q, "shift", None)):
            if torch.is_tensor(t):
                t = t.detach().reshape(-1)
                parts.append(t.view(torch.int32) if t.dtype == torch.float32 else t.to(torch.int32))
        parts_bits = getattr(q, "n_bits", None)
        if parts_bits is not None and parts:
            parts.append(torch.full((1,), int(parts_bits), dtype=torch.int32, device=parts[-1].device))
    if not parts:
        raise ValueError("quantizer_state: no calibrated tensor to snapshot")
    return torch.cat(parts) if len(parts) != 1 else parts[0].clone()


def begin_rounds(module):
    """call at the start (and end) of a module's hyperparameter search: forgets what its searches have seen"""
    module.__dict__["_round_inputs"] = {}


def round_is_redundant(module, tag: str, *quantizers) -> bool:
    """True when the search `tag` of this module last ran against exactly this state of the quantisers it reads (the other
    operand's; its own too when the search starts from its current parameters) -- then its committed result stands.
    Records the state either way."""
    if not SKIP_CONVERGED and not COLLECT_ROUND_STATS:
        return False                       # every search runs and nobody asked for the statistics: no snapshots, no compares
    seen = module.__dict__.setdefault("_round_inputs", {})
    state = quantizer_state(*quantizers)
    prev = seen.get(tag)
    seen[tag] = state
    if prev is None:
        return False
    ROUND_STATS["checked"] += 1
    if prev.shape != state.shape:
        return False
    if not SKIP_CONVERGED:                 # every search runs anyway: the comparison stays on the device, the host does not wait
        _ROUND_PENDING.append((prev == state).all())
        return False
    same = torch.equal(prev, state)
    ROUND_STATS["unchanged"] += int(same)
    return same


# A/B switch: 0 = scorers that can run the FPCS tail in their own launch hand their scores to a separate k_topk_next instead
FUSED_TAIL = __import__("os").environ.get("ADALOG_FUSED_TAIL", "1") != "0"


def fpcs(scale, zp, third, delta, score_fn: Callable, steps: int, width: int = 16, eq_n: int = 128,
         clamp_min: Optional[float] = None, commit_to=None):
    """Run the progressive search.  ``score_fn(scale, zp, third) -> scores [P, cols]`` (rank-local partial sums).

    ``score_fn`` may return an ``ops.PendingScores`` (partial sums not yet reduced) instead of the scores.  A ``score_fn`` with the
    attribute ``fused_tail`` also takes ``tail=`` (an ``ops.FpcsTail``): its kernel then ranks the scores and writes the next grid /
    commits the winner in the SAME launch (one GPU only: with several ranks the scores are all-reduced before the ranking).
    ``delta`` (the grid spacing, memoised with the grid) is only read: the narrowed spacing goes to a buffer of this call.
    ``commit_to``: (scale, zp | None, third | None) contiguous fp32 [cols] tensors -- the quantiser's own parameter storage -- that the
    last step's kernel writes the winner into (no copy afterwards).
    Returns the committed (scale [cols], zp [cols] | None, third [cols] | None); with steps == 1 nothing is committed
    (the reference's loop never reaches its top-1 branch then) and None is returned.
    """
    be = backend.get()
    new_cnt = int(eq_n / width)
    lin = linspace01(new_cnt, scale.device)
    remain = steps
    first = True
    pending_t = getattr(be, "PendingScores", ())
    Tail = getattr(be, "FpcsTail", None)
    one_gpu = Tail is not None and not parallel.is_dist()
    fused = one_gpu and FUSED_TAIL and getattr(score_fn, "fused_tail", False)
    # a scorer whose state was all-reduced when it was built (the Gram form of the weight searches: G, c, S0) returns the GLOBAL
    # scores on every rank: its steps need no collective; every other scorer's [P, cols] scores are rank-local partial sums
    global_scores = getattr(score_fn, "global_scores", False)
    d_in, d_buf = delta, None
    cols = scale.shape[1]

    def tail_for(last):
        nonlocal d_buf
        if last:
            return Tail(scale, zp, third, 1, 0, None, None, None, None, out=commit_to)
        if d_buf is None:
            d_buf = torch.empty_like(delta)
        return Tail(scale, zp, third, width, new_cnt, lin, d_in, d_buf, clamp_min)

    while remain > 0:
        last = (remain == 1) and not first
        if remain == 1 and first:    # steps == 1: survivors are selected but never committed (linear.py:490-491)
            score_fn(scale, zp, third)
            return None
        if one_gpu and not global_scores:
            parallel.note_planned(4 * scale.shape[0] * cols)       # (what an N-rank run all-reduces at this step)
        if fused:
            tail = tail_for(last)
            score_fn(scale, zp, third, tail=tail)
            res = None
        else:
            res = score_fn(scale, zp, third)
            if one_gpu:
                tail = tail_for(last)
                if isinstance(res, pending_t):
                    # the scoring kernel's partial sums are reduced, ranked and expanded into the next grid by ONE launch
                    be.finish_topk_next(res, None, None, None, 0, 0, None, None, None, tail=tail)
                else:
                    be.topk_next(res, None, None, None, 0, 0, None, None, None, tail=tail)
            else:
                # several ranks (or a backend without tails): the scores are all-reduced between the reduction and the ranking
                scores = res.finish() if isinstance(res, pending_t) else res
                if not global_scores:
                    scores = parallel.all_reduce_sum(scores)
                if last:
                    return be.topk_next(scores, scale, zp, third, 1, 0, None, None, None)
                if d_buf is None:
                    d_buf = delta.clone()            # (topk_next narrows its delta in place: never the memoised one)
                scale, zp, third = be.topk_next(scores, scale, zp, third, width, new_cnt, lin, d_buf, clamp_min)
                remain -= 1
                first = False
                continue
        if last:
            return tail.result()
        scale, zp, third = tail.result()
        d_in = d_buf
        remain -= 1
        first = False
    return None


def commit_param(param, value):
    """param <- value, unless the search's last kernel already wrote the winner into the parameter's storage (fpcs commit_to)"""
    if value.data_ptr() != param.data_ptr():
        param.data.copy_(value.view(param.shape))


def commit_targets(*params):
    """the flat fp32 views of the parameters a search commits to (fpcs commit_to), None for a plane the search does not have"""
    out = []
    for p_ in params:
        if p_ is None:
            out.append(None)
            continue
        d = p_.data
        if not (d.is_contiguous() and d.dtype == torch.float32):
            return None
        out.append(d.view(-1))
    return tuple(out)


def honour_tail(scores, tail):
    """a scorer that was handed an FPCS tail but produced plain scores: rank / expand / commit them in one more launch"""
    if tail is not None:
        backend.get().topk_next(scores, None, None, None, 0, 0, None, None, None, tail=tail)
    return scores


def argbest(scores, k: int = 1):
    """top-k candidate indices [k, cols] of an (all-reduced) score tensor."""
    return backend.get().topk(parallel.all_reduce_sum(scores), k)


# ------------------------------------------------------------------------------------------------ percentile grids
QS_HI = (0.9, 1.0)


def _pct_lists(lo=0.9, hi=1.0):
    pct = torch.tensor([lo, hi])
    return pct.tolist() + (1 - pct).tolist()            # fp32 arithmetic for 1 - pct, as in linear.py:439-441


def _chunk_rows(numel: int) -> int:
    """Rows ``mbs`` of the reference's chunked per-tensor quantile: the smallest power of two for which
    x.view(mbs, -1) is legal and the reduced dim fits torch.quantile's 2**24 limit (linear.py:465-471)."""
    mbs = 1
    while numel % mbs != 0 or numel // mbs > 16777216:
        mbs *= 2
        if mbs > numel:
            raise ValueError("cannot chunk tensor for quantile")
    return mbs


def _sharded_quantiles(x2, S: int, first: int, inner: int, outer: int, n_total: int, mbs: int):
    """torch.quantile over GLOBAL segments whose elements are spread over the ranks, without moving the data: every rank
    histograms its rows (4 radix passes), the [S, R, 256] int32 histograms are all-reduced, and all ranks descend
    identically (exact distributed radix select).  Returns [4, S / mbs] like backend.quantile_rows."""
    be = backend.get()
    qs = _pct_lists()
    lohi, w = be.quantile_ranks(qs, n_total)
    sel = be.ShardedSelect(x2, S, 2 * len(qs), first, inner, outer, ranks=lohi)
    for p in range(4):
        sel.hist_pass(p)
        parallel.all_reduce_sum(sel.hist)
        sel.pick(p)
    return sel.quantiles(w, mbs)


def _flat_layout(local_numel: int, rows: int = 1):
    """Where this rank's contiguous image shard sits among the reference's quantile chunks (x.view(mbs, -1) of the GLOBAL
    tensor, per `rows` leading rows such as heads): -> (mbs, local_rows_per_row, first, inner, outer, n_total) or None
    when chunk and shard boundaries do not nest (then the shards are gathered instead)."""
    ws, r = parallel.world_size(), parallel.rank()
    per_row_global = ws * (local_numel // rows)
    mbs = _chunk_rows(per_row_global)
    n_total = per_row_global // mbs
    if mbs <= ws and ws % mbs == 0:                        # a chunk spans ws/mbs ranks
        return mbs, 1, r // (ws // mbs), 1, mbs, n_total
    if mbs % ws == 0:                                      # a rank holds mbs/ws whole chunks
        per = mbs // ws
        return mbs, per, r * per, per, mbs, n_total
    return None


class _Memo(dict):
    pass


_memo_tls = __import__("threading").local()


def _memo_dict():
    """one memo per host thread: the calibrator runs two modules' searches side by side (adalog_amd.parallel lanes), and a
    module that finishes must only forget its own entries"""
    d = getattr(_memo_tls, "d", None)
    if d is None:
        d = _memo_tls.d = _Memo()
    return d



def _memo(kind, x, extra, make):
    """The percentile grid depends only on the tensor it is computed from, and every search round asks for it again
    (1 + search_round times per operand).  Keyed by storage pointer + in-place version + shape: a re-parameterised
    weight or a new raw_input gives a new key.  The entry holds a reference to the tensor it was computed from, so its
    storage cannot be freed and handed to another tensor (same address, same version) while the entry lives.
    ``delta`` is shared too: the FPCS driver reads it and narrows a buffer of its own.  Entries die with their tensor's
    search (forget_grids)."""
    key = (kind, x.data_ptr(), x._version, tuple(x.shape), tuple(x.stride())) + tuple(extra)
    memo = _memo_dict()
    hit = memo.get(key)
    if hit is None:
        hit = memo[key] = (x, make())
    hit = hit[1]
    return hit[0], hit[1], hit[2]          # (delta is read-only for its users: search.fpcs narrows a buffer of its own)


def memo_tensor_fn(kind, x, extra, make):
    """Same memo for other pure functions of a captured tensor (post-GELU positive percentiles, the sorted copy + prefix
    sums of the self-MSE searches); result returned as is."""
    key = (kind, x.data_ptr(), x._version, tuple(x.shape), tuple(x.stride())) + tuple(extra)
    memo = _memo_dict()
    if key not in memo:
        memo[key] = (x, make())
    return memo[key][1]


def forget_grids():
    """Drop the memoised grids (called when a module's search ends and its captures are released)."""
    _memo_dict().clear()


def weight_grid(w2, n_bits: int, eq_n: int, conv: bool = False):
    """linear.py:432-451 / conv.py:271-290 -> (scale [P, rows], zp [P, rows], delta [rows])."""
    return _memo("w", w2, (n_bits, eq_n, conv), lambda: _weight_grid(w2, n_bits, eq_n, conv))


def _weight_grid(w2, n_bits: int, eq_n: int, conv: bool = False):
    be = backend.get()
    L = 2 ** (n_bits - 1)
    num_zp = L if conv else min(16, L)
    num_scale = int(eq_n / num_zp)
    quant4 = be.quantile_rows(w2, _pct_lists(), 1)           # weights are replicated on every rank
    return be.candidate_grid(quant4, num_scale, num_zp, int(L - num_zp / 2), n_bits, linspace01(num_scale, w2.device), None)


def activation_grid(x, n_bits: int, eq_n: int, channel_wise: bool):
    return _memo("a", x, (n_bits, eq_n, channel_wise), lambda: _activation_grid(x, n_bits, eq_n, channel_wise))


def _activation_grid(x, n_bits: int, eq_n: int, channel_wise: bool):
    """linear.py:453-481 -> (scale [P, C], zp [P, C], delta [C]), C = in_features or 1.  ``x`` is the rank-local shard;
    quantiles are global order statistics: with several ranks they come from a distributed radix select (histograms
    all-reduced, data stays put)."""
    be = backend.get()
    L = 2 ** (n_bits - 1)
    num_zp = min(16, 2 * L)
    num_scale = int(eq_n / num_zp)
    lay = _flat_layout(x.numel()) if parallel.is_dist() else None
    if parallel.is_dist() and channel_wise:
        C = x.shape[-1]
        x2 = x.reshape(-1, C).t().contiguous()              # [C, local rows]: every rank holds a part of every channel
        quant4 = _sharded_quantiles(x2, C, 0, C, 0, parallel.world_size() * x2.shape[1], 1)
    elif lay is not None:
        mbs, per, first, inner, outer, n_total = lay
        quant4 = _sharded_quantiles(x.reshape(per, -1), mbs, first, inner, outer, n_total, mbs)
    else:
        xg = parallel.gather_images(x)
        if channel_wise:
            x2 = xg.reshape(-1, xg.shape[-1]).t().contiguous()
            quant4 = be.quantile_rows(x2, _pct_lists(), 1)
        else:
            mbs = _chunk_rows(xg.numel())
            quant4 = be.quantile_rows(xg.reshape(mbs, -1), _pct_lists(), mbs)
    return be.candidate_grid(quant4, num_scale, num_zp, int(L - num_zp / 2), n_bits, linspace01(num_scale, x.device), 1e-4)


def matmul_grid(x, n_bits_B: int, eq_n: int, head_wise: bool = True):
    return _memo("m", x, (n_bits_B, eq_n, head_wise), lambda: _matmul_grid(x, n_bits_B, eq_n, head_wise))


def _matmul_grid(x, n_bits_B: int, eq_n: int, head_wise: bool = True):
    """matmul.py:211-240 -> (scale [P, H], zp [P, H], delta [H]); both operands use B's level count."""
    be = backend.get()
    L = 2 ** (n_bits_B - 1)
    num_zp = min(16, L)
    num_scale = int(eq_n / num_zp)
    H = x.shape[1] if head_wise else 1
    lay = _flat_layout(x.numel(), H) if parallel.is_dist() else None
    if lay is not None:
        mbs, per, first, inner, outer, n_total = lay
        xl = x.transpose(0, 1).contiguous() if head_wise else x
        quant4 = _sharded_quantiles(xl.reshape(H * per, -1), H * mbs, first, inner, outer, n_total, mbs)
    else:
        xg = parallel.gather_images(x)
        if head_wise:
            xt = xg.transpose(0, 1).contiguous()
            mbs = _chunk_rows(xt.numel() // H)
            x2 = xt.view(H * mbs, -1)
        else:
            mbs = _chunk_rows(xg.numel())
            x2 = xg.reshape(mbs, -1)
        quant4 = be.quantile_rows(x2, _pct_lists(), mbs)
    return be.candidate_grid(quant4, num_scale, num_zp, int(L - num_zp / 2), n_bits_B, linspace01(num_scale, x.device), None)


def int_operand_dtype(bits_a: int, bits_b: int, chunk: int, zp_on_grid: bool = True, prefer_fp8: bool = False) -> int:
    """Storage type for a pair of uniformly quantised GEMM operands (ops.I8 or ops.FP8).

    fp8 (e4m3) holds every integer in [-16, 16] exactly, so ``q - z`` of a <= 4-bit operand whose zero point lies in
    [0, 2^bits - 1] (true for every FPCS grid, linear.py:446-449,476-479; matmul.py:236-239) is exact, products accumulate
    exactly in the fp32 MFMA accumulator (sums < 2^24), the f8f6f4 MFMA runs at the int8 rate and the epilogue needs no
    int->float conversion.  Requires all candidates of a call in one launch (64, 128 or 256).

    Measured on MI355X (deit_small W4A4): identical scores everywhere.  Whether it is faster depends on how much of a
    launch is epilogue: the attention q.k^T searches (one 64-byte K-step, group kernel) and the weight searches of the
    K <= 384 linear layers (slab kernel without a row scale: 2.29 -> 2.49 PFLOP/s) gain, so their callers pass
    ``prefer_fp8``; so do, since the 128-column slabs, the activation searches of K <= 768 (+2 %); the streaming kernel's
    launches (bound by the operand stream) do not.  ADALOG_INT_FP8=1 / 0 forces fp8 wherever it is exact /
    nowhere."""
    import os
    from .ops import FP8, I8
    env = os.environ.get("ADALOG_INT_FP8", "")
    want = env == "1" or (prefer_fp8 and env != "0")
    if want and bits_a <= 4 and bits_b <= 4 and zp_on_grid and chunk in (64, 128, 256):
        return FP8
    return I8
