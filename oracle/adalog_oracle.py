"""CPU oracle for the AdaLog calibration hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module.  Nothing under ``adalog_amd/`` imports it; the
product path runs on hand-written HIP kernels and fails loudly without them.

What it is: a restatement, as plain functions over CPU tensors, of the
arithmetic on the reference's calibration path.  Each
function cites the reference ``file:line`` it follows.  It is written against
torch *CPU* tensor ops on purpose: the reference's CPU path executes exactly
these ATen kernels (SLEEF ``log2``, ``torch.round`` half-to-even,
``torch.quantile`` with linear interpolation, ``torch.topk``), and numpy cannot
reproduce them bit for bit.  There is no device code and no memory-derived
candidate chunking (``parallel_eq_n``, linear.py:111-121, affects scheduling
only: candidates are scored independently).

Pinning: ``tests/test_oracle_golden.py`` checks every function here against
fixtures captured from the reference itself running in the build container
(``tools/make_golden.py`` -> ``tests/golden/*.npz``): quantiser outputs, candidate
grids, the full score vector and top-k indices of *every* FPCS scoring call, and
the final parameters of all six layer classes at W3A3/W4A4/W6A6.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Callable, List, Optional, Tuple

import torch
import torch.nn.functional as F

R = 37.0                                   # AdaLog fixed denominator, logarithm.py:71
GELU_SHIFT = 0.16997124254703522           # -min(gelu), linear.py:749


def n_levels(bits: int) -> int:
    """uniform.py:12 -- L = 2**(bits-1); the integer grid has 2L levels."""
    return 2 ** (bits - 1)


# ====================================================================== quantisers
def uniform_fake_quant(x, scale, zero_point=None, bits=4, sym=False):
    """uniform.py:25-36 (eval form).  Returns (dequantised, integer bins as float)."""
    if bits == 32:
        return x, None
    L = n_levels(bits)
    x_int = torch.round(x / scale)
    if sym:
        q = x_int.clamp(-L, L - 1)
        return q * scale, q
    z = torch.round(zero_point)
    q = (x_int + z).clamp(0, 2 * L - 1)
    return (q - z) * scale, q


def adalog_tables(q: int, bits: int):
    """logarithm.py:77-81 -- python-float64 loop, stored as fp32."""
    L = n_levels(bits)
    t1 = torch.zeros(2 * L)
    t2 = torch.zeros(2 * L)
    for i in range(2 * L):
        val = round((2 ** (-((q * i) % R) / R)) * (4 * L - 2)) / (4 * L - 2)
        t1[i] = torch.tensor(math.floor(i * q / R))
        t2[i] = torch.tensor(val)
    return t1, t2


def adalog_bins(x, scale, q, bits):
    """logarithm.py:87,94-96 -- k = rne(-log2(clamp(x/s,1e-15,1)) * 37 / q); mask = k < 2L."""
    L = n_levels(bits)
    qf = torch.as_tensor(q)
    u = (x / scale).clamp(min=1e-15, max=1.0)
    k = torch.round(-u.log2() * R / qf)
    mask = k < 2 * L
    return torch.clamp(k, 0, 2 * L - 1), mask


def adalog_fake_quant(x, scale, q: int, bits: int, tables=None):
    """logarithm.py:83-99 (eval form).  Returns (dequantised, bins, mask)."""
    t1, t2 = tables if tables is not None else adalog_tables(int(q), bits)
    k, mask = adalog_bins(x, scale, torch.tensor([int(q)]), bits)
    y = (2 ** (-t1[k.long()])) * t2[k.long()] * scale
    return y * mask, k, mask


def adalog_fake_quant_train(x, scale, q: int, bits: int):
    """logarithm.py:88-92 -- training form: no LUT rounding of the mantissa."""
    k, mask = adalog_bins(x, scale, torch.tensor([int(q)]), bits)
    y = 2 ** (-1 * k * torch.tensor([int(q)]) / R) * scale
    return y * mask, k, mask


def shift_adalog_fake_quant(x, scale, q, bits, shift, bias_reparamed=False, tables=None):
    """logarithm.py:127-135."""
    y, k, mask = adalog_fake_quant(x + shift, scale, q, bits, tables)
    return (y if bias_reparamed else y - shift), k, mask


def adaround_init_alpha(w, scale):
    """adaround.py:62-67."""
    gamma, zeta = -0.1, 1.1
    x_floor = torch.floor(w / scale)
    rest = (w / scale) - x_floor
    return -torch.log((zeta - gamma) / (rest - gamma) - 1)


def adaround_soft_targets(alpha):
    """adaround.py:59-60."""
    return torch.clamp(torch.sigmoid(alpha) * 1.2000000000000002 + (-0.1), 0, 1)


def adaround_fake_quant(w, scale, zero_point, alpha, bits, soft):
    """adaround.py:43-57 (asymmetric branch; note: no round() on zero_point here)."""
    L = n_levels(bits)
    x_floor = torch.floor(w / scale)
    x_int = x_floor + (adaround_soft_targets(alpha) if soft else (alpha >= 0).float())
    q = torch.clamp(x_int + zero_point, 0, 2 * L - 1)
    return (q - zero_point) * scale


def adaround_hard_value(w, scale, alpha):
    """adaround.py:71-73."""
    shp = w.shape
    return ((torch.floor(w.reshape_as(alpha) / scale) + (alpha >= 0).float()) * scale).reshape(*shp)


def search_table(bits: int):
    """linear.py:750-752 / matmul.py:313-315 -- the 120-entry search-time mantissa table."""
    L = n_levels(bits)
    table = torch.tensor([2 ** (-j / R) for j in range(120)])
    ts = 1.0 / (4 * L - 2)
    return torch.round(table / ts) * ts


def adalog_search_value(u_log2_neg, qf, bits, table):
    """linear.py:831-836 / matmul.py:337-342.

    ``u_log2_neg`` = -log2(.) already taken; ``qf`` broadcastable candidate bases.
    Returns the de-quantised *unit-scale* value (masked bins are 0).
    """
    L = n_levels(bits)
    k = torch.round(u_log2_neg * R / qf)
    mask = k >= 2 * L
    k = k.clamp_(0, 2 * L - 1)
    idx = torch.remainder(k * qf, R).round_().long()
    v = (2 ** (-1 * torch.floor(k * qf / R))) * table[idx]
    v[mask] = 0
    return v


# ====================================================================== candidate grids
def _grid(delta_min, delta_max, num_scale, num_zp, L, lead_shape):
    """Shared tail of linear.py:442-451, matmul.py:231-240, conv.py:281-290."""
    ones = [1] * (delta_min.dim() - 1)
    splits = torch.linspace(0, 1, steps=num_scale).view(-1, *ones) * (delta_max - delta_min)
    scales = (delta_min + splits).repeat(num_zp, *ones) / (2 * L - 1)
    zp_min = int(L - num_zp / 2)
    zp_max = int(L + num_zp / 2)
    zps = torch.tensor(range(zp_min, zp_max)).repeat_interleave(num_scale).view(-1, *ones)
    zps = zps.repeat(1, *lead_shape)
    return scales, zps


def weight_candidates(w3, bits, eq_n=128, conv=False, lo=0.9, hi=1.0):
    """linear.py:432-451 (w3 = weight.view(n_V, rows, I)); conv.py:271-290 (w3 = weight.view(oc,-1), conv=True)."""
    L = n_levels(bits)
    num_zp = L if conv else min(16, L)
    num_scale = int(eq_n / num_zp)
    pct = torch.tensor([lo, hi])
    up = torch.quantile(w3, pct, dim=-1).unsqueeze(-1)
    dn = torch.quantile(w3, 1 - pct, dim=-1).unsqueeze(-1)
    return _grid(up[0:1] - dn[0:1], up[1:] - dn[1:], num_scale, num_zp, L, tuple(w3.shape[:-1]) + (1,))


def _chunked_quantile_mean(x2_fn, pct):
    """linear.py:465-471 / matmul.py:223-230: double the number of rows until torch.quantile accepts."""
    mbs = 1
    while True:
        try:
            v = x2_fn(mbs)
            return torch.quantile(v, pct, dim=-1).mean(dim=-1), mbs
        except RuntimeError:
            mbs *= 2


def activation_candidates(x, bits, channel_wise, eq_n=128, lo=0.9, hi=1.0):
    """linear.py:453-481.  Returns (scales [C,eq_n], zps [C,eq_n]) with C = in_features or 1."""
    L = n_levels(bits)
    num_zp = min(16, 2 * L)
    num_scale = int(eq_n / num_zp)
    pct = torch.tensor([lo, hi])
    if channel_wise:
        up = torch.quantile(x.view(-1, x.shape[-1]), pct, dim=0).transpose(0, 1)
        dn = torch.quantile(x.view(-1, x.shape[-1]), 1 - pct, dim=0).transpose(0, 1)
    else:
        up, _ = _chunked_quantile_mean(lambda m: x.reshape(m, -1), pct)
        dn, _ = _chunked_quantile_mean(lambda m: x.reshape(m, -1), 1 - pct)
        up, dn = up.unsqueeze(0), dn.unsqueeze(0)
    dmin = up[:, 0:1] - dn[:, 0:1]
    dmax = up[:, 1:] - dn[:, 1:]
    splits = torch.linspace(0, 1, steps=num_scale)[None, :] * (dmax - dmin)
    scales = ((dmin + splits).repeat(1, num_zp) / (2 * L - 1)).clamp(min=1e-4)
    zp_min, zp_max = int(L - num_zp / 2), int(L + num_zp / 2)
    zps = torch.tensor(range(zp_min, zp_max)).repeat_interleave(num_scale)[None, :].repeat(scales.shape[0], 1)
    return scales, zps


def matmul_candidates(x, bits_B, head_wise=True, eq_n=128, lo=0.9, hi=1.0):
    """matmul.py:211-240 -- both operands use B's n_levels for the grid (matmul.py:212,234)."""
    L = n_levels(bits_B)
    num_zp = min(16, L)
    num_scale = int(eq_n / num_zp)
    pct = torch.tensor([lo, hi])
    if head_wise:
        xt = x.transpose(0, 1).contiguous()
        view = lambda m: xt.view(xt.shape[0], m, -1)
    else:
        view = lambda m: x.reshape(1, m, -1)
    up, _ = _chunked_quantile_mean(view, pct)
    dn, _ = _chunked_quantile_mean(view, 1 - pct)
    dmin = (up[0] - dn[0]).view(1, 1, -1, 1, 1)
    dmax = (up[1] - dn[1]).view(1, 1, -1, 1, 1)
    return _grid(dmin, dmax, num_scale, num_zp, L, tuple(dmin.shape[1:]))


def positive_percentile(t, q):
    """linear.py:763-798 for a flat tensor (dim=0): rank ceil(count*q)-1 of the sorted positive values."""
    pos = torch.where(t > 0, t, torch.tensor(float("nan")))
    srt, _ = pos.sort(dim=0)
    counts = (~torch.isnan(srt)).sum(dim=0, keepdim=True).float()
    ranks = ((counts * q.reshape(-1, 1)).ceil().long() - 1).clamp(min=0)
    res = torch.gather(srt.unsqueeze(0).expand(q.numel(), -1), 1, ranks).squeeze(1)
    res.masked_fill_(torch.isnan(res), 0)
    return res


def postgelu_candidates(x, shift, eq_n=128, lo=0.9, hi=1.0):
    """linear.py:800-814.  Returns (ud [1,2], scales [1,eq_n])."""
    cand = positive_percentile(x.reshape(-1), torch.tensor([lo, hi])) + shift
    cand = cand.unsqueeze(0)
    scales = cand[:, 0:1] + (cand[:, 1:] - cand[:, 0:1]) * torch.tensor(
        [i / (eq_n - 1) for i in range(eq_n)]).view(1, -1)
    return cand, scales


# ====================================================================== scoring (one call = eq_n candidates)
# Candidates in flight per scoring pass.  The reference derives this from GPU memory
# (parallel_eq_n, linear.py:111-121); with its 8 GiB shim value every call scores all eq_n=128 at once, which is
# what the golden traces were captured with.  bench.py lowers it to bound host memory at full layer sizes; the
# scores are independent of it up to fp32 GEMM blocking noise.
PCHUNK = 128


def _chunks(n, step=None):
    step = PCHUNK if step is None else step
    for s in range(0, n, step):
        yield s, min(n, s + step)


def score_w_self(w3, scales, zps, bits):
    """linear.py:296-309 -- [P, n_V, rows]."""
    L = n_levels(bits)
    rw = w3.unsqueeze(0)
    out = []
    for s, e in _chunks(scales.shape[0]):
        q = ((rw / scales[s:e]).round_() + zps[s:e]).clamp(0, 2 * L - 1)
        dq = (q - zps[s:e]) * scales[s:e]
        out.append(torch.mean(-(rw - dq) ** 2, dim=-1))
    return torch.cat(out, 0)


def score_a_self(x, scales, zps, bits, channel_wise, batch=32):
    """linear.py:320-345 -- [C, P] with C = in_features (channel-wise) or 1."""
    L = n_levels(bits)
    tot = None
    for bs, be in _chunks(x.shape[0], batch):
        xb = x[bs:be].unsqueeze(-1)
        parts = []
        for s, e in _chunks(scales.shape[-1]):
            q = ((xb / scales[:, s:e]).round_() + zps[:, s:e]).clamp_(0, 2 * L - 1)
            dq = (q - zps[:, s:e]) * scales[:, s:e]
            sim = -(xb - dq) ** 2
            if sim.dim() > 3:
                sim = torch.mean(sim, dim=list(range(1, sim.dim() - 2)))
            if not channel_wise:
                sim = torch.mean(sim, dim=1, keepdim=True)
            parts.append(torch.sum(sim, dim=0, keepdim=True))
        cur = torch.cat(parts, dim=-1)
        tot = cur if tot is None else torch.cat([tot, cur], 0)
    return tot.sum(dim=0)


def score_w(x_q, w3, bias, raw_out, scales, zps, bits, batch=32):
    """linear.py:355-384 -- [P, n_V, rows].  ``x_q`` is the already fake-quantised activation."""
    L = n_levels(bits)
    n_V, rows, I = w3.shape
    acc = []
    for bs, be in _chunks(x_q.shape[0], batch):
        xb = x_q[bs:be]
        ro = raw_out[bs:be].unsqueeze(-2)
        ro = ro.view(*ro.shape[:-1], n_V, -1)
        parts = []
        for s, e in _chunks(scales.shape[0]):
            q = ((w3.unsqueeze(0) / scales[s:e]).round_() + zps[s:e]).clamp(0, 2 * L - 1)
            wd = ((q - zps[s:e]) * scales[s:e]).view(-1, I)
            b = bias.repeat(e - s) if bias is not None else None
            o = F.linear(xb, wd, b)
            o = o.view(*o.shape[:-1], e - s, n_V, -1)
            sim = -(ro - o) ** 2
            if sim.dim() > 4:
                sim = torch.mean(sim, dim=list(range(1, sim.dim() - 3)))
            parts.append(sim.sum(dim=0, keepdim=True))
        acc.append(torch.cat(parts, dim=1))
    return torch.cat(acc, 0).sum(dim=0)


def score_a(x, w_q, bias, raw_out, scales, zps, bits, batch=32):
    """linear.py:394-423 -- [1, P] per-tensor activation candidates against the output."""
    L = n_levels(bits)
    acc = []
    for bs, be in _chunks(x.shape[0], batch):
        xb = x[bs:be].unsqueeze(-1)
        ro = raw_out[bs:be].unsqueeze(-2)
        parts = []
        for s, e in _chunks(scales.shape[-1]):
            q = ((xb / scales[:, s:e]).round_() + zps[:, s:e]).clamp_(0, 2 * L - 1)
            dq = (q - zps[:, s:e]) * scales[:, s:e]
            xs = dq.permute(*range(dq.dim() - 2), -1, -2)
            o = F.linear(xs, w_q, bias)
            sim = torch.mean(-(ro - o) ** 2, dim=-1)
            if sim.dim() > 2:
                sim = torch.mean(sim, dim=list(range(1, sim.dim() - 1)))
            parts.append(torch.sum(sim, dim=0, keepdim=True))
        acc.append(torch.cat(parts, dim=1))
    return torch.cat(acc, 0).sum(dim=0, keepdim=True)


def score_postgelu(x, w_q, bias, raw_out, scales, qs, shift, bits, table, batch=32):
    """linear.py:816-848 / 856-890 / 898-931 in one: per-candidate (scale_p, q_p), [1, P]."""
    acc = []
    for bs, be in _chunks(x.shape[0], batch):
        xb = x[bs:be].unsqueeze(-1)
        ro = raw_out[bs:be].unsqueeze(-2)
        parts = []
        for s, e in _chunks(scales.shape[-1]):
            cs, cq = scales[:, s:e], qs[:, s:e]
            u = ((xb + shift) / cs).clamp(min=1e-15, max=1.0)
            v = adalog_search_value(-u.log2(), cq, bits, table)
            xs = (v * cs - shift).permute(*range(v.dim() - 2), -1, -2)
            o = F.linear(xs, w_q, bias)
            sim = torch.mean(-(ro - o) ** 2, dim=-1)
            if sim.dim() > 2:
                sim = torch.mean(sim, dim=list(range(1, sim.dim() - 1)))
            parts.append(torch.sum(sim, dim=0, keepdim=True))
        acc.append(torch.cat(parts, dim=1))
    return torch.cat(acc, 0).sum(dim=0, keepdim=True)


def score_matmul(A, B, raw_out, scales, zps, bits, which, fixed_q, head_wise=True, batch=32):
    """matmul.py:135-163 (which='A') / 173-201 (which='B') -- [P, H] (or [P]).

    ``fixed_q`` is the other operand already fake-quantised with the current parameters.
    """
    L = n_levels(bits)
    acc = []
    for bs, be in _chunks(A.shape[0], batch):
        ro = raw_out[bs:be].unsqueeze(0)
        parts = []
        for s, e in _chunks(scales.shape[0]):
            src = (A if which == "A" else B)[bs:be]
            q = ((src / scales[s:e]).round_() + zps[s:e]).clamp(0, 2 * L - 1)
            sim_op = (q - zps[s:e]).mul_(scales[s:e])
            if which == "A":
                o = sim_op @ fixed_q[bs:be].unsqueeze(0)
            else:
                o = fixed_q[bs:be].unsqueeze(0) @ sim_op
            sim = -(ro - o) ** 2
            sim = torch.mean(sim, dim=list(range(3 if head_wise else 2, sim.dim())))
            parts.append(sim.sum(dim=1, keepdim=True))
        acc.append(torch.cat(parts, 0))
    return torch.cat(acc, dim=1).sum(dim=1)


def score_log_base_A(A, B_q, raw_out, qs, bits, table, batch=32):
    """matmul.py:321-351 -- per-tensor score [P,1] of post-softmax log bases (no clamp before log2)."""
    acc = []
    for bs, be in _chunks(A.shape[0], batch):
        ro = raw_out[bs:be].unsqueeze(0)
        Bq = B_q[bs:be].unsqueeze(0)
        nl = -A[bs:be].log2()
        parts = []
        for s, e in _chunks(qs.shape[0]):
            v = adalog_search_value(nl, qs[s:e], bits, table)
            o = v @ Bq
            sim = -(ro - o) ** 2
            sim = torch.mean(sim, dim=list(range(2, sim.dim())))
            parts.append(sim.sum(dim=1, keepdim=True))
        acc.append(torch.cat(parts, 0))
    return torch.cat(acc, dim=1).sum(dim=1, keepdim=True)


def score_conv_w(x, w2, bias, raw_out, scales, zps, bits, stride, ksize, batch=32):
    """conv.py:226-255 -- [P, oc]."""
    L = n_levels(bits)
    oc = w2.shape[0]
    ic = x.shape[1]
    acc = []
    for bs, be in _chunks(x.shape[0], batch):
        xb = x[bs:be]
        ro = raw_out[bs:be].unsqueeze(1)
        parts = []
        for s, e in _chunks(scales.shape[0]):
            q = ((w2.unsqueeze(0) / scales[s:e]).round_() + zps[s:e]).clamp(0, 2 * L - 1)
            wd = (q - zps[s:e]).mul_(scales[s:e]).view(-1, ic, *ksize)
            b = bias.repeat(e - s) if bias is not None else None
            o = F.conv2d(xb, wd, b, stride)
            o = torch.cat(torch.chunk(o.unsqueeze(1), chunks=e - s, dim=2), dim=1)
            sim = torch.mean(-(ro - o) ** 2, [3, 4])
            parts.append(torch.sum(sim, dim=0, keepdim=True))
        acc.append(torch.cat(parts, dim=1))
    return torch.cat(acc, 0).sum(dim=0)


# ====================================================================== FPCS driver
@dataclass
class Trace:
    """Every scoring call of a search, for step-by-step pinning against the golden traces."""
    scores: List[torch.Tensor] = field(default_factory=list)
    ks: List[int] = field(default_factory=list)
    idx: List[torch.Tensor] = field(default_factory=list)

    def add(self, s, k, i):
        self.scores.append(s.clone())
        self.ks.append(k)
        self.idx.append(i.clone())


def fpcs(scales, zps, score_fn: Callable, cand_dim: int, steps=6, width=16, eq_n=128,
         clamp_min: Optional[float] = None, trace: Optional[Trace] = None, topk_shape=None):
    """linear.py:483-523, matmul.py:243-262, conv.py:292-311.

    ``cand_dim`` is 0 (weights / matmul / conv) or -1 (activations).  ``score_fn(scales, zps)``
    returns scores with the candidate axis at ``cand_dim`` (possibly with fewer dims than the
    candidates); ``topk_shape(k)`` reshapes top-k indices to the candidates' rank.
    Returns the committed (scale, zero_point) = the top-1 of the last step.
    """
    new_cnt = int(eq_n / width)
    sl = lambda t, a, b: t[a:b] if cand_dim == 0 else t[..., a:b]
    delta = sl(scales, 1, 2) - sl(scales, 0, 1)

    def select(sc, zp, k):
        s = score_fn(sc, zp)
        _, idx = torch.topk(s, k=k, dim=cand_dim)
        if trace is not None:
            trace.add(s, k, idx)
        idx = topk_shape(idx, k) if topk_shape is not None else idx
        return torch.gather(sc, dim=cand_dim, index=idx), torch.gather(zp, dim=cand_dim, index=idx)

    top_s, top_z = select(scales, zps, width)
    remain = steps - 1
    while remain > 0:
        lin = torch.linspace(0, 1, steps=new_cnt)
        if cand_dim == 0:
            d = (lin.view(-1, *[1] * (scales.dim() - 1)) - 0.5) * delta
            delta = delta / (new_cnt - 0.5)
            scales = (top_s.unsqueeze(1) + d.unsqueeze(0)).reshape(-1, *scales.shape[1:])
            zps = top_z.repeat_interleave(new_cnt, dim=0)
        else:
            d = (lin[None, :] - 0.5) * delta
            delta = delta / (new_cnt - 0.5)
            scales = (top_s.unsqueeze(-1) + d.unsqueeze(-2)).reshape(*scales.shape[:-1], -1)
            zps = top_z.repeat_interleave(new_cnt, dim=-1)
        if clamp_min is not None:
            scales = scales.clamp(min=clamp_min)
        top_s, top_z = select(scales, zps, 1 if remain == 1 else width)
        remain -= 1
    return top_s, top_z


# ====================================================================== layer searches
def _obs(observer, kind, p, a, b, s):
    """Report one scoring call to a test observer: (kind, parameters in force, the two candidate tensors, scores).
    Called right after every score function, i.e. once per Trace entry and in the same order."""
    if observer is not None:
        observer(kind, p, a, b, s)
    return s


@dataclass
class LinearParams:
    w_scale: torch.Tensor = None      # [n_V, rows, 1]
    w_zp: torch.Tensor = None
    a_scale: torch.Tensor = None      # [1] or [I]
    a_zp: torch.Tensor = None
    a_q: int = 37                     # post-GELU only
    weight: torch.Tensor = None
    bias: torch.Tensor = None


def _w_fq(w3, p: LinearParams, bits):
    return uniform_fake_quant(w3, p.w_scale, p.w_zp, bits)[0]


def search_linear(weight, bias, x, raw_out, w_bit, a_bit, n_V=1, rounds=3, steps=6, eq_n=128, batch=32,
                  trace: Optional[Trace] = None, a_init=None, observer=None) -> LinearParams:
    """AsymmetricallyBatchingQuantLinear.hyperparameter_searching, linear.py:525-545 (fpcs=True)."""
    O, I = weight.shape
    w3 = weight.view(n_V, O // n_V, I)
    p = LinearParams(weight=weight, bias=bias)
    shape_w = lambda idx, k: idx.reshape(k, n_V, -1, 1)

    def w_search(score):
        sc, zp = weight_candidates(w3, w_bit, eq_n)
        s, z = fpcs(sc, zp, score, 0, steps, 16, eq_n, None, trace, shape_w)
        p.w_scale, p.w_zp = s.squeeze(0), z.squeeze(0).float()

    def a_search(score):
        sc, zp = activation_candidates(x, a_bit, False, eq_n)
        s, z = fpcs(sc, zp, score, -1, steps, 16, eq_n, 1e-4, trace)
        p.a_scale, p.a_zp = s.squeeze(-1), z.squeeze(-1).float()

    w_search(lambda sc, zp: _obs(observer, "w_self", p, sc, zp, score_w_self(w3, sc, zp, w_bit)))
    a_search(lambda sc, zp: _obs(observer, "a_self", p, sc, zp, score_a_self(x, sc, zp, a_bit, False, batch)))
    for _ in range(rounds):
        xq = uniform_fake_quant(x, p.a_scale, p.a_zp, a_bit)[0]
        w_search(lambda sc, zp: _obs(observer, "w_out", p, sc, zp, score_w(xq, w3, bias, raw_out, sc, zp, w_bit, batch)))
        wq = _w_fq(w3, p, w_bit).view(O, I)
        a_search(lambda sc, zp: _obs(observer, "a_out", p, sc, zp, score_a(x, wq, bias, raw_out, sc, zp, a_bit, batch)))
    return p


def search_linear_channelwise(x, a_bit, steps=6, eq_n=128, batch=32, trace=None, observer=None):
    """AsymmetricallyChannelWiseBatchingQuantLinear.hyperparameter_searching, linear.py:585-594."""
    sc, zp = activation_candidates(x, a_bit, True, eq_n)
    s, z = fpcs(sc, zp, lambda a, b: _obs(observer, "a_self_cw", None, a, b, score_a_self(x, a, b, a_bit, True, batch)),
                -1, steps, 16, eq_n, 1e-4, trace)
    return s.squeeze(-1), z.squeeze(-1).float()


def reparam_step1(a_scale, a_zp, ln_weight, ln_bias, weight, bias):
    """linear.py:596-612.  Returns (r, b, target_scale, target_zp, ln_w', ln_b', W', bias')."""
    channel_min = -a_zp * a_scale
    t_scale = torch.mean(a_scale).view(1)
    t_zp = torch.mean(a_zp).round().view(1)
    t_min = -t_zp * t_scale
    r = a_scale / t_scale
    b = channel_min / r - t_min
    ln_w = ln_weight / r
    ln_b = ln_bias / r.view(-1) - b
    W = weight * r.view(1, -1)
    add = torch.mm(W, b.reshape(-1, 1)).reshape(-1)
    new_bias = bias + add if bias is not None else add
    return r, b, t_scale, t_zp, ln_w, ln_b, W, new_bias


def search_postgelu(weight, bias, x, raw_out, w_bit, a_bit, rounds=3, steps=6, eq_n=128, batch=32,
                    trace: Optional[Trace] = None, observer=None) -> LinearParams:
    """PostGeluLogBasedBatchingQuantLinear.hyperparameter_searching, linear.py:969-997 (fpcs=True)."""
    O, I = weight.shape
    w3 = weight.view(1, O, I)
    p = LinearParams(weight=weight, bias=bias)
    shift = torch.tensor(GELU_SHIFT)
    table = search_table(a_bit)
    shape_w = lambda idx, k: idx.reshape(k, 1, -1, 1)

    def w_search(score):
        sc, zp = weight_candidates(w3, w_bit, eq_n)
        s, z = fpcs(sc, zp, score, 0, steps, 16, eq_n, None, trace, shape_w)
        p.w_scale, p.w_zp = s.squeeze(0), z.squeeze(0).float()

    w_search(lambda sc, zp: _obs(observer, "w_self", p, sc, zp, score_w_self(w3, sc, zp, w_bit)))
    ud, sc_all = postgelu_candidates(x, shift.item(), eq_n)
    p.a_scale = sc_all[:, -2].clone()
    p.a_q = 37
    for _ in range(rounds):
        wq = _w_fq(w3, p, w_bit).view(O, I)
        # activation_fpcs, linear.py:941-967
        q_all = torch.tensor([i for i in range(10, 11 + eq_n)]).view(1, -1)
        s0 = score_postgelu(x, wq, bias, raw_out, p.a_scale.view(1, 1).expand(1, eq_n), q_all[:, :eq_n],
                            shift, a_bit, table, batch)
        _obs(observer, "a_logbase", p, p.a_scale.view(1, 1).expand(1, eq_n), q_all[:, :eq_n], s0)
        _, qi = torch.topk(s0, k=8, dim=-1)
        if trace is not None:
            trace.add(s0, 8, qi)
        scs = ud[:, 0:1] + (ud[:, 1:] - ud[:, 0:1]) * torch.tensor([i / 15 for i in range(16)]).view(1, -1)
        delta = scs[:, 1:2] - scs[:, 0:1]
        scs = scs.repeat(1, 8)
        qs = torch.gather(q_all, dim=-1, index=qi).repeat_interleave(16, dim=-1)
        width, new_cnt = 32, int(eq_n / 32)

        def sel(scs, qs, k):
            s = _obs(observer, "a_logbase", p, scs, qs,
                     score_postgelu(x, wq, bias, raw_out, scs, qs, shift, a_bit, table, batch))
            _, idx = torch.topk(s, k=k, dim=-1)
            if trace is not None:
                trace.add(s, k, idx)
            return torch.gather(scs, -1, idx), torch.gather(qs, -1, idx)

        ts, tq = sel(scs, qs, width)
        remain = steps - 1
        while remain > 0:
            d = (torch.linspace(0, 1, steps=new_cnt)[None, :] - 0.5) * delta
            delta = delta / (new_cnt - 0.5)
            scs = (ts.unsqueeze(-1) + d.unsqueeze(-2)).reshape(1, -1)
            qs = tq.repeat_interleave(new_cnt, dim=-1)
            ts, tq = sel(scs, qs, 1 if remain == 1 else width)
            remain -= 1
        p.a_scale, p.a_q = ts.squeeze(-1), int(tq.item())
        t1, t2 = adalog_tables(p.a_q, a_bit)
        xq = shift_adalog_fake_quant(x, p.a_scale, p.a_q, a_bit, shift, False, (t1, t2))[0]
        w_search(lambda sc, zp: _obs(observer, "w_out", p, sc, zp, score_w(xq, w3, bias, raw_out, sc, zp, w_bit, batch)))
    return p


def reparam_bias(weight_q, bias, shift=GELU_SHIFT):
    """linear.py:999-1006."""
    x_ = torch.full((1, weight_q.shape[1]), -shift)
    return bias + (x_ @ weight_q.transpose(0, 1)).squeeze()


@dataclass
class MatMulParams:
    A_scale: torch.Tensor = None      # [1,H,1,1]
    A_zp: torch.Tensor = None
    B_scale: torch.Tensor = None
    B_zp: torch.Tensor = None
    A_q: int = 37


def search_matmul(A, B, raw_out, A_bit, B_bit, rounds=3, steps=6, eq_n=128, batch=32,
                  trace: Optional[Trace] = None, observer=None) -> MatMulParams:
    """AsymmetricallyBatchingQuantMatMul.hyperparameter_searching, matmul.py:264-283 (head-wise, fpcs)."""
    H = A.shape[1]
    p = MatMulParams()
    shp = lambda idx, k: idx.view(k, 1, -1, 1, 1)
    sA, zA = matmul_candidates(A, B_bit, True, eq_n)
    sB, zB = matmul_candidates(B, B_bit, True, eq_n)
    p.A_scale, p.A_zp = sA[-2].clone(), zA[-2].float()
    p.B_scale, p.B_zp = sB[-2].clone(), zB[-2].float()
    for _ in range(rounds):
        Bq = uniform_fake_quant(B, p.B_scale, p.B_zp, B_bit)[0]
        s, z = fpcs(*matmul_candidates(A, B_bit, True, eq_n),
                    lambda sc, zp: _obs(observer, "A", p, sc, zp,
                                        score_matmul(A, B, raw_out, sc, zp, A_bit, "A", Bq, True, batch)),
                    0, steps, 16, eq_n, None, trace, shp)
        p.A_scale, p.A_zp = s.view(1, H, 1, 1), z.view(1, H, 1, 1).float()
        Aq = uniform_fake_quant(A, p.A_scale, p.A_zp, A_bit)[0]
        s, z = fpcs(*matmul_candidates(B, B_bit, True, eq_n),
                    lambda sc, zp: _obs(observer, "B", p, sc, zp,
                                        score_matmul(A, B, raw_out, sc, zp, B_bit, "B", Aq, True, batch)),
                    0, steps, 16, eq_n, None, trace, shp)
        p.B_scale, p.B_zp = s.view(1, H, 1, 1), z.view(1, H, 1, 1).float()
    return p


def search_postsoftmax(A, B, raw_out, A_bit, B_bit, rounds=3, steps=6, eq_n=128, batch=32,
                       trace: Optional[Trace] = None, observer=None) -> MatMulParams:
    """PostSoftmaxAsymmetricallyBatchingQuantMatMul.hyperparameter_searching, matmul.py:360-378."""
    H = A.shape[1]
    p = MatMulParams(A_scale=torch.ones(1, 1, 1, 1))
    table = search_table(A_bit)
    shp = lambda idx, k: idx.view(k, 1, -1, 1, 1)
    sB, zB = matmul_candidates(B, B_bit, True, eq_n)
    p.B_scale, p.B_zp = sB[-2].clone(), zB[-2].float()
    for _ in range(rounds):
        Bq = uniform_fake_quant(B, p.B_scale, p.B_zp, B_bit)[0]
        qs = torch.tensor([i for i in range(10, 11 + eq_n)]).view(-1, 1, 1, 1, 1)
        s0 = _obs(observer, "A_logbase", p, None, qs[:eq_n], score_log_base_A(A, Bq, raw_out, qs[:eq_n], A_bit, table, batch))
        _, qi = torch.topk(s0, k=1, dim=0)
        if trace is not None:
            trace.add(s0, 1, qi)
        p.A_q = int(qs[qi.item()].item())
        Aq = adalog_fake_quant(A, p.A_scale, p.A_q, A_bit)[0]
        s, z = fpcs(*matmul_candidates(B, B_bit, True, eq_n),
                    lambda sc, zp: _obs(observer, "B", p, sc, zp,
                                        score_matmul(A, B, raw_out, sc, zp, B_bit, "B", Aq, True, batch)),
                    0, steps, 16, eq_n, None, trace, shp)
        p.B_scale, p.B_zp = s.view(1, H, 1, 1), z.view(1, H, 1, 1).float()
    return p


def search_conv(weight, bias, x, raw_out, w_bit, stride, steps=6, eq_n=128, batch=32, trace=None, observer=None):
    """AsymmetricallyBatchingQuantConv2d.hyperparameter_searching, conv.py:313-334.

    With ``qconv_a_bit = 8`` (configs/*.py:12) the input is left in fp32 (conv.py:55-58) and
    the round loop breaks after the first weight FPCS (conv.py:328-331).
    """
    oc = weight.shape[0]
    w2 = weight.view(oc, -1)
    sc, zp = weight_candidates(w2, w_bit, eq_n, conv=True)
    shp = lambda idx, k: idx.view(k, -1, 1)
    s, z = fpcs(sc, zp,
                lambda a, b: _obs(observer, "w_out", None, a, b,
                                  score_conv_w(x, w2, bias, raw_out, a, b, w_bit, stride, tuple(weight.shape[2:]), batch)),
                0, steps, 16, eq_n, None, trace, shp)
    return s.squeeze(0), z.squeeze(0).float()


# ====================================================================== quantised forward passes (a7)
def linear_quant_forward(x, p: LinearParams, w_bit, a_bit, n_V=1):
    """linear.py:46-51,90-92."""
    O, I = p.weight.shape
    wq = _w_fq(p.weight.view(n_V, O // n_V, I), p, w_bit).view(O, I)
    xq = uniform_fake_quant(x, p.a_scale, p.a_zp, a_bit)[0]
    return F.linear(xq, wq, p.bias)


def postgelu_quant_forward(x, p: LinearParams, w_bit, a_bit, bias_reparamed=False, bias=None):
    O, I = p.weight.shape
    wq = _w_fq(p.weight.view(1, O, I), p, w_bit).view(O, I)
    xq = shift_adalog_fake_quant(x, p.a_scale, p.a_q, a_bit, torch.tensor(GELU_SHIFT), bias_reparamed)[0]
    return F.linear(xq, wq, p.bias if bias is None else bias)


def matmul_quant_forward(A, B, p: MatMulParams, A_bit, B_bit, post_softmax=False):
    """matmul.py:43-45."""
    Bq = uniform_fake_quant(B, p.B_scale, p.B_zp, B_bit)[0]
    if post_softmax:
        Aq = adalog_fake_quant(A, p.A_scale, p.A_q, A_bit)[0]
    else:
        Aq = uniform_fake_quant(A, p.A_scale, p.A_zp, A_bit)[0]
    return Aq @ Bq


def conv_quant_forward(x, weight, bias, w_scale, w_zp, w_bit, stride):
    """conv.py:60-65,115-120 with an unquantised (>=8 bit) input."""
    oc = weight.shape[0]
    wq = uniform_fake_quant(weight.view(oc, -1), w_scale, w_zp, w_bit)[0].view_as(weight)
    return F.conv2d(x, wq, bias, stride)
