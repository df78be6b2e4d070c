#!/usr/bin/env python3
"""Generator of the hand-scheduled CDNA4 (gfx950) main loop of the fused activation-search kernel (gemm_fused.hip).

    FUSED_NRB=12 FUSED_FNS=4 python tools/gen_fused_asm.py   -> adalog_amd/csrc/fused_loop_nrb12_s4.inc  (+ .lst, a listing)
    tools/gen_fused_asm_all.sh                               -> the variants the library ships

Why a generator: the loop needs every register placed by hand and the issue order of a K-step laid out explicitly (the
compiler reads each A fragment right in front of its MFMA and cannot be steered into interleaving the fragment
generation with the matrix stream).

Machine model the layout follows (tools/lab/overlap_probe: measured on MI355X):
  * v_mfma_f32_32x32x16_bf16 keeps the SIMD's vector issue port for 16 of its 32 cycles, for EVERY wave of the SIMD: per
    MFMA only ~4 VALU instructions (4 cycles each) are free, a second wave does not add VALU throughput;
  * an LDS instruction costs the issuing wave ~6 cycles; with two waves on the SIMD about half of that overlaps;
  * a single wave per SIMD exposes every LDS round trip, the barrier skew and the near-tie blocks of any one wave to
    the whole workgroup (round-2 first form: 4 waves x 24 tiles, ~900 of 3100 cycles per K-step were such stalls).
Hence EIGHT waves per workgroup, two per SIMD, 256 registers each: wave w owns token (w >> 2) of the tile's pair and
candidate block (w & 3): NRB accumulator tiles of 32 x 32 (eight in a0..a127, four in v64..v127: the compiler splits a
256-register wave 128 + 128) and 64 VGPRs for everything else.  Per K-step and wave: 2 x NRB MFMAs, the 16 element-candidates of its B fragments per K half
(fma, med3, add, shift-add, LUT read, three more VALU for the near-tie test; two values pack into a dword), 2 x NRB
A-fragment reads three row blocks ahead,
NRB / 4 + 1 LDS-DMA requests, counted s_waitcnt from a model of the in-order LDS queue, one barrier.

Near-ties: each generated element adds |kf - rne(kf)| to a running maximum (3 more VALU per element); one compare per
unit branches to a cold block that re-derives the chunk's bins from the threshold table.  (A pre-pass that marked the
near-tie chunks per token and K half, leaving the loop without detection, was measured: loop 0.82 ms + pre-pass 0.17 ms
against 0.89 ms with the detection inside -- the loop is bound by latency, not by its VALU count.)

The text is one inline-asm block (v0..v127, a0..a127 and s8..s99 clobbered); the HIP kernel around it builds the LDS
tables, hands the scalars over through an LDS config array and turns the per-lane sums into the workgroup's output row.
See gemm_fused.hip for the algorithm; names here follow it.
"""
import os
import re

NRB = int(os.environ.get("FUSED_NRB", "12"))      # row blocks of 32 output channels per tile (12, 8 or 4)
FNS = int(os.environ.get("FUSED_FNS", "4"))       # weight-ring stages (3 where the 6-bit tables need the LDS)
NW = 8            # waves per workgroup
XS = FNS + 1      # x / log2 ring slots
XSLOT = 2048      # bytes reserved per x-ring slot (512 used: [token][x 128 B | log2 128 B])
AT = NRB * 2048   # weight tile bytes per stage
ROWS = 32 * NRB
RQ = NRB // 4     # weight DMA requests (16 rows x 64 B) per wave and K-step
assert NRB % 4 == 0 and NRB <= 12
MAGIC = 0x4B400000

# ---------------------------------------------------------------- config array in LDS (dword indices), written by the kernel
CFG = ["pW_lo", "pW_hi", "pX_lo", "pX_hi", "pL_lo", "pL_hi", "pRef_lo", "pRef_hi", "pRs_lo", "pRs_hi", "pRb_lo", "pRb_hi",
       "M", "T", "K", "Kb", "nk", "n_rt", "ntile", "nwg", "bid", "L2", "shift", "w",
       "oRing", "oXr", "oLut", "oThr", "oPar", "oRefb", "oRs", "oFin", "dpair", "drt", "pair0", "rt0", "tie"]
S = {n: 40 + i for i, n in enumerate(CFG)}          # s40 .. s75 (+ overrides below)
S.update(tie=19, cblk=37)
S.update(rW=76, rX=80, xw=84, aw=85,                 # buffer resources (4 SGPRs each); per-wave ring offsets
         a_tile=88, a_pair=89, a_rt=90, a_k=91, l_tile=92, l_pair=93, l_rt=94, l_k=95,
         c_tile=96, kt=97, stA=98, stX=99, c_pair=30, c_rt=31)
T0 = 8                                               # s8 .. s29: temporaries (s30, s31: compute cursor)


def s(name):
    return f"s{S[name]}"


def t(i):
    return f"s{T0 + i}"


def t2(i):
    return f"s[{T0 + i}:{T0 + i + 1}]"


# ---------------------------------------------------------------- VGPR map (64 registers)
V = dict(LANE=0, FROW=1, FKG=2, PAR=4, LUTC=8, THRC=9, LUTB=10, AOFF0=11, AOFF1=12, AS0=13, AS1=14, XOFF=15, XSC=16, XSN=17,
         XDMA=18, DMA=19, DX=22, DM=23, RUN=24, BA=26, BB=30, LV=34, ABUF=42, VAL=58, GT=62,
         TV=42)      # TV: 16 scratch registers shared with the A buffers (dead outside the MFMA stream)
CA, CC, CHI, AL = 0, 1, 2, 3


def v(name, off=0):
    return f"v{V[name] + off}"


def vr(name, off, n):
    a = V[name] + off
    return f"v[{a}:{a + n - 1}]"


NA = min(NRB, 8)   # accumulator tiles in AGPRs (a0..a127); the compiler splits a 256-register wave 128 + 128, so tiles
                   # 8..11 live in v64..v127


def acc(rb):
    if rb < NA:
        return f"a[{16 * rb}:{16 * rb + 15}]"
    b = 64 + 16 * (rb - NA)
    return f"v[{b}:{b + 15}]"


class Asm:
    def __init__(self):
        self.lines = []
        self.fifo = []          # outstanding LDS operations, oldest first (tags)
        self.uid = 0

    def e(self, text):
        self.lines.append(text)

    def c(self, text):
        self.lines.append("; " + text)

    def ds(self, text, tag):
        """an LDS operation whose completion someone will wait for"""
        self.e(text)
        self.fifo.append(tag)

    def wait(self, tag):
        """s_waitcnt lgkmcnt so that `tag` (and everything older) has returned"""
        if tag not in self.fifo:
            return
        i = self.fifo.index(tag)
        n = len(self.fifo) - 1 - i
        assert n <= 15, f"too many LDS operations in flight behind {tag}: {n}"
        self.e(f"s_waitcnt lgkmcnt({n})")
        self.fifo = self.fifo[i + 1:]

    def drain(self):
        if self.fifo:
            self.e("s_waitcnt lgkmcnt(0)")
            self.fifo = []

    def label(self, name):
        self.e(f"{name}_%=:")

    def new(self, base):
        self.uid += 1
        return f"{base}{self.uid}"


def cfg_load(A):
    A.c("scalars: LDS config array -> SGPRs (one dword per lane, then readlane)")
    A.e("v_mbcnt_lo_u32_b32 v0, -1, 0")
    A.e("v_mbcnt_hi_u32_b32 v0, -1, v0")
    A.e("v_lshlrev_b32 v3, 2, v0")
    A.e("v_add_u32 v3, %[cfg], v3")
    A.e("ds_read_b32 v4, v3")
    A.e("s_waitcnt lgkmcnt(0)")
    for i, n in enumerate(CFG):
        if n != "w":
            A.e(f"v_readlane_b32 {s(n)}, v4, {i}")
    A.e("s_nop 4")
    A.e(f"s_mov_b32 {s('w')}, s39")                               # wave index: handed over in s39 by the wrapper


def lane_setup(A):
    A.c("lane geometry: wave w = (token w >> 2 of the pair, candidate block w & 3); lane = (frow = candidate, fkg = k group)")
    A.e(f"v_and_b32 {v('FROW')}, 31, {v('LANE')}")
    A.e(f"v_lshrrev_b32 {v('FKG')}, 5, {v('LANE')}")
    A.e(f"s_lshr_b32 {t(0)}, {s('w')}, 2")                       # wtok
    A.e(f"s_and_b32 {s('cblk')}, {s('w')}, 3")
    A.e(f"s_lshl_b32 {t(2)}, {s('cblk')}, 5")
    A.e(f"v_add_u32 v3, {t(2)}, {v('FROW')}")                    # candidate index c
    # parameters {-37/q, log2(s)*37/q, hi, s*sa_mul}: float4 at oPar + c*16
    A.e(f"v_lshl_add_u32 {v('TV')}, v3, 4, {s('oPar')}")
    A.ds(f"ds_read_b128 {vr('PAR', 0, 4)}, {v('TV')}", "p0")
    # LUT / threshold lane bases
    A.e(f"v_lshl_add_u32 {v('LUTB')}, v3, 2, {s('oLut')}")
    A.e(f"s_mov_b32 {t(3)}, 0x{(-(MAGIC << 9)) & 0xFFFFFFFF:08x}")
    A.e(f"v_add_u32 {v('LUTC')}, {t(3)}, {v('LUTB')}")
    A.e(f"v_lshl_add_u32 {v('THRC')}, v3, 2, {s('oThr')}")
    # A fragment lane offsets: frow*64 + (((2h + fkg) ^ ((frow >> 2) & 3)) << 4)
    A.e(f"v_lshrrev_b32 {v('TV', 1)}, 2, {v('FROW')}")
    A.e(f"v_and_b32 {v('TV', 1)}, 3, {v('TV', 1)}")               # sw
    A.e(f"v_xor_b32 {v('TV', 2)}, {v('FKG')}, {v('TV', 1)}")      # h = 0: fkg ^ sw
    A.e(f"v_lshlrev_b32 {v('TV', 3)}, 6, {v('FROW')}")
    A.e(f"v_lshl_add_u32 {v('AOFF0')}, {v('TV', 2)}, 4, {v('TV', 3)}")
    A.e(f"v_xor_b32 {v('TV', 2)}, 2, {v('TV', 2)}")               # h = 1: (2 + fkg) ^ sw = (fkg ^ sw) ^ 2
    A.e(f"v_lshl_add_u32 {v('AOFF1')}, {v('TV', 2)}, 4, {v('TV', 3)}")
    # ring offsets of this wave: x slot part (token, array, half) and weight requests
    A.e(f"s_lshl_b32 {s('xw')}, {t(0)}, 8")                       # wtok * 256
    A.e(f"s_lshl_b32 {t(4)}, {s('cblk')}, 6")                     # (array, half) = cblk: 64 bytes each
    A.e(f"s_add_i32 {t(4)}, {t(4)}, {s('xw')}")
    A.e(f"s_mul_i32 {s('aw')}, {s('w')}, {RQ * 1024}")
    # x / log2 read offset inside a slot: token part + fkg * 32 bytes (log2 run at +128)
    A.e(f"s_add_i32 {t(5)}, {s('xw')}, {s('oXr')}")
    A.e(f"v_lshl_add_u32 {v('XOFF')}, {v('FKG')}, 5, {t(5)}")
    A.e(f"s_mov_b32 {s('xw')}, {t(4)}")                           # xw = DMA destination part of this wave inside a slot
    # x / log2 DMA voffset: (wtok * K + half * 16 + lane) * 4   (lanes 0..15 only)
    A.e(f"s_mul_i32 {t(5)}, {t(0)}, {s('K')}")
    A.e(f"s_and_b32 {t(6)}, {s('cblk')}, 1")
    A.e(f"s_lshl_b32 {t(6)}, {t(6)}, 4")
    A.e(f"s_add_i32 {t(5)}, {t(5)}, {t(6)}")
    A.e(f"v_add_u32 {v('TV', 1)}, {t(5)}, {v('LANE')}")
    A.e(f"v_lshlrev_b32 {v('XDMA')}, 2, {v('TV', 1)}")
    A.e(f"v_mov_b32 {v('RUN')}, 0")
    A.e(f"v_mov_b32 {v('RUN', 1)}, 0")
    A.drain()
    # buffer resources: W {ptr, 0 stride, M*Kb records, 0x00020000}
    A.e(f"s_mov_b32 s{S['rW']}, {s('pW_lo')}")
    A.e(f"s_and_b32 s{S['rW'] + 1}, {s('pW_hi')}, 0xffff")
    A.e(f"s_mul_i32 s{S['rW'] + 2}, {s('M')}, {s('Kb')}")
    A.e(f"s_mov_b32 s{S['rW'] + 3}, 0x00020000")
    A.e(f"s_mov_b32 s{S['rX'] + 3}, 0x00020000")


def set_x_rsrc(A):
    """resource of this wave's run (x or log2: bit 1 of cblk) for the L cursor's token pair: base + tok0*K*4,
    records = min(2, T - tok0)*K*4"""
    A.e(f"s_lshl_b32 {t(0)}, {s('l_pair')}, 1")                    # tok0
    A.e(f"s_mul_i32 {t(2)}, {t(0)}, {s('K')}")
    A.e(f"s_mul_hi_u32 {t(3)}, {t(0)}, {s('K')}")
    A.e(f"s_lshl_b64 {t2(2)}, {t2(2)}, 2")                         # byte offset (64 bit)
    A.e(f"s_bitcmp1_b32 {s('cblk')}, 1")
    A.e(f"s_cselect_b32 {t(4)}, {s('pL_lo')}, {s('pX_lo')}")
    A.e(f"s_cselect_b32 {t(5)}, {s('pL_hi')}, {s('pX_hi')}")
    A.e(f"s_add_u32 s{S['rX']}, {t(4)}, {t(2)}")
    A.e(f"s_addc_u32 {t(4)}, {t(5)}, {t(3)}")
    A.e(f"s_and_b32 s{S['rX'] + 1}, {t(4)}, 0xffff")
    A.e(f"s_sub_i32 {t(4)}, {s('T')}, {t(0)}")
    A.e(f"s_min_i32 {t(4)}, {t(4)}, 2")
    A.e(f"s_mul_i32 {t(4)}, {t(4)}, {s('K')}")
    A.e(f"s_lshl_b32 s{S['rX'] + 2}, {t(4)}, 2")


def set_a_rows(A):
    """weight DMA voffsets of the A cursor's row tile: min(rt*ROWS + (w*RQ + q)*16 + lrow, M-1) * Kb + lslot16"""
    a, b, c = v('VAL', 0), v('VAL', 1), v('VAL', 2)              # (runs at a step's top: the A buffers already receive fragments)
    A.e(f"v_lshrrev_b32 {a}, 2, {v('LANE')}")                      # lrow
    A.e(f"v_and_b32 {b}, 3, {v('LANE')}")
    A.e(f"v_lshrrev_b32 {c}, 4, {v('LANE')}")
    A.e(f"v_and_b32 {c}, 3, {c}")
    A.e(f"v_xor_b32 {b}, {b}, {c}")
    A.e(f"v_lshlrev_b32 {b}, 4, {b}")                              # lslot16
    A.e(f"s_mul_i32 {t(0)}, {s('a_rt')}, {ROWS}")
    A.e(f"s_mul_i32 {t(1)}, {s('w')}, {RQ * 16}")
    A.e(f"s_add_i32 {t(0)}, {t(0)}, {t(1)}")
    A.e(f"v_add_u32 {a}, {t(0)}, {a}")                             # row of q = 0
    A.e(f"s_sub_i32 {t(2)}, {s('M')}, 1")
    for q in range(RQ):
        A.e(f"v_add_u32 {c}, {16 * q}, {a}")
        A.e(f"v_min_i32 {c}, {t(2)}, {c}")
        A.e(f"v_mul_lo_u32 {c}, {c}, {s('Kb')}")
        A.e(f"v_add_u32 v{V['DMA'] + q}, {c}, {b}")


def cursor_step(A, pre, on_wrap):
    """advance cursor `pre` (a / l) by one K-step; `on_wrap(A)` emits the tile-change work (taken once per tile)"""
    lab = A.new("Lcur")
    A.e(f"s_add_i32 {s(pre + '_k')}, {s(pre + '_k')}, 1")
    A.e(f"s_cmp_lg_u32 {s(pre + '_k')}, {s('nk')}")
    A.e(f"s_cbranch_scc1 {lab}_%=")
    A.e(f"s_mov_b32 {s(pre + '_k')}, 0")
    A.e(f"s_add_i32 {t(0)}, {s(pre + '_tile')}, {s('nwg')}")
    A.e(f"s_cmp_ge_u32 {t(0)}, {s('ntile')}")
    A.e(f"s_cbranch_scc1 {lab}_%=")                                # past the last tile: stay (harmless re-fetch)
    A.e(f"s_mov_b32 {s(pre + '_tile')}, {t(0)}")
    A.e(f"s_add_i32 {s(pre + '_rt')}, {s(pre + '_rt')}, {s('drt')}")
    A.e(f"s_add_i32 {s(pre + '_pair')}, {s(pre + '_pair')}, {s('dpair')}")
    A.e(f"s_cmp_lt_u32 {s(pre + '_rt')}, {s('n_rt')}")
    lab2 = A.new("Lcur")
    A.e(f"s_cbranch_scc1 {lab2}_%=")
    A.e(f"s_sub_i32 {s(pre + '_rt')}, {s(pre + '_rt')}, {s('n_rt')}")
    A.e(f"s_add_i32 {s(pre + '_pair')}, {s(pre + '_pair')}, 1")
    A.label(lab2)
    on_wrap(A)
    A.label(lab)


def issue_x(A, slot):
    """this wave's 64 bytes (16 lanes) of the L cursor's step into x-ring slot `slot` (int or SGPR name)"""
    if isinstance(slot, int):
        A.e(f"s_add_i32 {t(6)}, {s('xw')}, {slot * XSLOT}")
    else:
        A.e(f"s_mul_i32 {t(6)}, {slot}, {XSLOT}")
        A.e(f"s_add_i32 {t(6)}, {t(6)}, {s('xw')}")
    A.e(f"s_add_i32 m0, {t(6)}, {s('oXr')}")
    A.e(f"s_lshl_b32 {t(7)}, {s('l_k')}, 7")                       # k * 128 bytes
    A.e(f"s_mov_b64 {t2(16)}, exec")
    A.e("s_mov_b64 exec, 0xffff")
    A.e(f"buffer_load_dword {v('XDMA')}, s[{S['rX']}:{S['rX'] + 3}], {t(7)} offen lds")
    A.e(f"s_mov_b64 exec, {t2(16)}")
    cursor_step(A, "l", set_x_rsrc)


def issue_a(A, slot):
    """weight tile of the A cursor's step: RQ requests of 16 rows x 64 B by this wave"""
    if isinstance(slot, int):
        A.e(f"s_add_i32 {t(6)}, {s('aw')}, {slot * AT}")
    else:
        A.e(f"s_mul_i32 {t(6)}, {slot}, {AT}")
        A.e(f"s_add_i32 {t(6)}, {t(6)}, {s('aw')}")
    A.e(f"s_add_i32 {t(6)}, {t(6)}, {s('oRing')}")
    A.e(f"s_lshl_b32 {t(7)}, {s('a_k')}, 6")                       # k * 64 bytes
    for q in range(RQ):
        if q == 0:
            A.e(f"s_mov_b32 m0, {t(6)}")
        else:
            A.e(f"s_add_i32 m0, {t(6)}, {q * 1024}")
        if not os.environ.get("FUSED_NODMA"):      # lab ablation: the weight ring is never refilled
            A.e(f"buffer_load_dwordx4 v{V['DMA'] + q}, s[{S['rW']}:{S['rW'] + 3}], {t(7)} offen lds")
    cursor_step(A, "a", set_a_rows)


# ---------------------------------------------------------------- fragment generation (one pair of element-candidates)
def gen_pair_ops(e, val0, first):
    """instruction list (strings or ('ds', text, tag)): LUT reads of elements e, e+1 from the log2 values LV[e], LV[e+1]
    (the 16-bit values land in VAL[val0], VAL[val0 + 1]) and |kf - rne(kf)| folded into the unit's running maximum DM"""
    ca, cc, chi = v('PAR', CA), v('PAR', CC), v('PAR', CHI)
    k, tt, x, dm = v('GT', 0), v('GT', 1), v('DX'), v('DM')
    ops = []
    nogen, nolut, notie = os.environ.get("FUSED_NOGEN"), os.environ.get("FUSED_NOLUT"), os.environ.get("FUSED_NOTIE")
    for j in range(2):
        if nogen:                                    # lab ablation: no generation at all (the B fragment keeps its last value)
            continue
        ops += [
            f"v_fma_f32 {k}, {v('LV', e + j)}, {ca}, {cc}",
            f"v_med3_f32 {k}, {k}, 0, {chi}",
            f"v_add_f32 {tt}, 0x{MAGIC:08x}, {k}",
        ]
        if not notie:                                # lab ablation: without the near-tie tracking (3 VALU per element)
            ops += [
                f"v_add_f32 {x}, 0x{(MAGIC ^ 0x80000000):08x}, {tt}",
                f"v_sub_f32 {x}, {k}, {x}",
                (f"v_max_f32 {dm}, abs({x}), abs({x})" if (first and j == 0) else f"v_max_f32 {dm}, abs({x}), {dm}"),
            ]
        ops.append(f"v_lshl_add_u32 {tt}, {tt}, 9, {v('LUTC')}")
        if not nolut:                                # lab ablation: without the LUT reads (VAL keeps its last value)
            ops.append(("ds", f"ds_read_b32 {v('VAL', val0 + j)}, {tt}", f"val{val0 + j}"))
    return ops


def pack_op(bn, i, val0):
    """dword i of B fragment bn from VAL[val0], VAL[val0+1]"""
    return ("pack", f"v_lshl_or_b32 v{V[bn] + i}, {v('VAL', val0 + 1)}, 16, {v('VAL', val0)}", (f"val{val0}", f"val{val0 + 1}"))


def fix_chunk(A, bn, xsrc_vgpr, eo):
    """cold block: the wave's chunk holds a near-tie.  Per element: is any lane within the zone?  (~1.3 % each) -- if so
    the element's bins come from the threshold table for every lane (exact wherever the fast bin is within one of it) and
    the 16-bit value is patched into B fragment bn.  LV holds the chunk's log2 values; x is read from the x run.
    Works in VAL0..3 and GT0..1 only: the A buffers may hold fragments read ahead for the next unit."""
    ca, cc, chi = v('PAR', CA), v('PAR', CC), v('PAR', CHI)
    ra, rb_, rc, rd, re_, rf = v('VAL', 0), v('VAL', 1), v('VAL', 2), v('VAL', 3), v('GT', 0), v('GT', 1)
    A.drain()
    A.e(f"s_sub_i32 {t(10)}, {s('L2')}, 1")
    for e in range(8):
        skip = A.new("Lfe")
        A.e(f"v_fma_f32 {ra}, {v('LV', e)}, {ca}, {cc}")
        A.e(f"v_med3_f32 {ra}, {ra}, 0, {chi}")                    # kf
        A.e(f"v_add_f32 {rb_}, 0x{MAGIC:08x}, {ra}")               # t
        A.e(f"v_add_f32 {rc}, 0x{(MAGIC ^ 0x80000000):08x}, {rb_}")
        A.e(f"v_sub_f32 {rc}, {ra}, {rc}")                         # d
        A.e(f"v_cmp_lt_f32 vcc, {s('tie')}, abs({rc})")
        A.e(f"s_cbranch_vccz {skip}_%=")
        A.ds(f"ds_read_b32 {rf}, {xsrc_vgpr} offset:{(eo + e) * 4}", "xe")
        A.e(f"v_and_b32 {ra}, 0xff, {rb_}")
        A.e(f"v_min_u32 {ra}, {s('L2')}, {ra}")                    # f: fast bin, clamped to 2^bits (= masked)
        A.e(f"v_min_u32 {rb_}, {t(10)}, {ra}")
        A.e(f"v_lshl_add_u32 {rb_}, {rb_}, 9, {v('THRC')}")
        A.e(f"v_subrev_u32 {rc}, 1, {ra}")
        A.e(f"v_max_i32 {rc}, 0, {rc}")
        A.e(f"v_lshl_add_u32 {rc}, {rc}, 9, {v('THRC')}")
        A.ds(f"ds_read_b32 {rd}, {rb_}", "tu")                     # thr[f]
        A.ds(f"ds_read_b32 {re_}, {rc}", "td")                     # thr[f - 1]
        A.drain()
        A.e(f"v_add_f32 {rf}, {s('shift')}, {rf}")                 # xs
        A.e(f"v_cmp_lt_f32 {t2(12)}, {rf}, {rd}")                  # xs < thr[f]
        A.e(f"v_cmp_lt_u32 {t2(14)}, {ra}, {s('L2')}")             # f < 2^bits
        A.e(f"s_and_b64 {t2(12)}, {t2(12)}, {t2(14)}")
        A.e(f"v_cndmask_b32 {rb_}, 0, 1, {t2(12)}")
        A.e(f"v_add_u32 {rb_}, {ra}, {rb_}")
        A.e(f"v_cmp_nlt_f32 {t2(12)}, {rf}, {re_}")                # !(xs < thr[f-1])
        A.e(f"v_cmp_lt_u32 {t2(14)}, 0, {ra}")                     # f > 0
        A.e(f"s_and_b64 {t2(12)}, {t2(12)}, {t2(14)}")
        A.e(f"v_cndmask_b32 {rc}, 0, 1, {t2(12)}")
        A.e(f"v_sub_u32 {rb_}, {rb_}, {rc}")
        A.e(f"v_lshl_add_u32 {rb_}, {rb_}, 9, {v('LUTB')}")
        A.ds(f"ds_read_b32 {rd}, {rb_}", "fx")
        A.drain()
        bdw = f"v{V[bn] + e // 2}"
        if e % 2 == 0:
            A.e(f"v_and_b32 {bdw}, 0xffff0000, {bdw}")
            A.e(f"v_or_b32 {bdw}, {bdw}, {rd}")
        else:
            A.e(f"v_and_b32 {bdw}, 0xffff, {bdw}")
            A.e(f"v_lshl_or_b32 {bdw}, {rd}, 16, {bdw}")
        A.label(skip)


def gen_only(A, bn, xs_vgpr, eo):
    """un-overlapped generation of one K half into bn (used once, before the first step)"""
    A.ds(f"ds_read_b128 {vr('LV', 0, 4)}, {xs_vgpr} offset:{128 + eo * 4}", "l0")
    A.ds(f"ds_read_b128 {vr('LV', 4, 4)}, {xs_vgpr} offset:{128 + eo * 4 + 16}", "l1")
    A.drain()
    for pi in range(4):
        for op in gen_pair_ops(2 * pi, 2 * (pi & 1), pi == 0):
            if isinstance(op, tuple):
                A.ds(op[1], op[2])
            else:
                A.e(op)
        A.drain()
        A.e(pack_op(bn, pi, 2 * (pi & 1))[1])
    lab = A.new("Lfix")
    A.e(f"v_cmp_lt_f32 vcc, {s('tie')}, {v('DM')}")
    A.e(f"s_cbranch_vccz {lab}_%=")
    fix_chunk(A, bn, xs_vgpr, eo)
    A.label(lab)


def abuf(h, rb):
    b = V['ABUF'] + 4 * ((h * NRB + rb) % 4)
    return f"v[{b}:{b + 3}]"


def a_read(A, h, rb):
    if os.environ.get("FUSED_NOAREAD"):              # lab ablation: the MFMAs run on whatever the A buffers hold (no LDS fragment reads)
        return
    A.ds(f"ds_read_b128 {abuf(h, rb)}, {v('AS0' if h == 0 else 'AS1')} offset:{rb * 2048}", f"a{h}_{rb}")


def unit_begin(A, h, xs_vgpr, eo, a_ahead):
    """the log2 values of the chunk this unit generates (and, when nothing was read ahead, its first A fragments)"""
    A.ds(f"ds_read_b128 {vr('LV', 0, 4)}, {xs_vgpr} offset:{128 + eo * 4}", f"l{h}_0")
    A.ds(f"ds_read_b128 {vr('LV', 4, 4)}, {xs_vgpr} offset:{128 + eo * 4 + 16}", f"l{h}_1")
    if not a_ahead:
        for rb in range(min(3, NRB)):
            a_read(A, h, rb)


def unit(A, h, bc, bn, xs_vgpr, eo, cold_blocks, first_fill_rb):
    """half a K-step: NRB MFMAs of K half h from B fragment bc, interleaved with the generation of the next half's
    fragment (bn) from the log2 values unit_begin requested.  A fragments run three row blocks ahead, across the boundary
    between the two units of a step (same ring slot).  Appends a cold block descriptor."""
    fill = []
    for pi in range(4):
        ops = gen_pair_ops(2 * pi, 2 * (pi & 1), pi == 0)
        if pi >= 2:
            # pair pi - 2 is packed right before this pair's LUT reads reuse its VAL registers: a full pair (and the MFMAs
            # in between) after its own reads were issued
            ops.insert(min(7, len(ops)), pack_op(bn, pi - 2, 2 * (pi & 1)))
        if pi == 0:
            ops.insert(0, ("waitfor", f"l{h}_0"))
        if pi == 2:
            ops.insert(0, ("waitfor", f"l{h}_1"))
        fill.extend(ops)
    tail = [pack_op(bn, 2, 0), pack_op(bn, 3, 2)]                   # after the last MFMA: their reads went out two MFMAs earlier
    nf = len(fill)
    fi = 0

    def emit_fill(k):
        nonlocal fi
        for _ in range(k):
            if fi >= nf:
                return
            op = fill[fi]
            fi += 1
            if isinstance(op, tuple):
                if op[0] == "ds":
                    A.ds(op[1], op[2])
                elif op[0] == "pack":
                    for tg in op[2]:
                        A.wait(tg)
                    A.e(op[1])
                elif op[0] == "waitfor":
                    A.wait(op[1])
            else:
                A.e(op)

    n_slots = max(1, NRB - 1 - first_fill_rb)                      # the generation ends one MFMA before the unit does
    per = (nf + n_slots - 1) // n_slots
    for rb in range(NRB):
        nxt = rb + 3
        if nxt < NRB:
            a_read(A, h, nxt)
        elif h == 0 and nxt - NRB < min(3, NRB):
            a_read(A, 1, nxt - NRB)                                # the second unit's first fragments (same ring slot)
        A.wait(f"a{h}_{rb}")
        if not os.environ.get("FUSED_NOMFMA"):
            A.e(f"v_mfma_f32_32x32x16_bf16 {acc(rb)}, {abuf(h, rb)}, v[{V[bc]}:{V[bc] + 3}], {acc(rb)}")
        if rb >= first_fill_rb:
            emit_fill(per)
    emit_fill(nf)
    fill.extend(tail)
    nf = len(fill)
    emit_fill(nf)
    if not os.environ.get("FUSED_NOCOLD"):
        lab = A.new("Lcold")
        A.e(f"v_cmp_lt_f32 vcc, {s('tie')}, {v('DM')}")             # any element-candidate of the chunk near a rounding tie?
        A.e(f"s_cbranch_vccnz {lab}_%=")
        A.label(lab + "r")
        cold_blocks.append((lab, bn, xs_vgpr, eo, list(A.fifo)))


def epilogue(A):
    A.c("epilogue: e = refb - (D * alpha) * rs ; s += e^2")
    A.e("s_nop 7")
    A.e("s_nop 7")
    A.e("s_nop 7")                                                 # MFMA results -> VALU reads
    E0 = V['ABUF']
    s0 = v('LV', 4)
    epi, epr, xora = v('LV', 5), v('LV', 6), v('LV', 7)
    # read bases: refb + wtok*ROWS*4 + fkg*16 ; rs + fkg*16 ; bpermute partner lane ^ 32
    A.e(f"s_lshr_b32 {t(5)}, {s('w')}, 2")
    A.e(f"s_mul_i32 {t(5)}, {t(5)}, {ROWS * 4}")
    A.e(f"s_add_i32 {t(5)}, {t(5)}, {s('oRefb')}")
    A.e(f"v_lshl_add_u32 {epi}, {v('FKG')}, 4, {t(5)}")
    A.e(f"v_lshl_add_u32 {epr}, {v('FKG')}, 4, {s('oRs')}")
    A.e(f"v_xor_b32 {xora}, 32, {v('LANE')}")
    A.e(f"v_lshlrev_b32 {xora}, 2, {xora}")
    A.e(f"v_mov_b32 {s0}, 0")
    reads = [(rb, i4) for rb in range(NRB) for i4 in range(4)]

    def issue(gi):
        rb, i4 = reads[gi]
        b = E0 + 8 * (gi & 1)
        off = (rb * 32 + 8 * i4) * 4
        A.ds(f"ds_read_b128 v[{b}:{b + 3}], {epi} offset:{off}", f"rf{gi}")
        A.ds(f"ds_read_b128 v[{b + 4}:{b + 7}], {epr} offset:{off}", f"rs{gi}")
    issue(0)
    for gi, (rb, i4) in enumerate(reads):
        if gi + 1 < len(reads):
            issue(gi + 1)
        A.wait(f"rs{gi}")
        b = E0 + 8 * (gi & 1)
        for j in range(4):
            tmp = v('LV', j)
            if rb < NA:
                A.e(f"v_accvgpr_read_b32 {tmp}, a{16 * rb + 4 * i4 + j}")
                A.e(f"v_mul_f32 {tmp}, {tmp}, {v('PAR', AL)}")
            else:
                A.e(f"v_mul_f32 {tmp}, v{64 + 16 * (rb - NA) + 4 * i4 + j}, {v('PAR', AL)}")
            A.e(f"v_fma_f32 {tmp}, -{tmp}, v{b + 4 + j}, v{b + j}")
            A.e(f"v_fma_f32 {s0}, {tmp}, {tmp}, {s0}")
    # lanes l and l + 32 hold the two row halves of a candidate
    A.ds(f"ds_bpermute_b32 {v('LV', 0)}, {xora}, {s0}", "bp0")
    A.drain()
    A.e(f"v_add_f32 {s0}, {s0}, {v('LV', 0)}")
    # run += (double) s  when this wave's token exists (t(9) = 1)
    lab = A.new("Lnotok")
    A.e(f"s_cmp_eq_u32 {t(9)}, 0")
    A.e(f"s_cbranch_scc1 {lab}_%=")
    A.e(f"v_cvt_f64_f32 {vr('LV', 0, 2)}, {s0}")
    A.e(f"v_add_f64 {vr('RUN', 0, 2)}, {vr('RUN', 0, 2)}, {vr('LV', 0, 2)}")
    A.label(lab)


def tile_setup(A):
    """epilogue operands of the compute tile -> LDS; accumulators zeroed.  Uses the compute cursor (c_tile)."""
    A.c("tile: m0, tok0, staging of ref - row_bias and row_scale, zero accumulators")
    A.e(f"s_mov_b32 {t(0)}, {s('c_pair')}")
    A.e(f"s_mov_b32 {t(1)}, {s('c_rt')}")
    A.e(f"s_lshl_b32 {t(2)}, {t(0)}, 1")                           # tok0
    A.e(f"s_mul_i32 {t(3)}, {t(1)}, {ROWS}")                       # m0
    A.e(f"s_lshr_b32 {t(4)}, {s('w')}, 2")
    A.e(f"s_add_i32 {t(4)}, {t(4)}, {t(2)}")
    A.e(f"s_cmp_lt_i32 {t(4)}, {s('T')}")
    A.e(f"s_cselect_b32 {t(9)}, 1, 0")                             # tok_ok for this wave: tok0 + wtok < T
    NT = 64 * NW
    tid, e_, ts, r_, row, tk, off, a4 = (v('TV', i) for i in range(8))
    A.e(f"s_lshl_b32 {t(5)}, {s('w')}, 6")
    A.e(f"v_add_u32 {tid}, {t(5)}, {v('LANE')}")                   # thread id within the workgroup
    EU = (2 * ROWS + NT - 1) // NT
    RU = (ROWS + NT - 1) // NT
    RF, RBV, RSV = V['TV'] + 8, V['TV'] + 10, V['TV'] + 12         # loaded ref / row_bias / row_scale values (EU, EU, RU <= 2)
    assert EU <= 2 and RU <= 2
    A.e("s_waitcnt vmcnt(0)")
    A.e(f"s_mul_i32 {t(20)}, {t(2)}, {s('M')}")                    # tok0 * M (64 bit) * 4 + pRef
    A.e(f"s_mul_hi_u32 {t(21)}, {t(2)}, {s('M')}")
    A.e(f"s_lshl_b64 {t2(20)}, {t2(20)}, 2")
    A.e(f"s_add_u32 {t(20)}, {t(20)}, {s('pRef_lo')}")
    A.e(f"s_addc_u32 {t(21)}, {t(21)}, {s('pRef_hi')}")
    A.e(f"s_movk_i32 {t(7)}, {2 * ROWS}")
    A.e(f"s_movk_i32 {t(8)}, {ROWS}")
    for u in range(EU):
        # e = tid + u*NT ; ts = e >= ROWS ; r = e - ts*ROWS ; row = m0 + r ; tk = tok0 + ts
        A.e(f"v_add_u32 {e_}, {u * NT}, {tid}")
        A.e(f"v_cmp_le_u32 vcc, {t(8)}, {e_}")
        A.e(f"v_cndmask_b32 {ts}, 0, 1, vcc")
        A.e(f"v_mul_u32_u24 {r_}, {t(8)}, {ts}")
        A.e(f"v_sub_u32 {r_}, {e_}, {r_}")
        A.e(f"v_add_u32 {row}, {t(3)}, {r_}")
        A.e(f"v_add_u32 {tk}, {t(2)}, {ts}")
        # ok = e < 2*ROWS && row < M && tk < T
        A.e(f"v_cmp_gt_u32 {t2(12)}, {t(7)}, {e_}")
        A.e(f"v_cmp_gt_i32 {t2(14)}, {s('M')}, {row}")
        A.e(f"s_and_b64 {t2(12)}, {t2(12)}, {t2(14)}")
        A.e(f"v_cmp_gt_i32 {t2(14)}, {s('T')}, {tk}")
        A.e(f"s_and_b64 {t2(12)}, {t2(12)}, {t2(14)}")
        # ref[(tok0 + ts)*M + row]: 64-bit tile base in s[t20:t21], 32-bit lane offset (ts*M + row)*4
        A.e(f"v_mul_lo_u32 {off}, {ts}, {s('M')}")
        A.e(f"v_add_u32 {off}, {off}, {row}")
        A.e(f"v_lshlrev_b32 {off}, 2, {off}")
        A.e(f"v_mov_b32 v{RF + u}, 0")
        A.e(f"v_mov_b32 v{RBV + u}, 0")
        A.e(f"s_and_saveexec_b64 {t2(16)}, {t2(12)}")
        A.e(f"global_load_dword v{RF + u}, {off}, {t2(20)}")
        labn = A.new("Lnorb")
        A.e(f"s_or_b32 {t(18)}, {s('pRb_lo')}, {s('pRb_hi')}")
        A.e(f"s_cmp_eq_u32 {t(18)}, 0")
        A.e(f"s_cbranch_scc1 {labn}_%=")
        A.e(f"v_lshlrev_b32 {a4}, 2, {row}")
        A.e(f"global_load_dword v{RBV + u}, {a4}, s[{S['pRb_lo']}:{S['pRb_hi']}]")
        A.label(labn)
        A.e(f"s_mov_b64 exec, {t2(16)}")
    for u in range(RU):
        A.e(f"v_add_u32 {e_}, {u * NT}, {tid}")                    # r
        A.e(f"v_add_u32 {row}, {t(3)}, {e_}")
        A.e(f"v_cmp_gt_u32 {t2(12)}, {t(8)}, {e_}")
        A.e(f"v_cmp_gt_i32 {t2(14)}, {s('M')}, {row}")
        A.e(f"s_and_b64 {t2(12)}, {t2(12)}, {t2(14)}")
        A.e(f"v_mov_b32 v{RSV + u}, 0")
        A.e(f"v_lshlrev_b32 {a4}, 2, {row}")
        A.e(f"s_and_saveexec_b64 {t2(16)}, {t2(12)}")
        A.e(f"global_load_dword v{RSV + u}, {a4}, s[{S['pRs_lo']}:{S['pRs_hi']}]")
        A.e(f"s_mov_b64 exec, {t2(16)}")
    A.e("s_waitcnt vmcnt(0)")
    A.e("s_barrier")                                               # every wave is past the previous tile's epilogue reads
    for u in range(EU):
        A.e(f"v_add_u32 {e_}, {u * NT}, {tid}")
        A.e(f"v_cmp_gt_u32 vcc, {t(7)}, {e_}")
        A.e(f"v_sub_f32 v{RF + u}, v{RF + u}, v{RBV + u}")
        A.e(f"v_lshl_add_u32 {r_}, {e_}, 2, {s('oRefb')}")
        A.e(f"s_and_saveexec_b64 {t2(16)}, vcc")
        A.e(f"ds_write_b32 {r_}, v{RF + u}")
        A.e(f"s_mov_b64 exec, {t2(16)}")
    for u in range(RU):
        A.e(f"v_add_u32 {e_}, {u * NT}, {tid}")
        A.e(f"v_cmp_gt_u32 vcc, {t(8)}, {e_}")
        A.e(f"v_lshl_add_u32 {r_}, {e_}, 2, {s('oRs')}")
        A.e(f"s_and_saveexec_b64 {t2(16)}, vcc")
        A.e(f"ds_write_b32 {r_}, v{RSV + u}")
        A.e(f"s_mov_b64 exec, {t2(16)}")
    A.c("zero the accumulators")
    for r in range(16 * NA):
        A.e(f"v_accvgpr_write_b32 a{r}, 0")
    for r in range(64, 64 + 16 * (NRB - NA)):
        A.e(f"v_mov_b32 v{r}, 0")
    A.e("s_nop 4")


STAGE = int(os.environ.get("FUSED_STAGE", "99"))      # debugging: stop after a phase (1 setup, 2 prologue, 3 tile setup, 4 one step)


def program():
    A = Asm()
    cfg_load(A)
    lane_setup(A)
    if STAGE <= 1:
        A.e("s_branch Lend_%=")
    for pre in ("a", "l"):
        A.e(f"s_mov_b32 {s(pre + '_tile')}, {s('bid')}")
        A.e(f"s_mov_b32 {s(pre + '_pair')}, {s('pair0')}")
        A.e(f"s_mov_b32 {s(pre + '_rt')}, {s('rt0')}")
        A.e(f"s_mov_b32 {s(pre + '_k')}, 0")
    A.e(f"s_mov_b32 {s('c_tile')}, {s('bid')}")
    A.e(f"s_mov_b32 {s('c_pair')}, {s('pair0')}")
    A.e(f"s_mov_b32 {s('c_rt')}, {s('rt0')}")
    set_x_rsrc(A)
    set_a_rows(A)
    A.c("prologue: x/log2 of step 0, then weights of steps 0..FNS-2 with the x/log2 of the following step")
    issue_x(A, 0)
    for s0 in range(FNS - 1):
        issue_a(A, s0)
        issue_x(A, s0 + 1)
    A.e(f"s_mov_b32 {s('stA')}, 0")
    A.e(f"s_mov_b32 {s('stX')}, 0")
    A.c("B fragment of the very first K half")
    A.e("s_waitcnt vmcnt(0)")
    A.e("s_barrier")                                               # the x / log2 parts of the slot come from four waves
    A.e(f"v_mov_b32 {v('XSC')}, {v('XOFF')}")
    gen_only(A, "BA", v("XSC"), 0)
    if STAGE <= 2:
        A.e("s_branch Lend_%=")

    A.label("Ltile")
    tile_setup(A)
    if STAGE <= 3:
        A.e("s_branch Lend_%=")
    A.e(f"s_mov_b32 {s('kt')}, {s('nk')}")
    A.label("Lstep")
    cold = []
    A.e(f"s_waitcnt vmcnt({(FNS - 2) * (RQ + 1)})")
    if not os.environ.get("FUSED_NOBAR"):
        A.e("s_barrier")
    # ring addresses of this step
    A.e(f"s_mul_i32 {t(0)}, {s('stA')}, {AT}")
    A.e(f"s_add_i32 {t(0)}, {t(0)}, {s('oRing')}")
    A.e(f"v_add_u32 {v('AS0')}, {t(0)}, {v('AOFF0')}")
    A.e(f"v_add_u32 {v('AS1')}, {t(0)}, {v('AOFF1')}")
    A.e(f"s_mul_i32 {t(1)}, {s('stX')}, {XSLOT}")
    A.e(f"v_add_u32 {v('XSC')}, {t(1)}, {v('XOFF')}")
    A.e(f"s_add_i32 {t(2)}, {s('stX')}, 1")
    A.e(f"s_cmp_eq_u32 {t(2)}, {XS}")
    A.e(f"s_cselect_b32 {t(2)}, 0, {t(2)}")                        # next x slot
    A.e(f"s_mul_i32 {t(3)}, {t(2)}, {XSLOT}")
    A.e(f"v_add_u32 {v('XSN')}, {t(3)}, {v('XOFF')}")
    A.e(f"s_mov_b32 {t(19)}, {t(2)}")                              # keep the next x slot
    unit_begin(A, 0, v("XSC"), 16, False)                          # this step's first reads fly while the DMA work issues
    # DMA for step n + FNS - 1 (weights -> the slot step n - 1 used) and n + FNS (x/log2 -> the x slot step n - 1 used)
    A.e(f"s_add_i32 {t(4)}, {s('stA')}, {FNS - 1}")
    A.e(f"s_cmp_ge_u32 {t(4)}, {FNS}")
    A.e(f"s_cselect_b32 {t(5)}, {FNS}, 0")
    A.e(f"s_sub_i32 {t(4)}, {t(4)}, {t(5)}")
    issue_a(A, t(4))
    A.e(f"s_add_i32 {t(4)}, {s('stX')}, {XS - 1}")
    A.e(f"s_cmp_ge_u32 {t(4)}, {XS}")
    A.e(f"s_cselect_b32 {t(5)}, {XS}, 0")
    A.e(f"s_sub_i32 {t(4)}, {t(4)}, {t(5)}")
    issue_x(A, t(4))
    unit(A, 0, "BA", "BB", v("XSC"), 16, cold, 0)
    unit_begin(A, 1, v("XSN"), 0, True)
    unit(A, 1, "BB", "BA", v("XSN"), 0, cold, 1)
    A.drain()
    # advance the ring slots
    A.e(f"s_add_i32 {s('stA')}, {s('stA')}, 1")
    A.e(f"s_cmp_eq_u32 {s('stA')}, {FNS}")
    A.e(f"s_cselect_b32 {s('stA')}, 0, {s('stA')}")
    A.e(f"s_mov_b32 {s('stX')}, {t(19)}")
    if STAGE <= 4:
        A.e("s_branch Lend_%=")
    A.e(f"s_sub_i32 {s('kt')}, {s('kt')}, 1")
    A.e(f"s_cmp_lg_u32 {s('kt')}, 0")
    A.e("s_cbranch_scc1 Lstep_%=")
    epilogue(A)
    A.e(f"s_add_i32 {s('c_tile')}, {s('c_tile')}, {s('nwg')}")
    A.e(f"s_add_i32 {s('c_pair')}, {s('c_pair')}, {s('dpair')}")
    A.e(f"s_add_i32 {s('c_rt')}, {s('c_rt')}, {s('drt')}")
    A.e(f"s_cmp_lt_u32 {s('c_rt')}, {s('n_rt')}")
    labc = A.new("Lcrt")
    A.e(f"s_cbranch_scc1 {labc}_%=")
    A.e(f"s_sub_i32 {s('c_rt')}, {s('c_rt')}, {s('n_rt')}")
    A.e(f"s_add_i32 {s('c_pair')}, {s('c_pair')}, 1")
    A.label(labc)
    A.e(f"s_cmp_lt_u32 {s('c_tile')}, {s('ntile')}")
    A.e("s_cbranch_scc1 Ltile_%=")
    A.e("s_branch Lend_%=")
    A.c("cold blocks: chunks with a near-tie")
    for lab, bn, xs_vgpr, eo, fifo in cold:
        A.label(lab)
        A.fifo = list(fifo)
        fix_chunk(A, bn, xs_vgpr, eo)
        A.e(f"s_branch {lab}r_%=")
    A.label("Lend")
    A.fifo = []
    A.c("per-lane sums -> s_fin[w][32] (lanes 0..31 hold the wave's candidates)")
    A.e("s_waitcnt vmcnt(0)")
    A.e(f"s_lshl_b32 {t(0)}, {s('w')}, 8")
    A.e(f"s_add_i32 {t(0)}, {t(0)}, {s('oFin')}")
    A.e(f"v_lshl_add_u32 {v('TV')}, {v('FROW')}, 3, {t(0)}")
    A.e(f"v_cmp_eq_u32 vcc, 0, {v('FKG')}")
    A.e(f"s_and_saveexec_b64 {t2(16)}, vcc")
    A.e(f"ds_write_b64 {v('TV')}, {vr('RUN', 0, 2)}")
    A.e(f"s_mov_b64 exec, {t2(16)}")
    A.e("s_waitcnt lgkmcnt(0)")
    return A


def _sgprs(text):
    """SGPR numbers (and 'vcc') named in an operand string"""
    out = set()
    for a, b in re.findall(r"s\[(\d+):(\d+)\]", text):
        out.update(range(int(a), int(b) + 1))
    out.update(int(n) for n in re.findall(r"(?<![a-z_\[\d:])s(\d+)", text))
    if "vcc" in text:
        out.add("vcc")
    return out


def hazard_pass(lines):
    """gfx940-class software hazards the compiler would pad and hand-written code must (GCNHazardRecognizer, VDecCoExec set):
      * VALU writes an SGPR / VCC  ->  VALU or SALU reading it: 2 wait states;  VMEM reading it: 5;  branch on vccz: 2
      * VALU writes a VGPR         ->  v_readlane / v_readfirstlane of it: 1 wait state
    Every instruction in between counts as one wait state."""
    out = []
    recent = []          # (age, written sgprs, written vgpr) of the latest VALU instructions that matter
    for ln in lines:
        tl = ln.strip()
        if not tl or tl.startswith(";") or tl.endswith(":"):
            out.append(ln)
            if tl.endswith(":"):
                recent = []                       # labels: be conservative the cheap way (targets are padded by their sources)
            continue
        op, _, rest = tl.partition(" ")
        ops = [o.strip() for o in rest.split(",")] if rest else []
        need = 0
        reads = _sgprs(",".join(ops[1:])) if op.startswith(("v_", "buffer_", "global_", "ds_")) else _sgprs(rest if not op.startswith("s_") else ",".join(ops[1:]) if len(ops) > 1 else rest)
        if op.startswith("s_cbranch_vcc"):
            reads = {"vcc"}
        if op in ("s_and_saveexec_b64",):
            reads = _sgprs(ops[1])
        is_vmem = op.startswith(("buffer_", "global_"))
        for age, wr_s, wr_v in recent:
            if wr_s & reads:
                need = max(need, (5 if is_vmem else 2) - age)
            if wr_v is not None and op in ("v_readlane_b32", "v_readfirstlane_b32") and len(ops) > 1 and ops[1] == wr_v:
                need = max(need, 1 - age)
        if need > 0:
            out.append(f"s_nop {need - 1}")
            recent = [(a + need, s_, v_) for a, s_, v_ in recent]
        out.append(ln)
        recent = [(a + 1, s_, v_) for a, s_, v_ in recent if a + 1 < 5]
        if op.startswith("v_"):
            wr_s = set()
            if op.startswith("v_cmp") or op in ("v_readlane_b32", "v_readfirstlane_b32"):
                wr_s = _sgprs(ops[0])
            if op in ("v_add_co_u32", "v_addc_co_u32", "v_sub_co_u32", "v_subb_co_u32"):
                wr_s = _sgprs(ops[1])
            wr_v = ops[0] if ops and re.fullmatch(r"v\d+", ops[0]) else None
            recent.append((0, wr_s, wr_v))
    return out


def main():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    A = program()
    A.lines = hazard_pass(A.lines)
    used_v = max(int(n) for ln in A.lines if not ln.startswith(";") for n in re.findall(r"(?<![a-z_\d])v\[?(\d+)", ln))
    used_v2 = max([int(b) for ln in A.lines if not ln.startswith(";") for _, b in re.findall(r"v\[(\d+):(\d+)\]", ln)] + [0])
    assert max(used_v, used_v2) < 128, f"VGPR budget exceeded: v{max(used_v, used_v2)}"
    out = os.path.join(root, "adalog_amd", "csrc", f"fused_loop_nrb{NRB}_s{FNS}.inc")
    with open(out, "w") as f:
        f.write(f"// GENERATED by tools/gen_fused_asm.py (NRB = {NRB}, FNS = {FNS}) -- do not edit.\n")
        f.write(f"// Config array (dword index): {', '.join(f'{i}={n}' for i, n in enumerate(CFG))}\n")
        for ln in A.lines:
            if ln.startswith(";"):
                f.write(f"// {ln[2:]}\n")
            else:
                f.write('"' + ln + '\\n\\t"\n')
    with open(out.replace(".inc", ".lst"), "w") as f:
        f.write("\n".join(A.lines) + "\n")
    n_mfma = sum(1 for l in A.lines if l.startswith("v_mfma"))
    print(f"{out}: {len(A.lines)} lines, {n_mfma} MFMAs, highest VGPR v{max(used_v, used_v2)}")


if __name__ == "__main__":
    main()
