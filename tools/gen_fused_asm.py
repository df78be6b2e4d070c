#!/usr/bin/env python3
"""Generator of the hand-scheduled CDNA4 (gfx950) main loop of the fused activation-search kernel (gemm_fused.hip).

    FUSED_NRB=12 FUSED_FNS=4 python tools/gen_fused_asm.py   -> adalog_amd/csrc/fused_loop_nrb12_s4.inc  (+ .lst, a listing)
    tools/gen_fused_asm_all.sh                               -> the five variants the library ships

Why a generator: the compiler keeps at most 256 accumulator registers per wave (one MFMA form per function), reads every
A fragment right in front of its MFMAs (exposed LDS latency) and cannot be steered into interleaving the fragment
generation with the matrix stream.  Here every register is placed by hand -- 16 accumulator tiles in AGPRs, 8 in
v128..v255, everything else below v128 -- and the issue order of one K-step is laid out explicitly: per MFMA about seven
other instructions (VALU of the fragment generation, LUT / fragment reads two MFMA pairs ahead, DMA requests), with
counted s_waitcnt lgkmcnt from a model of the in-order LDS queue.

The text is one inline-asm block (all VGPRs / AGPRs and s8..s99 clobbered); the HIP kernel around it builds the LDS
tables, hands the scalars over through an LDS config array and turns the per-lane sums into the workgroup's output row.
See gemm_fused.hip for the algorithm; names here follow it.
"""
import os
import sys

NRB = int(os.environ.get("FUSED_NRB", "12"))      # row blocks of 32 output channels per tile (12, 8, 6 or 4)
FNS = int(os.environ.get("FUSED_FNS", "4"))       # weight-ring stages (3 where the 6-bit tables need the LDS)
XS = FNS + 1      # x / log2 ring slots
AT = NRB * 2048   # weight tile bytes per stage
ROWS = 32 * NRB
NA = min(NRB, 8)  # row blocks whose tiles live in AGPRs (2 tiles each); the rest sit in v128..
MAGIC = 0x4B400000
TIE_BITS = 0x3EFFF2E5   # 0.4999f
# (v_pk_fma/add_f32 for the pairwise arithmetic was measured 22 % SLOWER beside the MFMAs: not generated any more)

# ---------------------------------------------------------------- config array in LDS (dword indices), written by the kernel
CFG = ["pW_lo", "pW_hi", "pX_lo", "pX_hi", "pL_lo", "pL_hi", "pRef_lo", "pRef_hi", "pRs_lo", "pRs_hi", "pRb_lo", "pRb_hi",
       "M", "T", "K", "Kb", "nk", "n_rt", "ntile", "nwg", "bid", "L2", "shift", "w",
       "oRing", "oXr", "oLut", "oThr", "oPar", "oRefb", "oRs", "oFin", "dpair", "drt", "pair0", "rt0", "tie",
       "pF_lo", "pF_hi", "fpitch"]
S = {n: 40 + i for i, n in enumerate(CFG)}          # s40 .. s75
S["tie"] = 27                                        # = t(19)'s old home is s27: see below (TIE lives in s19)
S.update(rW=76, rX=80, rL=84,                        # buffer resources (4 SGPRs each)
         a_tile=88, a_pair=89, a_rt=90, a_k=91, l_tile=92, l_pair=93, l_rt=94, l_k=95,
         c_tile=96, kt=97, stA=98, stX=99, c_pair=30, c_rt=31)
S["tie"] = 19                                        # t(11): the near-tie threshold, loaded from the config
S.update(pF_lo=34, pF_hi=35, fpitch=36, sh2=37)      # near-tie flag rows (k_tie_flags); sh2 = 2 * (w & 1)
T0 = 8                                               # s8 .. s31: temporaries


def s(name):
    return f"s{S[name]}"


def t(i):
    return f"s{T0 + i}"


def t2(i):
    return f"s[{T0 + i}:{T0 + i + 1}]"


# ---------------------------------------------------------------- VGPR map
V = dict(LANE=0, FROW=1, FKG=2, TMP=3, PAR0=4, PAR1=8, LUTC0=12, LUTC1=13, THRC0=14, THRC1=15, AOFF0=16, AOFF1=17, AS0=18, AS1=19,
         XOFF=20, XSC=21, XSN=22, XDMA=23, DMA=24, EPI=30, EPR=31, XORA=32, LUTB0=33, LUTB1=34, ZERO=35, RUN0=36, RUN1=38,
         BA=40, BB=48, LV=56, ABUF=64, VAL=84, GT=100, FLGC=116, FLGN=117, XV=120, E0=64, )
CA, CC, CHI, AL = 0, 1, 2, 3


def v(name, off=0):
    return f"v{V[name] + off}"


def vr(name, off, n):
    a = V[name] + off
    return f"v[{a}:{a + n - 1}]"


def acc(rb, cb):
    """register operand of accumulator tile (rb, cb)"""
    tix = rb * 2 + cb
    if rb < NA:
        return f"a[{16 * tix}:{16 * tix + 15}]"
    b = 128 + 16 * (tix - 2 * NA)
    return f"v[{b}:{b + 15}]"


def acc_elem(rb, cb, i):
    tix = rb * 2 + cb
    if rb < NA:
        return ("a", 16 * tix + i)
    return ("v", 128 + 16 * (tix - 2 * NA) + i)


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
    A.c("lane geometry")
    A.e(f"v_and_b32 {v('FROW')}, 31, {v('LANE')}")
    A.e(f"v_lshrrev_b32 {v('FKG')}, 5, {v('LANE')}")
    A.e(f"v_mov_b32 {v('ZERO')}, 0")
    A.e(f"s_lshr_b32 {t(0)}, {s('w')}, 1")                       # wtok
    A.e(f"s_and_b32 {t(1)}, {s('w')}, 1")                        # cbp
    # candidate index c0 = 64*cbp + frow ; c1 = c0 + 32
    A.e(f"s_lshl_b32 {t(2)}, {t(1)}, 6")
    A.e(f"v_add_u32 v3, {t(2)}, {v('FROW')}")                    # c0
    # parameters {-37/q, log2(s)*37/q, hi, s*sa_mul}: float4 at oPar + c*16
    A.e(f"v_lshl_add_u32 v60, v3, 4, {s('oPar')}")
    A.ds(f"ds_read_b128 {vr('PAR0', 0, 4)}, v60", "p0")
    A.ds(f"ds_read_b128 {vr('PAR1', 0, 4)}, v60 offset:512", "p1")
    # LUT / threshold lane bases
    A.e(f"v_lshl_add_u32 {v('LUTB0')}, v3, 2, {s('oLut')}")
    A.e(f"v_add_u32 {v('LUTB1')}, 128, {v('LUTB0')}")
    A.e(f"s_mov_b32 {t(3)}, 0x{(-(MAGIC << 9)) & 0xFFFFFFFF:08x}")
    A.e(f"v_add_u32 {v('LUTC0')}, {t(3)}, {v('LUTB0')}")
    A.e(f"v_add_u32 {v('LUTC1')}, {t(3)}, {v('LUTB1')}")
    A.e(f"v_lshl_add_u32 {v('THRC0')}, v3, 2, {s('oThr')}")
    A.e(f"v_add_u32 {v('THRC1')}, 128, {v('THRC0')}")
    # A fragment lane offsets: frow*64 + (((2h + fkg) ^ ((frow >> 2) & 3)) << 4)
    A.e(f"v_lshrrev_b32 v60, 2, {v('FROW')}")
    A.e("v_and_b32 v60, 3, v60")                                   # sw
    A.e(f"v_xor_b32 v61, {v('FKG')}, v60")                         # h = 0: fkg ^ sw
    A.e(f"v_lshlrev_b32 v62, 6, {v('FROW')}")
    A.e(f"v_lshl_add_u32 {v('AOFF0')}, v61, 4, v62")
    A.e("v_xor_b32 v61, 2, v61")                                   # h = 1: (2 + fkg) ^ sw = (fkg ^ sw) ^ 2
    A.e(f"v_lshl_add_u32 {v('AOFF1')}, v61, 4, v62")
    # x / log2 read offset inside the wave's 512-byte slot part: fkg * 32 bytes (+256 for log2)
    A.e(f"s_lshl_b32 {t(4)}, {s('w')}, 9")
    A.e(f"s_add_i32 {t(4)}, {t(4)}, {s('oXr')}")
    A.e(f"v_lshl_add_u32 {v('XOFF')}, {v('FKG')}, 5, {t(4)}")
    # x / log2 DMA voffset: (wtok * K + lane) * 4
    A.e(f"s_mul_i32 {t(5)}, {t(0)}, {s('K')}")
    A.e(f"v_add_u32 v60, {t(5)}, {v('LANE')}")
    A.e(f"v_lshlrev_b32 {v('XDMA')}, 2, v60")
    # epilogue read bases: refb + wtok*ROWS*4 + fkg*16 ; rs + fkg*16
    A.e(f"s_mul_i32 {t(5)}, {t(0)}, {ROWS * 4}")
    A.e(f"s_add_i32 {t(5)}, {t(5)}, {s('oRefb')}")
    A.e(f"v_lshl_add_u32 {v('EPI')}, {v('FKG')}, 4, {t(5)}")
    A.e(f"v_lshl_add_u32 {v('EPR')}, {v('FKG')}, 4, {s('oRs')}")
    A.e(f"v_xor_b32 v60, 32, {v('LANE')}")
    A.e(f"v_lshlrev_b32 {v('XORA')}, 2, v60")
    for r in range(4):
        A.e(f"v_mov_b32 v{V['RUN0'] + r}, 0")
    A.drain()
    # buffer resources: W {ptr, 0 stride, M*Kb records, 0x00020000}
    A.e(f"s_mov_b32 s{S['rW']}, {s('pW_lo')}")
    A.e(f"s_and_b32 s{S['rW'] + 1}, {s('pW_hi')}, 0xffff")
    A.e(f"s_mul_i32 s{S['rW'] + 2}, {s('M')}, {s('Kb')}")
    A.e(f"s_mov_b32 s{S['rW'] + 3}, 0x00020000")
    A.e(f"s_mov_b32 s{S['rX'] + 3}, 0x00020000")
    A.e(f"s_mov_b32 s{S['rL'] + 3}, 0x00020000")
    A.e(f"s_and_b32 {s('sh2')}, {s('w')}, 1")
    A.e(f"s_lshl_b32 {s('sh2')}, {s('sh2')}, 1")
    A.e(f"v_mov_b32 {v('FLGC')}, 0")
    A.e(f"v_mov_b32 {v('FLGN')}, 0")


def set_x_rsrc(A):
    """x / log2 resources of the L cursor's token pair: base + tok0*K*4, records = min(2, T - tok0)*K*4"""
    A.e(f"s_lshl_b32 {t(0)}, {s('l_pair')}, 1")                    # tok0
    A.e(f"s_mul_i32 {t(2)}, {t(0)}, {s('K')}")
    A.e(f"s_mul_hi_u32 {t(3)}, {t(0)}, {s('K')}")
    A.e(f"s_lshl_b64 {t2(2)}, {t2(2)}, 2")                         # byte offset (64 bit)
    A.e(f"s_add_u32 s{S['rX']}, {s('pX_lo')}, {t(2)}")
    A.e(f"s_addc_u32 {t(4)}, {s('pX_hi')}, {t(3)}")
    A.e(f"s_and_b32 s{S['rX'] + 1}, {t(4)}, 0xffff")
    A.e(f"s_add_u32 s{S['rL']}, {s('pL_lo')}, {t(2)}")
    A.e(f"s_addc_u32 {t(4)}, {s('pL_hi')}, {t(3)}")
    A.e(f"s_and_b32 s{S['rL'] + 1}, {t(4)}, 0xffff")
    A.e(f"s_sub_i32 {t(4)}, {s('T')}, {t(0)}")
    A.e(f"s_min_i32 {t(4)}, {t(4)}, 2")
    A.e(f"s_mul_i32 {t(4)}, {t(4)}, {s('K')}")
    A.e(f"s_lshl_b32 {t(4)}, {t(4)}, 2")
    A.e(f"s_mov_b32 s{S['rX'] + 2}, {t(4)}")
    A.e(f"s_mov_b32 s{S['rL'] + 2}, {t(4)}")


def set_a_rows(A):
    """weight DMA voffsets of the A cursor's row tile: min(rt*ROWS + (w + 4q)*16 + lrow, M-1) * Kb + lslot16"""
    A.e(f"v_lshrrev_b32 v60, 2, {v('LANE')}")                      # lrow
    A.e(f"v_and_b32 v61, 3, {v('LANE')}")
    A.e(f"v_lshrrev_b32 v62, 4, {v('LANE')}")
    A.e("v_and_b32 v62, 3, v62")
    A.e("v_xor_b32 v61, v61, v62")
    A.e("v_lshlrev_b32 v61, 4, v61")                               # lslot16
    A.e(f"s_mul_i32 {t(0)}, {s('a_rt')}, {ROWS}")
    A.e(f"s_lshl_b32 {t(1)}, {s('w')}, 4")
    A.e(f"s_add_i32 {t(0)}, {t(0)}, {t(1)}")
    A.e(f"v_add_u32 v60, {t(0)}, v60")                             # row of q = 0
    A.e(f"s_sub_i32 {t(2)}, {s('M')}, 1")
    for q in range(NRB // 2):
        A.e(f"v_add_u32 v62, {64 * q}, v60")
        A.e(f"v_min_i32 v62, {t(2)}, v62")
        A.e(f"v_mul_lo_u32 v62, v62, {s('Kb')}")
        A.e(f"v_add_u32 v{V['DMA'] + q}, v62, v61")


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


def issue_x(A, slot_sgpr_expr_setup):
    """x and log2 runs of the L cursor's step into x-ring slot (SGPR t(6) holds the slot's byte base for this wave)"""
    slot_sgpr_expr_setup(A)
    A.e(f"s_lshl_b32 {t(7)}, {s('l_k')}, 7")                       # k * 128 bytes
    A.e(f"s_mov_b32 m0, {t(6)}")
    A.e(f"buffer_load_dword {v('XDMA')}, s[{S['rX']}:{S['rX'] + 3}], {t(7)} offen lds")
    A.e(f"s_add_i32 m0, {t(6)}, 256")
    A.e(f"buffer_load_dword {v('XDMA')}, s[{S['rL']}:{S['rL'] + 3}], {t(7)} offen lds")
    cursor_step(A, "l", set_x_rsrc)


def issue_a(A, slot_setup):
    """weight tile of the A cursor's step: NRB/2 requests of 16 rows x 64 B by this wave (t(6) = slot base + w*1024)"""
    slot_setup(A)
    A.e(f"s_lshl_b32 {t(7)}, {s('a_k')}, 6")                       # k * 64 bytes
    for q in range(NRB // 2):
        if q == 0:
            A.e(f"s_mov_b32 m0, {t(6)}")
        else:
            A.e(f"s_add_i32 m0, {t(6)}, {q * 4096}")
        if os.environ.get("FUSED_NODMA") and q > 0:
            A.e(f"buffer_load_dword {v('XDMA')}, s[{S['rX']}:{S['rX'] + 3}], {t(7)} offen lds")   # keeps the vmcnt bookkeeping, moves 256 B
            continue
        A.e(f"buffer_load_dwordx4 v{V['DMA'] + q}, s[{S['rW']}:{S['rW'] + 3}], {t(7)} offen lds")
    cursor_step(A, "a", set_a_rows)


def x_slot_base(slot_reg_or_const):
    def f(A):
        # t(6) = oXr + slot*2048 + w*512
        if isinstance(slot_reg_or_const, int):
            A.e(f"s_lshl_b32 {t(6)}, {s('w')}, 9")
            A.e(f"s_add_i32 {t(6)}, {t(6)}, {slot_reg_or_const * 2048}")
        else:
            A.e(f"s_lshl_b32 {t(6)}, {slot_reg_or_const}, 11")
            A.e(f"s_lshl_b32 {t(8)}, {s('w')}, 9")
            A.e(f"s_add_i32 {t(6)}, {t(6)}, {t(8)}")
        A.e(f"s_add_i32 {t(6)}, {t(6)}, {s('oXr')}")
    return f


def a_slot_base(slot_reg_or_const):
    def f(A):
        if isinstance(slot_reg_or_const, int):
            A.e(f"s_lshl_b32 {t(6)}, {s('w')}, 10")
            A.e(f"s_add_i32 {t(6)}, {t(6)}, {slot_reg_or_const * AT}")
        else:
            A.e(f"s_mul_i32 {t(6)}, {slot_reg_or_const}, {AT}")
            A.e(f"s_lshl_b32 {t(8)}, {s('w')}, 10")
            A.e(f"s_add_i32 {t(6)}, {t(6)}, {t(8)}")
        A.e(f"s_add_i32 {t(6)}, {t(6)}, {s('oRing')}")
    return f


# ---------------------------------------------------------------- fragment generation (one pair of element-candidates)
def gen_pair_ops(cb, e, set_, val0):
    """instruction list (strings or ('ds', text, tag)) producing LUT reads of elements e, e+1 of candidate block cb
    from the log2 values LV[e], LV[e+1]; results land in VAL+val0, VAL+val0+1.  (Near-ties: k_tie_flags' bits.)"""
    par = "PAR0" if cb == 0 else "PAR1"
    ca, cc, chi = v(par, CA), v(par, CC), v(par, CHI)
    lutc = v("LUTC0" if cb == 0 else "LUTC1")
    g = V["GT"] + 4 * set_
    k0, k1, a0, a1 = (f"v{g + i}" for i in range(4))
    lut0 = ("ds", f"ds_read_b32 v{V['VAL'] + val0}, {a0}", f"val{val0}") if not os.environ.get("FUSED_NOLUT") else f"v_mov_b32 v{V['VAL'] + val0}, {a0}"
    lut1 = ("ds", f"ds_read_b32 v{V['VAL'] + val0 + 1}, {a1}", f"val{val0 + 1}") if not os.environ.get("FUSED_NOLUT") else f"v_mov_b32 v{V['VAL'] + val0 + 1}, {a1}"
    return [
        f"v_fma_f32 {k0}, {v('LV', e)}, {ca}, {cc}",
        f"v_fma_f32 {k1}, {v('LV', e + 1)}, {ca}, {cc}",
        f"v_med3_f32 {k0}, {k0}, 0, {chi}",
        f"v_med3_f32 {k1}, {k1}, 0, {chi}",
        f"v_add_f32 {k0}, 0x{MAGIC:08x}, {k0}",
        f"v_add_f32 {k1}, 0x{MAGIC:08x}, {k1}",
        f"v_lshl_add_u32 {a0}, {k0}, 9, {lutc}",
        f"v_lshl_add_u32 {a1}, {k1}, 9, {lutc}",
        lut0, lut1,
    ]


def pack_op(bn, cb, i, val0):
    """dword i of B fragment (cb) from VAL[val0], VAL[val0+1]"""
    return ("pack", f"v_lshl_or_b32 v{V[bn] + 4 * cb + i}, v{V['VAL'] + val0 + 1}, 16, v{V['VAL'] + val0}", (f"val{val0}", f"val{val0 + 1}"))


def fix_chunk(A, cb, bn, xsrc_vgpr, eo):
    """cold block: a chunk (cb) holding a near-tie.  Per element: is any lane within the zone?  (~1.3 % each) -- if so the
    element's bins come from the threshold table for every lane (exact wherever the fast bin is within one of it) and the
    16-bit value is patched into B fragment bn[cb].  LV holds the chunk's log2 values; x is read from the x run."""
    par = "PAR0" if cb == 0 else "PAR1"
    ca, cc, chi = v(par, CA), v(par, CC), v(par, CHI)
    thrc = v("THRC0" if cb == 0 else "THRC1")
    lutb = v("LUTB0" if cb == 0 else "LUTB1")
    A.drain()
    A.ds(f"ds_read_b128 {vr('XV', 0, 4)}, {xsrc_vgpr} offset:{eo * 4}", "x0")
    A.ds(f"ds_read_b128 {vr('XV', 4, 4)}, {xsrc_vgpr} offset:{eo * 4 + 16}", "x1")
    A.e(f"s_sub_i32 {t(10)}, {s('L2')}, 1")
    g = V["GT"]
    for e in range(8):
        kf, tt, dd, f_, xs, i0, i1, tu = (f"v{g + i}" for i in range(8))
        td, u_, d_ = (f"v{g + 8 + i}" for i in range(3))
        skip = A.new("Lfe")
        A.e(f"v_fma_f32 {kf}, {v('LV', e)}, {ca}, {cc}")
        A.e(f"v_med3_f32 {kf}, {kf}, 0, {chi}")
        A.e(f"v_add_f32 {tt}, 0x{MAGIC:08x}, {kf}")
        A.e(f"v_add_f32 {dd}, 0x{(MAGIC ^ 0x80000000):08x}, {tt}")
        A.e(f"v_sub_f32 {dd}, {kf}, {dd}")
        A.e(f"v_cmp_lt_f32 vcc, {s('tie')}, abs({dd})")
        A.e(f"s_cbranch_vccz {skip}_%=")
        saved = list(A.fifo)
        A.e(f"v_and_b32 {f_}, 0xff, {tt}")
        A.e(f"v_min_u32 {f_}, {s('L2')}, {f_}")                    # fast bin, clamped to 2^bits (= masked)
        A.e(f"v_min_u32 {i0}, {t(10)}, {f_}")
        A.e(f"v_lshl_add_u32 {i0}, {i0}, 9, {thrc}")
        A.e(f"v_subrev_u32 {i1}, 1, {f_}")
        A.e(f"v_max_i32 {i1}, 0, {i1}")
        A.e(f"v_lshl_add_u32 {i1}, {i1}, 9, {thrc}")
        A.ds(f"ds_read_b32 {tu}, {i0}", "tu")
        A.ds(f"ds_read_b32 {td}, {i1}", "td")
        A.drain()                                                  # (also covers the x reads)
        A.e(f"v_add_f32 {xs}, {s('shift')}, {v('XV', e)}")
        A.e(f"v_cmp_lt_f32 {t2(12)}, {xs}, {tu}")                  # xs < thr[f]
        A.e(f"v_cmp_lt_u32 {t2(14)}, {f_}, {s('L2')}")             # f < 2^bits
        A.e(f"s_and_b64 {t2(12)}, {t2(12)}, {t2(14)}")
        A.e(f"v_cndmask_b32 {u_}, 0, 1, {t2(12)}")
        A.e(f"v_cmp_nlt_f32 {t2(12)}, {xs}, {td}")                 # !(xs < thr[f-1])
        A.e(f"v_cmp_lt_u32 {t2(14)}, 0, {f_}")                     # f > 0
        A.e(f"s_and_b64 {t2(12)}, {t2(12)}, {t2(14)}")
        A.e(f"v_cndmask_b32 {d_}, 0, 1, {t2(12)}")
        A.e(f"v_add_u32 {f_}, {f_}, {u_}")
        A.e(f"v_sub_u32 {f_}, {f_}, {d_}")
        A.e(f"v_lshl_add_u32 {f_}, {f_}, 9, {lutb}")
        A.ds(f"ds_read_b32 {tu}, {f_}", "fx")
        A.drain()
        bdw = f"v{V[bn] + 4 * cb + e // 2}"
        if e % 2 == 0:
            A.e(f"v_and_b32 {bdw}, 0xffff0000, {bdw}")
            A.e(f"v_or_b32 {bdw}, {bdw}, {tu}")
        else:
            A.e(f"v_and_b32 {bdw}, 0xffff, {bdw}")
            A.e(f"v_lshl_or_b32 {bdw}, {tu}, 16, {bdw}")
        A.label(skip)
        A.fifo = saved if e == 0 and False else A.fifo             # after a taken element nothing is outstanding; if every
        # element is skipped the x reads are still in flight: harmless (XV is only read inside taken elements, behind a drain)
    A.drain()


PROF = bool(os.environ.get("FUSED_PROF"))          # phase timers (s_memtime) instead of results: see tools/lab/prof_fused.py
PV = dict(P0=118, P1=119, P2=35, P3=3)              # cycles: barrier wait / unit 0 / unit 1 / everything between steps


def stamp(A, which):
    """profiling build: add the cycles since the previous stamp to counter `which`"""
    if not PROF:
        return
    A.drain()
    A.e(f"s_memtime s[{T0 + 14}:{T0 + 15}]")
    A.e("s_waitcnt lgkmcnt(0)")
    A.e(f"s_sub_i32 {t(13)}, {t(14)}, {t(18)}")
    A.e(f"s_mov_b32 {t(18)}, {t(14)}")
    A.e(f"v_add_u32 v{PV[which]}, {t(13)}, v{PV[which]}")


def load_flag_row(A, pair_sgpr):
    """FLGN <- the near-tie flag row of token 2 * pair + (w >> 1) (dword j in lane j; zeros past T).  Issues one VMEM load."""
    lab = A.new("Lnofl")
    A.e(f"v_mov_b32 {v('FLGN')}, 0")
    A.e(f"s_lshl_b32 {t(0)}, {pair_sgpr}, 1")
    A.e(f"s_lshr_b32 {t(1)}, {s('w')}, 1")
    A.e(f"s_add_i32 {t(0)}, {t(0)}, {t(1)}")                        # token
    A.e(f"s_cmp_ge_i32 {t(0)}, {s('T')}")
    A.e(f"s_cbranch_scc1 {lab}_%=")
    A.e(f"s_mul_hi_u32 {t(3)}, {t(0)}, {s('fpitch')}")
    A.e(f"s_mul_i32 {t(2)}, {t(0)}, {s('fpitch')}")
    A.e(f"s_add_u32 {t(2)}, {t(2)}, {s('pF_lo')}")
    A.e(f"s_addc_u32 {t(3)}, {t(3)}, {s('pF_hi')}")
    A.e(f"s_lshr_b32 {t(4)}, {s('fpitch')}, 2")                     # dwords per row
    A.e(f"v_lshlrev_b32 v60, 2, {v('LANE')}")
    A.e(f"v_cmp_gt_u32 vcc, {t(4)}, {v('LANE')}")
    A.e(f"s_and_saveexec_b64 {t2(16)}, vcc")
    A.e(f"global_load_dword {v('FLGN')}, v60, {t2(2)}")
    A.e(f"s_mov_b64 exec, {t2(16)}")
    A.label(lab)


def first_nibble(A, reg, dst):
    """dst = nib(0) of the row in VGPR `reg` (byte nk), shifted right by 2 * (w & 1): bit cb = candidate block cb of this wave"""
    A.e(f"s_lshr_b32 {t(0)}, {s('nk')}, 2")
    A.e(f"v_readlane_b32 {dst}, {v(reg)}, {t(0)}")
    A.e(f"s_and_b32 {t(1)}, {s('nk')}, 3")
    A.e(f"s_lshl_b32 {t(1)}, {t(1)}, 3")
    A.e(f"s_lshr_b32 {dst}, {dst}, {t(1)}")
    A.e(f"s_and_b32 {dst}, {dst}, 15")
    A.e(f"s_lshr_b32 {dst}, {dst}, {s('sh2')}")


def rotate_flags(A):
    """after the next tile's row arrived in FLGN (and the current one sits in FLGC): put the next tile's nib(0) into the
    high nibble of the current row's last byte (the unit that runs ahead into the next tile tests it)"""
    A.e(f"s_lshr_b32 {t(0)}, {s('nk')}, 2")
    A.e(f"v_readlane_b32 {t(1)}, {v('FLGN')}, {t(0)}")
    A.e(f"s_and_b32 {t(2)}, {s('nk')}, 3")
    A.e(f"s_lshl_b32 {t(2)}, {t(2)}, 3")
    A.e(f"s_lshr_b32 {t(1)}, {t(1)}, {t(2)}")
    A.e(f"s_and_b32 {t(1)}, {t(1)}, 15")                            # nib(0) of the next tile's token
    A.e(f"s_sub_i32 {t(3)}, {s('nk')}, 1")
    A.e(f"s_and_b32 {t(4)}, {t(3)}, 3")
    A.e(f"s_lshl_b32 {t(4)}, {t(4)}, 3")
    A.e(f"s_add_i32 {t(4)}, {t(4)}, 4")
    A.e(f"s_lshl_b32 {t(1)}, {t(1)}, {t(4)}")
    A.e(f"s_lshr_b32 {t(3)}, {t(3)}, 2")
    A.e(f"v_mov_b32 v60, {t(1)}")
    A.e(f"v_cmp_eq_u32 vcc, {t(3)}, {v('LANE')}")
    A.e("v_cndmask_b32 v60, 0, v60, vcc")
    A.e(f"v_or_b32 {v('FLGC')}, {v('FLGC')}, v60")


def step_flags(A):
    """t(20) = flag byte of this K-step (k_tie_flags layout) >> 2 * (w & 1): bits cb = unit 0, bits 4 + cb = unit 1"""
    A.e(f"s_sub_i32 {t(12)}, {s('nk')}, {s('kt')}")                # step index
    A.e(f"s_lshr_b32 {t(13)}, {t(12)}, 2")
    A.e(f"v_readlane_b32 {t(20)}, {v('FLGC')}, {t(13)}")
    A.e(f"s_and_b32 {t(12)}, {t(12)}, 3")
    A.e(f"s_lshl_b32 {t(12)}, {t(12)}, 3")
    A.e(f"s_lshr_b32 {t(20)}, {t(20)}, {t(12)}")
    A.e(f"s_lshr_b32 {t(20)}, {t(20)}, {s('sh2')}")


def gen_only(A, bn, xs_vgpr, eo):
    """un-overlapped generation of one K half into bn (used once, before the first step)"""
    A.ds(f"ds_read_b128 {vr('LV', 0, 4)}, {xs_vgpr} offset:{256 + eo * 4}", "l0")
    A.ds(f"ds_read_b128 {vr('LV', 4, 4)}, {xs_vgpr} offset:{256 + eo * 4 + 16}", "l1")
    A.drain()
    for cb in range(2):
        for pi in range(4):
            for op in gen_pair_ops(cb, 2 * pi, pi & 3, 8 * cb + 2 * pi):
                if isinstance(op, tuple):
                    A.ds(op[1], op[2])
                else:
                    A.e(op)
        A.drain()
        for i in range(4):
            A.e(pack_op(bn, cb, i, 8 * cb + 2 * i)[1])
    first_nibble(A, "FLGN", t(20))                                # nib(0) of the first tile's token, >> 2 * (w & 1)
    for cb in range(2):
        lab = A.new("Lfix")
        A.e(f"s_bitcmp0_b32 {t(20)}, {cb}")
        A.e(f"s_cbranch_scc1 {lab}_%=")
        fix_chunk(A, cb, bn, xs_vgpr, eo)
        A.label(lab)


def unit(A, h, bc, bn, xs_vgpr, eo, cold_blocks):
    """half a K-step: 2*NRB MFMAs of K half h from B fragments bc, interleaved with the generation of the next half's
    fragments (bn) from the x / log2 run at xs_vgpr (+eo elements).  Appends (label, emitter) cold blocks."""
    aoff = v("AS0" if h == 0 else "AS1")
    # ---- the filler stream, in issue order
    fill = []
    fill.append(("ds", f"ds_read_b128 {vr('LV', 0, 4)}, {xs_vgpr} offset:{256 + eo * 4}", "l0"))
    fill.append(("ds", f"ds_read_b128 {vr('LV', 4, 4)}, {xs_vgpr} offset:{256 + eo * 4 + 16}", "l1"))
    pend_packs = []
    for cb in range(2):
        for pi in range(4):
            ops = gen_pair_ops(cb, 2 * pi, pi & 3, 8 * cb + 2 * pi)
            if cb == 0 and pi == 0:
                ops.insert(0, ("waitfor", "l1"))
            fill.extend(ops)
            # pack the pair generated two pairs ago (its LUT reads have had time to return)
            pend_packs.append(pack_op(bn, cb, pi, 8 * cb + 2 * pi))
            if len(pend_packs) > 3:
                fill.append(pend_packs.pop(0))
    fill.extend(pend_packs)
    # ---- MFMA stream with the A fragment reads three row blocks ahead (four rotating buffers)
    nf = len(fill)
    n_mfma = 2 * NRB
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

    def abuf(rb):
        b = V['ABUF'] + 4 * (rb % 4)
        return f"v[{b}:{b + 3}]"

    def a_read(rb):
        A.ds(f"ds_read_b128 {abuf(rb)}, {aoff} offset:{rb * 2048}", f"a{rb}")

    emit_fill(2)                                                   # the two log2 reads go first
    a_read(0)
    a_read(1)
    a_read(2)
    emit_fill(14)                                                  # first pair's arithmetic covers the fragment latency
    per = (nf - fi + n_mfma - 1) // n_mfma
    for rb in range(NRB):
        if rb + 3 < NRB:
            a_read(rb + 3)
        A.wait(f"a{rb}")
        for cb in range(2):
            if not os.environ.get("FUSED_NOMFMA"):
                A.e(f"v_mfma_f32_32x32x16_bf16 {acc(rb, cb)}, {abuf(rb)}, v[{V[bc] + 4 * cb}:{V[bc] + 4 * cb + 3}], {acc(rb, cb)}")
            emit_fill(per)
    emit_fill(nf)
    # ---- near-tie checks of the two chunks (cold blocks follow the loop)
    for cb in range(2):
        if os.environ.get("FUSED_NOCOLD"):
            continue
        lab = A.new("Lcold")
        A.e(f"s_bitcmp1_b32 {t(20)}, {4 * h + cb}")                 # k_tie_flags: unit h of this step, candidate block cb
        A.e(f"s_cbranch_scc1 {lab}_%=")
        A.label(lab + "r")
        cold_blocks.append((lab, cb, bn, xs_vgpr, eo, list(A.fifo)))


def epilogue(A):
    A.c("epilogue: e = refb - (D * alpha) * rs ; s += e^2")
    A.e("s_nop 7")
    A.e("s_nop 7")
    A.e("s_nop 7")                                                 # MFMA results -> VALU reads
    s0, s1 = "v96", "v97"
    A.e(f"v_mov_b32 {s0}, 0")
    A.e(f"v_mov_b32 {s1}, 0")
    idx = 0
    reads = []
    for rb in range(NRB):
        for i4 in range(4):
            reads.append((rb, i4))
    # software pipeline: issue the reads of group g+1 before the arithmetic of group g
    def issue(gi):
        rb, i4 = reads[gi]
        b = V["E0"] + 8 * (gi & 1)
        off = (rb * 32 + 8 * i4) * 4
        A.ds(f"ds_read_b128 v[{b}:{b + 3}], {v('EPI')} offset:{off}", f"rf{gi}")
        A.ds(f"ds_read_b128 v[{b + 4}:{b + 7}], {v('EPR')} offset:{off}", f"rs{gi}")
    issue(0)
    for gi, (rb, i4) in enumerate(reads):
        if gi + 1 < len(reads):
            issue(gi + 1)
        A.wait(f"rs{gi}")
        b = V["E0"] + 8 * (gi & 1)
        for j in range(4):
            for cb in range(2):
                kind, r = acc_elem(rb, cb, 4 * i4 + j)
                tmp = f"v{80 + 2 * j + cb}"
                if kind == "a":
                    A.e(f"v_accvgpr_read_b32 {tmp}, a{r}")
                    src = tmp
                else:
                    src = f"v{r}"
                A.e(f"v_mul_f32 {tmp}, {src}, {v('PAR0' if cb == 0 else 'PAR1', AL)}")
                A.e(f"v_fma_f32 {tmp}, -{tmp}, v{b + 4 + j}, v{b + j}")
                A.e(f"v_fma_f32 {s0 if cb == 0 else s1}, {tmp}, {tmp}, {s0 if cb == 0 else s1}")
    # lanes l and l + 32 hold the two row halves of a candidate
    A.ds(f"ds_bpermute_b32 v98, {v('XORA')}, {s0}", "bp0")
    A.ds(f"ds_bpermute_b32 v99, {v('XORA')}, {s1}", "bp1")
    A.drain()
    A.e(f"v_add_f32 {s0}, {s0}, v98")
    A.e(f"v_add_f32 {s1}, {s1}, v99")
    # run += (double) s  when this wave's token exists (t(9) = 1)
    lab = A.new("Lnotok")
    A.e(f"s_cmp_eq_u32 {t(9)}, 0")
    A.e(f"s_cbranch_scc1 {lab}_%=")
    A.e(f"v_cvt_f64_f32 v[100:101], {s0}")
    A.e(f"v_cvt_f64_f32 v[102:103], {s1}")
    A.e(f"v_add_f64 v[{V['RUN0']}:{V['RUN0'] + 1}], v[{V['RUN0']}:{V['RUN0'] + 1}], v[100:101]")
    A.e(f"v_add_f64 v[{V['RUN1']}:{V['RUN1'] + 1}], v[{V['RUN1']}:{V['RUN1'] + 1}], v[102:103]")
    A.label(lab)


def tile_setup(A):
    """epilogue operands of the compute tile -> LDS; accumulators zeroed.  Uses the compute cursor (c_tile)."""
    A.c("tile: near-tie flag rows (this tile's = the one fetched during the previous tile; fetch the next tile's)")
    A.e(f"v_mov_b32 {v('FLGC')}, {v('FLGN')}")
    A.e(f"s_add_i32 {t(5)}, {s('c_pair')}, {s('dpair')}")
    A.e(f"s_add_i32 {t(6)}, {s('c_rt')}, {s('drt')}")
    A.e(f"s_cmp_ge_u32 {t(6)}, {s('n_rt')}")
    A.e(f"s_cselect_b32 {t(6)}, 1, 0")
    A.e(f"s_add_i32 {t(5)}, {t(5)}, {t(6)}")                        # token pair of the next tile of this workgroup
    load_flag_row(A, t(5))
    A.c("tile: m0, tok0, staging of ref - row_bias and row_scale, zero accumulators")
    # (pair, rt) of the compute tile are kept incrementally (c_pair, c_rt), like the issue cursors
    A.e(f"s_mov_b32 {t(0)}, {s('c_pair')}")
    A.e(f"s_mov_b32 {t(1)}, {s('c_rt')}")
    A.e(f"s_lshl_b32 {t(2)}, {t(0)}, 1")                           # tok0
    A.e(f"s_mul_i32 {t(3)}, {t(1)}, {ROWS}")                       # m0
    # tok_ok for this wave: tok0 + wtok < T
    A.e(f"s_lshr_b32 {t(4)}, {s('w')}, 1")
    A.e(f"s_add_i32 {t(4)}, {t(4)}, {t(2)}")
    A.e(f"s_cmp_lt_i32 {t(4)}, {s('T')}")
    A.e(f"s_cselect_b32 {t(9)}, 1, 0")
    # thread id within the workgroup: tid = w*64 + lane ; staging elements e = tid + u*256
    A.e(f"s_lshl_b32 {t(5)}, {s('w')}, 6")
    A.e(f"v_add_u32 v60, {t(5)}, {v('LANE')}")                     # tid
    EU = (2 * ROWS + 255) // 256
    RU = (ROWS + 255) // 256
    A.e("s_waitcnt vmcnt(0)")
    A.e(f"s_mul_i32 {t(20)}, {t(2)}, {s('M')}")                    # tok0 * M (64 bit) * 4 + pRef
    A.e(f"s_mul_hi_u32 {t(21)}, {t(2)}, {s('M')}")
    A.e(f"s_lshl_b64 {t2(20)}, {t2(20)}, 2")
    A.e(f"s_add_u32 {t(20)}, {t(20)}, {s('pRef_lo')}")
    A.e(f"s_addc_u32 {t(21)}, {t(21)}, {s('pRef_hi')}")
    A.e(f"s_movk_i32 {t(7)}, {2 * ROWS}")
    A.e(f"s_movk_i32 {t(8)}, {ROWS}")
    for u in range(EU):
        # e = tid + u*256 ; ts = e >= ROWS ; r = e - ts*ROWS ; row = m0 + r ; tk = tok0 + ts
        A.e(f"v_add_u32 v61, {u * 256}, v60")
        A.e(f"v_cmp_le_u32 vcc, {ROWS}, v61")
        A.e("v_cndmask_b32 v62, 0, 1, vcc")                        # ts
        A.e(f"v_mul_u32_u24 v63, {ROWS}, v62")
        A.e("v_sub_u32 v63, v61, v63")                             # r
        A.e(f"v_add_u32 v64, {t(3)}, v63")                         # row
        A.e(f"v_add_u32 v65, {t(2)}, v62")                         # tk
        # ok = e < 2*ROWS && row < M && tk < T
        A.e(f"v_cmp_gt_u32 {t2(12)}, {t(7)}, v61")
        A.e(f"v_cmp_gt_i32 {t2(14)}, {s('M')}, v64")
        A.e(f"s_and_b64 {t2(12)}, {t2(12)}, {t2(14)}")
        A.e(f"v_cmp_gt_i32 {t2(14)}, {s('T')}, v65")
        A.e(f"s_and_b64 {t2(12)}, {t2(12)}, {t2(14)}")
        # ref[(tok0 + ts)*M + row]: 64-bit tile base in s[t20:t21], 32-bit lane offset (ts*M + row)*4
        A.e(f"v_mul_lo_u32 v66, v62, {s('M')}")
        A.e("v_add_u32 v66, v66, v64")
        A.e("v_lshlrev_b32 v66, 2, v66")
        A.e(f"v_mov_b32 v{70 + u}, 0")
        A.e(f"v_mov_b32 v{74 + u}, 0")
        A.e(f"s_and_saveexec_b64 {t2(16)}, {t2(12)}")
        if SUB >= 1:
            var = int(os.environ.get("FUSED_VAR", "0"))
            if var == 0:
                A.e(f"global_load_dword v{70 + u}, v66, {t2(20)}")
            elif var == 1:                                              # constant offset 0
                A.e(f"global_load_dword v{70 + u}, {v('ZERO')}, {t2(20)}")
            elif var == 2:                                              # plain 64-bit address of ref[0]
                A.e(f"v_mov_b32 v66, {s('pRef_lo')}")
                A.e(f"v_mov_b32 v67, {s('pRef_hi')}")
                A.e(f"global_load_dword v{70 + u}, v[66:67], off")
            elif var == 3:                                              # no exec masking around it
                A.e(f"s_mov_b64 exec, {t2(16)}")
                A.e(f"global_load_dword v{70 + u}, {v('ZERO')}, {t2(20)}")
            elif var == 5:                                              # s[28:29] = plain copy of pRef
                A.e(f"s_mov_b32 {t(20)}, {s('pRef_lo')}")
                A.e(f"s_mov_b32 {t(21)}, {s('pRef_hi')}")
                A.e(f"global_load_dword v{70 + u}, {v('ZERO')}, {t2(20)}")
            elif var == 6:                                              # same through another pair
                A.e(f"global_load_dword v{70 + u}, {v('ZERO')}, {t2(20)}")
            elif var == 7:                                              # pRef pair directly
                A.e(f"global_load_dword v{70 + u}, {v('ZERO')}, s[{S['pRef_lo']}:{S['pRef_hi']}]")
            elif var == 8:                                              # computed base, long settle time
                A.e("s_nop 7")
                A.e("s_nop 7")
                A.e(f"global_load_dword v{70 + u}, {v('ZERO')}, {t2(20)}")
            elif var == 10:                                             # pRef + constant inside the tensor
                A.e(f"s_add_u32 s28, {s('pRef_lo')}, 0x60000")
                A.e(f"s_addc_u32 s29, {s('pRef_hi')}, 0")
                A.e(f"global_load_dword v{70 + u}, {v('ZERO')}, s[28:29]")
            elif var == 11:                                             # computed base but only wave 0 / lane 0 loads
                A.e(f"s_mov_b64 exec, 1")
                A.e(f"global_load_dword v{70 + u}, {v('ZERO')}, {t2(20)}")
            elif var == 4:                                              # read W instead of ref
                A.e(f"global_load_dword v{70 + u}, {v('ZERO')}, s[{S['pW_lo']}:{S['pW_hi']}]")
        # row_bias (may be null)
        labn = A.new("Lnorb")
        A.e(f"s_or_b32 {t(18)}, {s('pRb_lo')}, {s('pRb_hi')}")
        A.e(f"s_cmp_eq_u32 {t(18)}, 0")
        A.e(f"s_cbranch_scc1 {labn}_%=")
        A.e("v_lshlrev_b32 v69, 2, v64")
        if SUB >= 2:
            A.e(f"global_load_dword v{74 + u}, v69, s[{S['pRb_lo']}:{S['pRb_hi']}]")
        A.label(labn)
        A.e(f"s_mov_b64 exec, {t2(16)}")
    for u in range(RU):
        A.e(f"v_add_u32 v61, {u * 256}, v60")                      # r
        A.e(f"v_add_u32 v64, {t(3)}, v61")                         # row
        A.e(f"v_cmp_gt_u32 {t2(12)}, {t(8)}, v61")
        A.e(f"v_cmp_gt_i32 {t2(14)}, {s('M')}, v64")
        A.e(f"s_and_b64 {t2(12)}, {t2(12)}, {t2(14)}")
        A.e(f"v_mov_b32 v{78 + u}, 0")
        A.e("v_lshlrev_b32 v69, 2, v64")
        A.e(f"s_and_saveexec_b64 {t2(16)}, {t2(12)}")
        if SUB >= 3:
            A.e(f"global_load_dword v{78 + u}, v69, s[{S['pRs_lo']}:{S['pRs_hi']}]")
        A.e(f"s_mov_b64 exec, {t2(16)}")
    A.e("s_waitcnt vmcnt(0)")
    A.e("s_barrier")                                               # every wave is past the previous tile's epilogue reads
    for u in range(EU):
        A.e(f"v_add_u32 v61, {u * 256}, v60")
        A.e(f"v_cmp_gt_u32 vcc, {2 * ROWS}, v61")
        A.e(f"v_sub_f32 v{70 + u}, v{70 + u}, v{74 + u}")
        A.e(f"v_lshl_add_u32 v62, v61, 2, {s('oRefb')}")
        A.e(f"s_and_saveexec_b64 {t2(16)}, vcc")
        A.e(f"ds_write_b32 v62, v{70 + u}")
        A.e(f"s_mov_b64 exec, {t2(16)}")
    for u in range(RU):
        A.e(f"v_add_u32 v61, {u * 256}, v60")
        A.e(f"v_cmp_gt_u32 vcc, {ROWS}, v61")
        A.e(f"v_lshl_add_u32 v62, v61, 2, {s('oRs')}")
        A.e(f"s_and_saveexec_b64 {t2(16)}, vcc")
        A.e(f"ds_write_b32 v62, v{78 + u}")
        A.e(f"s_mov_b64 exec, {t2(16)}")
    A.c("zero the accumulators")
    for r in range(16 * 2 * NA):
        A.e(f"v_accvgpr_write_b32 a{r}, 0")
    for r in range(128, 128 + 16 * 2 * (NRB - NA)):
        A.e(f"v_mov_b32 v{r}, 0")
    A.e("s_nop 4")
    rotate_flags(A)


SUB = int(os.environ.get("FUSED_SUB", "99"))            # debugging of tile_setup: 0 no loads, 1 ref, 2 + row_bias, 3 + row_scale
STAGE = int(os.environ.get("FUSED_STAGE", "99"))      # debugging: stop after a phase (1 setup, 2 prologue, 3 tile setup, 4 one step)


def program():
    A = Asm()
    cfg_load(A)
    lane_setup(A)
    if STAGE <= 1:
        A.e("s_branch Lend_%=")
    # cursors
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
    load_flag_row(A, s('pair0'))                                    # first tile's near-tie flags (waited with the first DMAs)
    A.c("prologue: x/log2 of step 0, then weights of steps 0..FNS-2 with the x/log2 of the following step")
    issue_x(A, x_slot_base(0))
    for s0 in range(FNS - 1):
        issue_a(A, a_slot_base(s0))
        issue_x(A, x_slot_base(s0 + 1))
    A.e(f"s_mov_b32 {s('stA')}, 0")
    A.e(f"s_mov_b32 {s('stX')}, 0")
    A.c("B fragments of the very first K half")
    A.e("s_waitcnt vmcnt(0)")
    A.e(f"v_mov_b32 {v('XSC')}, {v('XOFF')}")
    gen_only(A, "BA", v("XSC"), 0)
    if STAGE <= 2:
        A.e("s_branch Lend_%=")

    if PROF:
        for r in PV.values():
            A.e(f"v_mov_b32 v{r}, 0")
        A.e(f"s_memtime s[{T0 + 14}:{T0 + 15}]")
        A.e("s_waitcnt lgkmcnt(0)")
        A.e(f"s_mov_b32 {t(18)}, {t(14)}")
    A.label("Ltile")
    tile_setup(A)
    if STAGE <= 3:
        A.e("s_branch Lend_%=")
    A.e(f"s_mov_b32 {s('kt')}, {s('nk')}")
    A.label("Lstep")
    stamp(A, "P3")
    cold = []
    A.e(f"s_waitcnt vmcnt({(FNS - 2) * (NRB // 2 + 2)})")
    if not os.environ.get("FUSED_NOBAR"):
        A.e("s_barrier")
    stamp(A, "P0")
    # ring addresses of this step
    A.e(f"s_mul_i32 {t(0)}, {s('stA')}, {AT}")
    A.e(f"s_add_i32 {t(0)}, {t(0)}, {s('oRing')}")
    A.e(f"v_add_u32 {v('AS0')}, {t(0)}, {v('AOFF0')}")
    A.e(f"v_add_u32 {v('AS1')}, {t(0)}, {v('AOFF1')}")
    A.e(f"s_lshl_b32 {t(1)}, {s('stX')}, 11")
    A.e(f"v_add_u32 {v('XSC')}, {t(1)}, {v('XOFF')}")
    A.e(f"s_add_i32 {t(2)}, {s('stX')}, 1")
    A.e(f"s_cmp_eq_u32 {t(2)}, {XS}")
    A.e(f"s_cselect_b32 {t(2)}, 0, {t(2)}")                        # next x slot
    A.e(f"s_lshl_b32 {t(3)}, {t(2)}, 11")
    A.e(f"v_add_u32 {v('XSN')}, {t(3)}, {v('XOFF')}")
    A.e(f"s_mov_b32 {t(19)}, {t(2)}")                              # keep the next x slot
    step_flags(A)
    # DMA for step n + FNS - 1 (weights -> the slot step n - 1 used) and n + FNS (x/log2 -> the x slot step n - 1 used)
    A.e(f"s_add_i32 {t(4)}, {s('stA')}, {FNS - 1}")
    A.e(f"s_cmp_ge_u32 {t(4)}, {FNS}")
    A.e(f"s_cselect_b32 {t(5)}, {FNS}, 0")
    A.e(f"s_sub_i32 {t(4)}, {t(4)}, {t(5)}")
    issue_a(A, a_slot_base(t(4)))
    A.e(f"s_add_i32 {t(4)}, {s('stX')}, {XS - 1}")
    A.e(f"s_cmp_ge_u32 {t(4)}, {XS}")
    A.e(f"s_cselect_b32 {t(5)}, {XS}, 0")
    A.e(f"s_sub_i32 {t(4)}, {t(4)}, {t(5)}")
    issue_x(A, x_slot_base(t(4)))
    unit(A, 0, "BA", "BB", v("XSC"), 16, cold)
    stamp(A, "P1")
    unit(A, 1, "BB", "BA", v("XSN"), 0, cold)
    A.drain()
    stamp(A, "P2")
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
    for lab, cb, bn, xs_vgpr, eo, fifo in cold:
        A.label(lab)
        A.fifo = list(fifo)
        fix_chunk(A, cb, bn, xs_vgpr, eo)
        A.e(f"s_branch {lab}r_%=")
    A.label("Lend")
    A.fifo = []
    if os.environ.get("FUSED_DEBUG"):
        # dump scalars as raw dwords into s_fin[0..31] (wave 0), read back by the kernel's debug path
        dbg = [s('pRef_lo'), s('pRef_hi'), t(20), t(21), s('M'), s('T'), t(2), t(3), s('c_tile'), s('n_rt'), s('w'), s('oFin'),
               s('c_pair'), s('c_rt'), s('pair0'), s('rt0')]
        A.e("s_waitcnt vmcnt(0)")
        A.e(f"s_lshl_b32 {t(0)}, {s('w')}, 6")
        A.e(f"s_add_i32 {t(0)}, {t(0)}, {s('oFin')}")
        A.e(f"v_mov_b32 v60, {t(0)}")
        for i, r in enumerate(dbg):
            A.e(f"v_mov_b32 v61, {r}")
            A.e(f"ds_write_b32 v60, v61 offset:{4 * i}")
        A.e("s_waitcnt lgkmcnt(0)")
        A.e("s_branch Ldbgend_%=")
    A.c("per-lane sums -> s_fin[w][64] (lanes 0..31 hold the candidates of both blocks)")
    A.e("s_waitcnt vmcnt(0)")
    if PROF:
        A.e(f"v_cmp_gt_u32 vcc, 16, {v('FROW')}")
        A.e(f"v_cndmask_b32 v60, v{PV['P2']}, v{PV['P0']}, vcc")
        A.e(f"v_cndmask_b32 v61, v{PV['P3']}, v{PV['P1']}, vcc")
        A.e(f"v_cvt_f64_u32 v[{V['RUN0']}:{V['RUN0'] + 1}], v60")
        A.e(f"v_cvt_f64_u32 v[{V['RUN1']}:{V['RUN1'] + 1}], v61")
    A.e(f"s_lshl_b32 {t(0)}, {s('w')}, 9")
    A.e(f"s_add_i32 {t(0)}, {t(0)}, {s('oFin')}")
    A.e(f"v_lshl_add_u32 v60, {v('FROW')}, 3, {t(0)}")
    A.e(f"v_cmp_eq_u32 vcc, 0, {v('FKG')}")
    A.e(f"s_and_saveexec_b64 {t2(16)}, vcc")
    A.e(f"ds_write_b64 v60, v[{V['RUN0']}:{V['RUN0'] + 1}]")
    A.e(f"ds_write_b64 v60, v[{V['RUN1']}:{V['RUN1'] + 1}] offset:256")
    A.e(f"s_mov_b64 exec, {t2(16)}")
    A.e("s_waitcnt lgkmcnt(0)")
    A.label("Ldbgend")
    return A


import re


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
    print(f"{out}: {len(A.lines)} lines, {n_mfma} MFMAs")


if __name__ == "__main__":
    main()
