"""The C-ABI library loads and exports every symbol include/adalog_hip.h declares (no compute: CPU tier)."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "adalog_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(adalog_[a-z0-9_]+)\s*\(", src)))


def test_header_lists_functions():
    names = header_functions()
    assert "adalog_gemm_score" in names and "adalog_uniform_fake_quant_f32" in names and len(names) >= 18


def test_library_exports_every_declared_symbol():
    from adalog_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = _lib.load()
    for name in header_functions():
        assert hasattr(lib, name), f"{name} declared in include/adalog_hip.h but not exported"
        assert name in _lib.SIGNATURES, f"{name} has no ctypes signature"
    assert set(_lib.SIGNATURES) == set(header_functions())
    assert lib.adalog_abi_version() == 1


def test_argument_rejection_without_gpu():
    """Entry points validate arguments before touching the device (returns -1 + message)."""
    from adalog_amd import _lib
    lib = _lib.load()
    rc = lib.adalog_topk(None, 128, 4, 1, None, None)
    assert rc == -1 and b"topk" in lib.adalog_last_error()
    rc = lib.adalog_gemm_score(7, None, None, 0, 0, 0, 0, 1, 1, 64, 0, 1, 1, 1, None, 0, 0, 1, 1, None, 0, 0, 1.0,
                               None, 0, 0, 0, None, 0, 0, 0, None, None, None, 0, None, 0, 0, 0, 0, 0, None)
    assert rc == -1


def test_product_package_never_imports_oracle():
    """The oracle is test infrastructure: nothing under adalog_amd/ may reference it."""
    out = subprocess.run(["grep", "-rIl", "-E", r"(^|[^_a-zA-Z])oracle", os.path.join(ROOT, "adalog_amd"),
                          "--include=*.py", "--include=*.hip", "--include=*.h"], capture_output=True, text=True).stdout
    assert out.strip() == "", f"product files mention the oracle: {out}"


def test_torch_ops_registered_hip_only():
    """torch.ops.adalog.* (north_star: 'through PyTorch-ROCm custom ops'): the TORCH_LIBRARY registration loads, declares the
    expected schemas and has NO CPU kernel (calling it with CPU tensors is an error, not a fallback)."""
    import torch
    from adalog_amd import _torch_ops
    if not os.path.exists(_torch_ops.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    assert _torch_ops.available()
    for name in _torch_ops.OPS:
        assert hasattr(torch.ops.adalog, name), name
    with pytest.raises(NotImplementedError):
        torch.ops.adalog.topk(torch.zeros(4, 3), 2)
    with pytest.raises(NotImplementedError):
        torch.ops.adalog.uniform_fake_quant(torch.zeros(4), torch.ones(1), torch.zeros(1), 1, 4, 4, False)


def test_generated_asm_is_up_to_date(tmp_path):
    """The hand-scheduled loops of gemm_fused.hip are generated text (tools/gen_fused_asm.py); the committed .inc files
    must be what the generator emits today (edit the generator, run tools/gen_fused_asm_all.sh, commit both)."""
    import importlib
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    for nrb, fns in ((12, 4), (12, 3), (8, 4), (4, 4)):
        os.environ["FUSED_NRB"], os.environ["FUSED_FNS"] = str(nrb), str(fns)
        for k in ("FUSED_STAGE", "FUSED_NOCOLD", "FUSED_NOBAR", "FUSED_NOMFMA"):
            os.environ.pop(k, None)
        import gen_fused_asm as G
        G = importlib.reload(G)
        lines = G.hazard_pass(G.program().lines)
        want = ['"' + ln + '\\n\\t"' for ln in lines if not ln.startswith(";")]
        have = [ln.rstrip("\n") for ln in open(os.path.join(ROOT, "adalog_amd", "csrc", f"fused_loop_nrb{nrb}_s{fns}.inc"))
                if ln.startswith('"')]
        assert have == want, f"fused_loop_nrb{nrb}_s{fns}.inc is stale: run tools/gen_fused_asm_all.sh"
    os.environ.pop("FUSED_NRB", None)
    os.environ.pop("FUSED_FNS", None)


def test_brecq_ticket_is_ordered_after_the_partial_stores(tmp_path):
    """ISA shape of the 'last block finishes' reductions (csrc/brecq.hip last_block_arrives): between a block's partial
    store (global_store ... sc1) and its ticket (global_atomic_add) there must be an `s_waitcnt vmcnt(0)` -- a workgroup
    fence alone compiles to lgkmcnt(0), which does not order two VMEM operations to different addresses."""
    import shutil
    if shutil.which("hipcc") is None:
        pytest.skip("hipcc not available")
    asm = tmp_path / "brecq.s"
    csrc = os.path.join(ROOT, "adalog_amd", "csrc")
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "--cuda-device-only", "-S",
                    os.path.join(csrc, "brecq.hip"), "-o", str(asm)], check=True, capture_output=True)
    lines = open(asm).read().splitlines()
    atomics = [i for i, ln in enumerate(lines) if "global_atomic_add" in ln]
    assert len(atomics) >= 5, "expected one ticket per reducing kernel"
    for i in atomics:
        j = i - 1
        while j >= 0 and "global_store_dword" not in lines[j]:
            j -= 1
        assert j >= 0, "no partial store in front of a ticket"
        between = lines[j + 1:i]
        assert any("s_waitcnt" in ln and "vmcnt(0)" in ln for ln in between), \
            f"ticket at line {i} is not ordered after the store at line {j}"


def test_slab_register_loads_are_not_touched_before_their_wait(tmp_path):
    """ISA shape of the slab kernel's register path (csrc/gemm_k_slab.inc: 24 inline-asm `buffer_load_dwordx4` into the first
    six accumulator blocks, then an asm `s_waitcnt vmcnt(0)` naming them).  hipcc does not know the loads are asynchronous:
    nothing in the language stops it from copying, spilling or re-using a destination register between a load and the wait
    (it would read or clobber a register the memory system has not written yet).  The compiled kernels must show the 24
    loads and their wait with nothing but scalar instructions in between: no VALU, no `v_accvgpr_*`, no scratch access."""
    import re
    import shutil
    if shutil.which("hipcc") is None:
        pytest.skip("hipcc not available")
    asm = tmp_path / "gemm_score.s"
    csrc = os.path.join(ROOT, "adalog_amd", "csrc")
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "--cuda-device-only", "-S",
                    os.path.join(csrc, "gemm_score.hip"), "-o", str(asm)], check=True, capture_output=True)
    lines = [ln.strip() for ln in open(asm).read().splitlines()]
    is_reg_load = lambda ln: ln.startswith("buffer_load_dwordx4") and " lds" not in ln
    runs, i, fn = 0, 0, ""
    while i < len(lines):
        mfn = re.match(r"(_Z\S+):", lines[i])                 # a function label (a comment may follow it)
        if mfn:
            fn = mfn.group(1)
        if "k_gemm_slab" not in fn or not is_reg_load(lines[i]):
            i += 1
            continue
        j, loads, pending, first_bad = i, 0, set(), None
        while j < len(lines) and not (lines[j].startswith("s_waitcnt") and "vmcnt(0)" in lines[j]):
            ln = lines[j]
            if is_reg_load(ln):
                loads += 1
                m = re.match(r"buffer_load_dwordx4 v\[(\d+):(\d+)\], (\S+),", ln)
                assert m, ln
                dst = set(range(int(m.group(1)), int(m.group(2)) + 1))
                assert not dst & pending, f"line {j}: a destination is loaded twice in one run"
                addr = {int(x) for x in re.findall(r"v(\d+)", m.group(3))}
                if addr & pending and first_bad is None:
                    first_bad = (j, ln)
                pending |= dst
            elif ln and not ln.startswith((";", ".", "s_")):
                # VALU / scratch / accvgpr traffic is fine as long as it stays off the registers still in flight
                used = set()
                for lo, hi in re.findall(r"[va]\[(\d+):(\d+)\]", ln):
                    used |= set(range(int(lo), int(hi) + 1))
                used |= {int(x) for x in re.findall(r"\bv(\d+)\b", ln)}
                if used & pending and first_bad is None:
                    first_bad = (j, ln)
            j += 1
        if loads >= 24:
            runs += 1
            assert first_bad is None, f"line {first_bad[0]}: {first_bad[1]!r} touches a register whose load is still in flight"
        i = j + 1
    assert runs >= 1, "no run of 24 register loads found in k_gemm_slab: has the slab register path changed?"
