"""Image-sharded calibration over torch.distributed (gloo, world_size 2, CPU stand-in kernels).

The N > 1 path: each rank keeps its contiguous image shard of raw_input / raw_out, every scoring call's [P, cols] score
tensor is all-reduced (SUM), percentile candidates gather the shards once, and the deterministic top-k makes every rank
commit identical parameters.  Checked: (1) ranks agree bit for bit, (2) the sharded run reaches the single-process result.
"""
import os
import socket
import tempfile

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _build(kind, g):
    from adalog_amd import quant_layers as Q
    t = lambda a: torch.from_numpy(np.asarray(a))
    if kind == "linear":
        wb, ab, N, Tn, I, Oc, n_V, cbs = [int(v) for v in g["cfg"]]
        lay = Q.AsymmetricallyBatchingQuantLinear(I, Oc, True, "raw", wb, ab, calib_batch_size=cbs, search_round=2,
                                                  eq_n=128, n_V=n_V, fpcs=True, steps=4)
        lay.weight.data.copy_(t(g["weight"])); lay.bias.data.copy_(t(g["bias"]))
        return lay, [t(g["x"])]
    if kind == "postgelu":
        wb, ab, N, Tn, I, Oc, n_V, cbs = [int(v) for v in g["cfg"]]
        lay = Q.PostGeluLogBasedBatchingQuantLinear(I, Oc, True, "raw", wb, ab, calib_batch_size=cbs, search_round=1,
                                                    eq_n=128, n_V=1, quantizer="adalog", fpcs=True, steps=3)
        lay.weight.data.copy_(t(g["weight"])); lay.bias.data.copy_(t(g["bias"]))
        return lay, [t(g["x"])]
    _, _, N, H, S, C, cbs = [int(v) for v in g["cfg"]]
    cls = Q.PostSoftmaxAsymmetricallyBatchingQuantMatMul if kind == "postsoftmax" else Q.AsymmetricallyBatchingQuantMatMul
    kw = dict(quantizer="adalog") if kind == "postsoftmax" else {}
    lay = cls(A_bit=4, B_bit=4, mode="raw", calib_batch_size=cbs, search_round=1, eq_n=128, head_channel_wise=True,
              num_heads=H, fpcs=True, steps=3, **kw)
    return lay, [t(g["A"]), t(g["B"])]


def _search(lay, inputs, lo, hi):
    with torch.no_grad():
        full_out = lay(*inputs)
        shard = [x[lo:hi].contiguous() for x in inputs]
        lay.raw_input = shard[0] if len(shard) == 1 else shard
        lay.raw_out = full_out[lo:hi].contiguous()
        lay.hyperparameter_searching()
    return {k: v.clone() for k, v in lay.state_dict().items()}


def _worker(rank, world, port, kind, fixture, outdir):
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from adalog_amd import backend, parallel
    from tests import cpu_backend
    torch.set_num_threads(2)
    backend.set_backend(cpu_backend)
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    g = np.load(os.path.join(ROOT, "tests", "golden", fixture + ".npz"))
    lay, inputs = _build(kind, g)
    lo, hi = parallel.shard_slice(inputs[0].shape[0])
    sd = _search(lay, inputs, lo, hi)
    torch.save(sd, os.path.join(outdir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("kind,fixture", [("linear", "linear_w4a4"), ("postgelu", "postgelu_w4a4"),
                                          ("matmul", "matmul_a4b4"), ("postsoftmax", "postsoftmax_a4b4")])
def test_sharded_search_matches_single_process(kind, fixture):
    from adalog_amd import backend
    from tests import cpu_backend
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(2, _free_port(), kind, fixture, d), nprocs=2, join=True)
        r0, r1 = torch.load(os.path.join(d, "rank0.pt")), torch.load(os.path.join(d, "rank1.pt"))
    for k in r0:
        assert torch.equal(r0[k], r1[k]), f"ranks disagree on {k}"
    backend.set_backend(cpu_backend)
    try:
        g = np.load(os.path.join(ROOT, "tests", "golden", fixture + ".npz"))
        lay, inputs = _build(kind, g)
        single = _search(lay, inputs, 0, inputs[0].shape[0])
    finally:
        backend.set_backend(None)
    for k in single:
        if "zero_point" in k or k.endswith(".q"):
            assert (single[k] != r0[k]).float().mean().item() <= 0.1, k      # exact ties may resolve differently
        elif "scale" in k:
            torch.testing.assert_close(r0[k], single[k], rtol=2e-3, atol=0, msg=lambda m: f"{k}: {m}")


def _gram_w_case(seed=11, N=8, Tn=9, I=64, Oc=96, bits=4):
    """a Linear layer with a FIXED per-tensor activation quantiser (an FPCS grid point), its captures, and the weight search's state"""
    from adalog_amd import quant_layers as Q
    from oracle import adalog_oracle as O
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(N, Tn, I, generator=g)
    x[..., : I // 8] *= 3.0
    W = torch.randn(Oc, I, generator=g) * 0.05
    b = torch.randn(Oc, generator=g) * 0.1
    lay = Q.AsymmetricallyBatchingQuantLinear(I, Oc, True, "raw", bits, bits, calib_batch_size=N, search_round=1, eq_n=128, n_V=1,
                                              fpcs=True, steps=6)
    lay.weight.data.copy_(W); lay.bias.data.copy_(b)
    sca, zpa = O.activation_candidates(x, bits, False)
    aq = lay.a_quantizer
    aq.scale.data.copy_(sca[:, 60].view(aq.scale.shape)); aq.zero_point.data.copy_(zpa[:, 60].float().view(aq.zero_point.shape))
    aq.inited, aq._zp_on_grid = True, True
    return lay, x


def _gram_w_search(lay, x, lo, hi):
    from adalog_amd import parallel, search
    with torch.no_grad():
        full_out = lay(x)
        lay.raw_input, lay.raw_out = x[lo:hi].contiguous(), full_out[lo:hi].contiguous()
        lay._initialize_calib_parameters()
        search.forget_grids(); search.begin_rounds(lay)
        parallel.reset_stats()
        lay.weight_fpcs(steps=6, search_strategy="output")
        st = parallel.collective_stats()
    wq = lay.w_quantizer
    return wq.scale.detach().clone(), wq.zero_point.detach().clone(), st


def _gram_w_worker(rank, world, port, outdir):
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from adalog_amd import backend, parallel
    from tests import cpu_backend
    torch.set_num_threads(2)
    backend.set_backend(cpu_backend)
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    lay, x = _gram_w_case()
    lo, hi = parallel.shard_slice(x.shape[0])
    s_, z_, st = _gram_w_search(lay, x, lo, hi)
    torch.save({"scale": s_, "zp": z_, "collectives": st["collectives"]}, os.path.join(outdir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_gram_weight_search_with_the_state_all_reduced_once(world):
    """The Gram form of the output-MSE weight search (reference quant_layers/linear.py:355-392, 483-503) under image sharding: G, c
    and S0 are sums over tokens, so the build all-reduces them once (amax: MAX; G, c: int64 SUM, exact; S0: fp64 SUM) and every
    rank then scores the same FINAL scores -- the six FPCS steps issue NO collective.  Checked on gloo with the integer
    specification of the kernels (tests/cpu_backend.GramState): every rank commits bit-identical parameters, they EQUAL the
    one-process search's (the state is the same integers), and the search issued exactly the build's four all-reduces (plus the
    percentile grid's none: weights are replicated)."""
    from adalog_amd import backend
    from tests import cpu_backend
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_gram_w_worker, args=(world, _free_port(), d), nprocs=world, join=True)
        rs = [torch.load(os.path.join(d, f"rank{r}.pt")) for r in range(world)]
    for r in rs[1:]:
        assert torch.equal(r["scale"], rs[0]["scale"]) and torch.equal(r["zp"], rs[0]["zp"])
    assert all(r["collectives"] == 4 for r in rs), [r["collectives"] for r in rs]
    backend.set_backend(cpu_backend)
    try:
        lay, x = _gram_w_case()
        s1, z1, st1 = _gram_w_search(lay, x, 0, x.shape[0])
    finally:
        backend.set_backend(None)
    assert torch.equal(rs[0]["scale"], s1) and torch.equal(rs[0]["zp"], z1)


def test_shard_slice_and_gather_single_process():
    from adalog_amd import parallel
    assert parallel.world_size() == 1 and parallel.rank() == 0
    assert parallel.shard_slice(32) == (0, 32)
    x = torch.arange(6.0).view(3, 2)
    assert parallel.gather_images(x) is x and parallel.all_reduce_sum(x) is x


def _brecq_worker(rank, world, port, outdir):
    """BRECQ data parallelism: every rank trains the same block on its own shard of the block inputs; the
    gradients of alpha / activation scales are all-reduced (mean) each iteration, so the ranks' parameters stay identical."""
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from adalog_amd import backend, quant_layers as Q
    from adalog_amd.utils.block_recon import BlockReconstructor
    from tests import cpu_backend
    torch.set_num_threads(2)
    backend.set_backend(cpu_backend)
    if world > 1:
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    g = np.load(os.path.join(ROOT, "tests", "golden", "brecq_toy.npz"))
    I, Hd = 16, 32

    class Blk(torch.nn.Module):
        def __init__(self):
            super().__init__()
            kw = dict(mode="raw", w_bit=4, a_bit=4, calib_batch_size=4, search_round=1, eq_n=128, fpcs=True, steps=2)
            self.fc1 = Q.AsymmetricallyBatchingQuantLinear(I, Hd, True, n_V=1, **kw)
            self.fc2 = Q.PostGeluLogBasedBatchingQuantLinear(Hd, I, True, n_V=1, quantizer="adalog", **kw)

        def forward(self, x):
            return x + self.fc2(torch.nn.functional.gelu(self.fc1(x)))

    blk = Blk().eval()
    sd = {k[4:].replace("__", "."): torch.from_numpy(np.asarray(v)) for k, v in g.items() if k.startswith("cal_") and "matmul1" not in k}
    blk.load_state_dict(sd, strict=False)
    for m in blk.modules():
        if hasattr(m, "mode"):
            m.calibrated = True
            m.a_quantizer.inited = m.w_quantizer.inited = True
    gen = torch.Generator().manual_seed(99)
    x = torch.randn(16, 5, I, generator=gen)
    with torch.no_grad():
        y = blk(x)                                                 # FP targets (mode 'raw')
    per = 16 // world
    blk.raw_input, blk.raw_out = x[rank * per:(rank + 1) * per].clone(), y[rank * per:(rank + 1) * per].clone()
    rec = object.__new__(BlockReconstructor)
    rec.reconstruct_single_block("blk", blk, torch.device("cpu"), batch_size=8, iters=12, quant_act=True)
    out = {"alpha1": blk.fc1.w_quantizer.alpha.data.clone(), "alpha2": blk.fc2.w_quantizer.alpha.data.clone(),
           "s1": blk.fc1.a_quantizer.scale.data.clone(), "s2": blk.fc2.a_quantizer.scale.data.clone()}
    torch.save(out, os.path.join(outdir, f"brecq_rank{rank}_of{world}.pt"))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def test_brecq_gradient_all_reduce_keeps_ranks_in_lock_step():
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_brecq_worker, args=(2, _free_port(), d), nprocs=2, join=True)
        r0, r1 = torch.load(os.path.join(d, "brecq_rank0_of2.pt")), torch.load(os.path.join(d, "brecq_rank1_of2.pt"))
        mp.spawn(_brecq_worker, args=(1, _free_port(), d), nprocs=1, join=True)
        single = torch.load(os.path.join(d, "brecq_rank0_of1.pt"))
    for k in r0:
        assert torch.equal(r0[k], r1[k]), f"ranks diverged on {k}: the gradients were not all-reduced"
    # the ranks saw different samples than the single process, so the parameters moved differently -- but they did move
    assert not torch.equal(r0["alpha1"], single["alpha1"])
    init = torch.from_numpy(np.load(os.path.join(ROOT, "tests", "golden", "brecq_toy.npz"))["alpha_fc1"])
    assert (r0["alpha1"] - init).abs().max().item() > 1e-4


def _calib_worker(rank, world, port, interleave, outdir):
    """The whole calibrator over sharded images: a tiny wrapped ViT, `world` gloo ranks, the two-lane interleaved schedule
    (adalog_amd.parallel lanes: two modules' searches side by side, each lane on its own communicator) or the sequential one."""
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      ADALOG_INTERLEAVE="1" if interleave else "0")
    import torch.distributed as dist
    from adalog_amd import backend, parallel
    from adalog_amd.utils.calibrator import QuantCalibrator
    from adalog_amd.utils.wrap_net import wrap_modules_in_net, wrap_reparamed_modules_in_net
    from tests import cpu_backend, wrapper_cases as WC
    torch.set_num_threads(1)
    backend.set_backend(cpu_backend)
    if world > 1:
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    g = np.load(os.path.join(ROOT, "tests", "golden", "wrapper_rules.npz"))
    cfg = WC.cfg_of(4)
    cfg.search_round, cfg.steps, cfg.calib_batch_size = 1, 2, 2
    vit = wrap_modules_in_net(WC._load(WC.tiny_vit(), g, "vit_in_", torch.device("cpu")), cfg, reparam=True)
    x = torch.randn(8, 3, 32, 32, generator=torch.Generator().manual_seed(3))
    per = 8 // world
    xs = x[rank * per:(rank + 1) * per]
    parallel.reset_stats()
    QuantCalibrator(vit, [(xs, None)], capture="block").batching_quant_calib()
    stats = parallel.collective_stats()
    vit = wrap_reparamed_modules_in_net(vit)
    sd = {k: v.detach().clone() for k, v in vit.state_dict().items()}
    sd["__collectives"] = torch.tensor(stats["collectives"])
    torch.save(sd, os.path.join(outdir, f"calib_w{world}_i{int(interleave)}_r{rank}.pt"))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_interleaved_calibration_is_the_sequential_one(world):
    """Two lanes (module-parallel searches whose score all-reduces overlap the other lane's GEMMs) must change nothing: every
    rank holds bit-identical parameters, identical to the sequential schedule's on the same ranks, and the same number of
    collectives is issued; the sharded result agrees with the single-process one up to exact-tie resolution."""
    with tempfile.TemporaryDirectory() as d:
        for inter in (True, False):
            mp.spawn(_calib_worker, args=(world, _free_port(), inter, d), nprocs=world, join=True)
        mp.spawn(_calib_worker, args=(1, _free_port(), False, d), nprocs=1, join=True)
        runs = {(i, r): torch.load(os.path.join(d, f"calib_w{world}_i{i}_r{r}.pt")) for i in (0, 1) for r in range(world)}
        single = torch.load(os.path.join(d, "calib_w1_i0_r0.pt"))
    assert int(runs[(1, 0)]["__collectives"]) > 100
    for k in runs[(1, 0)]:
        for r in range(1, world):
            assert torch.equal(runs[(1, 0)][k], runs[(1, r)][k]), f"interleaved: ranks disagree on {k}"
        assert torch.equal(runs[(1, 0)][k], runs[(0, 0)][k]), f"interleaved and sequential schedules disagree on {k}"
    n_scale = n_off = 0
    for k, v in single.items():
        if k.endswith(".scale"):
            n_scale += v.numel()
            n_off += int(((v - runs[(1, 0)][k]).abs() > 2e-3 * v.abs()).sum())
    assert n_off <= 0.05 * n_scale, (n_off, n_scale)


def _brecq_blocks_worker(rank, world, port, outdir):
    """reconstruct_model over `world` gloo ranks with the optimisation images sharded: block-parallel mode (each rank trains the
    blocks it owns as a single process would, results broadcast)."""
    import copy
    import importlib.util
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from adalog_amd import backend
    from adalog_amd.utils.block_recon import BlockReconstructor
    from adalog_amd.utils.calibrator import QuantCalibrator
    from adalog_amd.utils.models import VisionTransformer
    from adalog_amd.utils.wrap_net import wrap_modules_in_net, wrap_reparamed_modules_in_net
    from tests import cpu_backend
    torch.set_num_threads(1)
    backend.set_backend(cpu_backend)
    spec = importlib.util.spec_from_file_location("cfg4d", os.path.join(ROOT, "configs", "4bit.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    cfg = mod.Config()
    cfg.search_round, cfg.steps = 1, 2
    torch.manual_seed(7)
    model = VisionTransformer(img_size=32, patch_size=8, embed_dim=32, depth=2, num_heads=2, num_classes=10).eval()
    for p_ in model.parameters():
        p_.data.mul_(8.0)
    full = copy.deepcopy(model)
    x = torch.randn(16, 3, 32, 32, generator=torch.Generator().manual_seed(11))
    model = wrap_modules_in_net(model, cfg, reparam=True)
    QuantCalibrator(model, [(x[:8], None), (x[8:], None)]).batching_quant_calib()     # (calibration itself: every rank, all images)
    model = wrap_reparamed_modules_in_net(model)
    if world > 1:
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    per = 16 // world
    xs = x[rank * per:(rank + 1) * per]
    br = BlockReconstructor(model, full, [(xs, None)])
    br.reconstruct_model(quant_act=True, keep_gpu=True, iters=10)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    sd["__mode"] = torch.tensor({"single": 0, "block": 1, "batch": 2}[br.dp_mode])
    torch.save(sd, os.path.join(outdir, f"brecq_blocks_w{world}_r{rank}.pt"))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def test_block_parallel_brecq_equals_the_single_process_run():
    """Blocks dealt to 2 ranks, images sharded: every rank ends with the model the single process trains (bit for bit: an owner
    trains its block exactly as one process would, on the all-gathered block data)."""
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_brecq_blocks_worker, args=(2, _free_port(), d), nprocs=2, join=True)
        mp.spawn(_brecq_blocks_worker, args=(1, _free_port(), d), nprocs=1, join=True)
        r0, r1 = (torch.load(os.path.join(d, f"brecq_blocks_w2_r{r}.pt")) for r in (0, 1))
        single = torch.load(os.path.join(d, "brecq_blocks_w1_r0.pt"))
    assert int(r0["__mode"]) == 1 and int(single["__mode"]) == 0
    for k in single:
        if k == "__mode":
            continue
        assert torch.equal(r0[k], r1[k]), f"ranks disagree on {k}"
        assert torch.equal(r0[k], single[k]), f"block-parallel and single-process runs disagree on {k}"


def test_sequencer_issues_one_global_order():
    """parallel.Sequencer: whatever the thread timing, the collectives of two lanes go out in the order (call index, lane)."""
    import random
    import threading
    import time
    from adalog_amd import parallel
    for trial in range(5):
        seq = parallel.Sequencer(2)
        counts = (7, 4)                                   # lane 1 retires early
        issued = []

        def lane(li):
            rnd = random.Random(100 * trial + li)
            for _ in range(counts[li]):
                time.sleep(rnd.random() * 0.003)
                seq.issue(li, lambda: issued.append(li))
            seq.finish(li)
        ts = [threading.Thread(target=lane, args=(li,)) for li in (0, 1)]
        for t in ts:
            t.start()
        for t in ts:
            t.join(timeout=10)
        assert not any(t.is_alive() for t in ts)
        assert seq.order == sorted(seq.order, key=lambda lc: (lc[1], lc[0])), seq.order
        assert seq.order == [(0, 0), (1, 0), (0, 1), (1, 1), (0, 2), (1, 2), (0, 3), (1, 3), (0, 4), (0, 5), (0, 6)]
