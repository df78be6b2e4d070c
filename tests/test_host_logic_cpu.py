"""Host-side logic of the product layers (FPCS driver, candidate flow, commits, reparam) on CPU.

The HIP kernels are replaced by the executable specs in tests/cpu_backend.py (injected through
adalog_amd.backend.set_backend, a test-only hook), so these tests pin everything *around* the kernels against the
golden fixtures captured from the reference.  The kernels themselves are pinned by the `-m gpu` tests, which run the
same cases (tests/layer_cases.py) on the HIP backend.
"""
import pytest
import torch

from adalog_amd import backend
from tests import cpu_backend, layer_cases as LC
from tests import cpu_backend as CB


@pytest.fixture(autouse=True)
def _cpu_backend():
    backend.set_backend(cpu_backend)
    yield
    backend.set_backend(None)


@pytest.mark.parametrize("name", ["linear_w3a3", "linear_w4a4", "linear_w6a6", "linear_w4a4_ragged"])
def test_linear_search(golden, name):
    LC.case_linear_search(golden, name)


@pytest.mark.parametrize("bits", [3, 4, 6])
def test_channelwise_reparam(golden, bits):
    LC.case_channelwise_reparam(golden, bits)


@pytest.mark.parametrize("bits", [3, 4, 6])
def test_postgelu_search(golden, bits):
    LC.case_postgelu_search(golden, bits)


@pytest.mark.parametrize("bits", [3, 4, 6])
def test_matmul_search(golden, bits):
    LC.case_matmul_search(golden, bits)


def test_matmul_searches_with_generated_candidates_commit_the_same_parameters():
    assert LC.case_matmul_gen_route() > 0


@pytest.mark.parametrize("bits", [3, 4, 6])
def test_postsoftmax_search(golden, bits):
    LC.case_postsoftmax_search(golden, bits)


@pytest.mark.parametrize("bits", [3, 4, 6])
def test_conv_search(golden, bits):
    LC.case_conv_search(golden, bits)


def test_modes_and_errors():
    LC.case_modes_and_errors()


def test_product_refuses_to_run_without_hip():
    """No silent fallback: with the test hook cleared and no GPU, resolving the backend raises."""
    import torch
    from adalog_amd._lib import AdalogHipError
    backend.set_backend(None)
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(AdalogHipError):
        backend.get()


def _emulated_ranks_quantiles(mod, shards, S, layouts, n_total, mbs, qs):
    """Drive mod.ShardedSelect for several emulated ranks in one process: the 'all-reduce' is the sum of their histograms."""
    lohi, w = mod.quantile_ranks(qs, n_total)
    sels = [mod.ShardedSelect(x2, S, 2 * len(qs), *lay, ranks=lohi.to(x2.device)) for x2, lay in zip(shards, layouts)]
    for p in range(4):
        for s in sels:
            s.hist_pass(p)
        total = sum(s.hist.clone() for s in sels)
        for s in sels:
            s.hist.copy_(total)
            s.pick(p)
    outs = [s.quantiles(w, mbs) for s in sels]
    for o in outs[1:]:
        assert torch.equal(o, outs[0])
    return outs[0]


def test_sharded_select_spec_matches_torch_quantile():
    """The radix-select specification with histograms summed over emulated ranks == torch.quantile on the gathered data:
    per-tensor chunks spanning ranks, chunks inside a rank, per-head chunks, and channel-wise segments."""
    gen = torch.Generator().manual_seed(11)
    qs = [0.9, 1.0, 1 - 0.9, 0.0]
    # (a) one global segment spread over 2 ranks
    x = torch.randn(2, 3000, generator=gen)
    got = _emulated_ranks_quantiles(CB, [x[0:1], x[1:2]], 1, [(0, 1, 0), (0, 1, 0)], 6000, 1, qs)
    want = torch.quantile(x.reshape(1, -1), torch.tensor(qs), dim=-1)
    torch.testing.assert_close(got, want, rtol=1e-6, atol=0)
    # (b) 4 chunks, 2 ranks: each rank holds two whole chunks; mean over the chunk quantiles
    x = torch.randn(4, 500, generator=gen)
    got = _emulated_ranks_quantiles(CB, [x[0:2], x[2:4]], 4, [(0, 2, 4), (2, 2, 4)], 500, 4, qs)
    want = torch.quantile(x, torch.tensor(qs), dim=-1).mean(-1, keepdim=True)
    torch.testing.assert_close(got, want, rtol=1e-6, atol=0)
    # (c) per-head (3 heads), 2 chunks per head, 2 ranks = one chunk per (head, rank)
    x = torch.randn(3, 2, 400, generator=gen)                       # [head, rank/chunk, elements]
    got = _emulated_ranks_quantiles(CB, [x[:, 0], x[:, 1]], 6, [(0, 1, 2), (1, 1, 2)], 400, 2, qs)
    want = torch.quantile(x.reshape(6, 400), torch.tensor(qs), dim=-1).view(4, 3, 2).mean(-1)
    torch.testing.assert_close(got, want, rtol=1e-6, atol=0)
    # (d) channel-wise: 5 channels, rows split 70 / 50 over the ranks
    x = torch.randn(5, 120, generator=gen)
    got = _emulated_ranks_quantiles(CB, [x[:, :70].contiguous(), x[:, 70:].contiguous()], 5, [(0, 5, 0), (0, 5, 0)], 120, 1, qs)
    want = torch.quantile(x, torch.tensor(qs), dim=-1)
    torch.testing.assert_close(got, want, rtol=1e-6, atol=0)
