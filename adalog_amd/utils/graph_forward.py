"""HIP-graph replay of a calibrated model's inference forward (validate() / fidelity checks).

A quant_forward pass of a ViT is ~25 small launches per block (operand packs, integer MFMA products, LayerNorm / softmax / GELU):
at 32 images it is launch-bound, not bandwidth-bound.  The forward is a fixed launch sequence for a fixed input shape -- no host
reads once the layers' caches are warm -- so it is captured once per input shape and replayed (torch.cuda.CUDAGraph: the capture
records the kernels the C ABI enqueues on torch's current stream, exactly as the BRECQ loop does, utils/block_recon.py).
No tracing compiler is involved: the graph holds the same hand-written kernels in the same order."""
import torch


class GraphedForward:
    """callable like ``model``; replays a captured forward where the input shape has been seen, else captures it (after one eager
    warm-up pass that fills the layers' packed-weight / table caches).  ``enabled=False`` or a CPU model: plain calls."""

    def __init__(self, model, enabled=True):
        self.model, self.enabled = model, enabled
        self._graphs = {}

    @torch.no_grad()
    def __call__(self, x):
        if not (self.enabled and x.is_cuda):
            return self.model(x)
        key = (tuple(x.shape), x.dtype, x.device)
        hit = self._graphs.get(key)
        if hit is None:
            static_in = x.clone()
            self.model(static_in)                               # warm-up: caches, lazily read host values (AdaLog base q)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                static_out = self.model(static_in)
            hit = self._graphs[key] = (graph, static_in, static_out)
        graph, static_in, static_out = hit
        static_in.copy_(x)
        graph.replay()
        return static_out.clone()
