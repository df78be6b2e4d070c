"""Prints the Python stack of an ATen op (by name and input shape) inside a BRECQ iteration."""
import os, sys, runpy
os.environ["ADALOG_BRECQ_GRAPH"] = "0"
import torch
want_name, want_shape = sys.argv[1], eval(sys.argv[2])
import traceback
from torch.utils._python_dispatch import TorchDispatchMode
seen = set()
class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if want_name in name:
            shapes = [tuple(a.shape) for a in args if torch.is_tensor(a)]
            if shapes and shapes[0] == tuple(want_shape):
                key = (name, tuple(shapes))
                if key not in seen:
                    seen.add(key)
                    node = torch._C._current_autograd_node()
                    print("====", name, shapes, "autograd node:", None if node is None else node.name())
                    print("".join(traceback.format_stack(limit=6)[:-1]))
        return func(*args, **(kwargs or {}))
sys.argv = [sys.argv[0]]
os.environ["ITERS"] = "6"
with Spy():
    runpy.run_path(os.path.join(os.path.dirname(__file__), "brecq_ops.py"), run_name="__main__")
