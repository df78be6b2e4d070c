"""Kernel backend selection.

The product backend is the HIP kernel library (adalog_amd.ops over libadalog_hip.so) and it is the ONLY one this
package contains: resolving it fails loudly when the library has not been built or no HIP device is present.
``set_backend`` exists so that the *host-side* logic (FPCS driver, calibrator, image sharding) can be exercised by CPU
unit tests with a stand-in that lives under tests/; the package never selects anything but HIP by itself.
"""
import torch

_backend = None


def _resolve_hip():
    from . import _lib, ops
    _lib.load()                                  # raises AdalogHipError when the .so is missing
    if not torch.cuda.is_available():
        raise _lib.AdalogHipError(
            "adalog_amd needs a HIP device (MI355X): torch.cuda.is_available() is False and there is no CPU fallback")
    return ops


def get():
    global _backend
    if _backend is None:
        _backend = _resolve_hip()
    return _backend


def set_backend(impl):
    """Test hook (host-logic unit tests only).  Pass None to restore strict HIP resolution."""
    global _backend
    _backend = impl
