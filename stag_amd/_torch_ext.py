"""PyTorch-ROCm front end of the hot calls (stag_amd/csrc/torch_ext.cpp): `torch.ops.stag.agg_fwd` /
`agg_bwd` — dispatcher ops with Meta kernels over the same C ABI as the ctypes binding (include/stag_hip.h).

Which front end runs: measured on the GPU box (tools/host_probe.py, a Cora-sized graph, 5000 calls, no autograd
node), one `ops.aggregate` costs 12.6 us of host time through ctypes and 12.5 us through the dispatcher (23
arguments to parse, box and match against the schema cost what ctypes' marshalling costs) — no gain, so eager mode
keeps ctypes and does not depend on a second shared object.  The dispatcher ops are what a traced or compiled graph
needs (they are visible to it, and their Meta kernels give it the output shapes), so they are used while
`torch.compiler.is_compiling()`, or always with STAG_TORCH_OPS=1.  The HIP library underneath is the same.
"""
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_stag_torch.so")
_state = {"loaded": None}


def loaded():
    """True once `_stag_torch.so` is registered with the dispatcher (loads it on first use; False if not built)."""
    if _state["loaded"] is None:
        ok = False
        if os.path.exists(_SO):
            from . import _lib
            _lib.lib()                       # libstag_hip.so first: the module links against it
            torch.ops.load_library(_SO)
            ok = int(torch.ops.stag.abi_version()) == int(_lib.lib().stag_abi_version())
            if not ok:
                raise _lib.StagHipError("_stag_torch.so was built against another ABI version: rebuild (make -C stag_amd/csrc)")
        _state["loaded"] = ok
    return _state["loaded"]


_is_compiling = getattr(getattr(torch, "compiler", None), "is_compiling", lambda: False)


def available():
    """True when ops.py should call torch.ops.stag.* instead of the ctypes binding."""
    if os.environ.get("STAG_TORCH_OPS") == "1" or _is_compiling():
        return loaded()
    return False
