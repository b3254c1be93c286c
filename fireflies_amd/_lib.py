"""Loader of libffx_hip.so — the ONLY compute backend of this package.

There is deliberately no CPU fallback: if the HIP library is missing or a tensor is not on a
HIP device, the ops raise.  (The scalar CPU restatement under oracle/ is test infrastructure and
is never imported from here.)
"""
import ctypes
import os
import subprocess

from . import _abi

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
# FFX_LIB lets an experiment load an alternative BUILD of the same HIP library (never another backend)
LIB_PATH = os.environ.get("FFX_LIB", os.path.join(CSRC, "libffx_hip.so"))

_api = None


def build(force=False):
    """Compile libffx_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-C", CSRC, "-j8"] + (["-B"] if force else [])
    subprocess.check_call(cmd)
    return LIB_PATH


def api():
    global _api
    if _api is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: the HIP extension has not been built. "
                "Run `python -c 'import __graft_entry__ as g; g.build()'` (or `make -C fireflies_amd/csrc`). "
                "fireflies_amd has no CPU fallback."
            )
        _api = _abi.Api(ctypes.CDLL(LIB_PATH))
        if _api.backend != "hip-gfx950":
            raise RuntimeError(f"unexpected backend {_api.backend!r} in {LIB_PATH}")
    return _api
