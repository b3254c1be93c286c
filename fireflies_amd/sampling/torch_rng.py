"""The `torch.rand` of a CUDA sampler bound, evaluated on the host (ffx_torch_rand_h, include/ffx.h).

The reference draws `torch.rand(a.shape, device=a.device)` per sampler (fireflies/utils/math.py:170-175)
and syncs per value (fireflies/scene.py:258-274).  For the default CUDA generator those numbers are a
pure function of (seed, offset), so the product computes them natively on the host and advances the
generator's offset by what the device launch would have consumed: same values, same stream for every
other consumer of the generator, no launch, no transfer, no sync.  FFX_HOST_PHILOX=0 switches back to
device draws (Scene then pre-draws them on a side stream, scene.py).
"""
import ctypes as C
import os

import numpy as np
import torch

from .. import _lib

_ENABLED = os.environ.get("FFX_HOST_PHILOX", "1") != "0"
_INC = C.c_uint64(0)
_BUF = (C.c_float * 256)()
_FN = None
_GENS = {}  # device (as the samplers hold it) -> its default CUDA generator
_CHECKED_DEVICES = set()  # devices (as the samplers hold them) whose host evaluation has been verified


def enabled() -> bool:
    return _ENABLED


_VERIFIED = {}  # device index -> bool


def verified(device) -> bool:
    """enabled() AND this process has seen, once per device, that the host evaluation reproduces `torch.rand` on that
    device bit for bit and leaves the generator at the same offset.  The host path hard-codes what PyTorch-ROCm's uniform
    kernel does (Philox-4x32-10, +4 offset per launch of <= 256 elements, rocrand's float conversion): a torch / rocrand
    upgrade that changes any of it would otherwise silently change seeded streams.  On a mismatch the host path is
    switched off for the process (device draws, as with FFX_HOST_PHILOX=0) with a warning.  The generator is left exactly
    as it was found."""
    global _ENABLED
    if not _ENABLED or not torch.cuda.is_available():
        return False
    dev = torch.device(device)
    if dev.type != "cuda":
        return False
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    ok = _VERIFIED.get(idx)
    if ok is None:
        gen = torch.cuda.default_generators[idx]
        state = gen.get_state()
        try:
            ok = True
            gen.manual_seed(0x5EED0FF5E7)
            for n in (3, 256):
                off0 = gen.get_offset()
                want = torch.rand(n, device=torch.device("cuda", idx)).cpu().numpy()
                off_t = gen.get_offset()
                gen.set_offset(off0)
                got = host_rand(n, torch.device("cuda", idx))
                ok = ok and got is not None and np.array_equal(want, got) and gen.get_offset() == off_t
        finally:
            gen.set_state(state)
        _VERIFIED[idx] = ok
        if not ok:
            import warnings

            warnings.warn("fireflies_amd: the host evaluation of torch.rand (ffx_torch_rand_h) does not reproduce this PyTorch build's CUDA "
                          "uniform kernel; sampler draws fall back to the device (FFX_HOST_PHILOX=0 behaviour)")
            _ENABLED = False
    return bool(ok)


def host_rand(numel: int, device):
    """the `numel` values torch.rand(..., device=device) would return now (float32 numpy array), with the
    device generator advanced accordingly; None if this draw has to be made on the device."""
    global _FN
    if not _ENABLED or numel < 1 or numel > 256 or torch.cuda.is_current_stream_capturing():
        return None
    if _FN is None:
        _FN = _lib.api().lib.ffx_torch_rand_h
    dev = torch.device(device)
    gen = torch.cuda.default_generators[dev.index if dev.index is not None else torch.cuda.current_device()]
    off = gen.get_offset()
    if _FN(gen.initial_seed(), off, numel, _BUF, C.byref(_INC)) != 0:
        return None
    gen.set_offset(off + _INC.value)
    return np.ctypeslib.as_array(_BUF)[:numel].copy()


class HostDraws:
    """the host-evaluated draws of one DrawBatch: `reserve` claims the generator's next offsets in program order (so that
    every other consumer of the generator sees the stream it would have seen), `resolve` computes all values with ONE
    native call (ffx_torch_rand_batch_h)."""

    __slots__ = ("seeds", "offsets", "counts", "_values")

    def __init__(self):
        self.seeds, self.offsets, self.counts, self._values = [], [], [], None

    def reserve(self, numel: int, device):
        """index of the reserved draw, or None if this draw has to be made on the device"""
        if not _ENABLED or numel < 1 or numel > 256 or torch.cuda.is_current_stream_capturing():
            return None
        if device not in _CHECKED_DEVICES:  # first draw on this device in this process: the one-time check against torch.rand
            if not verified(device):
                return None
            _CHECKED_DEVICES.add(device)
        gen = _GENS.get(device)
        if gen is None:
            dev = torch.device(device)
            gen = torch.cuda.default_generators[dev.index if dev.index is not None else torch.cuda.current_device()]
            if dev.index is not None:  # ("cuda" without an index follows the current device: not cached)
                _GENS[device] = gen
        off = gen.get_offset()
        if off & 3:
            return None
        gen.set_offset(off + 4)  # what the launch of a <= 256-element uniform kernel consumes (checked by the tests against torch.rand)
        self.seeds.append(gen.initial_seed())
        self.offsets.append(off)
        self.counts.append(numel)
        self._values = None
        return len(self.counts) - 1

    def resolve(self):
        """list of float32 numpy arrays, one per reserved draw"""
        if self._values is None:
            k = len(self.counts)
            total = sum(self.counts)
            out = np.empty(total, dtype=np.float32)
            if k:
                fn = _lib.api().lib.ffx_torch_rand_batch_h
                rc = fn(k, (C.c_uint64 * k)(*self.seeds), (C.c_uint64 * k)(*self.offsets), (C.c_int32 * k)(*self.counts), out.ctypes.data_as(C.POINTER(C.c_float)))
                if rc != 0:
                    raise RuntimeError("ffx_torch_rand_batch_h failed")
            vals, p = [], 0
            for n in self.counts:
                vals.append(out[p:p + n])
                p += n
            self._values = vals
        return self._values
