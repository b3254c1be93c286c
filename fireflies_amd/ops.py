"""Tensor-level wrappers over the C ABI (include/ffx.h) — plumbing only.

PyTorch is used for device memory and the current HIP stream; every arithmetic step happens in
libffx_hip.so.  All functions require contiguous tensors on a HIP device and raise otherwise.
"""
import ctypes as C
import os

import numpy as np
import torch

from . import _abi
from ._lib import api

REDUCE = {"sum": _abi.REDUCE_SUM, "softor": _abi.REDUCE_SOFTOR}


def _dev_index(device=None):
    if device is None:
        return torch.cuda.current_device()
    if isinstance(device, int):
        return device
    idx = torch.device(device).index
    return torch.cuda.current_device() if idx is None else idx


def _stream(device=None):
    """the caller's current HIP stream on `device` (default: the current device) as a raw handle.
    (torch.cuda.current_stream() builds a Stream object through several Python layers: 10 us per call,
    seven calls per render step; the raw query is one C call.)"""
    return C.c_void_p(torch._C._cuda_getCurrentRawStream(_dev_index(device)))


_STREAM_OBJECTS = {}


def _stream_obj(device=None):
    """torch Stream object of the caller's current stream (for event record / wait), cached by raw handle"""
    idx = _dev_index(device)
    raw = torch._C._cuda_getCurrentRawStream(idx)
    s = _STREAM_OBJECTS.get((idx, raw))
    if s is None:
        s = torch.cuda.current_stream(idx)
        if len(_STREAM_OBJECTS) > 64:
            _STREAM_OBJECTS.clear()
        _STREAM_OBJECTS[(idx, raw)] = s
    return s


def _dev(t, dtype=torch.float32, name="tensor"):
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a torch.Tensor")
    if not t.is_cuda:
        raise RuntimeError(f"{name}: must live on a HIP device (got {t.device}); fireflies_amd has no CPU path")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name}: must be contiguous")
    return C.c_void_p(t.data_ptr())


def _m16(m):
    if isinstance(m, torch.Tensor):
        m = m.detach().cpu().numpy()
    a = np.asarray(m, dtype=np.float32).reshape(-1)
    if a.size != 16:
        raise ValueError("expected a 4x4 matrix")
    return (C.c_float * 16)(*a.tolist())


def _reduce(reduce):
    if isinstance(reduce, str):
        return REDUCE[reduce]
    return int(reduce)


# ------------------------------------------------------------------ K1
def project_rays_fwd(rays, KF):
    out = torch.empty_like(rays)
    api().call("ffx_project_rays_fwd", _dev(rays, name="rays"), rays.shape[0], _m16(KF), _dev(out), _stream())
    return out


def project_rays_bwd(rays, KF, gpts):
    out = torch.empty_like(rays)
    api().call("ffx_project_rays_bwd", _dev(rays, name="rays"), rays.shape[0], _m16(KF), _dev(gpts, name="gpts"), _dev(out), _stream())
    return out


def transform_points(pts, M, mode=0):
    out = torch.empty_like(pts)
    api().call("ffx_transform_points", _dev(pts, name="pts"), pts.shape[0], _m16(M), mode, _dev(out), _stream())
    return out


def l1_value_grad(a, b, weight=1.0, acc=None):
    """weight * L1Loss(a, b) (a 0-dim view into the workspace) and its gradient with respect to `a`; acc (a float32 tensor): the value is
    also added to acc[0] by the same launches"""
    ws = torch.empty(257, dtype=torch.float32, device=a.device)
    g = torch.empty_like(a)
    if acc is not None:
        api().call("ffx_l1_value_grad_acc", _dev(a, name="a"), _dev(b, name="b"), a.numel(), float(weight), _dev(ws), _dev(g), _dev(acc.reshape(-1), name="acc"), _stream())
    else:
        api().call("ffx_l1_value_grad", _dev(a, name="a"), _dev(b, name="b"), a.numel(), float(weight), _dev(ws), _dev(g), _stream())
    return ws[0], g


def clamp_to_fov_(rays, KF, KF_inv, lo, hi, n_normalize=1):
    """in place: Laser.clamp_to_fov (+ n_normalize - 1 further normalisations) in one launch"""
    api().call("ffx_clamp_to_fov", _dev(rays, name="rays"), rays.shape[0], _m16(KF), _m16(KF_inv), float(lo), float(hi), int(n_normalize), _stream())
    return rays


# ------------------------------------------------------------------ fused pattern side of an optimisation step
def pattern_ws_floats(size0, size1):
    return int(api().lib.ffx_pattern_ws_floats(int(size0), int(size1)))


def pattern_fwd(rays, KF, sigma, size0, size1, want_softor=True, out=None, zero=None):
    """K1 + K2(sum) + K2(softor) + partial sums of L1(softor, sum) in one launch -> (pts [n,2], tsum, tsor, ws).
    `out`: optional tuple of tensors to reuse.  `zero`: a float32 tensor the same launch clears."""
    n = rays.shape[0]
    if out is None:
        pts = torch.empty((n, 2), dtype=torch.float32, device=rays.device)
        tsum = torch.empty((size1, size0), dtype=torch.float32, device=rays.device)
        tsor = torch.empty((size1, size0), dtype=torch.float32, device=rays.device) if want_softor else None
        ws = torch.empty(pattern_ws_floats(size0, size1), dtype=torch.float32, device=rays.device) if want_softor else None
    else:
        pts, tsum, tsor, ws = out
    api().call("ffx_pattern_fwd", _dev(rays, name="rays"), n, _m16(KF), float(sigma), int(size0), int(size1), int(bool(want_softor)), _dev(pts), _dev(tsum),
               _dev(tsor) if want_softor else None, _dev(ws) if want_softor else None, _dev(zero, name="zero") if zero is not None else None,
               int(zero.numel()) if zero is not None else 0, _stream())
    return pts, tsum, tsor, ws


def pattern_bwd(rays, KF, sigma, size0, size1, tsum, tsor, gts, reg_weight, ws, loss_in=None, loss_div=1.0):
    """K2-bwd of the data term (gts on the sum texture) and of reg_weight * L1(softor, sum), each through K1-bwd, in
    one launch -> (grays_data or None, grays_reg or None, [3] tensor: regulariser value, and — if `loss_in`, a tensor of
    partial sums of the data term (or one value), is given — sum(loss_in) / loss_div + regulariser, and sum(loss_in))"""
    n = rays.shape[0]
    gd = torch.empty((n, 3), dtype=torch.float32, device=rays.device) if gts is not None else None
    gr = torch.empty((n, 3), dtype=torch.float32, device=rays.device) if reg_weight > 0 else None
    val = torch.empty(3, dtype=torch.float32, device=rays.device)
    api().call("ffx_pattern_bwd", _dev(rays, name="rays"), n, _m16(KF), float(sigma), int(size0), int(size1), _dev(tsum, name="tsum"),
               _dev(tsor, name="tsor") if tsor is not None else None, _dev(gts, name="gts") if gts is not None else None, float(reg_weight),
               _dev(ws, name="ws") if ws is not None else None, _dev(gd) if gd is not None else None, _dev(gr) if gr is not None else None, _dev(val),
               _dev(loss_in, name="loss_in") if loss_in is not None else None, int(loss_in.numel()) if loss_in is not None else 0, float(loss_div), _stream())
    return gd, gr, val


def pattern_fwd_blur(rays, KF, sigma, size0, size1, ksize, blur_sigma, want_softor=True, out=None, zero=None):
    """pattern_fwd + tex = blur_fwd(tsum, ksize, blur_sigma) in one launch (ffx_pattern_fwd_blur) -> (pts, tsum, tsor, ws, tex); `out`: the five
    tensors to reuse."""
    n = rays.shape[0]
    if out is None:
        pts = torch.empty((n, 2), dtype=torch.float32, device=rays.device)
        tsum = torch.empty((size1, size0), dtype=torch.float32, device=rays.device)
        tsor = torch.empty((size1, size0), dtype=torch.float32, device=rays.device) if want_softor else None
        ws = torch.empty(pattern_ws_floats(size0, size1), dtype=torch.float32, device=rays.device) if want_softor else None
        tex = torch.empty((size1, size0), dtype=torch.float32, device=rays.device)
    else:
        pts, tsum, tsor, ws, tex = out
    api().call("ffx_pattern_fwd_blur", _dev(rays, name="rays"), n, _m16(KF), float(sigma), int(size0), int(size1), int(bool(want_softor)), _dev(pts), _dev(tsum),
               _dev(tsor) if want_softor else None, _dev(ws) if want_softor else None, _dev(zero, name="zero") if zero is not None else None,
               int(zero.numel()) if zero is not None else 0, int(ksize), float(blur_sigma), _dev(tex), _stream())
    return pts, tsum, tsor, ws, tex


def pattern_bwd_blur(rays, KF, sigma, size0, size1, tsum, tsor, gtex, reg_weight, ws, ksize, blur_sigma, loss_in=None, loss_div=1.0, adam=None, scratch=None):
    """pattern_bwd with K3^T in front (gtex: the gradient on the BLURRED texture; ksize 0: on tsum) and, with `adam` (an _abi.AdamArgs whose
    rays field names `rays`), the Adam + clamp_to_fov update behind it — one launch (ffx_pattern_bwd_blur).  -> (grays_data, grays_reg, [3])"""
    n = rays.shape[0]
    gd = torch.empty((n, 3), dtype=torch.float32, device=rays.device) if gtex is not None else None
    gr = torch.empty((n, 3), dtype=torch.float32, device=rays.device) if reg_weight > 0 else None
    val = torch.empty(3, dtype=torch.float32, device=rays.device)
    api().call("ffx_pattern_bwd_blur", _dev(rays, name="rays"), n, _m16(KF), float(sigma), int(size0), int(size1), _dev(tsum, name="tsum"),
               _dev(tsor, name="tsor") if tsor is not None else None, _dev(gtex, name="gtex") if gtex is not None else None, float(reg_weight),
               _dev(ws, name="ws") if ws is not None else None, _dev(gd) if gd is not None else None, _dev(gr) if gr is not None else None, _dev(val),
               _dev(loss_in, name="loss_in") if loss_in is not None else None, int(loss_in.numel()) if loss_in is not None else 0, float(loss_div),
               int(ksize), float(blur_sigma), _dev(scratch, name="scratch") if scratch is not None else None, C.byref(adam) if adam is not None else None, _stream())
    return gd, gr, val


_pattern_epochs = {}


_SHARED_STREAMS = {}


def shared_streams(device, role, n, priority=0):
    """the process's n streams of `role` on `device` (created on first use): every scene / geometry of the device takes the same ones"""
    dev = torch.device(device)
    key = (dev.type, dev.index if dev.index is not None else torch.cuda.current_device(), role, int(priority))
    got = _SHARED_STREAMS.setdefault(key, [])
    while len(got) < n:
        got.append(torch.cuda.Stream(dev, priority=int(priority)))
    return got[:n]


def pattern_step(rays, KF, sigma, size0, size1, bufs, gtex, reg_weight, ksize, blur_sigma, adam, zero, sync, rays_kept=None, check_kept=False, loss_in=None, loss_div=1.0,
                 epoch=None):
    """The pattern side of a step as ONE launch (ffx_pattern_step): pattern_bwd_blur(..., adam) on `bufs` = (pts, tsum, tsor, ws, tex) of THIS step,
    then — behind the update, in the same launch — pattern_fwd_blur of the NEXT step into the same five tensors, `zero` cleared.
    -> (grays_data, grays_reg, [3] loss values), or None when the library declines the shape (FFX_ERR_UNSUPPORTED: the caller issues the two launches).
    `sync`: uint8 tensor of _abi.PATTERN_SYNC_BYTES, zero before the first call; rays_kept: float32 [2, n, 3]; epoch: 1, 2, 3, ... per launch on this pair of
    buffers (None: counted here)."""
    pts, tsum, tsor, ws, tex = bufs
    n = rays.shape[0]
    gd = torch.empty((n, 3), dtype=torch.float32, device=rays.device) if gtex is not None else None
    gr = torch.empty((n, 3), dtype=torch.float32, device=rays.device) if reg_weight > 0 else None
    val = torch.empty(3, dtype=torch.float32, device=rays.device)
    if sync.numel() < _abi.PATTERN_SYNC_BYTES or sync.dtype != torch.uint8:
        raise ValueError("pattern_step: sync must be a uint8 tensor of PATTERN_SYNC_BYTES")
    if rays_kept is None or rays_kept.numel() != 6 * n:
        raise ValueError("pattern_step: rays_kept must be a float32 tensor [2, n, 3]")
    # the launch's epoch: the caller's count of the launches on this (sync, rays_kept) pair — or, for callers that keep none, a count kept here by the
    # buffers' addresses (a pair that was zeroed and is used again under the same addresses continues the old count: any value but the flags' works)
    key = None
    if epoch is None:
        key = (sync.data_ptr(), rays_kept.data_ptr())
        epoch = _pattern_epochs.get(key, 0) % 0xFFFFFFF0 + 1
        if len(_pattern_epochs) > 256 and key not in _pattern_epochs:
            _pattern_epochs.clear()
    rc = api().call_rc("ffx_pattern_step", _dev(rays, name="rays"), n, _m16(KF), float(sigma), int(size0), int(size1), _dev(tsum, name="tsum"),
                       _dev(tsor, name="tsor") if tsor is not None else None, _dev(gtex, name="gtex") if gtex is not None else None, float(reg_weight),
                       _dev(ws, name="ws") if ws is not None else None, _dev(gd) if gd is not None else None, _dev(gr) if gr is not None else None, _dev(val),
                       _dev(loss_in, name="loss_in") if loss_in is not None else None, int(loss_in.numel()) if loss_in is not None else 0, float(loss_div),
                       int(ksize), float(blur_sigma), C.byref(adam), _dev(pts, name="pts"), _dev(zero, name="zero") if zero is not None else None,
                       int(zero.numel()) if zero is not None else 0, _dev(tex, name="tex"), _dev(rays_kept, name="rays_kept"),
                       int(bool(check_kept)), _dev(sync, torch.uint8, "sync"), epoch, _stream(), allow=(_abi.FFX_ERR_UNSUPPORTED,))
    if rc != 0:
        return None
    if key is not None:
        _pattern_epochs[key] = epoch
    return gd, gr, val


def adam_args(rays, exp_avg, exp_avg_sq, step, counter, lr, beta1, beta2, eps, KF_inv, lo, hi, n_normalize=1, grad_div=1.0, grad_out=None, dot=None, guard=None):
    """ffx_adam_args for pattern_bwd_blur (the tensors must outlive the launch; `counter`: one zeroed int32 / uint32 device word).
    dot = (a, b, partial): the launch also evaluates <a, b> (two float32 tensors of equal size: the render and the constant gradient of a loss
    linear in it) as the step's data term; partial: float32 scratch of one element per point."""
    a = _abi.AdamArgs()
    a.rays = _dev(rays, name="rays").value
    if exp_avg is not None:  # (exp_avg = exp_avg_sq = step = None: no update, only the inner product — a multi-rank step)
        a.exp_avg, a.exp_avg_sq = _dev(exp_avg, name="exp_avg").value, _dev(exp_avg_sq, name="exp_avg_sq").value
        a.step = _dev(step, name="step").value
    a.grad_out = _dev(grad_out, name="grad_out").value if grad_out is not None else None
    a.counter = _dev(counter, torch.int32, "counter").value
    a.lr, a.beta1, a.beta2, a.eps = float(lr), float(beta1), float(beta2), float(eps)
    a.KF_inv = _m16(KF_inv)
    a.lo, a.hi, a.grad_div, a.n_normalize = float(lo), float(hi), float(grad_div), int(n_normalize)
    if dot is not None:
        da, db, part = dot
        if da.numel() % db.numel() != 0 or part.numel() < rays.shape[0]:
            raise ValueError("dot = (a, b, partial): a a whole number of b's long (b is repeated), partial with one float per point")
        a.dot_a, a.dot_b, a.dot_n, a.dot_partial = _dev(da, name="dot a").value, _dev(db, name="dot b").value, int(da.numel()), _dev(part, name="dot partial").value
        a.dot_b_n = int(db.numel())
    if guard is not None:  # an adjoint cache (uint8 tensor): the update is skipped when its header reports dropped samples
        a.guard = _dev(guard, torch.uint8, "guard").value
    # the struct holds raw addresses: it keeps the tensors alive for as long as it lives itself.  (A temporary passed as grad_out used to be
    # freed on return — the caching allocator then handed its block to the NEXT small torch.empty, e.g. pattern_bwd_blur's 3-float value
    # buffer, which the update's gradient store overwrote: an intermittent wrong regulariser value in the test suite, round 4)
    a._keep = (rays, exp_avg, exp_avg_sq, step, counter, grad_out, dot, guard)
    return a


def adam_clamp_step_(rays, grad, exp_avg, exp_avg_sq, step, lr, beta1, beta2, eps, KF, KF_inv, lo, hi, n_normalize=1, grad_b=None, grad_div=1.0, grad_out=None, guard=None):
    """in place: torch.optim.Adam's update of `rays` followed by Laser.clamp_to_fov + normalisation, one launch.
    With grad_out the gradient used is grad / grad_div + grad_b (stored in grad_out).
    guard: a device tensor (or a slice of one) whose 32-bit word at byte 8 says "skip": an adjoint cache header, or the tail of an
    all-reduced flat buffer (include/ffx.h)."""
    if guard is not None and (not guard.is_cuda or guard.numel() * guard.element_size() < 12):
        raise ValueError("guard must be a device tensor of at least 12 bytes")
    api().call("ffx_adam_clamp_step", _dev(rays, name="rays"), _dev(grad, name="grad"), _dev(grad_b, name="grad_b") if grad_b is not None else None, float(grad_div),
               _dev(grad_out, name="grad_out") if grad_out is not None else None, _dev(exp_avg, name="exp_avg"), _dev(exp_avg_sq, name="exp_avg_sq"),
               _dev(step, name="step"), rays.shape[0], float(lr), float(beta1), float(beta2), float(eps), _m16(KF), _m16(KF_inv), float(lo), float(hi),
               int(n_normalize), guard.data_ptr() if guard is not None else None, _stream())
    return rays


# ------------------------------------------------------------------ K2
def _check_pts(pts):
    if pts.dim() != 2 or pts.shape[1] != 2:
        raise ValueError(f"points must be [N,2], got {tuple(pts.shape)}")


def splat_dense_fwd(pts, sigma, size0, size1):
    _check_pts(pts)
    out = torch.empty((pts.shape[0], size1, size0), dtype=torch.float32, device=pts.device)
    api().call("ffx_splat_dense_fwd", _dev(pts, name="pts"), pts.shape[0], float(sigma), size0, size1, _dev(out), _stream())
    return out


def splat_dense_bwd(pts, sigma, size0, size1, gout):
    _check_pts(pts)
    if tuple(gout.shape) != (pts.shape[0], size1, size0):
        raise ValueError("gout shape mismatch")
    out = torch.empty_like(pts)
    api().call("ffx_splat_dense_bwd", _dev(pts, name="pts"), pts.shape[0], float(sigma), size0, size1, _dev(gout, name="gout"), _dev(out), _stream())
    return out


def splat_fwd(pts, sigma, reduce, half_window, size0, size1):
    _check_pts(pts)
    out = torch.empty((size1, size0), dtype=torch.float32, device=pts.device)
    api().call("ffx_splat_fwd", _dev(pts, name="pts"), pts.shape[0], float(sigma), _reduce(reduce), int(half_window), size0, size1, _dev(out), _stream())
    return out


def splat_bwd(pts, sigma, reduce, half_window, size0, size1, tex, gtex):
    _check_pts(pts)
    if tuple(gtex.shape) != (size1, size0):
        raise ValueError("gtex shape mismatch")
    out = torch.empty_like(pts)
    api().call(
        "ffx_splat_bwd", _dev(pts, name="pts"), pts.shape[0], float(sigma), _reduce(reduce), int(half_window), size0, size1,
        _dev(tex, name="tex") if tex is not None else None, _dev(gtex, name="gtex"), _dev(out), _stream(),
    )
    return out


def splat_depth_fwd(pts, depth, sigma, size0, size1):
    _check_pts(pts)
    depth = depth.reshape(-1)
    out = torch.empty((pts.shape[0], size1, size0), dtype=torch.float32, device=pts.device)
    api().call("ffx_splat_depth_fwd", _dev(pts, name="pts"), _dev(depth, name="depth"), pts.shape[0], float(sigma), size0, size1, _dev(out), _stream())
    return out


def splat_lines_fwd(lines, sigma, size0, size1):
    if lines.dim() != 3 or tuple(lines.shape[1:]) != (2, 2):
        raise ValueError("lines must be [N,2,2]")
    out = torch.empty((lines.shape[0], size1, size0), dtype=torch.float32, device=lines.device)
    api().call("ffx_splat_lines_fwd", _dev(lines, name="lines"), lines.shape[0], float(sigma), size0, size1, _dev(out), _stream())
    return out


def splat_lines_bwd(lines, sigma, size0, size1, gout):
    if tuple(gout.shape) != (lines.shape[0], size1, size0):
        raise ValueError("gout shape mismatch")
    out = torch.empty_like(lines)
    api().call("ffx_splat_lines_bwd", _dev(lines, name="lines"), lines.shape[0], float(sigma), size0, size1, _dev(gout, name="gout"), _dev(out), _stream())
    return out


# ------------------------------------------------------------------ K3
def blur_fwd(img, ksize=5, sigma=3.0):
    if img.dim() != 2:
        raise ValueError("blur expects [H,W]")
    out = torch.empty_like(img)
    api().call("ffx_blur_fwd", _dev(img, name="img"), img.shape[0], img.shape[1], ksize, float(sigma), _dev(out), _stream())
    return out


def silhouette(img, cx, cy, radius, ksize=11, sigma=5.0):
    """img [H,W] times the blurred filled circle of apply_silhouette.py — ffx_silhouette_fwd: blur_fwd(mask) * img bit for bit, one launch, no mask tensor"""
    if img.dim() != 2:
        raise ValueError("silhouette expects [H,W]")
    out = torch.empty_like(img)
    api().call("ffx_silhouette_fwd", _dev(img, name="img"), img.shape[0], img.shape[1], int(cx), int(cy), int(radius), int(ksize), float(sigma), _dev(out), _stream())
    return out


def noise_clamp(img, noise, mean, std, lo=0.0, hi=1.0):
    """clamp(img + (noise * std + mean), lo, hi) in one launch (ffx_noise_clamp); the result takes the noise tensor's place"""
    if img.shape != noise.shape:
        raise ValueError("noise must have the image's shape")
    api().call("ffx_noise_clamp", _dev(img, name="img"), _dev(noise, name="noise"), img.numel(), float(mean), float(std), float(lo), float(hi), _dev(noise), _stream())
    return noise


def rgb_to_gray(img, weights=(0.299, 0.587, 0.114)):
    """[..., 3] float32 / float16 -> [...] float32: (r wr + g wg) + b wb (ffx_rgb_to_gray; cv2.COLOR_RGB2GRAY's weights by default)"""
    if img.shape[-1] != 3 or img.dtype not in (torch.float32, torch.float16):
        raise ValueError("rgb_to_gray expects a [..., 3] float32 or float16 image")
    out = torch.empty(img.shape[:-1], dtype=torch.float32, device=img.device)
    api().call("ffx_rgb_to_gray", _dev(img, img.dtype, "img"), int(img.dtype == torch.float16), out.numel(), float(weights[0]), float(weights[1]), float(weights[2]), _dev(out), _stream())
    return out


def blur_bwd(g, ksize=5, sigma=3.0):
    out = torch.empty_like(g)
    api().call("ffx_blur_bwd", _dev(g, name="g"), g.shape[0], g.shape[1], ksize, float(sigma), _dev(out), _stream())
    return out


def render_cache_bytes(width, height, spp):
    return int(api().lib.ffx_render_cache_bytes(int(width), int(height), int(spp)))


def render_cache_bytes_sd(sd, spp):
    """the adjoint cache of a render of `sd` (larger with material rows: a second footprint per pixel)"""
    return int(api().lib.ffx_render_cache_bytes_sd(C.byref(sd), int(spp)))


def render_filter_bytes(sd):
    """scratch of ffx_render_{fwd,bwd}_filtered for a render of `sd` (400 + 16 bytes per pixel)"""
    return int(api().lib.ffx_render_filter_bytes(C.byref(sd)))


def render_dot_slots(width, height):
    return int(api().lib.ffx_render_dot_slots(int(width), int(height)))


def render_cache_status(cache):
    """(stray records used, arena capacity, dropped samples) of an adjoint cache written by render_fwd(..., cache=...) — for a filtered film's
    cache: (64-sample blocks taken, blocks the arena holds, pixels that found it full and kept no records).
    SYNCHRONISES the current stream (64-byte read).  dropped > 0: the cache is incomplete — ffx_render_bwd_cached then
    poisons gtex[0] with NaN; use the re-tracing adjoint."""
    out = (C.c_uint32 * 3)()
    api().call("ffx_render_cache_status", _dev(cache, torch.uint8, "cache"), out, _stream())
    return int(out[0]), int(out[1]), int(out[2])


def _check_materials(sd, albedo):
    """-> the pointer argument for shape_albedo: None when the scene description carries the rows itself (sd.n_mat_h)"""
    if sd.n_mat_h > 0:
        return None
    if albedo is None:
        raise ValueError("no material table: pass the device tensor or put the rows into the scene description (scene_desc.set_host_materials)")
    ms = int(sd.mat_stride) or 3
    if albedo.dim() != 2 or albedo.shape[1] != ms or albedo.shape[0] < sd.n_shapes:
        raise ValueError(f"material table {tuple(albedo.shape)} does not match the scene description (n_shapes {sd.n_shapes}, mat_stride {ms})")
    return _dev(albedo, name="albedo")


# ------------------------------------------------------------------ K5..K9
def camera_struct(to_world, camera_to_sample, near, far, width, height):
    c = _abi.Camera()
    c.to_world = _m16(to_world)
    c.camera_to_sample = _m16(camera_to_sample)
    c.near_clip, c.far_clip, c.width, c.height = float(near), float(far), int(width), int(height)
    return c


class _EventPair:
    """records a HIP event pair on the current stream around one launch (bench.py's live kernel
    timing); a no-op when `sink` is None."""

    def __init__(self, sink, name):
        self.sink, self.name = sink, name

    def __enter__(self):
        if self.sink is not None:
            self.a = torch.cuda.Event(enable_timing=True)
            self.b = torch.cuda.Event(enable_timing=True)
            self.a.record()
        return self

    def __exit__(self, *exc):
        if self.sink is not None:
            self.b.record()
            self.sink.append((self.name, self.a, self.b))
        return False


# the pre-pass knobs are read per call (a test flips them mid-process) — straight from os.environ's own dict where it has one (posix: bytes -> bytes;
# os.environ.get's encode + KeyError path was 0.6 us a look-up, fourteen of them per loop step: a tenth of the host's time at 4 spp)
_ENV_DATA = getattr(os.environ, "_data", None)
_ENV_NAMES = ("FFX_BINS", "FFX_BIN_TILE", "FFX_BIN_TILE_PROJ", "FFX_BIN_SPOT_N", "FFX_BIN_CAP", "FFX_SHADOW_CLEAR", "FFX_ENVELOPE")
if isinstance(_ENV_DATA, dict) and os.name == "posix":
    _ENV_KEYS = tuple(n.encode() for n in _ENV_NAMES)

    def _apex_env():
        g = _ENV_DATA.get
        return tuple(g(k) for k in _ENV_KEYS)
else:
    def _apex_env():
        g = os.environ.get
        return tuple(g(n) for n in _ENV_NAMES)


def apex_key(sd=None, cam=None):
    """what the blob's pre-pass areas must hold for a packet render / trace call (include/ffx.h ffx_apex_prepare): the apex records depend
    on the POSITIONS of the camera and of the enabled emitters, the tile bins (ffx_bvh_info.off_bins) on their whole projections — pose,
    field of view, film / texture size, the spot's cone — as the exact floats the library reads"""
    if sd is not None:
        k = getattr(sd, "_apex_key", None)  # (a description is never modified once built: mi.Scene makes a new one per pose — 5 us per call otherwise)
        if k is not None and k[1] == _apex_env():
            return k[0]
    c = sd.cam if sd is not None else cam
    # (the structs' bytes: ~1 us each — tuples of their 32 floats were 10 us per render call)
    key = [C.string_at(C.addressof(c), C.sizeof(c))]
    if sd is not None:
        key += [C.string_at(C.addressof(sd.proj), C.sizeof(sd.proj)) if sd.proj.enabled else None]
        # (not the spot's intensity: randomised per step, no part of the pre-pass; `shadows`: the pre-pass proves the "clear" triangles only for a scene that traces shadow rays)
        key += [(C.string_at(C.addressof(sd.spot), 64), sd.spot.cutoff_deg, int(sd.shadows)) if sd.spot.enabled else (None, int(sd.shadows))]
    else:
        key += [None, None]
    env = _apex_env()
    key = tuple(key) + env
    if sd is not None and getattr(sd, "_frozen", False):  # (only descriptions whose maker promises not to touch them again: mi.Scene.scene_desc)
        sd._apex_key = (key, env)
    return key


def _camera_part(key):
    """of an apex_key: what the camera's area (its apex records and tile bins) depends on"""
    return None if key is None else (key[0], key[3], key[4], key[7])  # (camera struct, FFX_BINS, FFX_BIN_TILE, FFX_BIN_CAP)


def deterministic_mode():
    """FFX_DETERMINISTIC=1: every texture gradient comes from ffx_render_bwd_det (64-bit fixed-point accumulation: bitwise reproducible, whatever
    the dispatch order or the number of ranks) instead of the float-atomic adjoints — functional.render's backward and PatternOptimizer.step
    then re-trace.  A cross-checking / debugging mode: two re-traces and one host synchronisation per adjoint."""
    return os.environ.get("FFX_DETERMINISTIC", "0") == "1"


def det_scale_log2(vmax_bits: int, n_taps: int):
    """the power-of-two scale of a deterministic accumulation whose largest |tap| has the float bits `vmax_bits`, over `n_taps` taps in all
    (include/ffx.h ffx_det_scale_log2) — None: nothing lit, or a non-finite tap"""
    sh = int(api().lib.ffx_det_scale_log2(C.c_uint32(int(vmax_bits) & 0xFFFFFFFF), C.c_uint64(int(n_taps))))
    return None if sh < -126 else sh


def det_finish_(acc: torch.Tensor, scale_log2: int, gtex: torch.Tensor):
    """gtex += acc * 2^-scale_log2 (acc: int64 fixed-point sums, one per texel and channel)"""
    if acc.dtype != torch.int64 or not acc.is_contiguous() or gtex.dtype != torch.float32 or not gtex.is_contiguous() or acc.numel() != gtex.numel():
        raise ValueError("det_finish_: int64 sums and a float32 gradient of the same size, both contiguous")
    api().call("ffx_det_finish", C.c_void_p(acc.data_ptr()), int(scale_log2), acc.numel(), C.c_void_p(gtex.data_ptr()), _stream())
    return gtex


def _lane_kernels():
    return os.environ.get("FFX_TRAVERSAL") == "lane"  # (the per-lane A/B kernels neither read nor write apex records)


class DeviceGeometry:
    """Triangle soup + BVH blob resident in HBM.

    src_verts [P,3] : vertex pool (all shapes, all animation frames)
    tris [F,3]      : shape-local vertex indices
    tri_shape [F]   : shape id per triangle
    vert_off [S]    : pool offset of each shape's current frame
    The topology is built once on the host (ffx_bvh_build_host) from `build_verts` (defaults to the
    pool at the given offsets); `update()` is the per-randomisation device pass (K5+K6).
    """

    def __init__(self, src_verts, tris, tri_shape, vert_off, device="cuda", build_xforms=None, smooth=None):
        """smooth: one flag per shape — interpolated shading normals, re-derived from the posed vertices by every update()
        (include/ffx.h ffx_smooth: what Mitsuba does for meshes that carry vertex normals)."""
        src = np.ascontiguousarray(src_verts, dtype=np.float32).reshape(-1, 3)
        tr = np.ascontiguousarray(tris, dtype=np.int32).reshape(-1, 3)
        ts = np.ascontiguousarray(tri_shape, dtype=np.int32).reshape(-1)
        vo = np.ascontiguousarray(vert_off, dtype=np.int32).reshape(-1)
        F, S = tr.shape[0], vo.shape[0]
        if ts.shape[0] != F:
            raise ValueError("tri_shape must have one entry per triangle")
        if F < 1 or S < 1:
            raise ValueError("need at least one triangle and one shape")
        if ts.min() < 0 or ts.max() >= S:
            raise ValueError("tri_shape out of range")
        if tr.min() < 0:
            raise ValueError("negative vertex index")
        self.n_tris, self.n_shapes = F, S
        self.device = torch.device(device)
        self._didx = _dev_index(self.device) if self.device.type == "cuda" and torch.cuda.is_available() else None
        self.timing = None  # set to a list to collect (name, start_event, end_event) per launch
        self._max_local = np.zeros(S, np.int64)
        np.maximum.at(self._max_local, ts, tr.max(axis=1))
        self._pool_size = src.shape[0]
        self._check_offsets(vo)
        # host build from the pose given by the current offsets (+ optional per-shape transforms)
        glob = (tr + vo[ts][:, None]).astype(np.int32)
        bverts = src
        if build_xforms is not None:
            bx = np.asarray(build_xforms, dtype=np.float32).reshape(S, 4, 4)
            bverts = src.copy()
            used = np.unique(glob)
            owner = np.zeros(src.shape[0], np.int64)
            owner[glob.reshape(-1)] = np.repeat(ts, 3)
            v = src[used]
            m = bx[owner[used]]
            bverts[used] = np.einsum("nij,nj->ni", m[:, :3, :3], v) + m[:, :3, 3]
        a = api()
        nbytes = a.lib.ffx_bvh_blob_bytes(F)
        blob = np.zeros(nbytes, np.uint8)
        self.info = _abi.BvhInfo()
        a.call("ffx_bvh_build_host", bverts.ctypes.data, bverts.shape[0], glob.ctypes.data, F, blob.ctypes.data, nbytes, C.byref(self.info))
        # Two copies of the blob: update() re-fits the one that is NOT being read, on a side stream, while
        # the render kernels of the previous step are still running on the caller's stream (the refit is
        # five small dependent launches, ~55 us with the GPU otherwise idle; overlapped it costs nothing).
        # FFX_ASYNC_UPDATE=0 falls back to one blob, everything on the caller's stream.
        # the host builder fills the first off_bins bytes; the tail (tile bins, apex records) is scratch of the render calls' pre-pass
        static = int(self.info.off_bins) if int(self.info.off_bins) > 0 else int(self.info.total_bytes)
        b0 = torch.zeros(int(self.info.total_bytes), dtype=torch.uint8, device=self.device)
        b0[:static].copy_(torch.from_numpy(blob[:static]))
        self._async = self.device.type == "cuda" and os.environ.get("FFX_ASYNC_UPDATE", "1") != "0"
        # (FFX_BLOB_COPIES >= 2: measured 2 / 3 / 4 copies = 2 050 / 2 043 / 2 038 renders/s — a longer window for the side stream's work buys
        # nothing: the GPU is busy either way, the re-fit and the pre-pass cost what they cost)
        # (round 5: four copies, of which renders at 33 samples per pixel and more use two — see _ring)
        n_copies = max(2, int(os.environ.get("FFX_BLOB_COPIES", "4"))) if self._async else 1
        self._blobs = [b0] + [b0.clone() for _ in range(n_copies - 1)]
        self._cur = 0
        # (FFX_SIDE_PRIORITY: -1 = a high-priority queue for the side stream; measured: renders/s unchanged, gradient steps 2 130 -> 1 540 per
        # second — the step's small launches on the main stream then wait behind it.  0 = default)
        # (round 6: ONE side stream — and one pair of render streams, shared_streams below — per device and process, not per geometry: the streams of a
        # scene that was used a moment ago keep their hardware queues, and a second scene's own streams then share what is left — its consecutive renders
        # lost their overlap: configs[4] behind a two-stream bracket of the vocal fold, 125 -> 118 renders/s, K8 alone unchanged)
        self._side = shared_streams(self.device, "side", 1, int(os.environ.get("FFX_SIDE_PRIORITY", "0")))[0] if self._async else None
        # (round 6, a measured negative result kept as a knob) FFX_SIDE_STREAMS=n: a side stream per blob copy, so that the chains of consecutive
        # poses — re-fit -> count -> scan -> fill [-> envelopes], five or six dependent launches of mostly latency — overlap each other instead of
        # queueing on one stream (each writes its own copy; what they share is read-only).  With 4 streams the loops LOSE a quarter (principled
        # 64 spp 2 890 -> 2 208 renders/s, diffuse 3 406 -> 2 506, 16 spp 5 516 -> 3 849): more side work in flight only takes issue slots and
        # dispatch turns from the render that the chain then waits for.  Default 1: the one stream of rounds 1-5.
        n_side = max(1, min(n_copies, int(os.environ.get("FFX_SIDE_STREAMS", "1")))) if self._async else 0
        self._sides = [self._side] + [torch.cuda.Stream(self.device, priority=int(os.environ.get("FFX_SIDE_PRIORITY", "0"))) for _ in range(n_side - 1)] if self._async else []
        self._side_handles = {}
        self._last_spp = 64  # samples per pixel of the last render call (how deep update() lets the poses run ahead: _ring)
        self._upd_done = [None] * n_copies   # event: the refit of blob i has been enqueued up to here (side stream)
        self._top = [None] * n_copies        # the top of blob i's tree: None = in place; True = owed (FFX_STEP_DEFER_TOP); a stream handle = launched there (_top_ev)
        self._top_ev = [None] * n_copies
        # (measured, 100-step render loops of the vocal fold: 1 / 4 / 8 spp +11 / +17 / +16 % with the top deferred, 10 / 12 / 16 spp -4 / -2 / -4 %, 64 spp
        # unchanged — a render of a few samples waits for the chain, a longer one for the GPU, where the extra launch in front of it only costs: deferred
        # while the last render had at most FFX_DEFER_TOP samples per pixel, default 8; 0: never)
        self._defer_top_spp = int(os.environ.get("FFX_DEFER_TOP", "8"))
        self._apex = [None] * n_copies       # apex_key of what blob i's apex areas hold (None: nothing usable)
        self._apex_written = [None] * n_copies  # (stream handle, event) behind the call whose own pre-pass last wrote blob i's apex areas
        self._apex_writer_pending = False
        self._last_use = [{} for _ in range(n_copies)]       # per blob: stream handle -> event behind that stream's last reader (renders may come from several streams)
        self.src_verts = torch.from_numpy(src).to(self.device)
        self.tris = torch.from_numpy(tr).to(self.device)
        self.tri_shape = torch.from_numpy(ts).to(self.device)
        self.vert_off = torch.from_numpy(vo).to(self.device)
        self._vert_off_host = vo.copy()
        self._vert_off_dev_stale = False
        self._pool_written = None  # event: the last write_verts() into the vertex pool (caller's stream)
        self.version = 0           # bumped by every update(): functional._Render pins the pose it traced
        self._smooth = None
        if smooth is not None and any(smooth):
            from . import scenes as _scenes

            if S > 32:
                raise NotImplementedError("interpolated normals: at most 32 shapes (the per-shape tables travel as kernel arguments)")
            flags, vbase, a0, adj, n_vn = _scenes.smooth_tables(tr, ts, smooth, S)
            keep = {"flags": np.ascontiguousarray(flags), "vbase": np.ascontiguousarray(vbase), "adj_start": torch.from_numpy(a0).to(self.device),
                    "adj": torch.from_numpy(adj if adj.size else np.zeros(1, np.int32)).to(self.device),
                    "vn": torch.zeros((n_vn, 3), dtype=torch.float32, device=self.device)}  # scratch of the update (side stream)
            sm = _abi.Smooth()
            sm.shape_smooth = keep["flags"].ctypes.data_as(C.POINTER(C.c_int32))
            sm.shape_vbase = keep["vbase"].ctypes.data_as(C.POINTER(C.c_int32))
            sm.adj_start, sm.adj, sm.n_vn, sm.vnormals = keep["adj_start"].data_ptr(), keep["adj"].data_ptr(), int(n_vn), keep["vn"].data_ptr()
            self._smooth = (sm, keep)
        self.smooth = [bool(f) for f in smooth] if smooth is not None else [False] * S
        if self._async:  # blob copies / uploads above were enqueued on the caller's stream
            if self._smooth is not None:
                self._sides = self._sides[:1]
            for s_ in self._sides:
                s_.wait_stream(_stream_obj(self._didx))
        self.update(torch.eye(4, dtype=torch.float32).repeat(S, 1, 1) if build_xforms is None else torch.from_numpy(np.asarray(build_xforms, np.float32)))

    @property
    def blob(self):
        """the blob the next render / trace call will read (ordered after its refit on the current stream)"""
        self._acquire()
        return self._blobs[self._cur]

    def _acquire(self):
        ev = self._upd_done[self._cur] if self._async else None
        if ev is not None:
            _stream_obj(self._didx).wait_event(ev)
        if self._async:
            # (round 6) a natively pushed pose leaves the TOP of its tree to the first call that walks it (FFX_STEP_DEFER_TOP: the chain a short render
            # waits for is one dependent launch shorter; the top — one workgroup — runs here, on the reader's own stream, beside the previous render).
            # A second reader on another stream waits for that launch.
            i = self._cur
            tp = self._top[i]
            if tp is True:
                so = _stream_obj(self._didx)
                self._call("ffx_scene_refit_top", _dev(self._blobs[i], torch.uint8, "blob"), C.byref(self.info), _stream(self._didx))
                e = self._top_ev[i]
                if e is None:
                    e = self._top_ev[i] = torch.cuda.Event()
                e.record(so)
                self._top[i] = so.cuda_stream
            elif tp is not None:
                so = _stream_obj(self._didx)
                if tp != so.cuda_stream:
                    so.wait_event(self._top_ev[i])

    def _release(self):
        if self.device.type == "cuda":
            so = _stream_obj(self._didx)
            d = self._last_use[self._cur if self._async else 0]
            ev = d.get(so.cuda_stream)
            if ev is None:
                ev = d[so.cuda_stream] = torch.cuda.Event()
            ev.record(so)
            if self._apex_writer_pending:  # this call wrote the apex areas itself: a render on ANOTHER stream that finds them "ready" waits for it
                self._apex_written[self._cur if self._async else 0] = (so.cuda_stream, ev)
                self._apex_writer_pending = False

    def _ring(self):
        """how many of the blob copies the next update() rotates through.  A render at 64 spp (0.39 ms) hides the whole side chain of the next pose
        (re-fit -> count -> scan -> fill, ~0.18 ms beside it) behind itself with two copies — more only cost (2 050 / 2 043 / 2 038 renders/s with 2 / 3 / 4).
        Below 33 spp the render is as short as the chain, and with two copies the chain of pose i + 2 has to wait for the render of pose i: 5 435 /
        5 735 / 6 002 renders/s at 1 spp, 5 107 / 5 384 / 5 602 at 10 spp with 2 / 3 / 4 copies."""
        n = len(self._blobs)
        if len(self._sides) > 1:  # (FFX_SIDE_STREAMS > 1: the chains of up to n - 1 poses ahead may overlap each other and the renders)
            return n if self._last_spp <= 32 else min(n, max(2, int(os.environ.get("FFX_RING_LONG", "3"))))
        return n if self._last_spp <= 32 else min(n, 2)

    def _wait_readers(self, i, stream_obj):
        """`stream_obj` is about to WRITE into blob i (a re-fit, an apex pre-pass): every reader on another stream must be done"""
        for handle, ev in self._last_use[i].items():
            if handle != stream_obj.cuda_stream:
                stream_obj.wait_event(ev)

    def write_verts(self, offset, verts):
        """copy caller-supplied vertices [V,3] into the pool at `offset` (animation functions, direct
        `vertex_positions` assignment).  Ordered against the refits on the side stream in both
        directions: the copy waits for every enqueued refit (one may still be reading this slot), and
        the next update() waits for the copy."""
        main = _stream_obj(self._didx)
        if self._async:
            for ev in self._upd_done:
                if ev is not None:
                    main.wait_event(ev)
        self.src_verts[offset : offset + verts.shape[0]].copy_(verts)
        if self._async:
            if self._pool_written is None:
                self._pool_written = torch.cuda.Event()
            self._pool_written.record(main)

    def _call(self, name, *args):
        """a launch on this geometry's device (which need not be the current one)"""
        if self.device.index is None or torch._C._cuda_getDevice() == self.device.index:
            return api().call(name, *args)
        with torch.cuda.device(self.device):
            return api().call(name, *args)

    def _check_offsets(self, vo):
        if (vo < 0).any() or ((vo.astype(np.int64) + self._max_local) >= self._pool_size).any():
            raise ValueError("vert_off + triangle index exceeds the vertex pool")

    def update(self, xforms, vert_off=None, apex_sd=None):
        """K5+K6.  xforms [S,4,4]; optional new frame offsets [S] (host ints).
        apex_sd: the scene description the next renders will use — its apex records are written right behind the re-fit (on the
        side stream, off the renders' critical path) and those renders are told FFX_RENDER_APEX_READY.
        Host xforms (CPU tensor / ndarray) with S <= 32 go through ffx_scene_update_h: the tables
        are kernel arguments, nothing is copied to the device and the call never blocks.  A device
        tensor of xforms uses ffx_scene_update (device-resident randomisers)."""
        if vert_off is not None:
            vo = np.ascontiguousarray(vert_off, dtype=np.int32).reshape(-1)
            if vo.shape[0] != self.n_shapes:
                raise ValueError("vert_off must have one entry per shape")
            self._check_offsets(vo)
            self._vert_off_host = vo.copy()
            self._vert_off_dev_stale = True
        on_device = isinstance(xforms, torch.Tensor) and xforms.is_cuda
        self.version += 1
        if not self._async:
            if self.device.type == "cuda":
                self._wait_readers(0, _stream_obj(self._didx))
            self._update_into(self._blobs[0], xforms, on_device)
            self._prepare_apex(0, apex_sd)
            return
        nxt = (self._cur + 1) % self._ring()
        main = _stream_obj(self._didx)
        side = self._side_of(nxt)
        self._wait_readers(nxt, side)  # its readers must be done before it is overwritten
        if self._pool_written is not None:
            side.wait_event(self._pool_written)  # caller-supplied vertices must have landed in the pool
        if on_device:
            side.wait_stream(main)  # the tables were produced on the caller's stream
            xforms.record_stream(side)
        ev = self._upd_done[nxt]
        if ev is None:
            ev = self._upd_done[nxt] = torch.cuda.Event()
        self._top[nxt] = None  # (this path re-fits the whole tree)
        if self.timing is None and not on_device:
            # (the two launch calls take the side stream's handle as an argument: making it the current stream first — a context manager and
            # two current-stream queries — cost 10 us of the 65 this method took per step; the timed / device-table paths keep that form)
            sh = self._side_handle_of(side)
            self._update_into(self._blobs[nxt], xforms, on_device, sh)
            self._prepare_apex(nxt, apex_sd, sh)
            ev.record(side)
        else:
            with torch.cuda.stream(side):
                self._update_into(self._blobs[nxt], xforms, on_device)
                self._prepare_apex(nxt, apex_sd)
                ev.record(side)
        self._cur = nxt

    def _side_of(self, i):
        """the side stream that re-fits blob copy i (one per copy: the poses' chains overlap; a copy is always written from the same stream)"""
        return self._sides[i % len(self._sides)]

    def _side_handle_of(self, side):
        h = self._side_handles.get(side.cuda_stream)
        if h is None:
            h = self._side_handles[side.cuda_stream] = C.c_void_p(side.cuda_stream)
        return h

    def update_native(self, launch, vert_off):
        """update() for a caller whose ONE native call enqueues the re-fit and the pre-pass (mi.Scene.step_native -> ffx_scene_step_h):
        the same blob rotation and ordering.  launch(blob index, side stream handle, with the pre-pass?) -> the scene description whose apex records and
        tile bins the blob then holds; vert_off: the frame offsets [S] the call used (host int32, already checked by it)."""
        self.version += 1
        nxt = (self._cur + 1) % self._ring()
        side = self._side_of(nxt)
        self._wait_readers(nxt, side)
        if self._pool_written is not None:
            side.wait_event(self._pool_written)
        ev = self._upd_done[nxt]
        if ev is None:
            ev = self._upd_done[nxt] = torch.cuda.Event()
        sh = self._side_handle_of(side)
        self._apex[nxt] = None
        self._apex_written[nxt] = None
        self._top[nxt] = None
        if self.timing is None:
            defer = self._last_spp <= self._defer_top_spp
            sd = launch(nxt, sh, 3 if defer else 1)  # (bit 1: FFX_STEP_DEFER_TOP — _acquire launches the top in front of the first reader)
            if defer:
                self._top[nxt] = True
            self._apex[nxt] = apex_key(sd)
        else:  # (bench.py's per-launch event pairs: around the re-fit alone, as on the Python path; the pre-pass as a call of its own behind them)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(side)
            sd = launch(nxt, sh, 0)
            b.record(side)
            self.timing.append(("scene_update", a, b))
            self._prepare_apex(nxt, sd, sh)
        np.copyto(self._vert_off_host, vert_off)
        self._vert_off_dev_stale = True
        ev.record(side)
        self._cur = nxt

    def _prepare_apex(self, i, sd, stream=None):
        """(on the stream the re-fit of blob i was enqueued on) the records changed: what the apex areas held is void"""
        self._apex[i] = None
        self._apex_written[i] = None
        if sd is not None and not _lane_kernels():
            self._call("ffx_apex_prepare", _dev(self._blobs[i], torch.uint8, "blob"), C.byref(self.info), C.byref(sd), stream if stream is not None else _stream(self._didx))
            self._apex[i] = apex_key(sd)

    def _apex_flag(self, key):
        """FFX_RENDER_APEX_READY if the current blob's apex areas hold `key`; they will after the call either way"""
        i = self._cur if self._async else 0
        ready = self._apex[i] == key and not _lane_kernels()
        if self.device.type == "cuda":
            so = _stream_obj(self._didx)
            if not ready:
                self._wait_readers(i, so)  # the call's pre-pass rewrites the apex areas: renders on other streams may still read them
                self._apex_writer_pending = True
            else:
                w = self._apex_written[i]
                if w is not None and w[0] != so.cuda_stream:
                    so.wait_event(w[1])  # written by a pre-pass on another stream (not by the re-fit, which _acquire has waited for)
        if _lane_kernels():
            self._apex[i] = None  # (a cache-writing render runs the packet kernel — and its pre-pass — even then: claim nothing afterwards)
            return 0
        self._apex[i] = key
        return _abi.RENDER_APEX_READY if ready else 0

    def _update_into(self, blob, xforms, on_device, stream=None):
        with self._timed("scene_update"):
            if not on_device and self.n_shapes <= 32:
                xf = xforms.detach().numpy() if isinstance(xforms, torch.Tensor) else np.asarray(xforms)
                # the host tables go through two persistent ctypes arrays with numpy views on them: `ndarray.ctypes.data_as`
                # per call cost more than the launch it fed (measured 120 us each inside the optimiser's loop)
                tabs = getattr(self, "_host_tabs", None)
                if tabs is None:
                    xc, oc = (C.c_float * (16 * self.n_shapes))(), (C.c_int32 * self.n_shapes)()
                    tabs = self._host_tabs = (xc, np.frombuffer(xc, dtype=np.float32).reshape(self.n_shapes, 16), oc, np.frombuffer(oc, dtype=np.int32))
                tabs[1][...] = np.asarray(xf, dtype=np.float32).reshape(self.n_shapes, 16)
                tabs[3][...] = self._vert_off_host
                self._call(
                    "ffx_scene_update_h", _dev(blob, torch.uint8, "blob"), C.byref(self.info), _dev(self.src_verts), _dev(self.tris, torch.int32),
                    _dev(self.tri_shape, torch.int32), tabs[2], tabs[0], self.n_shapes, C.byref(self._smooth[0]) if self._smooth is not None else None,
                    stream if stream is not None else _stream(self._didx),
                )
                return
            if self._vert_off_dev_stale:
                self.vert_off = torch.from_numpy(self._vert_off_host).to(self.device)
                self._vert_off_dev_stale = False
            xf = xforms if isinstance(xforms, torch.Tensor) else torch.as_tensor(np.asarray(xforms, np.float32))
            xf = xf.to(device=self.device, dtype=torch.float32).reshape(self.n_shapes, 16).contiguous()
            self._xf = xf  # keep alive until the stream has consumed it
            self._call(
                "ffx_scene_update", _dev(blob, torch.uint8, "blob"), C.byref(self.info), _dev(self.src_verts), _dev(self.tris, torch.int32),
                _dev(self.tri_shape, torch.int32), _dev(self.vert_off, torch.int32), _dev(xf), self.n_shapes,
                C.byref(self._smooth[0]) if self._smooth is not None else None, _stream(self._didx),
            )

    def trace_primary(self, cam, spp=1, jitter=0, seed=0, want_ids=True):
        n = cam.width * cam.height * spp
        t = torch.empty(n, dtype=torch.float32, device=self.device)
        shape = torch.empty(n, dtype=torch.int32, device=self.device) if want_ids else None
        prim = torch.empty(n, dtype=torch.int32, device=self.device) if want_ids else None
        flags = 0
        if not _lane_kernels():
            i = self._cur if self._async else 0
            self._acquire()
            so = _stream_obj(self._didx)
            key = apex_key(cam=cam)
            if _camera_part(self._apex[i]) == _camera_part(key):
                # the camera's area already holds this camera's records and tile bins (a render of this pose from it, or an earlier trace):
                # nothing is rewritten — whatever the areas were claimed to hold, they still do
                flags = _abi.RENDER_APEX_READY
                w = self._apex_written[i]
                if w is not None and w[0] != so.cuda_stream:
                    so.wait_event(w[1])
            else:
                self._apex[i] = key  # (it rewrites the camera's area; the emitters' areas keep what they had, but a render has to re-derive its own)
                self._wait_readers(i, so)
                self._apex_writer_pending = True
        self._call(
            "ffx_trace_primary", _dev(self.blob, torch.uint8), C.byref(self.info), C.byref(cam), int(spp), int(bool(jitter)) | flags, int(seed) & 0xFFFFFFFF,
            _dev(t), _dev(shape, torch.int32) if want_ids else None, _dev(prim, torch.int32) if want_ids else None, _stream(self._didx),
        )
        self._release()
        return t, shape, prim

    def trace_rays(self, origins, dirs, tmax=3.0e38):
        n = origins.shape[0]
        t = torch.empty(n, dtype=torch.float32, device=self.device)
        shape = torch.empty(n, dtype=torch.int32, device=self.device)
        prim = torch.empty(n, dtype=torch.int32, device=self.device)
        self._call(
            "ffx_trace_rays", _dev(self.blob, torch.uint8), C.byref(self.info), _dev(origins, name="origins"), _dev(dirs, name="dirs"), n, float(tmax),
            _dev(t), _dev(shape, torch.int32), _dev(prim, torch.int32), _stream(self._didx),
        )
        self._release()
        return t, shape, prim

    def _timed(self, name):
        return _EventPair(self.timing, name)

    def render_fwd(self, sd, albedo, tex, spp, seed=0, fp16=False, cache=None, sparse_adjoint=False, cache_zeroed=False, keep_dropped=False, img_out=None):
        """K8.  With `cache` (a uint8 tensor of render_cache_bytes(...) bytes) the kernel also stores one
        footprint of every pixel in the projector texture for render_bwd_cached (opaque layout, ffx.h).
        sparse_adjoint (with a cache): FFX_RENDER_SPARSE_ADJOINT — gradients are only wanted at texels whose value is
        not zero (a pattern optimiser's case), dark footprints are skipped.  cache_zeroed: FFX_RENDER_CACHE_ZEROED — the caller has
        cleared the first 64 bytes of `cache` on this stream.  keep_dropped: FFX_RENDER_CACHE_KEEP_DROPPED — the header's count of dropped
        samples survives this call's reset (the later scene samples of a step that reuses one cache).
        A filtered film (sd.rfilter) with a cache: ffx_render_fwd_cache_filtered (per-sample records; render_bwd_cached needs the seed)."""
        H, W = sd.cam.height, sd.cam.width
        self._last_spp = spp
        mats_arg = _check_materials(sd, albedo)
        img = torch.empty((H, W, 3), dtype=torch.float16 if fp16 else torch.float32, device=self.device) if img_out is None else img_out
        if tuple(img.shape) != (H, W, 3) or img.dtype != (torch.float16 if fp16 else torch.float32) or not img.is_contiguous():
            raise ValueError("img_out must be a contiguous [H, W, 3] tensor of the film's type")
        blob = self.blob  # (acquire first: the flag below speaks about the blob this call reads)
        flags = int(bool(fp16)) | self._apex_flag(apex_key(sd))
        if sd.rfilter:  # a reconstruction filter that spreads samples over neighbouring pixels: its own entry point and a scratch area
            scratch = torch.empty(render_filter_bytes(sd), dtype=torch.uint8, device=self.device)  # (caching allocator, stream-ordered: renders on two streams never share one)
            if cache is not None:  # ... and the per-sample records of its adjoint (ABI 7)
                if cache.numel() < render_cache_bytes_sd(sd, spp):
                    raise ValueError("cache tensor too small")
                flags |= (_abi.RENDER_SPARSE_ADJOINT if sparse_adjoint else 0) | (_abi.RENDER_CACHE_ZEROED if cache_zeroed else 0) | (_abi.RENDER_CACHE_KEEP_DROPPED if keep_dropped else 0)
                with self._timed("render_fwd"):
                    self._call("ffx_render_fwd_cache_filtered", _dev(blob, torch.uint8), C.byref(self.info), C.byref(sd), mats_arg,
                               _dev(tex, name="tex") if tex is not None else None, int(spp), int(seed) & 0xFFFFFFFF, flags, _dev(img, img.dtype),
                               _dev(cache, torch.uint8, "cache"), _dev(scratch, torch.uint8), _stream(self._didx))
                self._release()
                return img
            with self._timed("render_fwd"):
                self._call("ffx_render_fwd_filtered", _dev(blob, torch.uint8), C.byref(self.info), C.byref(sd), mats_arg,
                           _dev(tex, name="tex") if tex is not None else None, int(spp), int(seed) & 0xFFFFFFFF, flags, _dev(img, img.dtype), _dev(scratch, torch.uint8),
                           _stream(self._didx))
            self._release()
            return img
        if cache is not None:
            if cache.numel() < render_cache_bytes_sd(sd, spp):
                raise ValueError("cache tensor too small")
            flags |= (_abi.RENDER_SPARSE_ADJOINT if sparse_adjoint else 0) | (_abi.RENDER_CACHE_ZEROED if cache_zeroed else 0) | (_abi.RENDER_CACHE_KEEP_DROPPED if keep_dropped else 0)
            with self._timed("render_fwd"):
                self._call(
                    "ffx_render_fwd_cache", _dev(blob, torch.uint8), C.byref(self.info), C.byref(sd), mats_arg,
                    _dev(tex, name="tex") if tex is not None else None, int(spp), int(seed) & 0xFFFFFFFF, flags,
                    _dev(img, img.dtype), _dev(cache, torch.uint8, "cache"), _stream(self._didx),
                )
            self._release()
            return img
        with self._timed("render_fwd"):
          self._call(
            "ffx_render_fwd", _dev(blob, torch.uint8), C.byref(self.info), C.byref(sd), mats_arg,
            _dev(tex, name="tex") if tex is not None else None, int(spp), int(seed) & 0xFFFFFFFF, flags, _dev(img, img.dtype), _stream(self._didx),
          )
        self._release()
        return img

    def render_fwd_adjoint(self, sd, albedo, tex, spp, seed, gimg, out=None, dot_out=None, fp16=False, sparse_adjoint=False, img_out=None):
        """K8 with the adjoint folded in (ffx_render_fwd_adjoint): for a loss whose gradient `gimg` [H,W,3] does not depend on the image.
        -> (img, gtex): the render, and gtex (+)= its adjoint applied to gimg — `out`: accumulate into this [tex_h, tex_w, channels]
        tensor instead of a fresh zeroed one.  dot_out: _abi.ADJOINT_DOT_SLOTS float32 partial sums that <gimg, img> is added to."""
        H, W = sd.cam.height, sd.cam.width
        if sd.rfilter and (dot_out is not None or sd.proj.tex_channels != 1 or sd.n_base_tex > 0):
            raise ValueError("render_fwd_adjoint with a reconstruction filter: 1-channel projector textures, no textured base colours, no dot_out "
                             "(the image forms behind the render launch) — box-filtered pixels are what the general call folds; use render_fwd + render_bwd")
        mats_arg = _check_materials(sd, albedo)
        img = torch.empty((H, W, 3), dtype=torch.float16 if fp16 else torch.float32, device=self.device) if img_out is None else img_out
        if tuple(img.shape) != (H, W, 3) or img.dtype != (torch.float16 if fp16 else torch.float32):
            raise ValueError("img_out must be [H, W, 3] of the film's type")
        gtex = torch.zeros((sd.proj.tex_h, sd.proj.tex_w, sd.proj.tex_channels), dtype=torch.float32, device=self.device) if out is None else out
        if tuple(gimg.shape) != (H, W, 3):
            raise ValueError("gimg must be [H, W, 3]")
        if dot_out is not None and (dot_out.dtype != torch.float32 or dot_out.numel() != _abi.ADJOINT_DOT_SLOTS):
            raise ValueError(f"dot_out must hold {_abi.ADJOINT_DOT_SLOTS} float32 partial sums (the caller zeroes and sums them)")
        blob = self.blob
        flags = int(bool(fp16)) | (_abi.RENDER_SPARSE_ADJOINT if sparse_adjoint else 0) | self._apex_flag(apex_key(sd))
        if sd.rfilter:  # the filtered film: weights -> G, ONE render launch with the adjoint folded in, gather (ffx_render_fwd_adjoint_filtered)
            scratch = torch.empty(render_filter_bytes(sd), dtype=torch.uint8, device=self.device)
            with self._timed("render_fwd"):
                self._call("ffx_render_fwd_adjoint_filtered", _dev(blob, torch.uint8), C.byref(self.info), C.byref(sd), mats_arg, _dev(tex, name="tex"), int(spp),
                           int(seed) & 0xFFFFFFFF, flags, _dev(img, img.dtype), _dev(gimg, name="gimg"), _dev(gtex), _dev(scratch, torch.uint8), _stream(self._didx))
            self._release()
            return img, gtex
        with self._timed("render_fwd"):
            self._call("ffx_render_fwd_adjoint", _dev(blob, torch.uint8), C.byref(self.info), C.byref(sd), mats_arg, _dev(tex, name="tex"), int(spp),
                       int(seed) & 0xFFFFFFFF, flags, _dev(img, img.dtype), _dev(gimg, name="gimg"), _dev(gtex), _dev(dot_out) if dot_out is not None else None,
                       _stream(self._didx))
        self._release()
        return img, gtex

    def render_bwd_cached(self, sd, albedo, cache, spp, gimg, out=None, img=None, dot_out=None, seed=None):
        """K9 from the adjoint cache written by render_fwd(..., cache=...): scatters per-pixel footprints, no BVH.
        `out`: accumulate into this [tex_h, tex_w, channels] tensor instead of a fresh zeroed one.
        `img` + `dot_out` (render_dot_slots(W, H) float32 partial sums, one per 8x8-pixel block): the same launch adds
        <gimg, img> to them — their sum is the value of a loss that is linear in the image, whose gradient gimg is."""
        gtex = torch.zeros((sd.proj.tex_h, sd.proj.tex_w, sd.proj.tex_channels), dtype=torch.float32, device=self.device) if out is None else out
        mats_arg = _check_materials(sd, albedo)
        if (img is None) != (dot_out is None):
            raise ValueError("img and dot_out go together")
        if sd.rfilter:  # the filtered film's cache: per-sample records, filter weights re-derived from the forward's seed
            if seed is None or img is not None:
                raise ValueError("render_bwd_cached of a filtered render needs the forward's seed (and has no <gimg, img> output)")
            if tuple(gimg.shape) != (sd.cam.height, sd.cam.width, 3):
                raise ValueError("gimg must be [H, W, 3]")
            with self._timed("render_bwd_cached"):
                self._call("ffx_render_bwd_cached_filtered", C.byref(sd), mats_arg, _dev(cache, torch.uint8, "cache"), int(spp), int(seed) & 0xFFFFFFFF,
                           _dev(gimg, name="gimg"), _dev(gtex), _stream(self._didx))
            return gtex
        if img is not None and (tuple(img.shape) != tuple(gimg.shape) or dot_out.dtype != torch.float32 or dot_out.numel() != render_dot_slots(sd.cam.width, sd.cam.height)):
            raise ValueError("img must have gimg's shape and dot_out must be render_dot_slots(W, H) float32 partial sums (the caller zeroes and sums them)")
        with self._timed("render_bwd_cached"):
            self._call("ffx_render_bwd_cached", C.byref(sd), mats_arg, _dev(cache, torch.uint8, "cache"), int(spp),
                       _dev(gimg, name="gimg"), _dev(gtex), _dev(img, img.dtype, "img") if img is not None else None,
                       int(img is not None and img.dtype == torch.float16), _dev(dot_out) if dot_out is not None else None, _stream(self._didx))
        return gtex

    def render_bwd_cached_l1(self, sd, albedo, cache, spp, img, target, weight, out, loss_slots):
        """K9 under weight * L1Loss(img, target) (ffx_render_bwd_cached_l1): the loss's gradient image is formed inside the scatter launch and the loss value
        goes to `loss_slots` (render_dot_slots(W, H) float32 partial sums) — no loss launch, no gradient image.  -> `out`, or None when the library
        declines the case (FFX_ERR_UNSUPPORTED: the caller takes ffx_l1_value_grad + render_bwd_cached)."""
        if sd.rfilter or img.dtype != torch.float32 or target.dtype != torch.float32 or tuple(img.shape) != (sd.cam.height, sd.cam.width, 3) or tuple(target.shape) != tuple(img.shape):
            return None
        if loss_slots.dtype != torch.float32 or loss_slots.numel() != render_dot_slots(sd.cam.width, sd.cam.height):
            raise ValueError("loss_slots must be render_dot_slots(W, H) float32 partial sums")
        mats_arg = _check_materials(sd, albedo)
        with self._timed("render_bwd_cached"):
            rc = api().call_rc("ffx_render_bwd_cached_l1", C.byref(sd), mats_arg, _dev(cache, torch.uint8, "cache"), int(spp), _dev(img, name="img"), _dev(target, name="target"),
                               float(weight), _dev(out), _dev(loss_slots), _stream(self._didx), allow=(_abi.FFX_ERR_UNSUPPORTED,))
        return out if rc == 0 else None

    def render_bwd_det_part(self, sd, albedo, spp, seed, gimg, part, acc, scale_log2=0):
        """ONE pass of the deterministic re-tracing adjoint (ffx_render_bwd_det_part): part 1 — the largest |tap| of this render into the int32
        word `acc` (the float's bits; maximum over calls); part 2 — every tap as a 64-bit fixed-point integer at 2^scale_log2 into the int64
        tensor `acc` [tex_h, tex_w, channels] (sum over calls).  The caller clears `acc`, reduces it over samples and ranks, and converts
        (det_scale_log2, det_finish_): sums that come out bitwise equal whatever the order or the number of ranks."""
        if _lane_kernels():
            raise ValueError("the per-lane kernels (FFX_TRAVERSAL=lane) have no deterministic adjoint")
        want = torch.int32 if part == 1 else torch.int64
        if acc.dtype != want or not acc.is_contiguous() or (part == 2 and acc.numel() != sd.proj.tex_h * sd.proj.tex_w * sd.proj.tex_channels):
            raise ValueError("render_bwd_det_part: acc must be one int32 word (part 1) / an int64 tensor of the texture's size (part 2)")
        mats_arg = _check_materials(sd, albedo)
        blob = self.blob
        flags = self._apex_flag(apex_key(sd))
        scratch = torch.empty(render_filter_bytes(sd), dtype=torch.uint8, device=self.device) if sd.rfilter else None
        with self._timed("render_bwd"):
            self._call("ffx_render_bwd_det_part", _dev(blob, torch.uint8), C.byref(self.info), C.byref(sd), mats_arg, int(spp), int(seed) & 0xFFFFFFFF, flags,
                       _dev(gimg, name="gimg"), int(part), int(scale_log2), C.c_void_p(acc.data_ptr()), _dev(scratch, torch.uint8) if scratch is not None else None,
                       _stream(self._didx))
        self._release()

    def render_bwd(self, sd, albedo, spp, seed, gimg, deterministic=None):
        """the re-tracing adjoint.  deterministic (default: FFX_DETERMINISTIC=1 in the environment): ffx_render_bwd_det — bitwise
        reproducible accumulation (64-bit fixed point instead of float atomics; two re-traces and one host synchronisation)."""
        gtex = torch.zeros((sd.proj.tex_h, sd.proj.tex_w, sd.proj.tex_channels), dtype=torch.float32, device=self.device)
        mats_arg = _check_materials(sd, albedo)
        blob = self.blob
        if deterministic is None:
            deterministic = deterministic_mode()
        if deterministic:
            if _lane_kernels():
                raise ValueError("the per-lane kernels (FFX_TRAVERSAL=lane) have no deterministic adjoint")
            flags = self._apex_flag(apex_key(sd))
            wsb = api().lib.ffx_render_bwd_det_bytes(C.byref(sd))
            work = torch.empty(wsb, dtype=torch.uint8, device=self.device)
            with self._timed("render_bwd"):
                self._call("ffx_render_bwd_det", _dev(blob, torch.uint8), C.byref(self.info), C.byref(sd), mats_arg, int(spp), int(seed) & 0xFFFFFFFF, flags,
                           _dev(gimg, name="gimg"), _dev(gtex), _dev(work, torch.uint8), _stream(self._didx))
            self._release()
            return gtex
        # (ABI 7: the re-tracing adjoints take FFX_RENDER_APEX_READY like the renders.  Before, they always re-ran the pre-pass — and rewrote the
        # tile bins under the eyes of renders of the same pose on the scene's other render stream)
        flags = self._apex_flag(apex_key(sd))
        if sd.rfilter:
            scratch = torch.empty(render_filter_bytes(sd), dtype=torch.uint8, device=self.device)
            with self._timed("render_bwd"):
                self._call("ffx_render_bwd_filtered", _dev(blob, torch.uint8), C.byref(self.info), C.byref(sd), mats_arg, int(spp), int(seed) & 0xFFFFFFFF, flags,
                           _dev(gimg, name="gimg"), _dev(gtex), _dev(scratch, torch.uint8), _stream(self._didx))
            self._release()
            return gtex
        with self._timed("render_bwd"):
            self._call(
                "ffx_render_bwd", _dev(blob, torch.uint8), C.byref(self.info), C.byref(sd), mats_arg, int(spp),
                int(seed) & 0xFFFFFFFF, flags, _dev(gimg, name="gimg"), _dev(gtex), _stream(self._didx),
            )
        self._release()
        return gtex
