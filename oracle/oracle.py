"""numpy front-end of the CPU oracle (oracle/_build/libffx_oracle.so).

TEST INFRASTRUCTURE ONLY — see the header of oracle/ffx_oracle.c.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(_HERE))
from fireflies_amd import _abi  # noqa: E402  (ABI declarations only; loads nothing)

# FFX_ORACLE_LIB: use another build of the oracle (e.g. one compiled with -fsanitize=address,undefined)
LIB_PATH = os.environ.get("FFX_ORACLE_LIB") or os.path.join(_HERE, "_build", "libffx_oracle.so")


def build(force=False):
    if force or not os.path.exists(LIB_PATH):
        subprocess.check_call(["make", "-C", _HERE] + (["-B"] if force else []))
    return LIB_PATH


_api = None


def api():
    global _api
    if _api is None:
        build()
        _api = _abi.Api(C.CDLL(LIB_PATH))
        assert _api.backend == "cpu-oracle"
    return _api


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _p(a):
    return a.ctypes.data


def _m16(m):
    return (C.c_float * 16)(*np.asarray(m, dtype=np.float32).reshape(16).tolist())


# ---------------------------------------------------------------- K1
def project_rays_fwd(rays, KF):
    rays = _f32(rays)
    out = np.empty_like(rays)
    api().call("ffx_project_rays_fwd", _p(rays), rays.shape[0], _m16(KF), _p(out), None)
    return out


def project_rays_bwd(rays, KF, gpts):
    rays, gpts = _f32(rays), _f32(gpts)
    out = np.empty_like(rays)
    api().call("ffx_project_rays_bwd", _p(rays), rays.shape[0], _m16(KF), _p(gpts), _p(out), None)
    return out


def transform_points(pts, M, mode=0):
    pts = _f32(pts)
    out = np.empty_like(pts)
    api().call("ffx_transform_points", _p(pts), pts.shape[0], _m16(M), mode, _p(out), None)
    return out


def l1_value_grad(a, b, weight=1.0):
    a, b = _f32(a), _f32(b)
    ws = np.zeros(257, np.float32)
    g = np.empty_like(a)
    api().call("ffx_l1_value_grad", _p(a), _p(b), a.size, float(weight), _p(ws), _p(g), None)
    return float(ws[0]), g


def pattern_fwd(rays, KF, sigma, size0, size1, want_softor=True):
    rays = _f32(rays)
    n = rays.shape[0]
    pts = np.empty((n, 2), np.float32)
    tsum = np.empty((size1, size0), np.float32)
    tsor = np.empty((size1, size0), np.float32) if want_softor else None
    ws = np.zeros(int(api().lib.ffx_pattern_ws_floats(size0, size1)), np.float32)
    api().call("ffx_pattern_fwd", _p(rays), n, _m16(KF), float(sigma), size0, size1, int(want_softor), _p(pts), _p(tsum), _p(tsor) if want_softor else None, _p(ws), None, 0, None)
    return pts, tsum, tsor, ws


def pattern_bwd(rays, KF, sigma, size0, size1, tsum, tsor, gts, reg_weight, ws, loss_in=None, loss_div=1.0):
    """-> (grays_data, grays_reg, regulariser value); with `loss_in` (partial sums of the data term) a 4th item:
    (sum(loss_in) / loss_div + regulariser, sum(loss_in))"""
    rays = _f32(rays)
    n = rays.shape[0]
    gd = np.zeros((n, 3), np.float32)
    gr = np.zeros((n, 3), np.float32)
    val = np.zeros(3, np.float32)
    gts = None if gts is None else _f32(gts)
    li = None if loss_in is None else _f32(loss_in).reshape(-1)
    api().call("ffx_pattern_bwd", _p(rays), n, _m16(KF), float(sigma), size0, size1, _p(_f32(tsum)), _p(_f32(tsor)) if tsor is not None else None,
               _p(gts) if gts is not None else None, float(reg_weight), _p(_f32(ws)) if ws is not None else None, _p(gd), _p(gr), _p(val),
               _p(li) if li is not None else None, 0 if li is None else int(li.size), float(loss_div), None)
    if li is None:
        return gd, gr, float(val[0])
    return gd, gr, float(val[0]), (float(val[1]), float(val[2]))


def pattern_fwd_blur(rays, KF, sigma, size0, size1, ksize, blur_sigma, want_softor=True):
    """ffx_pattern_fwd_blur -> (pts, tsum, tsor, ws, tex)"""
    rays = _f32(rays)
    n = rays.shape[0]
    pts = np.empty((n, 2), np.float32)
    tsum = np.empty((size1, size0), np.float32)
    tsor = np.empty((size1, size0), np.float32) if want_softor else None
    ws = np.zeros(int(api().lib.ffx_pattern_ws_floats(size0, size1)), np.float32)
    tex = np.empty((size1, size0), np.float32)
    api().call("ffx_pattern_fwd_blur", _p(rays), n, _m16(KF), float(sigma), size0, size1, int(want_softor), _p(pts), _p(tsum), _p(tsor) if want_softor else None, _p(ws),
               None, 0, int(ksize), float(blur_sigma), _p(tex), None)
    return pts, tsum, tsor, ws, tex


def pattern_bwd_blur(rays, KF, sigma, size0, size1, tsum, tsor, gtex, reg_weight, ws, ksize, blur_sigma, loss_in=None, loss_div=1.0, adam=None):
    """ffx_pattern_bwd_blur -> (grays_data, grays_reg, [3] values).  adam: dict(exp_avg, exp_avg_sq, step (float32 arrays, updated in place —
    as `rays` is), lr, beta1, beta2, eps, KF_inv, lo, hi, grad_div, n_normalize) -> additionally the gradient used (grad_out)"""
    import ctypes as C

    from fireflies_amd import _abi

    assert rays.dtype == np.float32 and rays.flags.c_contiguous
    n = rays.shape[0]
    gd = np.zeros((n, 3), np.float32)
    gr = np.zeros((n, 3), np.float32)
    val = np.zeros(3, np.float32)
    gtex = None if gtex is None else _f32(gtex)
    li = None if loss_in is None else _f32(loss_in).reshape(-1)
    aa, gout = None, None
    if adam is not None:
        gout = np.zeros((n, 3), np.float32)
        cnt = np.zeros(1, np.uint32)
        aa = _abi.AdamArgs()
        aa.rays = rays.ctypes.data
        if adam["exp_avg"] is not None:  # (None: no update, only the inner product)
            aa.exp_avg, aa.exp_avg_sq, aa.step = adam["exp_avg"].ctypes.data, adam["exp_avg_sq"].ctypes.data, adam["step"].ctypes.data
        aa.grad_out, aa.counter = gout.ctypes.data, cnt.ctypes.data
        aa.lr, aa.beta1, aa.beta2, aa.eps = float(adam["lr"]), float(adam["beta1"]), float(adam["beta2"]), float(adam["eps"])
        aa.KF_inv = _m16(adam["KF_inv"])
        aa.lo, aa.hi, aa.grad_div, aa.n_normalize = float(adam["lo"]), float(adam["hi"]), float(adam.get("grad_div", 1.0)), int(adam.get("n_normalize", 1))
        if adam.get("dot") is not None:  # (a, b): the data term as an inner product
            da, db = (np.ascontiguousarray(v, np.float32).reshape(-1) for v in adam["dot"])
            part = np.zeros(n, np.float32)
            aa.dot_a, aa.dot_b, aa.dot_n, aa.dot_partial = da.ctypes.data, db.ctypes.data, int(da.size), part.ctypes.data
            aa.dot_b_n = int(db.size)
        if adam.get("guard") is not None:  # an adjoint cache's header (uint8 array): dropped != 0 -> the update is not applied
            aa.guard = adam["guard"].ctypes.data
    api().call("ffx_pattern_bwd_blur", _p(rays), n, _m16(KF), float(sigma), size0, size1, _p(_f32(tsum)), _p(_f32(tsor)) if tsor is not None else None,
               _p(gtex) if gtex is not None else None, float(reg_weight), _p(_f32(ws)) if ws is not None else None, _p(gd), _p(gr), _p(val),
               _p(li) if li is not None else None, 0 if li is None else int(li.size), float(loss_div), int(ksize), float(blur_sigma), None,
               C.byref(aa) if aa is not None else None, None)
    return (gd, gr, val) if adam is None else (gd, gr, val, gout)


def pattern_step(rays, KF, sigma, size0, size1, bufs, gtex, reg_weight, ksize, blur_sigma, adam, zero, sync, rays_kept, epoch, check_kept=False, loss_in=None, loss_div=1.0):
    """ffx_pattern_step: pattern_bwd_blur(..., adam) on bufs = (pts, tsum, tsor, ws, tex) of this step, then pattern_fwd_blur of the next into the same arrays (in place, as rays,
    adam's state, zero, sync [uint8, FFX_PATTERN_SYNC_BYTES] and rays_kept [2, n, 3] are) -> (grays_data, grays_reg, [3] values, grad_out)"""
    import ctypes as C

    from fireflies_amd import _abi

    pts, tsum, tsor, ws, tex = bufs
    assert rays.dtype == np.float32 and rays.flags.c_contiguous and sync.dtype == np.uint8 and sync.size >= _abi.PATTERN_SYNC_BYTES
    n = rays.shape[0]
    gd, gr, val, gout = np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32), np.zeros(3, np.float32), np.zeros((n, 3), np.float32)
    li = None if loss_in is None else _f32(loss_in).reshape(-1)
    aa = _abi.AdamArgs()
    aa.rays = rays.ctypes.data
    aa.exp_avg, aa.exp_avg_sq, aa.step = adam["exp_avg"].ctypes.data, adam["exp_avg_sq"].ctypes.data, adam["step"].ctypes.data
    aa.grad_out = gout.ctypes.data
    aa.lr, aa.beta1, aa.beta2, aa.eps = float(adam["lr"]), float(adam["beta1"]), float(adam["beta2"]), float(adam["eps"])
    aa.KF_inv = _m16(adam["KF_inv"])
    aa.lo, aa.hi, aa.grad_div, aa.n_normalize = float(adam["lo"]), float(adam["hi"]), float(adam.get("grad_div", 1.0)), int(adam.get("n_normalize", 1))
    guard = adam.get("guard")
    if guard is not None:
        aa.guard = guard.ctypes.data
    api().call("ffx_pattern_step", _p(rays), n, _m16(KF), float(sigma), size0, size1, _p(tsum), _p(tsor) if tsor is not None else None, _p(gtex) if gtex is not None else None,
               float(reg_weight), _p(ws) if ws is not None else None, _p(gd), _p(gr), _p(val), _p(li) if li is not None else None, 0 if li is None else int(li.size), float(loss_div),
               int(ksize), float(blur_sigma), C.byref(aa), _p(pts), _p(zero) if zero is not None else None, 0 if zero is None else int(zero.size), _p(tex), _p(rays_kept),
               int(bool(check_kept)), sync.ctypes.data, int(epoch), None)
    return gd, gr, val, gout


def adam_clamp_step(rays, grad, exp_avg, exp_avg_sq, step, lr, beta1, beta2, eps, KF, KF_inv, lo, hi, n_normalize=1, guard=None):
    """in place on rays / exp_avg / exp_avg_sq / step (float32 arrays)"""
    api().call("ffx_adam_clamp_step", _p(rays), _p(_f32(grad)), None, 1.0, None, _p(exp_avg), _p(exp_avg_sq), _p(step), rays.shape[0], float(lr), float(beta1), float(beta2), float(eps),
               _m16(KF), _m16(KF_inv), float(lo), float(hi), int(n_normalize), _p(guard) if guard is not None else None, None)
    return rays


def clamp_to_fov(rays, KF, KF_inv, lo, hi, n_normalize=1):
    out = _f32(rays).copy()
    api().call("ffx_clamp_to_fov", _p(out), out.shape[0], _m16(KF), _m16(KF_inv), float(lo), float(hi), int(n_normalize), None)
    return out


# ---------------------------------------------------------------- K2
def splat_dense_fwd(pts, sigma, size0, size1):
    pts = _f32(pts)
    out = np.empty((pts.shape[0], size1, size0), np.float32)
    api().call("ffx_splat_dense_fwd", _p(pts), pts.shape[0], sigma, size0, size1, _p(out), None)
    return out


def splat_dense_bwd(pts, sigma, size0, size1, gout):
    pts, gout = _f32(pts), _f32(gout)
    out = np.empty_like(pts)
    api().call("ffx_splat_dense_bwd", _p(pts), pts.shape[0], sigma, size0, size1, _p(gout), _p(out), None)
    return out


def splat_fwd(pts, sigma, reduce, half_window, size0, size1):
    pts = _f32(pts)
    out = np.empty((size1, size0), np.float32)
    api().call("ffx_splat_fwd", _p(pts), pts.shape[0], sigma, reduce, half_window, size0, size1, _p(out), None)
    return out


def splat_bwd(pts, sigma, reduce, half_window, size0, size1, tex, gtex):
    pts, tex, gtex = _f32(pts), _f32(tex), _f32(gtex)
    out = np.empty_like(pts)
    api().call("ffx_splat_bwd", _p(pts), pts.shape[0], sigma, reduce, half_window, size0, size1, _p(tex), _p(gtex), _p(out), None)
    return out


def splat_depth_fwd(pts, depth, sigma, size0, size1):
    pts, depth = _f32(pts), _f32(depth).reshape(-1)
    out = np.empty((pts.shape[0], size1, size0), np.float32)
    api().call("ffx_splat_depth_fwd", _p(pts), _p(depth), pts.shape[0], sigma, size0, size1, _p(out), None)
    return out


def splat_lines_fwd(lines, sigma, size0, size1):
    lines = _f32(lines)
    out = np.empty((lines.shape[0], size1, size0), np.float32)
    api().call("ffx_splat_lines_fwd", _p(lines), lines.shape[0], sigma, size0, size1, _p(out), None)
    return out


# ---------------------------------------------------------------- K3
def splat_lines_bwd(lines, sigma, size0, size1, gout):
    lines, gout = _f32(lines), _f32(gout)
    out = np.empty_like(lines)
    api().call("ffx_splat_lines_bwd", _p(lines), lines.shape[0], sigma, size0, size1, _p(gout), _p(out), None)
    return out


def blur_fwd(img, ksize=5, sigma=3.0):
    img = _f32(img)
    out = np.empty_like(img)
    api().call("ffx_blur_fwd", _p(img), img.shape[0], img.shape[1], ksize, sigma, _p(out), None)
    return out


def blur_bwd(g, ksize=5, sigma=3.0):
    g = _f32(g)
    out = np.empty_like(g)
    api().call("ffx_blur_bwd", _p(g), g.shape[0], g.shape[1], ksize, sigma, _p(out), None)
    return out


# ---------------------------------------------------------------- K5..K9
def camera_struct(to_world, camera_to_sample, near, far, width, height):
    c = _abi.Camera()
    c.to_world = _m16(to_world)
    c.camera_to_sample = _m16(camera_to_sample)
    c.near_clip, c.far_clip, c.width, c.height = near, far, width, height
    return c


class Geometry:
    """Triangle scene + the oracle's BVH blob (host memory)."""

    def __init__(self, src_verts, tris, tri_shape, vert_off, build_verts=None, smooth=None):
        """src_verts [*,3] pool, tris [F,3] shape-local indices, tri_shape [F], vert_off [S]; smooth: one flag per shape
        (interpolated shading normals, include/ffx.h ffx_smooth)."""
        self._smooth = None
        if smooth is not None and any(smooth):
            from fireflies_amd import scenes as _sc  # (pure numpy table builder shared with the product's host side)

            S = _i32(vert_off).shape[0]
            flags, vbase, a0, adj, n_vn = _sc.smooth_tables(tris, tri_shape, smooth, S)
            keep = (np.ascontiguousarray(flags), np.ascontiguousarray(vbase), np.ascontiguousarray(a0), np.ascontiguousarray(adj), np.zeros((n_vn, 3), np.float32))
            sm = _abi.Smooth()
            sm.shape_smooth = keep[0].ctypes.data_as(C.POINTER(C.c_int32))
            sm.shape_vbase = keep[1].ctypes.data_as(C.POINTER(C.c_int32))
            sm.adj_start, sm.adj, sm.n_vn, sm.vnormals = keep[2].ctypes.data, keep[3].ctypes.data, n_vn, keep[4].ctypes.data
            self._smooth = (sm, keep)
        self.src_verts = _f32(src_verts)
        self.tris = _i32(tris)
        self.tri_shape = _i32(tri_shape)
        self.vert_off = _i32(vert_off)
        self.n_shapes = self.vert_off.shape[0]
        F = self.tris.shape[0]
        glob = self.tris + self.vert_off[self.tri_shape][:, None]
        glob = _i32(glob)
        bv = self.src_verts if build_verts is None else _f32(build_verts)
        nbytes = api().lib.ffx_bvh_blob_bytes(F)
        self.blob = np.zeros(nbytes, np.uint8)
        self.info = _abi.BvhInfo()
        api().call("ffx_bvh_build_host", _p(bv), bv.shape[0], _p(glob), F, _p(self.blob), nbytes, C.byref(self.info))
        self.update(np.tile(np.eye(4, dtype=np.float32), (self.n_shapes, 1, 1)))

    def update(self, xforms, vert_off=None):
        if vert_off is not None:
            self.vert_off = _i32(vert_off)
        xf = _f32(xforms).reshape(self.n_shapes, 16)
        api().call(
            "ffx_scene_update", _p(self.blob), C.byref(self.info), _p(self.src_verts), _p(self.tris), _p(self.tri_shape),
            _p(self.vert_off), _p(xf), self.n_shapes, C.byref(self._smooth[0]) if self._smooth is not None else None, None,
        )

    @property
    def vertex_normals(self):
        """[n_vn, 3] world-space vertex normals of the last update (smooth shapes; zero rows elsewhere)"""
        return None if self._smooth is None else self._smooth[1][4]

    def trace_primary(self, cam, spp=1, jitter=0, seed=0):
        n = cam.width * cam.height * spp
        t = np.empty(n, np.float32)
        shape = np.empty(n, np.int32)
        prim = np.empty(n, np.int32)
        api().call("ffx_trace_primary", _p(self.blob), C.byref(self.info), C.byref(cam), spp, jitter, seed, _p(t), _p(shape), _p(prim), None)
        return t, shape, prim

    def trace_rays(self, origins, dirs, tmax=3.0e38):
        o, d = _f32(origins), _f32(dirs)
        n = o.shape[0]
        t = np.empty(n, np.float32)
        shape = np.empty(n, np.int32)
        prim = np.empty(n, np.int32)
        api().call("ffx_trace_rays", _p(self.blob), C.byref(self.info), _p(o), _p(d), n, tmax, _p(t), _p(shape), _p(prim), None)
        return t, shape, prim

    def render_fwd(self, sd, albedo, tex, spp, seed=0, fp16=False):
        albedo = _f32(albedo)
        tex = _f32(tex)
        H, W = sd.cam.height, sd.cam.width
        img = np.empty((H, W, 3), np.float16 if fp16 else np.float32)
        if sd.rfilter:  # a reconstruction filter that spreads samples over neighbouring pixels: its own entry points (include/ffx.h)
            scratch = np.empty(api().lib.ffx_render_filter_bytes(C.byref(sd)), np.uint8)
            api().call("ffx_render_fwd_filtered", _p(self.blob), C.byref(self.info), C.byref(sd), _p(albedo), _p(tex), spp, seed, int(fp16), _p(img), _p(scratch), None)
            return img
        api().call("ffx_render_fwd", _p(self.blob), C.byref(self.info), C.byref(sd), _p(albedo), _p(tex), spp, seed, int(fp16), _p(img), None)
        return img

    def render_fwd_cache(self, sd, albedo, tex, spp, seed=0, fp16=False):
        albedo, tex = _f32(albedo), _f32(tex)
        H, W = sd.cam.height, sd.cam.width
        img = np.empty((H, W, 3), np.float16 if fp16 else np.float32)
        cache = np.zeros(api().lib.ffx_render_cache_bytes_sd(C.byref(sd), spp), np.uint8)
        if sd.rfilter:  # the filtered film's cache (ffx_render_fwd_cache_filtered): per-sample records + the weight each pixel received
            scratch = np.empty(api().lib.ffx_render_filter_bytes(C.byref(sd)), np.uint8)
            api().call("ffx_render_fwd_cache_filtered", _p(self.blob), C.byref(self.info), C.byref(sd), _p(albedo), _p(tex), spp, seed, int(fp16), _p(img), _p(cache),
                       _p(scratch), None)
            return img, cache
        api().call("ffx_render_fwd_cache", _p(self.blob), C.byref(self.info), C.byref(sd), _p(albedo), _p(tex), spp, seed, int(fp16), _p(img), _p(cache), None)
        return img, cache

    def render_fwd_adjoint(self, sd, albedo, tex, spp, seed, gimg, fp16=False):
        """ffx_render_fwd_adjoint -> (img, gtex, <gimg, img>)"""
        albedo, tex, gimg = _f32(albedo), _f32(tex), _f32(gimg)
        H, W = sd.cam.height, sd.cam.width
        img = np.empty((H, W, 3), np.float16 if fp16 else np.float32)
        gtex = np.zeros((sd.proj.tex_h, sd.proj.tex_w, sd.proj.tex_channels), np.float32)
        if sd.rfilter:
            scratch = np.empty(api().lib.ffx_render_filter_bytes(C.byref(sd)), np.uint8)
            api().call("ffx_render_fwd_adjoint_filtered", _p(self.blob), C.byref(self.info), C.byref(sd), _p(albedo), _p(tex), spp, seed, int(fp16), _p(img), _p(gimg), _p(gtex),
                       _p(scratch), None)
            return img, gtex, float((img.astype(np.float64) * gimg.astype(np.float64)).sum())
        dot = np.zeros(4096, np.float32)
        api().call("ffx_render_fwd_adjoint", _p(self.blob), C.byref(self.info), C.byref(sd), _p(albedo), _p(tex), spp, seed, int(fp16), _p(img), _p(gimg), _p(gtex), _p(dot), None)
        return img, gtex, float(dot.astype(np.float64).sum())

    @staticmethod
    def render_bwd_cached(sd, albedo, cache, spp, gimg, img=None, seed=None):
        """-> gtex; with `img` (the forward's image, float32 or float16) also <gimg, img>: (gtex, dot).  A filtered film's cache
        (sd.rfilter) needs the forward's `seed` (the filter weights are re-derived from the jitter)."""
        albedo, gimg = _f32(albedo), _f32(gimg)
        gtex = np.zeros((sd.proj.tex_h, sd.proj.tex_w, sd.proj.tex_channels), np.float32)
        if sd.rfilter:
            if seed is None or img is not None:
                raise ValueError("filtered cache: pass the forward's seed; no <gimg, img> output")
            api().call("ffx_render_bwd_cached_filtered", C.byref(sd), _p(albedo), _p(np.ascontiguousarray(cache)), spp, seed, _p(gimg), _p(gtex), None)
            return gtex
        if img is None:
            api().call("ffx_render_bwd_cached", C.byref(sd), _p(albedo), _p(np.ascontiguousarray(cache)), spp, _p(gimg), _p(gtex), None, 0, None, None)
            return gtex
        img = np.ascontiguousarray(img)
        dot = np.zeros(1, np.float32)
        api().call("ffx_render_bwd_cached", C.byref(sd), _p(albedo), _p(np.ascontiguousarray(cache)), spp, _p(gimg), _p(gtex), _p(img), int(img.dtype == np.float16),
                   _p(dot), None)
        return gtex, float(dot[0])

    @staticmethod
    def render_bwd_cached_l1(sd, albedo, cache, spp, img, target, weight):
        """ffx_render_bwd_cached_l1 -> (gtex, weight * L1Loss(img, target))"""
        albedo, img, target = _f32(albedo), _f32(img), _f32(target)
        gtex = np.zeros((sd.proj.tex_h, sd.proj.tex_w, sd.proj.tex_channels), np.float32)
        slots = np.zeros(int(api().lib.ffx_render_dot_slots(sd.cam.width, sd.cam.height)), np.float32)
        api().call("ffx_render_bwd_cached_l1", C.byref(sd), _p(albedo), _p(np.ascontiguousarray(cache)), spp, _p(img), _p(target), float(weight), _p(gtex), _p(slots), None)
        return gtex, float(slots.sum())

    def render_bwd(self, sd, albedo, spp, seed, gimg):
        albedo, gimg = _f32(albedo), _f32(gimg)
        gtex = np.zeros((sd.proj.tex_h, sd.proj.tex_w, sd.proj.tex_channels), np.float32)
        if sd.rfilter:
            scratch = np.empty(api().lib.ffx_render_filter_bytes(C.byref(sd)), np.uint8)
            api().call("ffx_render_bwd_filtered", _p(self.blob), C.byref(self.info), C.byref(sd), _p(albedo), spp, seed, 0, _p(gimg), _p(gtex), _p(scratch), None)
            return gtex
        api().call("ffx_render_bwd", _p(self.blob), C.byref(self.info), C.byref(sd), _p(albedo), spp, seed, 0, _p(gimg), _p(gtex), None)
        return gtex
