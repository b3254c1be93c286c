"""Differentiable front-ends: PyTorch `autograd.Function`s whose forward and backward are single
calls into libffx_hip.so.  They replace the autograd graphs that the reference builds out of
eager torch ops (graphics/rasterization.py, projection/laser.py) and the Dr.Jit AD bridge around
`mi.render` (graphics/depth.py:9,33,128 show the `dr.wrap_ad` pattern).
"""
import os

import numpy as np
import torch

from . import ops

FLIP_Y = np.diag([1.0, -1.0, 1.0, 1.0]).astype(np.float32)


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


class _ProjectRays(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rays, KF):
        rays = _c(rays)
        ctx.save_for_backward(rays)
        ctx.KF = KF
        return ops.project_rays_fwd(rays, KF)

    @staticmethod
    def backward(ctx, g):
        (rays,) = ctx.saved_tensors
        return ops.project_rays_bwd(rays, ctx.KF, _c(g)), None


def project_rays(rays, KF):
    """K1: transform_points(rays, K @ FLIP_Y) (projection/laser.py:262-275); KF is a host 4x4."""
    return _ProjectRays.apply(rays, np.asarray(KF, np.float32))


class _RasterizePoints(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pts, sigma, size0, size1):
        pts = _c(pts)
        ctx.save_for_backward(pts)
        ctx.args = (sigma, size0, size1)
        return ops.splat_dense_fwd(pts, sigma, size0, size1)

    @staticmethod
    def backward(ctx, g):
        (pts,) = ctx.saved_tensors
        sigma, size0, size1 = ctx.args
        return ops.splat_dense_bwd(pts, sigma, size0, size1, _c(g)), None, None, None


def rasterize_points_dense(pts, sigma, size0, size1):
    """dense [N,size1,size0] layers of graphics/rasterization.py:7-37, differentiable in pts."""
    return _RasterizePoints.apply(pts, float(sigma), int(size0), int(size1))


class _RasterizeLines(torch.autograd.Function):
    @staticmethod
    def forward(ctx, lines, sigma, size0, size1):
        lines = _c(lines.float())
        ctx.save_for_backward(lines)
        ctx.args = (sigma, size0, size1)
        return ops.splat_lines_fwd(lines, sigma, size0, size1)

    @staticmethod
    def backward(ctx, g):
        (lines,) = ctx.saved_tensors
        sigma, size0, size1 = ctx.args
        return ops.splat_lines_bwd(lines, sigma, size0, size1, _c(g.float())), None, None, None


def rasterize_lines(lines, sigma, size0, size1):
    """soft line segments [N,2,2] -> [N,size1,size0] (graphics/rasterization.py:107-153), differentiable in the
    segments (the reference optimises them through it, :645-743)."""
    return _RasterizeLines.apply(lines, float(sigma), int(size0), int(size1))


class _Splat(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pts, sigma, reduce, half_window, size0, size1):
        pts = _c(pts)
        tex = ops.splat_fwd(pts, sigma, reduce, half_window, size0, size1)
        ctx.save_for_backward(pts, tex)
        ctx.args = (sigma, reduce, half_window, size0, size1)
        return tex

    @staticmethod
    def backward(ctx, g):
        pts, tex = ctx.saved_tensors
        sigma, reduce, half_window, size0, size1 = ctx.args
        return ops.splat_bwd(pts, sigma, reduce, half_window, size0, size1, tex, _c(g)), None, None, None, None, None


def splat(pts, sigma, size0, size1, reduce="sum", half_window=-1):
    """fused rasterize_points + sum/softor over the point axis -> [size1,size0]."""
    return _Splat.apply(pts, float(sigma), reduce, int(half_window), int(size0), int(size1))


class _Blur(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img, ksize, sigma):
        ctx.args = (ksize, sigma)
        return ops.blur_fwd(_c(img), ksize, sigma)

    @staticmethod
    def backward(ctx, g):
        ksize, sigma = ctx.args
        return ops.blur_bwd(_c(g), ksize, sigma), None, None


def gaussian_blur(img, ksize=5, sigma=3.0):
    """kornia.filters.gaussian_blur2d(img, (k,k), (s,s)) on a [H,W] tensor (vocalfold_scene.py:61-63)."""
    return _Blur.apply(img, int(ksize), float(sigma))


# renders whose adjoint cache would exceed this many bytes fall back to the re-tracing adjoint
CACHE_LIMIT_BYTES = int(float(os.environ.get("FFX_CACHE_LIMIT_GB", "32")) * (1 << 30))


# The cache keeps ONE 5x5-texel footprint per pixel; samples outside it go to an arena of 1/64 of all samples.  A projector
# texture much finer than the camera's pixel footprint makes most samples strays and the arena overflows (the gradient of the
# dropped samples would be lost: ffx_render_bwd_cached poisons gtex instead, ffx.h).  Expected texels per camera pixel along
# an axis ~ (texels per projector radian) / (pixels per camera radian) at equal depth; beyond this many the cache is refused
# up front and the adjoint re-traces.
# (the 256x256 camera / 500x500 texture of the parity tests is 4.2 and stays well inside the arena; the safety net behind
# this estimate is the `dropped` count: ffx_render_cache_status, checked by _Render.backward and PatternOptimizer)
def max_texels_per_pixel():
    return float(os.environ.get("FFX_CACHE_MAX_TEXELS_PER_PIXEL", "6.0"))


def texels_per_pixel(sd):
    """texels of the projector texture per camera pixel (per axis, at equal distance from both devices, on the optical
    axes): camera_to_sample[0][0] is 0.5 / tan(fov_x / 2) in sample units per unit tangent for both (mi.perspective_projection)"""
    kc, kp = abs(float(sd.cam.camera_to_sample[0])), abs(float(sd.proj.camera_to_sample[0]))
    if kc == 0.0 or kp == 0.0:
        return float("inf")
    return (kp * sd.proj.tex_w) / (kc * sd.cam.width)


def cache_supported(sd, spp):
    """whether ffx_render_fwd_cache accepts this render (its records pack the texel in 12+12 bits and
    the shape id in 8), the cache fits the FFX_CACHE_LIMIT_GB budget and a pixel's samples can be expected to
    stay inside one 5x5-texel footprint"""
    if not sd.proj.enabled or ops.deterministic_mode():  # (FFX_DETERMINISTIC=1: every adjoint re-traces with fixed-point accumulation)
        return False
    if sd.proj.tex_w > 4094 or sd.proj.tex_h > 4094 or sd.n_shapes > 255 or sd.n_base_tex > 0:  # (textured base colours: one colour per shape in the footprint)
        return False
    if sd.rfilter:
        # a filter that spreads samples over neighbouring pixels: ffx_render_fwd_cache_filtered keeps one 16-byte record (+ 4 with material rows) per
        # sample of the pixels that have a lit sample, in an arena of 64-sample blocks (a block per pass of every pixel up to 2^18 blocks — 344 MB
        # at 512 x 512 x 64 —, a quarter of them beyond: include/ffx.h) — no footprint a fine texture could overflow; up to 16 passes per pixel
        return spp <= 1024 and ops.render_cache_bytes_sd(sd, spp) <= CACHE_LIMIT_BYTES
    if texels_per_pixel(sd) > max_texels_per_pixel():
        return False
    return ops.render_cache_bytes_sd(sd, spp) <= CACHE_LIMIT_BYTES


class CacheOverflowError(RuntimeError):
    """the adjoint cache's arena of single-sample records filled up: its gradient would have holes"""


class _Render(torch.autograd.Function):
    @staticmethod
    def forward(ctx, tex, geom, sd, albedo, spp, seed, fp16):
        ctx.geom, ctx.sd, ctx.spp, ctx.seed = geom, sd, spp, seed
        ctx.tex_shape = tex.shape
        t = _c(tex)
        if t.dim() == 2:
            t = t.unsqueeze(-1)
        ctx.cache = None
        ctx.albedo = albedo
        if tex.requires_grad and sd.proj.enabled:
            # the adjoint needs the albedo as it was at the forward pass (Scene._apply overwrites the
            # tensor in place on the next randomisation): keep a private copy (3 or 16 floats per shape) — unless the scene
            # description carries the rows itself (sd.n_mat_h: a value, already a snapshot)
            ctx.albedo = albedo.clone() if albedo is not None and sd.n_mat_h == 0 else None
            if cache_supported(sd, spp):
                # store a texture footprint per pixel now instead of re-tracing the scene in backward
                ctx.cache = torch.empty(ops.render_cache_bytes_sd(sd, spp), dtype=torch.uint8, device=t.device)
        ctx.pose_version = geom.version
        return geom.render_fwd(sd, albedo, t, spp, seed, fp16, cache=ctx.cache)

    @staticmethod
    def backward(ctx, g):
        g = _c(g.float())
        gtex = None
        if ctx.cache is not None:
            # (one 64-byte read + stream sync per backward of this generic path; the optimiser's explicit step checks lazily)
            used, cap, dropped = ops.render_cache_status(ctx.cache)  # (box film: stray samples beyond the arena; filtered film: pixels that found no block)
            if dropped == 0:
                gtex = ctx.geom.render_bwd_cached(ctx.sd, ctx.albedo, ctx.cache, ctx.spp, g, seed=ctx.seed if ctx.sd.rfilter else None)
            elif ctx.geom.version != ctx.pose_version:
                raise CacheOverflowError(
                    f"render backward: the adjoint cache overflowed ({dropped} samples beyond its {cap} single-sample records — a projector "
                    "texture much finer than the camera's pixels?) and the scene has been re-fitted since the forward pass, so the adjoint "
                    "cannot re-trace: call backward before the next randomisation, or set FFX_CACHE_LIMIT_GB=0 to always re-trace")
            ctx.cache = None
        if gtex is None:  # replays the geometry: it must still be in the pose of the forward pass
            if ctx.geom.version != ctx.pose_version:
                raise RuntimeError(
                    "render backward: the scene was re-fitted (randomize()/update()) between forward and backward and this render "
                    "uses the re-tracing adjoint (cache over FFX_CACHE_LIMIT_GB, texture > 4094^2 or > 255 shapes): call backward "
                    "before the next randomisation"
                )
            gtex = ctx.geom.render_bwd(ctx.sd, ctx.albedo, ctx.spp, ctx.seed, g)
        return gtex.reshape(ctx.tex_shape), None, None, None, None, None, None


def render(tex, geom, sd, albedo, spp, seed=0, fp16=False):
    """K8/K9: image [H,W,3], differentiable w.r.t. the projector texture ([h,w] or [h,w,c]).
    When the texture requires grad the forward kernel also stores each pixel's footprint in the texture
    (128 B per pixel + a small arena: 40 MB at 512x512x64; under a gaussian film an arena of per-sample records — 16 / 20 bytes per sample
    of the pixels that have a lit sample, room for every pixel up to 344 MB, a quarter of them beyond) and the adjoint scatters those footprints; beyond FFX_CACHE_LIMIT_GB the adjoint re-traces instead (then the geometry must not be re-fitted between
    forward and backward)."""
    return _Render.apply(tex, geom, sd, albedo, int(spp), int(seed), bool(fp16))
