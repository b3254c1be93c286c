"""Pattern optimisation loop — the counterpart of the reference's EMPTY
examples/09_point_pattern_optimization.py and examples/11_domain_specific_pattern_optim.py
(SURVEY F3), reconstructed from the pieces the reference does ship:

    for step:
        pts  = laser.projectRaysToNDC()[:, :2]                    K1
        tex  = blur(sum(rasterize_points(pts, sigma, size)))      K2 + K3   (vocalfold_scene.py:56-63)
        for each scene sample k of this rank:
            seed RNGs; ff_scene.randomize()                       K5 + K6   (scene.py:360-384)
            img = mi.render(scene, spp)                           K8
            loss_k = task_loss(img); d loss_k / d tex             K9
        d loss / d rays through K3^T, K2-bwd, K1-bwd, + overlap regulariser L1(softor, sum)
                                                                  (rasterization.py:589-600)
        all-reduce [3N+1]; Adam step; laser.clamp_to_fov(); laser.normalize_rays()
                                                                  (laser.py:199-206,254-255)
"""
import os
import random

import torch

from . import _abi, dist
from . import functional as Fn
from .scene import StaleDrawError


def coverage_loss(img):
    """default task loss: minus the mean laser (green) radiance reaching the camera — the pattern
    is pulled toward surfaces that are visible and well lit; the overlap regulariser keeps the
    points apart."""
    return -img[..., 1].mean()


_COVERAGE_GRAD = {}


def _coverage_grad(img):
    key = (tuple(img.shape), img.device)
    g = _COVERAGE_GRAD.get(key)
    if g is None:
        g = torch.zeros(img.shape, dtype=torch.float32, device=img.device)
        g[..., 1] = -1.0 / float(img.shape[0] * img.shape[1])
        _COVERAGE_GRAD[key] = g
    return g


def _coverage_value_and_grad(img):
    """coverage_loss and its (constant) gradient without an autograd graph."""
    return -img[..., 1].float().mean(), _coverage_grad(img)


_coverage_value_and_grad.__wrapped_grad__ = _coverage_grad


def _coverage_accumulate(img, acc):
    """acc += coverage_loss(img) (two launches: reduction, subtract) -> its constant gradient"""
    g = _coverage_value_and_grad.__wrapped_grad__(img)
    acc.sub_(img[..., 1].mean(dtype=torch.float32))
    return g


# a task loss may carry `value_and_grad(img) -> (loss, d loss / d img)` and `accumulate_value_and_grad(img, acc) -> d loss / d img`
# (adds the value to the 0-dim tensor `acc`); otherwise autograd is used for it.  `linear_gradient(img) -> g` declares the loss
# LINEAR in the image, loss(img) = <g, img> with a constant g: the adjoint launch then also evaluates the loss
# (ffx_render_bwd_cached's dot_out) and the step needs no reduction launch at all
coverage_loss.value_and_grad = _coverage_value_and_grad
coverage_loss.accumulate_value_and_grad = _coverage_accumulate
coverage_loss.linear_gradient = _coverage_grad


def image_l1_loss(target):
    """A task loss that is NOT linear in the image: torch.nn.L1Loss()(img, target) against a fixed target render — the loss class
    the reference's own optimisation loop uses (fireflies/graphics/rasterization.py:579, 596-602), applied to the image as SURVEY 3.5's
    `task_loss(img | ...)`.  Its gradient sign(img - target) / n depends on the render, so a step takes the general adjoint path:
    cache-writing forward + ffx_render_bwd_cached (K9).  Value and gradient come from ONE launch (ffx_l1_value_grad)."""
    tgt = target.detach().float().contiguous()

    def loss(img):
        return (img.float() - tgt).abs().mean()

    def value_and_grad(img):
        from . import ops

        a = img if img.dtype == torch.float32 else img.float()
        v, g = ops.l1_value_grad(a.reshape(-1), tgt.reshape(-1))
        return v, g.view(a.shape)

    def accumulate(img, acc):
        from . import ops

        a = img if img.dtype == torch.float32 else img.float()
        if acc.dtype == torch.float32 and acc.is_contiguous() and acc.numel() >= 1:
            _, g = ops.l1_value_grad(a.reshape(-1), tgt.reshape(-1), acc=acc)  # (the value joins the step's running loss inside the reduction launch)
        else:
            v, g = ops.l1_value_grad(a.reshape(-1), tgt.reshape(-1))
            acc.add_(v)
        return g.view(a.shape)

    loss.value_and_grad = value_and_grad
    loss.accumulate_value_and_grad = accumulate
    loss.target = tgt
    loss.l1_target = (tgt, 1.0)  # (target, weight): PatternOptimizer folds value and gradient into K9 (ffx_render_bwd_cached_l1)
    return loss


class PatternOptimizer:
    def __init__(self, mi_scene, ff_scene, laser, sigma=10.0, tex_size=(500, 500), spp=64, lr=1e-3, reg_weight=0.1, samples_per_step=1,
                 base_seed=0, loss_fn=coverage_loss, blur=(5, 3.0)):
        self.mi_scene, self.ff_scene, self.laser = mi_scene, ff_scene, laser
        self.sigma, self.tex_size, self.spp = float(sigma), (int(tex_size[0]), int(tex_size[1])), int(spp)
        mi_scene.note_spp(self.spp)  # (the pre-pass of the poses to come: with or without the emitters' envelopes, mi.Scene.note_spp)
        self.reg_weight, self.samples_per_step, self.base_seed = float(reg_weight), int(samples_per_step), int(base_seed)
        self.loss_fn, self.blur = loss_fn, blur
        laser._rays = laser._rays.detach().clone().requires_grad_(True)
        try:  # one fused kernel for the single small parameter instead of ~10 launches
            self.opt = torch.optim.Adam([laser._rays], lr=lr, fused=laser._rays.is_cuda)
        except (RuntimeError, TypeError):
            self.opt = torch.optim.Adam([laser._rays], lr=lr)
        self.step_index = 0
        self._cache = None
        # scene samples per adjoint path (bench.py prints it): "fused" = forward + adjoint in one launch (a loss linear in the image),
        # "cache_k9" = cache-writing forward + ffx_render_bwd_cached (every other loss), "retrace" = ffx_render_bwd (no cache possible)
        self.step_paths = {"fused": 0, "cache_k9": 0, "retrace": 0}

    def _sample_seeds(self, step):
        """seeds of this rank's scene samples of optimisation step `step` (dist.sample_seed: independent of the world size)"""
        S = self.samples_per_step
        return [dist.sample_seed(self.base_seed, step, S, k) for k in dist.sample_ids(S, dist.rank(), dist.world_size())]

    # ------------------------------------------------------------------ texture from the current pattern
    def textures(self):
        pts = self.laser.projectRaysToNDC()[:, 0:2].contiguous()
        s0, s1 = self.tex_size
        tsum = Fn.splat(pts, self.sigma, s0, s1, "sum", -1)
        tex = Fn.gaussian_blur(tsum, self.blur[0], self.blur[1]) if self.blur else tsum
        return pts, tsum, tex

    def _render_sample(self, tex_value, seed):
        """one scene sample: randomise, render, adjoint.  Returns (d loss/d tex, loss)."""
        torch.manual_seed(seed)
        random.seed(seed)
        self.ff_scene.randomize()
        leaf = tex_value.detach().clone().requires_grad_(True)
        sd = self.mi_scene.scene_desc(tex_channels=1)
        img = Fn.render(leaf, self.mi_scene.geom, sd, self.mi_scene.materials_arg(sd), self.spp, seed)
        loss = self.loss_fn(img)
        (g,) = torch.autograd.grad(loss, leaf)
        return g, loss.detach()

    # ------------------------------------------------------------------ explicit-adjoint step (default)
    def _loss_and_grad(self, img):
        vg = getattr(self.loss_fn, "value_and_grad", None)
        if vg is not None:
            return vg(img)
        leaf = img.detach().float().requires_grad_(True)
        loss = self.loss_fn(leaf)
        (g,) = torch.autograd.grad(loss, leaf)
        return loss.detach(), g

    def _adam_state(self, rays):
        """exp_avg / exp_avg_sq / step of torch.optim.Adam for `rays` (created like Adam._init_group does), so that
        step() and step_autograd() — and anything that inspects self.opt — share one optimiser state"""
        st = self.opt.state[rays]
        if len(st) == 0:
            st["step"] = torch.zeros((), dtype=torch.float32, device=rays.device)
            st["exp_avg"] = torch.zeros_like(rays, memory_format=torch.preserve_format)
            st["exp_avg_sq"] = torch.zeros_like(rays, memory_format=torch.preserve_format)
        g = self.opt.param_groups[0]
        if g.get("weight_decay", 0) != 0 or g.get("amsgrad", False) or g.get("maximize", False):
            raise NotImplementedError("PatternOptimizer.step: plain Adam only (use step_autograd for other settings)")
        if not (isinstance(st["step"], torch.Tensor) and st["step"].is_cuda):
            st["step"] = torch.as_tensor(float(st["step"]), dtype=torch.float32, device=rays.device)
        return st, g

    @torch.no_grad()
    def step(self):
        """One optimisation step over `samples_per_step` scene samples (sharded over ranks).

        Same arithmetic, kernels and summation order as `step_autograd` (the tests compare the two
        trajectories), but every adjoint is called directly and the pattern side is three fused launches:
        ffx_pattern_fwd (K1 + K2 sum + K2 softor + the regulariser's partial sums), ffx_pattern_bwd (K2-bwd of the
        data and regulariser terms + K1-bwd) and ffx_adam_clamp_step (Adam + clamp_to_fov + normalize_rays); K9
        accumulates every sample's texture gradient into one buffer.  As ~30 separate launches the pattern side
        cost 0.11 ms per step next to a 0.75 ms render; through torch.autograd the step was host-bound."""
        from . import ops

        rays = self.laser._rays
        KF = self.laser._KF
        s0, s1 = self.tex_size
        S = self.samples_per_step
        r, w = dist.rank(), dist.world_size()
        geom, ms = self.mi_scene.geom, self.mi_scene
        want_reg = self.reg_weight > 0
        rd = rays.detach()
        # pattern -> texture (K1, K2, K3)
        buf = getattr(self, "_pat_buf", None)
        if buf is None or buf[0].shape[0] != rd.shape[0] or tuple(buf[1].shape) != (s1, s0) or (buf[2] is None) == want_reg:
            buf = None
        # the step's accumulator: the texture gradient, then the data term's partial sums (one slot per 8x8-pixel block of the film:
        # K9 adds <gimg, img> of its block to its own slot, other losses add to slot 0) — cleared by the pattern launch, summed by pattern_bwd
        # The adjoint cache sits right behind them in ONE allocation, so that the same launch also clears the cache's 64-byte header
        # (FFX_RENDER_CACHE_ZEROED: the step's first render then has nothing to reset — with the apex records written behind the
        # re-fit, ops.DeviceGeometry.update, it launches no pre-pass at all).
        cam = ms.data.camera
        linear = getattr(self.loss_fn, "linear_gradient", None)
        sd0 = ms.scene_desc(tex_channels=1)  # (sizes only: the pose of the samples comes later)
        # a loss that is linear in the image (its gradient does not depend on the render): forward and adjoint are ONE launch
        # (ffx_render_fwd_adjoint) — no cache, no K9; <gimg, img> goes to _abi.ADJOINT_DOT_SLOTS partial sums
        # (one process: the loss value then comes out of the gradient launch, from the step's renders stacked in one buffer; the fused
        # launch's own per-pixel partial sums would cost more than K9 — tools/adjprobe.py, 519 against 510 us per sample.  Several ranks
        # take the same launch; their gradient launch only evaluates the data term, the exchange and the update follow — below.
        # A filtered film (sd.rfilter) takes the same route through ffx_render_fwd_adjoint_filtered; with a non-linear loss it re-traces)
        fused = (linear is not None and int(sd0.n_base_tex) == 0 and bool(sd0.proj.enabled) and 1 <= len(self._sample_seeds(self.step_index)) <= 64
                 and os.environ.get("FFX_FUSED_ADJOINT", "1") != "0" and not ops.deterministic_mode())  # (FFX_DETERMINISTIC=1: ffx_render_bwd_det for every sample)
        # (round 5) under a gaussian film the fused launch is NOT the fast route: it needs two launches in front of the render (the weights every pixel
        # will receive, G = gimg / weight) and forms every lit sample's 25-term gradient inside K8 — 0.74 ms per sample against 0.57 for the filtered
        # forward that stores its per-sample records + the adjoint from them (tools/rfgrad.py).  A linear loss takes that pair too; its value
        # <gimg, img> still comes out of the gradient launch (the step's renders stacked in one buffer, as for the fused launch).
        lin_rf = fused and bool(sd0.rfilter) and Fn.cache_supported(sd0, self.spp) and os.environ.get("FFX_FUSED_ADJOINT_FILTERED", "0") != "1"
        if lin_rf:
            fused = False
        n_slots = ops.render_dot_slots(cam.width, cam.height)  # (K9's partial sums of the loss; the fused path evaluates it in the gradient launch)
        use_cache = (not fused) and Fn.cache_supported(sd0, self.spp) and not getattr(self, "_cache_overflowed", False)
        nbytes = ops.render_cache_bytes_sd(sd0, self.spp) if use_cache else 0
        acc_bytes = -(-4 * (s0 * s1 + n_slots) // 128) * 128
        if getattr(self, "_arena", None) is None or self._arena.numel() != acc_bytes + max(nbytes, 64):
            self._arena = torch.empty(acc_bytes + max(nbytes, 64), dtype=torch.uint8, device=rd.device)
            self._acc = self._arena[: acc_bytes + 64].view(torch.float32)  # what the pattern launch clears: accumulator + cache header
            self._cache = self._arena[acc_bytes:] if use_cache else None
        # (round 6) the texture of this step may already be there: the previous step's pattern launch (ffx_pattern_step) went on, behind its
        # update, to K1 + K2 + K3 of the updated pattern and cleared the accumulator — if the pattern, the buffers and the settings are still
        # the ones it saw (_premade_key; the launch itself checks the pattern's bits against what it kept: `stale`, read by _watch_cache)
        pre, self._premade = getattr(self, "_premade", None), None
        used_premade = False
        if self.blur:  # K1 + K2 + K3 in one launch (the blur rides on the splat's tiles)
            if buf is not None and len(buf) != 5:
                buf = None
            if pre is not None and buf is not None and pre == self._premade_key(rays, KF, want_reg, buf):
                pts, tsum, tsor, ws, tex = buf
                used_premade = True
            else:
                pts, tsum, tsor, ws, tex = self._pat_buf = ops.pattern_fwd_blur(rd, KF, self.sigma, s0, s1, self.blur[0], self.blur[1], want_softor=want_reg, out=buf, zero=self._acc)
        else:
            if buf is not None and len(buf) != 4:
                buf = None
            pts, tsum, tsor, ws = self._pat_buf = ops.pattern_fwd(rd, KF, self.sigma, s0, s1, want_softor=want_reg, out=buf, zero=self._acc)
            tex = tsum
        tex3 = tex.unsqueeze(-1)
        gtex, loss_slots = self._acc[: s0 * s1].view(tex3.shape), self._acc[s0 * s1: s0 * s1 + n_slots]
        loss_sum = loss_slots[0]
        header_clear = True  # (until the first cache-writing render of the step has used it)
        # this rank's scene samples: all their random draws up front (each under its own seed, as
        # manual_seed(s); randomize() would make them), ONE device-to-host transfer for the lot
        seeds = self._sample_seeds(self.step_index)
        ahead, self._ahead = getattr(self, "_ahead", None), None
        appliers = None
        if ahead is not None and ahead[0] == (self.step_index, tuple(seeds)):
            try:
                appliers = ahead[1]()
            except StaleDrawError:  # a sampler range was changed since: draw again (anything else — a failed native call, a HIP error — propagates)
                appliers = None
        if appliers is None:
            appliers = self.ff_scene.randomize_batch(seeds)
        # the NEXT step's samples are drawn now, ahead of this step's renders: device draws issued behind a
        # render that fills the GPU only complete when it ends, and waiting for them would serialise host and GPU.
        # Their seeds are a function of the step index alone; the generators are put back afterwards.
        if self.ff_scene._draw_stream() is not None and not self.ff_scene._host_drawable():  # (draws evaluated on the host have nothing to wait for)
            nxt = self._sample_seeds(self.step_index + 1)
            self._ahead = ((self.step_index + 1, tuple(nxt)), self.ff_scene.randomize_batch(nxt, lazy=True))
        fast_loss = getattr(self.loss_fn, "accumulate_value_and_grad", None)
        k_sample = 0
        used_rs = []
        det = ops.deterministic_mode() and bool(sd0.proj.enabled)
        if det:
            # FFX_DETERMINISTIC=1 (round 6): a step whose result does not depend on the number of ranks, BIT FOR BIT.  Every sample's texture gradient
            # is re-traced twice (ffx_render_bwd_det_part): first for the largest tap — maximum over this rank's samples, then over the ranks —, from
            # which every rank derives the SAME power-of-two scale; then as 64-bit fixed-point sums into one buffer per rank, summed over the ranks
            # as integers.  Integer sums are the same in any order and any grouping, the per-sample seeds do not depend on the world size
            # (dist.sample_seed), and from the summed gradient on every rank runs the same launches on the same bits: pattern, Adam state and loss
            # of a 1-, 2-, 4- or 8-rank run are equal.  (The float exchange of the default mode agrees to ~1e-7, not to the bit.)
            vmax = torch.zeros(1, dtype=torch.int32, device=rd.device)
            lossv = torch.zeros(max(S, 1), dtype=torch.float32, device=rd.device)  # one slot per sample of the STEP: each rank fills its own
            mine = dist.sample_ids(S, r, w)
            kept = []
            for kg, seed, apply_sample in zip(mine, seeds, appliers):
                apply_sample()
                sd = ms.scene_desc(tex_channels=1)
                mats = ms.materials_arg(sd)
                img = geom.render_fwd(sd, mats, tex3, self.spp, seed, False)
                if fast_loss is not None:
                    gimg = fast_loss(img, lossv[kg])
                else:
                    with torch.enable_grad():
                        l, gimg = self._loss_and_grad(img)
                    lossv[kg] += l
                    gimg = gimg.float().contiguous()
                geom.render_bwd_det_part(sd, mats, self.spp, seed, gimg, 1, vmax)
                kept.append((seed, apply_sample, gimg))
                self.step_paths["retrace"] += 1
            dist.allreduce_max_(vmax)
            dist.allreduce_sum_(lossv)  # (x + 0 is x: the slots arrive as their owners wrote them)
            sh = ops.det_scale_log2(int(vmax.item()), 4 * cam.width * cam.height * self.spp * max(S, 1))
            fix = torch.zeros(tex3.shape, dtype=torch.int64, device=rd.device)
            if sh is not None:
                for seed, apply_sample, gimg in kept:
                    apply_sample()  # (the pose again: the second re-trace of this sample)
                    sd = ms.scene_desc(tex_channels=1)
                    geom.render_bwd_det_part(sd, ms.materials_arg(sd), self.spp, seed, gimg, 2, fix, scale_log2=sh)
            dist.allreduce_sum_(fix)
            if sh is not None:
                ops.det_finish_(fix, sh, gtex)
            loss_sum += lossv.double().sum().float()  # (the same S floats in the same order on every rank, whatever the world size)
            seeds_run, seeds = seeds, []  # (the sample loop below has nothing left to do)
        for seed, apply_sample in zip(seeds, appliers):
            apply_sample()  # host 4x4 algebra + K5/K6 on the side stream
            sd = ms.scene_desc(tex_channels=1)
            if fused:  # K8 scatters the pixel's footprint x gimg straight into gtex and adds <gimg, img> to the loss slots
                if getattr(self, "_lin_g", None) is None or tuple(self._lin_g.shape) != (cam.height, cam.width, 3):
                    self._lin_g = linear(torch.empty((cam.height, cam.width, 3), device=tex.device)).float().contiguous()  # (constant by definition)
                # (<gimg, img>: with ONE sample per step the gradient launch evaluates it from the image — 3 us; K8's own partial sums are
                # a quarter of a million atomics per render, 27 us, and only used when a step has several samples)
                if getattr(self, "_img_stack", None) is None or tuple(self._img_stack.shape) != (len(seeds), cam.height, cam.width, 3):
                    self._img_stack = torch.empty((len(seeds), cam.height, cam.width, 3), dtype=torch.float32, device=tex.device)
                # (round 6) a step of SEVERAL samples: their fused launches are independent of each other — each adds its adjoint to gtex with atomics and
                # writes its own image of the stack — and take the scene's two render streams in turn, as consecutive mi.render calls do: the tail of one
                # launch runs beside the head of the next (FFX_STEP_STREAMS=1: one after the other on the caller's stream)
                rs_list = getattr(ms, "_render_streams", None)
                if len(seeds) > 1 and rs_list is not None and os.environ.get("FFX_STEP_STREAMS", "2") != "1":
                    if k_sample == 0:
                        step_ready = torch.cuda.Event()
                        step_ready.record()  # (the texture, the cleared accumulator, the constant gradient: everything issued on the caller's stream so far)
                    rs = rs_list[k_sample & 1]
                    if rs not in used_rs:
                        rs.wait_event(step_ready)
                        used_rs.append(rs)
                    with torch.cuda.stream(rs):
                        geom.render_fwd_adjoint(sd, ms.materials_arg(sd), tex3, self.spp, seed, self._lin_g, out=gtex, sparse_adjoint=True, img_out=self._img_stack[k_sample])
                else:
                    geom.render_fwd_adjoint(sd, ms.materials_arg(sd), tex3, self.spp, seed, self._lin_g, out=gtex, sparse_adjoint=True, img_out=self._img_stack[k_sample])
                k_sample += 1
                self.step_paths["fused"] += 1
                continue
            # the pattern gradient flows through the splat that produced this texture: texels whose value is exactly zero
            # (no splat within reach, nothing for the blur to spread) have no influence on it — sparse adjoint
            mats = ms.materials_arg(sd)  # (None: the rows are part of sd — no upload, no device tensor)
            if lin_rf and use_cache:  # linear loss, filtered film: records + their adjoint under the constant gradient; the image joins the step's stack
                if getattr(self, "_lin_g", None) is None or tuple(self._lin_g.shape) != (cam.height, cam.width, 3):
                    self._lin_g = linear(torch.empty((cam.height, cam.width, 3), device=tex.device)).float().contiguous()
                if getattr(self, "_img_stack", None) is None or tuple(self._img_stack.shape) != (len(seeds), cam.height, cam.width, 3):
                    self._img_stack = torch.empty((len(seeds), cam.height, cam.width, 3), dtype=torch.float32, device=tex.device)
                # (the pattern launch has cleared the cache's header: the step's first render has nothing to reset — no k_cache_reset launch in front of it)
                geom.render_fwd(sd, mats, tex3, self.spp, seed, False, cache=self._cache, sparse_adjoint=True, img_out=self._img_stack[k_sample],
                                cache_zeroed=header_clear, keep_dropped=not header_clear)
                header_clear = False
                geom.render_bwd_cached(sd, mats, self._cache, self.spp, self._lin_g, out=gtex, seed=seed)
                k_sample += 1
                self.step_paths["cache_k9"] += 1
                continue
            # (the cache is reused by the step's samples one after the other: only the first render of the step may clear the header's count of
            # dropped samples — the in-kernel Adam guard and _watch_cache read it at the END of the step, FFX_RENDER_CACHE_KEEP_DROPPED)
            img = geom.render_fwd(sd, mats, tex3, self.spp, seed, False, cache=self._cache if use_cache else None, sparse_adjoint=use_cache,
                                  cache_zeroed=header_clear, keep_dropped=not header_clear)
            header_clear = False
            self.step_paths["cache_k9" if use_cache else "retrace"] += 1
            if linear is not None and use_cache and not sd.rfilter:
                # loss(img) = <gimg, img>: K9 adds it to loss_sum while it scatters the footprints (no reduction launch)
                geom.render_bwd_cached(sd, mats, self._cache, self.spp, linear(img), out=gtex, img=img, dot_out=loss_slots)
                continue
            l1t = getattr(self.loss_fn, "l1_target", None)
            if l1t is not None and use_cache and not sd.rfilter and os.environ.get("FFX_K9_L1", "1") != "0":
                # the reference's own loss, L1 against a target image: K9 forms sign(img - target) / n per pixel itself and adds the loss value to the step's
                # slots — no loss launches (two), no gradient image (ffx_render_bwd_cached_l1; declined cases take the general route below)
                if geom.render_bwd_cached_l1(sd, mats, self._cache, self.spp, img, l1t[0], l1t[1], gtex, loss_slots) is not None:
                    continue
            if fast_loss is not None:
                gimg = fast_loss(img, loss_sum)
            else:
                with torch.enable_grad():
                    l, gimg = self._loss_and_grad(img)
                loss_sum += l
                gimg = gimg.float().contiguous()
            if use_cache:
                geom.render_bwd_cached(sd, mats, self._cache, self.spp, gimg, out=gtex, seed=seed if sd.rfilter else None)
            else:
                gtex += geom.render_bwd(sd, mats, self.spp, seed, gimg).reshape(gtex.shape)
        for rs in used_rs:  # (the step's renders on the render streams: the gradient launch waits for them)
            torch.cuda.current_stream().wait_stream(rs)
        # back through K3^T, K2-bwd, K1-bwd for this rank's share; the regulariser depends on the pattern only
        # K3^T is applied inside the gradient launch, over the points' footprints only (ffx_pattern_bwd_blur: the gradient of the separate
        # transpose blur + ffx_pattern_bwd, bit for bit)
        if det:
            seeds = seeds_run
        g2 = gtex.reshape(tex.shape) if (seeds or det) else None
        bk, bs = (self.blur[0], self.blur[1]) if self.blur else (0, 1.0)
        reg_w = self.reg_weight if want_reg else 0.0
        st, g = self._adam_state(rays)
        grad = torch.empty_like(rd)
        if getattr(self, "_scratch", None) is None or self._scratch.shape != tsum.shape:
            self._scratch = torch.empty_like(tsum)  # (only touched when a footprint does not fit the workgroup's LDS)
        if getattr(self, "_adam_counter", None) is None:
            self._adam_counter = torch.zeros(1, dtype=torch.int32, device=rd.device)
        dot = None
        if (fused or (lin_rf and use_cache)) and seeds:
            if getattr(self, "_dot_part", None) is None or self._dot_part.numel() < rd.shape[0]:
                self._dot_part = torch.empty(rd.shape[0], dtype=torch.float32, device=rd.device)
            dot = (self._img_stack, self._lin_g, self._dot_part)  # <gimg, img_k> summed over the step's renders (gimg repeated)
        if (w > 1 or dist.exchanging()) and not det:  # (det: the texture gradient has been exchanged — as integers; every rank holds the step's sum)
            # (several ranks: this rank's data term from the gradient launch — Adam arguments without state: no update —, then the exchange)
            aa = ops.adam_args(rd, None, None, None, self._adam_counter, 0.0, 0.0, 0.0, 0.0, self.laser._KF_inv, 0.0, 1.0, dot=dot) if dot is not None else None
            gd, gr, val = ops.pattern_bwd_blur(rd, KF, self.sigma, s0, s1, tsum, tsor, g2, reg_w, ws, bk, bs, loss_in=None if dot is not None else loss_slots,
                                               loss_div=float(S), adam=aa, scratch=self._scratch)
            # the ONE exchange of a step: [3N + 2] floats — the gradient, this rank's data term (val[2]) and (round 5) its adjoint cache's count of
            # dropped samples: K9 has then poisoned THIS rank's gradient with NaN, and after the sum every rank knows.  The update launch reads the
            # summed count as its guard (ffx_adam_clamp_step: the word at byte 8 of `guard` = the buffer's last float; any non-zero bit pattern,
            # a NaN's included, skips): no rank applies a poisoned update, rays and Adam state stay identical across ranks
            dropped = (self._cache[8:12].view(torch.int32).float() if (use_cache and self._cache is not None) else torch.zeros(1, device=rd.device))
            flat = torch.cat([(gd if gd is not None else torch.zeros_like(rd)).reshape(-1), val[2:3], dropped])
            dist.allreduce_sum_(flat)
            n3 = 3 * rd.shape[0]
            gsum = flat[:n3].reshape(rays.shape).contiguous()
            loss = flat[n3] / float(S) + val[0]
            # grad = gsum / S (+ regulariser, identical on every rank); Adam; Laser.clamp_to_fov() + normalize_rays()
            ops.adam_clamp_step_(rd, gsum, st["exp_avg"], st["exp_avg_sq"], st["step"], g["lr"], g["betas"][0], g["betas"][1], g["eps"], KF, self.laser._KF_inv,
                                 1 - 0.95, 0.95, 2, grad_b=gr, grad_div=float(S), grad_out=grad, guard=flat[n3 - 1:])
            self.laser._edits = getattr(self.laser, "_edits", 0) + 1
            self._last_flat = flat  # (kept alive until the update has run; tests read the exchanged count)
        else:
            # nothing to exchange: the whole backward half is ONE launch — gradient of the data term and of the regulariser, the step's total
            # loss, and (by the workgroup that finishes last) Adam + Laser.clamp_to_fov() + normalize_rays() on grad = gsum / S + regulariser
            # (guard: with the adjoint cache in play the in-kernel update is skipped when its header reports dropped samples — K9 has then
            # poisoned the gradient with NaN; rays and the Adam state stay intact and _watch_cache switches this optimiser to the re-tracing adjoint)
            aa = ops.adam_args(rd, st["exp_avg"], st["exp_avg_sq"], st["step"], self._adam_counter, g["lr"], g["betas"][0], g["betas"][1], g["eps"], self.laser._KF_inv,
                               1 - 0.95, 0.95, 2, grad_div=float(S), grad_out=grad, dot=dot, guard=self._cache if use_cache else None)
            res = None
            self._merged_last = False
            if self.blur and bk == 5 and g2 is not None and os.environ.get("FFX_PATTERN_STEP", "1") != "0":
                # ... and (round 6) the NEXT step's K1 + K2 + K3 behind the update, in the same launch (ffx_pattern_step): between two renders of a
                # one-sample step there is ONE launch.  The library declines footprints that do not fit its LDS window: the two launches then
                if getattr(self, "_rays_kept", None) is None or tuple(self._rays_kept.shape[1:]) != tuple(rd.shape):
                    # (the sync words, the kept pattern and the count of launches on them belong together: flags that still hold an old epoch must
                    # never meet a count that starts again)
                    self._pat_sync = torch.zeros(_abi.PATTERN_SYNC_BYTES, dtype=torch.uint8, device=rd.device)
                    self._rays_kept = torch.empty((2,) + tuple(rd.shape), dtype=torch.float32, device=rd.device)
                    self._pat_epoch = 0
                    used_premade = False  # (nothing kept to compare with)
                res = ops.pattern_step(rd, KF, self.sigma, s0, s1, self._pat_buf, g2, reg_w, bk, bs, aa, self._acc, self._pat_sync, rays_kept=self._rays_kept,
                                       check_kept=used_premade, loss_in=None if dot is not None else loss_slots, loss_div=float(S),
                                       epoch=self._pat_epoch % 0xFFFFFFF0 + 1)
                if res is not None:
                    self._pat_epoch += 1
                    self.laser._edits = getattr(self.laser, "_edits", 0) + 1  # (this update, counted before the key is taken: ANOTHER optimiser's is not in it)
                    self._premade = self._premade_key(rays, KF, want_reg, self._pat_buf)
                    self._merged_last = True
            if res is None:
                res = ops.pattern_bwd_blur(rd, KF, self.sigma, s0, s1, tsum, tsor, g2, reg_w, ws, bk, bs, loss_in=None if dot is not None else loss_slots,
                                           loss_div=float(S), adam=aa, scratch=self._scratch)
                self.laser._edits = getattr(self.laser, "_edits", 0) + 1  # (a native update of the pattern: torch's version counter does not see it)
            gd, gr, val = res
            loss = val[1]
        rays.grad = grad
        self.step_index += 1
        self._watch_cache(every=1 if self.step_index <= 4 else 32)  # (an arena that is too small shows in the first steps)
        return {"loss": loss}

    def _premade_key(self, rays, KF, want_reg, buf):
        """what must be unchanged for the texture made by the previous step's pattern launch to be THIS step's texture: the pattern tensor (storage and
        torch's version counter: in-place edits through torch bump it; the native update does not), the laser's own edit count, the accumulator the launch
        cleared, the five texture buffers, and every setting of K1-K3.  (Edits torch cannot see — `rays.data` written in place — are caught on the device:
        ffx_pattern_step compares the pattern's bits with the ones it kept and raises `stale`, which _watch_cache turns into an error.)"""
        return (rays.data_ptr(), rays._version, tuple(rays.shape), getattr(self.laser, "_edits", 0), self._acc.data_ptr(), self._acc.numel(),
                tuple(b.data_ptr() if b is not None else 0 for b in buf), float(self.sigma), tuple(self.tex_size), tuple(self.blur) if self.blur else None, bool(want_reg),
                KF.tobytes() if hasattr(KF, "tobytes") else tuple(float(v) for v in torch.as_tensor(KF).reshape(-1).tolist()))

    def invalidate_texture(self):
        """forget the texture the last step made ahead for the next one (call after editing the pattern behind torch's back, e.g. through `rays.data`)"""
        self._premade = None

    def _watch_cache(self, every=32):
        """The adjoint cache is lossy once its arena of single-sample records is full (include/ffx.h
        ffx_render_cache_status; K9 then poisons the gradient with NaN).  Every `every` steps the 64-byte cache header is
        copied to pinned host memory behind the step's kernels and inspected once it has landed — the host never waits."""
        w = getattr(self, "_watch", None)
        if w is not None and w[1].query():
            words = w[0].tolist()
            self._watch = None
            if w[3]:  # (behind ffx_pattern_step: its sync words — `stale` in word 4, the header as the step left it in words 18..33)
                if words[5]:
                    raise RuntimeError(f"PatternOptimizer: a pattern launch around step {w[2]} gave up waiting for its own update (ffx_pattern_step's `timeout`): "
                                       "the texture of the step after it is incomplete.  Set FFX_PATTERN_STEP=0 and report this.")
                if words[4]:
                    raise RuntimeError(
                        f"PatternOptimizer: the pattern was edited in place between two steps in a way torch does not record (around step {w[2]}: e.g. through "
                        "`laser._rays.data`), and the step after the edit rendered with the texture of the unedited pattern.  Call invalidate_texture() after "
                        "such an edit (or set FFX_PATTERN_STEP=0).")
                words = words[18:34]
            used, cap, dropped = words[:3] if w[4] else (0, 0, 0)
            if dropped:
                import warnings

                # the steps since then were NOT applied (ffx_adam_args.guard / ffx_adam_clamp_step's guard: the update launch skips when the header
                # — with several ranks: the exchanged sum of the ranks' headers — reports drops), so the optimiser state is intact and identical on
                # every rank.  One process switches to the re-tracing adjoint by itself; several ranks cannot decide that alone (a rank whose
                # OWN cache never overflowed would keep the cache: the ranks' launches must stay in step), so they raise — with clean state.
                if dist.world_size() > 1:
                    raise Fn.CacheOverflowError(
                        f"PatternOptimizer: the adjoint cache of step {w[2]} overflowed on this rank ({dropped} samples beyond its {cap} single-sample "
                        "records); the updates since then were skipped on every rank (rays and Adam state are intact). Set FFX_CACHE_LIMIT_GB=0 "
                        "(re-tracing adjoint) on all ranks and continue.")
                if getattr(self.mi_scene, "_rfilter", None) is not None and not getattr(self.mi_scene, "_cache_dense", False):
                    # the filtered film's cache: an arena with a share of the blocks (a quarter beyond 2^18) — a pattern that lights more of the film than
                    # that gets the dense layout (FFX_SHADOWS_CACHE_DENSE: cannot overflow) instead of the re-tracing adjoint
                    self.mi_scene.set_cache_dense(True)
                    self._arena = None
                    self._premade = None  # (the accumulator moves with the arena)
                    warnings.warn(f"PatternOptimizer: the filtered film's adjoint cache of step {w[2]} overflowed ({dropped} blocks beyond its {cap}); the updates of the "
                                  "affected steps were skipped; from now on the cache keeps a block for every pass of every pixel.", stacklevel=3)
                    return
                self._cache_overflowed = True
                self._arena = None
                warnings.warn(f"PatternOptimizer: the adjoint cache of step {w[2]} overflowed ({dropped} samples beyond its {cap} single-sample records: a projector "
                              "texture much finer than the camera's pixels, or grazing views).  The updates of the affected steps were skipped; this optimiser "
                              "now uses the re-tracing adjoint (ffx_render_bwd).", stacklevel=3)
        merged = getattr(self, "_merged_last", False)
        if getattr(self, "_watch", None) is None and (self._cache is not None or merged) and (self.step_index - 1) % every == 0:
            pin = getattr(self, "_watch_pin", None)
            if pin is None:
                pin = self._watch_pin = torch.empty(64, dtype=torch.int32, pin_memory=True)
            if merged:  # (the launch has cleared the cache's header for the next step; it kept a copy)
                pin.copy_(self._pat_sync[:256].view(torch.int32), non_blocking=True)
            else:
                pin[:16].copy_(self._cache[:64].view(torch.int32), non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            self._watch = (pin, ev, self.step_index - 1, merged, self._cache is not None)

    # ------------------------------------------------------------------ the same step through autograd
    def step_autograd(self):
        """reference implementation of `step` on torch.autograd (kept for validation and for task
        losses / pipelines that need the graph)."""
        self.opt.zero_grad(set_to_none=True)
        rays = self.laser._rays
        pts, tsum, tex = self.textures()
        S = self.samples_per_step
        r, w = dist.rank(), dist.world_size()
        gtex = torch.zeros_like(tex)
        loss_sum = torch.zeros((), device=tex.device)
        for k in dist.sample_ids(S, r, w):
            g, l = self._render_sample(tex, dist.sample_seed(self.base_seed, self.step_index, S, k))
            gtex += g
            loss_sum += l
        # back through K3^T, K2-bwd, K1-bwd for this rank's share
        tex.backward(gtex, retain_graph=self.reg_weight > 0)
        flat = torch.cat([rays.grad.reshape(-1), loss_sum.reshape(1)])
        dist.allreduce_sum_(flat)
        flat /= float(S)
        rays.grad = flat[:-1].reshape(rays.shape).clone()
        loss = flat[-1]
        if self.reg_weight > 0:  # identical on every rank (depends on the pattern only)
            s0, s1 = self.tex_size
            tsor = Fn.splat(pts, self.sigma, s0, s1, "softor", -1)
            reg = self.reg_weight * (tsor - tsum).abs().mean()
            (greg,) = torch.autograd.grad(reg, rays)
            rays.grad += greg
            loss = loss + reg.detach()
        self.opt.step()
        self.laser.clamp_to_fov()
        self.laser.normalize_rays()
        self.step_index += 1
        return {"loss": loss}
