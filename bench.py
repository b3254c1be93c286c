#!/usr/bin/env python3
"""bench.py — renders/sec and pattern-gradient-steps/sec at 512x512, 64 spp on the vocal-fold
scene (BASELINE.json metric), one process per GPU.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Without a torchrun environment `--gpus N` (N > 1) makes this process a LAUNCHER: it starts N fresh rank processes
(RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, one per device, backend nccl = RCCL), waits for them and exits with
the first non-zero code; it never touches a GPU itself and never restarts a rank.  More ranks than devices is refused
unless FFX_DIST_BACKEND=gloo asks for a single-device rehearsal of the N > 1 path.

A "step" of the headline number is one pass of the render hot path over one randomised scene:
    ff_scene.randomize()  ->  params.update() [K5+K6: vertex transform + BVH refit]
    mi.render(scene, spp) [K8]
with the laser texture (K1+K2+K3) built once before the loop, exactly like
examples/vocalfold_scene.py:56-69,100-102.  Inputs are resident in HBM before the timed region.
Ranks render disjoint scene samples with no communication (weak scaling); value = all renders of
all ranks / max-over-ranks time.  A second bracket times full pattern-gradient steps
(fireflies_amd.optim.PatternOptimizer: K1-K3 fwd, K5+K6, K8, K9, K3^T, K2-bwd, K1-bwd, one
all-reduce of [3N+1], Adam, clamp_to_fov) with one scene sample per rank per step.

Prints ONE JSON line on rank 0 (contract in the task description) with `roofline` (dominant kernel
= ffx render_fwd, live HIP-event timing on the launch stream) and `cpu_baseline` (the scalar CPU
oracle on a bounded sample of the same workload; test infrastructure, used here only as the
reported baseline).
"""
import argparse
import json
import os
import re
import random
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

dist = mi = workloads = PatternOptimizer = None  # the product, imported by the ranks only (_load_product)


def _load_product():
    """imports fireflies_amd (and with it libffx_hip.so).  Only a RANK does this: the launcher process of
    `python bench.py --gpus N` (launch_ranks) never loads the HIP library and never touches a GPU."""
    global dist, mi, workloads, PatternOptimizer
    from fireflies_amd import dist as _d, mi as _m, workloads as _w
    from fireflies_amd.optim import PatternOptimizer as _p

    dist, mi, workloads, PatternOptimizer = _d, _m, _w, _p

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)


def _max_over_ranks(x, device):
    t = torch.tensor([x], dtype=torch.float64, device=device)
    if dist.world_size() > 1:
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    return float(t.item())


def _images_agree(a, b, spp, what):
    """the bound of tests/conftest.assert_image_close: all but 2e-4 of the pixel channels within 1e-4 of the image scale, none off by more
    than one or two samples' worth"""
    a, b = a.float(), b.float()
    scale = float(b.max())
    err = (a - b).abs()
    bad = float((err > 1e-4 * scale).float().mean())
    if not (scale > 0 and bad <= 2e-4 and float(err.max()) <= 1.5 * scale / spp):
        raise SystemExit(f"bench.py preflight: {what}: {bad:.2e} of the pixel channels differ, worst {float(err.max()) / max(scale, 1e-30):.3g} of the scale — not timing a kernel whose output is in doubt")
    return bad


# The GPU's clocks follow its load: after an idle phase (the host-side scene set-up, the garbage collection in front of a bracket) the first
# ~40 launches of the render kernel run up to 6 % slower than the ones after them, whatever ran before the idle phase, and a latency-bound
# kernel does not bring them up (tools/posecost.py, profiles/r3_posecost.txt: ten-step windows 0.580 0.563 0.547 0.541 ... after 0.3 s of
# idle; 0.533 0.547 0.545 ... when 48 back-to-back launches of the same kernel precede the loop).  A 20-step bracket with 5 warm-up steps
# sits entirely inside that ramp and reads the ramp, not the kernel.  SETTLE_RENDERS launches of the render kernel on the scene's current
# pose (no randomisation, no re-fit, nothing of a step but the kernel: ~26 ms) are therefore issued right in front of the W warm-up steps
# of each bracket; the line says so ("clock_settle").  FFX_BENCH_SETTLE=0 turns it off (the cold-start figure: ~4 % lower over 20 steps,
# the same over 100).  The timed region is untouched: exactly K steps between barrier + synchronize on both sides.
SETTLE_RENDERS = int(os.environ.get("FFX_BENCH_SETTLE", "48"))


def _bracket(fn, steps, warmup, device, preflight=None, settle=None):
    """W untimed steps, then exactly K timed steps between barrier + synchronize on both sides.
    The cyclic garbage collector is paused over the timed steps, as timeit does: a generation-2 pass
    over the scene graphs built during set-up stalls the host for ~70 ms (seen once per ~200 steps),
    which is 60 steps' worth of GPU work."""
    import gc

    # collect BEFORE the warm-up steps: a full collection walks the whole heap and leaves the interpreter's
    # working set cold in the caches — done after the warm-up it made the first timed step 2x slower on the host
    # (1.1 instead of 0.5 ms, tools/startprof.py), which a 20-step bracket sees as +3 % per step
    gc.collect()
    gc.disable()
    try:
        if preflight is not None:
            preflight()
        if settle is not None:
            for k in range(SETTLE_RENDERS):
                settle(k)
        for i in range(warmup):
            fn(i)
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            fn(warmup + i)
        torch.cuda.synchronize()
        _bracket.local_s = time.perf_counter() - t0  # this rank's own K steps (before it waits for the others)
        dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    finally:
        gc.enable()
    return _max_over_ranks(dt, device)


def _kernel_ms(events, name):
    ms = [a.elapsed_time(b) for n, a, b in events if n == name]
    return (sum(ms) / len(ms), len(ms)) if ms else (None, 0)


# Workloads that tools/collect_profiles.sh has committed rocprofv3 passes for, and the key that names them in profiles/:
#   profiles/r<N>_*          bench.py's defaults (configs[1]/[2]: vocal fold, 512x512, 64 spp, principled material)
#   profiles/r<N>colon_*     configs[4] (--workload colon --res 1024 --spp 256 --grid 32 --fp16, principled mucosa)
# (+ "grad" in front of the underscore for the gradient bracket's PMC passes)
PROFILED_WORKLOADS = {
    "": {"workload": "vocalfold", "res": 512, "spp": 64, "fp16": False, "material": "principled"},
    "colon": {"workload": "colon", "res": 1024, "spp": 256, "fp16": True, "material": "principled"},
}
PROFILED_WORKLOAD = PROFILED_WORKLOADS[""]


def profile_key(args):
    """"" / "colon": which committed profile set describes this run's workload; None: none does"""
    if args.no_shadows or getattr(args, "rfilter", "box") != "box":
        return None
    for key, w in PROFILED_WORKLOADS.items():
        if all(getattr(args, k) == v for k, v in w.items()) and (key != "colon" or args.grid == 32):
            return key
    return None


def _is_profiled_workload(args):
    return profile_key(args) == ""


def _profile_files(suffix, key="", grad=False):
    """committed summaries profiles/r<N><key>[grad]_<suffix>, oldest round first"""
    import glob
    import re

    pat = re.compile(r"^r(\d+)" + re.escape(key or "") + ("grad" if grad else "") + "_" + re.escape(suffix) + "$")
    hits = [(int(pat.match(os.path.basename(f)).group(1)), f) for f in glob.glob(os.path.join(ROOT, "profiles", "*_" + suffix)) if pat.match(os.path.basename(f))]
    return [f for _, f in sorted(hits)]


def k8_instance(name):
    """(forward + adjoint in one launch?, filtered film?) of a k_render_fwd_pk<R, WIDE, MATM, ADJ[, RF]> instance name as rocprofv3 prints it"""
    m = re.search(r"k_render_fwd_pk<([^>]*)>", name)
    a = [t.strip() for t in m.group(1).split(",")] if m else []
    return (len(a) > 3 and a[3] == "true", len(a) > 4 and a[4] == "true")


def pmc_traffic(kernel_prefix, tag="r", key="", adjoint_instance=False):
    """HBM bytes per launch of a kernel from the newest committed rocprofv3 PMC passes (profiles/*_pmc_summary.json,
    written by tools/summarize_profile.py: separate --pmc FETCH_SIZE / WRITE_SIZE runs of this same bench at the workload
    `key` names; FETCH_SIZE doubled as MI355X_MICROARCH.md HBM section prescribes for gfx950).  The caller
    only asks for a workload those passes ran (profile_key).  None if no profile is present."""
    best = None
    # "<round><key>_pmc_summary.json" = render bracket only, "<round><key>grad_pmc_summary.json" = gradient bracket
    for f in _profile_files("pmc_summary.json", key, grad=(tag == "grad")):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        # (of the render kernel's instances: the plain forward `<..., false>` unless the forward + adjoint one `<..., true>` is asked for)
        for k, v in sorted(d.items(), key=lambda kv: k8_instance(kv[0])[0] != adjoint_instance):
            if adjoint_instance and not k8_instance(k)[0]:
                continue
            if k.startswith(kernel_prefix) and "hbm_bytes_corrected" in v:
                best = {"bytes": v["hbm_bytes_corrected"], "raw_bytes": v["hbm_bytes_raw"], "source": os.path.basename(f)}
                break
    return best


def step_traffic(key=""):
    """HBM bytes one RENDER STEP moves, everything it launches included — the pose's re-fit, the pre-pass (apex records + tile bins) and the render
    kernel — from the newest committed PMC passes of the render bracket: {kernel: corrected bytes per launch x launches per step}, their
    sum, and the step's algorithmic bytes are reported next to it (`traffic_ratio`).  Launches per step = a kernel's launch count over the
    render kernel's in the same pass."""
    ff = _profile_files("pmc_summary.json", key, grad=False)
    if not ff:
        return None
    try:
        d = json.load(open(ff[-1]))
    except Exception:
        return None
    k8 = [(k, v) for k, v in d.items() if k.startswith("k_render_fwd_pk") and "hbm_bytes_corrected" in v]
    if not k8:
        return None
    n_render = max(1, k8[0][1].get("FETCH_SIZE", {}).get("n", 1))
    per = {}
    for k, v in d.items():
        if "hbm_bytes_corrected" not in v:
            continue
        n = v.get("FETCH_SIZE", {}).get("n", 0)
        if n < n_render:  # (set-up launches: the texture, the first pose)
            continue
        per[k] = {"bytes_per_launch": v["hbm_bytes_corrected"], "launches_per_step": round(n / n_render, 2)}
    total = sum(x["bytes_per_launch"] * x["launches_per_step"] for x in per.values())
    return {"per_kernel": per, "bytes_per_step": total, "source": "profiles/" + os.path.basename(ff[-1])}


# cycles per wave64 instruction per SIMD: the NOMINAL issue rates, which the microbenchmark (tools/ubench/issue_rates.hip,
# profiles/r3_issue_rates.txt; instruction counts fixed by the asm bodies; >= 4 resident waves) approaches from above:
#   fast: v_fma/v_mul/v_add/v_fmac_f32 and v_mov on VGPR operands only ......... 2   (measured 2.17 - 2.47)
#   slow: any VALU op with an SGPR/constant source, min/max/med3, compares, v_cndmask, conversions, integer ops,
#         DPP, v_readlane/v_writelane, packed fp32 ................................. 4   (measured 4.07 - 4.8)
#   transcendental (v_rcp/v_rsq/v_sqrt/v_exp/v_log) ................................... 8   (measured 8.19 - 9.3)
VALU_CYCLES = {"fast": 2.0, "slow": 4.0, "trans": 8.0}
SIMDS, CLOCK_HZ = 1024, 2.4e9


def valu_issue(kernel_substr, kernel_ms, key=""):
    """Compute-side view of the dominant kernel (it is VALU-issue bound, which the contract's hbm|mfma roofline
    cannot express).  Ceiling = the time the kernel's OWN instruction mix needs if every instruction issued at the
    best rate measured for its class: the SQ counters of the newest committed pass (profiles/*_sq_instruction_mix.json,
    same workload) give wave-level counts per launch; every fma/mul/add is priced as the VGPR-only form (2 cycles)
    although many carry a scalar operand (4) — the ceiling is optimistic, so frac = ceiling / live kernel time
    is a lower bound of the issue utilisation and cannot exceed 1."""
    for f in reversed(_profile_files("sq_instruction_mix.json", key)):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        for k, v in sorted(d.items(), key=lambda kv: k8_instance(kv[0])):  # (the plain forward's instance first)
            if kernel_substr in k and "SQ_INSTS_VALU" in v:
                g = lambda c: v.get(c, {}).get("mean", 0.0)  # noqa: E731
                n = g("SQ_INSTS_VALU")
                fast = g("SQ_INSTS_VALU_FMA_F32") + g("SQ_INSTS_VALU_MUL_F32") + g("SQ_INSTS_VALU_ADD_F32")
                trans = g("SQ_INSTS_VALU_TRANS_F32")
                if fast == 0.0:  # a profile without the class counters: price everything at the fast rate
                    fast, trans = n, 0.0
                slow = n - fast - trans
                cyc = fast * VALU_CYCLES["fast"] + slow * VALU_CYCLES["slow"] + trans * VALU_CYCLES["trans"]
                ceil_ms = 1e3 * cyc / SIMDS / CLOCK_HZ
                # ... and with the share of the add/mul/fma class that carries a scalar or constant source (tools/isa_mix.py, from the
                # kernel's disassembly: static) priced at the 4-cycle rate such forms issue at — the figure that says how much a
                # pure scheduling improvement could still buy
                forms = {}
                ff = _profile_files("isa_operand_forms.json", "")
                if ff:
                    try:
                        fr = float(json.load(open(ff[-1]))["scalar_source_fraction"])
                        cyc2 = cyc + fast * fr * (VALU_CYCLES["slow"] - VALU_CYCLES["fast"])
                        # (an ESTIMATE, not a bound: the share is static — every instruction of the kernel once — and the counters come from the
                        # committed passes' poses, the time from this run's: a render costs +-6 % from pose to pose.  Nominal `frac` is the lower
                        # bound of the issue utilisation, this is the upper estimate; the truth lies between them)
                        forms = {"scalar_source_fraction_static": fr, "ceiling_ms_operand_forms": 1e3 * cyc2 / SIMDS / CLOCK_HZ,
                                 "frac_operand_forms": 1e3 * cyc2 / SIMDS / CLOCK_HZ / kernel_ms, "operand_forms_source": "profiles/" + os.path.basename(ff[-1])}
                        if forms["frac_operand_forms"] > 1.0:
                            forms["operand_forms_note"] = ("above 1: the share is STATIC (every instruction of the kernel text once); the loops that dominate since the "
                                                           "tile bins carry fewer scalar-source forms than the text as a whole, so this estimate over-counts — `frac` "
                                                           "(all fma / mul / add at the 2-cycle rate) is the bound")
                    except Exception:
                        forms = {}
                return {"valu_wave_instr_per_launch": n, "fp32_fma_mul_add": fast, "transcendental": trans, "other_valu": slow,
                        "cycles_per_instr": VALU_CYCLES, "simds": SIMDS, "clock_ghz": CLOCK_HZ / 1e9,
                        "ceiling_ms": ceil_ms, "kernel_ms": kernel_ms, "frac": ceil_ms / kernel_ms, **forms,
                        "salu_per_launch": g("SQ_INSTS_SALU"), "smem_per_launch": g("SQ_INSTS_SMEM"), "vmem_per_launch": g("SQ_INSTS_VMEM"),
                        "lds_per_launch": g("SQ_INSTS_LDS"),
                        "source": f"counts: profiles/{os.path.basename(f)} (committed rocprofv3 --pmc passes of this workload, not this run); rates: "
                                  f"profiles/{os.path.basename((_profile_files('issue_rates.txt', '') or ['r2_issue_rates.txt'])[-1])}; time: this run"}
    return None


def cited_profiles(pkey=""):
    """the committed files (profiles/...) a line of this bench cites for workload `pkey` — newest of each kind; tests/test_bench_cpu.py
    checks that each exists, is not empty and names kernels of the current tree"""
    if pkey is None:
        return []
    out = []
    for suffix, grad in (("pmc_summary.json", False), ("pmc_summary.json", True), ("sq_instruction_mix.json", False), ("kernel_stats.csv", False),
                         ("phaseclk.txt", False), ("cpu_oracle_table.json", False)):
        ff = _profile_files(suffix, pkey, grad=grad)
        if ff:
            out.append("profiles/" + os.path.basename(ff[-1]))
    for suffix in ("isa_operand_forms.json", "issue_rates.txt"):
        ff = _profile_files(suffix, "")
        if ff:
            out.append("profiles/" + os.path.basename(ff[-1]))
    return out


def algorithmic_bytes(wl, width, height, fp16=False):
    """K8 render_fwd  B = G + 4*T + 3*s_r*W*H,  G = 12*V + 12*F + 32*N_nodes
    (V = vertices of one pose, F = triangles, N_nodes = BVH nodes, T = texels of the 1-channel
    projector texture, s_r = bytes per radiance channel)."""
    V = sum(m.frames.shape[1] for m in wl.data.meshes)
    F = wl.mi_scene.geom.n_tris
    nodes = wl.mi_scene.geom.info.n_nodes
    G = 12 * V + 12 * F + 32 * nodes
    T = wl.tex_size[0] * wl.tex_size[1]
    s_r = 2 if fp16 else 4
    return {"render_fwd": G + 4 * T + 3 * s_r * width * height, "render_bwd": G + 4 * T + 12 * width * height + 4 * T,
            "scene_update": 24 * V + 32 * nodes, "G": G, "V": V, "F": F, "nodes": nodes}


def _omp_set_threads(n):
    """the oracle is linked against libgomp: set the team size of its `omp parallel for` loops"""
    import ctypes

    try:
        ctypes.CDLL("libgomp.so.1").omp_set_num_threads(int(n))
        return True
    except OSError:
        return False


def _effective_cores():
    """threads this process can actually run at once: the affinity mask, cut down to the cgroup's CPU quota where one is set (the GPU box
    shows 256 logical CPUs to a container that may use 16 of them: 256 OpenMP threads were then 12x one thread, and `cores: 256` wrong)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:  # cgroup v2
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = float(q) / float(per)
    except Exception:
        try:  # cgroup v1
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except Exception:
            pass
    if quota is not None:
        n = max(1, min(n, int(quota + 0.5)))
    return n, quota


def cpu_baseline(wl, tex, spp_full, cpu_spp, seed, grad_wl=None):
    """CPU oracle on the same pose / texture: all host threads (OpenMP over pixels) at `cpu_spp` samples per pixel,
    and ONE thread on a smaller bounded sample; both scaled to the full spp.  Rank 0, N = 1 only."""
    from fireflies_amd import scenes
    from oracle import oracle as orc

    pool, tris, shape, off, stride, nfr, alb = scenes.flatten(wl.data)
    geo = orc.Geometry(wl.mi_scene.geom.src_verts.cpu().numpy(), tris, shape, off)
    t0 = time.perf_counter()
    geo.update(wl.mi_scene._xforms.numpy(), wl.mi_scene._offs)
    t_upd = time.perf_counter() - t0
    sd = wl.mi_scene.scene_desc(tex_channels=1)
    alb, texh = wl.mi_scene.albedo.cpu().numpy(), tex.detach().cpu().numpy()
    cores, quota = _effective_cores()
    _omp_set_threads(cores)
    # whole renders of the same pose (different sample seeds) until >= 10 s of CPU work have been timed
    n, t_r = 0, 0.0
    while n < 8 and t_r < 10.0:
        t0 = time.perf_counter()
        geo.render_fwd(sd, alb, texh, cpu_spp, seed=seed + n)
        t_r += time.perf_counter() - t0
        n += 1
    per_render = t_upd + (t_r / n) * (spp_full / cpu_spp)
    out = {"value": 1.0 / per_render, "unit": "renders/sec", "cores": cores, "kind": "port",
           "sample": f"{n} renders of the same pose at {cpu_spp} of {spp_full} spp ({t_r:.1f} s measured in total, scaled x{spp_full / cpu_spp:g}) + refit "
                     f"{t_upd * 1e3:.1f} ms; gcc -O3 -march=x86-64-v3 scalar oracle (a restatement for checking, not a tuned CPU renderer), OpenMP over pixels on {cores} threads "
                     f"(logical CPUs {os.cpu_count()}, affinity {len(os.sched_getaffinity(0))}, cgroup CPU quota {'none' if quota is None else f'{quota:g}'})",
           "mitsuba_scalar_rgb": "unavailable (mitsuba 3.5.0 / drjit 0.4.4 are not installed here or on the GPU box and are not part of /root/reference)"}
    # one thread: a bounded sample (1/16 of the samples per pixel, >= 1), scaled
    spp1 = max(1, spp_full // 16)
    if _omp_set_threads(1):
        try:
            t0 = time.perf_counter()
            geo.update(wl.mi_scene._xforms.numpy(), wl.mi_scene._offs)
            t_upd1 = time.perf_counter() - t0
            t0 = time.perf_counter()
            geo.render_fwd(sd, alb, texh, spp1, seed=seed)
            t1 = time.perf_counter() - t0
        finally:
            _omp_set_threads(cores)
        out["one_thread"] = {"value": 1.0 / (t_upd1 + t1 * spp_full / spp1), "unit": "renders/sec", "cores": 1,
                             "sample": f"1 render at {spp1} of {spp_full} spp ({t1:.1f} s measured, scaled x{spp_full / spp1:g}) + refit {t_upd1 * 1e3:.1f} ms"}
    if grad_wl is not None:
        out["grad_steps_per_sec"] = cpu_grad_step(grad_wl, spp_full, cpu_spp, seed, cores)
    return out


def cpu_grad_step(wg, spp_full, cpu_spp, seed, cores):
    """the second half of the metric on the host: ONE pattern-gradient step chained through the oracle's entry points — K1 + K2 (sum,
    softor) + K3, re-fit, forward + adjoint of the coverage loss (the oracle's ffx_render_fwd_adjoint), K3^T, K2-bwd, K1-bwd, the
    overlap regulariser, Adam + clamp_to_fov — on the gradient bracket's scene (64-point pattern), all threads, one scene sample."""
    import numpy as np

    from fireflies_amd import scenes
    from oracle import oracle as orc

    pool, tris, shape, off, stride, nfr, alb = scenes.flatten(wg.data)
    geo = orc.Geometry(wg.mi_scene.geom.src_verts.cpu().numpy(), tris, shape, off)
    sd = wg.mi_scene.scene_desc(tex_channels=1)
    albh = wg.mi_scene.albedo.cpu().numpy()
    rays = wg.laser._rays.detach().cpu().numpy().copy()
    KF, KFi = wg.laser._KF, wg.laser._KF_inv
    s0, s1 = wg.tex_size
    H, W = sd.cam.height, sd.cam.width
    gimg = np.zeros((H, W, 3), np.float32)
    gimg[..., 1] = -1.0 / float(H * W)
    _omp_set_threads(cores)
    t0 = time.perf_counter()
    pts = np.ascontiguousarray(orc.project_rays_fwd(rays, KF)[:, :2])
    tsum = orc.splat_fwd(pts, wg.sigma, 0, -1, s0, s1)
    tsor = orc.splat_fwd(pts, wg.sigma, 1, -1, s0, s1)
    texh = orc.blur_fwd(tsum)
    t_pat_f = time.perf_counter() - t0
    t0 = time.perf_counter()
    geo.update(wg.mi_scene._xforms.numpy(), wg.mi_scene._offs)
    t_upd = time.perf_counter() - t0
    t0 = time.perf_counter()
    if sd.rfilter:  # (a filtered film: forward, then the re-traced filtered adjoint — the step the device takes, too)
        img = geo.render_fwd(sd, albh, texh, cpu_spp, seed)
        gtex = geo.render_bwd(sd, albh, cpu_spp, seed, gimg)
    else:
        img, gtex, _dot = geo.render_fwd_adjoint(sd, albh, texh, cpu_spp, seed, gimg)
    t_r = time.perf_counter() - t0
    t0 = time.perf_counter()
    gts = orc.blur_bwd(np.ascontiguousarray(gtex.reshape(texh.shape)))
    gp = orc.splat_bwd(pts, wg.sigma, 0, -1, s0, s1, tsum, gts)
    reg, gd = orc.l1_value_grad(tsor, tsum, 0.1)
    gp = gp + orc.splat_bwd(pts, wg.sigma, 1, -1, s0, s1, tsor, gd) - orc.splat_bwd(pts, wg.sigma, 0, -1, s0, s1, tsum, gd)
    grad = orc.project_rays_bwd(rays, KF, np.concatenate([gp, np.zeros((pts.shape[0], 1), np.float32)], 1))
    orc.adam_clamp_step(rays, grad, np.zeros_like(rays), np.zeros_like(rays), np.zeros(1, np.float32), 1e-3, 0.9, 0.999, 1e-8, KF, KFi, 0.05, 0.95, 2)
    t_pat_b = time.perf_counter() - t0
    per_step = t_pat_f + t_upd + t_r * (spp_full / cpu_spp) + t_pat_b
    return {"value": 1.0 / per_step, "unit": "pattern-gradient steps/sec", "cores": cores, "kind": "port",
            "sample": f"1 step, 1 scene sample: pattern forward {t_pat_f * 1e3:.0f} ms + refit {t_upd * 1e3:.1f} ms + forward-and-adjoint render at {cpu_spp} of {spp_full} spp "
                      f"{t_r:.1f} s (scaled x{spp_full / cpu_spp:g}) + pattern backward and update {t_pat_b * 1e3:.0f} ms; gcc -O3 -march=x86-64-v3 scalar oracle, OpenMP on {cores} threads"}


def launch_ranks(n, argv):
    """`python bench.py --gpus N` with no torchrun environment: N fresh child processes, one rank each.
    This parent makes NO GPU call (torch.cuda.device_count() does not initialise the runtime on this image) and
    does not import the product; a rank that fails ends the run with its exit code — nothing is restarted."""
    import socket
    import subprocess

    n_dev = torch.cuda.device_count()
    backend = os.environ.get("FFX_DIST_BACKEND")
    if n > n_dev and backend != "gloo":
        raise SystemExit(f"bench.py: --gpus {n} but only {n_dev} device(s) are visible (a single-device rehearsal of the N > 1 path needs FFX_DIST_BACKEND=gloo)")
    with socket.socket() as so:  # a free rendezvous port
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0", FFX_BENCH_LAUNCHED="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env))  # stdout/stderr inherited: rank 0 prints the line
    rc = 0
    try:
        pending = list(procs)
        while pending:
            for p in list(pending):
                code = p.poll()
                if code is None:
                    continue
                pending.remove(p)
                if code != 0 and rc == 0:
                    rc = code
                    for q in pending:  # one rank died: the others would wait in a collective forever
                        q.terminate()
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    raise SystemExit(rc)


def collective_probe(n_points, dev, iters=200):
    """what the optimisation step's ONE exchange costs on this node: all-reduce(sum) of the flat [3N+1] float buffer
    (DESIGN 6), mean over `iters` back-to-back calls, max over ranks; plus which device every rank sits on."""
    import torch.distributed as td

    world = dist.world_size()
    if world == 1:
        return None
    flat = torch.zeros(3 * n_points + 1, dtype=torch.float32, device=dev)
    for _ in range(20):
        td.all_reduce(flat)
    torch.cuda.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(iters):
        td.all_reduce(flat)
    torch.cuda.synchronize()
    us = _max_over_ranks((time.perf_counter() - t0) / iters * 1e6, dev)
    ids = [None] * world
    props = torch.cuda.get_device_properties(dev)
    td.all_gather_object(ids, {"rank": dist.rank(), "device": int(torch.cuda.current_device()), "name": props.name,
                               "pci_bus_id": getattr(props, "pci_bus_id", None), "uuid": str(getattr(props, "uuid", ""))})
    return {"world_size": world, "backend": td.get_backend(), "devices": ids, "allreduce_us": us, "allreduce_floats": int(flat.numel()),
            "launcher": "bench.py (self-launched ranks)" if os.environ.get("FFX_BENCH_LAUNCHED") else "torchrun"}


def _gather_floats(x, dev):
    """[x of rank 0, x of rank 1, ...] on every rank"""
    import torch.distributed as td

    if dist.world_size() == 1:
        return [float(x)]
    t = torch.tensor([x], dtype=torch.float64, device=dev)
    out = [torch.zeros_like(t) for _ in range(dist.world_size())]
    td.all_gather(out, t)
    return [float(o.item()) for o in out]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)  # ~0.1 s per bracket: long enough for the clocks to settle
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--res", type=int, default=512)
    ap.add_argument("--spp", type=int, default=64)
    ap.add_argument("--grid", type=int, default=16, help="laser grid (grid x grid points) for renders; 256 points by default")
    ap.add_argument("--grad-grid", type=int, default=8, help="laser grid of the gradient-step bracket (configs[1]: 64 points)")
    ap.add_argument("--cpu-spp", type=int, default=64, help="samples per pixel of the CPU-oracle baseline render (64 = the full workload, no scaling)")
    ap.add_argument("--grad-samples", type=int, default=0, help="scene samples per gradient step over all ranks (0 = one per rank; BASELINE configs[3]: 32)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-grad-steps", action="store_true")
    ap.add_argument("--no-render-steps", action="store_true", help="profiling aid: skip the render bracket (prints a reduced line)")
    ap.add_argument("--no-shadows", action="store_true")
    ap.add_argument("--workload", default="vocalfold", choices=["vocalfold", "colon"],
                    help="vocalfold = BASELINE configs[1]/[2] (the metric's configuration); colon = configs[4] "
                         "(use --res 1024 --spp 256 --grid 32 --fp16)")
    ap.add_argument("--fp16", action="store_true", help="fp16 radiance buffer (config 5)")
    ap.add_argument("--material", default="principled", choices=["principled", "diffuse"],
                    help="principled (default): the scene's material is Mitsuba's principled BSDF whose parameters the reference randomises "
                         "(examples/vocalfold_scene.py:86-93, main.py:97-107); diffuse: Lambert only")
    ap.add_argument("--rfilter", default="box", choices=["box", "gaussian"],
                    help="the film's reconstruction filter: box (the headline workload of every round) or Mitsuba's gaussian, hdrfilm's default, which a scene "
                         "FILE without <rfilter> gets (ffx_render_fwd_filtered; the gradient step then re-traces: ffx_render_bwd_filtered)")
    ap.add_argument("--entity-device", default="cuda", help="device argument of ff.Scene (where the sampler bounds live and whose generator is used): cuda (the reference default) or cpu")
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        launch_ranks(args.gpus, sys.argv[1:])  # never returns
    _load_product()
    pkey = profile_key(args)
    rank, world, local = dist.env_rank_world()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (there is no CPU path)")
    if world > torch.cuda.device_count() and os.environ.get("FFX_DIST_BACKEND") != "gloo":
        raise SystemExit(f"bench.py: {world} ranks but only {torch.cuda.device_count()} device(s) (FFX_DIST_BACKEND=gloo allows a single-device rehearsal)")
    local = local % max(torch.cuda.device_count(), 1)  # (rehearsals with more ranks than devices share a device)
    torch.cuda.set_device(local)
    dist.init(os.environ.get("FFX_DIST_BACKEND"))  # default: nccl (= RCCL); gloo only for single-GPU rehearsals of the N > 1 path
    dev = torch.device("cuda", local)
    W = H = args.res

    # ------------------------------------------------------------------ renders/sec
    make = workloads.vocalfold if args.workload == "vocalfold" else workloads.colon
    wl = make(device=dev, width=W, height=H, grid=args.grid, shadows=not args.no_shadows, entity_device=args.entity_device, principled=args.material == "principled")
    with torch.no_grad():
        tex = workloads.build_texture(wl).contiguous()
    wl.params["tex.data"] = tex
    wl.mi_scene.rfilter = args.rfilter
    geom = wl.mi_scene.geom
    base_seed = 1000

    # one seed per rank before the loop (disjoint scene samples per rank), then the reference's loop as it stands —
    # `ff_scene.randomize(); mi.render(...)` without reseeding (examples/vocalfold_scene.py:100-102)
    torch.manual_seed(base_seed + rank)
    random.seed(base_seed + rank)

    def render_step(i):
        # the reference's loop consumes every image (examples/vocalfold_scene.py:14-16,115: `mi.render(...).torch()`): the caller's stream
        # waits for the render before the next step's work is issued on it
        wl.ff_scene.randomize()
        return mi.render(wl.mi_scene, spp=args.spp, seed=base_seed + i * world + rank, fp16=args.fp16).torch()

    def render_step_unread(i):
        # the same step with the image handle dropped unread: consecutive renders then overlap on the scene's two render streams
        # (fireflies_amd/mi.py _RenderedXf) — reported as `value_overlapped`, never as `value`
        wl.ff_scene.randomize()
        return mi.render(wl.mi_scene, spp=args.spp, seed=base_seed + i * world + rank, fp16=args.fp16)

    events = []

    # Per-launch HIP event pairs (on the launch streams) are recorded for every 16th step of a timed bracket, at most
    # TIMED_STEPS of them (one in a 20-step bracket), and resolved right after it.  Every pair costs two marker packets around the kernel, which
    # keep the next step's refit from overlapping it: with all 20 steps of a short bracket instrumented the bracket
    # itself ran 8 % slower; and hundreds of unresolved timing events slow every later launch of the process down
    # (measured: the gradient bracket ran at half speed after 600 of them).
    TIMED_STEPS = int(os.environ.get("FFX_BENCH_TIMED_STEPS", "8"))

    def _timed(i, first, steps=None):  # steps 8, 24, 40, ... of a bracket (not its first steps, which start on an idle GPU)
        # (every 16th since round 5 — ONE step of the driver's 20: the marker packets of a timed step cost the bracket ~0.1 ms, more the further the host
        # runs ahead of the device, and the roofline takes its per-launch time from the drained launches of kernel_alone_ms, not from these)
        off = 8 if (steps is None or steps > 8) else 0  # (a bracket of fewer than nine steps: its first)
        return i >= first + off and (i - first - off) % 16 == 0 and (i - first - off) // 16 < TIMED_STEPS

    w_render = 0 if args.no_render_steps else args.warmup
    n_render = args.steps if not args.no_render_steps else 1

    def timed_render_step(i):
        geom.timing = events if _timed(i, w_render, n_render) else None
        return render_step(i)

    # Preflight (FFX_BENCH_PREFLIGHT=0: off): before anything is timed on this box, the kernel about to be timed must agree with the library's
    # OTHER implementation of the same render — the per-lane kernels (one ray per lane, LDS stack, binary BVH: FFX_TRAVERSAL=lane; ~20x
    # slower, the A/B baseline of DESIGN 5.1) — on the scene's current pose, three sample seeds.  The oracle is the checker of the test
    # suite and of smoke(); this is the cheap on-device cross-check that a broken build or a faulty box does not produce a benchmark line.
    preflight = {}

    def preflight_render():
        imgs = {}
        for mode in ("lane", None):
            if mode:
                os.environ["FFX_TRAVERSAL"] = mode
            try:
                imgs[mode] = [mi.render(wl.mi_scene, spp=args.spp, seed=77 + k, fp16=args.fp16).torch() for k in range(3 if args.workload == "vocalfold" else 1)]
            finally:
                os.environ.pop("FFX_TRAVERSAL", None)
        worst = max(_images_agree(a, b, args.spp, "packet kernel vs per-lane kernel") for a, b in zip(imgs[None], imgs["lane"]))
        preflight["render"] = {"checked": "k_render_fwd_pk against k_render_fwd (FFX_TRAVERSAL=lane) on the current pose", "images": len(imgs[None]),
                               "pixel_channels_beyond_1e-4_of_scale": worst}

    # (the per-lane kernels carry the box filter only: a filtered run is checked by the test suite and smoke(), not here)
    do_preflight = os.environ.get("FFX_BENCH_PREFLIGHT", "1") != "0" and "FFX_TRAVERSAL" not in os.environ and not args.no_render_steps and args.rfilter == "box"
    settle_render = (lambda k: mi.render(wl.mi_scene, spp=args.spp, seed=k, fp16=args.fp16)) if (SETTLE_RENDERS > 0 and not args.no_render_steps) else None
    # `value`: the consuming step, settled clocks (config.clock_settle)
    t_render = _bracket(timed_render_step, n_render, w_render, dev, preflight_render if do_preflight else None, settle_render)
    geom.timing = None
    torch.cuda.synchronize()
    renders_per_sec = world * args.steps / t_render
    per_rank_t = _gather_floats(_bracket.local_s, dev)
    rccl = collective_probe(args.grad_grid**2, dev)
    k8_ms, k8_n = _kernel_ms(events, "render_fwd")
    upd_ms, _ = _kernel_ms(events, "scene_update")
    events.clear()
    # a steadier kernel figure than the bracket's two instrumented launches: 16 more steps of the same loop AFTER the bracket (not part of
    # `value`), every launch between HIP events — the markers keep the next re-fit from overlapping the kernel, so this is K8 alone
    k8_post_ms = k8_post_n = k8_alone_ms = None
    value_overlapped = value_cold = None
    if not args.no_render_steps:
        geom.timing = events
        for i in range(16):
            render_step(args.warmup + args.steps + i)
        geom.timing = None
        torch.cuda.synchronize()
        k8_post_ms, k8_post_n = _kernel_ms(events, "render_fwd")
        events.clear()
        # ... and the kernel ALONE: the same 16 steps with the device drained after each — consecutive renders of the loop run beside each
        # other on the scene's two render streams, so an in-loop launch lasts longer than the kernel needs by itself (two share the GPU);
        # `avg_kernel_ms` above is what rocprofv3 reports for this command, this is what tools/binstats.py and DESIGN 5.1 quote
        geom.timing = events
        for i in range(16):
            render_step(args.warmup + args.steps + 16 + i)
            torch.cuda.synchronize()
        geom.timing = None
        k8_alone_ms, _ = _kernel_ms(events, "render_fwd")
        events.clear()
        if os.environ.get("FFX_BENCH_EXTRA_BRACKETS", "1") != "0":
            # (i) the same bracket with the image handles dropped unread (two render streams overlap consecutive renders)
            paths0 = dict(wl.mi_scene.render_paths)
            t_unread = _bracket(render_step_unread, args.steps, args.warmup, dev, None, settle_render)
            value_overlapped = world * args.steps / t_unread
            paths_unread = {k: v - paths0[k] for k, v in wl.mi_scene.render_paths.items()}
            # (ii) the consuming bracket again from a COLD start: the GPU idles first (the clocks fall back within ~0.3 s, tools/posecost.py),
            # no settle launches — what `python bench.py --steps K --warmup W` measured before round 3
            torch.cuda.synchronize()
            time.sleep(0.4)
            t_cold = _bracket(render_step, args.steps, args.warmup, dev, None, None)
            value_cold = world * args.steps / t_cold
    bytes_ = algorithmic_bytes(wl, W, H, fp16=args.fp16)

    # ------------------------------------------------------------------ pattern-gradient steps/sec
    grad = {}
    if not args.no_grad_steps:
        from fireflies_amd import functional as Fn
        from fireflies_amd import ops
        from fireflies_amd.optim import image_l1_loss

        wg = make(device=dev, width=W, height=H, grid=args.grad_grid, shadows=not args.no_shadows, entity_device=args.entity_device, principled=args.material == "principled")
        wg.mi_scene.rfilter = args.rfilter
        S = args.grad_samples if args.grad_samples > 0 else world
        opt = PatternOptimizer(wg.mi_scene, wg.ff_scene, wg.laser, sigma=wg.sigma, tex_size=wg.tex_size, spp=args.spp, samples_per_step=S, base_seed=7)
        gevents = []
        T = wg.tex_size[0] * wg.tex_size[1]

        def grad_step(i):
            wg.mi_scene.geom.timing = gevents if _timed(i, args.warmup, args.steps) else None
            return opt.step()

        def _adjoint_close(a, b, what):
            scale = float(b.abs().max())
            err = (a - b).abs()
            bad = float((err > 1e-3 * scale).float().mean())
            if not (scale > 0 and bad <= 1e-3 and float(err.max()) <= 0.1 * scale):
                raise SystemExit(f"bench.py preflight: {what}: {bad:.2e} of the texels differ — not timing an adjoint whose output is in doubt")
            return bad

        def _pose_inputs():
            ms, geom_g = wg.mi_scene, wg.mi_scene.geom
            sd = ms.scene_desc(tex_channels=1)
            with torch.no_grad():
                tex3 = workloads.build_texture(wg).contiguous()
            tex3 = tex3[..., 1:2].contiguous() if tex3.dim() == 3 else tex3.unsqueeze(-1).contiguous()
            return ms, geom_g, sd, tex3

        def preflight_grad_fused():
            # the kernel the default gradient bracket times — forward + adjoint in ONE launch (ffx_render_fwd_adjoint, the instance
            # k_render_fwd_pk<..., true>) — against the library's two other routes to the same pair: the plain forward for the image and
            # the re-tracing adjoint (ffx_render_bwd) for the texture gradient, on the scene's current pose, four sample seeds
            ms, geom_g, sd, tex3 = _pose_inputs()
            gimg = torch.zeros((H, W, 3), device=dev)
            gimg[..., 1] = -1.0 / (H * W)
            worst = worst_img = 0.0
            for k in range(4):
                mats = ms.materials_arg(sd)
                img_f, gt_f = geom_g.render_fwd_adjoint(sd, mats, tex3, args.spp, 55 + k, gimg)
                img_p = geom_g.render_fwd(sd, mats, tex3, args.spp, 55 + k, False)
                gt_r = geom_g.render_bwd(sd, mats, args.spp, 55 + k, gimg)
                worst_img = max(worst_img, _images_agree(img_f, img_p, args.spp, "forward + adjoint launch vs plain forward (image)"))
                worst = max(worst, _adjoint_close(gt_f, gt_r, "forward + adjoint launch vs re-traced adjoint"))
            preflight["gradient"] = {"checked": "k_render_fwd_pk<..., true> (ffx_render_fwd_adjoint: the launch the gradient bracket times) against k_render_fwd_pk<..., false> "
                                                "(image) and k_render_bwd_pk (re-tracing adjoint) on the current pose", "adjoints": 4,
                                     "texels_beyond_1e-3_of_scale": worst, "pixel_channels_beyond_1e-4_of_scale": worst_img}

        def preflight_grad_cached():
            # the adjoint the NON-LINEAR bracket times (K9 from the footprint cache K8 writes) against the re-tracing adjoint, four seeds
            ms, geom_g, sd, tex3 = _pose_inputs()
            if not Fn.cache_supported(sd, args.spp):
                return
            cache = torch.empty(ops.render_cache_bytes_sd(sd, args.spp), dtype=torch.uint8, device=dev)
            gimg = torch.randn((H, W, 3), device=dev).sign_() / (3.0 * H * W)  # (the shape of an L1 loss's gradient)
            worst = 0.0
            for k in range(4):
                mats = ms.materials_arg(sd)
                geom_g.render_fwd(sd, mats, tex3, args.spp, 55 + k, False, cache=cache)
                a = geom_g.render_bwd_cached(sd, mats, cache, args.spp, gimg)
                b = geom_g.render_bwd(sd, mats, args.spp, 55 + k, gimg)
                worst = max(worst, _adjoint_close(a, b, "cached adjoint vs re-traced adjoint"))
            preflight["gradient_nonlinear"] = {"checked": "k_render_bwd_cached_tiled16 (from the footprint cache k_render_fwd_pk writes) against k_render_bwd_pk (re-tracing) "
                                                          "on the current pose", "adjoints": 4, "texels_beyond_1e-3_of_scale": worst}

        pre_ok = os.environ.get("FFX_BENCH_PREFLIGHT", "1") != "0" and "FFX_TRAVERSAL" not in os.environ and args.rfilter == "box"
        settle_grad = (lambda k: mi.render(wg.mi_scene, spp=args.spp, seed=k)) if SETTLE_RENDERS > 0 else None
        fused_expected = os.environ.get("FFX_FUSED_ADJOINT", "1") != "0" and len(opt._sample_seeds(0)) <= 64
        t_grad = _bracket(grad_step, args.steps, args.warmup, dev, (preflight_grad_fused if fused_expected else preflight_grad_cached) if pre_ok else None, settle_grad)
        wg.mi_scene.geom.timing = None
        torch.cuda.synchronize()
        k9_ms, _ = _kernel_ms(gevents, "render_bwd")  # re-tracing adjoint (only above FFX_CACHE_LIMIT_GB)
        k9c_ms, _ = _kernel_ms(gevents, "render_bwd_cached")
        k8g_ms, k8g_n = _kernel_ms(gevents, "render_fwd")
        gevents.clear()
        lin_paths = dict(opt.step_paths)
        fused_ran = lin_paths["fused"] > 0
        bg = algorithmic_bytes(wg, W, H)
        bytes_fused = bg["G"] + 4 * T + 12 * W * H + 12 * W * H + 4 * T  # G + tex + img out + gimg in + gtex
        grad = {
            "grad_steps_per_sec": args.steps / t_grad,
            "grad_samples_per_sec": S * args.steps / t_grad,
            "grad_ms_per_step": 1e3 * t_grad / args.steps,
            "grad_config": {"points": args.grad_grid**2, "samples_per_step": S, "samples_per_rank": len(range(rank, S, world)),
                            "loss": "coverage_loss = -mean(green): linear in the image (forward + adjoint in one launch)" if fused_ran
                            else ("coverage_loss, filtered film: forward (ffx_render_fwd_filtered) + re-traced adjoint (ffx_render_bwd_filtered)" if lin_paths["retrace"] > 0
                                  else "coverage_loss through the cache + K9")},
            "grad_kernels_ms": {("render_fwd(+adjoint in the same launch: ffx_render_fwd_adjoint)" if fused_ran else "render_fwd(+cache write)"): k8g_ms,
                                "render_bwd_cached": k9c_ms, "render_bwd(retrace)": k9_ms},
            "grad_launches_per_step": ("render_fwd_adjoint, pattern_step<5> [gradient + Adam + clamp, then the next step's splat + blur] (+ re-fit and apex records on the side stream)" if args.rfilter == "box" else
                                       "[rf_weights, rf_gather (G), render_fwd_adjoint_filtered, rf_gather (image)], pattern_step<5> (+ re-fit and pre-pass on the side stream)") if fused_ran
            else "render_fwd_cache, render_bwd_cached, pattern_step<5> (+ re-fit and apex records on the side stream)",
            "grad_step_paths": lin_paths,
            "grad_update_paths": dict(wg.mi_scene.update_paths),  # (scene samples pushed by ffx_scene_step_h / by the Python path, up to the end of this bracket)
            "render_fwd_adjoint_roofline": None if not (fused_ran and k8g_ms) else {
                "kernel": "k_render_fwd_pk<1, wide, material rows, true> (ffx_render_fwd_adjoint: K8 with the adjoint folded in)", "bound": "hbm",
                "achieved": bytes_fused / (k8g_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": bytes_fused / (k8g_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "algorithmic_bytes_per_launch": bytes_fused, "bytes_formula": "G + 4T (tex) + 12WH (img) + 12WH (gimg) + 4T (gtex)",
                "avg_kernel_ms": k8g_ms, "launches_timed": k8g_n,
                "traffic": (pmc_traffic("k_render_fwd_pk", "grad", pkey, adjoint_instance=True) or {}).get("bytes") if pkey is not None else None,
                "traffic_source": (pmc_traffic("k_render_fwd_pk", "grad", pkey, adjoint_instance=True) or {}).get("source") if pkey is not None else None},
        }
        # ---- the same step with a loss that is NOT linear in the image (torch.nn.L1Loss against a fixed target render, the loss class of
        # the reference's loop: fireflies/graphics/rasterization.py:579,596-602): cache-writing forward + K9 (ffx_render_bwd_cached)
        if os.environ.get("FFX_BENCH_NONLINEAR", "1") != "0":
            with torch.no_grad():
                target = mi.render(wg.mi_scene, spp=args.spp, seed=4242).torch().clone()
            opt2 = PatternOptimizer(wg.mi_scene, wg.ff_scene, wg.laser, sigma=wg.sigma, tex_size=wg.tex_size, spp=args.spp, samples_per_step=S, base_seed=11,
                                    loss_fn=image_l1_loss(target))

            def grad_step2(i):
                wg.mi_scene.geom.timing = gevents if _timed(i, args.warmup, args.steps) else None
                return opt2.step()

            t_grad2 = _bracket(grad_step2, args.steps, args.warmup, dev, preflight_grad_cached if pre_ok else None, settle_grad)
            wg.mi_scene.geom.timing = None
            torch.cuda.synchronize()
            k9r2_ms, _ = _kernel_ms(gevents, "render_bwd")
            k9c2_ms, k9c2_n = _kernel_ms(gevents, "render_bwd_cached")
            k8g2_ms, _ = _kernel_ms(gevents, "render_fwd")
            gevents.clear()
            # streaming adjoint over the per-pixel footprint cache written by K8: the 8-byte header of every pixel, the 25-float
            # footprint(s) of the LIT pixels, the stray records, d(loss)/d(img); gtex written once.  Counted in the cache the last step left.
            n_lit = n_stray = 0
            if opt2._cache is not None:
                n_stray = int(opt2._cache[:4].view(torch.int32).item())
                n_lit = int((opt2._cache[64:64 + 8 * W * H].view(W * H, 8)[:, 6] != 0).sum().item())  # CachePix.lit (8-byte headers, dense)
            foot = 200 if args.material == "principled" else 100  # (material rows: a second footprint per lit pixel)
            bytes_k9c = 8 * W * H + foot * n_lit + 24 * n_stray + 12 * W * H + 4 * T
            grad.update({
                "grad_steps_per_sec_nonlinear": args.steps / t_grad2,
                "grad_ms_per_step_nonlinear": 1e3 * t_grad2 / args.steps,
                "grad_nonlinear_config": {"loss": "torch.nn.L1Loss()(img, target) against a fixed target render (optim.image_l1_loss): its gradient depends on the image",
                                          "step_paths": dict(opt2.step_paths),
                                          "launches_per_step": "render_fwd_cache, render_bwd_cached_l1 [the L1 loss's value and gradient formed inside K9], pattern_step<5> (+ re-fit and apex records on the side stream)"},
                "grad_nonlinear_kernels_ms": {"render_fwd(+cache write)": k8g2_ms, "render_bwd_cached": k9c2_ms, "render_bwd(retrace)": k9r2_ms},
                "render_bwd_cached_roofline": None if k9c2_ms is None else {
                    "kernel": "k_render_bwd_cached_tiled16 (scatters the per-pixel texture footprints written by K8)", "bound": "hbm",
                    "lit_pixels": n_lit, "stray_samples": n_stray,
                    "achieved": bytes_k9c / (k9c2_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": bytes_k9c / (k9c2_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": bytes_k9c,
                    "bytes_formula": f"8WH (pixel headers) + {foot} n_lit + 24 n_stray + 12WH (gimg) + 4T (gtex)",
                    "avg_kernel_ms": k9c2_ms, "launches_timed": k9c2_n,
                    "traffic": (pmc_traffic("k_render_bwd_cached", "grad", pkey) or {}).get("bytes") if pkey is not None else None,
                    "traffic_source": (pmc_traffic("k_render_bwd_cached", "grad", pkey) or {}).get("source") if pkey is not None else None},
            })
        grad["grad_config"]["algorithmic_bytes"] = {"render_fwd_adjoint": bytes_fused, "G": bg["G"]}

    # ------------------------------------------------------------------ the same three brackets through Mitsuba's DEFAULT film
    # Every scene the reference loads gets hdrfilm's gaussian reconstruction filter (examples/vocalfold_scene.py:20-22, main.py:26-29: no
    # <rfilter> in the files), and its own optimisation loop uses an L1 loss (fireflies/graphics/rasterization.py:579,596-602): the headline
    # `value` keeps the box film of rounds 1-4 (comparable across rounds), these are the reference-faithful figures beside it —
    # value_gaussian (ffx_render_fwd_filtered), grad_steps_per_sec_gaussian (coverage loss: ffx_render_fwd_adjoint_filtered) and
    # grad_steps_per_sec_gaussian_nonlinear (L1 loss: ffx_render_fwd_cache_filtered + ffx_render_bwd_cached_filtered, round 5; re-traced before).
    gauss = {}
    if args.rfilter == "box" and os.environ.get("FFX_BENCH_GAUSSIAN", "1") != "0":
        if not args.no_render_steps:
            wl.mi_scene.rfilter = "gaussian"
            try:
                t_g = _bracket(render_step, args.steps, args.warmup, dev, None, settle_render)
            finally:
                wl.mi_scene.rfilter = "box"
            gauss.update({"value_gaussian": world * args.steps / t_g, "ms_per_step_gaussian": 1e3 * t_g / args.steps})
        if not args.no_grad_steps:
            wg.mi_scene.rfilter = "gaussian"
            try:
                optg = PatternOptimizer(wg.mi_scene, wg.ff_scene, wg.laser, sigma=wg.sigma, tex_size=wg.tex_size, spp=args.spp, samples_per_step=S, base_seed=13)
                t_gg = _bracket(lambda i: optg.step(), args.steps, args.warmup, dev, None, settle_grad)
                gauss.update({"grad_steps_per_sec_gaussian": args.steps / t_gg, "grad_ms_per_step_gaussian": 1e3 * t_gg / args.steps, "grad_gaussian_step_paths": dict(optg.step_paths)})
                if os.environ.get("FFX_BENCH_NONLINEAR", "1") != "0":
                    with torch.no_grad():
                        target_g = mi.render(wg.mi_scene, spp=args.spp, seed=4242).torch().clone()
                    optg2 = PatternOptimizer(wg.mi_scene, wg.ff_scene, wg.laser, sigma=wg.sigma, tex_size=wg.tex_size, spp=args.spp, samples_per_step=S, base_seed=17,
                                             loss_fn=image_l1_loss(target_g))

                    def preflight_gauss_cached():
                        # the pair this bracket times — filtered forward that stores per-sample records, adjoint from the records — against the re-tracing
                        # filtered adjoint (which the full-size parity test holds against the oracle), two seeds on the current pose
                        ms, geom_g, sd, tex3 = _pose_inputs()
                        if not Fn.cache_supported(sd, args.spp):
                            return
                        cache = torch.empty(ops.render_cache_bytes_sd(sd, args.spp), dtype=torch.uint8, device=dev)
                        gimg = torch.randn((H, W, 3), device=dev).sign_() / (3.0 * H * W)
                        worst = 0.0
                        for k in range(2):
                            mats = ms.materials_arg(sd)
                            geom_g.render_fwd(sd, mats, tex3, args.spp, 55 + k, False, cache=cache)
                            used, cap, dropped = ops.render_cache_status(cache)
                            if dropped:  # (the arena of 64-sample blocks is a quarter of the dense size beyond 2^18 blocks: a pose that lights more of the film
                                # overflows it — counted, the adjoint poisoned, PatternOptimizer re-traces from then on: nothing to compare here)
                                preflight["gradient_gaussian_nonlinear"] = {"checked": f"the adjoint cache's arena overflowed on this pose ({dropped} blocks beyond {cap}): "
                                                                                       "the optimiser takes the re-tracing adjoint (ffx_render_bwd_filtered)", "adjoints": 0}
                                return
                            a = geom_g.render_bwd_cached(sd, mats, cache, args.spp, gimg, seed=55 + k)
                            b = geom_g.render_bwd(sd, mats, args.spp, 55 + k, gimg)
                            worst = max(worst, _adjoint_close(a, b, "filtered film: cached adjoint vs re-traced adjoint"))
                        preflight["gradient_gaussian_nonlinear"] = {"checked": "k_render_bwd_cached_filtered (from the per-sample records k_render_fwd_pk<..., RFC> writes) against "
                                                                               "k_render_bwd_pk<..., RF> (re-tracing) on the current pose", "adjoints": 2, "texels_beyond_1e-3_of_scale": worst}

                    def grad_step_g2(i):
                        wg.mi_scene.geom.timing = gevents if _timed(i, args.warmup, args.steps) else None
                        return optg2.step()

                    t_gg2 = _bracket(grad_step_g2, args.steps, args.warmup, dev, preflight_gauss_cached if (os.environ.get("FFX_BENCH_PREFLIGHT", "1") != "0" and "FFX_TRAVERSAL" not in os.environ) else None, settle_grad)
                    wg.mi_scene.geom.timing = None
                    torch.cuda.synchronize()
                    k9f_ms, _ = _kernel_ms(gevents, "render_bwd_cached")
                    k8f_ms, _ = _kernel_ms(gevents, "render_fwd")
                    k9fr_ms, _ = _kernel_ms(gevents, "render_bwd")
                    gevents.clear()
                    gauss.update({"grad_steps_per_sec_gaussian_nonlinear": args.steps / t_gg2, "grad_ms_per_step_gaussian_nonlinear": 1e3 * t_gg2 / args.steps,
                                  "grad_gaussian_nonlinear_config": {
                                      "loss": "torch.nn.L1Loss()(img, target) on the gaussian film (optim.image_l1_loss)", "step_paths": dict(optg2.step_paths),
                                      "launches_per_step": "render_fwd_cache_filtered [K8 + per-sample records, rf_gather], l1_value_grad, render_bwd_cached_filtered, "
                                                           "pattern_step<5> (+ re-fit and pre-pass on the side stream)",
                                      "kernels_ms": {"render_fwd_cache_filtered (K8 + gather)": k8f_ms, "render_bwd_cached_filtered": k9f_ms, "render_bwd_filtered(retrace)": k9fr_ms}}})
            finally:
                wg.mi_scene.rfilter = "box"

    # ------------------------------------------------------------------ the sample counts of the reference's own scripts
    # examples/01..06 render at 10 and 12 spp, main.py:144 draws its spp from 1..100, all through hdrfilm's gaussian film: below 33 spp the renders pack
    # several pixels into a wave (k_render_fwd_blk, round 5).  Two short brackets of the same randomise + render + read-back step, informational beside
    # the headline (FFX_BENCH_LOWSPP=0: off): 16 spp with the box film, 10 spp with the gaussian film.
    lowspp = {}
    if args.rfilter == "box" and args.spp > 32 and not args.no_render_steps and os.environ.get("FFX_BENCH_LOWSPP", "1") != "0" and os.environ.get("FFX_BENCH_EXTRA_BRACKETS", "1") != "0":
        def step_at(spp_):
            def f(i):
                wl.ff_scene.randomize()
                return mi.render(wl.mi_scene, spp=spp_, seed=base_seed + i * world + rank, fp16=args.fp16).torch()
            return f

        t16 = _bracket(step_at(16), args.steps, args.warmup, dev)
        wl.mi_scene.rfilter = "gaussian"
        try:
            t10 = _bracket(step_at(10), args.steps, args.warmup, dev)
        finally:
            wl.mi_scene.rfilter = "box"
        lowspp = {"value_16spp": world * args.steps / t16, "ms_per_step_16spp": 1e3 * t16 / args.steps,
                  "value_10spp_gaussian": world * args.steps / t10, "ms_per_step_10spp_gaussian": 1e3 * t10 / args.steps,
                  "lowspp_config": "the headline's step at 16 spp (box film) and at 10 spp with the gaussian film (the reference's examples/01_hello_world.py:29); "
                                   "below 33 spp a wave renders a compact block of pixels (k_render_fwd_blk), the same image bit for bit as a pixel per wave"}

    # ------------------------------------------------------------------ BASELINE configs[3] and [4], driver-timed beside the headline (round-5 review, item 5)
    # Two short informational brackets (FFX_BENCH_CONFIGS34=0: off; only behind the default workload, one rank): configs[3] — 32 scene samples per
    # gradient step (the step of examples/11_..., `--grad-samples 32`) — and configs[4] — the colon at 1024 x 1024 x 256 spp with an fp16 film and a
    # 1024-point pattern (`--workload colon --res 1024 --spp 256 --grid 32 --fp16`), with its own roofline from the committed r*colon_* passes.
    cfg34 = {}
    default_line = (args.workload == "vocalfold" and args.rfilter == "box" and args.res == 512 and args.spp == 64 and args.material == "principled" and not args.fp16
                    and not args.no_shadows and args.grad_samples == 0)
    if default_line and world == 1 and os.environ.get("FFX_BENCH_CONFIGS34", "1") != "0" and os.environ.get("FFX_BENCH_EXTRA_BRACKETS", "1") != "0":
        if not args.no_grad_steps:
            opt32 = PatternOptimizer(wg.mi_scene, wg.ff_scene, wg.laser, sigma=wg.sigma, tex_size=wg.tex_size, spp=args.spp, samples_per_step=32, base_seed=19)
            t32 = _bracket(lambda i: opt32.step(), 3, 1, dev, None, settle_grad)
            cfg34.update({"grad_samples_per_sec_s32": 32 * 3 / t32, "grad_ms_per_step_s32": 1e3 * t32 / 3,
                          "grad_s32_config": {"what": "BASELINE configs[3]: 32 randomised scene samples per gradient step (one rank: all 32 here; N ranks take 32 / N each and "
                                                      "exchange [3N+2] floats), 3 timed steps after 1", "step_paths": dict(opt32.step_paths)}})
        if not args.no_render_steps:
            t_set = time.perf_counter()
            wc = workloads.colon(device=dev, width=1024, height=1024, tex=1024, grid=32, entity_device=args.entity_device)
            with torch.no_grad():
                wc.params["tex.data"] = workloads.build_texture(wc).contiguous()
            t_set = time.perf_counter() - t_set

            def colon_step(i):
                wc.ff_scene.randomize()
                return mi.render(wc.mi_scene, spp=256, seed=base_seed + i, fp16=True).torch()

            tc = _bracket(colon_step, 5, 2, dev)
            cev = []
            wc.mi_scene.geom.timing = cev
            for i in range(3):  # the kernel alone: device drained between the launches (as kernel_alone_ms above)
                colon_step(7 + i)
                torch.cuda.synchronize()
            wc.mi_scene.geom.timing = None
            kc_ms, kc_n = _kernel_ms(cev, "render_fwd")
            bc = algorithmic_bytes(wc, 1024, 1024, fp16=True)
            trc = pmc_traffic("k_render_fwd_pk", "r", "colon")
            cfg34.update({"value_colon": 5 / tc, "ms_per_step_colon": 1e3 * tc / 5, "colon_setup_s": t_set,
                          "colon_config": f"BASELINE configs[4]: procedural colon scene, {bc['F']} triangles, 1024-point projector, 1024x1024, 256 spp, fp16 radiance buffer; "
                                          "5 timed steps after 2 (randomise + params.update() + mi.render(...).torch())",
                          "colon_primary_rays_per_sec": 5 * 1024 * 1024 * 256 / tc,
                          "colon_roofline": None if not kc_ms else {
                              "kernel": "k_render_fwd_pk<1, wide, material rows, false> (ffx_render_fwd, K8) on configs[4]", "bound": "hbm",
                              "achieved": bc["render_fwd"] / (kc_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": bc["render_fwd"] / (kc_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                              "traffic": None if trc is None else trc["bytes"], "traffic_ratio": None if trc is None else trc["bytes"] / bc["render_fwd"],
                              "traffic_source": None if trc is None else f"profiles/{trc['source']}",
                              "algorithmic_bytes_per_launch": bc["render_fwd"], "avg_kernel_ms": kc_ms, "kernel_alone_ms": kc_ms, "launches_timed": kc_n,
                              "valu_issue": valu_issue("k_render_fwd_pk", kc_ms, "colon")}})
            del wc
            torch.cuda.empty_cache()

    if rank != 0:
        return
    # the dominant kernel's time: the SIXTEEN launches timed behind the bracket (each between HIP events on its launch stream); the
    # bracket's own two instrumented launches are kept beside it
    # (round-4 review: the in-loop duration of a launch is concurrency-stretched — consecutive renders share the GPU on two render streams, 0.53 ms
    # "per launch" inside a 0.40 ms step — so the roofline is held against the kernel ALONE, device drained between launches; the in-loop
    # figure stays in the line as avg_kernel_ms_overlapped)
    k8_loop_ms = k8_post_ms if k8_post_ms else k8_ms
    k8_roof_ms = k8_alone_ms if k8_alone_ms else k8_loop_ms
    achieved = bytes_["render_fwd"] / (k8_roof_ms * 1e-3) / 1e9
    st_traffic = step_traffic(pkey) if pkey is not None else None
    step_alg = bytes_["render_fwd"] + bytes_["scene_update"] + 12 * bytes_["V"]  # SURVEY 8d "one render" without the once-per-loop K2: K5+K6 (in + out + nodes) + K8
    traffic = pmc_traffic("k_render_fwd_pk", "r", pkey) if pkey is not None else None  # only workloads the committed PMC passes ran
    phase_files = _profile_files("phaseclk.txt", pkey or "") if pkey is not None else []
    phase_files = [f for f in phase_files if os.path.getsize(f) > 0]
    out = {
        "metric": "renders/sec @512x512,64spp vocal-fold (+ pattern-grad-steps/sec in grad_steps_per_sec); HBM GB/s vs peak in roofline",
        "value": renders_per_sec,
        "unit": "renders/sec",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * t_render / args.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32" if not args.fp16 else "f32 (f16 film)",
        "data": "synthetic",
        # the same K steps without the clock-settle launches, after 0.4 s of idle GPU (what a bare `--steps K --warmup W` loop reads), and with
        # the image handles dropped unread (consecutive renders then overlap on two streams; the reference's loop reads every image)
        "value_cold": value_cold,
        "value_overlapped": value_overlapped,
        "config": {
            "workload": (f"BASELINE configs[2] (renders): procedural animated vocal-fold scene, {bytes_['F']} triangles, {args.grid**2}-point projector, "
                         f"{W}x{H}, {args.spp} spp, shadows {'on' if not args.no_shadows else 'off'}; configs[1] (grad steps): same scene, {args.grad_grid**2}-point pattern")
            if args.workload == "vocalfold" else
            (f"BASELINE configs[4]: procedural colon scene, {bytes_['F']} triangles, {args.grid**2}-point projector, {W}x{H}, {args.spp} spp, "
             f"{'fp16' if args.fp16 else 'fp32'} radiance buffer"),
            "step": "ff_scene.randomize() [draws + chains: ffx_scene_randomize_h; push: ffx_scene_step_h = params.update(), K5+K6 and the pre-pass] + mi.render(...).torch() [K8; the image is consumed on the caller's stream, as the reference's loop "
                    "does]; texture built once before the loop",
            "preflight": preflight or None,
            "clock_settle": {"renders_before_each_bracket": SETTLE_RENDERS,
                             "what": "launches of the render kernel on the current pose issued in front of the W warm-up steps (not steps: no randomisation, no re-fit, no adjoint)",
                             "why": "the first ~40 launches after an idle GPU run up to 6 % slower (clock ramp; tools/posecost.py, profiles/r3_posecost.txt): a 20-step bracket "
                                    "would time the ramp.  `value_cold` is the same bracket without them, after an idle phase"},
            "entity_device": args.entity_device,
            "rfilter": args.rfilter,
            "material": ("principled BSDF (Mitsuba's model, reflection side), parameters randomised as the reference's scripts do" if args.material == "principled"
                         else "diffuse (Lambert)"),
            "primary_rays_per_sec": world * args.steps * W * H * args.spp / t_render,
            "parallelism": f"dp{world}: independent scene samples per rank, no collective in the render loop",
            "render_paths": dict(wl.mi_scene.render_paths),
            # how the scene samples reached the device: "native" = one ffx_scene_step_h call each (ABI 8, the native params.update()), "python" = key writes + params.update()
            "update_paths": dict(wl.mi_scene.update_paths),
            "update_fallbacks": dict(wl.mi_scene.update_fallbacks),
        },
        "roofline": {
            "kernel": "k_render_fwd_pk<1, wide, %s, false> (ffx_render_fwd, K8)" % ("material rows" if args.material == "principled" else "albedo"),
            "bound": "hbm",
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "traffic": None if traffic is None else traffic["bytes"],
            "traffic_ratio": None if traffic is None else traffic["bytes"] / bytes_["render_fwd"],  # counter-corrected bytes / algorithmic bytes of the kernel
            "step_traffic": None if st_traffic is None else {**st_traffic, "algorithmic_bytes_per_step": step_alg, "traffic_ratio": st_traffic["bytes_per_step"] / step_alg},
            "traffic_source": ("no committed PMC pass for this workload" if pkey is None else f"no profiles/r*{pkey}_pmc_summary.json yet") if traffic is None else
            f"profiles/{traffic['source']} (committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this bench at this workload, not this run; raw {traffic['raw_bytes']:.0f} B, FETCH_SIZE x2 gfx950 correction)",
            "algorithmic_bytes_per_launch": bytes_["render_fwd"],
            "avg_kernel_ms": k8_roof_ms,  # = kernel_alone_ms: 16 launches of the loop's steps with the device drained between them
            "launches_timed": 16 if k8_alone_ms else (k8_post_n if k8_post_ms else k8_n),
            "kernel_alone_ms": k8_alone_ms,
            "avg_kernel_ms_overlapped": k8_loop_ms,  # 16 launches INSIDE the loop: each shares the GPU with its neighbour on the other render stream (what rocprofv3 averages)
            "avg_kernel_ms_in_bracket": k8_ms,
            "launches_timed_in_bracket": k8_n,
            # (against the kernel ALONE: the in-loop duration of a launch counts the time it shares the GPU with the previous render)
            "valu_issue": valu_issue("k_render_fwd_pk", k8_roof_ms, pkey) if pkey is not None else None,
            "note": "by design NOT HBM-bound: samples are reduced in registers, so the compulsory traffic per render is algorithmic_bytes_per_launch (geometry + "
                    "texture + film, SURVEY 8d); the kernel is bound by VALU / scalar issue (valu_issue below; SQ counters in profiles/r*_sq_instruction_mix.json"
                    + (f", phase shares in profiles/{os.path.basename(phase_files[-1])}" if phase_files else "") + ", DESIGN 8). rays/s is the meaningful secondary figure.",
            "kernel_ray_samples_per_sec": W * H * args.spp / (k8_roof_ms * 1e-3),
        },
        "kernels_ms": {"scene_update(K5+K6: side stream, overlapped with K8; elapsed incl. waiting for CUs)": upd_ms, "render_fwd(K8, alone)": k8_roof_ms,
                       "render_fwd(K8, in the loop: overlapped)": k8_loop_ms},
        "rccl": rccl,
    }
    if rccl is not None:
        rccl["renders_per_sec_per_rank"] = [args.steps / t for t in per_rank_t]
    out.update(grad)
    out.update(gauss)
    out.update(lowspp)
    out.update(cfg34)
    if not args.no_cpu_baseline and world == 1:
        out["cpu_baseline"] = cpu_baseline(wl, tex, args.spp, args.cpu_spp, seed=base_seed, grad_wl=wg if not args.no_grad_steps else None)
    else:
        out["cpu_baseline"] = None
    out["evidence"] = cited_profiles(pkey) + ([f for f in cited_profiles("colon") if "colon" in f] if "colon_roofline" in cfg34 else [])
    print(json.dumps(out))


if __name__ == "__main__":
    main()
