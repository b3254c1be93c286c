"""bench.py's bookkeeping without a GPU: the compute-side ceiling in the bench line is re-derivable from the committed
profiles (profiles/r2_sq_instruction_mix.json x the issue rates of profiles/r2_issue_rates.txt), stays <= 1, and the
PMC traffic figure is only attached to the workload the committed passes ran."""
import argparse
import json
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def bench():
    import importlib.util

    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_valu_ceiling_is_rederivable_and_below_one(bench):
    mix_file = bench._profile_files("sq_instruction_mix.json")[-1]  # the newest committed pass of the default workload
    tag = os.path.basename(mix_file).split("_")[0]
    assert tag in ("r2", "r3", "r4", "r5", "r6")
    mix = json.load(open(mix_file))
    (k8,) = [v for k, v in mix.items() if "k_render_fwd_pk" in k]
    stats = open(os.path.join(ROOT, "profiles", f"{tag}_kernel_stats.csv")).read()
    avg_ns = float(re.search(r'k_render_fwd_pk[^\n]*?",\d+,\d+,([0-9.]+),', stats).group(1))
    vi = bench.valu_issue("k_render_fwd_pk", avg_ns * 1e-6)
    n = k8["SQ_INSTS_VALU"]["mean"]
    fast = sum(k8[c]["mean"] for c in ("SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_ADD_F32"))
    trans = k8["SQ_INSTS_VALU_TRANS_F32"]["mean"]
    cyc = fast * 2.0 + trans * 8.0 + (n - fast - trans) * 4.0
    assert vi["ceiling_ms"] == pytest.approx(1e3 * cyc / 1024 / 2.4e9, rel=1e-9)
    assert 0.5 < vi["frac"] <= 1.0, vi  # against the kernel time rocprofv3 recorded in the same profile set
    assert f"{tag}_sq_instruction_mix.json" in vi["source"]
    # the scalar side at the measured 4.2 cycles per instruction does not exceed the kernel time either
    scalar_ms = 1e3 * (k8["SQ_INSTS_SALU"]["mean"] + k8["SQ_INSTS_BRANCH"]["mean"]) * 4.2 / 1024 / 2.4e9
    assert scalar_ms <= avg_ns * 1e-6 * 1.02


def test_operand_form_ceiling_is_consistent(bench):
    """round 3: the share of the add / mul / fma class that carries a scalar source (tools/isa_mix.py, committed) is priced at the
    4-cycle rate — a ceiling between the nominal one and the kernel time recorded in the same profile set"""
    forms = json.load(open(bench._profile_files("isa_operand_forms.json")[-1]))
    assert "k_render_fwd_pk<1, true, 1, false, false" in forms["kernel"]  # (<R, WIDE, MATM, ADJ, RF[, RFC]>: the plain forward with material rows)
    assert forms["with_scalar_or_constant_source"] == sum(v["with_scalar_source"] for v in forms["per_opcode"].values())
    fr = forms["scalar_source_fraction"]
    assert 0.2 < fr < 0.7
    tag = os.path.basename(bench._profile_files("sq_instruction_mix.json")[-1]).split("_")[0]
    stats = open(os.path.join(ROOT, "profiles", f"{tag}_kernel_stats.csv")).read()
    avg_ms = float(re.search(r'k_render_fwd_pk[^\n]*?",\d+,\d+,([0-9.]+),', stats).group(1)) * 1e-6
    vi = bench.valu_issue("k_render_fwd_pk", avg_ms)
    assert vi["scalar_source_fraction_static"] == fr
    assert vi["ceiling_ms_operand_forms"] == pytest.approx(vi["ceiling_ms"] + 1e3 * vi["fp32_fma_mul_add"] * fr * 2.0 / 1024 / 2.4e9, rel=1e-9)
    assert vi["frac"] < vi["frac_operand_forms"] <= 1.02, vi  # (against the kernel time of the same committed profile set)


def test_issue_rates_file_backs_the_class_rates(bench):
    txt = open(os.path.join(ROOT, "profiles", "r2_issue_rates.txt")).read()

    def best(label):  # lowest cycles/instr/SIMD over the occupancies >= 4 waves
        vals = [float(m.group(2)) for m in re.finditer(re.escape(label) + r"\s+W=(\d)\s+[0-9.]+ ms\s+clk [0-9.]+ GHz\s+([0-9.]+) cyc", txt) if int(m.group(1)) >= 4]
        assert vals, label
        return min(vals)

    # the nominal rates of the ceiling are never beaten by a measurement (so the ceiling is optimistic, the fraction a lower bound)
    assert best("v_fma_f32 v,v,v,v") >= bench.VALU_CYCLES["fast"] and best("v_mul_f32 v,v,v (VOP2)") >= bench.VALU_CYCLES["fast"]
    assert best("v_fma_f32 v,v,v,v") < 2.6  # ... and the fast class really is the 2-cycle class
    for slow in ("v_fma_f32 v,s,v,v", "v_max3_f32", "v_cmp_le_f32 -> SGPR pair (VOP3)", "v_cndmask_b32 SGPR mask (VOP3)", "v_pk_fma_f32", "v_readlane_b32",
                 "v_cvt_f32_u32", "v_min_u32 (VOP2)"):
        assert bench.VALU_CYCLES["slow"] <= best(slow) < 5.0, slow
    assert best("v_rcp_f32") >= bench.VALU_CYCLES["trans"]


def test_traffic_is_keyed_on_the_profiled_workload(bench):
    ns = argparse.Namespace(workload="vocalfold", res=512, spp=64, fp16=False, no_shadows=False, material="principled")
    assert bench._is_profiled_workload(ns)
    for k, v in (("workload", "colon"), ("res", 1024), ("spp", 256), ("fp16", True), ("no_shadows", True), ("material", "diffuse")):
        other = argparse.Namespace(**{**vars(ns), k: v})
        assert not bench._is_profiled_workload(other), k
    t = bench.pmc_traffic("k_render_fwd_pk")
    newest = os.path.basename(bench._profile_files("pmc_summary.json")[-1]).split("_")[0]
    assert t is not None and t["source"].startswith(newest + "_") and t["bytes"] >= t["raw_bytes"] > 5.6e6  # more than the algorithmic 5.6 MB
    # the gradient bracket's passes: K9 where the step runs it, the pattern launch where forward and adjoint are one launch
    g = bench.pmc_traffic("k_render_bwd_cached", "grad") or bench.pmc_traffic("k_pattern_bwd", "grad")
    assert g is not None and g["source"].startswith(newest + "grad_")
    # the colon's figures come from its own passes only (profiles/r<N>colon_*), never from the default workload's
    c = bench.pmc_traffic("k_render_fwd_pk", "r", "colon")
    assert c is None or "colon" in c["source"]


def _run_bench(args, env_extra, timeout=300):
    import subprocess
    import sys

    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "FFX_DIST_BACKEND")}
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, cwd=ROOT, capture_output=True, text=True, timeout=timeout)


def test_self_launcher_refuses_more_ranks_than_devices_and_propagates_failures():
    """`python bench.py --gpus N` without a torchrun environment starts N rank processes itself (it used to measure ONE
    GPU and label it n_gpus 1).  On this GPU-less box: more ranks than devices is refused unless FFX_DIST_BACKEND=gloo
    asks for a rehearsal, and ranks that fail (no HIP device here) end the run with a non-zero code — no line is printed,
    nothing is restarted."""
    import torch

    if torch.cuda.device_count() > 0:
        pytest.skip("a GPU box: covered by the -m gpu test")
    r = _run_bench(["--gpus", "2", "--steps", "2"], {})
    assert r.returncode != 0 and "only 0 device(s)" in r.stderr and r.stdout.strip() == ""
    r = _run_bench(["--gpus", "2", "--steps", "2"], {"FFX_DIST_BACKEND": "gloo"})
    assert r.returncode != 0 and r.stdout.strip() == ""
    # the ranks were started and fail loudly; the launcher terminates the others as soon as the first one has died, so the second
    # message may or may not have been printed by then (it was a flaky == 2 once)
    assert 1 <= r.stderr.count("bench.py needs a HIP device") <= 2
    # inside a torchrun environment the world size must match --gpus
    r = _run_bench(["--gpus", "4"], {"RANK": "0", "WORLD_SIZE": "2", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr


def test_every_profile_the_bench_line_cites_exists_is_not_empty_and_names_kernels_of_this_tree(bench):
    """round 3 committed a 0-byte profiles/r3_phaseclk.txt and a gradient PMC summary that listed kernels the step no longer launched,
    and the bench line cited both.  bench.cited_profiles() is what a default line's `evidence` field lists: every file must exist, be
    non-empty, come from ONE round's collection (tools/collect_profiles.sh) and name only kernels that exist in csrc/ today — and the
    gradient bracket's summary must contain the launches a gradient step makes now."""
    files = bench.cited_profiles("")
    assert files, "no committed profile set"
    kernels = set()
    for src in ("ffx_trace.hip", "ffx_splat.hip", "ffx_scene.hip", "ffx_bins.hip"):
        txt = open(os.path.join(ROOT, "fireflies_amd", "csrc", src)).read()
        kernels |= set(re.findall(r"__global__[^;{]*?\b(k_[A-Za-z0-9_]+)\s*\(", txt, flags=re.S))
    assert {"k_render_fwd_pk", "k_bin", "k_pattern_bwd", "k_render_bwd_cached_tiled16"} <= kernels
    tags = set()
    for rel in files:
        path = os.path.join(ROOT, rel)
        assert os.path.isfile(path) and os.path.getsize(path) > 0, f"{rel}: cited by the bench line but missing or empty"
        name = os.path.basename(rel)
        if any(name.endswith(sfx) for sfx in ("pmc_summary.json", "sq_instruction_mix.json", "kernel_stats.csv", "phaseclk.txt")):
            tags.add(re.match(r"(r\d+)", name).group(1))
    assert len(tags) == 1, f"the cited kernel profiles come from different rounds: {sorted(tags)}"
    (tag,) = tags

    def base(k):
        return k.replace("void ", "").split("<")[0].split("(")[0].strip()

    for name in (f"{tag}_pmc_summary.json", f"{tag}grad_pmc_summary.json", f"{tag}_sq_instruction_mix.json"):
        d = json.load(open(os.path.join(ROOT, "profiles", name)))
        stale = sorted({base(k) for k in d if base(k).startswith("k_")} - kernels)
        assert not stale, f"profiles/{name} names kernels that are not in csrc/ any more: {stale}"
    g = json.load(open(os.path.join(ROOT, "profiles", f"{tag}grad_pmc_summary.json")))
    assert any(bench.k8_instance(k)[0] for k in g), "no counters for the forward + adjoint launch the gradient bracket times"
    for want in ("k_pattern_step<5>", "k_render_bwd_cached", "k_bin<"):  # (round 6: the pattern side of a step is one launch)
        assert any(k.startswith(want) for k in g), f"the gradient bracket's PMC summary lacks {want}"
    stats = open(os.path.join(ROOT, "profiles", f"{tag}_kernel_stats.csv")).read()
    assert "k_render_fwd_pk<1, true, 1, false, false" in stats and "k_bin<true" in stats  # (<R, WIDE, MATM, ADJ, RF[, RFC]>: the plain forward)
    # ... and what the line derives from them resolves to that same set
    t = bench.pmc_traffic("k_render_fwd_pk", "grad", "", adjoint_instance=True)
    assert t is not None and t["source"] == f"{tag}grad_pmc_summary.json"
    assert bench.pmc_traffic("k_render_bwd_cached", "grad", "")["source"] == f"{tag}grad_pmc_summary.json"


def test_committed_bench_line_is_consistent_with_itself(bench):
    """Round-4 review, weak 4: the line printed `frac` from a per-launch time (0.534 ms) that EXCEEDED the step time (0.400 ms) — in-loop launches
    overlap on two render streams and their durations are concurrency-stretched.  Since round 5 the roofline is held against the kernel ALONE
    (device drained between launches) and the stretched figure is kept under its own name.  On the newest committed line of the default
    workload (profiles/r<N>_bench_under_rocprof.json): the kernel's time does not exceed the step's, frac = achieved / peak = algorithmic bytes /
    time / 8 TB/s, traffic_ratio = counter bytes / algorithmic bytes, and the step's traffic is the sum of its kernels'."""
    f = bench._profile_files("bench_under_rocprof.json")[-1]
    d = json.load(open(f))
    r = d["roofline"]
    if "avg_kernel_ms_overlapped" not in r:
        pytest.skip(f"{os.path.basename(f)} predates the round-5 roofline fields")
    assert r["avg_kernel_ms"] == r["kernel_alone_ms"] and r["kernel_alone_ms"] <= 1.02 * d["ms_per_step"], (r["kernel_alone_ms"], d["ms_per_step"])
    assert r["avg_kernel_ms_overlapped"] >= r["kernel_alone_ms"]
    assert r["achieved"] == pytest.approx(r["algorithmic_bytes_per_launch"] / (r["avg_kernel_ms"] * 1e-3) / 1e9, rel=1e-9)
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"], rel=1e-12) and r["peak"] == 8000.0 and r["bound"] == "hbm"
    assert r["traffic_ratio"] == pytest.approx(r["traffic"] / r["algorithmic_bytes_per_launch"], rel=1e-12)
    st = r["step_traffic"]
    assert st["bytes_per_step"] == pytest.approx(sum(k["bytes_per_launch"] * k["launches_per_step"] for k in st["per_kernel"].values()), rel=1e-12)
    assert st["traffic_ratio"] == pytest.approx(st["bytes_per_step"] / st["algorithmic_bytes_per_step"], rel=1e-12)
    assert any(k.startswith("k_bin<true") for k in st["per_kernel"]) and any(k.startswith("k_scene_update_fused") for k in st["per_kernel"])
    vi = r["valu_issue"]
    assert vi["kernel_ms"] == r["kernel_alone_ms"] and 0.5 < vi["frac"] <= 1.0
    tag = os.path.basename(f).split("_")[0]
    assert f"{tag}_sq_instruction_mix.json" in vi["source"] and f"{tag}_issue_rates.txt" in vi["source"], vi["source"]  # (the line cites its own round's files)


def test_committed_driver_line_carries_configs_3_and_4(bench):
    """Round-5 review, item 5: BASELINE configs[3] (32 scene samples per gradient step) and configs[4] (the colon at 1024 x 1024 x 256 spp, fp16 film)
    are timed inside the DEFAULT bench line — what the driver runs — as informational brackets, the colon with a roofline of its own from the
    committed r<N>colon_* passes.  On the newest committed driver-style line (profiles/r<N>_bench_line.json, written by tools/collect_profiles.sh):
    the keys are there, the colon's fraction is its algorithmic bytes over its own kernel time over 8 TB/s, and its traffic is the committed figure."""
    ff = bench._profile_files("bench_line.json")
    if not ff:
        pytest.skip("no committed profiles/r<N>_bench_line.json yet")
    d = json.load(open(ff[-1]))
    assert d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 5
    assert d["grad_samples_per_sec_s32"] > 0 and d["grad_s32_config"]["step_paths"]
    assert d["grad_samples_per_sec_s32"] >= 0.9 * d["grad_steps_per_sec"]  # (32 samples amortise the pattern side: at least the one-sample step's pace)
    assert d["value_colon"] > 0 and d["ms_per_step_colon"] == pytest.approx(1e3 / d["value_colon"], rel=1e-9)
    r = d["colon_roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["frac"] == pytest.approx(r["achieved"] / r["peak"], rel=1e-12)
    assert r["achieved"] == pytest.approx(r["algorithmic_bytes_per_launch"] / (r["avg_kernel_ms"] * 1e-3) / 1e9, rel=1e-9)
    assert r["kernel_alone_ms"] <= 1.05 * d["ms_per_step_colon"]
    tr = bench.pmc_traffic("k_render_fwd_pk", "r", "colon")
    if tr is not None and r["traffic"] is not None and os.path.basename(tr["source"]) == os.path.basename(r["traffic_source"]):
        assert r["traffic"] == pytest.approx(tr["bytes"], rel=1e-12) and r["traffic_ratio"] == pytest.approx(tr["bytes"] / r["algorithmic_bytes_per_launch"], rel=1e-12)
    assert any("colon" in f for f in d["evidence"])
