#!/usr/bin/env python3
"""Operand forms of the fp32 add / mul / fma instructions of a kernel, from its disassembly (no GPU needed).

The SQ counters price a wave instruction by its class, not by its operands — but on gfx950 a v_fma/v_mul/v_add/v_fmac/v_sub_f32 issues
every 2 cycles only when all sources are VGPRs; with an SGPR, a literal or an inline constant as a source it issues every 4
(profiles/r3_issue_rates.txt).  A wave-packet traversal keeps its uniform data (node planes, apex records, scene constants) in
SGPRs, so a good part of its "fast" class is not.  This tool compiles fireflies_amd/csrc/ffx_trace.hip to ISA with the library's
flags, takes the render kernel, and counts per opcode how many instructions carry a scalar source — statically (every instruction
once; the hot loops — the 64-wide step: 3 of 6 fmas, the exact triangle test: 9 of 9 — are not below the average).

    python tools/isa_mix.py [kernel-substring] > profiles/r<N>_isa_operand_forms.json
bench.py (valu_issue) uses the newest such file to price that share of the fast class at 4 cycles ("ceiling_ms_operand_forms").
"""
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAST = ("v_fma_f32", "v_fmac_f32", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mac_f32", "v_mad_f32")


def main():
    want = sys.argv[1] if len(sys.argv) > 1 else "k_render_fwd_pk<1, true, 1, false, false, false>"  # (one pixel per packet, 64-wide walk, material rows, plain forward)
    src = os.path.join(ROOT, "fireflies_amd", "csrc", "ffx_trace.hip")
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "t.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-slp-vectorize", "-Wno-inline-asm",
                        "-I" + os.path.join(ROOT, "include"), "-S", "--cuda-device-only", src, "-o", out], check=True, stderr=subprocess.DEVNULL)
        text = open(out).read().split("\n")
    starts = [i for i, l in enumerate(text) if re.match(r"^_Z\w+:", l)]
    names = subprocess.run(["c++filt"], input="\n".join(text[i].split(":")[0] for i in starts), capture_output=True, text=True).stdout.split("\n")
    body = None
    for k, i in enumerate(starts):
        if want in names[k]:
            end = next(j for j in range(i, len(text)) if text[j].strip().startswith(".Lfunc_end"))
            body, full = text[i:end], names[k]
            break
    if body is None:
        raise SystemExit(f"no kernel matching {want!r}")
    cnt, scal = collections.Counter(), collections.Counter()
    n_valu = n_salu = 0
    for l in body:
        l = l.split(";")[0].strip()
        if not l or l.startswith(".") or l.endswith(":"):
            continue
        op, _, rest = l.partition(" ")
        n_valu += op.startswith("v_")
        n_salu += op.startswith("s_")
        base = re.sub(r"_(e32|e64|dpp|sdwa)$", "", op)
        if base not in FAST:
            continue
        cnt[base] += 1
        srcs = [a.strip() for a in rest.split(",")][1:]  # (the first operand is the destination)
        is_scalar = any(re.match(r"^-?\|?(s\d+|s\[|vcc|exec|m0|0x|-?\d|lit)", a) or re.match(r"^-?\|?-?\d*\.\d", a) for a in srcs)
        scal[base] += bool(is_scalar)
    tot, sc = sum(cnt.values()), sum(scal.values())
    json.dump({"kernel": full[:120], "static_valu_instructions": n_valu, "static_salu_instructions": n_salu, "fp32_add_mul_fma": tot, "with_scalar_or_constant_source": sc,
               "scalar_source_fraction": sc / max(tot, 1), "per_opcode": {k: {"total": cnt[k], "with_scalar_source": scal[k]} for k in sorted(cnt)},
               "note": "static counts over the kernel's whole body (every instruction once), library flags; see tools/isa_mix.py"}, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
