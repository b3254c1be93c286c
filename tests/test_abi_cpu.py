"""CPU-side checks of the product library: it loads without a GPU, exports every symbol that
include/ffx.h declares, refuses non-device tensors loudly, and its host BVH builder emits a
structurally valid tree.  No device compute here."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

from fireflies_amd import _abi, _lib, ops, scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    txt = open(os.path.join(ROOT, "include", "ffx.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(ffx_[a-z0-9_]+)\s*\(", txt)))


def test_header_and_python_abi_agree():
    assert _header_symbols() == sorted(_abi.PROTOTYPES)


@pytest.mark.parametrize("which", ["hip", "oracle"])
def test_library_exports_every_symbol(which, oracle):
    path = _lib.LIB_PATH if which == "hip" else oracle.LIB_PATH
    assert os.path.exists(path), f"{path} not built"
    lib = C.CDLL(path)
    for name in _header_symbols():
        assert hasattr(lib, name), f"{which}: missing symbol {name}"
    lib.ffx_backend.restype = C.c_char_p
    assert lib.ffx_backend().decode() == ("hip-gfx950" if which == "hip" else "cpu-oracle")
    assert lib.ffx_abi_version() == _abi.FFX_ABI_VERSION


@pytest.mark.parametrize("which", ["hip", "oracle"])
def test_host_philox_known_answers(which, oracle):
    """ffx_torch_rand_h (a host function in both libraries): Random123's published known-answer vector for
    Philox-4x32-10 (counter 0, key 0 -> first word 0x6627e8d5) through the float mapping, the 1 -> 0 fold's domain,
    counter placement (offset / 4 in the low words, element index in the third), and the refusals.  The match
    with torch.rand itself is a GPU test (tests/test_api_gpu.py::test_host_philox_matches_torch_rand)."""
    lib = C.CDLL(_lib.LIB_PATH if which == "hip" else oracle.LIB_PATH)
    fn = lib.ffx_torch_rand_h
    fn.argtypes = [C.c_uint64, C.c_uint64, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_uint64)]
    out, inc = (C.c_float * 256)(), C.c_uint64()
    assert fn(0, 0, 3, out, C.byref(inc)) == 0 and inc.value == 4
    assert out[0] == np.float32(np.float32(0x6627E8D5) * np.float32(2.0**-32) + np.float32(2.0**-32))

    def philox(ctr, key):  # numpy restatement of the published round function, for the remaining words
        c, k = [int(x) for x in ctr], [int(x) for x in key]
        for r in range(10):
            p0, p1 = 0xD2511F53 * c[0], 0xCD9E8D57 * c[2]
            c = [(p1 >> 32) ^ c[1] ^ k[0], p1 & 0xFFFFFFFF, (p0 >> 32) ^ c[3] ^ k[1], p0 & 0xFFFFFFFF]
            k = [(k[0] + 0x9E3779B9) & 0xFFFFFFFF, (k[1] + 0xBB67AE85) & 0xFFFFFFFF]
        return c

    assert philox([0, 0, 0, 0], [0, 0]) == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]  # Random123 kat_vectors
    assert philox([0xFFFFFFFF] * 4, [0xFFFFFFFF] * 2) == [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]
    seed, off = (0x299F31D0 << 32) | 0xA4093822, 4 * ((0x05A308D3 << 32) | 0x243F6A88)
    assert fn(seed, off, 256, out, C.byref(inc)) == 0
    for i in (0, 1, 2, 100, 255):
        w = philox([0x243F6A88, 0x05A308D3, i, 0], [0xA4093822, 0x299F31D0])[0]
        want = np.float32(np.float32(w) * np.float32(2.0**-32) + np.float32(2.0**-32))
        assert out[i] == (np.float32(0.0) if want == 1.0 else want), i
    assert all(0.0 <= out[i] < 1.0 for i in range(256))
    assert fn(0, 0, 257, out, C.byref(inc)) == _abi_err("FFX_ERR_UNSUPPORTED") and fn(0, 2, 3, out, C.byref(inc)) == _abi_err("FFX_ERR_UNSUPPORTED")
    assert fn(0, 0, 0, out, C.byref(inc)) == _abi_err("FFX_ERR_ARG")
    # the batched entry point = the single one, draw by draw
    fb = lib.ffx_torch_rand_batch_h
    fb.argtypes = [C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_int32), C.POINTER(C.c_float)]
    seeds, offs, cnts = (C.c_uint64 * 3)(5, 5, 77), (C.c_uint64 * 3)(0, 4, 1024), (C.c_int32 * 3)(3, 1, 7)
    packed = (C.c_float * 11)()
    assert fb(3, seeds, offs, cnts, packed) == 0
    p = 0
    for sd_, of_, n in zip(seeds, offs, cnts):
        assert fn(sd_, of_, n, out, C.byref(inc)) == 0
        assert [packed[p + i] for i in range(n)] == [out[i] for i in range(n)]
        p += n
    assert fb(0, None, None, None, None) == 0 and fb(1, seeds, offs, (C.c_int32 * 1)(300), packed) == _abi_err("FFX_ERR_UNSUPPORTED")


def _abi_err(name):
    return {"FFX_ERR_ARG": -1, "FFX_ERR_LAUNCH": -2, "FFX_ERR_UNSUPPORTED": -3, "FFX_ERR_NOMEM": -4}[name]


def test_struct_sizes_match_header():
    # sizes implied by include/ffx.h (all members are 4- or 8-byte scalars)
    assert C.sizeof(_abi.Camera) == 4 * (16 + 16 + 2 + 2)
    assert C.sizeof(_abi.Projector) == 4 * (16 + 16 + 1 + 3 + 4)
    assert C.sizeof(_abi.Spot) == 4 * (16 + 3 + 2 + 1)
    # shadows, n_shapes, mat_stride; n_base_tex + 2 x 4 sizes, 4 texture pointers, slot_uv; n_mat_h + 128 inline material floats (+ 4 of padding)
    assert C.sizeof(_abi.SceneDesc) == C.sizeof(_abi.Camera) + C.sizeof(_abi.Projector) + C.sizeof(_abi.Spot) + 12 + 4 + 32 + 32 + 8 + 4 + 512 + 4 + 4 + 4  # (... n_mat_h, mat_h, rfilter, rfilter_stddev, tail padding)
    # 4 ints, 5 offsets, level_start, 4 ints of the wide overlay (+ 4 bytes of padding to 8), 4 offsets; ABI 4: the refit plan's offset + 2 ints
    assert C.sizeof(_abi.BvhInfo) == 16 + 5 * 8 + 4 * (_abi.FFX_MAX_LEVELS + 1) + 16 + 4 + 4 * 8 + 8 + 8 + 8 + 8 + 16  # (+ off_nrec, off_gn; ABI 5: off_bins, bins_stride)
    assert C.sizeof(_abi.Smooth) == 8 * 4 + 4 + 4 + 8  # four pointers, n_vn (+ padding), one pointer


def test_no_cpu_fallback():
    pts = torch.rand(4, 2)
    with pytest.raises(RuntimeError, match="HIP device"):
        ops.splat_fwd(pts, 10.0, "sum", -1, 16, 16)
    with pytest.raises(RuntimeError, match="HIP device"):
        ops.project_rays_fwd(torch.rand(4, 3), np.eye(4))


def test_error_reporting_through_the_abi():
    a = _lib.api()
    info = _abi.BvhInfo()
    with pytest.raises(_abi.FFXError, match="bad argument"):
        a.call("ffx_bvh_build_host", None, 0, None, 0, None, 0, C.byref(info))
    v = np.zeros((3, 3), np.float32)
    t = np.array([[0, 1, 7]], np.int32)
    blob = np.zeros(a.lib.ffx_bvh_blob_bytes(1), np.uint8)
    with pytest.raises(_abi.FFXError, match="out of range"):
        a.call("ffx_bvh_build_host", v.ctypes.data, 3, t.ctypes.data, 1, blob.ctypes.data, blob.size, C.byref(info))
    with pytest.raises(_abi.FFXError, match="too small"):
        a.call("ffx_bvh_build_host", v.ctypes.data, 3, np.array([[0, 1, 2]], np.int32).ctypes.data, 1, blob.ctypes.data, 8, C.byref(info))


def test_adjoint_cache_size_follows_the_material_table():
    """host-side sizing (no GPU work): Lambert scenes keep the 37.7 MB layout at the BASELINE size, material rows add one
    112-byte footprint per pixel behind the stray arena; the scene-description entry point agrees with the plain one"""
    a = _lib.api()
    sd = _abi.SceneDesc()
    sd.cam.width, sd.cam.height = 512, 512
    plain = a.lib.ffx_render_cache_bytes(512, 512, 64)
    assert plain <= 40 * 10**6
    assert a.lib.ffx_render_cache_bytes_sd(C.byref(sd), 64) == plain
    sd.mat_stride = 3
    assert a.lib.ffx_render_cache_bytes_sd(C.byref(sd), 64) == plain
    sd.mat_stride = _abi.MAT_STRIDE
    up = lambda v: ((v + 127) // 128) * 128  # noqa: E731
    assert a.lib.ffx_render_cache_bytes_sd(C.byref(sd), 64) == up(plain) + 112 * 512 * 512
    assert a.lib.ffx_render_cache_bytes_sd(None, 64) == 0 and a.lib.ffx_render_cache_bytes(0, 4, 4) == 0
    # the material columns of the Python mirror are the header's
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "ffx.h")).read()
    import re

    for name in ("MODEL", "ROUGHNESS", "ANISOTROPIC", "METALLIC", "SPEC_TRANS", "ETA", "SPEC_TINT", "SHEEN", "SHEEN_TINT", "FLATNESS", "CLEARCOAT",
                 "CLEARCOAT_GLOSS"):
        col = int(re.search(rf"#define FFX_MAT_{name} (\d+)", hdr).group(1))
        assert getattr(_abi, "MAT_" + name) == col
        from fireflies_amd import scenes

        assert scenes.MAT_COLUMN[name.lower()] == col
    assert int(re.search(r"#define FFX_MAT_STRIDE (\d+)", hdr).group(1)) == _abi.MAT_STRIDE


def _build(verts, tris):
    a = _lib.api()
    F = tris.shape[0]
    n = a.lib.ffx_bvh_blob_bytes(F)
    blob = np.zeros(n, np.uint8)
    info = _abi.BvhInfo()
    a.call("ffx_bvh_build_host", verts.ctypes.data, verts.shape[0], tris.ctypes.data, F, blob.ctypes.data, n, C.byref(info))
    return blob, info


@pytest.mark.parametrize("case", ["single", "leaf", "vocalfold", "degenerate"])
def test_host_bvh_structure(case):
    if case == "single":
        verts = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32)
        tris = np.array([[0, 1, 2]], np.int32)
    elif case == "leaf":
        verts, tris = scenes.make_plane(1.0, 1.0, 1, 1)
    elif case == "degenerate":  # every centroid identical -> median fallback must terminate
        verts = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32)
        tris = np.tile(np.array([[0, 1, 2]], np.int32), (300, 1))
    else:
        sc = scenes.vocalfold(frames=1)
        pool, tr, shape, off, stride, nfr, alb = scenes.flatten(sc)
        verts, tris = pool, (tr + off[shape][:, None]).astype(np.int32)
    verts = np.ascontiguousarray(verts, np.float32)
    tris = np.ascontiguousarray(tris, np.int32)
    blob, info = _build(verts, tris)
    F = tris.shape[0]
    assert info.n_tris == F and 1 <= info.n_nodes <= max(1, F)
    assert info.max_depth <= 48 and info.total_bytes <= blob.size
    nodes = blob[info.off_nodes : info.off_nodes + 64 * info.n_nodes].view(np.int32).reshape(-1, 16)
    order = blob[info.off_order : info.off_order + 4 * F].view(np.int32)
    refit = blob[info.off_refit : info.off_refit + 4 * info.n_nodes].view(np.int32)
    assert sorted(order.tolist()) == list(range(F))  # a permutation of the triangles
    assert sorted(refit.tolist()) == list(range(info.n_nodes))
    assert info.level_start[0] == 0 and info.level_start[info.n_levels] == info.n_nodes
    # every leaf slot covered exactly once; every inner node referenced exactly once (except root)
    covered = np.zeros(F, np.int32)
    refs = np.zeros(info.n_nodes, np.int32)
    height = np.zeros(info.n_nodes, np.int32)
    level_of = np.zeros(info.n_nodes, np.int32)
    for l in range(info.n_levels):
        level_of[refit[info.level_start[l] : info.level_start[l + 1]]] = l
    EMPTY = -(2**31)
    for k in range(info.n_nodes):
        for c in (int(nodes[k, 12]), int(nodes[k, 13])):
            if c == EMPTY:
                continue
            if c < 0:
                lc = (~c) & 0xFFFFFFFF
                first, cnt = lc >> 3, (lc & 7) + 1
                assert cnt <= 4 and first + cnt <= F
                covered[first : first + cnt] += 1
            else:
                assert 0 < c < info.n_nodes
                refs[c] += 1
                assert level_of[c] < level_of[k]  # children are refitted before parents
    assert (covered == 1).all()
    assert refs[0] == 0 and (refs[1:] == 1).all()
    # ---- the 64-wide overlay: clusters reachable from wide_root tile the leaf slots exactly once, every
    # wide node is referenced once, every child knows the binary node whose box it mirrors
    nw = info.n_wide
    ESZ = 32  # bytes per wide element: float32 box {lo[3], hi[3]}, ref, pad (ffx_common.h: WideChild)
    wn = blob[info.off_wnodes : info.off_wnodes + 64 * ESZ * nw].view(np.int32).reshape(nw, 64, ESZ // 4)
    wsrc = blob[info.off_wsrc : info.off_wsrc + 256 * nw].view(np.int32).reshape(nw, 64)
    wcov = np.zeros(F, np.int32)
    wrefs = np.zeros(max(nw, 1), np.int32)
    depth_seen = 0

    tq0 = 64 * nw
    assert info.off_tq == info.off_wnodes + ESZ * tq0

    def visit(ref, depth):
        nonlocal depth_seen
        ref &= 0xFFFFFFFF
        cluster, elem, cnt = ref >> 31, (ref >> 6) & 0x1FFFFFF, (ref & 63) + 1
        if cluster:
            first = elem - tq0
            assert 0 <= first and first + cnt <= F
            wcov[first : first + cnt] += 1
            return
        assert elem % 64 == 0
        idx = elem // 64
        assert 0 <= idx < nw and 2 <= cnt <= 64
        wrefs[idx] += 1
        depth_seen = max(depth_seen, depth + 1)
        for j in range(64):
            if j < cnt:
                src = int(wsrc[idx, j])
                assert 0 <= src < 2 * info.n_nodes
                visit(int(wn[idx, j, 6]), depth + 1)
            else:
                assert int(wsrc[idx, j]) == -1

    visit(int(info.wide_root), 0)
    assert (wcov == 1).all()
    assert nw == 0 or (wrefs == 1).all()
    assert depth_seen == info.wide_depth <= 6
    assert (info.n_wide == 0) == (F <= 64)
    assert info.off_whdr + 64 <= info.total_bytes and info.off_tq + ESZ * F <= info.off_wsrc
    # ---- the refit plan (ABI 4): treelets tile the leaf slots and the nodes exactly once; inside a treelet (and inside the
    # top) children are re-fitted before their parents; a treelet's nodes only depend on its own nodes and its own slots;
    # every used wide child is copied by the treelet (or the top) that owns the binary node holding its box
    T = info.n_treelets
    plan = blob[info.off_plan : info.off_plan + 4 * info.plan_ints].view(np.int32)
    assert T >= 1 and plan[-1] == 0 and info.off_plan + 4 * info.plan_ints <= info.total_bytes
    hdr = plan[: 8 * (T + 1)].reshape(T + 1, 8)
    slot_cov = np.zeros(F, np.int32)
    owner = np.full(info.n_nodes, -1, np.int32)
    step_of = np.zeros(info.n_nodes, np.int64)
    for t in range(T + 1):
        s0, sn, l0, nl, w0, wc = (int(v) for v in hdr[t, :6])
        slot_cov[s0 : s0 + sn] += 1
        assert (t < T) or sn == 0  # the top owns no slots
        for l in range(nl):
            b, e = int(plan[l0 + l]), int(plan[l0 + l + 1])
            assert b < e
            for k in plan[b:e]:
                assert owner[k] == -1
                owner[k] = t
                step_of[k] = l
    assert (slot_cov == 1).all() and (owner >= 0).all()
    for k in range(info.n_nodes):
        t = owner[k]
        for c in (int(nodes[k, 12]), int(nodes[k, 13])):
            if c == EMPTY:
                continue
            if c < 0:
                first = ((~c) & 0xFFFFFFFF) >> 3
                if t < T:  # a treelet only reads records it built itself
                    assert hdr[t, 0] <= first < hdr[t, 0] + hdr[t, 1]
            elif t < T:
                assert owner[c] == t and step_of[c] < step_of[k]
            else:  # the top reads finished treelets, or top nodes of a lower step
                assert owner[c] < T or step_of[c] < step_of[k]
    used = sorted(int(i) * 64 + j for i in range(nw) for j in range(64) if wsrc[i, j] >= 0)
    copied = []
    for t in range(T + 1):
        w0, wc = int(hdr[t, 4]), int(hdr[t, 5])
        for k in plan[w0 : w0 + wc]:
            copied.append(int(k))
            assert owner[int(wsrc.reshape(-1)[k]) >> 1] == t
    assert sorted(copied) == used
    if case == "vocalfold":
        assert 52 <= T <= 400 and int(hdr[:T, 1].max()) <= 1024


def test_ctypes_mirrors_have_the_compilers_layout(tmp_path):
    """sizeof and the offset of every field of every struct of include/ffx.h as gcc lays them out, against the ctypes mirrors of
    fireflies_amd/_abi.py (a field inserted on one side only would shift everything behind it silently)"""
    import subprocess

    pairs = {"ffx_bvh_info": _abi.BvhInfo, "ffx_smooth": _abi.Smooth, "ffx_camera": _abi.Camera, "ffx_projector": _abi.Projector, "ffx_spot": _abi.Spot,
             "ffx_scene_desc": _abi.SceneDesc, "ffx_adam_args": _abi.AdamArgs}
    lines = ["#include <stdio.h>", "#include <stddef.h>", '#include "ffx.h"', "int main(void) {"]
    for cname, cls in pairs.items():
        lines.append(f'  printf("{cname} %zu\\n", sizeof({cname}));')
        for fname, _ in cls._fields_:
            lines.append(f'  printf("{cname}.{fname} %zu\\n", offsetof({cname}, {fname}));')
    lines += ["  return 0;", "}"]
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    got = dict(l.split() for l in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    for cname, cls in pairs.items():
        assert int(got[cname]) == C.sizeof(cls), cname
        for fname, _ in cls._fields_:
            assert int(got[f"{cname}.{fname}"]) == getattr(cls, fname).offset, f"{cname}.{fname}"
