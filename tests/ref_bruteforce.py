"""Independent brute-force restatement of K7 / K8 / K9 — TEST INFRASTRUCTURE, float64 numpy.

Purpose: the oracle for the Mitsuba half of the path (oracle/ffx_oracle.c) and the HIP
kernels were written together "in the same operation order", so a shared conceptual error would be invisible to
every HIP-vs-oracle test.  This module was written from the *published conventions* (SURVEY.md Appendix A,
DESIGN.md §4) and NOT from ffx_oracle.c: no BVH (every ray against every triangle), the textbook
Moller-Trumbore test with the ray's own origin (not the apex form), float64 throughout, numpy broadcasting
instead of loops.  tests/test_bruteforce_cpu.py compares it with the oracle on small scenes: depth, ids,
radiance and the texture gradient.

What is restated (each with the convention it comes from):
  * sensor rays [SURVEY App. A `sensor.sample_ray`]: sample position ((x + jx)/W, (y + jy)/H), no half-pixel
    offset (fireflies/graphics/depth.py:61-69); near-plane point = camera_to_sample^-1 (sx, sy, 0); direction =
    to_world * normalize(that point); origin = camera position; valid range (near / d_z, far / d_z], reported t
    measured from the near plane.
  * jitter: the counter-based hash of DESIGN.md §4.2 (this is OUR definition; it has to be shared for a
    per-pixel comparison to be possible at all).
  * closest hit with ties broken by the smaller primitive id.
  * shading [SURVEY §7.3 / DESIGN §4.3]: direct light at the primary hit, geometric normal faced to the viewer,
    Lambert; projector = perspective frustum with a bilinear, clamp-to-edge irradiance texture, radiance
    albedo * tex(uv) * scale / (z_local^2 cos_p) * cos_s inside uv in [0,1]^2; spot = albedo / pi * I * falloff / d^2
    * cos_s with falloff 1 inside beam_width and linear in the ANGLE down to 0 at the cutoff; shadow rays from
    the emitter to the surface point lifted by (1 + max|P|) * 1500 * 2^-24 along the normal, occluded by any hit at
    0 < t < 1 - 10 * 1500 * 2^-24; box reconstruction filter (plain mean over the samples).
  * materials: rows of 3 floats are Lambert albedos; rows of 16 floats (include/ffx.h FFX_MAT_*) with model 1 are
    Mitsuba's `principled` BSDF, written here from the structure of its eval() (Burley 2012 / 2015 and the Mitsuba 3
    documentation of the plugin) as RGB arithmetic lobe by lobe — not as the oracle's "base_color * A + B" split.
  * adjoint: the render is linear in the texture, so d loss / d tex scatters gimg * albedo * colour * geometric
    factor / spp through the four bilinear weights.
"""
import numpy as np

EPS = 1500.0 * 2.0**-24


def _m(x, n):
    return np.asarray(list(x), np.float64).reshape(n, n)


def _hash32(x):
    x = np.asarray(x, np.uint64) & 0xFFFFFFFF
    x ^= x >> 16
    x = (x * 0x7FEB352D) & 0xFFFFFFFF
    x ^= x >> 15
    x = (x * 0x846CA68B) & 0xFFFFFFFF
    x ^= x >> 16
    return x


def jitter(seed, idx):
    key = _hash32(np.uint64((seed + 0x9E3779B9) & 0xFFFFFFFF))
    a = _hash32(((2 * idx) & 0xFFFFFFFF) ^ key)
    b = _hash32(((2 * idx + 1) & 0xFFFFFFFF) ^ key)
    return (a >> 8).astype(np.float64) / 16777216.0, (b >> 8).astype(np.float64) / 16777216.0


def camera_rays(cam, spp, use_jitter, seed):
    W, H = cam.width, cam.height
    idx = np.arange(W * H * spp, dtype=np.uint64)
    pix = idx // spp
    x, y = (pix % W).astype(np.float64), (pix // W).astype(np.float64)
    jx, jy = jitter(seed, idx) if use_jitter else (0.0, 0.0)
    sx, sy = (x + jx) / W, (y + jy) / H
    s2c = np.linalg.inv(_m(cam.camera_to_sample, 4))
    q = np.stack([sx, sy, np.zeros_like(sx), np.ones_like(sx)], 1) @ s2c.T
    near_p = q[:, :3] / q[:, 3:4]
    dl = near_p / np.linalg.norm(near_p, axis=1, keepdims=True)
    tw = _m(cam.to_world, 4)
    d = dl @ tw[:3, :3].T
    o = np.broadcast_to(tw[:3, 3], d.shape)
    return o, d, cam.near_clip / dl[:, 2], cam.far_clip / dl[:, 2]


def intersect(o, d, tris, tmin, tmax, chunk=4096):
    """every ray against every triangle (v0, e1, e2 of shape [F,3]); closest hit in (tmin, tmax], ties -> smaller id.
    Returns t (inf on a miss) and the primitive id (-1)."""
    v0, e1, e2 = tris
    n = o.shape[0]
    t_best = np.full(n, np.inf)
    prim = np.full(n, -1, np.int64)
    for a in range(0, n, chunk):
        oo, dd = o[a : a + chunk, None, :], d[a : a + chunk, None, :]
        pv = np.cross(dd, e2[None])
        det = (e1[None] * pv).sum(-1)
        with np.errstate(divide="ignore", invalid="ignore"):
            inv = 1.0 / det
            tv = oo - v0[None]
            u = (tv * pv).sum(-1) * inv
            qv = np.cross(tv, e1[None])
            v = (dd * qv).sum(-1) * inv
            t = (e2[None] * qv).sum(-1) * inv
        ok = (det != 0) & (u >= 0) & (v >= 0) & (u + v <= 1) & (t > np.asarray(tmin)[a : a + chunk, None]) & (t <= np.asarray(tmax)[a : a + chunk, None])
        t = np.where(ok, t, np.inf)
        tm = t.min(1)
        first = np.argmax(t == tm[:, None], axis=1)  # smallest primitive id among equal distances
        hit = np.isfinite(tm)
        t_best[a : a + chunk] = tm
        prim[a : a + chunk] = np.where(hit, first, -1)
    return t_best, prim


def any_hit(o, d, tris, tmax, chunk=4096):
    t, _ = intersect(o, d, tris, np.zeros(o.shape[0]), np.full(o.shape[0], np.nextafter(tmax, 0.0)), chunk)
    return np.isfinite(t)


def world_triangles(verts, tri_idx):
    p = np.asarray(verts, np.float64)[np.asarray(tri_idx)]
    return p[:, 0], p[:, 1] - p[:, 0], p[:, 2] - p[:, 0]


def trace_primary(verts, tri_idx, tri_shape, cam, spp, use_jitter, seed):
    tris = world_triangles(verts, tri_idx)
    o, d, nt, ft = camera_rays(cam, spp, use_jitter, seed)
    t, prim = intersect(o, d, tris, nt, ft)
    hit = prim >= 0
    return np.where(hit, t - nt, 0.0), np.where(hit, np.asarray(tri_shape)[np.maximum(prim, 0)], -1), prim


def _bilinear_setup(u, v, tw, th):
    fx, fy = u * tw - 0.5, v * th - 0.5
    x0, y0 = np.floor(fx), np.floor(fy)
    ax, ay = fx - x0, fy - y0
    xi0, xi1 = np.clip(x0, 0, tw - 1).astype(int), np.clip(x0 + 1, 0, tw - 1).astype(int)
    yi0, yi1 = np.clip(y0, 0, th - 1).astype(int), np.clip(y0 + 1, 0, th - 1).astype(int)
    return (xi0, xi1, yi0, yi1), ((1 - ay) * (1 - ax), (1 - ay) * ax, ay * (1 - ax), ay * ax)


def _schlick_w(c):
    return np.clip(1.0 - c, 0.0, 1.0) ** 5


def _fresnel_dielectric(c, eta):
    """unpolarised Fresnel reflectance for a ray arriving from outside (cos >= 0) at relative index eta"""
    st2 = (1.0 - c * c) / (eta * eta)
    ct = np.sqrt(np.maximum(1.0 - st2, 0.0))
    c = np.abs(c)
    with np.errstate(divide="ignore", invalid="ignore"):
        rs = (c - eta * ct) / (c + eta * ct)
        rp = (ct - eta * c) / (ct + eta * c)
    F = 0.5 * (rs * rs + rp * rp)
    F = np.where(c == 0.0, 1.0, F)
    return np.where(eta == 1.0, 0.0, F), ct


def _onb(n):
    """Duff et al. 2017, 'Building an Orthonormal Basis, Revisited' (what Mitsuba's coordinate_system uses)"""
    sg = np.copysign(1.0, n[:, 2])
    a = -1.0 / (sg + n[:, 2])
    b = n[:, 0] * n[:, 1] * a
    s = np.stack([1.0 + sg * n[:, 0] ** 2 * a, sg * b, -sg * n[:, 0]], 1)
    t = np.stack([b, sg + n[:, 1] ** 2 * a, -n[:, 1]], 1)
    return s, t


def bsdf_cos(rows, n, wv, wl):
    """f(wv, wl) * cos_o per colour channel, [N,3].  rows [N,3] (Lambert albedo) or [N,16] (material rows)."""
    rows = np.asarray(rows, np.float64)
    cos_i, cos_o = (n * wv).sum(1), (n * wl).sum(1)
    base = rows[:, :3]
    lambert = base / np.pi * np.maximum(cos_o, 0.0)[:, None]
    if rows.shape[1] == 3:
        return lambert
    model, rough, aniso, metallic, spec_trans, eta, spec_tint, sheen, sheen_tint, flat, cc, ccg = (rows[:, k] for k in range(3, 15))
    front = (cos_i > 0) & (cos_o > 0)
    s, t = _onb(n)
    wi = np.stack([(wv * s).sum(1), (wv * t).sum(1), cos_i], 1)
    wo = np.stack([(wl * s).sum(1), (wl * t).sum(1), cos_o], 1)
    wh = wi + wo
    wh = wh / np.linalg.norm(wh, axis=1, keepdims=True)
    ih, oh = (wi * wh).sum(1), (wo * wh).sum(1)
    facing = (ih * cos_i > 0) & (oh * cos_o > 0)
    lum = base @ np.array([0.212671, 0.715160, 0.072169])
    tint = np.where((lum > 0)[:, None], base / np.where(lum > 0, lum, 1.0)[:, None], 1.0)
    value = np.zeros_like(base)
    # ---- specular reflection: GGX (anisotropic), separable Smith, principled Fresnel
    aspect = np.sqrt(1.0 - 0.9 * aniso)
    ax, ay = np.maximum(0.001, rough**2 / aspect), np.maximum(0.001, rough**2 * aspect)
    with np.errstate(divide="ignore", invalid="ignore"):
        D = 1.0 / (np.pi * ax * ay * ((wh[:, 0] / ax) ** 2 + (wh[:, 1] / ay) ** 2 + wh[:, 2] ** 2) ** 2)
        D = np.where(D * wh[:, 2] > 1e-20, D, 0.0)

        def g1(v):
            xy = (ax * v[:, 0]) ** 2 + (ay * v[:, 1]) ** 2
            r = 2.0 / (1.0 + np.sqrt(1.0 + xy / v[:, 2] ** 2))
            r = np.where(xy == 0, 1.0, r)
            return np.where((v * wh).sum(1) * v[:, 2] <= 0, 0.0, r)

        G = g1(wi) * g1(wo)
        F_d, cos_t = _fresnel_dielectric(ih, eta)
        # Schlick weight the way the plugin's Schlick helper picks its cosine: the incident one for eta > 1, else the transmitted
        sw = np.where(eta > 1.0, _schlick_w(np.abs(ih)), _schlick_w(cos_t))[:, None]
        R0 = ((eta - 1.0) / (eta + 1.0)) ** 2
        F_metal = base + (1.0 - base) * sw
        F0t = tint * R0[:, None]
        F_tint = F0t + (1.0 - F0t) * sw
        F = ((1 - metallic) * (1 - spec_tint) * F_d)[:, None] + metallic[:, None] * F_metal + ((1 - metallic) * spec_tint)[:, None] * F_tint
        spec = F * (D * G / (4.0 * np.abs(cos_i)))[:, None]
    value += np.where((front & facing & (F_d > 0))[:, None], spec, 0.0)
    # ---- clearcoat: GTR1, Schlick with R0 = 0.04, GGX G with alpha 0.25
    with np.errstate(divide="ignore", invalid="ignore"):
        alpha = 0.1 + (0.001 - 0.1) * ccg
        a2 = alpha**2
        Dcc = (a2 - 1.0) / (np.pi * np.log(a2) * (1.0 + (a2 - 1.0) * wh[:, 2] ** 2))
        Dcc = np.where(Dcc * wh[:, 2] > 1e-20, Dcc, 0.0)

        def g1c(v):
            c2 = v[:, 2] ** 2
            r = 2.0 / (1.0 + np.sqrt(1.0 + 0.25**2 * (1.0 - c2) / c2))
            r = np.where(v[:, 2] == 1.0, 1.0, r)
            return np.where((v * wh).sum(1) * v[:, 2] <= 0, 0.0, r)

        Fcc = 0.04 + (1.0 - 0.04) * sw[:, 0]
        coat = cc * 0.25 * Fcc * Dcc * g1c(wi) * g1c(wo) * np.abs(cos_o)
    value += np.where((front & facing & (cc > 0))[:, None], coat[:, None], 0.0)
    # ---- diffuse with retro-reflection, flattened toward the Hanrahan-Krueger fake subsurface term
    brdf = (1 - metallic) * (1 - spec_trans)
    Fo, Fi = _schlick_w(np.abs(cos_o)), _schlick_w(np.abs(cos_i))
    f_diff = (1 - 0.5 * Fi) * (1 - 0.5 * Fo)
    Rr = 2.0 * rough * oh**2
    f_retro = Rr * (Fo + Fi + Fo * Fi * (Rr - 1.0))
    Fss90 = Rr / 2.0
    Fss = (1 + (Fss90 - 1) * Fo) * (1 + (Fss90 - 1) * Fi)
    with np.errstate(divide="ignore", invalid="ignore"):
        f_ss = 1.25 * (Fss * (1.0 / (np.abs(cos_o) + np.abs(cos_i)) - 0.5) + 0.5)
    dterm = (1 - flat) * (f_diff + f_retro) + flat * np.where(flat > 0, f_ss, 0.0)
    value += np.where((front & (brdf > 0))[:, None], (brdf * np.abs(cos_o) / np.pi * dterm)[:, None] * base, 0.0)
    # ---- sheen
    c_sheen = (1 - sheen_tint)[:, None] + sheen_tint[:, None] * tint
    sh = (sheen * (1 - metallic) * _schlick_w(np.abs(oh)) * np.abs(cos_o))[:, None] * c_sheen
    value += np.where((front & (sheen > 0) & (1 - metallic > 0))[:, None], sh, 0.0)
    return np.where((model != 0)[:, None], value, lambert)


def vertex_normals(verts, tri_idx, use=None):
    """Angle-weighted vertex normals, the way Mitsuba re-derives them after a vertex_positions update [EXT mesh.cpp
    recompute_vertex_normals; "Computing Vertex Normals from Polygonal Facets", Thuermer & Wuethrich 1998]: every face adds
    its unit normal, weighted by the interior angle at the corner, to each of its three vertices.  float64; the angle is
    arccos of the clipped cosine (the oracle uses Mitsuba's asin-based unit_angle).  `use` masks the faces that take part.
    -> [n_verts, 3] (zero rows for vertices no face touches)"""
    verts, tri_idx = np.asarray(verts, np.float64), np.asarray(tri_idx)
    p = verts[tri_idx]  # [F,3,3]
    fn = np.cross(p[:, 1] - p[:, 0], p[:, 2] - p[:, 0])
    fl = np.linalg.norm(fn, axis=1)
    good = fl > 0 if use is None else (fl > 0) & np.asarray(use, bool)
    fn = fn / np.where(fl > 0, fl, 1.0)[:, None]
    out = np.zeros_like(verts)
    for c in range(3):
        a, b = p[:, (c + 1) % 3] - p[:, c], p[:, (c + 2) % 3] - p[:, c]
        la, lb = np.linalg.norm(a, axis=1), np.linalg.norm(b, axis=1)
        ok = good & (la > 0) & (lb > 0)
        cosang = (a * b).sum(1) / np.where(ok, la * lb, 1.0)
        ang = np.arccos(np.clip(cosang, -1.0, 1.0))
        np.add.at(out, tri_idx[ok, c], fn[ok] * ang[ok, None])
    l = np.linalg.norm(out, axis=1)
    return out / np.where(l > 0, l, 1.0)[:, None]


def sample_texture(tex, u, v):
    """[h,w,3] texture at texture coordinates (u, v): repeat wrap, bilinear between texel centres, row 0 at v = 0
    [EXT Mitsuba `bitmap` defaults]"""
    tex = np.asarray(tex, np.float64)
    h, w = tex.shape[:2]
    fx, fy = (u % 1.0) * w - 0.5, (v % 1.0) * h - 0.5
    x0, y0 = np.floor(fx), np.floor(fy)
    ax, ay = (fx - x0)[:, None], (fy - y0)[:, None]
    x0, y0 = x0.astype(int), y0.astype(int)
    xa, xb, ya, yb = x0 % w, (x0 + 1) % w, y0 % h, (y0 + 1) % h
    return (1 - ay) * ((1 - ax) * tex[ya, xa] + ax * tex[ya, xb]) + ay * ((1 - ax) * tex[yb, xa] + ax * tex[yb, xb])


def _barycentric(verts, idx, P):
    """weights of P in the triangles idx [N,3] from sub-triangle areas (not from the intersection routine's u, v)"""
    a, b, c = verts[idx[:, 0]], verts[idx[:, 1]], verts[idx[:, 2]]
    nn = np.cross(b - a, c - a)
    den = (nn * nn).sum(1)
    with np.errstate(divide="ignore", invalid="ignore"):
        wa = (np.cross(c - b, P - b) * nn).sum(1) / den
        wb = (np.cross(a - c, P - c) * nn).sum(1) / den
    return wa, wb, 1.0 - wa - wb


def _shade_terms(verts, tri_idx, tri_shape, sd, spp, seed, mats, smooth=None, vert_uv=None, base_tex=None):
    """per sample: hit mask, shape id, projector taps / weights / rgb factor (irradiance scale x BSDF x cos), spot radiance.
    `smooth`: one flag per shape — shade in the frame of the interpolated vertex normal (Mitsuba meshes with vertex normals):
    BSDF and emitter cosines use the interpolated normal, faced to the viewer by its own cos(theta_i) (the `twosided`
    wrapper flips in the shading frame); the geometric normal keeps the shadow-ray offset and the requirement that the
    emitter lies on the viewer's geometric side."""
    mats = np.asarray(mats, np.float64)
    tris = world_triangles(verts, tri_idx)
    v0, e1, e2 = tris
    o, d, nt, ft = camera_rays(sd.cam, spp, True, seed)
    t, prim = intersect(o, d, tris, nt, ft)
    hit = prim >= 0
    pr = np.maximum(prim, 0)
    P = o + np.where(hit, t, 0.0)[:, None] * d
    n = np.cross(e1[pr], e2[pr])
    nl = np.linalg.norm(n, axis=1, keepdims=True)
    ok = hit & (nl[:, 0] > 0)
    n = n / np.where(nl > 0, nl, 1.0)
    n = np.where(((n * d).sum(1) > 0)[:, None], -n, n)
    Po = P + n * ((1.0 + np.abs(P).max(1)) * EPS)[:, None]
    shape = np.where(hit, np.asarray(tri_shape)[pr], -1)
    n_geo = n
    if smooth is not None and any(smooth):
        fl = np.asarray([bool(f) for f in smooth])
        vn = vertex_normals(verts, tri_idx, use=fl[np.asarray(tri_shape)])
        idx = np.asarray(tri_idx)[pr]
        wa, wb, wc = _barycentric(np.asarray(verts, np.float64), idx, P)
        ni = wa[:, None] * vn[idx[:, 0]] + wb[:, None] * vn[idx[:, 1]] + wc[:, None] * vn[idx[:, 2]]
        nil = np.linalg.norm(ni, axis=1)
        use = ok & fl[np.maximum(shape, 0)] & (nil > 0)
        ni = ni / np.where(nil > 0, nil, 1.0)[:, None]
        ni = np.where(((ni * d).sum(1) > 0)[:, None], -ni, ni)
        n = np.where(use[:, None], ni, n)
    out = {"hit": hit, "shape": shape, "pfac": np.zeros((len(P), 3)), "taps": None, "w": None, "spot": np.zeros((len(P), 3))}
    rows = mats[np.maximum(shape, 0)].copy()
    if base_tex is not None and rows.shape[1] == 16:
        # texture-valued base colour (Mitsuba: <mat>.brdf_0.base_color.data): rows[:, 15] = 1 + texture index; the hit's texture
        # coordinates are the vertices' interpolated with the barycentric weights
        idx = np.asarray(tri_idx)[pr]
        wa, wb, wc = _barycentric(np.asarray(verts, np.float64), idx, P)
        uvs = np.asarray(vert_uv, np.float64)
        uv = wa[:, None] * uvs[idx[:, 0]] + wb[:, None] * uvs[idx[:, 1]] + wc[:, None] * uvs[idx[:, 2]]
        for k, tex in enumerate(base_tex):
            sel = ok & (rows[:, 15] == k + 1)
            if sel.any():
                rows[sel, :3] = sample_texture(tex, uv[sel, 0], uv[sel, 1])
    if sd.proj.enabled:
        tw = _m(sd.proj.to_world, 4)
        w2l = np.linalg.inv(tw)
        pl = P @ w2l[:3, :3].T + w2l[:3, 3]
        Kp = _m(sd.proj.camera_to_sample, 4)
        q = np.concatenate([pl, np.ones((len(pl), 1))], 1) @ Kp.T
        with np.errstate(divide="ignore", invalid="ignore"):
            u, v = q[:, 0] / q[:, 3], q[:, 1] / q[:, 3]
        ppos, axis = tw[:3, 3], tw[:3, 2]
        wi = ppos - P
        dist = np.linalg.norm(wi, axis=1)
        wi = wi / np.where(dist > 0, dist, 1.0)[:, None]
        cos_s, cos_p = (n * wi).sum(1), -(wi @ axis)
        lit = ok & (pl[:, 2] > 0) & (u >= 0) & (u <= 1) & (v >= 0) & (v <= 1) & (cos_s > 0) & (cos_p > 0) & ((n_geo * wi).sum(1) > 0)
        if sd.shadows and lit.any():
            sel = np.where(lit)[0]
            occ = any_hit(np.broadcast_to(ppos, (len(sel), 3)), Po[sel] - ppos, tris, 1.0 - 10.0 * EPS)
            lit[sel[occ]] = False
        with np.errstate(divide="ignore", invalid="ignore"):
            fac = np.pi * sd.proj.scale / (pl[:, 2] ** 2 * cos_p)  # irradiance scale of the projector (its pi is Lambert's 1/pi)
            fac = fac[:, None] * bsdf_cos(rows, n, -d, wi)
        out["pfac"] = np.where(lit[:, None], fac, 0.0)
        out["taps"], out["w"] = _bilinear_setup(np.where(lit, u, 0.5), np.where(lit, v, 0.5), sd.proj.tex_w, sd.proj.tex_h)
    if sd.spot.enabled:
        tw = _m(sd.spot.to_world, 4)
        spos = tw[:3, 3]
        wi = spos - P
        d2 = (wi * wi).sum(1)
        wi = wi / np.sqrt(np.where(d2 > 0, d2, 1.0))[:, None]
        cos_s = (n * wi).sum(1)
        ll = (-wi) @ np.linalg.inv(tw)[:3, :3].T
        cos_t = ll[:, 2] / np.linalg.norm(ll, axis=1)
        cutoff, beam = np.deg2rad(sd.spot.cutoff_deg), np.deg2rad(sd.spot.beam_width_deg)
        ang = np.arccos(np.clip(cos_t, -1, 1))
        fall = np.where(ang <= beam, 1.0, np.where(ang < cutoff, (cutoff - ang) / (cutoff - beam), 0.0))
        lit = ok & (cos_s > 0) & (fall > 0) & ((n_geo * wi).sum(1) > 0)
        if sd.shadows and lit.any():
            sel = np.where(lit)[0]
            occ = any_hit(np.broadcast_to(spos, (len(sel), 3)), Po[sel] - spos, tris, 1.0 - 10.0 * EPS)
            lit[sel[occ]] = False
        with np.errstate(divide="ignore", invalid="ignore"):
            f = np.where(lit[:, None], (fall / d2)[:, None] * bsdf_cos(rows, n, -d, wi), 0.0)
        out["spot"] = f * np.asarray(list(sd.spot.intensity), np.float64)[None]
    return out


def _gaussian(x, stddev):
    """Mitsuba's `gaussian` reconstruction filter [EXT src/rfilters/gaussian.cpp]: radius 4 stddev, shifted so that it ends at zero"""
    alpha, r = -1.0 / (2.0 * stddev * stddev), 4.0 * stddev
    return np.maximum(0.0, np.exp(alpha * x * x) - np.exp(alpha * r * r))


def _film_positions(W, H, spp, seed):
    idx = np.arange(W * H * spp, dtype=np.uint64)
    pix = idx // spp
    jx, jy = jitter(seed, idx)
    return (pix % W).astype(np.int64), (pix // W).astype(np.int64), (pix % W) + jx, (pix // W) + jy  # pixel, film position in pixels


def _film_splat(W, H, spp, seed, stddev, values=None, gather=None):
    """the film of ImageBlock::put [EXT src/render/imageblock.cpp], formed from ABSOLUTE positions (sample position vs pixel centres):
    values [N,k]: -> (sum w values [H,W,k], sum w [H,W]);  gather [H,W,k]: -> per sample sum_pixels w gather[pixel] (the transpose)"""
    x, y, fx, fy = _film_positions(W, H, spp, seed)
    r = int(np.ceil(4.0 * stddev))
    num = np.zeros((H, W, values.shape[1])) if values is not None else None
    den = np.zeros((H, W))
    back = np.zeros((len(x), gather.shape[2])) if gather is not None else None
    for dy in range(-r, r + 1):
        for dx in range(-r, r + 1):
            tx, ty = x + dx, y + dy
            ok = (tx >= 0) & (tx < W) & (ty >= 0) & (ty < H)
            w = _gaussian((tx + 0.5) - fx, stddev) * _gaussian((ty + 0.5) - fy, stddev)
            w = np.where(ok, w, 0.0)
            txc, tyc = np.clip(tx, 0, W - 1), np.clip(ty, 0, H - 1)
            np.add.at(den, (tyc, txc), w)
            if num is not None:
                np.add.at(num, (tyc, txc), w[:, None] * values)
            if back is not None:
                back += w[:, None] * gather[tyc, txc]
    return num, den, back


def render_fwd(verts, tri_idx, tri_shape, sd, albedo, tex, spp, seed, smooth=None, vert_uv=None, base_tex=None, gaussian_stddev=None):
    s = _shade_terms(verts, tri_idx, tri_shape, sd, spp, seed, albedo, smooth, vert_uv, base_tex)
    W, H = sd.cam.width, sd.cam.height
    rad = s["spot"].copy()
    if sd.proj.enabled:
        tex = np.asarray(tex, np.float64)
        (x0, x1, y0, y1), (w00, w01, w10, w11) = s["taps"], s["w"]
        if tex.ndim == 2 or tex.shape[-1] == 1:
            t2 = tex.reshape(tex.shape[0], tex.shape[1])
            tv = w00 * t2[y0, x0] + w01 * t2[y0, x1] + w10 * t2[y1, x0] + w11 * t2[y1, x1]
            rad += tv[:, None] * s["pfac"] * np.asarray(list(sd.proj.color), np.float64)[None]
        else:
            tv = w00[:, None] * tex[y0, x0] + w01[:, None] * tex[y0, x1] + w10[:, None] * tex[y1, x0] + w11[:, None] * tex[y1, x1]
            rad += tv * s["pfac"]
    rad = np.where(s["hit"][:, None], rad, 0.0)
    if gaussian_stddev is not None:  # hdrfilm's default filter: weighted sum over weight [EXT src/films/hdrfilm.cpp develop()]
        num, den, _ = _film_splat(W, H, spp, seed, gaussian_stddev, values=rad)
        with np.errstate(divide="ignore", invalid="ignore"):
            return np.where(den[..., None] > 0, num / den[..., None], 0.0)
    return rad.reshape(H, W, spp, 3).mean(2)


def render_bwd(verts, tri_idx, tri_shape, sd, albedo, spp, seed, gimg, smooth=None, vert_uv=None, base_tex=None, gaussian_stddev=None):
    """d <img, gimg> / d tex for a 1-channel texture"""
    s = _shade_terms(verts, tri_idx, tri_shape, sd, spp, seed, albedo, smooth, vert_uv, base_tex)
    if gaussian_stddev is not None:
        W, H = sd.cam.width, sd.cam.height
        _, den, _ = _film_splat(W, H, spp, seed, gaussian_stddev)
        with np.errstate(divide="ignore", invalid="ignore"):
            G = np.where(den[..., None] > 0, np.asarray(gimg, np.float64).reshape(H, W, 3) / den[..., None], 0.0)
        _, _, g = _film_splat(W, H, spp, seed, gaussian_stddev, gather=G)
        ws = (g * s["pfac"] * np.asarray(list(sd.proj.color), np.float64)[None]).sum(1)
    else:
        g = np.repeat(np.asarray(gimg, np.float64).reshape(-1, 3), spp, axis=0)
        ws = (g * s["pfac"] * np.asarray(list(sd.proj.color), np.float64)[None]).sum(1) / spp
    ws = np.where(s["hit"], ws, 0.0)
    gt = np.zeros((sd.proj.tex_h, sd.proj.tex_w))
    (x0, x1, y0, y1), (w00, w01, w10, w11) = s["taps"], s["w"]
    for (yy, xx), w in (((y0, x0), w00), ((y0, x1), w01), ((y1, x0), w10), ((y1, x1), w11)):
        np.add.at(gt, (yy, xx), ws * w)
    return gt
