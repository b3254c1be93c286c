"""(CPU, numpy) prototype of the tile ENVELOPE proof for an emitter's any-hit stage: per tile of the emitter's grid one plane in front of
everything the tile lists; a shadow segment whose counted part ends in front of that plane cannot be occluded.  Prints the share of pixel
packets (all samples of a pixel) the proof settles on a random pose of the vocal fold.   python tools/envproto.py [n_tiles] [spp]"""
import sys

import numpy as np

sys.path.insert(0, ".")
from fireflies_amd import scenes  # noqa: E402
from oracle import oracle as orc  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 80
SPP = int(sys.argv[2]) if len(sys.argv) > 2 else 8
W = H = 512
RAY_EPS = 1500 * 2.0 ** -24
SHADOW_EPS = 10 * RAY_EPS


def rot_y(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, 0, s, 0], [0, 1, 0, 0], [-s, 0, c, 0], [0, 0, 0, 1.0]])


def pose(data, rng):
    out = []
    for m, (sx, ry) in zip(data.meshes, [(rng.uniform(0.8, 1.2), rng.uniform(-0.1, 0.1)), (rng.uniform(0.5, 2.0), rng.uniform(-0.25, 0.25))]):
        fr = m.frames[rng.integers(0, m.frames.shape[0])].astype(np.float64)
        c = fr.mean(0)
        M = np.eye(4)
        M[:3, 3] = c
        M = M @ rot_y(ry) @ np.diag([sx, 1, 1, 1.0])
        Mi = np.eye(4)
        Mi[:3, 3] = -c
        M = M @ Mi
        v = fr @ M[:3, :3].T + M[:3, 3]
        out.append(v[m.tris])
    return np.concatenate(out, 0)  # [F,3,3]


def main():
    rng = np.random.default_rng(int(sys.argv[3]) if len(sys.argv) > 3 else 0)
    data = scenes.vocalfold(width=W, height=H)
    tri = pose(data, rng)
    F = tri.shape[0]
    E = data.spot.to_world[:3, 3].astype(np.float64)
    w2l = np.linalg.inv(data.spot.to_world.astype(np.float64))
    tanc = np.tan(np.deg2rad(data.spot.cutoff_angle + 1.0))
    M = np.stack([0.5 * N * (w2l[0, :3] / tanc + w2l[2, :3]), 0.5 * N * (w2l[1, :3] / tanc + w2l[2, :3]), w2l[2, :3]])
    Minv = np.linalg.inv(M)
    # ---- per triangle: projection, plane n_j (n_j . (X - E) = 1), tile-space coefficients m_j
    rel = tri - E
    pz = rel @ M[2]
    px, py = (rel @ M[0]) / pz, (rel @ M[1]) / pz
    safe = (pz > 0).all(1)
    nn = np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0])
    den = (nn * rel[:, 0]).sum(1)
    nj = nn / den[:, None]
    mj = nj @ Minv  # (Minv^T n_j): w_j(q) = mj . (qx, qy, 1)
    PAD = 1 / 256
    bx0, bx1, by0, by1 = px.min(1) - PAD, px.max(1) + PAD, py.min(1) - PAD, py.max(1) + PAD
    # ---- per tile: lists by box overlap (a superset of the product's lists), gradient = mean, offset = max residual over clipped corners
    cnt = np.zeros((N, N))
    gsum = np.zeros((N, N, 2))
    bad = np.zeros((N, N), bool)
    ents = []
    for j in range(F):
        if not safe[j]:
            if (pz[j] > 0).any():
                X, Y, Zz = rel[j] @ M[0], rel[j] @ M[1], pz[j]
                if not ((X < 0).all() or (N * Zz - X < 0).all() or (Y < 0).all() or (N * Zz - Y < 0).all()):
                    bad[:] = True  # (prototype: an unsafe triangle inside the frustum spoils every tile)
                    print("unsafe triangle inside the frustum", j)
            continue
        if bx1[j] < 0 or by1[j] < 0 or bx0[j] >= N or by0[j] >= N:
            continue
        for ty in range(max(0, int(np.floor(by0[j]))), min(N - 1, int(np.floor(by1[j]))) + 1):
            for tx in range(max(0, int(np.floor(bx0[j]))), min(N - 1, int(np.floor(bx1[j]))) + 1):
                ents.append((ty, tx, j))
                cnt[ty, tx] += 1
                gsum[ty, tx] += mj[j, :2]
    grad = gsum / np.maximum(cnt, 1)[..., None]
    TWO = len(sys.argv) > 4 and sys.argv[4] == "2"
    # principal axis of the gradients' deviations per tile (2x2 covariance)
    cov = np.zeros((N, N, 3))
    for ty, tx, j in ents:
        dx, dy = mj[j, 0] - grad[ty, tx, 0], mj[j, 1] - grad[ty, tx, 1]
        cov[ty, tx] += (dx * dx, dx * dy, dy * dy)
    ang = 0.5 * np.arctan2(2 * cov[..., 1], cov[..., 0] - cov[..., 2])
    ax = np.stack([np.cos(ang), np.sin(ang)], -1)
    grp = {}
    g2sum = np.zeros((N, N, 2, 2)); g2cnt = np.zeros((N, N, 2))
    for ty, tx, j in ents:
        g = int(((mj[j, :2] - grad[ty, tx]) * ax[ty, tx]).sum() > 0) if TWO else 0
        grp[(ty, tx, j)] = g
        g2sum[ty, tx, g] += mj[j, :2]; g2cnt[ty, tx, g] += 1
    grad2 = g2sum / np.maximum(g2cnt, 1)[..., None]
    cmax2 = np.full((N, N, 2), -np.inf)
    call2 = np.full((N, N, 2), -np.inf)  # the same two gradients, each plane in front of ALL entries (min of the two: convex kinks)
    call1 = np.full((N, N), -np.inf)
    for ty, tx, j in ents:
        x0, x1, y0, y1 = max(bx0[j], tx), min(bx1[j], tx + 1), max(by0[j], ty), min(by1[j], ty + 1)
        for g in (0, 1):
            a, b = mj[j, 0] - grad2[ty, tx, g, 0], mj[j, 1] - grad2[ty, tx, g, 1]
            call2[ty, tx, g] = max(call2[ty, tx, g], mj[j, 2] + max(a * x0, a * x1) + max(b * y0, b * y1))
        a, b = mj[j, 0] - grad[ty, tx, 0], mj[j, 1] - grad[ty, tx, 1]
        call1[ty, tx] = max(call1[ty, tx], mj[j, 2] + max(a * x0, a * x1) + max(b * y0, b * y1))
    for ty, tx, j in ents:
        g = grp[(ty, tx, j)]
        x0, x1, y0, y1 = max(bx0[j], tx), min(bx1[j], tx + 1), max(by0[j], ty), min(by1[j], ty + 1)
        a, b = mj[j, 0] - grad2[ty, tx, g, 0], mj[j, 1] - grad2[ty, tx, g, 1]
        r = mj[j, 2] + max(a * x0, a * x1) + max(b * y0, b * y1)
        cmax2[ty, tx, g] = max(cmax2[ty, tx, g], r)
    # shared gradient per tile + an offset per SUB x SUB sub-cell
    SUB = int(sys.argv[6]) if len(sys.argv) > 6 else 4
    csub = np.full((N, N, SUB, SUB), -np.inf)
    for ty, tx, j in ents:
        a, b = mj[j, 0] - grad[ty, tx, 0], mj[j, 1] - grad[ty, tx, 1]
        for sy_ in range(SUB):
            y0, y1 = max(by0[j], ty + sy_ / SUB), min(by1[j], ty + (sy_ + 1) / SUB)
            if y0 > y1:
                continue
            for sx_ in range(SUB):
                x0, x1 = max(bx0[j], tx + sx_ / SUB), min(bx1[j], tx + (sx_ + 1) / SUB)
                if x0 > x1:
                    continue
                r = mj[j, 2] + max(a * x0, a * x1) + max(b * y0, b * y1)
                csub[ty, tx, sy_, sx_] = max(csub[ty, tx, sy_, sx_], r)
    # bilinear envelope on the (N SUB + 1)^2 vertices of the sub-cell grid: W[v] = max over the entries that overlap a cell next to v of plane_j(v)
    NV = N * SUB + 1
    Wv = np.full((NV, NV), -np.inf)
    seen = set()
    for ty, tx, j in ents:
        if j in seen:
            continue
        seen.add(j)
        cx0, cx1 = max(0, int(np.floor(bx0[j] * SUB))), min(N * SUB - 1, int(np.floor(bx1[j] * SUB)))
        cy0, cy1 = max(0, int(np.floor(by0[j] * SUB))), min(N * SUB - 1, int(np.floor(by1[j] * SUB)))
        vx = np.arange(cx0, cx1 + 2); vy = np.arange(cy0, cy1 + 2)
        val = mj[j, 0] * (vx[None, :] / SUB) + mj[j, 1] * (vy[:, None] / SUB) + mj[j, 2]
        Wv[cy0:cy1 + 2, cx0:cx1 + 2] = np.maximum(Wv[cy0:cy1 + 2, cx0:cx1 + 2], val)
    # per-cell planes: corner maxima over the entries that overlap THAT cell, then a plane in front of the bilinear patch
    NC = N * SUB
    Wc = np.full((NC, NC, 4), -np.inf)
    poison = np.zeros((NC, NC), bool)
    nlen = np.linalg.norm(nj, axis=1) * 1.35
    h = 1.0 / SUB
    for j in seen:
        cx0, cx1 = max(0, int(np.floor((bx0[j] - 2e-4) * SUB))), min(NC - 1, int(np.floor((bx1[j] + 2e-4) * SUB)))
        cy0, cy1 = max(0, int(np.floor((by0[j] - 2e-4) * SUB))), min(NC - 1, int(np.floor((by1[j] + 2e-4) * SUB)))
        X0 = np.arange(cx0, cx1 + 1)[None, :] * h; Y0 = np.arange(cy0, cy1 + 1)[:, None] * h
        p00 = mj[j, 0] * X0 + mj[j, 1] * Y0 + mj[j, 2]
        p = np.stack([p00, p00 + mj[j, 0] * h, p00 + mj[j, 1] * h, p00 + (mj[j, 0] + mj[j, 1]) * h], -1)
        Wc[cy0:cy1 + 1, cx0:cx1 + 1] = np.maximum(Wc[cy0:cy1 + 1, cx0:cx1 + 1], p)
        poison[cy0:cy1 + 1, cx0:cx1 + 1] |= ~(p.max(-1) * 40 >= nlen[j])
    axc = 0.5 * ((Wc[..., 1] - Wc[..., 0]) + (Wc[..., 3] - Wc[..., 2])) / h
    byc = 0.5 * ((Wc[..., 2] - Wc[..., 0]) + (Wc[..., 3] - Wc[..., 1])) / h
    with np.errstate(invalid="ignore"):
        c0 = np.maximum(np.maximum(Wc[..., 0], Wc[..., 1] - axc * h), np.maximum(Wc[..., 2] - byc * h, Wc[..., 3] - (axc + byc) * h))
    empty = ~np.isfinite(Wc[..., 0])
    print(f"cells {NC}x{NC}: empty {empty.mean():.3f}, poisoned {poison.mean():.4f}")
    print(f"grid {N}x{N}: {len(ents)} entries, {int((cnt > 0).sum())} tiles in use, mean list {cnt[cnt > 0].mean():.1f}")
    # ---- camera samples, primary hits (oracle), lifted points, the proof per sample and per pixel
    pool, tris, shape, off, stride, nfr, alb = scenes.flatten(data)
    verts = tri.reshape(-1, 3).astype(np.float32)
    go = orc.Geometry(verts, np.arange(3 * F, dtype=np.int32).reshape(F, 3), np.zeros(F, np.int32), np.zeros(1, np.int32))
    K = scenes.perspective_projection(W, H, data.camera.fov_x, data.camera.near, data.camera.far).astype(np.float64)
    Ki = np.linalg.inv(K)
    yy, xx = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    ok_pix = np.ones((H, W), bool)
    need_pix = np.zeros((H, W), bool)
    fails = np.zeros((H, W), int)
    C = data.camera.to_world[:3, 3].astype(np.float64)
    R = data.camera.to_world[:3, :3].astype(np.float64)
    n_s = n_ok = 0
    for s in range(SPP):
        jx, jy = rng.random((H, W)), rng.random((H, W))
        sx, sy = (xx + jx) / W, (yy + jy) / H
        npnt = np.stack([sx, sy, np.zeros_like(sx), np.ones_like(sx)], -1) @ Ki.T
        dl = npnt[..., :3] / npnt[..., 3:]
        dl /= np.linalg.norm(dl, axis=-1, keepdims=True)
        d = dl @ R.T
        o = np.broadcast_to(C, d.shape)
        t, sh, prim = go.trace_rays(o.reshape(-1, 3), d.reshape(-1, 3))
        hit = (prim >= 0).reshape(H, W)
        t = t.reshape(H, W).astype(np.float64)
        P = C + t[..., None] * d
        pr = np.maximum(prim.reshape(H, W), 0)
        ng = nn[pr] / np.linalg.norm(nn[pr], axis=-1, keepdims=True)
        flip = (ng * d).sum(-1) > 0
        ng[flip] *= -1
        offl = (1 + np.abs(P).max(-1)) * RAY_EPS
        Po = P + offl[..., None] * ng
        sd = Po - E
        cos_s = (ng * (-sd)).sum(-1)
        Z = sd @ M[2]
        fx, fy = (sd @ M[0]) / Z, (sd @ M[1]) / Z
        inside = (Z > 0) & (fx >= 0) & (fy >= 0) & (fx < N) & (fy < N)
        # spot cone (cutoff): only samples the spot lights need the walk
        ll = sd @ w2l[:3, :3].T
        cos_t = ll[..., 2] / np.linalg.norm(ll, axis=-1)
        need = hit & (cos_s > 0) & (cos_t > np.cos(np.deg2rad(data.spot.cutoff_angle)))
        tx, ty = np.clip(fx.astype(int), 0, N - 1), np.clip(fy.astype(int), 0, N - 1)
        e0 = grad2[ty, tx, 0, 0] * fx + grad2[ty, tx, 0, 1] * fy + cmax2[ty, tx, 0]
        e1 = grad2[ty, tx, 1, 0] * fx + grad2[ty, tx, 1, 1] * fy + cmax2[ty, tx, 1]
        env = Z * np.maximum(e0, e1)  # >= 1 / t_j for every entry of the tile (an empty group: -inf)
        MODE = sys.argv[5] if len(sys.argv) > 5 else "max"
        q0 = grad2[ty, tx, 0, 0] * fx + grad2[ty, tx, 0, 1] * fy + call2[ty, tx, 0]
        q1 = grad2[ty, tx, 1, 0] * fx + grad2[ty, tx, 1, 1] * fy + call2[ty, tx, 1]
        both = (g2cnt[ty, tx, 0] > 0) & (g2cnt[ty, tx, 1] > 0)
        envmin = Z * np.where(both, np.minimum(q0, q1), np.where(g2cnt[ty, tx, 0] > 0, q0, q1))
        env1 = Z * (grad[ty, tx, 0] * fx + grad[ty, tx, 1] * fy + call1[ty, tx])
        if MODE == "min":
            env = envmin
        elif MODE == "all":
            env = np.minimum(np.minimum(env, envmin), env1)
        elif MODE == "one":
            env = env1
        elif MODE == "bil":
            gx, gy = fx * SUB, fy * SUB
            ix, iy = np.clip(gx.astype(int), 0, N * SUB - 1), np.clip(gy.astype(int), 0, N * SUB - 1)
            ax_, ay_ = gx - ix, gy - iy
            w00, w10, w01, w11 = Wv[iy, ix], Wv[iy, ix + 1], Wv[iy + 1, ix], Wv[iy + 1, ix + 1]
            wq = (w00 * (1 - ax_) + w10 * ax_) * (1 - ay_) + (w01 * (1 - ax_) + w11 * ax_) * ay_
            wq = np.where(np.isfinite(w00) & np.isfinite(w10) & np.isfinite(w01) & np.isfinite(w11), wq, -np.inf)
            env = Z * wq
        elif MODE == "cell":
            gx, gy = fx * SUB, fy * SUB
            ix, iy = np.clip(gx.astype(int), 0, NC - 1), np.clip(gy.astype(int), 0, NC - 1)
            wq = c0[iy, ix] + axc[iy, ix] * (fx - ix * h) + byc[iy, ix] * (fy - iy * h)
            wq = np.where(empty[iy, ix], -np.inf, np.where(poison[iy, ix], np.inf, wq))
            env = Z * wq
        elif MODE == "sub":
            sxi = np.clip(((fx - tx) * SUB).astype(int), 0, SUB - 1); syi = np.clip(((fy - ty) * SUB).astype(int), 0, SUB - 1)
            env = Z * (grad[ty, tx, 0] * fx + grad[ty, tx, 1] * fy + csub[ty, tx, syi, sxi])
        proven = inside & ~bad[ty, tx] & (env <= (1.0 / (1.0 - SHADOW_EPS)) * (1 - 2e-5))
        n_s += int(need.sum())
        n_ok += int((need & proven).sum())
        ok_pix &= ~need | proven
        need_pix |= need
        fails += (need & ~proven)
    print(f"samples that need the spot: {n_s}, proven {n_ok} = {n_ok / max(n_s, 1):.3f}")
    print(f"pixels with such samples: {int(need_pix.sum())}, all {SPP} samples proven: {int((need_pix & ok_pix).sum())} = {(need_pix & ok_pix).sum() / max(need_pix.sum(), 1):.3f}")
    # where it fails: a coarse map
    blk = (need_pix & ~ok_pix).reshape(16, 32, 16, 32).mean((1, 3))
    print("share of unproven pixels per 32x32 block:")
    for r in blk:
        print(" ".join(f"{int(99 * v):2d}" for v in r))


if __name__ == "__main__":
    main()
