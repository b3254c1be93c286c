// ffx_bins.hip — tile bins: the binning stage in front of the packet render kernels (gfx950).
//
// No reference counterpart: Mitsuba builds and walks its own acceleration structure inside params.update() / mi.render
// (/root/reference/fireflies/scene.py:384, examples/vocalfold_scene.py:102) [EXT].  Here the three ray origins of a render — the
// camera, the projector and the spot light (rays towards an emitter are traced FROM it) — each get a perspective grid
// of tiles over their field of view, and this pre-pass lists, per tile, the triangles of the current pose whose projection touches
// it (ffx_common.h: BinEntry, BinGrid).  A pixel's packet of k_render_fwd_pk then tests the entries of its own tile against the
// packet's screen rectangle — lanes on the entries — and runs the exact ray / triangle test on the survivors: what were 4.2 + 3.9
// dependent steps through the 64-wide tree per pixel (closest-hit + any-hit walk) is one coalesced load and ~1.3 steps.
//
// Three launches on the stream of the apex records (normally the side stream behind the re-fit, off the renders' critical path):
//   k_bin<false>  a lane per triangle and apex: project, classify, count the tiles it touches (one atomic per tile)
//   k_bin_scan    a workgroup per apex: exclusive scan of the counts -> list starts; overflow check against the capacity
//   k_bin<true>   the same walk again, writing the 64-byte entries at start[tile] + cursor[tile]++
// Triangles that touch more than four tiles (slivers at grazing angles, large faces) and triangles the projection is not trusted
// for (a vertex behind / beside the apex) are handled by the whole wave, lanes on tiles.  Entry order inside a tile is the order
// of the atomics — it does not affect a result: closest hits carry the primitive-id tie-break, any-hit answers are booleans.
#include "ffx_common.h"

typedef unsigned long long wmask_t;

__device__ __forceinline__ void bin_make_entry(float x0, float y0, float x1, float y1, float x2, float y2, int slot, bool unsafe, BinEntry &en) {
  en.slot = slot;
  en.pad[0] = en.pad[1] = 0u;
  if (unsafe) {
    en.bb[0] = en.bb[1] = -INFINITY;
    en.bb[2] = en.bb[3] = INFINITY;
#pragma unroll
    for (int k = 0; k < 3; ++k) { en.e[3 * k] = 0.f; en.e[3 * k + 1] = 0.f; en.e[3 * k + 2] = 1.f; }
    return;
  }
  en.bb[0] = fminf(x0, fminf(x1, x2)) - FFX_BIN_PAD;
  en.bb[1] = fminf(y0, fminf(y1, y2)) - FFX_BIN_PAD;
  en.bb[2] = fmaxf(x0, fmaxf(x1, x2)) + FFX_BIN_PAD;
  en.bb[3] = fmaxf(y0, fmaxf(y1, y2)) + FFX_BIN_PAD;
  const float ax = x1 - x0, ay = y1 - y0, bx = x2 - x1, by = y2 - y1, cx = x0 - x2, cy = y0 - y2;
  const float area2 = ax * (y2 - y0) - ay * (x2 - x0);
  // the orientation of a sliver is not trusted (rounding of area2 ~ 1e-7 of the products): such a projection keeps its box only
  const bool degenerate = !(fabsf(area2) > 1e-5f * ((fabsf(ax) + fabsf(ay)) * (fabsf(cx) + fabsf(cy)))) || !(fabsf(area2) < 3.0e38f);
  const float s = area2 >= 0.f ? 1.0f : -1.0f;
  const float px[3] = {x0, x1, x2}, py[3] = {y0, y1, y2}, ex[3] = {ax, bx, cx}, ey[3] = {ay, by, cy};
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    // inside the edge P -> Q:  s * (e.x (y - P.y) - e.y (x - P.x)) >= 0   =   n.x x + n.y y + c >= 0
    const float nx = -s * ey[k], ny = s * ex[k];
    const float c = -(nx * px[k] + ny * py[k]) + FFX_BIN_PAD * (fabsf(nx) + fabsf(ny));
    en.e[3 * k] = degenerate ? 0.f : nx;
    en.e[3 * k + 1] = degenerate ? 0.f : ny;
    en.e[3 * k + 2] = degenerate ? 1.f : c;
  }
}

// does the (padded) projection of an entry touch the rectangle [rx0, rx1] x [ry0, ry1]?  Box first, then the rectangle's most-inside
// corner against each edge (the separating-axis test of a convex polygon against a box, edge normals only — conservative)
__device__ __forceinline__ bool bin_entry_touches(const BinEntry &en, float rx0, float ry0, float rx1, float ry1) {
  bool ok = en.bb[0] <= rx1 && en.bb[2] >= rx0 && en.bb[1] <= ry1 && en.bb[3] >= ry0;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float nx = en.e[3 * k], ny = en.e[3 * k + 1], c = en.e[3 * k + 2];
    ok = ok && fmaf(nx, nx >= 0.f ? rx1 : rx0, fmaf(ny, ny >= 0.f ? ry1 : ry0, c)) >= 0.f;
  }
  return ok;
}

__global__ void __launch_bounds__(256) k_bin_clear(BinBuild bb) {
  const int a = blockIdx.y;
  if (!bb.g[a].on) return;
  char *base = bb.base[a];
  const int i = blockIdx.x * 256 + threadIdx.x;
  uint32_t *starts = (uint32_t *)(base + ffx_bin_off_starts());
  uint32_t *cursors = (uint32_t *)(base + ffx_bin_off_cursors());
  const int nt = bb.g[a].nx * bb.g[a].ny;
  if (i <= nt) starts[i] = 0u;
  if (i < nt) cursors[i] = 0u;
  if (i == 0) { BinHdr *h = (BinHdr *)base; h->ok = 0u; h->total = 0u; h->cap = bb.cap; }
}

template <bool FILL>
__global__ void __launch_bounds__(256) k_bin(const TriRec *__restrict__ recs, int n_tris, BinBuild bb) {
  const int a = blockIdx.y;
  if (!bb.g[a].on) return;
  char *base = bb.base[a];
  const BinHdr *hdr = (const BinHdr *)base;
  if (FILL && hdr->ok == 0u) return; // the lists do not fit: this pose renders through the tree walks
  uint32_t *starts = (uint32_t *)(base + ffx_bin_off_starts());
  uint32_t *cursors = (uint32_t *)(base + ffx_bin_off_cursors());
  BinEntry *ents = (BinEntry *)(base + ffx_bin_off_entries());
  const int nx = bb.g[a].nx, ny = bb.g[a].ny;
  const int k = blockIdx.x * 256 + threadIdx.x;
  const int lane = threadIdx.x & 63;
  auto emit = [&](int tile, const BinEntry &en) {
    if (!FILL) atomicAdd(&starts[tile], 1u);
    else {
      const uint32_t i = atomicAdd(&cursors[tile], 1u);
      ents[starts[tile] + i] = en;
    }
  };
  float X[3] = {0.f, 0.f, 0.f}, Y[3] = {0.f, 0.f, 0.f}, Z[3] = {-1.f, -1.f, -1.f};
  if (k < n_tris) {
    const float4 *r4 = reinterpret_cast<const float4 *>(recs + k);
    const float4 ra = r4[0], rb = r4[1], rc = r4[2];
    const float vx[3] = {ra.x, ra.x + ra.w, ra.x + rb.z}, vy[3] = {ra.y, ra.y + rb.x, ra.y + rb.w}, vz[3] = {ra.z, ra.z + rb.y, ra.z + rc.x};
    const float *M = bb.g[a].M;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const float dx = vx[i] - bb.g[a].o[0], dy = vy[i] - bb.g[a].o[1], dz = vz[i] - bb.g[a].o[2];
      X[i] = fmaf(M[0], dx, fmaf(M[1], dy, M[2] * dz));
      Y[i] = fmaf(M[3], dx, fmaf(M[4], dy, M[5] * dz));
      Z[i] = fmaf(M[6], dx, fmaf(M[7], dy, M[8] * dz));
    }
  }
  // class 0: wholly behind the apex plane (no ray of the grid reaches it), 1: touches at most four tiles, 2: more, 3: the projection
  // is not trusted (a vertex behind / beside the apex, or far outside the grid): listed by plane tests
  int cls = 0;
  float x[3] = {0.f, 0.f, 0.f}, y[3] = {0.f, 0.f, 0.f};
  int tx0 = 0, tx1 = -1, ty0 = 0, ty1 = -1;
  if (Z[0] > 0.f || Z[1] > 0.f || Z[2] > 0.f) {
    bool safe = true;
#pragma unroll
    for (int i = 0; i < 3; ++i) safe = safe && Z[i] > 0.f && fabsf(X[i]) <= FFX_BIN_FAR * Z[i] && fabsf(Y[i]) <= FFX_BIN_FAR * Z[i];
    if (safe) {
#pragma unroll
      for (int i = 0; i < 3; ++i) { x[i] = X[i] / Z[i]; y[i] = Y[i] / Z[i]; }
      const float mnx = fminf(x[0], fminf(x[1], x[2])) - FFX_BIN_PAD, mxx = fmaxf(x[0], fmaxf(x[1], x[2])) + FFX_BIN_PAD;
      const float mny = fminf(y[0], fminf(y[1], y[2])) - FFX_BIN_PAD, mxy = fmaxf(y[0], fmaxf(y[1], y[2])) + FFX_BIN_PAD;
      if (mxx >= 0.f && mxy >= 0.f && mnx < (float)nx && mny < (float)ny) {
        tx0 = max(0, (int)floorf(mnx)); tx1 = min(nx - 1, (int)floorf(mxx));
        ty0 = max(0, (int)floorf(mny)); ty1 = min(ny - 1, (int)floorf(mxy));
        if (tx0 <= tx1 && ty0 <= ty1) cls = (tx1 - tx0 + 1) * (ty1 - ty0 + 1) <= 4 ? 1 : 2;
      }
    } else {
      cls = 3;
    }
  }
  if (cls == 1) {
    BinEntry en;
    bin_make_entry(x[0], y[0], x[1], y[1], x[2], y[2], k, false, en);
    for (int ty = ty0; ty <= ty1; ++ty)
      for (int tx = tx0; tx <= tx1; ++tx)
        if (bin_entry_touches(en, (float)tx, (float)ty, (float)(tx + 1), (float)(ty + 1))) emit(ty * nx + tx, en);
  }
  // ---- the whole wave on one triangle at a time: lanes on tiles
  wmask_t big = __ballot(cls >= 2);
  while (big != 0ull) {
    const int j = __builtin_ctzll(big);
    big &= big - 1ull;
    const int jc = __shfl(cls, j, 64), js = __shfl(k, j, 64);
    BinEntry en;
    if (jc == 2) {
      const float q0 = __shfl(x[0], j, 64), q1 = __shfl(y[0], j, 64), q2 = __shfl(x[1], j, 64), q3 = __shfl(y[1], j, 64), q4 = __shfl(x[2], j, 64), q5 = __shfl(y[2], j, 64);
      const int bx0 = __shfl(tx0, j, 64), bx1 = __shfl(tx1, j, 64), by0 = __shfl(ty0, j, 64), by1 = __shfl(ty1, j, 64);
      bin_make_entry(q0, q1, q2, q3, q4, q5, js, false, en);
      const int w = bx1 - bx0 + 1, n = w * (by1 - by0 + 1);
      for (int t = lane; t < n; t += 64) {
        const int ty = by0 + t / w, tx = bx0 + t % w;
        if (bin_entry_touches(en, (float)tx, (float)ty, (float)(tx + 1), (float)(ty + 1))) emit(ty * nx + tx, en);
      }
    } else {
      float PX[3], PY[3], PZ[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) { PX[i] = __shfl(X[i], j, 64); PY[i] = __shfl(Y[i], j, 64); PZ[i] = __shfl(Z[i], j, 64); }
      bin_make_entry(0.f, 0.f, 0.f, 0.f, 0.f, 0.f, js, true, en);
      // a column (row) of tiles is a wedge between two planes through the apex: the triangle misses it iff all three vertices lie
      // beyond the same plane (any vertex position, also behind the apex).  Columns and rows separate: the tiles are their product.
      wmask_t colm[2] = {0ull, 0ull}, rowm[2] = {0ull, 0ull};
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int c = lane + 64 * h;
        const float lo = (float)c - FFX_BIN_PAD, hi = (float)(c + 1) + FFX_BIN_PAD;
        bool outL = true, outR = true, outT = true, outB = true;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          outL = outL && (PX[i] - lo * PZ[i] < 0.f);
          outR = outR && (hi * PZ[i] - PX[i] < 0.f);
          outT = outT && (PY[i] - lo * PZ[i] < 0.f);
          outB = outB && (hi * PZ[i] - PY[i] < 0.f);
        }
        colm[h] = __ballot(c < nx && !outL && !outR);
        rowm[h] = __ballot(c < ny && !outT && !outB);
      }
      if ((colm[0] | colm[1]) != 0ull && (rowm[0] | rowm[1]) != 0ull) {
        for (int r = 0; r < ny; ++r) {
          if (!((rowm[r >> 6] >> (r & 63)) & 1ull)) continue;
#pragma unroll
          for (int h = 0; h < 2; ++h)
            if ((colm[h] >> lane) & 1ull) emit(r * nx + lane + 64 * h, en);
        }
      }
    }
  }
}

// exclusive scan of the per-tile counts of one apex -> list starts (in place), the total behind the last tile; the lists fit?
__global__ void __launch_bounds__(1024) k_bin_scan(BinBuild bb) {
  const int a = blockIdx.x;
  if (!bb.g[a].on) return;
  char *base = bb.base[a];
  uint32_t *starts = (uint32_t *)(base + ffx_bin_off_starts());
  const int nt = bb.g[a].nx * bb.g[a].ny;
  __shared__ uint32_t s_part[1024];
  const int per = (nt + 1023) / 1024; // <= 16
  const int t0 = threadIdx.x * per;
  uint32_t loc[16];
  uint32_t sum = 0;
  for (int i = 0; i < per; ++i) {
    const int t = t0 + i;
    loc[i] = t < nt ? starts[t] : 0u;
    sum += loc[i];
  }
  s_part[threadIdx.x] = sum;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) { // Hillis-Steele inclusive scan of the 1024 partial sums
    const uint32_t v = threadIdx.x >= (unsigned)off ? s_part[threadIdx.x - off] : 0u;
    __syncthreads();
    s_part[threadIdx.x] += v;
    __syncthreads();
  }
  uint32_t run = s_part[threadIdx.x] - sum; // exclusive
  for (int i = 0; i < per; ++i) {
    const int t = t0 + i;
    if (t < nt) starts[t] = run;
    run += loc[i];
  }
  if (threadIdx.x == 1023) {
    const uint32_t total = s_part[1023];
    starts[nt] = total;
    BinHdr *h = (BinHdr *)base;
    h->total = total;
    h->cap = bb.cap;
    h->ok = total <= bb.cap ? 1u : 0u;
  }
}

void ffx_bins_clear_launch(const BinBuild &bb, hipStream_t s) {
  hipLaunchKernelGGL(k_bin_clear, dim3(ffx_cdiv(FFX_BIN_MAX_TILES + 1, 256), FFX_N_APEX), dim3(256), 0, s, bb);
}

void ffx_bins_launch(const TriRec *recs, int n_tris, const BinBuild &bb, hipStream_t s) {
  const dim3 grid(ffx_cdiv(n_tris, 256), FFX_N_APEX);
  hipLaunchKernelGGL(k_bin<false>, grid, dim3(256), 0, s, recs, n_tris, bb);
  hipLaunchKernelGGL(k_bin_scan, dim3(FFX_N_APEX), dim3(1024), 0, s, bb);
  hipLaunchKernelGGL(k_bin<true>, grid, dim3(256), 0, s, recs, n_tris, bb);
}
