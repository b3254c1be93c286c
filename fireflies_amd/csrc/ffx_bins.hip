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
// Three launches on the stream of the re-fit (normally the side stream, off the renders' critical path):
//   k_bin<false>  a lane per triangle and apex: writes the triangle's APEX RECORD (ffx_common.h TriApex — what k_apex_records did as a
//                 launch of its own), projects, classifies and counts the tiles it touches (one atomic per tile into `cursors`)
//   k_bin_scan    a wave per grid: the counts into list starts (rows of 64 scanned on the DPP network), overflow check against the capacity
//   k_bin<true>   the same walk again, writing the 64-byte entries at start[tile] + --cursor[tile]: the cursors count back down to
//                 zero, which is what the next pose's counting pass needs to find (no clearing launch)
// Triangles that touch up to sixteen tiles loop over them per lane; larger ones (faces next to the apex) and triangles the projection
// is not trusted for (a vertex behind / beside the apex) are handled by the whole wave, lanes on tiles.  Entry order inside a tile is the order
// of the atomics — it does not affect a result: closest hits carry the primitive-id tie-break, any-hit answers are booleans.
#include <string.h>

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

// the scan of one grid's counts (in `cursors`) into list starts by ONE wave: rows of 64 tiles (coalesced), a row scanned across the lanes on
// the DPP network, the running total carried in a scalar, eight rows' counts fetched per trip.  One wave and no LDS, because the
// launch runs beside a render that keeps every CU full: with the counts copied to LDS (16 KB for 64x64 tiles, 64 KB for the colon's
// 128x128) the workgroup could not be placed until a render had drained (134 us on average inside the bench loop, rocprofv3, against 7
// alone), and four waves that must start on one CU together still waited 41 us.  The cursors keep the counts (the fill pass counts them
// down) unless the lists do not fit — then they are cleared here, because no fill pass will run.
__device__ void bin_scan_one(char *base, int nt, uint32_t cap, bool env) {
  uint32_t *starts = (uint32_t *)(base + ffx_bin_off_starts());
  uint32_t *cursors = (uint32_t *)(base + ffx_bin_off_cursors());
  const int lane = threadIdx.x;
  const int rows = (nt + 63) >> 6;
  uint32_t run = 0;
  auto load = [&](int r) { const int i = r * 64 + lane; return (r < rows && i < nt) ? cursors[i] : 0u; };
  for (int r8 = 0; r8 < rows; r8 += 8) { // eight rows per trip: their loads are in flight together (one memory round trip per trip, not per row)
    uint32_t cnt[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) cnt[u] = load(r8 + u);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = (r8 + u) * 64 + lane;
      const uint32_t c = cnt[u];
      // inclusive scan across the wave (six DPP adds): Hillis-Steele inside the rows of 16 (row_shr 1, 2, 4, 8: lanes without a source add
      // 0), then rows 1 and 3 take lane 15 of the row before (row_bcast:15) and rows 2, 3 lane 31 (row_bcast:31)
      uint32_t incl = c;
      incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x111, 0xf, 0xf, false);
      incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x112, 0xf, 0xf, false);
      incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x114, 0xf, 0xf, false);
      incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x118, 0xf, 0xf, false);
      incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x142, 0xa, 0xf, false);
      incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x143, 0xc, 0xf, false);
      if (i < nt) starts[i] = run + incl - c;
      run += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    }
  }
  const uint32_t total = run;
  const bool ok = total <= cap;
  if (!ok)
    for (int i = lane; i < nt; i += 64) cursors[i] = 0u;
  if (lane == 0) {
    starts[nt] = total;
    BinHdr *h = (BinHdr *)base;
    h->total = total;
    h->cap = cap;
    h->ok = ok ? 1u : 0u;
    h->env = env && ok ? 1u : 0u; // (k_bin_env leaves a grid whose lists do not fit alone)
  }
}

// apex records of the pre-pass (k_apex_records of ffx_trace.hip, same arithmetic): what to write for apex a
struct BinApex { float o[FFX_N_APEX][3]; int on[FFX_N_APEX]; TriApex *out; uint32_t astride; uint32_t *cache_hdr; uint32_t cap_stray;
                 uint32_t *gnw; int clear_on; }; // gnw: the per-slot normals as words (word 4 k + 3: shape / smooth / clear bits); clear_on: k_bin_clear will run

// One wave per workgroup whose registers fit the hole ONE retired render wave leaves: these launches run beside a render whose one-wave
// workgroups refill every slot the moment it frees.  A four-wave workgroup of 96-VGPR waves waited for four slots and enough registers on ONE
// compute unit at the same time and hardly ever found them (2 058 -> 2 400 renders/s when it became one wave); and a one-wave workgroup of 80
// VGPRs beside the Lambert render kernels — eight waves of 64 VGPRs per SIMD — still needed TWO of them to retire together: 141 + 143 us for
// the two binning launches inside that loop instead of 42 + 77, and the loop's period was this chain's (2 773 -> 3 168 renders/s with the
// diffuse material).  Hence WPE: 7 waves per SIMD (72 VGPRs, a few spilled) beside the material-row kernels (seven waves of 72), 8 (64 VGPRs)
// beside the Lambert ones — the launch knows which (ffx_bins_launch `beside_lambert`).
#define BIN_BLOCK 64
template <bool FILL, int WPE = 7>
__global__ void __launch_bounds__(BIN_BLOCK) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) k_bin(const TriRec *__restrict__ recs, int n_tris, BinBuild bb, BinApex ba) {
  FFX_SIDE_PRIO();
  const int a = blockIdx.y;
  const int k = blockIdx.x * BIN_BLOCK + threadIdx.x;
  const int lane = threadIdx.x & 63;
  float4 ra = make_float4(0.f, 0.f, 0.f, 0.f), rb = ra, rc = ra;
  if (k < n_tris) {
    const float4 *r4 = reinterpret_cast<const float4 *>(recs + k);
    ra = r4[0]; rb = r4[1]; rc = r4[2];
  }
  if (!FILL) {
    if (k == 0 && a == 0 && ba.cache_hdr) { // adjoint cache: stray arena empty (top bit of cap_stray: keep the `dropped` count, ffx_common.h)
      ba.cache_hdr[0] = 0u; ba.cache_hdr[1] = ba.cap_stray & ~FFX_CAP_KEEP_DROPPED;
      if (!(ba.cap_stray & FFX_CAP_KEEP_DROPPED)) ba.cache_hdr[2] = 0u;
      for (int i = 4; i < 12; ++i) ba.cache_hdr[i] = 0u; // (the filtered film's cache: its arena's eight counters, CacheHdr.pad[1..8])
    }
    if (k < n_tris && ba.out && ba.on[a]) { // the triangle as seen from apex a (ffx_common.h TriApex; the oracle's operation order)
      const v3 v0 = V3(ra.x, ra.y, ra.z), e1 = V3(ra.w, rb.x, rb.y), e2 = V3(rb.z, rb.w, rc.x);
      const v3 A = vcross(e2, e1);
      const v3 tv = vsub(V3(ba.o[a][0], ba.o[a][1], ba.o[a][2]), v0);
      const v3 B = vcross(e2, tv), Cc = vcross(tv, e1);
      const float T = vdot(e2, Cc);
      float4 *o4 = reinterpret_cast<float4 *>(reinterpret_cast<char *>(ba.out) + (size_t)a * ba.astride) + 3 * (size_t)k;
      o4[0] = make_float4(A.x, A.y, A.z, B.x);
      o4[1] = make_float4(B.y, B.z, Cc.x, Cc.y);
      o4[2] = make_float4(Cc.z, T, rc.y, rc.z); // prim, shape as in TriRec
    }
  }
  const bool grid_on = bb.g[a].on != 0 && bb.base[a] != nullptr;
  if (!FILL && a >= 1 && ba.gnw && ba.clear_on && k < n_tris) { // (clear_on == 0: the render kernels do not look at the bits)
    // the emitter's "clear" bit of this triangle (ffx_common.h FFX_GN_CLEAR_BIT): set here for every triangle when the proof will run,
    // taken back by k_bin_clear where it fails; cleared otherwise (a blob whose pre-pass ran before with other emitter positions).  A
    // degenerate triangle's word stays 0 — that IS its flag.
    uint32_t *w = ba.gnw + 4 * (size_t)k + 3;
    if ((*w & FFX_GN_SHAPE_MASK) != 0u) {
      if (grid_on && ((ba.clear_on >> (a - 1)) & 1) && ba.on[a]) atomicOr(w, FFX_GN_CLEAR_BIT(a));
      else atomicAnd(w, ~FFX_GN_CLEAR_BIT(a));
    }
  }
  char *base = bb.base[a];
  uint32_t *starts = (uint32_t *)(base + ffx_bin_off_starts());
  uint32_t *cursors = (uint32_t *)(base + ffx_bin_off_cursors());
  BinEntry *ents = (BinEntry *)(base + ffx_bin_off_entries());
  const int nx = bb.g[a].nx, ny = bb.g[a].ny;
  const bool work = grid_on && !(FILL && ((const BinHdr *)base)->ok == 0u); // (fill: the lists do not fit — this pose renders through the tree walks)
  auto emit = [&](int tile, const BinEntry &en) {
    if (!FILL) atomicAdd(&cursors[tile], 1u);
    else {
      const uint32_t i = atomicSub(&cursors[tile], 1u) - 1u;
      ents[starts[tile] + i] = en;
    }
  };
  float X[3] = {0.f, 0.f, 0.f}, Y[3] = {0.f, 0.f, 0.f}, Z[3] = {-1.f, -1.f, -1.f};
  if (work && k < n_tris) {
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
  // class 0: wholly behind the apex plane (no ray of the grid reaches it), 1: touches at most sixteen tiles, 2: more, 3: the projection
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
        if (tx0 <= tx1 && ty0 <= ty1) cls = (tx1 - tx0 + 1) * (ty1 - ty0 + 1) <= 16 ? 1 : 2;
      }
    } else {
      // not trusted — but most such triangles (the ring of the tube beside the apex, 85 - 90 degrees off its axis) miss the grid's whole
      // frustum: all three vertices beyond one of its four planes (any vertex position, also behind the apex).  The others take the wave.
      const float gx = (float)nx + FFX_BIN_PAD, gy = (float)ny + FFX_BIN_PAD;
      bool oL = true, oR = true, oT = true, oB = true;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        oL = oL && (X[i] + FFX_BIN_PAD * Z[i] < 0.f);
        oR = oR && (gx * Z[i] - X[i] < 0.f);
        oT = oT && (Y[i] + FFX_BIN_PAD * Z[i] < 0.f);
        oB = oB && (gy * Z[i] - Y[i] < 0.f);
      }
      cls = (oL || oR || oT || oB) ? 0 : 3;
    }
  }
#ifdef FFX_BINPROBE // timing experiments (tools/binprobe.py): 1 = no wave-cooperative path, 2 = no per-lane path, 3 = neither
  if (FFX_BINPROBE & 2) { if (cls == 1) cls = 0; }
  if (FFX_BINPROBE & 1) { if (cls >= 2) cls = 0; }
#endif
  {
    // ---- triangles that touch at most sixteen tiles, a lane each.  Which tiles: a bit mask over the lane's own window (no atomics
    // yet).  Then the wave walks its lanes' masks in rounds, every lane offering its next tile, and lanes that offer the SAME tile —
    // neighbouring leaf slots are neighbouring triangles: usually a dozen of them — are served by ONE atomic of their first lane:
    // a tile of the far wall receives 400 entries, and 400 same-address atomics from eight XCDs were the tail of both launches
    // (they serialise at the memory side, ~90 ns each).  Four rounds are in flight at a time: a fill-pass atomic returns the group's
    // place in the list, and one dependent round trip per round was the rest of that tail.
    // (round 6, fill pass) the lane's 64-byte entry waits in LDS instead of sixteen registers — the kernel had spilled 27 - 50 of them — and
    // leaves through it TRANSPOSED: four lanes write the four 16-byte quarters of ONE entry side by side, a whole 64-byte line per entry,
    // where every lane used to issue four scattered 16-byte stores of its own (write traffic of the launch 31.7 -> MB, profiles/r6_pmc_summary.json)
    __shared__ __attribute__((aligned(16))) float4 s_en[FILL ? 4 * BIN_BLOCK : 1];
    uint32_t touched = 0u;
    const int w = tx1 - tx0 + 1;
    if (cls == 1) {
      BinEntry en;
      bin_make_entry(x[0], y[0], x[1], y[1], x[2], y[2], k, false, en);
      const int nt_ = w * (ty1 - ty0 + 1);
#pragma nounroll
      for (int t = 0; t < nt_; ++t) {
        const int ty = ty0 + t / w, tx = tx0 + t % w;
        if (bin_entry_touches(en, (float)tx, (float)ty, (float)(tx + 1), (float)(ty + 1))) touched |= 1u << t;
      }
      if (FILL) {
        const float4 *e4 = reinterpret_cast<const float4 *>(&en);
#pragma unroll
        for (int p = 0; p < 4; ++p) s_en[4 * lane + p] = e4[p];
      }
    }
    if (FILL) __builtin_amdgcn_wave_barrier(); // (one wave per workgroup: program order + the LDS counter order the reads below behind these writes)
    const wmask_t below = (1ull << lane) - 1ull;
#pragma nounroll
    while (__ballot(touched != 0u) != 0ull) {
      int tl[4], leader[4], rank[4], cnt[4];
      uint32_t at[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        tl[q] = -1; leader[q] = lane; rank[q] = 0; cnt[q] = 0; at[q] = 0u;
        if (touched != 0u) {
          const int t = __builtin_ctz(touched);
          touched &= touched - 1u;
          tl[q] = (ty0 + t / w) * nx + tx0 + t % w;
        }
        wmask_t pend = __ballot(tl[q] >= 0);
#pragma nounroll
        while (pend != 0ull) { // group the lanes of this round by tile
          const int ld = __builtin_ctzll(pend);
          const int lt = __builtin_amdgcn_readlane(tl[q], ld);
          const wmask_t same = __ballot(tl[q] == lt);
          if (tl[q] == lt) { leader[q] = ld; rank[q] = __builtin_popcountll(same & below); cnt[q] = __builtin_popcountll(same); }
          pend &= ~same;
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (tl[q] >= 0 && leader[q] == lane) {
          if (!FILL) atomicAdd(&cursors[tl[q]], (uint32_t)cnt[q]);
          else at[q] = atomicSub(&cursors[tl[q]], (uint32_t)cnt[q]) - (uint32_t)cnt[q];
        }
      if (FILL) {
        float4 *ents4 = reinterpret_cast<float4 *>(ents);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const uint32_t base_ = (uint32_t)__shfl((int)at[q], leader[q], 64);
          const int dst = tl[q] >= 0 ? (int)(starts[tl[q]] + base_ + (uint32_t)rank[q]) : -1; // where this lane's entry goes (-1: nowhere this round)
#pragma unroll
          for (int i = 0; i < 4; ++i) { // entries of lanes 16 i .. 16 i + 15: lane l writes quarter l % 4 of lane 16 i + l / 4's
            const int src = 16 * i + (lane >> 2);
            const int d = __shfl(dst, src, 64);
            if (d >= 0) ents4[4 * (size_t)d + (lane & 3)] = s_en[4 * src + (lane & 3)];
          }
        }
      }
    }
  }
  // ---- the whole wave on one triangle at a time: lanes on tiles
  wmask_t big = __ballot(cls >= 2);
  while (big != 0ull) {
    const int j = __builtin_ctzll(big);
    big &= big - 1ull;
    // (j is wave-uniform: v_readlane — a shuffle would be an LDS round trip per value)
    auto rl = [&](float v) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), j)); };
    const int jc = __builtin_amdgcn_readlane(cls, j), js = __builtin_amdgcn_readlane(k, j);
    BinEntry en;
    if (jc == 2) {
      const float q0 = rl(x[0]), q1 = rl(y[0]), q2 = rl(x[1]), q3 = rl(y[1]), q4 = rl(x[2]), q5 = rl(y[2]);
      const int bx0 = __builtin_amdgcn_readlane(tx0, j), bx1 = __builtin_amdgcn_readlane(tx1, j), by0 = __builtin_amdgcn_readlane(ty0, j), by1 = __builtin_amdgcn_readlane(ty1, j);
      bin_make_entry(q0, q1, q2, q3, q4, q5, js, false, en);
      const int w = bx1 - bx0 + 1, n = w * (by1 - by0 + 1);
      for (int t = lane; t < n; t += 64) {
        const int ty = by0 + t / w, tx = bx0 + t % w;
        if (bin_entry_touches(en, (float)tx, (float)ty, (float)(tx + 1), (float)(ty + 1))) emit(ty * nx + tx, en);
      }
    } else {
      float PX[3], PY[3], PZ[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) { PX[i] = rl(X[i]); PY[i] = rl(Y[i]); PZ[i] = rl(Z[i]); }
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

// a wave per grid: its counts into list starts.  (A launch of its own: folded into the counting launch as the job of the workgroup
// that finishes last, it needed an agent-scope release fence in EVERY workgroup — which on this part writes back the whole L2, 7.7 MB of
// fresh apex records included: the counting launch took 65 us instead of 10.)
__global__ void __launch_bounds__(64) k_bin_scan(BinBuild bb) {
  FFX_SIDE_PRIO();
  const int g = blockIdx.x;
  if (bb.g[g].on && bb.base[g]) bin_scan_one(bb.base[g], bb.g[g].nx * bb.g[g].ny, bb.cap, g >= 1 && ((bb.env_mask >> (g - 1)) & 1));
}

// ---- "clear" triangles (ffx_common.h FFX_GN_CLEAR_BIT): the proof that nothing can shadow a triangle from an emitter.
// Shadow segments run from the emitter E to Po = P + off n_k, P on triangle k, n_k its unit normal on the viewer's side — which is E's
// side for every sample E lights — off = (1 + max|P|) 8.9e-5 >= 8.9e-5; hits at t >= 1 - 8.9e-4 do not count.  Triangle j cannot
// intersect the counted part of ANY such segment if
//   (H0) its projection from E is apart from k's (padded boxes in the grid: a segment is a POINT of E's image plane), or
//   (H1) all of j lies behind k's plane or within 2e-5 above it: the counted part of the segment stays >= 8.9e-5 (1 - 8.9e-4) above, or
//   (H2) j faces E, the two E-side normals agree (cos >= 0.5) and all of k lies in front of j's plane or within 2e-5 behind it: then
//        Po (lifted by >= 8.9e-5 x 0.5) and E are both strictly in front of j's plane, and so is the whole segment.
// The tolerances shrink by the rounding of the plane evaluations (3e-7 of the coordinates' magnitude: scenes beyond ~60 units lose them and
// keep the strict tests).  Two triangles whose projections overlap share a tile of E's grid, so testing k against the entries of its
// tiles is complete.  Unsafe projections (a vertex behind / beside the apex) have infinite boxes: H0 never holds for them, H1 / H2 are 3-D.
// A wave per (tile, chunk of 64 entries k): lanes on k, the tile's entries j one after the other (uniform: scalar loads).
// The entries j are walked 64 at a time: every lane loads ONE of them and derives its plane (normal towards the emitter, its length, the
// facing test, the coordinate magnitude) — one memory round trip per 64 entries — and the inner loop broadcasts entry after entry with
// v_readlane.  (First version: entry j and its record fetched inside the loop, two dependent loads per iteration: 357 us beside a render
// for 9 M instructions' worth of work, the loop's period.)
#define CLEAR_SPLIT 4
struct ClearTri { v3 a, b, c, n; float len, s, M; float4 bb; int slot; };
__device__ __forceinline__ ClearTri clear_load(const char *__restrict__ ents, const TriRec *__restrict__ recs, uint32_t i, v3 E) {
  ClearTri t;
  const float4 *e4 = reinterpret_cast<const float4 *>(ents + ((size_t)i << 6));
  t.bb = e4[0];
  t.slot = __float_as_int(e4[3].y);
  const float4 *r4 = reinterpret_cast<const float4 *>(recs + t.slot);
  const float4 ra = r4[0], rb = r4[1], rc = r4[2];
  const v3 e1 = V3(ra.w, rb.x, rb.y), e2 = V3(rb.z, rb.w, rc.x);
  t.a = V3(ra.x, ra.y, ra.z);
  t.b = V3(t.a.x + e1.x, t.a.y + e1.y, t.a.z + e1.z);
  t.c = V3(t.a.x + e2.x, t.a.y + e2.y, t.a.z + e2.z);
  t.n = vcross(e1, e2);
  t.s = vdot(t.n, vsub(E, t.a));
  if (t.s < 0.f) { t.n = V3(-t.n.x, -t.n.y, -t.n.z); t.s = -t.s; } // towards the emitter
  t.len = sqrtf(vdot(t.n, t.n));
  t.M = fmaxf(fmaxf(fmaxf(fabsf(t.a.x), fabsf(t.a.y)), fabsf(t.a.z)), fmaxf(fmaxf(fabsf(e1.x) + fabsf(e2.x), fabsf(e1.y) + fabsf(e2.y)), fabsf(e1.z) + fabsf(e2.z)));
  return t;
}
template <int WPE>
__global__ void __launch_bounds__(BIN_BLOCK) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) k_bin_clear(const TriRec *__restrict__ recs, BinBuild bb, BinApex ba) {
  FFX_SIDE_PRIO();
  const int a = 1 + (int)blockIdx.y;
  if (!bb.g[a].on || !bb.base[a] || !ba.on[a] || !ba.gnw || !((ba.clear_on >> (a - 1)) & 1)) return;
  const char *base = bb.base[a];
  if (((const BinHdr *)base)->ok == 0u) return; // (lists incomplete: the render kernels do not look at the bits then, bins_ready)
  const int tile = (int)blockIdx.x / CLEAR_SPLIT, part = (int)blockIdx.x % CLEAR_SPLIT;
  if (tile >= bb.g[a].nx * bb.g[a].ny) return;
  const uint32_t *starts = (const uint32_t *)(base + ffx_bin_off_starts());
  const uint32_t beg = starts[tile], n = starts[tile + 1] - beg;
  if (n < 2u) return; // alone in its tile
  const char *ents = base + ffx_bin_off_entries() + ((size_t)beg << 6);
  const uint32_t lane = threadIdx.x & 63u;
  const v3 E = V3(ba.o[a][0], ba.o[a][1], ba.o[a][2]);
  const uint32_t bit = FFX_GN_CLEAR_BIT(a);
  auto bc = [](float v, uint32_t l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), (int)l)); };
  for (uint32_t k0 = (uint32_t)part * 64u; k0 < n; k0 += 64u * CLEAR_SPLIT) {
    const uint32_t ki = k0 + lane;
    const bool valid = ki < n;
    const ClearTri K = clear_load(ents, recs, valid ? ki : n - 1u, E);
    // (already refuted in another tile: nothing to prove.  The emitter must stand clearly off k's plane — 1e-4: above the tolerances — or the
    // first part of a segment is no higher above it than j may be)
    bool open = valid && (ba.gnw[4 * (size_t)K.slot + 3] & bit) != 0u;
    bool unclear = open && !(K.len > 0.f && K.s > 1e-4f * K.len);
    open = open && !unclear;
    for (uint32_t j0 = 0; j0 < n && __ballot(open) != 0ull; j0 += 64u) { // (uniform)
      const uint32_t m = min(64u, n - j0);
      const ClearTri J = clear_load(ents, recs, j0 + lane < n ? j0 + lane : n - 1u, E); // this lane's entry of the chunk
      for (uint32_t t = 0; t < m && __ballot(open) != 0ull; ++t) { // (uniform) entry j0 + t, broadcast
        const int slot_j = __builtin_amdgcn_readlane(J.slot, (int)t);
        const float bx0 = bc(J.bb.x, t), by0 = bc(J.bb.y, t), bx1 = bc(J.bb.z, t), by1 = bc(J.bb.w, t);
        const bool other = open && slot_j != K.slot;
        const bool apart = K.bb.x > bx1 || K.bb.z < bx0 || K.bb.y > by1 || K.bb.w < by0; // (H0; an unsafe entry's infinite box is never apart)
        if (__ballot(other && !apart) == 0ull) continue;
        const v3 ja = V3(bc(J.a.x, t), bc(J.a.y, t), bc(J.a.z, t)), jb = V3(bc(J.b.x, t), bc(J.b.y, t), bc(J.b.z, t)), jc = V3(bc(J.c.x, t), bc(J.c.y, t), bc(J.c.z, t));
        const v3 nj = V3(bc(J.n.x, t), bc(J.n.y, t), bc(J.n.z, t));
        const float lenj = bc(J.len, t), sj = bc(J.s, t), Mj = bc(J.M, t);
        const float tol = 2e-5f - 3e-7f * (K.M + Mj); // world units; <= 0 for large coordinates: the strict tests remain
        // H1: j behind k's plane
        // (differences first: a neighbour's vertex minus k's is exact in float, Sterbenz — the products then carry no cancellation)
        const float tk = tol * K.len;
        const bool h1 = vdot(K.n, vsub(ja, K.a)) <= tk && vdot(K.n, vsub(jb, K.a)) <= tk && vdot(K.n, vsub(jc, K.a)) <= tk;
        // H2: j faces the emitter, the emitter-side normals agree, k in front of j's plane
        const float tj = -tol * lenj;
        const bool h2 = sj > 1e-5f * lenj && vdot(K.n, nj) >= 0.5f * K.len * lenj && vdot(nj, vsub(K.a, ja)) >= tj && vdot(nj, vsub(K.b, ja)) >= tj && vdot(nj, vsub(K.c, ja)) >= tj;
        if (other && !apart && !h1 && !h2) { unclear = true; open = false; }
      }
    }
    if (valid && unclear) atomicAnd(ba.gnw + 4 * (size_t)K.slot + 3, ~bit);
  }
}

// ---- the ENVELOPE of an emitter's grid (ffx_common.h FFX_ENV_SUB; round 6).  The spot's any-hit stage is a quarter of the render kernel for an
// emitter next to the camera that hardly anything shadows, and a per-triangle proof (k_bin_clear) settles a third of the
// triangles at best: the neighbours of a triangle on a curved surface do come within the shadow ray's ignored tail of it.  What CAN be stated
// cheaply is where the front of everything a tile lists lies.  Seen from the emitter E a triangle's plane is affine in 1 / depth:
//     1 / t_j(d) = d . n_j,  n_j = A_j / T_j  (the apex record's own numbers),      d = Z Minv (x, y, 1)   =>   1 / (t_j Z) = m_j . (x, y, 1)
// for the ray towards tile-space point (x, y).  A wave per tile, its 64 lanes on the 8 x 8 VERTICES of the tile's 7 x 7 cells; the tile's
// entries one after the other (lanes on the entries first: 64 planes per trip into LDS, then broadcast reads): a vertex keeps the largest
// plane value of the entries whose padded box touches one of the cells around it.  Over a cell each of those planes lies below the bilinear
// patch of the cell's four vertex values (affine functions are reproduced by bilinear interpolation, whose weights are non-negative), and
//     c0 + ax (x - X0) + ay (y - Y0),   ax, ay the patch's mean slopes,  c0 = max over the corners of (corner - slopes)
// lies above the patch (their difference is bilinear: extremal at the corners).  In world space that plane is N . d <= 1 with
//     N = kap (ax (M0 - X0 M2) + ay (M1 - Y0 M2) + c0 M2),   kap = (1 - 10 eps)(1 + 6e-5):
// a shadow direction d = Po - E with N . d <= 1 has kap / t_j <= 1 for every listed triangle that can contain its image point — t_j beyond
// the counted part of the ray with 6e-5 to spare for the roundings here (few 1e-6: everything is formed in tile-LOCAL coordinates) and of the
// exact test's own det (2.4e-7 |d||A_j| / |d . A_j|: an entry steeper than 1 : 40 against the rays of a vertex poisons it, +inf).
// Poisoned cells hold NaN (the render's `<= 1` fails), cells nothing touches hold 0 (nothing to hit: always proven).
// (First version: lanes on the CELLS, four corner values each — 24 VALU per entry instead of 10, 100 us beside a render; the launch sits in
// the chain the next render waits for.)
struct EnvArgs { const TriApex *arecs; uint32_t astride; int n_first; /* workgroups of emitter 1 (0: its envelope is off) */ };
template <int WPE>
__global__ void __launch_bounds__(BIN_BLOCK) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) k_bin_env(BinBuild bb, EnvBuild eb, EnvArgs ea) {
  FFX_SIDE_PRIO();
  // (one launch for both emitters: the projector's tiles first, then the spot's)
  const int n1 = ea.n_first;
  const int a = (int)blockIdx.x < n1 ? 1 : 2;
  const int tile = (int)blockIdx.x - (a == 1 ? 0 : n1);
  const int nx = bb.g[a].nx;
  const char *base = bb.base[a];
  if (((const BinHdr *)base)->ok == 0u) return; // (lists incomplete: this pose's packets walk the tree and never look here, bins_ready)
  const uint32_t *starts = (const uint32_t *)(base + ffx_bin_off_starts());
  const uint32_t beg = starts[tile], n = starts[tile + 1] - beg;
  const int lane = threadIdx.x & 63;
  const int ty = tile / nx, tx = tile - ty * nx;
  const int vx = lane & 7, vy = lane >> 3; // this lane's vertex; cell (vx, vy) for vx, vy < FFX_ENV_SUB
  float4 *env = reinterpret_cast<float4 *>(bb.base[a] + eb.env_off) + ((size_t)(ty * FFX_ENV_SUB + vy) * (size_t)(nx * FFX_ENV_SUB)) + (size_t)(tx * FFX_ENV_SUB + vx);
  const bool cell = vx < FFX_ENV_SUB && vy < FFX_ENV_SUB;
  if (n == 0u) { // (outside the scene as the emitter sees it: nothing listed, nothing to hit)
    if (cell) *env = make_float4(0.f, 0.f, 0.f, 0.f);
    return;
  }
  const float4 *ents = reinterpret_cast<const float4 *>(base + ffx_bin_off_entries()) + 4 * (size_t)beg;
  const char *rbase = reinterpret_cast<const char *>(ea.arecs) + (size_t)a * ea.astride;
  const float h = 1.0f / FFX_ENV_SUB, pe = 1.0f / 4096.0f; // (cells a little wider than they are: the render's cell index comes from a rounded quotient)
  const float lx = (float)vx * h, ly = (float)vy * h;       // the vertex, tile-local
  const float bx0 = (float)tx + lx - h - pe, bx1 = (float)tx + lx + h + pe, by0 = (float)ty + ly - h - pe, by1 = (float)ty + ly + h + pe; // the cells around it
  const float *Mi = eb.Minv[a];
  // direction through the tile's origin and the two tile-space axes:  Minv (x, y, 1) = u0 + (x - tx) ux + (y - ty) uy
  const v3 ux = V3(Mi[0], Mi[3], Mi[6]), uy = V3(Mi[1], Mi[4], Mi[7]);
  const v3 u0 = V3(fmaf(Mi[0], (float)tx, fmaf(Mi[1], (float)ty, Mi[2])), fmaf(Mi[3], (float)tx, fmaf(Mi[4], (float)ty, Mi[5])), fmaf(Mi[6], (float)tx, fmaf(Mi[7], (float)ty, Mi[8])));
  __shared__ __attribute__((aligned(16))) float4 s_bb[BIN_BLOCK];
  __shared__ __attribute__((aligned(16))) float4 s_pl[BIN_BLOCK];
  float w = -INFINITY;
  for (uint32_t j0 = 0; j0 < n; j0 += 64u) { // (uniform)
    const uint32_t i = j0 + (uint32_t)lane;
    float4 bbv = make_float4(INFINITY, INFINITY, -INFINITY, -INFINITY), pl = make_float4(0.f, 0.f, 0.f, 0.f); // (beyond the list: touches nothing)
    if (i < n) {
      bbv = ents[4 * (size_t)i];
      const int slot = __float_as_int(ents[4 * (size_t)i + 3].y);
      const float4 *r4 = reinterpret_cast<const float4 *>(rbase + (size_t)slot * 48u);
      const float4 rA = r4[0];
      const float T = r4[2].y;
      const float iT = 1.0f / T;
      const v3 nj = V3(rA.x * iT, rA.y * iT, rA.z * iT);
      pl = make_float4(vdot(nj, ux), vdot(nj, uy), vdot(nj, u0), sqrtf(vdot(nj, nj)) * eb.graz[a]);
    }
    if (j0 != 0u) __syncthreads(); // (one wave: orders this trip's writes behind the last trip's reads)
    s_bb[lane] = bbv;
    s_pl[lane] = pl;
    __syncthreads();
    const uint32_t m = min(64u, n - j0);
    for (uint32_t t = 0; t < m; ++t) { // (uniform: broadcast reads)
      const float4 b = s_bb[t], p = s_pl[t];
      asm volatile("" ::"v"(b.x), "v"(b.y), "v"(b.z), "v"(b.w), "v"(p.x), "v"(p.y), "v"(p.z), "v"(p.w)); // (both rows read up front: no lazy, branchy re-reads)
      const bool ov = (b.x <= bx1) & (b.z >= bx0) & (b.y <= by1) & (b.w >= by0); // (an unsafe entry's infinite box: every vertex)
      const float pv = fmaf(p.x, lx, fmaf(p.y, ly, p.z));
      const float q = pv >= p.w ? pv : INFINITY; // edge-on against this vertex's ray, behind the emitter there, or T = 0 (NaN): poison
      w = ov ? fmaxf(w, q) : w;
    }
  }
  // the cell's four vertices: this lane's, and those of the lanes 1, 8 and 9 further
  const float w00 = w, w10 = __shfl_down(w, 1, 64), w01 = __shfl_down(w, 8, 64), w11 = __shfl_down(w, 9, 64);
  if (!cell) return;
  float4 out = make_float4(0.f, 0.f, 0.f, 0.f); // a vertex nothing is near: nothing listed touches the cell
  if (w00 != -INFINITY && w10 != -INFINITY && w01 != -INFINITY && w11 != -INFINITY) {
    const float ih = (float)FFX_ENV_SUB;
    const float ax = 0.5f * ((w10 - w00) + (w11 - w01)) * ih, ay = 0.5f * ((w01 - w00) + (w11 - w10)) * ih;
    float c0 = fmaxf(fmaxf(w00, w10 - ax * h), fmaxf(w01 - ay * h, w11 - (ax + ay) * h));
    c0 += 2e-6f * (fabsf(c0) + (fabsf(ax) + fabsf(ay)) * h);
    const float gx0 = (float)tx + lx, gy0 = (float)ty + ly;
    const float *M = bb.g[a].M;
    const v3 m2 = V3(M[6], M[7], M[8]);
    const v3 mx = V3(fmaf(-gx0, m2.x, M[0]), fmaf(-gx0, m2.y, M[1]), fmaf(-gx0, m2.z, M[2]));
    const v3 my = V3(fmaf(-gy0, m2.x, M[3]), fmaf(-gy0, m2.y, M[4]), fmaf(-gy0, m2.z, M[5]));
    const v3 N = V3(fmaf(ax, mx.x, fmaf(ay, my.x, c0 * m2.x)), fmaf(ax, mx.y, fmaf(ay, my.y, c0 * m2.y)), fmaf(ax, mx.z, fmaf(ay, my.z, c0 * m2.z)));
    out = make_float4(N.x * eb.kap, N.y * eb.kap, N.z * eb.kap, 0.f);
    const bool fin = fabsf(out.x) < 3.0e38f && fabsf(out.y) < 3.0e38f && fabsf(out.z) < 3.0e38f; // (false for NaN, and for a poisoned vertex: inf - inf)
    if (!fin) out = make_float4(NAN, NAN, NAN, 0.f);
  }
  *env = out;
}

void ffx_bins_launch(const TriRec *recs, int n_tris, const BinBuild &bb, const void *apex_out, const float (*apex_o)[3], const int *apex_on, uint32_t astride,
                     uint32_t *cache_hdr, uint32_t cap_stray, hipStream_t s, int beside_lambert, uint32_t *gn_words, int clear_on, const EnvBuild *env) {
  BinApex ba;
  memset(&ba, 0, sizeof ba);
  for (int a = 0; a < FFX_N_APEX; ++a) { ba.on[a] = apex_on[a]; ba.o[a][0] = apex_o[a][0]; ba.o[a][1] = apex_o[a][1]; ba.o[a][2] = apex_o[a][2]; }
  ba.out = (TriApex *)apex_out; ba.astride = astride; ba.cache_hdr = cache_hdr; ba.cap_stray = cap_stray;
  ba.gnw = gn_words; ba.clear_on = clear_on;
  const dim3 grid(ffx_cdiv(n_tris, BIN_BLOCK), FFX_N_APEX);
  if (beside_lambert) hipLaunchKernelGGL((k_bin<false, 8>), grid, dim3(BIN_BLOCK), 0, s, recs, n_tris, bb, ba);
  else hipLaunchKernelGGL((k_bin<false, 7>), grid, dim3(BIN_BLOCK), 0, s, recs, n_tris, bb, ba);
  if (bb.g[0].on || bb.g[1].on || bb.g[2].on) {
    hipLaunchKernelGGL(k_bin_scan, dim3(FFX_N_APEX), dim3(64), 0, s, bb);
    if (beside_lambert) hipLaunchKernelGGL((k_bin<true, 8>), grid, dim3(BIN_BLOCK), 0, s, recs, n_tris, bb, ba);
    else hipLaunchKernelGGL((k_bin<true, 7>), grid, dim3(BIN_BLOCK), 0, s, recs, n_tris, bb, ba);
    if (gn_words && clear_on && (((clear_on & 1) && bb.g[1].on && apex_on[1]) || ((clear_on & 2) && bb.g[2].on && apex_on[2]))) { // the emitters' grids: which triangles nothing can shadow
      int nt = 0;
      for (int a = 1; a < FFX_N_APEX; ++a) if (((clear_on >> (a - 1)) & 1) && bb.g[a].on && apex_on[a]) nt = nt > bb.g[a].nx * bb.g[a].ny ? nt : bb.g[a].nx * bb.g[a].ny;
      const dim3 cgrid(nt * CLEAR_SPLIT, FFX_N_APEX - 1);
      if (beside_lambert) hipLaunchKernelGGL((k_bin_clear<8>), cgrid, dim3(BIN_BLOCK), 0, s, recs, bb, ba);
      else hipLaunchKernelGGL((k_bin_clear<7>), cgrid, dim3(BIN_BLOCK), 0, s, recs, bb, ba);
    }
    if (env && apex_out) { // the emitters' envelopes (ffx_common.h FFX_ENV_SUB): a wave per tile
      int nt[FFX_N_APEX] = {0, 0, 0};
      for (int a = 1; a < FFX_N_APEX; ++a) if (env->on[a] && bb.g[a].on && bb.base[a] && apex_on[a]) nt[a] = bb.g[a].nx * bb.g[a].ny;
      if (nt[1] + nt[2] > 0) {
        EnvArgs ea;
        ea.arecs = (const TriApex *)apex_out; ea.astride = astride; ea.n_first = nt[1];
        const dim3 egrid(nt[1] + nt[2]);
        if (beside_lambert) hipLaunchKernelGGL((k_bin_env<8>), egrid, dim3(BIN_BLOCK), 0, s, bb, *env, ea);
        else hipLaunchKernelGGL((k_bin_env<7>), egrid, dim3(BIN_BLOCK), 0, s, bb, *env, ea);
      }
    }
  }
}
