// ffx_common.h — shared declarations for libffx_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/ffx.h"

// ------------------------------------------------------------------ error handling
void ffx_set_error(const char *fmt, ...);
#define FFX_FAIL(code, ...)     \
  do {                          \
    ffx_set_error(__VA_ARGS__); \
    return (code);              \
  } while (0)
// checks the launch itself (not completion: all calls are asynchronous on the stream)
#define FFX_CHECK_LAUNCH(what)                                                           \
  do {                                                                                   \
    hipError_t e_ = hipGetLastError();                                                   \
    if (e_ != hipSuccess) FFX_FAIL(FFX_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e_)); \
  } while (0)

static inline int ffx_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// The kernels of a pose's preparation (re-fit, apex records, tile bins) run on a side stream BESIDE the previous pose's render, whose
// quarter of a million one-wave workgroups keep every SIMD's issue port busy: a wave of these latency-bound kernels then gets one issue
// slot in eight, and their dependent chain — which the next render waits for — took 0.46 ms instead of 0.12 (tools/steptrace.py: the
// step's period WAS that chain).  s_setprio raises the wave's priority at the SIMD's arbiter: their few instructions go first, the
// render loses nothing measurable.  -DFFX_NO_SIDE_PRIO switches it off (A/B).
#ifdef FFX_NO_SIDE_PRIO
#define FFX_SIDE_PRIO() do { } while (0)
#else
#define FFX_SIDE_PRIO() __builtin_amdgcn_s_setprio(3)
#endif

// ------------------------------------------------------------------ small POD blocks passed by value
struct Mat4 { float m[16]; };

// ------------------------------------------------------------------ vector helpers (same operation order as oracle/ffx_oracle.c)
struct v3 { float x, y, z; };
__host__ __device__ inline v3 V3(float x, float y, float z) { v3 r; r.x = x; r.y = y; r.z = z; return r; }
__host__ __device__ inline v3 vsub(v3 a, v3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ inline float diffprod(float a, float b, float c, float d) { return fmaf(a, b, -(c * d)); }
__device__ inline v3 vcross(v3 a, v3 b) {
  return V3(diffprod(a.y, b.z, a.z, b.y), diffprod(a.z, b.x, a.x, b.z), diffprod(a.x, b.y, a.y, b.x));
}
__device__ inline float vdot(v3 a, v3 b) { return fmaf(a.x, b.x, fmaf(a.y, b.y, a.z * b.z)); }
__device__ inline v3 xf_point(const float *m, v3 p) {
  return V3(fmaf(m[0], p.x, fmaf(m[1], p.y, fmaf(m[2], p.z, m[3]))), fmaf(m[4], p.x, fmaf(m[5], p.y, fmaf(m[6], p.z, m[7]))),
            fmaf(m[8], p.x, fmaf(m[9], p.y, fmaf(m[10], p.z, m[11]))));
}
__device__ inline v3 xf_dir(const float *m, v3 d) {
  return V3(fmaf(m[0], d.x, fmaf(m[1], d.y, m[2] * d.z)), fmaf(m[4], d.x, fmaf(m[5], d.y, m[6] * d.z)),
            fmaf(m[8], d.x, fmaf(m[9], d.y, m[10] * d.z)));
}

// counter-based per-sample jitter: lowbias32 integer hash
__host__ __device__ inline uint32_t hash32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}

// ------------------------------------------------------------------ BVH blob layout
// 64-byte node holding BOTH children's boxes: one fetch per traversal step.
// child >= 0: inner node index.  child < 0: leaf, ~child = (first_slot << 3) | (count - 1).
// An empty child has lo = +inf, hi = -inf and child = FFX_EMPTY_CHILD.
struct __attribute__((aligned(16))) BvhNode {
  float lo0[3], hi0[3];
  float lo1[3], hi1[3];
  int32_t c0, c1;
  int32_t pad0, pad1;
};
static_assert(sizeof(BvhNode) == 64, "node must be 64 bytes");
#define FFX_EMPTY_CHILD ((int32_t)0x80000000)
#ifndef FFX_LEAF_MAX
#define FFX_LEAF_MAX 4
#endif

// 48-byte triangle record in leaf order, written by ffx_scene_update
struct __attribute__((aligned(16))) TriRec {
  float v0[3];
  float e1[3];
  float e2[3];
  int32_t prim;
  int32_t shape;
  float pad;
};
static_assert(sizeof(TriRec) == 48, "record must be 48 bytes");

// 48-byte APEX record: a triangle as seen from a fixed ray origin ("apex") o — the camera position for
// primary rays, an emitter position for shadow rays (which are traced from the emitter towards the
// surface).  With tv = o - v0 the Moller-Trumbore scalars of a ray (o, d) are three dot products,
//     det = d . A,  U = d . B,  V = d . C,  t = T / det      A = e2 x e1, B = e2 x tv, C = tv x e1, T = e2 . C
// (DESIGN.md §4.1).  Written per render call by k_apex_records into the apex areas at the end of the
// blob; the oracle computes the same four quantities in the same order on the fly.
struct __attribute__((aligned(16))) TriApex {
  float A[3];
  float B[3];
  float C[3];
  float T;
  int32_t prim;
  int32_t shape;
};
static_assert(sizeof(TriApex) == 48, "apex record must be 48 bytes");
#define FFX_N_APEX 3 // 0 camera, 1 projector, 2 spot
// byte stride of one apex area / offset of area k (they are the last FFX_N_APEX areas of the blob)
static inline uint64_t ffx_apex_stride(int n_tris) { return ((((uint64_t)n_tris + FFX_LEAF_MAX) * 48u) + 63u) & ~(uint64_t)63; }
static inline uint64_t ffx_apex_offset(const ffx_bvh_info *info, int k) { return info->total_bytes - (uint64_t)(FFX_N_APEX - k) * ffx_apex_stride(info->n_tris); }

// ------------------------------------------------------------------ 64-wide overlay
// One child of a wide node, or one triangle of a cluster: a box on the 16-bit grid of the current pose
// (x = org + q * step per axis; q[0..2] round down, q[3..5] round up, one extra cell of slack each) and
// a reference: cluster << 31 | element << 6 | (count - 1), where `element` indexes the ONE array of 16-byte
// elements formed by the wide nodes (64 each) followed by the triangle boxes in leaf-slot order.  In the
// triangle part `ref` is unused.
#ifndef FFX_WIDE_F32
#define FFX_WIDE_F32 1
#endif
#if FFX_WIDE_F32
// float32 boxes (32-byte elements): the 16-bit grid cost six integer-to-float conversions per step of the walk —
// conversions issue at the 4-cycle rate, a third of the box test's issue time (tools/ubench/issue_rates.hip) — for
// half the bytes of a structure that lives in L2 anyway.  WideHdr then holds the identity grid (org 0, step 1).
struct __attribute__((aligned(16))) WideChild {
  float lo[3];
  float hi0;   // hi[0]
  float hi12[2]; // hi[1], hi[2]
  int32_t ref;
  int32_t pad;
};
static_assert(sizeof(WideChild) == 32, "wide child must be 32 bytes");
#define FFX_WIDE_ELEM_SHIFT 5
#else
struct __attribute__((aligned(16))) WideChild {
  uint16_t q[6];
  int32_t ref;
};
static_assert(sizeof(WideChild) == 16, "wide child must be 16 bytes");
#define FFX_WIDE_ELEM_SHIFT 4
#endif
#define FFX_WIDE 64
#define FFX_WIDE_MAX_DEPTH 6
// grid header written by the refit: org[3], step[3] (floats)
struct WideHdr { float org[3]; float step[3]; float pad[10]; };
static_assert(sizeof(WideHdr) == 64, "wide header must be 64 bytes");

// ------------------------------------------------------------------ tile bins (include/ffx.h ffx_bvh_info.off_bins)
// A render's rays leave three fixed points (camera, projector, spot).  Seen from such a point a triangle is a 2-D triangle on a
// perspective image plane, and a ray hits it iff the ray's image point lies inside: per apex the pre-pass bins every triangle of the
// current pose into the tiles of a grid over that plane (a software rasteriser's binning stage), and a pixel's packet — whose rays
// cover a known rectangle of the plane — tests the entries of its tile(s) with the LANES ON THE ENTRIES (box + three edge equations
// against the rectangle) and hands the survivors to the exact apex test.  Conservative by construction: an entry's box and edge
// offsets are padded by FFX_BIN_PAD tile units (~0.03 pixels, four orders of magnitude above the rounding of the projection), a
// triangle is listed in every tile its padded projection touches, and triangles the projection is ill-conditioned for (a vertex
// behind or beside the apex) are listed by 3-D plane tests with an entry that always passes.  The exact test alone decides hits.
#define FFX_BIN_MAX_TILES 16384 // per apex (128 x 128)
#define FFX_BIN_PAD 0.00390625f // 1/256 tile
#define FFX_BIN_FAR 512.0f      // projected coordinates beyond this many tiles are not trusted (float rounding would approach the pad)
struct __attribute__((aligned(16))) BinEntry {
  float bb[4];  // padded box of the projection: min x, min y, max x, max y (tile units); (-inf, -inf, inf, inf): always passes
  float e[9];   // three edge functions n.x * x + n.y * y + c >= 0 inside (offsets padded); a degenerate projection: (0, 0, 1)
  int32_t slot; // leaf slot of the triangle
  uint32_t pad[2];
};
static_assert(sizeof(BinEntry) == 64, "bin entry must be 64 bytes");
// tile coordinates of a world point p seen from the apex o: (M[0..2] . (p - o)) / Z, (M[3..5] . (p - o)) / Z with Z = M[6..8] . (p - o) > 0
struct BinGrid { float M[9]; float o[3]; int32_t nx, ny, on; };
struct BinHdr { uint32_t ok, total, cap, pad0 /* apex 0: the fill pass's arrival counter (BinBuild.arrive) */, env /* this pose's pre-pass wrote the grid's envelope */, pad[11]; };
static_assert(sizeof(BinHdr) == 64, "bin header must be 64 bytes");
__host__ __device__ static inline uint64_t ffx_bin_cap(int n_tris) { return (uint64_t)2 * (uint64_t)n_tris + FFX_BIN_MAX_TILES; } // (a few screen-filling triangles fit)
__host__ __device__ static inline uint64_t ffx_bin_off_starts() { return 64; }
__host__ __device__ static inline uint64_t ffx_bin_off_cursors() { return 64 + 4 * (uint64_t)(FFX_BIN_MAX_TILES + 16); }
__host__ __device__ static inline uint64_t ffx_bin_off_entries() { return (ffx_bin_off_cursors() + 4 * (uint64_t)FFX_BIN_MAX_TILES + 63) & ~(uint64_t)63; }
// Round 6: the ENVELOPE of an emitter's grid (k_bin_env, ffx_bins.hip; read by bins_shadow, ffx_trace.hip).  Every tile is cut into
// FFX_ENV_SUB x FFX_ENV_SUB cells (7: their 8 x 8 vertices are the 64 lanes of the wave that builds them) and every cell gets ONE plane through the emitter's space, N . (X - E) = 1, that lies in front of every
// triangle the tile lists over that cell (conservative: the cell's corner maxima of the triangles' own planes in 1 / depth, where planes are
// affine, then a plane above the bilinear patch of the four corners), pulled a little further forward than the ignored tail of a shadow ray
// (N carries the factor (1 - 10 eps)(1 + 6e-5)).  A shadow segment E -> Po with N . (Po - E) <= 1 ends in front of that plane: every
// triangle that could be hit at the segment's image point has its plane at ray parameter t >= (1 - 10 eps)(1 + 2e-5), which the exact test
// rejects — the packet skips the emitter's any-hit stage.  float4 per cell {N, 0}: N = 0 for a cell no listed triangle touches (nothing to
// hit), NaN where no proof is offered (a listed triangle seen edge-on, whose 1 / depth is ill-conditioned).  One row-major array over the
// grid's fine cells, [ny * SUB][nx * SUB], at the end of the apex's bins area.
#define FFX_ENV_SUB 7
__host__ __device__ static inline uint64_t ffx_bin_off_env(int n_tris) { return (ffx_bin_off_entries() + 64 * (ffx_bin_cap(n_tris) + 64) + 63) & ~(uint64_t)63; }
__host__ __device__ static inline uint64_t ffx_bin_stride(int n_tris) { return ffx_bin_off_env(n_tris) + 16 * (uint64_t)FFX_BIN_MAX_TILES * FFX_ENV_SUB * FFX_ENV_SUB; }
// what the render kernels need of the bins (part of their first kernel argument)
struct BinsK {
  const char *base[3]; // per apex: BinHdr, list starts, cursors, entries
  BinGrid g[3];
  float cam_inv_ts_x, cam_inv_ts_y; // camera pixels -> tile units
  int clear_on;                     // the emitters' "nothing can shadow this triangle" bits of the per-slot normals are valid (FFX_GN_CLEAR_BIT)
  int env_on;                       // mask over the emitters (bit 0 projector, bit 1 spot): the pre-pass built that grid's envelope
  uint32_t env_off;                 // ffx_bin_off_env(n_tris): the envelope's offset inside an apex's bins area
};
// Round 5: "clear" triangles.  The spot's any-hit stage was a quarter of the render kernel (K8 0.400 -> 0.302 ms without it, tools/k8ab.py) for an
// emitter next to the camera that hardly anything shadows.  The pre-pass (k_bin_clear, ffx_bins.hip) proves per triangle k and emitter E that
// NO other triangle can intersect a shadow segment from E to a lifted point of k — every triangle j listed in a tile of E's grid with k is
// (H0) apart from k in E's image plane, or (H1) wholly behind k's plane, or (H2) front-facing to E with k wholly in front of ITS plane (the
// concave neighbours of a tube seen from inside) — and leaves bit 27 + a (a = 1 projector, 2 spot) of the fourth word of the triangle's
// per-slot normal (ffx_bvh_info.off_gn) set.  A packet all of whose samples that need emitter a lie on such triangles skips the walk.
#define FFX_GN_SHAPE_MASK 0x0fffffff
#define FFX_GN_SMOOTH_BIT 0x40000000
#define FFX_GN_CLEAR_BIT(a) (1u << (27 + (a)))
// the pre-pass of a packet render on `s` (ffx_bins.hip): apex records + the bins of the enabled grids — three launches (count, scan,
// fill); with every grid off it is the apex records alone (one launch)
struct BinBuild { BinGrid g[3]; char *base[3]; uint32_t cap; uint32_t *arrive; /* [dev] one word, zero between builds: BinHdr.pad0 of apex 0 */
                  int env_mask; /* bit a - 1: k_bin_env will run for emitter a (k_bin_scan notes it in the grid's header) */ };
// the envelope launch's constants: Minv = inverse of BinGrid.M (direction of a tile-space point: d = Minv (x, y, 1) Z), graz = (largest |Minv (x, y, 1)|
// over the grid) / 40 — a listed triangle whose plane is steeper than 1 : 40 against a cell's rays poisons the cell —, kap = (1 - 10 eps)(1 + 6e-5)
struct EnvBuild { float Minv[FFX_N_APEX][9]; float graz[FFX_N_APEX]; int on[FFX_N_APEX]; uint32_t env_off; float kap; };
void ffx_bins_launch(const TriRec *recs, int n_tris, const BinBuild &bb, const void *apex_out, const float (*apex_o)[3], const int *apex_on, uint32_t astride,
                     uint32_t *cache_hdr, uint32_t cap_stray, hipStream_t s, int beside_lambert = 0, uint32_t *gn_words = nullptr, int clear_on = 0,
                     const EnvBuild *env = nullptr);

// top bit of the `cap_stray` argument of the launches that reset an adjoint cache's header (k_bin<false>, k_apex_records, k_cache_reset):
// FFX_RENDER_CACHE_KEEP_DROPPED — empty the arena, keep the `dropped` count of the step's earlier scene samples
#define FFX_CAP_KEEP_DROPPED 0x80000000u

// entry of the refit list (leaves-first by node height)
struct RefitEntry { int32_t node; };

#define FFX_STACK_DEPTH 48

int ffx_inv4(const float *m, float *out);  // double-precision cofactor inverse (host)
