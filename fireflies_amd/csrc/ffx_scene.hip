// ffx_scene.hip — K5 + K6: per-randomisation geometry update on the GPU.
//
// Replaces Mesh.get_randomized_vertices (fireflies/entity/mesh.py:158-165), Scene.update_meshes
// (fireflies/scene.py:243-251) and the acceleration-structure rebuild hidden in
// mitsuba_params.update() (fireflies/scene.py:384).
//
// Round 3: ONE launch (k_scene_update_fused).  The host builder cuts the tree into TREELETS (include/ffx.h: off_plan) —
// maximal subtrees of <= ~1024 triangles with a contiguous run of leaf slots.  A workgroup owns one treelet end to end:
// it builds its triangle records and per-triangle boxes, re-fits its nodes height by height behind workgroup barriers
// (one CU, one L1: workgroup-scope visibility is enough), copies the boxes of the wide overlay's children that live in
// its nodes, then publishes (agent-scope fence + one atomic).  The workgroup that arrives LAST re-fits the few hundred
// nodes above the treelets the same way.  The eight dependent launches this replaces (records, four levels, tail, wide
// boxes: 55-65 us of mostly launch gaps, 245 us when sharing the GPU with a render) remain below as the A/B baseline
// (FFX_REFIT=levels).
//
//   k_build_records : one lane per leaf slot.  Gathers the triangle's three source vertices
//                     (animation frame selected by vert_off[shape]), applies the shape's 4x4 and
//                     writes the 48-byte record {v0, e1, e2, prim, shape} in LEAF order, so that
//                     traversal reads a leaf's triangles as one contiguous run.  World-space
//                     vertices are never written back to HBM.
//   k_refit_level   : nodes grouped by height (leaves-first); a node's child boxes depend only
//                     on lower groups, so each group is one parallel pass.
//   k_refit_tail    : every group of <= 1024 nodes is handled by ONE workgroup that walks the
//                     remaining heights with a workgroup barrier in between (all waves of a
//                     workgroup share one CU's L1, so workgroup-scope visibility is enough);
//                     this replaces ~20 tiny dependent launches by one.
// HBM traffic per update: 12*V (gather, mostly L2 hits: each vertex is shared by ~6 triangles)
// + 48*F (records out) + 64*N_nodes (nodes in/out).
#include <stdlib.h>
#include <string.h>

#include "ffx_common.h"

#define UPD_BLOCK 256
#define TAIL_BLOCK 1024

// per-shape tables as a kernel argument (ffx_scene_update_h): 32 * (1 + 12) dwords = 1664 B of kernarg
struct ShapeTabH { int32_t off[FFX_MAX_SHAPES_H]; float m[FFX_MAX_SHAPES_H][12]; };
// ffx_smooth's two host tables as kernel arguments; `vn` / `nrec` NULL: no shape interpolates its normals
struct SmoothTab { int32_t on[FFX_MAX_SHAPES_H]; int32_t vbase[FFX_MAX_SHAPES_H]; const float *vn; float4 *nrec; float4 *gn; };
// unit geometric normal of a record, IEEE cross / sqrt / divide in the order of the oracle's shade_sample (ffx_bvh_info.off_gn)
// .w carries what else a hit needs of its record, as bits: 0 = degenerate triangle, else (shape + 1) | smooth << 30 — the render kernel then
// reads 16 bytes per hit sample instead of touching the 48-byte record as well (2.5 MB less working set on the vocal fold)
__device__ __forceinline__ float4 unit_normal_of(v3 e1, v3 e2, int shape, bool smooth) {
  v3 n = vcross(e1, e2);
  const float nl = sqrtf(vdot(n, n));
  if (!(nl > 0.f)) return make_float4(0.f, 0.f, 0.f, 0.f);
  const float inl = 1.0f / nl;
  return make_float4(n.x * inl, n.y * inl, n.z * inl, __int_as_float((shape + 1) | (smooth ? 0x40000000 : 0)));
}

// Vertex normals of the current pose (ffx.h ffx_smooth) [EXT Mitsuba mesh.cpp recompute_vertex_normals]: a lane per vertex
// row walks the corners incident to it in ascending triangle order — the oracle's face-major loop adds to a vertex in the
// same order, so the sums see the same operands (the results differ only where asinf / sqrtf differ in the last bit).
__device__ __forceinline__ float unit_angle_f(v3 a, v3 b) {
  const float dt = vdot(a, b);
  if (dt >= 0.f) {
    const v3 df = vsub(b, a);
    return 2.0f * asinf(fminf(0.5f * sqrtf(vdot(df, df)), 1.f));
  }
  const v3 sm = V3(a.x + b.x, a.y + b.y, a.z + b.z);
  return 3.14159265358979323846f - 2.0f * asinf(fminf(0.5f * sqrtf(vdot(sm, sm)), 1.f));
}
template <bool HOST_TAB>
__global__ void __launch_bounds__(UPD_BLOCK)
    k_vertex_normals(const int32_t *__restrict__ adj_start, const int32_t *__restrict__ adj, int n_vn, float *__restrict__ vn, const float *__restrict__ src_verts,
                     const int32_t *__restrict__ tris, const int32_t *__restrict__ tri_shape, const int32_t *__restrict__ vert_off,
                     const float *__restrict__ xform, int n_shapes, ShapeTabH tab) {
  const int row = blockIdx.x * UPD_BLOCK + threadIdx.x;
  if (row >= n_vn) return;
  float ax = 0.f, ay = 0.f, az = 0.f;
  const int b = adj_start[row], e = adj_start[row + 1];
  for (int i = b; i < e; ++i) {
    const int key = adj[i], t = key >> 2, c = key & 3;
    int sh = tri_shape[t];
    sh = min(max(sh, 0), n_shapes - 1);
    float mm[12];
    int base;
    if (HOST_TAB) {
#pragma unroll
      for (int j = 0; j < 12; ++j) mm[j] = tab.m[sh][j];
      base = tab.off[sh];
    } else {
      const float *m = xform + 16 * sh;
#pragma unroll
      for (int j = 0; j < 12; ++j) mm[j] = m[j];
      base = vert_off[sh];
    }
    v3 p[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float *sv = src_verts + 3 * ((size_t)base + tris[3 * t + k]);
      p[k] = xf_point(mm, V3(sv[0], sv[1], sv[2]));
    }
    v3 n = vcross(vsub(p[1], p[0]), vsub(p[2], p[0]));
    const float nl = sqrtf(vdot(n, n));
    if (!(nl > 0.f)) continue;
    n = V3(n.x / nl, n.y / nl, n.z / nl);
    const v3 pc = c == 0 ? p[0] : (c == 1 ? p[1] : p[2]), pa = c == 0 ? p[1] : (c == 1 ? p[2] : p[0]), pb = c == 0 ? p[2] : (c == 1 ? p[0] : p[1]);
    v3 d0 = vsub(pa, pc), d1 = vsub(pb, pc);
    const float l0 = sqrtf(vdot(d0, d0)), l1 = sqrtf(vdot(d1, d1));
    if (!(l0 > 0.f) || !(l1 > 0.f)) continue;
    d0 = V3(d0.x / l0, d0.y / l0, d0.z / l0);
    d1 = V3(d1.x / l1, d1.y / l1, d1.z / l1);
    const float w = unit_angle_f(d0, d1);
    ax = fmaf(n.x, w, ax); ay = fmaf(n.y, w, ay); az = fmaf(n.z, w, az);
  }
  const float l = sqrtf(fmaf(ax, ax, fmaf(ay, ay, az * az)));
  if (l > 0.f) { ax /= l; ay /= l; az /= l; }
  vn[3 * (size_t)row] = ax; vn[3 * (size_t)row + 1] = ay; vn[3 * (size_t)row + 2] = az;
}
// the record's flag and its three vertex normals (a lane that has just built the record of leaf slot k)
__device__ __forceinline__ float write_slot_normals(const SmoothTab &sm, int sh, int prim, int k, const int32_t *__restrict__ tris) {
  if (!sm.vn || !sm.on[sh]) return 0.f;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float *q = sm.vn + 3 * ((size_t)sm.vbase[sh] + tris[3 * prim + c]);
    sm.nrec[3 * (size_t)k + c] = make_float4(q[0], q[1], q[2], 0.f);
  }
  return 1.0f;
}

template <bool HOST_TAB>
__global__ void __launch_bounds__(UPD_BLOCK)
    k_build_records(const int32_t *__restrict__ order, TriRec *__restrict__ recs, int n_tris, const float *__restrict__ src_verts,
                    const int32_t *__restrict__ tris, const int32_t *__restrict__ tri_shape, const int32_t *__restrict__ vert_off,
                    const float *__restrict__ xform, int n_shapes, ShapeTabH tab, SmoothTab sm) {
  int k = blockIdx.x * UPD_BLOCK + threadIdx.x;
  if (k >= n_tris) return;
  int prim = order[k];
  int sh = tri_shape[prim];
  sh = min(max(sh, 0), n_shapes - 1); // host validated; clamp so a bad id can never fault
  float mm[12];
  int base;
  if (HOST_TAB) {
#pragma unroll
    for (int i = 0; i < 12; ++i) mm[i] = tab.m[sh][i];
    base = tab.off[sh];
  } else {
    const float *m = xform + 16 * sh;
#pragma unroll
    for (int i = 0; i < 12; ++i) mm[i] = m[i];
    base = vert_off[sh];
  }
  v3 p[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float *sv = src_verts + 3 * ((size_t)base + tris[3 * prim + c]);
    p[c] = xf_point(mm, V3(sv[0], sv[1], sv[2]));
  }
  v3 e1 = vsub(p[1], p[0]), e2 = vsub(p[2], p[0]);
  float4 *o = reinterpret_cast<float4 *>(recs + k);
  o[0] = make_float4(p[0].x, p[0].y, p[0].z, e1.x);
  o[1] = make_float4(e1.y, e1.z, e2.x, e2.y);
  const float sflag = write_slot_normals(sm, sh, prim, k, tris);
  o[2] = make_float4(e2.z, __int_as_float(prim), __int_as_float(sh), sflag);
  if (sm.gn) sm.gn[k] = unit_normal_of(e1, e2, sh, sflag != 0.f);
}

// conservative box of a leaf's triangles.  The intersection test works on (v0, e1, e2), whose
// corners v0+e1, v0+e2 are re-rounded here, so the box is widened by a few ulps.
__device__ __forceinline__ void leaf_box(const TriRec *__restrict__ recs, int32_t code, float lo[3], float hi[3]) {
  uint32_t lc = (uint32_t)~code;
  int first = (int)(lc >> 3), count = (int)(lc & 7u) + 1;
#pragma unroll
  for (int a = 0; a < 3; ++a) { lo[a] = INFINITY; hi[a] = -INFINITY; }
  for (int i = 0; i < count; ++i) {
    const TriRec &r = recs[first + i];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      float p0 = r.v0[a], p1 = r.v0[a] + r.e1[a], p2 = r.v0[a] + r.e2[a];
      float mn = fminf(p0, fminf(p1, p2)), mx = fmaxf(p0, fmaxf(p1, p2));
      float pad = 4e-7f * fmaxf(fabsf(mn), fabsf(mx));
      lo[a] = fminf(lo[a], mn - pad);
      hi[a] = fmaxf(hi[a], mx + pad);
    }
  }
}

__device__ __forceinline__ void child_box(const BvhNode *nodes, const TriRec *__restrict__ recs, int32_t c, float lo[3], float hi[3]) {
  if (c == FFX_EMPTY_CHILD) {
#pragma unroll
    for (int a = 0; a < 3; ++a) { lo[a] = INFINITY; hi[a] = -INFINITY; }
  } else if (c < 0) {
    leaf_box(recs, c, lo, hi);
  } else {
    const BvhNode &n = nodes[c];
#pragma unroll
    for (int a = 0; a < 3; ++a) { lo[a] = fminf(n.lo0[a], n.lo1[a]); hi[a] = fmaxf(n.hi0[a], n.hi1[a]); }
  }
}

__device__ __forceinline__ void refit_node(BvhNode *nodes, const TriRec *__restrict__ recs, int id) {
  BvhNode &n = nodes[id];
  float lo[3], hi[3];
  child_box(nodes, recs, n.c0, lo, hi);
#pragma unroll
  for (int a = 0; a < 3; ++a) { n.lo0[a] = lo[a]; n.hi0[a] = hi[a]; }
  child_box(nodes, recs, n.c1, lo, hi);
#pragma unroll
  for (int a = 0; a < 3; ++a) { n.lo1[a] = lo[a]; n.hi1[a] = hi[a]; }
}

__global__ void __launch_bounds__(UPD_BLOCK)
    k_refit_level(BvhNode *nodes, const TriRec *__restrict__ recs, const int32_t *__restrict__ refit, int begin, int end) {
  int i = begin + blockIdx.x * UPD_BLOCK + threadIdx.x;
  if (i >= end) return;
  refit_node(nodes, recs, refit[i]);
}

struct TailLevels { int n; int start[FFX_MAX_LEVELS + 1]; };

__global__ void __launch_bounds__(TAIL_BLOCK) k_refit_tail(BvhNode *nodes, const TriRec *__restrict__ recs, const int32_t *__restrict__ refit, TailLevels lv) {
  for (int l = 0; l < lv.n; ++l) {
    int i = lv.start[l] + threadIdx.x;
    if (i < lv.start[l + 1]) refit_node(nodes, recs, refit[i]);
    __threadfence_block();
    __syncthreads();
  }
}

// ---- 64-wide overlay (ffx_common.h): after the binary refit, every box the wave-packet kernels test is
// re-expressed on a 16-bit grid spanning the scene's bounding box of THIS pose (the root box the refit
// just produced): triangle boxes in leaf-slot order and the children of the wide inner nodes, each of which
// is the box of one binary node (wsrc: binary node * 2 + side).  q_lo = floor - 1, q_hi = ceil + 1: the
// extra cell (1.5e-5 of the scene extent) absorbs every rounding of the de-quantisation in the kernels.
__device__ __forceinline__ void grid_of_root(const BvhNode *__restrict__ nodes, float org[3], float step[3], float inv[3]) {
  const BvhNode &n = nodes[0];
  float ext[3], emax = 0.f;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float lo = fminf(n.lo0[a], n.lo1[a]), hi = fmaxf(n.hi0[a], n.hi1[a]);
    org[a] = lo;
    ext[a] = hi - lo;
    emax = fmaxf(emax, ext[a]);
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float e = fmaxf(ext[a], fmaxf(emax * 1e-6f, 1e-30f)); // a flat scene still gets a non-degenerate grid
    step[a] = e * (1.0f / 65533.0f);                            // cells 1 .. 65534 span the box: the slack cell never clips
    inv[a] = 65533.0f / e;
    org[a] -= step[a];
  }
}
__device__ __forceinline__ void quantise_box(const float lo[3], const float hi[3], const float org[3], const float inv[3], WideChild &c) {
#if FFX_WIDE_F32
  // the boxes as they are (leaf boxes carry the refit's relative padding; the walk's packet constants carry the
  // padding for its own roundings, make_widepk)
  c.lo[0] = lo[0]; c.lo[1] = lo[1]; c.lo[2] = lo[2];
  c.hi0 = hi[0]; c.hi12[0] = hi[1]; c.hi12[1] = hi[2];
#else
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float ql = floorf((lo[a] - org[a]) * inv[a]) - 1.0f, qh = ceilf((hi[a] - org[a]) * inv[a]) + 1.0f;
    c.q[a] = (uint16_t)fminf(fmaxf(ql, 0.f), 65535.f);
    c.q[3 + a] = (uint16_t)fminf(fmaxf(qh, 0.f), 65535.f);
  }
#endif
}
__global__ void __launch_bounds__(UPD_BLOCK)
    k_wide_quant(const BvhNode *__restrict__ nodes, const TriRec *__restrict__ recs, int n_tris, WideChild *__restrict__ tq, WideChild *__restrict__ wn,
                 const int32_t *__restrict__ wsrc, int n_wchild, WideHdr *__restrict__ hdr) {
  const int i = blockIdx.x * UPD_BLOCK + threadIdx.x;
  float org[3], step[3], inv[3];
#if FFX_WIDE_F32
#pragma unroll
  for (int a = 0; a < 3; ++a) { org[a] = 0.f; step[a] = 1.f; inv[a] = 1.f; }
#else
  grid_of_root(nodes, org, step, inv);
#endif
  if (i == 0) {
#pragma unroll
    for (int a = 0; a < 3; ++a) { hdr->org[a] = org[a]; hdr->step[a] = step[a]; }
  }
  if (i < n_tris) {
    float lo[3], hi[3];
    leaf_box(recs, ~(int32_t)((uint32_t)i << 3), lo, hi); // the padded box of the single triangle in slot i
    WideChild c;
    quantise_box(lo, hi, org, inv, c);
    c.ref = 0;
#if FFX_WIDE_F32
    c.pad = 0;
#endif
    tq[i] = c;
  } else if (i < n_tris + n_wchild) {
    const int k = i - n_tris;
    const int src = wsrc[k];
    if (src < 0) return;
    const BvhNode &n = nodes[src >> 1];
    WideChild c = wn[k];
    if (src & 1) quantise_box(n.lo1, n.hi1, org, inv, c);
    else quantise_box(n.lo0, n.hi0, org, inv, c);
    wn[k] = c;
  }
}

// ---- the fused update.  `recs` and `nodes` are written and read back inside this kernel: no __restrict__, no read-only loads.
// One wave per treelet (round 4): beside a render whose one-wave workgroups refill every slot at once, a four-wave workgroup waits for four
// free slots on one compute unit — the update's elapsed time was 128 us at 256 threads, 59 us at 64 (alone: 36 / ~60), renders/s +2.5 %
// (tools: bench.py under -DFUSED_BLOCK=64 / 128 / 256 builds and FFX_TREELET_TRIS).
#ifndef FUSED_BLOCK
#define FUSED_BLOCK 64
#endif
__device__ __forceinline__ void tri_wide_box(const float (&p0)[3], const float (&e1)[3], const float (&e2)[3], WideChild &c) {
  // the padded box of ONE triangle exactly as leaf_box forms it from the record (corners re-rounded as v0 + e)
  float lo[3], hi[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float q0 = p0[a], q1 = p0[a] + e1[a], q2 = p0[a] + e2[a];
    const float mn = fminf(q0, fminf(q1, q2)), mx = fmaxf(q0, fmaxf(q1, q2));
    const float pad = 4e-7f * fmaxf(fabsf(mn), fabsf(mx));
    lo[a] = mn - pad;
    hi[a] = mx + pad;
  }
  c.lo[0] = lo[0]; c.lo[1] = lo[1]; c.lo[2] = lo[2];
  c.hi0 = hi[0]; c.hi12[0] = hi[1]; c.hi12[1] = hi[2];
  c.ref = 0;
  c.pad = 0;
}

__device__ __forceinline__ void plan_levels(BvhNode *nodes, const TriRec *recs, const int32_t *__restrict__ plan, const int32_t *__restrict__ h) {
  const int lvl0 = h[2], n_l = h[3];
  for (int l = 0; l < n_l; ++l) {
    const int b = plan[lvl0 + l], e = plan[lvl0 + l + 1];
    for (int i = b + (int)threadIdx.x; i < e; i += FUSED_BLOCK) refit_node(nodes, recs, plan[i]);
    __threadfence_block();
    __syncthreads();
  }
}
__device__ __forceinline__ void plan_wide_children(const BvhNode *nodes, WideChild *wn, const int32_t *__restrict__ wsrc, const int32_t *__restrict__ plan,
                                                   const int32_t *__restrict__ h) {
  const int w0 = h[4], wc = h[5];
  for (int j = (int)threadIdx.x; j < wc; j += FUSED_BLOCK) {
    const int k = plan[w0 + j];
    const int src = wsrc[k];
    if (src < 0) continue;
    const BvhNode &n = nodes[src >> 1];
    // (only the box is rewritten: ref / pad are topology, written once by the host builder)
    const float *lo = (src & 1) ? n.lo1 : n.lo0, *hi = (src & 1) ? n.hi1 : n.hi0;
    WideChild &c = wn[k];
    c.lo[0] = lo[0]; c.lo[1] = lo[1]; c.lo[2] = lo[2];
    c.hi0 = hi[0]; c.hi12[0] = hi[1]; c.hi12[1] = hi[2];
  }
}

template <bool HOST_TAB>
__global__ void __launch_bounds__(FUSED_BLOCK)
    k_scene_update_fused(BvhNode *nodes, TriRec *recs, const int32_t *__restrict__ order, WideChild *tq, WideChild *wn, const int32_t *__restrict__ wsrc,
                         int32_t *plan, int n_treelets, int counter_at, const float *__restrict__ src_verts, const int32_t *__restrict__ tris,
                         const int32_t *__restrict__ tri_shape, const int32_t *__restrict__ vert_off, const float *__restrict__ xform, int n_shapes, ShapeTabH tab,
                         SmoothTab sm, int mode) {
  // mode 0: the whole update (the workgroup that arrives last re-fits the top); 1: the treelets only, 2: the top only — the same update as
  // two launches (FFX_REFIT=split).  Publishing needs an agent-scope release fence in every treelet's workgroup, and on this part that fence
  // writes back the workgroup's whole L2 — megabytes of fresh records; a kernel boundary does the same once.
  FFX_SIDE_PRIO();
  __shared__ int s_last;
  if (mode == 2) {
    const int32_t *ht = plan + 8 * n_treelets;
    plan_levels(nodes, recs, plan, ht);
    plan_wide_children(nodes, wn, wsrc, plan, ht);
    return;
  }
  const int32_t *h = plan + 8 * blockIdx.x;
  // ---- A: records and per-triangle boxes of this treelet's run of leaf slots
  const int s0 = h[0], sn = h[1];
  for (int k = s0 + (int)threadIdx.x; k < s0 + sn; k += FUSED_BLOCK) {
    const int prim = order[k];
    int sh = tri_shape[prim];
    sh = min(max(sh, 0), n_shapes - 1);
    float mm[12];
    int base;
    if (HOST_TAB) {
#pragma unroll
      for (int i = 0; i < 12; ++i) mm[i] = tab.m[sh][i];
      base = tab.off[sh];
    } else {
      const float *m = xform + 16 * sh;
#pragma unroll
      for (int i = 0; i < 12; ++i) mm[i] = m[i];
      base = vert_off[sh];
    }
    v3 p[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float *sv = src_verts + 3 * ((size_t)base + tris[3 * prim + c]);
      p[c] = xf_point(mm, V3(sv[0], sv[1], sv[2]));
    }
    const v3 e1 = vsub(p[1], p[0]), e2 = vsub(p[2], p[0]);
    float4 *o = reinterpret_cast<float4 *>(recs + k);
    o[0] = make_float4(p[0].x, p[0].y, p[0].z, e1.x);
    o[1] = make_float4(e1.y, e1.z, e2.x, e2.y);
    const float sflag = write_slot_normals(sm, sh, prim, k, tris);
    o[2] = make_float4(e2.z, __int_as_float(prim), __int_as_float(sh), sflag);
    if (sm.gn) sm.gn[k] = unit_normal_of(e1, e2, sh, sflag != 0.f);
    const float a0[3] = {p[0].x, p[0].y, p[0].z}, a1[3] = {e1.x, e1.y, e1.z}, a2[3] = {e2.x, e2.y, e2.z};
    WideChild c;
    tri_wide_box(a0, a1, a2, c);
    tq[k] = c;
  }
  __threadfence_block();
  __syncthreads();
  // ---- B, C: this treelet's nodes by height, then the wide children whose boxes live in them
  plan_levels(nodes, recs, plan, h);
  plan_wide_children(nodes, wn, wsrc, plan, h);
  if (mode == 1) return;
  // ---- D: publish; the last workgroup to arrive re-fits the top of the tree
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) s_last = (atomicAdd(plan + counter_at, 1) == n_treelets - 1);
  __syncthreads();
  if (!s_last) return;
  __threadfence();
  const int32_t *ht = plan + 8 * n_treelets;
  plan_levels(nodes, recs, plan, ht);
  plan_wide_children(nodes, wn, wsrc, plan, ht);
  if (threadIdx.x == 0) plan[counter_at] = 0; // ready for the next update of this blob (stream-ordered)
}

static int refit_split_enabled() { // FFX_REFIT=fused: the one-launch form (A/B); default: treelets + top as two launches
  const char *e = getenv("FFX_REFIT");
  return !(e && strcmp(e, "fused") == 0);
}
static int refit_fused_enabled() {
  const char *e = getenv("FFX_REFIT");
  return !(e && strcmp(e, "levels") == 0);
}

// top: 0 = the whole update; 1 = everything but the top of the tree (the treelets' records, boxes and nodes: all the pre-pass reads) — the caller owes
// ffx_scene_refit_top before anything walks the tree; 2 = the top alone.  (1 and 2 only with the fused kernel's two-launch form, the default; every
// other configuration does the whole update at 0 and 1 and nothing at 2: the top is then already in place.)
static int scene_update_impl(void *bvh, const ffx_bvh_info *info, const float *src_verts, const int32_t *tris, const int32_t *tri_shape,
                             const int32_t *vert_off, const float *xform, int n_shapes, const ffx_smooth *smooth, ffx_stream s, bool host_tab, int top = 0) {
  if (top == 2) {
    if (!bvh || !info) FFX_FAIL(FFX_ERR_ARG, "scene_refit_top: bad argument");
#if FFX_WIDE_F32
    if (refit_fused_enabled() && refit_split_enabled() && info->off_plan != 0 && info->n_treelets > 0 && info->off_tq != 0) {
      char *b = (char *)bvh;
      SmoothTab sm0;
      memset(&sm0, 0, sizeof sm0);
      ShapeTabH tab0;
      hipLaunchKernelGGL(k_scene_update_fused<true>, dim3(1), dim3(FUSED_BLOCK), 0, (hipStream_t)s, (BvhNode *)(b + info->off_nodes), (TriRec *)(b + info->off_recs),
                         (const int32_t *)(b + info->off_order), (WideChild *)(b + info->off_tq), (WideChild *)(b + info->off_wnodes), (const int32_t *)(b + info->off_wsrc),
                         (int32_t *)(b + info->off_plan), info->n_treelets, info->plan_ints - 1, (const float *)nullptr, (const int32_t *)nullptr, (const int32_t *)nullptr,
                         (const int32_t *)nullptr, (const float *)nullptr, 1, tab0, sm0, 2);
      FFX_CHECK_LAUNCH("scene_refit_top");
    }
#endif
    return FFX_OK;
  }
  if (!bvh || !info || !src_verts || !tris || !tri_shape || !vert_off || !xform || n_shapes < 1) FFX_FAIL(FFX_ERR_ARG, "scene_update: bad argument");
  if (host_tab && n_shapes > FFX_MAX_SHAPES_H) FFX_FAIL(FFX_ERR_UNSUPPORTED, "scene_update_h: more than %d shapes", FFX_MAX_SHAPES_H);
  if (info->n_tris < 1 || info->n_nodes < 1 || info->n_levels < 1 || info->n_levels > FFX_MAX_LEVELS)
    FFX_FAIL(FFX_ERR_ARG, "scene_update: bad bvh info");
  if (info->level_start[info->n_levels] != info->n_nodes) FFX_FAIL(FFX_ERR_ARG, "scene_update: refit list does not cover every node");
  hipStream_t st = (hipStream_t)s;
  char *base = (char *)bvh;
  BvhNode *nodes = (BvhNode *)(base + info->off_nodes);
  const int32_t *order = (const int32_t *)(base + info->off_order);
  const int32_t *refit = (const int32_t *)(base + info->off_refit);
  TriRec *recs = (TriRec *)(base + info->off_recs);

  ShapeTabH tab;
  if (host_tab) {
    for (int i = 0; i < n_shapes; ++i) {
      tab.off[i] = vert_off[i];
      for (int j = 0; j < 12; ++j) tab.m[i][j] = xform[16 * i + j];
    }
  }
  // interpolated shading normals (ffx_smooth): the vertex normals of this pose first (one launch, only when a shape asks)
  SmoothTab sm;
  memset(&sm, 0, sizeof sm);
  if (info->off_gn != 0 && info->off_gn + 16ull * (uint64_t)info->n_tris <= info->total_bytes) sm.gn = (float4 *)(base + info->off_gn);
  if (smooth) {
    if (!smooth->shape_smooth || !smooth->shape_vbase || !smooth->adj_start || !smooth->adj || !smooth->vnormals || smooth->n_vn < 1)
      FFX_FAIL(FFX_ERR_ARG, "scene_update: bad ffx_smooth");
    if (n_shapes > FFX_MAX_SHAPES_H) FFX_FAIL(FFX_ERR_UNSUPPORTED, "scene_update: ffx_smooth with more than %d shapes", FFX_MAX_SHAPES_H);
    if (info->off_nrec == 0 || info->off_nrec + 48ull * (uint64_t)info->n_tris > info->total_bytes) FFX_FAIL(FFX_ERR_ARG, "scene_update: blob without a normal area");
    bool any = false;
    for (int i = 0; i < n_shapes; ++i) { sm.on[i] = smooth->shape_smooth[i] != 0; sm.vbase[i] = smooth->shape_vbase[i]; any |= sm.on[i] != 0; }
    if (any) {
      sm.vn = smooth->vnormals;
      sm.nrec = (float4 *)(base + info->off_nrec);
      if (host_tab)
        hipLaunchKernelGGL(k_vertex_normals<true>, dim3(ffx_cdiv(smooth->n_vn, UPD_BLOCK)), dim3(UPD_BLOCK), 0, st, smooth->adj_start, smooth->adj, smooth->n_vn,
                           smooth->vnormals, src_verts, tris, tri_shape, (const int32_t *)nullptr, (const float *)nullptr, n_shapes, tab);
      else
        hipLaunchKernelGGL(k_vertex_normals<false>, dim3(ffx_cdiv(smooth->n_vn, UPD_BLOCK)), dim3(UPD_BLOCK), 0, st, smooth->adj_start, smooth->adj, smooth->n_vn,
                           smooth->vnormals, src_verts, tris, tri_shape, vert_off, xform, n_shapes, tab);
      FFX_CHECK_LAUNCH("scene_update/vertex_normals");
    }
  }
#if FFX_WIDE_F32
  if (info->off_plan != 0 && info->n_treelets > 0 && info->off_tq != 0 && refit_fused_enabled()) {
    if (info->plan_ints < 8 * (info->n_treelets + 1) + 1 || info->off_plan + 4ull * (uint64_t)info->plan_ints > info->total_bytes)
      FFX_FAIL(FFX_ERR_ARG, "scene_update: bad refit plan");
    WideChild *tq = (WideChild *)(base + info->off_tq), *wn = (WideChild *)(base + info->off_wnodes);
    const int32_t *wsrc = (const int32_t *)(base + info->off_wsrc);
    int32_t *plan = (int32_t *)(base + info->off_plan);
    const int split = refit_split_enabled();
    for (int pass = 0; pass < (split ? (top == 1 ? 1 : 2) : 1); ++pass) {
      const int mode = split ? pass + 1 : 0;
      const dim3 grid(mode == 2 ? 1 : info->n_treelets);
      if (host_tab)
        hipLaunchKernelGGL(k_scene_update_fused<true>, grid, dim3(FUSED_BLOCK), 0, st, nodes, recs, order, tq, wn, wsrc, plan, info->n_treelets,
                           info->plan_ints - 1, src_verts, tris, tri_shape, (const int32_t *)nullptr, (const float *)nullptr, n_shapes, tab, sm, mode);
      else
        hipLaunchKernelGGL(k_scene_update_fused<false>, grid, dim3(FUSED_BLOCK), 0, st, nodes, recs, order, tq, wn, wsrc, plan, info->n_treelets,
                           info->plan_ints - 1, src_verts, tris, tri_shape, vert_off, xform, n_shapes, tab, sm, mode);
    }
    FFX_CHECK_LAUNCH("scene_update/fused");
    return FFX_OK;
  }
#endif
  if (host_tab) {
    hipLaunchKernelGGL(k_build_records<true>, dim3(ffx_cdiv(info->n_tris, UPD_BLOCK)), dim3(UPD_BLOCK), 0, st, order, recs, info->n_tris, src_verts, tris,
                       tri_shape, (const int32_t *)nullptr, (const float *)nullptr, n_shapes, tab, sm);
  } else {
    hipLaunchKernelGGL(k_build_records<false>, dim3(ffx_cdiv(info->n_tris, UPD_BLOCK)), dim3(UPD_BLOCK), 0, st, order, recs, info->n_tris, src_verts, tris,
                       tri_shape, vert_off, xform, n_shapes, tab, sm);
  }
  FFX_CHECK_LAUNCH("scene_update/build_records");

  int l = 0;
  for (; l < info->n_levels; ++l) {
    int cnt = info->level_start[l + 1] - info->level_start[l];
    // once this and every later group fit one workgroup, hand the rest to the tail kernel
    bool tail_ok = true;
    for (int m = l; m < info->n_levels; ++m)
      if (info->level_start[m + 1] - info->level_start[m] > TAIL_BLOCK) { tail_ok = false; break; }
    if (tail_ok) break;
    hipLaunchKernelGGL(k_refit_level, dim3(ffx_cdiv(cnt, UPD_BLOCK)), dim3(UPD_BLOCK), 0, st, nodes, recs, refit, info->level_start[l],
                       info->level_start[l + 1]);
    FFX_CHECK_LAUNCH("scene_update/refit_level");
  }
  if (l < info->n_levels) {
    TailLevels lv;
    lv.n = info->n_levels - l;
    for (int m = 0; m <= lv.n; ++m) lv.start[m] = info->level_start[l + m];
    hipLaunchKernelGGL(k_refit_tail, dim3(1), dim3(TAIL_BLOCK), 0, st, nodes, recs, refit, lv);
    FFX_CHECK_LAUNCH("scene_update/refit_tail");
  }
  if (info->off_tq != 0) { // the 64-wide overlay of the wave-packet kernels
    const int n_wchild = info->n_wide * FFX_WIDE;
    hipLaunchKernelGGL(k_wide_quant, dim3(ffx_cdiv((long)info->n_tris + n_wchild, UPD_BLOCK)), dim3(UPD_BLOCK), 0, st, nodes, recs, info->n_tris,
                       (WideChild *)(base + info->off_tq), (WideChild *)(base + info->off_wnodes), (const int32_t *)(base + info->off_wsrc), n_wchild,
                       (WideHdr *)(base + info->off_whdr));
    FFX_CHECK_LAUNCH("scene_update/wide_quant");
  }
  return FFX_OK;
}

extern "C" int ffx_scene_update(void *bvh, const ffx_bvh_info *info, const float *src_verts, const int32_t *tris, const int32_t *tri_shape,
                                const int32_t *vert_off, const float *xform, int n_shapes, const ffx_smooth *smooth, ffx_stream s) {
  return scene_update_impl(bvh, info, src_verts, tris, tri_shape, vert_off, xform, n_shapes, smooth, s, false);
}

// (ffx_rng.cpp, ffx_scene_step_h: the whole update, or — top = 1 — without the top of the tree; not part of the C ABI)
int ffx_scene_update_h_top(void *bvh, const ffx_bvh_info *info, const float *src_verts, const int32_t *tris, const int32_t *tri_shape, const int32_t *vert_off,
                           const float *xform, int n_shapes, const ffx_smooth *smooth, int top, ffx_stream s) {
  return scene_update_impl(bvh, info, src_verts, tris, tri_shape, vert_off, xform, n_shapes, smooth, s, true, top);
}
extern "C" int ffx_scene_refit_top(void *bvh, const ffx_bvh_info *info, ffx_stream s) {
  return scene_update_impl(bvh, info, nullptr, nullptr, nullptr, nullptr, nullptr, 1, nullptr, s, true, 2);
}
extern "C" int ffx_scene_update_h(void *bvh, const ffx_bvh_info *info, const float *src_verts, const int32_t *tris, const int32_t *tri_shape,
                                  const int32_t *vert_off, const float *xform, int n_shapes, const ffx_smooth *smooth, ffx_stream s) {
  return scene_update_impl(bvh, info, src_verts, tris, tri_shape, vert_off, xform, n_shapes, smooth, s, true);
}
