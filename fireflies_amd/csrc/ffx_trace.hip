// ffx_trace.hip — K7 (primary visibility), K8 (render), K9 (render adjoint) for gfx950.
//
// Replaces sensor.sample_ray + scene.ray_intersect (fireflies/graphics/depth.py:35-46,72-84,
// 110-125,152-165) and mi.render / its Dr.Jit backward (examples/vocalfold_scene.py:69,102;
// main.py:156) — all Mitsuba [EXT].  The arithmetic of ray generation, the triangle test and
// shading follows the same documented operation order as oracle/ffx_oracle.c, so
// depths and ids agree with the oracle bit for bit apart from rare 1-ulp boundary cases.
//
// Execution model.  The render kernels and K7 are WAVE-PACKET kernels: one wavefront = the 64
// samples of one pixel (or a compact pixel block at low spp), wave-uniform control flow.  Their default
// traversal is the 64-wide walk (lanes test 64 child boxes of a wide node against the packet as a whole; exact
// triangle tests with the lanes back on the rays), falling back to the binary packet walk (scalar-cache node
// fetch, per-ray box tests with the node as SGPR operands, VGPR lane stack) for packets the interval test is bad
// at.  The per-lane kernels further down (one lane = one ray, LDS stack) serve arbitrary rays (k_trace_rays)
// and remain as the A/B baseline (FFX_TRAVERSAL=lane).
// XCD placement: giving each XCD a contiguous band of the image (FFX_XCD_REMAP=1) costs 30 % here: tile cost varies
// by 10x across the image, so whole XCDs idle while the one that owns the vocal folds finishes.  Default
// (FFX_XCD_REMAP=128): within every 1024 consecutive one-wave workgroups each XCD walks 128 CONSECUTIVE ones — a
// 16x8-pixel patch — so the four waves of a 2x2-pixel tile and their neighbours share an L2: as fast as the plain
// round-robin deal (+0..2 %), a third of its HBM traffic (FETCH 37 -> 13.5 MB, WRITE 10.2 -> 3.1 MB per launch).
#include <stdlib.h>
#include <string.h>

#include <limits.h>

#include "ffx_common.h"

#define TR_BLOCK 256
#define RAY_EPS 8.940696716308594e-05f
#define SHADOW_EPS (10.0f * RAY_EPS)

struct CamK {
  float s2c[16];
  float tw[12]; // rows 0..2 of to_world
  float near_clip, far_clip, inv_w, inv_h;
  int W, H;
  // a perspective sensor's sample_to_camera has m12 = m13 = 0: the homogeneous w of a near-plane point is the constant m15 and
  // 1 / w is formed once on the host (the same IEEE quotient the per-sample division gave: identical rays).  Measured with
  // tools/k8ab.py: within the noise (the ten VALU it saves are 0.7 % of a pixel's) — kept because it is free.
  int w_uniform;
  float iw_u;
};

// the gaussian reconstruction filter's constants (ffx_scene_desc.rfilter): g(x) = max(0, exp(alpha x^2) - bias).  A sample needs the five weights
// of its window along each axis, at x0 + a, a = 0..4: instead of five exponentials, exp(alpha (x + 1)^2) = exp(alpha x^2) exp(2 alpha x) exp(alpha)
// — two exponentials and eight multiplies with k[n] = exp(alpha (2 n + 1)) from the host (round 5; `rec`: |alpha| <= 8, i.e. stddev >= 0.25:
// the factor exp(2 alpha x0) stays inside float's range — narrower filters keep the five exponentials).  Every kernel that forms weights
// calls rf_weights, so forward, adjoint and cache agree bit for bit; the oracle's five expf differ in the last bits (1e-6 of a weight).
struct RfC { float alpha, bias, aL, aL2, k[4]; int rec; };
#define FFX_MAT_PRE 12 // floats per pre-row: eta, 1/eta^2, a2, 1/a2, metallic, c_sw, c_fd, brdf, 2 roughness, lobe flags (bits), roughness^2, tint term
#define FFX_PRE_ANISO 1u
#define FFX_PRE_TINT 2u
#define FFX_PRE_CLEARCOAT 4u
#define FFX_PRE_FLAT 8u
#define FFX_PRE_SHEEN 16u
struct ShadeK {
  CamK cam;
  int proj_on, spot_on, shadows;
  float p_w2l[12];
  float p_c2s[16];
  float p_pos[3], p_axis[3], p_color[3];
  float p_scale;
  int tw, th, tc;
  float s_w2l[12];
  float s_pos[3], s_int[3];
  float cos_cut, cos_beam, cutoff, inv_trans;
  int s_rigid; // the 3x3 part of s_w2l is a rotation (orthonormal rows to 2e-6): the packet kernels take the cosine to the spot's axis from its third row alone
  int mat_stride;    // floats per row of the material table: 3 (Lambert albedo) or FFX_MAT_STRIDE
  const float *mats; // the material table (the render calls' shape_albedo)
  // texture-valued base colours (ffx_scene_desc.base_tex): rows select one with FFX_MAT_BASE_TEX
  int n_base_tex;
  int btw[FFX_MAX_BASE_TEX], bth[FFX_MAX_BASE_TEX];
  const float *btex[FFX_MAX_BASE_TEX];
  const float *slot_uv;
  // the material table as a kernel argument (ffx_scene_desc.mat_h): the kernels then read the rows from their own kernarg segment
  int mat_inline;
  float mat_h[FFX_MAX_MAT_H];
  // per-row constants of the inline material rows (round 5; shade_prepare -> mat_pre_row): what every BSDF evaluation of a row re-derived from its
  // parameters — alpha^2 and its reciprocal, 1 / eta^2, the Fresnel mix's coefficients, the diffuse weight, which optional lobes are on
  int mat_pre_on;
  float mat_pre[8 * FFX_MAT_PRE];
  // tile bins of the three apexes (ffx_common.h BinsK): the packet kernels try them before the tree walks
  BinsK bins;
  // reconstruction filter (ffx_scene_desc.rfilter = gaussian; the *_filtered entry points only): g(x) = max(0, exp(rf_alpha x^2) - rf_bias)
  RfC rf;
};
// The first kernel argument, read in place.  The scene constants (ShadeK, ~100 dwords + the inline material rows) are the first
// argument of the render kernels.  Read through the by-value copy the compiler loads them all up front and, out of SGPRs, parks
// them in VGPR lanes (v_writelane / v_readlane: ~480 spill instructions on the VALU, the unit that bounds these kernels); through
// this pointer — which the compiler cannot see through — each phase re-reads what it needs with scalar loads from the kernarg
// segment (no VALU work at all), and per-lane indexed tables (the material rows) are ordinary global loads from it.
template <typename T>
__device__ __forceinline__ const T &kernarg_first() {
  const __attribute__((address_space(4))) char *p = (const __attribute__((address_space(4))) char *)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(p));
  return *(const T *)p;
}
__device__ __forceinline__ const ShadeK &kernarg_shade() { return kernarg_first<ShadeK>(); }
// the material table of a launch: its own kernel arguments, or the caller's device array
__device__ __forceinline__ const float *mat_table(const ShadeK &k) { return k.mat_inline ? k.mat_h : k.mats; }

struct Hit { float t; int prim, shape, slot; };

// ------------------------------------------------------------------------------------------ traversal
// Ray/box slab test, t = (plane - o) * (1/d): exact cancellation at the origin, so no absolute pad is
// needed; the far side is widened by 2 ulp.  A direction component of (nearly) zero is replaced by
// +-1e-20 FOR THE BOX TEST ONLY: 1/d stays finite, planes the origin lies between give t = -/+ huge
// (axis imposes no constraint) and planes it lies outside of give same-signed huge values (box
// rejected).  With 1/0 = inf the products 0*inf = NaN would silently drop the axis and such rays
// (every pixel-corner ray of the centre row / column when jitter is off) would walk the whole tree.
struct RayBox { v3 o, id; };
__device__ __forceinline__ float safe_rcp_dir(float d) {
  float a = fabsf(d) < 1e-20f ? copysignf(1e-20f, d) : d;
  return __builtin_amdgcn_rcpf(a);
}
__device__ __forceinline__ RayBox make_raybox(v3 o, v3 d) {
  RayBox r;
  r.o = o;
  r.id = V3(safe_rcp_dir(d.x), safe_rcp_dir(d.y), safe_rcp_dir(d.z));
  return r;
}
__device__ __forceinline__ bool slab(const float lo[3], const float hi[3], const RayBox &rb, float tmin, float tmax, float &tn_out) {
  float ax = (lo[0] - rb.o.x) * rb.id.x, bx = (hi[0] - rb.o.x) * rb.id.x;
  float ay = (lo[1] - rb.o.y) * rb.id.y, by = (hi[1] - rb.o.y) * rb.id.y;
  float az = (lo[2] - rb.o.z) * rb.id.z, bz = (hi[2] - rb.o.z) * rb.id.z;
  float tn = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fmaxf(fminf(az, bz), tmin));
  float tf = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz)) * 1.0000004f;
  tf = fminf(tf, tmax);
  tn_out = tn;
  return tn <= tf;
}

// Moller-Trumbore, division-free rejection; identical order to the oracle.
__device__ __forceinline__ bool tri_hit(const TriRec *__restrict__ rec, v3 o, v3 d, float tmin, float &t_out, int &prim, int &shape) {
  const float4 *r4 = reinterpret_cast<const float4 *>(rec);
  float4 a = r4[0], b = r4[1], c = r4[2];
  v3 v0 = V3(a.x, a.y, a.z), e1 = V3(a.w, b.x, b.y), e2 = V3(b.z, b.w, c.x);
  v3 pv = vcross(d, e2);
  float det = vdot(e1, pv);
  v3 tv = vsub(o, v0);
  v3 qv = vcross(tv, e1);
  float U = vdot(tv, pv), Vv = vdot(d, qv), T = vdot(e2, qv);
  if (det < 0.f) { det = -det; U = -U; Vv = -Vv; T = -T; }
  if (!(det > 0.f)) return false;
  if (!(U >= 0.f) || !(Vv >= 0.f) || !(U + Vv <= det)) return false;
  float t = T / det;
  if (!(t > tmin)) return false;
  t_out = t;
  prim = __float_as_int(c.y);
  shape = __float_as_int(c.z);
  return true;
}

// The same test for rays that share their origin with a whole batch (primary rays: the camera; shadow
// rays: the emitter they are traced from): Moller-Trumbore's scalars as three dot products with the
// triangle's APEX vectors (ffx_common.h).  The apex vectors are formed here in the same order as in
// k_apex_records and the oracle, so all three produce identical bits.
struct ApexVec { v3 A, B, C; float T; };
__device__ __forceinline__ ApexVec apex_vectors(v3 v0, v3 e1, v3 e2, v3 o) {
  ApexVec a;
  a.A = vcross(e2, e1);
  const v3 tv = vsub(o, v0);
  a.B = vcross(e2, tv);
  a.C = vcross(tv, e1);
  a.T = vdot(e2, a.C);
  return a;
}
__device__ __forceinline__ bool tri_hit_apex(const TriRec *__restrict__ rec, v3 o, v3 d, float tmin, float &t_out, int &prim, int &shape) {
  const float4 *r4 = reinterpret_cast<const float4 *>(rec);
  float4 a = r4[0], b = r4[1], c = r4[2];
  const ApexVec av = apex_vectors(V3(a.x, a.y, a.z), V3(a.w, b.x, b.y), V3(b.z, b.w, c.x), o);
  float det = vdot(d, av.A), U = vdot(d, av.B), Vv = vdot(d, av.C), T = av.T;
  if (det < 0.f) { det = -det; U = -U; Vv = -Vv; T = -T; }
  if (!(det > 0.f)) return false;
  if (!(U >= 0.f) || !(Vv >= 0.f) || !(U + Vv <= det)) return false;
  float t = T / det;
  if (!(t > tmin)) return false;
  t_out = t;
  prim = __float_as_int(c.y);
  shape = __float_as_int(c.z);
  return true;
}

// ANY = false: closest hit in (tmin, tmax], ties broken by the smaller primitive id.
// ANY = true : returns true as soon as a hit with tmin < t < tmax is found.
// `stack` points at this lane's first slot; consecutive entries are `stride` ints apart.
// APEX: the ray origin is a batch-wide apex (see tri_hit_apex); false for arbitrary rays.
template <bool ANY, bool APEX>
__device__ __forceinline__ bool traverse(const BvhNode *__restrict__ nodes, const TriRec *__restrict__ recs, v3 o, v3 d, float tmin, float tmax, Hit &h,
                                         int *stack, int stride) {
  h.t = tmax;
  h.prim = -1;
  h.shape = -1;
  h.slot = -1;
  const RayBox rb = make_raybox(o, d);
  int sp = 0;
  int cur = 0;
  while (true) {
    const float4 *n4 = reinterpret_cast<const float4 *>(nodes + cur);
    float4 q0 = n4[0], q1 = n4[1], q2 = n4[2];
    int4 ch = *reinterpret_cast<const int4 *>(n4 + 3);
    float lo0[3] = {q0.x, q0.y, q0.z}, hi0[3] = {q0.w, q1.x, q1.y};
    float lo1[3] = {q1.z, q1.w, q2.x}, hi1[3] = {q2.y, q2.z, q2.w};
    float t0, t1;
    bool h0 = (ch.x != FFX_EMPTY_CHILD) && slab(lo0, hi0, rb, tmin, h.t, t0);
    bool h1 = (ch.y != FFX_EMPTY_CHILD) && slab(lo1, hi1, rb, tmin, h.t, t1);
#pragma unroll
    for (int side = 0; side < 2; ++side) {
      int c = side ? ch.y : ch.x;
      bool hs = side ? h1 : h0;
      if (hs && c < 0) {
        uint32_t lc = (uint32_t)~c;
        int first = (int)(lc >> 3), count = (int)(lc & 7u) + 1;
        for (int i = 0; i < count; ++i) {
          float t;
          int prim, shape;
          if (APEX ? tri_hit_apex(recs + first + i, o, d, tmin, t, prim, shape) : tri_hit(recs + first + i, o, d, tmin, t, prim, shape)) {
            if (ANY) {
              if (t < tmax) return true;
            } else if (t <= tmax && (h.prim < 0 || t < h.t || (t == h.t && prim < h.prim))) {
              h.t = t; h.prim = prim; h.shape = shape; h.slot = first + i;
            }
          }
        }
        if (side) h1 = false; else h0 = false;
      }
    }
    // a leaf may have shortened the ray: drop inner children that now start behind the hit
    if (h0 && t0 > h.t) h0 = false;
    if (h1 && t1 > h.t) h1 = false;
    if (h0 && h1) {
      int nearc = ch.x, farc = ch.y;
      if (t1 < t0) { nearc = ch.y; farc = ch.x; }
      stack[sp * stride] = farc;
      ++sp;
      cur = nearc;
    } else if (h0) {
      cur = ch.x;
    } else if (h1) {
      cur = ch.y;
    } else {
      if (sp == 0) break;
      --sp;
      cur = stack[sp * stride];
    }
  }
  return h.prim >= 0;
}

// ------------------------------------------------------------------------------------------ camera
__device__ __forceinline__ void sample_jitter(uint32_t seed_key, uint32_t idx, float &jx, float &jy) {
  uint32_t a = hash32((2u * idx) ^ seed_key), b = hash32((2u * idx + 1u) ^ seed_key);
  jx = (float)(a >> 8) * (1.0f / 16777216.0f);
  jy = (float)(b >> 8) * (1.0f / 16777216.0f);
}

__device__ __forceinline__ void cam_ray(const CamK &k, float sx, float sy, v3 &o, v3 &d, float &near_t, float &far_t) {
  const float *m = k.s2c;
  float qx = fmaf(m[0], sx, fmaf(m[1], sy, m[3]));
  float qy = fmaf(m[4], sx, fmaf(m[5], sy, m[7]));
  float qz = fmaf(m[8], sx, fmaf(m[9], sy, m[11]));
  // one IEEE reciprocal + multiplies instead of a division per component (same order as the oracle)
  float iw;
  if (k.w_uniform) iw = k.iw_u; // (wave-uniform branch)
  else iw = 1.0f / fmaf(m[12], sx, fmaf(m[13], sy, m[15]));
  v3 np = V3(qx * iw, qy * iw, qz * iw);
  const float il = 1.0f / sqrtf(vdot(np, np));
  v3 dl = V3(np.x * il, np.y * il, np.z * il);
  d = xf_dir(k.tw, dl);
  o = V3(k.tw[3], k.tw[7], k.tw[11]);
  const float idz = 1.0f / dl.z;
  near_t = k.near_clip * idz;
  far_t = k.far_clip * idz;
}

// ------------------------------------------------------------------------------------------ K7 kernels
__global__ void __launch_bounds__(TR_BLOCK)
    k_trace_primary(CamK cam, const BvhNode *__restrict__ nodes, const TriRec *__restrict__ recs, int spp, int jitter, uint32_t seed_key, long total,
                    float *__restrict__ t_out, int32_t *__restrict__ shape_out, int32_t *__restrict__ prim_out) {
  extern __shared__ int s_stack[];
  long idx = (long)blockIdx.x * TR_BLOCK + threadIdx.x;
  if (idx >= total) return;
  long pix = idx / spp;
  int x = (int)(pix % cam.W), y = (int)(pix / cam.W);
  float jx = 0.f, jy = 0.f;
  if (jitter) sample_jitter(seed_key, (uint32_t)idx, jx, jy);
  v3 o, d;
  float nt, ft;
  cam_ray(cam, ((float)x + jx) * cam.inv_w, ((float)y + jy) * cam.inv_h, o, d, nt, ft);
  Hit h;
  bool hit = traverse<false, true>(nodes, recs, o, d, nt, ft, h, s_stack + threadIdx.x, TR_BLOCK);
  t_out[idx] = hit ? (h.t - nt) : 0.f;
  if (shape_out) shape_out[idx] = h.shape;
  if (prim_out) prim_out[idx] = h.prim;
}

__global__ void __launch_bounds__(TR_BLOCK)
    k_trace_rays(const BvhNode *__restrict__ nodes, const TriRec *__restrict__ recs, const float *__restrict__ org, const float *__restrict__ dir, int n,
                 float tmax, float *__restrict__ t_out, int32_t *__restrict__ shape_out, int32_t *__restrict__ prim_out) {
  extern __shared__ int s_stack[];
  int i = blockIdx.x * TR_BLOCK + threadIdx.x;
  if (i >= n) return;
  Hit h;
  bool hit = traverse<false, false>(nodes, recs, V3(org[3 * i], org[3 * i + 1], org[3 * i + 2]), V3(dir[3 * i], dir[3 * i + 1], dir[3 * i + 2]), 0.f, tmax, h,
                                    s_stack + threadIdx.x, TR_BLOCK);
  t_out[i] = hit ? h.t : 0.f;
  if (shape_out) shape_out[i] = h.shape;
  if (prim_out) prim_out[i] = h.prim;
}

// ------------------------------------------------------------------------------------------ shading
#define FFX_PI_F 3.14159265358979323846f
// Reciprocal, quotient and square root of the shading terms: the hardware seed (v_rcp / v_sqrt / v_rsq,
// 1 ulp) plus one fma Newton step.  The result equals the correctly rounded IEEE value except in rare
// last-bit cases (Markstein), at 3 / 5 / 7 VALU instructions instead of the 10 / 10 / 14 of the
// compiler's IEEE expansions (which also cover denormal and overflow ranges that cannot occur here:
// arguments are lengths and depths of visible surface points).  Ray generation and the triangle test
// keep the IEEE forms: they decide WHICH primitive is hit.
#ifdef FFX_EXP_SEED_LIGHT // timing experiment: the light terms on the bare hardware seeds (1 ulp), like the BSDF
__device__ __forceinline__ float rcp_nr(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float div_nr(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
__device__ __forceinline__ float sqrt_nr(float x) { return __builtin_amdgcn_sqrtf(x); }
#else
__device__ __forceinline__ float rcp_nr(float x) {
  const float r = __builtin_amdgcn_rcpf(x);
  return fmaf(r, fmaf(-x, r, 1.0f), r);
}
__device__ __forceinline__ float div_nr(float a, float b) {
  const float r = rcp_nr(b);
  const float q = a * r;
  return fmaf(fmaf(-b, q, a), r, q);
}
__device__ __forceinline__ float sqrt_nr(float x) {
  const float s = __builtin_amdgcn_sqrtf(x);
  const float h = 0.5f * __builtin_amdgcn_rsqf(x);
  const float s1 = fmaf(fmaf(-s, s, x), h, s);
  return x > 0.f ? s1 : s; // sqrt(0) = 0 (rsq(0) = inf would give NaN)
}
#endif
// 1 / sqrt(x) of a squared emitter distance (x > 0): the hardware seed plus one Newton step — five instructions where rcp_nr(sqrt_nr(x)) took ten
// (two transcendental seeds, two refinements), the same value to the last bit or two
__device__ __forceinline__ float rsqrt_nr(float x) {
  const float y = __builtin_amdgcn_rsqf(x);
  return fmaf(0.5f * y, fmaf(-(x * y), y, 1.0f), y);
}

// Shading normal of a hit on a record flagged by ffx_smooth (include/ffx.h): the three vertex normals stored next to the
// record, interpolated with Moller-Trumbore's barycentrics of the hit (P = v0 + u e1 + v e2, recomputed here from the record:
// the walks do not carry them), normalised and faced to the viewer by the sign of ITS OWN cosine (the `twosided` wrapper flips
// in the shading frame).  Same operation order as the oracle's shade_sample; FAST: the Newton-refined reciprocals of the packet
// kernels instead of IEEE division (equal except for rare last bits).  A zero-length interpolated normal keeps the geometric one.
template <bool FAST>
__device__ __forceinline__ void hit_barycentrics(float4 ra, float4 rb, float4 rc, v3 o, v3 d, float &bu, float &bv) {
  const v3 e1 = V3(ra.w, rb.x, rb.y), e2 = V3(rb.z, rb.w, rc.x);
  const v3 pv = vcross(d, e2);
  const float det = vdot(e1, pv);
  const v3 tv = vsub(o, V3(ra.x, ra.y, ra.z));
  const v3 qv = vcross(tv, e1);
  const float idet = FAST ? rcp_nr(det) : 1.0f / det;
  bu = vdot(tv, pv) * idet;
  bv = vdot(d, qv) * idet;
}
// base colour of a hit on a row with FFX_MAT_BASE_TEX = k + 1 (ffx.h ffx_scene_desc.base_tex): the slot's texture coordinates
// interpolated with the hit's barycentrics, repeat wrap, bilinear between texel centres — the oracle's base_tex_lookup
__device__ __forceinline__ void base_tex_sample(const ShadeK &c, int k, int slot, float bu, float bv, float (&rgb)[3]) {
  const float *uv = c.slot_uv + 6 * (size_t)slot;
  const float bw = (1.0f - bu) - bv;
  float u = fmaf(bw, uv[0], fmaf(bu, uv[2], bv * uv[4])), v = fmaf(bw, uv[1], fmaf(bu, uv[3], bv * uv[5]));
  u = u - floorf(u);
  v = v - floorf(v);
  const int w = c.btw[k], h = c.bth[k];
  const float fx = fmaf(u, (float)w, -0.5f), fy = fmaf(v, (float)h, -0.5f);
  const float x0f = floorf(fx), y0f = floorf(fy);
  const float ax = fx - x0f, ay = fy - y0f;
  int x0 = (int)x0f, y0 = (int)y0f, x1 = x0 + 1, y1 = y0 + 1; // in [-1, w]: one conditional step wraps them
  x0 = x0 < 0 ? x0 + w : x0; x1 = x1 >= w ? x1 - w : x1; y0 = y0 < 0 ? y0 + h : y0; y1 = y1 >= h ? y1 - h : y1;
  x0 = x0 >= w ? x0 - w : x0; y0 = y0 >= h ? y0 - h : y0; x1 = x1 < 0 ? x1 + w : x1; y1 = y1 < 0 ? y1 + h : y1;
  const float *t = c.btex[k];
#pragma unroll
  for (int ch = 0; ch < 3; ++ch) {
    const float t00 = t[((size_t)y0 * w + x0) * 3 + ch], t01 = t[((size_t)y0 * w + x1) * 3 + ch];
    const float t10 = t[((size_t)y1 * w + x0) * 3 + ch], t11 = t[((size_t)y1 * w + x1) * 3 + ch];
    rgb[ch] = (1.0f - ay) * ((1.0f - ax) * t00 + ax * t01) + ay * ((1.0f - ax) * t10 + ax * t11);
  }
}
template <bool FAST>
__device__ __forceinline__ v3 interpolated_normal(const float4 *__restrict__ nrec, int slot, float4 ra, float4 rb, float4 rc, v3 o, v3 d, v3 ng) {
  float bu, bv;
  hit_barycentrics<FAST>(ra, rb, rc, o, d, bu, bv);
  const float bw = (1.0f - bu) - bv;
  const float4 n0 = nrec[3 * (size_t)slot], n1 = nrec[3 * (size_t)slot + 1], n2 = nrec[3 * (size_t)slot + 2];
  const v3 ni = V3(fmaf(bw, n0.x, fmaf(bu, n1.x, bv * n2.x)), fmaf(bw, n0.y, fmaf(bu, n1.y, bv * n2.y)), fmaf(bw, n0.z, fmaf(bu, n1.z, bv * n2.z)));
  const float l2 = vdot(ni, ni);
  if (!(l2 > 0.f)) return ng;
  const float il = FAST ? rcp_nr(sqrt_nr(l2)) : 1.0f / sqrtf(l2);
  v3 ns = V3(ni.x * il, ni.y * il, ni.z * il);
  if (vdot(ns, d) > 0.f) ns = V3(-ns.x, -ns.y, -ns.z);
  return ns;
}

// BSDF of a material row (include/ffx.h FFX_MAT_*, model 1: the reflection side of Mitsuba's `principled`), for the
// viewer direction wv and an emitter direction wl at a surface with unit normal n facing the viewer:
//     pi * f(wv, wl) * cos_o = base_color * A + B   per colour channel.
// The model is material_eval of oracle/ffx_oracle.c (which cites its sources), arranged for the kernel:
//   * the directions are reduced to five cosines (MatGeo) first — for both emitters, so that the surface point, the normal
//     and the ray die before the lobes are evaluated; the lobes then run one after the other on those scalars;
//   * Smith's G1 is taken as 2 c / (c + sqrt(c^2 + xy)) (= 2 / (1 + sqrt(1 + xy / c^2))), so the specular lobe
//     D G / (4 cos_i) collapses to D cos_o / ((cos_i + s_i)(cos_o + s_o)): one quotient instead of five;
//   * an isotropic material (anisotropic = 0, the plugin's default) needs no tangent frame at all;
//   * quotients and roots are the bare hardware seeds (brcp / bdiv / bsqrt below).
// Equal to the oracle within a few ulp per term (the parity tests state the tolerance).  The caller handles model 0.
__device__ __forceinline__ float sqrf(float x) { return x * x; }
__device__ __forceinline__ float schlick_weight(float c) {
  float m = 1.0f - c;
  m = fminf(fmaxf(m, 0.f), 1.f);
  return sqrf(sqrf(m)) * m;
}
// reciprocal, quotient and square root inside the BSDF: the bare hardware seeds (1 ulp) — the BSDF scales a sample's
// radiance, it does not decide what is hit, and the parity tests state its tolerance
__device__ __forceinline__ float brcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float bdiv(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
__device__ __forceinline__ float bsqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
// geometry of one (viewer, emitter) pair — all the BSDF needs of the directions, as scalars: five cosines and the three
// GGX quadratic forms (which take the tangent frame for an anisotropic material: formed here, where the vectors are)
struct MatGeo {
  float cos_i, cos_o, ch, ci_h, co_h; // n.wv, n.wl, n.h, wv.h, wl.h
  float tmp, xy_i, xy_o, axay;        // (hx/ax)^2 + (hy/ay)^2 + hz^2, (ax ix)^2 + (ay iy)^2, the same for wl, ax * ay
  float s2;                           // sin^2(theta_h) = hx^2 + hy^2 — NEVER as 1 - ch^2: near-mirror rows (alpha = 0.0025) and a glossy
                                      // clearcoat (alpha = 0.001) divide it by alpha^2, which turned the rounding of ch into 0.4 % of a highlight
};
__device__ __forceinline__ void material_geometry(const float *__restrict__ m, v3 n, v3 wv, v3 wl, MatGeo &g) {
  g.cos_i = vdot(n, wv);
  g.cos_o = vdot(n, wl);
  v3 wh = V3(wv.x + wl.x, wv.y + wl.y, wv.z + wl.z);
  const float ihl = __builtin_amdgcn_rsqf(vdot(wh, wh));
  wh = V3(wh.x * ihl, wh.y * ihl, wh.z * ihl);
  g.ci_h = vdot(wv, wh);
  g.co_h = vdot(wl, wh);
  g.ch = vdot(n, wh);
  const float r2 = sqrf(m[FFX_MAT_ROUGHNESS]), aniso = m[FFX_MAT_ANISOTROPIC];
  if (aniso != 0.f) { // calc_dist_params + the shading frame coordinate_system(n)
    const float aspect = bsqrt(1.0f - 0.9f * aniso);
    const float ax = fmaxf(0.001f, bdiv(r2, aspect)), ay = fmaxf(0.001f, r2 * aspect);
    const float sg = copysignf(1.0f, n.z), ca = -brcp(sg + n.z), cb = n.x * n.y * ca;
    const v3 fs = V3(sg * (sqrf(n.x) * ca) + 1.0f, sg * cb, -sg * n.x), ft = V3(cb, fmaf(n.y, n.y * ca, sg), -n.y);
    const float hx = vdot(wh, fs), hy = vdot(wh, ft);
    g.s2 = sqrf(hx) + sqrf(hy);
    g.tmp = sqrf(bdiv(hx, ax)) + sqrf(bdiv(hy, ay)) + sqrf(g.ch);
    g.xy_i = sqrf(ax * vdot(wv, fs)) + sqrf(ay * vdot(wv, ft));
    g.xy_o = sqrf(ax * vdot(wl, fs)) + sqrf(ay * vdot(wl, ft));
    g.axay = ax * ay;
  } else {
    const float a2 = sqrf(fmaxf(0.001f, r2));
    const v3 cx = vcross(n, wh); // |n x h|^2 = sin^2(theta_h), well conditioned at the peak (same order as the oracle's material_geometry)
    g.s2 = vdot(cx, cx);
    g.tmp = bdiv(g.s2, a2) + sqrf(g.ch);
    g.xy_i = a2 * fmaxf(1.0f - sqrf(g.cos_i), 0.f);
    g.xy_o = a2 * fmaxf(1.0f - sqrf(g.cos_o), 0.f);
    g.axay = a2;
  }
}
__device__ __forceinline__ float ggx1_cc(float c, float c_dot_h) { // smith_ggx1 with alpha = 0.25 (clearcoat)
  const float c2 = sqrf(c);
  float r = bdiv(2.0f * c, c + bsqrt(c2 + 0.0625f * (1.0f - c2)));
  if (c_dot_h * c <= 0.f) r = 0.f;
  return r;
}
// the lobes, one after the other, each reading the material values it needs from the row when it needs them (the row is
// 64 bytes in the vector L1; holding all of it plus the directions in registers cost the kernel two waves of occupancy)
// TEX: the base colour is (b0, b1, b2) — the row's texture sampled at the hit — instead of the row's own m[0..2]
template <bool TEX = false>
__device__ __forceinline__ void material_terms(const float *__restrict__ m, const MatGeo &g, float &A, float &B, float b0 = 0.f, float b1 = 0.f, float b2 = 0.f) {
  const float cos_i = g.cos_i, cos_o = g.cos_o, ch = g.ch, ci_h = g.ci_h, co_h = g.co_h;
  A = 0.f; B = 0.f;
  if (!(cos_i > 0.f && cos_o > 0.f)) return;
  const bool facing = ci_h > 0.f && co_h > 0.f; // (cos_i, cos_o > 0)
  float a = 0.f, b = 0.f;
  const float eta = m[FFX_MAT_ETA];
  const float ct2 = 1.0f - (1.0f - ci_h * ci_h) * sqrf(brcp(eta));
  const float ct = ct2 > 0.f ? bsqrt(ct2) : 0.f; // cosine of the transmitted direction
  const float sw = schlick_weight(eta > 1.0f ? fabsf(ci_h) : ct);
  float F_d;
  {
    const float c = fabsf(ci_h);
    const float ds = c + eta * ct, dp = ct + eta * c, ir = brcp(ds * dp); // one reciprocal for both amplitudes
    const float a_s = (c - eta * ct) * dp * ir, a_p = (ct - eta * c) * ds * ir;
    F_d = 0.5f * (a_s * a_s + a_p * a_p);
    if (eta == 1.0f) F_d = 0.f;
    else if (c == 0.f) F_d = 1.f;
  }
  const float metallic = m[FFX_MAT_METALLIC];
  if (facing && F_d > 0.f) { // main specular reflection lobe: F D G / (4 cos_i)
    const float dden = FFX_PI_F * g.axay * sqrf(g.tmp);
    const float s_i = bsqrt(sqrf(cos_i) + g.xy_i), s_o = bsqrt(sqrf(cos_o) + g.xy_o);
    // D G / (4 cos_i) = cos_o / (dden (cos_i + s_i)(cos_o + s_o))
    float common = bdiv(cos_o, dden * ((cos_i + s_i) * (cos_o + s_o)));
    if (!(ch > 1e-20f * dden)) common = 0.f; // D * cos_h > 1e-20
    const float spec_tint = m[FFX_MAT_SPEC_TINT], m1 = 1.0f - metallic;
    float Fa = metallic * (1.0f - sw), Fb = metallic * sw + m1 * spec_tint * sw + m1 * (1.0f - spec_tint) * F_d;
    if (spec_tint != 0.f) {
      const float lum = TEX ? 0.212671f * b0 + 0.715160f * b1 + 0.072169f * b2 : 0.212671f * m[0] + 0.715160f * m[1] + 0.072169f * m[2];
      const float t = m1 * spec_tint * sqrf((eta - 1.0f) * brcp(eta + 1.0f)) * (1.0f - sw);
      if (lum > 0.f) Fa += bdiv(t, lum);
      else Fb += t;
    }
    a += Fa * common;
    b += Fb * common;
  }
  const float cc = m[FFX_MAT_CLEARCOAT];
  if (cc > 0.f && facing) { // clearcoat
    const float Fcc = sw + (1.0f - sw) * 0.04f;
    const float alpha = 0.1f + (0.001f - 0.1f) * m[FFX_MAT_CLEARCOAT_GLOSS], a2 = sqrf(alpha), c2 = sqrf(ch);
    float Dcc = bdiv(a2 - 1.0f, FFX_PI_F * logf(a2) * (g.s2 + a2 * c2)); // GTR1: 1 + (a2 - 1) cos^2 = sin^2 + a2 cos^2, without the cancellation
    if (!(Dcc * ch > 1e-20f)) Dcc = 0.f;
    const float Gcc = ggx1_cc(cos_i, ci_h) * ggx1_cc(cos_o, co_h);
    b += cc * 0.25f * Fcc * Dcc * Gcc * cos_o;
  }
  const float brdf = (1.0f - metallic) * (1.0f - m[FFX_MAT_SPEC_TRANS]);
  if (brdf > 0.f) { // diffuse + retro-reflection (+ fake subsurface)
    const float Fo = schlick_weight(cos_o), Fi = schlick_weight(cos_i);
    const float f_diff = (1.0f - 0.5f * Fi) * (1.0f - 0.5f * Fo);
    const float Rr = 2.0f * m[FFX_MAT_ROUGHNESS] * sqrf(co_h);
    const float f_retro = Rr * (Fo + Fi + Fo * Fi * (Rr - 1.0f));
    float dterm = f_diff + f_retro;
    const float flat = m[FFX_MAT_FLATNESS];
    if (flat > 0.f) {
      const float Fss90 = Rr * 0.5f;
      const float Fss = (1.0f + (Fss90 - 1.0f) * Fo) * (1.0f + (Fss90 - 1.0f) * Fi);
      const float f_ss = 1.25f * (Fss * (brcp(cos_o + cos_i) - 0.5f) + 0.5f);
      dterm = dterm + (f_ss - dterm) * flat;
    }
    a += brdf * cos_o * 0.3183098861837907f * dterm;
  }
  const float sheen = m[FFX_MAT_SHEEN];
  if (sheen > 0.f && 1.0f - metallic > 0.f) {
    const float sv = sheen * (1.0f - metallic) * schlick_weight(fabsf(co_h)) * cos_o;
    const float lum = TEX ? 0.212671f * b0 + 0.715160f * b1 + 0.072169f * b2 : 0.212671f * m[0] + 0.715160f * m[1] + 0.072169f * m[2], sheen_tint = m[FFX_MAT_SHEEN_TINT];
    if (lum > 0.f) { a += bdiv(sv * sheen_tint, lum); b += sv * (1.0f - sheen_tint); }
    else b += sv;
  }
  A = a * FFX_PI_F;
  B = b * FFX_PI_F;
}

// ---- the same two functions on a row whose constants the host has derived once (ShadeK.mat_pre, FFX_MAT_PRE floats per row): no squares, clamps
// and reciprocals of parameters per evaluation, one flag word instead of five parameter loads and compares.  The arithmetic of the lobes is
// material_terms', with the host's IEEE values where that one forms hardware seeds (a last-bit difference, inside the parity tolerance).
__device__ __forceinline__ void material_geometry_p(const float *__restrict__ m, const float *__restrict__ p, v3 n, v3 wv, v3 wl, MatGeo &g) {
  g.cos_i = vdot(n, wv);
  g.cos_o = vdot(n, wl);
  v3 wh = V3(wv.x + wl.x, wv.y + wl.y, wv.z + wl.z);
  const float ihl = __builtin_amdgcn_rsqf(vdot(wh, wh));
  wh = V3(wh.x * ihl, wh.y * ihl, wh.z * ihl);
  g.ci_h = vdot(wv, wh);
  g.co_h = vdot(wl, wh);
  g.ch = vdot(n, wh);
  if (__float_as_uint(p[9]) & FFX_PRE_ANISO) { // calc_dist_params + the shading frame coordinate_system(n): as material_geometry
    const float r2 = p[10], aniso = m[FFX_MAT_ANISOTROPIC];
    const float aspect = bsqrt(1.0f - 0.9f * aniso);
    const float ax = fmaxf(0.001f, bdiv(r2, aspect)), ay = fmaxf(0.001f, r2 * aspect);
    const float sg = copysignf(1.0f, n.z), ca = -brcp(sg + n.z), cb = n.x * n.y * ca;
    const v3 fs = V3(sg * (sqrf(n.x) * ca) + 1.0f, sg * cb, -sg * n.x), ft = V3(cb, fmaf(n.y, n.y * ca, sg), -n.y);
    const float hx = vdot(wh, fs), hy = vdot(wh, ft);
    g.s2 = sqrf(hx) + sqrf(hy);
    g.tmp = sqrf(bdiv(hx, ax)) + sqrf(bdiv(hy, ay)) + sqrf(g.ch);
    g.xy_i = sqrf(ax * vdot(wv, fs)) + sqrf(ay * vdot(wv, ft));
    g.xy_o = sqrf(ax * vdot(wl, fs)) + sqrf(ay * vdot(wl, ft));
    g.axay = ax * ay;
  } else {
    const float a2 = p[2];
    const v3 cx = vcross(n, wh);
    g.s2 = vdot(cx, cx);
    g.tmp = fmaf(g.s2, p[3], sqrf(g.ch));
    g.xy_i = a2 * fmaxf(1.0f - sqrf(g.cos_i), 0.f);
    g.xy_o = a2 * fmaxf(1.0f - sqrf(g.cos_o), 0.f);
    g.axay = a2;
  }
}
template <bool TEX = false>
__device__ __forceinline__ void material_terms_p(const float *__restrict__ m, const float *__restrict__ p, const MatGeo &g, float &A, float &B, float b0 = 0.f, float b1 = 0.f,
                                                 float b2 = 0.f) {
  const float cos_i = g.cos_i, cos_o = g.cos_o, ch = g.ch, ci_h = g.ci_h, co_h = g.co_h;
  A = 0.f; B = 0.f;
  if (!(cos_i > 0.f && cos_o > 0.f)) return;
  const bool facing = ci_h > 0.f && co_h > 0.f;
  float a = 0.f, b = 0.f;
  const float eta = p[0];
  const uint32_t flags = __float_as_uint(p[9]);
  const float ct2 = 1.0f - (1.0f - ci_h * ci_h) * p[1];
  const float ct = ct2 > 0.f ? bsqrt(ct2) : 0.f;
  const float sw = schlick_weight(eta > 1.0f ? fabsf(ci_h) : ct);
  float F_d;
  {
    const float c = fabsf(ci_h);
    const float ds = c + eta * ct, dp = ct + eta * c, ir = brcp(ds * dp);
    const float a_s = (c - eta * ct) * dp * ir, a_p = (ct - eta * c) * ds * ir;
    F_d = 0.5f * (a_s * a_s + a_p * a_p);
    if (eta == 1.0f) F_d = 0.f;
    else if (c == 0.f) F_d = 1.f;
  }
  const float metallic = p[4];
  if (facing && F_d > 0.f) { // main specular reflection lobe
    const float dden = FFX_PI_F * g.axay * sqrf(g.tmp);
    const float s_i = bsqrt(sqrf(cos_i) + g.xy_i), s_o = bsqrt(sqrf(cos_o) + g.xy_o);
    float common = bdiv(cos_o, dden * ((cos_i + s_i) * (cos_o + s_o)));
    if (!(ch > 1e-20f * dden)) common = 0.f;
    float Fa = metallic * (1.0f - sw), Fb = fmaf(p[5], sw, p[6] * F_d); // (metallic + m1 spec_tint) sw + m1 (1 - spec_tint) F_d
    if (flags & FFX_PRE_TINT) {
      const float lum = TEX ? 0.212671f * b0 + 0.715160f * b1 + 0.072169f * b2 : 0.212671f * m[0] + 0.715160f * m[1] + 0.072169f * m[2];
      const float t = p[11] * (1.0f - sw);
      if (lum > 0.f) Fa += bdiv(t, lum);
      else Fb += t;
    }
    a += Fa * common;
    b += Fb * common;
  }
  if ((flags & FFX_PRE_CLEARCOAT) && facing) {
    const float cc = m[FFX_MAT_CLEARCOAT];
    const float Fcc = sw + (1.0f - sw) * 0.04f;
    const float alpha = 0.1f + (0.001f - 0.1f) * m[FFX_MAT_CLEARCOAT_GLOSS], a2 = sqrf(alpha), c2 = sqrf(ch);
    float Dcc = bdiv(a2 - 1.0f, FFX_PI_F * logf(a2) * (g.s2 + a2 * c2));
    if (!(Dcc * ch > 1e-20f)) Dcc = 0.f;
    const float Gcc = ggx1_cc(cos_i, ci_h) * ggx1_cc(cos_o, co_h);
    b += cc * 0.25f * Fcc * Dcc * Gcc * cos_o;
  }
  const float brdf = p[7];
  if (brdf > 0.f) { // diffuse + retro-reflection (+ fake subsurface)
    const float Fo = schlick_weight(cos_o), Fi = schlick_weight(cos_i);
    const float f_diff = (1.0f - 0.5f * Fi) * (1.0f - 0.5f * Fo);
    const float Rr = p[8] * sqrf(co_h);
    const float f_retro = Rr * (Fo + Fi + Fo * Fi * (Rr - 1.0f));
    float dterm = f_diff + f_retro;
    if (flags & FFX_PRE_FLAT) {
      const float flat = m[FFX_MAT_FLATNESS];
      const float Fss90 = Rr * 0.5f;
      const float Fss = (1.0f + (Fss90 - 1.0f) * Fo) * (1.0f + (Fss90 - 1.0f) * Fi);
      const float f_ss = 1.25f * (Fss * (brcp(cos_o + cos_i) - 0.5f) + 0.5f);
      dterm = dterm + (f_ss - dterm) * flat;
    }
    a += brdf * cos_o * 0.3183098861837907f * dterm;
  }
  if (flags & FFX_PRE_SHEEN) {
    const float sheen = m[FFX_MAT_SHEEN];
    const float sv = sheen * (1.0f - metallic) * schlick_weight(fabsf(co_h)) * cos_o;
    const float lum = TEX ? 0.212671f * b0 + 0.715160f * b1 + 0.072169f * b2 : 0.212671f * m[0] + 0.715160f * m[1] + 0.072169f * m[2], sheen_tint = m[FFX_MAT_SHEEN_TINT];
    if (lum > 0.f) { a += bdiv(sv * sheen_tint, lum); b += sv * (1.0f - sheen_tint); }
    else b += sv;
  }
  A = a * FFX_PI_F;
  B = b * FFX_PI_F;
}

struct SampleTerms {
  int hit, shape, has_proj;
  int ubx, uby; // unclamped bilinear base texel (for the per-sample cache)
  int ix0, ix1, iy0, iy1;
  float wx0, wx1, wy0, wy1;
  float proj_fac;
  float spot[3];
  float proj_fac_b, spot_b[3]; // material rows: the part of the BSDF that does not scale with base_color
  float base[3];               // textured base colours only (MATM == 2 / the lane kernels): the sample's base colour
};

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

__device__ __forceinline__ void shade_sample(const ShadeK &c, const BvhNode *__restrict__ nodes, const TriRec *__restrict__ recs, const float4 *__restrict__ nrec, v3 o,
                                             v3 d, float nt, float ft, SampleTerms &st, int *stack, int stride) {
  Hit h;
  st.hit = traverse<false, true>(nodes, recs, o, d, nt, ft, h, stack, stride);
  st.has_proj = 0;
  st.proj_fac = 0.f; st.proj_fac_b = 0.f;
  st.spot[0] = st.spot[1] = st.spot[2] = 0.f;
  st.spot_b[0] = st.spot_b[1] = st.spot_b[2] = 0.f;
  st.shape = h.shape;
  if (!st.hit) return;
  const float4 *r4 = reinterpret_cast<const float4 *>(recs + h.slot);
  float4 ra = r4[0], rb = r4[1], rc = r4[2];
  v3 P = V3(fmaf(h.t, d.x, o.x), fmaf(h.t, d.y, o.y), fmaf(h.t, d.z, o.z));
  v3 ng = vcross(V3(ra.w, rb.x, rb.y), V3(rb.z, rb.w, rc.x));
  float nl = sqrtf(vdot(ng, ng));
  if (!(nl > 0.f)) return;
  const float inl = 1.0f / nl;
  ng = V3(ng.x * inl, ng.y * inl, ng.z * inl);
  if (vdot(ng, d) > 0.f) ng = V3(-ng.x, -ng.y, -ng.z);
  float pmax = fmaxf(fabsf(P.x), fmaxf(fabsf(P.y), fabsf(P.z)));
  float off = (1.0f + pmax) * RAY_EPS;
  v3 Po = V3(fmaf(off, ng.x, P.x), fmaf(off, ng.y, P.y), fmaf(off, ng.z, P.z));
  // shading normal (ffx_smooth): BSDF and emitter cosines use it; the geometric normal keeps the side tests
  v3 ns = ng;
  if (rc.w != 0.f && nrec) ns = interpolated_normal<false>(nrec, h.slot, ra, rb, rc, o, d, ng);
  // base colour: the shape's row, or (FFX_MAT_BASE_TEX) its texture at the hit
  const float *mt = mat_table(c);
  const float *row0 = mt + (size_t)c.mat_stride * h.shape;
  st.base[0] = row0[0]; st.base[1] = row0[1]; st.base[2] = row0[2];
  bool textured = false;
  if (c.mat_stride == FFX_MAT_STRIDE && c.n_base_tex > 0) {
    const int tix = (int)row0[FFX_MAT_BASE_TEX];
    if (tix > 0 && tix <= c.n_base_tex) {
      float bu, bv;
      hit_barycentrics<false>(ra, rb, rc, o, d, bu, bv);
      base_tex_sample(c, tix - 1, h.slot, bu, bv, st.base);
      textured = true;
    }
  }

  if (c.proj_on) {
    v3 pl = xf_point(c.p_w2l, P);
    if (pl.z > 0.f) {
      const float *m = c.p_c2s;
      float qx = fmaf(m[0], pl.x, fmaf(m[1], pl.y, fmaf(m[2], pl.z, m[3])));
      float qy = fmaf(m[4], pl.x, fmaf(m[5], pl.y, fmaf(m[6], pl.z, m[7])));
      float qw = fmaf(m[12], pl.x, fmaf(m[13], pl.y, fmaf(m[14], pl.z, m[15])));
      const float iqw = 1.0f / qw;
      float u = qx * iqw, v = qy * iqw;
      if (u >= 0.f && u <= 1.f && v >= 0.f && v <= 1.f) {
        v3 ppos = V3(c.p_pos[0], c.p_pos[1], c.p_pos[2]);
        v3 wi = vsub(ppos, P);
        float d2 = vdot(wi, wi);
        const float idist = 1.0f / sqrtf(d2);
        wi = V3(wi.x * idist, wi.y * idist, wi.z * idist);
        float cos_s = vdot(ns, wi);
        float cos_p = -vdot(V3(c.p_axis[0], c.p_axis[1], c.p_axis[2]), wi);
        if (cos_s > 0.f && cos_p > 0.f && vdot(ng, wi) > 0.f) {
          bool vis = true;
          if (c.shadows) { // traced FROM the emitter (the batch's apex) to the lifted surface point: t in (0, 1 - eps)
            Hit hs;
            vis = !traverse<true, true>(nodes, recs, ppos, vsub(Po, ppos), 0.f, 1.0f - SHADOW_EPS, hs, stack, stride);
          }
          if (vis) {
            float bA = cos_s, bB = 0.f; // Lambert; material rows: pi f cos = base_color * bA + bB
            if (c.mat_stride == FFX_MAT_STRIDE && mt[(size_t)FFX_MAT_STRIDE * h.shape + FFX_MAT_MODEL] != 0.f) {
              const float *mrow = mt + (size_t)FFX_MAT_STRIDE * h.shape;
              MatGeo mg;
              material_geometry(mrow, ns, V3(-d.x, -d.y, -d.z), wi, mg);
              if (textured) material_terms<true>(mrow, mg, bA, bB, st.base[0], st.base[1], st.base[2]);
              else material_terms(mrow, mg, bA, bB);
            }
            st.proj_fac = (c.p_scale / (pl.z * pl.z * cos_p)) * bA;
            st.proj_fac_b = (c.p_scale / (pl.z * pl.z * cos_p)) * bB;
            float fx = fmaf(u, (float)c.tw, -0.5f), fy = fmaf(v, (float)c.th, -0.5f);
            float x0 = floorf(fx), y0 = floorf(fy);
            float ax = fx - x0, ay = fy - y0;
            int ix0 = (int)x0, iy0 = (int)y0;
            st.ubx = ix0; st.uby = iy0;
            st.ix0 = clampi(ix0, 0, c.tw - 1);
            st.ix1 = clampi(ix0 + 1, 0, c.tw - 1);
            st.iy0 = clampi(iy0, 0, c.th - 1);
            st.iy1 = clampi(iy0 + 1, 0, c.th - 1);
            st.wx0 = 1.0f - ax; st.wx1 = ax;
            st.wy0 = 1.0f - ay; st.wy1 = ay;
            st.has_proj = 1;
          }
        }
      }
    }
  }
  if (c.spot_on) {
    v3 spos = V3(c.s_pos[0], c.s_pos[1], c.s_pos[2]);
    v3 wi = vsub(spos, P);
    float d2 = vdot(wi, wi);
    const float idist = 1.0f / sqrtf(d2);
    wi = V3(wi.x * idist, wi.y * idist, wi.z * idist);
    float cos_s = vdot(ns, wi);
    if (cos_s > 0.f && vdot(ng, wi) > 0.f) {
      v3 ll = xf_dir(c.s_w2l, V3(-wi.x, -wi.y, -wi.z));
      float ln = sqrtf(vdot(ll, ll));
      float cos_t = ll.z / ln;
      float fall = 0.f;
      if (cos_t >= c.cos_beam) fall = 1.f;
      else if (cos_t > c.cos_cut) fall = (c.cutoff - acosf(cos_t)) * c.inv_trans;
      if (fall > 0.f) {
        bool vis = true;
        if (c.shadows) {
          Hit hs;
          vis = !traverse<true, true>(nodes, recs, spos, vsub(Po, spos), 0.f, 1.0f - SHADOW_EPS, hs, stack, stride);
        }
        if (vis) {
          float bA = cos_s, bB = 0.f;
          if (c.mat_stride == FFX_MAT_STRIDE && mt[(size_t)FFX_MAT_STRIDE * h.shape + FFX_MAT_MODEL] != 0.f) {
            const float *mrow = mt + (size_t)FFX_MAT_STRIDE * h.shape;
            MatGeo mg;
            material_geometry(mrow, ns, V3(-d.x, -d.y, -d.z), wi, mg);
            if (textured) material_terms<true>(mrow, mg, bA, bB, st.base[0], st.base[1], st.base[2]);
            else material_terms(mrow, mg, bA, bB);
          }
          float f = fall * bA / d2 * 0.3183098861837907f, fb = fall * bB / d2 * 0.3183098861837907f;
          st.spot[0] = c.s_int[0] * f;
          st.spot[1] = c.s_int[1] * f;
          st.spot[2] = c.s_int[2] * f;
          st.spot_b[0] = c.s_int[0] * fb;
          st.spot_b[1] = c.s_int[1] * fb;
          st.spot_b[2] = c.s_int[2] * fb;
        }
      }
    }
  }
}

// blockIdx -> tile.  Workgroups b, b+8, b+16, ... share an XCD (and its L2).  mode 1 gives each XCD a
// contiguous 1/8 band of the image (best L2 locality, but bands differ a lot in cost: the XCD that
// owns the vocal folds finishes last while others idle); mode 0 is the identity (XCDs interleave at
// tile granularity: balanced; the whole BVH fits every XCD's L2 anyway).  Pure performance.
// mode >= 2: blocked interleave with block size B = mode: within every group of 8*B workgroups, XCD k
// walks B CONSECUTIVE work items — the waves resident on one CU then work on adjacent pixels and share
// scalar-cache / L2 lines of the lower tree levels, while the groups stay small enough to balance.
__device__ __forceinline__ int xcd_remap(int b, int nblocks, int mode) {
  if (mode == 0) return b;
  if (mode == 1) {
    int per = (nblocks + 7) / 8;
    return (b % 8) * per + (b / 8);
  }
  const int B = mode, G = 8 * B;
  if ((B & (B - 1)) == 0) { // a power of two (the default, 128): shifts instead of two integer divisions per wave
    const int lb = 31 - __builtin_clz(B);
    const int g = b >> (lb + 3), r = b & (G - 1);
    if (((g + 1) << (lb + 3)) > nblocks) return b; // ragged last group: identity
    return (g << (lb + 3)) + ((r & 7) << lb) + (r >> 3);
  }
  const int g = b / G, r = b % G;
  if ((g + 1) * G > nblocks) return b; // ragged last group: identity
  return g * G + (r % 8) * B + (r / 8);
}

__global__ void __launch_bounds__(TR_BLOCK)
    k_render_fwd(ShadeK c, const BvhNode *__restrict__ nodes, const TriRec *__restrict__ recs, const float4 *__restrict__ nrec, const float *__restrict__ albedo,
                 const float *__restrict__ tex, int spp, uint32_t seed_key, int tiles_x, int n_tiles, int remap, int fp16, void *__restrict__ img) {
  extern __shared__ int s_dyn[];
  __shared__ float s_red[3][3][64]; // waves 1..3 -> wave 0
  int tile = xcd_remap(blockIdx.x, gridDim.x, remap);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int W = c.cam.W, H = c.cam.H;
  int px = (tile % tiles_x) * 8 + (lane & 7), py = (tile / tiles_x) * 8 + (lane >> 3);
  bool live = tile < n_tiles && px < W && py < H;
  float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f;
  if (live) {
    const uint32_t pix = (uint32_t)py * (uint32_t)W + (uint32_t)px;
    for (int s = wave; s < spp; s += TR_BLOCK / 64) {
      uint32_t idx = pix * (uint32_t)spp + (uint32_t)s;
      float jx, jy;
      sample_jitter(seed_key, idx, jx, jy);
      v3 o, d;
      float nt, ft;
      cam_ray(c.cam, ((float)px + jx) * c.cam.inv_w, ((float)py + jy) * c.cam.inv_h, o, d, nt, ft);
      SampleTerms st;
      shade_sample(kernarg_shade(), nodes, recs, nrec, o, d, nt, ft, st, s_dyn + threadIdx.x, TR_BLOCK);
      if (!st.hit) continue;
      float r0 = st.spot[0], r1 = st.spot[1], r2 = st.spot[2];
      float b0 = st.spot_b[0], b1 = st.spot_b[1], b2 = st.spot_b[2];
      if (st.has_proj) {
        const int tc = c.tc;
        size_t o00 = ((size_t)st.iy0 * c.tw + st.ix0) * tc, o01 = ((size_t)st.iy0 * c.tw + st.ix1) * tc;
        size_t o10 = ((size_t)st.iy1 * c.tw + st.ix0) * tc, o11 = ((size_t)st.iy1 * c.tw + st.ix1) * tc;
        if (tc == 1) {
          float tv = st.wy0 * (st.wx0 * tex[o00] + st.wx1 * tex[o01]) + st.wy1 * (st.wx0 * tex[o10] + st.wx1 * tex[o11]);
          r0 += tv * c.p_color[0] * st.proj_fac;
          r1 += tv * c.p_color[1] * st.proj_fac;
          r2 += tv * c.p_color[2] * st.proj_fac;
          b0 += tv * c.p_color[0] * st.proj_fac_b;
          b1 += tv * c.p_color[1] * st.proj_fac_b;
          b2 += tv * c.p_color[2] * st.proj_fac_b;
        } else {
          float tv0 = st.wy0 * (st.wx0 * tex[o00] + st.wx1 * tex[o01]) + st.wy1 * (st.wx0 * tex[o10] + st.wx1 * tex[o11]);
          float tv1 = st.wy0 * (st.wx0 * tex[o00 + 1] + st.wx1 * tex[o01 + 1]) + st.wy1 * (st.wx0 * tex[o10 + 1] + st.wx1 * tex[o11 + 1]);
          float tv2 = st.wy0 * (st.wx0 * tex[o00 + 2] + st.wx1 * tex[o01 + 2]) + st.wy1 * (st.wx0 * tex[o10 + 2] + st.wx1 * tex[o11 + 2]);
          r0 += tv0 * 1.0f * st.proj_fac;
          r1 += tv1 * 1.0f * st.proj_fac;
          r2 += tv2 * 1.0f * st.proj_fac;
          b0 += tv0 * 1.0f * st.proj_fac_b;
          b1 += tv1 * 1.0f * st.proj_fac_b;
          b2 += tv2 * 1.0f * st.proj_fac_b;
        }
      }
      const float *alb = st.base; // (the shape's row, or its base-colour texture at the hit)
      if (c.mat_stride == 3) {
        acc0 += alb[0] * r0;
        acc1 += alb[1] * r1;
        acc2 += alb[2] * r2;
      } else {
        acc0 += alb[0] * r0 + b0;
        acc1 += alb[1] * r1 + b1;
        acc2 += alb[2] * r2 + b2;
      }
    }
  }
  if (wave > 0) {
    s_red[wave - 1][0][lane] = acc0;
    s_red[wave - 1][1][lane] = acc1;
    s_red[wave - 1][2][lane] = acc2;
  }
  __syncthreads();
  if (wave == 0 && live) {
#pragma unroll
    for (int w = 0; w < 3; ++w) {
      acc0 += s_red[w][0][lane];
      acc1 += s_red[w][1][lane];
      acc2 += s_red[w][2][lane];
    }
    float inv_spp = 1.0f / (float)spp;
    size_t o = ((size_t)py * W + px) * 3;
    if (fp16 & 1) {
      _Float16 *p = (_Float16 *)img;
      p[o] = (_Float16)(acc0 * inv_spp);
      p[o + 1] = (_Float16)(acc1 * inv_spp);
      p[o + 2] = (_Float16)(acc2 * inv_spp);
    } else {
      float *p = (float *)img;
      p[o] = acc0 * inv_spp;
      p[o + 1] = acc1 * inv_spp;
      p[o + 2] = acc2 * inv_spp;
    }
  }
}

// K9: replay the same samples, scatter d(loss)/d(img) * d(img)/d(tex) through the bilinear weights.
__global__ void __launch_bounds__(TR_BLOCK)
    k_render_bwd(ShadeK c, const BvhNode *__restrict__ nodes, const TriRec *__restrict__ recs, const float4 *__restrict__ nrec, const float *__restrict__ albedo, int spp,
                 uint32_t seed_key, int tiles_x, int n_tiles, int remap, const float *__restrict__ gimg, float *__restrict__ gtex) {
  extern __shared__ int s_dyn[];
  int tile = xcd_remap(blockIdx.x, gridDim.x, remap);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int W = c.cam.W, H = c.cam.H;
  int px = (tile % tiles_x) * 8 + (lane & 7), py = (tile / tiles_x) * 8 + (lane >> 3);
  if (!(tile < n_tiles && px < W && py < H)) return;
  const uint32_t pix = (uint32_t)py * (uint32_t)W + (uint32_t)px;
  const float g0 = gimg[(size_t)pix * 3], g1 = gimg[(size_t)pix * 3 + 1], g2 = gimg[(size_t)pix * 3 + 2];
  if (g0 == 0.f && g1 == 0.f && g2 == 0.f) return;
  const float inv_spp = 1.0f / (float)spp;
  const int tc = c.tc;
  for (int s = wave; s < spp; s += TR_BLOCK / 64) {
    uint32_t idx = pix * (uint32_t)spp + (uint32_t)s;
    float jx, jy;
    sample_jitter(seed_key, idx, jx, jy);
    v3 o, d;
    float nt, ft;
    cam_ray(c.cam, ((float)px + jx) * c.cam.inv_w, ((float)py + jy) * c.cam.inv_h, o, d, nt, ft);
    SampleTerms st;
    shade_sample(kernarg_shade(), nodes, recs, nrec, o, d, nt, ft, st, s_dyn + threadIdx.x, TR_BLOCK);
    if (!st.hit || !st.has_proj) continue;
    const float *alb = st.base;
    size_t o00 = ((size_t)st.iy0 * c.tw + st.ix0) * tc, o01 = ((size_t)st.iy0 * c.tw + st.ix1) * tc;
    size_t o10 = ((size_t)st.iy1 * c.tw + st.ix0) * tc, o11 = ((size_t)st.iy1 * c.tw + st.ix1) * tc;
    if (tc == 1) {
      float ws = (g0 * alb[0] * c.p_color[0] + g1 * alb[1] * c.p_color[1] + g2 * alb[2] * c.p_color[2]) * st.proj_fac * inv_spp;
      if (st.proj_fac_b != 0.f) ws += (g0 * c.p_color[0] + g1 * c.p_color[1] + g2 * c.p_color[2]) * st.proj_fac_b * inv_spp;
      atomicAdd(gtex + o00, ws * st.wy0 * st.wx0);
      atomicAdd(gtex + o01, ws * st.wy0 * st.wx1);
      atomicAdd(gtex + o10, ws * st.wy1 * st.wx0);
      atomicAdd(gtex + o11, ws * st.wy1 * st.wx1);
    } else {
      const float gg[3] = {g0, g1, g2};
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) {
        float ws = gg[ch] * alb[ch] * st.proj_fac * inv_spp;
        if (st.proj_fac_b != 0.f) ws += gg[ch] * st.proj_fac_b * inv_spp;
        atomicAdd(gtex + o00 + ch, ws * st.wy0 * st.wx0);
        atomicAdd(gtex + o01 + ch, ws * st.wy0 * st.wx1);
        atomicAdd(gtex + o10 + ch, ws * st.wy1 * st.wx0);
        atomicAdd(gtex + o11 + ch, ws * st.wy1 * st.wx1);
      }
    }
  }
}


// wave ballot of a predicate: the compiler's builtin reads the lane mask directly, whereas HIP's
// __ballot(int) materialises 0/1 in a VGPR and compares it again (two VALU instructions per call)
__device__ __forceinline__ unsigned long long wballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
// popcount of a lane mask as a 32-bit SCALAR (the compiler widens a comparison of two popcountll
// results to 64 bits, which has no scalar compare and lands on the VALU)
__device__ __forceinline__ int wpop(unsigned long long m) { int r; asm("s_bcnt1_i32_b64 %0, %1" : "=s"(r) : "s"(m) : "scc"); return r; }
// ------------------------------------------------------------------------------------------ wave-packet traversal
// One wavefront = one packet of 64 coherent rays (4x4 pixels x 4 samples, or their shadow rays).
// Control flow is WAVE-UNIFORM: the packet walks the union of its rays' paths.  Because the node
// index is uniform, a node / triangle record is fetched ONCE per wave through the scalar cache
// (s_load) into SGPRs and feeds the per-lane slab / Moller-Trumbore arithmetic as scalar operands;
// the vector memory pipe is not used at all during traversal.  The traversal stack is a single
// VGPR addressed with v_writelane / v_readlane (entry i lives in lane i): no LDS, no barriers.
// Per-lane results are identical to the per-lane traversal (same tests, closest hit with the
// primitive-id tie-break is order independent).
// v_writelane_b32: vec[lane] = val (val, lane wave-uniform, in SGPRs).  clang exposes readlane but not
// writelane; the s_nop covers the "VALU wrote the SGPR used as lane select" hazard, which hipcc
// cannot see inside an asm statement.
__device__ __forceinline__ int writelane_i32(int val, int lane, int vec) {
  // gfx9 VALU ops may read ONE SGPR over the constant bus: the lane select travels in M0
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 4\n\tv_writelane_b32 %0, %1, m0" : "+v"(vec) : "s"(val), "s"(lane) : "m0");
  return vec;
}
// the same, updating the stack register in place (no copy of the whole VGPR around the push)
__device__ __forceinline__ void push_lane(int &vec, int val, int lane) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 4\n\tv_writelane_b32 %0, %1, m0" : "+v"(vec) : "s"(val), "s"(lane) : "m0");
}
// dst = mask ? v : dst, in place (keeps a loop-carried value in one register across the branch that updates it)
__device__ __forceinline__ void msel_into(float &dst, unsigned long long mask, float v) {
  asm("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(dst) : "v"(v), "s"(mask));
}
__device__ __forceinline__ void msel_into(int &dst, unsigned long long mask, int v) {
  asm("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(dst) : "v"(v), "s"(mask));
}

// Generic packet walk (min/max slab test, no assumption on direction signs) for R rays per lane.  The
// kernels instantiate R = 1 only — a packed two-rays-per-lane variant was measured slower and removed —
// and reach this loop for the rare packets whose rays disagree on a direction sign; everything else
// takes the octant loops below.  Same apex triangle test, same results.
template <bool ANY, int R>
__device__ __forceinline__ void traverse_packet(const BvhNode *__restrict__ nodes, const TriApex *__restrict__ recs, const v3 (&o)[R], const v3 (&d)[R],
                                                const float (&tmin)[R], const float (&tmax)[R], const bool (&active)[R], Hit (&h)[R], bool (&found)[R]) {
  RayBox rb[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    h[r].t = active[r] ? tmax[r] : -INFINITY; // an inactive ray fails every slab test
    h[r].prim = -1;
    h[r].shape = -1;
    h[r].slot = -1;
    found[r] = false;
    rb[r] = make_raybox(o[r], d[r]);
  }
  int stack_v = 0;
  int sp = 0;
  int cur = 0;
  while (true) {
    cur = __builtin_amdgcn_readfirstlane(cur);
    // ONE 64-byte scalar fetch per node (uniform address -> scalar loads), no conditional loads
    const float4 *n4 = reinterpret_cast<const float4 *>(nodes + cur);
    const float4 q0 = n4[0], q1 = n4[1], q2 = n4[2];
    const int4 ch = *reinterpret_cast<const int4 *>(n4 + 3);
    const int c0 = ch.x, c1 = ch.y;
    const float lo0[3] = {q0.x, q0.y, q0.z}, hi0[3] = {q0.w, q1.x, q1.y};
    const float lo1[3] = {q1.z, q1.w, q2.x}, hi1[3] = {q2.y, q2.z, q2.w};
    float t0[R], t1[R];
    bool h0[R], h1[R];
    bool any0 = false, any1 = false;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      h0[r] = slab(lo0, hi0, rb[r], tmin[r], h[r].t, t0[r]) & (c0 != FFX_EMPTY_CHILD);
      h1[r] = slab(lo1, hi1, rb[r], tmin[r], h[r].t, t1[r]) & (c1 != FFX_EMPTY_CHILD);
      any0 |= h0[r];
      any1 |= h1[r];
    }
#pragma unroll
    for (int side = 0; side < 2; ++side) {
      const int c = side ? c1 : c0;
      if (c < 0 && c != FFX_EMPTY_CHILD && wballot(side ? any1 : any0) != 0ull) { // wave-uniform
        const uint32_t lc = (uint32_t)~c;
        const int first = (int)(lc >> 3), count = (int)(lc & 7u) + 1;
        for (int i = 0; i < count; ++i) {
          const float4 *r4 = reinterpret_cast<const float4 *>(recs + first + i);
          const float4 ra = r4[0], rb4 = r4[1], rc = r4[2];
          const int prim = __float_as_int(rc.z), shape = __float_as_int(rc.w);
          // apex record: det = d.A, U = d.B, V = d.C, t = T/det (ffx_common.h)
          const v3 A = V3(ra.x, ra.y, ra.z), B = V3(ra.w, rb4.x, rb4.y), C = V3(rb4.z, rb4.w, rc.x);
          const float Tq = rc.y;
          // three stages with WAVE-UNIFORM early-outs (same arithmetic and acceptance rule as
          // tri_hit_apex): the rays of a packet nearly always fail together at the first barycentric
          // test, so the remaining dot product and the IEEE division are skipped for the whole wave.
          // (U <= det is implied by V >= 0 and U + V <= det.)
          bool p1[R], neg[R], any1 = false;
          float detA[R], Us[R];
#pragma unroll
          for (int r = 0; r < R; ++r) {
            const float det = vdot(d[r], A);
            const float U = vdot(d[r], B);
            neg[r] = det < 0.f;
            detA[r] = neg[r] ? -det : det;
            Us[r] = neg[r] ? -U : U;
            p1[r] = (side ? h1[r] : h0[r]) & (detA[r] > 0.f) & (Us[r] >= 0.f) & (Us[r] <= detA[r]);
            any1 |= p1[r];
          }
          if (wballot(any1) == 0ull) continue;
          bool p2[R], any2 = false;
#pragma unroll
          for (int r = 0; r < R; ++r) {
            const float Vv = vdot(d[r], C);
            const float Vs = neg[r] ? -Vv : Vv;
            p2[r] = p1[r] & (Vs >= 0.f) & (Us[r] + Vs <= detA[r]);
            any2 |= p2[r];
          }
          if (wballot(any2) == 0ull) continue;
#pragma unroll
          for (int r = 0; r < R; ++r) {
            const float t = (neg[r] ? -Tq : Tq) / detA[r];
            const bool hit = p2[r] & (t > tmin[r]);
            if (ANY) {
              const bool occ = hit & (t < tmax[r]);
              found[r] = found[r] | occ;
              h[r].t = occ ? -INFINITY : h[r].t;
            } else {
              const bool better = hit & (t <= tmax[r]) & ((h[r].prim < 0) | (t < h[r].t) | ((t == h[r].t) & (prim < h[r].prim)));
              h[r].t = better ? t : h[r].t;
              h[r].prim = better ? prim : h[r].prim;
              h[r].shape = better ? shape : h[r].shape;
              h[r].slot = better ? first + i : h[r].slot;
            }
          }
        }
      }
    }
    if (ANY) {
      bool undecided = false;
#pragma unroll
      for (int r = 0; r < R; ++r) undecided |= active[r] & !found[r];
      if (wballot(undecided) == 0ull) break; // every ray of the packet is decided
    }
    bool g0 = false, g1 = false, first1 = false, first0 = false;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const bool a0 = h0[r] & (c0 >= 0) & (t0[r] <= h[r].t), a1 = h1[r] & (c1 >= 0) & (t1[r] <= h[r].t);
      g0 |= a0;
      g1 |= a1;
      first1 |= a1 & (!a0 | (t1[r] < t0[r]));
      first0 |= a0 & (!a1 | (t0[r] <= t1[r]));
    }
    const unsigned long long m0 = wballot(g0), m1 = wballot(g1);
    if (m0 != 0ull && m1 != 0ull) {
      // visit first the child that most lanes enter first
      const bool swap = wpop(wballot(first1)) > wpop(wballot(first0));
      stack_v = writelane_i32(swap ? c0 : c1, sp, stack_v);
      ++sp;
      cur = swap ? c1 : c0;
    } else if (m0 != 0ull) {
      cur = c0;
    } else if (m1 != 0ull) {
      cur = c1;
    } else {
      if (sp == 0) break;
      --sp;
      cur = __builtin_amdgcn_readlane(stack_v, sp);
    }
  }
}

// ------------------------------------------------------------------------------------------ octant-specialised packet walk
// All rays of a one-pixel packet (and of its shadow packets) almost always share the signs of their
// direction components.  Then, per axis, which of the two box planes is entered first is known for the
// whole packet, and the slab test needs neither min/max nor a subtract:
//     t_near[a] = near_plane[a] * id[a] - oidN[a]        t_far[a] = far_plane[a] * id[a] - oidF[a]
// one v_fma_f32 per plane with the node plane as the SGPR operand: 12 VALU per box instead of 27
// (3 v_mov + 6 sub + 6 mul + 6 min/max + ...).  oid = o*id; the fma form loses the exact cancellation
// of (plane - o)*id, so its rounding error |oid|*2^-24 is covered by a per-axis pad e = |oid|*2^-22
// folded into the two constants oidN = oid + e (entry can only move earlier) and oidF = oid - e (exit
// only later): boxes are hit at least as often as in exact arithmetic, and the triangle test — which
// alone decides the result — is unchanged.  The loop is compiled once per octant (OCT bit a set =
// direction component a negative) and selected per walk by a wave-uniform switch; packets with mixed
// signs take the generic loop.
#ifndef FFX_OCTANT_LOOPS
#define FFX_OCTANT_LOOPS 1
#endif
#ifndef FFX_PK1_WAVES
// resident waves per SIMD the 1-ray packet kernels are register-budgeted for.  8 -> 64 VGPRs / 78 SGPRs.  With
// the 64-wide walk the hot loops fit: what spills (9 dwords) are loop-invariant addresses and pixel indices,
// stored once per wave and reloaded in the per-pixel epilogue.  Measured through the bench loop (tools/k8sweep.sh):
// 5 / 6 / 7 / 8 waves: 1235 / 1325 / 1376 / 1404 renders/s — the walk waits on one vector load per step, more
// resident waves hide it.  (The binary walk preferred 7: its spills were inside the node loop.)
#define FFX_PK1_WAVES 8
#endif
__device__ __forceinline__ constexpr bool octant_loops() { return FFX_OCTANT_LOOPS != 0; }
// `scale` = 1/tmax of the ray: the octant box test works in units of the ray's own parameter range, so
// every distance it compares lies in [0, 1] (see slab_oct).  The scale multiplies 1/d once, hence enters
// plane*m and o*m alike: a purely relative change of the results, no new cancellation error.
struct RayOct { v3 id, oidN, oidF; };
__device__ __forceinline__ RayOct make_rayoct(v3 o, v3 d, float scale) {
  RayOct r;
  r.id = V3(safe_rcp_dir(d.x) * scale, safe_rcp_dir(d.y) * scale, safe_rcp_dir(d.z) * scale);
  const v3 oid = V3(o.x * r.id.x, o.y * r.id.y, o.z * r.id.z);
  const float k = 2.384185791015625e-07f; // 2^-22
  const v3 e = V3(fabsf(oid.x) * k, fabsf(oid.y) * k, fabsf(oid.z) * k);
  r.oidN = V3(oid.x + e.x, oid.y + e.y, oid.z + e.z);
  r.oidF = V3(oid.x - e.x, oid.y - e.y, oid.z - e.z);
  return r;
}
// Predicates are kept as explicit 64-bit lane masks: v_cmp writes them straight into an SGPR pair,
// they are combined on the scalar ALU, tested with s_cmp (no ballot) and consumed by v_cndmask through
// inverse_ballot.  (A `bool` that is not itself a compare costs v_cndmask + v_cmp_ne per ballot.)
typedef unsigned long long wmask;
// hides a uniform mask from common-subexpression elimination: the compiler otherwise evaluates one
// `g != 0` for two branches up front and keeps the outcome as s_cselect'ed lane masks (4 extra SALU)
__device__ __forceinline__ wmask launder_mask(wmask m) { asm("" : "+s"(m)); return m; }
typedef int v16i __attribute__((ext_vector_type(16)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v4i __attribute__((ext_vector_type(4)));
__device__ __forceinline__ wmask m_lt(float a, float b) { return __builtin_amdgcn_fcmpf(a, b, 4); }  // ordered <
__device__ __forceinline__ wmask m_le(float a, float b) { return __builtin_amdgcn_fcmpf(a, b, 5); }  // ordered <=
__device__ __forceinline__ wmask m_gt(float a, float b) { return __builtin_amdgcn_fcmpf(a, b, 2); }  // ordered >
__device__ __forceinline__ wmask m_ge(float a, float b) { return __builtin_amdgcn_fcmpf(a, b, 3); }  // ordered >=
__device__ __forceinline__ wmask m_eq(float a, float b) { return __builtin_amdgcn_fcmpf(a, b, 1); }  // ordered ==
__device__ __forceinline__ wmask m_ult(uint32_t a, uint32_t b) { return __builtin_amdgcn_uicmp(a, b, 36); }
template <typename T>
__device__ __forceinline__ T msel(wmask m, T a, T b) { return __builtin_amdgcn_inverse_ballot_w64(m) ? a : b; }
// min/max without the sNaN-quieting v_max(x,x) the compiler puts in front of fminf/fmaxf operands it
// cannot prove canonical (the loop-carried hit distance): inputs here are never NaN
__device__ __forceinline__ float vmin3(float a, float b, float c) { float r; asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
// max3 with the VOP3 clamp bit: the result is clamped to [0, 1] at no cost
__device__ __forceinline__ float vmax3_sat(float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3 clamp" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ float vmin2(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
// a * b with b wave-uniform, as ONE v_mul with a scalar operand (written out so that the vectoriser does not pair the
// multiplies of a pixel's channels into v_pk_mul_f32, whose VGPR copy of b it hoists out of every loop and spills)
__device__ __forceinline__ float vmul_s(float a, float b_uniform) { float r; asm("v_mul_f32 %0, %2, %1" : "=v"(r) : "v"(a), "s"(b_uniform)); return r; }
// b is wave-uniform (an SGPR): VOP2's first source may be scalar — no v_mov in front of the min
__device__ __forceinline__ float vmin2_s(float a, uint32_t b_bits) { float r; asm("v_min_f32 %0, %2, %1" : "=v"(r) : "v"(a), "s"(b_bits)); return r; }

// Box test of the octant loops: one v_fma_f32 per plane with the node plane as the SGPR operand.
// Per plane the ray carries a multiplier and a constant,
//     entered-first plane:  m = id[a],      c = oidN[a]
//     left-last plane:      m = id[a] * k,  c = oidF[a] * k      (k = 1 + 3*2^-23, the far-side widening)
// which of lo/hi is which is fixed by the octant at compile time.  (v_pk_fma_f32 with the node planes
// as an SGPR pair was measured: it occupies ~1.5 issue slots on gfx950 and needs three more VGPRs, and
// the kernel got 8% slower — profiles/r1_issue_mix.txt.)
struct RaySlab { v3 id, idk, cN, cF; };
__device__ __forceinline__ RaySlab make_rayslab(const RayOct &rb) {
  const float k = 1.0000004f;
  RaySlab r;
  r.id = rb.id;
  r.idk = V3(rb.id.x * k, rb.id.y * k, rb.id.z * k);
  r.cN = rb.oidN;
  r.cF = V3(rb.oidF.x * k, rb.oidF.y * k, rb.oidF.z * k);
  return r;
}
// Distances are in units of the ray's parameter range (RayOct::scale), so a box can only matter if it is
// entered within [0, 1): the lower bound max(tn, tmin) becomes the free `clamp` bit of v_max3_f32 (the
// near clip distance is tiny: boxes nearer than it cost a visit at most, the triangle test still
// enforces t > tmin), and the upper bound is the scaled hit distance `hts` <= 0.9990002 (so an entry
// distance clamped down to 1 still fails).  10 VALU per box.
template <int OCT>
__device__ __forceinline__ wmask slab_oct(const float lo[3], const float hi[3], const RaySlab &rs, float hts, float &tn_out) {
  const float nx = (OCT & 1) ? hi[0] : lo[0], fx = (OCT & 1) ? lo[0] : hi[0];
  const float ny = (OCT & 2) ? hi[1] : lo[1], fy = (OCT & 2) ? lo[1] : hi[1];
  const float nz = (OCT & 4) ? hi[2] : lo[2], fz = (OCT & 4) ? lo[2] : hi[2];
  const float tn = vmax3_sat(fmaf(nx, rs.id.x, -rs.cN.x), fmaf(ny, rs.id.y, -rs.cN.y), fmaf(nz, rs.id.z, -rs.cN.z));
  const float tf = vmin2(vmin3(fmaf(fx, rs.idk.x, -rs.cF.x), fmaf(fy, rs.idk.y, -rs.cF.y), fmaf(fz, rs.idk.z, -rs.cF.z)), hts);
  tn_out = tn;
  return m_le(tn, tf);
}

// The scalar ALU issues at the same rate as the VALU (one instruction per SIMD per 4 cycles), so the
// uniform bookkeeping is written to stay short: 32-bit byte offsets for node / record addressing, a
// fast path for nodes whose children are both inner nodes (no leaf or empty-child logic at all), and
// nested uniform branches instead of combined predicates.
#ifdef FFX_BINCHECK
// self-check build (-DFFX_BINCHECK): [0] binned primary walks whose result differs from the tree walk's (per wave), [1] first-record latch,
// [2..4] the first one (pixel, lane; primitives; slots); [8..11] projector shadow walks; [16..19] spot shadow walks
__device__ unsigned long long g_ffx_chk[24];
#endif
#ifdef FFX_STATS
// debug build only (-DFFX_STATS): per-launch totals of packet walks / node steps / triangle tests
__device__ unsigned long long g_ffx_stats[48]; // [32..47]: tile bins (closest-hit 32.., any-hit 40..: walks served, chunk steps, exact tests, second barycentric, fallbacks to the tree)
#define FFX_STAT(i) do { if (threadIdx.x % 64 == 0) atomicAdd(&g_ffx_stats[i], 1ull); } while (0)
#define FFX_STAT_MAX(i, v) do { if (threadIdx.x % 64 == 0) atomicMax(&g_ffx_stats[i], (unsigned long long)(v)); } while (0)
#else
#define FFX_STAT(i) do { } while (0)
#define FFX_STAT_MAX(i, v) do { } while (0)
#endif
#ifdef FFX_TIMERS // separate debug build (-DFFX_TIMERS): the counters above perturb the timing by two orders of magnitude
// phase timers (shader clock, s_memtime): FFX_TSTART(t); ... FFX_TSTOP(t, slot) adds the elapsed cycles of this wave to slot
// (accumulated per workgroup in LDS and flushed once at the end of the kernel: a global atomic per stamp
// would sit in front of every s_waitcnt vmcnt(0) of the walk and measure itself)
__device__ unsigned long long g_ffx_tim[32];
static __shared__ unsigned long long s_ffx_tim[32];
#define FFX_TSTART(t) unsigned long long t = __builtin_amdgcn_s_memtime()
#define FFX_TSTOP(t, i) do { const unsigned long long t_now_ = __builtin_amdgcn_s_memtime(); if (threadIdx.x % 64 == 0) s_ffx_tim[i] += t_now_ - t; t = t_now_; } while (0)
#define FFX_TINIT() do { if (threadIdx.x < 32) s_ffx_tim[threadIdx.x] = 0ull; __builtin_amdgcn_wave_barrier(); } while (0)
#define FFX_TFLUSH() do { __builtin_amdgcn_wave_barrier(); if (threadIdx.x < 32 && s_ffx_tim[threadIdx.x] != 0ull) atomicAdd(&g_ffx_tim[threadIdx.x], s_ffx_tim[threadIdx.x]); } while (0)
#else
#define FFX_TSTART(t) do { } while (0)
#define FFX_TSTOP(t, i) do { } while (0)
#define FFX_TINIT() do { } while (0)
#define FFX_TFLUSH() do { } while (0)
#endif
template <bool ANY, int OCT>
__device__ __forceinline__ void traverse_packet_oct(const BvhNode *__restrict__ nodes, const TriApex *__restrict__ recs, const RayOct &rb, float scale, v3 d,
                                                    float tmin, float tmax, wmask active, Hit &h, bool &found) {
  h.t = msel(active, tmax, -INFINITY); // an inactive ray fails every slab test
  // the hit distance in box-test units, widened by 2 ulp (a box touching the current hit must still be
  // entered: an equal-t triangle with a smaller id may be inside); refreshed after every leaf
  const float sw = scale * 1.0000002f;
  float hts = h.t * sw;
  h.prim = -1;
  h.shape = -1;
  h.slot = -1;
  wmask occluded = 0; // ANY: rays that have found an occluder
  int stack_v = 0;
  int sp = 0;
  int cur = 0; // wave-uniform throughout: node fetches are scalar loads
  const char *nbase = reinterpret_cast<const char *>(nodes);
  const char *rbase = reinterpret_cast<const char *>(recs);
  const RaySlab rs = make_rayslab(rb);
  FFX_STAT(ANY ? 4 : 0);
  while (true) {
    FFX_STAT(ANY ? 5 : 1);
    // the whole 64-byte node with ONE scalar load (s_load_dwordx16); the empty asm makes all 16 dwords
    // live, otherwise the compiler fetches the 14 it needs with three loads (x8 + x4 + x2) — and scalar
    // loads share the issue port that co-limits this loop
    const v16i nd = *reinterpret_cast<const v16i *>(nbase + ((uint32_t)cur << 6));
    asm volatile("" ::"s"(nd));
    const int c0 = nd.sc, c1 = nd.sd;
    float t0, t1;
    const float lo0[3] = {__int_as_float(nd.s0), __int_as_float(nd.s1), __int_as_float(nd.s2)};
    const float hi0[3] = {__int_as_float(nd.s3), __int_as_float(nd.s4), __int_as_float(nd.s5)};
    const float lo1[3] = {__int_as_float(nd.s6), __int_as_float(nd.s7), __int_as_float(nd.s8)};
    const float hi1[3] = {__int_as_float(nd.s9), __int_as_float(nd.sa), __int_as_float(nd.sb)};
    wmask g0 = slab_oct<OCT>(lo0, hi0, rs, hts, t0);
    wmask g1 = slab_oct<OCT>(lo1, hi1, rs, hts, t1);
    if (__builtin_expect((c0 | c1) < 0, 0)) { // a leaf or an empty child on at least one side (2 of 27 steps)
      FFX_STAT(ANY ? 12 : 8);
#pragma unroll
      for (int side = 0; side < 2; ++side) {
        const int c = side ? c1 : c0;
        if (c >= 0) continue;
        const wmask hs = side ? g1 : g0;
        if (side) g1 = 0ull; else g0 = 0ull; // never descended into
        if (c == FFX_EMPTY_CHILD) continue;
        if (hs == 0ull) continue;
        const uint32_t lc = (uint32_t)~c;
        const uint32_t first = lc >> 3, count = (lc & 7u) + 1u;
        uint32_t roff = first * 48u;
        for (uint32_t i = 0; i < count; ++i, roff += 48u) {
          FFX_STAT(ANY ? 6 : 2);
          // the 48-byte apex record with two scalar loads (x8 + x4) issued together, instead of four staged ones
          const v8i r8 = *reinterpret_cast<const v8i *>(rbase + roff);
          const v4i r4 = *reinterpret_cast<const v4i *>(rbase + roff + 32);
          asm volatile("" ::"s"(r8), "s"(r4));
          const int prim = r4.z; // (the shape id is read back through h.slot)
          // apex record: det = d.A, U = d.B, V = d.C, t = T/det; staged with wave-uniform early-outs
          // (same arithmetic and acceptance rule as tri_hit_apex)
          const v3 A = V3(__int_as_float(r8.s0), __int_as_float(r8.s1), __int_as_float(r8.s2));
          const v3 B = V3(__int_as_float(r8.s3), __int_as_float(r8.s4), __int_as_float(r8.s5));
          const v3 C = V3(__int_as_float(r8.s6), __int_as_float(r8.s7), __int_as_float(r4.x));
          const float det = vdot(d, A);
          const float U = vdot(d, B);
          const wmask neg = m_lt(det, 0.f);
          const float detA = fabsf(det);
          const float Us = msel(neg, -U, U);
          const wmask p1 = hs & m_gt(detA, 0.f) & m_ge(Us, 0.f) & m_le(Us, detA);
          if (p1 == 0ull) continue;
          FFX_STAT(ANY ? 7 : 3);
          const float Vv = vdot(d, C);
          const float Vs = msel(neg, -Vv, Vv);
          const wmask p2 = p1 & m_ge(Vs, 0.f) & m_le(Us + Vs, detA);
          if (p2 == 0ull) continue;
          const float T = __int_as_float(r4.y);
          const float t = msel(neg, -T, T) / detA;
          const wmask hit = p2 & m_gt(t, tmin);
          if (ANY) {
            const wmask occ = hit & m_lt(t, tmax);
            occluded |= occ;
            msel_into(h.t, occ, -INFINITY);
          } else {
            // h.t <= tmax always and h.prim == -1 (the largest unsigned) until the first hit, so
            // (t <= tmax) & (no hit yet | t < h.t | (t == h.t & prim < h.prim)) reduces to:
            const wmask better = hit & (m_lt(t, h.t) | (m_eq(t, h.t) & m_ult((uint32_t)prim, (uint32_t)h.prim)));
            msel_into(h.t, better, t);
            msel_into(h.prim, better, prim);
            msel_into(h.slot, better, (int)(first + i));
          }
        }
      }
      if (ANY && (active & ~occluded) == 0ull) break; // every ray of the packet is decided
      // the hit distance may have shrunk: re-test the (at most one) remaining inner child
      hts = h.t * sw;
      if (g0 != 0ull) g0 &= m_le(t0, hts);
      if (g1 != 0ull) g1 &= m_le(t1, hts);
    }
    if (g0 != 0ull) {
      if (g1 != 0ull) {
        // both: visit first the child that most lanes enter first, push the other
        FFX_STAT(ANY ? 13 : 9);
        const wmask lt = m_lt(t1, t0);
        const bool swap = wpop(g1 & (~g0 | lt)) > wpop(g0 & ~(g1 & lt));
        push_lane(stack_v, swap ? c0 : c1, sp);
        ++sp;
        cur = swap ? c1 : c0;
      } else {
        cur = c0;
      }
    } else if (launder_mask(g1) != 0ull) { // (laundered: tested here, not hoisted above the g0 branch as a lane mask)
      cur = c1;
    } else {
      if (sp == 0) break;
      FFX_STAT(ANY ? 14 : 10);
      --sp;
      cur = __builtin_amdgcn_readlane(stack_v, sp);
    }
  }
  found = __builtin_amdgcn_inverse_ballot_w64(occluded);
}

// ------------------------------------------------------------------------------------------ 64-wide packet walk
// tools/ubench/issue_rates.hip (profiles/r2_issue_rates.txt) measured what bounds the binary packet walk
// above: on gfx950 a VALU instruction with an SGPR source operand, and every min/max/compare/select,
// issues once per ~4.3 cycles per SIMD, while mul/add/fma on VGPR operands issue every ~2.3 — and the
// binary walk's box test is ten 4.3-cycle instructions PER BOX, each computing the same answer in 64 lanes
// (the rays of a one-pixel packet are practically one ray).  This walk turns the lanes around for the box
// tests: a node has up to 64 children and LANE j TESTS CHILD j against the packet as a whole,
//     enter >= max_a (near_plane_a - o_a) * N_a        leave <= min_a (far_plane_a - o_a) * F_a
// with N_a / F_a the smallest / largest |1/d_a| of the packet's rays (they share their origin o, so the
// interval test costs exactly what a single ray costs: six fma on VGPR operands + four 4.3-cycle ops for 64
// boxes).  Triangles sit in clusters of up to 64 consecutive leaf slots whose boxes are tested the same way;
// only the triangles that survive are tested exactly, with the lanes back on the rays (same apex test, same
// acceptance rule: results are identical to the binary walk, closest hit with the primitive-id tie-break
// is order independent).  Boxes live on a 16-bit grid (ffx_common.h: WideChild, 16 B per child: a node is one
// coalesced 1 KB load); the grid is folded into the packet constants once per walk, so de-quantisation is
// six integer-to-float conversions per step.  The traversal stack (reference, entry distance) is in LDS.
// `elems`: the wide nodes (64 children each) followed by the triangle boxes in leaf-slot order — ONE array of
// 16-byte elements, so that a reference (cluster << 31 | element << 6 | count - 1) addresses both kinds alike;
// `tq0` = element index of leaf slot 0.
struct WideScene { const WideChild *elems; const WideHdr *hdr; int32_t root; uint32_t tq0; };
#define FFX_WSTACK (63 * FFX_WIDE_MAX_DEPTH + 6)
// per-walk constants, uniform across the wave: tn_a = fma(q_near_a, mN_a, -kN_a), tf_a = fma(q_far_a, mF_a, -kF_a)
// (GEN walks only) a second entry term per axis from the FAR plane, tg_a = fma(q_far_a, mG_a, -kG_a): an axis on
// which the packet's rays disagree in sign has two entry bounds and no exit bound (see make_widepk).
struct WidePk { v3 mN, kN, mF, kF, mG, kG; wmask neg[3]; };

// wave-wide min / max of the bit patterns of NON-NEGATIVE floats (they order like unsigned integers): four
// fused DPP steps inside each row of 16 lanes (quad swap, quad-pair swap, half-row mirror, row mirror: one
// VALU instruction each; the s_nop covers the "VALU wrote the VGPR a DPP op reads" hazard the compiler
// cannot see inside an asm statement), then the four row results are combined on the scalar ALU.
#define FFX_DPP_STEP(OP, CTRL) asm("s_nop 1\n\t" OP " %0, %0, %0 " CTRL " row_mask:0xf bank_mask:0xf" : "+v"(v))
__device__ __forceinline__ uint32_t smin_u32(uint32_t a, uint32_t b) { uint32_t r; asm("s_min_u32 %0, %1, %2" : "=s"(r) : "s"(a), "s"(b) : "scc"); return r; }
__device__ __forceinline__ uint32_t smax_u32(uint32_t a, uint32_t b) { uint32_t r; asm("s_max_u32 %0, %1, %2" : "=s"(r) : "s"(a), "s"(b) : "scc"); return r; }
template <bool MAX>
__device__ __forceinline__ uint32_t wave_reduce_nn(uint32_t v) {
  // ONE asm block (the compiler pads every asm statement with its own hazard nop): four butterfly steps inside the rows,
  // row_bcast:15 / :31 carry the row results to lane 63 (see wave_reduce3_nn) — one readlane, no scalar min/max
#define FFX_R1(OP, CTRL, MASK) "s_nop 1\n\t" OP " %0, %0, %0 " CTRL " row_mask:" MASK " bank_mask:0xf\n\t"
#define FFX_R1_ALL(OP) FFX_R1(OP, "quad_perm:[1,0,3,2]", "0xf") FFX_R1(OP, "quad_perm:[2,3,0,1]", "0xf") FFX_R1(OP, "row_half_mirror", "0xf") \
    FFX_R1(OP, "row_mirror", "0xf") FFX_R1(OP, "row_bcast:15", "0xa") FFX_R1(OP, "row_bcast:31", "0xc")
  if (MAX) asm(FFX_R1_ALL("v_max_u32_dpp") : "+v"(v));
  else asm(FFX_R1_ALL("v_min_u32_dpp") : "+v"(v));
  return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
#ifndef FFX_WIDE_FAT
#define FFX_WIDE_FAT 0.04f // see traverse_wide
#endif
// three reductions at once (make_widepk: the three axes).  The chains are interleaved, so that the two instructions
// between a DPP write and the next DPP read of the same register ARE the wait states the hazard asks for (no s_nop per
// step), and the rows are combined by row_bcast:15 / row_bcast:31 (rows 1, 3 take lane 15 of the row before; rows 2, 3
// take lane 31): the wave's result sits in lane 63 — one readlane and no scalar min/max per value.
// Checked against a shuffle reduction on random data by tools/ubench/reduce3_check.hip.
#define FFX_R3_STEP(OP, CTRL, MASK)                               \
  OP " %0, %0, %0 " CTRL " row_mask:" MASK " bank_mask:0xf\n\t" \
  OP " %1, %1, %1 " CTRL " row_mask:" MASK " bank_mask:0xf\n\t" \
  OP " %2, %2, %2 " CTRL " row_mask:" MASK " bank_mask:0xf\n\t"
#define FFX_R3_ALL(OP)                                                                                                                            \
  "s_nop 1\n\t" FFX_R3_STEP(OP, "quad_perm:[1,0,3,2]", "0xf") FFX_R3_STEP(OP, "quad_perm:[2,3,0,1]", "0xf") FFX_R3_STEP(OP, "row_half_mirror", "0xf") \
      FFX_R3_STEP(OP, "row_mirror", "0xf") FFX_R3_STEP(OP, "row_bcast:15", "0xa") FFX_R3_STEP(OP, "row_bcast:31", "0xc")
template <bool MAX>
__device__ __forceinline__ void wave_reduce3_nn(uint32_t (&v)[3]) {
  if (MAX) asm(FFX_R3_ALL("v_max_u32_dpp") : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]));
  else asm(FFX_R3_ALL("v_min_u32_dpp") : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]));
#pragma unroll
  for (int a = 0; a < 3; ++a) v[a] = (uint32_t)__builtin_amdgcn_readlane((int)v[a], 63);
}
__device__ __forceinline__ uint32_t mbcnt64(wmask m) { return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)); }
__device__ __forceinline__ int wff1(wmask m) { return __builtin_ctzll(m); }

// the packet constants of one walk.  `aid[r]` = |scaled 1/d| of ray r per axis, `oct[r]` its direction signs,
// `oct0` the packet's octant (signs of its first active ray), `mixed` the axes on which the active rays
// disagree in sign.  On a mixed axis the packet is a wedge that opens both ways: it reaches a box beyond
// `lo` no earlier than (lo - o) * min(1/d+) and a box before `hi` no earlier than (o - hi) * min(1/|d-|), and
// it never leaves the slab for good — two entry bounds, no exit bound.  (Dropping the axis instead is
// conservative too, but a packet with two mixed axes — a pixel next to the image centre — then walks every
// box in its depth range: one such wave took 7 ms.)
template <int R>
__device__ __forceinline__ WidePk make_widepk(const WideHdr *__restrict__ hdr, v3 o, const v3 (&aid)[R], const uint32_t (&oct)[R], const wmask (&active)[R],
                                              uint32_t oct0, uint32_t mixed, float &spread) {
  float dmax_all = 0.f, dspread = 0.f; // largest |d_a| and largest (max |d_a| - min |d_a|) over the axes, in units of 1/scale
#if FFX_WIDE_F32
  const float org[3] = {0.f, 0.f, 0.f}, step[3] = {1.f, 1.f, 1.f}; // float boxes: no grid to fold in
#else
  const float org[3] = {hdr->org[0], hdr->org[1], hdr->org[2]}, step[3] = {hdr->step[0], hdr->step[1], hdr->step[2]};
#endif
  const float oo[3] = {o.x, o.y, o.z};
  float mN[3], kN[3], mF[3], kF[3], mG[3], kG[3];
  const float k22 = 2.384185791015625e-07f, k21 = 4.76837158203125e-07f, kw = 1.0000004f;
  WidePk pk;
  // smallest / largest |1/d_a| over the active rays; on a mixed axis: the smallest of each sign
  uint32_t lo[3], hi[3], lo2[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const bool is_mixed = (mixed >> a) & 1u;
    lo[a] = 0x7f800000u; hi[a] = 0u; lo2[a] = 0x7f800000u; // +inf, 0, +inf
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const uint32_t v = __float_as_uint(a == 0 ? aid[r].x : (a == 1 ? aid[r].y : aid[r].z));
      const wmask negr = __builtin_amdgcn_uicmp((oct[r] >> a) & 1u, 0u, 33);
      const wmask first = is_mixed ? (active[r] & ~negr) : active[r];
      const uint32_t vl = msel(first, v, 0x7f800000u), vh = msel(active[r], v, 0u), vl2 = msel(active[r] & negr, v, 0x7f800000u);
      if (R == 1) { lo[a] = vl; hi[a] = vh; lo2[a] = vl2; } // (aid is finite: safe_rcp_dir clamps tiny components)
      else {
        lo[a] = lo[a] < vl ? lo[a] : vl;
        hi[a] = hi[a] > vh ? hi[a] : vh;
        lo2[a] = lo2[a] < vl2 ? lo2[a] : vl2;
      }
    }
  }
  if (mixed == 0u) { // (a compile-time constant at the call sites) the three axes in interleaved chains
    wave_reduce3_nn<false>(lo);
    wave_reduce3_nn<true>(hi);
#if FFX_WIDE_F32
    // Everything from here on is wave-uniform float arithmetic, which this machine can only do on the vector ALU at the
    // 4-cycle rate of scalar-operand instructions: it is written to be short.  Signs are applied to the bit patterns
    // (scalar ALU); the two paddings of each side are one fma (2^-20 >= 2^-22 + 2^-21 + the roundings they cover:
    // of o * m, of m * k and of the box test's own fma); the fat-packet measure
    //     max_a (1/mn_a - 1/mx_a) / max_a (1/mn_a) <= FAT   <=>   for all a:  M (mx_a - mn_a) <= FAT mn_a mx_a,  M = min_a mn_a
    // needs no reciprocal.
    const float PAD = 9.5367431640625e-07f; // 2^-20
    const float M = fminf(__uint_as_float(lo[0]), fminf(__uint_as_float(lo[1]), __uint_as_float(lo[2])));
    bool fat = false;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const uint32_t sgn = ((oct0 >> a) & 1u) << 31;
      pk.neg[a] = sgn ? ~0ull : 0ull;
      const float mn = __uint_as_float(lo[a]), mx = __uint_as_float(hi[a]);
      fat |= M * (mx - mn) > FFX_WIDE_FAT * (mn * mx);
      const float sN = __uint_as_float(lo[a] ^ sgn), sFk = __uint_as_float(hi[a] ^ sgn) * kw;
      const float oidn = oo[a] * sN, oidf = oo[a] * sFk;
      mN[a] = sN;
      kN[a] = fmaf(fabsf(oidn), PAD, oidn);   // the entry can only move earlier
      mF[a] = sFk;
      kF[a] = fmaf(-fabsf(oidf), PAD, oidf);  // the exit only later (on top of the far-side widening kw)
      mG[a] = 0.f;
      kG[a] = 1e30f;
    }
    pk.mN = V3(mN[0], mN[1], mN[2]); pk.kN = V3(kN[0], kN[1], kN[2]);
    pk.mF = V3(mF[0], mF[1], mF[2]); pk.kF = V3(kF[0], kF[1], kF[2]);
    pk.mG = V3(mG[0], mG[1], mG[2]); pk.kG = V3(kG[0], kG[1], kG[2]);
    spread = fat ? 1.0f : 0.0f;
    return pk;
#endif
  } else {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      lo[a] = wave_reduce_nn<false>(lo[a]);
      if ((mixed >> a) & 1u) lo2[a] = wave_reduce_nn<false>(lo2[a]);
      else hi[a] = wave_reduce_nn<true>(hi[a]);
    }
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const bool is_mixed = (mixed >> a) & 1u;
    const float mn = __uint_as_float(lo[a]);
    const bool negd = !is_mixed && ((oct0 >> a) & 1u);
    pk.neg[a] = negd ? ~0ull : 0ull;
    const float sN = negd ? -mn : mn;
    // near side (same padding as make_rayoct: the entry can only move earlier)
    const float oidn = oo[a] * sN, cN = oidn + fabsf(oidn) * k22;
    const float gn = org[a] * sN;
    mN[a] = step[a] * sN;
    kN[a] = (cN - gn) + (fabsf(cN) + fabsf(gn)) * k21;
    if (!is_mixed) {
      // far side, widened (make_rayslab)
      const float mx = __uint_as_float(hi[a]);
      const float dhi = __builtin_amdgcn_rcpf(mn), dlo = __builtin_amdgcn_rcpf(mx); // |d_a| range of the packet
      dmax_all = fmaxf(dmax_all, dhi);
      dspread = fmaxf(dspread, dhi - dlo);
      const float sF = negd ? -mx : mx;
      const float oidf = oo[a] * sF, cF = (oidf - fabsf(oidf) * k22) * kw;
      const float sFk = sF * kw, gf = org[a] * sFk;
      mF[a] = step[a] * sFk;
      kF[a] = (cF - gf) - (fabsf(cF) + fabsf(gf)) * k21;
      mG[a] = 0.f;
      kG[a] = 1e30f; // tg_a = -1e30
    } else {
      // second entry bound from the far (= hi) plane: (o - x) * m2 = fma(q, -step * m2, -(org * m2 - o * m2))
      const float m2 = __uint_as_float(lo2[a]);
      const float og = oo[a] * m2, gg = org[a] * m2;
      mG[a] = -(step[a] * m2);
      kG[a] = (gg - og) + (fabsf(gg) + fabsf(og)) * k21;
      mF[a] = 0.f;
      kF[a] = -1e30f; // tf_a = +1e30
    }
  }
  pk.mN = V3(mN[0], mN[1], mN[2]); pk.kN = V3(kN[0], kN[1], kN[2]);
  pk.mF = V3(mF[0], mF[1], mF[2]); pk.kF = V3(kF[0], kF[1], kF[2]);
  pk.mG = V3(mG[0], mG[1], mG[2]); pk.kG = V3(kG[0], kG[1], kG[2]);
  spread = dspread * __builtin_amdgcn_rcpf(dmax_all);
  return pk;
}

// returns false if the walk was abandoned because it exceeded its budget of steps / exact tests (the caller
// then repeats it on the binary walk, whose per-ray box tests cope with incoherent packets)
#ifndef FFX_WIDE_MAX_WORK
#define FFX_WIDE_MAX_WORK 96 // steps + exact triangle tests of one walk (typical: 7 + 7; measured 24 / 48 / 64 / 96 / 192: 0.77 / 0.70 / 0.65 / 0.65 / 0.65 ms vocal fold, 4.2 / 3.5 / 3.4 / 3.3 / 3.3 ms colon)
#endif
template <bool ANY, int OCT, int R>
__device__ __forceinline__ bool traverse_wide_oct(const WideScene &ws, const TriApex *__restrict__ recs, const WidePk &pk, const v3 (&d)[R], const float (&tmin)[R],
                                                  const float (&tmax)[R], const float (&sw)[R], const wmask (&active)[R], Hit (&h)[R], wmask (&occluded)[R],
                                                  uint2 *__restrict__ stack) {
  const uint32_t lane16 = (threadIdx.x & 63u) << FFX_WIDE_ELEM_SHIFT; // byte offset of this lane's element within a node / cluster
  int budget = FFX_WIDE_MAX_WORK;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    h[r].t = msel(active[r], tmax[r], -INFINITY);
    h[r].prim = -1;
    h[r].shape = -1;
    h[r].slot = -1;
    occluded[r] = 0ull;
  }
  // the packet's hit distance in box-test units (the largest over its rays: a box matters while ANY ray can
  // still reach it), as the bit pattern of a non-negative float; refreshed whenever a ray's hit improves
  auto packet_hts = [&]() -> uint32_t {
    float m = 0.f;
#pragma unroll
    for (int r = 0; r < R; ++r) m = fmaxf(m, h[r].t * sw[r]); // -inf (inactive / occluded) drops out against 0
    return wave_reduce_nn<true>(__float_as_uint(m));
  };
  // before any hit every active ray's distance is its tmax, which the unit of the box test maps to 0.999 * (1 + a few
  // ulp) by construction (sw = 0.999 / tmax * 1.0000002): a constant just above that is a valid (conservative) packet
  // bound — and saves the wave-wide maximum at the start of every walk.  It stays below 1, the value the clamp gives
  // boxes beyond the rays' range.
  uint32_t hb = __float_as_uint(0.9991f);
  int sp = 0;
  int32_t cur = ws.root;
  const char *ebase = reinterpret_cast<const char *>(ws.elems);
  const char *rbase = reinterpret_cast<const char *>(recs);
  FFX_STAT(ANY ? 4 : 0);
#ifdef FFX_STATS
  unsigned n_steps = 0, n_tris = 0;
#endif
  FFX_TSTART(tw);
  // ONE loop with ONE exit: every way out of a step (walk complete, every ray decided, budget spent) is folded into the
  // wave-uniform flag `go`, and every way on (descend, pop) into `nxt`.  Written with returns / continues from inside the
  // nested loops, the compiler's control-flow structuriser turned the exits into state variables and a dispatch
  // chain of ~15 scalar instructions and half a dozen branches per step.
  bool go, completed = true;
  do {
    go = true;
    bool descend = false;
    int32_t nxt = 0;
    FFX_STAT(ANY ? 5 : 1);
#ifdef FFX_STATS
    ++n_steps;
    FFX_STAT_MAX(ANY ? 15 : 11, ((unsigned long long)n_steps << 32) | n_tris);
    if (n_steps == 13) FFX_STAT(ANY ? 28 : 24); // walks with more than 12 steps, and the steps beyond
    if (n_steps > 12) FFX_STAT(ANY ? 29 : 25);
#endif
    // ---- fetch: lane j reads child j of an inner node or box j of a cluster: one 16-byte load per lane, no
    // divergence (lanes beyond the count re-read the last element and are masked out of the result)
    const uint32_t cnt1 = (uint32_t)cur & 63u;
    const uint32_t eoff = ((uint32_t)cur & 0x7fffffc0u) >> (6 - FFX_WIDE_ELEM_SHIFT); // element index * element size
#if FFX_WIDE_F32
    // lane j reads element j whatever the count: the unused children of a wide node hold inverted boxes (never hit), a
    // cluster's run is followed by other triangles' boxes (masked below; the array is padded by 64 elements) — so the
    // scalar side of the step is three instructions: the lane's byte offset is ONE vector add on a loop-invariant base.
    const uint32_t voff = eoff + lane16;
    const float4 qa = *reinterpret_cast<const float4 *>(ebase + voff);
    const uint4 q = *reinterpret_cast<const uint4 *>(ebase + voff + 16); // hi.y, hi.z, ref, pad  (q.z = ref)
    const float lx = qa.x, ly = qa.y, lz = qa.z, hx = qa.w, hy = __uint_as_float(q.x), hz = __uint_as_float(q.y);
#define FFX_QREF q.z
#else
    const uint32_t lo16 = lane16 < (cnt1 << FFX_WIDE_ELEM_SHIFT) ? lane16 : (cnt1 << FFX_WIDE_ELEM_SHIFT);
    const wmask lanes = ~0ull >> (63u - cnt1);
    const uint4 q = *reinterpret_cast<const uint4 *>(ebase + eoff + lo16);
    const float lx = (float)(q.x & 0xffffu), ly = (float)(q.x >> 16), lz = (float)(q.y & 0xffffu);
    const float hx = (float)(q.y >> 16), hy = (float)(q.z & 0xffffu), hz = (float)(q.z >> 16);
#define FFX_QREF q.w
#endif
    float nx, ny, nz, fx, fy, fz, tn;
    if constexpr (OCT < 8) { // which plane is entered first is known at compile time
      nx = (OCT & 1) ? hx : lx; fx = (OCT & 1) ? lx : hx;
      ny = (OCT & 2) ? hy : ly; fy = (OCT & 2) ? ly : hy;
      nz = (OCT & 4) ? hz : lz; fz = (OCT & 4) ? lz : hz;
      tn = vmax3_sat(fmaf(nx, pk.mN.x, -pk.kN.x), fmaf(ny, pk.mN.y, -pk.kN.y), fmaf(nz, pk.mN.z, -pk.kN.z));
    } else { // generic instance (packets with mixed direction signs): run-time plane choice, second entry terms
      nx = msel(pk.neg[0], hx, lx); fx = msel(pk.neg[0], lx, hx);
      ny = msel(pk.neg[1], hy, ly); fy = msel(pk.neg[1], ly, hy);
      nz = msel(pk.neg[2], hz, lz); fz = msel(pk.neg[2], lz, hz);
      const float t1 = fmaxf(fmaxf(fmaf(nx, pk.mN.x, -pk.kN.x), fmaf(ny, pk.mN.y, -pk.kN.y)), fmaf(nz, pk.mN.z, -pk.kN.z));
      const float t2 = fmaxf(fmaxf(fmaf(fx, pk.mG.x, -pk.kG.x), fmaf(fy, pk.mG.y, -pk.kG.y)), fmaf(fz, pk.mG.z, -pk.kG.z));
      tn = vmax3_sat(t1, t2, t2);
    }
    const float tf = vmin2_s(vmin3(fmaf(fx, pk.mF.x, -pk.kF.x), fmaf(fy, pk.mF.y, -pk.kF.y), fmaf(fz, pk.mF.z, -pk.kF.z)), hb);
#if FFX_WIDE_F32
    wmask hit = m_le(tn, tf);
#else
    wmask hit = m_le(tn, tf) & lanes;
#endif
    FFX_TSTOP(tw, ANY ? 8 : 0);
    if (cur < 0) {
#if FFX_WIDE_F32
      hit &= ~0ull >> (63u - cnt1); // only the cluster's own triangles
#endif
      // ---- cluster: the surviving triangles are tested exactly, lanes back on the rays
      FFX_STAT(ANY ? 12 : 8);
      const uint32_t slot0 = (eoff >> FFX_WIDE_ELEM_SHIFT) - ws.tq0;
#ifdef FFX_EXP_ANY_NOTRIS // timing experiment: any-hit walks without the exact triangle tests
      if (ANY) hit = 0ull;
#endif
      while (hit != 0ull) {
        const uint32_t j = (uint32_t)wff1(hit);
        asm("s_bitset0_b64 %0, %1" : "+s"(hit) : "s"(j)); // hit &= hit - 1 in one scalar instruction instead of three
        --budget;
        FFX_STAT(ANY ? 6 : 2);
#ifdef FFX_STATS
        ++n_tris;
        if (n_tris == 17) FFX_STAT(ANY ? 20 : 16);  // walks with more than 16 exact tests ...
        if (n_tris > 16) FFX_STAT(ANY ? 21 : 17);   // ... and the tests beyond the 16th
        if (n_tris == 65) FFX_STAT(ANY ? 22 : 18);
        if (n_tris > 64) FFX_STAT(ANY ? 23 : 19);
#endif
        const uint32_t slot = slot0 + j;
        const uint32_t roff = slot * 48u;
        const v8i r8 = *reinterpret_cast<const v8i *>(rbase + roff);
        const v4i r4 = *reinterpret_cast<const v4i *>(rbase + roff + 32);
        asm volatile("" ::"s"(r8), "s"(r4));
        const int prim = r4.z;
        const v3 A = V3(__int_as_float(r8.s0), __int_as_float(r8.s1), __int_as_float(r8.s2));
        const v3 B = V3(__int_as_float(r8.s3), __int_as_float(r8.s4), __int_as_float(r8.s5));
        const v3 C = V3(__int_as_float(r8.s6), __int_as_float(r8.s7), __int_as_float(r4.x));
        const float T = __int_as_float(r4.y);
        wmask improved = 0ull;
#pragma unroll
        for (int r = 0; r < R; ++r) {
          // apex test, staged with wave-uniform early-outs (identical arithmetic to traverse_packet_oct)
          const float det = vdot(d[r], A);
          const float U = vdot(d[r], B);
          const wmask neg = m_lt(det, 0.f);
          const float detA = fabsf(det);
          const float Us = msel(neg, -U, U);
          const wmask alive = ANY ? (active[r] & ~occluded[r]) : active[r];
          const wmask p1 = alive & m_gt(detA, 0.f) & m_ge(Us, 0.f) & m_le(Us, detA);
          if (p1 == 0ull) continue;
          FFX_STAT(ANY ? 7 : 3);
          const float Vv = vdot(d[r], C);
          const float Vs = msel(neg, -Vv, Vv);
          const wmask p2 = p1 & m_ge(Vs, 0.f) & m_le(Us + Vs, detA);
          if (p2 == 0ull) continue;
          const float t = msel(neg, -T, T) / detA;
          const wmask hitm = p2 & m_gt(t, tmin[r]);
          if (ANY) {
            const wmask occ = hitm & m_lt(t, tmax[r]);
            occluded[r] |= occ;
            msel_into(h[r].t, occ, -INFINITY);
            improved |= occ;
          } else {
            const wmask better = hitm & (m_lt(t, h[r].t) | (m_eq(t, h[r].t) & m_ult((uint32_t)prim, (uint32_t)h[r].prim)));
            msel_into(h[r].t, better, t);
            msel_into(h[r].prim, better, prim);
            msel_into(h[r].slot, better, (int)slot);
            improved |= better;
          }
        }
        if (improved != 0ull) {
          if (ANY) {
            wmask left = 0ull;
#pragma unroll
            for (int r = 0; r < R; ++r) left |= active[r] & ~occluded[r];
            if (left == 0ull) { go = false; hit = 0ull; } // every ray of the packet is decided: leave the cluster, end the walk
            // (the undecided rays keep their full length: the packet's hit distance does not change)
          } else {
            hb = packet_hts();
            hit &= m_le(tn, __uint_as_float(hb)); // the remaining triangles of this cluster against the shorter rays
          }
        }
      }
      FFX_TSTOP(tw, ANY ? 9 : 1);
    } else if (hit != 0ull) {
      FFX_STAT(ANY ? 13 : 9);
      // ---- inner node: descend into the child entered first, push the others with their entry distances.
      // key = entry distance (6 low mantissa bits dropped: it only orders and culls, conservatively) | lane
      const uint32_t key = (__float_as_uint(tn) & ~63u) | (threadIdx.x & 63u);
      uint32_t near_lane = (uint32_t)wff1(hit);
      if (wpop(hit) != 1) { // several children: nearest first, the others onto the stack
#ifndef FFX_EXP_LANE_ORDER // (experiment: descend in lane order instead of nearest-first)
        near_lane = wave_reduce_nn<false>(msel(hit, key, 0xffffffffu)) & 63u;
#endif
        const wmask others = hit & ~(1ull << near_lane);
        if (__builtin_amdgcn_inverse_ballot_w64(others)) stack[sp + (int)mbcnt64(others)] = make_uint2(FFX_QREF, key);
        sp += wpop(others);
      }
      nxt = __builtin_amdgcn_readlane((int)FFX_QREF, (int)near_lane);
      descend = true;
      FFX_TSTOP(tw, ANY ? 10 : 2);
    }
    if (!descend && go) {
      // ---- pop: skip entries the rays can no longer reach; an empty stack ends the walk
      go = false;
      while (sp > 0) {
        FFX_STAT(ANY ? 14 : 10);
        --sp;
        const uint2 e = stack[sp];
        const uint32_t ref = (uint32_t)__builtin_amdgcn_readfirstlane((int)e.x), etn = (uint32_t)__builtin_amdgcn_readfirstlane((int)e.y) & ~63u;
        if (etn <= hb) { nxt = (int32_t)ref; go = true; break; }
      }
      FFX_TSTOP(tw, ANY ? 11 : 3);
    }
    if (--budget < 0) { completed = false; go = false; } // (a walk that ends on its last allowed step is repeated by the caller: harmless)
    cur = nxt;
  } while (go);
  return completed;
}

#undef FFX_QREF
// wide walk of R rays per lane that share their origin `o` (the apex the records `recs` were prepared for).
// Packets the interval test is bad at go to the binary walk instead (its box tests are per ray): packets whose
// rays disagree on a direction sign, packets whose directions spread more than FFX_WIDE_FAT of their length
// (the samples of a pixel on a depth discontinuity, seen from an emitter: a fan whose bounding wedge
// contains hundreds of boxes no ray comes near — colon, 1024^2: 15.9 ms with every packet on the wide walk,
// 4.2 ms on the binary walk), and walks that exceed FFX_WIDE_MAX_WORK.  Results are identical either way.
template <bool ANY, int R>
__device__ __forceinline__ void traverse_wide(const WideScene &ws, const BvhNode *__restrict__ nodes, const TriApex *__restrict__ recs, const v3 (&o)[R],
                                              const v3 (&d)[R], const float (&tmin)[R], const float (&tmax)[R], const bool (&act)[R], Hit (&h)[R], bool (&found)[R],
                                              uint2 *__restrict__ stack) {
  FFX_TSTART(ts);
  wmask active[R], any_active = 0ull;
#pragma unroll
  for (int r = 0; r < R; ++r) { active[r] = wballot(act[r]); any_active |= active[r]; }
  if (any_active == 0ull) {
#pragma unroll
    for (int r = 0; r < R; ++r) { h[r].t = -INFINITY; h[r].prim = -1; h[r].shape = -1; h[r].slot = -1; found[r] = false; }
    return;
  }
  v3 aid[R];
  float sw[R];
  uint32_t oct[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    // unit of the box test: tmax maps to 0.999 (traverse_packet1)
    const float scale = 0.999f * __builtin_amdgcn_rcpf(tmax[r]);
    const v3 id = V3(safe_rcp_dir(d[r].x) * scale, safe_rcp_dir(d[r].y) * scale, safe_rcp_dir(d[r].z) * scale);
    oct[r] = (__float_as_uint(id.x) >> 31) | ((__float_as_uint(id.y) >> 31) << 1) | ((__float_as_uint(id.z) >> 31) << 2);
    aid[r] = V3(fabsf(id.x), fabsf(id.y), fabsf(id.z));
    sw[r] = scale * 1.0000002f;
  }
  // the packet's octant: that of its first active ray
  uint32_t oct0 = 0;
  {
    bool got = false;
#pragma unroll
    for (int r = 0; r < R; ++r)
      if (!got && active[r] != 0ull) { oct0 = (uint32_t)__builtin_amdgcn_readlane((int)oct[r], wff1(active[r])); got = true; }
  }
  wmask dis = 0ull;
#pragma unroll
  for (int r = 0; r < R; ++r) dis |= active[r] & __builtin_amdgcn_uicmp(oct[r], oct0, 33);
  bool done = false;
  if (dis == 0ull) {
    float spread;
    const WidePk pk = make_widepk<R>(ws.hdr, o[0], aid, oct, active, oct0, 0u, spread);
    FFX_TSTOP(ts, ANY ? 12 : 4);
    if (spread <= FFX_WIDE_FAT) {
      wmask occ[R];
      switch (oct0) { // wave-uniform
        case 0: done = traverse_wide_oct<ANY, 0, R>(ws, recs, pk, d, tmin, tmax, sw, active, h, occ, stack); break;
        case 1: done = traverse_wide_oct<ANY, 1, R>(ws, recs, pk, d, tmin, tmax, sw, active, h, occ, stack); break;
        case 2: done = traverse_wide_oct<ANY, 2, R>(ws, recs, pk, d, tmin, tmax, sw, active, h, occ, stack); break;
        case 3: done = traverse_wide_oct<ANY, 3, R>(ws, recs, pk, d, tmin, tmax, sw, active, h, occ, stack); break;
        case 4: done = traverse_wide_oct<ANY, 4, R>(ws, recs, pk, d, tmin, tmax, sw, active, h, occ, stack); break;
        case 5: done = traverse_wide_oct<ANY, 5, R>(ws, recs, pk, d, tmin, tmax, sw, active, h, occ, stack); break;
        case 6: done = traverse_wide_oct<ANY, 6, R>(ws, recs, pk, d, tmin, tmax, sw, active, h, occ, stack); break;
        default: done = traverse_wide_oct<ANY, 7, R>(ws, recs, pk, d, tmin, tmax, sw, active, h, occ, stack); break;
      }
#pragma unroll
      for (int r = 0; r < R; ++r) found[r] = __builtin_amdgcn_inverse_ballot_w64(occ[r]);
    }
  }
  if (!done) {
    FFX_STAT(ANY ? 30 : 26);
    traverse_packet<ANY, R>(nodes, recs, o, d, tmin, tmax, act, h, found); // binary walk, per-ray box tests
  }
}

// one ray per lane: pick the octant loop if the packet's active rays agree on their direction signs
template <bool ANY>
__device__ __forceinline__ void traverse_packet1(const BvhNode *__restrict__ nodes, const TriApex *__restrict__ recs, const v3 (&o)[1], const v3 (&d)[1],
                                                 const float (&tmin)[1], const float (&tmax)[1], const bool (&active)[1], Hit (&h)[1], bool (&found)[1]) {
  const wmask am = wballot(active[0]);
  if (am == 0ull) { // nothing to trace
    h[0].t = -INFINITY; h[0].prim = -1; h[0].shape = -1; h[0].slot = -1; found[0] = false;
    return;
  }
  // unit of the box test: tmax maps to 0.999, so an entry distance beyond the ray's range — which the
  // clamp turns into exactly 1 — stays above the scaled hit distance and the box is rejected
  const float scale = 0.999f * __builtin_amdgcn_rcpf(tmax[0]);
  const RayOct rb = make_rayoct(o[0], d[0], scale);
  // octant from the reciprocals actually used (a clamped -0.0 component counts as negative)
  const uint32_t oct = (__float_as_uint(rb.id.x) >> 31) | ((__float_as_uint(rb.id.y) >> 31) << 1) | ((__float_as_uint(rb.id.z) >> 31) << 2);
  const uint32_t oct0 = (uint32_t)__builtin_amdgcn_readlane((int)oct, __builtin_ctzll(am)); // octant of the first active lane
  const bool uniform = (am & __builtin_amdgcn_uicmp(oct, oct0, 33)) == 0ull;
  if (uniform) {
    switch (oct0) { // wave-uniform
      case 0: traverse_packet_oct<ANY, 0>(nodes, recs, rb, scale, d[0], tmin[0], tmax[0], am, h[0], found[0]); break;
      case 1: traverse_packet_oct<ANY, 1>(nodes, recs, rb, scale, d[0], tmin[0], tmax[0], am, h[0], found[0]); break;
      case 2: traverse_packet_oct<ANY, 2>(nodes, recs, rb, scale, d[0], tmin[0], tmax[0], am, h[0], found[0]); break;
      case 3: traverse_packet_oct<ANY, 3>(nodes, recs, rb, scale, d[0], tmin[0], tmax[0], am, h[0], found[0]); break;
      case 4: traverse_packet_oct<ANY, 4>(nodes, recs, rb, scale, d[0], tmin[0], tmax[0], am, h[0], found[0]); break;
      case 5: traverse_packet_oct<ANY, 5>(nodes, recs, rb, scale, d[0], tmin[0], tmax[0], am, h[0], found[0]); break;
      case 6: traverse_packet_oct<ANY, 6>(nodes, recs, rb, scale, d[0], tmin[0], tmax[0], am, h[0], found[0]); break;
      default: traverse_packet_oct<ANY, 7>(nodes, recs, rb, scale, d[0], tmin[0], tmax[0], am, h[0], found[0]); break;
    }
  } else {
    traverse_packet<ANY, 1>(nodes, recs, o, d, tmin, tmax, active, h, found);
  }
}

// dispatch: WIDE — the 64-wide walk (default); otherwise the binary walks: octant loops for one ray per
// lane, the generic loop for more.  All packet walks are APEX walks (their rays share the origin o[0] whose
// records `arecs` were written by k_apex_records).
template <bool ANY, int R, bool WIDE>
__device__ __forceinline__ void traverse_packet_any(const BvhNode *__restrict__ nodes, const TriApex *__restrict__ arecs, const WideScene &ws, uint2 *__restrict__ stack,
                                                    const v3 (&o)[R], const v3 (&d)[R], const float (&tmin)[R], const float (&tmax)[R], const bool (&active)[R],
                                                    Hit (&h)[R], bool (&found)[R]) {
#ifdef FFX_EXP_NOWALK // timing experiment: every ray "hits" leaf slot 0 half-way along its range — what everything but the walks costs
#pragma unroll
  for (int r = 0; r < R; ++r) { h[r].t = ANY ? -INFINITY : 0.5f * (tmin[r] + tmax[r]); h[r].prim = ANY ? -1 : 0; h[r].shape = -1; h[r].slot = ANY ? -1 : 0; found[r] = false; }
  return;
#endif
  if constexpr (WIDE) traverse_wide<ANY, R>(ws, nodes, arecs, o, d, tmin, tmax, active, h, found, stack);
  else if constexpr (R == 1 && octant_loops()) traverse_packet1<ANY>(nodes, arecs, o, d, tmin, tmax, active, h, found);
  else traverse_packet<ANY, R>(nodes, arecs, o, d, tmin, tmax, active, h, found);
}

// ------------------------------------------------------------------------------------------ tile bins instead of a tree walk
// (ffx_common.h BinEntry / BinGrid, ffx_bins.hip.)  The rays of a packet leave one apex and cover a rectangle of that apex's image
// plane: one pixel for the primary rays, the (tiny) bounding box of the samples' image points for a packet of shadow rays.  Every
// triangle such a ray can hit is listed in the tile(s) the rectangle touches.  A step loads 64 entries — one per lane, coalesced —
// and tests each against the rectangle: its padded box, then the rectangle's most-inside corner against the three edge functions,
//     max over the rectangle of (n.x x + n.y y + c) = n.x cx + |n.x| hx + n.y cy + |n.y| hy + c      (centre c*, half extents h*)
// four fma per edge with the rectangle as scalar operands, no selects.  The survivors — the triangles whose projection really
// overlaps the pixel, 1 - 3 of them — get the exact apex test of the tree walks (same arithmetic, same acceptance rule, closest hit
// with the primitive-id tie-break: the result does not depend on the order, and is the tree walk's bit for bit).
// No stack, no descent, no packet constants: where the 64-wide walk took 4.2 dependent steps and 100 VALU of set-up per pixel
// this is ~1.3 steps.  tx0..ty1: the tiles the rectangle touches (at most four: the caller falls back to the tree otherwise).
template <bool ANY>
__device__ __forceinline__ void traverse_bins(const char *__restrict__ bbase, const int nx, const int tx0, const int ty0, const int tx1, const int ty1, const float rcx,
                                              const float rcy, const float rhx, const float rhy, const TriApex *__restrict__ recs, const v3 d, const float tmin,
                                              const float tmax, const wmask active, Hit &h, wmask &occluded) {
  h.t = msel(active, tmax, -INFINITY);
  h.prim = -1;
  h.shape = -1;
  h.slot = -1;
  occluded = 0ull;
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t *starts = reinterpret_cast<const uint32_t *>(bbase + 64); // ffx_bin_off_starts()
  const char *ents = bbase + ffx_bin_off_entries();
  const char *rbase = reinterpret_cast<const char *>(recs);
  // the rectangle's bounds for the box test — wave-uniform like centre and half extents: kept in SGPRs (they feed the step as scalar
  // operands; as VGPRs they would be eight more registers live across the exact tests)
  auto uni = [](float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); };
  const float rx0 = uni(rcx - rhx), rx1 = uni(rcx + rhx), ry0 = uni(rcy - rhy), ry1 = uni(rcy + rhy);
  FFX_STAT(ANY ? 40 : 32);
  const float tmax_up = tmax * 1.0000038f; // tmax (1 + 2^-18)
  bool go = true;
  for (int ty = ty0; ty <= ty1 && go; ++ty) {
    for (int tx = tx0; tx <= tx1 && go; ++tx) {
      const uint32_t tile = (uint32_t)(ty * nx + tx);
      const uint32_t beg = starts[tile], end = starts[tile + 1]; // (uniform address: scalar loads)
      for (uint32_t b0 = beg; b0 < end && go; b0 += 64u) {
        FFX_STAT(ANY ? 41 : 33);
        const uint32_t idx = b0 + lane;
        const uint32_t idc = idx < end ? idx : end - 1u; // lanes beyond the list re-read its last entry and are masked out
        const float4 *e4 = reinterpret_cast<const float4 *>(ents + ((size_t)idc << 6));
        const float4 A = e4[0], B = e4[1], C = e4[2], D = e4[3];
        wmask m = m_ult(idx, end) & m_le(A.x, rx1) & m_ge(A.z, rx0) & m_le(A.y, ry1) & m_ge(A.w, ry0);
        m &= m_ge(fmaf(B.x, rcx, fmaf(fabsf(B.x), rhx, fmaf(B.y, rcy, fmaf(fabsf(B.y), rhy, B.z)))), 0.f);
        m &= m_ge(fmaf(B.w, rcx, fmaf(fabsf(B.w), rhx, fmaf(C.x, rcy, fmaf(fabsf(C.x), rhy, C.y)))), 0.f);
        m &= m_ge(fmaf(C.z, rcx, fmaf(fabsf(C.z), rhx, fmaf(C.w, rcy, fmaf(fabsf(C.w), rhy, D.x)))), 0.f);
        const int slot_v = __float_as_int(D.y);
        while (m != 0ull) {
          const uint32_t j = (uint32_t)wff1(m);
          asm("s_bitset0_b64 %0, %1" : "+s"(m) : "s"(j));
          FFX_STAT(ANY ? 42 : 34);
          const uint32_t slot = (uint32_t)__builtin_amdgcn_readlane(slot_v, (int)j);
          const uint32_t roff = slot * 48u;
          const v8i r8 = *reinterpret_cast<const v8i *>(rbase + roff);
          const v4i r4 = *reinterpret_cast<const v4i *>(rbase + roff + 32);
          asm volatile("" ::"s"(r8), "s"(r4));
          const int prim = r4.z;
          // apex test, staged with wave-uniform early-outs (identical arithmetic to traverse_wide_oct / traverse_packet_oct)
          const v3 Av = V3(__int_as_float(r8.s0), __int_as_float(r8.s1), __int_as_float(r8.s2));
          const v3 Bv = V3(__int_as_float(r8.s3), __int_as_float(r8.s4), __int_as_float(r8.s5));
          const v3 Cv = V3(__int_as_float(r8.s6), __int_as_float(r8.s7), __int_as_float(r4.x));
          const float T = __int_as_float(r4.y);
          const float det = vdot(d, Av);
          const wmask neg = m_lt(det, 0.f);
          const float detA = fabsf(det);
          const wmask alive = ANY ? (active & ~occluded) : active;
          if (ANY) {
            // t = +-T / |det| must lie below tmax.  Most candidates of a shadow packet are the triangles of the surface the rays END on
            // (t = 1 up to rounding, beyond tmax = 1 - 10 eps): T and det alone settle them — |T| >= tmax (1 + 2^-18) |det| implies
            // fl(|T| / |det|) >= tmax — before either barycentric is formed (8 VALU instead of the 36 of a test that passes both)
            const float Ts = msel(neg, -T, T);
            if ((alive & m_lt(Ts, detA * tmax_up)) == 0ull) continue;
            FFX_STAT(46);
          }
          const float U = vdot(d, Bv);
          const float Us = msel(neg, -U, U);
          const wmask p1 = alive & m_gt(detA, 0.f) & m_ge(Us, 0.f) & m_le(Us, detA);
          if (p1 == 0ull) continue;
          FFX_STAT(ANY ? 43 : 35);
          const float Vv = vdot(d, Cv);
          const float Vs = msel(neg, -Vv, Vv);
          const wmask p2 = p1 & m_ge(Vs, 0.f) & m_le(Us + Vs, detA);
          if (p2 == 0ull) continue;
          const float t = msel(neg, -T, T) / detA;
          const wmask hitm = p2 & m_gt(t, tmin);
          if (ANY) {
            const wmask occ = hitm & m_lt(t, tmax);
            occluded |= occ;
            msel_into(h.t, occ, -INFINITY);
            if (occ != 0ull && (active & ~occluded) == 0ull) { go = false; m = 0ull; } // every ray of the packet is decided
          } else {
            const wmask better = hitm & (m_lt(t, h.t) | (m_eq(t, h.t) & m_ult((uint32_t)prim, (uint32_t)h.prim)));
            msel_into(h.t, better, t);
            msel_into(h.prim, better, prim);
            msel_into(h.slot, better, (int)slot);
          }
        }
      }
    }
  }
}

// the bins of apex `a` are usable for this launch and this pose?  (grid enabled by the host; lists complete: written by k_bin_scan)
__device__ __forceinline__ bool bins_ready(const BinsK &bk, const int a) {
  if (!bk.g[a].on) return false;
  return reinterpret_cast<const BinHdr *>(bk.base[a])->ok != 0u; // (uniform address: a scalar load)
}

// primary rays of pixel (px, py): its own tile, its own square of the image plane.  The square is taken a half pad short at the far
// side: a sample position of exactly px + 1 (rounding of px + jitter) still lies within the entries' padding, and the pixels of a
// tile's last column do not drag the next tile's list in.
__device__ __forceinline__ bool bins_primary(const BinsK &bk, const int px, const int py, const TriApex *__restrict__ recs, const v3 d, const float tmin, const float tmax,
                                             const wmask active, Hit &h) {
  if (!bins_ready(bk, 0)) return false;
  if (active == 0ull) { h.t = -INFINITY; h.prim = -1; h.shape = -1; h.slot = -1; return true; } // nothing to trace
  const float its_x = bk.cam_inv_ts_x, its_y = bk.cam_inv_ts_y; // 1 / tile side: a power of two
  const float x0 = (float)px * its_x, y0 = (float)py * its_y;
  const int tx = (int)x0, ty = (int)y0;
  const float hx = 0.5f * its_x - 0.25f * FFX_BIN_PAD, hy = 0.5f * its_y - 0.25f * FFX_BIN_PAD;
  const float cx = x0 + hx, cy = y0 + hy;
  // (uniform values computed on the vector ALU: say so, so that they feed the step as scalar operands)
  const int stx = __builtin_amdgcn_readfirstlane(tx), sty = __builtin_amdgcn_readfirstlane(ty);
  const float scx = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(cx))), scy = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(cy)));
  const float shx = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(hx))), shy = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(hy)));
  if (stx >= bk.g[0].nx || sty >= bk.g[0].ny) return false;
  wmask occ;
  traverse_bins<false>(bk.base[0], bk.g[0].nx, stx, sty, stx, sty, scx, scy, shx, shy, recs, d, tmin, tmax, active, h, occ);
  return true;
}

// primary rays of a compact block of bw x bh pixels (K7's packets at fewer than 64 samples per pixel: 4x4 pixels at 1 spp ... one pixel):
// the block is aligned to its own size and at most 4 pixels wide, so it lies inside ONE tile of >= 4 pixels; its rectangle is the block
__device__ __forceinline__ bool bins_block(const BinsK &bk, const int bx0, const int by0, const int bw, const int bh, const TriApex *__restrict__ recs, const v3 d,
                                           const float tmin, const float tmax, const wmask active, Hit &h) {
  if (!bins_ready(bk, 0)) return false;
  if (active == 0ull) { h.t = -INFINITY; h.prim = -1; h.shape = -1; h.slot = -1; return true; }
  const float its_x = bk.cam_inv_ts_x, its_y = bk.cam_inv_ts_y; // 1 / tile side: a power of two
  if ((float)bw * its_x > 1.0f || (float)bh * its_y > 1.0f) return false; // (tiles smaller than the block: FFX_BIN_TILE experiments)
  const float x0 = (float)bx0 * its_x, y0 = (float)by0 * its_y;
  const int tx = (int)x0, ty = (int)y0;
  const float hx = 0.5f * (float)bw * its_x - 0.25f * FFX_BIN_PAD, hy = 0.5f * (float)bh * its_y - 0.25f * FFX_BIN_PAD;
  const float cx = x0 + hx, cy = y0 + hy;
  const int stx = __builtin_amdgcn_readfirstlane(tx), sty = __builtin_amdgcn_readfirstlane(ty);
  const float scx = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(cx))), scy = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(cy)));
  const float shx = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(hx))), shy = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(hy)));
  if (stx >= bk.g[0].nx || sty >= bk.g[0].ny) return false;
  wmask occ;
  traverse_bins<false>(bk.base[0], bk.g[0].nx, stx, sty, stx, sty, scx, scy, shx, shy, recs, d, tmin, tmax, active, h, occ);
  return true;
}

// four wave-wide reductions of non-negative floats at once — two minima, two maxima — in interleaved DPP chains (wave_reduce3_nn's
// scheme: the three instructions between a DPP write and the next read of the same register are the wait states the hazard needs)
__device__ __forceinline__ void wave_reduce_minmax4(uint32_t &mn0, uint32_t &mn1, uint32_t &mx0, uint32_t &mx1) {
#define FFX_R4_STEP(CTRL, MASK)                                                   \
  "v_min_u32_dpp %0, %0, %0 " CTRL " row_mask:" MASK " bank_mask:0xf\n\t"       \
  "v_min_u32_dpp %1, %1, %1 " CTRL " row_mask:" MASK " bank_mask:0xf\n\t"       \
  "v_max_u32_dpp %2, %2, %2 " CTRL " row_mask:" MASK " bank_mask:0xf\n\t"       \
  "v_max_u32_dpp %3, %3, %3 " CTRL " row_mask:" MASK " bank_mask:0xf\n\t"
  asm("s_nop 1\n\t" FFX_R4_STEP("quad_perm:[1,0,3,2]", "0xf") FFX_R4_STEP("quad_perm:[2,3,0,1]", "0xf") FFX_R4_STEP("row_half_mirror", "0xf")
          FFX_R4_STEP("row_mirror", "0xf") FFX_R4_STEP("row_bcast:15", "0xa") FFX_R4_STEP("row_bcast:31", "0xc")
      : "+v"(mn0), "+v"(mn1), "+v"(mx0), "+v"(mx1));
#undef FFX_R4_STEP
  mn0 = (uint32_t)__builtin_amdgcn_readlane((int)mn0, 63);
  mn1 = (uint32_t)__builtin_amdgcn_readlane((int)mn1, 63);
  mx0 = (uint32_t)__builtin_amdgcn_readlane((int)mx0, 63);
  mx1 = (uint32_t)__builtin_amdgcn_readlane((int)mx1, 63);
}

// shadow rays from emitter apex `a` (1 projector, 2 spot) to the lifted surface points: direction sdir = Po - E per lane.  Their image
// points on the emitter's grid are (M sdir).xy / (M sdir).z; the packet's rectangle is their bounding box (the samples of one pixel:
// a pixel's footprint as the emitter sees it).  A packet that leaves the grid or spreads over more than four tiles (the two sides of
// a depth discontinuity, far apart as seen from the emitter) takes the tree walk.
__device__ __forceinline__ bool bins_shadow(const BinsK &bk, const int a, const TriApex *__restrict__ recs, const v3 sdir, const wmask active, wmask &occluded) {
  if (!bins_ready(bk, a)) return false;
  const float *M = bk.g[a].M;
  const float Z = fmaf(M[6], sdir.x, fmaf(M[7], sdir.y, M[8] * sdir.z));
  const float iz = __builtin_amdgcn_rcpf(Z);
  const float fx = fmaf(M[0], sdir.x, fmaf(M[1], sdir.y, M[2] * sdir.z)) * iz, fy = fmaf(M[3], sdir.x, fmaf(M[4], sdir.y, M[5] * sdir.z)) * iz;
  const float gx = (float)bk.g[a].nx, gy = (float)bk.g[a].ny;
  // inside the grid (NaN fails): everything else is the tree's business.  0 <= f < g as ONE unsigned comparison of the floats' bits: non-negative floats
  // order like their bits, negative ones (and NaN) have larger bits than any grid size (-0.0 counts as outside: the tree gives the same answer)
  const wmask inside = m_gt(Z, 0.f) & m_ult(__float_as_uint(fx), __float_as_uint(gx)) & m_ult(__float_as_uint(fy), __float_as_uint(gy));
  if ((active & ~inside) != 0ull) return false;
  // (round 6) the grid's ENVELOPE (ffx_common.h FFX_ENV_SUB, k_bin_env): the cell of each sample's image point holds a plane N . (X - E) = 1 in
  // front of every triangle the tile lists over that cell, pulled forward past the ignored tail of a shadow ray.  N . sdir <= 1: the segment
  // ends in front of it, no listed triangle can be hit within the counted part — and only listed triangles can be hit at all.  A packet all of
  // whose samples pass skips the stage (most do: the emitter stands next to the camera); NaN cells (no proof offered) fail the comparison.
  if (((bk.env_on >> (a - 1)) & 1) && reinterpret_cast<const BinHdr *>(bk.base[a])->env != 0u) { // (the header: THIS pose's pre-pass built it)
    const int nxf = bk.g[a].nx * FFX_ENV_SUB;
    const uint32_t last = (uint32_t)(nxf * bk.g[a].ny * FFX_ENV_SUB) - 1u;
    const uint32_t ci = (uint32_t)((int)(fy * (float)FFX_ENV_SUB) * nxf + (int)(fx * (float)FFX_ENV_SUB));
    const float4 Nc = reinterpret_cast<const float4 *>(bk.base[a] + bk.env_off)[ci < last ? ci : last]; // (lanes that are not active may hold anything)
    const float v = fmaf(Nc.x, sdir.x, fmaf(Nc.y, sdir.y, Nc.z * sdir.z));
    if ((active & ~m_le(v, 1.0f)) == 0ull) { occluded = 0ull; FFX_STAT(a == 1 ? 37 : 47); return true; }
  }
#ifdef FFX_EXP_NO_SPOT_WALK // timing experiment: what the any-hit stage of the packets the envelope does NOT settle costs (they count as unoccluded)
  if (a == 2) { occluded = 0ull; return true; }
#endif
  uint32_t mnx = msel(active, __float_as_uint(fx), 0x7f800000u), mny = msel(active, __float_as_uint(fy), 0x7f800000u);
  uint32_t mxx = msel(active, __float_as_uint(fx), 0u), mxy = msel(active, __float_as_uint(fy), 0u);
  wave_reduce_minmax4(mnx, mny, mxx, mxy);
  const float x0 = __uint_as_float(mnx), y0 = __uint_as_float(mny), x1 = __uint_as_float(mxx), y1 = __uint_as_float(mxy);
  const int tx0 = (int)x0, ty0 = (int)y0, tx1 = (int)x1, ty1 = (int)y1; // (non-negative: truncation is floor)
  const int stx0 = __builtin_amdgcn_readfirstlane(tx0), sty0 = __builtin_amdgcn_readfirstlane(ty0);
  const int stx1 = __builtin_amdgcn_readfirstlane(tx1), sty1 = __builtin_amdgcn_readfirstlane(ty1);
  if ((stx1 - stx0 + 1) * (sty1 - sty0 + 1) > 4) return false;
  const float cx = 0.5f * (x0 + x1), cy = 0.5f * (y0 + y1), hx = 0.5f * (x1 - x0) + 1e-5f * gx, hy = 0.5f * (y1 - y0) + 1e-5f * gy; // (+ the rounding of the centre form)
  const float scx = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(cx))), scy = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(cy)));
  const float shx = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(hx))), shy = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(hy)));
  Hit hs;
  traverse_bins<true>(bk.base[a], bk.g[a].nx, stx0, sty0, stx1, sty1, scx, scy, shx, shy, recs, sdir, 0.f, 1.0f - SHADOW_EPS, active, hs, occluded);
  return true;
}

// per-sample shading state between the three packet walks
struct ShadePre {
  bool ok, need_p, need_s;
  v3 P, ng, Po;
  float pfac, u, v, sfac;
  float pfac_b, sfac_b; // material rows only
};

// packet version of shade_sample for R samples per lane: every lane of the wave reaches every walk
// The scene constants (ShadeK, ~100 dwords) are the first kernel argument of the render kernels.  Read
// through `c` the compiler loads them all up front and, out of SGPRs, parks them in VGPR lanes
// (v_writelane / v_readlane: ~480 spill instructions on the VALU, the unit that bounds these kernels).
// kernarg_shade() hands out the same constants through a pointer the compiler cannot see through, so
// each phase re-reads what it needs with scalar loads from the kernarg segment (no VALU work at all).

// whether any of the four texels of the bilinear footprint at (u, v) is non-zero (same texel arithmetic as the footprint phase)
__device__ __forceinline__ bool tex_footprint_lit(const float *__restrict__ tex, int tw, int th, int tc, float u, float v) {
  const float fx = fmaf(u, (float)tw, -0.5f), fy = fmaf(v, (float)th, -0.5f);
  const int ix0 = (int)floorf(fx), iy0 = (int)floorf(fy);
  const int x0 = clampi(ix0, 0, tw - 1), x1 = clampi(ix0 + 1, 0, tw - 1), y0 = clampi(iy0, 0, th - 1), y1 = clampi(iy0 + 1, 0, th - 1);
  float m = 0.f;
  for (int ch = 0; ch < tc; ++ch)
    m = fmaxf(m, fmaxf(fmaxf(fabsf(tex[((size_t)y0 * tw + x0) * tc + ch]), fabsf(tex[((size_t)y0 * tw + x1) * tc + ch])),
                       fmaxf(fabsf(tex[((size_t)y1 * tw + x0) * tc + ch]), fabsf(tex[((size_t)y1 * tw + x1) * tc + ch]))));
  return m > 0.f; // (NaN texels: fmaxf drops them — as before, a NaN next to zeros reads as dark)
}

// MATM: 0 = [S,3] Lambert albedos, 1 = material rows, 2 = material rows some of which take their base colour from a texture
// px, py (wave-uniform): the pixel the packet's primary rays belong to — the tile bins are tried first (R == 1), the tree walks serve
// what they cannot
template <int R, bool WIDE, int MATM = 0>
__device__ __forceinline__ void shade_sample_pk(const BvhNode *__restrict__ nodes, const TriRec *__restrict__ recs, const TriApex *__restrict__ arecs,
                                                uint32_t astride, const WideScene &ws, uint2 *__restrict__ stack, const bool (&active)[R], const v3 (&o)[R],
                                                const v3 (&d)[R], const float (&nt)[R], const float (&ft)[R], SampleTerms (&st)[R],
                                                const float4 *__restrict__ nrec, const float4 *__restrict__ gn, const int px, const int py,
                                                const float *__restrict__ tex_probe = nullptr, const int blk_w = 1, const int blk_h = 1) {
  constexpr bool MAT = MATM != 0, TEX = MATM == 2;
  Hit h[R];
  bool fnd[R];
  FFX_TSTART(tp);
  bool binned = false;
  // (blk_w x blk_h > 1: the packet's primary rays belong to a compact block of pixels whose first is (px, py) — k_render_fwd_blk, renders at
  // fewer than 64 samples per pixel; the default arguments are constants of every other caller)
  if constexpr (R == 1) {
    if (blk_w * blk_h == 1) binned = bins_primary(kernarg_shade().bins, px, py, arecs, d[0], nt[0], ft[0], wballot(active[0]), h[0]);
    else binned = bins_block(kernarg_shade().bins, px, py, blk_w, blk_h, arecs, d[0], nt[0], ft[0], wballot(active[0]), h[0]);
  }
  if (!binned) {
    if constexpr (R == 1) FFX_STAT(36);
    traverse_packet_any<false, R, WIDE>(nodes, arecs, ws, stack, o, d, nt, ft, active, h, fnd); // apex 0: the camera
  }
#ifdef FFX_BINCHECK // self-check build: every binned walk is repeated on the tree and compared lane by lane (tools/bincheck.py)
  if constexpr (R == 1) {
    if (binned) {
      Hit h2[R];
      traverse_packet_any<false, R, WIDE>(nodes, arecs, ws, stack, o, d, nt, ft, active, h2, fnd);
      const bool bad = active[0] && (h2[0].prim != h[0].prim || (h2[0].prim >= 0 && h2[0].t != h[0].t));
      const wmask bm = wballot(bad);
      if (bm != 0ull && (threadIdx.x & 63) == (unsigned)wff1(bm)) {
        atomicAdd(&g_ffx_chk[0], 1ull);
        if (atomicAdd(&g_ffx_chk[1], 1ull) == 0ull) { g_ffx_chk[2] = (unsigned long long)px | ((unsigned long long)py << 16) | ((unsigned long long)(threadIdx.x & 63) << 32); g_ffx_chk[3] = (unsigned long long)(uint32_t)h[0].prim | ((unsigned long long)(uint32_t)h2[0].prim << 32); g_ffx_chk[4] = (unsigned long long)(uint32_t)h[0].slot | ((unsigned long long)(uint32_t)h2[0].slot << 32); g_ffx_chk[5] = (unsigned long long)__float_as_uint(h[0].t) | ((unsigned long long)__float_as_uint(h2[0].t) << 32); }
      }
      h[0] = h2[0];
    }
  }
#endif
  FFX_TSTOP(tp, 22);
  const ShadeK &c = kernarg_shade(); // phase: light terms at the hit point
  ShadePre pre[R];
  int cbits[R]; // the hit triangle's flag word (per-slot normal, word 3): the emitters' "clear" bits (ffx_common.h FFX_GN_CLEAR_BIT)
#pragma unroll
  for (int r = 0; r < R; ++r) cbits[r] = 0;
  bool any_p = false, any_s = false;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    ShadePre &q = pre[r];
    // shading normal of this sample (ffx_smooth): equal to the geometric one unless the hit record is flagged.  It takes the
    // geometric normal's place in q.ng once the geometric side tests below are done (one normal stays live across the walks).
    v3 ns = V3(0.f, 0.f, 1.f);
    bool smooth = false;
    st[r].hit = h[r].prim >= 0;
    st[r].has_proj = 0;
    st[r].proj_fac = 0.f; st[r].proj_fac_b = 0.f;
    st[r].spot[0] = st[r].spot[1] = st[r].spot[2] = 0.f;
    st[r].spot_b[0] = st[r].spot_b[1] = st[r].spot_b[2] = 0.f;
    st[r].shape = -1; // read from the hit's triangle record below (the octant loops do not carry it)
    q.P = V3(0.f, 0.f, 0.f);
    q.ng = V3(0.f, 0.f, 1.f);
    q.Po = V3(0.f, 0.f, 0.f);
    q.ok = st[r].hit != 0;
#ifdef FFX_EXPERIMENT_PRIMARY_ONLY // timing experiment: ray generation + primary walk only
    q.ok = false;
#endif
    if (q.ok) {
      const float4 *r4 = reinterpret_cast<const float4 *>(recs + h[r].slot);
      // the unit geometric normal comes from the update (ffx_bvh_info.off_gn: IEEE, the oracle's bits) — re-deriving it from the
      // record cost 22 VALU per sample (cross, dot, sqrt, reciprocal, scale) — with the shape id and the smooth flag in its fourth
      // word: the record itself is only touched by samples that interpolate normals or look up a base-colour texture
      const float4 gq = gn[h[r].slot];
      const int gbits = __float_as_int(gq.w); // 0: degenerate triangle; else (shape + 1) | smooth << 30
      st[r].shape = (gbits & FFX_GN_SHAPE_MASK) - 1;
      cbits[r] = gbits;
      q.P = V3(fmaf(h[r].t, d[r].x, o[r].x), fmaf(h[r].t, d[r].y, o[r].y), fmaf(h[r].t, d[r].z, o[r].z));
      v3 ng = V3(gq.x, gq.y, gq.z);
      q.ok = (gbits & FFX_GN_SHAPE_MASK) != 0;
      if (!q.ok) st[r].shape = 0;
      if (q.ok) {
        if (vdot(ng, d[r]) > 0.f) ng = V3(-ng.x, -ng.y, -ng.z);
        q.ng = ng;
        float pmax = fmaxf(fabsf(q.P.x), fmaxf(fabsf(q.P.y), fabsf(q.P.z)));
        float off = (1.0f + pmax) * RAY_EPS;
        q.Po = V3(fmaf(off, ng.x, q.P.x), fmaf(off, ng.y, q.P.y), fmaf(off, ng.z, q.P.z));
        ns = ng;
        smooth = (gbits & FFX_GN_SMOOTH_BIT) != 0;
        if (wballot(smooth) != 0ull) { // (wave-uniform: scenes without flagged records never enter)
          const float4 ra = r4[0], rb = r4[1], rc = r4[2];
          const v3 ni = interpolated_normal<true>(nrec, smooth ? h[r].slot : 0, ra, rb, rc, o[r], d[r], ng);
          if (smooth) ns = ni;
        }
      }
    }
    const bool any_smooth = wballot(smooth) != 0ull;
    // ---- projector terms
    q.need_p = false;
    q.pfac = 0.f; q.u = 0.f; q.v = 0.f;
    q.pfac_b = 0.f; q.sfac_b = 0.f;
    // material rows (MAT): pi f cos = base_color * bA + bB per emitter (material_eval); Lambert: bA = cos_s, bB = 0
    // (MAT: the BSDF is evaluated AFTER the shadow walks, for the samples the emitters reach — there the walk's registers
    // are free; before the walks only the geometric factors are formed, and P, ng stay live across them.  Parking P and ng
    // in LDS across the walks reaches 8 waves per SIMD instead of 7 but measured the same: the BSDF's own arithmetic, not
    // occupancy, is what the material rows cost — K8 0.57 ms against 0.48 ms for a diffuse scene)
    if (c.proj_on && q.ok) {
      v3 pl = xf_point(c.p_w2l, q.P);
      if (pl.z > 0.f) {
        const float *m = c.p_c2s;
        float qx = fmaf(m[0], pl.x, fmaf(m[1], pl.y, fmaf(m[2], pl.z, m[3])));
        float qy = fmaf(m[4], pl.x, fmaf(m[5], pl.y, fmaf(m[6], pl.z, m[7])));
        float qw = fmaf(m[12], pl.x, fmaf(m[13], pl.y, fmaf(m[14], pl.z, m[15])));
        const float iqw = rcp_nr(qw);
        q.u = qx * iqw;
        q.v = qy * iqw;
        if (q.u >= 0.f && q.u <= 1.f && q.v >= 0.f && q.v <= 1.f) {
          v3 ppos = V3(c.p_pos[0], c.p_pos[1], c.p_pos[2]);
          v3 wi = vsub(ppos, q.P);
          float d2 = vdot(wi, wi);
          const float idist = rsqrt_nr(d2);
          wi = V3(wi.x * idist, wi.y * idist, wi.z * idist);
          float cos_s = vdot(ns, wi);
          float cos_p = -vdot(V3(c.p_axis[0], c.p_axis[1], c.p_axis[2]), wi);
          bool lit = cos_s > 0.f && cos_p > 0.f;
          if (any_smooth) lit = lit && (!smooth || vdot(q.ng, wi) > 0.f); // the emitter on the viewer's GEOMETRIC side too
          if constexpr (MAT) {
            // (material rows: the texture probe — see below — comes first, it saves the BSDF of a dark footprint too)
            if (tex_probe && lit) lit = tex_footprint_lit(tex_probe, c.tw, c.th, c.tc, q.u, q.v);
          }
          if (lit) {
            q.need_p = true;
            if constexpr (MAT) {
              q.pfac = div_nr(c.p_scale, pl.z * pl.z * cos_p); // x (bA, bB) after the walks
            } else {
              q.pfac = div_nr(c.p_scale, pl.z * pl.z * cos_p) * cos_s;
            }
          }
        }
      }
    }
    // ---- spot terms
    q.need_s = false;
    q.sfac = 0.f;
    if (c.spot_on && q.ok) {
      v3 spos = V3(c.s_pos[0], c.s_pos[1], c.s_pos[2]);
      v3 wi = vsub(spos, q.P);
      float d2 = vdot(wi, wi);
      const float idist = rsqrt_nr(d2);
      wi = V3(wi.x * idist, wi.y * idist, wi.z * idist);
      float cos_s = vdot(ns, wi);
      bool front = cos_s > 0.f;
      if (any_smooth) front = front && (!smooth || vdot(q.ng, wi) > 0.f);
      if (front) {
        float cos_t;
        if (c.s_rigid) { // (the spot's world-to-local is a rotation — the usual case: |ll| = |wi| = 1, and ll.z is one row of it)
          cos_t = -fmaf(c.s_w2l[8], wi.x, fmaf(c.s_w2l[9], wi.y, c.s_w2l[10] * wi.z));
        } else {
          v3 ll = xf_dir(c.s_w2l, V3(-wi.x, -wi.y, -wi.z));
          float ln = sqrt_nr(vdot(ll, ll));
          cos_t = div_nr(ll.z, ln);
        }
        float fall = 0.f;
        if (cos_t >= c.cos_beam) fall = 1.f;
        else if (cos_t > c.cos_cut) fall = (c.cutoff - acosf(cos_t)) * c.inv_trans;
        if (fall > 0.f) {
          q.need_s = true;
          if constexpr (MAT) {
            q.sfac = div_nr(fall, d2) * 0.3183098861837907f; // x (bA, bB) after the walks
          } else {
            q.sfac = div_nr(fall * cos_s, d2) * 0.3183098861837907f;
          }
        }
      }
    }
    q.ng = ns; // from here on (BSDF after the walks) only the shading normal is needed
    // A plain forward render (no adjoint cache) does not need the projector's shadow ray where the projector shines
    // nothing: if the four texels of the sample's bilinear footprint are all exactly zero — most of a dot pattern is —
    // its contribution is zero whatever the walk finds.  (The cache-writing forward keeps every walk: the ADJOINT of a
    // dark texel is not zero.)  Same texel arithmetic as the footprint phase below.
    if constexpr (!MAT) {
      if (tex_probe && q.need_p) q.need_p = tex_footprint_lit(tex_probe, c.tw, c.th, c.tc, q.u, q.v);
    }
    any_p |= q.need_p;
    any_s |= q.need_s;
  }
  // ---- shadow walks (wave-uniform decisions).  A shadow ray is traced FROM the emitter — the apex its
  // triangle records were prepared for — to the lifted surface point: o = emitter, d = Po - emitter,
  // occluded iff some triangle is hit at 0 < t < 1 - eps.  No normalisation, and the packet's rays share
  // their origin exactly.
  bool occ_p[R], occ_s[R];
#pragma unroll
  for (int r = 0; r < R; ++r) occ_p[r] = occ_s[r] = false;
  FFX_TSTOP(tp, 18);
  // (round 5) a packet all of whose samples that need an emitter lie on triangles NOTHING can shadow from it (k_bin_clear's proof, the bit in
  // the per-slot normal's flag word; valid while that emitter's lists are: bins_ready) skips that emitter's any-hit stage altogether
  bool walk_p = c.shadows && wballot(any_p) != 0ull, walk_s = c.shadows && wballot(any_s) != 0ull;
  if constexpr (R == 1) {
    const BinsK &bkc = kernarg_shade().bins;
    if (bkc.clear_on) {
      if ((bkc.clear_on & 1) && walk_p && wballot(pre[0].need_p && !((uint32_t)cbits[0] & FFX_GN_CLEAR_BIT(1))) == 0ull && bins_ready(bkc, 1)) { walk_p = false; FFX_STAT(38); }
      if ((bkc.clear_on & 2) && walk_s && wballot(pre[0].need_s && !((uint32_t)cbits[0] & FFX_GN_CLEAR_BIT(2))) == 0ull && bins_ready(bkc, 2)) { walk_s = false; FFX_STAT(39); }
    }
  }
  if (walk_p) {
    const v3 ppos = V3(c.p_pos[0], c.p_pos[1], c.p_pos[2]);
    v3 so[R], sdir[R];
    float s0[R], s1[R];
    bool act[R];
    Hit hs[R];
#pragma unroll
    for (int r = 0; r < R; ++r) { so[r] = ppos; sdir[r] = vsub(pre[r].Po, ppos); s0[r] = 0.f; s1[r] = 1.0f - SHADOW_EPS; act[r] = pre[r].need_p; }
    const TriApex *arecs_p = reinterpret_cast<const TriApex *>(reinterpret_cast<const char *>(arecs) + astride);
    bool binned_p = false;
    if constexpr (R == 1) {
      wmask occm;
      binned_p = bins_shadow(kernarg_shade().bins, 1, arecs_p, sdir[0], wballot(act[0]), occm);
      if (binned_p) occ_p[0] = __builtin_amdgcn_inverse_ballot_w64(occm);
    }
    if (!binned_p) {
      if constexpr (R == 1) FFX_STAT(44);
      traverse_packet_any<true, R, WIDE>(nodes, arecs_p, ws, stack, so, sdir, s0, s1, act, hs, occ_p);
    }
#ifdef FFX_BINCHECK
    if constexpr (R == 1) {
      if (binned_p) {
        bool occ2[R];
        traverse_packet_any<true, R, WIDE>(nodes, arecs_p, ws, stack, so, sdir, s0, s1, act, hs, occ2);
        const wmask bm = wballot(act[0] && occ2[0] != occ_p[0]);
        if (bm != 0ull && (threadIdx.x & 63) == (unsigned)wff1(bm)) {
          atomicAdd(&g_ffx_chk[8], 1ull);
          if (atomicAdd(&g_ffx_chk[9], 1ull) == 0ull) { g_ffx_chk[10] = (unsigned long long)px | ((unsigned long long)py << 16) | ((unsigned long long)(threadIdx.x & 63) << 32) | ((unsigned long long)occ2[0] << 40); g_ffx_chk[11] = (unsigned long long)(uint32_t)hs[0].slot; }
        }
        occ_p[0] = occ2[0];
      }
    }
#endif
  }
  FFX_TSTOP(tp, 19);
#ifdef FFX_EXP_NO_SPOT_SHADOW // timing experiment: what the spot's any-hit stage costs (every spot sample counts as unoccluded)
  if (false) {
#else
  if (walk_s) {
#endif
    const v3 spos = V3(c.s_pos[0], c.s_pos[1], c.s_pos[2]);
    v3 so[R], sdir[R];
    float s0[R], s1[R];
    bool act[R];
    Hit hs[R];
#pragma unroll
    for (int r = 0; r < R; ++r) { so[r] = spos; sdir[r] = vsub(pre[r].Po, spos); s0[r] = 0.f; s1[r] = 1.0f - SHADOW_EPS; act[r] = pre[r].need_s; }
    const TriApex *arecs_s = reinterpret_cast<const TriApex *>(reinterpret_cast<const char *>(arecs) + 2u * astride);
    bool binned_s = false;
    if constexpr (R == 1) {
      wmask occm;
      binned_s = bins_shadow(kernarg_shade().bins, 2, arecs_s, sdir[0], wballot(act[0]), occm);
      if (binned_s) occ_s[0] = __builtin_amdgcn_inverse_ballot_w64(occm);
    }
    if (!binned_s) {
      if constexpr (R == 1) FFX_STAT(45);
      traverse_packet_any<true, R, WIDE>(nodes, arecs_s, ws, stack, so, sdir, s0, s1, act, hs, occ_s);
    }
#ifdef FFX_BINCHECK
    if constexpr (R == 1) {
      if (binned_s) {
        bool occ2[R];
        traverse_packet_any<true, R, WIDE>(nodes, arecs_s, ws, stack, so, sdir, s0, s1, act, hs, occ2);
        const wmask bm = wballot(act[0] && occ2[0] != occ_s[0]);
        if (bm != 0ull && (threadIdx.x & 63) == (unsigned)wff1(bm)) {
          atomicAdd(&g_ffx_chk[16], 1ull);
          if (atomicAdd(&g_ffx_chk[17], 1ull) == 0ull) { g_ffx_chk[18] = (unsigned long long)px | ((unsigned long long)py << 16) | ((unsigned long long)(threadIdx.x & 63) << 32) | ((unsigned long long)occ2[0] << 40); g_ffx_chk[19] = (unsigned long long)(uint32_t)hs[0].slot; }
        }
        occ_s[0] = occ2[0];
      }
    }
#endif
  }
  FFX_TSTOP(tp, 20);
  const ShadeK &c2 = kernarg_shade(); // phase: texture footprint and light intensities
#pragma unroll
  for (int r = 0; r < R; ++r) {
    ShadePre &q = pre[r];
    if constexpr (MAT) {
      // BSDF of the samples an emitter reaches: pi f cos = base_color * bA + bB (material_eval); Lambert rows: bA = cos_s
      const bool lit_p = q.need_p && !occ_p[r], lit_s = q.need_s && !occ_s[r];
      if constexpr (TEX) { st[r].base[0] = st[r].base[1] = st[r].base[2] = 0.f; }
      if (lit_p || lit_s) {
        const float *mrow = mat_table(c2) + (size_t)FFX_MAT_STRIDE * st[r].shape;
        const bool mat_on = mrow[FFX_MAT_MODEL] != 0.f;
        // (wave-uniform: the rows travel with the call and the host derived their constants — FFX_MAT_PRE=0 in the environment keeps the on-the-fly form)
        const float *prow = c2.mat_pre_on ? c2.mat_pre + FFX_MAT_PRE * st[r].shape : nullptr;
        const v3 wv = V3(-d[r].x, -d[r].y, -d[r].z);
        if constexpr (TEX) { // base colour of this sample: the row's, or its texture at the hit (only lit samples need one)
          st[r].base[0] = mrow[0]; st[r].base[1] = mrow[1]; st[r].base[2] = mrow[2];
          const int tix = (int)mrow[FFX_MAT_BASE_TEX];
          if (tix > 0 && tix <= c2.n_base_tex) {
            const float4 *r4 = reinterpret_cast<const float4 *>(recs + h[r].slot);
            float bu, bv;
            hit_barycentrics<true>(r4[0], r4[1], r4[2], o[r], d[r], bu, bv);
            base_tex_sample(c2, tix - 1, h[r].slot, bu, bv, st[r].base);
          }
        }
        // one emitter after the other: directions -> cosines (MatGeo) -> lobes
        if (lit_p) {
          v3 wi = vsub(V3(c2.p_pos[0], c2.p_pos[1], c2.p_pos[2]), q.P);
          const float idist = __builtin_amdgcn_rsqf(vdot(wi, wi));
          wi = V3(wi.x * idist, wi.y * idist, wi.z * idist);
          float bA = vdot(q.ng, wi), bB = 0.f;
          if (mat_on) {
            MatGeo g;
            if (prow) { material_geometry_p(mrow, prow, q.ng, wv, wi, g); material_terms_p<TEX>(mrow, prow, g, bA, bB, st[r].base[0], st[r].base[1], st[r].base[2]); }
            else { material_geometry(mrow, q.ng, wv, wi, g); material_terms<TEX>(mrow, g, bA, bB, st[r].base[0], st[r].base[1], st[r].base[2]); }
          }
          q.pfac_b = q.pfac * bB;
          q.pfac = q.pfac * bA;
        }
        if (lit_s) {
          v3 wi = vsub(V3(c2.s_pos[0], c2.s_pos[1], c2.s_pos[2]), q.P);
          const float idist = __builtin_amdgcn_rsqf(vdot(wi, wi));
          wi = V3(wi.x * idist, wi.y * idist, wi.z * idist);
          float bA = vdot(q.ng, wi), bB = 0.f;
          if (mat_on) {
            MatGeo g;
            if (prow) { material_geometry_p(mrow, prow, q.ng, wv, wi, g); material_terms_p<TEX>(mrow, prow, g, bA, bB, st[r].base[0], st[r].base[1], st[r].base[2]); }
            else { material_geometry(mrow, q.ng, wv, wi, g); material_terms<TEX>(mrow, g, bA, bB, st[r].base[0], st[r].base[1], st[r].base[2]); }
          }
          q.sfac_b = q.sfac * bB;
          q.sfac = q.sfac * bA;
        }
      }
    }
    if (q.need_p && !occ_p[r]) {
      st[r].proj_fac = q.pfac;
      if constexpr (MAT) st[r].proj_fac_b = q.pfac_b;
      float fx = fmaf(q.u, (float)c2.tw, -0.5f), fy = fmaf(q.v, (float)c2.th, -0.5f);
      float x0 = floorf(fx), y0 = floorf(fy);
      float ax = fx - x0, ay = fy - y0;
      int ix0 = (int)x0, iy0 = (int)y0;
      st[r].ubx = ix0; st[r].uby = iy0;
      st[r].ix0 = clampi(ix0, 0, c2.tw - 1);
      st[r].ix1 = clampi(ix0 + 1, 0, c2.tw - 1);
      st[r].iy0 = clampi(iy0, 0, c2.th - 1);
      st[r].iy1 = clampi(iy0 + 1, 0, c2.th - 1);
      st[r].wx0 = 1.0f - ax; st[r].wx1 = ax;
      st[r].wy0 = 1.0f - ay; st[r].wy1 = ay;
      st[r].has_proj = 1;
    }
    if (q.need_s && !occ_s[r]) {
      st[r].spot[0] = c2.s_int[0] * q.sfac;
      st[r].spot[1] = c2.s_int[1] * q.sfac;
      st[r].spot[2] = c2.s_int[2] * q.sfac;
      if constexpr (MAT) {
        st[r].spot_b[0] = c2.s_int[0] * q.sfac_b;
        st[r].spot_b[1] = c2.s_int[1] * q.sfac_b;
        st[r].spot_b[2] = c2.s_int[2] * q.sfac_b;
      }
    }
  }
  FFX_TSTOP(tp, 21);
}

// Packet kernels.  A wavefront owns one 2x2-pixel tile (65,536 work items at 512x512): tile costs vary
// by an order of magnitude (rays along the tube vs. rays that leave it), so fine-grained, independent
// waves let the hardware dispatcher keep every SIMD busy to the end.  The 64 lanes are the 64 samples
// of a pixel and the tile is walked as four single pixels (R = 1; the template parameter remains from
// a two-pixels-per-lane experiment).  All 64 rays of a packet are within a one-pixel frustum, so the
// union of their paths is practically one ray's path.
#define PK_BLOCK 64 // one independent wave per workgroup: the finest grain for the dispatcher (2 / 4 waves measured +1 % / +8 % time)

__device__ __forceinline__ float wave_sum64(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// K7 on the packet traversal.  A wavefront owns a compact block of bw x bh pixels with spp_w samples
// each (bw * bh * spp_w = 64): 8x8 pixels at 1 spp ... one pixel at >= 64 spp (then it loops over the
// pixel's samples 64 at a time).  Sample index and jitter are those of k_trace_primary.
template <bool WIDE>
__global__ void __launch_bounds__(PK_BLOCK) __attribute__((amdgpu_waves_per_eu(FFX_PK1_WAVES, FFX_PK1_WAVES)))
    k_trace_primary_pk(CamK cam, const BvhNode *__restrict__ nodes, const TriRec *__restrict__ recs, const TriApex *__restrict__ arecs, WideScene ws, int spp,
                       int jitter, uint32_t seed_key, int bw_log2,
                       int bh_log2, int blocks_x, int n_blocks, float *__restrict__ t_out, int32_t *__restrict__ shape_out, int32_t *__restrict__ prim_out, BinsK bins) {
  __shared__ uint2 s_wstack[WIDE ? FFX_WSTACK : 1];
  // (wave-uniform by construction; readfirstlane tells the compiler, which otherwise carries everything derived from it in VGPRs)
  const int blk = __builtin_amdgcn_readfirstlane(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
  if (blk >= n_blocks) return; // whole wave
  const int lane = threadIdx.x & 63;
  const int ppw_log2 = bw_log2 + bh_log2; // pixels per wave (log2)
  const int spp_w = 64 >> ppw_log2;       // samples of one pixel handled per pass
  const int pl = lane >> (6 - ppw_log2);  // pixel of the block (0 when the wave is a single pixel)
  const int sl = lane & (spp_w - 1);      // sample slot within the pass
  const int x = ((blk % blocks_x) << bw_log2) + (pl & ((1 << bw_log2) - 1));
  const int y = ((blk / blocks_x) << bh_log2) + (pl >> bw_log2);
  const bool in_img = x < cam.W && y < cam.H;
  const uint32_t pix = (uint32_t)y * (uint32_t)cam.W + (uint32_t)x;
  const int passes = (spp + spp_w - 1) / spp_w;
  for (int pass = 0; pass < passes; ++pass) {
    const int sidx = pass * spp_w + sl;
    const bool active[1] = {in_img && sidx < spp};
    const uint32_t idx = pix * (uint32_t)spp + (uint32_t)sidx;
    float jx = 0.f, jy = 0.f;
    if (jitter) sample_jitter(seed_key, idx, jx, jy);
    v3 o[1], d[1];
    float nt[1], ft[1];
    cam_ray(cam, ((float)x + jx) * cam.inv_w, ((float)y + jy) * cam.inv_h, o[0], d[0], nt[0], ft[0]);
    Hit h[1];
    bool fnd[1];
    // round 4: the block's candidates from the camera's tile bins (as K8's primary rays) — the tree walk when there are none
    // (no bins area, a grid that is off or overflowed)
    bool binned = false;
    if (bins.g[0].on)
      binned = bins_block(bins, (blk % blocks_x) << bw_log2, (blk / blocks_x) << bh_log2, 1 << bw_log2, 1 << bh_log2, arecs, d[0], nt[0], ft[0], wballot(active[0]), h[0]);
    if (!binned) traverse_packet_any<false, 1, WIDE>(nodes, arecs, ws, s_wstack, o, d, nt, ft, active, h, fnd);
    if (active[0]) {
      const bool hit = h[0].prim >= 0;
      t_out[idx] = hit ? (h[0].t - nt[0]) : 0.f;
      if (shape_out) shape_out[idx] = hit ? recs[h[0].slot].shape : -1;
      if (prim_out) prim_out[idx] = h[0].prim;
    }
  }
}

// tile index -> pixel.  tiles_x carries the enumeration: bits 0..23 the number of 2x2-pixel tiles per
// image row, bits 24..27 tb: tiles are enumerated in square blocks of 2^tb x 2^tb tiles (row-major
// inside a block, blocks row-major over the image; tb = 0: plain row-major), so that consecutive work
// items — which the dispatcher hands to neighbouring wave slots — are compact 2-D patches of the image.
template <int R>
__device__ __forceinline__ void packet_pixels(int tile, int tiles_x_tb, int sub, int (&px)[R], int (&py)[R]) {
  const int tb = (tiles_x_tb >> 24) & 15, tiles_x = tiles_x_tb & 0xffffff;
  const int bt1 = (1 << tb) - 1, blocks_x = (tiles_x + bt1) >> tb;
  const int blk = tile >> (2 * tb), within = tile & ((1 << (2 * tb)) - 1);
  // row / column of the block: an UNSIGNED division through a float reciprocal with one correction step (exact for
  // the < 2^22 blocks of any film) — the compiler's signed integer division and modulo were ~50 scalar instructions
  // per wave, a tenth of a one-pixel wave's scalar work
  uint32_t by_ = (uint32_t)((float)blk * __builtin_amdgcn_rcpf((float)blocks_x));
  by_ = __builtin_amdgcn_readfirstlane((int)by_);
  int rem = blk - (int)by_ * blocks_x;
  if (rem < 0) { --by_; rem += blocks_x; }
  if (rem >= blocks_x) { ++by_; rem -= blocks_x; }
  const int tx = (rem << tb) + (within & bt1), ty = ((int)by_ << tb) + (within >> tb);
  const int bx = tx * 2, by = ty * 2;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    if (R == 2) { px[r] = bx + r; py[r] = by + sub; }
    else { px[r] = bx + (sub & 1); py[r] = by + (sub >> 1); }
  }
}

// ---- adjoint cache (store instead of re-trace): what K9 needs of a pixel, written by K8.
// The render is linear in the texture and the 64 samples of a pixel land on a handful of neighbouring texels
// (measured: 87 % of the lit pixels within 4x4 texels, 99.5 % within 5x5), nearly always on ONE shape, so the
// adjoint of a pixel is  gtex[T] += (gimg[p] . albedo[shape] (. colour)) / spp * W[T]  with the per-pixel
// FOOTPRINT  W[T] = sum over its samples of fac * bilinear weight — 25 floats instead of 64 records of 16 B.
//   [0, 64)                        CacheHdr
//   64 + 8 * pixel                 CachePix: window origin, shape, lit flag — a DENSE array, so that the adjoint reads
//                                  2 MB of it instead of touching every pixel's footprint
//   off_foot + 112 * pixel         CacheFoot: the 5x5 footprint (written and read for lit pixels only)
//   off_arena + 24 * i             CacheStray: single samples that do not fit (outside the window: depth discontinuities,
//                                  grazing surfaces; a second shape in the pixel; later 64-sample passes of a pixel that
//                                  drift), allocated with one atomic per affected wave
// 37.7 MB at 512x512x64 (one 16-byte record per sample was 268 MB), of which K8 writes and K9 reads ~10 MB.
struct CacheHdr { uint32_t n_stray, cap_stray, dropped, pad[13]; };
struct __attribute__((aligned(8))) CachePix { int16_t x0, y0; uint16_t shape, lit; };
static_assert(sizeof(CachePix) == 8, "cache pixel header must be 8 bytes");
struct __attribute__((aligned(16))) CacheFoot { float w[25]; uint32_t pad[3]; };
static_assert(sizeof(CacheFoot) == 112, "cache footprint must be 112 bytes");
__host__ __device__ inline size_t cache_off_foot(size_t n_pix) { return (64 + 8 * n_pix + 127) & ~(size_t)127; }
__host__ __device__ inline size_t cache_off_arena(size_t n_pix) { return (cache_off_foot(n_pix) + sizeof(CacheFoot) * n_pix + 127) & ~(size_t)127; }
struct CacheStray { uint32_t pix, xy_shape; float ax, ay, fac; float fac_b; /* material rows only (else not written) */ }; // xy_shape = x0 | y0 << 12 ... see stray_pack
static_assert(sizeof(CacheStray) == 24, "stray record must be 24 bytes");
__host__ __device__ inline size_t cache_stray_capacity(int w, int h, int spp) {
  const size_t n = (size_t)w * h * spp / 64;
  return n < 4096 ? 4096 : n;
}
// material rows: a second footprint per pixel (the part of the BSDF that does not scale with base_color), behind the arena
__host__ __device__ inline size_t cache_off_foot_b(size_t n_pix, size_t cap_stray) {
  return (cache_off_arena(n_pix) + sizeof(CacheStray) * cap_stray + 127) & ~(size_t)127;
}

// ---- adjoint cache of a FILTERED render (ffx_render_fwd_cache_filtered, include/ffx.h).  A sample spreads over the 25 pixels of its
// window with weights of its own (the jitter), so a pixel's samples do not fold into one texture footprint as under the box film — its
// adjoint is a 25 x 25 matrix (window pixel x texel).  Kept instead: one 16-byte record PER SAMPLE of the pixels that have a lit sample
// {(ubx + 1) | (uby + 1) << 12 | shape << 24, ax, ay, fac} (+ fac_b in a second array with material rows) — one coalesced 1 KB store per
// wave and lit pass.  Round 6: the records live in an ARENA of 64-sample blocks (rounds 4-5: a dense [pixel][sample] array, 20 B x pixels x
// spp of address space — 5.4 GB at 1024 x 1024 x 256 — for the ~4 % of the pixels that hold a lit sample): the first lit pass of a pixel
// takes one block for itself and for every later pass with ONE atomic of its wave (on one of eight counters — sub-arenas —, neighbouring
// pixels on different ones; an arena with a block for every pass of every pixel hands each pixel its own, without atomics), the pixel's
// header holds the first block and that pass;
// capacity = rfc_cap_blocks; a pixel that finds the arena full is counted in `dropped` (sticky, as the
// box film's arena: ffx_render_cache_status, the NaN poison of the adjoint, ffx_adam_args.guard) and keeps no records.
//   [0, 64)                CacheHdr {-, capacity, dropped pixels, -, blocks taken from each of the eight sub-arenas (may exceed their share)}
//   64 + 8 * pixel         CachePix {x0 | y0 << 16 = first block, shape = first lit pass, lit = bit mask of the passes that hold records}
//   rfc_off_wsum           float per pixel
//   rfc_off_recs           uint4 per sample slot of the arena's blocks
//   rfc_off_facb           float per sample slot (material rows)
__host__ __device__ inline size_t rfc_cap_blocks(size_t n_pix, size_t spp, bool all = false) {
  // every pass of every pixel up to 2^18 blocks (335 MB with material rows: 512 x 512 x 64 spp keeps a block per pixel and cannot overflow,
  // whatever the texture), a QUARTER of them beyond: 1024 x 1024 x 256 — 1.05 M blocks, 1.34 GB instead of 5.4 GB.  (A tenth, 0.54 GB, was
  // tried first: BASELINE configs[4]'s own 1 024-point pattern lights 17 % of the film — 178 k pixels x 4 passes — and 73 k of them found no block.)
  // `all` (ffx_scene_desc.shadows & FFX_SHADOWS_CACHE_DENSE: the caller's answer to an overflow): a block for every pass of every pixel
  const size_t dense = n_pix * ((spp + 63) / 64), part = dense / 4, keep = (size_t)1 << 18;
  const size_t n = part > keep ? part : keep;
  return (all || n >= dense) ? dense : n;
}
__host__ __device__ inline size_t rfc_off_wsum(size_t n_pix) { return (64 + 8 * n_pix + 127) & ~(size_t)127; }
__host__ __device__ inline size_t rfc_off_recs(size_t n_pix) { return (rfc_off_wsum(n_pix) + 4 * n_pix + 127) & ~(size_t)127; }
__host__ __device__ inline size_t rfc_off_facb(size_t n_pix, size_t spp, bool all) { return (rfc_off_recs(n_pix) + 16 * 64 * rfc_cap_blocks(n_pix, spp, all) + 127) & ~(size_t)127; }
__host__ __device__ inline size_t rfc_bytes(size_t n_pix, size_t spp, bool mat, bool all) { return rfc_off_facb(n_pix, spp, all) + (mat ? ((4 * 64 * rfc_cap_blocks(n_pix, spp, all) + 127) & ~(size_t)127) : 0); }
static inline bool rfc_all(const ffx_scene_desc *sd) { return (sd->shadows & FFX_SHADOWS_CACHE_DENSE) != 0; }


// ---- reconstruction filter that spreads a sample over its 5x5-pixel window (include/ffx.h, ffx_scene_desc.rfilter) -------------------------
// A wave holds the 64 samples of ONE pixel, one per lane; what the film needs from it are 25 x (r, g, b, weight) sums over those samples —
// 100 cross-lane reductions (7 instructions each as DPP chains: ~700 at the 4-cycle rate, +65 % on the kernel).  Instead the lanes change
// roles through LDS: every lane parks its five weights per axis, its radiance and its "I am a sample" flag in 14 rows of 64 floats; then
// lane l becomes window entry l % 32 (25 of them) for the samples of half l / 32 and runs over its 32 samples with 16-byte LDS reads of the
// six rows it needs (its x weight, its y weight, the four channels): per four samples 6 reads, 4 multiplies and 16 fused multiply-adds,
// all in the 2-cycle class; the two halves meet in one cross-lane add per channel.  The rows alias the wide walk's stack (never live at the
// same time); the row pitch of 68 floats puts the rows a read touches into different banks.  The oracle sums in the same order (the two
// halves of every 64 samples apart, then together).  Measured (tools/rftime.py, 512^2 x 64 spp): see DESIGN 5.1.
// Tried and dropped: (1) lane = (entry, channel) over all 64 samples in two rounds — twice the LDS reads (96 x 1 KB per pixel: the CU's LDS
// port, shared by its four SIMDs, was then busy 80 % of the kernel's time) and 256 instead of 160 instructions: the kernel 0.386 -> 0.487 ms;
// (2) the same sums on the matrix pipe — v_mfma_f32_16x16x4_f32 with rows = channel (4 of 16 used), columns = window entry, K = samples
// (lanes ARE the K index: no role change, bit-identical k-ordered sums) — 32 instructions x 32 cycles of a pipe whose issue time adds to
// the VALU's here: 0.545 ms.
#define FFX_RF_ROW 68
#define FFX_RF_FLOATS (14 * FFX_RF_ROW)
static_assert(FFX_RF_ROW % 4 == 0, "rf_fold reads the rows as float4: the row pitch must keep them 16-byte aligned");
__device__ __forceinline__ void rf_weights(const RfC &c, float j, float (&w)[5]) {
  if (c.rec) { // (wave-uniform)
    const float x0 = -1.5f - j; // centre of window pixel 0 minus the sample position; pixel a: x0 + a
    const float e0 = __builtin_amdgcn_exp2f(c.aL * (x0 * x0)), r = __builtin_amdgcn_exp2f(c.aL2 * x0);
    const float e1 = e0 * r * c.k[0], e2 = e1 * r * c.k[1], e3 = e2 * r * c.k[2], e4 = e3 * r * c.k[3];
    w[0] = fmaxf(e0 - c.bias, 0.f); w[1] = fmaxf(e1 - c.bias, 0.f); w[2] = fmaxf(e2 - c.bias, 0.f); w[3] = fmaxf(e3 - c.bias, 0.f); w[4] = fmaxf(e4 - c.bias, 0.f);
    return;
  }
#pragma unroll
  for (int a = 0; a < 5; ++a) {
    const float x = ((float)(a - 2) + 0.5f) - j;
    w[a] = fmaxf(__expf(c.alpha * (x * x)) - c.bias, 0.f);
  }
}
static void rf_constants(float stddev, RfC &c) { // [EXT Mitsuba src/rfilters/gaussian.cpp] radius 4 stddev
  c.alpha = -1.0f / (2.0f * stddev * stddev);
  c.bias = expf(c.alpha * (4.0f * stddev) * (4.0f * stddev));
  c.aL = c.alpha * 1.4426950408889634f;
  c.aL2 = 2.0f * c.aL;
  for (int n = 0; n < 4; ++n) c.k[n] = (float)exp((double)c.alpha * (2 * n + 1));
  const char *e = getenv("FFX_RF_RECURRENCE");
  c.rec = (c.alpha >= -8.0f && !(e && strcmp(e, "0") == 0)) ? 1 : 0;
}
// acc[ch] of lane l += sum over the samples 32 (l / 32) .. + 31 of this pass of  gx[a] gy[b] L[ch]  for window entry n = 5 b + a = l % 32 (< 25)
__device__ __forceinline__ void rf_fold(float *__restrict__ s_rf, int lane, const float (&gx)[5], const float (&gy)[5], float l0, float l1, float l2, float l3,
                                        float (&acc)[4]) {
  int lz = lane;
  asm volatile("" : "+v"(lz)); // (keeps the row addresses out of long-lived registers, as for s_foot)
#pragma unroll
  for (int a = 0; a < 5; ++a) {
    s_rf[a * FFX_RF_ROW + lz] = gx[a];
    s_rf[(5 + a) * FFX_RF_ROW + lz] = gy[a];
  }
  s_rf[10 * FFX_RF_ROW + lz] = l0;
  s_rf[11 * FFX_RF_ROW + lz] = l1;
  s_rf[12 * FFX_RF_ROW + lz] = l2;
  s_rf[13 * FFX_RF_ROW + lz] = l3;
  __builtin_amdgcn_wave_barrier();
  const int n = min(lz & 31, 24), half = lz >> 5;
  const int b = (n * 13) >> 6, a = n - 5 * b; // n / 5, n % 5 for n < 25
  const float4 *rx = reinterpret_cast<const float4 *>(s_rf + a * FFX_RF_ROW) + 8 * half;
  const float4 *ry = reinterpret_cast<const float4 *>(s_rf + (5 + b) * FFX_RF_ROW) + 8 * half;
  const float4 *r0 = reinterpret_cast<const float4 *>(s_rf + 10 * FFX_RF_ROW) + 8 * half;
  float t0 = acc[0], t1 = acc[1], t2 = acc[2], t3 = acc[3];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const float4 X = rx[k], Y = ry[k];
    const float4 L0 = r0[k], L1 = r0[k + FFX_RF_ROW / 4], L2 = r0[k + 2 * (FFX_RF_ROW / 4)], L3 = r0[k + 3 * (FFX_RF_ROW / 4)];
    const float w0 = X.x * Y.x, w1 = X.y * Y.y, w2 = X.z * Y.z, w3 = X.w * Y.w;
    t0 = __builtin_fmaf(w0, L0.x, t0); t1 = __builtin_fmaf(w0, L1.x, t1); t2 = __builtin_fmaf(w0, L2.x, t2); t3 = __builtin_fmaf(w0, L3.x, t3);
    t0 = __builtin_fmaf(w1, L0.y, t0); t1 = __builtin_fmaf(w1, L1.y, t1); t2 = __builtin_fmaf(w1, L2.y, t2); t3 = __builtin_fmaf(w1, L3.y, t3);
    t0 = __builtin_fmaf(w2, L0.z, t0); t1 = __builtin_fmaf(w2, L1.z, t1); t2 = __builtin_fmaf(w2, L2.z, t2); t3 = __builtin_fmaf(w2, L3.z, t3);
    t0 = __builtin_fmaf(w3, L0.w, t0); t1 = __builtin_fmaf(w3, L1.w, t1); t2 = __builtin_fmaf(w3, L2.w, t2); t3 = __builtin_fmaf(w3, L3.w, t3);
  }
  acc[0] = t0; acc[1] = t1; acc[2] = t2; acc[3] = t3;
  __builtin_amdgcn_wave_barrier();
}
// a pixel's 25 outgoing sums -> scratch [pixel][25][4]: lane n < 25 adds its two halves (samples 0..31 in lanes 0..31, 32..63 in lanes 32..63)
__device__ __forceinline__ void rf_store(float *__restrict__ part, uint32_t pix, int lane, const float (&acc)[4]) {
  const float h0 = __shfl_down(acc[0], 32), h1 = __shfl_down(acc[1], 32), h2 = __shfl_down(acc[2], 32), h3 = __shfl_down(acc[3], 32);
  if (lane < 25) reinterpret_cast<float4 *>(part)[(size_t)pix * 25 + lane] = make_float4(acc[0] + h0, acc[1] + h1, acc[2] + h2, acc[3] + h3);
}

// rf_fold + rf_store for a wave that holds SEVERAL pixels (k_render_fwd_blk<..., RF>: 2^ppw_log2 pixels x spp_w = 64 >> ppw_log2 sample slots, pixel q in
// lanes [q spp_w, (q + 1) spp_w)): the same 14 rows of 64 floats; then work item (pixel q, window entry n) runs over ITS pixel's slots in order — the
// fma chain rf_fold runs over a half's 32 samples, where the slots beyond the sample count add exact zeros and the second half is empty: the same bits
// as the pixel-per-wave kernel's sums.  50 items per round (two pixels x 25 entries), 2^ppw_log2 / 2 rounds: 160 multiply-adds per lane as in rf_fold,
// for 2^ppw_log2 pixels instead of one.
__device__ __forceinline__ void rf_fold_blk(float *__restrict__ s_rf, int lane, const float (&gx)[5], const float (&gy)[5], float l0, float l1, float l2, float l3,
                                            int ppw_log2, int bw_log2, int bx0, int by0, int W, int H, float *__restrict__ part) {
  int lz = lane;
  asm volatile("" : "+v"(lz));
#pragma unroll
  for (int a = 0; a < 5; ++a) {
    s_rf[a * FFX_RF_ROW + lz] = gx[a];
    s_rf[(5 + a) * FFX_RF_ROW + lz] = gy[a];
  }
  s_rf[10 * FFX_RF_ROW + lz] = l0;
  s_rf[11 * FFX_RF_ROW + lz] = l1;
  s_rf[12 * FFX_RF_ROW + lz] = l2;
  s_rf[13 * FFX_RF_ROW + lz] = l3;
  __builtin_amdgcn_wave_barrier();
  const int spp_w = 64 >> ppw_log2, groups = spp_w >> 2; // float4 groups of a pixel's slots (spp_w >= 8: the block never exceeds 8 pixels)
  const int item = lz < 50 ? lz : 49, ql = item >= 25 ? 1 : 0, n = item - 25 * ql;
  const int b = (n * 13) >> 6, a = n - 5 * b; // n / 5, n % 5 for n < 25
  const int n_pix = 1 << ppw_log2;
  for (int q0 = 0; q0 < n_pix; q0 += 2) {
    const int q = q0 + ql; // (a block of one pixel does not get here: the host sends >= 2 pixels per wave, or the pixel-per-wave kernel)
    const float4 *rx = reinterpret_cast<const float4 *>(s_rf + a * FFX_RF_ROW) + q * groups;
    const float4 *ry = reinterpret_cast<const float4 *>(s_rf + (5 + b) * FFX_RF_ROW) + q * groups;
    const float4 *r0 = reinterpret_cast<const float4 *>(s_rf + 10 * FFX_RF_ROW) + q * groups;
    float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
    for (int k = 0; k < groups; ++k) {
      const float4 X = rx[k], Y = ry[k];
      const float4 L0 = r0[k], L1 = r0[k + FFX_RF_ROW / 4], L2 = r0[k + 2 * (FFX_RF_ROW / 4)], L3 = r0[k + 3 * (FFX_RF_ROW / 4)];
      const float w0 = X.x * Y.x, w1 = X.y * Y.y, w2 = X.z * Y.z, w3 = X.w * Y.w;
      t0 = __builtin_fmaf(w0, L0.x, t0); t1 = __builtin_fmaf(w0, L1.x, t1); t2 = __builtin_fmaf(w0, L2.x, t2); t3 = __builtin_fmaf(w0, L3.x, t3);
      t0 = __builtin_fmaf(w1, L0.y, t0); t1 = __builtin_fmaf(w1, L1.y, t1); t2 = __builtin_fmaf(w1, L2.y, t2); t3 = __builtin_fmaf(w1, L3.y, t3);
      t0 = __builtin_fmaf(w2, L0.z, t0); t1 = __builtin_fmaf(w2, L1.z, t1); t2 = __builtin_fmaf(w2, L2.z, t2); t3 = __builtin_fmaf(w2, L3.z, t3);
      t0 = __builtin_fmaf(w3, L0.w, t0); t1 = __builtin_fmaf(w3, L1.w, t1); t2 = __builtin_fmaf(w3, L2.w, t2); t3 = __builtin_fmaf(w3, L3.w, t3);
    }
    const int x = bx0 + (q & ((1 << bw_log2) - 1)), y = by0 + (q >> bw_log2);
    // (rf_store adds the second half's sums — here an empty half: + 0.f keeps the bits and turns a -0.f into the +0.f the other kernel stores)
    if (lz < 50 && q < n_pix && x < W && y < H)
      reinterpret_cast<float4 *>(part)[((size_t)y * W + x) * 25 + n] = make_float4(t0 + 0.f, t1 + 0.f, t2 + 0.f, t3 + 0.f);
  }
  __builtin_amdgcn_wave_barrier();
}

// the weights alone (the adjoint's first launch): one wave per pixel, the jitter decides everything
__global__ void __launch_bounds__(64) k_rf_weights(RfC rfc, int n_pix, int spp, uint32_t seed_key, float *__restrict__ part) {
  __shared__ __attribute__((aligned(16))) float s_rf[FFX_RF_FLOATS]; // (rf_fold reads it in 16-byte units)
  const int pix = blockIdx.x, lane = threadIdx.x;
  if (pix >= n_pix) return;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int pass = 0; pass < (spp + 63) >> 6; ++pass) {
    const int s = pass * 64 + lane;
    float jx, jy, gx[5], gy[5];
    sample_jitter(seed_key, (uint32_t)pix * (uint32_t)spp + (uint32_t)s, jx, jy);
    rf_weights(rfc, jx, gx);
    rf_weights(rfc, jy, gy);
    rf_fold(s_rf, lane, gx, gy, 0.f, 0.f, 0.f, s < spp ? 1.f : 0.f, acc);
  }
  rf_store(part, (uint32_t)pix, lane, acc);
}

// second launch of either direction: a pixel's 25 incoming sums.  Forward: img = (r, g, b) / weight.  Adjoint (gimg != NULL):
// G = gimg / weight as float4 per pixel.  Window entry n = (a, b) of source pixel (x - (a - 2), y - (b - 2)) is what that pixel's
// samples sent HERE; summed in window order like the oracle.  A wave owns 64 pixels of one film row; for each window row b it stages the
// 80-byte (b, 0..4) segments of the 68 source records it needs through LDS — five lanes per segment, every byte of the scratch area read
// once — and lane x then picks entry a from record x + 4 - a.  (Read straight from memory a lane's 25 loads were 400 bytes apart from its
// neighbours': 0.061 ms for 105 MB, the vector L1 thrashing.)
#define FFX_RFG_WAVES 1 // (one-wave workgroups: the gather of one render runs beside the next render's kernel on the other stream — with four waves and 22 KB of LDS
                        //  per workgroup it waited for room on one compute unit: 1 742 -> 1 913 filtered renders/s; the waves never cooperated anyway)
__global__ void __launch_bounds__(64 * FFX_RFG_WAVES) k_rf_gather(const float4 *__restrict__ part, int W, int H, int fp16, void *__restrict__ img,
                                                                   const float *__restrict__ gimg, float4 *__restrict__ G, float *__restrict__ wsum_out = nullptr) {
  __shared__ float4 s_seg[FFX_RFG_WAVES][68 * 5];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int x0 = blockIdx.x * 64, y = blockIdx.y * FFX_RFG_WAVES + wv;
  if (y >= H) return; // (wave-uniform; the waves of a workgroup never synchronise)
  float4 *seg = s_seg[wv];
  float r = 0.f, g = 0.f, b = 0.f, w = 0.f;
  for (int wb = 0; wb < 5; ++wb) {
    const int qy = y - (wb - 2);
    if (qy < 0 || qy >= H) continue; // (wave-uniform)
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const int e = lane + 64 * k; // element (record e / 5, entry e % 5) of the 68 x 5 segment table
      if (e < 340) {
        const int rec = e / 5, wa = e - 5 * rec;
        const int qx = x0 - 2 + rec;
        seg[e] = (qx >= 0 && qx < W) ? part[((size_t)qy * W + qx) * 25 + (wb * 5 + wa)] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int wa = 0; wa < 5; ++wa) {
      const float4 v = seg[(lane + 4 - wa) * 5 + wa];
      r += v.x; g += v.y; b += v.z; w += v.w;
    }
  }
  const int x = x0 + lane;
  if (x >= W) return;
  const size_t pix = (size_t)y * W + x;
  if (gimg) {
    const bool ok = w > 0.f;
    G[pix] = make_float4(ok ? gimg[pix * 3] / w : 0.f, ok ? gimg[pix * 3 + 1] / w : 0.f, ok ? gimg[pix * 3 + 2] / w : 0.f, 0.f);
    return;
  }
  const float v0 = w > 0.f ? r / w : 0.f, v1 = w > 0.f ? g / w : 0.f, v2 = w > 0.f ? b / w : 0.f;
  if (wsum_out) wsum_out[pix] = w; // (ffx_render_fwd_cache_filtered: the weight this pixel received, for the cached adjoint's G = gimg / weight)
  if (fp16 & 1) {
    _Float16 *p = (_Float16 *)img + pix * 3;
    p[0] = (_Float16)v0; p[1] = (_Float16)v1; p[2] = (_Float16)v2;
  } else {
    float *p = (float *)img + pix * 3;
    p[0] = v0; p[1] = v1; p[2] = v2;
  }
}

// lane n < 25: G of window pixel n = (a, b) of pixel (px, py), i.e. pixel (px + a - 2, py + b - 2); zero outside the film and in lanes >= 25
__device__ __forceinline__ float4 rf_window_g(const float4 *__restrict__ G, int px, int py, int W, int H, int lane, bool live) {
  const int b = (lane * 13) >> 6, a = lane - 5 * b;
  const int tx = px + a - 2, ty = py + b - 2;
  if (!live || lane >= 25 || tx < 0 || tx >= W || ty < 0 || ty >= H) return make_float4(0.f, 0.f, 0.f, 0.f);
  return G[(size_t)ty * W + tx];
}

#ifndef FFX_PK_MAT_WAVES
#define FFX_PK_MAT_WAVES 7 // material rows: 75 VGPRs (72 at this setting without spills; 8 waves spill 4)
#endif
// RF (ffx_render_fwd_filtered): instead of the pixel's mean, the wave leaves the 25 x 4 sums its samples send to its 5x5 window in `cache`
// (here: the scratch area, [pixel][25][4] floats; no adjoint cache in this mode) — see rf_fold above.
// RFC (ffx_render_fwd_cache_filtered; RF && !ADJ): additionally the per-sample records of the filtered film's adjoint cache (rfc_off_* above);
// its base travels in `adj_gtex`, the record areas' offsets (128-byte units) in cache_foot_off / cache_foot_b_off.
template <int R, bool WIDE, int MATM, bool ADJ = false, bool RF = false, bool RFC = false>
__global__ void __launch_bounds__(PK_BLOCK) __attribute__((amdgpu_waves_per_eu(MATM ? FFX_PK_MAT_WAVES : (R == 1 ? FFX_PK1_WAVES : 3), MATM ? FFX_PK_MAT_WAVES : (R == 1 ? FFX_PK1_WAVES : 4))))
    k_render_fwd_pk(ShadeK c, const BvhNode *__restrict__ nodes, const TriRec *__restrict__ recs, const TriApex *__restrict__ arecs, uint32_t astride,
                    WideScene ws, const float *__restrict__ albedo, const float *__restrict__ tex, int spp, uint32_t seed_key, int tiles_x, int n_tiles, int remap,
                    int fp16, void *__restrict__ img, char *__restrict__ cache, int ppw, float inv_spp_arg, uint32_t cache_foot_off, uint32_t cache_arena_off,
                    uint32_t cache_foot_b_off, const float4 *__restrict__ nrec, const float4 *__restrict__ gn, uint32_t cap_stray,
                    const float *__restrict__ adj_gimg, float *__restrict__ adj_gtex, float *__restrict__ adj_dot) {
  constexpr int NSUB = 4 / R;
  constexpr bool MAT = MATM != 0, TEX = MATM == 2;
  constexpr int MS = MAT ? FFX_MAT_STRIDE : 3; // floats per material row
  // adj_gtex (ffx_render_fwd_adjoint): the adjoint of a loss whose gradient gimg does not depend on the image (a loss linear in it) needs
  // no second pass — the pixel's footprint is scattered into gtex where it is formed instead of being stored for K9, stray samples at
  // once: no cache, no arena that could overflow, no launch behind the render.  `fold`: the footprint bookkeeping runs for either mode.
  // (ADJ is a template parameter: as a run-time mode its three pointers and the epilogue cost the plain forward a dozen scalar spills)
  // RF && ADJ (ffx_render_fwd_adjoint_filtered): adj_gimg is then G = gimg / weight as float4 per pixel (k_rf_gather) and every lit sample
  // enters the pixel's footprint with ITS OWN gradient — sum over its window of w_n G[pixel + n], times albedo and colour — already applied;
  // the epilogue scatters the footprint as it is.  1-channel textures only (the host refuses the other).
  const bool fold = RF ? ADJ : (ADJ || cache != nullptr);
  constexpr int WSTACK_N = WIDE ? FFX_WSTACK : 1, RF_N = (FFX_RF_FLOATS * 4 + 7) / 8;
  __shared__ __attribute__((aligned(16))) uint2 s_wstack[RF ? (WSTACK_N > RF_N ? WSTACK_N : RF_N) : WSTACK_N]; // (RF: the filter's rows alias the walk's stack and are read in 16-byte units)
  __shared__ float s_foot[32]; // the pixel's 5x5 texture footprint (adjoint cache)
  __shared__ float s_foot_b[MAT ? 32 : 1]; // material rows: the footprint of the base_color-independent part
  static_assert(R == 1, "the adjoint cache is written one pixel at a time");
  static_assert(!RFC || (RF && !ADJ), "RFC: the filtered forward that writes the per-sample adjoint records");
  FFX_TINIT();
  FFX_TSTART(twave);
  FFX_TSTART(tpro);
  // The per-pixel radiance sums are live across all three walks of every pass but touched once per pass:
  // they are parked in LDS (which these kernels do not otherwise use) instead of holding 3R VGPRs that
  // the allocator would spill to scratch at 8 waves per SIMD.  Each lane only ever reads its own slots.
  __shared__ float s_acc[R][RF ? 4 : 3][PK_BLOCK];
  // each wave of the workgroup owns its own tile; the waves never synchronise
  // a wave walks `ppw` (1, 2 or 4) of its tile's four pixels; 4 / ppw waves share a tile
  const int wv = __builtin_amdgcn_readfirstlane(xcd_remap(blockIdx.x, gridDim.x, remap) * (blockDim.x >> 6) + (threadIdx.x >> 6)); // wave-uniform: say so
  const int wpt_log2 = (R == 1) ? (ppw == 1 ? 2 : (ppw == 2 ? 1 : 0)) : 0; // waves per tile = NSUB / ppw, a power of two: shifts, not divisions
  const int tile = (R == 1) ? (wv >> wpt_log2) : wv / (NSUB / ppw), sub0 = (R == 1) ? (wv & ((1 << wpt_log2) - 1)) * ppw : (wv % (NSUB / ppw)) * ppw;
  const int lane = threadIdx.x & 63;
  const int W = c.cam.W, H = c.cam.H; // (the only direct use of the by-value copy)
  const int passes = (spp + 63) >> 6;
  // (wave-uniform, used once per pixel: kept in an SGPR — as a VGPR it was live across the whole kernel and spilled)
  const float inv_spp_u = inv_spp_arg; // 1 / spp from the host (a kernel argument is scalar by construction; the division here was ten vector instructions per wave)
  if (!ADJ && !RF && cache && wv == 0 && lane == 0) reinterpret_cast<CacheHdr *>(cache)->cap_stray = cap_stray; // (read back by K9 and ffx_render_cache_status; no wave of this launch reads it)
  if constexpr (RFC) { // (the arena's counters were cleared in front of this launch — launch_apex / FFX_RENDER_CACHE_ZEROED; nobody here reads the capacity word)
    if (wv == 0 && lane == 0) reinterpret_cast<CacheHdr *>(adj_gtex)->cap_stray = cap_stray;
  }
  FFX_TSTOP(tpro, 25);
  for (int sub = sub0; sub < sub0 + ppw; ++sub) {
    int px[R], py[R];
    packet_pixels<R>(tile, tiles_x, sub, px, py);
    bool live[R], any_live = false;
    uint32_t pix[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      live[r] = tile < n_tiles && px[r] < W && py[r] < H;
      any_live |= live[r];
      pix[r] = (uint32_t)py[r] * (uint32_t)W + (uint32_t)px[r];
    }
    if (wballot(any_live) == 0ull) continue;
    // adjoint cache: window origin (wave-uniform) and shape of this pixel's footprint, -1 until a sample is lit
    int fox = -1, foy = -1, fshape = -1;
    float fin0 = 0.f, fin1 = 0.f, fin2 = 0.f; // the pixel's value as stored (ffx_render_fwd_adjoint: <gimg, img>)
    float rfacc[4] = {0.f, 0.f, 0.f, 0.f}; // RF: this lane's window entry (lane % 32, samples of half lane / 32), four channels, summed over the passes
    uint32_t rfc_mask = 0u; // RFC: the passes of this pixel that hold a lit sample (wave-uniform)
    uint32_t rfc_first = 0u; // ... the arena block of the first of them (the later passes follow it), and that pass; rfc_drop: the arena was full
    int rfc_first_pass = -1;
    bool rfc_drop = false;
    for (int pass = 0; pass < passes; ++pass) {
      const int s = pass * 64 + lane;
      bool active[R];
      v3 o[R], d[R];
      float nt[R], ft[R];
      FFX_TSTART(tk);
      const CamK &cam = kernarg_shade().cam; // phase: ray generation
#pragma unroll
      for (int r = 0; r < R; ++r) {
        active[r] = live[r] && s < spp;
        uint32_t idx = pix[r] * (uint32_t)spp + (uint32_t)s;
        float jx, jy;
        sample_jitter(seed_key, idx, jx, jy);
        cam_ray(cam, ((float)px[r] + jx) * cam.inv_w, ((float)py[r] + jy) * cam.inv_h, o[r], d[r], nt[r], ft[r]);
      }
      FFX_TSTOP(tk, 16);
      SampleTerms st[R];
      // (fp16 carries the call's flags: bit 0 fp16 film, bit 1 FFX_RENDER_SPARSE_ADJOINT — then the cache-writing forward may
      // skip dark footprints too: the caller only wants gradients of texels whose value is not zero)
      shade_sample_pk<R, WIDE, MATM>(nodes, recs, arecs, astride, ws, s_wstack, active, o, d, nt, ft, st, nrec, gn, px[0], py[0], ((fold || RFC) && !(fp16 & 2)) ? nullptr : tex);
      FFX_TSTOP(tk, 17);
      if constexpr (RFC) {
        // ---- the filtered film's adjoint cache: this pass's 64 records, if any of its samples is lit (one 1 KB store per wave)
        const bool lit = active[0] && st[0].hit && st[0].has_proj;
        if (wballot(lit) != 0ull && !rfc_drop) {
          char *rfc = reinterpret_cast<char *>(adj_gtex);
          if (rfc_first_pass < 0) { // the pixel's first lit pass: blocks for it and for every pass behind it
            const uint32_t want = (uint32_t)(passes - pass);
            if (cap_stray >= (uint32_t)W * (uint32_t)H * (uint32_t)passes) {
              // the arena holds a block for every pass of every pixel (up to 2^18 blocks: 512 x 512 x 64 spp): the pixel's own — nothing to take,
              // nothing that can run out
              rfc_first = pix[0] * (uint32_t)passes + (uint32_t)pass;
              rfc_first_pass = pass;
            } else {
              // one atomic of the wave — on ONE of eight counters, picked so that neighbouring pixels take different ones: the lit pixels of a
              // laser dot are rendered together, and same-address atomics from eight XCDs serialise at ~90 ns each (one counter: 178 k of them
              // on configs[4], twice the kernel's own time; at 512 x 512 x 64 the filtered gradient step lost 8 %)
              const uint32_t sub = ((uint32_t)px[0] + 3u * (uint32_t)py[0]) & 7u, sub_cap = cap_stray >> 3;
              uint32_t b0 = 0u;
              if (lane == 0) b0 = atomicAdd(&reinterpret_cast<CacheHdr *>(rfc)->pad[1 + sub], want);
              b0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)b0);
              if (b0 + want > sub_cap) {
                rfc_drop = true; // (sub-arena full: this pixel keeps no records — counted, the adjoint is poisoned, the caller re-traces)
                if (lane == 0) atomicAdd(&reinterpret_cast<CacheHdr *>(rfc)->dropped, 1u);
              } else {
                rfc_first = sub * sub_cap + b0;
                rfc_first_pass = pass;
              }
            }
          }
        }
        if (wballot(lit) != 0ull && !rfc_drop) {
          rfc_mask |= 1u << pass;
          if (s < spp) {
            char *rfc = reinterpret_cast<char *>(adj_gtex);
            const size_t si = (size_t)(rfc_first + (uint32_t)(pass - rfc_first_pass)) * 64u + (size_t)(s & 63);
            uint4 rec = make_uint4(0u, 0u, 0u, 0u);
            if (lit) rec = make_uint4((uint32_t)(st[0].ubx + 1) | ((uint32_t)(st[0].uby + 1) << 12) | ((uint32_t)st[0].shape << 24), __float_as_uint(st[0].wx1),
                                      __float_as_uint(st[0].wy1), __float_as_uint(st[0].proj_fac));
            reinterpret_cast<uint4 *>(rfc + ((size_t)cache_foot_off << 7))[si] = rec;
            if constexpr (MAT) reinterpret_cast<float *>(rfc + ((size_t)cache_foot_b_off << 7))[si] = lit ? st[0].proj_fac_b : 0.f;
          }
        }
      }
      if (fold) {
        // ---- adjoint cache: fold this pass's lit samples into the pixel's footprint
        const bool lit = active[0] && st[0].hit && st[0].has_proj;
        const wmask litm = wballot(lit);
        if (litm != 0ull) {
          if (fox < 0) { // first lit samples of the pixel: the window starts at their smallest tap
            // (the footprint is cleared HERE, not at the start of every pixel: 96 % of the pixels of a dot pattern's render never get here.
            // Index laundered: the compiler otherwise keeps &s_foot[lane] in a VGPR across the whole kernel — and, at the 64-VGPR budget,
            // spills it: 256 B of scratch traffic per wave for an address that costs two instructions)
            int lz = lane;
            asm volatile("" : "+v"(lz));
            if (lz < 32) s_foot[lz] = 0.f;
            if constexpr (MAT) { if (lz < 32) s_foot_b[lz] = 0.f; }
            __builtin_amdgcn_wave_barrier();
            fox = (int)wave_reduce_nn<false>(lit ? (uint32_t)st[0].ix0 : 0xffffffffu);
            foy = (int)wave_reduce_nn<false>(lit ? (uint32_t)st[0].iy0 : 0xffffffffu);
            fshape = __builtin_amdgcn_readlane(st[0].shape, wff1(litm));
          }
          const bool in_win = lit && st[0].ix0 >= fox && st[0].ix1 <= fox + 4 && st[0].iy0 >= foy && st[0].iy1 <= foy + 4 && st[0].shape == fshape;
          float rf_pf = 0.f; // RF && ADJ: proj_fac (and proj_fac_b) of this sample with its filtered gradient, albedo and colour applied
          if constexpr (RF && ADJ) {
            const ShadeK &cr = kernarg_shade();
            const float4 gw = rf_window_g(reinterpret_cast<const float4 *>(adj_gimg), px[0], py[0], W, H, lane, live[0]);
            float jx, jy, gx[5], gy[5];
            sample_jitter(seed_key, pix[0] * (uint32_t)spp + (uint32_t)s, jx, jy);
            rf_weights(cr.rf, jx, gx);
            rf_weights(cr.rf, jy, gy);
            float a0 = 0.f, a1 = 0.f, a2 = 0.f;
#pragma unroll
            for (int n = 0; n < 25; ++n) {
              const float w = gx[n % 5] * gy[n / 5];
              a0 = __builtin_fmaf(w, __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(gw.x), n)), a0);
              a1 = __builtin_fmaf(w, __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(gw.y), n)), a1);
              a2 = __builtin_fmaf(w, __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(gw.z), n)), a2);
            }
            if (lit) { // (the arithmetic of k_render_bwd_pk's scatter, 1-channel texture)
              const float *alb = mat_table(cr) + MS * st[0].shape;
              rf_pf = (a0 * alb[0] * cr.p_color[0] + a1 * alb[1] * cr.p_color[1] + a2 * alb[2] * cr.p_color[2]) * st[0].proj_fac;
              if constexpr (MAT) {
                if (st[0].proj_fac_b != 0.f) rf_pf += (a0 * cr.p_color[0] + a1 * cr.p_color[1] + a2 * cr.p_color[2]) * st[0].proj_fac_b;
              }
            }
          }
          if (in_win) {
            const float pf = (RF && ADJ) ? rf_pf : st[0].proj_fac;
            const int bx0 = st[0].ix0 - fox, bx1 = st[0].ix1 - fox, by0 = (st[0].iy0 - foy) * 5, by1 = (st[0].iy1 - foy) * 5;
            atomicAdd(&s_foot[by0 + bx0], pf * st[0].wy0 * st[0].wx0);
            atomicAdd(&s_foot[by0 + bx1], pf * st[0].wy0 * st[0].wx1);
            atomicAdd(&s_foot[by1 + bx0], pf * st[0].wy1 * st[0].wx0);
            atomicAdd(&s_foot[by1 + bx1], pf * st[0].wy1 * st[0].wx1);
            if constexpr (MAT && !(RF && ADJ)) {
              const float pb = st[0].proj_fac_b;
              if (pb != 0.f) {
                atomicAdd(&s_foot_b[by0 + bx0], pb * st[0].wy0 * st[0].wx0);
                atomicAdd(&s_foot_b[by0 + bx1], pb * st[0].wy0 * st[0].wx1);
                atomicAdd(&s_foot_b[by1 + bx0], pb * st[0].wy1 * st[0].wx0);
                atomicAdd(&s_foot_b[by1 + bx1], pb * st[0].wy1 * st[0].wx1);
              }
            }
          }
          const wmask straym = wballot(lit && !in_win);
          if (ADJ && straym != 0ull) { // fused adjoint: the four taps of a sample that does not fit the footprint, at once (k9_stray's arithmetic)
            if constexpr (RF) {
              if (lit && !in_win && rf_pf != 0.f) {
                const ShadeK &ca = kernarg_shade();
                atomicAdd(adj_gtex + (size_t)st[0].iy0 * ca.tw + st[0].ix0, rf_pf * st[0].wy0 * st[0].wx0);
                atomicAdd(adj_gtex + (size_t)st[0].iy0 * ca.tw + st[0].ix1, rf_pf * st[0].wy0 * st[0].wx1);
                atomicAdd(adj_gtex + (size_t)st[0].iy1 * ca.tw + st[0].ix0, rf_pf * st[0].wy1 * st[0].wx0);
                atomicAdd(adj_gtex + (size_t)st[0].iy1 * ca.tw + st[0].ix1, rf_pf * st[0].wy1 * st[0].wx1);
              }
            } else
            if (lit && !in_win) {
              const ShadeK &ca = kernarg_shade();
              const float g0 = adj_gimg[(size_t)pix[0] * 3], g1 = adj_gimg[(size_t)pix[0] * 3 + 1], g2 = adj_gimg[(size_t)pix[0] * 3 + 2];
              const float *alb = mat_table(ca) + MS * st[0].shape;
              const float pf = st[0].proj_fac, pb = MAT ? st[0].proj_fac_b : 0.f;
              const float wx0 = st[0].wx0, wx1 = st[0].wx1, wy0 = st[0].wy0, wy1 = st[0].wy1;
              if (ca.tc == 1) {
                float wsv = (g0 * alb[0] * ca.p_color[0] + g1 * alb[1] * ca.p_color[1] + g2 * alb[2] * ca.p_color[2]) * pf * inv_spp_u;
                if (pb != 0.f) wsv += (g0 * ca.p_color[0] + g1 * ca.p_color[1] + g2 * ca.p_color[2]) * pb * inv_spp_u;
                if (wsv != 0.f) {
                  atomicAdd(adj_gtex + (size_t)st[0].iy0 * ca.tw + st[0].ix0, wsv * wy0 * wx0);
                  atomicAdd(adj_gtex + (size_t)st[0].iy0 * ca.tw + st[0].ix1, wsv * wy0 * wx1);
                  atomicAdd(adj_gtex + (size_t)st[0].iy1 * ca.tw + st[0].ix0, wsv * wy1 * wx0);
                  atomicAdd(adj_gtex + (size_t)st[0].iy1 * ca.tw + st[0].ix1, wsv * wy1 * wx1);
                }
              } else {
                const size_t o00 = ((size_t)st[0].iy0 * ca.tw + st[0].ix0) * 3, o01 = ((size_t)st[0].iy0 * ca.tw + st[0].ix1) * 3;
                const size_t o10 = ((size_t)st[0].iy1 * ca.tw + st[0].ix0) * 3, o11 = ((size_t)st[0].iy1 * ca.tw + st[0].ix1) * 3;
                const float gg[3] = {g0, g1, g2};
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) {
                  float wsv = gg[ch] * alb[ch] * pf * inv_spp_u;
                  if (pb != 0.f) wsv += gg[ch] * pb * inv_spp_u;
                  if (wsv == 0.f) continue;
                  atomicAdd(adj_gtex + o00 + ch, wsv * wy0 * wx0);
                  atomicAdd(adj_gtex + o01 + ch, wsv * wy0 * wx1);
                  atomicAdd(adj_gtex + o10 + ch, wsv * wy1 * wx0);
                  atomicAdd(adj_gtex + o11 + ch, wsv * wy1 * wx1);
                }
              }
            }
          } else if (!ADJ && straym != 0ull) { // single samples that do not fit the footprint: one allocation per wave
            CacheHdr *hdr = reinterpret_cast<CacheHdr *>(cache);
            const uint32_t n = (uint32_t)wpop(straym);
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(&hdr->n_stray, n);
            base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
            const uint32_t cap = cap_stray; // (a kernel argument: the header may have been cleared as a whole by the caller, FFX_RENDER_CACHE_ZEROED)
            if (base + n <= cap) {
              if (lit && !in_win) {
                CacheStray *rec = reinterpret_cast<CacheStray *>(cache + ((size_t)cache_arena_off << 7)) + base + mbcnt64(straym);
                rec->pix = pix[0];
                rec->xy_shape = (uint32_t)(st[0].ubx + 1) | ((uint32_t)(st[0].uby + 1) << 12) | ((uint32_t)st[0].shape << 24);
                rec->ax = st[0].wx1;
                rec->ay = st[0].wy1;
                rec->fac = st[0].proj_fac;
                if constexpr (MAT) rec->fac_b = st[0].proj_fac_b;
              }
            } else if (lane == 0) {
              atomicAdd(&hdr->dropped, n); // arena exhausted (never seen: it holds 1/64 of all samples, strays are ~0.1 %)
            }
          }
        }
      }
      const ShadeK &ct = kernarg_shade(); // phase: texture gather and accumulation
#pragma unroll
      for (int r = 0; r < R; ++r) {
        float c0 = 0.f, c1 = 0.f, c2 = 0.f; // this pass's contribution of the lane's sample
        if (st[r].hit) {
        float r0 = st[r].spot[0], r1 = st[r].spot[1], r2 = st[r].spot[2];
        float b0 = st[r].spot_b[0], b1 = st[r].spot_b[1], b2 = st[r].spot_b[2]; // (MAT only: dead otherwise)
        if (st[r].has_proj) {
          const int tc = ct.tc;
          size_t o00 = ((size_t)st[r].iy0 * ct.tw + st[r].ix0) * tc, o01 = ((size_t)st[r].iy0 * ct.tw + st[r].ix1) * tc;
          size_t o10 = ((size_t)st[r].iy1 * ct.tw + st[r].ix0) * tc, o11 = ((size_t)st[r].iy1 * ct.tw + st[r].ix1) * tc;
          const float wx0 = st[r].wx0, wx1 = st[r].wx1, wy0 = st[r].wy0, wy1 = st[r].wy1, pf = st[r].proj_fac, pb = st[r].proj_fac_b;
          if (tc == 1) {
            float tv = wy0 * (wx0 * tex[o00] + wx1 * tex[o01]) + wy1 * (wx0 * tex[o10] + wx1 * tex[o11]);
            r0 += tv * ct.p_color[0] * pf;
            r1 += tv * ct.p_color[1] * pf;
            r2 += tv * ct.p_color[2] * pf;
            if constexpr (MAT) {
              b0 += tv * ct.p_color[0] * pb;
              b1 += tv * ct.p_color[1] * pb;
              b2 += tv * ct.p_color[2] * pb;
            }
          } else {
            float tv0 = wy0 * (wx0 * tex[o00] + wx1 * tex[o01]) + wy1 * (wx0 * tex[o10] + wx1 * tex[o11]);
            float tv1 = wy0 * (wx0 * tex[o00 + 1] + wx1 * tex[o01 + 1]) + wy1 * (wx0 * tex[o10 + 1] + wx1 * tex[o11 + 1]);
            float tv2 = wy0 * (wx0 * tex[o00 + 2] + wx1 * tex[o01 + 2]) + wy1 * (wx0 * tex[o10 + 2] + wx1 * tex[o11 + 2]);
            r0 += tv0 * 1.0f * pf;
            r1 += tv1 * 1.0f * pf;
            r2 += tv2 * 1.0f * pf;
            if constexpr (MAT) {
              b0 += tv0 * 1.0f * pb;
              b1 += tv1 * 1.0f * pb;
              b2 += tv2 * 1.0f * pb;
            }
          }
        }
        if constexpr (TEX) { // (a textured row: the sample's own base colour; zero where nothing lit it, and then r0..2 are zero too)
          c0 = st[r].base[0] * r0;
          c1 = st[r].base[1] * r1;
          c2 = st[r].base[2] * r2;
        } else {
        const float *alb = mat_table(ct) + MS * st[r].shape;
        c0 = alb[0] * r0;
        c1 = alb[1] * r1;
        c2 = alb[2] * r2;
        }
        if constexpr (MAT) { c0 += b0; c1 += b1; c2 += b2; }
        }
        if constexpr (RF) {
          // the lane's sample -> its 25 window entries (weights from the jitter, re-derived: two hashes instead of two live registers)
          // (re-derived: two hashes instead of two live registers — parking the jitter in LDS instead was tried in round 5: the film's fold keeps
          // the CU's LDS port busy as it is, filtered render 0.531 -> 0.559 ms)
          float jx, jy, gx[5], gy[5];
          sample_jitter(seed_key, pix[r] * (uint32_t)spp + (uint32_t)s, jx, jy);
          rf_weights(ct.rf, jx, gx);
          rf_weights(ct.rf, jy, gy);
          if (pass > 0) { rfacc[0] = s_acc[r][0][threadIdx.x]; rfacc[1] = s_acc[r][1][threadIdx.x]; rfacc[2] = s_acc[r][2][threadIdx.x]; rfacc[3] = s_acc[r][3][threadIdx.x]; }
          rf_fold(reinterpret_cast<float *>(s_wstack), lane, gx, gy, c0, c1, c2, active[r] ? 1.f : 0.f, rfacc); // (the weight channel: every sample drawn counts)
          if (pass + 1 < passes) { s_acc[r][0][threadIdx.x] = rfacc[0]; s_acc[r][1][threadIdx.x] = rfacc[1]; s_acc[r][2][threadIdx.x] = rfacc[2]; s_acc[r][3][threadIdx.x] = rfacc[3]; }
          else if (live[r]) rf_store(reinterpret_cast<float *>(cache), pix[r], lane, rfacc);
          continue;
        }
        // running sums of a pixel that needs several 64-sample passes are parked in LDS between the passes; the
        // usual single pass never touches it (it had cost 15 LDS operations per pixel)
        if (pass > 0) { c0 += s_acc[r][0][threadIdx.x]; c1 += s_acc[r][1][threadIdx.x]; c2 += s_acc[r][2][threadIdx.x]; }
        if (pass + 1 < passes) {
          s_acc[r][0][threadIdx.x] = c0; s_acc[r][1][threadIdx.x] = c1; s_acc[r][2][threadIdx.x] = c2;
        } else {
          // last pass: combine the 64 lanes in a fixed order (deterministic, no atomics) — the three channel sums in
          // interleaved DPP chains (FFX_R3_ALL, see wave_reduce3_nn: butterfly inside the rows, row_bcast to lane 63)
          asm(FFX_R3_ALL("v_add_f32_dpp") : "+v"(c0), "+v"(c1), "+v"(c2));
          const float a0 = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(c0), 63));
          const float a1 = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(c1), 63));
          const float a2 = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(c2), 63));
          if (lane == 0 && live[r]) {
            size_t o = (size_t)pix[r] * 3;
            if (fp16 & 1) {
              _Float16 *p = (_Float16 *)img;
              p[o] = (_Float16)vmul_s(a0, inv_spp_u);
              p[o + 1] = (_Float16)vmul_s(a1, inv_spp_u);
              p[o + 2] = (_Float16)vmul_s(a2, inv_spp_u);
            } else {
              float *p = (float *)img;
              p[o] = vmul_s(a0, inv_spp_u);
              p[o + 1] = vmul_s(a1, inv_spp_u);
              p[o + 2] = vmul_s(a2, inv_spp_u);
            }
          }
          if (ADJ && adj_dot) { // (wave-uniform values: the same in every lane)
            fin0 = (fp16 & 1) ? (float)(_Float16)vmul_s(a0, inv_spp_u) : vmul_s(a0, inv_spp_u);
            fin1 = (fp16 & 1) ? (float)(_Float16)vmul_s(a1, inv_spp_u) : vmul_s(a1, inv_spp_u);
            fin2 = (fp16 & 1) ? (float)(_Float16)vmul_s(a2, inv_spp_u) : vmul_s(a2, inv_spp_u);
          }
        }
      }
      FFX_TSTOP(tk, 23);
    }
    if constexpr (RFC) { // the pixel's header: which of its passes hold records
      if (live[0] && lane == 0) {
        CachePix hp;
        hp.x0 = (int16_t)(rfc_first & 0xffffu); hp.y0 = (int16_t)(rfc_first >> 16); hp.shape = (uint16_t)(rfc_first_pass < 0 ? 0 : rfc_first_pass); hp.lit = (uint16_t)rfc_mask;
        reinterpret_cast<CachePix *>(reinterpret_cast<char *>(adj_gtex) + 64)[pix[0]] = hp;
      }
    }
    if (ADJ && RF && live[0]) { // filtered fused adjoint: the footprint already carries every sample's own gradient
      __builtin_amdgcn_wave_barrier();
      const ShadeK &ca = kernarg_shade();
      int lw = lane;
      asm volatile("" : "+v"(lw));
      if (fox >= 0 && lw < 25) {
        const float w = s_foot[lw];
        const int ey = (lw * 13) >> 6, ex = lw - 5 * ey;
        if (w != 0.f) atomicAdd(adj_gtex + (size_t)(foy + ey) * ca.tw + (fox + ex), w);
      }
    }
    if (ADJ && !RF && live[0] && (fox >= 0 || adj_dot)) { // fused adjoint: the pixel's footprint x (gimg . albedo . colour) / spp goes straight into gtex (K9's arithmetic)
      __builtin_amdgcn_wave_barrier();
      const ShadeK &ca = kernarg_shade();
      const float g0 = adj_gimg[(size_t)pix[0] * 3], g1 = adj_gimg[(size_t)pix[0] * 3 + 1], g2 = adj_gimg[(size_t)pix[0] * 3 + 2];
      if (fox >= 0) {
        int lw = lane;
        asm volatile("" : "+v"(lw));
        if (lw < 25) {
          const float *alb = mat_table(ca) + MS * fshape;
          const float w = s_foot[lw];
          float wb = 0.f;
          if constexpr (MAT) wb = s_foot_b[lw];
          const int ey = (lw * 13) >> 6, ex = lw - 5 * ey; // lw / 5, lw % 5 for lw < 25
          const int tx = fox + ex, ty = foy + ey;
          if (w != 0.f || wb != 0.f) {
            if (ca.tc == 1) {
              const float wsv = (g0 * alb[0] * ca.p_color[0] + g1 * alb[1] * ca.p_color[1] + g2 * alb[2] * ca.p_color[2]) * inv_spp_u;
              float val = wsv * w;
              if (wb != 0.f) val += (g0 * ca.p_color[0] + g1 * ca.p_color[1] + g2 * ca.p_color[2]) * inv_spp_u * wb;
              if (val != 0.f) atomicAdd(adj_gtex + (size_t)ty * ca.tw + tx, val);
            } else {
              float *t = adj_gtex + ((size_t)ty * ca.tw + tx) * 3;
              if (g0 != 0.f) atomicAdd(t, g0 * alb[0] * inv_spp_u * w + g0 * inv_spp_u * wb);
              if (g1 != 0.f) atomicAdd(t + 1, g1 * alb[1] * inv_spp_u * w + g1 * inv_spp_u * wb);
              if (g2 != 0.f) atomicAdd(t + 2, g2 * alb[2] * inv_spp_u * w + g2 * inv_spp_u * wb);
            }
          }
        }
      }
      if (adj_dot && lane == 0) { // <gimg, img> of this pixel into one of FFX_ADJOINT_DOT_SLOTS partial sums (256 cache lines: a quarter of a
        const float dd = g0 * fin0 + g1 * fin1 + g2 * fin2; // million atomics over the launch meet ~1000 times per line, not on one address)
        if (dd != 0.f) atomicAdd(adj_dot + (pix[0] & (FFX_ADJOINT_DOT_SLOTS - 1)), dd);
      }
    }
    if (!ADJ && !RF && cache && live[0]) { // the pixel's slot: header always, the footprint only if something was lit
      __builtin_amdgcn_wave_barrier();
      if (lane == 0) {
        CachePix hp;
        hp.x0 = (int16_t)fox; hp.y0 = (int16_t)foy;
        hp.shape = (uint16_t)(fshape < 0 ? 0 : fshape);
        hp.lit = fox >= 0 ? 1 : 0;
        reinterpret_cast<CachePix *>(cache + 64)[pix[0]] = hp; // one 8-byte store
      }
      int lw = lane;
      asm volatile("" : "+v"(lw));
      if (fox >= 0 && lw < 25) reinterpret_cast<CacheFoot *>(cache + ((size_t)cache_foot_off << 7))[pix[0]].w[lw] = s_foot[lw];
      if constexpr (MAT) {
        if (fox >= 0 && lw < 25) reinterpret_cast<CacheFoot *>(cache + ((size_t)cache_foot_b_off << 7))[pix[0]].w[lw] = s_foot_b[lw];
      }
    }
  }
  FFX_TSTOP(twave, 24);
  FFX_TFLUSH();
}

// ---- K8 at fewer than 64 samples per pixel (round 5): the plain forward, box film.  k_render_fwd_pk gives every pixel a wave of its own
// whatever the sample count — at 1 spp one lane of 64 works and the launch costs what it costs at 64 (0.37 ms at 512^2: the dataset loop of
// main.py:138-160 draws its spp from 1..100).  Here a wave owns a compact block of bw x bh pixels with spp_w sample slots each, as
// k_trace_primary_pk does (same layout, same caps: thinner packets mean fewer exact tests per walk and enough waves to fill the GPU); the
// block's primary rays take the block's rectangle to the camera's tile bins (bins_block), shadow packets their samples' bounding box as always.
// Shading is shade_sample_pk's, per sample; a pixel's sum runs over its own spp_w lanes (xor butterfly inside the group).  No adjoint cache,
// no fused adjoint, no reconstruction filter: those keep a pixel per wave.
// RF (ffx_render_fwd_filtered below 33 spp): instead of the pixels' means the wave leaves every pixel's 25 x 4 outgoing sums in the scratch area
// (`img`: [pixel][25][4] floats), as k_render_fwd_pk<..., RF> does — rf_fold_blk below.
template <bool WIDE, int MATM, bool RF = false>
__global__ void __launch_bounds__(PK_BLOCK) __attribute__((amdgpu_waves_per_eu(MATM ? FFX_PK_MAT_WAVES : FFX_PK1_WAVES, MATM ? FFX_PK_MAT_WAVES : FFX_PK1_WAVES)))
    k_render_fwd_blk(ShadeK c, const BvhNode *__restrict__ nodes, const TriRec *__restrict__ recs, const TriApex *__restrict__ arecs, uint32_t astride, WideScene ws,
                     const float *__restrict__ albedo, const float *__restrict__ tex, int spp, uint32_t seed_key, int bw_log2, int bh_log2, int blocks_x, int n_blocks,
                     int fp16, void *__restrict__ img, float inv_spp_arg, const float4 *__restrict__ nrec, const float4 *__restrict__ gn) {
  constexpr bool MAT = MATM != 0, TEX = MATM == 2;
  constexpr int MS = MAT ? FFX_MAT_STRIDE : 3;
  constexpr int WSTACK_N = WIDE ? FFX_WSTACK : 1, RF_N = (FFX_RF_FLOATS * 4 + 7) / 8;
  __shared__ __attribute__((aligned(16))) uint2 s_wstack[RF ? (WSTACK_N > RF_N ? WSTACK_N : RF_N) : WSTACK_N]; // (RF: the filter's rows alias the walk's stack)
  const int blk = __builtin_amdgcn_readfirstlane(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
  if (blk >= n_blocks) return; // whole wave
  const int lane = threadIdx.x & 63;
  const int ppw_log2 = bw_log2 + bh_log2;
  const int spp_w = 64 >> ppw_log2;      // sample slots per pixel (>= spp: the host chooses the block so)
  const int pl = lane >> (6 - ppw_log2); // pixel of the block
  const int sl = lane & (spp_w - 1);     // sample slot
  // blocks are enumerated in patches of 8 x 8 blocks (row-major inside a patch, patches row-major over the film: n_blocks counts whole patches), so that
  // the waves the dispatcher starts together work on neighbouring tiles' lists
  const int patches_x = (blocks_x + 7) >> 3, patch = blk >> 6, within = blk & 63;
  const int pby = patch / patches_x, pbx = patch - pby * patches_x;
  const int bxi = (pbx << 3) + (within & 7), byi = (pby << 3) + (within >> 3);
  const int bx0 = bxi << bw_log2, by0 = byi << bh_log2;
  if (bx0 >= c.cam.W || by0 >= c.cam.H) return; // (a patch that overhangs the film)
  const int x = bx0 + (pl & ((1 << bw_log2) - 1)), y = by0 + (pl >> bw_log2);
  const int W = c.cam.W, H = c.cam.H; // (the only direct use of the by-value copy)
  const bool in_img = x < W && y < H;
  const uint32_t pix = (uint32_t)y * (uint32_t)W + (uint32_t)x;
  const bool active[1] = {in_img && sl < spp};
  v3 o[1], d[1];
  float nt[1], ft[1];
  {
    const CamK &cam = kernarg_shade().cam;
    float jx, jy;
    sample_jitter(seed_key, pix * (uint32_t)spp + (uint32_t)sl, jx, jy);
    cam_ray(cam, ((float)x + jx) * cam.inv_w, ((float)y + jy) * cam.inv_h, o[0], d[0], nt[0], ft[0]);
  }
  SampleTerms st[1];
  shade_sample_pk<1, WIDE, MATM>(nodes, recs, arecs, astride, ws, s_wstack, active, o, d, nt, ft, st, nrec, gn, bx0, by0, tex, 1 << bw_log2, 1 << bh_log2);
  const ShadeK &ct = kernarg_shade(); // texture gather and the sample's colour: k_render_fwd_pk's arithmetic
  float c0 = 0.f, c1 = 0.f, c2 = 0.f;
  if (st[0].hit) {
    float r0 = st[0].spot[0], r1 = st[0].spot[1], r2 = st[0].spot[2];
    float b0 = st[0].spot_b[0], b1 = st[0].spot_b[1], b2 = st[0].spot_b[2]; // (MAT only: dead otherwise)
    if (st[0].has_proj) {
      const int tc = ct.tc;
      const size_t o00 = ((size_t)st[0].iy0 * ct.tw + st[0].ix0) * tc, o01 = ((size_t)st[0].iy0 * ct.tw + st[0].ix1) * tc;
      const size_t o10 = ((size_t)st[0].iy1 * ct.tw + st[0].ix0) * tc, o11 = ((size_t)st[0].iy1 * ct.tw + st[0].ix1) * tc;
      const float wx0 = st[0].wx0, wx1 = st[0].wx1, wy0 = st[0].wy0, wy1 = st[0].wy1, pf = st[0].proj_fac, pb = st[0].proj_fac_b;
      if (tc == 1) {
        const float tv = wy0 * (wx0 * tex[o00] + wx1 * tex[o01]) + wy1 * (wx0 * tex[o10] + wx1 * tex[o11]);
        r0 += tv * ct.p_color[0] * pf;
        r1 += tv * ct.p_color[1] * pf;
        r2 += tv * ct.p_color[2] * pf;
        if constexpr (MAT) {
          b0 += tv * ct.p_color[0] * pb;
          b1 += tv * ct.p_color[1] * pb;
          b2 += tv * ct.p_color[2] * pb;
        }
      } else {
        const float tv0 = wy0 * (wx0 * tex[o00] + wx1 * tex[o01]) + wy1 * (wx0 * tex[o10] + wx1 * tex[o11]);
        const float tv1 = wy0 * (wx0 * tex[o00 + 1] + wx1 * tex[o01 + 1]) + wy1 * (wx0 * tex[o10 + 1] + wx1 * tex[o11 + 1]);
        const float tv2 = wy0 * (wx0 * tex[o00 + 2] + wx1 * tex[o01 + 2]) + wy1 * (wx0 * tex[o10 + 2] + wx1 * tex[o11 + 2]);
        r0 += tv0 * 1.0f * pf;
        r1 += tv1 * 1.0f * pf;
        r2 += tv2 * 1.0f * pf;
        if constexpr (MAT) {
          b0 += tv0 * 1.0f * pb;
          b1 += tv1 * 1.0f * pb;
          b2 += tv2 * 1.0f * pb;
        }
      }
    }
    if constexpr (TEX) {
      c0 = st[0].base[0] * r0;
      c1 = st[0].base[1] * r1;
      c2 = st[0].base[2] * r2;
    } else {
      const float *alb = mat_table(ct) + MS * st[0].shape;
      c0 = alb[0] * r0;
      c1 = alb[1] * r1;
      c2 = alb[2] * r2;
    }
    if constexpr (MAT) { c0 += b0; c1 += b1; c2 += b2; }
  }
  if constexpr (RF) {
    float jx, jy, gx[5], gy[5];
    sample_jitter(seed_key, pix * (uint32_t)spp + (uint32_t)sl, jx, jy);
    rf_weights(ct.rf, jx, gx);
    rf_weights(ct.rf, jy, gy);
    rf_fold_blk(reinterpret_cast<float *>(s_wstack), lane, gx, gy, c0, c1, c2, active[0] ? 1.f : 0.f, ppw_log2, bw_log2, bx0, by0, W, H, reinterpret_cast<float *>(img));
    return;
  }
  // the pixel's samples sit in spp_w neighbouring lanes: a butterfly inside the group, nearest partners first — the tree of k_render_fwd_pk's DPP
  // chain (pairs, quads, half rows, rows, rows of rows; its lanes beyond the sample count add zeros), so the two kernels store the same bits
  for (int m = 1; m < spp_w; m <<= 1) {
    c0 += __shfl_xor(c0, m, 64);
    c1 += __shfl_xor(c1, m, 64);
    c2 += __shfl_xor(c2, m, 64);
  }
  if (in_img && sl == 0) {
    const size_t o3 = (size_t)pix * 3;
    if (fp16 & 1) {
      _Float16 *p = (_Float16 *)img;
      p[o3] = (_Float16)(c0 * inv_spp_arg);
      p[o3 + 1] = (_Float16)(c1 * inv_spp_arg);
      p[o3 + 2] = (_Float16)(c2 * inv_spp_arg);
    } else {
      float *p = (float *)img;
      p[o3] = c0 * inv_spp_arg;
      p[o3 + 1] = c1 * inv_spp_arg;
      p[o3 + 2] = c2 * inv_spp_arg;
    }
  }
}


// Deterministic accumulation (ffx_render_bwd_det, include/ffx.h): float atomics make gtex depend on the order in which the samples' taps arrive
// (reassociation: ~1e-7 relative, different from run to run).  INTEGER additions commute: mode 1 finds the largest |tap value| of the launch
// (atomicMax on the float's bits — order-independent), the host derives a power-of-two scale from it, mode 2 adds llrint(value * scale) to a
// 64-bit fixed-point accumulator per texel (global_atomic_add_x2) and k_det_finish converts back — bitwise the same gtex whatever the
// dispatch order, the number of XCDs or the rank count, at a resolution of 2^-(62 - b) of the largest tap, b = bits of the launch's tap count: 2^-36 at 512 x 512 x 64 spp (float32 carries 2^-24).
struct DetK { int mode; float scale; unsigned long long *fix; unsigned int *vmax; };
// one of the two passes alone (ffx_render_bwd_det_part, ABI 9): part 1 = the largest tap into *vmax (atomicMax: accumulates over calls), part 2 = the
// fixed-point sums at the caller's scale 2^scale_log2 into fix[] (accumulate over calls — over the scene samples of a step, over ranks)
struct DetPart { int part; int scale_log2; unsigned long long *fix; unsigned int *vmax; };
// the scale of a deterministic accumulation: vmax < 2^e, |sum| <= n_taps x vmax < 2^(b + e) with b = bits of the tap count; 2^(62 - b - e) keeps
// every sum below 2^62 and one unit at 2^-(62 - b) of the largest tap.  INT_MIN: nothing lit (vmax = 0) or a non-finite tap
static int det_scale_log2(uint32_t vmax_bits, unsigned long long n_taps) {
  float vmax;
  memcpy(&vmax, &vmax_bits, 4);
  if (!(vmax > 0.f) || !(vmax < 3.0e38f)) return INT_MIN;
  int e;
  frexpf(vmax, &e);
  int b = 0;
  for (unsigned long long ns = n_taps > 0 ? n_taps : 1; ns > 0; ns >>= 1) ++b;
  const int sh = 62 - b - e;
  return sh > 126 ? 126 : (sh < -126 ? -126 : sh);
}
__device__ __forceinline__ void det_emit(const DetK &det, float *__restrict__ gtex, size_t t, float v) {
  if (det.mode == 0) { atomicAdd(gtex + t, v); return; }
  if (det.mode == 1) { if (v != 0.f) atomicMax(det.vmax, __float_as_uint(fabsf(v))); return; }
  if (v != 0.f) atomicAdd(det.fix + t, (unsigned long long)__double2ll_rn((double)v * (double)det.scale));
}
__global__ void __launch_bounds__(256) k_det_finish(const unsigned long long *__restrict__ fix, float inv_scale, long n, float *__restrict__ gtex) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  if (t < n) gtex[t] += (float)((double)(long long)fix[t] * (double)inv_scale); // (gtex is ACCUMULATED into, as by every adjoint)
}
// RF (ffx_render_bwd_filtered): `gimg` is then G = gimg / weight as float4 per pixel (k_rf_gather) and a sample's radiance receives
// sum over its 5x5 window of  w_n G[pixel + n]  — the transpose of the filter — in place of gimg[pixel] / spp.
template <int R, bool WIDE, int MATM, bool RF = false>
__global__ void __launch_bounds__(PK_BLOCK) __attribute__((amdgpu_waves_per_eu(MATM ? FFX_PK_MAT_WAVES : (R == 1 ? FFX_PK1_WAVES : 3), MATM ? FFX_PK_MAT_WAVES : (R == 1 ? FFX_PK1_WAVES : 4))))
    k_render_bwd_pk(ShadeK c, const BvhNode *__restrict__ nodes, const TriRec *__restrict__ recs, const TriApex *__restrict__ arecs, uint32_t astride,
                    WideScene ws, const float *__restrict__ albedo, int spp, uint32_t seed_key, int tiles_x, int n_tiles, int remap,
                    const float *__restrict__ gimg, float *__restrict__ gtex, const float4 *__restrict__ nrec, const float4 *__restrict__ gn, DetK det) {
  constexpr int NSUB = 4 / R;
  constexpr bool MAT = MATM != 0, TEX = MATM == 2;
  __shared__ uint2 s_wstack[WIDE ? FFX_WSTACK : 1];
  const int tile = __builtin_amdgcn_readfirstlane(xcd_remap(blockIdx.x, gridDim.x, remap) * (blockDim.x >> 6) + (threadIdx.x >> 6)); // wave-uniform: say so
  const int lane = threadIdx.x & 63;
  const int W = c.cam.W, H = c.cam.H; // (the only direct use of the by-value copy; see kernarg_shade)
  const float inv_spp = RF ? 1.0f : 1.0f / (float)spp; // (RF: the normalisation is the weight inside G)
  const int passes = (spp + 63) >> 6;
  static_assert(!RF || R == 1, "the filtered adjoint walks one pixel per wave");
  for (int sub = 0; sub < NSUB; ++sub) {
    int px[R], py[R];
    packet_pixels<R>(tile, tiles_x, sub, px, py);
    bool live[R], any_live = false;
    uint32_t pix[R];
    float g[R][3];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      live[r] = tile < n_tiles && px[r] < W && py[r] < H;
      pix[r] = live[r] ? (uint32_t)py[r] * (uint32_t)W + (uint32_t)px[r] : 0u;
      g[r][0] = g[r][1] = g[r][2] = 0.f;
      if constexpr (RF) {
        // lane n < 25 looks at window pixel n: the pixel is skipped when no pixel of its window carries a gradient
        const float4 gw = rf_window_g(reinterpret_cast<const float4 *>(gimg), px[r], py[r], W, H, lane, live[r]);
        live[r] = wballot(gw.x != 0.f || gw.y != 0.f || gw.z != 0.f) != 0ull;
      } else {
      if (live[r]) { g[r][0] = gimg[(size_t)pix[r] * 3]; g[r][1] = gimg[(size_t)pix[r] * 3 + 1]; g[r][2] = gimg[(size_t)pix[r] * 3 + 2]; }
      live[r] = live[r] && !(g[r][0] == 0.f && g[r][1] == 0.f && g[r][2] == 0.f);
      }
      any_live |= live[r];
    }
    if (wballot(any_live) == 0ull) continue; // wave-uniform
    for (int pass = 0; pass < passes; ++pass) {
      const int s = pass * 64 + lane;
      bool active[R];
      v3 o[R], d[R];
      float nt[R], ft[R];
      const CamK &cam = kernarg_shade().cam; // phase: ray generation
#pragma unroll
      for (int r = 0; r < R; ++r) {
        active[r] = live[r] && s < spp;
        uint32_t idx = pix[r] * (uint32_t)spp + (uint32_t)s;
        float jx, jy;
        sample_jitter(seed_key, idx, jx, jy);
        cam_ray(cam, ((float)px[r] + jx) * cam.inv_w, ((float)py[r] + jy) * cam.inv_h, o[r], d[r], nt[r], ft[r]);
      }
      SampleTerms st[R];
      shade_sample_pk<R, WIDE, MATM>(nodes, recs, arecs, astride, ws, s_wstack, active, o, d, nt, ft, st, nrec, gn, px[0], py[0]);
      const ShadeK &ct = kernarg_shade(); // phase: scatter into the texture gradient
      const int tc = ct.tc;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        if constexpr (RF) {
          // every lane's sample gathers its own gradient through the filter's weights: window pixel n's G sits in lane n (re-read: cached),
          // broadcast with readlane and weighted per lane — window order, fused multiply-adds, as the oracle
          const float4 gw = rf_window_g(reinterpret_cast<const float4 *>(gimg), px[r], py[r], W, H, lane, live[r]);
          float jx, jy, gx[5], gy[5];
          sample_jitter(seed_key, pix[r] * (uint32_t)spp + (uint32_t)s, jx, jy);
          rf_weights(ct.rf, jx, gx);
          rf_weights(ct.rf, jy, gy);
          float a0 = 0.f, a1 = 0.f, a2 = 0.f;
#pragma unroll
          for (int n = 0; n < 25; ++n) {
            const float w = gx[n % 5] * gy[n / 5];
            a0 = __builtin_fmaf(w, __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(gw.x), n)), a0);
            a1 = __builtin_fmaf(w, __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(gw.y), n)), a1);
            a2 = __builtin_fmaf(w, __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(gw.z), n)), a2);
          }
          g[r][0] = a0; g[r][1] = a1; g[r][2] = a2;
        }
        if (!st[r].hit || !st[r].has_proj) continue;
        const float *alb = TEX ? st[r].base : mat_table(ct) + (MAT ? FFX_MAT_STRIDE : 3) * st[r].shape;
        size_t o00 = ((size_t)st[r].iy0 * ct.tw + st[r].ix0) * tc, o01 = ((size_t)st[r].iy0 * ct.tw + st[r].ix1) * tc;
        size_t o10 = ((size_t)st[r].iy1 * ct.tw + st[r].ix0) * tc, o11 = ((size_t)st[r].iy1 * ct.tw + st[r].ix1) * tc;
        const float wx0 = st[r].wx0, wx1 = st[r].wx1, wy0 = st[r].wy0, wy1 = st[r].wy1;
        if (tc == 1) {
          float ws = (g[r][0] * alb[0] * ct.p_color[0] + g[r][1] * alb[1] * ct.p_color[1] + g[r][2] * alb[2] * ct.p_color[2]) * st[r].proj_fac * inv_spp;
          if constexpr (MAT) {
            if (st[r].proj_fac_b != 0.f) ws += (g[r][0] * ct.p_color[0] + g[r][1] * ct.p_color[1] + g[r][2] * ct.p_color[2]) * st[r].proj_fac_b * inv_spp;
          }
          det_emit(det, gtex, o00, ws * wy0 * wx0);
          det_emit(det, gtex, o01, ws * wy0 * wx1);
          det_emit(det, gtex, o10, ws * wy1 * wx0);
          det_emit(det, gtex, o11, ws * wy1 * wx1);
        } else {
#pragma unroll
          for (int ch = 0; ch < 3; ++ch) {
            float ws = g[r][ch] * alb[ch] * st[r].proj_fac * inv_spp;
            if constexpr (MAT) {
              if (st[r].proj_fac_b != 0.f) ws += g[r][ch] * st[r].proj_fac_b * inv_spp;
            }
            det_emit(det, gtex, o00 + ch, ws * wy0 * wx0);
            det_emit(det, gtex, o01 + ch, ws * wy0 * wx1);
            det_emit(det, gtex, o10 + ch, ws * wy1 * wx0);
            det_emit(det, gtex, o11 + ch, ws * wy1 * wx1);
          }
        }
      }
    }
  }
}

// K9 from the adjoint cache.  Part 1: 32 lanes per pixel slot — lane e < 25 scatters footprint weight e scaled by
// the pixel's  gimg . albedo[shape] (. colour) / spp  with one global float atomic (a lit pixel touches ~16
// texels: ~0.8 M atomics per 512x512 render instead of 4 x 16.8 M sample taps).  Part 2 (the blocks past the
// pixel slots): one lane per stray sample record, four taps each.
struct BwdP { int tw, th, tc, spp; float color[3]; float inv_spp; int W, H; int ms; size_t off_foot_b; // ms: floats per material row (3 / FFX_MAT_STRIDE)
              const void *img; int img_fp16; float *dot_out; int dot_slots; // optional: sum(dot_out[0 .. dot_slots)) += <gimg, img> (the value of a linear loss whose gradient gimg is)
              const float *l1_tgt; float l1_gs, l1_vs; // optional (ffx_render_bwd_cached_l1): gimg = the gradient of l1_vs * sum |img - l1_tgt|, formed per pixel
              const float *mats; int mat_inline; float mat_h[FFX_MAX_MAT_H]; }; // the material table: the caller's device array or (ffx_scene_desc.mat_h) this kernel argument
__device__ __forceinline__ const float *mat_table(const BwdP &k) { return k.mat_inline ? k.mat_h : k.mats; }
// the loss gradient at a pixel: the caller's gimg, or (l1_tgt) the L1 loss's own — sign(img - target) * weight / n, the arithmetic of k_l1_partial
// (ffx_l1_value_grad) — with the pixel's |img - target| for the loss value
__device__ __forceinline__ void k9_g(const BwdP &p, const float *__restrict__ gimg, long pixel, float &g0, float &g1, float &g2, float *absum = nullptr) {
  if (p.l1_tgt) {
    const float *q = (const float *)p.img + pixel * 3, *t = p.l1_tgt + pixel * 3;
    const float d0 = q[0] - t[0], d1 = q[1] - t[1], d2 = q[2] - t[2];
    g0 = d0 > 0.f ? p.l1_gs : (d0 < 0.f ? -p.l1_gs : 0.f);
    g1 = d1 > 0.f ? p.l1_gs : (d1 < 0.f ? -p.l1_gs : 0.f);
    g2 = d2 > 0.f ? p.l1_gs : (d2 < 0.f ? -p.l1_gs : 0.f);
    if (absum) *absum = (fabsf(d0) + fabsf(d1)) + fabsf(d2);
    return;
  }
  g0 = gimg[pixel * 3]; g1 = gimg[pixel * 3 + 1]; g2 = gimg[pixel * 3 + 2];
}
__device__ __forceinline__ float k9_pixel_dot(const BwdP &p, long pixel, const float *__restrict__ gimg) {
  const float g0 = gimg[pixel * 3], g1 = gimg[pixel * 3 + 1], g2 = gimg[pixel * 3 + 2];
  if (p.img_fp16) {
    const _Float16 *q = (const _Float16 *)p.img + pixel * 3;
    return g0 * (float)q[0] + g1 * (float)q[1] + g2 * (float)q[2];
  }
  const float *q = (const float *)p.img + pixel * 3;
  return g0 * q[0] + g1 * q[1] + g2 * q[2];
}

// the stray records of the adjoint cache: thread i replays record i (four bilinear taps each)
__device__ __forceinline__ void k9_stray(const char *__restrict__ cache, long n_pix, uint32_t i, const BwdP &p, const float *__restrict__ gimg,
                                         const float *__restrict__ albedo, float *__restrict__ gtex) {
  const CacheHdr *hdr = reinterpret_cast<const CacheHdr *>(cache);
  // the forward pass ran out of stray records: the gradient of `dropped` samples is missing.  A wrong gradient must not
  // look like a right one: poison gtex (NaN) — ffx_render_cache_status tells the host why, which then re-traces
  // (functional._Render.backward) or raises (optim.PatternOptimizer)
  if (i == 0 && hdr->dropped != 0u) atomicAdd(gtex, __uint_as_float(0x7fc00000u));
  const uint32_t n = min(hdr->n_stray, hdr->cap_stray);
  if (i >= n) return;
  const CacheStray rec = reinterpret_cast<const CacheStray *>(cache + cache_off_arena((size_t)n_pix))[i];
  const long pixel = rec.pix;
  float g0, g1, g2;
  k9_g(p, gimg, pixel, g0, g1, g2);
  const int ix0 = (int)(rec.xy_shape & 0xfffu) - 1, iy0 = (int)((rec.xy_shape >> 12) & 0xfffu) - 1, shape = (int)(rec.xy_shape >> 24);
  const int x0 = clampi(ix0, 0, p.tw - 1), x1 = clampi(ix0 + 1, 0, p.tw - 1), y0 = clampi(iy0, 0, p.th - 1), y1 = clampi(iy0 + 1, 0, p.th - 1);
  const float wx0 = 1.0f - rec.ax, wx1 = rec.ax, wy0 = 1.0f - rec.ay, wy1 = rec.ay;
  const float *alb = albedo + p.ms * shape;
  const float fac_b = p.ms == 3 ? 0.f : rec.fac_b;
  if (p.tc == 1) {
    float ws = (g0 * alb[0] * p.color[0] + g1 * alb[1] * p.color[1] + g2 * alb[2] * p.color[2]) * rec.fac * p.inv_spp;
    if (fac_b != 0.f) ws += (g0 * p.color[0] + g1 * p.color[1] + g2 * p.color[2]) * fac_b * p.inv_spp;
    if (ws == 0.f) return;
    atomicAdd(gtex + (size_t)y0 * p.tw + x0, ws * wy0 * wx0);
    atomicAdd(gtex + (size_t)y0 * p.tw + x1, ws * wy0 * wx1);
    atomicAdd(gtex + (size_t)y1 * p.tw + x0, ws * wy1 * wx0);
    atomicAdd(gtex + (size_t)y1 * p.tw + x1, ws * wy1 * wx1);
  } else {
    const size_t o00 = ((size_t)y0 * p.tw + x0) * 3, o01 = ((size_t)y0 * p.tw + x1) * 3;
    const size_t o10 = ((size_t)y1 * p.tw + x0) * 3, o11 = ((size_t)y1 * p.tw + x1) * 3;
    const float gg[3] = {g0, g1, g2};
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      float ws = gg[ch] * alb[ch] * rec.fac * p.inv_spp;
      if (fac_b != 0.f) ws += gg[ch] * fac_b * p.inv_spp;
      if (ws == 0.f) continue;
      atomicAdd(gtex + o00 + ch, ws * wy0 * wx0);
      atomicAdd(gtex + o01 + ch, ws * wy0 * wx1);
      atomicAdd(gtex + o10 + ch, ws * wy1 * wx0);
      atomicAdd(gtex + o11 + ch, ws * wy1 * wx1);
    }
  }
}

// K9 for 1-channel textures, tiled: a workgroup owns an 8x8-pixel block of the image.  Neighbouring pixels land on
// neighbouring texels, so their 5x5 footprints overlap heavily: the plain kernel issued ~900 k contended global float
// atomics into a 500^2 texture and was bound by them (0.027 ms for 12 MB).  Here the block's footprints are summed
// in a 32x32-texel LDS tile anchored at the smallest window origin of its lit pixels, and every touched texel gets
// ONE global atomic; footprint elements that fall outside the tile (a block on a depth discontinuity) go to memory
// directly.  Summation order inside a block is not fixed (LDS float atomics) — like the global atomics it replaces.
#define K9_TILE 32
__global__ void __launch_bounds__(256)
    k_render_bwd_cached_tiled(BwdP p_by_value, const char *__restrict__ cache, int blocks_x, int tile_blocks, const float *__restrict__ gimg,
                              float *__restrict__ gtex) {
  __shared__ float s_tile[K9_TILE * K9_TILE];
  __shared__ int s_ox, s_oy, s_any;
  const BwdP &p = kernarg_first<BwdP>(); // (read in place: the inline material rows are indexed per lane)
  const float *albedo = mat_table(p);
  const long n_pix = (long)p.W * p.H;
  if ((int)blockIdx.x >= tile_blocks) { // the tail of the grid replays the stray records
    k9_stray(cache, n_pix, (uint32_t)((int)blockIdx.x - tile_blocks) * 256u + threadIdx.x, p, gimg, albedo, gtex);
    return;
  }
  const int bx = (int)blockIdx.x % blocks_x, by = (int)blockIdx.x / blocks_x;
  const CachePix *hdrs = reinterpret_cast<const CachePix *>(cache + 64);
  const CacheFoot *foots = reinterpret_cast<const CacheFoot *>(cache + cache_off_foot((size_t)n_pix));
  if (threadIdx.x == 0) { s_ox = 0x7fffffff; s_oy = 0x7fffffff; s_any = 0; }
  for (int i = threadIdx.x; i < K9_TILE * K9_TILE; i += 256) s_tile[i] = 0.f;
  __syncthreads();
  // pass 1 (threads 0..63, one per pixel): the tile origin
  if (threadIdx.x < 64) {
    const int x = bx * 8 + (threadIdx.x & 7), y = by * 8 + (threadIdx.x >> 3);
    float d = 0.f;
    if (x < p.W && y < p.H) {
      const CachePix hp = hdrs[(long)y * p.W + x];
      if (hp.lit) { atomicMin(&s_ox, (int)hp.x0); atomicMin(&s_oy, (int)hp.y0); s_any = 1; }
      if (p.dot_out) d = k9_pixel_dot(p, (long)y * p.W + x, gimg);
    }
    if (p.dot_out) { // <gimg, img> of this 8x8 block: one wave, one add into the block's OWN slot (the loss of a pattern optimiser's step:
      d = wave_sum64(d); // no separate reduction launch.  4096 atomics on ONE address cost 40 us here — a slot per block costs nothing)
      if (threadIdx.x == 0 && d != 0.f) atomicAdd(p.dot_out + (int)blockIdx.x % p.dot_slots, d);
    }
  }
  __syncthreads();
  if (!s_any) return; // (uniform: nothing lit in this block)
  const int ox = s_ox, oy = s_oy;
  // pass 2: 32 lanes per pixel (25 footprint elements), 8 pixels per iteration
  const int e = threadIdx.x & 31;
  for (int it = 0; it < 8; ++it) {
    const int pl = it * 8 + (threadIdx.x >> 5);
    const int x = bx * 8 + (pl & 7), y = by * 8 + (pl >> 3);
    if (x >= p.W || y >= p.H || e >= 25) continue;
    const long pixel = (long)y * p.W + x;
    const CachePix hp = hdrs[pixel];
    if (!hp.lit) continue;
    const float w = foots[pixel].w[e];
    const float wb = p.ms == 3 ? 0.f : reinterpret_cast<const CacheFoot *>(cache + p.off_foot_b)[pixel].w[e];
    if (w == 0.f && wb == 0.f) continue;
    const float g0 = gimg[pixel * 3], g1 = gimg[pixel * 3 + 1], g2 = gimg[pixel * 3 + 2];
    const float *alb = albedo + p.ms * (int)hp.shape;
    const float ws = (g0 * alb[0] * p.color[0] + g1 * alb[1] * p.color[1] + g2 * alb[2] * p.color[2]) * p.inv_spp;
    float val = ws * w;
    if (wb != 0.f) val += (g0 * p.color[0] + g1 * p.color[1] + g2 * p.color[2]) * p.inv_spp * wb;
    if (val == 0.f) continue;
    const int tx = (int)hp.x0 + e % 5, ty = (int)hp.y0 + e / 5;
    const int lx = tx - ox, ly = ty - oy;
    if (lx < K9_TILE && ly < K9_TILE) atomicAdd(&s_tile[ly * K9_TILE + lx], val); // (lx, ly >= 0 by construction)
    else atomicAdd(gtex + (size_t)ty * p.tw + tx, val);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < K9_TILE * K9_TILE; i += 256) {
    const float v = s_tile[i];
    if (v != 0.f) atomicAdd(gtex + (size_t)(oy + i / K9_TILE) * p.tw + (ox + i % K9_TILE), v);
  }
}

// The same with 16x16-pixel blocks (round 3, the default).  Most of K9's time was not the scatter but finding out that there is nothing to
// scatter: 96 % of the pixels of a dot pattern's render are unlit, and a workgroup of the 8x8 kernel spent an LDS clear, two barriers
// and a dependent header -> atomicMin chain on 64 of its 256 lanes to learn that (4096 workgroups in two rounds: 20 us for 8 MB).
// Here every lane owns a pixel in pass 1 — header, <gimg, img>, a ballot — and a workgroup without a lit pixel ends after ONE round of
// independent loads; the lit pixels are compacted into a list that pass 2 walks 8 at a time (32 lanes per pixel, 25 footprint weights),
// through a 48x48-texel LDS tile anchored at the smallest window origin.
#define K9_TILE16 48
__global__ void __launch_bounds__(256)
    k_render_bwd_cached_tiled16(BwdP p_by_value, const char *__restrict__ cache, int blocks_x, int tile_blocks, const float *__restrict__ gimg,
                                float *__restrict__ gtex) {
  __shared__ float s_tile[K9_TILE16 * K9_TILE16];
  __shared__ int s_ox, s_oy, s_n;
  // the lit pixels of the block, compacted: pixel (lane of pass 1), window origin, and the two per-pixel factors of its footprints
  __shared__ unsigned short s_px[256];
  __shared__ int s_xy[256];
  __shared__ float s_ws[256], s_wsb[256];
  const BwdP &p = kernarg_first<BwdP>();
  const float *albedo = mat_table(p);
  const long n_pix = (long)p.W * p.H;
  if ((int)blockIdx.x >= tile_blocks) { // the tail of the grid replays the stray records
    k9_stray(cache, n_pix, (uint32_t)((int)blockIdx.x - tile_blocks) * 256u + threadIdx.x, p, gimg, albedo, gtex);
    return;
  }
  const int bx = (int)blockIdx.x % blocks_x, by = (int)blockIdx.x / blocks_x;
  const CachePix *hdrs = reinterpret_cast<const CachePix *>(cache + 64);
  const CacheFoot *foots = reinterpret_cast<const CacheFoot *>(cache + cache_off_foot((size_t)n_pix));
  const CacheFoot *foots_b = reinterpret_cast<const CacheFoot *>(cache + p.off_foot_b);
  if (threadIdx.x == 0) { s_ox = 0x7fffffff; s_oy = 0x7fffffff; s_n = 0; }
  __syncthreads();
  // pass 1: one lane per pixel of the 16x16 block
  {
    const int lx = threadIdx.x & 15, ly = threadIdx.x >> 4;
    const int x = bx * 16 + lx, y = by * 16 + ly;
    const bool in = x < p.W && y < p.H;
    const long pixel = (long)y * p.W + x;
    CachePix hp;
    hp.lit = 0;
    if (in) hp = hdrs[pixel];
    const bool lit = in && hp.lit;
    float g0 = 0.f, g1 = 0.f, g2 = 0.f, l1_abs = 0.f;
    if (in && (p.dot_out || lit)) k9_g(p, gimg, pixel, g0, g1, g2, &l1_abs);
    if (p.l1_tgt) { // the L1 loss's value: this wave's 4 rows of |img - target|, one add into the block's own slot (summed by the gradient launch)
      const float d = wave_sum64(l1_abs);
      if ((threadIdx.x & 63) == 0 && d != 0.f) atomicAdd(p.dot_out + (int)blockIdx.x % p.dot_slots, d * p.l1_vs);
    }
    const wmask lm = wballot(lit);
    if (lm != 0ull) { // (per wave) origin of the tile and the wave's share of the list
      int base = 0;
      if ((threadIdx.x & 63) == 0) base = atomicAdd(&s_n, wpop(lm));
      base = __builtin_amdgcn_readfirstlane(base);
      if (lit) {
        atomicMin(&s_ox, (int)hp.x0);
        atomicMin(&s_oy, (int)hp.y0);
        const float *alb = albedo + p.ms * (int)hp.shape;
        const int k = base + (int)mbcnt64(lm);
        s_px[k] = (unsigned short)threadIdx.x;
        s_xy[k] = (int)hp.x0 | ((int)hp.y0 << 16);
        s_ws[k] = (g0 * alb[0] * p.color[0] + g1 * alb[1] * p.color[1] + g2 * alb[2] * p.color[2]) * p.inv_spp;
        s_wsb[k] = (g0 * p.color[0] + g1 * p.color[1] + g2 * p.color[2]) * p.inv_spp;
      }
    }
    if (p.dot_out && !p.l1_tgt) { // <gimg, img> of this wave's 4 rows: one add into the block's own slot (ffx_render_dot_slots)
      float d = 0.f;
      if (in) {
        if (p.img_fp16) { const _Float16 *q = (const _Float16 *)p.img + pixel * 3; d = g0 * (float)q[0] + g1 * (float)q[1] + g2 * (float)q[2]; }
        else { const float *q = (const float *)p.img + pixel * 3; d = g0 * q[0] + g1 * q[1] + g2 * q[2]; }
      }
      d = wave_sum64(d);
      if ((threadIdx.x & 63) == 0 && d != 0.f) atomicAdd(p.dot_out + (int)blockIdx.x % p.dot_slots, d);
    }
  }
  __syncthreads();
  const int n_lit = s_n;
  if (n_lit == 0) return; // (uniform: nothing lit in this block)
  for (int i = threadIdx.x; i < K9_TILE16 * K9_TILE16; i += 256) s_tile[i] = 0.f;
  __syncthreads();
  const int ox = s_ox, oy = s_oy;
  // pass 2: 32 lanes per lit pixel (25 footprint weights), 8 pixels side by side and four of those rounds in flight (the block of a
  // laser dot holds ~140 lit pixels: walked one round at a time its dependent loads were the tail of the whole launch)
  const int e = threadIdx.x & 31;
  const bool has_b = p.ms != 3;
  for (int it0 = threadIdx.x >> 5; it0 < n_lit; it0 += 32) {
    float w[4], wb[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int it = it0 + 8 * u;
      w[u] = 0.f; wb[u] = 0.f;
      if (it < n_lit && e < 25) {
        const int pl = (int)s_px[it];
        const long pixel = (long)(by * 16 + (pl >> 4)) * p.W + (bx * 16 + (pl & 15));
        w[u] = foots[pixel].w[e];
        if (has_b) wb[u] = foots_b[pixel].w[e];
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int it = it0 + 8 * u;
      if (it >= n_lit || (w[u] == 0.f && wb[u] == 0.f)) continue;
      float val = s_ws[it] * w[u];
      if (wb[u] != 0.f) val += s_wsb[it] * wb[u];
      if (val == 0.f) continue;
      const int xy = s_xy[it];
      const int tx = (xy & 0xffff) + e % 5, ty = (xy >> 16) + e / 5;
      const int lx = tx - ox, ly = ty - oy;
      if (lx < K9_TILE16 && ly < K9_TILE16) atomicAdd(&s_tile[ly * K9_TILE16 + lx], val); // (lx, ly >= 0 by construction)
      else atomicAdd(gtex + (size_t)ty * p.tw + tx, val);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < K9_TILE16 * K9_TILE16; i += 256) {
    const float v = s_tile[i];
    if (v != 0.f) atomicAdd(gtex + (size_t)(oy + i / K9_TILE16) * p.tw + (ox + i % K9_TILE16), v);
  }
}

__global__ void __launch_bounds__(256)
    k_render_bwd_cached(BwdP p_by_value, const char *__restrict__ cache, long n_pix, int slot_blocks, const float *__restrict__ gimg, float *__restrict__ gtex) {
  const BwdP &p = kernarg_first<BwdP>();
  const float *albedo = mat_table(p);
  if ((int)blockIdx.x < slot_blocks) {
    const long pixel = (long)blockIdx.x * 8 + (threadIdx.x >> 5);
    const int e = threadIdx.x & 31;
    if (p.dot_out) { // <gimg, img> of the block's 8 pixels: lanes 0..7 of wave 0, one atomic
      __shared__ float s_d[8];
      if (e == 0) s_d[threadIdx.x >> 5] = pixel < n_pix ? k9_pixel_dot(p, pixel, gimg) : 0.f;
      __syncthreads();
      if (threadIdx.x == 0) {
        const float d = ((s_d[0] + s_d[1]) + (s_d[2] + s_d[3])) + ((s_d[4] + s_d[5]) + (s_d[6] + s_d[7]));
        if (d != 0.f) atomicAdd(p.dot_out + (int)blockIdx.x % p.dot_slots, d);
      }
    }
    if (pixel >= n_pix) return;
    const CachePix hp = reinterpret_cast<const CachePix *>(cache + 64)[pixel];
    if (!hp.lit || e >= 25) return;
    const float w = reinterpret_cast<const CacheFoot *>(cache + cache_off_foot((size_t)n_pix))[pixel].w[e];
    const float wb = p.ms == 3 ? 0.f : reinterpret_cast<const CacheFoot *>(cache + p.off_foot_b)[pixel].w[e];
    if (w == 0.f && wb == 0.f) return;
    const float g0 = gimg[pixel * 3], g1 = gimg[pixel * 3 + 1], g2 = gimg[pixel * 3 + 2];
    const float *alb = albedo + p.ms * (int)hp.shape;
    const int x = (int)hp.x0 + e % 5, y = (int)hp.y0 + e / 5;
    if (p.tc == 1) {
      const float ws = (g0 * alb[0] * p.color[0] + g1 * alb[1] * p.color[1] + g2 * alb[2] * p.color[2]) * p.inv_spp;
      float val = ws * w;
      if (wb != 0.f) val += (g0 * p.color[0] + g1 * p.color[1] + g2 * p.color[2]) * p.inv_spp * wb;
      if (val != 0.f) atomicAdd(gtex + (size_t)y * p.tw + x, val);
    } else {
      float *t = gtex + ((size_t)y * p.tw + x) * 3;
      if (g0 != 0.f) atomicAdd(t, g0 * alb[0] * p.inv_spp * w + g0 * p.inv_spp * wb);
      if (g1 != 0.f) atomicAdd(t + 1, g1 * alb[1] * p.inv_spp * w + g1 * p.inv_spp * wb);
      if (g2 != 0.f) atomicAdd(t + 2, g2 * alb[2] * p.inv_spp * w + g2 * p.inv_spp * wb);
    }
    return;
  }
  k9_stray(cache, n_pix, (uint32_t)(blockIdx.x - slot_blocks) * 256u + threadIdx.x, p, gimg, albedo, gtex);
}

// K9 of the filtered film's cache (ffx_render_bwd_cached_filtered; rfc_off_* above).  An ITEM is one lit 64-sample pass of one pixel: the
// lanes take its 64 records (one coalesced 1 KB load), their ten filter weights from the jitter (the forward's hash), the pixel's window of
// G = gimg / weight (lanes 0..24 load a window pixel each, broadcast by v_readlane), the sample's own gradient sum_n w_n G[pixel + n] (the
// arithmetic and the order of k_render_fwd_pk<.., ADJ, RF>); the wave then sums its samples' taps per texel of the pixel's 5x5 texel window
// (25 DPP reductions) and 25 lanes add one value each to gtex — no per-sample atomics (samples outside the window: their four taps directly).
// ~800 VALU instructions per item, and the items of a dot pattern's render sit in ~500 of the film's 4096 8x8-pixel blocks: as ONE workgroup
// per block (first version: 16x16-pixel blocks, then 8x8, four to sixteen waves, an LDS tile per block) a block's ~64 items ran on ONE compute
// unit — 0.16 / 0.09 / 0.065 ms with 250 of 256 CUs idle.  Now every block is served by K9F_SUB independent ONE-WAVE workgroups (the
// dispatcher deals them to different CUs and XCDs): each reads the block's 64 pixel headers itself (512 bytes, L2) and takes every
// K9F_SUB-th lit item.  A block without lit pixels costs its waves one load.  3-channel textures: twelve direct atomics per sample.
struct BwdF { int tw, th, tc, spp; float color[3]; RfC rf; int W, H; int ms, n_shapes; uint32_t seed_key; size_t off_wsum, off_recs, off_facb;
              const float *mats; int mat_inline; float mat_h[FFX_MAX_MAT_H]; };
__device__ __forceinline__ const float *mat_table(const BwdF &k) { return k.mat_inline ? k.mat_h : k.mats; }
#define K9F_BLOCK 8 // pixels per side of a block
#ifndef K9F_SUB
#define K9F_SUB 16 // one-wave workgroups per block (a wave's items are a serial chain of ~1000 instructions each: 4 / 8 / 16 waves per block = 0.102 / 0.055 / 0.035 ms)
#endif
__global__ void __launch_bounds__(64)
    k_render_bwd_cached_filtered(BwdF p_by_value, const char *__restrict__ cache, int blocks_x, const float *__restrict__ gimg, float *__restrict__ gtex) {
  const BwdF &p = kernarg_first<BwdF>();
  const float *albedo = mat_table(p);
  const int blk = (int)blockIdx.x / K9F_SUB, sub = (int)blockIdx.x % K9F_SUB;
  const int bx = blk % blocks_x, by = blk / blocks_x;
  const CachePix *hdrs = reinterpret_cast<const CachePix *>(cache + 64);
  const float *wsum = reinterpret_cast<const float *>(cache + p.off_wsum);
  const uint4 *recs = reinterpret_cast<const uint4 *>(cache + p.off_recs);
  const float *facb = reinterpret_cast<const float *>(cache + p.off_facb);
  const int lane = threadIdx.x;
  uint32_t mask = 0u; // the lane's pixel: its lit passes, the arena block of the first of them, that pass
  uint32_t hfirst = 0u;
  int hpass = 0;
  {
    const int x = bx * K9F_BLOCK + lane % K9F_BLOCK, y = by * K9F_BLOCK + lane / K9F_BLOCK;
    if (x < p.W && y < p.H) {
      const CachePix hp = hdrs[(long)y * p.W + x];
      mask = hp.lit;
      hfirst = (uint32_t)(uint16_t)hp.x0 | ((uint32_t)(uint16_t)hp.y0 << 16);
      hpass = hp.shape;
    }
  }
  // (a forward that found the arena full left pixels without records: the gradient would have holes that look like values — poison it, NaN;
  // ffx_render_cache_status tells the host why, which re-traces (functional._Render.backward) or raises (optim.PatternOptimizer))
  if (blockIdx.x == 0 && lane == 0 && reinterpret_cast<const CacheHdr *>(cache)->dropped != 0u) atomicAdd(gtex, __uint_as_float(0x7fc00000u));
  if (wballot(mask != 0u) == 0ull) return; // (nothing lit in this block: 96 % of a dot pattern's film ends here)
  const int passes = (p.spp + 63) >> 6;
  const bool tiled = p.tc == 1, mat = p.ms != 3;
  // this wave's items: rank r of the block's lit (pass, pixel) pairs in (pass, pixel) order belongs to wave r % K9F_SUB.  State of the walk: the
  // pass, the lit pixels of that pass not yet visited, the rank of the next one — all wave-uniform
  int it_pass = 0, it_rank = 0;
  wmask it_left = wballot((mask & 1u) != 0u);
  auto next_item = [&](int &pl, int &pass) -> bool { // -> this wave's next item, false when the block is exhausted
    for (;;) {
      while (it_left == 0ull) {
        if (++it_pass >= passes) return false;
        it_left = wballot(((mask >> it_pass) & 1u) != 0u);
      }
      const int l = wff1(it_left);
      it_left &= it_left - 1ull;
      if ((it_rank++ % K9F_SUB) == sub) { pl = l; pass = it_pass; return true; }
    }
  };
  // an item's inputs: the lane's record (+ fac_b) and, in lanes 0..24, gimg and the weight of one pixel of the window.  Loaded ONE ITEM AHEAD
  struct Item { uint4 rec; float fb, q0, q1, q2, qw; int px, py; uint32_t sidx; bool ok; };
  auto fetch = [&](Item &o) {
    o.rec = make_uint4(0u, 0u, 0u, 0u); o.fb = 0.f; o.q0 = o.q1 = o.q2 = 0.f; o.qw = 0.f; o.px = o.py = 0; o.sidx = 0u;
    int pl = 0, pass = 0;
    o.ok = next_item(pl, pass);
    if (!o.ok) return;
    o.px = bx * K9F_BLOCK + pl % K9F_BLOCK; o.py = by * K9F_BLOCK + pl / K9F_BLOCK;
    const long pixel = (long)o.py * p.W + o.px;
    const int s = pass * 64 + lane;
    o.sidx = (uint32_t)pixel * (uint32_t)p.spp + (uint32_t)s;
    if (s < p.spp) {
      const uint32_t blk_ = (uint32_t)__builtin_amdgcn_readlane((int)hfirst, pl) + (uint32_t)(pass - __builtin_amdgcn_readlane(hpass, pl));
      o.rec = recs[(size_t)blk_ * 64u + (size_t)lane];
      if (mat) o.fb = facb[(size_t)blk_ * 64u + (size_t)lane];
    }
    const int wb = (lane * 13) >> 6, wa = lane - 5 * wb; // lane / 5, lane % 5 for lane < 25
    const int tx = o.px + wa - 2, ty = o.py + wb - 2;
    if (lane < 25 && tx >= 0 && tx < p.W && ty >= 0 && ty < p.H) {
      const long q = (long)ty * p.W + tx;
      o.qw = wsum[q];
      o.q0 = gimg[q * 3]; o.q1 = gimg[q * 3 + 1]; o.q2 = gimg[q * 3 + 2];
    }
  };
  Item cur, nxt;
  fetch(cur);
  while (cur.ok) { // (wave-uniform loop)
    fetch(nxt);
    const uint4 rec = cur.rec;
    const float fb = cur.fb;
    const float fac = __uint_as_float(rec.w);
    const bool lit = fac != 0.f || fb != 0.f;
    // the pixel's window of G = gimg / weight (lanes 0..24), zero outside the film
    const bool gok = cur.qw > 0.f;
    const float g0 = gok ? cur.q0 / cur.qw : 0.f, g1 = gok ? cur.q1 / cur.qw : 0.f, g2 = gok ? cur.q2 / cur.qw : 0.f;
    float jx, jy, gx[5], gy[5];
    sample_jitter(p.seed_key, cur.sidx, jx, jy);
    rf_weights(p.rf, jx, gx);
    rf_weights(p.rf, jy, gy);
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
#pragma unroll
    for (int n = 0; n < 25; ++n) {
      const float w = gx[n % 5] * gy[n / 5];
      a0 = __builtin_fmaf(w, __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(g0), n)), a0);
      a1 = __builtin_fmaf(w, __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(g1), n)), a1);
      a2 = __builtin_fmaf(w, __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(g2), n)), a2);
    }
    const int ubx = (int)(rec.x & 0xfffu) - 1, uby = (int)((rec.x >> 12) & 0xfffu) - 1, shape = (int)(rec.x >> 24);
    const int x0 = clampi(ubx, 0, p.tw - 1), x1 = clampi(ubx + 1, 0, p.tw - 1), y0 = clampi(uby, 0, p.th - 1), y1 = clampi(uby + 1, 0, p.th - 1);
    const float ax = __uint_as_float(rec.y), ay = __uint_as_float(rec.z);
    const float wx0 = 1.0f - ax, wx1 = ax, wy0 = 1.0f - ay, wy1 = ay;
    const float *alb = albedo + p.ms * (lit ? shape : 0);
    if (tiled) {
      float pf = 0.f;
      if (lit) {
        pf = (a0 * alb[0] * p.color[0] + a1 * alb[1] * p.color[1] + a2 * alb[2] * p.color[2]) * fac;
        if (fb != 0.f) pf += (a0 * p.color[0] + a1 * p.color[1] + a2 * p.color[2]) * fb;
      }
      const bool on = lit && pf != 0.f;
      if (wballot(on) != 0ull) { // (wave-uniform)
        const int fox = (int)wave_reduce_nn<false>(on ? (uint32_t)x0 : 0xffffffffu), foy = (int)wave_reduce_nn<false>(on ? (uint32_t)y0 : 0xffffffffu);
        const bool in_win = on && x1 - fox <= 4 && y1 - foy <= 4;
        const int cx0 = x0 - fox, cx1 = x1 - fox, cy0 = y0 - foy, cy1 = y1 - foy;
        float cw[5], rw[5];
#pragma unroll
        for (int c = 0; c < 5; ++c) {
          cw[c] = in_win ? ((cx0 == c ? wx0 : 0.f) + (cx1 == c ? wx1 : 0.f)) : 0.f; // (a clamped border tap may name the same texel twice: both weights count)
          rw[c] = in_win ? pf * ((cy0 == c ? wy0 : 0.f) + (cy1 == c ? wy1 : 0.f)) : 0.f;
        }
        float v[27];
#pragma unroll
        for (int n = 0; n < 25; ++n) v[n] = rw[n / 5] * cw[n % 5];
        v[25] = v[26] = 0.f;
#pragma unroll
        for (int k = 0; k < 9; ++k) asm(FFX_R3_ALL("v_add_f32_dpp") : "+v"(v[3 * k]), "+v"(v[3 * k + 1]), "+v"(v[3 * k + 2])); // (the wave's sums in lane 63)
        float mine = 0.f;
#pragma unroll
        for (int n = 0; n < 25; ++n) {
          const float sn = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(v[n]), 63));
          if (lane == n) mine = sn;
        }
        if (mine != 0.f) { // (lanes 0..24: one texel each)
          const int ey = (lane * 13) >> 6, ex = lane - 5 * ey;
          atomicAdd(gtex + (size_t)(foy + ey) * p.tw + (fox + ex), mine);
        }
        if (on && !in_win) { // a sample outside its pixel's window (depth discontinuities, grazing surfaces: ~0.1 %): its four taps directly
          atomicAdd(gtex + (size_t)y0 * p.tw + x0, pf * wy0 * wx0);
          atomicAdd(gtex + (size_t)y0 * p.tw + x1, pf * wy0 * wx1);
          atomicAdd(gtex + (size_t)y1 * p.tw + x0, pf * wy1 * wx0);
          atomicAdd(gtex + (size_t)y1 * p.tw + x1, pf * wy1 * wx1);
        }
      }
    } else if (lit) { // 3-channel textures: twelve direct atomics per sample
      const size_t o00 = ((size_t)y0 * p.tw + x0) * 3, o01 = ((size_t)y0 * p.tw + x1) * 3, o10 = ((size_t)y1 * p.tw + x0) * 3, o11 = ((size_t)y1 * p.tw + x1) * 3;
      const float aa[3] = {a0, a1, a2};
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) {
        float cv = aa[ch] * alb[ch] * fac;
        if (fb != 0.f) cv += aa[ch] * fb;
        if (cv != 0.f) {
          atomicAdd(gtex + o00 + ch, cv * wy0 * wx0);
          atomicAdd(gtex + o01 + ch, cv * wy0 * wx1);
          atomicAdd(gtex + o10 + ch, cv * wy1 * wx0);
          atomicAdd(gtex + o11 + ch, cv * wy1 * wx1);
        }
      }
    }
    cur = nxt;
  }
}

// ------------------------------------------------------------------------------------------ host side
static int cam_prepare(const ffx_camera *c, CamK &k) {
  if (c->width < 1 || c->height < 1) return 0;
  if (!ffx_inv4(c->camera_to_sample, k.s2c)) return 0;
  for (int i = 0; i < 12; ++i) k.tw[i] = c->to_world[i];
  k.near_clip = c->near_clip;
  k.far_clip = c->far_clip;
  k.W = c->width;
  k.H = c->height;
  k.inv_w = 1.0f / (float)c->width;
  k.inv_h = 1.0f / (float)c->height;
  k.w_uniform = (k.s2c[12] == 0.f && k.s2c[13] == 0.f && k.s2c[15] != 0.f) ? 1 : 0;
  k.iw_u = k.w_uniform ? 1.0f / k.s2c[15] : 0.f;
  return 1;
}

static int shade_prepare(const ffx_scene_desc *sd, ShadeK &c) {
  memset(&c, 0, sizeof c);
  if (!cam_prepare(&sd->cam, c.cam)) return 0;
  c.proj_on = sd->proj.enabled;
  c.spot_on = sd->spot.enabled;
  c.shadows = sd->shadows & FFX_SHADOWS_ON; // (the word's other bits are hints to the pre-pass and the caches)
  c.mat_stride = sd->mat_stride ? sd->mat_stride : 3;
  if (c.mat_stride != 3 && c.mat_stride != FFX_MAT_STRIDE) return 0;
  if (sd->n_mat_h > 0) { // the material table travels with the call (ffx_scene_desc.mat_h)
    if (sd->n_mat_h > FFX_MAX_MAT_H || sd->n_mat_h != sd->n_shapes * c.mat_stride) return 0;
    c.mat_inline = 1;
    for (int i = 0; i < sd->n_mat_h; ++i) c.mat_h[i] = sd->mat_h[i];
    const char *pe = getenv("FFX_MAT_PRE");
    if (c.mat_stride == FFX_MAT_STRIDE && sd->n_shapes <= 8 && !(pe && strcmp(pe, "0") == 0)) { // per-row constants of the principled rows (ShadeK.mat_pre)
      c.mat_pre_on = 1;
      for (int k = 0; k < sd->n_shapes; ++k) {
        const float *m = sd->mat_h + k * FFX_MAT_STRIDE;
        float *p = c.mat_pre + k * FFX_MAT_PRE;
        const float eta = m[FFX_MAT_ETA], r2 = m[FFX_MAT_ROUGHNESS] * m[FFX_MAT_ROUGHNESS], a = r2 > 0.001f ? r2 : 0.001f, a2 = a * a;
        const float metallic = m[FFX_MAT_METALLIC], m1 = 1.0f - metallic, tint = m[FFX_MAT_SPEC_TINT];
        const float r0 = (eta - 1.0f) / (eta + 1.0f);
        uint32_t flags = 0;
        if (m[FFX_MAT_ANISOTROPIC] != 0.f) flags |= FFX_PRE_ANISO;
        if (tint != 0.f) flags |= FFX_PRE_TINT;
        if (m[FFX_MAT_CLEARCOAT] > 0.f) flags |= FFX_PRE_CLEARCOAT;
        if (m[FFX_MAT_FLATNESS] > 0.f) flags |= FFX_PRE_FLAT;
        if (m[FFX_MAT_SHEEN] > 0.f && m1 > 0.f) flags |= FFX_PRE_SHEEN;
        p[0] = eta; p[1] = 1.0f / (eta * eta); p[2] = a2; p[3] = 1.0f / a2;
        p[4] = metallic; p[5] = metallic + m1 * tint; p[6] = m1 * (1.0f - tint); p[7] = m1 * (1.0f - m[FFX_MAT_SPEC_TRANS]);
        p[8] = 2.0f * m[FFX_MAT_ROUGHNESS]; memcpy(&p[9], &flags, 4); p[10] = r2; p[11] = m1 * tint * r0 * r0;
      }
    }
  }
  if (sd->rfilter == FFX_RFILTER_GAUSSIAN) { // [EXT Mitsuba src/rfilters/gaussian.cpp] radius 4 stddev; the 5x5 window holds radius <= 2
    const float sdv = sd->rfilter_stddev > 0.f ? sd->rfilter_stddev : 0.5f;
    if (!(sdv <= 0.5f)) return 0;
    rf_constants(sdv, c.rf);
  } else if (sd->rfilter != FFX_RFILTER_BOX) return 0;
  c.n_base_tex = sd->n_base_tex;
  if (c.n_base_tex < 0 || c.n_base_tex > FFX_MAX_BASE_TEX || (c.n_base_tex > 0 && (c.mat_stride != FFX_MAT_STRIDE || !sd->slot_uv))) return 0;
  c.slot_uv = sd->slot_uv;
  for (int k = 0; k < c.n_base_tex; ++k) {
    c.btex[k] = sd->base_tex[k]; c.btw[k] = sd->base_tex_w[k]; c.bth[k] = sd->base_tex_h[k];
    if (!c.btex[k] || c.btw[k] < 1 || c.bth[k] < 1) return 0;
  }
  float inv[16];
  if (c.proj_on) {
    if (!ffx_inv4(sd->proj.to_world, inv)) return 0;
    for (int i = 0; i < 12; ++i) c.p_w2l[i] = inv[i];
    for (int i = 0; i < 16; ++i) c.p_c2s[i] = sd->proj.camera_to_sample[i];
    c.p_pos[0] = sd->proj.to_world[3]; c.p_pos[1] = sd->proj.to_world[7]; c.p_pos[2] = sd->proj.to_world[11];
    c.p_axis[0] = sd->proj.to_world[2]; c.p_axis[1] = sd->proj.to_world[6]; c.p_axis[2] = sd->proj.to_world[10];
    c.p_scale = sd->proj.scale;
    for (int i = 0; i < 3; ++i) c.p_color[i] = sd->proj.color[i];
    c.tw = sd->proj.tex_w; c.th = sd->proj.tex_h; c.tc = sd->proj.tex_channels;
    if (c.tw < 1 || c.th < 1 || (c.tc != 1 && c.tc != 3)) return 0;
  }
  if (c.spot_on) {
    if (!ffx_inv4(sd->spot.to_world, inv)) return 0;
    for (int i = 0; i < 12; ++i) c.s_w2l[i] = inv[i];
    {
      double worst = 0.0;
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
          double d = 0.0;
          for (int k = 0; k < 3; ++k) d += (double)inv[4 * i + k] * (double)inv[4 * j + k];
          worst = fmax(worst, fabs(d - (i == j ? 1.0 : 0.0)));
        }
      c.s_rigid = worst < 2e-6 ? 1 : 0;
    }
    c.s_pos[0] = sd->spot.to_world[3]; c.s_pos[1] = sd->spot.to_world[7]; c.s_pos[2] = sd->spot.to_world[11];
    for (int i = 0; i < 3; ++i) c.s_int[i] = sd->spot.intensity[i];
    const float deg = 0.017453292519943295f;
    c.cutoff = sd->spot.cutoff_deg * deg;
    float beam = sd->spot.beam_width_deg * deg;
    c.cos_cut = cosf(c.cutoff);
    c.cos_beam = cosf(beam);
    c.inv_trans = 1.0f / (c.cutoff - beam);
  }
  return 1;
}

// Apex records (ffx_common.h): one lane per leaf slot writes the triangle's (A, B, C, T) for each of the
// requested apexes.  Runs in front of every packet render / trace launch on the same stream: the apex
// areas at the end of the blob are scratch owned by the most recent call (calls that share a blob must
// be stream-ordered, as they already are for the records themselves).  53 k triangles x 3 apexes: 7.7 MB.
struct ApexK { float o[FFX_N_APEX][3]; int on[FFX_N_APEX]; };
__global__ void __launch_bounds__(256)
    k_apex_records(const TriRec *__restrict__ recs, int n_tris, ApexK ak, TriApex *__restrict__ out, uint32_t astride, uint32_t *__restrict__ cache_hdr,
                   uint32_t cap_stray) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k == 0 && cache_hdr) { // adjoint cache: stray arena empty (CacheHdr)
    cache_hdr[0] = 0u; cache_hdr[1] = cap_stray & ~FFX_CAP_KEEP_DROPPED;
    if (!(cap_stray & FFX_CAP_KEEP_DROPPED)) cache_hdr[2] = 0u;
    for (int i = 4; i < 12; ++i) cache_hdr[i] = 0u; // (the filtered film's cache: its arena's eight counters)
  }
  if (k >= n_tris) return;
  const float4 *r4 = reinterpret_cast<const float4 *>(recs + k);
  const float4 a = r4[0], b = r4[1], c = r4[2];
  const v3 v0 = V3(a.x, a.y, a.z), e1 = V3(a.w, b.x, b.y), e2 = V3(b.z, b.w, c.x);
#pragma unroll
  for (int j = 0; j < FFX_N_APEX; ++j) {
    if (!ak.on[j]) continue;
    const ApexVec av = apex_vectors(v0, e1, e2, V3(ak.o[j][0], ak.o[j][1], ak.o[j][2]));
    float4 *o4 = reinterpret_cast<float4 *>(reinterpret_cast<char *>(out) + (size_t)j * astride) + 3 * (size_t)k;
    o4[0] = make_float4(av.A.x, av.A.y, av.A.z, av.B.x);
    o4[1] = make_float4(av.B.y, av.B.z, av.C.x, av.C.y);
    o4[2] = make_float4(av.C.z, av.T, c.y, c.z); // prim, shape as in TriRec
  }
}

// FFX_TRAVERSAL=lane selects the per-lane (LDS stack) kernels; default: wave-packet kernels
// (the knobs are read at every launch: cheap, and lets one process exercise every variant)
static int use_packet() {
  const char *e = getenv("FFX_TRAVERSAL");
  return (e && strcmp(e, "lane") == 0) ? 0 : 1;
}

// independent waves per workgroup of the packet kernels: FFX_PACKET_WAVES = 1 (default), 2 or 4.
// The waves of a workgroup never cooperate, so the smallest workgroup gives the dispatcher the finest
// grain (measured 1 / 2 / 4 waves: 0.882 / 0.891 / 0.951 ms per step on the vocal fold, 16.2 / 16.5 /
// 17.4 ms on the colon).
static int packet_waves() { return 1; }

// FFX_WIDE=0 selects the binary packet walks (A/B baseline); default: the 64-wide walk
static int use_wide(const ffx_bvh_info *info) {
  const char *e = getenv("FFX_WIDE");
  return (e && strcmp(e, "0") == 0) || info->off_tq == 0 ? 0 : 1;
}
static WideScene wide_scene(const void *bvh, const ffx_bvh_info *info) {
  WideScene ws;
  const char *base = (const char *)bvh;
  ws.elems = (const WideChild *)(base + info->off_wnodes);
  ws.hdr = (const WideHdr *)(base + info->off_whdr);
  ws.root = info->wide_root;
  ws.tq0 = (uint32_t)((info->off_tq - info->off_wnodes) / sizeof(WideChild));
  return ws;
}

// log2 of the side (in 2x2-pixel tiles) of the square blocks in which tiles are enumerated: FFX_TILE_BLOCK,
// default 4 = 32x32-pixel blocks since the tile bins (round 4: 3 / 4 / 5 = 0.4033 / 0.4001 / 0.42 ms for K8 alone, renders/s 2 469 / 2 500,
// gradient steps/s 2 288 / 2 354, colon 8.39 / 8.31 ms — a block of 16 bin tiles keeps a workgroup round inside fewer tile lists);
// before them 3 = 16x16-pixel blocks (+1 % on both workloads against row-major; the blocked XCD interleave on top of it,
// FFX_XCD_REMAP >= 2: B = 64 / 128 / 256 / 512 / 1024 measured +0.5 / +1 / -1 / -2 / -10 % against the round-robin deal)
// (films above 512 k pixels — config 5: 1024^2 x 256 spp, 141 MB of tile-list entries per pose — keep the round-3 parameters: the larger block and
// the second pixel per wave buy them nothing (8.39 / 8.31 ms) and double the kernel's HBM traffic, 168 -> 343 MB of FETCH_SIZE per launch)
// The kernels that fold footprints (cache / forward + adjoint) take the larger block there too: colon 114.5 -> 126.5 gradient steps/s.
static int tile_block_log2(long pixels = 0, bool with_cache = false) {
  const char *e = getenv("FFX_TILE_BLOCK");
  // (round 5, tools/trafficsweep.sh: 64x64-pixel blocks with 1024 workgroups per XCD and round — half a block, 32 camera tiles, on one L2 — read
  // 45 MB per launch instead of 67 at the same speed (2 530 vs 2 527 - 2 558 renders/s); the floor is 40 MB: the tile lists are 16.6 MB of
  // 64-byte entries that a render reads once, and the counter correction doubles them.  Coarser deals — one block or more per XCD — reach
  // the floor but lose 14 % to load imbalance)
  const int dflt = (pixels > 2L * 512 * 512) ? (with_cache ? 4 : 3) : 5;
  int t = e ? atoi(e) : dflt;
  return (t < 0 || t > 8) ? dflt : t;
}

// pixels of its 2x2 tile a wave of k_render_fwd_pk walks: FFX_PIXELS_PER_WAVE = 1, 2 (default) or 4.
// Measured 4 / 2 / 1: 1133 / 1180 / 1165 renders/s (vocal fold), 62.0 / 62.6 / 62.2 (colon): shorter
// waves even out the tail of the launch, one pixel per wave pays the wave start-up four times.
// default: one pixel per wave for the plain forward (finest grain: +1-2 %), two for the cache-writing forward (whose
// per-pixel footprint bookkeeping amortises better: one pixel per wave was 1.5 % slower there); colon: no difference
static int pixels_per_wave(bool with_cache, long pixels = 0) {
  const char *e = getenv("FFX_PIXELS_PER_WAVE");
  // (round 4, with the tile bins and 32x32-pixel enumeration blocks: two pixels per wave for the plain forward too — 1 / 2 / 4 pixels =
  // 2 450 / 2 610 / 2 600 renders/s over 100 steps, 2 390 / 2 500 / 2 500 over the driver's 20, K8 alone 0.399 / 0.387 / 0.419 ms; colon
  // 117.9 / 118.8 / 106.2 renders/s.  Half as many waves to dispatch, and a wave's second pixel finds its tile list in the L1)
  const int dflt = (with_cache || pixels <= 2L * 512 * 512) ? 2 : 1;
  int w = e ? atoi(e) : dflt;
  return (w == 1 || w == 2 || w == 4) ? w : dflt;
}

// experiment knob: dynamic LDS bytes per workgroup of the packet kernels (unused by the kernel; it only
// lowers occupancy so that latency- and throughput-bound behaviour can be told apart)
// renders below 33 samples per pixel take k_render_fwd_blk (several pixels per wave); FFX_RENDER_BLOCKS=0: a pixel per wave whatever the count
static bool lowspp_blocks() {
  const char *e = getenv("FFX_RENDER_BLOCKS");
  return !(e && strcmp(e, "0") == 0);
}
static size_t dummy_lds() {
  const char *e = getenv("FFX_DUMMY_LDS");
  return e ? (size_t)atol(e) : 0;
}

// ---- tile bins: the three grids of a scene description (ffx_common.h BinGrid).  A pure function of sd (and the environment), so that
// ffx_apex_prepare and the render calls that follow it derive the same grids.  A grid is switched off (its packets walk the tree)
// when its projection is not a central perspective, when a spot's cone is too wide for a perspective grid, or by FFX_BINS=0.
static int bins_enabled() {
  const char *e = getenv("FFX_BINS");
  return !(e && strcmp(e, "0") == 0);
}
static int env_pow2(const char *name, int dflt, int lo, int hi) {
  const char *e = getenv(name);
  int v = e ? atoi(e) : dflt;
  if (v < lo || v > hi || (v & (v - 1)) != 0) v = dflt;
  return v;
}
// rows of M for a sensor-like apex: tile coordinates = (film / tile) * sample coordinates of camera_to_sample * world_to_camera
static int grid_from_sensor(const float *to_world, const float *c2s, int width, int height, int tile, BinGrid &g) {
  memset(&g, 0, sizeof g);
  if (width < 1 || height < 1) return 0;
  float w2c[16];
  if (!ffx_inv4(to_world, w2c)) return 0;
  // a central projection: sample = (r0 . p, r1 . p) / (r3 . p) with no dependence on the homogeneous 1 of a camera-space point
  if (c2s[3] != 0.f || c2s[7] != 0.f || c2s[15] != 0.f || !(c2s[14] > 0.f)) return 0;
  while ((width + tile - 1) / tile > 128 || (height + tile - 1) / tile > 128) tile *= 2;
  g.nx = (width + tile - 1) / tile;
  g.ny = (height + tile - 1) / tile;
  if ((long)g.nx * g.ny > FFX_BIN_MAX_TILES) return 0;
  const double sx = (double)width / tile, sy = (double)height / tile;
  for (int j = 0; j < 3; ++j) {
    double x = 0, y = 0, z = 0;
    for (int k = 0; k < 3; ++k) {
      x += (double)c2s[k] * w2c[4 * k + j];
      y += (double)c2s[4 + k] * w2c[4 * k + j];
      z += (double)c2s[12 + k] * w2c[4 * k + j];
    }
    g.M[j] = (float)(sx * x); g.M[3 + j] = (float)(sy * y); g.M[6 + j] = (float)z;
  }
  g.o[0] = to_world[3]; g.o[1] = to_world[7]; g.o[2] = to_world[11];
  g.on = 1;
  return tile;
}
static void bins_grids(const ffx_scene_desc *sd, BinGrid (&g)[FFX_N_APEX], float &cam_its_x, float &cam_its_y) {
  memset(g, 0, sizeof g);
  cam_its_x = cam_its_y = 0.f;
  if (!sd || !bins_enabled()) return;
  const int ts = env_pow2("FFX_BIN_TILE", 8, 4, 32);
  const int tile = grid_from_sensor(sd->cam.to_world, sd->cam.camera_to_sample, sd->cam.width, sd->cam.height, ts, g[0]);
  if (tile > 0) cam_its_x = cam_its_y = 1.0f / (float)tile;
  if (sd->proj.enabled) grid_from_sensor(sd->proj.to_world, sd->proj.camera_to_sample, sd->proj.tex_w, sd->proj.tex_h, env_pow2("FFX_BIN_TILE_PROJ", 2 * ts, 4, 64), g[1]);
  if (sd->spot.enabled && sd->spot.cutoff_deg > 0.f && sd->spot.cutoff_deg <= 75.f) {
    // a square perspective grid around the cone's axis: half angle = cutoff + 1 degree, ~1 degree per tile at the centre
    float w2l[16];
    if (ffx_inv4(sd->spot.to_world, w2l)) {
      const char *e = getenv("FFX_BIN_SPOT_N");
      // (~1 degree per tile; 1.7 degrees under the caller's hint that this pose's renders are short — FFX_SHADOWS_PLAIN, below 33 samples per
      // pixel: such a render waits for the pre-pass chain, and a coarser grid lists a third fewer (triangle, tile) pairs: 16 spp 5 510 ->
      // 6 519 renders/s, 4 spp 5 657 -> 6 556; a long render prefers the finer grid's shorter lists: 64 spp 2 964 against 2 921)
      int n = e ? atoi(e) : (int)(((sd->shadows & FFX_SHADOWS_PLAIN) ? 1.2f : 2.0f) * sd->spot.cutoff_deg + 0.999f);
      n = n < 8 ? 8 : (n > 128 ? 128 : n);
      const double tanc = tan(((double)sd->spot.cutoff_deg + 1.0) * 0.017453292519943295);
      BinGrid &q = g[2];
      for (int j = 0; j < 3; ++j) {
        q.M[j] = (float)(0.5 * n * (w2l[j] / tanc + w2l[8 + j]));
        q.M[3 + j] = (float)(0.5 * n * (w2l[4 + j] / tanc + w2l[8 + j]));
        q.M[6 + j] = w2l[8 + j];
      }
      q.o[0] = sd->spot.to_world[3]; q.o[1] = sd->spot.to_world[7]; q.o[2] = sd->spot.to_world[11];
      q.nx = q.ny = n;
      q.on = 1;
    }
  }
}
// FFX_SHADOW_CLEAR=0: the render kernels walk every shadow packet (A/B and the tests' reference for the skip; the pre-pass then leaves the
// bits cleared).  A pure function of (sd, info, environment), like the grids: ffx_apex_prepare and the renders behind it agree.
// -> a mask over the emitters: bit 0 projector, bit 1 spot.  DEFAULT 0 (off) — a measured negative result, kept as an opt-in because it is
// exact and tested: without the spot's any-hit stage K8 runs 0.400 -> 0.302 ms (-DFFX_EXP_NO_SPOT_SHADOW), but the proof only succeeds for
// 22 - 36 % of the vocal fold's triangles (a smooth surface is half saddle: there a triangle's vertex neighbours straddle its plane and it
// theirs — neither H1 nor H2 — and the margin between the lift of a shadow ray's end point, 8.9e-5 (1 + |P|), and the ignored tail of the
// ray, 8.9e-4 |d|, is too small to settle them by distance), so ~20 % of the pixels skip the stage: K8 0.400 -> 0.381 ms (tools/k8ab.py),
// while the pairwise proof adds ~10 M instructions to the pre-pass on the side stream, 280 us elapsed beside a render (rocprofv3): the loop
// fell from 2 400 to 1 810 renders/s (spot only; 1 580 with the projector's 68-entry tiles too).  FFX_SHADOW_CLEAR=2 / 3 switches it on.
static int clear_enabled(const ffx_scene_desc *sd, const ffx_bvh_info *info) {
  const char *e = getenv("FFX_SHADOW_CLEAR");
  if (!(sd && (sd->shadows & FFX_SHADOWS_ON) && info->off_gn != 0 && bins_enabled())) return 0;
  const int m = e ? atoi(e) : 0;
  return (m < 0 || m > 3) ? 0 : m;
}
// FFX_ENVELOPE=0: no envelopes (every shadow packet runs its any-hit stage: the A/B baseline and the tests' reference).  -> a mask over the
// emitters (bit 0 projector, bit 1 spot).  A pure function of (sd, info, environment), like the grids: the pre-pass and the renders behind it agree.
static int env_enabled(const ffx_scene_desc *sd, const ffx_bvh_info *info) {
  const char *e = getenv("FFX_ENVELOPE");
  if (!(sd && (sd->shadows & FFX_SHADOWS_ON) && bins_enabled() && info->off_bins)) return 0;
  if (sd->shadows & FFX_SHADOWS_PLAIN) return 0; // (the caller's hint: this pose's renders are short — the envelope launch would only lengthen the pre-pass chain)
  if (info->bins_stride < ffx_bin_stride(info->n_tris) || ffx_bin_off_env(info->n_tris) >= (1ull << 32)) return 0; // (a blob of another library version)
  const int m = e ? atoi(e) : 3;
  return (m < 0 || m > 3) ? 0 : m;
}
// the envelope launch's constants for the grids g (ffx_common.h EnvBuild); an emitter whose grid is off or not invertible drops out of the mask
static int env_build(const BinGrid (&g)[FFX_N_APEX], int mask, int n_tris, EnvBuild &eb) {
  memset(&eb, 0, sizeof eb);
  eb.env_off = (uint32_t)ffx_bin_off_env(n_tris);
  eb.kap = (float)((1.0 - (double)SHADOW_EPS) * (1.0 + 6e-5));
  int out = 0;
  for (int a = 1; a < FFX_N_APEX; ++a) {
    if (!((mask >> (a - 1)) & 1) || !g[a].on) continue;
    const float *M = g[a].M;
    const double m[9] = {M[0], M[1], M[2], M[3], M[4], M[5], M[6], M[7], M[8]};
    const double det = m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
    if (!(fabs(det) > 1e-30)) continue;
    const double inv[9] = {(m[4] * m[8] - m[5] * m[7]) / det, (m[2] * m[7] - m[1] * m[8]) / det, (m[1] * m[5] - m[2] * m[4]) / det,
                           (m[5] * m[6] - m[3] * m[8]) / det, (m[0] * m[8] - m[2] * m[6]) / det, (m[2] * m[3] - m[0] * m[5]) / det,
                           (m[3] * m[7] - m[4] * m[6]) / det, (m[1] * m[6] - m[0] * m[7]) / det, (m[0] * m[4] - m[1] * m[3]) / det};
    double umax = 0.0; // |Minv (x, y, 1)| is convex in (x, y): its maximum over the grid sits at a corner
    for (int c = 0; c < 4; ++c) {
      const double x = (c & 1) ? g[a].nx : 0.0, y = (c & 2) ? g[a].ny : 0.0;
      const double ux = inv[0] * x + inv[1] * y + inv[2], uy = inv[3] * x + inv[4] * y + inv[5], uz = inv[6] * x + inv[7] * y + inv[8];
      umax = fmax(umax, sqrt(ux * ux + uy * uy + uz * uz));
    }
    for (int k = 0; k < 9; ++k) eb.Minv[a][k] = (float)inv[k];
    eb.graz[a] = (float)(umax * 1.001 / 40.0);
    eb.on[a] = 1;
    out |= 1 << (a - 1);
  }
  return out;
}
// the kernels' view of the bins of `sd` in the blob (grids as bins_grids gives them; built by launch_apex / ffx_apex_prepare)
static void bins_k(const void *bvh, const ffx_bvh_info *info, const ffx_scene_desc *sd, BinsK &bk) {
  memset(&bk, 0, sizeof bk);
  if (!info->off_bins || !info->bins_stride) return;
  bins_grids(sd, bk.g, bk.cam_inv_ts_x, bk.cam_inv_ts_y);
  for (int a = 0; a < FFX_N_APEX; ++a) bk.base[a] = (const char *)bvh + info->off_bins + (uint64_t)a * info->bins_stride;
  bk.clear_on = clear_enabled(sd, info);
  EnvBuild eb;
  bk.env_on = env_build(bk.g, env_enabled(sd, info), info->n_tris, eb);
  bk.env_off = eb.env_off;
}

// fills the blob's apex areas for the camera (and the enabled emitters of sd, if given) on stream s
// (cap_stray's top bit: FFX_RENDER_CACHE_KEEP_DROPPED — the arena is emptied, the header's `dropped` count stays: ffx_common.h FFX_CAP_KEEP_DROPPED)
__global__ void k_cache_reset(uint32_t *__restrict__ cache_hdr, uint32_t cap_stray) {
  cache_hdr[0] = 0u; cache_hdr[1] = cap_stray & ~FFX_CAP_KEEP_DROPPED;
  if (!(cap_stray & FFX_CAP_KEEP_DROPPED)) cache_hdr[2] = 0u;
  for (int i = 4; i < 12; ++i) cache_hdr[i] = 0u; // (the filtered film's cache: its arena's eight counters)
}
// flags: FFX_RENDER_APEX_READY — the areas already hold this call's apexes (ffx_apex_prepare, or an earlier call with the same
// origins on records that have not changed since): no launch; FFX_RENDER_CACHE_ZEROED — the caller has cleared the cache header.
static int launch_apex(const void *bvh, const ffx_bvh_info *info, const float *cam_to_world, const ffx_scene_desc *sd, const TriApex **arecs,
                       uint32_t *astride, hipStream_t s, void *cache = nullptr, uint32_t cap_stray = 0, int flags = 0) {
  ApexK ak;
  memset(&ak, 0, sizeof ak);
  ak.on[0] = 1;
  ak.o[0][0] = cam_to_world[3]; ak.o[0][1] = cam_to_world[7]; ak.o[0][2] = cam_to_world[11];
  if (sd && sd->proj.enabled) {
    ak.on[1] = 1;
    ak.o[1][0] = sd->proj.to_world[3]; ak.o[1][1] = sd->proj.to_world[7]; ak.o[1][2] = sd->proj.to_world[11];
  }
  if (sd && sd->spot.enabled) {
    ak.on[2] = 1;
    ak.o[2][0] = sd->spot.to_world[3]; ak.o[2][1] = sd->spot.to_world[7]; ak.o[2][2] = sd->spot.to_world[11];
  }
  const uint64_t stride = ffx_apex_stride(info->n_tris);
  if (info->total_bytes < info->off_recs + FFX_N_APEX * stride || stride >= (1ull << 32)) {
    ffx_set_error("blob has no apex areas (built by another library version?)");
    return 0;
  }
  TriApex *out = (TriApex *)((char *)bvh + ffx_apex_offset(info, 0));
  const TriRec *recs = (const TriRec *)((const char *)bvh + info->off_recs);
  if (flags & FFX_RENDER_CACHE_ZEROED) cache = nullptr;
  if (flags & FFX_RENDER_CACHE_KEEP_DROPPED) cap_stray |= FFX_CAP_KEEP_DROPPED; // (the reset launches' view of the flag; capacities stay far below 2^31)
  if (!(flags & FFX_RENDER_APEX_READY)) {
    // ONE pre-pass for both: the counting launch of the tile bins writes the apex records too (ffx_bins.hip k_bin<false>), the fill
    // launch follows when a grid is on.  A blob without a bins area (or a primary-visibility call: sd == NULL) gets the apex records alone.
    BinBuild bb;
    memset(&bb, 0, sizeof bb);
    if (sd && info->off_bins && info->bins_stride >= ffx_bin_stride(info->n_tris)) {
      float ix, iy;
      bins_grids(sd, bb.g, ix, iy);
      for (int a = 0; a < FFX_N_APEX; ++a) bb.base[a] = (char *)bvh + info->off_bins + (uint64_t)a * info->bins_stride;
      bb.cap = (uint32_t)ffx_bin_cap(info->n_tris);
      if (const char *ce = getenv("FFX_BIN_CAP")) { // (test knob, include/ffx.h: a smaller capacity makes a grid's lists overflow — its packets then walk the tree)
        const long cv = atol(ce);
        if (cv >= 0 && (uint64_t)cv < bb.cap) bb.cap = (uint32_t)cv;
      }
      bb.arrive = (uint32_t *)(bb.base[0] + offsetof(BinHdr, pad0));
    }
    if (!bb.arrive) { // no bins area to hold the arrival counter: the plain apex launch
      hipLaunchKernelGGL(k_apex_records, dim3(ffx_cdiv(info->n_tris, 256)), dim3(256), 0, s, recs, info->n_tris, ak, out, (uint32_t)stride, (uint32_t *)cache, cap_stray);
    } else {
      // (the render kernels these launches will run beside: material rows -> seven waves of 72 VGPRs per SIMD, Lambert -> eight of 64)
      EnvBuild eb;
      const int env_mask = env_build(bb.g, env_enabled(sd, info), info->n_tris, eb);
      bb.env_mask = env_mask;
      ffx_bins_launch(recs, info->n_tris, bb, out, ak.o, ak.on, (uint32_t)stride, (uint32_t *)cache, cap_stray, s, sd && sd->mat_stride != FFX_MAT_STRIDE,
                      info->off_gn ? (uint32_t *)((char *)bvh + info->off_gn) : nullptr, clear_enabled(sd, info), env_mask ? &eb : nullptr);
    }
  } else if (cache)
    hipLaunchKernelGGL(k_cache_reset, dim3(1), dim3(1), 0, s, (uint32_t *)cache, cap_stray);
  *arecs = out;
  *astride = (uint32_t)stride;
  return 1;
}

// default 128 workgroups per XCD per round of the blocked interleave; 256 for films above 512 k pixels: with four times the tiles the
// load still balances at the coarser grain and each L2 walks a smaller part of a larger scene (tools/xcdsweep_colon.sh, config 5:
// 64 / 128 / 256 / 512 / 1024 / 2048 = 90.1 / 90.2 / 89.7 / 88.2 / 84.2 / 67.7 renders/s at 181 / 128 / 80 / 64 / 56 / 54 MB of raw
// FETCH_SIZE per launch; the vocal fold at 512x512 loses 1 % at 256)
static int xcd_mode(long pixels = 0) {
  const char *e = getenv("FFX_XCD_REMAP");
  int m = e ? atoi(e) : (pixels > 2L * 512 * 512 ? 256 : 1024);
  return m < 0 ? 0 : m;
}

static inline uint32_t seed_key_of(uint32_t seed) { return hash32(seed + 0x9e3779b9U); }

static int check_info(const ffx_bvh_info *info, const char *what) {
  if (info->n_tris < 1 || info->n_nodes < 1 || info->max_depth < 1 || info->max_depth > FFX_STACK_DEPTH) {
    ffx_set_error("%s: bad bvh info (n_tris %d, n_nodes %d, max_depth %d)", what, info->n_tris, info->n_nodes, info->max_depth);
    return 0;
  }
  return 1;
}
static inline size_t stack_bytes(const ffx_bvh_info *info) {
  int depth = info->max_depth < 8 ? 8 : info->max_depth;
  return (size_t)depth * TR_BLOCK * sizeof(int);
}

#ifdef FFX_STATS
#endif
#ifdef FFX_BINCHECK
extern "C" int ffx_debug_bincheck(unsigned long long *out24, int reset) {
  if (hipMemcpyFromSymbol(out24, HIP_SYMBOL(g_ffx_chk), sizeof(unsigned long long) * 24) != hipSuccess) return -1;
  if (reset) {
    unsigned long long z[24] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_ffx_chk), z, sizeof(z)) != hipSuccess) return -1;
  }
  return 0;
}
#endif
#ifdef FFX_TIMERS
extern "C" int ffx_debug_timers(unsigned long long *out32, int reset) {
  if (hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_ffx_tim), sizeof(unsigned long long) * 32) != hipSuccess) return -1;
  if (reset) {
    unsigned long long z[32] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_ffx_tim), z, sizeof(z)) != hipSuccess) return -1;
  }
  return 0;
}
#endif
#ifdef FFX_STATS
extern "C" int ffx_debug_stats(unsigned long long *out16, int reset) { // the tree walks' counters [0, 32); resets all 48
  if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_ffx_stats), sizeof(unsigned long long) * 32) != hipSuccess) return -1;
  if (reset) {
    unsigned long long z[48] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_ffx_stats), z, sizeof(z)) != hipSuccess) return -1;
  }
  return 0;
}
extern "C" int ffx_debug_stats48(unsigned long long *out48, int reset) { // ... with the tile bins' counters [32, 48)
  if (hipMemcpyFromSymbol(out48, HIP_SYMBOL(g_ffx_stats), sizeof(unsigned long long) * 48) != hipSuccess) return -1;
  if (reset) {
    unsigned long long z[48] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_ffx_stats), z, sizeof(z)) != hipSuccess) return -1;
  }
  return 0;
}
#endif
extern "C" {

int ffx_trace_primary(const void *bvh, const ffx_bvh_info *info, const ffx_camera *cam, int spp, int jitter, uint32_t seed, float *t_out,
                      int32_t *shape_out, int32_t *prim_out, ffx_stream s) {
  if (!bvh || !info || !cam || !t_out || spp < 1) FFX_FAIL(FFX_ERR_ARG, "trace_primary: bad argument");
  if (!check_info(info, "trace_primary")) return FFX_ERR_ARG;
  CamK k;
  if (!cam_prepare(cam, k)) FFX_FAIL(FFX_ERR_ARG, "trace_primary: bad camera");
  long total = (long)k.W * k.H * spp;
  if (total >= (1L << 32)) FFX_FAIL(FFX_ERR_UNSUPPORTED, "trace_primary: more than 2^32 samples");
  const BvhNode *nodes = (const BvhNode *)((const char *)bvh + info->off_nodes);
  const TriRec *recs = (const TriRec *)((const char *)bvh + info->off_recs);
  if (use_packet()) {
    // pixels per wave: 64 / spp when spp divides 64 (a compact bw x bh block), else one pixel per wave
    int ppw_log2 = 0;
    if (spp < 64 && 64 % spp == 0)
      while ((spp << (ppw_log2 + 1)) <= 64) ++ppw_log2;
    {
      // ... but never more than 16 pixels (1 spp) / 8 pixels per wave: a smaller block leaves lanes idle, yet it makes
      // more and thinner packets — fewer exact tests per walk (24 per walk with 8x8-pixel packets) and enough waves to
      // hide the walk's latency (4096 waves of 64 pixels do not fill the GPU).  Measured at 512^2 (tools/k7time.py):
      // 1 / 2 / 4 spp 0.118 / 0.105 / 0.095 ms with full waves, 0.071 / 0.056 / 0.057 ms capped; >= 8 spp unchanged.
      // FFX_K7_PPW_LOG2 overrides the cap (experiment knob).
      static const int env_cap = getenv("FFX_K7_PPW_LOG2") ? atoi(getenv("FFX_K7_PPW_LOG2")) : -1;
      // (tools/lowspp.py, 512^2: 4 / 8 / 16 / 32 / 64 pixels per wave at 1 spp = 0.174 / 0.160 / 0.169 / 0.192 / 0.239 ms; at 4 spp 0.193 / 0.175 / 0.185)
      const int cap = env_cap >= 0 ? env_cap : 3;
      if (ppw_log2 > cap) ppw_log2 = cap;
    }
    const int bw_log2 = (ppw_log2 + 1) / 2, bh_log2 = ppw_log2 / 2;
    const int blocks_x = ffx_cdiv(k.W, 1 << bw_log2), n_blocks = blocks_x * ffx_cdiv(k.H, 1 << bh_log2);
    const int wpb = packet_waves();
    const TriApex *arecs;
    uint32_t astride;
    // the pre-pass of a camera-only scene: apex records and (round 4) the camera's tile bins; with FFX_RENDER_APEX_READY in `jitter` the
    // camera's area already holds both — a render of this pose from this camera wrote them — and nothing is launched in front of the kernel
    ffx_scene_desc sdt;
    memset(&sdt, 0, sizeof sdt);
    sdt.cam = *cam;
    if (!launch_apex(bvh, info, cam->to_world, &sdt, &arecs, &astride, (hipStream_t)s, nullptr, 0, jitter & FFX_RENDER_APEX_READY)) return FFX_ERR_ARG;
    BinsK bk;
    bins_k(bvh, info, &sdt, bk);
    const int jit = jitter & 1;
    const WideScene ws = wide_scene(bvh, info);
    if (use_wide(info))
      hipLaunchKernelGGL(k_trace_primary_pk<true>, dim3(ffx_cdiv(n_blocks, wpb)), dim3(64 * wpb), 0, (hipStream_t)s, k, nodes, recs, arecs, ws, spp, jit,
                         seed_key_of(seed), bw_log2, bh_log2, blocks_x, n_blocks, t_out, shape_out, prim_out, bk);
    else
      hipLaunchKernelGGL(k_trace_primary_pk<false>, dim3(ffx_cdiv(n_blocks, wpb)), dim3(64 * wpb), 0, (hipStream_t)s, k, nodes, recs, arecs, ws, spp, jit,
                         seed_key_of(seed), bw_log2, bh_log2, blocks_x, n_blocks, t_out, shape_out, prim_out, bk);
    FFX_CHECK_LAUNCH("trace_primary");
    return FFX_OK;
  }
  hipLaunchKernelGGL(k_trace_primary, dim3(ffx_cdiv(total, TR_BLOCK)), dim3(TR_BLOCK), stack_bytes(info), (hipStream_t)s, k, nodes, recs, spp, jitter & 1,
                     seed_key_of(seed), total, t_out, shape_out, prim_out);
  FFX_CHECK_LAUNCH("trace_primary");
  return FFX_OK;
}

int ffx_trace_rays(const void *bvh, const ffx_bvh_info *info, const float *origins, const float *dirs, int n, float tmax, float *t_out,
                   int32_t *shape_out, int32_t *prim_out, ffx_stream s) {
  if (n == 0) return FFX_OK;
  if (!bvh || !info || !origins || !dirs || !t_out || n < 0) FFX_FAIL(FFX_ERR_ARG, "trace_rays: bad argument");
  if (!check_info(info, "trace_rays")) return FFX_ERR_ARG;
  if (n == 0) return FFX_OK;
  const BvhNode *nodes = (const BvhNode *)((const char *)bvh + info->off_nodes);
  const TriRec *recs = (const TriRec *)((const char *)bvh + info->off_recs);
  hipLaunchKernelGGL(k_trace_rays, dim3(ffx_cdiv(n, TR_BLOCK)), dim3(TR_BLOCK), stack_bytes(info), (hipStream_t)s, nodes, recs, origins, dirs, n, tmax,
                     t_out, shape_out, prim_out);
  FFX_CHECK_LAUNCH("trace_rays");
  return FFX_OK;
}

static int render_fwd_impl(const void *bvh, const ffx_bvh_info *info, const ffx_scene_desc *sd, const float *shape_albedo, const float *tex, int spp,
                           uint32_t seed, int call_flags, void *img, void *cache, ffx_stream s, const float *adj_gimg = nullptr, float *adj_gtex = nullptr,
                           float *adj_dot = nullptr, void *rf_scratch = nullptr, void *rf_cache = nullptr) {
  const int img_fp16 = call_flags & (FFX_RENDER_FP16 | FFX_RENDER_SPARSE_ADJOINT); // what the kernels see; the other bits steer the pre-pass
  if (!bvh || !info || !sd || (!shape_albedo && sd->n_mat_h <= 0) || !img || spp < 1) FFX_FAIL(FFX_ERR_ARG, "render_fwd: bad argument");
  if (sd->proj.enabled && !tex) FFX_FAIL(FFX_ERR_ARG, "render_fwd: projector enabled but tex is NULL");
  if (!check_info(info, "render_fwd")) return FFX_ERR_ARG;
  if ((sd->rfilter != FFX_RFILTER_BOX) != (rf_scratch != nullptr))
    FFX_FAIL(FFX_ERR_UNSUPPORTED, rf_scratch ? "render_fwd_filtered: rfilter must be FFX_RFILTER_GAUSSIAN"
                                             : "render_fwd: the scene's reconstruction filter is not the box (use ffx_render_fwd_filtered)");
  if (rf_scratch && (!use_packet() || !use_wide(info))) FFX_FAIL(FFX_ERR_UNSUPPORTED, "render_fwd_filtered: only the default (wide packet) kernels carry the filter");
  ShadeK c;
  if (!shade_prepare(sd, c)) FFX_FAIL(rf_scratch ? FFX_ERR_UNSUPPORTED : FFX_ERR_ARG, "render_fwd: bad scene description%s", rf_scratch ? " (gaussian filter: stddev <= 0.5)" : "");
  c.mats = shape_albedo;
  const bool mat = c.mat_stride == FFX_MAT_STRIDE;
  if (mat && !c.mat_inline && ((uintptr_t)shape_albedo & 15) != 0) FFX_FAIL(FFX_ERR_ARG, "render_fwd: material rows must be 16-byte aligned");
  if ((long)c.cam.W * c.cam.H * spp >= (1L << 32)) FFX_FAIL(FFX_ERR_UNSUPPORTED, "render_fwd: more than 2^32 samples");
  const BvhNode *nodes = (const BvhNode *)((const char *)bvh + info->off_nodes);
  const TriRec *recs = (const TriRec *)((const char *)bvh + info->off_recs);
  const float4 *nrec = info->off_nrec ? (const float4 *)((const char *)bvh + info->off_nrec) : nullptr; // vertex normals per slot (ffx_smooth)
  const float4 *gn = info->off_gn ? (const float4 *)((const char *)bvh + info->off_gn) : nullptr;         // unit geometric normals per slot
  if (cache && sd->n_base_tex > 0) FFX_FAIL(FFX_ERR_UNSUPPORTED, "render_fwd_cache: textured base colours (the footprint folds one base colour per shape): use ffx_render_bwd");
  if (cache && sd->proj.enabled && (sd->proj.tex_w > 4094 || sd->proj.tex_h > 4094 || sd->n_shapes > 255))
    FFX_FAIL(FFX_ERR_UNSUPPORTED, "render_fwd_cache: texture larger than 4094^2 or more than 255 shapes");
  if (adj_gtex && sd->n_base_tex > 0) FFX_FAIL(FFX_ERR_UNSUPPORTED, "render_fwd_adjoint: textured base colours (the footprint folds one base colour per shape): use ffx_render_bwd");
  if ((use_packet() || cache || adj_gtex) && !gn) FFX_FAIL(FFX_ERR_ARG, "render_fwd: blob without per-slot normals (built by another library version?)");
  if (use_packet() || cache || adj_gtex) { // the per-sample cache / the fused adjoint are the packet kernels' 
    const int tb = tile_block_log2((long)c.cam.W * c.cam.H, cache != nullptr || adj_gtex != nullptr || rf_cache != nullptr);
    int ptx = ffx_cdiv(c.cam.W, 2), pty = ffx_cdiv(c.cam.H, 2);
    int pn = (ffx_cdiv(ptx, 1 << tb) * ffx_cdiv(pty, 1 << tb)) << (2 * tb); // whole blocks; tiles outside the image are skipped
    ptx |= tb << 24;
    const int wpb = packet_waves();
    const int ppw = pixels_per_wave(cache != nullptr || adj_gtex != nullptr || rf_cache != nullptr, (long)c.cam.W * c.cam.H);
    int pgrid = ((ffx_cdiv((long)pn * (4 / ppw), wpb) + 7) / 8) * 8; // multiple of 8 so the XCD remap is a bijection onto [0, grid)
    const TriApex *arecs;
    uint32_t astride;
    // (capacity of the cache's arena: single-sample records of the box film's footprint cache / 64-sample blocks of the filtered film's record cache)
    uint32_t cap_stray = cache ? (uint32_t)cache_stray_capacity(c.cam.W, c.cam.H, spp) : (rf_cache ? (uint32_t)rfc_cap_blocks((size_t)c.cam.W * c.cam.H, (size_t)spp, rfc_all(sd)) : 0u);
    if (rf_cache)
      if (const char *ce = getenv("FFX_RFC_CAP")) { // (test knob, include/ffx.h: fewer blocks than the cache has room for — the overflow path)
        const long cv = atol(ce);
        if (cv >= 0 && (unsigned long)cv < cap_stray) cap_stray = (uint32_t)cv;
      }
    if (!launch_apex(bvh, info, sd->cam.to_world, sd, &arecs, &astride, (hipStream_t)s, cache ? cache : rf_cache, cap_stray, call_flags)) return FFX_ERR_ARG;
    const WideScene ws = wide_scene(bvh, info);
    bins_k(bvh, info, sd, c.bins);
    // offsets of the cache areas in units of 128 bytes (both are multiples of 128; a 1024^2 x 256-spp cache is 160 MB)
    const uint32_t foot_off = (uint32_t)(cache_off_foot((size_t)c.cam.W * c.cam.H) >> 7), arena_off = (uint32_t)(cache_off_arena((size_t)c.cam.W * c.cam.H) >> 7);
    const uint32_t foot_b_off = (uint32_t)(cache_off_foot_b((size_t)c.cam.W * c.cam.H, cache_stray_capacity(c.cam.W, c.cam.H, spp)) >> 7);
#define FFX_LAUNCH_FWD_(WIDE_, MAT_, ADJ_)                                                                                                               \
  hipLaunchKernelGGL((k_render_fwd_pk<1, WIDE_, MAT_, ADJ_>), dim3(pgrid), dim3(64 * wpb), dummy_lds(), (hipStream_t)s, c, nodes, recs, arecs, astride, ws, \
                     shape_albedo, tex, spp, seed_key_of(seed), ptx, pn, xcd_mode((long)c.cam.W * c.cam.H), img_fp16, img, (char *)cache, ppw, 1.0f / (float)spp, foot_off,     \
                     arena_off, foot_b_off, nrec, gn, cap_stray, adj_gimg, adj_gtex, adj_dot)
#define FFX_LAUNCH_FWD(WIDE_, MAT_) do { if (adj_gtex) FFX_LAUNCH_FWD_(WIDE_, MAT_, true); else FFX_LAUNCH_FWD_(WIDE_, MAT_, false); } while (0)
    const int matm = !mat ? 0 : (c.n_base_tex > 0 ? 2 : 1); // (textured base colours: their own instantiation — the default kernels pay nothing)
    if (rf_scratch && adj_gtex) { // ... with the adjoint of a loss that is linear in the image folded in (ffx_render_fwd_adjoint_filtered)
      if (sd->proj.tex_channels != 1 || matm == 2) FFX_FAIL(FFX_ERR_UNSUPPORTED, "render_fwd_adjoint_filtered: 1-channel projector textures without textured base colours (use ffx_render_fwd_filtered + ffx_render_bwd_filtered)");
      // the weight every pixel will receive (the jitter alone decides it) -> G = gimg / weight behind the partial sums; then the render, whose
      // footprints take every sample's own gradient from G; then the image
      const int n_pix = c.cam.W * c.cam.H;
      float *part = (float *)rf_scratch;
      float4 *G = (float4 *)(part + (size_t)n_pix * 100);
      hipLaunchKernelGGL(k_rf_weights, dim3(n_pix), dim3(64), 0, (hipStream_t)s, c.rf, n_pix, spp, seed_key_of(seed), part);
      FFX_CHECK_LAUNCH("render_fwd_adjoint_filtered/weights");
      hipLaunchKernelGGL(k_rf_gather, dim3(ffx_cdiv(c.cam.W, 64), ffx_cdiv(c.cam.H, FFX_RFG_WAVES)), dim3(64 * FFX_RFG_WAVES), 0, (hipStream_t)s, (const float4 *)part, c.cam.W,
                         c.cam.H, 0, (void *)nullptr, adj_gimg, G);
      FFX_CHECK_LAUNCH("render_fwd_adjoint_filtered/gather G");
#define FFX_LAUNCH_RFA(MAT_)                                                                                                                             \
  hipLaunchKernelGGL((k_render_fwd_pk<1, true, MAT_, true, true>), dim3(pgrid), dim3(64 * wpb), dummy_lds(), (hipStream_t)s, c, nodes, recs, arecs, astride, ws, \
                     shape_albedo, tex, spp, seed_key_of(seed), ptx, pn, xcd_mode((long)c.cam.W * c.cam.H), img_fp16, img, (char *)rf_scratch, ppw,          \
                     1.0f / (float)spp, foot_off, arena_off, foot_b_off, nrec, gn, cap_stray, (const float *)G, adj_gtex, (float *)nullptr)
      if (matm == 1) FFX_LAUNCH_RFA(1); else FFX_LAUNCH_RFA(0);
#undef FFX_LAUNCH_RFA
      FFX_CHECK_LAUNCH("render_fwd_adjoint_filtered");
      hipLaunchKernelGGL(k_rf_gather, dim3(ffx_cdiv(c.cam.W, 64), ffx_cdiv(c.cam.H, FFX_RFG_WAVES)), dim3(64 * FFX_RFG_WAVES), 0, (hipStream_t)s, (const float4 *)rf_scratch, c.cam.W,
                         c.cam.H, img_fp16 & 1, img, (const float *)nullptr, (float4 *)nullptr);
      FFX_CHECK_LAUNCH("render_fwd_adjoint_filtered/gather");
      return FFX_OK;
    }
    if (rf_scratch && rf_cache) { // ... that also stores its adjoint's per-sample records (ffx_render_fwd_cache_filtered; rfc_off_* above)
      if (matm == 2) FFX_FAIL(FFX_ERR_UNSUPPORTED, "render_fwd_cache_filtered: textured base colours (the records carry one base colour per shape): use ffx_render_bwd_filtered");
      if (sd->proj.enabled && (sd->proj.tex_w > 4094 || sd->proj.tex_h > 4094 || sd->n_shapes > 255)) FFX_FAIL(FFX_ERR_UNSUPPORTED, "render_fwd_cache_filtered: texture larger than 4094^2 or more than 255 shapes");
      if (spp > 1024) FFX_FAIL(FFX_ERR_UNSUPPORTED, "render_fwd_cache_filtered: more than 1024 samples per pixel (16 passes)");
      const size_t n_pix = (size_t)c.cam.W * c.cam.H;
      const uint32_t recs_off = (uint32_t)(rfc_off_recs(n_pix) >> 7), facb_off = (uint32_t)(rfc_off_facb(n_pix, (size_t)spp, rfc_all(sd)) >> 7);
#define FFX_LAUNCH_RFC(MAT_)                                                                                                                              \
  hipLaunchKernelGGL((k_render_fwd_pk<1, true, MAT_, false, true, true>), dim3(pgrid), dim3(64 * wpb), dummy_lds(), (hipStream_t)s, c, nodes, recs, arecs, astride, ws, \
                     shape_albedo, tex, spp, seed_key_of(seed), ptx, pn, xcd_mode((long)c.cam.W * c.cam.H), img_fp16, img, (char *)rf_scratch, ppw,       \
                     1.0f / (float)spp, recs_off, arena_off, facb_off, nrec, gn, cap_stray, (const float *)nullptr, (float *)rf_cache, (float *)nullptr)
      if (matm == 1) FFX_LAUNCH_RFC(1); else FFX_LAUNCH_RFC(0);
#undef FFX_LAUNCH_RFC
      FFX_CHECK_LAUNCH("render_fwd_cache_filtered");
      hipLaunchKernelGGL(k_rf_gather, dim3(ffx_cdiv(c.cam.W, 64), ffx_cdiv(c.cam.H, FFX_RFG_WAVES)), dim3(64 * FFX_RFG_WAVES), 0, (hipStream_t)s, (const float4 *)rf_scratch, c.cam.W, c.cam.H,
                         img_fp16 & 1, img, (const float *)nullptr, (float4 *)nullptr, (float *)((char *)rf_cache + rfc_off_wsum(n_pix)));
      FFX_CHECK_LAUNCH("render_fwd_cache_filtered/gather");
      return FFX_OK;
    }
    if (rf_scratch && spp <= 32 && lowspp_blocks()) { // ... below 33 samples per pixel: several pixels per wave, the same scratch contents (k_render_fwd_blk<..., RF>)
      int slots = 8; // (at least 8 sample slots per pixel: rf_fold_blk reads them in groups of four, two pixels per round)
      while (slots < spp) slots <<= 1;
      int ppw_log2 = 0;
      while ((slots << (ppw_log2 + 1)) <= 64) ++ppw_log2; // 8 / 4 / 2 pixels per wave at <= 8 / 16 / 32 spp
      const int bw_log2 = (ppw_log2 + 1) / 2, bh_log2 = ppw_log2 / 2;
      const int blocks_x = ffx_cdiv(c.cam.W, 1 << bw_log2), n_blocks = 64 * ffx_cdiv(blocks_x, 8) * ffx_cdiv(ffx_cdiv(c.cam.H, 1 << bh_log2), 8); // (whole 8 x 8 patches)
#define FFX_LAUNCH_BLKF(MAT_)                                                                                                                            \
  hipLaunchKernelGGL((k_render_fwd_blk<true, MAT_, true>), dim3(ffx_cdiv(n_blocks, wpb)), dim3(64 * wpb), 0, (hipStream_t)s, c, nodes, recs, arecs, astride, ws, \
                     shape_albedo, tex, spp, seed_key_of(seed), bw_log2, bh_log2, blocks_x, n_blocks, img_fp16 & 1, rf_scratch, 1.0f / (float)spp, nrec, gn)
      if (matm == 2) FFX_LAUNCH_BLKF(2); else if (matm == 1) FFX_LAUNCH_BLKF(1); else FFX_LAUNCH_BLKF(0);
#undef FFX_LAUNCH_BLKF
      FFX_CHECK_LAUNCH("render_fwd_filtered (pixel blocks)");
      hipLaunchKernelGGL(k_rf_gather, dim3(ffx_cdiv(c.cam.W, 64), ffx_cdiv(c.cam.H, FFX_RFG_WAVES)), dim3(64 * FFX_RFG_WAVES), 0, (hipStream_t)s, (const float4 *)rf_scratch, c.cam.W, c.cam.H,
                         img_fp16 & 1, img, (const float *)nullptr, (float4 *)nullptr);
      FFX_CHECK_LAUNCH("render_fwd_filtered/gather");
      return FFX_OK;
    }
    if (rf_scratch) { // the filtered render: the kernel leaves every pixel's 25 x 4 outgoing sums in the scratch area, the gather forms the image
#define FFX_LAUNCH_RF(MAT_)                                                                                                                               \
  hipLaunchKernelGGL((k_render_fwd_pk<1, true, MAT_, false, true>), dim3(pgrid), dim3(64 * wpb), dummy_lds(), (hipStream_t)s, c, nodes, recs, arecs, astride, ws, \
                     shape_albedo, tex, spp, seed_key_of(seed), ptx, pn, xcd_mode((long)c.cam.W * c.cam.H), img_fp16 & 1, img, (char *)rf_scratch, ppw,       \
                     1.0f / (float)spp, foot_off, arena_off, foot_b_off, nrec, gn, cap_stray, adj_gimg, adj_gtex, adj_dot)
      if (matm == 2) FFX_LAUNCH_RF(2); else if (matm == 1) FFX_LAUNCH_RF(1); else FFX_LAUNCH_RF(0);
#undef FFX_LAUNCH_RF
      FFX_CHECK_LAUNCH("render_fwd_filtered");
      hipLaunchKernelGGL(k_rf_gather, dim3(ffx_cdiv(c.cam.W, 64), ffx_cdiv(c.cam.H, FFX_RFG_WAVES)), dim3(64 * FFX_RFG_WAVES), 0, (hipStream_t)s, (const float4 *)rf_scratch, c.cam.W, c.cam.H,
                         img_fp16 & 1, img, (const float *)nullptr, (float4 *)nullptr);
      FFX_CHECK_LAUNCH("render_fwd_filtered/gather");
      return FFX_OK;
    }
    if (!cache && !adj_gtex && spp <= 32 && lowspp_blocks()) {
      // fewer than 64 samples per pixel: several pixels per wave (k_render_fwd_blk).  Sample slots per pixel = the next power of two >= spp; the
      // block never exceeds 8 pixels (K7's reason, ffx_trace_primary: thinner packets, fewer exact tests per walk, more waves)
      int slots = 1;
      while (slots < spp) slots <<= 1;
      int ppw_log2 = 0;
      while ((slots << (ppw_log2 + 1)) <= 64) ++ppw_log2;
      static const int env_cap = getenv("FFX_RENDER_BLK_LOG2") ? atoi(getenv("FFX_RENDER_BLK_LOG2")) : -1; // (experiment knob: pixels per wave, log2)
      const int cap = env_cap >= 0 ? env_cap : (spp == 1 ? 4 : 3);
      if (ppw_log2 > cap) ppw_log2 = cap;
      const int bw_log2 = (ppw_log2 + 1) / 2, bh_log2 = ppw_log2 / 2;
      const int blocks_x = ffx_cdiv(c.cam.W, 1 << bw_log2), n_blocks = 64 * ffx_cdiv(blocks_x, 8) * ffx_cdiv(ffx_cdiv(c.cam.H, 1 << bh_log2), 8); // (whole 8 x 8 patches)
#define FFX_LAUNCH_BLK(WIDE_, MAT_)                                                                                                                     \
  hipLaunchKernelGGL((k_render_fwd_blk<WIDE_, MAT_>), dim3(ffx_cdiv(n_blocks, wpb)), dim3(64 * wpb), 0, (hipStream_t)s, c, nodes, recs, arecs, astride, ws, \
                     shape_albedo, tex, spp, seed_key_of(seed), bw_log2, bh_log2, blocks_x, n_blocks, img_fp16, img, 1.0f / (float)spp, nrec, gn)
      if (use_wide(info)) { if (matm == 2) FFX_LAUNCH_BLK(true, 2); else if (matm == 1) FFX_LAUNCH_BLK(true, 1); else FFX_LAUNCH_BLK(true, 0); }
      else { if (matm == 2) FFX_LAUNCH_BLK(false, 2); else if (matm == 1) FFX_LAUNCH_BLK(false, 1); else FFX_LAUNCH_BLK(false, 0); }
#undef FFX_LAUNCH_BLK
      FFX_CHECK_LAUNCH("render_fwd (pixel blocks)");
      return FFX_OK;
    }
    if (use_wide(info)) { if (matm == 2) FFX_LAUNCH_FWD_(true, 2, false); else if (matm == 1) FFX_LAUNCH_FWD(true, 1); else FFX_LAUNCH_FWD(true, 0); }
    else { if (matm == 2) FFX_LAUNCH_FWD_(false, 2, false); else if (matm == 1) FFX_LAUNCH_FWD(false, 1); else FFX_LAUNCH_FWD(false, 0); }
#undef FFX_LAUNCH_FWD
#undef FFX_LAUNCH_FWD_
    FFX_CHECK_LAUNCH("render_fwd");
    return FFX_OK;
  }
  int tiles_x = ffx_cdiv(c.cam.W, 8), tiles_y = ffx_cdiv(c.cam.H, 8);
  int n_tiles = tiles_x * tiles_y;
  int grid = ((n_tiles + 7) / 8) * 8;
  hipLaunchKernelGGL(k_render_fwd, dim3(grid), dim3(TR_BLOCK), stack_bytes(info), (hipStream_t)s, c, nodes, recs, nrec, shape_albedo, tex, spp,
                     seed_key_of(seed), tiles_x, n_tiles, xcd_mode(), img_fp16, img);
  FFX_CHECK_LAUNCH("render_fwd");
  return FFX_OK;
}

int ffx_render_fwd(const void *bvh, const ffx_bvh_info *info, const ffx_scene_desc *sd, const float *shape_albedo, const float *tex, int spp,
                   uint32_t seed, int img_fp16, void *img, ffx_stream s) {
  return render_fwd_impl(bvh, info, sd, shape_albedo, tex, spp, seed, img_fp16 & (FFX_RENDER_FP16 | FFX_RENDER_APEX_READY), img, nullptr, s);
}

size_t ffx_render_cache_bytes(int width, int height, int spp) {
  if (width < 1 || height < 1 || spp < 1) return 0;
  return cache_off_arena((size_t)width * height) + sizeof(CacheStray) * cache_stray_capacity(width, height, spp);
}

size_t ffx_render_cache_bytes_sd(const ffx_scene_desc *sd, int spp) {
  if (!sd || sd->cam.width < 1 || sd->cam.height < 1 || spp < 1) return 0;
  if (sd->rfilter != FFX_RFILTER_BOX) return rfc_bytes((size_t)sd->cam.width * sd->cam.height, (size_t)spp, sd->mat_stride == FFX_MAT_STRIDE, rfc_all(sd)); // the filtered film's cache
  if (sd->mat_stride != FFX_MAT_STRIDE) return ffx_render_cache_bytes(sd->cam.width, sd->cam.height, spp);
  const size_t n_pix = (size_t)sd->cam.width * sd->cam.height;
  return cache_off_foot_b(n_pix, cache_stray_capacity(sd->cam.width, sd->cam.height, spp)) + sizeof(CacheFoot) * n_pix;
}

int ffx_render_fwd_cache(const void *bvh, const ffx_bvh_info *info, const ffx_scene_desc *sd, const float *shape_albedo, const float *tex, int spp,
                         uint32_t seed, int img_fp16, void *img, void *cache, ffx_stream s) {
  if (!cache) FFX_FAIL(FFX_ERR_ARG, "render_fwd_cache: cache is NULL");
  if (((uintptr_t)cache & 15) != 0) FFX_FAIL(FFX_ERR_ARG, "render_fwd_cache: cache must be 16-byte aligned");
  return render_fwd_impl(bvh, info, sd, shape_albedo, tex, spp, seed, img_fp16 & 31, img, cache, s);
}

int ffx_render_cache_status(const void *cache, uint32_t *out3, ffx_stream s) {
  if (!cache || !out3) FFX_FAIL(FFX_ERR_ARG, "render_cache_status: bad argument");
  CacheHdr h;
  if (hipMemcpyAsync(&h, cache, sizeof h, hipMemcpyDeviceToHost, (hipStream_t)s) != hipSuccess || hipStreamSynchronize((hipStream_t)s) != hipSuccess)
    FFX_FAIL(FFX_ERR_LAUNCH, "render_cache_status: reading the cache header failed");
  out3[0] = h.n_stray; out3[1] = h.cap_stray; out3[2] = h.dropped;
  for (int i = 1; i <= 8; ++i) out3[0] += h.pad[i]; // (a filtered film's cache counts its blocks in eight sub-arenas; zero in a box film's)
  return FFX_OK;
}

int ffx_render_fwd_adjoint(const void *bvh, const ffx_bvh_info *info, const ffx_scene_desc *sd, const float *shape_albedo, const float *tex, int spp,
                           uint32_t seed, int img_fp16, void *img, const float *gimg, float *gtex, float *dot_out, ffx_stream s) {
  if (!gimg || !gtex) FFX_FAIL(FFX_ERR_ARG, "render_fwd_adjoint: gimg / gtex is NULL");
  if (sd && !sd->proj.enabled) FFX_FAIL(FFX_ERR_ARG, "render_fwd_adjoint: the scene has no projector (nothing to differentiate)");
  return render_fwd_impl(bvh, info, sd, shape_albedo, tex, spp, seed, img_fp16 & (FFX_RENDER_FP16 | FFX_RENDER_SPARSE_ADJOINT | FFX_RENDER_APEX_READY), img, nullptr, s, gimg,
                         gtex, dot_out);
}

int ffx_apex_prepare(void *bvh, const ffx_bvh_info *info, const ffx_scene_desc *sd, ffx_stream s) {
  if (!bvh || !info || !sd) FFX_FAIL(FFX_ERR_ARG, "apex_prepare: bad argument");
  if (!check_info(info, "apex_prepare")) return FFX_ERR_ARG;
  const TriApex *arecs;
  uint32_t astride;
  if (!launch_apex(bvh, info, sd->cam.to_world, sd, &arecs, &astride, (hipStream_t)s)) return FFX_ERR_ARG;
  FFX_CHECK_LAUNCH("apex_prepare");
  return FFX_OK;
}

// at most 256 slots (one per 8x8-pixel block, folded modulo 256): enough to take the contention out of the atomics (a 512x512 film
// adds 16 values to each) and few enough for ONE load per thread of the launch that sums them (ffx_pattern_bwd's block 0)
size_t ffx_render_dot_slots(int width, int height) {
  if (width < 1 || height < 1) return 0;
  const size_t b = (size_t)ffx_cdiv(width, 8) * (size_t)ffx_cdiv(height, 8);
  return b < 256 ? b : 256;
}

static int render_bwd_cached_impl(const ffx_scene_desc *sd, const float *shape_albedo, const void *cache, int spp, const float *gimg, float *gtex, const void *img,
                                  int img_fp16, float *dot_out, const float *l1_target, float l1_weight, ffx_stream s);
int ffx_render_bwd_cached(const ffx_scene_desc *sd, const float *shape_albedo, const void *cache, int spp, const float *gimg, float *gtex, const void *img,
                          int img_fp16, float *dot_out, ffx_stream s) {
  if (!gimg) FFX_FAIL(FFX_ERR_ARG, "render_bwd_cached: bad argument");
  return render_bwd_cached_impl(sd, shape_albedo, cache, spp, gimg, gtex, img, img_fp16, dot_out, nullptr, 0.f, s);
}
// K9 under an L1 loss against a target image (include/ffx.h): the loss launch (ffx_l1_value_grad) and its gradient image are folded into the scatter
int ffx_render_bwd_cached_l1(const ffx_scene_desc *sd, const float *shape_albedo, const void *cache, int spp, const float *img, const float *target, float weight,
                             float *gtex, float *loss_slots, ffx_stream s) {
  if (!img || !target || !loss_slots) FFX_FAIL(FFX_ERR_ARG, "render_bwd_cached_l1: bad argument");
  if (sd && (!sd->proj.enabled || sd->proj.tex_channels != 1)) FFX_FAIL(FFX_ERR_UNSUPPORTED, "render_bwd_cached_l1: a projector with a one-channel texture (else ffx_l1_value_grad + ffx_render_bwd_cached)");
  const char *k9e = getenv("FFX_K9_BLOCK");
  if (k9e && atoi(k9e) == 8) FFX_FAIL(FFX_ERR_UNSUPPORTED, "render_bwd_cached_l1: only the 16x16-block kernel carries the loss");
  return render_bwd_cached_impl(sd, shape_albedo, cache, spp, nullptr, gtex, img, 0, loss_slots, target, weight, s);
}
static int render_bwd_cached_impl(const ffx_scene_desc *sd, const float *shape_albedo, const void *cache, int spp, const float *gimg, float *gtex, const void *img,
                                  int img_fp16, float *dot_out, const float *l1_target, float l1_weight, ffx_stream s) {
  if (!sd || (!shape_albedo && sd->n_mat_h <= 0) || !cache || (!gimg && !l1_target) || !gtex || spp < 1 || (dot_out && !img)) FFX_FAIL(FFX_ERR_ARG, "render_bwd_cached: bad argument");
  if (sd->rfilter != FFX_RFILTER_BOX) FFX_FAIL(FFX_ERR_UNSUPPORTED, "render_bwd_cached: the scene's reconstruction filter is not the box (use ffx_render_bwd_filtered)");
  if (!sd->proj.enabled) {
    if (dot_out) FFX_FAIL(FFX_ERR_UNSUPPORTED, "render_bwd_cached: <gimg, img> is accumulated by the footprint kernel, which a scene without projector does not launch");
    return FFX_OK;
  }
  BwdP p;
  memset(&p, 0, sizeof p);
  p.mats = shape_albedo;
  if (sd->n_mat_h > 0) {
    const int ms_ = sd->mat_stride ? sd->mat_stride : 3;
    if (sd->n_mat_h > FFX_MAX_MAT_H || sd->n_mat_h != sd->n_shapes * ms_) FFX_FAIL(FFX_ERR_ARG, "render_bwd_cached: n_mat_h must be n_shapes x stride (<= %d)", FFX_MAX_MAT_H);
    p.mat_inline = 1;
    for (int i = 0; i < sd->n_mat_h; ++i) p.mat_h[i] = sd->mat_h[i];
  }
  p.img = dot_out ? img : nullptr; p.img_fp16 = img_fp16 & 1; p.dot_out = dot_out;
  p.dot_slots = (int)ffx_render_dot_slots(sd->cam.width, sd->cam.height);
  if (l1_target) { // (sign(img - target) * weight / n per element; value weight / n * sum |img - target|: ffx_l1_value_grad's)
    const float n_el = 3.0f * (float)sd->cam.width * (float)sd->cam.height;
    p.l1_tgt = l1_target; p.l1_gs = l1_weight / n_el; p.l1_vs = l1_weight / n_el;
  }
  p.tw = sd->proj.tex_w; p.th = sd->proj.tex_h; p.tc = sd->proj.tex_channels; p.spp = spp;
  if (p.tw < 1 || p.th < 1 || (p.tc != 1 && p.tc != 3) || sd->cam.width < 1 || sd->cam.height < 1) FFX_FAIL(FFX_ERR_ARG, "render_bwd_cached: bad scene description");
  for (int i = 0; i < 3; ++i) p.color[i] = sd->proj.color[i];
  p.inv_spp = 1.0f / (float)spp;
  const long n_pix = (long)sd->cam.width * sd->cam.height;
  p.W = sd->cam.width; p.H = sd->cam.height;
  p.ms = sd->mat_stride ? sd->mat_stride : 3;
  if (p.ms != 3 && p.ms != FFX_MAT_STRIDE) FFX_FAIL(FFX_ERR_ARG, "render_bwd_cached: bad material stride");
  p.off_foot_b = cache_off_foot_b((size_t)n_pix, cache_stray_capacity(sd->cam.width, sd->cam.height, spp));
  const int stray_blocks = ffx_cdiv((long)cache_stray_capacity(sd->cam.width, sd->cam.height, spp), 256);
  const char *k9e = getenv("FFX_K9_BLOCK"); // (8: the round-2 kernel, for A/B; read per call like the other knobs)
  const int k9_block = (k9e && atoi(k9e) == 8) ? 8 : 16;
  if (p.tc == 1 && k9_block == 16) {
    // footprints by 16x16-pixel blocks through an LDS tile; the tail of the grid replays the stray records
    const int blocks_x = ffx_cdiv(p.W, 16), blocks_y = ffx_cdiv(p.H, 16);
    hipLaunchKernelGGL(k_render_bwd_cached_tiled16, dim3(blocks_x * blocks_y + stray_blocks), dim3(256), 0, (hipStream_t)s, p, (const char *)cache, blocks_x,
                       blocks_x * blocks_y, gimg, gtex);
    FFX_CHECK_LAUNCH("render_bwd_cached/tiled16");
    return FFX_OK;
  }
  if (p.tc == 1) {
    const int blocks_x = ffx_cdiv(p.W, 8), blocks_y = ffx_cdiv(p.H, 8);
    hipLaunchKernelGGL(k_render_bwd_cached_tiled, dim3(blocks_x * blocks_y + stray_blocks), dim3(256), 0, (hipStream_t)s, p, (const char *)cache, blocks_x,
                       blocks_x * blocks_y, gimg, gtex);
    FFX_CHECK_LAUNCH("render_bwd_cached/tiled");
    return FFX_OK;
  }
  const int slot_blocks = ffx_cdiv(n_pix, 8);
  hipLaunchKernelGGL(k_render_bwd_cached, dim3(slot_blocks + stray_blocks), dim3(256), 0, (hipStream_t)s, p, (const char *)cache, n_pix, slot_blocks, gimg, gtex);
  FFX_CHECK_LAUNCH("render_bwd_cached");
  return FFX_OK;
}

static int render_bwd_impl(const void *bvh, const ffx_bvh_info *info, const ffx_scene_desc *sd, const float *shape_albedo, int spp, uint32_t seed, int flags,
                           const float *gimg, float *gtex, ffx_stream s, void *rf_scratch, void *det_ws = nullptr, const DetPart *det_part = nullptr) {
  if (!bvh || !info || !sd || (!shape_albedo && sd->n_mat_h <= 0) || !gimg || (!gtex && !det_part) || spp < 1) FFX_FAIL(FFX_ERR_ARG, "render_bwd: bad argument");
  if ((sd->rfilter != FFX_RFILTER_BOX) != (rf_scratch != nullptr))
    FFX_FAIL(FFX_ERR_UNSUPPORTED, rf_scratch ? "render_bwd_filtered: rfilter must be FFX_RFILTER_GAUSSIAN"
                                             : "render_bwd: the scene's reconstruction filter is not the box (use ffx_render_bwd_filtered)");
  if (rf_scratch && (!use_packet() || !use_wide(info))) FFX_FAIL(FFX_ERR_UNSUPPORTED, "render_bwd_filtered: only the default (wide packet) kernels carry the filter");
  if (!sd->proj.enabled) return FFX_OK;
  if (!check_info(info, "render_bwd")) return FFX_ERR_ARG;
  ShadeK c;
  if (!shade_prepare(sd, c)) FFX_FAIL(rf_scratch ? FFX_ERR_UNSUPPORTED : FFX_ERR_ARG, "render_bwd: bad scene description%s", rf_scratch ? " (gaussian filter: stddev <= 0.5)" : "");
  c.mats = shape_albedo;
  const bool mat = c.mat_stride == FFX_MAT_STRIDE;
  if (mat && !c.mat_inline && ((uintptr_t)shape_albedo & 15) != 0) FFX_FAIL(FFX_ERR_ARG, "render_bwd: material rows must be 16-byte aligned");
  if ((long)c.cam.W * c.cam.H * spp >= (1L << 32)) FFX_FAIL(FFX_ERR_UNSUPPORTED, "render_bwd: more than 2^32 samples");
  const BvhNode *nodes = (const BvhNode *)((const char *)bvh + info->off_nodes);
  const TriRec *recs = (const TriRec *)((const char *)bvh + info->off_recs);
  const float4 *nrec = info->off_nrec ? (const float4 *)((const char *)bvh + info->off_nrec) : nullptr;
  const float4 *gn = info->off_gn ? (const float4 *)((const char *)bvh + info->off_gn) : nullptr;
  if (use_packet() && !gn) FFX_FAIL(FFX_ERR_ARG, "render_bwd: blob without per-slot normals (built by another library version?)");
  if (use_packet()) {
    const int tb = tile_block_log2((long)c.cam.W * c.cam.H);
    int ptx = ffx_cdiv(c.cam.W, 2), pty = ffx_cdiv(c.cam.H, 2);
    int pn = (ffx_cdiv(ptx, 1 << tb) * ffx_cdiv(pty, 1 << tb)) << (2 * tb);
    ptx |= tb << 24;
    const int wpb = packet_waves();
    int pgrid = ((ffx_cdiv(pn, wpb) + 7) / 8) * 8;
    const TriApex *arecs;
    uint32_t astride;
    if (!launch_apex(bvh, info, sd->cam.to_world, sd, &arecs, &astride, (hipStream_t)s, nullptr, 0, flags & FFX_RENDER_APEX_READY)) return FFX_ERR_ARG;
    const WideScene ws = wide_scene(bvh, info);
    bins_k(bvh, info, sd, c.bins);
    DetK det;
    memset(&det, 0, sizeof det);
    const int matm = !mat ? 0 : (c.n_base_tex > 0 ? 2 : 1);
    const float *gsrc = gimg; // what the kernel gathers from: gimg, or (filtered film) G = gimg / weight as float4 per pixel
    if (rf_scratch) {
      // 1. the weight every pixel received (jitter only) -> G = gimg / weight behind the partial sums; 2. the re-trace gathers through the filter
      const int n_pix = c.cam.W * c.cam.H;
      float *part = (float *)rf_scratch;
      float4 *G = (float4 *)(part + (size_t)n_pix * 100);
      hipLaunchKernelGGL(k_rf_weights, dim3(n_pix), dim3(64), 0, (hipStream_t)s, c.rf, n_pix, spp, seed_key_of(seed), part);
      FFX_CHECK_LAUNCH("render_bwd_filtered/weights");
      hipLaunchKernelGGL(k_rf_gather, dim3(ffx_cdiv(c.cam.W, 64), ffx_cdiv(c.cam.H, FFX_RFG_WAVES)), dim3(64 * FFX_RFG_WAVES), 0, (hipStream_t)s, (const float4 *)part, c.cam.W, c.cam.H, 0,
                         (void *)nullptr, gimg, G);
      FFX_CHECK_LAUNCH("render_bwd_filtered/gather");
      gsrc = (const float *)G;
    }
    const bool wide = use_wide(info) != 0;
#define FFX_LAUNCH_BWD(WIDE_, MAT_, RF_)                                                                                                                  \
  hipLaunchKernelGGL((k_render_bwd_pk<1, WIDE_, MAT_, RF_>), dim3(pgrid), dim3(64 * wpb), 0, (hipStream_t)s, c, nodes, recs, arecs, astride, ws, shape_albedo, \
                     spp, seed_key_of(seed), ptx, pn, xcd_mode((long)c.cam.W * c.cam.H), gsrc, gtex, nrec, gn, det)
    auto launch = [&]() {
      if (rf_scratch) { if (matm == 2) FFX_LAUNCH_BWD(true, 2, true); else if (matm == 1) FFX_LAUNCH_BWD(true, 1, true); else FFX_LAUNCH_BWD(true, 0, true); }
      else if (wide) { if (matm == 2) FFX_LAUNCH_BWD(true, 2, false); else if (matm == 1) FFX_LAUNCH_BWD(true, 1, false); else FFX_LAUNCH_BWD(true, 0, false); }
      else { if (matm == 2) FFX_LAUNCH_BWD(false, 2, false); else if (matm == 1) FFX_LAUNCH_BWD(false, 1, false); else FFX_LAUNCH_BWD(false, 0, false); }
    };
#undef FFX_LAUNCH_BWD
    if (det_part) { // one pass of the deterministic accumulation alone: the caller owns the accumulators and the scale (ffx_render_bwd_det_part)
      det.mode = det_part->part;
      det.vmax = det_part->vmax;
      det.fix = det_part->fix;
      det.scale = ldexpf(1.0f, det_part->scale_log2);
      launch();
      FFX_CHECK_LAUNCH(det_part->part == 1 ? "render_bwd_det_part/max" : "render_bwd_det_part/sum");
      return FFX_OK;
    }
    if (det_ws) {
      // deterministic accumulation (DetK above): the largest tap of the launch -> a power-of-two scale -> 64-bit fixed-point sums -> gtex.  The
      // scale needs the first pass's result on the host: ONE 4-byte read and a stream synchronisation per call (a debugging / cross-checking mode)
      const long n_t = (long)c.tw * c.th * c.tc;
      det.fix = (unsigned long long *)det_ws;
      det.vmax = (unsigned int *)(det.fix + n_t);
      if (hipMemsetAsync(det_ws, 0, (size_t)n_t * 8 + 8, (hipStream_t)s) != hipSuccess) FFX_FAIL(FFX_ERR_LAUNCH, "render_bwd_det: clearing the workspace failed");
      det.mode = 1;
      launch();
      FFX_CHECK_LAUNCH("render_bwd_det/max");
      unsigned int vbits = 0;
      if (hipMemcpyAsync(&vbits, det.vmax, 4, hipMemcpyDeviceToHost, (hipStream_t)s) != hipSuccess || hipStreamSynchronize((hipStream_t)s) != hipSuccess)
        FFX_FAIL(FFX_ERR_LAUNCH, "render_bwd_det: reading the largest tap failed");
      float vmax;
      memcpy(&vmax, &vbits, 4);
      if (!(vmax > 0.f)) return FFX_OK; // nothing lit: gtex unchanged
      if (!(vmax < 3.0e38f)) { // a non-finite tap: the plain adjoint reports it (NaN / inf in gtex)
        det.mode = 0;
        launch();
        FFX_CHECK_LAUNCH("render_bwd_det/non-finite");
        return FFX_OK;
      }
      // (b = bits of this launch's tap count — 4 taps per sample: 2^-36 of the largest tap at 512 x 512 x 64 spp; a fixed b = 34 — room for 2^32
      // samples — had left 2^-28, round-5 advisor)
      const int sh = det_scale_log2(vbits, 4ull * (unsigned long long)c.cam.W * (unsigned long long)c.cam.H * (unsigned long long)(spp > 0 ? spp : 1));
      det.scale = ldexpf(1.0f, sh);
      det.mode = 2;
      launch();
      FFX_CHECK_LAUNCH("render_bwd_det/sum");
      hipLaunchKernelGGL(k_det_finish, dim3(ffx_cdiv(n_t, 256)), dim3(256), 0, (hipStream_t)s, det.fix, 1.0f / det.scale, n_t, gtex);
      FFX_CHECK_LAUNCH("render_bwd_det/finish");
      return FFX_OK;
    }
    launch();
    FFX_CHECK_LAUNCH(rf_scratch ? "render_bwd_filtered" : "render_bwd");
    return FFX_OK;
  }
  if (det_ws || det_part) FFX_FAIL(FFX_ERR_UNSUPPORTED, "render_bwd_det: the per-lane kernels (FFX_TRAVERSAL=lane) have no deterministic mode");
  int tiles_x = ffx_cdiv(c.cam.W, 8), tiles_y = ffx_cdiv(c.cam.H, 8);
  int n_tiles = tiles_x * tiles_y;
  int grid = ((n_tiles + 7) / 8) * 8;
  hipLaunchKernelGGL(k_render_bwd, dim3(grid), dim3(TR_BLOCK), stack_bytes(info), (hipStream_t)s, c, nodes, recs, nrec, shape_albedo, spp, seed_key_of(seed),
                     tiles_x, n_tiles, xcd_mode(), gimg, gtex);
  FFX_CHECK_LAUNCH("render_bwd");
  return FFX_OK;
}

int ffx_render_bwd(const void *bvh, const ffx_bvh_info *info, const ffx_scene_desc *sd, const float *shape_albedo, int spp, uint32_t seed, int flags,
                   const float *gimg, float *gtex, ffx_stream s) {
  return render_bwd_impl(bvh, info, sd, shape_albedo, spp, seed, flags, gimg, gtex, s, nullptr);
}

size_t ffx_render_bwd_det_bytes(const ffx_scene_desc *sd) {
  if (!sd || sd->proj.tex_w < 1 || sd->proj.tex_h < 1 || (sd->proj.tex_channels != 1 && sd->proj.tex_channels != 3)) return 0;
  return (size_t)sd->proj.tex_w * sd->proj.tex_h * sd->proj.tex_channels * 8 + 8 + (sd->rfilter != FFX_RFILTER_BOX ? ffx_render_filter_bytes(sd) : 0);
}

int ffx_render_bwd_det(const void *bvh, const ffx_bvh_info *info, const ffx_scene_desc *sd, const float *shape_albedo, int spp, uint32_t seed, int flags,
                       const float *gimg, float *gtex, void *workspace, ffx_stream s) {
  if (!workspace || ((uintptr_t)workspace & 15) != 0) FFX_FAIL(FFX_ERR_ARG, "render_bwd_det: workspace is NULL or not 16-byte aligned");
  if (!sd) FFX_FAIL(FFX_ERR_ARG, "render_bwd_det: bad argument");
  // workspace: [the filtered film's scratch (16-byte multiple)] [one 64-bit sum per texel and channel] [the largest tap]
  void *rf = sd->rfilter != FFX_RFILTER_BOX ? workspace : nullptr;
  void *det = (char *)workspace + (rf ? ffx_render_filter_bytes(sd) : 0);
  return render_bwd_impl(bvh, info, sd, shape_albedo, spp, seed, flags, gimg, gtex, s, rf, det);
}

int ffx_render_bwd_det_part(const void *bvh, const ffx_bvh_info *info, const ffx_scene_desc *sd, const float *shape_albedo, int spp, uint32_t seed, int flags,
                            const float *gimg, int part, int scale_log2, void *acc, void *workspace, ffx_stream s) {
  if (!sd || !acc || (part != 1 && part != 2) || ((uintptr_t)acc & (part == 1 ? 3 : 7)) != 0) FFX_FAIL(FFX_ERR_ARG, "render_bwd_det_part: bad argument");
  if (scale_log2 < -126 || scale_log2 > 126) FFX_FAIL(FFX_ERR_ARG, "render_bwd_det_part: scale_log2 out of range");
  void *rf = nullptr;
  if (sd->rfilter != FFX_RFILTER_BOX) {
    if (!workspace || ((uintptr_t)workspace & 15) != 0) FFX_FAIL(FFX_ERR_ARG, "render_bwd_det_part: a filtered film needs its scratch (ffx_render_filter_bytes), 16-byte aligned");
    rf = workspace;
  }
  DetPart dp;
  dp.part = part; dp.scale_log2 = scale_log2;
  dp.vmax = part == 1 ? (unsigned int *)acc : nullptr;
  dp.fix = part == 2 ? (unsigned long long *)acc : nullptr;
  return render_bwd_impl(bvh, info, sd, shape_albedo, spp, seed, flags, gimg, nullptr, s, rf, nullptr, &dp);
}
int ffx_det_scale_log2(uint32_t vmax_bits, uint64_t n_taps) { return det_scale_log2(vmax_bits, (unsigned long long)n_taps); }
int ffx_det_finish(const void *acc, int scale_log2, size_t n, float *gtex, ffx_stream s) {
  if (!acc || !gtex || n == 0 || scale_log2 < -126 || scale_log2 > 126) FFX_FAIL(FFX_ERR_ARG, "det_finish: bad argument");
  hipLaunchKernelGGL(k_det_finish, dim3(ffx_cdiv((long)n, 256)), dim3(256), 0, (hipStream_t)s, (const unsigned long long *)acc, ldexpf(1.0f, -scale_log2), (long)n, gtex);
  FFX_CHECK_LAUNCH("det_finish");
  return FFX_OK;
}

// scratch of the filtered calls: [pixel][25][4] outgoing sums (400 B per pixel) + G = gimg / weight as float4 per pixel (the adjoint)
size_t ffx_render_filter_bytes(const ffx_scene_desc *sd) {
  if (!sd || sd->cam.width < 1 || sd->cam.height < 1) return 0;
  return (size_t)sd->cam.width * sd->cam.height * (25 * 4 + 4) * sizeof(float);
}

int ffx_render_fwd_filtered(const void *bvh, const ffx_bvh_info *info, const ffx_scene_desc *sd, const float *shape_albedo, const float *tex, int spp,
                            uint32_t seed, int img_fp16, void *img, void *scratch, ffx_stream s) {
  if (!scratch || ((uintptr_t)scratch & 15) != 0) FFX_FAIL(FFX_ERR_ARG, "render_fwd_filtered: scratch is NULL or not 16-byte aligned");
  return render_fwd_impl(bvh, info, sd, shape_albedo, tex, spp, seed, img_fp16 & (FFX_RENDER_FP16 | FFX_RENDER_APEX_READY), img, nullptr, s, nullptr, nullptr, nullptr,
                         scratch);
}

int ffx_render_fwd_adjoint_filtered(const void *bvh, const ffx_bvh_info *info, const ffx_scene_desc *sd, const float *shape_albedo, const float *tex, int spp,
                                    uint32_t seed, int img_fp16, void *img, const float *gimg, float *gtex, void *scratch, ffx_stream s) {
  if (!gimg || !gtex) FFX_FAIL(FFX_ERR_ARG, "render_fwd_adjoint_filtered: gimg / gtex is NULL");
  if (!scratch || ((uintptr_t)scratch & 15) != 0) FFX_FAIL(FFX_ERR_ARG, "render_fwd_adjoint_filtered: scratch is NULL or not 16-byte aligned");
  if (sd && !sd->proj.enabled) FFX_FAIL(FFX_ERR_ARG, "render_fwd_adjoint_filtered: the scene has no projector (nothing to differentiate)");
  return render_fwd_impl(bvh, info, sd, shape_albedo, tex, spp, seed, img_fp16 & (FFX_RENDER_FP16 | FFX_RENDER_SPARSE_ADJOINT | FFX_RENDER_APEX_READY), img, nullptr, s, gimg,
                         gtex, nullptr, scratch);
}

int ffx_render_bwd_filtered(const void *bvh, const ffx_bvh_info *info, const ffx_scene_desc *sd, const float *shape_albedo, int spp, uint32_t seed, int flags,
                            const float *gimg, float *gtex, void *scratch, ffx_stream s) {
  if (!scratch || ((uintptr_t)scratch & 15) != 0) FFX_FAIL(FFX_ERR_ARG, "render_bwd_filtered: scratch is NULL or not 16-byte aligned");
  return render_bwd_impl(bvh, info, sd, shape_albedo, spp, seed, flags, gimg, gtex, s, scratch);
}

int ffx_render_fwd_cache_filtered(const void *bvh, const ffx_bvh_info *info, const ffx_scene_desc *sd, const float *shape_albedo, const float *tex, int spp,
                                  uint32_t seed, int img_fp16, void *img, void *cache, void *scratch, ffx_stream s) {
  if (!cache || ((uintptr_t)cache & 15) != 0) FFX_FAIL(FFX_ERR_ARG, "render_fwd_cache_filtered: cache is NULL or not 16-byte aligned");
  if (!scratch || ((uintptr_t)scratch & 15) != 0) FFX_FAIL(FFX_ERR_ARG, "render_fwd_cache_filtered: scratch is NULL or not 16-byte aligned");
  return render_fwd_impl(bvh, info, sd, shape_albedo, tex, spp, seed,
                         img_fp16 & (FFX_RENDER_FP16 | FFX_RENDER_SPARSE_ADJOINT | FFX_RENDER_APEX_READY | FFX_RENDER_CACHE_ZEROED | FFX_RENDER_CACHE_KEEP_DROPPED), img, nullptr, s,
                         nullptr, nullptr, nullptr, scratch, cache);
}

int ffx_render_bwd_cached_filtered(const ffx_scene_desc *sd, const float *shape_albedo, const void *cache, int spp, uint32_t seed, const float *gimg, float *gtex,
                                   ffx_stream s) {
  if (!sd || (!shape_albedo && sd->n_mat_h <= 0) || !cache || !gimg || !gtex || spp < 1) FFX_FAIL(FFX_ERR_ARG, "render_bwd_cached_filtered: bad argument");
  if (sd->rfilter != FFX_RFILTER_GAUSSIAN) FFX_FAIL(FFX_ERR_UNSUPPORTED, "render_bwd_cached_filtered: rfilter must be FFX_RFILTER_GAUSSIAN (the box film: ffx_render_bwd_cached)");
  if (spp > 1024) FFX_FAIL(FFX_ERR_UNSUPPORTED, "render_bwd_cached_filtered: more than 1024 samples per pixel");
  if (!sd->proj.enabled) return FFX_OK;
  BwdF p;
  memset(&p, 0, sizeof p);
  p.mats = shape_albedo;
  p.ms = sd->mat_stride ? sd->mat_stride : 3;
  if (p.ms != 3 && p.ms != FFX_MAT_STRIDE) FFX_FAIL(FFX_ERR_ARG, "render_bwd_cached_filtered: bad material stride");
  if (sd->n_mat_h > 0) {
    if (sd->n_mat_h > FFX_MAX_MAT_H || sd->n_mat_h != sd->n_shapes * p.ms) FFX_FAIL(FFX_ERR_ARG, "render_bwd_cached_filtered: n_mat_h must be n_shapes x stride (<= %d)", FFX_MAX_MAT_H);
    p.mat_inline = 1;
    for (int i = 0; i < sd->n_mat_h; ++i) p.mat_h[i] = sd->mat_h[i];
  }
  p.tw = sd->proj.tex_w; p.th = sd->proj.tex_h; p.tc = sd->proj.tex_channels; p.spp = spp;
  p.W = sd->cam.width; p.H = sd->cam.height;
  if (p.tw < 1 || p.th < 1 || (p.tc != 1 && p.tc != 3) || p.W < 1 || p.H < 1) FFX_FAIL(FFX_ERR_ARG, "render_bwd_cached_filtered: bad scene description");
  for (int i = 0; i < 3; ++i) p.color[i] = sd->proj.color[i];
  const float sdv = sd->rfilter_stddev > 0.f ? sd->rfilter_stddev : 0.5f;
  if (!(sdv <= 0.5f)) FFX_FAIL(FFX_ERR_UNSUPPORTED, "render_bwd_cached_filtered: gaussian filter: stddev <= 0.5");
  rf_constants(sdv, p.rf); // (as shade_prepare)
  p.seed_key = seed_key_of(seed);
  p.n_shapes = sd->n_shapes;
  if (p.n_shapes < 1 || p.n_shapes > 255) FFX_FAIL(FFX_ERR_UNSUPPORTED, "render_bwd_cached_filtered: 1 .. 255 shapes");
  const size_t n_pix = (size_t)p.W * p.H;
  p.off_wsum = rfc_off_wsum(n_pix); p.off_recs = rfc_off_recs(n_pix); p.off_facb = rfc_off_facb(n_pix, (size_t)spp, rfc_all(sd));
  const int blocks_x = ffx_cdiv(p.W, K9F_BLOCK), blocks_y = ffx_cdiv(p.H, K9F_BLOCK);
  hipLaunchKernelGGL(k_render_bwd_cached_filtered, dim3(blocks_x * blocks_y * K9F_SUB), dim3(64), 0, (hipStream_t)s, p, (const char *)cache, blocks_x, gimg, gtex);
  FFX_CHECK_LAUNCH("render_bwd_cached_filtered");
  return FFX_OK;
}

} // extern "C"
