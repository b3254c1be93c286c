// ffx_splat.hip — K1 (pattern projection), K2 (soft point splatting, forward + gradient),
// K3 (texture blur) for gfx950.  See include/ffx.h for the reference call each entry replaces
// and DESIGN.md §5 for the kernel design and roofline of each.
//
// All arithmetic follows the operation order of the reference's torch code (separate multiplies
// and adds; the file is compiled with -ffp-contract=off), so results agree with the reference to
// the last few ulps of expf.
#include "ffx_common.h"
#include <cstring>

#define SPLAT_BLOCK 256
#define TILE_W 32
#define TILE_H 8
#define CAND_MAX 512   // candidates staged per chunk in the fused forward kernel
#define NEIGH_MAX 2048 // neighbour list capacity of the gradient kernel
// exp(-q^2) is exactly 0 in binary32 for q^2 > 103.98; 10.3^2 = 106.1 leaves a margin, so culling
// on q > FFX_QCUT never drops a non-zero term.
#define FFX_QCUT 10.3f

// ------------------------------------------------------------------------------- helpers
// value of one splat at texel (fj, fi); rasterization.py:29-35
__device__ __forceinline__ float splat_val(float fj, float fi, float p0s, float p1s, float sigma, float inv_sigma, float &d, float &yd, float &xd) {
  yd = fj - p0s;
  xd = fi - p1s;
  d = yd * yd + xd * xd;
  if (d * inv_sigma > FFX_QCUT) return 0.f;
  float q = d / sigma;
  return expf(-(q * q));
}
// dv/d(p0*size0) = v*4*d*yd/sigma^2 (and xd for the other axis)
__device__ __forceinline__ float splat_gcoef(float v, float d, float sigma) { return v * 4.0f * d / (sigma * sigma); }

// window of baked_sum/baked_softor along one axis (rasterization.py:180-235); mirrors
// window_axis() of the oracle.  Returns false if the point contributes nothing on this axis.
struct Win1 { int lo, hi, off; float pm; };
__device__ __forceinline__ bool window_axis(float p, int half, int size, Win1 &w) {
  int fp = 2 * half + 1;
  int fo = (int)floorf(p - (float)half);
  int rs = 0, re = fp;
  if (fo < 0) { rs = -fo; fo = 0; }
  if (fo + fp >= size) re = size - fo;
  if (!(rs < re)) return false;
  w.pm = p - floorf(p) + (float)half;
  w.lo = fo;
  w.hi = fo + re - rs;
  w.off = rs - fo; // dist index a = A + off
  return true;
}
__device__ __forceinline__ float baked_val(int A, int B, const Win1 &w0, const Win1 &w1, float sigma, float &d, float &yd, float &xd) {
  yd = (float)(A + w0.off) - w0.pm;
  xd = (float)(B + w1.off) - w1.pm;
  d = yd * yd + xd * xd;
  float q = d / sigma;
  return expf(-(q * q));
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

// =================================================================================== K1
__global__ void __launch_bounds__(256) k_project_fwd(const float *__restrict__ rays, int n, Mat4 KF, float *__restrict__ pts) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float x = rays[3 * i], y = rays[3 * i + 1], z = rays[3 * i + 2];
  const float *K = KF.m;
  float q0 = K[0] * x + K[1] * y + K[2] * z + K[3];
  float q1 = K[4] * x + K[5] * y + K[6] * z + K[7];
  float q2 = K[8] * x + K[9] * y + K[10] * z + K[11];
  float q3 = K[12] * x + K[13] * y + K[14] * z + K[15];
  pts[3 * i] = q0 / q3;
  pts[3 * i + 1] = q1 / q3;
  pts[3 * i + 2] = q2 / q3;
}

__global__ void __launch_bounds__(256)
    k_project_bwd(const float *__restrict__ rays, int n, Mat4 KF, const float *__restrict__ gpts, float *__restrict__ grays) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float x = rays[3 * i], y = rays[3 * i + 1], z = rays[3 * i + 2];
  const float *K = KF.m;
  float q[4], gq[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) q[r] = K[4 * r] * x + K[4 * r + 1] * y + K[4 * r + 2] * z + K[4 * r + 3];
  float iw = 1.0f / q[3];
  float acc = 0.f;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    gq[k] = gpts[3 * i + k] * iw;
    acc += gpts[3 * i + k] * q[k];
  }
  gq[3] = -acc * iw * iw;
#pragma unroll
  for (int c = 0; c < 3; ++c) grays[3 * i + c] = gq[0] * K[c] + gq[1] * K[4 + c] + gq[2] * K[8 + c] + gq[3] * K[12 + c];
}

__global__ void __launch_bounds__(256) k_transform_points(const float *__restrict__ pts, int n, Mat4 M, int mode, float *__restrict__ out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float x = pts[3 * i], y = pts[3 * i + 1], z = pts[3 * i + 2];
  const float *K = M.m;
  if (mode == 0) {
    float q0 = K[0] * x + K[1] * y + K[2] * z + K[3];
    float q1 = K[4] * x + K[5] * y + K[6] * z + K[7];
    float q2 = K[8] * x + K[9] * y + K[10] * z + K[11];
    float q3 = K[12] * x + K[13] * y + K[14] * z + K[15];
    out[3 * i] = q0 / q3;
    out[3 * i + 1] = q1 / q3;
    out[3 * i + 2] = q2 / q3;
  } else {
#pragma unroll
    for (int r = 0; r < 3; ++r) out[3 * i + r] = K[4 * r] * x + K[4 * r + 1] * y + K[4 * r + 2] * z;
  }
}

// clamp_to_fov + normalize_rays in one launch (the pattern has 64..1024 rays: launch-bound; as separate
// torch ops this was 13 launches per optimiser step)
__global__ void __launch_bounds__(256) k_clamp_to_fov(float *__restrict__ rays, int n, Mat4 KF, Mat4 KI, float lo, float hi, int n_norm) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float x = rays[3 * i], y = rays[3 * i + 1], z = rays[3 * i + 2];
  const float *K = KF.m;
  float q0 = K[0] * x + K[1] * y + K[2] * z + K[3];
  float q1 = K[4] * x + K[5] * y + K[6] * z + K[7];
  float q2 = K[8] * x + K[9] * y + K[10] * z + K[11];
  float q3 = K[12] * x + K[13] * y + K[14] * z + K[15];
  float px = q0 / q3, py = q1 / q3, pz = q2 / q3;
  px = fminf(fmaxf(px, lo), hi);
  py = fminf(fmaxf(py, lo), hi);
  const float *I = KI.m;
  float w0 = I[0] * px + I[1] * py + I[2] * pz + I[3];
  float w1 = I[4] * px + I[5] * py + I[6] * pz + I[7];
  float w2 = I[8] * px + I[9] * py + I[10] * pz + I[11];
  float w3 = I[12] * px + I[13] * py + I[14] * pz + I[15];
  x = w0 / w3; y = w1 / w3; z = w2 / w3;
  for (int k = 0; k < n_norm; ++k) {
    float nrm = sqrtf(x * x + y * y + z * z);
    x /= nrm; y /= nrm; z /= nrm;
  }
  rays[3 * i] = x; rays[3 * i + 1] = y; rays[3 * i + 2] = z;
}

// L1Loss(a, b) * weight and its gradient in two launches (sub, abs, mean, mul, sign, mul, neg as torch ops
// were seven): partial sums per workgroup in a fixed order, then one workgroup adds the <= 256 partials
__global__ void __launch_bounds__(256) k_l1_partial(const float *__restrict__ a, const float *__restrict__ b, long n, float gscale, float *__restrict__ ws,
                                                     float *__restrict__ g) {
  __shared__ float s_part[4];
  float acc = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float d = a[i] - b[i];
    acc += fabsf(d);
    g[i] = d > 0.f ? gscale : (d < 0.f ? -gscale : 0.f);
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) acc += __shfl_down(acc, o, 64);
  if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) ws[1 + blockIdx.x] = (s_part[0] + s_part[1]) + (s_part[2] + s_part[3]);
}
__global__ void __launch_bounds__(256) k_l1_final(float *__restrict__ ws, int nblocks, float vscale, float *__restrict__ acc_out) {
  __shared__ float s_part[4];
  float acc = (int)threadIdx.x < nblocks ? ws[1 + threadIdx.x] : 0.f;
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) acc += __shfl_down(acc, o, 64);
  if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float v = ((s_part[0] + s_part[1]) + (s_part[2] + s_part[3])) * vscale;
    ws[0] = v;
    if (acc_out) acc_out[0] += v; // (ffx_l1_value_grad_acc: a step's running loss — launches on one stream are ordered, one thread writes)
  }
}

// =================================================================================== K2 dense forward
// rasterize_points: [n,size1,size0] layers.  HBM-write bound (4 B per (point, texel)); each lane
// produces 4 consecutive texels and issues one 16-byte store when rows are 16-byte aligned.
// MODE 0: plain; MODE 1: rasterize_depth (divide by the layer maximum, scale by depth).
template <int MODE>
__global__ void __launch_bounds__(SPLAT_BLOCK)
    k_splat_dense_fwd(const float *__restrict__ pts, const float *__restrict__ depth, float sigma, int size0, int size1, int w4, int vec_ok,
                      float *__restrict__ out) {
  int n = blockIdx.y;
  long t = (long)blockIdx.x * SPLAT_BLOCK + threadIdx.x;
  if (t >= (long)w4 * size1) return;
  int i = (int)(t / w4), j0 = (int)(t % w4) * 4;
  float p0s = pts[2 * n] * (float)size0, p1s = pts[2 * n + 1] * (float)size1;
  float inv_sigma = 1.0f / sigma;
  float scale_den = 1.f, scale_mul = 1.f;
  if (MODE == 1) {
    // the layer maximum sits at the texel nearest to the point (v is monotone in each |dist|)
    float jn = fminf(fmaxf(rintf(p0s), 0.f), (float)(size0 - 1));
    float in_ = fminf(fmaxf(rintf(p1s), 0.f), (float)(size1 - 1));
    float yd = jn - p0s, xd = in_ - p1s;
    float d = yd * yd + xd * xd;
    float q = d / sigma;
    scale_den = expf(-(q * q));
    scale_mul = depth[n];
  }
  float v[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    float d, yd, xd;
    float val = splat_val((float)(j0 + k), (float)i, p0s, p1s, sigma, inv_sigma, d, yd, xd);
    if (MODE == 1) val = (val / scale_den) * scale_mul;
    v[k] = val;
  }
  float *o = out + ((size_t)n * size1 + i) * size0 + j0;
  if (vec_ok) {
    *reinterpret_cast<float4 *>(o) = make_float4(v[0], v[1], v[2], v[3]);
  } else {
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (j0 + k < size0) o[k] = v[k];
  }
}

// rasterize_lines (rasterization.py:107-153)
__global__ void __launch_bounds__(SPLAT_BLOCK) k_splat_lines_fwd(const float *__restrict__ lines, float sigma, int size0, int size1, float *__restrict__ out) {
  int n = blockIdx.y;
  long t = (long)blockIdx.x * SPLAT_BLOCK + threadIdx.x;
  if (t >= (long)size0 * size1) return;
  int a = (int)(t / size0), b = (int)(t % size0);
  const float eps = 1.1920928955078125e-07f;
  float sx = lines[4 * n + 0] * (float)size0, sy = lines[4 * n + 1] * (float)size1;
  float ex = lines[4 * n + 2] * (float)size0, ey = lines[4 * n + 3] * (float)size1;
  float mx = ex - sx, my = ey - sy;
  float mm = mx * mx + my * my + eps;
  float X = (float)b, Y = (float)a;
  float pax = X - sx, pay = Y - sy, pbx = X - ex, pby = Y - ey;
  float t0 = (pax * mx + pay * my) / mm;
  float qx = X - (sx + t0 * mx), qy = Y - (sy + t0 * my);
  float d0 = (t0 <= 0.f) ? (pax * pax + pay * pay) : 0.f;
  float d1 = (t0 > 0.f && t0 < 1.f) ? (qx * qx + qy * qy) : 0.f;
  float d2 = (t0 >= 1.f) ? (pbx * pbx + pby * pby) : 0.f;
  float dist = d0 + d1 + d2;
  out[((size_t)n * size1 + a) * size0 + b] = expf(-(dist * dist) / (sigma * sigma));
}

// autograd of rasterize_lines w.r.t. the segment end points: one workgroup per segment walks the layer,
// per texel  g * d out / d dist2 * d dist2 / d (start, end)  with out = exp(-dist2^2 / sigma^2); fp64 partial
// sums, wave shuffles -> LDS -> one store per coordinate: no atomics, reproducible.
__global__ void __launch_bounds__(1024) k_splat_lines_bwd(const float *__restrict__ lines, float sigma, int size0, int size1, const float *__restrict__ gout,
                                                         float *__restrict__ glines) {
  __shared__ double s_part[16][4];
  const int n = blockIdx.x;
  const float eps = 1.1920928955078125e-07f;
  const float sx = lines[4 * n + 0] * (float)size0, sy = lines[4 * n + 1] * (float)size1;
  const float ex = lines[4 * n + 2] * (float)size0, ey = lines[4 * n + 3] * (float)size1;
  const float mx = ex - sx, my = ey - sy;
  const float mm = mx * mx + my * my + eps;
  const float inv_s2 = 1.0f / (sigma * sigma);
  double acc[4] = {0.0, 0.0, 0.0, 0.0}; // d/d(sx, sy, ex, ey) in texel units
  const long total = (long)size0 * size1;
  const float *g = gout + (size_t)n * total;
  for (long t = threadIdx.x; t < total; t += 1024) {
    const float go = g[t];
    if (go == 0.f) continue;
    const int a = (int)(t / size0), b = (int)(t % size0);
    const float X = (float)b, Y = (float)a;
    const float pax = X - sx, pay = Y - sy, pbx = X - ex, pby = Y - ey;
    const float t0 = (pax * mx + pay * my) / mm;
    float dist, dsx, dsy, dex, dey; // dist2 and its derivatives
    if (t0 <= 0.f) {
      dist = pax * pax + pay * pay;
      dsx = -2.f * pax; dsy = -2.f * pay; dex = 0.f; dey = 0.f;
    } else if (t0 < 1.f) {
      const float qx = X - (sx + t0 * mx), qy = Y - (sy + t0 * my);
      dist = qx * qx + qy * qy;
      // d dist2 = -2 q.dS - 2 t0 q.dm - 2 (q.m) dt0,  dt0 = (-m.dS + pa.dm) / mm - 2 t0 (m.dm) / mm,  dm = dE - dS
      const float qm = (qx * mx + qy * my) / mm;
      // coefficient vectors of dt0:  w.r.t. S: (-m - pa + 2 t0 m) / mm ; w.r.t. E: (pa - 2 t0 m) / mm   (the 1/mm is in qm)
      const float cSx = -mx - pax + 2.f * t0 * mx, cSy = -my - pay + 2.f * t0 * my;
      const float cEx = pax - 2.f * t0 * mx, cEy = pay - 2.f * t0 * my;
      dsx = -2.f * qx + 2.f * t0 * qx - 2.f * qm * cSx;
      dsy = -2.f * qy + 2.f * t0 * qy - 2.f * qm * cSy;
      dex = -2.f * t0 * qx - 2.f * qm * cEx;
      dey = -2.f * t0 * qy - 2.f * qm * cEy;
    } else {
      dist = pbx * pbx + pby * pby;
      dsx = 0.f; dsy = 0.f; dex = -2.f * pbx; dey = -2.f * pby;
    }
    const float o = expf(-(dist * dist) * inv_s2);
    const double k = (double)go * (double)(o * (-2.f * dist * inv_s2));
    acc[0] += k * dsx; acc[1] += k * dsy; acc[2] += k * dex; acc[3] += k * dey;
  }
#pragma unroll
  for (int c = 0; c < 4; ++c)
    for (int off = 32; off >= 1; off >>= 1) acc[c] += __shfl_down(acc[c], off, 64);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0)
    for (int c = 0; c < 4; ++c) s_part[wave][c] = acc[c];
  __syncthreads();
  if (threadIdx.x < 4) {
    double v = 0.0;
    for (int w = 0; w < 16; ++w) v += s_part[w][threadIdx.x];
    // chain through the scaling of the end points by the texture size (rasterization.py:122-123)
    glines[4 * n + threadIdx.x] = (float)(v * (double)((threadIdx.x & 1) ? size1 : size0));
  }
}

// =================================================================================== K2 fused forward
// One workgroup per 32x8 texel tile.  Wave 0 culls the points against the tile (ballot +
// prefix: the compacted list keeps ascending point order, so every texel accumulates in the
// same order as torch.sum/prod over dim 0) and stages the survivors in LDS; every lane then
// reads the staged points as LDS broadcasts.  The [n,size1,size0] tensor never exists.
template <bool BAKED>
__global__ void __launch_bounds__(SPLAT_BLOCK)
    k_splat_fused_fwd(const float *__restrict__ pts, int n, float sigma, int reduce, int half_window, int size0, int size1, float *__restrict__ tex) {
  __shared__ float c_p0[CAND_MAX], c_p1[CAND_MAX];
  __shared__ int c_lo0[BAKED ? CAND_MAX : 1], c_hi0[BAKED ? CAND_MAX : 1], c_lo1[BAKED ? CAND_MAX : 1], c_hi1[BAKED ? CAND_MAX : 1];
  __shared__ int c_of0[BAKED ? CAND_MAX : 1], c_of1[BAKED ? CAND_MAX : 1];
  __shared__ int c_count;

  const int tid = threadIdx.x;
  const int j0 = blockIdx.x * TILE_W, i0 = blockIdx.y * TILE_H;
  const int j = j0 + (tid % TILE_W), i = i0 + (tid / TILE_W);
  const float inv_sigma = 1.0f / sigma;
  float acc = (reduce == FFX_REDUCE_SUM) ? 0.f : 1.f;

  for (int chunk = 0; chunk < n; chunk += CAND_MAX) {
    if (tid < 64) { // wave 0: ordered compaction of this chunk
      int count = 0;
      int lim = min(CAND_MAX, n - chunk);
      for (int base = 0; base < lim; base += 64) {
        int k = chunk + base + tid;
        bool keep = false;
        float p0s = 0.f, p1s = 0.f;
        Win1 w0, w1;
        if (base + tid < lim) {
          p0s = pts[2 * k] * (float)size0;
          p1s = pts[2 * k + 1] * (float)size1;
          if (BAKED) {
            keep = window_axis(p0s, half_window, size0, w0) && window_axis(p1s, half_window, size1, w1);
            keep = keep && w0.lo < j0 + TILE_W && w0.hi > j0 && w1.lo < i0 + TILE_H && w1.hi > i0;
          } else {
            float dx = fmaxf(fmaxf((float)j0 - p0s, p0s - (float)(j0 + TILE_W - 1)), 0.f);
            float dy = fmaxf(fmaxf((float)i0 - p1s, p1s - (float)(i0 + TILE_H - 1)), 0.f);
            keep = (dx * dx + dy * dy) * inv_sigma <= FFX_QCUT;
          }
        }
        unsigned long long m = __ballot(keep);
        int pos = count + __popcll(m & ((1ull << tid) - 1ull));
        if (keep) {
          if (BAKED) {
            c_p0[pos] = w0.pm; c_p1[pos] = w1.pm;
            c_lo0[pos] = w0.lo; c_hi0[pos] = w0.hi; c_of0[pos] = w0.off;
            c_lo1[pos] = w1.lo; c_hi1[pos] = w1.hi; c_of1[pos] = w1.off;
          } else {
            c_p0[pos] = p0s; c_p1[pos] = p1s;
          }
        }
        count += __popcll(m);
      }
      if (tid == 0) c_count = count;
    }
    __syncthreads();
    int cnt = c_count;
    if (j < size0 && i < size1) {
      for (int c = 0; c < cnt; ++c) {
        float v, d, yd, xd;
        if (BAKED) {
          if (j < c_lo0[c] || j >= c_hi0[c] || i < c_lo1[c] || i >= c_hi1[c]) continue;
          Win1 w0, w1;
          w0.off = c_of0[c]; w0.pm = c_p0[c];
          w1.off = c_of1[c]; w1.pm = c_p1[c];
          v = baked_val(j, i, w0, w1, sigma, d, yd, xd);
        } else {
          v = splat_val((float)j, (float)i, c_p0[c], c_p1[c], sigma, inv_sigma, d, yd, xd);
        }
        if (reduce == FFX_REDUCE_SUM) acc += v; else acc *= (1.0f - v);
      }
    }
    __syncthreads();
  }
  if (j < size0 && i < size1) tex[(size_t)i * size0 + j] = (reduce == FFX_REDUCE_SUM) ? acc : (1.0f - acc);
}

// =================================================================================== K2 gradient
// One workgroup per point: sweeps only the texels where the point's value is non-zero in
// binary32 (or the baked window), so the traffic is the footprint of `gtex`, not [n,H,W].
// softor needs prod_{m != n}(1 - v_m) per texel: the points whose footprint overlaps this one
// are compacted into LDS once per workgroup.  Per-lane partial sums are reduced with wave
// shuffles, then across the 4 waves through LDS (fp64 partial sums): one plain store per point, no
// atomics, and the result is bitwise reproducible.
// LAYERED: upstream gradient is the dense [n,size1,size0] tensor (rasterize_points backward).
template <bool BAKED, bool LAYERED>
__global__ void __launch_bounds__(SPLAT_BLOCK)
    k_splat_bwd(const float *__restrict__ pts, int n, float sigma, int reduce, int half_window, int size0, int size1, const float *__restrict__ gtex,
                float *__restrict__ gpts) {
  __shared__ float nb_p0[NEIGH_MAX], nb_p1[NEIGH_MAX];
  __shared__ int nb_count;
  __shared__ double red0[SPLAT_BLOCK / 64], red1[SPLAT_BLOCK / 64];

  const int k = blockIdx.x, tid = threadIdx.x;
  const float inv_sigma = 1.0f / sigma;
  const float p0s = pts[2 * k] * (float)size0, p1s = pts[2 * k + 1] * (float)size1;
  const float R = sqrtf(FFX_QCUT * sigma) + 1.0f; // radius beyond which v is exactly 0
  int lo0, hi0, lo1, hi1;                          // texel range (exclusive hi)
  Win1 w0, w1;
  bool alive = true;
  if (BAKED) {
    alive = window_axis(p0s, half_window, size0, w0) && window_axis(p1s, half_window, size1, w1);
    lo0 = w0.lo; hi0 = w0.hi; lo1 = w1.lo; hi1 = w1.hi;
  } else {
    lo0 = max(0, (int)floorf(p0s - R)); hi0 = min(size0, (int)ceilf(p0s + R) + 1);
    lo1 = max(0, (int)floorf(p1s - R)); hi1 = min(size1, (int)ceilf(p1s + R) + 1);
    alive = hi0 > lo0 && hi1 > lo1;
  }
  const bool softor = (reduce == FFX_REDUCE_SOFTOR) && !LAYERED;
  const bool use_list = softor && n <= NEIGH_MAX;
  if (use_list) {
    if (tid < 64) {
      int count = 0;
      for (int base = 0; base < n; base += 64) {
        int m = base + tid;
        bool keep = false;
        float q0 = 0.f, q1 = 0.f;
        if (m < n && m != k) {
          q0 = pts[2 * m] * (float)size0;
          q1 = pts[2 * m + 1] * (float)size1;
          if (BAKED) keep = fabsf(q0 - p0s) <= (float)(2 * half_window + 2) && fabsf(q1 - p1s) <= (float)(2 * half_window + 2);
          else keep = fabsf(q0 - p0s) <= 2.f * R + 2.f && fabsf(q1 - p1s) <= 2.f * R + 2.f;
        }
        unsigned long long mk = __ballot(keep);
        int pos = count + __popcll(mk & ((1ull << tid) - 1ull));
        if (keep) { nb_p0[pos] = q0; nb_p1[pos] = q1; }
        count += __popcll(mk);
      }
      if (tid == 0) nb_count = count;
    }
    __syncthreads();
  }
  // per-texel terms are fp32 like the reference's autograd; their SUM is carried in fp64: these sums
  // cancel heavily (|terms| ~ 1e-2, result ~ 1e-4) and fp64 adds are cheap on CDNA
  double a0 = 0.0, a1 = 0.0;
  if (alive) {
    const int rw = hi0 - lo0, rh = hi1 - lo1;
    const float *g = gtex + (LAYERED ? (size_t)k * size0 * size1 : 0);
    for (int t = tid; t < rw * rh; t += SPLAT_BLOCK) {
      int j = lo0 + t % rw, i = lo1 + t / rw;
      float v, d, yd, xd;
      if (BAKED) v = baked_val(j, i, w0, w1, sigma, d, yd, xd);
      else v = splat_val((float)j, (float)i, p0s, p1s, sigma, inv_sigma, d, yd, xd);
      if (v == 0.f) continue;
      float w = g[(size_t)i * size0 + j];
      if (softor) {
        float prod = 1.f;
        int cnt = use_list ? nb_count : n;
        for (int c = 0; c < cnt; ++c) {
          float q0, q1;
          if (use_list) { q0 = nb_p0[c]; q1 = nb_p1[c]; }
          else {
            if (c == k) continue;
            q0 = pts[2 * c] * (float)size0; q1 = pts[2 * c + 1] * (float)size1;
          }
          float vm, dd, y2, x2;
          if (BAKED) {
            Win1 u0, u1;
            if (!window_axis(q0, half_window, size0, u0) || !window_axis(q1, half_window, size1, u1)) continue;
            if (j < u0.lo || j >= u0.hi || i < u1.lo || i >= u1.hi) continue;
            vm = baked_val(j, i, u0, u1, sigma, dd, y2, x2);
          } else {
            vm = splat_val((float)j, (float)i, q0, q1, sigma, inv_sigma, dd, y2, x2);
          }
          prod *= (1.0f - vm);
        }
        w *= prod;
      }
      float cf = splat_gcoef(v, d, sigma);
      a0 += (double)(w * (cf * yd));
      a1 += (double)(w * (cf * xd));
    }
  }
  a0 = wave_sum(a0);
  a1 = wave_sum(a1);
  if ((tid & 63) == 0) { red0[tid >> 6] = a0; red1[tid >> 6] = a1; }
  __syncthreads();
  if (tid == 0) {
    double s0 = 0.0, s1 = 0.0;
#pragma unroll
    for (int w = 0; w < SPLAT_BLOCK / 64; ++w) { s0 += red0[w]; s1 += red1[w]; }
    gpts[2 * k] = (float)s0 * (float)size0;
    gpts[2 * k + 1] = (float)s1 * (float)size1;
  }
}


// ---- K3 pieces shared by the blur kernels and by the pattern launches that carry the blur along (ffx_pattern_*_blur)
struct BlurW { float w[15]; int ksize; };
__device__ __forceinline__ int reflect_idx(int t, int n) {
  if (n == 1) return 0;
  while (t < 0 || t >= n) {
    if (t < 0) t = -t;
    if (t >= n) t = 2 * (n - 1) - t;
  }
  return t;
}
// K3^T at one pixel (y, x) of an h x w image (both > 2r + 2) from an LDS window of the upstream gradient: win[(gy - oy) * ww + (gx - ox)]
// holds gout(gy, gx), zero outside the image, and covers (at least) rows y - r .. y + r, columns x - r .. x + r and — for the pixels
// next to a border — the first / last r + 1 rows (columns), which the reflected candidates read.  The exact transpose of the
// reflect-padded correlation: gin[q] = sum over the padded positions t (|t - image| <= r) that reflect onto q of
// g_pad[t] = sum_k w[k] * gout[t - k + r], separately per axis; same terms in the same order wherever it is called from.
// KS: the kernel size as a compile-time constant (5: the size every call site of the reference uses, vocalfold_scene.py:61-63) or
// 0 = taken from bw at run time
template <int KS>
__device__ __forceinline__ float blur_bwd_texel(const float *__restrict__ win, int ww, int oy, int ox, int y, int x, int h, int w, const BlurW &bw) {
  const int ksize = KS ? KS : bw.ksize;
  const int r = ksize / 2;
  // Padded rows -r..-1 reflect onto rows 1..r and rows h..h+r-1 onto h-1-r..h-2: only those rows (columns) have
  // candidates besides themselves — every other pixel is a plain correlation.
  const bool edge_y = (y >= 1 && y <= r) || (y >= h - 1 - r && y <= h - 2);
  const bool edge_x = (x >= 1 && x <= r) || (x >= w - 1 - r && x <= w - 2);
  float acc = 0.f;
  if (!edge_y && !edge_x) {
    // no reflected candidate: a plain correlation over the zero-padded window (a tap outside the image adds w * 0, which leaves
    // the running sum as the skipped term of the general loop does).  With KS known the 25 taps are LDS reads at constant
    // offsets: the general loop below spends ~36 instructions per tap on bounds tests and integer multiplies.
    const float *t0 = win + (y - r - oy) * ww + (x - r - ox); // window element of (y - r, x - r)
    if constexpr (KS != 0) {
#pragma unroll
      for (int ky = 0; ky < KS; ++ky) {
        float row = 0.f;
#pragma unroll
        for (int kx = 0; kx < KS; ++kx) row = fmaf(bw.w[kx], t0[(2 * r - ky) * ww + (2 * r - kx)], row);
        acc = fmaf(bw.w[ky], row, acc);
      }
    } else {
      for (int ky = 0; ky < ksize; ++ky) {
        const float *tr = t0 + (2 * r - ky) * ww + 2 * r;
        float row = 0.f;
        for (int kx = 0; kx < ksize; ++kx) row = fmaf(bw.w[kx], tr[-kx], row);
        acc = fmaf(bw.w[ky], row, acc);
      }
    }
    return acc;
  }
  const int na = edge_y ? 2 * r : 0, nb = edge_x ? 2 * r : 0;
  // candidate padded rows: y itself, then the r rows above the image and the r rows below it
  for (int a = -1; a < na; ++a) {
    int ty = (a < 0) ? y : (a < r ? -(a + 1) : h + (a - r));
    if (a >= 0 && reflect_idx(ty, h) != y) continue;
    for (int b = -1; b < nb; ++b) {
      int tx = (b < 0) ? x : (b < r ? -(b + 1) : w + (b - r));
      if (b >= 0 && reflect_idx(tx, w) != x) continue;
      for (int ky = 0; ky < ksize; ++ky) {
        int py = ty - ky + r;
        if (py < 0 || py >= h) continue;
        float row = 0.f;
        for (int kx = 0; kx < ksize; ++kx) {
          int px = tx - kx + r;
          if (px < 0 || px >= w) continue;
          row = fmaf(bw.w[kx], win[(py - oy) * ww + (px - ox)], row);
        }
        acc = fmaf(bw.w[ky], row, acc);
      }
    }
  }
  return acc;
}

// Adam (the arithmetic of torch.optim.Adam's fused kernel: lerp of the first moment, bias corrections from the step count t kept
// on the device) followed by Laser.clamp_to_fov + normalize_rays on the updated ray i
// (pw1, pw2: beta1^t, beta2^t.  The operands of ray i come in an AdamIn, loaded by adam_load — k_pattern_step asks for them before it knows whether
// the update is applied; `kept`: a second place the updated ray is written to, or NULL)
struct AdamIn { float g[3], gb[3], m[3], v[3], r[3]; };
__device__ __forceinline__ void adam_load(int i, const float *rays, const float *__restrict__ grad, const float *__restrict__ grad_b, const float *__restrict__ m,
                                          const float *__restrict__ v, AdamIn &in) {
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    in.g[c] = grad ? grad[3 * i + c] : 0.f;
    in.gb[c] = grad_b ? grad_b[3 * i + c] : 0.f;
    in.m[c] = m[3 * i + c];
    in.v[c] = v[3 * i + c];
    in.r[c] = rays[3 * i + c];
  }
}
__device__ __forceinline__ void adam_clamp_core(int i, float t, double pw1, double pw2, const AdamIn &in, float *rays, bool have_grad_b, float scale_a,
                                                float *__restrict__ grad_out, float *__restrict__ m, float *__restrict__ v, double lr, double beta1, double beta2, double eps_d,
                                                const float *K, const float *I, float lo, float hi, int n_norm, float *kept) {
  // the scalars as torch forms them: in double from the Python floats, rounded to float where they meet the tensors
  const float b2 = (float)beta2, omb1 = (float)(1.0 - beta1), omb2 = (float)(1.0 - beta2), eps = (float)eps_d;
  const double bc1 = 1.0 - pw1, bc2 = 1.0 - pw2;
  const float step_size = (float)(lr / bc1), bc2s = (float)sqrt(bc2);
  float r[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float g = in.g[c];
    if (grad_out) { // g = grad / S + grad_b, rounded like the two torch ops it replaces; kept as the parameter's .grad
      if (scale_a != 1.0f) g = g / scale_a;
      if (have_grad_b) g = g + in.gb[c];
      grad_out[3 * i + c] = g;
    }
    float mm = in.m[c], vv = in.v[c];
    mm = mm + omb1 * (g - mm);
    vv = b2 * vv + omb2 * g * g;
    m[3 * i + c] = mm;
    v[3 * i + c] = vv;
    const float denom = sqrtf(vv) / bc2s + eps;
    r[c] = in.r[c] - step_size * mm / denom;
  }
  float x = r[0], y = r[1], z = r[2];
  const float q0 = K[0] * x + K[1] * y + K[2] * z + K[3];
  const float q1 = K[4] * x + K[5] * y + K[6] * z + K[7];
  const float q2 = K[8] * x + K[9] * y + K[10] * z + K[11];
  const float q3 = K[12] * x + K[13] * y + K[14] * z + K[15];
  float px = q0 / q3, py = q1 / q3, pz = q2 / q3;
  px = fminf(fmaxf(px, lo), hi);
  py = fminf(fmaxf(py, lo), hi);
  const float w0 = I[0] * px + I[1] * py + I[2] * pz + I[3];
  const float w1 = I[4] * px + I[5] * py + I[6] * pz + I[7];
  const float w2 = I[8] * px + I[9] * py + I[10] * pz + I[11];
  const float w3 = I[12] * px + I[13] * py + I[14] * pz + I[15];
  x = w0 / w3; y = w1 / w3; z = w2 / w3;
  for (int k = 0; k < n_norm; ++k) {
    const float nrm = sqrtf(x * x + y * y + z * z);
    x /= nrm; y /= nrm; z /= nrm;
  }
  rays[3 * i] = x; rays[3 * i + 1] = y; rays[3 * i + 2] = z;
  if (kept) { kept[3 * i] = x; kept[3 * i + 1] = y; kept[3 * i + 2] = z; }
}
__device__ __forceinline__ void adam_clamp_one(int i, float t, double pw1, double pw2, float *rays, const float *__restrict__ grad, const float *__restrict__ grad_b, float scale_a,
                                               float *__restrict__ grad_out, float *__restrict__ m, float *__restrict__ v, double lr, double beta1, double beta2, double eps_d,
                                               const float *K, const float *I, float lo, float hi, int n_norm) {
  AdamIn in;
  adam_load(i, rays, grad, grad_b, m, v, in);
  adam_clamp_core(i, t, pw1, pw2, in, rays, grad_b != nullptr, scale_a, grad_out, m, v, lr, beta1, beta2, eps_d, K, I, lo, hi, n_norm, nullptr);
}
__device__ __forceinline__ void adam_clamp_one(int i, float t, float *rays, const float *__restrict__ grad, const float *__restrict__ grad_b, float scale_a,
                                               float *__restrict__ grad_out, float *__restrict__ m, float *__restrict__ v, double lr, double beta1, double beta2, double eps_d,
                                               const float *K, const float *I, float lo, float hi, int n_norm) {
  adam_clamp_one(i, t, pow(beta1, (double)t), pow(beta2, (double)t), rays, grad, grad_b, scale_a, grad_out, m, v, lr, beta1, beta2, eps_d, K, I, lo, hi, n_norm);
}

// =================================================================================== fused pattern side of an optimisation step
// The pattern has 64..1024 points and the texture 500^2 texels: every kernel of the pattern side is launch-bound,
// and as separate entry points one gradient step issued ~30 of them (~0.11 ms next to a 0.75 ms render).  Three
// launches do the same arithmetic in the same order:
//   k_pattern_fwd   K1 + K2(sum) + K2(softor) (+ the partial sums of the overlap regulariser L1(softor, sum))
//   k_pattern_bwd   K2-bwd of the data term, of the regulariser's softor and sum terms, and K1-bwd of both
//   k_adam_clamp    Adam + Laser.clamp_to_fov + normalize_rays
__device__ __forceinline__ void project_xyz(const float x, const float y, const float z, const float *K, float &p0, float &p1) {
  const float q0 = K[0] * x + K[1] * y + K[2] * z + K[3];
  const float q1 = K[4] * x + K[5] * y + K[6] * z + K[7];
  const float q3 = K[12] * x + K[13] * y + K[14] * z + K[15];
  p0 = q0 / q3;
  p1 = q1 / q3;
}
__device__ __forceinline__ void project_xy(const float *rays, int k, const float *K, float &p0, float &p1) {
  const float x = rays[3 * k], y = rays[3 * k + 1], z = rays[3 * k + 2];
  const float q0 = K[0] * x + K[1] * y + K[2] * z + K[3];
  const float q1 = K[4] * x + K[5] * y + K[6] * z + K[7];
  const float q3 = K[12] * x + K[13] * y + K[14] * z + K[15];
  p0 = q0 / q3;
  p1 = q1 / q3;
}

// one workgroup per 32x8 texel tile, like k_splat_fused_fwd (ordered compaction of the points that reach the tile:
// every texel adds / multiplies in ascending point order); each texel keeps BOTH reductions of the same splat values.
__global__ void __launch_bounds__(SPLAT_BLOCK)
    k_pattern_fwd(const float *__restrict__ rays, int n, Mat4 KF, float sigma, int size0, int size1, int want_softor, float *__restrict__ pts,
                  float *__restrict__ tsum, float *__restrict__ tsor, float *__restrict__ ws, float *__restrict__ zero, long n_zero) {
  __shared__ float c_p0[CAND_MAX], c_p1[CAND_MAX];
  __shared__ int c_count;
  __shared__ float s_part[SPLAT_BLOCK / 64];
  const int tid = threadIdx.x;
  if (zero) { // the step's accumulation buffer (texture gradient + loss), cleared by the launch that opens the step instead of a fill of its own
    const long stride = (long)gridDim.x * gridDim.y * SPLAT_BLOCK;
    for (long t = ((long)blockIdx.y * gridDim.x + blockIdx.x) * SPLAT_BLOCK + tid; t < n_zero; t += stride) zero[t] = 0.f;
  }
  const int j0 = blockIdx.x * TILE_W, i0 = blockIdx.y * TILE_H;
  const int j = j0 + (tid % TILE_W), i = i0 + (tid / TILE_W);
  const float inv_sigma = 1.0f / sigma;
  float acc_s = 0.f, acc_p = 1.f;
  const bool first = blockIdx.x == 0 && blockIdx.y == 0;
  for (int chunk = 0; chunk < n; chunk += CAND_MAX) {
    if (tid < 64) {
      int count = 0;
      const int lim = min(CAND_MAX, n - chunk);
      for (int base = 0; base < lim; base += 64) {
        const int k = chunk + base + tid;
        bool keep = false;
        float p0s = 0.f, p1s = 0.f;
        if (base + tid < lim) {
          float p0, p1;
          project_xy(rays, k, KF.m, p0, p1);
          if (first) { pts[2 * k] = p0; pts[2 * k + 1] = p1; }
          p0s = p0 * (float)size0;
          p1s = p1 * (float)size1;
          const float dx = fmaxf(fmaxf((float)j0 - p0s, p0s - (float)(j0 + TILE_W - 1)), 0.f);
          const float dy = fmaxf(fmaxf((float)i0 - p1s, p1s - (float)(i0 + TILE_H - 1)), 0.f);
          keep = (dx * dx + dy * dy) * inv_sigma <= FFX_QCUT;
        }
        const unsigned long long m = __ballot(keep);
        const int pos = count + __popcll(m & ((1ull << tid) - 1ull));
        if (keep) { c_p0[pos] = p0s; c_p1[pos] = p1s; }
        count += __popcll(m);
      }
      if (tid == 0) c_count = count;
    }
    __syncthreads();
    const int cnt = c_count;
    if (j < size0 && i < size1) {
      for (int c = 0; c < cnt; ++c) {
        float d, yd, xd;
        const float v = splat_val((float)j, (float)i, c_p0[c], c_p1[c], sigma, inv_sigma, d, yd, xd);
        acc_s += v;
        acc_p *= (1.0f - v);
      }
    }
    __syncthreads();
  }
  float l1 = 0.f;
  if (j < size0 && i < size1) {
    tsum[(size_t)i * size0 + j] = acc_s;
    if (want_softor) {
      const float so = 1.0f - acc_p;
      tsor[(size_t)i * size0 + j] = so;
      l1 = fabsf(so - acc_s);
    }
  }
  if (want_softor) { // partial sum of |softor - sum| of this tile, fixed order
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) l1 += __shfl_down(l1, o, 64);
    if ((tid & 63) == 0) s_part[tid >> 6] = l1;
    __syncthreads();
    if (tid == 0) ws[blockIdx.y * gridDim.x + blockIdx.x] = (s_part[0] + s_part[1]) + (s_part[2] + s_part[3]);
  }
}

// k_pattern_fwd that also writes tex = K3(tsum), the (2R+1)^2 Gaussian blur with reflected borders (ffx_pattern_fwd_blur): the
// workgroup evaluates the splat sums of its 32x8 tile PLUS the halo of R texels around it (at the reflected positions where the
// halo leaves the image: exactly what k_blur_fwd stages) into LDS — 36x12 instead of 32x8 evaluations at R = 2 — and blurs from
// there.  Every value is formed as the separate launches form it (same candidate order per texel: points that cannot reach a
// texel add an exact 0; same thread-to-texel mapping for the regulariser's partial sums and for the blur), so tsum, tsor, ws
// and tex are bitwise those of ffx_pattern_fwd + ffx_blur_fwd; one launch (~5 us on the critical path of a step) less.
// (the tile (bx, by) of a grid nbx tiles wide: the body of k_pattern_fwd_blur, shared with k_pattern_step — `rays` carries no __restrict__: in
// k_pattern_step the same launch has written it)
template <int R>
__device__ __forceinline__ void pattern_fwd_blur_tile(int bx, int by, int nbx, const float *rays, int n, const Mat4 &KF, float sigma, int size0, int size1, int want_softor,
                                                      float *__restrict__ pts, float *__restrict__ tsum, float *__restrict__ tsor, float *__restrict__ ws, const BlurW &bw,
                                                      float *__restrict__ tex) {
  constexpr int HW = TILE_W + 2 * R, HH = TILE_H + 2 * R, NT = (HW * HH + SPLAT_BLOCK - 1) / SPLAT_BLOCK;
  __shared__ float c_p0[CAND_MAX], c_p1[CAND_MAX];
  __shared__ int c_count;
  __shared__ float s_part[SPLAT_BLOCK / 64];
  __shared__ float s_t[HW * HH];                 // tsum over the haloed tile (the blur's input tile)
  __shared__ float s_o[TILE_W * TILE_H];         // tsor of the tile itself
  const int tid = threadIdx.x;
  const int j0 = bx * TILE_W, i0 = by * TILE_H;
  const float inv_sigma = 1.0f / sigma;
  const bool first = bx == 0 && by == 0;
  // the texels of this thread: haloed-tile elements tid, tid + 256, ... at their reflected image positions
  float fj[NT], fi[NT], acc_s[NT], acc_p[NT];
#pragma unroll
  for (int u = 0; u < NT; ++u) {
    const int t = tid + u * SPLAT_BLOCK;
    const int lx = t % HW, ly = t / HW;
    fj[u] = (float)reflect_idx(j0 + lx - R, size0);
    fi[u] = (float)reflect_idx(i0 + ly - R, size1);
    acc_s[u] = 0.f;
    acc_p[u] = 1.f;
  }
  for (int chunk = 0; chunk < n; chunk += CAND_MAX) {
    if (tid < 64) {
      int count = 0;
      const int lim = min(CAND_MAX, n - chunk);
      for (int base = 0; base < lim; base += 64) {
        const int k = chunk + base + tid;
        bool keep = false;
        float p0s = 0.f, p1s = 0.f;
        if (base + tid < lim) {
          float p0, p1;
          project_xy(rays, k, KF.m, p0, p1);
          if (first) { pts[2 * k] = p0; pts[2 * k + 1] = p1; }
          p0s = p0 * (float)size0;
          p1s = p1 * (float)size1;
          const float dx = fmaxf(fmaxf((float)(j0 - R) - p0s, p0s - (float)(j0 + TILE_W - 1 + R)), 0.f); // distance to the HALOED tile
          const float dy = fmaxf(fmaxf((float)(i0 - R) - p1s, p1s - (float)(i0 + TILE_H - 1 + R)), 0.f);
          keep = (dx * dx + dy * dy) * inv_sigma <= FFX_QCUT;
        }
        const unsigned long long m = __ballot(keep);
        const int pos = count + __popcll(m & ((1ull << tid) - 1ull));
        if (keep) { c_p0[pos] = p0s; c_p1[pos] = p1s; }
        count += __popcll(m);
      }
      if (tid == 0) c_count = count;
    }
    __syncthreads();
    const int cnt = c_count;
#pragma unroll
    for (int u = 0; u < NT; ++u) {
      if (tid + u * SPLAT_BLOCK >= HW * HH) continue;
      for (int c = 0; c < cnt; ++c) {
        float d, yd, xd;
        const float v = splat_val(fj[u], fi[u], c_p0[c], c_p1[c], sigma, inv_sigma, d, yd, xd);
        acc_s[u] += v;
        acc_p[u] *= (1.0f - v);
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int u = 0; u < NT; ++u) {
    const int t = tid + u * SPLAT_BLOCK;
    if (t >= HW * HH) continue;
    s_t[t] = acc_s[u];
    const int lx = t % HW - R, ly = t / HW - R; // position inside the tile itself
    if (lx >= 0 && lx < TILE_W && ly >= 0 && ly < TILE_H) s_o[ly * TILE_W + lx] = 1.0f - acc_p[u];
  }
  __syncthreads();
  // the tile's own texels, one per thread as in k_pattern_fwd / k_blur_fwd
  const int lx = tid % TILE_W, ly = tid / TILE_W;
  const int j = j0 + lx, i = i0 + ly;
  const bool inside = j < size0 && i < size1;
  float l1 = 0.f;
  if (inside) {
    const float a = s_t[(ly + R) * HW + lx + R];
    tsum[(size_t)i * size0 + j] = a;
    if (want_softor) {
      const float so = s_o[tid];
      tsor[(size_t)i * size0 + j] = so;
      l1 = fabsf(so - a);
    }
    float acc = 0.f;
#pragma unroll
    for (int ky = 0; ky < 2 * R + 1; ++ky) {
      float row = 0.f;
#pragma unroll
      for (int kx = 0; kx < 2 * R + 1; ++kx) row = fmaf(bw.w[kx], s_t[(ly + ky) * HW + lx + kx], row);
      acc = fmaf(bw.w[ky], row, acc);
    }
    tex[(size_t)i * size0 + j] = acc;
  }
  if (want_softor) { // partial sum of |softor - sum| of this tile, fixed order
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) l1 += __shfl_down(l1, o, 64);
    if ((tid & 63) == 0) s_part[tid >> 6] = l1;
    __syncthreads();
    if (tid == 0) ws[by * nbx + bx] = (s_part[0] + s_part[1]) + (s_part[2] + s_part[3]);
  }
}
template <int R>
__global__ void __launch_bounds__(SPLAT_BLOCK)
    k_pattern_fwd_blur(const float *__restrict__ rays, int n, Mat4 KF, float sigma, int size0, int size1, int want_softor, float *__restrict__ pts,
                       float *__restrict__ tsum, float *__restrict__ tsor, float *__restrict__ ws, float *__restrict__ zero, long n_zero, BlurW bw,
                       float *__restrict__ tex) {
  if (zero) {
    const long stride = (long)gridDim.x * gridDim.y * SPLAT_BLOCK;
    for (long t = ((long)blockIdx.y * gridDim.x + blockIdx.x) * SPLAT_BLOCK + threadIdx.x; t < n_zero; t += stride) zero[t] = 0.f;
  }
  pattern_fwd_blur_tile<R>(blockIdx.x, blockIdx.y, gridDim.x, rays, n, KF, sigma, size0, size1, want_softor, pts, tsum, tsor, ws, bw, tex);
}

// one workgroup per point over its non-zero footprint, like k_splat_bwd.  Data term: upstream gts on the SUM
// texture.  Regulariser w * mean|softor - sum|: upstream gd = w * sign(softor - sum) / T on softor and -gd on
// sum, i.e. per texel gd * (prod_{m != k}(1 - v_m) - 1) on this point's splat value.  fp64 partial sums; then
// K1-bwd of both results (the chain through the perspective divide) by one lane.
// KS >= 0 (ffx_pattern_bwd_blur): gts is the gradient on the BLURRED texture and K3^T is applied here, over the point's footprint
// only — the footprint plus a halo of r texels is staged in (dynamic) LDS and every texel's value is formed by blur_bwd_texel, the
// code k_blur_bwd runs: bitwise the separate launches' gradient without the 250 000-texel transpose blur in front (11 us) —
// and `adam` (if it names rays): the workgroup that finishes last applies ffx_adam_clamp_step's update to every point.
struct AdamK { float *rays, *m, *v, *step, *grad_out; unsigned int *counter; double lr, beta1, beta2, eps; Mat4 KI; float lo, hi, grad_div; int n_norm;
               const float *dot_a, *dot_b; long dot_n; float *dot_partial; long dot_b_n; int no_update; const unsigned int *guard; };
// (point k of n: the gradient part of k_pattern_bwd — shared with k_pattern_step, where `rays` is written later in the same launch: no
// __restrict__ on it.  s_win: the launch's dynamic LDS, KS >= 0: gts over the footprint + halo)
template <int KS>
__device__ __forceinline__ void pattern_bwd_point(const int k, const float *rays, int n, const Mat4 &KF, float sigma, int size0, int size1, const float *__restrict__ tsum,
                                                  const float *__restrict__ tsor, const float *__restrict__ gts, float reg_weight, const float *__restrict__ ws, int n_ws,
                                                  float *__restrict__ grays_data, float *__restrict__ grays_reg, float *__restrict__ reg_value,
                                                  const float *__restrict__ loss_in, int loss_in_n, float loss_div, const BlurW &bw, float *s_win,
                                                  unsigned long long *stamps = nullptr) {
#ifdef FFX_PATSTAMP
#define PB_STAMP(slot) do { if (stamps && threadIdx.x == 0) stamps[slot] = wall_clock64(); } while (0)
#else
#define PB_STAMP(slot) do { } while (0)
#endif
  __shared__ float nb_p0[NEIGH_MAX], nb_p1[NEIGH_MAX];
  __shared__ int nb_count;
  __shared__ double red[4][SPLAT_BLOCK / 64];
  const int tid = threadIdx.x;
  const float inv_sigma = 1.0f / sigma;
  const bool reg = reg_weight > 0.f && tsor != nullptr;
  const bool use_list = reg && n <= NEIGH_MAX;
  // (round 6: the loads of this workgroup's dependent chain are asked for as early as their addresses are known — this point's ray together with
  // the first 64 candidate neighbours' rays here, the regulariser's first texture values together with the window of gts below: four memory
  // latencies in a row were most of a 6 us workgroup.  Same arithmetic in the same order.)
  float p0, p1;
  const float kx = rays[3 * k], ky = rays[3 * k + 1], kz = rays[3 * k + 2];
  float c_x = 0.f, c_y = 0.f, c_z = 0.f;
  if (use_list && tid < 64 && tid < n) { c_x = rays[3 * tid]; c_y = rays[3 * tid + 1]; c_z = rays[3 * tid + 2]; }
  project_xyz(kx, ky, kz, KF.m, p0, p1);
  const float p0s = p0 * (float)size0, p1s = p1 * (float)size1;
  const float R = sqrtf(FFX_QCUT * sigma) + 1.0f;
  const int lo0 = max(0, (int)floorf(p0s - R)), hi0 = min(size0, (int)ceilf(p0s + R) + 1);
  const int lo1 = max(0, (int)floorf(p1s - R)), hi1 = min(size1, (int)ceilf(p1s + R) + 1);
  const bool alive = hi0 > lo0 && hi1 > lo1;
  float n_so = 0.f, n_su = 0.f; // the regulariser's texture values of the first round of the texel loop
  if (alive && reg && tid < (hi0 - lo0) * (hi1 - lo1)) {
    const size_t Tn = (size_t)(lo1 + tid / (hi0 - lo0)) * size0 + (lo0 + tid % (hi0 - lo0));
    n_so = tsor[Tn]; n_su = tsum[Tn];
  }
  // the window of gts (KS >= 0: footprint + halo, for K3^T) is staged by waves 1..3 WHILE wave 0 compacts the neighbour list (round 6: one barrier
  // and one memory latency instead of two of each); without a list all four waves stage
  const int br = KS >= 0 ? (KS ? KS : bw.ksize) / 2 : 0;
  const int ww = hi0 - lo0 + 2 * br; // row length of the staged window
  if constexpr (KS >= 0) {
    if (alive && gts) {
      const int wh = hi1 - lo1 + 2 * br;
      const int first = use_list ? 64 : 0;
      if (tid >= first)
        for (int t = tid - first; t < ww * wh; t += SPLAT_BLOCK - first) {
          const int gy = lo1 - br + t / ww, gx = lo0 - br + t % ww;
          s_win[t] = (gy >= 0 && gy < size1 && gx >= 0 && gx < size0) ? gts[(size_t)gy * size0 + gx] : 0.f;
        }
    }
  }
  if (use_list) {
    if (tid < 64) {
      int count = 0;
      for (int base = 0; base < n; base += 64) {
        const int m = base + tid;
        bool keep = false;
        float q0 = 0.f, q1 = 0.f;
        if (m < n && m != k) {
          float a, b;
          if (base == 0) project_xyz(c_x, c_y, c_z, KF.m, a, b);
          else project_xy(rays, m, KF.m, a, b);
          q0 = a * (float)size0;
          q1 = b * (float)size1;
          keep = fabsf(q0 - p0s) <= 2.f * R + 2.f && fabsf(q1 - p1s) <= 2.f * R + 2.f;
        }
        const unsigned long long mk = __ballot(keep);
        const int pos = count + __popcll(mk & ((1ull << tid) - 1ull));
        if (keep) { nb_p0[pos] = q0; nb_p1[pos] = q1; }
        count += __popcll(mk);
      }
      if (tid == 0) nb_count = count;
    }
  }
  if (use_list || KS >= 0) __syncthreads();
  PB_STAMP(10);
  const float gscale = reg_weight / ((float)size0 * (float)size1);
  double a0 = 0.0, a1 = 0.0, b0 = 0.0, b1 = 0.0;
  if (alive) {
    const int rw = hi0 - lo0, rh = hi1 - lo1;
    // (the regulariser's two texture values of the NEXT round are asked for at the top of this one — the first round's: above)
    for (int t = tid; t < rw * rh; t += SPLAT_BLOCK) {
      const float c_so = n_so, c_su = n_su;
      if (reg && t + SPLAT_BLOCK < rw * rh) {
        const int tn = t + SPLAT_BLOCK;
        const size_t Tn = (size_t)(lo1 + tn / rw) * size0 + (lo0 + tn % rw);
        n_so = tsor[Tn]; n_su = tsum[Tn];
      }
      const int j = lo0 + t % rw, i = lo1 + t / rw;
      float d, yd, xd;
      const float v = splat_val((float)j, (float)i, p0s, p1s, sigma, inv_sigma, d, yd, xd);
      if (v == 0.f) continue;
      const size_t T = (size_t)i * size0 + j;
      const float cf = splat_gcoef(v, d, sigma);
      if (gts) {
        float w;
        if constexpr (KS >= 0) w = blur_bwd_texel<(KS > 0 ? KS : 0)>(s_win, ww, lo1 - br, lo0 - br, i, j, size1, size0, bw);
        else w = gts[T];
        a0 += (double)(w * (cf * yd));
        a1 += (double)(w * (cf * xd));
      }
      if (reg) {
        const float df = c_so - c_su;
        const float gd = df > 0.f ? gscale : (df < 0.f ? -gscale : 0.f);
        if (gd != 0.f) {
          float prod = 1.f;
          const int cnt = use_list ? nb_count : n;
          for (int c = 0; c < cnt; ++c) {
            float q0, q1;
            if (use_list) { q0 = nb_p0[c]; q1 = nb_p1[c]; }
            else {
              if (c == k) continue;
              float a, b;
              project_xy(rays, c, KF.m, a, b);
              q0 = a * (float)size0; q1 = b * (float)size1;
            }
            float dd, y2, x2;
            prod *= (1.0f - splat_val((float)j, (float)i, q0, q1, sigma, inv_sigma, dd, y2, x2));
          }
          // softor term (gd * prod) and sum term (-gd), each rounded like the separate kernels, summed in fp64
          b0 += (double)((gd * prod) * (cf * yd)) - (double)(gd * (cf * yd));
          b1 += (double)((gd * prod) * (cf * xd)) - (double)(gd * (cf * xd));
        }
      }
    }
  }
  PB_STAMP(11);
  a0 = wave_sum(a0); a1 = wave_sum(a1); b0 = wave_sum(b0); b1 = wave_sum(b1);
  if ((tid & 63) == 0) { red[0][tid >> 6] = a0; red[1][tid >> 6] = a1; red[2][tid >> 6] = b0; red[3][tid >> 6] = b1; }
  __syncthreads();
  if (tid == 0) {
    double sres[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int c = 0; c < 4; ++c)
      for (int w = 0; w < SPLAT_BLOCK / 64; ++w) sres[c] += red[c][w];
    // K1-bwd (k_project_bwd) with gpts = (g0, g1, 0)
    const float x = kx, y = ky, z = kz;
    const float *K = KF.m;
    float q[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) q[r] = K[4 * r] * x + K[4 * r + 1] * y + K[4 * r + 2] * z + K[4 * r + 3];
    const float iw = 1.0f / q[3];
#pragma unroll
    for (int part = 0; part < 2; ++part) {
      float *out = part == 0 ? grays_data : grays_reg;
      if (!out) continue;
      const float g0 = (float)sres[2 * part] * (float)size0, g1 = (float)sres[2 * part + 1] * (float)size1;
      float gq[4];
      gq[0] = g0 * iw;
      gq[1] = g1 * iw;
      gq[2] = 0.f * iw;
      float acc = 0.f;
      acc += g0 * q[0];
      acc += g1 * q[1];
      acc += 0.f * q[2];
      gq[3] = -acc * iw * iw;
#pragma unroll
      for (int c = 0; c < 3; ++c) out[3 * k + c] = gq[0] * K[c] + gq[1] * K[4 + c] + gq[2] * K[8 + c] + gq[3] * K[12 + c];
    }
  }
  PB_STAMP(12);
  if (k == 0 && reg_value) { // value of the regulariser from the forward's per-tile partial sums (fixed order)
    float acc = 0.f;
    if (reg && ws) // (no regulariser: the forward wrote no partial sums and ws may be NULL)
      for (int t = tid; t < n_ws; t += SPLAT_BLOCK) acc += ws[t];
    // the data term's partial sums (ffx_render_bwd_cached's dot slots, or one value), fixed order too
    float lacc = 0.f;
    if (loss_in)
      for (int t = tid; t < loss_in_n; t += SPLAT_BLOCK) lacc += loss_in[t];
    __shared__ float s_v[SPLAT_BLOCK / 64], s_l[SPLAT_BLOCK / 64];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { acc += __shfl_down(acc, o, 64); lacc += __shfl_down(lacc, o, 64); }
    if ((tid & 63) == 0) { s_v[tid >> 6] = acc; s_l[tid >> 6] = lacc; }
    __syncthreads();
    if (tid == 0) {
      const float rv = reg ? ((s_v[0] + s_v[1]) + (s_v[2] + s_v[3])) * gscale : 0.f;
      reg_value[0] = rv;
      if (loss_in) {
        const float ls = (s_l[0] + s_l[1]) + (s_l[2] + s_l[3]);
        reg_value[1] = ls / loss_div + rv; // the step's total loss: data term / S + regulariser
        reg_value[2] = ls;                 // the data term as accumulated (what a multi-rank step exchanges)
      }
    }
  }
}
// slice k of n_slices of <dot_a, dot_b> (the data term of a loss that is linear in the image) -> dot_partial[k]
__device__ __forceinline__ void pattern_dot_slice(const int k, const int n_slices, const AdamK &adam) {
  const int tid = threadIdx.x;
  {
    __shared__ double s_dot[SPLAT_BLOCK / 64];
    const long per = ((adam.dot_n + n_slices - 1) / n_slices + 3) & ~3L; // (slices of whole float4s)
    const long lo_i = min(adam.dot_n, per * k), hi_i = min(adam.dot_n, lo_i + per); // (workgroups past the end: an empty slice)
    double acc = 0.0;
    const long bn = adam.dot_b_n; // period of b (a whole number of float4s for the vector path: a quad of a never straddles two periods)
    const bool vec = ((((uintptr_t)adam.dot_a | (uintptr_t)adam.dot_b) & 15) == 0) && (bn & 3) == 0;
    if (vec) {
      for (long i = lo_i + 4 * tid; i + 3 < hi_i; i += 4 * SPLAT_BLOCK) {
        const float4 a = *reinterpret_cast<const float4 *>(adam.dot_a + i), b = *reinterpret_cast<const float4 *>(adam.dot_b + i % bn);
        acc += (double)(a.x * b.x) + (double)(a.y * b.y) + (double)(a.z * b.z) + (double)(a.w * b.w);
      }
      for (long i = lo_i + ((hi_i - lo_i) & ~3L) + tid; i < hi_i; i += SPLAT_BLOCK) acc += (double)(adam.dot_a[i] * adam.dot_b[i % bn]);
    } else {
      for (long i = lo_i + tid; i < hi_i; i += SPLAT_BLOCK) acc += (double)(adam.dot_a[i] * adam.dot_b[i % bn]);
    }
    acc = wave_sum(acc);
    if ((tid & 63) == 0) s_dot[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) adam.dot_partial[k] = (float)((s_dot[0] + s_dot[1]) + (s_dot[2] + s_dot[3]));
  }
}
// the step's data term from the slices, in slice order -> reg_value[1] (total loss: data term / loss_div + regulariser), reg_value[2]
__device__ __forceinline__ void pattern_dot_total(const int n_slices, const AdamK &adam, float *reg_value, float loss_div) {
  __shared__ double s_tot[SPLAT_BLOCK / 64];
  const int tid = threadIdx.x;
  double tot = 0.0;
  for (int i = tid; i < n_slices; i += SPLAT_BLOCK) tot += (double)adam.dot_partial[i];
  tot = wave_sum(tot);
  if ((tid & 63) == 0) s_tot[tid >> 6] = tot;
  __syncthreads();
  if (tid == 0) {
    const float ls = (float)((s_tot[0] + s_tot[1]) + (s_tot[2] + s_tot[3]));
    reg_value[1] = ls / loss_div + reg_value[0]; // (reg_value[0]: written by point 0's workgroup before it arrived at the counter)
    reg_value[2] = ls;
  }
}
template <int KS>
__global__ void __launch_bounds__(SPLAT_BLOCK)
    k_pattern_bwd(const float *__restrict__ rays, int n, Mat4 KF, float sigma, int size0, int size1, const float *__restrict__ tsum,
                  const float *__restrict__ tsor, const float *__restrict__ gts, float reg_weight, const float *__restrict__ ws, int n_ws,
                  float *__restrict__ grays_data, float *__restrict__ grays_reg, float *__restrict__ reg_value, const float *__restrict__ loss_in, int loss_in_n, float loss_div,
                  BlurW bw, AdamK adam) {
  extern __shared__ float s_win[]; // KS >= 0: gts over the footprint + halo
  const int k = blockIdx.x, tid = threadIdx.x;
  pattern_bwd_point<KS>(k, rays, n, KF, sigma, size0, size1, tsum, tsor, gts, reg_weight, ws, n_ws, grays_data, grays_reg, reg_value, loss_in, loss_in_n, loss_div, bw, s_win);
  if (adam.rays && adam.dot_a) pattern_dot_slice(k, (int)gridDim.x, adam);
  if (adam.rays) { // the update rides along: whoever finishes last sees every point's gradient (agent-scope fence + counter)
    __shared__ int s_last;
    __syncthreads();
    if (tid == 0) {
      __threadfence();
      s_last = atomicAdd(adam.counter, 1u) == gridDim.x - 1u;
    }
    __syncthreads();
    if (s_last) {
      __threadfence();
      // (ffx_adam_args.guard: an adjoint-cache header whose `dropped` word says the gradient of this step is incomplete — K9 has poisoned
      // it with NaN: the update is then NOT applied, so that rays and the optimiser state stay what they were)
      const bool skip = adam.no_update || (adam.guard && adam.guard[2] != 0u);
      const float t = skip ? 0.f : adam.step[0] + 1.0f;
      if (!skip)
        for (int i = tid; i < n; i += SPLAT_BLOCK)
          adam_clamp_one(i, t, adam.rays, grays_data, grays_reg, adam.grad_div, adam.grad_out, adam.m, adam.v, adam.lr, adam.beta1, adam.beta2, adam.eps, KF.m, adam.KI.m,
                         adam.lo, adam.hi, adam.n_norm);
      if (adam.dot_a && reg_value) pattern_dot_total((int)gridDim.x, adam, reg_value, loss_div); // the data term: the slices in workgroup order
      __syncthreads();
      if (tid == 0) { if (!skip) adam.step[0] = t; *adam.counter = 0u; } // (the counter is ready for the next launch)
    }
  }
}

// ------------------------------------------------------------------------------- the pattern side of a step as ONE launch
// k_pattern_step (ffx_pattern_step, round 6): k_pattern_bwd<KS> of step s — gradient, Adam, clamp_to_fov, the step's loss — AND
// k_pattern_fwd_blur<2> of step s + 1 (K1 + K2 + K3 on the updated pattern, the accumulator cleared) in one launch: the launch boundary between
// them (a drained GPU behind a 17 us kernel of 64 workgroups) and the serial tail of the gradient launch (every workgroup's slice of the data
// term behind its own point, two double-precision pow() in front of the update) leave the critical path between two renders.
//
//   workgroups 0 .. n-1     the gradient of point blockIdx.x (pattern_bwd_point); the one that arrives last applies the update to all points, copies
//                           the guard's header (the same launch clears it below) and publishes `go` = this launch's epoch.  They wait for nobody.
//   workgroups n .. G-1     helpers: slices of the data term's inner product while the gradient runs, then WAIT for `go`, then a forward tile each
//                           and the clearing of `zero`.
// What same-address device-scope atomics cost on this part decides the shape (measured here: ~27 ns each, one after the other — they execute where
// the eight XCDs meet): a ticket per workgroup to deal the roles in start order took 29 us for 1072 workgroups, a "who leaves last" counter another 29,
// a thousand helpers polling ONE flag kept the flag's line busy for longer than the wait.  Hence: roles by blockIdx (workgroups are dispatched in
// index order, per XCD as well: a helper only ever waits for workgroups dispatched before it; the wait is BOUNDED all the same — a helper that has
// polled for ~0.2 s raises `timeout` and leaves its tile undone, the host raises: never a hang), no exit counter (the flags carry the launch's epoch
// instead of being re-armed; the counters are re-armed by the workgroup that completes them), 64 copies of the flag on lines of their own, and a
// poll that is a read-modify-write (a plain load is answered by the XCD's own L2, possibly with the line from before: 130 us until it happened to be
// evicted; an acquire per poll invalidates that L2 at the polling rate: the gradient then misses on every load, 340 us).
#define PAT_GO_COPIES 64
struct PatSync {
  unsigned int pad0[4], stale, timeout, pad1[2]; double pw_t, pw_b1, pw_b2, pw1, pw2; unsigned int hdr[16]; unsigned int pad2[256 - 34]; // bytes 0..1023
  unsigned int arrived, pad3[255];
  unsigned int fin, pad4[255];
  struct { unsigned int v, pad[127]; } go[PAT_GO_COPIES];
};
static_assert(sizeof(PatSync) == FFX_PATTERN_SYNC_BYTES && offsetof(PatSync, stale) == 16 && offsetof(PatSync, timeout) == 20 && offsetof(PatSync, hdr) == 72,
              "ffx.h: FFX_PATTERN_SYNC_BYTES and the documented offsets");
struct FwdK { float *pts, *tsum, *tsor, *ws, *zero, *tex, *kept_new; const float *kept_old; long n_zero; int want_softor, nbx, nby, check_kept, pow_cache; unsigned int epoch; };
// the data term is complete when its slices AND the regulariser's value (point 0's workgroup) are: whoever of the slice helpers and the updating
// workgroup gets here last adds the slices up
__device__ __forceinline__ void pattern_step_fin(PatSync *sy, unsigned int participants, int n, const AdamK &adam, float *reg_value, float loss_div) {
  __shared__ int s_fin;
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    s_fin = atomicAdd(&sy->fin, 1u) == participants - 1u;
  }
  __syncthreads();
  if (s_fin) {
    __threadfence();
    pattern_dot_total(n, adam, reg_value, loss_div);
    if (threadIdx.x == 0) sy->fin = 0u;
  }
}
// -DFFX_PATSTAMP (tools/patstamp.py): s_memrealtime (100 MHz) at the launch's milestones, into the unused words of the sync block's first KB
#ifdef FFX_PATSTAMP
#define PAT_STAMP(slot) do { if (threadIdx.x == 0) ((unsigned long long *)sy->pad2)[slot] = wall_clock64(); } while (0)
#else
#define PAT_STAMP(slot) do { } while (0)
#endif
template <int KS>
__global__ void __launch_bounds__(SPLAT_BLOCK)
    k_pattern_step(float *rays, int n, Mat4 KF, float sigma, int size0, int size1, const float *tsum, const float *tsor, const float *gts, float reg_weight, const float *ws,
                   int n_ws, float *__restrict__ grays_data, float *__restrict__ grays_reg, float *__restrict__ reg_value, const float *__restrict__ loss_in, int loss_in_n,
                   float loss_div, BlurW bw, AdamK adam, FwdK fw, PatSync *sy) {
  extern __shared__ float s_win[];
  __shared__ int s_flag;
  const int tid = threadIdx.x;
  const unsigned int G = gridDim.x, H = G - (unsigned int)n;                          // H helpers
  const unsigned int dot_helpers = adam.dot_a ? (H < (unsigned int)n ? H : (unsigned int)n) : 0u; // the first of them take the data term's n slices
  if (blockIdx.x < (unsigned int)n) {
    const int k = (int)blockIdx.x;
    if (k == 0) PAT_STAMP(1);
    if (fw.check_kept && tid < 3) { // the texture this step rendered with was made from rays_kept: has anybody edited the pattern since?
      if (__float_as_uint(rays[3 * k + tid]) != __float_as_uint(fw.kept_old[3 * k + tid])) atomicOr(&sy->stale, 1u);
    }
#ifdef FFX_PATSTAMP
    pattern_bwd_point<KS>(k, rays, n, KF, sigma, size0, size1, tsum, tsor, gts, reg_weight, ws, n_ws, grays_data, grays_reg, reg_value, loss_in, loss_in_n, loss_div, bw, s_win,
                          k == 1 ? (unsigned long long *)sy->pad2 : nullptr);
#else
    pattern_bwd_point<KS>(k, rays, n, KF, sigma, size0, size1, tsum, tsor, gts, reg_weight, ws, n_ws, grays_data, grays_reg, reg_value, loss_in, loss_in_n, loss_div, bw, s_win);
#endif
    __syncthreads();
    if (k == 1) PAT_STAMP(13);
    if (tid == 0) {
      __threadfence();
      if (k == 1) PAT_STAMP(14);
      s_flag = atomicAdd(&sy->arrived, 1u) == (unsigned int)n - 1u;
      if (k == 1) PAT_STAMP(15);
    }
    __syncthreads();
    if (!s_flag) return;
    // ---- every point's gradient is in memory: the update
    PAT_STAMP(2);
    __threadfence();
    PAT_STAMP(3);
    // (the first round's operands are asked for together with the words that decide whether the update is applied: one memory latency, not two)
    AdamIn in0;
    if (tid < n && !adam.no_update) adam_load(tid, adam.rays, grays_data, grays_reg, adam.m, adam.v, in0);
    const bool skip = adam.no_update || (adam.guard && adam.guard[2] != 0u);
    const float t = skip ? 0.f : adam.step[0] + 1.0f;
    if (adam.guard && tid < 16) sy->hdr[tid] = adam.guard[tid]; // (what _watch_cache reads: the forward part below clears the header itself)
    // The pattern the texture is made from goes to the half of rays_kept that NO workgroup of this launch has read: the helpers take it from there with
    // plain loads and WITHOUT an acquire fence — their XCD's L2 cannot hold that half (it does hold `rays` as they were: the gradient's workgroups
    // read them).  A fence per helper wave — 4 000 at the same moment, each a write-back + invalidate of a whole L2 — was 60 us of an 80 us launch.
    if (!skip) {
      // beta^t: the running products of the last launch when they belong to step t - 1 and to these betas (one multiply instead of two
      // double-precision pow() in this workgroup's serial tail); pow() otherwise
      double p1, p2;
      if (fw.pow_cache && sy->pw_t == (double)t - 1.0 && sy->pw_b1 == adam.beta1 && sy->pw_b2 == adam.beta2 && t > 1.0f) { p1 = sy->pw1 * adam.beta1; p2 = sy->pw2 * adam.beta2; }
      else { p1 = pow(adam.beta1, (double)t); p2 = pow(adam.beta2, (double)t); }
      for (int i = tid; i < n; i += SPLAT_BLOCK) {
        AdamIn in = in0;
        if (i != tid) adam_load(i, adam.rays, grays_data, grays_reg, adam.m, adam.v, in);
        adam_clamp_core(i, t, p1, p2, in, adam.rays, grays_reg != nullptr, adam.grad_div, adam.grad_out, adam.m, adam.v, adam.lr, adam.beta1, adam.beta2, adam.eps, KF.m, adam.KI.m,
                        adam.lo, adam.hi, adam.n_norm, fw.kept_new);
      }
      // (the step count and the running products are written BEHIND the barrier below: every wave of this workgroup reads them above — a wave that
      // read them late would take the next step's count)
      __syncthreads();
      if (tid == 0) { sy->pw_t = (double)t; sy->pw_b1 = adam.beta1; sy->pw_b2 = adam.beta2; sy->pw1 = p1; sy->pw2 = p2; adam.step[0] = t; }
    } else {
      for (int i = tid; i < 3 * n; i += SPLAT_BLOCK) fw.kept_new[i] = rays[i]; // (the pattern as it was)
    }
    PAT_STAMP(4);
    __syncthreads();
    if (tid < PAT_GO_COPIES) {
      if (tid == 0) sy->arrived = 0u; // (re-armed for the next launch: nobody else touches it any more)
      __threadfence();
      __hip_atomic_exchange(&sy->go[tid].v, fw.epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
    PAT_STAMP(5);
    if (dot_helpers && reg_value) pattern_step_fin(sy, dot_helpers + 1u, n, adam, reg_value, loss_div);
    return;
  }
  const unsigned int h = blockIdx.x - (unsigned int)n; // helper h of H
  if (h == H - 1u) PAT_STAMP(6);
  if (h < dot_helpers) {
    for (unsigned int sl = h; sl < (unsigned int)n; sl += H) { pattern_dot_slice((int)sl, n, adam); __syncthreads(); }
    if (reg_value) pattern_step_fin(sy, dot_helpers + 1u, n, adam, reg_value, loss_div);
  }
  if (tid == 0) {
    unsigned int *flag = &sy->go[h % PAT_GO_COPIES].v;
    if (h >= dot_helpers) __builtin_amdgcn_s_sleep(100); // (nothing to do meanwhile: the gradient takes 7 us at least, this is 2.7)
    int ok = 1;
    for (unsigned int polls = 0; __hip_atomic_fetch_add(flag, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != fw.epoch; ++polls) {
      if (polls > 600000u) { ok = 0; atomicOr(&sy->timeout, 1u); break; } // (~0.2 s: see above)
      __builtin_amdgcn_s_sleep(8);
    }
    s_flag = ok;
  }
  __syncthreads();
  if (!s_flag) return;
  asm volatile("" ::: "memory"); // (no fence: see kept_new above; the loads below are issued behind the poll that saw the flag)
  if (h == H - 1u) PAT_STAMP(7);
  if (fw.zero)
    for (long t = (long)h * SPLAT_BLOCK + tid; t < fw.n_zero; t += (long)H * SPLAT_BLOCK) fw.zero[t] = 0.f;
  const unsigned int n_tiles = (unsigned int)(fw.nbx * fw.nby);
  for (unsigned int tile = h; tile < n_tiles; tile += H) {
    pattern_fwd_blur_tile<2>((int)(tile % (unsigned int)fw.nbx), (int)(tile / (unsigned int)fw.nbx), fw.nbx, fw.kept_new, n, KF, sigma, size0, size1, fw.want_softor, fw.pts, fw.tsum,
                             fw.tsor, fw.ws, bw, fw.tex);
    __syncthreads();
  }
  if (h == H - 1u) PAT_STAMP(8);
}

// Adam (the arithmetic of torch.optim.Adam's fused kernel: lerp of the first moment, bias corrections from the
// step count kept on the device) followed by Laser.clamp_to_fov + normalize_rays on the updated ray: one launch.
__global__ void __launch_bounds__(256)
    k_adam_clamp(float *__restrict__ rays, const float *__restrict__ grad, const float *__restrict__ grad_b, float scale_a, float *__restrict__ grad_out,
                 float *__restrict__ m, float *__restrict__ v, float *__restrict__ step, int n, double lr, double beta1, double beta2, double eps_d, Mat4 KF, Mat4 KI, float lo, float hi, int n_norm,
                 const unsigned int *__restrict__ guard) {
  // (guard, ABI 7: word 2 of an adjoint-cache header — or, behind a multi-rank exchange, of the all-reduced flat buffer, whose last float is the
  // sum of the ranks' `dropped` indicators: any non-zero bit pattern means some rank's gradient carries K9's NaN poison — the update is skipped
  // on EVERY rank, rays, both moments and the step count keep their values)
  if (guard && guard[2] != 0u) return;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const float t = step[0] + 1.0f;
  if (i < n) adam_clamp_one(i, t, rays, grad, grad_b, scale_a, grad_out, m, v, lr, beta1, beta2, eps_d, KF.m, KI.m, lo, hi, n_norm);
  // every lane has read the step count: the last workgroup to get here... (one workgroup covers up to 256 rays;
  // with more rays the count must not be bumped before every workgroup has read it: the host passes it instead)
  if (gridDim.x == 1) {
    __syncthreads();
    if (threadIdx.x == 0) step[0] = t;
  }
}
__global__ void k_bump_step(float *step, const unsigned int *guard) { if (!(guard && guard[2] != 0u)) step[0] += 1.0f; }

// =================================================================================== K3 blur
// (ksize x ksize) Gaussian, reflect border.  One workgroup per 32x8 output tile, halo staged in LDS.
__global__ void __launch_bounds__(SPLAT_BLOCK) k_blur_fwd(const float *__restrict__ in, int h, int w, BlurW bw, float *__restrict__ out) {
  __shared__ float tile[(TILE_H + 14) * (TILE_W + 14)];
  const int r = bw.ksize / 2, tw = TILE_W + 2 * r, th = TILE_H + 2 * r;
  const int x0 = blockIdx.x * TILE_W, y0 = blockIdx.y * TILE_H;
  for (int t = threadIdx.x; t < tw * th; t += SPLAT_BLOCK) {
    int ly = t / tw, lx = t % tw;
    tile[t] = in[(size_t)reflect_idx(y0 + ly - r, h) * w + reflect_idx(x0 + lx - r, w)];
  }
  __syncthreads();
  int lx = threadIdx.x % TILE_W, ly = threadIdx.x / TILE_W;
  int x = x0 + lx, y = y0 + ly;
  if (x >= w || y >= h) return;
  float acc = 0.f;
  for (int ky = 0; ky < bw.ksize; ++ky) {
    float row = 0.f;
    for (int kx = 0; kx < bw.ksize; ++kx) row = fmaf(bw.w[kx], tile[(ly + ky) * tw + lx + kx], row);
    acc = fmaf(bw.w[ky], row, acc);
  }
  out[(size_t)y * w + x] = acc;
}
// ---- dataset path (include/ffx.h: ffx_silhouette_fwd, ffx_noise_clamp, ffx_rgb_to_gray): one launch per post-processing step
// the image times the blurred disc: k_blur_fwd with the mask evaluated where it stages its tile (the mask is never stored)
__global__ void __launch_bounds__(SPLAT_BLOCK) k_silhouette_fwd(const float *__restrict__ img, int h, int w, int cx, int cy, int r2, BlurW bw, float *__restrict__ out) {
  __shared__ float tile[(TILE_H + 14) * (TILE_W + 14)];
  const int r = bw.ksize / 2, tw = TILE_W + 2 * r, th = TILE_H + 2 * r;
  const int x0 = blockIdx.x * TILE_W, y0 = blockIdx.y * TILE_H;
  for (int t = threadIdx.x; t < tw * th; t += SPLAT_BLOCK) {
    const int yy = reflect_idx(y0 + t / tw - r, h) - cy, xx = reflect_idx(x0 + t % tw - r, w) - cx;
    tile[t] = (xx * xx + yy * yy <= r2) ? 1.0f : 0.0f;
  }
  __syncthreads();
  const int lx = threadIdx.x % TILE_W, ly = threadIdx.x / TILE_W;
  const int x = x0 + lx, y = y0 + ly;
  if (x >= w || y >= h) return;
  float acc = 0.f;
  for (int ky = 0; ky < bw.ksize; ++ky) {
    float row = 0.f;
    for (int kx = 0; kx < bw.ksize; ++kx) row = fmaf(bw.w[kx], tile[(ly + ky) * tw + lx + kx], row);
    acc = fmaf(bw.w[ky], row, acc);
  }
  out[(size_t)y * w + x] = img[(size_t)y * w + x] * acc;
}
// (noise and out may be the SAME array — ops.noise_clamp writes the result over the noise it drew: no __restrict__ on either)
__global__ void __launch_bounds__(256) k_noise_clamp(const float *__restrict__ img, const float *noise, size_t n, float mean, float sd, float lo, float hi, float *out) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float nz = noise[i] * sd;
    const float v = img[i] + (nz + mean);
    out[i] = (v != v) ? v : fminf(fmaxf(v, lo), hi);
  }
}
template <bool F16>
__global__ void __launch_bounds__(256) k_rgb_to_gray(const void *__restrict__ img, size_t n, float wr, float wg, float wb, float *__restrict__ out) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float r, g, b;
    if (F16) { const _Float16 *p = (const _Float16 *)img + 3 * i; r = (float)p[0]; g = (float)p[1]; b = (float)p[2]; }
    else { const float *p = (const float *)img + 3 * i; r = p[0]; g = p[1]; b = p[2]; }
    const float a = r * wr, c = g * wg, d = b * wb;
    out[i] = (a + c) + d;
  }
}

// K3^T (blur_bwd_texel above).  One workgroup per 32x8 tile.  Whatever a pixel of the tile reads lies in the tile's halo
// [y0 - r, y0 + TILE_H + r) x [x0 - r, x0 + TILE_W + r) clipped to the image (the reflected candidates of a border pixel read the
// first / last r rows, which are in the halo of the border tiles): staged through LDS like the forward blur, zero outside the
// image.  Reading the 25 taps per pixel from global memory — and the candidate search of the border tiles — had made this kernel
// 3x as slow as k_blur_fwd (15 us against 5).  Images smaller than 2r + 3 (reflections of reflections) keep a direct loop.
template <int KS>
__global__ void __launch_bounds__(SPLAT_BLOCK) k_blur_bwd(const float *__restrict__ gout, int h, int w, BlurW bw, float *__restrict__ gin) {
  __shared__ float tile[(TILE_H + 14) * (TILE_W + 14)];
  const int ksize = KS ? KS : bw.ksize;
  const int r = ksize / 2;
  const int x0 = blockIdx.x * TILE_W, y0 = blockIdx.y * TILE_H;
  const int lx = threadIdx.x % TILE_W, ly = threadIdx.x / TILE_W;
  const int x = x0 + lx, y = y0 + ly;
  const bool small = h <= 2 * r + 2 || w <= 2 * r + 2; // (uniform)
  const int tw = TILE_W + 2 * r, th = TILE_H + 2 * r;
  if (!small) {
    for (int t = threadIdx.x; t < tw * th; t += SPLAT_BLOCK) {
      const int gy = y0 + t / tw - r, gx = x0 + t % tw - r;
      tile[t] = (gy >= 0 && gy < h && gx >= 0 && gx < w) ? gout[(size_t)gy * w + gx] : 0.f;
    }
    __syncthreads();
  }
  if (x >= w || y >= h) return;
  if (!small) {
    gin[(size_t)y * w + x] = blur_bwd_texel<KS>(tile, tw, y0 - r, x0 - r, y, x, h, w, bw);
    return;
  }
  float acc = 0.f;
  // candidate padded rows: y itself, then the r rows above the image and the r rows below it
  for (int a = -1; a < 2 * r; ++a) {
    int ty = (a < 0) ? y : (a < r ? -(a + 1) : h + (a - r));
    if (a >= 0 && reflect_idx(ty, h) != y) continue;
    for (int b = -1; b < 2 * r; ++b) {
      int tx = (b < 0) ? x : (b < r ? -(b + 1) : w + (b - r));
      if (b >= 0 && reflect_idx(tx, w) != x) continue;
      for (int ky = 0; ky < ksize; ++ky) {
        int py = ty - ky + r;
        if (py < 0 || py >= h) continue;
        float row = 0.f;
        for (int kx = 0; kx < ksize; ++kx) {
          int px = tx - kx + r;
          if (px < 0 || px >= w) continue;
          row = fmaf(bw.w[kx], gout[(size_t)py * w + px], row);
        }
        acc = fmaf(bw.w[ky], row, acc);
      }
    }
  }
  gin[(size_t)y * w + x] = acc;
}

// =================================================================================== host entry points
static int blur_weights(int ksize, float sg, BlurW &bw) {
  if (ksize < 1 || ksize > 15 || !(ksize & 1) || !(sg > 0.f)) return 0;
  int r = ksize / 2;
  double sum = 0, g[15];
  for (int k = 0; k < ksize; ++k) {
    double x = (double)(k - r);
    g[k] = exp(-(x * x) / (2.0 * (double)sg * (double)sg));
    bw.w[k] = (float)g[k];
    sum += g[k];
  }
  for (int k = 0; k < ksize; ++k) bw.w[k] = (float)((double)bw.w[k] / sum);
  for (int k = ksize; k < 15; ++k) bw.w[k] = 0.f;
  bw.ksize = ksize;
  return 1;
}

extern "C" {

int ffx_project_rays_fwd(const float *rays, int n, const float *KF, float *pts, ffx_stream s) {
  if (n == 0) return FFX_OK;
  if (!rays || !KF || !pts || n < 0) FFX_FAIL(FFX_ERR_ARG, "project_rays_fwd: bad argument");
  if (n == 0) return FFX_OK;
  Mat4 m;
  for (int i = 0; i < 16; ++i) m.m[i] = KF[i];
  hipLaunchKernelGGL(k_project_fwd, dim3(ffx_cdiv(n, 256)), dim3(256), 0, (hipStream_t)s, rays, n, m, pts);
  FFX_CHECK_LAUNCH("project_rays_fwd");
  return FFX_OK;
}

int ffx_project_rays_bwd(const float *rays, int n, const float *KF, const float *gpts, float *grays, ffx_stream s) {
  if (n == 0) return FFX_OK;
  if (!rays || !KF || !gpts || !grays || n < 0) FFX_FAIL(FFX_ERR_ARG, "project_rays_bwd: bad argument");
  if (n == 0) return FFX_OK;
  Mat4 m;
  for (int i = 0; i < 16; ++i) m.m[i] = KF[i];
  hipLaunchKernelGGL(k_project_bwd, dim3(ffx_cdiv(n, 256)), dim3(256), 0, (hipStream_t)s, rays, n, m, gpts, grays);
  FFX_CHECK_LAUNCH("project_rays_bwd");
  return FFX_OK;
}

int ffx_transform_points(const float *pts, int n, const float *M, int mode, float *out, ffx_stream s) {
  if (n == 0) return FFX_OK;
  if (!pts || !M || !out || n < 0 || (mode != 0 && mode != 1)) FFX_FAIL(FFX_ERR_ARG, "transform_points: bad argument");
  if (n == 0) return FFX_OK;
  Mat4 m;
  for (int i = 0; i < 16; ++i) m.m[i] = M[i];
  hipLaunchKernelGGL(k_transform_points, dim3(ffx_cdiv(n, 256)), dim3(256), 0, (hipStream_t)s, pts, n, m, mode, out);
  FFX_CHECK_LAUNCH("transform_points");
  return FFX_OK;
}

int ffx_l1_value_grad(const float *a, const float *b, long n, float weight, float *ws, float *g, ffx_stream s) {
  return ffx_l1_value_grad_acc(a, b, n, weight, ws, g, nullptr, s);
}
int ffx_l1_value_grad_acc(const float *a, const float *b, long n, float weight, float *ws, float *g, float *acc, ffx_stream s) {
  if (!a || !b || !ws || !g || n < 1) FFX_FAIL(FFX_ERR_ARG, "l1_value_grad: bad argument");
  int blocks = (int)((n + 255) / 256 < 256 ? (n + 255) / 256 : 256);
  hipLaunchKernelGGL(k_l1_partial, dim3(blocks), dim3(256), 0, (hipStream_t)s, a, b, n, weight / (float)n, ws, g);
  hipLaunchKernelGGL(k_l1_final, dim3(1), dim3(256), 0, (hipStream_t)s, ws, blocks, weight / (float)n, acc);
  FFX_CHECK_LAUNCH("l1_value_grad");
  return FFX_OK;
}


size_t ffx_pattern_ws_floats(int size0, int size1) {
  if (size0 < 1 || size1 < 1) return 0;
  return (size_t)ffx_cdiv(size0, TILE_W) * ffx_cdiv(size1, TILE_H);
}

int ffx_pattern_fwd(const float *rays, int n, const float *KF, float sigma, int size0, int size1, int want_softor, float *pts, float *tsum, float *tsor,
                    float *ws, float *zero, long n_zero, ffx_stream s) {
  if (!rays || !KF || !pts || !tsum || n < 1 || size0 <= 0 || size1 <= 0 || !(sigma > 0.f) || (want_softor && (!tsor || !ws)) || (zero && n_zero < 1))
    FFX_FAIL(FFX_ERR_ARG, "pattern_fwd: bad argument");
  Mat4 m;
  for (int i = 0; i < 16; ++i) m.m[i] = KF[i];
  dim3 grid(ffx_cdiv(size0, TILE_W), ffx_cdiv(size1, TILE_H));
  hipLaunchKernelGGL(k_pattern_fwd, grid, dim3(SPLAT_BLOCK), 0, (hipStream_t)s, rays, n, m, sigma, size0, size1, want_softor, pts, tsum, tsor, ws, zero,
                     zero ? n_zero : 0L);
  FFX_CHECK_LAUNCH("pattern_fwd");
  return FFX_OK;
}

int ffx_pattern_bwd(const float *rays, int n, const float *KF, float sigma, int size0, int size1, const float *tsum, const float *tsor, const float *gts,
                    float reg_weight, const float *ws, float *grays_data, float *grays_reg, float *reg_value, const float *loss_in, int loss_in_n, float loss_div, ffx_stream s) {
  if (!rays || !KF || n < 1 || size0 <= 0 || size1 <= 0 || !(sigma > 0.f) || (gts && !grays_data) || (reg_weight > 0.f && (!tsum || !tsor || !ws || !grays_reg)) ||
      (loss_in && (loss_in_n < 1 || !reg_value)))
    FFX_FAIL(FFX_ERR_ARG, "pattern_bwd: bad argument");
  if (n > 65535) FFX_FAIL(FFX_ERR_UNSUPPORTED, "pattern_bwd: more than 65535 points");
  Mat4 m;
  for (int i = 0; i < 16; ++i) m.m[i] = KF[i];
  BlurW nobw;
  AdamK noadam;
  memset(&nobw, 0, sizeof nobw);
  memset(&noadam, 0, sizeof noadam);
  hipLaunchKernelGGL(k_pattern_bwd<-1>, dim3(n), dim3(SPLAT_BLOCK), 0, (hipStream_t)s, rays, n, m, sigma, size0, size1, tsum, tsor, gts, reg_weight, ws,
                     ws ? (int)ffx_pattern_ws_floats(size0, size1) : 0, gts ? grays_data : nullptr, reg_weight > 0.f ? grays_reg : nullptr, reg_value, loss_in,
                     loss_in ? loss_in_n : 0, loss_div > 0.f ? loss_div : 1.0f, nobw, noadam);
  FFX_CHECK_LAUNCH("pattern_bwd");
  return FFX_OK;
}

int ffx_pattern_fwd_blur(const float *rays, int n, const float *KF, float sigma, int size0, int size1, int want_softor, float *pts, float *tsum, float *tsor,
                         float *ws, float *zero, long n_zero, int blur_ksize, float blur_sigma, float *tex, ffx_stream s) {
  if (!tex) FFX_FAIL(FFX_ERR_ARG, "pattern_fwd_blur: tex is NULL");
  BlurW bw;
  if (!blur_weights(blur_ksize, blur_sigma, bw)) FFX_FAIL(FFX_ERR_ARG, "pattern_fwd_blur: ksize must be odd, 1..15, sigma positive");
  if (blur_ksize != 5) { // other sizes: the separate launches (same values)
    const int rc = ffx_pattern_fwd(rays, n, KF, sigma, size0, size1, want_softor, pts, tsum, tsor, ws, zero, n_zero, s);
    return rc != FFX_OK ? rc : ffx_blur_fwd(tsum, size1, size0, blur_ksize, blur_sigma, tex, s);
  }
  if (!rays || !KF || !pts || !tsum || n < 1 || size0 <= 0 || size1 <= 0 || !(sigma > 0.f) || (want_softor && (!tsor || !ws)) || (zero && n_zero < 1))
    FFX_FAIL(FFX_ERR_ARG, "pattern_fwd_blur: bad argument");
  Mat4 m;
  for (int i = 0; i < 16; ++i) m.m[i] = KF[i];
  dim3 grid(ffx_cdiv(size0, TILE_W), ffx_cdiv(size1, TILE_H));
  hipLaunchKernelGGL(k_pattern_fwd_blur<2>, grid, dim3(SPLAT_BLOCK), 0, (hipStream_t)s, rays, n, m, sigma, size0, size1, want_softor, pts, tsum, tsor, ws, zero,
                     zero ? n_zero : 0L, bw, tex);
  FFX_CHECK_LAUNCH("pattern_fwd_blur");
  return FFX_OK;
}

int ffx_pattern_bwd_blur(const float *rays, int n, const float *KF, float sigma, int size0, int size1, const float *tsum, const float *tsor, const float *gtex,
                         float reg_weight, const float *ws, float *grays_data, float *grays_reg, float *reg_value, const float *loss_in, int loss_in_n, float loss_div,
                         int blur_ksize, float blur_sigma, float *gts_scratch, const ffx_adam_args *adam, ffx_stream s) {
  if (!rays || !KF || n < 1 || size0 <= 0 || size1 <= 0 || !(sigma > 0.f) || (gtex && !grays_data) || (reg_weight > 0.f && (!tsum || !tsor || !ws || !grays_reg)) ||
      (loss_in && (loss_in_n < 1 || !reg_value)))
    FFX_FAIL(FFX_ERR_ARG, "pattern_bwd_blur: bad argument");
  if (n > 65535) FFX_FAIL(FFX_ERR_UNSUPPORTED, "pattern_bwd_blur: more than 65535 points");
  BlurW bw;
  memset(&bw, 0, sizeof bw);
  if (blur_ksize != 0 && !blur_weights(blur_ksize, blur_sigma, bw)) FFX_FAIL(FFX_ERR_ARG, "pattern_bwd_blur: ksize must be 0 (no blur) or odd, 1..15, with a positive sigma");
  AdamK ak;
  memset(&ak, 0, sizeof ak);
  if (adam) {
    const bool no_update = !adam->exp_avg && !adam->exp_avg_sq && !adam->step;
    if (no_update && !adam->dot_a) FFX_FAIL(FFX_ERR_ARG, "pattern_bwd_blur: Adam arguments without state and without an inner product: nothing to do");
    if (!adam->rays || !adam->counter || (!no_update && (!adam->exp_avg || !adam->exp_avg_sq || !adam->step || !(adam->grad_div > 0.f) || adam->n_normalize < 0 || !(adam->lo <= adam->hi))))
      FFX_FAIL(FFX_ERR_ARG, "pattern_bwd_blur: bad Adam arguments");
    if (!no_update && (grays_reg || adam->grad_div != 1.0f) && !adam->grad_out) FFX_FAIL(FFX_ERR_ARG, "pattern_bwd_blur: combining gradients needs grad_out");
    ak.no_update = no_update ? 1 : 0;
    if (adam->rays != rays) FFX_FAIL(FFX_ERR_ARG, "pattern_bwd_blur: the update is applied to the rays the gradient was taken at");
    ak.guard = (const unsigned int *)adam->guard;
    ak.rays = adam->rays; ak.m = adam->exp_avg; ak.v = adam->exp_avg_sq; ak.step = adam->step; ak.grad_out = adam->grad_out; ak.counter = adam->counter;
    ak.lr = adam->lr; ak.beta1 = adam->beta1; ak.beta2 = adam->beta2; ak.eps = adam->eps;
    for (int i = 0; i < 16; ++i) ak.KI.m[i] = adam->KF_inv[i];
    ak.lo = adam->lo; ak.hi = adam->hi; ak.grad_div = adam->grad_div; ak.n_norm = adam->n_normalize;
    if (adam->dot_a) {
      if (!adam->dot_b || adam->dot_n < 1 || !adam->dot_partial || !reg_value) FFX_FAIL(FFX_ERR_ARG, "pattern_bwd_blur: the inner product needs dot_b, dot_n, dot_partial and reg_value");
      if (loss_in) FFX_FAIL(FFX_ERR_ARG, "pattern_bwd_blur: the data term comes either as partial sums (loss_in) or as an inner product (dot_a), not both");
      if (adam->dot_b_n < 0 || adam->dot_b_n > adam->dot_n) FFX_FAIL(FFX_ERR_ARG, "pattern_bwd_blur: dot_b_n must be 0 or a period <= dot_n");
      ak.dot_a = adam->dot_a; ak.dot_b = adam->dot_b; ak.dot_n = (long)adam->dot_n; ak.dot_partial = adam->dot_partial;
      ak.dot_b_n = adam->dot_b_n > 0 ? (long)adam->dot_b_n : (long)adam->dot_n;
    }
  }
  Mat4 m;
  for (int i = 0; i < 16; ++i) m.m[i] = KF[i];
  const int n_ws = ws ? (int)ffx_pattern_ws_floats(size0, size1) : 0;
  float *gd = gtex ? grays_data : nullptr, *gr = reg_weight > 0.f ? grays_reg : nullptr;
  const float ld = loss_div > 0.f ? loss_div : 1.0f;
  const int lin = loss_in ? loss_in_n : 0;
  // the footprint + halo must fit the workgroup's LDS next to the neighbour list; images too small for the windowed transpose
  // (reflections of reflections) and windows too large take the separate transpose blur into gts_scratch
  const int r = blur_ksize / 2;
  const int wmax = 2 * (int)ceilf(sqrtf(FFX_QCUT * sigma) + 1.0f) + 4 + 2 * r;
  const size_t lds = (size_t)wmax * wmax * sizeof(float);
  const bool windowed = blur_ksize > 0 && gtex && lds <= 40 * 1024 && size0 > 2 * r + 2 && size1 > 2 * r + 2;
  if (blur_ksize > 0 && gtex && !windowed) {
    if (!gts_scratch) FFX_FAIL(FFX_ERR_ARG, "pattern_bwd_blur: this size needs gts_scratch [size1,size0]");
    const int rc = ffx_blur_bwd(gtex, size1, size0, blur_ksize, blur_sigma, gts_scratch, s);
    if (rc != FFX_OK) return rc;
    gtex = gts_scratch;
  }
#define FFX_LAUNCH_PB(KS_, LDS_)                                                                                                                              \
  hipLaunchKernelGGL(k_pattern_bwd<KS_>, dim3(n), dim3(SPLAT_BLOCK), LDS_, (hipStream_t)s, rays, n, m, sigma, size0, size1, tsum, tsor, gtex, reg_weight, ws, \
                     n_ws, gd, gr, reg_value, loss_in, lin, ld, bw, ak)
  if (!windowed) FFX_LAUNCH_PB(-1, 0);
  else if (blur_ksize == 5) FFX_LAUNCH_PB(5, lds);
  else FFX_LAUNCH_PB(0, lds);
#undef FFX_LAUNCH_PB
  FFX_CHECK_LAUNCH("pattern_bwd_blur");
  return FFX_OK;
}

int ffx_pattern_step(float *rays, int n, const float *KF, float sigma, int size0, int size1, float *tsum, float *tsor, const float *gtex, float reg_weight, float *ws,
                     float *grays_data, float *grays_reg, float *reg_value, const float *loss_in, int loss_in_n, float loss_div, int blur_ksize, float blur_sigma,
                     const ffx_adam_args *adam, float *pts, float *zero, long n_zero, float *tex, float *rays_kept, int check_kept, void *sync, uint32_t epoch, ffx_stream s) {
  if (!rays || !KF || n < 1 || size0 <= 0 || size1 <= 0 || !(sigma > 0.f) || !tsum || !pts || !tex || !sync || !reg_value || (gtex && !grays_data) ||
      (reg_weight > 0.f && (!tsor || !ws || !grays_reg)) || (loss_in && loss_in_n < 1) || (zero && n_zero < 1) || !rays_kept || ((uintptr_t)sync & 7) != 0 || epoch == 0u)
    FFX_FAIL(FFX_ERR_ARG, "pattern_step: bad argument");
  if (n > 65535) FFX_FAIL(FFX_ERR_UNSUPPORTED, "pattern_step: more than 65535 points");
  if (blur_ksize != 5) FFX_FAIL(FFX_ERR_UNSUPPORTED, "pattern_step: blur_ksize must be 5 (other sizes: ffx_pattern_bwd_blur + ffx_pattern_fwd_blur)");
  BlurW bw;
  if (!blur_weights(blur_ksize, blur_sigma, bw)) FFX_FAIL(FFX_ERR_ARG, "pattern_step: blur sigma must be positive");
  if (!adam || !adam->exp_avg || !adam->exp_avg_sq || !adam->step || adam->rays != rays || !(adam->grad_div > 0.f) || adam->n_normalize < 0 || !(adam->lo <= adam->hi))
    FFX_FAIL(FFX_ERR_ARG, "pattern_step: needs Adam arguments with state, for the rays of the call");
  if ((grays_reg || adam->grad_div != 1.0f) && !adam->grad_out) FFX_FAIL(FFX_ERR_ARG, "pattern_step: combining gradients needs grad_out");
  if (adam->guard && ((uintptr_t)adam->guard & 3) != 0) FFX_FAIL(FFX_ERR_ARG, "pattern_step: guard must be 4-byte aligned");
  AdamK ak;
  memset(&ak, 0, sizeof ak);
  ak.guard = (const unsigned int *)adam->guard;
  ak.rays = adam->rays; ak.m = adam->exp_avg; ak.v = adam->exp_avg_sq; ak.step = adam->step; ak.grad_out = adam->grad_out;
  ak.lr = adam->lr; ak.beta1 = adam->beta1; ak.beta2 = adam->beta2; ak.eps = adam->eps;
  for (int i = 0; i < 16; ++i) ak.KI.m[i] = adam->KF_inv[i];
  ak.lo = adam->lo; ak.hi = adam->hi; ak.grad_div = adam->grad_div; ak.n_norm = adam->n_normalize;
  if (adam->dot_a) {
    if (!adam->dot_b || adam->dot_n < 1 || !adam->dot_partial) FFX_FAIL(FFX_ERR_ARG, "pattern_step: the inner product needs dot_b, dot_n and dot_partial");
    if (loss_in) FFX_FAIL(FFX_ERR_ARG, "pattern_step: the data term comes either as partial sums (loss_in) or as an inner product (dot_a), not both");
    if (adam->dot_b_n < 0 || adam->dot_b_n > adam->dot_n) FFX_FAIL(FFX_ERR_ARG, "pattern_step: dot_b_n must be 0 or a period <= dot_n");
    ak.dot_a = adam->dot_a; ak.dot_b = adam->dot_b; ak.dot_n = (long)adam->dot_n; ak.dot_partial = adam->dot_partial;
    ak.dot_b_n = adam->dot_b_n > 0 ? (long)adam->dot_b_n : (long)adam->dot_n;
  }
  const int r = blur_ksize / 2;
  const int wmax = 2 * (int)ceilf(sqrtf(FFX_QCUT * sigma) + 1.0f) + 4 + 2 * r;
  const size_t lds = (size_t)wmax * wmax * sizeof(float);
  if (gtex && !(lds <= 24 * 1024 && size0 > 2 * r + 2 && size1 > 2 * r + 2))
    FFX_FAIL(FFX_ERR_UNSUPPORTED, "pattern_step: this footprint / image size takes the separate launches (ffx_pattern_bwd_blur + ffx_pattern_fwd_blur)");
  Mat4 m;
  for (int i = 0; i < 16; ++i) m.m[i] = KF[i];
  FwdK fw;
  memset(&fw, 0, sizeof fw);
  fw.want_softor = tsor && ws ? 1 : 0; // (the next step's regulariser outputs whenever the caller keeps buffers for them)
  fw.pts = pts; fw.tsum = tsum; fw.tsor = fw.want_softor ? tsor : nullptr; fw.ws = fw.want_softor ? ws : nullptr; fw.zero = zero; fw.n_zero = zero ? n_zero : 0L; fw.tex = tex;
  fw.kept_new = rays_kept + (size_t)(epoch & 1u) * 3 * (size_t)n; fw.kept_old = rays_kept + (size_t)((epoch & 1u) ^ 1u) * 3 * (size_t)n; fw.check_kept = check_kept ? 1 : 0;
  fw.nbx = ffx_cdiv(size0, TILE_W); fw.nby = ffx_cdiv(size1, TILE_H);
  static const int pow_cache = [] { const char *e = getenv("FFX_ADAM_POW_CACHE"); return e ? atoi(e) : 1; }();
  fw.pow_cache = pow_cache;
  fw.epoch = epoch;
  // one helper per forward tile up to 1024 (5 workgroups of this kernel fit a CU by LDS: 1280 at a time; beyond that helpers take several tiles
  // each); the gradient's n workgroups in front
  const long tiles = (long)fw.nbx * fw.nby;
  const int helpers = (int)(tiles < 1024 ? tiles : 1024);
  const float *gd_in = gtex;
  hipLaunchKernelGGL(k_pattern_step<5>, dim3(n + helpers), dim3(SPLAT_BLOCK), gtex ? lds : 0, (hipStream_t)s, rays, n, m, sigma, size0, size1, tsum, tsor, gd_in, reg_weight, ws,
                     ws ? (int)ffx_pattern_ws_floats(size0, size1) : 0, gtex ? grays_data : nullptr, reg_weight > 0.f ? grays_reg : nullptr, reg_value, loss_in, loss_in ? loss_in_n : 0,
                     loss_div > 0.f ? loss_div : 1.0f, bw, ak, fw, (PatSync *)sync);
  FFX_CHECK_LAUNCH("pattern_step");
  return FFX_OK;
}

int ffx_adam_clamp_step(float *rays, const float *grad, const float *grad_b, float grad_div, float *grad_out, float *exp_avg, float *exp_avg_sq, float *step, int n,
                        double lr, double beta1, double beta2, double eps, const float *KF, const float *KF_inv, float lo, float hi, int n_normalize, const void *guard,
                        ffx_stream s) {
  if ((grad_b || grad_div != 1.0f) && !grad_out) FFX_FAIL(FFX_ERR_ARG, "adam_clamp_step: combining gradients needs grad_out");
  if (guard && ((uintptr_t)guard & 3) != 0) FFX_FAIL(FFX_ERR_ARG, "adam_clamp_step: guard must be 4-byte aligned");
  if (!(grad_div > 0.f)) FFX_FAIL(FFX_ERR_ARG, "adam_clamp_step: grad_div must be positive");
  if (!rays || !grad || !exp_avg || !exp_avg_sq || !step || !KF || !KF_inv || n < 1 || n_normalize < 0 || !(lo <= hi)) FFX_FAIL(FFX_ERR_ARG, "adam_clamp_step: bad argument");
  Mat4 a, b;
  for (int i = 0; i < 16; ++i) { a.m[i] = KF[i]; b.m[i] = KF_inv[i]; }
  const int blocks = ffx_cdiv(n, 256);
  hipLaunchKernelGGL(k_adam_clamp, dim3(blocks), dim3(256), 0, (hipStream_t)s, rays, grad, grad_b, grad_div, grad_out, exp_avg, exp_avg_sq, step, n, lr, beta1,
                     beta2, eps, a, b, lo, hi, n_normalize, (const unsigned int *)guard);
  if (blocks > 1) hipLaunchKernelGGL(k_bump_step, dim3(1), dim3(1), 0, (hipStream_t)s, step, (const unsigned int *)guard);
  FFX_CHECK_LAUNCH("adam_clamp_step");
  return FFX_OK;
}

int ffx_clamp_to_fov(float *rays, int n, const float *KF, const float *KF_inv, float lo, float hi, int n_normalize, ffx_stream s) {
  if (n == 0) return FFX_OK;
  if (!rays || !KF || !KF_inv || n < 0 || n_normalize < 0 || !(lo <= hi)) FFX_FAIL(FFX_ERR_ARG, "clamp_to_fov: bad argument");
  Mat4 a, b;
  for (int i = 0; i < 16; ++i) { a.m[i] = KF[i]; b.m[i] = KF_inv[i]; }
  hipLaunchKernelGGL(k_clamp_to_fov, dim3(ffx_cdiv(n, 256)), dim3(256), 0, (hipStream_t)s, rays, n, a, b, lo, hi, n_normalize);
  FFX_CHECK_LAUNCH("clamp_to_fov");
  return FFX_OK;
}

static int dense_fwd_common(const float *pts, const float *depth, int n, float sigma, int size0, int size1, float *out, ffx_stream s, int mode,
                            const char *what) {
  if (n == 0) return FFX_OK;
  if (!pts || !out || n < 0 || size0 <= 0 || size1 <= 0 || !(sigma > 0.f)) FFX_FAIL(FFX_ERR_ARG, "%s: bad argument", what);
  if (n == 0) return FFX_OK;
  if (n > 65535) FFX_FAIL(FFX_ERR_UNSUPPORTED, "%s: more than 65535 points", what);
  int w4 = (size0 + 3) / 4;
  int vec_ok = (size0 % 4 == 0) && (((uintptr_t)out & 15) == 0);
  dim3 grid(ffx_cdiv((long)w4 * size1, SPLAT_BLOCK), n);
  if (mode == 0)
    hipLaunchKernelGGL(k_splat_dense_fwd<0>, grid, dim3(SPLAT_BLOCK), 0, (hipStream_t)s, pts, depth, sigma, size0, size1, w4, vec_ok, out);
  else
    hipLaunchKernelGGL(k_splat_dense_fwd<1>, grid, dim3(SPLAT_BLOCK), 0, (hipStream_t)s, pts, depth, sigma, size0, size1, w4, vec_ok, out);
  FFX_CHECK_LAUNCH(what);
  return FFX_OK;
}

int ffx_splat_dense_fwd(const float *pts, int n, float sigma, int size0, int size1, float *out, ffx_stream s) {
  return dense_fwd_common(pts, nullptr, n, sigma, size0, size1, out, s, 0, "splat_dense_fwd");
}

int ffx_splat_depth_fwd(const float *pts, const float *depth, int n, float sigma, int size0, int size1, float *out, ffx_stream s) {
  if (!depth) FFX_FAIL(FFX_ERR_ARG, "splat_depth_fwd: bad argument");
  return dense_fwd_common(pts, depth, n, sigma, size0, size1, out, s, 1, "splat_depth_fwd");
}

int ffx_splat_lines_fwd(const float *lines, int n, float sigma, int size0, int size1, float *out, ffx_stream s) {
  if (n == 0) return FFX_OK;
  if (!lines || !out || n < 0 || size0 <= 0 || size1 <= 0 || !(sigma > 0.f)) FFX_FAIL(FFX_ERR_ARG, "splat_lines_fwd: bad argument");
  if (n == 0) return FFX_OK;
  if (n > 65535) FFX_FAIL(FFX_ERR_UNSUPPORTED, "splat_lines_fwd: more than 65535 lines");
  dim3 grid(ffx_cdiv((long)size0 * size1, SPLAT_BLOCK), n);
  hipLaunchKernelGGL(k_splat_lines_fwd, grid, dim3(SPLAT_BLOCK), 0, (hipStream_t)s, lines, sigma, size0, size1, out);
  FFX_CHECK_LAUNCH("splat_lines_fwd");
  return FFX_OK;
}

int ffx_splat_lines_bwd(const float *lines, int n, float sigma, int size0, int size1, const float *gout, float *glines, ffx_stream s) {
  if (n == 0) return FFX_OK;
  if (!lines || !gout || !glines || n < 0 || size0 <= 0 || size1 <= 0 || !(sigma > 0.f)) FFX_FAIL(FFX_ERR_ARG, "splat_lines_bwd: bad argument");
  hipLaunchKernelGGL(k_splat_lines_bwd, dim3(n), dim3(1024), 0, (hipStream_t)s, lines, sigma, size0, size1, gout, glines);
  FFX_CHECK_LAUNCH("splat_lines_bwd");
  return FFX_OK;
}

int ffx_splat_dense_bwd(const float *pts, int n, float sigma, int size0, int size1, const float *gout, float *gpts, ffx_stream s) {
  if (n == 0) return FFX_OK;
  if (!pts || !gout || !gpts || n < 0 || size0 <= 0 || size1 <= 0 || !(sigma > 0.f)) FFX_FAIL(FFX_ERR_ARG, "splat_dense_bwd: bad argument");
  if (n == 0) return FFX_OK;
  hipLaunchKernelGGL((k_splat_bwd<false, true>), dim3(n), dim3(SPLAT_BLOCK), 0, (hipStream_t)s, pts, n, sigma, FFX_REDUCE_SUM, -1, size0, size1, gout,
                     gpts);
  FFX_CHECK_LAUNCH("splat_dense_bwd");
  return FFX_OK;
}

int ffx_splat_fwd(const float *pts, int n, float sigma, int reduce, int half_window, int size0, int size1, float *tex, ffx_stream s) {
  if ((!pts && n > 0) || !tex || n < 0 || size0 <= 0 || size1 <= 0 || !(sigma > 0.f)) FFX_FAIL(FFX_ERR_ARG, "splat_fwd: bad argument");
  if (reduce != FFX_REDUCE_SUM && reduce != FFX_REDUCE_SOFTOR) FFX_FAIL(FFX_ERR_ARG, "splat_fwd: bad reduce %d", reduce);
  dim3 grid(ffx_cdiv(size0, TILE_W), ffx_cdiv(size1, TILE_H));
  if (grid.y > 65535) FFX_FAIL(FFX_ERR_UNSUPPORTED, "splat_fwd: texture too tall");
  if (half_window >= 0)
    hipLaunchKernelGGL(k_splat_fused_fwd<true>, grid, dim3(SPLAT_BLOCK), 0, (hipStream_t)s, pts, n, sigma, reduce, half_window, size0, size1, tex);
  else
    hipLaunchKernelGGL(k_splat_fused_fwd<false>, grid, dim3(SPLAT_BLOCK), 0, (hipStream_t)s, pts, n, sigma, reduce, half_window, size0, size1, tex);
  FFX_CHECK_LAUNCH("splat_fwd");
  return FFX_OK;
}

int ffx_splat_bwd(const float *pts, int n, float sigma, int reduce, int half_window, int size0, int size1, const float *tex, const float *gtex,
                  float *gpts, ffx_stream s) {
  (void)tex; // the product over the other points is recomputed; the forward output is not needed
  if (n == 0) return FFX_OK;
  if (!pts || !gtex || !gpts || n < 0 || size0 <= 0 || size1 <= 0 || !(sigma > 0.f)) FFX_FAIL(FFX_ERR_ARG, "splat_bwd: bad argument");
  if (reduce != FFX_REDUCE_SUM && reduce != FFX_REDUCE_SOFTOR) FFX_FAIL(FFX_ERR_ARG, "splat_bwd: bad reduce %d", reduce);
  if (n == 0) return FFX_OK;
  if (half_window >= 0)
    hipLaunchKernelGGL((k_splat_bwd<true, false>), dim3(n), dim3(SPLAT_BLOCK), 0, (hipStream_t)s, pts, n, sigma, reduce, half_window, size0, size1,
                       gtex, gpts);
  else
    hipLaunchKernelGGL((k_splat_bwd<false, false>), dim3(n), dim3(SPLAT_BLOCK), 0, (hipStream_t)s, pts, n, sigma, reduce, half_window, size0, size1,
                       gtex, gpts);
  FFX_CHECK_LAUNCH("splat_bwd");
  return FFX_OK;
}

int ffx_silhouette_fwd(const float *img, int h, int w, int cx, int cy, int radius, int ksize, float sg, float *out, ffx_stream s) {
  BlurW bw;
  if (!img || !out || h <= 0 || w <= 0 || h > 32768 || w > 32768 || radius < 0 || radius > 32768 || cx < -32768 || cx > 65536 || cy < -32768 || cy > 65536 ||
      !blur_weights(ksize, sg, bw))
    FFX_FAIL(FFX_ERR_ARG, "silhouette_fwd: bad argument");
  dim3 grid((w + TILE_W - 1) / TILE_W, (h + TILE_H - 1) / TILE_H);
  hipLaunchKernelGGL(k_silhouette_fwd, grid, dim3(SPLAT_BLOCK), 0, (hipStream_t)s, img, h, w, cx, cy, radius * radius, bw, out);
  FFX_CHECK_LAUNCH("silhouette_fwd");
  return FFX_OK;
}
int ffx_noise_clamp(const float *img, const float *noise, size_t n, float mean, float sd, float lo, float hi, float *out, ffx_stream s) {
  if (!img || !noise || !out || n == 0) FFX_FAIL(FFX_ERR_ARG, "noise_clamp: bad argument");
  const size_t blocks = (n + 255) / 256;
  hipLaunchKernelGGL(k_noise_clamp, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, (hipStream_t)s, img, noise, n, mean, sd, lo, hi, out);
  FFX_CHECK_LAUNCH("noise_clamp");
  return FFX_OK;
}
int ffx_rgb_to_gray(const void *img, int img_fp16, size_t n, float wr, float wg, float wb, float *out, ffx_stream s) {
  if (!img || !out || n == 0) FFX_FAIL(FFX_ERR_ARG, "rgb_to_gray: bad argument");
  const size_t blocks = (n + 255) / 256;
  const dim3 grid((unsigned)(blocks < 4096 ? blocks : 4096));
  if (img_fp16 & 1) hipLaunchKernelGGL(k_rgb_to_gray<true>, grid, dim3(256), 0, (hipStream_t)s, img, n, wr, wg, wb, out);
  else hipLaunchKernelGGL(k_rgb_to_gray<false>, grid, dim3(256), 0, (hipStream_t)s, img, n, wr, wg, wb, out);
  FFX_CHECK_LAUNCH("rgb_to_gray");
  return FFX_OK;
}
int ffx_blur_fwd(const float *in, int h, int w, int ksize, float sg, float *out, ffx_stream s) {
  BlurW bw;
  if (!in || !out || h <= 0 || w <= 0 || !blur_weights(ksize, sg, bw)) FFX_FAIL(FFX_ERR_ARG, "blur_fwd: bad argument");
  dim3 grid(ffx_cdiv(w, TILE_W), ffx_cdiv(h, TILE_H));
  hipLaunchKernelGGL(k_blur_fwd, grid, dim3(SPLAT_BLOCK), 0, (hipStream_t)s, in, h, w, bw, out);
  FFX_CHECK_LAUNCH("blur_fwd");
  return FFX_OK;
}

int ffx_blur_bwd(const float *gout, int h, int w, int ksize, float sg, float *gin, ffx_stream s) {
  BlurW bw;
  if (!gout || !gin || h <= 0 || w <= 0 || !blur_weights(ksize, sg, bw)) FFX_FAIL(FFX_ERR_ARG, "blur_bwd: bad argument");
  dim3 grid(ffx_cdiv(w, TILE_W), ffx_cdiv(h, TILE_H));
  if (ksize == 5) hipLaunchKernelGGL(k_blur_bwd<5>, grid, dim3(SPLAT_BLOCK), 0, (hipStream_t)s, gout, h, w, bw, gin);
  else hipLaunchKernelGGL(k_blur_bwd<0>, grid, dim3(SPLAT_BLOCK), 0, (hipStream_t)s, gout, h, w, bw, gin);
  FFX_CHECK_LAUNCH("blur_bwd");
  return FFX_OK;
}

} // extern "C"
