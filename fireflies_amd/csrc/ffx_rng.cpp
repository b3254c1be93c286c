// ffx_rng.cpp — the sampler draws of a scene randomisation, on the host.
//
// The reference draws every random transform / attribute with `torch.rand(shape, device="cuda")`
// (fireflies/utils/math.py:170-175 via fireflies/sampling/uniform.py:16-19) and then pulls each value back
// with `.tolist()` / `.item()` (fireflies/scene.py:258-274): one kernel launch plus one device-to-host sync
// per draw, for three floats.  A seeded script must keep seeing those numbers, so the draw itself is
// reproduced: PyTorch-ROCm's uniform kernel is Philox4x32-10 keyed by the generator's seed, with the
// element index as the subsequence and the generator's running offset as the counter
// (ATen/native/cuda/DistributionTemplates.h: distribution_elementwise_grid_stride_kernel; rocrand's
// philox4x32_10_engine + uniform_distribution).  Evaluating it here costs ~40 integer multiplies per value
// and no device work at all; the caller advances the generator's offset by what the kernel launch would
// have consumed, so every later consumer of the generator sees the same stream as in the reference program.
#include <math.h>
#include <stdint.h>

#include "../../include/ffx.h"
#include "ffx_common.h"

namespace {
struct U4 { uint32_t x, y, z, w; };

inline U4 philox_round(U4 c, uint32_t k0, uint32_t k1) {
  const uint64_t m0 = (uint64_t)0xD2511F53u * c.x, m1 = (uint64_t)0xCD9E8D57u * c.z;
  return U4{(uint32_t)(m1 >> 32) ^ c.y ^ k0, (uint32_t)m1, (uint32_t)(m0 >> 32) ^ c.w ^ k1, (uint32_t)m0};
}

inline U4 philox4x32_10(U4 c, uint32_t k0, uint32_t k1) {
  for (int r = 0; r < 9; ++r) {
    c = philox_round(c, k0, k1);
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return philox_round(c, k0, k1);
}
} // namespace

extern "C" int ffx_torch_rand_h(uint64_t seed, uint64_t offset, int n, float *out, uint64_t *offset_increment) {
  if (!out || !offset_increment || n <= 0) FFX_FAIL(FFX_ERR_ARG, "torch_rand_h: bad argument");
  // one 256-thread block serves n <= 256 elements (thread i writes element i from the first of its four
  // outputs); larger tensors spread over a device-dependent grid — not needed for sampler bounds
  if (n > 256) FFX_FAIL(FFX_ERR_UNSUPPORTED, "torch_rand_h: more than 256 elements (%d)", n);
  if (offset & 3u) FFX_FAIL(FFX_ERR_UNSUPPORTED, "torch_rand_h: offset not a multiple of 4");
  const uint64_t ctr = offset >> 2;
  for (int i = 0; i < n; ++i) {
    const U4 r = philox4x32_10(U4{(uint32_t)ctr, (uint32_t)(ctr >> 32), (uint32_t)i, 0u}, (uint32_t)seed, (uint32_t)(seed >> 32));
    // rocrand uniform_distribution: 2^-32 + x * 2^-32 in float, (0, 1]; torch maps 1.0 to 0.0 ([0, 1))
    const float two32inv = 2.3283064e-10f;
    const float u = two32inv + (float)r.x * two32inv;
    out[i] = u == 1.0f ? 0.0f : u;
  }
  *offset_increment = 4; // ((n - 1) / (256 * 1 * 4) + 1) * 4 engine calls
  return FFX_OK;
}

// The draws of a whole randomisation — or of the S scene samples of an optimisation step — in ONE call: draw i is the
// torch.rand of counts[i] elements under (seeds[i], offsets[i]); the values are packed one draw after the other.  The
// caller has advanced the generator by 4 per draw when it reserved them (what every draw of <= 256 elements consumes).
extern "C" int ffx_torch_rand_batch_h(int k, const uint64_t *seeds, const uint64_t *offsets, const int32_t *counts, float *out) {
  if (k < 0 || (k > 0 && (!seeds || !offsets || !counts || !out))) FFX_FAIL(FFX_ERR_ARG, "torch_rand_batch_h: bad argument");
  for (int i = 0; i < k; ++i) {
    uint64_t inc = 0;
    const int rc = ffx_torch_rand_h(seeds[i], offsets[i], counts[i], out, &inc);
    if (rc != FFX_OK) return rc;
    out += counts[i];
  }
  return FFX_OK;
}

// ------------------------------------------------------------------------------------------ f1: a whole Scene.randomize(), natively
// The reference's randomisation is Python around torch.rand: per entity a translation and a rotation draw (a Mesh: a scale draw too),
// per attribute a draw, each mapped to its interval; then 4x4 algebra — (T + centroid) @ R @ [S @] world — and the parent chains
// (/root/reference/fireflies/scene.py:344-384, entity/base.py:194-244, entity/mesh.py:141-165).  Here the whole of it — the draws of
// every entity in the reference's order from the generator's Philox stream, the interval maps, the matrices, the chains — is ONE host
// call for all the scene samples of a step.  The float32 arithmetic is that of the Python mirror (fireflies_amd/entity), which is
// torch's / numpy's: interval map as multiply then add; 3x3 and 4x4 products as an fma chain over k (what both libraries' sgemm
// kernels compute for these sizes — checked by tests/test_api_cpu.py against numpy on random matrices); Euler angles through the
// C library's double cos / sin like Python's math module.  So the two paths agree bit for bit and the goldens pin both.
namespace {
inline void mm4(const float *A, const float *B, float *C) { // (through a temporary: C may alias A or B)
  float t[16];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      float acc = A[4 * i] * B[j];
      for (int k = 1; k < 4; ++k) acc = fmaf(A[4 * i + k], B[4 * k + j], acc);
      t[4 * i + j] = acc;
    }
  for (int i = 0; i < 16; ++i) C[i] = t[i];
}
} // namespace

// out = a @ b for row-major 4x4 float32 matrices as an fma chain over k.  The Python mirror of the entity classes forms every matrix
// product through this call, so that it and ffx_scene_randomize_h agree bit for bit on any host (numpy's / torch's own sgemm kernels
// round a 4x4 product differently from CPU to CPU: OpenBLAS picks its kernel per micro-architecture).  out may alias a or b.
extern "C" int ffx_mat4_mul_h(const float *a, const float *b, float *out) {
  if (!a || !b || !out) FFX_FAIL(FFX_ERR_ARG, "mat4_mul_h: bad argument");
  mm4(a, b, out);
  return FFX_OK;
}

extern "C" int ffx_scene_randomize_h(int n_samples, const uint64_t *seeds, const uint64_t *offsets, const ffx_rand_draw *draws, int n_draws,
                                     const ffx_rand_entity *ents, int n_ents, float *values, float *local, float *chain, float *chain_unc) {
  if (n_samples < 0 || n_draws < 0 || n_ents < 0 || (n_samples > 0 && (!seeds || !offsets)) || (n_draws > 0 && (!draws || !values)) ||
      (n_ents > 0 && (!ents || !local || !chain || !chain_unc)))
    FFX_FAIL(FFX_ERR_ARG, "scene_randomize_h: bad argument");
  for (int d = 0; d < n_draws; ++d)
    if (draws[d].n < 1 || draws[d].n > 4) FFX_FAIL(FFX_ERR_UNSUPPORTED, "scene_randomize_h: draw %d has %d values (1..4)", d, draws[d].n);
  for (int e = 0; e < n_ents; ++e) {
    const ffx_rand_entity &q = ents[e];
    if (q.parent >= e || q.kind < 0 || q.kind > 2 || q.draw_t >= n_draws || q.draw_r >= n_draws || q.draw_s >= n_draws)
      FFX_FAIL(FFX_ERR_ARG, "scene_randomize_h: entity %d: parents come first, draw rows must exist", e);
  }
  for (int s = 0; s < n_samples; ++s) {
    if (offsets[s] & 3u) FFX_FAIL(FFX_ERR_UNSUPPORTED, "scene_randomize_h: generator offset not a multiple of 4");
    float *val = values + (size_t)s * n_draws * 4;
    for (int d = 0; d < n_draws; ++d) {
      float u[4];
      uint64_t inc;
      const int rc = ffx_torch_rand_h(seeds[s], offsets[s] + 4u * (uint64_t)d, draws[d].n, u, &inc); // (every draw of <= 256 values advances the generator by 4)
      if (rc != FFX_OK) return rc;
      for (int i = 0; i < 4; ++i) {
        if (i < draws[d].n) {
          const float w = draws[d].hi[i] - draws[d].lo[i];
          const float p = u[i] * w;
          val[4 * d + i] = p + draws[d].lo[i];
        } else {
          val[4 * d + i] = 0.f;
        }
      }
    }
    float *loc = local + (size_t)s * n_ents * 16, *ch = chain + (size_t)s * n_ents * 16, *cu = chain_unc + (size_t)s * n_ents * 16;
    for (int e = 0; e < n_ents; ++e) {
      const ffx_rand_entity &q = ents[e];
      float *L = loc + 16 * e;
      if (q.kind == 0 || q.draw_t < 0 || q.draw_r < 0) { // not randomised: its world as it stands
        for (int i = 0; i < 16; ++i) L[i] = q.world[i];
      } else {
        const float *t = val + 4 * q.draw_t, *r = val + 4 * q.draw_r;
        // names as in the reference: the "z" slot feeds getPitchTransform (about Y), the "y" slot getYawTransform (about Z); Z @ Y @ X
        const float cz = (float)cos((double)r[2]), sz = (float)sin((double)r[2]);
        const float cy = (float)cos((double)r[1]), sy = (float)sin((double)r[1]);
        const float cx = (float)cos((double)r[0]), sx = (float)sin((double)r[0]);
        // (as 4x4 matrices with a zero border, like the mirror's: one product routine for everything, identical zero signs)
        const float zM[16] = {cz, 0.f, sz, 0.f, 0.f, 1.f, 0.f, 0.f, -sz, 0.f, cz, 0.f, 0.f, 0.f, 0.f, 1.f};
        const float yM[16] = {cy, -sy, 0.f, 0.f, sy, cy, 0.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 0.f, 1.f};
        const float xM[16] = {1.f, 0.f, 0.f, 0.f, 0.f, cx, -sx, 0.f, 0.f, sx, cx, 0.f, 0.f, 0.f, 0.f, 1.f};
        float R[16];
        mm4(zM, yM, R);
        mm4(R, xM, R);
        // T + centroid matrix (identity + translation, plus a matrix that is zero but for the centroid in its last column)
        const float TC[16] = {1.f, 0.f, 0.f, t[0] + q.centroid[0], 0.f, 1.f, 0.f, t[1] + q.centroid[1], 0.f, 0.f, 1.f, t[2] + q.centroid[2], 0.f, 0.f, 0.f, 1.f};
        float M[16];
        mm4(TC, R, M);
        if (q.kind == 2 && q.draw_s >= 0) {
          const float *sc = val + 4 * q.draw_s;
          const float S[16] = {sc[0], 0.f, 0.f, 0.f, 0.f, sc[1], 0.f, 0.f, 0.f, 0.f, sc[2], 0.f, 0.f, 0.f, 0.f, 1.f};
          mm4(M, S, M);
        }
        mm4(M, q.world, L);
      }
      float *Cn = ch + 16 * e;
      if (q.parent >= 0) mm4(ch + 16 * q.parent, L, Cn);
      else for (int i = 0; i < 16; ++i) Cn[i] = L[i];
      const float U[16] = {1.f, 0.f, 0.f, -q.centroid[0], 0.f, 1.f, 0.f, -q.centroid[1], 0.f, 0.f, 1.f, -q.centroid[2], 0.f, 0.f, 0.f, 1.f};
      mm4(Cn, U, cu + 16 * e);
    }
  }
  return FFX_OK;
}

// ---- one scene sample from the randomiser's tables to the device (include/ffx.h ffx_scene_step_h, ABI 8): the native params.update().
// The reference pushes a randomisation key by key through Mitsuba's parameter map (fireflies/scene.py:243-342) and lets
// params.update() (scene.py:384) rebuild; here the key writes are compiled into `ops` once and a sample is: copy the template
// description, run the ops over the drawn values and chain matrices, copy the material rows, enqueue the re-fit and the pre-pass.
int ffx_scene_update_h_top(void *bvh, const ffx_bvh_info *info, const float *src_verts, const int32_t *tris, const int32_t *tri_shape, const int32_t *vert_off,
                           const float *xform, int n_shapes, const ffx_smooth *smooth, int top, ffx_stream s); // ffx_scene.hip
extern "C" int ffx_scene_step_h(const ffx_step_plan *plan, const float *values, const float *chain, const float *chain_unc, const int32_t *frames,
                                const ffx_scene_desc *tmpl, ffx_scene_desc *sd_out, float *mat_rows, float *xform, int32_t *vert_off,
                                const ffx_step_geom *geom, int prepare_apex, ffx_stream stream) {
  if (!plan || !tmpl || !sd_out || !xform || !vert_off || plan->n_ops < 0 || (plan->n_ops > 0 && !plan->ops) || plan->n_shapes < 1 ||
      plan->n_shapes > FFX_MAX_SHAPES_H || plan->n_draws < 0 || plan->n_ents < 0 || (plan->n_draws > 0 && !values) ||
      (plan->n_ents > 0 && (!chain || !chain_unc)))
    FFX_FAIL(FFX_ERR_ARG, "scene_step_h: bad argument");
  if (plan->n_mat_floats < 0 || plan->n_mat_floats > FFX_MAX_MAT_H || (plan->n_mat_floats > 0 && !mat_rows))
    FFX_FAIL(FFX_ERR_ARG, "scene_step_h: material table of %d floats (at most %d, and then mat_rows must be given)", plan->n_mat_floats, FFX_MAX_MAT_H);
  if (tmpl->n_mat_h > 0 && tmpl->n_mat_h != plan->n_mat_floats)
    FFX_FAIL(FFX_ERR_ARG, "scene_step_h: the template carries %d material floats, the plan %d", tmpl->n_mat_h, plan->n_mat_floats);
  const int sd_words = (int)(sizeof(ffx_scene_desc) / sizeof(float));
  // every op checked before anything is written: a refused call leaves the caller's tables as they were
  for (int i = 0; i < plan->n_ops; ++i) {
    const ffx_step_op &o = plan->ops[i];
    bool ok = false;
    switch (o.kind) {
      case FFX_STEP_POSE_SD: ok = o.src >= 0 && o.src < plan->n_ents && o.dst >= 0 && o.dst + 16 <= sd_words; break;
      case FFX_STEP_VALUE_SD: ok = o.src >= 0 && o.src < plan->n_draws && o.comp >= 0 && o.comp < 4 && o.dst >= 0 && o.dst < sd_words; break;
      case FFX_STEP_VALUE_MAT: ok = o.src >= 0 && o.src < plan->n_draws && o.comp >= 0 && o.comp < 4 && o.dst >= 0 && o.dst < plan->n_mat_floats && (o.conv == 0 || o.conv == 1); break;
      case FFX_STEP_MESH: ok = o.src >= 0 && o.src < plan->n_ents && o.dst >= 0 && o.dst < plan->n_shapes && (o.mode == 0 || o.mode == 1); break;
      default: break;
    }
    if (!ok) FFX_FAIL(FFX_ERR_ARG, "scene_step_h: op %d (kind %d, src %d, comp %d, dst %d) out of range", i, o.kind, o.src, o.comp, o.dst);
  }
  if (frames) {
    if (!plan->frame_base || !plan->frame_stride || !plan->n_frames) FFX_FAIL(FFX_ERR_ARG, "scene_step_h: frames without the frame tables");
    for (int s = 0; s < plan->n_shapes; ++s)
      if (frames[s] >= plan->n_frames[s]) FFX_FAIL(FFX_ERR_ARG, "scene_step_h: shape %d: frame %d out of range [0, %d)", s, frames[s], plan->n_frames[s]);
  }
  if (sd_out != tmpl) *sd_out = *tmpl;
  float *sdw = reinterpret_cast<float *>(sd_out);
  for (int i = 0; i < plan->n_ops; ++i) {
    const ffx_step_op &o = plan->ops[i];
    switch (o.kind) {
      case FFX_STEP_POSE_SD:
        for (int j = 0; j < 16; ++j) sdw[o.dst + j] = chain[16 * o.src + j];
        break;
      case FFX_STEP_VALUE_SD: sdw[o.dst] = values[4 * o.src + o.comp]; break;
      case FFX_STEP_VALUE_MAT: {
        const float v = values[4 * o.src + o.comp];
        // (conv 1, [EXT Mitsuba principled.cpp parameters_changed]: in double, rounded to float once — the Python expression's bits)
        mat_rows[o.dst] = o.conv ? (float)(2.0 / (1.0 - sqrt(0.08 * (double)v)) - 1.0) : v;
        break;
      }
      default: { // FFX_STEP_MESH
        const float *m = (o.mode ? chain : chain_unc) + 16 * o.src;
        for (int j = 0; j < 16; ++j) xform[16 * o.dst + j] = m[j];
      }
    }
  }
  if (tmpl->n_mat_h > 0)
    for (int j = 0; j < plan->n_mat_floats; ++j) sd_out->mat_h[j] = mat_rows[j];
  if (frames)
    for (int s = 0; s < plan->n_shapes; ++s)
      if (frames[s] >= 0) vert_off[s] = plan->frame_base[s] + frames[s] * plan->frame_stride[s];
  if (!geom) return FFX_OK;
  if (!geom->bvh || !geom->info || !geom->src_verts || !geom->tris || !geom->tri_shape) FFX_FAIL(FFX_ERR_ARG, "scene_step_h: incomplete geometry block");
  // prepare_apex: bit 0 — the pre-pass behind the re-fit; bit 1 (FFX_STEP_DEFER_TOP) — the re-fit leaves the top of the tree to ffx_scene_refit_top
  const int rc = ffx_scene_update_h_top(geom->bvh, geom->info, geom->src_verts, geom->tris, geom->tri_shape, vert_off, xform, plan->n_shapes, geom->smooth,
                                        (prepare_apex & 2) ? 1 : 0, stream);
  if (rc != FFX_OK || !(prepare_apex & 1)) return rc;
  return ffx_apex_prepare(geom->bvh, geom->info, sd_out, stream);
}
