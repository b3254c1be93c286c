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
