// ffx_bvh.cpp — host side of libffx_hip.so: error string, small host math, and the one-off
// BVH topology build (binned SAH).  The per-randomisation work (vertex transform, triangle
// records, bottom-up refit) runs on the GPU in ffx_scene.hip; this file is off the hot path.
//
// Replaces what Mitsuba does inside mi.load_file / params.update() (fireflies/scene.py:384) the
// first time a scene is seen [EXT]; there is no reference source for it.
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstring>
#include <limits>
#include <vector>

#include "ffx_common.h"

static thread_local char g_err[512] = "";

void ffx_set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
}

int ffx_inv4(const float *mf, float *outf) {
  double m[16], inv[16];
  for (int i = 0; i < 16; ++i) m[i] = mf[i];
  inv[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] + m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
  inv[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] - m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
  inv[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] + m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
  inv[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] - m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
  inv[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] - m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
  inv[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] + m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
  inv[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] - m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
  inv[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] + m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
  inv[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] + m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
  inv[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] - m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
  inv[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] + m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
  inv[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] - m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
  inv[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] - m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
  inv[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] + m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
  inv[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] - m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
  inv[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] + m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
  double det = m[0] * inv[0] + m[1] * inv[4] + m[2] * inv[8] + m[3] * inv[12];
  if (det == 0.0) return 0;
  det = 1.0 / det;
  for (int i = 0; i < 16; ++i) outf[i] = (float)(inv[i] * det);
  return 1;
}

namespace {

struct Box {
  float lo[3], hi[3];
  void reset() {
    for (int a = 0; a < 3; ++a) { lo[a] = std::numeric_limits<float>::infinity(); hi[a] = -std::numeric_limits<float>::infinity(); }
  }
  void grow(const Box &b) {
    for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], b.lo[a]); hi[a] = std::max(hi[a], b.hi[a]); }
  }
  void grow(const float *p) {
    for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], p[a]); hi[a] = std::max(hi[a], p[a]); }
  }
  float half_area() const {
    float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    if (dx < 0 || dy < 0 || dz < 0) return 0.f;
    return dx * dy + dy * dz + dz * dx;
  }
};

struct BuildNode {
  int left = -1, right = -1; // build-node indices, -1 for a leaf
  int first = 0, count = 0;
  int out_index = -1;        // index among the emitted (inner) nodes
  int height = 0;
};

struct Builder {
  const std::vector<Box> &tb;
  const std::vector<float> &cent;
  std::vector<int> &order;
  std::vector<BuildNode> nodes;
  int max_depth = 0;
  static constexpr int kBins = 16;
  static constexpr int kDepthLimit = FFX_STACK_DEPTH - 4;

  Builder(const std::vector<Box> &tb_, const std::vector<float> &c_, std::vector<int> &o_) : tb(tb_), cent(c_), order(o_) {}

  int build(int first, int count, int depth) {
    int id = (int)nodes.size();
    nodes.emplace_back();
    nodes[id].first = first;
    nodes[id].count = count;
    max_depth = std::max(max_depth, depth);
    if (count <= FFX_LEAF_MAX) return id;

    Box cb;
    cb.reset();
    for (int i = 0; i < count; ++i) cb.grow(&cent[3 * order[first + i]]);

    int log2c = 0;
    while ((1 << log2c) < count) ++log2c;
    bool force_median = depth + log2c >= kDepthLimit;

    int best_axis = -1, best_bin = -1;
    float best_cost = std::numeric_limits<float>::infinity();
    if (!force_median) {
      for (int axis = 0; axis < 3; ++axis) {
        float ext = cb.hi[axis] - cb.lo[axis];
        if (!(ext > 0.f)) continue;
        Box bb[kBins];
        int bc[kBins];
        for (int b = 0; b < kBins; ++b) { bb[b].reset(); bc[b] = 0; }
        float scale = (float)kBins / ext;
        for (int i = 0; i < count; ++i) {
          int t = order[first + i];
          int b = std::min(kBins - 1, std::max(0, (int)((cent[3 * t + axis] - cb.lo[axis]) * scale)));
          bb[b].grow(tb[t]);
          bc[b]++;
        }
        float right_area[kBins];
        int right_cnt[kBins];
        Box acc;
        acc.reset();
        int c = 0;
        for (int b = kBins - 1; b > 0; --b) {
          acc.grow(bb[b]);
          c += bc[b];
          right_area[b] = acc.half_area();
          right_cnt[b] = c;
        }
        acc.reset();
        c = 0;
        for (int b = 0; b < kBins - 1; ++b) {
          acc.grow(bb[b]);
          c += bc[b];
          if (c == 0 || right_cnt[b + 1] == 0) continue;
          float cost = acc.half_area() * (float)c + right_area[b + 1] * (float)right_cnt[b + 1];
          if (cost < best_cost) { best_cost = cost; best_axis = axis; best_bin = b; }
        }
      }
    }
    int mid;
    if (best_axis >= 0) {
      float ext = cb.hi[best_axis] - cb.lo[best_axis];
      float scale = (float)kBins / ext;
      float lo = cb.lo[best_axis];
      int axis = best_axis, bin = best_bin;
      auto it = std::partition(order.begin() + first, order.begin() + first + count, [&](int t) {
        int b = std::min(kBins - 1, std::max(0, (int)((cent[3 * t + axis] - lo) * scale)));
        return b <= bin;
      });
      mid = (int)(it - order.begin());
    } else {
      // all centroids coincide, or the depth budget is nearly used up: object median
      int axis = 0;
      for (int a = 1; a < 3; ++a)
        if (cb.hi[a] - cb.lo[a] > cb.hi[axis] - cb.lo[axis]) axis = a;
      mid = first + count / 2;
      std::nth_element(order.begin() + first, order.begin() + mid, order.begin() + first + count, [&](int x, int y) {
        float cx = cent[3 * x + axis], cy = cent[3 * y + axis];
        return cx < cy || (cx == cy && x < y);
      });
    }
    if (mid == first || mid == first + count) mid = first + count / 2;
    int l = build(first, mid - first, depth + 1);
    int r = build(mid, first + count - mid, depth + 1);
    nodes[id].left = l;
    nodes[id].right = r;
    nodes[id].count = 0;
    return id;
  }
};

inline int32_t leaf_code(int first, int count) { return ~(int32_t)(((uint32_t)first << 3) | (uint32_t)(count - 1)); }

} // namespace

extern "C" {

const char *ffx_last_error(void) { return g_err; }
int ffx_abi_version(void) { return FFX_ABI_VERSION; }
const char *ffx_backend(void) { return "hip-gfx950"; }

size_t ffx_bvh_blob_bytes(int n_tris) {
  size_t f = n_tris < 1 ? 1 : (size_t)n_tris;
  return 64 + f * sizeof(BvhNode) + f * 4 + f * 4 + 64 + (f + FFX_LEAF_MAX) * sizeof(TriRec) + 64 + FFX_N_APEX * (size_t)ffx_apex_stride(n_tris < 1 ? 1 : n_tris);
}

int ffx_bvh_build_host(const float *verts, int n_verts, const int32_t *tris, int n_tris, void *blob, size_t blob_bytes, ffx_bvh_info *info) {
  if (!verts || !tris || !blob || !info || n_tris < 1 || n_verts < 1) FFX_FAIL(FFX_ERR_ARG, "bvh_build_host: bad argument");
  if (n_tris >= (1 << 28)) FFX_FAIL(FFX_ERR_UNSUPPORTED, "bvh_build_host: more than 2^28 triangles");
  if (blob_bytes < ffx_bvh_blob_bytes(n_tris)) FFX_FAIL(FFX_ERR_NOMEM, "bvh_build_host: blob too small (%zu < %zu)", blob_bytes, ffx_bvh_blob_bytes(n_tris));
  for (long i = 0; i < 3L * n_tris; ++i)
    if (tris[i] < 0 || tris[i] >= n_verts) FFX_FAIL(FFX_ERR_ARG, "bvh_build_host: vertex index %d out of range at %ld", tris[i], i);

  std::vector<Box> tb(n_tris);
  std::vector<float> cent(3 * (size_t)n_tris);
  std::vector<int> order(n_tris);
  for (int t = 0; t < n_tris; ++t) {
    tb[t].reset();
    for (int c = 0; c < 3; ++c) tb[t].grow(verts + 3 * (size_t)tris[3 * t + c]);
    for (int a = 0; a < 3; ++a) cent[3 * t + a] = 0.5f * (tb[t].lo[a] + tb[t].hi[a]);
    order[t] = t;
  }
  Builder b(tb, cent, order);
  b.nodes.reserve(2 * (size_t)n_tris / 2 + 16);
  int root = b.build(0, n_tris, 0);

  // emit inner nodes in pre-order
  std::vector<int> emit; // build-node ids of inner nodes in output order
  {
    std::vector<int> st{root};
    while (!st.empty()) {
      int id = st.back();
      st.pop_back();
      if (b.nodes[id].left < 0) continue;
      b.nodes[id].out_index = (int)emit.size();
      emit.push_back(id);
      st.push_back(b.nodes[id].right);
      st.push_back(b.nodes[id].left);
    }
  }
  bool root_is_leaf = b.nodes[root].left < 0;
  int n_nodes = root_is_leaf ? 1 : (int)emit.size();

  memset(info, 0, sizeof *info);
  info->n_tris = n_tris;
  info->n_nodes = n_nodes;
  info->max_depth = b.max_depth + 1;
  uint64_t off = 64;
  info->off_nodes = off;
  off += (uint64_t)n_nodes * sizeof(BvhNode);
  info->off_order = off;
  off += (uint64_t)n_tris * 4;
  info->off_refit = off;
  off += (uint64_t)n_nodes * 4;
  off = (off + 63) & ~(uint64_t)63;
  info->off_recs = off;
  off += ((uint64_t)n_tris + FFX_LEAF_MAX) * sizeof(TriRec); // tail padding: leaf fetches always read FFX_LEAF_MAX records
  off = (off + 63) & ~(uint64_t)63;
  off += FFX_N_APEX * ffx_apex_stride(n_tris); // apex-record areas (ffx_common.h), zero until a render call fills them
  info->total_bytes = off;
  if (off > blob_bytes) FFX_FAIL(FFX_ERR_NOMEM, "bvh_build_host: internal size error");
  if (info->max_depth > FFX_STACK_DEPTH) FFX_FAIL(FFX_ERR_UNSUPPORTED, "bvh_build_host: tree depth %d exceeds the traversal stack", info->max_depth);

  memset(blob, 0, (size_t)off);
  BvhNode *out = (BvhNode *)((char *)blob + info->off_nodes);
  int32_t *ord = (int32_t *)((char *)blob + info->off_order);
  int32_t *refit = (int32_t *)((char *)blob + info->off_refit);
  memcpy(ord, order.data(), (size_t)n_tris * 4);

  const float inf = std::numeric_limits<float>::infinity();
  auto empty_boxes = [&](BvhNode &n) {
    for (int a = 0; a < 3; ++a) { n.lo0[a] = n.lo1[a] = inf; n.hi0[a] = n.hi1[a] = -inf; }
  };
  std::vector<int> height(n_nodes, 0);
  if (root_is_leaf) {
    empty_boxes(out[0]);
    out[0].c0 = leaf_code(0, n_tris);
    out[0].c1 = FFX_EMPTY_CHILD;
  } else {
    // heights bottom-up: emit[] is pre-order, so children come after parents
    for (int k = (int)emit.size() - 1; k >= 0; --k) {
      const BuildNode &bn = b.nodes[emit[k]];
      BvhNode &n = out[k];
      empty_boxes(n);
      int h = 0;
      const BuildNode &l = b.nodes[bn.left], &r = b.nodes[bn.right];
      if (l.left < 0) n.c0 = leaf_code(l.first, l.count);
      else { n.c0 = l.out_index; h = std::max(h, height[l.out_index] + 1); }
      if (r.left < 0) n.c1 = leaf_code(r.first, r.count);
      else { n.c1 = r.out_index; h = std::max(h, height[r.out_index] + 1); }
      height[k] = h;
    }
  }
  int max_h = 0;
  for (int k = 0; k < n_nodes; ++k) max_h = std::max(max_h, height[k]);
  if (max_h + 1 > FFX_MAX_LEVELS) FFX_FAIL(FFX_ERR_UNSUPPORTED, "bvh_build_host: %d refit levels exceed FFX_MAX_LEVELS", max_h + 1);
  info->n_levels = max_h + 1;
  std::vector<int> cnt(max_h + 2, 0);
  for (int k = 0; k < n_nodes; ++k) cnt[height[k] + 1]++;
  for (int h = 0; h <= max_h; ++h) cnt[h + 1] += cnt[h];
  for (int h = 0; h <= max_h + 1; ++h) info->level_start[h] = cnt[h];
  std::vector<int> cur(cnt.begin(), cnt.end() - 1);
  for (int k = 0; k < n_nodes; ++k) refit[cur[height[k]]++] = k;
  return FFX_OK;
}

} // extern "C"
