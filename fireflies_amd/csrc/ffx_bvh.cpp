// ffx_bvh.cpp — host side of libffx_hip.so: error string, small host math, and the one-off
// BVH topology build (binned SAH).  The per-randomisation work (vertex transform, triangle
// records, bottom-up refit) runs on the GPU in ffx_scene.hip; this file is off the hot path.
//
// Replaces what Mitsuba does inside mi.load_file / params.update() (fireflies/scene.py:384) the
// first time a scene is seen [EXT]; there is no reference source for it.
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdlib>
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
  int sfirst = 0, scount = 0; // the subtree's run of leaf slots (for leaves: first, count)
  int parent = -1, side = 0;  // build-node id of the parent and which child this is
  int out_index = -1;         // index among the emitted (inner) nodes
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
    nodes[id].sfirst = first;
    nodes[id].scount = count;
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
    nodes[l].parent = id; nodes[l].side = 0;
    nodes[r].parent = id; nodes[r].side = 1;
    return id;
  }
};

inline int32_t leaf_code(int first, int count) { return ~(int32_t)(((uint32_t)first << 3) | (uint32_t)(count - 1)); }

// ---- 64-wide overlay (ffx_common.h: WideChild), built in layers over the binary tree.  A CLUSTER is a
// maximal subtree with at most FFX_WIDE triangles: its leaf slots are one contiguous run, so it needs no
// node of its own.  A wide node of layer k is a maximal subtree that contains at most FFX_WIDE items of
// layer k-1 (clusters for k = 1; items that stay alone simply move up a layer), and those items are its
// children.  Every child's box is the box of ONE binary node, which the binary refit already maintains
// (in that node's parent): the overlay only re-quantises them, and it inherits the SAH quality of the
// binary tree.  Typical fill: ~45 of 64 (53 k triangles: 1 198 clusters, 27 + 1 wide nodes).
struct WideNode { std::vector<int> kids; int bnode = -1; };

// reference = cluster << 31 | element << 6 | (count - 1): `element` indexes the array of 16-byte WideChild
// elements that starts at off_wnodes — wide node w is elements [64 w, 64 w + count), the cluster of leaf slots
// [first, first + count) is elements [tq0 + first, ...) with tq0 = 64 * n_wide
inline int32_t wide_elem_ref(bool cluster, uint32_t element, int count) { return (int32_t)((cluster ? 0x80000000u : 0u) | (element << 6) | (uint32_t)(count - 1)); }

// returns the wide nodes (children = build-node ids of their items), `wide_of[b]` = wide node rooted at
// build node b (-1: b is a cluster root or not an item) and the number of layers
void build_wide(const std::vector<BuildNode> &bn, int root, std::vector<WideNode> &wide, std::vector<int> &wide_of, int &depth_out) {
  const int n = (int)bn.size();
  wide.clear();
  wide_of.assign(n, -1);
  depth_out = 0;
  if (bn[root].scount <= FFX_WIDE) return; // the whole scene is one cluster
  std::vector<char> item(n, 0);
  for (int id = 0; id < n; ++id)
    if (bn[id].scount <= FFX_WIDE && (bn[id].parent < 0 || bn[bn[id].parent].scount > FFX_WIDE)) item[id] = 1;
  std::vector<int> cnt(n, 0);
  while (true) {
    // items below every node (children have larger ids than their parent); nodes inside an item count 0
    for (int id = n - 1; id >= 0; --id) {
      if (item[id]) cnt[id] = 1;
      else if (bn[id].left >= 0) cnt[id] = cnt[bn[id].left] + cnt[bn[id].right];
      else cnt[id] = 0;
    }
    ++depth_out;
    // new wide nodes: maximal subtrees with 2..FFX_WIDE items
    std::vector<int> roots;
    for (int id = 0; id < n; ++id) {
      if (item[id] || cnt[id] < 2 || cnt[id] > FFX_WIDE) continue;
      if (bn[id].parent >= 0 && cnt[bn[id].parent] <= FFX_WIDE) continue; // not maximal
      roots.push_back(id);
    }
    for (int rb : roots) {
      WideNode w;
      w.bnode = rb;
      std::vector<int> st{rb};
      while (!st.empty()) {
        const int id = st.back();
        st.pop_back();
        if (item[id]) { w.kids.push_back(id); item[id] = 0; continue; }
        if (bn[id].left >= 0 && cnt[id] > 0) { st.push_back(bn[id].right); st.push_back(bn[id].left); }
      }
      std::sort(w.kids.begin(), w.kids.end(), [&](int a, int b) { return bn[a].sfirst < bn[b].sfirst; });
      wide_of[rb] = (int)wide.size();
      wide.push_back(w);
    }
    for (int rb : roots) item[rb] = 1;
    if (item[root]) return;
  }
}

// Top-down alternative (default): a wide node takes the CUT of the binary subtree below it that a greedy SAH
// collapse picks — start with the two children, keep replacing the child with the largest box area (rest pose) that is
// not a cluster by its own two children until there are FFX_WIDE of them — and every child that is not a cluster
// becomes a wide node the same way.  The layered build above hands the root ~27 children of ~2 000 triangles
// each: ring-shaped pieces of a hollow scene whose boxes a ray along the tube enters one after the other
// (measured: 2.6 of 6.9 steps per closest-hit walk visit a node none of whose children is hit); the greedy cut gives
// the root 64 smaller, area-balanced children.  `by_count` (fallback when the area-greedy tree gets deeper than
// FFX_WIDE_MAX_DEPTH): split the child with the most triangles instead, which bounds the depth by log32.
void build_wide_topdown(const std::vector<BuildNode> &bn, const std::vector<float> &area, int root, bool by_count, std::vector<WideNode> &wide,
                        std::vector<int> &wide_of, int &depth_out) {
  // experiment knobs (host side, read at build time): largest cluster, priority = area * count^FFX_WIDE_COST_EXP
  const int CL = getenv("FFX_WIDE_CLUSTER") ? std::max(4, std::min(FFX_WIDE, atoi(getenv("FFX_WIDE_CLUSTER")))) : FFX_WIDE;
  const float cexp = getenv("FFX_WIDE_COST_EXP") ? (float)atof(getenv("FFX_WIDE_COST_EXP")) : 0.f;
  auto prio = [&](int id) { return cexp == 0.f ? area[id] : area[id] * std::pow((float)bn[id].scount, cexp); };
  wide.clear();
  wide_of.assign(bn.size(), -1);
  depth_out = 0;
  if (bn[root].scount <= CL) return; // the whole scene is one cluster
  std::vector<std::pair<int, int>> queue{{root, 1}};
  for (size_t qi = 0; qi < queue.size(); ++qi) {
    const int b = queue[qi].first, depth = queue[qi].second;
    WideNode w;
    w.bnode = b;
    w.kids = {bn[b].left, bn[b].right};
    while ((int)w.kids.size() < FFX_WIDE) {
      int best = -1;
      for (int k = 0; k < (int)w.kids.size(); ++k) {
        const int id = w.kids[k];
        if (bn[id].scount <= CL) continue; // a cluster stays whole
        if (best < 0) { best = k; continue; }
        const int bid = w.kids[best];
        const bool better = by_count ? (bn[id].scount > bn[bid].scount) : (prio(id) > prio(bid) || (prio(id) == prio(bid) && bn[id].scount > bn[bid].scount));
        if (better) best = k;
      }
      if (best < 0) break;
      const int id = w.kids[best];
      w.kids[best] = bn[id].left;
      w.kids.push_back(bn[id].right);
    }
    std::sort(w.kids.begin(), w.kids.end(), [&](int a, int c) { return bn[a].sfirst < bn[c].sfirst; });
    wide_of[b] = (int)wide.size();
    for (int id : w.kids)
      if (bn[id].scount > CL) queue.push_back({id, depth + 1});
    wide.push_back(std::move(w));
    depth_out = std::max(depth_out, depth);
  }
}

} // namespace

extern "C" {

const char *ffx_last_error(void) { return g_err; }
int ffx_abi_version(void) { return FFX_ABI_VERSION; }
const char *ffx_backend(void) { return "hip-gfx950"; }

size_t ffx_bvh_blob_bytes(int n_tris) {
  size_t f = n_tris < 1 ? 1 : (size_t)n_tris;
  // wide overlay: at most f / 32 + 2 wide nodes (every wide node but the root's chain has >= 2 children
  // of more than FFX_WIDE triangles or is the parent of clusters; bounded generously), 64 x (16 + 4) B each
  const size_t wmax = f / 16 + 4;
  return 64 + f * sizeof(BvhNode) + f * 4 + f * 4 + 64 + (f + FFX_LEAF_MAX) * sizeof(TriRec) + 64 + FFX_N_APEX * (size_t)ffx_apex_stride(n_tris < 1 ? 1 : n_tris) +
         wmax * FFX_WIDE * (sizeof(WideChild) + 4) + (f + FFX_WIDE) * sizeof(WideChild) + 256 +
         // refit plan: <= f + 1 headers of 8 ints (every leaf under the top could be a treelet of its own), <= f nodes with <= 2 level
         // entries each, <= wmax * 64 wide children, the counter
         ((f + 2) * 8 + 3 * f + 8 + wmax * FFX_WIDE + 64) * 4 + (f + FFX_LEAF_MAX) * (48 + 16) + 192 +
         // tile bins of the three apexes (ffx_common.h)
         FFX_N_APEX * (size_t)ffx_bin_stride(n_tris < 1 ? 1 : n_tris) + 64;
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
  // 64-wide overlay (placed before the apex areas, which stay the LAST areas of the blob)
  // subtree slot runs of inner nodes (children have larger ids than their parent)
  for (int id = (int)b.nodes.size() - 1; id >= 0; --id) {
    BuildNode &n = b.nodes[id];
    if (n.left >= 0) { n.sfirst = b.nodes[n.left].sfirst; n.scount = b.nodes[n.left].scount + b.nodes[n.right].scount; }
  }
  std::vector<WideNode> wide;
  std::vector<int> wide_of;
  int wide_depth = 0;
  {
    // FFX_WIDE_BUILD=layers|area|count selects the overlay builder (default area; results do not depend on it)
    const char *mode = getenv("FFX_WIDE_BUILD");
    const size_t wcap = (size_t)n_tris / 16 + 4; // what ffx_bvh_blob_bytes reserves
    bool done = false;
    if (!mode || strcmp(mode, "layers") != 0) {
      std::vector<float> area(b.nodes.size(), 0.f);
      std::vector<Box> nb(b.nodes.size());
      for (int id = (int)b.nodes.size() - 1; id >= 0; --id) { // children have larger ids than their parent
        const BuildNode &n = b.nodes[id];
        nb[id].reset();
        if (n.left < 0) for (int k = n.first; k < n.first + n.count; ++k) nb[id].grow(tb[order[k]]);
        else { nb[id].grow(nb[n.left]); nb[id].grow(nb[n.right]); }
        area[id] = nb[id].half_area();
      }
      const bool by_count = mode && strcmp(mode, "count") == 0;
      build_wide_topdown(b.nodes, area, root, by_count, wide, wide_of, wide_depth);
      if (!by_count && (wide_depth > FFX_WIDE_MAX_DEPTH || wide.size() > wcap)) build_wide_topdown(b.nodes, area, root, true, wide, wide_of, wide_depth);
      done = wide_depth <= FFX_WIDE_MAX_DEPTH && wide.size() <= wcap;
    }
    if (!done) build_wide(b.nodes, root, wide, wide_of, wide_depth);
  }
  if (wide_depth > FFX_WIDE_MAX_DEPTH) FFX_FAIL(FFX_ERR_UNSUPPORTED, "bvh_build_host: wide tree depth %d exceeds %d", wide_depth, FFX_WIDE_MAX_DEPTH);
  info->n_wide = (int32_t)wide.size();
  info->wide_depth = wide_depth;
  info->off_wnodes = off; // wide nodes and triangle boxes form ONE array of 16-byte elements (see wide_elem_ref)
  off += (uint64_t)wide.size() * FFX_WIDE * sizeof(WideChild);
  info->off_tq = off;
  off += ((uint64_t)n_tris + FFX_WIDE) * sizeof(WideChild); // 64 elements of padding: every lane of a cluster step reads one
  off = (off + 63) & ~(uint64_t)63;
  info->off_wsrc = off;
  off += (uint64_t)wide.size() * FFX_WIDE * 4;
  off = (off + 63) & ~(uint64_t)63;
  info->off_whdr = off;
  off += 64;
  // ---- refit plan (ffx.h: off_plan).  Treelet roots: the highest inner nodes with at most `tl_max` triangles below them; a
  // leaf hanging directly off the top forms a treelet without nodes (its records still have to be built).
  std::vector<int32_t> plan;
  int n_treelets = 0;
  {
    // treelet size: <= 1024 triangles, grown (powers of two, up to 8192) until ~128-256 treelets are left — the top of the tree above
    // them is re-fitted by ONE workgroup after the last treelet has arrived, a serial tail that grows with their number
    // (tools/refittime.py, MI355X: colon, 524 k triangles: 3004 / 1519 / 751 / 379 / 191 / 95 treelets = 396 / 205 / 134 / 106 /
    // 100 / 128 us per update; vocal fold, 53 k: 302 / 152 / 76 / 38 / 21 treelets = 55 / 44 / 40 / 45 / 50 us)
    // (round 4: the update is two launches — treelets, then the top — without the per-workgroup release fence that made many treelets
    // expensive; 1024 triangles per treelet is then best for both scenes: colon 751 treelets 65 us (fused, 4096: 83), vocal fold 76: 36 us;
    // FFX_REFIT=fused with the sizes above remains the A/B baseline)
    int tl_auto = 512; // (one WAVE per treelet since the end of round 4: 512 / 1024 measure the same, 256 slightly worse)
    const int tl_max = getenv("FFX_TREELET_TRIS") ? std::max(FFX_LEAF_MAX, atoi(getenv("FFX_TREELET_TRIS"))) : tl_auto;
    // out-index heights (as the level refit uses them) per build node
    std::vector<int> hgt(b.nodes.size(), 0);
    for (int id = (int)b.nodes.size() - 1; id >= 0; --id) { // children have larger ids than their parent
      const BuildNode &n = b.nodes[id];
      if (n.left >= 0) hgt[id] = std::max(b.nodes[n.left].left >= 0 ? hgt[n.left] + 1 : 0, b.nodes[n.right].left >= 0 ? hgt[n.right] + 1 : 0);
    }
    std::vector<int> tl_of(b.nodes.size(), -1); // treelet of an inner build node; -2: top
    struct TL { int root, sfirst, scount; std::vector<int> nodes; std::vector<int> wc; };
    std::vector<TL> tls;
    std::vector<int> top_nodes;
    if (root_is_leaf) {
      tls.push_back({root, 0, n_tris, {}, {}});
    } else {
      std::vector<int> st{root};
      while (!st.empty()) {
        const int id = st.back();
        st.pop_back();
        const BuildNode &n = b.nodes[id];
        if (n.left < 0) { tls.push_back({id, n.first, n.count, {}, {}}); continue; } // a leaf directly under the top
        if (n.scount <= tl_max) {
          TL t{id, n.sfirst, n.scount, {}, {}};
          std::vector<int> sub{id};
          while (!sub.empty()) {
            const int x = sub.back();
            sub.pop_back();
            if (b.nodes[x].left < 0) continue;
            tl_of[x] = (int)tls.size();
            t.nodes.push_back(x);
            sub.push_back(b.nodes[x].left);
            sub.push_back(b.nodes[x].right);
          }
          tls.push_back(std::move(t));
        } else {
          tl_of[id] = -2;
          top_nodes.push_back(id);
          st.push_back(n.right);
          st.push_back(n.left);
        }
      }
    }
    n_treelets = (int)tls.size();
    TL top{-1, 0, 0, top_nodes, {}};
    // wide children go with the treelet (or the top) that owns the binary node holding their box: the PARENT of the child's
    // build node (ffx_scene.hip copies wn[k] from node wsrc[k] >> 1)
    for (size_t w = 0; w < wide.size(); ++w)
      for (int j = 0; j < (int)wide[w].kids.size(); ++j) {
        const int par = b.nodes[wide[w].kids[j]].parent;
        const int t = tl_of[par];
        (t >= 0 ? tls[t] : top).wc.push_back((int)(w * FFX_WIDE + j));
      }
    tls.push_back(std::move(top));
    // emit: headers, level starts, node lists (by height), wide children, counter
    const size_t n_hdr = tls.size();
    plan.assign(n_hdr * 8, 0);
    std::vector<int32_t> lvls, nodes_out, wc_out;
    for (size_t t = 0; t < n_hdr; ++t) {
      TL &tl = tls[t];
      const bool single = root_is_leaf && t == 0;
      std::vector<int> ids = tl.nodes;
      std::stable_sort(ids.begin(), ids.end(), [&](int a, int c) { return hgt[a] < hgt[c]; });
      int32_t *h = &plan[t * 8];
      h[0] = tl.sfirst; h[1] = tl.scount;
      h[2] = (int32_t)lvls.size();
      int n_l = 0;
      if (single) { // the one node of a scene that is a single leaf
        lvls.push_back((int32_t)nodes_out.size());
        nodes_out.push_back(0);
        n_l = 1;
      }
      for (size_t i = 0; i < ids.size(); ++i) {
        if (i == 0 || hgt[ids[i]] != hgt[ids[i - 1]]) { lvls.push_back((int32_t)nodes_out.size()); ++n_l; }
        nodes_out.push_back(b.nodes[ids[i]].out_index);
      }
      lvls.push_back((int32_t)nodes_out.size()); // end of the last level
      h[3] = n_l;
      h[4] = (int32_t)wc_out.size(); h[5] = (int32_t)tl.wc.size();
      for (int k : tl.wc) wc_out.push_back(k);
    }
    // absolute int offsets inside the plan: [headers][levels][nodes][wide children][counter]
    const int32_t o_lvl = (int32_t)plan.size(), o_nodes = o_lvl + (int32_t)lvls.size(), o_wc = o_nodes + (int32_t)nodes_out.size();
    for (size_t t = 0; t < n_hdr; ++t) { plan[t * 8 + 2] += o_lvl; plan[t * 8 + 4] += o_wc; }
    for (int32_t &v : lvls) v += o_nodes;
    plan.insert(plan.end(), lvls.begin(), lvls.end());
    plan.insert(plan.end(), nodes_out.begin(), nodes_out.end());
    plan.insert(plan.end(), wc_out.begin(), wc_out.end());
    plan.push_back(0); // arrival counter
  }
  off = (off + 63) & ~(uint64_t)63;
  info->off_plan = off;
  info->n_treelets = n_treelets;
  info->plan_ints = (int32_t)plan.size();
  off += (uint64_t)plan.size() * 4;
  off = (off + 63) & ~(uint64_t)63;
  info->off_nrec = off; // per-slot vertex normals of the shapes of ffx_smooth (ffx.h); untouched for flat scenes
  off += ((uint64_t)n_tris + FFX_LEAF_MAX) * 48;
  off = (off + 63) & ~(uint64_t)63;
  info->off_gn = off; // per-slot unit geometric normals
  off += ((uint64_t)n_tris + FFX_LEAF_MAX) * 16;
  off = (off + 63) & ~(uint64_t)63;
  // ---- everything from here on is SCRATCH of the render calls' pre-pass (written before it is read, never read by the host): the
  // tile bins and, last, the apex-record areas.  The host copy of the blob only needs its first off_bins bytes.
  info->off_bins = off;
  info->bins_stride = ffx_bin_stride(n_tris);
  const uint64_t static_bytes = off;
  off += FFX_N_APEX * info->bins_stride;
  off = (off + 63) & ~(uint64_t)63;
  off += FFX_N_APEX * ffx_apex_stride(n_tris); // apex-record areas (ffx_common.h)
  info->total_bytes = off;
  if (off > blob_bytes) FFX_FAIL(FFX_ERR_NOMEM, "bvh_build_host: internal size error");
  if (info->max_depth > FFX_STACK_DEPTH) FFX_FAIL(FFX_ERR_UNSUPPORTED, "bvh_build_host: tree depth %d exceeds the traversal stack", info->max_depth);

  memset(blob, 0, (size_t)static_bytes); // (the scratch tail is the device's business)
  memcpy((char *)blob + info->off_plan, plan.data(), plan.size() * 4);
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

  // wide overlay: child references and, per child, where its box lives in the binary tree
  // (binary node index * 2 + side of the child's parent; -1 for unused lanes)
  WideChild *wn = (WideChild *)((char *)blob + info->off_wnodes);
  int32_t *wsrc = (int32_t *)((char *)blob + info->off_wsrc);
  const uint32_t tq0 = (uint32_t)wide.size() * FFX_WIDE;
  if ((uint64_t)tq0 + (uint64_t)n_tris >= (1u << 25)) FFX_FAIL(FFX_ERR_UNSUPPORTED, "bvh_build_host: more than 2^25 wide elements");
  if (wide.empty()) {
    info->wide_root = wide_elem_ref(true, tq0, n_tris);
  } else {
    const int wroot = wide_of[root];
    info->wide_root = wide_elem_ref(false, (uint32_t)wroot * FFX_WIDE, (int)wide[wroot].kids.size());
    for (size_t w = 0; w < wide.size(); ++w) {
      const std::vector<int> &kids = wide[w].kids;
      for (int j = 0; j < FFX_WIDE; ++j) {
        WideChild &c = wn[w * FFX_WIDE + j];
#if FFX_WIDE_F32
        c.lo[0] = c.lo[1] = c.lo[2] = 3.0e38f; // inverted box: lanes beyond the child count are masked off anyway
        c.hi0 = c.hi12[0] = c.hi12[1] = -3.0e38f;
        c.pad = 0;
#else
        c.q[0] = c.q[1] = c.q[2] = 0xffff; // inverted box: lanes beyond the child count are masked off anyway
        c.q[3] = c.q[4] = c.q[5] = 0;
#endif
        c.ref = 0;
        wsrc[w * FFX_WIDE + j] = -1;
        if (j >= (int)kids.size()) continue;
        const BuildNode &k = b.nodes[kids[j]];
        const int kw = wide_of[kids[j]];
        c.ref = kw >= 0 ? wide_elem_ref(false, (uint32_t)kw * FFX_WIDE, (int)wide[kw].kids.size()) : wide_elem_ref(true, tq0 + (uint32_t)k.sfirst, k.scount);
        wsrc[w * FFX_WIDE + j] = b.nodes[k.parent].out_index * 2 + k.side;
      }
    }
  }
  return FFX_OK;
}

} // extern "C"
