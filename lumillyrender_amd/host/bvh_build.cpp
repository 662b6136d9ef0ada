// bvh_build.cpp -- host SAH BVH builder for the device traversal kernels.
//
// Replaces BVH::construct (bvh.rs:69-127): same full-sweep SAH over the three axes with the
// reference's cost model T = 2*T_aabb + (A(S1)*N(S1) + A(S2)*N(S2)) * T_tri / A(S), T_aabb = 1,
// T_tri = 2 (bvh.rs:71-72,111), decided on the primitives' EXACT boxes (triangle.rs:102-118, sphere.rs:31-38)
// with the reference's sequence of sorts -- x, y, z, then the chosen axis, each one STABLE (the reference's
// sort_unstable_by_key leaves the order of equal centres open; a stable sort is one valid outcome, and the
// one the parity tests' CPU restatement takes).
//
// THE LEAF ORDER IS THE REFERENCE'S CANDIDATE ORDER.  bvh.rs:131-141 keeps the FIRST minimum of the candidate
// list, and Node::may_intersect (bvh.rs:38-45) fills that list left subtree first: on an exact distance tie
// the primitive that comes first in a depth-first walk of the reference's tree wins.  `order` (the
// description's bvh_prim_order) lists the primitives in exactly that order -- lr_scene_create numbers them by
// it, so "lowest device id" IS "first candidate"  -- including inside a leaf of several
// primitives, whose range is ordered by the reference's own recursion (ref_order).
//
// Differences from the reference's tree, none of which can change an image:
//   * leaves may hold up to `max_leaf` primitives when the SAH says splitting does not pay
//     (the reference always splits down to one, bvh.rs:76-78);
//   * nodes are emitted in the flat two-boxes-per-node layout of LrBvhNode;
//   * every STORED box is padded (and a sphere's grown) so that the device slab test is strictly conservative:
//     the tree only decides how many primitive tests run; whether a primitive is a candidate is decided by
//     the literal slab test on its own exact box, on the device (bvh.rs:20-25, csrc/lr_kernels.h);
//   * below depth 32 the split falls back to the object median to bound the traversal stack; the leaf order
//     of such a subtree is the median split's, not the reference's (exact ties only, pathological inputs only).
#include "host_internal.h"
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>

namespace lrhost {
namespace {

struct Box {
  float mn[3], mx[3];
  void reset() { for (int a = 0; a < 3; ++a) { mn[a] = INFINITY; mx[a] = -INFINITY; } }
  void grow(const Box& b) { for (int a = 0; a < 3; ++a) { mn[a] = std::fmin(mn[a], b.mn[a]); mx[a] = std::fmax(mx[a], b.mx[a]); } }
  float area() const {                                        // aabb.rs:17-28
    float sx = std::fabs(mx[0] - mn[0]), sy = std::fabs(mx[1] - mn[1]), sz = std::fabs(mx[2] - mn[2]);
    return 2.0f * (sx * sy + sy * sz + sz * sx);
  }
};

Box prim_box(const LrPrimitive& p) {
  Box b;
  if (p.type == LR_PRIM_TRIANGLE) {                           // triangle.rs:102-118
    for (int a = 0; a < 3; ++a) {
      b.mn[a] = std::fmin(std::fmin(p.v[a], p.v[3 + a]), p.v[6 + a]);
      b.mx[a] = std::fmax(std::fmax(p.v[a], p.v[3 + a]), p.v[6 + a]);
    }
  } else {                                                    // sphere.rs:31-38
    for (int a = 0; a < 3; ++a) { b.mn[a] = p.v[a] - p.v[3]; b.mx[a] = p.v[a] + p.v[3]; }
  }
  return b;
}

struct Builder {
  const std::vector<Box>& boxes;        // exact boxes: every SAH decision and the order
  const std::vector<Box>& store;        // what is stored: spheres grown (build_bvh)
  std::vector<float> centre[3];
  std::vector<int> idx;
  std::vector<float> pre, suf;
  BvhResult& out;
  int max_leaf;
  float root_area = 1.0f;
  static constexpr float T_AABB = 1.0f, T_TRI = 2.0f;
  static constexpr int MEDIAN_DEPTH = 32;

  Builder(const std::vector<Box>& b, const std::vector<Box>& st, BvhResult& o, int ml) : boxes(b), store(st), out(o), max_leaf(ml) {}

  void sort_axis(int lo, int hi, int axis) {                   // bvh.rs:83-86,117-119 (stable: see the header)
    const std::vector<float>& c = centre[axis];
    std::stable_sort(idx.begin() + lo, idx.begin() + hi, [&c](int a, int b) { return c[a] < c[b]; });
  }
  Box range_box(int lo, int hi) const { Box b; b.reset(); for (int i = lo; i < hi; ++i) b.grow(boxes[idx[i]]); return b; }
  Box range_store(int lo, int hi) const { Box b; b.reset(); for (int i = lo; i < hi; ++i) b.grow(store[idx[i]]); return b; }

  // bvh.rs:80-115: the best (axis, split) of idx[lo, hi) under the SAH, first minimum over the splits of an axis and over the
  // axes; leaves the range sorted by z.  n >= 2.
  void find_split(int lo, int hi, const Box& box, int* axis_out, int* k_out, float* cost_out) {
    const int n = hi - lo;
    int best_axis = -1, best_k = -1; float best_cost = 0.0f;
    const float s_a = box.area();
    for (int axis = 0; axis < 3; ++axis) {
      sort_axis(lo, hi, axis);
      Box b; b.reset();
      for (int i = 0; i < n; ++i) { b.grow(boxes[idx[lo + i]]); pre[i] = b.area(); }
      b.reset();
      for (int i = n - 1; i >= 1; --i) { b.grow(boxes[idx[lo + i]]); suf[i] = b.area(); }
      for (int i = 0; i < n - 1; ++i) {                   // left = [0..i], right = [i+1..n)
        float c = 2.0f * T_AABB + (pre[i] * (float)(i + 1) + suf[i + 1] * (float)(n - i - 1)) * T_TRI / s_a;
        if (!(c == c)) c = INFINITY;                      // zero-area parents (degenerate input)
        if (best_axis < 0 || c < best_cost) { best_cost = c; best_axis = axis; best_k = i + 1; }
      }
    }
    *axis_out = best_axis; *k_out = best_k; *cost_out = best_cost;
  }
  // idx[lo, hi) in the order a depth-first walk of the REFERENCE's subtree over it visits the leaves (bvh.rs:69-127 down to
  // single primitives, no node emitted): the candidate order of bvh.rs:38-45 inside a leaf of several primitives
  void ref_order(int lo, int hi) {
    if (hi - lo < 2) return;
    int axis, k; float cost;
    find_split(lo, hi, range_box(lo, hi), &axis, &k, &cost);
    sort_axis(lo, hi, axis);
    ref_order(lo, lo + k); ref_order(lo + k, hi);
  }

  int32_t make_leaf(int lo, int hi, const Box& box, int split_axis = 0, int split_k = 0, bool median_below = false) {
    int first = (int)out.order.size();
    if (!median_below && hi - lo > 1) {
      // inside a leaf: the reference's candidate order (see the header).  The caller's find_split of this range IS the reference's
      // sweep (a second one would start from another order of the equal centres): apply its split, recurse as the reference does
      sort_axis(lo, hi, split_axis);
      ref_order(lo, lo + split_k); ref_order(lo + split_k, hi);
    }
    for (int i = lo; i < hi; ++i) out.order.push_back(idx[i]);
    out.sah_cost += (double)(box.area() / root_area) * (hi - lo) * T_TRI;
    return ~(int32_t)((first << 3) | (hi - lo));
  }

  void store_child(LrBvhNode& n, int side, const Box& b) {
    float p = out.pad;
    for (int a = 0; a < 3; ++a) {
      float* row = a == 0 ? n.x : (a == 1 ? n.y : n.z);
      float ext = 1e-6f * std::fmax(std::fabs(b.mn[a]), std::fabs(b.mx[a]));   // covers the f32 rounding of the padded bound itself
      row[2 * side] = b.mn[a] - p - ext;
      row[2 * side + 1] = b.mx[a] + p + ext;
    }
  }

  // returns the child reference for idx[lo, hi)
  int32_t build(int lo, int hi, const Box& box, int depth) {
    int n = hi - lo;
    if (n == 1) { out.max_depth = std::max(out.max_depth, depth); return make_leaf(lo, hi, box); }
    int best_axis = -1, best_k = -1; float best_cost = 0.0f;
    if (depth < MEDIAN_DEPTH) {
      find_split(lo, hi, box, &best_axis, &best_k, &best_cost);
      if (n <= max_leaf && (float)n * T_TRI <= best_cost) { out.max_depth = std::max(out.max_depth, depth); return make_leaf(lo, hi, box, best_axis, best_k); }
    } else {
      if (n <= max_leaf) { out.max_depth = std::max(out.max_depth, depth); return make_leaf(lo, hi, box, 0, 0, true); }
      float ext[3] = {box.mx[0] - box.mn[0], box.mx[1] - box.mn[1], box.mx[2] - box.mn[2]};
      best_axis = ext[0] >= ext[1] && ext[0] >= ext[2] ? 0 : (ext[1] >= ext[2] ? 1 : 2);
      best_k = n / 2;
    }
    sort_axis(lo, hi, best_axis);
    int mid = lo + best_k;
    Box lb = range_box(lo, mid), rb = range_box(mid, hi);
    const Box lbs = range_store(lo, mid), rbs = range_store(mid, hi);
    int32_t me = (int32_t)out.nodes.size();
    out.nodes.push_back(LrBvhNode());
    std::memset(&out.nodes[me], 0, sizeof(LrBvhNode));
    out.sah_cost += (double)(box.area() / root_area) * 2.0 * T_AABB;
    int32_t l = build(lo, mid, lb, depth + 1);
    int32_t r = build(mid, hi, rb, depth + 1);
    LrBvhNode& node = out.nodes[me];
    store_child(node, 0, lbs); store_child(node, 1, rbs);
    node.child[0] = l; node.child[1] = r;
    return me;
  }
};

}  // namespace

BvhResult build_bvh(const LrPrimitive* prims, int n, int max_leaf, const float* extra_point) {
  auto t0 = std::chrono::steady_clock::now();
  BvhResult out;
  if (n < 0 || (n > 0 && !prims)) fail(LR_EINVAL, "bvh: bad primitive array");
  if (n >= (1 << 28)) fail(LR_EUNSUPPORTED, "bvh: too many primitives");
  max_leaf = std::max(1, std::min(max_leaf, 7));
  std::vector<Box> boxes((size_t)n);
  Box all; all.reset();
  for (int i = 0; i < n; ++i) {
    if (prims[i].type != LR_PRIM_TRIANGLE && prims[i].type != LR_PRIM_SPHERE) fail(LR_EINVAL, "bvh: unknown primitive type");
    boxes[i] = prim_box(prims[i]);
    for (int a = 0; a < 3; ++a) if (!(std::fabs(boxes[i].mn[a]) < INFINITY) || !(std::fabs(boxes[i].mx[a]) < INFINITY)) fail(LR_EINVAL, "bvh: non-finite primitive");
    all.grow(boxes[i]);
  }
  Box ext = all;
  if (extra_point) { Box e; for (int a = 0; a < 3; ++a) e.mn[a] = e.mx[a] = extra_point[a]; ext.grow(e); }
  float diag = 0.0f;
  if (n > 0 || extra_point) {
    float dx = ext.mx[0] - ext.mn[0], dy = ext.mx[1] - ext.mn[1], dz = ext.mx[2] - ext.mn[2];
    diag = std::sqrt(dx * dx + dy * dy + dz * dz);
    float far = 0.0f;
    for (int a = 0; a < 3; ++a) far = std::fmax(far, std::fmax(std::fabs(ext.mn[a]), std::fabs(ext.mx[a])));
    diag = std::fmax(diag, far);
  }
  // f32 slab arithmetic is good to a few ulp of |bound - origin| <= scene extent; 4e-6 * extent is
  // ~20x that (see DESIGN.md "conservative boxes")
  out.pad = std::fmax(4e-6f * diag, 1e-30f);
  // Spheres need more: sphere.rs:45 forms cod^2 - |co|^2 + r^2, whose rounding error is a few ulp of
  // |co|^2 <= extent^2, so the test can accept rays that pass up to sqrt(r^2 + ~2e-7 extent^2) from
  // the centre (a radius-1 sphere seen from 1e5 away "grows" by tens of units).  The box must cover
  // every ray the primitive test can accept, so sphere boxes grow by that amount (2x margin).
  std::vector<Box> store(boxes);                      // what the nodes store; `boxes` stays exact: the SAH and the order are the reference's
  Box all_exact = all;
  for (int i = 0; i < n; ++i) {
    if (prims[i].type != LR_PRIM_SPHERE) continue;
    float r = std::fabs(prims[i].v[3]);
    float grow = std::sqrt(r * r + 4e-7f * diag * diag) - r;
    for (int a = 0; a < 3; ++a) { store[i].mn[a] -= grow; store[i].mx[a] += grow; }
    all.grow(store[i]);
  }

  LrBvhNode root; std::memset(&root, 0, sizeof(root));
  if (n == 0) {
    for (int s = 0; s < 2; ++s) { root.x[2 * s] = root.y[2 * s] = root.z[2 * s] = 0.0f; root.x[2 * s + 1] = root.y[2 * s + 1] = root.z[2 * s + 1] = 0.0f; }
    root.child[0] = ~0; root.child[1] = ~0;          // two empty leaves
    out.nodes.push_back(root);
    out.max_depth = 1;
  } else {
    Builder b(boxes, store, out, max_leaf);
    for (int a = 0; a < 3; ++a) {
      b.centre[a].resize((size_t)n);
      for (int i = 0; i < n; ++i)                       // triangle.rs:116 centre = (max + min) / 2; sphere.rs:36 centre = position
        b.centre[a][i] = prims[i].type == LR_PRIM_SPHERE ? prims[i].v[a] : (boxes[i].mx[a] + boxes[i].mn[a]) / 2.0f;
    }
    b.idx.resize((size_t)n); for (int i = 0; i < n; ++i) b.idx[i] = i;
    b.pre.resize((size_t)n); b.suf.resize((size_t)n + 1);
    b.root_area = std::fmax(all_exact.area(), 1e-30f);
    out.nodes.reserve((size_t)n);
    out.order.reserve((size_t)n);
    int32_t r = b.build(0, n, all_exact, 0);
    if (r < 0) {
      // a single leaf: wrap it so that node 0 is always an inner node
      LrBvhNode w; std::memset(&w, 0, sizeof(w));
      b.store_child(w, 0, all); b.store_child(w, 1, all);
      w.child[0] = r; w.child[1] = ~0;
      out.nodes.insert(out.nodes.begin(), w);
    }
    out.max_depth += 1;
  }
  out.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  return out;
}

}  // namespace lrhost
