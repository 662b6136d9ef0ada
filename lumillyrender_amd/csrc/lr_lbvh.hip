// lr_lbvh.hip -- BVH build on the device: replaces the host SAH build
// (bvh.rs:56-127, ~0.4 s for 1e5 primitives on one host core) when lr_scene_create gets a description without a tree.
//
//   1. primitive boxes + scene bounds            (triangle.rs:102-118, sphere.rs:31-38)
//   2. conservative padding, 30-bit Morton codes of the box centres
//   3. radix sort of (code, primitive)            four stable 8-bit passes: per-tile histograms, one scan, ranked scatter (below)
//   4. the tree over the sorted primitives:
//        PLOC (default, round 3)   bottom-up merging of mutually nearest clusters within +-16 positions, SAH cost carried
//                                  along, subtrees of <= 7 primitives collapsed into leaves when that is cheaper; the tree
//                                  renders within 2 % of the host SAH tree (100k-triangle scene: 3.0 ms on the device)
//        LBVH (LR_DEVICE_BVH=lbvh) Karras 2012 binary radix tree over the codes + bottom-up box fit; 1.9 ms, but the fused
//                                  traversal kernel renders 38 % slower through it
//   5. emit the two-boxes-per-node layout lr_scene_create collapses into 4-wide nodes + primitives in leaf order
//
// The tree only prunes (DESIGN.md "closest-hit semantics"): images are bit-identical whichever builder made it.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <utility>
#include <string>
#include <vector>

#include "../../include/lumilly_hip.h"
#include "lr_lbvh.h"
#include "lr_knobs.h"

namespace lr {
namespace {

constexpr int kB = 256;

__device__ __forceinline__ uint32_t fkey(float f) {            // order-preserving float -> uint
  uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
inline float fkey_inv(uint32_t k) {
  uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
  float f; std::memcpy(&f, &u, 4); return f;
}

// boxes[6*i..] = min xyz, max xyz of primitive i; bounds[0..2] = key(min), bounds[3..5] = key(max)
__global__ void k_prim_boxes(const LrPrimitive* prims, int n, float* boxes, uint32_t* bounds) {
  __shared__ uint32_t s_b[6];
  if (threadIdx.x < 3) s_b[threadIdx.x] = 0xffffffffu; else if (threadIdx.x < 6) s_b[threadIdx.x] = 0u;
  __syncthreads();
  int i = blockIdx.x * kB + threadIdx.x;
  if (i < n) {
    const LrPrimitive p = prims[i];
    float mn[3], mx[3];
    if (p.type == LR_PRIM_TRIANGLE) {
      for (int a = 0; a < 3; ++a) {
        mn[a] = fminf(fminf(p.v[a], p.v[3 + a]), p.v[6 + a]);
        mx[a] = fmaxf(fmaxf(p.v[a], p.v[3 + a]), p.v[6 + a]);
      }
    } else {
      for (int a = 0; a < 3; ++a) { mn[a] = p.v[a] - p.v[3]; mx[a] = p.v[a] + p.v[3]; }
    }
    for (int a = 0; a < 3; ++a) {
      boxes[6 * (size_t)i + a] = mn[a]; boxes[6 * (size_t)i + 3 + a] = mx[a];
      atomicMin(&s_b[a], fkey(mn[a])); atomicMax(&s_b[3 + a], fkey(mx[a]));
    }
  }
  __syncthreads();
  if (threadIdx.x < 3) atomicMin(&bounds[threadIdx.x], s_b[threadIdx.x]);
  else if (threadIdx.x < 6) atomicMax(&bounds[threadIdx.x], s_b[threadIdx.x]);
}

__device__ __forceinline__ uint32_t expand10(uint32_t v) {    // 10 bits -> every third bit
  v = (v * 0x00010001u) & 0xFF0000FFu;
  v = (v * 0x00000101u) & 0x0F00F00Fu;
  v = (v * 0x00000011u) & 0xC30C30C3u;
  v = (v * 0x00000005u) & 0x49249249u;
  return v;
}

// pad the boxes (same rule as the host builder: 4e-6 * extent everywhere, spheres grow by
// sqrt(r^2 + 4e-7 extent^2) - r) and compute the Morton key of the padded box centre
__global__ void k_pad_morton(const LrPrimitive* prims, int n, float* boxes, float pad, float diag,
                             float lo0, float lo1, float lo2, float inv0, float inv1, float inv2,
                             uint32_t* keys, uint32_t* vals) {
  int i = blockIdx.x * kB + threadIdx.x;
  if (i >= n) return;
  float grow = pad;
  if (prims[i].type == LR_PRIM_SPHERE) { float r = fabsf(prims[i].v[3]); grow += sqrtf(r * r + 4e-7f * diag * diag) - r; }
  float c[3];
  for (int a = 0; a < 3; ++a) {
    float mn = boxes[6 * (size_t)i + a], mx = boxes[6 * (size_t)i + 3 + a];
    float ext = 1e-6f * fmaxf(fabsf(mn), fabsf(mx));
    mn = mn - grow - ext; mx = mx + grow + ext;
    boxes[6 * (size_t)i + a] = mn; boxes[6 * (size_t)i + 3 + a] = mx;
    c[a] = 0.5f * (mn + mx);
  }
  float lo[3] = {lo0, lo1, lo2}, inv[3] = {inv0, inv1, inv2};
  uint32_t q[3];
  for (int a = 0; a < 3; ++a) {
    float t = (c[a] - lo[a]) * inv[a] * 1024.0f;
    q[a] = (uint32_t)fminf(fmaxf(t, 0.0f), 1023.0f);
  }
  keys[i] = (expand10(q[0]) << 2) | (expand10(q[1]) << 1) | expand10(q[2]);
  vals[i] = (uint32_t)i;
}

__device__ __forceinline__ int delta(const uint32_t* keys, int n, int i, int j) {
  if (j < 0 || j >= n) return -1;
  uint32_t a = keys[i], b = keys[j];
  if (a == b) return 32 + __clz((uint32_t)i ^ (uint32_t)j);
  return __clz(a ^ b);
}

// Karras, "Maximizing Parallelism in the Construction of BVHs, Octrees, and k-d Trees" (HPG 2012), section 3.
// child refs: >= 0 internal node, < 0 leaf ~c = sorted position.  parent[] for internal nodes and leaves.
__global__ void k_radix_tree(const uint32_t* keys, int n, int2* children, int* node_parent, int* leaf_parent) {
  int i = blockIdx.x * kB + threadIdx.x;
  if (i >= n - 1) return;
  int d = (delta(keys, n, i, i + 1) - delta(keys, n, i, i - 1)) >= 0 ? 1 : -1;
  int dmin = delta(keys, n, i, i - d);
  int lmax = 2;
  while (delta(keys, n, i, i + lmax * d) > dmin) lmax <<= 1;
  int l = 0;
  for (int t = lmax >> 1; t >= 1; t >>= 1)
    if (delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
  int j = i + l * d;
  int dnode = delta(keys, n, i, j);
  int s = 0, t = l;
  do {
    t = (t + 1) >> 1;
    if (delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
  } while (t > 1);
  int gamma = i + s * d + (d < 0 ? -1 : 0);
  int lo = i < j ? i : j, hi = i < j ? j : i;
  int left = (lo == gamma) ? ~gamma : gamma;
  int right = (hi == gamma + 1) ? ~(gamma + 1) : gamma + 1;
  children[i] = make_int2(left, right);
  if (left >= 0) node_parent[left] = i; else leaf_parent[gamma] = i;
  if (right >= 0) node_parent[right] = i; else leaf_parent[gamma + 1] = i;
  if (i == 0) node_parent[0] = -1;
}

__device__ __forceinline__ float ld_agent(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int ld_agent(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// One thread per leaf walks up; the second child to arrive at a node merges the two boxes (read with
// agent-scope loads: the sibling's box was written by another CU and must not come from a stale L1 line).
__global__ void k_fit(const int2* children, const int* node_parent, const int* leaf_parent, const uint32_t* vals,
                      const float* prim_boxes, int n, float* node_boxes, int* node_height, int* flags) {
  int p = blockIdx.x * kB + threadIdx.x;
  if (p >= n) return;
  int node = leaf_parent[p];
  while (node >= 0) {
    __threadfence();                                              // my child's box / height are out before I announce
    if (atomicAdd(&flags[node], 1) == 0) return;                  // first arrival: the sibling will finish this node
    __threadfence();
    int2 ch = children[node];
    float mn[3] = {3.0e38f, 3.0e38f, 3.0e38f}, mx[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    int h = 0;
    for (int side = 0; side < 2; ++side) {
      int c = side == 0 ? ch.x : ch.y;
      const float* b = c >= 0 ? node_boxes + 6 * (size_t)c : prim_boxes + 6 * (size_t)vals[~c];
      for (int a = 0; a < 3; ++a) { mn[a] = fminf(mn[a], ld_agent(b + a)); mx[a] = fmaxf(mx[a], ld_agent(b + 3 + a)); }
      int hc = c >= 0 ? ld_agent(node_height + c) : 0;
      h = hc > h ? hc : h;
    }
    for (int a = 0; a < 3; ++a) { node_boxes[6 * (size_t)node + a] = mn[a]; node_boxes[6 * (size_t)node + 3 + a] = mx[a]; }
    node_height[node] = h + 1;
    node = node_parent[node];
  }
}

// traversal layout: 4 float4 per node = x-row, y-row, z-row {l.min, l.max, r.min, r.max}, {child0, child1}
__global__ void k_emit_nodes(const int2* children, const uint32_t* vals, const float* prim_boxes, const float* node_boxes, int n, float4* out) {
  int i = blockIdx.x * kB + threadIdx.x;
  if (i >= n - 1) return;
  int2 ch = children[i];
  const float* bl = ch.x >= 0 ? node_boxes + 6 * (size_t)ch.x : prim_boxes + 6 * (size_t)vals[~ch.x];
  const float* br = ch.y >= 0 ? node_boxes + 6 * (size_t)ch.y : prim_boxes + 6 * (size_t)vals[~ch.y];
  for (int a = 0; a < 3; ++a) out[4 * (size_t)i + a] = make_float4(bl[a], bl[3 + a], br[a], br[3 + a]);
  int cl = ch.x >= 0 ? ch.x : ~(int)((((uint32_t)~ch.x) << 3) | 1u);
  int cr = ch.y >= 0 ? ch.y : ~(int)((((uint32_t)~ch.y) << 3) | 1u);
  out[4 * (size_t)i + 3] = make_float4(__int_as_float(cl), __int_as_float(cr), 0.0f, 0.0f);
}

// primitive rows in leaf (= sorted) order, same packing as the host path (triangle.rs:71-72: e1, e2)
__global__ void k_emit_prims(const LrPrimitive* prims, const uint32_t* vals, int n, float4* out) {
  int k = blockIdx.x * kB + threadIdx.x;
  if (k >= n) return;
  uint32_t id = vals[k];
  const LrPrimitive p = prims[id];
  if (p.type == LR_PRIM_TRIANGLE) {
    out[3 * (size_t)k] = make_float4(p.v[0], p.v[1], p.v[2], __uint_as_float(id));
    out[3 * (size_t)k + 1] = make_float4(p.v[3] - p.v[0], p.v[4] - p.v[1], p.v[5] - p.v[2], 0.0f);
    out[3 * (size_t)k + 2] = make_float4(p.v[6] - p.v[0], p.v[7] - p.v[1], p.v[8] - p.v[2], 0.0f);
  } else {
    out[3 * (size_t)k] = make_float4(p.v[0], p.v[1], p.v[2], __uint_as_float(id | 0x80000000u));
    out[3 * (size_t)k + 1] = make_float4(p.v[3], p.v[3] * p.v[3], 0.0f, 0.0f);
    out[3 * (size_t)k + 2] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  }
}

// ---- LSD radix sort of (key, value) pairs: 8-bit digits, stable ------------------------------------------------
// A pass = k_rs_hist (one 256-bin histogram per tile of kRsTile keys, stored bin-major so that one exclusive scan over
// the whole table yields every tile's first output position per digit) + k_rs_scan + k_rs_scatter (the tile again, in
// order: a key's rank = keys of its digit in earlier tiles + in earlier rounds / waves of this tile + in lower lanes of its
// wave, the last one from eight __ballot masks).  Not a hot operation: it runs once per lr_scene_create.
constexpr int kRsTile = kB * 16;
__global__ void __launch_bounds__(kB) k_rs_hist(const uint32_t* keys, int n, int shift, int n_tiles, uint32_t* hist) {
  __shared__ uint32_t h[256];
  h[threadIdx.x] = 0;
  __syncthreads();
  const int base = blockIdx.x * kRsTile;
  for (int r = 0; r < kRsTile / kB; ++r) {
    int i = base + r * kB + (int)threadIdx.x;
    if (i < n) atomicAdd(&h[(keys[i] >> shift) & 0xffu], 1u);
  }
  __syncthreads();
  hist[(size_t)threadIdx.x * n_tiles + blockIdx.x] = h[threadIdx.x];
}
__global__ void __launch_bounds__(kB) k_rs_scan(uint32_t* hist, int total) {      // one workgroup: exclusive scan in place
  __shared__ uint32_t s_w[kB / 64];
  __shared__ uint32_t s_carry;
  if (threadIdx.x == 0) s_carry = 0;
  __syncthreads();
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  for (int base = 0; base < total; base += kB) {
    int i = base + (int)threadIdx.x;
    uint32_t v = i < total ? hist[i] : 0u, x = v;
    for (int off = 1; off < 64; off <<= 1) { uint32_t t = __shfl_up(x, off, 64); if ((int)lane >= off) x += t; }
    if (lane == 63u) s_w[wave] = x;
    __syncthreads();
    uint32_t pre = s_carry;
    for (uint32_t w = 0; w < wave; ++w) pre += s_w[w];
    if (i < total) hist[i] = pre + x - v;
    __syncthreads();
    if (threadIdx.x == kB - 1) s_carry = pre + x;
    __syncthreads();
  }
}
__global__ void __launch_bounds__(kB) k_rs_scatter(const uint32_t* keys, const uint32_t* vals, int n, int shift, int n_tiles, const uint32_t* hist,
                                                   uint32_t* keys_out, uint32_t* vals_out) {
  __shared__ uint32_t s_next[256];                                  // next output position of each digit for this tile
  s_next[threadIdx.x] = hist[(size_t)threadIdx.x * n_tiles + blockIdx.x];
  __syncthreads();
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const int base = blockIdx.x * kRsTile;
  for (int r = 0; r < kRsTile / kB; ++r) {
    int i = base + r * kB + (int)threadIdx.x;
    const bool valid = i < n;
    uint32_t k = valid ? keys[i] : 0u, v = valid ? vals[i] : 0u;
    uint32_t d = (k >> shift) & 0xffu;
    // lanes of this wave with the same digit
    uint64_t same = __ballot(valid);
    for (int b = 0; b < 8; ++b) { uint64_t m = __ballot((d >> b) & 1u); same &= ((d >> b) & 1u) ? m : ~m; }
    const uint32_t below = (uint32_t)__builtin_popcountll(same & ((1ull << lane) - 1ull));
    const uint32_t count = (uint32_t)__builtin_popcountll(same);
    const int leader = same ? (int)__builtin_ctzll(same) : 0;      // lowest lane of this lane's digit group
    uint32_t pos = 0;
    for (uint32_t w = 0; w < kB / 64; ++w) {                        // waves take their positions in order: stable
      if (wave == w) {
        // ONE lane per digit group reads the group's base and advances it; the others get the base by shuffle -- no lane
        // reads s_next[d] beside another lane's update of it, whatever the compiler schedules
        uint32_t group_base = 0;
        if (valid && below == 0) { group_base = s_next[d]; s_next[d] = group_base + count; }
        group_base = (uint32_t)__shfl((int)group_base, leader, 64);
        pos = group_base + below;
      }
      __syncthreads();
    }
    if (valid) { keys_out[pos] = k; vals_out[pos] = v; }
  }
}


// ---- PLOC: parallel locally-ordered clustering (Meister & Bittner, "Parallel Locally-Ordered Clustering for Bounding Volume
// Hierarchy Construction", TVCG 2018) over the Morton-sorted primitives.  The LBVH above splits where the Morton codes say; PLOC
// builds the tree bottom-up by MERGING: every cluster looks at its kR neighbours on either side in the current (Morton) order,
// picks the one whose union box has the smallest surface area, and mutually nearest pairs become a node.  Trees come out close to
// a top-down SAH build (bvh.rs:69-127 is a full-sweep SAH) at the cost of a few dozen cheap iterations.  On top of the paper:
// the SAH cost of every subtree is carried along, and a subtree of <= 7 primitives that is cheaper as ONE leaf (bvh.rs:71-72's
// cost model, T_aabb = 1, T_tri = 2) is collapsed into a leaf, as the host builder does.
// Cluster arrays are ping-ponged between iterations; a cluster reference is >= 0 for a node (creation order), < 0 for a
// primitive ~(sorted position).
#ifndef LR_PLOC_R
#define LR_PLOC_R 16
#endif
constexpr int kPlocR = LR_PLOC_R;
constexpr float kTAabb = 1.0f, kTTri = 2.0f;
constexpr int kPlocMaxHeight = 88;

struct PlocNodes {                       // one entry per created node (n - 1 in all)
  int2* child;                           // cluster references of the two children
  float* lr_box;                         // 12 floats: left box, right box (as the clusters were when merged)
  int* count;                            // primitives below
  float* cost;                           // SAH cost of the subtree, area-weighted (not divided by the root's area)
  int* height;
  int* collapsed;                        // the subtree is emitted as one leaf
  int* parent;                           // (parent node << 1) | side
  int* leaf_parent;                      // per sorted primitive position: (parent node << 1) | side
};

__device__ __forceinline__ float box_area6(const float* b) {
  float sx = b[3] - b[0], sy = b[4] - b[1], sz = b[5] - b[2];
  return 2.0f * (sx * sy + sy * sz + sz * sx);
}

__global__ void __launch_bounds__(kB) k_ploc_init(const uint32_t* vals, const float* prim_boxes, int n, float* cbox, int* cref) {
  int i = blockIdx.x * kB + threadIdx.x;
  if (i >= n) return;
  const float* b = prim_boxes + 6 * (size_t)vals[i];
  for (int a = 0; a < 6; ++a) cbox[6 * (size_t)i + a] = b[a];
  cref[i] = ~i;
}

// nearest neighbour within +-kPlocR positions by the surface area of the union box; ties go to the lower position
__global__ void __launch_bounds__(kB) k_ploc_nn(const float* cbox, const int* m_ptr, int* nn) {
  __shared__ float s_box[(kB + 2 * kPlocR) * 6];
  const int m = *m_ptr;
  const int base = blockIdx.x * kB;
  if (base >= m) return;
  for (int t = threadIdx.x; t < kB + 2 * kPlocR; t += kB) {
    int j = base - kPlocR + t;
    if (j >= 0 && j < m) for (int a = 0; a < 6; ++a) s_box[6 * t + a] = cbox[6 * (size_t)j + a];
  }
  __syncthreads();
  const int i = base + threadIdx.x;
  if (i >= m) return;
  const float* bi = s_box + 6 * (threadIdx.x + kPlocR);
  float best = 3.0e38f; int bj = -1;
  for (int dlt = -kPlocR; dlt <= kPlocR; ++dlt) {
    int j = i + dlt;
    if (dlt == 0 || j < 0 || j >= m) continue;
    const float* bjx = s_box + 6 * (threadIdx.x + kPlocR + dlt);
    float u[6];
    for (int a = 0; a < 3; ++a) { u[a] = fminf(bi[a], bjx[a]); u[3 + a] = fmaxf(bi[3 + a], bjx[3 + a]); }
    float d = box_area6(u);
    if (d < best) { best = d; bj = j; }
  }
  nn[i] = bj;
}

// mutually nearest pairs become a node (written by the lower position, which keeps the merged cluster); keep[i] says whether
// position i survives into the next iteration; bsum[block] = survivors of the block
__global__ void __launch_bounds__(kB) k_ploc_merge(float* cbox, int* cref, const int* nn, const int* m_ptr, PlocNodes nd, int* node_counter,
                                                  int* keep, int* bsum) {
  __shared__ int s_cnt;
  if (threadIdx.x == 0) s_cnt = 0;
  __syncthreads();
  const int m = *m_ptr;
  const int i = blockIdx.x * kB + threadIdx.x;
  int k = 0;
  if (i < m) {
    k = 1;
    const int j = nn[i];
    if (j >= 0 && nn[j] == i) {
      if (i < j) {
        float bl[6], br[6], u[6];
        for (int a = 0; a < 6; ++a) { bl[a] = cbox[6 * (size_t)i + a]; br[a] = cbox[6 * (size_t)j + a]; }
        for (int a = 0; a < 3; ++a) { u[a] = fminf(bl[a], br[a]); u[3 + a] = fmaxf(bl[3 + a], br[3 + a]); }
        const int rl = cref[i], rr = cref[j];
        const int id = atomicAdd(node_counter, 1);
        const int cl = rl >= 0 ? nd.count[rl] : 1, cr = rr >= 0 ? nd.count[rr] : 1;
        const float al = box_area6(bl), ar = box_area6(br), au = box_area6(u);
        const float costl = rl >= 0 ? nd.cost[rl] : kTTri * al, costr = rr >= 0 ? nd.cost[rr] : kTTri * ar;
        const float split = 2.0f * kTAabb * au + costl + costr, leaf = kTTri * (float)(cl + cr) * au;
        const bool collapse = cl + cr <= 7 && leaf <= split;
        const int hl = rl >= 0 ? nd.height[rl] : 0, hr = rr >= 0 ? nd.height[rr] : 0;
        nd.child[id] = make_int2(rl, rr);
        for (int a = 0; a < 6; ++a) { nd.lr_box[12 * (size_t)id + a] = bl[a]; nd.lr_box[12 * (size_t)id + 6 + a] = br[a]; }
        nd.count[id] = cl + cr;
        nd.cost[id] = collapse ? leaf : split;
        nd.collapsed[id] = collapse ? 1 : 0;
        nd.height[id] = collapse ? 0 : (hl > hr ? hl : hr) + 1;
        nd.parent[id] = -1;
        if (rl >= 0) nd.parent[rl] = (id << 1); else nd.leaf_parent[~rl] = (id << 1);
        if (rr >= 0) nd.parent[rr] = (id << 1) | 1; else nd.leaf_parent[~rr] = (id << 1) | 1;
        for (int a = 0; a < 6; ++a) cbox[6 * (size_t)i + a] = u[a];
        cref[i] = id;
      } else k = 0;                                                  // the partner at the lower position carries the pair
    }
    keep[i] = k;
  }
  if (k) atomicAdd(&s_cnt, 1);
  __syncthreads();
  if (threadIdx.x == 0) bsum[blockIdx.x] = s_cnt;
}

// one workgroup: exclusive scan of the per-block survivor counts (blocks beyond the live range count 0); writes the new m
__global__ void __launch_bounds__(kB) k_ploc_scan(int* bsum, int n_blocks, const int* m_ptr, int* m_next) {
  __shared__ int s_w[kB / 64];
  __shared__ int s_carry;
  if (threadIdx.x == 0) s_carry = 0;
  __syncthreads();
  const int live_blocks = (*m_ptr + kB - 1) / kB;
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  for (int base = 0; base < n_blocks; base += kB) {
    int i = base + (int)threadIdx.x;
    int v = (i < n_blocks && i < live_blocks) ? bsum[i] : 0, x = v;
    for (int off = 1; off < 64; off <<= 1) { int t = __shfl_up(x, off, 64); if ((int)lane >= off) x += t; }
    if (lane == 63u) s_w[wave] = x;
    __syncthreads();
    int pre = s_carry;
    for (uint32_t w = 0; w < wave; ++w) pre += s_w[w];
    if (i < n_blocks) bsum[i] = pre + x - v;
    __syncthreads();
    if (threadIdx.x == kB - 1) s_carry = pre + x;
    __syncthreads();
  }
  if (threadIdx.x == 0) *m_next = s_carry;
}

__global__ void __launch_bounds__(kB) k_ploc_compact(const float* cbox, const int* cref, const int* keep, const int* bsum, const int* m_ptr,
                                                    float* cbox_out, int* cref_out) {
  __shared__ int s_w[kB / 64];
  const int m = *m_ptr;
  const int i = blockIdx.x * kB + threadIdx.x;
  if (blockIdx.x * kB >= m) return;
  const int k = i < m ? keep[i] : 0;
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const uint64_t mask = __ballot(k);
  const int below = (int)__builtin_popcountll(mask & ((1ull << lane) - 1ull));
  if (lane == 0) s_w[wave] = (int)__builtin_popcountll(mask);
  __syncthreads();
  int pre = bsum[blockIdx.x];
  for (uint32_t w = 0; w < wave; ++w) pre += s_w[w];
  if (k) {
    const int o = pre + below;
    for (int a = 0; a < 6; ++a) cbox_out[6 * (size_t)o + a] = cbox[6 * (size_t)i + a];
    cref_out[o] = cref[i];
  }
}

// position of every primitive in depth-first (left before right) leaf order: the primitives below any node are contiguous,
// so a collapsed subtree is a leaf range
__global__ void __launch_bounds__(kB) k_ploc_leaf_pos(PlocNodes nd, int n, int* leaf_pos) {
  int p = blockIdx.x * kB + threadIdx.x;
  if (p >= n) return;
  int pos = 0;
  int up = nd.leaf_parent[p];
  while (up >= 0) {
    const int node = up >> 1, side = up & 1;
    if (side) { const int l = nd.child[node].x; pos += l >= 0 ? nd.count[l] : 1; }
    up = nd.parent[node];
  }
  leaf_pos[p] = pos;
}

__device__ __forceinline__ int ploc_first(const PlocNodes& nd, const int* leaf_pos, int ref) {   // leaf position of the leftmost primitive below ref
  while (ref >= 0) ref = nd.child[ref].x;
  return leaf_pos[~ref];
}

// traversal layout, root at index 0 (the root is the node created last: index = n_nodes - 1 - creation id)
__global__ void __launch_bounds__(kB) k_ploc_emit_nodes(PlocNodes nd, const int* leaf_pos, int n_nodes, float4* out) {
  int id = blockIdx.x * kB + threadIdx.x;
  if (id >= n_nodes) return;
  const int o = n_nodes - 1 - id;
  const int2 ch = nd.child[id];
  const float* b = nd.lr_box + 12 * (size_t)id;
  for (int a = 0; a < 3; ++a) out[4 * (size_t)o + a] = make_float4(b[a], b[3 + a], b[6 + a], b[9 + a]);
  int refs[2];
  for (int side = 0; side < 2; ++side) {
    const int c = side == 0 ? ch.x : ch.y;
    if (c < 0) refs[side] = ~(int)((((uint32_t)leaf_pos[~c]) << 3) | 1u);
    else if (nd.collapsed[c]) refs[side] = ~(int)((((uint32_t)ploc_first(nd, leaf_pos, c)) << 3) | (uint32_t)nd.count[c]);
    else refs[side] = n_nodes - 1 - c;
  }
  out[4 * (size_t)o + 3] = make_float4(__int_as_float(refs[0]), __int_as_float(refs[1]), 0.0f, 0.0f);
}

__global__ void __launch_bounds__(kB) k_ploc_emit_prims(const LrPrimitive* prims, const uint32_t* vals, const int* leaf_pos, int n, float4* out) {
  int p = blockIdx.x * kB + threadIdx.x;
  if (p >= n) return;
  const uint32_t id = vals[p];
  const size_t k = (size_t)leaf_pos[p];
  const LrPrimitive q = prims[id];
  if (q.type == LR_PRIM_TRIANGLE) {
    out[3 * k] = make_float4(q.v[0], q.v[1], q.v[2], __uint_as_float(id));
    out[3 * k + 1] = make_float4(q.v[3] - q.v[0], q.v[4] - q.v[1], q.v[5] - q.v[2], 0.0f);
    out[3 * k + 2] = make_float4(q.v[6] - q.v[0], q.v[7] - q.v[1], q.v[8] - q.v[2], 0.0f);
  } else {
    out[3 * k] = make_float4(q.v[0], q.v[1], q.v[2], __uint_as_float(id | 0x80000000u));
    out[3 * k + 1] = make_float4(q.v[3], q.v[3] * q.v[3], 0.0f, 0.0f);
    out[3 * k + 2] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  }
}

struct Tmp {
  std::vector<void*> ptrs;
  template <class T> T* get(size_t n) { void* p = nullptr; if (hipMalloc(&p, std::max<size_t>(n, 1) * sizeof(T)) != hipSuccess) return nullptr; ptrs.push_back(p); return (T*)p; }
  ~Tmp() { for (void* p : ptrs) (void)hipFree(p); }
};

#define LB_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { err = std::string(#x) + ": " + hipGetErrorString(e_); return LR_EDEVICE; } } while (0)

}  // namespace

int lbvh_build(const LrPrimitive* host_prims, int n, const float* extra_point, hipStream_t st,
               float4* d_nodes, float4* d_prims, int* height_out, double* ms_out, std::string& err) {
  if (n < 2) { err = "lbvh_build needs at least two primitives"; return LR_EINVAL; }
  Tmp tmp;
  struct Events {                                                  // destroyed on every return path
    hipEvent_t a = nullptr, b = nullptr;
    ~Events() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); }
  } ev;
  LB_OK(hipEventCreate(&ev.a)); LB_OK(hipEventCreate(&ev.b));
  hipEvent_t e0 = ev.a, e1 = ev.b;
  LrPrimitive* d_in = tmp.get<LrPrimitive>(n);
  float* boxes = tmp.get<float>((size_t)n * 6);
  float* node_boxes = tmp.get<float>((size_t)n * 6);
  uint32_t* bounds = tmp.get<uint32_t>(8);
  uint32_t *keys = tmp.get<uint32_t>(n), *keys2 = tmp.get<uint32_t>(n), *vals = tmp.get<uint32_t>(n), *vals2 = tmp.get<uint32_t>(n);
  int2* children = tmp.get<int2>(n);
  int *node_parent = tmp.get<int>(n), *leaf_parent = tmp.get<int>(n), *height = tmp.get<int>(n), *flags = tmp.get<int>(n);
  if (!d_in || !boxes || !node_boxes || !bounds || !keys || !keys2 || !vals || !vals2 || !children || !node_parent || !leaf_parent || !height || !flags) {
    err = "out of device memory"; return LR_ENOMEM;
  }
  LB_OK(hipMemcpyAsync(d_in, host_prims, (size_t)n * sizeof(LrPrimitive), hipMemcpyHostToDevice, st));
  LB_OK(hipEventRecord(e0, st));
  uint32_t init[6] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u};
  LB_OK(hipMemcpyAsync(bounds, init, sizeof(init), hipMemcpyHostToDevice, st));
  const int grid = (n + kB - 1) / kB;
  hipLaunchKernelGGL(k_prim_boxes, dim3(grid), dim3(kB), 0, st, d_in, n, boxes, bounds);
  uint32_t hb[6];
  LB_OK(hipMemcpyAsync(hb, bounds, sizeof(hb), hipMemcpyDeviceToHost, st));
  LB_OK(hipStreamSynchronize(st));
  float lo[3], hi[3];
  for (int a = 0; a < 3; ++a) { lo[a] = fkey_inv(hb[a]); hi[a] = fkey_inv(hb[3 + a]); if (!(std::fabs(lo[a]) < INFINITY) || !(std::fabs(hi[a]) < INFINITY)) { err = "non-finite primitive"; return LR_EINVAL; } }
  // scene extent as in the host builder (incl. the camera position)
  float elo[3] = {lo[0], lo[1], lo[2]}, ehi[3] = {hi[0], hi[1], hi[2]};
  if (extra_point) for (int a = 0; a < 3; ++a) { elo[a] = std::fmin(elo[a], extra_point[a]); ehi[a] = std::fmax(ehi[a], extra_point[a]); }
  float dx = ehi[0] - elo[0], dy = ehi[1] - elo[1], dz = ehi[2] - elo[2];
  float diag = std::sqrt(dx * dx + dy * dy + dz * dz), far = 0.0f;
  for (int a = 0; a < 3; ++a) far = std::fmax(far, std::fmax(std::fabs(elo[a]), std::fabs(ehi[a])));
  diag = std::fmax(diag, far);
  float pad = std::fmax(4e-6f * diag, 1e-30f);
  float inv[3];
  for (int a = 0; a < 3; ++a) inv[a] = hi[a] > lo[a] ? 1.0f / (hi[a] - lo[a]) : 0.0f;
  hipLaunchKernelGGL(k_pad_morton, dim3(grid), dim3(kB), 0, st, d_in, n, boxes, pad, diag, lo[0], lo[1], lo[2], inv[0], inv[1], inv[2], keys, vals);
  {
    // sort (code, primitive) by code: 30 bits = four 8-bit passes, ping-pong; the sorted arrays end in keys2 / vals2
    const int n_tiles = (n + kRsTile - 1) / kRsTile;
    uint32_t* hist = tmp.get<uint32_t>((size_t)256 * n_tiles);
    if (!hist) { err = "out of device memory"; return LR_ENOMEM; }
    uint32_t *ka = keys, *va = vals, *kb = keys2, *vb = vals2;
    for (int pass = 0; pass < 4; ++pass) {
      hipLaunchKernelGGL(k_rs_hist, dim3(n_tiles), dim3(kB), 0, st, ka, n, 8 * pass, n_tiles, hist);
      hipLaunchKernelGGL(k_rs_scan, dim3(1), dim3(kB), 0, st, hist, 256 * n_tiles);
      hipLaunchKernelGGL(k_rs_scatter, dim3(n_tiles), dim3(kB), 0, st, ka, va, n, 8 * pass, n_tiles, hist, kb, vb);
      std::swap(ka, kb); std::swap(va, vb);
    }
    // four passes: the result is back in (keys, vals); the tree kernels read keys2 / vals2
    LB_OK(hipMemcpyAsync(keys2, keys, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToDevice, st));
    LB_OK(hipMemcpyAsync(vals2, vals, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToDevice, st));
    LB_OK(hipGetLastError());
    if (std::getenv("LR_DEBUG")) {                                  // self-check: sorted codes, values a permutation of 0..n-1
      std::vector<uint32_t> hk((size_t)n), hv((size_t)n);
      LB_OK(hipMemcpyAsync(hk.data(), keys2, (size_t)n * 4, hipMemcpyDeviceToHost, st));
      LB_OK(hipMemcpyAsync(hv.data(), vals2, (size_t)n * 4, hipMemcpyDeviceToHost, st));
      LB_OK(hipStreamSynchronize(st));
      std::vector<char> seen((size_t)n, 0);
      for (int i = 0; i < n; ++i) {
        if ((i > 0 && hk[i] < hk[i - 1]) || hv[i] >= (uint32_t)n || seen[hv[i]]) { err = "radix sort self-check failed at " + std::to_string(i); return LR_EDEVICE; }
        seen[hv[i]] = 1;
      }
      std::fprintf(stderr, "[lr] lbvh: radix sort of %d codes verified\n", n);
    }
  }
  const char* which = lr_knob("LR_DEVICE_BVH");
  if (!(which && std::strcmp(which, "lbvh") == 0)) {
    // ---- PLOC over the sorted primitives (default) ----
    const size_t nn1 = (size_t)n;
    float* cb[2] = {tmp.get<float>(nn1 * 6), tmp.get<float>(nn1 * 6)};
    int* cr[2] = {tmp.get<int>(nn1), tmp.get<int>(nn1)};
    int *nnb = tmp.get<int>(nn1), *keep = tmp.get<int>(nn1), *bsum = tmp.get<int>((size_t)grid + 1), *ctr = tmp.get<int>(4), *leaf_pos = tmp.get<int>(nn1);
    PlocNodes nd;
    nd.child = children; nd.lr_box = tmp.get<float>(nn1 * 12); nd.count = tmp.get<int>(nn1); nd.cost = tmp.get<float>(nn1); nd.height = height;
    nd.collapsed = flags; nd.parent = node_parent; nd.leaf_parent = leaf_parent;
    if (!cb[0] || !cb[1] || !cr[0] || !cr[1] || !nnb || !keep || !bsum || !ctr || !leaf_pos || !nd.lr_box || !nd.count || !nd.cost) { err = "out of device memory"; return LR_ENOMEM; }
    int hctr[4] = {n, 0, 0, 0};                                   // [0] / [1]: cluster count of the current / next iteration (ping-pong), [2]: nodes created
    LB_OK(hipMemcpyAsync(ctr, hctr, sizeof(hctr), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_ploc_init, dim3(grid), dim3(kB), 0, st, vals2, boxes, n, cb[0], cr[0]);
    int m_host = n, cur = 0, iters = 0;
    bool stalled = false;                                          // no cluster found a partner in a whole batch (or the iteration budget ran out)
    while (m_host > 1) {
      // A batch without a single merge cannot happen while union areas are finite and below the 3.0e38 `best` sentinel; it does
      // for coordinates of finite but absurd extent (> ~1e19: the area overflows).  The radix tree below builds such a scene
      // -- its splits are by key, not by area -- so PLOC gives up instead of failing the scene.
      if (iters > 4096) { stalled = true; break; }
      const int m_before = m_host;
      const int g = (m_host + kB - 1) / kB;                       // m only shrinks: the last known count bounds the grid
      for (int k = 0; k < 4; ++k, ++iters) {
        int* m_cur = ctr + (iters & 1);
        int* m_nxt = ctr + ((iters + 1) & 1);
        hipLaunchKernelGGL(k_ploc_nn, dim3(g), dim3(kB), 0, st, cb[cur], m_cur, nnb);
        hipLaunchKernelGGL(k_ploc_merge, dim3(g), dim3(kB), 0, st, cb[cur], cr[cur], nnb, m_cur, nd, ctr + 2, keep, bsum);
        hipLaunchKernelGGL(k_ploc_scan, dim3(1), dim3(kB), 0, st, bsum, g, m_cur, m_nxt);
        hipLaunchKernelGGL(k_ploc_compact, dim3(g), dim3(kB), 0, st, cb[cur], cr[cur], keep, bsum, m_cur, cb[cur ^ 1], cr[cur ^ 1]);
        cur ^= 1;
      }
      LB_OK(hipGetLastError());
      LB_OK(hipMemcpyAsync(&m_host, ctr + (iters & 1), sizeof(int), hipMemcpyDeviceToHost, st));
      LB_OK(hipStreamSynchronize(st));
      if (m_host < 1) { err = "PLOC lost its clusters"; return LR_EDEVICE; }
      if (m_host == m_before) { stalled = true; break; }
    }
    if (stalled) {
      if (std::getenv("LR_DEBUG")) std::fprintf(stderr, "[lr] ploc: no merge after %d iterations with %d clusters left, falling back to the radix tree\n", iters, m_host);
    } else {
    const int n_nodes = n - 1;
    hipLaunchKernelGGL(k_ploc_leaf_pos, dim3(grid), dim3(kB), 0, st, nd, n, leaf_pos);
    hipLaunchKernelGGL(k_ploc_emit_nodes, dim3(grid), dim3(kB), 0, st, nd, leaf_pos, n_nodes, d_nodes);
    hipLaunchKernelGGL(k_ploc_emit_prims, dim3(grid), dim3(kB), 0, st, d_in, vals2, leaf_pos, n, d_prims);
    LB_OK(hipGetLastError());
    LB_OK(hipEventRecord(e1, st));
    int created = 0, h = 0;
    LB_OK(hipMemcpyAsync(&created, ctr + 2, sizeof(int), hipMemcpyDeviceToHost, st));
    LB_OK(hipMemcpyAsync(&h, height + (n_nodes - 1), sizeof(int), hipMemcpyDeviceToHost, st));     // the root is the node created last
    LB_OK(hipStreamSynchronize(st));
    if (created != n_nodes) { err = "PLOC created " + std::to_string(created) + " nodes for " + std::to_string(n) + " primitives"; return LR_EDEVICE; }
    float ms = 0.0f; LB_OK(hipEventElapsedTime(&ms, e0, e1));
    if (std::getenv("LR_DEBUG")) std::fprintf(stderr, "[lr] ploc: %d primitives, %d iterations, height %d, %.3f ms on the device\n", n, iters, h, ms);
    // Needle-shaped primitives (a mesh stretched 40:1) make area-driven merging chain up: heights of ~100 were seen (fuzz seeds
    // 515, 529, 669, 711).  The traversal stack is sized for <= 95; such an input gets the radix tree below instead, whose
    // height is bounded by the key length.
    if (h <= kPlocMaxHeight) {
      *height_out = h < 1 ? 1 : h; *ms_out = ms;
      return LR_OK;
    }
    if (std::getenv("LR_DEBUG")) std::fprintf(stderr, "[lr] ploc: tree too deep (%d), falling back to the radix tree\n", h);
    }
  }
  LB_OK(hipMemsetAsync(flags, 0, (size_t)n * sizeof(int), st));
  LB_OK(hipMemsetAsync(height, 0, (size_t)n * sizeof(int), st));
  hipLaunchKernelGGL(k_radix_tree, dim3(grid), dim3(kB), 0, st, keys2, n, children, node_parent, leaf_parent);
  hipLaunchKernelGGL(k_fit, dim3(grid), dim3(kB), 0, st, children, node_parent, leaf_parent, vals2, boxes, n, node_boxes, height, flags);
  hipLaunchKernelGGL(k_emit_nodes, dim3(grid), dim3(kB), 0, st, children, vals2, boxes, node_boxes, n, d_nodes);
  hipLaunchKernelGGL(k_emit_prims, dim3(grid), dim3(kB), 0, st, d_in, vals2, n, d_prims);
  LB_OK(hipGetLastError());
  LB_OK(hipEventRecord(e1, st));
  int h = 0;
  LB_OK(hipMemcpyAsync(&h, height, sizeof(int), hipMemcpyDeviceToHost, st));
  LB_OK(hipStreamSynchronize(st));
  float ms = 0.0f; LB_OK(hipEventElapsedTime(&ms, e0, e1));
  *height_out = h; *ms_out = ms;
  return LR_OK;
}

}  // namespace lr
