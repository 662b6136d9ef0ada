// lr_lbvh.hip -- BVH build on the device: replaces the host SAH build
// (bvh.rs:56-127, ~0.5 s for 1e5 primitives) when lr_scene_create gets a description without a tree.
//
//   1. primitive boxes + scene bounds            (triangle.rs:102-118, sphere.rs:31-38)
//   2. conservative padding, 30-bit Morton codes of the box centres
//   3. radix sort of (code, primitive)            four stable 8-bit passes: per-tile histograms, one scan, ranked scatter (below)
//   4. Karras 2012 binary radix tree over the sorted codes (ties broken by position)
//   5. bottom-up box fit + height, agent-scope hand-off between the two children of a node
//   6. emit the two-boxes-per-node layout of the traversal kernels + primitives in leaf order
//
// The tree only prunes (DESIGN.md "closest-hit semantics"): images are bit-identical to the ones
// rendered with the host SAH tree; an LBVH is merely ~1.3-2x slower to traverse.
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
