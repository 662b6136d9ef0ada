// lumilly_hip.hip -- C ABI of liblumilly_hip.so (include/lumilly_hip.h): scene upload, the
// wavefront render loop and its measurement hooks.  gfx950 only.
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (see csrc/Makefile).  The -ffp-contract
// flag is part of the contract with the parity oracle: hit/miss decisions must round like the
// reference's f32 code.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../include/lumilly_hip.h"
#include "../../include/lumilly_hip_diag.h"
#include "lr_kernels.h"
#include "lr_path.h"
#include "lr_lbvh.h"
#include "lr_knobs.h"

// the flat-scene kernels are compiled in lr_flat.hip (their own scheduler switch, see there); this unit only launches them
namespace lr {
extern template __global__ void k_path_flat<1u, 1>(DevScene, DevState, DevParams, const float4*);
extern template __global__ void k_path_flat<9u, 1>(DevScene, DevState, DevParams, const float4*);
extern template __global__ void k_path_flat<31u, 1>(DevScene, DevState, DevParams, const float4*);
extern template __global__ void k_path_flat<1u, 2>(DevScene, DevState, DevParams, const float4*);
extern template __global__ void k_path_flat<9u, 2>(DevScene, DevState, DevParams, const float4*);
extern template __global__ void k_path_flat<31u, 2>(DevScene, DevState, DevParams, const float4*);
extern template __global__ void k_resident<true, 1u, 512>(DevScene, DevState, DevParams, uint32_t, const float4*);
extern template __global__ void k_resident<true, 1u, 256>(DevScene, DevState, DevParams, uint32_t, const float4*);
extern template __global__ void k_resident<true, 31u, 512>(DevScene, DevState, DevParams, uint32_t, const float4*);
extern template __global__ void k_resident<true, 31u, 256>(DevScene, DevState, DevParams, uint32_t, const float4*);
}  // namespace lr

using namespace lr;

namespace {

thread_local std::string g_err;

struct ApiError { int code; std::string msg; };
[[noreturn]] void fail(int code, const std::string& m) { throw ApiError{code, m}; }
void hip_check(hipError_t e, const char* what) {
  if (e != hipSuccess) fail(LR_EDEVICE, std::string(what) + ": " + hipGetErrorString(e));
}
#define HIP_OK(x) hip_check((x), #x)

// Owning device buffer: released by the destructor, so a temporary never leaks on an error path (every HIP_OK throws).
template <class T>
struct DevBuf {
  T* p = nullptr; size_t n = 0;
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  ~DevBuf() { release(); }
  void ensure(size_t count) {
    if (count <= n && p) return;
    release();
    HIP_OK(hipMalloc((void**)&p, std::max<size_t>(count, 1) * sizeof(T)));
    n = std::max<size_t>(count, 1);
  }
  void upload(const std::vector<T>& v, hipStream_t s) {
    ensure(v.size());
    if (!v.empty()) HIP_OK(hipMemcpyAsync(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, s));
  }
  void release() { if (p) { (void)hipFree(p); p = nullptr; n = 0; } }
};

constexpr double kSliverInvSin = 8.0;   // (statistics only since round 5: triangles with sin(angle at p0) < 1/8, lr_selftest_tree_info)
constexpr double kCullSlack = 8.0;      // distance culling keeps kCullSlack eps |e1||e2| (|o - p0| + |t|) / 1e-3 of slack per child (Wide4Builder).  First-order
                                        // analysis of triangle.rs:69-100 in f32: |dN| <= 7.5 eps A |tv|, |d det| <= 7.5 eps A, t = N / det => |dt| <= 7.5 eps A (|tv| + |t|) / |det|
                                        // + 2 eps |t|; measured against float64 on 1.6 M accepted grazing hits: at most 1.9 (tests/test_oracle_properties.py).  Cost
                                        // (interleaved, Msamples/s on config 4 | 5): 0: 4173 | 4074, 8: 4122 | 4026, 12: 4115 | 4018, 16: 4108 | 4019, 24: 4093 | 4014, 64: 4036 | 4003,
                                        // 200: 3912 | 3943 (round 4's sliver flag: 4101 | 4072); the 22 residual seeds are clean from 4 upwards
constexpr int kEventPool = 1024;      // timed launches per kernel type per render
constexpr int kProfileStride = 2;     // time every 2nd iteration when LR_FLAG_PROFILE is set

struct EventPool {
  std::vector<hipEvent_t> a, b; int used = 0;
  void init() {
    if (!a.empty()) return;
    a.resize(kEventPool); b.resize(kEventPool);
    for (int i = 0; i < kEventPool; ++i) { HIP_OK(hipEventCreate(&a[i])); HIP_OK(hipEventCreate(&b[i])); }
  }
  void destroy() { for (auto e : a) (void)hipEventDestroy(e); for (auto e : b) (void)hipEventDestroy(e); a.clear(); b.clear(); }
};

}  // namespace

struct LrScene {
  int device = 0;
  hipStream_t stream = nullptr;
  hipStream_t gstream[2] = {nullptr, nullptr};   // streams of the 2nd and 3rd slot group of the streaming pipeline (render_impl)
  hipEvent_t grp_ev[4] = {nullptr, nullptr, nullptr, nullptr};   // [0] setup done, [1], [2] group 1 / 2 batch done
  int n_cus = 0;
  // scene blob
  DevBuf<float4> nodes, prims, flat, shade, emit, texels;
  DevBuf<uint32_t> texels_rgbe;        // the IBL map as RGBE words when every texel re-encodes exactly (lr_device.h)
  DevBuf<uint8_t> prim_qid;
  DevScene dev;
  bool mat_present[kNumShadeQueues] = {false, false, false, false, false, true};
  int stack_depth = 2;
  int tree_info[4] = {0, 0, 0, 0};     // 4-wide nodes, nodes whose culling slack exceeds the distance itself (2 kappa >= 1: a wall-sized triangle below), sliver triangles, stack need
  double bvh_build_ms = 0.0;           // device LBVH build time (0 when the host supplied the tree)
  int film_w = 0, film_h = 0;
  int n_prims = 0;
  std::vector<int32_t> user_id;        // device primitive id -> the caller's index (pack_scene: device ids follow the tree's leaf order)
  // render state (kept between calls)
  DevBuf<float4> ray_o, ray_d, thr, rad, acc, sh_d, sh_w, partial;
  DevBuf<float2> hit;
  DevBuf<uint32_t> q_shade, c_shade, q_shadow, c_shadow, counters, tile_prefix, rank_pixel, stack_spill, chunk_start;
  DevBuf<uint16_t> sort_key, order;
  DevBuf<uint4> pool;
  DevBuf<int4> tiles;
  DevBuf<unsigned long long> stats_dev;
  DevBuf<float> film;
  DevBuf<float> packed;                // the rendered tiles' pixels in pixel-rank order (lr_render reads back only these)
#ifdef LR_TIMELINE
  DevBuf<unsigned long long> timeline;
#endif
  uint32_t* pinned = nullptr;         // [0..2], [4..6] retired-slot read-backs (two polls x up to three slot groups), [8..] stats
  hipEvent_t poll_ev[2] = {nullptr, nullptr};
  hipEvent_t t_begin = nullptr, t_end = nullptr;
  EventPool pools[LR_K_COUNT];
  LrStats stats;
  float* host_film = nullptr;           // pinned staging copy of the film (lr_render), host_film_cap floats
  size_t host_film_cap = 0;
};

namespace {

// Binary tree (4 float4 per node: x/y/z rows {l.min, l.max, r.min, r.max}, {child0, child1}) -> 4-wide tree of
// lr_device.h.  A node adopts its grandchildren: starting from its two children, the inner child with the largest
// surface area is replaced by its own two children until four slots are filled.  Leaves keep their encoding.
// Returns the worst-case traversal stack need (sum over a root-to-leaf path of children - 1).
//
// The child boxes are stored QUANTISED (lr_device.h: 64-B node): 8 bits per plane on a power-of-two grid anchored at
// the node's lower corner, lower planes rounded down and upper planes rounded up, so a stored box always CONTAINS the
// (already padded) box it came from.  Boxes only prune -- the closest hit is defined by the primitive tests alone
// (aabb.rs:74-92 decides nothing a primitive test would accept differently) -- so the quantisation changes how many
// nodes a ray visits, never a result.
struct Wide4Builder {
  const std::vector<float4>& in;
  std::vector<float4>& out;
  // leaf_area[k] = |e1||e2| of the triangle at leaf position k (0 for a sphere).  bvh.rs:131-141 tests every leaf whose box the
  // ray touches, however far behind the closest hit so far it begins; a traversal that culls by distance assumes an accepted hit
  // lies inside its primitive's box.  Moeller-Trumbore's distance t = (e2 . qv) / det carries an absolute error of up to
  // ~12 eps |e1||e2| (|o - p0| + |t|) / |det| and triangle.rs:75 accepts |det| down to an ABSOLUTE 1e-3, so a hit at grazing
  // incidence can be reported in front of its own (padded) box -- by 5 triangle sizes in round 4's fuzz.  Every node therefore
  // carries kappa = kCullSlack (= 8) eps max|e1||e2| / 1e-3 over the triangles below it, and a child is culled only when it begins
  // beyond   bound + kappa (2 t_far(child) + diagonal(node))   -- the error bound of any triangle inside that child, with
  // |o - p0| <= t_far + diagonal and |t| <= t_far.  For a well-conditioned mesh kappa is ~1e-2 (a few units of slack at the
  // scales of the BASELINE scenes: +2-3 % node visits); for a wall-sized triangle it is so large that nothing below the nodes that
  // hold it is culled by distance -- the reference's rule for that subtree.  Replaces round 3's sliver flag (sin(phi) < 1/8: a
  // heuristic on the shape; the bound above covers slivers through |det| >= 1e-3 like every other triangle).
  const std::vector<float>* leaf_area = nullptr;
  double cull_slack = kCullSlack;       // (pack_scene: the LR_CULL_SLACK knob of a diagnostic build, read once per scene)
  struct Cand { float lo[3], hi[3]; int ref; };
  static float area(const Cand& c) {
    float dx = c.hi[0] - c.lo[0], dy = c.hi[1] - c.lo[1], dz = c.hi[2] - c.lo[2];
    return dx * dy + dy * dz + dz * dx;
  }
  void children(int node, Cand* two) const {
    const float4 x = in[4 * (size_t)node], y = in[4 * (size_t)node + 1], z = in[4 * (size_t)node + 2], c = in[4 * (size_t)node + 3];
    two[0] = Cand{{x.x, y.x, z.x}, {x.y, y.y, z.y}, __builtin_bit_cast(int, c.x)};
    two[1] = Cand{{x.z, y.z, z.z}, {x.w, y.w, z.w}, __builtin_bit_cast(int, c.y)};
  }
  int build(int node, int* need_out, float* amax_out = nullptr) {
    const size_t me = out.size() / kNodeRows;
    float amax = 0.0f;                                                   // max |e1||e2| over the triangles below this node
    out.resize(out.size() + kNodeRows, make_float4(0, 0, 0, 0));
    Cand c[4]; int n = 2;
    children(node, c);
    while (n < 4) {
      int pick = -1; float best = -1.0f;
      for (int k = 0; k < n; ++k) if (c[k].ref >= 0 && area(c[k]) > best) { best = area(c[k]); pick = k; }
      if (pick < 0) break;
      Cand two[2]; children(c[pick].ref, two);
      c[pick] = two[0]; c[n++] = two[1];
    }
    int need = 0;
    int refs[4];
    for (int k = 0; k < 4; ++k) {
      int ref = kEmptyChild;
      if (k < n) {
        ref = c[k].ref;
        if (ref >= 0) { int sub = 0; float a = 0.0f; ref = build(ref, &sub, &a); amax = std::fmax(amax, a); need = std::max(need, n - 1 + sub); }
        else {
          need = std::max(need, n - 1);
          if (leaf_area && ref != kEmptyChild) {
            const uint32_t enc = (uint32_t)~ref, first = enc >> 3, count = enc & 7u;
            for (uint32_t q = first; q < first + count && q < leaf_area->size(); ++q) amax = std::fmax(amax, (*leaf_area)[q]);
          }
        }
      }
      refs[k] = ref;
    }
    // grid: origin = the lower corner of the union (an f32), step 2^e per axis with 255 steps covering the extent
    float org[3]; uint32_t ebits = 0; uint32_t qlo[3] = {0, 0, 0}, qhi[3] = {0, 0, 0};
    double diag2 = 0.0;
    for (int a = 0; a < 3; ++a) {
      float lo = c[0].lo[a], hi = c[0].hi[a];
      for (int k = 1; k < n; ++k) { lo = std::fmin(lo, c[k].lo[a]); hi = std::fmax(hi, c[k].hi[a]); }
      if (!(std::fabs(lo) < INFINITY) || !(std::fabs(hi) < INFINITY) || hi < lo) fail(LR_EINVAL, "BVH box is not finite");
      org[a] = lo;
      const double ext = (double)hi - (double)lo;
      diag2 += ext * ext;
      int e = -126;
      if (ext > 0.0) { int ex; (void)std::frexp(ext / 255.0, &ex); e = ex; }          // 2^ex > ext / 255 >= 2^(ex-1)
      if (e > 40) fail(LR_EUNSUPPORTED, "scene extent beyond 2^48: outside the range the traversal arithmetic keeps finite");
      e = std::max(-126, e);
      const double step = std::ldexp(1.0, e);
      if (255.0 * step < ext) fail(LR_EUNSUPPORTED, "BVH box extent outside the quantisation range");
      ebits |= (uint32_t)(e + 127) << (8 * a);
      for (int k = 0; k < 4; ++k) {
        uint32_t ql = 255, qh = 0;                                                     // empty slot: inverted (also skipped by its child reference)
        if (k < n) {
          double fl = std::floor(((double)c[k].lo[a] - (double)lo) / step), ce = std::ceil(((double)c[k].hi[a] - (double)lo) / step);
          fl = std::max(0.0, std::min(255.0, fl)); ce = std::max(0.0, std::min(255.0, ce));
          // decoded planes are exact in double (f32 origin + an integer multiple of a power of two): make sure they contain
          while (fl > 0.0 && (double)lo + fl * step > (double)c[k].lo[a]) fl -= 1.0;
          while (ce < 255.0 && (double)lo + ce * step < (double)c[k].hi[a]) ce += 1.0;
          if ((double)lo + fl * step > (double)c[k].lo[a] || (double)lo + ce * step < (double)c[k].hi[a]) fail(LR_EUNSUPPORTED, "BVH box cannot be quantised conservatively");
          ql = (uint32_t)fl; qh = (uint32_t)ce;
        }
        qlo[a] |= ql << (8 * k); qhi[a] |= qh << (8 * k);
      }
    }
    auto fb = [](uint32_t u) { return __builtin_bit_cast(float, u); };
    out[me * kNodeRows + 0] = make_float4(org[0], org[1], org[2], fb(ebits));
    out[me * kNodeRows + 1] = make_float4(fb(qlo[0]), fb(qlo[1]), fb(qlo[2]), fb(qhi[0]));
    // the culling slack of this node's children: lim = bound + kappa * diagonal + 2 kappa * t_far(child)  (finite: no inf - inf on the device)
    const double slack_k = cull_slack;
    const double kappa = std::isfinite(amax) ? slack_k * 5.9604644775390625e-8 * (double)amax / 1.0e-3 : 1.0e12;
    const float k2 = (float)std::fmin(2.0 * kappa, 1.0e12), kd = (float)std::fmin(kappa * std::sqrt(diag2), 1.0e30);
    out[me * kNodeRows + 2] = make_float4(fb(qhi[1]), fb(qhi[2]), k2, kd);
    out[me * kNodeRows + 3] = make_float4(fb((uint32_t)refs[0]), fb((uint32_t)refs[1]), fb((uint32_t)refs[2]), fb((uint32_t)refs[3]));
    *need_out = need;
    if (amax_out) *amax_out = amax;
    return (int)me;
  }
};

void pack_scene(LrScene& s, const LrSceneDesc& d) {
  if (d.abi_version != LR_ABI_VERSION) fail(LR_EINVAL, "LrSceneDesc.abi_version mismatch");
  if (d.n_prims < 0 || d.n_materials < 0 || (d.n_prims > 0 && (!d.prims || !d.materials))) fail(LR_EINVAL, "bad primitive / material arrays");
  const bool device_bvh = d.n_bvh_nodes == 0 && !d.bvh_nodes;            // no tree supplied: build an LBVH on the device
  if (!device_bvh && (d.n_bvh_nodes < 1 || !d.bvh_nodes || (d.n_prims > 0 && !d.bvh_prim_order))) fail(LR_EINVAL, "malformed BVH (build it with lr_host_build_bvh, or pass n_bvh_nodes = 0 for a device build)");
  if (d.camera.resolution[0] <= 0 || d.camera.resolution[1] <= 0) fail(LR_EINVAL, "bad film resolution");
  if ((uint64_t)d.camera.resolution[0] * (uint64_t)d.camera.resolution[1] > 0xffffffffull) fail(LR_EINVAL, "film too large");
  if (d.camera.type < 0 || d.camera.type > LR_CAMERA_OMNIDIRECTIONAL) fail(LR_EINVAL, "unknown camera type");
  const int np = d.n_prims;

  // DEVICE PRIMITIVE IDS = the tree's leaf order.  bvh.rs:131-141 keeps the FIRST minimum of the candidate list and bvh.rs:38-45 fills
  // that list left subtree first, so an exact distance tie goes to the primitive a depth-first walk of the reference's tree meets
  // first -- which is the order of the description's bvh_prim_order (lr_host_build_bvh reproduces the reference's SAH recursion and
  // its sorts, host/bvh_build.cpp).  Numbering the primitives by it makes the kernels' tie rule ("lowest id", explicit in the tree
  // walk, implicit in the flat loop's row order) the reference's.  Everything indexed by primitive id below (shading records, own
  // boxes, BSDF ids, the rows' id words) uses device ids; the emitter table keeps INSTANCE order (objects.rs:19-24); diagnostics
  // translate back (LrScene::user_id).  Without a caller's tree (device LBVH build) the ids are the caller's: ties by instance order.
  std::vector<LrPrimitive> prim_dev((size_t)np);
  std::vector<int32_t> rank((size_t)np), leaf_order((size_t)np);
  s.user_id.assign((size_t)np, 0);
  {
    std::vector<char> seen((size_t)np, 0);
    for (int k = 0; k < np; ++k) {
      int id = device_bvh ? k : d.bvh_prim_order[k];
      if (id < 0 || id >= np || seen[(size_t)id]) fail(LR_EINVAL, "BVH primitive order is not a permutation");
      seen[(size_t)id] = 1;
      rank[(size_t)id] = k; s.user_id[(size_t)k] = id; leaf_order[(size_t)k] = k;
      prim_dev[(size_t)k] = d.prims[id];
    }
  }
  const LrPrimitive* const dprims = prim_dev.data();                    // primitives by DEVICE id
  const int32_t* const dorder = leaf_order.data();                      // leaf position -> device id (the identity)

  // materials (weight: lambert.rs:27-30 and the other four `weight` impls)
  std::vector<float4> mats((size_t)d.n_materials * 3);
  for (int i = 0; i < d.n_materials; ++i) {
    const LrMaterial& m = d.materials[i];
    if (m.type < 0 || m.type > LR_MAT_IDEAL_REFRACTION) fail(LR_EINVAL, "unknown material type");
    float w = std::fmax(std::fmax(m.color[0], m.color[1]), m.color[2]);
    bool emits = m.type == LR_MAT_LAMBERT;                        // only Lambert has emission (lambert.rs:23-25)
    mats[3 * i] = make_float4(m.color[0], m.color[1], m.color[2], __builtin_bit_cast(float, (uint32_t)m.type));
    mats[3 * i + 1] = make_float4(emits ? m.emission[0] : 0.0f, emits ? m.emission[1] : 0.0f, emits ? m.emission[2] : 0.0f, w);
    mats[3 * i + 2] = make_float4(m.param[0], m.param[1], m.param[2], 0.0f);
    s.mat_present[m.type] = s.mat_present[m.type];               // presence is decided per primitive below
  }

  // per-primitive shading rows + emitter table (objects.rs:19-24) in instance order
  std::vector<float4> shade((size_t)std::max(np, 1) * kRecRows, make_float4(0, 0, 0, 0)), emit;      // the 128-B primitive records (lr_device.h): rows 0-3 shading, 4-5 own box
  std::vector<uint8_t> qid((size_t)np);
  std::vector<float> area((size_t)np);
  std::vector<uint8_t> sliver((size_t)np, 0);
  std::vector<float> tri_a((size_t)np, 0.0f);                        // |e1||e2| per triangle (0 for spheres): the scale of Moeller-Trumbore's absolute error
  for (int q = 0; q < kNumShadeQueues - 1; ++q) s.mat_present[q] = false;
  for (int i = 0; i < np; ++i) {
    const LrPrimitive& p = dprims[i];
    if (p.material < 0 || p.material >= d.n_materials) fail(LR_EINVAL, "primitive material index out of range");
    uint32_t mw = (uint32_t)p.material;
    if (p.type == LR_PRIM_TRIANGLE) {                            // triangle.rs:25-40
      float e1[3] = {p.v[3] - p.v[0], p.v[4] - p.v[1], p.v[5] - p.v[2]};
      float e2[3] = {p.v[6] - p.v[0], p.v[7] - p.v[1], p.v[8] - p.v[2]};
      float c[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
      // the device's 1/det is the IEEE quotient only below 2^126 (lr_math.h rcp_exact_mid); |det| <= |e1| |e2| |d|
      double l1 = std::sqrt((double)e1[0] * e1[0] + (double)e1[1] * e1[1] + (double)e1[2] * e1[2]);
      double l2 = std::sqrt((double)e2[0] * e2[0] + (double)e2[1] * e2[1] + (double)e2[2] * e2[2]);
      if (!(l1 * l2 < 1.329227995784916e36)) fail(LR_EUNSUPPORTED, "triangle " + std::to_string(i) + ": |e1| |e2| >= 2^120 is outside the exact-arithmetic range of the device path");
      float nrm = std::sqrt(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]);
      shade[kRecRows * (size_t)i] = make_float4(c[0] / nrm, c[1] / nrm, c[2] / nrm, __builtin_bit_cast(float, mw));
      area[i] = nrm * 0.5f;
      shade[kRecRows * (size_t)i + 4] = make_float4(std::fmin(std::fmin(p.v[0], p.v[3]), p.v[6]), std::fmin(std::fmin(p.v[1], p.v[4]), p.v[7]),      // triangle.rs:102-118
                                        std::fmin(std::fmin(p.v[2], p.v[5]), p.v[8]), 0.0f);
      shade[kRecRows * (size_t)i + 5] = make_float4(std::fmax(std::fmax(p.v[0], p.v[3]), p.v[6]), std::fmax(std::fmax(p.v[1], p.v[4]), p.v[7]),
                                            std::fmax(std::fmax(p.v[2], p.v[5]), p.v[8]), 0.0f);
      // triangle.rs:69-100 divides by det = e1 . (d x e2) = |e1||e2| sin(phi) cos(theta): with a small angle phi between the edges
      // at p0 the distance t = (e2 . qv) / det carries a relative error of ~eps / (sin(phi) cos(theta)) for EVERY direction, enough
      // to report the hit in front of the triangle's own (padded) box.  Such triangles are exempt from distance culling.
      {
        const double cx = (double)e1[1] * e2[2] - (double)e1[2] * e2[1], cy = (double)e1[2] * e2[0] - (double)e1[0] * e2[2], cz = (double)e1[0] * e2[1] - (double)e1[1] * e2[0];
        const double sinphi_l1l2 = std::sqrt(cx * cx + cy * cy + cz * cz);
        sliver[i] = !(sinphi_l1l2 * kSliverInvSin >= l1 * l2) ? 1 : 0;     // sin(phi) < 1 / kSliverInvSin (also degenerate / NaN)
        tri_a[i] = (float)(l1 * l2 * (1.0 + 1e-6));
      }
    } else if (p.type == LR_PRIM_SPHERE) {                       // sphere.rs:21-29
      shade[kRecRows * (size_t)i] = make_float4(p.v[0], p.v[1], p.v[2], __builtin_bit_cast(float, mw | 0x80000000u));
      area[i] = 4.0f * kPi * (p.v[3] * p.v[3]);
      shade[kRecRows * (size_t)i + 4] = make_float4(p.v[0] - p.v[3], p.v[1] - p.v[3], p.v[2] - p.v[3], p.v[3] * p.v[3]);   // sphere.rs:31-38; .w = r^2 (k_path_tree re-derives a parked hit's distance)
      shade[kRecRows * (size_t)i + 5] = make_float4(p.v[0] + p.v[3], p.v[1] + p.v[3], p.v[2] + p.v[3], 0.0f);
    } else fail(LR_EINVAL, "unknown primitive type");
    for (int k = 0; k < 3; ++k) shade[kRecRows * (size_t)i + 1 + k] = mats[3 * (size_t)p.material + k];
    int mt = d.materials[p.material].type;
    qid[i] = (uint8_t)mt;
    s.mat_present[mt] = true;
  }
  float emission_area = 0.0f;
  std::vector<int> emitters;
  for (int i = 0; i < np; ++i) {                                 // (i = the caller's index: instance order)
    const LrMaterial& m = d.materials[d.prims[i].material];
    if (m.type != LR_MAT_LAMBERT) continue;
    float e2 = m.emission[0] * m.emission[0] + m.emission[1] * m.emission[1] + m.emission[2] * m.emission[2];
    if (e2 > 0.0f) emitters.push_back(i);
  }
  for (int i : emitters) emission_area += area[rank[i]];         // objects.rs:24 (.sum() in instance order)
  {
    float cum = 0.0f;
    for (int i : emitters) {
      const LrPrimitive& p = d.prims[i];
      const float a_i = area[rank[i]];
      cum += a_i;                                                // objects.rs:41
      float pdf = (1.0f / a_i) * a_i / emission_area;            // objects.rs:46 with triangle.rs:147 / sphere.rs:82
      if (p.type == LR_PRIM_TRIANGLE) {
        emit.push_back(make_float4(p.v[0], p.v[1], p.v[2], __builtin_bit_cast(float, (uint32_t)LR_PRIM_TRIANGLE)));
        emit.push_back(make_float4(p.v[3], p.v[4], p.v[5], pdf));
        emit.push_back(make_float4(p.v[6], p.v[7], p.v[8], cum));
      } else {
        emit.push_back(make_float4(p.v[0], p.v[1], p.v[2], __builtin_bit_cast(float, (uint32_t)LR_PRIM_SPHERE)));
        emit.push_back(make_float4(p.v[3], 0.0f, 0.0f, pdf));
        emit.push_back(make_float4(0.0f, 0.0f, 0.0f, cum));
      }
    }
  }

  // BVH nodes + primitives in leaf order
  std::vector<float4> nodes, prims((size_t)std::max(np, 1) * 3, make_float4(0, 0, 0, 0));
  s.bvh_build_ms = 0.0;
  bool built_on_device = false;
  if (device_bvh && np >= 2) {
    s.nodes.ensure((size_t)(np - 1) * 4); s.prims.ensure((size_t)np * 3);
    int height = 0; std::string err;
    int rc = lbvh_build(d.prims, np, d.camera.aperture_position, s.stream, s.nodes.p, s.prims.p, &height, &s.bvh_build_ms, err);
    if (rc != LR_OK) fail(rc, "device BVH build: " + err);
    if (height < 1 || height > 95) fail(LR_EUNSUPPORTED, "device BVH too deep for the traversal stack (" + std::to_string(height) + ")");
    s.stack_depth = height + 1;
    built_on_device = true;
  } else if (device_bvh) {
    // 0 or 1 primitive: a root with (at most) one leaf
    nodes.assign(4, make_float4(0, 0, 0, 0));
    int leaf = np == 1 ? ~(int)((0u << 3) | 1u) : ~0;
    if (np == 1) {
      const LrPrimitive& p = d.prims[0];
      float big = 3.0e38f;
      nodes[0] = make_float4(-big, big, 0, 0); nodes[1] = make_float4(-big, big, 0, 0); nodes[2] = make_float4(-big, big, 0, 0);
      if (p.type == LR_PRIM_TRIANGLE) {
        prims[0] = make_float4(p.v[0], p.v[1], p.v[2], __builtin_bit_cast(float, 0u));
        prims[1] = make_float4(p.v[3] - p.v[0], p.v[4] - p.v[1], p.v[5] - p.v[2], 0.0f);
        prims[2] = make_float4(p.v[6] - p.v[0], p.v[7] - p.v[1], p.v[8] - p.v[2], 0.0f);
      } else {
        prims[0] = make_float4(p.v[0], p.v[1], p.v[2], __builtin_bit_cast(float, 0x80000000u));
        prims[1] = make_float4(p.v[3], p.v[3] * p.v[3], 0.0f, 0.0f);
      }
    }
    nodes[3] = make_float4(__builtin_bit_cast(float, leaf), __builtin_bit_cast(float, ~0), 0.0f, 0.0f);
    s.stack_depth = 2;
  } else {
    nodes.assign((size_t)d.n_bvh_nodes * 4, make_float4(0, 0, 0, 0));
    std::vector<char> seen((size_t)np, 0);
    std::vector<char> referenced((size_t)d.n_bvh_nodes, 0);        // a tree, not a DAG: every inner node has exactly one parent
    for (int i = 0; i < d.n_bvh_nodes; ++i) {
      const LrBvhNode& n = d.bvh_nodes[i];
      nodes[4 * i] = make_float4(n.x[0], n.x[1], n.x[2], n.x[3]);
      nodes[4 * i + 1] = make_float4(n.y[0], n.y[1], n.y[2], n.y[3]);
      nodes[4 * i + 2] = make_float4(n.z[0], n.z[1], n.z[2], n.z[3]);
      nodes[4 * i + 3] = make_float4(__builtin_bit_cast(float, n.child[0]), __builtin_bit_cast(float, n.child[1]), 0.0f, 0.0f);
      for (int c = 0; c < 2; ++c) {
        int ch = n.child[c];
        if (ch >= 0) {
          if (ch >= d.n_bvh_nodes || ch <= i) fail(LR_EINVAL, "BVH child index out of order");
          if (referenced[ch]) fail(LR_EINVAL, "BVH node referenced by two parents");     // a shared subtree would be expanded once per parent (exponential)
          referenced[ch] = 1;
          continue;
        }
        uint32_t enc = (uint32_t)~ch, first = enc >> 3, count = enc & 7u;
        if ((uint64_t)first + count > (uint64_t)np) fail(LR_EINVAL, "BVH leaf range out of bounds");
        for (uint32_t k = first; k < first + count; ++k) {
          int id = dorder[k];
          if (seen[id]) fail(LR_EINVAL, "BVH leaf ranges overlap");
          seen[id] = 1;
          const LrPrimitive& p = dprims[id];
          if (p.type == LR_PRIM_TRIANGLE) {                        // e1, e2 of triangle.rs:71-72
            prims[3 * k] = make_float4(p.v[0], p.v[1], p.v[2], __builtin_bit_cast(float, (uint32_t)id));
            prims[3 * k + 1] = make_float4(p.v[3] - p.v[0], p.v[4] - p.v[1], p.v[5] - p.v[2], 0.0f);
            prims[3 * k + 2] = make_float4(p.v[6] - p.v[0], p.v[7] - p.v[1], p.v[8] - p.v[2], 0.0f);
          } else {
            prims[3 * k] = make_float4(p.v[0], p.v[1], p.v[2], __builtin_bit_cast(float, (uint32_t)id | 0x80000000u));
            prims[3 * k + 1] = make_float4(p.v[3], p.v[3] * p.v[3], 0.0f, 0.0f);
            prims[3 * k + 2] = make_float4(0, 0, 0, 0);
          }
        }
      }
    }
    for (int i = 0; i < np; ++i) if (!seen[i]) fail(LR_EINVAL, "BVH does not reference every primitive");
    for (int i = 1; i < d.n_bvh_nodes; ++i) if (!referenced[i]) fail(LR_EINVAL, "BVH node without a parent");
    if (d.bvh_max_depth < 1 || d.bvh_max_depth > 96) fail(LR_EINVAL, "bvh_max_depth out of range (1..96)");
    s.stack_depth = d.bvh_max_depth + 1;
  }

  std::vector<float4> texels;
  std::vector<uint32_t> texels_rgbe;
  unsigned long long sky_sum = 0;                                    // checksum of the caller's floats (k_sky_checksum's formula)
  size_t n_texels = 0;
  if (d.sky.type == LR_SKY_IBL) {
    if (d.sky.height <= 0 || !d.sky.texels) fail(LR_EINVAL, "IBL sky without texels");
    const size_t n = n_texels = (size_t)d.sky.height * (size_t)d.sky.height * 2;
    // A map that came out of an .hdr file holds Radiance RGBE values, c * 2^(e - 136) per channel with one e per texel (the `image`
    // crate's decode behind sky.rs:45-48).  Re-encode every texel and keep 4 B instead of 16 B where the decode gives back the
    // caller's bits; one texel that does not (a procedural float map, a negative or denormal value) keeps the whole map as float4.
    bool rgbe_ok = !(lr_knob("LR_SKY_FLOAT4") && std::atoi(lr_knob("LR_SKY_FLOAT4")) == 1);
    texels_rgbe.resize(rgbe_ok ? n : 0);
    for (size_t i = 0; i < n; ++i) {
      const float c[3] = {d.sky.texels[3 * i], d.sky.texels[3 * i + 1], d.sky.texels[3 * i + 2]};
      uint32_t bits[3]; std::memcpy(bits, c, sizeof(bits));
      sky_sum += ((unsigned long long)bits[0] + 3ull * bits[1] + 5ull * bits[2]) * (2ull * i + 1ull);
      if (!rgbe_ok) continue;
      const float m = std::max(c[0], std::max(c[1], c[2]));
      uint32_t word = 10u << 24;                                      // (0, 0, 0): any exponent decodes to zero; 10 keeps the scale a normal float
      if ((bits[0] | bits[1] | bits[2]) != 0u) {
        int k = 0;
        if (!(m > 0.0f) || !std::isfinite(m)) { rgbe_ok = false; continue; }
        (void)std::frexp(m, &k);                                       // m = f * 2^k, f in [0.5, 1): the largest mantissa lands in [128, 256)
        const int e = k + 128;
        if (e < 10 || e > 255) { rgbe_ok = false; continue; }
        uint32_t q[3]; bool exact = true;
        for (int a = 0; a < 3; ++a) {
          const float scaled = std::ldexp(c[a], 136 - e);
          if (!(scaled >= 0.0f && scaled <= 255.0f) || scaled != std::floor(scaled)) { exact = false; break; }
          q[a] = (uint32_t)scaled;
          const float back = (float)q[a] * std::ldexp(1.0f, e - 136);
          if (std::memcmp(&back, &c[a], 4) != 0) { exact = false; break; }   // (also rejects -0.0)
        }
        if (!exact) { rgbe_ok = false; continue; }
        word = q[0] | (q[1] << 8) | (q[2] << 16) | ((uint32_t)e << 24);
      }
      texels_rgbe[i] = word;
    }
    if (!rgbe_ok) {
      texels_rgbe.clear();
      texels.resize(n);
      for (size_t i = 0; i < n; ++i) texels[i] = make_float4(d.sky.texels[3 * i], d.sky.texels[3 * i + 1], d.sky.texels[3 * i + 2], 0.0f);
    }
  } else if (d.sky.type != LR_SKY_UNIFORM) fail(LR_EINVAL, "unknown sky type");

  {
    // 4-wide nodes for the device (the binary tree of the description, or the one the LBVH kernels just wrote)
    if (built_on_device) {
      nodes.resize((size_t)(np - 1) * 4);
      HIP_OK(hipMemcpyAsync(nodes.data(), s.nodes.p, nodes.size() * sizeof(float4), hipMemcpyDeviceToHost, s.stream));
      HIP_OK(hipStreamSynchronize(s.stream));
    }
    // leaf position -> |e1||e2| of the triangle there (leaf order: the description's permutation, or the ids the device builder left in the rows)
    std::vector<float> leaf_area((size_t)std::max(np, 1), 0.0f);
    if (built_on_device) {
      std::vector<float4> rows((size_t)np * 3);
      HIP_OK(hipMemcpyAsync(rows.data(), s.prims.p, rows.size() * sizeof(float4), hipMemcpyDeviceToHost, s.stream));
      HIP_OK(hipStreamSynchronize(s.stream));
      for (int k = 0; k < np; ++k) { uint32_t id = __builtin_bit_cast(uint32_t, rows[3 * (size_t)k].w) & 0x7fffffffu; if (id < (uint32_t)np) leaf_area[k] = tri_a[id]; }
    } else if (!device_bvh) {
      for (int k = 0; k < np; ++k) leaf_area[k] = tri_a[dorder[k]];
    } else if (np == 1) leaf_area[0] = tri_a[0];
    std::vector<float4> wide;
    wide.reserve(nodes.size());
    int need = 0;
    Wide4Builder w4{nodes, wide};
    w4.leaf_area = &leaf_area;
    if (const char* e = lr_knob("LR_CULL_SLACK")) { const double v = std::atof(e); if (!(v >= 0.0)) fail(LR_EINVAL, "LR_CULL_SLACK must be >= 0"); w4.cull_slack = v; }
    w4.build(0, &need);
    if (need > 150) fail(LR_EUNSUPPORTED, "BVH too deep for the traversal stack");
    s.stack_depth = need + 1;
    {
      size_t n_sliver = 0, n_nocull = 0;
      for (int i = 0; i < np; ++i) n_sliver += sliver[i];
      for (size_t k = 0; k < wide.size() / kNodeRows; ++k) n_nocull += wide[k * kNodeRows + 2].z >= 1.0f;      // 2 kappa >= 1: the slack exceeds the child's own distance, i.e. no culling
      s.tree_info[0] = (int)(wide.size() / kNodeRows); s.tree_info[1] = (int)n_nocull; s.tree_info[2] = (int)n_sliver; s.tree_info[3] = need;
      if (std::getenv("LR_DEBUG")) std::fprintf(stderr, "[lr] wide BVH: %zu binary nodes -> %zu 4-wide nodes of %d B, stack need %d; %zu sliver triangles, %zu nodes without distance culling\n",
                   nodes.size() / 4, wide.size() / kNodeRows, kNodeRows * 16, need, n_sliver, n_nocull);
    }
    s.nodes.upload(wide, s.stream);
    if (!built_on_device) s.prims.upload(prims, s.stream);
  }
  if (np > 0 && np <= kFlatMax) {
    // small scenes are tested without a tree (traverse_flat): the same rows IN DEVICE-ID ORDER -- the reference's candidate order
    // (above) -- so that "first strictly nearer hit wins" is bvh.rs:131-141's first minimum; padded by three primitives (the loop
    // requests whole groups)
    std::vector<float4> flat((size_t)np * 3 + 9, make_float4(0, 0, 0, 0));
    for (int id = 0; id < np; ++id) {
      const LrPrimitive& p = dprims[id];
      if (p.type == LR_PRIM_TRIANGLE) {
        flat[3 * id] = make_float4(p.v[0], p.v[1], p.v[2], __builtin_bit_cast(float, (uint32_t)id));
        flat[3 * id + 1] = make_float4(p.v[3] - p.v[0], p.v[4] - p.v[1], p.v[5] - p.v[2], 0.0f);
        flat[3 * id + 2] = make_float4(p.v[6] - p.v[0], p.v[7] - p.v[1], p.v[8] - p.v[2], 0.0f);
      } else {
        flat[3 * id] = make_float4(p.v[0], p.v[1], p.v[2], __builtin_bit_cast(float, (uint32_t)id | 0x80000000u));
        flat[3 * id + 1] = make_float4(p.v[3], p.v[3] * p.v[3], 0.0f, 0.0f);
      }
    }
    s.flat.upload(flat, s.stream);
  }
  s.shade.upload(shade, s.stream);
  s.emit.upload(emit, s.stream); s.texels.upload(texels, s.stream);
  if (!texels_rgbe.empty()) s.texels_rgbe.upload(texels_rgbe, s.stream);
  s.prim_qid.upload(qid, s.stream);
  HIP_OK(hipStreamSynchronize(s.stream));

  DevScene& v = s.dev;
  std::memset(&v, 0, sizeof(v));
  v.nodes = s.nodes.p; v.prims = s.prims.p; v.shade = s.shade.p; v.pbox = s.shade.p + 4; v.emit = s.emit.p;
  v.texels = s.texels.p; v.prim_qid = s.prim_qid.p;
  v.texels_rgbe = texels_rgbe.empty() ? nullptr : s.texels_rgbe.p;
  v.n_flat = (np > 0 && np <= kFlatMax) ? np : 0;
  v.n_emitters = (int)emitters.size(); v.emission_area = emission_area;
  v.sky_type = d.sky.type; v.sky_color[0] = d.sky.color[0]; v.sky_color[1] = d.sky.color[1]; v.sky_color[2] = d.sky.color[2];
  v.sky_h = d.sky.height; v.sky_lon = d.sky.longitude_offset;
  const LrCamera& c = d.camera;
  DevCamera& dc = v.cam;
  dc.type = c.type; dc.res_w = c.resolution[0]; dc.res_h = c.resolution[1];
  for (int k = 0; k < 3; ++k) { dc.forward[k] = c.forward[k]; dc.right[k] = c.right[k]; dc.up[k] = c.up[k]; dc.position[k] = c.position[k]; dc.aperture_position[k] = c.aperture_position[k]; }
  dc.sensor_w = c.sensor_size[0]; dc.sensor_h = c.sensor_size[1];
  dc.aperture_sensor_distance = c.aperture_sensor_distance; dc.aperture_radius = c.aperture_radius;
  dc.focus_distance = c.focus_distance; dc.sensor_pixel_area = c.sensor_pixel_area;
  dc.weight2 = 1.0f;
  if (c.type == LR_CAMERA_THIN_LENS) {                           // camera.rs:426,438 pdfs; main.rs:118 sens / pdf
    float sensor_pdf = 1.0f / c.sensor_pixel_area;
    float aperture_pdf = 1.0f / (kPi * c.aperture_radius * c.aperture_radius);
    dc.weight2 = c.sensor_sensitivity / (sensor_pdf * aperture_pdf);
  }
  s.film_w = c.resolution[0]; s.film_h = c.resolution[1];
  s.n_prims = np;
  {
    // bounding box of the primitives for the ray sort's 4x4x4 origin cells (only groups rays; never decides anything)
    float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    for (int i = 0; i < np; ++i) {
      const LrPrimitive& p = d.prims[i];
      const int nv = p.type == LR_PRIM_TRIANGLE ? 3 : 1;
      const float r = p.type == LR_PRIM_TRIANGLE ? 0.0f : std::fabs(p.v[3]);
      for (int k = 0; k < nv; ++k) for (int a = 0; a < 3; ++a) { lo[a] = std::fmin(lo[a], p.v[3 * k + a] - r); hi[a] = std::fmax(hi[a], p.v[3 * k + a] + r); }
    }
    for (int a = 0; a < 3; ++a) {
      float ext = hi[a] - lo[a];
      v.key_lo[a] = np > 0 ? lo[a] : 0.0f;
      v.key_scale[a] = (np > 0 && ext > 0.0f && ext < 3.0e38f) ? 4.0f / ext : 0.0f;
    }
  }
  if (v.texels_rgbe) {
    // the DEVICE's decode of the whole RGBE map against the caller's floats (one position-weighted checksum over the f32 bits):
    // whatever the float mode of the kernels, the map the renders read is the map that was handed in -- or it is stored as float4
    DevBuf<unsigned long long> sum; sum.ensure(1);
    HIP_OK(hipMemsetAsync(sum.p, 0, sizeof(unsigned long long), s.stream));
    hipLaunchKernelGGL(k_sky_checksum, dim3(1024), dim3(256), 0, s.stream, s.dev, (uint64_t)n_texels, sum.p);
    HIP_OK(hipGetLastError());
    unsigned long long got = 0;
    HIP_OK(hipMemcpyAsync(&got, sum.p, sizeof(got), hipMemcpyDeviceToHost, s.stream));
    HIP_OK(hipStreamSynchronize(s.stream));
    if (std::getenv("LR_DEBUG")) std::fprintf(stderr, "[lr] IBL map: %zu texels as RGBE words (%.1f MB instead of %.1f MB), device decode checksum %s\n",
                                              n_texels, n_texels * 4e-6, n_texels * 16e-6, got == sky_sum ? "matches" : "DIFFERS -> float4");
    if (got != sky_sum) {
      texels.resize(n_texels);
      for (size_t i = 0; i < n_texels; ++i) texels[i] = make_float4(d.sky.texels[3 * i], d.sky.texels[3 * i + 1], d.sky.texels[3 * i + 2], 0.0f);
      s.texels.upload(texels, s.stream);
      HIP_OK(hipStreamSynchronize(s.stream));
      s.texels_rgbe.release();
      v.texels = s.texels.p; v.texels_rgbe = nullptr;
    }
  }
}

// Work items = (chunk, pixel); a pixel's spp samples are cut into chunks, each folded in order by one lane into a chunk sum, and
// k_resolve adds a pixel's chunk sums in chunk order (main.rs:92-121 is one flat fold; the difference is a few ulp of radiance).
// The cut is a function of spp ONLY, so the film does not depend on tiling, slot count, pipeline or GPU count.
//   body   chunks of L samples: L = 8 ... 16 up to 1024 spp (at most 64 of them), 16 ... 32 beyond (at most 256)
//   taper  the LAST samples of every pixel in chunks of L/2, L/4, ... 1 (kTaperRepeat of each, twice as many of 1): items are
//          dispensed chunk-major, so a render ends on single-sample items.  A lane ends the render inside its last item, and with
//          one chunk length throughout that tail was 2-3 chunk times of the slowest lanes with every other lane idle: 3 ms of a
//          177-ms frame on configs[1], 7.5 ms of 206 on config 3, 14-18 ms on config 5 -- a FIXED cost per call, paid in full by
//          each rank of an 8-GPU job that renders 1/8 of the frame (profiles/r05_timeline_before.txt: 12 % / 23 % / 6 % there).
constexpr uint32_t kTaperRepeat = 16;
constexpr uint32_t kSubBandShift = 17;    // work items are dealt in sub-bands of 2^17 consecutive pixel ranks (lr_device.h)
std::vector<uint32_t> chunk_schedule(uint32_t spp) {
  uint32_t n0 = spp <= 1024 ? std::min<uint32_t>(64, std::max<uint32_t>(1, spp / 8)) : std::min<uint32_t>(256, spp / 16);
  uint32_t L = (spp + n0 - 1) / n0;
  uint32_t level_sum = 0;
  uint32_t R = kTaperRepeat;
  if (const char* e = lr_knob("LR_TAPER")) { int v = std::atoi(e); if (v >= 0 && v <= 4096) R = (uint32_t)v; }            // diagnostic: chunks per taper level
  if (const char* e = lr_knob("LR_CHUNK_LEN")) { int v = std::atoi(e); if (v >= 1 && v <= 4096) L = std::min<uint32_t>((uint32_t)v, spp); }   // diagnostic: body chunk length
  for (uint32_t l = L / 2; l >= 1; l /= 2) level_sum += l;
  if (level_sum == 0) R = 0;
  else R = std::min<uint32_t>(R, spp / (2 * (level_sum + 1)));            // the taper takes at most half of the samples
  const uint32_t taper = R * (level_sum + 1), body = spp - taper;
  std::vector<uint32_t> start;
  uint32_t at = 0;
  for (; at + L <= body; at += L) start.push_back(at);
  if (at < body) { start.push_back(at); at = body; }                      // (a shorter last body chunk)
  for (uint32_t l = L / 2; l >= 1 && R > 0; l /= 2)
    for (uint32_t k = 0; k < R * (l == 1 ? 2u : 1u); ++k) { start.push_back(at); at += l; }
  start.push_back(spp);
  return start;
}

int grid_for(const void* kernel, int n_cus, size_t lds, uint32_t work_items) {
  int per_cu = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, kBlock, lds) != hipSuccess || per_cu < 1) per_cu = 1;
  per_cu = std::min(per_cu, 8);
  long long want = ((long long)work_items + kBlock - 1) / kBlock;   // callers pass segments * kBlock for segment kernels
  long long cap = (long long)n_cus * per_cu;
  return (int)std::max<long long>(1, std::min(want, cap));
}

struct Launcher {
  LrScene& s; bool profile; int iter = 0;
  bool timed(int k) const { return profile && (k == LR_K_GENERATE || k == LR_K_RESOLVE || k == LR_K_RESIDENT || k == LR_K_PATH || iter % kProfileStride == 0) && s.pools[k].used < kEventPool; }
  template <class F> void run(int k, F&& launch, hipStream_t on = nullptr) {
    bool t = timed(k);
    EventPool& p = s.pools[k];
    hipStream_t st = on ? on : s.stream;
    if (t) HIP_OK(hipEventRecord(p.a[p.used], st));
    launch();
    HIP_OK(hipGetLastError());
    if (t) { HIP_OK(hipEventRecord(p.b[p.used], st)); p.used++; }
    s.stats.kernel_launches[k]++;
  }
};

// LR_STACK_LDS=<n> (diagnostic): keep only n stack entries per lane in LDS so that tests reach the spill path
int stack_lds_limit() {
  if (const char* e = lr_knob("LR_STACK_LDS")) { int v = std::atoi(e); if (v >= 1 && v <= kStackLdsMax) return v; }
  return kStackLdsMax;
}

void render_impl(LrScene& s, const LrRenderParams& rp_in, const LrTile* tiles, int n_tiles, bool want_packed = false) {
  if (rp_in.spp <= 0) fail(LR_EINVAL, "spp must be positive");
  if (rp_in.integrator != LR_INTEGRATOR_PT && rp_in.integrator != LR_INTEGRATOR_PT_DIRECT) fail(LR_EINVAL, "unknown integrator");
  if (rp_in.depth < 0 || rp_in.depth_limit < 0) fail(LR_EINVAL, "negative depth");
  if (n_tiles < 0 || (n_tiles > 0 && !tiles)) fail(LR_EINVAL, "bad tile list");
  HIP_OK(hipSetDevice(s.device));
  HIP_OK(hipStreamSynchronize(s.stream));                                  // (a call that failed half-way may have left work on the streams)
  for (auto& g : s.gstream) if (g) HIP_OK(hipStreamSynchronize(g));
  const int W = s.film_w, H = s.film_h;
  // tile list -> prefix of pixel ranks
  std::vector<int4> tl; std::vector<uint32_t> prefix; uint64_t npix64 = 0;
  for (int i = 0; i < n_tiles; ++i) {
    const LrTile& t = tiles[i];
    if (t.w < 0 || t.h < 0 || t.x0 < 0 || t.y0 < 0 || (long long)t.x0 + t.w > W || (long long)t.y0 + t.h > H) fail(LR_EINVAL, "tile outside the film");
    if (t.w == 0 || t.h == 0) continue;
    tl.push_back(make_int4(t.x0, t.y0, t.w, t.h)); prefix.push_back((uint32_t)npix64);
    npix64 += (uint64_t)t.w * (uint64_t)t.h;
  }
  prefix.push_back((uint32_t)npix64);
  if (npix64 > (uint64_t)W * (uint64_t)H) fail(LR_EINVAL, "tiles overlap (more tile pixels than film pixels)");
  {
    // the header promises disjoint tiles: a pixel in two tiles would be rendered twice and resolved twice.  One occupancy bit per
    // film pixel, tested and set a 64-bit word at a time: linear in the pixels the tiles cover, whatever their shapes (a
    // per-pixel or per-scanline job list costs what a few large tiles cost)
    const size_t row_words = ((size_t)W + 63) / 64;
    std::vector<uint64_t> occ(row_words * (size_t)H, 0);
    for (const int4& t : tl) {                                        // int4 {x0, y0, w, h}
      const int x0 = t.x, x1 = t.x + t.z;
      for (int y = t.y; y < t.y + t.w; ++y) {
        uint64_t* row = occ.data() + (size_t)y * row_words;
        for (int wi = x0 >> 6; wi <= (x1 - 1) >> 6; ++wi) {
          const int lo = std::max(x0, wi << 6) & 63, hi = std::min(x1, (wi + 1) << 6) - (wi << 6);     // bits [lo, hi) of this word
          const uint64_t m = (hi >= 64 ? ~0ull : ((1ull << hi) - 1ull)) & ~((1ull << lo) - 1ull);
          if (row[wi] & m) fail(LR_EINVAL, "tiles overlap");
          row[wi] |= m;
        }
      }
    }
  }
  const uint32_t n_pix = (uint32_t)npix64;
  // chunks: a function of spp ONLY, so the image does not depend on tiling, slot count or GPU count
  // up to 1024 spp: chunks of >= 8 samples, at most 64; beyond: chunks of >= 16 samples, at most 256 -- a slot ends the
  // render inside its last chunk, so the chunk length is the tail of the render (weak scaling raises spp per pixel)
  const std::vector<uint32_t> chunks = chunk_schedule((uint32_t)rp_in.spp);
  const uint32_t n_chunks = (uint32_t)chunks.size() - 1;
  uint32_t chunk_spp = 1;                                                 // the longest chunk
  for (uint32_t c = 0; c < n_chunks; ++c) chunk_spp = std::max(chunk_spp, chunks[c + 1] - chunks[c]);
  // ---- pixel bands --------------------------------------------------------------------------------------------------------
  // Img::new is W x H (img.rs:13) whatever the spp; the chunk sums of a call are n_pix x n_chunks x 16 B (17 GB for config 5 at
  // 8192 spp in round 4).  A call whose sums exceed 3 GiB is rendered in BANDS of consecutive pixel ranks with at most 2 GiB of sums
  // each -- one launch + one k_resolve per band, one after the other on the call's stream.  A pixel's samples, chunks and fold
  // order do not depend on the band it falls in: same film bits.  The tail a band adds is short (chunk_schedule's taper) and the
  // bands keep the rays in flight within a strip of the film: measured against one launch for the whole frame, config 4 (8 bands)
  // 1317.6 vs 1315.3 ms, config 5 at 2048 spp (12 bands) 2038.8 vs 2055.8 ms.  (Two bands in flight on two streams -- the first
  // workgroups of band b + 1 starting on the compute units the last ones of band b leave -- was built and measured: 1314.9 / 2055.8 ms,
  // no better, and every launch's dispatch-to-end time then includes its wait for the previous band: gpurun_out/r05l.)
  const uint64_t row_bytes = (uint64_t)n_chunks * sizeof(float4);
  const uint64_t band_budget = 2ull << 30;                                 // chunk sums of one band
  uint32_t band_pix = n_pix;
  if ((uint64_t)n_pix * row_bytes > band_budget + (band_budget >> 1)) {          // (a call up to 3 GiB stays one launch)
    const uint64_t want = ((uint64_t)n_pix * row_bytes + band_budget - 1) / band_budget;
    band_pix = (uint32_t)(((uint64_t)n_pix + want - 1) / want);
  }
  if (const char* e = lr_knob("LR_BAND_PIX")) { long long v = std::atoll(e); if (v >= 1 && v < (long long)n_pix) band_pix = (uint32_t)v; }   // tests / diagnostics
  if (band_pix < n_pix) band_pix = (band_pix + 1023u) / 1024u * 1024u;
  const uint32_t n_bands = n_pix > 0 ? (n_pix + band_pix - 1) / band_pix : 1;
  uint64_t n_items64 = (uint64_t)std::min(band_pix, n_pix) * n_chunks;      // of one band (the last one may be smaller)
  if (n_items64 >= 0xffffffffull - (1ull << 24)) fail(LR_EUNSUPPORTED, "too many work items in one band (LR_BAND_PIX too large for this spp)");
  const uint32_t n_items = (uint32_t)n_items64;
  const bool count = (rp_in.flags & LR_FLAG_COUNT) != 0;
  // pipeline: resident (one launch, path state in LDS) when state + traversal stack stay under 40 KB per workgroup
  // (>= 4 workgroups per CU; 6 for flat scenes), streaming otherwise
  const size_t stack_lds = s.dev.n_flat > 0 ? 0 : (size_t)s.stack_depth * kBlock * 4;   // resident: the whole stack in LDS
  uint32_t present_mask = 0;
  for (int k = 0; k < kNumShadeQueues - 1; ++k) if (s.mat_present[k]) present_mask |= 1u << k;
  const int n_lists = __builtin_popcount(present_mask) + 2;                 // one list per BSDF present + shadow + finish
  // flat scenes with SEVERAL BSDF lists, whose lists leave room, run 512-slot workgroups (three per CU = the same 24 waves;
  // lr_kernels.h kRSeg): measured +3.9 % on the BRDF row (four lists), but -2.5 % on the Lambert-only headline scene, where
  // there is little chunk waste to win and the 8-wave barriers cost more than they save
  bool big_block = s.dev.n_flat > 0 && n_lists >= 4 && (size_t)3 * (resident_lds_bytes(512, n_lists) + 512) <= 160 * 1024;
  if (const char* e = lr_knob("LR_RES_BLOCK")) {                                       // diagnostic override
    if (std::atoi(e) == 256) big_block = false;
    if (std::atoi(e) == 512 && s.dev.n_flat > 0 && (size_t)3 * (resident_lds_bytes(512, n_lists) + 512) <= 160 * 1024) big_block = true;
  }
  const int RB = big_block ? 512 : 256;
  const size_t resident_lds = (size_t)resident_lds_bytes(RB, n_lists) + stack_lds;
  bool resident = resident_lds <= (big_block ? 54 : 40) * 1024 && !count;
  if (rp_in.flags & LR_FLAG_STREAMING) resident = false;
  if ((rp_in.flags & LR_FLAG_RESIDENT) && resident_lds <= 156 * 1024 && !count) resident = true;
  // fused: one persistent launch in which a lane carries its path in registers (lr_path.h); flat scenes
  bool fused = false;
  {
    // default for flat scenes with ONE BSDF (the headline class: +10 % over the resident pipeline); with several
    // BSDFs in a wave the per-lane material dispatch loses to the resident pipeline's per-BSDF lists (brdf row: 7.8 vs 9.6 G/s)
    const char* pe = lr_knob("LR_PIPELINE");
    const bool forced = (rp_in.flags & LR_FLAG_FUSED) || (pe && std::strcmp(pe, "fused") == 0);
    if (s.dev.n_flat > 0 && !count && (forced || __builtin_popcount(present_mask) == 1)) fused = true;
    if (s.dev.n_flat == 0 && !count) fused = true;                       // tree scenes: k_path_tree (+56...68 % over the streaming pipeline)
    if (pe && (std::strcmp(pe, "resident") == 0 || std::strcmp(pe, "streaming") == 0)) fused = false;
    if (rp_in.flags & LR_FLAG_STREAMING) fused = false;
    // LR_FLAG_RESIDENT on a scene whose state + stack do not fit the LDS: the request cannot be honoured, and what runs then
    // is the DEFAULT choice (fused where it applies), not the streaming pipeline (that was up to 68 % slower)
    if ((rp_in.flags & LR_FLAG_RESIDENT) && resident) fused = false;
    if (fused) resident = false;
  }
  const int resident_per_cu = big_block ? 3 : std::max(1, std::min(LR_RES_WAVES, (int)((160 * 1024) / (resident_lds + 768))));
  // streaming: enough slots that a k_trace workgroup pass covers many rays per lane (the run-down of a pass's last
  // rays is what idles lanes: 1 M slots = 4 rays per lane left 23 % of the lanes busy in a node step), but no more
  // than 1/8 of the work items so that the render as a whole still has many iterations; 136 B of state per slot (<= 32 M slots)
  const uint32_t stream_slots = (uint32_t)std::min<uint64_t>(32u << 20, std::max<uint64_t>(1u << 20, n_items64 / 8));
  uint32_t n_slots = rp_in.path_slots > 0 ? (uint32_t)rp_in.path_slots : (resident ? (uint32_t)(s.n_cus * resident_per_cu * RB) : stream_slots);
  if (resident) n_slots = std::min<uint32_t>(n_slots, (uint32_t)(s.n_cus * resident_per_cu * RB));   // every workgroup must be resident: no grid-stride
  // fused: one path per lane of every resident wave.  The tree kernel keeps 10 stack entries per lane in LDS (16 in the streaming
  // kernels; measured free down to 10) beside the spare camera samples, so that 7 (pt) / 6 (pt-direct, thin lens) workgroups fit a CU.
  const void* fused_kernel = nullptr; size_t fused_lds = 0; int fused_stack = 0;
  if (fused) {
    uint32_t mt_mask = 0;
    for (int k = 0; k < kNumShadeQueues - 1; ++k) if (s.mat_present[k]) mt_mask |= 1u << k;
    const uint32_t only = mt_mask | 1u;                                // the kernels are instantiated for the BASELINE material sets and for "anything"
    int want_waves;
    if (s.dev.n_flat > 0) {
      if (s.dev.cam.type == LR_CAMERA_THIN_LENS)                       // one spare camera sample per lane beside the aperture points, two otherwise
        fused_kernel = mt_mask == 1u ? (const void*)k_path_flat<1u, 1> : (only == 9u ? (const void*)k_path_flat<9u, 1> : (const void*)k_path_flat<31u, 1>);
      else
        fused_kernel = mt_mask == 1u ? (const void*)k_path_flat<1u, 2> : (only == 9u ? (const void*)k_path_flat<9u, 2> : (const void*)k_path_flat<31u, 2>);
      want_waves = LR_PATH_WAVES;
    } else {
      const bool nee_k = rp_in.integrator == LR_INTEGRATOR_PT_DIRECT;  // (a pt-direct scene without emitters runs the NEE kernel: the branch is then never taken)
      fused_kernel = only == 1u ? (nee_k ? (const void*)k_path_tree<1u, true> : (const void*)k_path_tree<1u, false>)
                   : only == 9u ? (nee_k ? (const void*)k_path_tree<9u, true> : (const void*)k_path_tree<9u, false>)
                                : (nee_k ? (const void*)k_path_tree<31u, true> : (const void*)k_path_tree<31u, false>);
      want_waves = path_tree_waves(nee_k);
      fused_stack = std::min(s.stack_depth, lr_knob("LR_STACK_LDS") ? stack_lds_limit() : kStackLdsFused);
    }
    int fit = 0;
    hipFuncAttributes fa; HIP_OK(hipFuncGetAttributes(&fa, fused_kernel));
    for (;;) {                                                         // a scene class whose LDS need leaves a workgroup out gives up stack entries (down to 8) for it
      fused_lds = (size_t)fused_stack * kBlock * 4 + (s.dev.n_flat == 0 && s.dev.cam.type == LR_CAMERA_THIN_LENS ? kSpareLensBytes : 0);
      if (fused_lds > 48 * 1024) HIP_OK(hipFuncSetAttribute(fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fused_lds));
      HIP_OK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&fit, fused_kernel, kBlock, fused_lds));
      // the CU hands out its 160 KB in 1280-B granules (tools/micro/lds_granule.hip: 23 040 B fit seven times and 26 880 B six
      // times, one byte more does not; 27 104 B per workgroup ran five per CU where the runtime reported six, and the sixth of
      // every CU waited for a slot: -6 %), the occupancy query rounds less
      const size_t granule = 1280, per_wg = (fa.sharedSizeBytes + fused_lds + granule - 1) / granule * granule;
      fit = std::min<int>(fit, (int)((160 * 1024) / std::max<size_t>(per_wg, granule)));
      if (fit >= want_waves || fused_stack <= 8 || s.dev.n_flat > 0 || lr_knob("LR_STACK_LDS")) break;
      --fused_stack;
    }
    if (fit < 1) fail(LR_EUNSUPPORTED, "the fused kernel does not fit a compute unit");
    if (std::getenv("LR_DEBUG")) std::fprintf(stderr, "[lumilly_hip] fused kernel: %d workgroups per CU fit (%d wanted), %zu B of dynamic LDS\n", fit, want_waves, fused_lds);
    n_slots = (uint32_t)(s.n_cus * std::min(fit, want_waves) * kBlock);
    // LrRenderParams.path_slots is an UPPER bound here (one path per lane of a resident wave is the most the kernel can use)
    if (rp_in.path_slots > 0) n_slots = std::min<uint32_t>(n_slots, ((uint32_t)rp_in.path_slots + kBlock - 1) / kBlock * kBlock);
  }
  n_slots = std::max<uint32_t>(kSeg, std::min<uint32_t>(n_slots, ((n_items + kSeg - 1) / kSeg) * kSeg));
  n_slots = (n_slots + kSeg - 1) / kSeg * kSeg;
  const uint32_t n_seg = n_slots / kSeg;

  hipStream_t st = s.stream;
  s.ray_o.ensure(n_slots); s.ray_d.ensure(n_slots); s.hit.ensure(n_slots); s.thr.ensure(n_slots); s.rad.ensure(n_slots);
  s.acc.ensure(n_slots); s.sh_d.ensure(n_slots); s.sh_w.ensure(n_slots);
  // dense shading (k_shade_all): one launch per iteration over the slots themselves instead of one per class over lists;
  // then k_trace writes no lists and there is ONE shadow list per range (4 B per slot instead of 44)
  const bool dense_shade = !resident && !(lr_knob("LR_DENSE") && std::atoi(lr_knob("LR_DENSE")) == 0);
  const size_t shadow_lists = dense_shade ? 1 : (size_t)(kNumShadeQueues - 1);
  s.q_shade.ensure(dense_shade ? 1 : (size_t)kNumShadeQueues * n_slots); s.c_shade.ensure((size_t)kNumShadeQueues * n_seg);
  s.q_shadow.ensure(shadow_lists * n_slots); s.c_shadow.ensure((size_t)(kNumShadeQueues - 1) * n_seg);
  s.pool.ensure(n_seg);
  // Ray sort before trace / shadow (lr_kernels.h "Ray sort"): measured on the 100k-triangle configs it LOSES 6 % (lanes per
  // VALU instruction 20.9 -> 23.0, but HBM fetch x3 and L2 hit rate 0.67 -> 0.54: the sorted order gathers 16-B rows from all
  // over the range), so it is opt-in: LR_SORT=1.  DESIGN.md section 6 has the numbers.
  const bool sort_rays = !resident && s.dev.n_flat == 0 && lr_knob("LR_SORT") && std::atoi(lr_knob("LR_SORT")) == 1;
  if (sort_rays) { s.sort_key.ensure(n_slots); s.order.ensure(n_slots); }
  s.counters.ensure(4 + (size_t)n_bands);                                     // [0] dispenser (streaming), [1..3] retired slots per group, [4 + b] dispenser of band b
  s.stats_dev.ensure((size_t)kStatShards * kStatStride + 64);
  s.partial.ensure(n_items); s.rank_pixel.ensure(std::max<uint32_t>(n_pix, 1));
  if (s.film.n < (size_t)W * H * 3 || !s.film.p) {
    s.film.ensure((size_t)W * H * 3);
    HIP_OK(hipMemsetAsync(s.film.p, 0, (size_t)W * H * 3 * sizeof(float), st));
  }
  s.tiles.upload(tl, st); s.tile_prefix.upload(prefix, st); s.chunk_start.upload(chunks, st);
  if (!s.pinned) HIP_OK(hipHostMalloc((void**)&s.pinned, (8 + 2 * kStatShards * kStatStride) * sizeof(uint64_t)));
  if (!s.poll_ev[0]) { HIP_OK(hipEventCreate(&s.poll_ev[0])); HIP_OK(hipEventCreate(&s.poll_ev[1])); HIP_OK(hipEventCreate(&s.t_begin)); HIP_OK(hipEventCreate(&s.t_end)); }
  const bool profile = (rp_in.flags & LR_FLAG_PROFILE) != 0;
  if (profile) for (auto& p : s.pools) p.init();
  for (auto& p : s.pools) p.used = 0;

  DevState ds; std::memset(&ds, 0, sizeof(ds));
  ds.ray_o = s.ray_o.p; ds.ray_d = s.ray_d.p; ds.hit = s.hit.p; ds.thr = s.thr.p; ds.rad = s.rad.p; ds.acc = s.acc.p;
  ds.sh_d = s.sh_d.p; ds.sh_w = s.sh_w.p;
  ds.q_shade = s.q_shade.p; ds.c_shade = s.c_shade.p; ds.q_shadow = s.q_shadow.p; ds.c_shadow = s.c_shadow.p; ds.pool = s.pool.p;
  ds.sort_key = sort_rays ? s.sort_key.p : nullptr; ds.order = sort_rays ? s.order.p : nullptr;
  const bool shade_ordered = !resident && !(lr_knob("LR_SHADE_ORDER") && std::atoi(lr_knob("LR_SHADE_ORDER")) == 0);
  ds.shade_ordered = shade_ordered ? 1u : 0u;
  const size_t shade_lds = shade_ordered ? sizeof(ShadeOrderLds) : 0;
  ds.dense_shade = dense_shade ? 1u : 0u;
  ds.next_item = s.counters.p; ds.n_retired = s.counters.p + 1;
  if (want_packed) s.packed.ensure((size_t)std::max<uint32_t>(n_pix, 1) * 3);
  ds.stats = s.stats_dev.p; ds.partial = s.partial.p; ds.film = s.film.p; ds.packed = want_packed ? s.packed.p : nullptr;
  ds.tiles = s.tiles.p; ds.tile_prefix = s.tile_prefix.p; ds.n_tiles = (int)tl.size(); ds.rank_pixel = s.rank_pixel.p;
  ds.n_slots = n_slots; ds.n_seg = n_seg; ds.n_pix = n_pix; ds.n_chunks = n_chunks; ds.chunk_start = s.chunk_start.p; ds.n_items = n_items;
  ds.stack_depth = s.stack_depth;
#ifdef LR_TIMELINE
  s.timeline.ensure((size_t)3 * (n_slots / 64 + 8));
  HIP_OK(hipMemsetAsync(s.timeline.p, 0, (size_t)3 * (n_slots / 64 + 8) * sizeof(unsigned long long), st));
  ds.timeline = s.timeline.p;
#endif
  {
    const uint64_t per_block = n_items64 / (4ull * std::max<uint32_t>(1u, n_slots / kRSeg));   // per 256 slots (a 512-slot workgroup refills twice as much: below)
    ds.pool_batch = (uint32_t)std::min<uint64_t>(256, std::max<uint64_t>(64, per_block));
    ds.pool_low = ds.pool_batch >= 128 ? ds.pool_batch / 2 : 24;
    ds.pool_shift = 0;
    while ((1ull << ds.pool_shift) < 2ull * std::max<uint32_t>(1u, n_slots / (uint32_t)RB)) ++ds.pool_shift;   // 2 x the workgroups that draw from the dispenser
  }
  DevParams dp; dp.integrator = rp_in.integrator; dp.spp = rp_in.spp; dp.seed = rp_in.seed; dp.depth = rp_in.depth;
  dp.depth_limit = rp_in.depth_limit; dp.no_direct_emitter = rp_in.no_direct_emitter ? 1 : 0;

  LrStats& S = s.stats;
  double keep_upload = S.upload_ms, keep_bvh = S.bvh_build_ms;
  std::memset(&S, 0, sizeof(S)); S.upload_ms = keep_upload; S.bvh_build_ms = keep_bvh;
  S.path_slots = n_slots; S.pipeline = fused ? 2 : (resident ? 1 : 0);

  HIP_OK(hipMemsetAsync(s.counters.p, 0, (4 + (size_t)n_bands) * sizeof(uint32_t), st));
  HIP_OK(hipMemsetAsync(s.stats_dev.p, 0, ((size_t)kStatShards * kStatStride + 64) * sizeof(unsigned long long), st));
  HIP_OK(hipEventRecord(s.t_begin, st));

  // streaming traversal kernels: at most kStackLdsMax stack entries per lane in LDS (6 workgroups of 25 KB per CU, the
  // occupancy their 80 VGPRs allow), the rest of the worst case in a spill buffer that near-first traversal rarely reaches
  const int stack_in_lds = resident ? s.stack_depth : std::min(s.stack_depth, stack_lds_limit());
  const size_t lds = std::max((size_t)stack_in_lds * kBlock * 4, sort_rays ? sizeof(SortLds) : (size_t)0);   // the sort's histogram borrows the stack
  const void* ktrace = count ? (sort_rays ? (const void*)k_trace<true, true> : (const void*)k_trace<true, false>) : (sort_rays ? (const void*)k_trace<false, true> : (const void*)k_trace<false, false>);
  const void* kshadow = count ? (sort_rays ? (const void*)k_shadow<true, true> : (const void*)k_shadow<true, false>) : (sort_rays ? (const void*)k_shadow<false, true> : (const void*)k_shadow<false, false>);
  if (lds > 156 * 1024) fail(LR_EUNSUPPORTED, "BVH too deep: the per-lane traversal stack does not fit the CU's LDS");
  if (lds > 48 * 1024) {
    HIP_OK(hipFuncSetAttribute(ktrace, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    HIP_OK(hipFuncSetAttribute(kshadow, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  }
  DevScene dsc = s.dev;
  dsc.stack_lds = stack_in_lds; dsc.spill_depth = s.stack_depth - stack_in_lds; dsc.stack_spill = nullptr;
  Launcher L{s, profile};
  if (n_pix > 0) hipLaunchKernelGGL(k_rank_table, dim3(grid_for((const void*)k_rank_table, s.n_cus, 0, n_pix)), dim3(kBlock), 0, st, dsc, ds);
  const DevState ds_all = ds;
  for (uint32_t band = 0; band < n_bands; ++band) {
    const uint32_t r0 = band * band_pix, bn = std::min(band_pix, n_pix - r0);
    ds = ds_all;
    ds.n_pix = bn; ds.n_items = (uint32_t)((uint64_t)bn * n_chunks);
    {
      // sub-bands of 2^17 pixel ranks inside the launch (DevState): the rays in flight stay within one strip of the film.  Config 5:
      // a rank's 1/8 share (524 k pixels spread over the whole film) 271 -> 257 ms, the whole frame -1.2 % (gpurun_out/r05g)
      uint32_t shift = s.dev.n_flat > 0 ? 0u : kSubBandShift;           // flat scenes have no locality to win (14 primitives in the scalar cache) and pay for the longer decode: config 3 +1.1 %
      if (const char* e = lr_knob("LR_SUB_SHIFT")) { int v = std::atoi(e); if (v == 0 || (v >= 6 && v <= 30)) shift = (uint32_t)v; }   // diagnostic
      ds.sub_shift = 0; ds.sub_last_item0 = 0; ds.sub_last_rank0 = 0; ds.sub_last_pix = bn;
      if (shift > 0 && shift < 31 && (bn >> shift) >= 2u) {
        const uint32_t full = (bn >> shift) - 1u;                            // the last sub-band takes the remainder too
        ds.sub_shift = shift; ds.sub_last_rank0 = full << shift; ds.sub_last_pix = bn - ds.sub_last_rank0;
        ds.sub_last_item0 = (uint32_t)(((uint64_t)full * n_chunks) << shift);
      }
    }
    ds.rank_pixel = s.rank_pixel.p + r0; ds.packed = want_packed ? s.packed.p + (size_t)r0 * 3 : nullptr;
    if (fused || resident) ds.next_item = s.counters.p + 4 + band;
    else if (band > 0) HIP_OK(hipMemsetAsync(s.counters.p, 0, 4 * sizeof(uint32_t), st));
    if (ds.n_items > 0 && fused) {
      const uint32_t blocks = n_slots / kBlock, n_waves = blocks * (kBlock / 64);
      // a wave reserves pool_batch work items per trip to the dispenser, one trip ahead of need
      ds.pool_batch = (uint32_t)std::min<uint64_t>(64, std::max<uint64_t>(1, n_items64 / (4ull * n_waves)));
      ds.pool_low = std::max<uint32_t>(1, ds.pool_batch / 2);
      ds.pool_shift = 0;
      while ((1ull << ds.pool_shift) < 2ull * n_waves) ++ds.pool_shift;     // 2 x the waves that draw from the dispenser
      if (s.dev.n_flat == 0) {
        dsc.stack_lds = fused_stack; dsc.spill_depth = s.stack_depth - fused_stack; dsc.stack_spill = nullptr;
        if (dsc.spill_depth > 0) {
          s.stack_spill.ensure((size_t)blocks * dsc.spill_depth * kBlock);
          dsc.stack_spill = s.stack_spill.p;
        }
      }
      const float4* flat_rows = (const float4*)s.flat.p;
      void* args[4] = {&dsc, &ds, &dp, (void*)&flat_rows};                // k_path_tree takes the first three
      L.run(LR_K_PATH, [&] { HIP_OK(hipLaunchKernel(fused_kernel, dim3(blocks), dim3(kBlock), args, fused_lds, st)); });
      S.iterations = 1;
    } else if (ds.n_items > 0 && resident) {
      uint32_t mt_mask = 0;
      for (int k = 0; k < kNumShadeQueues - 1; ++k) if (s.mat_present[k]) mt_mask |= 1u << k;
      if (RB == 512) { ds.pool_batch *= 2; ds.pool_low *= 2; }
      auto launch_resident = [&](auto kernel) {
        HIP_OK(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)resident_lds));
        L.run(LR_K_RESIDENT, [&] { hipLaunchKernelGGL(kernel, dim3(n_slots / RB), dim3(RB), resident_lds, st, dsc, ds, dp, mt_mask, (const float4*)s.flat.p); });
      };
      if (s.dev.n_flat > 0 && mt_mask == 1u) { if (RB == 512) launch_resident(k_resident<true, 1u, 512>); else launch_resident(k_resident<true, 1u, 256>); }   // flat, Lambert only
      else if (s.dev.n_flat > 0) { if (RB == 512) launch_resident(k_resident<true, 31u, 512>); else launch_resident(k_resident<true, 31u, 256>); }
      else launch_resident(k_resident<false, 31u, 256>);
      S.iterations = 1;
    } else if (ds.n_items > 0) {
      // Two slot groups on two streams: the traversal kernels are latency- and divergence-bound, the shade kernels
      // bandwidth-bound, so whenever the two groups are out of phase one's k_trace overlaps the other's k_shade
      // (+10 % on the mesh configs, free-running; chaining the traces with events so that they alternate strictly,
      // or halving the grids, was slower).  The groups share the item dispenser, the chunk sums and the statistics;
      // everything indexed by slot or segment is split.  Small jobs and the counting mode keep one group.
      // k_shade_all is instantiated for the material sets of the BASELINE scenes (Lambert only; Lambert + GGX) and for "anything"
      uint32_t present_bsdf = 0;
      for (int k = 0; k < kNumShadeQueues - 1; ++k) if (s.mat_present[k]) present_bsdf |= 1u << k;
      const int dense_variant = (present_bsdf | 1u) == 1u ? 0 : ((present_bsdf | 9u) == 9u ? 1 : 2);
      const void* kdense = dense_variant == 0 ? (const void*)k_shade_all<1u> : (dense_variant == 1 ? (const void*)k_shade_all<9u> : (const void*)k_shade_all<31u>);
      constexpr int kMaxGroups = 3;
      // two slot groups on two streams; three when an iteration is only trace + shade (no shadow stage): measured +4 % on the
      // 100k-triangle pt scene, -3 % on the pt-direct one
      const bool has_shadow_stage = dp.integrator == LR_INTEGRATOR_PT_DIRECT && s.dev.n_emitters > 0;
      int G = (!count && n_seg >= 64) ? ((dense_shade && !has_shadow_stage && n_seg >= 96) ? 3 : 2) : 1;
      if (const char* e = lr_knob("LR_GROUPS")) { int v = std::atoi(e); if (v == 1 || ((v == 2 || v == 3) && n_seg >= (uint32_t)v)) G = v; }
      for (int g = 1; g < G; ++g) if (!s.gstream[g - 1]) HIP_OK(hipStreamCreateWithFlags(&s.gstream[g - 1], hipStreamNonBlocking));
      if (G > 1 && !s.grp_ev[0]) for (auto& e : s.grp_ev) HIP_OK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
      struct Group {
        DevState ds; DevScene dsc; hipStream_t st; uint32_t n_slots, n_seg, spb; int g_trace, g_shadow, g_gen, g_shade[kNumShadeQueues], g_dense;
      } grp[kMaxGroups];
      uint32_t spill_per_group = 0;
      for (int g = 0; g < G; ++g) {
        Group& q = grp[g];
        const uint32_t base_seg = (uint32_t)((uint64_t)n_seg * g / G);
        q.n_seg = (uint32_t)((uint64_t)n_seg * (g + 1) / G) - base_seg;
        q.n_slots = q.n_seg * kSeg;
        const size_t base = (size_t)base_seg * kSeg;
        q.st = g == 0 ? st : s.gstream[g - 1];
        q.ds = ds;
        q.ds.ray_o += base; q.ds.ray_d += base; q.ds.hit += base; q.ds.thr += base; q.ds.rad += base; q.ds.acc += base;
        q.ds.sh_d += base; q.ds.sh_w += base;
        if (!dense_shade) q.ds.q_shade += (size_t)kNumShadeQueues * base;
        q.ds.c_shade += (size_t)kNumShadeQueues * base_seg;
        q.ds.q_shadow += shadow_lists * base; q.ds.c_shadow += (size_t)(kNumShadeQueues - 1) * base_seg;
        q.ds.pool += base_seg;
        if (q.ds.order) { q.ds.order += base; q.ds.sort_key += base; }
        q.ds.n_retired = s.counters.p + 1 + g;
        q.ds.n_slots = q.n_slots; q.ds.n_seg = q.n_seg;
        q.g_trace = grid_for(ktrace, s.n_cus, lds, q.n_seg * kBlock);
        q.g_shadow = grid_for(kshadow, s.n_cus, lds, q.n_seg * kBlock);
        uint32_t max_group = kMaxGroup;
        if (const char* e = lr_knob("LR_MAXGROUP")) { int v = std::atoi(e); if (v >= 1 && v <= kMaxGroup) max_group = (uint32_t)v; }   // diagnostic: shorter passes / smaller sort windows
        q.spb = std::min<uint32_t>(max_group, (q.n_seg + q.g_trace - 1) / q.g_trace);   // segments per workgroup pass; k_shade and k_shadow walk the same ranges
        q.ds.trace_spb = q.spb;
        q.g_gen = grid_for((const void*)k_generate, s.n_cus, 0, q.n_seg * kBlock);
        const uint32_t n_ranges = (q.n_seg + q.spb - 1) / q.spb;                           // k_shade: one workgroup per trace range
        q.g_shade[0] = grid_for((const void*)k_shade<0>, s.n_cus, shade_lds, n_ranges * kBlock);
        q.g_shade[1] = grid_for((const void*)k_shade<1>, s.n_cus, shade_lds, n_ranges * kBlock);
        q.g_shade[2] = grid_for((const void*)k_shade<2>, s.n_cus, shade_lds, n_ranges * kBlock);
        q.g_shade[3] = grid_for((const void*)k_shade<3>, s.n_cus, shade_lds, n_ranges * kBlock);
        q.g_shade[4] = grid_for((const void*)k_shade<4>, s.n_cus, shade_lds, n_ranges * kBlock);
        q.g_shade[5] = grid_for((const void*)k_shade<5>, s.n_cus, shade_lds, n_ranges * kBlock);
        q.g_dense = grid_for(kdense, s.n_cus, 0, n_ranges * kBlock);
        q.dsc = dsc;
        spill_per_group = std::max<uint32_t>(spill_per_group, (uint32_t)std::max(q.g_trace, q.g_shadow));
      }
      if (dsc.spill_depth > 0) {
        const size_t per = (size_t)spill_per_group * dsc.spill_depth * kBlock;
        s.stack_spill.ensure(per * G);
        for (int g = 0; g < G; ++g) grp[g].dsc.stack_spill = s.stack_spill.p + per * g;
      }
      if (G > 1) { HIP_OK(hipEventRecord(s.grp_ev[0], st)); for (int g = 1; g < G; ++g) HIP_OK(hipStreamWaitEvent(s.gstream[g - 1], s.grp_ev[0], 0)); }   // uploads, memsets, rank table
      for (int g = 0; g < G; ++g) {
        Group& q = grp[g];
        L.run(LR_K_GENERATE, [&] { hipLaunchKernelGGL(k_generate, dim3(q.g_gen), dim3(kBlock), 0, q.st, q.dsc, q.ds, dp); }, q.st);
      }
      const bool nee = dp.integrator == LR_INTEGRATOR_PT_DIRECT && s.dev.n_emitters > 0;
      const int kCheck = 8;
      int batch = 0;
      uint32_t mt_mask = 0;
      for (int k = 0; k < kNumShadeQueues - 1; ++k) if (s.mat_present[k]) mt_mask |= 1u << k;
      bool done = false;
      // hard stop: every iteration advances every live path by one vertex; depth_limit bounds path
      // length statistically, this bounds the loop against a logic error
      const uint64_t max_iter = ((uint64_t)n_items * chunk_spp / n_slots + 64) * 4096ull;
      auto retired_all = [&](const uint32_t* per_group) {
        for (int g = 0; g < G; ++g) if (per_group[g] < grp[g].n_slots) return false;
        return true;
      };
      while (!done) {
        for (int k = 0; k < kCheck; ++k) {
          for (int g = 0; g < G; ++g) {
            Group& q = grp[g];
            auto launch_trace = [&](auto kernel) {
              L.run(LR_K_TRACE, [&] { hipLaunchKernelGGL(kernel, dim3(q.g_trace), dim3(kBlock), lds, q.st, q.dsc, q.ds, (const float4*)s.flat.p, q.spb); }, q.st);
            };
            if (count) { if (sort_rays) launch_trace(k_trace<true, true>); else launch_trace(k_trace<true, false>); }
            else { if (sort_rays) launch_trace(k_trace<false, true>); else launch_trace(k_trace<false, false>); }
            if (dense_shade) {
              L.run(LR_K_SHADE, [&] {
                if (dense_variant == 0) hipLaunchKernelGGL(k_shade_all<1u>, dim3(q.g_dense), dim3(kBlock), 0, q.st, q.dsc, q.ds, dp);
                else if (dense_variant == 1) hipLaunchKernelGGL(k_shade_all<9u>, dim3(q.g_dense), dim3(kBlock), 0, q.st, q.dsc, q.ds, dp);
                else hipLaunchKernelGGL(k_shade_all<31u>, dim3(q.g_dense), dim3(kBlock), 0, q.st, q.dsc, q.ds, dp);
              }, q.st);
            } else {
            if (s.mat_present[0]) L.run(LR_K_SHADE, [&] { hipLaunchKernelGGL(k_shade<0>, dim3(q.g_shade[0]), dim3(kBlock), shade_lds, q.st, q.dsc, q.ds, dp); }, q.st);
            if (s.mat_present[1]) L.run(LR_K_SHADE, [&] { hipLaunchKernelGGL(k_shade<1>, dim3(q.g_shade[1]), dim3(kBlock), shade_lds, q.st, q.dsc, q.ds, dp); }, q.st);
            if (s.mat_present[2]) L.run(LR_K_SHADE, [&] { hipLaunchKernelGGL(k_shade<2>, dim3(q.g_shade[2]), dim3(kBlock), shade_lds, q.st, q.dsc, q.ds, dp); }, q.st);
            if (s.mat_present[3]) L.run(LR_K_SHADE, [&] { hipLaunchKernelGGL(k_shade<3>, dim3(q.g_shade[3]), dim3(kBlock), shade_lds, q.st, q.dsc, q.ds, dp); }, q.st);
            if (s.mat_present[4]) L.run(LR_K_SHADE, [&] { hipLaunchKernelGGL(k_shade<4>, dim3(q.g_shade[4]), dim3(kBlock), shade_lds, q.st, q.dsc, q.ds, dp); }, q.st);
            L.run(LR_K_SHADE, [&] { hipLaunchKernelGGL(k_shade<5>, dim3(q.g_shade[5]), dim3(kBlock), shade_lds, q.st, q.dsc, q.ds, dp); }, q.st);
            }
            if (nee) {
              auto launch_shadow = [&](auto kernel) {
                L.run(LR_K_SHADOW, [&] { hipLaunchKernelGGL(kernel, dim3(q.g_shadow), dim3(kBlock), lds, q.st, q.dsc, q.ds, dense_shade ? 1u : mt_mask, (const float4*)s.flat.p, q.spb); }, q.st);
              };
              if (count) { if (sort_rays) launch_shadow(k_shadow<true, true>); else launch_shadow(k_shadow<true, false>); }
              else { if (sort_rays) launch_shadow(k_shadow<false, true>); else launch_shadow(k_shadow<false, false>); }
            }
          }
          L.iter++; S.iterations++;
        }
        // poll the retired-slot counters one batch behind so the queue never drains
        for (int g = 1; g < G; ++g) { HIP_OK(hipEventRecord(s.grp_ev[g], s.gstream[g - 1])); HIP_OK(hipStreamWaitEvent(st, s.grp_ev[g], 0)); }
        HIP_OK(hipMemcpyAsync(&s.pinned[(batch & 1) * 4], s.counters.p + 1, 3 * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        HIP_OK(hipEventRecord(s.poll_ev[batch & 1], st));
        if (batch > 0) {
          HIP_OK(hipEventSynchronize(s.poll_ev[(batch - 1) & 1]));
          if (retired_all(&s.pinned[((batch - 1) & 1) * 4])) done = true;
        }
        ++batch;
        if (S.iterations > max_iter) fail(LR_EDEVICE, "render loop did not terminate (internal error)");
      }
      for (int g = 1; g < G; ++g) HIP_OK(hipStreamSynchronize(s.gstream[g - 1]));
      HIP_OK(hipStreamSynchronize(st));
      if (!retired_all(&s.pinned[((batch - 1) & 1) * 4])) fail(LR_EDEVICE, "render loop ended with live paths (internal error)");
    }
    if (bn > 0) {
      int g_res = grid_for((const void*)k_resolve, s.n_cus, 0, bn);
      L.run(LR_K_RESOLVE, [&] { hipLaunchKernelGGL(k_resolve, dim3(g_res), dim3(kBlock), 0, st, dsc, ds, dp); });
    }
  }   // bands
  HIP_OK(hipEventRecord(s.t_end, st));
  unsigned long long* hshards = (unsigned long long*)(s.pinned + 8);
  HIP_OK(hipMemcpyAsync(hshards, s.stats_dev.p, (size_t)kStatShards * kStatStride * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
  HIP_OK(hipStreamSynchronize(st));
#ifdef LR_TIMELINE
  if (n_items > 0 && (fused || resident)) {
    // per wave {entry, dispenser seen dry, exit} in 100-MHz ticks -> the ramp and the tail of the one launch (ms relative to the first entry)
    const size_t nw = n_slots / 64;
    std::vector<unsigned long long> tl(3 * nw);
    HIP_OK(hipMemcpy(tl.data(), s.timeline.p, tl.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    std::vector<double> en, dr, ex; unsigned long long t0 = ~0ull;
    for (size_t w = 0; w < nw; ++w) if (tl[3 * w] && tl[3 * w] < t0) t0 = tl[3 * w];
    double busy = 0.0;
    for (size_t w = 0; w < nw; ++w) {
      if (!tl[3 * w] || !tl[3 * w + 2]) continue;
      en.push_back((tl[3 * w] - t0) * 1e-5); ex.push_back((tl[3 * w + 2] - t0) * 1e-5);
      if (tl[3 * w + 1] && (!resident || w % (RB / 64) == 0)) dr.push_back((tl[3 * w + 1] - t0) * 1e-5);
      busy += (tl[3 * w + 2] - tl[3 * w]) * 1e-5;
    }
    std::sort(en.begin(), en.end()); std::sort(dr.begin(), dr.end()); std::sort(ex.begin(), ex.end());
    auto q = [](const std::vector<double>& v, double f) { return v.empty() ? -1.0 : v[std::min(v.size() - 1, (size_t)(f * (v.size() - 1) + 0.5))]; };
    const double span = ex.empty() ? 0.0 : ex.back();
    std::fprintf(stderr, "[LR_TIMELINE] {\"waves\": %zu, \"span_ms\": %.3f, \"entry_ms\": [%.3f, %.3f, %.3f, %.3f], \"dry_ms\": [%.3f, %.3f, %.3f], "
                 "\"exit_ms\": [%.3f, %.3f, %.3f, %.3f, %.3f, %.3f, %.3f], \"wave_time_over_span\": %.4f, \"quantiles\": \"entry 0.5 0.9 0.99 1 | dry 0 0.5 1 | exit 0 0.01 0.1 0.5 0.9 0.99 1\"}\n",
                 en.size(), span, q(en, 0.5), q(en, 0.9), q(en, 0.99), q(en, 1.0), q(dr, 0.0), q(dr, 0.5), q(dr, 1.0),
                 q(ex, 0.0), q(ex, 0.01), q(ex, 0.1), q(ex, 0.5), q(ex, 0.9), q(ex, 0.99), q(ex, 1.0), en.empty() ? 0.0 : busy / (en.size() * span));
    if (resident) {                                                    // per workgroup: iterations after dry, items in the pool at dry, drawn but not yet accounted
      std::vector<double> it, po, tk;
      for (size_t w = 0; w + 3 < nw; w += RB / 64) { if (tl[3 * (w + 1) + 1]) it.push_back((double)tl[3 * (w + 1) + 1] - 1); if (tl[3 * (w + 2) + 1]) po.push_back((double)tl[3 * (w + 2) + 1] - 1); if (tl[3 * (w + 3) + 1]) tk.push_back((double)tl[3 * (w + 3) + 1] - 1); }
      std::sort(it.begin(), it.end()); std::sort(po.begin(), po.end()); std::sort(tk.begin(), tk.end());
      std::fprintf(stderr, "[LR_TIMELINE] resident workgroups (%d slots): iterations after dry [%.0f, %.0f, %.0f, %.0f], pooled items at dry [%.0f, %.0f, %.0f], taken [%.0f, %.0f] (quantiles 0.1 0.5 0.9 1 | 0.1 0.5 1 | 0.5 1)\n",
                   RB, q(it, 0.1), q(it, 0.5), q(it, 0.9), q(it, 1.0), q(po, 0.1), q(po, 0.5), q(po, 1.0), q(tk, 0.5), q(tk, 1.0));
    }
  }
#endif
#ifdef LR_STAMP
  {
    unsigned long long tk[8];
    HIP_OK(hipMemcpy(tk, s.stats_dev.p + (size_t)kStatShards * kStatStride, sizeof(tk), hipMemcpyDeviceToHost));
    double tot = 0; for (int i = 0; i < 6; ++i) tot += (double)tk[i];
    std::fprintf(stderr, "[LR_STAMP] k_resident wave-cycle shares: trace %.1f%% + barrier %.1f%%  shade %.1f%% + barrier %.1f%%  shadow / finish %.1f%% + barrier %.1f%%\n",
                 100 * tk[2] / tot, 100 * tk[1] / tot, 100 * tk[3] / tot, 100 * tk[5] / tot, 100 * tk[4] / tot, 100 * tk[0] / tot);
    std::fprintf(stderr, "[LR_STAMP] of the last barrier, %.2f%% of all wave cycles came after every one of the wave's own 64 slots had completed phase 3 (what per-slot ready flags could return)\n",
                 100 * tk[6] / tot);
  }
#endif
#ifdef LR_DIAG
  {
    TravDiag d;
    HIP_OK(hipMemcpy(&d, s.stats_dev.p + (size_t)kStatShards * kStatStride + 8, sizeof(d), hipMemcpyDeviceToHost));
    if (d.node_steps) std::fprintf(stderr, "[LR_DIAG] k_trace waves: node steps %llu (%.1f lanes), leaf steps %llu (%.1f lanes, %.2f prims/lane, max %.2f/step); "
        "wave cycles: node %.1f%% (%.1f%% waiting for its rows, %.0f cycles per step of %.0f) leaf %.1f%% (%.1f%% waiting for the first rows, %.0f of %.0f) retire %.1f%% fetch %.1f%% other %.1f%%; per ray: %.2f node-lane-steps %.2f leaf-lane-steps\n",
        d.node_steps, (double)d.node_lanes / d.node_steps, d.leaf_steps, (double)d.leaf_lanes / std::max<unsigned long long>(d.leaf_steps, 1), (double)d.leaf_prims / std::max<unsigned long long>(d.leaf_lanes, 1),
        (double)d.leaf_prims_max / std::max<unsigned long long>(d.leaf_steps, 1),
        100.0 * d.cyc_node / d.cyc_total, 100.0 * d.cyc_node_wait / d.cyc_total, (double)d.cyc_node_wait / d.node_steps, (double)d.cyc_node / d.node_steps,
        100.0 * d.cyc_leaf / d.cyc_total, 100.0 * d.cyc_leaf_wait / d.cyc_total, (double)d.cyc_leaf_wait / std::max<unsigned long long>(d.leaf_steps, 1), (double)d.cyc_leaf / std::max<unsigned long long>(d.leaf_steps, 1),
        100.0 * d.cyc_retire / d.cyc_total, 100.0 * d.cyc_fetch / d.cyc_total,
        100.0 * (double)(d.cyc_total - d.cyc_node - d.cyc_leaf - d.cyc_retire - d.cyc_fetch) / d.cyc_total,
        (double)d.node_lanes / std::max<unsigned long long>(d.rays, 1), (double)d.leaf_lanes / std::max<unsigned long long>(d.rays, 1));
  }
#endif
#ifdef LR_DIAG
  {
    PathDiag d;
    HIP_OK(hipMemcpy(&d, s.stats_dev.p + (size_t)kStatShards * kStatStride + 24, sizeof(d), hipMemcpyDeviceToHost));
    auto pc = [&](unsigned long long c) { return 100.0 * (double)c / (double)d.cyc_total; };
    auto per = [](unsigned long long a, unsigned long long b) { return (double)a / (double)std::max<unsigned long long>(b, 1); };
    if (d.walks && !d.node_steps) std::fprintf(stderr, "[LR_DIAG] k_path_flat wave cycles: finish %.1f%% (%.1f lanes ending, %.0f cyc) trace %.1f%% (%.1f lanes live, %.1f with a connection, %.0f cyc) "
        "vertex %.1f%% (%.1f lanes at a hit, %.0f cyc); spare batches %.3f per iteration (%.1f lanes, %.0f cyc, %.1f%% of the cycles, forced %.1f%%)\n",
        pc(d.cyc_finish), per(d.l_finish, d.n_finish), per(d.cyc_finish, d.n_finish), pc(d.cyc_walk), per(d.walk_lanes, d.walks), per(d.l_resolve, d.n_resolve), per(d.cyc_walk, d.walks),
        pc(d.cyc_vertex), per(d.l_vertex, d.n_vertex), per(d.cyc_vertex, d.n_vertex),
        per(d.n_batch, d.n_finish), per(d.l_batch, d.n_batch), per(d.cyc_batch, d.n_batch), pc(d.cyc_batch), 100.0 * per(d.n_forced, d.n_batch));
    if (d.walks && d.node_steps) std::fprintf(stderr, "[LR_DIAG] k_path_tree wave cycles: resolve %.1f%% (%.1f lanes, %.0f cyc) vertex %.1f%% (%.1f lanes, %.0f cyc) finish %.1f%% (%.1f lanes, %.0f cyc) "
        "walk %.1f%% [node %.1f%% (%.1f lanes, %.0f cyc/step) leaf %.1f%% (%.1f lanes, %.0f cyc/step, %.2f primitives per lane, longest %.2f: %.1f lanes per test)] other %.1f%%; per walk: %.1f lanes in, %.1f node steps, %.1f leaf steps; spare batches %.3f per retire point (%.1f lanes, %.0f cyc, %.1f%% of the cycles, forced %.1f%%)\n",
        pc(d.cyc_resolve), per(d.l_resolve, d.n_resolve), per(d.cyc_resolve, d.n_resolve), pc(d.cyc_vertex), per(d.l_vertex, d.n_vertex), per(d.cyc_vertex, d.n_vertex),
        pc(d.cyc_finish), per(d.l_finish, d.n_finish), per(d.cyc_finish, d.n_finish), pc(d.cyc_walk), pc(d.cyc_node), per(d.node_lanes, d.node_steps), per(d.cyc_node, d.node_steps),
        pc(d.cyc_leaf), per(d.leaf_lanes, d.leaf_steps), per(d.cyc_leaf, d.leaf_steps), per(d.leaf_prims, d.leaf_lanes), per(d.leaf_prims_max, d.leaf_steps), per(d.leaf_prims, d.leaf_prims_max),
        pc(d.cyc_total - d.cyc_resolve - d.cyc_vertex - d.cyc_finish - d.cyc_walk), per(d.walk_lanes, d.walks), per(d.node_steps, d.walks), per(d.leaf_steps, d.walks),
        per(d.n_batch, d.n_finish), per(d.l_batch, d.n_batch), per(d.cyc_batch, d.n_batch), pc(d.cyc_batch), 100.0 * per(d.n_forced, d.n_batch));
  }
#endif
  unsigned long long hstats[ST_COUNT] = {0};
  for (int sh = 0; sh < kStatShards; ++sh) for (int k = 0; k < ST_COUNT; ++k) hstats[k] += hshards[sh * kStatStride + k];
  float ms = 0.0f; HIP_OK(hipEventElapsedTime(&ms, s.t_begin, s.t_end));
  S.render_ms = ms;
  S.samples = hstats[ST_SAMPLES]; S.segments = hstats[ST_SEGMENTS]; S.shadow_rays = hstats[ST_SHADOW];
  S.node_visits = hstats[ST_NODE_VISITS]; S.prim_tests = hstats[ST_PRIM_TESTS]; S.sky_fetches = hstats[ST_SKY];
  S.shadow_node_visits = hstats[ST_SHADOW_VISITS]; S.shadow_prim_tests = hstats[ST_SHADOW_TESTS];
  for (int k = 0; k < LR_K_COUNT; ++k) {
    EventPool& p = s.pools[k];
    for (int i = 0; i < p.used; ++i) { float e = 0.0f; HIP_OK(hipEventElapsedTime(&e, p.a[i], p.b[i])); S.kernel_ms[k] += e; }
    S.kernel_timed[k] = (uint64_t)p.used;
  }
}

}  // namespace

#define LR_TRY(...)                                                                     \
  try { __VA_ARGS__; return LR_OK; }                                                    \
  catch (const ApiError& e) { g_err = e.msg; return e.code; }                           \
  catch (const std::bad_alloc&) { g_err = "out of host memory"; return LR_ENOMEM; }     \
  catch (const std::exception& e) { g_err = e.what(); return LR_EINVAL; }           \
  catch (...) { g_err = "unknown error"; return LR_EINVAL; }

extern "C" {

const char* lr_last_error(void) { return g_err.c_str(); }
#ifndef LR_BUILD_ID
#define LR_BUILD_ID "unknown"          /* diagnostic variants (make diag / stamp / timeline) are not built through the id header */
#endif
const char* lr_build_info(void) {
#ifdef LR_DIAG_KNOBS
  return "lumilly_hip gfx950 wave64 -ffp-contract=off abi=2 knobs=on build=" LR_BUILD_ID;
#else
  return "lumilly_hip gfx950 wave64 -ffp-contract=off abi=2 knobs=off build=" LR_BUILD_ID;
#endif
}

int lr_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) { g_err = "hipGetDeviceCount failed"; return 0; }
  return n;
}

int lr_scene_create(int device, const LrSceneDesc* desc, LrScene** out) {
  LrScene* s = nullptr;
  try {
    if (!desc || !out) fail(LR_EINVAL, "null argument");
    int n = 0; HIP_OK(hipGetDeviceCount(&n));
    if (device < 0 || device >= n) fail(LR_EINVAL, "no such device");
    auto t0 = std::chrono::steady_clock::now();
    HIP_OK(hipSetDevice(device));
    s = new LrScene();
    s->device = device;
    hipDeviceProp_t prop; HIP_OK(hipGetDeviceProperties(&prop, device));
    s->n_cus = prop.multiProcessorCount;
    HIP_OK(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
    std::memset(&s->stats, 0, sizeof(s->stats));
    pack_scene(*s, *desc);
    s->stats.upload_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    s->stats.bvh_build_ms = s->bvh_build_ms;
    *out = s;
    return LR_OK;
  } catch (const ApiError& e) { g_err = e.msg; if (s) lr_scene_destroy(s); return e.code; }
  catch (const std::bad_alloc&) { g_err = "out of host memory"; if (s) lr_scene_destroy(s); return LR_ENOMEM; }
  catch (const std::exception& e) { g_err = e.what(); if (s) lr_scene_destroy(s); return LR_EINVAL; }      // e.g. std::length_error from a vector sized by a corrupt count
  catch (...) { g_err = "unknown error"; if (s) lr_scene_destroy(s); return LR_EINVAL; }                   // nothing crosses the extern "C" boundary
}

int lr_scene_destroy(LrScene* s) {
  if (!s) return LR_OK;
  (void)hipSetDevice(s->device);
  if (s->stream) (void)hipStreamSynchronize(s->stream);
  for (auto& g : s->gstream) if (g) (void)hipStreamSynchronize(g);
  s->nodes.release(); s->prims.release(); s->flat.release(); s->shade.release(); s->emit.release(); s->texels.release(); s->texels_rgbe.release(); s->prim_qid.release();
  s->ray_o.release(); s->ray_d.release(); s->thr.release(); s->rad.release(); s->acc.release(); s->sh_d.release(); s->sh_w.release();
  s->partial.release(); s->hit.release(); s->q_shade.release(); s->c_shade.release(); s->q_shadow.release(); s->c_shadow.release(); s->pool.release(); s->counters.release(); s->tile_prefix.release(); s->tiles.release(); s->rank_pixel.release(); s->stack_spill.release(); s->chunk_start.release();
  s->stats_dev.release(); s->film.release(); s->packed.release(); s->sort_key.release(); s->order.release();
  if (s->pinned) (void)hipHostFree(s->pinned);
  if (s->host_film) (void)hipHostFree(s->host_film);
  for (auto e : s->poll_ev) if (e) (void)hipEventDestroy(e);
  if (s->t_begin) (void)hipEventDestroy(s->t_begin);
  if (s->t_end) (void)hipEventDestroy(s->t_end);
  for (auto& p : s->pools) p.destroy();
  if (s->stream) (void)hipStreamDestroy(s->stream);
  for (auto& g : s->gstream) if (g) (void)hipStreamDestroy(g);
  for (auto& e : s->grp_ev) if (e) (void)hipEventDestroy(e);
  delete s;
  return LR_OK;
}

int lr_render_device(LrScene* s, const LrRenderParams* params, const LrTile* tiles, int n_tiles, void** film_dev) {
  LR_TRY({
    if (!s || !params) fail(LR_EINVAL, "null argument");
    render_impl(*s, *params, tiles, n_tiles);
    if (film_dev) *film_dev = s->film.p;
  })
}

int lr_render(LrScene* s, const LrRenderParams* params, const LrTile* tiles, int n_tiles, float* rgb_out, size_t row_stride_floats) {
  LR_TRY({
    if (!s || !params || !rgb_out) fail(LR_EINVAL, "null argument");
    if (row_stride_floats < (size_t)s->film_w * 3) fail(LR_EINVAL, "row stride smaller than one film row");
    render_impl(*s, *params, tiles, n_tiles, true);
    // read back only the pixels this call rendered (1/world of the film for a rank of a multi-GPU job): k_resolve left
    // them packed in pixel-rank order = tile list order, row-major inside a tile
    size_t n_val = 0;
    for (int i = 0; i < n_tiles; ++i) if (tiles[i].w > 0 && tiles[i].h > 0) n_val += (size_t)tiles[i].w * tiles[i].h * 3;
    if (s->host_film_cap < n_val) {                                 // pinned: the device-to-host copy runs at link speed, not through a bounce buffer
      if (s->host_film) (void)hipHostFree(s->host_film);
      s->host_film = nullptr; s->host_film_cap = 0;
      HIP_OK(hipHostMalloc((void**)&s->host_film, n_val * sizeof(float)));
      s->host_film_cap = n_val;
    }
    if (n_val > 0) {
      HIP_OK(hipMemcpyAsync(s->host_film, s->packed.p, n_val * sizeof(float), hipMemcpyDeviceToHost, s->stream));
      HIP_OK(hipStreamSynchronize(s->stream));
    }
    const float* src = s->host_film;
    for (int i = 0; i < n_tiles; ++i) {                            // only tile pixels are written (Img::set per job, main.rs:129-132)
      const LrTile& t = tiles[i];
      if (t.w <= 0 || t.h <= 0) continue;
      for (int y = t.y0; y < t.y0 + t.h; ++y, src += (size_t)t.w * 3)
        std::memcpy(rgb_out + (size_t)y * row_stride_floats + (size_t)t.x0 * 3, src, (size_t)t.w * 3 * sizeof(float));
    }
  })
}

int lr_film_quantize(LrScene* s, int mode, float gamma, uint8_t* out, size_t row_stride_bytes) {
  LR_TRY({
    if (!s || !out || (mode != LR_QUANT_RGB8 && mode != LR_QUANT_RGBE)) fail(LR_EINVAL, "bad argument");
    if (!s->film.p || s->film_w <= 0) fail(LR_EINVAL, "no film: render first");
    if (mode == LR_QUANT_RGB8 && !(gamma > 0.0f)) fail(LR_EINVAL, "gamma must be positive");
    const int W = s->film_w, H = s->film_h, bpp = mode == LR_QUANT_RGB8 ? 3 : 4;
    if (row_stride_bytes < (size_t)W * bpp) fail(LR_EINVAL, "row stride smaller than one row");
    HIP_OK(hipSetDevice(s->device));
    DevBuf<uint8_t> q; q.ensure((size_t)W * H * bpp);
    uint32_t n_pix = (uint32_t)((size_t)W * H);
    int grid = (int)std::min<size_t>(((size_t)n_pix + kBlock - 1) / kBlock, (size_t)s->n_cus * 8);
    hipLaunchKernelGGL(k_quantize, dim3(grid), dim3(kBlock), 0, s->stream, s->film.p, q.p, n_pix, mode, 1.0f / gamma);
    HIP_OK(hipGetLastError());
    std::vector<uint8_t> host((size_t)W * H * bpp);
    HIP_OK(hipMemcpyAsync(host.data(), q.p, host.size(), hipMemcpyDeviceToHost, s->stream));
    HIP_OK(hipStreamSynchronize(s->stream));
    for (int y = 0; y < H; ++y) std::memcpy(out + (size_t)y * row_stride_bytes, host.data() + (size_t)y * W * bpp, (size_t)W * bpp);
  })
}

int lr_get_stats(LrScene* s, LrStats* out) {
  if (!s || !out) { g_err = "null argument"; return LR_EINVAL; }
  *out = s->stats;
  return LR_OK;
}

// ---- diagnostics: run the device math / RNG / traversal on caller data (parity tests) -------------
int lr_selftest_math(int device, int fn, const float* a, const float* b, float* out, int n) {
  LR_TRY({
    if (!a || !out || n < 0) fail(LR_EINVAL, "bad argument");
    HIP_OK(hipSetDevice(device));
    DevBuf<float> da, db, dout;
    da.ensure(n); db.ensure(n); dout.ensure(n);
    HIP_OK(hipMemcpy(da.p, a, (size_t)n * 4, hipMemcpyHostToDevice));
    if (b) HIP_OK(hipMemcpy(db.p, b, (size_t)n * 4, hipMemcpyHostToDevice));
    if (n > 0) hipLaunchKernelGGL(k_selftest_math, dim3((n + 255) / 256), dim3(256), 0, 0, fn, da.p, b ? db.p : (const float*)nullptr, dout.p, n);
    HIP_OK(hipGetLastError()); HIP_OK(hipDeviceSynchronize());
    HIP_OK(hipMemcpy(out, dout.p, (size_t)n * 4, hipMemcpyDeviceToHost));
  })
}
int lr_selftest_rcp(int device, uint32_t lo_exp, uint32_t hi_exp, uint64_t* out4) {
  LR_TRY({
    if (!out4) fail(LR_EINVAL, "bad argument");
    HIP_OK(hipSetDevice(device));
    DevBuf<unsigned long long> d; d.ensure(4);
    HIP_OK(hipMemset(d.p, 0, 32));
    hipLaunchKernelGGL(k_selftest_rcp, dim3(256 * 16), dim3(256), 0, 0, lo_exp, hi_exp, d.p);
    HIP_OK(hipGetLastError()); HIP_OK(hipDeviceSynchronize());
    HIP_OK(hipMemcpy(out4, d.p, 32, hipMemcpyDeviceToHost));
  })
}
int lr_selftest_rng(int device, uint32_t seed, const uint32_t* pixel, const uint32_t* sample, const uint32_t* block, float* out4, int n) {
  LR_TRY({
    if (!pixel || !sample || !block || !out4 || n < 0) fail(LR_EINVAL, "bad argument");
    HIP_OK(hipSetDevice(device));
    DevBuf<uint32_t> dp, dsm, dbk; DevBuf<float> dout;
    dp.ensure(n); dsm.ensure(n); dbk.ensure(n); dout.ensure((size_t)n * 4);
    HIP_OK(hipMemcpy(dp.p, pixel, (size_t)n * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(dsm.p, sample, (size_t)n * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(dbk.p, block, (size_t)n * 4, hipMemcpyHostToDevice));
    if (n > 0) hipLaunchKernelGGL(k_selftest_rng, dim3((n + 255) / 256), dim3(256), 0, 0, seed, dp.p, dsm.p, dbk.p, dout.p, n);
    HIP_OK(hipGetLastError()); HIP_OK(hipDeviceSynchronize());
    HIP_OK(hipMemcpy(out4, dout.p, (size_t)n * 16, hipMemcpyDeviceToHost));
  })
}
int lr_selftest_intersect(LrScene* s, int n, const float* origins, const float* dirs, int32_t* prim_out, float* t_out) {
  LR_TRY({
    if (!s || !origins || !dirs || !prim_out || !t_out || n < 0) fail(LR_EINVAL, "bad argument");
    HIP_OK(hipSetDevice(s->device));
    DevBuf<float> dor, ddr, dt; DevBuf<int> dpr;
    dor.ensure((size_t)n * 3); ddr.ensure((size_t)n * 3); dt.ensure(n); dpr.ensure(n);
    HIP_OK(hipMemcpy(dor.p, origins, (size_t)n * 12, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(ddr.p, dirs, (size_t)n * 12, hipMemcpyHostToDevice));
    const int in_lds = std::min(s->stack_depth, stack_lds_limit());
    size_t lds = (size_t)in_lds * kBlock * 4;
    DevScene dsc = s->dev;
    dsc.stack_lds = in_lds; dsc.spill_depth = s->stack_depth - in_lds; dsc.stack_spill = nullptr;
    if (dsc.spill_depth > 0) { s->stack_spill.ensure((size_t)((n + kBlock - 1) / kBlock) * dsc.spill_depth * kBlock); dsc.stack_spill = s->stack_spill.p; }
    if (lds > 48 * 1024) HIP_OK(hipFuncSetAttribute((const void*)k_selftest_intersect, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    if (n > 0) hipLaunchKernelGGL(k_selftest_intersect, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), lds, s->stream, dsc, (const float4*)s->flat.p, s->stack_depth, dor.p, ddr.p, dpr.p, dt.p, n);
    HIP_OK(hipGetLastError()); HIP_OK(hipStreamSynchronize(s->stream));
    HIP_OK(hipMemcpy(prim_out, dpr.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    for (int i = 0; i < n; ++i) if (prim_out[i] >= 0 && (size_t)prim_out[i] < s->user_id.size()) prim_out[i] = s->user_id[(size_t)prim_out[i]];   // device id -> the caller's index
    HIP_OK(hipMemcpy(t_out, dt.p, (size_t)n * 4, hipMemcpyDeviceToHost));
  })
}

static int selftest_brute(LrScene* s, bool own_box, int n, const float* origins, const float* dirs, int32_t* prim_out, float* t_out) {
  LR_TRY({
    if (!s || !origins || !dirs || !prim_out || !t_out || n < 0) fail(LR_EINVAL, "bad argument");
    HIP_OK(hipSetDevice(s->device));
    DevBuf<float> dor, ddr, dt; DevBuf<int> dpr;
    dor.ensure((size_t)n * 3); ddr.ensure((size_t)n * 3); dt.ensure(n); dpr.ensure(n);
    HIP_OK(hipMemcpy(dor.p, origins, (size_t)n * 12, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(ddr.p, dirs, (size_t)n * 12, hipMemcpyHostToDevice));
    if (n > 0) hipLaunchKernelGGL(k_selftest_brute, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, s->stream, (const float4*)s->prims.p, own_box ? (const float4*)s->dev.pbox : (const float4*)nullptr, s->n_prims, dor.p, ddr.p, dpr.p, dt.p, n);
    HIP_OK(hipGetLastError()); HIP_OK(hipStreamSynchronize(s->stream));
    HIP_OK(hipMemcpy(prim_out, dpr.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    for (int i = 0; i < n; ++i) if (prim_out[i] >= 0 && (size_t)prim_out[i] < s->user_id.size()) prim_out[i] = s->user_id[(size_t)prim_out[i]];   // device id -> the caller's index
    HIP_OK(hipMemcpy(t_out, dt.p, (size_t)n * 4, hipMemcpyDeviceToHost));
  })
}
int lr_selftest_brute(LrScene* s, int n, const float* origins, const float* dirs, int32_t* prim_out, float* t_out) { return selftest_brute(s, false, n, origins, dirs, prim_out, t_out); }
int lr_selftest_brute_own_box(LrScene* s, int n, const float* origins, const float* dirs, int32_t* prim_out, float* t_out) { return selftest_brute(s, true, n, origins, dirs, prim_out, t_out); }
int lr_selftest_sky(LrScene* s, int n, const float* dirs, float* rgb_out) {
  LR_TRY({
    if (!s || !dirs || !rgb_out || n < 0) fail(LR_EINVAL, "bad argument");
    HIP_OK(hipSetDevice(s->device));
    DevBuf<float> dd, dc;
    dd.ensure((size_t)n * 3); dc.ensure((size_t)n * 3);
    HIP_OK(hipMemcpy(dd.p, dirs, (size_t)n * 12, hipMemcpyHostToDevice));
    if (n > 0) hipLaunchKernelGGL(k_selftest_sky, dim3((n + 255) / 256), dim3(256), 0, s->stream, s->dev, dd.p, dc.p, n);
    HIP_OK(hipGetLastError()); HIP_OK(hipStreamSynchronize(s->stream));
    HIP_OK(hipMemcpy(rgb_out, dc.p, (size_t)n * 12, hipMemcpyDeviceToHost));
  })
}
int lr_selftest_material(int device, const LrMaterial* m, int n, const float* in13, float* out10) {
  LR_TRY({
    if (!m || !in13 || !out10 || n < 0) fail(LR_EINVAL, "bad argument");
    if (m->type < 0 || m->type > LR_MAT_IDEAL_REFRACTION) fail(LR_EINVAL, "unknown material type");
    HIP_OK(hipSetDevice(device));
    DevBuf<float> di, dout;
    di.ensure((size_t)n * 13); dout.ensure((size_t)n * 10);
    HIP_OK(hipMemcpy(di.p, in13, (size_t)n * 13 * 4, hipMemcpyHostToDevice));
    // the three material rows exactly as pack_scene lays them out
    const float w = std::fmax(std::fmax(m->color[0], m->color[1]), m->color[2]);
    const bool emits = m->type == LR_MAT_LAMBERT;
    const float4 m0 = make_float4(m->color[0], m->color[1], m->color[2], __builtin_bit_cast(float, (uint32_t)m->type));
    const float4 m1 = make_float4(emits ? m->emission[0] : 0.0f, emits ? m->emission[1] : 0.0f, emits ? m->emission[2] : 0.0f, w);
    const float4 m2 = make_float4(m->param[0], m->param[1], m->param[2], 0.0f);
    if (n > 0) hipLaunchKernelGGL(k_selftest_material, dim3((n + 255) / 256), dim3(256), 0, 0, m0, m1, m2, di.p, dout.p, n);
    HIP_OK(hipGetLastError()); HIP_OK(hipDeviceSynchronize());
    HIP_OK(hipMemcpy(out10, dout.p, (size_t)n * 10 * 4, hipMemcpyDeviceToHost));
  })
}
int lr_selftest_camera(LrScene* s, int n, const int32_t* xy, const float* xi4, float* out8) {
  LR_TRY({
    if (!s || !xy || !xi4 || !out8 || n < 0) fail(LR_EINVAL, "bad argument");
    HIP_OK(hipSetDevice(s->device));
    DevBuf<int> dxy; DevBuf<float> dxi, dout;
    dxy.ensure((size_t)n * 2); dxi.ensure((size_t)n * 4); dout.ensure((size_t)n * 8);
    HIP_OK(hipMemcpy(dxy.p, xy, (size_t)n * 8, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(dxi.p, xi4, (size_t)n * 16, hipMemcpyHostToDevice));
    if (n > 0) hipLaunchKernelGGL(k_selftest_camera, dim3((n + 255) / 256), dim3(256), 0, s->stream, s->dev, dxy.p, dxi.p, dout.p, n);
    HIP_OK(hipGetLastError()); HIP_OK(hipStreamSynchronize(s->stream));
    HIP_OK(hipMemcpy(out8, dout.p, (size_t)n * 32, hipMemcpyDeviceToHost));
  })
}
int lr_selftest_emission_sample(LrScene* s, int n, const float* xi4, float* out4) {
  LR_TRY({
    if (!s || !xi4 || !out4 || n < 0) fail(LR_EINVAL, "bad argument");
    if (s->dev.n_emitters <= 0) fail(LR_EINVAL, "scene has no emitters");
    HIP_OK(hipSetDevice(s->device));
    DevBuf<float> dxi, dout;
    dxi.ensure((size_t)n * 4); dout.ensure((size_t)n * 4);
    HIP_OK(hipMemcpy(dxi.p, xi4, (size_t)n * 16, hipMemcpyHostToDevice));
    DevState ds; std::memset(&ds, 0, sizeof(ds));
    if (n > 0) hipLaunchKernelGGL(k_selftest_emission_sample, dim3((n + 255) / 256), dim3(256), 0, s->stream, s->dev, ds, dxi.p, dout.p, n);
    HIP_OK(hipGetLastError()); HIP_OK(hipStreamSynchronize(s->stream));
    HIP_OK(hipMemcpy(out4, dout.p, (size_t)n * 16, hipMemcpyDeviceToHost));
  })
}
int lr_selftest_tree_info(LrScene* s, int32_t* out4) {
  if (!s || !out4) return LR_EINVAL;
  for (int k = 0; k < 4; ++k) out4[k] = s->tree_info[k];
  return LR_OK;
}
int lr_selftest_sky_texel_bytes(LrScene* s) {
  if (!s) return LR_EINVAL;
  return s->dev.sky_type != LR_SKY_IBL ? 0 : (s->dev.texels_rgbe ? 4 : 16);
}
int lr_selftest_emitter_pick(LrScene* s, int n, const float* xi, int32_t* k_out) {
  LR_TRY({
    if (!s || !xi || !k_out || n < 0) fail(LR_EINVAL, "bad argument");
    if (s->dev.n_emitters <= 0) fail(LR_EINVAL, "scene has no emitters");
    HIP_OK(hipSetDevice(s->device));
    DevBuf<float> dx; DevBuf<int> dk;
    dx.ensure(n); dk.ensure(n);
    HIP_OK(hipMemcpy(dx.p, xi, (size_t)n * 4, hipMemcpyHostToDevice));
    if (n > 0) hipLaunchKernelGGL(k_selftest_emitter_pick, dim3((n + 255) / 256), dim3(256), 0, s->stream, s->dev, dx.p, dk.p, n);
    HIP_OK(hipGetLastError()); HIP_OK(hipStreamSynchronize(s->stream));
    HIP_OK(hipMemcpy(k_out, dk.p, (size_t)n * 4, hipMemcpyDeviceToHost));
  })
}

}  // extern "C"
