// lr_kernels.h -- device functions of the path (primitive tests, 4-wide node step, BSDFs, cameras, sky, emitter sampling, the
// vertex of scene.rs:153-193) and the kernels of two of the three pipelines that run them; the third, default one -- the
// FUSED pipeline, one persistent launch in which a lane carries its path in registers -- is lr_path.h.
//
// STREAMING pipeline (state in HBM, one launch per stage and iteration; since round 3 the counting / reference pipeline):
// one iteration of the render loop is   trace -> shade -> shadow.   The shade stage is k_shade_all (one launch over the
// slots themselves, every class of vertex under its lane mask) or, with LR_DENSE=0, one k_shade<bsdf> launch per BSDF
// present plus the miss/sky launch over the lists k_trace then writes.  Paths never leave their slot: a path that ends regenerates
// the next camera sample of its work item in place, so every slot is live until the work-item
// dispenser runs dry ("persistent" path slots; workgroups are grid-strided over them).
// RESIDENT pipeline (k_resident: state in LDS, the stages as phases of one launch; flat scenes with several BSDFs, see its
// header further down).
//
//   k_generate   camera.rs:64-115 / :411-476 / :168-188   first camera sample of every slot
//   k_trace      bvh.rs:130-141 + aabb.rs:74-92 + triangle.rs:69-100 + sphere.rs:42-63
//                closest hit; LDS-staged per-lane traversal stack; epilogue compacts slot ids into
//                one queue per BSDF with __ballot / popcount prefix sums
//   k_shade_all  the same vertex code as k_shade<M> below for every slot of a range in slot order
//   k_shade<M>   scene.rs:153-193 (emission, Russian roulette, direct-light sample, BSDF sample),
//                material/*.rs for M, sky.rs for the miss queue, main.rs:92-121 for the per-sample
//                accumulation; pushes NEE-eligible slots to the shadow queue
//   k_shadow     scene.rs:127-147: visibility by closest hit within EPS of the sampled point,
//                light-side cosine and emission of the surface actually hit
//   k_resolve    main.rs:104,121 + img.rs:25-27: chunk sums -> pixel mean -> film
#pragma once
#include <type_traits>
#include "lr_math.h"
#include "lr_device.h"

namespace lr {

// ------------------------------------------------------------------------------------------
// wave / workgroup helpers.  Queue heads, item pools and statistics are aggregated per 512-slot
// segment in LDS: one global atomic per wave on a shared word costs ~12 ns and serialises (the first
// version of these kernels spent >90 % of its time there), an LDS atomic does not.
// ------------------------------------------------------------------------------------------
// A loop-invariant value, made opaque at the point of use: whatever is computed from it (an integer reciprocal, a float
// conversion, an LDS address) is computed THERE instead of once at kernel entry.  In the persistent kernels everything hoisted
// to the entry lives across the whole loop, and what does not fit the registers is reloaded from scratch -- a round trip
// through the vector-memory path (3000+ cycles under a tree walk's load) where ten instructions would have done.
template <class T> LR_DEV T fresh_s(T v) { asm volatile("" : "+s"(v)); return v; }     // wave-uniform value
LR_DEV uint32_t fresh_v(uint32_t v) { asm volatile("" : "+v"(v)); return v; }
// the lane id: two instructions wherever it is needed (the mask is opaque, so it is not an entry-block value either)
LR_DEV uint32_t lane_id() { const uint32_t m = fresh_s(~0u); return __builtin_amdgcn_mbcnt_hi(m, __builtin_amdgcn_mbcnt_lo(m, 0u)); }
// threadIdx.x from the wave's first thread id (a scalar, kept by the caller) -- threadIdx.x itself is an entry value too
LR_DEV uint32_t tid_of(uint32_t wave_base) { return wave_base + lane_id(); }
LR_DEV uint32_t uniform(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
// make timeline (-DLR_TIMELINE): when every wave of the one-launch kernels entered, first saw the item dispenser dry, and left
// (lr_render prints the distribution: the ramp and the tail of a render, i.e. its fixed cost per call)
#ifdef LR_TIMELINE
#define LR_TL(ST, SLOT) { if (lane_id() == 0) (ST).timeline[3 * ((size_t)blockIdx.x * (blockDim.x >> 6) + (uniform(threadIdx.x) >> 6)) + (SLOT)] = __builtin_amdgcn_s_memrealtime(); }
#else
#define LR_TL(ST, SLOT)
#endif
// count `mask`'s lanes into a workgroup statistic (converged wave): one LDS atomic without return, no register kept
LR_DEV void stat_count(uint32_t* lds_stat, uint64_t mask) {
  if (mask != 0 && lane_id() == 0) atomicAdd(lds_stat, (uint32_t)__builtin_popcountll(mask));
}
LR_DEV uint32_t rank_in_mask(uint64_t mask) {
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}
// Reserve popcount(pred) consecutive entries with ONE atomic per wave (on an LDS or global word);
// returns this lane's index.  Must be reached by the whole (converged) wave.
LR_DEV uint32_t wave_reserve(uint32_t* counter, bool pred) {
  uint64_t mask = __ballot(pred);
  if (mask == 0) return 0;
  uint32_t leader = (uint32_t)__builtin_ctzll(mask);
  uint32_t base = 0;
  if (lane_id() == leader) base = atomicAdd(counter, (uint32_t)__builtin_popcountll(mask));
  base = __shfl(base, (int)leader, 64);
  return base + rank_in_mask(mask);
}
// per-lane counter -> workgroup total in LDS (flushed to a sharded global counter at kernel end)
LR_DEV void stat_accumulate(uint32_t* lds_stat, uint32_t v) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  if (lane_id() == 0 && v) atomicAdd(lds_stat, v);
}
LR_DEV void stat_flush(unsigned long long* stats, const uint32_t* lds_stat) {
  // call after __syncthreads(); 64 shards of kStatStride words spread the atomics over L2 channels
  if (threadIdx.x < ST_COUNT && lds_stat[threadIdx.x])
    atomicAdd(stats + (size_t)(blockIdx.x & (kStatShards - 1)) * kStatStride + threadIdx.x, (unsigned long long)lds_stat[threadIdx.x]);
}

// ------------------------------------------------------------------------------------------
// primitive tests -- exact restatements (no contraction in this TU)
// ------------------------------------------------------------------------------------------
// triangle.rs:69-100 with e1, e2 precomputed
LR_DEV bool tri_test(V3 p0, V3 e1, V3 e2, V3 o, V3 d, float* t_out) {
  V3 pv = cross(d, e2);
  float det = dot(e1, pv);
  if (__builtin_fabsf(det) < kEps) return false;
  float invdet = rcp_exact_mid(det);                  // == 1.0f / det, see lr_math.h
  V3 tv = o - p0;
  float u = dot(tv, pv) * invdet;
  if (u < 0.0f || u > 1.0f) return false;
  V3 qv = cross(tv, e1);
  float v = dot(d, qv) * invdet;
  if (v < 0.0f || u + v > 1.0f) return false;
  float t = dot(e2, qv) * invdet;
  if (t < kEps) return false;
  *t_out = t;
  return true;
}
// sphere.rs:42-55
LR_DEV bool sphere_test(V3 c, float r2, V3 o, V3 d, float* t_out) {
  V3 co = o - c;
  float cod = dot(co, d);
  float det = cod * cod - sqr_norm(co) + r2;
  if (det <= 0.0f) return false;
  float sq = __builtin_sqrtf(det);
  float t1 = -cod - sq;
  float t2 = -cod + sq;
  if (t1 < kEps && t2 < kEps) return false;
  *t_out = t1 > kEps ? t1 : t2;
  return true;
}

// ------------------------------------------------------------------------------------------
// The leaf's OWN box (bvh.rs:20-25): Leaf::may_intersect pushes a primitive onto the candidate list only if
// aabb.rs:74-92 passes on the primitive's own exact box -- a slab test in plain f32, seeded with [-1e5, 1e5]
// (constant.rs:3).  It DECIDES: a ray that triangle.rs:69-100 accepts within a few ulp of an edge lying on a face of a
// flat box (every axis-aligned wall) is dropped by the reference, and so is any hit beyond 1e5.  The boxes of the inner
// nodes never reject what the leaf's box accepts (rounding is monotone in the planes), so the reference's closest hit is
//     min t over { p : own_box(p) passes  and  own_test(p) accepts },
// whatever the tree.  The device keeps its fast search over ALL primitives (conservative boxes that only prune) and then
// SETTLES the one primitive that decides the query -- the closest hit, or a connection's occluder:
//   own_box_surely   one approximate slab test with a margin on every comparison: true = the literal test certainly
//                    passes, so the winner of the unfiltered search is the winner of the filtered one (every other
//                    candidate is no nearer);
//   otherwise        (~1e-3 of the hits on unit-sized triangles, ~1e-5 on walls: the hit lies within ~1e-6 relative of a
//                    face of its own box as seen along the ray, or a direction component is ~0) the query is repeated
//                    LITERALLY: every primitive test followed by own_box_exact, the reference's operations in the
//                    reference's order (retrace_flat / retrace_tree: cold code; k_path_tree: the lane walks again among
//                    the others in literal mode, lr_path.h PathCtl::lit).
// pbox: rows 4 and 5 of the primitive's 128-B record (lr_device.h), {min.xyz, r^2 of a sphere} {max.xyz, -} as
// triangle.rs:102-118 / sphere.rs:31-38 compute them.
// ------------------------------------------------------------------------------------------
LR_DEV bool own_box_exact(float4 lo, float4 hi, V3 o, V3 d) {                   // aabb.rs:74-92, line by line
  float mn = -kInf, mx = kInf;
#define LR_AXIS(C)                                                                            \
  {                                                                                           \
    float inv_d = 1.0f / d.C;                                                                 \
    float t1 = (lo.C - o.C) * inv_d;                                                          \
    float t2 = (hi.C - o.C) * inv_d;                                                          \
    float t_min = t1, t_max = t2;                                                             \
    if (t1 > t2) { t_min = t2; t_max = t1; }                                                  \
    if (mn < t_min) mn = t_min;                                                               \
    if (mx > t_max) mx = t_max;                                                               \
    if (mn > mx) return false;                                                                \
  }
  LR_AXIS(x) LR_AXIS(y) LR_AXIS(z)
#undef LR_AXIS
  return true;
}
// true = own_box_exact(lo, hi, o, d) is certainly true.  (ix, iy, iz) = v_rcp_f32 of d's components: 1/d to 1 ulp, and +-inf for a
// zero or denormal component -- the slab parameters of that axis are then +-inf (or NaN when o lies in one of its planes) and the
// margin arithmetic below turns them into NaN: "undecided", which is where the literal test's own inf / NaN rules must speak.
// (The traversal's clamped reciprocals do NOT qualify: their finite products would be compared.)  The literal test's slab
// parameters are fl(fl(plane - o) * fl(1/d)); the ones below share the exact difference and differ by < 2.5e-7 relative, the
// margin is 2^-20 relative + 1e-30 absolute on each side of every comparison the literal test makes between DIFFERENT axes (its
// own axis always passes by construction: t_min <= t_max after the swap).  Every comparison is written so that a NaN gives
// "undecided".
LR_DEV bool own_box_surely(float4 lo, float4 hi, V3 o, V3 d, float ix, float iy, float iz) {
  const float g = 9.5367431640625e-7f;                                          // 2^-20
  float ax = (lo.x - o.x) * ix, bx = (hi.x - o.x) * ix;
  float ay = (lo.y - o.y) * iy, by = (hi.y - o.y) * iy;
  float az = (lo.z - o.z) * iz, bz = (hi.z - o.z) * iz;
  float nx = __builtin_fminf(ax, bx), fx = __builtin_fmaxf(ax, bx);
  float ny = __builtin_fminf(ay, by), fy = __builtin_fmaxf(ay, by);
  float nz = __builtin_fminf(az, bz), fz = __builtin_fmaxf(az, bz);
  {
#pragma clang fp contract(fast)
    // entry parameters pushed up, exit parameters pulled down by the margin
    nx = __builtin_fmaf(__builtin_fabsf(nx), g, nx); ny = __builtin_fmaf(__builtin_fabsf(ny), g, ny); nz = __builtin_fmaf(__builtin_fabsf(nz), g, nz);
    fx = __builtin_fmaf(__builtin_fabsf(fx), -g, fx) - 1e-30f; fy = __builtin_fmaf(__builtin_fabsf(fy), -g, fy) - 1e-30f; fz = __builtin_fmaf(__builtin_fabsf(fz), -g, fz) - 1e-30f;
  }
  const float hi5 = kInf * (1.0f - 2.0f * g);
  const bool ok = bool(nx <= __builtin_fminf(__builtin_fminf(fy, fz), hi5)) & bool(ny <= __builtin_fminf(__builtin_fminf(fx, fz), hi5)) &
                  bool(nz <= __builtin_fminf(__builtin_fminf(fx, fy), hi5)) & bool(__builtin_fminf(__builtin_fminf(fx, fy), fz) >= -hi5);
  return ok;
}

#ifndef LR_NO_SETTLE
#define LR_NO_SETTLE 0                 // measurement only (tools/build_variant.sh): skip the own-box stage = the closest hit over ALL primitives of rounds 1-5
#endif
// The winner `prim` of an unfiltered search is NOT certainly a candidate (lo, hi = its own box rows; prim < 0: nothing to settle):
// the approximate test, every lane, ~40 instructions, no branch.  Where it cannot tell -- the hit lies within ~1e-6 relative of a
// face of its own box as seen along the ray: ~1e-3 of the hits on unit-sized triangles, ~1e-5 on walls -- the query is repeated
// LITERALLY (every primitive test followed by own_box_exact); that costs one more walk for one ray in a thousand and keeps the
// three IEEE divisions of the literal test out of the hot code.
LR_DEV bool own_box_unsure(float4 lo, float4 hi, int prim, V3 o, V3 d) {
#if LR_NO_SETTLE
  return false;
#endif
  return bool(prim >= 0) & !own_box_surely(lo, hi, o, d, __builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y), __builtin_amdgcn_rcpf(d.z));
}

// ------------------------------------------------------------------------------------------
// BVH traversal.  Closest hit = min t over the primitives whose own box passes (above) and whose own test accepts,
// ties to the lowest primitive id (order independent).  The tree's boxes are padded by the host builder, so ITS slab
// test may use fused / approximate arithmetic: it only prunes, never decides.
//   SHADOW: accept only hits with t - dist <= EPS; stop at the first hit with t - dist < -EPS
//   (then the closest hit is at least that near and scene.rs:129 rejects the connection); the occluder's id and
//   distance are kept, because its own box has the last word (own_box_settle_tree / the vertex of k_path_tree).
// ------------------------------------------------------------------------------------------
struct TraceResult { float t; int prim; bool occluded; uint32_t visits, tests; };

// Traversal state of one ray.  trav_node() / trav_leaf() advance it by one inner node or one leaf, so a
// kernel can interleave rays of very different depth in one wave
// (dynamic ray fetch, see k_trace) instead of idling until the deepest ray of the wave is done.
template <bool SHADOW>
struct Trav {
  V3 o, d;
  float ix, iy, iz;
  float dist, t;
  int prim, cur, sp;
  bool occluded;
  uint32_t visits, tests;
#ifdef LR_DIAG
  uint32_t last_wait;               // cycles the last node / leaf step spent between issuing its fetch and having the rows
#endif
};

template <bool SHADOW>
LR_DEV void trav_begin(Trav<SHADOW>& s, V3 o, V3 d, float dist) {
  s.o = o; s.d = d; s.dist = dist;
  s.t = 3.0e38f; s.prim = -1; s.occluded = false; s.visits = 0; s.tests = 0;
  s.cur = 0; s.sp = 0;
  {
#pragma clang fp contract(fast)
    // a zero direction component would give inf * 0 = NaN below, and a NaN beside an inf makes the
    // min/max chain reject boxes the ray is inside of; 1e-20 is "parallel" at any scene scale and
    // keeps every product finite, so the test stays conservative
    float dx = __builtin_fabsf(d.x) < 1e-20f ? __builtin_copysignf(1e-20f, d.x) : d.x;
    float dy = __builtin_fabsf(d.y) < 1e-20f ? __builtin_copysignf(1e-20f, d.y) : d.y;
    float dz = __builtin_fabsf(d.z) < 1e-20f ? __builtin_copysignf(1e-20f, d.z) : d.z;
    s.ix = __builtin_amdgcn_rcpf(dx); s.iy = __builtin_amdgcn_rcpf(dy); s.iz = __builtin_amdgcn_rcpf(dz);
  }
}

// (stack entries carry no entry distance: a stale subtree costs one node fetch whose boxes then fail
//  the cull test, but 4 B per entry instead of 8 doubles the workgroups an LDS-bound CU can hold)
// The two homes of a stack entry are addressed through pointers of their own address spaces: with generic pointers the
// compiler folds the branches into one FLAT load of a selected address, and every pop then takes the flat path to LDS
// (longer than ds_read and it waits for all outstanding vector-memory loads as well).
typedef __attribute__((address_space(3))) uint32_t lds_u32;
typedef __attribute__((address_space(1))) uint32_t glb_u32;
LR_DEV void stack_store(const DevScene& sc, uint32_t* stk_n, int e, uint32_t v) {
  if (e < sc.stack_lds) ((lds_u32*)stk_n)[e * kBlock + threadIdx.x] = v;
  else ((glb_u32*)sc.stack_spill)[((size_t)blockIdx.x * sc.spill_depth + (e - sc.stack_lds)) * kBlock + threadIdx.x] = v;
}
LR_DEV uint32_t stack_load(const DevScene& sc, const uint32_t* stk_n, int e) {
  if (e < sc.stack_lds) return ((const lds_u32*)stk_n)[e * kBlock + threadIdx.x];
  return ((const glb_u32*)sc.stack_spill)[((size_t)blockIdx.x * sc.spill_depth + (e - sc.stack_lds)) * kBlock + threadIdx.x];
}
template <bool SHADOW>
LR_DEV bool trav_pop(const DevScene& sc, Trav<SHADOW>& s, const uint32_t* stk_n) {
  if (s.sp > 0) { --s.sp; s.cur = (int)stack_load(sc, stk_n, s.sp); return true; }
  return false;
}

// One inner node (s.cur >= 0) of the 4-wide tree: one 64-B fetch tests four child boxes, the hit children are
// ordered by entry distance (5-comparator network), the nearest becomes the next node and the others go on the
// stack far-first.  Half the dependent fetch rounds of a binary tree -- the traversal is latency-bound, not
// bandwidth-bound.  The boxes are 8-bit planes on the node's own power-of-two grid
// (lr_device.h); instead of decoding them to world space the RAY is moved into the grid: per axis
// t(q) = q * (step * 1/d) + (origin - o) * 1/d, one conversion and one FMA per plane.  false = ray finished.
constexpr int kEmptyChild = 0x7fffffff;
LR_DEV void order2(float& ka, int& ra, float& kb, int& rb) {
  bool sw = kb < ka;
  float k0 = sw ? kb : ka, k1 = sw ? ka : kb;
  int r0 = sw ? rb : ra, r1 = sw ? ra : rb;
  ka = k0; kb = k1; ra = r0; rb = r1;
}
// LR_NODE_ORDER 1: a node step finds the NEAREST hit child (three compare + select pairs on the entry distances) and pushes the
// other hit children in slot order, instead of sorting all four by entry distance with a 5-comparator network and pushing them
// far-first (0).  The visit order of the remaining children changes, never a result.
#ifndef LR_NODE_ORDER
#define LR_NODE_ORDER 0
#endif
// in: entry distances k0..k3 (inf = miss), children r0..r3.  out: r0 = the nearest hit child; r1, r2, r3 = the other hit children
// compacted to the front in slot order (the entries behind them are dead).
LR_DEV void nearest_then_slot_order(float k0, float k1, float k2, float k3, int& r0, int& r1, int& r2, int& r3) {
  const float inf = __builtin_inff();
  const bool m0 = k0 < inf, m1 = k1 < inf, m2 = k2 < inf, m3 = k3 < inf;
  const bool c01 = k1 < k0, c23 = k3 < k2;
  const float ka = __builtin_fminf(k0, k1), kb = __builtin_fminf(k2, k3);
  const int ra = c01 ? r1 : r0, rb = c23 ? r3 : r2;
  const bool cab = kb < ka;
  const int rn = cab ? rb : ra;
  const bool n0 = !cab & !c01, n1 = !cab & c01, n3 = cab & c23;        // which slot holds the nearest (n2 = cab & !c23)
  const bool n01 = !cab;
  const int o1 = n0 ? r1 : r0, o2 = n01 ? r2 : r1, o3 = n3 ? r2 : r3;  // the other three slots, in slot order
  const bool h1 = (n0 & m1) | (!n0 & m0), h2 = (n01 & m2) | (!n01 & m1), h3 = (n3 & m2) | (!n3 & m3);
  (void)n1; (void)h3;
  r0 = rn;
  r1 = h1 ? o1 : (h2 ? o2 : o3);
  r2 = (h1 & h2) ? o2 : o3;
  r3 = o3;
}
LR_DEV float qbyte(uint32_t w, int k) {                                  // byte k of w as a float (v_cvt_f32_ubyte0..3)
  return (float)((w >> (8 * k)) & 0xffu);
}
template <bool SHADOW>
LR_DEV bool trav_node(const DevScene& sc, Trav<SHADOW>& s, uint32_t* stk_n) {
  // (Tried: a breadth-first copy of the top 208 nodes in LDS served 78 % of the node fetches and changed nothing -- those
  //  nodes were L1 hits already; the step is bound by its ~150 dependent VALU instructions, not by the fetch.)
  const float4* n = sc.nodes + kNodeRows * (size_t)s.cur;
  float4 g = n[0], qa = n[1], qb = n[2], rc = n[3];
#ifdef LR_DIAG
  { unsigned long long w0 = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); s.last_wait = (uint32_t)(__builtin_amdgcn_s_memtime() - w0); }
#endif
  s.visits += 4;
  float k0, k1, k2, k3;
  int r0 = __float_as_int(rc.x), r1 = __float_as_int(rc.y), r2 = __float_as_int(rc.z), r3 = __float_as_int(rc.w);
  {
#pragma clang fp contract(fast)
    const float inf = __builtin_inff();
    const uint32_t eb = __float_as_uint(g.w);
    // step * 1/d is exact (a power of two times a float); the offset costs two roundings of magnitude |o| ulp, far inside
    // the padding the boxes carry
    const float ax = __uint_as_float((eb & 0xffu) << 23) * s.ix, bx = (g.x - s.o.x) * s.ix;
    const float ay = __uint_as_float(((eb >> 8) & 0xffu) << 23) * s.iy, by = (g.y - s.o.y) * s.iy;
    const float az = __uint_as_float(((eb >> 16) & 0xffu) << 23) * s.iz, bz = (g.z - s.o.z) * s.iz;
    // t(q) is monotonic in q with the sign of 1/d: the ray enters a slab through the lower plane when it travels up the
    // axis and through the upper plane otherwise -- pick the words once per node instead of a min and a max per plane pair
    // box-pruning bound: the light's distance (+ the visibility window), or the closest hit so far, + the culling slack of this
    // node (lumilly_hip.hip Wide4Builder: qb.z = 2 kappa, qb.w = kappa * diagonal): a child is culled when it begins beyond
    // bound + kappa * diagonal + 2 kappa * t_far(child) -- the error bound of any Moeller-Trumbore distance inside it, which at
    // grazing incidence may land in front of the triangle's own box
    const float bound = (SHADOW ? s.dist + 2.0f * kEps : s.t) + qb.w;
    const float slack2 = qb.z;
    const bool upx = s.ix >= 0.0f, upy = s.iy >= 0.0f, upz = s.iz >= 0.0f;
    const uint32_t wlx = __float_as_uint(qa.x), wly = __float_as_uint(qa.y), wlz = __float_as_uint(qa.z);
    const uint32_t whx = __float_as_uint(qa.w), why = __float_as_uint(qb.x), whz = __float_as_uint(qb.y);
    const uint32_t nx = upx ? wlx : whx, fx = upx ? whx : wlx;
    const uint32_t ny = upy ? wly : why, fy = upy ? why : wly;
    const uint32_t nz = upz ? wlz : whz, fz = upz ? whz : wlz;
#define LR_SLAB(K, C, R)                                                                                         \
    {                                                                                                            \
      float a0 = __builtin_fmaf(qbyte(nx, C), ax, bx), a1 = __builtin_fmaf(qbyte(fx, C), ax, bx);                \
      float b0 = __builtin_fmaf(qbyte(ny, C), ay, by), b1 = __builtin_fmaf(qbyte(fy, C), ay, by);                \
      float c0 = __builtin_fmaf(qbyte(nz, C), az, bz), c1 = __builtin_fmaf(qbyte(fz, C), az, bz);                \
      float tn = __builtin_fmaxf(__builtin_fmaxf(a0, b0), __builtin_fmaxf(c0, 0.0f));                            \
      float tfr = __builtin_fminf(__builtin_fminf(a1, b1), c1);                                                  \
      float tf = __builtin_fminf(tfr, __builtin_fmaf(slack2, tfr, bound));                                       \
      K = (tn <= tf && R != kEmptyChild) ? tn : inf;                                                             \
    }
    LR_SLAB(k0, 0, r0) LR_SLAB(k1, 1, r1) LR_SLAB(k2, 2, r2) LR_SLAB(k3, 3, r3)
#undef LR_SLAB
    int n_hit = (k0 < inf ? 1 : 0) + (k1 < inf ? 1 : 0) + (k2 < inf ? 1 : 0) + (k3 < inf ? 1 : 0);
    if (n_hit == 0) return trav_pop<SHADOW>(sc, s, stk_n);
#if LR_NODE_ORDER
    nearest_then_slot_order(k0, k1, k2, k3, r0, r1, r2, r3);
#else
    order2(k0, r0, k1, r1); order2(k2, r2, k3, r3); order2(k0, r0, k2, r2); order2(k1, r1, k3, r3); order2(k1, r1, k2, r2);
#endif
    if (n_hit == 4) { stack_store(sc, stk_n, s.sp, (uint32_t)r3); stack_store(sc, stk_n, s.sp + 1, (uint32_t)r2); stack_store(sc, stk_n, s.sp + 2, (uint32_t)r1); }
    else if (n_hit == 3) { stack_store(sc, stk_n, s.sp, (uint32_t)r2); stack_store(sc, stk_n, s.sp + 1, (uint32_t)r1); }
    else if (n_hit == 2) { stack_store(sc, stk_n, s.sp, (uint32_t)r1); }
    s.sp += n_hit - 1;
    s.cur = r0;
  }
  return true;
}

// One leaf (s.cur < 0): run the primitive tests of its range.  false = ray finished.
// LITERAL (retrace_tree): a primitive whose own test accepts counts only if aabb.rs:74-92 passes on its own box (bvh.rs:20-25).
template <bool SHADOW, bool LITERAL = false>
LR_DEV bool trav_leaf(const DevScene& sc, Trav<SHADOW>& s, const uint32_t* stk_n) {
  uint32_t enc = (uint32_t)~s.cur;
  uint32_t first = enc >> 3, count = enc & 7u;
  // the rows of primitive k+1 are requested before primitive k is tested: a leaf of n primitives costs one exposed
  // fetch round trip plus n tests, not n round trips
  const float4* q = sc.prims + 3 * (size_t)first;
  float4 n0 = q[0], n1 = q[1], n2 = q[2];
#ifdef LR_DIAG
  { unsigned long long w0 = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); s.last_wait = (uint32_t)(__builtin_amdgcn_s_memtime() - w0); }
#endif
  for (uint32_t k = 0; k < count; ++k) {
    float4 q0 = n0, q1 = n1, q2 = n2;
    if (k + 1 < count) { n0 = q[3 * k + 3]; n1 = q[3 * k + 4]; n2 = q[3 * k + 5]; }
    uint32_t idw = __float_as_uint(q0.w);
    int id = (int)(idw & 0x7fffffffu);
    float t; bool hit;
    s.tests += 1;
    if (idw >> 31) hit = sphere_test(v3(q0), q1.y, s.o, s.d, &t);
    else hit = tri_test(v3(q0), v3(q1), v3(q2), s.o, s.d, &t);
    if (!hit) continue;
    if (LITERAL && !own_box_exact(sc.pbox[kRecRows * (size_t)id], sc.pbox[kRecRows * (size_t)id + 1], s.o, s.d)) continue;
    if (SHADOW) {
      float diff = t - s.dist;
      if (diff < -kEps) { s.occluded = true; s.t = t; s.prim = id; return false; }   // (the occluder: own_box_settle_tree looks at its box)
      if (diff > kEps) continue;
    }
    if (t < s.t || (t == s.t && id < s.prim)) { s.t = t; s.prim = id; }   // closest-hit queries prune with s.t itself
  }
  return trav_pop<SHADOW>(sc, s, stk_n);
}

// The literal query on a tree scene (cold, see own_box_surely): the same walk -- the tree's padded boxes contain every
// primitive's own box, so a primitive whose own box passes is reached -- with own_box_exact behind every primitive test.
// A lane runs it alone (the lanes of its wave wait); its traversal stack is free, its walk being over.
template <bool SHADOW>
LR_DEV void retrace_tree(const DevScene& sc, Trav<SHADOW>& s, uint32_t* stk_n) {
  const uint32_t visits = s.visits, tests = s.tests;
  trav_begin<SHADOW>(s, s.o, s.d, s.dist);
  bool go = true;
#pragma unroll 1
  while (go) go = s.cur >= 0 ? trav_node<SHADOW>(sc, s, stk_n) : trav_leaf<SHADOW, true>(sc, s, stk_n);
  s.visits += visits; s.tests += tests;
}
// settle the primitive that decided a finished walk (closest hit, in-window hit or occluder): certainly a candidate, or the literal query
template <bool SHADOW>
LR_DEV void own_box_settle_tree(const DevScene& sc, Trav<SHADOW>& s, uint32_t* stk_n) {
  const size_t pi = s.prim < 0 ? 0 : (size_t)s.prim;
  const float4 lo = sc.pbox[kRecRows * pi], hi = sc.pbox[kRecRows * pi + 1];
  if (own_box_unsure(lo, hi, s.prim, s.o, s.d)) retrace_tree<SHADOW>(sc, s, stk_n);
}

// A burst of traversal for the lanes with `go` set (while-while: the wave first descends inner nodes
// together, then the lanes that reached a leaf run their primitive tests together).  Clears `go` of lanes
// whose ray is finished.
#ifndef LR_DESCEND_BURST
#define LR_DESCEND_BURST 3
#endif
constexpr int kDescendBurst = LR_DESCEND_BURST;
// LR_DIAG build (make diag): wave-uniform step / lane / cycle counters of the traversal loop, printed by lr_render
struct TravDiag { unsigned long long node_steps, node_lanes, leaf_steps, leaf_lanes, leaf_prims_max, leaf_prims, cyc_node, cyc_leaf, cyc_retire, cyc_fetch, cyc_total, rays, cyc_node_wait, cyc_leaf_wait; };
#ifdef LR_DIAG
#define LR_DIAG_ONLY(...) __VA_ARGS__
#else
#define LR_DIAG_ONLY(...)
#endif
template <bool SHADOW>
LR_DEV void trav_burst(const DevScene& sc, Trav<SHADOW>& s, uint32_t* stk_n, bool& go, TravDiag* dg = nullptr) {
  (void)dg;
#pragma unroll 1
  for (int it = 0; it < kDescendBurst; ++it) {
    bool nm = go && s.cur >= 0;
    uint64_t bm = __ballot(nm);
    if (bm == 0) break;
    LR_DIAG_ONLY(unsigned long long t0 = __builtin_amdgcn_s_memtime();)
    if (nm) go = trav_node<SHADOW>(sc, s, stk_n);
    LR_DIAG_ONLY(dg->node_steps += 1; dg->node_lanes += (unsigned)__builtin_popcountll(bm); dg->cyc_node += __builtin_amdgcn_s_memtime() - t0;)
    LR_DIAG_ONLY(dg->cyc_node_wait += (unsigned)__shfl((int)s.last_wait, (int)__builtin_ctzll(bm), 64);)
  }
#ifdef LR_DIAG
  {
    bool lm = go && s.cur < 0;
    uint64_t bl = __ballot(lm);
    if (bl) {
      uint32_t cnt = lm ? ((uint32_t)~s.cur & 7u) : 0u, mx = cnt, sm = cnt;
      for (int off = 32; off > 0; off >>= 1) { mx = max(mx, (uint32_t)__shfl_xor((int)mx, off, 64)); sm += (uint32_t)__shfl_xor((int)sm, off, 64); }
      dg->leaf_steps += 1; dg->leaf_lanes += (unsigned)__builtin_popcountll(bl); dg->leaf_prims_max += mx; dg->leaf_prims += sm;
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
#endif
  LR_DIAG_ONLY(uint64_t blw = __ballot(go && s.cur < 0);)
  if (go && s.cur < 0) go = trav_leaf<SHADOW>(sc, s, stk_n);
  LR_DIAG_ONLY(dg->cyc_leaf += __builtin_amdgcn_s_memtime() - t1; if (blw) dg->cyc_leaf_wait += (unsigned)__shfl((int)s.last_wait, (int)__builtin_ctzll(blw), 64);)
}

template <bool SHADOW>
LR_DEV TraceResult traverse(const DevScene& sc, V3 o, V3 d, float dist, uint32_t* stk_n, float* stk_t) {
  (void)stk_t;
  Trav<SHADOW> s;
  trav_begin<SHADOW>(s, o, d, dist);
  bool go = true;
  while (__ballot(go) != 0) trav_burst<SHADOW>(sc, s, stk_n, go);
  own_box_settle_tree<SHADOW>(sc, s, stk_n);
  TraceResult res; res.t = s.t; res.prim = s.prim; res.occluded = s.occluded; res.visits = s.visits; res.tests = s.tests;
  return res;
}

// wave-level skips of the v and t stages of the flat triangle test (2 = both, 1 = t only, 0 = none): as in the pair loop of
// lr_path.h they almost never skip with 64 unrelated rays per wave and their branches cost a little (config 3: 10 340 -> 10 370)
#ifndef LR_FLAT_BALLOTS
#define LR_FLAT_BALLOTS 0
#endif
// the shadow loop may stop when every lane of the wave already knows it is occluded: it almost never happens with 64 unrelated
// connections, and the test is a ballot + branch per primitive
#ifndef LR_FLAT_SHADOW_EXIT
#define LR_FLAT_SHADOW_EXIT 0
#endif
typedef float RowVec __attribute__((ext_vector_type(4)));
typedef RowVec __attribute__((address_space(4))) ConstRow;
LR_DEV float4 row4(RowVec v) { return make_float4(v.x, v.y, v.z, v.w); }

// One primitive of the flat loop: triangle.rs:69-100 / sphere.rs:42-55 and the closest-hit fold.  Returns
// true when the loop may stop (SHADOW: every lane of the wave already knows it is occluded).
template <bool SHADOW>
LR_DEV bool flat_test(float4 q0, float4 q1, float4 q2, V3 o, V3 d, float dist, TraceResult& res) {
  uint32_t idw = __float_as_uint(q0.w);
  int id = (int)(idw & 0x7fffffffu);
  float t = 0.0f; bool hit = false;
  // booleans are combined with & and | (no short circuit): the lane masks stay in SGPRs and the only
  // branches left are the wave-uniform ones
  if (idw >> 31) {
    hit = sphere_test(v3(q0), q1.y, o, d, &t);
  } else {
    // same operations as tri_test(); the __ballot tests only skip work that no lane of the wave
    // needs (all lanes already rejected), they never change a result
    V3 p0 = v3(q0), e1 = v3(q1), e2 = v3(q2);
    V3 pv = cross(d, e2);
    float det = dot(e1, pv);
    float invdet = rcp_exact_mid(det);                  // == 1.0f / det, see lr_math.h
    V3 tv = o - p0;
    float u = dot(tv, pv) * invdet;
    bool ok = bool(!(__builtin_fabsf(det) < kEps)) & bool(!(u < 0.0f)) & bool(!(u > 1.0f));
#if LR_FLAT_BALLOTS >= 2
    if (__ballot(ok) != 0)
#endif
    {
      V3 qv = cross(tv, e1);
      float v = dot(d, qv) * invdet;
      ok = ok & bool(!(v < 0.0f)) & bool(!(u + v > 1.0f));
#if LR_FLAT_BALLOTS >= 1
      if (__ballot(ok) != 0)
#endif
      {
        t = dot(e2, qv) * invdet;
        hit = ok & bool(!(t < kEps));
      }
    }
  }
  // (SHADOW too: the closest hit of the connection; scene.rs:127-131's window is applied to it after its own box has been settled)
  (void)dist;
  bool better = hit & bool(t < res.t);                   // rows come in primitive-id order: the first of equal hits is the lowest id
  res.t = better ? t : res.t;
  res.prim = better ? id : res.prim;
  return false;
}

// The literal query on a flat scene (cold: see own_box_surely): bvh.rs:20-25,131-141 over every primitive in id order --
// the primitive's own test, then aabb.rs:74-92 on its own box.  Runs under the lane mask of the undecided rays; the loop
// index is wave-uniform, so the rows still arrive as scalar loads.
LR_DEV void retrace_flat(const float4* __restrict__ prims, const float4* __restrict__ pbox, int n, V3 o, V3 d, float& t_out, int& prim_out) {
  const ConstRow* rows = (const ConstRow*)prims;
  const ConstRow* boxes = (const ConstRow*)pbox;
  float bt = 3.0e38f; int bp = -1;
#pragma unroll 1
  for (int k = 0; k < n; ++k) {
    const float4 q0 = row4(rows[3 * k]), q1 = row4(rows[3 * k + 1]), q2 = row4(rows[3 * k + 2]);
    const uint32_t idw = __float_as_uint(q0.w);
    const int id = (int)(idw & 0x7fffffffu);
    float t = 0.0f; bool hit;
    if (idw >> 31) hit = sphere_test(v3(q0), q1.y, o, d, &t);
    else hit = tri_test(v3(q0), v3(q1), v3(q2), o, d, &t);
    if (hit && t < bt && own_box_exact(row4(boxes[kRecRows * id]), row4(boxes[kRecRows * id + 1]), o, d)) { bt = t; bp = id; }
  }
  t_out = bt; prim_out = bp;
}
// settle the winner of the unfiltered search (t, prim) of a flat scene: certainly a candidate, or the literal query
LR_DEV void own_box_settle_flat(const float4* __restrict__ prims, const float4* __restrict__ pbox, int n, V3 o, V3 d, float& t, int& prim) {
  const int pi = prim < 0 ? 0 : prim;
  const float4 lo = pbox[kRecRows * pi], hi = pbox[kRecRows * pi + 1];
  if (own_box_unsure(lo, hi, prim, o, d)) retrace_flat(prims, pbox, n, o, d, t, prim);
}
// scene.rs:127-131 on the settled closest hit of a connection: nearer than the window = occluded, beyond it = no hit
LR_DEV void shadow_window(float dist, float t, int& prim, bool& occluded) {
  const float diff = t - dist;
  occluded = bool(prim >= 0) & bool(diff < -kEps);
  prim = diff > kEps ? -1 : prim;
}

// Small scenes (n_flat = number of primitives when <= kFlatMax, else 0): the SAH says a tree over a
// dozen primitives saves almost nothing, so the "tree" is one leaf and every lane tests every
// primitive.  The loop index is wave-uniform, so the primitive rows arrive through the scalar cache
// into SGPRs (s_load_dwordx4) and the loop is pure, fully converged VALU: no vector memory traffic,
// no stack, no divergence.  Same tests, same tie rule => same result as traverse().
// (traverse_flat_raw: the closest hit over all primitives, its own box not yet settled)
template <bool SHADOW>
LR_DEV TraceResult traverse_flat_raw(const float4* __restrict__ prims, int n, V3 o, V3 d, float dist) {
  TraceResult res; res.t = 3.0e38f; res.prim = -1; res.occluded = false; res.visits = 0; res.tests = (uint32_t)n;
  // constant address space: uniform loads from it are always scalar (s_load), whatever the alias analysis thinks
  const ConstRow* rows = (const ConstRow*)prims;
  // groups of two: both primitives' rows are requested together (one scalar-cache round trip per pair) and nothing
  // is carried around the loop -- a register-rotating software pipeline made the compiler copy every row into
  // VGPRs (10 v_mov per primitive), which cost more than the exposed latency
  for (int k = 0; k < n; k += 2) {
    const ConstRow* nx = rows + 3 * k;
    float4 a0 = row4(nx[0]), a1 = row4(nx[1]), a2 = row4(nx[2]);
    float4 b0 = row4(nx[3]), b1 = row4(nx[4]), b2 = row4(nx[5]);    // unconditional: the array is padded
    asm volatile("" :: "s"(a2.x), "s"(a2.y), "s"(a2.z));            // keeps the third row's s_load up here (it would sink into the triangle branch and stall there)
    if (flat_test<SHADOW>(a0, a1, a2, o, d, dist, res)) break;
    if (k + 1 >= n) break;
    if (flat_test<SHADOW>(b0, b1, b2, o, d, dist, res)) break;
  }
  return res;
}
template <bool SHADOW>
LR_DEV TraceResult traverse_flat(const float4* __restrict__ prims, const float4* __restrict__ pbox, int n, V3 o, V3 d, float dist) {
  TraceResult res = traverse_flat_raw<SHADOW>(prims, n, o, d, dist);
  own_box_settle_flat(prims, pbox, n, o, d, res.t, res.prim);
  if (SHADOW) shadow_window(dist, res.t, res.prim, res.occluded);
  return res;
}

// ------------------------------------------------------------------------------------------
// util.rs
// ------------------------------------------------------------------------------------------
LR_DEV void orthonormal_basis(V3 w, V3* t_out, V3* b_out) {            // util.rs:12-21
  V3 a = __builtin_fabsf(w.x) > kEps ? v3(0.0f, 1.0f, 0.0f) : v3(1.0f, 0.0f, 0.0f);
  V3 tangent = normalize(cross(a, w));
  *t_out = tangent; *b_out = cross(w, tangent);
}
LR_DEV V3 reflect(V3 self, V3 normal) { return -self + normal * (dot(self, normal) * 2.0f); }   // util.rs:30-32
LR_DEV bool refract(V3 self, V3 normal, float from_per_to_ior, V3* out) {                       // util.rs:34-42
  float dn = dot(self, normal);
  float cos2theta = 1.0f - (from_per_to_ior * from_per_to_ior) * (1.0f - (dn * dn));
  if (cos2theta > 0.0f) {
    *out = -self * from_per_to_ior - normal * (from_per_to_ior * -dn + __builtin_sqrtf(cos2theta));
    return true;
  }
  return false;
}
LR_DEV V3 orienting_normal(V3 out_, V3 normal) {                       // lambert.rs:14-21
  if (dot(normal, out_) < 0.0f) return normal * -1.0f;
  return normal;
}

// ------------------------------------------------------------------------------------------
// materials.  Mat = the three float4 rows of the material table.
// ------------------------------------------------------------------------------------------
struct Mat { float4 m0, m1, m2; };

// Radiance-only arithmetic.  BSDF values, pdfs, throughput and radiance never feed a decision (hit / miss,
// Russian roulette, light pick, ray directions): they only scale what is added to the film, and the
// parity bar on the film is 1e-4, not bit equality.  So their divisions use v_rcp_f32 (1 ulp) instead of
// the ten-instruction IEEE sequence.  Everything that builds a ray, a hit or a branch stays exact.
constexpr float kInvPi = 0.318309886183790671537767526745028724f;
LR_DEV float rcp_r(float x) { return __builtin_amdgcn_rcpf(x); }
LR_DEV V3 div_r(V3 a, float s) { float r = __builtin_amdgcn_rcpf(s); return v3(a.x * r, a.y * r, a.z * r); }
LR_DEV V3 mcolor(const Mat& m) { return v3(m.m0); }

LR_DEV float signed_mod(float base, float module) {                    // lambert.rs:58-64
  if (base > 0.0f) return det_fmod_pos(base, module);
  return module - det_fmod_pos(-base, module);
}
LR_DEV float checker_level(float lu, float lv, float su, float sv, float cu, float cv) {   // lambert.rs:72-90
  const float lw = 2.0f, sw = 1.0f, cw = 150.0f;
  if (lu < lw || lv < lw) return 0.5f;
  else if (su < sw || sv < sw) return 0.6f;
  else if ((cu < cw || cv < cw) && !(cu < cw && cv < cw)) return 0.8f;
  return 1.0f;
}
// the general form: six independent remainders, any magnitude (kept out of line: it only runs for |u|, |v| >= 2^24)
__attribute__((noinline)) LR_DEV float checker_general(float u, float v) {
  const float li = 150.0f, si = 30.0f, ci = 300.0f;
  return checker_level(signed_mod(u, li), signed_mod(v, li), signed_mod(u, si), signed_mod(v, si), signed_mod(u, ci), signed_mod(v, ci));
}
// f = |x| mod 300, then mod 150 and mod 30 FROM it: 300 = 2 * 150 and 150 = 5 * 30, every remainder is exactly
// representable and every step below is an exact operation (Sterbenz subtraction, small-integer products), so the
// three values are the same bits as three independent fmods -- two full remainders per call instead of six, and
// no per-remainder fallback branch in the shading code.
LR_DEV void mods_300_150_30(float ax, float* m300, float* m150, float* m30) {
  float q = __builtin_floorf(ax * (1.0f / 300.0f));
  float r = ax - q * 300.0f;                       // exact for ax < 2^24 whether q is right or one off
  r = r < 0.0f ? r + 300.0f : r;
  r = r >= 300.0f ? r - 300.0f : r;
  float h = r >= 150.0f ? r - 150.0f : r;
  float q3 = __builtin_floorf(h * (1.0f / 30.0f));
  float t = h - q3 * 30.0f;
  t = t < 0.0f ? t + 30.0f : t;
  t = t >= 30.0f ? t - 30.0f : t;
  *m300 = r; *m150 = h; *m30 = t;
}
LR_DEV float checker(float u, float v) {                               // lambert.rs:66-90 (grey level)
  float au = __builtin_fabsf(u), av = __builtin_fabsf(v);
  if (__ballot(!(au < 16777216.0f && av < 16777216.0f)) != 0) return checker_general(u, v);
  float u300, u150, u30, v300, v150, v30;
  mods_300_150_30(au, &u300, &u150, &u30);
  mods_300_150_30(av, &v300, &v150, &v30);
  // signed_mod: base > 0 ? fmod(base, m) : m - fmod(-base, m)
  bool up = u > 0.0f, vp = v > 0.0f;
  float lu = up ? u150 : 150.0f - u150, lv = vp ? v150 : 150.0f - v150;
  float su = up ? u30 : 30.0f - u30, sv = vp ? v30 : 30.0f - v30;
  float cu = up ? u300 : 300.0f - u300, cv = vp ? v300 : 300.0f - v300;
  return checker_level(lu, lv, su, sv, cu, cv);
}
// The three GGX terms are radiance-only (they end up in the BRDF value and in the pdf that divides it, never in a direction or
// a decision), so their divisions and the square root are the 1-ulp hardware forms (see rcp_r): 15 IEEE sequences of ~10
// instructions each per GGX vertex otherwise.  The half vector and the sampled direction (material_sample) stay exact.
LR_DEV float sqrt_r(float x) { return __builtin_amdgcn_sqrtf(x); }
// The half vector of a BRDF *evaluation* (ggx.rs:75) feeds the Fresnel term and the distribution only -- the value of the BRDF,
// never a direction or a decision -- so v_rsq_f32 and three multiplies could replace the IEEE square root and three IEEE divisions
// (~40 instructions per evaluation).  MEASURED AND NOT KEPT (round 4): +1.5 % on config 5, +2.0 % on config 3, but ggx.rs:34-39's
// D = a^2 / (pi ((a^2 - 1)(m.n)^2 + 1)^2) cancels catastrophically at the peak of a smooth lobe (roughness 0.2: one ulp of m.n
// is 1.5e-4 of D), and the BRDF row's film moved to 1.10e-4 from the oracle -- over the 1e-4 bar.  The exact form stays.
#ifndef LR_FAST_EVAL_HALF
#define LR_FAST_EVAL_HALF 0
#endif
LR_DEV V3 normalize_r(V3 a) {
#if LR_FAST_EVAL_HALF
  const float r = __builtin_amdgcn_rsqf(sqr_norm(a));
  return v3(a.x * r, a.y * r, a.z * r);
#else
  return normalize(a);
#endif
}
LR_DEV float ggx_g(float alpha, V3 v, V3 n) {                          // ggx.rs:27-32
  float a2 = alpha * alpha;
  float c = dot(v, n);
  float tan = rcp_r(c * c) - 1.0f;
  return 2.0f * rcp_r(1.0f + sqrt_r(1.0f + a2 * tan * tan));
}
LR_DEV float ggx_ndf(float alpha, V3 mm, V3 n) {                       // ggx.rs:34-39
  float a2 = alpha * alpha;
  float mdn = dot(mm, n);
  float x = (a2 - 1.0f) * mdn * mdn + 1.0f;
  return a2 * rcp_r(kPi * x * x);
}
LR_DEV float ggx_fresnel(float ior, V3 in_, V3 mm) {                   // ggx.rs:41-47
  float nnn = 1.0f - ior, nnp = 1.0f + ior;
  float f_0 = (nnn * nnn) * rcp_r(nnp * nnp);
  float c = dot(in_, mm);
  float c1 = 1.0f - c;
  float c2 = c1 * c1, c4 = c2 * c2;                                    // powi(5) = c1 * (c1^2)^2
  return f_0 + (1.0f - f_0) * (c1 * c4);
}
LR_DEV void ior_pair(float ior, V3 out_, V3 n, float* from_ior, float* to_ior) {   // ideal_refraction.rs:117-135
  if (dot(out_, n) > 0.0f) { *from_ior = 1.0f; *to_ior = ior; }
  else { *from_ior = ior; *to_ior = 1.0f; }
}
LR_DEV float fresnel_exact(float n1, float n2, V3 out_, V3 in_, V3 on) {           // ideal_refraction.rs:137-149
  float cos1 = dot(out_, on);
  float cos2 = dot(in_, -on);
  float a = (n1 * cos1 - n2 * cos2) / (n1 * cos1 + n2 * cos2);
  float b = (n1 * cos2 - n2 * cos1) / (n1 * cos2 + n2 * cos1);
  return (a * a + b * b) / 2.0f;
}

// lam_pre: the Lambert value of this vertex, computed once by the caller (it depends on the position only and a vertex asks
// for it twice: for the light sample and for the BSDF sample)
template <int MT>
LR_DEV V3 material_brdf(const Mat& m, V3 out_, V3 in_, V3 n, V3 pos, const V3* lam_pre = nullptr) {
  if (MT == LR_MAT_LAMBERT) {                                          // lambert.rs:32-35
    if (lam_pre) return *lam_pre;
    float g = checker(pos.x, pos.z) * kInvPi;
    return mcolor(m) * g;
  } else if (MT == LR_MAT_PHONG) {                                     // phong.rs:37-45
    V3 on = orienting_normal(out_, n);
    if (dot(in_, on) <= 0.0f) return v3(0, 0, 0);
    V3 r = reflect(out_, on);
    float c = dot(r, in_);
    float a = m.m2.x;
    return mcolor(m) * ((a + 2.0f) / (2.0f * kPi) * det_pow(c, a));
  } else if (MT == LR_MAT_BLINN_PHONG) {                               // blinn_phong.rs:37-47
    V3 on = orienting_normal(out_, n);
    if (dot(in_, on) <= 0.0f) return v3(0, 0, 0);
    V3 h = normalize(in_ + out_);
    float c = dot(h, on);
    float a = m.m2.x;
    return mcolor(m) * ((a + 2.0f) * (a + 4.0f) / (8.0f * kPi * (det_pow(2.0f, -a / 2.0f) + a)) * det_pow(c, a));
  } else if (MT == LR_MAT_GGX) {                                       // ggx.rs:71-85
    V3 on = orienting_normal(out_, n);
    if (dot(in_, on) <= 0.0f) return v3(0, 0, 0);
    V3 h = normalize_r(in_ + out_);
    float alpha = m.m2.x * m.m2.x;
    float f = ggx_fresnel(m.m2.y, in_, h);
    float g = ggx_g(alpha, in_, on) * ggx_g(alpha, out_, on);
    float d = ggx_ndf(alpha, h, on);
    return div_r(mcolor(m) * f * g * d, 4.0f * dot(in_, on) * dot(out_, on));
  } else {                                                             // ideal_refraction.rs:39-66
    V3 on = orienting_normal(out_, n);
    float from_ior, to_ior; ior_pair(m.m2.x, out_, n, &from_ior, &to_ior);
    float ratio = from_ior / to_ior;
    V3 r;
    if (refract(out_, on, ratio, &r)) {
      float fr = fresnel_exact(from_ior, to_ior, out_, r, on);
      if (dot(in_, on) > 0.0f) return mcolor(m) * 1.0f / dot(in_, n) * fr;
      float q = to_ior / from_ior;
      float ft = (1.0f - fr) * (q * q);
      return mcolor(m) * 1.0f / dot(in_, n) * ft;
    }
    return mcolor(m) * 1.0f / dot(in_, n);
  }
}

template <int MT>
LR_DEV void material_sample(const Mat& m, V3 out_, V3 n, const float* xi, V3* in_out, float* pdf_out) {
  if (MT == LR_MAT_LAMBERT) {                                          // lambert.rs:37-55, util.rs:87-96
    V3 on = orienting_normal(out_, n);
    V3 w = on, u, v; orthonormal_basis(w, &u, &v);
    float r1 = 2.0f * kPi * xi[0];
    float r2 = xi[1];
    float r2s = __builtin_sqrtf(r2);
    float s1, c1; det_sincos(r1, &s1, &c1);
    V3 s = v3(c1 * r2s, s1 * r2s, __builtin_sqrtf(1.0f - r2));
    V3 in_ = u * s.x + v * s.y + w * s.z;
    *in_out = in_; *pdf_out = dot(in_, n) * kInvPi;
  } else if (MT == LR_MAT_PHONG) {                                     // phong.rs:47-68
    V3 on = orienting_normal(out_, n);
    float a = m.m2.x;
    V3 r = reflect(out_, on);
    V3 w = r, u, v; orthonormal_basis(w, &u, &v);
    float r1 = 2.0f * kPi * xi[0];
    float r2 = xi[1];
    float t = det_pow(r2, 1.0f / (a + 2.0f));
    float ts = __builtin_sqrtf(1.0f - t * t);
    float s1, c1; det_sincos(r1, &s1, &c1);
    V3 in_ = u * c1 * ts + v * s1 * ts + w * t;
    float c = dot(r, in_);
    *in_out = in_; *pdf_out = (a + 2.0f) / (2.0f * kPi) * det_pow(c, a);
  } else if (MT == LR_MAT_BLINN_PHONG) {                               // blinn_phong.rs:49-72
    V3 on = orienting_normal(out_, n);
    float a = m.m2.x;
    V3 w = on, u, v; orthonormal_basis(w, &u, &v);
    float r1 = 2.0f * kPi * xi[0];
    float r2 = xi[1];
    float t = det_pow(r2, 1.0f / (a + 2.0f));
    float ts = __builtin_sqrtf(1.0f - t * t);
    float s1, c1; det_sincos(r1, &s1, &c1);
    V3 h = u * c1 * ts + v * s1 * ts + w * t;
    V3 in_ = h * (2.0f * dot(out_, h)) - out_;
    float c = dot(on, h);
    *in_out = in_; *pdf_out = (a + 2.0f) / (2.0f * kPi) * det_pow(c, a);
  } else if (MT == LR_MAT_GGX) {                                       // ggx.rs:87-113
    V3 on = orienting_normal(out_, n);
    V3 w = on, u, v; orthonormal_basis(w, &u, &v);
    float alpha = m.m2.x * m.m2.x;
    float r1 = 2.0f * kPi * xi[0];
    float r2 = xi[1];
    float tan = alpha * __builtin_sqrtf(r2 / (1.0f - r2));
    float x = 1.0f + tan * tan;
    float c = 1.0f / __builtin_sqrtf(x);
    float s = tan / __builtin_sqrtf(x);
    float s1, c1; det_sincos(r1, &s1, &c1);
    V3 h = u * c1 * s + v * s1 * s + w * c;
    float o_h = dot(out_, h);
    V3 in_ = h * (2.0f * o_h) - out_;
    float jacobian = rcp_r(4.0f * o_h);
    *in_out = in_; *pdf_out = ggx_ndf(alpha, h, on) * dot(h, on) * jacobian;
  } else {                                                             // ideal_refraction.rs:68-104
    float from_ior, to_ior; ior_pair(m.m2.x, out_, n, &from_ior, &to_ior);
    float ratio = from_ior / to_ior;
    V3 on = orienting_normal(out_, n);
    V3 r;
    if (refract(out_, on, ratio, &r)) {
      float fr = fresnel_exact(from_ior, to_ior, out_, r, on);
      if (xi[2] < fr) { *in_out = reflect(out_, on); *pdf_out = 1.0f * fr; }
      else { *in_out = r; *pdf_out = 1.0f * (1.0f - fr); }
    } else { *in_out = reflect(out_, on); *pdf_out = 1.0f; }
  }
}

template <int MT>
LR_DEV V3 material_coef(const Mat& m, V3 out_, V3 n, float fly_distance) {   // traits.rs:20-22, ideal_refraction.rs:106-113
  if (MT == LR_MAT_IDEAL_REFRACTION) {
    if (dot(out_, n) < 0.0f) {
      V3 v = -(v3(1.0f, 1.0f, 1.0f) - mcolor(m)) * m.m2.y * fly_distance;
      return v3(det_exp(v.x), det_exp(v.y), det_exp(v.z));
    }
  }
  return v3(1.0f, 1.0f, 1.0f);
}

// Per-lane BSDF type inside one wave (k_shade_all on a scene with several BSDFs): each type present among the active
// lanes runs under its own lane mask, so the code AROUND the material calls -- emission, roulette, the light sample, both
// RNG blocks, the stores -- is executed once for all hit lanes instead of once per type.  Same functions, same inputs.
constexpr int kMtDyn = 8;
#define LR_MT_CASES(CALL)                                                          \
  if ((MASK & 1u) && __ballot(mt == 0)) { if (mt == 0) { CALL(0) } }                \
  if ((MASK & 2u) && __ballot(mt == 1)) { if (mt == 1) { CALL(1) } }                \
  if ((MASK & 4u) && __ballot(mt == 2)) { if (mt == 2) { CALL(2) } }                \
  if ((MASK & 8u) && __ballot(mt == 3)) { if (mt == 3) { CALL(3) } }                \
  if ((MASK & 16u) && __ballot(mt == 4)) { if (mt == 4) { CALL(4) } }
template <uint32_t MASK>
LR_DEV V3 material_brdf_dyn(int mt, const Mat& m, V3 out_, V3 in_, V3 n, V3 pos, const V3* lam_pre = nullptr) {
  V3 r = v3(0.0f, 0.0f, 0.0f);
#define LR_CALL(K) r = material_brdf<K>(m, out_, in_, n, pos, lam_pre);
  LR_MT_CASES(LR_CALL)
#undef LR_CALL
  return r;
}
template <uint32_t MASK>
LR_DEV void material_sample_dyn(int mt, const Mat& m, V3 out_, V3 n, const float* xi, V3* in_out, float* pdf_out) {
#define LR_CALL(K) material_sample<K>(m, out_, n, xi, in_out, pdf_out);
  LR_MT_CASES(LR_CALL)
#undef LR_CALL
}
template <uint32_t MASK>
LR_DEV V3 material_coef_dyn(int mt, const Mat& m, V3 out_, V3 n, float fly_distance) {
  V3 r = v3(1.0f, 1.0f, 1.0f);
#define LR_CALL(K) r = material_coef<K>(m, out_, n, fly_distance);
  LR_MT_CASES(LR_CALL)
#undef LR_CALL
  return r;
}

// ------------------------------------------------------------------------------------------
// cameras  (camera.rs)
// ------------------------------------------------------------------------------------------
LR_DEV V3 arr3(const float* a) { return v3(a[0], a[1], a[2]); }

// the ray origin of a camera sample, given the aperture point's coordinates in the lens plane (thin lens only)
LR_DEV V3 camera_origin(const DevCamera& c, float apx, float apy) {
  V3 aperture_position = arr3(c.aperture_position);
  if (c.type == LR_CAMERA_THIN_LENS) return aperture_position + arr3(c.right) * apx + arr3(c.up) * apy;
  return aperture_position;
}
LR_DEV void camera_sample(const DevCamera& c, int x, int y, const Draw4& d, V3* o_out, V3* d_out, float* g_out, float* lens_out = nullptr) {
  V3 position = arr3(c.position), right = arr3(c.right), up = arr3(c.up);
  V3 aperture_position = arr3(c.aperture_position);
  const int res_w = fresh_s(c.res_w), res_h = fresh_s(c.res_h);
  if (c.type == LR_CAMERA_IDEAL_PINHOLE) {                             // camera.rs:64-115
    float px = ((((float)x + d.v[0]) / (float)res_w) - 0.5f) * c.sensor_w;
    float py = ((((float)y + d.v[1]) / (float)res_h) - 0.5f) * c.sensor_h;
    V3 point = position - right * px + up * py;
    *o_out = aperture_position;
    *d_out = normalize(aperture_position - point);
    *g_out = 1.0f;
  } else if (c.type == LR_CAMERA_THIN_LENS) {                          // camera.rs:411-476
    V3 forward = arr3(c.forward);
    float px = ((((float)x + d.v[0]) / (float)res_w) - 0.5f) * c.sensor_w;
    float py = ((((float)y + d.v[1]) / (float)res_h) - 0.5f) * c.sensor_h;
    V3 point = position - right * px + up * py;
    float au = 2.0f * kPi * d.v[2];
    float av = __builtin_sqrtf(d.v[3]) * c.aperture_radius;
    float s1, c1; det_sincos(au, &s1, &c1);
    float apx = c1 * av, apy = s1 * av;
    V3 apoint = aperture_position + right * apx + up * apy;
    if (lens_out) { lens_out[0] = apx; lens_out[1] = apy; }
    V3 sensor_center = aperture_position - point;
    V3 object_plane = sensor_center * (c.focus_distance / dot(sensor_center, forward));
    *o_out = apoint;
    *d_out = normalize(aperture_position + object_plane - apoint);
    V3 dir = normalize(apoint - point);                                // geometry_term :446-455
    float cos_term = dot(dir, forward);
    float dd = c.aperture_sensor_distance * rcp_r(cos_term);          // the geometry term only scales the sample's radiance: 1-ulp forms (rcp_r)
    *g_out = cos_term * cos_term * rcp_r(dd * dd);
  } else {                                                             // camera.rs:168-188
    float p = ((float)x + d.v[0]) / (float)res_w * kPi * 2.0f;
    float t = ((float)y + d.v[1]) / (float)res_h * kPi;
    float sp, cp, st, ct; det_sincos(p, &sp, &cp); det_sincos(t, &st, &ct);
    *o_out = aperture_position;
    *d_out = v3(st * cp, st * sp, ct);
    *g_out = 1.0f;
  }
}

// ------------------------------------------------------------------------------------------
// sky  (sky.rs)
// ------------------------------------------------------------------------------------------
// IBL: the texel a direction looks up (sky.rs:57-78)
// maps taller than 32768 rows (2^31 texels and up): sky.rs:72-77 in 64 bits, kept out of line -- it would otherwise size the registers of every caller
__attribute__((noinline)) LR_DEV uint64_t sky_index_wide(uint32_t width, uint32_t height, float fx, float fy) {
  uint64_t all = (uint64_t)width * height;
  uint64_t x = fx > 0.0f ? (uint64_t)fx : 0, y = fy > 0.0f ? (uint64_t)fy : 0;   // `as usize` saturates
  return (y * width + x) % all;
}
LR_DEV uint64_t sky_texel_index(const DevScene& sc, V3 dir) {
  float theta = det_acos(dir.y);
  float phi = det_atan2(dir.z, dir.x);
  float uu = (phi + kPi + sc.sky_lon) / (2.0f * kPi);
  float ru = det_fmod1_pos(__builtin_fabsf(uu));                       // Rust `%`: the remainder keeps the dividend's sign
  float u = uu >= 0.0f ? ru : -ru;
  float vv = theta / kPi;
  float rv = det_fmod1_pos(__builtin_fabsf(vv));
  float v = vv >= 0.0f ? rv : -rv;
  uint32_t height = (uint32_t)fresh_s(sc.sky_h), width = height * 2u;
  float fx = __builtin_floorf((float)width * u), fy = __builtin_floorf((float)height * v);
  if (height <= 32768u) {
    // |u|, |v| < 1, so x <= width and y <= height: the index stays below 2 * width * height and `% all` is one conditional
    // subtraction in 32 bits (a 64-bit remainder is a ~200-instruction routine; `as usize` saturates: negative / NaN -> 0)
    const uint32_t all = width * height;
    const uint32_t x = fx > 0.0f ? (uint32_t)fx : 0u, y = fy > 0.0f ? (uint32_t)fy : 0u;
    const uint32_t i = y * width + x;
    return i >= all ? i - all : i;
  }
  return sky_index_wide(width, height, fx, fy);
}
// one texel of the map.  RGBE words decode as the `image` crate does (c * 2^(e - 136), sky.rs:45-48): the scale is a power of two
// built from e (lr_scene_create guarantees e >= 10, so it is a normal float) and c < 256, so the product is exact -- the same
// f32 bits the float4 map would have held, from a quarter of the bytes
LR_DEV float4 sky_texel(const DevScene& sc, uint64_t i) {
  if (sc.texels_rgbe) {                                                  // (wave-uniform)
    const uint32_t w = sc.texels_rgbe[i];
    const float scale = __uint_as_float(((w >> 24) - 9u) << 23);
    return make_float4((float)(w & 0xffu) * scale, (float)((w >> 8) & 0xffu) * scale, (float)((w >> 16) & 0xffu) * scale, 0.0f);
  }
  return sc.texels[i];
}
LR_DEV V3 sky_radiance(const DevScene& sc, V3 dir) {
  if (sc.sky_type == LR_SKY_UNIFORM) return v3(sc.sky_color[0], sc.sky_color[1], sc.sky_color[2]);   // sky.rs:17-21
  return v3(sky_texel(sc, sky_texel_index(sc, dir)));
}

// ------------------------------------------------------------------------------------------
// emitter sampling  (objects.rs:37-51, triangle.rs:140-149, sphere.rs:79-84, util.rs:108-116)
// ------------------------------------------------------------------------------------------
// objects.rs:37-51: the first emitter k (instance order) with roulette <= cumulative area
LR_DEV int emitter_index(const DevScene& sc, float roulette) {
  int k = 0;
  int n = sc.n_emitters;
  if (n <= 8) {
    // the cumulative areas are nondecreasing, so "first k with roulette <= cum[k]" is a count; the rows are
    // read with wave-uniform indices from the constant address space (s_load), no dependent vector loads
    const ConstRow* er = (const ConstRow*)sc.emit;
    for (int j = 0; j < n - 1; ++j) k += !(roulette <= er[3 * j + 2].w) ? 1 : 0;
  } else {
    // emissive meshes (scene_loader.rs:254-262 binds a light to every triangle of an object): binary search for the
    // same "first k" -- the sums are nondecreasing, so it finds what the reference's linear scan finds; the last
    // emitter catches a roulette that rounding left above the final sum (the reference would panic there, objects.rs:50)
    int lo = 0, hi = n - 1;
    while (lo < hi) { int mid = (lo + hi) >> 1; if (roulette <= sc.emit[3 * mid + 2].w) hi = mid; else lo = mid + 1; }
    k = lo;
  }
  return k;
}
// `st` says where the emitter rows are read from: the scene blob (DevState) or a copy the kernel staged in LDS (lr_path.h)
LR_DEV float4 emit_row(const DevState&, const DevScene& sc, int i) { return sc.emit[i]; }
template <class ST>
LR_DEV void sample_emission(const DevScene& sc, const ST& st, const Draw4& d, V3* value, float* pdf) {
  float roulette = sc.emission_area * d.v[1];
  int k = emitter_index(sc, roulette);
  float4 e0 = emit_row(st, sc, 3 * k), e1 = emit_row(st, sc, 3 * k + 1), e2 = emit_row(st, sc, 3 * k + 2);
  if (__float_as_uint(e0.w) == LR_PRIM_TRIANGLE) {
    float u = d.v[2], v = d.v[3];
    float mn = fmin_rs(u, v), mx = fmax_rs(u, v);
    *value = v3(e0) * mn + v3(e1) * (1.0f - mx) + v3(e2) * (mx - mn);
  } else {
    float r1 = 2.0f * kPi * d.v[2];
    float r2 = d.v[3] * 2.0f - 1.0f;
    float r2s = __builtin_sqrtf(1.0f - r2 * r2);
    float s1, c1; det_sincos(r1, &s1, &c1);
    *value = v3(e0) + e1.x * v3(c1 * r2s, s1 * r2s, r2);
  }
  *pdf = e1.w;
}

LR_DEV float russian_roulette(float init, int d, const DevParams& rp) {   // scene.rs:64-76
  float p = init;
  if (d > rp.depth_limit) {
    int k = d - rp.depth_limit;
    float h = 1.0f;                                                     // 0.5^k, exact
    for (int i = 0; i < k && i < 200; ++i) h = h * 0.5f;
    p = p * h;
  }
  if (d <= rp.depth && p > 0.0f) p = 1.0f;
  return p;
}

// ------------------------------------------------------------------------------------------
// work items: item = chunk * n_pix + pixel rank; pixel rank -> (x, y) through the tile list
// ------------------------------------------------------------------------------------------
LR_DEV uint32_t rank_to_pixel(const DevState& st, const DevCamera& cam, uint32_t rank) {
  int lo = 0, hi = st.n_tiles - 1;
  while (lo < hi) { int mid = (lo + hi + 1) >> 1; if (st.tile_prefix[mid] <= rank) lo = mid; else hi = mid - 1; }
  int4 t = st.tiles[lo];
  uint32_t off = rank - st.tile_prefix[lo];
  uint32_t x = (uint32_t)t.x + off % (uint32_t)t.z, y = (uint32_t)t.y + off / (uint32_t)t.z;
  return y * (uint32_t)cam.res_w + x;
}
// the search runs once per pixel rank per render (k_rank_table); the path kernels read the table
LR_DEV uint32_t item_pixel(const DevState& st, const DevCamera& cam, uint32_t rank) { (void)cam; return st.rank_pixel[rank]; }

// work item <-> (chunk, pixel rank): DevState "Order of the work items of a launch"
struct ItemRef { uint32_t chunk, rank; };
LR_DEV ItemRef item_decode(const DevState& st, uint32_t item) {
  ItemRef r;
  const uint32_t shift = fresh_s(st.sub_shift);
  if (shift == 0u) { const uint32_t n_pix = fresh_s(st.n_pix); r.chunk = item / n_pix; r.rank = item - r.chunk * n_pix; return r; }
  const uint32_t last0 = st.sub_last_item0;
  const bool last = item >= last0;                                      // one division serves both cases
  const uint32_t num = last ? item - last0 : item >> shift, den = last ? st.sub_last_pix : st.n_chunks, quo = num / den, rem = num - quo * den;
  r.chunk = last ? quo : rem;
  r.rank = last ? st.sub_last_rank0 + rem : (quo << shift) + (item & ((1u << shift) - 1u));
  return r;
}
LR_DEV size_t item_index(const DevState& st, uint32_t chunk, uint32_t rank) {
  if (st.sub_shift == 0u) return (size_t)chunk * st.n_pix + rank;
  if (rank >= st.sub_last_rank0) return (size_t)st.sub_last_item0 + (size_t)chunk * st.sub_last_pix + (rank - st.sub_last_rank0);
  const uint32_t b = rank >> st.sub_shift;
  return (((size_t)b * st.n_chunks + chunk) << st.sub_shift) + (rank & ((1u << st.sub_shift) - 1u));
}

// Start the camera sample (pixel, sample) in `slot`.
LR_DEV void start_sample(const DevScene& sc, const DevState& st, const DevParams& rp, uint32_t slot, uint32_t pixel, uint32_t sample) {
  Draw4 d0 = rng_block(rp.seed, pixel, sample, 0u);
  int x = (int)(pixel % (uint32_t)sc.cam.res_w), y = (int)(pixel / (uint32_t)sc.cam.res_w);
  V3 o, d; float g;
  camera_sample(sc.cam, x, y, d0, &o, &d, &g);
  st.ray_o[slot] = make_float4(o.x, o.y, o.z, __int_as_float(0));
  st.ray_d[slot] = make_float4(d.x, d.y, d.z, g);
  st.thr[slot] = make_float4(1.0f, 1.0f, 1.0f, __uint_as_float(pixel));
  st.rad[slot] = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(sample));
}

// Per-segment pool of work items (the device end of the pixel tile queue).  Each segment keeps up to
// two ranges of item ids it reserved from the global dispenser; thread 0 tops the pool up at the
// start of a segment pass so that the pass can never run short (at most one item per slot per
// pass), lanes then draw from it with LDS atomics only.
struct PoolLds { uint32_t r0, a0, r1, a1, taken; };

LR_DEV void pool_begin(const DevState& st, uint32_t seg, uint32_t need_max, PoolLds* pl, bool reset, uint32_t batch = kSeg) {
  uint4 p = reset ? make_uint4(0, 0, 0, 0) : st.pool[seg];       // {r0 next, r0 end, r1 next, r1 end}
  if (p.x >= p.y) { p.x = p.z; p.y = p.w; p.z = p.w = 0; }
  if (p.y - p.x < need_max && p.z >= p.w) {
    uint32_t cur = __hip_atomic_load(st.next_item, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (cur < st.n_items) {
      uint32_t nb = atomicAdd(st.next_item, batch);
      if (nb < st.n_items) { p.z = nb; p.w = st.n_items - nb > batch ? nb + batch : st.n_items; }
    }
    if (p.x >= p.y) { p.x = p.z; p.y = p.w; p.z = p.w = 0; }
  }
  pl->r0 = p.x; pl->a0 = p.y - p.x; pl->r1 = p.z; pl->a1 = p.w - p.z; pl->taken = 0;
}
LR_DEV void pool_end(const DevState& st, uint32_t seg, const PoolLds* pl) {
  uint32_t t0 = pl->taken < pl->a0 ? pl->taken : pl->a0;
  uint32_t rest = pl->taken - t0;
  uint32_t t1 = rest < pl->a1 ? rest : pl->a1;
  st.pool[seg] = make_uint4(pl->r0 + t0, pl->r0 + pl->a0, pl->r1 + t1, pl->r1 + pl->a1);
}

// End-of-path bookkeeping for the whole wave (main.rs:92-121): fold the finished sample into the
// chunk sum, draw new work items from the segment pool, start the next camera sample.
//   finished : this lane's path just ended with radiance L (sample index `sample`, pixel `pixel`)
//   fresh    : this lane has no work item yet (k_generate)
// Returns true when the lane found the pool (and the dispenser) empty and retired its slot.
LR_DEV bool finish_and_regenerate(const DevScene& sc, const DevState& st, const DevParams& rp, PoolLds* pl, uint32_t slot,
                                  bool finished, bool fresh, V3 L, float g_term, uint32_t pixel, uint32_t sample,
                                  const float4* acc_early = nullptr) {   // acc_early: the slot's chunk-sum row, loaded by the caller ahead of the shading (one round trip less)
  bool need_item = fresh;
  uint32_t item = 0;
  V3 sum = v3(0, 0, 0);
  if (finished) {
    float4 a = acc_early ? *acc_early : st.acc[slot];
    item = __float_as_uint(a.w);
    V3 delta = L;
    if (sc.cam.type == LR_CAMERA_THIN_LENS) delta = (L * g_term) * sc.cam.weight2;   // e * (sens / pdf); 1 * 1 for the others
    sum = v3(a) + delta;
    sample += 1;
    uint32_t chunk = item_decode(st, item).chunk;
    uint32_t end = st.chunk_start[chunk + 1];
    if (sample >= end) { st.partial[item] = make_float4(sum.x, sum.y, sum.z, 0.0f); need_item = true; }
  }
  uint32_t k = wave_reserve(&pl->taken, need_item);
  bool retired = false;
  // the pool is a cache of reserved item ids; a lane that finds it empty goes to the global dispenser itself
  // (one atomic per wave, rare), so a slot only retires when the dispenser is really dry
  bool short_of = need_item && !(k < pl->a0 + pl->a1);
  uint32_t direct = wave_reserve(st.next_item, short_of);
  if (need_item) {
    if (k < pl->a0) item = pl->r0 + k;
    else if (k - pl->a0 < pl->a1) item = pl->r1 + (k - pl->a0);
    else if (direct < st.n_items) item = direct;
    else retired = true;
    if (!retired) {
      const ItemRef ir = item_decode(st, item);
      const uint32_t rank = ir.rank, chunk = ir.chunk;
      pixel = item_pixel(st, sc.cam, rank);
      sample = st.chunk_start[chunk];
      sum = v3(0, 0, 0);
    }
  }
  if (finished || fresh) {
    if (retired) {
      st.ray_o[slot] = make_float4(0, 0, 0, __int_as_float(-1));
    } else {
      st.acc[slot] = make_float4(sum.x, sum.y, sum.z, __uint_as_float(item));
      start_sample(sc, st, rp, slot, pixel, sample);
    }
  }
  return retired;
}

// ==========================================================================================
// kernels.  Every kernel walks 512-slot segments (grid-stride over segments, 256 threads).
// ==========================================================================================
#ifndef LR_TEMPLATE_KERNELS_ONLY      // (lr_flat.hip instantiates template kernels only: the plain ones live in lumilly_hip.hip)
__global__ void __launch_bounds__(kBlock) k_generate(DevScene sc, DevState st, DevParams rp) {
  __shared__ PoolLds pl;
  __shared__ uint32_t s_retired;
  for (uint32_t seg = blockIdx.x; seg < st.n_seg; seg += gridDim.x) {
    if (threadIdx.x == 0) { pool_begin(st, seg, kSeg, &pl, true); s_retired = 0; }
    __syncthreads();
    for (uint32_t step = 0; step < kSeg / kBlock; ++step) {
      uint32_t slot = seg * kSeg + step * kBlock + threadIdx.x;
      bool r = finish_and_regenerate(sc, st, rp, &pl, slot, false, true, v3(0, 0, 0), 1.0f, 0, 0);
      (void)wave_reserve(&s_retired, r);
    }
    __syncthreads();
    if (threadIdx.x == 0) { pool_end(st, seg, &pl); if (s_retired) atomicAdd(st.n_retired, s_retired); }
    __syncthreads();
  }
}
#endif

// ------------------------------------------------------------------------------------------
// Ray sort (north star: "ray sort/compaction").  Before a workgroup walks the rays of its range (up to 16 K path slots)
// it bins them by (direction octant, 4x4x4 cell of the origin in Morton order): a 9-bit counting sort through an LDS
// histogram, the sorted order left as 16-bit local slot numbers in `order`.  Waves then draw 64 consecutive entries of
// that order at a time, so the lanes of a wave start next to each other, leave in the same octant, descend the same
// top of the tree (same cache lines, same near-first child order) and stay in step between node and leaf phases.
// The order only changes WHICH lane walks WHICH ray: hits, lists and films are the same bits.
// ------------------------------------------------------------------------------------------
constexpr int kSortBins = 512;
constexpr uint32_t kDeadKey = 0xffffu;
struct SortLds { uint32_t hist[kSortBins]; uint32_t wsum[kBlock / 64]; };

LR_DEV uint32_t ray_key(const DevScene& sc, V3 o, V3 d) {
  uint32_t oct = (d.x < 0.0f ? 1u : 0u) | (d.y < 0.0f ? 2u : 0u) | (d.z < 0.0f ? 4u : 0u);
  float fx = (o.x - sc.key_lo[0]) * sc.key_scale[0], fy = (o.y - sc.key_lo[1]) * sc.key_scale[1], fz = (o.z - sc.key_lo[2]) * sc.key_scale[2];
  uint32_t cx = (uint32_t)__builtin_fminf(__builtin_fmaxf(fx, 0.0f), 3.0f);      // fmax(NaN, 0) = 0: any ray gets a valid bin
  uint32_t cy = (uint32_t)__builtin_fminf(__builtin_fmaxf(fy, 0.0f), 3.0f);
  uint32_t cz = (uint32_t)__builtin_fminf(__builtin_fmaxf(fz, 0.0f), 3.0f);
  uint32_t m = (cx & 1u) | ((cy & 1u) << 1) | ((cz & 1u) << 2) | ((cx & 2u) << 2) | ((cy & 2u) << 3) | ((cz & 2u) << 4);
  return (oct << 6) | m;
}
// histogram -> exclusive offsets, in place (call with the whole workgroup, between barriers); returns the number of keys
LR_DEV uint32_t sort_scan(SortLds& sl) {
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  uint32_t a = sl.hist[2 * tid], b = sl.hist[2 * tid + 1];
  uint32_t v = a + b;
  for (int off = 1; off < 64; off <<= 1) { uint32_t t = __shfl_up(v, off, 64); if ((int)lane >= off) v += t; }
  if (lane == 63u) sl.wsum[wave] = v;
  __syncthreads();
  uint32_t pre = 0, all = 0;
  for (uint32_t w = 0; w < kBlock / 64; ++w) { uint32_t x = sl.wsum[w]; all += x; if (w < wave) pre += x; }
  uint32_t excl = pre + v - (a + b);
  sl.hist[2 * tid] = excl; sl.hist[2 * tid + 1] = excl + a;
  __syncthreads();
  return all;
}
static_assert(kSortBins == 2 * kBlock, "sort_scan gives two bins to each thread");

// A wave's window on the sorted order: 64 entries at a time (one per lane in `cv`), the next 64 requested one draw
// ahead (`nv`) so that a refill never waits for the order list.  pos / len / nlen are wave-uniform.  Written as a macro
// pair over plain locals: as a struct passed by reference the compiler kept the window in scratch memory and waited
// for the prefetch right behind its issue.
#define LR_CHUNK_DECL uint32_t ck_cv = 0, ck_nv = 0, ck_pos = 0, ck_len = 0, ck_nlen = 0;
// request the next 64 entries (no-op once the range's dispenser has run dry: ck_nlen stays 0)
#define LR_CHUNK_FETCH(S_NEXT, ORD, N_LIVE)                                                             \
  {                                                                                                     \
    uint32_t base_ = 0;                                                                                 \
    if (lane_id() == 0) base_ = atomicAdd(S_NEXT, 64u);                                                 \
    base_ = (uint32_t)__builtin_amdgcn_readfirstlane((int)base_);                                       \
    ck_nlen = base_ < (N_LIVE) ? ((N_LIVE) - base_ < 64u ? (N_LIVE) - base_ : 64u) : 0u;               \
    ck_nv = lane_id() < ck_nlen ? (uint32_t)(ORD)[base_ + lane_id()] : 0u;                              \
  }
// Hands entries of the sorted order to the lanes with NEED set: GOT = this lane received one (its local slot number in
// LOCAL), EXHAUSTED = nothing is left for this wave to draw.  Whole (converged) wave.
#define LR_CHUNK_TAKE(S_NEXT, ORD, N_LIVE, NEED, GOT, LOCAL, EXHAUSTED)                                 \
  {                                                                                                     \
    bool need_ = (NEED);                                                                                \
    uint64_t nm_ = __ballot(need_);                                                                     \
    while (nm_ != 0) {                                                                                  \
      if (ck_pos == ck_len) {                                                                           \
        if (ck_nlen == 0) break;                                                                        \
        ck_cv = ck_nv; ck_len = ck_nlen; ck_pos = 0;                                                    \
        LR_CHUNK_FETCH(S_NEXT, ORD, N_LIVE)                                                             \
      }                                                                                                 \
      const uint32_t avail_ = ck_len - ck_pos;                                                          \
      const uint32_t r_ = rank_in_mask(nm_);                                                            \
      const bool take_ = need_ && r_ < avail_;                                                          \
      const uint32_t v_ = (uint32_t)__shfl((int)ck_cv, (int)(take_ ? ck_pos + r_ : 0u), 64);            \
      if (take_) { LOCAL = v_; GOT = true; need_ = false; }                                             \
      const uint32_t cnt_ = (uint32_t)__builtin_popcountll(nm_);                                        \
      ck_pos += cnt_ < avail_ ? cnt_ : avail_;                                                          \
      nm_ = __ballot(need_);                                                                            \
    }                                                                                                   \
    EXHAUSTED = ck_pos == ck_len && ck_nlen == 0;                                                       \
  }

// Closest-hit stage of the streaming pipeline.  A workgroup owns `spb` consecutive segments (spb * 512
// path slots) per pass and hands them to its waves through an LDS dispenser:
//   * tree scenes: persistent while-while traversal with DYNAMIC RAY FETCH -- a lane whose ray is done
//     writes its hit, appends its slot to the range's list for its BSDF and draws the next slot, as soon
//     as at most kRefillBelow lanes of the wave are still walking.  Ray depths differ by 10x in one
//     wave (box walls vs. the 100k-triangle mesh); without refill the wave idles at ~14 % lane use, and
//     a pass must be long (up to kMaxGroup segments) or the run-down of its last rays does the same.
//   * flat scenes (<= 32 primitives): every lane tests every primitive, nothing diverges, plain loop.
#ifndef LR_TRACE_WAVES
#define LR_TRACE_WAVES 6
#endif
#ifndef LR_SHADOW_WAVES
#define LR_SHADOW_WAVES 6
#endif
constexpr int kMaxGroup = 32;            // segments a workgroup may own at once (16 K rays per pass: long passes amortise the run-down of the last rays)
#ifndef LR_REFILL_BELOW
#define LR_REFILL_BELOW 32
#endif
constexpr int kRefillBelow = LR_REFILL_BELOW;         // refill the wave when at most this many lanes are still traversing

template <bool COUNT, bool SORTED>
__global__ void __launch_bounds__(kBlock, LR_TRACE_WAVES) k_trace(DevScene sc, DevState st, const float4* __restrict__ flat_prims, uint32_t spb) {
  extern __shared__ uint32_t lds[];
  __shared__ uint32_t s_cnt[8];                                    // one list per BSDF for the whole range of this pass (k_shade cuts it into 512-entry slices)
  __shared__ uint32_t s_next;
  __shared__ uint32_t s_stat[ST_COUNT];
  SortLds& s_sort = *(SortLds*)lds;                                 // the histogram borrows the traversal stack's LDS: the sort runs before any ray walks
  uint32_t* stk_n = lds;
  const uint32_t tid = threadIdx.x;
  if (tid < ST_COUNT) s_stat[tid] = 0;
  uint32_t n_rays = 0, n_vis = 0, n_tst = 0;
  for (uint32_t seg0 = blockIdx.x * spb; seg0 < st.n_seg; seg0 += gridDim.x * spb) {
    const uint32_t nsegs = st.n_seg - seg0 < spb ? st.n_seg - seg0 : spb;
    const uint32_t total = nsegs * kSeg, slot0 = seg0 * kSeg;
    if (tid < 8) s_cnt[tid] = 0;
    if (tid == 0) s_next = 0;
    __syncthreads();
    if (sc.n_flat > 0) {
      for (uint32_t base = 0; base < total; base += kBlock) {
        uint32_t slot = slot0 + base + tid;
        bool active = false; int qid = -1;
        float4 ro = st.ray_o[slot];
        if (__float_as_int(ro.w) >= 0) {
          float4 rd = st.ray_d[slot];
          active = true;
          TraceResult r = traverse_flat<false>(flat_prims, sc.pbox, sc.n_flat, v3(ro), v3(rd), 0.0f);
          st.hit[slot] = make_float2(r.t, __int_as_float(r.prim));
          if (!st.dense_shade) qid = r.prim < 0 ? kQMiss : (int)sc.prim_qid[r.prim];
          if (COUNT) n_tst += r.tests;
        }
        n_rays += (uint32_t)__builtin_popcountll(__ballot(active));
        uint64_t todo = st.dense_shade ? 0ull : __ballot(active);
        while (todo) {
          int lead = (int)__builtin_ctzll(todo);
          int q = __shfl(qid, lead, 64);
          bool mine = active && qid == q;
          uint32_t idx = wave_reserve(&s_cnt[q], mine);
          if (mine) st.q_shade[((size_t)q * st.n_seg + seg0) * kSeg + idx] = slot;
          todo &= ~__ballot(mine);
        }
      }
    } else {
      // ---- ray sort: bin the range's live rays, leave the sorted order in st.order[slot0 ...] ----
      constexpr bool sorted = SORTED;
      uint32_t n_live = total;
      if (sorted) {
        for (uint32_t b = tid; b < (uint32_t)kSortBins; b += kBlock) s_sort.hist[b] = 0;
        __syncthreads();
        for (uint32_t i = tid; i < total; i += kBlock) {
          float4 ro = st.ray_o[slot0 + i];
          uint32_t key = kDeadKey;
          if (__float_as_int(ro.w) >= 0) {
            float4 rd = st.ray_d[slot0 + i];
            key = ray_key(sc, v3(ro), v3(rd));
            atomicAdd(&s_sort.hist[key], 1u);
          }
          st.sort_key[slot0 + i] = (uint16_t)key;
        }
        __syncthreads();
        n_live = sort_scan(s_sort);
        for (uint32_t i = tid; i < total; i += kBlock) {
          uint32_t key = st.sort_key[slot0 + i];                    // this thread's own store
          if (key != kDeadKey) { uint32_t pos = atomicAdd(&s_sort.hist[key], 1u); st.order[slot0 + pos] = (uint16_t)i; }
        }
        __threadfence_block();
        __syncthreads();
      }
      const uint16_t* ord = sorted ? st.order + slot0 : nullptr;
      LR_CHUNK_DECL
      if (sorted) LR_CHUNK_FETCH(&s_next, ord, n_live)
      Trav<false> tr;
      bool has = false, fin = false;
      uint32_t slot = 0;
      LR_DIAG_ONLY(TravDiag dg = {}; unsigned long long tq0 = __builtin_amdgcn_s_memtime(), tq = tq0;)
      while (true) {
        // ---- converged point: retire finished rays (hit record + compaction into the (segment, BSDF) lists) ----
        LR_DIAG_ONLY(tq = __builtin_amdgcn_s_memtime();)
        {
          bool f = has && fin;
          int key = -1;
          if (f) {
            own_box_settle_tree<false>(sc, tr, stk_n);               // bvh.rs:20-25: the own box of the closest hit has the last word
            st.hit[slot] = make_float2(tr.t, __int_as_float(tr.prim));
            if (!st.dense_shade) key = tr.prim < 0 ? kQMiss : (int)sc.prim_qid[tr.prim];
            if (COUNT) { n_vis += tr.visits; n_tst += tr.tests; }
          }
          uint64_t todo = st.dense_shade ? 0ull : __ballot(f);     // k_shade_all walks the slots themselves: no lists
          while (todo) {
            int lead = (int)__builtin_ctzll(todo);
            int k = __shfl(key, lead, 64);
            bool mine = f && key == k;
            uint32_t idx = wave_reserve(&s_cnt[k], mine);
            if (mine) st.q_shade[((size_t)k * st.n_seg + seg0) * kSeg + idx] = slot;
            todo &= ~__ballot(mine);
          }
          if (f) { has = false; fin = false; }
        }
        // ---- dynamic fetch: free lanes draw the next rays of this workgroup's range ----
        LR_DIAG_ONLY(dg.cyc_retire += __builtin_amdgcn_s_memtime() - tq; tq = __builtin_amdgcn_s_memtime();)
        bool need = !has;
        bool exhausted;
        if (sorted) {
          uint32_t local = 0; bool got = false;
          LR_CHUNK_TAKE(&s_next, ord, n_live, need, got, local, exhausted)
          if (got) {
            slot = slot0 + local;
            float4 ro = st.ray_o[slot], rd = st.ray_d[slot];         // every entry of the order is a live ray
            trav_begin<false>(tr, v3(ro), v3(rd), 0.0f);
            has = true;
          }
          n_rays += (uint32_t)__builtin_popcountll(__ballot(got));   // counted per wave (a scalar), not per lane
        } else {
          uint32_t idx = wave_reserve(&s_next, need);
          exhausted = __ballot(need && idx >= total) != 0;           // the dispenser is monotonic: one lane past the end = empty for all
          bool started = false;
          if (need && idx < total) {
            slot = slot0 + idx;
            float4 ro = st.ray_o[slot], rd = st.ray_d[slot];         // both rows in one round trip (a retired slot costs a wasted 16 B, only at the end of a render)
            if (__float_as_int(ro.w) >= 0) {
              trav_begin<false>(tr, v3(ro), v3(rd), 0.0f);
              has = true; started = true;
            }
          }
          n_rays += (uint32_t)__builtin_popcountll(__ballot(started));   // counted per wave (a scalar), not per lane
        }
        LR_DIAG_ONLY(dg.cyc_fetch += __builtin_amdgcn_s_memtime() - tq;)
        if (__ballot(has) == 0) { if (exhausted) break; continue; }
        // ---- walk until the wave thins out (or, with nothing left to fetch, until it is done) ----
        const int thresh = exhausted ? 0 : kRefillBelow;
        bool go = has && !fin;
#ifdef LR_DIAG
        do { trav_burst<false>(sc, tr, stk_n, go, &dg); } while (__builtin_popcountll(__ballot(go)) > thresh);
#else
        do { trav_burst<false>(sc, tr, stk_n, go); } while (__builtin_popcountll(__ballot(go)) > thresh);
#endif
        fin = has && !go;
      }
#ifdef LR_DIAG
      dg.cyc_total = __builtin_amdgcn_s_memtime() - tq0;
      dg.rays = n_rays;
      if (lane_id() == 0) {
        unsigned long long* o = st.stats + (size_t)kStatShards * kStatStride + 8;
        const unsigned long long* v = (const unsigned long long*)&dg;
        for (int i = 0; i < (int)(sizeof(TravDiag) / 8); ++i) atomicAdd(o + i, v[i]);
      }
#endif
    }
    __syncthreads();
    if (tid < kNumShadeQueues) st.c_shade[tid * st.n_seg + seg0] = s_cnt[tid];      // the range's lists start in its first segment's storage
    __syncthreads();
  }
  stat_accumulate(&s_stat[ST_SEGMENTS], lane_id() == 0 ? n_rays : 0u);
  if (COUNT) { stat_accumulate(&s_stat[ST_NODE_VISITS], n_vis); stat_accumulate(&s_stat[ST_PRIM_TESTS], n_tst); }
  __syncthreads();
  stat_flush(st.stats, s_stat);
}

// One path vertex (scene.rs:153-193): MT in 0..4 = BSDF of the hit material, MT == kQMiss = sky.
// Reads the slot's ray / hit / throughput / radiance, handles emission, Russian roulette, the
// direct-light sample (its occlusion test is left to the shadow stage: *has_shadow) and the BSDF sample.
// Returns true when the path ended here; L / g_term / pixel / sample then describe the finished sample.
// `st` may point at global memory (streaming pipeline) or at LDS (resident pipeline).
// which spelling of the generator's multiply-adds a path-state type wants (lr_math.h mad32): the tree kernel's lane state says LO
template <class ST> struct RngLo { static constexpr bool value = false; };
struct VertexOut { bool finished, has_shadow; V3 L; float g_term; uint32_t pixel, sample; bool sky_fetch; };

// The rows a vertex reads: the slot's state and (for a hit) the primitive's shading record.  k_shade_all requests them for
// all classes at once, ahead of the per-class code; the other callers load them where they always did.
struct VertexIn { float4 ro, rd, th, ra; float2 h; float4 sh, m0, m1, m2; };

// MT = the BSDF type, kQMiss, or kMtDyn: type per lane (`mt`, one of MASK's bits), see material_*_dyn
// ST = where the vertex's outputs go: DevState (rows of the slot in HBM or LDS) or the lane's own registers (lr_path.h)
// NEE = 1 / 0: the integrator is known when the kernel is instantiated (pt-direct / pt) and the other one's code -- and its
// registers -- are not compiled in; -1: read it from rp (the kernels that serve both)
template <int MT, uint32_t MASK = 0, int NEE = -1, class ST>
LR_DEV VertexOut shade_vertex_core(const DevScene& sc, ST& st, const DevParams& rp, uint32_t slot, const VertexIn& in, int mt = MT) {
  VertexOut out; out.finished = false; out.has_shadow = false; out.sky_fetch = false;
  const bool nee_mode = NEE < 0 ? rp.integrator == LR_INTEGRATOR_PT_DIRECT : NEE != 0;
  const float4 ro = in.ro, rd = in.rd, th = in.th, ra = in.ra;
  int depth = __float_as_int(ro.w);
  out.pixel = __float_as_uint(th.w); out.sample = __float_as_uint(ra.w);
  V3 o = v3(ro), d = v3(rd), T = v3(th);
  V3 L = v3(ra); out.g_term = rd.w;
  if (MT == kQMiss) {                                              // scene.rs:29 / :43
    // k_shade_all requests the IBL texel of a miss together with the shading records of the hits (mt = 1: it is in in.sh)
    L = L + T * (mt == 1 ? v3(in.sh) : sky_radiance(sc, d));
    out.sky_fetch = sc.sky_type == LR_SKY_IBL;
    out.finished = true;
  } else {
    float t = in.h.x;
    V3 pos = o + d * t;                                            // triangle.rs:93 / sphere.rs:55
    float4 sh = in.sh;
    Mat m; m.m0 = in.m0; m.m1 = in.m1; m.m2 = in.m2;
    uint32_t mw = __float_as_uint(sh.w);
    V3 nrm = (mw >> 31) ? normalize(pos - v3(sh)) : v3(sh);        // sphere.rs:56 / triangle.rs:36
    V3 out_ = -d;
    V3 emission = v3(m.m1);
    bool no_emission = nee_mode && depth > 0;                      // scene.rs:189 passes `true` below depth 0
    // scene.rs:155-159 / :175-179: l_e is the emission or ZERO, and the recursion multiplies it by every factor above it
    // (`l_e + (.. + brdf * coef * L_i * cos / pdf) / p`).  Adding T * 0 changes nothing while T is finite, but once a factor
    // was inf or NaN (a sampled direction whose pdf underflowed to 0: 0 * c / 0) the reference's pixel is NaN whatever the
    // rest of the path returns -- the throughput form must poison the sample the same way (found by the fuzzer, seed 400649)
    const bool emits = !(rp.no_direct_emitter && depth == 0) && !no_emission && dot(out_, nrm) > 0.0f;
    L = L + T * (emits ? emission : v3(0.0f, 0.0f, 0.0f));
    float p = russian_roulette(m.m1.w, depth, rp);                 // scene.rs:161 / :181
    constexpr bool rng_lo = NEE == 1 && RngLo<ST>::value;
    Draw4 d1 = rng_block<rng_lo>(rp.seed, out.pixel, out.sample, 1u + 2u * (uint32_t)depth);
    if (p != 1.0f && d1.v[0] >= p) {                               // scene.rs:162-164 / :182-184
      out.finished = true;
    } else {
      // lambert.rs:32-35 depends on the vertex position only: one evaluation serves the light sample and the BSDF sample
      V3 lam_val = v3(0.0f, 0.0f, 0.0f);
      const V3* lam_pre = nullptr;
      if constexpr (NEE != 0) {                                      // without a light sample the value is asked for once anyway
        if constexpr (MT == LR_MAT_LAMBERT) { lam_val = material_brdf<LR_MAT_LAMBERT>(m, out_, out_, nrm, pos); lam_pre = &lam_val; }
        else if constexpr (MT == kMtDyn && (MASK & 1u) != 0u) {
          if (__ballot(mt == LR_MAT_LAMBERT) != 0) { if (mt == LR_MAT_LAMBERT) lam_val = material_brdf<LR_MAT_LAMBERT>(m, out_, out_, nrm, pos); }
          lam_pre = &lam_val;
        }
      }
      // ---- direct light (scene.rs:104-151); the occlusion test itself is the shadow stage ----
      if (nee_mode && !(sqr_norm(emission) > 0.0f) && sc.emission_area > 0.0f) {
        V3 lp; float lpdf;
        sample_emission(sc, st, d1, &lp, &lpdf);
        V3 direct_path = lp - pos;
        float d2 = sqr_norm(direct_path);
        float dist = __builtin_sqrtf(d2);
        V3 dir = direct_path / dist;
        V3 point_normal = orienting_normal(out_, nrm);
        float point_cos = dot(dir, point_normal);
        if (point_cos > 0.0f) {
          V3 brdf;
          if constexpr (MT == kMtDyn) brdf = material_brdf_dyn<MASK>(mt, m, out_, dir, point_normal, pos, lam_pre);
          else brdf = material_brdf<MT>(m, out_, dir, point_normal, pos, lam_pre);
          V3 W = T * (brdf * (point_cos * rcp_r(d2 * lpdf * p)));
          st.sh_d[slot] = make_float4(dir.x, dir.y, dir.z, dist);
          st.sh_w[slot] = make_float4(W.x, W.y, W.z, 0.0f);
          out.has_shadow = true;
        }
      }
      // ---- BSDF sample (scene.rs:78-102) ----
      Draw4 d2r = rng_block<rng_lo>(rp.seed, out.pixel, out.sample, 2u + 2u * (uint32_t)depth);
      V3 in_; float pdf;
      V3 brdf, coef;
      if constexpr (MT == kMtDyn) {
        material_sample_dyn<MASK>(mt, m, out_, nrm, d2r.v, &in_, &pdf);
        brdf = material_brdf_dyn<MASK>(mt, m, out_, in_, nrm, pos, lam_pre);
        coef = material_coef_dyn<MASK>(mt, m, out_, nrm, t);
      } else {
        material_sample<MT>(m, out_, nrm, d2r.v, &in_, &pdf);
        brdf = material_brdf<MT>(m, out_, in_, nrm, pos, lam_pre);
        coef = material_coef<MT>(m, out_, nrm, t);
      }
      float c = dot(in_, nrm);
      const float q = pdf * p;
      V3 f;
      if (__builtin_expect(!(__builtin_fabsf(q) >= 1.17549435e-38f), 0)) {
        // a pdf in the DENORMAL range (or 0 / NaN): a Phong / Blinn-Phong direction almost across the lobe, c^20 ~ 1e-40 -- six samples of the
        // 2.1e9 of the stated row.  v_rcp_f32 reads a denormal as 0 and returns inf, where scene.rs:101's `brdf * coef * L_i * cos / pdf` -- the
        // BRDF value as small as the pdf that divides it -- is finite (found by the whole-frame check against the literal restatement):
        // the reference's operations in its order, IEEE divisions (cold)
        f = ((brdf * coef) * c) / pdf / p;
      } else {
        f = brdf * coef * (c * rcp_r(q));
      }
      T = T * f;
      st.ray_o[slot] = make_float4(pos.x, pos.y, pos.z, __int_as_float(depth + 1));
      st.ray_d[slot] = make_float4(in_.x, in_.y, in_.z, out.g_term);
      st.thr[slot] = make_float4(T.x, T.y, T.z, th.w);
      st.rad[slot] = make_float4(L.x, L.y, L.z, ra.w);
    }
  }
  out.L = L;
  return out;
}

template <int MT>
LR_DEV VertexOut shade_vertex(const DevScene& sc, const DevState& st, const DevParams& rp, uint32_t slot) {
  VertexIn in;
  in.ro = st.ray_o[slot]; in.rd = st.ray_d[slot]; in.th = st.thr[slot]; in.ra = st.rad[slot];
  in.h = make_float2(0.0f, 0.0f); in.sh = in.m0 = in.m1 = in.m2 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  if (MT != kQMiss) {
    // resident pipeline: the hit record lives in the slot's (not yet written) shadow-weight row
    if (st.hit) in.h = st.hit[slot]; else { float4 w = st.sh_w[slot]; in.h = make_float2(w.x, w.y); }
    const float4* rec = sc.shade + kRecRows * (size_t)__float_as_int(in.h.y);   // one 64-B record: no dependent second fetch for the material
    in.sh = rec[0]; in.m0 = rec[1]; in.m1 = rec[2]; in.m2 = rec[3];
  }
  return shade_vertex_core<MT>(sc, st, rp, slot, in);
}

// The shadow stage for one slot (scene.rs:127-147): adds the direct-light term if the sampled point is visible.
LR_DEV void shadow_resolve(const DevScene& sc, const DevState& st, uint32_t slot, V3 o, V3 dir, const TraceResult& r) {
  if (!r.occluded && r.prim >= 0) {                                // scene.rs:127-131
    V3 pos = o + dir * r.t;
    const float4* rec = sc.shade + kRecRows * (size_t)r.prim;
    float4 sh = rec[0], em = rec[2];
    uint32_t mw = __float_as_uint(sh.w);
    V3 light_normal = (mw >> 31) ? normalize(pos - v3(sh)) : v3(sh);
    float light_cos = dot(-dir, light_normal);
    if (light_cos > 0.0f) {                                        // scene.rs:133-139
      V3 l_i = v3(em);                                             // emission of what was hit (scene.rs:144)
      float4 w = st.sh_w[slot];
      float4 ra = st.rad[slot];
      V3 L = v3(ra) + v3(w) * l_i * light_cos;
      st.rad[slot] = make_float4(L.x, L.y, L.z, ra.w);
    }
  }
}

// Slot-ordered shading.  k_trace appends a slot to its BSDF's list when the ray retires, i.e. in traversal-completion
// order (and, with the ray sort, in sorted-ray order): shading in that order reads and writes the 16-B state rows of
// scattered slots, eight unrelated wave-instructions per 128-B line.  The prologue below turns the list back into slot
// order -- a bitmap of the range's slots in LDS (one atomic OR per entry), then every thread writes the set bits of its
// two words into an LDS list behind a workgroup prefix sum -- so that 64 lanes shade (nearly) consecutive slots.
constexpr int kRangeSlots = kMaxGroup * kSeg;                       // 16384: slots of the largest range
struct ShadeOrderLds { uint32_t bits[kRangeSlots / 32]; uint32_t wsum[kBlock / 64]; uint16_t list[kRangeSlots]; };
static_assert(kRangeSlots / 32 == 2 * kBlock, "two bitmap words per thread");

template <int MT>
__global__ void __launch_bounds__(kBlock) k_shade(DevScene sc, DevState st, DevParams rp) {
  extern __shared__ uint32_t shade_lds[];                           // ShadeOrderLds when st.shade_ordered, else nothing
  __shared__ PoolLds pl;
  __shared__ uint32_t s_shadow, s_retired;
  __shared__ uint32_t s_stat[ST_COUNT];
  ShadeOrderLds& so = *(ShadeOrderLds*)shade_lds;
  if (threadIdx.x < ST_COUNT) s_stat[threadIdx.x] = 0;
  uint32_t n_done = 0, n_sky = 0;
  // k_trace left ONE list per BSDF for each range of trace_spb segments (contiguous storage).  A workgroup shades a
  // whole range: the range's slots then all draw their work items from ONE pool (the range's first segment's) that
  // this workgroup owns, so a pool can only run dry, never be stranded with items nobody asks for.
  const uint32_t n_ranges = (st.n_seg + st.trace_spb - 1) / st.trace_spb;
  for (uint32_t g = blockIdx.x; g < n_ranges; g += gridDim.x) {
    const uint32_t g0 = g * st.trace_spb;
    const uint32_t n = st.c_shade[MT * st.n_seg + g0];
    const uint32_t* queue = st.q_shade + ((size_t)MT * st.n_seg + g0) * kSeg;
    uint32_t* shadow_q = st.q_shadow + ((size_t)(MT == kQMiss ? 0 : MT) * st.n_seg + g0) * kSeg;   // same contiguous layout: one list per range
    // reservations of a quarter of the list (512..8192 items): draws are a fraction of the finishing lanes; a pool
    // that still runs short falls back to the global dispenser (finish_and_regenerate), it never loses work
    const uint32_t batch = n / 4 < (uint32_t)kSeg ? (uint32_t)kSeg : (n / 4 > 8192u ? 8192u : n / 4);
    if (threadIdx.x == 0) { pool_begin(st, g0, n < batch ? n : batch, &pl, false, batch); s_shadow = 0; s_retired = 0; }
    const bool ordered = st.shade_ordered != 0;
    const uint32_t slot0 = g0 * kSeg;
    if (ordered) {
      const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
      so.bits[2 * tid] = 0; so.bits[2 * tid + 1] = 0;
      __syncthreads();
      for (uint32_t i = tid; i < n; i += kBlock) { uint32_t l = queue[i] - slot0; atomicOr(&so.bits[l >> 5], 1u << (l & 31u)); }
      __syncthreads();
      uint32_t w0 = so.bits[2 * tid], w1 = so.bits[2 * tid + 1];
      uint32_t c = (uint32_t)__builtin_popcount(w0) + (uint32_t)__builtin_popcount(w1), v = c;
      for (int off = 1; off < 64; off <<= 1) { uint32_t t = __shfl_up(v, off, 64); if ((int)lane >= off) v += t; }
      if (lane == 63u) so.wsum[wave] = v;
      __syncthreads();
      uint32_t pre = v - c;
      for (uint32_t w = 0; w < wave; ++w) pre += so.wsum[w];
      while (w0) { uint32_t b = (uint32_t)__builtin_ctz(w0); w0 &= w0 - 1; so.list[pre++] = (uint16_t)(64u * tid + b); }
      while (w1) { uint32_t b = (uint32_t)__builtin_ctz(w1); w1 &= w1 - 1; so.list[pre++] = (uint16_t)(64u * tid + 32u + b); }
    }
    __syncthreads();
    for (uint32_t base = 0; base < n; base += kBlock) {
      uint32_t i = base + threadIdx.x;
      bool valid = i < n;
      uint32_t slot = valid ? (ordered ? slot0 + so.list[i] : queue[i]) : 0;
      VertexOut v; v.finished = false; v.has_shadow = false; v.L = v3(0, 0, 0); v.g_term = 1.0f; v.pixel = 0; v.sample = 0; v.sky_fetch = false;
      float4 acc_row = st.acc[slot];                                // every wave finishes some path: fetch the chunk sum with the rest of the state
      if (valid) v = shade_vertex<MT>(sc, st, rp, slot);
      if (v.finished) n_done += 1;
      if (v.sky_fetch) n_sky += 1;
      bool r = finish_and_regenerate(sc, st, rp, &pl, slot, v.finished, false, v.L, v.g_term, v.pixel, v.sample, &acc_row);
      (void)wave_reserve(&s_retired, r);
      if (MT != kQMiss) {
        uint32_t idx = wave_reserve(&s_shadow, v.has_shadow);
        if (v.has_shadow) shadow_q[idx] = slot;
      }
    }
    __syncthreads();
    if (MT != kQMiss && threadIdx.x == 0) st.c_shadow[MT * st.n_seg + g0] = s_shadow;
    if (threadIdx.x == 0) {
      pool_end(st, g0, &pl);
      if (s_retired) atomicAdd(st.n_retired, s_retired);
    }
    __syncthreads();
  }
  stat_accumulate(&s_stat[ST_SAMPLES], n_done);
  if (MT == kQMiss) stat_accumulate(&s_stat[ST_SKY], n_sky);
  __syncthreads();
  stat_flush(st.stats, s_stat);
}

// Dense shading: ONE launch per iteration walks the slots themselves, 64 consecutive slots per wave, and runs the vertex
// code of every class (BSDF or miss) present among them under its lane mask.  The per-class lists of k_shade keep the
// lanes of a wave on one code path, but a wave then touches the 16-B state rows of 64 slots spread over 64 / density
// consecutive ones: a class that holds a quarter of the slots (the misses of an open scene) moves four times the bytes
// it needs, and every line is read and written once per class that has a slot in it.  The stage is bandwidth-bound and
// has VALU time to spare, so it pays divergence instead: every state line is fetched once and written back once.
// MASK = the BSDF types that can occur (a scene's materials); bit kQMiss is implied.
// 5 waves per SIMD = 96 VGPRs (32-96 B of scratch): not for this kernel's own occupancy but for what fits BESIDE it -- one
// of its waves leaves room for five 80-VGPR traversal waves of another slot group on the same SIMD, two for four.  At 4 waves
// the allocation follows the code (112 or 120 registers) and a 4-register change cost the mesh scene 8 %.
#ifndef LR_DENSE_WAVES
#define LR_DENSE_WAVES 5
#endif
template <uint32_t MASK>
__global__ void __launch_bounds__(kBlock, LR_DENSE_WAVES) k_shade_all(DevScene sc, DevState st, DevParams rp) {
  __shared__ PoolLds pl;
  __shared__ uint32_t s_shadow, s_retired;
  __shared__ uint32_t s_stat[ST_COUNT];
  if (threadIdx.x < ST_COUNT) s_stat[threadIdx.x] = 0;
  uint32_t n_done = 0, n_sky = 0;
  const uint32_t n_ranges = (st.n_seg + st.trace_spb - 1) / st.trace_spb;
  const bool ibl = sc.sky_type == LR_SKY_IBL;
  for (uint32_t g = blockIdx.x; g < n_ranges; g += gridDim.x) {
    const uint32_t g0 = g * st.trace_spb;
    const uint32_t nsegs = st.n_seg - g0 < st.trace_spb ? st.n_seg - g0 : st.trace_spb;
    const uint32_t n = nsegs * kSeg, slot0 = g0 * kSeg;
    uint32_t* shadow_q = st.q_shadow + (size_t)g0 * kSeg;          // one shadow list per range (list 0's storage; k_shadow gets mt_mask = 1)
    const uint32_t batch = n / 4 < (uint32_t)kSeg ? (uint32_t)kSeg : (n / 4 > 8192u ? 8192u : n / 4);
    if (threadIdx.x == 0) { pool_begin(st, g0, n < batch ? n : batch, &pl, false, batch); s_shadow = 0; s_retired = 0; }
    __syncthreads();
    for (uint32_t base = 0; base < n; base += kBlock) {
      const uint32_t slot = slot0 + base + threadIdx.x;           // n is a multiple of kBlock: every lane has a slot
      // two round trips per slot: every state row at once, then the shading record of the primitive that was hit
      VertexIn in;
      in.h = st.hit[slot];
      in.ro = st.ray_o[slot]; in.rd = st.ray_d[slot]; in.th = st.thr[slot]; in.ra = st.rad[slot];
      float4 acc_row = st.acc[slot];
      int key = -1;
      in.sh = in.m0 = in.m1 = in.m2 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      if (__float_as_int(in.ro.w) >= 0) {
        int prim = __float_as_int(in.h.y);
        key = kQMiss;
        if (prim >= 0) {
          const float4* rec = sc.shade + kRecRows * (size_t)prim;
          in.sh = rec[0]; in.m0 = rec[1]; in.m1 = rec[2]; in.m2 = rec[3];
          key = (int)__float_as_uint(in.m0.w);                      // {color.rgb, type bits}: the record names its own class
        } else if (ibl) {
          in.sh = sky_texel(sc, sky_texel_index(sc, v3(in.rd)));        // same round trip as the records of the hit lanes
        }
      }
      VertexOut v; v.finished = false; v.has_shadow = false; v.L = v3(0, 0, 0); v.g_term = 1.0f; v.pixel = 0; v.sample = 0; v.sky_fetch = false;
      if constexpr ((MASK & (MASK - 1u)) == 0u) {                    // one BSDF in the scene: its code, statically
        constexpr int K = MASK == 1u ? 0 : (MASK == 2u ? 1 : (MASK == 4u ? 2 : (MASK == 8u ? 3 : 4)));
        if (key == K) v = shade_vertex_core<K>(sc, st, rp, slot, in);
      } else {
        if (key >= 0 && key < kQMiss) v = shade_vertex_core<kMtDyn, MASK>(sc, st, rp, slot, in, key);
      }
      if (__ballot(key == kQMiss)) { if (key == kQMiss) v = shade_vertex_core<kQMiss>(sc, st, rp, slot, in, ibl ? 1 : 0); }
      if (v.finished) n_done += 1;
      if (v.sky_fetch) n_sky += 1;
      bool r = finish_and_regenerate(sc, st, rp, &pl, slot, v.finished, false, v.L, v.g_term, v.pixel, v.sample, &acc_row);
      (void)wave_reserve(&s_retired, r);
      uint32_t idx = wave_reserve(&s_shadow, v.has_shadow);
      if (v.has_shadow) shadow_q[idx] = slot;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      st.c_shadow[g0] = s_shadow;
      pool_end(st, g0, &pl);
      if (s_retired) atomicAdd(st.n_retired, s_retired);
    }
    __syncthreads();
  }
  stat_accumulate(&s_stat[ST_SAMPLES], n_done);
  stat_accumulate(&s_stat[ST_SKY], n_sky);
  __syncthreads();
  stat_flush(st.stats, s_stat);
}

// Shadow stage of the streaming pipeline (scene.rs:127-147).  mt_mask = BSDF types present in the scene
// (their k_shade wrote this iteration's shadow lists).  Same workgroup ranges and the same dynamic ray
// fetch as k_trace; the work list is the concatenation of the range's per-BSDF shadow lists.
template <bool COUNT, bool SORTED>
__global__ void __launch_bounds__(kBlock, LR_SHADOW_WAVES) k_shadow(DevScene sc, DevState st, uint32_t mt_mask, const float4* __restrict__ flat_prims, uint32_t spb) {
  extern __shared__ uint32_t lds[];
  __shared__ uint32_t s_pref[8];                                    // prefix over the range's per-BSDF shadow lists (k_shade writes one per range)
  __shared__ uint32_t s_next;
  __shared__ uint32_t s_stat[ST_COUNT];
  SortLds& s_sort = *(SortLds*)lds;                                 // borrows the traversal stack's LDS (see k_trace)
  uint32_t* stk_n = lds;
  const uint32_t tid = threadIdx.x;
  if (tid < ST_COUNT) s_stat[tid] = 0;
  uint32_t n_q = 0, n_vis = 0, n_tst = 0;
  for (uint32_t seg0 = blockIdx.x * spb; seg0 < st.n_seg; seg0 += gridDim.x * spb) {
    __syncthreads();
    if (tid == 0) {
      uint32_t acc = 0;
      for (int k = 0; k < kNumShadeQueues - 1; ++k) {
        s_pref[k] = acc;
        if (mt_mask & (1u << k)) acc += st.c_shadow[k * st.n_seg + seg0];
      }
      s_pref[kNumShadeQueues - 1] = acc;
      s_next = 0;
    }
    __syncthreads();
    const uint32_t total = s_pref[kNumShadeQueues - 1];
    const uint32_t slot0 = seg0 * kSeg;
    const uint32_t p1 = s_pref[1], p2 = s_pref[2], p3 = s_pref[3], p4 = s_pref[4];
    auto entry_slot = [&](uint32_t i) -> uint32_t {                 // i-th shadow ray of the range -> slot id
      uint32_t k = (i >= p1 ? 1u : 0u) + (i >= p2 ? 1u : 0u) + (i >= p3 ? 1u : 0u) + (i >= p4 ? 1u : 0u);
      uint32_t base = k == 0 ? 0u : (k == 1 ? p1 : (k == 2 ? p2 : (k == 3 ? p3 : p4)));
      return st.q_shadow[((size_t)k * st.n_seg + seg0) * kSeg + (i - base)];
    };
    if (sc.n_flat > 0) {
      for (uint32_t i = tid; i < total; i += kBlock) {
        uint32_t slot = entry_slot(i);
        float4 ro = st.ray_o[slot], sd = st.sh_d[slot];             // shade advanced ray_o to the hit point = shadow origin (scene.rs:114-117)
        V3 o = v3(ro), dir = v3(sd);
        TraceResult r = traverse_flat<true>(flat_prims, sc.pbox, sc.n_flat, o, dir, sd.w);
        n_q += (uint32_t)__builtin_popcountll(__ballot(true));
        if (COUNT) n_tst += r.tests;
        shadow_resolve(sc, st, slot, o, dir, r);
      }
    } else {
      // ---- ray sort over the range's shadow entries (origin = the path vertex, direction = towards the sampled light point) ----
      constexpr bool sorted = SORTED;
      uint32_t n_live = total;
      if (sorted) {
        for (uint32_t b = tid; b < (uint32_t)kSortBins; b += kBlock) s_sort.hist[b] = 0;
        __syncthreads();
        for (uint32_t i = tid; i < total; i += kBlock) {
          uint32_t sl = entry_slot(i);
          float4 ro = st.ray_o[sl], sd = st.sh_d[sl];
          uint32_t key = ray_key(sc, v3(ro), v3(sd));
          atomicAdd(&s_sort.hist[key], 1u);
          st.sort_key[slot0 + i] = (uint16_t)key;
        }
        __syncthreads();
        n_live = sort_scan(s_sort);
        for (uint32_t i = tid; i < total; i += kBlock) {
          uint32_t sl = entry_slot(i);
          uint32_t key = st.sort_key[slot0 + i];                    // this thread's own store
          uint32_t pos = atomicAdd(&s_sort.hist[key], 1u);
          st.order[slot0 + pos] = (uint16_t)(sl - slot0);
        }
        __threadfence_block();
        __syncthreads();
      }
      const uint16_t* ord = sorted ? st.order + slot0 : nullptr;
      LR_CHUNK_DECL
      if (sorted) LR_CHUNK_FETCH(&s_next, ord, n_live)
      Trav<true> tr;
      bool has = false, fin = false;
      uint32_t slot = 0;
      while (true) {
        if (has && fin) {
          own_box_settle_tree<true>(sc, tr, stk_n);
          TraceResult r; r.t = tr.t; r.prim = tr.prim; r.occluded = tr.occluded; r.visits = tr.visits; r.tests = tr.tests;
          if (COUNT) { n_vis += tr.visits; n_tst += tr.tests; }
          shadow_resolve(sc, st, slot, tr.o, tr.d, r);
          has = false; fin = false;
        }
        bool need = !has;
        bool exhausted, drew = false;
        if (sorted) {
          uint32_t local = 0;
          LR_CHUNK_TAKE(&s_next, ord, n_live, need, drew, local, exhausted)
          if (drew) slot = slot0 + local;
        } else {
          uint32_t idx = wave_reserve(&s_next, need);
          exhausted = __ballot(need && idx >= total) != 0;
          drew = need && idx < total;
          if (drew) slot = entry_slot(idx);
        }
        if (drew) {
          float4 ro = st.ray_o[slot], sd = st.sh_d[slot];
          trav_begin<true>(tr, v3(ro), v3(sd), sd.w);
          has = true;
        }
        n_q += (uint32_t)__builtin_popcountll(__ballot(drew));       // counted per wave (a scalar)
        if (__ballot(has) == 0) break;                              // every drawn entry is a ray: empty wave = list exhausted
        const int thresh = exhausted ? 0 : kRefillBelow;
        bool go = has && !fin;
        do { trav_burst<true>(sc, tr, stk_n, go); } while (__builtin_popcountll(__ballot(go)) > thresh);
        fin = has && !go;
      }
    }
  }
  stat_accumulate(&s_stat[ST_SHADOW], lane_id() == 0 ? n_q : 0u);
  if (COUNT) { stat_accumulate(&s_stat[ST_SHADOW_VISITS], n_vis); stat_accumulate(&s_stat[ST_SHADOW_TESTS], n_tst); }
  __syncthreads();
  stat_flush(st.stats, s_stat);
}

// ==========================================================================================
// Resident pipeline: ONE launch for the whole render.  A workgroup owns 256 path slots from the first camera
// sample to its last retired slot and keeps their whole path state in LDS (six 16-B rows per slot: ray origin,
// direction, throughput, radiance, shadow direction, shadow weight -- the hit record borrows the shadow-weight row
// between trace and shade -- plus seven byte lists: 25.75 KB, six workgroups per CU).  The stages of the streaming
// pipeline survive as PHASES separated by three workgroup barriers per iteration:
//   1 trace all live slots, __ballot / mbcnt compaction into per-BSDF lists, misses onto the finish list
//   2 shade<BSDF> per list (chunks dealt round-robin to the waves), roulette deaths onto the finish list, pool top-up
//   3 shadow list, and on the waves it leaves idle the finish pass (fold, next work item, next camera sample)
// Workgroups never exchange data (only the global item dispenser and the chunk-sum array are shared), so there is
// no grid barrier and no path-state traffic to HBM at all.  Used when the LDS budget allows (flat scenes, shallow
// trees); results are bit-identical to the streaming pipeline (same device functions, same RNG keys, same
// chunk order).
// ==========================================================================================
LR_DEV void pool_step(const DevState& st, PoolLds* pl, uint32_t need, uint32_t batch) {
  uint32_t t0 = pl->taken < pl->a0 ? pl->taken : pl->a0;
  pl->r0 += t0; pl->a0 -= t0;
  uint32_t rest = pl->taken - t0;
  uint32_t t1 = rest < pl->a1 ? rest : pl->a1;
  pl->r1 += t1; pl->a1 -= t1;
  pl->taken = 0;
  if (pl->a0 == 0) { pl->r0 = pl->r1; pl->a0 = pl->a1; pl->a1 = 0; }
  if (pl->a0 < need && pl->a1 == 0) {
    uint32_t cur = __hip_atomic_load(st.next_item, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (cur < st.n_items) {
      // towards the end of the render a workgroup asks for less (DevState::pool_shift), never less than what it is short of now
      const uint32_t fair = (st.n_items - cur) >> st.pool_shift, floor_ = need > pl->a0 ? need - pl->a0 : 1u;
      if (batch > fair) batch = fair > floor_ ? fair : floor_;
      uint32_t nb = atomicAdd(st.next_item, batch);
      if (nb < st.n_items) { pl->r1 = nb; pl->a1 = st.n_items - nb < batch ? st.n_items - nb : batch; }
    }
#ifdef LR_TIMELINE
    else {                                                            // first sight of a dry dispenser: when, and what the workgroup still holds
      unsigned long long* e = st.timeline + 3 * ((size_t)blockIdx.x * (blockDim.x >> 6));
      if (e[1] == 0) { e[1] = __builtin_amdgcn_s_memrealtime(); e[3 * 2 + 1] = 1 + pl->a0 + pl->a1; e[3 * 3 + 1] = 1 + pl->taken; }
    }
#endif
    if (pl->a0 == 0) { pl->r0 = pl->r1; pl->a0 = pl->a1; pl->a1 = 0; }
  }
}

// The lists of one iteration are cut into 64-entry chunks and the chunks of ALL lists are dealt round-robin to the
// four waves (*next_chunk carries the running chunk number from list to list): with one wave per list start, a
// scene with five BSDFs of ~40 hits each would run its whole shade phase on wave 0.
template <int MT, int RB, class LT>
LR_DEV void resident_shade_list(const DevScene& sc, const DevState& st, const DevParams& rp,
                                const LT* list, uint32_t n, LT* shadow_list, uint32_t* shadow_cnt,
                                LT* finish_list, uint32_t* finish_cnt, uint32_t wave, uint32_t lane, uint32_t* next_chunk) {
  const uint32_t chunks = (n + 63u) >> 6;
  for (uint32_t c = (wave - *next_chunk) & (RB / 64 - 1); c < chunks; c += RB / 64) {
    uint32_t i = c * 64u + lane;
    bool valid = i < n;
    uint32_t slot = valid ? list[i] : 0;
    VertexOut v; v.finished = false; v.has_shadow = false; v.L = v3(0, 0, 0); v.g_term = 1.0f; v.pixel = 0; v.sample = 0; v.sky_fetch = false;
    if (valid) v = shade_vertex<MT>(sc, st, rp, slot);
    // a path that ended here (Russian roulette) leaves its final radiance in the slot and queues for the finish pass
    if (v.finished) st.rad[slot] = make_float4(v.L.x, v.L.y, v.L.z, __uint_as_float(v.sample));
    uint32_t fidx = wave_reserve(finish_cnt, v.finished);
    if (v.finished) finish_list[fidx] = (LT)slot;
    uint32_t idx = wave_reserve(shadow_cnt, v.has_shadow);
    if (v.has_shadow) shadow_list[idx] = (LT)slot;
  }
  *next_chunk += chunks;
}

// Slots per resident workgroup = its thread count RB: 256 (4 waves, slot numbers in bytes, 25.75 KB: six workgroups per
// CU) or, for flat scenes whose lists fit, 512 (8 waves, 16-bit slot numbers, <= 53 KB: three per CU -- the same 24 waves).
// A phase cuts each list into 64-entry chunks, and on average half a chunk per list runs on dead lanes: with twice the
// slots per workgroup the lists are twice as long and that waste halves.  Chosen by the host for scenes with several BSDF
// lists (BRDF row: +3.9 %); the Lambert-only headline scene is 2.5 % faster with 256.  Lists are allocated only for the
// BSDFs present (+ shadow + finish).
constexpr int kRSeg = 256;               // the small form; also the unit n_slots is rounded to
LR_DEV constexpr int resident_state_bytes(int rb, int n_lists) { return 6 * rb * 16 + n_lists * rb * (rb > 256 ? 2 : 1); }
inline int resident_lds_bytes(int rb, int n_lists) { return 6 * rb * 16 + n_lists * rb * (rb > 256 ? 2 : 1); }

#ifndef LR_RES_WAVES
#define LR_RES_WAVES 6
#endif
// FLAT: the scene is tested without a tree (n_flat > 0); the instantiation carries no traversal code, so its
// register allocation is not burdened by the 4-wide node step it never runs.
// MTS: the BSDF bodies compiled in (bit k = LR_MAT_* k); the Lambert-only instantiation (the headline scene class)
// allocates registers for one shading body instead of the most demanding of five.
template <bool FLAT, uint32_t MTS, int RB>
__global__ void __launch_bounds__(RB, LR_RES_WAVES) k_resident(DevScene sc, DevState gst, DevParams rp, uint32_t mt_mask, const float4* __restrict__ flat_prims) {
  static_assert(RB == 256 || (RB == 512 && FLAT), "the traversal stack is laid out for 256-thread workgroups");
  typedef typename std::conditional<(RB > 256), uint16_t, uint8_t>::type LT;      // list entry = slot number
  extern __shared__ float4 lds4[];
  __shared__ PoolLds pl;
  __shared__ uint32_t s_cnt2[2][8];        // list lengths, double-buffered by iteration parity: [0..4] shade lists, [6] shadow list, [7] finish list
  __shared__ uint32_t s_retired;
  __shared__ uint32_t s_stat[ST_COUNT];
  __shared__ uint8_t s_qid[kFlatMax];
  DevState st = gst;
  st.ray_o = lds4; st.ray_d = lds4 + RB; st.thr = lds4 + 2 * RB; st.rad = lds4 + 3 * RB;
  st.sh_d = lds4 + 4 * RB; st.sh_w = lds4 + 5 * RB;
  st.acc = gst.acc + (size_t)blockIdx.x * RB;                    // chunk sums are touched once per finished sample: they stay in HBM/L2
  st.hit = nullptr;                                                 // {t, prim} of a slot is kept in sh_w[slot].xy between trace and shade (sh_w is dead then)
  // lists of slot numbers, one per BSDF present in the scene, then the shadow list and the finish list
  LT* lists = (LT*)(lds4 + 6 * RB);
  const uint32_t present = mt_mask & MTS;
  const uint32_t n_b = (uint32_t)__builtin_popcount(present);
  auto lpos = [&](int q) { return (uint32_t)__builtin_popcount(present & ((1u << q) - 1u)); };
  uint32_t* stk_n = (uint32_t*)(lists + (n_b + 2) * RB);
  LT* shq = lists + n_b * RB;
  LT* finq = lists + (n_b + 1) * RB;
  const uint32_t tid = threadIdx.x;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63u;
  if (tid < ST_COUNT) s_stat[tid] = 0;
  if (tid < 16) s_cnt2[tid >> 3][tid & 7] = 0;
  if ((int)tid < sc.n_flat) s_qid[tid] = sc.prim_qid[tid];          // flat scenes: the BSDF id of a hit comes from LDS, not from an L2 round trip
  if (tid == 0) { pl.r0 = pl.a0 = pl.r1 = pl.a1 = pl.taken = 0; pool_step(st, &pl, RB, RB); s_retired = 0; }   // first fill: one item per slot
  __syncthreads();
  for (uint32_t step = 0; step < 1; ++step) {
    bool r = finish_and_regenerate(sc, st, rp, &pl, step * kBlock + tid, false, true, v3(0, 0, 0), 1.0f, 0, 0);
    (void)wave_reserve(&s_retired, r);
  }
  __syncthreads();
  if (tid == 0) pool_step(st, &pl, st.pool_low, st.pool_batch);     // later top-ups happen in phase 2
  uint32_t n_seg = 0, n_shq = 0, n_done = 0, n_sky = 0, parity = 0;
  LR_TL(gst, 0)
#ifdef LR_TIMELINE
  uint32_t tl_iter = 0, tl_iter_dry = 0;
#endif
#ifdef LR_STAMP
  unsigned long long tk[7] = {0, 0, 0, 0, 0, 0, 0}, t_prev = __builtin_amdgcn_s_memtime();
#define LR_TICK(i) { unsigned long long t_now = __builtin_amdgcn_s_memtime(); tk[i] += t_now - t_prev; t_prev = t_now; }
  // what per-slot ready flags could give back of the top barrier: phase 3 stamps every slot it completes (sh_d[slot].x is
  // dead then); after the barrier a wave takes the newest stamp among ITS 64 slots: the part of its wait that came after that is tk[6]
#define LR_READY_STAMP(ok, slot) { if (ok) ((float*)(lds4 + 4 * RB))[4 * (slot)] = __uint_as_float((uint32_t)__builtin_amdgcn_s_memtime() | 1u); }
#else
#define LR_TICK(i)
#define LR_READY_STAMP(ok, slot)
#endif
  while (true) {
#ifdef LR_STAMP
    const unsigned long long t_arrive = __builtin_amdgcn_s_memtime();
#endif
    __syncthreads();                                                // previous iteration (or generation) complete
    LR_TICK(0)
#ifdef LR_STAMP
    {
      const uint32_t stamp = __float_as_uint(((float*)(lds4 + 4 * RB))[4 * tid]);
      ((float*)(lds4 + 4 * RB))[4 * tid] = 0.0f;
      uint32_t age = (stamp & 1u) ? (uint32_t)t_prev - stamp : 0x7fffffffu;        // cycles since the slot's phase-3 work completed
      for (int o = 32; o > 0; o >>= 1) { uint32_t x = (uint32_t)__shfl_xor((int)age, o, 64); age = x < age ? x : age; }
      const uint32_t waited = (uint32_t)(t_prev - t_arrive);
      tk[6] += age < waited ? age : waited;
    }
#endif
    const uint32_t retired = s_retired;
#ifdef LR_TIMELINE
    tl_iter += 1;
    if (tl_iter_dry == 0 && gst.timeline[3 * ((size_t)blockIdx.x * (RB >> 6)) + 1] != 0) tl_iter_dry = tl_iter;      // (a global read per iteration: diagnostic build)
#endif
    if (retired >= (uint32_t)RB) break;                          // wave-uniform
    uint32_t* s_cnt = s_cnt2[parity];
    uint32_t* s_cnt_next = s_cnt2[parity ^ 1u];
    parity ^= 1u;
    // ---- phase 1: closest hit for every live slot, compaction by BSDF ----
    for (uint32_t step = 0; step < 1; ++step) {
      uint32_t slot = step * kBlock + tid;
      bool active = false; int qid = -1;
      float4 ro = st.ray_o[slot];
      if (__float_as_int(ro.w) >= 0) {
        float4 rd = st.ray_d[slot];
        active = true;
        TraceResult r;
        if (FLAT) r = traverse_flat<false>(flat_prims, sc.pbox, sc.n_flat, v3(ro), v3(rd), 0.0f);
        else r = traverse<false>(sc, v3(ro), v3(rd), 0.0f, stk_n, nullptr);
        st.sh_w[slot] = make_float4(r.t, __int_as_float(r.prim), 0.0f, 0.0f);
        qid = r.prim < 0 ? kQMiss : (FLAT ? (int)s_qid[r.prim] : (int)sc.prim_qid[r.prim]);
        n_seg += 1;
      }
      // a ray that left the scene queues for the finish pass of phase 3 (sky lookup, fold, next camera sample).
      // (Finishing dense waves of misses right here was faster only while the work-item pool ran dry in open
      // scenes; with pools sized for the job the dense finish pass wins everywhere.)
      {
        bool miss = active && qid == kQMiss;
        uint32_t fidx = wave_reserve(&s_cnt[7], miss);
        if (miss) { finq[fidx] = (LT)slot; active = false; }
      }
      uint64_t todo = __ballot(active);
      while (todo) {
        int lead = (int)__builtin_ctzll(todo);
        int q = __shfl(qid, lead, 64);
        bool mine = active && qid == q;
        uint32_t idx = wave_reserve(&s_cnt[q], mine);
        if (mine) lists[lpos(q) * RB + idx] = (LT)slot;
        todo &= ~__ballot(mine);
      }
    }
    LR_TICK(2)
    __syncthreads();
    LR_TICK(1)
    // ---- phase 2: one BSDF-specialised body per list ----
    // nobody draws work items in this phase, so the last thread (its wave has the least shade work: the lists fill
    // from wave 0 up) tops the pool up here and the dispenser round trip hides behind the shading
    if (tid == RB - 1) pool_step(st, &pl, st.pool_low, st.pool_batch);   // sized by the host, see DevState
    uint32_t next_chunk = 0;
    // the dearest bodies first (GGX, Blinn-Phong, Phong, refraction), Lambert last: chunks are dealt round-robin across the lists, so the
    // chunks beyond one per wave are then the cheap ones (round 5, BRDF row: 102.3 -> 100.4 ms per 2048 spp, +1.9 %; same film)
    if ((MTS & 8u) && (mt_mask & 8u)) resident_shade_list<3, RB, LT>(sc, st, rp, lists + lpos(3) * RB, s_cnt[3], shq, &s_cnt[6], finq, &s_cnt[7], wave, lane, &next_chunk);
    if ((MTS & 4u) && (mt_mask & 4u)) resident_shade_list<2, RB, LT>(sc, st, rp, lists + lpos(2) * RB, s_cnt[2], shq, &s_cnt[6], finq, &s_cnt[7], wave, lane, &next_chunk);
    if ((MTS & 2u) && (mt_mask & 2u)) resident_shade_list<1, RB, LT>(sc, st, rp, lists + lpos(1) * RB, s_cnt[1], shq, &s_cnt[6], finq, &s_cnt[7], wave, lane, &next_chunk);
    if ((MTS & 16u) && (mt_mask & 16u)) resident_shade_list<4, RB, LT>(sc, st, rp, lists + lpos(4) * RB, s_cnt[4], shq, &s_cnt[6], finq, &s_cnt[7], wave, lane, &next_chunk);
    if ((MTS & 1u) && (mt_mask & 1u)) resident_shade_list<0, RB, LT>(sc, st, rp, lists + lpos(0) * RB, s_cnt[0], shq, &s_cnt[6], finq, &s_cnt[7], wave, lane, &next_chunk);
    LR_TICK(3)
    __syncthreads();
    LR_TICK(5)
    // ---- phase 3: shadow rays of this iteration, and -- on the waves the shadow list leaves idle -- the
    // finish pass: every path that ended in phase 1 (miss) or phase 2 (roulette) is folded into its chunk
    // sum and its slot starts the next camera sample.  Dense waves instead of ~20 % of the lanes of every
    // wave in both earlier phases; the two lists never share a slot.
    const uint32_t nsh = s_cnt[6], nfin = s_cnt[7];
    if (tid < 8) s_cnt_next[tid] = 0;                               // the other parity's counters are idle during this iteration
    const uint32_t wsh = (nsh + 63u) >> 6, wfin = (nfin + 63u) >> 6;
    for (uint32_t v = wave; v < wsh + wfin; v += RB / 64) {
      if (v < wsh) {
        uint32_t i = v * 64u + lane;
        if (i < nsh) {
          uint32_t slot = shq[i];
          float4 ro = st.ray_o[slot];
          float4 sd = st.sh_d[slot];
          V3 o = v3(ro), dir = v3(sd);
          TraceResult r;
          if (FLAT) r = traverse_flat<true>(flat_prims, sc.pbox, sc.n_flat, o, dir, sd.w);
          else r = traverse<true>(sc, o, dir, sd.w, stk_n, nullptr);
          n_shq += 1;
          shadow_resolve(sc, st, slot, o, dir, r);
          LR_READY_STAMP(true, slot)
        }
      } else {
        uint32_t i = (v - wsh) * 64u + lane;
        bool valid = i < nfin;
        uint32_t slot = valid ? finq[i] : 0;
        V3 L = v3(0, 0, 0); float g = 1.0f; uint32_t pixel = 0, sample = 0;
        if (valid) {
          float4 ra = st.rad[slot], th = st.thr[slot], rd = st.ray_d[slot], h = st.sh_w[slot];
          L = v3(ra); sample = __float_as_uint(ra.w); pixel = __float_as_uint(th.w); g = rd.w;
          if (__float_as_int(h.y) < 0) {                               // scene.rs:29 / :43: the ray left the scene
            L = L + v3(th) * sky_radiance(sc, v3(rd));
            if (sc.sky_type == LR_SKY_IBL) n_sky += 1;
          }
          n_done += 1;
        }
        bool rr = finish_and_regenerate(sc, st, rp, &pl, slot, valid, false, L, g, pixel, sample);
        (void)wave_reserve(&s_retired, rr);
        LR_READY_STAMP(valid, slot)
      }
    }
    LR_TICK(4)
  }
#ifdef LR_STAMP
  // diagnostic build only: lane 0 of every wave adds its cycle shares to the tail of the stats buffer
  if (lane_id() == 0) for (int i = 0; i < 7; ++i) atomicAdd(gst.stats + (size_t)kStatShards * kStatStride + i, tk[i]);
#endif
  LR_TL(gst, 2)
#ifdef LR_TIMELINE
  if (tid == 0) gst.timeline[3 * ((size_t)blockIdx.x * (RB >> 6) + 1) + 1] = 1 + (tl_iter_dry ? tl_iter - tl_iter_dry : 0);   // iterations after the dispenser ran dry
#endif
  stat_accumulate(&s_stat[ST_SEGMENTS], n_seg);
  stat_accumulate(&s_stat[ST_SHADOW], n_shq);
  stat_accumulate(&s_stat[ST_SAMPLES], n_done);
  stat_accumulate(&s_stat[ST_SKY], n_sky);
  __syncthreads();
  stat_flush(gst.stats, s_stat);
}

#ifndef LR_TEMPLATE_KERNELS_ONLY
__global__ void __launch_bounds__(kBlock) k_rank_table(DevScene sc, DevState st) {
  uint32_t stride = gridDim.x * kBlock;
  for (uint32_t rank = blockIdx.x * kBlock + threadIdx.x; rank < st.n_pix; rank += stride) st.rank_pixel[rank] = rank_to_pixel(st, sc.cam, rank);
}

__global__ void __launch_bounds__(kBlock) k_resolve(DevScene sc, DevState st, DevParams rp) {
  uint32_t stride = gridDim.x * kBlock;
  for (uint32_t rank = blockIdx.x * kBlock + threadIdx.x; rank < st.n_pix; rank += stride) {
    V3 sum = v3(0, 0, 0);
    for (uint32_t c = 0; c < st.n_chunks; ++c) sum = sum + v3(st.partial[item_index(st, c, rank)]);
    V3 px = sum / (float)rp.spp;                                       // main.rs:104 / :121
    uint32_t pixel = item_pixel(st, sc.cam, rank);
    float* o = st.film + (size_t)pixel * 3;
    o[0] = px.x; o[1] = px.y; o[2] = px.z;
    if (st.packed) { float* q = st.packed + (size_t)rank * 3; q[0] = px.x; q[1] = px.y; q[2] = px.z; }
  }
}

// ---- film output stage on the device (main.rs:171-173 `to_color`, img.rs:40-50 RGBE) -------------------
// mode 0: 8-bit RGB = trunc(clamp(x, 0, 1)^(1/gamma) * 255); mode 1: Radiance RGBE of the linear film.
__global__ void __launch_bounds__(kBlock) k_quantize(const float* film, uint8_t* out, uint32_t n_pix, int mode, float inv_gamma) {
  uint32_t stride = gridDim.x * kBlock;
  for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n_pix; i += stride) {
    float r = film[3 * (size_t)i], g = film[3 * (size_t)i + 1], b = film[3 * (size_t)i + 2];
    if (mode == 0) {
      float v[3] = {r, g, b};
      for (int k = 0; k < 3; ++k) {
        float c = __builtin_fminf(__builtin_fmaxf(v[k], 0.0f), 1.0f);       // f32::max / min: NaN -> 0
        float q = det_pow(c, inv_gamma) * 255.0f;
        out[3 * (size_t)i + k] = !(q > 0.0f) ? 0 : (q >= 255.0f ? 255 : (uint8_t)q);
      }
    } else {
      float mx = __builtin_fmaxf(r, __builtin_fmaxf(g, b));
      uint8_t c[4] = {0, 0, 0, 0};
      if (mx > 0.0f && mx < 3.0e38f) {
        int e; (void)__builtin_frexpf(mx, &e);                            // mx = m * 2^e, m in [0.5, 1)
        float scale = __builtin_ldexpf(1.0f, 8 - e);                        // exact power of two: v / 2^e * 256
        float v[3] = {r, g, b};
        for (int k = 0; k < 3; ++k) {
          float t = __builtin_truncf(v[k] * scale);
          c[k] = t <= 0.0f ? 0 : (t >= 255.0f ? 255 : (uint8_t)t);
        }
        c[3] = (uint8_t)(e + 128);
      }
      out[4 * (size_t)i] = c[0]; out[4 * (size_t)i + 1] = c[1]; out[4 * (size_t)i + 2] = c[2]; out[4 * (size_t)i + 3] = c[3];
    }
  }
}

// exhaustive check of rcp_exact_mid against the IEEE quotient: out[0..1] mismatch counts of the 2- and 3-step
// variants over all finite d with biased exponent in [lo_exp, hi_exp], out[2..3] = one offending bit pattern each
__global__ void k_selftest_rcp(uint32_t lo_exp, uint32_t hi_exp, unsigned long long* out) {
  unsigned long long bad2 = 0, bad3 = 0;
  uint32_t ex2 = 0, ex3 = 0;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (1ull << 32); i += stride) {
    uint32_t bits = (uint32_t)i, ex = (bits >> 23) & 0xffu;
    if (ex < lo_exp || ex > hi_exp) continue;
    float d = __uint_as_float(bits);
    uint32_t want = __float_as_uint(1.0f / d);
    if (__float_as_uint(rcp_exact_mid(d)) != want) { bad2 += 1; ex2 = bits; }
    if (__float_as_uint(rcp_exact_mid3(d)) != want) { bad3 += 1; ex3 = bits; }
  }
  if (bad2) { atomicAdd(out, bad2); out[2] = ex2; }
  if (bad3) { atomicAdd(out + 1, bad3); out[3] = ex3; }
}

// ---- diagnostics kernels (lr_selftest_*) ----------------------------------------------------------
__global__ void k_selftest_math(int fn, const float* a, const float* b, float* out, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float x = a[i], y = b ? b[i] : 0.0f, r = 0.0f, s, c;
  switch (fn) {
    case 0: det_sincos(x, &s, &c); r = s; break;
    case 1: det_sincos(x, &s, &c); r = c; break;
    case 2: r = det_acos(x); break;
    case 3: r = det_atan2(x, y); break;
    case 4: r = det_pow(x, y); break;
    case 5: r = det_exp(x); break;
    case 6: r = det_fmod_pos(x, y); break;
    case 7: r = x / y; break;
    case 8: r = __builtin_sqrtf(x); break;
    case 9: r = checker(x, y); break;
  }
  out[i] = r;
}
__global__ void k_selftest_rng(uint32_t seed, const uint32_t* pixel, const uint32_t* sample, const uint32_t* block, float* out4, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Draw4 d = rng_block(seed, pixel[i], sample[i], block[i]);
  out4[4 * i] = d.v[0]; out4[4 * i + 1] = d.v[1]; out4[4 * i + 2] = d.v[2]; out4[4 * i + 3] = d.v[3];
}
__global__ void __launch_bounds__(kBlock) k_selftest_intersect(DevScene sc, const float4* __restrict__ flat_prims, int stack_depth, const float* origins, const float* dirs, int* prim_out, float* t_out, int n) {
  extern __shared__ uint32_t lds[];
  uint32_t* stk_n = lds;
  int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  V3 o = v3(origins[3 * i], origins[3 * i + 1], origins[3 * i + 2]), d = v3(dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2]);
  TraceResult r = sc.n_flat > 0 ? traverse_flat<false>(flat_prims, sc.pbox, sc.n_flat, o, d, 0.0f) : traverse<false>(sc, o, d, 0.0f, stk_n, nullptr);
  prim_out[i] = r.prim; t_out[i] = r.prim >= 0 ? r.t : 0.0f;
}

// bvh.rs:131-141 with the candidate set widened to every primitive: the definition the tree must reproduce.  The loop
// index is wave-uniform, so the rows arrive by scalar loads; tri_test / sphere_test are the render path's own.
// pbox != null: the DEFINITION of the render path's closest hit (bvh.rs:20-25 + 131-141): a primitive counts only if aabb.rs:74-92
// passes on its own exact box (own_box_exact, per primitive, no shortcut).
__global__ void __launch_bounds__(kBlock) k_selftest_brute(const float4* __restrict__ prims, const float4* __restrict__ pbox, int n_prims, const float* origins, const float* dirs, int* prim_out, float* t_out, int n) {
  int i = blockIdx.x * kBlock + threadIdx.x;
  const bool valid = i < n;
  const int ii = valid ? i : n - 1;
  V3 o = v3(origins[3 * (size_t)ii], origins[3 * (size_t)ii + 1], origins[3 * (size_t)ii + 2]);
  V3 d = v3(dirs[3 * (size_t)ii], dirs[3 * (size_t)ii + 1], dirs[3 * (size_t)ii + 2]);
  float best = 3.0e38f; int bp = -1;
  const ConstRow* rows = (const ConstRow*)prims;
  for (int k = 0; k < n_prims; ++k) {
    float4 q0 = row4(rows[3 * (size_t)k]), q1 = row4(rows[3 * (size_t)k + 1]), q2 = row4(rows[3 * (size_t)k + 2]);
    uint32_t idw = __float_as_uint(q0.w);
    int id = (int)(idw & 0x7fffffffu);
    float t = 0.0f; bool hit;
    if (idw >> 31) hit = sphere_test(v3(q0), q1.y, o, d, &t);
    else hit = tri_test(v3(q0), v3(q1), v3(q2), o, d, &t);
    if (hit && pbox) hit = own_box_exact(row4(((const ConstRow*)pbox)[kRecRows * (size_t)id]), row4(((const ConstRow*)pbox)[kRecRows * (size_t)id + 1]), o, d);
    if (hit && (t < best || (t == best && id < bp))) { best = t; bp = id; }
  }
  if (valid) { prim_out[i] = bp; t_out[i] = bp >= 0 ? best : 0.0f; }
}
__global__ void k_selftest_sky(DevScene sc, const float* dirs, float* rgb, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  V3 c = sky_radiance(sc, v3(dirs[3 * (size_t)i], dirs[3 * (size_t)i + 1], dirs[3 * (size_t)i + 2]));
  rgb[3 * (size_t)i] = c.x; rgb[3 * (size_t)i + 1] = c.y; rgb[3 * (size_t)i + 2] = c.z;
}
// lr_scene_create: checksum of the DECODED map (every texel through sky_texel), compared with the same sum over the caller's floats
__global__ void k_sky_checksum(DevScene sc, uint64_t n, unsigned long long* out) {
  unsigned long long acc = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const float4 t = sky_texel(sc, i);
    acc += ((unsigned long long)__float_as_uint(t.x) + 3ull * __float_as_uint(t.y) + 5ull * __float_as_uint(t.z)) * (2ull * i + 1ull);
  }
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if ((threadIdx.x & 63u) == 0 && acc) atomicAdd(out, acc);
}
// material/*.rs on the device for n inputs of ONE material: in13[13*i..] = out_.xyz, n.xyz, pos.xyz, xi.xyz, fly distance;
// out10[10*i..] = sampled in_.xyz, pdf, brdf(out_, in_).rgb, coef.rgb -- the per-lane dispatch the fused kernels run (material_*_dyn<31>)
__global__ void k_selftest_material(float4 m0, float4 m1, float4 m2, const float* in13, float* out10, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* a = in13 + 13 * (size_t)i;
  Mat m; m.m0 = m0; m.m1 = m1; m.m2 = m2;
  const int mt = (int)__float_as_uint(m0.w);
  V3 out_ = v3(a[0], a[1], a[2]), nrm = v3(a[3], a[4], a[5]), pos = v3(a[6], a[7], a[8]);
  V3 in_ = v3(0, 0, 0); float pdf = 0.0f;
  material_sample_dyn<31u>(mt, m, out_, nrm, a + 9, &in_, &pdf);
  V3 f = material_brdf_dyn<31u>(mt, m, out_, in_, nrm, pos);
  V3 c = material_coef_dyn<31u>(mt, m, out_, nrm, a[12]);
  float* o = out10 + 10 * (size_t)i;
  o[0] = in_.x; o[1] = in_.y; o[2] = in_.z; o[3] = pdf; o[4] = f.x; o[5] = f.y; o[6] = f.z; o[7] = c.x; o[8] = c.y; o[9] = c.z;
}
// camera.rs sample() of the scene's camera: xy[2*i..] = pixel, xi4 = the four draws; out8[8*i..] = origin.xyz, direction.xyz, geometry term, 0
__global__ void k_selftest_camera(DevScene sc, const int* xy, const float* xi4, float* out8, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Draw4 d; for (int k = 0; k < 4; ++k) d.v[k] = xi4[4 * (size_t)i + k];
  V3 o, dir; float g;
  camera_sample(sc.cam, xy[2 * (size_t)i], xy[2 * (size_t)i + 1], d, &o, &dir, &g);
  float* q = out8 + 8 * (size_t)i;
  q[0] = o.x; q[1] = o.y; q[2] = o.z; q[3] = dir.x; q[4] = dir.y; q[5] = dir.z; q[6] = g; q[7] = 0.0f;
}
// Objects::sample_emission (objects.rs:37-51, triangle.rs:140-149, sphere.rs:79-84): xi4 = (-, pick, u, v); out4 = point.xyz, pdf
__global__ void k_selftest_emission_sample(DevScene sc, DevState st, const float* xi4, float* out4, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Draw4 d; for (int k = 0; k < 4; ++k) d.v[k] = xi4[4 * (size_t)i + k];
  V3 p; float pdf;
  sample_emission(sc, st, d, &p, &pdf);
  out4[4 * (size_t)i] = p.x; out4[4 * (size_t)i + 1] = p.y; out4[4 * (size_t)i + 2] = p.z; out4[4 * (size_t)i + 3] = pdf;
}
__global__ void k_selftest_emitter_pick(DevScene sc, const float* xi, int* k_out, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) k_out[i] = emitter_index(sc, sc.emission_area * xi[i]);
}

#endif  // LR_TEMPLATE_KERNELS_ONLY

}  // namespace lr
