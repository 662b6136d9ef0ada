// lr_path.h -- the FUSED pipeline: one persistent launch in which every LANE carries its path from the camera sample to
// its end (scene.rs:78-102,173-193 is one recursion per sample; here one loop per lane).
//
// The streaming and resident pipelines of lr_kernels.h move path state between stages -- through HBM, or through LDS with
// three workgroup barriers per iteration.  Here the state (ray, throughput, radiance, pixel, sample, the pending
// direct-light connection) never leaves the lane's registers, waves never wait for each other, and the stages are plain
// code in the lane's loop:
//
//   finish     main.rs:92-121 in two halves: a lane whose path ended folds its radiance into the chunk sum and installs the
//              camera sample it holds ready in LDS (path_consume); next work item + RNG + camera.rs sample() for every lane
//              that lacks such a spare run as a batch at 20+ of 64 lanes (path_spare_batch; the flat kernel queues two per lane)
//   trace      closest hit of the lane's ray AND, in the same pass over the primitives, the visibility test of the
//              direct-light connection the previous vertex left behind: both rays leave the same point (scene.rs:94-97,
//              112-117), so per triangle  tv = o - p0,  qv = tv x e1  and  e2 . qv  (17 of the 57 operations of
//              triangle.rs:69-100), per sphere  co  and  |co|^2,  and the scalar row loads are computed once for the pair.
//              Every remaining operation keeps its operands and its order: the hits are the same bits.
//   settle     bvh.rs:20-25: the own box of what each of the two rays hit has the last word (lr_kernels.h own_box_unsure: one
//              approximate slab test per winner; the rays it cannot settle are traced again literally)
//   resolve    scene.rs:127-147 for the connection just tested (tree scenes: for the connection whose outcome the lane parked
//              when its walk ended -- settle and resolve run at the lane's next vertex, where the wave's finished rays are densest)
//   shade      scene.rs:153-193 (shade_vertex_core, the same function the other pipelines call)
//
// Same device functions, same RNG keys, same chunk order and the same order of additions into a sample's radiance as the
// other pipelines => bit-identical films (tests/test_gpu_pipelines.py).
//
// Work items come from a per-WAVE pool of four LDS words, topped up by one global atomic per 64 items.  Nothing that is only
// needed at the finish stage is kept in a register across the walk (fresh_s / fresh_v, lr_kernels.h): in a persistent kernel
// every entry-block value is live for the whole render, and what the allocator parks in scratch comes back through the
// vector-memory path at 3000+ cycles a reload.
#pragma once
#include "lr_kernels.h"

#ifndef LR_PUSH_BRANCHFREE
#define LR_PUSH_BRANCHFREE 1
#endif
// Wave priorities by phase (s_setprio; round 4).  The SIMD's arbiter picks among its ready waves by priority, then age.  The bulk
// phase of a kernel -- the node steps of the tree walk, the pass over the primitive rows of a flat scene -- is where a wave spends
// most of its instructions and, in the tree walk, waits for memory between them; the other phases (leaf tests, the vertex, the
// finish stage) are shorter ALU-dense stretches after which the wave goes back to requesting rows.  Letting those go first shortens
// every wave's own serial path without starving anybody (a wave at priority 0 still issues whenever the others wait).
// Measured, interleaved (gpurun_out/r04s .. r04u; retire / node / leaf): 0/0/0 4031 | 3862 Msamples/s on config 4 | 5;
// 1/0/0 4049 | 3906;  1/0/1 4110 | 3953 (+2.0 % | +2.4 %);  2/0/1 4099 | 3948;  1/0/2 4117 | 3943;  2/1/0 3986 | 3877;
// the WALK above the retire point 3949 | 3767.  Flat kernel (finish / trace / vertex): 0/0/0 6019 on configs[1]; 1/0/1 6120 (+1.7 %);
// 1/0/0 6037; 0/1/0 5855; 2/0/1 6115.  Finer splits measure the same or worse (the four row requests of a node step at their own
// priority: 4102 | 3941; the finish stage above / below the vertex: 4108 | 3945 and 4074 | 3898; k_resident's phases: within +-0.6 %).
// The priorities only move instructions in time: films are bit-identical.
#ifndef LR_PRIO_RETIRE
#define LR_PRIO_RETIRE 1
#endif
#ifndef LR_PRIO_NODE
#define LR_PRIO_NODE 0
#endif
#ifndef LR_PRIO_LEAF
#define LR_PRIO_LEAF 1
#endif
#define LR_PRIO_ANY (LR_PRIO_RETIRE || LR_PRIO_NODE || LR_PRIO_LEAF)
// the same for k_path_flat's three phases: finish / trace (the pass over the primitive rows) / vertex
#ifndef LR_PRIO_F_FINISH
#define LR_PRIO_F_FINISH 1
#endif
#ifndef LR_PRIO_F_TRACE
#define LR_PRIO_F_TRACE 0
#endif
#ifndef LR_PRIO_F_VERTEX
#define LR_PRIO_F_VERTEX 1
#endif
#define LR_PRIO_F_ANY (LR_PRIO_F_FINISH || LR_PRIO_F_TRACE || LR_PRIO_F_VERTEX)
#define LR_SETPRIO(P) asm volatile("s_setprio %0" :: "n"(P) : "memory")

namespace lr {

typedef RowVec __attribute__((address_space(3))) LdsRow;

// The lane's path state in the row format shade_vertex_core reads and writes (lr_device.h "Path state").
struct LaneRow { float4 v; LR_DEV float4& operator[](uint32_t) { return v; } };
template <bool LDS_TABLES>
struct LaneStateT {
  LaneRow ray_o, ray_d, thr, rad, sh_d, sh_w;
  const LdsRow* emit;                  // the emitter rows, staged in LDS by the kernel (LDS_TABLES: always; otherwise null = read the scene blob)
};
template <> struct RngLo<LaneStateT<false>> { static constexpr bool value = true; };   // k_path_tree (with NEE: see shade_vertex_core)
constexpr int kEmitLds = 8;            // k_path_tree stages up to this many emitters (the scalar-load case of emitter_index)
template <bool LDS_TABLES>
LR_DEV float4 emit_row(const LaneStateT<LDS_TABLES>& st, const DevScene& sc, int i) {
  if constexpr (LDS_TABLES) return row4(st.emit[i]);
  else if (st.emit) return row4(st.emit[i]);      // (wave-uniform)
  else return sc.emit[i];
}

// LR_DIAG build: wave-uniform phase counters of k_path_tree (calls, lanes, cycles), printed by lr_render
struct PathDiag { unsigned long long cyc_total, cyc_resolve, n_resolve, l_resolve, cyc_vertex, n_vertex, l_vertex, cyc_finish, n_finish, l_finish,
                  cyc_walk, walks, walk_lanes, node_steps, node_lanes, cyc_node, leaf_steps, leaf_lanes, cyc_leaf, leaf_prims, n_batch, l_batch, n_forced, cyc_batch, leaf_prims_max; };

// ---- wave-level work-item pool: a range of reserved item ids in scalar registers ----
// Four LDS words per wave {next reserved item, how many are left, the dispenser has run out, -}: only the spare batch touches
// them, and as registers they would live across the whole walk (the allocator had them in scratch).
constexpr int kPoolWords = 4;

// ---- the pair test: closest hit of (o, d) and in-window visibility of (o, sd, sdist) against one primitive ----
// Wave-level skips of the v and t stages of the pair test when no lane of the wave needs them (they never change a result):
// 2 = both, 1 = the t stage only, 0 = none.  With 64 unrelated rays per wave some lane almost always passes, so the skips
// save nothing and their branches cost: measured on configs[1] 5452 / 5500 / 5554 Msamples/s for 2 / 1 / 0.
#ifndef LR_PAIR_BALLOTS
#define LR_PAIR_BALLOTS 0
#endif
struct PairHit { float t; int prim; float st; int sprim; bool occluded; };
LR_DEV uint64_t lane_mask(bool b) { return __builtin_amdgcn_ballot_w64(b); }    // the compare's own mask as a branch condition

LR_DEV bool sphere_test_co(V3 co, float co2, float r2, V3 d, float* t_out) {     // sphere.rs:42-55 given co = o - c and |co|^2
  // every lane runs every operation (no early return: a wave of 64 unrelated rays never skips as a whole); the same
  // comparisons decide on the same values, so a miss is a miss exactly where sphere.rs:46,51 return None
  float cod = dot(co, d);
  float det = cod * cod - co2 + r2;
  float sq = __builtin_sqrtf(det);
  float t1 = -cod - sq;
  float t2 = -cod + sq;
  *t_out = t1 > kEps ? t1 : t2;
  return bool(!(det <= 0.0f)) & bool(!(bool(t1 < kEps) & bool(t2 < kEps)));
}

LR_DEV void flat_test_pair(float4 q0, float4 q1, float4 q2, V3 o, V3 d, V3 sd, float sdist, bool has_sh, PairHit& r) {
  uint32_t idw = __float_as_uint(q0.w);
  int id = (int)(idw & 0x7fffffffu);
  float tA = 0.0f, tB = 0.0f; bool hitA = false, hitB = false;
  if (idw >> 31) {
    V3 co = o - v3(q0);
    float co2 = sqr_norm(co);
    hitA = sphere_test_co(co, co2, q1.y, d, &tA);
    hitB = sphere_test_co(co, co2, q1.y, sd, &tB) & has_sh;
  } else {
    // triangle.rs:69-100 twice; tv, qv and e2 . qv depend on the origin only
    V3 p0 = v3(q0), e1 = v3(q1), e2 = v3(q2);
    V3 tv = o - p0;
    V3 pvA = cross(d, e2);
    float detA = dot(e1, pvA);
    float invA = rcp_exact_mid(detA);
    float uA = dot(tv, pvA) * invA;
    bool okA = bool(!(__builtin_fabsf(detA) < kEps)) & bool(!(uA < 0.0f)) & bool(!(uA > 1.0f));
    V3 pvB = cross(sd, e2);
    float detB = dot(e1, pvB);
    float invB = rcp_exact_mid(detB);
    float uB = dot(tv, pvB) * invB;
    bool okB = has_sh & bool(!(__builtin_fabsf(detB) < kEps)) & bool(!(uB < 0.0f)) & bool(!(uB > 1.0f));
#if LR_PAIR_BALLOTS >= 2
    if (lane_mask(okA | okB) != 0)
#endif
    {
      V3 qv = cross(tv, e1);
      float vA = dot(d, qv) * invA;
      okA = okA & bool(!(vA < 0.0f)) & bool(!(uA + vA > 1.0f));
      float vB = dot(sd, qv) * invB;
      okB = okB & bool(!(vB < 0.0f)) & bool(!(uB + vB > 1.0f));
#if LR_PAIR_BALLOTS >= 1
      if (lane_mask(okA | okB) != 0)
#endif
      {
        float eq = dot(e2, qv);
        tA = eq * invA;
        hitA = okA & bool(!(tA < kEps));
        tB = eq * invB;
        hitB = okB & bool(!(tB < kEps));
      }
    }
  }
  bool betterA = hitA & bool(tA < r.t);                  // rows come in primitive-id order: the first of equal hits is the lowest id
  r.t = betterA ? tA : r.t;
  r.prim = betterA ? id : r.prim;
  (void)sdist;                                           // the connection's CLOSEST hit, as flat_test<true>: scene.rs:127-131's window
  bool betterB = hitB & bool(tB < r.st);                 // is applied after its own box has been settled (traverse_flat_pair)
  r.st = betterB ? tB : r.st;
  r.sprim = betterB ? id : r.sprim;
}

LR_DEV PairHit traverse_flat_pair(const float4* __restrict__ prims, int n, V3 o, V3 d, V3 sd, float sdist, bool has_sh) {
  PairHit r; r.t = 3.0e38f; r.prim = -1; r.st = 3.0e38f; r.sprim = -1; r.occluded = false;
  const ConstRow* rows = (const ConstRow*)prims;
  for (int k = 0; k < n; k += 2) {
    const ConstRow* nx = rows + 3 * k;
    float4 a0 = row4(nx[0]), a1 = row4(nx[1]), a2 = row4(nx[2]);
    float4 b0 = row4(nx[3]), b1 = row4(nx[4]), b2 = row4(nx[5]);    // unconditional: the array is padded
    asm volatile("" :: "s"(a2.x), "s"(a2.y), "s"(a2.z));
    flat_test_pair(a0, a1, a2, o, d, sd, sdist, has_sh, r);
    if (k + 1 >= n) break;
    flat_test_pair(b0, b1, b2, o, d, sd, sdist, has_sh, r);
  }
  return r;                                               // (both winners still to be settled: k_path_flat)
}

// ---- stages shared by the flat and the tree kernel -------------------------------------------------------------------

// scene.rs:127-147 for the connection the lane carries: (o, dir = sh_d.xyz) reached primitive `sprim` at distance `t`
template <class RecFn>
LR_DEV V3 path_shadow_resolve(V3 L, V3 o, V3 dir, V3 W, bool occluded, int sprim, float t, RecFn rec) {
  if (!occluded && sprim >= 0) {
    V3 pos = o + dir * t;
    float4 sh = rec(sprim, 0), em = rec(sprim, 2);
    uint32_t mw = __float_as_uint(sh.w);
    V3 light_normal = (mw >> 31) ? normalize(pos - v3(sh)) : v3(sh);
    float light_cos = dot(-dir, light_normal);
    if (light_cos > 0.0f) L = L + W * v3(em) * light_cos;
  }
  return L;
}

// Everything a lane keeps besides its LaneState rows; chunk sum, item and chunk end live in LDS (touched once per sample).
struct PathCtl {
  bool has_sh;                         // sh_d / sh_w hold a direct-light connection that is still to be tested
  bool finished;                       // the sample in rad ended (its radiance is complete)
  bool fresh;                          // no work item yet
  bool pend;                           // k_path_tree: the connection's walk is over, its outcome is parked in LDS until the next vertex
  bool lit;                            // k_path_tree: the lane repeats its walk(s) LITERALLY -- own_box_exact behind every primitive test (bvh.rs:20-25) --
                                       // because the own box of what the first walk found rejected the ray (~1e-6 of the rays)
};

#ifndef LR_PATH_WAVES
#define LR_PATH_WAVES 6
#endif
constexpr int kRecStride = 7;          // float4 rows per staged record: the 4 shading rows + the 2 rows of the primitive's own box (+ 1: 112 B, an odd number of
                                       // 16-B bank groups, so consecutive records start in different ones)

// ---- spare camera samples: the finish stage as a dense batch ---------------------------------------------------------
// Paths of a wave end a few at a time (8 of 64 lanes per retire point on the 100k-triangle scene), and fold + next work item +
// camera sample at that density was 17 % of the frame.  So a lane keeps the NEXT sample of its sequence ready in LDS: the
// camera ray's direction and weight (16 B), pixel, work item and sample number (12 B) and, for the thin lens, the point on
// the aperture (8 B) -- the traversal stack gives up 6 of its 16 LDS entries for them (measured: free down to 10).  A lane
// whose path ends folds its radiance and installs its spare (path_consume: a few instructions); spares are produced for every
// lane that lacks one as soon as LR_SPARE_BATCH lanes do, or at once when a path ends without one (path_spare_batch: the pool
// draw, the RNG block and camera.rs sample(), now at 20+ of 64 lanes).  Which lane renders which work item changes; a work
// item's samples are still folded in order by one lane, so the film does not.  (First built with the spares in global memory,
// three 16-B rows: a round trip through the vector-memory path of a tree scene costs 3000+ cycles, -5 %.)
#ifndef LR_SPARE_BATCH
#define LR_SPARE_BATCH 24              // one spare per lane (tree kernels): produce when this many lanes lack theirs
#endif
#ifndef LR_SPARE_BATCH2
#define LR_SPARE_BATCH2 40             // two spares per lane (flat kernel, pinhole): produce when this many lanes hold fewer than two
#endif
// With ONE spare per lane most batches are forced by a lane that ends again before LR_SPARE_BATCH others have, and run at 21-28
// lanes.  The flat kernel has LDS for a SECOND spare per lane (not beside the thin lens' aperture points, and not the tree
// kernels beside their stack): a two-entry queue, the batch adds one entry to every lane that holds fewer than two.
constexpr uint32_t kNoItem = 0xffffffffu;     // spare of a lane that found pool and dispenser empty: consuming it retires the lane
struct SpareLds {
  LdsRow* acc;                                // the sample in flight: chunk sum + work item
  uint32_t* end;                              // chunk end of the NEWEST sample the lane was assigned (in flight or queued)
  LdsRow* dir;                                // the spares [D][kBlock]: camera ray direction, weight
  uint32_t *pix, *item, *smp;                 //            pixel, work item, sample number
  float *lu, *lv;                             //            thin lens: the aperture point in the lens plane (null otherwise)
  uint32_t wave_base;                         // threadIdx.x of the wave's lane 0
  uint32_t* pool;                             // this wave's kPoolWords
};
constexpr size_t kSpareLensBytes = 2 * kBlock * sizeof(float);
// the lane's queue of D spares in one register: entries held (bits 0-1), slot of the oldest (bit 2).  A retired lane stays "full".
template <int D> LR_DEV uint32_t sq_count(uint32_t sq) { return sq & 3u; }
template <int D> LR_DEV uint32_t sq_head(uint32_t sq) { return D == 1 ? 0u : (sq >> 2) & 1u; }
template <int D> LR_DEV uint32_t sq_make(uint32_t count, uint32_t head) { return D == 1 ? count : (count | (head << 2)); }

// main.rs:92-121, first half: which sample follows the newest one the lane was assigned (next of the chunk, or the first of a
// new work item from the wave's pool), its camera ray (camera.rs sample()); appended to the lane's queue.  Whole (converged)
// wave; `want` = the lane holds fewer than D spares.
template <int D, class LS>
LR_DEV void path_spare_batch(const DevScene& sc, const DevState& st, const DevParams& rp, const LS& ls, const PathCtl& c,
                             const SpareLds& sp, bool want, uint32_t& sq) {
  const uint32_t tid = tid_of(sp.wave_base);
  const uint32_t count = sq_count<D>(sq), head = sq_head<D>(sq);
  const uint32_t tail = D == 1 ? 0u : (head + count) & 1u;            // where the new entry goes
  uint32_t pixel = __float_as_uint(ls.thr.v.w), sample = __float_as_uint(ls.rad.v.w);
  uint32_t item = 0, end = 0;
  if (want && !c.fresh) {
    item = __float_as_uint(sp.acc[tid].w); end = sp.end[tid];
    if (D > 1 && count > 0) {                                          // the newest assigned sample is the queued one
      const uint32_t nw = (tail ^ 1u) * kBlock + tid;
      pixel = sp.pix[nw]; sample = sp.smp[nw]; item = sp.item[nw];
    }
  }
  sample += 1u;
  const bool need_item = want && (c.fresh || sample >= end);
  const uint64_t nm = __ballot(need_item);
  bool retired = false;
  if (nm != 0) {
    const uint32_t cnt = (uint32_t)__builtin_popcountll(nm);
    uint32_t r0 = uniform(sp.pool[0]), a0 = uniform(sp.pool[1]), dry = uniform(sp.pool[2]);
    uint32_t r1 = 0, a1 = 0;
    if (cnt > a0 && !dry) {                                         // one trip to the dispenser per pool_batch items: the wave waits for it
      // ... fewer towards the end of the render: at most (items left when this wave last looked) >> pool_shift, so that the last
      // items are not parked in one wave's pool while the lanes of other waves retire (the tail of a render is a fixed cost per
      // call: 1/8 of a frame on each of 8 GPUs pays it in full)
      const uint32_t n_items = st.n_items, seen = uniform(sp.pool[3]);
      uint32_t batch = st.pool_batch;
      const uint32_t fair = (n_items - seen) >> st.pool_shift;
      if (batch > fair) batch = fair;
      const uint32_t ask = batch > cnt - a0 ? batch : cnt - a0;
      uint32_t nb = 0;
      if (lane_id() == 0) nb = atomicAdd(st.next_item, ask);
      r1 = uniform(nb);
      if (r1 < n_items) a1 = n_items - r1 < ask ? n_items - r1 : ask;
      if (a1 < ask) { dry = 1u; LR_TL(st, 1) }
      if (lane_id() == 0) sp.pool[3] = r1 < n_items ? r1 : n_items;
    }
    const uint32_t k = rank_in_mask(nm);
    if (need_item) {
      if (k < a0) item = r0 + k;
      else if (k - a0 < a1) item = r1 + (k - a0);
      else retired = true;
      if (!retired) {
        const ItemRef ir = item_decode(st, item);
        const uint32_t chunk = ir.chunk, rank = ir.rank;
        pixel = st.rank_pixel[rank];
        sample = st.chunk_start[chunk];
        sp.end[tid] = st.chunk_start[chunk + 1];
      }
    }
    // advance the pool by what was handed out (wave-uniform)
    if (cnt < a0) { r0 += cnt; a0 -= cnt; }
    else { const uint32_t u = cnt - a0 < a1 ? cnt - a0 : a1; r0 = r1 + u; a0 = a1 - u; }
    if (lane_id() == 0) { sp.pool[0] = r0; sp.pool[1] = a0; sp.pool[2] = dry; }
  }
  if (want) {
    const uint32_t at = tail * kBlock + tid;
    if (retired) {
      sp.item[at] = kNoItem;
      sq = sq_make<D>((uint32_t)D, head);                              // full: the marker is the last entry the lane will ever get
    } else {
      Draw4 d0 = rng_block(rp.seed, pixel, sample, 0u);
      const uint32_t res_w = (uint32_t)fresh_s(sc.cam.res_w);
      uint32_t y = pixel / res_w, x = pixel - y * res_w;
      V3 o, d; float g, lens[2] = {0.0f, 0.0f};
      camera_sample(sc.cam, (int)x, (int)y, d0, &o, &d, &g, lens);
      sp.dir[at] = (RowVec){d.x, d.y, d.z, g};
      sp.pix[at] = pixel; sp.item[at] = item; sp.smp[at] = sample;
      if (sp.lu) { sp.lu[at] = lens[0]; sp.lv[at] = lens[1]; }
      sq = sq_make<D>(count + 1u, head);
    }
  }
}

// main.rs:92-121, second half, for the lanes in `fin` (their sample ended, or they have no item yet; each holds a spare): fold the
// radiance into the chunk sum, write the sum out if the oldest spare belongs to another work item, install that spare.
template <int D, class LS>
LR_DEV void path_consume(const DevScene& sc, const DevState& st, LS& ls, PathCtl& c, const SpareLds& sp, bool fin, uint32_t& sq, uint32_t* s_stat) {
  const uint32_t tid = tid_of(sp.wave_base);
  stat_count(&s_stat[ST_SAMPLES], __ballot(fin && c.finished));
  if (fin) {
    const uint32_t head = sq_head<D>(sq), at = head * kBlock + tid;
    const uint32_t item2 = sp.item[at];
    V3 sum = v3(0, 0, 0);
    if (c.finished) {
      float4 a = row4(sp.acc[tid]);
      V3 delta = v3(ls.rad.v);
      if (sc.cam.type == LR_CAMERA_THIN_LENS) delta = (delta * ls.ray_d.v.w) * sc.cam.weight2;   // e * (sens / pdf); 1 * 1 for the others
      sum = v3(a) + delta;
      if (item2 != __float_as_uint(a.w)) { st.partial[__float_as_uint(a.w)] = make_float4(sum.x, sum.y, sum.z, 0.0f); sum = v3(0, 0, 0); }
    }
    if (item2 == kNoItem) {
      ls.ray_o.v = make_float4(0, 0, 0, __int_as_float(-1));
      sq = sq_make<D>((uint32_t)D, 0u);                                // retired: never short of spares again
    } else {
      const float4 dg = row4(sp.dir[at]);
      const uint32_t sample = sp.smp[at];
      sp.acc[tid] = (RowVec){sum.x, sum.y, sum.z, __uint_as_float(item2)};
      V3 o = camera_origin(sc.cam, sp.lu ? sp.lu[at] : 0.0f, sp.lv ? sp.lv[at] : 0.0f);
      ls.ray_o.v = make_float4(o.x, o.y, o.z, __int_as_float(0));
      ls.ray_d.v = dg;
      ls.thr.v = make_float4(1.0f, 1.0f, 1.0f, __uint_as_float(sp.pix[at]));
      ls.rad.v = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(sample));
      sq = sq_make<D>(sq_count<D>(sq) - 1u, head ^ 1u);
    }
    c.finished = false; c.fresh = false;
  }
}

// The finish stage of both fused kernels (converged wave): consume, produce spares where the wave is short of them, and
// consume again for the lanes that held none.  `sq` = the lane's queue of D spares.
template <int D, class LS>
LR_DEV void path_finish_spares(const DevScene& sc, const DevState& st, const DevParams& rp, LS& ls, PathCtl& c,
                               const SpareLds& sp, uint32_t& sq, uint32_t* s_stat, PathDiag* dg = nullptr) {
  (void)dg;
  const bool fin1 = (c.finished || c.fresh) && sq_count<D>(sq) != 0u;
  if (__ballot(fin1) != 0) path_consume<D>(sc, st, ls, c, sp, fin1, sq, s_stat);
  const bool unserved = c.finished || c.fresh;
  const uint64_t um = __ballot(unserved);
  const bool want = sq_count<D>(sq) < (uint32_t)D;
  if (um != 0 || (int)__builtin_popcountll(__ballot(want)) >= (D == 1 ? LR_SPARE_BATCH : LR_SPARE_BATCH2)) {
    LR_DIAG_ONLY(unsigned long long tb = __builtin_amdgcn_s_memtime(); if (dg) { dg->n_batch += 1; dg->l_batch += (unsigned)__builtin_popcountll(__ballot(want)); dg->n_forced += um != 0; })
    path_spare_batch<D>(sc, st, rp, ls, c, sp, want, sq);
    LR_DIAG_ONLY(if (dg) dg->cyc_batch += __builtin_amdgcn_s_memtime() - tb;)
    if (um != 0) path_consume<D>(sc, st, ls, c, sp, unserved, sq, s_stat);
  }
  c.finished = false; c.fresh = false;                               // (every such lane was served: nothing of these crosses the walk)
}

// One vertex for the lanes whose ray is done: hit -> scene.rs:153-193, miss -> sky.  `rec(prim, row)` reads the 64-B shading
// record.  Leaves the next ray (and possibly a connection to test) in ls, or c.finished with the final radiance in ls.rad.
template <uint32_t MTS, int NEE = -1, class LS, class RecFn>
LR_DEV void path_vertex(const DevScene& sc, const DevParams& rp, LS& ls, PathCtl& c, bool live, float t, int prim, RecFn rec, uint32_t* s_stat) {
  const bool hitv = live && prim >= 0, miss = live && prim < 0;
  if (hitv) {
    VertexIn in;
    in.ro = ls.ray_o.v; in.rd = ls.ray_d.v; in.th = ls.thr.v; in.ra = ls.rad.v;
    in.h = make_float2(t, __int_as_float(prim));
    in.sh = rec(prim, 0); in.m0 = rec(prim, 1); in.m1 = rec(prim, 2); in.m2 = rec(prim, 3);
    VertexOut v;
    if constexpr ((MTS & (MTS - 1u)) == 0u) {
      constexpr int K = MTS == 1u ? 0 : (MTS == 2u ? 1 : (MTS == 4u ? 2 : (MTS == 8u ? 3 : 4)));
      v = shade_vertex_core<K, 0u, NEE>(sc, ls, rp, 0u, in);
    } else {
      v = shade_vertex_core<kMtDyn, MTS, NEE>(sc, ls, rp, 0u, in, (int)__float_as_uint(in.m0.w));
    }
    c.has_sh = v.has_shadow;
    if (v.finished) { c.finished = true; ls.rad.v = make_float4(v.L.x, v.L.y, v.L.z, in.ra.w); }
  }
  if (__ballot(miss) != 0) {
    if (miss) {                                                      // scene.rs:29 / :43
      V3 L = v3(ls.rad.v) + v3(ls.thr.v) * sky_radiance(sc, v3(ls.ray_d.v));
      ls.rad.v = make_float4(L.x, L.y, L.z, ls.rad.v.w);
      c.finished = true;
    }
    if (sc.sky_type == LR_SKY_IBL) stat_count(&s_stat[ST_SKY], __ballot(miss));
  }
}

// =====================================================================================================================
// Flat scenes (<= kFlatMax primitives): every lane tests every primitive, rows through the scalar cache (traverse_flat).
// Nothing in the loop diverges except the stages' own lane masks.
// =====================================================================================================================
// D: spares per lane (2 unless the camera is a thin lens: its aperture points leave no room for the second set)
template <uint32_t MTS, int D>
__global__ void __launch_bounds__(kBlock, LR_PATH_WAVES) k_path_flat(DevScene sc, DevState st, DevParams rp, const float4* __restrict__ flat_prims) {
  __shared__ RowVec s_rec[kFlatMax * kRecStride];
  __shared__ RowVec s_emit[kFlatMax * 3];
  __shared__ RowVec s_acc[kBlock], s_sdir[D * kBlock];
  __shared__ uint32_t s_end[kBlock], s_spix[D * kBlock], s_sitem[D * kBlock], s_ssmp[D * kBlock];
  __shared__ float s_lens[D == 1 ? 2 * kBlock : 1];
  __shared__ uint32_t s_stat[ST_COUNT], s_pool[kPoolWords * kBlock / 64];
  const uint32_t tid = threadIdx.x;
  const bool lens = D == 1 && sc.cam.type == LR_CAMERA_THIN_LENS;    // (the host picks D = 1 for a thin lens)
  const SpareLds sp = {(LdsRow*)s_acc, s_end, (LdsRow*)s_sdir, s_spix, s_sitem, s_ssmp, lens ? s_lens : nullptr, lens ? s_lens + kBlock : nullptr,
                       uniform(threadIdx.x), s_pool + kPoolWords * (uniform(threadIdx.x) >> 6)};
  if (tid < ST_COUNT) s_stat[tid] = 0;
  if (tid < kPoolWords * kBlock / 64) s_pool[tid] = 0;
  for (uint32_t i = tid; i < (uint32_t)sc.n_flat * 4u; i += kBlock) {
    float4 v = sc.shade[(i >> 2) * kRecRows + (i & 3u)];
    s_rec[(i >> 2) * kRecStride + (i & 3u)] = (RowVec){v.x, v.y, v.z, v.w};
  }
  for (uint32_t i = tid; i < (uint32_t)sc.n_flat * 2u; i += kBlock) {      // rows 4, 5: the primitive's own box (bvh.rs:20-25)
    float4 v = sc.pbox[(i >> 1) * kRecRows + (i & 1u)];
    s_rec[(i >> 1) * kRecStride + 4u + (i & 1u)] = (RowVec){v.x, v.y, v.z, v.w};
  }
  for (uint32_t i = tid; i < (uint32_t)sc.n_emitters * 3u && i < (uint32_t)kFlatMax * 3u; i += kBlock) {
    float4 v = sc.emit[i];
    s_emit[i] = (RowVec){v.x, v.y, v.z, v.w};
  }
  __syncthreads();
  auto rec = [&](int prim, int row) -> float4 { return row4(((const LdsRow*)s_rec)[prim * kRecStride + row]); };

  LaneStateT<true> ls;
  ls.emit = (const LdsRow*)s_emit;
  ls.ray_o.v = make_float4(0, 0, 0, __int_as_float(-1));
  ls.ray_d.v = ls.thr.v = ls.rad.v = ls.sh_d.v = ls.sh_w.v = make_float4(0, 0, 0, 0);
  PathCtl c; c.has_sh = false; c.finished = false; c.fresh = true; c.pend = false; c.lit = false;
  uint32_t sq = 0;                                                   // the lane's queue of spares (sq_count / sq_head)
  LR_DIAG_ONLY(PathDiag dg = {}; const unsigned long long tq0 = __builtin_amdgcn_s_memtime(); unsigned long long tq;)
  LR_TL(st, 0)
  while (true) {
    // ---- finish: fold, install the spare camera sample; new spares where the wave is short of them ----
#if LR_PRIO_F_ANY
    LR_SETPRIO(LR_PRIO_F_FINISH);
#endif
#ifdef LR_DIAG
    tq = __builtin_amdgcn_s_memtime(); dg.n_finish += 1; dg.l_finish += (unsigned)__builtin_popcountll(__ballot(c.finished || c.fresh));
    path_finish_spares<D>(sc, st, rp, ls, c, sp, sq, s_stat, &dg);
    dg.cyc_finish += __builtin_amdgcn_s_memtime() - tq;
#else
    path_finish_spares<D>(sc, st, rp, ls, c, sp, sq, s_stat);
#endif
    const bool live = __float_as_int(ls.ray_o.v.w) >= 0;
    const uint64_t lm = __ballot(live);
    if (lm == 0) break;
    stat_count(&s_stat[ST_SEGMENTS], lm);
    const uint64_t sm = __ballot(live && c.has_sh);
    stat_count(&s_stat[ST_SHADOW], sm);
    // ---- trace: closest hit + the pending connection, one pass over the primitive rows ----
#if LR_PRIO_F_ANY
    LR_SETPRIO(LR_PRIO_F_TRACE);
#endif
    float t = 3.0e38f; int prim = -1;
    LR_DIAG_ONLY(tq = __builtin_amdgcn_s_memtime(); dg.walks += 1; dg.walk_lanes += (unsigned)__builtin_popcountll(lm); dg.n_resolve += 1; dg.l_resolve += (unsigned)__builtin_popcountll(sm);)
    if (live) {
      const V3 o = v3(ls.ray_o.v), d = v3(ls.ray_d.v);
      if (sm != 0) {
        const V3 sd = c.has_sh ? v3(ls.sh_d.v) : d;
        PairHit h = traverse_flat_pair(flat_prims, sc.n_flat, o, d, sd, ls.sh_d.v.w, c.has_sh);
        // bvh.rs:20-25: the own box of each winner has the last word (lr_kernels.h own_box_surely); box rows from LDS
        {
          const int pa = h.prim < 0 ? 0 : h.prim, pb = h.sprim < 0 ? 0 : h.sprim;
          const bool ua = own_box_unsure(rec(pa, 4), rec(pa, 5), h.prim, o, d);
          const bool ub = own_box_unsure(rec(pb, 4), rec(pb, 5), h.sprim, o, sd);
          if (ua | ub) {                                             // cold: ~1e-5 of the rays
#pragma unroll 1
            for (int w = 0; w < 2; ++w) {
              if (w ? ub : ua) {
                float t_; int p_;
                retrace_flat(flat_prims, sc.pbox, sc.n_flat, o, w ? sd : d, t_, p_);
                if (w) { h.st = t_; h.sprim = p_; } else { h.t = t_; h.prim = p_; }
              }
            }
          }
          shadow_window(ls.sh_d.v.w, h.st, h.sprim, h.occluded);
        }
        t = h.t; prim = h.prim;
        if (c.has_sh) {
          V3 L = path_shadow_resolve(v3(ls.rad.v), o, sd, v3(ls.sh_w.v), h.occluded, h.sprim, h.st, rec);
          ls.rad.v = make_float4(L.x, L.y, L.z, ls.rad.v.w);
          c.has_sh = false;
        }
      } else {
        TraceResult r = traverse_flat_raw<false>(flat_prims, sc.n_flat, o, d, 0.0f);
        const int pa = r.prim < 0 ? 0 : r.prim;
        if (own_box_unsure(rec(pa, 4), rec(pa, 5), r.prim, o, d)) retrace_flat(flat_prims, sc.pbox, sc.n_flat, o, d, r.t, r.prim);
        t = r.t; prim = r.prim;
      }
    }
    // ---- vertex: shade the hit or fold the sky ----
#if LR_PRIO_F_ANY
    LR_SETPRIO(LR_PRIO_F_VERTEX);
#endif
    LR_DIAG_ONLY(dg.cyc_walk += __builtin_amdgcn_s_memtime() - tq; tq = __builtin_amdgcn_s_memtime(); dg.n_vertex += 1; dg.l_vertex += (unsigned)__builtin_popcountll(__ballot(live && prim >= 0));)
    path_vertex<MTS>(sc, rp, ls, c, live, t, prim, rec, s_stat);
    LR_DIAG_ONLY(dg.cyc_vertex += __builtin_amdgcn_s_memtime() - tq;)
  }
#ifdef LR_DIAG
  dg.cyc_total = __builtin_amdgcn_s_memtime() - tq0;
  if (lane_id() == 0) {
    unsigned long long* od = st.stats + (size_t)kStatShards * kStatStride + 24;
    const unsigned long long* v = (const unsigned long long*)&dg;
    for (int i = 0; i < (int)(sizeof(PathDiag) / 8); ++i) atomicAdd(od + i, v[i]);
  }
#endif
  LR_TL(st, 2)
  __syncthreads();
  stat_flush(st.stats, s_stat);
}


// =====================================================================================================================
// Tree scenes: the persistent while-while traversal of k_trace, with SHADING AS THE REFILL.  A lane walks the rays of its
// own path -- first the direct-light connection its last vertex left (a visibility query, scene.rs:127-131), then the
// continuation ray -- and whenever at most half of the wave is still walking, the wave stops at a converged RETIRE POINT:
// finished connections are resolved and their lanes start the continuation ray; finished continuation rays are shaded
// right there (scene.rs:153-193) or fold the sky, paths that ended fold into their chunk sum and start the next camera
// sample; every such lane leaves the retire point with a new ray.  No ray, hit, throughput or radiance row ever goes to
// HBM, and an iteration is not three launches but a branch.
// The traversal differs from trav_node / trav_leaf (lr_kernels.h) in two ways only: the query kind is a per-lane flag
// (lanes of one wave walk connections and continuation rays side by side), and the origin is the path's (ls.ray_o).
// =====================================================================================================================
struct PTrav {                         // (the direction and, for a connection, the distance stay in the lane's rows: ray_d / sh_d)
  float ix, iy, iz;
  float t; int prim, cur, sp;
  bool occluded;
};
// the ray a lane is walking: its connection (conn: sh_d = direction, distance) or its continuation ray (ray_d).  `conn` is the
// lane's PathCtl::has_sh: a connection is walked from the retire point that follows its vertex until it is resolved.
template <bool CONN, class LS> LR_DEV V3 ptrav_dir(bool conn, const LS& ls) {
  if constexpr (CONN) return conn ? v3(ls.sh_d.v) : v3(ls.ray_d.v);
  else return v3(ls.ray_d.v);
}

LR_DEV void ptrav_begin(PTrav& s, V3 d) {
  s.t = 3.0e38f; s.prim = -1; s.occluded = false; s.cur = 0; s.sp = 0;
  {
#pragma clang fp contract(fast)
    float dx = __builtin_fabsf(d.x) < 1e-20f ? __builtin_copysignf(1e-20f, d.x) : d.x;     // see trav_begin
    float dy = __builtin_fabsf(d.y) < 1e-20f ? __builtin_copysignf(1e-20f, d.y) : d.y;
    float dz = __builtin_fabsf(d.z) < 1e-20f ? __builtin_copysignf(1e-20f, d.z) : d.z;
    s.ix = __builtin_amdgcn_rcpf(dx); s.iy = __builtin_amdgcn_rcpf(dy); s.iz = __builtin_amdgcn_rcpf(dz);
  }
}
LR_DEV bool ptrav_pop(const DevScene& sc, PTrav& s, const uint32_t* stk_n) {
  if (s.sp > 0) { --s.sp; s.cur = (int)stack_load(sc, stk_n, s.sp); return true; }
  return false;
}
// one 4-wide node (trav_node): boxes only prune, so fused / approximate arithmetic is allowed here
template <bool CONN, class LS>                                    // CONN = false: no lane ever walks a connection (integrator pt)
LR_DEV bool ptrav_node(const DevScene& sc, PTrav& s, const LS& ls, bool conn, uint32_t* stk_n) {
  const V3 o = v3(ls.ray_o.v);
  const float4* n = sc.nodes + kNodeRows * (size_t)s.cur;
  float4 g = n[0], qa = n[1], qb = n[2], rc = n[3];
  float k0, k1, k2, k3;
  int r0 = __float_as_int(rc.x), r1 = __float_as_int(rc.y), r2 = __float_as_int(rc.z), r3 = __float_as_int(rc.w);
  {
#pragma clang fp contract(fast)
    const float inf = __builtin_inff();
    const uint32_t eb = __float_as_uint(g.w);
    const float ax = __uint_as_float((eb & 0xffu) << 23) * s.ix, bx = (g.x - o.x) * s.ix;
    const float ay = __uint_as_float(((eb >> 8) & 0xffu) << 23) * s.iy, by = (g.y - o.y) * s.iy;
    const float az = __uint_as_float(((eb >> 16) & 0xffu) << 23) * s.iz, bz = (g.z - o.z) * s.iz;
    const float bound = ((CONN && conn) ? ls.sh_d.v.w + 2.0f * kEps : s.t) + qb.w;   // as trav_node
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
    if (n_hit == 0) return ptrav_pop(sc, s, stk_n);
#if LR_NODE_ORDER
    nearest_then_slot_order(k0, k1, k2, k3, r0, r1, r2, r3);
#else
    order2(k0, r0, k1, r1); order2(k2, r2, k3, r3); order2(k0, r0, k2, r2); order2(k1, r1, k3, r3); order2(k1, r1, k2, r2);
#endif
#if LR_PUSH_BRANCHFREE
    // the far children go on the stack far-first.  Entries at and above the new top are dead, so while three more fit the LDS
    // part every lane stores three words -- which ones is a select on n_hit -- instead of walking three blocks of predicated
    // stores with an LDS-or-spill branch in each (lanes of one wave differ in n_hit, so the wave used to walk them all)
    if (s.sp + 3 <= sc.stack_lds) {
      lds_u32* e = (lds_u32*)stk_n + s.sp * kBlock + threadIdx.x;
#if LR_NODE_ORDER
      e[0] = (uint32_t)r1; e[kBlock] = (uint32_t)r2; e[2 * kBlock] = (uint32_t)r3;      // hits first, in slot order: no select at all
#else
      e[0] = (uint32_t)(n_hit == 4 ? r3 : (n_hit == 3 ? r2 : r1));
      e[kBlock] = (uint32_t)(n_hit == 4 ? r2 : r1);
      e[2 * kBlock] = (uint32_t)r1;
#endif
    } else
#endif
    if (n_hit == 4) { stack_store(sc, stk_n, s.sp, (uint32_t)r3); stack_store(sc, stk_n, s.sp + 1, (uint32_t)r2); stack_store(sc, stk_n, s.sp + 2, (uint32_t)r1); }
    else if (n_hit == 3) { stack_store(sc, stk_n, s.sp, (uint32_t)r2); stack_store(sc, stk_n, s.sp + 1, (uint32_t)r1); }
    else if (n_hit == 2) { stack_store(sc, stk_n, s.sp, (uint32_t)r1); }
    s.sp += n_hit - 1;
    s.cur = r0;
  }
  return true;
}
// triangle.rs:69-100 without early returns: every lane runs every operation and the same comparisons decide on the same values
#ifndef LR_LEAF_BRANCHFREE
#define LR_LEAF_BRANCHFREE 1
#endif
LR_DEV bool tri_test_bf(V3 p0, V3 e1, V3 e2, V3 o, V3 d, float* t_out) {
  V3 pv = cross(d, e2);
  float det = dot(e1, pv);
  float invdet = rcp_exact_mid(det);
  V3 tv = o - p0;
  float u = dot(tv, pv) * invdet;
  V3 qv = cross(tv, e1);
  float v = dot(d, qv) * invdet;
  float t = dot(e2, qv) * invdet;
  *t_out = t;
  return bool(!(__builtin_fabsf(det) < kEps)) & bool(!(u < 0.0f)) & bool(!(u > 1.0f)) & bool(!(v < 0.0f)) & bool(!(u + v > 1.0f)) & bool(!(t < kEps));
}
// one primitive of a leaf against the lane's ray; true = a connection found its occluder (the walk is over)
// lit (PathCtl::lit, almost never set: a lane-mask skip per primitive.  Choosing between two instantiations of the leaf once per
// leaf step with a wave-uniform flag measured 1-1.4 % SLOWER on configs 4 / 5: the second copy of the leaf cost registers): a
// primitive whose own test accepts counts only if aabb.rs:74-92 passes on its own box (bvh.rs:20-25)
template <bool CONN, class LS>
LR_DEV bool ptrav_prim(const DevScene& sc, PTrav& s, const LS& ls, bool conn, bool lit, V3 o, V3 d, float4 q0, float4 q1, float4 q2) {
  uint32_t idw = __float_as_uint(q0.w);
  int id = (int)(idw & 0x7fffffffu);
  float t; bool hit;
#if LR_LEAF_BRANCHFREE
  if (idw >> 31) { V3 co = o - v3(q0); hit = sphere_test_co(co, sqr_norm(co), q1.y, d, &t); }
  else hit = tri_test_bf(v3(q0), v3(q1), v3(q2), o, d, &t);
#else
  if (idw >> 31) hit = sphere_test(v3(q0), q1.y, o, d, &t);
  else hit = tri_test(v3(q0), v3(q1), v3(q2), o, d, &t);
#endif
  if (!hit) return false;
#if !LR_NO_LIT      // (LR_NO_LIT: measurement only -- what the literal-mode test costs the leaf loop: 1.7 % on config 4, 2.6 % on config 5)
  if (lit) { if (!own_box_exact(sc.pbox[kRecRows * (size_t)id], sc.pbox[kRecRows * (size_t)id + 1], o, d)) return false; }
#endif
  if (CONN && conn) {
    float diff = t - ls.sh_d.v.w;
    if (diff < -kEps) { s.occluded = true; s.t = t; s.prim = id; return true; }   // (the occluder: its own box is settled at the lane's next vertex)
    if (diff > kEps) return false;
  }
  if (t < s.t || (t == s.t && id < s.prim)) { s.t = t; s.prim = id; }
  return false;
}
// one leaf (trav_leaf): the primitive tests decide, exact arithmetic.  The rows of the next primitive are requested before the
// current one is tested; the loop is written two primitives long so that the two row sets swap roles instead of being copied
// (the rotating form spent 19 v_mov per primitive on it, a fifth of the loop)
#ifndef LR_LEAF_UNROLL2
#define LR_LEAF_UNROLL2 1
#endif
template <bool CONN, class LS>
LR_DEV bool ptrav_leaf(const DevScene& sc, PTrav& s, const LS& ls, bool conn, bool lit, const uint32_t* stk_n) {
  const V3 o = v3(ls.ray_o.v), d = ptrav_dir<CONN>(conn, ls);
  uint32_t enc = (uint32_t)~s.cur;
  uint32_t first = enc >> 3, count = enc & 7u;
  const float4* q = sc.prims + 3 * (size_t)first;
#if LR_LEAF_UNROLL2
  float4 a0 = q[0], a1 = q[1], a2 = q[2], b0 = a0, b1 = a1, b2 = a2;
  for (uint32_t k = 0; ; k += 2) {
    const bool more1 = k + 1 < count;
    if (more1) { b0 = q[3 * k + 3]; b1 = q[3 * k + 4]; b2 = q[3 * k + 5]; }
    if (ptrav_prim<CONN>(sc, s, ls, conn, lit, o, d, a0, a1, a2)) return false;
    if (!more1) break;
    const bool more2 = k + 2 < count;
    if (more2) { a0 = q[3 * k + 6]; a1 = q[3 * k + 7]; a2 = q[3 * k + 8]; }
    if (ptrav_prim<CONN>(sc, s, ls, conn, lit, o, d, b0, b1, b2)) return false;
    if (!more2) break;
  }
#else
  float4 n0 = q[0], n1 = q[1], n2 = q[2];
  for (uint32_t k = 0; k < count; ++k) {
    float4 q0 = n0, q1 = n1, q2 = n2;
    if (k + 1 < count) { n0 = q[3 * k + 3]; n1 = q[3 * k + 4]; n2 = q[3 * k + 5]; }
    if (ptrav_prim<CONN>(sc, s, ls, conn, lit, o, d, q0, q1, q2)) return false;
  }
#endif
  return ptrav_pop(sc, s, stk_n);
}
// node steps in a row before the lanes that reached a leaf get their turn: 2 in the pt kernel (7 waves per SIMD; 100k-triangle
// scene 3865 vs 3794 Msamples/s with 3, 3444 with 4), 3 in the pt-direct one (thin-lens / IBL scene 3416 vs 3337 with 2)
#ifndef LR_BURST_PT
#define LR_BURST_PT 2
#endif
#ifndef LR_BURST_NEE
#define LR_BURST_NEE 3
#endif
template <bool CONN, class LS>
LR_DEV void ptrav_burst(const DevScene& sc, PTrav& s, const LS& ls, bool conn, bool lit, uint32_t* stk_n, bool& go, PathDiag* dg = nullptr) {
  (void)dg;
#if LR_PRIO_ANY
  LR_SETPRIO(LR_PRIO_NODE);
#endif
#pragma unroll 1
  for (int it = 0; it < (CONN ? LR_BURST_NEE : LR_BURST_PT); ++it) {
    bool nm = go && s.cur >= 0;
    const uint64_t bm = __ballot(nm);
    if (bm == 0) break;
    LR_DIAG_ONLY(unsigned long long t0 = __builtin_amdgcn_s_memtime();)
    if (nm) go = ptrav_node<CONN>(sc, s, ls, conn, stk_n);
    LR_DIAG_ONLY(dg->node_steps += 1; dg->node_lanes += (unsigned)__builtin_popcountll(bm); dg->cyc_node += __builtin_amdgcn_s_memtime() - t0;)
  }
#ifdef LR_DIAG
  const uint64_t lm = __ballot(go && s.cur < 0);
  if (lm) {
    uint32_t cnt = (go && s.cur < 0) ? ((uint32_t)~s.cur & 7u) : 0u, mx = cnt, sm = cnt;
    for (int off = 32; off > 0; off >>= 1) { mx = max(mx, (uint32_t)__shfl_xor((int)mx, off, 64)); sm += (uint32_t)__shfl_xor((int)sm, off, 64); }
    dg->leaf_prims_max += mx; dg->leaf_prims += sm;
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
#endif
#if LR_PRIO_ANY
  if (LR_PRIO_LEAF != LR_PRIO_NODE) LR_SETPRIO(LR_PRIO_LEAF);
#endif
  if (go && s.cur < 0) go = ptrav_leaf<CONN>(sc, s, ls, conn, lit, stk_n);
  LR_DIAG_ONLY(if (lm) { dg->leaf_steps += 1; dg->leaf_lanes += (unsigned)__builtin_popcountll(lm); dg->cyc_leaf += __builtin_amdgcn_s_memtime() - t1; })
}

#ifndef LR_FAKE_SETTLE
#define LR_FAKE_SETTLE 0
#endif
#ifndef LR_NO_LIT
#define LR_NO_LIT 0
#endif
// the outcome of a connection's walk in one word: 0 = nothing in the window, w + 1 = primitive w hit inside it, -(x + 1) = occluded by x
LR_DEV uint32_t conn_word(const PTrav& s) { return s.prim < 0 ? 0u : (s.occluded ? (uint32_t)-(s.prim + 1) : (uint32_t)(s.prim + 1)); }

#ifndef LR_RETIRE_EIGHTHS
#define LR_RETIRE_EIGHTHS 4            // the walk stops for a retire point when this many eighths of the wave's rays are still under way
#endif
#ifndef LR_PATHT_WAVES
#define LR_PATHT_WAVES 6
#endif
#ifndef LR_PATHT_WAVES_PT
#define LR_PATHT_WAVES_PT 7
#endif
// waves per SIMD of k_path_tree: the pt instantiation carries no connection code and fits 72 VGPRs (48 B of scratch): 7 waves
// render the 100k-triangle scene 1.8 % faster than 6 (8 do not fit the LDS); the pt-direct one is best at 6 (80 VGPRs)
LR_DEV constexpr int path_tree_waves_c(bool nee) { return nee ? LR_PATHT_WAVES : LR_PATHT_WAVES_PT; }
inline int path_tree_waves(bool nee) { return nee ? LR_PATHT_WAVES : LR_PATHT_WAVES_PT; }

// NEE: the integrator is pt-direct (connections are produced, walked and resolved) or pt (none of that code exists)
template <uint32_t MTS, bool NEE>
__global__ void __launch_bounds__(kBlock, path_tree_waves_c(NEE)) k_path_tree(DevScene sc, DevState st, DevParams rp) {
  extern __shared__ uint32_t lds[];                                  // traversal stack: sc.stack_lds entries per lane; thin lens: + kSpareLensBytes
  __shared__ RowVec s_acc[kBlock], s_sdir[kBlock];
  __shared__ uint32_t s_end[kBlock], s_spix[kBlock], s_sitem[kBlock], s_ssmp[kBlock];
  __shared__ uint32_t s_stat[ST_COUNT], s_pool[kPoolWords * kBlock / 64];
  __shared__ RowVec s_emit[NEE ? kEmitLds * 3 : 1];
  __shared__ uint32_t s_conn[NEE ? kBlock : 1];                      // the lane's parked connection (PathCtl::pend): conn_word
  uint32_t* stk_n = lds;
  const uint32_t tid = threadIdx.x;
  float* s_lens = sc.cam.type == LR_CAMERA_THIN_LENS ? (float*)(lds + (size_t)sc.stack_lds * kBlock) : nullptr;
  const SpareLds sp = {(LdsRow*)s_acc, s_end, (LdsRow*)s_sdir, s_spix, s_sitem, s_ssmp, s_lens, s_lens ? s_lens + kBlock : nullptr, uniform(threadIdx.x),
                       s_pool + kPoolWords * (uniform(threadIdx.x) >> 6)};
  if (tid < ST_COUNT) s_stat[tid] = 0;
  if (tid < kPoolWords * kBlock / 64) s_pool[tid] = 0;
  // a few emitters (area lights): their rows in LDS -- from the scene blob the vertex waited for them twice (the rows both
  // primitive classes read, then the triangle's other two), two round trips of the four a pt-direct vertex made
  const bool emit_lds = NEE && sc.n_emitters <= kEmitLds;
  if (emit_lds && tid < (uint32_t)sc.n_emitters * 3u) { float4 v = sc.emit[tid]; s_emit[tid] = (RowVec){v.x, v.y, v.z, v.w}; }
  __syncthreads();
  auto rec = [&](int prim, int row) -> float4 { return sc.shade[kRecRows * (size_t)prim + row]; };

  LaneStateT<false> ls;
  ls.emit = emit_lds ? (const LdsRow*)s_emit : nullptr;
  ls.ray_o.v = make_float4(0, 0, 0, __int_as_float(-1));
  ls.ray_d.v = ls.thr.v = ls.rad.v = ls.sh_d.v = ls.sh_w.v = make_float4(0, 0, 0, 0);
  PathCtl c; c.has_sh = false; c.finished = false; c.fresh = true; c.pend = false; c.lit = false;
  PTrav tr; ptrav_begin(tr, v3(1.0f, 0.0f, 0.0f));
  bool go = false;                                                   // the lane's walk is under way
  uint32_t sq = 0;                                                   // the lane's queue of one spare
  LR_DIAG_ONLY(PathDiag dg = {}; const unsigned long long tq0 = __builtin_amdgcn_s_memtime(); unsigned long long tq;)
  LR_TL(st, 0)
  while (true) {
    // ================= retire point (converged) =================
    // A lane with depth >= 0 has a ray; without `go` its walk is over.  (Few flags cross the walk: go, the spare count, has_sh, occluded.)
    // (a) connections whose walk is over: the outcome is PARKED (one word in LDS) and the lane starts its continuation ray.  It is
    // settled and resolved (scene.rs:127-147) at the lane's next vertex, where the wave's finished rays are densest: resolving the
    // one or two connections that end per burst on the spot ran ~70 instructions at 1-3 lanes, and the own-box test (bvh.rs:20-25)
    // would have doubled that.  The radiance still receives the connection before anything the next vertex adds: same sums.
    if constexpr (NEE) {
      const bool fs = __float_as_int(ls.ray_o.v.w) >= 0 && !go && c.has_sh;
      if (fs) {
        s_conn[threadIdx.x] = conn_word(tr);
        c.has_sh = false; c.pend = true;
        ptrav_begin(tr, v3(ls.ray_d.v));
        go = true;
      }
    }
    // (b) continuation rays whose walk is over: the own boxes of what they and the parked connection hit (bvh.rs:20-25), the
    // connection's radiance, then the vertex
    const bool fm = __float_as_int(ls.ray_o.v.w) >= 0 && !go;         // (a) left only closest-hit walks among these
    if (__ballot(fm) != 0) {
      LR_DIAG_ONLY(tq = __builtin_amdgcn_s_memtime(); if (!NEE) { dg.n_resolve += 1; dg.l_resolve += (unsigned)__builtin_popcountll(__ballot(fm && tr.prim >= 0)); })
      // ---- settle (lr_kernels.h own_box_unsure): is each winner certainly a candidate under bvh.rs:20-25?  Straight-line code.  (The
      // shading rows of the vertex are NOT requested up here with the own-box rows: measured 2.3 % slower on configs 4 and 5 --
      // 16 more registers live across the settle arithmetic cost more than the second round trip.)  A lane in literal mode has been
      // through this: its walks tested every primitive's own box themselves ----
      const V3 o = v3(ls.ray_o.v);
      const bool chk = fm && !c.lit;
      const bool hitv = chk && tr.prim >= 0;
      const size_t pr = (fm && tr.prim >= 0) ? (size_t)tr.prim : 0;
#if LR_FAKE_SETTLE      // measurement only (wrong results): the settle arithmetic on rows the vertex requests anyway -- what the own-box rows' round trip costs
      const float4 blo = sc.shade[kRecRows * pr + 1], bhi = sc.shade[kRecRows * pr + 2];
#else
      const float4 blo = sc.pbox[kRecRows * pr], bhi = sc.pbox[kRecRows * pr + 1];
#endif
      bool pm = false, cocc = false; int cp = -1; float ct = 0.0f;
      bool uc = false;
      if constexpr (NEE) {
        pm = fm && c.pend;
        LR_DIAG_ONLY(dg.n_resolve += 1; dg.l_resolve += (unsigned)__builtin_popcountll(__ballot(pm));)
        if (pm) { const int w = (int)s_conn[threadIdx.x]; cocc = w < 0; cp = (w < 0 ? -w : w) - 1; }
        const size_t pc = cp >= 0 ? (size_t)cp : 0;
        const float4 clo = sc.pbox[kRecRows * pc], chi = sc.pbox[kRecRows * pc + 1];
        const float4 csh = sc.shade[kRecRows * pc], cem = sc.shade[kRecRows * pc + 2];
        uc = own_box_unsure(clo, chi, (pm && chk) ? cp : -1, o, v3(ls.sh_d.v));
        // ---- the parked connection's radiance (scene.rs:127-147), unless its own box is in doubt ----
        const bool rm = pm && !uc;
        if (rm) {
          // the distance is not parked: only a sphere's normal needs it (scene.rs:135, sphere.rs:57-62), and the sphere's own test gives it
          // again, the same bits (centre from the shading record, r^2 from the box rows)
          if (cp >= 0 && !cocc && (__float_as_uint(csh.w) >> 31)) { const V3 co = o - v3(csh); (void)sphere_test_co(co, sqr_norm(co), clo.w, v3(ls.sh_d.v), &ct); }
          auto crec = [&](int, int row) -> float4 { return row == 0 ? csh : cem; };
          V3 L = path_shadow_resolve(v3(ls.rad.v), o, v3(ls.sh_d.v), v3(ls.sh_w.v), cocc, cp, ct, crec);
          ls.rad.v = make_float4(L.x, L.y, L.z, ls.rad.v.w);
          c.pend = false;
        }
      }
#if LR_FAKE_SETTLE
      bool ur = own_box_unsure(blo, bhi, hitv ? tr.prim : -1, o, v3(ls.ray_d.v));
      { int keep = ur ? 1 : 0; asm volatile("" :: "v"(keep)); ur = false; }
#else
      const bool ur = own_box_unsure(blo, bhi, hitv ? tr.prim : -1, o, v3(ls.ray_d.v));
#endif
      // ---- a winner in doubt (~1e-3 of the hits on unit-sized triangles, ~1e-5 on walls): the lane walks again among the others,
      // LITERALLY (ptrav_prim tests every primitive's own box).  A connection in doubt is walked first, the continuation ray after
      // it as always.  (Handled behind the vertex of the other lanes: no branch between the row requests and their use.) ----
      const bool redo = ur | uc;
      const bool vm = fm && !redo;                                   // the lanes whose vertex runs now
      LR_DIAG_ONLY(dg.cyc_resolve += __builtin_amdgcn_s_memtime() - tq; tq = __builtin_amdgcn_s_memtime(); dg.n_vertex += 1; dg.l_vertex += (unsigned)__builtin_popcountll(__ballot(vm));)
      path_vertex<MTS, NEE ? 1 : 0>(sc, rp, ls, c, vm, tr.t, tr.prim, rec, s_stat);
      c.lit = c.lit && !vm;
      if (redo) {
        if (NEE && uc) { c.has_sh = true; c.pend = false; ptrav_begin(tr, v3(ls.sh_d.v)); }
        else ptrav_begin(tr, v3(ls.ray_d.v));
        c.lit = true; go = true;
      }
      LR_DIAG_ONLY(dg.cyc_vertex += __builtin_amdgcn_s_memtime() - tq;)
    }
    // (c) paths that ended (or lanes without a work item yet): fold, next item, next camera sample
    {

      LR_DIAG_ONLY(tq = __builtin_amdgcn_s_memtime(); dg.n_finish += 1; dg.l_finish += (unsigned)__builtin_popcountll(__ballot(c.finished || c.fresh));)
#ifdef LR_DIAG
      path_finish_spares<1>(sc, st, rp, ls, c, sp, sq, s_stat, &dg);
#else
      path_finish_spares<1>(sc, st, rp, ls, c, sp, sq, s_stat);
#endif
      LR_DIAG_ONLY(dg.cyc_finish += __builtin_amdgcn_s_memtime() - tq;)
    }
    // (d) every lane with a path and no walk under way starts its next ray: the connection first, if its vertex left one
    const bool live = __float_as_int(ls.ray_o.v.w) >= 0;
    {
      const bool start = live && !go;
      stat_count(&s_stat[ST_SEGMENTS], __ballot(start));
      if constexpr (NEE) stat_count(&s_stat[ST_SHADOW], __ballot(start && c.has_sh));
      if (start) {
        if (NEE && c.has_sh) ptrav_begin(tr, v3(ls.sh_d.v));
        else ptrav_begin(tr, v3(ls.ray_d.v));
        go = true;
      }
    }
    const uint64_t hm = __ballot(live);
    if (hm == 0) break;
    // ================= walk until at most half of the wave's rays are still under way =================
    const int live_n = __builtin_popcountll(hm);
    const int thresh = live_n * LR_RETIRE_EIGHTHS / 8;
#ifdef LR_DIAG
    tq = __builtin_amdgcn_s_memtime(); dg.walks += 1; dg.walk_lanes += (unsigned)__builtin_popcountll(__ballot(go));
#endif
    do {
#ifdef LR_DIAG
      ptrav_burst<NEE>(sc, tr, ls, NEE && c.has_sh, c.lit, stk_n, go, &dg);
#else
      ptrav_burst<NEE>(sc, tr, ls, NEE && c.has_sh, c.lit, stk_n, go);
#endif
      if constexpr (NEE) {
        // a connection whose walk is over does not wait for the retire point: its outcome is parked and the lane walks on with the
        // continuation ray (see (a))
        const bool dc = live && !go && c.has_sh;
        if (dc) {
          s_conn[threadIdx.x] = conn_word(tr);
          c.has_sh = false; c.pend = true;
          ptrav_begin(tr, v3(ls.ray_d.v));
          go = true;
        }
      }
    } while (__builtin_popcountll(__ballot(go)) > thresh);
#if LR_PRIO_ANY
    LR_SETPRIO(LR_PRIO_RETIRE);
#endif
    LR_DIAG_ONLY(dg.cyc_walk += __builtin_amdgcn_s_memtime() - tq;)
  }
#ifdef LR_DIAG
  dg.cyc_total = __builtin_amdgcn_s_memtime() - tq0;
  if (lane_id() == 0) {
    unsigned long long* od = st.stats + (size_t)kStatShards * kStatStride + 24;
    const unsigned long long* v = (const unsigned long long*)&dg;
    for (int i = 0; i < (int)(sizeof(PathDiag) / 8); ++i) atomicAdd(od + i, v[i]);
  }
#endif
  LR_TL(st, 2)
  __syncthreads();
  stat_flush(st.stats, s_stat);
}

}  // namespace lr
