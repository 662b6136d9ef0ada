// lr_device.h -- HBM layouts shared by the kernels and the C ABI implementation.
//
// Scene blob (read-only, replicated per GPU; DESIGN.md "data layout in HBM"):
//   nodes   4 x float4 per 4-wide BVH node (64 B: two nodes per 128-B cache line), child boxes QUANTISED to 8 bits per
//           plane on a power-of-two grid anchored at the node's lower corner:
//             {origin.xyz, ex | ey << 8 | ez << 16}       e* = IEEE biased exponents of the grid steps 2^(e - 127)
//             {qlo.x of child 0..3 (one byte each), qlo.y, qlo.z, qhi.x}
//             {qhi.y, qhi.z, 2 kappa, kappa * diagonal}     the distance-culling slack of this node's children (lr_scene_create, Wide4Builder)
//             {child 0..3 as int bits}
//           plane = origin + q * step; lower planes are rounded down, upper planes up, so a stored box contains the
//           padded f32 box of the description.  child >= 0 inner node, child < 0 leaf ~c = first<<3 | count,
//           0x7fffffff = empty slot.  lr_scene_create collapses the binary tree of the description (or of the device
//           LBVH build) into this form: one fetch per two binary levels.
//   prims   3 x float4 per primitive (48 B) IN LEAF ORDER, so a leaf reads consecutive rows:
//             triangle  {p0.xyz, id} {e1.xyz, -} {e2.xyz, -}     e1 = p1-p0, e2 = p2-p0 (triangle.rs:71-72)
//             sphere    {c.xyz, id | 1<<31} {r, r*r, -, -} {-}
//   shade / pbox   ONE 128-B record per primitive id (kRecRows = 8 rows: one cache line, so the own-box rows the settle stage asks for
//           bring the shading rows of the vertex that follows into L1 -- two arrays meant two lines and two full round trips per vertex):
//             rows 0-3 (DevScene::shade points at row 0; the four loads of a vertex go out together):
//               {n.xyz | c.xyz, material | sphere<<31}   (flat normal, triangle.rs:36)
//               {color.rgb, type} {emission.rgb, weight} {param0..2, -}    = the primitive's material, denormalised
//             rows 4-5 (DevScene::pbox points at row 4): the primitive's OWN exact box {min.xyz, r^2 of a sphere} {max.xyz, -} as
//               triangle.rs:102-118 / sphere.rs:31-38 compute it.  bvh.rs:20-25 makes a primitive a candidate only if aabb.rs:74-92
//               passes on it: read once per query for the primitive that decides it (lr_kernels.h own_box_surely), and per primitive
//               by the literal re-trace of the few undecided rays
//             rows 6-7 unused
//   emit    3 x float4 per emitter (objects.rs:19-24, instance order):
//             {p0|c .xyz, type} {p1.xyz | r, pdf} {p2.xyz, cumulative area}
//   texels  float4 per IBL texel (rgb, -); or, when every texel of the map is a Radiance RGBE value (c * 2^(e - 136), the
//           `image` crate's decode, sky.rs:45-48) -- true of any map that was loaded from an .hdr file -- the RGBE word itself,
//           4 B per texel, decoded on the fly to the SAME f32 bits (texels_rgbe; lr_scene_create re-encodes, checks every texel
//           and the device decode of the whole map, and keeps float4 when one fails): 75 MB -> 19 MB for a 3072 x 1536 map
//
// Path state (SoA, one entry per resident path slot; every slot always carries a live path because
// a finished path regenerates the next sample in place):
//   ray_o {o.xyz, depth as int (-1 = slot retired)}   ray_d {d.xyz, camera g_term}
//   hit   {t, primitive id as int (-1 = miss)}
//   thr   {throughput.rgb, pixel index}               rad {radiance of this sample.rgb, sample index}
//   acc   {sum over the current chunk.rgb, work item}
//   sh_d  {shadow dir.xyz, distance to the light point} sh_w {weight.rgb, -}
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/lumilly_hip.h"

namespace lr {

constexpr int kBlock = 256;            // 4 waves of 64
constexpr int kStackLdsFused = 10;     // ... by k_path_tree, whose LDS also holds the spare camera samples (lr_path.h)
constexpr int kStackLdsMax = 16;       // traversal stack entries per lane kept in LDS by the streaming kernels; near-first order rarely goes deeper
                                       // (12 / 16 / 25 entries render the 100k-triangle configs at the same speed), the rest of the worst case spills
constexpr int kNodeRows = 4;         // float4 rows per 4-wide node (64 B)
constexpr int kRecRows = 8;          // float4 rows per primitive RECORD (128 B = one cache line): rows 0-3 shading, 4-5 the own box, 6-7 unused
constexpr int kQMiss = 5;              // queue ids 0..4 = LR_MAT_*, 5 = miss
constexpr int kNumShadeQueues = 6;
constexpr int kFlatMax = 32;          // scenes up to this many primitives skip the tree
constexpr int kSeg = 512;              // path slots per segment (queue / pool / counter granularity)
constexpr int kStatShards = 64;        // statistics are sharded over 64 cache lines
constexpr int kStatStride = 16;        // u64 words per shard (128 B)

enum StatSlot { ST_SAMPLES = 0, ST_SEGMENTS, ST_SHADOW, ST_NODE_VISITS, ST_PRIM_TESTS, ST_SHADOW_VISITS, ST_SHADOW_TESTS, ST_SKY, ST_COUNT };

struct DevCamera {
  int type; int res_w, res_h;
  float forward[3], right[3], up[3], position[3], aperture_position[3];
  float sensor_w, sensor_h;
  float aperture_sensor_distance, aperture_radius, focus_distance, sensor_pixel_area;
  float weight2;                       // sensor_sensitivity / (sensor pdf * aperture pdf)  (main.rs:101,118)
};

struct DevScene {
  const float4* nodes;
  const float4* prims;
  const float4* shade;                 // row 0 of the 128-B primitive records (kRecRows rows apart)
  const float4* pbox;                  // row 4 of the same records: the primitive's own exact box {min.xyz, r^2 of a sphere} {max.xyz, -} (bvh.rs:20-25 decides with it)
  const float4* emit;
  const float4* texels;
  const uint32_t* texels_rgbe;         // non-null: the map as RGBE words r | g << 8 | b << 16 | e << 24 (e >= 10: every value normal or zero), texels unused
  const uint8_t* prim_qid;             // per primitive id: shade queue (= material type)
  int   n_flat;                        // > 0: test all n_flat primitives with scalar loads instead of walking the tree
  // traversal stack of one lane: entries [0, stack_lds) in LDS (entry e of thread t at [e * 256 + t]), deeper ones --
  // rare with near-first order -- in stack_spill (HBM/L2, [(block * spill_depth + e - stack_lds) * 256 + t]).  The
  // worst case of a 4-wide tree (3 pushes per level) would otherwise cap a CU at 3 workgroups.  Set per launch.
  int   stack_lds, spill_depth;
  uint32_t* stack_spill;
  float key_lo[3], key_scale[3];       // ray sort: cell = (origin - key_lo) * key_scale, 4 cells per axis over the scene's bounding box
  int   n_emitters;
  float emission_area;
  int   sky_type;
  float sky_color[3];
  int   sky_h;
  float sky_lon;
  DevCamera cam;
};

struct DevParams {
  int integrator, spp; uint32_t seed; int depth, depth_limit, no_direct_emitter;
};

// A column of the path state.  Indexing yields a proxy whose load / store carry the non-temporal hint when the library
// is built with -DLR_STATE_NT: the state (136 B x up to 32 M slots) streams through once per stage and should not evict
// the BVH rows and shading records from L2.
template <class T>
struct StateArr {
  T* p;
  StateArr() = default;
  __host__ __device__ StateArr(T* q) : p(q) {}
  __host__ __device__ explicit operator bool() const { return p != nullptr; }
  __host__ __device__ StateArr& operator+=(size_t n) { p += n; return *this; }
  __host__ __device__ StateArr operator+(size_t n) const { return StateArr(p + n); }
  typedef float native_t __attribute__((ext_vector_type(sizeof(T) / 4)));
  struct Ref {
    T* q;
    __device__ operator T() const {
#ifdef LR_STATE_NT
      native_t v = __builtin_nontemporal_load(reinterpret_cast<const native_t*>(q));
      T r; __builtin_memcpy(&r, &v, sizeof(T)); return r;
#else
      return *q;
#endif
    }
    __device__ void operator=(const T& v) const {
#ifdef LR_STATE_NT
      native_t n; __builtin_memcpy(&n, &v, sizeof(T));
      __builtin_nontemporal_store(n, reinterpret_cast<native_t*>(q));
#else
      *q = v;
#endif
    }
  };
  __device__ Ref operator[](size_t i) const { return Ref{p + i}; }
};

struct DevState {
  StateArr<float4> ray_o, ray_d; StateArr<float2> hit;
  StateArr<float4> thr, rad, acc;
  StateArr<float4> sh_d, sh_w;
  uint32_t* q_shade;                   // [queue][segment][kSeg] slot ids, written by k_trace
  uint32_t* c_shade;                   // [queue][segment] counts
  uint32_t* q_shadow;                  // [bsdf][segment][kSeg] slot ids with a pending shadow ray, written by k_shade<bsdf>
  uint32_t* c_shadow;                  // [bsdf][segment] counts
  uint16_t* sort_key;                  // [slot] scratch of the ray sort (k_trace / k_shadow): the bin of the i-th entry of a range
  uint16_t* order;                     // [slot] sorted order of a range's rays as local slot numbers (< 16384); null = no sort
  uint4*    pool;                      // per-segment work-item pool {r0 next, r0 end, r1 next, r1 end}
  uint32_t* next_item;                 // global work-item dispenser
  uint32_t* n_retired;                 // slots that found pool and dispenser empty
  unsigned long long* stats;           // kStatShards * kStatStride
  float4* partial;                     // n_items chunk sums
  float*  film;                        // W*H*3
  float*  packed;                      // n_pix*3 in pixel-rank order, or null (lr_render's read-back source)
  const int4* tiles;                   // x0, y0, w, h
  const uint32_t* tile_prefix;         // n_tiles + 1
  uint32_t* rank_pixel;                // n_pix: film pixel of every pixel rank (k_rank_table), one load instead of a search per work item
  int n_tiles;
  uint32_t n_slots, n_seg, n_pix, n_chunks, n_items;
  // a pixel's samples [chunk_start[c], chunk_start[c + 1]) form chunk c (n_chunks + 1 entries; lumilly_hip.hip chunk_schedule: a
  // function of spp only -- long chunks first, a taper of short ones at the end of the render)
  const uint32_t* chunk_start;
  // Order of the work items of a launch (lr_kernels.h item_decode): SUB-BANDS of 2^sub_shift consecutive pixel ranks, all chunks
  // of sub-band 0 (chunk-major inside it), then sub-band 1, ...; the last sub-band takes the remainder (one to two sub-bands'
  // worth of pixels: sub_last_pix from rank sub_last_rank0, its items from sub_last_item0).  sub_shift = 0: one band, item =
  // chunk * n_pix + rank.  The rays in flight then belong to one strip of the film (L2 locality of tree scenes; the chunk sums
  // and the film do not depend on it).
  uint32_t sub_shift, sub_last_item0, sub_last_rank0, sub_last_pix;
  // resident pipeline: a workgroup tops its work-item pool up to pool_low by pool_batch items per iteration.  Paths of
  // one workgroup end their chunks in bursts (they all started together), so an open scene at low spp can ask for a
  // hundred items in one iteration, and a lane that finds the pool empty pays a global atomic round trip inside the
  // finish pass: 128 / 256 instead of 24 / 64 tripled such scenes.  Small jobs keep small batches (tail balance).
  uint32_t pool_low, pool_batch;
  // reservations shrink towards the end of the render: a trip to the dispenser asks for at most (items left) >> pool_shift
  // (about half a fair share of what is left per wave / workgroup), so the last items are not parked in one pool while other
  // lanes have retired
  uint32_t pool_shift;
  uint32_t trace_spb;                  // segments per k_trace workgroup pass: its per-BSDF lists span that many segments
  uint32_t shade_ordered;              // k_shade turns each list back into slot order in LDS before shading (coalesced state rows)
  uint32_t dense_shade;                // k_shade_all shades the slots in place: k_trace writes no lists
  int stack_depth;                     // LDS traversal stack entries per lane
#ifdef LR_TIMELINE
  unsigned long long* timeline;        // diagnostic build (make timeline): per wave {entry, dispenser seen dry, exit} in s_memrealtime ticks (100 MHz)
#endif
};

}  // namespace lr
