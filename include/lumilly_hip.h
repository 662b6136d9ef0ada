/*
 * lumilly_hip.h -- C ABI of liblumilly_hip.so, the MI355X (gfx950) replacement for the
 * per-pixel sampling loop of pnlybubbles/LumillyRender.
 *
 * The reference has no FFI; the seam this header cuts is the trait surface its render loop
 * consumes (all citations are into the reference tree):
 *
 *   src/main.rs:70-132        the per-pixel task farm (one closure per pixel folding `spp` samples)
 *   src/camera.rs:9-13        trait Camera { sample(x,y) -> (Sample<Ray>, f32); sensor_sensitivity() }
 *   src/scene.rs:20,34        Scene::radiance / Scene::radiance_nee
 *   src/shape.rs:9-18         Shape::intersect, SurfaceShape::{material, area, sample}
 *   src/material/traits.rs:7-23  Material::{emission, orienting_normal, brdf, sample, weight, coef}
 *   src/sky.rs:9-11           Sky::radiance
 *   src/img.rs:25-27          Img::set(x, y, Vector3)  -- the result sink
 *
 * A Rust host binds these with `extern "C"` (INTEGRATION.md shows the stub); everything is plain
 * pointers and sizes, no C++ or torch types.  All functions return 0 on success and a negative
 * LR_E* code on failure; they never throw or abort.  lr_last_error() returns a thread-local
 * message for the last failure on the calling thread.
 *
 * Threading: one LrScene per device; calls on one handle are serialised by the caller; different
 * handles may be driven concurrently from different host threads / processes (one per GPU).
 * Ownership: the library copies everything it needs out of LrSceneDesc during lr_scene_create and
 * never retains host pointers after a call returns.
 */
#ifndef LUMILLY_HIP_H
#define LUMILLY_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LR_ABI_VERSION 2

/* error codes */
#define LR_OK            0
#define LR_EINVAL       -1   /* bad argument / malformed scene description            */
#define LR_EDEVICE      -2   /* HIP runtime error (message in lr_last_error)          */
#define LR_ENOMEM       -3
#define LR_EUNSUPPORTED -4   /* feature outside the hot-path scope                    */
#define LR_EIO          -5   /* host library: file not found / parse error            */

/* ---- camera: the public fields of the reference camera structs, already derived ------------
 * IdealPinholeCamera  src/camera.rs:16-31 (derived by ::new :34-62)
 * LensCamera          src/camera.rs:340-364 (derived by ::new :366-409)
 * OmnidirectionalCamera src/camera.rs:137-146
 */
#define LR_CAMERA_IDEAL_PINHOLE   0
#define LR_CAMERA_THIN_LENS       1
#define LR_CAMERA_OMNIDIRECTIONAL 2

typedef struct LrCamera {
  int32_t type;
  int32_t resolution[2];             /* film width, height                                  */
  float   forward[3], right[3], up[3];
  float   position[3];               /* sensor centre                                       */
  float   aperture_position[3];
  float   sensor_size[2];
  float   aperture_sensor_distance;
  float   aperture_radius;           /* thin lens only                                      */
  float   focus_distance;            /* thin lens only                                      */
  float   sensor_pixel_area;         /* thin lens only                                      */
  float   sensor_sensitivity;        /* 1.0 for pinhole / omnidirectional                   */
} LrCamera;

/* ---- materials: src/material/{lambert,phong,blinn_phong,ggx,ideal_refraction}.rs ----------- */
#define LR_MAT_LAMBERT          0   /* color = albedo; the only material with emission       */
#define LR_MAT_PHONG            1   /* color = reflectance, param[0] = alpha                 */
#define LR_MAT_BLINN_PHONG      2   /* color = reflectance, param[0] = alpha                 */
#define LR_MAT_GGX              3   /* color = reflectance, param[0] = roughness, [1] = ior  */
#define LR_MAT_IDEAL_REFRACTION 4   /* color = reflectance, param[0] = ior, [1] = absorbtance*/

typedef struct LrMaterial {
  int32_t type;
  float   color[3];
  float   emission[3];               /* light emission * intensity (scene_loader.rs:254-262)  */
  float   param[3];
} LrMaterial;

/* ---- primitives, in the reference's instance order (description.rs:89-148) ------------------
 * The index of a primitive in this array is its identity: ties in hit distance resolve to the
 * LOWEST index (see DESIGN.md "closest-hit semantics").
 */
#define LR_PRIM_TRIANGLE 0          /* v = p0.xyz, p1.xyz, p2.xyz   (triangle.rs:14-40)       */
#define LR_PRIM_SPHERE   1          /* v = centre.xyz, radius       (sphere.rs:13-29)         */

typedef struct LrPrimitive {
  int32_t type;
  int32_t material;                  /* index into LrSceneDesc.materials                     */
  float   v[9];
  float   pad;
} LrPrimitive;

/* ---- sky: src/sky.rs:13-21 (uniform), :35-78 (IBL, nearest texel, width == 2*height) -------- */
#define LR_SKY_UNIFORM 0
#define LR_SKY_IBL     1

typedef struct LrSky {
  int32_t      type;
  float        color[3];             /* uniform sky emission                                 */
  int32_t      height;               /* IBL: texel rows; columns = 2*height                  */
  float        longitude_offset;     /* IBL: added to phi + pi (radians)                     */
  const float* texels;               /* IBL: height * 2*height * 3 linear f32 RGB, row-major  */
} LrSky;

/* ---- BVH built by the host (replaces bvh.rs:56-127; traversal replaces :130-141) ------------
 * Two-child node carrying BOTH child boxes, one axis per 16-byte row so that one node fetch
 * tests two boxes.  A child reference c >= 0 is an inner node index; c < 0 is a leaf,
 * ~c = (first << 3) | count, covering prim_order[first .. first+count) with count in 0..7.
 * Boxes must be CONSERVATIVE (inflated, see lr_host_build_bvh): they only decide how many primitive tests run.  What makes a
 * primitive a CANDIDATE is the reference's own rule, evaluated by the library on the primitive's own exact box (bvh.rs:20-25:
 * aabb.rs:74-92 must pass on it, seeded with [-1e5, 1e5]); the result is the closest hit over the candidates (bvh.rs:131-141).
 */
typedef struct LrBvhNode {
  float   x[4];                      /* left.min.x, left.max.x, right.min.x, right.max.x     */
  float   y[4];
  float   z[4];
  int32_t child[2];
  int32_t pad[2];
} LrBvhNode;

typedef struct LrSceneDesc {
  uint32_t           abi_version;    /* = LR_ABI_VERSION                                     */
  LrCamera           camera;
  int32_t            n_materials;
  const LrMaterial*  materials;
  int32_t            n_prims;
  const LrPrimitive* prims;
  LrSky              sky;
  int32_t            n_bvh_nodes;    /* >= 1 (node 0 is the root); 0 with bvh_nodes == NULL: the library
                                        builds an LBVH on the device (replaces bvh.rs:56-127 there)   */
  const LrBvhNode*   bvh_nodes;
  const int32_t*     bvh_prim_order; /* n_prims entries: leaf ranges index into this.  ALSO the order that breaks exact distance
                                        ties: bvh.rs:131-141 keeps the first minimum of a candidate list that bvh.rs:38-45 fills
                                        depth-first, and lr_host_build_bvh emits the leaves in exactly that order (the reference's
                                        SAH recursion with stable sorts).  A device-built tree breaks ties by primitive index      */
  int32_t            bvh_max_depth;  /* max number of inner nodes on a root-to-leaf path      */
} LrSceneDesc;

/* ---- render request: the [renderer] table (scene_loader.rs:8-18, description.rs:74-79) ------ */
#define LR_INTEGRATOR_PT        0   /* scene.rs:20-32,153-171   */
#define LR_INTEGRATOR_PT_DIRECT 1   /* scene.rs:34-46,173-193   */

typedef struct LrRenderParams {
  int32_t  integrator;
  int32_t  spp;                      /* samples per pixel (renderer.samples)                 */
  uint32_t seed;                     /* counter-based RNG key (the reference is unseeded)    */
  int32_t  depth;                    /* forced-continue depth, default 5                     */
  int32_t  depth_limit;              /* default 64                                           */
  int32_t  no_direct_emitter;        /* bool                                                 */
  int32_t  path_slots;               /* 0 = library default; path-state slots (paths in flight).  The fused and the resident
                                        pipeline treat it as an UPPER bound: they cannot use more than one path per lane of
                                        the waves a GPU holds */
  int32_t  flags;                    /* LR_FLAG_*                                            */
} LrRenderParams;

#define LR_FLAG_PROFILE 1            /* bracket kernel launches with HIP events (lr_get_stats) */
#define LR_FLAG_COUNT   2            /* count segments / shadow rays / node visits / prim tests */
#define LR_FLAG_STREAMING 4          /* force the multi-kernel streaming pipeline (state in HBM)      */
#define LR_FLAG_RESIDENT  8          /* force the single-launch resident pipeline (state in LDS) if it fits; if it does not, the default choice runs */
#define LR_FLAG_FUSED     16         /* force the fused pipeline: one persistent launch, every lane carries its path in registers */

typedef struct LrTile { int32_t x0, y0, w, h; } LrTile;

/* kernel ids for LrStats.kernel_* */
#define LR_K_GENERATE 0
#define LR_K_TRACE    1
#define LR_K_SHADE    2
#define LR_K_SHADOW   3
#define LR_K_RESOLVE  4
#define LR_K_RESIDENT 5              /* the resident pipeline's single launch (trace/shade/shadow phases) */
#define LR_K_PATH     6              /* the fused pipeline's single launch (k_path_*: a lane carries its path)   */
#define LR_K_COUNT    7

typedef struct LrStats {
  uint64_t samples;                  /* camera samples completed                             */
  uint64_t segments;                 /* S: closest-hit queries                               */
  uint64_t shadow_rays;              /* Q                                                    */
  uint64_t node_visits;              /* V: child boxes tested by closest-hit queries (LR_FLAG_COUNT) */
  uint64_t prim_tests;               /* T: primitive tests of closest-hit queries    (LR_FLAG_COUNT) */
  uint64_t shadow_node_visits;       /* same, shadow-ray queries                     (LR_FLAG_COUNT) */
  uint64_t shadow_prim_tests;
  uint64_t sky_fetches;              /* M                                                    */
  uint64_t iterations;               /* wavefront loop iterations                            */
  uint64_t kernel_launches[LR_K_COUNT];
  double   kernel_ms[LR_K_COUNT];    /* sum of event-timed launches  (LR_FLAG_PROFILE)       */
  uint64_t kernel_timed[LR_K_COUNT]; /* number of launches that were event-timed             */
  double   render_ms;                /* wall time of the last lr_render, device work only    */
  double   upload_ms;                /* lr_scene_create                                      */
  double   bvh_build_ms;             /* device LBVH build inside lr_scene_create (0 = host tree) */
  uint64_t path_slots;               /* path-state slots the last render ran with                */
  uint64_t pipeline;                 /* 2 = fused (one launch, state in registers), 1 = resident (one launch, state in LDS), 0 = streaming */
} LrStats;

typedef struct LrScene LrScene;      /* opaque */

int         lr_device_count(void);
int         lr_scene_create(int device, const LrSceneDesc* desc, LrScene** out);
int         lr_scene_destroy(LrScene* scene);

/* Renders the given pixel tiles (disjoint, any subset of the film) and writes them into the
 * caller's film: rgb_out[(y*row_stride) + 3*x + c], f32 linear radiance, row 0 = top
 * (img.rs:21-27).  Only tile pixels are written.  Blocks until done.  Replaces main.rs:70-132. */
int         lr_render(LrScene* scene, const LrRenderParams* params,
                      const LrTile* tiles, int n_tiles,
                      float* rgb_out, size_t row_stride_floats);

/* Same, but the film stays on the device (W*H*3 f32, dense rows); *film_dev receives the device
 * pointer, valid until the next call on this handle.  Used when the caller gathers with its own
 * copies (bench: inputs and outputs resident in HBM). */
int         lr_render_device(LrScene* scene, const LrRenderParams* params,
                             const LrTile* tiles, int n_tiles, void** film_dev);

/* Film output stage on the device, applied to the film of the last render (whole film, W*H pixels):
 *   LR_QUANT_RGB8  3 bytes/pixel, trunc(clamp(x,0,1)^(1/gamma) * 255)          (main.rs:171-173)
 *   LR_QUANT_RGBE  4 bytes/pixel, Radiance RGBE of the linear value            (img.rs:40-50)
 * out receives dense rows (row_stride_bytes >= W * bytes/pixel).  lr_host_write_png_rgb8 /
 * lr_host_write_hdr_rgbe turn the bytes into files. */
#define LR_QUANT_RGB8 0
#define LR_QUANT_RGBE 1
int         lr_film_quantize(LrScene* scene, int mode, float gamma, uint8_t* out, size_t row_stride_bytes);

int         lr_get_stats(LrScene* scene, LrStats* out);
const char* lr_last_error(void);
const char* lr_build_info(void);     /* "gfx950 ..." */

#ifdef __cplusplus
}
#endif
#endif /* LUMILLY_HIP_H */
