/*
 * lumilly_host.h -- C ABI of liblumilly_host.so: the host side of the render path (no GPU code).
 *
 * It owns what the reference's Rust host owns around the sampling loop and hands the device
 * library (lumilly_hip.h) a flat LrSceneDesc:
 *
 *   scene file -> Config          src/scene_loader.rs:8-270 (schema, name lookup, light binding)
 *   Config -> camera / primitives src/description.rs:32-197 (OBJ/MTL via tobj, transforms)
 *   matrix conventions            src/math/matrix4.rs:9-68,193-222
 *   camera constructors           src/camera.rs:34-62 (ideal pinhole), :366-409 (thin lens), :148-166
 *   SAH BVH build                 src/bvh.rs:56-127  (any conservative tree gives the same image)
 *   film output                   src/img.rs:40-63, src/main.rs:147-173 (png with gamma, Radiance hdr)
 *   IBL decode                    src/sky.rs:40-55 (image::hdr::HDRDecoder)
 *   pixel tile queue              src/main.rs:73-126 (one job per pixel -> tiles, sharded by rank)
 *
 * Every function returns 0 on success or a negative LR_E* code; lr_host_last_error() gives the
 * thread-local message.  The reference panics instead (description.rs:34,38,139,156,178).
 */
#ifndef LUMILLY_HOST_H
#define LUMILLY_HOST_H

#include "lumilly_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* [renderer] table with the reference defaults applied (description.rs:75-79, main.rs:62-66) */
typedef struct LrRendererConfig {
  int32_t samples;
  int32_t depth;              /* default 5  */
  int32_t depth_limit;        /* default 64 */
  int32_t no_direct_emitter;  /* default 0  */
  int32_t threads;            /* default 0 (= all cores); unused by the GPU path */
  int32_t integrator;         /* LR_INTEGRATOR_*, default pt-direct */
} LrRendererConfig;

#define LR_OUTPUT_PNG 0
#define LR_OUTPUT_HDR 1

/* [film] table (scene_loader.rs:20-27); gamma default 2.2 (main.rs:136) */
typedef struct LrFilmConfig {
  int32_t resolution[2];
  int32_t output;
  float   gamma;
  float   sensitivity[3];     /* parsed, unused -- as in the reference */
} LrFilmConfig;

typedef struct LrHostScene LrHostScene;   /* owns every array LrSceneDesc points at */

/* Description::new (description.rs:32-44).  Mesh / HDR paths are tried as written (relative to
 * the current directory, like the reference) and then relative to asset_root (may be NULL). */
int  lr_host_load_scene(const char* toml_path, const char* asset_root, LrHostScene** out);
int  lr_host_load_scene_string(const char* toml_text, const char* asset_root, LrHostScene** out);
void lr_host_scene_free(LrHostScene* scene);

const LrSceneDesc* lr_host_scene_desc(const LrHostScene* scene);
int  lr_host_scene_renderer(const LrHostScene* scene, LrRendererConfig* out);
int  lr_host_scene_film(const LrHostScene* scene, LrFilmConfig* out);
/* Re-derives the camera for another film size (Description::camera, description.rs:46-55). */
int  lr_host_scene_set_resolution(LrHostScene* scene, int width, int height);
/* BVH build statistics of the scene: seconds, nodes, max depth, SAH cost. */
int  lr_host_scene_bvh_info(const LrHostScene* scene, double* seconds, int* n_nodes, int* max_depth, double* sah_cost);
/* JSON dump of the parsed Config + derived camera + per-primitive data; used by the loader tests. */
int  lr_host_scene_dump_json(const LrHostScene* scene, int max_prims, char** out_json);

/* Stand-alone SAH build over a primitive array (bvh.rs:56-127): the reference's recursion -- full sweep over the three axes,
 * T = 2 T_aabb + (A(S1) N(S1) + A(S2) N(S2)) T_tri / A(S), decided on the primitives' exact boxes, the reference's sequence of
 * sorts (stable) -- so that prim_order_out is the depth-first leaf order of the reference's tree, i.e. the candidate order of
 * bvh.rs:38-45 that decides exact distance ties (bvh.rs:131-141 keeps the first minimum), also inside a leaf of up to max_leaf
 * primitives.  The STORED boxes are padded (conservative).  extra_point (3 floats, may be NULL) is included in the scene extent
 * that sizes the padding (camera position).  Outputs are malloc'd; release with lr_host_free. */
int  lr_host_build_bvh(const LrPrimitive* prims, int n_prims, int max_leaf, const float* extra_point,
                       LrBvhNode** nodes_out, int* n_nodes_out, int32_t** prim_order_out, int* max_depth_out);
void lr_host_free(void* p);

/* Film output.  rgb = linear f32 radiance, row 0 = top (img.rs:21-27). */
int  lr_host_save_png(const char* path, const float* rgb, int width, int height, size_t row_stride_floats, float gamma);
int  lr_host_save_hdr(const char* path, const float* rgb, int width, int height, size_t row_stride_floats);
/* Same files from already quantized pixels (lr_film_quantize on the device). */
int  lr_host_write_png_rgb8(const char* path, const uint8_t* rgb8, int width, int height, size_t row_stride_bytes);
int  lr_host_write_hdr_rgbe(const char* path, const uint8_t* rgbe, int width, int height, size_t row_stride_bytes);
int  lr_host_to_color(const float* rgb, size_t n, float gamma, uint8_t* out);   /* main.rs:171-173 */
int  lr_host_load_hdr(const char* path, float** texels_out, int* width_out, int* height_out);

/* Pixel tile queue (main.rs:65-80: a shared pool of per-pixel jobs): cuts the film into tile x tile blocks and returns the
 * blocks owned by `rank` of `world`, in row-major order.  Block (i, j) of the tile grid belongs to rank
 * lr_host_tile_rank(i, j, world) = (i + k * j) mod world with k = lr_host_tile_stride(world) coprime with world: every rank
 * appears once in any `world` consecutive blocks of a row and of a column, whatever the width of the grid (no stripes).
 * Returns the count; writes at most `cap` tiles.  With out == NULL only counts. */
#define LR_HOST_DEFAULT_TILE 16
int  lr_host_tiles(int width, int height, int tile, int rank, int world, LrTile* out, int cap);
int  lr_host_tile_rank(int tile_i, int tile_j, int world);
int  lr_host_tile_stride(int world);
int  lr_host_default_tile(void);

/* Barrier of `world` processes of one node on THREE zero-initialised uint32 words of memory they all map -- arrivals, round,
 * broken -- (the shared film of bench.py / multigpu.SharedFilm: the counterpart of main.rs:129-132's channel drain across
 * processes).  Everything a process wrote before it arrived is visible to every process that leaves.  LR_EDEVICE after
 * `timeout_s` without the others; that marks the barrier broken, and every waiting or later call on it fails at once. */
int  lr_host_shm_barrier(uint32_t* state, int world, double timeout_s);

size_t      lr_host_sizeof(const char* struct_name);   /* ABI self-check for bindings */
const char* lr_host_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* LUMILLY_HOST_H */
