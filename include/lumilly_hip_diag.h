/*
 * lumilly_hip_diag.h -- DIAGNOSTIC entry points of liblumilly_hip.so.
 *
 * Not part of the drop-in surface (include/lumilly_hip.h is the reference-shaped boundary): these run single
 * device functions of the render path on caller data so that the parity tests can compare them, bit for bit,
 * with the CPU oracle -- the device-side counterpart of the reference's inline unit tests
 * (src/triangle.rs:152-236, src/util.rs:45-82).  Same conventions: plain pointers and sizes, 0 / LR_E* return
 * codes, never throw.
 */
#ifndef LUMILLY_HIP_DIAG_H
#define LUMILLY_HIP_DIAG_H

#include "lumilly_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* The deterministic math spec on the device (csrc/lr_math.h): out[i] = fn(a[i] [, b[i]]).
 * fn: 0 sin, 1 cos, 2 acos, 3 atan2(a, b), 4 pow(a, b), 5 exp, 6 fmod_pos(a, b), 7 a / b, 8 sqrt,
 *     9 Lambert grid level at (a, b) (lambert.rs:66-90).  b may be NULL for the unary ones. */
int lr_selftest_math(int device, int fn, const float* a, const float* b, float* out, int n);

/* Exhaustive check of the five-instruction exact reciprocal used by the triangle test against the IEEE quotient
 * over every float whose biased exponent lies in [lo_exp, hi_exp]: out4 = {mismatches of the 2-step form,
 * mismatches of the 3-step form, one offending bit pattern each}. */
int lr_selftest_rcp(int device, uint32_t lo_exp, uint32_t hi_exp, uint64_t* out4);

/* The counter-based generator: out4[4*i..] = the four draws of block[i] of (seed, pixel[i], sample[i]). */
int lr_selftest_rng(int device, uint32_t seed, const uint32_t* pixel, const uint32_t* sample, const uint32_t* block,
                    float* out4, int n);

/* Closest hit of n rays through the scene's traversal path (flat loop or 4-wide tree, as lr_render would choose):
 * prim_out[i] = primitive index or -1, t_out[i] = distance (0 on a miss).  Replaces bvh.rs:130-141 for a batch. */
int lr_selftest_intersect(LrScene* scene, int n, const float* origins, const float* dirs, int32_t* prim_out, float* t_out);

/* Closest hit of n rays through the scene's traversal path (flat loop or 4-wide tree, as lr_render would choose) --
 * see above -- is checked against two brute-force forms on the device, both O(n * n_prims), same primitive tests as
 * the render path, ties to the lowest primitive index:
 *   lr_selftest_brute_own_box  the DEFINITION of the render path's closest hit (bvh.rs:20-25 + 131-141): every ray
 *                              against every primitive whose OWN exact box passes aabb.rs:74-92 (the literal slab test,
 *                              evaluated per primitive, no tree, no shortcut);
 *   lr_selftest_brute          the candidate set widened to EVERY primitive, no box at all (what rounds 1-5 defined
 *                              the closest hit by; differs from the reference where a leaf's own box rejects a ray its
 *                              primitive test accepts). */
int lr_selftest_brute(LrScene* scene, int n, const float* origins, const float* dirs, int32_t* prim_out, float* t_out);
int lr_selftest_brute_own_box(LrScene* scene, int n, const float* origins, const float* dirs, int32_t* prim_out, float* t_out);

/* Emitter pick of objects.rs:37-51 on the device: for n uniform draws xi in [0,1), k_out[i] = index (into the scene's
 * emitter list, instance order) of the emitter chosen by roulette = total_area * xi[i]. */
int lr_selftest_emitter_pick(LrScene* scene, int n, const float* xi, int32_t* k_out);

/* Sky::radiance (sky.rs:13-21 uniform, :57-78 IBL nearest texel) for n unit directions: rgb_out[3*i..]. */
int lr_selftest_sky(LrScene* scene, int n, const float* dirs, float* rgb_out);

/* material/{lambert,phong,blinn_phong,ggx,ideal_refraction}.rs for n inputs of one material, through the per-lane dispatch the
 * render kernels use: in13[13*i..] = out_.xyz, normal.xyz, position.xyz, xi[3], fly distance;
 * out10[10*i..] = sample() -> in_.xyz, pdf; brdf(out_, in_, normal, position).rgb; coef(out_, normal, fly distance).rgb. */
int lr_selftest_material(int device, const LrMaterial* material, int n, const float* in13, float* out10);

/* Camera::sample (camera.rs:64-115 pinhole, :411-476 thin lens, :168-188 omnidirectional) of the scene's camera:
 * xy[2*i..] = pixel, xi4[4*i..] = the four draws; out8[8*i..] = ray origin.xyz, direction.xyz, geometry term, 0. */
int lr_selftest_camera(LrScene* scene, int n, const int32_t* xy, const float* xi4, float* out8);

/* Objects::sample_emission in full (objects.rs:37-51 + triangle.rs:140-149 / sphere.rs:79-84): xi4[4*i..] = (-, pick, u, v);
 * out4[4*i..] = sampled point.xyz, pdf. */
int lr_selftest_emission_sample(LrScene* scene, int n, const float* xi4, float* out4);

/* The scene's 4-wide tree: out4 = {nodes, nodes whose distance-culling slack exceeds the distance itself (2 kappa >= 1: a
 * wall-sized triangle somewhere below them, DESIGN.md section 2 -- this count is how much culling the error bound gives up),
 * sliver triangles (sin of the angle at p0 below 1/8: statistics only), worst-case traversal stack entries}. */
int lr_selftest_tree_info(LrScene* scene, int32_t* out4);

/* How the scene's IBL map is stored in HBM: 4 = RGBE words (every texel of the caller's map is a Radiance value
 * c * 2^(e - 136) and re-encodes exactly; decoded on the fly to the same f32 bits), 16 = float4, 0 = no map. */
int lr_selftest_sky_texel_bytes(LrScene* scene);

#ifdef __cplusplus
}
#endif
#endif /* LUMILLY_HIP_DIAG_H */
