// description.cpp -- Config -> flat LrSceneDesc, plus the C ABI of liblumilly_host.so.
// Follows Description::{new, camera, scene} and Loader (description.rs:32-197): object order is
// instance order, OBJ faces become triangles in file order, spheres take the transformed origin
// and an UNSCALED radius, emission binds to objects by light.object == object.name.
#include "host_internal.h"
#include <chrono>
#include <cmath>
#include <thread>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <algorithm>
#include <map>
#include <memory>
#include <sstream>
#include <sys/stat.h>

namespace lrhost {

static thread_local std::string g_last_error;
void set_last_error(const std::string& m) { g_last_error = m; }
void fail(int code, const std::string& msg) { throw HostError{code, msg}; }

std::string resolve_path(const std::string& path, const std::string& asset_root) {
  struct stat st;
  if (stat(path.c_str(), &st) == 0 && S_ISREG(st.st_mode)) return path;
  if (!asset_root.empty()) {
    std::string p = asset_root + "/" + path;
    if (stat(p.c_str(), &st) == 0 && S_ISREG(st.st_mode)) return p;
  }
  fail(LR_EIO, "File `" + path + "` is not found.");               // description.rs:34
}

// camera.rs:34-62 (IdealPinholeCamera::new), :366-409 (LensCamera::new), :148-166 (Omnidirectional)
LrCamera make_camera(const CameraCfg& cfg, int width, int height) {
  LrCamera c; std::memset(&c, 0, sizeof(c));
  Mat4 matrix = compose(cfg.transform);
  c.type = cfg.type;
  c.resolution[0] = width; c.resolution[1] = height;
  Vec3 aperture_position = matrix.row3(3);
  Vec3 forward = mul(matrix, vec3(0.0f, 0.0f, -1.0f));
  Vec3 right = mul(matrix, vec3(1.0f, 0.0f, 0.0f));
  Vec3 up = mul(matrix, vec3(0.0f, 1.0f, 0.0f));
  auto put = [](float* d, Vec3 v) { d[0] = v.x; d[1] = v.y; d[2] = v.z; };
  put(c.forward, forward); put(c.right, right); put(c.up, up); put(c.aperture_position, aperture_position);
  c.sensor_sensitivity = 1.0f;
  if (cfg.type == LR_CAMERA_OMNIDIRECTIONAL) { put(c.position, aperture_position); return c; }
  Vec3 direction = forward * 50.0f;
  Vec3 position = aperture_position - direction;
  put(c.position, position);
  float aperture_sensor_distance = norm(direction);
  c.aperture_sensor_distance = aperture_sensor_distance;
  float sensor_size_x = 2.0f * aperture_sensor_distance * std::tan(cfg.fov * kPi / 180.0f / 2.0f);
  float sensor_size_y = sensor_size_x * (float)height / (float)width;
  c.sensor_size[0] = sensor_size_x; c.sensor_size[1] = sensor_size_y;
  if (cfg.type == LR_CAMERA_THIN_LENS) {
    float focal_length = 1.0f / (1.0f / aperture_sensor_distance + 1.0f / cfg.focus_distance);
    float aperture_radius = focal_length / cfg.f_number / 2.0f;
    float sensor_pixel_area = (sensor_size_x * sensor_size_y) / (float)((size_t)width * (size_t)height);
    c.aperture_radius = aperture_radius;
    c.focus_distance = cfg.focus_distance;
    c.sensor_pixel_area = sensor_pixel_area;
    c.sensor_sensitivity = aperture_sensor_distance * aperture_sensor_distance / (sensor_pixel_area * kPi * aperture_radius * aperture_radius);
  }
  return c;
}

namespace {

const MeshCfg& find_mesh(const Config& c, const std::string& name) {          // scene_loader.rs:239-242
  for (const MeshCfg& m : c.mesh) if (m.name == name) return m;
  fail(LR_EINVAL, "Mesh named `" + name + "` is not found.");
}
const MaterialCfg& find_material(const Config& c, const std::string& name) {  // scene_loader.rs:244-247
  for (const MaterialCfg& m : c.material) if (m.name == name) return m;
  fail(LR_EINVAL, "Material named `" + name + "` is not found.");
}

void instantiate(LrHostScene& s) {
  const Config& cfg = s.config;
  std::map<std::string, ObjFile> obj;                                         // Loader::load_obj description.rs:150-162
  for (const ObjectCfg& o : cfg.object) {
    const MeshCfg& m = find_mesh(cfg, o.mesh);
    if (m.is_obj && !obj.count(m.name)) obj[m.name] = load_obj(resolve_path(m.path, s.asset_root));
  }
  s.materials.clear(); s.prims.clear();
  for (const ObjectCfg& o : cfg.object) {                                     // Loader::new description.rs:89-148
    const MeshCfg& mesh = find_mesh(cfg, o.mesh);
    Mat4 transform = compose(o.transform);
    Vec3 emission = vec3(0, 0, 0);                                            // scene_loader.rs:254-262
    if (o.has_name)
      for (const LightCfg& l : cfg.light)
        if (l.object == o.name) { emission = l.emission * (l.has_intensity ? l.intensity : 1.0f); break; }
    int default_material = -1;
    if (o.has_material) {
      const MaterialCfg& mc = find_material(cfg, o.material);
      LrMaterial lm; std::memset(&lm, 0, sizeof(lm));
      lm.type = mc.type;
      lm.color[0] = mc.color.x; lm.color[1] = mc.color.y; lm.color[2] = mc.color.z;
      lm.param[0] = mc.p0; lm.param[1] = mc.p1;
      if (mc.type == LR_MAT_LAMBERT) { lm.emission[0] = emission.x; lm.emission[1] = emission.y; lm.emission[2] = emission.z; }   // only Lambert emits (description.rs:98-101)
      s.materials.push_back(lm);
      default_material = (int)s.materials.size() - 1;
    }
    if (mesh.is_obj) {                                                        // Loader::obj description.rs:164-197
      const ObjFile& f = obj[mesh.name];
      int mtl_base = (int)s.materials.size();
      if (default_material < 0) {
        for (const ObjMaterial& om : f.materials) {
          LrMaterial lm; std::memset(&lm, 0, sizeof(lm));
          lm.type = LR_MAT_LAMBERT;
          lm.color[0] = om.diffuse[0]; lm.color[1] = om.diffuse[1]; lm.color[2] = om.diffuse[2];
          lm.emission[0] = emission.x; lm.emission[1] = emission.y; lm.emission[2] = emission.z;
          s.materials.push_back(lm);
        }
      }
      for (const ObjModel& m : f.models) {
        int mat = default_material;
        if (mat < 0) {
          if (m.material_id < 0) {
            if (m.indices.empty()) continue;
            fail(LR_EINVAL, "Specified material is not found in mlt file. (mesh `" + mesh.name + "`, model `" + m.name + "`)");
          }
          mat = mtl_base + m.material_id;
        }
        for (size_t fi = 0; fi + 2 < m.indices.size(); fi += 3) {
          LrPrimitive p; std::memset(&p, 0, sizeof(p));
          p.type = LR_PRIM_TRIANGLE; p.material = mat;
          for (int i = 0; i < 3; ++i) {
            uint32_t vi = m.indices[fi + i];
            Vec3 q = mul(transform, vec3(m.positions[3 * vi], m.positions[3 * vi + 1], m.positions[3 * vi + 2]));
            p.v[3 * i] = q.x; p.v[3 * i + 1] = q.y; p.v[3 * i + 2] = q.z;
          }
          s.prims.push_back(p);
        }
      }
    } else {                                                                  // description.rs:137-142
      if (default_material < 0) fail(LR_EINVAL, "Material must be specified for object `" + mesh.name + "`");
      Vec3 position = mul(transform, vec3(0, 0, 0));
      LrPrimitive p; std::memset(&p, 0, sizeof(p));
      p.type = LR_PRIM_SPHERE; p.material = default_material;
      p.v[0] = position.x; p.v[1] = position.y; p.v[2] = position.z; p.v[3] = mesh.radius;
      s.prims.push_back(p);
    }
  }
}

void finish_desc(LrHostScene& s) {
  const Config& cfg = s.config;
  LrSceneDesc& d = s.desc;
  std::memset(&d, 0, sizeof(d));
  d.abi_version = LR_ABI_VERSION;
  d.camera = make_camera(cfg.camera, cfg.film.resolution[0], cfg.film.resolution[1]);
  d.n_materials = (int)s.materials.size(); d.materials = s.materials.data();
  d.n_prims = (int)s.prims.size(); d.prims = s.prims.data();
  d.sky.type = LR_SKY_UNIFORM;                                                // description.rs:58-65: no [sky] -> black uniform
  if (cfg.sky.present) {
    d.sky.type = cfg.sky.type;
    d.sky.color[0] = cfg.sky.color.x; d.sky.color[1] = cfg.sky.color.y; d.sky.color[2] = cfg.sky.color.z;
    if (cfg.sky.type == LR_SKY_IBL) {
      d.sky.height = s.sky_h;
      d.sky.longitude_offset = cfg.sky.longitude_offset;
      d.sky.texels = s.texels.data();
    }
  }
  d.n_bvh_nodes = (int)s.bvh.nodes.size(); d.bvh_nodes = s.bvh.nodes.data();
  d.bvh_prim_order = s.bvh.order.data(); d.bvh_max_depth = s.bvh.max_depth;
}

LrHostScene* load_from_text(const std::string& text, const char* asset_root) {
  std::unique_ptr<LrHostScene> s(new LrHostScene());
  s->asset_root = asset_root ? asset_root : "";
  s->config = parse_config(text);
  if (s->config.film.resolution[0] <= 0 || s->config.film.resolution[1] <= 0) fail(LR_EINVAL, "film.resolution must be positive");
  instantiate(*s);
  if (s->config.sky.present && s->config.sky.type == LR_SKY_IBL) {            // IBLSky::new sky.rs:42-55
    int w = 0, h = 0;
    load_hdr(resolve_path(s->config.sky.path, s->asset_root), s->texels, w, h);
    s->sky_w = w; s->sky_h = h;
    if (w != 2 * h) fail(LR_EUNSUPPORTED, "ibl: the lookup assumes width == 2*height (sky.rs:66-67)");
  }
  LrCamera cam = make_camera(s->config.camera, s->config.film.resolution[0], s->config.film.resolution[1]);
  s->bvh = build_bvh(s->prims.data(), (int)s->prims.size(), 4, cam.aperture_position);
  finish_desc(*s);
  return s.release();
}

void json_f(std::ostringstream& o, float v) {
  char b[64];
  if (v != v) std::snprintf(b, sizeof(b), "\"nan\"");
  else if (std::isinf(v)) std::snprintf(b, sizeof(b), v > 0 ? "\"inf\"" : "\"-inf\"");
  else std::snprintf(b, sizeof(b), "%.9g", (double)v);
  o << b;
}
void json_v(std::ostringstream& o, const float* v, int n) { o << "["; for (int i = 0; i < n; ++i) { if (i) o << ","; json_f(o, v[i]); } o << "]"; }

}  // namespace
}  // namespace lrhost

using namespace lrhost;

#define LR_HOST_TRY(...)                                                             \
  try { __VA_ARGS__; return LR_OK; }                                                      \
  catch (const HostError& e) { set_last_error(e.msg); return e.code; }               \
  catch (const std::bad_alloc&) { set_last_error("out of memory"); return LR_ENOMEM; } \
  catch (const std::exception& e) { set_last_error(e.what()); return LR_EINVAL; }

extern "C" {

const char* lr_host_last_error(void) { return g_last_error.c_str(); }

int lr_host_load_scene_string(const char* text, const char* asset_root, LrHostScene** out) {
  LR_HOST_TRY({
    if (!text || !out) fail(LR_EINVAL, "null argument");
    *out = load_from_text(text, asset_root);
  })
}
int lr_host_load_scene(const char* path, const char* asset_root, LrHostScene** out) {
  LR_HOST_TRY({
    if (!path || !out) fail(LR_EINVAL, "null argument");
    std::ifstream f(path, std::ios::binary);
    if (!f) fail(LR_EIO, std::string("File `") + path + "` is not found.");
    std::stringstream ss; ss << f.rdbuf();
    *out = load_from_text(ss.str(), asset_root);
  })
}
void lr_host_scene_free(LrHostScene* s) { delete s; }
const LrSceneDesc* lr_host_scene_desc(const LrHostScene* s) { return s ? &s->desc : nullptr; }
int lr_host_scene_renderer(const LrHostScene* s, LrRendererConfig* out) {
  if (!s || !out) { set_last_error("null argument"); return LR_EINVAL; }
  *out = s->config.renderer; return LR_OK;
}
int lr_host_scene_film(const LrHostScene* s, LrFilmConfig* out) {
  if (!s || !out) { set_last_error("null argument"); return LR_EINVAL; }
  *out = s->config.film; return LR_OK;
}
int lr_host_scene_set_resolution(LrHostScene* s, int w, int h) {
  LR_HOST_TRY({
    if (!s || w <= 0 || h <= 0) fail(LR_EINVAL, "bad resolution");
    s->config.film.resolution[0] = w; s->config.film.resolution[1] = h;
    finish_desc(*s);
  })
}
int lr_host_scene_bvh_info(const LrHostScene* s, double* seconds, int* n_nodes, int* max_depth, double* sah_cost) {
  if (!s) { set_last_error("null argument"); return LR_EINVAL; }
  if (seconds) *seconds = s->bvh.seconds;
  if (n_nodes) *n_nodes = (int)s->bvh.nodes.size();
  if (max_depth) *max_depth = s->bvh.max_depth;
  if (sah_cost) *sah_cost = s->bvh.sah_cost;
  return LR_OK;
}
int lr_host_scene_dump_json(const LrHostScene* s, int max_prims, char** out_json) {
  LR_HOST_TRY({
    if (!s || !out_json) fail(LR_EINVAL, "null argument");
    std::ostringstream o;
    const LrSceneDesc& d = s->desc; const LrCamera& c = d.camera;
    o << "{\"renderer\":{\"samples\":" << s->config.renderer.samples << ",\"depth\":" << s->config.renderer.depth
      << ",\"depth_limit\":" << s->config.renderer.depth_limit << ",\"no_direct_emitter\":" << s->config.renderer.no_direct_emitter
      << ",\"threads\":" << s->config.renderer.threads << ",\"integrator\":" << s->config.renderer.integrator << "},";
    o << "\"film\":{\"resolution\":[" << s->config.film.resolution[0] << "," << s->config.film.resolution[1] << "],\"output\":" << s->config.film.output << ",\"gamma\":";
    json_f(o, s->config.film.gamma); o << "},";
    o << "\"camera\":{\"type\":" << c.type << ",\"forward\":"; json_v(o, c.forward, 3);
    o << ",\"right\":"; json_v(o, c.right, 3); o << ",\"up\":"; json_v(o, c.up, 3);
    o << ",\"position\":"; json_v(o, c.position, 3); o << ",\"aperture_position\":"; json_v(o, c.aperture_position, 3);
    o << ",\"sensor_size\":"; json_v(o, c.sensor_size, 2);
    o << ",\"aperture_sensor_distance\":"; json_f(o, c.aperture_sensor_distance);
    o << ",\"aperture_radius\":"; json_f(o, c.aperture_radius);
    o << ",\"focus_distance\":"; json_f(o, c.focus_distance);
    o << ",\"sensor_pixel_area\":"; json_f(o, c.sensor_pixel_area);
    o << ",\"sensor_sensitivity\":"; json_f(o, c.sensor_sensitivity); o << "},";
    o << "\"sky\":{\"type\":" << d.sky.type << ",\"color\":"; json_v(o, d.sky.color, 3);
    o << ",\"height\":" << d.sky.height << ",\"longitude_offset\":"; json_f(o, d.sky.longitude_offset); o << "},";
    o << "\"materials\":[";
    for (int i = 0; i < d.n_materials; ++i) {
      const LrMaterial& m = d.materials[i];
      if (i) o << ",";
      o << "{\"type\":" << m.type << ",\"color\":"; json_v(o, m.color, 3);
      o << ",\"emission\":"; json_v(o, m.emission, 3); o << ",\"param\":"; json_v(o, m.param, 3); o << "}";
    }
    o << "],\"n_prims\":" << d.n_prims << ",\"prims\":[";
    int np = max_prims < 0 ? d.n_prims : std::min(max_prims, d.n_prims);
    for (int i = 0; i < np; ++i) {
      const LrPrimitive& p = d.prims[i];
      if (i) o << ",";
      o << "{\"type\":" << p.type << ",\"material\":" << p.material << ",\"v\":"; json_v(o, p.v, p.type == LR_PRIM_TRIANGLE ? 9 : 4); o << "}";
    }
    o << "],\"bvh\":{\"nodes\":" << d.n_bvh_nodes << ",\"max_depth\":" << d.bvh_max_depth << "}}";
    std::string str = o.str();
    char* buf = (char*)std::malloc(str.size() + 1);
    if (!buf) fail(LR_ENOMEM, "out of memory");
    std::memcpy(buf, str.c_str(), str.size() + 1);
    *out_json = buf;
  })
}

int lr_host_build_bvh(const LrPrimitive* prims, int n, int max_leaf, const float* extra_point,
                      LrBvhNode** nodes_out, int* n_nodes_out, int32_t** order_out, int* max_depth_out) {
  LR_HOST_TRY({
    if (!nodes_out || !n_nodes_out || !order_out || !max_depth_out) fail(LR_EINVAL, "null argument");
    BvhResult r = build_bvh(prims, n, max_leaf, extra_point);
    LrBvhNode* nodes = (LrBvhNode*)std::malloc(sizeof(LrBvhNode) * std::max<size_t>(r.nodes.size(), 1));
    int32_t* order = (int32_t*)std::malloc(sizeof(int32_t) * std::max<size_t>(r.order.size(), 1));
    if (!nodes || !order) { std::free(nodes); std::free(order); fail(LR_ENOMEM, "out of memory"); }
    std::memcpy(nodes, r.nodes.data(), sizeof(LrBvhNode) * r.nodes.size());
    if (!r.order.empty()) std::memcpy(order, r.order.data(), sizeof(int32_t) * r.order.size());
    *nodes_out = nodes; *n_nodes_out = (int)r.nodes.size(); *order_out = order; *max_depth_out = r.max_depth;
  })
}
void lr_host_free(void* p) { std::free(p); }

int lr_host_save_png(const char* path, const float* rgb, int w, int h, size_t stride, float gamma) {
  LR_HOST_TRY({ if (!path || !rgb) fail(LR_EINVAL, "null argument"); save_png(path, rgb, w, h, stride, gamma); })
}
int lr_host_save_hdr(const char* path, const float* rgb, int w, int h, size_t stride) {
  LR_HOST_TRY({ if (!path || !rgb) fail(LR_EINVAL, "null argument"); save_hdr(path, rgb, w, h, stride); })
}
int lr_host_write_png_rgb8(const char* path, const uint8_t* rgb8, int w, int h, size_t stride) {
  LR_HOST_TRY({ if (!path || !rgb8) fail(LR_EINVAL, "null argument"); write_png_rgb8(path, rgb8, w, h, stride); })
}
int lr_host_write_hdr_rgbe(const char* path, const uint8_t* rgbe, int w, int h, size_t stride) {
  LR_HOST_TRY({ if (!path || !rgbe) fail(LR_EINVAL, "null argument"); write_hdr_rgbe(path, rgbe, w, h, stride); })
}
int lr_host_to_color(const float* rgb, size_t n, float gamma, uint8_t* out) {
  if (!rgb || !out) { set_last_error("null argument"); return LR_EINVAL; }
  for (size_t i = 0; i < n; ++i) out[i] = to_color(rgb[i], gamma);
  return LR_OK;
}
int lr_host_load_hdr(const char* path, float** texels_out, int* w_out, int* h_out) {
  LR_HOST_TRY({
    if (!path || !texels_out || !w_out || !h_out) fail(LR_EINVAL, "null argument");
    std::vector<float> t; int w, h; load_hdr(path, t, w, h);
    float* buf = (float*)std::malloc(sizeof(float) * t.size());
    if (!buf) fail(LR_ENOMEM, "out of memory");
    std::memcpy(buf, t.data(), sizeof(float) * t.size());
    *texels_out = buf; *w_out = w; *h_out = h;
  })
}

// The reference farms one job per pixel from a shared pool (main.rs:65-80,104,121: whichever worker is free takes the next
// pixel), balanced by construction.  One process per GPU has no shared pool, so the deal itself must not be able to
// degenerate: tile (i, j) of the tile grid goes to rank (i + k * j) mod world, k = the stride nearest 0.382 * world that
// is coprime with world (world 8: k = 3, world 4: k = 1, world 2: checkerboard).  Every rank then appears once in any `world`
// consecutive tiles of a ROW and of a COLUMN whatever the width of the grid -- `id % world` (round 4) gave rank r whole
// columns whenever the grid was a multiple of `world` wide (the 16- and 32-tile-wide films of configs 2 and 5: heaviest
// rank +12 % / +10 %, profiles/r05_strong_rank_before_*.json).
int lr_host_tile_stride(int world) {
  if (world <= 2) return 1;
  int best = 1; double bd = 1e30;
  for (int k = 1; k < world; ++k) {
    int a = k, b = world; while (b) { int t = a % b; a = b; b = t; }
    if (a != 1) continue;
    const double d = std::fabs((double)k - 0.382 * (double)world);
    if (d < bd) { bd = d; best = k; }
  }
  return best;
}
int lr_host_tile_rank(int i, int j, int world) {
  if (world <= 0 || i < 0 || j < 0) return -1;
  return (int)(((long long)i + (long long)lr_host_tile_stride(world) * j) % world);
}
int lr_host_default_tile(void) { return LR_HOST_DEFAULT_TILE; }

int lr_host_tiles(int width, int height, int tile, int rank, int world, LrTile* out, int cap) {
  if (width <= 0 || height <= 0 || tile <= 0 || world <= 0 || rank < 0 || rank >= world) { set_last_error("bad tiling arguments"); return LR_EINVAL; }
  const int tx = (width + tile - 1) / tile, ty = (height + tile - 1) / tile, k = lr_host_tile_stride(world);
  int count = 0;
  for (int j = 0; j < ty; ++j)
    for (int i = 0; i < tx; ++i) {
      if ((int)(((long long)i + (long long)k * j) % world) != rank) continue;
      if (out && count < cap) {
        LrTile t; t.x0 = i * tile; t.y0 = j * tile;
        t.w = std::min(tile, width - t.x0); t.h = std::min(tile, height - t.y0);
        out[count] = t;
      }
      ++count;
    }
  return count;
}

// Barrier of `world` processes on two words of shared memory (the film every rank of a node maps: multigpu.SharedFilm):
// state[0] = arrivals of the current round, state[1] = round number, state[2] = broken.  Sense-reversing: the last arrival resets the
// count and advances the round, the others spin on the round (pause, then yield, then sleep 50 us) until it moves or `timeout_s` runs
// out.  A rank that times out marks the barrier BROKEN: every rank still waiting and every later call returns LR_EDEVICE at once --
// a late rank must not complete the round alone and render into a film nobody is synchronised on.
// Sequentially consistent atomics: everything a rank wrote into the film before it arrived is visible to every rank that leaves.
static inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
  __builtin_ia32_pause();
#elif defined(__aarch64__)
  asm volatile("yield");
#endif
}
int lr_host_shm_barrier(uint32_t* state, int world, double timeout_s) {
  if (!state || world <= 0) { set_last_error("bad barrier arguments"); return LR_EINVAL; }
  if (world == 1) return LR_OK;
  if (__atomic_load_n(&state[2], __ATOMIC_SEQ_CST) != 0u) { set_last_error("shared-memory barrier is broken (a rank timed out on it earlier)"); return LR_EDEVICE; }
  const uint32_t round = __atomic_load_n(&state[1], __ATOMIC_SEQ_CST);
  if (__atomic_add_fetch(&state[0], 1u, __ATOMIC_SEQ_CST) == (uint32_t)world) {
    __atomic_store_n(&state[0], 0u, __ATOMIC_SEQ_CST);
    __atomic_add_fetch(&state[1], 1u, __ATOMIC_SEQ_CST);
    return LR_OK;
  }
  const auto t0 = std::chrono::steady_clock::now();
  for (uint64_t spins = 0;; ++spins) {
    if (__atomic_load_n(&state[1], __ATOMIC_SEQ_CST) != round) return LR_OK;
    if (spins < 2000) cpu_relax();
    else if (spins < 20000) std::this_thread::yield();
    else {
      std::this_thread::sleep_for(std::chrono::microseconds(50));
      if (__atomic_load_n(&state[2], __ATOMIC_SEQ_CST) != 0u) { set_last_error("shared-memory barrier was broken by another rank's timeout"); return LR_EDEVICE; }
      if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s) {
        __atomic_store_n(&state[2], 1u, __ATOMIC_SEQ_CST);
        if (__atomic_load_n(&state[1], __ATOMIC_SEQ_CST) != round) return LR_OK;      // (the round completed while this rank gave up: it stands; the flag stops the next one)
        set_last_error("shared-memory barrier timed out (a rank of the node did not arrive)");
        return LR_EDEVICE;
      }
    }
  }
}

size_t lr_host_sizeof(const char* name) {
  if (!name) return 0;
  std::string n(name);
  if (n == "LrCamera") return sizeof(LrCamera);
  if (n == "LrMaterial") return sizeof(LrMaterial);
  if (n == "LrPrimitive") return sizeof(LrPrimitive);
  if (n == "LrSky") return sizeof(LrSky);
  if (n == "LrBvhNode") return sizeof(LrBvhNode);
  if (n == "LrSceneDesc") return sizeof(LrSceneDesc);
  if (n == "LrRenderParams") return sizeof(LrRenderParams);
  if (n == "LrTile") return sizeof(LrTile);
  if (n == "LrStats") return sizeof(LrStats);
  if (n == "LrRendererConfig") return sizeof(LrRendererConfig);
  if (n == "LrFilmConfig") return sizeof(LrFilmConfig);
  return 0;
}

}  // extern "C"
