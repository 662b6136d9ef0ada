// host_internal.h -- declarations shared by the host translation units.
#pragma once
#include <string>
#include <vector>
#include <cstdint>
#include "../../include/lumilly_host.h"
#include "host_math.h"

namespace lrhost {

struct HostError { int code; std::string msg; };
[[noreturn]] void fail(int code, const std::string& msg);
void set_last_error(const std::string& msg);

// ---- Config (scene_loader.rs:8-222) ---------------------------------------------------------
struct Transform {                    // scene_loader.rs:68-97
  enum Kind { TRANSLATE, SCALE, AXIS_ANGLE, LOOK_AT } kind;
  Vec3 a, b, c;                       // vector | axis | origin,target,up
  float angle = 0.0f;                 // degrees
  Mat4 matrix() const;
};
Mat4 compose(const std::vector<Transform>& ts);   // HasTransform::matrix scene_loader.rs:99-104

struct CameraCfg {                    // scene_loader.rs:106-125
  int type = LR_CAMERA_IDEAL_PINHOLE;
  float fov = 0, focus_distance = 0, f_number = 0;
  std::vector<Transform> transform;
};
struct SkyCfg { bool present = false; int type = LR_SKY_UNIFORM; Vec3 color; std::string path; float longitude_offset = 0; };
struct LightCfg { std::string object; Vec3 emission; bool has_intensity = false; float intensity = 1.0f; };
struct ObjectCfg { bool has_name = false; std::string name, mesh; bool has_material = false; std::string material; std::vector<Transform> transform; };
struct MaterialCfg { int type = LR_MAT_LAMBERT; std::string name; Vec3 color; float p0 = 0, p1 = 0; };
struct MeshCfg { bool is_obj = true; std::string name, path; float radius = 0; };
struct Config {                       // scene_loader.rs:207-222
  LrRendererConfig renderer;
  bool has_integrator = false; std::string integrator_name;
  LrFilmConfig film;
  bool has_gamma = false;
  SkyCfg sky;
  CameraCfg camera;
  std::vector<LightCfg> light;
  std::vector<ObjectCfg> object;
  std::vector<MaterialCfg> material;
  std::vector<MeshCfg> mesh;
};
Config parse_config(const std::string& toml_text);

// ---- OBJ / MTL (stands in for tobj 0.1.6: description.rs:150-162) ------------------------------
struct ObjModel {
  std::string name;
  std::vector<float> positions;       // xyz per vertex (only the vertices this model uses, tobj-style)
  std::vector<uint32_t> indices;      // 3 per triangle
  int material_id = -1;               // index into ObjFile::materials, -1 = none
};
struct ObjMaterial { std::string name; float diffuse[3] = {0, 0, 0}; };
struct ObjFile { std::vector<ObjModel> models; std::vector<ObjMaterial> materials; };
ObjFile load_obj(const std::string& path);

// ---- BVH ---------------------------------------------------------------------------------------
struct BvhResult {
  std::vector<LrBvhNode> nodes;
  std::vector<int32_t> order;
  int max_depth = 0;
  double sah_cost = 0.0;
  double seconds = 0.0;
  float pad = 0.0f;
};
BvhResult build_bvh(const LrPrimitive* prims, int n, int max_leaf, const float* extra_point);

// ---- images --------------------------------------------------------------------------------------
void load_hdr(const std::string& path, std::vector<float>& texels, int& w, int& h);
void save_hdr(const std::string& path, const float* rgb, int w, int h, size_t stride);
void save_png(const std::string& path, const float* rgb, int w, int h, size_t stride, float gamma);
uint8_t to_color(float x, float gamma);
void write_png_rgb8(const std::string& path, const uint8_t* rgb8, int w, int h, size_t stride);
void write_hdr_rgbe(const std::string& path, const uint8_t* rgbe, int w, int h, size_t stride);

std::string resolve_path(const std::string& path, const std::string& asset_root);

// ---- camera constructors (camera.rs) ------------------------------------------------------------
LrCamera make_camera(const CameraCfg& cfg, int width, int height);

}  // namespace lrhost

struct LrHostScene {
  lrhost::Config config;
  std::string asset_root;
  std::vector<LrMaterial> materials;
  std::vector<LrPrimitive> prims;
  std::vector<float> texels;
  int sky_w = 0, sky_h = 0;
  lrhost::BvhResult bvh;
  LrSceneDesc desc;
};
