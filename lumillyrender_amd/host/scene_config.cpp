// scene_config.cpp -- scene file schema: TOML text -> Config.
// Mirrors the serde-derived structs of the reference (scene_loader.rs:8-222): kebab-case keys,
// internally tagged enums on `type`, Option<> fields optional, #[serde(default)] fields optional,
// unknown keys ignored, integers accepted where floats are expected.
#include "host_internal.h"
#include "toml_lite.h"

namespace lrhost {
namespace {

using lrtoml::Value;

[[noreturn]] void bad(const std::string& where, const std::string& what) { fail(LR_EINVAL, "scene: " + where + ": " + what); }

const Value* field(const Value& t, const std::string& where, const char* key, bool required, const char* alt_key = nullptr) {
  const Value* v = t.get(key);
  if (!v && alt_key) v = t.get(alt_key);
  if (!v && required) bad(where, std::string("missing field `") + key + "`");
  return v;
}
float as_f32(const Value& v, const std::string& where) {
  if (!v.is_number()) bad(where, "expected a number");
  return (float)v.number();          // toml parses f64, serde narrows with `as f32`
}
int32_t as_usize(const Value& v, const std::string& where) {
  if (v.kind != Value::INT) bad(where, "expected an integer");
  if (v.i < 0 || v.i > 0x7fffffff) bad(where, "integer out of range");
  return (int32_t)v.i;
}
bool as_bool(const Value& v, const std::string& where) {
  if (v.kind != Value::BOOL) bad(where, "expected a boolean");
  return v.b;
}
std::string as_string(const Value& v, const std::string& where) {
  if (v.kind != Value::STRING) bad(where, "expected a string");
  return v.s;
}
Vec3 as_vec3(const Value& v, const std::string& where) {        // type Vec3 = (f32, f32, f32)
  if (v.kind != Value::ARRAY || v.arr.size() != 3) bad(where, "expected an array of 3 numbers");
  return vec3(as_f32(*v.arr[0], where), as_f32(*v.arr[1], where), as_f32(*v.arr[2], where));
}
const Value& as_table(const Value& v, const std::string& where) {
  if (v.kind != Value::TABLE) bad(where, "expected a table");
  return v;
}
std::string tag_of(const Value& t, const std::string& where) {
  return as_string(*field(t, where, "type", true), where + ".type");
}

std::vector<Transform> parse_transforms(const Value* v, const std::string& where) {
  std::vector<Transform> out;
  if (!v) return out;                                           // #[serde(default)]
  if (v->kind != Value::ARRAY) bad(where, "expected an array of tables");
  for (size_t i = 0; i < v->arr.size(); ++i) {
    std::string w = where + "[" + std::to_string(i) + "]";
    const Value& t = as_table(*v->arr[i], w);
    std::string tag = tag_of(t, w);
    Transform tr;
    if (tag == "translate") { tr.kind = Transform::TRANSLATE; tr.a = as_vec3(*field(t, w, "vector", true), w + ".vector"); }
    else if (tag == "scale") { tr.kind = Transform::SCALE; tr.a = as_vec3(*field(t, w, "vector", true), w + ".vector"); }
    else if (tag == "axis-angle") {
      tr.kind = Transform::AXIS_ANGLE;
      tr.a = as_vec3(*field(t, w, "axis", true), w + ".axis");
      tr.angle = as_f32(*field(t, w, "angle", true), w + ".angle");
    } else if (tag == "look-at") {
      tr.kind = Transform::LOOK_AT;
      tr.a = as_vec3(*field(t, w, "origin", true), w + ".origin");
      tr.b = as_vec3(*field(t, w, "target", true), w + ".target");
      tr.c = as_vec3(*field(t, w, "up", true), w + ".up");
    } else bad(w, "unknown transform type `" + tag + "`");
    out.push_back(tr);
  }
  return out;
}

}  // namespace

Mat4 Transform::matrix() const {                                 // scene_loader.rs:88-97
  switch (kind) {
    case TRANSLATE: return Mat4::translate(a);
    case SCALE: return Mat4::scale(a);
    case AXIS_ANGLE: return Mat4::axis_angle(a, angle * kPi / 180.0f);
    case LOOK_AT: return Mat4::look_at(a, b, c);
  }
  return Mat4::unit();
}
Mat4 compose(const std::vector<Transform>& ts) {                 // fold(unit, |p, c| c * p)
  Mat4 p = Mat4::unit();
  for (const Transform& t : ts) p = mul(t.matrix(), p);
  return p;
}

Config parse_config(const std::string& text) {
  lrtoml::ValuePtr rootp;
  try { rootp = lrtoml::parse(text); }
  catch (const lrtoml::ParseError& e) { fail(LR_EINVAL, e.what()); }
  const Value& root = *rootp;
  Config c;

  {  // [renderer]  scene_loader.rs:8-18
    const Value& t = as_table(*field(root, "root", "renderer", true), "renderer");
    c.renderer.samples = as_usize(*field(t, "renderer", "samples", true), "renderer.samples");
    const Value* v;
    c.renderer.depth = (v = field(t, "renderer", "depth", false)) ? as_usize(*v, "renderer.depth") : 5;                        // description.rs:75
    c.renderer.depth_limit = (v = field(t, "renderer", "depth-limit", false)) ? as_usize(*v, "renderer.depth-limit") : 64;       // description.rs:76
    c.renderer.no_direct_emitter = (v = field(t, "renderer", "no-direct-emitter", false)) ? (as_bool(*v, "renderer.no-direct-emitter") ? 1 : 0) : 0;
    c.renderer.threads = (v = field(t, "renderer", "threads", false)) ? as_usize(*v, "renderer.threads") : 0;
    c.renderer.integrator = LR_INTEGRATOR_PT_DIRECT;                                                                          // main.rs:66
    if ((v = field(t, "renderer", "integrator", false))) {
      c.has_integrator = true; c.integrator_name = as_string(*v, "renderer.integrator");
      if (c.integrator_name == "pt") c.renderer.integrator = LR_INTEGRATOR_PT;
      else if (c.integrator_name == "pt-direct") c.renderer.integrator = LR_INTEGRATOR_PT_DIRECT;
      else bad("renderer.integrator", "Unknown integrator type `" + c.integrator_name + "`");                                 // main.rs:124
    }
  }
  {  // [film]  scene_loader.rs:20-27
    const Value& t = as_table(*field(root, "root", "film", true), "film");
    const Value& r = *field(t, "film", "resolution", true);
    if (r.kind != Value::ARRAY || r.arr.size() != 2) bad("film.resolution", "expected [width, height]");
    c.film.resolution[0] = as_usize(*r.arr[0], "film.resolution"); c.film.resolution[1] = as_usize(*r.arr[1], "film.resolution");
    std::string o = as_string(*field(t, "film", "output", true), "film.output");
    if (o == "png") c.film.output = LR_OUTPUT_PNG;
    else if (o == "hdr") c.film.output = LR_OUTPUT_HDR;
    else bad("film.output", "Unsupported output type `" + o + "`");                                                           // main.rs:165-167
    const Value* v;
    c.film.gamma = 2.2f;                                                                                                      // main.rs:136
    if ((v = field(t, "film", "gamma", false))) { c.film.gamma = as_f32(*v, "film.gamma"); c.has_gamma = true; }
    c.film.sensitivity[0] = c.film.sensitivity[1] = c.film.sensitivity[2] = 1.0f;
    if ((v = field(t, "film", "sensitivity", false))) { Vec3 s = as_vec3(*v, "film.sensitivity"); c.film.sensitivity[0] = s.x; c.film.sensitivity[1] = s.y; c.film.sensitivity[2] = s.z; }
  }
  if (const Value* sv = field(root, "root", "sky", false)) {   // scene_loader.rs:29-42
    const Value& t = as_table(*sv, "sky");
    std::string tag = tag_of(t, "sky");
    c.sky.present = true;
    if (tag == "uniform") { c.sky.type = LR_SKY_UNIFORM; c.sky.color = as_vec3(*field(t, "sky", "color", true), "sky.color"); }
    else if (tag == "ibl") {
      c.sky.type = LR_SKY_IBL;
      c.sky.path = as_string(*field(t, "sky", "path", true), "sky.path");
      const Value* lo = field(t, "sky", "longitude-offset", false, "longitude_offset");
      c.sky.longitude_offset = lo ? as_f32(*lo, "sky.longitude-offset") : 0.0f;
    } else bad("sky", "unknown sky type `" + tag + "`");
  }
  {  // [camera]  scene_loader.rs:106-125
    const Value& t = as_table(*field(root, "root", "camera", true), "camera");
    std::string tag = tag_of(t, "camera");
    if (tag == "ideal-pinhole") {
      c.camera.type = LR_CAMERA_IDEAL_PINHOLE;
      c.camera.fov = as_f32(*field(t, "camera", "fov", true), "camera.fov");
      c.camera.transform = parse_transforms(field(t, "camera", "transform", false), "camera.transform");
    } else if (tag == "thin-lens") {
      c.camera.type = LR_CAMERA_THIN_LENS;
      c.camera.fov = as_f32(*field(t, "camera", "fov", true), "camera.fov");
      // the v0 schema is kebab-case; scenes/welcome-2018.toml spells these with underscores, so both are accepted
      c.camera.focus_distance = as_f32(*field(t, "camera", "focus-distance", true, "focus_distance"), "camera.focus-distance");
      c.camera.f_number = as_f32(*field(t, "camera", "f-number", true, "f_number"), "camera.f-number");
      c.camera.transform = parse_transforms(field(t, "camera", "transform", false), "camera.transform");
    } else if (tag == "omnidirectional") {
      c.camera.type = LR_CAMERA_OMNIDIRECTIONAL;
      c.camera.transform = parse_transforms(field(t, "camera", "transform", true), "camera.transform");
    } else bad("camera", "unknown camera type `" + tag + "`");
  }
  if (const Value* lv = field(root, "root", "light", false)) {   // scene_loader.rs:44-52
    if (lv->kind != Value::ARRAY) bad("light", "expected an array of tables");
    for (size_t i = 0; i < lv->arr.size(); ++i) {
      std::string w = "light[" + std::to_string(i) + "]";
      const Value& t = as_table(*lv->arr[i], w);
      std::string tag = tag_of(t, w);
      if (tag != "area") bad(w, "unknown light type `" + tag + "`");
      LightCfg l;
      l.object = as_string(*field(t, w, "object", true), w + ".object");
      l.emission = as_vec3(*field(t, w, "emission", true), w + ".emission");
      if (const Value* iv = field(t, w, "intensity", false)) { l.has_intensity = true; l.intensity = as_f32(*iv, w + ".intensity"); }
      c.light.push_back(l);
    }
  }
  if (const Value* ov = field(root, "root", "object", false)) {  // scene_loader.rs:54-62
    if (ov->kind != Value::ARRAY) bad("object", "expected an array of tables");
    for (size_t i = 0; i < ov->arr.size(); ++i) {
      std::string w = "object[" + std::to_string(i) + "]";
      const Value& t = as_table(*ov->arr[i], w);
      ObjectCfg o;
      if (const Value* nv = field(t, w, "name", false)) { o.has_name = true; o.name = as_string(*nv, w + ".name"); }
      o.mesh = as_string(*field(t, w, "mesh", true), w + ".mesh");
      if (const Value* mv = field(t, w, "material", false)) { o.has_material = true; o.material = as_string(*mv, w + ".material"); }
      o.transform = parse_transforms(field(t, w, "transform", false), w + ".transform");
      c.object.push_back(o);
    }
  }
  if (const Value* mv = field(root, "root", "material", false)) {   // scene_loader.rs:143-173
    if (mv->kind != Value::ARRAY) bad("material", "expected an array of tables");
    for (size_t i = 0; i < mv->arr.size(); ++i) {
      std::string w = "material[" + std::to_string(i) + "]";
      const Value& t = as_table(*mv->arr[i], w);
      std::string tag = tag_of(t, w);
      MaterialCfg m;
      m.name = as_string(*field(t, w, "name", true), w + ".name");
      if (tag == "lambert") { m.type = LR_MAT_LAMBERT; m.color = as_vec3(*field(t, w, "albedo", true), w + ".albedo"); }
      else if (tag == "phong" || tag == "blinn-phong") {
        m.type = tag == "phong" ? LR_MAT_PHONG : LR_MAT_BLINN_PHONG;
        m.color = as_vec3(*field(t, w, "reflectance", true), w + ".reflectance");
        m.p0 = as_f32(*field(t, w, "alpha", true), w + ".alpha");
      } else if (tag == "ggx") {
        m.type = LR_MAT_GGX;
        m.color = as_vec3(*field(t, w, "reflectance", true), w + ".reflectance");
        m.p0 = as_f32(*field(t, w, "roughness", true), w + ".roughness");
        m.p1 = as_f32(*field(t, w, "ior", true), w + ".ior");
      } else if (tag == "ideal-refraction") {
        m.type = LR_MAT_IDEAL_REFRACTION;
        m.color = as_vec3(*field(t, w, "reflectance", true), w + ".reflectance");
        const Value* av = field(t, w, "absorbtance", false);
        m.p1 = av ? as_f32(*av, w + ".absorbtance") : 0.0f;
        m.p0 = as_f32(*field(t, w, "ior", true), w + ".ior");
      } else bad(w, "unknown material type `" + tag + "`");
      c.material.push_back(m);
    }
  }
  if (const Value* mv = field(root, "root", "mesh", false)) {   // scene_loader.rs:186-205
    if (mv->kind != Value::ARRAY) bad("mesh", "expected an array of tables");
    for (size_t i = 0; i < mv->arr.size(); ++i) {
      std::string w = "mesh[" + std::to_string(i) + "]";
      const Value& t = as_table(*mv->arr[i], w);
      std::string tag = tag_of(t, w);
      MeshCfg m;
      m.name = as_string(*field(t, w, "name", true), w + ".name");
      if (tag == "obj") { m.is_obj = true; m.path = as_string(*field(t, w, "path", true), w + ".path"); }
      else if (tag == "sphere") { m.is_obj = false; m.radius = as_f32(*field(t, w, "radius", true), w + ".radius"); }
      else bad(w, "unknown mesh type `" + tag + "`");
      c.mesh.push_back(m);
    }
  }
  return c;
}

}  // namespace lrhost
