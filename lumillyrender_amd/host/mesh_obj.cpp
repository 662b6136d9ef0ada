// mesh_obj.cpp -- Wavefront OBJ / MTL reader with the behaviour the reference gets from
// tobj 0.1.6 (description.rs:150-197 consumes positions, indices and material_id only):
//   * one model per `o`/`g` group, and a new model whenever `usemtl` changes mid-group;
//   * faces triangulated as a fan from the first vertex (quads -> (0,1,2),(0,2,3));
//   * negative (relative) indices; `v/vt/vn` forms (only the position index is used);
//   * `mtllib` resolved next to the .obj; `Kd` -> diffuse (-> Lambert albedo, description.rs:165-170).
// Only positions are kept: the reference never reads normals or texture coordinates.
#include "host_internal.h"
#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>

namespace lrhost {
namespace {

std::string dirname_of(const std::string& p) {
  size_t k = p.find_last_of('/');
  return k == std::string::npos ? std::string(".") : p.substr(0, k);
}
std::string trim(const std::string& s) {
  size_t a = s.find_first_not_of(" \t\r\n"), b = s.find_last_not_of(" \t\r\n");
  return a == std::string::npos ? std::string() : s.substr(a, b - a + 1);
}

void load_mtl(const std::string& path, ObjFile& out, std::map<std::string, int>& mat_map) {
  std::ifstream f(path);
  if (!f) fail(LR_EIO, "mtl: cannot open `" + path + "`");
  std::string line;
  int cur = -1;
  while (std::getline(f, line)) {
    line = trim(line);
    if (line.empty() || line[0] == '#') continue;
    std::istringstream ss(line);
    std::string key; ss >> key;
    if (key == "newmtl") {
      std::string name; std::getline(ss, name); name = trim(name);
      ObjMaterial m; m.name = name;
      out.materials.push_back(m);
      cur = (int)out.materials.size() - 1;
      mat_map[name] = cur;
    } else if (key == "Kd" && cur >= 0) {
      std::string a, b, c; ss >> a >> b >> c;
      out.materials[cur].diffuse[0] = std::strtof(a.c_str(), nullptr);
      out.materials[cur].diffuse[1] = std::strtof(b.c_str(), nullptr);
      out.materials[cur].diffuse[2] = std::strtof(c.c_str(), nullptr);
    }
  }
}

}  // namespace

ObjFile load_obj(const std::string& path) {
  std::ifstream f(path);
  if (!f) fail(LR_EIO, "obj: cannot open `" + path + "`");
  ObjFile out;
  std::map<std::string, int> mat_map;
  std::vector<float> pos;                       // all `v` records of the file
  ObjModel cur; cur.name = "unnamed_object";
  std::vector<uint32_t> tmp;                    // global vertex indices, 3 per triangle
  int mat_id = -1;
  auto flush = [&]() {
    ObjModel m; m.name = cur.name; m.material_id = mat_id;
    // tobj re-indexes per model; keep that shape (positions used by this model + local indices)
    std::map<uint32_t, uint32_t> remap;
    for (uint32_t g : tmp) {
      auto it = remap.find(g);
      if (it == remap.end()) {
        uint32_t l = (uint32_t)(m.positions.size() / 3);
        remap[g] = l;
        m.positions.push_back(pos[3 * g]); m.positions.push_back(pos[3 * g + 1]); m.positions.push_back(pos[3 * g + 2]);
        m.indices.push_back(l);
      } else m.indices.push_back(it->second);
    }
    out.models.push_back(m);
    tmp.clear();
  };
  std::string line;
  int lineno = 0;
  while (std::getline(f, line)) {
    ++lineno;
    line = trim(line);
    if (line.empty() || line[0] == '#') continue;
    std::istringstream ss(line);
    std::string key; ss >> key;
    if (key == "v") {
      std::string a, b, c; ss >> a >> b >> c;
      if (c.empty()) fail(LR_EIO, "obj: `" + path + "` line " + std::to_string(lineno) + ": bad vertex");
      pos.push_back(std::strtof(a.c_str(), nullptr)); pos.push_back(std::strtof(b.c_str(), nullptr)); pos.push_back(std::strtof(c.c_str(), nullptr));
    } else if (key == "f") {
      std::vector<uint32_t> face;
      std::string tok;
      while (ss >> tok) {
        long vi = std::strtol(tok.c_str(), nullptr, 10);     // stops at '/'
        long nverts = (long)(pos.size() / 3);
        long idx = vi < 0 ? nverts + vi : vi - 1;
        if (vi == 0 || idx < 0 || idx >= nverts) fail(LR_EIO, "obj: `" + path + "` line " + std::to_string(lineno) + ": vertex index out of range");
        face.push_back((uint32_t)idx);
      }
      if (face.size() < 3) fail(LR_EIO, "obj: `" + path + "` line " + std::to_string(lineno) + ": face with fewer than 3 vertices");
      for (size_t i = 1; i + 1 < face.size(); ++i) { tmp.push_back(face[0]); tmp.push_back(face[i]); tmp.push_back(face[i + 1]); }
    } else if (key == "o" || key == "g") {
      if (!tmp.empty()) flush();
      std::string name; std::getline(ss, name); name = trim(name);
      cur.name = name.empty() ? "unnamed_object" : name;
    } else if (key == "usemtl") {
      std::string name; std::getline(ss, name); name = trim(name);
      if (!name.empty()) {
        auto it = mat_map.find(name);
        int new_mat = it == mat_map.end() ? -1 : it->second;
        if (new_mat != mat_id && !tmp.empty()) flush();
        mat_id = new_mat;
      }
    } else if (key == "mtllib") {
      std::string name; std::getline(ss, name); name = trim(name);
      if (!name.empty()) load_mtl(dirname_of(path) + "/" + name, out, mat_map);
    }
    // vt, vn, s, l, ... are irrelevant to the render path
  }
  if (!tmp.empty() || out.models.empty()) flush();
  return out;
}

}  // namespace lrhost
