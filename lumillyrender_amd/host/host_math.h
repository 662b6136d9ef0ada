// host_math.h -- f32 vector / matrix helpers for scene instantiation.  Operation order follows
// the reference (math/vector3.rs, math/vector4.rs, math/matrix4.rs) because the results feed the
// device as vertex positions and camera bases.  Compiled with -ffp-contract=off.
#pragma once
#include <cmath>

namespace lrhost {

const float kPi = 3.14159265358979323846264338327950288f;   // constant.rs:1

struct Vec3 { float x = 0, y = 0, z = 0; };
inline Vec3 vec3(float x, float y, float z) { Vec3 v; v.x = x; v.y = y; v.z = z; return v; }
inline Vec3 operator+(Vec3 a, Vec3 b) { return vec3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline Vec3 operator-(Vec3 a, Vec3 b) { return vec3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline Vec3 operator*(Vec3 a, float s) { return vec3(a.x * s, a.y * s, a.z * s); }
inline Vec3 operator/(Vec3 a, float s) { return vec3(a.x / s, a.y / s, a.z / s); }
inline float dot(Vec3 a, Vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline Vec3 cross(Vec3 a, Vec3 b) { return vec3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
inline float norm(Vec3 a) { return std::sqrt(dot(a, a)); }
inline Vec3 normalize(Vec3 a) { return a / norm(a); }

// Row-major 4x4 (matrix4.rs:4-6)
struct Mat4 {
  float v[16];
  static Mat4 unit() {                                        // matrix4.rs:9-18
    Mat4 m = {{1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1}};
    return m;
  }
  static Mat4 translate(Vec3 t) {                             // matrix4.rs:20-29
    Mat4 m = {{1, 0, 0, t.x, 0, 1, 0, t.y, 0, 0, 1, t.z, 0, 0, 0, 1}};
    return m;
  }
  static Mat4 scale(Vec3 s) {                                 // matrix4.rs:31-40
    Mat4 m = {{s.x, 0, 0, 0, 0, s.y, 0, 0, 0, 0, s.z, 0, 0, 0, 0, 1}};
    return m;
  }
  static Mat4 axis_angle(Vec3 a, float t) {                   // matrix4.rs:42-54 (Rodrigues)
    float c = std::cos(t), s = std::sin(t);
    Mat4 m = {{
        c + a.x * a.x * (1.0f - c), a.x * a.y * (1.0f - c) - a.z * s, a.x * a.z * (1.0f - c) + a.y * s, 0.0f,
        a.y * a.x * (1.0f - c) + a.z * s, c + a.y * a.y * (1.0f - c), a.y * a.z * (1.0f - c) - a.x * s, 0.0f,
        a.z * a.x * (1.0f - c) - a.y * s, a.z * a.y * (1.0f - c) + a.x * s, c + a.z * a.z * (1.0f - c), 0.0f,
        0.0f, 0.0f, 0.0f, 1.0f}};
    return m;
  }
  // matrix4.rs:56-68: rows 0-2 are the camera basis, ROW 3 is the origin (not a column)
  static Mat4 look_at(Vec3 origin, Vec3 target, Vec3 up) {
    Vec3 za = normalize(origin - target);
    Vec3 xa = normalize(cross(up, za));
    Vec3 ya = cross(za, xa);
    Mat4 m = {{xa.x, xa.y, xa.z, 0.0f, ya.x, ya.y, ya.z, 0.0f, za.x, za.y, za.z, 0.0f, origin.x, origin.y, origin.z, 1.0f}};
    return m;
  }
  Vec3 row3(int r) const { return vec3(v[4 * r], v[4 * r + 1], v[4 * r + 2]); }   // matrix4.rs:74-76 + From<Vector4>
};

inline float dot4(const float* r, float x, float y, float z, float w) {    // vector4.rs:73-75
  return r[0] * x + r[1] * y + r[2] * z + r[3] * w;
}
// Matrix4 * Vector3 (matrix4.rs:193-199): w = 1, keeps xyz of the 4 row dots
inline Vec3 mul(const Mat4& m, Vec3 p) {
  return vec3(dot4(m.v + 0, p.x, p.y, p.z, 1.0f), dot4(m.v + 4, p.x, p.y, p.z, 1.0f), dot4(m.v + 8, p.x, p.y, p.z, 1.0f));
}
// Matrix4 * Matrix4 (matrix4.rs:201-222): rows of lhs dot columns of rhs
inline Mat4 mul(const Mat4& a, const Mat4& b) {
  Mat4 r;
  for (int y = 0; y < 4; ++y)
    for (int x = 0; x < 4; ++x)
      r.v[4 * y + x] = a.v[4 * y] * b.v[x] + a.v[4 * y + 1] * b.v[4 + x] + a.v[4 * y + 2] * b.v[8 + x] + a.v[4 * y + 3] * b.v[12 + x];
  return r;
}

}  // namespace lrhost
