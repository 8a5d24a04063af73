// Host-side vector / matrix helpers for the scene loader.
//
// The reference builds its transforms with cyCodeBase (cyVector.h / cyMatrix.h), which is absent
// from the reference checkout.  The operation order below is therefore this
// build's own definition of those conventions: column-major 3x4 matrices, post-multiplied column
// vectors, Rodrigues rotation, cofactor inverse.  Only the loader uses it; the renderer core and
// the oracle consume the resulting matrices as data, so both see identical numbers.
#pragma once
#include <cmath>
#include <cstring>
#include "rtcore.h"

namespace rth {

struct V3 {
  float x = 0, y = 0, z = 0;
  V3() = default;
  V3(float a, float b, float c) : x(a), y(b), z(c) {}
  float& operator[](int i) { return (&x)[i]; }
  float operator[](int i) const { return (&x)[i]; }
};
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline V3 operator/(V3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
inline V3 operator-(V3 a) { return {-a.x, -a.y, -a.z}; }
inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline float length(V3 a) { return std::sqrt(dot(a, a)); }
inline V3 normalized(V3 a) { return a / length(a); }

// column-major 3x4
struct M34 {
  float c[12];
  M34() { set_identity(); }
  void set_identity() {
    std::memset(c, 0, sizeof c);
    c[0] = c[4] = c[8] = 1.0f;
  }
  static M34 translation(V3 t) {
    M34 m;
    m.c[9] = t.x; m.c[10] = t.y; m.c[11] = t.z;
    return m;
  }
  static M34 scale(V3 s) {
    M34 m;
    m.c[0] = s.x; m.c[4] = s.y; m.c[8] = s.z;
    return m;
  }
  // Rotation about a unit axis by `angle` radians (Rodrigues).
  static M34 rotation(V3 axis, float angle) {
    float s = std::sin(angle), co = std::cos(angle);
    float t = 1.0f - co;
    V3 a = axis * t;
    float txy = a.x * axis.y, txz = a.x * axis.z, tyz = a.y * axis.z;
    V3 sv = axis * s;
    M34 m;
    m.c[0] = a.x * axis.x + co; m.c[1] = txy + sv.z;        m.c[2] = txz - sv.y;
    m.c[3] = txy - sv.z;        m.c[4] = a.y * axis.y + co; m.c[5] = tyz + sv.x;
    m.c[6] = txz + sv.y;        m.c[7] = tyz - sv.x;        m.c[8] = a.z * axis.z + co;
    m.c[9] = m.c[10] = m.c[11] = 0.0f;
    return m;
  }
  V3 col(int i) const { return {c[3 * i], c[3 * i + 1], c[3 * i + 2]}; }
  // point transform
  V3 mul_point(V3 p) const {
    return {c[0] * p.x + c[3] * p.y + c[6] * p.z + c[9], c[1] * p.x + c[4] * p.y + c[7] * p.z + c[10],
            c[2] * p.x + c[5] * p.y + c[8] * p.z + c[11]};
  }
  V3 mul_dir(V3 p) const {
    return {c[0] * p.x + c[3] * p.y + c[6] * p.z, c[1] * p.x + c[4] * p.y + c[7] * p.z,
            c[2] * p.x + c[5] * p.y + c[8] * p.z};
  }
  // (this * r): apply r first, then this.
  M34 operator*(const M34& r) const {
    M34 o;
    for (int j = 0; j < 3; j++) {
      V3 v = mul_dir(r.col(j));
      o.c[3 * j] = v.x; o.c[3 * j + 1] = v.y; o.c[3 * j + 2] = v.z;
    }
    V3 t = mul_point(r.col(3));
    o.c[9] = t.x; o.c[10] = t.y; o.c[11] = t.z;
    return o;
  }
  M34 inverse() const {
    // cofactor inverse of the 3x3 block, then -inv*t
    float a00 = c[0], a10 = c[1], a20 = c[2];
    float a01 = c[3], a11 = c[4], a21 = c[5];
    float a02 = c[6], a12 = c[7], a22 = c[8];
    float c00 = a11 * a22 - a12 * a21;
    float c01 = a12 * a20 - a10 * a22;
    float c02 = a10 * a21 - a11 * a20;
    float det = a00 * c00 + a01 * c01 + a02 * c02;
    float id = 1.0f / det;
    M34 o;
    o.c[0] = c00 * id;
    o.c[1] = c01 * id;
    o.c[2] = c02 * id;
    o.c[3] = (a02 * a21 - a01 * a22) * id;
    o.c[4] = (a00 * a22 - a02 * a20) * id;
    o.c[5] = (a01 * a20 - a00 * a21) * id;
    o.c[6] = (a01 * a12 - a02 * a11) * id;
    o.c[7] = (a02 * a10 - a00 * a12) * id;
    o.c[8] = (a00 * a11 - a01 * a10) * id;
    V3 t = o.mul_dir(col(3));
    o.c[9] = -t.x; o.c[10] = -t.y; o.c[11] = -t.z;
    return o;
  }
  void store(rt_mat34& d) const { std::memcpy(d.m, c, sizeof c); }
};

}  // namespace rth
