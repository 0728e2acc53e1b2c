// Device-side value types with WGSL semantics for the N-wide geometry kernels (projection, geometry backward,
// densify).  Column-major matrices, M[c][r]; products evaluated left to right with one rounding per multiply and
// per add (the library is built with -ffp-contract=off).  These kernels are HBM-bound and run once per Gaussian,
// so the arithmetic is written for exactness and readability, not for instruction count.
#pragma once
#include "dmath.h"

struct vec2 { float x, y; };
struct vec3 { float x, y, z; };
struct vec4 { float x, y, z, w; };

WD_DEV vec2 V2(float a, float b) { return vec2{a, b}; }
WD_DEV vec3 V3(float a, float b, float c) { return vec3{a, b, c}; }
WD_DEV vec3 V3(float a) { return vec3{a, a, a}; }
WD_DEV vec4 V4(float a, float b, float c, float d) { return vec4{a, b, c, d}; }
WD_DEV vec4 V4(vec3 v, float d) { return vec4{v.x, v.y, v.z, d}; }
WD_DEV vec3 xyz(vec4 v) { return vec3{v.x, v.y, v.z}; }

WD_DEV vec2 operator+(vec2 a, vec2 b) { return vec2{a.x + b.x, a.y + b.y}; }
WD_DEV vec2 operator-(vec2 a, vec2 b) { return vec2{a.x - b.x, a.y - b.y}; }
WD_DEV vec2 operator*(vec2 a, vec2 b) { return vec2{a.x * b.x, a.y * b.y}; }
WD_DEV vec2 operator+(vec2 a, float b) { return vec2{a.x + b, a.y + b}; }
WD_DEV vec2 operator-(vec2 a, float b) { return vec2{a.x - b, a.y - b}; }
WD_DEV vec2 operator*(vec2 a, float b) { return vec2{a.x * b, a.y * b}; }

WD_DEV vec3 operator+(vec3 a, vec3 b) { return vec3{a.x + b.x, a.y + b.y, a.z + b.z}; }
WD_DEV vec3 operator-(vec3 a, vec3 b) { return vec3{a.x - b.x, a.y - b.y, a.z - b.z}; }
WD_DEV vec3 operator*(vec3 a, vec3 b) { return vec3{a.x * b.x, a.y * b.y, a.z * b.z}; }
WD_DEV vec3 operator+(vec3 a, float b) { return vec3{a.x + b, a.y + b, a.z + b}; }
WD_DEV vec3 operator-(vec3 a, float b) { return vec3{a.x - b, a.y - b, a.z - b}; }
WD_DEV vec3 operator*(vec3 a, float b) { return vec3{a.x * b, a.y * b, a.z * b}; }
WD_DEV vec3 operator*(float a, vec3 b) { return vec3{a * b.x, a * b.y, a * b.z}; }
WD_DEV vec3 operator/(vec3 a, float b) { return vec3{wd_div(a.x, b), wd_div(a.y, b), wd_div(a.z, b)}; }

WD_DEV vec4 operator+(vec4 a, vec4 b) { return vec4{a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
WD_DEV vec4 operator*(vec4 a, float b) { return vec4{a.x * b, a.y * b, a.z * b, a.w * b}; }
WD_DEV vec4 operator/(vec4 a, float b) { return vec4{wd_div(a.x, b), wd_div(a.y, b), wd_div(a.z, b), wd_div(a.w, b)}; }

WD_DEV float dot(vec3 a, vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
WD_DEV float dot(vec4 a, vec4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
WD_DEV vec3 cross(vec3 a, vec3 b) { return vec3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
WD_DEV vec3 normalize(vec3 a) { return a / wd_sqrt(dot(a, a)); }
WD_DEV vec4 normalize(vec4 a) { return a / wd_sqrt(dot(a, a)); }
WD_DEV vec3 vexp(vec3 v) { return vec3{wd_exp(v.x), wd_exp(v.y), wd_exp(v.z)}; }
WD_DEV vec3 vmax(vec3 a, vec3 b) { return vec3{wd_max(a.x, b.x), wd_max(a.y, b.y), wd_max(a.z, b.z)}; }
WD_DEV vec3 vclamp(vec3 v, float lo, float hi) { return vec3{wd_clamp(v.x, lo, hi), wd_clamp(v.y, lo, hi), wd_clamp(v.z, lo, hi)}; }

struct mat3 { vec3 c[3]; };
struct mat4 { vec4 c[4]; };
WD_DEV float el(const vec3& v, int r) { return r == 0 ? v.x : (r == 1 ? v.y : v.z); }
WD_DEV float el(const mat3& m, int c, int r) { return el(m.c[c], r); }
WD_DEV mat3 M3(vec3 a, vec3 b, vec3 c) { return mat3{{a, b, c}}; }
WD_DEV vec3 operator*(const mat3& m, vec3 v) { return m.c[0] * v.x + m.c[1] * v.y + m.c[2] * v.z; }
WD_DEV vec4 operator*(const mat4& m, vec4 v) { return m.c[0] * v.x + m.c[1] * v.y + m.c[2] * v.z + m.c[3] * v.w; }
WD_DEV mat3 operator*(const mat3& a, const mat3& b) { return mat3{{a * b.c[0], a * b.c[1], a * b.c[2]}}; }
WD_DEV mat4 operator*(const mat4& a, const mat4& b) { return mat4{{a * b.c[0], a * b.c[1], a * b.c[2], a * b.c[3]}}; }
WD_DEV mat3 operator*(float s, const mat3& m) { return mat3{{s * m.c[0], s * m.c[1], s * m.c[2]}}; }
WD_DEV mat3 transpose(const mat3& m) {
    return mat3{{vec3{m.c[0].x, m.c[1].x, m.c[2].x}, vec3{m.c[0].y, m.c[1].y, m.c[2].y}, vec3{m.c[0].z, m.c[1].z, m.c[2].z}}};
}
WD_DEV mat4 transpose(const mat4& m) {
    return mat4{{vec4{m.c[0].x, m.c[1].x, m.c[2].x, m.c[3].x}, vec4{m.c[0].y, m.c[1].y, m.c[2].y, m.c[3].y},
                 vec4{m.c[0].z, m.c[1].z, m.c[2].z, m.c[3].z}, vec4{m.c[0].w, m.c[1].w, m.c[2].w, m.c[3].w}}};
}

struct CameraUniforms { mat4 view, view_inv, proj, proj_inv; vec2 viewport, focal; };

// 3D covariance from (w,x,y,z) quaternion and scale: Sigma = (S R)^T (S R), upper triangle.
struct Cov3D { float v[6]; };
WD_DEV mat3 quat_to_R(vec4 q) {
    const float x = q.y, y = q.z, z = q.w, r = q.x;
    return M3(V3(1.0f - 2.0f * (y * y + z * z), 2.0f * (x * y - r * z), 2.0f * (x * z + r * y)),
              V3(2.0f * (x * y + r * z), 1.0f - 2.0f * (x * x + z * z), 2.0f * (y * z - r * x)),
              V3(2.0f * (x * z - r * y), 2.0f * (y * z + r * x), 1.0f - 2.0f * (x * x + y * y)));
}
WD_DEV mat3 diag3(vec3 s) { return M3(V3(s.x, 0.0f, 0.0f), V3(0.0f, s.y, 0.0f), V3(0.0f, 0.0f, s.z)); }
WD_DEV Cov3D covariance3D(vec4 q, vec3 scale) {
    const mat3 M = diag3(scale) * quat_to_R(q);
    const mat3 cov = transpose(M) * M;
    return Cov3D{{cov.c[0].x, cov.c[0].y, cov.c[0].z, cov.c[1].y, cov.c[1].z, cov.c[2].z}};
}
