// ORACLE (test infrastructure only) -- WGSL value-semantics shim for the CPU restatement.
//
// Nothing under oracle/ is product code: only tests/, __graft_entry__.smoke() and
// bench.py's cpu_baseline leg may load it.  PARITY UNPINNED: the reference
// (krispy-kenay/WebDGS) ships no tests, golden vectors or runnable CPU path
// (SURVEY.md section 8c), so this restatement is pinned only by line-by-line review
// against the cited WGSL and by its own self-consistency tests.
//
// What this header fixes, where WGSL leaves behaviour implementation-defined:
//  * matrices are column-major, M[c][r] (WGSL spec 6.2.6); products are evaluated
//    left-to-right with one rounding per multiply and per add (no FMA contraction;
//    the file is compiled with -ffp-contract=off) except where fmaf() is written out.
//  * exp/log are the deterministic algorithms documented in DESIGN.md ("dmath"):
//    every step is an IEEE-754 binary32 operation, so a GPU implementation that
//    performs the same steps is bit-identical.  (WGSL only bounds exp to 3+2|x| ULP
//    and log to 3 ULP; these are within ~1 ULP, checked in tests/test_oracle_math.py.)
//  * sqrt and '/' are correctly rounded; inverseSqrt(x) = 1/sqrt(x); normalize(v) = v/length(v).
//  * pack2x16float rounds to nearest even, overflows to infinity, keeps subnormals.
//  * f32->u32 / f32->i32 conversions truncate toward zero and saturate (NaN -> 0).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

namespace wgsl {

typedef uint32_t u32;
typedef int32_t i32;
typedef float f32;

static inline u32 f2bits(f32 f) { u32 u; std::memcpy(&u, &f, 4); return u; }
static inline f32 bits2f(u32 u) { f32 f; std::memcpy(&f, &u, 4); return f; }

// ---------------------------------------------------------------- dmath: exp
// exp(x) = 2^n * P(r), n = rint(x*log2e), r = x - n*ln2 (two-step, FMA),
// P(r) = 1 + r*(1 + r*(c2 + r*(c3 + r*(c4 + r*(c5 + r*c6))))) by FMA Horner.
// x < -86 -> +0 and x > 88 -> +inf, so n is in [-124, 127] and the single multiply by 2^n is exact;  NaN -> NaN.
static inline f32 wd_exp(f32 x) {
    const f32 LOG2E  = bits2f(0x3FB8AA3Bu);   // 1.44269502
    const f32 LN2_HI = bits2f(0x3F318000u);   // 0.693359375
    const f32 LN2_LO = bits2f(0xB95E8083u);   // -2.12194440e-4
    const f32 C2 = bits2f(1056964604u), C3 = bits2f(1042983495u), C4 = bits2f(1026207148u),
              C5 = bits2f(1007230415u), C6 = bits2f(984890875u);
    if (x != x) return x;
    if (x < -86.0f) return 0.0f;
    if (x > 88.0f) return INFINITY;
    f32 n = std::nearbyintf(x * LOG2E);
    f32 r = std::fmaf(-n, LN2_HI, x);
    r = std::fmaf(-n, LN2_LO, r);
    f32 p = std::fmaf(C6, r, C5);
    p = std::fmaf(p, r, C4);
    p = std::fmaf(p, r, C3);
    p = std::fmaf(p, r, C2);
    p = std::fmaf(p, r, 1.0f);
    p = std::fmaf(p, r, 1.0f);
    i32 ni = (i32)n;                       // -124 <= n <= 127
    return p * bits2f((u32)(ni + 127) << 23);
}

// ---------------------------------------------------------------- dmath: log
// Cephes-style logf: x = m*2^e, m in [sqrt(1/2), sqrt(2)); log(m) by a degree-8
// polynomial in (m-1); all steps binary32 with explicit FMA.
static inline f32 wd_log(f32 x) {
    if (x != x) return x;
    if (x < 0.0f) return NAN;
    if (x == 0.0f) return -INFINITY;
    if (x == INFINITY) return x;
    i32 eadj = 0;
    if (x < bits2f(0x00800000u)) { x = x * 8388608.0f; eadj = -23; }   // subnormal: scale by 2^23
    u32 b = f2bits(x);
    i32 e = (i32)((b >> 23) & 0xFFu) - 126 + eadj;                       // x = m*2^e, m in [0.5,1)
    f32 m = bits2f((b & 0x007FFFFFu) | 0x3F000000u);
    if (m < 0.707106781186547524f) { e = e - 1; m = m + m; }
    m = m - 1.0f;
    f32 z = m * m;
    f32 y = 7.0376836292E-2f;
    y = std::fmaf(y, m, -1.1514610310E-1f);
    y = std::fmaf(y, m, 1.1676998740E-1f);
    y = std::fmaf(y, m, -1.2420140846E-1f);
    y = std::fmaf(y, m, 1.4249322787E-1f);
    y = std::fmaf(y, m, -1.6668057665E-1f);
    y = std::fmaf(y, m, 2.0000714765E-1f);
    y = std::fmaf(y, m, -2.4999993993E-1f);
    y = std::fmaf(y, m, 3.3333331174E-1f);
    y = (y * m) * z;
    f32 fe = (f32)e;
    y = std::fmaf(fe, -2.12194440e-4f, y);
    y = std::fmaf(-0.5f, z, y);
    f32 r = m + y;
    r = std::fmaf(fe, 0.693359375f, r);
    return r;
}

// Evaluation-order switch for the three expressions whose contraction the oracle pins (raster q / power, C += c*alpha*vis):
// 0 (default) = the pinned FMA order the HIP kernels reproduce bit for bit; 1 = the WGSL source read literally, left to right, one
// rounding per operator, no FMA (tiled-rasterizer.wgsl:228-238, tiled-backward-rasterize.wgsl:108-110).  Both are legal WGSL
// evaluations; tests/test_oracle_tolerance.py measures how far apart their results are (the "stated fp32 tolerance").
extern int g_literal_order;
static inline f32 wd_sqrt(f32 x) { return std::sqrt(x); }          // correctly rounded
static inline f32 wd_inverseSqrt(f32 x) { return 1.0f / std::sqrt(x); }

// WGSL u32(f32) / i32(f32): truncate toward zero, saturate, NaN -> 0.
static inline u32 to_u32(f32 v) {
    if (!(v > 0.0f)) return 0u;
    if (v >= 4294967296.0f) return 0xFFFFFFFFu;
    return (u32)v;
}
static inline i32 to_i32(f32 v) {
    if (v != v) return 0;
    if (v >= 2147483648.0f) return 2147483647;
    if (v <= -2147483648.0f) return (i32)0x80000000u;
    return (i32)v;
}

// ---------------------------------------------------------------- fp16
static inline uint16_t f32_to_f16(f32 f) {
    u32 x = f2bits(f);
    u32 sign = (x >> 16) & 0x8000u;
    u32 ax = x & 0x7FFFFFFFu;
    if (ax > 0x7F800000u) return (uint16_t)(sign | 0x7E00u);          // NaN
    if (ax >= 0x477FF000u) return (uint16_t)(sign | 0x7C00u);         // >= 65520 -> inf
    if (ax < 0x33000001u) return (uint16_t)sign;                       // <= 2^-25 -> 0 (ties to even)
    if (ax < 0x38800000u) {                                            // subnormal half
        u32 mant = (ax & 0x007FFFFFu) | 0x00800000u;
        int shift = 126 - (int)(ax >> 23);                             // 14..24
        u32 hm = mant >> shift;
        u32 rem = mant & ((1u << shift) - 1u);
        u32 half = 1u << (shift - 1);
        if (rem > half || (rem == half && (hm & 1u))) hm++;
        return (uint16_t)(sign | hm);
    }
    u32 e = (ax >> 23) - 112u;
    u32 m = (ax >> 13) & 0x3FFu;
    u32 rem = ax & 0x1FFFu;
    u32 h = (e << 10) | m;
    if (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) h++;            // may carry into exponent (correct)
    return (uint16_t)(sign | h);
}
static inline f32 f16_to_f32(uint16_t h) {
    u32 sign = ((u32)h & 0x8000u) << 16;
    u32 e = (h >> 10) & 0x1Fu, m = h & 0x3FFu;
    if (e == 0) {
        if (m == 0) return bits2f(sign);
        f32 v = (f32)m * bits2f(0x33800000u);                          // m * 2^-24
        return bits2f(f2bits(v) | sign);
    }
    if (e == 31) return bits2f(sign | 0x7F800000u | (m << 13));
    return bits2f(sign | ((e + 112u) << 23) | (m << 13));
}

// ---------------------------------------------------------------- vectors
struct vec2 { f32 x, y; f32& operator[](int i) { return (&x)[i]; } f32 operator[](int i) const { return (&x)[i]; } };
struct vec3 { f32 x, y, z; f32& operator[](int i) { return (&x)[i]; } f32 operator[](int i) const { return (&x)[i]; } };
struct vec4 { f32 x, y, z, w; f32& operator[](int i) { return (&x)[i]; } f32 operator[](int i) const { return (&x)[i]; }
              vec3 xyz() const { return vec3{x, y, z}; } };

static inline vec2 V2(f32 a, f32 b) { return vec2{a, b}; }
static inline vec2 V2(f32 a) { return vec2{a, a}; }
static inline vec3 V3(f32 a, f32 b, f32 c) { return vec3{a, b, c}; }
static inline vec3 V3(f32 a) { return vec3{a, a, a}; }
static inline vec4 V4(f32 a, f32 b, f32 c, f32 d) { return vec4{a, b, c, d}; }
static inline vec4 V4(vec3 v, f32 d) { return vec4{v.x, v.y, v.z, d}; }

#define WGSL_VOP(T, N, OP)                                                              \
    static inline T operator OP(T a, T b) { T r; for (int i = 0; i < N; i++) r[i] = a[i] OP b[i]; return r; } \
    static inline T operator OP(T a, f32 b) { T r; for (int i = 0; i < N; i++) r[i] = a[i] OP b; return r; }  \
    static inline T operator OP(f32 a, T b) { T r; for (int i = 0; i < N; i++) r[i] = a OP b[i]; return r; }
WGSL_VOP(vec2, 2, +) WGSL_VOP(vec2, 2, -) WGSL_VOP(vec2, 2, *) WGSL_VOP(vec2, 2, /)
WGSL_VOP(vec3, 3, +) WGSL_VOP(vec3, 3, -) WGSL_VOP(vec3, 3, *) WGSL_VOP(vec3, 3, /)
WGSL_VOP(vec4, 4, +) WGSL_VOP(vec4, 4, -) WGSL_VOP(vec4, 4, *) WGSL_VOP(vec4, 4, /)
#undef WGSL_VOP
static inline vec3 operator-(vec3 a) { return vec3{-a.x, -a.y, -a.z}; }

static inline f32 dot(vec2 a, vec2 b) { return a.x * b.x + a.y * b.y; }
static inline f32 dot(vec3 a, vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline f32 dot(vec4 a, vec4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
static inline vec3 cross(vec3 a, vec3 b) { return vec3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
static inline f32 length(vec3 a) { return wd_sqrt(dot(a, a)); }
static inline f32 length(vec4 a) { return wd_sqrt(dot(a, a)); }
static inline vec3 normalize(vec3 a) { return a / length(a); }
static inline vec4 normalize(vec4 a) { return a / length(a); }
// WGSL spec 17.5: min(e1,e2) = e2 if e2 < e1 else e1;  max(e1,e2) = e2 if e1 < e2 else e1 (pins +-0 ties and NaN).
static inline f32 wmin(f32 a, f32 b) { return (b < a) ? b : a; }
static inline f32 wmax(f32 a, f32 b) { return (a < b) ? b : a; }
static inline f32 clamp(f32 v, f32 lo, f32 hi) { return wmin(wmax(v, lo), hi); }
static inline vec2 min(vec2 a, vec2 b) { return vec2{wmin(a.x, b.x), wmin(a.y, b.y)}; }
static inline vec2 max(vec2 a, vec2 b) { return vec2{wmax(a.x, b.x), wmax(a.y, b.y)}; }
static inline vec3 max(vec3 a, vec3 b) { return vec3{wmax(a.x, b.x), wmax(a.y, b.y), wmax(a.z, b.z)}; }
static inline vec2 clamp(vec2 v, vec2 lo, vec2 hi) { return min(max(v, lo), hi); }
static inline vec3 clamp(vec3 v, vec3 lo, vec3 hi) {
    return vec3{clamp(v.x, lo.x, hi.x), clamp(v.y, lo.y, hi.y), clamp(v.z, lo.z, hi.z)};
}
static inline vec3 exp(vec3 v) { return vec3{wd_exp(v.x), wd_exp(v.y), wd_exp(v.z)}; }
static inline f32 sign(f32 v) { return v > 0.0f ? 1.0f : (v < 0.0f ? -1.0f : 0.0f); }
static inline vec3 sign(vec3 v) { return vec3{sign(v.x), sign(v.y), sign(v.z)}; }
static inline vec3 abs(vec3 v) { return vec3{std::fabs(v.x), std::fabs(v.y), std::fabs(v.z)}; }

static inline u32 pack2x16float(vec2 v) { return (u32)f32_to_f16(v.x) | ((u32)f32_to_f16(v.y) << 16); }
static inline vec2 unpack2x16float(u32 w) { return vec2{f16_to_f32((uint16_t)(w & 0xFFFFu)), f16_to_f32((uint16_t)(w >> 16))}; }

// ---------------------------------------------------------------- matrices (column-major)
struct mat3 {
    vec3 c[3];
    vec3& operator[](int i) { return c[i]; }
    const vec3& operator[](int i) const { return c[i]; }
};
struct mat4 {
    vec4 c[4];
    vec4& operator[](int i) { return c[i]; }
    const vec4& operator[](int i) const { return c[i]; }
};
static inline mat3 M3(vec3 a, vec3 b, vec3 c) { return mat3{{a, b, c}}; }
static inline mat3 M3(f32 a, f32 b, f32 c, f32 d, f32 e, f32 f, f32 g, f32 h, f32 i) {
    return mat3{{vec3{a, b, c}, vec3{d, e, f}, vec3{g, h, i}}};
}
static inline vec3 operator*(const mat3& m, vec3 v) { return m[0] * v.x + m[1] * v.y + m[2] * v.z; }
static inline vec4 operator*(const mat4& m, vec4 v) { return m[0] * v.x + m[1] * v.y + m[2] * v.z + m[3] * v.w; }
static inline mat3 operator*(const mat3& a, const mat3& b) { return mat3{{a * b[0], a * b[1], a * b[2]}}; }
static inline mat4 operator*(const mat4& a, const mat4& b) { return mat4{{a * b[0], a * b[1], a * b[2], a * b[3]}}; }
static inline mat3 operator*(f32 s, const mat3& m) { return mat3{{s * m[0], s * m[1], s * m[2]}}; }
static inline mat3 transpose(const mat3& m) {
    return mat3{{vec3{m[0].x, m[1].x, m[2].x}, vec3{m[0].y, m[1].y, m[2].y}, vec3{m[0].z, m[1].z, m[2].z}}};
}
static inline mat4 transpose(const mat4& m) {
    return mat4{{vec4{m[0].x, m[1].x, m[2].x, m[3].x}, vec4{m[0].y, m[1].y, m[2].y, m[3].y},
                 vec4{m[0].z, m[1].z, m[2].z, m[3].z}, vec4{m[0].w, m[1].w, m[2].w, m[3].w}}};
}

}  // namespace wgsl
