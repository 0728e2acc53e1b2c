// Deterministic device math ("dmath"): every step is an IEEE-754 binary32 operation with a fixed
// order, so results are bit-identical to any host that performs the same steps (the parity oracle does, in its
// own independent implementation).  Compiled with -ffp-contract=off; FMA only where written.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cstdint>

#define WD_DEV __device__ __forceinline__

WD_DEV float wd_bits2f(uint32_t u) { return __uint_as_float(u); }
WD_DEV uint32_t wd_f2bits(float f) { return __float_as_uint(f); }

// exp(x) = 2^n * P(r); n = rint(x*log2e); r = x - n*ln2 (hi/lo FMA); P degree 6 (FMA Horner).
// x < -86 -> 0 and x > 88 -> inf, so n is in [-124, 127] and the multiply by 2^n is exact; NaN propagates through r.
WD_DEV float wd_exp(float x) {
    const float LOG2E = wd_bits2f(0x3FB8AA3Bu);
    const float LN2_HI = wd_bits2f(0x3F318000u);
    const float LN2_LO = wd_bits2f(0xB95E8083u);
    const float C2 = wd_bits2f(1056964604u), C3 = wd_bits2f(1042983495u), C4 = wd_bits2f(1026207148u),
                C5 = wd_bits2f(1007230415u), C6 = wd_bits2f(984890875u);
    float n = __builtin_rintf(x * LOG2E);
    float r = __builtin_fmaf(-n, LN2_HI, x);
    r = __builtin_fmaf(-n, LN2_LO, r);
    float p = __builtin_fmaf(C6, r, C5);
    p = __builtin_fmaf(p, r, C4);
    p = __builtin_fmaf(p, r, C3);
    p = __builtin_fmaf(p, r, C2);
    p = __builtin_fmaf(p, r, 1.0f);
    p = __builtin_fmaf(p, r, 1.0f);
    // clamp the exponent so out-of-range x cannot build an invalid bit pattern before the selects below.  (The selects must act
    // on the product: for |x| far outside the range r is no longer small and p may be inf or NaN, so folding the range cases into
    // the power-of-two factor -- p*0, p*inf -- is NOT equivalent; tests/test_gpu_math.py holds the cases.)
    float nc = fminf(fmaxf(n, -126.0f), 127.0f);
    float res = p * wd_bits2f((uint32_t)((int)nc + 127) << 23);
    res = (x < -86.0f) ? 0.0f : res;
    res = (x > 88.0f) ? __builtin_inff() : res;
    return res;
}

// wd_exp for an argument the caller has brought into [-86, 87]: the same operations in the same order, minus the exponent clamp and
// the two range selects, which cannot fire there (n is in [-124, 126]).  Bit-identical to wd_exp on that interval
// (tests/test_gpu_math.py walks it); 13 instead of 19 instructions in the rasterization kernels' inner loops.
WD_DEV float wd_exp_inrange(float x) {
    const float LOG2E = wd_bits2f(0x3FB8AA3Bu);
    const float LN2_HI = wd_bits2f(0x3F318000u);
    const float LN2_LO = wd_bits2f(0xB95E8083u);
    const float C2 = wd_bits2f(1056964604u), C3 = wd_bits2f(1042983495u), C4 = wd_bits2f(1026207148u),
                C5 = wd_bits2f(1007230415u), C6 = wd_bits2f(984890875u);
    // n = rint(x * log2e) by adding and subtracting 1.5 * 2^23: in round-to-nearest-even the sum keeps exactly the integer rint would
    // return (|x * log2e| < 2^22), and its low mantissa bits are n in two's complement -- so 2^n is applied by adding those bits,
    // shifted to the exponent field, to p's (one v_lshl_add_u32; everything of the constant is shifted out).  p is in [0.7, 1.42] and n in
    // [-124, 126]: the result is a normal number and the addition is the exact multiplication.  Three ordinary operations where
    // v_rndne_f32, v_cvt_i32_f32 and v_ldexp_f32 were (same bits; tests/test_gpu_math.py walks the interval).
    const float MAGIC = 12582912.0f;
    const float t = x * LOG2E + MAGIC;
    const float n = t - MAGIC;
    float r = __builtin_fmaf(-n, LN2_HI, x);
    r = __builtin_fmaf(-n, LN2_LO, r);
    float p = __builtin_fmaf(C6, r, C5);
    p = __builtin_fmaf(p, r, C4);
    p = __builtin_fmaf(p, r, C3);
    p = __builtin_fmaf(p, r, C2);
    p = __builtin_fmaf(p, r, 1.0f);
    p = __builtin_fmaf(p, r, 1.0f);
    return wd_bits2f(wd_f2bits(p) + (wd_f2bits(t) << 23));
}

// Two-component float value with component-wise IEEE operations, written out as scalar instructions.  (On gfx950 a wave64 VALU
// instruction issues at 32 lanes/cycle; v_pk_*_f32 costs two issue slots plus hazard nops, so packing two fp32 operations into
// one instruction gains nothing -- measured: the same kernels run 2-16 % faster with scalar code and -fno-slp-vectorize.)
struct wd_pair {
    float x, y;
};
WD_DEV wd_pair operator+(wd_pair a, wd_pair b) { return {a.x + b.x, a.y + b.y}; }
WD_DEV wd_pair operator-(wd_pair a, wd_pair b) { return {a.x - b.x, a.y - b.y}; }
WD_DEV wd_pair operator*(wd_pair a, wd_pair b) { return {a.x * b.x, a.y * b.y}; }
WD_DEV wd_pair operator*(float s, wd_pair b) { return {s * b.x, s * b.y}; }
WD_DEV wd_pair operator*(wd_pair a, float s) { return {a.x * s, a.y * s}; }
WD_DEV wd_pair operator-(wd_pair a) { return {-a.x, -a.y}; }
WD_DEV wd_pair& operator+=(wd_pair& a, wd_pair b) { a.x += b.x; a.y += b.y; return a; }

WD_DEV float wd_log(float x) {
    if (x != x) return x;
    if (x < 0.0f) return __builtin_nanf("");
    if (x == 0.0f) return -__builtin_inff();
    if (x == __builtin_inff()) return x;
    int eadj = 0;
    if (x < wd_bits2f(0x00800000u)) { x = x * 8388608.0f; eadj = -23; }
    uint32_t b = wd_f2bits(x);
    int e = (int)((b >> 23) & 0xFFu) - 126 + eadj;
    float m = wd_bits2f((b & 0x007FFFFFu) | 0x3F000000u);
    if (m < 0.707106781186547524f) { e = e - 1; m = m + m; }
    m = m - 1.0f;
    float z = m * m;
    float y = 7.0376836292E-2f;
    y = __builtin_fmaf(y, m, -1.1514610310E-1f);
    y = __builtin_fmaf(y, m, 1.1676998740E-1f);
    y = __builtin_fmaf(y, m, -1.2420140846E-1f);
    y = __builtin_fmaf(y, m, 1.4249322787E-1f);
    y = __builtin_fmaf(y, m, -1.6668057665E-1f);
    y = __builtin_fmaf(y, m, 2.0000714765E-1f);
    y = __builtin_fmaf(y, m, -2.4999993993E-1f);
    y = __builtin_fmaf(y, m, 3.3333331174E-1f);
    y = (y * m) * z;
    float fe = (float)e;
    y = __builtin_fmaf(fe, -2.12194440e-4f, y);
    y = __builtin_fmaf(-0.5f, z, y);
    float r = m + y;
    r = __builtin_fmaf(fe, 0.693359375f, r);
    return r;
}

// __fsqrt_rn is the approximate v_sqrt_f32 unless OCML_BASIC_ROUNDED_OPERATIONS is defined; sqrtf and "/" are correctly
// rounded under -fhip-fp32-correctly-rounded-divide-sqrt (set explicitly in the Makefile).
WD_DEV float wd_sqrt(float x) { return __builtin_sqrtf(x); }
WD_DEV float wd_div(float a, float b) { return a / b; }
// a / b, correctly rounded, for operands the caller knows to be ordinary: the Newton-Raphson core of the compiler's own IEEE sequence
// without its operand pre-scaling (v_div_scale_f32) and special-case fix-up (v_div_fixup_f32), which only act when an operand, the
// reciprocal or the quotient leaves the normal range.  Callers: T / (1 - alpha) with T in [1e-4, 1], 1 - alpha in [0.01, 1]; the window
// means, variances and the SSIM quotient of loss.hip (numerators 0 or of either sign up to 1e2, denominators 25 or at least 1e-8).
// Bit-identical to wd_div there (tests/test_gpu_math.py), a numerator of -0 excepted (it yields +0; none of the callers can produce
// one); 8 instead of 11 instructions.
WD_DEV float wd_div_inrange(float a, float b) {
    float r = __builtin_amdgcn_rcpf(b);
    r = __builtin_fmaf(__builtin_fmaf(-b, r, 1.0f), r, r);
    float q = a * r;
    q = __builtin_fmaf(__builtin_fmaf(-b, q, a), r, q);
    const float e = __builtin_fmaf(-b, q, a);
    // the last fused multiply-add as a three-address instruction: left to itself the compiler emits a destructive v_fmac_f32 on q and,
    // when the quotient replaces the dividend in a loop-carried register (T = T / x), a v_mov_b32 to get it there
    float out;
    asm("v_fma_f32 %0, %1, %2, %3" : "=v"(out) : "v"(e), "v"(r), "v"(q));
    return out;
}
// WGSL min/max: min(e1,e2) = e2 if e2 < e1 else e1; max(e1,e2) = e2 if e1 < e2 else e1 (pins +-0 ties and NaN).
WD_DEV float wd_min(float a, float b) { return (b < a) ? b : a; }
WD_DEV float wd_max(float a, float b) { return (a < b) ? b : a; }
WD_DEV float wd_clamp(float v, float lo, float hi) { return wd_min(wd_max(v, lo), hi); }

// WGSL u32(f32)/i32(f32): truncate, saturate, NaN -> 0.
WD_DEV uint32_t wd_to_u32(float v) {
    if (!(v > 0.0f)) return 0u;
    if (v >= 4294967296.0f) return 0xFFFFFFFFu;
    return (uint32_t)v;
}
WD_DEV int32_t wd_to_i32(float v) {
    if (v != v) return 0;
    if (v >= 2147483648.0f) return 2147483647;
    if (v <= -2147483648.0f) return (int32_t)0x80000000u;
    return (int32_t)v;
}

// fp16 pack/unpack (round to nearest even, subnormals kept, overflow to inf).
// f32 -> f16, round to nearest even, of the ROUNDED f32 value.  The empty asm makes that value opaque: otherwise the compiler
// may fold a preceding f32 multiply into v_fma_mixlo_f16, which rounds the exact product once, straight to f16 -- not what the
// reference's f32 arithmetic followed by pack2x16float does (seen in K17: a product that is an exact f16 tie after its f32
// rounding packed to the other neighbour).
WD_DEV uint32_t wd_f16bits(float f) {
    asm("" : "+v"(f));
    return (uint32_t)__half_as_ushort(__float2half_rn(f));
}
WD_DEV uint32_t wd_pack2(float x, float y) { return wd_f16bits(x) | (wd_f16bits(y) << 16); }
WD_DEV float wd_unpack_lo(uint32_t w) { return __half2float(__ushort_as_half((unsigned short)(w & 0xFFFFu))); }
WD_DEV float wd_unpack_hi(uint32_t w) { return __half2float(__ushort_as_half((unsigned short)(w >> 16))); }
