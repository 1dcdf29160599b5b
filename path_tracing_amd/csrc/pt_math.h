// pt_math.h — device arithmetic of the path tracer, with a fixed evaluation order.
//
// WGSL leaves fused multiply-add, the evaluation order of dot/cross/matrix
// products and the accuracy of sin/cos/pow/normalize to the implementation. This
// backend fixes one evaluation so that every accept/reject decision and every
// argmin is reproducible across machines (and against a CPU checker built with
// the same rules): IEEE-754 binary32, round to nearest even, subnormals kept,
// correctly rounded divide and sqrt; the translation unit is compiled with
// -ffp-contract=off, so the only fused operations are the explicit
// __builtin_fmaf calls in this file:
//   dot(a,b)   = fma(a.z,b.z, fma(a.y,b.y, a.x*b.x))
//   cross(a,b) = ( fma(a.y,b.z,-(a.z*b.y)), fma(a.z,b.x,-(a.x*b.z)), fma(a.x,b.y,-(a.y*b.x)) )
//   M*(v,0)    : r_i = fma(M[2][i],v.z, fma(M[1][i],v.y, M[0][i]*v.x)); M*(v,1) adds M[3][i] last
//   a + b*s    = fma(b,s,a);  mix(a,b,t) = fma(b,t, a*(1-t));  reflect(I,N) = fma(-(2*dot(N,I)), N, I)
//   normalize(v) = v * inverseSqrt(dot(v,v)), inverseSqrt by the fixed recipe of pt_rsqrt (integer seed, three Newton
//   steps; within the 2 ulp WGSL allows inverseSqrt);  pow(x,5) = ((x*x)*(x*x))*x
//   sin/cos: Cody-Waite reduction by pi/2 and fixed polynomials (pt_sincos).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#define PT_DEV __device__ __forceinline__

// reference src/math.wgsl:3-6
#define PT_EPSILON 4.88e-4f
#define PT_WGSL_INFINITY 10000000.0f
#define PT_T_MIN 0.0001f
#define PT_T_MAX 1000.0f           // reference src/raytrace.wgsl:230
#define PT_PI 3.14159265358979323846f  // bevy_render::maths::PI
#define PT_NONE 0xFFFFFFFFu

struct v3 {
    float x, y, z;
};

PT_DEV v3 V3(float x, float y, float z) { v3 r; r.x = x; r.y = y; r.z = z; return r; }
PT_DEV float pt_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
PT_DEV float dot3(v3 a, v3 b) { return pt_fma(a.z, b.z, pt_fma(a.y, b.y, a.x * b.x)); }
PT_DEV v3 cross3(v3 a, v3 b) {
    return V3(pt_fma(a.y, b.z, -(a.z * b.y)), pt_fma(a.z, b.x, -(a.x * b.z)), pt_fma(a.x, b.y, -(a.y * b.x)));
}
PT_DEV v3 sub3(v3 a, v3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
PT_DEV v3 add3(v3 a, v3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
PT_DEV v3 mul3(v3 a, v3 b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
PT_DEV v3 scale3(v3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }
PT_DEV v3 neg3(v3 a) { return V3(-a.x, -a.y, -a.z); }
PT_DEV v3 madd3(v3 a, v3 b, float s) { return V3(pt_fma(b.x, s, a.x), pt_fma(b.y, s, a.y), pt_fma(b.z, s, a.z)); }
PT_DEV float pt_div(float a, float b) { return a / b; }       // correctly rounded (hipcc default)
PT_DEV float pt_sqrt(float a) { return __builtin_sqrtf(a); }  // correctly rounded (hipcc default)
// Correctly rounded square root for arguments that are 0 or in [2^-96, 2^64): the hardware's 1-ulp v_sqrt_f32 and the
// same final step the compiler's expansion of sqrtf takes (the two neighbouring values are tried with an exact fused
// residual), without its rescaling of tiny arguments and its special-case pass for zero, infinity and NaN, which
// those arguments never need (a zero argument comes through as zero: both neighbour tests fail). The arguments of
// cosine_sample, r and 1 - r with r = fract(u32 / 3141.59...), are 0 or at least 2^-24. Seven instructions fewer.
PT_DEV float pt_sqrt_in_range(float x) {
    const float s = __builtin_amdgcn_sqrtf(x);
    const float s_down = __uint_as_float(__float_as_uint(s) - 1u), s_up = __uint_as_float(__float_as_uint(s) + 1u);
    const float r_down = pt_fma(-s_down, s, x), r_up = pt_fma(-s_up, s, x);
    float r = (r_down <= 0.0f) ? s_down : s;
    r = (r_up > 0.0f) ? s_up : r;
    return r;
}
// 1/sqrt(d) by a fixed recipe of f32 operations (identical in the CPU restatement): integer seed, three Newton steps.
// At most 1.92 ulp from the exact value for d in [2^-100, 2^100] (checked over every f32 in that range), inside the 2 ulp
// WGSL allows inverseSqrt. No division, no square root instruction (both are quarter rate and need long fix-up
// sequences to be correctly rounded): a third of the cycles of v * (1 / sqrt(d)).
PT_DEV float pt_rsqrt(float d) {
    float y = __uint_as_float(0x5f375a86u - (__float_as_uint(d) >> 1));
    const float hd = 0.5f * d;
    y = y * pt_fma(-(hd * y), y, 1.5f);
    y = y * pt_fma(-(hd * y), y, 1.5f);
    y = y * pt_fma(-(hd * y), y, 1.5f);
    return y;
}
PT_DEV v3 normalize3(v3 a) {
    const float r = pt_rsqrt(dot3(a, a));
    return V3(a.x * r, a.y * r, a.z * r);
}
PT_DEV float pt_min(float a, float b) { return __builtin_fminf(a, b); }
PT_DEV float pt_max(float a, float b) { return __builtin_fmaxf(a, b); }
PT_DEV float pt_saturate(float x) { return pt_min(pt_max(x, 0.0f), 1.0f); }
PT_DEV float pt_pow5(float x) { float x2 = x * x; return (x2 * x2) * x; }

// WGSL u32(f32): truncate toward zero, saturate, NaN -> 0.
PT_DEV uint32_t pt_f2u(float f) {
    if (!(f > 0.0f)) return 0u;
    if (f >= 4294967296.0f) return 0xFFFFFFFFu;
    return (uint32_t)f;
}

// A 3x4 affine map stored as rows: row_i = (m[i], m[4+i], m[8+i], m[12+i]) of a column-major mat4.
struct Affine {
    float4 r0, r1, r2;
};
PT_DEV v3 xform_dir(const Affine& m, v3 v) {
    return V3(pt_fma(m.r0.z, v.z, pt_fma(m.r0.y, v.y, m.r0.x * v.x)),
              pt_fma(m.r1.z, v.z, pt_fma(m.r1.y, v.y, m.r1.x * v.x)),
              pt_fma(m.r2.z, v.z, pt_fma(m.r2.y, v.y, m.r2.x * v.x)));
}
PT_DEV v3 xform_point(const Affine& m, v3 v) {
    v3 r = xform_dir(m, v);
    return V3(r.x + m.r0.w, r.y + m.r1.w, r.z + m.r2.w);
}

// sin and cos of one angle: k = rint(x*2/pi); y = x - k*pi/2 (two-constant Cody-Waite, fused);
// minimax polynomials on [-pi/4, pi/4]; quadrant fix-up.
PT_DEV void pt_sincos(float x, float& s_out, float& c_out) {
    float kf = __builtin_rintf(x * 0.636619772367581343f);
    int k = (__builtin_fabsf(kf) < 1073741824.0f) ? (int)kf : 0;
    float y = pt_fma(-kf, 1.57079637050628662109375f, x);
    y = pt_fma(-kf, -4.37113900018624283e-8f, y);
    float z = y * y;
    float ps = pt_fma(pt_fma(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f);
    float s = pt_fma(ps * z, y, y);
    float pc = pt_fma(pt_fma(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f);
    float c = pt_fma(pc * z, z, pt_fma(-0.5f, z, 1.0f));
    int q = k & 3;
    float ss = (q & 1) ? c : s;
    float cc = (q & 1) ? s : c;
    s_out = (q & 2) ? -ss : ss;              // q: 0 s, 1 c, 2 -s, 3 -c
    c_out = (q == 1 || q == 2) ? -cc : cc;   // q: 0 c, 1 -s, 2 -c, 3 s
}

// ---- RNG: reference src/raytrace.wgsl:60-82 -----------------------------------
struct Rng {
    uint32_t x, y, z;
};
// src/raytrace.wgsl:60-78 (PCG3D, JCGT 9(3) 2020)
PT_DEV void pcg3d(Rng& st) {
    uint32_t x = st.x * 1664525u + 1013904223u;
    uint32_t y = st.y * 1664525u + 1013904223u;
    uint32_t z = st.z * 1664525u + 1013904223u;
    x += y * z; y += z * x; z += x * y;
    x ^= x >> 16; y ^= y >> 16; z ^= z >> 16;
    x += y * z; y += z * x; z += x * y;
    st.x = x; st.y = y; st.z = z;
}
// one component of src/raytrace.wgsl:80-82: abs(fract(f32(v) / 3141.592653589793))
// The division is done as x*r, one exact fma residual and one fma correction (r = the f32 nearest 1/c). For every f32
// an unsigned 32-bit integer can convert to (0 and everything in [1, 2^32]) this is bit for bit the correctly rounded
// x / c; tools/verify_div_const.c checks all of them (tests/test_oracle.py runs it).
PT_DEV float rand_component(uint32_t v) {
    const float c = 3141.592653589793f, r = 1.0f / 3141.592653589793f;
    const float x = (float)v;
    const float q0 = x * r;
    const float q = __builtin_fmaf(__builtin_fmaf(-c, q0, x), r, q0);
    return __builtin_fabsf(q - __builtin_floorf(q));
}
