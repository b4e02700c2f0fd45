/*
 * detmath.h — deterministic fp32 elementary functions shared by the HIP kernels (device) and by
 * the CPU oracle in its bit-exact ("det") math mode.
 *
 * Why: the reference evaluates exp/atan/cos/sin/pow/log2 with whatever the GLSL implementation
 * provides (ComputeOrientation.comp:75-81,105-106; ComputeDescriptors.comp:111-124,146,162;
 * ExtractKeypoints.comp:215-219) — precision is implementation-defined and differs per GPU vendor
 * (SURVEY.md quirk Q14). This build pins them to the functions below, written with +,-,*,fmaf,
 * IEEE division and integer bit operations only, so that gfx950 and x86-64 produce the same bits.
 * Accuracy (tools/fit_detmath.py, tests/test_detmath.py): <= 2 ulp for exp/exp2/atan/sin/cos on
 * the ranges the pipeline uses.
 *
 * Build rules for bit-exactness: compile every translation unit that includes this header with
 * -ffp-contract=off (fused multiply-adds are spelled out as fmaf), no -ffast-math, and on the
 * device keep hipcc's default correctly-rounded fp32 '/' and sqrtf.
 */
#ifndef VKSIFT_DETMATH_H
#define VKSIFT_DETMATH_H

#include <math.h>
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define DM_FN static __device__ __host__ __forceinline__
#else
#define DM_FN static inline
#endif

#define DM_PI_F 3.14159265358979323846f     /* == (float)pi, the value GLSL's float literal takes */
#define DM_TWO_PI_F (2.f * DM_PI_F)

DM_FN uint32_t dm_f2u(float f)
{
  uint32_t u;
  memcpy(&u, &f, 4);
  return u;
}
DM_FN float dm_u2f(uint32_t u)
{
  float f;
  memcpy(&f, &u, 4);
  return f;
}

/* 2^n for integer n in [-126, 127] (exact). */
DM_FN float dm_pow2i(int n) { return dm_u2f((uint32_t)(n + 127) << 23); }

/* Round to nearest integer, ties to even, for |x| < 2^22 (exact, no libm call). */
DM_FN float dm_rint(float x)
{
  const float magic = 12582912.f; /* 1.5 * 2^23 */
  float t = x + magic;
  return t - magic;
}

/* e^x. Relative error <= ~1.5 ulp for x in [-87, 87]; returns 0 below -87.3 (the callers only
 * ever scale the result to a <= 2^30 fixed-point integer, so the flushed tail is always 0). */
/* dm_expf for -87.3 <= x <= 88.7 (no range handling, straight-line code): same operations, same bits */
DM_FN float dm_expf_core(float x)
{
  float n = dm_rint(x * 0x1.715476p+0f); /* x / ln2 */
  float r = fmaf(n, -0x1.62e400p-1f, x);  /* ln2 high part: 12 trailing zero bits -> n*hi exact */
  r = fmaf(n, -0x1.7f7d1cp-20f, r);       /* ln2 low part */
  float p = 0x1.6d4922p-10f;
  p = fmaf(p, r, 0x1.121072p-7f);
  p = fmaf(p, r, 0x1.5554e4p-5f);
  p = fmaf(p, r, 0x1.5554d8p-3f);
  p = fmaf(p, r, 0.5f);
  float e = fmaf(p * r, r, r) + 1.f;
  int ni = (int)n;
  /* split the scaling so that ni in [-126-..,128] never builds an out-of-range exponent field */
  int h = ni / 2;
  return (e * dm_pow2i(h)) * dm_pow2i(ni - h);
}

/* dm_expf_core for -87.3 <= x <= 0 (every exponent the kernels form): n = rint(x / ln2) is in [-126, 0], so 2^n is a
 * normal number and ONE exact power-of-two scaling replaces the split one (e * 2^h is exact, so both forms round the same
 * real number once): same bits, 6 instructions less. Checked against dm_expf for every float in the range. */
DM_FN float dm_expf_core_nonpos(float x)
{
  float n = dm_rint(x * 0x1.715476p+0f);
  float r = fmaf(n, -0x1.62e400p-1f, x);
  r = fmaf(n, -0x1.7f7d1cp-20f, r);
  float p = 0x1.6d4922p-10f;
  p = fmaf(p, r, 0x1.121072p-7f);
  p = fmaf(p, r, 0x1.5554e4p-5f);
  p = fmaf(p, r, 0x1.5554d8p-3f);
  p = fmaf(p, r, 0.5f);
  float e = fmaf(p * r, r, r) + 1.f;
  return e * dm_pow2i((int)n);
}

/* dm_expf for x <= 0 without branches (kernels): clamp, evaluate, select — the same bits */
DM_FN float dm_expf_nb_nonpos(float x)
{
  const float e = dm_expf_core_nonpos(x < -87.3f ? -87.3f : x);
  return x < -87.3f ? 0.f : e;
}

/* dm_expf without branches (kernels): clamp, evaluate, select — the same bits for every x */
DM_FN float dm_expf_nb(float x)
{
  const float e = dm_expf_core(x < -87.3f ? -87.3f : (x > 88.7f ? 88.7f : x));
  return x < -87.3f ? 0.f : e;
}

DM_FN float dm_expf(float x)
{
  if (x < -87.3f)
    return 0.f;
  if (x > 88.7f)
    x = 88.7f;
  return dm_expf_core(x);
}

/* 2^x for |x| < 120. Relative error <= ~1 ulp. */
DM_FN float dm_exp2f(float x)
{
  float n = dm_rint(x);
  float r = x - n; /* exact */
  float q = 0x1.00c54ep-16f;
  q = fmaf(q, r, 0x1.444646p-13f);
  q = fmaf(q, r, 0x1.5d8770p-10f);
  q = fmaf(q, r, 0x1.3b2a16p-7f);
  q = fmaf(q, r, 0x1.c6b08ep-5f);
  q = fmaf(q, r, 0x1.ebfbe0p-3f);
  q = fmaf(q, r, 0x1.62e430p-1f);
  float e = fmaf(q, r, 1.f);
  return e * dm_pow2i((int)n);
}

/* x / (2*pi), bit-identical to the IEEE division, in 3 operations instead of the 11 of the device's division expansion:
 * q0 = x * RN(1/c), one exact remainder, one correction (Markstein). Verified exhaustively against x / c for every float in
 * [2^-103, 256] (931 135 489 values; tests/test_detmath.py re-checks a sample and the guard); below 2^-96 the remainder
 * would underflow, so those inputs (never seen in practice) take the division, and zeros keep their sign through x * rc.
 * Used by the kernels only: the oracle keeps the plain division. */
DM_FN float dm_div_2pi(float x)
{
  const float c = DM_TWO_PI_F, rc = 0x1.45f306p-3f; /* RN(1 / c) */
  float q = x * rc;
  if (fabsf(x) >= 0x1p-96f)
    q = fmaf(fmaf(-q, c, x), rc, q);
  else if (x != 0.f)
    q = x / c;
  return q;
}

/* x / 3 for the integer-valued x in [0, 2^32] that the orientation histogram's smoothing divides (ComputeOrientation.comp:130-147: sums of
 * three uint32 bins, converted to float), bit-identical to the IEEE division in 3 operations: q0 = x * RN(1/3), one exact remainder, one
 * correction. Verified against x / 3.f for EVERY such float (83 886 081 values: tests/test_detmath.py through orc_check_div_3).
 * Used by the kernels only: the oracle keeps the plain division. */
DM_FN float dm_div_3(float x)
{
  const float rc = 0x1.555556p-2f; /* RN(1 / 3) */
  const float q = x * rc;
  return fmaf(fmaf(-q, 3.f, x), rc, q);
}

/* dm_atan2f behind its division: a = min(|x|, |y|) / max(|x|, |y|) (a = 0 for x = y = 0), ax = |x|, ay = |y| */
DM_FN float dm_atan2f_ratio(float a, float ax, float ay, float x, float y)
{
  float z = a * a;
  float p = -0x1.f76bccp-11f;
  p = fmaf(p, z, 0x1.9eb02ep-8f);
  p = fmaf(p, z, -0x1.3fccb4p-6f);
  p = fmaf(p, z, 0x1.3c7ccep-5f);
  p = fmaf(p, z, -0x1.d9b870p-5f);
  p = fmaf(p, z, 0x1.3038d4p-4f);
  p = fmaf(p, z, -0x1.724100p-4f);
  p = fmaf(p, z, 0x1.c6dda6p-4f);
  p = fmaf(p, z, -0x1.249062p-3f);
  p = fmaf(p, z, 0x1.99998ep-3f);
  p = fmaf(p, z, -0x1.555556p-2f);
  float t = fmaf(p * z, a, a); /* atan(a), a in [0,1] */
  if (ay > ax)
    t = 0x1.921fb6p+0f - t; /* pi/2 - t */
  if (x < 0.f)
    t = 0x1.921fb6p+1f - t; /* pi - t */
  return y < 0.f ? -t : t;
}

/* atan2(y, x) in (-pi, pi]; atan2(0,0) = 0 (GLSL leaves it undefined, quirk Q13). <= ~2 ulp. */
DM_FN float dm_atan2f(float y, float x)
{
  float ax = fabsf(x), ay = fabsf(y);
  float mx = ax > ay ? ax : ay;
  float mn = ax > ay ? ay : ax;
  /* atan2(+-0, +-0) = +0: with the divisor replaced by 1 the straight-line path below yields exactly that
   * (a = 0, t = +0, neither x < 0 nor y < 0 holds for a signed zero) — no early return, no branch in the kernels */
  float a = mn / (mx == 0.f ? 1.f : mx); /* in [0,1] */
  return dm_atan2f_ratio(a, ax, ay, x, y);
}

/* sin and cos of t for |t| <= ~16 (the pipeline passes orientations in [0, 2*pi]). <= ~1.5 ulp. */
DM_FN void dm_sincosf(float t, float *s_out, float *c_out)
{
  float k = dm_rint(t * 0x1.45f306p-1f);  /* t * 2/pi */
  float r = fmaf(k, -0x1.921e00p+0f, t);  /* pi/2 high, 16 significant bits: k*hi exact for small k */
  r = fmaf(k, -0x1.b54400p-16f, r);       /* pi/2 mid, 16 bits */
  r = fmaf(k, -0x1.0b4600p-34f, r);       /* pi/2 low */
  float z = r * r;
  float sp = 0x1.6da7b6p-19f;
  sp = fmaf(sp, z, -0x1.a01360p-13f);
  sp = fmaf(sp, z, 0x1.11110ep-7f);
  sp = fmaf(sp, z, -0x1.555556p-3f);
  float s = fmaf(sp * z, r, r);
  float cp = -0x1.23e1f2p-22f;
  cp = fmaf(cp, z, 0x1.a00f70p-16f);
  cp = fmaf(cp, z, -0x1.6c16b6p-10f);
  cp = fmaf(cp, z, 0x1.555556p-5f);
  cp = fmaf(cp, z, -0.5f);
  float c = fmaf(cp, z, 1.f);
  int q = ((int)k) & 3;
  float ss = (q & 1) ? c : s;
  float cc = (q & 1) ? s : c;
  if (q == 1 || q == 2)
    cc = -cc;
  if (q >= 2)
    ss = -ss;
  *s_out = ss;
  *c_out = cc;
}

/* ceil(log2(m)) for finite m > 0, exact (the reference takes ceil of an approximate GPU log2,
 * ComputeOrientation.comp:81, ComputeDescriptors.comp:124). */
DM_FN int dm_ceil_log2f(float m)
{
  uint32_t u = dm_f2u(m);
  int e = (int)((u >> 23) & 0xff) - 127;
  uint32_t man = u & 0x7fffffu;
  if (((u >> 23) & 0xff) == 0)
  { /* subnormal: normalise */
    float mm = m * 8388608.f;
    u = dm_f2u(mm);
    e = (int)((u >> 23) & 0xff) - 127 - 23;
    man = u & 0x7fffffu;
  }
  return man ? e + 1 : e;
}

#endif /* VKSIFT_DETMATH_H */
