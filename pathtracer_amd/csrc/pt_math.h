// Arithmetic shared by the host layer and the gfx950 kernels.
//
// The reference does all of its vector arithmetic through thBase (vec3, Ray, Triangle, Plane,
// mat3/mat4, FloatEpsilon, fastsqrt), which is not part of the reference checkout.  Every operation
// here therefore has one fixed evaluation order, written with explicit fmaf where a fused operation
// is meant and compiled with -ffp-contract=off, so that host code, device code and any independent
// CPU restatement produce the same bits for +,-,*,/ and sqrt:
//   dot(a,b)      = fma(az,bz, fma(ay,by, ax*bx))
//   cross(a,b).x  = fma(ay,bz, -(az*by))   (y, z cyclic)
//   normalized(v) = v * (1 / sqrt(dot(v,v)))
//   mat3(c0,c1,c2) * v = fma chain over columns
// Anything else follows the reference source left to right, one rounding per operator.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>

#define PT_HD __host__ __device__ __forceinline__

namespace pt {

struct V3 { float x, y, z; };
PT_HD V3 v3(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
PT_HD V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
PT_HD V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
PT_HD V3 operator-(V3 a) { return v3(-a.x, -a.y, -a.z); }
PT_HD V3 operator*(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
PT_HD V3 operator*(float s, V3 a) { return v3(s * a.x, s * a.y, s * a.z); }
PT_HD V3 operator*(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }   // component-wise (rendering.d:391,418,424)
PT_HD V3 operator/(V3 a, float s) { return v3(a.x / s, a.y / s, a.z / s); }
PT_HD float dot(V3 a, V3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
PT_HD V3 cross(V3 a, V3 b) {
  return v3(fmaf(a.y, b.z, -(a.z * b.y)), fmaf(a.z, b.x, -(a.x * b.z)), fmaf(a.x, b.y, -(a.y * b.x)));
}
PT_HD float length(V3 a) { return sqrtf(dot(a, a)); }
PT_HD V3 normalized(V3 a) { float inv = 1.0f / sqrtf(dot(a, a)); return a * inv; }
PT_HD V3 mat3cols(V3 c0, V3 c1, V3 c2, V3 v) {
  return v3(fmaf(c2.x, v.z, fmaf(c1.x, v.y, c0.x * v.x)),
            fmaf(c2.y, v.z, fmaf(c1.y, v.y, c0.y * v.x)),
            fmaf(c2.z, v.z, fmaf(c1.z, v.y, c0.z * v.x)));
}

#define PT_PI_F 3.14159265358979323846f
#define PT_PI_2_F 1.57079632679489661923f
#define PT_BRDF (1.0f / PT_PI_F)          /* rendering.d:368 */
#define PT_FLT_MAX 3.402823466e+38f

// Camera's state, camera.d:11-12
struct Camera { V3 pos, dir, right, up; float tanFov, aspect; };

// Camera.getScreenRay, camera.d:38-42
PT_HD V3 screenRayDir(const Camera& c, float x, float y) {
  V3 a = (c.right * x) * c.tanFov;
  V3 b = ((c.up * y) * c.tanFov) * c.aspect;
  return normalized((c.dir + a) + b);
}

// ------------------------------------------------------------------ RNG
// Counter-keyed streams: (seed, pixel, sample, stream) -> xoroshiro64** state.  The reference uses one
// Mt19937 per 4-scanline band (main.d:109,117), a serial dependence that cannot be kept on a GPU; keyed
// streams make the image independent of how pixels are partitioned over wavefronts or GPUs.
PT_HD uint32_t rotl32(uint32_t x, int k) { return (x << k) | (x >> (32 - k)); }
PT_HD uint32_t mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x21f0aaadu; x ^= x >> 15; x *= 0x735a2d97u; x ^= x >> 15; return x;
}
struct Rng {
  uint32_t s0, s1;
  PT_HD void seed(uint32_t seed, uint32_t pixel, uint32_t sample, uint32_t stream) {
    s0 = mix32(mix32(mix32(seed ^ 0x9E3779B9u) + pixel) ^ (sample * 0x85EBCA6Bu + stream));
    s1 = mix32(mix32(mix32(seed ^ 0x7F4A7C15u) + sample) ^ (pixel * 0xC2B2AE35u + stream * 0x27D4EB2Fu));
    if ((s0 | s1) == 0) s0 = 1;
  }
  PT_HD uint32_t next() {
    uint32_t r = rotl32(s0 * 0x9E3779BBu, 5) * 5u;
    s1 ^= s0; s0 = rotl32(s0, 26) ^ s1 ^ (s1 << 9); s1 = rotl32(s1, 13);
    return r;
  }
  PT_HD float u01() { return (float)(next() >> 8) * (1.0f / 16777216.0f); }
  PT_HD float uniform(float a, float b) { return a + (b - a) * u01(); }   // std.random.uniform, "[)"
};

// getViewRay's screen coordinates, rendering.d:280-283 (x first, then y); col = pixelIndex % W, row = pixelIndex / W
PT_HD void viewRayXY(uint32_t col, uint32_t row, uint32_t W, uint32_t H, Rng* gen, float& x, float& y) {
  float psx = 1.0f / (float)W, psy = 1.0f / (float)H;
  x = (float)col / (float)W * 2.0f - 1.0f;
  if (gen) x = x + gen->uniform(-psx, psx);
  y = (float)row / (float)H * 2.0f - 1.0f;
  if (gen) y = y + gen->uniform(-psy, psy);
}

// Shading normal of Scene.trace, scene.d:711-723 (fastsqrt -> sqrtf).  u weights vertex 2, v vertex 1.
PT_HD V3 interpNormal(V3 n0, V3 n1, V3 n2, float u, float v) {
  float x = 1.0f / (u + v);
  float u1 = x * u, v1 = x * v;
  const float sqrt2 = 1.414213562f;
  float d1 = sqrtf((1.0f - u1) * (1.0f - u1) + v1 * v1) / sqrt2;
  float d2 = sqrtf(u1 * u1 + (1.0f - v1) * (1.0f - v1)) / sqrt2;
  V3 interpolated1 = n1 * d1 + n2 * d2;
  float len = sqrtf(u1 * u1 + v1 * v1);
  float i1 = sqrtf(u * u + v * v) / len;
  float i2 = 1.0f - i1;
  return normalized(n0 * i2 + interpolated1 * i1);
}

// interpolate(u, v, val0, val1, val2) of the texcoord variant of Scene.trace, one component (scene.d:68-105):
// the weights of interpNormal with u, v clamped at 0 and the u + v == 0 corner defined
PT_HD float interpolateComponent(float u, float v, float val0, float val1, float val2) {
  if (u < 0.0f) u = 0.0f;
  if (v < 0.0f) v = 0.0f;
  float uv = (u + v);
  float u1, v1;
  if (uv != 0.0f) { float x = 1.0f / uv; u1 = x * u; v1 = x * v; }
  else { u1 = u; v1 = v; }
  const float sqrt2 = 1.414213562f;
  float d1 = sqrtf((1.0f - u1) * (1.0f - u1) + v1 * v1) / sqrt2;
  float d2 = sqrtf(u1 * u1 + (1.0f - v1) * (1.0f - v1)) / sqrt2;
  float interpolated1 = val1 * d1 + val2 * d2;
  float i1;
  if (uv != 0.0f) { float len = sqrtf(u1 * u1 + v1 * v1); i1 = sqrtf(u * u + v * v) / len; }
  else i1 = 0.0f;
  float i2 = 1.0f - i1;
  return val0 * i2 + interpolated1 * i1;
}

// ------------------------------------------------------------------ sin, cos, pow with one fixed evaluation
// cosf / sinf (rendering.d:309-330) and powf (main.d:92) are library calls in the reference; two libraries differ
// in the last bit, and one bit of a direction can flip a visibility decision.  These evaluate one explicit sequence
// of fma / * / + / rint (and, for pow, IEEE division), so that host and device (and any restatement that uses the
// same constants and order) return the same bits.
//
// sin and cos of x, binary32 throughout (the integrator evaluates them eleven times per shaded vertex, and binary64
// vector arithmetic runs at half rate with twice the registers): k = rint(x * 2/pi); r = x - k * pi/2 with pi/2 in
// four parts (Cody-Waite: the first two have 9 and 11 significant bits, so k * part is exact and so is the
// difference; the last two carry pi/2 to 2^-76, which keeps r accurate when x lies next to a multiple of pi/2);
// Taylor polynomials in r on [-pi/4, pi/4] (sin through r^9, cos through r^10: truncation < 0.03 ulp), Horner with
// fma from the highest coefficient; quadrant k & 3.  Checked against the correctly rounded value for EVERY binary32
// in [0, 100] (scripts/probes/sincos_f32_check.cpp): never more than 1 ulp away, and the correctly rounded value
// itself for 89 % (sin) and 83 % (cos) of arguments drawn uniformly from [0, 2 pi].
PT_HD void sincosFixed(float x, float* sinOut, float* cosOut) {
  const float k = rintf(x * 0.636619772f);                           // 2 / pi
  float r = fmaf(-k, 1.5703125f, x);                                 // pi / 2: 0x3FC90000
  r = fmaf(-k, 4.837512969970703125e-4f, r);                         //         0x39FDA000
  r = fmaf(-k, 7.549790126404332e-8f, r);                            //         0x33A22169
  r = fmaf(-k, -1.7151245100058819e-15f, r);                         //         0xA6F72CED
  const float z = r * r;
  float ps = 2.7557314297e-6f;                                       //  1/9!
  ps = fmaf(ps, z, -1.9841270114e-4f);                               // -1/7!
  ps = fmaf(ps, z, 8.3333337680e-3f);                                //  1/5!
  ps = fmaf(ps, z, -1.6666667163e-1f);                               // -1/3!
  const float s = fmaf(r * z, ps, r);
  float pc = -2.7557314297e-7f;                                      // -1/10!
  pc = fmaf(pc, z, 2.4801587642e-5f);                                //  1/8!
  pc = fmaf(pc, z, -1.3888889225e-3f);                               // -1/6!
  pc = fmaf(pc, z, 4.1666667908e-2f);                                //  1/4!
  pc = fmaf(pc, z, -0.5f);                                           // -1/2!
  const float c = fmaf(z, pc, 1.0f);
  const int q = (int)k & 3;
  const float sq = (q & 1) ? c : s, cq = (q & 1) ? s : c;
  *sinOut = (q & 2) ? -sq : sq;
  *cosOut = ((q + 1) & 2) ? -cq : cq;
}

// pow: binary64 with an explicit sequence of fma / * / + / rint / IEEE division, rounded to binary32 once (error
// before the final rounding < 1e-13 relative: the correctly rounded result except in about one case in 10^4, where
// it is its neighbour).  One evaluation per channel of a displayed pixel, not per sample.
// x^(1/2.4) for x in (0, 1] (the sRGB curve of drawScreen, main.d:92): x = m * 2^e with m in [sqrt(1/2), sqrt(2));
// ln m = 2 atanh(f), f = (m - 1) / (m + 1), series through f^13; y = (e ln 2 + ln m) * (1 / 2.4); n = rint(y / ln 2);
// g = y - n ln 2; exp(g) by Taylor through g^11; result scaled by 2^n.
PT_HD float powInv24Fixed(float x) {
  const double xd = (double)x;
  unsigned long long bits; __builtin_memcpy(&bits, &xd, 8);
  int e = (int)((bits >> 52) & 0x7FFu) - 1023;
  bits = (bits & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull;     // mantissa in [1, 2)
  double m; __builtin_memcpy(&m, &bits, 8);
  if (m > 1.41421356237309514547) { m = m * 0.5; e = e + 1; }
  const double f = (m - 1.0) / (m + 1.0), f2 = f * f;
  double pl = 1.0 / 13.0;
  pl = fma(pl, f2, 1.0 / 11.0);
  pl = fma(pl, f2, 1.0 / 9.0);
  pl = fma(pl, f2, 1.0 / 7.0);
  pl = fma(pl, f2, 1.0 / 5.0);
  pl = fma(pl, f2, 1.0 / 3.0);
  pl = fma(pl, f2, 1.0);
  const double lnm = 2.0 * f * pl;
  const double lnx = fma((double)e, 0.69314718055994528623, lnm);
  const double y = lnx * (1.0 / 2.4);
  const double n = rint(y * 1.44269504088896338700);                 // 1 / ln 2
  const double g = fma(-n, 0.69314718055994528623, y);               // |g| <= 0.35
  double pe = 1.0 / 39916800.0;                                      // 1/11!
  pe = fma(pe, g, 1.0 / 3628800.0);
  pe = fma(pe, g, 1.0 / 362880.0);
  pe = fma(pe, g, 1.0 / 40320.0);
  pe = fma(pe, g, 1.0 / 5040.0);
  pe = fma(pe, g, 1.0 / 720.0);
  pe = fma(pe, g, 1.0 / 120.0);
  pe = fma(pe, g, 1.0 / 24.0);
  pe = fma(pe, g, 1.0 / 6.0);
  pe = fma(pe, g, 0.5);
  pe = fma(pe, g, 1.0);
  pe = fma(pe, g, 1.0);
  unsigned long long sb = (unsigned long long)((long long)n + 1023) << 52;   // 2^n, n in [-1100/2.4, 0]
  double scale; __builtin_memcpy(&scale, &sb, 8);
  return (float)(pe * scale);
}

// drawScreen's scalar path for one channel (main.d:87-93): saturate, sRGB curve, cast(ubyte)(x * 255)
PT_HD uint8_t tonemapChannel(float x) {
  const float A = 0.055f, aPlusOne = 1 + 0.055f;
  if (!(x == x)) return 0;                                           // NaN (the cast is undefined for it)
  x = x < 0.0f ? 0.0f : (x > 1.0f ? 1.0f : x);
  x = (x <= 0.0031308f) ? x * 12.92f : aPlusOne * powInv24Fixed(x) - A;
  return (uint8_t)(x * 255.0f);
}

// rendering.d:309-314
PT_HD V3 angleToLocalDirection(float phi, float psi) {
  float sinPhi, cosPhi, sinPsi, cosPsi;
  sincosFixed(phi, &sinPhi, &cosPhi); sincosFixed(psi, &sinPsi, &cosPsi);
  return v3(cosPsi * cosPhi, sinPsi * cosPhi, sinPhi);
}
// rendering.d:316-330
PT_HD V3 angleToDirection(float phi, float psi, V3 normal) {
  V3 localDir = angleToLocalDirection(phi, psi);
  V3 up = normal, dir = v3(1, 0, 0);
  if (fabsf(dot(dir, up)) > 0.9f) dir = v3(0, 1, 0);
  V3 right = normalized(cross(up, dir));
  dir = normalized(cross(up, right));
  return mat3cols(dir, right, up, localDir);
}
// rendering.d:332-339 (its own sun direction (1,1,6), independent of g_sunDir)
PT_HD V3 sampleSky(V3 dir) {
  V3 sunDir = normalized(v3(1, 1, 6));
  if (dot(sunDir, dir) > 0.997f) return v3(1.0f, 0.984f, 0.8f) * 30.0f;
  return v3(0.682f, 0.977f, 1.0f);
}

}  // namespace pt
