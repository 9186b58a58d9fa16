// Exhaustive check of the fp32 sin/cos sequence of pt_math.h (sincosFixed) against the correctly rounded value:
// every binary32 in [lo, hi].  Prints the largest distance in units of the last place and how often the result is not
// the correctly rounded one.   g++ -O2 -ffp-contract=off -fopenmp sincos_f32_check.cpp -o /tmp/sincos_check
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <cstdlib>

static inline void sincosFixed(float x, float* sinOut, float* cosOut) {
  const float k = rintf(x * 0.636619772f);
  float r = fmaf(-k, 1.5703125f, x);
  r = fmaf(-k, 4.837512969970703125e-4f, r);
  r = fmaf(-k, 7.549790126404332e-8f, r);
  r = fmaf(-k, -1.7151245100058819e-15f, r);
  const float z = r * r;
  float ps = 2.7557314297e-6f;
  ps = fmaf(ps, z, -1.9841270114e-4f);
  ps = fmaf(ps, z, 8.3333337680e-3f);
  ps = fmaf(ps, z, -1.6666667163e-1f);
  const float s = fmaf(r * z, ps, r);
  float pc = -2.7557314297e-7f;
  pc = fmaf(pc, z, 2.4801587642e-5f);
  pc = fmaf(pc, z, -1.3888889225e-3f);
  pc = fmaf(pc, z, 4.1666667908e-2f);
  pc = fmaf(pc, z, -0.5f);
  const float c = fmaf(z, pc, 1.0f);
  const int q = (int)k & 3;
  const float sq = (q & 1) ? c : s, cq = (q & 1) ? s : c;
  *sinOut = (q & 2) ? -sq : sq;
  *cosOut = ((q + 1) & 2) ? -cq : cq;
}
static inline int64_t ulps(float a, float b) {
  int32_t ia, ib; memcpy(&ia, &a, 4); memcpy(&ib, &b, 4);
  if (ia < 0) ia = (int32_t)0x80000000 - ia;
  if (ib < 0) ib = (int32_t)0x80000000 - ib;
  return llabs((int64_t)ia - (int64_t)ib);
}
int main(int argc, char** argv) {
  float lo = argc > 1 ? (float)atof(argv[1]) : 0.0f, hi = argc > 2 ? (float)atof(argv[2]) : 6.2831855f;
  uint32_t a, b; memcpy(&a, &lo, 4); memcpy(&b, &hi, 4);
  int64_t maxS = 0, maxC = 0, offS = 0, offC = 0, n = 0;
#pragma omp parallel for reduction(max : maxS, maxC) reduction(+ : offS, offC, n) schedule(static)
  for (int64_t i = a; i <= (int64_t)b; i++) {
    uint32_t u = (uint32_t)i; float x; memcpy(&x, &u, 4);
    float s, c; sincosFixed(x, &s, &c);
    float es = (float)sin((double)x), ec = (float)cos((double)x);
    int64_t ds = ulps(s, es), dc = ulps(c, ec);
    if (ds > maxS) maxS = ds;
    if (dc > maxC) maxC = dc;
    offS += ds != 0; offC += dc != 0; n++;
  }
  printf("[%g, %g]: %lld values, sin max %lld ulp (%.3f%% not correctly rounded), cos max %lld ulp (%.3f%%)\n", lo, hi,
         (long long)n, (long long)maxS, 100.0 * offS / n, (long long)maxC, 100.0 * offC / n);
  return 0;
}
