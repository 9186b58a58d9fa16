// Arithmetic of the view-ray masks (scene_host.h: buildCoverageMask), shared by the host builder and the kernels of
// mask_gpu.hip so that both set the same bits: binary64 throughout, one IEEE operation per operator
// (-ffp-contract=off), division, sqrt, ceil and floor correctly rounded on both sides.
#pragma once
#include "pt_math.h"

namespace pt {

// A view ray through screen position (x, y) has direction dir + right*(x*tanFov) + up*(y*tanFov*aspect)
// (camera.d:38-42).  Point p (relative to the camera) lies on it iff p = t*(dir + a*right + b*up) with t > 0,
// x = a/tanFov, y = b/(tanFov*aspect): solved with the inverse of the matrix whose columns are dir, right, up.
struct MaskCam {
  double inv[3][3];
  double pos[3], dir[3], right[3], up[3];
  double sx, sy;            // tanFov, tanFov * aspect
  double tMin;              // a vertex closer to the camera plane than this is treated as "not in front"
  double delta;             // length of the largest jitter offset of a sample's un-normalised direction
  uint32_t W, H;
};

// scale: largest finite |coordinate| of the scene.  Returns false when the camera basis is singular.
PT_HD bool maskCamSetup(const Camera& cam, uint32_t W, uint32_t H, double scale, MaskCam& mc) {
  const double m[3][3] = {{cam.dir.x, cam.right.x, cam.up.x}, {cam.dir.y, cam.right.y, cam.up.y}, {cam.dir.z, cam.right.z, cam.up.z}};
  double det = m[0][0] * (m[1][1] * m[2][2] - m[1][2] * m[2][1]) - m[0][1] * (m[1][0] * m[2][2] - m[1][2] * m[2][0]) +
               m[0][2] * (m[1][0] * m[2][1] - m[1][1] * m[2][0]);
  mc.sx = (double)cam.tanFov; mc.sy = (double)cam.tanFov * (double)cam.aspect;
  if (!(fabs(det) > 1e-12) || !(fabs(mc.sx) > 1e-12) || !(fabs(mc.sy) > 1e-12)) return false;
  mc.inv[0][0] = (m[1][1] * m[2][2] - m[1][2] * m[2][1]) / det; mc.inv[0][1] = (m[0][2] * m[2][1] - m[0][1] * m[2][2]) / det;
  mc.inv[0][2] = (m[0][1] * m[1][2] - m[0][2] * m[1][1]) / det; mc.inv[1][0] = (m[1][2] * m[2][0] - m[1][0] * m[2][2]) / det;
  mc.inv[1][1] = (m[0][0] * m[2][2] - m[0][2] * m[2][0]) / det; mc.inv[1][2] = (m[0][2] * m[1][0] - m[0][0] * m[1][2]) / det;
  mc.inv[2][0] = (m[1][0] * m[2][1] - m[1][1] * m[2][0]) / det; mc.inv[2][1] = (m[0][1] * m[2][0] - m[0][0] * m[2][1]) / det;
  mc.inv[2][2] = (m[0][0] * m[1][1] - m[0][1] * m[1][0]) / det;
  mc.pos[0] = cam.pos.x; mc.pos[1] = cam.pos.y; mc.pos[2] = cam.pos.z;
  mc.dir[0] = cam.dir.x; mc.dir[1] = cam.dir.y; mc.dir[2] = cam.dir.z;
  mc.right[0] = cam.right.x; mc.right[1] = cam.right.y; mc.right[2] = cam.right.z;
  mc.up[0] = cam.up.x; mc.up[1] = cam.up.y; mc.up[2] = cam.up.z;
  double s = scale > 1e-30 ? scale : 1e-30;
  for (int k = 0; k < 3; k++) { double a = fabs(mc.pos[k]); if (a > s) s = a; }
  mc.tMin = 1e-6 * s;
  mc.W = W; mc.H = H;
  const double rl = sqrt((double)dot(cam.right, cam.right)), ul = sqrt((double)dot(cam.up, cam.up));
  mc.delta = sqrt((rl * mc.sx / W) * (rl * mc.sx / W) + (ul * mc.sy / H) * (ul * mc.sy / H));
  return true;
}

// Screen footprint of one triangle: the rows and columns of its bounding rectangle, grown by the sample jitter, and
// its three edge functions E_k(x, y) = A_k x + B_k y + C_k >= 0 inside (a sliver without usable orientation keeps the
// whole rectangle).
struct MaskTri { uint32_t ca, cb, ra, rb; int edges; double A[3], B[3], C[3]; };
enum { MASK_TRI_SKIP = 0, MASK_TRI_OK = 1, MASK_TRI_BEHIND = 2 };

// sample positions of pixel (col, row): x in col*2/W - 1 + [-1/W, 1/W), i.e. within half a pixel of the pixel's
// nominal position; margin: 0.05 pixel for the fp32 rounding of the device's ray set-up and triangle test
#define PT_MASK_MARGIN 0.55

PT_HD int maskTriSetup(const MaskCam& mc, const float* p, MaskTri& t) {
  for (int k = 0; k < 9; k++) if (!(fabs(p[k]) <= 3.4028234663852886e38)) return MASK_TRI_SKIP;     // never hit (no BVH leaf holds it)
  const double W = (double)mc.W, H = (double)mc.H, margin = PT_MASK_MARGIN;
  double x0 = 1e300, x1 = -1e300, y0 = 1e300, y1 = -1e300, vx[3], vy[3];
  for (int v = 0; v < 3; v++) {
    double q[3] = {(double)p[3 * v] - mc.pos[0], (double)p[3 * v + 1] - mc.pos[1], (double)p[3 * v + 2] - mc.pos[2]};
    double tt = mc.inv[0][0] * q[0] + mc.inv[0][1] * q[1] + mc.inv[0][2] * q[2];
    double a = mc.inv[1][0] * q[0] + mc.inv[1][1] * q[1] + mc.inv[1][2] * q[2];
    double b = mc.inv[2][0] * q[0] + mc.inv[2][1] * q[1] + mc.inv[2][2] * q[2];
    if (!(tt > mc.tMin)) return MASK_TRI_BEHIND;
    double px = (a / (tt * mc.sx) + 1.0) * 0.5 * W, py = (b / (tt * mc.sy) + 1.0) * 0.5 * H;      // in pixels
    vx[v] = px; vy[v] = py;
    x0 = x0 < px ? x0 : px; x1 = x1 > px ? x1 : px; y0 = y0 < py ? y0 : py; y1 = y1 > py ? y1 : py;
  }
  double c0 = ceil(x0 - margin), c1 = floor(x1 + margin), r0 = ceil(y0 - margin), r1 = floor(y1 + margin);
  if (c1 < 0 || r1 < 0 || c0 > W - 1 || r0 > H - 1) return MASK_TRI_SKIP;
  t.ca = (uint32_t)(c0 > 0.0 ? c0 : 0.0); t.cb = (uint32_t)(c1 < W - 1 ? c1 : W - 1);
  t.ra = (uint32_t)(r0 > 0.0 ? r0 : 0.0); t.rb = (uint32_t)(r1 < H - 1 ? r1 : H - 1);
  if (t.ca > t.cb || t.ra > t.rb) return MASK_TRI_SKIP;
  // All vertices are in front of the camera, so the projection of the triangle is the triangle of the projections.
  double area2 = (vx[1] - vx[0]) * (vy[2] - vy[0]) - (vx[2] - vx[0]) * (vy[1] - vy[0]);
  t.edges = fabs(area2) > 1e-9 ? 1 : 0;
  for (int k = 0; k < 3; k++) {
    int k1 = (k + 1) % 3;
    double sgn = area2 > 0.0 ? 1.0 : -1.0;
    t.A[k] = sgn * (vy[k] - vy[k1]); t.B[k] = sgn * (vx[k1] - vx[k]); t.C[k] = -(t.A[k] * vx[k] + t.B[k] * vy[k]);
  }
  return MASK_TRI_OK;
}

// Columns [a, b] of row r whose pixels' sample squares (half size = margin) reach the triangle: for every edge, the
// corner of the square farthest to the inner side must not be outside of it.  False when the row holds none.
PT_HD bool maskRowSpan(const MaskTri& t, uint32_t r, uint32_t& a, uint32_t& b) {
  const double margin = PT_MASK_MARGIN;
  double lo = (double)t.ca, hi = (double)t.cb;
  for (int k = 0; k < 3 && t.edges; k++) {
    double reach = margin * (fabs(t.A[k]) + fabs(t.B[k]));
    double kk = t.B[k] * (double)r + t.C[k] + reach + 1e-9 * (fabs(t.A[k]) + fabs(t.B[k]) + fabs(t.C[k]) + 1.0);
    if (t.A[k] > 0.0) { double c = ceil(-kk / t.A[k] - 1e-9); lo = lo > c ? lo : c; }
    else if (t.A[k] < 0.0) { double c = floor(-kk / t.A[k] + 1e-9); hi = hi < c ? hi : c; }
    else if (kk < 0.0) hi = lo - 1.0;
  }
  if (!(lo <= hi)) return false;
  a = (uint32_t)lo; b = (uint32_t)hi;
  return true;
}

// Whether no sample position of the (uncovered) pixel (c, r) can see sampleSky's sun disc (dot(normalized(1,1,6), d) >
// 0.997, rendering.d:332-339).  The direction of a sample differs from that of the pixel's nominal position by an
// angle of at most delta / (|v| - delta), v the un-normalised direction and delta the length of the jitter's offset to
// it; a cosine changes by no more than the angle does.
PT_HD bool maskConstantSky(const MaskCam& mc, uint32_t c, uint32_t r) {
  const double s38 = sqrt(38.0);
  const double sun[3] = {1.0 / s38, 1.0 / s38, 6.0 / s38};
  double y = (double)r / mc.H * 2.0 - 1.0, x = (double)c / mc.W * 2.0 - 1.0;
  double v[3];
  for (int k = 0; k < 3; k++) v[k] = mc.dir[k] + mc.right[k] * x * mc.sx + mc.up[k] * y * mc.sy;
  double len = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
  if (!(len > 2.0 * mc.delta)) return false;                                // degenerate direction: leave it to the kernel
  double cosSun = (v[0] * sun[0] + v[1] * sun[1] + v[2] * sun[2]) / len;
  return cosSun + mc.delta / (len - mc.delta) + 1e-5 < 0.997;
}

}  // namespace pt
