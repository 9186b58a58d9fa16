// =====================================================================================
// TEST INFRASTRUCTURE ONLY.  This file is the CPU oracle: a plain C++ restatement of the
// reference's per-pixel Monte Carlo integrator and ray-scene intersection.  Only tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the product path
// (pathtracer_amd/) never links, imports or calls anything in this directory.
//
// PARITY UNPINNED.  The reference (D2) has no tests, golden vectors or fixtures for this
// path, cannot be compiled here (no D compiler), and all of its arithmetic lives in
// Ingrater/thBase, a sibling checkout that is not part of /root/reference and has no
// version pin (PathTracer.visualdproj:50,86-87; PathTracer.sln:9).  The functions below
// follow the reference's own source line by line where it exists and *chosen*, documented
// semantics (S1..S12) where it calls into thBase.
//
// What follows what:
//   Camera               camera.d:16-42
//   getViewRay           rendering.d:278-285
//   pickRandomLightPoint rendering.d:164-193
//   light discovery      rendering.d:116-161
//   angleTo*Direction    rendering.d:309-330
//   sampleSky            rendering.d:332-339
//   computeLindirect     rendering.d:376-394   (dormant at HEAD; mode 1 here)
//   computeLdirect       rendering.d:396-462
//   computeL             rendering.d:464-484
//   computeOutputColor   rendering.d:494-500
//   sphere tree build    scene.d:589-609, 353-500, linearizeNodes scene.d:1089-1150
//   Scene.trace          scene.d:681-743      Scene.hitsNothing scene.d:801-838
//   Scene.trace (texcoord variant) scene.d:745-796, interpolate scene.d:68-105 (exhaustive form only)
//   bands + per-band RNG main.d:117,184-192   tone map main.d:82-95
//
// Chosen semantics (thBase is absent):
//   S1  dot(a,b)   = fma(az,bz, fma(ay,by, ax*bx));  cross(a,b).x = fma(ay,bz, -(az*by)) etc.
//   S2  normalized = v * (1.0f / sqrtf(dot(v,v)))
//   S3  Triangle.intersects = two-sided Moller-Trumbore, u along (v2-v0), v along (v1-v0)
//       (implied by rendering.d:177 and scene.d:711-723), |det| < FloatEpsilon rejects,
//       inv = 1/det (correctly rounded), u = dot(s,p)*inv ...
//   S4  FloatEpsilon = 1e-5 (configurable)
//   S5  Sphere.intersects(ray): origin inside, or nearest approach ahead and within r
//   S6  front face:  plane.normal . dir < 0.  plane.normal = normalize((v1-v0)x(v2-v0)), so
//       this is sign(det) < 0 with det = dir . ((v1-v0)x(v2-v0)); the oracle uses det's sign
//       (orc_frontface_disagreements() counts rays where the normalised form would differ).
//   S7  fastsqrt = sqrtf
//   S8  mat3(a,b,c) * v = a*v.x + b*v.y + c*v.z (columns), fma chain like S1
//   S9  vec3 default-initialises to 0
//   S10 uniform(a,b) = a + (b-a)*r, r = (x>>8)*2^-24 in [0,1)
//   S11 Triangle.area = 0.5*|(v1-v0)x(v2-v0)|
//   S12 A in B (spheres) <=> |cA-cB| + rA <= rB ; Sphere.radius = sqrtf(radiusSquared)
//   S13 cosf / sinf: one fixed binary32 evaluation (sincosS13), within 1 ulp of libm
//   S14 powf(x, 1/2.4) of the tone map: one fixed binary64 evaluation rounded once (powInv24S14)
// Everything else is evaluated left to right as the D source writes it, one IEEE float
// operation per operator, no contraction (compile with -ffp-contract=off).
// =====================================================================================
#include <cmath>
#include <cstdint>
#include <cstring>
#include <cfloat>
#include <vector>
#include <random>
#include <thread>
#include <atomic>
#include <algorithm>
#include <chrono>

namespace {

struct V3 { float x, y, z; };
inline V3 v3(float x, float y, float z) { return V3{x, y, z}; }
inline V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline V3 operator-(V3 a) { return v3(-a.x, -a.y, -a.z); }
inline V3 operator*(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
inline V3 operator*(float s, V3 a) { return v3(s * a.x, s * a.y, s * a.z); }
inline V3 operator*(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }   // component-wise, rendering.d:391,418,424
inline V3 operator/(V3 a, float s) { return v3(a.x / s, a.y / s, a.z / s); }
inline float dot(V3 a, V3 b) { return std::fmaf(a.z, b.z, std::fmaf(a.y, b.y, a.x * b.x)); }        // S1
inline V3 cross(V3 a, V3 b) {
  return v3(std::fmaf(a.y, b.z, -(a.z * b.y)), std::fmaf(a.z, b.x, -(a.x * b.z)), std::fmaf(a.x, b.y, -(a.y * b.x)));
}
inline float length(V3 a) { return sqrtf(dot(a, a)); }
inline V3 normalized(V3 a) { float inv = 1.0f / sqrtf(dot(a, a)); return a * inv; }                  // S2
inline V3 mat3cols(V3 c0, V3 c1, V3 c2, V3 v) {                                                      // S8
  return v3(std::fmaf(c2.x, v.z, std::fmaf(c1.x, v.y, c0.x * v.x)),
            std::fmaf(c2.y, v.z, std::fmaf(c1.y, v.y, c0.y * v.x)),
            std::fmaf(c2.z, v.z, std::fmaf(c1.z, v.y, c0.z * v.x)));
}

struct Ray { V3 o, d; };
struct Tri { V3 v0, v1, v2; };
struct TriData { V3 n[3]; uint32_t mat; float tex[3][2] = {{0, 0}, {0, 0}, {0, 0}}; };
struct Material { float emissive; V3 color; };
struct LightTri { V3 v0, e1, e2; float area, probability; };   // rendering.d:40-46
struct Pixel { float color[3]; float n; float sum[3]; };        // rendering.d:24-30
static_assert(sizeof(Pixel) == 28, "Pixel is 7 packed floats");

const float PI_F = 3.14159265358979323846f;
const float PI_2_F = 1.57079632679489661923f;
const float BRDF = 1.0f / PI_F;                                  // rendering.d:368

// ---------------------------------------------------------------- RNG
inline uint32_t rotl32(uint32_t x, int k) { return (x << k) | (x >> (32 - k)); }
inline uint32_t mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x21f0aaadu; x ^= x >> 15; x *= 0x735a2d97u; x ^= x >> 15; return x;
}
struct Rng {
  bool keyed = false;
  std::mt19937 mt;                 // Phobos Random == Mt19937 (main.d:109)
  uint32_t s0 = 1, s1 = 0;         // xoroshiro64** for the counter-keyed streams
  void seed_mt(uint32_t s) { keyed = false; mt.seed(s); }
  void seed_keyed(uint32_t seed, uint32_t pixel, uint32_t sample, uint32_t stream) {
    keyed = true;
    s0 = mix32(mix32(mix32(seed ^ 0x9E3779B9u) + pixel) ^ (sample * 0x85EBCA6Bu + stream));
    s1 = mix32(mix32(mix32(seed ^ 0x7F4A7C15u) + sample) ^ (pixel * 0xC2B2AE35u + stream * 0x27D4EB2Fu));
    if ((s0 | s1) == 0) s0 = 1;
  }
  uint32_t next() {
    if (!keyed) return (uint32_t)mt();
    uint32_t r = rotl32(s0 * 0x9E3779BBu, 5) * 5u;
    s1 ^= s0; s0 = rotl32(s0, 26) ^ s1 ^ (s1 << 9); s1 = rotl32(s1, 13);
    return r;
  }
  float u01() { return (float)(next() >> 8) * (1.0f / 16777216.0f); }
  float uniform(float a, float b) { return a + (b - a) * u01(); }     // S10
};

// ---------------------------------------------------------------- config + camera
struct OrcConfig {          // plain-old-data mirror of the source-level switches, rendering.d:19-21,54-55,72-74,379-380,466
  uint32_t width, height;
  uint32_t mode;            // 0 = HEAD (direct only), 1 = + 5 indirect chains (rendering.d:474-477)
  uint32_t hasSky, hasLights;
  float sunDir[3], sunRadiance[3], skyRadiance[3];
  float floatEpsilon, rrProb;
  uint32_t maxDepth, numIndirect, seed;
};

struct Camera { float tanFov, aspect; V3 dir, up, right, pos; };

Ray getScreenRay(const Camera& c, float x, float y) {             // camera.d:38-42
  V3 a = (c.right * x) * c.tanFov;
  V3 b = ((c.up * y) * c.tanFov) * c.aspect;
  V3 rd = normalized((c.dir + a) + b);
  return Ray{c.pos, rd};
}

// ---------------------------------------------------------------- triangle + sphere tests
inline bool triIntersects(const Tri& tr, const Ray& r, float eps, float& t, float& u, float& v, float& det_out) {  // S3
  V3 A = tr.v2 - tr.v0, B = tr.v1 - tr.v0;
  V3 p = cross(r.d, B);
  float det = dot(A, p);
  det_out = det;
  if (fabsf(det) < eps) return false;
  float inv = 1.0f / det;
  V3 s = r.o - tr.v0;
  u = dot(s, p) * inv;
  if (u < 0.0f || u > 1.0f) return false;
  V3 q = cross(s, A);
  v = dot(r.d, q) * inv;
  if (v < 0.0f || u + v > 1.0f) return false;
  t = dot(B, q) * inv;
  return true;
}

struct Node {               // scene.d:145-156 with indices instead of pointers; leaf <=> child[0] < 0
  V3 c; float r2; bool same; int child[2]; int tri;
};

inline bool sphereIntersects(const Node& n, const Ray& r) {         // S5
  V3 L = n.c - r.o;
  float ll = dot(L, L);
  if (ll <= n.r2) return true;
  float tca = dot(L, r.d);
  if (tca < 0.0f) return false;
  return ll - tca * tca <= n.r2;
}
inline bool sphereIn(const Node& a, const Node& b) {                 // S12
  return length(a.c - b.c) + sqrtf(a.r2) <= sqrtf(b.r2);
}

struct Counters { uint64_t trace = 0, occl = 0, nodes = 0, tris = 0; };

struct Scene {
  std::vector<Tri> tris; std::vector<TriData> data; std::vector<Material> mats;
  std::vector<Node> nodes; int root = -1; bool haveTree = false;
  std::vector<std::vector<LightTri>> lights; std::vector<float> lightArea;
  OrcConfig cfg; Camera cam;
  double buildSeconds = 0;

  // ------------------------------------------------------------ sphere tree (scene.d:329-637)
  int nextNode = 0;
  int creationViolations = 0;      // scene.d:383-384: both children inside the sphere made for them
  int nodeDepth(int n, int depth) const {                           // scene.d:388-394
    if (nodes[n].child[0] < 0) return depth;
    return std::max(nodeDepth(nodes[n].child[0], depth + 1), nodeDepth(nodes[n].child[1], depth + 1));
  }
  int mergeHelper(int a, int b, int newNode) {                      // scene.d:364-386
    if (newNode < 0) {
      newNode = nextNode++;
      float rA = sqrtf(nodes[a].r2), rB = sqrtf(nodes[b].r2);
      V3 through = nodes[b].c - nodes[a].c;
      float dist = length(through);
      through = normalized(through);
      nodes[newNode].c = ((nodes[a].c - (through * rA)) + (nodes[b].c + (through * rB))) * 0.5f;
      float nr = (rA + dist + rB) * 0.5f + 0.01f;
      nodes[newNode].r2 = nr * nr;
    }
    nodes[newNode].child[0] = a; nodes[newNode].child[1] = b; nodes[newNode].same = false; nodes[newNode].tri = -1;
    if (!sphereIn(nodes[a], nodes[newNode]) || !sphereIn(nodes[b], nodes[newNode])) creationViolations++;
    return newNode;
  }
  int insertInto(int ins, int bounding) {                           // scene.d:396-423
    int cur = bounding;
    while (true) {
      int c0 = nodes[cur].child[0], c1 = nodes[cur].child[1];
      if (nodes[c0].child[0] >= 0 && sphereIn(nodes[ins], nodes[c0])) { cur = c0; continue; }
      if (nodes[c1].child[0] >= 0 && sphereIn(nodes[ins], nodes[c1])) { cur = c1; continue; }
      break;
    }
    if (nodeDepth(nodes[cur].child[0], 0) < nodeDepth(nodes[cur].child[1], 0)) {
      int m = mergeNode(nodes[cur].child[0], ins); nodes[cur].child[0] = m;
    } else {
      int m = mergeNode(nodes[cur].child[1], ins); nodes[cur].child[1] = m;
    }
    return bounding;
  }
  int mergeNode(int a, int b) {                                     // scene.d:353-455
    if (a < 0 && b < 0) return -1;
    if (a < 0) return b;
    if (b < 0) return a;
    int newNode = -1;
    if (sphereIn(nodes[a], nodes[b])) {
      if (nodes[b].child[0] < 0) {
        newNode = nextNode++; nodes[newNode].c = nodes[b].c; nodes[newNode].r2 = nodes[b].r2; nodes[b].same = true;
      } else return insertInto(a, b);
    } else if (sphereIn(nodes[b], nodes[a])) {
      if (nodes[a].child[0] < 0) {
        newNode = nextNode++; nodes[newNode].c = nodes[a].c; nodes[newNode].r2 = nodes[a].r2; nodes[a].same = true;
      } else return insertInto(b, a);
    }
    return mergeHelper(a, b, newNode);
  }
  void buildTree() {
    auto t0 = std::chrono::steady_clock::now();
    size_t T = tris.size();
    nodes.assign(2 * T, Node{v3(0, 0, 0), 0, false, {-1, -1}, -1});  // scene.d:329
    nextNode = (int)T;
    std::vector<int> rem(T);
    for (size_t i = 0; i < T; i++) {                                  // scene.d:592-607
      const Tri& tr = tris[i];
      V3 c = (tr.v0 + tr.v1 + tr.v2) / 3.0f;
      Node& n = nodes[i]; rem[i] = (int)i;
      n.c = c; n.same = false;
      V3 d0 = tr.v0 - c, d1 = tr.v1 - c, d2 = tr.v2 - c;
      n.r2 = std::max(std::max(dot(d0, d0), dot(d1, d1)), dot(d2, d2)) + 0.01f;
      n.child[0] = n.child[1] = -1; n.tri = (int)i;
    }
    size_t k = 0;                                                     // bruteForceMerge, scene.d:457-500
    while (rem.size() > 1) {
      int a = rem[k]; V3 cp = nodes[a].c;
      size_t smallest = (k == 0) ? 1 : 0;
      V3 d = nodes[rem[smallest]].c - cp; float best = dot(d, d);
      for (size_t i = 0; i < rem.size(); i++) {
        if (i == k) continue;
        V3 e = nodes[rem[i]].c - cp; float dist = dot(e, e);
        if (dist < best) { best = dist; smallest = i; }
      }
      int b = rem[smallest];
      rem[k] = mergeNode(a, b);
      rem[smallest] = rem.back(); rem.pop_back();                     // removeAtIndexUnordered
      k++;
      if (k >= rem.size()) k = 0;
    }
    root = T ? rem[0] : -1;
    linearize();
    haveTree = true;
    buildSeconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  }
  void linearize() {                                                  // scene.d:1089-1150 (level order, root = 0)
    if (root < 0) return;
    std::vector<Node> out; out.reserve(nodes.size());
    std::vector<int> cur{root}, nxt, mapping(nodes.size(), -1);
    // writeNodeAtDepth visits each level left to right in child order == a BFS that keeps order
    while (!cur.empty()) {
      nxt.clear();
      for (int n : cur) {
        mapping[n] = (int)out.size(); out.push_back(nodes[n]);
        if (nodes[n].child[0] >= 0) { nxt.push_back(nodes[n].child[0]); nxt.push_back(nodes[n].child[1]); }
      }
      cur.swap(nxt);
    }
    for (Node& n : out) if (n.child[0] >= 0) { n.child[0] = mapping[n.child[0]]; n.child[1] = mapping[n.child[1]]; }
    nodes.swap(out); root = 0;
  }

  // ------------------------------------------------------------ shading normal (scene.d:711-723, S7)
  V3 interpNormal(const TriData& ld, float u, float v) const {
    float x = 1.0f / (u + v);
    float u1 = x * u, v1 = x * v;
    const float sqrt2 = 1.414213562f;
    float d1 = sqrtf((1.0f - u1) * (1.0f - u1) + v1 * v1) / sqrt2;
    float d2 = sqrtf(u1 * u1 + (1.0f - v1) * (1.0f - v1)) / sqrt2;
    V3 interpolated1 = ld.n[1] * d1 + ld.n[2] * d2;
    float len = sqrtf(u1 * u1 + v1 * v1);
    float i1 = sqrtf(u * u + v * v) / len;
    float i2 = 1.0f - i1;
    return normalized(ld.n[0] * i2 + interpolated1 * i1);
  }

  struct Hit { float t, u, v; int tri; };

  // Scene.trace over the sphere tree, scene.d:681-743.  Frontier arrays are per caller (thread local in D).
  bool traceTree(const Ray& ray, Hit& h, std::vector<int>& readFrom, std::vector<int>& writeTo, Counters& cn) const {
    float rayPos = FLT_MAX; bool result = false; h.tri = -1;
    float eps = cfg.floatEpsilon;
    size_t numReads = 1, nextWrite = 0; readFrom[0] = root;
    while (numReads > 0) {
      for (size_t i = 0; i < numReads; i++) {
        const Node& node = nodes[readFrom[i]];
        cn.nodes++;
        if (node.same || sphereIntersects(node, ray)) {
          if (node.child[0] < 0) {
            float pos = -1.0f, u = 0, v = 0, det;
            cn.tris++;
            if (triIntersects(tris[node.tri], ray, eps, pos, u, v, det) && pos < rayPos && pos >= eps) {
              if (det < 0.0f) {                                         // S6 (scene.d:704-705)
                rayPos = pos; h.t = pos; h.u = u; h.v = v; h.tri = node.tri; result = true;
              }
            }
          } else { writeTo[nextWrite++] = node.child[0]; writeTo[nextWrite++] = node.child[1]; }
        }
      }
      numReads = nextWrite; nextWrite = 0; readFrom.swap(writeTo);
    }
    return result;
  }
  bool hitsNothingTree(const Ray& ray, std::vector<int>& readFrom, std::vector<int>& writeTo, Counters& cn) const {  // scene.d:801-838
    float eps = cfg.floatEpsilon;
    size_t numReads = 1, nextWrite = 0; readFrom[0] = root;
    while (numReads > 0) {
      for (size_t i = 0; i < numReads; i++) {
        const Node& node = nodes[readFrom[i]];
        cn.nodes++;
        if (node.same || sphereIntersects(node, ray)) {
          if (node.child[0] < 0) {
            float pos = -1.0f, u = 0, v = 0, det;
            cn.tris++;
            if (triIntersects(tris[node.tri], ray, eps, pos, u, v, det) && pos >= eps) return false;
          } else { writeTo[nextWrite++] = node.child[0]; writeTo[nextWrite++] = node.child[1]; }
        }
      }
      numReads = nextWrite; nextWrite = 0; readFrom.swap(writeTo);
    }
    return true;
  }
  // Oracle-B: the same acceptance predicates over ALL triangles; ties keep the lowest triangle index.
  bool traceBrute(const Ray& ray, Hit& h, Counters& cn) const {
    float rayPos = FLT_MAX; bool result = false; h.tri = -1; float eps = cfg.floatEpsilon;
    for (size_t i = 0; i < tris.size(); i++) {
      float pos = -1.0f, u = 0, v = 0, det; cn.tris++;
      if (triIntersects(tris[i], ray, eps, pos, u, v, det) && pos < rayPos && pos >= eps && det < 0.0f) {
        rayPos = pos; h.t = pos; h.u = u; h.v = v; h.tri = (int)i; result = true;
      }
    }
    return result;
  }
  // interpolate(u, v, val0, val1, val2) for one float component, scene.d:68-105
  static float interpolate1(float u, float v, float val0, float val1, float val2) {
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
  // Scene.trace, texcoord variant (scene.d:745-796) over ALL triangles; ties keep the lowest triangle index.
  // Returns TraceResult: 0 FrontFaceHit, 1 BackFaceHit, 2 NoHit (scene.d:55-60).
  int traceTexBrute(const Ray& ray, bool ignoreBackfaces, float& rayPos, float tc[2], int& tri) const {
    rayPos = FLT_MAX; int result = 2; tri = -1; tc[0] = tc[1] = 0.0f; float eps = cfg.floatEpsilon;
    for (size_t i = 0; i < tris.size(); i++) {
      float pos = -1.0f, u = 0, v = 0, det;
      if (triIntersects(tris[i], ray, eps, pos, u, v, det) && pos < rayPos && pos >= eps) {
        bool isFrontFace = det < 0.0f;                                   // S6
        if (!ignoreBackfaces || isFrontFace) {
          rayPos = pos; tri = (int)i;
          const TriData& ld = data[i];
          tc[0] = interpolate1(u, v, ld.tex[0][0], ld.tex[1][0], ld.tex[2][0]);
          tc[1] = interpolate1(u, v, ld.tex[0][1], ld.tex[1][1], ld.tex[2][1]);
          result = isFrontFace ? 0 : 1;
        }
      }
    }
    return result;
  }
  bool hitsNothingBrute(const Ray& ray, Counters& cn) const {
    float eps = cfg.floatEpsilon;
    for (size_t i = 0; i < tris.size(); i++) {
      float pos = -1.0f, u = 0, v = 0, det; cn.tris++;
      if (triIntersects(tris[i], ray, eps, pos, u, v, det) && pos >= eps) return false;
    }
    return true;
  }

  // ------------------------------------------------------------ lights (rendering.d:116-161)
  void findLights() {
    std::vector<int> matToLight(mats.size(), -1); int nl = 0;
    for (size_t m = 0; m < mats.size(); m++) if (mats[m].emissive > 0.0f) matToLight[m] = nl++;
    lights.assign(nl, {}); lightArea.assign(nl, 0.0f);
    for (size_t i = 0; i < tris.size(); i++) {
      const Material& mt = mats[data[i].mat];
      if (mt.emissive > 0.0f) {
        int li = matToLight[data[i].mat];
        LightTri lt; lt.v0 = tris[i].v0; lt.e1 = tris[i].v1 - tris[i].v0; lt.e2 = tris[i].v2 - tris[i].v0;
        lt.area = 0.5f * length(cross(lt.e1, lt.e2));                  // S11
        lt.probability = 0;
        lights[li].push_back(lt); lightArea[li] += lt.area;
      }
    }
    for (int i = 0; i < nl; i++) for (LightTri& l : lights[i]) l.probability = l.area / lightArea[i];
  }
};

// ---------------------------------------------------------------- integrator
struct Tracer {             // one per worker thread: D's thread-local frontiers, scene.d:32-47
  const Scene& sc; bool useTree; std::vector<int> a, b; Counters cn;
  Tracer(const Scene& s, bool tree) : sc(s), useTree(tree), a(std::max<size_t>(s.tris.size(), 2)), b(std::max<size_t>(s.tris.size(), 2)) {}
  bool trace(const Ray& r, float& rayPos, V3& normal, uint32_t& mat) {
    Scene::Hit h; cn.trace++;
    bool ok = useTree ? sc.traceTree(r, h, a, b, cn) : sc.traceBrute(r, h, cn);
    if (ok) { rayPos = h.t; normal = sc.interpNormal(sc.data[h.tri], h.u, h.v); mat = sc.data[h.tri].mat; }
    else rayPos = FLT_MAX;
    return ok;
  }
  bool hitsNothing(const Ray& r) { cn.occl++; return useTree ? sc.hitsNothingTree(r, a, b, cn) : sc.hitsNothingBrute(r, cn); }
};

V3 pickRandomLightPoint(const Scene& sc, Rng& gen, uint32_t li) {   // rendering.d:164-193
  float a = gen.uniform(0.0f, 1.0f);
  const std::vector<LightTri>& L = sc.lights[li];
  for (const LightTri& light : L) {
    if (light.probability > a) {
      while (true) {
        float u = gen.uniform(0.0f, 1.0f), v = gen.uniform(0.0f, 1.0f);
        if (u + v <= 1.0f) return light.v0 + light.e1 * v + light.e2 * u;
      }
    }
    a -= light.probability;
  }
  const LightTri& light = L[L.size() - 1];
  while (true) {
    float u = gen.uniform(0.0f, 1.0f), v = gen.uniform(0.0f, 1.0f);
    if (u + v <= 1.0f) return light.v0 + light.e1 * v + light.e2 * u;
  }
}

// cosf / sinf of the reference (rendering.d:309-330), S13: one fixed binary32 evaluation -- k = rint(x * 2/pi),
// r = x - k*pi/2 with pi/2 in four pieces (the first two short enough for k * piece to be exact), Taylor sin (through
// r^9) and cos (through r^10) by Horner with fma, quadrant from k -- so that every implementation of this sequence
// returns the same bits.  Never more than 1 ulp from the correctly rounded value on [0, 100] (every binary32 was
// checked) and hence within 1 ulp of libm (test_oracle.py); the product evaluates the same sequence
// (pt_math.h: sincosFixed).
// g_libmTrig: the library calls the reference makes instead (orc_set_libm_trig; never set by the parity tests -- it
// exists to bound, over whole images, what the choice of S13 over this host's libm changes).
bool g_libmTrig = false;
void sincosS13(float x, float& sn, float& cs) {
  if (g_libmTrig) { sn = sinf(x); cs = cosf(x); return; }
  static const float PIO2[4] = {1.5703125f, 4.837512969970703125e-4f, 7.549790126404332e-8f, -1.7151245100058819e-15f};
  static const float SC[4] = {2.7557314297e-6f, -1.9841270114e-4f, 8.3333337680e-3f, -1.6666667163e-1f};
  static const float CC[5] = {-2.7557314297e-7f, 2.4801587642e-5f, -1.3888889225e-3f, 4.1666667908e-2f, -0.5f};
  float k = std::rint(x * 0.636619772f);
  float r = x;
  for (int i = 0; i < 4; i++) r = std::fma(-k, PIO2[i], r);
  float r2 = r * r;
  float ps = SC[0]; for (int i = 1; i < 4; i++) ps = std::fma(ps, r2, SC[i]);
  float pc = CC[0]; for (int i = 1; i < 5; i++) pc = std::fma(pc, r2, CC[i]);
  float s = std::fma(r * r2, ps, r), c = std::fma(r2, pc, 1.0f);
  switch ((int)k & 3) {
    case 0: sn = s; cs = c; break;
    case 1: sn = c; cs = -s; break;
    case 2: sn = -s; cs = -c; break;
    default: sn = -c; cs = s; break;
  }
}
V3 angleToLocalDirection(float phi, float psi) {                    // rendering.d:309-314
  float sinPhi, cosPhi, sinPsi, cosPsi;
  sincosS13(phi, sinPhi, cosPhi); sincosS13(psi, sinPsi, cosPsi);
  return v3(cosPsi * cosPhi, sinPsi * cosPhi, sinPhi);
}
V3 angleToDirection(float phi, float psi, V3 normal) {              // rendering.d:316-330
  V3 localDir = angleToLocalDirection(phi, psi);
  V3 up = normal, dir = v3(1, 0, 0);
  if (fabsf(dot(dir, up)) > 0.9f) dir = v3(0, 1, 0);
  V3 right = normalized(cross(up, dir));
  dir = normalized(cross(up, right));
  return mat3cols(dir, right, up, localDir);
}
V3 sampleSky(V3 dir) {                                              // rendering.d:332-339
  V3 sunDir = normalized(v3(1, 1, 6));
  if (dot(sunDir, dir) > 0.997f) return v3(1.0f, 0.984f, 0.8f) * 30.0f;
  return v3(0.682f, 0.977f, 1.0f);
}

V3 computeLdirect(Tracer& tr, V3 x, V3 normalX, uint32_t mat, Rng& gen) {   // rendering.d:396-462
  const Scene& sc = tr.sc; const OrcConfig& c = sc.cfg; float eps = c.floatEpsilon;
  V3 L = v3(0, 0, 0);                                                // S9
  V3 color = sc.mats[mat].color;
  if (c.hasLights) {
    for (uint32_t li = 0; li < sc.lights.size(); li++) {
      V3 y = pickRandomLightPoint(sc, gen, li);
      float lightDistance = length(y - x);
      V3 phi = normalized(y - x);
      Ray shadowRay{x + normalX * eps, phi};
      float distanceY = 0.0f; V3 normalY = v3(0, 0, 0); uint32_t matY = 0;
      if (dot(normalX, phi) > eps && tr.trace(shadowRay, distanceY, normalY, matY)) {
        if (distanceY > lightDistance - eps) {
          V3 Le = sc.mats[matY].emissive * sc.mats[matY].color;
          float G = dot(normalX, phi) * dot(normalY, -phi) / (lightDistance * lightDistance);
          if (!(G > 0.0f)) G = 0.0f;
          L = L + (BRDF * color) * Le * G * sc.lightArea[li];
        }
      }
    }
  }
  if (c.hasSky) {
    V3 sunDir = v3(c.sunDir[0], c.sunDir[1], c.sunDir[2]);
    {
      float psi = gen.uniform(0.0f, 2.0f * PI_F);
      float phi = gen.uniform(PI_2_F - PI_2_F * 0.01f, PI_2_F);
      V3 sampleDir = angleToDirection(phi, psi, sunDir);
      if (dot(normalX, sampleDir) > eps) {
        Ray sunRay{x + normalX * 0.1f, sampleDir};
        if (tr.hitsNothing(sunRay))
          L = L + (BRDF * color) * v3(c.sunRadiance[0], c.sunRadiance[1], c.sunRadiance[2]) * dot(normalX, sunDir);
      }
    }
    {
      float psi = gen.uniform(0.0f, 2.0f * PI_F);
      float phi = gen.uniform(0.0f, PI_2_F);
      V3 sampleDir = angleToLocalDirection(phi, psi);
      if (dot(normalX, sampleDir) > eps) {
        Ray sampleRay{x + normalX * 0.1f, sampleDir};
        if (tr.hitsNothing(sampleRay))
          L = L + (BRDF * color) * v3(c.skyRadiance[0], c.skyRadiance[1], c.skyRadiance[2]) * dot(normalX, sampleDir);
      }
    }
  }
  return L;
}

V3 computeLindirect(Tracer& tr, V3 pos, V3 normal, uint32_t mat, Rng& gen, uint32_t depth);
V3 computeLrefl(Tracer& tr, V3 pos, V3 normal, uint32_t mat, Rng& gen, uint32_t depth) {   // rendering.d:370-374
  V3 d = computeLdirect(tr, pos, normal, mat, gen);
  return d + computeLindirect(tr, pos, normal, mat, gen, depth);
}
V3 computeLindirect(Tracer& tr, V3 pos, V3 normal, uint32_t mat, Rng& gen, uint32_t depth) {   // rendering.d:376-394
  const OrcConfig& c = tr.sc.cfg;
  float a = gen.uniform(0.0f, 1.0f);
  if (a > c.rrProb || depth > c.maxDepth) return v3(0, 0, 0);
  float psi = gen.uniform(0.0f, 2.0f * PI_F);
  float phi = gen.uniform(0.0f, PI_2_F);
  V3 sampleDir = angleToDirection(phi, psi, normal);
  Ray sampleRay{pos + normal * 0.1f, sampleDir};
  float hitDistance = 0.0f; V3 hitNormal = v3(0, 0, 0); uint32_t hitMat = 0;
  if (tr.trace(sampleRay, hitDistance, hitNormal, hitMat)) {
    V3 hitPos = sampleRay.o + sampleRay.d * hitDistance;
    // BRDF * PI * color: (1/pi)*pi is 1 to within an ulp in D's real arithmetic -> color
    return tr.sc.mats[mat].color * computeLrefl(tr, hitPos, hitNormal, hitMat, gen, depth + 1);
  }
  return v3(0, 0, 0);
}

// getViewRay, rendering.d:278-285.  jitter=false is the parity probe (no RNG draw).
Ray getViewRay(const Scene& sc, uint32_t pixelIndex, Rng* gen) {
  uint32_t W = sc.cfg.width, H = sc.cfg.height;
  float psx = 1.0f / (float)W, psy = 1.0f / (float)H;
  float x = (float)(pixelIndex % W) / (float)W * 2.0f - 1.0f;
  if (gen) x = x + gen->uniform(-psx, psx);
  float y = (float)(pixelIndex / W) / (float)H * 2.0f - 1.0f;
  if (gen) y = y + gen->uniform(-psy, psy);
  return getScreenRay(sc.cam, x, y);
}

// computeL, rendering.d:464-484.  In keyed mode chain i draws from its own stream (i+1); in
// reference-shaped mode everything draws from the band's generator in source order.
void computeL(Tracer& tr, uint32_t pixelIndex, Pixel& px, Rng& gen, bool keyed, uint32_t sampleIndex) {
  const Scene& sc = tr.sc; const OrcConfig& c = sc.cfg;
  if (keyed) gen.seed_keyed(c.seed, pixelIndex, sampleIndex, 0);
  Ray viewRay = getViewRay(sc, pixelIndex, &gen);
  float rayPos = 0.0f; V3 normal = v3(0, 0, 0); uint32_t mat = 0;
  V3 add;
  if (tr.trace(viewRay, rayPos, normal, mat)) {
    V3 hitPos = viewRay.o + viewRay.d * rayPos;
    V3 sum = v3(0, 0, 0);
    if (c.mode == 1) {
      for (uint32_t i = 0; i < c.numIndirect; i++) {
        if (keyed) { Rng cg; cg.seed_keyed(c.seed, pixelIndex, sampleIndex, i + 1); sum = sum + computeLindirect(tr, hitPos, normal, mat, cg, 0); }
        else sum = sum + computeLindirect(tr, hitPos, normal, mat, gen, 0);
      }
    }
    add = sc.mats[mat].emissive * sc.mats[mat].color + computeLdirect(tr, hitPos, normal, mat, gen);
    if (c.mode == 1) add = add + sum / (float)c.numIndirect;
  } else {
    add = sampleSky(viewRay.d);
  }
  px.sum[0] += add.x; px.sum[1] += add.y; px.sum[2] += add.z;
  px.n += 1;
  px.color[0] = px.sum[0] / px.n; px.color[1] = px.sum[1] / px.n; px.color[2] = px.sum[2] / px.n;
}

}  // namespace

// =====================================================================================
// C interface (ctypes)
// =====================================================================================
extern "C" {

void* orc_create(const float* v, const float* n, const uint32_t* matIdx, const float* mats, uint32_t T, uint32_t M,
                 int buildTree) {
  Scene* s = new Scene();
  s->tris.resize(T); s->data.resize(T); s->mats.resize(M);
  for (uint32_t i = 0; i < T; i++) {
    const float* p = v + 9 * i; const float* q = n + 9 * i;
    s->tris[i] = Tri{v3(p[0], p[1], p[2]), v3(p[3], p[4], p[5]), v3(p[6], p[7], p[8])};
    for (int k = 0; k < 3; k++) s->data[i].n[k] = v3(q[3 * k], q[3 * k + 1], q[3 * k + 2]);
    s->data[i].mat = matIdx[i];
  }
  for (uint32_t i = 0; i < M; i++) s->mats[i] = Material{mats[4 * i], v3(mats[4 * i + 1], mats[4 * i + 2], mats[4 * i + 3])};
  std::memset(&s->cfg, 0, sizeof(s->cfg));
  s->cfg.width = 320; s->cfg.height = 240; s->cfg.hasLights = 1; s->cfg.floatEpsilon = 1e-5f; s->cfg.rrProb = 0.7f;
  s->cfg.maxDepth = 5; s->cfg.numIndirect = 5;
  s->findLights();
  if (buildTree) s->buildTree();
  return s;
}
void orc_destroy(void* h) { delete (Scene*)h; }
void orc_set_config(void* h, const OrcConfig* c) { ((Scene*)h)->cfg = *c; }
void orc_set_camera(void* h, const float* pos, const float* dir, const float* right, const float* up, float tanFov, float aspect) {
  Camera& c = ((Scene*)h)->cam;
  c.pos = v3(pos[0], pos[1], pos[2]); c.dir = v3(dir[0], dir[1], dir[2]); c.right = v3(right[0], right[1], right[2]);
  c.up = v3(up[0], up[1], up[2]); c.tanFov = tanFov; c.aspect = aspect;
}
// Camera ctor + setTransform, camera.d:16-29.  out = pos[3] dir[3] right[3] up[3] tanFov aspect
void orc_camera_lookat(float fovDeg, float aspect, const float* from, const float* to, const float* up, float* out) {
  V3 f = v3(from[0], from[1], from[2]), t = v3(to[0], to[1], to[2]), u = v3(up[0], up[1], up[2]);
  V3 dir = normalized(t - f);
  V3 right = normalized(cross(u, dir));
  V3 up2 = normalized(cross(right, dir));
  float tanFov = (float)std::tan((double)(fovDeg / 180.0f) * 3.14159265358979323846);
  float r[14] = {f.x, f.y, f.z, dir.x, dir.y, dir.z, right.x, right.y, right.z, up2.x, up2.y, up2.z, tanFov, aspect};
  std::memcpy(out, r, sizeof(r));
}
double orc_tree_build_seconds(void* h) { return ((Scene*)h)->buildSeconds; }
uint32_t orc_num_nodes(void* h) { return (uint32_t)((Scene*)h)->nodes.size(); }
uint32_t orc_num_light_groups(void* h) { return (uint32_t)((Scene*)h)->lights.size(); }
uint32_t orc_light_group(void* h, uint32_t g, float* out11, uint32_t cap, float* totalArea) {   // v0 e1 e2 area prob
  Scene* s = (Scene*)h; const auto& L = s->lights[g];
  for (uint32_t i = 0; i < L.size() && i < cap; i++) std::memcpy(out11 + 11 * i, &L[i], 44);
  *totalArea = s->lightArea[g];
  return (uint32_t)L.size();
}
// Sphere-tree invariants the reference asserts (scene.d:347,383-384,498): returns 0 when they hold.
// Containment is asserted by the reference only when a parent sphere is created; insertInto
// (scene.d:414-421) may later replace a child by a larger merged node, so it is not re-checked here.
int orc_check_tree(void* h) {
  Scene* s = (Scene*)h; if (!s->haveTree) return -1;
  size_t T = s->tris.size(); if (s->nodes.size() != 2 * T - 1) return 1;
  std::vector<int> seen(T, 0); size_t leaves = 0;
  for (const Node& n : s->nodes) {
    if (!(n.r2 > 0.0f)) return 2;
    if (n.child[0] < 0) { if (n.tri < 0 || n.tri >= (int)T || seen[n.tri]++) return 3; leaves++; }
  }
  if (s->creationViolations) return 4;
  return leaves == T ? 0 : 5;
}

// Primary-ray probe without jitter: closest front-face hit per pixel.  useTree=0 is Oracle-B.
void orc_trace_primary(void* h, uint32_t pixelOffset, uint32_t count, int useTree, int32_t* triId, float* t, float* u, float* v) {
  Scene* s = (Scene*)h;
  unsigned nt = std::max(1u, std::min(8u, std::thread::hardware_concurrency()));
  std::vector<std::thread> th;
  for (unsigned k = 0; k < nt; k++) th.emplace_back([=]() {
    Tracer tr(*s, useTree != 0);
    for (uint32_t i = k; i < count; i += nt) {
      Ray r = getViewRay(*s, pixelOffset + i, nullptr);
      Scene::Hit hh; bool ok = useTree ? s->traceTree(r, hh, tr.a, tr.b, tr.cn) : s->traceBrute(r, hh, tr.cn);
      triId[i] = ok ? hh.tri : -1; t[i] = ok ? hh.t : FLT_MAX; u[i] = ok ? hh.u : 0; v[i] = ok ? hh.v : 0;
    }
  });
  for (auto& x : th) x.join();
}
// S6 cross-check: number of primary rays (no jitter) whose accepted hit would change if the
// front-face test used the normalised plane normal (scene.d:704-705 as written) instead of sign(det).
uint32_t orc_frontface_disagreements(void* h, uint32_t pixelOffset, uint32_t count) {
  Scene* s = (Scene*)h; uint32_t bad = 0; float eps = s->cfg.floatEpsilon;
  for (uint32_t i = 0; i < count; i++) {
    Ray r = getViewRay(*s, pixelOffset + i, nullptr);
    for (const Tri& tr : s->tris) {
      float pos, u, v, det;
      if (triIntersects(tr, r, eps, pos, u, v, det) && pos >= eps) {
        V3 pn = normalized(cross(tr.v1 - tr.v0, tr.v2 - tr.v0));
        bool a = det < 0.0f, b = dot(pn, r.d) < 0.0f;
        if (a != b) bad++;
      }
    }
  }
  return bad;
}
// View rays themselves (no jitter), for bit-comparison with the product's ray generation: out = o[3] d[3]
void orc_view_rays(void* h, uint32_t pixelOffset, uint32_t count, float* out6) {
  Scene* s = (Scene*)h;
  for (uint32_t i = 0; i < count; i++) {
    Ray r = getViewRay(*s, pixelOffset + i, nullptr);
    float o[6] = {r.o.x, r.o.y, r.o.z, r.d.x, r.d.y, r.d.z}; std::memcpy(out6 + 6 * i, o, 24);
  }
}
// Arbitrary rays: kind 0 = trace (closest front face), kind 1 = hitsNothing.  rays = o[3] d[3] each.
void orc_trace_rays(void* h, const float* rays6, uint32_t count, int kind, int useTree, int32_t* triId, float* t, float* u, float* v) {
  Scene* s = (Scene*)h; Tracer tr(*s, useTree != 0);
  for (uint32_t i = 0; i < count; i++) {
    const float* p = rays6 + 6 * i; Ray r{v3(p[0], p[1], p[2]), v3(p[3], p[4], p[5])};
    if (kind == 0) {
      Scene::Hit hh; bool ok = useTree ? s->traceTree(r, hh, tr.a, tr.b, tr.cn) : s->traceBrute(r, hh, tr.cn);
      triId[i] = ok ? hh.tri : -1; t[i] = ok ? hh.t : FLT_MAX; u[i] = ok ? hh.u : 0; v[i] = ok ? hh.v : 0;
    } else {
      triId[i] = tr.hitsNothing(r) ? -1 : 0;
    }
  }
}
// Texture coordinates per triangle vertex (scene.d:305-307): tex6 = T x 3 x (s, t)
void orc_set_texcoords(void* h, const float* tex6) {
  Scene* s = (Scene*)h;
  for (size_t i = 0; i < s->data.size(); i++)
    for (int k = 0; k < 3; k++) { s->data[i].tex[k][0] = tex6[6 * i + 2 * k]; s->data[i].tex[k][1] = tex6[6 * i + 2 * k + 1]; }
}
// Scene.trace, texcoord variant: result = TraceResult (0 front, 1 back, 2 none), tc2 = interpolated texcoords
void orc_trace_tex(void* h, const float* rays6, uint32_t count, int ignoreBackfaces, int32_t* result, int32_t* triId, float* t,
                   float* tc2) {
  Scene* s = (Scene*)h;
  for (uint32_t i = 0; i < count; i++) {
    const float* p = rays6 + 6 * i; Ray r{v3(p[0], p[1], p[2]), v3(p[3], p[4], p[5])};
    int tri; result[i] = s->traceTexBrute(r, ignoreBackfaces != 0, t[i], tc2 + 2 * i, tri); triId[i] = tri;
  }
}
void orc_interp_normal(void* h, uint32_t tri, float u, float v, float* out3) {
  Scene* s = (Scene*)h; V3 n = s->interpNormal(s->data[tri], u, v); out3[0] = n.x; out3[1] = n.y; out3[2] = n.z;
}

// Reference-shaped renderer: bands of 4 scanlines, one Mt19937 per band seeded with
// pixelOffset + pixels[0].n (main.d:117,184-192), carried across the spp passes, nthreads workers
// pulling bands with a frame-wide barrier per pass (main.d:207-215,273-288).
// counters = {trace calls, hitsNothing calls, nodes visited, triangles tested}
double orc_render_mt(void* h, Pixel* pixels, uint32_t spp, uint32_t nthreads, int useTree, uint64_t* counters) {
  Scene* s = (Scene*)h; uint32_t W = s->cfg.width, H = s->cfg.height;
  uint32_t step = W * 4, steps = (W * H) / step;
  std::vector<Rng> gens(steps);
  for (uint32_t i = 0; i < steps; i++) gens[i].seed_mt(i * step + (uint32_t)pixels[i * step].n);
  if (nthreads < 1) nthreads = 1;
  std::vector<Counters> cns(nthreads);
  auto t0 = std::chrono::steady_clock::now();
  for (uint32_t pass = 0; pass < spp; pass++) {
    std::atomic<uint32_t> next{0};
    auto work = [&](uint32_t tid) {
      Tracer tr(*s, useTree != 0);
      while (true) {
        uint32_t b = next.fetch_add(1); if (b >= steps) break;
        uint32_t off = b * step;
        for (uint32_t i = 0; i < step; i++) computeL(tr, off + i, pixels[off + i], gens[b], false, 0);   // computeOutputColor
      }
      cns[tid].trace += tr.cn.trace; cns[tid].occl += tr.cn.occl; cns[tid].nodes += tr.cn.nodes; cns[tid].tris += tr.cn.tris;
    };
    std::vector<std::thread> th;
    for (uint32_t k = 1; k < nthreads; k++) th.emplace_back(work, k);
    work(0);
    for (auto& x : th) x.join();
  }
  double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  if (counters) { counters[0] = counters[1] = counters[2] = counters[3] = 0;
    for (auto& c : cns) { counters[0] += c.trace; counters[1] += c.occl; counters[2] += c.nodes; counters[3] += c.tris; } }
  return sec;
}

// Same estimator with counter-keyed RNG streams (seed, pixel, sample, stream): sample index of a
// pixel = sampleBase + (uint)pixel.n * sampleStride.  Independent of threads and of how pixels are partitioned.
double orc_render_keyed(void* h, uint32_t pixelOffset, uint32_t count, Pixel* pixels, uint32_t spp, uint32_t sampleBase,
                        uint32_t sampleStride, uint32_t nthreads, int useTree, uint64_t* counters) {
  Scene* s = (Scene*)h;
  if (nthreads < 1) nthreads = 1;
  std::vector<Counters> cns(nthreads);
  auto t0 = std::chrono::steady_clock::now();
  std::atomic<uint32_t> next{0};
  const uint32_t chunk = 256;
  auto work = [&](uint32_t tid) {
    Tracer tr(*s, useTree != 0); Rng gen;
    while (true) {
      uint32_t b = next.fetch_add(chunk); if (b >= count) break;
      uint32_t e = std::min(count, b + chunk);
      for (uint32_t i = b; i < e; i++)
        for (uint32_t k = 0; k < spp; k++)
          computeL(tr, pixelOffset + i, pixels[i], gen, true, sampleBase + (uint32_t)pixels[i].n * sampleStride);
    }
    cns[tid] = tr.cn;
  };
  std::vector<std::thread> th;
  for (uint32_t k = 1; k < nthreads; k++) th.emplace_back(work, k);
  work(0);
  for (auto& x : th) x.join();
  double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  if (counters) { counters[0] = counters[1] = counters[2] = counters[3] = 0;
    for (auto& c : cns) { counters[0] += c.trace; counters[1] += c.occl; counters[2] += c.nodes; counters[3] += c.tris; } }
  return sec;
}

// The raw RNG stream, so the product's device generator can be compared word for word.
void orc_keyed_stream(uint32_t seed, uint32_t pixel, uint32_t sample, uint32_t stream, uint32_t n, uint32_t* out) {
  Rng g; g.seed_keyed(seed, pixel, sample, stream);
  for (uint32_t i = 0; i < n; i++) out[i] = g.next();
}

// pow(c, 1/2.4) of drawScreen (main.d:92), S14: binary64, one fixed order, rounded to binary32 once.  c = m * 2^e
// with m in [sqrt(1/2), sqrt(2)); ln m = 2 f (1 + f^2/3 + ... + f^12/13), f = (m-1)/(m+1); y = (e ln2 + ln m) / 2.4
// (as a product with the double 1/2.4); n = rint(y / ln2); exp(y - n ln2) by Taylor through the 11th power; * 2^n.
// Within 1 ulp of libm's powf (test_oracle.py); the product evaluates the same sequence (pt_math.h: powInv24Fixed).
float powInv24S14(float c) {
  double xd = (double)c;
  uint64_t bits; std::memcpy(&bits, &xd, 8);
  int e = (int)((bits >> 52) & 0x7FFu) - 1023;
  bits = (bits & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull;
  double m; std::memcpy(&m, &bits, 8);
  if (m > 1.41421356237309514547) { m = m * 0.5; e = e + 1; }
  double f = (m - 1.0) / (m + 1.0), f2 = f * f;
  static const double LC[7] = {1.0 / 13.0, 1.0 / 11.0, 1.0 / 9.0, 1.0 / 7.0, 1.0 / 5.0, 1.0 / 3.0, 1.0};
  double pl = LC[0]; for (int i = 1; i < 7; i++) pl = std::fma(pl, f2, LC[i]);
  double lnm = 2.0 * f * pl;
  double lnx = std::fma((double)e, 0.69314718055994528623, lnm);
  double y = lnx * (1.0 / 2.4);
  double n = std::rint(y * 1.44269504088896338700);
  double g = std::fma(-n, 0.69314718055994528623, y);
  static const double EC[12] = {1.0 / 39916800.0, 1.0 / 3628800.0, 1.0 / 362880.0, 1.0 / 40320.0, 1.0 / 5040.0, 1.0 / 720.0,
                                1.0 / 120.0, 1.0 / 24.0, 1.0 / 6.0, 0.5, 1.0, 1.0};
  double pe = EC[0]; for (int i = 1; i < 12; i++) pe = std::fma(pe, g, EC[i]);
  uint64_t sb = (uint64_t)((int64_t)n + 1023) << 52;
  double scale; std::memcpy(&scale, &sb, 8);
  return (float)(pe * scale);
}

// drawScreen's scalar path, main.d:82-95: saturate -> sRGB OETF -> cast(ubyte)(x*255) (truncation).  A NaN channel
// (the cast is undefined for it) is written as 0.
void orc_tonemap(const Pixel* pixels, uint32_t count, uint8_t* rgb) {
  const float a = 0.055f, aPlusOne = 1 + 0.055f;
  for (uint32_t i = 0; i < count; i++)
    for (int k = 0; k < 3; k++) {
      float c = pixels[i].color[k];
      if (!(c == c)) { rgb[3 * i + k] = 0; continue; }
      c = c < 0.0f ? 0.0f : (c > 1.0f ? 1.0f : c);
      c = (c <= 0.0031308f) ? c * 12.92f : aPlusOne * powInv24S14(c) - a;
      rgb[3 * i + k] = (uint8_t)(c * 255.0f);
    }
}
// Probes of S13 / S14 and of the direction helpers, for cross-checks against libm and against the device.
void orc_set_libm_trig(int on) { g_libmTrig = on != 0; }
void orc_sincos(const float* x, uint32_t n, float* s, float* c) { for (uint32_t i = 0; i < n; i++) sincosS13(x[i], s[i], c[i]); }
void orc_pow_inv24(const float* x, uint32_t n, float* y) { for (uint32_t i = 0; i < n; i++) y[i] = powInv24S14(x[i]); }
// angleToDirection(phi, psi, normal) (rendering.d:316-330); a zero normal selects angleToLocalDirection (:309-314).
// in = n x (phi, psi, nx, ny, nz), out = n x 3
void orc_angle_directions(const float* in5, uint32_t n, float* out3) {
  for (uint32_t i = 0; i < n; i++) {
    const float* p = in5 + 5 * i; V3 nn = v3(p[2], p[3], p[4]);
    V3 d = (nn.x == 0.0f && nn.y == 0.0f && nn.z == 0.0f) ? angleToLocalDirection(p[0], p[1]) : angleToDirection(p[0], p[1], nn);
    out3[3 * i] = d.x; out3[3 * i + 1] = d.y; out3[3 * i + 2] = d.z;
  }
}

// Traversal-cost model for the roofline: walks the PRODUCT's flattened BVH (nodes8 = {min.xyz, ref, max.xyz, 0},
// ref = count << 28 | index, sibling pairs adjacent; triOrder = original triangle id per leaf slot) with the same
// order as the kernels -- both children of a pair tested per step, near child first, far child stacked, boxes
// culled against the closest accepted hit -- and counts box tests and triangle tests.  anyHit: 0 closest hit; 1 first
// hit, near child first; 2 first hit, children in storage order (what the kernels do when the scene is in LDS).  The kernels use the
// hardware reciprocal for the slab test, this uses 1/x, so counts agree to a small fraction of a percent.
// out = {nodes tested, triangles tested}; triId (optional) receives the accepted triangle (-1 = none).
void orc_bvh_count(void* h, const float* nodes8, uint32_t rootRef, const uint32_t* triOrder, const float* rays6,
                   uint32_t count, int anyHit, uint64_t* out, int32_t* triId) {
  Scene* s = (Scene*)h; float eps = s->cfg.floatEpsilon;
  uint64_t cN = 0, cT = 0;
  for (uint32_t ri = 0; ri < count; ri++) {
    const float* p = rays6 + 6 * (size_t)ri; Ray r{v3(p[0], p[1], p[2]), v3(p[3], p[4], p[5])};
    float d[3] = {r.d.x, r.d.y, r.d.z}, o[3] = {r.o.x, r.o.y, r.o.z}, inv[3], a0[3], a1[3];
    float om = fmaxf(fmaxf(fabsf(o[0]), fabsf(o[1])), fabsf(o[2]));
    for (int k = 0; k < 3; k++) {      // the kernels' addends: -o/d moved apart by 4 ulp of max|o| / |d| towards "enters earlier, leaves later"
      float dk = fabsf(d[k]) < 1e-20f ? copysignf(1e-20f, d[k]) : d[k]; inv[k] = 1.0f / dk;
      float oi = -o[k] * inv[k], s = om * fabsf(inv[k]) * 2.4e-7f;
      a0[k] = inv[k] > 0.0f ? oi - s : oi + s; a1[k] = inv[k] > 0.0f ? oi + s : oi - s;
    }
    float bt = FLT_MAX; int btri = -1; uint32_t stack[64]; int sp = 0; uint32_t ref = rootRef; bool live = true, done = false;
    auto slab = [&](const float* n, float& tn) {
      float lo[3], hi[3];
      for (int k = 0; k < 3; k++) { float a = std::fmaf(n[k], inv[k], a0[k]), b = std::fmaf(n[4 + k], inv[k], a1[k]); lo[k] = fminf(a, b); hi[k] = fmaxf(a, b); }
      tn = fmaxf(fmaxf(lo[0], lo[1]), fmaxf(lo[2], 0.0f));
      float tf = fminf(fminf(hi[0], hi[1]), hi[2]) * 1.0000005f;
      if (anyHit) return !(tn > tf);
      return tn <= fminf(tf, bt);
    };
    while (live && !done) {
      while ((ref >> 28) == 0) {
        const float* n0 = nodes8 + (size_t)ref * 16; const float* n1 = n0 + 8; cN += 2;
        float tn0, tn1; bool h0 = slab(n0, tn0), h1 = slab(n1, tn1);
        uint32_t r0, r1; std::memcpy(&r0, n0 + 3, 4); std::memcpy(&r1, n1 + 3, 4);
        if (h0 && h1) { bool sw = anyHit != 2 && tn1 < tn0; stack[sp++] = sw ? r0 : r1; ref = sw ? r1 : r0; }
        else if (h0 || h1) ref = h0 ? r0 : r1;
        else { if (sp == 0) { live = false; break; } ref = stack[--sp]; }
      }
      if (!live) break;
      uint32_t cnt = ref >> 28, first = ref & ((1u << 28) - 1);
      for (uint32_t i = 0; i < cnt && !done; i++) {
        uint32_t orig = triOrder[first + i]; cT++;
        if (orig == 0xFFFFFFFFu) continue;
        float pos = -1.0f, u = 0, v = 0, det;
        if (triIntersects(s->tris[orig], r, eps, pos, u, v, det)) {
          if (anyHit) { if (pos >= eps) { btri = (int)orig; done = true; } }
          else if ((pos < bt || (pos == bt && (int)orig < btri)) && pos >= eps && det < 0.0f) { bt = pos; btri = (int)orig; }
        }
      }
      if (sp == 0) break;
      ref = stack[--sp];
    }
    if (triId) triId[ri] = btri;
  }
  out[0] = cN; out[1] = cT;
}

}  // extern "C"
