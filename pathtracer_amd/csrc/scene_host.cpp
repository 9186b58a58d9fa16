#include "scene_host.h"
#include "mask_math.h"
#include <algorithm>
#include <cmath>
#include <cstring>
#include <functional>

namespace pt {

PtMaterial materialFromName(int table, const std::string& name) {
  PtMaterial m; m.emissive = 0.0f; m.color[0] = m.color[1] = m.color[2] = 0.7f;
  auto set = [&](float r, float g, float b, float e) { m.color[0] = r; m.color[1] = g; m.color[2] = b; m.emissive = e; };
  if (table == 1) {                                                     // fillMaterialCitymap, rendering.d:195-244
    if (name == "Material #10088/City_HotelLight_mat") set(0.7f, 0.7f, 0.7f, 10.0f);
    else if (name == "Material #10088/GreenLight") set(0.0f, 0.7f, 0.0f, 10.0f);
    else if (name == "Material #10088/City_CinemaDisplay_mat") set(0.7f, 0.7f, 0.7f, 20.0f);
    else if (name == "Material #10088/City_CinemaEntrancelamp_mat") set(0.8f, 0.8f, 0.2f, 160.0f);
    else if (name == "Material #10088/City_FC_Light_mat") set(0.7f, 0.0f, 0.0f, 40.0f);
    else if (name == "Material #10088/City_FC_Logo_mat") set(0.7f, 0.0f, 0.0f, 30.0f);
  } else {                                                              // fillMaterial, rendering.d:247-275
    if (name == "Light") m.emissive = 0.5f;
    else if (name == "Red") set(0.7f, 0.0f, 0.0f, 0.0f);
    else if (name == "Green") set(0.0f, 0.7f, 0.0f, 0.0f);
    else if (name == "Blue") set(0.0f, 0.0f, 0.7f, 0.0f);
  }
  return m;
}

void mat4Mul(const float* a, const float* b, float* out) {
  float r[16];
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) {
      float acc = a[4 * i] * b[j];
      for (int k = 1; k < 4; k++) acc = acc + a[4 * i + k] * b[4 * k + j];
      r[4 * i + j] = acc;
    }
  std::memcpy(out, r, sizeof(r));
}

bool mat4Inverse(const float* m, float* out) {                         // Gauss-Jordan in double, rounded once
  double a[4][8];
  for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) { a[i][j] = m[4 * i + j]; a[i][4 + j] = (i == j) ? 1.0 : 0.0; }
  for (int c = 0; c < 4; c++) {
    int piv = c;
    for (int r = c + 1; r < 4; r++) if (std::fabs(a[r][c]) > std::fabs(a[piv][c])) piv = r;
    if (a[piv][c] == 0.0) return false;
    if (piv != c) for (int j = 0; j < 8; j++) std::swap(a[piv][j], a[c][j]);
    double d = a[c][c];
    for (int j = 0; j < 8; j++) a[c][j] /= d;
    for (int r = 0; r < 4; r++) if (r != c) { double f = a[r][c]; if (f != 0.0) for (int j = 0; j < 8; j++) a[r][j] -= f * a[c][j]; }
  }
  for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) out[4 * i + j] = (float)a[i][4 + j];
  return true;
}

void flattenModel(const ModelData& model, int materialTable, const float* modelMatrix, FlatScene& out) {
  out = FlatScene();
  const NodeDrawData& root = model.nodes[0];
  float mm[16];
  if (modelMatrix) std::memcpy(mm, modelMatrix, sizeof(mm));
  else if (!mat4Inverse(root.transform, mm)) throw IoError("root node transform is singular");

  // findLeafs, scene.d:213-225: depth first from node 0, children in file order.  The reference recurses and trusts
  // the file; a node reached twice (a cycle, or a node with two parents) would send it into unbounded recursion or
  // duplicate geometry, so such a file is refused.
  std::vector<uint32_t> leafs;
  {
    std::vector<uint8_t> seen(model.nodes.size(), 0);
    std::vector<std::pair<uint32_t, size_t>> stack;                     // (node, next child position)
    stack.push_back({0u, 0});
    seen[0] = 1;
    if (!model.nodes[0].meshes.empty()) leafs.push_back(0);
    while (!stack.empty()) {
      auto& top = stack.back();
      const NodeDrawData& nd = model.nodes[top.first];
      if (top.second == nd.children.size()) { stack.pop_back(); continue; }
      uint32_t c = nd.children[top.second++];
      if (seen[c]) throw IoError("thModel: node " + std::to_string(c) + " is reached twice (the node graph is not a tree)");
      seen[c] = 1;
      if (!model.nodes[c].meshes.empty()) leafs.push_back(c);
      stack.push_back({c, 0});
    }
  }
  if (leafs.empty()) throw IoError("no leaf nodes found");              // scene.d:226

  for (const MaterialData& md : model.materials) {                      // scene.d:239-245
    out.matNames.push_back(md.name);
    out.mats.push_back(materialFromName(materialTable, md.name));
  }
  for (uint32_t leaf : leafs) {
    float tr[16];
    mat4Mul(root.transform, mm, tr);                                    // scene.d:257
    int32_t cur = (int32_t)leaf;
    size_t steps = 0;
    while (cur >= 0 && cur != 0) {                                      // scene.d:258-262
      if (++steps > model.nodes.size()) throw IoError("thModel: parent links of node " + std::to_string(leaf) + " form a cycle");
      mat4Mul(model.nodes[cur].transform, tr, tr);
      cur = model.nodes[cur].parent;
    }
    // NormalMatrix(): (M3^-1)^T = cofactor(M3) / det(M3) of the upper-left 3x3 (scene.d:264)
    float a[3][3], c[3][3], N[3][3];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) a[i][j] = tr[4 * i + j];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
      int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
      c[i][j] = a[i1][j1] * a[i2][j2] - a[i1][j2] * a[i2][j1];
    }
    float det = (a[0][0] * c[0][0] + a[0][1] * c[0][1]) + a[0][2] * c[0][2];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) N[i][j] = c[i][j] / det;

    for (uint32_t mi : model.nodes[leaf].meshes) {
      const MeshData& mesh = model.meshes[mi];
      size_t nv = mesh.vertices.size() / 3;
      if (mesh.normals.size() != mesh.vertices.size()) throw IoError("mesh has no per-vertex normals");
      std::vector<float> vt(nv * 3), nt(nv * 3);
      for (size_t i = 0; i < nv; i++) {                                 // transformPosition, scene.d:280
        float x = mesh.vertices[3 * i], y = mesh.vertices[3 * i + 1], z = mesh.vertices[3 * i + 2];
        for (int j = 0; j < 3; j++) vt[3 * i + j] = ((x * tr[j] + y * tr[4 + j]) + z * tr[8 + j]) + tr[12 + j];
        float nx = mesh.normals[3 * i], ny = mesh.normals[3 * i + 1], nz = mesh.normals[3 * i + 2];
        V3 t = v3((nx * N[0][0] + ny * N[1][0]) + nz * N[2][0], (nx * N[0][1] + ny * N[1][1]) + nz * N[2][1],
                  (nx * N[0][2] + ny * N[1][2]) + nz * N[2][2]);
        t = normalized(t);                                              // scene.d:288
        nt[3 * i] = t.x; nt[3 * i + 1] = t.y; nt[3 * i + 2] = t.z;
      }
      size_t nf = mesh.faces.size() / 3;
      for (size_t f = 0; f < nf; f++) {                                 // scene.d:291-307
        for (int k = 0; k < 3; k++) {
          uint32_t idx = mesh.faces[3 * f + k];
          for (int j = 0; j < 3; j++) { out.v.push_back(vt[3 * idx + j]); }
        }
        for (int k = 0; k < 3; k++) {
          uint32_t idx = mesh.faces[3 * f + k];
          for (int j = 0; j < 3; j++) { out.n.push_back(nt[3 * idx + j]); }
        }
        out.matIdx.push_back(mesh.materialIndex);
      }
      // TriangleData.tex, scene.d:305-307.  The reference indexes texcoord set 1; a mesh that has fewer sets
      // (every model shipped with it) would fail its bounds check there, so the last set present is taken
      // instead, and (0, 0) when the mesh has none.
      const std::vector<float>* tc = nullptr;
      if (mesh.texcoords[1].size() == nv * 2) tc = &mesh.texcoords[1];
      else for (int s = 3; s >= 0 && !tc; s--) if (mesh.texcoords[s].size() == nv * 2 && nv) tc = &mesh.texcoords[s];
      for (size_t f = 0; f < nf; f++)
        for (int k = 0; k < 3; k++) {
          uint32_t idx = mesh.faces[3 * f + k];
          out.tex.push_back(tc ? (*tc)[2 * idx] : 0.0f);
          out.tex.push_back(tc ? (*tc)[2 * idx + 1] : 0.0f);
        }
    }
  }
}

void findLights(const FlatScene& s, LightTable& out) {
  out = LightTable();
  std::vector<int> matToLight(s.mats.size(), -1);                       // rendering.d:117-126
  int nl = 0;
  for (size_t m = 0; m < s.mats.size(); m++) if (s.mats[m].emissive > 0.0f) matToLight[m] = nl++;
  std::vector<std::vector<LightTriangle>> groups(nl);
  out.totalArea.assign(nl, 0.0f);
  for (size_t i = 0; i < s.numTriangles(); i++) {                       // rendering.d:135-150
    uint32_t mi = s.matIdx[i];
    if (s.mats[mi].emissive > 0.0f) {
      const float* p = &s.v[9 * i];
      V3 v0 = v3(p[0], p[1], p[2]), v1 = v3(p[3], p[4], p[5]), v2 = v3(p[6], p[7], p[8]);
      V3 e1 = v1 - v0, e2 = v2 - v0;
      LightTriangle lt;
      lt.v0[0] = v0.x; lt.v0[1] = v0.y; lt.v0[2] = v0.z;
      lt.e1[0] = e1.x; lt.e1[1] = e1.y; lt.e1[2] = e1.z;
      lt.e2[0] = e2.x; lt.e2[1] = e2.y; lt.e2[2] = e2.z;
      lt.area = 0.5f * length(cross(e1, e2));                           // Triangle.area
      lt.probability = 0.0f;
      int g = matToLight[mi];
      groups[g].push_back(lt);
      out.totalArea[g] += lt.area;
    }
  }
  for (int g = 0; g < nl; g++) {                                        // rendering.d:153-160
    out.first.push_back((uint32_t)out.tris.size());
    out.count.push_back((uint32_t)groups[g].size());
    for (LightTriangle& lt : groups[g]) { lt.probability = lt.area / out.totalArea[g]; out.tris.push_back(lt); }
  }
}

Camera cameraLookAt(float fovDegrees, float aspect, V3 from, V3 to, V3 up) {
  Camera c;
  c.tanFov = (float)std::tan((double)(fovDegrees / 180.0f) * 3.14159265358979323846);   // camera.d:18
  c.aspect = aspect;
  c.pos = from;                                                                          // camera.d:24-28
  c.dir = normalized(to - from);
  c.right = normalized(cross(up, c.dir));
  c.up = normalized(cross(c.right, c.dir));
  return c;
}

double sceneScale(const FlatScene& scene) {
  double scale = 1e-30;
  for (float f : scene.v) if (std::isfinite(f)) scale = std::max(scale, (double)std::fabs(f));
  return scale;
}

bool buildCoverageMask(const FlatScene& scene, const Camera& cam, uint32_t W, uint32_t H, std::vector<uint32_t>& bits,
                       std::vector<uint32_t>& constantSky) {
  MaskCam mc;
  if (!maskCamSetup(cam, W, H, sceneScale(scene), mc)) return false;
  const size_t wordsPerRow = ((size_t)W + 31) / 32;           // rows are padded to whole words while marking
  std::vector<uint32_t> rows((size_t)H * wordsPerRow, 0u);
  size_t T = scene.numTriangles();
  for (size_t i = 0; i < T; i++) {
    MaskTri t;
    int st = maskTriSetup(mc, &scene.v[9 * i], t);
    if (st == MASK_TRI_BEHIND) return false;
    if (st != MASK_TRI_OK) continue;
    for (uint32_t r = t.ra; r <= t.rb; r++) {
      uint32_t a, b;
      if (!maskRowSpan(t, r, a, b)) continue;
      uint32_t wa = a >> 5, wb = b >> 5;
      uint32_t first = 0xFFFFFFFFu << (a & 31), last = 0xFFFFFFFFu >> (31 - (b & 31));
      uint32_t* row = &rows[(size_t)r * wordsPerRow];
      if (wa == wb) row[wa] |= first & last;
      else { row[wa] |= first; for (uint32_t w = wa + 1; w < wb; w++) row[w] = 0xFFFFFFFFu; row[wb] |= last; }
    }
  }
  // repack to one bit per pixel index (row * W + col)
  size_t npix = (size_t)W * H;
  bits.assign((npix + 31) / 32, 0u);
  constantSky.assign((npix + 31) / 32, 0u);
  for (uint32_t r = 0; r < H; r++)
    for (uint32_t c = 0; c < W; c++) {
      size_t px = (size_t)r * W + c;
      if ((rows[(size_t)r * wordsPerRow + (c >> 5)] >> (c & 31)) & 1u) { bits[px >> 5] |= 1u << (px & 31); continue; }
      if (maskConstantSky(mc, c, r)) constantSky[px >> 5] |= 1u << (px & 31);
    }
  return true;
}

}  // namespace pt
