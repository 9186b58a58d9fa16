// Host-side scene preparation: what Scene's constructor and loadScene() do before the first ray
// (scene.d:193-321, rendering.d:116-161, 195-275).
#pragma once
#include "thmodel.h"
#include "pt_math.h"
#include "../../include/ptgpu.h"
#include <vector>
#include <string>

namespace pt {

struct FlatScene {                       // Scene.m_triangles / m_data / m_materials as plain arrays
  std::vector<float> v;                  // T x 9  (v0, v1, v2)
  std::vector<float> n;                  // T x 9  vertex normals
  std::vector<float> tex;                // T x 6  TriangleData.tex (scene.d:305-307); empty = all zero
  std::vector<uint32_t> matIdx;          // T
  std::vector<PtMaterial> mats;          // M
  std::vector<std::string> matNames;     // M
  size_t numTriangles() const { return matIdx.size(); }
};

struct LightTriangle { float v0[3], e1[3], e2[3], area, probability; };   // rendering.d:40-46
static_assert(sizeof(LightTriangle) == 44, "LightTriangle is 11 packed floats");
struct LightTable {                      // g_lightTriangles / g_totalLightSourceArea, rendering.d:47-48
  std::vector<LightTriangle> tris;       // groups back to back
  std::vector<uint32_t> first, count;    // per group
  std::vector<float> totalArea;          // per group
  size_t numGroups() const { return first.size(); }
};

// name -> Material tables: 0 = fillMaterial (rendering.d:247-275), 1 = fillMaterialCitymap (rendering.d:195-244)
PtMaterial materialFromName(int table, const std::string& name);

// row-major 4x4, row-vector convention (v' = v * M)
void mat4Mul(const float* a, const float* b, float* out);
bool mat4Inverse(const float* m, float* out);

// Scene's flattening of the model's node graph, scene.d:209-321.  modelMatrix == nullptr selects
// inverse(root.transform), i.e. raw mesh space with only non-root node transforms applied.
void flattenModel(const ModelData& model, int materialTable, const float* modelMatrix, FlatScene& out);

// loadScene()'s light discovery, rendering.d:116-161: one group per emissive material.
void findLights(const FlatScene& scene, LightTable& out);

// Camera(fov, aspect) + setTransform(from, to, up), camera.d:16-29
Camera cameraLookAt(float fovDegrees, float aspect, V3 from, V3 to, V3 up);

// Conservative coverage mask of the view rays (getViewRay, rendering.d:278-285): bit p of `bits` is set when some
// triangle's projection reaches the square of pixel p's sample positions (half a pixel of jitter each way plus a
// margin; tested edge by edge inside the projection's bounding rectangle).  A view ray of a pixel whose bit is clear cannot hit anything.  Bit p of `constantSky` is set when
// in addition no sample position of the pixel sees sampleSky's sun disc (rendering.d:332-339): every sample of that
// pixel is the sky constant.  Returns false (outputs unspecified) when the masks cannot be built: a triangle is not
// strictly in front of the camera, or the camera basis is singular.
// Largest finite |coordinate| of the scene (the masks' notion of its scale).
double sceneScale(const FlatScene& scene);
bool buildCoverageMask(const FlatScene& scene, const Camera& cam, uint32_t W, uint32_t H, std::vector<uint32_t>& bits,
                       std::vector<uint32_t>& constantSky);

}  // namespace pt
