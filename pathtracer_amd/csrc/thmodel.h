// .thModel reader: the host-side data the reference gets from ModelLoader.LoadFile
// (modelloader.d:125-704).  Names follow the reference's structs (modelloader.d:34-85).
#pragma once
#include <stdint.h>
#include <string>
#include <vector>
#include <stdexcept>

namespace pt {

struct IoError : std::runtime_error { using std::runtime_error::runtime_error; };

// modeltypes.d:28-38
enum PerVertexData : uint32_t {
  PVD_Position = 0x01, PVD_Normal = 0x02, PVD_Tangent = 0x04, PVD_Bitangent = 0x08,
  PVD_TexCoord0 = 0x10, PVD_TexCoord1 = 0x20, PVD_TexCoord2 = 0x40, PVD_TexCoord3 = 0x80
};

struct TextureReference { uint8_t semantic; uint32_t textureIndex; };   // modelloader.d:72-76, modeltypes.d:6-20
struct MaterialData { std::string name; std::vector<TextureReference> textures; };
struct MeshData {
  uint32_t materialIndex = 0;
  float bboxMin[3], bboxMax[3];
  std::vector<uint32_t> faces;       // 3 per face
  std::vector<float> vertices;       // 3 per vertex
  std::vector<float> normals;        // 3 per vertex, int16/32767 (modelloader.d:492-497)
  std::vector<float> texcoords[4];   // 2 per vertex
};
struct NodeDrawData {
  std::string name;
  float transform[16];
  int32_t parent = -1;               // -1 == none (0xFFFFFFFF in the file)
  std::vector<uint32_t> meshes, children;
};
struct ModelData {
  uint32_t version = 0;
  std::vector<std::string> textures;
  std::vector<MaterialData> materials;
  std::vector<MeshData> meshes;
  std::vector<NodeDrawData> nodes;   // root = nodes[0] (modelloader.d:692)
};

// Throws IoError with the reference's messages where it has one.
void LoadFile(const std::string& path, ModelData& out);

}  // namespace pt
