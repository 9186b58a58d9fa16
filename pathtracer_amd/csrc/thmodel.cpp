// .thModel reader (modelloader.d:125-704).  The chunk framing itself lives in thBase.Chunkfile, which is
// not part of the reference checkout; it was recovered from the three model files the reference ships:
//   Chunk := u8 nameLen, char name[nameLen], u32 payloadLen, payload
//   File  := Chunk("thModel"){ u32 version, u32 debugFlag, sizeinfo textures materials meshes nodes }
#include "thmodel.h"
#include <cstdio>
#include <cstring>

namespace pt {
namespace {

struct Cursor {
  const uint8_t* d; size_t p, end; std::string name;
  void need(size_t n) const { if (n > end - p) throw IoError("thModel: unexpected end of chunk '" + name + "'"); }
  // A count read from the file is only believed when the chunk still holds that many records of at least minBytes
  // each: nothing is allocated for a count the payload cannot back.
  uint32_t count(size_t minBytes) { uint32_t n = u32(); need((size_t)n * minBytes); return n; }
  uint8_t u8() { need(1); return d[p++]; }
  uint32_t u32() { need(4); uint32_t v; std::memcpy(&v, d + p, 4); p += 4; return v; }
  void raw(void* dst, size_t n) { need(n); std::memcpy(dst, d + p, n); p += n; }
  std::string str(size_t n) { need(n); std::string s((const char*)d + p, n); p += n; return s; }
  Cursor chunk() {
    uint8_t n = u8(); std::string nm = str(n); uint32_t len = u32(); need(len);
    Cursor c{d, p, p + len, nm}; p += len; return c;
  }
  Cursor chunk(const char* expect) {
    Cursor c = chunk();
    if (c.name != expect) throw IoError(std::string("Expected '") + expect + "' chunk but got '" + c.name + "'");
    return c;
  }
  bool done() const { return p == end; }
};

}  // namespace

void LoadFile(const std::string& path, ModelData& out) {
  FILE* f = std::fopen(path.c_str(), "rb");
  if (!f) throw IoError("File '" + path + "' does not exist");                        // modelloader.d:132-135
  std::vector<uint8_t> buf;
  std::fseek(f, 0, SEEK_END); long sz = std::ftell(f); std::fseek(f, 0, SEEK_SET);
  buf.resize(sz > 0 ? (size_t)sz : 0);
  size_t got = buf.empty() ? 0 : std::fread(buf.data(), 1, buf.size(), f);
  std::fclose(f);
  if (got != buf.size() || buf.size() < 2) throw IoError("File '" + path + "' is not a thModel format");

  Cursor top{buf.data(), 0, buf.size(), "file"};
  uint8_t n0 = buf[0];
  if ((size_t)n0 + 1 > buf.size() || std::string((const char*)buf.data() + 1, n0) != "thModel")
    throw IoError("File '" + path + "' is not a thModel format");                     // modelloader.d:139-142
  Cursor file = top.chunk();
  out = ModelData();
  out.version = file.u32();
  file.u32();                                                                          // Chunkfile debug flag
  if (out.version > 2) throw IoError("Model '" + path + "' does have old format, please reexport");   // modelloader.d:144-147

  {  // sizeinfo (modelloader.d:196-327): only needed by the reference to size its allocator; validated, not used
    Cursor s = file.chunk();
    if (s.name != "sizeinfo") throw IoError("Expected sizeinfo chunk, got '" + s.name + "' chunk in file '" + path + "'");
    s.u32(); s.u32(); if (out.version >= 2) s.u32();
    s.u32(); uint32_t nm = s.u32();
    for (uint32_t i = 0; i < nm; i++) {
      s.u32(); uint32_t flags = s.u32();
      for (uint32_t b = PVD_TexCoord0; b <= PVD_TexCoord3; b <<= 1) if (flags & b) s.u8();
      s.u32();
    }
    for (int i = 0; i < 5; i++) s.u32();
  }
  {  // textures (modelloader.d:338-374)
    Cursor t = file.chunk("textures");
    uint32_t n = t.count(4); out.textures.resize(n);
    for (auto& s : out.textures) { uint32_t len = t.u32(); s = t.str(len); }
  }
  {  // materials (modelloader.d:377-445)
    Cursor m = file.chunk("materials");
    uint32_t n = m.count(8); out.materials.resize(n);                       // an empty "mat" chunk is 8 bytes
    for (auto& mat : out.materials) {
      Cursor c = m.chunk("mat");
      if (out.version >= 2) { uint32_t len = c.u32(); mat.name = c.str(len); }
      uint32_t nt = c.count(5); mat.textures.resize(nt);
      for (auto& tr : mat.textures) {
        tr.textureIndex = c.u32(); tr.semantic = c.u8();
        if (tr.textureIndex >= out.textures.size()) throw IoError("thModel: texture index out of range");
      }
    }
  }
  {  // meshes (modelloader.d:448-633)
    Cursor ms = file.chunk("meshes");
    uint32_t n = ms.count(9); out.meshes.resize(n);                         // an empty "mesh" chunk is 9 bytes
    for (auto& mesh : out.meshes) {
      Cursor c = ms.chunk("mesh");
      mesh.materialIndex = c.u32();
      c.raw(mesh.bboxMin, 12); c.raw(mesh.bboxMax, 12);
      uint32_t nv = c.u32();
      Cursor v = c.chunk("vertices");
      v.need((size_t)nv * 12);
      mesh.vertices.resize((size_t)nv * 3); v.raw(mesh.vertices.data(), (size_t)nv * 12);
      bool faces = false;
      while (!faces) {
        Cursor sub = c.chunk();
        if (sub.name == "normals" || sub.name == "tangents" || sub.name == "bitangents") {
          sub.need((size_t)nv * 6);
          std::vector<int16_t> q((size_t)nv * 3); sub.raw(q.data(), q.size() * 2);
          if (sub.name == "normals") {
            mesh.normals.resize(q.size());
            for (size_t i = 0; i < q.size(); i++) mesh.normals[i] = (float)q[i] / (float)32767;   // UncompressFloat
          }
        } else if (sub.name == "texcoords") {
          uint8_t nsets = sub.u8();
          for (uint8_t i = 0; i < nsets; i++) {
            uint8_t comps = sub.u8();
            if (comps != 2) throw IoError("Currently only 2 component texture coordinates are supported got " + std::to_string(comps));
            sub.need((size_t)nv * 8);
            std::vector<float> tc((size_t)nv * 2); sub.raw(tc.data(), tc.size() * 4);
            if (i < 4) mesh.texcoords[i].swap(tc);
          }
        } else if (sub.name == "faces") {
          uint32_t nf = sub.count(nv > 65535 ? 12 : 6); mesh.faces.resize((size_t)nf * 3);
          if (nv > 65535) sub.raw(mesh.faces.data(), (size_t)nf * 12);                           // modelloader.d:602-605
          else { std::vector<uint16_t> s16((size_t)nf * 3); sub.raw(s16.data(), s16.size() * 2);
                 for (size_t i = 0; i < s16.size(); i++) mesh.faces[i] = s16[i]; }
          for (uint32_t idx : mesh.faces) if (idx >= nv) throw IoError("thModel: face index out of range");
          faces = true;
        } else {
          throw IoError("Unexpected chunk '" + sub.name + "'");                                  // modelloader.d:619
        }
      }
      if (mesh.materialIndex >= out.materials.size()) throw IoError("thModel: material index out of range");
    }
  }
  {  // nodes (modelloader.d:636-692)
    Cursor nd = file.chunk("nodes");
    uint32_t n = nd.count(80); out.nodes.resize(n);                        // name length, transform, parent, two counts
    for (auto& node : out.nodes) {
      uint32_t len = nd.u32(); node.name = nd.str(len);
      nd.raw(node.transform, 64);
      uint32_t parent = nd.u32(); node.parent = parent == 0xFFFFFFFFu ? -1 : (int32_t)parent;
      uint32_t nmesh = nd.count(4); node.meshes.resize(nmesh);
      for (auto& m : node.meshes) { m = nd.u32(); if (m >= out.meshes.size()) throw IoError("thModel: mesh index out of range"); }
      uint32_t nch = nd.count(4); node.children.resize(nch);
      for (auto& ch : node.children) { ch = nd.u32(); if (ch >= n) throw IoError("thModel: child index out of range"); }
      if (node.parent >= (int32_t)n) throw IoError("thModel: parent index out of range");
    }
    if (out.nodes.empty()) throw IoError("thModel: no nodes");
  }
}

}  // namespace pt
