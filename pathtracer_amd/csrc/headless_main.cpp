// Headless stand-in for the reference's main() (main.d:142-345): the same loop shape -- loadScene, optional
// resume from a project file, one whole-frame pass per iteration with "pass %d done in %f seconds",
// tone map, total time, project file dump -- driving libptgpu through its C ABI instead of SDL + the D
// task queue.  The image goes to a binary PPM instead of an SDL surface.
#include "../../include/ptgpu.h"
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

static double now() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

static int usage() {
  std::fprintf(stderr,
               "usage: ptgpu_headless MODEL.thModel [--size W H] [--passes N] [--spp-per-pass K] [--from x y z] [--to x y z]\n"
               "         [--fov deg] [--sky] [--no-lights] [--mode 0|1] [--seed s] [--device d] [--out image.ppm]\n"
               "         [--project file]   (raw Pixel[] as main.d:227-234 reads and main.d:333-340 writes)\n"
               "         [--display sync|overlap]  tone map + copy the frame to the host after every pass, as drawScreen is\n"
               "                                   called per pass (main.d:275); overlap: beside the next pass's rendering\n");
  return 2;
}

#define CHECK(call)                                                                       \
  do {                                                                                    \
    int rc_ = (call);                                                                     \
    if (rc_ != PT_OK) { std::fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, ptgpu_last_error()); return 1; } \
  } while (0)

int main(int argc, char** argv) {
  if (argc < 2) return usage();
  std::string model = argv[1], out, project, display;
  uint32_t W = 320, H = 240, passes = 16, sppPerPass = 1;        // rendering.d:19-20; one sample per pass (main.d:273-287)
  float from[3] = {-1, 26.5f, 10}, to[3] = {0, 0, 9}, up[3] = {0, 0, 1}, fov = 30.0f;   // rendering.d:76,81
  PtConfig cfg; std::memset(&cfg, 0, sizeof(cfg));
  cfg.hasLights = 1;                                             // rendering.d:54-55
  for (int i = 2; i < argc; i++) {
    std::string a = argv[i];
    auto need = [&](int n) { if (i + n >= argc) { usage(); std::exit(2); } };
    if (a == "--size") { need(2); W = std::atoi(argv[++i]); H = std::atoi(argv[++i]); }
    else if (a == "--passes") { need(1); passes = std::atoi(argv[++i]); }
    else if (a == "--spp-per-pass") { need(1); sppPerPass = std::atoi(argv[++i]); }
    else if (a == "--from") { need(3); for (int k = 0; k < 3; k++) from[k] = (float)std::atof(argv[++i]); }
    else if (a == "--to") { need(3); for (int k = 0; k < 3; k++) to[k] = (float)std::atof(argv[++i]); }
    else if (a == "--fov") { need(1); fov = (float)std::atof(argv[++i]); }
    else if (a == "--sky") cfg.hasSky = 1;
    else if (a == "--no-lights") cfg.hasLights = 0;
    else if (a == "--mode") { need(1); cfg.mode = std::atoi(argv[++i]); }
    else if (a == "--seed") { need(1); cfg.seed = (uint32_t)std::strtoul(argv[++i], nullptr, 10); }
    else if (a == "--device") { need(1); cfg.device = std::atoi(argv[++i]); }
    else if (a == "--out") { need(1); out = argv[++i]; }
    else if (a == "--project") { need(1); project = argv[++i]; }
    else if (a == "--display") { need(1); display = argv[++i]; if (display != "sync" && display != "overlap") return usage(); }
    else return usage();
  }
  cfg.width = W; cfg.height = H;
  // loadScene(), rendering.d:72-74
  float inv = 1.0f / std::sqrt(11.0f);
  cfg.sunDir[0] = 1 * inv; cfg.sunDir[1] = 1 * inv; cfg.sunDir[2] = 3 * inv;
  cfg.sunRadiance[0] = 1.0f; cfg.sunRadiance[1] = 0.984f; cfg.sunRadiance[2] = 0.8f;
  cfg.skyRadiance[0] = 0.682f * 0.1f; cfg.skyRadiance[1] = 0.977f * 0.1f; cfg.skyRadiance[2] = 1.0f * 0.1f;

  PtHandle* h = nullptr;
  CHECK(ptgpu_create(&cfg, &h));
  double t0 = now();
  CHECK(ptgpu_load_thmodel(h, model.c_str(), 0, nullptr));
  CHECK(ptgpu_camera_look_at(h, fov, (float)H / (float)W, from, to, up, nullptr));
  PtStats st; CHECK(ptgpu_get_stats(h, &st));
  std::printf("Scene has %u triangles\n", st.numTriangles);                       // scene.d:237
  for (uint32_t g = 0; g < st.numLightGroups; g++) {
    uint32_t cnt = 0; float area = 0; CHECK(ptgpu_get_light_group(h, g, nullptr, 0, &cnt, &area));
    std::printf("Light %u has %u light triangles\n", g, cnt);                     // rendering.d:159
  }
  std::printf("Building tree took %f seconds\n", now() - t0);                     // scene.d:635

  uint32_t npix = W * H;
  std::vector<PtPixel> pixels(npix);
  if (!project.empty()) {                                                         // main.d:227-234
    if (FILE* f = std::fopen(project.c_str(), "rb")) {
      size_t got = std::fread(pixels.data(), sizeof(PtPixel), npix, f); std::fclose(f);
      if (got == npix) { CHECK(ptgpu_write_pixels(h, 0, npix, pixels.data())); std::printf("project file loaded\n"); }
    }
  }
  // the frame as drawScreen would show it (main.d:31-99), in page-locked memory so that the device writes it directly
  uint8_t* screen[2] = {nullptr, nullptr};
  if (!display.empty())
    for (auto& s : screen) { void* p = nullptr; CHECK(ptgpu_host_alloc(h, (uint64_t)npix * 3, &p)); s = (uint8_t*)p; }
  std::vector<double> passSeconds;
  double startRendering = now(), startPass = startRendering;
  bool pendingDraw = false;
  for (uint32_t progress = 1; progress <= passes; progress++) {                    // main.d:239-288
    CHECK(ptgpu_render(h, 0, npix, sppPerPass));
    if (display == "sync") {                                                       // render, then draw (main.d:273-287)
      CHECK(ptgpu_tonemap(h, 0, npix, screen[0]));
    } else if (display == "overlap") {                                             // draw pass k beside rendering pass k + 1
      if (pendingDraw) CHECK(ptgpu_tonemap_end(h));
      CHECK(ptgpu_tonemap_begin(h, 0, npix, screen[progress & 1]));
      pendingDraw = true;
    } else {
      CHECK(ptgpu_synchronize(h));
    }
    double endPass = now();
    std::printf("pass %u done in %f seconds\n", progress, endPass - startPass);   // main.d:279
    passSeconds.push_back(endPass - startPass);
    startPass = endPass;
  }
  if (pendingDraw) CHECK(ptgpu_tonemap_end(h));
  CHECK(ptgpu_synchronize(h));
  std::printf("Rendering took %f seconds\n", now() - startRendering);             // main.d:331
  if (passSeconds.size() > 2) {                 // steady state: the first pass builds masks and allocates
    std::vector<double> s(passSeconds.begin() + 1, passSeconds.end());
    std::sort(s.begin(), s.end());
    std::printf("pass_ms median %.4f min %.4f (%ux%u, %u spp per pass, display %s)\n", 1e3 * s[s.size() / 2], 1e3 * s[0], W, H,
                sppPerPass, display.empty() ? "none" : display.c_str());
  }
  for (auto& s : screen) CHECK(ptgpu_host_free(h, s));
  CHECK(ptgpu_get_stats(h, &st));
  uint64_t rays = st.raysPrimary + st.raysLight + st.raysIndirect + st.raysOcclusion;
  std::printf("%llu rays, %.1f Mrays/s (device time %.3f ms)\n", (unsigned long long)rays,
              st.renderMs > 0 ? rays / (st.renderMs * 1e-3) / 1e6 : 0.0, st.renderMs);

  if (!out.empty()) {                                                             // drawScreen, main.d:31-99
    std::vector<uint8_t> rgb((size_t)npix * 3);
    CHECK(ptgpu_tonemap(h, 0, npix, rgb.data()));
    FILE* f = std::fopen(out.c_str(), "wb");
    if (!f) { std::fprintf(stderr, "cannot write %s\n", out.c_str()); return 1; }
    std::fprintf(f, "P6\n%u %u\n255\n", W, H);
    std::fwrite(rgb.data(), 1, rgb.size(), f); std::fclose(f);
  }
  if (!project.empty()) {                                                         // main.d:333-340
    CHECK(ptgpu_read_pixels(h, 0, npix, pixels.data()));
    if (FILE* f = std::fopen(project.c_str(), "wb")) { std::fwrite(pixels.data(), sizeof(PtPixel), npix, f); std::fclose(f); }
    std::printf("written project file\n");
  }
  CHECK(ptgpu_destroy(h));
  return 0;
}
