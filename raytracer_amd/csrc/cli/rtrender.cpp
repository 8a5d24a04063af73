// rtrender — headless replacement for the reference's main.cpp + viewport (RayTracer/main.cpp:9-14,
// viewport.cpp:776-798): load a scene file, render it on one MI355X through the C-ABI, write the image
// the way StopRender does (RayTracer.cpp:60-90: outputs/rawImage.png, outputs/denoised.png).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "rtcore.h"
#include "rthost.h"

static void usage() {
  std::fprintf(stderr,
               "usage: rtrender scene.xml [--out raw.png] [--denoised out.png] [--zbuffer z.png] [--samples n.png] [--width W --height H]\n"
               "                [--spp N | --spp-min A --spp-max B]\n"
               "                [--bounces N] [--mc-bounces N] [--no-photons] [--no-jitter] [--seed S] [--device D] [--assets DIR]\n"
               "                [--extensions BITS]   1: real Phong / microfacet materials, 2: filtered texture lookups (ray differentials)\n");
}

int main(int argc, char** argv) {
  if (argc < 2) { usage(); return 2; }
  const char* scene_path = argv[1];
  std::string out = "outputs/rawImage.png", den, zimg, simg;
  const char* assets = nullptr;
  int device = 0;
  rt_render_params P;
  rt_render_params_default(&P);
  for (int i = 2; i < argc; i++) {
    auto arg = [&](const char* n) { return !std::strcmp(argv[i], n) && i + 1 < argc; };
    if (arg("--out")) out = argv[++i];
    else if (arg("--denoised")) den = argv[++i];
    else if (arg("--zbuffer")) zimg = argv[++i];
    else if (arg("--samples")) simg = argv[++i];
    else if (arg("--width")) P.width = std::atoi(argv[++i]);
    else if (arg("--height")) P.height = std::atoi(argv[++i]);
    else if (arg("--spp")) P.spp_min = P.spp_max = std::atoi(argv[++i]);
    else if (arg("--spp-min")) P.spp_min = std::atoi(argv[++i]);
    else if (arg("--spp-max")) P.spp_max = std::atoi(argv[++i]);
    else if (arg("--bounces")) P.max_bounces = std::atoi(argv[++i]);
    else if (arg("--mc-bounces")) P.mc_bounces = std::atoi(argv[++i]);
    else if (arg("--seed")) P.seed_lo = (uint32_t)std::strtoul(argv[++i], nullptr, 10);
    else if (arg("--device")) device = std::atoi(argv[++i]);
    else if (arg("--assets")) assets = argv[++i];
    else if (arg("--extensions")) P.extensions = (uint32_t)std::strtoul(argv[++i], nullptr, 10);   // RT_EXT_* bits
    else if (!std::strcmp(argv[i], "--no-photons")) P.photons = 0;
    else if (!std::strcmp(argv[i], "--no-jitter")) P.pixel_jitter = 0;
    else { usage(); return 2; }
  }
  rth_scene* scene = rth_scene_load_file(scene_path, assets);
  if (!scene) { std::fprintf(stderr, "ERROR: %s\n", rth_last_error()); return 1; }
  const rt_scene_desc* d = rth_scene_desc(scene);
  int W = P.width > 0 ? P.width : d->camera.width, H = P.height > 0 ? P.height : d->camera.height;
  rt_context* ctx = nullptr;
  if (rt_context_create(device, &ctx) != RT_OK) { std::fprintf(stderr, "ERROR: no usable GPU (device %d)\n", device); return 1; }
  if (rt_scene_commit(ctx, d) != RT_OK) { std::fprintf(stderr, "ERROR: %s\n", rt_last_error(ctx)); return 1; }
  std::vector<uint8_t> rgb((size_t)W * H * 3);
  std::vector<float> rad((size_t)W * H * 4);
  rt_framebuffer fb;
  std::memset(&fb, 0, sizeof fb);
  std::vector<float> depth((size_t)W * H), normal((size_t)W * H * 3), albedo((size_t)W * H * 3);
  std::vector<int32_t> nsamples((size_t)W * H);
  fb.rgb8 = rgb.data();
  fb.radiance = rad.data();
  fb.depth = depth.data();
  fb.normal = normal.data();
  fb.albedo = albedo.data();
  fb.sample_count = nsamples.data();
  auto t0 = std::chrono::steady_clock::now();
  int rc = rt_render(ctx, &P, &fb);
  double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  if (rc != RT_OK) { std::fprintf(stderr, "ERROR: %s\n", rt_last_error(ctx)); return 1; }
  std::printf("Render time is %.3f s (%dx%d)\n", sec, W, H);
  if (rth_write_png(out.c_str(), rgb.data(), W, H, 3) != 0) std::fprintf(stderr, "WARNING: cannot write %s\n", out.c_str());
  if (!den.empty()) {
    std::vector<float> in((size_t)W * H * 3), o((size_t)W * H * 3);
    for (size_t i = 0; i < (size_t)W * H * 3; i++) in[i] = rgb[i] / 255.0f;  // RayTracer.cpp:70-74: the 8-bit image re-floated
    // colour + the albedo / normal guides of the first primary hit (the overload the reference declares, denoiser.h:16-17)
    if (rt_denoise(ctx, in.data(), o.data(), W, H, albedo.data(), normal.data()) == RT_OK) {
      for (size_t i = 0; i < (size_t)W * H * 3; i++) {
        float v = o[i] * 255.0f + 0.5f;
        rgb[i] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
      }
      rth_write_png(den.c_str(), rgb.data(), W, H, 3);
    } else {
      std::fprintf(stderr, "WARNING: denoise: %s\n", rt_last_error(ctx));
    }
  }
  // RenderImage::ComputeZBufferImage / ComputeSampleCountImage (renderer.h:74-75): grey images scaled to the range present
  auto write_grey = [&](const std::string& path, auto value, bool invert) {
    float lo = 1e30f, hi = -1e30f;
    for (size_t i = 0; i < (size_t)W * H; i++) {
      float v = value(i);
      if (v >= 1e30f) continue;  // BIGFLOAT: nothing hit
      lo = v < lo ? v : lo; hi = v > hi ? v : hi;
    }
    std::vector<uint8_t> g((size_t)W * H, 0);
    for (size_t i = 0; i < (size_t)W * H; i++) {
      float v = value(i);
      if (v >= 1e30f || hi <= lo) continue;
      float f = (v - lo) / (hi - lo);
      if (invert) f = 1.0f - f;
      g[i] = (uint8_t)(f * 255.0f + 0.5f);
    }
    if (rth_write_png(path.c_str(), g.data(), W, H, 1) != 0) std::fprintf(stderr, "WARNING: cannot write %s\n", path.c_str());
  };
  if (!zimg.empty()) write_grey(zimg, [&](size_t i) { return depth[i]; }, true);
  if (!simg.empty()) write_grey(simg, [&](size_t i) { return (float)nsamples[i]; }, false);
  rt_context_destroy(ctx);
  rth_scene_free(scene);
  return 0;
}
