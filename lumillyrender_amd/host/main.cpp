// main.cpp -- lumilly_render: stand-alone driver, the counterpart of the reference binary
// (src/main.rs:43-145): load the scene file, render it on every visible GPU (one host thread per
// device, pixel tiles dealt diagonally (lr_host_tiles), replicated scene), save png/hdr.
//
//   lumilly_render <scene.toml> [--seed N] [--gpus N] [--spp N] [--out FILE] [--assets DIR]
//
// Prints the same kind of lines as the reference (resolution, spp, integrator, elapse).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <string>
#include <thread>
#include <vector>

#include "../../include/lumilly_host.h"

int main(int argc, char** argv) {
  auto t_start = std::chrono::steady_clock::now();
  if (argc <= 1) { std::fprintf(stderr, "Path for .toml must be specified.\n"); return 2; }      // main.rs:47-49
  std::string scene_path = argv[1], out_path, assets = "assets";
  unsigned seed = 0; int gpus = 0, spp_override = 0;
  for (int i = 2; i < argc; ++i) {
    std::string a = argv[i];
    auto next = [&](const char* what) -> const char* { if (i + 1 >= argc) { std::fprintf(stderr, "%s needs a value\n", what); std::exit(2); } return argv[++i]; };
    if (a == "--seed") seed = (unsigned)std::strtoul(next("--seed"), nullptr, 10);
    else if (a == "--gpus") gpus = std::atoi(next("--gpus"));
    else if (a == "--spp") spp_override = std::atoi(next("--spp"));
    else if (a == "--out") out_path = next("--out");
    else if (a == "--assets") assets = next("--assets");
    else { std::fprintf(stderr, "unknown option %s\n", a.c_str()); return 2; }
  }
  std::printf("loading: %s\n", scene_path.c_str());
  LrHostScene* hs = nullptr;
  if (lr_host_load_scene(scene_path.c_str(), assets.c_str(), &hs) != LR_OK) { std::fprintf(stderr, "error: %s\n", lr_host_last_error()); return 1; }
  LrRendererConfig rc; LrFilmConfig fc;
  lr_host_scene_renderer(hs, &rc); lr_host_scene_film(hs, &fc);
  const LrSceneDesc* desc = lr_host_scene_desc(hs);
  const int W = fc.resolution[0], H = fc.resolution[1];
  const int spp = spp_override > 0 ? spp_override : rc.samples;
  double bvh_s = 0; int nodes = 0, depth = 0;
  lr_host_scene_bvh_info(hs, &bvh_s, &nodes, &depth, nullptr);
  std::printf("resolution: %dx%d\nspp: %d\npolygons: %d\nbvh construction: %.3fs (%d nodes, depth %d)\n", W, H, spp, desc->n_prims, bvh_s, nodes, depth);
  std::printf("integrator: %s\n", rc.integrator == LR_INTEGRATOR_PT ? "pt" : "pt-direct");
  int avail = lr_device_count();
  if (avail <= 0) { std::fprintf(stderr, "error: no HIP device (there is no CPU fallback)\n"); return 1; }
  if (gpus <= 0 || gpus > avail) gpus = avail;
  std::printf("gpus: %d\n", gpus);

  LrRenderParams rp; std::memset(&rp, 0, sizeof(rp));
  rp.integrator = rc.integrator; rp.spp = spp; rp.seed = seed; rp.depth = rc.depth; rp.depth_limit = rc.depth_limit;
  rp.no_direct_emitter = rc.no_direct_emitter;
  std::vector<float> film((size_t)W * H * 3, 0.0f);
  std::vector<int> status((size_t)gpus, LR_OK);
  std::vector<std::string> errors((size_t)gpus);
  auto t_render = std::chrono::steady_clock::now();
  std::vector<std::thread> th;
  for (int g = 0; g < gpus; ++g) th.emplace_back([&, g] {
    LrScene* sc = nullptr;
    int rcode = lr_scene_create(g, desc, &sc);
    if (rcode == LR_OK) {
      int n = lr_host_tiles(W, H, lr_host_default_tile(), g, gpus, nullptr, 0);
      std::vector<LrTile> tiles((size_t)std::max(n, 1));
      lr_host_tiles(W, H, lr_host_default_tile(), g, gpus, tiles.data(), n);
      rcode = lr_render(sc, &rp, tiles.data(), n, film.data(), (size_t)W * 3);   // disjoint tiles: threads share the film
    }
    if (rcode != LR_OK) errors[(size_t)g] = lr_last_error();
    status[(size_t)g] = rcode;
    lr_scene_destroy(sc);
  });
  for (auto& t : th) t.join();
  for (int g = 0; g < gpus; ++g) if (status[(size_t)g] != LR_OK) { std::fprintf(stderr, "error (gpu %d): %s\n", g, errors[(size_t)g].c_str()); return 1; }
  double render_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_render).count();
  std::printf("render: %.3fs  %.1f Msamples/s\n", render_s, (double)W * H * spp / render_s / 1e6);

  std::printf("saving...\n");
  if (out_path.empty()) {                                     // main.rs:148-153 (the directory must exist)
    char stamp[32]; std::time_t now = std::time(nullptr); std::strftime(stamp, sizeof(stamp), "%Y%m%d%H%M%S", std::localtime(&now));
    out_path = std::string("images/image_") + stamp + "_" + std::to_string(spp) + (fc.output == LR_OUTPUT_HDR ? ".hdr" : ".png");
  }
  int src = fc.output == LR_OUTPUT_HDR ? lr_host_save_hdr(out_path.c_str(), film.data(), W, H, (size_t)W * 3)
                                        : lr_host_save_png(out_path.c_str(), film.data(), W, H, (size_t)W * 3, fc.gamma);
  if (src != LR_OK) { std::fprintf(stderr, "error: %s\n", lr_host_last_error()); return 1; }
  std::printf("wrote %s\n", out_path.c_str());
  lr_host_scene_free(hs);
  std::printf("elapse: %.3fs\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count());
  return 0;
}
