// `ray` -- headless driver with the reference's command line
// (src/main.cpp:20-38, src/ui/CommandLineUI.cpp:18-52,68-122):
//
//     ray [-r depth] [-w width] in.ray out.bmp
//
// It loads the scene like RayTracer::loadScene, sizes the frame like
// CommandLineUI::run (height = (int)(width / aspect + 0.5)), and replaces the
// std::thread fan-out over column strips by one rt_render call on the GPU.
// Extra switches (not in the reference): --reference-traversal walks the
// reference's KD-trees with its exact rules instead of the library's BVH;
// --aa N renders N x N jittered samples per pixel (tracePixelAntiAlias);
// --threshold T with --aa selects the GUI's adaptive mode; --gpus N spreads the
// frame over N GPUs of the node (one host thread per GPU where the reference
// has one per column strip, include/rt_multi.h).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <iostream>
#include <string>
#include <vector>

#include "../../include/rt_amd.h"
#include "../../include/rt_multi.h"
#include "host_api.h"

namespace {
void usage(const char* prog, int depth, int size) {
  std::cerr << "usage: " << prog << " [options] [input.ray output.bmp]" << std::endl;
  std::cerr << "  -r <#>      set recursion level (default " << depth << ")" << std::endl;
  std::cerr << "  -w <#>      set output image width (default " << size << ")" << std::endl;
}
}  // namespace

int main(int argc, char** argv) {
  int depth = 2, size = 512;  // TraceUI defaults (src/ui/TraceUI.h:21)
  int aa = 0, threshold = -1, gpus = 0;
  bool referenceTraversal = false;
  std::vector<std::string> files;
  for (int i = 1; i < argc; i++) {
    std::string a = argv[i];
    if (a == "-r" && i + 1 < argc) depth = std::atoi(argv[++i]);
    else if (a == "-w" && i + 1 < argc) size = std::atoi(argv[++i]);
    else if (a == "--aa" && i + 1 < argc) aa = std::atoi(argv[++i]);
    else if (a == "--threshold" && i + 1 < argc) threshold = std::atoi(argv[++i]);
    else if (a == "--gpus" && i + 1 < argc) gpus = std::atoi(argv[++i]);
    else if (a == "--reference-traversal") referenceTraversal = true;
    else if (!a.empty() && a[0] == '-') {
      std::cerr << "Invalid argument: '" << a << "'." << std::endl;
      usage(argv[0], depth, size);
      return 1;
    } else files.push_back(a);
  }
  if (files.size() < 2) {
    std::cerr << "no input and/or output name." << std::endl;
    return 1;
  }
  RthOptions opt;
  rth_default_options(&opt);
  char err[1024] = {0};
  RthScene* hs = rth_scene_load(files[0].c_str(), &opt, err, sizeof err);
  if (!hs) {
    std::cerr << err << std::endl;
    std::cerr << "Unable to load ray file '" << files[0] << "'" << std::endl;
    return 1;
  }
  const int width = size;
  const int height = (int)(width / rth_scene_aspect_ratio(hs) + 0.5);

  if (gpus < 0 || (gpus > 0 && aa > 0 && threshold >= 0)) {
    std::cerr << "--gpus takes a positive count and does not go with --threshold" << std::endl;
    return 1;
  }
  RtScene* scene = nullptr;
  RtMulti* multi = nullptr;
  if (gpus > 0) {
    if (rt_multi_create(rth_scene_desc(hs), gpus, nullptr, &multi) != RT_OK) {
      std::cerr << "rt_multi_create: " << rt_multi_last_error() << std::endl;
      return 1;
    }
  } else if (rt_scene_create(rth_scene_desc(hs), &scene) != RT_OK) {
    std::cerr << "rt_scene_create: " << rt_last_error() << std::endl;
    return 1;
  }
  RtRenderParams p;
  std::memset(&p, 0, sizeof p);
  p.width = width;
  p.height = height;
  p.depth = depth;
  p.filter_width = 1;
  p.traversal = referenceTraversal ? RT_TRAVERSE_REFERENCE : RT_TRAVERSE_FAST;
  p.tile_w = p.tile_h = 32;
  p.shard_count = 1;
  std::vector<double> jitter;
  if (aa > 0) {
    p.aa_mode = threshold >= 0 ? RT_AA_ADAPTIVE : RT_AA_FULL;
    p.aa_samples = aa;
    p.aa_threshold = threshold;
    jitter.resize((size_t)2 * aa * aa);
    rth_jitter_table((unsigned)time(NULL), aa, jitter.data());  // srand(time(NULL)), RayTracer.cpp:70
    p.jitter = jitter.data();
  }
  std::vector<unsigned char> buf((size_t)width * height * 3);
  RtStats st;
  const auto t0 = std::chrono::steady_clock::now();
  if (multi) {
    if (rt_multi_render(multi, &p, buf.data(), &st) != RT_OK) {
      std::cerr << "rt_multi_render: " << rt_multi_last_error() << std::endl;
      return 1;
    }
  } else if (rt_render(scene, &p, buf.data(), nullptr, &st) != RT_OK) {
    std::cerr << "rt_render: " << rt_last_error() << std::endl;
    return 1;
  }
  const double t = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  if (rth_write_bmp(files[1].c_str(), width, height, buf.data()) != 0) {
    std::cerr << "cannot write " << files[1] << std::endl;
    return 1;
  }
  const unsigned long long rays = st.rays_primary + st.rays_reflect + st.rays_refract + st.rays_shadow;
  std::cout << "total time = " << t << " seconds, rays traced = " << rays << std::endl;
  if (multi) rt_multi_destroy(multi);
  if (scene) rt_scene_destroy(scene);
  rth_scene_free(hs);
  return 0;
}
