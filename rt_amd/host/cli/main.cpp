// rt_cli -- headless driver of the demos on the HIP backend.
//
// Keeps the reference's command-line flags (main.cpp:66-77: -w -h -s -t -d -c) and its demo selector
// (main.cpp:85-107: one Demo per scene, each built around a Frame of the chosen backend), and replaces the
// GLFW window by a file dump.  The frame is driven exactly as App drives it (app.cpp:61-109): zero the
// buffer, TraceAllRays, then poll InProgress/CopyToBuffer every 0.25 s and HasFinished for completion.
// This binary has no CPU renderer: without a usable GPU it reports the error and exits with status 2.
#include "demos/glass_spheres.hpp"
#include "demos/pretty_spheres.hpp"
#include "demos/scene_file.hpp"
#include "demos/synthetic_spheres.hpp"
#include "rt_hip/demo_frame_hip.hpp"

#include "rt_host.h"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <string>
#include <thread>
#include <vector>

namespace
{
struct Options
{
  uint32_t width = 1024;
  uint32_t height = 768;
  uint32_t threads = 4;  // accepted for compatibility; the HIP backend has no host worker threads
  uint32_t samplesInRow = 2;
  uint32_t demoIndex = 1;
  bool enableGpu = true;
  uint32_t sceneSeed = 1;
  uint32_t seed = 1;
  bool deterministic = false;
  uint32_t maxDepth = 0;
  uint32_t spheres = 10000;
  bool highQuality = false;
  bool progress = false;
  uint32_t frames = 1;
  int device = -1;
  std::string out;
  std::string json;
  std::string scene;      // --scene FILE: render this scene file instead of a built-in demo
  std::string dumpScene;  // --dump-scene FILE: write the chosen scene to a scene file
};

bool ParseUint(char const * text, uint32_t & value)
{
  char * end = nullptr;
  unsigned long const v = std::strtoul(text, &end, 10);
  if (end == text || *end != '\0')
    return false;
  value = static_cast<uint32_t>(v);
  return true;
}

// Unknown flags are tolerated, as cxxopts is configured to do in the reference (main.cpp:67).
bool ParseArgs(int argc, char * argv[], Options & o)
{
  for (int i = 1; i < argc; ++i)
  {
    std::string const a = argv[i];
    auto const value = [&](uint32_t & dst) {
      if (i + 1 >= argc || !ParseUint(argv[i + 1], dst))
      {
        std::cout << "Error parsing options: bad value for " << a << std::endl;
        return false;
      }
      ++i;
      return true;
    };
    if (a == "-w" || a == "--width") { if (!value(o.width)) return false; }
    else if (a == "-h" || a == "--height") { if (!value(o.height)) return false; }
    else if (a == "-s" || a == "--samples") { if (!value(o.samplesInRow)) return false; }
    else if (a == "-t" || a == "--threads") { if (!value(o.threads)) return false; }
    else if (a == "-d" || a == "--demo") { if (!value(o.demoIndex)) return false; }
    else if (a == "-c" || a == "--cuda" || a == "--gpu")
    {
      // Optional boolean value, default true.
      if (i + 1 < argc && (!std::strcmp(argv[i + 1], "false") || !std::strcmp(argv[i + 1], "0")))
      {
        o.enableGpu = false;
        ++i;
      }
      else if (i + 1 < argc && (!std::strcmp(argv[i + 1], "true") || !std::strcmp(argv[i + 1], "1")))
        ++i;
    }
    else if (a == "--backend")
    {
      if (i + 1 >= argc)
        return false;
      o.enableGpu = std::string(argv[++i]) == "hip";
    }
    else if (a == "--scene-seed") { if (!value(o.sceneSeed)) return false; }
    else if (a == "--seed") { if (!value(o.seed)) return false; }
    else if (a == "--max-depth") { if (!value(o.maxDepth)) return false; }
    else if (a == "--spheres") { if (!value(o.spheres)) return false; }
    else if (a == "--frames") { if (!value(o.frames)) return false; }
    else if (a == "--device") { uint32_t d = 0; if (!value(d)) return false; o.device = static_cast<int>(d); }
    else if (a == "--deterministic") o.deterministic = true;
    else if (a == "--hq") o.highQuality = true;
    else if (a == "--progress") o.progress = true;
    else if (a == "--out") { if (i + 1 >= argc) return false; o.out = argv[++i]; }
    else if (a == "--bench-json") { if (i + 1 >= argc) return false; o.json = argv[++i]; }
    else if (a == "--scene") { if (i + 1 >= argc) return false; o.scene = argv[++i]; }
    else if (a == "--dump-scene") { if (i + 1 >= argc) return false; o.dumpScene = argv[++i]; }
  }
  return true;
}

bool EndsWith(std::string const & s, char const * suffix)
{
  size_t const n = std::strlen(suffix);
  return s.size() >= n && s.compare(s.size() - n, n, suffix) == 0;
}

// PFM stores rows bottom-up; the ColorBuffer has row 0 on top.
bool WritePfm(std::string const & path, ray_tracing::ColorBuffer const & buffer, uint32_t w, uint32_t h)
{
  std::ofstream f(path, std::ios::binary);
  if (!f)
    return false;
  f << "PF\n" << w << " " << h << "\n-1.0\n";
  for (uint32_t y = 0; y < h; ++y)
    f.write(reinterpret_cast<char const *>(buffer.data() + static_cast<size_t>(h - 1 - y) * w),
            static_cast<std::streamsize>(w * sizeof(ray_tracing::vec3)));
  return static_cast<bool>(f);
}

// 8-bit output through the viewer's own conversion (App::UpdateTexture, app.cpp:128-138).
bool WritePpm(std::string const & path, ray_tracing::ColorBuffer const & buffer, uint32_t w, uint32_t h)
{
  std::vector<uint8_t> rgba(static_cast<size_t>(w) * h * 4);
  rt_host_update_texture(reinterpret_cast<float const *>(buffer.data()), static_cast<size_t>(w) * h, rgba.data());
  std::ofstream f(path, std::ios::binary);
  if (!f)
    return false;
  f << "P6\n" << w << " " << h << "\n255\n";
  for (size_t i = 0; i < static_cast<size_t>(w) * h; ++i)
    f.write(reinterpret_cast<char const *>(&rgba[i * 4]), 3);
  return static_cast<bool>(f);
}
}  // namespace

int main(int argc, char * argv[])
{
  Options o;
  if (!ParseArgs(argc, argv, o))
    return 1;

  if (!o.enableGpu)
  {
    std::cout << "This build renders on the HIP backend only (-c false / --backend cpu is not available)."
              << std::endl;
    return 2;
  }

  // The demo selector: one vector of demos per backend, each demo built around a Frame of that backend
  // (main.cpp:85-107).
  std::vector<ray_tracing_hip::DemoFrameHIP *> frames;
  auto makeFrame = [&]() {
    auto f = std::make_unique<ray_tracing_hip::DemoFrameHIP>(o.device);
    f->SetSeed(o.seed);
    f->SetDeterministic(o.deterministic);
    f->SetMaxDepth(o.maxDepth);
    frames.push_back(f.get());
    return f;
  };
  std::vector<std::unique_ptr<demo::Demo>> demosGpu;
  demosGpu.emplace_back(std::make_unique<demo::PrettySpheres>(makeFrame(), o.sceneSeed));
  demosGpu.emplace_back(std::make_unique<demo::GlassSpheres>(makeFrame(), o.sceneSeed));
  demosGpu.emplace_back(std::make_unique<demo::SyntheticSpheres>(makeFrame(), o.spheres, o.sceneSeed));

  demo::SceneFile * sceneFile = nullptr;
  if (!o.scene.empty())
  {
    auto d = std::make_unique<demo::SceneFile>(makeFrame(), o.scene);
    sceneFile = d.get();
    demosGpu.emplace_back(std::move(d));
    o.demoIndex = static_cast<uint32_t>(demosGpu.size());
  }

  o.demoIndex = std::max(static_cast<uint32_t>(1), o.demoIndex);
  if (o.demoIndex > demosGpu.size())
  {
    std::cout << "Incorrect demo index." << std::endl;
    return 1;
  }
  auto & demo = demosGpu[o.demoIndex - 1];
  auto * hip = frames[o.demoIndex - 1];

  auto buffer = std::make_shared<ray_tracing::ColorBuffer>(static_cast<size_t>(o.width) * o.height);
  ray_tracing::ColorBuffer realtime(buffer->size());
  if (!demo->Initialize(buffer, o.width, o.height, o.samplesInRow))
  {
    std::cout << "Demo initialization failed." << std::endl;
    if (sceneFile != nullptr)
      std::cout << sceneFile->Error() << std::endl;
    return 1;
  }
  auto & frame = demo->GetFrame();
  if (!o.dumpScene.empty())
  {
    auto const & flat = hip->Scene();
    auto const cam = hip->Camera();
    if (!demo::WriteSceneFile(o.dumpScene, cam, flat.CentersRadii().data(), flat.MaterialIndex().data(),
                              flat.SphereCount(), flat.Materials().data(),
                              static_cast<uint32_t>(flat.Materials().size()), flat.Lights().data(),
                              static_cast<uint32_t>(flat.Lights().size())))
    {
      std::cout << "Cannot write " << o.dumpScene << std::endl;
      return 1;
    }
  }

  // High-quality mode raises the sample count to at least 20 per row on a GPU backend (app.cpp:92-99).
  if (o.highQuality)
    frame->SetSamplesInRowCount(std::max(frame->GetSamplesInRowCount(), 20u));
  uint32_t const samples = frame->GetSamplesInRowCount();

  double bestMs = 0.0;
  uint32_t snapshots = 0;
  for (uint32_t i = 0; i < std::max(o.frames, 1u); ++i)
  {
    std::fill(buffer->begin(), buffer->end(), ray_tracing::vec3(0.0f, 0.0f, 0.0f));
    auto const t0 = std::chrono::steady_clock::now();
    auto lastUpdate = t0;
    frame->TraceAllRays();
    if (hip->Failed())
    {
      std::cout << "HIP backend error: " << hip->LastError() << std::endl;
      return 2;
    }
    while (!frame->HasFinished())
    {
      auto const now = std::chrono::steady_clock::now();
      if (o.progress && frame->InProgress() && std::chrono::duration<double>(now - lastUpdate).count() >= 0.25)
      {
        lastUpdate = now;
        frame->CopyToBuffer(realtime);
        ++snapshots;
      }
      std::this_thread::sleep_for(std::chrono::microseconds(100));
    }
    if (hip->Failed())
    {
      std::cout << "HIP backend error: " << hip->LastError() << std::endl;
      return 2;
    }
    double const ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    bestMs = (i == 0) ? ms : std::min(bestMs, ms);
  }

  auto const & st = hip->LastStats();
  double const msamples = static_cast<double>(st.samples) / 1e6;
  char line[768];
  std::snprintf(line, sizeof(line),
                "{\"backend\": \"hip\", \"demo\": %u, \"width\": %u, \"height\": %u, \"samples_in_row\": %u, "
                "\"ms_per_frame\": %.3f, \"kernel_ms\": %.3f, \"msamples_per_s\": %.2f, "
                "\"samples\": %llu, \"ray_queries\": %llu, \"hit_queries\": %llu, \"shaded_bounces\": %llu, "
                "\"sphere_tests\": %llu, \"snapshots\": %u}",
                o.demoIndex, o.width, o.height, samples, bestMs, st.kernel_ms,
                bestMs > 0.0 ? msamples / (bestMs * 1e-3) : 0.0, static_cast<unsigned long long>(st.samples),
                static_cast<unsigned long long>(st.ray_queries), static_cast<unsigned long long>(st.hit_queries),
                static_cast<unsigned long long>(st.shaded_bounces),
                static_cast<unsigned long long>(st.sphere_tests), snapshots);
  std::cout << line << std::endl;
  if (!o.json.empty())
    std::ofstream(o.json) << line << "\n";

  if (!o.out.empty())
  {
    bool const ok = EndsWith(o.out, ".pfm") ? WritePfm(o.out, *buffer, o.width, o.height)
                                            : WritePpm(o.out, *buffer, o.width, o.height);
    if (!ok)
    {
      std::cout << "Cannot write " << o.out << std::endl;
      return 1;
    }
  }
  frame->Uninitialize();
  return 0;
}
