// restir_demo — the reference application's call order without the window (src/main.cpp:50-264, src/sample_example.cpp):
//   loadEnvironmentHdr -> loadScene (Scene::load + AccelStructure::create) -> createRender -> per frame:
//   updateFrame / Scene::updateCamera -> Renderer::run -> (post.frag's sum of the two HDR images, written to disk)
// Usage mirrors main.cpp:52-54:  restir_demo [-f scene.gltf | -p cornell|helmet|sponza|bistro|interior] [-e env.hdr]
//                                            [-w 1920] [-h 1080] [-n frames] [-o out] [-s scale] [-a autoExposure]
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include "renderer.hpp"

using namespace rth;

static const char* arg(int argc, char** argv, const char* key, const char* def)
{
  for(int i = 1; i + 1 < argc; i++) if(strcmp(argv[i], key) == 0) return argv[i + 1];
  return def;
}

int main(int argc, char** argv)
{
  const int W = atoi(arg(argc, argv, "-w", "1920")), H = atoi(arg(argc, argv, "-h", "1080")), frames = atoi(arg(argc, argv, "-n", "32"));
  const std::string file = arg(argc, argv, "-f", ""), proc = arg(argc, argv, "-p", "cornell"), envFile = arg(argc, argv, "-e", ""), out = arg(argc, argv, "-o", "frame");
  const float scale = float(atof(arg(argc, argv, "-s", "1.0")));

  HdrSampling sky;
  bool haveSky = false;
  if(!envFile.empty()) haveSky = sky.loadEnvironment(envFile);
  else if(proc != "cornell") { sky.makeSyntheticSky(2048, 1024, 5e4f, 7); haveSky = true; }

  Scene scene;
  scene.setup();
  if(!file.empty()) { if(!scene.load(file)) { fprintf(stderr, "cannot load %s\n", file.c_str()); return 1; } }
  else {
    const char* names[] = {"cornell", "helmet", "sponza", "bistro", "interior"};
    int kind = 0;
    for(int i = 0; i < 5; i++) if(proc == names[i]) kind = i;
    scene.loadFromGltfScene(makeProceduralScene(ProcScene(kind), scale, 1), proc);
  }
  const SceneStats& s = scene.getStat();
  printf("scene %s: %llu triangles (%llu instanced), %u materials, %u textures, %u emissive triangles\n", scene.getSceneName().c_str(),
         (unsigned long long)s.triangles, (unsigned long long)s.instancedTriangles, s.materials, s.textures, s.trigLights);

  Renderer render;
  if(!render.setup(0)) return 2;
  AccelStructure accel;
  accel.setup(render.context());
  if(!accel.create(scene, haveSky ? &sky : nullptr)) return 3;
  if(!render.create(W, H, &scene)) return 4;

  // RtxState as SampleExample fills it (sample_example.hpp:154-184, sample_example.cpp:87, 104-105)
  rt_state st{};
  st.maxDepth = 4; st.modulate = 1; st.fireflyClampThreshold = haveSky ? sky.getIntegral() * 4.f : 100.f; st.hdrMultiplier = 1.f;
  st.environmentProb = haveSky ? 0.25f : 0.f; st.ReSTIRState = RT_RESTIR_TEMPORAL; st.RISSampleNum = 4; st.reservoirClamp = 80;
  st.size = rt_ivec2{W, H}; st.envMapLuminIntegInv = haveSky ? 1.f / sky.getIntegral() : 0.f;
  st.lightLuminIntegInv = 1.f / (scene.m_trigLightWeight + scene.m_puncLightWeight); st.MIS = 1;
  st.sigLuminDirect = 0.4f; st.sigNormalDirect = 0.1f; st.sigDepthDirect = 0.02f; st.denoise = 1;
  st.sigLuminIndirect = 4.f; st.sigNormalIndirect = 0.4f; st.sigDepthIndirect = 1.f;

  auto t0 = std::chrono::steady_clock::now();
  for(int f = 0; f < frames; f++) {
    scene.updateCamera(W, H);
    render.setCamera(scene.getCamera());
    st.frame = f; st.time = 1000u + unsigned(f);
    if(!render.run(st, f)) return 5;
  }
  rt_sync(render.context());
  const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  printf("%d frames %dx%d: %.3f ms/frame\n", frames, W, H, ms / frames);

  std::vector<float> d, i;
  if(!render.readResult(frames - 1, d, i)) return 6;
  {  // PFM (bottom-up scanlines, little endian) of direct + indirect = the HDR frame
    std::ofstream pfm(out + ".pfm", std::ios::binary);
    pfm << "PF\n" << W << " " << H << "\n-1.0\n";
    for(int y = H - 1; y >= 0; y--) for(int x = 0; x < W; x++) { float c[3]; for(int k = 0; k < 3; k++) c[k] = d[(size_t(y) * W + x) * 4 + k] + i[(size_t(y) * W + x) * 4 + k]; pfm.write(reinterpret_cast<const char*>(c), 12); }
  }
  {  // the displayed frame: RenderOutput::run (post.frag: Uncharted 2 tone curve + dither) -> PPM + PNG
    RenderOutput offscreen;
    offscreen.setup(render.context());
    offscreen.create(W, H);
    offscreen.m_tm.autoExposure = atoi(arg(argc, argv, "-a", "0"));
    std::vector<uint8_t> rgba;
    if(!offscreen.run(st, 1.0f, rt_vec2{1.0f, 1.0f}, frames - 1) || !offscreen.readImage(rgba)) return 7;
    std::ofstream ppm(out + ".ppm", std::ios::binary);
    ppm << "P6\n" << W << " " << H << "\n255\n";
    for(size_t p = 0; p < size_t(W) * H; p++) ppm.write(reinterpret_cast<const char*>(&rgba[p * 4]), 3);
    if(!writePng(out + ".png", rgba.data(), W, H)) fprintf(stderr, "cannot write %s.png\n", out.c_str());
  }
  render.destroy();
  return 0;
}
