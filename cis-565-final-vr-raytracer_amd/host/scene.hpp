// scene.hpp — host scene container with the reference's Scene interface (src/scene.hpp:59-117).
//
// Kept from the reference: setup / load / updateCamera / destroy, getStat, public m_puncLightWeight /
// m_trigLightWeight, and the data products of scene.cpp:179-448, 700-826 (packed vertices, instance table,
// materials, light lists with alias tables, camera block with last-frame matrices).
// Replaced: every Vulkan buffer/descriptor (scene.cpp:453-508, 650-695) — the products are exposed as one
// rt_scene_desc for rt_upload_scene instead.
#pragma once
#include <array>
#include <string>
#include <vector>
#include "../../include/rt_abi.h"
#include "host_math.h"
#include "hdr_sampling.hpp"

namespace rth {

// nvh::GltfMaterial subset actually consumed by the reference (scene.cpp:415-448, accelstruct.cpp:140-149)
struct GltfMaterial {
  float baseColorFactor[4] = {1, 1, 1, 1};
  int baseColorTexture = -1;
  float metallicFactor = 1.f, roughnessFactor = 1.f;
  int metallicRoughnessTexture = -1;
  int emissiveTexture = -1;
  float emissiveFactor[3] = {0, 0, 0};
  int normalTexture = -1;
  float normalTextureScale = 1.f;
  float transmissionFactor = 0.f;
  int transmissionTexture = -1;
  float ior = 1.5f;
  int alphaMode = RT_ALPHA_OPAQUE;
  float alphaCutoff = 0.5f;
  int doubleSided = 0;
};
struct GltfPrimMesh { uint32_t firstIndex = 0, indexCount = 0, vertexOffset = 0, vertexCount = 0; int materialIndex = 0; };
struct GltfNode { M4 worldMatrix = M4::identity(); int primMesh = 0; };
struct GltfLight {  // KHR_lights_punctual
  M4 worldMatrix = M4::identity();
  float color[3] = {1, 1, 1};
  float intensity = 1.f, range = 0.f, innerConeAngle = 0.f, outerConeAngle = 0.785398163f;
  int type = 1;  // LightType_Point (host_device.h:252-254)
};
struct GltfCamera { V3 eye{0, 0, 1}, center{0, 0, 0}, up{0, 1, 0}; float yfovDeg = 45.f; };
// host/png_writer.cpp: the displayed frame (RT_BUF_LDR, R first) as an 8-bit PNG
bool encodePng(const uint8_t* rgba, int width, int height, bool keepAlpha, std::vector<uint8_t>& out);
bool writePng(const std::string& path, const uint8_t* rgba, int width, int height, bool keepAlpha = false);
struct TextureImage { int width = 1, height = 1; std::vector<uint8_t> bgra; int wrapS = RT_WRAP_REPEAT, wrapT = RT_WRAP_REPEAT, magFilter = RT_FILTER_LINEAR; };

// nvh::GltfScene stand-in: flattened arrays shared by all primitive meshes
struct GltfScene {
  std::vector<V3> positions, normals;
  std::vector<std::array<float, 4>> tangents, colors0;
  std::vector<std::array<float, 2>> texcoords0;
  std::vector<uint32_t> indices;
  std::vector<GltfPrimMesh> primMeshes;
  std::vector<GltfNode> nodes;
  std::vector<GltfMaterial> materials;
  std::vector<GltfLight> lights;
  std::vector<GltfCamera> cameras;
  std::vector<TextureImage> textures;
  V3 bboxMin{0, 0, 0}, bboxMax{0, 0, 0};
};

struct SceneStats { uint64_t triangles = 0, instancedTriangles = 0, vertices = 0; uint32_t primMeshes = 0, nodes = 0, materials = 0, textures = 0, puncLights = 0, trigLights = 0; };

class Scene {
 public:
  // scene.hpp:62-64 — the Vulkan handles the reference passes here have no equivalent; kept for call-site shape
  void setup() {}
  // scene.hpp:65 / scene.cpp:57-125.  .gltf (external or base64 buffers) and .glb; returns false on failure.
  bool load(const std::string& filename);
  // Build from an in-memory GltfScene (used by the procedural stand-ins for the absent assets)
  bool loadFromGltfScene(GltfScene&& gltf, const std::string& name);
  void destroy();
  // scene.cpp:777-826 — shifts current matrices into last*, applies the half-pixel jitter, refreshes `m_camera`
  void updateCamera(int width, int height);
  void setCamera(V3 eye, V3 center, V3 up, float fovDeg) { m_eye = eye; m_center = center; m_up = up; m_fov = fovDeg; }
  void fitCamera();  // CameraManip.fit stand-in (scene.cpp:311)

  const rt_scene_camera& getCamera() const { return m_camera; }
  const SceneStats& getStat() const { return m_stats; }
  const std::string& getSceneName() const { return m_sceneName; }
  const GltfScene& getScene() const { return m_gltf; }
  // The upload payload; `env` may be null (=> 1x1 black environment).  Pointers stay valid until destroy()/load().
  rt_scene_desc getDesc(const HdrSampling* env) const;

  float m_puncLightWeight = 0.f, m_trigLightWeight = 0.f;  // scene.hpp:79-80
  rt_light_buf_info m_lightBufInfo{};                      // scene.hpp:113 (zero-initialised here: quirk 10)
  V3 m_eye{0, 0, 1}, m_center{0, 0, 0}, m_up{0, 1, 0};
  float m_fov = 45.f;

 private:
  void createMaterialBuffer();      // scene.cpp:415-448
  void createPuncLightBuffer();     // scene.cpp:319-353, 700-726
  void createVertexBuffer();        // scene.cpp:209-289
  void createInstanceDataBuffer();  // scene.cpp:179-195 + accelstruct.cpp:132-162 (instance flags)
  void createTrigLightBuffer();     // scene.cpp:355-409, 741-772

  GltfScene m_gltf;
  std::string m_sceneName;
  SceneStats m_stats;
  rt_scene_camera m_camera{};
  bool m_cameraInit = false;
  V3 m_lastEye{0, 0, 0};
  // upload products
  std::vector<rt_prim_mesh> m_primMeshes;
  std::vector<rt_vertex> m_vertices;
  std::vector<rt_instance> m_instances;
  std::vector<rt_material> m_materials;
  std::vector<rt_texture> m_textures;
  std::vector<rt_punc_light> m_puncLights;
  std::vector<rt_trig_light> m_trigLights;
};

// ---- procedural stand-ins for the assets the reference downloads / the benchmark names (scene_gen.cpp) ----
// 5 / 6 (round 5): the SAME classes with the memory footprint of the assets they stand in for — the reference uploads every image at full size, BGRA8, no mips
// (src/scene.cpp:554-646): the exterior street with 128 materials of 2k^2 texture sets (3.3 GB), 16 distinct 1k^2 foliage cards and long thin triangles;
// the atrium with SURVEY 8(d)'s 1k^2 textures.  0-4 are unchanged (`lite`): every fixture and digest of rounds 1-4 keeps its scene.
enum ProcScene { PROC_CORNELL = 0, PROC_HELMET = 1, PROC_SPONZA = 2, PROC_BISTRO_EXT = 3, PROC_BISTRO_INT = 4, PROC_BISTRO_EXT_REAL = 5, PROC_SPONZA_1K = 6 };
// scale in (0,1] shrinks tessellation (triangle count ~ scale) for quick tests; seed fixes every random choice
GltfScene makeProceduralScene(ProcScene kind, float scale, uint32_t seed);

// gltf_loader.cpp — tinygltf + nvh::GltfScene::importMaterials/importDrawableNodes stand-in (scene.cpp:72-74, 130-173)
bool loadGltfFile(const std::string& filename, GltfScene& out, std::string& error);
// self-contained .gltf (base64 buffers + PNG images) from an in-memory scene; exchange/test utility, no reference counterpart
bool saveGltfFile(const std::string& filename, const GltfScene& g, std::string& error);

}  // namespace rth
