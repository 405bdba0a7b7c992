// scene.cpp — host Scene: the data products of the reference's Scene class (src/scene.cpp), minus Vulkan.
#include "scene.hpp"
#include "alias_table.hpp"
#include "pack.h"
#include <algorithm>
#include <cstdio>

namespace rth {

void Scene::destroy()
{
  m_gltf = GltfScene{};
  m_primMeshes.clear(); m_vertices.clear(); m_instances.clear(); m_materials.clear(); m_textures.clear();
  m_puncLights.clear(); m_trigLights.clear();
  m_lightBufInfo = rt_light_buf_info{};
  m_puncLightWeight = m_trigLightWeight = 0.f;
  m_stats = SceneStats{};
}

bool Scene::load(const std::string& filename)  // scene.cpp:57-125
{
  GltfScene g;
  std::string err;
  if(!loadGltfFile(filename, g, err)) { fprintf(stderr, "Scene::load: %s\n", err.c_str()); return false; }
  size_t slash = filename.find_last_of("/\\"), dot = filename.find_last_of('.');
  std::string stem = filename.substr(slash == std::string::npos ? 0 : slash + 1, dot == std::string::npos ? std::string::npos : dot - (slash == std::string::npos ? 0 : slash + 1));
  return loadFromGltfScene(std::move(g), stem);
}

bool Scene::loadFromGltfScene(GltfScene&& gltf, const std::string& name)
{
  destroy();
  m_gltf = std::move(gltf);
  m_sceneName = name;
  GltfScene& g = m_gltf;
  if(g.positions.empty() || g.primMeshes.empty() || g.nodes.empty()) return false;
  if(g.materials.empty()) g.materials.push_back(GltfMaterial{});
  const size_t nv = g.positions.size();
  // importDrawableNodes(..., Normal | Texcoord_0 | Tangent | Color_0) fills missing attributes (scene.cpp:73-74).
  // nvpro_core is not vendored; the fill rules below are this build's (DESIGN.md §Scene ingest).
  if(g.normals.size() != nv) {
    g.normals.assign(nv, V3{0, 0, 0});
    for(const GltfPrimMesh& pm : g.primMeshes)
      for(uint32_t i = 0; i + 2 < pm.indexCount; i += 3) {
        uint32_t a = pm.vertexOffset + g.indices[pm.firstIndex + i], b = pm.vertexOffset + g.indices[pm.firstIndex + i + 1], c = pm.vertexOffset + g.indices[pm.firstIndex + i + 2];
        V3 n = cross(g.positions[b] - g.positions[a], g.positions[c] - g.positions[a]);
        g.normals[a] = g.normals[a] + n; g.normals[b] = g.normals[b] + n; g.normals[c] = g.normals[c] + n;
      }
    for(V3& n : g.normals) n = length(n) > 0 ? normalize(n) : V3{0, 1, 0};
  }
  if(g.texcoords0.size() != nv) g.texcoords0.assign(nv, {0.f, 0.f});
  if(g.colors0.size() != nv) g.colors0.assign(nv, {1.f, 1.f, 1.f, 1.f});
  if(g.tangents.size() != nv) {
    g.tangents.resize(nv);
    for(size_t i = 0; i < nv; i++) {
      V3 n = g.normals[i];
      V3 c1 = cross(n, V3{0, 0, 1}), c2 = cross(n, V3{0, 1, 0});
      V3 t = normalize(length(c1) > length(c2) ? c1 : c2);
      g.tangents[i] = {t.x, t.y, t.z, 1.f};
    }
  }
  g.bboxMin = V3{3e38f, 3e38f, 3e38f}; g.bboxMax = V3{-3e38f, -3e38f, -3e38f};
  for(const GltfNode& nd : g.nodes) {
    const GltfPrimMesh& pm = g.primMeshes[nd.primMesh];
    for(uint32_t v = 0; v < pm.vertexCount; v++) {
      V3 p = xformPoint(nd.worldMatrix, g.positions[pm.vertexOffset + v]);
      g.bboxMin = {std::min(g.bboxMin.x, p.x), std::min(g.bboxMin.y, p.y), std::min(g.bboxMin.z, p.z)};
      g.bboxMax = {std::max(g.bboxMax.x, p.x), std::max(g.bboxMax.y, p.y), std::max(g.bboxMax.z, p.z)};
    }
  }
  // setCameraFromScene, scene.cpp:295-314
  if(!g.cameras.empty()) setCamera(g.cameras[0].eye, g.cameras[0].center, g.cameras[0].up, g.cameras[0].yfovDeg);
  else fitCamera();
  m_camera = rt_scene_camera{};
  m_camera.nbLights = int(g.lights.size());
  m_cameraInit = false;
  m_lastEye = V3{0, 0, 0};

  // scene.cpp:94-103
  createMaterialBuffer();
  createPuncLightBuffer();
  m_textures.clear();
  if(g.textures.empty()) { TextureImage w; w.bgra = {255, 255, 255, 255}; g.textures.push_back(w); }  // "cannot be empty" default (scene.cpp:575-583)
  for(const TextureImage& t : g.textures) m_textures.push_back(rt_texture{t.bgra.data(), t.width, t.height, t.wrapS, t.wrapT, t.magFilter, 0});
  createVertexBuffer();
  createInstanceDataBuffer();
  createTrigLightBuffer();
  if(m_lightBufInfo.puncLightSize > 0 || m_lightBufInfo.trigLightSize > 0)
    m_lightBufInfo.trigSampProb = m_trigLightWeight / (m_trigLightWeight + m_puncLightWeight);

  m_stats = SceneStats{};
  for(const GltfPrimMesh& pm : g.primMeshes) m_stats.triangles += pm.indexCount / 3;
  for(const GltfNode& nd : g.nodes) m_stats.instancedTriangles += g.primMeshes[nd.primMesh].indexCount / 3;
  m_stats.vertices = nv; m_stats.primMeshes = uint32_t(g.primMeshes.size()); m_stats.nodes = uint32_t(g.nodes.size());
  m_stats.materials = uint32_t(g.materials.size()); m_stats.textures = uint32_t(g.textures.size());
  m_stats.puncLights = m_lightBufInfo.puncLightSize; m_stats.trigLights = m_lightBufInfo.trigLightSize;
  return true;
}

void Scene::fitCamera()
{
  const GltfScene& g = m_gltf;
  V3 c = (g.bboxMin + g.bboxMax) * 0.5f;
  float r = length(g.bboxMax - g.bboxMin) * 0.5f;
  m_fov = 45.f;
  float d = r / std::tan(m_fov * 0.5f * 3.14159265f / 180.f);
  setCamera(c + V3{0, 0, d}, c, V3{0, 1, 0}, m_fov);
}

void Scene::createMaterialBuffer()  // scene.cpp:415-448
{
  m_materials.clear();
  for(const GltfMaterial& m : m_gltf.materials) {
    rt_material s{};
    s.pbrBaseColorFactor = rt_vec4{m.baseColorFactor[0], m.baseColorFactor[1], m.baseColorFactor[2], m.baseColorFactor[3]};
    s.pbrBaseColorTexture = m.baseColorTexture;
    s.pbrMetallicFactor = m.metallicFactor;
    s.pbrRoughnessFactor = m.roughnessFactor;
    s.pbrMetallicRoughnessTexture = m.metallicRoughnessTexture;
    s.emissiveTexture = m.emissiveTexture;
    s.emissiveFactor = rt_vec3{m.emissiveFactor[0], m.emissiveFactor[1], m.emissiveFactor[2]};
    s.normalTexture = m.normalTexture;
    s.normalTextureScale = m.normalTextureScale;
    s.transmissionFactor = m.transmissionFactor;
    s.transmissionTexture = m.transmissionTexture;
    s.ior = std::min(std::max(m.ior, 1.f), RT_MAX_IOR_MINUS_ONE + 1.f);
    s.alphaMode = m.alphaMode;
    s.alphaCutoff = m.alphaCutoff;
    m_materials.push_back(s);
  }
}

void Scene::createPuncLightBuffer()  // scene.cpp:319-353 + createPuncLightImptSampAccel :700-726
{
  m_puncLights.clear();
  for(const GltfLight& l : m_gltf.lights) {
    rt_punc_light p{};
    p.position = R3(xformPoint(l.worldMatrix, V3{0, 0, 0}));
    p.direction = R3(xformDir(l.worldMatrix, V3{0, 0, -1}));
    p.color = rt_vec3{l.color[0], l.color[1], l.color[2]};
    p.innerConeCos = std::cos(l.innerConeAngle);
    p.outerConeCos = std::cos(l.outerConeAngle);
    p.range = l.range;
    p.intensity = l.intensity;
    p.type = l.type;
    m_puncLights.push_back(p);
  }
  m_lightBufInfo.puncLightSize = uint32_t(m_puncLights.size());
  m_puncLightWeight = 0.f;
  if(!m_puncLights.empty()) {
    std::vector<float> distrib;
    float total = 0.f;
    for(const rt_punc_light& l : m_puncLights) {
      float power = luminance(&l.color.x) * l.intensity * 3.1416f * 4.f;
      distrib.push_back(power);
      total += power;
    }
    std::vector<AliasBucket> table = buildAliasTable(distrib);
    for(size_t i = 0; i < distrib.size(); i++)
      m_puncLights[i].impSamp = rt_impt_samp{table[i].failId, table[i].prob, distrib[i] / total, distrib[size_t(table[i].failId)] / total};
    m_puncLightWeight = total;
  }
}

void Scene::createVertexBuffer()  // scene.cpp:209-289
{
  const GltfScene& g = m_gltf;
  m_vertices.resize(g.positions.size());
  for(size_t i = 0; i < g.positions.size(); i++) {
    rt_vertex v{};
    v.position = R3(g.positions[i]);
    v.normal = compressUnitVec(g.normals[i].x, g.normals[i].y, g.normals[i].z);
    v.tangent = compressUnitVec(g.tangents[i][0], g.tangents[i][1], g.tangents[i][2]);
    v.texcoord = rt_vec2{g.texcoords0[i][0], g.texcoords0[i][1]};
    v.color = packUnorm4x8(g.colors0[i][0], g.colors0[i][1], g.colors0[i][2], g.colors0[i][3]);
    uint32_t bits = floatBits(v.texcoord.y);  // tangent handedness in the LSB of V (scene.cpp:249-257)
    if(g.tangents[i][3] > 0) bits |= 1u; else bits &= ~1u;
    v.texcoord.y = bitsFloat(bits);
    m_vertices[i] = v;
  }
  m_primMeshes.clear();
  for(const GltfPrimMesh& pm : g.primMeshes) m_primMeshes.push_back(rt_prim_mesh{pm.vertexOffset, pm.vertexCount, pm.firstIndex, pm.indexCount, pm.materialIndex});
}

void Scene::createInstanceDataBuffer()  // scene.cpp:179-195 (InstanceData rows == m_primMeshes) + accelstruct.cpp:132-162
{
  const GltfScene& g = m_gltf;
  m_instances.clear();
  for(const GltfNode& nd : g.nodes) {
    rt_instance in{};
    for(int r = 0; r < 3; r++) for(int c = 0; c < 4; c++) in.objectToWorld[r * 4 + c] = nd.worldMatrix.at(r, c);
    in.primMesh = uint32_t(nd.primMesh);
    const GltfMaterial& mat = g.materials[size_t(std::max(0, g.primMeshes[nd.primMesh].materialIndex))];
    if(mat.alphaMode == 0 || (mat.baseColorFactor[3] == 1.0f && mat.baseColorTexture == -1)) in.flags |= RT_INST_FORCE_OPAQUE;
    if(mat.doubleSided == 1) in.flags |= RT_INST_CULL_DISABLE;
    m_instances.push_back(in);
  }
}

void Scene::createTrigLightBuffer()  // scene.cpp:355-409 + createTrigLightImptSampAccel :741-772
{
  const GltfScene& g = m_gltf;
  m_trigLights.clear();
  for(const GltfNode& node : g.nodes) {
    const GltfPrimMesh& pm = g.primMeshes[node.primMesh];
    const GltfMaterial& mat = g.materials[size_t(std::max(0, pm.materialIndex))];
    if(luminance(mat.emissiveFactor) > 1e-2f) {
      for(uint32_t idx = pm.firstIndex; idx + 1 < pm.firstIndex + pm.indexCount; idx += 3) {  // `idx < first+count-1` (quirk 11)
        rt_trig_light t{};
        uint32_t i0 = g.indices[idx] + pm.vertexOffset, i1 = g.indices[idx + 1] + pm.vertexOffset, i2 = g.indices[idx + 2] + pm.vertexOffset;
        t.transformIndex = 0xffffffffu;  // `transforms.size()-1` on an empty vector (quirk 11); unused by the shaders
        t.matIndex = uint32_t(pm.materialIndex);
        t.v0 = R3(xformPoint(node.worldMatrix, g.positions[i0])); t.uv0 = rt_vec2{g.texcoords0[i0][0], g.texcoords0[i0][1]};
        t.v1 = R3(xformPoint(node.worldMatrix, g.positions[i1])); t.uv1 = rt_vec2{g.texcoords0[i1][0], g.texcoords0[i1][1]};
        t.v2 = R3(xformPoint(node.worldMatrix, g.positions[i2])); t.uv2 = rt_vec2{g.texcoords0[i2][0], g.texcoords0[i2][1]};
        m_trigLights.push_back(t);
      }
    }
  }
  m_trigLightWeight = 0.f;
  if(!m_trigLights.empty()) {
    std::vector<float> distrib;
    float total = 0.f;
    for(const rt_trig_light& t : m_trigLights) {
      float power = luminance(g.materials[t.matIndex].emissiveFactor);  // not area-weighted (scene.cpp:750-756)
      distrib.push_back(power);
      total += power;
    }
    std::vector<AliasBucket> table = buildAliasTable(distrib);
    for(size_t i = 0; i < m_trigLights.size(); i++)
      m_trigLights[i].impSamp = rt_impt_samp{table[i].failId, table[i].prob, distrib[i] / total, distrib[size_t(table[i].failId)] / total};
    m_trigLightWeight = total;
  }
  m_lightBufInfo.trigLightSize = uint32_t(m_trigLights.size());
}

void Scene::updateCamera(int width, int height)  // scene.cpp:777-795
{
  const float aspect = float(width) / float(height);
  const M4 view = lookAt(m_eye, m_center, m_up);
  M4 proj = perspectiveVK(m_fov, aspect, 0.001f, 1000.0f);  // CAMERA_NEAR / CAMERA_FAR, host_device.h:151-152
  proj.at(0, 2) += .5f / float(width);                      // fixed half-pixel jitter (scene.cpp:783-787)
  proj.at(1, 2) += .5f / float(height);
  M4 prevViewInverse, prevProjView;
  memcpy(prevViewInverse.m, m_camera.viewInverse.m, sizeof(prevViewInverse.m));
  memcpy(prevProjView.m, m_camera.projView.m, sizeof(prevProjView.m));
  m_camera.lastProjView = toRt(prevProjView);
  m_camera.lastView = toRt(invert(prevViewInverse));
  m_camera.viewInverse = toRt(invert(view));
  m_camera.projInverse = toRt(invert(proj));
  m_camera.projView = toRt(proj * view);
  m_camera.lastPosition = R3(m_lastEye);  // `static eye` starts at 0 (scene.cpp:780)
  m_lastEye = m_eye;
  m_cameraInit = true;
}

rt_scene_desc Scene::getDesc(const HdrSampling* env) const
{
  rt_scene_desc d{};
  d.numPrimMeshes = uint32_t(m_primMeshes.size()); d.primMeshes = m_primMeshes.data();
  d.numVertices = m_vertices.size(); d.vertices = m_vertices.data();
  d.numIndices = m_gltf.indices.size(); d.indices = m_gltf.indices.data();
  d.numInstances = uint32_t(m_instances.size()); d.instances = m_instances.data();
  d.numMaterials = uint32_t(m_materials.size()); d.materials = m_materials.data();
  d.numTextures = uint32_t(m_textures.size()); d.textures = m_textures.data();
  d.puncLights = m_puncLights.empty() ? nullptr : m_puncLights.data();
  d.trigLights = m_trigLights.empty() ? nullptr : m_trigLights.data();
  d.lightInfo = m_lightBufInfo;
  if(env && env->width() > 0) {
    d.envWidth = env->width(); d.envHeight = env->height();
    d.envRgba32f = env->pixels().data(); d.envAccel = env->accel().data();
  }
  return d;
}

}  // namespace rth
