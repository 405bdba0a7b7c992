// ref_ctx.h — what one dispatch of a reference shader is bound to (TEST INFRASTRUCTURE; included inside namespace glsl after the
// reference's own host_device.h, so the struct types below ARE the reference's: shaders/host_device.h:153-333).
// The names are the resource names of shaders/layouts.glsl:38-75 and of the push-constant block every .comp declares.
#pragma once
struct RefCtx {
  const orc::Scene* scene;
  RtxState rtxState;                 // push constant (re-pushed with denoiseLevel per pass, renderer.cpp:183,196)
  SceneCamera sceneCamera;
  SunAndSky sunAndSky;
  LightBufInfo lightBufInfo;
  InstanceData* geoInfo;
  GltfShadeMaterial* materials;
  PuncLight* puncLights;
  TrigLight* trigLights;
  ImptSampData* envSamplingData;
  sampler2D* texturesMap;
  sampler2D environmentTexture;
  image2D lastDirectResultImage, lastIndirectResultImage, thisDirectResultImage, thisIndirectResultImage;
  uimage2D lastGbuffer, thisGbuffer;
  iimage2D motionVector;
  DirectReservoir *lastDirectResv, *thisDirectResv, *tempDirectResv;
  IndirectReservoir *lastIndirectResv, *thisIndirectResv, *tempIndirectResv;
  image2D denoiseDirTempA, denoiseDirTempB, denoiseIndTempA, denoiseIndTempB;
};
