// ref_tu.cpp — one translation unit per reference compute shader (TEST INFRASTRUCTURE, authoring container only).
// Built by build_ref.sh as   g++ -DREF_SHADER='"direct_stage.comp"' -DREF_ENTRY=ref_run_direct_stage [-DREF_HAS_PRD] ref_tu.cpp
// with -I pointing at the scratch directory that holds glsl2cpp.py's rewrite of /root/reference/shaders.  The shader text is
// compiled as it stands there, inside an anonymous namespace (every .comp defines main(), raySpawn(), ... of its own).
#include "glsl_cpu.h"
namespace glsl {
#include "host_device.h"
#include "ref_ctx.h"
namespace {
#include REF_SHADER

// descriptor-set binding: point the shader's resource globals (layouts.glsl) at this dispatch's buffers
static void ref_bind(const RefCtx& c)
{
  topLevelAS.scene = c.scene;
  lastDirectResultImage = c.lastDirectResultImage; lastIndirectResultImage = c.lastIndirectResultImage;
  thisDirectResultImage = c.thisDirectResultImage; thisIndirectResultImage = c.thisIndirectResultImage;
  geoInfo = c.geoInfo; sceneCamera = c.sceneCamera; materials = c.materials; puncLights = c.puncLights; trigLights = c.trigLights;
  lightBufInfo = c.lightBufInfo; texturesMap = c.texturesMap;
  _sunAndSky = c.sunAndSky; environmentTexture = c.environmentTexture; envSamplingData = c.envSamplingData;
  lastGbuffer = c.lastGbuffer; thisGbuffer = c.thisGbuffer; motionVector = c.motionVector;
  lastDirectResv = c.lastDirectResv; thisDirectResv = c.thisDirectResv; tempDirectResv = c.tempDirectResv;
  lastIndirectResv = c.lastIndirectResv; thisIndirectResv = c.thisIndirectResv; tempIndirectResv = c.tempIndirectResv;
  denoiseDirTempA = c.denoiseDirTempA; denoiseDirTempB = c.denoiseDirTempB; denoiseIndTempA = c.denoiseIndTempA; denoiseIndTempB = c.denoiseIndTempB;
  rtxState = c.rtxState;
#ifdef REF_HAS_PRD
  rq_seed = &prd.seed;
#else
  rq_seed = nullptr;
#endif
}
}  // namespace
}  // namespace glsl

// vkCmdDispatch(groupsX, groupsY, 1) with local size 8 x 8 (host_device.h:31-38): workgroups in order, the 64 invocations of a
// workgroup one after another in gl_LocalInvocationIndex order, each to completion (barrier() is a no-op; thread 0 runs first,
// which is all indirect_stage.comp's `shared bool multiBounce` needs).
extern "C" void REF_ENTRY(const glsl::RefCtx* c, int groupsX, int groupsY)
{
  using namespace glsl;
  ref_bind(*c);
  for(int gy = 0; gy < groupsY; gy++)
    for(int gx = 0; gx < groupsX; gx++) {
      gl_WorkGroupID = uvec3(gx, gy, 0);
      for(int ly = 0; ly < 8; ly++)
        for(int lx = 0; lx < 8; lx++) {
          gl_LocalInvocationID = uvec3(lx, ly, 0);
          gl_LocalInvocationIndex = uint(ly * 8 + lx);
          gl_GlobalInvocationID = uvec3(gx * 8 + lx, gy * 8 + ly, 0);
          shader_main();
        }
    }
  rq_seed = nullptr;
}
