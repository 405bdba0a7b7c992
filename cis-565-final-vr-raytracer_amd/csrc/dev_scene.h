// dev_scene.h — device-resident scene (HBM): what the reference binds as descriptor sets S_ACCEL / S_SCENE / S_ENV
// (shaders/host_device.h:67-110, shaders/layouts.glsl:38-55) as one POD of raw pointers passed by value to every kernel.
#pragma once
#include <hip/hip_runtime.h>
#include "bvh8.h"
#include "../../include/rt_abi.h"

namespace rt {

struct DevInstance {   // TLAS instance + the two matrices rayQueryGetIntersection{ObjectToWorld,WorldToObject}EXT return
  float o2w[12];
  float w2o[12];
  uint32_t primMesh;   // instanceCustomIndex
  uint32_t flags;
  uint32_t pad[2];
};

struct DevTexture {
  const uint8_t* bgra;
  int32_t w, h, wrapS, wrapT, filter, pad;
};

struct DevScene {
  const Node8* nodes;
  const Tri48* tris;
  const TriRef* triRef;
  const AlphaRec* alphaRec;         // indexed by Tri48::alphaIdx
  const AlphaRec* alphaByTri;       // the same records indexed by TRIANGLE index (zero for opaque triangles): the latency build fetches record and AlphaRec together
  const DevInstance* instances;
  const rt_prim_mesh* primMeshes;   // geoInfo[] (InstanceData rows): vertex/index offsets + materialIndex
  const rt_vertex* vertices;
  const uint32_t* indices;
  const rt_material* materials;
  const DevTexture* textures;
  const rt_punc_light* puncLights;
  const rt_trig_light* trigLights;
  const float* env;                 // RGBA32F
  const rt_impt_samp* envAccel;
  rt_light_buf_info lightInfo;
  int32_t envW, envH;
  uint32_t numTris, numNodes;
  const struct SkyPre* sky;         // != nullptr: procedural sun & sky replaces the HDR map (_sunAndSky.in_use == 1)
  int32_t stackEntries;             // traversal stack entries per lane held in LDS (8 B each); entries beyond that live in stackOvf
  int32_t coopLive;                 // traversal: cooperative triangle steps when at most this many rays of a wave are live (0 = off)
  float triPad;                     // box padding of the build (2e-5 x largest |coordinate|): an accepted hit point lies inside its triangle's padded box
  int32_t stackTotal;               // stack entries a ray of this tree can need (multiple of 4, > max depth); stackTotal - stackEntries per thread sit in HBM
  // overflow part of the traversal stacks: (stackTotal - stackEntries) entries per thread of a launch, thread = blockIdx.x * blockDim.x + threadIdx.x.
  // Two areas, because a direct-kind and an indirect-kind kernel can be in flight together; the launchers point stackOvf at the one of their stage.
  uint2* stackOvf; uint2* stackOvfInd;
  uint32_t stackOvfThreads;
  int32_t gangMax;                  // latency build: at most this many live rays in a wave whose pool is dry -> its idle ray slots work for them (traverse.h gangTail); 0 = off
};

// per-pixel scratch record (internal; never crosses the ABI)
struct SurfRec {   // what k_direct_spatial needs from the primary hit's shading state: 64 B
  rt_vec3 position, normal, ffnormal, emission;
  float roughness, metallic;
  uint32_t matID, seed;
};

// per-frame screen-space state (renderer.cpp:227-302), "this"/"last" already resolved from the frame parity
struct DevFrame {
  uint4* thisG; const uint4* lastG;
  short2* motion;
  rt_direct_reservoir* thisDirectResv; const rt_direct_reservoir* lastDirectResv;
  rt_indirect_reservoir* thisIndirectResv; const rt_indirect_reservoir* lastIndirectResv;
  uint32_t* thisLightId; const uint32_t* lastLightId;
  float4* thisDirectResult; float4* thisIndirectResult;
  float4* denoiseDirA; float4* denoiseDirB; float4* denoiseIndA; float4* denoiseIndB;
  unsigned long long* counters;     // 6 x u64 (rt_counters order) or nullptr
  // scratch (ctx-owned, sized by rt_resize): per-pixel records between k_direct_stage and k_direct_spatial (spatial reuse modes), the
  // indirect stage's tile-list counters + the history-miss flags (qcount), per-wave profile records of measurement builds (RT_WAVEPROF)
  SurfRec* surf; uint32_t* status; uint32_t* qcount; uint32_t* waveProf;
  // A-Trous geometry decoded once per frame: (normal.xyz, matHash bits) and (world position.xyz, 0); full-res and half-res grids
  float4* geomN; float4* geomP; float4* geomNh; float4* geomPh;
  rt_direct_reservoir* tempDirectResv;  // RT_BUF_DIRECT_RESV_TEMP: cacheTempReservoir target of the spatial reuse (direct_stage.comp:127-129)
  double* postRowSums; float* postMean;  // rt_tonemap: per-row colour sums [2][H][3], image means [2][4]
  float4* postMipD; float4* postMipI;    // rt_tonemap: mip levels 1..7 of the two result images (toneLocalExposure)
  uint32_t* tileOrder;              // 8 per-XCD lists of half-res tile ids, longest (multi-bounce) first
  int32_t W, H;
  // rows of the LAST-frame buffers that are valid on this GPU (row-tiled multi-GPU: own band + received halos).  A temporal
  // lookup that lands inside the image but outside [histRow0, histRow1) raises *histMiss so the host can fetch the full
  // history and redo the frame; single-GPU: [0, H) and the flag never fires.
  int32_t histRow0, histRow1;
  uint32_t* histMiss;
  int32_t stackLds;                 // traversal stack entries per lane kept in LDS by this frame's traced launches (0 = the whole stack); the rest sits in DevScene::stackOvf
  int32_t pad4;
  // Direct stage, full-frame launches (round 5): rowCost[tile row] = wave cycles the row's tiles took in the LAST direct-stage launch (>> 8, summed by atomics, one
  // per wave); k_row_order turns it into rowOrder = the tile rows by descending cost, and the launch deals them to the XCDs in that order: the horizon rows, whose
  // rays cross every tree of the street and whose waves end the launch, start first.  nullptr: identity (row bands, the first frame's all-zero costs sort to identity).
  uint32_t* rowCost; uint16_t* rowOrder;
};

}  // namespace rt
