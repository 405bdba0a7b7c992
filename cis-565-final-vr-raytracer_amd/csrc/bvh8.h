// bvh8.h — flat BVH8 in HBM: the replacement for VK_KHR_acceleration_structure (src/accelstruct.cpp:55-162).
//
// One level, world space: every (TLAS instance, triangle) pair of the reference becomes one record.
// Node = 80 B compressed wide node (8 children, 8-bit quantised child boxes on a per-node power-of-two grid,
// after Ylitie, Karras, Laine, "Efficient Incoherent Ray Traversal on GPUs Through Compressed Wide BVHs",
// HPG 2017): 5 x 16 B loads per node visit.  Triangle = 64 B (v0, e1, e2, ids, 8x8 opacity micro-map): 4 x 16 B loads per test.
#pragma once
#include <cstdint>

namespace rt {

struct alignas(16) Node8 {
  float px, py, pz;    // quantisation origin = node box minimum
  uint8_t ex, ey, ez;  // per-axis grid step 2^(e-127) (IEEE exponent byte)
  uint8_t imask;       // bit s set: slot s holds an internal child
  uint32_t childBase;  // index of the first internal child; children are stored in slot order
  uint32_t triBase;    // index of this node's first leaf triangle
  uint8_t meta[8];     // internal: 0b001'11sss (sss = slot); leaf: unary count (1..3) << 5 | first triangle offset; empty: 0
  uint8_t qlox[8], qloy[8], qloz[8];
  uint8_t qhix[8], qhiy[8], qhiz[8];
};
static_assert(sizeof(Node8) == 80, "Node8 must be 80 bytes");

enum : uint32_t { TRI_OPAQUE = 1u, TRI_NOCULL = 2u, TRI_FLIP = 4u };

struct alignas(16) Tri48 {   // (name kept from the first 48-byte layout; the record is 64 B since the opacity map moved in)
  float v0x, v0y, v0z;
  float e1x, e1y, e1z;  // v1 - v0
  float e2x, e2y, e2z;  // v2 - v0
  uint32_t globalId;    // index in (instance, primitive) order: the tie-break key and the key into triRef[]
  uint32_t flags;       // TRI_*
  uint32_t alphaIdx;    // index into alphaRec[] for triangles that need the alpha test (TRI_OPAQUE clear), else 0
  // Opacity micro-map: 8x8 cells over the barycentrics (u,v), 2 bits each: 0 = unknown (run HitTest on the texture),
  // 1 = HitTest accepts everywhere in the cell, 2 = opacity is 0 everywhere in the cell.  Built conservatively from the
  // alpha channel (rt_api.cpp buildOpacityMap), so resolving a candidate from it gives exactly HitTest's answer without
  // the two dependent gathers (AlphaRec, texels) — the cost that dominated foliage rays.
  uint32_t omm[4];
};
static_assert(sizeof(Tri48) == 64, "triangle record must be 64 bytes");

// Everything HitTest (traceray_rq.glsl:32-102) needs for one non-opaque triangle, gathered at build time so the
// alpha test is one 64 B record + the texel fetches instead of triRef -> instance -> primMesh -> indices -> 3 vertices ->
// material -> texture descriptor (7 dependent gathers per candidate; foliage rays test tens of candidates).
struct alignas(16) AlphaRec {
  float uv0x, uv0y, uv1x, uv1y, uv2x, uv2y;  // raw vertex texcoords (HitTest does not strip the handedness bit)
  float baseAlpha;                            // pbrBaseColorFactor.a
  float cutoff;
  const uint8_t* bgra;                        // nullptr: no base colour texture
  int32_t w, h, wrapS, wrapT, filter;
  int32_t alphaMode;
};
static_assert(sizeof(AlphaRec) == 64, "AlphaRec must be 64 bytes");

// globalId -> (instance, primitive)
struct TriRef { uint32_t inst, prim; };

}  // namespace rt
