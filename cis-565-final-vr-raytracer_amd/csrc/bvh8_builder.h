#pragma once
#include <vector>
#include "bvh8.h"
#include "dev_scene.h"

namespace rt {

struct BuildInput {
  const rt_scene_desc* scene;
};
struct BuildOutput {
  std::vector<Node8> nodes;
  std::vector<Tri48> tris;       // BVH leaf order
  std::vector<TriRef> triRef;    // by globalId
  std::vector<DevInstance> instances;
  int maxDepth = 0;
  double sahCost = 0;
  double sahNodeStepsQ = 0, sahTriStepsQ = 0;  // the same with the 8-bit quantised child boxes the traversal tests
  double sahNodeSteps = 0, sahTriSteps = 0;   // SAH expectation of node / triangle steps of a ray that hits the root box (wide tree, build-time boxes)
  uint64_t references = 0;                   // leaf records (= tris.size()): > the triangle count when spatial splits duplicated references
  uint64_t spatialSplits = 0, rotations = 0, reinsertions = 0;
  float pad = 0;
};
// Host-side build: flatten (instance, triangle) pairs to world space, binned-SAH BVH2, greedy collapse to 8-wide,
// octant-ordered slot assignment, conservative 8-bit quantisation.  Replaces AccelStructure::create
// (src/accelstruct.cpp:55-162: BLAS per prim mesh + TLAS per node, built by the Vulkan driver).
// plainTree: object splits only, no rotations / reinsertion — the tree of rounds 1-4, whatever the environment says (rt_build_accel's fallback when the quality passes
// made a tree deeper than the traversal stack).
bool buildBvh8(const rt_scene_desc& scene, BuildOutput& out, int threads, bool plainTree = false);

}  // namespace rt
