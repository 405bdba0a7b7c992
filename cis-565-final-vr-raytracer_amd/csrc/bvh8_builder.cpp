// bvh8_builder.cpp — host build of the flat BVH8 (see bvh8.h).  Replaces the driver-side BLAS/TLAS build of
// src/accelstruct.cpp:110-162.  Box culling is made strictly weaker than the triangle test by padding every
// triangle box with 2e-5 x (largest |coordinate|) before anything else (DESIGN.md §Traversal soundness), so the
// closest hit is a function of the triangle set only, never of the tree.
#include "bvh8_builder.h"
#include "dev_math.h"
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <thread>

namespace rt {
namespace {

struct Box {
  float lo[3], hi[3];
  void reset() { for(int a = 0; a < 3; a++) { lo[a] = 3e38f; hi[a] = -3e38f; } }
  void grow(const Box& b) { for(int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], b.lo[a]); hi[a] = std::max(hi[a], b.hi[a]); } }
  void grow(const float* p) { for(int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], p[a]); hi[a] = std::max(hi[a], p[a]); } }
  float area() const { float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2]; return (dx < 0) ? 0.f : 2.f * (dx * dy + dy * dz + dz * dx); }
};
struct Prim { Box b; float c[3]; };
struct N2 { Box b; uint32_t a, n; bool leaf; };  // internal: children a, a+1; leaf: prims [a, a+n)

struct Builder2 {
  const std::vector<Prim>& prims;
  std::vector<uint32_t>& idx;
  std::vector<N2> nodes;
  std::atomic<uint32_t> nodeCount{1};
  std::atomic<int> liveThreads{0};
  int maxThreads;
  Builder2(const std::vector<Prim>& p, std::vector<uint32_t>& i, int threads) : prims(p), idx(i), nodes(std::max<size_t>(2, 2 * p.size() + 2)), maxThreads(threads) {}

  void build(uint32_t node, uint32_t b, uint32_t e)
  {
    for(;;) {
      N2& N = nodes[node];
      N.b.reset();
      Box cb; cb.reset();
      for(uint32_t k = b; k < e; k++) { N.b.grow(prims[idx[k]].b); cb.grow(prims[idx[k]].c); }
      const uint32_t cnt = e - b;
      auto makeLeaf = [&] { N.leaf = true; N.a = b; N.n = cnt; };
      if(cnt <= 1) { makeLeaf(); return; }
      // binned SAH over the three axes
      constexpr int NBMAX = 64;
      static const int envNB = getenv("RESTIR_BVH_BINS") ? std::min(NBMAX, std::max(4, atoi(getenv("RESTIR_BVH_BINS")))) : 16;
      static const float leafSlotCost = getenv("RESTIR_BVH_SLOTCOST") ? float(atof(getenv("RESTIR_BVH_SLOTCOST"))) : 0.25f;  // measured: 0.25 beats 0.5 by 2 % on the Bistro-class scene, bins 16 vs 32 vs 64 make no difference
      const int NB = envNB;
      float best = 3e38f; int bestAxis = -1, bestBin = 0;
      for(int ax = 0; ax < 3; ax++) {
        float ext = cb.hi[ax] - cb.lo[ax];
        if(!(ext > 0)) continue;
        Box bb[NBMAX]; uint32_t bc[NBMAX];
        for(int i = 0; i < NB; i++) { bb[i].reset(); bc[i] = 0; }
        const float k1 = NB * (1.f - 1e-6f) / ext;
        for(uint32_t k = b; k < e; k++) {
          const Prim& p = prims[idx[k]];
          int bi = std::min(NB - 1, std::max(0, int((p.c[ax] - cb.lo[ax]) * k1)));
          bb[bi].grow(p.b); bc[bi]++;
        }
        float ra[NBMAX]; uint32_t rc[NBMAX];
        Box acc; acc.reset(); uint32_t c = 0;
        for(int i = NB - 1; i > 0; i--) { acc.grow(bb[i]); c += bc[i]; ra[i] = acc.area(); rc[i] = c; }
        acc.reset(); c = 0;
        for(int i = 0; i < NB - 1; i++) {
          acc.grow(bb[i]); c += bc[i];
          if(c == 0 || rc[i + 1] == 0) continue;
          float cost = acc.area() * c + ra[i + 1] * rc[i + 1];
          if(cost < best) { best = cost; bestAxis = ax; bestBin = i; }
        }
      }
      const float pa = std::max(N.b.area(), 1e-30f);
      if(cnt <= 3) {
        // leaf of <= 3 triangles unless splitting is clearly cheaper (one wide-node slot costs a fraction of a triangle test)
        if(bestAxis < 0 || leafSlotCost * pa + best >= float(cnt) * pa) { makeLeaf(); return; }
      }
      uint32_t mid;
      if(bestAxis >= 0) {
        const float ext = cb.hi[bestAxis] - cb.lo[bestAxis], k1 = NB * (1.f - 1e-6f) / ext, lo = cb.lo[bestAxis];
        const int ax = bestAxis, bin = bestBin;
        auto it = std::partition(idx.begin() + b, idx.begin() + e, [&](uint32_t i) {
          int bi = std::min(NB - 1, std::max(0, int((prims[i].c[ax] - lo) * k1)));
          return bi <= bin;
        });
        mid = uint32_t(it - idx.begin());
      } else {
        mid = b;
      }
      if(mid == b || mid == e) mid = b + cnt / 2;  // identical centroids: split by index
      const uint32_t child = nodeCount.fetch_add(2);
      N.leaf = false; N.a = child; N.n = 0;
      // recurse: big subtrees on their own thread while the budget lasts
      if(cnt > 100000 && liveThreads.load() < maxThreads) {
        liveThreads++;
        std::thread th([this, child, b, mid] { build(child, b, mid); liveThreads--; });
        build(child + 1, mid, e);
        th.join();
        return;
      }
      build(child, b, mid);
      node = child + 1; b = mid;  // tail-iterate on the right half
    }
  }
};

inline int slotSign(int slot, int axis) { return (slot >> axis) & 1 ? 1 : -1; }

}  // namespace

// 3x4 affine inverse: adjugate / determinant for the 3x3 block, then -inv3*t  (same expression order as DESIGN.md §Numerics)
static void inverseAffine(const float* M, float* R, float* detOut)
{
  float a = M[0], b = M[1], c = M[2], d = M[4], e = M[5], f = M[6], g = M[8], h = M[9], i = M[10];
  float A = e * i - f * h, B = f * g - d * i, C = d * h - e * g;
  float det = (a * A + b * B) + c * C;
  *detOut = det;
  float inv = 1.0f / det;
  R[0] = A * inv; R[1] = (c * h - b * i) * inv; R[2] = (b * f - c * e) * inv;
  R[4] = B * inv; R[5] = (a * i - c * g) * inv; R[6] = (c * d - a * f) * inv;
  R[8] = C * inv; R[9] = (b * g - a * h) * inv; R[10] = (a * e - b * d) * inv;
  float tx = M[3], ty = M[7], tz = M[11];
  R[3] = -((R[0] * tx + R[1] * ty) + R[2] * tz);
  R[7] = -((R[4] * tx + R[5] * ty) + R[6] * tz);
  R[11] = -((R[8] * tx + R[9] * ty) + R[10] * tz);
}

bool buildBvh8(const rt_scene_desc& sc, BuildOutput& out, int threads)
{
  // ---- 1. flatten: one world-space triangle per (instance, primitive); globalId = running index ------------
  out.instances.resize(sc.numInstances);
  size_t total = 0;
  for(uint32_t i = 0; i < sc.numInstances; i++) total += sc.primMeshes[sc.instances[i].primMesh].indexCount / 3;
  std::vector<Tri48> flat(total);
  out.triRef.resize(total);
  size_t g = 0;
  float scale = 1e-3f;
  for(uint32_t i = 0; i < sc.numInstances; i++) {
    const rt_instance& in = sc.instances[i];
    DevInstance& di = out.instances[i];
    memcpy(di.o2w, in.objectToWorld, sizeof(di.o2w));
    float det;
    inverseAffine(di.o2w, di.w2o, &det);
    di.primMesh = in.primMesh; di.flags = in.flags; di.pad[0] = di.pad[1] = 0;
    uint32_t f = 0;
    if(in.flags & RT_INST_FORCE_OPAQUE) f |= TRI_OPAQUE;
    if(in.flags & RT_INST_CULL_DISABLE) f |= TRI_NOCULL;
    if(det < 0.0f) f |= TRI_FLIP;
    const rt_prim_mesh& pm = sc.primMeshes[in.primMesh];
    for(uint32_t p = 0; p < pm.indexCount / 3; p++, g++) {
      const uint32_t* ix = &sc.indices[pm.firstIndex + 3 * p];
      float w[3][3];
      for(int k = 0; k < 3; k++) {
        const rt_vec3& q = sc.vertices[pm.vertexOffset + ix[k]].position;
        xformPointRaw(di.o2w, q.x, q.y, q.z, w[k]);
        for(int a = 0; a < 3; a++) scale = std::max(scale, std::fabs(w[k][a]));
      }
      Tri48& T = flat[g];
      T.v0x = w[0][0]; T.v0y = w[0][1]; T.v0z = w[0][2];
      T.e1x = w[1][0] - w[0][0]; T.e1y = w[1][1] - w[0][1]; T.e1z = w[1][2] - w[0][2];
      T.e2x = w[2][0] - w[0][0]; T.e2y = w[2][1] - w[0][1]; T.e2z = w[2][2] - w[0][2];
      T.globalId = uint32_t(g); T.flags = f; T.alphaIdx = 0; T.omm[0] = T.omm[1] = T.omm[2] = T.omm[3] = 0;
      out.triRef[g] = TriRef{i, p};
    }
  }
  const size_t n = total;
  out.nodes.clear(); out.tris.clear(); out.maxDepth = 0;
  if(n == 0) {  // a single empty node keeps the kernels branch-free
    Node8 e{}; e.ex = e.ey = e.ez = 127;
    out.nodes.push_back(e);
    return true;
  }
  const float pad = 2e-5f * scale;
  out.pad = pad;

  // ---- 2. padded triangle boxes + binned-SAH BVH2 ---------------------------------------------------------------
  std::vector<Prim> prims(n);
  for(size_t i = 0; i < n; i++) {
    const Tri48& T = flat[i];
    const float v[3][3] = {{T.v0x, T.v0y, T.v0z}, {T.v0x + T.e1x, T.v0y + T.e1y, T.v0z + T.e1z}, {T.v0x + T.e2x, T.v0y + T.e2y, T.v0z + T.e2z}};
    Prim& P = prims[i];
    P.b.reset();
    for(int k = 0; k < 3; k++) P.b.grow(v[k]);
    // e1/e2 were rounded when stored: v0+e1 may differ from v1 by an ulp; the pad (>= 300 ulp) absorbs it
    for(int a = 0; a < 3; a++) { P.b.lo[a] -= pad; P.b.hi[a] += pad; P.c[a] = 0.5f * (P.b.lo[a] + P.b.hi[a]); }
  }
  std::vector<uint32_t> idx(n);
  for(size_t i = 0; i < n; i++) idx[i] = uint32_t(i);
  Builder2 B2(prims, idx, std::max(1, threads));
  B2.build(0, 0, uint32_t(n));
  const std::vector<N2>& N = B2.nodes;

  // ---- 3a. which BVH2 nodes become wide nodes: SAH-optimal collapse (Ylitie, Karras, Laine 2017, §3.1) -------------------------
  // cost[n][i] = cheapest way to hang BVH2 subtree n under a parent using at most i of the parent's 8 slots, where a slot holds
  // either a BVH2 leaf (<= 3 triangles, area x triangles x cTri) or a wide node (area x cNode + the cost of ITS 8 slots).
  //   cost[n][1] = A_n cNode + min_k cost[l][k] + cost[r][8-k]          (n becomes a wide node)
  //   cost[n][i] = min(cost[n][i-1], min_k cost[l][k] + cost[r][i-k])   (n dissolves: its children share i slots)
  // The greedy "open the largest child" rule leaves wide nodes 2.8 of 8 slots full on average (636 k nodes for 2.8 M
  // triangles, depth 11, 16.5 node visits per ray); the tree does not change any result, only the step counts.
  // Measured on the 2.8 M-triangle bench scene (scripts/bvh_ab.py, profiles/r02_bvh_collapse_ab.txt): 408 k nodes instead of 636 k, but
  // node visits per ray only 16.49 -> 16.18 (the visits are in the upper and middle levels, not in the under-filled bottom nodes), depth
  // 11 -> 13, direct stage -2 %, indirect stage +4.6 %: no net gain, so the greedy rule stays the default (RESTIR_BVH_COLLAPSE=dp selects this).
  static const bool useDp = getenv("RESTIR_BVH_COLLAPSE") && strcmp(getenv("RESTIR_BVH_COLLAPSE"), "dp") == 0;
  static const float cNode = 2.3f;   // a node step is ~230 instructions,
  static const float cTri = 1.0f;                                                                            // a triangle step ~100
  const uint32_t n2count = B2.nodeCount.load();
  std::vector<float> cost;        // [n][i], i = 1..7 at [n * 8 + i]
  std::vector<uint8_t> choice;    // [n][i]: 0 = single slot (leaf / wide node), 0xff = same as i - 1, else k = slots of the left child
  std::vector<uint8_t> k8;        // [n]: left child's share of the 8 slots when n becomes a wide node
  if(useDp) {
    cost.assign(size_t(n2count) * 8, 0.f); choice.assign(size_t(n2count) * 8, 0); k8.assign(n2count, 1);
    for(uint32_t n = n2count; n-- > 0;) {   // children have larger indices than their parent: this order is bottom-up
      const N2& x = N[n];
      const float A = x.b.area();
      float* c = &cost[size_t(n) * 8];
      uint8_t* ch = &choice[size_t(n) * 8];
      if(x.leaf) { for(int i = 1; i < 8; i++) { c[i] = A * float(x.n) * cTri; ch[i] = 0; } continue; }
      const float* cl = &cost[size_t(x.a) * 8];
      const float* cr = &cost[size_t(x.a + 1) * 8];
      auto dist = [&](int j, int& kBest) { float b = 3e38f; kBest = 1; for(int k = std::max(1, j - 7); k <= std::min(7, j - 1); k++) { const float v = cl[k] + cr[j - k]; if(v < b) { b = v; kBest = k; } } return b; };
      int kb;
      const float d8 = dist(8, kb);
      k8[n] = uint8_t(kb);
      c[1] = A * cNode + d8; ch[1] = 0;
      for(int i = 2; i < 8; i++) {
        const float d = dist(i, kb);
        if(d < c[i - 1]) { c[i] = d; ch[i] = uint8_t(kb); } else { c[i] = c[i - 1]; ch[i] = 0xff; }
      }
    }
  }
  // children of a wide node: follow the recorded choices
  auto expand = [&](auto&& self, uint32_t n, int i, uint32_t* out, int& nc) -> void {
    while(i > 1 && choice[size_t(n) * 8 + i] == 0xff) i--;
    const uint8_t c = (N[n].leaf || i == 1) ? uint8_t(0) : choice[size_t(n) * 8 + i];
    if(c == 0) { out[nc++] = n; return; }
    self(self, N[n].a, int(c), out, nc);
    self(self, N[n].a + 1, i - int(c), out, nc);
  };

  // ---- 3b. build the 8-wide nodes, breadth-first so that a node's internal children are contiguous ------------------
  struct Work { uint32_t n2; uint32_t wide; int depth; };
  std::vector<Work> queue;
  out.nodes.reserve(n / 2 + 16);
  out.tris.reserve(n);
  out.nodes.push_back(Node8{});
  // a root that is itself a leaf gets wrapped by a one-child wide node (handled by the generic path below)
  queue.push_back({0, 0, 1});
  for(size_t qi = 0; qi < queue.size(); qi++) {
    const Work w = queue[qi];
    out.maxDepth = std::max(out.maxDepth, w.depth);
    uint32_t ch[8]; int nc = 0;
    if(N[w.n2].leaf) ch[nc++] = w.n2;
    else if(useDp) {
      // the optimal 8 children of this wide node under the cost model: left subtree in k8 slots, right subtree in the rest
      const int k = k8[w.n2];
      expand(expand, N[w.n2].a, k, ch, nc);
      expand(expand, N[w.n2].a + 1, 8 - k, ch, nc);
    } else {
      ch[nc++] = N[w.n2].a; ch[nc++] = N[w.n2].a + 1;
      for(;;) {  // greedily open the internal child with the largest surface area
        int pick = -1; float bestA = -1.f;
        for(int i = 0; i < nc; i++) if(!N[ch[i]].leaf && N[ch[i]].b.area() > bestA) { bestA = N[ch[i]].b.area(); pick = i; }
        if(pick < 0 || nc == 8) break;
        uint32_t c = ch[pick];
        ch[pick] = N[c].a; ch[nc++] = N[c].a + 1;
      }
    }
    // node box = union of child boxes
    Box nb; nb.reset();
    for(int i = 0; i < nc; i++) nb.grow(N[ch[i]].b);
    // slot assignment: child whose centroid lies furthest towards corner s gets slot s (greedy best-pair)
    int slotOf[8]; bool slotUsed[8] = {false, false, false, false, false, false, false, false}; bool done[8] = {false, false, false, false, false, false, false, false};
    float cen[3] = {0.5f * (nb.lo[0] + nb.hi[0]), 0.5f * (nb.lo[1] + nb.hi[1]), 0.5f * (nb.lo[2] + nb.hi[2])};
    for(int r = 0; r < nc; r++) {
      float bestC = -3e38f; int bc = -1, bs = -1;
      for(int i = 0; i < nc; i++) {
        if(done[i]) continue;
        const Box& cb = N[ch[i]].b;
        float d[3] = {0.5f * (cb.lo[0] + cb.hi[0]) - cen[0], 0.5f * (cb.lo[1] + cb.hi[1]) - cen[1], 0.5f * (cb.lo[2] + cb.hi[2]) - cen[2]};
        for(int s = 0; s < 8; s++) {
          if(slotUsed[s]) continue;
          float c = d[0] * slotSign(s, 0) + d[1] * slotSign(s, 1) + d[2] * slotSign(s, 2);
          if(c > bestC) { bestC = c; bc = i; bs = s; }
        }
      }
      done[bc] = true; slotUsed[bs] = true; slotOf[bc] = bs;
    }
    int childInSlot[8]; for(int s = 0; s < 8; s++) childInSlot[s] = -1;
    for(int i = 0; i < nc; i++) childInSlot[slotOf[i]] = i;

    // quantisation grid: smallest power-of-two step with extent/step <= 255, bumped until every child fits
    Node8 W{};
    W.px = nb.lo[0]; W.py = nb.lo[1]; W.pz = nb.lo[2];
    int ex[3];
    for(int a = 0; a < 3; a++) {
      float ext = nb.hi[a] - nb.lo[a];
      int e = -60;
      if(ext > 0) { int fe; std::frexp(ext / 255.f, &fe); e = fe; }  // ext/255 <= 2^fe
      ex[a] = std::max(-100, std::min(100, e));
    }
    uint8_t qlo[3][8], qhi[3][8];
    for(int a = 0; a < 3; a++) {
      for(;;) {
        const float step = std::ldexp(1.0f, ex[a]);
        const float p = nb.lo[a];
        bool ok = true;
        for(int s = 0; s < 8 && ok; s++) {
          qlo[a][s] = 0; qhi[a][s] = 0;
          if(childInSlot[s] < 0) continue;
          const Box& cb = N[ch[childInSlot[s]]].b;
          int ql = int(std::floor((double(cb.lo[a]) - double(p)) / double(step)));
          ql = std::max(0, std::min(255, ql));
          while(ql > 0 && p + float(ql) * step > cb.lo[a]) ql--;
          int qh = int(std::ceil((double(cb.hi[a]) - double(p)) / double(step)));
          qh = std::max(0, qh);
          while(qh <= 255 && p + float(qh) * step < cb.hi[a]) qh++;
          if(qh > 255) { ok = false; break; }
          qlo[a][s] = uint8_t(ql); qhi[a][s] = uint8_t(qh);
        }
        if(ok) break;
        ex[a]++;
      }
    }
    W.ex = uint8_t(ex[0] + 127); W.ey = uint8_t(ex[1] + 127); W.ez = uint8_t(ex[2] + 127);
    W.childBase = uint32_t(out.nodes.size());
    W.triBase = uint32_t(out.tris.size());
    uint32_t triOff = 0;
    for(int s = 0; s < 8; s++) {
      W.qlox[s] = qlo[0][s]; W.qloy[s] = qlo[1][s]; W.qloz[s] = qlo[2][s];
      W.qhix[s] = qhi[0][s]; W.qhiy[s] = qhi[1][s]; W.qhiz[s] = qhi[2][s];
      if(childInSlot[s] < 0) { W.meta[s] = 0; continue; }
      const N2& c = N[ch[childInSlot[s]]];
      if(c.leaf) {
        // c.n <= 3 by construction, except the degenerate "root is one big leaf" case which cannot happen for n > 3
        uint32_t cntT = std::min<uint32_t>(c.n, 3u);
        W.meta[s] = uint8_t((((1u << cntT) - 1u) << 5) | triOff);
        for(uint32_t k = 0; k < cntT; k++) out.tris.push_back(flat[idx[c.a + k]]);
        triOff += cntT;
      } else {
        W.imask |= uint8_t(1u << s);
        W.meta[s] = uint8_t((1u << 5) | (24u + uint32_t(s)));
        const uint32_t wi = uint32_t(out.nodes.size());
        out.nodes.push_back(Node8{});
        queue.push_back({ch[childInSlot[s]], wi, w.depth + 1});
      }
    }
    out.nodes[w.wide] = W;
  }
  return true;
}

}  // namespace rt
