// bvh8_builder.cpp — host build of the flat BVH8 (see bvh8.h).  Replaces the driver-side BLAS/TLAS build of
// src/accelstruct.cpp:110-162.  Box culling is made strictly weaker than the triangle test by padding every
// triangle box with 2e-5 x (largest |coordinate|) before anything else (DESIGN.md §Traversal soundness), so the
// closest hit is a function of the triangle set only, never of the tree.
#include "bvh8_builder.h"
#include "dev_math.h"
#include "../../include/rt_cpus.h"
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>
#include <thread>

// Defaults of the tree-quality passes, from profiles/r05_bvh_quality_ab.txt (same-box A/Bs on the real-footprint exterior scene, the lite scene, configs 3 and 5):
// spatial splits (alpha 1e-5, at most +30 % references) followed by four rotation passes: frame in flight -23 % / -5 % / -5 % / -1 %.  The insertion-based pass lowers
// the SAH estimate most and the frame not at all (alone: +-0, on top of splits + rotations: +6 %), so it stays off; the estimate is not what the GPU pays.
#ifndef RT_BVH_REINSERT_DEFAULT
#define RT_BVH_REINSERT_DEFAULT 0
#endif
#ifndef RT_BVH_ROTATE_DEFAULT
#define RT_BVH_ROTATE_DEFAULT 4
#endif
#ifndef RT_BVH_SPLIT_DEFAULT
#define RT_BVH_SPLIT_DEFAULT 1
#endif

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

// Memory of the build (round 6).  Inside bench.py the BVH2 phase took 3-4.7 s where the same build (same tree hash) took 1.3 s in a process that had just freed a
// gigabyte (scripts/r06_build_in_process.sh, profiles/r06_bvh_build.txt section 5): the split builder hands every node's references down in fresh vectors — 2.2 GB
// allocated, touched once and returned over the depth of the tree, from up to 512 threads — and what that costs is the process's address space (mmap / munmap / page faults
// under one lock, slower still once the HIP runtime has registered its notifiers on it), not the builder's arithmetic.  Two allocators take the operating system out of it:
//  * BlockPool / PoolAlloc: blocks of power-of-two size on per-size free lists, reused for the whole build and released when the last build in flight ends;
//  * NoInitAlloc: the 600 MB of BVH2 records are filled by all threads (first touch) instead of by vector::assign on one.
// Under ASan / TSan the pool passes through to operator new so that the sanitizers keep seeing every list's lifetime.
struct BlockPool {
  static constexpr int MINLOG = 16, MAXLOG = 44;   // (64 KB and up: what glibc would take to mmap sooner or later; smaller lists stay in the calling thread's malloc arena)
  std::mutex mu[MAXLOG + 1];
  std::vector<void*> fr[MAXLOG + 1];
  std::mutex muUsers; int users = 0;
  static BlockPool& get() { static BlockPool P; return P; }
  static int cls(size_t bytes) { int k = MINLOG; while(k < MAXLOG && (size_t(1) << k) < bytes) k++; return k; }
  static bool pooled(size_t bytes)
  {
#if defined(__SANITIZE_ADDRESS__) || defined(__SANITIZE_THREAD__)
    (void)bytes; return false;
#else
    return bytes >= (size_t(1) << MINLOG) && bytes <= (size_t(1) << MAXLOG);
#endif
  }
  void* alloc(size_t bytes)
  {
    if(!pooled(bytes)) return ::operator new(bytes);
    const int k = cls(bytes);
    { std::lock_guard<std::mutex> g(mu[k]); if(!fr[k].empty()) { void* p = fr[k].back(); fr[k].pop_back(); return p; } }
    return ::operator new(size_t(1) << k);
  }
  void free(void* p, size_t bytes)
  {
    if(!pooled(bytes)) { ::operator delete(p); return; }
    const int k = cls(bytes);
    std::lock_guard<std::mutex> g(mu[k]); fr[k].push_back(p);
  }
  void enter() { std::lock_guard<std::mutex> g(muUsers); users++; }
  void leave()   // the last build in flight returns the cached blocks (every list of a build is gone before its scope ends)
  {
    std::lock_guard<std::mutex> g(muUsers);
    if(--users > 0) return;
    for(int k = 0; k <= MAXLOG; k++) { std::lock_guard<std::mutex> h(mu[k]); for(void* p : fr[k]) ::operator delete(p); fr[k].clear(); fr[k].shrink_to_fit(); }
  }
  struct Scope { Scope() { BlockPool::get().enter(); } ~Scope() { BlockPool::get().leave(); } };
};
template <class T> struct PoolAlloc {
  using value_type = T;
  PoolAlloc() = default;
  template <class U> PoolAlloc(const PoolAlloc<U>&) {}
  T* allocate(size_t n) { return static_cast<T*>(BlockPool::get().alloc(n * sizeof(T))); }
  void deallocate(T* p, size_t n) { BlockPool::get().free(p, n * sizeof(T)); }
  template <class U> bool operator==(const PoolAlloc<U>&) const { return true; }
  template <class U> bool operator!=(const PoolAlloc<U>&) const { return false; }
};
template <class T> struct NoInitAlloc : std::allocator<T> {   // resize(n) leaves trivially constructible elements untouched
  template <class U> struct rebind { using other = NoInitAlloc<U>; };
  NoInitAlloc() = default;
  template <class U> NoInitAlloc(const NoInitAlloc<U>&) {}
  template <class U> void construct(U* p) { ::new(static_cast<void*>(p)) U; }
  template <class U, class... A> void construct(U* p, A&&... a) { ::new(static_cast<void*>(p)) U(std::forward<A>(a)...); }
};
using N2Vec = std::vector<N2, NoInitAlloc<N2>>;

struct Builder2 {
  const std::vector<Prim>& prims;
  std::vector<uint32_t>& idx;
  N2Vec nodes;
  alignas(64) std::atomic<uint32_t> nodeCount{1};   // (a line of its own: every node of every thread draws from it)
  alignas(64) std::atomic<int> liveThreads{0};
  alignas(64) int maxThreads;
  // (read per build, not once per process: the settings are part of rt_build_accel's cache key)
  const int envNB = getenv("RESTIR_BVH_BINS") ? std::min(64, std::max(4, atoi(getenv("RESTIR_BVH_BINS")))) : 16;
  const float leafSlotCost = getenv("RESTIR_BVH_SLOTCOST") ? float(atof(getenv("RESTIR_BVH_SLOTCOST"))) : 0.25f;  // measured: 0.25 beats 0.5 by 2 % on the Bistro-class scene, bins 16 vs 32 vs 64 make no difference
  Builder2(const std::vector<Prim>& p, std::vector<uint32_t>& i, int threads) : prims(p), idx(i), nodes(std::max<size_t>(2, 2 * p.size() + 2), N2{}), maxThreads(threads) {}

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

// ---- BVH2 with spatial splits (Stich, Friedrich, Dietrich, "Spatial Splits in Bounding Volume Hierarchies", HPG 2009) -----------------------------------
// The reference asks its driver for PREFER_FAST_TRACE trees (src/accelstruct.cpp:125-126, 161).  An object-split SAH tree cannot separate a long thin
// triangle (a rail, a cable, an awning strip, a facade quad) from the small geometry next to it: the boxes of its children overlap and every ray through
// the overlap walks both.  Here a node may instead be cut by a PLANE: a triangle that straddles it is referenced from both sides, each reference bounded
// by the box of the part of the triangle inside its cell (the triangle is clipped, not its box).  References are bounded (RESTIR_BVH_SPLIT_BUDGET x the
// triangle count); a reference is a copy of the 64 B triangle record in leaf order, `triRef` and ids stay per TRIANGLE.
// Results do not depend on it: the verdict of a candidate is a function of (ray, triangle), the closest hit the minimum over (t, id) — testing
// a triangle twice changes neither; every point of a triangle lies in the (padded) box of at least one of its references.
struct Ref { Box b; uint32_t tri; };   // b: UNPADDED bounds of the part of triangle `tri` this reference stands for
using RefVec = std::vector<Ref, PoolAlloc<Ref>>;   // (reference lists come from the build's block pool: see BlockPool)

// fixed-size chunks of [0, n) dealt to at most `threads` threads; f(chunk, begin, end).  The chunking never depends on the thread count, and every use below merges the
// per-chunk results in chunk order (or with min / max / integer sums, which do not care), so what is built is a function of the input alone.
template <class F> void parallelChunks(size_t n, size_t chunk, int threads, F f)
{
  const size_t nc = (n + chunk - 1) / chunk;
  if(nc <= 1 || threads <= 1) { for(size_t c = 0; c < nc; c++) f(c, c * chunk, std::min(n, (c + 1) * chunk)); return; }
  std::atomic<size_t> next{0};
  auto work = [&] { for(;;) { const size_t c = next.fetch_add(1); if(c >= nc) return; f(c, c * chunk, std::min(n, (c + 1) * chunk)); } };
  std::vector<std::thread> pool;
  const int nt = int(std::min<size_t>(size_t(threads), nc));
  for(int i = 1; i < nt; i++) pool.emplace_back(work);
  work();
  for(auto& t : pool) t.join();
}

// Round 6: the build is DETERMINISTIC — the same tree for any thread count and any run.  Until round 5 the threads drew node indices, leaf ranges and the reference
// budget from shared atomics in whatever order they got there: the tree, its SAH figures and the frame time varied from run to run, and subtrees built late found the
// budget spent (advisor finding; a race for the budget had already overrun `leafTris` once, commit 12d774f).  Now every subtree owns what it may use, handed down from
// its parent: a share of the reference budget (what is left after the parent's own split, divided between the children in proportion to their reference counts) and,
// derived from it, a range of node records and of leaf entries sized for the worst case (a subtree of r references and budget b has at most r + b leaves and
// 2 (r + b) - 1 nodes).  Subtrees of more than SEQ_MAX references are cut this way and may be built by different threads; a subtree of at most SEQ_MAX references is
// built by ONE thread, depth first, and inside it the budget is a pool again (what the left child leaves goes to the right one, records and leaf entries are taken
// from the subtree's range in the order of the walk) — sequential, hence reproducible, and the budget does not strand in parts of the scene that need no splits (with
// fixed shares all the way down only 7 k of the 33 k spatial splits of the round-5 tree were taken).  The ranges leave gaps in `nodes` / `leafTris` (unused records
// stay marked as empty leaves); nothing is shared, nothing is atomic.
struct BuilderS {
  const std::vector<Tri48>& flat;
  const float pad;
  N2Vec nodes;
  std::vector<uint32_t> leafTris;
  // Statistics (integer sums: the order of the additions does not matter).  On lines of their own, and a sequential subtree counts in its Pool and adds ONCE when it is done:
  // until the end of round 6 every leaf of every subtree did `leafRefs += n` on a line it shared with `nodes` / `leafTris` / `flat` — 2.7 M read-modify-writes from ~300
  // threads on the line every other access of the builder starts from.  The 289 sequential subtrees of the headline scene (20 ms of work each) took 1.4-2.1 s EACH, 415-620
  // CPU seconds in all, and the phase 1.2 s or 5 s depending on where the scheduler had put the threads (scripts/r06_build_profile.sh, profiles/r06_bvh_build.txt section 5).
  alignas(64) std::atomic<int> liveThreads{0};
  alignas(64) std::atomic<uint64_t> spatialSplits{0};
  alignas(64) std::atomic<uint64_t> leafRefs{0};
  alignas(64) int64_t rootBudget;
  int maxThreads;
  float rootArea = 1.f, alpha = 1e-5f;
  int NB = 16, NBS = 16;   // bins of the object / of the spatial split search
  float leafSlotCost = 0.25f;
  bool areaRule = false;
  // nodes with more than PAR_MIN references bin them in parallel, in chunks of PAR_CHUNK; subtrees of at most SEQ_MAX references: one thread, budget pooled (see above).
  // (RESTIR_BVH_PAR_MIN / RESTIR_BVH_SEQ_MAX: test hooks — the sanitizer jobs of tests/test_bvh_quality.py reach every parallel path on a 50 k-triangle scene.  The
  //  thresholds are part of the build's definition: another value is another, equally reproducible, tree.)
  const size_t PAR_MIN = getenv("RESTIR_BVH_PAR_MIN") ? size_t(std::max(256, atoi(getenv("RESTIR_BVH_PAR_MIN")))) : size_t(1) << 18;
  const size_t PAR_CHUNK = std::min<size_t>(size_t(1) << 16, std::max<size_t>(64, PAR_MIN / 4));
  // (SEQ_MAX 100 000 left sequential subtrees of up to 3 s each — the ones dense with thin triangles — and the build did not scale past 16 threads: 4.8 s on the 256
  //  threads of the GPU box's host; with 16 384 the longest one is a few tenths of a second, profiles/r06_bvh_build.txt)
  const size_t SEQ_MAX = getenv("RESTIR_BVH_SEQ_MAX") ? size_t(std::max(16, atoi(getenv("RESTIR_BVH_SEQ_MAX")))) : 16384;
  struct Pool { int64_t budget; uint32_t nextNode, nextLeaf; uint64_t leafRefs = 0, splits = 0; };     // what a sequential subtree still owns, and what it has counted
  // RESTIR_BVH_TIMING: where the wall time of this phase goes — the own work (bounds, bins, partition) of the nodes above the sequential subtrees by size class, when the
  // last of them was done, and the sequential subtrees (count, CPU seconds, the longest, when the last one ended).  Microseconds since the builder was constructed.
  const bool prof = getenv("RESTIR_BVH_TIMING") && atoi(getenv("RESTIR_BVH_TIMING")) != 0;
  const std::chrono::steady_clock::time_point profT0 = std::chrono::steady_clock::now();
  alignas(64) std::atomic<int64_t> cutUs[32] = {}, cutMaxUs[32] = {}, cutEndUs[32] = {}; std::atomic<int> cutN[32] = {};
  alignas(64) std::atomic<int64_t> seqUs{0}, seqMaxUs{0}, seqEndUs{0}, seqFirstUs{int64_t(1) << 60}; std::atomic<int> seqN{0};
  int64_t nowUs() const { return std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - profT0).count(); }
  static void amax(std::atomic<int64_t>& a, int64_t v) { int64_t o = a.load(); while(o < v && !a.compare_exchange_weak(o, v)) {} }
  static void amin(std::atomic<int64_t>& a, int64_t v) { int64_t o = a.load(); while(o > v && !a.compare_exchange_weak(o, v)) {} }
  void report() const
  {
    if(!prof) return;
    for(int k = 31; k >= 0; k--) if(cutN[k].load())
      fprintf(stderr, "[bvh8 build]   nodes of 2^%d..2^%d references above the sequential subtrees: %5d, own work %.3f s in all, the longest %.3f s, the last one done at %.3f s\n", k, k + 1,
              cutN[k].load(), cutUs[k].load() * 1e-6, cutMaxUs[k].load() * 1e-6, cutEndUs[k].load() * 1e-6);
    fprintf(stderr, "[bvh8 build]   sequential subtrees: %d, %.3f thread-seconds (wall), the longest %.3f s, first started at %.3f s, last ended at %.3f s\n", seqN.load(), seqUs.load() * 1e-6, seqMaxUs.load() * 1e-6,
            seqFirstUs.load() * 1e-6, seqEndUs.load() * 1e-6);
  }
  BuilderS(const std::vector<Tri48>& f, float pad_, size_t n, double budgetFrac, int threads)
    : flat(f), pad(pad_), rootBudget(int64_t(double(n) * budgetFrac)), maxThreads(threads)
  {
    N2 empty; empty.b.reset(); empty.a = 0; empty.n = 0; empty.leaf = true;
    nodes.resize(2 * (n + size_t(rootBudget)) + 16);     // (NoInitAlloc: untouched; filled — and first touched — by all threads)
    parallelChunks(nodes.size(), size_t(1) << 18, std::max(1, threads), [&](size_t, size_t b, size_t e) { for(size_t i = b; i < e; i++) nodes[i] = empty; });
    leafTris.assign(n + size_t(rootBudget) + 16, 0u);
  }

  // bounds of (triangle t) AND box `cell` (closed), by clipping the polygon in double; false: empty
  bool clipBounds(uint32_t t, const Box& cell, Box& out) const
  {
    const Tri48& T = flat[t];
    double poly[2][12][3]; int np = 3, cur = 0;
    poly[0][0][0] = T.v0x; poly[0][0][1] = T.v0y; poly[0][0][2] = T.v0z;
    poly[0][1][0] = double(T.v0x) + double(T.e1x); poly[0][1][1] = double(T.v0y) + double(T.e1y); poly[0][1][2] = double(T.v0z) + double(T.e1z);
    poly[0][2][0] = double(T.v0x) + double(T.e2x); poly[0][2][1] = double(T.v0y) + double(T.e2y); poly[0][2][2] = double(T.v0z) + double(T.e2z);
    for(int ax = 0; ax < 3 && np > 0; ax++)
      for(int side = 0; side < 2 && np > 0; side++) {
        const double plane = side ? double(cell.hi[ax]) : double(cell.lo[ax]);
        const double sg = side ? -1.0 : 1.0;           // inside: sg * (x - plane) >= 0
        int nq = 0;
        for(int i = 0; i < np; i++) {
          const double* a = poly[cur][i]; const double* b = poly[cur][(i + 1) % np];
          const double da = sg * (a[ax] - plane), db = sg * (b[ax] - plane);
          if(da >= 0) { for(int k = 0; k < 3; k++) poly[cur ^ 1][nq][k] = a[k]; nq++; }
          if((da >= 0) != (db >= 0)) {
            const double tt = da / (da - db);
            for(int k = 0; k < 3; k++) poly[cur ^ 1][nq][k] = a[k] + (b[k] - a[k]) * tt;
            poly[cur ^ 1][nq][ax] = plane;
            nq++;
          }
        }
        cur ^= 1; np = nq;
      }
    if(np == 0) return false;
    out.reset();
    for(int i = 0; i < np; i++) {
      float q[3];
      for(int k = 0; k < 3; k++) q[k] = float(poly[cur][i][k]);
      // (double -> float rounds to nearest: may fall short of the true bound by half an ulp; the pad — >= 300 ulp of the largest coordinate — covers it)
      out.grow(q);
    }
    for(int k = 0; k < 3; k++) { out.lo[k] = std::max(out.lo[k], cell.lo[k]); out.hi[k] = std::min(out.hi[k], cell.hi[k]); if(out.lo[k] > out.hi[k]) return false; }
    return true;
  }

  void makeLeaf(N2& N, const RefVec& refs, uint32_t leafBase, Pool* pool)
  {
    for(size_t k = 0; k < refs.size(); k++) leafTris[leafBase + k] = refs[k].tri;
    N.leaf = true; N.a = leafBase; N.n = uint32_t(refs.size());
    if(pool) pool->leafRefs += refs.size(); else leafRefs += refs.size();
  }

  static constexpr int NBMAX = 64;
  struct Bins {   // what one pass over (a chunk of) a node's references collects: object bins by centroid and spatial bins by clipped extent, for the three axes
    Box ob[3][NBMAX]; uint32_t oc[3][NBMAX];
    Box sb[3][NBMAX]; uint32_t en[3][NBMAX], ex[3][NBMAX];
    void reset(int nb, int nbs) { for(int a = 0; a < 3; a++) { for(int i = 0; i < nb; i++) { ob[a][i].reset(); oc[a][i] = 0; } for(int i = 0; i < nbs; i++) { sb[a][i].reset(); en[a][i] = ex[a][i] = 0; } } }
    void merge(const Bins& o, int nb, int nbs) { for(int a = 0; a < 3; a++) { for(int i = 0; i < nb; i++) { ob[a][i].grow(o.ob[a][i]); oc[a][i] += o.oc[a][i]; } for(int i = 0; i < nbs; i++) { sb[a][i].grow(o.sb[a][i]); en[a][i] += o.en[a][i]; ex[a][i] += o.ex[a][i]; } } }
  };
  // which: 1 = the object bins, 2 = the spatial bins (a second pass, only for the nodes whose object split leaves its children overlapping: it clips every straddling
  // reference per bin — until round 6's first restructuring merged the two passes, and with a working budget that is never exhausted EVERY node paid for it)
  void binRefs(const Ref* r, size_t count, const Box& nb, const Box& cb, int which, Bins& B) const
  {
    float k1o[3], k1s[3], w[3]; bool okO[3], okS[3];
    for(int ax = 0; ax < 3; ax++) {
      const float extO = cb.hi[ax] - cb.lo[ax], extS = nb.hi[ax] - nb.lo[ax];
      okO[ax] = which == 1 && extO > 0; okS[ax] = which == 2 && extS > 0;
      k1o[ax] = okO[ax] ? NB * (1.f - 1e-6f) / extO : 0.f;
      k1s[ax] = okS[ax] ? NBS * (1.f - 1e-6f) / extS : 0.f; w[ax] = extS / NBS;
    }
    for(size_t q = 0; q < count; q++) {
      const Ref& R = r[q];
      for(int ax = 0; ax < 3; ax++) {
        if(okO[ax]) {
          const int bi = std::min(NB - 1, std::max(0, int((0.5f * (R.b.lo[ax] + R.b.hi[ax]) - cb.lo[ax]) * k1o[ax])));
          B.ob[ax][bi].grow(R.b); B.oc[ax][bi]++;
        }
        if(okS[ax]) {
          const float lo = nb.lo[ax];
          const int b0 = std::min(NBS - 1, std::max(0, int((R.b.lo[ax] - lo) * k1s[ax]))), b1 = std::min(NBS - 1, std::max(b0, int((R.b.hi[ax] - lo) * k1s[ax])));
          B.en[ax][b0]++; B.ex[ax][b1]++;
          if(b0 == b1) { B.sb[ax][b0].grow(R.b); continue; }
          for(int bi = b0; bi <= b1; bi++) {
            Box cell = R.b, part;
            cell.lo[ax] = std::max(cell.lo[ax], lo + w[ax] * float(bi)); cell.hi[ax] = std::min(cell.hi[ax], bi == NBS - 1 ? nb.hi[ax] : lo + w[ax] * float(bi + 1));
            if(cell.lo[ax] <= cell.hi[ax] && clipBounds(R.tri, cell, part)) B.sb[ax][bi].grow(part);
          }
        }
      }
    }
  }

  // subtree of `refs` rooted at record `node`; its descendants use the records [nodeBase, nodeBase + 2 (|refs| + budget) - 2), its leaves the entries
  // [leafBase, leafBase + |refs| + budget)
  void build(uint32_t node, RefVec refs, int64_t budget, uint32_t nodeBase, uint32_t leafBase, Pool* pool = nullptr)
  {
    Pool own;
    for(;;) {
      N2& N = nodes[node];
      const uint32_t cnt = uint32_t(refs.size());
      if(!pool && cnt <= SEQ_MAX) {   // from here down: one thread, pooled budget — the subtree as one call on its own pool, its counts added when it returns
        own.budget = budget; own.nextNode = nodeBase; own.nextLeaf = leafBase;
        const int64_t t0 = prof ? nowUs() : 0;
        build(node, std::move(refs), budget, nodeBase, leafBase, &own);
        leafRefs += own.leafRefs; if(own.splits) spatialSplits += own.splits;
        if(prof) { const int64_t t1 = nowUs(); seqN++; seqUs += t1 - t0; amax(seqMaxUs, t1 - t0); amax(seqEndUs, t1); amin(seqFirstUs, t0); }
        return;
      }
      const int64_t profStart = (prof && !pool) ? nowUs() : 0;
      if(pool) budget = pool->budget;
      const int par = cnt >= PAR_MIN ? std::max(1, std::min(maxThreads, 32)) : 1;   // (a node has cnt / PAR_CHUNK chunks: 43 at the root of the benchmark scene)
      Box nb, cb; nb.reset(); cb.reset();
      if(par > 1) {
        const size_t nch = (cnt + PAR_CHUNK - 1) / PAR_CHUNK;
        std::vector<Box> pn(nch), pc(nch);
        parallelChunks(cnt, PAR_CHUNK, par, [&](size_t c, size_t b, size_t e) {
          Box x, y; x.reset(); y.reset();
          for(size_t k = b; k < e; k++) { const Ref& r = refs[k]; x.grow(r.b); const float ce[3] = {0.5f * (r.b.lo[0] + r.b.hi[0]), 0.5f * (r.b.lo[1] + r.b.hi[1]), 0.5f * (r.b.lo[2] + r.b.hi[2])}; y.grow(ce); }
          pn[c] = x; pc[c] = y;
        });
        for(size_t c = 0; c < nch; c++) { nb.grow(pn[c]); cb.grow(pc[c]); }
      } else
        for(const Ref& r : refs) { nb.grow(r.b); const float c[3] = {0.5f * (r.b.lo[0] + r.b.hi[0]), 0.5f * (r.b.lo[1] + r.b.hi[1]), 0.5f * (r.b.lo[2] + r.b.hi[2])}; cb.grow(c); }
      N.b = nb;
      for(int a = 0; a < 3; a++) { N.b.lo[a] -= pad; N.b.hi[a] += pad; }
      auto leafHere = [&] { if(pool) { makeLeaf(N, refs, pool->nextLeaf, pool); pool->nextLeaf += cnt; } else makeLeaf(N, refs, leafBase, nullptr); };
      if(cnt <= 1) { leafHere(); return; }
      // ---- object bins on the reference centroids (the rule of Builder2) ----
      const bool mayAdd = budget > 0;
      Bins B;   // (on the stack: 23 KB; the recursion is as deep as the tree)
      B.reset(NB, NBS);
      auto binAll = [&](int which) {
        if(par > 1) {
          const size_t nch = (cnt + PAR_CHUNK - 1) / PAR_CHUNK;
          std::vector<std::unique_ptr<Bins>> part(nch);
          parallelChunks(cnt, PAR_CHUNK, par, [&](size_t c, size_t b, size_t e) { part[c].reset(new Bins); part[c]->reset(NB, NBS); binRefs(refs.data() + b, e - b, nb, cb, which, *part[c]); });
          for(size_t c = 0; c < nch; c++) B.merge(*part[c], NB, NBS);
        } else binRefs(refs.data(), cnt, nb, cb, which, B);
      };
      binAll(1);
      // ---- object split ----
      float best = 3e38f; int bestAxis = -1, bestBin = 0; Box bestL, bestR; uint32_t bestNL = 0, bestNR = 0;
      for(int ax = 0; ax < 3; ax++) {
        const float ext = cb.hi[ax] - cb.lo[ax];
        if(!(ext > 0)) continue;
        const Box* bb = B.ob[ax]; const uint32_t* bc = B.oc[ax];
        Box rb[NBMAX]; uint32_t rc[NBMAX];
        Box acc; acc.reset(); uint32_t c = 0;
        for(int i = NB - 1; i > 0; i--) { acc.grow(bb[i]); c += bc[i]; rb[i] = acc; rc[i] = c; }
        acc.reset(); c = 0;
        for(int i = 0; i < NB - 1; i++) {
          acc.grow(bb[i]); c += bc[i];
          if(c == 0 || rc[i + 1] == 0) continue;
          const float cost = acc.area() * c + rb[i + 1].area() * rc[i + 1];
          if(cost < best) { best = cost; bestAxis = ax; bestBin = i; bestL = acc; bestR = rb[i + 1]; bestNL = c; bestNR = rc[i + 1]; }
        }
      }
      const float pa = std::max(nb.area(), 1e-30f);
      // ---- spatial split: only where the object split leaves its children overlapping (Stich et al., 4.5), and while references may be added ----
      float sBest = 3e38f; int sAxis = -1; float sPlane = 0.f; uint32_t sNL = 0, sNR = 0; Box sL, sR;
      bool trySpatial = cnt >= 2 && mayAdd;
      if(trySpatial && bestAxis >= 0) {
        Box ov;
        for(int a = 0; a < 3; a++) { ov.lo[a] = std::max(bestL.lo[a], bestR.lo[a]); ov.hi[a] = std::min(bestL.hi[a], bestR.hi[a]); }
        trySpatial = ov.area() > alpha * rootArea;   // (area() is 0 for an empty intersection)
      }
      if(trySpatial) {
        binAll(2);
        for(int ax = 0; ax < 3; ax++) {
          const float lo = nb.lo[ax], ext = nb.hi[ax] - nb.lo[ax];
          if(!(ext > 0)) continue;
          const Box* bb = B.sb[ax]; const uint32_t* en = B.en[ax]; const uint32_t* exx = B.ex[ax];
          const float w = ext / NBS;
          Box rb[NBMAX]; uint32_t rc[NBMAX];
          Box acc; acc.reset(); uint32_t c = 0;
          for(int i = NBS - 1; i > 0; i--) { acc.grow(bb[i]); c += exx[i]; rb[i] = acc; rc[i] = c; }
          acc.reset(); c = 0;
          for(int i = 0; i < NBS - 1; i++) {
            acc.grow(bb[i]); c += en[i];
            if(c == 0 || rc[i + 1] == 0 || c >= cnt || rc[i + 1] >= cnt) continue;   // a cut that leaves every reference on one side makes no progress
            const float cost = acc.area() * c + rb[i + 1].area() * rc[i + 1];
            if(cost < sBest) { sBest = cost; sAxis = ax; sPlane = lo + w * float(i + 1); sNL = c; sNR = rc[i + 1]; sL = acc; sR = rb[i + 1]; }
          }
        }
      }
      const bool spatial = sAxis >= 0 && sBest < best && int64_t(sNL + sNR) - int64_t(cnt) <= budget;
      const float chosen = spatial ? sBest : best;
      if(cnt <= 3) {
        if((bestAxis < 0 && !spatial) || leafSlotCost * pa + chosen >= float(cnt) * pa) { leafHere(); return; }
      }
      RefVec left, right;
      if(spatial) {
        left.reserve(sNL); right.reserve(sNR);
        Box L = sL, R = sR; uint32_t nl = sNL, nr = sNR;   // running estimates for the unsplitting rule (Stich et al., 4.4)
        for(const Ref& r : refs) {
          if(r.b.hi[sAxis] <= sPlane) { left.push_back(r); continue; }
          if(r.b.lo[sAxis] >= sPlane) { right.push_back(r); continue; }
          Box cl = r.b, cr = r.b, pl, pr;
          cl.hi[sAxis] = sPlane; cr.lo[sAxis] = sPlane;
          const bool okL = clipBounds(r.tri, cl, pl), okR = clipBounds(r.tri, cr, pr);
          if(!okL && !okR) { left.push_back(r); continue; }      // (cannot happen for a non-empty reference; keep it whole)
          if(!okR) { Ref q = r; q.b = pl; left.push_back(q); nr--; continue; }
          if(!okL) { Ref q = r; q.b = pr; right.push_back(q); nl--; continue; }
          // split it, or keep it whole on one side when that is cheaper
          Box Lw = L, Rw = R; Lw.grow(r.b); Rw.grow(r.b);
          const float cSplit = L.area() * nl + R.area() * nr, cLeft = Lw.area() * nl + R.area() * (nr - 1), cRight = L.area() * (nl - 1) + Rw.area() * nr;
          if(cLeft < cSplit && cLeft <= cRight && nr > 1) { left.push_back(r); L = Lw; nr--; }
          else if(cRight < cSplit && nl > 1) { right.push_back(r); R = Rw; nl--; }
          else { Ref a = r, b = r; a.b = pl; b.b = pr; left.push_back(a); right.push_back(b); }
        }
        // no progress, or more references than this subtree may add (cannot happen: the unsplitting rule only lowers the binned estimate checked above): the object split
        if(left.empty() || right.empty() || left.size() >= cnt || right.size() >= cnt || int64_t(left.size() + right.size()) - int64_t(cnt) > budget) { left.clear(); right.clear(); }
        else if(pool) pool->splits++; else spatialSplits++;
      }
      if(left.empty()) {
        if(bestAxis >= 0) {
          left.reserve(bestNL); right.reserve(bestNR);   // (the bin counts of the chosen split: one allocation per list)
          const float ext = cb.hi[bestAxis] - cb.lo[bestAxis], k1 = NB * (1.f - 1e-6f) / ext, lo = cb.lo[bestAxis];
          for(const Ref& r : refs) {
            const int bi = std::min(NB - 1, std::max(0, int((0.5f * (r.b.lo[bestAxis] + r.b.hi[bestAxis]) - lo) * k1)));
            (bi <= bestBin ? left : right).push_back(r);
          }
        }
        if(left.empty() || right.empty()) {   // identical centroids: split by index
          left.assign(refs.begin(), refs.begin() + cnt / 2); right.assign(refs.begin() + cnt / 2, refs.end());
        }
      }
      RefVec().swap(refs);
      const int64_t rest = budget - (int64_t(left.size() + right.size()) - int64_t(cnt));
      if(pool) {   // sequential subtree: the two child records from the pool, the left child first; what it leaves of the budget is the right child's
        pool->budget = rest;
        const uint32_t child = pool->nextNode; pool->nextNode += 2;
        N.leaf = false; N.a = child; N.n = 0;
        build(child, std::move(left), 0, 0u, 0u, pool);
        node = child + 1; refs = std::move(right);
        continue;
      }
      // what this split did not use of the subtree's budget goes to the children in proportion to their reference counts; their record / leaf ranges follow from it
      // (RESTIR_BVH_BUDGET_RULE=area weighs a child's count with the area of its references' bounds: measured against the count rule in profiles/r06_bvh_build.txt)
      double wl = double(left.size()), wr = double(right.size());
      if(areaRule) { Box bL, bR; bL.reset(); bR.reset(); for(const Ref& r : left) bL.grow(r.b); for(const Ref& r : right) bR.grow(r.b); wl *= double(bL.area()) + 1e-30; wr *= double(bR.area()) + 1e-30; }
      const int64_t bl = std::min<int64_t>(rest, std::max<int64_t>(0, int64_t(double(rest) * (wl / (wl + wr))))), br = rest - bl;
      const uint32_t child = nodeBase;
      const uint32_t lNodes = uint32_t(2 * (int64_t(left.size()) + bl) - 2), lLeaves = uint32_t(int64_t(left.size()) + bl);
      if(prof) { int k = 0; while((uint32_t(2) << k) <= cnt && k < 31) k++; const int64_t t1 = nowUs(); cutN[k]++; cutUs[k] += t1 - profStart; amax(cutMaxUs[k], t1 - profStart); amax(cutEndUs[k], t1); }
      N.leaf = false; N.a = child; N.n = 0;
      const uint32_t lBase = nodeBase + 2, rBase = nodeBase + 2 + lNodes, lLeaf = leafBase, rLeaf = leafBase + lLeaves;
      if(liveThreads.load() < 2 * maxThreads) {   // (half of these threads wait in join() for their children)
        liveThreads++;
        std::thread th([this, child, bl, lBase, lLeaf, l = std::move(left)]() mutable { build(child, std::move(l), bl, lBase, lLeaf); liveThreads--; });
        build(child + 1, std::move(right), br, rBase, rLeaf);
        th.join();
        return;
      }
      build(child, std::move(left), bl, lBase, lLeaf);
      node = child + 1; refs = std::move(right); budget = br; nodeBase = rBase; leafBase = rLeaf;
    }
  }
};

// ---- tree rotations on the finished BVH2 (Kensler, "Tree Rotations for Improving Bounding Volume Hierarchies", RT 2008) ---------------------------------------
// Bottom-up, for every internal node n with children (L, R): exchanging a child with a grandchild on the other side changes only the box of the child that
// loses / gains a subtree; the exchange with the largest area reduction is applied.  Children are adjacent records (a, a + 1) and a record carries its
// subtree by index, so an exchange is a swap of two records.  RESTIR_BVH_ROTATE = number of passes (0 = off).
inline Box unite(const Box& x, const Box& y) { Box r = x; r.grow(y); return r; }
// the best exchange at internal record n, applied; false: none improves
inline bool rotateAt(N2Vec& N, uint32_t n, bool gg)
{
  N2& P = N[n];
  if(P.leaf) return false;
  const uint32_t L = P.a, R = P.a + 1;
  float best = 0.f; int which = -1;
  if(!N[R].leaf) {
    const uint32_t RL = N[R].a, RR = N[R].a + 1;
    const float base = N[R].b.area();
    const float d0 = unite(N[L].b, N[RR].b).area() - base;   // L <-> RL
    const float d1 = unite(N[RL].b, N[L].b).area() - base;   // L <-> RR
    if(d0 < best) { best = d0; which = 0; }
    if(d1 < best) { best = d1; which = 1; }
  }
  if(!N[L].leaf) {
    const uint32_t LL = N[L].a, LR = N[L].a + 1;
    const float base = N[L].b.area();
    const float d2 = unite(N[R].b, N[LR].b).area() - base;   // R <-> LL
    const float d3 = unite(N[LL].b, N[R].b).area() - base;   // R <-> LR
    if(d2 < best) { best = d2; which = 2; }
    if(d3 < best) { best = d3; which = 3; }
  }
  if(gg && !N[L].leaf && !N[R].leaf) {   // grandchild <-> grandchild across the two children: both child boxes change (LL <-> RL, LL <-> RR; the other two are mirror images)
    const uint32_t LL = N[L].a, LR = N[L].a + 1, RL = N[R].a, RR = N[R].a + 1;
    const float base = N[L].b.area() + N[R].b.area();
    const float d4 = unite(N[RL].b, N[LR].b).area() + unite(N[LL].b, N[RR].b).area() - base;   // LL <-> RL
    const float d5 = unite(N[RR].b, N[LR].b).area() + unite(N[RL].b, N[LL].b).area() - base;   // LL <-> RR
    if(d4 < best) { best = d4; which = 4; }
    if(d5 < best) { best = d5; which = 5; }
  }
  if(which < 0) return false;
  if(which >= 4) {
    const uint32_t LL = N[L].a, g = N[R].a + uint32_t(which - 4);
    std::swap(N[LL], N[g]);
    N[L].b = unite(N[N[L].a].b, N[N[L].a + 1].b);
    N[R].b = unite(N[N[R].a].b, N[N[R].a + 1].b);
  } else if(which < 2) {
    const uint32_t g = N[R].a + uint32_t(which);
    std::swap(N[L], N[g]);
    N[R].b = unite(N[N[R].a].b, N[N[R].a + 1].b);
  } else {
    const uint32_t g = N[L].a + uint32_t(which - 2);
    std::swap(N[R], N[g]);
    N[L].b = unite(N[N[L].a].b, N[N[L].a + 1].b);
  }
  return true;
}
// One bottom-up pass.  The order comes from the TREE — reverse pre-order: every record after all of its descendants — not from the record indices: "children have
// larger indices than their parent" stops being true with the first exchange (a record carries its subtree by index, the one moved up keeps children with smaller
// indices; advisor finding of round 5) and never held for the ranges the deterministic builder hands out.  An exchange at n only permutes records INSIDE n's subtree, all
// of which the pass has already visited, so the order taken at the start of the pass stays bottom-up.  Parallel (round 6): the tree is cut ROT_CUT levels below the
// root; the subtrees under the cut are independent — one task each — and the few records above it follow serially.  Which thread runs a subtree changes nothing in it:
// the result is the sequential pass's, for any thread count.
uint64_t rotatePass(N2Vec& N, bool gg, int threads)
{
  constexpr int ROT_CUT = 9;   // up to 512 subtrees
  std::vector<uint32_t> top, roots;
  {
    std::vector<std::pair<uint32_t, int>> st; st.push_back({0u, 0});
    while(!st.empty()) {
      const auto [q, d] = st.back(); st.pop_back();
      if(N[q].leaf) continue;
      if(d == ROT_CUT) { roots.push_back(q); continue; }
      top.push_back(q);   // pre-order
      st.push_back({N[q].a + 1, d + 1}); st.push_back({N[q].a, d + 1});
    }
  }
  std::vector<uint64_t> applied(roots.size(), 0);
  parallelChunks(roots.size(), 1, threads, [&](size_t c, size_t, size_t) {
    std::vector<uint32_t> pre, st2; st2.push_back(roots[c]);
    while(!st2.empty()) { const uint32_t q = st2.back(); st2.pop_back(); if(N[q].leaf) continue; pre.push_back(q); st2.push_back(N[q].a); st2.push_back(N[q].a + 1); }
    uint64_t k = 0;
    for(size_t i = pre.size(); i-- > 0;) k += rotateAt(N, pre[i], gg) ? 1 : 0;
    applied[c] = k;
  });
  uint64_t total = 0;
  for(uint64_t k : applied) total += k;
  for(size_t i = top.size(); i-- > 0;) total += rotateAt(N, top[i], gg) ? 1 : 0;
  return total;
}

// ---- insertion-based optimisation of the finished BVH2 (after Bittner, Hapala, Havran, "Fast Insertion-Based Optimization of Bounding Volume Hierarchies",
//      CGF 2013) ----------------------------------------------------------------------------------------------------------------------------------------------
// A top-down SAH build decides every split with what it knows at that level; subtrees that ended up in the wrong place (a rail that runs through three
// buildings' boxes) stay there.  Here the nodes whose boxes are the least efficient (large, and much larger than their children) are taken apart: each of
// their children is removed from the tree — its sibling moves up into the parent's place — and put back where it increases the tree's area the least, found by
// a branch-and-bound search from the root.  Records carry their subtree by index (children are the adjacent records a, a + 1), so removal and insertion move
// three records and refit the boxes on two root paths.  RESTIR_BVH_REINSERT = passes (0 = off), each over the worst 2 % of the internal nodes.
struct Reinserter {
  N2Vec& N;
  uint32_t count;
  std::vector<int32_t> parent;
  uint64_t moved = 0;
  Reinserter(N2Vec& n, uint32_t c) : N(n), count(c), parent(c, -1)
  {
    for(uint32_t i = 0; i < c; i++) if(!N[i].leaf) { parent[N[i].a] = int32_t(i); parent[N[i].a + 1] = int32_t(i); }
  }
  void put(uint32_t dst, const N2& rec)
  {
    N[dst] = rec;
    if(!rec.leaf) { parent[rec.a] = int32_t(dst); parent[rec.a + 1] = int32_t(dst); }
  }
  void refit(int32_t a)
  {
    for(; a >= 0; a = parent[size_t(a)]) {
      const Box nb = unite(N[N[size_t(a)].a].b, N[N[size_t(a)].a + 1].b);
      if(memcmp(&nb, &N[size_t(a)].b, sizeof(Box)) == 0) break;
      N[size_t(a)].b = nb;
    }
  }
  // remove the subtree in slot x and put it back at the best place; false: left where it was
  bool reinsert(uint32_t x)
  {
    const int32_t p = parent[x];
    if(p <= 0) return false;                       // the root's children stay (the root record does not move)
    const uint32_t pairBase = N[size_t(p)].a, s = pairBase + (x == pairBase ? 1u : 0u);
    const N2 X = N[x], S = N[s];
    const float ax = X.b.area();
    const int32_t g = parent[size_t(p)];
    put(uint32_t(p), S);
    refit(g);
    // branch and bound: cost of inserting X as the sibling of t = area(t u X) + the growth of every ancestor of t
    struct Item { float induced; uint32_t node; };
    auto cmp = [](const Item& a, const Item& b) { return a.induced > b.induced; };
    std::vector<Item> heap;
    heap.push_back({0.f, 0u});
    float best = 3e38f; uint32_t bestT = uint32_t(p);
    while(!heap.empty()) {
      std::pop_heap(heap.begin(), heap.end(), cmp);
      const Item it = heap.back(); heap.pop_back();
      if(it.induced + ax >= best) break;            // every remaining candidate costs at least its induced part + area(X)
      const N2& T = N[it.node];
      const float direct = unite(T.b, X.b).area();
      const float total = it.induced + direct;
      if(total < best) { best = total; bestT = it.node; }
      const float childInduced = total - T.b.area();
      if(!T.leaf && childInduced + ax < best) {
        heap.push_back({childInduced, T.a}); std::push_heap(heap.begin(), heap.end(), cmp);
        heap.push_back({childInduced, T.a + 1}); std::push_heap(heap.begin(), heap.end(), cmp);
      }
    }
    const N2 T = N[bestT];
    put(pairBase, T); put(pairBase + 1, X);
    parent[pairBase] = parent[pairBase + 1] = int32_t(bestT);
    N2 J; J.leaf = false; J.a = pairBase; J.n = 0; J.b = unite(T.b, X.b);
    N[bestT] = J;
    refit(parent[bestT]);
    const bool changed = bestT != uint32_t(p);
    if(changed) moved++;
    return changed;
  }
  void pass(float fraction)
  {
    std::vector<std::pair<float, uint32_t>> cand;
    cand.reserve(count / 2);
    for(uint32_t i = 1; i < count; i++) {
      if(N[i].leaf || parent[i] < 0) continue;
      const float a = N[i].b.area(), l = N[N[i].a].b.area(), r = N[N[i].a + 1].b.area();
      const float m = a * (a / std::max(1e-30f, 0.5f * (l + r))) * (a / std::max(1e-30f, std::min(l, r)));   // Bittner et al.: area x (area / mean child) x (area / smallest child)
      cand.push_back({m, i});
    }
    const size_t k = std::max<size_t>(1, size_t(double(cand.size()) * fraction));
    std::partial_sort(cand.begin(), cand.begin() + std::min(k, cand.size()), cand.end(), [](const auto& x, const auto& y) { return x.first > y.first; });
    for(size_t c = 0; c < std::min(k, cand.size()); c++) {
      const uint32_t n = cand[c].second;
      if(N[n].leaf) continue;                       // (an earlier move may have put a leaf record here)
      const uint32_t l = N[n].a;
      reinsert(l);
      if(!N[n].leaf && N[n].a == l) reinsert(l + 1);   // (n still holds the same pair: its other child; otherwise the record at n is another subtree now)
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

namespace {
// RESTIR_BVH_TIMING=1: seconds per phase of the build on stderr (round 6: where the 4.6 s of the headline scene went)
struct PhaseTimer {
  const bool on = getenv("RESTIR_BVH_TIMING") && atoi(getenv("RESTIR_BVH_TIMING")) != 0;
  std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
  void lap(const char* what)
  {
    if(!on) return;
    const auto n = std::chrono::steady_clock::now();
    fprintf(stderr, "[bvh8 build] %-28s %.3f s\n", what, std::chrono::duration<double>(n - t).count());
    t = n;
  }
};
}  // namespace

bool buildBvh8(const rt_scene_desc& sc, BuildOutput& out, int threads, bool plainTree)
{
  PhaseTimer timer;
  // ---- 1. flatten: one world-space triangle per (instance, primitive); globalId = running index ------------
  out.instances.resize(sc.numInstances);
  size_t total = 0;
  for(uint32_t i = 0; i < sc.numInstances; i++) total += sc.primMeshes[sc.instances[i].primMesh].indexCount / 3;
  std::vector<Tri48> flat(total);
  out.triRef.resize(total);
  // (round 6: instances in parallel — a task per instance, its first record from a prefix sum; the largest |coordinate| is a maximum: any order gives the same value)
  std::vector<size_t> first(size_t(sc.numInstances) + 1, 0);
  for(uint32_t i = 0; i < sc.numInstances; i++) first[i + 1] = first[i] + sc.primMeshes[sc.instances[i].primMesh].indexCount / 3;
  std::vector<float> scaleOf(sc.numInstances, 0.f);
  parallelChunks(sc.numInstances, 1, std::max(1, threads), [&](size_t ii, size_t, size_t) {
    const uint32_t i = uint32_t(ii);
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
    float sm = 0.f;
    size_t g = first[i];
    for(uint32_t p = 0; p < pm.indexCount / 3; p++, g++) {
      const uint32_t* ix = &sc.indices[pm.firstIndex + 3 * p];
      float w[3][3];
      for(int k = 0; k < 3; k++) {
        const rt_vec3& q = sc.vertices[pm.vertexOffset + ix[k]].position;
        xformPointRaw(di.o2w, q.x, q.y, q.z, w[k]);
        for(int a = 0; a < 3; a++) sm = std::max(sm, std::fabs(w[k][a]));
      }
      Tri48& T = flat[g];
      T.v0x = w[0][0]; T.v0y = w[0][1]; T.v0z = w[0][2];
      T.e1x = w[1][0] - w[0][0]; T.e1y = w[1][1] - w[0][1]; T.e1z = w[1][2] - w[0][2];
      T.e2x = w[2][0] - w[0][0]; T.e2y = w[2][1] - w[0][1]; T.e2z = w[2][2] - w[0][2];
      T.globalId = uint32_t(g); T.flags = f; T.alphaIdx = 0; T.omm[0] = T.omm[1] = T.omm[2] = T.omm[3] = 0;
      out.triRef[g] = TriRef{i, p};
    }
    scaleOf[i] = sm;
  });
  float scale = 1e-3f;
  for(float v : scaleOf) scale = std::max(scale, v);
  timer.lap("flatten");
  const size_t n = total;
  out.nodes.clear(); out.tris.clear(); out.maxDepth = 0; out.sahNodeSteps = out.sahTriSteps = out.sahNodeStepsQ = out.sahTriStepsQ = 0; out.references = 0; out.spatialSplits = 0; out.rotations = 0; out.reinsertions = 0;
  if(n == 0) {  // a single empty node keeps the kernels branch-free
    Node8 e{}; e.ex = e.ey = e.ez = 127;
    out.nodes.push_back(e);
    return true;
  }
  const float pad = 2e-5f * scale;
  out.pad = pad;

  // ---- 2. padded triangle boxes + binned-SAH BVH2 ---------------------------------------------------------------
  // RESTIR_BVH_SPLIT: 0 = object splits only (the tree of rounds 1-4), 1 = spatial splits (BuilderS); RESTIR_BVH_SPLIT_BUDGET: added references / triangles
  const int splitMode = plainTree ? 0 : (getenv("RESTIR_BVH_SPLIT") ? atoi(getenv("RESTIR_BVH_SPLIT")) : RT_BVH_SPLIT_DEFAULT);
  const bool useSplits = splitMode > 0 && n > 1;
  std::vector<Prim> prims(useSplits ? 0 : n);   // (the padded boxes + centroids are the object-split builder's input; the spatial-split builder works on references)
  std::vector<uint32_t> idx(useSplits ? 0 : n);
  if(!useSplits) parallelChunks(n, size_t(1) << 16, std::max(1, threads), [&](size_t, size_t b0, size_t e0) {
    for(size_t i = b0; i < e0; i++) {
      const Tri48& T = flat[i];
      const float v[3][3] = {{T.v0x, T.v0y, T.v0z}, {T.v0x + T.e1x, T.v0y + T.e1y, T.v0z + T.e1z}, {T.v0x + T.e2x, T.v0y + T.e2y, T.v0z + T.e2z}};
      Prim& P = prims[i];
      P.b.reset();
      for(int k = 0; k < 3; k++) P.b.grow(v[k]);
      // e1/e2 were rounded when stored: v0+e1 may differ from v1 by an ulp; the pad (>= 300 ulp) absorbs it
      for(int a = 0; a < 3; a++) { P.b.lo[a] -= pad; P.b.hi[a] += pad; P.c[a] = 0.5f * (P.b.lo[a] + P.b.hi[a]); }
      idx[i] = uint32_t(i);
    }
  });
  timer.lap("triangle boxes");
  const double splitBudget = getenv("RESTIR_BVH_SPLIT_BUDGET") ? std::max(0.0, std::min(2.0, atof(getenv("RESTIR_BVH_SPLIT_BUDGET")))) : 0.3;
  const float splitAlpha = getenv("RESTIR_BVH_SPLIT_ALPHA") ? float(atof(getenv("RESTIR_BVH_SPLIT_ALPHA"))) : 1e-5f;
  std::unique_ptr<Builder2> B2;   // (built only when it builds the tree: its constructor allocates 2 n records — advisor finding of round 5)
  std::unique_ptr<BuilderS> BS;
  uint32_t n2count = 0;
  if(useSplits) {
    // The budget is a cap on the TOTAL of added references.  Shares of exactly that total strand most of it in parts of the scene that need no splits (see
    // BuilderS), so the build first runs with shares of a generous working budget (every subtree may triple its references: no sequential subtree of the benchmark scenes comes near its share): on the benchmark scenes the overlap
    // criterion (alpha) then decides alone — 0.26 n added references on the real exterior scene, the tree of round 5 — and the total stays under the cap.  Only when it
    // does not is the build redone with shares of the cap itself, which bound the total by construction.  Both attempts are deterministic.
    const size_t RCH = size_t(1) << 16;
    Box root; root.reset();
    BlockPool::Scope poolScope;
    RefVec refs0(n);
    {
      std::vector<Box> rootOf((n + RCH - 1) / RCH);
      parallelChunks(n, RCH, std::max(1, threads), [&](size_t c, size_t b0, size_t e0) {
        Box rb; rb.reset();
        for(size_t i = b0; i < e0; i++) {
          const Tri48& T = flat[i];
          const float v[3][3] = {{T.v0x, T.v0y, T.v0z}, {T.v0x + T.e1x, T.v0y + T.e1y, T.v0z + T.e1z}, {T.v0x + T.e2x, T.v0y + T.e2y, T.v0z + T.e2z}};
          refs0[i].tri = uint32_t(i); refs0[i].b.reset();
          for(int k = 0; k < 3; k++) refs0[i].b.grow(v[k]);   // the UNPADDED bounds (the same expressions as the padded boxes of the object-split builder)
          rb.grow(refs0[i].b);
        }
        rootOf[c] = rb;
      });
      for(const Box& b : rootOf) root.grow(b);
    }
    const double workBudget = getenv("RESTIR_BVH_SPLIT_WORK") ? std::max(0.0, atof(getenv("RESTIR_BVH_SPLIT_WORK"))) : 2.0;
    for(int attempt = 0; attempt < 2; attempt++) {
      const bool strict = attempt == 1 || workBudget <= splitBudget;
      BS.reset();   // (release the first attempt's arrays before the second one allocates)
      BS.reset(new BuilderS(flat, pad, n, strict ? splitBudget : workBudget, std::max(1, threads)));
      RefVec refs = refs0;
      if(strict) RefVec().swap(refs0);
      BS->rootArea = std::max(root.area(), 1e-30f); BS->alpha = splitAlpha;
      BS->NB = getenv("RESTIR_BVH_BINS") ? std::min(64, std::max(4, atoi(getenv("RESTIR_BVH_BINS")))) : 16;
      BS->NBS = getenv("RESTIR_BVH_SBINS") ? std::min(64, std::max(4, atoi(getenv("RESTIR_BVH_SBINS")))) : 16;
      BS->leafSlotCost = getenv("RESTIR_BVH_SLOTCOST") ? float(atof(getenv("RESTIR_BVH_SLOTCOST"))) : 0.25f;
      BS->areaRule = getenv("RESTIR_BVH_BUDGET_RULE") && strcmp(getenv("RESTIR_BVH_BUDGET_RULE"), "area") == 0;
      BS->build(0, std::move(refs), BS->rootBudget, 1u, 0u);
      BS->report();
      n2count = uint32_t(BS->nodes.size());   // (record ranges are handed out per subtree: the records in use are not contiguous)
      out.references = BS->leafRefs.load(); out.spatialSplits = BS->spatialSplits.load();
      if(strict || double(out.references - n) <= double(n) * splitBudget) break;
    }
  } else {
    B2.reset(new Builder2(prims, idx, std::max(1, threads)));
    B2->build(0, 0, uint32_t(n));
    n2count = B2->nodeCount.load();
    out.references = n; out.spatialSplits = 0;
  }
  timer.lap("BVH2 (splits)");
  {
    const bool dp = plainTree;   // (plain tree: no quality passes.  The SAH-optimal collapse takes its bottom-up order from the tree itself since round 5, so the passes may run before it)
    const int rotate = dp ? 0 : (getenv("RESTIR_BVH_ROTATE") ? atoi(getenv("RESTIR_BVH_ROTATE")) : RT_BVH_ROTATE_DEFAULT);
    N2Vec& M = BS ? BS->nodes : B2->nodes;
    const bool rotateGG = getenv("RESTIR_BVH_ROTATE_GG") && atoi(getenv("RESTIR_BVH_ROTATE_GG")) != 0;
    const int reins = dp ? 0 : (getenv("RESTIR_BVH_REINSERT") ? atoi(getenv("RESTIR_BVH_REINSERT")) : RT_BVH_REINSERT_DEFAULT);
    if(reins > 0) {
      Reinserter RI(M, n2count);
      for(int pass = 0; pass < reins; pass++) RI.pass(0.02f);
      out.reinsertions = RI.moved;
    }
    for(int pass = 0; pass < rotate; pass++) { const uint64_t k = rotatePass(M, rotateGG, std::max(1, threads)); out.rotations += k; if(k == 0) break; }
  }
  timer.lap("rotations / reinsertion");
  const N2Vec& N = BS ? BS->nodes : B2->nodes;
  const std::vector<uint32_t>& leafTris = BS ? BS->leafTris : idx;

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
  const bool useDp = getenv("RESTIR_BVH_COLLAPSE") && strcmp(getenv("RESTIR_BVH_COLLAPSE"), "dp") == 0;
  static const float cNode = 2.3f;   // a node step is ~230 instructions,
  static const float cTri = 1.0f;                                                                            // a triangle step ~100
  std::vector<float> cost;        // [n][i], i = 1..7 at [n * 8 + i]
  std::vector<uint8_t> choice;    // [n][i]: 0 = single slot (leaf / wide node), 0xff = same as i - 1, else k = slots of the left child
  std::vector<uint8_t> k8;        // [n]: left child's share of the 8 slots when n becomes a wide node
  if(useDp) {
    cost.assign(size_t(n2count) * 8, 0.f); choice.assign(size_t(n2count) * 8, 0); k8.assign(n2count, 1);
    // bottom-up order: children before their parent.  The builders allocate children after their parent, but the quality passes move records (a rotation swaps
    // two records, a reinsertion moves three), so the order is taken from the tree itself
    std::vector<uint32_t> post; post.reserve(n2count);
    {
      std::vector<uint32_t> st2; st2.push_back(0u);
      while(!st2.empty()) { const uint32_t q = st2.back(); st2.pop_back(); post.push_back(q); if(!N[q].leaf) { st2.push_back(N[q].a); st2.push_back(N[q].a + 1); } }
    }
    for(size_t pi = post.size(); pi-- > 0;) {   // reverse pre-order: every node comes after all of its descendants
      const uint32_t n = post[pi];
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
  // Round 6: level by level, two parallel passes per level around a prefix sum.  Pass 1 picks a wide node's children (collapse), assigns them to slots and counts its
  // internal children and leaf triangles; the prefix sum over the level (in level order) gives every node its child base and triangle base — the same numbers the
  // sequential breadth-first queue of rounds 1-5 produced; pass 2 quantises, writes the node, copies its triangle records and names its children's records for the
  // next level.  The statistics are summed per fixed chunk and then in chunk order: the same bits for any thread count.
  struct Work { uint32_t n2; uint32_t wide; };
  struct Wide { uint32_t ch[8]; int8_t childInSlot[8]; uint8_t nc, nInner, nTris; };
  const bool optSlots = getenv("RESTIR_BVH_SLOTS") && strcmp(getenv("RESTIR_BVH_SLOTS"), "opt") == 0;
  const int T = std::max(1, threads);
  constexpr size_t WCH = 512;
  std::vector<Work> level, nextLevel;
  out.nodes.reserve(n / 2 + 16);
  out.tris.reserve(size_t(out.references));
  out.nodes.push_back(Node8{});
  // a root that is itself a leaf gets wrapped by a one-child wide node (handled by the generic path below)
  level.push_back({0, 0});
  const double ra = std::max(1e-30, double(N[0].b.area()));
  int depth = 0;
  while(!level.empty()) {
    depth++;
    out.maxDepth = std::max(out.maxDepth, depth);
    const size_t L = level.size();
    std::vector<Wide> wide(L);
    // ---- pass 1: children, slots, counts ----
    parallelChunks(L, WCH, T, [&](size_t, size_t b0, size_t e0) {
      for(size_t wi = b0; wi < e0; wi++) {
        const Work w = level[wi];
        Wide& X = wide[wi];
        uint32_t* ch = X.ch; int nc = 0;
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
        X.nc = uint8_t(nc);
        // node box = union of child boxes
        Box nb; nb.reset();
        for(int i = 0; i < nc; i++) nb.grow(N[ch[i]].b);
        // slot assignment: child whose centroid lies furthest towards corner s gets slot s (greedy best-pair)
        int slotOf[8]; bool slotUsed[8] = {false, false, false, false, false, false, false, false}; bool done[8] = {false, false, false, false, false, false, false, false};
        const float cen[3] = {0.5f * (nb.lo[0] + nb.hi[0]), 0.5f * (nb.lo[1] + nb.hi[1]), 0.5f * (nb.lo[2] + nb.hi[2])};
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
        // RESTIR_BVH_SLOTS=opt (experiment, profiles/r05_bvh_quality_ab.txt): the assignment that MAXIMISES the summed projection (Ylitie et al. 2017 solve it with an
        // auction; with 8 x 8 an exact subset DP is 2 k steps per node) instead of the greedy best pair above
        if(optSlots && nc > 1) {
          float score[8][8];
          for(int i = 0; i < nc; i++) {
            const Box& cb = N[ch[i]].b;
            const float d[3] = {0.5f * (cb.lo[0] + cb.hi[0]) - cen[0], 0.5f * (cb.lo[1] + cb.hi[1]) - cen[1], 0.5f * (cb.lo[2] + cb.hi[2]) - cen[2]};
            for(int s2 = 0; s2 < 8; s2++) score[i][s2] = d[0] * slotSign(s2, 0) + d[1] * slotSign(s2, 1) + d[2] * slotSign(s2, 2);
          }
          // dp[mask] = best total for children 0 .. popcount(mask) - 1 placed in the slots of mask
          float dp[256]; int8_t from[256];
          for(int m = 0; m < 256; m++) { dp[m] = -3e38f; from[m] = -1; }
          dp[0] = 0.f;
          for(int m = 0; m < 256; m++) {
            const int i = __builtin_popcount(unsigned(m));
            if(i >= nc || dp[m] < -1e38f) continue;
            for(int s2 = 0; s2 < 8; s2++) {
              if(m & (1 << s2)) continue;
              const float v = dp[m] + score[i][s2];
              if(v > dp[m | (1 << s2)]) { dp[m | (1 << s2)] = v; from[m | (1 << s2)] = int8_t(s2); }
            }
          }
          int bestMask = -1; float bestV = -3e38f;
          for(int m = 0; m < 256; m++) if(__builtin_popcount(unsigned(m)) == nc && dp[m] > bestV) { bestV = dp[m]; bestMask = m; }
          for(int i = nc - 1, m = bestMask; i >= 0; i--) { const int s2 = from[m]; slotOf[i] = s2; m &= ~(1 << s2); }
        }
        for(int s = 0; s < 8; s++) X.childInSlot[s] = -1;
        for(int i = 0; i < nc; i++) X.childInSlot[slotOf[i]] = int8_t(i);
        int inner = 0, tris = 0;
        for(int i = 0; i < nc; i++) { if(N[ch[i]].leaf) tris += int(std::min<uint32_t>(N[ch[i]].n, 3u)); else inner++; }
        X.nInner = uint8_t(inner); X.nTris = uint8_t(tris);
      }
    });
    // ---- prefix sums in level order: where every node's internal children and leaf triangles go ----
    std::vector<uint32_t> childBase(L), triBase(L);
    uint32_t nodeTop = uint32_t(out.nodes.size()), triTop = uint32_t(out.tris.size());
    for(size_t wi = 0; wi < L; wi++) { childBase[wi] = nodeTop; triBase[wi] = triTop; nodeTop += wide[wi].nInner; triTop += wide[wi].nTris; }
    const uint32_t firstChild = uint32_t(out.nodes.size());
    out.nodes.resize(nodeTop);
    out.tris.resize(triTop);
    nextLevel.assign(size_t(nodeTop - firstChild), Work{0u, 0u});
    const size_t nChunks = (L + WCH - 1) / WCH;
    std::vector<double> sNode(nChunks, 0.0), sTri(nChunks, 0.0), sNodeQ(nChunks, 0.0), sTriQ(nChunks, 0.0);
    // ---- pass 2: quantise, write the node, copy its triangles, name its children ----
    parallelChunks(L, WCH, T, [&](size_t chunkId, size_t b0, size_t e0) {
      double aN = 0, aT = 0, aNQ = 0, aTQ = 0;
      for(size_t wi = b0; wi < e0; wi++) {
        const Work w = level[wi];
        const Wide& X = wide[wi];
        const uint32_t* ch = X.ch; const int nc = X.nc;
        const int8_t* childInSlot = X.childInSlot;
        Box nb; nb.reset();
        for(int i = 0; i < nc; i++) nb.grow(N[ch[i]].b);
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
        W.childBase = childBase[wi];
        W.triBase = triBase[wi];
        uint32_t triOff = 0, rel = 0;
        for(int s = 0; s < 8; s++) {
          W.qlox[s] = qlo[0][s]; W.qloy[s] = qlo[1][s]; W.qloz[s] = qlo[2][s];
          W.qhix[s] = qhi[0][s]; W.qhiy[s] = qhi[1][s]; W.qhiz[s] = qhi[2][s];
          if(childInSlot[s] < 0) { W.meta[s] = 0; continue; }
          const N2& c = N[ch[childInSlot[s]]];
          if(c.leaf) {
            // c.n <= 3 by construction, except the degenerate "root is one big leaf" case which cannot happen for n > 3
            uint32_t cntT = std::min<uint32_t>(c.n, 3u);
            W.meta[s] = uint8_t((((1u << cntT) - 1u) << 5) | triOff);
            for(uint32_t k = 0; k < cntT; k++) out.tris[size_t(triBase[wi]) + triOff + k] = flat[leafTris[c.a + k]];
            triOff += cntT;
          } else {
            W.imask |= uint8_t(1u << s);
            W.meta[s] = uint8_t((1u << 5) | (24u + uint32_t(s)));
            const uint32_t widx = childBase[wi] + rel; rel++;
            nextLevel[size_t(widx - firstChild)] = Work{ch[childInSlot[s]], widx};
          }
        }
        out.nodes[w.wide] = W;
        // SAH statistics of the WIDE tree (expected steps of a random ray that hits the root box): a node step per wide node whose box is hit, a triangle step
        // per triangle of a leaf slot whose box is hit
        aN += double(nb.area()) / ra;
        for(int i = 0; i < nc; i++) if(N[ch[i]].leaf) aT += double(N[ch[i]].b.area()) / ra * double(std::min<uint32_t>(N[ch[i]].n, 3u));
        // ... and the same expectation with the boxes the GPU tests: the children's boxes on this node's 8-bit grid (a thin triangle in a wide node grows to the grid step)
        for(int s2 = 0; s2 < 8; s2++) {
          if(childInSlot[s2] < 0) continue;
          const N2& cq = N[ch[childInSlot[s2]]];
          Box q;
          for(int a = 0; a < 3; a++) { const float step = std::ldexp(1.0f, ex[a]); q.lo[a] = nb.lo[a] + float(qlo[a][s2]) * step; q.hi[a] = nb.lo[a] + float(qhi[a][s2]) * step; }
          if(cq.leaf) aTQ += double(q.area()) / ra * double(std::min<uint32_t>(cq.n, 3u));
          else aNQ += double(q.area()) / ra;
        }
        if(w.wide == 0) aNQ += 1.0;   // the root itself
      }
      sNode[chunkId] = aN; sTri[chunkId] = aT; sNodeQ[chunkId] = aNQ; sTriQ[chunkId] = aTQ;
    });
    for(size_t c = 0; c < nChunks; c++) { out.sahNodeSteps += sNode[c]; out.sahTriSteps += sTri[c]; out.sahNodeStepsQ += sNodeQ[c]; out.sahTriStepsQ += sTriQ[c]; }
    level.swap(nextLevel);
  }
  timer.lap("collapse + wide nodes");
  return true;
}

}  // namespace rt

// ---- determinism pin (tests/test_bvh_quality.py; no GPU): the tree as a 64-bit FNV-1a hash of its node and leaf records, built with a given number of threads --------
// out[0] = hash, out[1] = nodes, out[2] = leaf records, out[3] = spatial splits, out[4] = rotations, out[5] = depth; seconds = build time
extern "C" int rt_bvh8_build_hash(const rt_scene_desc* scene, int threads, uint64_t* out, double* seconds)
{
  if(!scene || !out) return -1;
  rt::BuildOutput bo;
  const auto t0 = std::chrono::steady_clock::now();
  if(!rt::buildBvh8(*scene, bo, std::max(1, threads))) return -2;
  if(seconds) *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  uint64_t h = 1469598103934665603ull;
  auto eat = [&](const void* p, size_t bytes) { const uint8_t* b = static_cast<const uint8_t*>(p); for(size_t i = 0; i < bytes; i++) { h ^= b[i]; h *= 1099511628211ull; } };
  eat(bo.nodes.data(), bo.nodes.size() * sizeof(rt::Node8));
  eat(bo.tris.data(), bo.tris.size() * sizeof(rt::Tri48));
  eat(&bo.sahNodeSteps, sizeof(double)); eat(&bo.sahTriSteps, sizeof(double)); eat(&bo.sahNodeStepsQ, sizeof(double)); eat(&bo.sahTriStepsQ, sizeof(double));
  out[0] = h; out[1] = bo.nodes.size(); out[2] = bo.tris.size(); out[3] = bo.spatialSplits; out[4] = bo.rotations; out[5] = uint64_t(bo.maxDepth);
  return 0;
}

// ---- host-side self check of a built tree (tests/test_bvh_quality.py; no GPU) ----------------------------------------------------------------------------
// The property every parity claim rests on: the answer of a query is a function of the triangle set, never of the tree.  For that every point of
// every triangle must be reachable: walking down from the root through the child boxes that contain the point has to arrive at a leaf slot that holds the
// triangle.  With spatial splits a triangle has several references, each bounded by the part inside its cell: the check samples points on every triangle
// (vertices, edge and interior points) and counts the points that no reference covers.  Decodes the nodes the way csrc/traverse.h does.
extern "C" int rt_bvh8_selfcheck(const rt_scene_desc* scene, int samplesPerTri, uint64_t* out /* [8]: triangles, references, nodes, depth, spatial splits, uncovered points, points, 0 */,
                                 double* outF /* [5]: SAH node steps, SAH triangle steps, build seconds, the two SAH figures with the quantised child boxes */)
{
  if(!scene || !out || !outF) return -1;
  rt::BuildOutput bo;
  const auto t0 = std::chrono::steady_clock::now();
  const int threads = rt_cpu_budget();
  if(!rt::buildBvh8(*scene, bo, threads)) return -2;
  outF[2] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  outF[0] = bo.sahNodeSteps; outF[1] = bo.sahTriSteps; outF[3] = bo.sahNodeStepsQ; outF[4] = bo.sahTriStepsQ;
  const size_t n = bo.triRef.size();
  // geometry by globalId (any reference of a triangle carries its full record)
  std::vector<const rt::Tri48*> byId(n, nullptr);
  for(const rt::Tri48& T : bo.tris) if(T.globalId < n) byId[T.globalId] = &T;
  std::atomic<uint64_t> uncovered{0}, points{0};
  std::atomic<size_t> next{0};
  const int S = std::max(3, samplesPerTri);
  auto worker = [&] {
    std::vector<uint32_t> stack;
    for(;;) {
      const size_t g0 = next.fetch_add(1024);
      if(g0 >= n) return;
      for(size_t g = g0; g < std::min(n, g0 + 1024); g++) {
        const rt::Tri48* T = byId[g];
        if(!T) { uncovered += uint64_t(S); points += uint64_t(S); continue; }   // a triangle without any reference
        uint32_t rng = uint32_t(g) * 2654435761u + 12345u;
        for(int k = 0; k < S; k++) {
          float u, v;
          if(k == 0) { u = 0; v = 0; } else if(k == 1) { u = 1; v = 0; } else if(k == 2) { u = 0; v = 1; }
          else {
            rng = rng * 747796405u + 2891336453u; u = float(rng >> 8) * (1.0f / 16777216.0f);
            rng = rng * 747796405u + 2891336453u; v = float(rng >> 8) * (1.0f / 16777216.0f);
            if(u + v > 1.0f) { u = 1.0f - u; v = 1.0f - v; }
            if(k % 3 == 0) v = 0.0f; else if(k % 3 == 1) u = 1.0f - v;   // two thirds of the random points on edges
          }
          const double p[3] = {double(T->v0x) + double(u) * T->e1x + double(v) * T->e2x, double(T->v0y) + double(u) * T->e1y + double(v) * T->e2y,
                               double(T->v0z) + double(u) * T->e1z + double(v) * T->e2z};
          bool found = false;
          stack.clear(); stack.push_back(0u);
          while(!stack.empty() && !found) {
            const rt::Node8& N = bo.nodes[stack.back()]; stack.pop_back();
            const double sx = std::ldexp(1.0, int(N.ex) - 127), sy = std::ldexp(1.0, int(N.ey) - 127), sz = std::ldexp(1.0, int(N.ez) - 127);
            uint32_t rel = 0;
            for(int s = 0; s < 8 && !found; s++) {
              const bool inner = (N.imask >> s) & 1u;
              const uint32_t myRel = rel; if(inner) rel++;
              if(N.meta[s] == 0) continue;
              const bool in = p[0] >= double(N.px) + N.qlox[s] * sx && p[0] <= double(N.px) + N.qhix[s] * sx && p[1] >= double(N.py) + N.qloy[s] * sy &&
                              p[1] <= double(N.py) + N.qhiy[s] * sy && p[2] >= double(N.pz) + N.qloz[s] * sz && p[2] <= double(N.pz) + N.qhiz[s] * sz;
              if(!in) continue;
              if(inner) stack.push_back(N.childBase + myRel);
              else {
                const uint32_t cnt = uint32_t(__builtin_popcount(N.meta[s] >> 5)), off = N.meta[s] & 31u;
                for(uint32_t q = 0; q < cnt; q++) if(bo.tris[N.triBase + off + q].globalId == uint32_t(g)) found = true;
              }
            }
          }
          points++;
          if(!found) uncovered++;
        }
      }
    }
  };
  std::vector<std::thread> pool;
  for(int i = 0; i < threads; i++) pool.emplace_back(worker);
  for(auto& t : pool) t.join();
  out[0] = n; out[1] = bo.tris.size(); out[2] = bo.nodes.size(); out[3] = uint64_t(bo.maxDepth); out[4] = bo.spatialSplits; out[5] = uncovered.load(); out[6] = points.load(); out[7] = bo.rotations + (bo.reinsertions << 32);
  return 0;
}
