// traverse.h — software ray query on gfx950: ClosestHit / AnyHit of shaders/traceray_rq.glsl:108-185 over the flat
// BVH8 of bvh8.h.  One ray per lane, one wave64 per 8x8 pixel tile; the traversal stack lives in LDS
// (STACK_N x 64 lanes x 8 B, entry-major so each lane owns one 8-byte column => conflict-free ds_read/write_b64).
//
// Semantics kept from the reference's ray flags: tmin 0 (exclusive), gl_RayFlagsCullBackFacingTrianglesEXT,
// per-instance FORCE_OPAQUE / TRIANGLE_FACING_CULL_DISABLE (accelstruct.cpp:145-149), candidates of non-opaque
// instances go through HitTest (traceray_rq.glsl:32-102).  Closest hit = lexicographic minimum of (t, globalId);
// box tests use FMAs and are conservative w.r.t. the exact (non-contracted) triangle test.
#pragma once
#include "dev_math.h"
#include "dev_scene.h"
#include "stages.h"

namespace rt {



struct RayHit {
  float t;
  uint32_t gid;  // globalId of the hit triangle (0xffffffff = miss)
  float u, v;
};

// RT_WAVEPROF (compile-time, measurement builds only: scripts/wave_profile.py): per-wave cycle breakdown of the traversal rounds
#ifndef RT_WAVEPROF
#define RT_WAVEPROF 0
#endif
struct TravCounters {
  uint32_t nodes, tris, rounds, live;
#if RT_WAVEPROF
  uint32_t rN, rT, rC, cN, cT, cC, aTex, aOmm;   // rounds / cycles by kind (node, triangle, cooperative tail), alpha candidates by resolution
  uint32_t ph[8];                                // wide traversal: cycles by phase of the triangle step (see travTriW), [7] = steps
  uint32_t nph[4];                               // and of the node step (travRoundW), [3] = steps
  uint32_t hl[8], he[8];                         // throughput build: rounds by live lanes (1-8, 9-16, .. 57-64) and by lanes that EXECUTE the round (the majority kind)
#endif
};

// ---- texture fetch: Vulkan sampler restatement, LOD 0 (scene.cpp:513-548) -----------------
// i mod p, result in [0, p).  `%` with a run-time divisor is a ~30-instruction software division on the GPU; for a power of two
// (the usual texture size) the two's-complement mask is the same number.  The slow path sits behind a per-lane test, i.e. a
// wave that only meets power-of-two textures skips it (s_cbranch_execz).  Measured on the foliage-heavy benchmark scene, where the
// alpha test of the traversal samples a texture per unresolved candidate: direct stage -4 %, frame -2.5 %.
RT_DEV int floorMod(int i, int p)
{
  if((p & (p - 1)) == 0) return i & (p - 1);
  int m = i % p; if(m < 0) m += p;
  return m;
}
RT_DEV int wrapCoord(int i, int n, int mode)
{
  if(mode == RT_WRAP_CLAMP) return i < 0 ? 0 : (i >= n ? n - 1 : i);
  if(mode == RT_WRAP_MIRROR) {
    const int p = 2 * n;
    const int m = floorMod(i, p);
    return m < n ? m : p - 1 - m;
  }
  return floorMod(i, n);
}
// REPEAT on power-of-two sizes (every texture of the benchmark scenes, most glTF assets) is a mask; anything else takes the general path, whose integer
// modulo costs ~80 instructions per coordinate.  Callers choose per WAVE (a scalar branch: the general path is skipped, not predicated, when no active lane
// needs it); both paths return the same values.
RT_DEV bool wrapIsMask(int w, int h, int wrapS, int wrapT) { return (((w & (w - 1)) | (h & (h - 1))) == 0) && wrapS == RT_WRAP_REPEAT && wrapT == RT_WRAP_REPEAT; }
RT_DEV bool waveNone(bool c) { return __builtin_amdgcn_ballot_w64(c) == 0ull; }
RT_DEV f4 texelBGRA(const DevTexture& t, int x, int y)
{
  uint32_t p = gLoadU32(t.bgra + (size_t(y) * t.w + x) * 4);   // (t.bgra was loaded from the texture table: generic to the compiler)
  return mk4(unorm8ToFloat((p >> 16) & 0xffu), unorm8ToFloat((p >> 8) & 0xffu), unorm8ToFloat(p & 0xffu), unorm8ToFloat(p >> 24));
}
RT_DEV f4 sampleTexture(const DevScene& S, int id, f2 uv)
{
  const DevTexture t = S.textures[id];
  float fx = uv.x * float(t.w), fy = uv.y * float(t.h);
  const bool masks = waveNone(!wrapIsMask(t.w, t.h, t.wrapS, t.wrapT));
  if(t.filter == RT_FILTER_NEAREST) {
    const int xi = rt_ftoi(rt_floor(fx)), yi = rt_ftoi(rt_floor(fy));
    if(masks) return texelBGRA(t, xi & (t.w - 1), yi & (t.h - 1));
    return texelBGRA(t, wrapCoord(xi, t.w, t.wrapS), wrapCoord(yi, t.h, t.wrapT));
  }
  fx = fx - 0.5f; fy = fy - 0.5f;
  float x0f = rt_floor(fx), y0f = rt_floor(fy);
  float ax = fx - x0f, ay = fy - y0f;
  int x0 = rt_ftoi(x0f), y0 = rt_ftoi(y0f);
  int xa, xb, ya, yb;
  if(masks) { xa = x0 & (t.w - 1); xb = (x0 + 1) & (t.w - 1); ya = y0 & (t.h - 1); yb = (y0 + 1) & (t.h - 1); }
  else {
    xa = wrapCoord(x0, t.w, t.wrapS); xb = wrapCoord(x0 + 1, t.w, t.wrapS);
    ya = wrapCoord(y0, t.h, t.wrapT); yb = wrapCoord(y0 + 1, t.h, t.wrapT);
  }
  f4 top = mix(texelBGRA(t, xa, ya), texelBGRA(t, xb, ya), ax);
  f4 bot = mix(texelBGRA(t, xa, yb), texelBGRA(t, xb, yb), ax);
  return mix(top, bot, ay);
}
RT_DEV f4 envTexel(const DevScene& S, int x, int y)
{
  const float4 p = gLoadF4(S.env + (size_t(y) * S.envW + x) * 4);
  return mk4(p.x, p.y, p.z, p.w);
}
// environmentTexture: linear, U repeat, V clamp-to-edge (hdr_sampling.cpp:69-77)
RT_DEV f4 sampleEnv(const DevScene& S, f2 uv)
{
  float fx = uv.x * float(S.envW) - 0.5f, fy = uv.y * float(S.envH) - 0.5f;
  float x0f = rt_floor(fx), y0f = rt_floor(fy);
  float ax = fx - x0f, ay = fy - y0f;
  int x0 = rt_ftoi(x0f), y0 = rt_ftoi(y0f);
  int xa = wrapCoord(x0, S.envW, RT_WRAP_REPEAT), xb = wrapCoord(x0 + 1, S.envW, RT_WRAP_REPEAT);
  int ya = wrapCoord(y0, S.envH, RT_WRAP_CLAMP), yb = wrapCoord(y0 + 1, S.envH, RT_WRAP_CLAMP);
  f4 top = mix(envTexel(S, xa, ya), envTexel(S, xb, ya), ax);
  f4 bot = mix(envTexel(S, xa, yb), envTexel(S, xb, yb), ax);
  return mix(top, bot, ay);
}

RT_DEV uint32_t pcgNext(uint32_t& state)  // random.glsl:59-65
{
  uint32_t prev = state * 747796405u + 2891336453u;
  uint32_t word = ((prev >> ((prev >> 28u) + 4u)) ^ prev) * 277803737u;
  state = prev;
  return (word >> 22u) ^ word;
}
RT_DEV float rnd(uint32_t& seed)  // rand(), random.glsl:98-102
{
  uint32_t r = pcgNext(seed);
  return rt_u2f(0x3f800000u | (r >> 9)) - 1.0f;
}

// HitTest, traceray_rq.glsl:32-102, on the pre-gathered AlphaRec.  The stochastic draw comes from a hash of
// (ray seed, triangle id) instead of advancing prd.seed per candidate, so the outcome does not depend on candidate
// order.  Only the alpha channel of the bilinear fetch is evaluated (same arithmetic).
// seed of the draw for one (ray, triangle) pair; never the ray's own seed (triangle 0 included), so the pixel's next rand() does
// not repeat the alpha draw
RT_DEV uint32_t candidateSeed(uint32_t raySeed, uint32_t gid) { return (raySeed ^ 0x9e3779b9u) + (gid + 1u) * 2654435761u; }
// HitTest in three pieces so that the latency-mode traversal (RT_LAT) can issue the record loads of several candidates, then all their
// texel loads, before it evaluates any of them; hitTestAlpha() below is the three pieces back to back — same operations, same order.
struct AlphaRegs { uint4 r0, r1, r2, r3; };
struct AlphaFetch { const uint8_t* p00; const uint8_t* p10; const uint8_t* p01; const uint8_t* p11; float ax, ay; int kind; };   // kind: 0 no texture, 1 nearest (p00), 2 bilinear
RT_DEV AlphaRegs alphaLoad(const DevScene& S, uint32_t alphaIdx)
{
  const uint4* rp = reinterpret_cast<const uint4*>(S.alphaRec + alphaIdx);
  AlphaRegs A; A.r0 = gLoadU4(rp); A.r1 = gLoadU4(rp + 1); A.r2 = gLoadU4(rp + 2); A.r3 = gLoadU4(rp + 3);
  return A;
}
// texel addresses of the fetch; without a texture every address is `safe` (a readable dummy) so that callers may load unconditionally
RT_DEV AlphaFetch alphaAddr(const AlphaRegs& A, float u, float v, const uint8_t* safe)
{
  AlphaFetch F; F.p00 = F.p10 = F.p01 = F.p11 = safe; F.ax = F.ay = 0.0f; F.kind = 0;
  const uint8_t* bgra = reinterpret_cast<const uint8_t*>((uint64_t(A.r2.y) << 32) | uint64_t(A.r2.x));
  if(bgra) {
    const f3 bary = mk3((1.0f - u) - v, u, v);
    const f2 uv = (mk2(rt_u2f(A.r0.x), rt_u2f(A.r0.y)) * bary.x + mk2(rt_u2f(A.r0.z), rt_u2f(A.r0.w)) * bary.y) + mk2(rt_u2f(A.r1.x), rt_u2f(A.r1.y)) * bary.z;
    const int w = int(A.r2.z), h = int(A.r2.w), wrapS = int(A.r3.x), wrapT = int(A.r3.y), filter = int(A.r3.z);
    float fx = uv.x * float(w), fy = uv.y * float(h);
    const bool masks = waveNone(!wrapIsMask(w, h, wrapS, wrapT));
    if(filter == RT_FILTER_NEAREST) {
      const int xi = rt_ftoi(rt_floor(fx)), yi = rt_ftoi(rt_floor(fy));
      int x, y;
      if(masks) { x = xi & (w - 1); y = yi & (h - 1); }
      else { x = wrapCoord(xi, w, wrapS); y = wrapCoord(yi, h, wrapT); }
      F.p00 = F.p10 = F.p01 = F.p11 = bgra + (size_t(y) * w + x) * 4 + 3;
      F.kind = 1;
    } else {
      fx = fx - 0.5f; fy = fy - 0.5f;
      const float x0f = rt_floor(fx), y0f = rt_floor(fy);
      F.ax = fx - x0f; F.ay = fy - y0f;
      const int x0 = rt_ftoi(x0f), y0 = rt_ftoi(y0f);
      int xa, xb, ya, yb;
      if(masks) { xa = x0 & (w - 1); xb = (x0 + 1) & (w - 1); ya = y0 & (h - 1); yb = (y0 + 1) & (h - 1); }
      else {
        xa = wrapCoord(x0, w, wrapS); xb = wrapCoord(x0 + 1, w, wrapS);
        ya = wrapCoord(y0, h, wrapT); yb = wrapCoord(y0 + 1, h, wrapT);
      }
      F.p00 = bgra + (size_t(ya) * w + xa) * 4 + 3; F.p10 = bgra + (size_t(ya) * w + xb) * 4 + 3;
      F.p01 = bgra + (size_t(yb) * w + xa) * 4 + 3; F.p11 = bgra + (size_t(yb) * w + xb) * 4 + 3;
      F.kind = 2;
    }
  }
  return F;
}
RT_DEV bool alphaFinish(const AlphaRegs& A, const AlphaFetch& F, uint8_t a00, uint8_t a10, uint8_t a01, uint8_t a11, uint32_t gid, uint32_t raySeed)
{
  float baseColorAlpha = rt_u2f(A.r1.z);
  if(F.kind != 0) {
    float a;
    if(F.kind == 1) a = unorm8ToFloat(a00);
    else {
      const float top = mixf(unorm8ToFloat(a00), unorm8ToFloat(a10), F.ax);
      const float bot = mixf(unorm8ToFloat(a01), unorm8ToFloat(a11), F.ax);
      a = mixf(top, bot, F.ay);
    }
    baseColorAlpha = baseColorAlpha * a;
  }
  float opacity;
  if(int(A.r3.w) == RT_ALPHA_MASK) opacity = baseColorAlpha > rt_u2f(A.r1.w) ? 1.0f : 0.0f;
  else opacity = baseColorAlpha;
  uint32_t s = candidateSeed(raySeed, gid);
  const float r = rnd(s);
  return !(r > opacity);
}
RT_DEV bool hitTestAlpha(const DevScene& S, uint32_t alphaIdx, uint32_t gid, float u, float v, uint32_t raySeed)
{
  const AlphaRegs A = alphaLoad(S, alphaIdx);
  const AlphaFetch F = alphaAddr(A, u, v, reinterpret_cast<const uint8_t*>(S.alphaRec));
  uint8_t a00 = 0, a10 = 0, a01 = 0, a11 = 0;
  // (the texel addresses derive from AlphaRec::bgra, a pointer loaded from memory)
  if(F.kind == 1) a00 = gLoadU8(F.p00);
  else if(F.kind == 2) { a00 = gLoadU8(F.p00); a10 = gLoadU8(F.p10); a01 = gLoadU8(F.p01); a11 = gLoadU8(F.p11); }
  return alphaFinish(A, F, a00, a10, a01, a11, gid, raySeed);
}

// Möller–Trumbore in the fixed operation order of DESIGN.md §Numerics (this TU is compiled with -ffp-contract=off)
// The last test makes an accepted hit a property of (ray, triangle) alone: for a sliver (two vertices an ulp apart) the quotients
// above are rounding noise and can "hit" anywhere along the ray — whether such a triangle is tested at all would then depend on how
// the tree groups it.  A hit therefore only counts if its point o + t*d lies inside the triangle's bounding box widened by the
// build's box padding; every box of the tree contains that padded box, so no conservative box test can have culled it.
RT_DEV bool intersectTri(const Tri48& T, f3 o, f3 d, float& t, float& u, float& v)
{
  const f3 e1 = mk3(T.e1x, T.e1y, T.e1z), e2 = mk3(T.e2x, T.e2y, T.e2z), v0 = mk3(T.v0x, T.v0y, T.v0z);
  const f3 p = cross(d, e2);
  const float det = dot(e1, p);
  if(T.flags & TRI_NOCULL) { if(det == 0.0f || rt_isnan(det)) return false; }
  else {
    const float sdet = (T.flags & TRI_FLIP) ? -det : det;
    if(!(sdet > 0.0f)) return false;
  }
  const float inv = 1.0f / det;
  const f3 tv = o - v0;
  u = dot(tv, p) * inv;
  if(!(u >= 0.0f && u <= 1.0f)) return false;
  const f3 q = cross(tv, e1);
  v = dot(d, q) * inv;
  if(!(v >= 0.0f && u + v <= 1.0f)) return false;
  t = dot(e2, q) * inv;
  return !rt_isnan(t);
}
// second half of the test (see above).  The box is widened by the build's pad PLUS a term for the rounding error of the hit point itself,
// 2^-21 x (largest |origin component| + t x largest |direction component|): with the build's pad alone a camera far outside the scene lost genuine
// hits (the error of o + t*d grows with |o|; DESIGN.md deviation 6; tests/test_trace_pin.py).  Every candidate that survives the range / closer-than-best
// tests is checked.  (Checking only the FINAL hit of a closest-hit ray and re-tracing strictly when it fails is equivalent — a candidate accepted wrongly on
// the way can only be displaced by closer ones — and was built in round 3: the extra record fetch and the re-trace path cost more than the 35 instructions
// per surviving candidate they save: frame +3.3 %, indirect stage +8 %, profiles/r03_final_hit_ab.txt.)
RT_DEV float hitPointPad(f3 o, f3 d, float pad, float t)
{
  const float om = fmaxf(fmaxf(rt_abs(o.x), rt_abs(o.y)), rt_abs(o.z)), dm = fmaxf(fmaxf(rt_abs(d.x), rt_abs(d.y)), rt_abs(d.z));
  return pad + (om + t * dm) * 4.76837158203125e-07f;
}
RT_DEV bool hitInsidePaddedBox(const Tri48& T, f3 o, f3 d, float buildPad, float t)
{
  const float pad = hitPointPad(o, d, buildPad, t);
  const f3 e1 = mk3(T.e1x, T.e1y, T.e1z), e2 = mk3(T.e2x, T.e2y, T.e2z), v0 = mk3(T.v0x, T.v0y, T.v0z);
  // min(v0, v0 + e1, v0 + e2) = v0 + min(0, e1, e2) exactly (rounded addition is monotonic): one v_min3 / v_max3 per bound
  const f3 h = o + d * t;
  const f3 lo = mk3((v0.x + fminf(fminf(0.0f, e1.x), e2.x)) - pad, (v0.y + fminf(fminf(0.0f, e1.y), e2.y)) - pad, (v0.z + fminf(fminf(0.0f, e1.z), e2.z)) - pad);
  const f3 hi = mk3((v0.x + fmaxf(fmaxf(0.0f, e1.x), e2.x)) + pad, (v0.y + fmaxf(fmaxf(0.0f, e1.y), e2.y)) + pad, (v0.z + fmaxf(fmaxf(0.0f, e1.z), e2.z)) + pad);
  return h.x >= lo.x && h.x <= hi.x && h.y >= lo.y && h.y <= hi.y && h.z >= lo.z && h.z <= hi.z;
}

RT_DEV float cvtByte(uint32_t w, int j) { return float((w >> (8 * j)) & 0xffu); }  // v_cvt_f32_ubyte{j}

// Traversal state of one ray.  Work is advanced in two kinds of steps — travNode() (pop the nearest pending child, test
// its 8 children) and travTri() (test ONE pending triangle) — so that a wave can run, at every iteration, the kind of
// step most of its lanes are ready for.  With a fused "node, then all of its triangles" step, 84 % of the VALU lanes idled
// on incoherent rays (measured): a node exposes 0..24 triangles and the wave looped for the maximum over its lanes.
struct Trav {
  f3 o, d;
  float idx, idy, idz;
  uint32_t octinv;
  uint2 ngroup;    // pending children of the current node: (child base, hit bits 31..24 | internal mask 7..0)
  uint2 tgroup;    // pending triangles of the current node: (triangle base, hit bits 23..0)
  int sp;
  float tmax;      // ANY: ray tmax; closest: unused (RT_INFINITY)
  uint32_t seed;
  RayHit hit;
  bool found;
  bool isAny;      // MODE 2 only (rays of both kinds in one wave: tracePool)
};

RT_DEV bool travHasTris(const Trav& T) { return T.tgroup.y != 0u; }
RT_DEV bool travHasNodes(const Trav& T) { return T.ngroup.y > 0x00FFFFFFu || T.sp > 0; }

// ANY = false: closest hit in (0, 1e28); ANY = true: first accepted hit in (0, tmax).  Returns false when there is nothing to traverse.
// MODE 2: the kind is a per-ray run-time property (T.isAny, set by the caller before travInit).
template <int MODE>
RT_DEV bool travInit(Trav& T, f3 o, f3 d, float tmax, uint32_t raySeed)
{
  const bool ANY = MODE == 2 ? T.isAny : MODE == 1;
  T.o = o; T.d = d; T.tmax = tmax; T.seed = raySeed; T.found = false; T.sp = 0;
  T.hit.t = ANY ? tmax : RT_INFINITY;
  T.hit.gid = 0xffffffffu; T.hit.u = 0.f; T.hit.v = 0.f;
  T.ngroup = make_uint2(0u, 0u);
  T.tgroup = make_uint2(0u, 0u);
  if(hasNan(o) || hasNan(d) || !(T.hit.t > 0.0f)) return false;
  // reciprocal direction with a floor on |d| (a zero component must not produce inf*0 = NaN in the slab test)
  const float eps = 1e-20f;
  T.idx = 1.0f / (rt_abs(d.x) > eps ? d.x : (d.x < 0.0f ? -eps : eps));
  T.idy = 1.0f / (rt_abs(d.y) > eps ? d.y : (d.y < 0.0f ? -eps : eps));
  T.idz = 1.0f / (rt_abs(d.z) > eps ? d.z : (d.z < 0.0f ? -eps : eps));
  T.octinv = (((d.x < 0.0f) ? 1u : 0u) | ((d.y < 0.0f) ? 2u : 0u) | ((d.z < 0.0f) ? 4u : 0u)) ^ 7u;
  T.ngroup = make_uint2(0u, 0x80000000u);
  return true;
}

// The traversal stack of a lane: the first S.stackEntries entries live in LDS (most rays never go deeper), the rest of the S.stackTotal entries a
// ray of this tree can need in HBM, one small area per thread of the launch.  The short LDS stack is what lets LDS-staged filter kernels run beside
// the traversal kernels of the neighbouring frames in flight (profiles/r02_short_stack_ab.txt; a ring of the TOP entries in LDS was measured too:
// it needs a power-of-two size, and 4 KB per wave loses more to LDS pressure than the rarer HBM accesses gain).
// Round 6: the two homes of a stack entry are accessed through TYPED pointers (ldsLoadU2 / gLoadU2, dev_math.h) on two explicit paths.  Written as
// `cond ? stack[..] : ovf[..]` the compiler merged both into one generic pointer and every pop became flat_load_dwordx2 + s_waitcnt vmcnt(0) lgkmcnt(0) —
// a wait that also drains whatever node / triangle fetch is in flight; now the common pop is ds_read_b64 + lgkmcnt only (scripts/isa_census.py,
// profiles/r06_addrspace_ab.txt).
RT_DEV uint2* stackOverflowSlot(const DevScene& S, int sp)
{
  const size_t thread = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
  return S.stackOvf + (thread * size_t(S.stackTotal - S.stackEntries) + size_t(sp - S.stackEntries));
}
RT_DEV uint2 stackPop(const DevScene& S, Trav& T, uint2* stack)
{
  --T.sp;
  if(__builtin_expect(T.sp < S.stackEntries, 1)) return ldsLoadU2(stack + T.sp * 64);
  return gLoadU2(stackOverflowSlot(S, T.sp));
}
RT_DEV void stackPush(const DevScene& S, Trav& T, uint2* stack, uint2 g)
{
  if(__builtin_expect(T.sp < S.stackEntries, 1)) ldsStoreU2(stack + (T.sp++) * 64, g);
  else if(T.sp < S.stackTotal) { gStoreU2(stackOverflowSlot(S, T.sp), g); T.sp++; }   // (deeper than the tree: cannot happen, rt_build_accel sizes stackTotal)
}

// Node step (precondition: no pending triangles, travHasNodes).  `stack` = this lane's LDS column (stride 64 entries).
// In two halves — pick the nearest pending child (travNodeSelect: stack traffic only, yields the node to fetch), test the fetched node's eight
// children (travNodeTest) — so that the latency-mode round (RT_LAT) can put a round's triangle work between the fetch and the test.
struct NodeRegs { uint4 n0, n1, n2, n3, n4; };
RT_DEV uint32_t travNodeSelect(const DevScene& S, Trav& T, uint2* stack)
{
  uint2 ngroup = T.ngroup;
  if(ngroup.y <= 0x00FFFFFFu) ngroup = stackPop(S, T, stack);
  const uint32_t hits = ngroup.y;
  const uint32_t bit = 31u - uint32_t(__clz(int(hits)));
  ngroup.y &= ~(1u << bit);
  if(ngroup.y > 0x00FFFFFFu) stackPush(S, T, stack, ngroup);
  const uint32_t slot = (bit - 24u) ^ T.octinv;
  const uint32_t rel = uint32_t(__popc(hits & ~(0xFFFFFFFFu << slot) & 0xFFu));
  T.ngroup.y = 0u;   // consumed: what is left of the group sits on the stack
  return ngroup.x + rel;
}
RT_DEV NodeRegs nodeLoad(const DevScene& S, uint32_t index)
{
  const uint4* np = reinterpret_cast<const uint4*>(S.nodes + index);
  NodeRegs N; N.n0 = gLoadU4(np); N.n1 = gLoadU4(np + 1); N.n2 = gLoadU4(np + 2); N.n3 = gLoadU4(np + 3); N.n4 = gLoadU4(np + 4);
  return N;
}
RT_DEV void travNodeTest(Trav& T, const NodeRegs& N)
{
  const uint4 n0 = N.n0, n1 = N.n1, n2 = N.n2, n3 = N.n3, n4 = N.n4;
  const bool nx = T.d.x < 0.0f, ny = T.d.y < 0.0f, nz = T.d.z < 0.0f;
  const uint32_t octinv = T.octinv;
  const uint32_t octinv4 = octinv * 0x01010101u;
  const float adjx = rt_u2f((n0.w & 0xffu) << 23) * T.idx;
  const float adjy = rt_u2f(((n0.w >> 8) & 0xffu) << 23) * T.idy;
  const float adjz = rt_u2f(((n0.w >> 16) & 0xffu) << 23) * T.idz;
  const float orgx = (rt_u2f(n0.x) - T.o.x) * T.idx, orgy = (rt_u2f(n0.y) - T.o.y) * T.idy, orgz = (rt_u2f(n0.z) - T.o.z) * T.idz;
  const uint32_t imask = n0.w >> 24;
  uint32_t hitmask = 0;
#pragma unroll
  for(int half = 0; half < 2; half++) {
    const uint32_t meta4 = half ? n1.w : n1.z;
    const uint32_t isInner4 = (meta4 & (meta4 << 1)) & 0x10101010u;
    const uint32_t innerMask4 = (isInner4 >> 4) * 0xffu;
    const uint32_t bitIndex4 = (meta4 ^ (octinv4 & innerMask4)) & 0x1F1F1F1Fu;
    const uint32_t childBits4 = (meta4 >> 5) & 0x07070707u;
    const uint32_t qlx = half ? n2.y : n2.x, qly = half ? n2.w : n2.z, qlz = half ? n3.y : n3.x;
    const uint32_t qhx = half ? n3.w : n3.z, qhy = half ? n4.y : n4.x, qhz = half ? n4.w : n4.z;
    const uint32_t nearx = nx ? qhx : qlx, farx = nx ? qlx : qhx;
    const uint32_t neary = ny ? qhy : qly, fary = ny ? qly : qhy;
    const uint32_t nearz = nz ? qhz : qlz, farz = nz ? qlz : qhz;
#pragma unroll
    for(int j = 0; j < 4; j++) {
      const float tlx = __builtin_fmaf(cvtByte(nearx, j), adjx, orgx), thx = __builtin_fmaf(cvtByte(farx, j), adjx, orgx);
      const float tly = __builtin_fmaf(cvtByte(neary, j), adjy, orgy), thy = __builtin_fmaf(cvtByte(fary, j), adjy, orgy);
      const float tlz = __builtin_fmaf(cvtByte(nearz, j), adjz, orgz), thz = __builtin_fmaf(cvtByte(farz, j), adjz, orgz);
      const float tn = fmaxf(fmaxf(tlx, tly), fmaxf(tlz, 0.0f));
      const float tf = fminf(fminf(thx, thy), fminf(thz, T.hit.t));
      if(tn <= tf) hitmask |= ((childBits4 >> (8 * j)) & 0xffu) << ((bitIndex4 >> (8 * j)) & 0xffu);
    }
  }
  T.ngroup = make_uint2(n1.x, (hitmask & 0xFF000000u) | imask);
  T.tgroup = make_uint2(n1.y, hitmask & 0x00FFFFFFu);
}
RT_DEV void travNode(const DevScene& S, Trav& T, uint2* stack, TravCounters& tc)
{
  const uint32_t index = travNodeSelect(S, T, stack);
  const NodeRegs N = nodeLoad(S, index);
  tc.nodes++;
  travNodeTest(T, N);
}

// One triangle candidate of a ray: intersect, range / closest-so-far test, opacity (micro-map first, texture only when the
// micro-map cell is mixed).  The verdict depends on (ray, triangle) only — never on the order candidates are visited in.
struct TriRegs { uint4 a, b, c, om; };
RT_DEV TriRegs triLoad(const DevScene& S, uint32_t triIndex)
{
  const uint4* tp = reinterpret_cast<const uint4*>(S.tris + triIndex);
  TriRegs Q; Q.a = gLoadU4(tp); Q.b = gLoadU4(tp + 1); Q.c = gLoadU4(tp + 2); Q.om = gLoadU4(tp + 3);
  return Q;
}
// everything but the texture fetch: 0 = rejected, 1 = accepted, 2 = accepted iff HitTest on the triangle's AlphaRec (alphaIdx) accepts
RT_DEV int triCandidateGeom(const DevScene& S, const TriRegs& Q, f3 o, f3 d, bool ANY, float tmax, float curT, uint32_t curG, uint32_t seed, float& t, float& u, float& v,
                            uint32_t& gid, uint32_t& alphaIdx, TravCounters& tc)
{
  const uint4 a = Q.a, b = Q.b, c = Q.c, om = Q.om;
  Tri48 R;
  R.v0x = rt_u2f(a.x); R.v0y = rt_u2f(a.y); R.v0z = rt_u2f(a.z); R.e1x = rt_u2f(a.w);
  R.e1y = rt_u2f(b.x); R.e1z = rt_u2f(b.y); R.e2x = rt_u2f(b.z); R.e2y = rt_u2f(b.w);
  R.e2z = rt_u2f(c.x); R.globalId = c.y; R.flags = c.z; R.alphaIdx = c.w;
  gid = R.globalId; alphaIdx = R.alphaIdx;
  if(!intersectTri(R, o, d, t, u, v)) return 0;
  if(ANY) {
    if(!(t > 0.0f && t < tmax)) return 0;
  } else {
    if(!(t > 0.0f && t < RT_INFINITY)) return 0;
    if(!(t < curT || (t == curT && R.globalId < curG))) return 0;
  }
  if(!hitInsidePaddedBox(R, o, d, S.triPad, t)) return 0;
  if(!(R.flags & TRI_OPAQUE)) {
    // opacity micro-map first: most candidates resolve without touching the texture
    const int ci = min(int(u * 8.0f), 7), cj = min(int(v * 8.0f), 7);
    const int cell = cj * 8 + ci;
    const uint32_t word = (cell < 16) ? om.x : ((cell < 32) ? om.y : ((cell < 48) ? om.z : om.w));
    const uint32_t state = (word >> ((cell & 15) * 2)) & 3u;
#if RT_WAVEPROF
    if(state == 0u) tc.aTex++; else tc.aOmm++;
#else
    (void)tc;
#endif
    if(state == 1u) return 1;
    if(state == 2u) { uint32_t hs = candidateSeed(seed, R.globalId); return !(rnd(hs) > 0.0f) ? 1 : 0; }
    return 2;
  }
  return 1;
}
RT_DEV bool triCandidate(const DevScene& S, uint32_t triIndex, f3 o, f3 d, bool ANY, float tmax, float curT, uint32_t curG, uint32_t seed, float& t, float& u, float& v,
                         uint32_t& gid, TravCounters& tc)
{
  const TriRegs Q = triLoad(S, triIndex);
  uint32_t alphaIdx;
  const int s = triCandidateGeom(S, Q, o, d, ANY, tmax, curT, curG, seed, t, u, v, gid, alphaIdx, tc);
  if(s == 2) return hitTestAlpha(S, alphaIdx, gid, u, v, seed);
  return s == 1;
}

// Triangle step (precondition: travHasTris): test one pending triangle.
template <int MODE>
RT_DEV void travTri(const DevScene& S, Trav& T, TravCounters& tc)
{
  const bool ANY = MODE == 2 ? T.isAny : MODE == 1;
  const uint32_t bit = 31u - uint32_t(__clz(int(T.tgroup.y)));
  T.tgroup.y &= ~(1u << bit);
  tc.tris++;
  float t, u, v; uint32_t gid;
  if(!triCandidate(S, T.tgroup.x + bit, T.o, T.d, ANY, T.tmax, T.hit.t, T.hit.gid, T.seed, t, u, v, gid, tc)) return;
  T.hit.t = t; T.hit.gid = gid; T.hit.u = u; T.hit.v = v;
  T.found = true;
  if(ANY) { T.tgroup.y = 0u; T.ngroup.y = 0u; T.sp = 0; }  // first accepted hit terminates the query
}

RT_DEV float laneF(float v, int l) { return rt_u2f(uint32_t(__builtin_amdgcn_readlane(int(rt_f2u(v)), l))); }
RT_DEV uint32_t laneU(uint32_t v, int l) { return uint32_t(__builtin_amdgcn_readlane(int(v), l)); }

// Cooperative triangle step for the tail of a wave.  When only a few rays of a wave are still live, a round is bound by the
// latency of one dependent load, not by issue slots, and a ray that crosses foliage holds 5-20 pending triangles per node:
// one per round, it keeps its whole wave waiting.  Here every lane that is in the loop (live or not) takes ONE of the pending
// triangles of ray `l` — ray and triangle list are broadcast with readlane — and the verdicts are merged with the rule the
// sequential step applies one by one: closest = lexicographic minimum of (t, globalId); any-hit = the first accepted in list
// order.  Verdicts do not depend on visiting order (triCandidate), so the result is the one travTri would reach.
template <int MODE>
RT_DEV void travTriCoop(const DevScene& S, Trav& T, unsigned long long triMask, TravCounters& tc)
{
  const int lane = int(threadIdx.x) & 63;
  const unsigned long long present = __ballot(1);
  const int rank = __popcll(present & ((1ull << lane) - 1ull)), np = __popcll(present);
  while(triMask != 0ull) {
    const int l = __builtin_ctzll(triMask);
    triMask &= triMask - 1ull;
    const f3 o = mk3(laneF(T.o.x, l), laneF(T.o.y, l), laneF(T.o.z, l)), d = mk3(laneF(T.d.x, l), laneF(T.d.y, l), laneF(T.d.z, l));
    const uint32_t tbase = laneU(T.tgroup.x, l), seed = laneU(T.seed, l);
    uint32_t tbits = laneU(T.tgroup.y, l);
    const float tmax = laneF(T.tmax, l);
    const bool ANY = MODE == 2 ? (laneU(T.isAny ? 1u : 0u, l) != 0u) : MODE == 1;
    float bt = laneF(T.hit.t, l), bu = 0.0f, bv = 0.0f;
    uint32_t bg = laneU(T.hit.gid, l);
    bool found = false;
    if(lane == l) tc.tris += uint32_t(__popc(tbits));
    while(tbits != 0u && !(ANY && found)) {
      int mine = -1, j = 0;
      while(tbits != 0u && j < np) {  // the j-th pending triangle (list order: highest bit first) goes to the j-th lane present
        const int k = 31 - __clz(int(tbits));
        tbits &= ~(1u << k);
        if(rank == j) mine = k;
        j++;
      }
      float t = 0.0f, u = 0.0f, v = 0.0f; uint32_t gid = 0u;
      bool ok = false;
      if(mine >= 0) ok = triCandidate(S, tbase + uint32_t(mine), o, d, ANY, tmax, bt, bg, seed, t, u, v, gid, tc);
      unsigned long long okm = __ballot(ok ? 1 : 0);
      while(okm != 0ull) {
        const int w = __builtin_ctzll(okm);
        okm &= okm - 1ull;
        const float wt = laneF(t, w); const uint32_t wg = laneU(gid, w);
        if(ANY || wt < bt || (wt == bt && wg < bg)) { bt = wt; bg = wg; bu = laneF(u, w); bv = laneF(v, w); found = true; }
        if(ANY) break;  // list order = rank order = lane order: the lowest accepted lane is the one travTri would have stopped at
      }
    }
    if(lane == l) {
      T.tgroup.y = 0u;
      if(found) {
        T.hit.t = bt; T.hit.gid = bg; T.hit.u = bu; T.hit.v = bv; T.found = true;
        if(ANY) { T.ngroup.y = 0u; T.sp = 0; }
      }
    }
  }
}

// One scheduling round for a (partial) wave: every lane that is `live` votes for the kind of step it is ready for; the
// majority kind runs, the other lanes wait one round.  Returns whether this lane still has work.
template <int ANY>
RT_DEV bool travRound(const DevScene& S, Trav& T, bool live, uint2* stack, TravCounters& tc)
{
  tc.rounds++; tc.live += live ? 1u : 0u;
  const bool wantTri = live && travHasTris(T);
  const bool wantNode = live && !wantTri;
  const int nT = __popcll(__ballot(wantTri ? 1 : 0)), nN = __popcll(__ballot(wantNode ? 1 : 0));
  // (running both kinds every round, or the minority kind above a 25 % threshold, measured 5-10 % slower in both the
  // throughput-bound full frame and the latency-bound row-band case)
  if(nT >= nN) { if(wantTri) travTri<ANY>(S, T, tc); }
  else { if(wantNode) travNode(S, T, stack, tc); }
  return live && (travHasTris(T) || travHasNodes(T));
}

// travRound for callers that already hold the wave's live mask (saves one ballot per round)
template <int ANY>
RT_DEV bool travRoundMasked(const DevScene& S, Trav& T, bool live, unsigned long long liveMask, uint2* stack, TravCounters& tc)
{
  tc.rounds++; tc.live += live ? 1u : 0u;
  const bool wantTri = live && travHasTris(T);
  const unsigned long long triMask = __ballot(wantTri ? 1 : 0);
  const int nLive = __popcll(liveMask), nT = __popcll(triMask), nN = nLive - nT;
#if RT_WAVEPROF
  const uint64_t c0 = clock64();
#endif
  if(nLive <= S.coopLive) {  // tail of the wave (wave-uniform): all pending triangles at once, then a node step for every live ray
    if(nT > 0) travTriCoop<ANY>(S, T, triMask, tc);
    if(live && travHasNodes(T)) travNode(S, T, stack, tc);
  } else if(nT >= nN) { if(wantTri) travTri<ANY>(S, T, tc); }
  else { if(live && !wantTri) travNode(S, T, stack, tc); }
#if RT_WAVEPROF
  {
    const uint32_t dc = uint32_t(clock64() - c0);
    if(nLive <= S.coopLive) { tc.rC++; tc.cC += dc; } else if(nT >= nN) { tc.rT++; tc.cT += dc; } else { tc.rN++; tc.cN += dc; }
    const int ex = nLive <= S.coopLive ? nLive : (nT >= nN ? nT : nN);
    tc.hl[(nLive - 1) >> 3]++; tc.he[(max(ex, 1) - 1) >> 3]++;
  }
#endif
  return live && (travHasTris(T) || travHasNodes(T));
}

template <int ANY>
RT_DEV bool traceRay(const DevScene& S, f3 o, f3 d, float tmax, uint32_t raySeed, uint2* stack, RayHit& hit, TravCounters& tc)
{
  Trav T;
  bool live = travInit<ANY>(T, o, d, tmax, raySeed);
  // all lanes that entered this call stay in the loop until the last one is done, so the ballots of travRound see them
  // (travRound with its ballots shared between the vote and the loop condition: two per round instead of three)
  unsigned long long liveMask = __ballot(live ? 1 : 0);
  while(liveMask != 0ull) {
    live = travRoundMasked<ANY>(S, T, live, liveMask, stack, tc);
    liveMask = __ballot(live ? 1 : 0);
  }
  hit = T.hit;
  return T.found;
}

// ---- wave-level ray pool ---------------------------------------------------------------------------------------
// A path vertex spawns up to two independent rays (the NEE shadow ray and the BSDF bounce ray); in a multi-bounce tile many
// lanes are dead (path left the scene) while the live ones would trace their two rays back to back.  tracePool() lets the 64
// lanes of the wave share the wave's rays: lane L parks its rays in LDS slots 2L (closest-hit) and 2L+1 (any-hit), the set
// slots are compacted into a work list, and every lane — dead or alive — pulls the next ray whenever its current one
// finishes.  Results come back through the same slots.  A ray's result does not depend on the lane that traced it (the
// HitTest RNG is keyed by ray seed and triangle id), so the frame is unchanged bit for bit.
#ifndef RT_POOL_ORDER
#define RT_POOL_ORDER 0
#endif
constexpr int POOL_SLOT_F4 = 2;                 // 32 B per slot: ray (o.xyz, d.x | d.yz, tmax, seed) then result (t, gid, u, v)
constexpr int POOL_BYTES = 128 * 32 + 128;      // 128 slots + the work list (u8 slot ids)

RT_DEV void poolPut(float4* pool, int slot, f3 o, f3 d, float tmax, uint32_t seed)
{
  pool[slot * POOL_SLOT_F4] = make_float4(o.x, o.y, o.z, d.x);
  pool[slot * POOL_SLOT_F4 + 1] = make_float4(d.y, d.z, tmax, rt_u2f(seed));
}
RT_DEV RayHit poolGet(const float4* pool, int slot)
{
  const float4 r = pool[slot * POOL_SLOT_F4];
  RayHit h; h.t = r.x; h.gid = rt_f2u(r.y); h.u = r.z; h.v = r.w;
  return h;
}
RT_DEV void waveLdsSync() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); }

// Every lane of the wave must call this (uniform control flow).  hasC / hasS: this lane parked a closest-hit / any-hit ray.
RT_DEV void tracePool(const DevScene& S, float4* pool, bool hasC, bool hasS, uint2* stack, TravCounters& tc)
{
  unsigned char* list = reinterpret_cast<unsigned char*>(pool + 128 * POOL_SLOT_F4);
  const int lane = int(threadIdx.x) & 63;
  const unsigned long long lt = (1ull << lane) - 1ull;
  const unsigned long long mC = __ballot(hasC ? 1 : 0), mS = __ballot(hasS ? 1 : 0);
  const int nC = __popcll(mC), n = nC + __popcll(mS);
  if(n == 0) return;
#if RT_POOL_ORDER & 2
  // experiment (round 5, profiles/r05_pool_order_ab.txt): the shadow rays first — they end at their first accepted hit — then the bounce rays
  const int nS = n - nC;
  if(hasS) list[__popcll(mS & lt)] = (unsigned char)(lane * 2 + 1);
  if(hasC) list[nS + __popcll(mC & lt)] = (unsigned char)(lane * 2);
#else
  if(hasC) list[__popcll(mC & lt)] = (unsigned char)(lane * 2);
  if(hasS) list[nC + __popcll(mS & lt)] = (unsigned char)(lane * 2 + 1);
#endif
  waveLdsSync();
  int next = 0, mySlot = 0;
  bool live = false;
  Trav T;
  for(;;) {
    unsigned long long liveMask = __ballot(live ? 1 : 0);
    const unsigned long long idle = ~liveMask;   // every lane of the wave is in this loop
    if(next < n && idle != 0ull) {
      const int item = next + __popcll(idle & lt);
      if(!live && item < n) {
        mySlot = int(list[item]);
        const float4 a = pool[mySlot * POOL_SLOT_F4], b = pool[mySlot * POOL_SLOT_F4 + 1];
        T.isAny = (mySlot & 1) != 0;
        live = travInit<2>(T, mk3(a.x, a.y, a.z), mk3(a.w, b.x, b.y), b.z, rt_f2u(b.w));
        if(!live) pool[mySlot * POOL_SLOT_F4] = make_float4(T.hit.t, rt_u2f(T.hit.gid), T.hit.u, T.hit.v);
      }
      next = min(n, next + __popcll(idle));
      liveMask = __ballot(live ? 1 : 0);
    }
    if(liveMask == 0ull) {
      if(next >= n) break;
      continue;
    }
    const bool still = travRoundMasked<2>(S, T, live, liveMask, stack, tc);
    if(live && !still) pool[mySlot * POOL_SLOT_F4] = make_float4(T.hit.t, rt_u2f(T.hit.gid), T.hit.u, T.hit.v);
    live = still;
  }
  waveLdsSync();
}

// The same pool for rays of ONE kind parked by K pixels per lane (slot k*64 + lane, K <= 4): used by the multi-tile
// single-bounce body of the indirect stage, where a wave owns K tiles so that the pool holds more rays than the wave has
// lanes and a lane whose ray ends early pulls another one instead of waiting for the slowest ray of the wave.
template <int K, bool ANY>
RT_DEV void tracePoolTiles(const DevScene& S, float4* pool, uint32_t have /* bit k: this lane parked a ray for its k-th pixel */, uint2* stack, TravCounters& tc)
{
  unsigned char* list = reinterpret_cast<unsigned char*>(pool + K * 64 * POOL_SLOT_F4);
  const int lane = int(threadIdx.x) & 63;
  const unsigned long long lt = (1ull << lane) - 1ull;
  int n = 0;
#if RT_POOL_ORDER & 1
  // experiment (round 5, profiles/r05_pool_order_ab.txt): the work list ordered by direction octant, so that the rays a wave traces at the same time head the
  // same way through the tree (the pool of a K-tile wave holds up to K x 64 rays of incoherent directions)
  uint32_t oct = 0;
#pragma unroll
  for(int k = 0; k < K; k++) if((have >> k) & 1u) {
    const float4 a = pool[(k * 64 + lane) * POOL_SLOT_F4], b = pool[(k * 64 + lane) * POOL_SLOT_F4 + 1];
    oct |= ((a.w < 0.0f ? 1u : 0u) | (b.x < 0.0f ? 2u : 0u) | (b.y < 0.0f ? 4u : 0u)) << (4 * k);
  }
  for(uint32_t oc = 0; oc < 8u; oc++) {
#pragma unroll
    for(int k = 0; k < K; k++) {
      const bool h = ((have >> k) & 1u) && ((oct >> (4 * k)) & 7u) == oc;
      const unsigned long long m = __ballot(h ? 1 : 0);
      if(h) list[n + __popcll(m & lt)] = (unsigned char)(k * 64 + lane);
      n += __popcll(m);
    }
  }
#else
#pragma unroll
  for(int k = 0; k < K; k++) {
    const bool h = (have >> k) & 1u;
    const unsigned long long m = __ballot(h ? 1 : 0);
    if(h) list[n + __popcll(m & lt)] = (unsigned char)(k * 64 + lane);
    n += __popcll(m);
  }
#endif
  if(n == 0) return;
  waveLdsSync();
  int next = 0, mySlot = 0;
  bool live = false;
  Trav T;
  for(;;) {
    unsigned long long liveMask = __ballot(live ? 1 : 0);
    const unsigned long long idle = ~liveMask;   // every lane of the wave is in this loop
    if(next < n && idle != 0ull) {
      const int item = next + __popcll(idle & lt);
      if(!live && item < n) {
        mySlot = int(list[item]);
        const float4 a = pool[mySlot * POOL_SLOT_F4], b = pool[mySlot * POOL_SLOT_F4 + 1];
        live = travInit<ANY>(T, mk3(a.x, a.y, a.z), mk3(a.w, b.x, b.y), b.z, rt_f2u(b.w));
        if(!live) pool[mySlot * POOL_SLOT_F4] = make_float4(T.hit.t, rt_u2f(T.hit.gid), T.hit.u, T.hit.v);
      }
      next = min(n, next + __popcll(idle));
      liveMask = __ballot(live ? 1 : 0);
    }
    if(liveMask == 0ull) {
      if(next >= n) break;
      continue;
    }
    const bool still = travRoundMasked<ANY>(S, T, live, liveMask, stack, tc);
    if(live && !still) pool[mySlot * POOL_SLOT_F4] = make_float4(T.hit.t, rt_u2f(T.hit.gid), T.hit.u, T.hit.v);
    live = still;
  }
  waveLdsSync();
}

// ---- latency-mode traversal: eight lanes per ray (RT_LAT = 1: the traced kernels of SMALL launches, csrc/stages_lat.hip) ------------------------------
// A row band of a multi-GPU frame (or a small image) puts about one wave on a SIMD, and its launch takes what its slowest wave takes.  Measured on the
// horizon bands of the benchmark frame (profiles/r03a_wave_profile_baseline.txt, profiles/r03_lat_narrow_band_ab.txt): a wave64 instruction costs its
// wave ~4 cycles however empty the SIMD is, so a ray's chain is (node steps + triangle steps) x (memory latency + the step's OWN instruction time), and
// with one ray per lane a node step is 230 instructions, a triangle step 170 per triangle — while 1000 SIMDs idle.  Here a ray owns EIGHT lanes:
//   * node step: all eight lanes fetch the node (one request), lane j decodes and slab-tests child j, the hit bits are OR-ed with three DPP moves
//     (~70 instructions instead of 230);
//   * triangle step: lane j tests the j-th pending triangle of the leaf group (8 at a time, each with its own alpha chain), the closest accepted candidate
//     is the minimum of (t, id) over the group (three DPP exchange steps);
//   * the ray's state is replicated in its eight lanes, its stack is one column in LDS.
// A wave carries 8 rays; the waves of a workgroup share an LDS ray pool and pull rays group by group as theirs finish (tracePoolWide).  Verdicts are functions
// of (ray, triangle) and the closest hit is a minimum over (t, id), so the result is bit-identical to the one-lane-per-ray traversal.
#ifndef RT_LAT
#define RT_LAT 0
#endif
#if RT_LAT
constexpr int WIDE_G = 8;            // lanes per ray
constexpr int WIDE_RAYS = 64 / WIDE_G;  // rays per wave = stack columns per wave
// DPP controls (gfx9): quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror (lane i <-> 7 - i inside each group of eight)
#define RT_DPP_XOR1 0xB1
#define RT_DPP_XOR2 0x4E
#define RT_DPP_HALF_MIRROR 0x141
template <int CTRL> RT_DEV uint32_t dppU(uint32_t v) { return uint32_t(__builtin_amdgcn_update_dpp(0, int(v), CTRL, 0xf, 0xf, true)); }
template <int CTRL> RT_DEV float dppF(float v) { return rt_u2f(dppU<CTRL>(rt_f2u(v))); }
// OR over the eight lanes of a group (all eight must be active)
RT_DEV uint32_t groupOr(uint32_t v)
{
  v |= dppU<RT_DPP_XOR1>(v);
  v |= dppU<RT_DPP_XOR2>(v);
  v |= dppU<RT_DPP_HALF_MIRROR>(v);
  return v;
}
struct WideCand { float t; uint32_t gid; float u, v; };   // t = +inf (IEEE): no candidate
template <int CTRL> RT_DEV void candMinStep(WideCand& c)
{
  const float ot = dppF<CTRL>(c.t); const uint32_t og = dppU<CTRL>(c.gid); const float ou = dppF<CTRL>(c.u), ov = dppF<CTRL>(c.v);
  const bool take = ot < c.t || (ot == c.t && og < c.gid);
  c.t = take ? ot : c.t; c.gid = take ? og : c.gid; c.u = take ? ou : c.u; c.v = take ? ov : c.v;
}
// lexicographic minimum of (t, gid) over the group, with its (u, v)
RT_DEV void groupMinCand(WideCand& c) { candMinStep<RT_DPP_XOR1>(c); candMinStep<RT_DPP_XOR2>(c); candMinStep<RT_DPP_HALF_MIRROR>(c); }

// the ray's stack: one 8-byte column per ray, stride WIDE_RAYS entries (every lane of the group reads the same address; lane 0 of the group writes)
RT_DEV uint2 stackPopW(Trav& T, const uint2* stack) { --T.sp; return ldsLoadU2(stack + T.sp * WIDE_RAYS); }
RT_DEV void stackPushW(Trav& T, uint2* stack, uint2 g, int j) { if(j == 0) ldsStoreU2(stack + T.sp * WIDE_RAYS, g); T.sp++; }
RT_DEV uint32_t travNodeSelectW(Trav& T, uint2* stack, int j)
{
  uint2 ngroup = T.ngroup;
  if(ngroup.y <= 0x00FFFFFFu) ngroup = stackPopW(T, stack);
  const uint32_t hits = ngroup.y;
  const uint32_t bit = 31u - uint32_t(__clz(int(hits)));
  ngroup.y &= ~(1u << bit);
  if(ngroup.y > 0x00FFFFFFu) stackPushW(T, stack, ngroup, j);
  const uint32_t slot = (bit - 24u) ^ T.octinv;
  const uint32_t rel = uint32_t(__popc(hits & ~(0xFFFFFFFFu << slot) & 0xFFu));
  T.ngroup.y = 0u;
  return ngroup.x + rel;
}
RT_DEV uint32_t byteOf(uint32_t lo, uint32_t hi, int j) { return ((j < 4 ? lo : hi) >> (8 * (j & 3))) & 0xffu; }
// lane j of the group tests child j of the fetched node (the arithmetic of travNodeTest for that child)
RT_DEV void travNodeTestW(Trav& T, const NodeRegs& N, int j)
{
  const uint4 n0 = N.n0, n1 = N.n1, n2 = N.n2, n3 = N.n3, n4 = N.n4;
  const bool nx = T.d.x < 0.0f, ny = T.d.y < 0.0f, nz = T.d.z < 0.0f;
  const float adjx = rt_u2f((n0.w & 0xffu) << 23) * T.idx;
  const float adjy = rt_u2f(((n0.w >> 8) & 0xffu) << 23) * T.idy;
  const float adjz = rt_u2f(((n0.w >> 16) & 0xffu) << 23) * T.idz;
  const float orgx = (rt_u2f(n0.x) - T.o.x) * T.idx, orgy = (rt_u2f(n0.y) - T.o.y) * T.idy, orgz = (rt_u2f(n0.z) - T.o.z) * T.idz;
  const uint32_t imask = n0.w >> 24;
  const uint32_t meta = byteOf(n1.z, n1.w, j);
  const uint32_t isInner = (meta & (meta << 1)) & 0x10u;
  const uint32_t bitIndex = (meta ^ (isInner ? T.octinv : 0u)) & 0x1Fu;
  const uint32_t childBits = (meta >> 5) & 0x07u;
  const float qlx = float(byteOf(n2.x, n2.y, j)), qly = float(byteOf(n2.z, n2.w, j)), qlz = float(byteOf(n3.x, n3.y, j));
  const float qhx = float(byteOf(n3.z, n3.w, j)), qhy = float(byteOf(n4.x, n4.y, j)), qhz = float(byteOf(n4.z, n4.w, j));
  const float tlx = __builtin_fmaf(nx ? qhx : qlx, adjx, orgx), thx = __builtin_fmaf(nx ? qlx : qhx, adjx, orgx);
  const float tly = __builtin_fmaf(ny ? qhy : qly, adjy, orgy), thy = __builtin_fmaf(ny ? qly : qhy, adjy, orgy);
  const float tlz = __builtin_fmaf(nz ? qhz : qlz, adjz, orgz), thz = __builtin_fmaf(nz ? qlz : qhz, adjz, orgz);
  const float tn = fmaxf(fmaxf(tlx, tly), fmaxf(tlz, 0.0f));
  const float tf = fminf(fminf(thx, thy), fminf(thz, T.hit.t));
  const uint32_t hitmask = groupOr((tn <= tf) ? (childBits << bitIndex) : 0u);
  T.ngroup = make_uint2(n1.x, (hitmask & 0xFF000000u) | imask);
  T.tgroup = make_uint2(n1.y, hitmask & 0x00FFFFFFu);
}
// lane j tests the j-th pending triangle (list order: highest bit first); up to eight per step, the rest stay pending
template <int MODE>
RT_DEV void travTriW(const DevScene& S, Trav& T, int j, TravCounters& tc)
{
  const bool ANY = MODE == 2 ? T.isAny : MODE == 1;
  uint32_t rem = T.tgroup.y;
  int mine = -1;
#pragma unroll
  for(int k = 0; k < WIDE_G; k++) {
    if(rem != 0u) {
      const int b = 31 - __clz(int(rem));
      if(k == j) mine = b;
      rem &= ~(1u << b);
    }
  }
  T.tgroup.y = rem;
  WideCand c; c.t = __builtin_huge_valf(); c.gid = 0xffffffffu; c.u = 0.0f; c.v = 0.0f;
  bool ok = false;
#if RT_WAVEPROF
#define RT_PH(k) { __builtin_amdgcn_s_waitcnt(0); const uint64_t pn = clock64(); tc.ph[k] += uint32_t(pn - pc); pc = pn; }
  uint64_t pc = clock64(); tc.ph[7]++;
#else
#define RT_PH(k)
#endif
  if(mine >= 0) {
    float t, u, v; uint32_t gid, alphaIdx;
    const uint32_t ti = T.tgroup.x + uint32_t(mine);
    const TriRegs Q = triLoad(S, ti);
    const uint4* ap = reinterpret_cast<const uint4*>(S.alphaByTri + ti);   // fetched with the record, whether needed or not: no second dependent access
    AlphaRegs A; A.r0 = gLoadU4(ap); A.r1 = gLoadU4(ap + 1); A.r2 = gLoadU4(ap + 2); A.r3 = gLoadU4(ap + 3);
    RT_PH(0)   // record + AlphaRec arrive
    const int s = triCandidateGeom(S, Q, T.o, T.d, ANY, T.tmax, T.hit.t, T.hit.gid, T.seed, t, u, v, gid, alphaIdx, tc);
    ok = s == 1;
    RT_PH(1)   // intersection, padded-box check, opacity micro-map
    if(s == 2) {
      const AlphaFetch Fh = alphaAddr(A, u, v, reinterpret_cast<const uint8_t*>(S.alphaRec));
      RT_PH(2) // texel addresses
      uint8_t a00 = 0, a10 = 0, a01 = 0, a11 = 0;
      if(Fh.kind == 1) a00 = gLoadU8(Fh.p00);
      else if(Fh.kind == 2) { a00 = gLoadU8(Fh.p00); a10 = gLoadU8(Fh.p10); a01 = gLoadU8(Fh.p01); a11 = gLoadU8(Fh.p11); }
      RT_PH(3) // texels arrive
      ok = alphaFinish(A, Fh, a00, a10, a01, a11, gid, T.seed);
      RT_PH(4) // filter + draw
    }
    if(ok) { c.t = t; c.gid = gid; c.u = u; c.v = v; }
  }
  // (all eight lanes of the group are back together here)
  RT_PH(5)
  groupMinCand(c);
  const bool any = c.t < __builtin_huge_valf();   // an accepted candidate has a finite t
  if(any && (ANY || c.t < T.hit.t || (c.t == T.hit.t && c.gid < T.hit.gid))) {
    T.hit.t = c.t; T.hit.gid = c.gid; T.hit.u = c.u; T.hit.v = c.v;
    T.found = true;
    if(ANY) { T.tgroup.y = 0u; T.ngroup.y = 0u; T.sp = 0; }
  }
  RT_PH(6)     // minimum over the group, hit update
#undef RT_PH
  (void)ok;
}
// one round of a wave: a ray wants a triangle step if it has pending triangles, a node step otherwise; the wave takes the kind most of its rays want
template <int MODE>
RT_DEV bool travRoundW(const DevScene& S, Trav& T, bool live, int j, uint2* stack, TravCounters& tc)
{
  bool wantTri = live && travHasTris(T);
  bool wantNode = live && !wantTri;
  {  // One kind of step per round, by vote of the wave's rays; the rays that wanted the other kind wait a round.  A round that serves both kinds costs the
     // sum of the two (the groups diverge), i.e. a ray's cheap node steps are billed at the price of another ray's triangle step: with the vote the direct
     // stage of a band takes 10-15 % less, the indirect stage 3-8 % (profiles/r03_lat_wide_ab.txt).  The result does not depend on the order of the steps.
    const int nT = __popcll(__ballot(wantTri ? 1 : 0)), nN = __popcll(__ballot(wantNode ? 1 : 0));
    if(nT >= nN) wantNode = false; else wantTri = false;
  }
  if(wantTri) travTriW<MODE>(S, T, j, tc);
#if RT_WAVEPROF
  if(wantNode) {   // ph[0..2] of node rounds: stack pop + select | node arrives | test
    __builtin_amdgcn_s_waitcnt(0); const uint64_t q0 = clock64();
    const uint32_t ni = travNodeSelectW(T, stack, j);
    __builtin_amdgcn_s_waitcnt(0); const uint64_t q1 = clock64();
    const NodeRegs N = nodeLoad(S, ni);
    __builtin_amdgcn_s_waitcnt(0); const uint64_t q2 = clock64();
    travNodeTestW(T, N, j);
    const uint64_t q3 = clock64();
    tc.nph[0] += uint32_t(q1 - q0); tc.nph[1] += uint32_t(q2 - q1); tc.nph[2] += uint32_t(q3 - q2); tc.nph[3]++;
  }
#else
  if(wantNode) { const NodeRegs N = nodeLoad(S, travNodeSelectW(T, stack, j)); travNodeTestW(T, N, j); }
#endif
  return live && (travHasTris(T) || travHasNodes(T));
}

// ---- gang mode: the tail of a wave ------------------------------------------------------------------------------------------------------------------
// When the pool is dry and a wave still holds a few long rays, its other ray slots idle while those rays walk on, one dependent step at a time: the tail of
// a tile — and of a small launch — is the chain of its slowest ray (370 steps through the foliage of the benchmark scene).  In gang mode the idle groups of
// the wave become WORKERS of the live rays: a ray's pending work lives entirely on its LDS stack column, every worker (the owner group included) takes a
// DIFFERENT pending child of it per round — worker i the i-th nearest of the top two stack entries —, tests that node's eight children with its eight
// lanes, pushes the hit group back and tests its node's leaf triangles itself; after a triangle round the workers of a ray merge their best hits through an
// LDS mailbox (minimum over (t, id); any-hit: any accepted).  A round then advances a ray by up to eight nodes instead of one.  Verdicts are functions of
// (ray, triangle), the closest hit a minimum over (t, id), boxes are only culled against hits already found: the result does not depend on the order the
// nodes are visited in, so the frame is bit-identical with and without (tests: every latency-build parity case).  S.gangMax = 0 switches it off.
constexpr int GANG_BOX_UINT2 = 16;   // per wave, behind its stacks: 8 x float4 mailbox
constexpr int GANG_SCAN = 2;         // stack entries a round looks at for pending children
constexpr int GANG_EXTRA = 20;       // stack entries per ray on top of the tree's own bound: the room parallel expansion may use
RT_DEV uint32_t leaderBits(unsigned long long m) { return uint32_t(((m & 0x0101010101010101ull) * 0x0102040810204080ull) >> 56); }   // lanes 0, 8, .., 56 -> bits 0..7
RT_DEV int nthSetBitFromTop(uint32_t v, int n)   // position of the n-th (0 = highest) set bit of v; v has more than n bits set
{
  for(int k = 0; k < n; k++) v &= ~(1u << (31 - __clz(int(v))));
  return 31 - __clz(int(v));
}
RT_DEV uint32_t dropTopBits(uint32_t v, int n) { for(int k = 0; k < n; k++) v &= ~(1u << (31 - __clz(int(v)))); return v; }

// Runs the live rays of this wave to completion.  Every lane of the wave calls it (uniform); on entry `live` marks the owner groups, T their state.
// Stack space: a ray's column holds S.stackEntries = S.stackTotal + GANG_EXTRA entries in the latency build.  Several children are expanded per round only while
// the column has more than S.stackTotal entries to spare; beyond that one child per round is taken — plain depth-first order, which from any state needs at most
// the tree's depth (< S.stackTotal) more entries — so the column cannot overflow.
RT_DEV void gangTail(const DevScene& S, float4* pool, uint2* waveStack, Trav& T, bool& live, int mySlot, TravCounters& tc)
{
  const int lane = int(threadIdx.x) & 63, j = lane & (WIDE_G - 1), g = lane >> 3;
  float4* gbox = reinterpret_cast<float4*>(waveStack + size_t(S.stackEntries) * WIDE_RAYS);
  const int wideRoom = S.stackEntries - S.stackTotal;   // parallel expansion while sp + taken <= wideRoom
  bool working = false;          // this group works on a ray (its own or an adopted one)
  int og = g;                    // owner group of that ray = its stack column
  uint32_t wm = 0u;              // groups that work on the same ray
  uint32_t lastOwners = 0u;
  for(;;) {
    const uint32_t owners = leaderBits(__ballot(live ? 1 : 0));
    if(owners == 0u) break;
    if(owners != lastOwners) {
      // ---- (re)assignment: owners keep their ray; the groups without work are dealt round-robin to the live rays and adopt the ray's state; groups that are
      //      still busy with a ray (its owner lives) keep everything they hold -------------------------------------------------------------------------------
      lastOwners = owners;
      const int n = __popc(owners);
      const bool mine = (owners >> g) & 1u;
      const uint32_t freeM = leaderBits(__ballot((!working && !mine) ? 1 : 0));
      const bool adopt = !working && !mine;
      if(mine && !working) {   // first round of an owner: what is pending moves to the stack column (the workers share it); pending triangles stay with their worker
        if(T.ngroup.y > 0x00FFFFFFu) { if(__builtin_expect(T.sp >= S.stackEntries, 0)) __builtin_trap(); stackPushW(T, waveStack + g, T.ngroup, j); T.ngroup.y = 0u; }
        og = g;
      }
      if(adopt) {
        uint32_t o = owners;
        for(int k = __popc(freeM & ((1u << g) - 1u)) % n; k > 0; k--) o &= o - 1u;
        og = __ffs(int(o)) - 1;
      }
      waveLdsSync();
      const int src = og * WIDE_G;
      const int slot = __shfl(mySlot, src);
      const int sp = __shfl(T.sp, src);
      const float bt = __shfl(T.hit.t, src), bu = __shfl(T.hit.u, src), bv = __shfl(T.hit.v, src);
      const uint32_t bg = uint32_t(__shfl(int(T.hit.gid), src));
      const int any = __shfl(T.isAny ? 1 : 0, src);
      if(adopt) {
        const float4 a = pool[slot * POOL_SLOT_F4], b = pool[slot * POOL_SLOT_F4 + 1];
        T.isAny = any != 0;
        (void)travInit<2>(T, mk3(a.x, a.y, a.z), mk3(a.w, b.x, b.y), b.z, rt_f2u(b.w));
        T.ngroup.y = 0u; T.tgroup.y = 0u;
        T.sp = sp; T.hit.t = bt; T.hit.gid = bg; T.hit.u = bu; T.hit.v = bv;
      }
      working = true;
      wm = 0u;
#pragma unroll
      for(int k = 0; k < WIDE_RAYS; k++) if(__builtin_amdgcn_readlane(og, k * WIDE_G) == og) wm |= 1u << k;
    }
    uint2* stack = waveStack + og;
    // ---- triangle phase: every worker tests (up to eight of) the leaf triangles of ITS node ----------------------------------------------------------------
    const bool hasTri = working && T.tgroup.y != 0u;
    if(__ballot(hasTri ? 1 : 0) != 0ull) {
      if(hasTri) travTriW<2>(S, T, j, tc);
      if(j == 0) gbox[g] = make_float4(T.hit.t, rt_u2f(T.hit.gid), T.hit.u, T.hit.v);
      waveLdsSync();
      uint32_t rest = wm & ~(1u << g);
      while(rest != 0u) {   // merge: the workers of a ray end up with the same best hit
        const int k = __ffs(int(rest)) - 1; rest &= rest - 1u;
        const float4 c = gbox[k];
        const uint32_t cg = rt_f2u(c.y);
        if(cg != 0xffffffffu && (T.hit.gid == 0xffffffffu || (!T.isAny && (c.x < T.hit.t || (c.x == T.hit.t && cg < T.hit.gid))))) { T.hit.t = c.x; T.hit.gid = cg; T.hit.u = c.z; T.hit.v = c.w; }
      }
      if(T.isAny && T.hit.gid != 0xffffffffu) { T.sp = 0; T.tgroup.y = 0u; T.ngroup.y = 0u; }   // first accepted hit terminates the query, for every worker
      waveLdsSync();
    }
    // ---- node phase: the workers without pending triangles take the nearest pending children of their ray, one each ---------------------------------------
    const bool wantNode = working && T.tgroup.y == 0u && T.sp > 0;
    const uint32_t wantM = leaderBits(__ballot(wantNode ? 1 : 0)) & wm;
    bool took = false; uint32_t child = 0u;
    if(wantNode) {
      // The ray's pending children, nearest first: the hit bits of its top GANG_SCAN stack entries, top entry first, high bit first.  Worker i takes the i-th.
      // Entries above the deepest one touched are used up and dropped; that one is written back with its remaining bits (or dropped as well).
      const int iw = __popc(wantM & ((1u << g) - 1u)), nw = __popc(wantM);
      uint2 e[GANG_SCAN]; int cnt[GANG_SCAN]; int avail = 0;
#pragma unroll
      for(int k = 0; k < GANG_SCAN; k++) {
        e[k] = make_uint2(0u, 0u);
        if(T.sp > k) e[k] = stack[(T.sp - 1 - k) * WIDE_RAYS];
        cnt[k] = __popc(e[k].y >> 24); avail += cnt[k];
      }
      int nsel = min(nw, avail);
      if(T.sp + nsel > wideRoom) nsel = 1;
      int before = 0, nsp = T.sp; bool wrote = false; uint2 back = make_uint2(0u, 0u); int backAt = 0;
#pragma unroll
      for(int k = 0; k < GANG_SCAN; k++) {
        const int takeHere = max(0, min(cnt[k], nsel - before));   // children of entry k that are taken this round
        if(iw >= before && iw < before + takeHere) {
          took = true;
          const int bit = nthSetBitFromTop(e[k].y & 0xFF000000u, iw - before);
          const uint32_t slotI = uint32_t(bit - 24) ^ T.octinv;
          child = e[k].x + uint32_t(__popc(e[k].y & ~(0xFFFFFFFFu << slotI) & 0xFFu));
        }
        if(takeHere > 0) {
          const uint32_t h = dropTopBits(e[k].y & 0xFF000000u, takeHere) | (e[k].y & 0x00FFFFFFu);
          if(h > 0x00FFFFFFu) { nsp = T.sp - k; wrote = true; back = make_uint2(e[k].x, h); backAt = T.sp - 1 - k; }   // partially used: stays (it is the deepest one touched)
          else nsp = T.sp - 1 - k;
        }
        before += takeHere;
      }
      if(wrote && iw == 0 && j == 0) stack[backAt * WIDE_RAYS] = back;
      T.sp = nsp;
    }
    waveLdsSync();
    // (workers of the ray that sat this phase out — pending triangles — follow the stack pointer)
    { const int sp2 = __shfl(T.sp, wantM != 0u ? (__ffs(int(wantM)) - 1) * WIDE_G : lane); if(working && !wantNode && wantM != 0u) T.sp = sp2; }   // (every lane takes part in the shuffle)
    if(took) { const NodeRegs N = nodeLoad(S, child); travNodeTestW(T, N, j); }
    const bool push = took && T.ngroup.y > 0x00FFFFFFu;
    const uint32_t pushM = leaderBits(__ballot(push ? 1 : 0)) & wm;
    if(working) {
      // (the bound above is an argument, not a check: a column that would overflow into the gang mailbox / the next wave's stacks aborts the launch — the
      //  library then reports RT_ERR_HIP — instead of corrupting LDS silently; advisor finding of round 4)
      if(__builtin_expect(T.sp + __popc(pushM) > S.stackEntries, 0)) __builtin_trap();
      if(push && j == 0) stack[(T.sp + __popc(pushM & ~((2u << g) - 1u))) * WIDE_RAYS] = T.ngroup;   // farthest first: the nearest group ends on top
      T.sp += __popc(pushM);
      T.ngroup.y = 0u;
    }
    waveLdsSync();
    // ---- finished rays: the owner publishes the result, its workers are free for the next assignment -------------------------------------------------------
    const uint32_t busyM = leaderBits(__ballot((working && (T.tgroup.y != 0u || T.sp > 0)) ? 1 : 0)) & wm;
    if(working && busyM == 0u) {
      if(live) { if(j == 0) pool[mySlot * POOL_SLOT_F4] = make_float4(T.hit.t, rt_u2f(T.hit.gid), T.hit.u, T.hit.v); live = false; }
      working = false;
    }
  }
}

// ---- workgroup-wide ray pool ------------------------------------------------------------------------------------------------------------------------
// Rays wait in LDS slots (poolPut: ray in, hit out, 32 B), `list` holds the n occupied slot ids (kind of the ray in bit 7: any-hit), *next is the shared cursor.
// EVERY wave of the workgroup calls this between two workgroup barriers; a wave serves up to eight rays at a time and refills group by group.
RT_DEV void tracePoolWide(const DevScene& S, float4* pool, const unsigned char* list, int n, uint32_t* next, uint2* waveStack, TravCounters& tc)
{
  const int lane = int(threadIdx.x) & 63, j = lane & (WIDE_G - 1), r = lane >> 3;
  uint2* stack = waveStack + r;
  const unsigned long long leaders = 0x0101010101010101ull;
  const unsigned long long below = (1ull << (lane & ~(WIDE_G - 1))) - 1ull;   // lanes of lower groups
  bool live = false, exhausted = n == 0;
  int mySlot = 0;
  Trav T;
  T.isAny = false;
  for(;;) {
    unsigned long long liveMask = __ballot(live ? 1 : 0);
    const unsigned long long idle = ~liveMask & leaders;
    if(!exhausted && idle != 0ull) {
      const int nIdle = __popcll(idle);
      uint32_t base = 0u;
      if(lane == 0) base = atomicAdd(next, uint32_t(nIdle));
      base = uint32_t(__builtin_amdgcn_readfirstlane(int(base)));
      const int item = int(base) + __popcll(idle & below);
      if(!live && item < n) {
        const uint32_t e = list[item];
        mySlot = int(e & 0x7fu);
        const float4 a = pool[mySlot * POOL_SLOT_F4], b = pool[mySlot * POOL_SLOT_F4 + 1];
        T.isAny = (e & 0x80u) != 0u;
        live = travInit<2>(T, mk3(a.x, a.y, a.z), mk3(a.w, b.x, b.y), b.z, rt_f2u(b.w));
        if(!live && j == 0) pool[mySlot * POOL_SLOT_F4] = make_float4(T.hit.t, rt_u2f(T.hit.gid), T.hit.u, T.hit.v);
      }
      if(int(base) + nIdle >= n) exhausted = true;
      liveMask = __ballot(live ? 1 : 0);
    }
    if(liveMask == 0ull) {
      if(exhausted) break;
      continue;
    }
    if(exhausted && S.gangMax > 0 && __popcll(liveMask & leaders) <= S.gangMax && __popcll(liveMask & leaders) < WIDE_RAYS) {   // the tail: idle slots become workers of the live rays
      gangTail(S, pool, waveStack, T, live, mySlot, tc);
      break;
    }
#if RT_WAVEPROF
    const uint64_t pc0 = clock64();
    const bool anyTri = __ballot((live && travHasTris(T)) ? 1 : 0) != 0ull;
    tc.rC += uint32_t(__popcll(liveMask & leaders));   // rays served this round (of WIDE_RAYS)
#endif
    const bool still = travRoundW<2>(S, T, live, j, stack, tc);
#if RT_WAVEPROF
    { const uint32_t dc = uint32_t(clock64() - pc0); if(anyTri) { tc.rT++; tc.cT += dc; } else { tc.rN++; tc.cN += dc; } }
#endif
    if(live && !still && j == 0) pool[mySlot * POOL_SLOT_F4] = make_float4(T.hit.t, rt_u2f(T.hit.gid), T.hit.u, T.hit.v);
    live = still;
  }
}
#endif  // RT_LAT

}  // namespace rt
