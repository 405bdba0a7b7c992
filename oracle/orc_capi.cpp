// orc_capi.cpp — ctypes-facing C entry points of the CPU oracle (liboracle.so).
// TEST INFRASTRUCTURE: loaded only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
#include "orc_stages.h"
#include "../include/rt_cpus.h"
#include <cstdio>
#include <thread>

using namespace orc;

namespace {
struct Ctx {
  Scene scene;
  Frame frame;
  bool haveScene = false;
};
template <class T> size_t vbytes(const std::vector<T>& v) { return v.size() * sizeof(T); }
bool isHalfRes(int id) { return id == RT_BUF_INDIRECT_RESV0 || id == RT_BUF_INDIRECT_RESV1 || id == RT_BUF_INDIRECT_RESV_TEMP; }
size_t elemSize(int id)
{
  switch(id) {
    case RT_BUF_MOTION: case RT_BUF_LIGHT_ID0: case RT_BUF_LIGHT_ID1: case RT_BUF_LDR: return 4;
    case RT_BUF_DIRECT_RESV0: case RT_BUF_DIRECT_RESV1: case RT_BUF_DIRECT_RESV_TEMP: return sizeof(rt_direct_reservoir);
    case RT_BUF_INDIRECT_RESV0: case RT_BUF_INDIRECT_RESV1: case RT_BUF_INDIRECT_RESV_TEMP: return sizeof(rt_indirect_reservoir);
    default: return 16;
  }
}

void* bufPtr(Ctx* c, int id, size_t& bytes)
{
  Frame& f = c->frame;
  switch(id) {
    case RT_BUF_GBUFFER0: bytes = vbytes(f.gbuffer[0]); return f.gbuffer[0].data();
    case RT_BUF_GBUFFER1: bytes = vbytes(f.gbuffer[1]); return f.gbuffer[1].data();
    case RT_BUF_MOTION: bytes = vbytes(f.motion); return f.motion.data();
    case RT_BUF_DIRECT_RESV0: bytes = vbytes(f.directResv[0]); return f.directResv[0].data();
    case RT_BUF_DIRECT_RESV1: bytes = vbytes(f.directResv[1]); return f.directResv[1].data();
    case RT_BUF_DIRECT_RESV_TEMP: bytes = vbytes(f.directResvTemp); return f.directResvTemp.data();
    case RT_BUF_INDIRECT_RESV0: bytes = vbytes(f.indirectResv[0]); return f.indirectResv[0].data();
    case RT_BUF_INDIRECT_RESV1: bytes = vbytes(f.indirectResv[1]); return f.indirectResv[1].data();
    case RT_BUF_INDIRECT_RESV_TEMP: bytes = vbytes(f.indirectResvTemp); return f.indirectResvTemp.data();
    case RT_BUF_DENOISE_DIR_A: bytes = vbytes(f.denoiseTemp[0]); return f.denoiseTemp[0].data();
    case RT_BUF_DENOISE_DIR_B: bytes = vbytes(f.denoiseTemp[1]); return f.denoiseTemp[1].data();
    case RT_BUF_DENOISE_IND_A: bytes = vbytes(f.denoiseTemp[2]); return f.denoiseTemp[2].data();
    case RT_BUF_DENOISE_IND_B: bytes = vbytes(f.denoiseTemp[3]); return f.denoiseTemp[3].data();
    case RT_BUF_DIRECT_RESULT0: bytes = vbytes(f.directResult[0]); return f.directResult[0].data();
    case RT_BUF_DIRECT_RESULT1: bytes = vbytes(f.directResult[1]); return f.directResult[1].data();
    case RT_BUF_INDIRECT_RESULT0: bytes = vbytes(f.indirectResult[0]); return f.indirectResult[0].data();
    case RT_BUF_INDIRECT_RESULT1: bytes = vbytes(f.indirectResult[1]); return f.indirectResult[1].data();
    case RT_BUF_LIGHT_ID0: bytes = vbytes(f.lightId2[0]); return f.lightId2[0].data();
    case RT_BUF_LIGHT_ID1: bytes = vbytes(f.lightId2[1]); return f.lightId2[1].data();
    case RT_BUF_LDR: bytes = vbytes(f.ldr); return f.ldr.data();
  }
  bytes = 0;
  return nullptr;
}
}  // namespace

extern "C" {

void* orc_create(int threads)
{
  Ctx* c = new Ctx();
  if(threads <= 0) threads = rt_cpu_budget();
  c->frame.threads = threads > 0 ? threads : 1;
  c->frame.scene = &c->scene;
  return c;
}
void orc_destroy(void* p) { delete static_cast<Ctx*>(p); }
int orc_threads(void* p) { return static_cast<Ctx*>(p)->frame.threads; }
// bench.py's cpu_baseline leg: another thread count for the same context (scaling points), workers pinned to the CPUs of the process's affinity mask
void orc_set_threads(void* p, int threads, int pin)
{
  Ctx* c = static_cast<Ctx*>(p);
  if(threads <= 0) threads = rt_cpu_budget();
  c->frame.threads = threads > 0 ? threads : 1; c->frame.pin = pin != 0;
}
int orc_upload_scene(void* p, const rt_scene_desc* d)
{
  Ctx* c = static_cast<Ctx*>(p);
  if(!d) return RT_ERR_INVALID_ARG;
  c->scene.upload(d);
  c->scene.build();
  c->haveScene = true;
  return RT_OK;
}
int orc_resize(void* p, int w, int h)
{
  if(w <= 0 || h <= 0) return RT_ERR_INVALID_ARG;
  static_cast<Ctx*>(p)->frame.resize(w, h);
  return RT_OK;
}
int orc_set_camera(void* p, const rt_scene_camera* cam) { static_cast<Ctx*>(p)->frame.cam = *cam; return RT_OK; }
int orc_render_frame(void* p, const rt_state* st, int frames)
{
  Ctx* c = static_cast<Ctx*>(p);
  if(!c->haveScene) return RT_ERR_NO_SCENE;
  if(st->size.x != c->frame.W || st->size.y != c->frame.H) return RT_ERR_NO_TARGET;
  c->frame.renderFrame(*st, frames);
  c->scene.counters.flush();
  return RT_OK;
}
int orc_run_stage(void* p, const rt_state* st, int frames, int stage, int level, int rowBegin, int rowEnd)
{
  Ctx* c = static_cast<Ctx*>(p);
  if(!c->haveScene) return RT_ERR_NO_SCENE;
  if(st->size.x != c->frame.W || st->size.y != c->frame.H) return RT_ERR_NO_TARGET;
  c->frame.runStage(*st, frames, stage, level, rowBegin, rowEnd);
  c->scene.counters.flush();
  return RT_OK;
}
// logical size (W x H elements); the vectors carry slack rows behind it
size_t orc_buffer_bytes(void* p, int id)
{
  Ctx* c = static_cast<Ctx*>(p);
  if(id < 0 || id >= RT_BUF_COUNT) return 0;
  return (isHalfRes(id) ? size_t(c->frame.W / 2) * (c->frame.H / 2) : size_t(c->frame.W) * c->frame.H) * elemSize(id);
}
int orc_buffer_ptr(void* p, int id, void** ptr, size_t* allocBytes, size_t* rowPitch)
{
  Ctx* c = static_cast<Ctx*>(p);
  size_t b; void* q = bufPtr(c, id, b);
  if(!q) return RT_ERR_INVALID_ARG;
  *ptr = q; *allocBytes = b; *rowPitch = size_t(isHalfRes(id) ? c->frame.W / 2 : c->frame.W) * elemSize(id);
  return RT_OK;
}
int orc_readback(void* p, int id, void* dst, size_t bytes)
{
  size_t b; void* src = bufPtr(static_cast<Ctx*>(p), id, b);
  b = orc_buffer_bytes(p, id);
  if(!src || bytes != b) return RT_ERR_INVALID_ARG;
  memcpy(dst, src, b);
  return RT_OK;
}
int orc_upload_history(void* p, int id, const void* src, size_t bytes)
{
  size_t b; void* dst = bufPtr(static_cast<Ctx*>(p), id, b);
  b = orc_buffer_bytes(p, id);
  if(!dst || bytes != b) return RT_ERR_INVALID_ARG;
  memcpy(dst, src, b);
  return RT_OK;
}
int orc_get_counters(void* p, rt_counters* out)
{
  Ctx* c = static_cast<Ctx*>(p);
  memset(out, 0, sizeof(*out));
  c->scene.counters.flush();   // (what the calling thread itself counted: single-threaded stages, orc_trace_*)
  out->closestHitRays = c->scene.counters.closestHitRays; out->anyHitRays = c->scene.counters.anyHitRays;
  out->nodesVisited = c->scene.counters.nodesVisited; out->trisTested = c->scene.counters.trisTested;
  out->hitsShaded = c->scene.counters.hitsShaded; out->risCandidates = c->scene.counters.risCandidates;
  return RT_OK;
}
int orc_set_sun_and_sky(void* p, const rt_sun_and_sky* ss) { if(!ss) return RT_ERR_INVALID_ARG; static_cast<Ctx*>(p)->scene.sunAndSky = *ss; return RT_OK; }
void orc_sun_and_sky_eval(const rt_sun_and_sky* ss, int n, const float* dirs, float* out)
{
  for(int i = 0; i < n; i++) { vec3 r = sky::sun_and_sky(*ss, V3(dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2])); out[3 * i] = r.x; out[3 * i + 1] = r.y; out[3 * i + 2] = r.z; }
}
int orc_pick(void* p, const rt_mat4* viewInv, const rt_mat4* projInv, float pickX, float pickY, rt_pick_result* out)
{
  Ctx* c = static_cast<Ctx*>(p);
  if(!c->haveScene) return RT_ERR_NO_SCENE;
  const vec2 d = V2(pickX, pickY) * 2.0f - 1.0f;
  const vec4 origin = mul(*reinterpret_cast<const mat4*>(viewInv), V4(0, 0, 0, 1));
  const vec4 target = mul(*reinterpret_cast<const mat4*>(projInv), V4(d.x, d.y, 1, 1));
  const vec4 direction = mul(*reinterpret_cast<const mat4*>(viewInv), V4(normalize(xyz(target)), 0));
  const vec3 o = xyz(origin), dir = normalize(xyz(direction));
  const Hit h = c->scene.closestHit(o, dir, 0u);
  c->scene.counters.flush();   // (every entry point that traces on the CALLER's thread hands its counts to its own context before it returns: the thread-local block is
                               //  shared by all contexts of the thread — advisor finding of round 5: context A's rays ended in whichever context flushed next)
  out->worldRayOrigin = rt_vec4{o.x, o.y, o.z, 1.0f}; out->worldRayDirection = rt_vec4{dir.x, dir.y, dir.z, 0.0f};
  out->hitT = h.t; out->primitiveID = 0; out->instanceID = -1; out->instanceCustomIndex = 0; out->baryCoord = rt_vec3{0, 0, 0};
  if(h.tri != 0xffffffffu) {
    const Tri& t = c->scene.tris[h.tri];
    out->primitiveID = int32_t(t.prim); out->instanceID = int32_t(t.inst); out->instanceCustomIndex = int32_t(c->scene.instances[t.inst].primMesh);
    out->baryCoord = rt_vec3{(1.0f - h.u) - h.v, h.u, h.v};
  }
  return RT_OK;
}
int orc_tonemap(void* p, const rt_tonemapper* tm, int dbg, int frames)
{
  Ctx* c = static_cast<Ctx*>(p);
  if(!tm || c->frame.W == 0) return RT_ERR_INVALID_ARG;
  c->frame.tonemap(*tm, dbg, frames);
  c->scene.counters.flush();
  return RT_OK;
}
int orc_set_history_rows(void* p, int r0, int r1) { Ctx* c = static_cast<Ctx*>(p); c->frame.histRow0 = r0; c->frame.histRow1 = r1; return RT_OK; }
int orc_history_miss(void* p) { Ctx* c = static_cast<Ctx*>(p); return int(c->frame.histMiss.exchange(0u) | c->frame.histMissInd.exchange(0u)); }
int orc_history_miss_stage(void* p, int stage) { Ctx* c = static_cast<Ctx*>(p); return int(stage == RT_STAGE_INDIRECT ? c->frame.histMissInd.exchange(0u) : c->frame.histMiss.exchange(0u)); }
void orc_reset_counters(void* p) { static_cast<Ctx*>(p)->scene.counters.reset(); }
uint64_t orc_num_triangles(void* p) { return static_cast<Ctx*>(p)->scene.tris.size(); }

// ---- ray-level access for traversal parity tests ----------------------------------------------------------
// rays: n x {ox,oy,oz,dx,dy,dz,tmax,seedbits}; out: n x {t, tri(u32 bits), u, v}
void orc_trace_closest(void* p, int n, const float* rays, float* out)
{
  Ctx* c = static_cast<Ctx*>(p);
  for(int i = 0; i < n; i++) {
    const float* r = rays + 8 * i;
    Hit h = c->scene.closestHit(V3(r[0], r[1], r[2]), V3(r[3], r[4], r[5]), rt_f2u(r[7]));
    out[4 * i + 0] = h.t; out[4 * i + 1] = rt_u2f(h.tri); out[4 * i + 2] = h.u; out[4 * i + 3] = h.v;
  }
  c->scene.counters.flush();
}
void orc_trace_any(void* p, int n, const float* rays, int32_t* out)
{
  Ctx* c = static_cast<Ctx*>(p);
  for(int i = 0; i < n; i++) {
    const float* r = rays + 8 * i;
    out[i] = c->scene.anyHit(V3(r[0], r[1], r[2]), V3(r[3], r[4], r[5]), r[6], rt_f2u(r[7])) ? 1 : 0;
  }
  c->scene.counters.flush();
}
// brute force over all triangles (no BVH): validates the oracle's own BVH
void orc_trace_closest_brute(void* p, int n, const float* rays, float* out)
{
  Ctx* c = static_cast<Ctx*>(p);
  const Scene& S = c->scene;
  for(int i = 0; i < n; i++) {
    const float* r = rays + 8 * i;
    vec3 o = V3(r[0], r[1], r[2]), d = V3(r[3], r[4], r[5]);
    Hit best;
    for(uint32_t ti = 0; ti < S.tris.size(); ti++) {
      float t, u, v;
      if(!S.intersectTri(S.tris[ti], o, d, t, u, v)) continue;
      if(!(t > 0.0f && t < RT_INFINITY)) continue;
      if(!(t < best.t || (t == best.t && ti < best.tri))) continue;
      if(!(S.tris[ti].flags & TRI_OPAQUE) && !S.hitTest(S.tris[ti], ti, u, v, rt_f2u(r[7]))) continue;
      best.t = t; best.tri = ti; best.u = u; best.v = v;
    }
    out[4 * i + 0] = best.t; out[4 * i + 1] = rt_u2f(best.tri); out[4 * i + 2] = best.u; out[4 * i + 3] = best.v;
  }
  c->scene.counters.flush();
}

// the sampler on its own (tests/test_trace_pin.py: against a float64 statement of the Vulkan rules): n uv pairs -> n RGBA values
void orc_sample_texture(const uint8_t* bgra, int w, int h, int wrapS, int wrapT, int filter, int n, const float* uv, float* out)
{
  Scene S;
  Texture t;
  t.bgra.assign(bgra, bgra + size_t(w) * h * 4); t.w = w; t.h = h; t.wrapS = wrapS; t.wrapT = wrapT; t.filter = filter;
  S.textures.push_back(t);
  for(int i = 0; i < n; i++) {
    const vec4 c = S.sampleTexture(0, V2(uv[2 * i], uv[2 * i + 1]));
    out[4 * i] = c.x; out[4 * i + 1] = c.y; out[4 * i + 2] = c.z; out[4 * i + 3] = c.w;
  }
}

// ---- known-answer entry points (random.glsl, compress.glsl, common.glsl) ------------------------------------
uint32_t orc_tea(uint32_t a, uint32_t b) { return tea(a, b); }
uint32_t orc_pcg(uint32_t* s) { return pcg(*s); }
float orc_rand(uint32_t* s) { return rnd(*s); }
uint32_t orc_compress_unit_vec(float x, float y, float z) { return compress_unit_vec(V3(x, y, z)); }
void orc_decompress_unit_vec(uint32_t p, float* out) { vec3 v = decompress_unit_vec(p); out[0] = v.x; out[1] = v.y; out[2] = v.z; }
uint32_t orc_pack_unorm4x8(float x, float y, float z, float w) { return packUnorm4x8(V4(x, y, z, w)); }
uint32_t orc_hash8bit(uint32_t a) { return hash8bit(a); }
void orc_offset_ray(const float* p, const float* n, float* out) { vec3 r = OffsetRay(V3(p[0], p[1], p[2]), V3(n[0], n[1], n[2])); out[0] = r.x; out[1] = r.y; out[2] = r.z; }
void orc_concentric_disk(float a, float b, float* out) { vec2 r = toConcentricDisk(V2(a, b)); out[0] = r.x; out[1] = r.y; }

// ---- BSDF access for property tests ------------------------------------------------------------------------
// m = {albedo.rgb, metallic, roughness}
static State mkState(const float* m) { State s; s.mat.albedo = V3(m[0], m[1], m[2]); s.mat.metallic = m[3]; s.mat.roughness = m[4]; return s; }
void orc_bsdf_eval(const float* m, const float* n, const float* wo, const float* wi, float* out)
{
  State s = mkState(m); float pdf = 0;
  vec3 f = metallicWorkflowEval(s, V3(n[0], n[1], n[2]), V3(wo[0], wo[1], wo[2]), V3(wi[0], wi[1], wi[2]), pdf);
  out[0] = f.x; out[1] = f.y; out[2] = f.z; out[3] = pdf;
}
void orc_bsdf_sample(const float* m, const float* n, const float* wo, const float* r, float* out)
{
  State s = mkState(m); vec3 bsdf = V3(0.0f), dir = V3(0.0f);
  float pdf = metallicWorkflowSample(s, V3(n[0], n[1], n[2]), V3(wo[0], wo[1], wo[2]), V3(r[0], r[1], r[2]), bsdf, dir);
  out[0] = dir.x; out[1] = dir.y; out[2] = dir.z; out[3] = pdf; out[4] = bsdf.x; out[5] = bsdf.y; out[6] = bsdf.z;
}

// ---- scalar helpers + reservoir arithmetic for the known-answer vectors minted from the reference's GLSL (tests/test_kat_float.py)
void orc_spherical_uv(const float* d, float* out) { vec2 uv = GetSphericalUv(V3(d[0], d[1], d[2])); out[0] = uv.x; out[1] = uv.y; }
void orc_coordinate_system(const float* n, float* out) { vec3 t, b; CreateCoordinateSystem(V3(n[0], n[1], n[2]), t, b); out[0] = t.x; out[1] = t.y; out[2] = t.z; out[3] = b.x; out[4] = b.y; out[5] = b.z; }
float orc_power_heuristic(float f, float g) { return powerHeuristic(f, g); }
float orc_luminance(const float* c) { return luminance(V3(c[0], c[1], c[2])); }
void orc_hdr_to_ldr(const float* c, float* out) { vec3 r = HDRToLDR(V3(c[0], c[1], c[2])); out[0] = r.x; out[1] = r.y; out[2] = r.z; }
void orc_ldr_to_hdr(const float* c, float* out) { vec3 r = LDRToHDR(V3(c[0], c[1], c[2])); out[0] = r.x; out[1] = r.y; out[2] = r.z; }
// One step of the op stream of oracle/kat/kat_float.cpp over a (DirectReservoir, IndirectReservoir) pair.
// ops: 0 update(w, r, tag)  1 merge(rhs{num, weight, tag}, r) [direct only]  2 clamp(c)  3 checkValidity  4 reset
void orc_resv_op(rt_direct_reservoir* d, rt_indirect_reservoir* g, int op, float w, float r, float tag, uint32_t rn, int c, int* invalid)
{
  switch(op) {
    case 0: {
      rt_light_sample ls = zeroLightSample(); ls.Li = rt_vec3{tag, tag, tag}; ls.wi = rt_vec3{0, 0, 1}; ls.dist = tag; resvUpdate(*d, ls, w, r);
      rt_gi_sample gs; memset(&gs, 0, sizeof(gs)); gs.L = rt_vec3{tag, tag, tag}; gs.pHat = tag; resvUpdate(*g, gs, w, r);
    } break;
    case 1: {
      rt_direct_reservoir rhs; memset(&rhs, 0, sizeof(rhs)); rhs.lightSample.Li = rt_vec3{tag, tag, tag}; rhs.lightSample.wi = rt_vec3{0, 0, 1}; rhs.lightSample.dist = tag;
      rhs.num = rn; rhs.weight = w;
      if(!resvInvalid(rhs)) resvMerge(*d, rhs, r);
    } break;
    case 2: resvClamp(*d, c); resvClamp(*g, c); break;
    case 3: resvCheckValidity(*d); resvCheckValidity(*g); break;
    default: resvReset(*d); resvReset(*g); break;
  }
  invalid[0] = resvInvalid(*d) ? 1 : 0; invalid[1] = resvInvalid(*g) ? 1 : 0;
}

// ---- numerics contract (include/rt_detmath.h) on arrays -----------------------------------------------------
void orc_detmath(int op, int n, const float* a, const float* b, float* out)
{
  for(int i = 0; i < n; i++) {
    switch(op) {
      case 0: out[i] = rt_exp(a[i]); break;
      case 1: out[i] = rt_log(a[i]); break;
      case 2: out[i] = rt_pow(a[i], b[i]); break;
      case 3: out[i] = rt_sin(a[i]); break;
      case 4: out[i] = rt_cos(a[i]); break;
      case 5: out[i] = rt_asin(a[i]); break;
      case 6: out[i] = rt_acos(a[i]); break;
      case 7: out[i] = rt_atan2(a[i], b[i]); break;
      case 8: out[i] = rt_tan(a[i]); break;
      // 9: the product's division shortcut for a uniform divisor (csrc/stages.hip divUniform): q' = fma(a - b*q, y, q), y = RN(1/b).
      //    Restated here only so that tests/test_detmath.py can check it against IEEE division on the CPU.
      case 9: { const float y = 1.0f / b[i], q = a[i] * y, r = fmaf(-b[i], q, a[i]); out[i] = (a[i] <= 1.0e30f) ? fmaf(r, y, q) : a[i] / b[i]; break; }
      // 10: the product's branch-free exp for x <= 0 (csrc/stages.hip expNonPositive) must equal rt_exp there
      case 10: {
        const float x = a[i], z = rt_floor(rt_fma(x, 1.44269504088896341f, 0.5f));
        const int n = int(fmaxf(z, -127.0f));
        float r = rt_fma(z, -0.693359375f, x); r = rt_fma(z, 2.12194440e-4f, r);
        const float rr = r * r;
        float p = 1.9875691500E-4f; p = rt_fma(p, r, 1.3981999507E-3f); p = rt_fma(p, r, 8.3334519073E-3f); p = rt_fma(p, r, 4.1665795894E-2f);
        p = rt_fma(p, r, 1.6666665459E-1f); p = rt_fma(p, r, 5.0000001201E-1f); p = rt_fma(p, rr, r); p = p + 1.0f;
        float e = p * rt_u2f(uint32_t(n + 127) << 23);
        out[i] = (x < -87.33654475055310f) ? 0.0f : e; break; }
      // 11 / 12: rt_ftoi / rt_ftou as floats (every value they return is exactly representable)
      case 11: out[i] = float(rt_ftoi(a[i])); break;
      case 12: out[i] = float(rt_ftou(a[i])); break;
      // 13: the product's b / 255 for 8-bit texel and G-buffer channels (csrc/dev_math.h unorm8ToFloat), restated for the CPU test
      case 13: { const float y = 1.0f / 255.0f, q = a[i] * y; out[i] = fmaf(fmaf(-255.0f, q, a[i]), y, q); break; }
      default: out[i] = 0;
    }
  }
}

}  // extern "C"
