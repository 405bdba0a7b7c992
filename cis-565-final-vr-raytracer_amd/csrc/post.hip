// post.hip — the display pass: shaders/post.frag:103-175 + tonemapping.glsl:25-105 as compute kernels writing RGBA8.
//   k_post_rowsum / k_post_mean  image mean for auto-exposure (stands in for textureLod(.., 20) on the mip pyramid that
//                                RenderOutput::genMipmap builds, render_output.cpp:243-254)
//   k_post_mip                   one level of the mip pyramid RenderOutput::genMipmap blits (linear filter), levels 1..7 of both result images:
//                                toneLocalExposure (post.frag:70-101, autoExposure bit 1) samples them
//   k_tonemap                    one thread per output pixel
// Arithmetic follows include/rt_detmath.h (rt_pow for every pow) and the left-to-right evaluation rule, like every other stage.
#include "stage_common.h"
#include <algorithm>

namespace rt {

// Mean in a fixed association order (double precision): lane L of a row accumulates x = L, L+64, ... ; lane 0 adds the 64
// partial sums in lane order; one thread adds the rows top to bottom.
__global__ __launch_bounds__(64) void k_post_rowsum(const float4* direct, const float4* indirect, int W, int H, double* rowSums)
{
  __shared__ double part[64][3];
  const int y = int(blockIdx.x) % H, img = int(blockIdx.x) / H, lane = int(threadIdx.x);
  const float4* src = (img ? indirect : direct) + size_t(y) * W;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0;
  for(int x = lane; x < W; x += 64) { const float4 v = src[x]; s0 += double(v.x); s1 += double(v.y); s2 += double(v.z); }
  part[lane][0] = s0; part[lane][1] = s1; part[lane][2] = s2;
  __syncthreads();
  if(lane == 0) {
    double t0 = 0.0, t1 = 0.0, t2 = 0.0;
    for(int l = 0; l < 64; l++) { t0 += part[l][0]; t1 += part[l][1]; t2 += part[l][2]; }
    double* o = rowSums + (size_t(img) * H + y) * 3;
    o[0] = t0; o[1] = t1; o[2] = t2;
  }
}
__global__ void k_post_mean(const double* rowSums, int W, int H, float* mean /* [2][4] */)
{
  const int img = int(threadIdx.x) / 3, ch = int(threadIdx.x) % 3;
  if(img >= 2) return;
  double t = 0.0;
  for(int y = 0; y < H; y++) t += rowSums[(size_t(img) * H + y) * 3 + ch];
  mean[img * 4 + ch] = float(t / (double(W) * double(H)));
}

// vkCmdBlitImage with VK_FILTER_LINEAR from a (sw x sh) level to a (dw x dh) level, as nvvk::cmdGenerateMipmaps records it
// (render_output.cpp:243-254): the destination texel centre maps to the source coordinate (x + 0.5) * sw / dw, which is then sampled with
// a clamped bilinear filter.  For even sizes this is the 2 x 2 box average.  The arithmetic order is fixed.
struct MipLevels { const float4* img[8]; int w[8], h[8]; };   // level 0 = the result image itself
RT_DEV float4 bilinearClamp(const float4* src, int sw, int sh, float fx, float fy)
{
  const float x0f = rt_floor(fx), y0f = rt_floor(fy);
  const float ax = fx - x0f, ay = fy - y0f;
  const int x0 = rt_ftoi(x0f), y0 = rt_ftoi(y0f);
  const int xa = min(max(x0, 0), sw - 1), xb = min(max(x0 + 1, 0), sw - 1), ya = min(max(y0, 0), sh - 1), yb = min(max(y0 + 1, 0), sh - 1);
  const float4 a = src[size_t(ya) * sw + xa], b = src[size_t(ya) * sw + xb], c = src[size_t(yb) * sw + xa], d = src[size_t(yb) * sw + xb];
  auto mx = [](float p, float q, float t) { return p * (1.0f - t) + q * t; };
  return make_float4(mx(mx(a.x, b.x, ax), mx(c.x, d.x, ax), ay), mx(mx(a.y, b.y, ax), mx(c.y, d.y, ax), ay), mx(mx(a.z, b.z, ax), mx(c.z, d.z, ax), ay),
                     mx(mx(a.w, b.w, ax), mx(c.w, d.w, ax), ay));
}
__global__ __launch_bounds__(256) void k_post_mip(const float4* src, int sw, int sh, float4* dst, int dw, int dh)
{
  const int x = int(blockIdx.x * 64 + (threadIdx.x & 63)), y = int(blockIdx.y * 4 + (threadIdx.x >> 6));
  if(x >= dw || y >= dh) return;
  const float fx = (float(x) + 0.5f) * (float(sw) / float(dw)) - 0.5f, fy = (float(y) + 0.5f) * (float(sh) / float(dh)) - 0.5f;
  dst[size_t(y) * dw + x] = bilinearClamp(src, sw, sh, fx, fy);
}
// texture(img, uv, bias = level): the level's bilinear sample at the fragment's uv (the display is 1:1, so the implicit LOD is 0)
RT_DEV f3 sampleLevel(const MipLevels& M, int level, float u, float v)
{
  const float4 t = bilinearClamp(M.img[level], M.w[level], M.h[level], u * float(M.w[level]) - 0.5f, v * float(M.h[level]) - 0.5f);
  return mk3(t.x, t.y, t.z);
}

RT_DEV f3 pow3(f3 c, float e) { return mk3(rt_pow(c.x, e), rt_pow(c.y, e), rt_pow(c.z, e)); }
RT_DEV f3 linearTosRGB(f3 c) { return pow3(c, 1.0f / 2.2f); }  // tonemapping.glsl:25-33
RT_DEV f3 sRGBToLinear(f3 c) { return pow3(c, 2.2f); }         // :37-40
RT_DEV f3 uncharted2Impl(f3 color)                             // :50-59
{
  const float A = 0.15f, B = 0.50f, C = 0.10f, D = 0.20f, E = 0.02f, F = 0.30f;
  return ((color * (color * A + C * B) + D * E) / (color * (color * A + B) + D * F)) + (-(E / F));  // x - c == x + (-c) in IEEE arithmetic
}
RT_DEV f3 toneMapUncharted(f3 color)  // :61-67
{
  color = uncharted2Impl(color * 2.0f);
  const f3 whiteScale = mk3(1.0f) / uncharted2Impl(mk3(11.2f));
  return linearTosRGB(color * whiteScale);
}
RT_DEV float postLuminance(f3 c) { return (c.x * 0.2126f + c.y * 0.7152f) + c.z * 0.0722f; }  // post.frag:59-61
RT_DEV f3 toneExposure(const rt_tonemapper& tm, f3 RGB, float logAvgLum)                       // post.frag:63-68
{
  const float XYZy = (0.3575761f * RGB.x + 0.7151522f * RGB.y) + 0.1191920f * RGB.z;  // second row of the column-major RGB2XYZ (post.frag:58)
  const float Y = (tm.key / logAvgLum) * XYZy;
  const float Yd = (Y * (1.0f + Y / (tm.Ywhite * tm.Ywhite))) / (1.0f + Y);
  return RGB / XYZy * Yd;
}
// toneLocalExposure, post.frag:70-101.  v1 / v2 are the luminances of two consecutive pyramid levels at the fragment.  In the default view the
// reference's `v2 == luminance(...)` (a comparison, :91) leaves v2 undefined; undefined values are 0 here, so the
// adaptation luminance is the first v1 that clears the threshold, else 0.  The two single-image views assign v2 properly.
RT_DEV f3 toneLocalExposure(const rt_tonemapper& tm, int dbg, const MipLevels& MD, const MipLevels& MI, float u, float v, f3 RGB, float logAvgLum)
{
  const float XYZy = (0.3575761f * RGB.x + 0.7151522f * RGB.y) + 0.1191920f * RGB.z;
  const float Y = (tm.key / logAvgLum) * XYZy;
  const float factor = tm.key / logAvgLum;
  const float epsilon = 0.05f, phi = 2.0f;
  float La = 0.0f;
  float scale = 1.0f;
  for(int i = 0; i < 7; ++i) {
    float v1, v2 = 0.0f;
    if(dbg == RT_DBG_DIRECT_STAGE) { v1 = postLuminance(sampleLevel(MD, i, u, v)) * factor; v2 = postLuminance(sampleLevel(MD, i + 1, u, v)) * factor; }
    else if(dbg == RT_DBG_INDIRECT_STAGE) { v1 = postLuminance(sampleLevel(MI, i, u, v)) * factor; v2 = postLuminance(sampleLevel(MI, i + 1, u, v)) * factor; }
    else v1 = postLuminance(sampleLevel(MD, i, u, v) + sampleLevel(MI, i, u, v)) * factor;
    if(rt_abs(v1 - v2) / ((tm.key * rt_pow(2.0f, phi) / (scale * scale)) + v1) > epsilon) { La = v1; break; }
    La = v2;
    scale = scale * 2.0f;
  }
  const float Yd = Y / (1.0f + La);
  return RGB / XYZy * Yd;
}
RT_DEV f3 ditherColor(f3 linear, f3 noise, float quant)  // post.frag:50-55
{
  const f3 s = linearTosRGB(linear) / quant;
  const f3 c0 = mk3(rt_floor(s.x), rt_floor(s.y), rt_floor(s.z)) * quant;
  const f3 c1 = c0 + quant;
  const f3 a = sRGBToLinear(c0), b = sRGBToLinear(c1);
  const f3 discr = mk3(a.x * (1.0f - noise.x) + b.x * noise.x, a.y * (1.0f - noise.y) + b.y * noise.y, a.z * (1.0f - noise.z) + b.z * noise.z);
  return mk3(discr.x < linear.x ? c1.x : c0.x, discr.y < linear.y ? c1.y : c0.y, discr.z < linear.z ? c1.z : c0.z);
}
RT_DEV uint32_t toUnorm8(float c) { return rt_ftou(rt_floor(rt_clamp(c, 0.0f, 1.0f) * 255.0f + 0.5f)); }

__global__ __launch_bounds__(256) void k_tonemap(const float4* direct, const float4* indirect, const float* mean, rt_tonemapper tm, int dbg, int W, int H, uint32_t* ldr,
                                                 MipLevels MD, MipLevels MI)
{
  const int x = int(blockIdx.x * 64 + (threadIdx.x & 63)), y = int(blockIdx.y * 4 + (threadIdx.x >> 6));
  if(x >= W || y >= H) return;
  const float u = (float(x) + 0.5f) / float(W), v = (float(y) + 0.5f) / float(H);
  const int sx = min(max(rt_ftoi(u * tm.zoom * float(W)), 0), W - 1), sy = min(max(rt_ftoi(v * tm.zoom * float(H)), 0), H - 1);
  const float4 D = direct[size_t(sy) * W + sx], I = indirect[size_t(sy) * W + sx];
  f3 color;
  if(dbg == RT_DBG_DEPTH) {  // post.frag:106-112
    float depth = D.w;
    depth *= rt_pow(2.0f, tm.brightness);
    depth += tm.saturation;
    depth = rt_clamp(rt_pow(depth, 1.0f / tm.contrast), 0.0f, 1.0f);
    color = mk3(depth);
  } else if(dbg > RT_DBG_INDIRECT_STAGE) {  // :113-118
    color = mk3(D.x, D.y, D.z);
    if(dbg == RT_DBG_BASECOLOR) { const f3 p = pow3(color, 0.45454545454545f); color = mk3(rt_clamp(p.x, 0.f, 1.f), rt_clamp(p.y, 0.f, 1.f), rt_clamp(p.z, 0.f, 1.f)); }
  } else {
    f3 hdr = (dbg == RT_DBG_DIRECT_STAGE) ? mk3(D.x, D.y, D.z) : (dbg == RT_DBG_INDIRECT_STAGE) ? mk3(I.x, I.y, I.z) : mk3(D.x + I.x, D.y + I.y, D.z + I.z);
    if(tm.autoExposure & 1) {  // :133-153
      const f3 aD = mk3(mean[0], mean[1], mean[2]), aI = mk3(mean[4], mean[5], mean[6]);
      const f3 avg = (dbg == RT_DBG_DIRECT_STAGE) ? aD : (dbg == RT_DBG_INDIRECT_STAGE) ? aI : aD + aI;
      const float avgLum2 = postLuminance(avg);
      hdr = (tm.autoExposure & 2) ? toneLocalExposure(tm, dbg, MD, MI, u * tm.zoom, v * tm.zoom, hdr, avgLum2) : toneExposure(tm, hdr, avgLum2);
    }
    color = toneMapUncharted(hdr * tm.avgLum);  // toneMap(), tonemapping.glsl:89-105 with TONEMAP_UNCHARTED
    uint32_t rx = uint32_t(x) * 1664525u + 1013904223u, ry = uint32_t(y) * 1664525u + 1013904223u, rz = 1013904223u;  // pcg3d, random.glsl:81-92
    rx += ry * rz; ry += rz * rx; rz += rx * ry;
    rx ^= rx >> 16; ry ^= ry >> 16; rz ^= rz >> 16;
    rx += ry * rz; ry += rz * rx; rz += rx * ry;
    const f3 noise = mk3(rt_u2f(0x3f800000u | (rx >> 9)) - 1.0f, rt_u2f(0x3f800000u | (ry >> 9)) - 1.0f, rt_u2f(0x3f800000u | (rz >> 9)) - 1.0f);
    color = ditherColor(sRGBToLinear(color), noise, 1.0f / 255.0f);
    color = mk3(0.5f) * (1.0f - tm.contrast) + color * tm.contrast;  // mix(vec3(0.5), color, contrast)
    color = mk3(rt_clamp(color.x, 0.f, 1.f), rt_clamp(color.y, 0.f, 1.f), rt_clamp(color.z, 0.f, 1.f));
    color = pow3(color, 1.0f / tm.brightness);
    const float i = (color.x * 0.299f + color.y * 0.587f) + color.z * 0.114f;
    color = mk3(i) * (1.0f - tm.saturation) + color * tm.saturation;
    const float ux = ((u * tm.renderingRatio.x) - 0.5f) * 2.0f, uy = ((v * tm.renderingRatio.y) - 0.5f) * 2.0f;
    color *= 1.0f - (ux * ux + uy * uy) * tm.vignette;
  }
  ldr[size_t(y) * W + x] = toUnorm8(color.x) | (toUnorm8(color.y) << 8) | (toUnorm8(color.z) << 16) | 0xff000000u;
}

static_assert(sizeof(SkyPre) <= 512, "rt_set_sun_and_sky allocates 512 B for the precomputed sky terms");
__global__ void k_sky_prepare(rt_sun_and_sky ss, SkyPre* out) { if(threadIdx.x == 0 && blockIdx.x == 0) { SkyPre P; skyfn::prepare(ss, P); *out = P; } }

hipError_t launchSkyPrepare(hipStream_t stream, const rt_sun_and_sky& ss, SkyPre* out)
{
  hipLaunchKernelGGL(k_sky_prepare, dim3(1), dim3(64), 0, stream, ss, out);
  return hipGetLastError();
}

// nvvk::RayPickerKHR stand-in: one ray, lane 0 of one wave (the other lanes stay out of the traversal ballots)
__global__ __launch_bounds__(64) void k_pick(DevScene S, rt_mat4 viewInv, rt_mat4 projInv, float pickX, float pickY, rt_pick_result* out)
{
  extern __shared__ uint2 s_stack[];
  if(threadIdx.x != 0) return;
  const f2 d = mk2(pickX, pickY) * 2.0f - 1.0f;
  const f4 origin = mul(viewInv, mk4(0, 0, 0, 1));
  const f4 target = mul(projInv, mk4(d.x, d.y, 1, 1));
  const f4 direction = mul(viewInv, mk4(normalize(xyz(target)), 0));
  const f3 o = xyz(origin), dir = normalize(xyz(direction));
  RayHit hit; TravCounters tc{0, 0};
  traceRay<false>(S, o, dir, RT_INFINITY, 0u, s_stack, hit, tc);
  rt_pick_result r;
  r.worldRayOrigin = rt_vec4{o.x, o.y, o.z, 1.0f}; r.worldRayDirection = rt_vec4{dir.x, dir.y, dir.z, 0.0f};
  r.hitT = hit.t; r.primitiveID = 0; r.instanceID = -1; r.instanceCustomIndex = 0; r.baryCoord = rt_vec3{0, 0, 0};
  if(hit.gid != 0xffffffffu) {
    const TriRef tr = S.triRef[hit.gid];
    r.primitiveID = int32_t(tr.prim); r.instanceID = int32_t(tr.inst); r.instanceCustomIndex = int32_t(S.instances[tr.inst].primMesh);
    r.baryCoord = rt_vec3{(1.0f - hit.u) - hit.v, hit.u, hit.v};
  }
  *out = r;
}
// The ray query on its own (rt_trace_rays): one ray per lane, ClosestHit / AnyHit of traceray_rq.glsl:108-185 with the traversal the stage kernels use.
__global__ __launch_bounds__(64) void k_trace_rays(DevScene S, int n, const float4* rays, float4* out, int anyHit)
{
  extern __shared__ uint2 s_stack[];
  const int i = int(blockIdx.x) * 64 + int(threadIdx.x);
  const float nanv = rt_u2f(0x7fc00000u);
  const float4 a = i < n ? rays[2 * i] : make_float4(nanv, nanv, nanv, nanv), b = i < n ? rays[2 * i + 1] : make_float4(nanv, nanv, 0.f, 0.f);
  RayHit hit; TravCounters tc{0, 0};
  hit.t = RT_INFINITY; hit.gid = 0xffffffffu; hit.u = hit.v = 0.f;
  const f3 o = mk3(a.x, a.y, a.z), d = mk3(a.w, b.x, b.y);
  bool found;
  if(anyHit) found = traceRay<true>(S, o, d, b.z, rt_f2u(b.w), s_stack + threadIdx.x, hit, tc);
  else found = traceRay<false>(S, o, d, RT_INFINITY, rt_f2u(b.w), s_stack + threadIdx.x, hit, tc);
  if(i < n) out[i] = anyHit ? make_float4(found ? 1.0f : 0.0f, 0.f, 0.f, 0.f) : make_float4(hit.t, rt_u2f(hit.gid), hit.u, hit.v);
}
hipError_t launchTraceRays(hipStream_t stream, const DevScene& S, int n, const float4* rays, float4* out, int anyHit)
{
  hipLaunchKernelGGL(k_trace_rays, dim3(unsigned((n + 63) / 64)), dim3(64), size_t(S.stackEntries) * 64 * sizeof(uint2), stream, S, n, rays, out, anyHit);
  return hipGetLastError();
}

hipError_t launchPick(hipStream_t stream, const DevScene& S, const rt_mat4& viewInv, const rt_mat4& projInv, float pickX, float pickY, rt_pick_result* out)
{
  hipLaunchKernelGGL(k_pick, dim3(1), dim3(64), size_t(S.stackEntries) * 64 * sizeof(uint2), stream, S, viewInv, projInv, pickX, pickY, out);
  return hipGetLastError();
}

hipError_t launchTonemap(hipStream_t stream, const float4* direct, const float4* indirect, double* rowSums, float* mean, const rt_tonemapper& tm, int dbg, int W, int H,
                         uint32_t* ldr, float4* mipD, float4* mipI)
{
  if(tm.autoExposure & 1) {
    hipLaunchKernelGGL(k_post_rowsum, dim3(unsigned(2 * H)), dim3(64), 0, stream, direct, indirect, W, H, rowSums);
    hipLaunchKernelGGL(k_post_mean, dim3(1), dim3(64), 0, stream, (const double*)rowSums, W, H, mean);
  }
  MipLevels MD{}, MI{};
  MD.img[0] = direct; MI.img[0] = indirect; MD.w[0] = MI.w[0] = W; MD.h[0] = MI.h[0] = H;
  if((tm.autoExposure & 3) == 3 && mipD && mipI) {   // RenderOutput::genMipmap: levels 1..7 of both images, each from the one above
    float4* pd = mipD; float4* pi = mipI;
    for(int l = 1; l < 8; l++) {
      const int w = std::max(1, MD.w[l - 1] / 2), h = std::max(1, MD.h[l - 1] / 2);
      MD.w[l] = MI.w[l] = w; MD.h[l] = MI.h[l] = h;
      const dim3 g(unsigned((w + 63) / 64), unsigned((h + 3) / 4));
      hipLaunchKernelGGL(k_post_mip, g, dim3(256), 0, stream, MD.img[l - 1], MD.w[l - 1], MD.h[l - 1], pd, w, h);
      hipLaunchKernelGGL(k_post_mip, g, dim3(256), 0, stream, MI.img[l - 1], MI.w[l - 1], MI.h[l - 1], pi, w, h);
      MD.img[l] = pd; MI.img[l] = pi;
      pd += size_t(w) * h; pi += size_t(w) * h;
    }
  } else {
    for(int l = 1; l < 8; l++) { MD.img[l] = direct; MI.img[l] = indirect; MD.w[l] = MI.w[l] = W; MD.h[l] = MI.h[l] = H; }
  }
  hipLaunchKernelGGL(k_tonemap, dim3(unsigned((W + 63) / 64), unsigned((H + 3) / 4)), dim3(256), 0, stream, direct, indirect, (const float*)mean, tm, dbg, W, H, ldr, MD, MI);
  return hipGetLastError();
}

}  // namespace rt
