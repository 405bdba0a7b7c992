// orc_post.cpp — CPU restatement of the display pass (TEST INFRASTRUCTURE; see oracle/README.md).
//   shaders/post.frag          :50-55 dither, :57-68 luminance / toneExposure, :103-175 main
//   shaders/tonemapping.glsl   :25-40 gamma curves, :48-66 Uncharted 2, :89-105 toneMap
//   shaders/random.glsl        :81-92 pcg3d
//   src/render_output.cpp      :224-254 RenderOutput::run / genMipmap
// PARITY UNPINNED for this file: the reference draws this pass with a fragment shader into a swapchain (needs a Vulkan
// device + glslang); no golden image exists.  Restated choices that the reference leaves to the driver are listed in
// include/rt_abi.h at rt_tonemap (image mean for the top mip level, nearest texel for tm.zoom, the blit formula of the
// mip levels the "local" auto-exposure bit samples).  The pass's pure helpers ARE pinned by vectors minted from the GLSL itself (tests/test_kat_float.py): pcg3d,
// toneExposure and the dither step bit-exactly, toneMapUncharted to 1 ulp; main()'s sequence is restated.
#include "orc_stages.h"

namespace orc {

static inline vec3 vpow(vec3 c, float e) { return {rt_pow(c.x, e), rt_pow(c.y, e), rt_pow(c.z, e)}; }
static inline vec3 linearTosRGB(vec3 c) { const float INV_GAMMA = 1.0f / 2.2f; return vpow(c, INV_GAMMA); }
static inline vec3 sRGBToLinear(vec3 c) { const float GAMMA = 2.2f; return vpow(c, GAMMA); }

static vec3 toneMapUncharted2Impl(vec3 color)
{
  const float A = 0.15f, B = 0.50f, C = 0.10f, D = 0.20f, E = 0.02f, F = 0.30f;
  vec3 num = color * (A * color + C * B) + D * E;
  vec3 den = color * (A * color + B) + D * F;
  vec3 q = num / den;
  const float ef = E / F;
  return {q.x - ef, q.y - ef, q.z - ef};
}
static vec3 toneMapUncharted(vec3 color)
{
  const float Wp = 11.2f;
  color = toneMapUncharted2Impl(color * 2.0f);
  vec3 w = toneMapUncharted2Impl(vec3{Wp, Wp, Wp});
  vec3 whiteScale{1.0f / w.x, 1.0f / w.y, 1.0f / w.z};
  return linearTosRGB(color * whiteScale);
}
static inline float lumPost(vec3 c) { return (c.x * 0.2126f + c.y * 0.7152f) + c.z * 0.0722f; }

static vec3 toneExposure(const rt_tonemapper& tm, vec3 RGB, float logAvgLum)
{
  // RGB2XYZ is built column by column (mat3 constructor), so (RGB2XYZ * RGB).y = m[0][1]*R + m[1][1]*G + m[2][1]*B
  const float XYZy = (0.3575761f * RGB.x + 0.7151522f * RGB.y) + 0.1191920f * RGB.z;
  float Y = (tm.key / logAvgLum) * XYZy;
  float Yd = (Y * (1.0f + Y / (tm.Ywhite * tm.Ywhite))) / (1.0f + Y);
  return (RGB / XYZy) * Yd;
}

static vec3 dither(vec3 linear_color, vec3 noise, float quant)
{
  vec3 s = linearTosRGB(linear_color) / quant;
  vec3 c0 = vec3{rt_floor(s.x), rt_floor(s.y), rt_floor(s.z)} * quant;
  vec3 c1 = c0 + quant;
  vec3 discr = mix(sRGBToLinear(c0), sRGBToLinear(c1), noise);
  return {discr.x < linear_color.x ? c1.x : c0.x, discr.y < linear_color.y ? c1.y : c0.y, discr.z < linear_color.z ? c1.z : c0.z};
}

static void pcg3d(uint32_t v[3])
{
  for(int i = 0; i < 3; i++) v[i] = v[i] * 1664525u + 1013904223u;
  v[0] += v[1] * v[2]; v[1] += v[2] * v[0]; v[2] += v[0] * v[1];
  for(int i = 0; i < 3; i++) v[i] ^= v[i] >> 16u;
  v[0] += v[1] * v[2]; v[1] += v[2] * v[0]; v[2] += v[0] * v[1];
}

// mean colour of an image in the product's association order (csrc/post.hip): 64 interleaved column partial sums per row,
// added in lane order, rows added top to bottom, all in double
static vec3 imageMean(const std::vector<float>& img, int W, int H)
{
  double tot[3] = {0, 0, 0};
  for(int y = 0; y < H; y++) {
    double part[64][3];
    for(int l = 0; l < 64; l++) {
      double s[3] = {0, 0, 0};
      for(int x = l; x < W; x += 64) for(int c = 0; c < 3; c++) s[c] += double(img[(size_t(y) * W + x) * 4 + c]);
      for(int c = 0; c < 3; c++) part[l][c] = s[c];
    }
    double row[3] = {0, 0, 0};
    for(int l = 0; l < 64; l++) for(int c = 0; c < 3; c++) row[c] += part[l][c];
    for(int c = 0; c < 3; c++) tot[c] += row[c];
  }
  const double n = double(W) * double(H);
  return {float(tot[0] / n), float(tot[1] / n), float(tot[2] / n)};
}

// RenderOutput::genMipmap (render_output.cpp:243-254): vkCmdBlitImage with VK_FILTER_LINEAR level by level — the destination texel centre
// maps to (x + 0.5) * sw / dw in the source, sampled with a clamped bilinear filter (2 x 2 box average for even sizes)
struct Mip { std::vector<float> img; int w = 0, h = 0; };
static vec4 bilinearClamp(const float* src, int sw, int sh, float fx, float fy)
{
  const float x0f = rt_floor(fx), y0f = rt_floor(fy);
  const float ax = fx - x0f, ay = fy - y0f;
  const int x0 = rt_ftoi(x0f), y0 = rt_ftoi(y0f);
  const int xa = std::min(std::max(x0, 0), sw - 1), xb = std::min(std::max(x0 + 1, 0), sw - 1), ya = std::min(std::max(y0, 0), sh - 1), yb = std::min(std::max(y0 + 1, 0), sh - 1);
  auto px = [&](int x, int y) { const float* p = src + (size_t(y) * sw + x) * 4; return vec4{p[0], p[1], p[2], p[3]}; };
  return mix(mix(px(xa, ya), px(xb, ya), ax), mix(px(xa, yb), px(xb, yb), ax), ay);
}
static void buildMips(const std::vector<float>& img, int W, int H, Mip (&M)[8])
{
  M[0].w = W; M[0].h = H;   // level 0 is read from `img` directly
  for(int l = 1; l < 8; l++) {
    const int sw = M[l - 1].w, sh = M[l - 1].h, w = std::max(1, sw / 2), h = std::max(1, sh / 2);
    const float* src = l == 1 ? img.data() : M[l - 1].img.data();
    M[l].w = w; M[l].h = h; M[l].img.resize(size_t(w) * h * 4);
    for(int y = 0; y < h; y++)
      for(int x = 0; x < w; x++) {
        const vec4 t = bilinearClamp(src, sw, sh, (float(x) + 0.5f) * (float(sw) / float(w)) - 0.5f, (float(y) + 0.5f) * (float(sh) / float(h)) - 0.5f);
        float* o = &M[l].img[(size_t(y) * w + x) * 4]; o[0] = t.x; o[1] = t.y; o[2] = t.z; o[3] = t.w;
      }
  }
}
static vec3 sampleLevel(const std::vector<float>& img, const Mip (&M)[8], int level, float u, float v)
{
  const float* src = level == 0 ? img.data() : M[level].img.data();
  return xyz(bilinearClamp(src, M[level].w, M[level].h, u * float(M[level].w) - 0.5f, v * float(M[level].h) - 0.5f));
}
// toneLocalExposure, post.frag:70-101 (v2 of the default view is undefined in the reference — `==` instead of `=` at :91 — and 0 here)
static vec3 toneLocalExposure(const rt_tonemapper& tm, int dbg, const std::vector<float>& D, const Mip (&MD)[8], const std::vector<float>& I, const Mip (&MI)[8], float u, float v,
                              vec3 RGB, float logAvgLum)
{
  const float XYZy = (0.3575761f * RGB.x + 0.7151522f * RGB.y) + 0.1191920f * RGB.z;
  const float Y = (tm.key / logAvgLum) * XYZy;
  const float factor = tm.key / logAvgLum;
  const float epsilon = 0.05f, phi = 2.0f;
  float La = 0.0f, scale = 1.0f;
  for(int i = 0; i < 7; ++i) {
    float v1, v2 = 0.0f;
    if(dbg == RT_DBG_DIRECT_STAGE) { v1 = lumPost(sampleLevel(D, MD, i, u, v)) * factor; v2 = lumPost(sampleLevel(D, MD, i + 1, u, v)) * factor; }
    else if(dbg == RT_DBG_INDIRECT_STAGE) { v1 = lumPost(sampleLevel(I, MI, i, u, v)) * factor; v2 = lumPost(sampleLevel(I, MI, i + 1, u, v)) * factor; }
    else v1 = lumPost(sampleLevel(D, MD, i, u, v) + sampleLevel(I, MI, i, u, v)) * factor;
    if(rt_abs(v1 - v2) / ((tm.key * rt_pow(2.0f, phi) / (scale * scale)) + v1) > epsilon) { La = v1; break; }
    La = v2;
    scale = scale * 2.0f;
  }
  const float Yd = Y / (1.0f + La);
  return (RGB / XYZy) * Yd;
}

static inline uint32_t unorm8(float c) { return rt_ftou(rt_floor(rt_clamp(c, 0.0f, 1.0f) * 255.0f + 0.5f)); }

void Frame::tonemap(const rt_tonemapper& tm, int dbg, int frames)
{
  const int cur = frames & 1;
  const std::vector<float>& D = directResult[cur];
  const std::vector<float>& I = indirectResult[cur];
  vec3 avgD{0, 0, 0}, avgI{0, 0, 0};
  if(tm.autoExposure & 1) { avgD = imageMean(D, W, H); avgI = imageMean(I, W, H); }
  Mip MD[8], MI[8];
  const bool local = (tm.autoExposure & 3) == 3;
  if(local) { buildMips(D, W, H, MD); buildMips(I, W, H, MI); }
  parallelRows(H, 0, 0, [&](int y) {
    for(int x = 0; x < W; x++) {
      const float u = (float(x) + 0.5f) / float(W), v = (float(y) + 0.5f) / float(H);  // passthrough.vert: uv at the fragment centre
      int sx = rt_ftoi(u * tm.zoom * float(W)), sy = rt_ftoi(v * tm.zoom * float(H));
      sx = std::min(std::max(sx, 0), W - 1); sy = std::min(std::max(sy, 0), H - 1);
      const vec4 d = loadImg(D, ivec2{sx, sy}), in = loadImg(I, ivec2{sx, sy});
      vec3 color;
      if(dbg == RT_DBG_DEPTH) {
        float depth = d.w;
        depth *= rt_pow(2.0f, tm.brightness);
        depth += tm.saturation;
        depth = rt_clamp(rt_pow(depth, 1.0f / tm.contrast), 0.0f, 1.0f);
        color = {depth, depth, depth};
      } else if(dbg > RT_DBG_INDIRECT_STAGE) {
        color = xyz(d);
        if(dbg == RT_DBG_BASECOLOR) { vec3 p = vpow(color, 0.45454545454545f); color = {rt_clamp(p.x, 0.f, 1.f), rt_clamp(p.y, 0.f, 1.f), rt_clamp(p.z, 0.f, 1.f)}; }
      } else {
        vec3 hdr = dbg == RT_DBG_DIRECT_STAGE ? xyz(d) : (dbg == RT_DBG_INDIRECT_STAGE ? xyz(in) : xyz(d) + xyz(in));
        if(tm.autoExposure & 1) {
          vec3 avg = dbg == RT_DBG_DIRECT_STAGE ? avgD : (dbg == RT_DBG_INDIRECT_STAGE ? avgI : avgD + avgI);
          hdr = local ? toneLocalExposure(tm, dbg, D, MD, I, MI, u * tm.zoom, v * tm.zoom, hdr, lumPost(avg)) : toneExposure(tm, hdr, lumPost(avg));
        }
        color = toneMapUncharted(hdr * tm.avgLum);
        uint32_t r[3] = {uint32_t(x), uint32_t(y), 0u};  // uvec3(gl_FragCoord.xy, 0): x+0.5 truncates to x
        pcg3d(r);
        vec3 noise{rt_u2f(0x3f800000u | (r[0] >> 9)) - 1.0f, rt_u2f(0x3f800000u | (r[1] >> 9)) - 1.0f, rt_u2f(0x3f800000u | (r[2] >> 9)) - 1.0f};
        color = dither(sRGBToLinear(color), noise, 1.0f / 255.0f);
        color = mix(vec3{0.5f, 0.5f, 0.5f}, color, tm.contrast);
        color = {rt_clamp(color.x, 0.f, 1.f), rt_clamp(color.y, 0.f, 1.f), rt_clamp(color.z, 0.f, 1.f)};
        color = vpow(color, 1.0f / tm.brightness);
        const float i = (color.x * 0.299f + color.y * 0.587f) + color.z * 0.114f;
        color = mix(vec3{i, i, i}, color, tm.saturation);
        const float ux = ((u * tm.renderingRatio.x) - 0.5f) * 2.0f, uy = ((v * tm.renderingRatio.y) - 0.5f) * 2.0f;
        color = color * (1.0f - (ux * ux + uy * uy) * tm.vignette);
      }
      ldr[size_t(y) * W + x] = unorm8(color.x) | (unorm8(color.y) << 8) | (unorm8(color.z) << 16) | 0xff000000u;
    }
  });
}

}  // namespace orc

// ---- display-pass helpers for the known-answer vectors minted from the reference's GLSL (tests/test_kat_float.py) ----
// op 0: toneMapUncharted(in[0..2])                       1: dither(sRGBToLinear(in[0..2]), noise = in[3..5], 1/255)
// op 2: toneExposure(key = in[0], Ywhite = in[1], RGB = in[2..4], logAvgLum = in[5])
extern "C" void orc_post_fn(int op, const float* in, float* out)
{
  using namespace orc;
  vec3 r{0, 0, 0};
  if(op == 0) r = toneMapUncharted(vec3{in[0], in[1], in[2]});
  else if(op == 1) r = dither(sRGBToLinear(vec3{in[0], in[1], in[2]}), vec3{in[3], in[4], in[5]}, 1.0f / 255.0f);
  else if(op == 2) { rt_tonemapper tm; memset(&tm, 0, sizeof(tm)); tm.key = in[0]; tm.Ywhite = in[1]; r = toneExposure(tm, vec3{in[2], in[3], in[4]}, in[5]); }
  out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
extern "C" void orc_pcg3d(uint32_t* v) { orc::pcg3d(v); }
