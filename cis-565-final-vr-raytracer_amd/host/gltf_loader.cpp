// gltf_loader.cpp — glTF 2.0 ingest for Scene::load (src/scene.cpp:57-173).
//
// The reference delegates this to tinygltf + nvh::GltfScene::importMaterials / importDrawableNodes (scene.cpp:72-74) +
// FreeImage — all third-party, un-vendored, absent here.  This is a from-scratch reader for the subset the
// path consumes: .gltf (external or data: URIs) and .glb containers; scene graph (matrix / TRS) flattened to world matrices,
// one GltfPrimMesh per (mesh, primitive) and one GltfNode per drawable instance like nvh::GltfScene; triangle primitives
// with POSITION / NORMAL / TANGENT / TEXCOORD_0 / COLOR_0 and any index type; pbrMetallicRoughness materials with
// KHR_materials_transmission, KHR_materials_ior, KHR_materials_emissive_strength; KHR_lights_punctual; perspective cameras;
// samplers; PNG images (every colour type / bit depth, Adam7 included, via zlib) and JPEG images (baseline / progressive, jpeg_decoder.cpp).
#include "scene.hpp"
#include "../../include/rt_cpus.h"
#include <zlib.h>
#include <atomic>
#include <mutex>
#include <thread>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <map>
#include <memory>
#include <sstream>

namespace rth {
bool decodeJpeg(const uint8_t* d, size_t n, TextureImage& img);  // jpeg_decoder.cpp
namespace {

// ---------------------------------------------------------------------------------------------------------- JSON
struct Json {
  enum Type { Null, Bool, Num, Str, Arr, Obj } type = Null;
  bool b = false;
  double num = 0;
  std::string str;
  std::vector<Json> arr;
  std::vector<std::pair<std::string, Json>> obj;

  const Json* get(const char* key) const
  {
    if(type != Obj) return nullptr;
    for(const auto& kv : obj) if(kv.first == key) return &kv.second;
    return nullptr;
  }
  double number(const char* key, double def) const { const Json* j = get(key); return (j && j->type == Num) ? j->num : def; }
  int integer(const char* key, int def) const { return int(number(key, def)); }
  std::string string(const char* key, const std::string& def = "") const { const Json* j = get(key); return (j && j->type == Str) ? j->str : def; }
  bool boolean(const char* key, bool def) const { const Json* j = get(key); return (j && j->type == Bool) ? j->b : def; }
  size_t size() const { return type == Arr ? arr.size() : 0; }
  const Json& at(size_t i) const { static const Json nul; return (type == Arr && i < arr.size()) ? arr[i] : nul; }
};

struct JsonParser {
  const char* p; const char* end; bool ok = true;
  void ws() { while(p < end && (*p == ' ' || *p == '\n' || *p == '\r' || *p == '\t')) p++; }
  Json parse()
  {
    ws();
    Json j;
    if(p >= end) { ok = false; return j; }
    if(*p == '{') {
      j.type = Json::Obj; p++; ws();
      if(p < end && *p == '}') { p++; return j; }
      while(ok && p < end) {
        ws(); Json k = parse();
        if(k.type != Json::Str) { ok = false; break; }
        ws(); if(p >= end || *p != ':') { ok = false; break; } p++;
        j.obj.emplace_back(k.str, parse());
        ws(); if(p < end && *p == ',') { p++; continue; }
        if(p < end && *p == '}') { p++; break; }
        ok = false;
      }
    } else if(*p == '[') {
      j.type = Json::Arr; p++; ws();
      if(p < end && *p == ']') { p++; return j; }
      while(ok && p < end) {
        j.arr.push_back(parse());
        ws(); if(p < end && *p == ',') { p++; continue; }
        if(p < end && *p == ']') { p++; break; }
        ok = false;
      }
    } else if(*p == '"') {
      j.type = Json::Str; p++;
      while(p < end && *p != '"') {
        if(*p == '\\' && p + 1 < end) {
          p++;
          switch(*p) {
            case 'n': j.str += '\n'; break; case 't': j.str += '\t'; break; case 'r': j.str += '\r'; break;
            case 'b': j.str += '\b'; break; case 'f': j.str += '\f'; break;
            case 'u': { unsigned cp = 0; if(p + 4 < end) { cp = unsigned(strtoul(std::string(p + 1, p + 5).c_str(), nullptr, 16)); p += 4; }
                        if(cp < 0x80) j.str += char(cp); else if(cp < 0x800) { j.str += char(0xC0 | (cp >> 6)); j.str += char(0x80 | (cp & 0x3F)); }
                        else { j.str += char(0xE0 | (cp >> 12)); j.str += char(0x80 | ((cp >> 6) & 0x3F)); j.str += char(0x80 | (cp & 0x3F)); } break; }
            default: j.str += *p;
          }
          p++;
        } else j.str += *p++;
      }
      if(p < end) p++; else ok = false;
    } else if(!strncmp(p, "true", 4)) { j.type = Json::Bool; j.b = true; p += 4; }
    else if(!strncmp(p, "false", 5)) { j.type = Json::Bool; j.b = false; p += 5; }
    else if(!strncmp(p, "null", 4)) { p += 4; }
    else {
      char* e = nullptr;
      j.num = strtod(p, &e);
      if(e == p) { ok = false; return j; }
      j.type = Json::Num; p = e;
    }
    return j;
  }
};

// -------------------------------------------------------------------------------------------------------- helpers
bool readFile(const std::string& path, std::vector<uint8_t>& out)
{
  std::ifstream f(path, std::ios::binary);
  if(!f) return false;
  f.seekg(0, std::ios::end); std::streamoff n = f.tellg(); f.seekg(0);
  out.resize(size_t(n));
  if(n) f.read(reinterpret_cast<char*>(out.data()), n);
  return bool(f) || n == 0;
}
std::vector<uint8_t> base64(const std::string& s, size_t from)
{
  std::vector<uint8_t> out; unsigned acc = 0; int bits = 0;
  for(size_t i = from; i < s.size(); i++) {
    char c = s[i]; int v;
    if(c >= 'A' && c <= 'Z') v = c - 'A'; else if(c >= 'a' && c <= 'z') v = c - 'a' + 26; else if(c >= '0' && c <= '9') v = c - '0' + 52;
    else if(c == '+' || c == '-') v = 62; else if(c == '/' || c == '_') v = 63; else continue;
    acc = (acc << 6) | unsigned(v); bits += 6;
    if(bits >= 8) { bits -= 8; out.push_back(uint8_t((acc >> bits) & 0xff)); }
  }
  return out;
}
bool loadUri(const std::string& uri, const std::string& dir, std::vector<uint8_t>& out)
{
  if(uri.rfind("data:", 0) == 0) { size_t c = uri.find(','); if(c == std::string::npos) return false; out = base64(uri, c + 1); return true; }
  std::string path;  // percent-decoding of the few characters exporters escape
  for(size_t i = 0; i < uri.size(); i++) {
    if(uri[i] == '%' && i + 2 < uri.size()) { path += char(strtol(uri.substr(i + 1, 2).c_str(), nullptr, 16)); i += 2; }
    else path += uri[i];
  }
  return readFile(dir + path, out);
}

// PNG: every colour type and bit depth of the specification (gray 1/2/4/8/16, RGB 8/16, palette 1/2/4/8, gray+alpha 8/16, RGBA 8/16), non-interlaced and
// Adam7-interlaced, tRNS for palette images.  Output BGRA8 (scene.cpp:559: VK_FORMAT_B8G8R8A8_UNORM, FreeImage's native channel order); 16-bit samples keep
// their high byte, sub-byte gray samples are scaled to 0..255.  Every length is checked before it is used: a malformed file is `false`, never a fault
// (tests/test_ingest_fuzz.py).
constexpr uint32_t PNG_MAX_DIM = 16384;
bool decodePng(const uint8_t* d, size_t n, TextureImage& img)
{
  static const uint8_t sig[8] = {137, 80, 78, 71, 13, 10, 26, 10};
  if(n < 8 || memcmp(d, sig, 8) != 0) return false;
  auto be32 = [](const uint8_t* p) { return (uint32_t(p[0]) << 24) | (uint32_t(p[1]) << 16) | (uint32_t(p[2]) << 8) | p[3]; };
  uint32_t w = 0, h = 0; int depth = 0, ctype = -1, interlace = 0; bool haveHdr = false;
  std::vector<uint8_t> idat, plte, trns;
  size_t pos = 8;
  while(pos + 12 <= n) {
    const uint32_t len = be32(d + pos);
    const uint8_t* tag = d + pos + 4; const uint8_t* body = d + pos + 8;
    if(len > n || pos + 12 + size_t(len) > n) return false;
    if(!memcmp(tag, "IHDR", 4)) {
      if(len < 13) return false;
      w = be32(body); h = be32(body + 4); depth = body[8]; ctype = body[9]; interlace = body[12]; haveHdr = true;
      if(body[10] != 0 || body[11] != 0) return false;   // compression / filter method
    }
    else if(!memcmp(tag, "PLTE", 4)) plte.assign(body, body + len);
    else if(!memcmp(tag, "tRNS", 4)) trns.assign(body, body + len);
    else if(!memcmp(tag, "IDAT", 4)) idat.insert(idat.end(), body, body + len);
    else if(!memcmp(tag, "IEND", 4)) break;
    pos += 12 + size_t(len);
  }
  if(!haveHdr || !w || !h || w > PNG_MAX_DIM || h > PNG_MAX_DIM || interlace > 1 || idat.empty()) return false;
  const int chn = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
  if(!chn) return false;
  const bool depthOk = (ctype == 0 && (depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16)) || (ctype == 3 && (depth == 1 || depth == 2 || depth == 4 || depth == 8)) ||
                       ((ctype == 2 || ctype == 4 || ctype == 6) && (depth == 8 || depth == 16));
  if(!depthOk) return false;
  const int bpp = chn * depth;                       // bits per pixel
  const int fd = std::max(1, bpp / 8);               // filter distance in bytes
  auto rowBytes = [&](uint32_t pw) { return (size_t(pw) * size_t(bpp) + 7) / 8; };
  // the passes: {x0, y0, dx, dy}; a non-interlaced image is one pass
  static const int adam7[7][4] = {{0, 0, 8, 8}, {4, 0, 8, 8}, {0, 4, 4, 8}, {2, 0, 4, 4}, {0, 2, 2, 4}, {1, 0, 2, 2}, {0, 1, 1, 2}};
  static const int single[1][4] = {{0, 0, 1, 1}};
  const int (*passes)[4] = interlace ? adam7 : single; const int npass = interlace ? 7 : 1;
  size_t total = 0;
  for(int k = 0; k < npass; k++) {
    const uint32_t pw = (w - uint32_t(passes[k][0]) + uint32_t(passes[k][2]) - 1) / uint32_t(passes[k][2]), ph = (h - uint32_t(passes[k][1]) + uint32_t(passes[k][3]) - 1) / uint32_t(passes[k][3]);
    if(uint32_t(passes[k][0]) >= w || uint32_t(passes[k][1]) >= h || !pw || !ph) continue;
    total += (rowBytes(pw) + 1) * ph;
  }
  // deflate expands by at most 1032 : 1 (a stored length-258 match per ~2 bits): a header that promises more than the IDAT stream can hold is rejected BEFORE
  // the multi-GB buffers it asks for are allocated (round-3 advisor: a 100-byte PNG claiming 16384 x 16384 x 64 bpp forced ~4 GB of zero-filled vectors)
  if(total > idat.size() * size_t(1032) + 1024) return false;
  std::vector<uint8_t> raw(total);
  uLongf rawLen = uLongf(raw.size());
  if(uncompress(raw.data(), &rawLen, idat.data(), uLong(idat.size())) != Z_OK || rawLen != raw.size()) return false;
  std::vector<uint8_t> px(size_t(w) * h * chn);      // one byte per sample
  const int maxv = (1 << std::min(depth, 8)) - 1;
  size_t off = 0;
  std::vector<uint8_t> prev, cur;
  for(int k = 0; k < npass; k++) {
    if(uint32_t(passes[k][0]) >= w || uint32_t(passes[k][1]) >= h) continue;
    const uint32_t pw = (w - uint32_t(passes[k][0]) + uint32_t(passes[k][2]) - 1) / uint32_t(passes[k][2]), ph = (h - uint32_t(passes[k][1]) + uint32_t(passes[k][3]) - 1) / uint32_t(passes[k][3]);
    if(!pw || !ph) continue;
    const size_t stride = rowBytes(pw);
    prev.assign(stride, 0); cur.assign(stride, 0);
    for(uint32_t y = 0; y < ph; y++) {  // un-filter one scanline of the pass, then scatter its samples
      const uint8_t ft = raw[off]; const uint8_t* in = &raw[off + 1];
      off += stride + 1;
      if(ft > 4) return false;
      for(size_t x = 0; x < stride; x++) {
        const int a = x >= size_t(fd) ? cur[x - size_t(fd)] : 0, b = prev[x], c = x >= size_t(fd) ? prev[x - size_t(fd)] : 0;
        int v = in[x];
        switch(ft) {
          case 1: v += a; break; case 2: v += b; break; case 3: v += (a + b) >> 1; break;
          case 4: { const int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c); v += (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); break; }
          default: break;
        }
        cur[x] = uint8_t(v);
      }
      const size_t Y = size_t(passes[k][1]) + size_t(y) * size_t(passes[k][3]);
      for(uint32_t x = 0; x < pw; x++) {
        const size_t X = size_t(passes[k][0]) + size_t(x) * size_t(passes[k][2]);
        uint8_t* o = &px[(Y * w + X) * size_t(chn)];
        for(int s2 = 0; s2 < chn; s2++) {
          const size_t bit = (size_t(x) * size_t(chn) + size_t(s2)) * size_t(depth);
          if(depth == 8) o[s2] = cur[bit / 8];
          else if(depth == 16) o[s2] = cur[bit / 8];          // big-endian: the high byte
          else { const int v = (cur[bit / 8] >> (8 - depth - int(bit % 8))) & maxv; o[s2] = (ctype == 3) ? uint8_t(v) : uint8_t(v * 255 / maxv); }
        }
      }
      prev.swap(cur);
    }
  }
  img.width = int(w); img.height = int(h); img.bgra.resize(size_t(w) * h * 4);
  for(size_t i = 0; i < size_t(w) * h; i++) {
    uint8_t r, g, b, a = 255;
    const uint8_t* s = &px[i * chn];
    if(ctype == 0) { r = g = b = s[0]; }
    else if(ctype == 4) { r = g = b = s[0]; a = s[1]; }
    else if(ctype == 3) { size_t k = s[0]; if(k * 3 + 2 >= plte.size()) { r = g = b = 0; } else { r = plte[k * 3]; g = plte[k * 3 + 1]; b = plte[k * 3 + 2]; } if(k < trns.size()) a = trns[k]; }
    else { r = s[0]; g = s[1]; b = s[2]; if(chn == 4) a = s[3]; }
    uint8_t* o = &img.bgra[i * 4];
    o[0] = b; o[1] = g; o[2] = r; o[3] = a;
  }
  return true;
}

struct Accessor { const uint8_t* data = nullptr; size_t count = 0, stride = 0; int comp = 5126, ncomp = 1; bool normalized = false; };

float readComp(const Accessor& a, size_t i, int c)
{
  const uint8_t* p = a.data + i * a.stride;
  switch(a.comp) {
    case 5126: { float f; memcpy(&f, p + 4 * c, 4); return f; }
    case 5121: { uint8_t v = p[c]; return a.normalized ? v / 255.f : float(v); }
    case 5120: { int8_t v = int8_t(p[c]); return a.normalized ? std::max(v / 127.f, -1.f) : float(v); }
    case 5123: { uint16_t v; memcpy(&v, p + 2 * c, 2); return a.normalized ? v / 65535.f : float(v); }
    case 5122: { int16_t v; memcpy(&v, p + 2 * c, 2); return a.normalized ? std::max(v / 32767.f, -1.f) : float(v); }
    case 5125: { uint32_t v; memcpy(&v, p + 4 * c, 4); return float(v); }
  }
  return 0.f;
}
uint32_t readIndex(const Accessor& a, size_t i)
{
  const uint8_t* p = a.data + i * a.stride;
  switch(a.comp) { case 5121: return p[0]; case 5123: { uint16_t v; memcpy(&v, p, 2); return v; } case 5125: { uint32_t v; memcpy(&v, p, 4); return v; } }
  return 0;
}

M4 nodeLocalMatrix(const Json& n)
{
  if(const Json* m = n.get("matrix")) if(m->size() == 16) { M4 r; for(int i = 0; i < 16; i++) r.m[i] = float(m->at(size_t(i)).num); return r; }  // column-major like glTF
  float t[3] = {0, 0, 0}, q[4] = {0, 0, 0, 1}, s[3] = {1, 1, 1};
  if(const Json* j = n.get("translation")) for(int i = 0; i < 3 && i < int(j->size()); i++) t[i] = float(j->at(size_t(i)).num);
  if(const Json* j = n.get("rotation")) for(int i = 0; i < 4 && i < int(j->size()); i++) q[i] = float(j->at(size_t(i)).num);
  if(const Json* j = n.get("scale")) for(int i = 0; i < 3 && i < int(j->size()); i++) s[i] = float(j->at(size_t(i)).num);
  const float x = q[0], y = q[1], z = q[2], w = q[3];
  M4 r = M4::identity();
  r.at(0, 0) = (1 - 2 * (y * y + z * z)) * s[0]; r.at(0, 1) = (2 * (x * y - z * w)) * s[1]; r.at(0, 2) = (2 * (x * z + y * w)) * s[2];
  r.at(1, 0) = (2 * (x * y + z * w)) * s[0]; r.at(1, 1) = (1 - 2 * (x * x + z * z)) * s[1]; r.at(1, 2) = (2 * (y * z - x * w)) * s[2];
  r.at(2, 0) = (2 * (x * z - y * w)) * s[0]; r.at(2, 1) = (2 * (y * z + x * w)) * s[1]; r.at(2, 2) = (1 - 2 * (x * x + y * y)) * s[2];
  r.at(0, 3) = t[0]; r.at(1, 3) = t[1]; r.at(2, 3) = t[2];
  return r;
}

}  // namespace
// exported for the host C API (PNG round-trip test of host/png_writer.cpp)
bool decodePngImage(const uint8_t* d, size_t n, TextureImage& img) { return decodePng(d, n, img); }

bool loadGltfFile(const std::string& filename, GltfScene& out, std::string& error)
{
  std::vector<uint8_t> file;
  if(!readFile(filename, file)) { error = "cannot read " + filename; return false; }
  const size_t slash = filename.find_last_of("/\\");
  const std::string dir = slash == std::string::npos ? "" : filename.substr(0, slash + 1);
  std::vector<uint8_t> glbBin;
  const char* jb = reinterpret_cast<const char*>(file.data());
  size_t jl = file.size();
  if(file.size() >= 20 && !memcmp(file.data(), "glTF", 4)) {  // .glb: 12-byte header, JSON chunk, optional BIN chunk
    uint32_t len0; memcpy(&len0, &file[12], 4);
    if(20 + size_t(len0) > file.size()) { error = "truncated glb"; return false; }
    jb = reinterpret_cast<const char*>(&file[20]); jl = len0;
    const size_t p = 20 + len0;
    if(p + 8 <= file.size()) { uint32_t len1; memcpy(&len1, &file[p], 4); if(p + 8 + len1 <= file.size()) glbBin.assign(file.begin() + long(p) + 8, file.begin() + long(p) + 8 + len1); }
  }
  JsonParser jp{jb, jb + jl};
  const Json root = jp.parse();
  if(!jp.ok || root.type != Json::Obj) { error = "JSON parse error in " + filename; return false; }
  static const Json none;
  auto A = [&](const char* k) -> const Json& { const Json* j = root.get(k); return j ? *j : none; };

  // ---- buffers / views / accessors --------------------------------------------------------------------------------
  std::vector<std::vector<uint8_t>> buffers(A("buffers").size());
  for(size_t i = 0; i < buffers.size(); i++) {
    const std::string uri = A("buffers").at(i).string("uri");
    if(uri.empty()) buffers[i] = glbBin;
    else if(!loadUri(uri, dir, buffers[i])) { error = "cannot load buffer " + uri; return false; }
  }
  auto accessor = [&](int idx, Accessor& a) -> bool {
    if(idx < 0 || size_t(idx) >= A("accessors").size()) return false;
    const Json& ac = A("accessors").at(size_t(idx));
    const int bv = ac.integer("bufferView", -1);
    if(bv < 0 || size_t(bv) >= A("bufferViews").size()) return false;  // sparse / zero-filled accessors are not supported
    const Json& view = A("bufferViews").at(size_t(bv));
    const int b = view.integer("buffer", 0);
    if(b < 0 || size_t(b) >= buffers.size()) return false;
    const std::string type = ac.string("type", "SCALAR");
    a.ncomp = type == "SCALAR" ? 1 : type == "VEC2" ? 2 : type == "VEC3" ? 3 : type == "VEC4" ? 4 : type == "MAT4" ? 16 : 1;
    a.comp = ac.integer("componentType", 5126);
    const size_t csz = (a.comp == 5120 || a.comp == 5121) ? 1 : (a.comp == 5122 || a.comp == 5123) ? 2 : 4;
    // count / stride / offsets come from the file as doubles: reject anything that is not a finite, non-negative integer small
    // enough that the byte range below cannot wrap (a malformed file must not become an out-of-bounds read)
    auto asSize = [](double v, size_t limit, size_t& out) { if(!(v >= 0.0) || !(v <= double(limit)) || v != std::floor(v)) return false; out = size_t(v); return true; };
    const size_t bufSize = buffers[size_t(b)].size();
    size_t viewOff = 0, accOff = 0, strideIn = 0;
    if(!asSize(ac.number("count", 0), size_t(1) << 40, a.count) || !asSize(view.number("byteStride", 0), 4096, strideIn) ||
       !asSize(view.number("byteOffset", 0), bufSize, viewOff) || !asSize(ac.number("byteOffset", 0), bufSize, accOff)) return false;
    a.normalized = ac.boolean("normalized", false);
    a.stride = strideIn ? strideIn : csz * size_t(a.ncomp);
    const size_t off = viewOff + accOff, elem = csz * size_t(a.ncomp);
    if(off > bufSize || elem > bufSize - off) return false;
    if(a.count && (a.count - 1) > (bufSize - off - elem) / a.stride) return false;   // division form: no multiplication that could overflow
    a.data = buffers[size_t(b)].data() + off;
    return true;
  };

  // ---- images -> textures (scene.cpp:554-646) ----------------------------------------------------------------------
  // (round 6: the images of an asset are decoded by all host cores — Bistro Exterior has 251 of them, 3.4 GB of texels: one core took a minute)
  std::vector<TextureImage> images(A("images").size());
  {
    std::atomic<size_t> nextImage{0};
    std::mutex logLock;
    auto work = [&] {
      for(;;) {
        const size_t i = nextImage.fetch_add(1);
        if(i >= images.size()) return;
        const Json& im = A("images").at(i);
        std::vector<uint8_t> bytes;
        const uint8_t* data = nullptr; size_t size = 0;
        bool have = false;
        const std::string uri = im.string("uri");
        if(!uri.empty()) { have = loadUri(uri, dir, bytes); data = bytes.data(); size = bytes.size(); }
        else {
          const int bv = im.integer("bufferView", -1);
          if(bv >= 0 && size_t(bv) < A("bufferViews").size()) {
            const Json& view = A("bufferViews").at(size_t(bv));
            const int b = view.integer("buffer", 0);
            const size_t off = size_t(view.number("byteOffset", 0)), len = size_t(view.number("byteLength", 0));
            if(b >= 0 && size_t(b) < buffers.size() && off + len <= buffers[size_t(b)].size()) { data = buffers[size_t(b)].data() + off; size = len; have = true; }
          }
        }
        TextureImage t;
        if(!have || !(decodePng(data, size, t) || decodeJpeg(data, size, t))) {
          if(have) { std::lock_guard<std::mutex> l(logLock); fprintf(stderr, "gltf: image %zu is neither an 8-bit PNG nor a Huffman-coded 8-bit JPEG: using a white texel\n", i); }
          t.width = t.height = 1; t.bgra = {255, 255, 255, 255};  // addDefaultImage, scene.cpp:566-572
        }
        images[i] = std::move(t);
      }
    };
    const size_t nt = std::min<size_t>(images.size(), size_t(rt_cpu_budget()));
    std::vector<std::thread> pool;
    for(size_t k = 1; k < nt; k++) pool.emplace_back(work);
    work();
    for(auto& th : pool) th.join();
  }
  // (an image used by exactly one texture — the usual case — is moved, not copied: 16 MB per 2048^2 image)
  std::vector<int> imageUses(images.size(), 0);
  for(size_t i = 0; i < A("textures").size(); i++) { const int src = A("textures").at(i).integer("source", -1); if(src >= 0 && size_t(src) < images.size()) imageUses[size_t(src)]++; }
  for(size_t i = 0; i < A("textures").size(); i++) {
    const Json& tx = A("textures").at(i);
    const int src = tx.integer("source", -1);
    TextureImage t;
    if(src >= 0 && size_t(src) < images.size()) { if(--imageUses[size_t(src)] == 0) t = std::move(images[size_t(src)]); else t = images[size_t(src)]; }
    else { t.width = t.height = 1; t.bgra = {255, 255, 255, 255}; }
    const int smp = tx.integer("sampler", -1);
    if(smp >= 0 && size_t(smp) < A("samplers").size()) {  // gltfSamplerToVulkan, scene.cpp:513-548
      const Json& s = A("samplers").at(size_t(smp));
      t.wrapS = s.integer("wrapS", RT_WRAP_REPEAT); t.wrapT = s.integer("wrapT", RT_WRAP_REPEAT);
      const int mag = s.integer("magFilter", RT_FILTER_LINEAR);
      t.magFilter = (mag == RT_FILTER_NEAREST) ? RT_FILTER_NEAREST : RT_FILTER_LINEAR;
    }
    out.textures.push_back(std::move(t));
  }

  // ---- materials (nvh::GltfScene::importMaterials; fields consumed by scene.cpp:415-448) ----------------------------------
  auto texIndex = [](const Json* j) { return j ? j->integer("index", -1) : -1; };
  for(size_t i = 0; i < A("materials").size(); i++) {
    const Json& m = A("materials").at(i);
    GltfMaterial g;
    if(const Json* pbr = m.get("pbrMetallicRoughness")) {
      if(const Json* f = pbr->get("baseColorFactor")) for(int k = 0; k < 4 && k < int(f->size()); k++) g.baseColorFactor[k] = float(f->at(size_t(k)).num);
      g.baseColorTexture = texIndex(pbr->get("baseColorTexture"));
      g.metallicFactor = float(pbr->number("metallicFactor", 1.0));
      g.roughnessFactor = float(pbr->number("roughnessFactor", 1.0));
      g.metallicRoughnessTexture = texIndex(pbr->get("metallicRoughnessTexture"));
    }
    if(const Json* nt = m.get("normalTexture")) { g.normalTexture = nt->integer("index", -1); g.normalTextureScale = float(nt->number("scale", 1.0)); }
    g.emissiveTexture = texIndex(m.get("emissiveTexture"));
    if(const Json* e = m.get("emissiveFactor")) for(int k = 0; k < 3 && k < int(e->size()); k++) g.emissiveFactor[k] = float(e->at(size_t(k)).num);
    const std::string am = m.string("alphaMode", "OPAQUE");
    g.alphaMode = am == "MASK" ? RT_ALPHA_MASK : am == "BLEND" ? RT_ALPHA_BLEND : RT_ALPHA_OPAQUE;
    g.alphaCutoff = float(m.number("alphaCutoff", 0.5));
    g.doubleSided = m.boolean("doubleSided", false) ? 1 : 0;
    if(const Json* ext = m.get("extensions")) {
      if(const Json* t = ext->get("KHR_materials_transmission")) { g.transmissionFactor = float(t->number("transmissionFactor", 0.0)); g.transmissionTexture = texIndex(t->get("transmissionTexture")); }
      if(const Json* t = ext->get("KHR_materials_ior")) g.ior = float(t->number("ior", 1.5));
      if(const Json* t = ext->get("KHR_materials_emissive_strength")) { const float s = float(t->number("emissiveStrength", 1.0)); for(float& e : g.emissiveFactor) e *= s; }
    }
    out.materials.push_back(g);
  }
  if(out.materials.empty()) out.materials.push_back(GltfMaterial{});

  // ---- meshes: one GltfPrimMesh per triangle primitive --------------------------------------------------------------
  std::vector<std::vector<int>> meshPrims(A("meshes").size());
  bool haveNormals = true, haveTangents = true;
  for(size_t mi = 0; mi < A("meshes").size(); mi++) {
    const Json* prims = A("meshes").at(mi).get("primitives");
    if(!prims) continue;
    for(size_t pi = 0; pi < prims->size(); pi++) {
      const Json& pr = prims->at(pi);
      if(pr.integer("mode", 4) != 4) continue;  // triangles only (nvh::GltfScene skips the rest as well)
      const Json* attr = pr.get("attributes");
      Accessor pos;
      if(!attr || !accessor(attr->integer("POSITION", -1), pos) || pos.ncomp != 3 || pos.comp != 5126) continue;
      GltfPrimMesh pm;
      pm.vertexOffset = uint32_t(out.positions.size()); pm.vertexCount = uint32_t(pos.count);
      pm.firstIndex = uint32_t(out.indices.size());
      pm.materialIndex = std::max(0, pr.integer("material", 0));
      if(size_t(pm.materialIndex) >= out.materials.size()) pm.materialIndex = 0;
      for(size_t v = 0; v < pos.count; v++) out.positions.push_back(V3{readComp(pos, v, 0), readComp(pos, v, 1), readComp(pos, v, 2)});
      Accessor nrm, tng, uv, col;
      const bool hn = accessor(attr->integer("NORMAL", -1), nrm) && nrm.count == pos.count && nrm.ncomp >= 3;
      const bool ht = accessor(attr->integer("TANGENT", -1), tng) && tng.count == pos.count && tng.ncomp == 4;
      const bool hu = accessor(attr->integer("TEXCOORD_0", -1), uv) && uv.count == pos.count && uv.ncomp >= 2;
      const bool hc = accessor(attr->integer("COLOR_0", -1), col) && col.count == pos.count && (col.ncomp == 3 || col.ncomp == 4);
      haveNormals = haveNormals && hn; haveTangents = haveTangents && ht;
      for(size_t v = 0; v < pos.count; v++) {
        out.normals.push_back(hn ? V3{readComp(nrm, v, 0), readComp(nrm, v, 1), readComp(nrm, v, 2)} : V3{0, 0, 0});
        out.tangents.push_back(ht ? std::array<float, 4>{readComp(tng, v, 0), readComp(tng, v, 1), readComp(tng, v, 2), readComp(tng, v, 3)} : std::array<float, 4>{0, 0, 0, 0});
        out.texcoords0.push_back(hu ? std::array<float, 2>{readComp(uv, v, 0), readComp(uv, v, 1)} : std::array<float, 2>{0, 0});
        out.colors0.push_back(hc ? std::array<float, 4>{readComp(col, v, 0), readComp(col, v, 1), readComp(col, v, 2), col.ncomp == 4 ? readComp(col, v, 3) : 1.f}
                                 : std::array<float, 4>{1, 1, 1, 1});
      }
      Accessor idx;
      if(accessor(pr.integer("indices", -1), idx)) { for(size_t k = 0; k + 3 <= idx.count; k += 3) for(int c = 0; c < 3; c++) out.indices.push_back(readIndex(idx, k + size_t(c))); }
      else for(uint32_t k = 0; k + 3 <= pm.vertexCount; k += 3) { out.indices.push_back(k); out.indices.push_back(k + 1); out.indices.push_back(k + 2); }
      pm.indexCount = uint32_t(out.indices.size()) - pm.firstIndex;
      for(uint32_t k = pm.firstIndex; k < pm.firstIndex + pm.indexCount; k++) if(out.indices[k] >= pm.vertexCount) { error = "index out of range in mesh " + std::to_string(mi); return false; }
      if(pm.indexCount == 0) { out.positions.resize(pm.vertexOffset); out.normals.resize(pm.vertexOffset); out.tangents.resize(pm.vertexOffset); out.texcoords0.resize(pm.vertexOffset); out.colors0.resize(pm.vertexOffset); continue; }
      meshPrims[mi].push_back(int(out.primMeshes.size()));
      out.primMeshes.push_back(pm);
    }
  }
  // per-vertex synthesis of missing NORMAL / TANGENT happens for whole attribute arrays in Scene::loadFromGltfScene; a file
  // that mixes primitives with and without them gets them synthesised for every primitive (documented in DESIGN.md §Scene ingest)
  if(!haveNormals) out.normals.clear();
  if(!haveTangents) out.tangents.clear();

  // ---- scene graph -> drawable nodes, cameras, lights (nvh::GltfScene::importDrawableNodes) -------------------------------
  const Json* lightDefs = nullptr;
  if(const Json* ext = root.get("extensions")) if(const Json* kl = ext->get("KHR_lights_punctual")) lightDefs = kl->get("lights");
  const Json& nodes = A("nodes");
  std::vector<int> roots;
  const Json& scenes = A("scenes");
  const int sceneIdx = root.integer("scene", 0);
  if(scenes.size() && size_t(sceneIdx) < scenes.size()) { if(const Json* rn = scenes.at(size_t(sceneIdx)).get("nodes")) for(size_t i = 0; i < rn->size(); i++) roots.push_back(int(rn->at(i).num)); }
  else { std::vector<bool> child(nodes.size(), false); for(size_t i = 0; i < nodes.size(); i++) if(const Json* c = nodes.at(i).get("children")) for(size_t k = 0; k < c->size(); k++) { int ci = int(c->at(k).num); if(ci >= 0 && size_t(ci) < child.size()) child[size_t(ci)] = true; } for(size_t i = 0; i < nodes.size(); i++) if(!child[i]) roots.push_back(int(i)); }
  struct Item { int node; M4 parent; int depth; };
  std::vector<Item> stack;
  for(auto it = roots.rbegin(); it != roots.rend(); ++it) stack.push_back({*it, M4::identity(), 0});
  while(!stack.empty()) {
    const Item it = stack.back(); stack.pop_back();
    if(it.node < 0 || size_t(it.node) >= nodes.size() || it.depth > 256) continue;
    const Json& n = nodes.at(size_t(it.node));
    const M4 world = it.parent * nodeLocalMatrix(n);
    const int mesh = n.integer("mesh", -1);
    if(mesh >= 0 && size_t(mesh) < meshPrims.size()) for(int pm : meshPrims[size_t(mesh)]) { GltfNode gn; gn.worldMatrix = world; gn.primMesh = pm; out.nodes.push_back(gn); }
    const int cam = n.integer("camera", -1);
    if(cam >= 0 && size_t(cam) < A("cameras").size()) {
      const Json& c = A("cameras").at(size_t(cam));
      if(const Json* p = c.get("perspective")) {
        GltfCamera gc;
        gc.eye = xformPoint(world, V3{0, 0, 0}); gc.center = xformPoint(world, V3{0, 0, -1}); gc.up = normalize(xformDir(world, V3{0, 1, 0}));
        gc.yfovDeg = float(p->number("yfov", 0.785398)) * 180.f / 3.14159265f;
        out.cameras.push_back(gc);
      }
    }
    if(lightDefs) if(const Json* ext = n.get("extensions")) if(const Json* kl = ext->get("KHR_lights_punctual")) {
      const int li = kl->integer("light", -1);
      if(li >= 0 && size_t(li) < lightDefs->size()) {
        const Json& l = lightDefs->at(size_t(li));
        GltfLight gl; gl.worldMatrix = world;
        if(const Json* c = l.get("color")) for(int k = 0; k < 3 && k < int(c->size()); k++) gl.color[k] = float(c->at(size_t(k)).num);
        gl.intensity = float(l.number("intensity", 1.0)); gl.range = float(l.number("range", 0.0));
        const std::string ty = l.string("type", "point");
        gl.type = ty == "directional" ? 0 : ty == "spot" ? 2 : 1;  // host_device.h:252-254
        if(const Json* sp = l.get("spot")) { gl.innerConeAngle = float(sp->number("innerConeAngle", 0.0)); gl.outerConeAngle = float(sp->number("outerConeAngle", 0.785398)); }
        out.lights.push_back(gl);
      }
    }
    if(const Json* c = n.get("children")) for(size_t k = c->size(); k-- > 0;) stack.push_back({int(c->at(k).num), world, it.depth + 1});
  }
  if(out.primMeshes.empty() || out.nodes.empty()) { error = "no drawable triangle meshes in " + filename; return false; }
  return true;
}

}  // namespace rth

// ---------------------------------------------------------------------------------------------------------------------
// Writer: serialises a GltfScene back to a self-contained .gltf (base64 buffers, PNG images).  Not part of the reference;
// it exists so that the procedural stand-in scenes can be exchanged with other glTF tools and so that the reader above
// is exercised on every attribute/extension it understands (tests/test_gltf.py round-trips through it).
namespace rth {
namespace {
std::string b64(const uint8_t* d, size_t n)
{
  static const char* T = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789+/";
  std::string o; o.reserve((n + 2) / 3 * 4);
  for(size_t i = 0; i < n; i += 3) {
    unsigned v = unsigned(d[i]) << 16 | (i + 1 < n ? unsigned(d[i + 1]) << 8 : 0) | (i + 2 < n ? d[i + 2] : 0);
    o += T[(v >> 18) & 63]; o += T[(v >> 12) & 63]; o += i + 1 < n ? T[(v >> 6) & 63] : '='; o += i + 2 < n ? T[v & 63] : '=';
  }
  return o;
}
std::vector<uint8_t> encodePng(const TextureImage& t)
{
  std::vector<uint8_t> raw; raw.reserve(size_t(t.height) * (size_t(t.width) * 4 + 1));
  for(int y = 0; y < t.height; y++) {
    raw.push_back(0);
    for(int x = 0; x < t.width; x++) { const uint8_t* p = &t.bgra[(size_t(y) * t.width + x) * 4]; raw.push_back(p[2]); raw.push_back(p[1]); raw.push_back(p[0]); raw.push_back(p[3]); }
  }
  uLongf clen = compressBound(uLong(raw.size()));
  std::vector<uint8_t> z(clen);
  compress2(z.data(), &clen, raw.data(), uLong(raw.size()), raw.size() > (size_t(1) << 22) ? 1 : 6);   // (asset-sized images: speed over ratio)
  z.resize(clen);
  std::vector<uint8_t> out = {137, 80, 78, 71, 13, 10, 26, 10};
  auto chunk = [&](const char* tag, const std::vector<uint8_t>& body) {
    auto be = [&](uint32_t v) { out.push_back(uint8_t(v >> 24)); out.push_back(uint8_t(v >> 16)); out.push_back(uint8_t(v >> 8)); out.push_back(uint8_t(v)); };
    be(uint32_t(body.size()));
    const size_t s = out.size();
    out.insert(out.end(), tag, tag + 4); out.insert(out.end(), body.begin(), body.end());
    be(uint32_t(crc32(0L, out.data() + s, uInt(out.size() - s))));
  };
  std::vector<uint8_t> ihdr(13, 0);
  ihdr[0] = uint8_t(t.width >> 24); ihdr[1] = uint8_t(t.width >> 16); ihdr[2] = uint8_t(t.width >> 8); ihdr[3] = uint8_t(t.width);
  ihdr[4] = uint8_t(t.height >> 24); ihdr[5] = uint8_t(t.height >> 16); ihdr[6] = uint8_t(t.height >> 8); ihdr[7] = uint8_t(t.height);
  ihdr[8] = 8; ihdr[9] = 6;
  chunk("IHDR", ihdr); chunk("IDAT", z); chunk("IEND", {});
  return out;
}
std::string num(float f) { char b[40]; snprintf(b, sizeof b, "%.9g", double(f)); return b; }
}  // namespace

bool saveGltfFile(const std::string& filename, const GltfScene& g, std::string& error)
{
  const size_t nv = g.positions.size();
  std::vector<uint8_t> bin;
  struct View { size_t off, len; };
  auto append = [&](const void* p, size_t n) { while(bin.size() % 4) bin.push_back(0); View v{bin.size(), n}; const uint8_t* b = static_cast<const uint8_t*>(p); bin.insert(bin.end(), b, b + n); return v; };
  const bool hn = g.normals.size() == nv, ht = g.tangents.size() == nv, hu = g.texcoords0.size() == nv, hc = g.colors0.size() == nv;
  std::vector<View> views;
  views.push_back(append(g.positions.data(), nv * 12));
  const int vN = hn ? (views.push_back(append(g.normals.data(), nv * 12)), int(views.size()) - 1) : -1;
  const int vT = ht ? (views.push_back(append(g.tangents.data(), nv * 16)), int(views.size()) - 1) : -1;
  const int vU = hu ? (views.push_back(append(g.texcoords0.data(), nv * 8)), int(views.size()) - 1) : -1;
  const int vC = hc ? (views.push_back(append(g.colors0.data(), nv * 16)), int(views.size()) - 1) : -1;
  views.push_back(append(g.indices.data(), g.indices.size() * 4));
  const int vI = int(views.size()) - 1;

  std::ostringstream o;
  o << "{\"asset\":{\"version\":\"2.0\",\"generator\":\"restir_amd host\"},\n";
  o << "\"extensionsUsed\":[\"KHR_lights_punctual\",\"KHR_materials_transmission\",\"KHR_materials_ior\"],\n";
  // accessors: per prim-mesh one accessor per attribute + one for the indices
  std::ostringstream acc, meshes;
  int nAcc = 0;
  auto addAcc = [&](int view, size_t byteOff, size_t count, int comp, const char* type, const V3* mn, const V3* mx) {
    if(nAcc) acc << ",\n";
    acc << "{\"bufferView\":" << view << ",\"byteOffset\":" << byteOff << ",\"componentType\":" << comp << ",\"count\":" << count << ",\"type\":\"" << type << "\"";
    if(mn) acc << ",\"min\":[" << num(mn->x) << "," << num(mn->y) << "," << num(mn->z) << "],\"max\":[" << num(mx->x) << "," << num(mx->y) << "," << num(mx->z) << "]";
    acc << "}";
    return nAcc++;
  };
  for(size_t i = 0; i < g.primMeshes.size(); i++) {
    const GltfPrimMesh& pm = g.primMeshes[i];
    V3 mn{3e38f, 3e38f, 3e38f}, mx{-3e38f, -3e38f, -3e38f};
    for(uint32_t v = 0; v < pm.vertexCount; v++) { V3 p = g.positions[pm.vertexOffset + v]; mn = {std::min(mn.x, p.x), std::min(mn.y, p.y), std::min(mn.z, p.z)}; mx = {std::max(mx.x, p.x), std::max(mx.y, p.y), std::max(mx.z, p.z)}; }
    if(i) meshes << ",\n";
    meshes << "{\"primitives\":[{\"attributes\":{\"POSITION\":" << addAcc(0, size_t(pm.vertexOffset) * 12, pm.vertexCount, 5126, "VEC3", &mn, &mx);
    if(hn) meshes << ",\"NORMAL\":" << addAcc(vN, size_t(pm.vertexOffset) * 12, pm.vertexCount, 5126, "VEC3", nullptr, nullptr);
    if(ht) meshes << ",\"TANGENT\":" << addAcc(vT, size_t(pm.vertexOffset) * 16, pm.vertexCount, 5126, "VEC4", nullptr, nullptr);
    if(hu) meshes << ",\"TEXCOORD_0\":" << addAcc(vU, size_t(pm.vertexOffset) * 8, pm.vertexCount, 5126, "VEC2", nullptr, nullptr);
    if(hc) meshes << ",\"COLOR_0\":" << addAcc(vC, size_t(pm.vertexOffset) * 16, pm.vertexCount, 5126, "VEC4", nullptr, nullptr);
    meshes << "},\"indices\":" << addAcc(vI, size_t(pm.firstIndex) * 4, pm.indexCount, 5125, "SCALAR", nullptr, nullptr);
    meshes << ",\"material\":" << pm.materialIndex << ",\"mode\":4}]}";
  }
  // Small scenes are written self-contained (one data: URI; images inside the binary buffer).  An asset-sized scene (round 6: the benchmark's 2.8 M triangles and
  // 3.4 GB of texels, bench.py --via-gltf) is written the way such assets ship: `<name>.bin` for the geometry and one `<name>_img<i>.png` per image beside the .gltf —
  // a base64 buffer of that size would be a 5 GB JSON string, and a .glb chunk cannot exceed 4 GB.  RESTIR_GLTF_EXTERNAL=0 / 1 forces either form.
  size_t texelBytes = 0;
  for(const TextureImage& t : g.textures) texelBytes += t.bgra.size();
  const char* extEnv = getenv("RESTIR_GLTF_EXTERNAL");
  const bool external = extEnv ? atoi(extEnv) != 0 : (bin.size() + texelBytes > (size_t(256) << 20));
  const size_t slash = filename.find_last_of("/\\");
  const std::string dirOut = slash == std::string::npos ? "" : filename.substr(0, slash + 1);
  std::string stem = slash == std::string::npos ? filename : filename.substr(slash + 1);
  if(stem.size() > 5 && stem.substr(stem.size() - 5) == ".gltf") stem.resize(stem.size() - 5);
  std::vector<int> imgView;
  if(!external) {
    for(const TextureImage& t : g.textures) { std::vector<uint8_t> png = encodePng(t); views.push_back(append(png.data(), png.size())); imgView.push_back(int(views.size()) - 1); }
    o << "\"buffers\":[{\"byteLength\":" << bin.size() << ",\"uri\":\"data:application/octet-stream;base64," << b64(bin.data(), bin.size()) << "\"}],\n";
  } else {
    std::atomic<size_t> nextImage{0};
    std::atomic<bool> failed{false};
    auto work = [&] {
      for(;;) {
        const size_t i = nextImage.fetch_add(1);
        if(i >= g.textures.size()) return;
        const std::vector<uint8_t> png = encodePng(g.textures[i]);
        std::ofstream pf(dirOut + stem + "_img" + std::to_string(i) + ".png", std::ios::binary);
        pf.write(reinterpret_cast<const char*>(png.data()), std::streamsize(png.size()));
        if(!pf) failed = true;
      }
    };
    const size_t nt = std::min<size_t>(std::max<size_t>(1, g.textures.size()), size_t(rt_cpu_budget()));
    std::vector<std::thread> pool;
    for(size_t k = 1; k < nt; k++) pool.emplace_back(work);
    work();
    for(auto& th : pool) th.join();
    std::ofstream bf(dirOut + stem + ".bin", std::ios::binary);
    bf.write(reinterpret_cast<const char*>(bin.data()), std::streamsize(bin.size()));
    if(!bf || failed) { error = "cannot write the external files of " + filename; return false; }
    o << "\"buffers\":[{\"byteLength\":" << bin.size() << ",\"uri\":\"" << stem << ".bin\"}],\n";
  }
  o << "\"bufferViews\":[";
  for(size_t i = 0; i < views.size(); i++) o << (i ? "," : "") << "{\"buffer\":0,\"byteOffset\":" << views[i].off << ",\"byteLength\":" << views[i].len << "}";
  o << "],\n\"accessors\":[" << acc.str() << "],\n\"meshes\":[" << meshes.str() << "],\n";
  if(!g.textures.empty()) {
    o << "\"images\":[";
    if(external) for(size_t i = 0; i < g.textures.size(); i++) o << (i ? "," : "") << "{\"uri\":\"" << stem << "_img" << i << ".png\"}";
    else for(size_t i = 0; i < imgView.size(); i++) o << (i ? "," : "") << "{\"bufferView\":" << imgView[i] << ",\"mimeType\":\"image/png\"}";
    o << "],\n\"samplers\":[";
    for(size_t i = 0; i < g.textures.size(); i++) o << (i ? "," : "") << "{\"wrapS\":" << g.textures[i].wrapS << ",\"wrapT\":" << g.textures[i].wrapT << ",\"magFilter\":" << g.textures[i].magFilter << "}";
    o << "],\n\"textures\":[";
    for(size_t i = 0; i < g.textures.size(); i++) o << (i ? "," : "") << "{\"source\":" << i << ",\"sampler\":" << i << "}";
    o << "],\n";
  }
  o << "\"materials\":[";
  for(size_t i = 0; i < g.materials.size(); i++) {
    const GltfMaterial& m = g.materials[i];
    o << (i ? ",\n" : "") << "{\"pbrMetallicRoughness\":{\"baseColorFactor\":[" << num(m.baseColorFactor[0]) << "," << num(m.baseColorFactor[1]) << "," << num(m.baseColorFactor[2]) << "," << num(m.baseColorFactor[3])
      << "],\"metallicFactor\":" << num(m.metallicFactor) << ",\"roughnessFactor\":" << num(m.roughnessFactor);
    if(m.baseColorTexture >= 0) o << ",\"baseColorTexture\":{\"index\":" << m.baseColorTexture << "}";
    if(m.metallicRoughnessTexture >= 0) o << ",\"metallicRoughnessTexture\":{\"index\":" << m.metallicRoughnessTexture << "}";
    o << "},\"emissiveFactor\":[" << num(m.emissiveFactor[0]) << "," << num(m.emissiveFactor[1]) << "," << num(m.emissiveFactor[2]) << "]";
    if(m.emissiveTexture >= 0) o << ",\"emissiveTexture\":{\"index\":" << m.emissiveTexture << "}";
    if(m.normalTexture >= 0) o << ",\"normalTexture\":{\"index\":" << m.normalTexture << ",\"scale\":" << num(m.normalTextureScale) << "}";
    o << ",\"alphaMode\":\"" << (m.alphaMode == RT_ALPHA_MASK ? "MASK" : m.alphaMode == RT_ALPHA_BLEND ? "BLEND" : "OPAQUE") << "\",\"alphaCutoff\":" << num(m.alphaCutoff)
      << ",\"doubleSided\":" << (m.doubleSided ? "true" : "false");
    // emissive factors above 1 are legal for this loader (the strength extension multiplies into them on load), so they are
    // written as they are rather than split into factor x strength
    o << ",\"extensions\":{\"KHR_materials_transmission\":{\"transmissionFactor\":" << num(m.transmissionFactor);
    if(m.transmissionTexture >= 0) o << ",\"transmissionTexture\":{\"index\":" << m.transmissionTexture << "}";
    o << "},\"KHR_materials_ior\":{\"ior\":" << num(m.ior) << "}}}";
  }
  o << "],\n";
  // nodes: drawables, then lights, then cameras — all at the root with world matrices
  std::ostringstream nodes; int nNodes = 0;
  auto mat = [&](const M4& m) { std::string s = "["; for(int i = 0; i < 16; i++) { s += (i ? "," : ""); s += num(m.m[i]); } return s + "]"; };
  for(const GltfNode& n : g.nodes) nodes << (nNodes++ ? ",\n" : "") << "{\"mesh\":" << n.primMesh << ",\"matrix\":" << mat(n.worldMatrix) << "}";
  for(size_t i = 0; i < g.lights.size(); i++) nodes << (nNodes++ ? ",\n" : "") << "{\"matrix\":" << mat(g.lights[i].worldMatrix) << ",\"extensions\":{\"KHR_lights_punctual\":{\"light\":" << i << "}}}";
  for(size_t i = 0; i < g.cameras.size(); i++) {
    const GltfCamera& c = g.cameras[i];
    const V3 f = normalize(c.center - c.eye), r = normalize(cross(f, c.up)), u = cross(r, f);
    M4 m = M4::identity();
    m.at(0, 0) = r.x; m.at(1, 0) = r.y; m.at(2, 0) = r.z; m.at(0, 1) = u.x; m.at(1, 1) = u.y; m.at(2, 1) = u.z;
    m.at(0, 2) = -f.x; m.at(1, 2) = -f.y; m.at(2, 2) = -f.z; m.at(0, 3) = c.eye.x; m.at(1, 3) = c.eye.y; m.at(2, 3) = c.eye.z;
    nodes << (nNodes++ ? ",\n" : "") << "{\"camera\":" << i << ",\"matrix\":" << mat(m) << "}";
  }
  o << "\"nodes\":[" << nodes.str() << "],\n\"scenes\":[{\"nodes\":[";
  for(int i = 0; i < nNodes; i++) o << (i ? "," : "") << i;
  o << "]}],\"scene\":0";
  if(!g.cameras.empty()) {
    o << ",\n\"cameras\":[";
    for(size_t i = 0; i < g.cameras.size(); i++) o << (i ? "," : "") << "{\"type\":\"perspective\",\"perspective\":{\"yfov\":" << num(g.cameras[i].yfovDeg * 3.14159265f / 180.f) << ",\"znear\":0.01}}";
    o << "]";
  }
  if(!g.lights.empty()) {
    o << ",\n\"extensions\":{\"KHR_lights_punctual\":{\"lights\":[";
    for(size_t i = 0; i < g.lights.size(); i++) {
      const GltfLight& l = g.lights[i];
      o << (i ? "," : "") << "{\"type\":\"" << (l.type == 0 ? "directional" : l.type == 2 ? "spot" : "point") << "\",\"color\":[" << num(l.color[0]) << "," << num(l.color[1]) << "," << num(l.color[2])
        << "],\"intensity\":" << num(l.intensity);
      if(l.range > 0) o << ",\"range\":" << num(l.range);
      if(l.type == 2) o << ",\"spot\":{\"innerConeAngle\":" << num(l.innerConeAngle) << ",\"outerConeAngle\":" << num(l.outerConeAngle) << "}";
      o << "}";
    }
    o << "]}}";
  }
  o << "}\n";
  std::ofstream f(filename, std::ios::binary);
  if(!f) { error = "cannot write " + filename; return false; }
  const std::string s = o.str();
  f.write(s.data(), std::streamsize(s.size()));
  return bool(f);
}
}  // namespace rth
