// png_writer.cpp — 8-bit RGB / RGBA PNG writer for the displayed frame (RenderOutput::run's image; the reference saves its
// screenshots through stb_image_write, src/sample_example.cpp "save image").  One IDAT chunk, filter type 0 on every scanline,
// zlib deflate + CRC-32 from zlib (already a dependency of the glTF loader's PNG reader, which reads these files back).
#include "scene.hpp"
#include <zlib.h>
#include <cstdio>

namespace rth {

static void put32(std::vector<uint8_t>& v, uint32_t x) { v.push_back(uint8_t(x >> 24)); v.push_back(uint8_t(x >> 16)); v.push_back(uint8_t(x >> 8)); v.push_back(uint8_t(x)); }
static void chunk(std::vector<uint8_t>& out, const char* type, const std::vector<uint8_t>& data)
{
  put32(out, uint32_t(data.size()));
  const size_t start = out.size();
  out.insert(out.end(), type, type + 4);
  out.insert(out.end(), data.begin(), data.end());
  put32(out, uint32_t(crc32(0L, out.data() + start, uInt(out.size() - start))));
}

// rgba: width*height*4 bytes, R first (RT_BUF_LDR layout).  keepAlpha = false writes colour type 2 (RGB).
bool encodePng(const uint8_t* rgba, int width, int height, bool keepAlpha, std::vector<uint8_t>& out)
{
  if(!rgba || width <= 0 || height <= 0) return false;
  const int ch = keepAlpha ? 4 : 3;
  std::vector<uint8_t> raw(size_t(height) * (1 + size_t(width) * ch));
  size_t o = 0;
  for(int y = 0; y < height; y++) {
    raw[o++] = 0;  // filter: none
    for(int x = 0; x < width; x++) { const uint8_t* p = rgba + (size_t(y) * width + x) * 4; for(int k = 0; k < ch; k++) raw[o++] = p[k]; }
  }
  uLongf bound = compressBound(uLong(raw.size()));
  std::vector<uint8_t> z(bound);
  if(compress2(z.data(), &bound, raw.data(), uLong(raw.size()), 6) != Z_OK) return false;
  z.resize(bound);
  out.clear();
  const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
  out.insert(out.end(), sig, sig + 8);
  std::vector<uint8_t> ihdr;
  put32(ihdr, uint32_t(width)); put32(ihdr, uint32_t(height));
  ihdr.push_back(8); ihdr.push_back(keepAlpha ? 6 : 2); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
  chunk(out, "IHDR", ihdr);
  chunk(out, "IDAT", z);
  chunk(out, "IEND", {});
  return true;
}

bool writePng(const std::string& path, const uint8_t* rgba, int width, int height, bool keepAlpha)
{
  std::vector<uint8_t> bytes;
  if(!encodePng(rgba, width, height, keepAlpha, bytes)) return false;
  FILE* f = fopen(path.c_str(), "wb");
  if(!f) return false;
  const bool ok = fwrite(bytes.data(), 1, bytes.size(), f) == bytes.size();
  fclose(f);
  return ok;
}

}  // namespace rth
