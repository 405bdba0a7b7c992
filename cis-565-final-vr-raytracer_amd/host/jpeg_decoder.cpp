// jpeg_decoder.cpp — JPEG (ITU-T T.81) decoder for glTF texture images: baseline / extended sequential (SOF0, SOF1) and
// progressive (SOF2) Huffman DCT, 8-bit samples, 1 (grey) or 3 (YCbCr, JFIF) components, any sampling factors up to 4x4,
// restart intervals.  Arithmetic coding, lossless and 12-bit JPEGs are rejected (the caller substitutes the white texel).
//
// The reference decodes images with stb_image / FreeImage through nvpro_core (scene.cpp:554-646) — third-party code that is
// not vendored — so this is a from-scratch reader.  Choices the standard leaves to the decoder: the inverse DCT is the
// separable floating-point definition (double), chroma is upsampled by pixel replication, YCbCr -> RGB uses the JFIF
// matrix with rounding.  tests/test_gltf.py compares it with libjpeg (through PIL) on smooth images.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>
#include "scene.hpp"

namespace rth {
namespace {

struct Huff {
  // canonical Huffman table (T.81 Annex C / F.2.2.3): for each code length the first code, the index of its first symbol
  int mincode[17], maxcode[18], valptr[17];
  uint8_t vals[256];
  bool present = false;
  void build(const uint8_t counts[16], const uint8_t* symbols)
  {
    int code = 0, k = 0;
    for(int l = 1; l <= 16; l++) {
      valptr[l] = k;
      mincode[l] = code;
      code += counts[l - 1];
      k += counts[l - 1];
      maxcode[l] = counts[l - 1] ? code - 1 : -1;
      code <<= 1;
    }
    maxcode[17] = 0x7fffffff;
    memcpy(vals, symbols, size_t(k));
    present = true;
  }
};

struct Component {
  int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
  int blocksW = 0, blocksH = 0;      // allocated size in blocks (padded to whole MCUs)
  std::vector<int16_t> coef;         // blocksW * blocksH * 64, natural (de-zigzagged) order
  int pred = 0;
};

const uint8_t kZigzag[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                             35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct BitReader {
  const uint8_t* p; const uint8_t* end;
  uint32_t acc = 0; int bits = 0;
  bool hitMarker = false;
  void reset() { acc = 0; bits = 0; hitMarker = false; }
  void fill()
  {
    while(bits <= 24) {
      int b = 0;
      if(!hitMarker && p < end) {
        b = *p;
        if(b == 0xFF) {
          if(p + 1 < end && p[1] == 0x00) p += 2;        // stuffed zero
          else { hitMarker = true; b = 0; }              // a marker: feed zeros until the caller handles it
        } else p++;
      }
      acc |= uint32_t(b) << (24 - bits);
      bits += 8;
    }
  }
  int get(int n)
  {
    if(n == 0) return 0;
    fill();
    const int v = int(acc >> (32 - n));
    acc <<= n; bits -= n;
    return v;
  }
  int bit() { return get(1); }
  int decode(const Huff& h)
  {
    fill();
    int code = 0;
    for(int l = 1; l <= 16; l++) {
      code = (code << 1) | int(acc >> 31);
      acc <<= 1; bits--;
      if(h.maxcode[l] >= 0 && code <= h.maxcode[l] && code >= h.mincode[l]) return h.vals[h.valptr[l] + code - h.mincode[l]];
      if(bits <= 0) fill();
    }
    return -1;
  }
  static int extend(int v, int t) { return (t && v < (1 << (t - 1))) ? v - (1 << t) + 1 : v; }  // F.2.2.1
};

struct Decoder {
  const uint8_t* data; size_t size;
  int width = 0, height = 0, ncomp = 0, hmax = 1, vmax = 1, mcuW = 0, mcuH = 0, mcusX = 0, mcusY = 0;
  bool progressive = false;
  uint16_t qt[4][64];
  bool qtPresent[4] = {false, false, false, false};
  Huff dc[4], ac[4];
  Component comp[3];
  int restartInterval = 0;
  int eobrun = 0;
  bool adobe = false; int adobeTransform = -1;

  static int be16(const uint8_t* p) { return (p[0] << 8) | p[1]; }

  bool parse()
  {
    if(size < 4 || data[0] != 0xFF || data[1] != 0xD8) return false;
    size_t pos = 2;
    bool haveFrame = false;
    while(pos + 4 <= size) {
      if(data[pos] != 0xFF) { pos++; continue; }
      const int m = data[pos + 1];
      if(m == 0xFF) { pos++; continue; }
      pos += 2;
      if(m == 0xD8 || (m >= 0xD0 && m <= 0xD7) || m == 0x01) continue;
      if(m == 0xD9) break;
      if(pos + 2 > size) return false;
      const int len = be16(data + pos);
      if(len < 2 || pos + size_t(len) > size) return false;
      const uint8_t* seg = data + pos + 2;
      const int n = len - 2;
      switch(m) {
        case 0xDB: {  // DQT
          int o = 0;
          while(o < n) {
            const int pq = seg[o] >> 4, tq = seg[o] & 15; o++;
            if(tq > 3 || o + (pq ? 128 : 64) > n) return false;
            for(int i = 0; i < 64; i++) { qt[tq][kZigzag[i]] = uint16_t(pq ? be16(seg + o + 2 * i) : seg[o + i]); }
            o += pq ? 128 : 64;
            qtPresent[tq] = true;
          }
          break;
        }
        case 0xC4: {  // DHT
          int o = 0;
          while(o + 17 <= n) {
            const int tc = seg[o] >> 4, th = seg[o] & 15; o++;
            if(th > 3 || tc > 1) return false;
            int total = 0;
            for(int i = 0; i < 16; i++) total += seg[o + i];
            if(total > 256 || o + 16 + total > n) return false;
            (tc ? ac[th] : dc[th]).build(seg + o, seg + o + 16);
            o += 16 + total;
          }
          break;
        }
        case 0xC0: case 0xC1: case 0xC2: {  // SOF0/1/2
          if(haveFrame || n < 6) return false;
          progressive = (m == 0xC2);
          if(seg[0] != 8) return false;  // 8-bit only
          height = be16(seg + 1); width = be16(seg + 3); ncomp = seg[5];
          if(width <= 0 || height <= 0 || (ncomp != 1 && ncomp != 3) || n < 6 + 3 * ncomp) return false;
          if(size_t(width) * size_t(height) > (size_t(1) << 28)) return false;
          for(int i = 0; i < ncomp; i++) {
            comp[i].id = seg[6 + 3 * i]; comp[i].h = seg[7 + 3 * i] >> 4; comp[i].v = seg[7 + 3 * i] & 15; comp[i].tq = seg[8 + 3 * i];
            if(comp[i].h < 1 || comp[i].h > 4 || comp[i].v < 1 || comp[i].v > 4 || comp[i].tq > 3) return false;
            hmax = std::max(hmax, comp[i].h); vmax = std::max(vmax, comp[i].v);
          }
          mcuW = 8 * hmax; mcuH = 8 * vmax;
          mcusX = (width + mcuW - 1) / mcuW; mcusY = (height + mcuH - 1) / mcuH;
          for(int i = 0; i < ncomp; i++) {
            comp[i].blocksW = mcusX * comp[i].h; comp[i].blocksH = mcusY * comp[i].v;
            comp[i].coef.assign(size_t(comp[i].blocksW) * comp[i].blocksH * 64, 0);
          }
          haveFrame = true;
          break;
        }
        case 0xC3: case 0xC5: case 0xC6: case 0xC7: case 0xC9: case 0xCA: case 0xCB: case 0xCD: case 0xCE: case 0xCF: return false;  // lossless / hierarchical / arithmetic
        case 0xDD: if(n >= 2) restartInterval = be16(seg); break;
        case 0xEE: if(n >= 12 && !memcmp(seg, "Adobe", 5)) { adobe = true; adobeTransform = seg[11]; } break;
        case 0xDA: {  // SOS
          if(!haveFrame || n < 1) return false;
          const int ns = seg[0];
          if(ns < 1 || ns > ncomp || n < 1 + 2 * ns + 3) return false;
          int order[3];
          for(int i = 0; i < ns; i++) {
            int ci = -1;
            for(int k = 0; k < ncomp; k++) if(comp[k].id == seg[1 + 2 * i]) ci = k;
            if(ci < 0) return false;
            order[i] = ci;
            comp[ci].td = seg[2 + 2 * i] >> 4; comp[ci].ta = seg[2 + 2 * i] & 15;
            if(comp[ci].td > 3 || comp[ci].ta > 3) return false;
          }
          const int ss = seg[1 + 2 * ns], se = seg[2 + 2 * ns], ah = seg[3 + 2 * ns] >> 4, al = seg[3 + 2 * ns] & 15;
          size_t scanStart = pos + size_t(len);
          size_t consumed = 0;
          if(!decodeScan(order, ns, ss, se, ah, al, data + scanStart, data + size, consumed)) return false;
          pos = scanStart + consumed;
          continue;
        }
        default: break;
      }
      pos += size_t(len);
    }
    return haveFrame;
  }

  // ---- entropy-coded segment ---------------------------------------------------------------------------------------------
  bool decodeBlockBaseline(BitReader& br, Component& c, int16_t* blk)
  {
    const Huff& hd = dc[c.td]; const Huff& ha = ac[c.ta];
    if(!hd.present || !ha.present) return false;
    const int t = br.decode(hd);
    if(t < 0 || t > 11) return false;
    c.pred += BitReader::extend(br.get(t), t);
    blk[0] = int16_t(c.pred);
    for(int k = 1; k < 64;) {
      const int rs = br.decode(ha);
      if(rs < 0) return false;
      const int r = rs >> 4, s = rs & 15;
      if(s == 0) { if(r == 15) { k += 16; continue; } break; }
      k += r;
      if(k > 63) return false;
      blk[kZigzag[k]] = int16_t(BitReader::extend(br.get(s), s));
      k++;
    }
    return true;
  }
  bool decodeBlockDCFirst(BitReader& br, Component& c, int16_t* blk, int al)
  {
    const Huff& hd = dc[c.td];
    if(!hd.present) return false;
    const int t = br.decode(hd);
    if(t < 0 || t > 11) return false;
    c.pred += BitReader::extend(br.get(t), t);
    blk[0] = int16_t(c.pred * (1 << al));
    return true;
  }
  static void decodeBlockDCRefine(BitReader& br, int16_t* blk, int al) { if(br.bit()) blk[0] = int16_t(blk[0] | (1 << al)); }
  bool decodeBlockACFirst(BitReader& br, Component& c, int16_t* blk, int ss, int se, int al)
  {
    if(eobrun > 0) { eobrun--; return true; }
    const Huff& ha = ac[c.ta];
    if(!ha.present) return false;
    for(int k = ss; k <= se;) {
      const int rs = br.decode(ha);
      if(rs < 0) return false;
      const int r = rs >> 4, s = rs & 15;
      if(s == 0) {
        if(r < 15) { eobrun = (1 << r) - 1; if(r) eobrun += br.get(r); break; }
        k += 16;
        continue;
      }
      k += r;
      if(k > 63) return false;
      blk[kZigzag[k]] = int16_t(BitReader::extend(br.get(s), s) * (1 << al));
      k++;
    }
    return true;
  }
  bool decodeBlockACRefine(BitReader& br, Component& c, int16_t* blk, int ss, int se, int al)  // T.81 G.1.2.3
  {
    const int p1 = 1 << al, m1 = -1 * (1 << al);
    int k = ss;
    if(eobrun <= 0) {
      const Huff& ha = ac[c.ta];
      if(!ha.present) return false;
      for(; k <= se;) {
        const int rs = br.decode(ha);
        if(rs < 0) return false;
        int r = rs >> 4;
        const int s = rs & 15;
        int value = 0;
        if(s == 0) {
          if(r < 15) { eobrun = (1 << r); if(r) eobrun += br.get(r); break; }
        } else {
          if(s != 1) return false;
          value = br.bit() ? p1 : m1;
        }
        while(k <= se) {
          int16_t& z = blk[kZigzag[k]];
          if(z != 0) {
            if(br.bit()) { if((z & p1) == 0) z = int16_t(z >= 0 ? z + p1 : z + m1); }
          } else {
            if(r == 0) { if(value) z = int16_t(value); k++; break; }
            r--;
          }
          k++;
        }
      }
    }
    if(eobrun > 0) {
      for(; k <= se; k++) {
        int16_t& z = blk[kZigzag[k]];
        if(z != 0 && br.bit()) { if((z & p1) == 0) z = int16_t(z >= 0 ? z + p1 : z + m1); }
      }
      eobrun--;
    }
    return true;
  }

  bool decodeScan(const int* order, int ns, int ss, int se, int ah, int al, const uint8_t* p, const uint8_t* end, size_t& consumed)
  {
    if(!progressive) { if(ss != 0 || se != 63 || ah != 0 || al != 0) { ss = 0; se = 63; ah = al = 0; } }
    else if(ss > se || se > 63 || (ss == 0 && se != 0) || (ss > 0 && ns != 1)) return false;
    BitReader br{p, end};
    for(int i = 0; i < ncomp; i++) comp[i].pred = 0;
    eobrun = 0;
    auto block = [&](Component& c, int bx, int by) -> bool {
      if(bx >= c.blocksW || by >= c.blocksH) return false;
      int16_t* blk = &c.coef[(size_t(by) * c.blocksW + bx) * 64];
      if(!progressive) return decodeBlockBaseline(br, c, blk);
      if(ss == 0) { if(ah == 0) return decodeBlockDCFirst(br, c, blk, al); decodeBlockDCRefine(br, blk, al); return true; }
      return ah == 0 ? decodeBlockACFirst(br, c, blk, ss, se, al) : decodeBlockACRefine(br, c, blk, ss, se, al);
    };
    int total, perRow = 0;
    Component* single = nullptr;
    if(ns == 1) {  // non-interleaved: the component's own blocks, only those that cover the image (A.2.3)
      single = &comp[order[0]];
      perRow = (((width * single->h + hmax - 1) / hmax) + 7) / 8;
      const int rows = (((height * single->v + vmax - 1) / vmax) + 7) / 8;
      total = perRow * rows;
    } else total = mcusX * mcusY;
    int untilRestart = restartInterval;
    for(int m = 0; m < total; m++) {
      if(restartInterval && untilRestart == 0) {
        // byte-align, expect RSTn
        br.reset();
        const uint8_t* q = br.p;
        while(q + 1 < end && !(q[0] == 0xFF && q[1] >= 0xD0 && q[1] <= 0xD7)) { if(q[0] == 0xFF && q[1] != 0 && q[1] != 0xFF) break; q++; }
        if(q + 1 < end && q[0] == 0xFF && q[1] >= 0xD0 && q[1] <= 0xD7) q += 2;
        br.p = q;
        for(int i = 0; i < ncomp; i++) comp[i].pred = 0;
        eobrun = 0;
        untilRestart = restartInterval;
      }
      if(single) { if(!block(*single, m % perRow, m / perRow)) return false; }
      else {
        const int mx = m % mcusX, my = m / mcusX;
        for(int i = 0; i < ns; i++) {
          Component& c = comp[order[i]];
          for(int v = 0; v < c.v; v++) for(int h = 0; h < c.h; h++) if(!block(c, mx * c.h + h, my * c.v + v)) return false;
        }
      }
      if(restartInterval) untilRestart--;
    }
    // advance to the next marker
    const uint8_t* q = br.p;
    while(q + 1 < end && !(q[0] == 0xFF && q[1] != 0x00 && q[1] != 0xFF && !(q[1] >= 0xD0 && q[1] <= 0xD7))) q++;
    consumed = size_t(q - p);
    return true;
  }

  // ---- reconstruction ------------------------------------------------------------------------------------------------------
  void idctPlane(const Component& c, std::vector<uint8_t>& plane) const
  {
    static double C[8][8];
    static bool init = false;
    if(!init) {
      for(int x = 0; x < 8; x++) for(int u = 0; u < 8; u++) C[x][u] = (u == 0 ? std::sqrt(0.125) : 0.5) * std::cos((2 * x + 1) * u * 3.14159265358979323846 / 16.0);
      init = true;
    }
    const int pw = c.blocksW * 8, ph = c.blocksH * 8;
    plane.assign(size_t(pw) * ph, 0);
    const uint16_t* q = qt[c.tq];
    for(int by = 0; by < c.blocksH; by++)
      for(int bx = 0; bx < c.blocksW; bx++) {
        const int16_t* blk = &c.coef[(size_t(by) * c.blocksW + bx) * 64];
        double f[64], t[64];
        for(int i = 0; i < 64; i++) f[i] = double(blk[i]) * double(q[i]);
        for(int y = 0; y < 8; y++) for(int u = 0; u < 8; u++) { double s = 0; for(int v = 0; v < 8; v++) s += C[y][v] * f[v * 8 + u]; t[y * 8 + u] = s; }
        for(int y = 0; y < 8; y++) for(int x = 0; x < 8; x++) {
          double s = 0;
          for(int u = 0; u < 8; u++) s += C[x][u] * t[y * 8 + u];
          const double v = std::floor(s + 128.0 + 0.5);
          plane[size_t(by * 8 + y) * pw + bx * 8 + x] = uint8_t(v < 0 ? 0 : (v > 255 ? 255 : v));
        }
      }
  }

  bool toImage(TextureImage& img) const
  {
    for(int i = 0; i < ncomp; i++) if(!qtPresent[comp[i].tq]) return false;
    std::vector<uint8_t> planes[3];
    for(int i = 0; i < ncomp; i++) idctPlane(comp[i], planes[i]);
    img.width = width; img.height = height;
    img.bgra.resize(size_t(width) * height * 4);
    auto clamp8 = [](double v) { v = std::floor(v + 0.5); return uint8_t(v < 0 ? 0 : (v > 255 ? 255 : v)); };
    const bool ycc = ncomp == 3 && !(adobe && adobeTransform == 0);
    for(int y = 0; y < height; y++)
      for(int x = 0; x < width; x++) {
        uint8_t s[3] = {0, 0, 0};
        for(int i = 0; i < ncomp; i++) {
          const Component& c = comp[i];
          const int sx = x * c.h / hmax, sy = y * c.v / vmax;  // replication upsampling
          s[i] = planes[i][size_t(sy) * (c.blocksW * 8) + sx];
        }
        uint8_t r, g, b;
        if(ncomp == 1) r = g = b = s[0];
        else if(ycc) {
          const double Y = s[0], cb = double(s[1]) - 128.0, cr = double(s[2]) - 128.0;
          r = clamp8(Y + 1.402 * cr); g = clamp8(Y - 0.344136 * cb - 0.714136 * cr); b = clamp8(Y + 1.772 * cb);
        } else { r = s[0]; g = s[1]; b = s[2]; }
        uint8_t* o = &img.bgra[(size_t(y) * width + x) * 4];
        o[0] = b; o[1] = g; o[2] = r; o[3] = 255;
      }
    return true;
  }
};

}  // namespace

bool decodeJpeg(const uint8_t* d, size_t n, TextureImage& img)
{
  Decoder dec;
  dec.data = d; dec.size = n;
  memset(dec.qt, 0, sizeof(dec.qt));
  if(!dec.parse()) return false;
  return dec.toImage(img);
}

}  // namespace rth
