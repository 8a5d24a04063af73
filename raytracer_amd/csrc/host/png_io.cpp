// PNG / PPM reading to RGB8 and PNG writing, on top of the system zlib.
// Covers what the reference uses lodepng for: lodepng::decode(..., LCT_RGB) in
// TextureFile::LoadFile (reference RayTracer/texture.cpp:83-93, plus LoadPPM :43-64) and
// lodepng::encode(..., LCT_RGB / LCT_GREY, 8) in RenderImage::Save*Image (RayTracer/renderer.h:77-79).
#include <zlib.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <exception>
#include <vector>

#include "rthost.h"

namespace {

uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
void put32(std::vector<uint8_t>& v, uint32_t x) {
  v.push_back(x >> 24); v.push_back(x >> 16); v.push_back(x >> 8); v.push_back(x);
}
void chunk(std::vector<uint8_t>& out, const char* type, const uint8_t* data, size_t n) {
  put32(out, (uint32_t)n);
  size_t s = out.size();
  out.insert(out.end(), type, type + 4);
  if (n) out.insert(out.end(), data, data + n);
  put32(out, (uint32_t)crc32(0L, out.data() + s, (uInt)(n + 4)));
}
int paeth(int a, int b, int c) {
  int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
  return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

bool read_file(const char* path, std::vector<uint8_t>& d) {
  FILE* f = std::fopen(path, "rb");
  if (!f) return false;
  std::fseek(f, 0, SEEK_END);
  long n = std::ftell(f);
  std::fseek(f, 0, SEEK_SET);
  d.resize(n > 0 ? n : 0);
  size_t r = n > 0 ? std::fread(d.data(), 1, n, f) : 0;
  std::fclose(f);
  return r == d.size();
}

bool decode_png(const std::vector<uint8_t>& d, uint8_t** rgb, int* w, int* h) {
  static const uint8_t sig[8] = {137, 80, 78, 71, 13, 10, 26, 10};
  if (d.size() < 8 || std::memcmp(d.data(), sig, 8)) return false;
  size_t i = 8;
  uint32_t W = 0, H = 0;
  int depth = 0, ctype = 0, interlace = 0;
  std::vector<uint8_t> idat, plte, trns;
  while (i + 12 <= d.size()) {
    uint32_t n = be32(&d[i]);
    const uint8_t* t = &d[i + 4];
    if (i + 12 + n > d.size()) return false;
    const uint8_t* p = &d[i + 8];
    if (!std::memcmp(t, "IHDR", 4)) {
      if (n < 13) return false;   // truncated header
      W = be32(p); H = be32(p + 4); depth = p[8]; ctype = p[9]; interlace = p[12];
    } else if (!std::memcmp(t, "PLTE", 4)) plte.assign(p, p + n);
    else if (!std::memcmp(t, "IDAT", 4)) idat.insert(idat.end(), p, p + n);
    else if (!std::memcmp(t, "IEND", 4)) break;
    i += 12 + n;
  }
  if (!W || !H || interlace) return false;
  if ((uint64_t)W * (uint64_t)H > (1ull << 28)) return false;   // (a scene file can name any file: bound what it may allocate)
  int ch = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
  if (!ch) return false;
  // bit depths the PNG specification allows per colour type
  const bool depth_ok = ctype == 0 ? (depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16)
                      : ctype == 3 ? (depth == 1 || depth == 2 || depth == 4 || depth == 8)
                                   : (depth == 8 || depth == 16);
  if (!depth_ok) return false;
  size_t bpp_bits = (size_t)ch * depth;
  size_t stride = (W * bpp_bits + 7) / 8;
  size_t bpp = (bpp_bits + 7) / 8;
  std::vector<uint8_t> raw((stride + 1) * H);
  uLongf rl = (uLongf)raw.size();
  if (uncompress(raw.data(), &rl, idat.data(), (uLong)idat.size()) != Z_OK || rl != raw.size()) return false;
  std::vector<uint8_t> cur(stride), prev(stride, 0);
  uint8_t* out = (uint8_t*)std::malloc((size_t)W * H * 3);
  if (!out) return false;
  for (uint32_t y = 0; y < H; y++) {
    const uint8_t* row = &raw[(stride + 1) * y];
    int ft = row[0];
    for (size_t x = 0; x < stride; x++) {
      int a = x >= bpp ? cur[x - bpp] : 0, b = prev[x], c = x >= bpp ? prev[x - bpp] : 0;
      int v = row[1 + x];
      switch (ft) {
        case 1: v += a; break;
        case 2: v += b; break;
        case 3: v += (a + b) >> 1; break;
        case 4: v += paeth(a, b, c); break;
        default: break;
      }
      cur[x] = (uint8_t)v;
    }
    uint8_t* o = out + (size_t)y * W * 3;
    for (uint32_t x = 0; x < W; x++) {
      auto sample = [&](int k) -> int {  // k-th channel of pixel x as 8 bit
        if (depth == 8) return cur[(size_t)x * ch + k];
        if (depth == 16) return cur[((size_t)x * ch + k) * 2];
        size_t bit = ((size_t)x * ch + k) * depth;
        int v = (cur[bit >> 3] >> (8 - depth - (bit & 7))) & ((1 << depth) - 1);
        return ctype == 3 ? v : v * 255 / ((1 << depth) - 1);
      };
      if (ctype == 3) {
        int idx = sample(0);
        for (int k = 0; k < 3; k++) o[3 * x + k] = (size_t)(3 * idx + k) < plte.size() ? plte[3 * idx + k] : 0;
      } else if (ctype == 0 || ctype == 4) {
        int g = sample(0);
        o[3 * x] = o[3 * x + 1] = o[3 * x + 2] = (uint8_t)g;
      } else {
        for (int k = 0; k < 3; k++) o[3 * x + k] = (uint8_t)sample(k);
      }
    }
    prev.swap(cur);
  }
  *rgb = out; *w = (int)W; *h = (int)H;
  return true;
}

// P6 PPM as LoadPPM reads it (texture.cpp:43-64): header lines, comments, then raw RGB.
bool decode_ppm(const std::vector<uint8_t>& d, uint8_t** rgb, int* w, int* h) {
  size_t i = 0;
  auto line = [&]() {
    std::string s;
    while (i < d.size() && d[i] != '\n' && d[i] != '\r') s += (char)d[i++];
    if (i < d.size()) ++i;
    return s;
  };
  std::string l = line();
  if (l.size() < 2 || l[0] != 'P' || l[1] != '6') return false;
  l = line();
  while (!l.empty() && l[0] == '#') l = line();
  int W = 0, H = 0;
  if (std::sscanf(l.c_str(), "%d %d", &W, &H) != 2 || W <= 0 || H <= 0) return false;
  l = line();
  while (!l.empty() && l[0] == '#') l = line();
  size_t need = (size_t)W * H * 3;
  if (i + need > d.size()) return false;
  uint8_t* out = (uint8_t*)std::malloc(need);
  std::memcpy(out, &d[i], need);
  *rgb = out; *w = W; *h = H;
  return true;
}

}  // namespace

extern "C" int rth_write_png(const char* path, const uint8_t* px, int w, int h, int channels) {
  if (!path || !px || w <= 0 || h <= 0 || (channels != 1 && channels != 3)) return -1;
  std::vector<uint8_t> raw;
  size_t stride = (size_t)w * channels;
  raw.reserve((stride + 1) * h);
  for (int y = 0; y < h; y++) {
    raw.push_back(0);
    raw.insert(raw.end(), px + stride * y, px + stride * (y + 1));
  }
  uLongf cl = compressBound((uLong)raw.size());
  std::vector<uint8_t> comp(cl);
  if (compress2(comp.data(), &cl, raw.data(), (uLong)raw.size(), 6) != Z_OK) return -1;
  std::vector<uint8_t> out = {137, 80, 78, 71, 13, 10, 26, 10};
  uint8_t ihdr[13];
  ihdr[0] = w >> 24; ihdr[1] = w >> 16; ihdr[2] = w >> 8; ihdr[3] = w;
  ihdr[4] = h >> 24; ihdr[5] = h >> 16; ihdr[6] = h >> 8; ihdr[7] = h;
  ihdr[8] = 8; ihdr[9] = channels == 3 ? 2 : 0; ihdr[10] = ihdr[11] = ihdr[12] = 0;
  chunk(out, "IHDR", ihdr, 13);
  chunk(out, "IDAT", comp.data(), cl);
  chunk(out, "IEND", nullptr, 0);
  FILE* f = std::fopen(path, "wb");
  if (!f) return -1;
  size_t wr = std::fwrite(out.data(), 1, out.size(), f);
  std::fclose(f);
  return wr == out.size() ? 0 : -1;
}

extern "C" int rth_read_image_rgb(const char* path, uint8_t** rgb, int* w, int* h) {
  if (!path || !rgb || !w || !h) return -1;
  try {   // (no exception may cross the C boundary: a hostile file must fail the load, not the process)
  std::vector<uint8_t> d;
  if (!read_file(path, d)) return -1;
  size_t len = std::strlen(path);
  bool ppm = len >= 3 && (path[len - 3] == 'p' || path[len - 3] == 'P') && (path[len - 2] == 'p' || path[len - 2] == 'P') &&
             (path[len - 1] == 'm' || path[len - 1] == 'M');
  bool ok = ppm ? decode_ppm(d, rgb, w, h) : decode_png(d, rgb, w, h);
  return ok ? 0 : -1;
  } catch (const std::exception&) {
    return -1;
  }
}

extern "C" void rth_free(void* p) { std::free(p); }
