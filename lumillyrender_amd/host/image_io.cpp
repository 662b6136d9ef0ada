// image_io.cpp -- film output and IBL input (stands in for the `image` 0.18 crate).
//   save_png : img.rs:52-63 + main.rs:158-164,171-173 (gamma, clamp, TRUNCATING u8 quantisation)
//   save_hdr : img.rs:40-50 (Radiance RGBE, run-length encoded scanlines)
//   load_hdr : sky.rs:40-55 (HDRDecoder::read_image_hdr -> linear f32 RGB)
// RGBE conversions follow the crate's published behaviour (UNPINNED: Cargo.lock is not in the
// reference tree): decode c * 2^(e-136); encode e = floor(log2(max)) + 1 + 128, c = trunc(v / 2^(e-128) * 256).
#include "host_internal.h"
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <zlib.h>

namespace lrhost {

uint8_t to_color(float x, float gamma) {                         // main.rs:171-173
  float c = std::fmin(std::fmax(x, 0.0f), 1.0f);               // f32::max/min return the non-NaN operand
  float v = std::pow(c, 1.0f / gamma) * 255.0f;
  if (!(v > 0.0f)) return 0;                                     // `as u8` saturates; NaN -> 0
  if (v >= 255.0f) return 255;
  return (uint8_t)v;
}

namespace {

void put_u32(std::vector<uint8_t>& b, uint32_t v) { b.push_back(v >> 24); b.push_back(v >> 16); b.push_back(v >> 8); b.push_back(v); }
void png_chunk(std::vector<uint8_t>& out, const char* type, const std::vector<uint8_t>& data) {
  put_u32(out, (uint32_t)data.size());
  size_t start = out.size();
  out.insert(out.end(), type, type + 4);
  out.insert(out.end(), data.begin(), data.end());
  uint32_t crc = (uint32_t)crc32(0L, out.data() + start, (uInt)(out.size() - start));
  put_u32(out, crc);
}

struct Rgbe { uint8_t c[3], e; };
Rgbe to_rgbe8(float r, float g, float b) {
  Rgbe o; std::memset(&o, 0, sizeof(o));
  float mx = std::fmax(r, std::fmax(g, b));
  if (!(mx > 0.0f)) return o;
  int exp; (void)std::frexp(mx, &exp);                       // mx = m * 2^exp, m in [0.5, 1)  (= floor(log2 mx) + 1)
  float mul = std::ldexp(1.0f, exp);
  float v[3] = {r, g, b};
  for (int k = 0; k < 3; ++k) {
    float t = std::trunc(v[k] / mul * 256.0f);
    o.c[k] = t <= 0.0f ? 0 : (t >= 255.0f ? 255 : (uint8_t)t);
  }
  o.e = (uint8_t)(exp + 128);
  return o;
}

void rle_channel(std::vector<uint8_t>& out, const uint8_t* d, int n) {
  int i = 0;
  while (i < n) {
    int run = 1;
    while (i + run < n && run < 127 && d[i + run] == d[i]) ++run;
    if (run >= 3) { out.push_back((uint8_t)(128 + run)); out.push_back(d[i]); i += run; continue; }
    // literal span until the next run of >= 3
    int start = i, len = 0;
    while (i < n && len < 128) {
      int r = 1;
      while (i + r < n && r < 3 && d[i + r] == d[i]) ++r;
      if (r >= 3) break;
      ++i; ++len;
    }
    out.push_back((uint8_t)len);
    out.insert(out.end(), d + start, d + start + len);
  }
}

}  // namespace

void save_png(const std::string& path, const float* rgb, int w, int h, size_t stride, float gamma) {
  if (w <= 0 || h <= 0) fail(LR_EINVAL, "png: empty image");
  std::vector<uint8_t> q((size_t)w * h * 3);
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w * 3; ++x) q[(size_t)y * w * 3 + x] = to_color(rgb[(size_t)y * stride + x], gamma);
  write_png_rgb8(path, q.data(), w, h, (size_t)w * 3);
}

void write_png_rgb8(const std::string& path, const uint8_t* rgb8, int w, int h, size_t stride) {
  if (w <= 0 || h <= 0) fail(LR_EINVAL, "png: empty image");
  std::vector<uint8_t> raw((size_t)h * ((size_t)w * 3 + 1));
  for (int y = 0; y < h; ++y) {
    uint8_t* row = raw.data() + (size_t)y * ((size_t)w * 3 + 1);
    row[0] = 0;                                                  // filter: none
    std::memcpy(row + 1, rgb8 + (size_t)y * stride, (size_t)w * 3);
  }
  uLongf clen = compressBound((uLong)raw.size());
  std::vector<uint8_t> comp(clen);
  if (compress2(comp.data(), &clen, raw.data(), (uLong)raw.size(), 6) != Z_OK) fail(LR_EIO, "png: zlib failure");
  comp.resize(clen);
  std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
  std::vector<uint8_t> ihdr;
  put_u32(ihdr, (uint32_t)w); put_u32(ihdr, (uint32_t)h);
  ihdr.push_back(8); ihdr.push_back(2); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);   // 8-bit RGB
  png_chunk(out, "IHDR", ihdr);
  png_chunk(out, "IDAT", comp);
  png_chunk(out, "IEND", std::vector<uint8_t>());
  std::ofstream f(path, std::ios::binary);
  if (!f) fail(LR_EIO, "png: cannot create `" + path + "`");     // File::create().unwrap() in the reference
  f.write((const char*)out.data(), (std::streamsize)out.size());
}

void save_hdr(const std::string& path, const float* rgb, int w, int h, size_t stride) {
  if (w <= 0 || h <= 0) fail(LR_EINVAL, "hdr: empty image");
  std::vector<uint8_t> q((size_t)w * h * 4);
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) {
      const float* p = rgb + (size_t)y * stride + (size_t)x * 3;
      Rgbe e = to_rgbe8(p[0], p[1], p[2]);
      uint8_t* o = q.data() + ((size_t)y * w + x) * 4;
      o[0] = e.c[0]; o[1] = e.c[1]; o[2] = e.c[2]; o[3] = e.e;
    }
  write_hdr_rgbe(path, q.data(), w, h, (size_t)w * 4);
}

void write_hdr_rgbe(const std::string& path, const uint8_t* rgbe, int w, int h, size_t stride) {
  if (w <= 0 || h <= 0) fail(LR_EINVAL, "hdr: empty image");
  std::ofstream f(path, std::ios::binary);
  if (!f) fail(LR_EIO, "hdr: cannot create `" + path + "`");
  char hdr[128];
  int n = std::snprintf(hdr, sizeof(hdr), "#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y %d +X %d\n", h, w);
  f.write(hdr, n);
  std::vector<Rgbe> line((size_t)w);
  std::vector<uint8_t> chan((size_t)w), out;
  for (int y = 0; y < h; ++y) {
    for (int x = 0; x < w; ++x) {
      const uint8_t* p = rgbe + (size_t)y * stride + (size_t)x * 4;
      line[x].c[0] = p[0]; line[x].c[1] = p[1]; line[x].c[2] = p[2]; line[x].e = p[3];
    }
    out.clear();
    if (w < 8 || w > 32767) {
      for (int x = 0; x < w; ++x) { out.push_back(line[x].c[0]); out.push_back(line[x].c[1]); out.push_back(line[x].c[2]); out.push_back(line[x].e); }
    } else {
      out.push_back(2); out.push_back(2); out.push_back((uint8_t)(w >> 8)); out.push_back((uint8_t)(w & 255));
      for (int k = 0; k < 4; ++k) {
        for (int x = 0; x < w; ++x) chan[x] = k < 3 ? line[x].c[k] : line[x].e;
        rle_channel(out, chan.data(), w);
      }
    }
    f.write((const char*)out.data(), (std::streamsize)out.size());
  }
}

void load_hdr(const std::string& path, std::vector<float>& texels, int& w, int& h) {
  std::ifstream f(path, std::ios::binary);
  if (!f) fail(LR_EIO, "hdr: cannot open `" + path + "`");
  std::string line;
  bool magic = false; w = h = 0;
  while (std::getline(f, line)) {
    if (!magic) { if (line.compare(0, 2, "#?") != 0) fail(LR_EIO, "hdr: `" + path + "` is not a Radiance file"); magic = true; continue; }
    if (line.empty()) break;
  }
  if (!std::getline(f, line)) fail(LR_EIO, "hdr: missing resolution line");
  if (std::sscanf(line.c_str(), "-Y %d +X %d", &h, &w) != 2 || w <= 0 || h <= 0) fail(LR_EUNSUPPORTED, "hdr: unsupported orientation `" + line + "`");
  texels.assign((size_t)w * h * 3, 0.0f);
  std::vector<uint8_t> sl((size_t)w * 4);
  auto rd = [&f]() -> int { return f.get(); };
  for (int y = 0; y < h; ++y) {
    int c0 = rd(), c1 = rd(), c2 = rd(), c3 = rd();
    if (c3 < 0) fail(LR_EIO, "hdr: truncated file");
    if (c0 == 2 && c1 == 2 && (c2 & 0x80) == 0 && w >= 8 && w < 32768) {
      if (((c2 << 8) | c3) != w) fail(LR_EIO, "hdr: scanline width mismatch");
      for (int k = 0; k < 4; ++k) {
        int x = 0;
        while (x < w) {
          int cnt = rd();
          if (cnt < 0) fail(LR_EIO, "hdr: truncated file");
          if (cnt > 128) { int v = rd(); cnt -= 128; if (x + cnt > w) fail(LR_EIO, "hdr: bad run"); for (int i = 0; i < cnt; ++i) sl[(size_t)(x++) * 4 + k] = (uint8_t)v; }
          else { if (cnt == 0 || x + cnt > w) fail(LR_EIO, "hdr: bad literal"); for (int i = 0; i < cnt; ++i) sl[(size_t)(x++) * 4 + k] = (uint8_t)rd(); }
        }
      }
    } else {
      sl[0] = (uint8_t)c0; sl[1] = (uint8_t)c1; sl[2] = (uint8_t)c2; sl[3] = (uint8_t)c3;
      f.read((char*)sl.data() + 4, (std::streamsize)((size_t)w * 4 - 4));
      if (!f) fail(LR_EIO, "hdr: truncated file");
    }
    for (int x = 0; x < w; ++x) {
      const uint8_t* p = sl.data() + (size_t)x * 4;
      float* o = texels.data() + ((size_t)y * w + x) * 3;
      if (p[3] == 0) { o[0] = o[1] = o[2] = 0.0f; }
      else { float e = std::ldexp(1.0f, (int)p[3] - 136); o[0] = e * (float)p[0]; o[1] = e * (float)p[1]; o[2] = e * (float)p[2]; }
    }
  }
}

}  // namespace lrhost
