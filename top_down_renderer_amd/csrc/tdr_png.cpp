// tdr_png.cpp — the 8-bit greyscale PNG files of the reference's raster cache (TopDownMap::saveRasterizedMaps /
// loadRasterizedMaps, src/top_down_map.cpp:197-224: cv::imwrite / cv::imread(IMREAD_GRAYSCALE) of CV_8UC1 images), read and
// written over zlib.  Reader: colour type 0, bit depth 8, non-interlaced — what cv::imwrite produces for these images —
// every filter type; chunk CRCs checked; any other PNG is refused by name.  Writer: filter 0, one IDAT.
#include <zlib.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <exception>
#include <vector>

#include "tdr_common.h"

static uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
static void put32(uint8_t* p, uint32_t v) { p[0] = v >> 24; p[1] = v >> 16; p[2] = v >> 8; p[3] = v; }
static const uint8_t PNG_SIG[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};

int tdr_png_read_gray8(const char* path, std::vector<uint8_t>& px, int& w, int& h) {
  FILE* fh = fopen(path, "rb");
  if (!fh) return fail(TDR_ERR_ARG, "png: cannot open %s", path);
  std::vector<uint8_t> file;
  uint8_t buf[65536];
  size_t got;
  while ((got = fread(buf, 1, sizeof(buf), fh)) > 0) file.insert(file.end(), buf, buf + got);
  fclose(fh);
  if (file.size() < 8 + 25 || memcmp(file.data(), PNG_SIG, 8) != 0) return fail(TDR_ERR_ARG, "png: %s is not a PNG file", path);
  size_t at = 8;
  bool have_hdr = false, done = false;
  std::vector<uint8_t> idat;
  w = h = 0;
  while (!done) {
    if (at + 12 > file.size()) return fail(TDR_ERR_ARG, "png: %s is truncated", path);
    const uint32_t len = be32(&file[at]);
    if (len > 0x7FFFFFFFu || at + 12 + (size_t)len > file.size()) return fail(TDR_ERR_ARG, "png: %s is truncated", path);
    const uint8_t* type = &file[at + 4];
    const uint8_t* data = &file[at + 8];
    if ((uint32_t)crc32(crc32(0L, Z_NULL, 0), type, 4 + len) != be32(data + len))
      return fail(TDR_ERR_ARG, "png: %s has a chunk with a wrong CRC", path);
    if (!memcmp(type, "IHDR", 4)) {
      if (len != 13) return fail(TDR_ERR_ARG, "png: %s has a malformed header", path);
      const uint32_t ww = be32(data), hh = be32(data + 4);
      if (ww < 1 || hh < 1 || ww > (1u << 24) || hh > (1u << 24)) return fail(TDR_ERR_ARG, "png: %s has an unusable size", path);
      if (data[8] != 8 || data[9] != 0 || data[10] != 0 || data[11] != 0 || data[12] != 0)
        return fail(TDR_ERR_ARG, "png: %s is not an 8-bit greyscale, non-interlaced image (bit depth %d, colour type %d, "
                    "interlace %d)", path, data[8], data[9], data[12]);
      w = (int)ww; h = (int)hh;
      have_hdr = true;
    } else if (!memcmp(type, "IDAT", 4)) {
      if (!have_hdr) return fail(TDR_ERR_ARG, "png: %s has image data before its header", path);
      idat.insert(idat.end(), data, data + len);
    } else if (!memcmp(type, "IEND", 4)) {
      done = true;
    } else if (!(type[0] & 0x20)) {   // an unknown CRITICAL chunk (PLTE has no place in a greyscale image either)
      return fail(TDR_ERR_ARG, "png: %s holds a critical chunk this reader does not know", path);
    }
    at += 12 + (size_t)len;
  }
  if (!have_hdr || idat.empty()) return fail(TDR_ERR_ARG, "png: %s has no image data", path);
  const size_t stride = (size_t)w + 1;
  // deflate expands at most ~1032 : 1: a header that announces more pixels than the image data can hold is refused before
  // anything of that size is allocated (a damaged or hostile file must not be able to ask for 2^48 bytes)
  if ((uint64_t)stride * (uint64_t)h > (uint64_t)idat.size() * 1032u + 64u)
    return fail(TDR_ERR_ARG, "png: %s announces %d x %d pixels but holds %zu bytes of image data", path, w, h, idat.size());
  std::vector<uint8_t> raw(stride * (size_t)h);
  uLongf out_len = (uLongf)raw.size();
  if (uncompress(raw.data(), &out_len, idat.data(), (uLong)idat.size()) != Z_OK || out_len != raw.size())
    return fail(TDR_ERR_ARG, "png: the image data of %s does not inflate to %d x %d bytes", path, w, h);
  px.assign((size_t)w * h, 0);
  for (int y = 0; y < h; y++) {   // unfilter, one byte per pixel
    const uint8_t* in = &raw[stride * y];
    uint8_t* out = &px[(size_t)w * y];
    const uint8_t* up = y ? &px[(size_t)w * (y - 1)] : nullptr;
    const int ft = in[0];
    if (ft > 4) return fail(TDR_ERR_ARG, "png: %s uses filter type %d", path, ft);
    for (int x = 0; x < w; x++) {
      const int a = x ? out[x - 1] : 0, b = up ? up[x] : 0, c = (x && up) ? up[x - 1] : 0;
      int pred = 0;
      if (ft == 1) pred = a;
      else if (ft == 2) pred = b;
      else if (ft == 3) pred = (a + b) >> 1;
      else if (ft == 4) {
        const int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
        pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
      }
      out[x] = (uint8_t)(in[x + 1] + pred);
    }
  }
  return TDR_OK;
}

int tdr_png_write_gray8(const char* path, const uint8_t* px, int w, int h) {
  if (!px || w < 1 || h < 1) return fail(TDR_ERR_ARG, "png: nothing to write");
  const size_t stride = (size_t)w + 1;
  std::vector<uint8_t> raw(stride * (size_t)h);
  for (int y = 0; y < h; y++) {
    raw[stride * y] = 0;   // filter type 0
    memcpy(&raw[stride * y + 1], px + (size_t)w * y, (size_t)w);
  }
  uLongf zlen = compressBound((uLong)raw.size());
  std::vector<uint8_t> z(zlen);
  if (compress2(z.data(), &zlen, raw.data(), (uLong)raw.size(), 6) != Z_OK) return fail(TDR_ERR_ARG, "png: deflate failed");
  FILE* fh = fopen(path, "wb");
  if (!fh) return fail(TDR_ERR_ARG, "png: cannot write %s", path);
  auto chunk = [&](const char* type, const uint8_t* data, uint32_t len) {
    std::vector<uint8_t> c(12 + (size_t)len);
    put32(&c[0], len);
    memcpy(&c[4], type, 4);
    if (len) memcpy(&c[8], data, len);
    put32(&c[8 + len], (uint32_t)crc32(crc32(0L, Z_NULL, 0), &c[4], 4 + len));
    return fwrite(c.data(), 1, c.size(), fh) == c.size();
  };
  uint8_t hdr[13];
  put32(hdr, (uint32_t)w);
  put32(hdr + 4, (uint32_t)h);
  hdr[8] = 8; hdr[9] = 0; hdr[10] = 0; hdr[11] = 0; hdr[12] = 0;
  const bool ok = fwrite(PNG_SIG, 1, 8, fh) == 8 && chunk("IHDR", hdr, 13) && chunk("IDAT", z.data(), (uint32_t)zlen) &&
                  chunk("IEND", nullptr, 0);
  fclose(fh);
  return ok ? TDR_OK : fail(TDR_ERR_ARG, "png: short write to %s", path);
}

// C-ABI face of the codec (host only, no device): px_out holds capacity bytes; *w / *h are set even when the image does not fit
extern "C" int tdr_png_read_gray8_host(const char* path, uint8_t* px_out, int64_t capacity, int* w, int* h) {
  if (!path || !w || !h) return fail(TDR_ERR_ARG, "png: null pointer");
  try {   // (no exception crosses the C ABI: an allocation failure is an error code)
    std::vector<uint8_t> px;
    if (int rc = tdr_png_read_gray8(path, px, *w, *h)) return rc;
    if (!px_out || capacity < (int64_t)px.size()) return fail(TDR_ERR_ARG, "png: %s needs %zu bytes", path, px.size());
    memcpy(px_out, px.data(), px.size());
    return TDR_OK;
  } catch (const std::exception& e) {
    return fail(TDR_ERR_NOMEM, "png: %s: %s", path, e.what());
  }
}
extern "C" int tdr_png_write_gray8_host(const char* path, const uint8_t* px, int w, int h) {
  if (!path) return fail(TDR_ERR_ARG, "png: null pointer");
  try {
    return tdr_png_write_gray8(path, px, w, h);
  } catch (const std::exception& e) {
    return fail(TDR_ERR_NOMEM, "png: %s: %s", path, e.what());
  }
}
