// 24-bit uncompressed BMP reader/writer with the reference's conventions
// (src/fileio/bitmap.cpp:17-155): rows are stored bottom-up, the in-memory
// buffer's row 0 is written first, pixels are RGB in memory and BGR on disk,
// rows are padded to a multiple of four bytes, header = 14 + 40 bytes.
// Two habits of the reference's writer are kept so that files compare equal:
// bfSize counts the file header as sizeof(BMP_BITMAPFILEHEADER), which is 16 on
// x86-64 (two bytes more than the file holds, bitmap.cpp:109), and the padding
// bytes of a row are the leading bytes of the next row in memory, because the
// scanline is copied with its padded length (bitmap.cpp:138-149).  The last
// row's padding, which the reference reads from past the end of its buffer, is
// zero here.
#include <cstdint>
#include <cstdio>
#include <cstring>

#include "scene.h"

namespace rta {
namespace {
void put16(unsigned char* p, uint16_t v) { p[0] = v & 0xff; p[1] = v >> 8; }
void put32(unsigned char* p, uint32_t v) { for (int i = 0; i < 4; i++) p[i] = (v >> (8 * i)) & 0xff; }
uint16_t get16(const unsigned char* p) { return (uint16_t)(p[0] | (p[1] << 8)); }
uint32_t get32(const unsigned char* p) {
  return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}
}  // namespace

bool writeBMP(const std::string& path, int width, int height, const unsigned char* rgb) {
  int rowBytes = width * 3;
  int pad = (rowBytes % 4) ? 4 - (rowBytes % 4) : 0;
  rowBytes += pad;
  unsigned char hdr[54];
  std::memset(hdr, 0, sizeof hdr);
  put16(hdr + 0, 0x4d42);
  put32(hdr + 2, 16 + 40 + (uint32_t)rowBytes * height);
  put32(hdr + 10, 54);
  put32(hdr + 14, 40);
  put32(hdr + 18, (uint32_t)width);
  put32(hdr + 22, (uint32_t)height);
  put16(hdr + 26, 1);
  put16(hdr + 28, 24);
  put32(hdr + 38, (uint32_t)(int)(100 / 2.54 * 72));
  put32(hdr + 42, (uint32_t)(int)(100 / 2.54 * 72));
  FILE* f = std::fopen(path.c_str(), "wb");
  if (!f) return false;
  std::fwrite(hdr, 1, sizeof hdr, f);
  std::vector<unsigned char> line(rowBytes, 0);
  for (int j = 0; j < height; j++) {
    const unsigned char* src = rgb + (size_t)j * 3 * width;
    for (int i = 0; i < width; i++) {
      line[i * 3 + 0] = src[i * 3 + 2];
      line[i * 3 + 1] = src[i * 3 + 1];
      line[i * 3 + 2] = src[i * 3 + 0];
    }
    for (int k = 0; k < pad; k++) line[width * 3 + k] = (j + 1 < height) ? src[width * 3 + k] : 0;
    std::fwrite(line.data(), 1, rowBytes, f);
  }
  std::fclose(f);
  return true;
}

bool readBMP(const std::string& path, int& width, int& height, std::vector<unsigned char>& rgb) {
  FILE* f = std::fopen(path.c_str(), "rb");
  if (!f) return false;
  unsigned char hdr[54];
  if (std::fread(hdr, 1, sizeof hdr, f) != sizeof hdr || get16(hdr) != 0x4d42 ||
      get16(hdr + 28) != 24) {
    std::fclose(f);
    return false;
  }
  uint32_t off = get32(hdr + 10);
  width = (int)get32(hdr + 18);
  height = (int)get32(hdr + 22);
  int rowBytes = width * 3;
  int pad = (rowBytes % 4) ? 4 - (rowBytes % 4) : 0;
  std::fseek(f, (long)off, SEEK_SET);
  rgb.assign((size_t)width * height * 3, 0);
  std::vector<unsigned char> line(rowBytes + pad);
  for (int j = 0; j < height; j++) {
    if (std::fread(line.data(), 1, line.size(), f) != line.size()) {
      std::fclose(f);
      return false;
    }
    unsigned char* dst = rgb.data() + (size_t)j * rowBytes;
    for (int i = 0; i < width; i++) {
      dst[i * 3 + 0] = line[i * 3 + 2];
      dst[i * 3 + 1] = line[i * 3 + 1];
      dst[i * 3 + 2] = line[i * 3 + 0];
    }
  }
  std::fclose(f);
  return true;
}

}  // namespace rta
