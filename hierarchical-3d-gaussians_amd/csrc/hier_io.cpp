// .hier reader / writer (host only).  Replaces gaussian_hierarchy._C.load_hierarchy /
// write_hierarchy (scene/gaussian_model.py:329,420-427).
//
// Layouts (all little-endian):
//
//  HGS_HIER_UPSTREAM (default for M = 16): the layout of graphdeco-inria/gaussian-hierarchy's loader / writer as
//  publicly documented by its source.  That submodule is NOT vendored in the reference checkout
//  (/root/reference/.gitmodules:10-12, empty directory), so this is a restatement from the public repository, not a
//  copy of anything available here, and it could not be checked against a file produced by the upstream tools
//  (DESIGN.md section 4 says so):
//      int32  P                       number of Gaussians; NEGATIVE = half-precision variant with -P Gaussians
//      float  pos[P][3]
//      float  rot[P][4]               (half in the compressed variant)
//      float  log_scale[P][3]         (half)
//      float  alpha[P]                (half)
//      float  sh[P][16][3]            (half)
//      int32  N                       number of nodes
//      int32  nodes[N][7]             depth, parent, start, count_leafs, count_merged, start_children, count_children
//      float  boxes[N][2][4]          min.xyz + extent, max.xyz + pad
//  No magic: a file is accepted as this layout only if its size matches the sizes it declares, exactly.
//
//  HGS_HIER_PRIVATE: "HGSHIER1", int32 P, N, M, 0, then xyz, shs[P][M][3], alpha, log_scales, rots, nodes, boxes --
//  kept for SH counts other than 16, which the upstream layout cannot express.
#include "common.h"

#include <stdlib.h>
#include <string.h>

using namespace hgs;

namespace {
const char kMagic[8] = {'H', 'G', 'S', 'H', 'I', 'E', 'R', '1'};

bool read_all(FILE* f, void* dst, size_t bytes) { return bytes == 0 || fread(dst, 1, bytes, f) == bytes; }
bool write_all(FILE* f, const void* src, size_t bytes) { return bytes == 0 || fwrite(src, 1, bytes, f) == bytes; }

float half_to_float(uint16_t h) {
  const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
  uint32_t exp = (h >> 10) & 0x1fu, man = h & 0x3ffu, bits;
  if (exp == 0) {
    if (man == 0) bits = sign;
    else {                                   // subnormal half -> normal float
      int e = -1;
      do { ++e; man <<= 1; } while (!(man & 0x400u));
      bits = sign | (uint32_t)(127 - 15 - e) << 23 | (man & 0x3ffu) << 13;
    }
  } else if (exp == 31) bits = sign | 0x7f800000u | man << 13;
  else bits = sign | (exp + 112u) << 23 | man << 13;
  float f;
  memcpy(&f, &bits, 4);
  return f;
}

int alloc_parts(hgs_hier_host* out) {
  const size_t p = (size_t)out->P, n = (size_t)out->N, m = (size_t)out->M;
  struct { void** dst; size_t bytes; } parts[] = {
      {(void**)&out->xyz, p * 3 * 4}, {(void**)&out->shs, p * m * 3 * 4}, {(void**)&out->alpha, p * 4},
      {(void**)&out->log_scales, p * 3 * 4}, {(void**)&out->rots, p * 4 * 4}, {(void**)&out->nodes, n * 7 * 4},
      {(void**)&out->boxes, n * 8 * 4}};
  for (auto& part : parts) {
    *part.dst = malloc(part.bytes ? part.bytes : 1);
    if (!*part.dst) return HGS_ERR_NOMEM;
  }
  return HGS_OK;
}

// reads `count` values stored as float or as IEEE half into a float array
bool read_values(FILE* f, float* dst, size_t count, bool half) {
  if (!half) return read_all(f, dst, count * 4);
  uint16_t* tmp = static_cast<uint16_t*>(malloc(count ? count * 2 : 1));
  if (!tmp) return false;
  const bool ok = read_all(f, tmp, count * 2);
  if (ok) for (size_t i = 0; i < count; ++i) dst[i] = half_to_float(tmp[i]);
  free(tmp);
  return ok;
}

int load_private(FILE* f, const char* path, hgs_hier_host* out) {
  int32_t hdr[4];
  if (fseek(f, 8, SEEK_SET) != 0 || !read_all(f, hdr, sizeof(hdr))) { set_error("%s: truncated header", path); return HGS_ERR_IO; }
  if (hdr[0] < 0 || hdr[1] < 0 || hdr[2] < 0 || hdr[2] > 64) { set_error("%s: corrupt header", path); return HGS_ERR_IO; }
  out->P = hdr[0]; out->N = hdr[1]; out->M = hdr[2]; out->reserved = HGS_HIER_PRIVATE;
  int rc = alloc_parts(out);
  if (rc) { set_error("out of memory"); return rc; }
  const size_t p = (size_t)out->P, n = (size_t)out->N, m = (size_t)out->M;
  if (!read_all(f, out->xyz, p * 12) || !read_all(f, out->shs, p * m * 12) || !read_all(f, out->alpha, p * 4) ||
      !read_all(f, out->log_scales, p * 12) || !read_all(f, out->rots, p * 16) || !read_all(f, out->nodes, n * 28) ||
      !read_all(f, out->boxes, n * 32)) {
    set_error("%s: truncated file", path);
    return HGS_ERR_IO;
  }
  return HGS_OK;
}

int load_upstream(FILE* f, const char* path, long long file_bytes, hgs_hier_host* out) {
  int32_t P_raw = 0, N = 0;
  if (fseek(f, 0, SEEK_SET) != 0 || !read_all(f, &P_raw, 4)) { set_error("%s: empty file", path); return HGS_ERR_IO; }
  const bool half = P_raw < 0;
  const long long P = half ? -(long long)P_raw : (long long)P_raw;
  const long long per = half ? (12 + 2 * (4 + 3 + 1 + 48)) : (12 + 16 + 12 + 4 + 192);
  const long long n_at = 4 + P * per;
  if (P > 0x7fffffffLL || n_at + 4 > file_bytes || fseek(f, (long)n_at, SEEK_SET) != 0 || !read_all(f, &N, 4) ||
      N < 0 || n_at + 4 + (long long)N * 60 != file_bytes) {
    set_error("%s is neither an upstream .hier file (declared sizes do not match the file size) nor an HGSHIER1 file", path);
    return HGS_ERR_IO;
  }
  out->P = (int32_t)P; out->N = N; out->M = 16; out->reserved = half ? HGS_HIER_UPSTREAM_HALF : HGS_HIER_UPSTREAM;
  int rc = alloc_parts(out);
  if (rc) { set_error("out of memory"); return rc; }
  const size_t p = (size_t)P, n = (size_t)N;
  if (fseek(f, 4, SEEK_SET) != 0 || !read_all(f, out->xyz, p * 12) || !read_values(f, out->rots, p * 4, half) ||
      !read_values(f, out->log_scales, p * 3, half) || !read_values(f, out->alpha, p, half) ||
      !read_values(f, out->shs, p * 48, half) || fseek(f, 4, SEEK_CUR) != 0 || !read_all(f, out->nodes, n * 28) ||
      !read_all(f, out->boxes, n * 32)) {
    set_error("%s: truncated file", path);
    return HGS_ERR_IO;
  }
  return HGS_OK;
}
}  // namespace

extern "C" {

void hgs_hier_free(hgs_hier_host* h) {
  if (!h) return;
  free(h->xyz); free(h->shs); free(h->alpha); free(h->log_scales); free(h->rots); free(h->nodes); free(h->boxes);
  memset(h, 0, sizeof(*h));
}

int hgs_hier_load(const char* path, hgs_hier_host* out) {
  if (!path || !out) { set_error("null argument"); return HGS_ERR_INVALID; }
  memset(out, 0, sizeof(*out));
  FILE* f = fopen(path, "rb");
  if (!f) { set_error("cannot open %s", path); return HGS_ERR_IO; }
  char magic[8] = {0};
  const size_t got = fread(magic, 1, 8, f);
  long long bytes = -1;
  if (fseek(f, 0, SEEK_END) == 0) bytes = ftell(f);
  int rc;
  if (got == 8 && memcmp(magic, kMagic, 8) == 0) rc = load_private(f, path, out);
  else rc = load_upstream(f, path, bytes, out);
  fclose(f);
  if (rc) hgs_hier_free(out);
  return rc;
}

int hgs_hier_write(const char* path, const hgs_hier_host* in) {
  if (!path || !in) { set_error("null argument"); return HGS_ERR_INVALID; }
  if (in->P < 0 || in->N < 0 || in->M < 0) { set_error("negative sizes"); return HGS_ERR_INVALID; }
  const int fmt = in->reserved;
  if (fmt != HGS_HIER_UPSTREAM && fmt != HGS_HIER_PRIVATE) { set_error("hgs_hier_write: format %d cannot be written (0 = upstream float layout, 1 = HGSHIER1)", fmt); return HGS_ERR_INVALID; }
  if (fmt == HGS_HIER_UPSTREAM && in->M != 16) { set_error("the upstream .hier layout stores exactly 16 SH coefficients per Gaussian (got %d); use HGS_HIER_PRIVATE", in->M); return HGS_ERR_INVALID; }
  FILE* f = fopen(path, "wb");
  if (!f) { set_error("cannot create %s", path); return HGS_ERR_IO; }
  const size_t p = (size_t)in->P, n = (size_t)in->N, m = (size_t)in->M;
  bool ok;
  if (fmt == HGS_HIER_UPSTREAM) {
    ok = write_all(f, &in->P, 4) && write_all(f, in->xyz, p * 12) && write_all(f, in->rots, p * 16) &&
         write_all(f, in->log_scales, p * 12) && write_all(f, in->alpha, p * 4) && write_all(f, in->shs, p * 192) &&
         write_all(f, &in->N, 4) && write_all(f, in->nodes, n * 28) && write_all(f, in->boxes, n * 32);
  } else {
    const int32_t hdr[4] = {in->P, in->N, in->M, 0};
    ok = write_all(f, kMagic, 8) && write_all(f, hdr, sizeof(hdr)) && write_all(f, in->xyz, p * 12) &&
         write_all(f, in->shs, p * m * 12) && write_all(f, in->alpha, p * 4) && write_all(f, in->log_scales, p * 12) &&
         write_all(f, in->rots, p * 16) && write_all(f, in->nodes, n * 28) && write_all(f, in->boxes, n * 32);
  }
  const bool closed = fclose(f) == 0;
  if (!ok || !closed) { set_error("short write to %s", path); return HGS_ERR_IO; }
  return HGS_OK;
}

}  // extern "C"
