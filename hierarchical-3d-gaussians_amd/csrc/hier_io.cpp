// .hier reader / writer (host only).  Replaces gaussian_hierarchy._C.load_hierarchy /
// write_hierarchy (scene/gaussian_model.py:329,420-427).  The reference's binary layout
// lives in the absent gaussian-hierarchy submodule and cannot be recovered here, so this
// is our own documented little-endian layout (DESIGN.md '.hier layout'):
//
//   char[8]  magic "HGSHIER1"
//   int32    P, N, M, reserved(0)
//   float    xyz[P*3], shs[P*M*3], alpha[P], log_scales[P*3], rots[P*4]
//   int32    nodes[N*7]
//   float    boxes[N*8]
#include "common.h"

#include <stdlib.h>
#include <string.h>

using namespace hgs;

namespace {
const char kMagic[8] = {'H', 'G', 'S', 'H', 'I', 'E', 'R', '1'};

bool read_all(FILE* f, void* dst, size_t bytes) { return bytes == 0 || fread(dst, 1, bytes, f) == bytes; }
bool write_all(FILE* f, const void* src, size_t bytes) { return bytes == 0 || fwrite(src, 1, bytes, f) == bytes; }
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
  char magic[8];
  int32_t hdr[4];
  if (!read_all(f, magic, 8) || memcmp(magic, kMagic, 8) != 0 || !read_all(f, hdr, sizeof(hdr))) {
    fclose(f);
    set_error("%s is not an HGSHIER1 file", path);
    return HGS_ERR_IO;
  }
  const int32_t P = hdr[0], N = hdr[1], M = hdr[2];
  if (P < 0 || N < 0 || M < 0 || M > 64) { fclose(f); set_error("%s: corrupt header", path); return HGS_ERR_IO; }
  out->P = P; out->N = N; out->M = M;
  const size_t p = (size_t)P, n = (size_t)N, m = (size_t)M;
  struct { void** dst; size_t bytes; } parts[] = {
      {(void**)&out->xyz, p * 3 * 4}, {(void**)&out->shs, p * m * 3 * 4}, {(void**)&out->alpha, p * 4},
      {(void**)&out->log_scales, p * 3 * 4}, {(void**)&out->rots, p * 4 * 4}, {(void**)&out->nodes, n * 7 * 4},
      {(void**)&out->boxes, n * 8 * 4}};
  for (auto& part : parts) {
    *part.dst = malloc(part.bytes ? part.bytes : 1);
    if (!*part.dst) { fclose(f); hgs_hier_free(out); set_error("out of memory"); return HGS_ERR_NOMEM; }
    if (!read_all(f, *part.dst, part.bytes)) {
      fclose(f);
      hgs_hier_free(out);
      set_error("%s: truncated file", path);
      return HGS_ERR_IO;
    }
  }
  fclose(f);
  return HGS_OK;
}

int hgs_hier_write(const char* path, const hgs_hier_host* in) {
  if (!path || !in) { set_error("null argument"); return HGS_ERR_INVALID; }
  if (in->P < 0 || in->N < 0 || in->M < 0) { set_error("negative sizes"); return HGS_ERR_INVALID; }
  FILE* f = fopen(path, "wb");
  if (!f) { set_error("cannot create %s", path); return HGS_ERR_IO; }
  const int32_t hdr[4] = {in->P, in->N, in->M, 0};
  const size_t p = (size_t)in->P, n = (size_t)in->N, m = (size_t)in->M;
  const bool ok = write_all(f, kMagic, 8) && write_all(f, hdr, sizeof(hdr)) && write_all(f, in->xyz, p * 3 * 4) &&
                  write_all(f, in->shs, p * m * 3 * 4) && write_all(f, in->alpha, p * 4) &&
                  write_all(f, in->log_scales, p * 3 * 4) && write_all(f, in->rots, p * 4 * 4) &&
                  write_all(f, in->nodes, n * 7 * 4) && write_all(f, in->boxes, n * 8 * 4);
  const bool closed = fclose(f) == 0;
  if (!ok || !closed) { set_error("short write to %s", path); return HGS_ERR_IO; }
  return HGS_OK;
}

}  // extern "C"
