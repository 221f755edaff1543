// Flat-buffer VTK writer (host code).
//
// Replaces, for callers that hold the flat mesh buffers, the itk::Mesh fill + itk::VTKPolyDataWriter pass of
// Testing/CuberilleTest01.cxx:161-187: same legacy-ASCII POLYDATA bytes as the writer the unchanged driver
// uses (9 significant digits, "x y z" per point, "k id0 .. idk-1" per polygon), formatted by a pool of host
// threads straight from float[3n] / uint64[k m], without a per-cell heap object in between.
#include "../../include/cuberille_hip.h"

#include <algorithm>
#include <charconv>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace {

inline char *put_float(char *p, float v) {
  if (std::isfinite(v)) {
    // == printf("%.9g", (double)v), which is what an ostream with precision(9) prints
    return std::to_chars(p, p + 32, v, std::chars_format::general, 9).ptr;
  }
  return p + std::snprintf(p, 32, "%.9g", (double)v);
}

inline char *put_u64(char *p, uint64_t v) { return std::to_chars(p, p + 24, v).ptr; }

void format_points(const float *pts, size_t i0, size_t i1, std::string &out) {
  out.resize((i1 - i0) * 3 * 20 + 16);
  char *p = &out[0];
  for (size_t i = i0; i < i1; i++) {
    p = put_float(p, pts[3 * i]);
    *p++ = ' ';
    p = put_float(p, pts[3 * i + 1]);
    *p++ = ' ';
    p = put_float(p, pts[3 * i + 2]);
    *p++ = '\n';
  }
  out.resize(p - &out[0]);
}

void format_cells(const uint64_t *cells, int k, size_t i0, size_t i1, std::string &out) {
  out.resize((i1 - i0) * (size_t)(k * 21 + 4) + 16);
  char *p = &out[0];
  for (size_t i = i0; i < i1; i++) {
    p = put_u64(p, (uint64_t)k);
    for (int j = 0; j < k; j++) {
      *p++ = ' ';
      p = put_u64(p, cells[i * k + j]);
    }
    *p++ = '\n';
  }
  out.resize(p - &out[0]);
}

// items [0,n) in rounds of T chunks: threads format, the caller's thread writes the chunks in order
template <class F> bool write_section(std::FILE *f, size_t n, int threads, F format) {
  const size_t chunk = 1u << 18;
  std::vector<std::string> buf(threads);
  for (size_t base = 0; base < n; base += chunk * threads) {
    std::vector<std::thread> pool;
    int used = 0;
    for (int t = 0; t < threads; t++) {
      const size_t i0 = base + chunk * t, i1 = std::min(n, i0 + chunk);
      if (i0 >= n) break;
      used++;
      if (t == 0) continue;  // the calling thread takes chunk 0
      pool.emplace_back([&, t, i0, i1] { format(i0, i1, buf[t]); });
    }
    format(base, std::min(n, base + chunk), buf[0]);
    for (auto &th : pool) th.join();
    for (int t = 0; t < used; t++)
      if (std::fwrite(buf[t].data(), 1, buf[t].size(), f) != buf[t].size()) return false;
  }
  return true;
}

}  // namespace

extern "C" int cuberille_write_vtk_buffers(const char *path, const float *points, uint64_t n_points,
                                           const uint64_t *cells, uint64_t n_cells, int verts_per_cell,
                                           int n_threads) {
  if (!path || (n_points && !points) || (n_cells && !cells)) return CUBERILLE_ERR_ARGUMENT;
  if (n_cells && verts_per_cell != 3 && verts_per_cell != 4) return CUBERILLE_ERR_ARGUMENT;
  if (n_threads <= 0) n_threads = (int)std::min(32u, std::max(1u, std::thread::hardware_concurrency()));
  std::FILE *f = std::fopen(path, "wb");
  if (!f) return CUBERILLE_ERR_ARGUMENT;
  std::vector<char> iobuf(8u << 20);
  std::setvbuf(f, iobuf.data(), _IOFBF, iobuf.size());
  bool ok = std::fprintf(f, "# vtk DataFile Version 2.0\nFile written by itkVTKPolyDataWriter\nASCII\nDATASET POLYDATA\n"
                            "POINTS %llu float\n", (unsigned long long)n_points) > 0;
  ok = ok && write_section(f, n_points, n_threads,
                           [&](size_t a, size_t b, std::string &s) { format_points(points, a, b, s); });
  ok = ok && std::fprintf(f, "POLYGONS %llu %llu\n", (unsigned long long)n_cells,
                          (unsigned long long)(n_cells * (uint64_t)(verts_per_cell + 1))) > 0;
  ok = ok && write_section(f, n_cells, n_threads, [&](size_t a, size_t b, std::string &s) {
         format_cells(cells, verts_per_cell, a, b, s);
       });
  ok = (std::fclose(f) == 0) && ok;
  return ok ? CUBERILLE_OK : CUBERILLE_ERR_STATE;
}
