// Internal declarations shared by the HIP kernels (cuberille_kernels.hip) and the
// C-ABI host layer (cuberille_api.hip).  Not installed; the public surface is
// include/cuberille_hip.h.
#ifndef CUBERILLE_INTERNAL_H
#define CUBERILLE_INTERNAL_H

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cuberille {

typedef unsigned long long u64;
typedef unsigned int u32;

// Geometry of the image, precomputed on the host in double (reference:
// ImageBase::m_IndexToPhysicalPoint / m_PhysicalPointToIndex as used through
// txx:266 and the interpolators' Evaluate(point), txx:451,455).
struct Geo {
  double i2p[9];      // Direction * diag(spacing)
  double p2i[9];      // its inverse
  double origin[3];
  double spacing[3];
  double dir[9];
  float gcoef[3];     // derivative tap coefficient per axis: float(0.5 * (1/spacing))
};

// Layout of the packed inside-bit volume and of the slab being processed.
//   bits[(z*ny + y)*W + k] : bit b = inside(x = 64k+b, y, z), z local to the buffer,
//   tail bits (x >= nx) of the last word are 0.
struct Grid {
  int nx, ny, nzb;     // buffer dims (voxels)
  int W;               // 64-voxel words per x-row
  int lastpos;         // (nx-1) & 63
  int wShift, yShift;  // log2(W), log2(ny) when those are powers of two, else -1 (word index -> k, y, z without dividing)
  int cz0;             // first local slice whose words are counted (own_z0-1 when that exists)
  int oz0, oz1;        // local slices [oz0, oz1) this rank emits
  long long zglob0;    // global z of local slice 0
  long long gnz;       // global Nz
};

struct Totals {        // device-resident, mirrored to pinned host memory
  u64 totV, totQ;      // created vertices / quads in the counted range
  u64 V0, Q0;          // of which before the first owned slice
  u64 iters;           // projection iterations (atomic)
  u32 err;             // device-side error flags
  u32 nVertexWords;   // entries in the vertex-word queue (words that create at least one vertex)
};

enum { ERRF_ALIAS_UNKNOWN = 1 };

struct Workspace {     // device pointers valid for one count/emit pair
  const void *vox;
  u64 *bits;
  u64 *flatBits;       // scratch for rows that are not whole words: inside bits in flat voxel order (or null)
  u32 *sliceOcc;
  int *alias;          // per local slice: source slice of the empty-slice aliasing or -1
  u32 *prefix;         // per counted word: exclusive in-segment prefix, V | Q<<16
  u64 *segV, *segQ;    // per 64-word segment totals
  u64 *segBaseV, *segBaseQ;
  Totals *totals;
  float *points;
  u64 *cells;
  u32 *cmap;           // dense lattice-corner -> vertex index map (null: recompute ids instead)
  u32 *headV, *headQ;  // word that produces output 64*i (null: per-lane binary search instead)
  u32 *vqueue;         // counted-range indices of the words that create vertices, in no particular order (or null)
};

struct Params {
  double iso;
  double thr, step, relax;
  u32 max_steps;
  int triangles, project, q1;
};

// launchers (cuberille_kernels.hip); all asynchronous on `s`
hipError_t launch_classify(int pixel_type, const Workspace &w, const Grid &g, double iso, int z0, int z1, hipStream_t s);
hipError_t launch_alias(const Workspace &w, const Grid &g, int q1, hipStream_t s);
hipError_t launch_count(const Workspace &w, const Grid &g, size_t nwords, hipStream_t s);
hipError_t launch_heads(const Workspace &w, size_t nwords, u64 totV, u64 totQ, hipStream_t s);
hipError_t launch_finalize(const Workspace &w, const Grid &g, size_t nwords, hipStream_t s);
hipError_t launch_emit_points(const Workspace &w, const Grid &g, const Geo &geo, u64 nV, u32 nVertexWords, hipStream_t s);
hipError_t launch_emit_cells(const Workspace &w, const Grid &g, int triangles, u64 pointOffset, u64 nQ, hipStream_t s);
hipError_t launch_project(int pixel_type, const Workspace &w, const Grid &g, const Geo &geo,
                          const Params &p, u64 nPoints, u64 nGhost, hipStream_t s);
size_t scan_temp_bytes(size_t nseg);
hipError_t launch_scan(void *temp, size_t tempBytes, const u64 *in, u64 *out, size_t n, hipStream_t s);

}  // namespace cuberille
#endif
