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
  int istart[3];      // index of the first buffered pixel (itk::ImageRegion::GetIndex): ITK's index <-> point transforms and
                      // its interpolators work on INDICES = position + start.  (Only these three scalars: the walk kernel
                      // keeps the whole struct in SGPRs and spills what does not fit into its hot loop.)
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
  int cmapLinear;      // 0: the corner map bricked (corner_map_index); 1: row-major (development switch)
  int extAlias;        // 1: slice nzb of the bit volume (one past the buffer) holds the inside bits of the occupied slice
                       //    the rank below reported, the source of quirk Q1's vertex re-use for this slab's first
                       //    occupied slice (cuberille_recount)
};

// Is this context one rank of several (its row then carries the three slices the ranks above judge their flags by)?  NOT
// "the buffer is shorter than the volume": with a halo as deep as the rest of the volume a rank's BUFFER is the whole volume
// while it owns a part of it (round 5, tests/fuzz_ranks.py: a rank that owned slices 0-10 of 13 said "no occupied slice" and
// the rank two above kept a count that quirk Q1 should have changed).
__host__ __device__ inline bool part_of_a_volume(const Grid &g) {
  return g.gnz != (long long)g.nzb || g.oz0 != 0 || g.oz1 != g.nzb;
}

struct Totals {        // device-resident, zeroed before every count, mirrored to pinned host memory
  u64 totV, totQ;      // created vertices / quads in the counted range
  u64 V0, Q0;          // of which before the first owned slice
  u64 iters;           // projection iterations (atomic)
  u64 g0pre;           // in-block prefix (V | Q << 32) at the first owned word, left by the count block that holds it
  u32 err;             // device-side error flags
  u32 nVertexWords;    // entries in the vertex-word queue (words that create at least one vertex)
  u32 nEscaped;        // THIN_HALO slabs: vertices whose walk asked for a slice the buffer lacks (entries of the escape list,
                       // or more than it holds: ERRF_ESCAPE_OVERFLOW)
  u32 go;              // cuberille_step_begin: 1 when the kernels launched blindly behind the count may run (k_gate)
  u64 stopThr, stopSteps;   // walks of owned vertices that ended within the threshold / out of steps (txx:457-459, 470-472)
  // what the ranks need of each other to judge an ERRF_ALIAS_BELOW_BUFFER without the host (row_flags; global slices,
  // -1: none; written by k_block_scan): the first occupied slice of the counted range -- the one such a flag is about --
  // and the highest and second-highest occupied OWNED slices (cuberille_slab_status has the same three for the host)
  int aliasZ, topZ, top2Z;
  u32 ticket;          // count blocks that have published their totals (the last one scans them where the count kernel does
                       // the block scan itself: launches whose blocks are all resident at once)
};

enum {
  ERRF_ALIAS_UNKNOWN = 1,        // quirk Q1: the aliased source slice is in the buffer but below the counted range
  ERRF_ALIAS_BELOW_BUFFER = 2,   // quirk Q1: the search for the source slice ran off the bottom of a slab buffer
  ERRF_CAPACITY = 4,             // cuberille_step_begin: the counts exceed what the blind launches / buffers were sized for
  ERRF_ESCAPE = 8,               // THIN_HALO: at least one walk left the buffer (Totals::nEscaped)
  ERRF_ESCAPE_OVERFLOW = 16,     // ... and more of them than the escape list holds
  ERRF_RANK_FAILED = 32          // never raised by a kernel: the row a driver contributes for a rank whose
                                 // cuberille_step_begin failed (cuberille_failed_row)
};

// The flags of rank r's row as the step sees them (k_emit_cells on the device, cuberille_step_end on the host: every
// rank decides alike from the same gathered rows).  ERRF_ALIAS_BELOW_BUFFER says "my first occupied slice, aliasZ, has only
// empty slices below it in my buffer: I counted as if nothing were occupied further down" -- which is TRUE, and no flag at
// all, unless some rank below owns an occupied slice strictly below aliasZ (the rule of the driver's alias_plan: a rank's
// highest occupied slice, or its second highest when the highest IS aliasZ -- the ghost slice of the rank above).
__host__ __device__ inline u32 row_flags(const Totals *rows, int r) {
  u32 e = rows[r].err;
  if (e & (u32)ERRF_ALIAS_BELOW_BUFFER) {
    const int az = rows[r].aliasZ;
    bool source = az < 0;                    // (a row that does not say what the flag is about: leave it standing)
    for (int s = 0; s < r; s++) {
      const int h = rows[s].topZ < az ? rows[s].topZ : rows[s].top2Z;
      if (h >= 0 && h < az) source = true;
    }
    if (!source) e &= ~(u32)ERRF_ALIAS_BELOW_BUFFER;
  }
  return e;
}

// A count block owns COUNT_WB consecutive words of the flat raster order = 32 scan segments of 64 words.
// Absolute exclusive prefix of word gi = blockBase[gi >> COUNT_LG] + segPre[gi >> 6] + prefix[gi].
constexpr unsigned ESCAPE_LIST_CAP = 1u << 20;   // entries of the THIN_HALO escape list (4 MiB)
constexpr int COUNT_LG = 11;
constexpr int COUNT_WB = 1 << COUNT_LG;

// Quirk Q3 reproduced on request (cuberille_hold_gradient): the float gradient image of the volume of a context's FIRST
// projecting extraction and that volume's geometry -- what the reference's cached gradient interpolator goes on
// evaluating for the life of the filter object (txx:484).  img == null: none held, the walk evaluates the current volume.
struct HeldGradient {
  const float *img;    // CovariantVector<float,3> per pixel, x fastest
  Geo geo;
  int n[3];
};

struct Workspace {     // device pointers valid for one count/emit pair
  const void *vox;
  u64 *bits;
  u64 *flatBits;       // scratch for rows that are not whole words: inside bits in flat voxel order (or null)
  u32 *sliceOcc;       // per buffer slice: does it hold an inside voxel (quirk Q1 needs it)
  u32 *prefix;         // per counted word: exclusive in-segment prefix, V | Q<<16
  u64 *segPre;         // per 64-word segment: exclusive in-block prefix, V | Q<<32
  u64 *blockTot;       // per count block: its totals V | Q<<32
  u64 *blockBase;      // per count block: absolute exclusive prefixes, [2b] = V, [2b+1] = Q (k_block_scan)
  Totals *totals;
  float *points;
  u64 *cells;
  u32 *cmap;           // dense lattice-corner -> vertex index map (null: recompute ids instead)
  u32 *headV, *headQ;  // word that produces output 64*i (null: per-lane binary search instead)
  u32 *vqueue;         // counted-range indices of the words that create vertices, in no particular order (or null)
  double *gradImg;     // gradient_variant 1: the recursive-Gaussian gradient image, 3 doubles per voxel (else null)
  float *rgA, *rgB;    //   ... and what its passes go through: two float volumes and a double one
  double *rgScratch;
  u32 *escList;        // THIN_HALO: indices (in this rank's point buffer) of the vertices whose walk left the buffer
  u32 escCap;
  const HeldGradient *held;   // (host pointer) the held gradient image the walk follows instead of the volume's own, or null
};

// Development switches, set per context through cuberille_debug_set_option (never read from the environment).
// The defaults are the measured best; the parity tests flip the fallbacks on to cover them.
struct Tuning {
  int no_cmap = 0, no_heads = 0, no_vqueue = 0, no_stream_classify = 0;   // drop a scratch table / the flat-stream path
  int classify_variant = 0;   // 0: staged spans with write-through stores where the volume is large, 1: always the plain sweep
  int classify_grid = 0;      // workgroups of the sweep (0 = default)
  int classify_keep_tail = 0; // 1: a partly filled last round of spans stays with the span kernel (A/B of the balancing)
  int proj_chunk64_below = 0; // vertices under which the walk deals batches of 64 (0: the default, 8 M)
  int points_no_split = 0;    // 1: the point pass runs one lane per vertex word however short the queue
  int points_split = 0;       // lanes per vertex word of the point pass: 1, 2, 4 or 8 (0: by the length of the queue)
  int points_variant = 3;     // 3 dense two-phase, 2 queue walk, 1 wave-window search, 0 block form
  int cmap_linear = 0;        // 0: the corner map in 4 x 4 x 2 bricks of one 128-byte line, 1: row-major as in round 2
  int count_variant = -1;     // 1: the count kernel reads its bit rows from an LDS tile (in columns of 8 blocks where slices are
                              // whole blocks), 2: the tile, one block per workgroup, 4 .. 31: columns of that many, 0: from memory,
                              // 3 / 32 + one of those: the dense form (k_count_dense: one phase, corner logic per lattice corner,
                              // pipelined columns) where rows are a power of two of words, else the tile as 1 / that,
                              // -1: the dense form when the previous extraction on the context found vertices in a quarter of its
                              //     words, else from memory
  // the walk: vertices per batch (0: 64 when the launch leaves wave slots empty, else 128), waves in the grid, idle lanes
  // at which a wave refills (0: 16; 64 = only when empty, where the previous extraction's walks took under four passes
  // per vertex), XCD-contiguous batches, the reference's interpolation loop to the letter on every pass
  // (proj_waves 0: 16 384, or 65 536 waves dealing batches of 64 where walks are short -- see launch_project)
  int proj_chunk = 0, proj_waves = 0, proj_refill = 0, proj_xcd = 0, proj_literal = 0;
  int proj_ident = -1;        // 1 / 0: the walk kernel specialised for identity geometry where the geometry allows / never (A/B); -1: the default
  int proj_short = -1;        // the launch shapes of SHORT walks (1) or of long ones (0); -1: by the previous extraction's passes per vertex
  int count_no_fold = 0;      // 1: the block scan always as a launch of its own (A/B of the scan folded into small count launches)
  int stage_timing = 0;       // 1: events between the stages too (cuberille_result::ms_classify ... ms_emit_cells)
};

struct Params {
  double iso;
  long long isoInt;           // the iso value of the 64-bit integer pixel types (a double cannot hold it past 2^53)
  double thr, step, relax;
  u32 max_steps;
  int triangles, project, q1;
  int variant;                // CUBERILLE_PROJECT_*
  int gradVariant;            // CUBERILLE_GRADIENT_*
};

// cuberille_step_begin: what the launches behind the count were sized for (k_block_scan sets Totals::go accordingly)
struct Gate {
  int on;
  u64 coverV, coverQ;
  u32 coverVW;
};

// launchers (cuberille_kernels.hip); all asynchronous on `s`.  dyn: the launch is sized for an estimate, the kernel
// reads the real sizes from the device totals and runs only when Totals::go says so.
hipError_t launch_classify(int pixel_type, const Workspace &w, const Grid &g, const Params &p, int z0, int z1, const Tuning &t,
                           hipStream_t s);
hipError_t launch_occupancy(int pixel_type, const Workspace &w, const Grid &g, const Tuning &t, hipStream_t s);
hipError_t launch_occupancy_range(const Workspace &w, const Grid &g, int z0, int z1, hipStream_t s);
hipError_t launch_count(const Workspace &w, const Grid &g, size_t nwords, int q1, const Gate &gate, int tiled, int noFold, hipStream_t s);
hipError_t launch_heads(const Workspace &w, const Grid &g, u64 totV, u64 totQ, int dyn, hipStream_t s);
hipError_t launch_emit_points(const Workspace &w, const Grid &g, const Geo &geo, int q1, u64 nV, u32 nVertexWords,
                              const Tuning &t, int dyn, hipStream_t s);
hipError_t launch_emit_cells(const Workspace &w, const Grid &g, int triangles, int q1, u64 pointOffset, u64 nQ,
                             const u64 *extIds, const Totals *rows, int nRanks, int rank, int dyn, hipStream_t s);
hipError_t launch_slice_prefix(const Workspace &w, const Grid &g, u64 *out, hipStream_t s);
hipError_t launch_density_probe(const Workspace &w, const Grid &g, size_t nwords, u64 *out, hipStream_t s);
hipError_t launch_alias_plane(const Workspace &w, const Grid &g, int zLocal, u64 pointOffset, u64 *idsOut, float *ptsOut,
                              hipStream_t s);
hipError_t launch_recursive_gaussian(int pixel_type, const Workspace &w, const Grid &g, const Geo &geo, const double coef[3][2][20],
                                     hipStream_t s);
hipError_t launch_gradient_image(int pixel_type, const Workspace &w, const Grid &g, const Geo &geo, float *out, hipStream_t s);
hipError_t launch_project(int pixel_type, const Workspace &w, const Grid &g, const Geo &geo,
                          const Params &p, u64 nPoints, u64 nGhost, const Tuning &t, int mode, int dyn, hipStream_t s);

}  // namespace cuberille
#endif
