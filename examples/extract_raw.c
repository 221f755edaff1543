/* extract_raw.c -- the C ABI of include/cuberille_hip.h from plain C (C99, no C++ runtime on this side).
 *
 *   extract_raw <volume.raw> <nx> <ny> <nz> <u8|u16|f32> <iso> <out.vtk> [quads]
 *
 * Reads a headerless little-endian volume (x fastest) through cuberille_extract_stream -- fread() fills the
 * library's pinned staging slots chunk by chunk while the previous chunk is uploaded and thresholded -- with the
 * reference's default parameters (Testing/CuberilleTest01.cxx:98-109: triangles, projection, threshold 0.5, step 0.25,
 * relaxation 0.95, 50 steps) and writes the mesh as legacy VTK polydata.
 * Build:  gcc -std=c99 -O2 -Iinclude examples/extract_raw.c -Lmidas-journal-740_amd/csrc -lcuberille_hip \
 *             -Wl,-rpath,$PWD/midas-journal-740_amd/csrc -o extract_raw
 * Exit status: 0 ok, 1 usage / file trouble, 2 no MI355X device, 3 library error. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "cuberille_hip.h"

struct reader {
  FILE *f;
  size_t slice_bytes;
};

static int read_slices(void *user, void *dst, int64_t z0, int64_t z1) {
  struct reader *r = (struct reader *)user;
  const size_t want = (size_t)(z1 - z0) * r->slice_bytes;
  return fread(dst, 1, want, r->f) == want ? 0 : 1;          /* non-zero: CUBERILLE_ERR_SOURCE */
}

int main(int argc, char **argv) {
  if (argc < 8) {
    fprintf(stderr, "usage: %s volume.raw nx ny nz u8|u16|f32 iso out.vtk [quads]\n", argv[0]);
    return 1;
  }
  if (cuberille_abi_version() != CUBERILLE_ABI_VERSION) {
    fprintf(stderr, "header is ABI %d, library is ABI %d\n", CUBERILLE_ABI_VERSION, cuberille_abi_version());
    return 3;
  }
  if (cuberille_device_count() < 1) {
    fprintf(stderr, "no gfx950 device: the cuberille path has no CPU fallback\n");
    return 2;
  }
  cuberille_image_desc img;
  memset(&img, 0, sizeof img);
  size_t pixel = 0;
  if (!strcmp(argv[5], "u8")) { img.pixel_type = CUBERILLE_PIX_U8; pixel = 1; }
  else if (!strcmp(argv[5], "u16")) { img.pixel_type = CUBERILLE_PIX_U16; pixel = 2; }
  else if (!strcmp(argv[5], "f32")) { img.pixel_type = CUBERILLE_PIX_F32; pixel = 4; }
  else { fprintf(stderr, "pixel type must be u8, u16 or f32\n"); return 1; }
  for (int i = 0; i < 3; i++) {
    img.dims[i] = atoll(argv[2 + i]);
    img.spacing[i] = 1.0;
    img.direction[4 * i] = 1.0;
  }
  cuberille_params prm;
  memset(&prm, 0, sizeof prm);
  prm.iso_value = atof(argv[6]);
  prm.generate_triangles = !(argc > 8 && !strcmp(argv[8], "quads"));
  prm.project_vertices = 1;
  prm.distance_threshold = 0.5;
  prm.step_length = 0.25;
  prm.relaxation = 0.95;
  prm.max_steps = 50;
  prm.emulate_empty_slice_aliasing = 1;
  prm.projection_variant = CUBERILLE_PROJECT_DEFAULT;

  struct reader rd;
  rd.f = fopen(argv[1], "rb");
  rd.slice_bytes = (size_t)img.dims[0] * (size_t)img.dims[1] * pixel;
  if (!rd.f) { perror(argv[1]); return 1; }

  cuberille_ctx *ctx = NULL;
  int rc = cuberille_create(&ctx, 0);
  if (rc != CUBERILLE_OK) {
    fprintf(stderr, "cuberille_create: %s\n", cuberille_last_error(NULL));
    fclose(rd.f);
    return rc == CUBERILLE_ERR_NO_DEVICE ? 2 : 3;
  }
  cuberille_result res;
  rc = cuberille_extract_stream(ctx, &img, read_slices, &rd, &prm, &res);
  fclose(rd.f);
  if (rc == CUBERILLE_OK) rc = cuberille_mesh_write_vtk(ctx, argv[7], 0);
  if (rc != CUBERILLE_OK) {
    fprintf(stderr, "cuberille error %d: %s\n", rc, cuberille_last_error(ctx));
    cuberille_destroy(ctx);
    return 3;
  }
  printf("Mesh has %llu vertices and %llu cells (%.3f ms on the device)\n", (unsigned long long)res.n_points,
         (unsigned long long)res.n_cells, (double)res.ms_total);
  cuberille_destroy(ctx);
  return 0;
}
