"""ctypes loader for the CPU oracle.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this
module (see oracle/cuberille_oracle.h).  It also holds a numpy closed-form
counter used to cross-check the C++ sweep.
"""
import ctypes as C
import math
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libcuberille_oracle.so")

PIXEL_CODES = {
    np.dtype(np.uint8): 0, np.dtype(np.int8): 1, np.dtype(np.uint16): 2, np.dtype(np.int16): 3,
    np.dtype(np.uint32): 4, np.dtype(np.int32): 5, np.dtype(np.float32): 6, np.dtype(np.float64): 7,
    np.dtype(np.int64): 8, np.dtype(np.uint64): 9,
}


class _Image(C.Structure):
    _fields_ = [("pixel_type", C.c_int32), ("dims", C.c_int64 * 3), ("spacing", C.c_double * 3),
                ("origin", C.c_double * 3), ("direction", C.c_double * 9), ("voxels", C.c_void_p),
                ("index_start", C.c_int64 * 3)]


class _Params(C.Structure):
    _fields_ = [("iso_value", C.c_double), ("generate_triangles", C.c_int32), ("project_vertices", C.c_int32),
                ("distance_threshold", C.c_double), ("step_length", C.c_double), ("relaxation", C.c_double),
                ("max_steps", C.c_uint32), ("gradient_threads", C.c_int32), ("faithful_cells", C.c_int32),
                ("projection_variant", C.c_int32), ("gradient_variant", C.c_int32), ("iso_value_int", C.c_int64)]


class _Mesh(C.Structure):
    _fields_ = [("n_points", C.c_uint64), ("n_cells", C.c_uint64), ("verts_per_cell", C.c_int32),
                ("points", C.POINTER(C.c_float)), ("cells", C.POINTER(C.c_uint64)),
                ("seconds_gradient", C.c_double), ("seconds_sweep", C.c_double),
                ("proj_iterations", C.c_uint64), ("proj_stop_threshold", C.c_uint64),
                ("proj_stop_steps", C.c_uint64)]


def build(force=False):
    """Compile the oracle with g++ (oracle/Makefile)."""
    if force or not os.path.exists(_SO) or any(
            os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_SO)
            for f in ("cuberille_oracle.cpp", "cuberille_oracle.h")):
        subprocess.check_call(["make", "-s", "-C", _HERE])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _lib.cuberille_oracle_run.argtypes = [C.POINTER(_Image), C.POINTER(_Params), C.POINTER(_Mesh)]
        _lib.cuberille_oracle_run.restype = C.c_int
        _lib.cuberille_oracle_run_after.argtypes = [C.POINTER(_Image), C.POINTER(_Image), C.POINTER(_Params), C.POINTER(_Mesh)]
        _lib.cuberille_oracle_run_after.restype = C.c_int
        _lib.cuberille_oracle_free.argtypes = [C.POINTER(_Mesh)]
        _lib.cuberille_oracle_interpolate.argtypes = [C.POINTER(_Image), C.POINTER(C.c_double)]
        _lib.cuberille_oracle_interpolate.restype = C.c_double
        _lib.cuberille_oracle_gradient_at_index.argtypes = [C.POINTER(_Image), C.POINTER(C.c_int64), C.POINTER(C.c_float)]
        _lib.cuberille_oracle_index_to_point.argtypes = [C.POINTER(_Image), C.POINTER(C.c_int64), C.POINTER(C.c_float)]
    return _lib


def _image(vol, spacing, origin, direction, index_start=(0, 0, 0)):
    vol = np.ascontiguousarray(vol)
    assert vol.ndim == 3, "volume is indexed [z, y, x]"
    img = _Image()
    img.pixel_type = PIXEL_CODES[vol.dtype]
    nz, ny, nx = vol.shape
    img.dims[:] = [nx, ny, nz]
    img.spacing[:] = list(spacing)
    img.origin[:] = list(origin)
    img.direction[:] = list(np.asarray(direction, dtype=np.float64).reshape(9))
    img.voxels = vol.ctypes.data
    img.index_start[:] = [int(v) for v in index_start]
    return img, vol


class OracleMesh:
    def __init__(self, points, cells, info):
        self.points = points      # float32 [n,3]
        self.cells = cells        # uint64 [m, 3|4]
        self.info = info


def run(vol, iso, triangles=True, project=True, threshold=0.5, step=-1.0, relax=0.95, max_steps=50,
        spacing=(1.0, 1.0, 1.0), origin=(0.0, 0.0, 0.0), direction=np.eye(3), gradient_threads=1,
        faithful_cells=False, variant=0, gradient=0, first=None, index_start=(0, 0, 0)):
    """Run the restated reference sweep on `vol` ([z,y,x] numpy array).  first: None, or the input of the filter object's
    FIRST projecting Update() as (vol, spacing, origin, direction) -- quirk Q3 (txx:484): this update then walks along
    that image's gradient.  variant: 0 the default projection,
    1 / 2 the reference's compiled-out USE_ADVANCED_PROJECTION / USE_LINESEARCH_PROJECTION branches.  gradient: 0 the
    central differences of itk::GradientImageFilter, 1 USE_GRADIENT_RECURSIVE_GAUSSIAN (compiled out upstream too)."""
    img, keep = _image(vol, spacing, origin, direction, index_start)
    # 64-bit integer pixels: the iso value as the C cast of the reference takes it (h:180-181), truncated toward zero
    try:
        iso_int = int(iso) if isinstance(iso, (int, np.integer)) else math.trunc(float(iso))
    except (OverflowError, ValueError):
        iso_int = None
    if keep.dtype in (np.dtype(np.int64), np.dtype(np.uint64)):
        lo, hi = (-(1 << 63), 1 << 63) if keep.dtype == np.dtype(np.int64) else (0, 1 << 64)
        if iso_int is None or not (lo <= iso_int < hi):
            raise ValueError("iso value is not representable in the pixel type")
    iso_int = 0 if iso_int is None else ((iso_int + (1 << 63)) % (1 << 64)) - (1 << 63)   # uint64 above 2^63: the same 64 bits
    prm = _Params(float(iso), int(bool(triangles)), int(bool(project)), float(threshold), float(step),
                  float(relax), int(max_steps), int(gradient_threads), int(bool(faithful_cells)), int(variant), int(gradient), iso_int)
    mesh = _Mesh()
    if first is not None:
        fvol, *fgeo = first if isinstance(first, (tuple, list)) else (first,)
        fgeo = list(fgeo) + [(1.0, 1.0, 1.0), (0.0, 0.0, 0.0), np.eye(3), (0, 0, 0)][len(fgeo):]
        fimg, fkeep = _image(np.asarray(fvol), *fgeo)
        rc = lib().cuberille_oracle_run_after(C.byref(img), C.byref(fimg), C.byref(prm), C.byref(mesh))
        del fkeep
    else:
        rc = lib().cuberille_oracle_run(C.byref(img), C.byref(prm), C.byref(mesh))
    if rc != 0:
        raise ValueError("cuberille_oracle_run failed: %d" % rc)
    try:
        npnt, ncell, vpc = int(mesh.n_points), int(mesh.n_cells), int(mesh.verts_per_cell)
        pts = np.ctypeslib.as_array(mesh.points, shape=(max(npnt * 3, 1),))[:npnt * 3].copy().reshape(npnt, 3)
        cells = np.ctypeslib.as_array(mesh.cells, shape=(max(ncell * vpc, 1),))[:ncell * vpc].copy().reshape(ncell, vpc)
        info = dict(seconds_gradient=mesh.seconds_gradient, seconds_sweep=mesh.seconds_sweep,
                    proj_iterations=int(mesh.proj_iterations), proj_stop_threshold=int(mesh.proj_stop_threshold),
                    proj_stop_steps=int(mesh.proj_stop_steps))
    finally:
        lib().cuberille_oracle_free(C.byref(mesh))
    del keep
    return OracleMesh(pts, cells, info)


def interpolate(vol, point, spacing=(1.0, 1.0, 1.0), origin=(0.0, 0.0, 0.0), direction=np.eye(3), index_start=(0, 0, 0)):
    img, keep = _image(vol, spacing, origin, direction, index_start)
    p = (C.c_double * 3)(*point)
    return lib().cuberille_oracle_interpolate(C.byref(img), p)


def gradient_at_index(vol, idx, spacing=(1.0, 1.0, 1.0), origin=(0.0, 0.0, 0.0), direction=np.eye(3)):
    img, keep = _image(vol, spacing, origin, direction)
    i = (C.c_int64 * 3)(*idx)
    g = (C.c_float * 3)()
    lib().cuberille_oracle_gradient_at_index(C.byref(img), i, g)
    return np.array(list(g), dtype=np.float32)


def index_to_point(vol, idx, spacing=(1.0, 1.0, 1.0), origin=(0.0, 0.0, 0.0), direction=np.eye(3), index_start=(0, 0, 0)):
    """idx: the pixel's position in the buffer ([x, y, z]); its ITK index is idx + index_start."""
    img, keep = _image(vol, spacing, origin, direction, index_start)
    i = (C.c_int64 * 3)(*idx)
    p = (C.c_float * 3)()
    lib().cuberille_oracle_index_to_point(C.byref(img), i, p)
    return np.array(list(p), dtype=np.float32)


def closed_form_counts(vol, iso):
    """(#points, #quads) by the closed form of SURVEY.md section 8a, valid when quirk Q1 does not
    fire (no empty slice between occupied slices): quads = inside voxel x face whose
    clamped neighbour is outside; points = lattice corners whose clamped 2x2x2 block is mixed."""
    ins = np.asarray(vol) >= np.asarray(iso, dtype=np.asarray(vol).dtype)
    quads = 0
    for ax in range(3):
        a = np.moveaxis(ins, ax, 0)
        quads += int(np.count_nonzero(a[1:] & ~a[:-1])) + int(np.count_nonzero(a[:-1] & ~a[1:]))
    p = np.pad(ins, 1, mode="edge")
    blk_and = np.ones(tuple(s + 1 for s in ins.shape), dtype=bool)
    blk_or = np.zeros_like(blk_and)
    for dz in (0, 1):
        for dy in (0, 1):
            for dx in (0, 1):
                s = p[dz:dz + blk_and.shape[0], dy:dy + blk_and.shape[1], dx:dx + blk_and.shape[2]]
                blk_and &= s
                blk_or |= s
    points = int(np.count_nonzero(blk_or & ~blk_and))
    return points, quads
