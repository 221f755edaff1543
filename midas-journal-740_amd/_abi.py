"""ctypes binding of include/cuberille_hip.h (the C-ABI drop-in boundary).

The shared library is built in-tree by csrc/Makefile (hipcc --offload-arch=gfx950).
There is no CPU fallback anywhere in this package: if the library is missing, or no
gfx950 device is usable, the calls raise.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
# CUBERILLE_LIB: load another build of the same ABI (same-box A/B timing of kernel changes).  The one environment variable
# this BINDING reads; the library itself reads none (include/cuberille_hip.h).
LIB_PATH = os.environ.get("CUBERILLE_LIB") or os.path.join(CSRC, "libcuberille_hip.so")

OK, ERR_ARGUMENT, ERR_NO_DEVICE, ERR_HIP, ERR_STATE, ERR_HALO, ERR_LIMIT, ERR_SOURCE, RETRY = range(9)
SLAB_THIN_HALO = 1
ESCAPED_OVERFLOW = 1 << 62          # Extractor.escaped_count(): more walks escaped than the library's list holds

# every symbol include/cuberille_hip.h declares (tests check the built library exports them all)
EXPORTS = [
    "cuberille_abi_version", "cuberille_device_count", "cuberille_last_error", "cuberille_create",
    "cuberille_destroy", "cuberille_set_stream", "cuberille_extract_host", "cuberille_extract_device",
    "cuberille_count", "cuberille_emit", "cuberille_mesh_device", "cuberille_mesh_download",
    "cuberille_debug_bits", "cuberille_slice_occupancy", "cuberille_write_vtk_buffers", "cuberille_mesh_write_vtk",
    "cuberille_required_halo", "cuberille_slab_info", "cuberille_debug_set_option", "cuberille_debug_h2d_seconds",
    "cuberille_extract_stream", "cuberille_emit_points", "cuberille_slice_bits_device", "cuberille_recount", "cuberille_alias_plane_device", "cuberille_set_alias_plane",
    "cuberille_minimum_halo", "cuberille_escaped_count", "cuberille_reproject_escaped", "cuberille_step_begin", "cuberille_step_end",
    "cuberille_slice_counts", "cuberille_failed_row", "cuberille_warm_up", "cuberille_mesh_host", "cuberille_step_classify", "cuberille_step_count",
    "cuberille_release_host_mesh", "cuberille_hold_gradient", "cuberille_gradient_held",
]
ABI_VERSION = 13


class ImageDesc(C.Structure):
    _fields_ = [("pixel_type", C.c_int32), ("dims", C.c_int64 * 3), ("spacing", C.c_double * 3),
                ("origin", C.c_double * 3), ("direction", C.c_double * 9), ("index_start", C.c_int64 * 3)]


class Params(C.Structure):
    _fields_ = [("iso_value", C.c_double), ("generate_triangles", C.c_int32), ("project_vertices", C.c_int32),
                ("distance_threshold", C.c_double), ("step_length", C.c_double), ("relaxation", C.c_double),
                ("max_steps", C.c_uint32), ("emulate_empty_slice_aliasing", C.c_int32),
                ("projection_variant", C.c_int32), ("gradient_variant", C.c_int32), ("iso_value_int", C.c_int64)]


class Slab(C.Structure):
    _fields_ = [("global_nz", C.c_int64), ("z_begin", C.c_int64), ("own_z0", C.c_int64), ("own_z1", C.c_int64),
                ("point_id_offset", C.c_uint64), ("flags", C.c_uint64), ("halo_ready_event", C.c_void_p),
                ("voxels_ready_event", C.c_void_p)]


class SlabStatus(C.Structure):
    _fields_ = [("alias_source_below_buffer", C.c_int32), ("reserved", C.c_int32), ("lowest_occupied_z", C.c_int64),
                ("highest_occupied_z", C.c_int64), ("second_highest_occupied_z", C.c_int64), ("alias_z", C.c_int64)]


class Result(C.Structure):
    _fields_ = [("n_points", C.c_uint64), ("n_cells", C.c_uint64), ("verts_per_cell", C.c_int32),
                ("reserved", C.c_int32), ("ms_classify", C.c_float), ("ms_count", C.c_float),
                ("ms_scan", C.c_float), ("ms_emit_points", C.c_float), ("ms_project", C.c_float),
                ("ms_emit_cells", C.c_float), ("ms_total", C.c_float), ("ms_pass", C.c_float), ("proj_iterations", C.c_uint64),
                ("proj_stop_threshold", C.c_uint64), ("proj_stop_steps", C.c_uint64), ("n_escaped", C.c_uint64)]

    def as_dict(self):
        return {name: getattr(self, name) for name, _ in self._fields_ if name != "reserved"}


# cuberille_chunk_source: int (*)(void *user, void *dst, int64_t z0, int64_t z1)
CHUNK_SOURCE = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64)


class CuberilleError(RuntimeError):
    def __init__(self, code, text):
        RuntimeError.__init__(self, "cuberille error %d: %s" % (code, text))
        self.code = code


def build(force=False):
    """Compile csrc/*.hip for gfx950 with hipcc (works without a GPU)."""
    srcs = [os.path.join(CSRC, f) for f in ("cuberille_kernels.hip", "cuberille_api.hip", "cuberille_vtk.cpp", "cuberille_internal.h")]
    srcs.append(os.path.join(_HERE, "..", "include", "cuberille_hip.h"))
    if not force and os.path.exists(LIB_PATH) and all(os.path.getmtime(s) <= os.path.getmtime(LIB_PATH) for s in srcs):
        return LIB_PATH
    subprocess.check_call(["make", "-s", "-j3", "-C", CSRC])
    return LIB_PATH


_lib = None


def lib():
    """Load the C-ABI library; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    # torch bundles its own HIP runtime (libamdhip64.so.7).  Import it FIRST so that this library
    # binds to the runtime already in the process: two HIP runtimes in one process cannot both
    # see the GPU, and bench.py / the multi-GPU driver need torch tensors and RCCL next to us.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(LIB_PATH):
        raise ImportError("%s is missing: run __graft_entry__.build() (hipcc --offload-arch=gfx950); "
                          "the cuberille hot path has no CPU fallback" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp, u64p = C.c_void_p, C.POINTER(C.c_uint64)
    L.cuberille_abi_version.restype = C.c_int
    L.cuberille_device_count.restype = C.c_int
    L.cuberille_last_error.restype = C.c_char_p
    L.cuberille_last_error.argtypes = [vp]
    L.cuberille_create.argtypes = [C.POINTER(vp), C.c_int]
    L.cuberille_destroy.argtypes = [vp]
    L.cuberille_destroy.restype = None
    L.cuberille_set_stream.argtypes = [vp, vp]
    L.cuberille_extract_host.argtypes = [vp, C.POINTER(ImageDesc), vp, C.POINTER(Params), C.POINTER(Result)]
    L.cuberille_extract_stream.argtypes = [vp, C.POINTER(ImageDesc), CHUNK_SOURCE, vp, C.POINTER(Params), C.POINTER(Result)]
    L.cuberille_extract_device.argtypes = [vp, C.POINTER(ImageDesc), vp, C.POINTER(Params), C.POINTER(Slab),
                                           C.POINTER(Result)]
    L.cuberille_count.argtypes = [vp, C.POINTER(ImageDesc), vp, C.POINTER(Params), C.POINTER(Slab), u64p, u64p]
    L.cuberille_emit.argtypes = [vp, C.c_uint64, C.POINTER(Result)]
    L.cuberille_emit_points.argtypes = [vp]
    L.cuberille_mesh_device.argtypes = [vp, C.POINTER(vp), C.POINTER(vp)]
    L.cuberille_mesh_download.argtypes = [vp, vp, vp]
    L.cuberille_mesh_host.argtypes = [vp, C.POINTER(vp), C.POINTER(vp)]
    L.cuberille_release_host_mesh.argtypes = [vp]
    L.cuberille_hold_gradient.argtypes = [vp, C.c_int]
    L.cuberille_gradient_held.argtypes = [vp, C.POINTER(C.c_int64)]
    L.cuberille_debug_bits.argtypes = [vp, vp, C.c_size_t]
    L.cuberille_slice_occupancy.argtypes = [vp, vp, C.c_size_t]
    L.cuberille_write_vtk_buffers.argtypes = [C.c_char_p, vp, C.c_uint64, vp, C.c_uint64, C.c_int, C.c_int]
    L.cuberille_mesh_write_vtk.argtypes = [vp, C.c_char_p, C.c_int]
    L.cuberille_required_halo.argtypes = [C.POINTER(ImageDesc), C.POINTER(Params), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    L.cuberille_slab_info.argtypes = [vp, C.POINTER(SlabStatus)]
    L.cuberille_minimum_halo.argtypes = [C.POINTER(ImageDesc), C.POINTER(Params), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    L.cuberille_escaped_count.argtypes = [vp, u64p]
    L.cuberille_reproject_escaped.argtypes = [vp, vp, C.c_int64, C.c_int64]
    L.cuberille_step_begin.argtypes = [vp, C.POINTER(ImageDesc), vp, C.POINTER(Params), C.POINTER(Slab), C.POINTER(vp),
                                       C.POINTER(C.c_size_t)]
    L.cuberille_step_classify.argtypes = [vp, C.POINTER(ImageDesc), vp, C.POINTER(Params), C.POINTER(Slab), C.POINTER(vp),
                                          C.POINTER(C.c_size_t)]
    L.cuberille_step_count.argtypes = [vp, vp, vp, C.POINTER(vp), C.POINTER(C.c_size_t)]
    L.cuberille_step_end.argtypes = [vp, vp, C.c_int, C.c_int, C.POINTER(Result)]
    L.cuberille_slice_counts.argtypes = [vp, vp, vp, C.c_size_t]
    L.cuberille_warm_up.argtypes = [vp, C.POINTER(ImageDesc), C.POINTER(Params)]
    L.cuberille_failed_row.argtypes = [vp, C.c_size_t, C.POINTER(C.c_size_t)]
    L.cuberille_debug_set_option.argtypes = [vp, C.c_char_p, C.c_int64]
    L.cuberille_debug_h2d_seconds.argtypes = [vp, C.c_size_t, C.POINTER(C.c_double)]
    L.cuberille_slice_bits_device.argtypes = [vp, C.c_int64, C.POINTER(vp), C.POINTER(C.c_size_t)]
    L.cuberille_recount.argtypes = [vp, vp, u64p, u64p]
    L.cuberille_alias_plane_device.argtypes = [vp, C.c_int64, vp, vp]
    L.cuberille_set_alias_plane.argtypes = [vp, vp, vp]
    _lib = L
    return L


def failed_row():
    """The row a rank whose cuberille_step_begin failed contributes to the all-gather of the rows (uint8 numpy array)."""
    import numpy as np
    n = C.c_size_t()
    lib().cuberille_failed_row(None, 0, C.byref(n))
    row = np.zeros(int(n.value), dtype=np.uint8)
    rc = lib().cuberille_failed_row(C.c_void_p(row.ctypes.data), row.nbytes, C.byref(n))
    if rc != OK:
        raise CuberilleError(rc, "cuberille_failed_row")
    return row


def check(ctx, rc):
    if rc != OK:
        text = lib().cuberille_last_error(ctx)
        raise CuberilleError(rc, text.decode("utf-8", "replace") if text else "")
