"""Python host side of the MI355X cuberille path.

`CuberilleImageToMeshFilter` mirrors the public surface of the reference class
(/root/reference/Source/itkCuberilleImageToMeshFilter.h:180-228: the same
Set/Get names, defaults of txx:33-40, clamps of h:210,216,223, On/Off helpers,
Update()/GetOutput()) so the parity tests read like the reference's own driver
(/root/reference/Testing/CuberilleTest01.cxx:144-162).  `Extractor` is the thin
layer over the C ABI (include/cuberille_hip.h) used by bench.py and the multi-GPU
driver.  Everything computes on the GPU through libcuberille_hip.so; nothing here
falls back to a CPU implementation.
"""
import collections
import ctypes as C
import math
import os

import numpy as np

from . import _abi
from .mha import Volume

PIXEL_CODES = {
    np.dtype(np.uint8): 0, np.dtype(np.int8): 1, np.dtype(np.uint16): 2, np.dtype(np.int16): 3,
    np.dtype(np.uint32): 4, np.dtype(np.int32): 5, np.dtype(np.float32): 6, np.dtype(np.float64): 7,
    np.dtype(np.int64): 8, np.dtype(np.uint64): 9,
}
PIXEL_DTYPES = {code: dt for dt, code in PIXEL_CODES.items()}


def _torch():
    import torch
    return torch


_TORCH_TO_NP = None


def _np_dtype_of(t):
    global _TORCH_TO_NP
    torch = _torch()
    if _TORCH_TO_NP is None:
        _TORCH_TO_NP = {torch.uint8: np.uint8, torch.int8: np.int8, torch.int16: np.int16, torch.int32: np.int32,
                        torch.int64: np.int64, torch.float32: np.float32, torch.float64: np.float64}
        for name, npd in (("uint16", np.uint16), ("uint32", np.uint32), ("uint64", np.uint64)):
            if hasattr(torch, name):
                _TORCH_TO_NP[getattr(torch, name)] = npd
    return np.dtype(_TORCH_TO_NP[t.dtype])


class Mesh:
    """Flat mesh: points float32 [n,3]; cells uint64 [m,3|4] of GLOBAL point ids."""

    def __init__(self, points, cells, point_id_offset=0):
        self.points = points
        self.cells = cells
        self.point_id_offset = point_id_offset

    def GetNumberOfPoints(self):
        return int(self.points.shape[0])

    def GetNumberOfCells(self):
        return int(self.cells.shape[0])

    def write_vtk(self, path, threads=0):
        """Legacy-ASCII VTK POLYDATA straight from the flat buffers (include/cuberille_hip.h:
        cuberille_write_vtk_buffers), the bytes itk::VTKPolyDataWriter gives at CuberilleTest01.cxx:180-187."""
        pts = np.ascontiguousarray(self.points, dtype=np.float32)
        cells = np.ascontiguousarray(self.cells, dtype=np.uint64)
        rc = _abi.lib().cuberille_write_vtk_buffers(os.fsencode(path), C.c_void_p(pts.ctypes.data), pts.shape[0],
                                                    C.c_void_p(cells.ctypes.data), cells.shape[0],
                                                    int(cells.shape[1]) if cells.ndim == 2 else 3, int(threads))
        if rc != _abi.OK:
            raise _abi.CuberilleError(rc, "cannot write %s" % path)


SlabInfo = collections.namedtuple("SlabInfo", "alias_below lowest highest second_highest alias_z")

PROJECT_DEFAULT, PROJECT_ADVANCED, PROJECT_LINESEARCH = 0, 1, 2    # include/cuberille_hip.h CUBERILLE_PROJECT_*


GRADIENT_CENTRAL, GRADIENT_RECURSIVE_GAUSSIAN = 0, 1               # include/cuberille_hip.h CUBERILLE_GRADIENT_*


def make_params(iso, triangles=True, project=True, threshold=0.5, step=-1.0, relax=0.95, max_steps=50, q1=True,
                variant=PROJECT_DEFAULT, gradient=GRADIENT_CENTRAL):
    """variant picks the branch of ProjectVertexToIsoSurface: the shipped one (txx:439-474) or one of the two the
    reference compiles out (USE_ADVANCED_PROJECTION txx:340-397, USE_LINESEARCH_PROJECTION txx:398-437; h:22-23).
    gradient: the shipped central differences, or USE_GRADIENT_RECURSIVE_GAUSSIAN (h:21; txx:488-491; parity unpinned)."""
    # The 64-bit integer pixel types take the iso value as an integer (a double cannot hold it past 2^53): converted like
    # the C cast the reference makes of it (m_IsoSurfaceValue IS an InputPixelType, h:180-181), i.e. truncated toward
    # zero -- 100.5 is 100, as for every other integer pixel type.  A value no 64-bit type can hold (NaN, an infinity,
    # beyond -2^63 .. 2^64 - 1) leaves `iso_int_exact` None, and the entry points refuse it for those pixel types.
    exact = iso_int_of(iso)
    iso_int = 0 if exact is None else ((exact + (1 << 63)) % (1 << 64)) - (1 << 63)   # uint64 above 2^63: the same 64 bits
    prm = _abi.Params(float(iso), int(bool(triangles)), int(bool(project)), float(threshold), float(step),
                      float(relax), int(max_steps), int(bool(q1)), int(variant), int(gradient), iso_int)
    prm.iso_int_exact = exact
    return prm


def iso_int_of(iso):
    """The iso value as the integer a C cast to a 64-bit integer pixel type gives: truncated toward zero; None when no
    such type holds it."""
    try:
        v = int(iso) if isinstance(iso, (int, np.integer)) else math.trunc(float(iso))
    except (OverflowError, ValueError):          # inf, NaN
        return None
    return v if -(1 << 63) <= v < (1 << 64) else None


def check_iso(pixel_type, params):
    """64-bit integer pixels: the iso value must be one the pixel type holds (the other integer types are range-checked
    by the library against iso_value itself).  Parameters not made by make_params are taken as they are."""
    if pixel_type not in (8, 9) or not hasattr(params, "iso_int_exact"):
        return
    v = params.iso_int_exact
    lo, hi = (-(1 << 63), 1 << 63) if pixel_type == 8 else (0, 1 << 64)
    if v is None or not (lo <= v < hi):
        raise _abi.CuberilleError(_abi.ERR_ARGUMENT, "iso value is not representable in the pixel type")


def make_desc(np_dtype, dims_xyz, spacing=(1.0, 1.0, 1.0), origin=(0.0, 0.0, 0.0), direction=None, index_start=(0, 0, 0)):
    d = _abi.ImageDesc()
    d.pixel_type = PIXEL_CODES[np.dtype(np_dtype)]
    d.dims[:] = [int(v) for v in dims_xyz]
    d.spacing[:] = [float(v) for v in spacing]
    d.origin[:] = [float(v) for v in origin]
    dm = np.eye(3) if direction is None else np.asarray(direction, dtype=np.float64)
    d.direction[:] = [float(v) for v in dm.reshape(9)]
    d.index_start[:] = [int(v) for v in index_start]        # (x, y, z) of GetBufferedRegion().GetIndex(); 0 for files
    return d


class Extractor:
    """One context (stream + workspace) on one GPU."""

    def __init__(self, device=0):
        self._lib = _abi.lib()
        self._ctx = C.c_void_p()
        rc = self._lib.cuberille_create(C.byref(self._ctx), int(device))
        if rc != _abi.OK:
            text = self._lib.cuberille_last_error(None)
            self._ctx = C.c_void_p()
            raise _abi.CuberilleError(rc, text.decode() if text else "")
        self.device = int(device)
        self.result = None

    def close(self):
        if getattr(self, "_ctx", None) is not None and self._ctx:
            self._lib.cuberille_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def warm_up(self, desc=None, params=None):
        """cuberille_warm_up: load the code objects and, with an image description, reserve the workspace for such a volume
        -- what the first extraction on a fresh context would otherwise pay inside its own call."""
        _abi.check(self._ctx, self._lib.cuberille_warm_up(self._ctx, C.byref(desc) if desc is not None else None,
                                                          C.byref(params) if params is not None else None))

    def use_torch_stream(self):
        """Order this context's work on torch's current stream."""
        torch = _torch()
        s = torch.cuda.current_stream(self.device).cuda_stream
        _abi.check(self._ctx, self._lib.cuberille_set_stream(self._ctx, C.c_void_p(s)))

    def use_stream(self, torch_stream):
        """Order this context's work on a torch.cuda.Stream of the caller's."""
        _abi.check(self._ctx, self._lib.cuberille_set_stream(self._ctx, C.c_void_p(torch_stream.cuda_stream)))

    def use_own_stream(self):
        """Back to the context's own stream (the default)."""
        _abi.check(self._ctx, self._lib.cuberille_set_stream(self._ctx, C.c_void_p(0)))

    # -- whole-path entry points -----------------------------------------------------------
    def extract_host(self, vol, params):
        """vol: mha.Volume in host memory.  Upload + extract (PCIe-inclusive)."""
        vox = np.ascontiguousarray(vol.voxels)
        desc = make_desc(vox.dtype, vol.dims, vol.spacing, vol.origin, vol.direction, getattr(vol, "index_start", (0, 0, 0)))
        check_iso(int(desc.pixel_type), params)
        res = _abi.Result()
        _abi.check(self._ctx, self._lib.cuberille_extract_host(
            self._ctx, C.byref(desc), C.c_void_p(vox.ctypes.data), C.byref(params), C.byref(res)))
        self.result = res
        return res

    def extract_stream(self, desc, source, params):
        """The upload pipeline fed by a producer (include/cuberille_hip.h: cuberille_extract_stream).  source(dst, z0, z1)
        fills the writable numpy array dst ([z1-z0, Ny, Nx], pinned staging memory of the library) with slices [z0, z1)
        -- e.g. mha.open_stream(path), which inflates a compressed MetaImage stretch by stretch.  An exception raised
        by source ends the call (CuberilleError ERR_SOURCE, the exception chained as its cause)."""
        nx, ny, _ = (int(v) for v in desc.dims)
        dtype = np.dtype(PIXEL_DTYPES[int(desc.pixel_type)])
        check_iso(int(desc.pixel_type), params)
        raised = []

        def trampoline(_user, dst, z0, z1):
            try:
                n = (z1 - z0) * ny * nx
                buf = (C.c_char * (n * dtype.itemsize)).from_address(dst)
                source(np.frombuffer(buf, dtype=dtype).reshape(z1 - z0, ny, nx), int(z0), int(z1))
                return 0
            except BaseException as e:      # noqa: BLE001 -- must not unwind through the C frame
                raised.append(e)
                return 1

        res = _abi.Result()
        rc = self._lib.cuberille_extract_stream(self._ctx, C.byref(desc), _abi.CHUNK_SOURCE(trampoline), None,
                                                C.byref(params), C.byref(res))
        if rc == _abi.ERR_SOURCE and raised:
            text = self._lib.cuberille_last_error(self._ctx)
            raise _abi.CuberilleError(rc, text.decode("utf-8", "replace") if text else "") from raised[0]
        _abi.check(self._ctx, rc)
        self.result = res
        return res

    def extract_mha(self, path, params):
        """File -> mesh: the MetaImage at `path` is read (and inflated) stretch by stretch straight into the upload
        pipeline; returns (result, stream info with dims / spacing / origin / direction)."""
        from .mha import open_stream
        with open_stream(path) as st:
            desc = make_desc(st.dtype, st.dims, st.spacing, st.origin, st.direction)
            return self.extract_stream(desc, st, params), st

    def extract_device(self, dev_ptr, desc, params, slab=None):
        check_iso(int(desc.pixel_type), params)
        res = _abi.Result()
        _abi.check(self._ctx, self._lib.cuberille_extract_device(
            self._ctx, C.byref(desc), C.c_void_p(dev_ptr), C.byref(params),
            C.byref(slab) if slab is not None else None, C.byref(res)))
        self.result = res
        return res

    def count(self, dev_ptr, desc, params, slab=None):
        check_iso(int(desc.pixel_type), params)
        npnt, ncell = C.c_uint64(), C.c_uint64()
        _abi.check(self._ctx, self._lib.cuberille_count(
            self._ctx, C.byref(desc), C.c_void_p(dev_ptr), C.byref(params),
            C.byref(slab) if slab is not None else None, C.byref(npnt), C.byref(ncell)))
        return int(npnt.value), int(ncell.value)

    def emit_points(self):
        """Start the offset-free part of the emit (vertex scatter, projection) right after count(); returns at once."""
        _abi.check(self._ctx, self._lib.cuberille_emit_points(self._ctx))

    def emit(self, point_id_offset=0):
        res = _abi.Result()
        _abi.check(self._ctx, self._lib.cuberille_emit(self._ctx, int(point_id_offset), C.byref(res)))
        self.result = res
        return res

    def step_begin(self, dev_ptr, desc, params, slab=None):
        """Count and the offset-free part of the emit, launched back to back without waiting (cuberille_step_begin).
        Returns (device pointer, bytes) of this rank's row, to be all-gathered in rank order."""
        check_iso(int(desc.pixel_type), params)
        p, n = C.c_void_p(), C.c_size_t()
        _abi.check(self._ctx, self._lib.cuberille_step_begin(
            self._ctx, C.byref(desc), C.c_void_p(dev_ptr), C.byref(params),
            C.byref(slab) if slab is not None else None, C.byref(p), C.byref(n)))
        return p.value, int(n.value)

    def step_classify(self, dev_ptr, desc, params, slab=None):
        """First half of step_begin for the bits-first halo (cuberille_step_classify): thresholds the owned slices and
        returns (device pointer of the buffer's bit volume, words per slice) without waiting."""
        check_iso(int(desc.pixel_type), params)
        p, n = C.c_void_p(), C.c_size_t()
        _abi.check(self._ctx, self._lib.cuberille_step_classify(
            self._ctx, C.byref(desc), C.c_void_p(dev_ptr), C.byref(params),
            C.byref(slab) if slab is not None else None, C.byref(p), C.byref(n)))
        return p.value, int(n.value)

    def step_count(self, halo_bits_event=None, halo_voxels_event=None):
        """Second half (cuberille_step_count): the count behind halo_bits_event, the walk behind halo_voxels_event (raw
        hipEvent_t handles, e.g. torch.cuda.Event.cuda_event; None: nothing to wait for).  Returns the row like step_begin."""
        p, n = C.c_void_p(), C.c_size_t()
        _abi.check(self._ctx, self._lib.cuberille_step_count(
            self._ctx, C.c_void_p(halo_bits_event or 0), C.c_void_p(halo_voxels_event or 0), C.byref(p), C.byref(n)))
        return p.value, int(n.value)

    def step_end(self, dev_rows, n_ranks, rank):
        """The cells with the id offset computed on the device from the gathered rows, then the one wait of the step.
        Returns (result, done): done is False (CUBERILLE_RETRY) when a flag on some rank sends every rank to the
        synchronous calls; the result then only holds the counts."""
        res = _abi.Result()
        rc = self._lib.cuberille_step_end(self._ctx, C.c_void_p(dev_rows), int(n_ranks), int(rank), C.byref(res))
        if rc == _abi.RETRY:
            return res, False
        _abi.check(self._ctx, rc)
        self.result = res
        return res, True

    def slice_counts(self, n_slices):
        """(points, quads) created / emitted by every owned slice of the last count: two uint64 arrays."""
        pts = np.empty(n_slices, dtype=np.uint64)
        quads = np.empty(n_slices, dtype=np.uint64)
        _abi.check(self._ctx, self._lib.cuberille_slice_counts(self._ctx, C.c_void_p(pts.ctypes.data),
                                                                C.c_void_p(quads.ctypes.data), int(n_slices)))
        return pts, quads

    def escaped_count(self):
        """THIN_HALO slabs, after emit_points(): walks that left the buffer (waits for the vertex phase)."""
        n = C.c_uint64()
        _abi.check(self._ctx, self._lib.cuberille_escaped_count(self._ctx, C.byref(n)))
        return _abi.ESCAPED_OVERFLOW if n.value == 2 ** 64 - 1 else int(n.value)

    def reproject_escaped(self, dev_ptr, z_begin, nz):
        """Walk the escaped vertices again in a buffer that holds the full halo (slices [z_begin, z_begin + nz))."""
        _abi.check(self._ctx, self._lib.cuberille_reproject_escaped(self._ctx, C.c_void_p(dev_ptr), int(z_begin), int(nz)))

    # -- results -----------------------------------------------------------------------------
    def download(self, out=None):
        """The last mesh part as numpy arrays.  out: a Mesh of a previous download whose arrays are written again when
        the shapes match -- memory that has been touched before takes the copy at the link's rate, a fresh array is
        bound by its page faults."""
        res = self.result
        npnt, ncell, vpc = int(res.n_points), int(res.n_cells), int(res.verts_per_cell)
        if out is not None and out.points.shape == (npnt, 3) and out.cells.shape == (ncell, vpc) \
                and out.points.dtype == np.float32 and out.cells.dtype == np.uint64 \
                and out.points.flags.c_contiguous and out.cells.flags.c_contiguous:
            pts, cells = out.points, out.cells
        else:
            pts = np.empty((npnt, 3), dtype=np.float32)
            cells = np.empty((ncell, vpc), dtype=np.uint64)
        _abi.check(self._ctx, self._lib.cuberille_mesh_download(
            self._ctx, C.c_void_p(pts.ctypes.data), C.c_void_p(cells.ctypes.data)))
        return Mesh(pts, cells)

    def mesh_host(self):
        """The last mesh part in host memory of the CONTEXT (cuberille_mesh_host): numpy views, valid until the next count
        or extraction on this extractor -- copy what must live longer.  The context keeps the memory across extractions
        (huge pages where the system has them), so this is the fast way to the host for a series of meshes."""
        res = self.result
        npnt, ncell, vpc = int(res.n_points), int(res.n_cells), int(res.verts_per_cell)
        pp, cp = C.c_void_p(), C.c_void_p()
        _abi.check(self._ctx, self._lib.cuberille_mesh_host(self._ctx, C.byref(pp), C.byref(cp)))
        pts = np.ctypeslib.as_array(C.cast(pp, C.POINTER(C.c_float)), shape=(max(npnt * 3, 1),))[:npnt * 3].reshape(npnt, 3)
        cells = np.ctypeslib.as_array(C.cast(cp, C.POINTER(C.c_uint64)), shape=(max(ncell * vpc, 1),))[:ncell * vpc].reshape(ncell, vpc)
        return Mesh(pts, cells)

    def release_host_mesh(self):
        """Give the host memory behind mesh_host() back to the system (views handed out before become invalid)."""
        _abi.check(self._ctx, self._lib.cuberille_release_host_mesh(self._ctx))

    def hold_gradient(self, hold=True):
        """Quirk Q3 of the reference on request (cuberille_hold_gradient; txx:484): like the reference's filter object, this
        extractor keeps the gradient image (and geometry) of its next projecting extraction and walks every later volume
        along it.  hold=False drops the image and returns to the default: each volume's own gradient."""
        _abi.check(self._ctx, self._lib.cuberille_hold_gradient(self._ctx, 1 if hold else 0))

    @property
    def gradient_held(self):
        """None, or the (nz, ny, nx) of the volume whose gradient image the extractor holds."""
        dims = (C.c_int64 * 3)()
        if not self._lib.cuberille_gradient_held(self._ctx, dims):
            return None
        return (int(dims[2]), int(dims[1]), int(dims[0]))

    def write_vtk(self, path, threads=0):
        """Download the last whole-volume mesh and write it as legacy-ASCII VTK POLYDATA."""
        _abi.check(self._ctx, self._lib.cuberille_mesh_write_vtk(self._ctx, os.fsencode(path), int(threads)))

    def device_pointers(self):
        p, c = C.c_void_p(), C.c_void_p()
        _abi.check(self._ctx, self._lib.cuberille_mesh_device(self._ctx, C.byref(p), C.byref(c)))
        return p.value, c.value

    def debug_bits(self, dims_xyz):
        nx, ny, nz = dims_xyz
        W = (nx + 63) // 64
        words = np.empty((nz, ny, W), dtype=np.uint64)
        _abi.check(self._ctx, self._lib.cuberille_debug_bits(self._ctx, C.c_void_p(words.ctypes.data), words.size))
        return words

    def slice_occupancy(self, nz):
        occ = np.empty(nz, dtype=np.uint32)
        _abi.check(self._ctx, self._lib.cuberille_slice_occupancy(self._ctx, C.c_void_p(occ.ctypes.data), nz))
        return occ != 0

    def slab_info(self):
        """After count() on a slab: SlabInfo(alias_below, lowest, highest, second_highest, alias_z) -- the fields of
        cuberille_slab_status (global slices, -1 = none)."""
        st = _abi.SlabStatus()
        _abi.check(self._ctx, self._lib.cuberille_slab_info(self._ctx, C.byref(st)))
        return SlabInfo(bool(st.alias_source_below_buffer), int(st.lowest_occupied_z), int(st.highest_occupied_z),
                        int(st.second_highest_occupied_z), int(st.alias_z))

    # -- quirk Q1 across a slab boundary (include/cuberille_hip.h) ------------------------------------------------
    def slice_bits_device(self, z_global):
        """(device pointer, n_words) of the inside bits of one buffer slice."""
        p, n = C.c_void_p(), C.c_size_t()
        _abi.check(self._ctx, self._lib.cuberille_slice_bits_device(self._ctx, int(z_global), C.byref(p), C.byref(n)))
        return p.value, int(n.value)

    def recount(self, dev_source_bits):
        npnt, ncell = C.c_uint64(), C.c_uint64()
        _abi.check(self._ctx, self._lib.cuberille_recount(self._ctx, C.c_void_p(dev_source_bits), C.byref(npnt), C.byref(ncell)))
        return int(npnt.value), int(ncell.value)

    def alias_plane_device(self, z_global, dev_ids, dev_points):
        _abi.check(self._ctx, self._lib.cuberille_alias_plane_device(self._ctx, int(z_global), C.c_void_p(dev_ids),
                                                                     C.c_void_p(dev_points)))

    def set_alias_plane(self, dev_ids, dev_points):
        _abi.check(self._ctx, self._lib.cuberille_set_alias_plane(self._ctx, C.c_void_p(dev_ids), C.c_void_p(dev_points)))

    def debug_option(self, name, value):
        """Development switch of this context (cuberille_debug_set_option); "defaults" resets them all."""
        _abi.check(self._ctx, self._lib.cuberille_debug_set_option(self._ctx, name.encode(), int(value)))


def required_halo(desc, params):
    """Slices a slab buffer must hold (below, above) its owned range for these parameters (needs no GPU)."""
    lo, hi = C.c_int64(), C.c_int64()
    rc = _abi.lib().cuberille_required_halo(C.byref(desc), C.byref(params), C.byref(lo), C.byref(hi))
    if rc != _abi.OK:
        raise _abi.CuberilleError(rc, "cuberille_required_halo: bad image description or parameters")
    return int(lo.value), int(hi.value)


def minimum_halo(desc, params):
    """The least a THIN_HALO slab must hold (below, above) its owned range (needs no GPU)."""
    lo, hi = C.c_int64(), C.c_int64()
    rc = _abi.lib().cuberille_minimum_halo(C.byref(desc), C.byref(params), C.byref(lo), C.byref(hi))
    if rc != _abi.OK:
        raise _abi.CuberilleError(rc, "cuberille_minimum_halo: bad image description or parameters")
    return int(lo.value), int(hi.value)


def _clamp(v, lo, hi):
    return lo if v < lo else (hi if v > hi else v)


def _pixel_max(dt):
    dt = np.dtype(dt)
    return float(np.iinfo(dt).max) if dt.kind in "iu" else float(np.finfo(dt).max)


class CuberilleImageToMeshFilter:
    """Python mirror of itk::CuberilleImageToMeshFilter (h:110-339).

    Input: mha.Volume (host memory, like the itk::Image the reference driver hands
    over).  Output: Mesh with the reference's vertex ids, cell order and coordinates.
    """

    def __init__(self, device=0):
        self._device = device
        self._extractor = None
        self._input = None
        self._output = None
        self._dtype = np.dtype(np.uint8)
        # txx:33-40
        self._iso = 1
        self._triangles = True
        self._project = True
        self._threshold = self._threshold_asked = 0.5
        self._step = -1.0
        self._relax = 0.95
        self._max_steps = 50
        self._q1 = True
        self._variant = PROJECT_DEFAULT           # h:22-23: both alternative branches are compiled out
        self._gradient = GRADIENT_CENTRAL         # h:21: and so is the recursive-Gaussian gradient
        self._stale_gradient = False
        self.last_result = None
        # like the C++ drop-in: the GPU context and the code objects are set up when the filter is made, not inside the
        # first Update() (the reference's driver times one cold Update(), test:158-160); silent without a device --
        # Update() tries again and raises
        self._acquire(False)

    def _acquire(self, must):
        if self._extractor is None:
            try:
                self._extractor = Extractor(self._device)
                self._extractor.warm_up()
            except (_abi.CuberilleError, ImportError, OSError):
                self._extractor = None
                if must:
                    raise
        return self._extractor is not None

    # h:184 / txx:53-56
    def SetInput(self, image):
        if not isinstance(image, Volume):
            raise TypeError("SetInput expects an mha.Volume")
        self._input = image
        self._dtype = image.voxels.dtype
        self._threshold = _clamp(self._threshold_asked, 0.0, _pixel_max(self._dtype))
        if self._acquire(False) and self._dtype in PIXEL_CODES:
            self._extractor.warm_up(make_desc(self._dtype, image.dims, image.spacing, image.origin, image.direction, image.index_start))

    # h:180-181
    def SetIsoSurfaceValue(self, v):
        self._iso = v

    def GetIsoSurfaceValue(self):
        return self._iso

    # h:193-195
    def SetGenerateTriangleFaces(self, b):
        self._triangles = bool(b)

    def GetGenerateTriangleFaces(self):
        return self._triangles

    def GenerateTriangleFacesOn(self):
        self._triangles = True

    def GenerateTriangleFacesOff(self):
        self._triangles = False

    # h:199-201
    def SetProjectVerticesToIsoSurface(self, b):
        self._project = bool(b)

    def GetProjectVerticesToIsoSurface(self):
        return self._project

    def ProjectVerticesToIsoSurfaceOn(self):
        self._project = True

    def ProjectVerticesToIsoSurfaceOff(self):
        self._project = False

    # h:209-210 clamp [0, max pixel].  The reference knows the pixel type at compile time; here it is known once
    # an input is set, so the upper clamp is (re)applied against the input's type by SetInput and Update
    def SetProjectVertexSurfaceDistanceThreshold(self, v):
        self._threshold_asked = float(v)
        self._threshold = _clamp(float(v), 0.0, _pixel_max(self._dtype) if self._input is not None else float("inf"))

    def GetProjectVertexSurfaceDistanceThreshold(self):
        return self._threshold

    # h:215-216 clamp [0, 100000]
    def SetProjectVertexStepLength(self, v):
        self._step = _clamp(float(v), 0.0, 100000.0)

    def GetProjectVertexStepLength(self):
        return self._step

    # h:222-223 clamp [0, 1]
    def SetProjectVertexStepLengthRelaxationFactor(self, v):
        self._relax = _clamp(float(v), 0.0, 1.0)

    def GetProjectVertexStepLengthRelaxationFactor(self):
        return self._relax

    # h:227-228
    def SetProjectVertexMaximumNumberOfSteps(self, n):
        self._max_steps = int(n)

    def GetProjectVertexMaximumNumberOfSteps(self):
        return self._max_steps

    def SetEmulateEmptySliceAliasing(self, b):
        """Not in the reference: switches the reproduction of its quirk Q1."""
        self._q1 = bool(b)

    def SetProjectionVariant(self, variant):
        """Not in the reference's API: stands for building it with USE_ADVANCED_PROJECTION (1) or
        USE_LINESEARCH_PROJECTION (2) set (h:22-23); 0 is what it ships."""
        if variant not in (PROJECT_DEFAULT, PROJECT_ADVANCED, PROJECT_LINESEARCH):
            raise ValueError("projection variant must be 0, 1 or 2")
        self._variant = int(variant)

    def SetGradientVariant(self, gradient):
        """Not in the reference's API: stands for building it with USE_GRADIENT_RECURSIVE_GAUSSIAN set (h:21)."""
        if gradient not in (GRADIENT_CENTRAL, GRADIENT_RECURSIVE_GAUSSIAN):
            raise ValueError("gradient variant must be 0 or 1")
        self._gradient = int(gradient)

    def SetReproduceStaleGradient(self, b):
        """Not in the reference -- its behaviour, on request (like the C++ drop-in's switch of the same name): the reference's
        gradient interpolator is created once per filter object (txx:484), so every Update() after the first projecting
        one walks along the FIRST input's gradient.  Default off: each input's own gradient."""
        self._stale_gradient = bool(b)

    def GetReproduceStaleGradient(self):
        return self._stale_gradient

    def Update(self):
        if self._input is None:
            # the ITK pipeline throws for a missing required input (txx:33)
            raise RuntimeError("CuberilleImageToMeshFilter: input 0 is required but not set")
        self._acquire(True)
        vol = self._input
        if self._step < 0.0:                      # txx:82-85, sticky like the reference (quirk Q3)
            self._step = max(vol.spacing) * 0.25
        prm = make_params(self._iso, self._triangles, self._project, self._threshold, self._step, self._relax,
                          self._max_steps, self._q1, self._variant, self._gradient)
        self._extractor.hold_gradient(self._stale_gradient)
        self.last_result = self._extractor.extract_host(vol, prm)
        self._output = self._extractor.download()

    def GetOutput(self):
        return self._output
