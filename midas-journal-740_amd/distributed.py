"""Z-slab sharding of the cuberille path over torch.distributed (one process per GPU).

The reference has no distributed code.  The raster order of its sweep
(/root/reference/Source/itkCuberilleImageToMeshFilter.txx:136) is z-major, so cutting
the volume into Z-slabs in rank order gives every rank a CONTIGUOUS, ORDERED range of
vertex ids and of cell ids; the path needs exactly two exchanges:

  1. halo: each rank receives `HALO` boundary slices from its two neighbours
     (point-to-point send/recv: on GPUs each pair rides one direct xGMI link; this is
     a chain, not a ring collective).  2 slices below + 1 above are needed for the
     topology (ids of corners created one slice down); the projection walk can travel
     ~4.8 voxels (step * sum(relax^k)), hence 8.
  2. counts: one all-gather of (n_points, n_cells) per rank -> exclusive prefix = the
     rank's id offsets.

Backend "nccl" is RCCL on ROCm; the same code runs under "gloo" on CPU tensors, which
is how tests/test_distributed.py covers it without a GPU.
"""
import numpy as np

HALO = 8


def slab_range(global_nz, world, rank):
    """Balanced contiguous slices [z0, z1) owned by `rank`."""
    base, rem = divmod(int(global_nz), int(world))
    z0 = rank * base + min(rank, rem)
    return z0, z0 + base + (1 if rank < rem else 0)


def buffer_range(global_nz, z0, z1, halo=HALO):
    """Slices [lo, hi) a rank keeps in memory: its own plus the halo that exists."""
    return max(z0 - halo, 0), min(z1 + halo, int(global_nz))


def exchange_halos(buf, lo, hi, z0, z1, rank, world, group=None, wait=True):
    """buf[z - lo] holds slice z for z in [lo, hi); the owned part [z0, z1) is valid on entry.
    Fills [lo, z0) from rank-1 and [z1, hi) from rank+1.  All ranks call it together.
    wait=False (RCCL only): return the outstanding requests instead of waiting for them."""
    import os
    import torch.distributed as dist
    if buf.is_cuda and dist.get_backend(group) == "gloo" and wait:
        # rehearsal mode (several ranks sharing one GPU under gloo): stage the halos through the host
        host = buf.cpu()
        exchange_halos(host, lo, hi, z0, z1, rank, world, group)
        if z0 > lo:
            buf[:z0 - lo].copy_(host[:z0 - lo])
        if hi > z1:
            buf[z1 - lo:].copy_(host[z1 - lo:])
        return buf
    ops, keep = [], []
    nlo, nhi = z0 - lo, hi - z1                     # halo depth below / above
    if rank > 0 and nlo > 0:
        ops.append(dist.P2POp(dist.irecv, buf[0:nlo], rank - 1, group))
    if rank < world - 1 and nhi > 0:
        ops.append(dist.P2POp(dist.irecv, buf[z1 - lo:hi - lo], rank + 1, group))
    # what the neighbours miss: the upper neighbour wants my top `its_nlo` slices, the lower my bottom ones
    if rank < world - 1:
        n = min(HALO, z1 - z0) if nhi > 0 else 0
        if n > 0:
            t = buf[z1 - lo - n:z1 - lo].contiguous()
            keep.append(t)
            ops.append(dist.P2POp(dist.isend, t, rank + 1, group))
    if rank > 0:
        n = min(HALO, z1 - z0) if nlo > 0 else 0
        if n > 0:
            t = buf[z0 - lo:z0 - lo + n].contiguous()
            keep.append(t)
            ops.append(dist.P2POp(dist.isend, t, rank - 1, group))
    reqs = dist.batch_isend_irecv(ops) if ops else []
    if not wait:
        return reqs, keep
    for req in reqs:
        req.wait()
    return buf


def gather_counts(n_points, n_cells, device, group=None):
    """All-gather of every rank's (n_points, n_cells); returns an int64 array [world, 2]."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    if dist.get_backend(group) == "gloo":
        device = "cpu"
    mine = torch.tensor([int(n_points), int(n_cells)], dtype=torch.int64, device=device)
    out = torch.empty(world * 2, dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(out, mine, group=group)
    return out.cpu().numpy().reshape(world, 2)


class _DeviceArray:
    """Zero-copy view of library-owned device memory for torch (CUDA array interface, version 2)."""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False),
                                         "version": 2}


def mesh_tensors(extractor, device):
    """The last mesh part of `extractor` as torch tensors WITHOUT copying: points float32 [n,3], cells int64 [m,k]
    (the uint64 ids reinterpreted; they are < 2^63).  Valid until the next call on the extractor."""
    import torch
    res = extractor.result
    n, m, k = int(res.n_points), int(res.n_cells), int(res.verts_per_cell)
    pp, cp = extractor.device_pointers()
    pts = torch.as_tensor(_DeviceArray(pp, (n, 3), "<f4"), device=device) if n else \
        torch.empty((0, 3), dtype=torch.float32, device=device)
    cells = torch.as_tensor(_DeviceArray(cp, (m, k), "<i8"), device=device) if m else \
        torch.empty((0, k), dtype=torch.int64, device=device)
    return pts, cells


def id_offsets(counts, rank):
    """Exclusive prefix over ranks of the gathered counts -> (point_id_offset, cell_id_offset)."""
    c = np.asarray(counts, dtype=np.int64)
    return int(c[:rank, 0].sum()), int(c[:rank, 1].sum())


def aliasing_crosses_slabs(occupied, bounds):
    """Quirk Q1 check on the gathered per-slice occupancy (bool[global_nz]).  The reference re-uses
    vertices across an EMPTY slice (txx:139-141 precede 156-161); a rank can only reproduce that
    when the previous occupied slice lies inside its own counted range.  Returns the global z of
    the first slice where it does not, or -1."""
    occ = np.asarray(occupied, dtype=bool)
    prev = -1
    for z in range(occ.shape[0]):
        if not occ[z]:
            continue
        if prev >= 0 and prev < z - 1:
            for (z0, z1) in bounds:
                if z0 <= z < z1 and prev < max(z0 - 1, 0):
                    return z
        prev = z
    return -1


class ShardedExtractor:
    """Multi-GPU driver: one instance per rank, wraps one Extractor."""

    def __init__(self, extractor, global_dims, np_dtype, rank, world, group=None, spacing=(1.0, 1.0, 1.0),
                 origin=(0.0, 0.0, 0.0), direction=None, check_aliasing=False):
        from . import _abi
        from .cuberille import make_desc
        self.ex = extractor
        self.rank, self.world, self.group = rank, world, group
        self.nx, self.ny, self.nz = (int(v) for v in global_dims)
        self.z0, self.z1 = slab_range(self.nz, world, rank)
        self.lo, self.hi = buffer_range(self.nz, self.z0, self.z1)
        if world > 1 and self.z1 - self.z0 < HALO:
            raise ValueError("slabs thinner than the halo (%d slices) are not supported" % HALO)
        self.desc = make_desc(np_dtype, (self.nx, self.ny, self.hi - self.lo), spacing, origin, direction)
        self.slab = _abi.Slab(self.nz, self.lo, self.z0, self.z1, 0, 0, None)
        self._halo_event = None
        self.check_aliasing = check_aliasing
        self.counts = None

    def extract(self, buf, params):
        """buf: device tensor [hi-lo, ny, nx] whose owned slices are valid.  Runs halo exchange,
        count, the count all-gather and emit; leaves this rank's mesh part on its device."""
        import os
        import torch
        import torch.distributed as dist
        keep = None
        if self.world > 1:
            # (CUBERILLE_FORCE_EVENT_PATH: take this branch under gloo too -- test hook for one-GPU boxes)
            if buf.is_cuda and (dist.get_backend(self.group) == "nccl" or os.environ.get("CUBERILLE_FORCE_EVENT_PATH")):
                # RCCL: do not wait on the host.  req.wait() only orders torch's current stream behind the
                # transfer; an event recorded there tells the library when the halo slices are in, and it
                # thresholds the owned slices meanwhile.
                reqs, keep = exchange_halos(buf, self.lo, self.hi, self.z0, self.z1, self.rank, self.world, self.group,
                                            wait=False)
                for req in reqs:
                    req.wait()
                if self._halo_event is None:
                    self._halo_event = torch.cuda.Event()
                self._halo_event.record(torch.cuda.current_stream())
                self.slab.halo_ready_event = self._halo_event.cuda_event
            else:
                exchange_halos(buf, self.lo, self.hi, self.z0, self.z1, self.rank, self.world, self.group)
                if buf.is_cuda:
                    torch.cuda.current_stream().synchronize()
                self.slab.halo_ready_event = None
        n_p, n_c = self.ex.count(buf.data_ptr(), self.desc, params, self.slab if self.world > 1 else None)
        del keep
        if self.world > 1:
            self.counts = gather_counts(n_p, n_c, buf.device, self.group)
            poff, coff = id_offsets(self.counts, self.rank)
            if self.check_aliasing:
                occ_local = self.ex.slice_occupancy(self.hi - self.lo)[self.z0 - self.lo:self.z1 - self.lo]
                occ = [None] * self.world
                dist.all_gather_object(occ, occ_local, group=self.group)
                bounds = [slab_range(self.nz, self.world, r) for r in range(self.world)]
                bad = aliasing_crosses_slabs(np.concatenate(occ), bounds)
                if bad >= 0:
                    raise RuntimeError("empty-slice aliasing (reference quirk Q1) crosses a slab boundary at z=%d" % bad)
        else:
            self.counts = np.array([[n_p, n_c]], dtype=np.int64)
            poff, coff = 0, 0
        return self.ex.emit(poff, coff)

    def gather_mesh(self, dst=0, on_device=None):
        """Concatenate the rank parts in rank order on rank `dst` (collective 3): the
        gathered counts give every part's place, so `dst` posts one receive per rank straight into its slice of
        the whole buffers and every other rank one send of its part -- device to device over RCCL (xGMI), or
        host buffers over gloo.  Returns a Mesh of numpy arrays on `dst` (cells hold global ids already), None
        elsewhere.  on_device: transport device tensors (default: exactly when the backend is nccl)."""
        import torch
        import torch.distributed as dist
        from .cuberille import Mesh
        if self.counts is None:
            raise RuntimeError("gather_mesh before extract")
        counts = np.asarray(self.counts, dtype=np.int64)
        vpc = int(self.ex.result.verts_per_cell)
        if self.world == 1:
            return self.ex.download()
        if on_device is None:
            on_device = dist.get_backend(self.group) == "nccl"
        if on_device:
            dev = torch.device("cuda", torch.cuda.current_device())
            torch.cuda.synchronize()
            pts, cells = mesh_tensors(self.ex, dev)
        else:
            dev = torch.device("cpu")
            part = self.ex.download()
            pts = torch.from_numpy(part.points)
            cells = torch.from_numpy(part.cells.view(np.int64))
        poff = np.concatenate([[0], np.cumsum(counts[:, 0])])
        coff = np.concatenate([[0], np.cumsum(counts[:, 1])])
        ops = []
        if self.rank == dst:
            all_pts = torch.empty((int(poff[-1]), 3), dtype=torch.float32, device=dev)
            all_cells = torch.empty((int(coff[-1]), vpc), dtype=torch.int64, device=dev)
            for r in range(self.world):
                ps, cs = all_pts[int(poff[r]):int(poff[r + 1])], all_cells[int(coff[r]):int(coff[r + 1])]
                if r == dst:
                    ps.copy_(pts)
                    cs.copy_(cells)
                else:
                    if ps.numel():
                        ops.append(dist.P2POp(dist.irecv, ps, r, self.group))
                    if cs.numel():
                        ops.append(dist.P2POp(dist.irecv, cs, r, self.group))
        else:
            if pts.numel():
                ops.append(dist.P2POp(dist.isend, pts.contiguous(), dst, self.group))
            if cells.numel():
                ops.append(dist.P2POp(dist.isend, cells.contiguous(), dst, self.group))
        for req in (dist.batch_isend_irecv(ops) if ops else []):
            req.wait()
        if on_device:
            torch.cuda.synchronize()
        if self.rank != dst:
            return None
        return Mesh(all_pts.cpu().numpy(), all_cells.cpu().numpy().view(np.uint64))
