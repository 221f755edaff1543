"""Z-slab sharding of the cuberille path over torch.distributed (one process per GPU).

The reference has no distributed code.  The raster order of its sweep
(/root/reference/Source/itkCuberilleImageToMeshFilter.txx:136) is z-major, so cutting
the volume into Z-slabs in rank order gives every rank a CONTIGUOUS, ORDERED range of
vertex ids and of cell ids; the path needs exactly two exchanges:

  1. halo: each rank receives `halo` boundary slices from its two neighbours
     (point-to-point send/recv: on GPUs each pair rides one direct xGMI link; this is
     a chain, not a ring collective).  2 slices below + 1 above are needed for the
     topology (ids of corners created one slice down); the projection walk can travel
     step * sum(relax^k) / z-spacing slices (4.65 at the defaults), plus the interpolation
     cell and the gradient ring: 8 at the defaults, more for long walks or thin slices --
     the library computes it (cuberille_required_halo) and refuses a slab that holds less.
  2. counts: one all-gather of (n_points, n_cells, ...) per rank -> exclusive prefix =
     the rank's id offsets.

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


def _pair(halo):
    """A halo as (slices below, slices above); a plain number means both."""
    if isinstance(halo, (tuple, list)):
        return int(halo[0]), int(halo[1])
    return int(halo), int(halo)


def buffer_range(global_nz, z0, z1, halo=HALO):
    """Slices [lo, hi) a rank keeps in memory: its own plus the halo (a number, or (below, above)) that exists."""
    below, above = _pair(halo)
    return max(z0 - below, 0), min(z1 + above, int(global_nz))


def balanced_bounds(work, world):
    """Cut slices 0 .. len(work) into `world` contiguous slabs of (nearly) equal total work: [(z0, z1)] per rank, every
    slab at least one slice.  work: a non-negative number per slice -- e.g. ShardedExtractor.slice_work() after a step on
    a similar volume: the surface of a volume is rarely spread evenly over z, and a rank's time follows its surface."""
    w = np.maximum(np.asarray(work, dtype=np.float64), 0.0)
    nz = int(w.shape[0])
    if world > nz:
        raise ValueError("more ranks (%d) than slices (%d)" % (world, nz))
    if not w.sum() > 0.0:
        w = np.ones(nz)
    acc = np.concatenate([[0.0], np.cumsum(w)])
    cuts = [0]
    for r in range(1, world):
        z = int(np.searchsorted(acc, acc[-1] * r / world, side="left"))
        z = max(z, cuts[-1] + 1)                    # at least one slice each ...
        z = min(z, nz - (world - r))                # ... and room for the ranks above
        cuts.append(z)
    cuts.append(nz)
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


def halo_transfers(global_nz, world, rank, halo=HALO, held=0, ranges=None):
    """The point-to-point transfers of one halo exchange as this rank sees them: (recvs, sends), each a list of
    (peer, z_first, z_last_exclusive) in global slices.  A rank needs [z0 - below, z0) and [z1, z1 + above) clipped to
    the volume; every slice of that comes from the rank that owns it -- the two neighbours when the slabs are at least
    `halo` thick, more ranks when they are thinner (thin slabs, long walks).  held: the part of the halo every rank has
    already (the thin halo of a first exchange): only what lies beyond it is moved."""
    if ranges is None:
        ranges = [slab_range(global_nz, world, r) for r in range(world)]
    bufs = [buffer_range(global_nz, a, b, halo) for a, b in ranges]
    have = [buffer_range(global_nz, a, b, held) for a, b in ranges]

    def needs(r):                                   # the two halo parts of rank r
        (lo, hi), (hlo, hhi) = bufs[r], have[r]
        return [(lo, hlo), (hhi, hi)]
    recvs, sends = [], []
    z0, z1 = ranges[rank]
    for s in range(world):
        if s == rank:
            continue
        a, b = ranges[s]
        for p, q in needs(rank):                    # what I need of what s owns
            o0, o1 = max(p, a), min(q, b)
            if o1 > o0:
                recvs.append((s, o0, o1))
        for p, q in needs(s):                       # what s needs of what I own
            o0, o1 = max(p, z0), min(q, z1)
            if o1 > o0:
                sends.append((s, o0, o1))
    return recvs, sends


def halo_bytes(global_nz, world, rank, slice_bytes, halo=HALO, held=0, ranges=None):
    """Bytes this rank receives in one halo exchange of that shape."""
    recvs, _ = halo_transfers(global_nz, world, rank, halo, held, ranges)
    return sum(b - a for _, a, b in recvs) * int(slice_bytes)


def _gloo_reads_device_memory_now(tensor, group):
    """gloo's point-to-point operations know nothing of streams: handed a device tensor (the rehearsals of the RCCL event path
    on a one-GPU box, ShardedExtractor.force_event_path) its threads read the bytes straight away -- not behind what torch's
    current stream was told to wait for, as an RCCL operation does.  So under gloo the host waits for that stream first: the
    sweep that produces the bit planes, the copy that filled the owned slices.  (Found by tests/fuzz_ranks.py, round 5: two
    meshes in six thousand random three-rank steps carried a neighbour's bit planes from BEFORE its sweep.)  A no-op under
    RCCL and for host tensors."""
    import torch
    import torch.distributed as dist
    if tensor is not None and tensor.is_cuda and dist.get_backend(group) == "gloo":
        torch.cuda.current_stream(tensor.device).synchronize()


def exchange_halos(buf, lo, hi, z0, z1, rank, world, group=None, wait=True, halo=HALO, global_nz=None, held=0, ranges=None):
    """buf[z - lo] holds slice z for z in [lo, hi); the owned part [z0, z1) is valid on entry (and, held, that much halo
    around it).  Fills the rest of the halo (a number, or (below, above)) from the ranks that own those slices.  All
    ranks call it together.  wait=False (RCCL only): return the outstanding requests instead of waiting for them."""
    import torch.distributed as dist
    if buf.is_cuda and dist.get_backend(group) == "gloo" and wait:
        # rehearsal mode (several ranks sharing one GPU under gloo): stage the halos through the host
        host = buf.cpu()
        exchange_halos(host, lo, hi, z0, z1, rank, world, group, halo=halo, global_nz=global_nz, held=held, ranges=ranges)
        buf.copy_(host)
        return buf
    if global_nz is None:
        # callers that only know their own ranges: the classic two-neighbour exchange (slabs >= halo thick)
        h = _pair(halo)[0]
        recvs = ([(rank - 1, lo, z0)] if rank > 0 and z0 > lo else []) + ([(rank + 1, z1, hi)] if rank < world - 1 and hi > z1 else [])
        sends = ([(rank + 1, z1 - min(h, z1 - z0), z1)] if rank < world - 1 and hi > z1 else []) + \
                ([(rank - 1, z0, z0 + min(h, z1 - z0))] if rank > 0 and z0 > lo else [])
    else:
        recvs, sends = halo_transfers(global_nz, world, rank, halo, held, ranges)
    ops, keep = [], []
    for peer, a, b in recvs:
        ops.append(dist.P2POp(dist.irecv, buf[a - lo:b - lo], peer, group))
    for peer, a, b in sends:
        t = buf[a - lo:b - lo].contiguous()
        keep.append(t)
        ops.append(dist.P2POp(dist.isend, t, peer, group))
    _gloo_reads_device_memory_now(buf, group)
    reqs = dist.batch_isend_irecv(ops) if ops else []
    if not wait:
        return reqs, keep
    for req in reqs:
        req.wait()
    return buf


def gather_counts(n_points, n_cells, device, group=None, extra=()):
    """All-gather of every rank's (n_points, n_cells, *extra); returns an int64 array [world, 2 + len(extra)]."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    if dist.get_backend(group) == "gloo":
        device = "cpu"
    row = [int(n_points), int(n_cells)] + [int(v) for v in extra]
    mine = torch.tensor(row, dtype=torch.int64, device=device)
    out = torch.empty(world * len(row), dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(out, mine, group=group)
    return out.cpu().numpy().reshape(world, len(row))


class _DeviceArray:
    """Zero-copy view of library-owned device memory for torch (CUDA array interface, version 2)."""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False),
                                         "version": 2}


def _words_view(ptr, n, device):
    """n int64 words at `ptr` as a tensor without copying (device memory; host memory for the CPU stand-ins of the tests)."""
    import torch
    if device.type == "cpu":
        import ctypes
        return torch.frombuffer((ctypes.c_int64 * n).from_address(ptr), dtype=torch.int64)
    return torch.as_tensor(_DeviceArray(ptr, (n,), "<i8"), device=device)


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


# columns of the per-rank row that travels in the count all-gather
ROW_POINTS, ROW_CELLS, ROW_ALIAS_Z, ROW_TOP, ROW_TOP2, ROW_FAILED, ROW_ESCAPED = range(7)


def alias_plan(rows, bounds=None):
    """From the gathered per-rank rows [n_points, n_cells, alias_z, highest occupied owned slice, second highest,
    failed]: the (consumer rank, source rank, source slice, plane needed) entries of quirk Q1 crossing slab boundaries.
    alias_z >= 0 on rank r: the first occupied slice of r's counted range (its ghost slice own_z0 - 1 included) has
    nothing but empty slices below it in r's buffer, so the reference would re-use the vertices of the highest occupied
    slice STRICTLY below alias_z -- owned by some rank below r -- if there is one.  When alias_z is the ghost slice
    (it is the highest occupied slice of the rank below, so that rank's SECOND highest one is the candidate there) the
    consumer needs the source's inside bits only: it emits no cells for that slice.  bounds: the ranks' (z0, z1)."""
    plan = []
    rows = np.asarray(rows)
    for r in range(1, rows.shape[0]):
        az = int(rows[r, ROW_ALIAS_Z])
        if az < 0:
            continue
        src, zp = -1, -1
        for s in range(r):
            h = int(rows[s, ROW_TOP]) if rows[s, ROW_TOP] < az else int(rows[s, ROW_TOP2])
            if h >= 0 and h < az and h > zp:
                src, zp = s, h
        if src >= 0:
            ghost = bounds is not None and az < bounds[r][0]
            plan.append((r, src, zp, not ghost))
    return plan


def _p2p_send(t, dst, group):
    import torch.distributed as dist
    if t.is_cuda and dist.get_backend(group) == "gloo":
        t = t.cpu()
    dist.send(t.contiguous(), dst, group=group)


def _p2p_recv(shape, dtype, device, src, group):
    import torch
    import torch.distributed as dist
    staged = device.type == "cuda" and dist.get_backend(group) == "gloo"
    t = torch.empty(shape, dtype=dtype, device="cpu" if staged else device)
    dist.recv(t, src, group=group)
    return t.to(device) if staged else t


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


_BITS_GROUPS = {}


def _bits_group_of(group):
    """The second communicator of the bits-first halo: the same ranks as `group`, made once per group and process and
    re-used by every ShardedExtractor built on it (bench.py builds three: calibration rounds).  torch wants new_group
    entered by every process of the default group, whatever the ranks: build the first bits_first extractor everywhere."""
    import torch.distributed as dist
    key = None if group is None else id(group)
    if key not in _BITS_GROUPS:
        ranks = dist.get_process_group_ranks(dist.group.WORLD if group is None else group)
        _BITS_GROUPS[key] = dist.new_group(ranks=ranks, backend=dist.get_backend(group))
    return _BITS_GROUPS[key]


class StepMonitor:
    """A wall-clock watch over the phases of a multi-rank step.  Under RCCL a collective is only enqueued by the call that
    names it; what blocks, if a peer never arrives, is the step's one host wait (or torch's own watchdog, minutes later, with
    a sequence number for a message).  The driver tells the monitor which phase it enters (`enter`: the exchange with which
    ranks, the row all-gather, the host wait, ...); a daemon thread says ONCE, on stderr, which rank has been in which phase
    of which step for how long when that exceeds `timeout` seconds -- on every rank that is stuck, so the rank that is
    missing shows by its absence or by its own, different, phase -- and, abort=True, ends the process with exit code 3 (the
    launcher then ends the other ranks).  timeout None or 0: nothing is started."""

    def __init__(self, rank, world, timeout=None, abort=False, out=None):
        import threading
        self.rank, self.world = rank, world
        self.timeout = float(timeout) if timeout else 0.0
        self.abort = bool(abort)
        self.step = 0
        self._phase = None            # (text, since)
        self._said = None
        self._lock = threading.Lock()
        self._out = out
        self._thread = None
        self._stop = threading.Event()

    def enter(self, text):
        import time
        if not self.timeout:
            return
        with self._lock:
            self._phase = (text, time.monotonic())
        if self._thread is None:
            import threading
            self._thread = threading.Thread(target=self._run, daemon=True)
            self._thread.start()

    def leave(self):
        if self.timeout:
            with self._lock:
                self._phase = None

    def close(self):
        self._stop.set()

    def message(self, text, seconds):
        return ("cuberille: rank %d of %d: step %d has been in '%s' for %.0f s (limit %.0f s)%s" % (
            self.rank, self.world, self.step, text, seconds, self.timeout, " -- giving up (exit 3)" if self.abort else ""))

    def _run(self):
        import os
        import sys
        import time
        while not self._stop.wait(min(1.0, self.timeout / 4.0)):
            with self._lock:
                ph = self._phase
            if ph is None or ph is self._said:
                continue
            waited = time.monotonic() - ph[1]
            if waited > self.timeout:
                self._said = ph
                out = self._out or sys.stderr
                out.write(self.message(ph[0], waited) + "\n")
                out.flush()
                if self.abort:
                    os._exit(3)


class ShardedExtractor:
    """Multi-GPU driver: one instance per rank, wraps one Extractor."""

    def __init__(self, extractor, global_dims, np_dtype, rank, world, group=None, spacing=(1.0, 1.0, 1.0),
                 origin=(0.0, 0.0, 0.0), direction=None, check_aliasing=True, params=None, halo=None,
                 cross_slab_aliasing=True, thin_halo=False, guard=1, device_offsets=True, bounds=None, close_steps=False,
                 bits_first=False, step_timeout=None, abort_on_timeout=False, index_start=(0, 0, 0)):
        """params: the extraction parameters the slabs will be used with -- the halo is sized for them
        (cuberille_required_halo); without them it is the HALO of the default parameters on unit spacing.
        check_aliasing: look for quirk Q1 (vertex re-use across a run of empty slices) crossing a slab boundary -- costs
        nothing extra, the flags ride in the count all-gather -- and, cross_slab_aliasing, reproduce it: the rank below
        sends the source slice's inside bits (the rank above counts again), later the ids and positions of that slice's
        top-plane vertices; with cross_slab_aliasing off the case raises instead.  check_aliasing off returns whatever
        the ranks computed on their own (a mesh that differs from the single-GPU one in that case).
        thin_halo (needs params): every step exchanges only the slices a vertex needs where it STARTS
        (cuberille_minimum_halo) plus `guard` more on each side; a walk that wants a slice beyond that is put aside by
        the library, and only then -- the count of such walks rides in the count all-gather -- the ranks fetch the rest
        of the full halo and walk those vertices again.  The buffer is sized for the full halo either way.
        device_offsets (GPU buffers only): the step without a host round trip between count and emit
        (cuberille_step_begin / _end) -- the all-gather of the per-rank rows lands in device memory and the cell pass sums
        its id offset there; the host waits once per step.  A flag in any row (quirk Q1 across slabs, counts beyond the
        sizes guessed from the previous step, an escaped walk) sends every rank through the synchronous protocol.
        bits_first (with device_offsets): the neighbours get the inside BITS of the boundary slices right behind the
        sweep of the owned slices (1/32 of a float32 slice's bytes, on a communicator of their own), the count waits only
        for those; the halo's voxels, sent from the start of the step, are waited for by the projection walk alone
        (cuberille_step_classify / cuberille_step_count).  All ranks must construct their ShardedExtractor together: a
        second process group is made for the bit planes.
        close_steps: end every one-wait step in a gather of the ranks' outcomes, so that a failure inside
        cuberille_step_end is raised on every rank (one more collective and host wait per step).
        bounds: the ranks' slices [(z0, z1)] (contiguous, in rank order) instead of slabs of equal thickness -- e.g.
        balanced_bounds(slice_work()) from a step on a similar volume.
        step_timeout (seconds): a StepMonitor watches the phases of every step and says on stderr which rank sits in which
        exchange, collective or wait for longer than that -- and, abort_on_timeout, ends the process (exit code 3)."""
        from . import _abi
        from .cuberille import make_desc, minimum_halo, required_halo
        self.ex = extractor
        self.rank, self.world, self.group = rank, world, group
        self.nx, self.ny, self.nz = (int(v) for v in global_dims)
        self.bounds = [slab_range(self.nz, world, r) for r in range(world)] if bounds is None else \
            [(int(a), int(b)) for a, b in bounds]
        if len(self.bounds) != world or self.bounds[0][0] != 0 or self.bounds[-1][1] != self.nz or \
                any(a >= b for a, b in self.bounds) or any(p[1] != q[0] for p, q in zip(self.bounds[:-1], self.bounds[1:])):
            raise ValueError("bounds must cut slices 0 .. %d into %d contiguous, non-empty slabs" % (self.nz, world))
        self.z0, self.z1 = self.bounds[rank]
        self._geo = (np_dtype, spacing, origin, direction)
        whole = make_desc(np_dtype, (self.nx, self.ny, self.nz), spacing, origin, direction, index_start)
        if halo is None:
            halo = HALO
            if params is not None:
                halo = max(required_halo(whole, params))
        self.halo = int(halo)
        self.lo, self.hi = buffer_range(self.nz, self.z0, self.z1, self.halo)
        if world > self.nz:
            raise ValueError("more ranks (%d) than slices (%d)" % (world, self.nz))
        self.desc = make_desc(np_dtype, (self.nx, self.ny, self.hi - self.lo), spacing, origin, direction, index_start)
        self.slab = _abi.Slab(self.nz, self.lo, self.z0, self.z1, 0, 0, None, None)
        # the thin form of the same slab: a window of the same buffer
        self.thin = None
        if thin_halo and world > 1:
            if params is None:
                raise ValueError("thin_halo needs params (the halo's two sizes follow from them)")
            below, above = minimum_halo(whole, params)
            t = (min(below + int(guard), self.halo), min(above + int(guard), self.halo))
            if t[0] < self.halo or t[1] < self.halo:          # (else there is nothing to save)
                self.thin = t
                self.tlo, self.thi = buffer_range(self.nz, self.z0, self.z1, t)
                self.thin_desc = make_desc(np_dtype, (self.nx, self.ny, self.thi - self.tlo), spacing, origin, direction, index_start)
                self.thin_slab = _abi.Slab(self.nz, self.tlo, self.z0, self.z1, 0, _abi.SLAB_THIN_HALO, None, None)
        self.itemsize = int(np.dtype(np_dtype).itemsize)
        self._halo_event = None
        self.force_event_path = False             # tests: the non-blocking exchange + event hand-off without RCCL
        self._vox_event = None
        self.check_aliasing = check_aliasing
        self.cross_slab_aliasing = cross_slab_aliasing
        self.counts = None
        self.device_offsets = bool(device_offsets)
        self.close_steps = bool(close_steps)
        self.bits_first = bool(bits_first) and world > 1
        self._bits_group = None
        self._side = None                         # bits_first: side streams and events of the two exchanges
        if self.bits_first:
            self._bits_group = _bits_group_of(group)
        self.force_step_path = False              # tests: the one-wait step over CPU tensors and gloo (a stand-in extractor)
        self._lib_stream = None                   # device_offsets: the library's work goes to a torch stream of ours
        self._rows = None
        self._lazy_counts = None
        self._ev = None
        # what the last extract() cost besides kernels (bench.py prints them)
        self.stats = {"halo_bytes": 0, "halo_bit_bytes": 0, "host_syncs": 0, "collectives": 0, "escaped": 0, "deep_halo_fetched": False}
        self.monitor = StepMonitor(rank, world, step_timeout, abort_on_timeout)

    def _peers(self, halo, held=0):
        recvs, sends = halo_transfers(self.nz, self.world, self.rank, halo, held, self.bounds)
        return "receives from ranks %s, sends to ranks %s" % (sorted({p for p, _, _ in recvs}), sorted({p for p, _, _ in sends}))

    # -- halo exchange -------------------------------------------------------------------------------------------
    def _exchange(self, buf, halo, held, slab):
        """Fill the halo (beyond what `held` says is there) and tell `slab` how to wait for it: on the host (gloo), or
        through two events (RCCL: the library runs on its own stream and thresholds the owned slices meanwhile)."""
        import torch
        import torch.distributed as dist
        self.monitor.enter("halo exchange, %s slices beyond %s held: %s" % (_pair(halo), _pair(held), self._peers(halo, held)))
        self.stats["halo_bytes"] += halo_bytes(self.nz, self.world, self.rank, self.nx * self.ny * self.itemsize, halo, held,
                                               self.bounds)
        # (self.force_event_path: take the event branch under gloo too -- what the tests on one-GPU boxes set)
        if buf.is_cuda and (dist.get_backend(self.group) == "nccl" or self.force_event_path):
            if self._vox_event is None:
                self._vox_event = torch.cuda.Event()
                self._halo_event = torch.cuda.Event()
            self._vox_event.record(torch.cuda.current_stream())
            reqs, keep = exchange_halos(buf, self.lo, self.hi, self.z0, self.z1, self.rank, self.world, self.group,
                                        wait=False, halo=halo, global_nz=self.nz, held=held, ranges=self.bounds)
            for req in reqs:
                req.wait()                     # orders torch's current stream behind the transfer, not the host
            self._halo_event.record(torch.cuda.current_stream())
            slab.voxels_ready_event = self._vox_event.cuda_event
            slab.halo_ready_event = self._halo_event.cuda_event
            return keep
        exchange_halos(buf, self.lo, self.hi, self.z0, self.z1, self.rank, self.world, self.group, halo=halo,
                       global_nz=self.nz, held=held, ranges=self.bounds)
        if buf.is_cuda:
            torch.cuda.current_stream().synchronize()
            self.stats["host_syncs"] += 1
        slab.halo_ready_event = None
        slab.voxels_ready_event = None
        return None

    def extract(self, buf, params):
        """buf: device tensor [hi-lo, ny, nx] whose owned slices are valid (written on torch's current stream).
        Runs halo exchange, count, the count all-gather and emit; leaves this rank's mesh part on its device."""
        self.stats = {"halo_bytes": 0, "halo_bit_bytes": 0, "host_syncs": 0, "collectives": 0, "escaped": 0, "deep_halo_fetched": False}
        self.monitor.step += 1
        try:
            return self._extract(buf, params)
        finally:
            self.monitor.leave()

    def _extract(self, buf, params):
        import torch
        from .cuberille import required_halo
        if self.world == 1:
            # (cuberille_extract_device is the one-wait step with a single rank: from the second extraction on a context
            #  on, everything is launched back to back and the host waits once.  The library runs on a stream of its own:
            #  an event recorded behind whatever wrote the buffer on torch's current stream orders it -- no host wait)
            slab = None
            if buf.is_cuda:
                from . import _abi
                if self._vox_event is None:
                    self._vox_event = torch.cuda.Event()
                self._vox_event.record(torch.cuda.current_stream())
                slab = _abi.Slab(0, 0, 0, 0, 0, 0, None, self._vox_event.cuda_event)    # all-zero ranges: the whole volume
            res = self.ex.extract_device(buf.data_ptr(), self.desc, params, slab)
            self.counts = np.array([[int(res.n_points), int(res.n_cells)]], dtype=np.int64)
            self.stats["host_syncs"] = 1 if buf.is_cuda else 0
            return res
        need = max(required_halo(self.desc, params))
        if need > self.halo:
            raise ValueError("these parameters let the projection reach %d slices; this ShardedExtractor was built "
                             "with a halo of %d (pass params= or halo= to its constructor)" % (need, self.halo))
        thin = self.thin is not None and bool(params.project_vertices) and int(params.projection_variant) == 0
        if self.device_offsets and (buf.is_cuda or self.force_step_path) and hasattr(self.ex, "step_begin"):
            return self._extract_step(buf, params, thin)
        return self._extract_sync(buf, params, thin, held=0)

    def _extract_step(self, buf, params, thin):
        """One step with ONE host wait: exchange -> [count, vertex phase] -> all-gather of the rows in device memory ->
        cells with the offset summed on the device.  The library works on a stream of ours; events order it against
        torch's current stream, where the collectives are enqueued.
        A failure on one rank strands nobody: a rank whose cuberille_step_begin failed still joins the row all-gather,
        with a row that carries the library's "this rank failed" flag (cuberille_failed_row) -- every peer's step_end
        then comes back with CUBERILLE_RETRY, all ranks meet in the count all-gather of the synchronous protocol and
        raise there together.  cuberille_step_end is the last thing of a step: a rank that fails in it raises, and the
        others learn of it in the closing gather when close_steps is on (one more host-synchronised collective per step;
        off by default: what can fail there -- the cell launch, the wait itself -- is a device fault that ends the
        process anyway)."""
        import torch
        import torch.distributed as dist
        from . import _abi
        dev = buf.device
        cuda = dev.type == "cuda"                  # (CPU tensors: the gloo stand-ins of tests/test_distributed.py)
        if cuda and self._lib_stream is None:
            self._lib_stream = torch.cuda.Stream(device=dev)
            self._ev = (torch.cuda.Event(), torch.cuda.Event())
            self.ex.use_stream(self._lib_stream)
        cur, ls = (torch.cuda.current_stream(), self._lib_stream) if cuda else (None, None)
        if thin:
            slab, desc, halo = self.thin_slab, self.thin_desc, self.thin
            base = buf[self.tlo - self.lo:self.thi - self.lo]
        else:
            slab, desc, halo, base = self.slab, self.desc, self.halo, buf
        failed = None
        try:
            if self.bits_first and hasattr(self.ex, "step_classify"):
                ptr, nbytes, keep = self._begin_bits_first(buf, base, desc, params, slab, halo, dev, cur, ls)
            else:
                keep = self._exchange(buf, halo, 0, slab)
                if cuda and slab.voxels_ready_event is None:
                    ls.wait_stream(cur)            # (host-waited exchange: the buffer is ready, order the streams all the same)
                self.monitor.enter("cuberille_step_begin (sweep, count and vertex phase behind the halo exchange)")
                ptr, nbytes = self.ex.step_begin(base.data_ptr(), desc, params, slab)
            nw = nbytes // 8
            row = _words_view(ptr, nw, dev)
        except _abi.CuberilleError as e:
            keep = None
            failed = e
            row = torch.from_numpy(_abi.failed_row().view(np.int64).copy()).to(dev)
            nw = row.numel()
        if self._rows is None or self._rows.numel() != self.world * nw or self._rows.device != dev:
            self._rows = torch.empty(self.world * nw, dtype=torch.int64, device=dev)
        self.monitor.enter("all-gather of the %d ranks' rows (behind the halo exchange: %s)" % (self.world, self._peers(halo)))
        if cuda:
            self._ev[0].record(ls)
            cur.wait_event(self._ev[0])
        if cuda and dist.get_backend(self.group) == "gloo":
            # rehearsal on one GPU: the rows go through the host (a wait that RCCL does not need)
            host = torch.empty(self.world * nw, dtype=torch.int64)
            dist.all_gather_into_tensor(host, row.cpu(), group=self.group)
            self._rows.copy_(host)
            self.stats["host_syncs"] += 1
        else:
            dist.all_gather_into_tensor(self._rows, row, group=self.group)
        self.stats["collectives"] += 1
        if cuda:
            self._ev[1].record(cur)
            ls.wait_event(self._ev[1])
            if self._side is not None:             # bits_first: whatever the caller does to the buffer next comes behind both exchanges
                cur.wait_stream(self._side[0])
                cur.wait_stream(self._side[1])
        res, done, end_failed = None, False, None
        if failed is None:
            try:
                self.monitor.enter("cuberille_step_end: the one host wait of the step -- for this rank's kernels, the halo exchange "
                                   "(%s) and the all-gather of %d rows" % (self._peers(halo), self.world))
                res, done = self.ex.step_end(self._rows.data_ptr(), self.world, self.rank)
                self.stats["host_syncs"] += 1
            except _abi.CuberilleError as e:
                end_failed = e
        del keep
        if self.close_steps:
            flag = 1 if end_failed is not None else 0
            self.monitor.enter("closing gather of the step's outcomes (%d ranks)" % self.world)
            ok = gather_counts(0, 0, dev, self.group, extra=(flag,))
            self.stats["collectives"] += 1
            if cuda:
                self.stats["host_syncs"] += 1
            self._raise_if_any_failed(ok[:, 2], "cuberille_step_end", end_failed)
        elif end_failed is not None:
            raise end_failed
        if done:
            # (the offsets were never on the host: gather_mesh reads the counts from the rows that step_end brought back)
            self.counts = None
            self._lazy_counts = (nw, 2 if int(params.generate_triangles) else 1)
            return res
        # a flag somewhere: every rank goes on from its finished count with the host in the loop (a rank whose step_begin
        # failed has no count: it only carries its failure into the gather, where every rank raises)
        return self._extract_sync(buf, params, thin, held=halo,
                                  resume=(int(res.n_points), int(res.n_cells)) if failed is None else None, failed=failed)

    def _exchange_planes(self, bits_ptr, wps, z_begin, recvs, sends, dev, group, wait):
        """The inside-bit planes of the halo: sends views of the library's bit volume (owned slices), receives straight into
        it (halo slices).  Slice z (global) is the wps words at bits_ptr + (z - z_begin) * wps * 8."""
        import torch
        import torch.distributed as dist

        def plane(a, b):
            if bits_ptr is None:       # this rank's sweep failed: its neighbours still get (and send) their planes
                return torch.zeros((b - a) * wps, dtype=torch.int64, device=dev)
            return _words_view(bits_ptr + (a - z_begin) * wps * 8, (b - a) * wps, dev)
        staged = dev.type == "cuda" and dist.get_backend(group) == "gloo" and wait
        ops, keep, landed = [], [], []
        for peer, a, b in recvs:
            t = torch.empty((b - a) * wps, dtype=torch.int64) if staged else plane(a, b)
            landed.append((t, a, b))
            ops.append(dist.P2POp(dist.irecv, t, peer, group))
        for peer, a, b in sends:
            t = plane(a, b).cpu() if staged else plane(a, b)
            keep.append(t)
            ops.append(dist.P2POp(dist.isend, t, peer, group))
        if not staged and keep:
            _gloo_reads_device_memory_now(keep[0], group)
        reqs = dist.batch_isend_irecv(ops) if ops else []
        if not wait:
            return reqs, keep
        for req in reqs:
            req.wait()
        if staged:
            for t, a, b in landed:
                plane(a, b).copy_(t)
        return [], keep

    def _begin_bits_first(self, buf, base, desc, params, slab, halo, dev, cur, ls):
        """The first half of a one-wait step with the bits-first halo.  Returns (row pointer, row bytes, tensors to keep
        alive until the step is over).

            side stream V : wait voxels_ready; halo VOXELS over the main communicator ........ record halo_voxels
            library       : wait voxels_ready; sweep of the owned slices; record bits_ready
            side stream B : wait bits_ready; halo BIT PLANES over the second communicator; record halo_bits
            library       : wait halo_bits; count, scan, gate, heads, vertex scatter; wait halo_voxels; walk
        """
        import torch
        import torch.distributed as dist
        from . import _abi
        cuda = dev.type == "cuda"
        slice_bytes = self.nx * self.ny * self.itemsize
        wps = self.ny * ((self.nx + 63) // 64)
        recvs, sends = halo_transfers(self.nz, self.world, self.rank, halo, 0, self.bounds)
        self.stats["halo_bytes"] += sum(b - a for _, a, b in recvs) * slice_bytes
        self.stats["halo_bit_bytes"] += sum(b - a for _, a, b in recvs) * wps * 8
        z_begin = int(slab.z_begin)
        if cuda and (dist.get_backend(self.group) == "nccl" or self.force_event_path):
            if self._side is None:
                self._side = (torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev),
                              torch.cuda.Event(), torch.cuda.Event(), torch.cuda.Event(), torch.cuda.Event())
            sV, sB, ev_vox, ev_halo_vox, ev_bits, ev_halo_bits = self._side
            self.monitor.enter("bits-first halo: voxels on the main communicator, bit planes on the second (%s)" % self._peers(halo))
            ev_vox.record(cur)
            slab.voxels_ready_event = ev_vox.cuda_event
            slab.halo_ready_event = None
            sV.wait_event(ev_vox)
            with torch.cuda.stream(sV):
                reqs, keep_v = exchange_halos(buf, self.lo, self.hi, self.z0, self.z1, self.rank, self.world, self.group,
                                              wait=False, halo=halo, global_nz=self.nz, held=0, ranges=self.bounds)
                for req in reqs:
                    req.wait()                 # orders side stream V behind the transfer, not the host
                ev_halo_vox.record(sV)
            bits_ptr, swept = None, None
            try:
                bits_ptr, got = self.ex.step_classify(base.data_ptr(), desc, params, slab)
                assert got == wps
                ev_bits.record(ls)
                sB.wait_event(ev_bits)
            except _abi.CuberilleError as e:       # the neighbours are on their way into the plane exchange: take part, then fail
                swept = e
            with torch.cuda.stream(sB):
                reqs, keep_b = self._exchange_planes(bits_ptr, wps, z_begin, recvs, sends, dev, self._bits_group, wait=False)
                for req in reqs:
                    req.wait()
                ev_halo_bits.record(sB)
            if swept is not None:
                raise swept
            ptr, nbytes = self.ex.step_count(ev_halo_bits.cuda_event, ev_halo_vox.cuda_event)
            return ptr, nbytes, (keep_v, keep_b)
        # host-waited exchanges (gloo: CPU stand-ins, or several ranks rehearsing on one GPU)
        self.monitor.enter("bits-first halo, host-waited: sweep, bit planes, voxels (%s)" % self._peers(halo))
        slab.voxels_ready_event = None
        slab.halo_ready_event = None
        if cuda:
            cur.synchronize()
            self.stats["host_syncs"] += 1
        bits_ptr, swept = None, None
        try:
            bits_ptr, got = self.ex.step_classify(base.data_ptr(), desc, params, slab)
            assert got == wps
        except _abi.CuberilleError as e:
            swept = e
        if cuda:
            ls.synchronize()
            self.stats["host_syncs"] += 1
        _, keep_b = self._exchange_planes(bits_ptr, wps, z_begin, recvs, sends, dev, self._bits_group, wait=True)
        exchange_halos(buf, self.lo, self.hi, self.z0, self.z1, self.rank, self.world, self.group, halo=halo,
                       global_nz=self.nz, held=0, ranges=self.bounds)
        if cuda:
            cur.synchronize()
            ls.wait_stream(cur)
        if swept is not None:
            raise swept
        ptr, nbytes = self.ex.step_count(None, None)
        return ptr, nbytes, keep_b

    def _extract_sync(self, buf, params, thin, held, resume=None, failed=None):
        """One step with the host in the loop: exchange, count, (vertex phase), all-gather of the counts, cells.
        resume=(n_points, n_cells): the count is done already (a step that came back with CUBERILLE_RETRY).
        failed: this rank's cuberille_step_begin failed -- it has nothing to count or emit and goes straight to the
        gather that tells the others."""
        import torch
        from . import _abi
        dev = buf.device
        if thin:
            slab, desc, halo = self.thin_slab, self.thin_desc, self.thin
            base = buf[self.tlo - self.lo:self.thi - self.lo]
        else:
            slab, desc, halo, base = self.slab, self.desc, self.halo, buf
        keep = self._exchange(buf, halo, held, slab) if _pair(halo) != _pair(held) else None
        # a failure on one rank must not leave the others waiting in the all-gather: it travels with the counts
        n_p = n_c = 0
        n_esc = 0
        info = None
        try:
            if failed is not None:
                pass
            elif resume is not None:
                n_p, n_c = resume
            else:
                self.monitor.enter("cuberille_count (sweep and count, a host wait)")
                n_p, n_c = self.ex.count(base.data_ptr(), desc, params, slab)
                self.stats["host_syncs"] += 1
            if failed is None and self.check_aliasing and params.emulate_empty_slice_aliasing:
                info = self.ex.slab_info()
            if failed is None and (info is None or info.alias_z < 0):
                # nothing another rank says can change this rank's counts: the vertices are scattered and projected
                # while the counts are gathered, only the cells wait for the id offsets
                self.ex.emit_points()
                if thin:
                    # ... unless walks may have left the thin halo: their number has to travel with the counts
                    n_esc = self.ex.escaped_count()
                    self.stats["host_syncs"] += 1
        except _abi.CuberilleError as e:
            failed = e
        del keep
        self.monitor.enter("all-gather of the %d ranks' counts (host in the loop)" % self.world)
        rows = gather_counts(n_p, n_c, dev, self.group, extra=(
            info.alias_z if info is not None else -1, info.highest if info is not None else -1,
            info.second_highest if info is not None else -1, 0 if failed is None else 1, n_esc))
        self.stats["collectives"] += 1
        if dev.type == "cuda":
            self.stats["host_syncs"] += 1
        self.counts = rows[:, :2]
        self._raise_if_any_failed(rows[:, ROW_FAILED], "cuberille_count", failed)
        # quirk Q1 across slab boundaries: rank r assumed that nothing is occupied below its buffer; a rank below says
        # otherwise.  Every rank derives the same plan from the gathered rows.
        plan = alias_plan(rows, self.bounds) if self.check_aliasing else []
        if plan and not self.cross_slab_aliasing:
            raise RuntimeError("empty-slice aliasing (reference quirk Q1) crosses the slab boundary below rank %d" % plan[0][0])
        if thin:
            escaped = rows[:, ROW_ESCAPED].copy()
            late = rows[:, ROW_ALIAS_Z] >= 0
            if not plan and late.any():
                # (round-4 advisor finding) A rank whose first occupied slice has only empty slices below it in its buffer
                # could not say before the gather whether its counts stand -- so its vertex phase, and with it the number
                # of its walks that left the thin halo, did not travel with the counts (on a resumed step the walks HAVE
                # run, blindly, and may have escaped).  No rank below holds a source: the counts stand.  Those ranks run
                # (or look at) their vertex phase now and every rank learns of their escapes in a second, small gather --
                # every rank sees `late` in the rows, so all of them take this turn together.
                if late[self.rank] and failed is None:
                    try:
                        self.ex.emit_points()
                        n_esc = self.ex.escaped_count()
                        self.stats["host_syncs"] += 1
                    except _abi.CuberilleError as e:
                        failed = e
                self.monitor.enter("all-gather of the escapes of ranks %s" % [int(r) for r in np.nonzero(late)[0]])
                rows_e = gather_counts(0, 0, dev, self.group, extra=(n_esc, 0 if failed is None else 1))
                self.stats["collectives"] += 1
                self._raise_if_any_failed(rows_e[:, 3], "cuberille_emit_points", failed)
                escaped = np.where(late, rows_e[:, 2], escaped)
            self.stats["escaped"] = int(n_esc)
            overflow = (escaped >= _abi.ESCAPED_OVERFLOW).any()
            if plan or overflow or escaped.any():
                # the rare way: every rank completes its halo to the full one ...
                self.stats["deep_halo_fetched"] = True
                keep = self._exchange(buf, self.halo, self.thin, self.slab)
                if dev.type == "cuda":
                    torch.cuda.current_stream().synchronize()     # (the re-projection reads the new slices at once)
                del keep
                if plan or overflow:
                    # ... and the step is taken again on it (quirk Q1 crossing a boundary, or more escapes than the
                    # library's list holds): the hand-over protocol below then runs on full buffers only
                    return self._extract_sync(buf, params, False, held=self.halo)
                try:
                    if n_esc:                                  # ... or only the escaped vertices are walked again
                        self.ex.reproject_escaped(buf.data_ptr(), self.lo, self.hi - self.lo)
                except _abi.CuberilleError as e:
                    failed = e
        n_words = self.ny * ((self.nx + 63) // 64)
        n_corners = (self.nx + 1) * (self.ny + 1)
        for r, src, zp, _ in plan:
            self.monitor.enter("quirk-Q1 hand-over: inside bits of slice %d from rank %d to rank %d" % (zp, src, r))
            # the consumer counts again with the source slice's inside bits at hand: the re-used vertices are no
            # longer created.  A rank that fails here still takes part in every transfer of the plan (nobody waits
            # for a message that never comes); the failure travels in the second gather and is raised everywhere.
            if self.rank == src:
                try:
                    ptr, n = self.ex.slice_bits_device(zp)
                    bits = _words_view(ptr, n, dev)
                except _abi.CuberilleError as e:
                    failed = failed or e
                    bits = torch.zeros((n_words,), dtype=torch.int64, device=dev)
                _p2p_send(bits, r, self.group)
            if self.rank == r:
                bits = _p2p_recv((n_words,), torch.int64, dev, src, self.group)
                if dev.type == "cuda":
                    torch.cuda.current_stream().synchronize()
                try:
                    n_p, n_c = self.ex.recount(bits.data_ptr())
                except _abi.CuberilleError as e:
                    failed = failed or e
                del bits
        if plan:
            self.monitor.enter("second all-gather of the counts (after the quirk-Q1 recount)")
            rows2 = gather_counts(n_p, n_c, dev, self.group, extra=(0 if failed is None else 1,))
            self.counts = rows2[:, :2]
            self._raise_if_any_failed(rows2[:, 2], "cuberille_recount", failed)
        poff, coff = id_offsets(self.counts, self.rank)
        mine = [e for e in plan if e[0] == self.rank and e[3]]
        serve = [e for e in plan if e[1] == self.rank and e[3]]
        planes, res = None, None
        self.monitor.enter("cuberille_emit (and the quirk-Q1 planes it waits for: %s)" % ([(e[1], e[2]) for e in mine] or "none"))
        try:
            if mine:
                # ids and final positions of the vertices under the (x, y) corner keys of the source slice's top plane
                # (a plane whose first id is -2 says that the serving rank failed)
                _, src, _, _ = mine[0]
                ids = _p2p_recv((n_corners,), torch.int64, dev, src, self.group)
                pts = _p2p_recv((n_corners, 3), torch.float32, dev, src, self.group)
                if dev.type == "cuda":
                    torch.cuda.current_stream().synchronize()
                if n_corners and int(ids[0]) == -2:
                    raise RuntimeError("the rank serving the re-used vertices (rank %d) failed" % src)
                self.ex.set_alias_plane(ids.data_ptr(), pts.data_ptr())
                planes = (ids, pts)
            res = self.ex.emit(poff)
        except (_abi.CuberilleError, RuntimeError) as e:
            failed = failed or e
        del planes
        for r, _, zp, _ in serve:
            ids = torch.empty((n_corners,), dtype=torch.int64, device=dev)
            pts = torch.zeros((n_corners, 3), dtype=torch.float32, device=dev)
            if dev.type == "cuda":
                torch.cuda.current_stream().synchronize()
            try:
                if failed is not None:
                    raise failed
                self.ex.alias_plane_device(zp, ids.data_ptr(), pts.data_ptr())
            except (_abi.CuberilleError, RuntimeError) as e:
                failed = failed or e
                ids.fill_(-2)
            _p2p_send(ids, r, self.group)
            _p2p_send(pts, r, self.group)
        # close the step together: a hand-over made ranks depend on each other after the counts were agreed -- and without
        # one, a rank whose emit fails alone (out of memory for its part of the mesh, say) must not leave the others on
        # their way into the next step's collectives (round-4 advisor finding).  This is the protocol with the host in
        # the loop: one more small gather does not change what it costs.
        self.monitor.enter("closing gather of the step's outcomes (%d ranks)" % self.world)
        ok = gather_counts(0, 0, dev, self.group, extra=(0 if failed is None else 1,))
        self.stats["collectives"] += 1
        self._raise_if_any_failed(ok[:, 2], "cuberille_emit", failed)
        return res

    def slice_work(self, result):
        """After extract(): an estimate of the device time every slice of the volume cost, the same vector (float64,
        one entry per global slice, milliseconds) on every rank -- from each rank's two measured intervals (the pass over
        its voxels, the emit phase over its surface) spread over its slices by voxels and by vertices created.  Feed it to
        balanced_bounds() to cut the next, similar volume into slabs of equal work.  (Two small collectives; not part
        of a step.)"""
        import torch
        import torch.distributed as dist
        own = self.z1 - self.z0
        pts, _ = self.ex.slice_counts(own)
        ms_pass, ms_emit = float(result.ms_pass), max(float(result.ms_total) - float(result.ms_pass), 0.0)
        if ms_pass <= 0.0:                                       # (small volumes carry one interval only)
            ms_pass, ms_emit = 0.25 * ms_emit, 0.75 * ms_emit
        per_slice = np.full(own, ms_pass / own)
        tot = float(pts.sum())
        per_slice += ms_emit * (pts.astype(np.float64) / tot if tot > 0 else 1.0 / own)
        work = np.zeros(self.nz, dtype=np.float64)
        work[self.z0:self.z1] = per_slice
        if self.world > 1:
            t = torch.from_numpy(work)
            if dist.get_backend(self.group) != "gloo":
                t = t.cuda()
            dist.all_reduce(t, group=self.group)
            work = t.cpu().numpy()
        return work

    def _raise_if_any_failed(self, flags, what, mine):
        if np.asarray(flags).any():
            bad = [int(r) for r in np.nonzero(np.asarray(flags))[0]]
            raise RuntimeError("%s failed on rank(s) %s%s" % (what, bad, ": %s" % mine if mine is not None else ""))

    def gather_mesh(self, dst=0, on_device=None):
        """Concatenate the rank parts in rank order on rank `dst` (collective 3): the
        gathered counts give every part's place, so `dst` posts one receive per rank straight into its slice of
        the whole buffers and every other rank one send of its part -- device to device over RCCL (xGMI), or
        host buffers over gloo.  Returns a Mesh of numpy arrays on `dst` (cells hold global ids already), None
        elsewhere.  on_device: transport device tensors (default: exactly when the backend is nccl)."""
        import torch
        import torch.distributed as dist
        from .cuberille import Mesh
        if self.counts is None and self._lazy_counts is not None:
            nw, tri = self._lazy_counts
            t = self._rows.view(self.world, nw).cpu().numpy()       # the rows of the last step: totV, totQ, V0, Q0, ...
            self.counts = np.stack([t[:, 0] - t[:, 2], (t[:, 1] - t[:, 3]) * tri], 1).astype(np.int64)
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
