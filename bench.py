#!/usr/bin/env python3
"""bench.py -- Mvoxels/s polygonized + achieved HBM GB/s on MI355X (BASELINE.json metric).

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W
  python bench.py --gpus N ...        (no launcher: bench.py starts that same command itself, as a child process, before it
                                       touches the GPU; an attempt that ends without a line is followed by the plain protocol)

A "step" is one complete extraction (the whole GenerateData() path: classify, count,
scan, emit, projection, triangle split) of a volume already resident in HBM.
Workload: BASELINE.json configs[3], the 1024^3 float32 Marschner-Lobb volume, iso 0.5,
triangles + vertex projection on (thr 0.002, step 0.25, relax 0.95, 50 steps).
N>1, --scaling strong (the default: it is what the metric names, "1024^3 @1/2/4/8 GPU", and
what configs[3] and [4] describe): that ONE volume is cut into N Z-slabs, every rank generates
only its own slices; per step a thin halo (3 + 3 slices: what a vertex needs where it starts, plus
one) crosses the links over RCCL -- the rest of the 8-slice halo only when a walk really leaves it --
and one all-gather of the per-rank rows that lands in device memory (the cell pass sums its id
offset there: one host wait per step).  --full-halo / --host-offsets switch the two back.  --scaling weak: each rank owns one 1024^3 block of a 1024x1024x(1024 N) volume (the
block repeats along z).  configs[4]: --workload noise --size 2048.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md; ~6.3 TB/s achievable)
STAGE_REPS = 3          # extractions with per-stage events, after the timed region (stages_ms)


def generate_block(pkg, torch, workload, n, lo, hi, period, device):
    """Slices [lo, hi) of the global volume, generated in HBM in chunks (float64 temporaries)."""
    chunks = []
    step = max(1, (1 << 26) // (n * n))
    for a in range(lo, hi, step):
        b = min(a + step, hi)
        if workload == "marschner_lobb":
            chunks.append(pkg.volumes.marschner_lobb(n, a, b, xp=torch, device=device, period=period))
        elif workload == "sphere":
            assert period is None
            chunks.append(pkg.volumes.sphere_sdf(n, a, b, xp=torch, device=device))
        elif workload == "noise":
            chunks.append(pkg.volumes.gradient_noise(n, n, n * 1000000, a, b, xp=torch, device=device))
        else:
            raise ValueError(workload)
    return torch.cat(chunks, 0).contiguous()


WORKLOADS = {
    # name: (dtype, iso, threshold)
    "marschner_lobb": (np.float32, 0.5, 0.002),
    "sphere": (np.float32, 0.0, 0.05),
    "noise": (np.uint8, 128, 0.5),
}


# the rocprofv3 PMC summary (profiles/collect.sh) of exactly the configuration a line reports: workload, edge -> file
# (file, scale, note): configs[4]'s counters are taken on a 2048 x 2048 x 256 slab of the same field -- rocprofv3's counter passes
# die on the 8.6 GB volume -- and scaled by the slice ratio: every pass kernel's traffic is proportional to the slices it sweeps
PMC_FILES = {("marschner_lobb", 1024): ("r5_pmc_hbm.csv", 1.0, None), ("sphere", 512): ("r5_config3_sphere512_pmc_hbm.csv", 1.0, None),
             ("noise", 2048): ("r5_config5_slab2048x2048x256_pmc_hbm.csv", 8.0,
                               "counted on a 2048 x 2048 x 256 slab of the same field (the counter passes do not survive the "
                               "8.6 GB volume), scaled by 2048 / 256")}
PASS_KERNELS = ("k_classify_span<", "k_classify_flat<", "k_classify_span_rows<", "k_count<", "k_count_dense", "k_block_scan", "k_block_partial")


def measured_traffic(args, world, alg_bytes):
    """HBM bytes per classify+count+scan pass from the committed rocprofv3 PMC passes (profiles/*_pmc_hbm.csv:
    FETCH_SIZE and WRITE_SIZE in KiB per launch, collected in separate runs; on gfx950 FETCH_SIZE counts half
    the bytes of a 16 B/lane coalesced stream, so the sweep's is doubled -- MI355X_MICROARCH.md, HBM section; the
    count kernel's 8-byte accesses are uncalibrated and taken as counted).  Only valid for the configuration the
    profile was taken on; otherwise null."""
    entry = PMC_FILES.get((args.workload, args.size))
    if world != 1 or entry is None or args.no_project or args.thr is not None:
        return None, None
    import csv
    name, scale, note = entry
    path = os.path.join(ROOT, "profiles", name)
    if not os.path.exists(path):
        return None, None
    total, seen = 0.0, set()
    for row in csv.DictReader(open(path)):
        k = next((p for p in PASS_KERNELS if p in row["kernel"]), None)
        if k is None:
            continue
        kb = float(row["mean_value_KB"])
        if row["counter"] == "FETCH_SIZE":
            total += kb * (2.0 if "classify" in k else 1.0)
            seen.add((k, "F"))
        elif row["counter"] == "WRITE_SIZE":
            total += kb
            seen.add((k, "W"))
    if not any(k.startswith("k_classify") for k, _ in seen) or not any(k.startswith("k_count") and f == "F" for k, f in seen):
        return None, None
    return total * 1024.0 * scale, os.path.relpath(path, ROOT) + ("" if note is None else " -- " + note)


def slab_probe(pkg, torch, ex, buf, n, dtype, prm, reps=20):
    """What ONE rank of an 8-GPU strong-scaling run does per step, measured on this GPU (no exchange, no collective: its
    device work and the host's share): slices [3n/8, 4n/8) of the resident volume as a THIN_HALO slab, through
    cuberille_step_begin / _end with itself as the only rank.  wall - device is what the host adds per step."""
    from midas_journal_740_amd.cuberille import minimum_halo
    whole = pkg.make_desc(dtype, (n, n, n))
    below, above = minimum_halo(whole, prm)
    a, b = 3 * n // 8, 4 * n // 8
    lo, hi = a - below - 1, b + above + 1
    desc = pkg.make_desc(dtype, (n, n, hi - lo))
    slab = pkg._abi.Slab(n, lo, a, b, 0, pkg._abi.SLAB_THIN_HALO, None, None)
    sub = buf[lo:hi]
    wall = dev = 0.0
    escaped = retries = 0
    for i in range(reps + 2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ptr, _ = ex.step_begin(sub.data_ptr(), desc, prm, slab)
        res, done = ex.step_end(ptr, 1, 0)
        t1 = time.perf_counter()
        if not done:
            # CUBERILLE_RETRY: some flag in the row.  The probe has no neighbours to ask: continue with the synchronous
            # calls as a driver would -- the vertex phase again unless quirk Q1 wants a source slice from below (then
            # nothing occupied lies below in this volume: the count stands), escaped walks in the whole volume
            retries += 1
            info = ex.slab_info()
            if not info.alias_source_below_buffer:
                ex.emit_points()
                escaped = ex.escaped_count()
                if escaped:
                    ex.reproject_escaped(buf.data_ptr(), 0, n)
            res = ex.emit(0)
        elif i >= 2:
            wall += t1 - t0
            dev += res.ms_total
    k = max(reps - retries, 1)
    return {"retries": retries, "slices": [a, b], "halo": [below + 1, above + 1], "points": int(res.n_points), "cells": int(res.n_cells),
            "wall_ms": round(wall / k * 1e3, 4), "device_ms": round(dev / k, 4),
            "wall_minus_device_ms": round(wall / k * 1e3 - dev / k, 4), "host_waits_per_step": 1,
            "halo_bytes_a_rank_would_receive": (below + above + 2) * n * n * int(np.dtype(dtype).itemsize),
            "escaped_walks": int(escaped)}


def series_probe(pkg, torch, ex, buf, desc, prm, device_index, volumes=20):
    """A SERIES of volumes through two contexts (two streams), the next one's step opened (cuberille_step_begin returns without
    waiting) before the previous one's is closed: the next sweep and the previous tail share the GPU.  Beside the line's
    `value`, which stays one extraction after the other on one context.  Here the series is the resident volume again and
    again; what it shows is the ceiling of overlapping the HBM-bound sweep with the f64-bound walk."""
    ex2 = pkg.Extractor(device_index)
    try:
        for _ in range(2):             # both contexts with this volume's sizes behind them (the slab probe left others)
            ex.extract_device(buf.data_ptr(), desc, prm)
            ex2.extract_device(buf.data_ptr(), desc, prm)
        ctx = (ex, ex2)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        row = ctx[0].step_begin(buf.data_ptr(), desc, prm)[0]
        for i in range(volumes):
            nxt = ctx[(i + 1) & 1].step_begin(buf.data_ptr(), desc, prm)[0] if i + 1 < volumes else None
            res, done = ctx[i & 1].step_end(row, 1, 0)
            if not done:
                return {"error": "a step came back with CUBERILLE_RETRY"}
            row = nxt
        dt = time.perf_counter() - t0
        return {"volumes": volumes, "ms_per_volume": round(dt / volumes * 1e3, 4), "contexts": 2,
                "points": int(res.n_points), "cells": int(res.n_cells)}
    finally:
        ex2.close()


def first_calls_probe(pkg, torch, buf, desc, prm, device_index, calls=4):
    """The first extractions on a FRESH context (set up, as the drop-in filter's constructor does it: cuberille_warm_up): the
    launch shapes of an extraction follow the previous one on its context (the count from memory / in its dense form, waves
    of the walk that refill at 16 idle lanes / only when empty, launches sized without the host learning the counts), so a
    context's first call is not its steady state.  Results never depend on any of it; this is what it costs."""
    ex2 = pkg.Extractor(device_index)
    try:
        ex2.warm_up(desc)
        torch.cuda.synchronize()
        wall, dev, count = [], [], []
        ex2.debug_option("stage_timing", 1)
        for _ in range(calls):
            t0 = time.perf_counter()
            r = ex2.extract_device(buf.data_ptr(), desc, prm)
            wall.append(round((time.perf_counter() - t0) * 1e3, 4))
            dev.append(round(r.ms_total, 4))
            count.append(round(r.ms_count, 4))
        return {"wall_ms": wall, "device_ms": dev, "count_stage_ms": count,
                "what": "call 1, 2, ... on a fresh context after cuberille_warm_up, per-stage events on"}
    finally:
        ex2.close()


def cpu_baseline(pkg, torch, args, device, gpu_mesh=None, gpu_iterations=None):
    """The oracle restatement of the reference ("port"), timed on this box's host cores on a
    bounded sample of the same workload (the reference itself needs ITK).
    When the sample IS the bench volume (the default) the mesh the oracle just built is also held against the mesh of
    the timed region, byte for byte: the second return value (parity_at_bench_size), None otherwise."""
    oracle = graft.load_oracle()
    oracle.build()
    n = args.cpu_sample
    dtype, iso, thr = WORKLOADS[args.workload]
    vol = generate_block(pkg, torch, args.workload, n, 0, n, None, device).cpu().numpy()
    try:
        cores = len(os.sched_getaffinity(0))      # the cores this process may really use
    except AttributeError:
        cores = os.cpu_count() or 1
    t0 = time.perf_counter()
    m = oracle.run(vol, iso, triangles=True, project=True, threshold=thr, step=0.25, relax=0.95, max_steps=50,
                   gradient_threads=cores, faithful_cells=True)
    dt = time.perf_counter() - t0
    secs = m.info["seconds_gradient"] + m.info["seconds_sweep"]
    parity = None
    if gpu_mesh is not None and n == args.size:
        import hashlib
        same_shape = gpu_mesh.points.shape == m.points.shape and gpu_mesh.cells.shape == m.cells.shape
        same_cells = bool(same_shape and np.array_equal(gpu_mesh.cells, m.cells))
        same_points = bool(same_shape and np.array_equal(gpu_mesh.points.view(np.uint32), m.points.view(np.uint32)))
        parity = {"identical": bool(same_cells and same_points and gpu_iterations == m.info["proj_iterations"]),
                  "cells_identical": same_cells, "point_bits_identical": same_points,
                  "walk_passes": [int(gpu_iterations), int(m.info["proj_iterations"])],
                  "points_sha256": hashlib.sha256(gpu_mesh.points.tobytes()).hexdigest()[:16],
                  "cells_sha256": hashlib.sha256(gpu_mesh.cells.tobytes()).hexdigest()[:16],
                  "oracle_points_sha256": hashlib.sha256(m.points.tobytes()).hexdigest()[:16],
                  "oracle_cells_sha256": hashlib.sha256(m.cells.tobytes()).hexdigest()[:16],
                  "what": "the mesh on the device after the timed region (downloaded) against the oracle's mesh of the same "
                          "volume: cell ids and order, float bits of every coordinate, passes through the walk loop"}
    return parity, {
        "value": round(n ** 3 / secs / 1e6, 3), "unit": "Mvoxels/s", "cores": cores, "kind": "port",
        "sample": "%s %d^3 %s (%s, same generator and parameters); sweep single-threaded like the reference, "
                  "gradient pre-pass on %d threads like ITK; %.1f s gradient + %.1f s sweep, %d points / %d cells; "
                  "wall %.1f s" % (
                      args.workload, n, np.dtype(dtype).name,
                      "the bench volume itself" if n == args.size else "%.3g x fewer voxels than the GPU workload" % ((args.size / float(n)) ** 3),
                      cores, m.info["seconds_gradient"], m.info["seconds_sweep"], len(m.points), len(m.cells), dt),
    }


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launcher_command(n, argv, port):
    """The command `bench.py --gpus N` starts when nobody wrapped it in torch.distributed.run: the contract's own launch
    line, one rank per GPU, the same arguments."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


# the ladder a self-launched N>1 run goes down when an attempt ends without a JSON line (non-zero exit, or no end within
# --launch-timeout): the default step first, then the protocol with the host in the loop and the full halo, equal slabs
# (nothing blind, nothing on a side stream, no second communicator).  A slower curve is worth more than none.
LAUNCH_LADDER = [
    ([], None),
    (["--host-offsets", "--full-halo", "--partition", "uniform"],
     "the default N>1 step did not finish (%s): plain protocol -- host-side offsets, full halo, slabs of equal thickness"),
]


def self_launch(args, argv):
    """`python bench.py --gpus N` from a bare shell (WORLD_SIZE unset, N > 1): start the ranks as a CHILD process -- this
    process never touches HIP, so nothing that has initialised the GPU is ever replaced -- pass their output through,
    return their exit code.  The JSON line is the child's rank 0's."""
    import signal
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    rc, why = 1, ""
    for extra, note in LAUNCH_LADDER:
        cmd = launcher_command(args.gpus, argv + extra + (["--fallback-note", note % why] if note else []), free_port())
        print("bench.py: starting %d ranks: %s" % (args.gpus, " ".join(cmd)), file=sys.stderr, flush=True)
        proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, start_new_session=True, text=True)
        try:
            out, _ = proc.communicate(timeout=args.launch_timeout)
            rc = proc.returncode
            why = "exit code %d" % rc
        except subprocess.TimeoutExpired:
            # the whole process group of the launcher: its ranks must not outlive it on the GPUs
            try:
                os.killpg(proc.pid, signal.SIGTERM)
                try:
                    proc.wait(timeout=20)
                except subprocess.TimeoutExpired:
                    os.killpg(proc.pid, signal.SIGKILL)
            except ProcessLookupError:
                pass
            out, _ = proc.communicate()
            rc, why = 124, "no end within %d s" % args.launch_timeout
        lines = [ln for ln in (out or "").splitlines() if ln.startswith("{") and '"metric"' in ln]
        sys.stdout.write(out or "")
        sys.stdout.flush()
        if rc == 0 and lines:
            return 0
        if lines:            # a line was printed and the run failed afterwards (e.g. a parity check): that is the result
            return rc
        print("bench.py: %d ranks ended without a result (%s)" % (args.gpus, why), file=sys.stderr, flush=True)
        if args.no_launch_fallback:
            break
    return rc or 1


class Watchdog:
    """A wall-clock guard around a stretch that can only hang, never raise (the first collectives of an N>1 run): if it
    is not left within `seconds`, say where, and end the process with exit code 3 -- torch.distributed.run then ends the
    other ranks -- instead of sitting in a collective until somebody's patience runs out."""

    def __init__(self, seconds, what):
        import threading
        self.what, self.seconds = what, seconds
        self._done = threading.Event()
        self._t = threading.Thread(target=self._run, daemon=True)

    def _run(self):
        if not self._done.wait(self.seconds):
            sys.stderr.write("bench.py: rank %s: %s did not finish within %d s -- giving up (exit 3)\n" % (
                os.environ.get("RANK", "0"), self.what, self.seconds))
            sys.stderr.flush()
            os._exit(3)

    def __enter__(self):
        if self.seconds > 0:
            self._t.start()
        return self

    def __exit__(self, *exc):
        self._done.set()
        return False


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--workload", default="marschner_lobb", choices=sorted(WORKLOADS))
    ap.add_argument("--cpu-sample", type=int, default=-1,
                    help="edge of the cube the CPU baseline is timed on (default: --size, i.e. the bench volume itself; 0 = skip)")
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"],
                    help="N>1: strong = one size^3 volume cut into N slabs (default), weak = one size^3 block per rank")
    ap.add_argument("--check-against-single", action="store_true",
                    help="N>1: after the timed region gather the mesh on rank 0 and compare it, bit for bit, with a one-shot "
                         "extraction of the whole volume on rank 0's GPU")
    ap.add_argument("--no-project", action="store_true")
    ap.add_argument("--gather-mesh", action="store_true",
                    help="N>1: after the timed region also concatenate the rank parts on rank 0 and report its time")
    ap.add_argument("--thr", type=float, default=None, help="override the projection threshold (experiments)")
    ap.add_argument("--full-halo", action="store_true", help="N>1: exchange the whole projection-reach halo every step")
    ap.add_argument("--host-offsets", action="store_true",
                    help="N>1: the count all-gather through the host (two more host waits per step) instead of device-resident rows")
    ap.add_argument("--bits-first", action="store_true",
                    help="N>1: the neighbours' BIT PLANES cross right behind the owned sweep, on a communicator of their own, and the "
                         "count waits for those alone; only the walk waits for the halo's voxels (DESIGN.md section 6: rehearsed "
                         "bit-identical, never run under RCCL -- a switch until a node has timed it against the default)")
    ap.add_argument("--opt", action="append", default=[],
                    help="development: name=value for cuberille_debug_set_option (kernel variants; results never depend on them)")
    ap.add_argument("--partition", default="balanced", choices=["balanced", "uniform"],
                    help="N>1: z-slabs of equal WORK, cut from the per-slice work a calibration step measures before the "
                         "timed region (default), or of equal thickness")
    ap.add_argument("--calibration-rounds", type=int, default=2,
                    help="N>1, --partition balanced: calibrate-and-cut rounds before the timed region (each on the cuts of the one before)")
    ap.add_argument("--no-warm-up", action="store_true",
                    help="skip cuberille_warm_up (its toy extraction launches a few tiny kernels: profiles/collect.sh keeps them out "
                         "of the per-kernel averages this way; the untimed warm-up steps do the warming then)")
    ap.add_argument("--no-slab-probe", action="store_true",
                    help="N=1: skip the extra measurement of one 1/8 slab (what a rank of an 8-GPU run does per step)")
    ap.add_argument("--launch-timeout", type=int, default=600,
                    help="--gpus N > 1 from a bare shell: seconds one attempt of the self-started ranks may take")
    ap.add_argument("--no-launch-fallback", action="store_true",
                    help="--gpus N > 1 from a bare shell: do not try the plain protocol when the default step ends without a line")
    ap.add_argument("--step-timeout", type=int, default=240,
                    help="N>1: seconds the start of the process group, the first step, or the timed region may take before "
                         "the rank gives up with exit code 3 (0 = wait forever)")
    ap.add_argument("--fallback-note", default=None, help=argparse.SUPPRESS)
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # a bare `python bench.py --gpus N`: the ranks are started HERE, as a child process, before this process has
        # imported torch or touched HIP (a process that has initialised the GPU is never replaced by another program)
        raise SystemExit(self_launch(args, sys.argv[1:]))

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d "
                         "(or with no launcher at all: bench.py starts its own ranks)" % (args.gpus, world, args.gpus))

    def guard(what, scale=1):
        return Watchdog(args.step_timeout * scale if world > 1 else 0, what)
    # CUBERILLE_BENCH_REHEARSAL=1: all ranks share GPU 0 and talk over gloo -- a functional rehearsal of
    # the N>1 path on a one-GPU box (never a measurement)
    rehearsal = os.environ.get("CUBERILLE_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        with guard("the start of the process group"):
            import datetime
            # (the collectives' own limit: torch's watchdog ends a rank whose RCCL collective has not completed by then,
            #  naming the collective; the StepMonitor of the driver says earlier, and in the path's own words, who waits where)
            limit = datetime.timedelta(seconds=max(2 * args.step_timeout, 120)) if args.step_timeout > 0 else None
            if rehearsal:
                dist.init_process_group("gloo", timeout=limit)
            else:
                dist.init_process_group("nccl", device_id=device, timeout=limit)
    pkg = graft.load_package()
    # the library is built by __graft_entry__.build(); here only make sure it exists, and never let
    # several ranks run the compiler on the same output at once
    if not os.path.exists(pkg._abi.LIB_PATH):
        if local_rank == 0:
            pkg._abi.build()
        if world > 1:
            dist.barrier()
    from midas_journal_740_amd.distributed import ShardedExtractor

    n = args.size
    if args.cpu_sample < 0:
        args.cpu_sample = n
    dtype, iso, thr = WORKLOADS[args.workload]
    if args.thr is not None:
        thr = args.thr
    strong = world == 1 or args.scaling == "strong"
    gnz = n if strong else n * world
    ex = pkg.Extractor(local_rank)
    for kv in args.opt:
        ex.debug_option(kv.split("=")[0], int(kv.split("=")[1]))
    # (weak mode stacks copies of the block along z; for Marschner-Lobb every copy ends in a run of empty slices, across
    #  which the reference re-uses vertex ids -- quirk Q1, DESIGN.md -- at every slab boundary: the driver hands the
    #  source slices between the ranks, two more small exchanges per step)
    prm = pkg.make_params(iso, triangles=True, project=not args.no_project, threshold=thr, step=0.25, relax=0.95,
                          max_steps=50)
    mon_timeout = args.step_timeout if (world > 1 and args.step_timeout > 0) else None
    sh = ShardedExtractor(ex, (n, n, gnz), dtype, rank, world, params=prm, thin_halo=not args.full_halo,
                          device_offsets=not args.host_offsets, bits_first=args.bits_first and not args.host_offsets,
                          step_timeout=mon_timeout, abort_on_timeout=True)
    if not args.no_warm_up:
        ex.warm_up(sh.desc)         # code objects, workspace and staging ring for this rank's buffer: before any step
    period = None if strong else n
    if args.workload == "sphere" and not strong:
        raise SystemExit("sphere workload: strong scaling or one GPU only")
    buf = generate_block(pkg, torch, args.workload, n, sh.lo, sh.hi, period, device)
    if world > 1:       # halos arrive through the exchange, every step
        buf[:sh.z0 - sh.lo].zero_()
        buf[sh.z1 - sh.lo:].zero_()
    torch.cuda.synchronize()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    res = None
    fallback = args.fallback_note
    first_step = guard("the first step of the N>1 path (the first halo exchange and row all-gather between the ranks)")
    first_step.__enter__()
    if world > 1 and not args.host_offsets:
        # the first step of the N>1 path on this node: should the one-wait step (device-resident rows, thin halo) raise on
        # ANY rank, every rank learns of it here and all take the plain protocol (host in the loop, full halo) instead -- a
        # slower curve is worth more than none.  (A hang cannot be caught: that is what the rehearsals are for.)
        try:
            res = sh.extract(buf, prm)
            mine, why = 1, ""
        except Exception as e:       # noqa: BLE001
            mine, why = 0, "%s: %s" % (type(e).__name__, e)
        okt = torch.tensor([mine], dtype=torch.int32, device="cpu" if rehearsal else device)
        dist.all_reduce(okt, op=dist.ReduceOp.MIN)
        if int(okt.item()) == 0:
            fallback = "one-wait step failed on some rank (%s): host-side offsets, full halo" % (why or "another rank")
            args.host_offsets, args.full_halo, args.bits_first = True, True, False
            ex.use_own_stream()
            sh = ShardedExtractor(ex, (n, n, gnz), dtype, rank, world, params=prm, thin_halo=False, device_offsets=False,
                                  step_timeout=mon_timeout, abort_on_timeout=True)
            res = None
    for _ in range(args.warmup):
        res = sh.extract(buf, prm)
    first_step.__exit__(None, None, None)
    partition = "uniform"
    rest = guard("the calibration, the timed region or what follows it", scale=4)
    rest.__enter__()
    if world > 1 and strong and args.partition == "balanced":
        # the surface of a volume is rarely spread evenly over z (this field's rippled sheet lies in a sixth of the
        # slices): one calibration step on slabs of equal thickness measures what every slice costs, the slabs of the
        # timed region are cut for equal work.  Outside the timed region, like any warm-up; the same volume, other cuts.
        # (a vertex does not cost the same everywhere -- the walks of this field's flat caps at the bottom of the volume run
        #  out of steps, those of the sheet converge in a few passes: measured on one GPU, profiles/r4_slab_stages.log -- and a
        #  rank's time is spread over its slices by vertices created: a second round on the first round's cuts refines them)
        from midas_journal_740_amd.distributed import balanced_bounds
        if res is None:
            res = sh.extract(buf, prm)
        for _round in range(max(args.calibration_rounds, 1)):
            bounds = balanced_bounds(sh.slice_work(res), world)
            del buf
            sh = ShardedExtractor(ex, (n, n, gnz), dtype, rank, world, params=prm, thin_halo=not args.full_halo,
                                  device_offsets=not args.host_offsets, bounds=bounds,
                                  bits_first=args.bits_first and not args.host_offsets,
                          step_timeout=mon_timeout, abort_on_timeout=True)
            buf = generate_block(pkg, torch, args.workload, n, sh.lo, sh.hi, period, device)
            buf[:sh.z0 - sh.lo].zero_()
            buf[sh.z1 - sh.lo:].zero_()
            torch.cuda.synchronize()
            for _ in range(max(args.warmup, 2)):
                res = sh.extract(buf, prm)
        partition = "balanced"
    stage_keys = ["ms_classify", "ms_count", "ms_scan", "ms_emit_points", "ms_project", "ms_emit_cells", "ms_total"]
    # the timed region carries the two event pairs every extraction has (the pass over the volume, the emit phase);
    # the per-stage events cost the stream about 8 us each and are switched on for a few extra extractions afterwards
    live = {"ms_pass": 0.0, "ms_total": 0.0}
    host_side = {"halo_bytes": 0, "halo_bit_bytes": 0, "host_syncs": 0, "collectives": 0, "escaped": 0, "deep_halo_fetched": 0}
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = sh.extract(buf, prm)
        for k in live:
            live[k] += getattr(res, k)
        for k in host_side:
            host_side[k] += int(sh.stats[k])
    barrier()
    dt = time.perf_counter() - t0
    acc = {k: 0.0 for k in stage_keys}
    ex.debug_option("stage_timing", 1)
    for _ in range(STAGE_REPS):
        r2 = sh.extract(buf, prm)
        for k in stage_keys:
            acc[k] += getattr(r2, k)
    ex.debug_option("stage_timing", 0)
    barrier()
    if world > 1:
        cdev = "cpu" if rehearsal else device
        t = torch.tensor([dt], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        tot = torch.tensor([int(res.n_points), int(res.n_cells)], dtype=torch.int64, device=cdev)
        dist.all_reduce(tot)
        n_points, n_cells = int(tot[0]), int(tot[1])
    else:
        n_points, n_cells = int(res.n_points), int(res.n_cells)

    gather_ms = None
    single_check = None
    if world > 1 and (args.gather_mesh or args.check_against_single):
        # outside the timed region: mesh concatenation in rank order on rank 0 (device to device over RCCL)
        barrier()
        t0 = time.perf_counter()
        whole = sh.gather_mesh(dst=0)
        gather_ms = (time.perf_counter() - t0) * 1e3
        if rank == 0:
            assert (whole.GetNumberOfPoints(), whole.GetNumberOfCells()) == (n_points, n_cells)
            if args.check_against_single:
                # the same volume in one piece on this rank's GPU: the slabs must give its buffers bit for bit
                import hashlib
                full = generate_block(pkg, torch, args.workload, n, 0, gnz, period, device)
                torch.cuda.synchronize()           # the library runs on its own stream
                one = pkg.Extractor(local_rank)
                one.extract_device(full.data_ptr(), pkg.make_desc(dtype, (n, n, gnz)), prm)
                ref = one.download()
                same = (np.array_equal(ref.cells, whole.cells)
                        and np.array_equal(ref.points.view(np.uint32), whole.points.view(np.uint32)))
                single_check = {"identical_to_one_shot": bool(same),
                                "cells_sha256": hashlib.sha256(whole.cells.tobytes()).hexdigest()[:16],
                                "points_sha256": hashlib.sha256(whole.points.tobytes()).hexdigest()[:16]}
                one.close()
                del full, ref
                assert same, "slab decomposition differs from the one-shot extraction"
        del whole

    if rank == 0:
        voxels = float(n) * n * gnz
        stages = {k: acc[k] / STAGE_REPS for k in stage_keys}
        # B_A = Nx*Ny*Nz*sizeof(pixel), every voxel this rank's launch reads counted once
        launch_slices = (sh.thi - sh.tlo) if (world > 1 and sh.thin) else (sh.hi - sh.lo)
        alg_bytes = float(n) * n * launch_slices * np.dtype(dtype).itemsize
        classify_gbs = alg_bytes / (stages["ms_classify"] * 1e-3) / 1e9
        pass_ms = live["ms_pass"] / args.steps          # HIP events around the pass, every step of the timed region
        if pass_ms <= 0.0:                              # (volumes of a few M voxels get one event pair only: take the stage events)
            pass_ms = stages["ms_classify"] + stages["ms_count"]
        pass_gbs = alg_bytes / (pass_ms * 1e-3) / 1e9
        traffic, traffic_src = measured_traffic(args, world, alg_bytes)
        out = {
            "metric": "Mvoxels/s polygonized + achieved HBM GB/s, 1024^3 float32 @1/2/4/8 GPU",
            "value": round(voxels * args.steps / dt / 1e6, 1),
            "unit": "Mvoxels/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "strong" if strong else "weak",
            "vs_baseline": None,
            "dtype": "f32" if dtype == np.float32 else "u8",
            "data": "synthetic" + (" (REHEARSAL: ranks share one GPU over gloo, not a measurement)" if rehearsal else ""),
            "config": {"workload": "%s %dx%dx%d %s iso=%g, triangles+projection (thr %g, step 0.25, relax 0.95, max 50)"
                                   % (args.workload, n, n, gnz, np.dtype(dtype).name, iso, thr),
                       "per_gpu": "%dx%dx%d slab + %s-slice halo%s" % (
                           n, n, sh.z1 - sh.z0, ("%d+%d" % sh.thin) if sh.thin else str(sh.halo if world > 1 else 0),
                           " (rest of the %d-slice halo on demand)" % sh.halo if sh.thin else ""),
                       "parallelism": "zslab%d" % world,
                       "partition": partition + (" (slices per rank %s, cut for equal work from a calibration step before "
                                                 "the timed region)" % [b - a for a, b in sh.bounds] if partition == "balanced" else ""),
                       "points": n_points, "cells": n_cells,
                       "projection_iterations_rank0": int(res.proj_iterations)},
            # the HBM-bound pass the north star's target is defined on: threshold sweep +
            # count + prefix sums, three launches; algorithmic bytes over the HIP-event duration of the pass on the
            # library's own stream, averaged over the steps of the timed region
            "roofline": {"bound": "hbm",
                         "kernel": "classify+count+scan pass: k_classify_span/_flat + k_count + k_block_scan",
                         "achieved": round(pass_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(pass_gbs / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "traffic_source": traffic_src, "algorithmic_bytes": alg_bytes, "pass_ms": round(pass_ms, 4),
                         "sweep_kernel_alone": {"ms": round(stages["ms_classify"], 4), "achieved": round(classify_gbs, 1),
                                                "frac": round(classify_gbs / HBM_PEAK_GBS, 4)}},
            "stages_ms": {k: round(v, 4) for k, v in stages.items()},
            "stages_ms_source": "%d extractions after the timed region with the per-stage events on "
                                "(cuberille_debug_set_option stage_timing); ms_total of the timed region: %.4f"
                                % (STAGE_REPS, live["ms_total"] / args.steps),
        }
        # what a step costs besides kernels (rank 0's view; per step, averaged over the timed region)
        out["host_side"] = {"halo_bytes_per_rank": host_side["halo_bytes"] // args.steps,
                            "halo_bit_plane_bytes_per_rank": host_side["halo_bit_bytes"] // args.steps,
                            "halo_order": "bit planes first (count), voxels for the walk" if sh.bits_first else "voxels",
                            "host_syncs_per_step": round(host_side["host_syncs"] / args.steps, 2),
                            "collectives_per_step": round(host_side["collectives"] / args.steps, 2),
                            "escaped_walks_per_step": round(host_side["escaped"] / args.steps, 2),
                            "steps_that_fetched_the_deep_halo": host_side["deep_halo_fetched"],
                            "wall_minus_device_ms": round(dt / args.steps * 1e3 - live["ms_total"] / args.steps, 4)}
        out["roofline"]["note"] = ("largest kernel by time is the projection walk (%.0f %% of device time): f64 VALU-bound, "
                                   "neither an HBM nor an MFMA roofline applies to it" % (
                                       100.0 * stages["ms_project"] / stages["ms_total"]))
        if fallback is not None:
            out["fallback"] = fallback
        if single_check is not None:
            out["check_against_single"] = single_check
        if gather_ms is not None:
            out["gather_mesh_ms"] = round(gather_ms, 1)
        mesh = None
        if world == 1:
            # outside the timed region, reported separately: copying the mesh to the host
            # (1) into host memory the context owns and keeps (cuberille_mesh_host: what the drop-in filter fills its
            # itk::Mesh from) -- the first mesh of the process, then, after one more extraction, the same buffers again;
            # (2) into arrays of the caller's, freshly allocated (cuberille_mesh_download)
            t0 = time.perf_counter()
            view = ex.mesh_host()
            out["d2h_mesh_ms"] = round((time.perf_counter() - t0) * 1e3, 1)
            out["mesh_bytes"] = int(view.points.nbytes + view.cells.nbytes)
            sh.extract(buf, prm)
            t0 = time.perf_counter()
            view = ex.mesh_host()
            again = time.perf_counter() - t0
            out["d2h_mesh_again_ms"] = round(again * 1e3, 1)
            out["d2h_mesh_GBps"] = round(out["mesh_bytes"] / again / 1e9, 1)
            del view
            t0 = time.perf_counter()
            mesh = ex.download()
            out["d2h_mesh_into_fresh_caller_arrays_ms"] = round((time.perf_counter() - t0) * 1e3, 1)
        if world == 1 and not args.no_slab_probe and n >= 64:
            try:                     # a probe beside the measurement: whatever it runs into, the line is printed
                out["slab_eighth_probe"] = slab_probe(pkg, torch, ex, buf, n, dtype, prm)
            except Exception as e:   # noqa: BLE001
                out["slab_eighth_probe"] = {"error": "%s: %s" % (type(e).__name__, e)}
        if world == 1 and not args.no_slab_probe and n >= 256:
            try:
                out["series_two_contexts"] = series_probe(pkg, torch, ex, buf, sh.desc, prm, local_rank, max(args.steps, 10))
            except Exception as e:   # noqa: BLE001
                out["series_two_contexts"] = {"error": "%s: %s" % (type(e).__name__, e)}
        if world == 1 and not args.no_slab_probe and n >= 256:
            try:
                out["first_calls_on_a_fresh_context"] = first_calls_probe(pkg, torch, buf, sh.desc, prm, local_rank)
            except Exception as e:   # noqa: BLE001
                out["first_calls_on_a_fresh_context"] = {"error": "%s: %s" % (type(e).__name__, e)}
        parity = None
        if world == 1 and args.cpu_sample > 0:
            parity, out["cpu_baseline"] = cpu_baseline(pkg, torch, args, device, mesh, int(res.proj_iterations))
            if parity is not None:
                out["parity_at_bench_size"] = parity
        print(json.dumps(out), flush=True)
        if parity is not None and not parity["identical"]:
            raise SystemExit("bench: the mesh of the timed region differs from the oracle's at the bench size")
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    rest.__exit__(None, None, None)


if __name__ == "__main__":
    main()
