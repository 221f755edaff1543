#!/usr/bin/env python3
"""The multi-GPU step (midas-journal-740_amd/distributed.py) under random volumes: W processes share ONE GPU over gloo, every case
cuts a fresh random volume into W Z-slabs (equal or random bounds), runs ShardedExtractor.extract in a random protocol (host in
the loop / device-resident offsets, full / thin halo, bits first, event path, a second step on the same contexts) and holds the
gathered mesh against the oracle's mesh of the whole volume -- ids, cell order, float bits.  Blanked slices make quirk Q1 cross the
cuts (alias planes, recounts), long steps make walks leave the thin halo (escapes).  Run by hand on a GPU box:

    python tests/fuzz_ranks.py --world 3 --seconds 240 --seed 1

TEST INFRASTRUCTURE: the oracle is the checker.  One progress line every ~20 s from the last rank; exit code 1 with the case's recipe
on a difference (or on a rank that raised)."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def worker(rank, world, port, seconds, seed, out_path, replay=-1, start=0):
    import torch
    import torch.distributed as dist
    import __graft_entry__ as graft
    from conftest import assert_same_mesh
    import fuzz_campaign as fz
    pkg = graft.load_package()
    from midas_journal_740_amd.distributed import ShardedExtractor
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    oracle = None
    if rank == world - 1:
        oracle = graft.load_oracle()
        oracle.build()
    rng = np.random.default_rng(seed)          # the same stream on every rank: the same cases
    ex = pkg.Extractor(0)
    stats = {"cases": 0, "points": 0, "cells": 0, "modes": {}, "escaped": 0, "second_steps": 0, "q1_cases": 0}
    t0 = last = time.time()
    rc = 0
    case = -1
    try:
        while True:
            # all ranks agree on going on (rank 0's clock)
            go = torch.tensor([1 if (time.time() - t0 < seconds or case < replay or case < start) else 0])
            if case >= replay - 1 and case >= start - 1:
                dist.broadcast(go, 0)
            if not int(go[0]):
                break
            case += 1
            nx = int(rng.choice([5, 31, 64, 65, 100, 128, 200]))
            ny = int(rng.integers(2, 33))
            nz = int(rng.integers(4 * world, 70))
            dt = fz.DTYPES[int(rng.integers(0, len(fz.DTYPES)))]
            vox, iso = fz.draw_field(rng, (nz, ny, nx), dt)
            if rng.random() < 0.35:                # more empty slices than the single-GPU campaign draws: Q1 across the cuts
                for _ in range(int(rng.integers(1, 4))):
                    z = int(rng.integers(0, nz))
                    vox[z:z + int(rng.integers(1, 4))] = vox.min()
            spacing, origin, direction = fz.draw_geometry(rng)
            if rng.random() < 0.5:
                direction = np.eye(3)
            mode = str(rng.choice(["sync", "step", "thin", "step_thin", "step_bits", "step_thin_bits"]))
            kw = dict(triangles=bool(rng.integers(0, 2)), project=True,
                      threshold=float(rng.choice([0.01, 0.2, 5.0])) * (1.0 if np.dtype(dt).kind != "f" else 0.05),
                      step=float(rng.choice([0.1, 0.25, 0.6])) * min(spacing), relax=float(rng.choice([0.9, 0.95, 1.0])),
                      max_steps=int(rng.choice([4, 25, 50])))
            event_path = bool(rng.integers(0, 2))
            random_bounds = bool(rng.random() < 0.5)
            bounds = None
            if random_bounds:
                inner = sorted(rng.choice(np.arange(1, nz), size=world - 1, replace=False).tolist())
                cuts = [0] + [int(v) for v in inner] + [nz]
                bounds = list(zip(cuts[:-1], cuts[1:]))
            two_steps = mode != "sync" and rng.random() < 0.4
            recipe = dict(case=case, seed=seed, world=world, shape=[nz, ny, nx], dtype=np.dtype(dt).name, iso=iso, mode=mode, kw=kw,
                          spacing=spacing, origin=origin, direction=np.asarray(direction).tolist(), bounds=bounds,
                          event_path=event_path, two_steps=two_steps)
            if case < replay or case < start:
                continue
            if case == replay:
                # the one case in every protocol, equal and drawn bounds: which of them differ
                for m2 in ["sync", "step", "thin", "step_thin", "step_bits", "step_thin_bits"]:
                    for ev in (False, True):
                        for bd in (bounds, None):
                            verdict = "ok"
                            try:
                                prm = pkg.make_params(iso, **kw)
                                sh = ShardedExtractor(ex, (nx, ny, nz), vox.dtype, rank, world, spacing=spacing, origin=origin,
                                                      direction=direction, check_aliasing=True, params=prm, thin_halo="thin" in m2,
                                                      device_offsets=m2.startswith("step"), bits_first=m2.endswith("_bits"), bounds=bd,
                                                      step_timeout=120, abort_on_timeout=True)
                                sh.force_event_path = ev
                                host = np.zeros((sh.hi - sh.lo, ny, nx), dtype=vox.dtype)
                                host[sh.z0 - sh.lo:sh.z1 - sh.lo] = vox[sh.z0:sh.z1]
                                res = sh.extract(fz.to_device(torch, host), prm)
                                whole = sh.gather_mesh(dst=world - 1, on_device=ev)
                                if rank == world - 1:
                                    ref = oracle.run(vox, iso, spacing=spacing, origin=origin, direction=direction, **kw)
                                    try:
                                        assert_same_mesh(whole, ref)
                                    except AssertionError as e:
                                        verdict = "DIFFERS %s" % str(e)[:120]
                            except Exception as e:  # noqa: BLE001
                                verdict = "RAISED %s: %s" % (type(e).__name__, str(e)[:200])
                            print("rank %d  mode %-15s event_path %-5s bounds %-28s %s  stats %s" % (rank, m2, ev, bd, verdict, dict(sh.stats)), flush=True)
                            dist.barrier()
                if rank == world - 1:
                    np.save(out_path + ".vox.npy", vox)
                    occ = (vox.reshape(nz, -1) >= iso).any(axis=1)
                    print("recipe", json.dumps(recipe), "occupied slices", "".join("1" if o else "0" for o in occ), flush=True)
                rc = 2
                break
            err = None
            whole = None
            sh = None
            prm = None
            try:
                prm = pkg.make_params(iso, **kw)
                sh = ShardedExtractor(ex, (nx, ny, nz), vox.dtype, rank, world, spacing=spacing, origin=origin, direction=direction,
                                      check_aliasing=True, params=prm, thin_halo="thin" in mode, device_offsets=mode.startswith("step"),
                                      bits_first=mode.endswith("_bits"), bounds=bounds, step_timeout=120, abort_on_timeout=True)
                sh.force_event_path = event_path
                host = np.zeros((sh.hi - sh.lo, ny, nx), dtype=vox.dtype)
                host[sh.z0 - sh.lo:sh.z1 - sh.lo] = vox[sh.z0:sh.z1]            # owned slices only: the exchange brings the halo
                buf = fz.to_device(torch, host)
                res = sh.extract(buf, prm)
                if two_steps:
                    buf[:sh.z0 - sh.lo].zero_()
                    buf[sh.z1 - sh.lo:].zero_()
                    keep = (int(res.n_points), int(res.n_cells), int(res.proj_iterations))
                    res = sh.extract(buf, prm)
                    assert (int(res.n_points), int(res.n_cells), int(res.proj_iterations)) == keep, "second step differs from the first"
                esc = int(sh.stats.get("escaped", 0))
                whole = sh.gather_mesh(dst=world - 1, on_device=event_path)
                if rank == world - 1:
                    ref = oracle.run(vox, iso, spacing=spacing, origin=origin, direction=direction, **kw)
                    assert_same_mesh(whole, ref)
                    stats["cases"] += 1
                    stats["points"] += int(ref.points.shape[0])
                    stats["cells"] += int(ref.cells.shape[0])
                    stats["modes"][mode] = stats["modes"].get(mode, 0) + 1
                    stats["second_steps"] += int(two_steps)
                    ins_any = (vox.reshape(nz, -1) != vox.min()).any(axis=1)
                    stats["q1_cases"] += int((~ins_any[1:-1]).any())
                stats["escaped"] += esc
            except Exception as e:  # noqa: BLE001
                err = "%s: %s" % (type(e).__name__, str(e)[:400])
            # every rank learns whether any rank failed (a rank that raised must not leave the others in a collective)
            flag = torch.tensor([1 if err else 0])
            dist.all_reduce(flag)
            if int(flag[0]):
                print(json.dumps({"FAILED": recipe, "rank": rank, "error": err, "counts": np.asarray(getattr(sh, "counts", [])).tolist(),
                                  "stats": dict(sh.stats) if sh is not None else None}), flush=True)
                # the same case again on the same contexts (same history): a state left behind fails again, a race may not
                for attempt in range(3 if prm is not None else 0):
                    verdict = "identical"
                    sh2 = None
                    try:
                        sh2 = ShardedExtractor(ex, (nx, ny, nz), vox.dtype, rank, world, spacing=spacing, origin=origin, direction=direction,
                                               check_aliasing=True, params=prm, thin_halo="thin" in mode, device_offsets=mode.startswith("step"),
                                               bits_first=mode.endswith("_bits"), bounds=bounds, step_timeout=120, abort_on_timeout=True)
                        sh2.force_event_path = event_path
                        host = np.zeros((sh2.hi - sh2.lo, ny, nx), dtype=vox.dtype)
                        host[sh2.z0 - sh2.lo:sh2.z1 - sh2.lo] = vox[sh2.z0:sh2.z1]
                        sh2.extract(fz.to_device(torch, host), prm)
                        w2 = sh2.gather_mesh(dst=world - 1, on_device=event_path)
                        if rank == world - 1:
                            try:
                                assert_same_mesh(w2, oracle.run(vox, iso, spacing=spacing, origin=origin, direction=direction, **kw))
                            except AssertionError as e2:
                                verdict = "DIFFERS " + str(e2)[:100]
                    except Exception as e2:  # noqa: BLE001
                        verdict = "RAISED %s: %s" % (type(e2).__name__, str(e2)[:200])
                    if rank == world - 1:
                        print("again %d: %s counts %s stats %s" % (attempt, verdict, np.asarray(getattr(sh2, "counts", [])).tolist(),
                                                                    dict(sh2.stats) if sh2 is not None else None), flush=True)
                    dist.barrier()
                rc = 1
                break
            if rank == world - 1 and time.time() - last > 20:
                last = time.time()
                print("t %.0f s: %d cases, %d points, %d cells, all identical" % (last - t0, stats["cases"], stats["points"], stats["cells"]), flush=True)
    finally:
        ex.close()
        if rank == world - 1 and rc == 0:
            stats.update(seconds=round(time.time() - t0, 1), seed=seed, world=world, identical=True)
            print(json.dumps(stats), flush=True)
        with open(out_path + ".%d" % rank, "w") as f:
            f.write(str(rc))
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=3)
    ap.add_argument("--seconds", type=float, default=240.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--start", type=int, default=0, help="draw the cases before this one without running them")
    ap.add_argument("--replay", type=int, default=-1, help="draw up to this case without running, then run IT in every protocol and say which differ")
    args = ap.parse_args()
    if not 2 <= args.world <= 5:
        raise SystemExit("--world 2..5 (the GPU box allows six processes on its card)")
    import tempfile
    import torch.multiprocessing as mp
    import bench
    port = bench.free_port()
    out = os.path.join(tempfile.mkdtemp(), "rc")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    mp.spawn(worker, args=(args.world, port, args.seconds, args.seed, out, args.replay, args.start), nprocs=args.world, join=True)
    rcs = [int(open(out + ".%d" % r).read()) for r in range(args.world)]
    return 1 if any(rcs) else 0


if __name__ == "__main__":
    sys.exit(main())
