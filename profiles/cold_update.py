#!/usr/bin/env python3
"""What the reference's own probe prints: one cold Update() per process (Testing/CuberilleTest01.cxx:158-160,190,
"Polygonization took ... seconds"), for the reference's volumes at its CTest settings, through the UNCHANGED driver built
against the drop-in header -- beside the oracle's time for the same volume and parameters on this host.

  python profiles/cold_update.py [--exe NAME=PATH ...] [--runs 3] > profiles/rN_cold_update.log

--exe may be given several times (e.g. a binary kept from before a change, for a same-box A/B).  Every run is a fresh
process; the driver constructs the filter and sets its input before it starts its clock, as the reference's does."""
import argparse
import json
import os
import re
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--exe", action="append", default=[])
    ap.add_argument("--runs", type=int, default=3)
    ap.add_argument("--cases", default="Cuberille_Blob0_00,Cuberille_Nucleon_01,Cuberille_Fuel_01,Cuberille_MarschnerLobb_01,"
                                       "Cuberille_Silicium_01,Cuberille_HydrogenAtom_01,Cuberille_Neghip_01")
    args = ap.parse_args()
    exes = [e.split("=", 1) for e in args.exe] or [["drop-in", os.path.join(ROOT, "midas-journal-740_amd", "itk", "build", "CuberilleTest01")]]
    cases = {c["name"]: c for c in json.load(open(os.path.join(ROOT, "tests", "golden", "ctest_cases.json")))}
    pkg = graft.load_package()
    oracle = graft.load_oracle()
    oracle.build()
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    print("# one Update() per fresh process: 'Polygonization took' of the unchanged CuberilleTest01.cxx, seconds -> ms; %d runs each" % args.runs)
    print("# oracle: gradient pre-pass on %d threads + single-threaded sweep, best of 3 in one process" % cores)
    tmp = tempfile.mkdtemp()
    for name in args.cases.split(","):
        c = cases[name]
        path = os.path.join(ROOT, "tests", "golden", "data", c["input"])
        vol = pkg.read_mha(path)
        kw = dict(triangles=c["triangles"], project=c["project"], threshold=c["threshold"], step=c["step"], relax=c["relax"],
                  max_steps=c["max_steps"])
        best = 1e9
        for _ in range(3):
            m = oracle.run(vol.voxels, c["iso"], gradient_threads=cores, faithful_cells=True, **kw)
            best = min(best, m.info["seconds_gradient"] + m.info["seconds_sweep"])
        line = "%-28s %-16s %3dx%3dx%3d  %6d pts  oracle %8.2f ms |" % (name, c["input"], *vol.dims, len(m.points), best * 1e3)
        for label, exe in exes:
            took, wall = [], []
            for _ in range(args.runs):
                argv = [exe, "Test01", path, os.path.join(tmp, "o.vtk"), str(c["iso"]), str(c["points"]), str(c["cells"]),
                        str(c["triangles"]), str(c["project"]), repr(c["threshold"]), repr(c["step"]), repr(c["relax"]), str(c["max_steps"])]
                t0 = time.perf_counter()
                r = subprocess.run(argv, capture_output=True, text=True, timeout=300)
                wall.append(time.perf_counter() - t0)
                if r.returncode != 0:
                    raise SystemExit("%s failed: %s %s" % (label, r.stdout[-300:], r.stderr[-300:]))
                took.append(float(re.search(r"Polygonization took ([0-9.eE+-]+) seconds", r.stdout).group(1)))
            line += "  %s: Update() %s ms (process %s ms) |" % (label, " ".join("%.2f" % (t * 1e3) for t in took),
                                                                 " ".join("%.0f" % (w * 1e3) for w in wall))
        print(line, flush=True)


if __name__ == "__main__":
    main()
