#!/usr/bin/env python3
"""One dense-field volume of given dimensions through a few extractions, for the counter passes of profiles/walk_counters.sh
and for plain timing: uint8 gradient noise (configs[4]'s field), iso 128, the bench's parameters.
  python3 profiles/walk_row_width.py NX NY NZ [reps] [name=value ...]
Prints one line: dimensions, vertices, passes per vertex, per-stage ms, ps per vertex of the walk."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import __graft_entry__ as g

pkg = g.load_package()
nx, ny, nz = (int(v) for v in sys.argv[1:4])
rest = sys.argv[4:]
reps = int(rest[0]) if rest and "=" not in rest[0] else 4
opts = [a for a in rest if "=" in a]
dev = torch.device("cuda", 0)
step = max(1, (1 << 26) // (nx * ny))
vol = torch.cat([pkg.volumes.gradient_noise(nx, ny, nz, a, min(a + step, nz), xp=torch, device=dev) for a in range(0, nz, step)], 0).contiguous()
torch.cuda.synchronize()
desc = pkg.make_desc(np.uint8, (nx, ny, nz))
prm = pkg.make_params(128, triangles=True, project=True, threshold=0.5, step=0.25, relax=0.95, max_steps=50)
ex = pkg.Extractor(0)
for kv in opts:
    ex.debug_option(kv.split("=")[0], int(kv.split("=")[1]))
ex.debug_option("stage_timing", 1)
acc, n = None, 0
for i in range(reps + 2):
    res = ex.extract_device(vol.data_ptr(), desc, prm)
    if i >= 2:                      # (the first two calls of a context pick its launch shapes)
        d = res.as_dict()
        acc = d if acc is None else {k: acc[k] + d[k] for k in d}
        n += 1
V = int(res.n_points)
ms = {k: acc[k] / n for k in ("ms_classify", "ms_count", "ms_emit_points", "ms_project", "ms_emit_cells", "ms_total")}
print("noise u8 %d x %d x %d  ptr %% 2MiB = %d  V %d  passes/V %.2f  classify %.4f count %.4f points %.4f project %.4f cells %.4f total %.4f ms  walk %.1f ps/V" % (
    nx, ny, nz, vol.data_ptr() % (2 << 20), V, res.proj_iterations / max(V, 1), ms["ms_classify"], ms["ms_count"], ms["ms_emit_points"],
    ms["ms_project"], ms["ms_emit_cells"], ms["ms_total"], ms["ms_project"] * 1e9 / max(V, 1)), flush=True)
ex.close()
