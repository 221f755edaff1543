#!/usr/bin/env python3
"""Regenerate tests/golden/bench_field_digests.json: ORACLE output (not reference output -- the reference needs
ITK) on the bench's own synthetic fields at sizes the oracle finishes in seconds, with the bench's parameters:

  sphere_sdf       64^3, 128^3   iso 0.0   thr 0.05      (BASELINE.json configs[2] generator)
  gradient_noise   128^3 uint8   iso 128   thr 0.5       (configs[4] generator)

Both generators are bit-portable (IEEE +,-,*,sqrt / integer arithmetic only), so the digests hold on any host.
The Marschner-Lobb field (configs[3]) goes through sin/cos, which are not bit-portable between libm / numpy SIMD
paths: it is compared with the oracle on the spot in tests/test_gpu_parity.py and has no frozen digest.
Every row: quads and triangles, projection on; counts, projection loop passes, SHA-256 of points and cells.
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import __graft_entry__ as graft  # noqa: E402
from make_mesh_digests import digest  # noqa: E402

CASES = [
    dict(field="sphere_sdf", n=64, iso=0.0, threshold=0.05),
    dict(field="sphere_sdf", n=128, iso=0.0, threshold=0.05),
    dict(field="gradient_noise", n=128, iso=128, threshold=0.5),
]


def volume(pkg, field, n):
    if field == "sphere_sdf":
        return pkg.volumes.sphere_sdf(n)
    if field == "gradient_noise":
        return pkg.volumes.gradient_noise(n, n, n * 1000000, 0, n)      # the bench's call (bench.py generate_block)
    raise ValueError(field)


def main():
    pkg, oracle = graft.load_package(), graft.load_oracle()
    oracle.build()
    rows = []
    for c in CASES:
        vol = volume(pkg, c["field"], c["n"])
        for tri in (0, 1):
            kw = dict(triangles=tri, project=1, threshold=c["threshold"], step=0.25, relax=0.95, max_steps=50)
            m = oracle.run(vol, c["iso"], **kw)
            rows.append(dict(field=c["field"], n=c["n"], iso=c["iso"], **kw, proj_iterations=m.info["proj_iterations"],
                             volume_sha256=hashlib.sha256(np.ascontiguousarray(vol).tobytes()).hexdigest(), **digest(m)))
    with open(os.path.join(HERE, "bench_field_digests.json"), "w") as f:
        json.dump(rows, f, indent=1)
    print(len(rows), "rows")


if __name__ == "__main__":
    main()
