"""-m "not gpu": the hand-run campaign and profile scripts stay importable, and the campaign's generators stay inside what the
oracle accepts (every drawn volume, geometry, start index and held first volume runs through it) -- so that the scripts do not rot
between the rounds in which somebody runs them on a GPU box."""
import glob
import os
import py_compile

import numpy as np

from conftest import ROOT


def test_scripts_compile():
    files = glob.glob(os.path.join(ROOT, "profiles", "*.py")) + glob.glob(os.path.join(ROOT, "tests", "*.py")) + \
        glob.glob(os.path.join(ROOT, "midas-journal-740_amd", "*.py")) + [os.path.join(ROOT, "bench.py"), os.path.join(ROOT, "__graft_entry__.py")]
    assert len(files) > 30
    for f in files:
        py_compile.compile(f, doraise=True)


def test_campaign_generators_run_through_the_oracle(oracle):
    import fuzz_campaign as fz
    rng = np.random.default_rng(11)
    kinds = set()
    for case in range(40):
        nx = int(rng.choice(fz.XS[:10]))
        ny, nz = int(rng.integers(1, 12)), int(rng.integers(1, 12))
        dt = fz.DTYPES[case % len(fz.DTYPES)]
        vox, iso = fz.draw_field(rng, (nz, ny, nx), dt)
        assert vox.shape == (nz, ny, nx) and vox.dtype == np.dtype(dt) and vox.flags["C_CONTIGUOUS"]
        spacing, origin, direction = fz.draw_geometry(rng)
        assert abs(abs(np.linalg.det(direction)) - 1.0) < 1e-9
        start = tuple(int(v) for v in rng.integers(-3000, 3000, size=3))
        kw = dict(triangles=bool(case & 1), project=True, threshold=0.01, step=0.25 * min(spacing), relax=0.9, max_steps=6)
        mesh = oracle.run(vox, iso, spacing=spacing, origin=origin, direction=direction, index_start=start, **kw)
        assert mesh.cells.shape[1] == (3 if case & 1 else 4)
        if case % 5 == 0 and mesh.points.shape[0]:
            fvox, _ = fz.draw_field(rng, (5, 6, 7), dt)
            fs, fo, fd = fz.draw_geometry(rng)
            held = oracle.run(vox, iso, spacing=spacing, origin=origin, direction=direction, index_start=start,
                              first=(fvox, fs, fo, fd, (3, -2, 1)), **kw)
            assert held.points.shape == mesh.points.shape and np.array_equal(held.cells.shape, mesh.cells.shape)
        kinds.add(np.dtype(dt).kind)
    assert kinds == {"u", "i", "f"}
    # the device copy helper keeps the bytes of the unsigned types torch cannot hold as such
    class _T:
        @staticmethod
        def from_numpy(a):
            class _R:
                def __init__(self, arr): self.arr = arr
                def cuda(self): return self.arr
            return _R(a)
    for dt in (np.uint16, np.uint32, np.uint64, np.uint8, np.float32):
        a = (np.arange(24).reshape(2, 3, 4) * 1000).astype(dt)
        b = fz.to_device(_T, a)
        assert b.tobytes() == a.tobytes() and b.dtype.itemsize == a.dtype.itemsize
