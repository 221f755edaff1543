"""CPU tests of the oracle (the checker): pinned against every known-answer vector the reference
holds for the hot path, cross-checked against an independent numpy closed form, plus the [ITK]
contract pieces one by one (H7)."""
import os

import numpy as np
import pytest

from conftest import point_bytes as _point_bytes


def test_reference_ctest_table(oracle, volumes, ctest_cases):
    """All 19 (points, cells) pairs of /root/reference/Testing/CMakeLists.txt:10-331."""
    assert len(ctest_cases) == 19
    for c in ctest_cases:
        vol = volumes(c["input"])
        m = oracle.run(vol.voxels, c["iso"], c["triangles"], c["project"], c["threshold"], c["step"], c["relax"],
                       c["max_steps"])
        assert len(m.points) == c["points"], c["name"]
        assert len(m.cells) == c["cells"], c["name"]


def test_closed_form_counts_agree_on_data(oracle, volumes, ctest_cases):
    for name, iso in sorted({(c["input"], c["iso"]) for c in ctest_cases}):
        vol = volumes(name)
        m = oracle.run(vol.voxels, iso, triangles=False, project=False)
        assert (len(m.points), len(m.cells)) == oracle.closed_form_counts(vol.voxels, iso)


def test_baseline_iso_values(oracle, volumes):
    """Counts at BASELINE.json's iso 128 (derived)."""
    for name, pts, quads in [("nucleon.mha", 3640, 3636), ("fuel.mha", 1218, 1208), ("marschnerlobb.mha", 14726, 15744)]:
        m = oracle.run(volumes(name).voxels, 128)
        assert (len(m.points), len(m.cells)) == (pts, 2 * quads)


def test_closed_form_on_dense_random_volumes(oracle):
    """Dense noise never leaves a slice empty, so the closed form must hold (also on the border: Q2)."""
    rng = np.random.default_rng(1)
    for _ in range(10):
        shape = tuple(int(v) for v in rng.integers(2, 12, size=3))
        vox = rng.integers(0, 255, size=shape, dtype=np.uint8)
        m = oracle.run(vox, 128, triangles=False, project=False)
        if (vox >= 128).any(axis=(1, 2)).all():
            assert (len(m.points), len(m.cells)) == oracle.closed_form_counts(vox, 128)


def test_quirks(oracle):
    vox = np.zeros((10, 8, 8), dtype=np.uint8)
    vox[5, 3, 3] = vox[7, 3, 3] = 255                       # Q1: 12 points instead of 16
    m = oracle.run(vox, 128, triangles=False, project=False)
    assert (len(m.points), len(m.cells)) == (12, 12)
    vox = np.zeros((4, 4, 4), dtype=np.uint8)
    vox[0, 0, 0] = 255                                       # Q2: open border
    m = oracle.run(vox, 128, triangles=False, project=False)
    assert (len(m.points), len(m.cells)) == (7, 3)
    vox[:] = 255
    m = oracle.run(vox, 128)
    assert (len(m.points), len(m.cells)) == (0, 0)


def test_topology_blob_cases(oracle, volumes):
    """blob0 is one voxel: its 8 corners in corner order, 6 quads in face order (txx:197-202)."""
    vol = volumes("blob0.mha")
    m = oracle.run(vol.voxels, 200, triangles=False, project=False)
    z, y, x = (int(v[0]) for v in np.nonzero(vol.voxels >= 200))
    VO = [(0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 1, 0), (0, 0, 1), (1, 0, 1), (1, 1, 1), (0, 1, 1)]
    want = np.array([(x + a - 0.5, y + b - 0.5, z + c - 0.5) for a, b, c in VO], dtype=np.float32)
    assert np.array_equal(m.points, want)
    assert m.cells.tolist() == [[0, 4, 7, 3], [0, 1, 5, 4], [1, 2, 6, 5], [2, 3, 7, 6], [0, 3, 2, 1], [4, 5, 6, 7]]
    # unprojected unit quads always tie on the diagonal test -> first split form (txx:298-302)
    t = oracle.run(vol.voxels, 200, triangles=True, project=False)
    assert t.cells[:2].tolist() == [[0, 4, 3], [4, 7, 3]]


def test_interpolation_contract(oracle):
    rng = np.random.default_rng(0)
    vox = rng.random((5, 6, 7)).astype(np.float32)
    # at pixel centres: the pixel itself
    assert oracle.interpolate(vox, (3.0, 2.0, 1.0)) == float(vox[1, 2, 3])
    # trilinear weights
    p = (2.25, 3.5, 1.75)
    x0, y0, z0 = 2, 3, 1
    dx, dy, dz = 0.25, 0.5, 0.75
    want = 0.0
    for k in range(8):
        ux, uy, uz = k & 1, (k >> 1) & 1, k >> 2
        w = (dx if ux else 1 - dx) * (dy if uy else 1 - dy) * (dz if uz else 1 - dz)
        want += w * float(vox[z0 + uz, y0 + uy, x0 + ux])
    assert oracle.interpolate(vox, p) == pytest.approx(want, rel=1e-15)
    # outside the image: clamped neighbours, no exception
    assert np.isfinite(oracle.interpolate(vox, (-3.0, 100.0, 2.0)))
    # geometry: spacing and origin
    assert oracle.interpolate(vox, (10.0 + 3 * 0.5, -1.0 + 2 * 2.0, 1 * 1.5), spacing=(0.5, 2.0, 1.5),
                              origin=(10.0, -1.0, 0.0)) == float(vox[1, 2, 3])


def test_gradient_contract(oracle):
    vox = np.arange(4 * 5 * 6, dtype=np.float32).reshape(4, 5, 6) ** 1.5
    g = oracle.gradient_at_index(vox, (2, 2, 1))
    want = [(vox[1, 2, 3] - vox[1, 2, 1]) / 2, (vox[1, 3, 2] - vox[1, 1, 2]) / 2, (vox[2, 2, 2] - vox[0, 2, 2]) / 2]
    assert np.allclose(g, want, rtol=1e-6)
    # ZeroFluxNeumann border: one-sided difference halved
    g = oracle.gradient_at_index(vox, (0, 0, 0))
    assert np.allclose(g, [(vox[0, 0, 1] - vox[0, 0, 0]) / 2, (vox[0, 1, 0] - vox[0, 0, 0]) / 2,
                           (vox[1, 0, 0] - vox[0, 0, 0]) / 2], rtol=1e-6)
    # spacing scales the derivative
    g2 = oracle.gradient_at_index(vox, (2, 2, 1), spacing=(2.0, 1.0, 0.5))
    assert np.allclose(g2, [want[0] / 2, want[1], want[2] * 2], rtol=1e-6)


def test_index_to_point(oracle):
    vox = np.zeros((2, 2, 2), dtype=np.uint8)
    p = oracle.index_to_point(vox, (3, 4, 5), spacing=(0.5, 2.0, 1.0), origin=(1.0, -2.0, 0.25))
    assert np.array_equal(p, np.array([2.5, 6.0, 5.25], dtype=np.float32))


def test_projection_moves_vertices_onto_the_surface(oracle):
    n = 24
    z, y, x = np.meshgrid(*(np.arange(n, dtype=np.float64),) * 3, indexing="ij")
    c = (n - 1) / 2
    sdf = (8.0 - np.sqrt((x - c - 0.25) ** 2 + (y - c - 0.125) ** 2 + (z - c) ** 2)).astype(np.float32)
    m = oracle.run(sdf, 0.0, triangles=True, project=True, threshold=0.02, step=0.25, relax=0.95, max_steps=50)
    r = np.sqrt(((m.points - np.array([c + 0.25, c + 0.125, c], dtype=np.float32)) ** 2).sum(1))
    assert np.abs(r - 8.0).max() < 0.08      # thr 0.02 + trilinear bias on an r=8 sphere
    assert m.info["proj_stop_threshold"] == len(m.points)
    # triangles: two per quad, every triangle uses three distinct vertices
    assert (m.cells[:, 0] != m.cells[:, 1]).all() and (m.cells[:, 1] != m.cells[:, 2]).all()


def test_pixel_types_agree(oracle, volumes):
    """The template is instantiated per pixel type; integer types must give identical meshes."""
    vox = volumes("nucleon.mha").voxels
    ref = oracle.run(vox, 128)
    for dt in (np.int16, np.uint16, np.int32, np.uint32, np.float32, np.float64):
        m = oracle.run(vox.astype(dt), 128)
        assert np.array_equal(m.cells, ref.cells)
        assert np.array_equal(m.points.view(np.uint32), ref.points.view(np.uint32))


def test_faithful_cells_mode_is_identical(oracle, volumes):
    vox = volumes("fuel.mha").voxels
    a = oracle.run(vox, 15, faithful_cells=False, gradient_threads=1)
    b = oracle.run(vox, 15, faithful_cells=True, gradient_threads=4)
    assert np.array_equal(a.cells, b.cells) and np.array_equal(a.points.view(np.uint32), b.points.view(np.uint32))


def test_oracle_reproduces_committed_mesh_digests(pkg, oracle, volumes):
    """tests/golden/mesh_digests.json (made by make_mesh_digests.py from this oracle): every bit of every
    Data mesh -- ids, order, float coordinates -- is frozen, so the checker cannot drift unnoticed."""
    import hashlib
    import json
    rows = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "mesh_digests.json")))
    assert len(rows) == 44
    for r in rows:
        vol = volumes(r["input"])
        m = oracle.run(vol.voxels, r["iso"], triangles=r["triangles"], project=r["project"], threshold=r["threshold"],
                       step=r["step"], relax=r["relax"], max_steps=r["max_steps"], spacing=vol.spacing,
                       origin=vol.origin, direction=vol.direction)
        assert (m.points.shape[0], m.cells.shape[0]) == (r["points"], r["cells"]), r["input"]
        assert hashlib.sha256(_point_bytes(m.points)).hexdigest() == r["points_sha256"], r["input"]
        assert hashlib.sha256(m.cells.astype("<u8").tobytes()).hexdigest() == r["cells_sha256"], r["input"]


def test_oracle_reproduces_committed_variant_digests(pkg, oracle, volumes):
    """tests/golden/variant_digests.json: the oracle's restatements of the reference's compiled-out variants (the two
    other projection branches, the recursive-Gaussian gradient) frozen on every Data volume."""
    import hashlib
    import json
    rows = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "variant_digests.json")))
    assert len(rows) == 33
    for r in rows:
        vol = volumes(r["input"])
        m = oracle.run(vol.voxels, r["iso"], triangles=r["triangles"], project=r["project"], threshold=r["threshold"],
                       step=r["step"], relax=r["relax"], max_steps=r["max_steps"], variant=r["variant"],
                       gradient=r["gradient"], spacing=vol.spacing, origin=vol.origin, direction=vol.direction)
        assert (m.points.shape[0], m.cells.shape[0]) == (r["points"], r["cells"]), r["input"]
        assert hashlib.sha256(_point_bytes(m.points)).hexdigest() == r["points_sha256"], (r["input"], r["variant"], r["gradient"])
        assert hashlib.sha256(m.cells.astype("<u8").tobytes()).hexdigest() == r["cells_sha256"], r["input"]


def test_oracle_reproduces_later_update_and_start_index_digests(pkg, oracle, volumes):
    """tests/golden/later_update_digests.json (round 5): a LATER Update() of a filter object -- quirk Q3, the walk along the first
    input's gradient (cuberille_oracle_run_after) -- and a buffered region that starts at a non-zero index, frozen on every Data
    volume: the restatements cannot drift, and the HIP path is held to the same bytes where only the fixtures travel."""
    import hashlib
    import json
    rows = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "later_update_digests.json")))
    assert len(rows) == 22
    for r in rows:
        vol = volumes(r["input"])
        kw = dict(triangles=r["triangles"], project=r["project"], threshold=r["threshold"], step=r["step"], relax=r["relax"],
                  max_steps=r["max_steps"])
        if r["first"]:
            f = volumes(r["first"])
            m = oracle.run(vol.voxels, r["iso"], first=(f.voxels, f.spacing, f.origin, f.direction), **kw)
        else:
            m = oracle.run(vol.voxels, r["iso"], spacing=tuple(r["spacing"]), origin=tuple(r["origin"]), index_start=tuple(r["index_start"]), **kw)
        assert (m.points.shape[0], m.cells.shape[0]) == (r["points"], r["cells"]), r["input"]
        assert hashlib.sha256(_point_bytes(m.points)).hexdigest() == r["points_sha256"], (r["input"], r["first"], r["index_start"])
        assert hashlib.sha256(m.cells.astype("<u8").tobytes()).hexdigest() == r["cells_sha256"], r["input"]


def test_oracle_reproduces_bench_field_digests(pkg, oracle):
    """tests/golden/bench_field_digests.json (made by make_bench_field_digests.py): the oracle on the bench's own
    bit-portable fields is frozen too, so the checker of the GPU parity tests on those fields cannot drift."""
    import hashlib
    import json
    from conftest import GOLDEN, point_bytes
    rows = json.load(open(os.path.join(GOLDEN, "bench_field_digests.json")))
    for r in rows:
        if r["n"] > 64 and r["field"] == "sphere_sdf":
            continue                                   # one size per field keeps the CPU suite short
        if r["field"] == "sphere_sdf":
            vox = pkg.volumes.sphere_sdf(r["n"])
        else:
            vox = pkg.volumes.gradient_noise(r["n"], r["n"], r["n"] * 1000000, 0, r["n"])
        assert hashlib.sha256(np.ascontiguousarray(vox).tobytes()).hexdigest() == r["volume_sha256"]
        m = oracle.run(vox, r["iso"], triangles=r["triangles"], project=r["project"], threshold=r["threshold"], step=r["step"],
                       relax=r["relax"], max_steps=r["max_steps"])
        assert (len(m.points), len(m.cells), m.info["proj_iterations"]) == (r["points"], r["cells"], r["proj_iterations"])
        assert hashlib.sha256(point_bytes(m.points)).hexdigest() == r["points_sha256"]
        assert hashlib.sha256(m.cells.astype("<u8").tobytes()).hexdigest() == r["cells_sha256"]


# ---- the two projection branches the reference compiles out (h:22-23; txx:340-397 and 398-437) ----------------------
# No fixture of the reference covers them (its CTest table only runs the shipped branch), so the C++ restatement is
# checked against a second, independent restatement: the control flow below in plain Python floats (IEEE double, like
# the reference's arithmetic) over the two primitives the contract tests above pin (I5 interpolation, I6 gradient).

from restate import f32 as _f32, py_normal as _py_normal, py_default_walk as _py_default_walk  # noqa: E402


def _py_advanced(oracle, vol, iso, v, thr, step, relax, max_steps):
    v = [float(c) for c in v]
    number_of_steps, swaps, previous, passes = 0, 0, -1, 0
    while True:
        passes += 1
        nrm = _py_normal(oracle, vol, v)
        temp = [[_f32(v[k] + (nrm[k] * s * step)) for k in range(3)] for s in (+1.0, -1.0)]
        step *= relax
        diff = [abs(oracle.interpolate(vol, tuple(t)) - iso) for t in temp]
        i = 0 if diff[0] <= diff[1] else 1
        if previous < 0:
            previous = i
        swaps += int(previous != i)
        v = temp[i]
        if diff[i] < thr:
            break
        number_of_steps += 1
        if number_of_steps - 1 > max_steps:
            break
        if swaps >= 5:
            break
    return v, passes


def _py_linesearch(oracle, vol, iso, v, step, max_steps):
    v = [float(c) for c in v]
    nrm = _py_normal(oracle, vol, v)
    best, best_metric, passes = list(v), 10000.0, 0
    for sign in (-1.0, 1.0):
        for j in range(1, max_steps // 2):
            passes += 1
            d = float(j) / (float(max_steps) / 2.0)
            temp = [_f32(v[k] + (nrm[k] * sign * step * d)) for k in range(3)]
            metric = abs(oracle.interpolate(vol, tuple(temp)) - iso)
            if metric < best_metric:
                best_metric, best = metric, temp
    return best, passes


def _small_field():
    n = 11
    z, y, x = np.meshgrid(*(np.arange(n, dtype=np.float64),) * 3, indexing="ij")
    c = (n - 1) / 2
    return (3.3 - np.sqrt((x - c - 0.25) ** 2 + (y - c - 0.125) ** 2 + (z - c + 0.3) ** 2)
            + 0.15 * np.sin(1.7 * x) * np.cos(1.3 * y + z)).astype(np.float32)


@pytest.mark.parametrize("variant", [1, 2])
def test_compiled_out_projection_branches_against_python_restatement(oracle, variant):
    vol = _small_field()
    kw = dict(threshold=0.01, step=0.3, relax=0.9, max_steps=14)
    flat = oracle.run(vol, 0.0, triangles=False, project=False, **kw)
    got = oracle.run(vol, 0.0, triangles=False, project=True, variant=variant, **kw)
    assert np.array_equal(got.cells, flat.cells)              # which vertices a quad joins never depends on the walk
    passes = 0
    for i, v in enumerate(flat.points):
        if variant == 1:
            want, n = _py_advanced(oracle, vol, 0.0, v, kw["threshold"], kw["step"], kw["relax"], kw["max_steps"])
        else:
            want, n = _py_linesearch(oracle, vol, 0.0, v, kw["step"], kw["max_steps"])
        passes += n
        assert np.array_equal(np.asarray(want, dtype=np.float32).view(np.uint32), got.points[i].view(np.uint32)), i
    assert got.info["proj_iterations"] == passes


def test_compiled_out_projection_branches_properties(oracle):
    n = 32
    z, y, x = np.meshgrid(*(np.arange(n, dtype=np.float64),) * 3, indexing="ij")
    c = (n - 1) / 2
    sdf = (11.0 - np.sqrt((x - c - 0.25) ** 2 + (y - c - 0.125) ** 2 + (z - c) ** 2)).astype(np.float32)
    kw = dict(triangles=True, threshold=0.02, step=0.25, relax=0.95, max_steps=50)
    flat = oracle.run(sdf, 0.0, project=False, **kw)
    centre = np.array([c + 0.25, c + 0.125, c], dtype=np.float32)
    adv = oracle.run(sdf, 0.0, project=True, variant=1, **kw)
    r = np.sqrt(((adv.points - centre) ** 2).sum(1))
    assert np.abs(r - 11.0).max() < 0.3 and np.abs(r - 11.0).mean() < 0.06
    assert adv.info["proj_iterations"] <= len(adv.points) * (kw["max_steps"] + 2)
    ls = oracle.run(sdf, 0.0, project=True, variant=2, **kw)
    moved = np.sqrt(((ls.points.astype(np.float64) - flat.points) ** 2).sum(1))
    assert moved.max() < kw["step"] + 1e-6                    # d = j / (max_steps / 2) stays below 1 (txx:415-418)
    assert ls.info["proj_iterations"] == len(ls.points) * 2 * (kw["max_steps"] // 2 - 1)
    # no sample at all (max_steps < 4): the vertex stays where it is
    none = oracle.run(sdf, 0.0, project=True, variant=2, **dict(kw, max_steps=3))
    assert np.array_equal(none.points.view(np.uint32), flat.points.view(np.uint32))


def test_default_walk_restatement_matches_the_oracle(oracle):
    """tests/restate.py:py_default_walk (txx:439-474 in Python over the pinned primitives) against the oracle's own
    default branch: the restatement the host-walk tests of the drop-in filter lean on is itself held to the oracle."""
    vol = _small_field()
    kw = dict(threshold=0.01, step=0.3, relax=0.9, max_steps=14)
    flat = oracle.run(vol, 0.0, triangles=False, project=False, **kw)
    got = oracle.run(vol, 0.0, triangles=False, project=True, **kw)
    passes = 0
    for i, v in enumerate(flat.points):
        want, n = _py_default_walk(oracle, vol, lambda p: oracle.interpolate(vol, p), 0.0, v, kw["threshold"], kw["step"],
                                   kw["relax"], kw["max_steps"])
        passes += n
        assert np.array_equal(np.asarray(want, dtype=np.float32).view(np.uint32), got.points[i].view(np.uint32)), i
    assert got.info["proj_iterations"] == passes


def _other_field():
    """A smaller, shifted neighbour of _small_field (9 x 10 x 8 voxels): as the FIRST input of a filter object its gradient image
    does not cover the second input's surface everywhere -- the cached interpolator clamps to its own extent (I5)."""
    z, y, x = np.meshgrid(np.arange(8, dtype=np.float64), np.arange(10, dtype=np.float64), np.arange(9, dtype=np.float64), indexing="ij")
    return (3.0 - np.sqrt((x - 4.5) ** 2 + 0.8 * (y - 4.0) ** 2 + (z - 3.75) ** 2) + 0.1 * np.cos(1.1 * x + 0.7 * z)).astype(np.float32)


def test_second_update_walks_along_the_first_inputs_gradient(oracle):
    """Quirk Q3 (txx:484): ComputeGradientImage() builds the gradient interpolator only while it is null, so a LATER Update()
    of the same filter object walks along the gradient image -- through the geometry -- of the first projecting update's
    input.  cuberille_oracle_run_after against the Python restatement: normal from the FIRST volume's gradient image at the
    point's continuous index IN THAT IMAGE (clamped to its extent), value from the current volume."""
    vol, first = _small_field(), _other_field()
    kw = dict(threshold=0.01, step=0.3, relax=0.9, max_steps=14)
    flat = oracle.run(vol, 0.0, triangles=False, project=False, **kw)
    own = oracle.run(vol, 0.0, triangles=False, project=True, **kw)
    # the first update itself, and a "later" update whose first input was the same image: nothing to see
    for same in (None, vol, (vol,), (vol, (1.0, 1.0, 1.0), (0.0, 0.0, 0.0), np.eye(3))):
        again = oracle.run(vol, 0.0, triangles=False, project=True, first=same, **kw)
        assert np.array_equal(again.points.view(np.uint32), own.points.view(np.uint32)) and np.array_equal(again.cells, own.cells)
    for origin in [(0.0, 0.0, 0.0), (1.0, -0.5, 2.0)]:
        got = oracle.run(vol, 0.0, triangles=False, project=True, first=(first, (1.0, 1.0, 1.0), origin), **kw)
        assert np.array_equal(got.cells, flat.cells)
        assert not np.array_equal(got.points, own.points)
        passes = 0
        for i, v in enumerate(flat.points):
            v, step, steps, n = [float(c) for c in v], kw["step"], 0, 0
            while True:
                n += 1
                nrm = _py_normal(oracle, first, [v[k] - origin[k] for k in range(3)])
                value = oracle.interpolate(vol, tuple(v))
                if abs(value - 0.0) < kw["threshold"]:
                    break
                sign = 1.0 if value < 0.0 else -1.0
                with np.errstate(all="ignore"):
                    v = [_f32(v[k] + (nrm[k] * sign * step)) for k in range(3)]
                step *= kw["relax"]
                steps += 1
                if steps - 1 > kw["max_steps"]:
                    break
            passes += n
            assert np.array_equal(np.asarray(v, dtype=np.float32).view(np.uint32), got.points[i].view(np.uint32)), (origin, i)
        assert got.info["proj_iterations"] == passes
    # not projecting: the gradient image is never looked at; another pixel type than the filter's is not a later update of it
    off = oracle.run(vol, 0.0, triangles=True, project=False, first=first, **kw)
    assert_flat = oracle.run(vol, 0.0, triangles=True, project=False, **kw)
    assert np.array_equal(off.points, assert_flat.points) and np.array_equal(off.cells, assert_flat.cells)
    with pytest.raises(ValueError):
        oracle.run(vol, 0.0, first=first.astype(np.float64), **kw)
    # the compiled-out branches and the other gradient go through the same cached interpolator
    for extra in (dict(variant=1), dict(variant=2), dict(gradient=1)):
        a = oracle.run(vol, 0.0, triangles=False, project=True, first=first, **dict(kw, **extra))
        b = oracle.run(vol, 0.0, triangles=False, project=True, **dict(kw, **extra))
        assert np.array_equal(a.cells, b.cells) and not np.array_equal(a.points, b.points), extra


def test_buffered_region_start_index(oracle):
    """A buffered region that does not start at index 0 (a cropped or pasted itk::Image): TransformIndexToPhysicalPoint (txx:266)
    and the interpolators (txx:451,455) work on INDICES, buffer position + start.  With arithmetic that is exact either way
    (spacings that are powers of two, origins on their grid) the mesh equals, bit for bit, the mesh of the same pixels described
    with start 0 and the origin moved to the first buffered pixel; index_to_point is the transform of position + start; with
    awkward geometry the two descriptions still give the same mesh up to the last bits of the doubles involved."""
    vol = _small_field()
    kw = dict(threshold=0.01, step=0.125, relax=0.9, max_steps=14)
    spacing, origin, start = (0.5, 2.0, 1.0), (4.5, -6.0, 3.0), (1000, -37, 512)
    moved = tuple(origin[k] + spacing[k] * start[k] for k in range(3))              # exact
    for tri in (False, True):
        a = oracle.run(vol, 0.0, triangles=tri, spacing=spacing, origin=origin, index_start=start, **kw)
        b = oracle.run(vol, 0.0, triangles=tri, spacing=spacing, origin=moved, **kw)
        assert np.array_equal(a.cells, b.cells) and np.array_equal(a.points.view(np.uint32), b.points.view(np.uint32))
        assert a.info["proj_iterations"] == b.info["proj_iterations"]
    p = oracle.index_to_point(vol, (3, 4, 5), spacing=spacing, origin=origin, index_start=start)
    assert np.array_equal(p, np.array([origin[k] + spacing[k] * ((3, 4, 5)[k] + start[k]) for k in range(3)], dtype=np.float32))
    # interpolation at a point: the same pixel values whichever way the region is described
    q = tuple(moved[k] + spacing[k] * (2.25, 3.5, 4.75)[k] for k in range(3))
    assert oracle.interpolate(vol, q, spacing=spacing, origin=origin, index_start=start) == oracle.interpolate(vol, q, spacing=spacing, origin=moved)
    # ... and outside the region the neighbours clamp to ITS ends, not to index 0
    far = tuple(moved[k] - 5.0 * spacing[k] for k in range(3))
    assert oracle.interpolate(vol, far, spacing=spacing, origin=origin, index_start=start) == float(vol[0, 0, 0])
    # awkward geometry: a rotation, spacings and origins that round -- equal to a relative 1e-6 (mostly to the bit)
    c, s_ = np.cos(0.37), np.sin(0.37)
    d = np.array([[c, -s_, 0.0], [s_, c, 0.0], [0.0, 0.0, 1.0]])
    spacing, origin = (0.7, 1.3, 0.9), (3.3, -2.1, 0.77)
    moved = tuple(np.asarray(origin) + (d @ np.diag(spacing)) @ np.asarray(start, dtype=np.float64))
    a = oracle.run(vol, 0.0, spacing=spacing, origin=origin, direction=d, index_start=start, **dict(kw, step=0.1))
    b = oracle.run(vol, 0.0, spacing=spacing, origin=moved, direction=d, **dict(kw, step=0.1))
    assert np.array_equal(a.cells, b.cells)
    np.testing.assert_allclose(a.points, b.points, rtol=1e-5, atol=1e-4)


def test_recursive_gaussian_gradient_against_second_restatement(oracle):
    """USE_GRADIENT_RECURSIVE_GAUSSIAN (h:21,163-164; txx:488-491), compiled out upstream and with no fixture: the
    oracle's restatement of ITK's recursive Gaussian gradient + the walk through it against a second restatement in
    numpy (tests/restate.py): every projected vertex bit for bit, on unit and on anisotropic spacing.  PARITY UNPINNED
    against ITK itself (oracle/cuberille_oracle.h)."""
    from restate import py_default_walk_normal_fn, py_normal_from_gradient_image, recursive_gaussian_gradient
    vol = _small_field()
    for spacing in [(1.0, 1.0, 1.0), (0.5, 1.0, 2.0)]:
        kw = dict(threshold=0.01, step=0.3 * min(spacing), relax=0.9, max_steps=14, spacing=spacing)
        flat = oracle.run(vol, 0.0, triangles=False, project=False, **kw)
        got = oracle.run(vol, 0.0, triangles=False, project=True, gradient=1, **kw)
        plain = oracle.run(vol, 0.0, triangles=False, project=True, gradient=0, **kw)
        assert not np.array_equal(got.points, plain.points)           # it is another gradient
        grad = recursive_gaussian_gradient(vol, spacing)
        passes = 0
        for i, v in enumerate(flat.points):
            want, n = py_default_walk_normal_fn(lambda p: py_normal_from_gradient_image(grad, p, spacing),
                                                lambda p: oracle.interpolate(vol, p, spacing=spacing), 0.0, v,
                                                kw["threshold"], kw["step"], kw["relax"], kw["max_steps"])
            passes += n
            assert np.array_equal(np.asarray(want, dtype=np.float32).view(np.uint32), got.points[i].view(np.uint32)), (spacing, i)
        assert got.info["proj_iterations"] == passes
    # a derivative filter: on a linear ramp the gradient is the slope, whatever the spacing (NormalizeAcrossScale
    # multiplies by sigma; the walk only uses the direction)
    z, y, x = np.meshgrid(*(np.arange(12, dtype=np.float64),) * 3, indexing="ij")
    ramp = (2.0 * x + 3.0 * y - 1.0 * z).astype(np.float32)
    g = recursive_gaussian_gradient(ramp)
    mid = g[4:8, 4:8, 4:8]
    assert np.allclose(mid[..., 0], 2.0, atol=1e-2) and np.allclose(mid[..., 1], 3.0, atol=1e-2) and np.allclose(mid[..., 2], -1.0, atol=1e-2)
