"""-m gpu: the HIP path (through the C ABI) against the CPU oracle and the reference's
known-answer table, on the reference's own volumes and on synthetic ones: every pixel type, geometry, quirk and launch shape.
Bar: identical vertex ids, cell order and counts; coordinates bit-identical to the oracle (the north star only asks for
1e-5 relative).  (Split by subject in round 5: full-size configs in test_gpu_fullsize.py, slabs and the multi-rank step in
test_gpu_slabs.py, the drop-in filter and the C ABI's behaviours in test_gpu_boundary.py, the threshold sweep in
test_gpu_sweep.py; shared helpers in gpu_helpers.py.)"""
import json
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, assert_same_mesh
from conftest import point_bytes as _point_bytes
from gpu_helpers import _bench_field, _closed_form_counts_torch, _host_threads, _read_vtk_polydata, run_gpu  # noqa: F401

pytestmark = pytest.mark.gpu


def test_reference_ctest_table(pkg, extractor, volumes, ctest_cases):
    """The 19 pinned (points, cells) pairs of Testing/CMakeLists.txt:10-331."""
    for c in ctest_cases:
        vol = volumes(c["input"])
        mesh = run_gpu(pkg, extractor, vol, c["iso"], triangles=c["triangles"], project=c["project"],
                       threshold=c["threshold"], step=c["step"], relax=c["relax"], max_steps=c["max_steps"])
        assert mesh.GetNumberOfPoints() == c["points"], c["name"]
        assert mesh.GetNumberOfCells() == c["cells"], c["name"]


@pytest.mark.parametrize("name,iso,max_steps", [
    ("blob0.mha", 200, 100), ("blob1.mha", 200, 100), ("blob2.mha", 200, 100), ("blob3.mha", 200, 100),
    ("blob4.mha", 200, 100), ("marschnerlobb.mha", 55, 200), ("fuel.mha", 15, 100), ("hydrogenAtom.mha", 15, 100),
    ("neghip.mha", 55, 100), ("nucleon.mha", 140, 100), ("silicium.mha", 85, 100)])
def test_data_volumes_match_oracle(pkg, oracle, extractor, volumes, name, iso, max_steps):
    """Every Data/*.mha at its CTest iso: ids, order and float bits equal the oracle's for
    quads/triangles x projection off/on."""
    vol = volumes(name)
    for tri in (0, 1):
        for proj in (0, 1):
            kw = dict(triangles=tri, project=proj, threshold=0.2, step=0.24, relax=0.95, max_steps=max_steps)
            mesh = run_gpu(pkg, extractor, vol, iso, **kw)
            ref = oracle.run(vol.voxels, iso, **kw)
            assert_same_mesh(mesh, ref)
            # same number of passes through the walk loop (txx:449-470) as the oracle, vertex for vertex in total
            assert int(extractor.result.proj_iterations) == ref.info["proj_iterations"]


@pytest.mark.parametrize("name,points,quads", [("nucleon.mha", 3640, 3636), ("fuel.mha", 1218, 1208),
                                               ("marschnerlobb.mha", 14726, 15744)])
def test_baseline_configs_1_and_2(pkg, oracle, extractor, volumes, name, points, quads):
    """BASELINE.json configs[0..1]: iso 128 ("50 %"), CLI defaults of CuberilleTest01.cxx:98-109."""
    vol = volumes(name)
    mesh = run_gpu(pkg, extractor, vol, 128)
    ref = oracle.run(vol.voxels, 128)
    assert mesh.GetNumberOfPoints() == points and mesh.GetNumberOfCells() == 2 * quads
    assert_same_mesh(mesh, ref)


@pytest.mark.parametrize("dtype", [np.uint8, np.int8, np.uint16, np.int16, np.uint32, np.int32, np.float32, np.float64])
@pytest.mark.parametrize("shape", [(5, 6, 7), (9, 8, 64), (6, 5, 128), (7, 9, 130), (4, 3, 200), (12, 11, 63)])
def test_random_noise_volumes(pkg, oracle, extractor, dtype, shape):
    """White noise: every voxel is surface, the image border is inside (quirk Q2), ragged x
    sizes take both classify paths (nx % 64 == 0 and not)."""
    rng = np.random.default_rng(hash((np.dtype(dtype).name, shape)) % (2 ** 32))
    if np.dtype(dtype).kind == "f":
        vox = rng.normal(0.0, 1.0, size=shape).astype(dtype)
        iso = 0.25
    else:
        info = np.iinfo(dtype)
        vox = rng.integers(max(info.min, -100), min(info.max, 100), size=shape, endpoint=True).astype(dtype)
        iso = 10
    vol = pkg.Volume(vox)
    for tri in (0, 1):
        kw = dict(triangles=tri, project=1, threshold=0.05, step=0.25, relax=0.95, max_steps=50)
        mesh = run_gpu(pkg, extractor, vol, iso, **kw)
        ref = oracle.run(vox, iso, **kw)
        assert_same_mesh(mesh, ref)


def test_sparse_volumes_empty_slice_aliasing(pkg, oracle, extractor):
    """Quirk Q1: an empty slice between occupied slices makes the reference re-use the ids of the
    slice below (txx:139-141 precede 156-161).  Sparse random volumes hit it constantly."""
    rng = np.random.default_rng(7)
    hits = 0
    for trial in range(40):
        shape = (int(rng.integers(4, 14)), int(rng.integers(3, 12)), int(rng.choice([5, 64, 70, 130])))
        vox = (rng.random(shape) < rng.choice([0.002, 0.01, 0.05])).astype(np.uint8) * 255
        # blank whole slices to force gaps
        for z in range(shape[0]):
            if rng.random() < 0.4:
                vox[z] = 0
        vol = pkg.Volume(vox)
        mesh = run_gpu(pkg, extractor, vol, 128, triangles=1, project=0)
        ref = oracle.run(vox, 128, triangles=True, project=False)
        assert_same_mesh(mesh, ref)
        p_closed, _ = oracle.closed_form_counts(vox, 128)
        hits += int(p_closed != len(ref.points))
    assert hits > 0, "no trial exercised the aliasing quirk"


def test_quirk_micro_volumes(pkg, oracle, extractor):
    # Q1: two isolated voxels two slices apart -> 12 points instead of 16
    vox = np.zeros((10, 8, 8), dtype=np.uint8)
    vox[5, 3, 3] = 255
    vox[7, 3, 3] = 255
    mesh = run_gpu(pkg, extractor, pkg.Volume(vox), 128, triangles=0, project=0)
    assert (mesh.GetNumberOfPoints(), mesh.GetNumberOfCells()) == (12, 12)
    assert_same_mesh(mesh, oracle.run(vox, 128, triangles=False, project=False))
    # same volume with the emulation switched off: the geometrically expected 16 points
    mesh = run_gpu(pkg, extractor, pkg.Volume(vox), 128, triangles=0, project=0, q1=False)
    assert (mesh.GetNumberOfPoints(), mesh.GetNumberOfCells()) == (16, 12)
    # Q2: inside voxel in the image corner -> 7 points, 3 quads; all-inside volume -> empty mesh
    vox = np.zeros((4, 4, 4), dtype=np.uint8)
    vox[0, 0, 0] = 255
    mesh = run_gpu(pkg, extractor, pkg.Volume(vox), 128, triangles=0, project=0)
    assert (mesh.GetNumberOfPoints(), mesh.GetNumberOfCells()) == (7, 3)
    assert_same_mesh(mesh, oracle.run(vox, 128, triangles=False, project=False))
    vox[:] = 255
    mesh = run_gpu(pkg, extractor, pkg.Volume(vox), 128)
    assert (mesh.GetNumberOfPoints(), mesh.GetNumberOfCells()) == (0, 0)
    # 1x1x1 and all-outside
    mesh = run_gpu(pkg, extractor, pkg.Volume(np.zeros((1, 1, 1), dtype=np.uint8)), 128)
    assert (mesh.GetNumberOfPoints(), mesh.GetNumberOfCells()) == (0, 0)


def test_zero_gradient_plateau_goes_nan_like_the_oracle(pkg, oracle, extractor):
    """Quirk Q4: Normalize() has no zero guard (txx:452).  A flat plateau gives a zero gradient at
    vertices that are not within the threshold: NaN coordinates on both sides."""
    vox = np.zeros((9, 9, 9), dtype=np.float32)
    vox[2:7, 2:7, 2:7] = 10.0
    kw = dict(triangles=1, project=1, threshold=0.01, step=0.25, relax=0.95, max_steps=20)
    mesh = run_gpu(pkg, extractor, pkg.Volume(vox), 5.0, **kw)
    ref = oracle.run(vox, 5.0, **kw)
    assert np.array_equal(np.isnan(mesh.points), np.isnan(ref.points))
    assert_same_mesh(mesh, ref)


def test_anisotropic_geometry(pkg, oracle, extractor, volumes):
    """Spacing, origin and a rotated direction matrix go through I3/I4/I6 on both sides."""
    src = volumes("nucleon.mha")
    th = 0.3
    direction = np.array([[np.cos(th), -np.sin(th), 0.0], [np.sin(th), np.cos(th), 0.0], [0.0, 0.0, 1.0]])
    vol = pkg.Volume(src.voxels, spacing=(0.5, 1.25, 2.0), origin=(-3.0, 10.5, 0.125), direction=direction)
    kw = dict(triangles=1, project=1, threshold=0.2, step=-1.0, relax=0.95, max_steps=60)
    mesh = run_gpu(pkg, extractor, vol, 128, **kw)
    ref = oracle.run(vol.voxels, 128, spacing=vol.spacing, origin=vol.origin, direction=vol.direction, **kw)
    assert_same_mesh(mesh, ref)


@pytest.mark.parametrize("options", [("no_cmap",), ("no_heads",), ("no_vqueue",), ("no_vqueue", "no_heads"), ("count_no_fold",),
                                     ("proj_xcd=2", "proj_waves=64"), ("proj_xcd=1", "proj_waves=64"), ("proj_xcd=3", "proj_waves=100", "proj_refill=64"), ("proj_short=1", "proj_refill=16"),
                                     ("no_cmap", "no_heads", "no_vqueue"), ("classify_variant",),
                                     ("points_no_split", "proj_chunk=128"), ("cmap_linear",), ("proj_refill=64",), ("proj_refill=3", "proj_chunk=256"),
                                     ("points_split=2", "classify_keep_tail"), ("points_split=4", "proj_chunk64_below=1")])
def test_fallback_paths_without_scratch_tables(pkg, oracle, extractor, volumes, options):
    """When the dense corner map (4 B per lattice corner), the head tables or the vertex-word queue cannot be
    allocated the kernels recompute ids / search the prefix arrays instead; the sweep also runs without its staged
    spans; the launch shapes of large volumes (one lane per vertex word, 128 vertices per wave of the walk, the corner map
    in raster order) on small ones; waves of the walk that refill only when empty, or at three idle lanes; same mesh every way.  The switches are per-context
    options of the C ABI (cuberille_debug_set_option), not environment variables."""
    rng = np.random.default_rng(11)
    vox = rng.integers(0, 255, size=(9, 10, 130), dtype=np.uint8)
    vox[4] = 0                                    # an empty slice: exercises the aliasing redirect too
    try:
        for o in options:
            name, _, value = o.partition("=")
            extractor.debug_option(name, int(value or 1))
        for vol, iso in [(volumes("nucleon.mha"), 128), (volumes("silicium.mha"), 85), (pkg.Volume(vox), 128)]:
            for tri in (0, 1):
                kw = dict(triangles=tri, project=1, threshold=0.2, step=0.24, relax=0.95, max_steps=100)
                mesh = run_gpu(pkg, extractor, vol, iso, **kw)
                assert_same_mesh(mesh, oracle.run(vol.voxels, iso, **kw))
    finally:
        extractor.debug_option("defaults", 0)
    with pytest.raises(pkg._abi.CuberilleError):
        extractor.debug_option("no_such_switch", 1)


@pytest.mark.parametrize("shape", [(128, 1024, 20), (256, 96, 24), (192, 64, 40), (64, 96, 40), (512, 24, 36), (1000, 48, 20),
                                   (100, 64, 24), (40, 50, 30)])
def test_count_forms_agree_with_oracle(pkg, oracle, extractor, shape):
    """The three forms of the count kernel -- untiled, LDS-tiled one block per workgroup, LDS-tiled with a workgroup walking
    up a column of blocks (slices that are whole count blocks: the first shape; the others have blocks that straddle rows
    and slices; rows of ONE word, where every word is a row's first and last; rows of eight words in slices of 192; rows
    that end inside their last word: 16 words for 1000 voxels, two for 100, one for 40) -- on the
    same fields: whole volumes against the oracle, and a slab with a ghost slice against the whole."""
    import torch
    nx, ny, nz = shape
    vox = pkg.volumes.gradient_noise(nx, ny, nz, base_period=32)
    vox[nz // 2] = 0                              # an empty slice: quirk Q1 inside a column of blocks
    vol = pkg.Volume(vox)
    kw = dict(triangles=1, project=1, threshold=0.5, step=0.25, relax=0.95, max_steps=50)
    ref = oracle.run(vox, 128, **kw)
    prm = pkg.make_params(128, **kw)
    dev = torch.from_numpy(vox).cuda()
    a, b = 3, nz - 2
    try:
        # (1: the tile, in columns of 8 where those fill the chip -- not on volumes this small: one block per workgroup, like 2;
        #  4 and 8: columns of that many blocks whatever their number)
        # (3, 32 + form: the dense form of the tile -- one phase, the corner logic per lattice corner -- where rows are a power of
        #  two of whole words: the first two shapes; the third takes the two-phase tile)
        for form in (0, 1, 2, 4, 8, 3, 34, 36, 40):
            extractor.debug_option("count_variant", form)
            assert_same_mesh(run_gpu(pkg, extractor, vol, 128, **kw), ref)
            # a slab: ghost slice below its owned range, halo above
            slab = pkg._abi.Slab(nz, 0, a, b, 0, 0)
            n_p, n_c = extractor.count(dev.data_ptr(), pkg.make_desc(vox.dtype, (nx, ny, nz)), prm, slab)
            extractor.emit(0)
            m = extractor.download()
            if form == 0:
                first = m
            else:
                assert np.array_equal(m.cells, first.cells)
                assert np.array_equal(m.points.view(np.uint32), first.points.view(np.uint32))
    finally:
        extractor.debug_option("defaults", 0)


def test_fuzz_shapes_types_parameters(pkg, oracle, extractor):
    """120 seeded random cases: degenerate and ragged shapes (1-thick volumes, nx around the 64-voxel word
    and the 1 KiB load granule), every pixel type, sparse to dense occupancy, blanked slices (quirk Q1),
    random spacing/origin, quads/triangles, projection on/off with random knobs."""
    rng = np.random.default_rng(20261003)
    dtypes = [np.uint8, np.int8, np.uint16, np.int16, np.uint32, np.int32, np.float32, np.float64]
    xs = [1, 2, 3, 31, 63, 64, 65, 127, 128, 129, 191, 192, 256, 257, 300]
    for case in range(120):
        nx = int(rng.choice(xs))
        ny = int(rng.integers(1, 9))
        nz = int(rng.integers(1, 9))
        dt = dtypes[case % len(dtypes)]
        dens = float(rng.choice([0.02, 0.2, 0.5, 0.9]))
        smooth = rng.random((nz, ny, nx))
        if np.dtype(dt).kind == "f":
            vox = (smooth - (1.0 - dens)).astype(dt)
            iso = 0.0
        else:
            hi = 100
            vox = np.where(smooth < dens, hi, 0).astype(dt) + rng.integers(0, 20, size=smooth.shape).astype(dt)
            iso = 50
        if rng.random() < 0.3 and nz > 2:
            vox[rng.integers(0, nz)] = vox.min()                      # an empty slice
        kw = dict(triangles=bool(rng.integers(0, 2)), project=bool(rng.integers(0, 2)),
                  threshold=float(rng.choice([0.01, 0.2, 5.0])), step=float(rng.choice([-1.0, 0.1, 0.25, 0.6])),
                  relax=float(rng.choice([0.5, 0.95, 1.0])), max_steps=int(rng.choice([0, 3, 50])))
        spacing = tuple(float(v) for v in rng.choice([0.5, 1.0, 1.7], size=3))
        origin = tuple(float(v) for v in rng.normal(0, 5, size=3).round(3))
        vol = pkg.Volume(vox, spacing=spacing, origin=origin)
        mesh = run_gpu(pkg, extractor, vol, iso, **kw)
        ref = oracle.run(vox, iso, spacing=spacing, origin=origin, **kw)
        try:
            assert_same_mesh(mesh, ref)
        except AssertionError as e:
            raise AssertionError("case %d: shape %s dtype %s %s spacing %s: %s" % (
                case, vox.shape, np.dtype(dt).name, kw, spacing, e))


def test_hip_path_reproduces_committed_mesh_digests(pkg, extractor, volumes):
    """The 44 committed digests (11 Data volumes x quads/triangles x projection off/on; oracle output frozen by
    tests/golden/make_mesh_digests.py): the HIP path gives the same bytes without the oracle in the loop."""
    import hashlib
    import json
    rows = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "mesh_digests.json")))
    for r in rows:
        mesh = run_gpu(pkg, extractor, volumes(r["input"]), r["iso"], triangles=r["triangles"], project=r["project"],
                       threshold=r["threshold"], step=r["step"], relax=r["relax"], max_steps=r["max_steps"])
        assert (mesh.GetNumberOfPoints(), mesh.GetNumberOfCells()) == (r["points"], r["cells"]), r["input"]
        assert hashlib.sha256(_point_bytes(mesh.points)).hexdigest() == r["points_sha256"], r
        assert hashlib.sha256(mesh.cells.astype("<u8").tobytes()).hexdigest() == r["cells_sha256"], r


def test_hip_path_reproduces_committed_variant_digests(pkg, extractor, volumes):
    """tests/golden/variant_digests.json (33 rows: 11 Data volumes x {advanced, line-search projection, recursive-Gaussian
    gradient}; frozen oracle output of restated code): the HIP path gives the same bytes without the oracle in the loop."""
    import hashlib
    import json
    rows = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "variant_digests.json")))
    for r in rows:
        mesh = run_gpu(pkg, extractor, volumes(r["input"]), r["iso"], triangles=r["triangles"], project=r["project"],
                       threshold=r["threshold"], step=r["step"], relax=r["relax"], max_steps=r["max_steps"],
                       variant=r["variant"], gradient=r["gradient"])
        assert (mesh.GetNumberOfPoints(), mesh.GetNumberOfCells()) == (r["points"], r["cells"]), r["input"]
        assert hashlib.sha256(_point_bytes(mesh.points)).hexdigest() == r["points_sha256"], r
        assert hashlib.sha256(mesh.cells.astype("<u8").tobytes()).hexdigest() == r["cells_sha256"], r


def test_hip_path_reproduces_later_update_and_start_index_digests(pkg, volumes):
    """tests/golden/later_update_digests.json (22 rows: every Data volume as a LATER update of a filter object whose first input
    was another Data volume -- cuberille_hold_gradient -- and as a region that starts at a non-zero index -- index_start, ABI 13;
    frozen oracle output): the HIP path gives the same bytes without the oracle in the loop."""
    import hashlib
    import json
    rows = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "later_update_digests.json")))
    ex = pkg.Extractor(0)
    try:
        for r in rows:
            vol = volumes(r["input"])
            prm = pkg.make_params(r["iso"], triangles=r["triangles"], project=r["project"], threshold=r["threshold"], step=r["step"],
                                  relax=r["relax"], max_steps=r["max_steps"])
            ex.hold_gradient(False)
            if r["first"]:
                ex.hold_gradient(True)
                ex.extract_host(volumes(r["first"]), prm)
                ex.extract_host(vol, prm)
            else:
                ex.extract_host(pkg.Volume(vol.voxels, spacing=r["spacing"], origin=r["origin"], index_start=r["index_start"]), prm)
            mesh = ex.download()
            assert (mesh.points.shape[0], mesh.cells.shape[0]) == (r["points"], r["cells"]), r["input"]
            assert hashlib.sha256(_point_bytes(mesh.points)).hexdigest() == r["points_sha256"], r
            assert hashlib.sha256(mesh.cells.astype("<u8").tobytes()).hexdigest() == r["cells_sha256"], r
    finally:
        ex.close()


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_nonfinite_and_signed_zero_voxels_walk_like_the_oracle(pkg, oracle, extractor, dtype):
    """The projection shortens the gradient of a cell with finite taps to (-c)f(-1) + c f(+1) (equal to the
    reference's four-term sum up to the sign of a zero) and replays the reference's formula to the letter when
    any cached number is not finite.  Volumes built to sit on both sides of that switch: many exact +0 / -0
    voxels (zero and negative-zero gradient components), infinities and NaNs next to the surface, and
    anisotropic spacing with a rotated direction matrix (the general direction transform)."""
    rng = np.random.default_rng(21)
    th = 0.3
    rot = np.array([[np.cos(th), -np.sin(th), 0.0], [np.sin(th), np.cos(th), 0.0], [0.0, 0.0, 1.0]])
    for trial in range(6):
        shape = (9, 10, 70) if trial % 2 else (12, 9, 33)
        vol = rng.standard_normal(shape)
        vol[rng.random(shape) < 0.35] = 0.0
        vol[rng.random(shape) < 0.15] = -0.0
        if trial >= 2:
            bad = rng.random(shape)
            vol[bad < 0.01] = np.inf
            vol[(bad >= 0.01) & (bad < 0.02)] = -np.inf
            vol[(bad >= 0.02) & (bad < 0.03)] = np.nan
        vol = vol.astype(dtype)
        geo = {} if trial % 3 else dict(spacing=(0.7, 1.3, 2.1), origin=(-3.0, 4.0, 0.5), direction=rot)
        for iso in (0.0, 0.25):
            kw = dict(triangles=1, project=1, threshold=0.01, step=0.25, relax=0.95, max_steps=30)
            want = oracle.run(vol, iso, **kw, **geo)
            got = run_gpu(pkg, extractor, pkg.Volume(vol, **geo), iso, **kw)
            assert_same_mesh(got, want)


def test_bench_fields_match_oracle_and_frozen_digests(pkg, oracle, extractor):
    """The bench's own generators at sizes the oracle finishes in seconds, with the bench's parameters: smooth float32
    fields whose walks take many in-cell iterations and change cells often -- the regime of the headline number.
    sphere_sdf 64^3 / 128^3 (thr 0.05) and uint8 gradient_noise 128^3 (iso 128): ids, order, float bits and the number
    of passes through the walk loop equal the oracle's, and the bytes equal the committed digests
    (tests/golden/bench_field_digests.json, bit-portable generators)."""
    import hashlib
    rows = json.load(open(os.path.join(GOLDEN, "bench_field_digests.json")))
    assert len(rows) == 6
    for r in rows:
        vox = _bench_field(pkg, r["field"], r["n"])
        assert hashlib.sha256(np.ascontiguousarray(vox).tobytes()).hexdigest() == r["volume_sha256"]
        kw = dict(triangles=r["triangles"], project=r["project"], threshold=r["threshold"], step=r["step"], relax=r["relax"],
                  max_steps=r["max_steps"])
        mesh = run_gpu(pkg, extractor, pkg.Volume(vox), r["iso"], **kw)
        ref = oracle.run(vox, r["iso"], **kw)
        assert_same_mesh(mesh, ref)
        assert int(extractor.result.proj_iterations) == ref.info["proj_iterations"] == r["proj_iterations"]
        assert (mesh.GetNumberOfPoints(), mesh.GetNumberOfCells()) == (r["points"], r["cells"])
        assert hashlib.sha256(_point_bytes(mesh.points)).hexdigest() == r["points_sha256"], r
        assert hashlib.sha256(mesh.cells.astype("<u8").tobytes()).hexdigest() == r["cells_sha256"], r


@pytest.mark.parametrize("n", [96, 128])
def test_marschner_lobb_bench_field_matches_oracle(pkg, oracle, extractor, n):
    """BASELINE.json configs[3]'s field (the headline workload) at 96^3 / 128^3 with the bench's parameters -- iso 0.5,
    thr 0.002, step 0.25, relax 0.95, 50 steps: long walks, ~28 % of the vertices on the zero shell -- against the
    oracle: ids, order, float bits, loop passes; whole, period-stacked (the weak-scaling volume) and cut into Z-slabs
    with the halo the library asks for.  (sin/cos are not bit-portable: whoever generates the field hands the same
    bytes to both sides; no frozen digest.)"""
    import torch
    kw = dict(triangles=1, project=1, threshold=0.002, step=0.25, relax=0.95, max_steps=50)
    prm = pkg.make_params(0.5, **kw)
    for vox in (pkg.volumes.marschner_lobb(n), pkg.volumes.marschner_lobb(n, 0, 2 * n, period=n)):
        ref = oracle.run(vox, 0.5, **kw)
        mesh = run_gpu(pkg, extractor, pkg.Volume(vox), 0.5, **kw)
        assert_same_mesh(mesh, ref)
        assert int(extractor.result.proj_iterations) == ref.info["proj_iterations"]
        assert len(ref.points) > 20000
        nz = vox.shape[0]
        halo = max(pkg.required_halo(pkg.make_desc(np.float32, (n, n, nz)), prm))
        assert halo == 8
        dev = torch.from_numpy(vox).cuda()
        # the stacked volume has a run of empty slices between its two copies (the upper part of the field is outside,
        # then two zero shells): quirk Q1 re-uses vertices across it, which slabs reproduce when the cuts leave the run
        # and the occupied slice below it inside one slab
        occupied = np.nonzero((vox >= 0.5).any(axis=(1, 2)))[0]
        first_gap = int(occupied[np.nonzero(np.diff(occupied) > 1)[0][0]]) if (np.diff(occupied) > 1).any() else nz
        c1 = min(nz // 3, first_gap // 2)
        cuts = [0, c1, c1 + 9, nz]
        pts, cells, poff, iters = [], [], 0, 0
        for a, b in zip(cuts[:-1], cuts[1:]):
            lo, hi = max(a - halo, 0), min(b + halo, nz)
            slab = pkg._abi.Slab(nz, lo, a, b, 0, 0)
            n_p, n_c = extractor.count(dev[lo:hi].data_ptr(), pkg.make_desc(np.float32, (n, n, hi - lo)), prm, slab)
            assert not extractor.slab_info()[0]
            res = extractor.emit(poff)
            m = extractor.download()
            pts.append(m.points)
            cells.append(m.cells)
            poff += n_p
            iters += int(res.proj_iterations)
        assert_same_mesh(pkg.Mesh(np.concatenate(pts), np.concatenate(cells)), ref)
        assert iters == ref.info["proj_iterations"]
        if first_gap < nz:
            # a slab whose buffer starts inside the empty run cannot know what lies below: the count says so (the
            # multi-GPU driver resolves it with the other ranks' occupancy) instead of silently skipping the re-use
            a = first_gap + 12
            slab = pkg._abi.Slab(nz, a - halo, a, nz, 0, 0)
            extractor.count(dev[a - halo:].data_ptr(), pkg.make_desc(np.float32, (n, n, nz - a + halo)), prm, slab)
            below, lowest, highest = extractor.slab_info()[:3]
            assert below and lowest > first_gap and highest == int(occupied[-1])


@pytest.mark.parametrize("variant", [1, 2])
def test_compiled_out_projection_branches_match_oracle(pkg, oracle, extractor, volumes, variant):
    """cuberille_params::projection_variant = ADVANCED / LINESEARCH against the oracle's restatement of the same
    branch (itself checked against a second restatement in tests/test_oracle.py): data volumes, every pixel type on
    ragged noise, anisotropic rotated geometry, non-finite voxels, max_steps with no or one sample per side."""
    for name, iso in [("nucleon.mha", 128), ("fuel.mha", 15), ("blob2.mha", 200)]:
        vol = volumes(name)
        for tri in (0, 1):
            kw = dict(triangles=tri, project=1, threshold=0.2, step=0.24, relax=0.95, max_steps=24)
            want = oracle.run(vol.voxels, iso, variant=variant, **kw)
            got = run_gpu(pkg, extractor, vol, iso, variant=variant, **kw)
            assert_same_mesh(got, want)
            assert int(extractor.result.proj_iterations) == want.info["proj_iterations"]
    rng = np.random.default_rng(40 + variant)
    th = 0.3
    rot = np.array([[np.cos(th), -np.sin(th), 0.0], [np.sin(th), np.cos(th), 0.0], [0.0, 0.0, 1.0]])
    for dtype in (np.uint8, np.int8, np.uint16, np.int16, np.uint32, np.int32, np.float32, np.float64):
        shape = (7, 9, 70)
        if np.dtype(dtype).kind == "f":
            vox = rng.normal(0.0, 1.0, size=shape)
            vox[rng.random(shape) < 0.2] = 0.0
            bad = rng.random(shape)
            vox[bad < 0.01] = np.inf
            vox[(bad >= 0.01) & (bad < 0.02)] = np.nan
            vox, iso = vox.astype(dtype), 0.25
        else:
            info = np.iinfo(dtype)
            vox = rng.integers(max(info.min, -100), min(info.max, 100), size=shape, endpoint=True).astype(dtype)
            iso = 10
        for geo in ({}, dict(spacing=(0.7, 1.3, 2.1), origin=(-3.0, 4.0, 0.5), direction=rot)):
            for max_steps in (3, 4, 9, 50):
                kw = dict(triangles=1, project=1, threshold=0.05, step=0.25, relax=0.9, max_steps=max_steps)
                want = oracle.run(vox, iso, variant=variant, **kw, **geo)
                got = run_gpu(pkg, extractor, pkg.Volume(vox, **geo), iso, variant=variant, **kw)
                assert_same_mesh(got, want)
                assert int(extractor.result.proj_iterations) == want.info["proj_iterations"]


def test_unknown_projection_variant_is_refused(pkg, extractor, volumes):
    vol = volumes("blob0.mha")
    with pytest.raises(pkg._abi.CuberilleError) as e:
        extractor.extract_host(vol, pkg.make_params(200, variant=3))
    assert e.value.code == pkg._abi.ERR_ARGUMENT
    # an iso value outside the (integer) pixel type: the reference cannot even express it (h:180-181)
    for iso in (256.0, -1.0, float("nan"), 1e30):
        with pytest.raises(pkg._abi.CuberilleError) as e:
            extractor.extract_host(vol, pkg.make_params(iso))
        assert e.value.code == pkg._abi.ERR_ARGUMENT and "iso value" in str(e.value)
    for iso in (255.9, -0.5, 0.0):                            # cut off like a C cast: 255, 0, 0
        extractor.extract_host(vol, pkg.make_params(iso))
    extractor.extract_host(vol, pkg.make_params(200))        # the context stays usable


def test_without_the_aliasing_quirk_the_mesh_is_the_geometric_one(pkg, oracle, extractor):
    """emulate_empty_slice_aliasing = 0 (the one deliberate departure on offer, include/cuberille_hip.h): on sparse
    volumes with blanked slices -- where the reference re-uses vertices across the gap -- the vertices are exactly the
    lattice corners whose 2x2x2 block is mixed (the closed form evaluated with numpy), each at corner - spacing/2, and
    every quad is a unit square of four distinct vertices around a face between an inside and an outside voxel."""
    rng = np.random.default_rng(99)
    differs = 0
    for trial in range(12):
        shape = (int(rng.integers(5, 12)), int(rng.integers(3, 10)), int(rng.choice([7, 64, 70])))
        vox = (rng.random(shape) < rng.choice([0.01, 0.05, 0.2])).astype(np.uint8) * 255
        for z in range(shape[0]):
            if rng.random() < 0.4:
                vox[z] = 0
        mesh = run_gpu(pkg, extractor, pkg.Volume(vox), 128, triangles=0, project=0, q1=False)
        n_closed, q_closed = oracle.closed_form_counts(vox, 128)
        assert (mesh.GetNumberOfPoints(), mesh.GetNumberOfCells()) == (n_closed, q_closed)
        differs += int(len(oracle.run(vox, 128, triangles=False, project=False).points) != n_closed)
        if n_closed == 0:
            continue
        # the vertex set: mixed corners, at index - 0.5
        ins = vox >= 128
        p = np.pad(ins, 1, mode="edge")
        nz, ny, nx = ins.shape
        blk = np.stack([p[dz:dz + nz + 1, dy:dy + ny + 1, dx:dx + nx + 1] for dz in (0, 1) for dy in (0, 1) for dx in (0, 1)])
        mixed = blk.any(0) & ~blk.all(0)
        cz, cy, cx = np.nonzero(mixed)
        want = set(zip((cx - 0.5).tolist(), (cy - 0.5).tolist(), (cz - 0.5).tolist()))
        got = [tuple(v) for v in mesh.points.astype(np.float64).tolist()]
        assert len(set(got)) == len(got) and set(got) == want
        # every quad: four distinct corners of one unit face, between an inside voxel and an outside one
        q = mesh.points[mesh.cells.astype(np.int64)].astype(np.float64)      # [n, 4, 3]
        centre = q.mean(axis=1)
        assert np.allclose(np.abs(q - centre[:, None, :]).sum(axis=2), 1.0)   # (0.5, 0.5, 0) in some order
        normal_axis = np.argmin(np.ptp(q, axis=1), axis=1)
        assert (np.ptp(q, axis=1)[np.arange(len(q)), normal_axis] == 0).all()
        for c, ax in zip(centre, normal_axis):
            a, b = c.copy(), c.copy()
            a[ax] -= 0.5
            b[ax] += 0.5
            va, vb = (int(round(v)) for v in a), (int(round(v)) for v in b)
            xa, ya, za = va
            xb, yb, zb = vb
            assert ins[za, ya, xa] != ins[zb, yb, xb]
    assert differs > 0, "no trial had the quirk change the vertex count"


def test_extreme_aspect_ratios(pkg, oracle, extractor):
    """Needles, sheets and single rows: one-voxel axes in every position, row lengths around the word and the staging
    granules, thousands of slices of a few voxels."""
    rng = np.random.default_rng(1)
    shapes = [(5000, 1, 1), (1, 3000, 1), (1, 1, 3000), (20000, 2, 2), (300, 1, 70), (1, 70, 300), (2, 3, 4097), (3, 2, 8191),
              (7, 5, 1025), (2, 1, 1), (1, 2, 64), (9, 1, 128), (4000, 3, 65)]
    for shape in shapes:
        for dens in (0.1, 0.5):
            vox = (rng.random(shape) < dens).astype(np.uint8) * 200
            for tri, proj in ((0, 0), (1, 1)):
                kw = dict(triangles=tri, project=proj, threshold=0.2, step=0.25, relax=0.95, max_steps=10)
                mesh = run_gpu(pkg, extractor, pkg.Volume(vox), 100, **kw)
                try:
                    assert_same_mesh(mesh, oracle.run(vox, 100, **kw))
                except AssertionError as e:
                    raise AssertionError("shape %s density %s %s: %s" % (shape, dens, kw, e))


@pytest.mark.parametrize("dtype", [np.int64, np.uint64])
def test_64bit_integer_pixels_match_oracle(pkg, oracle, extractor, dtype):
    """itk::Image<long,3> / <unsigned long,3> (h:150 takes any InputPixelType): compared in the pixel type, through
    (float) into the gradient taps and (double) into the interpolation -- small values (the same mesh as int32), values
    past 2^24 and 2^53 where both conversions round, an iso value a double cannot hold, rows of every kind."""
    rng = np.random.default_rng(21)
    kw = dict(triangles=1, project=1, threshold=0.5, step=0.25, relax=0.95, max_steps=30)
    for shape in [(6, 7, 9), (5, 4, 64), (4, 3, 130)]:
        small = rng.integers(0, 200, size=shape).astype(dtype)
        a = run_gpu(pkg, extractor, pkg.Volume(small), 100, **kw)
        assert_same_mesh(a, oracle.run(small, 100, **kw))
        assert_same_mesh(a, oracle.run(small.astype(np.int32), 100, **kw))
        # a fractional iso value is cast like the reference's InputPixelType member (h:180-181): truncated toward zero,
        # as for the narrower integer types -- 100.5 is 100 (round-3 advisor finding: it used to become 0)
        b = run_gpu(pkg, extractor, pkg.Volume(small), 100.5, **kw)
        assert_same_mesh(b, a)
        assert_same_mesh(b, oracle.run(small, 100.5, **kw))
        assert_same_mesh(b, oracle.run(small.astype(np.int32), 100.5, **kw))
        for bad in (float("nan"), float("inf"), 2.0 ** 64, -2.0 ** 63 - 4096.0, -1.0 if dtype == np.uint64 else 2.0 ** 63):
            with pytest.raises(pkg._abi.CuberilleError):
                run_gpu(pkg, extractor, pkg.Volume(small), bad, **kw)
        # magnitudes where (float)pixel and (double)pixel round: a smooth field scaled to 2^55, low bits noisy
        zz, yy, xx = np.meshgrid(*[np.arange(n, dtype=np.float64) for n in shape], indexing="ij")
        f = np.sin(zz * 0.9) + np.sin(yy * 0.7 + 1.0) + np.sin(xx * 0.3 + 2.0)
        big = ((f + 3.0) * 2.0 ** 55).astype(dtype) + rng.integers(0, 1 << 20, size=shape).astype(dtype)
        iso = (3 << 55) + 12345677                               # not a double
        assert float(iso) != iso
        kwb = dict(kw, threshold=2.0 ** 50)
        m = run_gpu(pkg, extractor, pkg.Volume(big), iso, **kwb)
        ref = oracle.run(big, iso, **kwb)
        assert len(ref.points) > 20
        assert_same_mesh(m, ref)
    if dtype == np.uint64:
        top = (rng.integers(0, 200, size=(5, 6, 70)).astype(np.uint64) << np.uint64(56)) + np.uint64(99)   # above 2^63
        iso = (100 << 56) + 5
        m = run_gpu(pkg, extractor, pkg.Volume(top), iso, **dict(kw, threshold=2.0 ** 58))
        assert_same_mesh(m, oracle.run(top, iso, **dict(kw, threshold=2.0 ** 58)))
    else:
        neg = rng.integers(-(1 << 40), 1 << 40, size=(5, 6, 70)).astype(np.int64)
        m = run_gpu(pkg, extractor, pkg.Volume(neg), -12345, **dict(kw, threshold=2.0 ** 30))
        assert_same_mesh(m, oracle.run(neg, -12345, **dict(kw, threshold=2.0 ** 30)))


def test_termination_counters_match_oracle(pkg, oracle, extractor, volumes):
    """cuberille_result::proj_stop_threshold / proj_stop_steps = the reference's DEBUG_PRINT counters (h:336-338;
    txx:457-459, 470-472) as the oracle counts them, for all three projection branches and in slabs."""
    for name, iso, max_steps in [("nucleon.mha", 140, 100), ("marschnerlobb.mha", 55, 200), ("fuel.mha", 15, 5)]:
        vol = volumes(name)
        for variant in (0, 1, 2):
            kw = dict(triangles=1, project=1, threshold=0.2, step=0.24, relax=0.95, max_steps=max_steps)
            ref = oracle.run(vol.voxels, iso, variant=variant, **kw)
            res = extractor.extract_host(vol, pkg.make_params(iso, variant=variant, **kw))
            got = (int(res.proj_iterations), int(res.proj_stop_threshold), int(res.proj_stop_steps))
            assert got == (ref.info["proj_iterations"], ref.info["proj_stop_threshold"], ref.info["proj_stop_steps"]), (name, variant)
            if variant == 0:
                assert got[1] + got[2] == len(ref.points)
    # without projection nothing is counted; in slabs the owned vertices only
    res = extractor.extract_host(volumes("fuel.mha"), pkg.make_params(15, project=False))
    assert (int(res.proj_stop_threshold), int(res.proj_stop_steps)) == (0, 0)
    import torch
    vol = volumes("marschnerlobb.mha")
    kw = dict(triangles=1, project=1, threshold=0.2, step=0.24, relax=0.95, max_steps=200)
    ref = oracle.run(vol.voxels, 55, **kw)
    nx, ny, nz = vol.dims
    dev = torch.from_numpy(vol.voxels).cuda()
    torch.cuda.synchronize()
    prm = pkg.make_params(55, **kw)
    thr = steps = poff = 0
    for a, b in [(0, 17), (17, 30), (30, nz)]:
        lo, hi = max(a - 8, 0), min(b + 8, nz)
        n_p, _ = extractor.count(dev[lo:hi].data_ptr(), pkg.make_desc(np.uint8, (nx, ny, hi - lo)), prm, pkg._abi.Slab(nz, lo, a, b, 0, 0))
        r = extractor.emit(poff)
        thr += int(r.proj_stop_threshold)
        steps += int(r.proj_stop_steps)
        poff += n_p
    assert (thr, steps) == (ref.info["proj_stop_threshold"], ref.info["proj_stop_steps"])


@pytest.mark.parametrize("variant", [0, 1, 2])
def test_recursive_gaussian_gradient_matches_oracle(pkg, oracle, extractor, volumes, variant):
    """cuberille_params::gradient_variant = RECURSIVE_GAUSSIAN (USE_GRADIENT_RECURSIVE_GAUSSIAN, h:21; txx:488-491: compiled
    out upstream, ITK's Deriche filter restated -- parity unpinned against ITK, the oracle's restatement is held to a second
    one in tests/test_oracle.py): the HIP gradient image + walk against the oracle, bit for bit, with each of the three
    projection branches; data volumes, float fields, anisotropic spacing and a tilted direction matrix."""
    kw = dict(triangles=1, project=1, threshold=0.2, step=0.24, relax=0.95, max_steps=30, variant=variant)
    for name, iso in [("nucleon.mha", 140), ("fuel.mha", 15)]:
        vol = volumes(name)
        ref = oracle.run(vol.voxels, iso, gradient=1, **kw)
        res = extractor.extract_host(vol, pkg.make_params(iso, gradient=1, **kw))
        assert_same_mesh(extractor.download(), ref)
        assert (int(res.proj_iterations), int(res.proj_stop_threshold), int(res.proj_stop_steps)) == \
            (ref.info["proj_iterations"], ref.info["proj_stop_threshold"], ref.info["proj_stop_steps"])
        plain = oracle.run(vol.voxels, iso, gradient=0, **kw)
        assert not np.array_equal(plain.points, ref.points)             # it is another gradient
    from restate import blend_field
    fld, _ = blend_field(18)
    c, s_ = np.cos(0.3), np.sin(0.3)
    tilt = np.array([[c, -s_, 0.0], [s_, c, 0.0], [0.0, 0.0, 1.0]])
    for dtype in (np.float32, np.float64, np.int16):
        v = (fld * 40).astype(dtype)
        for spacing, direction in [((1.0, 1.0, 1.0), np.eye(3)), ((0.5, 1.0, 2.0), np.eye(3)), ((0.7, 0.7, 1.3), tilt)]:
            kw2 = dict(kw, threshold=0.5, step=0.2 * min(spacing))
            geo = dict(spacing=spacing, origin=(3.0, -2.0, 0.5), direction=direction)
            ref = oracle.run(v, 0, gradient=1, **kw2, **geo)
            extractor.extract_host(pkg.Volume(v, spacing, geo["origin"], direction), pkg.make_params(0, gradient=1, **kw2))
            assert_same_mesh(extractor.download(), ref)
    # refused: a slab (the filter needs whole lines), fewer than 4 voxels along an axis, an unknown variant
    import torch
    vol = volumes("nucleon.mha")
    nx, ny, nz = vol.dims
    dev = torch.from_numpy(vol.voxels).cuda()
    torch.cuda.synchronize()
    prm = pkg.make_params(140, gradient=1, **kw)
    with pytest.raises(pkg._abi.CuberilleError) as e:
        extractor.count(dev[:30].data_ptr(), pkg.make_desc(np.uint8, (nx, ny, 30)), prm, pkg._abi.Slab(nz, 0, 0, 20, 0, 0))
    assert e.value.code == pkg._abi.ERR_ARGUMENT
    with pytest.raises(pkg._abi.CuberilleError) as e:
        extractor.extract_host(pkg.Volume(np.zeros((3, 8, 8), dtype=np.uint8)), prm)
    assert e.value.code == pkg._abi.ERR_ARGUMENT
    with pytest.raises(pkg._abi.CuberilleError) as e:
        extractor.extract_host(vol, pkg.make_params(140, gradient=2))
    assert e.value.code == pkg._abi.ERR_ARGUMENT
