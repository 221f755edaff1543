"""-m gpu: the drop-in boundary -- the reference's unchanged driver built against the drop-in header, other instantiations of
the template, the flat writer, a plain C client, and the behaviours the C ABI promises (warm-up, streams, allocation failures,
two contexts on two threads, stage timing on request)."""
import json
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, assert_same_mesh
from conftest import point_bytes as _point_bytes
from gpu_helpers import _bench_field, _closed_form_counts_torch, _host_threads, _read_vtk_polydata, run_gpu  # noqa: F401

pytestmark = pytest.mark.gpu


def test_reference_driver_unchanged(pkg, extractor, oracle, volumes, ctest_cases, tmp_path):
    """The reference's own CuberilleTest01.cxx, compiled UNCHANGED against the drop-in filter header
    (midas-journal-740_amd/itk; built by __graft_entry__.build() where /root/reference exists), run
    exactly as its CTest table runs it: `CuberilleTest01 Test01 <in> <out> <iso> <pts> <cells> ...`.
    The driver itself asserts the two counts; the .vtk it writes is compared with the oracle."""
    import os
    import subprocess
    from conftest import GOLDEN, ROOT
    exe = os.path.join(ROOT, "midas-journal-740_amd", "itk", "build", "CuberilleTest01")
    if not os.path.exists(exe):
        pytest.skip("drop-in driver binary not built (needs /root/reference at build time)")
    for c in ctest_cases:
        out = str(tmp_path / (c["name"] + ".vtk"))
        args = [exe, "Test01", os.path.join(GOLDEN, "data", c["input"]), out, str(c["iso"]), str(c["points"]),
                str(c["cells"]), str(c["triangles"]), str(c["project"]), repr(c["threshold"]), repr(c["step"]),
                repr(c["relax"]), str(c["max_steps"])]
        r = subprocess.run(args, capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, (c["name"], r.stdout[-400:], r.stderr[-400:])
        assert "Mesh has %d vertices and %d cells" % (c["points"], c["cells"]) in r.stdout
        pts, cells = _read_vtk_polydata(out)
        ref = oracle.run(volumes(c["input"]).voxels, c["iso"], c["triangles"], c["project"], c["threshold"], c["step"],
                         c["relax"], c["max_steps"])
        assert np.array_equal(cells, ref.cells.astype(np.int64)), c["name"]
        np.testing.assert_allclose(pts, ref.points, rtol=1e-6, atol=0)     # 9 significant digits in the file
        # the flat-buffer writer (no itk::Mesh in between) gives the driver's file byte for byte
        run_gpu(pkg, extractor, volumes(c["input"]), c["iso"], triangles=c["triangles"], project=c["project"],
                threshold=c["threshold"], step=c["step"], relax=c["relax"], max_steps=c["max_steps"])
        flat = str(tmp_path / "flat.vtk")
        extractor.write_vtk(flat, threads=3)
        assert open(flat, "rb").read() == open(out, "rb").read(), c["name"]
    # the example main of Source/examples.cxx takes the same arguments without the test name
    exe2 = os.path.join(ROOT, "midas-journal-740_amd", "itk", "build", "Examples")
    c = ctest_cases[-1]
    r = subprocess.run([exe2, os.path.join(GOLDEN, "data", c["input"]), str(tmp_path / "e.vtk"), str(c["iso"]),
                        str(c["points"]), str(c["cells"]), str(c["triangles"]), str(c["project"]), "0.2", "0.24", "0.95",
                        str(c["max_steps"])], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, (r.stdout[-400:], r.stderr[-400:])
    # a wrong expectation must fail like the reference driver does (CuberilleTest01.cxx:193-204)
    r = subprocess.run([exe2, os.path.join(GOLDEN, "data", "blob0.mha"), str(tmp_path / "f.vtk"), "200", "9", "6", "0", "0"],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "Expected mesh with 9 points" in r.stderr


def test_cxx_dropin_instantiates_for_other_pixel_types(oracle, tmp_path):
    """itk/tests/instantiations.cxx: the filter template instantiated for uchar/short/ushort/int/float/double/long/
    unsigned long/long long images (and a float mesh) through the C ABI; each mesh must be a closed genus-0 quad surface.  Also a
    user-defined TInterpolator class: the filter keeps the GPU for the topology and walks the vertices on the host
    through that class (midas-journal-740_amd/itk/itkCuberilleImageToMeshFilter.txx, HostWalk); with a class that
    inherits the linear Evaluate the mesh must equal the all-GPU one bit for bit."""
    import os
    import subprocess
    from conftest import ROOT
    exe = os.path.join(ROOT, "midas-journal-740_amd", "itk", "build", "instantiations")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", os.path.dirname(os.path.dirname(exe)), "build/instantiations"])
    stale_vtk = str(tmp_path / "stale.vtk")
    r = subprocess.run([exe, stale_vtk], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, (r.stdout, r.stderr[-500:])
    lines = r.stdout.strip().splitlines()
    # (ABI 12: SetReproduceStaleGradient -- the reference's quirk Q3 on request: the second Update() of a filter object walks
    #  along the first input's gradient; the program checks the switch against fresh filters, the mesh it wrote is held here
    #  against the oracle's run_after on the same two fields)
    stale = [l for l in lines if l.startswith("stale-gradient")]
    assert len(stale) == 1 and stale[0].endswith(" consistent") and int(stale[0].split()[4]) > 0, stale
    lines = [l for l in lines if not l.startswith("stale-gradient")]
    # (ABI 13: a buffered region that starts at a non-zero index goes to the library as it is)
    reg = [l for l in lines if l.startswith("region-index")]
    assert len(reg) == 1 and reg[0].endswith(" same"), reg
    lines = [l for l in lines if not l.startswith("region-index")]

    def stale_field(nx, ny, nz, cx, radius):
        z, y, x = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")
        r_ = np.sqrt((x - cx) * (x - cx) + (y - 10.25) * (y - 10.25) + (z - 9.5) * (z - 9.5))
        return (radius - r_ + 0.03125 * ((x * 7 + y * 13 + z * 5) % 11)).astype(np.float32)

    first, second = stale_field(26, 22, 20, 12.5, 7.0), stale_field(31, 24, 21, 15.0, 8.5)
    kw = dict(triangles=False, project=True, threshold=0.01, step=0.25, relax=0.95, max_steps=50)
    want = oracle.run(second, 0.0, first=first, **kw)
    pts, cells = _read_vtk_polydata(stale_vtk)
    assert np.array_equal(cells, want.cells.astype(np.int64))
    np.testing.assert_allclose(pts, want.points, rtol=1e-6, atol=1e-9)     # 9 significant digits in the file
    own = oracle.run(second, 0.0, **kw)
    assert np.abs(own.points.astype(np.float64) - pts).max() > 1e-3        # ... and it is not the second field's own gradient
    # the mesh that outlives its filter (cells in one slab the mesh carries in its MetaDataDictionary), 10 pixel-type
    # instantiations (long / unsigned long / long long among them) + the user-defined interpolator type (host walk == GPU
    # walk, quads and triangles)
    # (round 5: + a mesh of DefaultDynamicMeshTraits -- MapContainers -- filled element by element, equal to the static one)
    dyn = [l for l in lines if l.startswith("dynamic-traits")]
    assert len(dyn) == 2 and all(l.endswith(" same") for l in dyn), dyn
    lines = [l for l in lines if not l.startswith("dynamic-traits")]
    assert len(lines) == 13 and all(l.split()[3] == "2" for l in lines)
    assert lines[0].startswith("mesh-outlives-filter")


def test_cxx_flat_writer_route_matches_mesh_route(tmp_path):
    """midas-journal-740_amd/itk/tests/end_to_end.cxx: itk::Mesh fill + itk::VTKPolyDataWriter vs
    WriteLastMeshAsVTKPolyData (flat device buffers -> file) write the same bytes, quads and triangles."""
    exe = os.path.join(ROOT, "midas-journal-740_amd", "itk", "build", "end_to_end")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", os.path.dirname(os.path.dirname(exe)), "build/end_to_end"])
    for tri in ("0", "1"):
        r = subprocess.run([exe, "72", str(tmp_path / "e2e"), tri], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, (r.stdout, r.stderr[-500:])
        info = json.loads(r.stdout.strip().splitlines()[-1])
        assert info["same_bytes"] and info["points"] > 1000 and info["cells"] > 1000


def test_caller_stream_orders_the_extraction(pkg, oracle, extractor, volumes):
    """cuberille_set_stream: with the context on the caller's stream, a volume produced on that stream by
    asynchronous work (here a long chain of torch kernels ending in the real voxels) needs no host
    synchronisation before the extraction; afterwards the context goes back to its own stream."""
    import torch
    vol = volumes("hydrogenAtom.mha")
    want = oracle.run(vol.voxels, 15, triangles=1, project=1, threshold=0.2, step=0.24, relax=0.95, max_steps=100)
    nx, ny, nz = vol.dims
    desc = pkg.make_desc(np.uint8, (nx, ny, nz))
    prm = pkg.make_params(15, triangles=True, project=True, threshold=0.2, step=0.24, relax=0.95, max_steps=100)
    host = torch.from_numpy(vol.voxels).pin_memory()
    side = torch.cuda.Stream()
    try:
        with torch.cuda.stream(side):
            extractor.use_torch_stream()
            dev = torch.zeros((nz, ny, nx), dtype=torch.uint8, device="cuda")
            big = torch.ones((4096, 4096), device="cuda")
            for _ in range(20):                       # keep the stream busy so the copy below lands late
                big = big @ big * 1e-4
            dev.copy_(host, non_blocking=True)
            extractor.extract_device(dev.data_ptr(), desc, prm)      # stream-ordered behind the copy
            mesh = extractor.download()
    finally:
        extractor.use_own_stream()
    assert_same_mesh(mesh, want)
    torch.cuda.synchronize()
    extractor.extract_device(dev.data_ptr(), desc, prm)
    assert_same_mesh(extractor.download(), want)


@pytest.mark.parametrize("variant,suffix", [(1, "advanced"), (2, "linesearch")])
def test_reference_driver_built_with_a_projection_macro(oracle, volumes, ctest_cases, tmp_path, variant, suffix):
    """The reference's CuberilleTest01.cxx compiled unchanged with -DUSE_ADVANCED_PROJECTION=1 /
    -DUSE_LINESEARCH_PROJECTION=1 against the drop-in header: the macro reaches the device as projection_variant."""
    exe = os.path.join(ROOT, "midas-journal-740_amd", "itk", "build", "CuberilleTest01_" + suffix)
    if not os.path.exists(exe):
        pytest.skip("drop-in driver binary not built (needs /root/reference at build time)")
    ran = 0
    for c in ctest_cases:
        if not c["project"] or ran >= 4:
            continue
        ran += 1
        out = str(tmp_path / (c["name"] + ".vtk"))
        args = [exe, "Test01", os.path.join(GOLDEN, "data", c["input"]), out, str(c["iso"]), str(c["points"]),
                str(c["cells"]), str(c["triangles"]), str(c["project"]), repr(c["threshold"]), repr(c["step"]),
                repr(c["relax"]), str(c["max_steps"])]
        r = subprocess.run(args, capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, (c["name"], r.stdout[-400:], r.stderr[-400:])
        pts, cells = _read_vtk_polydata(out)
        ref = oracle.run(volumes(c["input"]).voxels, c["iso"], c["triangles"], c["project"], c["threshold"], c["step"],
                         c["relax"], c["max_steps"], variant=variant)
        shipped = oracle.run(volumes(c["input"]).voxels, c["iso"], c["triangles"], c["project"], c["threshold"],
                             c["step"], c["relax"], c["max_steps"])
        assert np.array_equal(cells, ref.cells.astype(np.int64)), c["name"]
        np.testing.assert_allclose(pts, ref.points, rtol=1e-6, atol=0)
        assert not np.allclose(pts, shipped.points, rtol=1e-6, atol=0)      # it is not the shipped branch
    assert ran > 0


def test_allocation_failure_drill(pkg, oracle, volumes):
    """Every device allocation of an extraction fails once (debug option fail_alloc_at = n: the n-th allocation of this
    thread reports out-of-memory): a required buffer gives CUBERILLE_ERR_HIP with a message and leaves the context
    usable -- the very next call gives the oracle's mesh; an optional table (corner map, head tables, vertex-word
    queue, flat bit stream of ragged rows) is done without and the mesh is still the oracle's."""
    rng = np.random.default_rng(77)
    vox = (rng.random((9, 11, 70)) < 0.3).astype(np.uint8) * 200          # ragged rows: the flat-stream scratch is in play
    vol = pkg.Volume(vox)
    kw = dict(triangles=True, project=True, threshold=0.2, step=0.24, relax=0.95, max_steps=30)
    want = oracle.run(vox, 100, **kw)
    prm = pkg.make_params(100, **kw)
    failed = degraded = 0
    for n in range(40):
        ex = pkg.Extractor(0)                                # a fresh context: nothing is allocated yet
        try:
            ex.debug_option("fail_alloc_at", n)
            try:
                ex.extract_host(vol, prm)
                hit = False
            except pkg._abi.CuberilleError as e:
                assert e.code == pkg._abi.ERR_HIP and "reserve" in str(e), str(e)
                hit = True
            if hit:
                failed += 1
                ex.extract_host(vol, prm)                    # the drill has fired: this one goes through
            assert_same_mesh(ex.download(), want)
            # (the countdown is still armed when the extraction made fewer than n allocations)
            ex.debug_option("fail_alloc_at", 0)
            try:
                ex.extract_host(pkg.Volume(np.zeros((40, 40, 200), dtype=np.uint8)), prm)   # bigger: must allocate
                past_the_end = False
            except pkg._abi.CuberilleError:
                past_the_end = True
            assert past_the_end
            if not hit:
                degraded += 1
        finally:
            ex.debug_option("defaults", 0)
            ex.close()
    assert failed >= 8, failed          # voxels, bits, occupancy, prefix, segment and block tables, points, cells
    assert degraded >= 3, degraded      # optional tables skipped


def test_two_contexts_on_two_threads(pkg, oracle, volumes):
    """"Distinct contexts are independent" (include/cuberille_hip.h): two host threads, one context each (own stream,
    own workspace), extracting different volumes at the same time; every result is the oracle's."""
    import threading
    cases = [("nucleon.mha", 128), ("fuel.mha", 15), ("silicium.mha", 85), ("hydrogenAtom.mha", 15)]
    kw = dict(triangles=True, project=True, threshold=0.2, step=0.24, relax=0.95, max_steps=40)
    want = {name: oracle.run(volumes(name).voxels, iso, **kw) for name, iso in cases}
    errors = []

    def worker(tid):
        try:
            ex = pkg.Extractor(0)
            for rep in range(12):
                name, iso = cases[(tid + rep) % len(cases)]
                ex.extract_host(volumes(name), pkg.make_params(iso, **kw))
                assert_same_mesh(ex.download(), want[name])
            ex.close()
        except Exception as e:          # noqa: BLE001 -- reported by the main thread
            errors.append((tid, repr(e)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


def test_streamed_ingestion_equals_resident_volume(pkg, oracle, extractor, volumes, tmp_path):
    """cuberille_extract_stream (decode overlapped with the upload): a compressed
    MetaImage inflated stretch by stretch into the library's pinned staging memory gives the mesh of the same volume
    handed over whole -- on every shipped volume, on a five-chunk float volume, on slices larger than a chunk; a
    source that gives up ends the call with CUBERILLE_ERR_SOURCE and the context goes on working."""
    import glob
    import torch
    kw = dict(triangles=True, project=True, threshold=0.2, step=0.24, relax=0.95, max_steps=40)
    for path in sorted(glob.glob(os.path.join(GOLDEN, "data", "*.mha"))):
        vol = pkg.read_mha(path)
        iso = 128 if "blob" not in path else 200
        prm = pkg.make_params(iso, **kw)
        res, info = extractor.extract_mha(path, prm)
        got = extractor.download()
        assert info.dims == vol.dims
        assert_same_mesh(got, oracle.run(vol.voxels, iso, **kw))
    # several 32 MiB chunks, compressed float payload with geometry
    n = (600, 256, 256)
    z, y, x = np.meshgrid(*(np.arange(v, dtype=np.float32) for v in n), indexing="ij")
    vox = (np.sin(x * 0.11) + np.cos(y * 0.07) * np.sin(z * 0.05) + 0.1 * np.sin(0.9 * x + 0.7 * y + z)).astype(np.float32)
    del x, y, z
    vol = pkg.Volume(vox, spacing=(0.5, 1.0, 1.5), origin=(3.0, -1.0, 2.0))
    path = str(tmp_path / "waves.mha")
    pkg.write_mha(path, vol, compress=True)
    prm = pkg.make_params(0.25, triangles=True, project=True, threshold=0.002, step=-1.0, relax=0.95, max_steps=50)
    extractor.extract_mha(path, prm)
    streamed = extractor.download()
    dev = torch.from_numpy(vox).cuda()
    extractor.extract_device(dev.data_ptr(), pkg.make_desc(np.float32, vol.dims, vol.spacing, vol.origin), prm)
    whole = extractor.download()
    assert streamed.points.shape[0] > 500000
    assert np.array_equal(streamed.cells, whole.cells) and _point_bytes(streamed.points) == _point_bytes(whole.points)
    # a slice larger than the 32 MiB chunk: one slice per chunk
    big = np.zeros((3, 2100, 4096), dtype=np.float32)
    big[1, 500:1500, 1000:3000] = 1.0
    calls = []

    def source(dst, z0, z1):
        calls.append((z0, z1))
        dst[...] = big[z0:z1]

    prm = pkg.make_params(0.5, triangles=False, project=False)
    extractor.extract_stream(pkg.make_desc(np.float32, (4096, 2100, 3)), source, prm)
    assert calls == [(0, 1), (1, 2), (2, 3)]
    assert extractor.download().cells.shape[0] == 2 * 1000 * 2000 + 2 * 1000 + 2 * 2000
    # the producer gives up half way
    def failing(dst, z0, z1):
        if z0 > 0:
            raise OSError("disk on fire")
        dst[...] = big[z0:z1]

    with pytest.raises(pkg._abi.CuberilleError) as e:
        extractor.extract_stream(pkg.make_desc(np.float32, (4096, 2100, 3)), failing, prm)
    assert e.value.code == pkg._abi.ERR_SOURCE and isinstance(e.value.__cause__, OSError)
    vol = volumes("nucleon.mha")
    extractor.extract_host(vol, pkg.make_params(128, **kw))
    assert_same_mesh(extractor.download(), oracle.run(vol.voxels, 128, **kw))


def test_plain_c_program_through_the_stream_entry(pkg, oracle, extractor, volumes, tmp_path):
    """examples/extract_raw.c (C99, built by __graft_entry__.build()): a raw volume read with fread() into
    cuberille_extract_stream, the reference driver's default parameters, the mesh written by cuberille_mesh_write_vtk --
    the file equals the one the Python host side writes for the same call, and the mesh is the oracle's."""
    exe = os.path.join(ROOT, "examples", "build", "extract_raw")
    if not os.path.exists(exe):
        pytest.skip("examples/build/extract_raw not built")
    for name, iso, mode in [("nucleon.mha", 128, "tri"), ("fuel.mha", 15, "quads")]:
        vol = volumes(name)
        raw = str(tmp_path / "v.raw")
        vol.voxels.tofile(raw)
        out = str(tmp_path / "c.vtk")
        nx, ny, nz = vol.dims
        args = [exe, raw, str(nx), str(ny), str(nz), "u8", str(iso), out] + (["quads"] if mode == "quads" else [])
        r = subprocess.run(args, capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, (r.stdout, r.stderr)
        kw = dict(triangles=mode == "tri", project=True, threshold=0.5, step=0.25, relax=0.95, max_steps=50)
        want = oracle.run(vol.voxels, iso, **kw)
        assert "Mesh has %d vertices and %d cells" % (len(want.points), len(want.cells)) in r.stdout
        run_gpu(pkg, extractor, vol, iso, **kw)
        py = str(tmp_path / "py.vtk")
        extractor.write_vtk(py, threads=2)
        assert open(out, "rb").read() == open(py, "rb").read()
        pts, cells = _read_vtk_polydata(out)
        assert np.array_equal(cells, want.cells.astype(np.int64))
    # a file that ends early: the source gives up, the program reports the library's error
    open(str(tmp_path / "short.raw"), "wb").write(b"\0" * 1000)
    r = subprocess.run([exe, str(tmp_path / "short.raw"), "41", "41", "41", "u8", "128", str(tmp_path / "x.vtk")],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 3 and "chunk source gave up" in r.stderr


def test_stage_timing_is_on_request(pkg, volumes):
    """cuberille_result: ms_total is measured by every extraction, ms_pass by every extraction of more than 4 Mi voxels (a
    smaller one gets ONE event pair: every event between two kernels costs the stream about as much as its kernels do);
    the five per-stage figures only with the context's stage_timing switch, 0 otherwise."""
    ex = pkg.Extractor(0)
    stages = ("ms_classify", "ms_count", "ms_emit_points", "ms_project", "ms_emit_cells")
    prm = pkg.make_params(15, triangles=True, project=True)
    big = np.zeros((160, 192, 192), dtype=np.uint8)                  # 5.9 M voxels
    big[40:120, 50:140, 60:130] = 200
    for vol, light in [(volumes("hydrogenAtom.mha"), True), (pkg.Volume(big), False)]:
        for _ in range(2):                                          # (sized by a host read, then blind: both report alike)
            r = ex.extract_host(vol, prm)
            assert r.ms_total > 0 and (r.ms_pass == 0.0 if light else 0 < r.ms_pass < r.ms_total)
            assert all(getattr(r, k) == 0.0 for k in stages)
        ex.debug_option("stage_timing", 1)
        r = ex.extract_host(vol, prm)
        assert all(getattr(r, k) > 0.0 for k in stages)
        assert abs(r.ms_classify + r.ms_count - r.ms_pass) < 0.02 * r.ms_pass + 0.005
        assert abs(r.ms_pass + r.ms_emit_points + r.ms_project + r.ms_emit_cells - r.ms_total) < 0.02 * r.ms_total + 0.01
        ex.debug_option("defaults", 0)
        r = ex.extract_host(vol, prm)
        assert all(getattr(r, k) == 0.0 for k in stages) and r.ms_total > 0
    ex.close()


@pytest.mark.parametrize("triangles,threads", [(0, 1), (1, 1), (1, 4)])
def test_filter_with_a_nonlinear_interpolator(oracle, tmp_path, triangles, threads):
    """The whole drop-in filter with a TInterpolator that is not the linear one (h:110; B-spline in the reference's
    driver, Testing/CuberilleTest01.cxx:73-75): topology and start points from the GPU, the walk on the host through the
    user's Evaluate() -- here a blend with a second, smoothed image.  Points equal a Python restatement of txx:439-474
    over the oracle's pinned primitives, quads equal the oracle's, triangles follow txx:286-321 on those points; four
    host threads (opt-in) give the same mesh."""
    from restate import blend_field, blend_value, py_default_walk, split_quads
    exe = os.path.join(ROOT, "midas-journal-740_amd", "itk", "build", "host_walk")
    vol, smooth = blend_field()
    n = vol.shape[0]
    kw = dict(threshold=0.02, step=0.25, relax=0.95, max_steps=30)
    flat = oracle.run(vol, 0.0, triangles=False, project=False, **kw)
    vol.tofile(str(tmp_path / "v.raw"))
    smooth.tofile(str(tmp_path / "s.raw"))
    r = subprocess.run([exe, "filter", str(tmp_path / "v.raw"), str(tmp_path / "s.raw"), str(n), "0.0", "0.02", "0.25", "0.95", "30",
                        str(tmp_path / "p.raw"), str(tmp_path / "c.raw"), str(triangles), str(threads)],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout, r.stderr)
    pts = np.fromfile(str(tmp_path / "p.raw"), dtype=np.float32).reshape(-1, 3)
    cells = np.fromfile(str(tmp_path / "c.raw"), dtype=np.uint64).reshape(-1, 3 if triangles else 4)
    assert pts.shape == flat.points.shape
    value = blend_value(oracle, vol, smooth)
    want = np.array([py_default_walk(oracle, vol, value, 0.0, v, kw["threshold"], kw["step"], kw["relax"], kw["max_steps"])[0]
                     for v in flat.points], dtype=np.float32)
    assert np.array_equal(want.view(np.uint32), pts.view(np.uint32))
    if triangles:
        assert np.array_equal(cells, split_quads(want, flat.cells.astype(np.int64)).astype(np.uint64))
    else:
        assert np.array_equal(cells, flat.cells)


def test_throwing_interpolator_leaves_through_update(tmp_path):
    """An exception thrown by the user's Evaluate() -- on the calling thread or inside one of the opt-in worker threads --
    comes out of Update() as an exception (round-2 advisor finding: a worker's exception used to end in std::terminate)."""
    from restate import blend_field
    exe = os.path.join(ROOT, "midas-journal-740_amd", "itk", "build", "host_walk")
    vol, smooth = blend_field(20)
    vol.tofile(str(tmp_path / "v.raw"))
    smooth.tofile(str(tmp_path / "s.raw"))
    for threads in (1, 4):
        r = subprocess.run([exe, "throw", str(tmp_path / "v.raw"), str(tmp_path / "s.raw"), "20", "0.0", str(threads)],
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and "caught: interpolator gave up" in r.stdout, (r.returncode, r.stdout, r.stderr)


def test_warm_up_and_host_mesh(pkg, oracle, volumes):
    """cuberille_warm_up (what the drop-in filter calls from its constructor and from SetInput, so that the one cold
    Update() the reference's driver times -- test:158-160 -- does not pay for the context) leaves no trace in the results:
    the first extraction after it equals the oracle; cuberille_mesh_host hands out the context's own host copy of the
    mesh, the same bytes as cuberille_mesh_download, the same pointers when asked twice, refreshed by the next extraction."""
    ex = pkg.Extractor(0)
    try:
        vol = volumes("nucleon.mha")
        desc = pkg.make_desc(np.uint8, vol.dims)
        ex.warm_up()                       # code objects only
        ex.warm_up(desc)                   # + the workspace for this image
        kw = dict(triangles=1, project=1, threshold=0.2, step=0.24, relax=0.95, max_steps=100)
        for name, iso in (("nucleon.mha", 140), ("fuel.mha", 15), ("nucleon.mha", 140)):
            v = volumes(name)
            ex.extract_host(v, pkg.make_params(iso, **kw))
            ref = oracle.run(v.voxels, iso, **kw)
            view = ex.mesh_host()
            assert_same_mesh(view, ref)
            again = ex.mesh_host()
            assert again.points.ctypes.data == view.points.ctypes.data and again.cells.ctypes.data == view.cells.ctypes.data
            assert_same_mesh(ex.download(), ref)
        ex.warm_up(pkg.make_desc(np.float32, (64, 64, 64)))        # between extractions: only reserves
        assert_same_mesh(ex.mesh_host(), ref)
        # cuberille_release_host_mesh (ABI 11): the host copy goes back to the system, the mesh on the device stays -- the
        # next cuberille_mesh_host maps fresh memory and copies again
        ex.release_host_mesh()
        assert_same_mesh(ex.mesh_host(), ref)
        ex.release_host_mesh()
        ex.release_host_mesh()
        assert_same_mesh(ex.download(), ref)
    finally:
        ex.close()
    # the Python mirror of the filter warms up the same way (constructor, SetInput) and gives the oracle's mesh
    f = pkg.CuberilleImageToMeshFilter(device=0)
    f.SetInput(volumes("fuel.mha"))
    f.SetIsoSurfaceValue(128)
    f.Update()
    assert_same_mesh(f.GetOutput(), oracle.run(volumes("fuel.mha").voxels, 128))


def test_warm_up_leaves_a_live_count_and_mesh_alone(pkg, oracle, volumes):
    """Advisor finding (round 4): cuberille_warm_up(img) for a LARGER image on a context that holds a count or a mesh must not
    move the workspace under it (DevBuf::reserve frees, then allocates): everything that reads the count's tables and the bit
    volume afterwards -- the bits, the slice's bit plane on the device, the plane of ids, the emit behind a count -- still
    gives what it gave before the call.  (The drop-in filter calls warm_up from every SetInput, also after an Update().)"""
    import torch
    ex = pkg.Extractor(0)
    try:
        vol = volumes("nucleon.mha")
        nx, ny, nz = vol.dims
        desc = pkg.make_desc(np.uint8, vol.dims)
        prm = pkg.make_params(140, triangles=1, project=1, threshold=0.2, step=0.24, relax=0.95, max_steps=100)
        ref = oracle.run(vol.voxels, 140, triangles=1, project=1, threshold=0.2, step=0.24, relax=0.95, max_steps=100)
        big = pkg.make_desc(np.float32, (320, 320, 320))          # every workspace buffer would have to grow
        dev = torch.from_numpy(vol.voxels).cuda()
        torch.cuda.synchronize()
        # (1) between a count and its emit
        n_p, n_c = ex.count(dev.data_ptr(), desc, prm)
        assert (n_p, n_c) == (len(ref.points), len(ref.cells))
        bits_before = ex.debug_bits(vol.dims).copy()
        ex.warm_up(big)
        assert np.array_equal(ex.debug_bits(vol.dims), bits_before)
        # (the plane of ids is defined for a slice with nothing occupied above it: the top corners of its inside voxels are
        #  all vertices there)
        zmid = int(np.nonzero((vol.voxels >= 140).any(axis=(1, 2)))[0].max())
        ptr, n = ex.slice_bits_device(zmid)
        W = (nx + 63) // 64
        plane = torch.empty(n, dtype=torch.int64, device="cuda")
        from midas_journal_740_amd.distributed import _words_view
        plane.copy_(_words_view(ptr, n, plane.device))
        torch.cuda.synchronize()
        assert np.array_equal(plane.cpu().numpy().view(np.uint64).reshape(ny, W), bits_before.reshape(nz, ny, W)[zmid])
        ex.emit(0)
        assert_same_mesh(ex.download(), ref)
        # (2) behind a finished mesh
        ex.warm_up(big)
        assert np.array_equal(ex.debug_bits(vol.dims), bits_before)
        ids = torch.empty((nx + 1) * (ny + 1), dtype=torch.int64, device="cuda")
        pts = torch.zeros(((nx + 1) * (ny + 1), 3), dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        ex.alias_plane_device(zmid, ids.data_ptr(), pts.data_ptr())
        torch.cuda.synchronize()
        got = ids.cpu().numpy()
        live = got >= 0
        assert live.any() and got[live].max() < n_p
        # the plane's positions are the mesh's points under those ids
        assert np.array_equal(pts.cpu().numpy()[live].view(np.uint32), ref.points[got[live]].view(np.uint32))
        assert_same_mesh(ex.mesh_host(), ref)
        # the next extraction (of the larger image's size class) grows the workspace itself
        v2 = volumes("hydrogenAtom.mha")
        ex.extract_host(v2, pkg.make_params(15, triangles=1, project=1, threshold=0.2, step=0.24, relax=0.95, max_steps=100))
        assert_same_mesh(ex.download(), oracle.run(v2.voxels, 15, triangles=1, project=1, threshold=0.2, step=0.24, relax=0.95,
                                                   max_steps=100))
    finally:
        ex.close()


def _geom(vol):
    return (vol.voxels, vol.spacing, vol.origin, vol.direction)


def test_hold_gradient_reproduces_the_reference_second_update(pkg, oracle, volumes):
    """Quirk Q3 on request (cuberille_hold_gradient, ABI 12; txx:484): a context asked to hold behaves like the reference's
    filter object -- the gradient interpolator of its FIRST projecting extraction serves every later one, whatever their
    size, geometry or projection branch.  Every mesh against the oracle's cuberille_oracle_run_after, bit for bit; without
    the call (the default) every extraction follows its own volume's gradient."""
    kw = dict(triangles=1, project=1, threshold=0.2, step=0.24, relax=0.95, max_steps=60)
    ex = pkg.Extractor(0)
    try:
        assert ex.gradient_held is None
        ex.hold_gradient(True)
        assert ex.gradient_held is None                    # nothing yet: the next projecting extraction is the first
        # an extraction that does not project builds no gradient image (txx:484: m_ProjectVerticesToIsoSurface && IsNull)
        flat = dict(kw, project=0)
        ex.extract_host(volumes("blob0.mha"), pkg.make_params(128, **flat))
        assert_same_mesh(ex.download(), oracle.run(volumes("blob0.mha").voxels, 128, **flat))
        assert ex.gradient_held is None
        first = volumes("fuel.mha")
        ex.extract_host(first, pkg.make_params(15, **kw))
        assert_same_mesh(ex.download(), oracle.run(first.voxels, 15, **kw))
        assert ex.gradient_held == first.voxels.shape
        moved = 0
        for name, iso, extra in (("nucleon.mha", 140, {}), ("hydrogenAtom.mha", 15, {}), ("silicium.mha", 128, {}),
                                 ("neghip.mha", 64, dict(variant=1)), ("blob2.mha", 128, dict(variant=2)),
                                 ("fuel.mha", 15, dict(triangles=0)), ("fuel.mha", 100, {})):
            v = volumes(name)
            k = dict(kw, **extra)
            ex.extract_host(v, pkg.make_params(iso, **k))
            got = ex.download()
            assert_same_mesh(got, oracle.run(v.voxels, iso, first=_geom(first), **k))
            own = oracle.run(v.voxels, iso, **k)
            moved += int(not np.array_equal(got.points.view(np.uint32), own.points.view(np.uint32)))
            assert ex.gradient_held == first.voxels.shape
            if name == "fuel.mha" and iso == 15:           # the first image again: its own gradient, the same mesh as ever
                assert np.array_equal(got.points.view(np.uint32), own.points.view(np.uint32))
        assert moved >= 5
        # a volume resident on the device takes the same path
        import torch
        v = volumes("hydrogenAtom.mha")
        dev = torch.from_numpy(v.voxels).cuda()
        ex.extract_device(dev.data_ptr(), pkg.make_desc(np.uint8, v.dims), pkg.make_params(15, **kw))
        assert_same_mesh(ex.download(), oracle.run(v.voxels, 15, first=_geom(first), **kw))
        # refused: slabs (the held image is a whole volume's), the other gradient filter
        with pytest.raises(pkg._abi.CuberilleError) as e:
            ex.extract_device(dev.data_ptr(), pkg.make_desc(np.uint8, (128, 128, 64)), pkg.make_params(15, **kw),
                              pkg._abi.Slab(128, 0, 0, 60, 0, 0))
        assert e.value.code == pkg._abi.ERR_ARGUMENT and "cuberille_hold_gradient" in str(e.value)
        with pytest.raises(pkg._abi.CuberilleError) as e:
            ex.extract_host(v, pkg.make_params(15, gradient=1, **kw))
        assert e.value.code == pkg._abi.ERR_ARGUMENT and "cuberille_hold_gradient" in str(e.value)
        # ... but a slab that does not project never looks at a gradient
        ex.count(dev.data_ptr(), pkg.make_desc(np.uint8, (128, 128, 64)), pkg.make_params(15, **flat), pkg._abi.Slab(128, 0, 0, 60, 0, 0))
        # dropped: the default again, every volume its own gradient; asked again: the NEXT projecting extraction is the first
        ex.hold_gradient(False)
        assert ex.gradient_held is None
        v = volumes("nucleon.mha")
        ex.extract_host(v, pkg.make_params(140, **kw))
        assert_same_mesh(ex.download(), oracle.run(v.voxels, 140, **kw))
        assert ex.gradient_held is None
        ex.hold_gradient(True)
        ex.extract_host(v, pkg.make_params(140, **kw))
        assert_same_mesh(ex.download(), oracle.run(v.voxels, 140, **kw))
        assert ex.gradient_held == v.voxels.shape
        w = volumes("fuel.mha")
        ex.extract_host(w, pkg.make_params(15, **kw))
        assert_same_mesh(ex.download(), oracle.run(w.voxels, 15, first=_geom(v), **kw))
    finally:
        ex.close()


def test_hold_gradient_across_geometries_and_sizes(pkg, oracle):
    """The cached interpolator maps a point through the FIRST image's geometry (origin, spacing, direction) and clamps to ITS
    extent: float volumes of different sizes with anisotropic, rotated and shifted geometry on either side; a first volume
    with ragged rows; a second one of a few million voxels (the production launch shapes of every kernel but the walk, which
    takes the plain one-lane-per-vertex form behind a held gradient)."""
    import torch
    c, s = np.cos(0.3), np.sin(0.3)
    rot = np.array([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]])

    def field(nz, ny, nx, cx, r):
        z, y, x = np.meshgrid(np.arange(nz, dtype=np.float64), np.arange(ny, dtype=np.float64), np.arange(nx, dtype=np.float64), indexing="ij")
        return (r - np.sqrt((x - cx) ** 2 + (y - ny / 2.1) ** 2 + (z - nz / 1.9) ** 2) + 0.2 * np.sin(0.9 * x) * np.cos(0.7 * y + 0.5 * z)).astype(np.float32)

    first = pkg.Volume(field(21, 30, 37, 17.0, 9.0), spacing=(0.8, 1.0, 1.3), origin=(-2.0, 1.5, 0.25), direction=rot)
    later = [pkg.Volume(field(40, 33, 29, 14.0, 10.0), spacing=(1.0, 1.0, 1.0)),
             pkg.Volume(field(18, 18, 70, 30.0, 7.0), spacing=(0.5, 1.1, 0.9), origin=(3.0, -1.0, 2.0), direction=rot.T),
             pkg.Volume(field(160, 160, 160, 80.0, 61.0), spacing=(0.25, 0.25, 0.25), origin=(-8.0, -6.0, -4.0))]
    ex = pkg.Extractor(0)
    try:
        ex.hold_gradient(True)
        for i, v in enumerate([first] + later):
            kw = dict(triangles=1, project=1, threshold=0.01, step=0.25 * min(v.spacing), relax=0.9, max_steps=25)
            ex.extract_host(v, pkg.make_params(0.0, **kw))
            ref = oracle.run(v.voxels, 0.0, spacing=v.spacing, origin=v.origin, direction=v.direction, gradient_threads=_host_threads(),
                             first=None if i == 0 else _geom(first), **kw)
            assert_same_mesh(ex.download(), ref)
            assert ex.gradient_held == first.voxels.shape
            if i:
                own = oracle.run(v.voxels, 0.0, spacing=v.spacing, origin=v.origin, direction=v.direction, **kw)
                assert not np.array_equal(own.points.view(np.uint32), ref.points.view(np.uint32))
        assert ex.result.n_points > 50000
    finally:
        ex.close()
    # the Python mirror of the filter: the switch of the C++ drop-in, update for update
    f = pkg.CuberilleImageToMeshFilter(device=0)
    f.SetReproduceStaleGradient(True)
    f.SetIsoSurfaceValue(0.0)
    f.SetProjectVertexSurfaceDistanceThreshold(0.01)
    f.SetProjectVertexStepLength(0.2)
    f.SetInput(first)
    f.Update()
    f.SetInput(later[0])
    f.Update()
    kw = dict(triangles=1, project=1, threshold=0.01, step=0.2, relax=0.95, max_steps=50)
    assert_same_mesh(f.GetOutput(), oracle.run(later[0].voxels, 0.0, first=_geom(first), **kw))
    f.SetReproduceStaleGradient(False)
    f.Update()
    assert_same_mesh(f.GetOutput(), oracle.run(later[0].voxels, 0.0, **kw))


def test_buffered_region_start_index_matches_oracle(pkg, oracle, extractor):
    """cuberille_image_desc::index_start (ABI 13): a region that starts at a non-zero ITK index (a cropped image) goes through the
    index <-> point transforms and the interpolators as buffer position + start, like ITK -- host upload, resident volume, slabs and
    a held gradient against the oracle with the same start; a start outside +-2^30 is refused."""
    import torch
    c, s_ = np.cos(0.37), np.sin(0.37)
    d = np.array([[c, -s_, 0.0], [s_, c, 0.0], [0.0, 0.0, 1.0]])
    z, y, x = np.meshgrid(np.arange(30.0), np.arange(26.0), np.arange(70.0), indexing="ij")
    vox = (9.0 - np.sqrt((x - 33.3) ** 2 * 0.2 + (y - 12.5) ** 2 + (z - 14.2) ** 2) + 0.4 * np.sin(0.8 * x) * np.cos(0.6 * y + 0.3 * z)).astype(np.float32)
    nz, ny, nx = vox.shape
    for geo in (dict(spacing=(0.7, 1.3, 0.9), origin=(3.3, -2.1, 0.77), direction=d, index_start=(1000, -37, 512)),
                dict(spacing=(1.0, 1.0, 1.0), origin=(0.0, 0.0, 0.0), direction=np.eye(3), index_start=(-5, 7, 123456)),
                dict(spacing=(2.0, 0.5, 1.0), origin=(1e6, -3.0, 0.25), direction=d.T, index_start=(0, 0, -999))):
        kw = dict(triangles=1, project=1, threshold=0.01, step=0.2 * min(geo["spacing"]), relax=0.9, max_steps=20)
        ref = oracle.run(vox, 0.0, **geo, **kw)
        prm = pkg.make_params(0.0, **kw)
        extractor.extract_host(pkg.Volume(vox, **geo), prm)
        assert_same_mesh(extractor.download(), ref)
        dev = torch.from_numpy(vox).cuda()
        extractor.extract_device(dev.data_ptr(), pkg.make_desc(np.float32, (nx, ny, nz), **geo), prm)
        assert_same_mesh(extractor.download(), ref)
        below, above = pkg.required_halo(pkg.make_desc(np.float32, (nx, ny, nz), **geo), prm)
        pts, cells, poff = [], [], 0
        for a, b in [(0, 11), (11, 12), (12, 30)]:
            lo, hi = max(a - below, 0), min(b + above, nz)
            n_p, _ = extractor.count(dev[lo:hi].data_ptr(), pkg.make_desc(np.float32, (nx, ny, hi - lo), **geo), prm, pkg._abi.Slab(nz, lo, a, b, 0, 0))
            extractor.emit(poff)
            m = extractor.download()
            pts.append(m.points)
            cells.append(m.cells)
            poff += n_p
        assert_same_mesh(pkg.Mesh(np.concatenate(pts), np.concatenate(cells)), ref)
    # a held gradient keeps the first volume's start index with its geometry
    first = dict(spacing=(1.1, 0.9, 1.0), origin=(2.0, 2.0, -1.0), direction=d, index_start=(40, -2, 9))
    later = dict(spacing=(1.0, 1.0, 1.0), origin=(0.5, 0.0, 0.0), direction=np.eye(3), index_start=(3, 3, 3))
    kw = dict(triangles=1, project=1, threshold=0.01, step=0.2, relax=0.9, max_steps=20)
    ex = pkg.Extractor(0)
    try:
        ex.hold_gradient(True)
        ex.extract_host(pkg.Volume(vox[:20, :20, :40], **first), pkg.make_params(0.0, **kw))
        ex.extract_host(pkg.Volume(vox, **later), pkg.make_params(0.0, **kw))
        assert_same_mesh(ex.download(), oracle.run(vox, 0.0, first=(vox[:20, :20, :40], first["spacing"], first["origin"], first["direction"],
                                                                    first["index_start"]), **later, **kw))
    finally:
        ex.close()
    with pytest.raises(pkg._abi.CuberilleError) as e:
        extractor.extract_host(pkg.Volume(vox, index_start=(0, 0, 1 << 31)), pkg.make_params(0.0))
    assert e.value.code == pkg._abi.ERR_LIMIT
