"""CPU tests of the host side: the C-ABI library builds for gfx950 without a GPU, loads, exports
every symbol include/cuberille_hip.h declares and fails loudly (no fallback) when there is no
device; the filter mirror keeps the reference's defaults and clamps; MHA io; volume generators."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _has_gpu(pkg):
    return pkg._abi.lib().cuberille_device_count() > 0


def test_library_builds_and_exports_the_whole_header(pkg):
    pkg._abi.build()
    lib = pkg._abi.lib()
    header = open(os.path.join(ROOT, "include", "cuberille_hip.h")).read()
    declared = set(re.findall(r"\b(cuberille_[a-z_0-9]+)\s*\(", header))
    assert declared == set(pkg._abi.EXPORTS)
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.cuberille_abi_version() == pkg._abi.ABI_VERSION == 13


def test_struct_layouts_match_the_header(pkg):
    assert C.sizeof(pkg._abi.ImageDesc) == 8 + 24 + 24 + 24 + 72 + 24      # (+ index_start, ABI 13)
    assert C.sizeof(pkg._abi.Params) == 8 + 8 + 8 + 8 + 8 + 8 + 8 + 8
    assert C.sizeof(pkg._abi.Slab) == 64
    assert C.sizeof(pkg._abi.SlabStatus) == 40
    assert C.sizeof(pkg._abi.Result) == 8 + 8 + 8 + 8 * 4 + 8 + 3 * 8


def test_no_device_means_error_not_fallback(pkg):
    if _has_gpu(pkg):
        pytest.skip("a GPU is present")
    with pytest.raises(pkg._abi.CuberilleError) as e:
        pkg.Extractor(0)
    assert e.value.code == pkg._abi.ERR_NO_DEVICE
    f = pkg.CuberilleImageToMeshFilter()
    f.SetInput(pkg.Volume(np.zeros((4, 4, 4), dtype=np.uint8)))
    with pytest.raises(pkg._abi.CuberilleError):
        f.Update()


def test_contexts_from_two_threads(pkg):
    """include/cuberille_hip.h: distinct contexts are independent, also while they are being created and destroyed
    on different threads; the text of a failed create is kept per thread.  Without a GPU every create fails with
    CUBERILLE_ERR_NO_DEVICE and its own message; with one, every thread gets a working context."""
    import threading
    lib = pkg._abi.lib()
    out = {}

    def work(i):
        res = []
        for _ in range(20):
            ctx = C.c_void_p()
            rc = lib.cuberille_create(C.byref(ctx), 0)
            text = lib.cuberille_last_error(None) if rc else b""
            res.append((rc, bytes(text)))
            if rc == 0:
                assert lib.cuberille_debug_set_option(ctx, b"no_cmap", 1) == 0
                lib.cuberille_destroy(ctx)
        out[i] = res
    threads = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for i in range(2):
        for rc, text in out[i]:
            if _has_gpu(pkg):
                assert rc == 0
            else:
                assert rc == pkg._abi.ERR_NO_DEVICE and b"no CPU fallback" in text


def test_library_never_reads_the_environment():
    """Tuning and test switches go through cuberille_debug_set_option: no getenv in the product sources."""
    csrc = os.path.join(ROOT, "midas-journal-740_amd", "csrc")
    for fn in os.listdir(csrc):
        if fn.endswith((".hip", ".cpp", ".h")):
            assert "getenv" not in open(os.path.join(csrc, fn)).read(), fn


def test_required_halo_follows_the_parameters(pkg):
    """cuberille_required_halo (host arithmetic, no GPU): 8 slices for the defaults on unit spacing; thin slices,
    a longer step or more steps need more; projection off needs the topology's 2 / 1 only."""
    d = pkg.make_desc(np.float32, (64, 64, 64))
    assert pkg.required_halo(d, pkg.make_params(0.5)) == (8, 8)
    assert pkg.required_halo(d, pkg.make_params(0.5, project=False)) == (2, 1)
    thin = pkg.make_desc(np.float32, (64, 64, 64), spacing=(1.0, 1.0, 0.25))
    assert pkg.required_halo(thin, pkg.make_params(0.5)) == (22, 22)
    assert pkg.required_halo(d, pkg.make_params(0.5, step=1.0, relax=1.0, max_steps=10))[0] == 12 + 3
    assert pkg.required_halo(d, pkg.make_params(0.5, step=0.25, relax=0.95, max_steps=0))[0] == 1 + 3


def test_required_halo_covers_the_start_of_the_walk_under_a_tilted_direction(pkg, oracle):
    """The reference takes half a spacing off every PHYSICAL axis of a corner's position (txx:266-270): under a tilted
    direction matrix with unequal spacings the walk starts slices away from the lattice corner, not half a voxel.  A slab cut
    to the old figure clamped such walks (found by tests/fuzz_campaign.py, seed 1 case 1955).  Property, on the oracle's
    meshes: no vertex starts or ends further (in slices) from the voxel slice that created it than the halo allows for
    -- the cell's far side and the gradient ring (2) taken off."""
    direction = np.array([[-0.938553308377216, 0.29130696674791057, 0.18508900145150348],
                          [-0.2721283976390384, -0.29475640305213213, -0.9160048024209143],
                          [-0.21228241220739966, -0.9100873111871742, 0.35591749533214434]])
    rng = np.random.default_rng(5)
    z, y, x = np.meshgrid(np.arange(19.0), np.arange(8.0), np.arange(40.0), indexing="ij")
    vox = np.rint(40.0 * np.sin(0.35 * x + 0.5 * y) * np.cos(0.45 * z) + rng.normal(0, 6, size=z.shape)).astype(np.int8)
    cases = [((3.0, 1.7, 0.25), direction), ((3.0, 1.7, 0.25), direction.T), ((0.5, 2.0, 1.0), direction),
             ((1.0, 1.0, 1.0), direction), ((3.0, 1.7, 0.25), np.eye(3))]
    for spacing, d in cases:
        kw = dict(triangles=False, threshold=0.0, step=0.25 * min(spacing), relax=0.95, max_steps=25)
        desc = pkg.make_desc(np.int8, (40, 8, 19), spacing, (1.0, -2.0, 0.5), d)
        below, above = pkg.required_halo(desc, pkg.make_params(0, **kw))
        geo = dict(spacing=spacing, origin=(1.0, -2.0, 0.5), direction=d)
        start = oracle.run(vox, 0, project=False, **geo, **kw).points.astype(np.float64)
        end = oracle.run(vox, 0, project=True, **geo, **kw).points.astype(np.float64)
        minv = np.linalg.inv(d @ np.diag(spacing))
        zs = ((start - np.array(geo["origin"])) @ minv.T)[:, 2]
        ze = ((end - np.array(geo["origin"])) @ minv.T)[:, 2]
        # a vertex is created by a voxel of slice k with its corner at lattice k or k + 1: with the identity that is index
        # k - 1/2 or k + 1/2; whatever the direction, the creating slice lies within one of the lattice plane
        ident = np.array_equal(d, np.eye(3))
        lattice = np.rint(zs + 0.5) if ident else None
        if ident:
            assert np.abs(zs - (lattice - 0.5)).max() < 1e-6
            assert (below, above) == pkg.required_halo(pkg.make_desc(np.int8, (40, 8, 19), spacing), pkg.make_params(0, **kw))
        else:
            # the start positions under the identity name the lattice corners (same volume, same creation order)
            s0 = oracle.run(vox, 0, project=False, spacing=spacing, **kw).points.astype(np.float64)
            lattice = np.rint(s0[:, 2] / spacing[2] + 0.5)
        far = np.maximum(np.abs(zs - (lattice - 0.5)), np.abs(ze - (lattice - 0.5))).max()
        assert far + 2.0 <= min(below, above), (spacing, far, below, above)
    # ... and the figure the campaign's case needed
    desc = pkg.make_desc(np.int8, (65, 8, 19), (3.0, 1.7, 0.25), (-2.451, 5.471, 0.531), direction)
    assert pkg.required_halo(desc, pkg.make_params(0, threshold=0.0, step=0.0625, relax=0.95, max_steps=25)) == (10, 10)


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under the package may reference it."""
    pkg_dir = os.path.join(ROOT, "midas-journal-740_amd")
    for dirpath, _, files in os.walk(pkg_dir):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h", ".cpp", ".cxx", ".txx")) or fn == "Makefile":
                text = open(os.path.join(dirpath, fn), errors="replace").read()
                assert "oracle/" not in text and "import oracle" not in text and "cuberille_oracle" not in text, fn


def test_filter_defaults_and_clamps(pkg):
    f = pkg.CuberilleImageToMeshFilter()
    # txx:33-40
    assert f.GetIsoSurfaceValue() == 1
    assert f.GetGenerateTriangleFaces() is True and f.GetProjectVerticesToIsoSurface() is True
    assert f.GetProjectVertexSurfaceDistanceThreshold() == 0.5
    assert f.GetProjectVertexStepLength() == -1.0
    assert f.GetProjectVertexStepLengthRelaxationFactor() == 0.95
    assert f.GetProjectVertexMaximumNumberOfSteps() == 50
    # h:210,216,223
    f.SetInput(pkg.Volume(np.zeros((2, 2, 2), dtype=np.uint8)))
    f.SetProjectVertexSurfaceDistanceThreshold(1e9)
    assert f.GetProjectVertexSurfaceDistanceThreshold() == 255.0
    f.SetProjectVertexSurfaceDistanceThreshold(-1)
    assert f.GetProjectVertexSurfaceDistanceThreshold() == 0.0
    f.SetProjectVertexStepLength(1e9)
    assert f.GetProjectVertexStepLength() == 100000.0
    f.SetProjectVertexStepLengthRelaxationFactor(3)
    assert f.GetProjectVertexStepLengthRelaxationFactor() == 1.0
    f.GenerateTriangleFacesOff()
    assert f.GetGenerateTriangleFaces() is False
    f.ProjectVerticesToIsoSurfaceOff()
    assert f.GetProjectVerticesToIsoSurface() is False
    g = pkg.CuberilleImageToMeshFilter()
    with pytest.raises(RuntimeError):
        g.Update()                      # missing required input


def test_mha_reader_on_reference_data(pkg, volumes):
    v = volumes("nucleon.mha")
    assert v.dims == (41, 41, 41) and v.voxels.dtype == np.uint8
    assert v.spacing == (1.0, 1.0, 1.0) and v.origin == (0.0, 0.0, 0.0)
    assert np.array_equal(v.direction, np.eye(3))
    assert volumes("silicium.mha").dims == (104, 40, 40)
    assert int((v.voxels >= 140).sum()) == 6996          # SURVEY.md section 4 [PROBE]


def test_mha_roundtrip(pkg, tmp_path):
    rng = np.random.default_rng(5)
    for dt in (np.uint8, np.int16, np.float32):
        vox = (rng.random((3, 4, 5)) * 100).astype(dt)
        vol = pkg.Volume(vox, spacing=(0.5, 1.5, 2.0), origin=(1.0, 2.0, 3.0))
        for compress in (True, False):
            p = str(tmp_path / ("v_%s_%d.mha" % (np.dtype(dt).name, compress)))
            pkg.write_mha(p, vol, compress)
            back = pkg.read_mha(p)
            assert np.array_equal(back.voxels, vox) and back.spacing == vol.spacing and back.origin == vol.origin


def test_volume_generators_numpy_vs_torch(pkg):
    import torch
    vol = pkg.volumes
    a = vol.sphere_sdf(24, 3, 17)
    b = vol.sphere_sdf(24, 3, 17, xp=torch).numpy()
    assert a.dtype == np.float32 and np.array_equal(a.view(np.uint32), b.view(np.uint32))
    a = vol.gradient_noise(70, 33, 40, 5, 29)
    b = vol.gradient_noise(70, 33, 40, 5, 29, xp=torch).numpy()
    assert a.dtype == np.uint8 and np.array_equal(a, b)
    assert a.std() > 5 and 20 < a.mean() < 235
    m = vol.marschner_lobb(32)
    assert m.dtype == np.float32 and m[0].max() == 0 and m[:, 0].max() == 0 and m[:, :, -1].max() == 0
    assert 0.0 <= m.min() and m.max() <= 1.0 and (m[1:-1, 1:-1, 1:-1] > 0.5).any()
    # slabs of the generators tile the whole volume
    whole = vol.marschner_lobb(32)
    parts = np.concatenate([vol.marschner_lobb(32, 0, 10), vol.marschner_lobb(32, 10, 32)])
    assert np.array_equal(whole, parts)
    stacked = vol.marschner_lobb(16, 0, 32, period=16)
    assert np.allclose(stacked[:16], stacked[16:]) and np.allclose(stacked[:16], vol.marschner_lobb(16))


def test_dropin_header_compiles_the_unchanged_reference_driver(pkg):
    """Where the reference tree is present (the build container), CuberilleTest01.cxx and examples.cxx
    compile unchanged against itk/itkCuberilleImageToMeshFilter.h; without a GPU the binary fails the
    reference's way: itk::ExceptionObject caught, EXIT_FAILURE (CuberilleTest01.cxx:207-212)."""
    import subprocess
    ref = "/root/reference/Testing/CuberilleTest01.cxx"
    if not os.path.exists(ref):
        pytest.skip("reference tree not present on this machine")
    pkg._abi.build()
    itk_dir = os.path.join(ROOT, "midas-journal-740_amd", "itk")
    subprocess.check_call(["make", "-s", "-C", itk_dir])
    exe = os.path.join(itk_dir, "build", "CuberilleTest01")
    assert os.path.exists(exe) and os.path.exists(os.path.join(itk_dir, "build", "Examples"))
    # the reference's own filter sources must not have been compiled in
    syms = subprocess.run(["nm", "-D", "--undefined-only", exe], capture_output=True, text=True).stdout
    assert "cuberille_extract_host" in syms
    if _has_gpu(pkg):
        return
    r = subprocess.run([exe, "Test01", os.path.join(ROOT, "tests", "golden", "data", "blob0.mha"), "/tmp/_blob0.vtk", "200",
                        "8", "6", "0", "0"], capture_output=True, text=True)
    assert r.returncode != 0 and "ExceptionObject caught" in r.stderr and "no CPU fallback" in r.stderr


def test_flat_vtk_writer_bytes(pkg, tmp_path):
    """cuberille_write_vtk_buffers (host code of the C-ABI library, no GPU needed): the legacy-ASCII POLYDATA
    layout of itk::VTKPolyDataWriter as used at CuberilleTest01.cxx:180-187 -- 9 significant digits -- for
    triangles and quads, any thread count, including more items than one formatting chunk."""
    rng = np.random.default_rng(5)
    for n_pts, n_cells, k, threads in [(0, 0, 3, 1), (7, 3, 4, 1), (1000, 1999, 3, 3), (300000, 600001, 3, 4)]:
        pts = (rng.standard_normal((n_pts, 3)) * 10.0 ** rng.integers(-6, 7, (n_pts, 1))).astype(np.float32)
        if n_pts > 6:
            pts[0] = (0.0, -0.0, 1.0)
            pts[1] = (np.nan, np.inf, -np.inf)
            pts[2] = (1e-5, 123456792.0, 0.1)
            pts[3] = (-0.5, 40.5, 3.4028235e38)
        cells = rng.integers(0, max(n_pts, 1), (n_cells, k)).astype(np.uint64)
        if n_cells:
            cells[0, 0] = 2 ** 40 + 5
        path = str(tmp_path / "m.vtk")
        pkg.Mesh(pts, cells).write_vtk(path, threads)
        lines = ["# vtk DataFile Version 2.0", "File written by itkVTKPolyDataWriter", "ASCII", "DATASET POLYDATA",
                 "POINTS %d float" % n_pts]
        lines += ["%.9g %.9g %.9g" % tuple(float(v) for v in p) for p in pts]
        lines.append("POLYGONS %d %d" % (n_cells, n_cells * (k + 1)))
        lines += [" ".join([str(k)] + [str(int(i)) for i in c]) for c in cells]
        assert open(path, "rb").read() == ("\n".join(lines) + "\n").encode()


def test_streaming_metaimage_reader(pkg, tmp_path):
    """mha.MhaStream hands out the slices of a MetaImage stretch by stretch (the producer of
    cuberille_extract_stream): equal to read_mha on every shipped volume for any cut of the z range, on uncompressed
    and big-endian files too; out-of-order requests and short payloads are errors."""
    import glob
    import os
    from conftest import GOLDEN
    rng = np.random.default_rng(5)
    files = sorted(glob.glob(os.path.join(GOLDEN, "data", "*.mha")))
    assert len(files) == 11
    for path in files:
        vol = pkg.read_mha(path)
        with pkg.open_stream(path) as st:
            assert st.dims == vol.dims and st.dtype == vol.voxels.dtype
            assert st.spacing == vol.spacing and st.origin == vol.origin and np.array_equal(st.direction, vol.direction)
            nz = st.dims[2]
            cuts = sorted(set([0, nz] + [int(v) for v in rng.integers(1, nz, size=3)]))
            out = np.empty_like(vol.voxels)
            for a, b in zip(cuts[:-1], cuts[1:]):
                st(out[a:b], a, b)
        assert np.array_equal(out, vol.voxels), path
    vox = (rng.normal(size=(6, 5, 9)) * 1000).astype(np.int16)
    for compress in (False, True):
        p = str(tmp_path / ("v%d.mha" % compress))
        pkg.write_mha(p, pkg.Volume(vox, spacing=(0.5, 1.0, 2.0), origin=(1.0, -2.0, 3.0)), compress=compress)
        with pkg.open_stream(p, ) as st:
            out = np.empty_like(vox)
            st(out[0:4], 0, 4)
            with pytest.raises(ValueError):
                st(out[5:6], 5, 6)                       # slice 4 skipped
            st(out[4:6], 4, 6)
            assert np.array_equal(out, vox) and st.spacing == (0.5, 1.0, 2.0)
    # big-endian payload
    raw = open(str(tmp_path / "v0.mha"), "rb").read()
    head, data = raw[:raw.index(b"ElementDataFile")], raw[raw.index(b"ElementDataFile"):]
    body = data[data.index(b"\n") + 1:]
    swapped = np.frombuffer(body, dtype="<i2").astype(">i2").tobytes()
    with open(str(tmp_path / "be.mha"), "wb") as f:
        f.write(head.replace(b"BinaryDataByteOrderMSB = False", b"BinaryDataByteOrderMSB = True") + b"ElementDataFile = LOCAL\n" + swapped)
    assert np.array_equal(pkg.read_mha(str(tmp_path / "be.mha")).voxels, vox)
    with pkg.open_stream(str(tmp_path / "be.mha")) as st:
        out = np.empty_like(vox)
        st(out, 0, 6)
        assert np.array_equal(out, vox)
    # a payload that ends early
    with open(str(tmp_path / "short.mha"), "wb") as f:
        f.write(raw[:-40])
    with pkg.open_stream(str(tmp_path / "short.mha")) as st:
        with pytest.raises(ValueError):
            st(np.empty_like(vox), 0, 6)


def test_c_abi_from_plain_c(pkg):
    """include/cuberille_hip.h is C99 and the library links into a plain C program (examples/extract_raw.c, built by
    __graft_entry__.build()): without a device it says so and exits with 2 -- no fallback on that side either."""
    import subprocess
    import __graft_entry__ as graft
    exe = graft.build_c_example()
    r = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert r.returncode == 1 and "usage" in r.stderr
    if not _has_gpu(pkg):
        r = subprocess.run([exe, "nothing.raw", "4", "4", "4", "u8", "1", "out.vtk"], capture_output=True, text=True, timeout=60)
        assert r.returncode == 2 and "no gfx950 device" in r.stderr, (r.returncode, r.stderr)


def _host_walk_exe():
    import subprocess
    exe = os.path.join(ROOT, "midas-journal-740_amd", "itk", "build", "host_walk")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "midas-journal-740_amd", "itk"), "build/host_walk"])
    return exe


@pytest.mark.parametrize("threads", [1, 3])
def test_host_walk_through_a_nonlinear_interpolator(pkg, oracle, tmp_path, threads):
    """The drop-in filter's host walk (itkCuberilleImageToMeshFilter.txx: HostGradient + HostWalk, taken for any
    TInterpolator the kernels do not implement, h:110 / txx:455) through an interpolator whose Evaluate() is NOT the
    linear one -- it blends the image with a second, smoothed image: every vertex lands, bit for bit, where a Python
    restatement of txx:439-474 over the oracle's pinned primitives puts it.  Needs no GPU: the start points are the
    oracle's unprojected vertices.  threads 3: the opt-in threaded form gives the same points."""
    import subprocess
    from restate import blend_field, blend_value, py_default_walk
    vol, smooth = blend_field()
    n = vol.shape[0]
    kw = dict(threshold=0.02, step=0.25, relax=0.95, max_steps=30)
    flat = oracle.run(vol, 0.0, triangles=False, project=False, **kw)
    assert len(flat.points) > 300
    vol.tofile(str(tmp_path / "v.raw"))
    smooth.tofile(str(tmp_path / "s.raw"))
    flat.points.tofile(str(tmp_path / "start.raw"))
    r = subprocess.run([_host_walk_exe(), "walk", str(tmp_path / "v.raw"), str(tmp_path / "s.raw"), str(n), "0.0", "0.02", "0.25",
                        "0.95", "30", str(tmp_path / "start.raw"), str(len(flat.points)), str(tmp_path / "out.raw"), str(threads)],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout, r.stderr)
    got = np.fromfile(str(tmp_path / "out.raw"), dtype=np.float32).reshape(-1, 3)
    value = blend_value(oracle, vol, smooth)
    moved = 0
    for i, v in enumerate(flat.points):
        want, _ = py_default_walk(oracle, vol, value, 0.0, v, kw["threshold"], kw["step"], kw["relax"], kw["max_steps"])
        assert np.array_equal(np.asarray(want, dtype=np.float32).view(np.uint32), got[i].view(np.uint32)), i
        moved += int(not np.array_equal(got[i], v))
    assert moved > len(flat.points) // 2
    # ... and it is not the linear interpolator's walk
    lin = oracle.run(vol, 0.0, triangles=False, project=True, **kw)
    assert not np.array_equal(lin.points, got)


def test_iso_value_of_64bit_pixels_is_cast_like_the_reference(pkg, oracle):
    """m_IsoSurfaceValue is an InputPixelType (h:180-181): a fractional value handed to a `long` image is truncated
    toward zero by the C cast, like for every narrower integer type -- not dropped to 0 (round-3 advisor finding).  Values
    no 64-bit type holds are refused by the entry points; the oracle's loader follows the same rule."""
    prm = pkg.make_params(100.5)
    assert prm.iso_value_int == 100 and prm.iso_value == 100.5
    assert pkg.make_params(-100.5).iso_value_int == -100
    assert pkg.make_params(2 ** 63 + 5).iso_value_int == (2 ** 63 + 5) - 2 ** 64      # uint64: the same 64 bits
    assert pkg.make_params((3 << 55) + 1).iso_value_int == (3 << 55) + 1               # not a double: travels as an integer
    from midas_journal_740_amd.cuberille import check_iso
    for code, bad in ((8, float("nan")), (8, float("inf")), (8, 2.0 ** 63), (8, -2.0 ** 63 - 4096.0), (9, -1.0), (9, 2.0 ** 64)):
        with pytest.raises(pkg._abi.CuberilleError) as e:
            check_iso(code, pkg.make_params(bad))
        assert e.value.code == pkg._abi.ERR_ARGUMENT
    check_iso(8, pkg.make_params(-2.0 ** 63))
    check_iso(9, pkg.make_params(2 ** 64 - 1))
    check_iso(6, pkg.make_params(float("nan")))     # a float image takes any iso value (txx:139-141: every compare false)
    rng = np.random.default_rng(5)
    small = rng.integers(0, 200, size=(6, 7, 9))
    kw = dict(triangles=1, project=1, threshold=0.5, step=0.25, relax=0.95, max_steps=30)
    a = oracle.run(small.astype(np.int64), 100.5, **kw)
    for other in (oracle.run(small.astype(np.int64), 100, **kw), oracle.run(small.astype(np.int32), 100.5, **kw),
                  oracle.run(small.astype(np.uint64), 100.9, **kw)):
        assert np.array_equal(a.cells, other.cells) and np.array_equal(a.points.view(np.uint32), other.points.view(np.uint32))
    with pytest.raises(ValueError):
        oracle.run(small.astype(np.uint64), -1.0, **kw)


def test_bulk_mesh_fill_under_sanitizers():
    """The drop-in's mesh fill (all cells in one slab the mesh carries in its MetaDataDictionary, CellsAllocatedAsStaticArray:
    what replaces the reference's heap object per face, txx:309-329) needs no GPU: itk/tests/mesh_fill.cxx built with
    -fsanitize=address,undefined -- fill, read back, refill the same mesh, Initialize(), a mesh that outlives every other
    owner, destruction; leaks count as failures (LeakSanitizer)."""
    import subprocess
    itk = os.path.join(ROOT, "midas-journal-740_amd", "itk")
    exe = os.path.join(itk, "build", "mesh_fill_asan")
    subprocess.check_call(["make", "-s", "-C", itk, "build/mesh_fill_asan"])
    r = subprocess.run([exe, "200000"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "mesh fill ok" in r.stdout, (r.stdout[-300:], r.stderr[-2000:])


def test_lattice_corner_closed_forms_equal_the_activation_chains():
    """The dense form of the count kernel (csrc/cuberille_kernels.hip: lat_row_closed) reads "voxel X / voxel X-1 creates
    this lattice corner" off closed forms over the eight inside bits of the 2x2x2 block around the corner; the per-voxel
    form and the border waves evaluate the chains of activation terms they were derived from (SURVEY.md section 8a items
    2-3; txx:164-194: the first voxel in raster order that has a face at the corner creates it).  All 256 blocks x the four
    corner rows, away from the image border."""
    import itertools

    def T(b, e):        # member e is inside and one of its three neighbours inside the block is not
        return b[e] & (1 - (b[e ^ 1] & b[e ^ 2] & b[e ^ 4]))

    def chains(b, d):   # a larger member code comes earlier in raster order
        s = 0
        for e in range(d + 2, 8):
            s |= T(b, e)
        return T(b, d) & (1 - (T(b, d + 1) | s)), T(b, d + 1) & (1 - s)

    def closed(b, d):
        b0, b1, b2, b3, b4, b5, b6, b7 = b
        n = lambda v: 1 - v     # noqa: E731
        if d == 0:
            z = b2 | b3 | b4 | b5 | b6 | b7
            return b0 & n(b1 | z), b1 & n(z)
        if d == 2:
            o = b4 | b5 | b6 | b7
            return b2 & n(b3 | o), b3 & n(o)
        if d == 4:
            w = b2 & b3 & b4 & b5 & b6 & b7
            return (b4 & n(b5 | b6 | b7)) | (w & b1 & n(b0)), (b5 & n(b6 | b7)) | (w & n(b1))
        return b6 & (n(b7) | (b5 & b3 & n(b4 & b2))), b7 & n(b6 & b5 & b3)

    for b in itertools.product((0, 1), repeat=8):
        for d in (0, 2, 4, 6):
            assert chains(b, d) == closed(b, d), (b, d)
