"""-m gpu: the threshold sweep (txx:139-141: inside <=> !(pixel < iso), in the pixel type) at the sizes where it is a kernel of
its own -- buffers of 256 MiB and more -- for rows that are NOT whole 64-voxel words and for pointers that are not 16-byte
aligned (k_classify_span_rows): packed bits against a torch threshold, slice occupancy, counts against the closed form, and one
such volume byte for byte against the oracle."""
import numpy as np
import pytest

from conftest import assert_same_mesh
from gpu_helpers import _closed_form_counts_torch, _host_threads

pytestmark = pytest.mark.gpu


def _field(shape, dtype, seed=5):
    """Smooth blobs + noise in the pixel type (neither empty nor everything), a torch tensor on the GPU, and an iso value."""
    import torch
    nz, ny, nx = shape
    tdt = {np.uint16: torch.int32, np.int16: torch.int16, np.int8: torch.int8, np.uint8: torch.uint8, np.uint32: torch.int64,
           np.int32: torch.int32, np.float32: torch.float32, np.float64: torch.float64, np.int64: torch.int64,
           np.uint64: torch.int64}[dtype]
    g = torch.Generator(device="cuda").manual_seed(seed)
    z = torch.arange(nz, device="cuda", dtype=torch.float32)[:, None, None]
    y = torch.arange(ny, device="cuda", dtype=torch.float32)[None, :, None]
    x = torch.arange(nx, device="cuda", dtype=torch.float32)[None, None, :]
    field = torch.sin(z * 0.11) + torch.sin(y * 0.07 + 1.0) + torch.sin(x * 0.05 + 2.0)
    field += (torch.rand(shape, device="cuda", generator=g) - 0.5) * 0.02
    field.clamp_(-2.99, 2.99)
    info = np.iinfo(dtype) if np.dtype(dtype).kind in "iu" else None
    if info is not None:
        lo, hi = (float(info.min) * 0.9, float(info.max) * 0.9) if np.dtype(dtype).itemsize < 8 else (-2.0 ** 40, 2.0 ** 40)
        if info.min == 0:
            lo = 0.0
        vol = ((field + 3.0) / 6.0 * (hi - lo) + lo).to(torch.float64).round().to(tdt)
        iso = int(round((lo + hi) / 2.0))
    else:
        vol = field.to(tdt)
        iso = 0.125
    return vol, iso


@pytest.mark.parametrize("dtype,shape,skew", [
    (np.uint8, (300, 1000, 1000), 0), (np.int8, (1100, 500, 500), 3), (np.uint16, (540, 500, 500), 0),
    (np.int16, (300, 700, 650), 1), (np.uint32, (280, 500, 500), 0), (np.int32, (70, 1000, 1000), 1),
    (np.float32, (270, 500, 500), 0), (np.float32, (270, 500, 500), 3), (np.float64, (140, 500, 500), 1),
    (np.int64, (100, 600, 600), 0), (np.uint64, (100, 600, 600), 1),
    # slices far shorter than a span's rows (65 slices per span: the occupancy of a word takes the division)
    (np.uint8, (70000, 3, 1300), 5),
    # rows of ONE word (17 voxels) and of two (65): every word is a row's first and, or, last
    (np.uint8, (4096, 4096, 17), 1), (np.float32, (1024, 1024, 65), 0),
    # whole-word rows behind a pointer that is not 16-byte aligned take the same kernel
    (np.float32, (257, 512, 512), 1), (np.uint8, (1025, 512, 512), 7)])
def test_ragged_span_sweep_every_pixel_type(pkg, extractor, dtype, shape, skew):
    """k_classify_span_rows<T> for every pixel type, rows of 8 ... 21 words that end inside their last word, the pointer
    `skew` elements off its allocation: packed inside bits equal a torch threshold (tail bits of every row zero), the slice
    occupancy equals "any inside voxel in the slice" (with empty slices at both ends and in the middle), counts equal the
    closed form (txx:139-141, 164-173)."""
    import torch
    nz, ny, nx = shape
    vol, iso = _field(shape, dtype)
    # empty slices: quirk Q1's occupancy must see them (everything below the iso value)
    lowest = vol.min()
    for a, b in ((0, 2), (nz // 2, nz // 2 + 3), (nz - 1, nz)):
        vol[a:b] = lowest
    narrow = {np.uint16: torch.int16, np.uint32: torch.int32}
    dev = vol.to(narrow[dtype]) if dtype in narrow else vol
    item = np.dtype(dtype).itemsize
    assert dev.element_size() == item and dev.numel() * item >= (256 << 20)
    raw = torch.empty(dev.numel() + 16, dtype=dev.dtype, device="cuda")
    raw[skew:skew + dev.numel()] = dev.reshape(-1)
    del dev
    torch.cuda.synchronize()
    ptr = raw.data_ptr() + skew * item
    assert nx % 64 != 0 or ptr % 16 != 0
    inside = vol >= iso
    want_pts, want_quads = _closed_form_counts_torch(inside)
    assert 1000 < want_quads
    # (the closed form knows nothing of quirk Q1, and the empty slices in the middle make the reference re-use vertices:
    #  switched off for the count, on for a second count that must find fewer vertices and the same quads)
    plain = pkg.make_params(iso, triangles=False, project=False, q1=False)
    res = extractor.extract_device(ptr, pkg.make_desc(dtype, (nx, ny, nz)), plain)
    assert (int(res.n_points), int(res.n_cells)) == (want_pts, want_quads)
    res = extractor.extract_device(ptr, pkg.make_desc(dtype, (nx, ny, nz)), pkg.make_params(iso, triangles=False, project=False))
    assert int(res.n_points) < want_pts and int(res.n_cells) == want_quads
    W = (nx + 63) // 64
    words = torch.from_numpy(extractor.debug_bits((nx, ny, nz)).view(np.int64)).cuda()
    shifts = torch.arange(64, device="cuda", dtype=torch.int64)
    step = max(1, (1 << 24) // (ny * W * 64))
    for z0 in range(0, nz, step):
        bits = ((words[z0:z0 + step, :, :, None] >> shifts) & 1).bool().reshape(-1, ny, W * 64)
        assert torch.equal(bits[:, :, :nx], inside[z0:z0 + step]), "packed bits differ from the threshold in slices %d.." % z0
        assert not bits[:, :, nx:].any(), "bits beyond the end of a row in slices %d.." % z0
    occ = extractor.slice_occupancy(nz)
    assert np.array_equal(occ != 0, inside.reshape(nz, -1).any(1).cpu().numpy())
    # the plain sweep (development switch) packs the same words
    extractor.debug_option("classify_variant", 1)
    try:
        res2 = extractor.extract_device(ptr, pkg.make_desc(dtype, (nx, ny, nz)), plain)
        assert (int(res2.n_points), int(res2.n_cells)) == (want_pts, want_quads)
        assert torch.equal(torch.from_numpy(extractor.debug_bits((nx, ny, nz)).view(np.int64)).cuda(), words)
    finally:
        extractor.debug_option("defaults", 0)
    del vol, raw, inside, words


def test_ragged_1000_wide_volume_matches_oracle(pkg, oracle, extractor):
    """A volume of the size and shape real data has -- 1000 x 1000 voxels per slice, 300 slices, float32, 1.2 GB: rows of 15 5/8
    words -- through the production launch shapes (k_classify_span_rows, the count, blind launches from the second call on),
    byte for byte against the oracle: slices 350 .. 650 of the 1000^3 sphere field of bench.py (configs[2]'s generator and
    parameters), triangles and projection on.  The band cuts the sphere open at both ends: the reference's open border (Q2)."""
    import torch
    n, a, b = 1000, 350, 650
    vol = pkg.volumes.sphere_sdf(n, a, b, xp=torch, device=torch.device("cuda", 0)).contiguous()
    # (the generator's z is absolute; the extractor sees a volume of its own with origin 0: same voxels, that is all that matters)
    torch.cuda.synchronize()
    desc = pkg.make_desc(np.float32, (n, n, b - a))
    kw = dict(triangles=1, project=1, threshold=0.05, step=0.25, relax=0.95, max_steps=50)
    prm = pkg.make_params(0.0, **kw)
    for _ in range(3):                         # the third call runs blind, sized by the second
        res = extractor.extract_device(vol.data_ptr(), desc, prm)
    mesh = extractor.download()
    ref = oracle.run(vol.cpu().numpy(), 0.0, gradient_threads=_host_threads(), **kw)
    assert int(res.n_points) == len(ref.points) > 1000000
    assert_same_mesh(mesh, ref)
    assert int(res.proj_iterations) == ref.info["proj_iterations"]
    # the same volume from HOST memory: above a GiB the upload is chunked (32 MiB = 8 slices at a time, each chunk swept by
    # k_classify_span_rows while the next one crosses the link) -- the same mesh
    host = vol.cpu().numpy()
    del vol
    res = extractor.extract_host(pkg.Volume(host), prm)
    assert int(res.proj_iterations) == ref.info["proj_iterations"]
    assert_same_mesh(extractor.download(), ref)


# ---- moved here from test_gpu_parity.py in round 5's split: the sweep at smaller sizes and for whole-word rows -----------

def test_packed_bits_match_threshold(pkg, extractor):
    rng = np.random.default_rng(3)
    for shape in [(3, 4, 70), (2, 3, 128), (5, 2, 64)]:
        vox = rng.integers(0, 255, size=shape, dtype=np.uint8)
        vol = pkg.Volume(vox)
        extractor.extract_host(vol, pkg.make_params(100, project=False))
        words = extractor.debug_bits(vol.dims)
        nx = shape[2]
        bits = np.unpackbits(words.view(np.uint8), axis=-1, bitorder="little")[..., :nx].astype(bool)
        assert np.array_equal(bits, vox >= 100)
        tail = np.unpackbits(words.view(np.uint8), axis=-1, bitorder="little")[..., nx:]
        assert not tail.any()


@pytest.mark.parametrize("dtype", [np.uint8, np.int16, np.float32, np.float64])
def test_ragged_rows_at_every_pointer_alignment(pkg, oracle, extractor, dtype):
    """Rows that are not whole 64-voxel words go through the flat-stream threshold + row repack; the stream
    starts at the 16-byte boundary below the first voxel, so every misalignment of the device pointer (and
    the old one-voxel-per-lane kernel, option no_stream_classify) must give the oracle's mesh."""
    import torch
    rng = np.random.default_rng(11)
    item = np.dtype(dtype).itemsize
    for shape in [(3, 5, 71), (2, 3, 1), (4, 2, 129), (1, 1, 300), (5, 7, 63)]:
        vol = (rng.random(shape) * 200).astype(dtype)
        want = oracle.run(vol, 100, triangles=1, project=1, threshold=0.5, step=0.25, relax=0.95, max_steps=20)
        nz, ny, nx = shape
        desc = pkg.make_desc(dtype, (nx, ny, nz))
        prm = pkg.make_params(100, triangles=True, project=True, threshold=0.5, step=0.25, relax=0.95, max_steps=20)
        raw = torch.zeros(vol.nbytes + 64, dtype=torch.uint8, device="cuda")
        for skew in range(0, 16, item):
            raw.zero_()
            raw[skew:skew + vol.nbytes] = torch.from_numpy(vol.view(np.uint8).reshape(-1)).cuda()
            torch.cuda.synchronize()
            extractor.extract_device(raw.data_ptr() + skew, desc, prm)
            assert_same_mesh(extractor.download(), want)
    extractor.debug_option("no_stream_classify", 1)
    try:
        extractor.extract_device(raw.data_ptr() + skew, desc, prm)
        assert_same_mesh(extractor.download(), want)
    finally:
        extractor.debug_option("defaults", 0)


@pytest.mark.parametrize("dtype,shape", [
    (np.uint16, (512, 512, 512)), (np.int16, (512, 512, 512)), (np.int8, (1024, 512, 512)), (np.uint8, (1024, 512, 512)),
    (np.uint32, (320, 512, 512)), (np.int32, (320, 512, 512)), (np.float64, (256, 512, 512)),
    (np.int64, (128, 512, 512)), (np.uint64, (128, 512, 512))])
def test_span_sweep_every_pixel_type(pkg, extractor, dtype, shape):
    """k_classify_span<T> -- the sweep every launch of 256 MiB or more takes -- for every pixel type the library is
    instantiated for (round 2 only ever ran it for float and uint8; the 2-voxels-per-lane group OR of the 8-byte types
    ran nowhere): packed inside bits equal a torch threshold, counts equal the closed form (txx:139-141, 164-173)."""
    import torch
    nz, ny, nx = shape
    tdt = {np.uint16: torch.int32, np.int16: torch.int16, np.int8: torch.int8, np.uint8: torch.uint8, np.uint32: torch.int64,
           np.int32: torch.int32, np.float64: torch.float64, np.int64: torch.int64, np.uint64: torch.int64}[dtype]
    g = torch.Generator(device="cuda").manual_seed(5)
    # smooth blobs + noise, so that the surface is neither empty nor everything
    z = torch.arange(nz, device="cuda", dtype=torch.float32)[:, None, None]
    y = torch.arange(ny, device="cuda", dtype=torch.float32)[None, :, None]
    x = torch.arange(nx, device="cuda", dtype=torch.float32)[None, None, :]
    field = torch.sin(z * 0.11) + torch.sin(y * 0.07 + 1.0) + torch.sin(x * 0.05 + 2.0)
    field += (torch.rand(shape, device="cuda", generator=g) - 0.5) * 0.02
    field.clamp_(-2.99, 2.99)
    info = np.iinfo(dtype) if np.dtype(dtype).kind in "iu" else None
    if info is not None:
        lo, hi = (float(info.min) * 0.9, float(info.max) * 0.9) if np.dtype(dtype).itemsize < 8 else (-2.0 ** 40, 2.0 ** 40)
        if info.min == 0:
            lo = 0.0
        vol = ((field + 3.0) / 6.0 * (hi - lo) + lo).to(torch.float64).round().to(tdt)
        iso = int(round((lo + hi) / 2.0))
    else:
        vol = field.to(tdt)
        iso = 0.125
    del field
    # (the unsigned types as the signed tensor of the same width: a narrowing torch conversion wraps like a C cast, so the
    #  bits are the unsigned value's)
    dev = vol.to({np.uint16: torch.int16, np.uint32: torch.int32}[dtype]) if dtype in (np.uint16, np.uint32) else vol
    assert dev.element_size() == np.dtype(dtype).itemsize and dev.numel() * dev.element_size() >= (256 << 20)
    torch.cuda.synchronize()
    inside = vol >= iso
    want_pts, want_quads = _closed_form_counts_torch(inside)
    assert 1000 < want_quads
    res = extractor.extract_device(dev.data_ptr(), pkg.make_desc(dtype, (nx, ny, nz)), pkg.make_params(iso, triangles=False, project=False))
    assert (int(res.n_points), int(res.n_cells)) == (want_pts, want_quads)
    words = torch.from_numpy(extractor.debug_bits((nx, ny, nz)).view(np.int64)).cuda()
    shifts = torch.arange(64, device="cuda", dtype=torch.int64)
    for z0 in range(0, nz, 32):
        bits = ((words[z0:z0 + 32, :, :, None] >> shifts) & 1).bool().reshape(-1, ny, nx)
        assert torch.equal(bits, inside[z0:z0 + 32]), "packed bits differ from the threshold in slices %d.." % z0
    # (round 5: volumes of fewer than two rounds of 4096-word spans -- all of these -- are swept in quarter spans; the whole
    #  spans of larger volumes, forced here, pack the same words)
    extractor.debug_option("classify_keep_tail", 1)
    try:
        res = extractor.extract_device(dev.data_ptr(), pkg.make_desc(dtype, (nx, ny, nz)), pkg.make_params(iso, triangles=False, project=False))
        assert (int(res.n_points), int(res.n_cells)) == (want_pts, want_quads)
        assert torch.equal(torch.from_numpy(extractor.debug_bits((nx, ny, nz)).view(np.int64)).cuda(), words)
    finally:
        extractor.debug_option("defaults", 0)
    del vol, dev, inside, words


@pytest.mark.parametrize("dtype", [np.uint8, np.int16, np.float32, np.float64])
def test_whole_word_rows_at_every_pointer_alignment(pkg, oracle, extractor, dtype):
    """Rows of whole 64-voxel words behind a device pointer that is NOT 16-byte aligned leave the vector sweep for the
    flat-stream path (or, without its scratch, the one-voxel-per-lane kernel): every byte skew, same mesh."""
    import torch
    rng = np.random.default_rng(12)
    item = np.dtype(dtype).itemsize
    prm = pkg.make_params(100, triangles=True, project=True, threshold=0.5, step=0.25, relax=0.95, max_steps=20)
    for shape in [(3, 5, 64), (2, 3, 128), (4, 2, 192)]:
        vol = (rng.random(shape) * 200).astype(dtype)
        want = oracle.run(vol, 100, triangles=1, project=1, threshold=0.5, step=0.25, relax=0.95, max_steps=20)
        nz, ny, nx = shape
        desc = pkg.make_desc(dtype, (nx, ny, nz))
        raw = torch.zeros(vol.nbytes + 64, dtype=torch.uint8, device="cuda")
        for variant in (0, 1):
            extractor.debug_option("no_stream_classify", variant)
            for skew in range(0, 16, item):
                raw.zero_()
                raw[skew:skew + vol.nbytes] = torch.from_numpy(vol.view(np.uint8).reshape(-1)).cuda()
                torch.cuda.synchronize()
                extractor.extract_device(raw.data_ptr() + skew, desc, prm)
                assert_same_mesh(extractor.download(), want)
        extractor.debug_option("defaults", 0)
