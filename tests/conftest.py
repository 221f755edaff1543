import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import __graft_entry__ as graft  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    return graft.load_package()


@pytest.fixture(scope="session")
def oracle():
    o = graft.load_oracle()
    o.build()
    return o


@pytest.fixture(scope="session")
def ctest_cases():
    with open(os.path.join(GOLDEN, "ctest_cases.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def volumes(pkg):
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = pkg.read_mha(os.path.join(GOLDEN, "data", name))
        return cache[name]
    return get


@pytest.fixture(scope="session")
def extractor(pkg):
    """One GPU context shared by the -m gpu tests (they run in one process)."""
    pkg._abi.build()
    ex = pkg.Extractor(0)
    # The library works on a NON-BLOCKING stream of its own: a volume that torch has just generated or copied on ITS stream is
    # only ordered before the extraction by an event the caller hands over (cuberille_slab::voxels_ready_event) or by a wait.
    # The tests that generate volumes on the GPU mostly do neither -- they used to pass because a workspace that had to grow
    # (hipFree) synchronised the device on the way -- so the shared extractor waits for torch here, except where a test hands
    # over events of its own (those test exactly that ordering).
    def waits_for_torch(method):
        inner = getattr(ex, method)

        def call(*args, **kw):
            handed_over = any(getattr(a, "voxels_ready_event", None) or getattr(a, "halo_ready_event", None)
                              for a in list(args) + list(kw.values()))
            if not handed_over and "torch" in sys.modules:
                import torch
                if torch.cuda.is_available():
                    torch.cuda.synchronize()
            return inner(*args, **kw)
        setattr(ex, method, call)
    for m in ("extract_device", "count", "step_begin", "step_classify"):
        waits_for_torch(m)
    yield ex
    ex.close()


def point_bytes(points):
    """As tests/golden/make_mesh_digests.py: float32 bits with every NaN replaced by 0x7fc00000."""
    bits = points.astype("<f4").view("<u4").copy()
    bits[np.isnan(points)] = 0x7fc00000
    return bits.tobytes()


def assert_same_mesh(mesh, ref, coords="bits"):
    """GPU mesh vs oracle mesh: identical ids and cell order; coordinates bit-identical
    (or within the stated relative tolerance when coords is a float)."""
    assert mesh.points.shape == ref.points.shape, (mesh.points.shape, ref.points.shape)
    assert mesh.cells.shape == ref.cells.shape, (mesh.cells.shape, ref.cells.shape)
    assert np.array_equal(mesh.cells, ref.cells), "cell topology / order differs"
    if coords == "bits":
        a = mesh.points.view(np.uint32)
        b = ref.points.view(np.uint32)
        nan = np.isnan(mesh.points) & np.isnan(ref.points)
        bad = (a != b) & ~nan
        assert not bad.any(), "%d coordinates differ bitwise, first at %s" % (bad.sum(), np.argwhere(bad)[0])
    else:
        np.testing.assert_allclose(mesh.points, ref.points, rtol=coords, atol=0)
