"""The N>1 host path on CPU: world_size-2 (and 3) gloo process groups exercise the slab plan, the
halo exchange (P2P send/recv) and the count all-gather / id-offset prefix without a GPU."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, nz, out_dir):
    import sys
    sys.path.insert(0, ROOT)
    import __graft_entry__ as graft
    graft.load_package()
    from midas_journal_740_amd import distributed as D
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ny, nx = 5, 7
        full = torch.arange(nz * ny * nx, dtype=torch.float32).reshape(nz, ny, nx)
        z0, z1 = D.slab_range(nz, world, rank)
        lo, hi = D.buffer_range(nz, z0, z1)
        buf = torch.full((hi - lo, ny, nx), -1.0)
        buf[z0 - lo:z1 - lo] = full[z0:z1]
        D.exchange_halos(buf, lo, hi, z0, z1, rank, world)
        assert torch.equal(buf, full[lo:hi]), "halo exchange did not reproduce the neighbours' slices"
        # counts all-gather -> exclusive prefix
        counts = D.gather_counts(100 * (rank + 1), 7 * (rank + 1), torch.device("cpu"))
        assert counts.tolist() == [[100 * (r + 1), 7 * (r + 1)] for r in range(world)]
        poff, coff = D.id_offsets(counts, rank)
        assert poff == sum(100 * (r + 1) for r in range(rank)) and coff == sum(7 * (r + 1) for r in range(rank))
        np.save(os.path.join(out_dir, "ok%d.npy" % rank), np.array([z0, z1, lo, hi]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,nz", [(2, 40), (3, 50)])
def test_halo_exchange_and_offsets_gloo(tmp_path, world, nz):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, nz, str(tmp_path)), nprocs=world, join=True)
    spans = [np.load(str(tmp_path / ("ok%d.npy" % r))) for r in range(world)]
    assert spans[0][0] == 0 and spans[-1][1] == nz
    for a, b in zip(spans[:-1], spans[1:]):
        assert a[1] == b[0]


def test_slab_plan():
    import sys
    sys.path.insert(0, ROOT)
    import __graft_entry__ as graft
    graft.load_package()
    from midas_journal_740_amd import distributed as D
    for nz, world in [(1024, 8), (1000, 3), (64, 8), (17, 2)]:
        spans = [D.slab_range(nz, world, r) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == nz
        assert all(a[1] == b[0] for a, b in zip(spans[:-1], spans[1:]))
        sizes = [b - a for a, b in spans]
        assert max(sizes) - min(sizes) <= 1
        for z0, z1 in spans:
            lo, hi = D.buffer_range(nz, z0, z1)
            assert lo == max(z0 - D.HALO, 0) and hi == min(z1 + D.HALO, nz)


def test_aliasing_check():
    import sys
    sys.path.insert(0, ROOT)
    import __graft_entry__ as graft
    graft.load_package()
    from midas_journal_740_amd import distributed as D
    occ = np.zeros(40, dtype=bool)
    occ[5:15] = True
    occ[25:30] = True                                    # gap 15..24, re-entry at 25
    assert D.aliasing_crosses_slabs(occ, [(0, 40)]) == -1                 # one rank sees everything
    assert D.aliasing_crosses_slabs(occ, [(0, 20), (20, 40)]) == 25       # prev occupied (14) is below rank 1's range
    occ2 = np.zeros(40, dtype=bool)
    occ2[5:30] = True
    assert D.aliasing_crosses_slabs(occ2, [(0, 20), (20, 40)]) == -1      # no gap, no aliasing
