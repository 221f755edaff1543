#!/usr/bin/env python3
"""File -> mesh on a compressed MetaImage: whole-file read + inflate + extract_host against the streamed pipeline
(mha.MhaStream feeding cuberille_extract_stream).  python profiles/stream_ingest.py [nz]   (1024 x 1024 x nz, uint8)"""
import os
import sys
import tempfile
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402


def main():
    nz = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    pkg = graft.load_package()
    ml = pkg.volumes.marschner_lobb(256)                         # float field in [0, 1]
    tile = (np.clip(ml, 0.0, 1.0) * 255.0).astype(np.uint8)
    vox = np.tile(tile, (nz // 256, 4, 4))[:, :1024, :1024]
    vox = np.ascontiguousarray(vox)
    path = os.path.join(tempfile.gettempdir(), "stream_ingest.mha")
    payload = zlib.compress(vox.tobytes(), 1)
    with open(path, "wb") as f:
        f.write(("ObjectType = Image\nNDims = 3\nBinaryData = True\nBinaryDataByteOrderMSB = False\nCompressedData = True\n"
                 "CompressedDataSize = %d\nTransformMatrix = 1 0 0 0 1 0 0 0 1\nOffset = 0 0 0\nElementSpacing = 1 1 1\n"
                 "DimSize = 1024 1024 %d\nElementType = MET_UCHAR\nElementDataFile = LOCAL\n" % (len(payload), vox.shape[0])).encode())
        f.write(payload)
    print("volume 1024x1024x%d uint8, %.0f MB, %.0f MB compressed" % (vox.shape[0], vox.nbytes / 1e6, len(payload) / 1e6), flush=True)
    del payload
    ex = pkg.Extractor(0)
    prm = pkg.make_params(128, triangles=True, project=True, threshold=0.5, step=0.25, relax=0.95, max_steps=50)
    ex.extract_host(pkg.Volume(vox[:64]), prm)                   # warm the context
    for rep in range(2):
        t0 = time.perf_counter()
        vol = pkg.read_mha(path)
        t1 = time.perf_counter()
        ex.extract_host(vol, prm)
        t2 = time.perf_counter()
        whole = (ex.result.n_points, ex.result.n_cells)
        del vol
        t3 = time.perf_counter()
        ex.extract_mha(path, prm)
        t4 = time.perf_counter()
        assert (ex.result.n_points, ex.result.n_cells) == whole
        print("whole file: read+inflate %.3f s, upload+extract %.3f s, total %.3f s | streamed: %.3f s | %d points %d cells" %
              (t1 - t0, t2 - t1, t2 - t0, t4 - t3, whole[0], whole[1]), flush=True)
    os.unlink(path)


if __name__ == "__main__":
    main()
