"""MetaImage (.mha) reader/writer for the volumes the cuberille path consumes.

The reference reads its inputs with itk::ImageFileReader (MetaImageIO)
(/root/reference/Testing/CuberilleTest01.cxx:113-117); every shipped volume
(/root/reference/Data/*.mha) is a text header followed by a zlib-compressed local
payload (`ElementDataFile = LOCAL`, `CompressedData = True`).  This is the Python
host-side equivalent (the C++ one lives in itk_lite/itkImageFileReader.h).
"""
import zlib

import numpy as np

_ELEMENT_TYPES = {
    "MET_UCHAR": np.uint8, "MET_CHAR": np.int8, "MET_USHORT": np.uint16, "MET_SHORT": np.int16,
    "MET_UINT": np.uint32, "MET_INT": np.int32, "MET_FLOAT": np.float32, "MET_DOUBLE": np.float64,
}
_ELEMENT_NAMES = {np.dtype(v): k for k, v in _ELEMENT_TYPES.items()}


class Volume:
    """A 3-D image: voxels[z, y, x] plus ITK-style geometry."""

    def __init__(self, voxels, spacing=(1.0, 1.0, 1.0), origin=(0.0, 0.0, 0.0), direction=None, index_start=(0, 0, 0)):
        # index_start: (x, y, z) ITK index of voxels[0, 0, 0] -- GetBufferedRegion().GetIndex(); 0 for an image read from a file
        self.index_start = tuple(int(v) for v in index_start)
        self.voxels = np.ascontiguousarray(voxels)
        self.spacing = tuple(float(s) for s in spacing)
        self.origin = tuple(float(o) for o in origin)
        self.direction = np.eye(3) if direction is None else np.asarray(direction, dtype=np.float64).reshape(3, 3)

    @property
    def dims(self):
        nz, ny, nx = self.voxels.shape
        return (nx, ny, nz)


def _parse_header(path, raw):
    """Header fields of a MetaImage whose first bytes are `raw`; returns (fields, offset of the pixel data)."""
    header = {}
    pos = 0
    while True:
        end = raw.index(b"\n", pos)
        line = raw[pos:end].decode("ascii", "replace").strip()
        pos = end + 1
        if "=" not in line:
            continue
        key, val = (t.strip() for t in line.split("=", 1))
        header[key] = val
        if key == "ElementDataFile":
            break
    if header.get("ObjectType", "Image") != "Image":
        raise ValueError("%s: ObjectType %r is not Image" % (path, header.get("ObjectType")))
    ndims = int(header.get("NDims", "3"))
    if ndims != 3:
        raise ValueError("%s: only NDims = 3 is supported, got %d" % (path, ndims))
    if header["ElementDataFile"] != "LOCAL":
        raise ValueError("%s: only ElementDataFile = LOCAL is supported" % path)
    if int(header.get("ElementNumberOfChannels", "1")) != 1:
        raise ValueError("%s: only scalar pixels are supported" % path)
    if header["ElementType"] not in _ELEMENT_TYPES:
        raise ValueError("%s: unsupported ElementType %s" % (path, header["ElementType"]))
    return header, pos


def _geometry(header):
    dtype = np.dtype(_ELEMENT_TYPES[header["ElementType"]])
    msb = header.get("BinaryDataByteOrderMSB", header.get("ElementByteOrderMSB", "False")).lower() == "true"
    dtype = dtype.newbyteorder(">" if msb else "<")
    dims = tuple(int(t) for t in header["DimSize"].split())
    spacing = tuple(float(t) for t in header.get("ElementSpacing", "1 1 1").split())
    origin = tuple(float(t) for t in header.get("Offset", header.get("Position", "0 0 0")).split())
    tm = header.get("TransformMatrix", header.get("Orientation", "1 0 0 0 1 0 0 0 1"))
    # MetaIO stores the direction cosines column-wise: row i of TransformMatrix is axis i's direction.
    direction = np.array([float(t) for t in tm.split()], dtype=np.float64).reshape(3, 3).T
    return dtype, dims, spacing, origin, direction


def read_mha(path):
    with open(path, "rb") as f:
        raw = f.read()
    header, pos = _parse_header(path, raw)
    dtype, (nx, ny, nz), spacing, origin, direction = _geometry(header)
    payload = raw[pos:]
    if header.get("CompressedData", "False").lower() == "true":
        size = header.get("CompressedDataSize")
        if size is not None:
            payload = payload[:int(size)]
        payload = zlib.decompress(payload)
    need = nx * ny * nz * dtype.itemsize
    if len(payload) < need:
        raise ValueError("%s: payload has %d bytes, header needs %d" % (path, len(payload), need))
    vox = np.frombuffer(payload[:need], dtype=dtype).reshape(nz, ny, nx).astype(dtype.newbyteorder("="))
    return Volume(vox, spacing, origin, direction)


class MhaStream:
    """A MetaImage read stretch by stretch: `info` is a Volume-like record without voxels (dims, dtype, spacing,
    origin, direction); calling the object as source(dst, z0, z1) fills dst with slices [z0, z1), which must be asked
    for in ascending order without gaps -- the contract of Extractor.extract_stream.  A zlib-compressed payload is
    inflated as it is read (zlib.decompressobj), so neither the compressed nor the inflated volume is ever held whole."""

    def __init__(self, path, read_bytes=8 << 20):
        self.path = path
        self._f = open(path, "rb")
        head = self._f.read(1 << 16)
        header, pos = _parse_header(path, head)
        self.file_dtype, self.dims, self.spacing, self.origin, self.direction = _geometry(header)
        self.dtype = np.dtype(self.file_dtype.newbyteorder("="))
        self._f.seek(pos)
        self._compressed = header.get("CompressedData", "False").lower() == "true"
        size = header.get("CompressedDataSize")
        self._left = int(size) if (self._compressed and size is not None) else None     # compressed bytes still in the file
        self._inflate = zlib.decompressobj() if self._compressed else None
        self._read_bytes = int(read_bytes)
        self._z = 0

    def _read(self, n):
        if self._left is not None:
            n = min(n, self._left)
            self._left -= n
        return self._f.read(n) if n > 0 else b""

    def __call__(self, dst, z0, z1):
        if z0 != self._z:
            raise ValueError("%s: slices must be read in order: asked for %d, at %d" % (self.path, z0, self._z))
        out = dst.reshape(-1).view(np.uint8)
        need, have = out.size, 0
        while have < need:
            if self._compressed:
                piece = b""
                if self._inflate.unconsumed_tail:
                    piece = self._inflate.decompress(self._inflate.unconsumed_tail, need - have)
                else:
                    raw = self._read(self._read_bytes)
                    if not raw:
                        raise ValueError("%s: compressed payload ends %d bytes short" % (self.path, need - have))
                    piece = self._inflate.decompress(raw, need - have)
            else:
                piece = self._f.read(min(self._read_bytes, need - have))
                if not piece:
                    raise ValueError("%s: payload ends %d bytes short" % (self.path, need - have))
            out[have:have + len(piece)] = np.frombuffer(piece, dtype=np.uint8)
            have += len(piece)
        if self.file_dtype.byteorder == ">":
            dst.byteswap(inplace=True)
        self._z = z1

    def close(self):
        self._f.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def open_stream(path):
    return MhaStream(path)


def write_mha(path, vol, compress=True):
    vox = np.ascontiguousarray(vol.voxels)
    nz, ny, nx = vox.shape
    payload = vox.astype(vox.dtype.newbyteorder("<")).tobytes()
    lines = ["ObjectType = Image", "NDims = 3", "BinaryData = True", "BinaryDataByteOrderMSB = False"]
    if compress:
        payload = zlib.compress(payload)
        lines += ["CompressedData = True", "CompressedDataSize = %d" % len(payload)]
    else:
        lines += ["CompressedData = False"]
    d = np.asarray(vol.direction).T.reshape(9)
    lines += ["TransformMatrix = " + " ".join("%.17g" % v for v in d),
              "Offset = " + " ".join("%.17g" % v for v in vol.origin),
              "CenterOfRotation = 0 0 0", "AnatomicalOrientation = RAI",
              "ElementSpacing = " + " ".join("%.17g" % v for v in vol.spacing),
              "DimSize = %d %d %d" % (nx, ny, nz),
              "ElementType = " + _ELEMENT_NAMES[vox.dtype.newbyteorder("=")],
              "ElementDataFile = LOCAL"]
    with open(path, "wb") as f:
        f.write(("\n".join(lines) + "\n").encode("ascii"))
        f.write(payload)
