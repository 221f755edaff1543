"""MI355X-native cuberille iso-surface extraction (the hot path of midas-journal-740).

The directory name carries a hyphen, so import it through `__graft_entry__.load_package()`
(which registers it as `midas_journal_740_amd`).  Layout:

  csrc/            HIP kernels + the C ABI of include/cuberille_hip.h (libcuberille_hip.so)
  _abi.py          ctypes binding of that ABI
  cuberille.py     host side: Extractor (thin) and CuberilleImageToMeshFilter (reference surface)
  distributed.py   Z-slab sharding over torch.distributed (RCCL on GPUs)
  mha.py           MetaImage reader/writer, and the streaming reader that feeds the upload pipeline
  volumes.py       synthetic volumes of the benchmark configs
  itk/             C++ drop-in: itkCuberilleImageToMeshFilter.h + the ITK-lite shim headers
"""
from . import _abi, mha, volumes  # noqa: F401
from .cuberille import CuberilleImageToMeshFilter, Extractor, Mesh, make_desc, make_params, required_halo  # noqa: F401
from .mha import MhaStream, Volume, open_stream, read_mha, write_mha  # noqa: F401
