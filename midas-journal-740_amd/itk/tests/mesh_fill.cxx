// The bulk fill of the output mesh (itkCuberilleImageToMeshFilter.txx: CellSlab + the bulk FillCells, what replaces the reference's
// one-heap-cell-per-face loop, txx:309-329) without a GPU, so that it can run under AddressSanitizer / UBSan on any host:
// cells in one slab that the mesh carries in its MetaDataDictionary, CellsAllocatedAsStaticArray.
//   fill -> read back -> fill the same mesh again (the old slab must go) -> Initialize() -> a mesh that outlives every
//   other reference to the slab -> destruction.
#include <cstdlib>
#include <iostream>
#include <vector>

#include "itkMesh.h"
#include "itkTriangleCell.h"
#include "itkQuadrilateralCell.h"
#include "itkCuberilleImageToMeshFilter.h"

typedef itk::Mesh<float, 3> MeshType;
typedef MeshType::CellType CellType;
typedef itk::TriangleCell<CellType> TriangleCellType;
typedef itk::QuadrilateralCell<CellType> QuadCellType;

template <class TCell, unsigned int K> static bool fill_and_check(MeshType *mesh, uint64_t n, uint64_t salt)
{
  std::vector<uint64_t> ids(static_cast<size_t>(n) * K);
  for (size_t i = 0; i < ids.size(); i++) ids[i] = (i * 2654435761ull + salt) % 1000003ull;
  itk::cuberille_detail::FillCells<MeshType, TCell, K>(mesh, n ? &ids[0] : 0, n, itk::cuberille_detail::FillTag<true>());
  if (mesh->GetNumberOfCells() != n) return false;
  if (mesh->GetCellsAllocationMethod() != MeshType::CellsAllocatedAsStaticArray) return false;
  for (uint64_t c = 0; c < n; c++)
    {
    MeshType::CellAutoPointer cell;
    if (!mesh->GetCell(c, cell) || cell->GetNumberOfPoints() != K) return false;
    CellType::PointIdConstIterator it = cell->PointIdsBegin();
    for (unsigned int k = 0; k < K; k++) if (it[k] != ids[K * c + k]) return false;
    }
  typename itk::cuberille_detail::CellSlab<TCell>::Pointer slab;
  if (n && !itk::ExposeMetaData(mesh->GetMetaDataDictionary(), std::string("CuberilleCellSlab"), slab)) return false;
  return true;
}

int main(int argc, char *argv[])
{
  const uint64_t n = argc > 1 ? std::strtoull(argv[1], 0, 10) : 300000;
  bool ok = true;
  MeshType::Pointer kept;
  {
    MeshType::Pointer mesh = MeshType::New();
    ok &= fill_and_check<TriangleCellType, 3>(mesh, n, 1);        // several threads above 65 536 cells
    ok &= fill_and_check<QuadCellType, 4>(mesh, n / 2, 2);        // the same mesh again: the triangle slab is released
    ok &= fill_and_check<TriangleCellType, 3>(mesh, 0, 3);        // an empty mesh
    ok &= fill_and_check<TriangleCellType, 3>(mesh, 1000, 4);
    kept = mesh;
  }
  ok &= kept->GetNumberOfCells() == 1000;
  MeshType::CellAutoPointer cell;
  ok &= kept->GetCell(999, cell) && cell->GetNumberOfPoints() == 3;
  kept->Initialize();                                            // forgets the pointers; the slab goes with the dictionary
  ok &= kept->GetNumberOfCells() == 0;
  ok &= fill_and_check<QuadCellType, 4>(kept, 70000, 5);         // and the mesh can be filled again
  kept = 0;
  std::cout << (ok ? "mesh fill ok" : "mesh fill FAILED") << std::endl;
  return ok ? 0 : 1;
}
