// ITK-lite: itk::VTKPolyDataWriter -- legacy ASCII VTK polydata from an itk::Mesh
// (call sites: Testing/CuberilleTest01.cxx:180-187).
#ifndef ITK_LITE_VTK_POLYDATA_WRITER_H
#define ITK_LITE_VTK_POLYDATA_WRITER_H
#include "itkLite.h"

namespace itk {
template <class TInputMesh> class VTKPolyDataWriter : public Object {
public:
  typedef VTKPolyDataWriter Self;
  typedef SmartPointer<Self> Pointer;
  itkNewMacro(Self);
  itkTypeMacro(VTKPolyDataWriter, Object);
  typedef TInputMesh InputMeshType;
  void SetInput(InputMeshType *m) { m_Input = m; }
  void SetFileName(const char *n) { m_FileName = n; }
  void SetFileName(const std::string &n) { m_FileName = n; }
  void Update() { Write(); }
  void Write() {
    if (m_Input.IsNull()) itkExceptionMacro(<< "No input to writer");
    if (m_FileName.empty()) itkExceptionMacro(<< "No FileName");
    std::ofstream os(m_FileName.c_str());
    if (!os) itkExceptionMacro(<< "Unable to open file: " << m_FileName);
    os << "# vtk DataFile Version 2.0\nFile written by itkVTKPolyDataWriter\nASCII\nDATASET POLYDATA\n";
    const unsigned long np = m_Input->GetNumberOfPoints(), nc = m_Input->GetNumberOfCells();
    os << "POINTS " << np << " float\n";
    os.precision(9);
    for (unsigned long i = 0; i < np; i++) {
      const typename InputMeshType::PointType &p = m_Input->GetPoints()->GetElement(i);
      os << p[0] << " " << p[1] << " " << p[2] << "\n";
    }
    unsigned long total = 0;
    for (unsigned long i = 0; i < nc; i++) total += 1 + m_Input->GetCells()->GetElement(i)->GetNumberOfPoints();
    os << "POLYGONS " << nc << " " << total << "\n";
    for (unsigned long i = 0; i < nc; i++) {
      const typename InputMeshType::CellType *c = m_Input->GetCells()->GetElement(i);
      os << c->GetNumberOfPoints();
      for (typename InputMeshType::CellType::PointIdConstIterator it = c->PointIdsBegin(); it != c->PointIdsEnd(); ++it) os << " " << *it;
      os << "\n";
    }
    if (!os) itkExceptionMacro(<< "Error writing: " << m_FileName);
  }
protected:
  VTKPolyDataWriter() {}
  typename InputMeshType::Pointer m_Input;
  std::string m_FileName;
};
}  // namespace itk
#endif
