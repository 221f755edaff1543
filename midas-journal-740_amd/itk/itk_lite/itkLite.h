// itkLite.h -- a minimal, from-scratch stand-in for the handful of ITK 3.x classes that the
// reference's UNCHANGED driver (/root/reference/Testing/CuberilleTest01.cxx:31-48,57-213 and
// Source/examples.cxx) and our drop-in filter header name.  Used only when real ITK is not
// installed (it is not in this image; the reference's CMakeLists.txt:7 needs it).  It is NOT
// ITK: only the members those two translation units touch exist, with ITK's signatures.
// The forwarding headers next to this file (itkImage.h, itkMesh.h, ...) carry the ITK file
// names so that `#include "itkImage.h"` resolves here.
#ifndef ITK_LITE_H
#define ITK_LITE_H

#include <cmath>
#include <cstddef>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <limits>
#include <map>
#include <sstream>
#include <stdexcept>
#include <string>
#include <sys/time.h>
#include <vector>

#define ITK_LITE 1
#define ITK_EXPORT

namespace itk {

// ------------------------------------------------------------------------------------------
// basics: Indent, ExceptionObject, NumericTraits, SmartPointer, LightObject
// ------------------------------------------------------------------------------------------
class Indent {
public:
  explicit Indent(int n = 0) : m_N(n) {}
  Indent GetNextIndent() const { return Indent(m_N + 2); }
  int m_N;
};
inline std::ostream &operator<<(std::ostream &os, const Indent &i) {
  for (int k = 0; k < i.m_N; k++) os << ' ';
  return os;
}

class ExceptionObject : public std::exception {
public:
  ExceptionObject() {}
  ExceptionObject(const char *file, unsigned int line, const char *desc = "None", const char *loc = "Unknown")
      : m_File(file), m_Line(line), m_Description(desc), m_Location(loc) {}
  virtual ~ExceptionObject() throw() {}
  virtual const char *what() const throw() { return m_Description.c_str(); }
  virtual const char *GetDescription() const { return m_Description.c_str(); }
  virtual void Print(std::ostream &os) const {
    os << "itk::ExceptionObject\n  Location: \"" << m_Location << "\"\n  File: " << m_File << "\n  Line: " << m_Line
       << "\n  Description: " << m_Description << "\n";
  }
  std::string m_File;
  unsigned int m_Line = 0;
  std::string m_Description, m_Location;
};
inline std::ostream &operator<<(std::ostream &os, const ExceptionObject &e) { e.Print(os); return os; }

#define itkExceptionMacro(x)                                                             \
  {                                                                                      \
    std::ostringstream itk_lite_msg;                                                     \
    itk_lite_msg << "itk::ERROR: " << this->GetNameOfClass() << "(" << this << "): " x;  \
    throw ::itk::ExceptionObject(__FILE__, __LINE__, itk_lite_msg.str().c_str(), "");    \
  }
#define itkGenericExceptionMacro(x)                                                      \
  {                                                                                      \
    std::ostringstream itk_lite_msg;                                                     \
    itk_lite_msg << "itk::ERROR: " x;                                                    \
    throw ::itk::ExceptionObject(__FILE__, __LINE__, itk_lite_msg.str().c_str(), "");    \
  }

template <class T> struct NumericTraits {
  typedef T ValueType;
  typedef T PrintType;
  typedef double RealType;
  static const T Zero;
  static const T One;
  static T max() { return std::numeric_limits<T>::max(); }
  static T min() { return std::numeric_limits<T>::min(); }
  static T NonpositiveMin() { return std::numeric_limits<T>::is_integer ? std::numeric_limits<T>::min() : -std::numeric_limits<T>::max(); }
};
template <class T> const T NumericTraits<T>::Zero = T(0);
template <class T> const T NumericTraits<T>::One = T(1);
template <> struct NumericTraits<unsigned char> {
  typedef unsigned char ValueType;
  typedef int PrintType;
  typedef double RealType;
  static const unsigned char Zero = 0;
  static const unsigned char One = 1;
  static unsigned char max() { return 255; }
  static unsigned char min() { return 0; }
  static unsigned char NonpositiveMin() { return 0; }
};
template <> struct NumericTraits<signed char> {
  typedef signed char ValueType;
  typedef int PrintType;
  typedef double RealType;
  static const signed char Zero = 0;
  static const signed char One = 1;
  static signed char max() { return 127; }
  static signed char min() { return -128; }
  static signed char NonpositiveMin() { return -128; }
};
template <> struct NumericTraits<bool> {
  typedef bool ValueType;
  typedef bool PrintType;
  static const bool Zero = false;
  static const bool One = true;
};

template <class T> class SmartPointer {
public:
  typedef T ObjectType;
  SmartPointer() : m_P(0) {}
  SmartPointer(T *p) : m_P(p) { Reg(); }
  SmartPointer(const SmartPointer &o) : m_P(o.m_P) { Reg(); }
  ~SmartPointer() { Unreg(); }
  SmartPointer &operator=(const SmartPointer &o) { return operator=(o.m_P); }
  SmartPointer &operator=(T *p) {
    if (p != m_P) { T *old = m_P; m_P = p; Reg(); if (old) old->UnRegister(); }
    return *this;
  }
  T *operator->() const { return m_P; }
  operator T *() const { return m_P; }
  T *GetPointer() const { return m_P; }
  bool IsNull() const { return m_P == 0; }
  bool IsNotNull() const { return m_P != 0; }
private:
  void Reg() { if (m_P) m_P->Register(); }
  void Unreg() { if (m_P) m_P->UnRegister(); m_P = 0; }
  T *m_P;
};

class LightObject {
public:
  typedef LightObject Self;
  typedef SmartPointer<Self> Pointer;
  virtual const char *GetNameOfClass() const { return "LightObject"; }
  virtual void Register() const { ++m_Ref; }
  virtual void UnRegister() const { if (--m_Ref <= 0) delete this; }
  int GetReferenceCount() const { return m_Ref; }
  void Print(std::ostream &os, Indent indent = Indent(0)) const {
    os << indent << GetNameOfClass() << " (" << this << ")\n";
    PrintSelf(os, indent.GetNextIndent());
  }
protected:
  LightObject() : m_Ref(0) {}
  virtual ~LightObject() {}
  virtual void PrintSelf(std::ostream &os, Indent indent) const { os << indent << "Reference Count: " << m_Ref << std::endl; }
  mutable int m_Ref;
};

// itk::MetaDataDictionary / MetaDataObject<T> / EncapsulateMetaData / ExposeMetaData (itkMetaDataObject.h): string-keyed,
// reference-counted values that live and die with the itk::Object that carries them
class MetaDataObjectBase : public LightObject {
public:
  typedef MetaDataObjectBase Self;
  typedef SmartPointer<Self> Pointer;
  virtual const char *GetNameOfClass() const { return "MetaDataObjectBase"; }
protected:
  MetaDataObjectBase() {}
};
template <class T> class MetaDataObject : public MetaDataObjectBase {
public:
  typedef MetaDataObject Self;
  typedef SmartPointer<Self> Pointer;
  static Pointer New() { Pointer p = new Self; return p; }
  virtual const char *GetNameOfClass() const { return "MetaDataObject"; }
  void SetMetaDataObjectValue(const T &v) { m_Value = v; }
  const T &GetMetaDataObjectValue() const { return m_Value; }
protected:
  MetaDataObject() : m_Value() {}
  T m_Value;
};
class MetaDataDictionary {
public:
  MetaDataObjectBase::Pointer &operator[](const std::string &key) { return m_Map[key]; }
  bool HasKey(const std::string &key) const { return m_Map.find(key) != m_Map.end(); }
  const MetaDataObjectBase *Get(const std::string &key) const {
    std::map<std::string, MetaDataObjectBase::Pointer>::const_iterator it = m_Map.find(key);
    return it == m_Map.end() ? 0 : it->second.GetPointer();
  }
  bool Erase(const std::string &key) { return m_Map.erase(key) != 0; }
private:
  std::map<std::string, MetaDataObjectBase::Pointer> m_Map;
};
template <class T> inline void EncapsulateMetaData(MetaDataDictionary &dict, const std::string &key, const T &value) {
  typename MetaDataObject<T>::Pointer o = MetaDataObject<T>::New();
  o->SetMetaDataObjectValue(value);
  dict[key] = o.GetPointer();
}
template <class T> inline bool ExposeMetaData(const MetaDataDictionary &dict, const std::string &key, T &out) {
  const MetaDataObject<T> *o = dynamic_cast<const MetaDataObject<T> *>(dict.Get(key));
  if (!o) return false;
  out = o->GetMetaDataObjectValue();
  return true;
}

class Object : public LightObject {
public:
  typedef Object Self;
  virtual const char *GetNameOfClass() const { return "Object"; }
  virtual void Modified() const { ++m_MTime; }
  unsigned long GetMTime() const { return m_MTime; }
  MetaDataDictionary &GetMetaDataDictionary() { if (!m_Dictionary) m_Dictionary = new MetaDataDictionary; return *m_Dictionary; }
protected:
  Object() : m_MTime(1), m_Dictionary(0) {}
  ~Object() { delete m_Dictionary; }          // after the derived classes' destructors: what the dictionary holds goes last
  mutable unsigned long m_MTime;
  MetaDataDictionary *m_Dictionary;
private:
  Object(const Object &);
  void operator=(const Object &);
};

#define itkNewMacro(x)                                  \
  static Pointer New(void) {                            \
    Pointer smartPtr = new x;                           \
    return smartPtr;                                    \
  }
#define itkTypeMacro(thisClass, superclass) \
  virtual const char *GetNameOfClass() const { return #thisClass; }
#define itkSetMacro(name, type)                 \
  virtual void Set##name(const type _arg) {     \
    if (this->m_##name != _arg) {               \
      this->m_##name = _arg;                    \
      this->Modified();                         \
    }                                           \
  }
#define itkGetMacro(name, type) \
  virtual type Get##name() { return this->m_##name; }
#define itkGetConstMacro(name, type) \
  virtual type Get##name() const { return this->m_##name; }
#define itkSetClampMacro(name, type, min, max)                              \
  virtual void Set##name(type _arg) {                                       \
    if (this->m_##name != (_arg < min ? min : (_arg > max ? max : _arg))) { \
      this->m_##name = (_arg < min ? min : (_arg > max ? max : _arg));      \
      this->Modified();                                                     \
    }                                                                       \
  }
#define itkBooleanMacro(name)                      \
  virtual void name##On() { this->Set##name(true); } \
  virtual void name##Off() { this->Set##name(false); }
#define itkSetObjectMacro(name, type)      \
  virtual void Set##name(type *_arg) {     \
    if (this->m_##name != _arg) {          \
      this->m_##name = _arg;               \
      this->Modified();                    \
    }                                      \
  }
#define itkGetObjectMacro(name, type) \
  virtual type *Get##name() { return this->m_##name.GetPointer(); }

// ------------------------------------------------------------------------------------------
// fixed-size value types
// ------------------------------------------------------------------------------------------
template <class T, unsigned int N> class FixedArray {
public:
  typedef T ValueType;
  T &operator[](unsigned int i) { return m_V[i]; }
  const T &operator[](unsigned int i) const { return m_V[i]; }
  void Fill(const T &v) { for (unsigned int i = 0; i < N; i++) m_V[i] = v; }
  bool operator==(const FixedArray &o) const { for (unsigned int i = 0; i < N; i++) if (m_V[i] != o.m_V[i]) return false; return true; }
  bool operator!=(const FixedArray &o) const { return !(*this == o); }
  static unsigned int GetLength() { return N; }
  T m_V[N];
};

template <class T, unsigned int N> class Vector : public FixedArray<T, N> {};

template <class T, unsigned int N> class CovariantVector : public FixedArray<T, N> {
public:
  typedef double RealValueType;
  RealValueType GetSquaredNorm() const { RealValueType s = 0; for (unsigned int i = 0; i < N; i++) { const RealValueType v = (*this)[i]; s += v * v; } return s; }
  RealValueType GetNorm() const { return std::sqrt(GetSquaredNorm()); }
  void Normalize() { const RealValueType n = GetNorm(); for (unsigned int i = 0; i < N; i++) (*this)[i] = static_cast<T>((*this)[i] / n); }
};

template <class T, unsigned int N> class Point : public FixedArray<T, N> {
public:
  typedef double RealType;
  Point() {}
  template <class U> Point(const Point<U, N> &o) { for (unsigned int i = 0; i < N; i++) (*this)[i] = static_cast<T>(o[i]); }
  template <class U> RealType SquaredEuclideanDistanceTo(const Point<U, N> &pa) const {
    RealType sum = 0;
    for (unsigned int i = 0; i < N; i++) { const RealType d = static_cast<RealType>(pa[i]) - static_cast<RealType>((*this)[i]); sum += d * d; }
    return sum;
  }
};

template <unsigned int N> class Index {
public:
  typedef long IndexValueType;
  IndexValueType &operator[](unsigned int i) { return m_Index[i]; }
  const IndexValueType &operator[](unsigned int i) const { return m_Index[i]; }
  void Fill(IndexValueType v) { for (unsigned int i = 0; i < N; i++) m_Index[i] = v; }
  IndexValueType m_Index[N];
};
template <unsigned int N> class Size {
public:
  typedef unsigned long SizeValueType;
  SizeValueType &operator[](unsigned int i) { return m_Size[i]; }
  const SizeValueType &operator[](unsigned int i) const { return m_Size[i]; }
  void Fill(SizeValueType v) { for (unsigned int i = 0; i < N; i++) m_Size[i] = v; }
  SizeValueType m_Size[N];
};
template <unsigned int N> class Offset {
public:
  typedef long OffsetValueType;
  OffsetValueType &operator[](unsigned int i) { return m_Offset[i]; }
  const OffsetValueType &operator[](unsigned int i) const { return m_Offset[i]; }
  OffsetValueType m_Offset[N];
};
template <unsigned int N> class ImageRegion {
public:
  typedef Index<N> IndexType;
  typedef Size<N> SizeType;
  const IndexType &GetIndex() const { return m_Index; }
  const SizeType &GetSize() const { return m_Size; }
  void SetIndex(const IndexType &i) { m_Index = i; }
  void SetSize(const SizeType &s) { m_Size = s; }
  unsigned long GetNumberOfPixels() const { unsigned long n = 1; for (unsigned int i = 0; i < N; i++) n *= m_Size[i]; return n; }
  IndexType m_Index;
  SizeType m_Size;
};
template <class T, unsigned int R, unsigned int C> class Matrix {
public:
  Matrix() { SetIdentity(); }
  void SetIdentity() { for (unsigned int r = 0; r < R; r++) for (unsigned int c = 0; c < C; c++) m[r][c] = (r == c) ? T(1) : T(0); }
  T *operator[](unsigned int r) { return m[r]; }
  const T *operator[](unsigned int r) const { return m[r]; }
  T m[R][C];
};

// ------------------------------------------------------------------------------------------
// DataObject / ProcessObject : a one-input, one-output pull pipeline
// ------------------------------------------------------------------------------------------
class ProcessObject;
class DataObject : public Object {
public:
  typedef DataObject Self;
  typedef SmartPointer<Self> Pointer;
  virtual const char *GetNameOfClass() const { return "DataObject"; }
  void DisconnectPipeline() { m_Source = 0; }
  virtual void Initialize() {}
  virtual void Update();
  ProcessObject *m_Source;   // not reference counted (like ITK's weak source link)
protected:
  DataObject() : m_Source(0) {}
};

class ProcessObject : public Object {
public:
  typedef ProcessObject Self;
  typedef SmartPointer<Self> Pointer;
  virtual const char *GetNameOfClass() const { return "ProcessObject"; }
  virtual void Update() {
    if (m_Inputs.size() < m_RequiredInputs || (m_RequiredInputs > 0 && m_Inputs[0].IsNull()))
      itkExceptionMacro(<< "Input 0 is required but not set");
    for (size_t i = 0; i < m_Inputs.size(); i++) if (m_Inputs[i].IsNotNull() && m_Inputs[i]->m_Source) m_Inputs[i]->Update();
    this->GenerateOutputInformation();
    if (m_Output.IsNotNull()) m_Output->Initialize();      // I12: the output is re-initialised each run
    this->GenerateData();
  }
  virtual void UpdateLargestPossibleRegion() { this->Update(); }
  virtual void SetNthInput(unsigned int idx, DataObject *input) {
    if (m_Inputs.size() <= idx) m_Inputs.resize(idx + 1);
    if (m_Inputs[idx].GetPointer() != input) { m_Inputs[idx] = input; this->Modified(); }
  }
  void SetNumberOfRequiredInputs(unsigned int n) { m_RequiredInputs = n; }
protected:
  ProcessObject() : m_RequiredInputs(0) {}
  ~ProcessObject() { if (m_Output.IsNotNull() && m_Output->m_Source == this) m_Output->m_Source = 0; }
  virtual void GenerateData() {}
  virtual void GenerateOutputInformation() {}
  void SetPrimaryOutput(DataObject *o) { m_Output = o; if (o) o->m_Source = this; }
  std::vector<DataObject::Pointer> m_Inputs;
  DataObject::Pointer m_Output;
  unsigned int m_RequiredInputs;
};
inline void DataObject::Update() { if (m_Source) m_Source->Update(); }

// ------------------------------------------------------------------------------------------
// Image
// ------------------------------------------------------------------------------------------
template <class TPixel, unsigned int VDim = 2> class Image : public DataObject {
public:
  typedef Image Self;
  typedef SmartPointer<Self> Pointer;
  typedef SmartPointer<const Self> ConstPointer;
  itkNewMacro(Self);
  itkTypeMacro(Image, DataObject);
  static const unsigned int ImageDimension = VDim;
  typedef TPixel PixelType;
  typedef TPixel ValueType;
  typedef TPixel InternalPixelType;
  typedef Index<VDim> IndexType;
  typedef typename IndexType::IndexValueType IndexValueType;
  typedef Size<VDim> SizeType;
  typedef Offset<VDim> OffsetType;
  typedef ImageRegion<VDim> RegionType;
  typedef double SpacingValueType;
  typedef Vector<double, VDim> SpacingType;
  typedef Point<double, VDim> PointType;
  typedef Matrix<double, VDim, VDim> DirectionType;

  void SetRegions(const RegionType &r) { m_Region = r; }
  const RegionType &GetLargestPossibleRegion() const { return m_Region; }
  const RegionType &GetBufferedRegion() const { return m_Region; }
  const RegionType &GetRequestedRegion() const { return m_Region; }
  void Allocate() { m_Buffer.assign(m_Region.GetNumberOfPixels(), TPixel()); }
  void FillBuffer(const TPixel &v) { m_Buffer.assign(m_Region.GetNumberOfPixels(), v); }
  TPixel *GetBufferPointer() { return m_Buffer.empty() ? 0 : &m_Buffer[0]; }
  const TPixel *GetBufferPointer() const { return m_Buffer.empty() ? 0 : &m_Buffer[0]; }
  const SpacingType &GetSpacing() const { return m_Spacing; }
  const PointType &GetOrigin() const { return m_Origin; }
  const DirectionType &GetDirection() const { return m_Direction; }
  void SetSpacing(const SpacingType &s) { m_Spacing = s; }
  void SetOrigin(const PointType &o) { m_Origin = o; }
  void SetDirection(const DirectionType &d) { m_Direction = d; }
  size_t ComputeOffset(const IndexType &ind) const {
    size_t off = 0, stride = 1;
    for (unsigned int i = 0; i < VDim; i++) { off += stride * (size_t)(ind[i] - m_Region.GetIndex()[i]); stride *= m_Region.GetSize()[i]; }
    return off;
  }
  const TPixel &GetPixel(const IndexType &i) const { return m_Buffer[ComputeOffset(i)]; }
  void SetPixel(const IndexType &i, const TPixel &v) { m_Buffer[ComputeOffset(i)] = v; }
  template <class TCoord> void TransformIndexToPhysicalPoint(const IndexType &index, Point<TCoord, VDim> &point) const {
    for (unsigned int r = 0; r < VDim; r++) {
      double sum = 0.0;
      for (unsigned int c = 0; c < VDim; c++) sum += m_Direction[r][c] * m_Spacing[c] * static_cast<double>(index[c]);
      point[r] = static_cast<TCoord>(sum + m_Origin[r]);
    }
  }
protected:
  Image() { m_Spacing.Fill(1.0); m_Origin.Fill(0.0); for (unsigned int i = 0; i < VDim; i++) { m_Region.m_Index[i] = 0; m_Region.m_Size[i] = 0; } }
  RegionType m_Region;
  SpacingType m_Spacing;
  PointType m_Origin;
  DirectionType m_Direction;
  std::vector<TPixel> m_Buffer;
};

// ------------------------------------------------------------------------------------------
// Mesh, cells
// ------------------------------------------------------------------------------------------
template <class TId, class TElement> class VectorContainer : public Object {
public:
  typedef VectorContainer Self;
  typedef SmartPointer<Self> Pointer;
  itkNewMacro(Self);
  typedef TId ElementIdentifier;
  typedef TElement Element;
  void InsertElement(ElementIdentifier id, const Element &e) { if (m_V.size() <= id) m_V.resize(id + 1); m_V[id] = e; }
  const Element &GetElement(ElementIdentifier id) const { return m_V[id]; }
  Element &ElementAt(ElementIdentifier id) { return m_V[id]; }
  void Reserve(ElementIdentifier n) { m_V.reserve(n); }
  ElementIdentifier Size() const { return m_V.size(); }
  void Initialize() { m_V.clear(); }
  std::vector<Element> &CastToSTLContainer() { return m_V; }
protected:
  VectorContainer() {}
  std::vector<Element> m_V;
};

// itk::MapContainer as DefaultDynamicMeshTraits uses it: identifier -> element in a std::map (no contiguous storage: a
// filter that fills meshes in bulk has to take the element-by-element road for it)
template <class TId, class TElement> class MapContainer : public Object {
public:
  typedef MapContainer Self;
  typedef SmartPointer<Self> Pointer;
  itkNewMacro(Self);
  typedef TId ElementIdentifier;
  typedef TElement Element;
  typedef std::map<TId, TElement> STLContainerType;
  void InsertElement(ElementIdentifier id, const Element &e) { m_M[id] = e; }
  const Element &GetElement(ElementIdentifier id) const { return m_M.find(id)->second; }
  Element &ElementAt(ElementIdentifier id) { return m_M[id]; }
  bool IndexExists(ElementIdentifier id) const { return m_M.count(id) != 0; }
  void Reserve(ElementIdentifier) {}
  ElementIdentifier Size() const { return m_M.size(); }
  void Initialize() { m_M.clear(); }
  STLContainerType &CastToSTLContainer() { return m_M; }
protected:
  MapContainer() {}
  STLContainerType m_M;
};

template <class TPixelType, class TCellTraits> class CellInterface;

template <class TPixelType, unsigned int VPointDimension = 3, unsigned int VMaxTopologicalDimension = VPointDimension,
          class TCoordRep = float, class TInterpolationWeight = float, class TCellPixelType = TPixelType>
class DefaultStaticMeshTraits {
public:
  typedef TPixelType PixelType;
  typedef TCellPixelType CellPixelType;
  typedef TCoordRep CoordRepType;
  typedef unsigned long PointIdentifier;
  typedef unsigned long CellIdentifier;
  typedef Point<CoordRepType, VPointDimension> PointType;
  static const unsigned int PointDimension = VPointDimension;
  struct CellTraits {
    typedef TCellPixelType PixelType;
    typedef TCoordRep CoordRepType;
    typedef unsigned long PointIdentifier;
    typedef unsigned long CellIdentifier;
    typedef PointIdentifier *PointIdIterator;
    typedef const PointIdentifier *PointIdConstIterator;
  };
  typedef CellInterface<CellPixelType, CellTraits> CellType;
  typedef VectorContainer<PointIdentifier, PointType> PointsContainer;
  typedef VectorContainer<CellIdentifier, CellType *> CellsContainer;
};

// itk::DefaultDynamicMeshTraits: the same names, points and cells in MapContainers
template <class TPixelType, unsigned int VPointDimension = 3, unsigned int VMaxTopologicalDimension = VPointDimension,
          class TCoordRep = float, class TInterpolationWeight = float, class TCellPixelType = TPixelType>
class DefaultDynamicMeshTraits
  : public DefaultStaticMeshTraits<TPixelType, VPointDimension, VMaxTopologicalDimension, TCoordRep, TInterpolationWeight, TCellPixelType> {
  typedef DefaultStaticMeshTraits<TPixelType, VPointDimension, VMaxTopologicalDimension, TCoordRep, TInterpolationWeight, TCellPixelType> Base;
public:
  typedef MapContainer<typename Base::PointIdentifier, typename Base::PointType> PointsContainer;
  typedef MapContainer<typename Base::CellIdentifier, typename Base::CellType *> CellsContainer;
};

template <class TPixelType, class TCellTraits> class CellInterface {
public:
  typedef CellInterface Self;
  typedef TPixelType PixelType;
  typedef TCellTraits CellTraits;
  typedef typename CellTraits::PointIdentifier PointIdentifier;
  typedef const PointIdentifier *PointIdConstIterator;
  class AutoPointer {          // SelfAutoPointer: owns the cell when asked to
  public:
    AutoPointer() : m_P(0), m_Own(false) {}
    ~AutoPointer() { Reset(); }
    void TakeOwnership(Self *p) { Reset(); m_P = p; m_Own = true; }
    void TakeNoOwnership(Self *p) { Reset(); m_P = p; m_Own = false; }
    Self *ReleaseOwnership() { m_Own = false; return m_P; }
    Self *GetPointer() const { return m_P; }
    Self *operator->() const { return m_P; }
    bool IsOwner() const { return m_Own; }
  private:
    AutoPointer(const AutoPointer &);
    void operator=(const AutoPointer &);
    void Reset() { if (m_Own && m_P) delete m_P; m_P = 0; m_Own = false; }
    Self *m_P;
    bool m_Own;
  };
  typedef AutoPointer CellAutoPointer;
  typedef AutoPointer SelfAutoPointer;
  virtual ~CellInterface() {}
  virtual unsigned int GetNumberOfPoints() const = 0;
  virtual void SetPointIds(PointIdConstIterator first) = 0;
  virtual PointIdConstIterator PointIdsBegin() const = 0;
  virtual PointIdConstIterator PointIdsEnd() const { return PointIdsBegin() + GetNumberOfPoints(); }
};

template <class TCellInterface, unsigned int NPoints> class FixedCell : public TCellInterface {
public:
  typedef typename TCellInterface::PointIdentifier PointIdentifier;
  typedef typename TCellInterface::PointIdConstIterator PointIdConstIterator;
  typedef typename TCellInterface::CellAutoPointer CellAutoPointer;
  typedef typename TCellInterface::SelfAutoPointer SelfAutoPointer;
  static const unsigned int NumberOfPoints = NPoints;
  virtual unsigned int GetNumberOfPoints() const { return NPoints; }
  virtual void SetPointIds(PointIdConstIterator first) { for (unsigned int i = 0; i < NPoints; i++) m_PointIds[i] = first[i]; }
  virtual PointIdConstIterator PointIdsBegin() const { return m_PointIds; }
protected:
  PointIdentifier m_PointIds[NPoints];
};
template <class TCellInterface> class TriangleCell : public FixedCell<TCellInterface, 3> {};
template <class TCellInterface> class QuadrilateralCell : public FixedCell<TCellInterface, 4> {};


template <class TPixelType, unsigned int VDimension = 3,
          class TMeshTraits = DefaultStaticMeshTraits<TPixelType, VDimension, VDimension> >
class Mesh : public DataObject {
public:
  typedef Mesh Self;
  typedef SmartPointer<Self> Pointer;
  typedef SmartPointer<const Self> ConstPointer;
  itkNewMacro(Self);
  itkTypeMacro(Mesh, DataObject);
  static const unsigned int PointDimension = VDimension;
  typedef TMeshTraits MeshTraits;
  typedef typename MeshTraits::PixelType PixelType;
  typedef typename MeshTraits::CellTraits CellTraits;
  typedef typename MeshTraits::CoordRepType CoordRepType;
  typedef typename MeshTraits::PointIdentifier PointIdentifier;
  typedef typename MeshTraits::CellIdentifier CellIdentifier;
  typedef typename MeshTraits::PointType PointType;
  typedef CellInterface<typename MeshTraits::CellPixelType, CellTraits> CellType;
  typedef typename CellType::CellAutoPointer CellAutoPointer;
  typedef typename MeshTraits::PointsContainer PointsContainer;
  typedef typename PointsContainer::Pointer PointsContainerPointer;
  typedef typename MeshTraits::CellsContainer CellsContainer;
  typedef typename CellsContainer::Pointer CellsContainerPointer;

  PointsContainer *GetPoints() { if (m_Points.IsNull()) m_Points = PointsContainer::New(); return m_Points; }
  CellsContainer *GetCells() { if (m_Cells.IsNull()) m_Cells = CellsContainer::New(); return m_Cells; }
  unsigned long GetNumberOfPoints() const { return m_Points.IsNull() ? 0 : m_Points->Size(); }
  unsigned long GetNumberOfCells() const { return m_Cells.IsNull() ? 0 : m_Cells->Size(); }
  void SetPoint(PointIdentifier id, const PointType &p) { GetPoints()->InsertElement(id, p); }
  bool GetPoint(PointIdentifier id, PointType *p) const { if (m_Points.IsNull() || id >= m_Points->Size()) return false; *p = m_Points->GetElement(id); return true; }   // (ids are dense here: 0 .. Size() - 1)
  void SetCells(CellsContainer *cells) { if (m_Cells.GetPointer() != cells) { ReleaseCellsMemory(); m_Cells = cells; this->Modified(); } }
  // How the cell objects were allocated, hence who frees them (names and meaning of itk::Mesh::ReleaseCellsMemory):
  //   CellsAllocatedDynamicallyCellByCell  each cell came from its own `new` and the mesh deletes it (txx:310-313);
  //   CellsAllocatedAsStaticArray          the cells live in storage that is not the mesh's to free: it only forgets the
  //                                        pointers.  (The MI355X filter's bulk fill keeps that storage in an object it
  //                                        hangs into the mesh's MetaDataDictionary, so it still lives and dies with the mesh.)
  //   CellsAllocatedAsADynamicArray        one `new[]` whose first cell is the array: the mesh delete[]s it.
  enum CellsAllocationMethodType { CellsAllocationMethodUndefined, CellsAllocatedAsStaticArray,
                                   CellsAllocatedAsADynamicArray, CellsAllocatedDynamicallyCellByCell };
  void SetCellsAllocationMethod(CellsAllocationMethodType m) { m_CellsAllocationMethod = m; }
  CellsAllocationMethodType GetCellsAllocationMethod() const { return m_CellsAllocationMethod; }
  // the mesh takes over the cell object and deletes it later (ITK ownership rule, txx:310-313)
  void SetCell(CellIdentifier id, CellAutoPointer &cell) {
    CellsContainer *c = GetCells();
    if (m_CellsAllocationMethod == CellsAllocatedDynamicallyCellByCell && id < c->Size() && c->ElementAt(id)) delete c->ElementAt(id);
    c->InsertElement(id, cell.ReleaseOwnership());
  }
  bool GetCell(CellIdentifier id, CellAutoPointer &cell) const {
    if (m_Cells.IsNull() || id >= m_Cells->Size()) return false;
    cell.TakeNoOwnership(m_Cells->GetElement(id));
    return true;
  }
  virtual void Initialize() {
    ReleaseCellsMemory();
    m_CellsAllocationMethod = CellsAllocatedDynamicallyCellByCell;
    if (m_Points.IsNotNull()) m_Points->Initialize();
  }
protected:
  Mesh() : m_CellsAllocationMethod(CellsAllocatedDynamicallyCellByCell) {}
  ~Mesh() { ReleaseCellsMemory(); }
  void ReleaseCellsMemory() {
    if (m_Cells.IsNull()) return;
    if (m_CellsAllocationMethod == CellsAllocatedDynamicallyCellByCell)
      for (CellIdentifier i = 0; i < m_Cells->Size(); i++) delete m_Cells->ElementAt(i);
    else if (m_CellsAllocationMethod == CellsAllocatedAsADynamicArray && m_Cells->Size())
      delete[] m_Cells->ElementAt(0);
    m_Cells->Initialize();
  }
  PointsContainerPointer m_Points;
  CellsContainerPointer m_Cells;
  CellsAllocationMethodType m_CellsAllocationMethod;
};

// declared so that the driver's typedefs and includes resolve; never instantiated
// (CuberilleTest01.cxx:23-26 compiles those pipelines out)
template <class TPixel, unsigned int VDimension = 3, class TTraits = void> class QuadEdgeMesh;
template <class TIn, class TOut> class BinaryThresholdImageFilter;
template <class TIn, class TOut> class BinaryMask3DMeshSource;
template <class TImage> class ImageFileWriter;
template <class TImage, class TCoordRep = double, class TCoefficientType = double> class BSplineInterpolateImageFunction;
template <class TIn, class TOut, class TCriterion> class QuadEdgeMeshQuadricDecimation;
template <class TMesh> class NumberOfFacesCriterion;
template <class TImage> class ConstShapedNeighborhoodIterator;
template <class TImage, class TOperatorValue = float, class TOutputValue = float> class GradientImageFilter {
public:
  typedef GradientImageFilter Self;
  typedef SmartPointer<Self> Pointer;
  typedef CovariantVector<TOutputValue, TImage::ImageDimension> OutputPixelType;
  typedef Image<OutputPixelType, TImage::ImageDimension> OutputImageType;
};
template <class TImage> class GradientRecursiveGaussianImageFilter;
template <class TImage, class TCoordRep = double> class VectorLinearInterpolateImageFunction {
public:
  typedef VectorLinearInterpolateImageFunction Self;
  typedef SmartPointer<Self> Pointer;
};

// ------------------------------------------------------------------------------------------
// LinearInterpolateImageFunction (ITK 3.x N-d form: 2^N neighbour weighted sum in double)
// ------------------------------------------------------------------------------------------
template <class TInputImage, class TCoordRep = double> class LinearInterpolateImageFunction : public Object {
public:
  typedef LinearInterpolateImageFunction Self;
  typedef SmartPointer<Self> Pointer;
  itkNewMacro(Self);
  itkTypeMacro(LinearInterpolateImageFunction, InterpolateImageFunction);
  typedef TInputImage InputImageType;
  typedef double OutputType;
  typedef double RealType;
  typedef Point<TCoordRep, TInputImage::ImageDimension> PointType;
  typedef typename TInputImage::IndexType IndexType;
  void SetInputImage(const TInputImage *img) { m_Image = img; }
  const TInputImage *GetInputImage() const { return m_Image; }
  OutputType Evaluate(const PointType &point) const {
    const unsigned int N = TInputImage::ImageDimension;
    const typename TInputImage::RegionType &reg = m_Image->GetBufferedRegion();
    double ci[8];
    // identity-direction form of TransformPhysicalPointToContinuousIndex
    for (unsigned int k = 0; k < N; k++) ci[k] = (point[k] - m_Image->GetOrigin()[k]) / m_Image->GetSpacing()[k];
    long base[8];
    double dist[8];
    for (unsigned int k = 0; k < N; k++) { base[k] = (long)std::floor(ci[k]); dist[k] = ci[k] - (double)base[k]; }
    double value = 0.0, total = 0.0;
    for (unsigned int counter = 0; counter < (1u << N); counter++) {
      double overlap = 1.0;
      IndexType ni;
      for (unsigned int k = 0; k < N; k++) {
        long lo = reg.GetIndex()[k], hi = lo + (long)reg.GetSize()[k] - 1, v;
        if (counter & (1u << k)) { v = base[k] + 1; overlap *= dist[k]; } else { v = base[k]; overlap *= 1.0 - dist[k]; }
        ni[k] = v < lo ? lo : (v > hi ? hi : v);
      }
      if (overlap) { value += overlap * static_cast<double>(m_Image->GetPixel(ni)); total += overlap; }
      if (total == 1.0) break;
    }
    return value;
  }
protected:
  LinearInterpolateImageFunction() : m_Image(0) {}
  const TInputImage *m_Image;
};

// ------------------------------------------------------------------------------------------
// ImageToMeshFilter
// ------------------------------------------------------------------------------------------
template <class TOutputMesh> class MeshSource : public ProcessObject {
public:
  typedef MeshSource Self;
  typedef SmartPointer<Self> Pointer;
  typedef TOutputMesh OutputMeshType;
  typedef typename OutputMeshType::Pointer OutputMeshPointer;
  itkTypeMacro(MeshSource, ProcessObject);
  OutputMeshType *GetOutput() { return static_cast<OutputMeshType *>(this->m_Output.GetPointer()); }
protected:
  MeshSource() { OutputMeshPointer o = OutputMeshType::New(); this->SetPrimaryOutput(o.GetPointer()); }
};

template <class TInputImage, class TOutputMesh> class ImageToMeshFilter : public MeshSource<TOutputMesh> {
public:
  typedef ImageToMeshFilter Self;
  typedef MeshSource<TOutputMesh> Superclass;
  typedef SmartPointer<Self> Pointer;
  typedef SmartPointer<const Self> ConstPointer;
  itkTypeMacro(ImageToMeshFilter, MeshSource);
  typedef TInputImage InputImageType;
  typedef TOutputMesh OutputMeshType;
  void SetInput(unsigned int idx, const InputImageType *input) { this->ProcessObject::SetNthInput(idx, const_cast<InputImageType *>(input)); }
  const InputImageType *GetInput(unsigned int idx = 0) {
    if (this->m_Inputs.size() <= idx) return 0;
    return static_cast<const InputImageType *>(this->m_Inputs[idx].GetPointer());
  }
  OutputMeshType *GetOutput() { return Superclass::GetOutput(); }
protected:
  ImageToMeshFilter() { this->SetNumberOfRequiredInputs(1); }
  void PrintSelf(std::ostream &os, Indent indent) const { Superclass::PrintSelf(os, indent); }
};

// ------------------------------------------------------------------------------------------
// TimeProbe
// ------------------------------------------------------------------------------------------
class TimeProbe {
public:
  TimeProbe() : m_Total(0), m_Starts(0), m_Stops(0), m_T0(0) {}
  void Start() { m_T0 = Now(); m_Starts++; }
  void Stop() { m_Total += Now() - m_T0; m_Stops++; }
  double GetMeanTime() const { return m_Stops ? m_Total / m_Stops : 0.0; }
  double GetTotal() const { return m_Total; }
private:
  static double Now() { timeval tv; gettimeofday(&tv, 0); return tv.tv_sec + 1e-6 * tv.tv_usec; }
  double m_Total;
  unsigned long m_Starts, m_Stops;
  double m_T0;
};

}  // namespace itk

#define vnl_math_max(a, b) ((a) > (b) ? (a) : (b))
#define vnl_math_abs(a) ((a) < 0 ? -(a) : (a))

#endif
