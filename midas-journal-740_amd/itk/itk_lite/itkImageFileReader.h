// ITK-lite: itk::ImageFileReader restricted to MetaImage (.mha/.mhd-with-LOCAL-data) files, the
// only format the reference's tests use (Testing/CuberilleTest01.cxx:113-117; Data/*.mha are
// zlib-compressed MET_UCHAR volumes).  Needs -lz.
#ifndef ITK_LITE_IMAGE_FILE_READER_H
#define ITK_LITE_IMAGE_FILE_READER_H
#include "itkLite.h"
#include <zlib.h>

namespace itk {

namespace lite {
template <class T> struct MetaTypeName;
template <> struct MetaTypeName<unsigned char> { static const char *Name() { return "MET_UCHAR"; } };
template <> struct MetaTypeName<signed char> { static const char *Name() { return "MET_CHAR"; } };
template <> struct MetaTypeName<char> { static const char *Name() { return "MET_CHAR"; } };
template <> struct MetaTypeName<unsigned short> { static const char *Name() { return "MET_USHORT"; } };
template <> struct MetaTypeName<short> { static const char *Name() { return "MET_SHORT"; } };
template <> struct MetaTypeName<unsigned int> { static const char *Name() { return "MET_UINT"; } };
template <> struct MetaTypeName<int> { static const char *Name() { return "MET_INT"; } };
template <> struct MetaTypeName<float> { static const char *Name() { return "MET_FLOAT"; } };
template <> struct MetaTypeName<double> { static const char *Name() { return "MET_DOUBLE"; } };

inline size_t MetaTypeSize(const std::string &t) {
  if (t == "MET_UCHAR" || t == "MET_CHAR") return 1;
  if (t == "MET_USHORT" || t == "MET_SHORT") return 2;
  if (t == "MET_UINT" || t == "MET_INT" || t == "MET_FLOAT") return 4;
  if (t == "MET_DOUBLE") return 8;
  return 0;
}
template <class TOut> void ConvertBuffer(const std::string &t, const unsigned char *src, size_t n, TOut *dst) {
#define ITK_LITE_CONV(NAME, TYPE) \
  if (t == NAME) { const TYPE *s = reinterpret_cast<const TYPE *>(src); for (size_t i = 0; i < n; i++) dst[i] = static_cast<TOut>(s[i]); return; }
  ITK_LITE_CONV("MET_UCHAR", unsigned char) ITK_LITE_CONV("MET_CHAR", signed char)
  ITK_LITE_CONV("MET_USHORT", unsigned short) ITK_LITE_CONV("MET_SHORT", short)
  ITK_LITE_CONV("MET_UINT", unsigned int) ITK_LITE_CONV("MET_INT", int)
  ITK_LITE_CONV("MET_FLOAT", float) ITK_LITE_CONV("MET_DOUBLE", double)
#undef ITK_LITE_CONV
}
}  // namespace lite

template <class TOutputImage> class ImageFileReader : public ProcessObject {
public:
  typedef ImageFileReader Self;
  typedef SmartPointer<Self> Pointer;
  itkNewMacro(Self);
  itkTypeMacro(ImageFileReader, ImageSource);
  typedef TOutputImage OutputImageType;
  typedef typename TOutputImage::PixelType PixelType;
  void SetFileName(const char *name) { m_FileName = name; this->Modified(); }
  void SetFileName(const std::string &name) { m_FileName = name; this->Modified(); }
  const char *GetFileName() const { return m_FileName.c_str(); }
  TOutputImage *GetOutput() { return static_cast<TOutputImage *>(this->m_Output.GetPointer()); }

protected:
  ImageFileReader() { typename TOutputImage::Pointer o = TOutputImage::New(); this->SetPrimaryOutput(o.GetPointer()); }

  virtual void GenerateData() {
    const unsigned int N = TOutputImage::ImageDimension;
    std::ifstream f(m_FileName.c_str(), std::ios::binary);
    if (!f) itkExceptionMacro(<< "Could not open file: " << m_FileName);
    std::map<std::string, std::string> h;
    std::string line;
    while (std::getline(f, line)) {
      const size_t eq = line.find('=');
      if (eq == std::string::npos) continue;
      std::string key = Trim(line.substr(0, eq)), val = Trim(line.substr(eq + 1));
      h[key] = val;
      if (key == "ElementDataFile") break;
    }
    if (h["ElementDataFile"] != "LOCAL") itkExceptionMacro(<< "Only ElementDataFile = LOCAL is supported: " << m_FileName);
    if (h.count("NDims") && (unsigned int)std::atoi(h["NDims"].c_str()) != N)
      itkExceptionMacro(<< "NDims mismatch in " << m_FileName);
    if (h.count("ElementNumberOfChannels") && std::atoi(h["ElementNumberOfChannels"].c_str()) != 1)
      itkExceptionMacro(<< "Only scalar pixels are supported: " << m_FileName);
    const std::string et = h["ElementType"];
    const size_t esz = lite::MetaTypeSize(et);
    if (!esz) itkExceptionMacro(<< "Unsupported ElementType '" << et << "' in " << m_FileName);
    if (Lower(h["BinaryDataByteOrderMSB"]) == "true" || Lower(h["ElementByteOrderMSB"]) == "true")
      itkExceptionMacro(<< "Big-endian MetaImage data is not supported: " << m_FileName);
    typename TOutputImage::RegionType region;
    typename TOutputImage::SpacingType spacing;
    typename TOutputImage::PointType origin;
    typename TOutputImage::DirectionType dir;
    spacing.Fill(1.0);
    origin.Fill(0.0);
    {
      std::istringstream s(h["DimSize"]);
      for (unsigned int i = 0; i < N; i++) { unsigned long v = 0; s >> v; region.m_Size[i] = v; region.m_Index[i] = 0; }
    }
    if (h.count("ElementSpacing")) { std::istringstream s(h["ElementSpacing"]); for (unsigned int i = 0; i < N; i++) s >> spacing[i]; }
    const std::string okey = h.count("Offset") ? "Offset" : (h.count("Position") ? "Position" : "");
    if (!okey.empty()) { std::istringstream s(h[okey]); for (unsigned int i = 0; i < N; i++) s >> origin[i]; }
    const std::string tkey = h.count("TransformMatrix") ? "TransformMatrix" : (h.count("Orientation") ? "Orientation" : "");
    if (!tkey.empty()) {   // MetaIO stores the direction cosines column-wise
      std::istringstream s(h[tkey]);
      for (unsigned int c = 0; c < N; c++) for (unsigned int r = 0; r < N; r++) s >> dir[r][c];
    }
    const size_t npix = region.GetNumberOfPixels();
    std::vector<unsigned char> payload((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    std::vector<unsigned char> raw;
    if (Lower(h["CompressedData"]) == "true") {
      size_t csize = payload.size();
      if (h.count("CompressedDataSize")) csize = std::min(csize, (size_t)std::strtoull(h["CompressedDataSize"].c_str(), 0, 10));
      raw.resize(npix * esz);
      uLongf dlen = (uLongf)raw.size();
      const int rc = uncompress(raw.empty() ? 0 : &raw[0], &dlen, payload.empty() ? 0 : &payload[0], (uLong)csize);
      if (rc != Z_OK || dlen != raw.size()) itkExceptionMacro(<< "zlib could not inflate the pixel data of " << m_FileName);
    } else {
      if (payload.size() < npix * esz) itkExceptionMacro(<< "Pixel data truncated in " << m_FileName);
      raw.swap(payload);
    }
    TOutputImage *out = GetOutput();
    out->SetRegions(region);
    out->SetSpacing(spacing);
    out->SetOrigin(origin);
    out->SetDirection(dir);
    out->Allocate();
    lite::ConvertBuffer(et, raw.empty() ? 0 : &raw[0], npix, out->GetBufferPointer());
  }

private:
  static std::string Trim(const std::string &s) {
    const size_t a = s.find_first_not_of(" \t\r\n"), b = s.find_last_not_of(" \t\r\n");
    return a == std::string::npos ? std::string() : s.substr(a, b - a + 1);
  }
  static std::string Lower(std::string s) { for (size_t i = 0; i < s.size(); i++) s[i] = (char)std::tolower(s[i]); return s; }
  std::string m_FileName;
};

}  // namespace itk
#endif
