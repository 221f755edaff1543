// ITK-lite forwarding header: carries the ITK file name, the declarations live in itkLite.h
#ifndef ITK_LITE_FWD_itkDefaultDynamicMeshTraits_H
#define ITK_LITE_FWD_itkDefaultDynamicMeshTraits_H
#include "itkLite.h"
#endif
