// Version / error-string entry points of libdiffuvolume_hip.so.
#include "dv_common.h"

extern "C" int dv_version(void) { return 100; }  // 0.1.0

extern "C" const char* dv_error_string(int code) {
  switch (code) {
    case DV_OK: return "ok";
    case DV_ERR_NULL: return "required pointer is NULL";
    case DV_ERR_SHAPE: return "bad or inconsistent dimensions";
    case DV_ERR_UNSUPPORTED: return "shape/option not implemented by the gfx950 kernels";
    case DV_ERR_ALIGN: return "pointer not 16-byte aligned";
    default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown diffuvolume error";
  }
}
