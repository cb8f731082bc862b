// api.hip -- error reporting, version and device query of libfgs_hip.so.
#include "fgs_common.h"

#include <stdarg.h>
#include <string.h>

namespace {
thread_local char g_last_error[512] = "";
}

int fgs_set_error(int code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_last_error, sizeof(g_last_error), fmt, ap);
  va_end(ap);
  return code;
}

FGS_API const char *fgs_last_error(void) { return g_last_error; }

FGS_API int fgs_version(void) { return FGS_ABI_VERSION; }

FGS_API int fgs_device_info(int device, char *name, int name_len, int *cu_count, int *wave_size, int64_t *lds_bytes) {
  hipDeviceProp_t prop;
  hipError_t e = hipGetDeviceProperties(&prop, device);
  if (e != hipSuccess) return fgs_set_error(FGS_E_NODEV, "fgs_device_info: %s", hipGetErrorString(e));
  if (name && name_len > 0) {
    snprintf(name, (size_t)name_len, "%s (%s)", prop.name, prop.gcnArchName);
  }
  if (cu_count) *cu_count = prop.multiProcessorCount;
  if (wave_size) *wave_size = prop.warpSize;
  if (lds_bytes) *lds_bytes = (int64_t)prop.maxSharedMemoryPerMultiProcessor;
  return 0;
}
