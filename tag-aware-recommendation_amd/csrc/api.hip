// Library-wide entry points: ABI version, per-thread error slot, device probe.
#include <cstring>

#include "common.h"

namespace tagrec {
namespace {
thread_local std::string g_last_error;
}
void set_error(const std::string& msg) { g_last_error = msg; }
}  // namespace tagrec

extern "C" int tagrec_abi_version(void) { return TAGREC_ABI_VERSION; }

extern "C" const char* tagrec_last_error(void) { return tagrec::g_last_error.c_str(); }

extern "C" int tagrec_device_info(int* n_cu, int* wave_size, char* arch, int arch_len) {
  int dev = 0;
  TAGREC_HIP(hipGetDevice(&dev));
  hipDeviceProp_t prop;
  TAGREC_HIP(hipGetDeviceProperties(&prop, dev));
  if (n_cu) *n_cu = prop.multiProcessorCount;
  if (wave_size) *wave_size = prop.warpSize;
  if (arch && arch_len > 0) {
    std::strncpy(arch, prop.gcnArchName, static_cast<size_t>(arch_len) - 1);
    arch[arch_len - 1] = '\0';
  }
  return TAGREC_OK;
}
