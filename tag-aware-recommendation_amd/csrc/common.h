// Shared host-side helpers of libtagrec_hip (error slot, HIP call checking).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>

#include "../../include/tagrec.h"

namespace tagrec {

void set_error(const std::string& msg);

inline int fail(int code, const std::string& msg) {
  set_error(msg);
  return code;
}

#define TAGREC_HIP(expr)                                                                   \
  do {                                                                                     \
    hipError_t _e = (expr);                                                                \
    if (_e != hipSuccess)                                                                  \
      return ::tagrec::fail(TAGREC_E_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
  } while (0)

#define TAGREC_REQUIRE(cond, msg)                                        \
  do {                                                                   \
    if (!(cond)) return ::tagrec::fail(TAGREC_E_INVALID, std::string(msg)); \
  } while (0)

#define TAGREC_LAUNCH_CHECK() TAGREC_HIP(hipGetLastError())

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

constexpr int kWave = 64;

}  // namespace tagrec
