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

// splitmix64 finaliser: the counter-based generator behind the negative sampler and the dropout masks
__host__ __device__ inline uint64_t mix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

// Message dropout (F.dropout on a layer's output, /root/reference/model/lightgcn.py:56): whether element `index` of
// the tensor drawn under `seed` survives is a pure function of (seed, index), so the backward pass re-creates the
// forward's mask instead of storing it.  Four consecutive elements (one float4) share one hash: 4 x 16 bits.
struct DropMask {
  float p;          // drop probability; 0 = off
  uint64_t seed;
};
__device__ __forceinline__ void drop4(const DropMask& m, int64_t float4_index, float& a, float& b, float& c, float& d) {
  if (m.p <= 0.f) return;
  const uint64_t h = mix64(m.seed ^ mix64(static_cast<uint64_t>(float4_index)));
  const unsigned thr = static_cast<unsigned>(m.p * 65536.0f);          // keep iff 16-bit draw >= p * 2^16
  const float keep = 1.0f / (1.0f - m.p);
  a = ((h >> 0) & 0xFFFFu) >= thr ? a * keep : 0.f;
  b = ((h >> 16) & 0xFFFFu) >= thr ? b * keep : 0.f;
  c = ((h >> 32) & 0xFFFFu) >= thr ? c * keep : 0.f;
  d = ((h >> 48) & 0xFFFFu) >= thr ? d * keep : 0.f;
}


// torch.optim.Adam's update (_single_tensor_adam: exp_avg.lerp_(grad, 1 - b1); exp_avg_sq.mul_(b2).addcmul_(grad, grad,
// 1 - b2); param.addcdiv_(exp_avg, sqrt(exp_avg_sq) / sqrt(bc2) + eps, -step_size)), shared by the Adam kernels (rowops.hip)
// and the last backward hop's fused epilogue (spmm.hip).  The roundings are PINNED -- contraction off, the three fused
// multiply-adds written out -- so that every kernel that applies the update gives the same bits whatever surrounds the call
// (left to the compiler, the vector kernel and a scalar kernel contracted differently).
__device__ __forceinline__ float adam_update1(float& m, float& v, float p, const float g, float w1, float b2, float w2, float step_size,
                                              float bc2_sqrt, float eps) {
#pragma clang fp contract(off)
  m = fmaf(w1, g - m, m);
  v = fmaf(w2 * g, g, v * b2);
  const float denom = sqrtf(v) / bc2_sqrt + eps;
  return fmaf(-step_size, m / denom, p);
}
typedef float adam_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ adam_f4 adam_update(adam_f4& mi, adam_f4& vi, adam_f4 pi, const adam_f4 gi, float w1, float b2, float w2,
                                               float step_size, float bc2_sqrt, float eps) {
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    float m = mi[c], v = vi[c];
    pi[c] = adam_update1(m, v, pi[c], gi[c], w1, b2, w2, step_size, bc2_sqrt, eps);
    mi[c] = m;
    vi[c] = v;
  }
  return pi;
}
// Optional Adam update in place of a gradient store (EpiArgs::adam): p == nullptr -> off
struct AdamRow {
  float* p; float* m; float* v;
  float w1, b2, w2, step_size, bc2_sqrt, eps;
  const float* coef = nullptr;     // device [step_size, bc2_sqrt] advanced on the device (graph replay); overrides the two floats
};

}  // namespace tagrec
