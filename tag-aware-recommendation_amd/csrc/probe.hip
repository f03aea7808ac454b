// Bandwidth probes ("re-measure achievable copy bandwidth on the box ... and report fraction of both").
// Two minimal kernels that bench.py runs before its timed region, so that the roofline line carries MEASURED ceilings of
// the machine it ran on next to the 8 TB/s specification:
//   * stream triad a = b + s c over large arrays (the HBM streaming ceiling), 16 B per lane, grid-stride;
//   * random whole-row gather: one wavefront fetches 64 rows of D floats named by an index list -- the access shape of
//     the SpMM's neighbour gather (csrc/spmm.hip: D/4 lanes x 16 B per row, 64 / (D/4) rows per wave-instruction) with
//     everything else stripped: no values, no epilogue, indices read coalesced, one store per wave at the end.  From a
//     table that fits the 256 MB Infinity Cache it measures the cache-resident gather ceiling, from a multi-GB table
//     the HBM gather ceiling.
#include "common.h"

namespace tagrec {
namespace {

typedef float pf4 __attribute__((ext_vector_type(4)));

// NT = non-temporal loads / stores (lines not retained in the caches); c == nullptr: plain copy a = b (8 bytes per element)
template <bool NT>
__global__ __launch_bounds__(256) void probe_triad_kernel(pf4* __restrict__ a, const pf4* __restrict__ b,
                                                           const pf4* __restrict__ c, float s, int64_t n4) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * 256;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; i < n4; i += stride) {
    pf4 x = NT ? __builtin_nontemporal_load(b + i) : b[i];
    if (c) x += s * (NT ? __builtin_nontemporal_load(c + i) : c[i]);
    if (NT) __builtin_nontemporal_store(x, a + i); else a[i] = x;
  }
}

// read-only stream: every thread sums its float4s and stores one value at the end (the SpMM moves 50x more bytes in than out)
__global__ __launch_bounds__(256) void probe_read_kernel(const pf4* __restrict__ b, int64_t n4, pf4* __restrict__ out) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * 256;
  pf4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
  int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  for (; i + stride < n4; i += 2 * stride) {
    acc0 += __builtin_nontemporal_load(b + i);
    acc1 += __builtin_nontemporal_load(b + i + stride);
  }
  if (i < n4) acc0 += __builtin_nontemporal_load(b + i);
  out[static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x] = acc0 + acc1;
}

// LPR = lanes per row (D / 4).  A wave walks chunks of 64 indices: every lane loads one index, the indices are handed
// round by ds_bpermute, and each of the 64 / (64 / LPR) gather instructions of a chunk fetches 64 / LPR whole rows.
template <int LPR>
__global__ __launch_bounds__(256) void probe_gather_kernel(const pf4* __restrict__ table, const int* __restrict__ idx,
                                                            int64_t n_chunks, pf4* __restrict__ out) {
  constexpr int RPI = 64 / LPR;
  const int lane = threadIdx.x & 63;
  const int64_t wave = (static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x) >> 6;
  const int64_t n_waves = static_cast<int64_t>(gridDim.x) * 4;
  const int sub = lane / LPR, c = lane % LPR;
  pf4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int64_t ch = wave; ch < n_chunks; ch += n_waves) {
    const int mine = idx[ch * 64 + lane];
#pragma unroll
    for (int j = 0; j < 64 / RPI; ++j) {
      const int r = __builtin_amdgcn_ds_bpermute((j * RPI + sub) * 4, mine);
      acc += table[static_cast<int64_t>(r) * LPR + c];
    }
  }
  out[wave * 64 + lane] = acc;
}


// Shader clock while other work runs: one wavefront reads the shader-cycle counter (s_memtime) and the constant 100 MHz
// counter (s_memrealtime) at both ends of a timed spin.  Launched on a second stream BEFORE the kernel under study, it
// shares the chip's clock with it: cycles / ticks x 100 MHz = the frequency the kernel actually ran at (the peaks of
// MI355X_MICROARCH.md are quoted at 2.4 GHz).  Ends by the wall clock, whatever else happens.
__global__ void probe_clock_kernel(long long spin_ticks, long long* __restrict__ out) {
  if (threadIdx.x != 0) return;
  const long long t0 = wall_clock64();
  const long long c0 = clock64();
  long long t1 = t0;
  while (t1 - t0 < spin_ticks) {
    __builtin_amdgcn_s_sleep(32);
    t1 = wall_clock64();
  }
  const long long c1 = clock64();
  out[0] = c1 - c0;
  out[1] = t1 - t0;
}

}  // namespace
}  // namespace tagrec

using namespace tagrec;

extern "C" int tagrec_probe_triad_f32(float* a, const float* b, const float* c, float s, int64_t n, int non_temporal, void* stream) {
  TAGREC_REQUIRE(a && b, "probe_triad: null pointer");
  TAGREC_REQUIRE(n >= 0 && n % 4 == 0 && aligned16(a) && aligned16(b) && (!c || aligned16(c)), "probe_triad: need a multiple of 4 elements, 16-byte aligned");
  if (n == 0) return TAGREC_OK;
  int64_t blocks = (n / 4 + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (non_temporal)
    probe_triad_kernel<true><<<static_cast<unsigned>(blocks), 256, 0, st>>>(reinterpret_cast<pf4*>(a), reinterpret_cast<const pf4*>(b),
                                                                           reinterpret_cast<const pf4*>(c), s, n / 4);
  else
    probe_triad_kernel<false><<<static_cast<unsigned>(blocks), 256, 0, st>>>(reinterpret_cast<pf4*>(a), reinterpret_cast<const pf4*>(b),
                                                                            reinterpret_cast<const pf4*>(c), s, n / 4);
  TAGREC_LAUNCH_CHECK();
  return TAGREC_OK;
}

extern "C" int tagrec_probe_read_f32(const float* b, int64_t n, float* out, void* stream) {
  TAGREC_REQUIRE(b && out, "probe_read: null pointer");
  TAGREC_REQUIRE(n >= 0 && n % 4 == 0 && aligned16(b) && aligned16(out), "probe_read: need a multiple of 4 elements, 16-byte aligned");
  if (n == 0) return TAGREC_OK;
  probe_read_kernel<<<256 * 8, 256, 0, static_cast<hipStream_t>(stream)>>>(reinterpret_cast<const pf4*>(b), n / 4, reinterpret_cast<pf4*>(out));
  TAGREC_LAUNCH_CHECK();
  return TAGREC_OK;
}

extern "C" int64_t tagrec_probe_gather_out_floats(void) { return static_cast<int64_t>(256) * 8 * 4 * 64 * 4; }

extern "C" int tagrec_probe_gather_rows_f32(const float* table, int64_t n_rows, int D, const int32_t* idx, int64_t n_idx, float* out,
                                            void* stream) {
  TAGREC_REQUIRE(table && idx && out, "probe_gather: null pointer");
  TAGREC_REQUIRE(n_rows >= 1 && n_rows <= 0x7fffffff && n_idx >= 0 && n_idx % 64 == 0, "probe_gather: n_idx must be a multiple of 64");
  TAGREC_REQUIRE(aligned16(table) && aligned16(out), "probe_gather: 16-byte aligned buffers expected");
  if (n_idx == 0) return TAGREC_OK;
  const unsigned blocks = 256 * 8;                      // out holds blocks * 4 waves * 64 lanes * 4 floats
  hipStream_t s = static_cast<hipStream_t>(stream);
  const pf4* t = reinterpret_cast<const pf4*>(table);
  pf4* o = reinterpret_cast<pf4*>(out);
  switch (D) {
    case 32: probe_gather_kernel<8><<<blocks, 256, 0, s>>>(t, idx, n_idx / 64, o); break;
    case 64: probe_gather_kernel<16><<<blocks, 256, 0, s>>>(t, idx, n_idx / 64, o); break;
    case 128: probe_gather_kernel<32><<<blocks, 256, 0, s>>>(t, idx, n_idx / 64, o); break;
    case 256: probe_gather_kernel<64><<<blocks, 256, 0, s>>>(t, idx, n_idx / 64, o); break;
    default: return fail(TAGREC_E_UNSUPPORTED, "probe_gather: D must be 32, 64, 128 or 256");
  }
  TAGREC_LAUNCH_CHECK();
  return TAGREC_OK;
}

extern "C" int tagrec_probe_clock(int64_t spin_us, int64_t* out2, void* stream) {
  TAGREC_REQUIRE(out2 && spin_us >= 1 && spin_us <= 1000000, "probe_clock: 1 us .. 1 s");
  probe_clock_kernel<<<1, 64, 0, static_cast<hipStream_t>(stream)>>>(spin_us * 100, reinterpret_cast<long long*>(out2));
  TAGREC_LAUNCH_CHECK();
  return TAGREC_OK;
}
