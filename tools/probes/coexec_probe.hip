// Do VALU instructions hide under MFMAs on gfx950?  One block per CU, W waves per SIMD; every wave runs ITER rounds of
// 8 independent v_mfma_f32_16x16x4_f32 (4 accumulators) with V independent v_fma_f32 per MFMA in between.
//   hipcc --offload-arch=gfx950 -O3 -o coexec_probe coexec_probe.hip && ./coexec_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int V>
__global__ __launch_bounds__(1024) void probe(float* out, int iters) {
  f32x4 acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float a = threadIdx.x * 1e-3f, b = 1.0f + blockIdx.x * 1e-6f;
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = a + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      acc[m & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[m & 3], 0, 0, 0);
#pragma unroll
      for (int k = 0; k < V; ++k) v[(m + k) & 7] = __builtin_fmaf(v[(m + k) & 7], 1.0001f, 0.5f);
    }
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += v[i];
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 123.456f) out[0] = s;
}

template <int V>
float run(int waves_per_simd, int iters) {
  float* d;
  hipMalloc(&d, 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int threads = 64 * 4 * waves_per_simd;
  probe<V><<<256, threads>>>(d, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  probe<V><<<256, threads>>>(d, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  hipFree(d);
  return ms;
}

int main() {
  const int iters = 20000;
  for (int w = 1; w <= 4; w *= 2) {
    const double mf = 8.0 * iters * w;          // MFMAs per SIMD
    float t0 = run<0>(w, iters), t1 = run<1>(w, iters), t2 = run<2>(w, iters), t4 = run<4>(w, iters), t6 = run<6>(w, iters);
    printf("waves/SIMD %d: cycles per MFMA per SIMD at 2.4 GHz: V=0 %.1f  V=1 %.1f  V=2 %.1f  V=4 %.1f  V=6 %.1f\n", w,
           t0 * 2.4e6 / mf, t1 * 2.4e6 / mf, t2 * 2.4e6 / mf, t4 * 2.4e6 / mf, t6 * 2.4e6 / mf);
  }
  return 0;
}
