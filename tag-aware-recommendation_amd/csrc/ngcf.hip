// K3: the dense half of an NGCF layer on the gfx950 matrix cores (exact-fp32 MFMA 16x16x4).
//
// Replaces, per layer k of /root/reference/model/ngcf.py:73-90 (after `nei = split_mm(A, X)`):
//     S  = LeakyReLU_0.2((nei + X) (W1 + b1));   Bi = LeakyReLU_0.2((nei * X) (W2 + b2))
//     X' = S + Bi;   Z = F.normalize(X')         (W + b: the 1 x Dout bias is added to the WEIGHT)
// and the autograd backward of that block.  W' = W + b is formed by the caller (tiny).
//
// Layout trick: every product is computed TRANSPOSED (P^T = W'^T A^T) so that the MFMA "column"
// index (lane & 15) is always the node row and the 4 accumulator registers are 4 consecutive
// features.  One register layout -- lane (r = lane&15, q = lane>>4) owns features {16 b + 4 q + v} of
// row r -- then serves as MFMA B-operand of the next product, as the float4 global load/store
// shape, and as the row-norm reduction shape (xor 16, 32); nothing is transposed through LDS.
//   forward      P^T  = W'^T . A^T      A-operand W'[k][m]  from LDS (k-major, stride Dout+4)
//   backward dA  dA^T = W'   . dP^T     A-operand W'[i][k]  from LDS (transposed copy, stride Din+4)
//   backward dW  dW'  = A^T  . dP       both operands straight from global (rows on the k axis)
// HBM-bound (about 2 GB per layer and pass at C3); the MFMA work is ~0.2 ms per product at C3.
//
// 128 -> 128 layers: both LDS copies of both matrices would take 270 KB, and the weight-gradient accumulators of both
// matrices 512 registers.  That shape runs the backward and the weight gradient ONE MATRIX PER LAUNCH (template MAT):
// the k-major and the transposed copy of one matrix fit (135 KB), pass 1 writes dN = dX_direct = dA1, pass 2 adds
// dA2 * X and dA2 * N to them; the forward keeps both k-major copies (135 KB), one block per CU.
#include "common.h"

namespace tagrec {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kNgcfThreads = 256;     // 4 waves, 16 rows each -> 64-row tiles
constexpr float kSlope = 0.2f;

__device__ __forceinline__ float lrelu(float x) { return x > 0.f ? x : kSlope * x; }
__device__ __forceinline__ float lrelu_grad(float x) { return x > 0.f ? 1.f : kSlope; }

template <int DIN, int DOUT>
struct NgcfShape {
  static constexpr bool kBig = DIN * DOUT >= 128 * 128;      // one matrix per launch in the backward kernels
  static constexpr int kOcc = kBig ? 1 : 2;                  // blocks per CU the register budget is set for
  static constexpr int IB = DIN / 16, MB = DOUT / 16;
  static constexpr int LDW = DOUT + 4;   // k-major W' rows: lanes run over the Dout index
  static constexpr int LDT = DIN + 4;    // transposed copy: lanes run over the Din index
};

// cooperative fill of the LDS copies of W' (row-major [DIN][DOUT] in global memory)
template <int DIN, int DOUT, bool WITH_T>
__device__ __forceinline__ void load_weights(const float* __restrict__ W1, const float* __restrict__ W2, float* lds) {
  using S = NgcfShape<DIN, DOUT>;
  float* w1 = lds;
  float* w2 = w1 + DIN * S::LDW;
  float* t1 = w2 + DIN * S::LDW;
  float* t2 = t1 + DOUT * S::LDT;
  for (int e = threadIdx.x; e < DIN * DOUT; e += kNgcfThreads) {
    const int k = e / DOUT, m = e % DOUT;
    const float a = W1[e], b = W2[e];
    w1[k * S::LDW + m] = a;
    w2[k * S::LDW + m] = b;
    if constexpr (WITH_T) {
      t1[m * S::LDT + k] = a;
      t2[m * S::LDT + k] = b;
    }
  }
  __syncthreads();
}

// one matrix: k-major copy at lds, transposed copy behind it
template <int DIN, int DOUT>
__device__ __forceinline__ void load_weights_one(const float* __restrict__ W, float* lds) {
  using S = NgcfShape<DIN, DOUT>;
  float* w = lds;
  float* t = w + DIN * S::LDW;
  for (int e = threadIdx.x; e < DIN * DOUT; e += kNgcfThreads) {
    const int k = e / DOUT, m = e % DOUT;
    const float a = W[e];
    w[k * S::LDW + m] = a;
    t[m * S::LDT + k] = a;
  }
  __syncthreads();
}

template <int DIN, int DOUT>
__device__ __forceinline__ void product_one(const float* w, const f32x4 (&a)[DIN / 16], f32x4 (&p)[DOUT / 16], int lane) {
  using S = NgcfShape<DIN, DOUT>;
  const int m = lane & 15, q = lane >> 4;
#pragma unroll
  for (int mb = 0; mb < S::MB; ++mb) p[mb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ib = 0; ib < S::IB; ++ib) {
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int k = ib * 16 + q * 4 + v;
#pragma unroll
      for (int mb = 0; mb < S::MB; ++mb)
        p[mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[k * S::LDW + mb * 16 + m], a[ib][v], p[mb], 0, 0, 0);
    }
  }
}

// P1^T, P2^T for the wave's 16 rows; a1/a2 are the lane's operand registers (see header comment)
template <int DIN, int DOUT>
__device__ __forceinline__ void product_pair(const float* lds, const f32x4 (&a1)[DIN / 16], const f32x4 (&a2)[DIN / 16],
                                             f32x4 (&p1)[DOUT / 16], f32x4 (&p2)[DOUT / 16], int lane) {
  using S = NgcfShape<DIN, DOUT>;
  const float* w1 = lds;
  const float* w2 = w1 + DIN * S::LDW;
  const int m = lane & 15, q = lane >> 4;
#pragma unroll
  for (int mb = 0; mb < S::MB; ++mb) { p1[mb] = f32x4{0.f, 0.f, 0.f, 0.f}; p2[mb] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
  for (int ib = 0; ib < S::IB; ++ib) {
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int k = ib * 16 + q * 4 + v;
#pragma unroll
      for (int mb = 0; mb < S::MB; ++mb) {
        p1[mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1[k * S::LDW + mb * 16 + m], a1[ib][v], p1[mb], 0, 0, 0);
        p2[mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(w2[k * S::LDW + mb * 16 + m], a2[ib][v], p2[mb], 0, 0, 0);
      }
    }
  }
}

template <int DIN>
__device__ __forceinline__ void load_inputs(const float* __restrict__ Nn, const float* __restrict__ X, int64_t row,
                                            bool ok, int q, f32x4 (&nn)[DIN / 16], f32x4 (&x)[DIN / 16]) {
#pragma unroll
  for (int ib = 0; ib < DIN / 16; ++ib) {
    if (ok) {
      nn[ib] = *reinterpret_cast<const f32x4*>(Nn + row * DIN + ib * 16 + q * 4);
      x[ib] = *reinterpret_cast<const f32x4*>(X + row * DIN + ib * 16 + q * 4);
    } else {
      nn[ib] = f32x4{0.f, 0.f, 0.f, 0.f};
      x[ib] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
}

// ---- forward -----------------------------------------------------------------------------------
template <int DIN, int DOUT>
__global__ __launch_bounds__(kNgcfThreads, (NgcfShape<DIN, DOUT>::kOcc)) void ngcf_fwd_kernel(const float* __restrict__ Nn, const float* __restrict__ X,
                                                                const float* __restrict__ W1, const float* __restrict__ W2,
                                                                int64_t n_rows, float* __restrict__ Xp,
                                                                float* __restrict__ inv_norm, float* __restrict__ Z,
                                                                int64_t ldz, const uint8_t* __restrict__ row_mask) {
  using S = NgcfShape<DIN, DOUT>;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  load_weights<DIN, DOUT, false>(W1, W2, lds);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int64_t n_tiles = (n_rows + 63) / 64;
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int64_t row = tile * 64 + wave * 16 + r;
    const bool ok = row < n_rows && (!row_mask || row_mask[row]);
    if (row_mask && !__any(ok)) continue;                    // none of the wave's 16 rows is wanted (wave-uniform)
    f32x4 nn[S::IB], x[S::IB], a1[S::IB], a2[S::IB], p1[S::MB], p2[S::MB];
    load_inputs<DIN>(Nn, X, row, ok, q, nn, x);
#pragma unroll
    for (int ib = 0; ib < S::IB; ++ib) { a1[ib] = nn[ib] + x[ib]; a2[ib] = nn[ib] * x[ib]; }
    product_pair<DIN, DOUT>(lds, a1, a2, p1, p2, lane);
    float ss = 0.f;
#pragma unroll
    for (int mb = 0; mb < S::MB; ++mb) {
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const float o = lrelu(p1[mb][v]) + lrelu(p2[mb][v]);
        p1[mb][v] = o;
        ss = fmaf(o, o, ss);
      }
    }
    ss += __shfl_xor(ss, 16);
    ss += __shfl_xor(ss, 32);
    const float den = fmaxf(sqrtf(ss), 1e-12f);
    if (ok) {
#pragma unroll
      for (int mb = 0; mb < S::MB; ++mb) {
        *reinterpret_cast<f32x4*>(Xp + row * DOUT + mb * 16 + q * 4) = p1[mb];
        if (Z) {
          f32x4 z = p1[mb];
          z[0] /= den; z[1] /= den; z[2] /= den; z[3] /= den;
          *reinterpret_cast<f32x4*>(Z + row * ldz + mb * 16 + q * 4) = z;
        }
      }
      if (q == 0) inv_norm[row] = 1.0f / den;
    }
  }
}

// ---- backward, activation part: dN, dX_direct and the pre-activation gradients dP1, dP2 ----------
// The gradient w.r.t. Xp is either given (dXp) or formed here (NORM): dXp = G + normalize-backward(Xp, inv, dZ), G
// (may be null) = what the NEXT layer sent back to this layer's output, dZ = this layer's slot of the concat
// gradient (row stride ldz).  Forming it here saves writing and re-reading an [n, Dout] tensor per layer.
struct NormGrad {
  const float* Xp;
  const float* inv;
  const float* dZ;
  int64_t ldz;
  const uint8_t* dz_flags;     // may be null; rows whose byte is 0 have dZ == 0: neither Xp nor dZ is read there
};

template <int DIN, int DOUT, bool NORM>
__global__ __launch_bounds__(kNgcfThreads) void ngcf_bwd_kernel(const float* __restrict__ dXp, NormGrad ng,
                                                                const float* __restrict__ Nn,
                                                                const float* __restrict__ X, const float* __restrict__ W1,
                                                                const float* __restrict__ W2, int64_t n_rows,
                                                                float* __restrict__ dNn, float* __restrict__ dXd,
                                                                float* __restrict__ dP1, float* __restrict__ dP2,
                                                                const uint8_t* __restrict__ row_mask) {
  using S = NgcfShape<DIN, DOUT>;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  load_weights<DIN, DOUT, true>(W1, W2, lds);
  const float* t1 = lds + 2 * DIN * S::LDW;
  const float* t2 = t1 + DOUT * S::LDT;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int64_t n_tiles = (n_rows + 63) / 64;
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int64_t row = tile * 64 + wave * 16 + r;
    const bool ok = row < n_rows && (!row_mask || row_mask[row]);
    if (row_mask && !__any(ok)) continue;                    // rows outside the mask: nothing read, nothing written
    f32x4 nn[S::IB], x[S::IB], a1[S::IB], a2[S::IB], p1[S::MB], p2[S::MB];
    load_inputs<DIN>(Nn, X, row, ok, q, nn, x);
#pragma unroll
    for (int ib = 0; ib < S::IB; ++ib) { a1[ib] = nn[ib] + x[ib]; a2[ib] = nn[ib] * x[ib]; }
    product_pair<DIN, DOUT>(lds, a1, a2, p1, p2, lane);      // recompute the pre-activations
    f32x4 gx[S::MB];
#pragma unroll
    for (int mb = 0; mb < S::MB; ++mb) {
      gx[mb] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (ok && dXp) gx[mb] = *reinterpret_cast<const f32x4*>(dXp + row * DOUT + mb * 16 + q * 4);
    }
    if constexpr (NORM) {
      f32x4 z[S::MB], dz[S::MB];
      const bool nz = ok && (!ng.dz_flags || ng.dz_flags[row]);
      const float inv = nz ? ng.inv[row] : 0.f;
      float dot = 0.f;
#pragma unroll
      for (int mb = 0; mb < S::MB; ++mb) {
        z[mb] = f32x4{0.f, 0.f, 0.f, 0.f};
        dz[mb] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (nz) {
          z[mb] = *reinterpret_cast<const f32x4*>(ng.Xp + row * DOUT + mb * 16 + q * 4) * inv;
          dz[mb] = *reinterpret_cast<const f32x4*>(ng.dZ + row * ng.ldz + mb * 16 + q * 4);
        }
#pragma unroll
        for (int v = 0; v < 4; ++v) dot = fmaf(z[mb][v], dz[mb][v], dot);
      }
      dot += __shfl_xor(dot, 16);
      dot += __shfl_xor(dot, 32);
      if (inv >= 1e12f) dot = 0.f;                         // the clamp is constant where ||Xp|| <= eps
#pragma unroll
      for (int mb = 0; mb < S::MB; ++mb)
#pragma unroll
        for (int v = 0; v < 4; ++v) gx[mb][v] += inv * (dz[mb][v] - z[mb][v] * dot);
    }
#pragma unroll
    for (int mb = 0; mb < S::MB; ++mb) {
      const f32x4 g = gx[mb];
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        p1[mb][v] = g[v] * lrelu_grad(p1[mb][v]);
        p2[mb][v] = g[v] * lrelu_grad(p2[mb][v]);
      }
      if (ok) {
        *reinterpret_cast<f32x4*>(dP1 + row * DOUT + mb * 16 + q * 4) = p1[mb];
        *reinterpret_cast<f32x4*>(dP2 + row * DOUT + mb * 16 + q * 4) = p2[mb];
      }
    }
    // dA^T = W' . dP^T : contraction over the Dout index, operand = the registers just formed
#pragma unroll
    for (int ib = 0; ib < S::IB; ++ib) {
      f32x4 d1 = f32x4{0.f, 0.f, 0.f, 0.f}, d2 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int mb = 0; mb < S::MB; ++mb) {
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int k = mb * 16 + q * 4 + v;
          d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(t1[k * S::LDT + ib * 16 + r], p1[mb][v], d1, 0, 0, 0);
          d2 = __builtin_amdgcn_mfma_f32_16x16x4f32(t2[k * S::LDT + ib * 16 + r], p2[mb][v], d2, 0, 0, 0);
        }
      }
      if (ok) {
        *reinterpret_cast<f32x4*>(dNn + row * DIN + ib * 16 + q * 4) = d1 + d2 * x[ib];
        *reinterpret_cast<f32x4*>(dXd + row * DIN + ib * 16 + q * 4) = d1 + d2 * nn[ib];
      }
    }
  }
}

// ---- backward, activation part, ONE matrix per launch (128 -> 128: see the header) ----------------------------------
// MAT = 1: dP1, dN = dX_direct = dA1.   MAT = 2: dP2, dN += dA2 * X, dX_direct += dA2 * N (read-modify-write).
template <int DIN, int DOUT, bool NORM, int MAT>
__global__ __launch_bounds__(kNgcfThreads, 1) void ngcf_bwd_one_kernel(const float* __restrict__ dXp, NormGrad ng,
                                                                    const float* __restrict__ Nn, const float* __restrict__ X,
                                                                    const float* __restrict__ W, int64_t n_rows,
                                                                    float* __restrict__ dNn, float* __restrict__ dXd,
                                                                    float* __restrict__ dP, const uint8_t* __restrict__ row_mask) {
  using S = NgcfShape<DIN, DOUT>;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  load_weights_one<DIN, DOUT>(W, lds);
  const float* tw = lds + DIN * S::LDW;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int64_t n_tiles = (n_rows + 63) / 64;
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int64_t row = tile * 64 + wave * 16 + r;
    const bool ok = row < n_rows && (!row_mask || row_mask[row]);
    if (row_mask && !__any(ok)) continue;
    f32x4 nn[S::IB], x[S::IB], a[S::IB], p[S::MB];
    load_inputs<DIN>(Nn, X, row, ok, q, nn, x);
#pragma unroll
    for (int ib = 0; ib < S::IB; ++ib) a[ib] = MAT == 1 ? nn[ib] + x[ib] : nn[ib] * x[ib];
    product_one<DIN, DOUT>(lds, a, p, lane);                 // recompute this matrix's pre-activations
    f32x4 gx[S::MB];
#pragma unroll
    for (int mb = 0; mb < S::MB; ++mb) {
      gx[mb] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (ok && dXp) gx[mb] = *reinterpret_cast<const f32x4*>(dXp + row * DOUT + mb * 16 + q * 4);
    }
    if constexpr (NORM) {
      const bool nz = ok && (!ng.dz_flags || ng.dz_flags[row]);
      const float inv = nz ? ng.inv[row] : 0.f;
      float dot = 0.f;
#pragma unroll
      for (int mb = 0; mb < S::MB; ++mb) {
        if (nz) {
          const f32x4 z = *reinterpret_cast<const f32x4*>(ng.Xp + row * DOUT + mb * 16 + q * 4) * inv;
          const f32x4 dz = *reinterpret_cast<const f32x4*>(ng.dZ + row * ng.ldz + mb * 16 + q * 4);
#pragma unroll
          for (int v = 0; v < 4; ++v) dot = fmaf(z[v], dz[v], dot);
        }
      }
      dot += __shfl_xor(dot, 16);
      dot += __shfl_xor(dot, 32);
      if (inv >= 1e12f) dot = 0.f;
#pragma unroll
      for (int mb = 0; mb < S::MB; ++mb) {
        if (nz) {                                              // (re-read: holding z and dz would cost 64 registers)
          const f32x4 z = *reinterpret_cast<const f32x4*>(ng.Xp + row * DOUT + mb * 16 + q * 4) * inv;
          const f32x4 dz = *reinterpret_cast<const f32x4*>(ng.dZ + row * ng.ldz + mb * 16 + q * 4);
#pragma unroll
          for (int v = 0; v < 4; ++v) gx[mb][v] += inv * (dz[v] - z[v] * dot);
        }
      }
    }
#pragma unroll
    for (int mb = 0; mb < S::MB; ++mb) {
#pragma unroll
      for (int v = 0; v < 4; ++v) p[mb][v] = gx[mb][v] * lrelu_grad(p[mb][v]);
      if (ok) *reinterpret_cast<f32x4*>(dP + row * DOUT + mb * 16 + q * 4) = p[mb];
    }
#pragma unroll
    for (int ib = 0; ib < S::IB; ++ib) {
      f32x4 d = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int mb = 0; mb < S::MB; ++mb) {
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int k = mb * 16 + q * 4 + v;
          d = __builtin_amdgcn_mfma_f32_16x16x4f32(tw[k * S::LDT + ib * 16 + r], p[mb][v], d, 0, 0, 0);
        }
      }
      if (ok) {
        f32x4* pn = reinterpret_cast<f32x4*>(dNn + row * DIN + ib * 16 + q * 4);
        f32x4* px = reinterpret_cast<f32x4*>(dXd + row * DIN + ib * 16 + q * 4);
        if constexpr (MAT == 1) {
          *pn = d;
          *px = d;
        } else {
          *pn = *pn + d * x[ib];
          *px = *px + d * nn[ib];
        }
      }
    }
  }
}

// weight gradient of ONE matrix (128 -> 128: 256 accumulator registers per matrix)
template <int DIN, int DOUT, int MAT>
__global__ __launch_bounds__(kNgcfThreads, 1) void ngcf_wgrad_one_kernel(const float* __restrict__ Nn, const float* __restrict__ X,
                                                                      const float* __restrict__ dP, int64_t n_rows,
                                                                      int64_t steps_per_wave, float* __restrict__ slab,
                                                                      const uint8_t* __restrict__ row_mask) {
  constexpr int IB = DIN / 16, JB = DOUT / 16;
  const int lane = threadIdx.x & 63;
  const int m = lane & 15, q = lane >> 4;
  const int64_t w = static_cast<int64_t>(blockIdx.x) * (kNgcfThreads / 64) + (threadIdx.x >> 6);
  f32x4 acc[IB][JB];
#pragma unroll
  for (int i = 0; i < IB; ++i)
#pragma unroll
    for (int j = 0; j < JB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int64_t s0 = w * steps_per_wave;
  for (int64_t s = s0; s < s0 + steps_per_wave; ++s) {
    const int64_t row = s * 4 + q;
    if (s * 4 >= n_rows) break;
    const bool ok = row < n_rows && (!row_mask || row_mask[row]);
    if (row_mask && !__any(ok)) continue;                    // rows outside the mask contribute nothing
    float a[IB], b[JB];
#pragma unroll
    for (int i = 0; i < IB; ++i) {
      const float nv = ok ? Nn[row * DIN + i * 16 + m] : 0.f;
      const float xv = ok ? X[row * DIN + i * 16 + m] : 0.f;
      a[i] = MAT == 1 ? nv + xv : nv * xv;
    }
#pragma unroll
    for (int j = 0; j < JB; ++j) b[j] = ok ? dP[row * DOUT + j * 16 + m] : 0.f;
#pragma unroll
    for (int i = 0; i < IB; ++i)
#pragma unroll
      for (int j = 0; j < JB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
  }
  float* o = slab + w * 2 * DIN * DOUT + (MAT == 1 ? 0 : DIN * DOUT);
#pragma unroll
  for (int i = 0; i < IB; ++i)
#pragma unroll
    for (int j = 0; j < JB; ++j)
#pragma unroll
      for (int v = 0; v < 4; ++v) o[(i * 16 + q * 4 + v) * DOUT + j * 16 + m] = acc[i][j][v];
}

// ---- backward, weight part: dW1' = (N+X)^T dP1, dW2' = (N*X)^T dP2 -------------------------------
// Each wave owns a contiguous strip of rows and a full set of DIN x DOUT accumulators; partial sums
// go to a slab [wave][2][DIN*DOUT] that a second kernel folds in wave order (deterministic).
template <int DIN, int DOUT>
__global__ __launch_bounds__(kNgcfThreads, 2) void ngcf_wgrad_kernel(const float* __restrict__ Nn, const float* __restrict__ X,
                                                                  const float* __restrict__ dP1, const float* __restrict__ dP2,
                                                                  int64_t n_rows, int64_t steps_per_wave,
                                                                  float* __restrict__ slab, const uint8_t* __restrict__ row_mask) {
  constexpr int IB = DIN / 16, JB = DOUT / 16;
  const int lane = threadIdx.x & 63;
  const int m = lane & 15, q = lane >> 4;
  const int64_t w = static_cast<int64_t>(blockIdx.x) * (kNgcfThreads / 64) + (threadIdx.x >> 6);
  f32x4 acc1[IB][JB], acc2[IB][JB];
#pragma unroll
  for (int i = 0; i < IB; ++i)
#pragma unroll
    for (int j = 0; j < JB; ++j) { acc1[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; acc2[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  const int64_t s0 = w * steps_per_wave;
  for (int64_t s = s0; s < s0 + steps_per_wave; ++s) {
    const int64_t row = s * 4 + q;
    if (s * 4 >= n_rows) break;
    const bool ok = row < n_rows && (!row_mask || row_mask[row]);
    if (row_mask && !__any(ok)) continue;                    // rows outside the mask contribute nothing
    float a1[IB], a2[IB], b1[JB], b2[JB];
#pragma unroll
    for (int i = 0; i < IB; ++i) {
      const float nv = ok ? Nn[row * DIN + i * 16 + m] : 0.f;
      const float xv = ok ? X[row * DIN + i * 16 + m] : 0.f;
      a1[i] = nv + xv;
      a2[i] = nv * xv;
    }
#pragma unroll
    for (int j = 0; j < JB; ++j) {
      b1[j] = ok ? dP1[row * DOUT + j * 16 + m] : 0.f;
      b2[j] = ok ? dP2[row * DOUT + j * 16 + m] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < IB; ++i)
#pragma unroll
      for (int j = 0; j < JB; ++j) {
        acc1[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[i], b1[j], acc1[i][j], 0, 0, 0);
        acc2[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[i], b2[j], acc2[i][j], 0, 0, 0);
      }
  }
  float* o1 = slab + w * 2 * DIN * DOUT;
  float* o2 = o1 + DIN * DOUT;
#pragma unroll
  for (int i = 0; i < IB; ++i)
#pragma unroll
    for (int j = 0; j < JB; ++j)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        o1[(i * 16 + q * 4 + v) * DOUT + j * 16 + m] = acc1[i][j][v];
        o2[(i * 16 + q * 4 + v) * DOUT + j * 16 + m] = acc2[i][j][v];
      }
}

// fold the per-wave partials: 16 threads per element take every 16th wave, then a fixed-order combine ->
// deterministic, and 16 independent load streams per element (at C3 the slab is 64 MB per layer)
constexpr int kReduceParts = 16;
__global__ __launch_bounds__(64 * kReduceParts) void ngcf_wgrad_reduce_kernel(const float* __restrict__ slab, int n_waves, int elems,
                                                                             float* __restrict__ dW1, float* __restrict__ dW2) {
  __shared__ float sh[kReduceParts][64];
  const int e = blockIdx.x * 64 + (threadIdx.x & 63);
  const int part = threadIdx.x >> 6;
  float s = 0.f;
  if (e < 2 * elems)
    for (int w = part; w < n_waves; w += kReduceParts) s += slab[static_cast<int64_t>(w) * 2 * elems + e];
  sh[part][threadIdx.x & 63] = s;
  __syncthreads();
  if (part == 0 && e < 2 * elems) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < kReduceParts; ++k) t += sh[k][threadIdx.x];
    if (e < elems) dW1[e] = t; else dW2[e - elems] = t;
  }
}

constexpr int kWgradBlocks = 512;   // 2 blocks per CU, 2048 waves
constexpr int kWgradWaves = kWgradBlocks * (kNgcfThreads / 64);

template <int DIN, int DOUT>
int launch_fwd(const float* Nn, const float* X, const float* W1, const float* W2, int64_t n, float* Xp, float* inv,
               float* Z, int64_t ldz, const uint8_t* row_mask, hipStream_t s) {
  using S = NgcfShape<DIN, DOUT>;
  const size_t lds = sizeof(float) * 2 * DIN * S::LDW;
  const int64_t tiles = (n + 63) / 64;
  const unsigned grid = static_cast<unsigned>(tiles < 1024 ? tiles : 1024);
  if (lds > 64 * 1024)
    TAGREC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ngcf_fwd_kernel<DIN, DOUT>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
  ngcf_fwd_kernel<DIN, DOUT><<<grid, kNgcfThreads, lds, s>>>(Nn, X, W1, W2, n, Xp, inv, Z, ldz, row_mask);
  TAGREC_LAUNCH_CHECK();
  return TAGREC_OK;
}

template <int DIN, int DOUT>
int launch_bwd(const float* dXp, const NormGrad& ng, const float* Nn, const float* X, const float* W1, const float* W2, int64_t n,
               float* dNn, float* dXd, float* dP1, float* dP2, const uint8_t* row_mask, hipStream_t s) {
  using S = NgcfShape<DIN, DOUT>;
  const int64_t tiles = (n + 63) / 64;
  if constexpr (S::kBig) {
    // one matrix per launch: k-major + transposed copy of ONE matrix in LDS (see the header)
    const size_t lds1 = sizeof(float) * (DIN * S::LDW + DOUT * S::LDT);
    const unsigned grid1 = static_cast<unsigned>(tiles < 256 ? tiles : 256);
    auto k1 = ng.Xp ? ngcf_bwd_one_kernel<DIN, DOUT, true, 1> : ngcf_bwd_one_kernel<DIN, DOUT, false, 1>;
    auto k2 = ng.Xp ? ngcf_bwd_one_kernel<DIN, DOUT, true, 2> : ngcf_bwd_one_kernel<DIN, DOUT, false, 2>;
    TAGREC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k1), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds1)));
    TAGREC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k2), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds1)));
    k1<<<grid1, kNgcfThreads, lds1, s>>>(dXp, ng, Nn, X, W1, n, dNn, dXd, dP1, row_mask);
    TAGREC_LAUNCH_CHECK();
    k2<<<grid1, kNgcfThreads, lds1, s>>>(dXp, ng, Nn, X, W2, n, dNn, dXd, dP2, row_mask);
    TAGREC_LAUNCH_CHECK();
    return TAGREC_OK;
  } else {
    const size_t lds = sizeof(float) * (2 * DIN * S::LDW + 2 * DOUT * S::LDT);
    const unsigned grid = static_cast<unsigned>(tiles < 1024 ? tiles : 1024);
    auto kern = ng.Xp ? ngcf_bwd_kernel<DIN, DOUT, true> : ngcf_bwd_kernel<DIN, DOUT, false>;
    if (lds > 64 * 1024)
      TAGREC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     static_cast<int>(lds)));
    kern<<<grid, kNgcfThreads, lds, s>>>(dXp, ng, Nn, X, W1, W2, n, dNn, dXd, dP1, dP2, row_mask);
    TAGREC_LAUNCH_CHECK();
    return TAGREC_OK;
  }
}

template <int DIN, int DOUT>
int launch_wgrad(const float* Nn, const float* X, const float* dP1, const float* dP2, int64_t n, float* dW1, float* dW2,
                 float* ws, const uint8_t* row_mask, hipStream_t s) {
  const int64_t steps = (n + 3) / 4;
  int64_t per = (steps + kWgradWaves - 1) / kWgradWaves;
  if (per < 8) per = 8;                 // small graphs: fewer waves, so the fold below walks fewer partials
  constexpr int kWavesPerBlock = kNgcfThreads / 64;
  const int64_t waves = steps > 0 ? (steps + per - 1) / per : 1;      // n == 0: one block writes zero partials
  const unsigned blocks = static_cast<unsigned>((waves + kWavesPerBlock - 1) / kWavesPerBlock);
  if constexpr (NgcfShape<DIN, DOUT>::kBig) {
    ngcf_wgrad_one_kernel<DIN, DOUT, 1><<<blocks, kNgcfThreads, 0, s>>>(Nn, X, dP1, n, per, ws, row_mask);
    TAGREC_LAUNCH_CHECK();
    ngcf_wgrad_one_kernel<DIN, DOUT, 2><<<blocks, kNgcfThreads, 0, s>>>(Nn, X, dP2, n, per, ws, row_mask);
  } else {
    ngcf_wgrad_kernel<DIN, DOUT><<<blocks, kNgcfThreads, 0, s>>>(Nn, X, dP1, dP2, n, per, ws, row_mask);
  }
  TAGREC_LAUNCH_CHECK();
  const int elems = DIN * DOUT;
  ngcf_wgrad_reduce_kernel<<<(2 * elems + 63) / 64, 64 * kReduceParts, 0, s>>>(ws, static_cast<int>(blocks) * kWavesPerBlock, elems, dW1, dW2);
  TAGREC_LAUNCH_CHECK();
  return TAGREC_OK;
}

#define TAGREC_NGCF_DISPATCH(CALL)                                                          \
  switch (Din * 1000 + Dout) {                                                              \
    case 16016: return CALL(16, 16);   case 16032: return CALL(16, 32);                     \
    case 16064: return CALL(16, 64);   case 16128: return CALL(16, 128);                    \
    case 32016: return CALL(32, 16);   case 32032: return CALL(32, 32);                     \
    case 32064: return CALL(32, 64);   case 32128: return CALL(32, 128);                    \
    case 64016: return CALL(64, 16);   case 64032: return CALL(64, 32);                     \
    case 64064: return CALL(64, 64);   case 64128: return CALL(64, 128);                    \
    case 128016: return CALL(128, 16); case 128032: return CALL(128, 32);                   \
    case 128064: return CALL(128, 64); case 128128: return CALL(128, 128);                  \
    default: break;                                                                         \
  }                                                                                         \
  return fail(TAGREC_E_UNSUPPORTED, "ngcf: layer widths must be 16, 32, 64 or 128 (got " +  \
                                        std::to_string(Din) + " -> " + std::to_string(Dout) + ")")

}  // namespace tagrec

using namespace tagrec;

extern "C" int64_t tagrec_ngcf_wgrad_workspace(int Din, int Dout) {
  return static_cast<int64_t>(kWgradWaves) * 2 * Din * Dout;
}

extern "C" int tagrec_ngcf_dense_fwd_rows_f32(const float* Nn, const float* X, const float* W1p, const float* W2p,
                                              int64_t n_rows, int Din, int Dout, float* Xp, float* inv_norm, float* Z,
                                              int64_t ldz, const uint8_t* row_mask, void* stream) {
  TAGREC_REQUIRE(Nn && X && W1p && W2p && Xp && inv_norm, "ngcf_dense_fwd: null pointer");
  TAGREC_REQUIRE(n_rows >= 0 && (!Z || (ldz >= Dout && (ldz % 4) == 0)), "ngcf_dense_fwd: bad shape (ldz must be a multiple of 4)");
  TAGREC_REQUIRE(aligned16(Nn) && aligned16(X) && aligned16(Xp) && aligned16(Z), "ngcf_dense_fwd: rows must be 16-byte aligned");
  if (n_rows == 0) return TAGREC_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
#define CALL(A, B) launch_fwd<A, B>(Nn, X, W1p, W2p, n_rows, Xp, inv_norm, Z, ldz, row_mask, s)
  TAGREC_NGCF_DISPATCH(CALL);
#undef CALL
}

extern "C" int tagrec_ngcf_dense_fwd_f32(const float* Nn, const float* X, const float* W1p, const float* W2p,
                                         int64_t n_rows, int Din, int Dout, float* Xp, float* inv_norm, float* Z,
                                         int64_t ldz, void* stream) {
  TAGREC_REQUIRE(Z, "ngcf_dense_fwd: null pointer");
  return tagrec_ngcf_dense_fwd_rows_f32(Nn, X, W1p, W2p, n_rows, Din, Dout, Xp, inv_norm, Z, ldz, nullptr, stream);
}

extern "C" int tagrec_ngcf_dense_bwd_f32(const float* dXp, const float* Nn, const float* X, const float* W1p,
                                         const float* W2p, int64_t n_rows, int Din, int Dout, float* dNn, float* dXd,
                                         float* dP1, float* dP2, void* stream) {
  TAGREC_REQUIRE(dXp && Nn && X && W1p && W2p && dNn && dXd && dP1 && dP2, "ngcf_dense_bwd: null pointer");
  TAGREC_REQUIRE(n_rows >= 0, "ngcf_dense_bwd: bad shape");
  TAGREC_REQUIRE(aligned16(dXp) && aligned16(Nn) && aligned16(X) && aligned16(dNn) && aligned16(dXd) && aligned16(dP1) &&
                     aligned16(dP2), "ngcf_dense_bwd: rows must be 16-byte aligned");
  if (n_rows == 0) return TAGREC_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const NormGrad ng{nullptr, nullptr, nullptr, 0, nullptr};
#define CALL(A, B) launch_bwd<A, B>(dXp, ng, Nn, X, W1p, W2p, n_rows, dNn, dXd, dP1, dP2, nullptr, s)
  TAGREC_NGCF_DISPATCH(CALL);
#undef CALL
}

extern "C" int tagrec_ngcf_dense_bwd_rows_f32(const float* G, const float* Xp, const float* inv_norm, const float* dZ, int64_t ldz,
                                              const uint8_t* dz_flags, const float* Nn, const float* X, const float* W1p,
                                              const float* W2p, int64_t n_rows, int Din, int Dout, float* dNn, float* dXd,
                                              float* dP1, float* dP2, const uint8_t* row_mask, void* stream) {
  TAGREC_REQUIRE(Xp && inv_norm && dZ && Nn && X && W1p && W2p && dNn && dXd && dP1 && dP2, "ngcf_dense_bwd_norm: null pointer");
  TAGREC_REQUIRE(n_rows >= 0 && ldz >= Dout && ldz % 4 == 0, "ngcf_dense_bwd_norm: bad shape");
  TAGREC_REQUIRE(aligned16(G) && aligned16(Xp) && aligned16(dZ) && aligned16(Nn) && aligned16(X) && aligned16(dNn) &&
                     aligned16(dXd) && aligned16(dP1) && aligned16(dP2), "ngcf_dense_bwd_norm: rows must be 16-byte aligned");
  if (n_rows == 0) return TAGREC_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const NormGrad ng{Xp, inv_norm, dZ, ldz, dz_flags};
#define CALL(A, B) launch_bwd<A, B>(G, ng, Nn, X, W1p, W2p, n_rows, dNn, dXd, dP1, dP2, row_mask, s)
  TAGREC_NGCF_DISPATCH(CALL);
#undef CALL
}

extern "C" int tagrec_ngcf_dense_bwd_norm_f32(const float* G, const float* Xp, const float* inv_norm, const float* dZ, int64_t ldz,
                                              const float* Nn, const float* X, const float* W1p, const float* W2p,
                                              int64_t n_rows, int Din, int Dout, float* dNn, float* dXd, float* dP1,
                                              float* dP2, void* stream) {
  return tagrec_ngcf_dense_bwd_rows_f32(G, Xp, inv_norm, dZ, ldz, nullptr, Nn, X, W1p, W2p, n_rows, Din, Dout, dNn, dXd, dP1, dP2,
                                        nullptr, stream);
}

extern "C" int tagrec_ngcf_wgrad_rows_f32(const float* Nn, const float* X, const float* dP1, const float* dP2, int64_t n_rows,
                                          int Din, int Dout, float* dW1p, float* dW2p, float* workspace,
                                          int64_t workspace_floats, const uint8_t* row_mask, void* stream) {
  TAGREC_REQUIRE(Nn && X && dP1 && dP2 && dW1p && dW2p && workspace, "ngcf_wgrad: null pointer");
  TAGREC_REQUIRE(workspace_floats >= tagrec_ngcf_wgrad_workspace(Din, Dout), "ngcf_wgrad: workspace too small");
  hipStream_t s = static_cast<hipStream_t>(stream);
#define CALL(A, B) launch_wgrad<A, B>(Nn, X, dP1, dP2, n_rows, dW1p, dW2p, workspace, row_mask, s)
  TAGREC_NGCF_DISPATCH(CALL);
#undef CALL
}

extern "C" int tagrec_ngcf_wgrad_f32(const float* Nn, const float* X, const float* dP1, const float* dP2, int64_t n_rows,
                                     int Din, int Dout, float* dW1p, float* dW2p, float* workspace,
                                     int64_t workspace_floats, void* stream) {
  return tagrec_ngcf_wgrad_rows_f32(Nn, X, dP1, dP2, n_rows, Din, Dout, dW1p, dW2p, workspace, workspace_floats, nullptr, stream);
}
