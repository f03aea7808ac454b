// K7: TGCN neighbour-level attention over fixed-width neighbour tables, forward and backward.
//
// Replaces `Attention1.forward` (/root/reference/model/tgcn.py:20-37) after the algebraic split
//     [ev || eNw] W1 + eNj W2 + b  =  (ev W1[:D] + b)  +  (ew' W1[D:])[widx]  +  (ej W2)[idx]
//                                  =        P[v]       +       WT[widx]       +     Q[idx]
// P, Q, WT are small dense products formed by the caller (plain GEMMs); what is left is the part that
// touches k neighbours per node: score -> softmax over k (pad index 0 takes part as a zero row and is
// NOT masked, exactly as the reference) -> weighted sum of the gathered D-wide neighbour rows.
//
// One wavefront per (node, relation).  Two lane layouts, both row-contiguous:
//   * scores: A lanes per neighbour (A = attention width, a power of two <= 64), 64/A neighbours per
//     pass -> the Q / WT rows are read as contiguous 4A-byte segments, and in backward the dQ scatter is
//     a contiguous float-atomic segment per row (the fast shape on gfx950);
//   * embeddings: D/4 lanes x 16 B per neighbour row, 64/(D/4) rows per wave-instruction (as the SpMM),
//     for the weighted sum and for d a_n = dOut . e_n; the dEj scatter uses one lane per column so every
//     atomic wave-instruction covers 256 contiguous bytes.
// HBM-bound: about k (4D + 4A + 8) bytes per (node, relation) forward, three times that backward.
// The tiny dWT / dv tables are accumulated per block in LDS and folded by a second kernel in block order.
#include "common.h"

namespace tagrec {

constexpr int kAttnThreads = 256;
constexpr int kAttnWaves = kAttnThreads / kWave;
constexpr int kAttnBlocks = 2048;    // persistent blocks of the backward kernel

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int m = 1; m < kWave; m <<= 1) v = fmaxf(v, __shfl_xor(v, m));
  return v;
}
__device__ __forceinline__ float wave_add(float v) {
#pragma unroll
  for (int m = 1; m < kWave; m <<= 1) v += __shfl_xor(v, m);
  return v;
}

// softmax weights of node v's k neighbours; on return lane n < k holds a_n (0 elsewhere)
__device__ __forceinline__ float attention_weights(const float* __restrict__ p, const float* __restrict__ Q,
                                                   const float* __restrict__ WT, const float* __restrict__ vv, int j, int w,
                                                   int k, int A, int lane) {
  const int npi = kWave / A, grp = lane / A, c = lane % A;
  const float pc = p[c], vc = vv[c];
  float s = -INFINITY;
  for (int n0 = 0; n0 < k; n0 += npi) {
    const int n = n0 + grp;
    const int jn = __shfl(j, n & (kWave - 1));
    const int wn = __shfl(w, n & (kWave - 1));
    float t = 0.f;
    if (n < k) {
      float h = pc + WT[static_cast<int64_t>(wn) * A + c];
      if (jn) h += Q[static_cast<int64_t>(jn - 1) * A + c];
      t = fmaxf(h, 0.f) * vc;
    }
    for (int m = 1; m < A; m <<= 1) t += __shfl_xor(t, m);
    for (int gg = 0; gg < npi; ++gg) {
      const float tg = __shfl(t, gg * A);
      if (lane == n0 + gg && lane < k) s = tg;
    }
  }
  const float mx = wave_max(s);
  const float e = lane < k ? expf(s - mx) : 0.f;
  return e / wave_add(e);
}

template <int LPR>
__global__ __launch_bounds__(kAttnThreads) void tgcn_attn_fwd_kernel(const float* __restrict__ P, const float* __restrict__ Q,
                                                                     const float* __restrict__ WT, const float* __restrict__ vv,
                                                                     const float* __restrict__ Ej, const int32_t* __restrict__ idx,
                                                                     const int32_t* __restrict__ widx, int64_t n, int k, int A,
                                                                     float* __restrict__ attn, float* __restrict__ out) {
  constexpr int RPI = kWave / LPR;          // neighbour rows per wave-instruction
  const int lane = threadIdx.x & (kWave - 1);
  const int64_t v = static_cast<int64_t>(blockIdx.x) * kAttnWaves + (threadIdx.x >> 6);
  if (v >= n) return;
  int j = 0, w = 0;
  if (lane < k) {
    j = idx[v * k + lane];
    w = widx[v * k + lane];
  }
  const float a = attention_weights(P + v * A, Q, WT, vv, j, w, k, A, lane);
  if (lane < k) attn[v * k + lane] = a;
  const int g = lane / LPR, c4 = lane % LPR;
  const float4* __restrict__ E4 = reinterpret_cast<const float4*>(Ej) + c4;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int n0 = 0; n0 < k; n0 += RPI) {
    const int nn = n0 + g;
    const int jj = __shfl(j, nn & (kWave - 1));
    const float aa = __shfl(a, nn & (kWave - 1));
    if (nn < k && jj != 0) {
      const float4 x = E4[static_cast<int64_t>(jj - 1) * LPR];
      acc.x = fmaf(aa, x.x, acc.x); acc.y = fmaf(aa, x.y, acc.y); acc.z = fmaf(aa, x.z, acc.z); acc.w = fmaf(aa, x.w, acc.w);
    }
  }
#pragma unroll
  for (int m = LPR; m < kWave; m <<= 1) {
    acc.x += __shfl_xor(acc.x, m); acc.y += __shfl_xor(acc.y, m); acc.z += __shfl_xor(acc.z, m); acc.w += __shfl_xor(acc.w, m);
  }
  if (lane < LPR) reinterpret_cast<float4*>(out)[v * LPR + lane] = acc;
}

// PULL = false: dQ / dEj are scattered with float atomics.  PULL = true: the per-(node, neighbour) pre-activation
// gradients are written to dh [n, k, A] instead and the caller finishes dQ and dEj as two pull products over the
// inverted neighbour table with the SpMM kernel (no atomics, deterministic, about 3x faster at C4).
template <int LPR, bool PULL>
__global__ __launch_bounds__(kAttnThreads) void tgcn_attn_bwd_kernel(const float* __restrict__ P, const float* __restrict__ Q,
                                                                     const float* __restrict__ WT, const float* __restrict__ vv,
                                                                     const float* __restrict__ Ej, const int32_t* __restrict__ idx,
                                                                     const int32_t* __restrict__ widx, const float* __restrict__ attn,
                                                                     const float* __restrict__ dOut, int64_t n, int k, int A, int n_wt,
                                                                     float* __restrict__ dP, float* __restrict__ dQ,
                                                                     float* __restrict__ dEj, float* __restrict__ dh,
                                                                     float* __restrict__ part) {
  constexpr int RPI = kWave / LPR;
  constexpr int D = LPR * 4;
  extern __shared__ float sh[];                 // [n_wt * A] dWT partial, then [A] dv partial
  float* sh_wt = sh;
  float* sh_v = sh + n_wt * A;
  for (int i = threadIdx.x; i < (n_wt + 1) * A; i += kAttnThreads) sh[i] = 0.f;
  __syncthreads();
  const int lane = threadIdx.x & (kWave - 1);
  const int g = lane / LPR, c4 = lane % LPR;
  const int npi = kWave / A, grp = lane / A, c = lane % A;
  const float vc = vv[c];
  const float4* __restrict__ E4 = reinterpret_cast<const float4*>(Ej) + c4;
  float dv_acc = 0.f;                            // this lane's share of dv[c], over all its nodes
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kAttnWaves;
  for (int64_t v = static_cast<int64_t>(blockIdx.x) * kAttnWaves + (threadIdx.x >> 6); v < n; v += stride) {
    int j = 0, w = 0;
    float a = 0.f;
    if (lane < k) {
      j = idx[v * k + lane];
      w = widx[v * k + lane];
      a = attn[v * k + lane];
    }
    // (1) d a_n = dOut . e_n  (grouped float4 gather)
    const float4 go = reinterpret_cast<const float4*>(dOut)[v * LPR + c4];
    float da = 0.f;
    for (int n0 = 0; n0 < k; n0 += RPI) {
      const int nn = n0 + g;
      const int jj = __shfl(j, nn & (kWave - 1));
      float d = 0.f;
      if (nn < k && jj != 0) {
        const float4 x = E4[static_cast<int64_t>(jj - 1) * LPR];
        d = fmaf(go.x, x.x, fmaf(go.y, x.y, fmaf(go.z, x.z, go.w * x.w)));
      }
#pragma unroll
      for (int m = 1; m < LPR; m <<= 1) d += __shfl_xor(d, m);
#pragma unroll
      for (int gg = 0; gg < RPI; ++gg) {
        const float dg = __shfl(d, gg * LPR);
        if (lane == n0 + gg) da = dg;
      }
    }
    // (2) dEj[idx_n] += a_n dOut: one lane per column, 256 contiguous bytes per atomic instruction
    if constexpr (!PULL)
    for (int nn = 0; nn < k; ++nn) {
      const int jj = __shfl(j, nn);
      const float aa = __shfl(a, nn);
      if (jj != 0)
        for (int col = lane; col < D; col += kWave)
          atomicAdd(&dEj[static_cast<int64_t>(jj - 1) * D + col], aa * dOut[v * D + col]);
    }
    // (3) softmax backward; ds = 0 on lanes >= k because a = 0 there
    const float dot = wave_add(a * da);
    const float ds = a * (da - dot);
    // (4) pre-activation gradients in the A-lanes-per-neighbour layout
    const float pc = P[v * A + c];
    float dp_acc = 0.f;
    for (int n0 = 0; n0 < k; n0 += npi) {
      const int nn = n0 + grp;
      const int jn = __shfl(j, nn & (kWave - 1));
      const int wn = __shfl(w, nn & (kWave - 1));
      const float dsn = __shfl(ds, nn & (kWave - 1));
      if (nn < k) {
        float h = pc + WT[static_cast<int64_t>(wn) * A + c];
        if (jn) h += Q[static_cast<int64_t>(jn - 1) * A + c];
        float dhv = 0.f;
        if (h > 0.f) {
          dhv = dsn * vc;
          dp_acc += dhv;
          dv_acc = fmaf(dsn, h, dv_acc);
          atomicAdd(&sh_wt[wn * A + c], dhv);
          if constexpr (!PULL) {
            if (jn) atomicAdd(&dQ[static_cast<int64_t>(jn - 1) * A + c], dhv);
          }
        }
        if constexpr (PULL) dh[(v * k + nn) * A + c] = dhv;
      }
    }
    for (int m = A; m < kWave; m <<= 1) dp_acc += __shfl_xor(dp_acc, m);
    if (lane < A) dP[v * A + lane] = dp_acc;
  }
  for (int m = A; m < kWave; m <<= 1) dv_acc += __shfl_xor(dv_acc, m);
  if (lane < A) atomicAdd(&sh_v[lane], dv_acc);
  __syncthreads();
  float* o = part + static_cast<int64_t>(blockIdx.x) * (n_wt + 1) * A;
  for (int i = threadIdx.x; i < (n_wt + 1) * A; i += kAttnThreads) o[i] = sh[i];
}

// fold the per-block partial tables: 16 threads per element take every 16th block, then a fixed-order combine
// (deterministic given the partials; a single thread per element would walk up to 2048 dependent loads)
__global__ __launch_bounds__(256) void tgcn_attn_fold_kernel(const float* __restrict__ part, int n_blocks, int elems, int split,
                                                              float* __restrict__ dWT, float* __restrict__ dv) {
  __shared__ float sh[16][16];
  const int el = threadIdx.x & 15, p = threadIdx.x >> 4;
  const int e = blockIdx.x * 16 + el;
  float s = 0.f;
  if (e < elems)
    for (int b = p; b < n_blocks; b += 16) s += part[static_cast<int64_t>(b) * elems + e];
  sh[p][el] = s;
  __syncthreads();
  if (p == 0 && e < elems) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += sh[k][el];
    if (e < split) dWT[e] = t; else dv[e - split] = t;
  }
}

}  // namespace tagrec

using namespace tagrec;

namespace {
bool attn_shape_ok(int k, int D, int A) {
  const bool d_ok = D == 16 || D == 32 || D == 64 || D == 128 || D == 256;
  const bool a_ok = A == 4 || A == 8 || A == 16 || A == 32 || A == 64;
  return d_ok && a_ok && k >= 1 && k <= kWave;
}
const char* kShapeMsg = "tgcn_attn: need D in {16,32,64,128,256}, A in {4,8,16,32,64}, 1 <= k <= 64";
}  // namespace

extern "C" int64_t tagrec_tgcn_attn_workspace(int n_wt, int A) {
  return static_cast<int64_t>(kAttnBlocks) * (n_wt + 1) * A;
}

extern "C" int tagrec_tgcn_attn_fwd_f32(const float* P, const float* Q, const float* WT, const float* v, const float* Ej,
                                        const int32_t* idx, const int32_t* widx, int64_t n, int k, int D, int A,
                                        float* attn, float* out, void* stream) {
  TAGREC_REQUIRE(P && Q && WT && v && Ej && idx && widx && attn && out, "tgcn_attn_fwd: null pointer");
  if (!attn_shape_ok(k, D, A)) return fail(TAGREC_E_UNSUPPORTED, kShapeMsg);
  TAGREC_REQUIRE(aligned16(Ej) && aligned16(out), "tgcn_attn_fwd: embedding rows must be 16-byte aligned");
  if (n <= 0) return TAGREC_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const unsigned blocks = static_cast<unsigned>((n + kAttnWaves - 1) / kAttnWaves);
#define LAUNCH(L) tgcn_attn_fwd_kernel<L><<<blocks, kAttnThreads, 0, s>>>(P, Q, WT, v, Ej, idx, widx, n, k, A, attn, out)
  switch (D) {
    case 16: LAUNCH(4); break;
    case 32: LAUNCH(8); break;
    case 64: LAUNCH(16); break;
    case 128: LAUNCH(32); break;
    default: LAUNCH(64); break;
  }
#undef LAUNCH
  TAGREC_LAUNCH_CHECK();
  return TAGREC_OK;
}

extern "C" int tagrec_tgcn_attn_bwd_f32(const float* P, const float* Q, const float* WT, const float* v, const float* Ej,
                                        const int32_t* idx, const int32_t* widx, const float* attn, const float* dOut,
                                        int64_t n, int k, int D, int A, int n_wt, float* dP, float* dQ, float* dEj,
                                        float* dh, float* dWT, float* dv, float* workspace, int64_t workspace_floats,
                                        void* stream) {
  TAGREC_REQUIRE(P && Q && WT && v && Ej && idx && widx && attn && dOut && dP && dWT && dv && workspace,
                 "tgcn_attn_bwd: null pointer");
  TAGREC_REQUIRE(dh || (dQ && dEj), "tgcn_attn_bwd: give either dh (pull form) or dQ and dEj (scatter form)");
  if (!attn_shape_ok(k, D, A)) return fail(TAGREC_E_UNSUPPORTED, kShapeMsg);
  TAGREC_REQUIRE(n_wt >= 1 && static_cast<size_t>(n_wt + 1) * A * sizeof(float) <= 48 * 1024,
                 "tgcn_attn_bwd: weight table too large for LDS");
  TAGREC_REQUIRE(workspace_floats >= tagrec_tgcn_attn_workspace(n_wt, A), "tgcn_attn_bwd: workspace too small");
  TAGREC_REQUIRE(aligned16(Ej) && aligned16(dOut), "tgcn_attn_bwd: embedding rows must be 16-byte aligned");
  if (n <= 0) return TAGREC_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int elems = (n_wt + 1) * A;
  const size_t lds = static_cast<size_t>(elems) * sizeof(float);
  const int64_t want = (n + kAttnWaves - 1) / kAttnWaves;
  const unsigned blocks = static_cast<unsigned>(want < 1 ? 1 : (want < kAttnBlocks ? want : kAttnBlocks));
#define LAUNCH(L)                                                                                       \
  do {                                                                                                  \
    if (dh)                                                                                             \
      tgcn_attn_bwd_kernel<L, true><<<blocks, kAttnThreads, lds, s>>>(P, Q, WT, v, Ej, idx, widx, attn, dOut, n, k, A, n_wt, \
                                                                     dP, dQ, dEj, dh, workspace);       \
    else                                                                                                \
      tgcn_attn_bwd_kernel<L, false><<<blocks, kAttnThreads, lds, s>>>(P, Q, WT, v, Ej, idx, widx, attn, dOut, n, k, A, n_wt, \
                                                                      dP, dQ, dEj, dh, workspace);      \
  } while (0)
  switch (D) {
    case 16: LAUNCH(4); break;
    case 32: LAUNCH(8); break;
    case 64: LAUNCH(16); break;
    case 128: LAUNCH(32); break;
    default: LAUNCH(64); break;
  }
#undef LAUNCH
  TAGREC_LAUNCH_CHECK();
  tgcn_attn_fold_kernel<<<(elems + 15) / 16, 256, 0, s>>>(workspace, static_cast<int>(blocks), elems, n_wt * A, dWT, dv);
  TAGREC_LAUNCH_CHECK();
  return TAGREC_OK;
}
