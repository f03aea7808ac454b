// K2 / K4 / K5: row normalisation, BPR triplet loss (forward + scatter backward), fused Adam.
//
// Replaces, in the reference (/root/reference):
//   F.normalize(p=2, dim=1)                      model/ngcf.py:86, model/tgcn.py:220-222
//   advanced-index gathers + mul_loss + l2reg    model/lightgcn.py:68-82, model/help/loss.py:4-12, 27-32
//   index_put_(accumulate=True) in autograd      (backward of the gathers above)
//   torch.optim.Adam.step                        com.py:14,25,69
// All HBM-bound; one wavefront (or a power-of-two slice of one) owns one row / triplet, reductions
// are cross-lane shuffles, the loss sum is a fixed-order two-stage reduction.
#include "common.h"

namespace tagrec {

constexpr int kThreads = 256;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int m = 1; m < kWave; m <<= 1) v += __shfl_xor(v, m);
  return v;
}

// ------------------------------------------------------------------------------------------------
// F.normalize: one wave per row, lanes stride the columns.
__global__ __launch_bounds__(kThreads) void rownorm_fwd_kernel(const float* __restrict__ X, float* __restrict__ Z,
                                                               int64_t ldz, float* __restrict__ inv_norm,
                                                               int64_t n_rows, int D) {
  const int lane = threadIdx.x & (kWave - 1);
  const int64_t r = static_cast<int64_t>(blockIdx.x) * (kThreads / kWave) + (threadIdx.x >> 6);
  if (r >= n_rows) return;
  const float* x = X + r * D;
  float ss = 0.f;
  for (int k = lane; k < D; k += kWave) ss = fmaf(x[k], x[k], ss);
  ss = wave_sum(ss);
  const float den = fmaxf(sqrtf(ss), 1e-12f);
  float* z = Z + r * ldz;
  for (int k = lane; k < D; k += kWave) z[k] = x[k] / den;
  if (lane == 0 && inv_norm) inv_norm[r] = 1.0f / den;
}

__global__ __launch_bounds__(kThreads) void rownorm_bwd_kernel(const float* __restrict__ Xraw,
                                                               const float* __restrict__ inv_norm,
                                                               const float* __restrict__ dZ, int64_t lddz, float s,
                                                               float* __restrict__ dX, int accumulate,
                                                               int64_t n_rows, int D, uint8_t* __restrict__ row_flags) {
  const int lane = threadIdx.x & (kWave - 1);
  const int64_t r = static_cast<int64_t>(blockIdx.x) * (kThreads / kWave) + (threadIdx.x >> 6);
  if (r >= n_rows) return;
  const float* dz = dZ + r * lddz;
  float* o = dX + r * D;
  if (!accumulate) {
    // the gradient is zero outside the batch rows: such a row's result is zero whatever x is, so x is not read
    bool any_dz = false;
    for (int k = lane; k < D; k += kWave) any_dz |= dz[k] != 0.f;
    if (!__any(any_dz)) {
      for (int k = lane; k < D; k += kWave) o[k] = 0.f;
      if (row_flags && lane == 0) row_flags[r] = 0;
      return;
    }
  }
  const float inv = inv_norm[r];
  const float* x = Xraw + r * D;
  float dot = 0.f;
  for (int k = lane; k < D; k += kWave) dot = fmaf(x[k] * inv, dz[k] * s, dot);
  dot = wave_sum(dot);
  if (inv >= 1e12f) dot = 0.f;  // norm was clamped to eps: the denominator is a constant
  bool nz = false;
  for (int k = lane; k < D; k += kWave) {
    float g = inv * (dz[k] * s - x[k] * inv * dot);
    if (accumulate) g += o[k];
    o[k] = g;
    nz |= g != 0.f;
  }
  if (row_flags) {                        // 1 iff the row holds a non-zero (see EpiArgs::in_flags in spmm.hip)
    const bool any = __any(nz);
    if (lane == 0) row_flags[r] = any;
  }
}

// The same for D = 4 LPR in {8 .. 256}, not accumulating: LPR lanes x 16 B per row, 64 / LPR rows per wavefront.  At the
// head of the backward chain dZ is zero outside the <= 3 B batch rows, so nearly every row is "read 4 D bytes, write 4 D
// zero bytes": one float per lane and one row per wave (the kernel above) leaves too few bytes in flight for that.
template <int LPR>
__global__ __launch_bounds__(kThreads) void rownorm_bwd_vec_kernel(const float* __restrict__ Xraw,
                                                                   const float* __restrict__ inv_norm,
                                                                   const float* __restrict__ dZ, int64_t lddz, float s,
                                                                   float* __restrict__ dX, int64_t n_rows,
                                                                   uint8_t* __restrict__ row_flags) {
  constexpr int NPI = kWave / LPR;
  const int lane = threadIdx.x & (kWave - 1);
  const int grp = lane / LPR, c4 = lane % LPR;
  const int64_t r = (static_cast<int64_t>(blockIdx.x) * (kThreads / kWave) + (threadIdx.x >> 6)) * NPI + grp;
  const bool ok = r < n_rows;
  float4 dz = make_float4(0.f, 0.f, 0.f, 0.f);
  if (ok) dz = *reinterpret_cast<const float4*>(dZ + r * lddz + 4 * c4);
  const unsigned long long lanes = __ballot(dz.x != 0.f || dz.y != 0.f || dz.z != 0.f || dz.w != 0.f);
  const unsigned long long mine = LPR == kWave ? lanes : (lanes >> (grp * LPR)) & ((1ull << (LPR % kWave)) - 1);
  float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
  bool nz = false;
  if (mine != 0) {                                   // uniform over the row's lane group
    const float inv = inv_norm[r];
    const float4 x = *reinterpret_cast<const float4*>(Xraw + r * (4 * LPR) + 4 * c4);
    float dot = fmaf(x.x * inv, dz.x * s, fmaf(x.y * inv, dz.y * s, fmaf(x.z * inv, dz.z * s, (x.w * inv) * (dz.w * s))));
#pragma unroll
    for (int m = 1; m < LPR; m <<= 1) dot += __shfl_xor(dot, m);
    if (inv >= 1e12f) dot = 0.f;                     // norm was clamped to eps: the denominator is a constant
    g = make_float4(inv * (dz.x * s - x.x * inv * dot), inv * (dz.y * s - x.y * inv * dot),
                    inv * (dz.z * s - x.z * inv * dot), inv * (dz.w * s - x.w * inv * dot));
    nz = g.x != 0.f || g.y != 0.f || g.z != 0.f || g.w != 0.f;
  }
  const unsigned long long nzl = __ballot(nz);
  if (!ok) return;
  *reinterpret_cast<float4*>(dX + r * (4 * LPR) + 4 * c4) = g;
  if (row_flags && c4 == 0)
    row_flags[r] = (LPR == kWave ? nzl : (nzl >> (grp * LPR)) & ((1ull << (LPR % kWave)) - 1)) != 0;
}

template <int LPR>
static void launch_rownorm_bwd_vec(const float* X_raw, const float* inv_norm, const float* dZ, int64_t lddz, float s, float* dX,
                                   int64_t n_rows, uint8_t* row_flags, hipStream_t st) {
  const int64_t rows_per_block = (kThreads / kWave) * (kWave / LPR);
  rownorm_bwd_vec_kernel<LPR><<<static_cast<unsigned>((n_rows + rows_per_block - 1) / rows_per_block), kThreads, 0, st>>>(
      X_raw, inv_norm, dZ, lddz, s, dX, n_rows, row_flags);
}

// true (and launched) when the vector kernel applies: D = 8 .. 256 a power of two, rows 16-byte aligned
static bool rownorm_bwd_vec(const float* X_raw, const float* inv_norm, const float* dZ, int64_t lddz, float s, float* dX,
                            int64_t n_rows, int D, uint8_t* row_flags, hipStream_t st) {
  if (!(aligned16(X_raw) && aligned16(dZ) && aligned16(dX) && lddz % 4 == 0)) return false;
  switch (D) {
    case 8: launch_rownorm_bwd_vec<2>(X_raw, inv_norm, dZ, lddz, s, dX, n_rows, row_flags, st); return true;
    case 16: launch_rownorm_bwd_vec<4>(X_raw, inv_norm, dZ, lddz, s, dX, n_rows, row_flags, st); return true;
    case 32: launch_rownorm_bwd_vec<8>(X_raw, inv_norm, dZ, lddz, s, dX, n_rows, row_flags, st); return true;
    case 64: launch_rownorm_bwd_vec<16>(X_raw, inv_norm, dZ, lddz, s, dX, n_rows, row_flags, st); return true;
    case 128: launch_rownorm_bwd_vec<32>(X_raw, inv_norm, dZ, lddz, s, dX, n_rows, row_flags, st); return true;
    case 256: launch_rownorm_bwd_vec<64>(X_raw, inv_norm, dZ, lddz, s, dX, n_rows, row_flags, st); return true;
    default: return false;
  }
}

// row_flags[r] = 1 iff row r of X [n, D] holds a non-zero
__global__ __launch_bounds__(kThreads) void row_flags_kernel(const float* __restrict__ X, int64_t n_rows, int D,
                                                             uint8_t* __restrict__ row_flags) {
  const int lane = threadIdx.x & (kWave - 1);
  const int64_t r = static_cast<int64_t>(blockIdx.x) * (kThreads / kWave) + (threadIdx.x >> 6);
  if (r >= n_rows) return;
  bool nz = false;
  for (int k = lane; k < D; k += kWave) nz |= X[r * D + k] != 0.f;
  const bool any = __any(nz);
  if (lane == 0) row_flags[r] = any;
}

// number of non-zero entries of a byte array (the row flags above), added to *count
__global__ __launch_bounds__(kThreads) void count_flags_kernel(const uint8_t* __restrict__ flags, int64_t n, unsigned* __restrict__ count) {
  unsigned c = 0;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kThreads;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; i < n; i += stride) c += flags[i] != 0;
#pragma unroll
  for (int m = 1; m < kWave; m <<= 1) c += __shfl_xor(c, m);
  if ((threadIdx.x & (kWave - 1)) == 0 && c) atomicAdd(count, c);
}

// ------------------------------------------------------------------------------------------------
// Column-sharded tables (dist.py, feature sharding): each rank holds D/G columns of every row, so everything that
// reduces over a row's columns is split into "local partial" + all-reduce + "apply".
//   row_scale_acc : acc[r,:] += s * inv[r] * y[r,:]                         (layer mean, after the norm all-reduce)
//   row_dot       : out[r]    = inv[r] * s * sum_c x[r,c] dZ[r,c]           (local part of z . (s dZ))
//   rownorm_bwd_dot: dX[r,:]  = inv[r] * (s dZ[r,:] - x[r,:] inv[r] dot[r]) (dot all-reduced by the caller)
__global__ __launch_bounds__(kThreads) void row_scale_acc_kernel(const float* __restrict__ Y, const float* __restrict__ inv,
                                                                 float s, float* __restrict__ acc, int64_t n_rows, int D) {
  const int64_t total = n_rows * D;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kThreads;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; i < total; i += stride)
    acc[i] = fmaf(s * inv[i / D], Y[i], acc[i]);
}

__global__ __launch_bounds__(kThreads) void row_dot_kernel(const float* __restrict__ X, const float* __restrict__ inv,
                                                           const float* __restrict__ dZ, float s, float* __restrict__ out,
                                                           int64_t n_rows, int D) {
  const int lane = threadIdx.x & (kWave - 1);
  const int64_t r = static_cast<int64_t>(blockIdx.x) * (kThreads / kWave) + (threadIdx.x >> 6);
  if (r >= n_rows) return;
  float d = 0.f;
  for (int k = lane; k < D; k += kWave) d = fmaf(X[r * D + k], dZ[r * D + k], d);
  d = wave_sum(d);
  if (lane == 0) out[r] = inv[r] * s * d;
}

__global__ __launch_bounds__(kThreads) void rownorm_bwd_dot_kernel(const float* __restrict__ X, const float* __restrict__ inv,
                                                                   const float* __restrict__ dZ, const float* __restrict__ dot,
                                                                   float s, float* __restrict__ dX, int64_t n_rows, int D) {
  const int64_t total = n_rows * D;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kThreads;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; i < total; i += stride) {
    const int64_t r = i / D;
    const float iv = inv[r];
    const float dt = iv >= 1e12f ? 0.f : dot[r];
    dX[i] = iv * (s * dZ[i] - X[i] * iv * dt);
  }
}

// local part of the BPR scores and of the L2 term: dots[b] = (u.p, u.n, 0.5(|u|^2+|p|^2+|n|^2)) over this shard's columns
__global__ __launch_bounds__(kThreads) void bpr_dots_kernel(const float* __restrict__ U, const float* __restrict__ I, int64_t ld,
                                                            int D, const float* __restrict__ Ur, const float* __restrict__ Ir,
                                                            int64_t ldr, int Dr, const int64_t* __restrict__ trip, int64_t B,
                                                            float* __restrict__ dots) {
  const int lane = threadIdx.x & (kWave - 1);
  const int64_t b = static_cast<int64_t>(blockIdx.x) * (kThreads / kWave) + (threadIdx.x >> 6);
  if (b >= B) return;
  const int64_t u = trip[3 * b], p = trip[3 * b + 1], n = trip[3 * b + 2];
  float pos = 0.f, neg = 0.f, ss = 0.f;
  for (int k = lane; k < D; k += kWave) {
    const float uv = U[u * ld + k];
    pos = fmaf(uv, I[p * ld + k], pos);
    neg = fmaf(uv, I[n * ld + k], neg);
  }
  if (Ur)
    for (int k = lane; k < Dr; k += kWave) {
      const float a = Ur[u * ldr + k], bb = Ir[p * ldr + k], cc = Ir[n * ldr + k];
      ss = fmaf(a, a, fmaf(bb, bb, fmaf(cc, cc, ss)));
    }
  pos = wave_sum(pos); neg = wave_sum(neg); ss = wave_sum(ss);
  if (lane == 0) { dots[3 * b] = pos; dots[3 * b + 1] = neg; dots[3 * b + 2] = 0.5f * ss; }
}

// ------------------------------------------------------------------------------------------------
// BPR forward: one wave per triplet.  pos = u.p, neg = u.n; per-triplet loss + sigmoid coefficient;
// block partial sums (loss, 0.5*||.||^2) in triplet order.
__device__ __forceinline__ float softplus_torch(float x) {  // F.softplus(beta=1, threshold=20)
  return x > 20.f ? x : log1pf(expf(x));
}
__device__ __forceinline__ float neg_logsigmoid_torch(float y) {  // -F.logsigmoid(y)
  return -(fminf(y, 0.f) - log1pf(expf(-fabsf(y))));
}

__global__ __launch_bounds__(kThreads) void bpr_fwd_kernel(const float* __restrict__ U, const float* __restrict__ I,
                                                           int64_t ld, int D, const float* __restrict__ Ur,
                                                           const float* __restrict__ Ir, int64_t ldr, int Dr,
                                                           const int64_t* __restrict__ trip, int64_t B, int loss_kind,
                                                           float* __restrict__ coef, float* __restrict__ partials) {
  __shared__ float sh[2][kThreads / kWave];
  const int lane = threadIdx.x & (kWave - 1);
  const int w = threadIdx.x >> 6;
  const int64_t b = static_cast<int64_t>(blockIdx.x) * (kThreads / kWave) + w;
  float loss = 0.f, reg = 0.f;
  if (b < B) {
    const int64_t u = trip[3 * b], p = trip[3 * b + 1], n = trip[3 * b + 2];
    const float* ur = U + u * ld;
    const float* pr = I + p * ld;
    const float* nr = I + n * ld;
    float pos = 0.f, neg = 0.f;
    for (int k = lane; k < D; k += kWave) {
      const float uv = ur[k];
      pos = fmaf(uv, pr[k], pos);
      neg = fmaf(uv, nr[k], neg);
    }
    pos = wave_sum(pos);
    neg = wave_sum(neg);
    const float x = neg - pos;
    loss = (loss_kind == TAGREC_LOSS_LOGSIGMOID) ? neg_logsigmoid_torch(pos - neg) : softplus_torch(x);
    // d loss / d x = sigmoid(x); torch's softplus passes the gradient through past the threshold
    const float c = (loss_kind == TAGREC_LOSS_SOFTPLUS && x > 20.f) ? 1.f : 1.f / (1.f + expf(-x));
    if (lane == 0) coef[b] = c;
    if (Ur) {
      const float* a = Ur + u * ldr;
      const float* bb = Ir + p * ldr;
      const float* cc = Ir + n * ldr;
      float ss = 0.f;
      for (int k = lane; k < Dr; k += kWave) ss = fmaf(a[k], a[k], fmaf(bb[k], bb[k], fmaf(cc[k], cc[k], ss)));
      reg = 0.5f * wave_sum(ss);
    }
  }
  if (lane == 0) { sh[0][w] = loss; sh[1][w] = reg; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float l = 0.f, g = 0.f;
    for (int i = 0; i < kThreads / kWave; ++i) { l += sh[0][i]; g += sh[1][i]; }
    partials[2 * blockIdx.x] = l;
    partials[2 * blockIdx.x + 1] = g;
  }
}

// second stage: one block, fixed order
__global__ __launch_bounds__(kThreads) void bpr_reduce_kernel(const float* __restrict__ partials, int64_t n_part,
                                                              float inv_b, float* __restrict__ loss_out) {
  __shared__ double sh[2][kThreads];
  double l = 0.0, g = 0.0;
  for (int64_t i = threadIdx.x; i < n_part; i += kThreads) { l += partials[2 * i]; g += partials[2 * i + 1]; }
  sh[0][threadIdx.x] = l;
  sh[1][threadIdx.x] = g;
  __syncthreads();
  for (int s = kThreads / 2; s > 0; s >>= 1) {
    if (threadIdx.x < s) { sh[0][threadIdx.x] += sh[0][threadIdx.x + s]; sh[1][threadIdx.x] += sh[1][threadIdx.x + s]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    loss_out[0] = static_cast<float>(sh[0][0] * inv_b);
    loss_out[1] = static_cast<float>(sh[1][0] * inv_b);
  }
}

// BPR backward: purely per-column once coef is known, so lanes map to consecutive columns and every
// atomic wave-instruction covers 256 contiguous bytes of one row (the fast shape for
// global_atomic_add_f32 on gfx950).
__global__ __launch_bounds__(kThreads) void bpr_bwd_kernel(const float* __restrict__ U, const float* __restrict__ I,
                                                           int64_t ld, int D, const float* __restrict__ Ur,
                                                           const float* __restrict__ Ir, int64_t ldr, int Dr,
                                                           const int64_t* __restrict__ trip, int64_t B,
                                                           const float* __restrict__ coef, const float* __restrict__ g,
                                                           float reg, float inv_b, float* __restrict__ dU,
                                                           float* __restrict__ dI, float* __restrict__ dUr,
                                                           float* __restrict__ dIr) {
  const int lane = threadIdx.x & (kWave - 1);
  const int64_t b = static_cast<int64_t>(blockIdx.x) * (kThreads / kWave) + (threadIdx.x >> 6);
  if (b >= B) return;
  const float g0 = g ? g[0] : 1.f;
  const float g1 = g ? g[1] : 1.f;
  const int64_t u = trip[3 * b], p = trip[3 * b + 1], n = trip[3 * b + 2];
  const float c = g0 * coef[b] * inv_b;
  if (dU) {
    for (int k = lane; k < D; k += kWave) {
      const float uv = U[u * ld + k], pv = I[p * ld + k], nv = I[n * ld + k];
      atomicAdd(&dU[u * ld + k], c * (nv - pv));
      atomicAdd(&dI[p * ld + k], -c * uv);
      atomicAdd(&dI[n * ld + k], c * uv);
    }
  }
  const float cr = g1 * reg * inv_b;
  if (Ur && cr != 0.f) {
    for (int k = lane; k < Dr; k += kWave) {
      atomicAdd(&dUr[u * ldr + k], cr * Ur[u * ldr + k]);
      atomicAdd(&dIr[p * ldr + k], cr * Ir[p * ldr + k]);
      atomicAdd(&dIr[n * ldr + k], cr * Ir[n * ldr + k]);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Negative sampler (train_data/utils.py:19-28 `sample_neg_item`, :31-40 `sample_neg_tail`): one uniform draw in
// [0, n_right) per positive row, re-drawn while (left id, draw) is a positive pair.  The positives of a left id are
// the sorted row `cols[rowptr[l] .. rowptr[l+1])`, so membership is a binary search in that row.  Counter-based
// generator: draw t of row e under `seed` is a pure function of (seed, e, t) -- reproducible, order-free.
__global__ __launch_bounds__(kThreads) void sample_negative_kernel(const int64_t* __restrict__ left, int64_t n_rows,
                                                                   const int64_t* __restrict__ rowptr,
                                                                   const int32_t* __restrict__ cols, int64_t n_right,
                                                                   uint64_t seed, int64_t* __restrict__ neg) {
  const int64_t e = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
  if (e >= n_rows) return;
  const int64_t l = left[e];
  const int64_t lo0 = rowptr[l], hi0 = rowptr[l + 1];
  const uint64_t base = mix64(seed ^ mix64(static_cast<uint64_t>(e)));
  int64_t draw = 0;
  for (uint32_t t = 0; t < 4096u; ++t) {       // the row cannot cover all of [0, n_right): terminates; bound anyway
    // 64-bit multiply-shift maps a uniform 64-bit word onto [0, n_right) without modulo bias worth measuring
    const uint64_t r = mix64(base + t);
    draw = static_cast<int64_t>(__umul64hi(r, static_cast<uint64_t>(n_right)));
    int64_t lo = lo0, hi = hi0;
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if (cols[mid] < draw) lo = mid + 1; else hi = mid;
    }
    if (lo == hi0 || cols[lo] != draw) break;
  }
  neg[e] = draw;
}

// ------------------------------------------------------------------------------------------------
// TransTag phase (tgcn.py:251-261, loss.py:35-41): rows (user, tag, pos_item, neg_item) of the EGO tables,
//   loss = mean relu(margin + ||u + t - p||_2 - ||u + t - n||_2),  reg = 0.5 (|u|^2 + |t|^2 + |p|^2 + |n|^2) / B.
// One wave per row; forward keeps the two distances for the backward scatter.
__global__ __launch_bounds__(kThreads) void transtag_fwd_kernel(const float* __restrict__ Eu, const float* __restrict__ Ei,
                                                                const float* __restrict__ Et, int64_t ld, int D,
                                                                const int64_t* __restrict__ quad, int64_t B, float margin,
                                                                float* __restrict__ dist, float* __restrict__ partials) {
  __shared__ float sh[2][kThreads / kWave];
  const int lane = threadIdx.x & (kWave - 1);
  const int w = threadIdx.x >> 6;
  const int64_t b = static_cast<int64_t>(blockIdx.x) * (kThreads / kWave) + w;
  float loss = 0.f, reg = 0.f;
  if (b < B) {
    const float* u = Eu + quad[4 * b] * ld;
    const float* t = Et + quad[4 * b + 1] * ld;
    const float* p = Ei + quad[4 * b + 2] * ld;
    const float* n = Ei + quad[4 * b + 3] * ld;
    float sp = 0.f, sn = 0.f, ss = 0.f;
    for (int k = lane; k < D; k += kWave) {
      const float uv = u[k], tv = t[k], pv = p[k], nv = n[k];
      const float h = uv + tv;
      sp = fmaf(h - pv, h - pv, sp);
      sn = fmaf(h - nv, h - nv, sn);
      ss = fmaf(uv, uv, fmaf(tv, tv, fmaf(pv, pv, fmaf(nv, nv, ss))));
    }
    const float ps = sqrtf(wave_sum(sp)), ns = sqrtf(wave_sum(sn));
    loss = fmaxf(margin + ps - ns, 0.f);
    reg = 0.5f * wave_sum(ss);
    if (lane == 0) { dist[2 * b] = ps; dist[2 * b + 1] = ns; }
  }
  if (lane == 0) { sh[0][w] = loss; sh[1][w] = reg; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float l = 0.f, g = 0.f;
    for (int i = 0; i < kThreads / kWave; ++i) { l += sh[0][i]; g += sh[1][i]; }
    partials[2 * blockIdx.x] = l;
    partials[2 * blockIdx.x + 1] = g;
  }
}

__global__ __launch_bounds__(kThreads) void transtag_bwd_kernel(const float* __restrict__ Eu, const float* __restrict__ Ei,
                                                                const float* __restrict__ Et, int64_t ld, int D,
                                                                const int64_t* __restrict__ quad, int64_t B, float margin,
                                                                const float* __restrict__ dist, const float* __restrict__ g,
                                                                float inv_b, float* __restrict__ dEu, float* __restrict__ dEi,
                                                                float* __restrict__ dEt) {
  const int lane = threadIdx.x & (kWave - 1);
  const int64_t b = static_cast<int64_t>(blockIdx.x) * (kThreads / kWave) + (threadIdx.x >> 6);
  if (b >= B) return;
  const float g0 = (g ? g[0] : 1.f) * inv_b, g1 = (g ? g[1] : 1.f) * inv_b;
  const int64_t iu = quad[4 * b] * ld, it = quad[4 * b + 1] * ld, ip = quad[4 * b + 2] * ld, in_ = quad[4 * b + 3] * ld;
  const float ps = dist[2 * b], ns = dist[2 * b + 1];
  const bool live = margin + ps - ns > 0.f;
  // d||x||/dx = x / ||x||, taken as 0 at x = 0 (what torch's norm backward does)
  const float cp = (live && ps > 0.f) ? g0 / ps : 0.f;
  const float cn = (live && ns > 0.f) ? g0 / ns : 0.f;
  for (int k = lane; k < D; k += kWave) {
    const float uv = Eu[iu + k], tv = Et[it + k], pv = Ei[ip + k], nv = Ei[in_ + k];
    const float h = uv + tv;
    const float dh = cp * (h - pv) - cn * (h - nv);
    atomicAdd(&dEu[iu + k], dh + g1 * uv);
    atomicAdd(&dEt[it + k], dh + g1 * tv);
    atomicAdd(&dEi[ip + k], -cp * (h - pv) + g1 * pv);
    atomicAdd(&dEi[in_ + k], cn * (h - nv) + g1 * nv);
  }
}

// ------------------------------------------------------------------------------------------------
// Adam, 16 B per lane, grid-stride, two iterations (eight 16-byte loads) in flight per lane; every array is streamed
// once per step and is far larger than the caches, so all accesses are non-temporal.  28 B of traffic per element.
__global__ __launch_bounds__(kThreads) void adam_kernel(float4* __restrict__ p_, const float4* __restrict__ g_,
                                                        float4* __restrict__ m_, float4* __restrict__ v_, int64_t n4,
                                                        float w1, float b2, float w2, float step_size,
                                                        float bc2_sqrt, float eps, const float* __restrict__ coef) {
  if (coef) { step_size = coef[0]; bc2_sqrt = coef[1]; }      // step-dependent factors kept on the device (graph capture)
  adam_f4* p = reinterpret_cast<adam_f4*>(p_);
  const adam_f4* g = reinterpret_cast<const adam_f4*>(g_);
  adam_f4* m = reinterpret_cast<adam_f4*>(m_);
  adam_f4* v = reinterpret_cast<adam_f4*>(v_);
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kThreads;
  int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
  for (; i + stride < n4; i += 2 * stride) {
    const int64_t j = i + stride;
    const adam_f4 g0 = __builtin_nontemporal_load(g + i), g1 = __builtin_nontemporal_load(g + j);
    adam_f4 m0 = __builtin_nontemporal_load(m + i), m1 = __builtin_nontemporal_load(m + j);
    adam_f4 v0 = __builtin_nontemporal_load(v + i), v1 = __builtin_nontemporal_load(v + j);
    const adam_f4 p0 = __builtin_nontemporal_load(p + i), p1 = __builtin_nontemporal_load(p + j);
    const adam_f4 q0 = adam_update(m0, v0, p0, g0, w1, b2, w2, step_size, bc2_sqrt, eps);
    const adam_f4 q1 = adam_update(m1, v1, p1, g1, w1, b2, w2, step_size, bc2_sqrt, eps);
    __builtin_nontemporal_store(m0, m + i); __builtin_nontemporal_store(v0, v + i); __builtin_nontemporal_store(q0, p + i);
    __builtin_nontemporal_store(m1, m + j); __builtin_nontemporal_store(v1, v + j); __builtin_nontemporal_store(q1, p + j);
  }
  if (i < n4) {
    const adam_f4 g0 = __builtin_nontemporal_load(g + i);
    adam_f4 m0 = __builtin_nontemporal_load(m + i), v0 = __builtin_nontemporal_load(v + i);
    const adam_f4 q0 = adam_update(m0, v0, __builtin_nontemporal_load(p + i), g0, w1, b2, w2, step_size, bc2_sqrt, eps);
    __builtin_nontemporal_store(m0, m + i); __builtin_nontemporal_store(v0, v + i); __builtin_nontemporal_store(q0, p + i);
  }
}

__global__ __launch_bounds__(kThreads) void adam_tail_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                             float* __restrict__ m, float* __restrict__ v,
                                                             int64_t start, int64_t n, float w1, float b2, float w2,
                                                             float step_size, float bc2_sqrt, float eps,
                                                             const float* __restrict__ coef) {
  if (coef) { step_size = coef[0]; bc2_sqrt = coef[1]; }
  const int64_t i = start + static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
  if (i >= n) return;
  float mi = m[i], vi = v[i];
  p[i] = adam_update1(mi, vi, p[i], g[i], w1, b2, w2, step_size, bc2_sqrt, eps);
  m[i] = mi;
  v[i] = vi;
}

// out = mask(seed) * x / (1 - p), the mask of common.h's DropMask (x may alias out)
__global__ __launch_bounds__(kThreads) void dropout_kernel(const float4* x, float4* out, int64_t n4,
                                                           DropMask m) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kThreads;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; i < n4; i += stride) {
    float4 v = x[i];
    drop4(m, i, v.x, v.y, v.z, v.w);
    out[i] = v;
  }
}

// The same mask on a COMPACT set of rows: element (j, c) of x [T, d] takes the draw of element (rows[j], c) of the full [N, d]
// tensor, so every slot of a node that a row list names several times gets that node's mask (the reference draws one mask
// per node: F.dropout on the full layer output, /root/reference/model/ngcf.py:85).
__global__ __launch_bounds__(kThreads) void dropout_rows_kernel(const float4* x, float4* out,
                                                                const int64_t* __restrict__ rows, int64_t T, int d4, DropMask m) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kThreads;
  const int64_t n4 = T * d4;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; i < n4; i += stride) {
    const int64_t j = i / d4;
    float4 v = x[i];
    drop4(m, rows[j] * d4 + (i - j * d4), v.x, v.y, v.z, v.w);
    out[i] = v;
  }
}

// step counter and the two step-dependent factors of Adam, advanced ON the device so a captured graph replays correctly
__global__ void adam_advance_kernel(int64_t* __restrict__ step, float* __restrict__ coef, double lr, double b1, double b2) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const int64_t t = *step + 1;
  *step = t;
  const double bc1 = 1.0 - pow(b1, static_cast<double>(t));
  const double bc2 = 1.0 - pow(b2, static_cast<double>(t));
  coef[0] = static_cast<float>(lr / bc1);
  coef[1] = static_cast<float>(sqrt(bc2));
}

}  // namespace tagrec

using namespace tagrec;

extern "C" int tagrec_rownorm_fwd_f32(const float* X, float* Z, int64_t ldz, float* inv_norm, int64_t n_rows, int D,
                                      void* stream) {
  TAGREC_REQUIRE(X && Z, "rownorm_fwd: null pointer");
  TAGREC_REQUIRE(n_rows >= 0 && D >= 1 && ldz >= D, "rownorm_fwd: bad shape");
  if (n_rows == 0) return TAGREC_OK;
  const unsigned blocks = static_cast<unsigned>((n_rows + 3) / 4);
  rownorm_fwd_kernel<<<blocks, kThreads, 0, static_cast<hipStream_t>(stream)>>>(X, Z, ldz, inv_norm, n_rows, D);
  TAGREC_LAUNCH_CHECK();
  return TAGREC_OK;
}

extern "C" int tagrec_rownorm_bwd_f32(const float* X_raw, const float* inv_norm, const float* dZ, int64_t lddz,
                                      float d_scale, float* dX, int accumulate, int64_t n_rows, int D, void* stream) {
  TAGREC_REQUIRE(X_raw && inv_norm && dZ && dX, "rownorm_bwd: null pointer");
  TAGREC_REQUIRE(n_rows >= 0 && D >= 1 && lddz >= D, "rownorm_bwd: bad shape");
  if (n_rows == 0) return TAGREC_OK;
  const unsigned blocks = static_cast<unsigned>((n_rows + 3) / 4);
  if (accumulate || !rownorm_bwd_vec(X_raw, inv_norm, dZ, lddz, d_scale, dX, n_rows, D, nullptr, static_cast<hipStream_t>(stream)))
    rownorm_bwd_kernel<<<blocks, kThreads, 0, static_cast<hipStream_t>(stream)>>>(X_raw, inv_norm, dZ, lddz, d_scale, dX,
                                                                                  accumulate, n_rows, D, nullptr);
  TAGREC_LAUNCH_CHECK();
  return TAGREC_OK;
}

namespace tagrec {
int count_flags(const uint8_t* flags, int64_t n, unsigned* count, hipStream_t s) {
  TAGREC_HIP(hipMemsetAsync(count, 0, sizeof(unsigned), s));
  if (n == 0) return TAGREC_OK;
  int64_t blocks = (n + kThreads * 16 - 1) / (kThreads * 16);
  if (blocks > 256) blocks = 256;
  count_flags_kernel<<<static_cast<unsigned>(blocks), kThreads, 0, s>>>(flags, n, count);
  TAGREC_LAUNCH_CHECK();
  return TAGREC_OK;
}
}  // namespace tagrec

extern "C" int tagrec_rownorm_bwd_flags_f32(const float* X_raw, const float* inv_norm, const float* dZ, int64_t lddz,
                                            float d_scale, float* dX, int accumulate, int64_t n_rows, int D,
                                            uint8_t* row_flags, unsigned* count, void* stream) {
  TAGREC_REQUIRE(X_raw && inv_norm && dZ && dX && row_flags && count, "rownorm_bwd_flags: null pointer");
  TAGREC_REQUIRE(n_rows >= 0 && D >= 1 && lddz >= D, "rownorm_bwd_flags: bad shape");
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (n_rows > 0) {
    if (accumulate || !rownorm_bwd_vec(X_raw, inv_norm, dZ, lddz, d_scale, dX, n_rows, D, row_flags, s))
      rownorm_bwd_kernel<<<static_cast<unsigned>((n_rows + 3) / 4), kThreads, 0, s>>>(X_raw, inv_norm, dZ, lddz, d_scale, dX,
                                                                                     accumulate, n_rows, D, row_flags);
    TAGREC_LAUNCH_CHECK();
  }
  return count_flags(row_flags, n_rows, count, s);
}

extern "C" int tagrec_bpr_fwd_f32(const float* U, const float* I, int64_t ld, int D, const float* Ureg,
                                  const float* Ireg, int64_t ldreg, int Dreg, const int64_t* trip, int64_t B,
                                  int loss_kind, float* coef, float* partials, float* loss_out, void* stream) {
  TAGREC_REQUIRE(U && I && trip && coef && partials && loss_out, "bpr_fwd: null pointer");
  TAGREC_REQUIRE(B >= 1 && D >= 1 && ld >= D, "bpr_fwd: bad shape");
  TAGREC_REQUIRE((Ureg == nullptr) == (Ireg == nullptr), "bpr_fwd: Ureg/Ireg must both be given or both null");
  TAGREC_REQUIRE(!Ureg || (Dreg >= 1 && ldreg >= Dreg), "bpr_fwd: bad reg shape");
  TAGREC_REQUIRE(loss_kind == TAGREC_LOSS_SOFTPLUS || loss_kind == TAGREC_LOSS_LOGSIGMOID, "bpr_fwd: unknown loss_kind");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int64_t blocks = (B + 3) / 4;
  bpr_fwd_kernel<<<static_cast<unsigned>(blocks), kThreads, 0, s>>>(U, I, ld, D, Ureg, Ireg, ldreg, Dreg, trip, B,
                                                                     loss_kind, coef, partials);
  TAGREC_LAUNCH_CHECK();
  bpr_reduce_kernel<<<1, kThreads, 0, s>>>(partials, blocks, 1.0f / static_cast<float>(B), loss_out);
  TAGREC_LAUNCH_CHECK();
  return TAGREC_OK;
}

extern "C" int tagrec_bpr_bwd_f32(const float* U, const float* I, int64_t ld, int D, const float* Ureg,
                                  const float* Ireg, int64_t ldreg, int Dreg, const int64_t* trip, int64_t B,
                                  const float* coef, const float* g, float reg, float* dU, float* dI, float* dUreg,
                                  float* dIreg, void* stream) {
  TAGREC_REQUIRE(U && I && trip && coef, "bpr_bwd: null pointer");
  TAGREC_REQUIRE((dU == nullptr) == (dI == nullptr), "bpr_bwd: dU/dI must both be given or both null (reg-only pass)");
  TAGREC_REQUIRE(B >= 1 && D >= 1 && ld >= D, "bpr_bwd: bad shape");
  TAGREC_REQUIRE(!Ureg || (Ireg && dUreg && dIreg && Dreg >= 1 && ldreg >= Dreg), "bpr_bwd: bad reg arguments");
  const int64_t blocks = (B + 3) / 4;
  bpr_bwd_kernel<<<static_cast<unsigned>(blocks), kThreads, 0, static_cast<hipStream_t>(stream)>>>(
      U, I, ld, D, Ureg, Ireg, ldreg, Dreg, trip, B, coef, g, reg, 1.0f / static_cast<float>(B), dU, dI, dUreg, dIreg);
  TAGREC_LAUNCH_CHECK();
  return TAGREC_OK;
}

extern "C" int tagrec_row_scale_acc_f32(const float* Y, const float* inv, float s, float* acc, int64_t n_rows, int D,
                                        void* stream) {
  TAGREC_REQUIRE(Y && inv && acc && n_rows >= 0 && D >= 1, "row_scale_acc: bad argument");
  if (n_rows == 0) return TAGREC_OK;
  int64_t blocks = (n_rows * D + kThreads - 1) / kThreads;
  if (blocks > 256 * 16) blocks = 256 * 16;
  row_scale_acc_kernel<<<static_cast<unsigned>(blocks), kThreads, 0, static_cast<hipStream_t>(stream)>>>(Y, inv, s, acc, n_rows, D);
  TAGREC_LAUNCH_CHECK();
  return TAGREC_OK;
}

extern "C" int tagrec_row_dot_f32(const float* X, const float* inv, const float* dZ, float s, float* out, int64_t n_rows,
                                  int D, void* stream) {
  TAGREC_REQUIRE(X && inv && dZ && out && n_rows >= 0 && D >= 1, "row_dot: bad argument");
  if (n_rows == 0) return TAGREC_OK;
  row_dot_kernel<<<static_cast<unsigned>((n_rows + 3) / 4), kThreads, 0, static_cast<hipStream_t>(stream)>>>(X, inv, dZ, s, out,
                                                                                                             n_rows, D);
  TAGREC_LAUNCH_CHECK();
  return TAGREC_OK;
}

extern "C" int tagrec_rownorm_bwd_dot_f32(const float* X, const float* inv, const float* dZ, const float* dot, float s,
                                          float* dX, int64_t n_rows, int D, void* stream) {
  TAGREC_REQUIRE(X && inv && dZ && dot && dX && n_rows >= 0 && D >= 1, "rownorm_bwd_dot: bad argument");
  if (n_rows == 0) return TAGREC_OK;
  int64_t blocks = (n_rows * D + kThreads - 1) / kThreads;
  if (blocks > 256 * 16) blocks = 256 * 16;
  rownorm_bwd_dot_kernel<<<static_cast<unsigned>(blocks), kThreads, 0, static_cast<hipStream_t>(stream)>>>(X, inv, dZ, dot, s, dX,
                                                                                                          n_rows, D);
  TAGREC_LAUNCH_CHECK();
  return TAGREC_OK;
}

extern "C" int tagrec_bpr_dots_f32(const float* U, const float* I, int64_t ld, int D, const float* Ureg, const float* Ireg,
                                   int64_t ldreg, int Dreg, const int64_t* trip, int64_t B, float* dots, void* stream) {
  TAGREC_REQUIRE(U && I && trip && dots, "bpr_dots: null pointer");
  TAGREC_REQUIRE(B >= 1 && D >= 1 && ld >= D, "bpr_dots: bad shape");
  TAGREC_REQUIRE((Ureg == nullptr) == (Ireg == nullptr), "bpr_dots: Ureg/Ireg must both be given or both null");
  bpr_dots_kernel<<<static_cast<unsigned>((B + 3) / 4), kThreads, 0, static_cast<hipStream_t>(stream)>>>(U, I, ld, D, Ureg, Ireg,
                                                                                                        ldreg, Dreg, trip, B, dots);
  TAGREC_LAUNCH_CHECK();
  return TAGREC_OK;
}

extern "C" int tagrec_sample_negative_i64(const int64_t* left, int64_t n_rows, const int64_t* rowptr, const int32_t* cols,
                                          int64_t n_left, int64_t n_right, uint64_t seed, int64_t* neg, void* stream) {
  TAGREC_REQUIRE(left && rowptr && neg && (cols || n_rows == 0), "sample_negative: null pointer");
  TAGREC_REQUIRE(n_rows >= 0 && n_left >= 1 && n_right >= 1, "sample_negative: bad shape");
  if (n_rows == 0) return TAGREC_OK;
  const int64_t blocks = (n_rows + kThreads - 1) / kThreads;
  sample_negative_kernel<<<static_cast<unsigned>(blocks), kThreads, 0, static_cast<hipStream_t>(stream)>>>(
      left, n_rows, rowptr, cols, n_right, seed, neg);
  TAGREC_LAUNCH_CHECK();
  return TAGREC_OK;
}

extern "C" int tagrec_transtag_fwd_f32(const float* Eu, const float* Ei, const float* Et, int64_t ld, int D,
                                       const int64_t* quad, int64_t B, float margin, float* dist, float* partials,
                                       float* loss_out, void* stream) {
  TAGREC_REQUIRE(Eu && Ei && Et && quad && dist && partials && loss_out, "transtag_fwd: null pointer");
  TAGREC_REQUIRE(B >= 1 && D >= 1 && ld >= D, "transtag_fwd: bad shape");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int64_t blocks = (B + 3) / 4;
  transtag_fwd_kernel<<<static_cast<unsigned>(blocks), kThreads, 0, s>>>(Eu, Ei, Et, ld, D, quad, B, margin, dist, partials);
  TAGREC_LAUNCH_CHECK();
  bpr_reduce_kernel<<<1, kThreads, 0, s>>>(partials, blocks, 1.0f / static_cast<float>(B), loss_out);
  TAGREC_LAUNCH_CHECK();
  return TAGREC_OK;
}

extern "C" int tagrec_transtag_bwd_f32(const float* Eu, const float* Ei, const float* Et, int64_t ld, int D,
                                       const int64_t* quad, int64_t B, float margin, const float* dist, const float* g,
                                       float* dEu, float* dEi, float* dEt, void* stream) {
  TAGREC_REQUIRE(Eu && Ei && Et && quad && dist && dEu && dEi && dEt, "transtag_bwd: null pointer");
  TAGREC_REQUIRE(B >= 1 && D >= 1 && ld >= D, "transtag_bwd: bad shape");
  const int64_t blocks = (B + 3) / 4;
  transtag_bwd_kernel<<<static_cast<unsigned>(blocks), kThreads, 0, static_cast<hipStream_t>(stream)>>>(
      Eu, Ei, Et, ld, D, quad, B, margin, dist, g, 1.0f / static_cast<float>(B), dEu, dEi, dEt);
  TAGREC_LAUNCH_CHECK();
  return TAGREC_OK;
}

namespace {
int launch_adam(float* p, const float* g, float* m, float* v, int64_t n, float b1, float b2, float eps, float step_size,
                float bc2_sqrt, const float* coef, hipStream_t s) {
  const float w1 = static_cast<float>(1.0 - static_cast<double>(b1));
  const float w2 = static_cast<float>(1.0 - static_cast<double>(b2));
  const bool vec = aligned16(p) && aligned16(g) && aligned16(m) && aligned16(v);
  const int64_t n4 = vec ? n / 4 : 0;
  if (n4 > 0) {
    int64_t blocks = (n4 + kThreads - 1) / kThreads;
    if (blocks > 256 * 16) blocks = 256 * 16;
    adam_kernel<<<static_cast<unsigned>(blocks), kThreads, 0, s>>>(
        reinterpret_cast<float4*>(p), reinterpret_cast<const float4*>(g), reinterpret_cast<float4*>(m),
        reinterpret_cast<float4*>(v), n4, w1, b2, w2, step_size, bc2_sqrt, eps, coef);
    TAGREC_LAUNCH_CHECK();
  }
  const int64_t done = n4 * 4;
  if (done < n) {
    const int64_t blocks = (n - done + kThreads - 1) / kThreads;
    adam_tail_kernel<<<static_cast<unsigned>(blocks), kThreads, 0, s>>>(p, g, m, v, done, n, w1, b2, w2, step_size,
                                                                        bc2_sqrt, eps, coef);
    TAGREC_LAUNCH_CHECK();
  }
  return TAGREC_OK;
}
}  // namespace

extern "C" int tagrec_adam_f32(float* p, const float* g, float* m, float* v, int64_t n, float lr, float b1, float b2,
                               float eps, int64_t step, void* stream) {
  TAGREC_REQUIRE(p && g && m && v, "adam: null pointer");
  TAGREC_REQUIRE(n >= 0 && step >= 1, "adam: bad n or step");
  if (n == 0) return TAGREC_OK;
  // same host arithmetic as torch's _single_tensor_adam (python floats = doubles)
  const double bc1 = 1.0 - pow(static_cast<double>(b1), static_cast<double>(step));
  const double bc2 = 1.0 - pow(static_cast<double>(b2), static_cast<double>(step));
  return launch_adam(p, g, m, v, n, b1, b2, eps, static_cast<float>(static_cast<double>(lr) / bc1),
                     static_cast<float>(sqrt(bc2)), nullptr, static_cast<hipStream_t>(stream));
}

// Many small parameter tensors in ONE launch (a TGCN layer has 21; one launch each costs more in launch latency than in
// work, and the burst lets the GPU catch up with the host at the end of every step).  Scalar loads / stores: the tensors
// are small and need not be 16-byte aligned.
namespace {
constexpr int kAdamMulti = 64;
struct AdamMultiArgs {
  float* p[kAdamMulti];
  const float* g[kAdamMulti];
  float* m[kAdamMulti];
  float* v[kAdamMulti];
  int n[kAdamMulti];
};
__global__ __launch_bounds__(kThreads) void adam_multi_kernel(AdamMultiArgs a, float w1, float b2, float w2, float step_size, float bc2_sqrt,
                                                              float eps) {
  const int t = blockIdx.y;
  const int n = a.n[t];
  float* __restrict__ p = a.p[t];
  const float* __restrict__ g = a.g[t];
  float* __restrict__ m = a.m[t];
  float* __restrict__ v = a.v[t];
  // four consecutive elements per thread through adam_update (common.h): the arithmetic -- and its contraction into FMAs --
  // is the vector kernel's and the fused epilogue's, so the three ways of applying an update agree bit for bit
  for (int i = (blockIdx.x * kThreads + threadIdx.x) * 4; i < n; i += gridDim.x * kThreads * 4) {
    adam_f4 gi = {0.f, 0.f, 0.f, 0.f}, mi = gi, vi = gi, pi = gi;
#pragma unroll
    for (int c = 0; c < 4; ++c)
      if (i + c < n) { gi[c] = g[i + c]; mi[c] = m[i + c]; vi[c] = v[i + c]; pi[c] = p[i + c]; }
    const adam_f4 qi = adam_update(mi, vi, pi, gi, w1, b2, w2, step_size, bc2_sqrt, eps);
#pragma unroll
    for (int c = 0; c < 4; ++c)
      if (i + c < n) { m[i + c] = mi[c]; v[i + c] = vi[c]; p[i + c] = qi[c]; }
  }
}
}  // namespace

extern "C" int tagrec_adam_multi_f32(int n_tensors, float* const* p, const float* const* g, float* const* m, float* const* v,
                                     const int64_t* n, float lr, float b1, float b2, float eps, int64_t step, void* stream) {
  TAGREC_REQUIRE(n_tensors >= 0 && (n_tensors == 0 || (p && g && m && v && n)), "adam_multi: null pointer");
  TAGREC_REQUIRE(step >= 1, "adam_multi: bad step");
  const double bc1 = 1.0 - pow(static_cast<double>(b1), static_cast<double>(step));
  const double bc2 = 1.0 - pow(static_cast<double>(b2), static_cast<double>(step));
  const float step_size = static_cast<float>(static_cast<double>(lr) / bc1), bc2_sqrt = static_cast<float>(sqrt(bc2));
  const float w1 = static_cast<float>(1.0 - static_cast<double>(b1)), w2 = static_cast<float>(1.0 - static_cast<double>(b2));
  for (int t0 = 0; t0 < n_tensors; t0 += kAdamMulti) {
    AdamMultiArgs a;
    const int cnt = n_tensors - t0 < kAdamMulti ? n_tensors - t0 : kAdamMulti;
    int64_t most = 0;
    for (int i = 0; i < cnt; ++i) {
      TAGREC_REQUIRE(n[t0 + i] >= 0 && n[t0 + i] < (1ll << 31), "adam_multi: tensor sizes must fit int32 (use tagrec_adam_f32 for tables)");
      TAGREC_REQUIRE(n[t0 + i] == 0 || (p[t0 + i] && g[t0 + i] && m[t0 + i] && v[t0 + i]), "adam_multi: null tensor");
      a.p[i] = p[t0 + i]; a.g[i] = g[t0 + i]; a.m[i] = m[t0 + i]; a.v[i] = v[t0 + i];
      a.n[i] = static_cast<int>(n[t0 + i]);
      if (n[t0 + i] > most) most = n[t0 + i];
    }
    if (most == 0) continue;
    int64_t bx = (most + 4 * kThreads - 1) / (4 * kThreads);
    if (bx > 1024) bx = 1024;
    adam_multi_kernel<<<dim3(static_cast<unsigned>(bx), static_cast<unsigned>(cnt)), kThreads, 0, static_cast<hipStream_t>(stream)>>>(
        a, w1, b2, w2, step_size, bc2_sqrt, eps);
    TAGREC_LAUNCH_CHECK();
  }
  return TAGREC_OK;
}

extern "C" int tagrec_adam_graph_f32(float* p, const float* g, float* m, float* v, int64_t n, float lr, float b1, float b2,
                                     float eps, int64_t* step_dev, float* coef_dev, void* stream) {
  TAGREC_REQUIRE(p && g && m && v && step_dev && coef_dev, "adam_graph: null pointer");
  TAGREC_REQUIRE(n >= 0, "adam_graph: bad n");
  hipStream_t s = static_cast<hipStream_t>(stream);
  adam_advance_kernel<<<1, 1, 0, s>>>(step_dev, coef_dev, static_cast<double>(lr), static_cast<double>(b1), static_cast<double>(b2));
  TAGREC_LAUNCH_CHECK();
  if (n == 0) return TAGREC_OK;
  return launch_adam(p, g, m, v, n, b1, b2, eps, 0.f, 1.f, coef_dev, s);
}

extern "C" int tagrec_adam_advance(int64_t* step_dev, float* coef_dev, float lr, float b1, float b2, void* stream) {
  TAGREC_REQUIRE(step_dev && coef_dev, "adam_advance: null pointer");
  adam_advance_kernel<<<1, 1, 0, static_cast<hipStream_t>(stream)>>>(step_dev, coef_dev, static_cast<double>(lr), static_cast<double>(b1),
                                                                     static_cast<double>(b2));
  TAGREC_LAUNCH_CHECK();
  return TAGREC_OK;
}

extern "C" int tagrec_dropout_f32(const float* x, float* out, int64_t n, float p, uint64_t seed, void* stream) {
  TAGREC_REQUIRE(x && out, "dropout: null pointer");
  TAGREC_REQUIRE(n >= 0 && n % 4 == 0 && aligned16(x) && aligned16(out), "dropout: need a multiple of 4 elements, 16-byte aligned");
  TAGREC_REQUIRE(p >= 0.f && p < 1.f, "dropout: p must be in [0, 1)");
  if (n == 0) return TAGREC_OK;
  int64_t blocks = (n / 4 + kThreads - 1) / kThreads;
  if (blocks > 256 * 16) blocks = 256 * 16;
  dropout_kernel<<<static_cast<unsigned>(blocks), kThreads, 0, static_cast<hipStream_t>(stream)>>>(
      reinterpret_cast<const float4*>(x), reinterpret_cast<float4*>(out), n / 4, DropMask{p, seed});
  TAGREC_LAUNCH_CHECK();
  return TAGREC_OK;
}

extern "C" int tagrec_dropout_rows_f32(const float* x, float* out, const int64_t* rows, int64_t n_rows, int D, float p, uint64_t seed,
                                       void* stream) {
  TAGREC_REQUIRE(x && out && rows, "dropout_rows: null pointer");
  TAGREC_REQUIRE(n_rows >= 0 && D >= 4 && D % 4 == 0 && aligned16(x) && aligned16(out),
                 "dropout_rows: need a width that is a multiple of 4 and 16-byte aligned buffers");
  TAGREC_REQUIRE(p >= 0.f && p < 1.f, "dropout_rows: p must be in [0, 1)");
  if (n_rows == 0) return TAGREC_OK;
  int64_t blocks = (n_rows * (D / 4) + kThreads - 1) / kThreads;
  if (blocks > 256 * 16) blocks = 256 * 16;
  dropout_rows_kernel<<<static_cast<unsigned>(blocks), kThreads, 0, static_cast<hipStream_t>(stream)>>>(
      reinterpret_cast<const float4*>(x), reinterpret_cast<float4*>(out), rows, n_rows, D / 4, DropMask{p, seed});
  TAGREC_LAUNCH_CHECK();
  return TAGREC_OK;
}

extern "C" int tagrec_row_flags_f32(const float* X, int64_t n_rows, int D, uint8_t* row_flags, unsigned* count, void* stream) {
  TAGREC_REQUIRE(X && row_flags && count, "row_flags: null pointer");
  TAGREC_REQUIRE(n_rows >= 0 && D >= 1, "row_flags: bad shape");
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (n_rows > 0) {
    row_flags_kernel<<<static_cast<unsigned>((n_rows + 3) / 4), kThreads, 0, s>>>(X, n_rows, D, row_flags);
    TAGREC_LAUNCH_CHECK();
  }
  return count_flags(row_flags, n_rows, count, s);
}

// ---- out = s_0 + s_1 + ... + s_{k-1} in ONE pass (k <= 8), summed left to right --------------------------------------
// A table read by several consumers (TGCN: the Q projection, two neighbour attentions, the node's own slot) receives
// one gradient per consumer; autograd adds them pairwise -- three passes over the table per extra gradient.  Here the
// k gradients are read once and the sum written once.  `out` may alias any source (element-wise).
namespace tagrec {
typedef float sum_f4 __attribute__((ext_vector_type(4)));
struct SumSrcs { const sum_f4* p[8]; };
template <int K>
__global__ __launch_bounds__(256) void sum_n_kernel(SumSrcs s, sum_f4* out, int64_t n4) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n4; i += stride) {
    sum_f4 a = __builtin_nontemporal_load(s.p[0] + i);
#pragma unroll
    for (int k = 1; k < K; ++k) a += __builtin_nontemporal_load(s.p[k] + i);
    out[i] = a;
  }
}
__global__ void sum_n_tail_kernel(SumSrcs s, int k, float* out, int64_t lo, int64_t n) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = lo + static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
    float a = reinterpret_cast<const float*>(s.p[0])[i];
    for (int j = 1; j < k; ++j) a += reinterpret_cast<const float*>(s.p[j])[i];
    out[i] = a;
  }
}
}  // namespace tagrec

extern "C" int tagrec_sum_n_f32(float* out, const float* const* srcs, int n_srcs, int64_t n, void* stream) {
  TAGREC_REQUIRE(out != nullptr && srcs != nullptr && n_srcs >= 1 && n_srcs <= 8 && n >= 0, "sum_n: 1 .. 8 sources expected");
  tagrec::SumSrcs s{};
  bool al = aligned16(out);
  for (int k = 0; k < n_srcs; ++k) {
    TAGREC_REQUIRE(srcs[k] != nullptr, "sum_n: null source");
    s.p[k] = reinterpret_cast<const tagrec::sum_f4*>(srcs[k]);
    al = al && aligned16(srcs[k]);
  }
  if (n == 0) return TAGREC_OK;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int64_t n4 = al ? n / 4 : 0;
  if (n4 > 0) {
    const int64_t want = (n4 + 255) / 256;
    const unsigned blocks = static_cast<unsigned>(want < 8192 ? want : 8192);
    tagrec::sum_f4* o = reinterpret_cast<tagrec::sum_f4*>(out);
    switch (n_srcs) {
      case 1: tagrec::sum_n_kernel<1><<<blocks, 256, 0, st>>>(s, o, n4); break;
      case 2: tagrec::sum_n_kernel<2><<<blocks, 256, 0, st>>>(s, o, n4); break;
      case 3: tagrec::sum_n_kernel<3><<<blocks, 256, 0, st>>>(s, o, n4); break;
      case 4: tagrec::sum_n_kernel<4><<<blocks, 256, 0, st>>>(s, o, n4); break;
      case 5: tagrec::sum_n_kernel<5><<<blocks, 256, 0, st>>>(s, o, n4); break;
      case 6: tagrec::sum_n_kernel<6><<<blocks, 256, 0, st>>>(s, o, n4); break;
      case 7: tagrec::sum_n_kernel<7><<<blocks, 256, 0, st>>>(s, o, n4); break;
      default: tagrec::sum_n_kernel<8><<<blocks, 256, 0, st>>>(s, o, n4); break;
    }
    TAGREC_LAUNCH_CHECK();
  }
  if (n4 * 4 < n) {                                        // unaligned operands or the last n % 4 elements
    const int64_t want = (n - n4 * 4 + 255) / 256;
    tagrec::sum_n_tail_kernel<<<static_cast<unsigned>(want < 4096 ? want : 4096), 256, 0, st>>>(s, n_srcs, out, n4 * 4, n);
    TAGREC_LAUNCH_CHECK();
  }
  return TAGREC_OK;
}
