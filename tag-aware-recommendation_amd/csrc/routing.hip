// N4: propagation with DYNAMIC per-factor edge weights (the "neighbourhood routing" of DGCF / DisenGCN).
//
// Replaces, per routing iteration (/root/reference/model/dgcf.py:70-110, model/disengcn.py:28-44):
//     softmax over the K factors of the per-edge logits            dgcf.py:75, disengcn.py:34
//     torch.sparse.sum(adj, dim=1) -> 1/sqrt -> sparse diagonal    dgcf.py:95-100
//     K x torch.sparse.mm with a freshly built sparse tensor       dgcf.py:101-103, disengcn.py:37-40
//     factor_emb[head] / ego[tail] gathers, normalise, tanh, dot   dgcf.py:105-110, disengcn.py:31-33
//     F.normalize of every factor slice                            dgcf.py:87, disengcn.py:26,42
// The reference builds K sparse tensors per iteration and runs K D/K-wide products; here an embedding row is
// gathered ONCE per stored entry and each lane weighs it with the weight of the factor its columns belong to, so
// a routed product costs what a plain one does.  Edge data is factor-interleaved, `W[nnz][K]`, so one 4K-byte
// read brings an entry's K weights.
//
// Layout: embeddings [N, D] with factor k owning columns [k D/K, (k+1) D/K) (the reference's torch.split / cat
// along dim 1); per-node per-factor scalars [N, K]; per-entry per-factor scalars [nnz, K] in CSR entry order.
// One wavefront per CSR row (LPR = D/4 lanes per embedding row, SL = LPR/K lanes per factor slice); long rows are
// cut into chunks exactly as in spmm.hip (same work list, fixed-order fold).  HBM-bound: per stored entry
// 4 + 4K + 4D bytes for a product, 4 + 4K (+4K) + 4D for a score pass.
#include <math.h>

#include <type_traits>

#include "common.h"
#include "graph.h"

namespace tagrec {

__device__ __forceinline__ float4 r4_zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ float r4_dot(const float4& a, const float4& b) {
  return fmaf(a.x, b.x, fmaf(a.y, b.y, fmaf(a.z, b.z, a.w * b.w)));
}
template <int W>
__device__ __forceinline__ float lanes_sum(float v) {
#pragma unroll
  for (int m = 1; m < W; m <<= 1) v += __shfl_xor(v, m);
  return v;
}

// Per-wave LDS staging of 64 entries' column index and K weights (a lane needs the weight of ITS factor for the
// entry its lane group handles, which a register broadcast cannot select).
template <int K>
struct WaveStage {
  int col[kWave];
  float w[kWave * K];
};

template <int K>
__device__ __forceinline__ void stage_entries(WaveStage<K>& st, const int32_t* __restrict__ col, const float* __restrict__ W,
                                              int64_t base, int n, int lane, const uint8_t* __restrict__ flags = nullptr) {
  __builtin_amdgcn_wave_barrier();          // earlier reads of the stage are done (one wave, in-order LDS)
  if (lane < n) {
    const int c = __builtin_nontemporal_load(col + base + lane);
    st.col[lane] = c;
    if (W) {
      // flags[c] == 0: row c of the operand is all zero -> weight 0, and entries of weight 0 are not gathered
      const bool live = !flags || flags[c];
#pragma unroll
      for (int k = 0; k < K; ++k) st.w[lane * K + k] = live ? __builtin_nontemporal_load(W + (base + lane) * K + k) : 0.f;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// sum_j W[j][factor of my columns] * X[col[j], my columns] over entries [start, end).
template <int LPR, int K>
__device__ __forceinline__ float4 routed_gather(const GraphView& g, const float* __restrict__ W, const float* __restrict__ X,
                                                int64_t start, int64_t end, int lane, WaveStage<K>& st,
                                                const uint8_t* __restrict__ flags) {
  constexpr int NPI = kWave / LPR;
  constexpr int SL = LPR / K;
  const int q = lane / LPR;
  const int kf = (lane % LPR) / SL;
  const float4* __restrict__ Xv = reinterpret_cast<const float4*>(X) + (lane % LPR);
  float4 acc = r4_zero();
  for (int64_t base = start; base < end; base += kWave) {
    const int n = (end - base) < kWave ? static_cast<int>(end - base) : kWave;
    stage_entries<K>(st, g.col, W, base, n, lane, flags);
    const int groups = (n + NPI - 1) / NPI;
    for (int gi = 0; gi < groups; gi += 4) {
      float4 x[4];
      float v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int j = (gi + u) * NPI + q;
        const int jj = j < n ? j : 0;
        const int c = st.col[jj];
        v[u] = j < n ? st.w[jj * K + kf] : 0.f;
        x[u] = v[u] != 0.f ? Xv[static_cast<int64_t>(c) * LPR] : r4_zero();      // a x 0 adds exactly 0: not fetched
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        acc.x = fmaf(v[u], x[u].x, acc.x); acc.y = fmaf(v[u], x[u].y, acc.y);
        acc.z = fmaf(v[u], x[u].z, acc.z); acc.w = fmaf(v[u], x[u].w, acc.w);
      }
    }
  }
#pragma unroll
  for (int m = LPR; m < kWave; m <<= 1) {
    acc.x += __shfl_xor(acc.x, m); acc.y += __shfl_xor(acc.y, m);
    acc.z += __shfl_xor(acc.z, m); acc.w += __shfl_xor(acc.w, m);
  }
  return acc;
}

struct RouteEpi {
  float* Y;             // y = post * acc + self + b_scale * B         (nullable)
  float* Yn;            // y / max(||y_slice||, 1e-12) per factor slice (nullable)
  float* inv;           // [N, K] 1 / max(||y_slice||, 1e-12)          (nullable)
  const float* post;    // [N, K] per-row per-factor scale             (nullable = 1)
  const float* self;    // [N, D] added to the product                 (nullable)
  const float* B;       // [N, D] added with b_scale                   (nullable)
  float b_scale;
  const uint8_t* row_mask;   // output rows to compute (nullable = all); the others are left untouched
  const uint8_t* in_flags;   // rows of X that hold a non-zero (nullable); consulted while *in_count < 4/5 of the rows
  const unsigned* in_count;
};

template <int LPR, int K>
__device__ __forceinline__ void route_epilogue(float4 acc, int64_t r, int lane, const RouteEpi& e) {
  constexpr int SL = LPR / K;
  const int kf = (lane % LPR) / SL;
  const int64_t off = r * LPR + (lane % LPR);
  if (e.post) {
    const float p = e.post[r * K + kf];
    acc.x *= p; acc.y *= p; acc.z *= p; acc.w *= p;
  }
  if (e.self) {
    const float4 s = reinterpret_cast<const float4*>(e.self)[off];
    acc.x += s.x; acc.y += s.y; acc.z += s.z; acc.w += s.w;
  }
  if (e.B) {
    const float4 b = reinterpret_cast<const float4*>(e.B)[off];
    acc.x = fmaf(e.b_scale, b.x, acc.x); acc.y = fmaf(e.b_scale, b.y, acc.y);
    acc.z = fmaf(e.b_scale, b.z, acc.z); acc.w = fmaf(e.b_scale, b.w, acc.w);
  }
  const bool writer = lane < LPR;
  if (e.Y && writer) reinterpret_cast<float4*>(e.Y)[off] = acc;
  if (e.Yn || e.inv) {
    const float ss = lanes_sum<SL>(r4_dot(acc, acc));
    const float den = fmaxf(sqrtf(ss), 1e-12f);
    if (e.Yn && writer)
      reinterpret_cast<float4*>(e.Yn)[off] = make_float4(acc.x / den, acc.y / den, acc.z / den, acc.w / den);
    if (e.inv && writer && (lane % SL) == 0) e.inv[r * K + kf] = 1.0f / den;
  }
}

template <int LPR, int K>
__global__ __launch_bounds__(kWavesPerBlock * kWave) void route_spmm_kernel(GraphView g, const float* __restrict__ W,
                                                                             const float* __restrict__ X, RouteEpi e,
                                                                             LongView lv) {
  __shared__ WaveStage<K> stage[kWavesPerBlock];
  const int lane = threadIdx.x & (kWave - 1);
  WaveStage<K>& st = stage[threadIdx.x >> 6];
  const uint8_t* flags = (e.in_flags && 5ull * (*e.in_count) < 4ull * static_cast<unsigned long long>(g.n_cols)) ? e.in_flags : nullptr;
  if (blockIdx.x < lv.chunk_blocks) {
    const int64_t c = static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + (threadIdx.x >> 6);
    if (c >= lv.n_chunks) return;
    const int2 d = lv.chunk_desc[c];
    const int64_t r = lv.long_rows[d.x];
    if (e.row_mask && !e.row_mask[r]) return;
    const int64_t start = g.rowptr[r] + static_cast<int64_t>(d.y) * kChunk;
    const int64_t row_end = g.rowptr[r + 1];
    const int64_t end = (start + kChunk < row_end) ? start + kChunk : row_end;
    const float4 acc = routed_gather<LPR, K>(g, W, X, start, end, lane, st, flags);
    if (lane < LPR) reinterpret_cast<float4*>(lv.slab)[c * LPR + lane] = acc;
    return;
  }
  const int64_t r = static_cast<int64_t>(blockIdx.x - lv.chunk_blocks) * kWavesPerBlock + (threadIdx.x >> 6);
  if (r >= g.n_rows) return;
  if (e.row_mask && !e.row_mask[r]) return;
  const int64_t start = g.rowptr[r], end = g.rowptr[r + 1];
  if (end - start > kLongRow) return;
  const float4 acc = routed_gather<LPR, K>(g, W, X, start, end, lane, st, flags);
  route_epilogue<LPR, K>(acc, r, lane, e);
}

template <int LPR, int K>
__global__ __launch_bounds__(kWavesPerBlock * kWave) void route_spmm_finish_kernel(GraphView g, const int32_t* __restrict__ long_rows,
                                                                                    const int32_t* __restrict__ long_base,
                                                                                    int64_t n_long, const float* __restrict__ slab,
                                                                                    RouteEpi e) {
  const int lane = threadIdx.x & (kWave - 1);
  const int64_t li = static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + (threadIdx.x >> 6);
  if (li >= n_long) return;
  const int64_t r = long_rows[li];
  if (e.row_mask && !e.row_mask[r]) return;
  const int64_t deg = g.rowptr[r + 1] - g.rowptr[r];
  const int nc = static_cast<int>((deg + kChunk - 1) / kChunk);
  const float4* p = reinterpret_cast<const float4*>(slab) + static_cast<int64_t>(long_base[li]) * LPR + (lane % LPR);
  // lane group q sums the chunks q, q + NPI, ... in ascending order, then an xor butterfly over the groups (fixed order;
  // see spmm_finish_kernel)
  constexpr int NPI = kWave / LPR;
  auto add = [](float4& a, const float4& x) { a.x += x.x; a.y += x.y; a.z += x.z; a.w += x.w; };
  float4 acc = r4_zero();
  int k = lane / LPR;
  for (; k + 3 * NPI < nc; k += 4 * NPI) {
    const float4 x0 = p[static_cast<int64_t>(k) * LPR], x1 = p[static_cast<int64_t>(k + NPI) * LPR];
    const float4 x2 = p[static_cast<int64_t>(k + 2 * NPI) * LPR], x3 = p[static_cast<int64_t>(k + 3 * NPI) * LPR];
    add(acc, x0); add(acc, x1); add(acc, x2); add(acc, x3);
  }
  for (; k < nc; k += NPI) add(acc, p[static_cast<int64_t>(k) * LPR]);
#pragma unroll
  for (int m = LPR; m < kWave; m <<= 1) {
    const float4 o = make_float4(__shfl_xor(acc.x, m), __shfl_xor(acc.y, m), __shfl_xor(acc.z, m), __shfl_xor(acc.w, m));
    add(acc, o);
  }
  route_epilogue<LPR, K>(acc, r, lane, e);
}

// Per-entry per-factor score  <H[row, slice k], T[col, slice k]>  written to (or added to) logits[nnz][K].
// Entries are independent, so long rows need no fold: chunk waves write their own entries.
template <int LPR, int K>
__global__ __launch_bounds__(kWavesPerBlock * kWave) void route_score_kernel(GraphView g, const float* __restrict__ H,
                                                                              const float* __restrict__ T,
                                                                              float* __restrict__ logits, int accumulate,
                                                                              LongView lv, const uint8_t* __restrict__ row_mask) {
  constexpr int NPI = kWave / LPR;
  constexpr int SL = LPR / K;
  __shared__ WaveStage<1> stage[kWavesPerBlock];
  const int lane = threadIdx.x & (kWave - 1);
  WaveStage<1>& st = stage[threadIdx.x >> 6];
  int64_t r, start, end;
  if (blockIdx.x < lv.chunk_blocks) {
    const int64_t c = static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + (threadIdx.x >> 6);
    if (c >= lv.n_chunks) return;
    const int2 d = lv.chunk_desc[c];
    r = lv.long_rows[d.x];
    if (row_mask && !row_mask[r]) return;
    start = g.rowptr[r] + static_cast<int64_t>(d.y) * kChunk;
    const int64_t row_end = g.rowptr[r + 1];
    end = (start + kChunk < row_end) ? start + kChunk : row_end;
  } else {
    r = static_cast<int64_t>(blockIdx.x - lv.chunk_blocks) * kWavesPerBlock + (threadIdx.x >> 6);
    if (r >= g.n_rows) return;
    if (row_mask && !row_mask[r]) return;
    start = g.rowptr[r];
    end = g.rowptr[r + 1];
    if (end - start > kLongRow) return;
  }
  const int q = lane / LPR;
  const int kf = (lane % LPR) / SL;
  const float4 h = reinterpret_cast<const float4*>(H)[r * LPR + (lane % LPR)];
  const float4* __restrict__ Tv = reinterpret_cast<const float4*>(T) + (lane % LPR);
  for (int64_t base = start; base < end; base += kWave) {
    const int n = (end - base) < kWave ? static_cast<int>(end - base) : kWave;
    stage_entries<1>(st, g.col, nullptr, base, n, lane);
    const int groups = (n + NPI - 1) / NPI;
    for (int gi = 0; gi < groups; gi += 4) {
      float4 t[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int j = (gi + u) * NPI + q;
        const bool ok = j < n;
        t[u] = ok ? Tv[static_cast<int64_t>(st.col[ok ? j : 0]) * LPR] : r4_zero();
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int j = (gi + u) * NPI + q;
        const float sc = lanes_sum<SL>(r4_dot(h, t[u]));
        if (j < n && (lane % SL) == 0) {
          float* dst = logits + (base + j) * K + kf;
          *dst = accumulate ? *dst + sc : sc;
        }
      }
    }
  }
}

// softmax over the K logits of every entry (thread per entry)
template <int K>
__global__ void route_softmax_kernel(const float* __restrict__ logits, float* __restrict__ w, int64_t nnz) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= nnz) return;
  float v[K];
  float mx = -INFINITY;
#pragma unroll
  for (int k = 0; k < K; ++k) { v[k] = logits[i * K + k]; mx = fmaxf(mx, v[k]); }
  float sum = 0.f;
#pragma unroll
  for (int k = 0; k < K; ++k) { v[k] = expf(v[k] - mx); sum += v[k]; }
#pragma unroll
  for (int k = 0; k < K; ++k) w[i * K + k] = v[k] / sum;
}

// d[r][k] = 1 / sqrt(sum of W[j][k] over the row), inf -> 0 (dgcf.py:95-99)
template <int K>
__device__ __forceinline__ void rowsum_partial(const float* __restrict__ W, int64_t start, int64_t end, int lane, float (&s)[K]) {
#pragma unroll
  for (int k = 0; k < K; ++k) s[k] = 0.f;
  for (int64_t j = start + lane; j < end; j += kWave) {
#pragma unroll
    for (int k = 0; k < K; ++k) s[k] += W[j * K + k];
  }
#pragma unroll
  for (int k = 0; k < K; ++k) s[k] = lanes_sum<kWave>(s[k]);
}

__device__ __forceinline__ float inv_sqrt_or_zero(float s) {
  const float d = 1.0f / sqrtf(s);
  return isinf(d) ? 0.f : d;
}

template <int K>
__global__ __launch_bounds__(kWavesPerBlock * kWave) void route_rowsum_kernel(GraphView g, const float* __restrict__ W,
                                                                               float* __restrict__ d, LongView lv) {
  const int lane = threadIdx.x & (kWave - 1);
  float s[K];
  if (blockIdx.x < lv.chunk_blocks) {
    const int64_t c = static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + (threadIdx.x >> 6);
    if (c >= lv.n_chunks) return;
    const int2 cd = lv.chunk_desc[c];
    const int64_t r = lv.long_rows[cd.x];
    const int64_t start = g.rowptr[r] + static_cast<int64_t>(cd.y) * kChunk;
    const int64_t row_end = g.rowptr[r + 1];
    rowsum_partial<K>(W, start, (start + kChunk < row_end) ? start + kChunk : row_end, lane, s);
    if (lane == 0) {
#pragma unroll
      for (int k = 0; k < K; ++k) lv.slab[c * K + k] = s[k];
    }
    return;
  }
  const int64_t r = static_cast<int64_t>(blockIdx.x - lv.chunk_blocks) * kWavesPerBlock + (threadIdx.x >> 6);
  if (r >= g.n_rows) return;
  const int64_t start = g.rowptr[r], end = g.rowptr[r + 1];
  if (end - start > kLongRow) return;
  rowsum_partial<K>(W, start, end, lane, s);
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < K; ++k) d[r * K + k] = inv_sqrt_or_zero(s[k]);
  }
}

template <int K>
__global__ void route_rowsum_finish_kernel(GraphView g, const int32_t* __restrict__ long_rows, const int32_t* __restrict__ long_base,
                                           int64_t n_long, const float* __restrict__ slab, float* __restrict__ d) {
  const int64_t t = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= n_long * K) return;
  const int64_t li = t / K;
  const int k = static_cast<int>(t % K);
  const int64_t r = long_rows[li];
  const int nc = static_cast<int>((g.rowptr[r + 1] - g.rowptr[r] + kChunk - 1) / kChunk);
  float s = 0.f;
  for (int c = 0; c < nc; ++c) s += slab[(static_cast<int64_t>(long_base[li]) + c) * K + k];
  d[r * K + k] = inv_sqrt_or_zero(s);
}

// Wt[j] = W[perm[j]]  (the weights in the entry order of the transposed matrix)
template <int K>
__global__ void route_permute_kernel(const float* __restrict__ W, const int32_t* __restrict__ perm, float* __restrict__ Wt,
                                     int64_t nnz) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= nnz) return;
  const int64_t src = perm[i];
#pragma unroll
  for (int k = 0; k < K; ++k) Wt[i * K + k] = W[src * K + k];
}

// ---- softmax over the stored entries of every CSR row (torch.sparse.softmax(adj, dim=1), kgat.py:96) and its backward ----
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int m = 1; m < kWave; m <<= 1) v = fmaxf(v, __shfl_xor(v, m));
  return v;
}

__global__ __launch_bounds__(kWavesPerBlock * kWave) void row_softmax_fwd_kernel(GraphView g, const float* __restrict__ logits,
                                                                                  float* __restrict__ a) {
  const int lane = threadIdx.x & (kWave - 1);
  const int64_t r = static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + (threadIdx.x >> 6);
  if (r >= g.n_rows) return;
  const int64_t start = g.rowptr[r], end = g.rowptr[r + 1];
  float mx = -INFINITY;
  for (int64_t j = start + lane; j < end; j += kWave) mx = fmaxf(mx, logits[j]);
  mx = wave_max(mx);
  float sum = 0.f;
  for (int64_t j = start + lane; j < end; j += kWave) sum += expf(logits[j] - mx);
  sum = lanes_sum<kWave>(sum);
  for (int64_t j = start + lane; j < end; j += kWave) a[j] = expf(logits[j] - mx) / sum;
}

// dlogit = a * (da - sum_row(a * da))
__global__ __launch_bounds__(kWavesPerBlock * kWave) void row_softmax_bwd_kernel(GraphView g, const float* __restrict__ a,
                                                                                  const float* __restrict__ da,
                                                                                  float* __restrict__ dlogits) {
  const int lane = threadIdx.x & (kWave - 1);
  const int64_t r = static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + (threadIdx.x >> 6);
  if (r >= g.n_rows) return;
  const int64_t start = g.rowptr[r], end = g.rowptr[r + 1];
  float dot = 0.f;
  for (int64_t j = start + lane; j < end; j += kWave) dot = fmaf(a[j], da[j], dot);
  dot = lanes_sum<kWave>(dot);
  for (int64_t j = start + lane; j < end; j += kWave) dlogits[j] = a[j] * (da[j] - dot);
}

// ---- per-slice row operations on [N, D] (thread per float4; a slice = SL consecutive lanes) -------------------
enum SliceOp { SLICE_SCALE = 0, SLICE_NORM = 1, SLICE_NORM_TANH = 2, SLICE_NORM_BWD = 3 };

template <int LPR, int K, int OP>
__global__ __launch_bounds__(256) void slice_kernel(const float* __restrict__ X, const float* __restrict__ P,
                                                    const float* __restrict__ dZ, float* __restrict__ Y, float* __restrict__ inv_out,
                                                    int64_t n_rows) {
  constexpr int SL = LPR / K;
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;   // float4 index
  const bool ok = i < n_rows * LPR;
  const int64_t r = ok ? i / LPR : 0;
  const int kf = static_cast<int>((i % LPR) / SL);
  const float4 x = ok ? reinterpret_cast<const float4*>(X)[i] : r4_zero();
  float4 y;
  if constexpr (OP == SLICE_SCALE) {
    const float p = ok ? P[r * K + kf] : 0.f;
    y = make_float4(x.x * p, x.y * p, x.z * p, x.w * p);
  } else if constexpr (OP == SLICE_NORM || OP == SLICE_NORM_TANH) {
    const float den = fmaxf(sqrtf(lanes_sum<SL>(r4_dot(x, x))), 1e-12f);
    y = make_float4(x.x / den, x.y / den, x.z / den, x.w / den);
    if constexpr (OP == SLICE_NORM_TANH) y = make_float4(tanhf(y.x), tanhf(y.y), tanhf(y.z), tanhf(y.w));
    if (inv_out && ok && (i % SL) == 0) inv_out[r * K + kf] = 1.0f / den;
  } else {
    // dX = inv * (dZ - z (z . dZ)), z = x * inv; the clamp is constant where ||x|| <= eps (inv == 1e12)
    const float inv = ok ? P[r * K + kf] : 0.f;
    const float4 dz = ok ? reinterpret_cast<const float4*>(dZ)[i] : r4_zero();
    const float4 z = make_float4(x.x * inv, x.y * inv, x.z * inv, x.w * inv);
    float dot = lanes_sum<SL>(r4_dot(z, dz));
    if (inv >= 1e12f) dot = 0.f;
    y = make_float4(inv * (dz.x - z.x * dot), inv * (dz.y - z.y * dot), inv * (dz.z - z.z * dot), inv * (dz.w - z.w * dot));
  }
  if (ok) reinterpret_cast<float4*>(Y)[i] = y;
}

}  // namespace tagrec

using namespace tagrec;

namespace {

bool route_shape_ok(int D, int K) {
  if (!(D == 16 || D == 32 || D == 64 || D == 128 || D == 256)) return false;
  if (!(K == 1 || K == 2 || K == 4 || K == 8)) return false;
  return (D / 4) % K == 0;
}

// Dispatch on (D, K) to a functor templated <LPR, K>.
template <typename F>
int route_dispatch(int D, int K, const char* who, F&& f) {
  if (!route_shape_ok(D, K))
    return fail(TAGREC_E_UNSUPPORTED, std::string(who) + ": needs D in {16,32,64,128,256}, K in {1,2,4,8}, D/K a multiple of 4 (got D=" +
                                          std::to_string(D) + ", K=" + std::to_string(K) + ")");
#define ROUTE_CASE(LPRV, KV) if (D == LPRV * 4 && K == KV) return f(std::integral_constant<int, LPRV>{}, std::integral_constant<int, KV>{});
  ROUTE_CASE(4, 1) ROUTE_CASE(4, 2) ROUTE_CASE(4, 4)
  ROUTE_CASE(8, 1) ROUTE_CASE(8, 2) ROUTE_CASE(8, 4) ROUTE_CASE(8, 8)
  ROUTE_CASE(16, 1) ROUTE_CASE(16, 2) ROUTE_CASE(16, 4) ROUTE_CASE(16, 8)
  ROUTE_CASE(32, 1) ROUTE_CASE(32, 2) ROUTE_CASE(32, 4) ROUTE_CASE(32, 8)
  ROUTE_CASE(64, 1) ROUTE_CASE(64, 2) ROUTE_CASE(64, 4) ROUTE_CASE(64, 8)
#undef ROUTE_CASE
  return fail(TAGREC_E_UNSUPPORTED, std::string(who) + ": unsupported shape");
}

template <typename F>
int k_dispatch(int K, const char* who, F&& f) {
  switch (K) {
    case 1: return f(std::integral_constant<int, 1>{});
    case 2: return f(std::integral_constant<int, 2>{});
    case 4: return f(std::integral_constant<int, 4>{});
    case 8: return f(std::integral_constant<int, 8>{});
    default: return fail(TAGREC_E_UNSUPPORTED, std::string(who) + ": K must be 1, 2, 4 or 8 (got " + std::to_string(K) + ")");
  }
}

int long_view(const tagrec_graph* g, int width, LongView* lv) {
  TAGREC_REQUIRE(!g->deferred, "routing kernels need a handle created with a host read (tagrec_graph_create / _ws)");
  *lv = LongView{g->long_rows, g->chunk_desc, g->n_chunks, nullptr, 0};
  if (g->n_long > 0) {
    int rc = ensure_slab(g, width);
    if (rc != TAGREC_OK) return rc;
    lv->slab = g->slab;
    lv->chunk_blocks = static_cast<unsigned>((g->n_chunks + kWavesPerBlock - 1) / kWavesPerBlock);
  }
  return TAGREC_OK;
}

}  // namespace

extern "C" int tagrec_route_softmax_f32(const float* logits, float* w, int64_t nnz, int K, void* stream) {
  TAGREC_REQUIRE(logits && w, "route_softmax: null pointer");
  TAGREC_REQUIRE(nnz >= 0, "route_softmax: negative nnz");
  if (nnz == 0) return TAGREC_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  return k_dispatch(K, "route_softmax", [&](auto kc) {
    constexpr int KK = decltype(kc)::value;
    route_softmax_kernel<KK><<<static_cast<unsigned>((nnz + 255) / 256), 256, 0, s>>>(logits, w, nnz);
    TAGREC_LAUNCH_CHECK();
    return TAGREC_OK;
  });
}

extern "C" int tagrec_route_rowsum_rsqrt_f32(const tagrec_graph* g, const float* w, int K, float* d, void* stream) {
  TAGREC_REQUIRE(g && d, "route_rowsum_rsqrt: null pointer");
  TAGREC_REQUIRE(g->nnz == 0 || w, "route_rowsum_rsqrt: null weights");
  if (g->n_rows == 0) return TAGREC_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  return k_dispatch(K, "route_rowsum_rsqrt", [&](auto kc) {
    constexpr int KK = decltype(kc)::value;
    LongView lv;
    int rc = long_view(g, KK, &lv);
    if (rc != TAGREC_OK) return rc;
    const GraphView gv{g->n_rows, g->rowptr, g->col, g->val, g->n_cols};
    const unsigned blocks = static_cast<unsigned>((g->n_rows + kWavesPerBlock - 1) / kWavesPerBlock);
    route_rowsum_kernel<KK><<<blocks + lv.chunk_blocks, kWavesPerBlock * kWave, 0, s>>>(gv, w, d, lv);
    TAGREC_LAUNCH_CHECK();
    if (g->n_long > 0) {
      const int64_t n = g->n_long * KK;
      route_rowsum_finish_kernel<KK><<<static_cast<unsigned>((n + 255) / 256), 256, 0, s>>>(gv, g->long_rows, g->long_base,
                                                                                            g->n_long, g->slab, d);
      TAGREC_LAUNCH_CHECK();
    }
    return TAGREC_OK;
  });
}

extern "C" int tagrec_route_permute_f32(const float* w, const int32_t* perm, float* wt, int64_t nnz, int K, void* stream) {
  TAGREC_REQUIRE(nnz == 0 || (w && perm && wt), "route_permute: null pointer");
  TAGREC_REQUIRE(w != wt, "route_permute: in-place permutation is not supported");
  if (nnz == 0) return TAGREC_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  return k_dispatch(K, "route_permute", [&](auto kc) {
    constexpr int KK = decltype(kc)::value;
    route_permute_kernel<KK><<<static_cast<unsigned>((nnz + 255) / 256), 256, 0, s>>>(w, perm, wt, nnz);
    TAGREC_LAUNCH_CHECK();
    return TAGREC_OK;
  });
}

extern "C" int tagrec_route_spmm_f32(const tagrec_graph* g, const float* W, int K, const float* X, const float* post,
                                     const float* self, const float* B, float b_scale, float* Y, float* Yn, float* inv,
                                     int D, void* stream) {
  return tagrec_route_spmm_ex_f32(g, W, K, X, post, self, B, b_scale, Y, Yn, inv, nullptr, nullptr, nullptr, D, stream);
}

extern "C" int tagrec_route_spmm_ex_f32(const tagrec_graph* g, const float* W, int K, const float* X, const float* post,
                                        const float* self, const float* B, float b_scale, float* Y, float* Yn, float* inv,
                                        const uint8_t* row_mask, const uint8_t* in_flags, const unsigned* in_count, int D,
                                        void* stream) {
  TAGREC_REQUIRE((in_flags == nullptr) == (in_count == nullptr), "route_spmm: in_flags and in_count go together");
  TAGREC_REQUIRE(g && X, "route_spmm: null graph or X");
  TAGREC_REQUIRE(g->nnz == 0 || W, "route_spmm: null weights");
  TAGREC_REQUIRE(Y || Yn, "route_spmm: no output requested");
  TAGREC_REQUIRE(static_cast<const void*>(X) != Y && static_cast<const void*>(X) != Yn, "route_spmm: output aliases the gathered input");
  TAGREC_REQUIRE(aligned16(X) && aligned16(Y) && aligned16(Yn) && aligned16(self) && aligned16(B), "route_spmm: rows must be 16-byte aligned");
  if (g->n_rows == 0) return TAGREC_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const RouteEpi e{Y, Yn, inv, post, self, B, b_scale, row_mask, in_flags, in_count};
  return route_dispatch(D, K, "route_spmm", [&](auto lc, auto kc) {
    constexpr int LPR = decltype(lc)::value, KK = decltype(kc)::value;
    LongView lv;
    int rc = long_view(g, LPR * 4, &lv);
    if (rc != TAGREC_OK) return rc;
    const GraphView gv{g->n_rows, g->rowptr, g->col, g->val, g->n_cols};
    const unsigned blocks = static_cast<unsigned>((g->n_rows + kWavesPerBlock - 1) / kWavesPerBlock);
    route_spmm_kernel<LPR, KK><<<blocks + lv.chunk_blocks, kWavesPerBlock * kWave, 0, s>>>(gv, W, X, e, lv);
    TAGREC_LAUNCH_CHECK();
    if (g->n_long > 0) {
      const unsigned fb = static_cast<unsigned>((g->n_long + kWavesPerBlock - 1) / kWavesPerBlock);
      route_spmm_finish_kernel<LPR, KK><<<fb, kWavesPerBlock * kWave, 0, s>>>(gv, g->long_rows, g->long_base, g->n_long, g->slab, e);
      TAGREC_LAUNCH_CHECK();
    }
    return TAGREC_OK;
  });
}

extern "C" int tagrec_route_score_f32(const tagrec_graph* g, const float* H, const float* T, float* logits, int K,
                                      int accumulate, int D, void* stream) {
  return tagrec_route_score_rows_f32(g, H, T, logits, K, accumulate, nullptr, D, stream);
}

extern "C" int tagrec_route_score_rows_f32(const tagrec_graph* g, const float* H, const float* T, float* logits, int K,
                                           int accumulate, const uint8_t* row_mask, int D, void* stream) {
  TAGREC_REQUIRE(g && H && T, "route_score: null pointer");
  TAGREC_REQUIRE(g->nnz == 0 || logits, "route_score: null logits");
  TAGREC_REQUIRE(aligned16(H) && aligned16(T), "route_score: rows must be 16-byte aligned");
  TAGREC_REQUIRE(!g->deferred, "route_score: needs a handle created with a host read (tagrec_graph_create / _ws)");
  if (g->n_rows == 0 || g->nnz == 0) return TAGREC_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  return route_dispatch(D, K, "route_score", [&](auto lc, auto kc) {
    constexpr int LPR = decltype(lc)::value, KK = decltype(kc)::value;
    LongView lv{g->long_rows, g->chunk_desc, g->n_chunks, nullptr,
                g->n_long > 0 ? static_cast<unsigned>((g->n_chunks + kWavesPerBlock - 1) / kWavesPerBlock) : 0u};
    const GraphView gv{g->n_rows, g->rowptr, g->col, g->val, g->n_cols};
    const unsigned blocks = static_cast<unsigned>((g->n_rows + kWavesPerBlock - 1) / kWavesPerBlock);
    route_score_kernel<LPR, KK><<<blocks + lv.chunk_blocks, kWavesPerBlock * kWave, 0, s>>>(gv, H, T, logits, accumulate, lv, row_mask);
    TAGREC_LAUNCH_CHECK();
    return TAGREC_OK;
  });
}

namespace {
template <int OP>
int launch_slice(const float* X, const float* P, const float* dZ, float* Y, float* inv_out, int64_t n_rows, int D, int K,
                 void* stream, const char* who) {
  TAGREC_REQUIRE(X && Y, std::string(who) + ": null pointer");
  TAGREC_REQUIRE(n_rows >= 0, std::string(who) + ": negative row count");
  TAGREC_REQUIRE(aligned16(X) && aligned16(Y) && aligned16(dZ), std::string(who) + ": rows must be 16-byte aligned");
  if (n_rows == 0) return TAGREC_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  return route_dispatch(D, K, who, [&](auto lc, auto kc) {
    constexpr int LPR = decltype(lc)::value, KK = decltype(kc)::value;
    const int64_t n4 = n_rows * LPR;
    slice_kernel<LPR, KK, OP><<<static_cast<unsigned>((n4 + 255) / 256), 256, 0, s>>>(X, P, dZ, Y, inv_out, n_rows);
    TAGREC_LAUNCH_CHECK();
    return TAGREC_OK;
  });
}
}  // namespace

extern "C" int tagrec_slice_scale_f32(const float* X, const float* scale, float* Y, int64_t n_rows, int D, int K, void* stream) {
  TAGREC_REQUIRE(scale, "slice_scale: null scale");
  return launch_slice<SLICE_SCALE>(X, scale, nullptr, Y, nullptr, n_rows, D, K, stream, "slice_scale");
}

extern "C" int tagrec_slice_norm_fwd_f32(const float* X, float* Y, float* inv, int64_t n_rows, int D, int K, int apply_tanh,
                                         void* stream) {
  if (apply_tanh) return launch_slice<SLICE_NORM_TANH>(X, nullptr, nullptr, Y, inv, n_rows, D, K, stream, "slice_norm_fwd");
  return launch_slice<SLICE_NORM>(X, nullptr, nullptr, Y, inv, n_rows, D, K, stream, "slice_norm_fwd");
}

extern "C" int tagrec_slice_norm_bwd_f32(const float* X_raw, const float* inv, const float* dZ, float* dX, int64_t n_rows, int D,
                                         int K, void* stream) {
  TAGREC_REQUIRE(inv && dZ, "slice_norm_bwd: null inv or dZ");
  return launch_slice<SLICE_NORM_BWD>(X_raw, inv, dZ, dX, nullptr, n_rows, D, K, stream, "slice_norm_bwd");
}

extern "C" int tagrec_row_softmax_fwd_f32(const tagrec_graph* g, const float* logits, float* a, void* stream) {
  TAGREC_REQUIRE(g, "row_softmax_fwd: null graph");
  TAGREC_REQUIRE(g->nnz == 0 || (logits && a), "row_softmax_fwd: null pointer");
  if (g->n_rows == 0 || g->nnz == 0) return TAGREC_OK;
  const GraphView gv{g->n_rows, g->rowptr, g->col, g->val, g->n_cols};
  const unsigned blocks = static_cast<unsigned>((g->n_rows + kWavesPerBlock - 1) / kWavesPerBlock);
  row_softmax_fwd_kernel<<<blocks, kWavesPerBlock * kWave, 0, static_cast<hipStream_t>(stream)>>>(gv, logits, a);
  TAGREC_LAUNCH_CHECK();
  return TAGREC_OK;
}

extern "C" int tagrec_row_softmax_bwd_f32(const tagrec_graph* g, const float* a, const float* da, float* dlogits, void* stream) {
  TAGREC_REQUIRE(g, "row_softmax_bwd: null graph");
  TAGREC_REQUIRE(g->nnz == 0 || (a && da && dlogits), "row_softmax_bwd: null pointer");
  if (g->n_rows == 0 || g->nnz == 0) return TAGREC_OK;
  const GraphView gv{g->n_rows, g->rowptr, g->col, g->val, g->n_cols};
  const unsigned blocks = static_cast<unsigned>((g->n_rows + kWavesPerBlock - 1) / kWavesPerBlock);
  row_softmax_bwd_kernel<<<blocks, kWavesPerBlock * kWave, 0, static_cast<hipStream_t>(stream)>>>(gv, a, da, dlogits);
  TAGREC_LAUNCH_CHECK();
  return TAGREC_OK;
}
