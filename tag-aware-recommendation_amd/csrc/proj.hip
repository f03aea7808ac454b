// K10: the tall-skinny dense products of the TGCN step on the exact-fp32 MFMA (v_mfma_f32_16x16x4_f32).
//
// `Attention1` (/root/reference/model/tgcn.py:20-37) multiplies every (node, neighbour) pair by W_1 / W_2; after the
// split of csrc/tgcn.hip what is left are products of a node table with a small matrix:
//     Q = X W2                    [n, D] x [D, A]            one per node type and layer, n up to 2 M
//     P = X[self rows] W1[:D] + b [m, D] x [D, 2A]           both neighbour types of a source type in one pass
//     dX += dQ W2^T               [n, A] x [A, D]            accumulated into the table gradient
//     dXs += [dP1 | dP2] [W1a[:D]^T ; W1b[:D]^T]             [m, 2A] x [2A, D]
//     dW = X^T dY, db = 1^T dY    contraction over the n rows
// n is in the millions and the small matrix has 16..128 rows / columns, so every one of them is HBM-bound (read the
// tall operand once, write the tall result once); the library GEMMs they replace picked wave-starved kernels for these
// shapes (0.5 TB/s for the weight gradient).
//
//   tall_mm_kernel<K, NO>: one wavefront per 16-row tile, grid-stride.  The product is computed TRANSPOSED (A operand = the
//     small matrix from LDS, B operand = the node rows), so `lane & 15` is the node and the four accumulator registers of a
//     lane are four consecutive output features: float4 loads of the node rows (lane (node, g) reads columns
//     16 j + 4 g .. + 3 for j = 0 .. K/16 - 1), float4 stores of the result.  The k axis is consumed in the order
//     (j, i, g) -> 16 j + 4 g + i; the small matrix is staged in LDS once per block in exactly that fragment order
//     (conflict-free ds_read_b32).  Optional: row gather of the tall operand, the k axis split over two tall operands,
//     the small matrix split over two pointers (along k or along the output), two output tensors, bias, accumulate.
//   tall_wgrad_kernel<KI, NO>: rows on the MFMA k axis straight from global memory (lane (m, q): row 4 s + q, column
//     16 i + m), every wave owns a strip of rows and a full [KI (+16), NO] accumulator set; the column sums of dY (the bias
//     gradient) ride along as one extra row block whose A operand is 1 on m == 0.  Per-wave partials are folded in wave
//     order by proj_fold_kernel (deterministic).
//   small_mm_kernel: C = op(A) op(B) for the (n_weight + 1) x dim_weight look-up tables (a few hundred flops).
#include "common.h"

namespace tagrec {
namespace {

typedef float pj4 __attribute__((ext_vector_type(4)));
constexpr int kProjThreads = 256;

struct SmallMat {        // W(k, c) = (k < ksplit ? p1 : p2 - ksplit rows)[k * sk + c * sc]   or split along c
  const float* p1; const float* p2;
  int64_t sk, sc;
  int split;             // 0: one matrix; 1: p2 holds the rows k >= K/2; 2: p2 holds the columns c >= NO/2
};

template <int K, int NO>
__device__ __forceinline__ float small_at(const SmallMat& w, int k, int c) {
  if (w.split == 1 && k >= K / 2) return w.p2[(k - K / 2) * w.sk + c * w.sc];
  if (w.split == 2 && c >= NO / 2) return w.p2[k * w.sk + (c - NO / 2) * w.sc];
  return w.p1[k * w.sk + c * w.sc];
}

struct TallArgs {
  const float* X1; const float* X2;      // X2 != nullptr: columns [K/2, K) of the tall operand (both with row stride K/2)
  const int64_t* sel;                    // optional row gather of X1 / X2
  const float* b1; const float* b2;      // bias (b2: the columns >= NO/2 when the output is split)
  float* Y1; float* Y2;                  // Y2 != nullptr: columns [NO/2, NO) go there (both with row stride NO/2)
  int accumulate;
  AdamRow adam;                          // adam.p != nullptr: the result (+ what Y1 holds, with `accumulate`) is the GRADIENT of the
                                         // table adam.p [n, NO]; its Adam update is applied here and nothing is stored to Y1
};

template <int K, int NO>
__global__ __launch_bounds__(kProjThreads) void tall_mm_kernel(TallArgs a, SmallMat w, int64_t n) {
  constexpr int KJ = K / 16, CB = NO / 16;
  __shared__ float Wl[K * NO];
  for (int e = threadIdx.x; e < K * NO; e += kProjThreads) {
    const int l = e & 63, cb = (e >> 6) % CB, ji = (e >> 6) / CB;
    const int j = ji >> 2, i = ji & 3;
    Wl[e] = small_at<K, NO>(w, 16 * j + 4 * (l >> 4) + i, 16 * cb + (l & 15));
  }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int node = lane & 15, g = lane >> 4;
  const int64_t tiles = (n + 15) / 16;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * (kProjThreads / 64);
  constexpr int XS = K;                         // row stride of a single tall operand
  for (int64_t tile = static_cast<int64_t>(blockIdx.x) * (kProjThreads / 64) + (threadIdx.x >> 6); tile < tiles; tile += stride) {
    const int64_t row = tile * 16 + node;
    const bool ok = row < n;
    const int64_t src = ok ? (a.sel ? a.sel[row] : row) : 0;
    pj4 x[KJ];
#pragma unroll
    for (int j = 0; j < KJ; ++j) {
      const int col = 16 * j + 4 * g;
      const float* base;
      if (a.X2) base = col < K / 2 ? a.X1 + src * (K / 2) + col : a.X2 + src * (K / 2) + (col - K / 2);
      else base = a.X1 + src * XS + col;
      x[j] = ok ? *reinterpret_cast<const pj4*>(base) : pj4{0.f, 0.f, 0.f, 0.f};
    }
    pj4 acc[CB];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) {
      const int c = 16 * cb + 4 * g;
      const float* b = a.b1;
      int cc = c;
      if (a.Y2 && c >= NO / 2) { b = a.b2; cc = c - NO / 2; }
      acc[cb] = b ? *reinterpret_cast<const pj4*>(b + cc) : pj4{0.f, 0.f, 0.f, 0.f};
    }
    // accumulate form: what the destination holds is fetched NOW, beside the operand rows, not behind the MFMA chain
    pj4 old[CB];
    if (a.accumulate) {
#pragma unroll
      for (int cb = 0; cb < CB; ++cb) {
        const int c = 16 * cb + 4 * g;
        const float* dst;
        if (a.Y2) dst = c < NO / 2 ? a.Y1 + row * (NO / 2) + c : a.Y2 + row * (NO / 2) + (c - NO / 2);
        else dst = a.Y1 + row * NO + c;
        old[cb] = ok ? *reinterpret_cast<const pj4*>(dst) : pj4{0.f, 0.f, 0.f, 0.f};
      }
    }
#pragma unroll
    for (int j = 0; j < KJ; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
          acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(Wl[((j * 4 + i) * CB + cb) * 64 + lane], x[j][i], acc[cb], 0, 0, 0);
    if (ok) {
#pragma unroll
      for (int cb = 0; cb < CB; ++cb) {
        const int c = 16 * cb + 4 * g;
        float* dst;
        if (a.Y2) dst = c < NO / 2 ? a.Y1 + row * (NO / 2) + c : a.Y2 + row * (NO / 2) + (c - NO / 2);
        else dst = a.Y1 + row * NO + c;
        pj4 o = acc[cb];
        if (a.accumulate) o += old[cb];
        if (a.adam.p) {                     // (wave-uniform) the table's update in place of the gradient store
          const int64_t off = row * NO + c;
          adam_f4 mi = *reinterpret_cast<const adam_f4*>(a.adam.m + off), vi = *reinterpret_cast<const adam_f4*>(a.adam.v + off);
          const adam_f4 pi = *reinterpret_cast<const adam_f4*>(a.adam.p + off);
          const adam_f4 qi = adam_update(mi, vi, pi, adam_f4{o[0], o[1], o[2], o[3]}, a.adam.w1, a.adam.b2, a.adam.w2, a.adam.step_size,
                                         a.adam.bc2_sqrt, a.adam.eps);
          *reinterpret_cast<adam_f4*>(a.adam.m + off) = mi;
          *reinterpret_cast<adam_f4*>(a.adam.v + off) = vi;
          *reinterpret_cast<adam_f4*>(a.adam.p + off) = qi;
        } else {
          *reinterpret_cast<pj4*>(dst) = o;
        }
      }
    }
  }
}

// dW [KI, NO] = X^T [dY1 | dY2], db [NO] = column sums of [dY1 | dY2]; partials per wave: [(KI + 16) * NO] (the first row of the
// extra block holds the column sums).
template <int KI, int NO>
__global__ __launch_bounds__(kProjThreads, 2) void tall_wgrad_kernel(const float* __restrict__ X, const float* __restrict__ dY1,
                                                                      const float* __restrict__ dY2, int64_t n, int64_t steps_per_wave,
                                                                      float* __restrict__ slab) {
  constexpr int IB = KI / 16, JB = NO / 16;
  const int lane = threadIdx.x & 63;
  const int m = lane & 15, q = lane >> 4;
  const int64_t wv = static_cast<int64_t>(blockIdx.x) * (kProjThreads / 64) + (threadIdx.x >> 6);
  pj4 acc[IB + 1][JB];
#pragma unroll
  for (int i = 0; i <= IB; ++i)
#pragma unroll
    for (int j = 0; j < JB; ++j) acc[i][j] = pj4{0.f, 0.f, 0.f, 0.f};
  const float one = m == 0 ? 1.f : 0.f;
  const int64_t s0 = wv * steps_per_wave;
  for (int64_t s = s0; s < s0 + steps_per_wave; ++s) {
    if (s * 4 >= n) break;
    const int64_t row = s * 4 + q;
    const bool ok = row < n;
    float xa[IB], yb[JB];
#pragma unroll
    for (int i = 0; i < IB; ++i) xa[i] = ok ? X[row * KI + i * 16 + m] : 0.f;
#pragma unroll
    for (int j = 0; j < JB; ++j) {
      const int c = j * 16 + m;
      float v = 0.f;
      if (ok) v = dY2 ? (c < NO / 2 ? dY1[row * (NO / 2) + c] : dY2[row * (NO / 2) + c - NO / 2]) : dY1[row * NO + c];
      yb[j] = v;
    }
#pragma unroll
    for (int j = 0; j < JB; ++j) {
#pragma unroll
      for (int i = 0; i < IB; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[i], yb[j], acc[i][j], 0, 0, 0);
      acc[IB][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(ok ? one : 0.f, yb[j], acc[IB][j], 0, 0, 0);
    }
  }
  float* o = slab + wv * (KI + 16) * NO;
#pragma unroll
  for (int i = 0; i <= IB; ++i)
#pragma unroll
    for (int j = 0; j < JB; ++j)
#pragma unroll
      for (int v = 0; v < 4; ++v) o[(i * 16 + q * 4 + v) * NO + j * 16 + m] = acc[i][j][v];
}

// fold per-wave partials in wave order: 16 threads per element take every 16th wave, then a fixed-order combine.
// Elements [0, KI*NO) -> dW (optionally added to what is there), the next NO -> db (the rest of the extra block is zero).
__global__ __launch_bounds__(1024) void proj_fold_kernel(const float* __restrict__ slab, int n_waves, int stride_elems, int w_elems,
                                                          int b_elems, float* __restrict__ dW, float* __restrict__ db,
                                                          int acc_w, int acc_b, int b_split, float* __restrict__ db2) {
  __shared__ float sh[16][64];
  const int e = blockIdx.x * 64 + (threadIdx.x & 63);
  const int part = threadIdx.x >> 6;
  const int total = w_elems + b_elems;
  float s = 0.f;
  if (e < total)
    for (int w = part; w < n_waves; w += 16) s += slab[static_cast<int64_t>(w) * stride_elems + e];
  sh[part][threadIdx.x & 63] = s;
  __syncthreads();
  if (part == 0 && e < total) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += sh[k][threadIdx.x];
    if (e < w_elems) {
      if (dW) dW[e] = acc_w ? dW[e] + t : t;
    } else {
      int c = e - w_elems;
      float* dst = db;
      if (db2 && c >= b_split) { dst = db2; c -= b_split; }
      if (dst) dst[c] = acc_b ? dst[c] + t : t;
    }
  }
}

__global__ void small_mm_kernel(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int M, int N, int K,
                                int64_t sam, int64_t sak, int64_t sbk, int64_t sbn, int accumulate) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= M * N) return;
  const int i = idx / N, j = idx % N;
  float s = 0.f;
  for (int k = 0; k < K; ++k) s = fmaf(A[i * sam + k * sak], B[k * sbk + j * sbn], s);
  C[idx] = accumulate ? C[idx] + s : s;
}

// rows of src added into dst at the listed positions (positions are DISTINCT: plain read-modify-write, no atomics)
__global__ __launch_bounds__(256) void row_add_at_kernel(float* __restrict__ dst, const int64_t* __restrict__ pos,
                                                          const float* __restrict__ src, int64_t n, int d4) {
  const int64_t total = n * d4;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * 256;
  for (int64_t e = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; e < total; e += stride) {
    const int64_t r = e / d4;
    const int c = static_cast<int>(e - r * d4);
    pj4* p = reinterpret_cast<pj4*>(dst) + pos[r] * d4 + c;
    *p = *p + reinterpret_cast<const pj4*>(src)[e];
  }
}

// column sums of dOut (.) [out > 0]: the bias gradient of a ReLU layer (tgcn.py:104-106, dbf of the fusion layer).
// 256 threads = (256 / (D/4)) row lanes x D/4 float4 columns; per-block partials [blocks, D] folded by proj_fold_kernel.
__global__ __launch_bounds__(256) void masked_colsum_kernel(const float* __restrict__ dOut, const float* __restrict__ outv, int64_t n,
                                                             int d4, float* __restrict__ slab) {
  __shared__ pj4 sh[256];
  const int rpp = 256 / d4;                             // rows per pass of the block
  const int c = threadIdx.x % d4, ty = threadIdx.x / d4;
  pj4 acc = {0.f, 0.f, 0.f, 0.f};
  if (ty < rpp)
    for (int64_t r = static_cast<int64_t>(blockIdx.x) * rpp + ty; r < n; r += static_cast<int64_t>(gridDim.x) * rpp) {
      const pj4 g = __builtin_nontemporal_load(reinterpret_cast<const pj4*>(dOut) + r * d4 + c);
      const pj4 o = __builtin_nontemporal_load(reinterpret_cast<const pj4*>(outv) + r * d4 + c);
      acc.x += o.x > 0.f ? g.x : 0.f; acc.y += o.y > 0.f ? g.y : 0.f; acc.z += o.z > 0.f ? g.z : 0.f; acc.w += o.w > 0.f ? g.w : 0.f;
    }
  sh[threadIdx.x] = acc;
  __syncthreads();
  if (ty == 0) {
    pj4 t = sh[c];
    for (int y = 1; y < rpp; ++y) t += sh[y * d4 + c];                 // fixed order
    reinterpret_cast<pj4*>(slab)[static_cast<int64_t>(blockIdx.x) * d4 + c] = t;
  }
}

constexpr int kColsumBlocks = 1024;
constexpr int kWgradWaveCap = 2048;

int64_t wgrad_waves(int64_t n, int64_t* steps_per_wave) {
  const int64_t steps = (n + 3) / 4;
  int64_t spw = (steps + kWgradWaveCap - 1) / kWgradWaveCap;
  if (spw < 16) spw = 16;
  *steps_per_wave = spw;
  int64_t waves = (steps + spw - 1) / spw;
  const int per_block = kProjThreads / 64;
  return (waves + per_block - 1) / per_block * per_block;
}

template <int K, int NO>
int launch_tall(const TallArgs& a, const SmallMat& w, int64_t n, hipStream_t s) {
  const int64_t tiles = (n + 15) / 16;
  int64_t blocks = (tiles + 3) / 4;
  if (blocks > 256 * 5) blocks = 256 * 5;
  tall_mm_kernel<K, NO><<<static_cast<unsigned>(blocks), kProjThreads, 0, s>>>(a, w, n);
  TAGREC_LAUNCH_CHECK();
  return TAGREC_OK;
}

template <int KI, int NO>
int launch_wgrad(const float* X, const float* dY1, const float* dY2, int64_t n, float* slab, hipStream_t s, int64_t* n_waves) {
  int64_t spw;
  *n_waves = wgrad_waves(n, &spw);
  tall_wgrad_kernel<KI, NO><<<static_cast<unsigned>(*n_waves / (kProjThreads / 64)), kProjThreads, 0, s>>>(X, dY1, dY2, n, spw, slab);
  TAGREC_LAUNCH_CHECK();
  return TAGREC_OK;
}

bool dim_ok(int d) { return d == 16 || d == 32 || d == 64 || d == 128; }

}  // namespace
}  // namespace tagrec

using namespace tagrec;

#define PROJ_DISPATCH(K, NO, CALL)                                           \
  do {                                                                       \
    switch ((K) * 1000 + (NO)) {                                             \
      case 16016: return CALL(16, 16);   case 16032: return CALL(16, 32);    \
      case 16064: return CALL(16, 64);   case 16128: return CALL(16, 128);   \
      case 32016: return CALL(32, 16);   case 32032: return CALL(32, 32);    \
      case 32064: return CALL(32, 64);   case 32128: return CALL(32, 128);   \
      case 64016: return CALL(64, 16);   case 64032: return CALL(64, 32);    \
      case 64064: return CALL(64, 64);   case 64128: return CALL(64, 128);   \
      case 128016: return CALL(128, 16); case 128032: return CALL(128, 32);  \
      case 128064: return CALL(128, 64); case 128128: return CALL(128, 128); \
      default: break;                                                        \
    }                                                                        \
  } while (0)

extern "C" int tagrec_tall_mm_f32(const float* X1, const float* X2, const int64_t* sel, int64_t n, int K, int NO, const float* W1,
                                  const float* W2, int64_t w_sk, int64_t w_sc, int w_split, const float* b1, const float* b2,
                                  float* Y1, float* Y2, int accumulate, void* stream) {
  TAGREC_REQUIRE(n >= 0 && dim_ok(K) && dim_ok(NO), "tall_mm: K and NO must be 16, 32, 64 or 128");
  if (n == 0) return TAGREC_OK;                                   // (empty tensors carry null pointers)
  TAGREC_REQUIRE(X1 && W1 && Y1, "tall_mm: null pointer");
  TAGREC_REQUIRE(w_split >= 0 && w_split <= 2 && (w_split == 0) == (W2 == nullptr), "tall_mm: W2 goes with w_split 1 (rows) or 2 (columns)");
  TAGREC_REQUIRE(!(X2 && K < 32) && !(Y2 && NO < 32), "tall_mm: a split operand needs at least 32 columns");
  TAGREC_REQUIRE(aligned16(X1) && aligned16(Y1) && (!X2 || aligned16(X2)) && (!Y2 || aligned16(Y2)) && (!b1 || aligned16(b1)) &&
                     (!b2 || aligned16(b2)),
                 "tall_mm: 16-byte aligned rows expected");
  TAGREC_REQUIRE(!b2 || Y2, "tall_mm: a second bias goes with a second output");
  if (n == 0) return TAGREC_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const TallArgs a{X1, X2, sel, b1, b2, Y1, Y2, accumulate, AdamRow{nullptr, nullptr, nullptr, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}};
  const SmallMat w{W1, W2, w_sk, w_sc, w_split};
#define CALL(KK, NN) launch_tall<KK, NN>(a, w, n, s)
  PROJ_DISPATCH(K, NO, CALL);
#undef CALL
  return fail(TAGREC_E_UNSUPPORTED, "tall_mm: shape not covered");
}

// dX = G_in + X W with the Adam update of the table `p` [n, NO] in the epilogue: the gradient row is consumed where its last
// term is formed (no gradient tensor is written, the optimizer does not read one).  G_in may be NULL (no earlier terms).
extern "C" int tagrec_tall_mm_adam_f32(const float* X, int64_t n, int K, int NO, const float* W, int64_t w_sk, int64_t w_sc,
                                       const float* G_in, float* p, float* m, float* v, float lr, float b1, float b2, float eps,
                                       int64_t step, void* stream) {
  TAGREC_REQUIRE(n >= 0 && dim_ok(K) && dim_ok(NO), "tall_mm_adam: K and NO must be 16, 32, 64 or 128");
  TAGREC_REQUIRE(step >= 1, "tall_mm_adam: bad step");
  if (n == 0) return TAGREC_OK;
  TAGREC_REQUIRE(X && W && p && m && v, "tall_mm_adam: null pointer");
  TAGREC_REQUIRE(aligned16(X) && aligned16(p) && aligned16(m) && aligned16(v) && (!G_in || aligned16(G_in)),
                 "tall_mm_adam: 16-byte aligned rows expected");
  hipStream_t s = static_cast<hipStream_t>(stream);
  // same host arithmetic as tagrec_adam_f32 (torch's _single_tensor_adam: python floats = doubles)
  const double bc1 = 1.0 - pow(static_cast<double>(b1), static_cast<double>(step));
  const double bc2 = 1.0 - pow(static_cast<double>(b2), static_cast<double>(step));
  AdamRow ad{p, m, v, static_cast<float>(1.0 - static_cast<double>(b1)), b2, static_cast<float>(1.0 - static_cast<double>(b2)),
             static_cast<float>(static_cast<double>(lr) / bc1), static_cast<float>(sqrt(bc2)), eps};
  // Y1 = G_in: read as the accumulate source, never written (the Adam branch stores to p / m / v instead)
  const TallArgs a{X, nullptr, nullptr, nullptr, nullptr, const_cast<float*>(G_in), nullptr, G_in ? 1 : 0, ad};
  const SmallMat w{W, nullptr, w_sk, w_sc, 0};
#define CALL(KK, NN) launch_tall<KK, NN>(a, w, n, s)
  PROJ_DISPATCH(K, NO, CALL);
#undef CALL
  return fail(TAGREC_E_UNSUPPORTED, "tall_mm_adam: shape not covered");
}

extern "C" int64_t tagrec_tall_wgrad_workspace(int KI, int NO) {
  return static_cast<int64_t>(kWgradWaveCap + kProjThreads / 64) * (KI + 16) * NO;
}

extern "C" int tagrec_tall_wgrad_f32(const float* X, const float* dY1, const float* dY2, int64_t n, int KI, int NO, float* dW,
                                     float* db1, float* db2, int acc_w, int acc_b, float* workspace,
                                     int64_t workspace_floats, void* stream) {
  TAGREC_REQUIRE(n >= 0 && dim_ok(KI) && dim_ok(NO), "tall_wgrad: KI and NO must be 16, 32, 64 or 128");
  TAGREC_REQUIRE(workspace && (n == 0 || (X && dY1)), "tall_wgrad: null pointer");       // (empty tensors carry null pointers)
  TAGREC_REQUIRE(!(dY2 && NO < 32), "tall_wgrad: a split dY needs at least 32 columns");
  TAGREC_REQUIRE(db2 == nullptr || dY2 != nullptr, "tall_wgrad: a second bias gradient goes with a second dY");
  TAGREC_REQUIRE(workspace_floats >= tagrec_tall_wgrad_workspace(KI, NO), "tall_wgrad: workspace too small");
  hipStream_t s = static_cast<hipStream_t>(stream);
  int64_t n_waves = 0;
  if (n > 0) {
#define CALL(KK, NN) launch_wgrad<KK, NN>(X, dY1, dY2, n, workspace, s, &n_waves)
    int rc = TAGREC_E_UNSUPPORTED;
    do {
      switch (KI * 1000 + NO) {
        case 16016: rc = CALL(16, 16); break;   case 16032: rc = CALL(16, 32); break;
        case 16064: rc = CALL(16, 64); break;   case 16128: rc = CALL(16, 128); break;
        case 32016: rc = CALL(32, 16); break;   case 32032: rc = CALL(32, 32); break;
        case 32064: rc = CALL(32, 64); break;   case 32128: rc = CALL(32, 128); break;
        case 64016: rc = CALL(64, 16); break;   case 64032: rc = CALL(64, 32); break;
        case 64064: rc = CALL(64, 64); break;   case 64128: rc = CALL(64, 128); break;
        case 128016: rc = CALL(128, 16); break; case 128032: rc = CALL(128, 32); break;
        case 128064: rc = CALL(128, 64); break; case 128128: rc = CALL(128, 128); break;
        default: break;
      }
    } while (0);
#undef CALL
    if (rc != TAGREC_OK) return rc == TAGREC_E_UNSUPPORTED ? fail(rc, "tall_wgrad: shape not covered") : rc;
  }
  // dW receives the whole [KI, NO] block (with a split dY its columns [NO/2, NO) belong to dY2); db1 / db2 the two halves of
  // the column sums.  n == 0: zeros (or nothing, when accumulating).
  const int w_elems = KI * NO, b_elems = NO;
  const int total = w_elems + b_elems;
  proj_fold_kernel<<<(total + 63) / 64, 1024, 0, s>>>(workspace, static_cast<int>(n_waves), (KI + 16) * NO, w_elems, b_elems, dW, db1,
                                                       acc_w, acc_b, dY2 ? NO / 2 : NO, db2);
  TAGREC_LAUNCH_CHECK();
  return TAGREC_OK;
}

extern "C" int tagrec_small_mm_f32(const float* A, const float* B, float* C, int M, int N, int K, int64_t sam, int64_t sak,
                                   int64_t sbk, int64_t sbn, int accumulate, void* stream) {
  TAGREC_REQUIRE(A && B && C, "small_mm: null pointer");
  TAGREC_REQUIRE(M >= 0 && N >= 0 && K >= 0 && static_cast<int64_t>(M) * N <= (1 << 22) && K <= 4096, "small_mm: meant for small tables");
  if (M * N == 0) return TAGREC_OK;
  small_mm_kernel<<<(M * N + 255) / 256, 256, 0, static_cast<hipStream_t>(stream)>>>(A, B, C, M, N, K, sam, sak, sbk, sbn, accumulate);
  TAGREC_LAUNCH_CHECK();
  return TAGREC_OK;
}

extern "C" int tagrec_row_add_at_f32(float* dst, const int64_t* pos, const float* src, int64_t n_rows, int D, void* stream) {
  TAGREC_REQUIRE(dst && pos && src, "row_add_at: null pointer");
  TAGREC_REQUIRE(n_rows >= 0 && D >= 4 && D % 4 == 0 && aligned16(dst) && aligned16(src), "row_add_at: need D % 4 == 0, 16-byte aligned rows");
  if (n_rows == 0) return TAGREC_OK;
  int64_t blocks = (n_rows * (D / 4) + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;
  row_add_at_kernel<<<static_cast<unsigned>(blocks), 256, 0, static_cast<hipStream_t>(stream)>>>(dst, pos, src, n_rows, D / 4);
  TAGREC_LAUNCH_CHECK();
  return TAGREC_OK;
}

extern "C" int64_t tagrec_masked_colsum_workspace(int D) { return static_cast<int64_t>(kColsumBlocks) * D; }

extern "C" int tagrec_masked_colsum_f32(const float* dOut, const float* out, int64_t n_rows, int D, float* result, float* workspace,
                                        int64_t workspace_floats, void* stream) {
  TAGREC_REQUIRE(n_rows >= 0 && D >= 4 && D % 4 == 0 && D <= 1024 && 256 % (D / 4) == 0, "masked_colsum: D / 4 must divide 256");
  TAGREC_REQUIRE(result && workspace && (n_rows == 0 || (dOut && out)), "masked_colsum: null pointer");
  TAGREC_REQUIRE(workspace_floats >= tagrec_masked_colsum_workspace(D), "masked_colsum: workspace too small");
  TAGREC_REQUIRE(aligned16(dOut) && aligned16(out) && aligned16(workspace), "masked_colsum: 16-byte aligned rows expected");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int rpp = 256 / (D / 4);
  int64_t blocks = (n_rows + rpp - 1) / rpp;
  if (blocks > kColsumBlocks) blocks = kColsumBlocks;
  if (blocks > 0) {
    masked_colsum_kernel<<<static_cast<unsigned>(blocks), 256, 0, s>>>(dOut, out, n_rows, D / 4, workspace);
    TAGREC_LAUNCH_CHECK();
  }
  proj_fold_kernel<<<(D + 63) / 64, 1024, 0, s>>>(workspace, static_cast<int>(blocks), D, D, 0, result, nullptr, 0, 0, D, nullptr);
  TAGREC_LAUNCH_CHECK();
  return TAGREC_OK;
}
