// K1: CSR sparse-adjacency x dense-embedding product for gfx950, with the fused row
// epilogues of the LightGCN / NGCF layers.
//
// Replaces torch.sparse.mm inside `split_mm` (/root/reference/model/help/adj.py:158-167) and,
// fused behind it, F.normalize + the layer mean of `LightGCN.forward`
// (/root/reference/model/lightgcn.py:54-60) and their autograd backward.
//
// Work decomposition (wave64):
//   * one wavefront per CSR row.  A row of D fp32 = D/4 lanes x 16 B, so one
//     global_load_dwordx4 wave-instruction gathers 64/(D/4) neighbour rows at once
//     (D=64: 4 neighbours, 1 KiB per instruction), 4 instructions in flight per wave.
//   * column indices / values are read 64 at a time, coalesced, and broadcast to the
//     lane groups with ds_bpermute (__shfl).
//   * per-lane-group partial sums are folded with cross-lane xor shuffles; the row norm
//     is reduced the same way, so the epilogue needs no LDS and no second pass.
//   * rows longer than kLongRow entries are cut into kChunk-entry chunks, one wave each,
//     written to a partial-sum slab and folded in chunk order by a finishing kernel
//     (deterministic; no float atomics).
// HBM-bound: algorithmic bytes per stored entry = 8 + 4*D.
#include <cmath>
#include <new>

#include "common.h"
#include "graph.h"

namespace tagrec {

constexpr int kMaxGenericBlocks = 8;  // scalar kernel keeps D <= 512 in registers

enum Epi { EPI_NONE = 0, EPI_NORM_ACC = 1, EPI_NORMBWD = 2, EPI_AXPY = 3, EPI_SS = 4, EPI_NORMBWD_DOT = 5 };

struct EpiArgs {
  float* Y;               // [n_rows, D] product (or gradient) out
  float* inv_norm;        // NORM_ACC: out; NORMBWD / NORMBWD_DOT: in; SS: out = this shard's sum of squares per row
  float* accum;           // NORM_ACC: accum += s * normalize(y)
  const float* Xraw;      // NORMBWD
  const float* B;         // NORMBWD / NORMBWD_DOT: dZ ; AXPY: B
  const float* dot;       // NORMBWD_DOT: z . (s dZ) per row, summed over ALL column shards by the caller
  float s;
  DropMask drop;          // NORM_ACC: message dropout of the product before it is stored / normalised (lightgcn.py:56);
                          // NORMBWD: the same mask on the gradient leaving this layer.  p = 0: off
  // Row-sparse operand (the gradient at the start of the backward chain is non-zero on the <= 3 B batch rows only, and
  // one hop later on their neighbours): in_flags[c] != 0 iff row c of the gathered matrix has a non-zero; rows flagged
  // zero are not fetched (a x 0 adds exactly 0, so the result is unchanged).  *in_count = number of flagged rows; the
  // flags are consulted only while they cover less than 4/5 of the rows (each check is an extra L2 read per stored entry).
  // out_flags: the same for this product's output.
  const uint8_t* in_flags;
  const unsigned* in_count;
  uint8_t* out_flags;
  // Output rows to compute (NORM_ACC): row r is skipped entirely when row_mask[r] == 0 -- its outputs are neither
  // written nor accumulated.  nullptr = every row.  The last forward layer is needed on the batch rows only and the one
  // before it on their neighbours (lightgcn.py: the loss reads `out` at the batch rows).
  const uint8_t* row_mask;
  // NORMBWD / AXPY: b_flags[r] == 0 promises that row r of B (dZ) is zero, so the row's epilogue term vanishes and
  // neither B nor X_raw is read for it (the BPR gradient w.r.t. the layer mean lives on the <= 3 B batch rows).
  // nullptr = read every row.
  const uint8_t* b_flags;
  // AXPY: when adam.p is set the row's result is the gradient of parameter row r and is consumed on the spot -- the
  // Adam update of that row (torch's operation order, common.h) replaces the store to Y, which may then be null.  Saves
  // the gradient's round trip through memory and the optimizer's own launch (the last hop of a LightGCN step).
  AdamRow adam = AdamRow{nullptr, nullptr, nullptr, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
};

// Streamed-once data (indices, values, epilogue operands, outputs) is moved with non-temporal accesses so it does
// not displace the gathered embedding rows -- the only data with reuse -- from L2 / Infinity Cache.
typedef float nt_f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld_stream(const float4* p) {
  const nt_f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const nt_f32x4*>(p));
  return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void st_stream(float4* p, const float4& a) {
  __builtin_nontemporal_store(nt_f32x4{a.x, a.y, a.z, a.w}, reinterpret_cast<nt_f32x4*>(p));
}
template <typename T>
__device__ __forceinline__ T ld_stream(const T* p) { return __builtin_nontemporal_load(p); }

__device__ __forceinline__ float4 f4_zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ void f4_fma(float4& a, float s, const float4& x) {
  a.x = fmaf(s, x.x, a.x); a.y = fmaf(s, x.y, a.y); a.z = fmaf(s, x.z, a.z); a.w = fmaf(s, x.w, a.w);
}
__device__ __forceinline__ float4 f4_shfl_xor(const float4& a, int m) {
  return make_float4(__shfl_xor(a.x, m), __shfl_xor(a.y, m), __shfl_xor(a.z, m), __shfl_xor(a.w, m));
}
__device__ __forceinline__ float f4_dot(const float4& a, const float4& b) {
  return fmaf(a.x, b.x, fmaf(a.y, b.y, fmaf(a.z, b.z, a.w * b.w)));
}

// Sum of val[j] * X[col[j], :] over j in [start, end).  LPR = lanes per row = D/4.  On return every
// lane holds the full sum for its float4 column (lane % LPR).
template <int LPR, bool FLAGS = false>
__device__ __forceinline__ float4 gather_rows(const GraphView& g, const float* __restrict__ X,
                                              int64_t start, int64_t end, int lane, const uint8_t* __restrict__ flags = nullptr) {
  constexpr int NPI = kWave / LPR;  // neighbour rows per wave-instruction
  const int q = lane / LPR;
  const float4* __restrict__ Xv = reinterpret_cast<const float4*>(X) + (lane % LPR);
  float4 acc = f4_zero();
  for (int64_t base = start; base < end; base += kWave) {
    int n = (end - base) < kWave ? static_cast<int>(end - base) : kWave;
    int my_col = 0;
    float my_val = 0.f;
    if (lane < n) {
      my_col = ld_stream(g.col + base + lane);
      if constexpr (FLAGS) {
        if (flags[my_col]) my_val = ld_stream(g.val + base + lane);    // a zero weight marks "row not needed"
      } else {
        my_val = ld_stream(g.val + base + lane);
      }
    }
    if constexpr (FLAGS) {
      // Keep only the entries whose operand row is flagged: they move to the first lanes (order preserved) and the
      // gather loop below runs over them alone -- a batch of 64 entries with nothing flagged costs its index / flag reads
      // and nothing else.  (dest is a permutation of the 64 lanes: flagged entries first, the rest behind them.)
      const unsigned long long m = __ballot(my_val != 0.f);
      const int cnt = __popcll(m);
      if (cnt == 0) continue;
      if (cnt < n) {
        const int before = __builtin_amdgcn_mbcnt_hi(static_cast<unsigned>(m >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<unsigned>(m), 0u));
        const int dest = (my_val != 0.f) ? before : cnt + (lane - before);
        my_col = __builtin_amdgcn_ds_permute(dest << 2, my_col);
        my_val = __int_as_float(__builtin_amdgcn_ds_permute(dest << 2, __float_as_int(my_val)));
        n = cnt;
      }
    }
    const int groups = (n + NPI - 1) / NPI;
    for (int gi = 0; gi < groups; gi += 4) {
      float4 x[4];
      float v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int j = (gi + u) * NPI + q;
        const int c = __shfl(my_col, j & (kWave - 1));
        const float w = __shfl(my_val, j & (kWave - 1));
        const bool ok = FLAGS ? (j < n && w != 0.f) : (j < n);
        v[u] = ok ? w : 0.f;
        x[u] = ok ? Xv[static_cast<int64_t>(c) * LPR] : f4_zero();
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) f4_fma(acc, v[u], x[u]);
    }
  }
#pragma unroll
  for (int m = LPR; m < kWave; m <<= 1) {
    const float4 o = f4_shfl_xor(acc, m);
    acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w;
  }
  return acc;
}

template <int LPR>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int m = 1; m < LPR; m <<= 1) v += __shfl_xor(v, m);
  return v;
}

// Gradient of z = x / max(||x||, eps) given inv = 1/max(||x||, eps):
//   ||x|| >  eps : inv * (dz - z (z . dz))
//   ||x|| <= eps : inv * dz            (the clamp is constant there; inv == 1e12)
template <int LPR>
__device__ __forceinline__ float4 normalize_bwd(const float4& xr, float inv, const float4& dz) {
  const float4 z = make_float4(xr.x * inv, xr.y * inv, xr.z * inv, xr.w * inv);
  float dot = group_sum<LPR>(f4_dot(z, dz));
  if (inv >= 1e12f) dot = 0.f;
  return make_float4(inv * (dz.x - z.x * dot), inv * (dz.y - z.y * dot), inv * (dz.z - z.z * dot),
                     inv * (dz.w - z.w * dot));
}

template <int LPR, int EPI>
__device__ __forceinline__ void row_epilogue(float4 acc, int64_t r, int lane, const EpiArgs& e) {
  const bool writer = lane < LPR;
  const int64_t off = r * LPR + (lane % LPR);  // float4 index of this lane's columns
  if constexpr (EPI == EPI_NONE) {
    if (writer) st_stream(reinterpret_cast<float4*>(e.Y) + off, acc);
    if (e.out_flags) {
      const float nz = group_sum<LPR>((acc.x != 0.f || acc.y != 0.f || acc.z != 0.f || acc.w != 0.f) ? 1.f : 0.f);
      if (lane == 0) e.out_flags[r] = nz != 0.f;
    }
  } else if constexpr (EPI == EPI_NORM_ACC) {
    drop4(e.drop, off, acc.x, acc.y, acc.z, acc.w);
    const float ss = group_sum<LPR>(f4_dot(acc, acc));
    const float den = fmaxf(sqrtf(ss), 1e-12f);
    if (writer) {
      st_stream(reinterpret_cast<float4*>(e.Y) + off, acc);
      if (e.accum) {                      // nullptr: the caller forms the layer mean itself, on the rows it needs
        float4 a = ld_stream(reinterpret_cast<const float4*>(e.accum) + off);
        a.x = fmaf(e.s, acc.x / den, a.x);
        a.y = fmaf(e.s, acc.y / den, a.y);
        a.z = fmaf(e.s, acc.z / den, a.z);
        a.w = fmaf(e.s, acc.w / den, a.w);
        st_stream(reinterpret_cast<float4*>(e.accum) + off, a);
      }
    }
    if (lane == 0) e.inv_norm[r] = 1.0f / den;
  } else if constexpr (EPI == EPI_NORMBWD) {
    float4 o = acc;
    if (!e.b_flags || e.b_flags[r]) {     // (r is the same for every lane of the wave)
      const float4 xr = ld_stream(reinterpret_cast<const float4*>(e.Xraw) + off);
      float4 dz = ld_stream(reinterpret_cast<const float4*>(e.B) + off);
      dz.x *= e.s; dz.y *= e.s; dz.z *= e.s; dz.w *= e.s;
      const float4 gz = normalize_bwd<LPR>(xr, e.inv_norm[r], dz);
      o = make_float4(acc.x + gz.x, acc.y + gz.y, acc.z + gz.z, acc.w + gz.w);
    }
    drop4(e.drop, off, o.x, o.y, o.z, o.w);
    if (writer) st_stream(reinterpret_cast<float4*>(e.Y) + off, o);
    if (e.out_flags) {
      const float nz = group_sum<LPR>((o.x != 0.f || o.y != 0.f || o.z != 0.f || o.w != 0.f) ? 1.f : 0.f);
      if (lane == 0) e.out_flags[r] = nz != 0.f;
    }
  } else if constexpr (EPI == EPI_AXPY) {
    float4 o = acc;
    if (!(e.b_flags && !e.b_flags[r])) {
      const float4 b = ld_stream(reinterpret_cast<const float4*>(e.B) + off);
      o = make_float4(fmaf(e.s, b.x, acc.x), fmaf(e.s, b.y, acc.y), fmaf(e.s, b.z, acc.z), fmaf(e.s, b.w, acc.w));
    }
    if (e.adam.p) {
      if (writer) {
        adam_f4 m = __builtin_nontemporal_load(reinterpret_cast<const adam_f4*>(e.adam.m) + off);
        adam_f4 v = __builtin_nontemporal_load(reinterpret_cast<const adam_f4*>(e.adam.v) + off);
        const adam_f4 p = __builtin_nontemporal_load(reinterpret_cast<const adam_f4*>(e.adam.p) + off);
        const float step_size = e.adam.coef ? e.adam.coef[0] : e.adam.step_size;
        const float bc2_sqrt = e.adam.coef ? e.adam.coef[1] : e.adam.bc2_sqrt;
        const adam_f4 q = adam_update(m, v, p, adam_f4{o.x, o.y, o.z, o.w}, e.adam.w1, e.adam.b2, e.adam.w2, step_size, bc2_sqrt,
                                      e.adam.eps);
        __builtin_nontemporal_store(m, reinterpret_cast<adam_f4*>(e.adam.m) + off);
        __builtin_nontemporal_store(v, reinterpret_cast<adam_f4*>(e.adam.v) + off);
        __builtin_nontemporal_store(q, reinterpret_cast<adam_f4*>(e.adam.p) + off);
      }
    } else if (writer) {
      st_stream(reinterpret_cast<float4*>(e.Y) + off, o);
    }
  } else if constexpr (EPI == EPI_SS) {
    // column-sharded tables: the row norm needs every shard's columns, so only the local sum of squares is formed
    const float ss = group_sum<LPR>(f4_dot(acc, acc));
    if (writer) st_stream(reinterpret_cast<float4*>(e.Y) + off, acc);
    if (lane == 0) e.inv_norm[r] = ss;
  } else {  // EPI_NORMBWD_DOT: normalize-backward with the row dot product supplied (already summed over shards)
    const float4 xr = ld_stream(reinterpret_cast<const float4*>(e.Xraw) + off);
    const float4 dz = ld_stream(reinterpret_cast<const float4*>(e.B) + off);
    const float inv = e.inv_norm[r];
    const float dot = inv >= 1e12f ? 0.f : e.dot[r];
    const float4 o = make_float4(acc.x + inv * (e.s * dz.x - xr.x * inv * dot), acc.y + inv * (e.s * dz.y - xr.y * inv * dot),
                                 acc.z + inv * (e.s * dz.z - xr.z * inv * dot), acc.w + inv * (e.s * dz.w - xr.w * inv * dot));
    if (writer) st_stream(reinterpret_cast<float4*>(e.Y) + off, o);
    if (e.out_flags) {
      const float nz = group_sum<LPR>((o.x != 0.f || o.y != 0.f || o.z != 0.f || o.w != 0.f) ? 1.f : 0.f);
      if (lane == 0) e.out_flags[r] = nz != 0.f;
    }
  }
}

// ---- main kernel: one wave per short row; the FIRST blocks of the grid take the long-row chunks ----
// (one wave per kChunk entries, partial sums to a slab) so the heavy items start first and the same
// launch covers every stored entry of the matrix.
// MASKED = e.row_mask given (a separate instantiation, so that profiles tell the all-rows launches from the
// restricted ones by name)
template <int LPR, int EPI, bool MASKED = false>
__global__ __launch_bounds__(kWavesPerBlock * kWave) void spmm_rows_kernel(GraphView g, const float* __restrict__ X,
                                                                            EpiArgs e, LongView lv) {
  const int lane = threadIdx.x & (kWave - 1);
  if (blockIdx.x < lv.chunk_blocks) {
    const int64_t c = static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + (threadIdx.x >> 6);
    if (c >= lv.n_chunks) return;
    const int2 d = lv.chunk_desc[c];
    if (d.x < 0) return;                    // unused slot of a handle created without a host read
    const int64_t r = lv.long_rows[d.x];
    if (MASKED && !e.row_mask[r]) return;
    const int64_t start = g.rowptr[r] + static_cast<int64_t>(d.y) * kChunk;
    const int64_t row_end = g.rowptr[r + 1];
    const int64_t end = (start + kChunk < row_end) ? start + kChunk : row_end;
    const bool sparse = e.in_flags && (!e.in_count || 5ull * (*e.in_count) < 4ull * static_cast<unsigned long long>(g.n_cols));
    const float4 acc = sparse ? gather_rows<LPR, true>(g, X, start, end, lane, e.in_flags) : gather_rows<LPR>(g, X, start, end, lane);
    if (lane < LPR) reinterpret_cast<float4*>(lv.slab)[c * LPR + lane] = acc;
    return;
  }
  const int64_t r = static_cast<int64_t>(blockIdx.x - lv.chunk_blocks) * kWavesPerBlock + (threadIdx.x >> 6);
  if (r >= g.n_rows) return;
  if (MASKED && !e.row_mask[r]) return;
  const int64_t start = g.rowptr[r], end = g.rowptr[r + 1];
  if (end - start > kLongRow) return;  // chunked above, folded by spmm_finish_kernel
  const bool sparse = e.in_flags && (!e.in_count || 5ull * (*e.in_count) < 4ull * static_cast<unsigned long long>(g.n_cols));
  const float4 acc = sparse ? gather_rows<LPR, true>(g, X, start, end, lane, e.in_flags) : gather_rows<LPR>(g, X, start, end, lane);
  row_epilogue<LPR, EPI>(acc, r, lane, e);
}

// ---- masked hop with a handful of flagged operand rows: one row per LANE GROUP ------------------------------------------
// The hop below the top layer of a restricted backward pass visits the rows of a mask (the batch rows' neighbours: ~40 % of
// the graph at C2) and gathers only FLAGGED operand rows (the <= 3 B batch rows): a row reads its ~50 column ids and flag
// bytes and finds 0-3 of them flagged.  With one wavefront per row that is three dependent memory latencies per row and
// nothing to amortise them (0.73 ms for 160 MB of indices at C2).  Here every lane group of LPR lanes walks ITS OWN row
// (64 / LPR rows per wave): the latencies of 4 rows (D = 64) overlap, the flagged entries of a group are taken in entry
// order (a fixed order: the result is reproducible), and the epilogue runs per group.  Long rows: chunks first, as in
// spmm_rows_kernel, folded by spmm_finish_kernel.
template <int LPR, int EPI>
__global__ __launch_bounds__(kWavesPerBlock * kWave) void spmm_rows_grouped_kernel(GraphView g, const float* __restrict__ X,
                                                                                    EpiArgs e, LongView lv) {
  static_assert(EPI == EPI_NORMBWD || EPI == EPI_AXPY, "grouped rows: the two masked backward hops");
  static_assert(LPR <= 32, "grouped rows: at least two rows per wavefront");
  constexpr int NPI = kWave / LPR;
  const int lane = threadIdx.x & (kWave - 1);
  if (blockIdx.x < lv.chunk_blocks) {
    const int64_t c = static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + (threadIdx.x >> 6);
    if (c >= lv.n_chunks) return;
    const int2 d = lv.chunk_desc[c];
    if (d.x < 0) return;
    const int64_t r = lv.long_rows[d.x];
    if (!e.row_mask[r]) return;
    const int64_t start = g.rowptr[r] + static_cast<int64_t>(d.y) * kChunk;
    const int64_t row_end = g.rowptr[r + 1];
    const int64_t end = (start + kChunk < row_end) ? start + kChunk : row_end;
    const float4 acc = gather_rows<LPR, true>(g, X, start, end, lane, e.in_flags);
    if (lane < LPR) reinterpret_cast<float4*>(lv.slab)[c * LPR + lane] = acc;
    return;
  }
  const int q = lane / LPR, c = lane % LPR;
  const int64_t wv = static_cast<int64_t>(blockIdx.x - lv.chunk_blocks) * kWavesPerBlock + (threadIdx.x >> 6);
  const int64_t r = wv * NPI + q;
  bool valid = r < g.n_rows && e.row_mask[r];
  int64_t start = 0;
  int len = 0;
  if (valid) {
    start = g.rowptr[r];
    const int64_t deg = g.rowptr[r + 1] - start;
    if (deg > kLongRow) valid = false;          // chunked above, folded by spmm_finish_kernel
    else len = static_cast<int>(deg);
  }
  int maxlen = len;
#pragma unroll
  for (int m = LPR; m < kWave; m <<= 1) maxlen = max(maxlen, __shfl_xor(maxlen, m));
  const float4* __restrict__ Xv = reinterpret_cast<const float4*>(X) + c;
  float4 acc = f4_zero();
  for (int base = 0; base < maxlen; base += LPR) {
    const int n = len - base;                    // entries of this group's row in the batch (may be <= 0)
    int my_col = 0;
    float my_val = 0.f;
    if (c < n) {
      my_col = ld_stream(g.col + start + base + c);
      if (e.in_flags[my_col]) my_val = ld_stream(g.val + start + base + c);      // a zero weight marks "row not needed"
    }
    const unsigned long long m = __ballot(my_val != 0.f);
    unsigned gm = static_cast<unsigned>((m >> (q * LPR)) & ((1ull << LPR) - 1ull));   // the group's flagged entries
    int maxcnt = __popc(gm);
#pragma unroll
    for (int mm = LPR; mm < kWave; mm <<= 1) maxcnt = max(maxcnt, __shfl_xor(maxcnt, mm));
    for (int t = 0; t < maxcnt; t += 2) {
      float4 x[2];
      float v[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const bool ok = gm != 0u;
        const int bit = ok ? __ffs(static_cast<int>(gm)) - 1 : 0;
        gm &= gm - 1u;                                                            // (0 stays 0)
        const int col = __shfl(my_col, q * LPR + bit);
        const float w = __shfl(my_val, q * LPR + bit);
        v[u] = ok ? w : 0.f;
        x[u] = ok ? Xv[static_cast<int64_t>(col) * LPR] : f4_zero();
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) f4_fma(acc, v[u], x[u]);
    }
  }
  if (!valid) return;                            // (whole lane groups leave: the group reductions below stay inside a group)
  const int64_t off = r * LPR + c;
  float4 o = acc;
  if constexpr (EPI == EPI_NORMBWD) {
    if (!e.b_flags || e.b_flags[r]) {
      const float4 xr = ld_stream(reinterpret_cast<const float4*>(e.Xraw) + off);
      float4 dz = ld_stream(reinterpret_cast<const float4*>(e.B) + off);
      dz.x *= e.s; dz.y *= e.s; dz.z *= e.s; dz.w *= e.s;
      const float4 gz = normalize_bwd<LPR>(xr, e.inv_norm[r], dz);
      o = make_float4(acc.x + gz.x, acc.y + gz.y, acc.z + gz.z, acc.w + gz.w);
    }
    drop4(e.drop, off, o.x, o.y, o.z, o.w);
    st_stream(reinterpret_cast<float4*>(e.Y) + off, o);
    if (e.out_flags) {
      const float nz = group_sum<LPR>((o.x != 0.f || o.y != 0.f || o.z != 0.f || o.w != 0.f) ? 1.f : 0.f);
      if (c == 0) e.out_flags[r] = nz != 0.f;
    }
  } else {
    if (!(e.b_flags && !e.b_flags[r])) {
      const float4 b = ld_stream(reinterpret_cast<const float4*>(e.B) + off);
      o = make_float4(fmaf(e.s, b.x, acc.x), fmaf(e.s, b.y, acc.y), fmaf(e.s, b.z, acc.z), fmaf(e.s, b.w, acc.w));
    }
    st_stream(reinterpret_cast<float4*>(e.Y) + off, o);
  }
}

template <int LPR, int EPI, bool MASKED = false>
__global__ __launch_bounds__(kWavesPerBlock * kWave) void spmm_finish_kernel(GraphView g,
                                                                              const int32_t* __restrict__ long_rows,
                                                                              const int32_t* __restrict__ long_base,
                                                                              int64_t n_long, const float* __restrict__ slab,
                                                                              EpiArgs e) {
  const int lane = threadIdx.x & (kWave - 1);
  const int64_t li = static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + (threadIdx.x >> 6);
  if (li >= n_long) return;
  const int64_t r = long_rows[li];
  if (r < 0) return;                        // unused slot of a handle created without a host read
  if (MASKED && !e.row_mask[r]) return;
  const int64_t deg = g.rowptr[r + 1] - g.rowptr[r];
  const int nc = static_cast<int>((deg + kChunk - 1) / kChunk);
  const float4* p = reinterpret_cast<const float4*>(slab) + static_cast<int64_t>(long_base[li]) * LPR + (lane % LPR);
  // Lane group q = lane / LPR sums the chunks q, q + NPI, ... in ascending order (four loads in flight), then the groups
  // are combined by an xor butterfly: a fixed order, so the sum is reproducible -- and a row of 1e5 entries (200 chunks)
  // is 13 dependent steps instead of 200.
  constexpr int NPI = kWave / LPR;
  auto add = [](float4& a, const float4& x) { a.x += x.x; a.y += x.y; a.z += x.z; a.w += x.w; };
  float4 acc = f4_zero();
  int k = lane / LPR;
  for (; k + 3 * NPI < nc; k += 4 * NPI) {
    const float4 x0 = p[static_cast<int64_t>(k) * LPR], x1 = p[static_cast<int64_t>(k + NPI) * LPR];
    const float4 x2 = p[static_cast<int64_t>(k + 2 * NPI) * LPR], x3 = p[static_cast<int64_t>(k + 3 * NPI) * LPR];
    add(acc, x0); add(acc, x1); add(acc, x2); add(acc, x3);
  }
  for (; k < nc; k += NPI) add(acc, p[static_cast<int64_t>(k) * LPR]);
#pragma unroll
  for (int m = LPR; m < kWave; m <<= 1) add(acc, f4_shfl_xor(acc, m));
  row_epilogue<LPR, EPI>(acc, r, lane, e);
}

// ---- any width D <= 512: scalar columns, one neighbour row per instruction ---------------------
template <int EPI>
__global__ __launch_bounds__(kWavesPerBlock * kWave) void spmm_rows_generic_kernel(GraphView g, const float* __restrict__ X,
                                                                                    EpiArgs e, int D) {
  const int lane = threadIdx.x & (kWave - 1);
  const int64_t r = static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + (threadIdx.x >> 6);
  if (r >= g.n_rows) return;
  const int64_t start = g.rowptr[r], end = g.rowptr[r + 1];
  float acc[kMaxGenericBlocks];
#pragma unroll
  for (int b = 0; b < kMaxGenericBlocks; ++b) acc[b] = 0.f;
  for (int64_t base = start; base < end; base += kWave) {
    const int n = (end - base) < kWave ? static_cast<int>(end - base) : kWave;
    int my_col = 0;
    float my_val = 0.f;
    if (lane < n) {
      my_col = g.col[base + lane];
      my_val = g.val[base + lane];
    }
    for (int j = 0; j < n; ++j) {
      const int64_t c = __shfl(my_col, j);
      const float w = __shfl(my_val, j);
      const float* xr = X + c * D;
#pragma unroll
      for (int b = 0; b < kMaxGenericBlocks; ++b) {
        const int k = b * kWave + lane;
        if (k < D) acc[b] = fmaf(w, xr[k], acc[b]);
      }
    }
  }
  const int64_t row = r * D;
  if constexpr (EPI == EPI_NONE) {
#pragma unroll
    for (int b = 0; b < kMaxGenericBlocks; ++b) {
      const int k = b * kWave + lane;
      if (k < D) e.Y[row + k] = acc[b];
    }
  } else if constexpr (EPI == EPI_NORM_ACC) {
    float ss = 0.f;
#pragma unroll
    for (int b = 0; b < kMaxGenericBlocks; ++b) ss = fmaf(acc[b], acc[b], ss);  // lanes past D hold 0
    ss = group_sum<kWave>(ss);
    const float den = fmaxf(sqrtf(ss), 1e-12f);
#pragma unroll
    for (int b = 0; b < kMaxGenericBlocks; ++b) {
      const int k = b * kWave + lane;
      if (k < D) {
        e.Y[row + k] = acc[b];
        e.accum[row + k] = fmaf(e.s, acc[b] / den, e.accum[row + k]);
      }
    }
    if (lane == 0) e.inv_norm[r] = 1.0f / den;
  } else if constexpr (EPI == EPI_NORMBWD) {
    const float inv = e.inv_norm[r];
    float z[kMaxGenericBlocks], dz[kMaxGenericBlocks];
    float dot = 0.f;
#pragma unroll
    for (int b = 0; b < kMaxGenericBlocks; ++b) {
      const int k = b * kWave + lane;
      z[b] = (k < D) ? e.Xraw[row + k] * inv : 0.f;
      dz[b] = (k < D) ? e.B[row + k] * e.s : 0.f;
      dot = fmaf(z[b], dz[b], dot);
    }
    dot = group_sum<kWave>(dot);
    if (inv >= 1e12f) dot = 0.f;
#pragma unroll
    for (int b = 0; b < kMaxGenericBlocks; ++b) {
      const int k = b * kWave + lane;
      if (k < D) e.Y[row + k] = acc[b] + inv * (dz[b] - z[b] * dot);
    }
  } else if constexpr (EPI == EPI_AXPY) {
#pragma unroll
    for (int b = 0; b < kMaxGenericBlocks; ++b) {
      const int k = b * kWave + lane;
      if (k < D) e.Y[row + k] = fmaf(e.s, e.B[row + k], acc[b]);
    }
  } else if constexpr (EPI == EPI_SS) {
    float ss = 0.f;
#pragma unroll
    for (int b = 0; b < kMaxGenericBlocks; ++b) {
      const int k = b * kWave + lane;
      ss = fmaf(acc[b], acc[b], ss);
      if (k < D) e.Y[row + k] = acc[b];
    }
    ss = group_sum<kWave>(ss);
    if (lane == 0) e.inv_norm[r] = ss;
  } else {
    const float inv = e.inv_norm[r];
    const float dot = inv >= 1e12f ? 0.f : e.dot[r];
#pragma unroll
    for (int b = 0; b < kMaxGenericBlocks; ++b) {
      const int k = b * kWave + lane;
      if (k < D) e.Y[row + k] = acc[b] + inv * (e.s * e.B[row + k] - e.Xraw[row + k] * inv * dot);
    }
  }
}


// ---- product on a short LIST of rows, compact output: Y[k, :] = sum_j val_j X[col_j, :] over row rows[k] ----------
// This is the top layer of a training step (the loss reads that layer at the <= 3 B batch rows only) and, with
// A = the column slice A[:, rows_g], a rank's share of it in the row-sharded step (dist.py).  The listed rows are the
// batch's users and items, and positive items are popular: rows of 1e5 entries are the common case.  So every listed row
// is cut into kListedSplits equal ranges (multiples of 64 entries); block (k, s) sums range s of row k with its four
// waves (each a contiguous quarter, folded in wave order through LDS) into ws[k][s], and a second kernel adds the
// kListedSplits partial rows in order: a fixed summation order, and the longest row costs 1/128 of its length per wave.
constexpr int kListedSplits = 32;

template <int LPR>
__global__ __launch_bounds__(kWavesPerBlock * kWave) void spmm_listed_kernel(GraphView g, const int64_t* __restrict__ rows,
                                                                              const float* __restrict__ X, float* __restrict__ ws) {
  __shared__ float4 part[kWavesPerBlock][LPR];
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
  const int64_t k = blockIdx.x / kListedSplits;
  const int sp = blockIdx.x % kListedSplits;
  const int64_t r = rows[k];
  const int64_t start = g.rowptr[r], end = g.rowptr[r + 1];
  const int64_t per_split = ((end - start + kListedSplits * kWave - 1) / (kListedSplits * kWave)) * kWave;
  const int64_t b0 = (start + sp * per_split < end) ? start + sp * per_split : end;
  const int64_t b1 = (b0 + per_split < end) ? b0 + per_split : end;
  float4* out = reinterpret_cast<float4*>(ws) + (k * kListedSplits + sp) * LPR;
  if (b0 >= b1) {                                   // nothing in this range (short rows use the first ranges only)
    if (threadIdx.x < LPR) out[threadIdx.x] = f4_zero();
    return;
  }
  const int64_t per = ((b1 - b0 + kWavesPerBlock * kWave - 1) / (kWavesPerBlock * kWave)) * kWave;
  const int64_t s0 = (b0 + wave * per < b1) ? b0 + wave * per : b1;
  const int64_t s1 = (s0 + per < b1) ? s0 + per : b1;
  const float4 acc = gather_rows<LPR>(g, X, s0, s1, lane);
  if (lane < LPR) part[wave][lane] = acc;
  __syncthreads();
  if (wave == 0 && lane < LPR) {
    float4 a = part[0][lane];
#pragma unroll
    for (int w = 1; w < kWavesPerBlock; ++w) {
      const float4 b = part[w][lane];
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    out[lane] = a;
  }
}

template <int LPR>
__global__ __launch_bounds__(256) void spmm_listed_fold_kernel(const float* __restrict__ ws, float* __restrict__ Y, int64_t n_listed) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;      // one float4 of the output per thread
  if (i >= n_listed * LPR) return;
  const int64_t k = i / LPR;
  const int c = static_cast<int>(i % LPR);
  const float4* p = reinterpret_cast<const float4*>(ws) + k * kListedSplits * LPR + c;
  float4 a = p[0];
#pragma unroll 4
  for (int sp = 1; sp < kListedSplits; ++sp) {
    const float4 b = p[static_cast<int64_t>(sp) * LPR];
    a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
  }
  reinterpret_cast<float4*>(Y)[i] = a;
}

// flags[c] = 1 for every column index stored in the listed rows, and for the rows themselves (block per listed row)
template <bool SELF>
__global__ __launch_bounds__(256) void mark_rows_kernel(GraphView g, const int64_t* __restrict__ rows, int64_t n_listed,
                                                        uint8_t* __restrict__ flags) {
  // every listed row is cut into kListedSplits ranges (batch rows are popular items: 1e5 entries), one block each
  const int64_t i = blockIdx.x / kListedSplits;
  const int part = blockIdx.x % kListedSplits;
  if (i >= n_listed) return;
  const int64_t r = rows[i];
  if (SELF && part == 0 && threadIdx.x == 0) flags[r] = 1;
  const int64_t start = g.rowptr[r], end = g.rowptr[r + 1];
  const int64_t per = (end - start + kListedSplits - 1) / kListedSplits;
  const int64_t b0 = start + part * per;
  const int64_t b1 = (b0 + per < end) ? b0 + per : end;
  for (int64_t j = b0 + threadIdx.x; j < b1; j += blockDim.x) flags[g.col[j]] = 1;
}

// ---- TGCN attention backward over an inverted neighbour table (csrc/tgcn.hip, model/tgcn.py:20-37) ------------------------
// Row j of the table lists the (source node v, slot n) pairs that point at destination j, as pair ids p = v k + n in
// `g.col` (g.val is not read).  Two pulls walk it, both on the long-row machinery above (chunks first, fixed-order fold):
//
//   (attn_pull_da reads the pair's source row from g.col and its attention weight from g.val, and the pair id from a
//   parallel array; attn_pull_dq reads pair ids from g.col)
//   attn_pull_da : dEj[j] = sum_p attn[p] dOut[p / k]  (+ B[j])  -- and, since the wave that walks row j holds BOTH
//                  operands of it, da[p] = dOut[p / k] . Ej[j]: the softmax-backward input of pair p.  The source-centric
//                  backward kernel then does not gather the D-wide neighbour rows a second time (512 of its 776 bytes per
//                  pair at D = 128).
//   attn_pull_dq : dQ[j] = v (.) sum_p ds[p] mask[p]  (+ B[j]) from the COMPRESSED pre-activation gradients: per pair one
//                  float ds and the A relu bits, 8 bytes instead of the 4 A bytes of dh.
// sum over the LPR lanes of a lane group with DPP inside a 16-lane row (register speed: no LDS round trip), bpermutes across rows
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
  return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}
template <int LPR>
__device__ __forceinline__ float group_sum_dpp(float v) {
  if constexpr (LPR >= 2) v = dpp_add<0xB1>(v);      // quad_perm [1,0,3,2]
  if constexpr (LPR >= 4) v = dpp_add<0x4E>(v);      // quad_perm [2,3,0,1]
  if constexpr (LPR >= 8) v = dpp_add<0x141>(v);     // row_half_mirror
  if constexpr (LPR >= 16) v = dpp_add<0x140>(v);    // row_mirror
  if constexpr (LPR >= 32) v += __shfl_xor(v, 16);
  if constexpr (LPR >= 64) v += __shfl_xor(v, 32);
  return v;
}

template <int LPR>
__device__ __forceinline__ float4 gather_pairs_da(const GraphView& g, const int32_t* __restrict__ pair, int64_t start, int64_t end,
                                                  int lane, const float* __restrict__ dOut, const float4 own,
                                                  float* __restrict__ da) {
  constexpr int NPI = kWave / LPR;
  const int q = lane / LPR, c = lane % LPR;
  const float4* __restrict__ Xv = reinterpret_cast<const float4*>(dOut) + c;
  float4 acc = f4_zero();
  for (int64_t base = start; base < end; base += kWave) {
    const int n = (end - base) < kWave ? static_cast<int>(end - base) : kWave;
    int my_src = 0, my_pair = 0;
    float my_val = 0.f, my_da = 0.f;
    if (lane < n) {                      // source row and attention weight of the pair, streamed (filled by attn_invert_fill)
      my_src = ld_stream(g.col + base + lane);
      my_val = ld_stream(g.val + base + lane);
      my_pair = ld_stream(pair + base + lane);
    }
    const int groups = (n + NPI - 1) / NPI;
    for (int gi = 0; gi < groups; gi += 4) {
      float4 x[4];
      float v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int j = (gi + u) * NPI + q;
        const int src = __shfl(my_src, j & (kWave - 1));
        const float w = __shfl(my_val, j & (kWave - 1));
        const bool ok = j < n;
        v[u] = ok ? w : 0.f;
        x[u] = ok ? Xv[static_cast<int64_t>(src) * LPR] : f4_zero();
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        f4_fma(acc, v[u], x[u]);
        const float d = group_sum_dpp<LPR>(f4_dot(x[u], own));
        // entry (gi + u) NPI + q' sits in lane group q': lane L of the batch collects its own entry's dot product, so that the
        // 64 results leave with ONE store instruction (two active lanes per store cost a third of the kernel)
        const float dd = __shfl(d, (lane % NPI) * LPR);
        if (lane / NPI == gi + u) my_da = dd;
      }
    }
    if (lane < n) da[my_pair] = my_da;
  }
#pragma unroll
  for (int m = LPR; m < kWave; m <<= 1) {
    const float4 o = f4_shfl_xor(acc, m);
    acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w;
  }
  return acc;
}

// Short rows, 64 / LPR of them per wavefront: lane group q walks ITS OWN row (row = wave * NPI + q), so the fixed cost of a
// row -- row pointer -> entries -> gather -> store, three dependent memory latencies that a destination with ~6 incoming
// pairs cannot amortise -- is paid once per NPI rows, and the dot product of entry e is kept by lane e of the group without a
// shuffle.  Long rows: one chunk per wave as in spmm_rows_kernel (the first blocks), folded by spmm_finish_kernel.
template <int LPR, int EPI>
__global__ __launch_bounds__(kWavesPerBlock * kWave) void attn_pull_da_kernel(GraphView g, const int32_t* __restrict__ pair,
                                                                               const float* __restrict__ dOut,
                                                                               const float* __restrict__ Ej, float* __restrict__ da,
                                                                               EpiArgs e, LongView lv) {
  constexpr int NPI = kWave / LPR;
  const int lane = threadIdx.x & (kWave - 1);
  if (blockIdx.x < lv.chunk_blocks) {
    const int64_t chunk = static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + (threadIdx.x >> 6);
    if (chunk >= lv.n_chunks) return;
    const int2 d = lv.chunk_desc[chunk];
    if (d.x < 0) return;
    const int64_t r = lv.long_rows[d.x];
    const int64_t start = g.rowptr[r] + static_cast<int64_t>(d.y) * kChunk;
    const int64_t row_end = g.rowptr[r + 1];
    const int64_t end = (start + kChunk < row_end) ? start + kChunk : row_end;
    const float4 own = reinterpret_cast<const float4*>(Ej)[r * LPR + (lane % LPR)];
    const float4 acc = gather_pairs_da<LPR>(g, pair, start, end, lane, dOut, own, da);
    if (lane < LPR) reinterpret_cast<float4*>(lv.slab)[chunk * LPR + lane] = acc;
    return;
  }
  const int q = lane / LPR, c = lane % LPR;
  const int64_t wv = static_cast<int64_t>(blockIdx.x - lv.chunk_blocks) * kWavesPerBlock + (threadIdx.x >> 6);
  const int64_t r = wv * NPI + q;
  const bool valid = r < g.n_rows;
  int64_t start = 0;
  int len = 0;
  bool is_long = false;
  if (valid) {
    start = g.rowptr[r];
    const int64_t deg = g.rowptr[r + 1] - start;
    is_long = deg > kLongRow;
    len = is_long ? 0 : static_cast<int>(deg);
  }
  int maxlen = len;
#pragma unroll
  for (int m = LPR; m < kWave; m <<= 1) maxlen = max(maxlen, __shfl_xor(maxlen, m));
  const float4 own = len > 0 ? reinterpret_cast<const float4*>(Ej)[r * LPR + c] : f4_zero();
  const float4* __restrict__ Xv = reinterpret_cast<const float4*>(dOut) + c;
  float4 acc = f4_zero();
  for (int base = 0; base < maxlen; base += LPR) {
    const int n = len - base;                           // entries of this group's row in the batch (may be <= 0)
    int my_src = 0, my_pair = 0;
    float my_val = 0.f, my_da = 0.f;
    if (c < n) {
      my_src = ld_stream(g.col + start + base + c);
      my_val = ld_stream(g.val + start + base + c);
      my_pair = ld_stream(pair + start + base + c);
    }
    const int nmax = (maxlen - base) < LPR ? (maxlen - base) : LPR;
    for (int i = 0; i < nmax; i += 4) {
      float4 x[4];
      float v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int en = i + u;
        const int src = __shfl(my_src, q * LPR + (en & (LPR - 1)));
        const float w = __shfl(my_val, q * LPR + (en & (LPR - 1)));
        const bool ok = en < n;
        v[u] = ok ? w : 0.f;
        x[u] = ok ? Xv[static_cast<int64_t>(src) * LPR] : f4_zero();
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        f4_fma(acc, v[u], x[u]);
        const float d = group_sum_dpp<LPR>(f4_dot(x[u], own));
        if (c == i + u) my_da = d;                      // lane e of the group keeps entry e's dot product
      }
    }
    if (c < n) da[my_pair] = my_da;
  }
  if (!valid || is_long) return;
  const int64_t off = r * LPR + c;
  if constexpr (EPI == EPI_AXPY) {
    const float4 b = ld_stream(reinterpret_cast<const float4*>(e.B) + off);
    acc = make_float4(fmaf(e.s, b.x, acc.x), fmaf(e.s, b.y, acc.y), fmaf(e.s, b.z, acc.z), fmaf(e.s, b.w, acc.w));
  }
  st_stream(reinterpret_cast<float4*>(e.Y) + off, acc);
}

// A lanes per pair, 64 / A pairs per step; on return lanes [0, A/4) hold the row's float4 columns of v (.) sum
template <int A>
__device__ __forceinline__ float4 gather_pairs_dq(const int32_t* __restrict__ pair, int64_t start, int64_t end, int lane,
                                                  const float2* __restrict__ comp, const float* __restrict__ vv) {
  constexpr int SLOTS = kWave / A;
  const int slot = lane / A, c = lane % A;
  float acc = 0.f;
  for (int64_t base = start; base < end; base += kWave) {
    const int n = (end - base) < kWave ? static_cast<int>(end - base) : kWave;
    float my_ds = 0.f;
    int my_bits = 0;
    if (lane < n) {
      const float2 cv = comp[ld_stream(pair + base + lane)];
      my_ds = cv.x;
      my_bits = __float_as_int(cv.y);
    }
    const int steps = (n + SLOTS - 1) / SLOTS;
    for (int i = 0; i < steps; ++i) {
      const int j = i * SLOTS + slot;                 // (lanes past n hold ds = 0)
      const float ds = __shfl(my_ds, j & (kWave - 1));
      const int bits = __shfl(my_bits, j & (kWave - 1));
      acc += ((bits >> c) & 1) ? ds : 0.f;
    }
  }
#pragma unroll
  for (int m = A; m < kWave; m <<= 1) acc += __shfl_xor(acc, m);
  acc *= vv[c];
  // scalar-per-lane -> float4 columns on the first A/4 lanes
  const int b = (lane * 4) & (kWave - 1);
  return make_float4(__shfl(acc, b), __shfl(acc, b + 1), __shfl(acc, b + 2), __shfl(acc, b + 3));
}

template <int A, int EPI>
__global__ __launch_bounds__(kWavesPerBlock * kWave) void attn_pull_dq_kernel(GraphView g, const float2* __restrict__ comp,
                                                                               const float* __restrict__ vv, EpiArgs e, LongView lv) {
  constexpr int LPR = A / 4;
  const int lane = threadIdx.x & (kWave - 1);
  if (blockIdx.x < lv.chunk_blocks) {
    const int64_t c = static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + (threadIdx.x >> 6);
    if (c >= lv.n_chunks) return;
    const int2 d = lv.chunk_desc[c];
    if (d.x < 0) return;                    // unused slot of a handle created without a host read
    const int64_t r = lv.long_rows[d.x];
    const int64_t start = g.rowptr[r] + static_cast<int64_t>(d.y) * kChunk;
    const int64_t row_end = g.rowptr[r + 1];
    const int64_t end = (start + kChunk < row_end) ? start + kChunk : row_end;
    const float4 acc = gather_pairs_dq<A>(g.col, start, end, lane, comp, vv);
    if (lane < LPR) reinterpret_cast<float4*>(lv.slab)[c * LPR + lane] = acc;
    return;
  }
  const int64_t r = static_cast<int64_t>(blockIdx.x - lv.chunk_blocks) * kWavesPerBlock + (threadIdx.x >> 6);
  if (r >= g.n_rows) return;
  const int64_t start = g.rowptr[r], end = g.rowptr[r + 1];
  if (end - start > kLongRow) return;
  const float4 acc = gather_pairs_dq<A>(g.col, start, end, lane, comp, vv);
  row_epilogue<LPR, EPI>(acc, r, lane, e);
}

// ---- row-length scan at graph creation ---------------------------------------------------------
__global__ void count_long_kernel(const int64_t* __restrict__ rowptr, int64_t n_rows, unsigned long long* counters) {
  const int64_t r = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (r >= n_rows) return;
  const int64_t deg = rowptr[r + 1] - rowptr[r];
  if (deg > kLongRow) {
    atomicAdd(&counters[0], 1ull);
    atomicAdd(&counters[1], static_cast<unsigned long long>((deg + kChunk - 1) / kChunk));
  }
}

__global__ void fill_long_kernel(const int64_t* __restrict__ rowptr, int64_t n_rows, unsigned long long* counters,
                                 int32_t* long_rows, int32_t* long_base, int2* chunk_desc) {
  const int64_t r = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (r >= n_rows) return;
  const int64_t deg = rowptr[r + 1] - rowptr[r];
  if (deg > kLongRow) {
    const int nc = static_cast<int>((deg + kChunk - 1) / kChunk);
    const int li = static_cast<int>(atomicAdd(&counters[2], 1ull));
    const int base = static_cast<int>(atomicAdd(&counters[3], static_cast<unsigned long long>(nc)));
    long_rows[li] = static_cast<int32_t>(r);
    long_base[li] = base;
    for (int k = 0; k < nc; ++k) chunk_desc[base + k] = make_int2(li, k);
  }
}

}  // namespace tagrec

using namespace tagrec;

extern "C" int tagrec_graph_create(tagrec_graph** out, int64_t n_rows, int64_t n_cols, int64_t nnz,
                                   const int64_t* rowptr, const int32_t* colidx, const float* vals, void* stream) {
  TAGREC_REQUIRE(out != nullptr, "graph_create: out is null");
  TAGREC_REQUIRE(n_rows >= 0 && n_cols >= 0 && nnz >= 0, "graph_create: negative size");
  TAGREC_REQUIRE(n_rows < (1ll << 31) && n_cols < (1ll << 31), "graph_create: node ids must fit int32");
  TAGREC_REQUIRE(rowptr != nullptr, "graph_create: rowptr is null");
  TAGREC_REQUIRE(nnz == 0 || (colidx != nullptr && vals != nullptr), "graph_create: colidx/vals null with nnz > 0");
  hipStream_t s = static_cast<hipStream_t>(stream);
  tagrec_graph* g = new (std::nothrow) tagrec_graph();
  if (!g) return fail(TAGREC_E_NOMEM, "graph_create: host allocation failed");
  *g = tagrec_graph{n_rows, n_cols, nnz, rowptr, colidx, vals, 0, 0, nullptr, nullptr, nullptr, nullptr, 0, true, true};
  if (n_rows > 0) {
    unsigned long long* counters = nullptr;
    unsigned long long host[4] = {0, 0, 0, 0};
    hipError_t err = hipMalloc(&counters, sizeof(host));
    if (err != hipSuccess) { delete g; return fail(TAGREC_E_HIP, std::string("graph_create: hipMalloc: ") + hipGetErrorString(err)); }
    auto bail = [&](const char* what, hipError_t e2) {
      (void)hipFree(counters); (void)hipFree(g->long_rows); (void)hipFree(g->long_base); (void)hipFree(g->chunk_desc);
      (void)hipFree(g->slab);
      delete g;
      return fail(TAGREC_E_HIP, std::string("graph_create: ") + what + ": " + hipGetErrorString(e2));
    };
    const int threads = 256;
    const unsigned blocks = static_cast<unsigned>((n_rows + threads - 1) / threads);
    if ((err = hipMemsetAsync(counters, 0, sizeof(host), s)) != hipSuccess) return bail("memset", err);
    count_long_kernel<<<blocks, threads, 0, s>>>(rowptr, n_rows, counters);
    if ((err = hipGetLastError()) != hipSuccess) return bail("count_long launch", err);
    if ((err = hipMemcpyAsync(host, counters, sizeof(host), hipMemcpyDeviceToHost, s)) != hipSuccess) return bail("memcpy", err);
    if ((err = hipStreamSynchronize(s)) != hipSuccess) return bail("sync", err);
    g->n_long = static_cast<int64_t>(host[0]);
    g->n_chunks = static_cast<int64_t>(host[1]);
    if (g->n_long > 0) {
      if (g->n_chunks >= (1ll << 31)) { (void)hipFree(counters); delete g; return fail(TAGREC_E_UNSUPPORTED, "graph_create: too many long-row chunks"); }
      if ((err = hipMalloc(&g->long_rows, sizeof(int32_t) * g->n_long)) != hipSuccess) return bail("hipMalloc long_rows", err);
      if ((err = hipMalloc(&g->long_base, sizeof(int32_t) * g->n_long)) != hipSuccess) return bail("hipMalloc long_base", err);
      if ((err = hipMalloc(&g->chunk_desc, sizeof(int2) * g->n_chunks)) != hipSuccess) return bail("hipMalloc chunk_desc", err);
      fill_long_kernel<<<blocks, threads, 0, s>>>(rowptr, n_rows, counters, g->long_rows, g->long_base, g->chunk_desc);
      if ((err = hipGetLastError()) != hipSuccess) return bail("fill_long launch", err);
      if ((err = hipStreamSynchronize(s)) != hipSuccess) return bail("sync", err);
      // partial-sum slab of the long-row chunks, sized once for the widest vector kernel: no later call allocates
      if ((err = hipMalloc(&g->slab, sizeof(float) * kSlabWidth * g->n_chunks)) != hipSuccess) return bail("hipMalloc slab", err);
      g->slab_floats = static_cast<size_t>(kSlabWidth) * g->n_chunks;
    }
    (void)hipFree(counters);
  }
  *out = g;
  return TAGREC_OK;
}

extern "C" int tagrec_graph_create_like(tagrec_graph** out, const tagrec_graph* like, int64_t n_cols, const int32_t* colidx,
                                        const float* vals) {
  TAGREC_REQUIRE(out != nullptr && like != nullptr, "graph_create_like: null handle");
  TAGREC_REQUIRE(n_cols >= 0 && n_cols < (1ll << 31), "graph_create_like: node ids must fit int32");
  TAGREC_REQUIRE(like->nnz == 0 || (colidx != nullptr && vals != nullptr), "graph_create_like: colidx/vals null with nnz > 0");
  tagrec_graph* g = new (std::nothrow) tagrec_graph(*like);
  if (!g) return fail(TAGREC_E_NOMEM, "graph_create_like: host allocation failed");
  g->n_cols = n_cols;
  g->col = colidx;
  g->val = vals;
  g->slab = nullptr;
  g->slab_floats = 0;
  g->owns_long = false;
  g->owns_slab = true;
  if (g->n_chunks > 0) {   // its own slab: the two matrices may be multiplied back to back on one stream
    hipError_t err = hipMalloc(&g->slab, sizeof(float) * kSlabWidth * g->n_chunks);
    if (err != hipSuccess) { delete g; return fail(TAGREC_E_HIP, std::string("graph_create_like: hipMalloc slab: ") + hipGetErrorString(err)); }
    g->slab_floats = static_cast<size_t>(kSlabWidth) * g->n_chunks;
  }
  *out = g;
  return TAGREC_OK;
}

// ---- handles on a caller-provided workspace: no hipMalloc / hipFree (short-lived matrices built inside a training step,
// e.g. the inverted neighbour tables of the TGCN attention backward).  Upper bounds: a long row has > kLongRow entries, so
// there are at most nnz / kLongRow of them, and a row of d entries has ceil(d / kChunk) <= d / kChunk + 1 chunks.
namespace {
struct WsLayout {
  int64_t n_long_max, n_chunks_max;
  size_t off_counters, off_long_rows, off_long_base, off_chunk_desc, off_slab, total;
};
WsLayout ws_layout(int64_t nnz) {
  WsLayout w;
  w.n_long_max = nnz / kLongRow + 1;
  w.n_chunks_max = nnz / kChunk + w.n_long_max + 1;
  auto up = [](size_t x) { return (x + 255) & ~static_cast<size_t>(255); };
  size_t o = 0;
  w.off_counters = o;   o = up(o + 4 * sizeof(unsigned long long));
  w.off_long_rows = o;  o = up(o + sizeof(int32_t) * w.n_long_max);
  w.off_long_base = o;  o = up(o + sizeof(int32_t) * w.n_long_max);
  w.off_chunk_desc = o; o = up(o + sizeof(int2) * w.n_chunks_max);
  w.off_slab = o;       o = up(o + sizeof(float) * kSlabWidth * w.n_chunks_max);
  w.total = o;
  return w;
}
}  // namespace

extern "C" int64_t tagrec_graph_workspace(int64_t nnz) { return nnz < 0 ? 0 : static_cast<int64_t>(ws_layout(nnz).total); }

extern "C" int tagrec_graph_create_ws(tagrec_graph** out, int64_t n_rows, int64_t n_cols, int64_t nnz, const int64_t* rowptr,
                                      const int32_t* colidx, const float* vals, void* workspace, int64_t workspace_bytes,
                                      void* stream) {
  TAGREC_REQUIRE(out != nullptr && rowptr != nullptr, "graph_create_ws: null pointer");
  TAGREC_REQUIRE(n_rows >= 0 && n_cols >= 0 && nnz >= 0 && n_rows < (1ll << 31) && n_cols < (1ll << 31), "graph_create_ws: bad size");
  TAGREC_REQUIRE(nnz == 0 || (colidx != nullptr && vals != nullptr), "graph_create_ws: colidx/vals null with nnz > 0");
  const WsLayout w = ws_layout(nnz);
  TAGREC_REQUIRE(workspace != nullptr && workspace_bytes >= static_cast<int64_t>(w.total) &&
                     (reinterpret_cast<uintptr_t>(workspace) & 255u) == 0,
                 "graph_create_ws: workspace smaller than tagrec_graph_workspace(nnz) or not 256-byte aligned");
  hipStream_t s = static_cast<hipStream_t>(stream);
  tagrec_graph* g = new (std::nothrow) tagrec_graph();
  if (!g) return fail(TAGREC_E_NOMEM, "graph_create_ws: host allocation failed");
  *g = tagrec_graph{n_rows, n_cols, nnz, rowptr, colidx, vals, 0, 0, nullptr, nullptr, nullptr, nullptr, 0, false, false};
  if (n_rows > 0) {
    char* base = static_cast<char*>(workspace);
    unsigned long long* counters = reinterpret_cast<unsigned long long*>(base + w.off_counters);
    unsigned long long host[4] = {0, 0, 0, 0};
    auto bail = [&](const char* what, hipError_t e2) {
      delete g;
      return fail(TAGREC_E_HIP, std::string("graph_create_ws: ") + what + ": " + hipGetErrorString(e2));
    };
    hipError_t err;
    const int threads = 256;
    const unsigned blocks = static_cast<unsigned>((n_rows + threads - 1) / threads);
    if ((err = hipMemsetAsync(counters, 0, sizeof(host), s)) != hipSuccess) return bail("memset", err);
    count_long_kernel<<<blocks, threads, 0, s>>>(rowptr, n_rows, counters);
    if ((err = hipGetLastError()) != hipSuccess) return bail("count_long launch", err);
    g->long_rows = reinterpret_cast<int32_t*>(base + w.off_long_rows);
    g->long_base = reinterpret_cast<int32_t*>(base + w.off_long_base);
    g->chunk_desc = reinterpret_cast<int2*>(base + w.off_chunk_desc);
    g->slab = reinterpret_cast<float*>(base + w.off_slab);
    // the fill runs unconditionally (it writes nothing when no row is long), so one read-back serves both kernels
    fill_long_kernel<<<blocks, threads, 0, s>>>(rowptr, n_rows, counters, g->long_rows, g->long_base, g->chunk_desc);
    if ((err = hipGetLastError()) != hipSuccess) return bail("fill_long launch", err);
    if ((err = hipMemcpyAsync(host, counters, sizeof(host), hipMemcpyDeviceToHost, s)) != hipSuccess) return bail("memcpy", err);
    if ((err = hipStreamSynchronize(s)) != hipSuccess) return bail("sync", err);
    g->n_long = static_cast<int64_t>(host[0]);
    g->n_chunks = static_cast<int64_t>(host[1]);
    if (g->n_long > w.n_long_max || g->n_chunks > w.n_chunks_max) {
      delete g;
      return fail(TAGREC_E_INVALID, "graph_create_ws: row pointer inconsistent with nnz (more long rows than nnz allows)");
    }
    g->slab_floats = static_cast<size_t>(kSlabWidth) * g->n_chunks;
  }
  *out = g;
  return TAGREC_OK;
}

// The same handle WITHOUT the host read of the two counters (a read drains the launch queue: the TGCN step builds twelve
// inverted tables per step).  The work list is sized by its upper bounds and its unused slots hold -1, which the chunk /
// finish waves of every kernel in this file skip.
extern "C" int tagrec_graph_create_ws_deferred(tagrec_graph** out, int64_t n_rows, int64_t n_cols, int64_t nnz, const int64_t* rowptr,
                                               const int32_t* colidx, const float* vals, void* workspace, int64_t workspace_bytes,
                                               void* stream) {
  TAGREC_REQUIRE(out != nullptr && rowptr != nullptr, "graph_create_ws_deferred: null pointer");
  TAGREC_REQUIRE(n_rows >= 0 && n_cols >= 0 && nnz >= 0 && n_rows < (1ll << 31) && n_cols < (1ll << 31), "graph_create_ws_deferred: bad size");
  TAGREC_REQUIRE(nnz == 0 || (colidx != nullptr && vals != nullptr), "graph_create_ws_deferred: colidx/vals null with nnz > 0");
  const WsLayout w = ws_layout(nnz);
  TAGREC_REQUIRE(workspace != nullptr && workspace_bytes >= static_cast<int64_t>(w.total) &&
                     (reinterpret_cast<uintptr_t>(workspace) & 255u) == 0,
                 "graph_create_ws_deferred: workspace smaller than tagrec_graph_workspace(nnz) or not 256-byte aligned");
  hipStream_t s = static_cast<hipStream_t>(stream);
  tagrec_graph* g = new (std::nothrow) tagrec_graph();
  if (!g) return fail(TAGREC_E_NOMEM, "graph_create_ws_deferred: host allocation failed");
  *g = tagrec_graph{n_rows, n_cols, nnz, rowptr, colidx, vals, 0, 0, nullptr, nullptr, nullptr, nullptr, 0, false, false};
  if (n_rows > 0) {
    char* base = static_cast<char*>(workspace);
    unsigned long long* counters = reinterpret_cast<unsigned long long*>(base + w.off_counters);
    auto bail = [&](const char* what, hipError_t e2) {
      delete g;
      return fail(TAGREC_E_HIP, std::string("graph_create_ws_deferred: ") + what + ": " + hipGetErrorString(e2));
    };
    hipError_t err;
    g->long_rows = reinterpret_cast<int32_t*>(base + w.off_long_rows);
    g->long_base = reinterpret_cast<int32_t*>(base + w.off_long_base);
    g->chunk_desc = reinterpret_cast<int2*>(base + w.off_chunk_desc);
    g->slab = reinterpret_cast<float*>(base + w.off_slab);
    if ((err = hipMemsetAsync(counters, 0, 4 * sizeof(unsigned long long), s)) != hipSuccess) return bail("memset", err);
    if ((err = hipMemsetAsync(g->long_rows, 0xFF, sizeof(int32_t) * w.n_long_max, s)) != hipSuccess) return bail("memset", err);
    if ((err = hipMemsetAsync(g->chunk_desc, 0xFF, sizeof(int2) * w.n_chunks_max, s)) != hipSuccess) return bail("memset", err);
    const int threads = 256;
    const unsigned blocks = static_cast<unsigned>((n_rows + threads - 1) / threads);
    fill_long_kernel<<<blocks, threads, 0, s>>>(rowptr, n_rows, counters, g->long_rows, g->long_base, g->chunk_desc);
    if ((err = hipGetLastError()) != hipSuccess) return bail("fill_long launch", err);
    g->n_long = w.n_long_max;            // upper bounds (ws_layout): the slots past the real counts stay -1
    g->n_chunks = w.n_chunks_max;
    g->slab_floats = static_cast<size_t>(kSlabWidth) * g->n_chunks;
    g->deferred = true;
  }
  *out = g;
  return TAGREC_OK;
}

extern "C" int tagrec_graph_create_like_ws(tagrec_graph** out, const tagrec_graph* like, int64_t n_cols, const int32_t* colidx,
                                           const float* vals, void* workspace, int64_t workspace_bytes) {
  TAGREC_REQUIRE(out != nullptr && like != nullptr, "graph_create_like_ws: null handle");
  TAGREC_REQUIRE(n_cols >= 0 && n_cols < (1ll << 31), "graph_create_like_ws: node ids must fit int32");
  TAGREC_REQUIRE(like->nnz == 0 || (colidx != nullptr && vals != nullptr), "graph_create_like_ws: colidx/vals null with nnz > 0");
  const WsLayout w = ws_layout(like->nnz);
  TAGREC_REQUIRE(workspace != nullptr && workspace_bytes >= static_cast<int64_t>(w.total) &&
                     (reinterpret_cast<uintptr_t>(workspace) & 255u) == 0,
                 "graph_create_like_ws: workspace smaller than tagrec_graph_workspace(nnz) or not 256-byte aligned");
  tagrec_graph* g = new (std::nothrow) tagrec_graph(*like);
  if (!g) return fail(TAGREC_E_NOMEM, "graph_create_like_ws: host allocation failed");
  g->n_cols = n_cols;
  g->col = colidx;
  g->val = vals;
  g->owns_long = false;
  g->owns_slab = false;
  g->slab = reinterpret_cast<float*>(static_cast<char*>(workspace) + w.off_slab);   // only the slab part is used
  g->slab_floats = static_cast<size_t>(kSlabWidth) * g->n_chunks;
  *out = g;
  return TAGREC_OK;
}

extern "C" int tagrec_graph_destroy(tagrec_graph* g) {
  if (!g) return TAGREC_OK;
  if (g->owns_long) {
    (void)hipFree(g->long_rows);
    (void)hipFree(g->long_base);
    (void)hipFree(g->chunk_desc);
  }
  if (g->owns_slab) (void)hipFree(g->slab);
  delete g;
  return TAGREC_OK;
}

extern "C" int tagrec_graph_info(const tagrec_graph* g, int64_t* n_rows, int64_t* n_cols, int64_t* nnz,
                                 int64_t* n_long_rows, int64_t* n_chunks) {
  TAGREC_REQUIRE(g != nullptr, "graph_info: null handle");
  if (n_rows) *n_rows = g->n_rows;
  if (n_cols) *n_cols = g->n_cols;
  if (nnz) *nnz = g->nnz;
  if (n_long_rows) *n_long_rows = g->n_long;
  if (n_chunks) *n_chunks = g->n_chunks;
  return TAGREC_OK;
}

int tagrec::ensure_slab(const tagrec_graph* g, int D) {
  // The slab is allocated when the handle is created (kSlabWidth floats per chunk); nothing is allocated or freed inside
  // an asynchronous entry point, so calls are safe under stream capture.
  const size_t need = static_cast<size_t>(g->n_chunks) * D;
  if (need <= g->slab_floats) return TAGREC_OK;
  return fail(TAGREC_E_UNSUPPORTED, "long-row slab: row width " + std::to_string(D) + " exceeds the " +
                                        std::to_string(kSlabWidth) + " floats per chunk reserved at graph creation");
}

namespace {

template <int LPR, int EPI>
int launch_vec(const tagrec_graph* g, const float* X, const EpiArgs& e, hipStream_t s) {
  const GraphView gv{g->n_rows, g->rowptr, g->col, g->val, g->n_cols};
  const int threads = kWavesPerBlock * kWave;
  const unsigned blocks = static_cast<unsigned>((g->n_rows + kWavesPerBlock - 1) / kWavesPerBlock);
  LongView lv{g->long_rows, g->chunk_desc, g->n_chunks, nullptr, 0};
  if (g->n_long > 0) {
    int rc = ensure_slab(g, LPR * 4);
    if (rc != TAGREC_OK) return rc;
    lv.slab = g->slab;
    lv.chunk_blocks = static_cast<unsigned>((g->n_chunks + kWavesPerBlock - 1) / kWavesPerBlock);
  }
  if (e.row_mask) {                       // rows whose mask byte is 0 are left alone (a separate instantiation)
    constexpr bool kGroupable = (EPI == EPI_NORMBWD || EPI == EPI_AXPY) && LPR <= 32;
    bool grouped = false;
    if constexpr (kGroupable) {
      // flags always consulted (in_count == NULL) = the hop below the top layer: a handful of flagged operand rows
      if (e.in_flags && !e.in_count && !e.adam.p) {
        constexpr int rows_per_block = kWavesPerBlock * (kWave / LPR);
        const unsigned gblocks = static_cast<unsigned>((g->n_rows + rows_per_block - 1) / rows_per_block);
        spmm_rows_grouped_kernel<LPR, EPI><<<gblocks + lv.chunk_blocks, threads, 0, s>>>(gv, X, e, lv);
        grouped = true;
      }
    }
    if (!grouped) spmm_rows_kernel<LPR, EPI, true><<<blocks + lv.chunk_blocks, threads, 0, s>>>(gv, X, e, lv);
    TAGREC_LAUNCH_CHECK();
    if (g->n_long > 0) {
      const unsigned fblocks = static_cast<unsigned>((g->n_long + kWavesPerBlock - 1) / kWavesPerBlock);
      spmm_finish_kernel<LPR, EPI, true><<<fblocks, threads, 0, s>>>(gv, g->long_rows, g->long_base, g->n_long, g->slab, e);
      TAGREC_LAUNCH_CHECK();
    }
    return TAGREC_OK;
  }
  spmm_rows_kernel<LPR, EPI><<<blocks + lv.chunk_blocks, threads, 0, s>>>(gv, X, e, lv);
  TAGREC_LAUNCH_CHECK();
  if (g->n_long > 0) {
    const unsigned fblocks = static_cast<unsigned>((g->n_long + kWavesPerBlock - 1) / kWavesPerBlock);
    spmm_finish_kernel<LPR, EPI><<<fblocks, threads, 0, s>>>(gv, g->long_rows, g->long_base, g->n_long, g->slab, e);
    TAGREC_LAUNCH_CHECK();
  }
  return TAGREC_OK;
}

template <int EPI>
int launch_spmm(const tagrec_graph* g, const float* X, const EpiArgs& e, int D, void* stream, const char* who) {
  TAGREC_REQUIRE(g != nullptr, std::string(who) + ": null graph handle");
  TAGREC_REQUIRE(X != nullptr && e.Y != nullptr, std::string(who) + ": null X or output");
  TAGREC_REQUIRE(D >= 1, std::string(who) + ": D must be >= 1");
  TAGREC_REQUIRE(static_cast<const void*>(X) != static_cast<const void*>(e.Y), std::string(who) + ": output aliases the gathered input");
  if (g->n_rows == 0) return TAGREC_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  bool vec_ok = aligned16(X) && aligned16(e.Y);
  if (e.accum) vec_ok = vec_ok && aligned16(e.accum);
  if (e.Xraw) vec_ok = vec_ok && aligned16(e.Xraw);
  if (e.B) vec_ok = vec_ok && aligned16(e.B);
  if (e.drop.p > 0.f && !vec_ok) return fail(TAGREC_E_INVALID, std::string(who) + ": dropout needs 16-byte aligned rows");
  if (vec_ok) {
    switch (D) {
      case 8: return launch_vec<2, EPI>(g, X, e, s);
      case 16: return launch_vec<4, EPI>(g, X, e, s);
      case 32: return launch_vec<8, EPI>(g, X, e, s);
      case 64: return launch_vec<16, EPI>(g, X, e, s);
      case 128: return launch_vec<32, EPI>(g, X, e, s);
      case 256: return launch_vec<64, EPI>(g, X, e, s);
      default: break;
    }
  }
  if (EPI == EPI_NORM_ACC && !e.accum)
    return fail(TAGREC_E_INVALID, std::string(who) + ": acc may be NULL only with the vector kernels (D in {8,...,256}, aligned rows)");
  if (e.adam.p)       // the scalar kernel stores the raw gradient into e.Y == p and never touches m / v
    return fail(TAGREC_E_INVALID, std::string(who) + ": the fused Adam epilogue needs the vector kernels (D in {8,...,256}, "
                                                     "every operand 16-byte aligned)");
  if (e.row_mask || e.in_flags || e.out_flags || e.b_flags)
    return fail(TAGREC_E_INVALID, std::string(who) + ": row masks / row flags need the vector kernels (D in {8,...,256}, "
                                                     "every operand 16-byte aligned); the scalar kernel would ignore them");
  if (D > kMaxGenericBlocks * kWave)
    return fail(TAGREC_E_UNSUPPORTED, std::string(who) + ": row width " + std::to_string(D) + " > 512 is not covered");
  const GraphView gv{g->n_rows, g->rowptr, g->col, g->val, g->n_cols};
  const unsigned blocks = static_cast<unsigned>((g->n_rows + kWavesPerBlock - 1) / kWavesPerBlock);
  spmm_rows_generic_kernel<EPI><<<blocks, kWavesPerBlock * kWave, 0, s>>>(gv, X, e, D);
  TAGREC_LAUNCH_CHECK();
  return TAGREC_OK;
}

}  // namespace

extern "C" int tagrec_spmm_f32(const tagrec_graph* g, const float* X, float* Y, int D, void* stream) {
  EpiArgs e{Y, nullptr, nullptr, nullptr, nullptr, nullptr, 0.f, DropMask{0.f, 0}, nullptr, nullptr, nullptr, nullptr};
  return launch_spmm<EPI_NONE>(g, X, e, D, stream, "spmm");
}

extern "C" int tagrec_spmm_norm_acc_f32(const tagrec_graph* g, const float* X, float* Y_raw, float* inv_norm,
                                        float* acc, float acc_scale, int D, void* stream) {
  TAGREC_REQUIRE(inv_norm != nullptr && acc != nullptr, "spmm_norm_acc: null inv_norm or acc");
  EpiArgs e{Y_raw, inv_norm, acc, nullptr, nullptr, nullptr, acc_scale, DropMask{0.f, 0}, nullptr, nullptr, nullptr, nullptr};
  return launch_spmm<EPI_NORM_ACC>(g, X, e, D, stream, "spmm_norm_acc");
}

extern "C" int tagrec_spmm_normbwd_f32(const tagrec_graph* g, const float* G_in, const float* X_raw,
                                       const float* inv_norm, const float* dZ, float d_scale, float* G_out,
                                       int D, void* stream) {
  TAGREC_REQUIRE(X_raw != nullptr && inv_norm != nullptr && dZ != nullptr, "spmm_normbwd: null X_raw, inv_norm or dZ");
  EpiArgs e{G_out, const_cast<float*>(inv_norm), nullptr, X_raw, dZ, nullptr, d_scale, DropMask{0.f, 0}, nullptr, nullptr, nullptr, nullptr};
  return launch_spmm<EPI_NORMBWD>(g, G_in, e, D, stream, "spmm_normbwd");
}

extern "C" int tagrec_spmm_axpy_f32(const tagrec_graph* g, const float* G_in, const float* B, float b_scale,
                                    float* G_out, int D, void* stream) {
  TAGREC_REQUIRE(B != nullptr, "spmm_axpy: null B");
  EpiArgs e{G_out, nullptr, nullptr, nullptr, B, nullptr, b_scale, DropMask{0.f, 0}, nullptr, nullptr, nullptr, nullptr};
  return launch_spmm<EPI_AXPY>(g, G_in, e, D, stream, "spmm_axpy");
}

extern "C" int tagrec_spmm_ss_f32(const tagrec_graph* g, const float* X, float* Y, float* ss, int D, void* stream) {
  TAGREC_REQUIRE(ss != nullptr, "spmm_ss: null ss");
  EpiArgs e{Y, ss, nullptr, nullptr, nullptr, nullptr, 0.f, DropMask{0.f, 0}, nullptr, nullptr, nullptr, nullptr};
  return launch_spmm<EPI_SS>(g, X, e, D, stream, "spmm_ss");
}

extern "C" int tagrec_spmm_normbwd_dot_f32(const tagrec_graph* g, const float* G_in, const float* X_raw,
                                           const float* inv_norm, const float* dZ, const float* dot, float d_scale,
                                           float* G_out, int D, void* stream) {
  TAGREC_REQUIRE(X_raw != nullptr && inv_norm != nullptr && dZ != nullptr && dot != nullptr,
                 "spmm_normbwd_dot: null X_raw, inv_norm, dZ or dot");
  EpiArgs e{G_out, const_cast<float*>(inv_norm), nullptr, X_raw, dZ, dot, d_scale, DropMask{0.f, 0}, nullptr, nullptr, nullptr, nullptr};
  return launch_spmm<EPI_NORMBWD_DOT>(g, G_in, e, D, stream, "spmm_normbwd_dot");
}

extern "C" int tagrec_spmm_norm_acc_drop_f32(const tagrec_graph* g, const float* X, float* Y_raw, float* inv_norm,
                                             float* acc, float acc_scale, float drop_p, uint64_t seed, int D, void* stream) {
  TAGREC_REQUIRE(inv_norm != nullptr && acc != nullptr, "spmm_norm_acc_drop: null inv_norm or acc");
  TAGREC_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "spmm_norm_acc_drop: p must be in [0, 1)");
  TAGREC_REQUIRE(drop_p == 0.f || (D % 4 == 0 && D <= 256 && (D & (D - 1)) == 0 && D >= 8),
                 "spmm_norm_acc_drop: dropout needs a vector-kernel width (8..256, power of two)");
  EpiArgs e{Y_raw, inv_norm, acc, nullptr, nullptr, nullptr, acc_scale, DropMask{drop_p, seed}, nullptr, nullptr, nullptr, nullptr};
  return launch_spmm<EPI_NORM_ACC>(g, X, e, D, stream, "spmm_norm_acc_drop");
}

extern "C" int tagrec_spmm_normbwd_drop_f32(const tagrec_graph* g, const float* G_in, const float* X_raw,
                                            const float* inv_norm, const float* dZ, float d_scale, float drop_p,
                                            uint64_t seed, float* G_out, int D, void* stream) {
  TAGREC_REQUIRE(X_raw != nullptr && inv_norm != nullptr && dZ != nullptr, "spmm_normbwd_drop: null X_raw, inv_norm or dZ");
  TAGREC_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "spmm_normbwd_drop: p must be in [0, 1)");
  TAGREC_REQUIRE(drop_p == 0.f || (D % 4 == 0 && D <= 256 && (D & (D - 1)) == 0 && D >= 8),
                 "spmm_normbwd_drop: dropout needs a vector-kernel width (8..256, power of two)");
  EpiArgs e{G_out, const_cast<float*>(inv_norm), nullptr, X_raw, dZ, nullptr, d_scale, DropMask{drop_p, seed}, nullptr, nullptr, nullptr, nullptr};
  return launch_spmm<EPI_NORMBWD>(g, G_in, e, D, stream, "spmm_normbwd_drop");
}

// ---- backward layers on a row-sparse gradient (see EpiArgs::in_flags) -----------------------------------------------
extern "C" int tagrec_spmm_normbwd_sparse_f32(const tagrec_graph* g, const float* G_in, const uint8_t* in_flags,
                                              const unsigned* in_count, const float* X_raw, const float* inv_norm,
                                              const float* dZ, float d_scale, float drop_p, uint64_t seed, float* G_out,
                                              uint8_t* out_flags, unsigned* out_count, const uint8_t* row_mask,
                                              const uint8_t* dz_flags, int D, void* stream) {
  TAGREC_REQUIRE(X_raw != nullptr && inv_norm != nullptr && dZ != nullptr, "spmm_normbwd_sparse: null X_raw, inv_norm or dZ");
  TAGREC_REQUIRE(in_flags != nullptr || in_count == nullptr, "spmm_normbwd_sparse: in_count without in_flags");
  TAGREC_REQUIRE((out_flags == nullptr) == (out_count == nullptr), "spmm_normbwd_sparse: out_flags and out_count go together");
  TAGREC_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "spmm_normbwd_sparse: p must be in [0, 1)");
  TAGREC_REQUIRE(D == 8 || D == 16 || D == 32 || D == 64 || D == 128 || D == 256, "spmm_normbwd_sparse: D must be 8 .. 256, a power of two");
  EpiArgs e{G_out, const_cast<float*>(inv_norm), nullptr, X_raw, dZ, nullptr, d_scale, DropMask{drop_p, seed}, in_flags, in_count, out_flags, row_mask, dz_flags};
  int rc = launch_spmm<EPI_NORMBWD>(g, G_in, e, D, stream, "spmm_normbwd_sparse");
  if (rc != TAGREC_OK || !out_flags) return rc;
  return count_flags(out_flags, g->n_rows, out_count, static_cast<hipStream_t>(stream));
}

extern "C" int tagrec_spmm_axpy_sparse_f32(const tagrec_graph* g, const float* G_in, const uint8_t* in_flags,
                                           const unsigned* in_count, const float* B, float b_scale, float* G_out,
                                           const uint8_t* row_mask, const uint8_t* b_flags, int D, void* stream) {
  TAGREC_REQUIRE(B != nullptr, "spmm_axpy_sparse: null B");
  TAGREC_REQUIRE(in_flags != nullptr || in_count == nullptr, "spmm_axpy_sparse: in_count without in_flags");
  TAGREC_REQUIRE(D == 8 || D == 16 || D == 32 || D == 64 || D == 128 || D == 256, "spmm_axpy_sparse: D must be 8 .. 256, a power of two");
  EpiArgs e{G_out, nullptr, nullptr, nullptr, B, nullptr, b_scale, DropMask{0.f, 0}, in_flags, in_count, nullptr, row_mask, b_flags};
  return launch_spmm<EPI_AXPY>(g, G_in, e, D, stream, "spmm_axpy_sparse");
}

// The last hop of a training step with the optimizer folded in: row r of (A G_in + b_scale B) is the gradient of
// parameter row r, and Adam (torch.optim.Adam defaults, same arithmetic as tagrec_adam_f32 at step `step`) is applied
// to p / m / v right there; no gradient tensor is written.  The gathered operand must not alias p.
extern "C" int tagrec_spmm_axpy_adam_f32(const tagrec_graph* g, const float* G_in, const uint8_t* in_flags,
                                         const unsigned* in_count, const float* B, float b_scale, const uint8_t* b_flags,
                                         float* p, float* m, float* v, float lr, float b1, float b2, float eps, int64_t step,
                                         int D, void* stream) {
  TAGREC_REQUIRE(B != nullptr && p != nullptr && m != nullptr && v != nullptr, "spmm_axpy_adam: null pointer");
  TAGREC_REQUIRE(in_flags != nullptr || in_count == nullptr, "spmm_axpy_adam: in_count without in_flags");
  TAGREC_REQUIRE(step >= 1, "spmm_axpy_adam: step counts from 1");
  TAGREC_REQUIRE(D == 8 || D == 16 || D == 32 || D == 64 || D == 128 || D == 256, "spmm_axpy_adam: D must be 8 .. 256, a power of two");
  TAGREC_REQUIRE(static_cast<const void*>(G_in) != static_cast<const void*>(p), "spmm_axpy_adam: the gathered operand aliases the parameters");
  TAGREC_REQUIRE(aligned16(p) && aligned16(m) && aligned16(v) && aligned16(G_in) && aligned16(B),
                 "spmm_axpy_adam: every operand (p, m, v, G_in, B) must be 16-byte aligned");
  const double bc1 = 1.0 - pow(static_cast<double>(b1), static_cast<double>(step));      // as tagrec_adam_f32
  const double bc2 = 1.0 - pow(static_cast<double>(b2), static_cast<double>(step));
  EpiArgs e{p, nullptr, nullptr, nullptr, B, nullptr, b_scale, DropMask{0.f, 0}, in_flags, in_count, nullptr, nullptr, b_flags};
  e.adam = AdamRow{p, m, v, static_cast<float>(1.0 - static_cast<double>(b1)), b2, static_cast<float>(1.0 - static_cast<double>(b2)),
                   static_cast<float>(static_cast<double>(lr) / bc1), static_cast<float>(sqrt(bc2)), eps};
  return launch_spmm<EPI_AXPY>(g, G_in, e, D, stream, "spmm_axpy_adam");
}

// The same with the step counter and the two step-dependent factors in DEVICE memory (advanced by a one-thread kernel in
// front of the product), so the whole training step -- optimizer included -- can be captured once as a HIP graph and replayed.
extern "C" int tagrec_adam_advance(int64_t* step_dev, float* coef_dev, float lr, float b1, float b2, void* stream);
extern "C" int tagrec_spmm_axpy_adam_graph_f32(const tagrec_graph* g, const float* G_in, const uint8_t* in_flags,
                                               const unsigned* in_count, const float* B, float b_scale, const uint8_t* b_flags,
                                               float* p, float* m, float* v, float lr, float b1, float b2, float eps,
                                               int64_t* step_dev, float* coef_dev, int D, void* stream) {
  TAGREC_REQUIRE(B != nullptr && p != nullptr && m != nullptr && v != nullptr && step_dev != nullptr && coef_dev != nullptr,
                 "spmm_axpy_adam_graph: null pointer");
  TAGREC_REQUIRE(in_flags != nullptr || in_count == nullptr, "spmm_axpy_adam_graph: in_count without in_flags");
  TAGREC_REQUIRE(D == 8 || D == 16 || D == 32 || D == 64 || D == 128 || D == 256, "spmm_axpy_adam_graph: D must be 8 .. 256, a power of two");
  TAGREC_REQUIRE(static_cast<const void*>(G_in) != static_cast<const void*>(p), "spmm_axpy_adam_graph: the gathered operand aliases the parameters");
  TAGREC_REQUIRE(aligned16(p) && aligned16(m) && aligned16(v) && aligned16(G_in) && aligned16(B),
                 "spmm_axpy_adam_graph: every operand (p, m, v, G_in, B) must be 16-byte aligned");
  int rc = tagrec_adam_advance(step_dev, coef_dev, lr, b1, b2, stream);
  if (rc != TAGREC_OK) return rc;
  EpiArgs e{p, nullptr, nullptr, nullptr, B, nullptr, b_scale, DropMask{0.f, 0}, in_flags, in_count, nullptr, nullptr, b_flags};
  e.adam = AdamRow{p, m, v, static_cast<float>(1.0 - static_cast<double>(b1)), b2, static_cast<float>(1.0 - static_cast<double>(b2)),
                   0.f, 1.f, eps, coef_dev};
  return launch_spmm<EPI_AXPY>(g, G_in, e, D, stream, "spmm_axpy_adam_graph");
}

extern "C" int tagrec_spmm_normbwd_dot_sparse_f32(const tagrec_graph* g, const float* G_in, const uint8_t* in_flags,
                                                  const unsigned* in_count, const float* X_raw, const float* inv_norm,
                                                  const float* dZ, const float* dot, float d_scale, float* G_out,
                                                  uint8_t* out_flags, unsigned* out_count, const uint8_t* row_mask, int D,
                                                  void* stream) {
  TAGREC_REQUIRE(X_raw != nullptr && inv_norm != nullptr && dZ != nullptr && dot != nullptr,
                 "spmm_normbwd_dot_sparse: null X_raw, inv_norm, dZ or dot");
  TAGREC_REQUIRE(in_flags != nullptr && in_count != nullptr, "spmm_normbwd_dot_sparse: null in_flags or in_count");
  TAGREC_REQUIRE((out_flags == nullptr) == (out_count == nullptr), "spmm_normbwd_dot_sparse: out_flags and out_count go together");
  TAGREC_REQUIRE(D == 8 || D == 16 || D == 32 || D == 64 || D == 128 || D == 256, "spmm_normbwd_dot_sparse: D must be 8 .. 256, a power of two");
  EpiArgs e{G_out, const_cast<float*>(inv_norm), nullptr, X_raw, dZ, dot, d_scale, DropMask{0.f, 0}, in_flags, in_count, out_flags, row_mask};
  int rc = launch_spmm<EPI_NORMBWD_DOT>(g, G_in, e, D, stream, "spmm_normbwd_dot_sparse");
  if (rc != TAGREC_OK || !out_flags) return rc;
  return count_flags(out_flags, g->n_rows, out_count, static_cast<hipStream_t>(stream));
}

// Plain product on a row-sparse operand, writing the row flags of its result: the backward hop of a COLUMN-sharded table
// (the normalize-backward term lives on the batch rows and needs row dots over every rank's columns, so the caller adds
// it to those rows afterwards).
extern "C" int tagrec_spmm_flags_f32(const tagrec_graph* g, const float* G_in, const uint8_t* in_flags, const unsigned* in_count,
                                     float* G_out, uint8_t* out_flags, unsigned* out_count, const uint8_t* row_mask, int D,
                                     void* stream) {
  TAGREC_REQUIRE(in_flags != nullptr || in_count == nullptr, "spmm_flags: in_count without in_flags");
  TAGREC_REQUIRE(out_flags != nullptr || out_count == nullptr, "spmm_flags: out_count without out_flags");
  TAGREC_REQUIRE(D == 8 || D == 16 || D == 32 || D == 64 || D == 128 || D == 256, "spmm_flags: D must be 8 .. 256, a power of two");
  EpiArgs e{G_out, nullptr, nullptr, nullptr, nullptr, nullptr, 0.f, DropMask{0.f, 0}, in_flags, in_count, out_flags, row_mask, nullptr};
  int rc = launch_spmm<EPI_NONE>(g, G_in, e, D, stream, "spmm_flags");
  if (rc != TAGREC_OK || !out_count) return rc;
  return count_flags(out_flags, g->n_rows, out_count, static_cast<hipStream_t>(stream));
}

// ---- forward layer on a subset of the output rows -------------------------------------------------------------------
extern "C" int tagrec_graph_mark_rows_u8(const tagrec_graph* g, const int64_t* rows, int64_t n_listed, uint8_t* flags,
                                         void* stream) {
  TAGREC_REQUIRE(g != nullptr && flags != nullptr && (n_listed == 0 || rows != nullptr), "graph_mark_rows: null pointer");
  TAGREC_REQUIRE(g->n_rows == g->n_cols, "graph_mark_rows: square adjacency expected (flags are indexed by node)");
  TAGREC_REQUIRE(n_listed >= 0 && n_listed * kListedSplits < (1ll << 31), "graph_mark_rows: bad row count");
  if (n_listed == 0) return TAGREC_OK;
  const GraphView gv{g->n_rows, g->rowptr, g->col, g->val, g->n_cols};
  mark_rows_kernel<true><<<static_cast<unsigned>(n_listed * kListedSplits), 256, 0, static_cast<hipStream_t>(stream)>>>(gv, rows, n_listed, flags);
  TAGREC_LAUNCH_CHECK();
  return TAGREC_OK;
}

extern "C" int tagrec_graph_mark_cols_u8(const tagrec_graph* g, const int64_t* rows, int64_t n_listed, uint8_t* flags,
                                         void* stream) {
  TAGREC_REQUIRE(g != nullptr && flags != nullptr && (n_listed == 0 || rows != nullptr), "graph_mark_cols: null pointer");
  TAGREC_REQUIRE(n_listed >= 0 && n_listed * kListedSplits < (1ll << 31), "graph_mark_cols: bad row count");
  if (n_listed == 0) return TAGREC_OK;
  const GraphView gv{g->n_rows, g->rowptr, g->col, g->val, g->n_cols};
  mark_rows_kernel<false><<<static_cast<unsigned>(n_listed * kListedSplits), 256, 0, static_cast<hipStream_t>(stream)>>>(gv, rows, n_listed, flags);
  TAGREC_LAUNCH_CHECK();
  return TAGREC_OK;
}

extern "C" int64_t tagrec_spmm_listed_workspace(int64_t n_listed, int D) {
  return n_listed < 0 || D < 1 ? 0 : n_listed * kListedSplits * static_cast<int64_t>(D);
}

extern "C" int tagrec_spmm_listed_f32(const tagrec_graph* g, const int64_t* rows, int64_t n_listed, const float* X, float* Y,
                                      int D, float* ws, int64_t ws_floats, void* stream) {
  TAGREC_REQUIRE(g != nullptr && X != nullptr && Y != nullptr && (n_listed == 0 || rows != nullptr), "spmm_listed: null pointer");
  TAGREC_REQUIRE(n_listed >= 0 && n_listed * kListedSplits < (1ll << 31), "spmm_listed: bad row count");
  TAGREC_REQUIRE(aligned16(X) && aligned16(Y) && aligned16(ws), "spmm_listed: rows must be 16-byte aligned");
  TAGREC_REQUIRE(n_listed == 0 || (ws != nullptr && ws_floats >= tagrec_spmm_listed_workspace(n_listed, D)),
                 "spmm_listed: workspace smaller than tagrec_spmm_listed_workspace(n_listed, D)");
  if (n_listed == 0) return TAGREC_OK;
  const GraphView gv{g->n_rows, g->rowptr, g->col, g->val, g->n_cols};
  hipStream_t s = static_cast<hipStream_t>(stream);
  const unsigned blocks = static_cast<unsigned>(n_listed * kListedSplits);
  const int threads = kWavesPerBlock * kWave;
#define LISTED(L)                                                                                   \
  spmm_listed_kernel<L><<<blocks, threads, 0, s>>>(gv, rows, X, ws);                                \
  TAGREC_LAUNCH_CHECK();                                                                            \
  spmm_listed_fold_kernel<L><<<static_cast<unsigned>((n_listed * L + 255) / 256), 256, 0, s>>>(ws, Y, n_listed); \
  break
  switch (D) {
    case 8: LISTED(2);
    case 16: LISTED(4);
    case 32: LISTED(8);
    case 64: LISTED(16);
    case 128: LISTED(32);
    case 256: LISTED(64);
    default: return fail(TAGREC_E_UNSUPPORTED, "spmm_listed: D must be 8 .. 256, a power of two");
  }
#undef LISTED
  TAGREC_LAUNCH_CHECK();
  return TAGREC_OK;
}

extern "C" int tagrec_spmm_norm_acc_rows_f32(const tagrec_graph* g, const float* X, float* Y_raw, float* inv_norm,
                                             float* acc, float acc_scale, const uint8_t* row_mask, float drop_p,
                                             uint64_t seed, int D, void* stream) {
  TAGREC_REQUIRE(inv_norm != nullptr, "spmm_norm_acc_rows: null inv_norm");
  TAGREC_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "spmm_norm_acc_rows: p must be in [0, 1)");
  TAGREC_REQUIRE(D == 8 || D == 16 || D == 32 || D == 64 || D == 128 || D == 256, "spmm_norm_acc_rows: D must be 8 .. 256, a power of two");
  EpiArgs e{Y_raw, inv_norm, acc, nullptr, nullptr, nullptr, acc_scale, DropMask{drop_p, seed}, nullptr, nullptr, nullptr, row_mask, nullptr};
  return launch_spmm<EPI_NORM_ACC>(g, X, e, D, stream, "spmm_norm_acc_rows");
}

extern "C" int tagrec_spmm_rows_f32(const tagrec_graph* g, const float* X, float* Y, const uint8_t* row_mask, int D, void* stream) {
  TAGREC_REQUIRE(row_mask != nullptr, "spmm_rows: null row_mask");
  TAGREC_REQUIRE(D == 8 || D == 16 || D == 32 || D == 64 || D == 128 || D == 256, "spmm_rows: D must be 8 .. 256, a power of two");
  EpiArgs e{Y, nullptr, nullptr, nullptr, nullptr, nullptr, 0.f, DropMask{0.f, 0}, nullptr, nullptr, nullptr, row_mask};
  return launch_spmm<EPI_NONE>(g, X, e, D, stream, "spmm_rows");
}

extern "C" int tagrec_spmm_ss_rows_f32(const tagrec_graph* g, const float* X, float* Y, float* ss, const uint8_t* row_mask, int D,
                                       void* stream) {
  TAGREC_REQUIRE(ss != nullptr && row_mask != nullptr, "spmm_ss_rows: null ss or row_mask");
  TAGREC_REQUIRE(D == 8 || D == 16 || D == 32 || D == 64 || D == 128 || D == 256, "spmm_ss_rows: D must be 8 .. 256, a power of two");
  EpiArgs e{Y, ss, nullptr, nullptr, nullptr, nullptr, 0.f, DropMask{0.f, 0}, nullptr, nullptr, nullptr, row_mask};
  return launch_spmm<EPI_SS>(g, X, e, D, stream, "spmm_ss_rows");
}

// ---- TGCN attention backward pulls (see attn_pull_da_kernel / attn_pull_dq_kernel) ------------------------------------
namespace {
template <int LPR, int EPI>
int launch_pull_da(const tagrec_graph* g, const int32_t* pair, const float* dOut, const float* Ej, float* da, const EpiArgs& e,
                   hipStream_t s) {
  const GraphView gv{g->n_rows, g->rowptr, g->col, g->val, g->n_cols};
  const int threads = kWavesPerBlock * kWave;
  constexpr int rows_per_block = kWavesPerBlock * (kWave / LPR);          // every lane group of a wave walks its own row
  const unsigned blocks = static_cast<unsigned>((g->n_rows + rows_per_block - 1) / rows_per_block);
  LongView lv{g->long_rows, g->chunk_desc, g->n_chunks, nullptr, 0};
  if (g->n_long > 0) {
    int rc = ensure_slab(g, LPR * 4);
    if (rc != TAGREC_OK) return rc;
    lv.slab = g->slab;
    lv.chunk_blocks = static_cast<unsigned>((g->n_chunks + kWavesPerBlock - 1) / kWavesPerBlock);
  }
  attn_pull_da_kernel<LPR, EPI><<<blocks + lv.chunk_blocks, threads, 0, s>>>(gv, pair, dOut, Ej, da, e, lv);
  TAGREC_LAUNCH_CHECK();
  if (g->n_long > 0) {
    const unsigned fblocks = static_cast<unsigned>((g->n_long + kWavesPerBlock - 1) / kWavesPerBlock);
    spmm_finish_kernel<LPR, EPI><<<fblocks, threads, 0, s>>>(gv, g->long_rows, g->long_base, g->n_long, g->slab, e);
    TAGREC_LAUNCH_CHECK();
  }
  return TAGREC_OK;
}

template <int A, int EPI>
int launch_pull_dq(const tagrec_graph* g, const float2* comp, const float* vv, const EpiArgs& e, hipStream_t s) {
  constexpr int LPR = A / 4;
  const GraphView gv{g->n_rows, g->rowptr, g->col, g->val, g->n_cols};
  const int threads = kWavesPerBlock * kWave;
  const unsigned blocks = static_cast<unsigned>((g->n_rows + kWavesPerBlock - 1) / kWavesPerBlock);
  LongView lv{g->long_rows, g->chunk_desc, g->n_chunks, nullptr, 0};
  if (g->n_long > 0) {
    int rc = ensure_slab(g, A);
    if (rc != TAGREC_OK) return rc;
    lv.slab = g->slab;
    lv.chunk_blocks = static_cast<unsigned>((g->n_chunks + kWavesPerBlock - 1) / kWavesPerBlock);
  }
  attn_pull_dq_kernel<A, EPI><<<blocks + lv.chunk_blocks, threads, 0, s>>>(gv, comp, vv, e, lv);
  TAGREC_LAUNCH_CHECK();
  if (g->n_long > 0) {
    const unsigned fblocks = static_cast<unsigned>((g->n_long + kWavesPerBlock - 1) / kWavesPerBlock);
    spmm_finish_kernel<LPR, EPI><<<fblocks, threads, 0, s>>>(gv, g->long_rows, g->long_base, g->n_long, g->slab, e);
    TAGREC_LAUNCH_CHECK();
  }
  return TAGREC_OK;
}
}  // namespace

extern "C" int tagrec_attn_pull_da_f32(const tagrec_graph* g, const int32_t* pair, const float* dOut, const float* Ej,
                                       const float* B, float* dEj, float* da, int D, void* stream) {
  TAGREC_REQUIRE(g != nullptr, "attn_pull_da: null graph handle");
  if (g->n_rows == 0) return TAGREC_OK;
  TAGREC_REQUIRE(Ej && dEj && (g->nnz == 0 || (dOut && pair && da)), "attn_pull_da: null pointer");
  TAGREC_REQUIRE(aligned16(dOut) && aligned16(Ej) && aligned16(dEj) && (!B || aligned16(B)), "attn_pull_da: rows must be 16-byte aligned");
  hipStream_t s = static_cast<hipStream_t>(stream);
  EpiArgs e{dEj, nullptr, nullptr, nullptr, B, nullptr, 1.0f, DropMask{0.f, 0}, nullptr, nullptr, nullptr, nullptr, nullptr};
#define PULL(L) (B ? launch_pull_da<L, EPI_AXPY>(g, pair, dOut, Ej, da, e, s) : launch_pull_da<L, EPI_NONE>(g, pair, dOut, Ej, da, e, s))
  switch (D) {
    case 16: return PULL(4);
    case 32: return PULL(8);
    case 64: return PULL(16);
    case 128: return PULL(32);
    case 256: return PULL(64);
    default: break;
  }
#undef PULL
  return fail(TAGREC_E_UNSUPPORTED, "attn_pull_da: D must be 16, 32, 64, 128 or 256");
}

extern "C" int tagrec_attn_pull_dq_f32(const tagrec_graph* g, const float* comp, const float* v, int A, const float* B, float* dQ,
                                       void* stream) {
  TAGREC_REQUIRE(g != nullptr, "attn_pull_dq: null graph handle");
  if (g->n_rows == 0) return TAGREC_OK;
  TAGREC_REQUIRE(v && dQ && (g->nnz == 0 || comp), "attn_pull_dq: null pointer");
  TAGREC_REQUIRE(aligned16(dQ) && (!B || aligned16(B)) && (reinterpret_cast<uintptr_t>(comp) & 7u) == 0, "attn_pull_dq: misaligned buffer");
  hipStream_t s = static_cast<hipStream_t>(stream);
  EpiArgs e{dQ, nullptr, nullptr, nullptr, B, nullptr, 1.0f, DropMask{0.f, 0}, nullptr, nullptr, nullptr, nullptr, nullptr};
  const float2* c2 = reinterpret_cast<const float2*>(comp);
#define PULL(AA) (B ? launch_pull_dq<AA, EPI_AXPY>(g, c2, v, e, s) : launch_pull_dq<AA, EPI_NONE>(g, c2, v, e, s))
  switch (A) {
    case 16: return PULL(16);
    case 32: return PULL(32);
    default: break;
  }
#undef PULL
  return fail(TAGREC_E_UNSUPPORTED, "attn_pull_dq: A must be 16 or 32 (the relu bits of a pair travel in one 32-bit word)");
}
