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
  // UB steps at a time: their Q / WT reads are all issued before the first is used (a wave walks one node, and the steps
  // were a chain of dependent memory latencies)
  constexpr int UB = 8;
  for (int n1 = 0; n1 < k; n1 += UB * npi) {
    float hq[UB], hw[UB];
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const int n = n1 + u * npi + grp;
      const int jn = __shfl(j, n & (kWave - 1));
      const int wn = __shfl(w, n & (kWave - 1));
      const bool in = n < k;
      hw[u] = in ? WT[static_cast<int64_t>(wn) * A + c] : 0.f;
      hq[u] = (in && jn) ? Q[static_cast<int64_t>(jn - 1) * A + c] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const int n0 = n1 + u * npi;
      if (n0 < k) {                                    // (wave-uniform)
        const int n = n0 + grp;
        float t = n < k ? fmaxf(pc + hw[u] + hq[u], 0.f) * vc : 0.f;
        for (int m = 1; m < A; m <<= 1) t += __shfl_xor(t, m);
        for (int gg = 0; gg < npi; ++gg) {
          const float tg = __shfl(t, gg * A);
          if (lane == n0 + gg && lane < k) s = tg;
        }
      }
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
  for (int n1 = 0; n1 < k; n1 += 4 * RPI) {            // four gather instructions in flight
    float4 x[4];
    float aa[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int nn = n1 + u * RPI + g;
      const int jj = __shfl(j, nn & (kWave - 1));
      const float av = __shfl(a, nn & (kWave - 1));
      const bool ok = nn < k && jj != 0;
      aa[u] = ok ? av : 0.f;
      x[u] = ok ? E4[static_cast<int64_t>(jj - 1) * LPR] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      acc.x = fmaf(aa[u], x[u].x, acc.x); acc.y = fmaf(aa[u], x[u].y, acc.y);
      acc.z = fmaf(aa[u], x[u].z, acc.z); acc.w = fmaf(aa[u], x[u].w, acc.w);
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

// Source-centric half of the backward pass when the D-wide work has been done by the destination-centric pull
// (attn_pull_da_kernel, csrc/spmm.hip): da[v, n] = dOut[v] . Ej[idx_n] arrives as an input, so this kernel touches no
// embedding row at all -- per (node, neighbour) it reads the A-wide Q row and writes EIGHT bytes: the softmax-backward
// scalar ds and the A relu bits of the pre-activation, from which the second pull (attn_pull_dq_kernel) rebuilds
// dh = ds v (.) [h > 0].  dP, dWT, dv as in the kernel above.  A <= 32.
__global__ __launch_bounds__(kAttnThreads) void tgcn_attn_bwd_ds_kernel(const float* __restrict__ P, const float* __restrict__ Q,
                                                                        const float* __restrict__ WT, const float* __restrict__ vv,
                                                                        const int32_t* __restrict__ idx, const int32_t* __restrict__ widx,
                                                                        const float* __restrict__ attn, const float* __restrict__ da,
                                                                        int64_t n, int k, int A, int n_wt, float* __restrict__ dP,
                                                                        float2* __restrict__ comp, float* __restrict__ part, int copies,
                                                                        int w_major) {
  // dWT partials in LDS, one private copy per (wave, lane group) when they fit (`copies` = 4 * 64 / A): edge weights are
  // mostly 1, so the lane groups of a wave -- and the waves of a block -- would otherwise add into the SAME few addresses
  // and serialise (measured: 56 % of this kernel's time with one shared copy).  Then [A] dv partial.
  extern __shared__ float sh[];
  const int wt_elems = n_wt * A;
  float* sh_v = sh + copies * wt_elems;
  for (int i = threadIdx.x; i < copies * wt_elems + A; i += kAttnThreads) sh[i] = 0.f;
  __syncthreads();
  const int lane = threadIdx.x & (kWave - 1);
  const int npi = kWave / A, grp = lane / A, c = lane % A;
  float* sh_wt = sh + (copies > 1 ? ((threadIdx.x >> 6) * npi + grp) * wt_elems : 0);
  const float vc = vv[c];
  const unsigned long long gmask = A >= 32 ? 0xFFFFFFFFull : ((1ull << A) - 1ull);
  float dv_acc = 0.f;
  float wt_major = 0.f;        // the share of dWT[w_major]: the most frequent edge weight stays in a register for the whole launch
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kAttnWaves;
  // the node's scalars (neighbour ids, weight ids, attention weights, da) are fetched one node ahead
  int64_t v = static_cast<int64_t>(blockIdx.x) * kAttnWaves + (threadIdx.x >> 6);
  int nj = 0, nw = 0;
  float na = 0.f, nd = 0.f, npc = 0.f;
  if (v < n) {
    if (lane < k) { nj = idx[v * k + lane]; nw = widx[v * k + lane]; na = attn[v * k + lane]; nd = da[v * k + lane]; }
    npc = P[v * A + c];
  }
  for (; v < n; v += stride) {
    const int j = nj, w = nw;
    const float a = na, d = (j != 0) ? nd : 0.f;   // a pad slot is a zero row: its da is 0 (the slot holds garbage)
    const float pc = npc;
    const int64_t vn = v + stride;
    if (vn < n) {
      if (lane < k) { nj = idx[vn * k + lane]; nw = widx[vn * k + lane]; na = attn[vn * k + lane]; nd = da[vn * k + lane]; }
      npc = P[vn * A + c];
    }
    const float dot = wave_add(a * d);
    const float ds = a * (d - dot);
    float dp_acc = 0.f;
    int my_bits = 0;                  // lane n < k collects the relu bits of neighbour n: one coalesced store at the end
    // UB steps at a time: all their Q / WT reads are issued before the first is used (the steps of a node are otherwise a
    // chain of dependent memory latencies, and a wave walks one node at a time)
    constexpr int UB = 8;
    for (int n0 = 0; n0 < k; n0 += UB * npi) {
      float hq[UB], hw[UB], dsn[UB];
      int wnv[UB];
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        const int nn = n0 + u * npi + grp;
        const int jn = __shfl(j, nn & (kWave - 1));
        wnv[u] = __shfl(w, nn & (kWave - 1));
        dsn[u] = __shfl(ds, nn & (kWave - 1));
        const bool in = nn < k;
        hw[u] = in ? WT[static_cast<int64_t>(wnv[u]) * A + c] : 0.f;
        hq[u] = (in && jn) ? Q[static_cast<int64_t>(jn - 1) * A + c] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        const int nn = n0 + u * npi + grp;
        if (n0 + u * npi < k) {                           // (wave-uniform)
          const float h = pc + hw[u] + hq[u];
          const bool pos = nn < k && h > 0.f;
          if (pos) {
            const float dhv = dsn[u] * vc;
            dp_acc += dhv;
            dv_acc = fmaf(dsn[u], h, dv_acc);
            if (wnv[u] == w_major) wt_major += dhv;
            else atomicAdd(&sh_wt[wnv[u] * A + c], dhv);
          }
          const unsigned long long m = __ballot(pos);
          const int first = n0 + u * npi;                 // this step covers neighbours [first, first + npi)
          if (lane >= first && lane < first + npi) my_bits = static_cast<int>((m >> ((lane - first) * A)) & gmask);
        }
      }
    }
    if (lane < k) comp[v * k + lane] = make_float2(ds, __int_as_float(my_bits));
    for (int m = A; m < kWave; m <<= 1) dp_acc += __shfl_xor(dp_acc, m);
    if (lane < A) dP[v * A + lane] = dp_acc;
  }
  for (int m = A; m < kWave; m <<= 1) dv_acc += __shfl_xor(dv_acc, m);
  if (lane < A) atomicAdd(&sh_v[lane], dv_acc);
  if (w_major >= 0 && w_major < n_wt) atomicAdd(&sh_wt[w_major * A + c], wt_major);
  __syncthreads();
  float* o = part + static_cast<int64_t>(blockIdx.x) * (n_wt + 1) * A;
  for (int i = threadIdx.x; i < wt_elems; i += kAttnThreads) {
    float t = 0.f;
    for (int cp = 0; cp < copies; ++cp) t += sh[cp * wt_elems + i];          // fixed order
    o[i] = t;
  }
  for (int i = threadIdx.x; i < A; i += kAttnThreads) o[wt_elems + i] = sh_v[i];
}

// dQ[j] += v (.) sum over the pairs that point at j of ds[p] bits[p], from the pair list SORTED by destination: every wave
// takes kSegEntries consecutive entries (perfectly balanced, no row pointer, no per-row latency floor -- a destination
// collects ~6 pairs on average and the row-per-wave pull spent its time waiting on rowptr -> pair -> comp chains), walks
// them with the destination in a scalar register and adds a finished segment to dQ with one 4 A-byte float-atomic
// instruction.  Destinations that straddle two waves are added twice (hence atomics; dQ is zeroed by the caller).
constexpr int kSegEntries = 256;
template <int A>
__global__ __launch_bounds__(256) void attn_seg_dq_kernel(const int32_t* __restrict__ dest, const int32_t* __restrict__ pair,
                                                           int64_t n_entries, int32_t n_dst, const float2* __restrict__ comp,
                                                           const float* __restrict__ vv, float* __restrict__ dQ) {
  const int lane = threadIdx.x & (kWave - 1);
  const int64_t wv = static_cast<int64_t>(blockIdx.x) * 4 + (threadIdx.x >> 6);
  const int64_t e0 = wv * kSegEntries;
  if (e0 >= n_entries) return;
  const int64_t e1 = e0 + kSegEntries < n_entries ? e0 + kSegEntries : n_entries;
  const int c = lane % A;
  const bool worker = lane < A;
  const float vc = vv[c];
  int cur = -1;
  float acc = 0.f;
  bool done = false;
  for (int64_t base = e0; base < e1 && !done; base += kWave) {
    const int n = (e1 - base) < kWave ? static_cast<int>(e1 - base) : kWave;
    int my_dest = n_dst;
    float my_ds = 0.f;
    int my_bits = 0;
    if (lane < n) {
      my_dest = dest[base + lane];
      if (my_dest < n_dst) {
        const float2 cv = comp[pair[base + lane]];
        my_ds = cv.x;
        my_bits = __float_as_int(cv.y);
      }
    }
    for (int i = 0; i < n; ++i) {
      const int d = __builtin_amdgcn_readlane(my_dest, i);
      if (d >= n_dst) { done = true; break; }            // pads sort behind the last destination: nothing follows
      if (d != cur) {
        if (cur >= 0 && worker) atomicAdd(&dQ[static_cast<int64_t>(cur) * A + c], acc * vc);
        acc = 0.f;
        cur = d;
      }
      const float ds = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_ds), i));
      const int bits = __builtin_amdgcn_readlane(my_bits, i);
      acc += ((bits >> c) & 1) ? ds : 0.f;
    }
  }
  if (cur >= 0 && worker) atomicAdd(&dQ[static_cast<int64_t>(cur) * A + c], acc * vc);
}

// after the sort: pair id (as int32), source row and attention weight of every sorted entry, in one streaming pass
__global__ __launch_bounds__(256) void attn_invert_fill_kernel(const int64_t* __restrict__ order, const float* __restrict__ attn, int k,
                                                                int64_t n, int32_t* __restrict__ pair, int32_t* __restrict__ src,
                                                                float* __restrict__ val) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * 256;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; i < n; i += stride) {
    const int64_t p = order[i];
    pair[i] = static_cast<int32_t>(p);
    src[i] = static_cast<int32_t>(p / k);
    val[i] = attn[p];
  }
}

// neighbour / weight ids of a row subset in compact numbering: out[i, :] = pos[idx[rows[i], :]] (pos == nullptr: ids as they
// are; 0 stays the pad), wout[i, :] = widx[rows[i], :] -- one pass instead of two index_selects and a look-up per relation
__global__ __launch_bounds__(256) void nbr_gather_kernel(const int32_t* __restrict__ idx, const int32_t* __restrict__ widx,
                                                          const int64_t* __restrict__ rows, const int32_t* __restrict__ pos,
                                                          int64_t n_rows, int k, int32_t* __restrict__ out, int32_t* __restrict__ wout) {
  const int64_t total = n_rows * k;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * 256;
  for (int64_t e = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; e < total; e += stride) {
    const int64_t i = e / k;
    const int64_t srcpos = rows[i] * k + (e - i * k);
    const int32_t j = idx[srcpos];
    out[e] = pos ? pos[j] : j;
    wout[e] = widx[srcpos];
  }
}

// ---- a step's inverted table WITHOUT a sort -----------------------------------------------------------------------------
// The neighbour tables are static, so every relation's pair list sorted by destination is built ONCE (model start-up).  A
// step needs the sub-list of the pairs whose SOURCE row is among the rows it computes: an order-preserving compaction of
// the static list which writes, already in destination order, everything the pulls read: destination (renumbered into the
// step's compact table), pair id p' = row' k + slot (row' = the source row's position in the step's row subset), source
// row, attention weight.  Replaces one radix sort per relation, layer and step (the library's kernels) plus the passes that
// followed it.
// ONE pass over the static list (it is 25 M entries per relation at C4, read twice by a count / scan / scatter scheme):
// blocks take tickets, publish their count and look back over their predecessors' published counts / prefixes (a
// "chained scan"); a block waits only for blocks with smaller tickets, which are running by construction (they drew
// their ticket earlier) and publish their own count BEFORE they look back, so every wait ends.
constexpr int kFiltThreads = 256;
constexpr int kFiltPerThread = 16;
constexpr int kFiltBlock = kFiltThreads * kFiltPerThread;     // static entries per block
constexpr unsigned long long kFiltValueMask = (1ull << 62) - 1;    // status word: flag (0 empty, 1 count, 2 prefix) << 62 | value

// p / k for 0 <= p < 2^31, 1 <= k <= 64 without a division: magic = floor(2^40 / k) + 1 (error term p (magic k - 2^40) < 2^40)
__device__ __forceinline__ int div_magic(int p, unsigned long long magic) {
  return static_cast<int>((static_cast<unsigned long long>(static_cast<unsigned>(p)) * magic) >> 40);
}

__device__ __forceinline__ bool filt_active(const int32_t* __restrict__ perm, const int32_t* __restrict__ pos_src, int k,
                                            unsigned long long magic, int64_t e, int64_t n_entries, int& row_c, int& slot) {
  if (e >= n_entries) return false;
  const int p = perm[e];
  const int v = div_magic(p, magic);
  slot = p - v * k;
  row_c = pos_src[v + 1] - 1;                                  // -1: the source row is not computed this step
  return row_c >= 0;
}

__global__ __launch_bounds__(kFiltThreads) void inv_filter_kernel(const int32_t* __restrict__ perm, const int32_t* __restrict__ dest,
                                                                   const int32_t* __restrict__ pos_src, const int32_t* __restrict__ pos_dst,
                                                                   const float* __restrict__ attn, int k, unsigned long long magic,
                                                                   int64_t n_entries, int32_t* __restrict__ ws, int n_blocks,
                                                                   int32_t* __restrict__ skey, int32_t* __restrict__ pair,
                                                                   int32_t* __restrict__ src, float* __restrict__ val) {
  // wave w of the block takes the contiguous quarter [w 1024, (w + 1) 1024) of the block's entries, 64 per pass: the 16
  // ballot masks and the active lanes' (row, slot) stay in registers
  __shared__ int wave_tot[kFiltThreads / 64];
  __shared__ int bid_s;
  __shared__ long long excl_s;
  unsigned* ticket = reinterpret_cast<unsigned*>(ws);
  unsigned long long* status = reinterpret_cast<unsigned long long*>(ws + 4);
  if (threadIdx.x == 0) bid_s = static_cast<int>(atomicAdd(ticket, 1u));
  __syncthreads();
  const int bid = bid_s;                                        // the block's place in the list = the order it started in
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t wbase = static_cast<int64_t>(bid) * kFiltBlock + static_cast<int64_t>(wave) * (64 * kFiltPerThread);
  int rcs[kFiltPerThread], sls[kFiltPerThread];
  unsigned long long masks[kFiltPerThread];
  int tot = 0;
#pragma unroll
  for (int i = 0; i < kFiltPerThread; ++i) {
    const bool act = filt_active(perm, pos_src, k, magic, wbase + i * 64 + lane, n_entries, rcs[i], sls[i]);
    masks[i] = __ballot(act);
    tot += __popcll(masks[i]);
  }
  if (lane == 0) wave_tot[wave] = tot;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long mine = 0;
    for (int w = 0; w < kFiltThreads / 64; ++w) mine += static_cast<unsigned long long>(wave_tot[w]);
    unsigned long long excl = 0;
    if (bid > 0) {
      __hip_atomic_store(&status[bid], (1ull << 62) | mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // count first, then look back
      for (int j = bid - 1;; --j) {                            // ends at the latest at block 0, which publishes a prefix at once
        unsigned long long st;
        while (((st = __hip_atomic_load(&status[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 62) == 0) __builtin_amdgcn_s_sleep(1);
        excl += st & kFiltValueMask;
        if ((st >> 62) == 2) break;
      }
    }
    __hip_atomic_store(&status[bid], (2ull << 62) | (excl + mine), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    excl_s = static_cast<long long>(excl);
    if (bid == n_blocks - 1) ws[1] = static_cast<int32_t>(excl + mine);        // the number of listed entries
  }
  __syncthreads();
  int64_t out = excl_s;
  for (int w = 0; w < wave; ++w) out += wave_tot[w];
#pragma unroll
  for (int i = 0; i < kFiltPerThread; ++i) {
    const unsigned long long m = masks[i];
    if ((m >> lane) & 1ull) {
      const int before = __builtin_amdgcn_mbcnt_hi(static_cast<unsigned>(m >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<unsigned>(m), 0u));
      const int64_t o = out + before;
      const int d = dest[wbase + i * 64 + lane];
      const int pc = rcs[i] * k + sls[i];
      skey[o] = pos_dst ? pos_dst[d + 1] - 1 : d;
      pair[o] = pc;
      src[o] = rcs[i];
      val[o] = attn[pc];
    }
    out += __popcll(m);
  }
}

// sort keys of an on-the-spot table inversion: destination row of every (node, slot) pair, pads behind the last row
__global__ __launch_bounds__(256) void attn_keys_kernel(const int32_t* __restrict__ idx, int64_t n, int32_t n_dst, int32_t* __restrict__ key) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * 256;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; i < n; i += stride) {
    const int32_t j = idx[i];
    key[i] = j > 0 ? j - 1 : n_dst;
  }
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

extern "C" int tagrec_tgcn_attn_bwd_ds_f32(const float* P, const float* Q, const float* WT, const float* v, const int32_t* idx,
                                           const int32_t* widx, const float* attn, const float* da, int64_t n, int k, int A,
                                           int n_wt, int w_major, float* dP, float* comp, float* dWT, float* dv, float* workspace,
                                           int64_t workspace_floats, void* stream) {
  TAGREC_REQUIRE(P && Q && WT && v && idx && widx && attn && da && dP && comp && dWT && dv && workspace, "tgcn_attn_bwd_ds: null pointer");
  TAGREC_REQUIRE((A == 4 || A == 8 || A == 16 || A == 32) && k >= 1 && k <= kWave, "tgcn_attn_bwd_ds: need A in {4,8,16,32}, 1 <= k <= 64");
  TAGREC_REQUIRE(n_wt >= 1 && static_cast<size_t>(n_wt + 1) * A * sizeof(float) <= 48 * 1024, "tgcn_attn_bwd_ds: weight table too large for LDS");
  TAGREC_REQUIRE(workspace_floats >= tagrec_tgcn_attn_workspace(n_wt, A), "tgcn_attn_bwd_ds: workspace too small");
  TAGREC_REQUIRE((reinterpret_cast<uintptr_t>(comp) & 7u) == 0, "tgcn_attn_bwd_ds: comp must be 8-byte aligned");
  if (n <= 0) return TAGREC_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int elems = (n_wt + 1) * A;
  int copies = kAttnWaves * (kWave / A);                        // one dWT partial per (wave, lane group) while that fits
  if (static_cast<size_t>(copies) * n_wt * A * sizeof(float) > 40 * 1024) copies = 1;
  const size_t lds = (static_cast<size_t>(copies) * n_wt * A + A) * sizeof(float);
  const int64_t want = (n + kAttnWaves - 1) / kAttnWaves;
  const unsigned blocks = static_cast<unsigned>(want < kAttnBlocks ? want : kAttnBlocks);
  tgcn_attn_bwd_ds_kernel<<<blocks, kAttnThreads, lds, s>>>(P, Q, WT, v, idx, widx, attn, da, n, k, A, n_wt, dP,
                                                             reinterpret_cast<float2*>(comp), workspace, copies, w_major);
  TAGREC_LAUNCH_CHECK();
  tgcn_attn_fold_kernel<<<(elems + 15) / 16, 256, 0, s>>>(workspace, static_cast<int>(blocks), elems, n_wt * A, dWT, dv);
  TAGREC_LAUNCH_CHECK();
  return TAGREC_OK;
}

extern "C" int tagrec_attn_keys_i32(const int32_t* idx, int64_t n, int32_t n_dst, int32_t* key, void* stream) {
  TAGREC_REQUIRE(n >= 0 && (n == 0 || (idx && key)), "attn_keys: null pointer");
  if (n == 0) return TAGREC_OK;
  int64_t blocks = (n + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;
  attn_keys_kernel<<<static_cast<unsigned>(blocks), 256, 0, static_cast<hipStream_t>(stream)>>>(idx, n, n_dst, key);
  TAGREC_LAUNCH_CHECK();
  return TAGREC_OK;
}

extern "C" int tagrec_attn_seg_dq_f32(const int32_t* dest_sorted, const int32_t* pair_sorted, int64_t n_entries, int32_t n_dst,
                                      const float* comp, const float* v, int A, float* dQ, void* stream) {
  TAGREC_REQUIRE(n_entries >= 0 && n_dst >= 0, "attn_seg_dq: bad size");
  if (n_entries == 0) return TAGREC_OK;
  TAGREC_REQUIRE(dest_sorted && pair_sorted && comp && v && dQ, "attn_seg_dq: null pointer");
  TAGREC_REQUIRE((reinterpret_cast<uintptr_t>(comp) & 7u) == 0, "attn_seg_dq: comp must be 8-byte aligned");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int64_t waves = (n_entries + kSegEntries - 1) / kSegEntries;
  const unsigned blocks = static_cast<unsigned>((waves + 3) / 4);
  const float2* c2 = reinterpret_cast<const float2*>(comp);
  switch (A) {
    case 4: attn_seg_dq_kernel<4><<<blocks, 256, 0, s>>>(dest_sorted, pair_sorted, n_entries, n_dst, c2, v, dQ); break;
    case 8: attn_seg_dq_kernel<8><<<blocks, 256, 0, s>>>(dest_sorted, pair_sorted, n_entries, n_dst, c2, v, dQ); break;
    case 16: attn_seg_dq_kernel<16><<<blocks, 256, 0, s>>>(dest_sorted, pair_sorted, n_entries, n_dst, c2, v, dQ); break;
    case 32: attn_seg_dq_kernel<32><<<blocks, 256, 0, s>>>(dest_sorted, pair_sorted, n_entries, n_dst, c2, v, dQ); break;
    default: return fail(TAGREC_E_UNSUPPORTED, "attn_seg_dq: A must be 4, 8, 16 or 32");
  }
  TAGREC_LAUNCH_CHECK();
  return TAGREC_OK;
}

extern "C" int tagrec_attn_invert_fill(const int64_t* order, const float* attn, int k, int64_t n, int32_t* pair, int32_t* src,
                                       float* val, void* stream) {
  TAGREC_REQUIRE(n >= 0 && k >= 1 && (n == 0 || (order && attn && pair && src && val)), "attn_invert_fill: null pointer");
  if (n == 0) return TAGREC_OK;
  int64_t blocks = (n + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;
  attn_invert_fill_kernel<<<static_cast<unsigned>(blocks), 256, 0, static_cast<hipStream_t>(stream)>>>(order, attn, k, n, pair, src, val);
  TAGREC_LAUNCH_CHECK();
  return TAGREC_OK;
}

extern "C" int tagrec_nbr_gather_i32(const int32_t* idx, const int32_t* widx, const int64_t* rows, const int32_t* pos, int64_t n_rows,
                                     int k, int32_t* out_idx, int32_t* out_widx, void* stream) {
  TAGREC_REQUIRE(n_rows >= 0 && k >= 1, "nbr_gather: bad shape");
  if (n_rows == 0) return TAGREC_OK;
  TAGREC_REQUIRE(idx && widx && rows && out_idx && out_widx, "nbr_gather: null pointer");
  int64_t blocks = (n_rows * k + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;
  nbr_gather_kernel<<<static_cast<unsigned>(blocks), 256, 0, static_cast<hipStream_t>(stream)>>>(idx, widx, rows, pos, n_rows, k, out_idx,
                                                                                                  out_widx);
  TAGREC_LAUNCH_CHECK();
  return TAGREC_OK;
}

extern "C" int64_t tagrec_inv_filter_workspace(int64_t n_entries) { return 2 * ((n_entries + kFiltBlock - 1) / kFiltBlock + 1) + 4; }

extern "C" int tagrec_inv_filter_i32(const int32_t* perm_sorted, const int32_t* dest_sorted, int64_t n_entries, int k,
                                     const int32_t* pos_src, const int32_t* pos_dst, const float* attn, int32_t n_dst,
                                     int64_t capacity, int32_t* skey, int32_t* pair, int32_t* src, float* val,
                                     int32_t* workspace, int64_t workspace_ints, void* stream) {
  TAGREC_REQUIRE(n_entries >= 0 && n_entries < (1ll << 31) && k >= 1 && k <= 64 && capacity >= 0, "inv_filter: bad size (k <= 64, < 2^31 entries)");
  TAGREC_REQUIRE(pos_src && workspace && (capacity == 0 || (skey && pair && src && val)) && (n_entries == 0 || (perm_sorted && dest_sorted && attn)),
                 "inv_filter: null pointer");
  TAGREC_REQUIRE(workspace_ints >= tagrec_inv_filter_workspace(n_entries), "inv_filter: workspace too small");
  TAGREC_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 7u) == 0, "inv_filter: workspace must be 8-byte aligned");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int n_blocks = static_cast<int>((n_entries + kFiltBlock - 1) / kFiltBlock);
  // ticket, total and the blocks' status words start at zero; the keys start as "no entry" (>= n_dst: what the pads behind
  // a sorted list were), the listed ones are written over the head
  TAGREC_HIP(hipMemsetAsync(workspace, 0, sizeof(int32_t) * static_cast<size_t>(tagrec_inv_filter_workspace(n_entries)), s));
  if (capacity > 0) TAGREC_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(skey), n_dst, static_cast<size_t>(capacity), s));
  if (n_blocks > 0) {
    const unsigned long long magic = (1ull << 40) / static_cast<unsigned long long>(k) + 1;
    inv_filter_kernel<<<n_blocks, kFiltThreads, 0, s>>>(perm_sorted, dest_sorted, pos_src, pos_dst, attn, k, magic, n_entries, workspace,
                                                        n_blocks, skey, pair, src, val);
    TAGREC_LAUNCH_CHECK();
  }
  return TAGREC_OK;
}
