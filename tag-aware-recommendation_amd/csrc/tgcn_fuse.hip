// K8: TGCN type-level attention + bit-/vector-level convolutions + fusion layer, fused, on the gfx950
// matrix cores (exact-fp32 MFMA 16x16x4).
//
// Replaces `BasicLayer._atten2`, `_conv`, `_fusion` (/root/reference/model/tgcn.py:78-106) for one node type:
//     s_j  = relu(t_j U + q) . p            j = 0..2 (user-, item-, tag-side vector of the node)
//     e_j  = softmax_j(s)_j * t_j           (scaled, NOT summed)
//     y    = [ relu(sum_j wb[c][j] e_j[d])  for c < 32, d < D          bit-level  Conv2d(1,32,(3,1))
//            | relu(w1[c] . e_h)            c < 8, h < 3               vector-level Conv2d(1,8,(1,D))
//            | relu(w2[c] . [e_h ; e_h+1])  c < 8, h < 2                             Conv2d(1,8,(2,D))
//            | relu(w3[c] . [e_0;e_1;e_2])  c < 8 ]                                  Conv2d(1,8,(3,D))
//     out  = relu(y Wf + bf)                Wf [32 D + 48, Dout]
// The reference materialises y for ALL nodes (N x (32 D + 48) floats); here y exists only as the MFMA
// B-operand of the step that consumes it.  MFMA-bound: 2 (32 D + 48) Dout flop per node.
//
// Register layout (everything is computed transposed, as in ngcf.hip): lane (r = lane & 15, q = lane >> 4)
// owns, for node r of the wave's 16-node subtile, the contiguous quarter [q W/4, (q+1) W/4) of every
// width-W vector (inputs, attention pre-activations, outputs).  With the MFMA "row" index i = 4 q' + v of
// block b mapped to element q' W/4 + v W/16 + b, an accumulator is directly the next product's B-operand,
// loads and stores are contiguous W-byte runs per lane, and the A-operand (weights, streamed from L2) of a
// lane is W/16 consecutive floats per k-step.
#include "common.h"

namespace tagrec {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kFuseThreads = 256;
constexpr int kBitC = 32;      // num_bit_conv  (utility/config.py:44)
constexpr int kVecC = 8;       // num_vec_conv  (utility/config.py:45)

__device__ __forceinline__ f32x4 zero4() { return f32x4{0.f, 0.f, 0.f, 0.f}; }
// sum over the 16 lanes of a DPP row (= the 16 nodes of a lane row q) at register speed: __shfl_xor compiles to
// ds_bpermute_b32 -- an LDS round trip and two address instructions per step
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
  return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float row16_sum(float v) {
  v = dpp_add<0xB1>(v);       // quad_perm [1,0,3,2]
  v = dpp_add<0x4E>(v);       // quad_perm [2,3,0,1]
  v = dpp_add<0x141>(v);      // row_half_mirror
  return dpp_add<0x140>(v);   // row_mirror
}
__device__ __forceinline__ float quad_sum(float v) {   // over the 4 lanes that share a node (q = 0..3)
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 32);
  return v;
}

// Message dropout of the layer's output (F.dropout on eu / ei / et, /root/reference/model/tgcn.py:217-219) in the forward
// kernel's epilogue.  One launch covers up to three row segments (the node types of a layer, merged: [0, lo1) users,
// [lo1, lo2) items, [lo2, n) tags), each with its own seed and -- when the launch computes a row SUBSET -- its list of
// node ids: the mask of element (row, column) is the counter-based one of common.h keyed by (seed, NODE id, column), so
// the restricted step on compact tables and an all-rows pass drop the same elements.  The backward pass needs nothing
// new: with out' = mask out / (1 - p) stored, [out' > 0] = mask [out > 0], so g = dOut' / (1 - p) [out' > 0].
struct FuseDrop {
  float p;                         // 0 = off
  int64_t lo1, lo2;
  const int64_t* rows[3];          // nullptr: the segment's rows are nodes 0 .. in order
  uint64_t seed[3];
};
template <int OS, int DOUT>
__device__ __forceinline__ void fuse_drop(const FuseDrop& d, int64_t node, int q, float (&o)[OS]) {
  if (d.p <= 0.f) return;
  const int seg = node >= d.lo2 ? 2 : (node >= d.lo1 ? 1 : 0);
  const int64_t local = node - (seg == 2 ? d.lo2 : (seg == 1 ? d.lo1 : 0));
  const int64_t* rows = seg == 2 ? d.rows[2] : (seg == 1 ? d.rows[1] : d.rows[0]);
  const uint64_t seed = seg == 2 ? d.seed[2] : (seg == 1 ? d.seed[1] : d.seed[0]);
  const int64_t id = rows ? rows[local] : local;
  const DropMask m{d.p, seed};
#pragma unroll
  for (int i = 0; i < OS; i += 4) drop4(m, id * (DOUT / 4) + (q * OS + i) / 4, o[i], o[i + 1], o[i + 2], o[i + 3]);
}

// Launder a (wave-uniform) pointer inside the tile loop: without it the compiler proves the weight loads
// loop-invariant, hoists hundreds of them out of the loop and holds them in registers (spilling the rest).
__device__ __forceinline__ const float* fresh(const float* p) {
  asm volatile("" : "+s"(p));
  return p;
}

// N consecutive floats from p (N in {1,2,4,8}); p is N*4-byte aligned
template <int N>
__device__ __forceinline__ void load_run(const float* __restrict__ p, float (&a)[N]) {
  if constexpr (N == 1) {
    a[0] = p[0];
  } else if constexpr (N == 2) {
    const float2 t = *reinterpret_cast<const float2*>(p);
    a[0] = t.x; a[1] = t.y;
  } else {
#pragma unroll
    for (int i = 0; i < N; i += 4) {
      const float4 t = *reinterpret_cast<const float4*>(p + i);
      a[i] = t.x; a[i + 1] = t.y; a[i + 2] = t.z; a[i + 3] = t.w;
    }
  }
}

// The lane's quarter of a width-W row: W/4 consecutive floats
template <int SEG>
__device__ __forceinline__ void load_seg(const float* __restrict__ p, bool ok, float (&a)[SEG]) {
  if (ok) {
#pragma unroll
    for (int i = 0; i < SEG; i += 4) {
      const float4 t = *reinterpret_cast<const float4*>(p + i);
      a[i] = t.x; a[i + 1] = t.y; a[i + 2] = t.z; a[i + 3] = t.w;
    }
  } else {
#pragma unroll
    for (int i = 0; i < SEG; ++i) a[i] = 0.f;
  }
}

// Type-level attention for one subtile: t[j][e] (e < D/4) -> softmax weights bw[j]; scales t in place.
// Also returns the pre-activations sc[j][ab] (element a = q A/4 + v A/16 + ab) for the backward pass.
template <int D, int A>
__device__ __forceinline__ void type_attention(float (&t)[3][D / 4], const float* __restrict__ U,
                                               const float* __restrict__ qv, const float* __restrict__ pv, int r, int q,
                                               f32x4 (&sc)[3][A / 16], float (&bw)[3]) {
  constexpr int DS = D / 4, AB = A / 16, AS = A / 4;
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int ab = 0; ab < AB; ++ab) sc[j][ab] = zero4();
  const float* urow = U + static_cast<int64_t>(q * DS) * A + r * AB;
#pragma unroll
  for (int e = 0; e < DS; ++e) {
    float ua[AB];
    load_run<AB>(urow + e * A, ua);
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int ab = 0; ab < AB; ++ab) sc[j][ab] = __builtin_amdgcn_mfma_f32_16x16x4f32(ua[ab], t[j][e], sc[j][ab], 0, 0, 0);
    if ((e & 7) == 7) __builtin_amdgcn_sched_barrier(0);
  }
  float s[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    float part = 0.f;
#pragma unroll
    for (int ab = 0; ab < AB; ++ab)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int a = q * AS + v * AB + ab;
        sc[j][ab][v] += qv[a];
        part = fmaf(fmaxf(sc[j][ab][v], 0.f), pv[a], part);
      }
    s[j] = quad_sum(part);
  }
  const float mx = fmaxf(s[0], fmaxf(s[1], s[2]));
  const float e0 = expf(s[0] - mx), e1 = expf(s[1] - mx), e2 = expf(s[2] - mx);
  const float inv = 1.0f / (e0 + e1 + e2);
  bw[0] = e0 * inv; bw[1] = e1 * inv; bw[2] = e2 * inv;
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int e = 0; e < DS; ++e) t[j][e] *= bw[j];
}

// Vector-level convolutions of one subtile through the matrix cores: the 8 filters sit on MFMA rows 0..7, so
// lanes q = 0, 1 end up with filter c = 4 q + v in register v; six groups g: v1 h=0..2, v2 h=0..1, v3.
// pre[g] holds the PRE-activation.
template <int D>
__device__ __forceinline__ void vector_conv(const float (&e3)[3][D / 4], const float* __restrict__ w1,
                                            const float* __restrict__ w2, const float* __restrict__ w3, int r, int q,
                                            f32x4 (&pre)[6]) {
  constexpr int DS = D / 4;
#pragma unroll
  for (int g = 0; g < 6; ++g) pre[g] = zero4();
  const bool row_ok = r < kVecC;
  const int d0 = q * DS;
#pragma unroll
  for (int e = 0; e < DS; ++e) {
    const float a1 = row_ok ? w1[r * D + d0 + e] : 0.f;
    const float a20 = row_ok ? w2[(r * 2 + 0) * D + d0 + e] : 0.f;
    const float a21 = row_ok ? w2[(r * 2 + 1) * D + d0 + e] : 0.f;
    const float a30 = row_ok ? w3[(r * 3 + 0) * D + d0 + e] : 0.f;
    const float a31 = row_ok ? w3[(r * 3 + 1) * D + d0 + e] : 0.f;
    const float a32 = row_ok ? w3[(r * 3 + 2) * D + d0 + e] : 0.f;
#pragma unroll
    for (int h = 0; h < 3; ++h) pre[h] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, e3[h][e], pre[h], 0, 0, 0);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      pre[3 + h] = __builtin_amdgcn_mfma_f32_16x16x4f32(a20, e3[h][e], pre[3 + h], 0, 0, 0);
      pre[3 + h] = __builtin_amdgcn_mfma_f32_16x16x4f32(a21, e3[h + 1][e], pre[3 + h], 0, 0, 0);
    }
    pre[5] = __builtin_amdgcn_mfma_f32_16x16x4f32(a30, e3[0][e], pre[5], 0, 0, 0);
    pre[5] = __builtin_amdgcn_mfma_f32_16x16x4f32(a31, e3[1][e], pre[5], 0, 0, 0);
    pre[5] = __builtin_amdgcn_mfma_f32_16x16x4f32(a32, e3[2][e], pre[5], 0, 0, 0);
    if ((e & 3) == 3) __builtin_amdgcn_sched_barrier(0);   // keep the unrolled loop from hoisting every weight load
  }
}

// index of vector feature (group g, filter c) inside y, after the 32 D bit-level features
__device__ __forceinline__ int vec_feature(int g, int c) {
  if (g < 3) return c * 3 + g;                        // conv_1: [c][h]
  if (g < 5) return 3 * kVecC + c * 2 + (g - 3);      // conv_2: [c][h]
  return 5 * kVecC + c;                               // conv_3: [c]
}

// ---- forward -----------------------------------------------------------------------------------
template <int D, int DOUT, int A, int NS, bool DROP = false>
__global__ __launch_bounds__(kFuseThreads) void tgcn_fuse_fwd_kernel(
    const float* __restrict__ T0, const float* __restrict__ T1, const float* __restrict__ T2, int64_t n,
    const float* __restrict__ U, const float* __restrict__ qv, const float* __restrict__ pv,
    const float* __restrict__ wb, const float* __restrict__ w1, const float* __restrict__ w2,
    const float* __restrict__ w3, const float* __restrict__ Wf, const float* __restrict__ bf,
    float* __restrict__ bw_out, float* __restrict__ out, FuseDrop drop) {
  constexpr int DS = D / 4, OS = DOUT / 4, OB = DOUT / 16, AB = A / 16;
  const int lane = threadIdx.x & 63;
  const int r = lane & 15, q = lane >> 4;
  const int64_t n_tiles = (n + 16 * NS - 1) / (16 * NS);
  const float* Tj[3] = {T0, T1, T2};
  for (int64_t tile = static_cast<int64_t>(blockIdx.x) * (kFuseThreads / 64) + (threadIdx.x >> 6); tile < n_tiles;
       tile += static_cast<int64_t>(gridDim.x) * (kFuseThreads / 64)) {
    U = fresh(U); qv = fresh(qv); pv = fresh(pv); wb = fresh(wb); w1 = fresh(w1); w2 = fresh(w2); w3 = fresh(w3);
    Wf = fresh(Wf); bf = fresh(bf);
    float t[NS][3][DS];
    int64_t node[NS];
    bool ok[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      node[s] = (tile * NS + s) * 16 + r;
      ok[s] = node[s] < n;
#pragma unroll
      for (int j = 0; j < 3; ++j) load_seg<DS>(Tj[j] + node[s] * D + q * DS, ok[s], t[s][j]);
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      f32x4 sc[3][AB];
      float bw[3];
      type_attention<D, A>(t[s], U, qv, pv, r, q, sc, bw);
      if (ok[s] && q == 0) {
        bw_out[node[s] * 3 + 0] = bw[0];
        bw_out[node[s] * 3 + 1] = bw[1];
        bw_out[node[s] * 3 + 2] = bw[2];
      }
    }
    f32x4 acc[NS][OB];
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
      for (int ob = 0; ob < OB; ++ob) acc[s][ob] = zero4();
    // bit-level features: k = c D + q DS + e
    for (int c = 0; c < kBitC; ++c) {
      const float c0 = wb[c * 3], c1 = wb[c * 3 + 1], c2 = wb[c * 3 + 2];
      const float* wrow = Wf + (static_cast<int64_t>(c) * D + q * DS) * DOUT + r * OB;
#pragma unroll
      for (int e = 0; e < DS; ++e) {
        float a[OB];
        load_run<OB>(wrow + e * DOUT, a);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          const float y = fmaxf(fmaf(c0, t[s][0][e], fmaf(c1, t[s][1][e], c2 * t[s][2][e])), 0.f);
#pragma unroll
          for (int ob = 0; ob < OB; ++ob) acc[s][ob] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ob], y, acc[s][ob], 0, 0, 0);
        }
      }
    }
    // vector-level features: group g, register v -> filter c = 4 q + v on lanes q < 2 (zero elsewhere)
    {
      f32x4 pre[NS][6];
#pragma unroll
      for (int s = 0; s < NS; ++s) vector_conv<D>(t[s], w1, w2, w3, r, q, pre[s]);
#pragma unroll
      for (int g = 0; g < 6; ++g)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int f = vec_feature(g, 4 * (q & 1) + v);
          float a[OB];
          load_run<OB>(Wf + (static_cast<int64_t>(kBitC) * D + f) * DOUT + r * OB, a);
#pragma unroll
          for (int s = 0; s < NS; ++s) {
            const float y = q < 2 ? fmaxf(pre[s][g][v], 0.f) : 0.f;
#pragma unroll
            for (int ob = 0; ob < OB; ++ob) acc[s][ob] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ob], y, acc[s][ob], 0, 0, 0);
          }
        }
    }
    // epilogue: element o = q OS + v OB + ob
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      if (!ok[s]) continue;
      float o[OS];
#pragma unroll
      for (int ob = 0; ob < OB; ++ob)
#pragma unroll
        for (int v = 0; v < 4; ++v) o[v * OB + ob] = fmaxf(acc[s][ob][v] + bf[q * OS + v * OB + ob], 0.f);
      if constexpr (DROP) fuse_drop<OS, DOUT>(drop, node[s], q, o);
      float* dst = out + node[s] * DOUT + q * OS;
#pragma unroll
      for (int i = 0; i < OS; i += 4) *reinterpret_cast<float4*>(dst + i) = make_float4(o[i], o[i + 1], o[i + 2], o[i + 3]);
    }
  }
}

// ---- forward, fusion weights staged through LDS --------------------------------------------------------------
// In the kernel above every wave streams the whole of Wf (32 D + 48 rows of Dout floats) from L2 for its 16 nodes;
// with the weight loads removed it runs twice as fast (measured at D = Dout = 128), i.e. it is bound by that stream.
// Here the block's four waves -- four 16-node subtiles walking the same weight rows in the same order -- share
// each 32-row chunk of Wf through LDS: the L2 stream shrinks 4x.  Chunks are filled by LDS-DMA
// (global_load_lds_dwordx4: no staging registers, the kernel is at the VGPR limit for two waves per SIMD), two
// buffers, chunk i+1 in flight while chunk i feeds the MFMAs, one barrier per chunk.
//   chunk (c, ec): rows k = c D + q D/4 + ec*kEC + el  (q < 4, el < kEC)  ->  LDS row q*kEC + el
//   a DMA wave-instruction writes 1 KiB = kRP consecutive LDS rows (one q); piece p sits at p*1024 + q*32 + (q&1)*SHIFT
//   bytes: with Dout = 128 the 16-byte shift of odd q makes the two quarters a ds_read_b128 lane group spans
//   (q = 0,1 or q = 2,3) fall on disjoint banks; with Dout = 64 they already do.
constexpr int kEC = 8;

// DROP: message dropout in the epilogue -- a separate instantiation, so that the p = 0 kernel keeps its register allocation
// (with the mask code compiled in it grew from 232 to 252 VGPRs and ran 4-8 % slower)
template <int D, int DOUT, int A, bool DROP = false>
__global__ __launch_bounds__(kFuseThreads, 2) void tgcn_fuse_fwd_lds_kernel(
    const float* __restrict__ T0, const float* __restrict__ T1, const float* __restrict__ T2, int64_t n,
    const float* __restrict__ U, const float* __restrict__ qv, const float* __restrict__ pv,
    const float* __restrict__ wb, const float* __restrict__ w1, const float* __restrict__ w2,
    const float* __restrict__ w3, const float* __restrict__ Wf, const float* __restrict__ bf,
    float* __restrict__ bw_out, float* __restrict__ out, FuseDrop drop) {
  constexpr int DS = D / 4, OS = DOUT / 4, OB = DOUT / 16, AB = A / 16;
  constexpr int NEC = DS / kEC;                         // chunks per bit-level filter
  constexpr int RP = 256 / DOUT;                        // LDS rows per 1 KiB piece
  constexpr int PIECES = 4 * kEC / RP, PPW = PIECES / 4;   // per chunk, per wave
  constexpr int SHIFT = DOUT == 128 ? 16 : 0;
  constexpr int BUF_BYTES = PIECES * 1024 + 3 * 32 + 32;
  static_assert(NEC >= 2 && NEC % 2 == 0 && kEC % RP == 0 && PPW >= 1, "tgcn_fuse_fwd_lds: unsupported shape");
  __shared__ __attribute__((aligned(1024))) char wbuf[2][BUF_BYTES];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int64_t n_groups = (n + 63) / 64;
  const float* Tj[3] = {T0, T1, T2};
  // DMA source of this lane inside a chunk: piece p = wave*PPW + j covers LDS rows p*RP ..; the lane's row / column
  const int lanes_per_row = DOUT / 4;
  const int row_in_piece = lane / lanes_per_row, c4 = lane % lanes_per_row;
  // this lane's read base inside a buffer (see the layout above)
  const int read_base = q * (kEC / RP) * 1024 + q * 32 + (q & 1) * SHIFT + r * OB * 4;
  // lane part of the DMA source offsets, once (a VALU instruction costs 1/8 of an MFMA and does not overlap with one)
  int dma_off[PPW];
#pragma unroll
  for (int j = 0; j < PPW; ++j) {
    const int lrow = (wave * PPW + j) * RP + row_in_piece;            // LDS row -> (q', el)
    dma_off[j] = ((lrow / kEC) * DS + lrow % kEC) * DOUT + c4 * 4;
  }
  const int wave_s = __builtin_amdgcn_readfirstlane(wave);
  for (int64_t grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
    U = fresh(U); qv = fresh(qv); pv = fresh(pv); wb = fresh(wb); w1 = fresh(w1); w2 = fresh(w2); w3 = fresh(w3);
    Wf = fresh(Wf); bf = fresh(bf);
    auto issue = [&](int c, int ec, int buf) {
      const float* base = Wf + (static_cast<int64_t>(c) * D + ec * kEC) * DOUT;
#pragma unroll
      for (int j = 0; j < PPW; ++j) {
        const int p = wave_s * PPW + j;
        const int pq = (p * RP) / kEC;
        char* dst = &wbuf[buf][p * 1024 + pq * 32 + (pq & 1) * SHIFT];
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + dma_off[j]),
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
      }
    };
    issue(0, 0, 0);
    float t[3][DS];
    const int64_t node = (grp * 4 + wave) * 16 + r;
    const bool ok = node < n;
#pragma unroll
    for (int j = 0; j < 3; ++j) load_seg<DS>(Tj[j] + node * D + q * DS, ok, t[j]);
    {
      f32x4 sc[3][AB];
      float bw[3];
      type_attention<D, A>(t, U, qv, pv, r, q, sc, bw);
      if (ok && q == 0) {
        bw_out[node * 3 + 0] = bw[0];
        bw_out[node * 3 + 1] = bw[1];
        bw_out[node * 3 + 2] = bw[2];
      }
    }
    f32x4 acc[OB];
#pragma unroll
    for (int ob = 0; ob < OB; ++ob) acc[ob] = zero4();
    for (int c = 0; c < kBitC; ++c) {
      const float c0 = wb[c * 3], c1 = wb[c * 3 + 1], c2 = wb[c * 3 + 2];
#pragma unroll
      for (int ec = 0; ec < NEC; ++ec) {
        __syncthreads();                                 // chunk (c, ec) has landed; the other buffer is free again
        if (ec + 1 < NEC) issue(c, ec + 1, (ec + 1) & 1);
        else if (c + 1 < kBitC) issue(c + 1, 0, 0);
        const char* rb = &wbuf[ec & 1][read_base];
#pragma unroll
        for (int el = 0; el < kEC; ++el) {
          float a[OB];
          const float* ap = reinterpret_cast<const float*>(rb + (el / RP) * 1024 + (el % RP) * DOUT * 4);
#pragma unroll
          for (int i = 0; i < OB; i += 4) {
            const float4 w4 = *reinterpret_cast<const float4*>(ap + i);
            a[i] = w4.x; a[i + 1] = w4.y; a[i + 2] = w4.z; a[i + 3] = w4.w;
          }
          const int e = ec * kEC + el;
          const float y = fmaxf(fmaf(c0, t[0][e], fmaf(c1, t[1][e], c2 * t[2][e])), 0.f);
#pragma unroll
          for (int ob = 0; ob < OB; ++ob) acc[ob] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ob], y, acc[ob], 0, 0, 0);
        }
      }
    }
    {
      f32x4 pre[6];
      vector_conv<D>(t, w1, w2, w3, r, q, pre);
#pragma unroll
      for (int g = 0; g < 6; ++g)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int f = vec_feature(g, 4 * (q & 1) + v);
          float a[OB];
          load_run<OB>(Wf + (static_cast<int64_t>(kBitC) * D + f) * DOUT + r * OB, a);
          const float y = q < 2 ? fmaxf(pre[g][v], 0.f) : 0.f;
#pragma unroll
          for (int ob = 0; ob < OB; ++ob) acc[ob] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ob], y, acc[ob], 0, 0, 0);
        }
    }
    if (ok) {
      float o[OS];
#pragma unroll
      for (int ob = 0; ob < OB; ++ob)
#pragma unroll
        for (int v = 0; v < 4; ++v) o[v * OB + ob] = fmaxf(acc[ob][v] + bf[q * OS + v * OB + ob], 0.f);
      if constexpr (DROP) fuse_drop<OS, DOUT>(drop, node, q, o);
      float* dst = out + node * DOUT + q * OS;
#pragma unroll
      for (int i = 0; i < OS; i += 4) *reinterpret_cast<float4*>(dst + i) = make_float4(o[i], o[i + 1], o[i + 2], o[i + 3]);
    }
    __syncthreads();     // every wave is done with both buffers before the next group's first DMA overwrites buffer 0
  }
}

template <int D, int DOUT>
int launch_fuse_fwd(const float* T0, const float* T1, const float* T2, int64_t n, const float* U, const float* qv,
                    const float* pv, const float* wb, const float* w1, const float* w2, const float* w3, const float* Wf,
                    const float* bf, float* bw_out, float* out, const FuseDrop& drop, hipStream_t s) {
  if constexpr ((D == 64 || D == 128) && (DOUT == 64 || DOUT == 128)) {
    const int64_t groups = (n + 63) / 64;
    const unsigned grid = static_cast<unsigned>(groups < 512 ? groups : 512);
    if (drop.p > 0.f)
      tgcn_fuse_fwd_lds_kernel<D, DOUT, 32, true><<<grid, kFuseThreads, 0, s>>>(T0, T1, T2, n, U, qv, pv, wb, w1, w2, w3, Wf, bf, bw_out,
                                                                              out, drop);
    else
      tgcn_fuse_fwd_lds_kernel<D, DOUT, 32, false><<<grid, kFuseThreads, 0, s>>>(T0, T1, T2, n, U, qv, pv, wb, w1, w2, w3, Wf, bf, bw_out,
                                                                               out, drop);
    TAGREC_LAUNCH_CHECK();
    return TAGREC_OK;
  }
  constexpr int NS = (D >= 64) ? 1 : 2;   // measured at D = 128: two waves per SIMD with one subtile beat one wave with two
  const int64_t tiles = (n + 16 * NS - 1) / (16 * NS);
  int64_t blocks = (tiles + 3) / 4;
  if (blocks > 256 * 2) blocks = 256 * 2;
  if (drop.p > 0.f)
    tgcn_fuse_fwd_kernel<D, DOUT, 32, NS, true><<<static_cast<unsigned>(blocks), kFuseThreads, 0, s>>>(
        T0, T1, T2, n, U, qv, pv, wb, w1, w2, w3, Wf, bf, bw_out, out, drop);
  else
    tgcn_fuse_fwd_kernel<D, DOUT, 32, NS, false><<<static_cast<unsigned>(blocks), kFuseThreads, 0, s>>>(
        T0, T1, T2, n, U, qv, pv, wb, w1, w2, w3, Wf, bf, bw_out, out, drop);
  TAGREC_LAUNCH_CHECK();
  return TAGREC_OK;
}


// ---- backward, data part -------------------------------------------------------------------------
// From dOut and the saved `out` (ReLU mask): gradients w.r.t. the three input vectors, plus what the weight
// gradients need (kept small): yvec [n, 48] post-ReLU vector features, dfeat [n, 48] their pre-activation
// gradients, dS [n, 3 A] type-attention pre-activation gradients.  The tiny parameter gradients that are plain
// sums over nodes (dwb [32,3], dq [A], dp [A]) are accumulated per block in LDS and written as
// per-block partials [block][96 + 2 A]; dbf is a column sum the caller takes.
//
// Where the time goes at D = Dout = 128 (1 M nodes, 18.2 ms; ablations on an MI355X, each part removed in turn):
// MFMAs of the main loop 7.5 ms, its VALU part 2.7 ms, its LDS reads 2.0 ms, barriers + DMA waits 2.0 ms, per-tile
// prologue / epilogue 4.4 ms -- the parts ADD UP: with 512 registers per wave only one wave fits a SIMD and nothing
// overlaps.  Pipelining inside the wave (the piece rotation and the three-buffer ring below) recovers ~3 %; the lever
// is a second wave per SIMD, i.e. the e / de accumulators (192 registers) split over two waves or two passes.
template <int D, int DOUT, int A, bool LDSW>
__global__ __launch_bounds__(kFuseThreads) void tgcn_fuse_bwd_kernel(
    const float* __restrict__ T0, const float* __restrict__ T1, const float* __restrict__ T2, int64_t n,
    const float* __restrict__ U, const float* __restrict__ qv, const float* __restrict__ pv,
    const float* __restrict__ wb, const float* __restrict__ w1, const float* __restrict__ w2,
    const float* __restrict__ w3, const float* __restrict__ Wf, const float* __restrict__ outv,
    const float* __restrict__ dOut, float* __restrict__ dT0, float* __restrict__ dT1, float* __restrict__ dT2,
    float* __restrict__ yvec, float* __restrict__ dfeat, float* __restrict__ dS, float* __restrict__ part) {
  constexpr int DS = D / 4, OS = DOUT / 4, IB = D / 16, AB = A / 16, AS = A / 4;
  constexpr int NSM = 3 * kBitC + 2 * A;             // dwb | dq | dp
  __shared__ float sh[NSM];
  // LDSW: the 16 rows of Wf that one (c, b) step multiplies by are the same for the block's four waves; they are
  // staged once per block through LDS (LDS-DMA, two buffers, as in tgcn_fuse_fwd_lds_kernel) instead of being
  // streamed from L2 by every wave -- without the stream this kernel runs 2.3x faster (measured, D = Dout = 128).
  // A row is Dout floats; its 16-byte units are XOR-swizzled with g(row) so that the b128 reads of a lane group
  // (rows {0-3,12-15} of one quarter q and {4-11} of the next) fall on 16 distinct bank slots.
  constexpr int NB = 2;                      // b-steps (16 weight rows each) per staged chunk: a barrier every 2 x Dout/4 MFMAs
  constexpr int RP = 256 / DOUT, PPW = (NB * 16 / RP) / 4, CHUNK = NB * 16 * DOUT * 4;
  // The chunks form ONE periodic stream over the whole kernel (the weight rows do not depend on the tile): a ring of
  // three buffers, chunk i + 2 issued when chunk i is entered, so a chunk has two chunk-times (~1.7 us of MFMA work)
  // to arrive from L2; waits are counted (`s_waitcnt vmcnt(PPW)` + a raw s_barrier), never the vmcnt(0) of
  // __syncthreads(), which would drain the chunk just issued.
  constexpr int NBUF = 3, CPC = IB / NB, NCH = kBitC * CPC;
  static_assert(!LDSW || CPC >= 2, "ring prologue issues two chunks of filter 0");
  __shared__ __attribute__((aligned(1024))) char wbuf[LDSW ? NBUF : 1][LDSW ? CHUNK : 16];
  for (int i = threadIdx.x; i < NSM; i += kFuseThreads) sh[i] = 0.f;
  __syncthreads();
  float* sh_wb = sh;
  float* sh_q = sh + 3 * kBitC;
  float* sh_p = sh_q + A;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int64_t n_tiles = (n + 15) / 16;
  const float* Tj[3] = {T0, T1, T2};
  float* dTj[3] = {dT0, dT1, dT2};
  auto swz = [](int row) {        // see above: bit (row3 ^ row2) goes where the quarter stride (in 16-byte units) has its bit
    const int x = ((row >> 3) ^ (row >> 2)) & 1;
    return DOUT == 128 ? ((row & 0xB) | (x << 2)) : ((row & 0x7) | (x << 3));
  };
  const int lanes_per_row = DOUT / 4;
  const int dma_row = lane / lanes_per_row, dma_unit = lane % lanes_per_row;
  // row of the weight matrices this lane feeds as MFMA A-operand when the OUTPUT rows are input features:
  // row i = r of block b  <->  d = (r >> 2) DS + 4 b + (r & 3)
  const int drow = (r >> 2) * DS + (r & 3);
  float q_acc[AS], p_acc[AS];
#pragma unroll
  for (int i = 0; i < AS; ++i) { q_acc[i] = 0.f; p_acc[i] = 0.f; }
  // LDSW: the four waves of a block step through tile GROUPS together (block-uniform trip count for the barriers)
  const int64_t it_first = LDSW ? static_cast<int64_t>(blockIdx.x) : static_cast<int64_t>(blockIdx.x) * 4 + wave;
  const int64_t it_step = LDSW ? static_cast<int64_t>(gridDim.x) : static_cast<int64_t>(gridDim.x) * 4;
  const int64_t it_end = LDSW ? (n_tiles + 3) / 4 : n_tiles;
  auto issue = [&](int c, int b, int buf) {              // rows of steps b .. b + NB - 1 of filter c
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
      const int p = wave * PPW + j;
      const int lrow = p * RP + dma_row;                                      // chunk row = (step, MFMA row i)
      const int sub = lrow >> 4, i = lrow & 15;
      const int u = dma_unit ^ swz(i);
      const float* src = Wf + (static_cast<int64_t>(c) * D + (i >> 2) * DS + (i & 3) + 4 * (b + sub)) * DOUT + u * 4;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)&wbuf[buf][p * 1024], 16, 0, 0);
    }
  };
  int cur = 0;                                           // ring slot of the chunk being consumed (block-uniform)
  if constexpr (LDSW)
    if (it_first < it_end) { issue(0, 0, 0); issue(0, NB, 1); }
  for (int64_t it = it_first; it < it_end; it += it_step) {
    const int64_t tile = LDSW ? it * 4 + wave : it;
    const bool has_next = it + it_step < it_end;
    U = fresh(U); qv = fresh(qv); pv = fresh(pv); wb = fresh(wb); w1 = fresh(w1); w2 = fresh(w2); w3 = fresh(w3);
    Wf = fresh(Wf);
    const int64_t node = tile * 16 + r;
    const bool ok = node < n;
    float e3[3][DS];
#pragma unroll
    for (int j = 0; j < 3; ++j) load_seg<DS>(Tj[j] + node * D + q * DS, ok, e3[j]);
    f32x4 sc[3][AB];
    float bw[3];
    type_attention<D, A>(e3, U, qv, pv, r, q, sc, bw);       // e3 now holds softmax_j * t_j
    // g = dOut * [out > 0], the lane's quarter of the Dout outputs
    float g[OS];
    {
      float o[OS];
      load_seg<OS>(dOut + node * DOUT + q * OS, ok, g);
      load_seg<OS>(outv + node * DOUT + q * OS, ok, o);
#pragma unroll
      for (int i = 0; i < OS; ++i) g[i] = o[i] > 0.f ? g[i] : 0.f;
    }
    float de3[3][DS];
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int e = 0; e < DS; ++e) de3[j][e] = 0.f;
    // bit-level: dy^T[d][node] = sum_o Wf[c D + d][o] g[node][o], then through the ReLU and the 3 -> 1 mix.
    // Software pipeline: the weight rows of block (c, b+1) are in flight while block (c, b) runs its MFMAs (one
    // wave per SIMD here, nothing else hides the L2 latency); two accumulators break the dependent MFMA chain.
    float a_next[OS];
    if constexpr (!LDSW) load_run<OS>(Wf + static_cast<int64_t>(drow) * DOUT + q * OS, a_next);
    for (int c = 0; c < kBitC; ++c) {
      const float c0 = wb[c * 3], c1 = wb[c * 3 + 1], c2 = wb[c * 3 + 2];
      float a0 = 0.f, a1 = 0.f, a2 = 0.f;
      const float* wrow = Wf + (static_cast<int64_t>(c) * D + drow) * DOUT + q * OS;
      const int cn = c + 1 < kBitC ? c + 1 : c;
      const float* wrow_next = Wf + (static_cast<int64_t>(cn) * D + drow) * DOUT + q * OS;
#pragma unroll
      for (int b = 0; b < IB; ++b) {
        float a[OS];
        if constexpr (LDSW) {
          if (b % NB == 0) {
            const int j = c * CPC + b / NB;              // chunk number inside the tile
            // chunk j + 1 may still be in flight (it exists unless this is the very last chunk of the block's work)
            if (j + 1 < NCH || has_next) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(PPW) : "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                // chunk j is in LDS for every wave; everyone is done with chunk j - 1
            int j2 = j + 2;
            const bool wrap = j2 >= NCH;                 // the stream continues with the next tile group's first chunks
            if (wrap) j2 -= NCH;
            if (!wrap || has_next) issue(j2 / CPC, (j2 % CPC) * NB, cur >= 1 ? cur - 1 : NBUF - 1);
          }
        }
        f32x4 dy = zero4(), dy1 = zero4();
        if constexpr (LDSW) {
          // The allocator has no room for the 32 weight values of a step (the kernel sits at the register limit), and
          // left to itself reads one 16-byte piece, waits, issues its 4 MFMAs, reads the next: the LDS latency is
          // exposed 8 times per step.  Written out as a rotation over three pieces -- two reads always in flight
          // behind the MFMAs of the current piece -- it costs 8 more registers and hides it.
          const char* rowp = &wbuf[cur][((b % NB) * 16 + r) * DOUT * 4];
          const int sw = swz(r);
          auto piece = [&](int t4) { return *reinterpret_cast<const float4*>(rowp + (((q * (OS / 4) + t4) ^ sw) << 4)); };
          float4 w0 = piece(0), w1_ = piece(1 < OS / 4 ? 1 : 0);
#pragma unroll
          for (int t4 = 0; t4 < OS / 4; ++t4) {
            const float4 w2_ = piece(t4 + 2 < OS / 4 ? t4 + 2 : t4);
            const int t = 4 * t4;
            dy = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.x, g[t], dy, 0, 0, 0);
            dy1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.y, g[t + 1], dy1, 0, 0, 0);
            dy = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.z, g[t + 2], dy, 0, 0, 0);
            dy1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.w, g[t + 3], dy1, 0, 0, 0);
            w0 = w1_; w1_ = w2_;
          }
          // pin that order (the scheduler otherwise sinks every read down to its use to save registers)
          __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
          for (int t4 = 0; t4 < OS / 4; ++t4) {
            if (t4 + 2 < OS / 4) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
          }
        } else {
#pragma unroll
          for (int t = 0; t < OS; ++t) a[t] = a_next[t];
          load_run<OS>(b + 1 < IB ? wrow + static_cast<int64_t>(4 * (b + 1)) * DOUT : wrow_next, a_next);
#pragma unroll
          for (int t = 0; t < OS; t += 2) {
            dy = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], g[t], dy, 0, 0, 0);
            dy1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t + 1], g[t + 1], dy1, 0, 0, 0);
          }
        }
        dy += dy1;
        if constexpr (LDSW)
          if (b % NB == NB - 1) cur = cur == NBUF - 1 ? 0 : cur + 1;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int e = 4 * b + v;
          const float pre = fmaf(c0, e3[0][e], fmaf(c1, e3[1][e], c2 * e3[2][e]));
          const float dp_ = pre > 0.f ? dy[v] : 0.f;
          de3[0][e] = fmaf(c0, dp_, de3[0][e]);
          de3[1][e] = fmaf(c1, dp_, de3[1][e]);
          de3[2][e] = fmaf(c2, dp_, de3[2][e]);
          a0 = fmaf(dp_, e3[0][e], a0);
          a1 = fmaf(dp_, e3[1][e], a1);
          a2 = fmaf(dp_, e3[2][e], a2);
        }
        if constexpr (!LDSW) __builtin_amdgcn_sched_barrier(0);      // one block's loads / MFMAs / VALU at a time
      }
      // dwb[c][j] += sum over the wave (rows past n contribute 0: their g is 0)
#pragma unroll
      for (int m = 1; m < 64; m <<= 1) { a0 += __shfl_xor(a0, m); a1 += __shfl_xor(a1, m); a2 += __shfl_xor(a2, m); }
      if (lane == 0) { atomicAdd(&sh_wb[c * 3], a0); atomicAdd(&sh_wb[c * 3 + 1], a1); atomicAdd(&sh_wb[c * 3 + 2], a2); }
    }
    // vector-level: recompute pre-activations, gradient of the 48 features, then back to de3
    {
      f32x4 pre[6];
      vector_conv<D>(e3, w1, w2, w3, r, q, pre);
      f32x4 dpv[6];
#pragma unroll
      for (int gi = 0; gi < 6; ++gi) {
        // rows of this product are filters: lane r < 8 feeds Wf row of feature (gi, r)
        const int f = vec_feature(gi, r & 7);
        float a[OS];
        load_run<OS>(Wf + (static_cast<int64_t>(kBitC) * D + f) * DOUT + q * OS, a);
        f32x4 dy = zero4();
#pragma unroll
        for (int t = 0; t < OS; ++t) dy = __builtin_amdgcn_mfma_f32_16x16x4f32(r < kVecC ? a[t] : 0.f, g[t], dy, 0, 0, 0);
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const bool live = q < 2;
          const float y = live ? fmaxf(pre[gi][v], 0.f) : 0.f;
          dpv[gi][v] = (live && pre[gi][v] > 0.f) ? dy[v] : 0.f;
          if (ok && live) {
            const int ff = vec_feature(gi, 4 * q + v);
            yvec[node * (6 * kVecC) + ff] = y;
            dfeat[node * (6 * kVecC) + ff] = dpv[gi][v];
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      // de3_h[d] += sum_c w[c][a][d] dpre[c]: contraction over the 8 filters = 4 k-steps' worth in 2 live slots
#pragma unroll
      for (int b = 0; b < IB; ++b) {
        const int d = drow + 4 * b;
        f32x4 acc0 = zero4(), acc1 = zero4(), acc2 = zero4();
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int c = 4 * (q & 1) + v;                 // filter fed by this k-slot (slots 2,3 carry zeros)
          const float z = q < 2 ? 1.f : 0.f;
          const float k1 = z * w1[c * D + d];
          const float k20 = z * w2[(c * 2 + 0) * D + d], k21 = z * w2[(c * 2 + 1) * D + d];
          const float k30 = z * w3[(c * 3 + 0) * D + d], k31 = z * w3[(c * 3 + 1) * D + d], k32 = z * w3[(c * 3 + 2) * D + d];
          // conv_1: feature (h) touches e3_h
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(k1, dpv[0][v], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(k1, dpv[1][v], acc1, 0, 0, 0);
          acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(k1, dpv[2][v], acc2, 0, 0, 0);
          // conv_2: window h covers rows h (tap 0) and h+1 (tap 1)
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(k20, dpv[3][v], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(k21, dpv[3][v], acc1, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(k20, dpv[4][v], acc1, 0, 0, 0);
          acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(k21, dpv[4][v], acc2, 0, 0, 0);
          // conv_3: one window over rows 0,1,2
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(k30, dpv[5][v], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(k31, dpv[5][v], acc1, 0, 0, 0);
          acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(k32, dpv[5][v], acc2, 0, 0, 0);
        }
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          de3[0][4 * b + v] += acc0[v];
          de3[1][4 * b + v] += acc1[v];
          de3[2][4 * b + v] += acc2[v];
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // through e_j = bw_j t_j and the type-level softmax
    float tj[3][DS];
    float db[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      load_seg<DS>(Tj[j] + node * D + q * DS, ok, tj[j]);
      float s_ = 0.f;
#pragma unroll
      for (int e = 0; e < DS; ++e) s_ = fmaf(de3[j][e], tj[j][e], s_);
      db[j] = quad_sum(s_);
    }
    const float mix = bw[0] * db[0] + bw[1] * db[1] + bw[2] * db[2];
    float dsv[3][AS];                                        // dS_j[a] in the lane's quarter, order t = v AB + ab
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const float ds = bw[j] * (db[j] - mix);
#pragma unroll
      for (int ab = 0; ab < AB; ++ab)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int t = v * AB + ab;
          const float h = sc[j][ab][v];
          const float x = h > 0.f ? ds * pv[q * AS + t] : 0.f;
          dsv[j][t] = x;
          q_acc[t] += x;
          p_acc[t] = fmaf(ds, fmaxf(h, 0.f), p_acc[t]);
        }
      if (ok) {
        float* dst = dS + node * (3 * A) + j * A + q * AS;
#pragma unroll
        for (int t = 0; t < AS; t += 4) *reinterpret_cast<float4*>(dst + t) = make_float4(dsv[j][t], dsv[j][t + 1], dsv[j][t + 2], dsv[j][t + 3]);
      }
    }
    // dt_j = bw_j de3_j + U dS_j
#pragma unroll
    for (int b = 0; b < IB; ++b) {
      float ua[AS];
      load_run<AS>(U + static_cast<int64_t>(drow + 4 * b) * A + q * AS, ua);
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        f32x4 acc = zero4();
#pragma unroll
        for (int t = 0; t < AS; ++t) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ua[t], dsv[j][t], acc, 0, 0, 0);
#pragma unroll
        for (int v = 0; v < 4; ++v) de3[j][4 * b + v] = fmaf(bw[j], de3[j][4 * b + v], acc[v]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (ok) {
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        float* dst = dTj[j] + node * D + q * DS;
#pragma unroll
        for (int e = 0; e < DS; e += 4) *reinterpret_cast<float4*>(dst + e) = make_float4(de3[j][e], de3[j][e + 1], de3[j][e + 2], de3[j][e + 3]);
      }
    }
  }
  // fold the per-lane sums over the 16 node lanes, then into LDS
#pragma unroll
  for (int i = 0; i < AS; ++i) {
    float a = q_acc[i], b = p_acc[i];
    a = row16_sum(a);
    b = row16_sum(b);
    if (r == 0) { atomicAdd(&sh_q[q * AS + i], a); atomicAdd(&sh_p[q * AS + i], b); }
  }
  __syncthreads();
  float* o = part + static_cast<int64_t>(blockIdx.x) * NSM;
  for (int i = threadIdx.x; i < NSM; i += kFuseThreads) o[i] = sh[i];
}

// ---- backward, data part, TWO WAVES PER SIMD (D = 128) ------------------------------------------------------------
// The kernel above keeps the whole e / de tile of a subtile in registers (2 x 3 x D/4 = 192 at D = 128) and lands at
// 512 registers per wave: one wave per SIMD, nothing hides the LDS / MFMA / VALU latencies of the main loop (MFMA pipe
// 41 % busy).  Here a tile is processed in two passes over HALVES of the feature blocks, so that only 2 x 3 x D/8 = 96
// accumulator / operand registers are live in the main loop and two waves fit a SIMD:
//   A  type attention (full vectors, transient) -> bw; vector-level pre-activations -> dpv, yvec, dfeat
//   B  for half h: e_h = bw t_h; main (c, b) loop over the b-blocks of the half (same LDS ring of Wf rows, the stream now
//      runs half 0 of every filter, then half 1); vector-level contribution; partial softmax dots; de_h -> dT (scratch)
//   C  the type attention's pre-activations (kept in LDS since A) -> dS, dq, dp
//   D  dt = bw de + U dS, reading de back from dT
template <int D, int DOUT, int A>
__global__ __launch_bounds__(kFuseThreads, 2) void tgcn_fuse_bwd2_kernel(
    const float* __restrict__ T0, const float* __restrict__ T1, const float* __restrict__ T2, int64_t n,
    const float* __restrict__ U, const float* __restrict__ qv, const float* __restrict__ pv,
    const float* __restrict__ wb, const float* __restrict__ w1, const float* __restrict__ w2,
    const float* __restrict__ w3, const float* __restrict__ Wf, const float* __restrict__ outv,
    const float* __restrict__ dOut, float* __restrict__ dT0, float* __restrict__ dT1, float* __restrict__ dT2,
    float* __restrict__ yvec, float* __restrict__ dfeat, float* __restrict__ dS, float* __restrict__ part) {
  constexpr int DS = D / 4, HS = DS / 2, OS = DOUT / 4, IB = D / 16, HB = IB / 2, AB = A / 16, AS = A / 4;
  constexpr int NSM = 3 * kBitC + 2 * A;             // dwb | dq | dp
  __shared__ float sh[NSM];
  __shared__ float sh_wb4[3 * kBitC * 4];            // dwb partials, one slot per lane row q: no same-address LDS atomics
  constexpr int NB = 2;
  constexpr int RP = 256 / DOUT, PPW = (NB * 16 / RP) / 4, CHUNK = NB * 16 * DOUT * 4;
  constexpr int NBUF = 3, CPH = HB / NB, NCH = 2 * kBitC * CPH;   // chunks per (filter, half); chunks per tile group
  static_assert(HB % NB == 0 && CPH >= 1 && NCH >= 4, "tgcn_fuse_bwd2: unsupported shape");
  __shared__ __attribute__((aligned(1024))) char wbuf[NBUF][CHUNK];
  // the type attention's pre-activations of the tile (3 A / 4 floats per lane), kept from phase A for phase C: recomputing
  // them there (192 MFMAs fed by 96 row loads and the U stream) cost 3.5 % of the kernel, the registers to hold them
  // through phase B do not exist
  __shared__ f32x4 sc_keep[3 * AB][kFuseThreads];
  for (int i = threadIdx.x; i < NSM; i += kFuseThreads) sh[i] = 0.f;
  for (int i = threadIdx.x; i < 3 * kBitC * 4; i += kFuseThreads) sh_wb4[i] = 0.f;
  __syncthreads();
  float* sh_wb = sh;
  float* sh_q = sh + 3 * kBitC;
  float* sh_p = sh_q + A;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int64_t n_tiles = (n + 15) / 16;
  const float* Tj[3] = {T0, T1, T2};
  float* dTj[3] = {dT0, dT1, dT2};
  auto swz = [](int row) {
    const int x = ((row >> 3) ^ (row >> 2)) & 1;
    return DOUT == 128 ? ((row & 0xB) | (x << 2)) : ((row & 0x7) | (x << 3));
  };
  const int lanes_per_row = DOUT / 4;
  const int dma_row = lane / lanes_per_row, dma_unit = lane % lanes_per_row;
  const int drow = (r >> 2) * DS + (r & 3);
  float q_acc[AS], p_acc[AS];
#pragma unroll
  for (int i = 0; i < AS; ++i) { q_acc[i] = 0.f; p_acc[i] = 0.f; }
  const int64_t it_first = static_cast<int64_t>(blockIdx.x), it_step = static_cast<int64_t>(gridDim.x);
  const int64_t it_end = (n_tiles + 3) / 4;
  // lane part of the DMA source offsets, once (a VALU instruction costs 1/8 of an MFMA and does not overlap with one)
  int dma_off[PPW];
#pragma unroll
  for (int jj = 0; jj < PPW; ++jj) {
    const int lrow = (wave * PPW + jj) * RP + dma_row;
    const int sub = lrow >> 4, i = lrow & 15;
    dma_off[jj] = ((i >> 2) * DS + (i & 3) + 4 * sub) * DOUT + (dma_unit ^ swz(i)) * 4;
  }
  const int wave_s = __builtin_amdgcn_readfirstlane(wave);
  auto issue = [&](int j, int buf) {                     // chunk j of the tile group's stream
    const int h = j / (kBitC * CPH), rem = j % (kBitC * CPH);
    const int c = rem / CPH, b = h * HB + (rem % CPH) * NB;
    const float* base = Wf + (static_cast<int64_t>(c) * D + 4 * b) * DOUT;
#pragma unroll
    for (int jj = 0; jj < PPW; ++jj)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + dma_off[jj]),
                                       (__attribute__((address_space(3))) void*)&wbuf[buf][(wave_s * PPW + jj) * 1024], 16, 0, 0);
  };
  int cur = 0;
  if (it_first < it_end) { issue(0, 0); issue(1, 1); }
  for (int64_t it = it_first; it < it_end; it += it_step) {
    const int64_t tile = it * 4 + wave;
    const bool has_next = it + it_step < it_end;
    U = fresh(U); qv = fresh(qv); pv = fresh(pv); wb = fresh(wb); w1 = fresh(w1); w2 = fresh(w2); w3 = fresh(w3);
    Wf = fresh(Wf);
    const int64_t node = tile * 16 + r;
    const bool ok = node < n;
    float g[OS];
    {
      float o[OS];
      load_seg<OS>(dOut + node * DOUT + q * OS, ok, g);
      load_seg<OS>(outv + node * DOUT + q * OS, ok, o);
#pragma unroll
      for (int i = 0; i < OS; ++i) g[i] = o[i] > 0.f ? g[i] : 0.f;
    }
    // ---- A: attention weights and the vector-level features (full vectors, transient)
    float bw[3];
    f32x4 dpv[6];
    {
      float e3[3][DS];
#pragma unroll
      for (int j = 0; j < 3; ++j) load_seg<DS>(Tj[j] + node * D + q * DS, ok, e3[j]);
      f32x4 sc[3][AB];
      type_attention<D, A>(e3, U, qv, pv, r, q, sc, bw);
#pragma unroll
      for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int ab = 0; ab < AB; ++ab) sc_keep[j * AB + ab][threadIdx.x] = sc[j][ab];      // (read back by this thread only)
      f32x4 pre[6];
      vector_conv<D>(e3, w1, w2, w3, r, q, pre);
#pragma unroll
      for (int gi = 0; gi < 6; ++gi) {
        const int f = vec_feature(gi, r & 7);
        float a[OS];
        load_run<OS>(Wf + (static_cast<int64_t>(kBitC) * D + f) * DOUT + q * OS, a);
        f32x4 dy = zero4();
#pragma unroll
        for (int t = 0; t < OS; ++t) dy = __builtin_amdgcn_mfma_f32_16x16x4f32(r < kVecC ? a[t] : 0.f, g[t], dy, 0, 0, 0);
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const bool live = q < 2;
          const float y = live ? fmaxf(pre[gi][v], 0.f) : 0.f;
          dpv[gi][v] = (live && pre[gi][v] > 0.f) ? dy[v] : 0.f;
          if (ok && live) {
            const int ff = vec_feature(gi, 4 * q + v);
            yvec[node * (6 * kVecC) + ff] = y;
            dfeat[node * (6 * kVecC) + ff] = dpv[gi][v];
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // ---- B: the two halves of the feature blocks
    float db[3] = {0.f, 0.f, 0.f};
    for (int h = 0; h < 2; ++h) {
      float e3h[3][HS], de3h[3][HS];
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        load_seg<HS>(Tj[j] + node * D + q * DS + h * HS, ok, e3h[j]);
#pragma unroll
        for (int e = 0; e < HS; ++e) { e3h[j][e] *= bw[j]; de3h[j][e] = 0.f; }
      }
      for (int c = 0; c < kBitC; ++c) {
        const float c0 = wb[c * 3], c1 = wb[c * 3 + 1], c2 = wb[c * 3 + 2];
        float a0 = 0.f, a1 = 0.f, a2 = 0.f;
#pragma unroll
        for (int bl = 0; bl < HB; ++bl) {
          if (bl % NB == 0) {
            const int j = (h * kBitC + c) * CPH + bl / NB;      // chunk number inside the tile group
            if (j + 1 < NCH || has_next) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(PPW) : "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            int j2 = j + 2;
            const bool wrap = j2 >= NCH;
            if (wrap) j2 -= NCH;
            if (!wrap || has_next) issue(j2, cur >= 1 ? cur - 1 : NBUF - 1);
          }
          f32x4 dy = zero4(), dy1 = zero4();
          const char* rowp = &wbuf[cur][((bl % NB) * 16 + r) * DOUT * 4];
          const int sw = swz(r);
          auto piece = [&](int t4) { return *reinterpret_cast<const float4*>(rowp + (((q * (OS / 4) + t4) ^ sw) << 4)); };
          float4 w0 = piece(0), w1_ = piece(1 < OS / 4 ? 1 : 0);
#pragma unroll
          for (int t4 = 0; t4 < OS / 4; ++t4) {
            const float4 w2_ = piece(t4 + 2 < OS / 4 ? t4 + 2 : t4);
            const int t = 4 * t4;
            dy = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.x, g[t], dy, 0, 0, 0);
            dy1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.y, g[t + 1], dy1, 0, 0, 0);
            dy = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.z, g[t + 2], dy, 0, 0, 0);
            dy1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.w, g[t + 3], dy1, 0, 0, 0);
            w0 = w1_; w1_ = w2_;
          }
          dy += dy1;
          if (bl % NB == NB - 1) cur = cur == NBUF - 1 ? 0 : cur + 1;
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            const int e = 4 * bl + v;
            const float pre = fmaf(c0, e3h[0][e], fmaf(c1, e3h[1][e], c2 * e3h[2][e]));
            const float dp_ = pre > 0.f ? dy[v] : 0.f;
            de3h[0][e] = fmaf(c0, dp_, de3h[0][e]);
            de3h[1][e] = fmaf(c1, dp_, de3h[1][e]);
            de3h[2][e] = fmaf(c2, dp_, de3h[2][e]);
            a0 = fmaf(dp_, e3h[0][e], a0);
            a1 = fmaf(dp_, e3h[1][e], a1);
            a2 = fmaf(dp_, e3h[2][e], a2);
          }
        }
        // over the 16 nodes of a lane row by DPP (register speed); the four rows add into their own LDS slots without a
        // return value (the two cross-row shuffle rounds went through the LDS pipe and stalled the wave once per filter;
        // four lanes adding into ONE address serialise and cost more than they save)
        a0 = row16_sum(a0); a1 = row16_sum(a1); a2 = row16_sum(a2);
        if (r == 0) {
          atomicAdd(&sh_wb4[(c * 3) * 4 + q], a0); atomicAdd(&sh_wb4[(c * 3 + 1) * 4 + q], a1); atomicAdd(&sh_wb4[(c * 3 + 2) * 4 + q], a2);
        }
      }
      // vector-level contribution to the half: de_h[d] += sum_c w[c][a][d] dpre[c]
#pragma unroll
      for (int bl = 0; bl < HB; ++bl) {
        const int d = drow + 4 * (h * HB + bl);
        f32x4 acc0 = zero4(), acc1 = zero4(), acc2 = zero4();
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int c = 4 * (q & 1) + v;
          const float z = q < 2 ? 1.f : 0.f;
          const float k1 = z * w1[c * D + d];
          const float k20 = z * w2[(c * 2 + 0) * D + d], k21 = z * w2[(c * 2 + 1) * D + d];
          const float k30 = z * w3[(c * 3 + 0) * D + d], k31 = z * w3[(c * 3 + 1) * D + d], k32 = z * w3[(c * 3 + 2) * D + d];
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(k1, dpv[0][v], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(k1, dpv[1][v], acc1, 0, 0, 0);
          acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(k1, dpv[2][v], acc2, 0, 0, 0);
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(k20, dpv[3][v], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(k21, dpv[3][v], acc1, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(k20, dpv[4][v], acc1, 0, 0, 0);
          acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(k21, dpv[4][v], acc2, 0, 0, 0);
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(k30, dpv[5][v], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(k31, dpv[5][v], acc1, 0, 0, 0);
          acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(k32, dpv[5][v], acc2, 0, 0, 0);
        }
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          de3h[0][4 * bl + v] += acc0[v];
          de3h[1][4 * bl + v] += acc1[v];
          de3h[2][4 * bl + v] += acc2[v];
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      // the half's share of db_j = de_j . t_j, and de_h out to dT (read back in D)
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        float th[HS];
        load_seg<HS>(Tj[j] + node * D + q * DS + h * HS, ok, th);
        float s_ = 0.f;
#pragma unroll
        for (int e = 0; e < HS; ++e) s_ = fmaf(de3h[j][e], th[e], s_);
        db[j] += s_;
        if (ok) {
          float* dst = dTj[j] + node * D + q * DS + h * HS;
#pragma unroll
          for (int e = 0; e < HS; e += 4)
            *reinterpret_cast<float4*>(dst + e) = make_float4(de3h[j][e], de3h[j][e + 1], de3h[j][e + 2], de3h[j][e + 3]);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) db[j] = quad_sum(db[j]);
    // ---- C: through the type-level softmax (pre-activations kept in LDS since phase A)
    float dsv[3][AS];
    {
      f32x4 sc[3][AB];
#pragma unroll
      for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int ab = 0; ab < AB; ++ab) sc[j][ab] = sc_keep[j * AB + ab][threadIdx.x];
      const float mix = bw[0] * db[0] + bw[1] * db[1] + bw[2] * db[2];
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const float ds = bw[j] * (db[j] - mix);
#pragma unroll
        for (int ab = 0; ab < AB; ++ab)
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            const int t_ = v * AB + ab;
            const float hh = sc[j][ab][v];
            const float x = hh > 0.f ? ds * pv[q * AS + t_] : 0.f;
            dsv[j][t_] = x;
            q_acc[t_] += x;
            p_acc[t_] = fmaf(ds, fmaxf(hh, 0.f), p_acc[t_]);
          }
        if (ok) {
          float* dst = dS + node * (3 * A) + j * A + q * AS;
#pragma unroll
          for (int t_ = 0; t_ < AS; t_ += 4)
            *reinterpret_cast<float4*>(dst + t_) = make_float4(dsv[j][t_], dsv[j][t_ + 1], dsv[j][t_ + 2], dsv[j][t_ + 3]);
        }
      }
    }
    // ---- D: dt_j = bw_j de_j + U dS_j  (de_j read back from dT: written by this lane above)
#pragma unroll
    for (int b = 0; b < IB; ++b) {
      float ua[AS];
      load_run<AS>(U + static_cast<int64_t>(drow + 4 * b) * A + q * AS, ua);
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        f32x4 acc = zero4();
#pragma unroll
        for (int t_ = 0; t_ < AS; ++t_) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ua[t_], dsv[j][t_], acc, 0, 0, 0);
        if (ok) {
          float4* p4 = reinterpret_cast<float4*>(dTj[j] + node * D + q * DS + 4 * b);
          const float4 de = *p4;
          *p4 = make_float4(fmaf(bw[j], de.x, acc[0]), fmaf(bw[j], de.y, acc[1]), fmaf(bw[j], de.z, acc[2]), fmaf(bw[j], de.w, acc[3]));
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
#pragma unroll
  for (int i = 0; i < AS; ++i) {
    float a = q_acc[i], b = p_acc[i];
    a = row16_sum(a);
    b = row16_sum(b);
    if (r == 0) { atomicAdd(&sh_q[q * AS + i], a); atomicAdd(&sh_p[q * AS + i], b); }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 3 * kBitC; i += kFuseThreads)
    sh_wb[i] = (sh_wb4[i * 4] + sh_wb4[i * 4 + 1]) + (sh_wb4[i * 4 + 2] + sh_wb4[i * 4 + 3]);
  __syncthreads();
  float* o = part + static_cast<int64_t>(blockIdx.x) * NSM;
  for (int i = threadIdx.x; i < NSM; i += kFuseThreads) o[i] = sh[i];
}

__global__ void fuse_fold_kernel(const float* __restrict__ part, int n_parts, int elems, float* __restrict__ out) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= elems) return;
  float s = 0.f;
  for (int b = 0; b < n_parts; ++b) s += part[static_cast<int64_t>(b) * elems + e];
  out[e] = s;
}

constexpr int kFuseBwdBlocks = 512;
constexpr int kLightBatch = 4;      // k-steps whose operands the light paths of the weight-gradient kernel load at once

template <int D, int DOUT>
int launch_fuse_bwd(const float* T0, const float* T1, const float* T2, int64_t n, const float* U, const float* qv,
                    const float* pv, const float* wb, const float* w1, const float* w2, const float* w3, const float* Wf,
                    const float* outv, const float* dOut, float* dT0, float* dT1, float* dT2, float* yvec, float* dfeat,
                    float* dS, float* small, float* ws, hipStream_t s) {
  constexpr int A = 32;
  constexpr int NSM = 3 * kBitC + 2 * A;
  const int64_t tiles = (n + 15) / 16;
  int64_t blocks = (tiles + 3) / 4;
  if (blocks > kFuseBwdBlocks) blocks = kFuseBwdBlocks;
  constexpr bool kLds = (D == 64 || D == 128) && (DOUT == 64 || DOUT == 128);
  if constexpr (D == 128 && kLds) {
    // two waves per SIMD: the tile's feature blocks in two passes (see tgcn_fuse_bwd2_kernel)
    tgcn_fuse_bwd2_kernel<D, DOUT, A><<<static_cast<unsigned>(blocks), kFuseThreads, 0, s>>>(
        T0, T1, T2, n, U, qv, pv, wb, w1, w2, w3, Wf, outv, dOut, dT0, dT1, dT2, yvec, dfeat, dS, ws);
  } else {
    tgcn_fuse_bwd_kernel<D, DOUT, A, kLds><<<static_cast<unsigned>(blocks), kFuseThreads, 0, s>>>(
        T0, T1, T2, n, U, qv, pv, wb, w1, w2, w3, Wf, outv, dOut, dT0, dT1, dT2, yvec, dfeat, dS, ws);
  }
  TAGREC_LAUNCH_CHECK();
  fuse_fold_kernel<<<(NSM + 255) / 256, 256, 0, s>>>(ws, static_cast<int>(blocks), NSM, small);
  TAGREC_LAUNCH_CHECK();
  return TAGREC_OK;
}


// ---- backward, fusion weight: dWf[k][o] = sum_nodes y[node][k] g[node][o] ---------------------------
// y is re-formed on the fly (it is never stored): here the contraction runs over nodes, so node rows sit on the
// MFMA k axis and both operands are read straight from global memory in their natural layout.
// grid = (10 k-groups) x (node groups): k-group < 8 = four bit-level channels (one per wave, a full D x Dout
// accumulator tile each), k-group 8 = the 48 vector-level rows (16 per wave), k-group 9 = the small weight
// gradients G (vector-level filters) and dU (type attention), which are the same kind of node-axis product.  Per-node-group partial sums are
// folded in group order by fuse_fold_kernel (deterministic).
template <int D, int DOUT, int NH>
__global__ __launch_bounds__(kFuseThreads * NH) void tgcn_fuse_wf_kernel(
    const float* __restrict__ T0, const float* __restrict__ T1, const float* __restrict__ T2, const float* __restrict__ bw,
    const float* __restrict__ yvec, const float* __restrict__ wb, const float* __restrict__ outv,
    const float* __restrict__ dOut, const float* __restrict__ dfeat, const float* __restrict__ dS, int64_t n,
    int64_t rows_per_group, int main_groups, int64_t rows_per_small_group, float* __restrict__ part_main,
    float* __restrict__ part_small) {
  constexpr int IB = D / 16, OB = DOUT / 16;
  constexpr int A = 32, AB = A / 16, NF = 6 * kVecC, FB = NF / 16;
  constexpr int64_t KBIT = static_cast<int64_t>(kBitC) * D * DOUT;                       // the bit-level rows of dWf
  constexpr int64_t SMALL = static_cast<int64_t>(NF) * DOUT + static_cast<int64_t>(NF) * 3 * D + static_cast<int64_t>(D) * A;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int m = lane & 15, q = lane >> 4;
  // The first 8 * main_groups blocks are the heavy ones (k-groups 0..7: four bit-level channels each, operands shared
  // through LDS); the remaining blocks take the two light k-groups (vector-level rows; G and dU) over their own,
  // finer node groups: those are load-latency bound, so they get more blocks with fewer nodes each.
  const bool heavy = static_cast<int>(blockIdx.x) < 8 * main_groups;
  const int rest = static_cast<int>(blockIdx.x) - 8 * main_groups;
  const int kg = heavy ? blockIdx.x % 8 : 8 + rest % 2;
  const int64_t ng = heavy ? blockIdx.x / 8 : rest / 2;
  const int64_t rpg = heavy ? rows_per_group : rows_per_small_group;
  const int64_t lo = ng * rpg;
  const int64_t hi = (lo + rpg < n) ? lo + rpg : n;
  // partial results: heavy blocks [group][KBIT]; light blocks [group][vec rows of dWf | G | dU] (indexed as in [dWf | G | dU])
  float* dst = heavy ? part_main + ng * KBIT : part_small + ng * SMALL - KBIT;
  if (!heavy && wave >= 4) return;               // the light paths are written for four waves
  if (kg == 9) {
    // small weight gradients with nodes on the k axis:
    //   G[f][j][d] = sum_nodes dfeat[node][f] e_j[node][d]   (the caller folds G into dw1, dw2, dw3)
    //   dU[d][a]   = sum_nodes sum_j t_j[node][d] dS_j[node][a]
    float* gdst = dst + KBIT + static_cast<int64_t>(NF) * DOUT;
    float* udst = gdst + static_cast<int64_t>(NF) * 3 * D;
    const float* Tj[3] = {T0, T1, T2};
    constexpr int IBW = (IB + 3) / 4;              // input-feature blocks per wave: ib = wave + 4 i
    f32x4 accg[IBW][FB][3], accu[IBW][AB];
#pragma unroll
    for (int i = 0; i < IBW; ++i) {
#pragma unroll
      for (int f = 0; f < FB; ++f)
#pragma unroll
        for (int j = 0; j < 3; ++j) accg[i][f][j] = zero4();
#pragma unroll
      for (int a = 0; a < AB; ++a) accu[i][a] = zero4();
    }
    struct RawS { float df[FB], b[3], t[3][IBW], ds[3][AB]; };
    auto fetch_s = [&](int64_t node0, RawS& rw) {
      const int64_t node = node0 + q;
      const bool ok = node < hi;
#pragma unroll
      for (int f = 0; f < FB; ++f) rw.df[f] = ok ? dfeat[node * NF + f * 16 + m] : 0.f;
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        rw.b[j] = ok ? bw[node * 3 + j] : 0.f;
#pragma unroll
        for (int i = 0; i < IBW; ++i) {
          const int ib = wave + 4 * i;
          rw.t[j][i] = (ok && ib < IB) ? Tj[j][node * D + ib * 16 + m] : 0.f;
        }
#pragma unroll
        for (int a = 0; a < AB; ++a) rw.ds[j][a] = ok ? dS[node * (3 * A) + j * A + a * 16 + m] : 0.f;
      }
    };
    // load-latency bound: the operands of kLightBatch k-steps are fetched together, then multiplied
    for (int64_t node0 = lo; node0 < hi; node0 += 4 * kLightBatch) {
      RawS rs[kLightBatch];
#pragma unroll
      for (int u = 0; u < kLightBatch; ++u) fetch_s(node0 + 4 * u, rs[u]);
#pragma unroll
      for (int u = 0; u < kLightBatch; ++u) {
        const RawS& cur = rs[u];
#pragma unroll
        for (int i = 0; i < IBW; ++i) {
#pragma unroll
          for (int j = 0; j < 3; ++j) {
            const float e = cur.t[j][i] * cur.b[j];
#pragma unroll
            for (int f = 0; f < FB; ++f) accg[i][f][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.df[f], e, accg[i][f][j], 0, 0, 0);
#pragma unroll
            for (int a = 0; a < AB; ++a) accu[i][a] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.t[j][i], cur.ds[j][a], accu[i][a], 0, 0, 0);
          }
        }
      }
    }
#pragma unroll
    for (int i = 0; i < IBW; ++i) {
      const int ib = wave + 4 * i;
      if (ib >= IB) continue;
#pragma unroll
      for (int f = 0; f < FB; ++f)
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
          for (int v = 0; v < 4; ++v) gdst[(static_cast<int64_t>(f * 16 + q * 4 + v) * 3 + j) * D + ib * 16 + m] = accg[i][f][j][v];
#pragma unroll
      for (int a = 0; a < AB; ++a)
#pragma unroll
        for (int v = 0; v < 4; ++v) udst[static_cast<int64_t>(ib * 16 + q * 4 + v) * A + a * 16 + m] = accu[i][a][v];
    }
    return;
  }
  if (kg < 8) {
    // NH = 2 (D = Dout = 128): a channel's D x Dout accumulator tile is split over two waves (rows [half IBH, ..)), so a
    // wave holds 128 accumulator registers instead of 256 and two waves fit a SIMD -- with one, every wait was exposed
    constexpr int IBH = IB / NH;
    const int c = 4 * kg + (wave & 3), half = wave >> 2;
    const float c0 = wb[c * 3], c1 = wb[c * 3 + 1], c2 = wb[c * 3 + 2];
    f32x4 acc[IBH][OB];
#pragma unroll
    for (int i = 0; i < IBH; ++i)
#pragma unroll
      for (int o = 0; o < OB; ++o) acc[i][o] = zero4();
    if constexpr ((D == 64 || D == 128) && (DOUT == 64 || DOUT == 128)) {
      // The four waves (four channels) need the SAME rows of T0, T1, T2, dOut, out and bw; loading them per wave made
      // 32 waves re-read every node row from L2 (80 GB per call at C4: the kernel sat at 36 % MFMA-busy on that).
      // They are staged once per block through LDS instead: 8 nodes (two MFMA k-steps) per stage, LDS-DMA, two
      // buffers.  A row's 16-byte units are rotated by 4 * (node % 4) so that the b32 reads of the four nodes a
      // k-step touches (lanes q = 0..3) fall on disjoint 16-bank ranges.
      constexpr int SN = 8;                                   // nodes per stage
      constexpr int TB = SN * D * 4, GB = SN * DOUT * 4;      // bytes per staged tensor
      // T0 T1 T2 | dOut | out | bw (8 x 3 floats, padded) | yvec [8, 48] | dfeat [8, 48] | dS [8, 3 A]: the last three
      // feed the LIGHT products (vector-level rows of dWf, G, dU), which the heavy blocks carry along (see below)
      constexpr int BWO = 3 * TB + 2 * GB, YVO = BWO + 1024, DFO = YVO + 2048, DSO = DFO + 2048;
      constexpr int PV = (SN * NF * 4 + 1023) / 1024, PS = (SN * 3 * A * 4 + 1023) / 1024;   // 1 KiB pieces of yvec / dfeat, of dS
      constexpr int STAGE = DSO + PS * 1024;
      static_assert(PV * 1024 <= 2048 && SN * NF % 4 == 0, "stage layout");
      static_assert(D != 128 || DOUT != 128 || STAGE == 28672, "stage size");
      // Node rows come from HBM the first time (eight blocks share a node range, the other seven hit L2): a stage is
      // two k-steps of MFMAs, far shorter than that miss, so THREE stages are kept in flight (four buffers) and the
      // waits are counted -- `s_waitcnt vmcnt(k * CNT)` + a raw s_barrier, never the vmcnt(0) of __syncthreads().
      constexpr int AHEAD = 3, NBUF = AHEAD + 1;
      __shared__ __attribute__((aligned(1024))) char sbuf[NBUF][STAGE];
      constexpr int WPB = 4 * NH;
      constexpr int PT = TB / 1024, PG = GB / 1024;           // 1 KiB pieces per tensor
      constexpr int NMAINP = 3 * PT + 2 * PG + 1;             // + the bw piece
      constexpr int NPIECE = NMAINP + 2 * PV + PS;            // + yvec, dfeat, dS
      constexpr int CNT = (NPIECE + WPB - 1) / WPB;           // DMA instructions per wave and stage (uniform: see below)
      // LIGHT products, carried along: the vector-level rows of dWf (yvec^T g), G (dfeat^T e_j) and dU (t_j^T dS_j) are 7 %
      // of the flops but were 28 % of the blocks when they had blocks of their own (load-latency bound: one float per lane
      // straight from global).  Their accumulator tiles are dealt to the 8 x WPB waves that share a node group -- tile ids
      // u, u + NW, ..: [0, NTU) dU (three MFMAs per k-step), then the vector rows, then G -- and their operands come out
      // of the stage the heavy product already has in LDS (+ yvec, dfeat, dS: 6 KB per stage).
      constexpr int NTU = IB * AB, NTV = FB * OB, NTG = FB * 3 * IB, NT = NTU + NTV + NTG;
      constexpr int NW = 8 * WPB, MAXT = (NT + NW - 1) / NW;
      const int u = kg * WPB + __builtin_amdgcn_readfirstlane(wave);    // wave-uniform, in a scalar register
      // per tile: kind (0 dU, 1 vector rows, 2 G, 3 none), the relation j of a G tile, and the lane's float offsets of its
      // A / B operands inside a stage for k-step 0 (a k-step later = 4 node rows further)
      f32x4 lacc[MAXT];
      int lkind[MAXT], lj[MAXT], la[MAXT], lb[MAXT];
      auto rot = [&](int blk, int W) { return q * W + 4 * ((4 * blk + (m >> 2) + 4 * q) & (W / 4 - 1)) + (m & 3); };
      // a dU tile costs three MFMAs per k-step, the others one: where the other tiles fit on the remaining waves, a wave
      // with a dU tile carries nothing else (D = Dout = 128: 16 waves x 3 and 48 waves x 2 instead of up to 4)
      constexpr bool kSoloU = NTU <= NW && (NT - NTU) <= MAXT * (NW - NTU);
      auto tile_of = [&](int it) {
        if constexpr (kSoloU) return u < NTU ? (it == 0 ? u : NT) : NTU + it * (NW - NTU) + (u - NTU);
        else return u + it * NW;
      };
#pragma unroll
      for (int it = 0; it < MAXT; ++it) {
        lacc[it] = zero4();
        const int tile = tile_of(it);
        lkind[it] = 3; lj[it] = 0; la[it] = 0; lb[it] = 0;
        if (tile < NTU) {
          lkind[it] = 0;
          la[it] = rot(tile / AB, D);                                  // t_j row block db (A operand, rows = d)
          lb[it] = q * (3 * A) + (tile % AB) * 16 + m;                 // dS_j block ab (B operand, cols = a)
        } else if (tile < NTU + NTV) {
          const int t = tile - NTU;
          lkind[it] = 1;
          la[it] = q * NF + (t / OB) * 16 + m;                         // yvec block fb
          lb[it] = rot(t % OB, DOUT);                                  // masked dOut block ob
        } else if (tile < NT) {
          const int t = tile - NTU - NTV;
          lkind[it] = 2;
          lj[it] = (t / IB) % 3;
          la[it] = q * NF + (t / (3 * IB)) * 16 + m;                   // dfeat block fb
          lb[it] = rot(t % IB, D);                                     // e_j block db
        }
      }
      const int64_t n_stage = (hi - lo + SN - 1) / SN;
      // iterations st = -AHEAD .. -1 only issue (stages 0 .. AHEAD-1); the DMA issue is written once, inline -- as a
      // lambda with two call sites its captures ended up in scratch memory, and a scratch load's wait is a vmcnt(0)
      for (int64_t st = -AHEAD; st < n_stage; ++st) {
        if (st >= 0) {
          const int64_t after = n_stage - 1 - st < AHEAD - 1 ? n_stage - 1 - st : AHEAD - 1;   // stages in flight beyond st
          if (after >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * CNT) : "memory");
          else if (after == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CNT) : "memory");
          else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();                                // stage st is in LDS for everyone; buffer (st-1) % NBUF is free
          asm volatile("" ::: "memory");
        }
        if (st + AHEAD < n_stage) {
          const int64_t node0 = lo + (st + AHEAD) * SN;
          const int buf = static_cast<int>((st + AHEAD) % NBUF);
        // piece ids: [0, 3 PT) rows of T0..T2, then PG of dOut, PG of out, last = bw.  Wave w takes ids w, w + WPB, ..;
        // a wave whose last slot has no piece repeats its previous one, so that every wave issues exactly CNT.
#pragma unroll
        for (int j = 0; j < CNT; ++j) {
          int id = wave + j * WPB;
          if (id >= NPIECE) id -= WPB;
          if (id == NMAINP - 1) {                               // bw: 24 consecutive floats of [n, 3]
            int64_t e = node0 * 3 + (lane < 3 * SN ? lane : 0);
            if (e >= n * 3) e = n * 3 - 1;
            if (lane < 3 * SN)
              __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(bw + e),
                                               (__attribute__((address_space(3))) void*)&sbuf[buf][BWO], 4, 0, 0);
            continue;
          }
          if (id >= NMAINP) {                                   // yvec / dfeat / dS: the stage's 8 rows are one contiguous run
            const int x = id - NMAINP;
            const float* src = x < PV ? yvec : (x < 2 * PV ? dfeat : dS);
            const int rf = x < 2 * PV ? NF : 3 * A;               // floats per node row
            const int piece = x < PV ? x : (x < 2 * PV ? x - PV : x - 2 * PV);
            const int off = x < PV ? YVO : (x < 2 * PV ? DFO : DSO);
            const int unit = piece * 64 + lane;                   // 16-byte unit inside the run
            int64_t e = node0 * rf + unit * 4;
            if (e > n * rf - 4) e = n * rf - 4;                   // clamped; rows past n are masked out when read
            if (unit * 4 < SN * rf)
              __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + e),
                                               (__attribute__((address_space(3))) void*)&sbuf[buf][off + piece * 1024], 16, 0, 0);
            continue;
          }
          const bool is_t = id < 3 * PT;
          const int gi = is_t ? 0 : (id - 3 * PT) / PG;                                  // 0: dOut, 1: out
          const int ti = id / PT;                                                          // (selects, not a pointer table in scratch)
          const float* base = is_t ? (ti == 0 ? T0 : (ti == 1 ? T1 : T2)) : (gi == 0 ? dOut : outv);
          const int W = is_t ? D : DOUT;
          const int piece = is_t ? id % PT : (id - 3 * PT) % PG;
          const int off_bytes = is_t ? (id / PT) * TB : 3 * TB + gi * GB;
          const int lanes_per_row = W / 4, rpp = 256 / W;
          const int nd = piece * rpp + lane / lanes_per_row;          // node slot 0..7 inside the stage
          const int upos = lane % lanes_per_row;                       // unit position inside the LDS row
          const int u = (upos - 4 * (nd & 3)) & (lanes_per_row - 1);   // source unit: rows are stored rotated
          int64_t node = node0 + nd;
          if (node >= n) node = n - 1;                                 // clamped; masked out when read
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + node * W + u * 4),
                                           (__attribute__((address_space(3))) void*)&sbuf[buf][off_bytes + piece * 1024], 16, 0, 0);
        }
        }
        if (st < 0) continue;
        const char* sb = sbuf[st % NBUF];
#pragma unroll
        for (int ks = 0; ks < SN / 4; ++ks) {
          const int nd = ks * 4 + q;                                   // node slot of this lane's k index
          const bool ok = lo + st * SN + nd < hi;
          const float* bwp = reinterpret_cast<const float*>(sb + BWO) + nd * 3;
          const float b0 = bwp[0], b1 = bwp[1], b2 = bwp[2];
          float y[IBH], g[OB];
#pragma unroll
          for (int i = 0; i < IBH; ++i) {
            const int w = nd * D + 4 * ((4 * (half * IBH + i) + (m >> 2) + 4 * q) & (D / 4 - 1)) + (m & 3);
            const float e0 = reinterpret_cast<const float*>(sb)[w] * b0;
            const float e1 = reinterpret_cast<const float*>(sb + TB)[w] * b1;
            const float e2 = reinterpret_cast<const float*>(sb + 2 * TB)[w] * b2;
            // same operation order as the forward kernel, so the ReLU mask is the forward's
            y[i] = fmaxf(fmaf(c0, e0, fmaf(c1, e1, c2 * e2)), 0.f);
          }
#pragma unroll
          for (int o = 0; o < OB; ++o) {
            const int w = nd * DOUT + 4 * ((4 * o + (m >> 2) + 4 * q) & (DOUT / 4 - 1)) + (m & 3);
            const float go = reinterpret_cast<const float*>(sb + 3 * TB)[w];
            const float ov = reinterpret_cast<const float*>(sb + 3 * TB + GB)[w];
            g[o] = (ok && ov > 0.f) ? go : 0.f;
          }
#pragma unroll
          for (int i = 0; i < IBH; ++i)
#pragma unroll
            for (int o = 0; o < OB; ++o) acc[i][o] = __builtin_amdgcn_mfma_f32_16x16x4f32(y[i], g[o], acc[i][o], 0, 0, 0);
#pragma unroll
          for (int it = 0; it < MAXT; ++it) {                           // this wave's light tiles (scalar branches)
            const int ao = la[it] + ks * 4 * (lkind[it] == 0 ? D : NF);
            if (lkind[it] == 0) {
              const int bo = lb[it] + ks * 4 * 3 * A;
#pragma unroll
              for (int j = 0; j < 3; ++j) {
                const float tv = reinterpret_cast<const float*>(sb + j * TB)[ao];
                const float dsv = reinterpret_cast<const float*>(sb + DSO)[bo + j * A];
                lacc[it] = __builtin_amdgcn_mfma_f32_16x16x4f32(ok ? tv : 0.f, dsv, lacc[it], 0, 0, 0);
              }
            } else if (lkind[it] == 1) {
              const int bo = lb[it] + ks * 4 * DOUT;
              const float yv = reinterpret_cast<const float*>(sb + YVO)[ao];
              const float go = reinterpret_cast<const float*>(sb + 3 * TB)[bo];
              const float ov = reinterpret_cast<const float*>(sb + 3 * TB + GB)[bo];
              lacc[it] = __builtin_amdgcn_mfma_f32_16x16x4f32(yv, (ok && ov > 0.f) ? go : 0.f, lacc[it], 0, 0, 0);
            } else if (lkind[it] == 2) {
              const int bo = lb[it] + ks * 4 * D;
              const float df = reinterpret_cast<const float*>(sb + DFO)[ao];
              const float e = reinterpret_cast<const float*>(sb + lj[it] * TB)[bo] * bwp[lj[it]];
              lacc[it] = __builtin_amdgcn_mfma_f32_16x16x4f32(ok ? df : 0.f, e, lacc[it], 0, 0, 0);
            }
          }
        }
      }
      // the light tiles' partial sums: [vector-level rows of dWf | G | dU] of this node group
      float* sdst = part_small + ng * SMALL;
#pragma unroll
      for (int it = 0; it < MAXT; ++it) {
        const int tile = tile_of(it);
        if (tile < NTU) {
          const int db = tile / AB, ab = tile % AB;
          float* udst = sdst + static_cast<int64_t>(NF) * DOUT + static_cast<int64_t>(NF) * 3 * D;
#pragma unroll
          for (int v = 0; v < 4; ++v) udst[static_cast<int64_t>(db * 16 + q * 4 + v) * A + ab * 16 + m] = lacc[it][v];
        } else if (tile < NTU + NTV) {
          const int t = tile - NTU, fb = t / OB, ob = t % OB;
#pragma unroll
          for (int v = 0; v < 4; ++v) sdst[static_cast<int64_t>(fb * 16 + q * 4 + v) * DOUT + ob * 16 + m] = lacc[it][v];
        } else if (tile < NT) {
          const int t = tile - NTU - NTV, fb = t / (3 * IB), j = (t / IB) % 3, db = t % IB;
          float* gdst = sdst + static_cast<int64_t>(NF) * DOUT;
#pragma unroll
          for (int v = 0; v < 4; ++v) gdst[(static_cast<int64_t>(fb * 16 + q * 4 + v) * 3 + j) * D + db * 16 + m] = lacc[it][v];
        }
      }
    } else {
      static_assert(NH == 1 || ((D == 64 || D == 128) && (DOUT == 64 || DOUT == 128)), "row halves only with staged operands");
    // software pipeline: the raw operands of step s+1 are in flight while step s runs its IB x OB MFMAs
    // (one wave per SIMD here -- the accumulator tile fills the register file -- so nothing else hides the latency)
    struct Raw { float t0[IB], t1[IB], t2[IB], go[OB], ov[OB], b0, b1, b2; };
    auto fetch = [&](int64_t node0, Raw& rw) {
      const int64_t node = node0 + q;
      const bool ok = node < hi;
      rw.b0 = ok ? bw[node * 3] : 0.f; rw.b1 = ok ? bw[node * 3 + 1] : 0.f; rw.b2 = ok ? bw[node * 3 + 2] : 0.f;
#pragma unroll
      for (int i = 0; i < IB; ++i) {
        const int64_t off = node * D + i * 16 + m;
        rw.t0[i] = ok ? T0[off] : 0.f; rw.t1[i] = ok ? T1[off] : 0.f; rw.t2[i] = ok ? T2[off] : 0.f;
      }
#pragma unroll
      for (int o = 0; o < OB; ++o) {
        const int64_t off = node * DOUT + o * 16 + m;
        rw.go[o] = ok ? dOut[off] : 0.f; rw.ov[o] = ok ? outv[off] : 0.f;
      }
    };
    Raw cur, nxt;
    fetch(lo, cur);
    for (int64_t node0 = lo; node0 < hi; node0 += 4) {
      fetch(node0 + 4, nxt);                       // rows past `hi` come back as zeros
      float y[IB], g[OB];
#pragma unroll
      for (int i = 0; i < IB; ++i) {
        // same operation order as the forward kernel, so the ReLU mask is the forward's
        const float e0 = cur.t0[i] * cur.b0, e1 = cur.t1[i] * cur.b1, e2 = cur.t2[i] * cur.b2;
        y[i] = fmaxf(fmaf(c0, e0, fmaf(c1, e1, c2 * e2)), 0.f);
      }
#pragma unroll
      for (int o = 0; o < OB; ++o) g[o] = cur.ov[o] > 0.f ? cur.go[o] : 0.f;
#pragma unroll
      for (int i = 0; i < IB; ++i)
#pragma unroll
        for (int o = 0; o < OB; ++o) acc[i][o] = __builtin_amdgcn_mfma_f32_16x16x4f32(y[i], g[o], acc[i][o], 0, 0, 0);
      cur = nxt;
    }
    }
#pragma unroll
    for (int i = 0; i < IBH; ++i)
#pragma unroll
      for (int o = 0; o < OB; ++o)
#pragma unroll
        for (int v = 0; v < 4; ++v)
          dst[(static_cast<int64_t>(c) * D + (half * IBH + i) * 16 + q * 4 + v) * DOUT + o * 16 + m] = acc[i][o][v];
  } else if (wave < 3) {
    f32x4 acc[OB];
#pragma unroll
    for (int o = 0; o < OB; ++o) acc[o] = zero4();
    // load-latency bound (16 MFMAs per 4 nodes): the operands of kLightBatch k-steps are fetched together
    for (int64_t node0 = lo; node0 < hi; node0 += 4 * kLightBatch) {
      float y[kLightBatch], go[kLightBatch][OB], ov[kLightBatch][OB];
#pragma unroll
      for (int u = 0; u < kLightBatch; ++u) {
        const int64_t node = node0 + 4 * u + q;
        const bool ok = node < hi;
        y[u] = ok ? yvec[node * (6 * kVecC) + wave * 16 + m] : 0.f;
#pragma unroll
        for (int o = 0; o < OB; ++o) {
          const int64_t off = node * DOUT + o * 16 + m;
          go[u][o] = ok ? dOut[off] : 0.f;
          ov[u][o] = ok ? outv[off] : 0.f;
        }
      }
#pragma unroll
      for (int u = 0; u < kLightBatch; ++u)
#pragma unroll
        for (int o = 0; o < OB; ++o)
          acc[o] = __builtin_amdgcn_mfma_f32_16x16x4f32(y[u], ov[u][o] > 0.f ? go[u][o] : 0.f, acc[o], 0, 0, 0);
    }
#pragma unroll
    for (int o = 0; o < OB; ++o)
#pragma unroll
      for (int v = 0; v < 4; ++v)
        dst[(static_cast<int64_t>(kBitC) * D + wave * 16 + q * 4 + v) * DOUT + o * 16 + m] = acc[o][v];
  }
}

constexpr int kWfGroups = 23;        // unstaged shapes: 8 x 23 heavy blocks + 2 x 36 light ones = 256: one block per CU; the split
constexpr int kWfSmallGroups = 36;   // equalises their measured times (heavy-only 13.4 ms at 26 groups, light-only 11.3 ms at 48; 1 M nodes, D = 128)
constexpr int kWfMergedGroups = 32;  // staged shapes (D, Dout in {64, 128}): 8 x 32 blocks, light products carried along
static_assert(kWfMergedGroups <= kWfSmallGroups && kWfMergedGroups >= kWfGroups, "workspace layout");

template <int D, int DOUT>
int launch_fuse_wf(const float* T0, const float* T1, const float* T2, const float* bw, const float* yvec, const float* wb,
                   const float* outv, const float* dOut, const float* dfeat, const float* dS, int64_t n, float* dWf,
                   float* ws, hipStream_t s) {
  constexpr int64_t KBIT = static_cast<int64_t>(kBitC) * D * DOUT;
  constexpr int64_t SMALL = static_cast<int64_t>(6 * kVecC) * DOUT + static_cast<int64_t>(6 * kVecC) * 3 * D + static_cast<int64_t>(D) * 32;
  auto split = [&](int64_t want, int64_t* per) {
    int64_t groups = (n + 255) / 256;
    if (groups > want) groups = want;
    if (groups < 1) groups = 1;
    *per = ((n + groups - 1) / groups + 7) / 8 * 8;
    return (n + *per - 1) / *per;
  };
  // staged shapes: the heavy blocks carry the light products along (no light blocks), 8 x 32 blocks = one per CU
  constexpr bool kMerged = (D == 64 || D == 128) && (DOUT == 64 || DOUT == 128);
  int64_t per = 0, per_small = 0;
  const int64_t groups = split(kMerged ? kWfMergedGroups : kWfGroups, &per);
  const int64_t small_groups = kMerged ? 0 : split(kWfSmallGroups, &per_small);
  float* ws_small = ws + static_cast<int64_t>(kWfMergedGroups) * KBIT;
  constexpr int NH = (D == 128 && DOUT == 128) ? 2 : 1;
  tgcn_fuse_wf_kernel<D, DOUT, NH><<<static_cast<unsigned>(8 * groups + 2 * small_groups), kFuseThreads * NH, 0, s>>>(
      T0, T1, T2, bw, yvec, wb, outv, dOut, dfeat, dS, n, per, static_cast<int>(groups), per_small, ws, ws_small);
  TAGREC_LAUNCH_CHECK();
  // one contiguous result [dWf | G | dU] (the caller hands a buffer of that size): bit-level rows, then the rest
  fuse_fold_kernel<<<static_cast<unsigned>((KBIT + 255) / 256), 256, 0, s>>>(ws, static_cast<int>(groups), static_cast<int>(KBIT), dWf);
  TAGREC_LAUNCH_CHECK();
  fuse_fold_kernel<<<static_cast<unsigned>((SMALL + 255) / 256), 256, 0, s>>>(ws_small, static_cast<int>(kMerged ? groups : small_groups),
                                                                            static_cast<int>(SMALL), dWf + KBIT);
  TAGREC_LAUNCH_CHECK();
  return TAGREC_OK;
}

#define TAGREC_FUSE_DISPATCH(CALL)                                                        \
  switch (D * 1000 + Dout) {                                                              \
    case 16016: return CALL(16, 16);   case 16032: return CALL(16, 32);                   \
    case 16064: return CALL(16, 64);   case 16128: return CALL(16, 128);                  \
    case 32016: return CALL(32, 16);   case 32032: return CALL(32, 32);                   \
    case 32064: return CALL(32, 64);   case 32128: return CALL(32, 128);                  \
    case 64016: return CALL(64, 16);   case 64032: return CALL(64, 32);                   \
    case 64064: return CALL(64, 64);   case 64128: return CALL(64, 128);                  \
    case 128016: return CALL(128, 16); case 128032: return CALL(128, 32);                 \
    case 128064: return CALL(128, 64); case 128128: return CALL(128, 128);                \
    default: break;                                                                       \
  }                                                                                       \
  return fail(TAGREC_E_UNSUPPORTED, "tgcn_fuse: D and Dout must be 16, 32, 64 or 128")

}  // namespace tagrec

using namespace tagrec;

extern "C" int tagrec_tgcn_fuse_fwd_f32(const float* T0, const float* T1, const float* T2, int64_t n, int D, int Dout,
                                        int A, int C, int V, const float* U, const float* q, const float* p,
                                        const float* wb, const float* w1, const float* w2, const float* w3,
                                        const float* Wf, const float* bf, float* bw_out, float* out, void* stream) {
  TAGREC_REQUIRE(T0 && T1 && T2 && U && q && p && wb && w1 && w2 && w3 && Wf && bf && bw_out && out,
                 "tgcn_fuse_fwd: null pointer");
  if (A != 32 || C != kBitC || V != kVecC)
    return fail(TAGREC_E_UNSUPPORTED, "tgcn_fuse: built for dim_atten 32, num_bit_conv 32, num_vec_conv 8");
  TAGREC_REQUIRE(aligned16(T0) && aligned16(T1) && aligned16(T2) && aligned16(Wf) && aligned16(U) && aligned16(out),
                 "tgcn_fuse_fwd: rows must be 16-byte aligned");
  if (n <= 0) return TAGREC_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const FuseDrop drop{0.f, n, n, {nullptr, nullptr, nullptr}, {0, 0, 0}};
#define CALL(DD, OO) launch_fuse_fwd<DD, OO>(T0, T1, T2, n, U, q, p, wb, w1, w2, w3, Wf, bf, bw_out, out, drop, s)
  TAGREC_FUSE_DISPATCH(CALL);
#undef CALL
}

extern "C" int tagrec_tgcn_fuse_fwd_drop_f32(const float* T0, const float* T1, const float* T2, int64_t n, int D, int Dout,
                                             int A, int C, int V, const float* U, const float* q, const float* p,
                                             const float* wb, const float* w1, const float* w2, const float* w3,
                                             const float* Wf, const float* bf, float drop_p, const uint64_t* seeds3,
                                             const int64_t* rows0, const int64_t* rows1, const int64_t* rows2, int64_t lo1,
                                             int64_t lo2, float* bw_out, float* out, void* stream) {
  TAGREC_REQUIRE(T0 && T1 && T2 && U && q && p && wb && w1 && w2 && w3 && Wf && bf && bw_out && out && seeds3,
                 "tgcn_fuse_fwd_drop: null pointer");
  if (A != 32 || C != kBitC || V != kVecC)
    return fail(TAGREC_E_UNSUPPORTED, "tgcn_fuse: built for dim_atten 32, num_bit_conv 32, num_vec_conv 8");
  TAGREC_REQUIRE(aligned16(T0) && aligned16(T1) && aligned16(T2) && aligned16(Wf) && aligned16(U) && aligned16(out),
                 "tgcn_fuse_fwd_drop: rows must be 16-byte aligned");
  TAGREC_REQUIRE(drop_p >= 0.f && drop_p < 1.f && 0 <= lo1 && lo1 <= lo2 && lo2 <= n, "tgcn_fuse_fwd_drop: bad p or segment bounds");
  if (n <= 0) return TAGREC_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const FuseDrop drop{drop_p, lo1, lo2, {rows0, rows1, rows2}, {seeds3[0], seeds3[1], seeds3[2]}};
#define CALL(DD, OO) launch_fuse_fwd<DD, OO>(T0, T1, T2, n, U, q, p, wb, w1, w2, w3, Wf, bf, bw_out, out, drop, s)
  TAGREC_FUSE_DISPATCH(CALL);
#undef CALL
}

extern "C" int64_t tagrec_tgcn_fuse_bwd_workspace(int Dout) {
  (void)Dout;
  return static_cast<int64_t>(kFuseBwdBlocks) * (3 * kBitC + 2 * 32);
}

extern "C" int tagrec_tgcn_fuse_bwd_f32(const float* T0, const float* T1, const float* T2, int64_t n, int D, int Dout,
                                        int A, int C, int V, const float* U, const float* q, const float* p,
                                        const float* wb, const float* w1, const float* w2, const float* w3,
                                        const float* Wf, const float* out, const float* dOut, float* dT0, float* dT1,
                                        float* dT2, float* yvec, float* dfeat, float* dS, float* small,
                                        float* workspace, int64_t workspace_floats, void* stream) {
  TAGREC_REQUIRE(T0 && T1 && T2 && U && q && p && wb && w1 && w2 && w3 && Wf && out && dOut && dT0 && dT1 && dT2 && yvec &&
                     dfeat && dS && small && workspace, "tgcn_fuse_bwd: null pointer");
  if (A != 32 || C != kBitC || V != kVecC)
    return fail(TAGREC_E_UNSUPPORTED, "tgcn_fuse: built for dim_atten 32, num_bit_conv 32, num_vec_conv 8");
  TAGREC_REQUIRE(workspace_floats >= tagrec_tgcn_fuse_bwd_workspace(Dout), "tgcn_fuse_bwd: workspace too small");
  TAGREC_REQUIRE(aligned16(T0) && aligned16(T1) && aligned16(T2) && aligned16(Wf) && aligned16(U) && aligned16(out) &&
                     aligned16(dOut) && aligned16(dT0) && aligned16(dT1) && aligned16(dT2) && aligned16(dS),
                 "tgcn_fuse_bwd: rows must be 16-byte aligned");
  if (n <= 0) return TAGREC_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
#define CALL(DD, OO) launch_fuse_bwd<DD, OO>(T0, T1, T2, n, U, q, p, wb, w1, w2, w3, Wf, out, dOut, dT0, dT1, dT2, yvec, dfeat, dS, small, workspace, s)
  TAGREC_FUSE_DISPATCH(CALL);
#undef CALL
}

extern "C" int64_t tagrec_tgcn_fuse_wf_result(int D, int Dout) {      // floats in [dWf | G | dU]
  return (static_cast<int64_t>(kBitC) * D + 6 * kVecC) * Dout + static_cast<int64_t>(6 * kVecC) * 3 * D +
         static_cast<int64_t>(D) * 32;
}

extern "C" int64_t tagrec_tgcn_fuse_wf_workspace(int D, int Dout) {
  const int64_t kbit = static_cast<int64_t>(kBitC) * D * Dout;
  return static_cast<int64_t>(kWfMergedGroups) * kbit + static_cast<int64_t>(kWfSmallGroups) * (tagrec_tgcn_fuse_wf_result(D, Dout) - kbit);
}

extern "C" int tagrec_tgcn_fuse_wf_f32(const float* T0, const float* T1, const float* T2, const float* bw,
                                       const float* yvec, const float* wb, const float* out, const float* dOut,
                                       const float* dfeat, const float* dS, int64_t n, int D, int Dout, int C, int V,
                                       float* result, float* workspace, int64_t workspace_floats, void* stream) {
  float* dWf = result;
  TAGREC_REQUIRE(T0 && T1 && T2 && bw && yvec && wb && out && dOut && dfeat && dS && dWf && workspace,
                 "tgcn_fuse_wf: null pointer");
  if (C != kBitC || V != kVecC) return fail(TAGREC_E_UNSUPPORTED, "tgcn_fuse: built for num_bit_conv 32, num_vec_conv 8");
  TAGREC_REQUIRE(workspace_floats >= tagrec_tgcn_fuse_wf_workspace(D, Dout), "tgcn_fuse_wf: workspace too small");
  TAGREC_REQUIRE(n >= 1, "tgcn_fuse_wf: empty input");
  hipStream_t s = static_cast<hipStream_t>(stream);
#define CALL(DD, OO) launch_fuse_wf<DD, OO>(T0, T1, T2, bw, yvec, wb, out, dOut, dfeat, dS, n, dWf, workspace, s)
  TAGREC_FUSE_DISPATCH(CALL);
#undef CALL
}
