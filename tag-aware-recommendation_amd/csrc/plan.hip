// K12: the row plan of a restricted TGCN training step.
//
// TGCN.loss() reads the top layer at the <= 3 B batch rows (/root/reference/model/tgcn.py:236-249); layer l is therefore
// needed on the rows layer l + 1 needs plus their k sampled neighbours under every relation (the fixed-width tables of
// tgcn.py:194-202).  The host used to build these row sets from ~40 torch kernels and three host reads per level
// (zeros / index_put / index_select / nonzero / unique); here a level is TWO launches and ONE read:
//   mark    : every descriptor -- "the rows of this list" or "the neighbours, under this table, of the rows of this list" --
//             sets byte flags of its node type (races write the same value);
//   compact : flags -> ascending row list + position map (int32 [n + 1], 1-based position at slot row + 1, 0 elsewhere: the
//             renumbering the compact tables of the step use; slot 0 absorbs the pad id of the neighbour tables) + counts,
//             by count / scan / scatter over 4096-flag chunks (order-preserving: the sort-free inversion of the attention
//             backward relies on ascending rows).
// The flags of the three node types live in one buffer, each segment padded to whole chunks, so that one compaction
// serves all three and a chunk never straddles two types.
#include "common.h"

namespace tagrec {
namespace {

constexpr int kChunk = 4096;        // flags per compaction block: 256 threads x 16 bytes
constexpr int kMaxDesc = 12;

struct PlanMark {
  const int32_t* idx;      // neighbour table [*, k] (ids 1-based, 0 = pad) or nullptr: mark the listed rows themselves
  const int64_t* rows;     // list (element i at rows[i * stride]) or nullptr: rows 0 .. n_rows - 1
  int64_t stride, n_rows, n_src, n_dst;
  int k;
  uint8_t* flags;          // the destination type's segment: [n_dst + 1]
};
struct PlanMarkArgs {
  PlanMark d[kMaxDesc];
  int64_t* bad;            // counts[3]: ids out of range
};

__global__ __launch_bounds__(256) void plan_mark_kernel(PlanMarkArgs a) {
  const PlanMark& d = a.d[blockIdx.y];
  const int64_t total = d.idx ? d.n_rows * d.k : d.n_rows;
  const int64_t step = static_cast<int64_t>(gridDim.x) * 256;
  for (int64_t e = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; e < total; e += step) {
    if (d.idx) {
      const int64_t i = e / d.k;
      const int s = static_cast<int>(e - i * d.k);
      const int64_t row = d.rows ? d.rows[i * d.stride] : i;
      if (row < 0 || row >= d.n_src) { atomicAdd(reinterpret_cast<unsigned long long*>(a.bad), 1ull); continue; }
      const int64_t id = d.idx[row * d.k + s];
      if (id < 0 || id > d.n_dst) { atomicAdd(reinterpret_cast<unsigned long long*>(a.bad), 1ull); continue; }
      d.flags[id] = 1;
    } else {
      const int64_t row = d.rows[e * d.stride];
      if (row < 0 || row >= d.n_dst) { atomicAdd(reinterpret_cast<unsigned long long*>(a.bad), 1ull); continue; }
      d.flags[row + 1] = 1;
    }
  }
}

struct PlanLayout {
  int64_t off[4];          // byte offset of each type's segment in the flag buffer (multiples of kChunk); off[3] = total
  int64_t row_base[3];     // where each type's rows start in rows_out
};

__device__ __forceinline__ int chunk_type(const PlanLayout& L, int64_t byte0) { return byte0 >= L.off[2] ? 2 : (byte0 >= L.off[1] ? 1 : 0); }

// the 16 flags of a thread with the pad slot (first byte of a segment) cleared
__device__ __forceinline__ uint4 load_flags(const uint8_t* flags, const PlanLayout& L, int64_t byte0, int t) {
  uint4 v = *reinterpret_cast<const uint4*>(flags + byte0);
  if (byte0 == L.off[t]) v.x &= ~0xFFu;
  return v;
}
__device__ __forceinline__ int count16(const uint4& v) { return __popc(v.x) + __popc(v.y) + __popc(v.z) + __popc(v.w); }

__global__ __launch_bounds__(256) void plan_count_kernel(const uint8_t* __restrict__ flags, PlanLayout L, int* __restrict__ chunk_count) {
  __shared__ int part[4];
  const int64_t byte0 = static_cast<int64_t>(blockIdx.x) * kChunk + threadIdx.x * 16;
  const int t = chunk_type(L, static_cast<int64_t>(blockIdx.x) * kChunk);
  int c = count16(load_flags(flags, L, byte0, t));
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) c += __shfl_xor(c, m);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) chunk_count[blockIdx.x] = (part[0] + part[1]) + (part[2] + part[3]);
}

// one block: exclusive prefix of the chunk counts inside each type's segment, and the three totals
__global__ __launch_bounds__(1024) void plan_scan_kernel(const int* __restrict__ chunk_count, PlanLayout L, int* __restrict__ chunk_off,
                                                         int64_t* __restrict__ counts) {
  __shared__ int wave_sum[16];
  __shared__ int carry_s;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int t = 0; t < 3; ++t) {
    const int64_t c0 = L.off[t] / kChunk, c1 = L.off[t + 1] / kChunk;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int64_t base = c0; base < c1; base += 1024) {
      const int64_t i = base + threadIdx.x;
      const int v = i < c1 ? chunk_count[i] : 0;
      int x = v;
#pragma unroll
      for (int m = 1; m < 64; m <<= 1) {
        const int y = __shfl_up(x, m);
        if (lane >= m) x += y;
      }
      if (lane == 63) wave_sum[wave] = x;
      __syncthreads();
      int before = 0;
      for (int w = 0; w < wave; ++w) before += wave_sum[w];
      const int carry = carry_s;
      if (i < c1) chunk_off[i] = carry + before + x - v;
      __syncthreads();
      if (threadIdx.x == 1023) carry_s = carry + before + x;
      __syncthreads();
    }
    if (threadIdx.x == 0) counts[t] = carry_s;
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void plan_scatter_kernel(const uint8_t* __restrict__ flags, PlanLayout L, const int* __restrict__ chunk_off,
                                                            int64_t* __restrict__ rows_out, int32_t* __restrict__ pos_out) {
  __shared__ int part[4];
  const int64_t chunk0 = static_cast<int64_t>(blockIdx.x) * kChunk;
  const int64_t byte0 = chunk0 + threadIdx.x * 16;
  const int t = chunk_type(L, chunk0);
  const uint4 v = load_flags(flags, L, byte0, t);
  const int c = count16(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int x = c;
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) {
    const int y = __shfl_up(x, m);
    if (lane >= m) x += y;
  }
  if (lane == 63) part[wave] = x;
  __syncthreads();
  int before = 0;
  for (int w = 0; w < wave; ++w) before += part[w];
  int pos = chunk_off[blockIdx.x] + before + x - c;          // exclusive position of this thread's first flagged row in its type
  int64_t* rows_t = rows_out + L.row_base[t];
  const int64_t local0 = byte0 - L.off[t];                   // slot = row + 1
  const unsigned w4[4] = {v.x, v.y, v.z, v.w};
  int p[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const bool on = (w4[j >> 2] >> (8 * (j & 3))) & 0xFFu;
    p[j] = on ? pos + 1 : 0;
    if (on) { rows_t[pos] = local0 + j - 1; ++pos; }
  }
  int4* dst = reinterpret_cast<int4*>(pos_out + byte0);
#pragma unroll
  for (int j = 0; j < 4; ++j) dst[j] = make_int4(p[4 * j], p[4 * j + 1], p[4 * j + 2], p[4 * j + 3]);
}

// out[i] = pos[rows[i * stride] + 1] - 1: a row id -> its position in a compact table (-1: not there)
__global__ __launch_bounds__(256) void plan_lookup_kernel(const int32_t* __restrict__ pos, const int64_t* __restrict__ rows, int64_t stride,
                                                           int64_t n, int64_t* __restrict__ out, int64_t out_stride) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (i < n) out[i * out_stride] = static_cast<int64_t>(pos[rows[i * stride] + 1]) - 1;
}

int64_t pad_chunk(int64_t n) { return (n + kChunk - 1) / kChunk * kChunk; }

PlanLayout layout_of(const int64_t* sizes3) {
  PlanLayout L;
  L.off[0] = 0;
  L.row_base[0] = 0;
  for (int t = 0; t < 3; ++t) {
    L.off[t + 1] = L.off[t] + pad_chunk(sizes3[t] + 1);
    if (t < 2) L.row_base[t + 1] = L.row_base[t] + sizes3[t];
  }
  return L;
}

}  // namespace
}  // namespace tagrec

using namespace tagrec;

extern "C" {

int64_t tagrec_plan_flags_workspace(const int64_t* sizes3) {
  if (!sizes3) return 0;
  return layout_of(sizes3).off[3];
}

int64_t tagrec_plan_segment_result(const int64_t* sizes3, int type) {
  if (!sizes3 || type < 0 || type > 3) return -1;
  return layout_of(sizes3).off[type];
}

int64_t tagrec_plan_scan_workspace(const int64_t* sizes3) {
  if (!sizes3) return 0;
  return 2 * (layout_of(sizes3).off[3] / kChunk);
}

int tagrec_plan_mark_u8(int n_desc, const int32_t* const* idx, const int64_t* const* rows, const int64_t* stride, const int64_t* n_rows,
                        const int64_t* n_src, const int* k, const int* dst_type, const int* all3, const int64_t* sizes3,
                        uint8_t* flags, int64_t* counts, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  TAGREC_REQUIRE(n_desc >= 0 && n_desc <= kMaxDesc, "plan_mark: at most 12 descriptors");
  TAGREC_REQUIRE(sizes3 && flags && counts && all3, "plan_mark: null pointer");
  TAGREC_REQUIRE(aligned16(flags), "plan_mark: flags must be 16-byte aligned");
  const PlanLayout L = layout_of(sizes3);
  TAGREC_HIP(hipMemsetAsync(flags, 0, static_cast<size_t>(L.off[3]), stream));
  TAGREC_HIP(hipMemsetAsync(counts, 0, 4 * sizeof(int64_t), stream));
  for (int t = 0; t < 3; ++t)
    if (all3[t]) TAGREC_HIP(hipMemsetAsync(flags + L.off[t], 1, static_cast<size_t>(sizes3[t] + 1), stream));
  PlanMarkArgs a;
  a.bad = counts + 3;
  int64_t most = 0;
  int live = 0;
  for (int i = 0; i < n_desc; ++i) {
    const int t = dst_type[i];
    TAGREC_REQUIRE(t >= 0 && t < 3, "plan_mark: destination type must be 0, 1 or 2");
    TAGREC_REQUIRE(n_rows[i] >= 0 && stride[i] >= 1, "plan_mark: negative row count or stride < 1");
    TAGREC_REQUIRE(idx[i] || rows[i] || n_rows[i] == 0, "plan_mark: a descriptor without a table needs a row list");
    TAGREC_REQUIRE(!idx[i] || k[i] >= 1, "plan_mark: table width must be positive");
    if (n_rows[i] == 0 || all3[t]) continue;               // nothing to add / the type is already complete
    PlanMark& d = a.d[live++];
    d.idx = idx[i]; d.rows = rows[i]; d.stride = stride[i]; d.n_rows = n_rows[i]; d.n_src = n_src[i]; d.n_dst = sizes3[t];
    d.k = idx[i] ? k[i] : 0;
    d.flags = flags + L.off[t];
    const int64_t total = idx[i] ? n_rows[i] * k[i] : n_rows[i];
    if (total > most) most = total;
  }
  if (live == 0) return TAGREC_OK;
  int64_t bx = (most + 255) / 256;
  if (bx > 8192) bx = 8192;
  plan_mark_kernel<<<dim3(static_cast<unsigned>(bx), static_cast<unsigned>(live)), 256, 0, stream>>>(a);
  TAGREC_LAUNCH_CHECK();
  return TAGREC_OK;
}

int tagrec_plan_compact_i64(const uint8_t* flags, const int64_t* sizes3, int64_t* rows_out, int32_t* pos_out, int64_t* counts,
                            int32_t* workspace, int64_t workspace_ints, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  TAGREC_REQUIRE(flags && sizes3 && rows_out && pos_out && counts && workspace, "plan_compact: null pointer");
  TAGREC_REQUIRE(aligned16(flags) && aligned16(pos_out), "plan_compact: flags / pos_out must be 16-byte aligned");
  const PlanLayout L = layout_of(sizes3);
  const int64_t chunks = L.off[3] / kChunk;
  TAGREC_REQUIRE(workspace_ints >= 2 * chunks, "plan_compact: workspace too small (tagrec_plan_scan_workspace)");
  TAGREC_REQUIRE(chunks < (int64_t{1} << 31), "plan_compact: too many nodes");
  plan_count_kernel<<<static_cast<unsigned>(chunks), 256, 0, stream>>>(flags, L, workspace);
  TAGREC_LAUNCH_CHECK();
  plan_scan_kernel<<<1, 1024, 0, stream>>>(workspace, L, workspace + chunks, counts);
  TAGREC_LAUNCH_CHECK();
  plan_scatter_kernel<<<static_cast<unsigned>(chunks), 256, 0, stream>>>(flags, L, workspace + chunks, rows_out, pos_out);
  TAGREC_LAUNCH_CHECK();
  return TAGREC_OK;
}

int tagrec_plan_lookup_i64(const int32_t* pos, const int64_t* rows, int64_t stride, int64_t n, int64_t* out, int64_t out_stride,
                           void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (n == 0) return TAGREC_OK;
  TAGREC_REQUIRE(pos && rows && out, "plan_lookup: null pointer");
  TAGREC_REQUIRE(n > 0 && stride >= 1 && out_stride >= 1, "plan_lookup: bad sizes");
  plan_lookup_kernel<<<static_cast<unsigned>((n + 255) / 256), 256, 0, stream>>>(pos, rows, stride, n, out, out_stride);
  TAGREC_LAUNCH_CHECK();
  return TAGREC_OK;
}

}  // extern "C"
