// N1: evaluation scoring -> train-item mask -> top-K, fused (one pass over the item table per block of users).
//
// Replaces, per 512-user batch of `epoch_test` (/root/reference/training/basic_test.py:36-50):
//     rating = sigmoid(U_b I^T)        model/lightgcn.py:84-89 (predict_rating)
//     rating[train positives] = -1024  basic_test.py:42-47
//     _, top = torch.topk(rating, k)   basic_test.py:48
// The reference materialises the [512, n_item] rating matrix; here scores live in MFMA accumulators: a wave
// keeps 16 users' embeddings in registers (B-operand), streams 16-item tiles of the item table as A-operand
// (exact-fp32 MFMA 16x16x4, D/4 per tile), and keeps each user's running top-K in LDS.  A score is looked at
// again only if it beats the user's current K-th best (about K ln(n_item / K) times per user), and only then is
// the train-item mask consulted (binary search in the user's sorted train list), so the mask costs nothing per
// score.  Ranking uses the sigmoid value, as the reference does (fp32 sigmoid saturates for scores >~ 17, which
// creates ties there too); ties resolve to the lower item id.
#include <math.h>

#include "common.h"

namespace tagrec {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kEvalThreads = 256;                 // 4 waves x 16 users
constexpr int kEvalUsers = 64;
constexpr int kMaxTopK = 64;

template <int D>
__global__ __launch_bounds__(kEvalThreads) void eval_topk_kernel(const float* __restrict__ U, const float* __restrict__ I,
                                                                 int64_t n_item, const int64_t* __restrict__ users,
                                                                 int64_t n_users, const int64_t* __restrict__ train_ptr,
                                                                 const int32_t* __restrict__ train_items, int K,
                                                                 int64_t* __restrict__ top_idx, float* __restrict__ top_val) {
  constexpr int DS = D / 4;                        // the lane's quarter of an embedding row
  extern __shared__ float lds[];                   // [64][K] scores, then [64][K] item ids
  float* sh_sc = lds;
  int* sh_id = reinterpret_cast<int*>(lds + kEvalUsers * K);
  for (int i = threadIdx.x; i < kEvalUsers * K; i += kEvalThreads) { sh_sc[i] = -INFINITY; sh_id[i] = -1; }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int uslot = wave * 16 + r;
  const int64_t upos = static_cast<int64_t>(blockIdx.x) * kEvalUsers + uslot;
  const bool uok = upos < n_users;
  const int64_t user = uok ? users[upos] : 0;
  float ub[DS];
#pragma unroll
  for (int s = 0; s < DS; s += 4) {
    const float4 t = uok ? *reinterpret_cast<const float4*>(U + user * D + q * DS + s) : make_float4(0.f, 0.f, 0.f, 0.f);
    ub[s] = t.x; ub[s + 1] = t.y; ub[s + 2] = t.z; ub[s + 3] = t.w;
  }
  const int64_t tlo = uok ? train_ptr[user] : 0, thi = uok ? train_ptr[user + 1] : 0;
  // The four lanes q = 0..3 of a user slot share its list and take turns (wave_barrier below).  That intrinsic orders
  // execution, not memory, so the list is accessed through volatile pointers: every threshold read and every insertion
  // goes to LDS and sees what the previous lane wrote.
  volatile float* my_sc = sh_sc + uslot * K;
  volatile int* my_id = sh_id + uslot * K;
  for (int64_t item0 = 0; item0 < n_item; item0 += 16) {
    // A-operand: row m = r is item item0 + r; k-slot q covers features q*DS .. q*DS+DS-1 (same split as ub)
    const int64_t it = item0 + r;
    float a[DS];
#pragma unroll
    for (int s = 0; s < DS; s += 4) {
      const float4 t = it < n_item ? *reinterpret_cast<const float4*>(I + it * D + q * DS + s) : make_float4(0.f, 0.f, 0.f, 0.f);
      a[s] = t.x; a[s + 1] = t.y; a[s + 2] = t.z; a[s + 3] = t.w;
    }
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < DS; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], ub[s], acc, 0, 0, 0);
    // acc[v] = score of (user slot r, item item0 + 4 q + v)
    float sg[4];
    bool cand = false;
    const float thr = my_sc[K - 1];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      sg[v] = 1.0f / (1.0f + expf(-acc[v]));
      cand |= uok && (item0 + 4 * q + v < n_item) && sg[v] > thr;
    }
    if (__any(cand)) {
      // serialise the (rare) insertions: one k-slot and one register at a time, so a user's list has one writer
      for (int qq = 0; qq < 4; ++qq) {
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int64_t item = item0 + 4 * q + v;
          if (q == qq && uok && item < n_item && sg[v] > my_sc[K - 1]) {
            // train positives are masked out (basic_test.py:47): binary search in the user's sorted train list
            int64_t lo = tlo, hi = thi;
            while (lo < hi) {
              const int64_t mid = (lo + hi) >> 1;
              if (train_items[mid] < item) lo = mid + 1; else hi = mid;
            }
            if (!(lo < thi && train_items[lo] == item)) {
              int p = K - 1;
              while (p > 0 && my_sc[p - 1] < sg[v]) {
                const float ps = my_sc[p - 1];
                const int pi = my_id[p - 1];
                my_sc[p] = ps;
                my_id[p] = pi;
                --p;
              }
              my_sc[p] = sg[v];
              my_id[p] = static_cast<int>(item);
            }
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }
    }
  }
  __syncthreads();
  if (uok && q == 0) {
    for (int p = 0; p < K; ++p) {
      top_idx[upos * K + p] = my_id[p];
      if (top_val) top_val[upos * K + p] = my_sc[p];
    }
  }
}

}  // namespace tagrec

using namespace tagrec;

extern "C" int tagrec_eval_topk_f32(const float* U, const float* I, int64_t n_item, int D, const int64_t* users,
                                    int64_t n_users, const int64_t* train_ptr, const int32_t* train_items, int K,
                                    int64_t* top_idx, float* top_val, void* stream) {
  TAGREC_REQUIRE(U && I && users && train_ptr && top_idx, "eval_topk: null pointer");
  TAGREC_REQUIRE(n_item >= 1 && n_users >= 0 && K >= 1 && K <= kMaxTopK, "eval_topk: bad shape (1 <= K <= 64)");
  TAGREC_REQUIRE(n_item < (1ll << 31), "eval_topk: item ids must fit int32");
  TAGREC_REQUIRE(aligned16(U) && aligned16(I), "eval_topk: rows must be 16-byte aligned");
  if (n_users == 0) return TAGREC_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const unsigned blocks = static_cast<unsigned>((n_users + kEvalUsers - 1) / kEvalUsers);
  const size_t lds = static_cast<size_t>(kEvalUsers) * K * 8;
#define LAUNCH(DD) \
  eval_topk_kernel<DD><<<blocks, kEvalThreads, lds, s>>>(U, I, n_item, users, n_users, train_ptr, train_items, K, top_idx, top_val)
  switch (D) {
    case 16: LAUNCH(16); break;
    case 32: LAUNCH(32); break;
    case 64: LAUNCH(64); break;
    case 128: LAUNCH(128); break;
    case 192: LAUNCH(192); break;
    case 256: LAUNCH(256); break;
    case 384: LAUNCH(384); break;
    case 512: LAUNCH(512); break;
    default:
      return fail(TAGREC_E_UNSUPPORTED,
                  "eval_topk: embedding width must be 16, 32, 64, 128, 192, 256, 384 or 512 (got " + std::to_string(D) + ")");
  }
#undef LAUNCH
  TAGREC_LAUNCH_CHECK();
  return TAGREC_OK;
}
