/* tagrec.h -- C ABI of libtagrec_hip.so (MI355X / gfx950 only).
 *
 * The reference (chenzheng5555/tag-aware-recommendation) has no FFI: its hot path
 * is in-process Python calling PyTorch operators.  Each entry point below replaces
 * the PyTorch operator sequence at the cited reference lines; the Python host in
 * `tag-aware-recommendation_amd/` mirrors the reference's operator / model / step
 * surface and binds these symbols with ctypes (binding shown in INTEGRATION.md).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer into caller-owned memory (PyTorch storage);
 *     the library never frees caller memory.  Row-major, contiguous, fp32.
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream);
 *     every call is asynchronous on that stream unless stated otherwise.
 *   - return value: 0 = TAGREC_OK, negative = TAGREC_E_*; the message for the last
 *     failing call of the calling thread is tagrec_last_error().  No C++ exception
 *     crosses the ABI.
 *   - a handle may be used from one thread AND one stream at a time (a graph handle owns the
 *     partial-sum scratch of the launch in flight; launches on one stream are ordered, two
 *     streams sharing a handle would race on it); distinct handles are independent.
 *   - no entry point that takes a `stream` allocates or frees device memory: scratch is
 *     either allocated when a handle is created or passed in by the caller (the
 *     `*_workspace` size queries), so every such call may be captured in a HIP graph.
 *   - D (row width) must be a multiple of 4 and rows 16-byte aligned for the vector
 *     kernels; other widths take a scalar kernel (correct, slower).
 */
#ifndef TAGREC_H
#define TAGREC_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TAGREC_OK 0
#define TAGREC_E_INVALID (-1)     /* bad argument (null pointer, negative size, bad enum) */
#define TAGREC_E_HIP (-2)         /* a HIP runtime call failed; message holds hipGetErrorString */
#define TAGREC_E_UNSUPPORTED (-3) /* shape outside what the kernels cover */
#define TAGREC_E_NOMEM (-4)

#define TAGREC_LOSS_SOFTPLUS 0    /* mean softplus(neg - pos)        (loss.py:11) */
#define TAGREC_LOSS_LOGSIGMOID 1  /* -mean logsigmoid(pos - neg)     (loss.py:9)  */

#define TAGREC_ABI_VERSION 2

typedef struct tagrec_graph tagrec_graph;

int tagrec_abi_version(void);
const char* tagrec_last_error(void);
/* number of CUs / wavefront size / gcnArchName of the current device (synchronous) */
int tagrec_device_info(int* n_cu, int* wave_size, char* arch, int arch_len);

/* ---- sparse normalised adjacency ------------------------------------------------
 * A CSR matrix (int64 rowptr[n_rows+1], int32 colidx[nnz], fp32 vals[nnz]) as built by
 * model/help/adj.py:38-46 `creat_adj` (+ :144-150 `sp2tensor`).  The arrays are BORROWED:
 * they must outlive the handle.  Creation scans the row lengths once (synchronises the
 * stream) and splits rows longer than 1024 entries into 512-entry chunks that are
 * reduced in a fixed order, so results do not depend on scheduling.  The partial-sum
 * scratch of those chunks (256 floats per chunk) is allocated here, once. */
int tagrec_graph_create(tagrec_graph** out, int64_t n_rows, int64_t n_cols, int64_t nnz,
                        const int64_t* rowptr, const int32_t* colidx, const float* vals, void* stream);
/* A second matrix over the SAME row pointer (same rows, same entries per row) with its own column indices / values,
 * e.g. the two inverted neighbour tables of the TGCN attention backward: shares `like`'s long-row work list instead of
 * scanning again (no synchronisation).  `like` must outlive the new handle. */
int tagrec_graph_create_like(tagrec_graph** out, const tagrec_graph* like, int64_t n_cols, const int32_t* colidx,
                             const float* vals);
/* The same two constructors on a CALLER-PROVIDED workspace of tagrec_graph_workspace(nnz) bytes (256-byte aligned, must
 * outlive the handle): nothing is allocated or freed on the device -- for matrices built and dropped inside a training
 * step (the inverted neighbour tables of the TGCN attention backward, tgcn.py:20-37).  create_ws still reads two counters
 * back (one stream synchronisation); create_like_ws is asynchronous. */
int64_t tagrec_graph_workspace(int64_t nnz);
int tagrec_graph_create_ws(tagrec_graph** out, int64_t n_rows, int64_t n_cols, int64_t nnz, const int64_t* rowptr,
                           const int32_t* colidx, const float* vals, void* workspace, int64_t workspace_bytes, void* stream);
/* The same, without the host read of the long-row counters (a read waits for the stream: the TGCN step builds twelve
 * inverted tables per step).  The work list is sized by its upper bounds, unused slots are marked and skipped by the
 * kernels; accepted by the tagrec_spmm_*, tagrec_attn_pull_* entry points, refused by the tagrec_route_* ones. */
int tagrec_graph_create_ws_deferred(tagrec_graph** out, int64_t n_rows, int64_t n_cols, int64_t nnz, const int64_t* rowptr,
                                    const int32_t* colidx, const float* vals, void* workspace, int64_t workspace_bytes,
                                    void* stream);
int tagrec_graph_create_like_ws(tagrec_graph** out, const tagrec_graph* like, int64_t n_cols, const int32_t* colidx,
                                const float* vals, void* workspace, int64_t workspace_bytes);
int tagrec_graph_destroy(tagrec_graph* g);
int tagrec_graph_info(const tagrec_graph* g, int64_t* n_rows, int64_t* n_cols, int64_t* nnz,
                      int64_t* n_long_rows, int64_t* n_chunks);

/* Y[n_rows,D] = A @ X[n_cols,D]            -- `split_mm` / torch.sparse.mm, adj.py:158-167.
 * The autograd backward dX = A^T dY is the same call on the transposed graph
 * (bi_norm adjacency is symmetric: the same handle). */
int tagrec_spmm_f32(const tagrec_graph* g, const float* X, float* Y, int D, void* stream);

/* One LightGCN layer, fused (lightgcn.py:54-58 + the mean of :60):
 *   Y_raw = A @ X;  inv_norm[r] = 1 / max(||Y_raw[r]||_2, 1e-12);
 *   acc[r] += acc_scale * Y_raw[r] / max(||Y_raw[r]||_2, 1e-12)
 * Y_raw feeds the next layer; acc is the running layer mean. */
int tagrec_spmm_norm_acc_f32(const tagrec_graph* g, const float* X, float* Y_raw, float* inv_norm,
                             float* acc, float acc_scale, int D, void* stream);

/* Backward of one layer, fused:  G_out = A @ G_in + normalize_bwd(X_raw, inv_norm, d_scale * dZ)
 * where normalize_bwd is the gradient of F.normalize(p=2, eps=1e-12) (lightgcn.py:57) at X_raw. */
int tagrec_spmm_normbwd_f32(const tagrec_graph* g, const float* G_in, const float* X_raw,
                            const float* inv_norm, const float* dZ, float d_scale, float* G_out,
                            int D, void* stream);

/* G_out = A @ G_in + b_scale * B      (last backward hop into the ego table; NGCF dX = dX_direct + A^T dN) */
int tagrec_spmm_axpy_f32(const tagrec_graph* g, const float* G_in, const float* B, float b_scale,
                         float* G_out, int D, void* stream);

/* The forward / backward LightGCN layers with MESSAGE DROPOUT (F.dropout on the product, model/lightgcn.py:56) inside
 * the epilogue: Y_raw = mask * (A @ X) / (1 - p) before it is normalised and accumulated; in the backward layer the
 * same mask multiplies the gradient that leaves the layer.  Whether element i survives is a pure function of
 * (seed, i) (counter-based generator, csrc/common.h), so nothing is stored between the passes; the caller uses one
 * seed per layer and step.  tagrec_dropout_f32 applies the identical mask to a plain [n] buffer (n % 4 == 0): the
 * last layer's normalize-backward, NGCF / TGCN layer outputs, tests.  Training-mode parity with the reference is
 * statistical (it draws torch's generator). */
int tagrec_spmm_norm_acc_drop_f32(const tagrec_graph* g, const float* X, float* Y_raw, float* inv_norm,
                                  float* acc, float acc_scale, float drop_p, uint64_t seed, int D, void* stream);
int tagrec_spmm_normbwd_drop_f32(const tagrec_graph* g, const float* G_in, const float* X_raw,
                                 const float* inv_norm, const float* dZ, float d_scale, float drop_p,
                                 uint64_t seed, float* G_out, int D, void* stream);
int tagrec_dropout_f32(const float* x, float* out, int64_t n, float p, uint64_t seed, void* stream);
/* The same mask on a compact row list: element (j, c) of x [n_rows, D] takes the draw of element (rows[j], c) of the full
 * [N, D] layer output, so a node named by several list slots gets ONE mask (model/ngcf.py:85 drops the full output). */
int tagrec_dropout_rows_f32(const float* x, float* out, const int64_t* rows, int64_t n_rows, int D, float p, uint64_t seed,
                            void* stream);

/* Backward layers on a ROW-SPARSE gradient.  The gradient that enters the backward chain is non-zero on the <= 3 B rows
 * of the batch only, and one hop later on their neighbours, so most rows a backward product would gather are zero.
 * row_flags[c] (uint8) != 0 iff row c holds a non-zero; *count (uint32, device) = how many rows are flagged.  A product
 * given in_flags / in_count does not fetch rows flagged zero -- a x 0 adds exactly 0 -- and consults the flags only
 * while they cover less than 4/5 of the rows (decided on the device, no host read).  in_flags with in_count == NULL:
 * the flags are ALWAYS consulted, so rows flagged zero may hold anything (an operand written on a row subset only).
 *   rownorm_bwd_flags   : tagrec_rownorm_bwd_f32 + flags / count of its output rows (the head of the chain)
 *   spmm_normbwd_sparse : tagrec_spmm_normbwd_drop_f32 reading in_flags (may be NULL) and writing out_flags / out_count
 *                         (may be NULL).  row_mask (may be NULL): rows whose byte is 0 are left alone -- for a caller
 *                         who knows their result is zero (no flagged neighbour, and X_raw / inv_norm / dZ zero there, as
 *                         after the row-restricted forward) and has zeroed those rows of G_out / out_flags itself
 *   spmm_axpy_sparse    : tagrec_spmm_axpy_f32 reading in_flags (may be NULL)
 * dz_flags / b_flags (may be NULL): dz_flags[r] == 0 promises that row r of dZ (of B) is zero; the row's epilogue term
 * then vanishes and neither dZ nor X_raw is read for it.  The gradient of a BPR batch w.r.t. the layer mean is non-zero
 * on the <= 3 B batch rows only, so every backward layer skips 2 N D floats of epilogue reads.
 * A product given in_flags walks a row's entries 64 at a time, keeps the entries whose operand row is flagged (order
 * preserved) and gathers those alone. */
int tagrec_rownorm_bwd_flags_f32(const float* X_raw, const float* inv_norm, const float* dZ, int64_t lddz,
                                 float d_scale, float* dX, int accumulate, int64_t n_rows, int D,
                                 uint8_t* row_flags, unsigned* count, void* stream);
int tagrec_spmm_normbwd_sparse_f32(const tagrec_graph* g, const float* G_in, const uint8_t* in_flags,
                                   const unsigned* in_count, const float* X_raw, const float* inv_norm,
                                   const float* dZ, float d_scale, float drop_p, uint64_t seed, float* G_out,
                                   uint8_t* out_flags, unsigned* out_count, const uint8_t* row_mask,
                                   const uint8_t* dz_flags, int D, void* stream);
int tagrec_spmm_axpy_sparse_f32(const tagrec_graph* g, const float* G_in, const uint8_t* in_flags /* may be NULL */,
                                const unsigned* in_count, const float* B, float b_scale, float* G_out,
                                const uint8_t* row_mask /* may be NULL; as in spmm_normbwd_sparse */,
                                const uint8_t* b_flags, int D, void* stream);
/* G_out = A @ G_in on a row-sparse operand (in_flags / in_count as above; row_mask optional), writing the row flags of
 * the result (out_flags; out_count optional).  The backward hop of a column-sharded table, whose normalize-backward term
 * needs row dots over every rank's columns and is added on the batch rows by the caller. */
int tagrec_spmm_flags_f32(const tagrec_graph* g, const float* G_in, const uint8_t* in_flags, const unsigned* in_count,
                          float* G_out, uint8_t* out_flags, unsigned* out_count, const uint8_t* row_mask, int D, void* stream);
/* Forward layer on a SUBSET of the output rows.  The loss reads the propagated table at the batch rows only
 * (model/lightgcn.py:71-75), so the last layer is needed on those rows and the layer below it on their neighbours.
 *   graph_mark_rows   : flags[c] = 1 for every column index stored in the listed rows and for the rows themselves
 *                       (flags uint8 [n_rows], zeroed / pre-marked by the caller; square adjacency)
 *   spmm_norm_acc_rows: tagrec_spmm_norm_acc_drop_f32 for the rows with row_mask[r] != 0 (NULL = every row); the other
 *                       rows of Y_raw, inv_norm and acc are left untouched.  acc may be NULL: the layer mean is then
 *                       not accumulated (a caller that reads it at a few rows forms it there from Y_raw and inv_norm) */
int tagrec_graph_mark_rows_u8(const tagrec_graph* g, const int64_t* rows, int64_t n_listed, uint8_t* flags, void* stream);
/* the same for a rectangular matrix: flags (uint8 [n_cols]) of the columns stored in the listed rows, rows not marked */
int tagrec_graph_mark_cols_u8(const tagrec_graph* g, const int64_t* rows, int64_t n_listed, uint8_t* flags, void* stream);
int tagrec_spmm_norm_acc_rows_f32(const tagrec_graph* g, const float* X, float* Y_raw, float* inv_norm,
                                  float* acc, float acc_scale, const uint8_t* row_mask, float drop_p,
                                  uint64_t seed, int D, void* stream);
/* the plain product (NGCF) and the column-sharded forward product on the rows with row_mask[r] != 0 */
int tagrec_spmm_rows_f32(const tagrec_graph* g, const float* X, float* Y, const uint8_t* row_mask, int D, void* stream);
int tagrec_spmm_ss_rows_f32(const tagrec_graph* g, const float* X, float* Y, float* ss, const uint8_t* row_mask, int D,
                            void* stream);
/* The plain product on a short LIST of rows with a compact result: Y[k, :] = (A @ X)[rows[k], :], k < n_listed
 * (rows int64, may repeat; rows[k] is NOT range-checked on the device -- the caller guarantees 0 <= rows[k] < n_rows).
 * Every listed row is cut into 32 ranges summed by separate blocks (batch rows are popular items: 1e5 entries), partial
 * rows go to `ws` (tagrec_spmm_listed_workspace(n_listed, D) floats, caller-provided) and are added in a fixed order.
 * This is the top layer of a training step -- the loss reads it at the <= 3 B batch rows (model/lightgcn.py:71-75) -- and,
 * with A = the column slice A[:, rows_g] of a row-sharded table, rank g's share of it (the 1-D fold partition of
 * adj.py:114-140,158-164 in push form where the output is 3 B rows); the shares are summed by an all-reduce of [3 B, D]. */
int64_t tagrec_spmm_listed_workspace(int64_t n_listed, int D);
int tagrec_spmm_listed_f32(const tagrec_graph* g, const int64_t* rows, int64_t n_listed, const float* X, float* Y,
                           int D, float* ws, int64_t ws_floats, void* stream);
/* tagrec_spmm_normbwd_dot_f32 (column-sharded tables) on a row-sparse G_in */
int tagrec_spmm_normbwd_dot_sparse_f32(const tagrec_graph* g, const float* G_in, const uint8_t* in_flags,
                                       const unsigned* in_count, const float* X_raw, const float* inv_norm,
                                       const float* dZ, const float* dot, float d_scale, float* G_out,
                                       uint8_t* out_flags, unsigned* out_count, const uint8_t* row_mask, int D,
                                       void* stream);
/* row_flags[r] = (row r of X [n_rows, D] holds a non-zero), *count = number of such rows: for operands whose producer
 * does not write flags itself (the NGCF backward's dN). */
int tagrec_row_flags_f32(const float* X, int64_t n_rows, int D, uint8_t* row_flags, unsigned* count, void* stream);

/* ---- column-sharded tables (feature sharding over GPUs: every rank holds D/G columns of every row) ---------------
 * Whatever reduces over a row's columns is split into a local part, an all-reduce done by the caller, and an apply:
 *   spmm_ss          : Y = A @ X;  ss[r] = sum_c Y[r,c]^2 over the local columns
 *   row_scale_acc    : acc[r,:] += s * inv[r] * Y[r,:]                          (inv from the all-reduced ss)
 *   row_dot          : out[r] = inv[r] * s * sum_c X[r,c] dZ[r,c]               (local part of z . (s dZ))
 *   rownorm_bwd_dot  : dX[r,:] = inv[r] * (s dZ[r,:] - X[r,:] inv[r] dot[r])    (dot all-reduced)
 *   spmm_normbwd_dot : G_out = A @ G_in + the same expression                   (fused backward layer)
 *   bpr_dots         : dots[b] = (u.p, u.n, 0.5(|u|^2+|p|^2+|n|^2)) over the local columns */
int tagrec_spmm_ss_f32(const tagrec_graph* g, const float* X, float* Y, float* ss, int D, void* stream);
int tagrec_spmm_normbwd_dot_f32(const tagrec_graph* g, const float* G_in, const float* X_raw, const float* inv_norm,
                                const float* dZ, const float* dot, float d_scale, float* G_out, int D, void* stream);
int tagrec_row_scale_acc_f32(const float* Y, const float* inv, float s, float* acc, int64_t n_rows, int D, void* stream);
int tagrec_row_dot_f32(const float* X, const float* inv, const float* dZ, float s, float* out, int64_t n_rows, int D,
                       void* stream);
int tagrec_rownorm_bwd_dot_f32(const float* X, const float* inv, const float* dZ, const float* dot, float s, float* dX,
                               int64_t n_rows, int D, void* stream);
int tagrec_bpr_dots_f32(const float* U, const float* I, int64_t ld, int D, const float* Ureg, const float* Ireg,
                        int64_t ldreg, int Dreg, const int64_t* trip, int64_t B, float* dots, void* stream);

/* ---- F.normalize(p=2, dim=1, eps=1e-12) as standalone kernels (ngcf.py:86, tgcn.py:220-222) ----
 * fwd: Z (row stride ldz floats, so it can write one slot of a concat buffer) and inv_norm.
 * bwd: dX = normalize_bwd(X_raw, inv_norm, d_scale * dZ); dZ has row stride lddz; if `accumulate`
 *      the result is added to dX. */
int tagrec_rownorm_fwd_f32(const float* X, float* Z, int64_t ldz, float* inv_norm, int64_t n_rows, int D, void* stream);
int tagrec_rownorm_bwd_f32(const float* X_raw, const float* inv_norm, const float* dZ, int64_t lddz,
                           float d_scale, float* dX, int accumulate, int64_t n_rows, int D, void* stream);

/* ---- BPR triplet loss (lightgcn.py:68-82, ngcf.py:95-105, loss.py:4-12, 27-32) -------------------
 * trip: int64 [B,3] = (user, pos_item, neg_item), item ids local to the item table.
 * U/I: propagated user/item tables (row stride ld floats, width D); Ureg/Ireg (stride ldreg, width Dreg):
 * the rows the L2 term reads (ego tables for LightGCN, the propagated ones for NGCF/TGCN).
 * fwd writes loss_out[0] = mul_loss, loss_out[1] = l2reg_loss (unweighted), and coef[b] =
 * d mul_loss_b / d(neg_b - pos_b) (a sigmoid) for the backward.  `partials` is scratch of
 * 2*ceil(B/256) floats; the two-stage reduction is deterministic. */
int tagrec_bpr_fwd_f32(const float* U, const float* I, int64_t ld, int D,
                       const float* Ureg, const float* Ireg, int64_t ldreg, int Dreg,
                       const int64_t* trip, int64_t B, int loss_kind,
                       float* coef, float* partials, float* loss_out, void* stream);
/* bwd scatter-adds (float atomics) into dU/dI (stride ld) and dUreg/dIreg (stride ldreg):
 *   dU[u] += g0*coef/B * (I[n]-I[p]);  dI[p] -= g0*coef/B * U[u];  dI[n] += g0*coef/B * U[u]
 *   dXreg[row] += g1 * reg / B * Xreg[row]     for the three rows of every triplet
 * g = device pointer to two floats (upstream gradients of the two loss parts) or NULL for (1,1).
 * Callers zero the gradient buffers.  dU = dI = NULL runs the L2 part only (so it can be added
 * after the propagation backward has overwritten the ego gradient). */
int tagrec_bpr_bwd_f32(const float* U, const float* I, int64_t ld, int D,
                       const float* Ureg, const float* Ireg, int64_t ldreg, int Dreg,
                       const int64_t* trip, int64_t B, const float* coef, const float* g, float reg,
                       float* dU, float* dI, float* dUreg, float* dIreg, void* stream);

/* ---- N4: propagation with dynamic per-factor edge weights (DGCF model/dgcf.py:70-110, DisenGCN model/disengcn.py:23-46) --
 * The graph handle supplies the STRUCTURE (rowptr / colidx; its values are not read).  K factors own the column
 * slices [k D/K, (k+1) D/K) of an embedding row (torch.split / torch.cat along dim 1 in the reference).  Per-entry
 * data is factor-interleaved [nnz, K] in CSR entry order; per-node data [N, K].  D in {32,64,128,256}, K in {1,2,4,8},
 * D/K a multiple of 4.
 *   route_softmax      : w[j, :] = softmax(logits[j, :])                      (torch.softmax(A_values, dim=0))
 *   route_rowsum_rsqrt : d[r, k] = 1 / sqrt(sum_j w[j, k] over row r), inf -> 0  (dgcf.py:95-99)
 *   route_permute      : wt[j, :] = w[perm[j], :]                              (weights in the transposed matrix's order)
 *   route_spmm         : y[r, slice k] = post[r,k] * sum_j w[j,k] X[col_j, slice k] + self[r] + b_scale * B[r];
 *                        Y = y, Yn = y / max(||y slice||, 1e-12), inv[r,k] = 1 / max(||y slice||, 1e-12)
 *                        (post / self / B / Y / Yn / inv may each be NULL)
 *   route_score        : logits[j, k] (+)= < H[row_j, slice k], T[col_j, slice k] >
 *   slice_scale        : Y[r, slice k] = scale[r, k] * X[r, slice k]
 *   slice_norm_fwd     : Y = X / max(||X slice||, 1e-12) per slice (then tanh if apply_tanh); inv (may be NULL) as above
 *   slice_norm_bwd     : gradient of slice_norm_fwd (without tanh) given X_raw, inv and dZ
 *   row_softmax_fwd    : a[j] = softmax of logits over the stored entries of j's row (torch.sparse.softmax(adj, dim=1),
 *                        model/kgat.py:96);  row_softmax_bwd : dlogits = a * (da - sum_row(a * da))
 * D in {16,32,64,128,256} for the routed kernels. */
int tagrec_route_softmax_f32(const float* logits, float* w, int64_t nnz, int K, void* stream);
int tagrec_route_rowsum_rsqrt_f32(const tagrec_graph* g, const float* w, int K, float* d, void* stream);
int tagrec_route_permute_f32(const float* w, const int32_t* perm, float* wt, int64_t nnz, int K, void* stream);
int tagrec_route_spmm_f32(const tagrec_graph* g, const float* W, int K, const float* X, const float* post,
                          const float* self, const float* B, float b_scale, float* Y, float* Yn, float* inv,
                          int D, void* stream);
int tagrec_route_score_f32(const tagrec_graph* g, const float* H, const float* T, float* logits, int K,
                           int accumulate, int D, void* stream);
/* The same two with the restrictions of the plain products: row_mask (may be NULL) = output rows / entries' rows to
 * compute, the others are left untouched; in_flags / in_count (may be NULL) = rows of X that hold a non-zero. */
int tagrec_route_spmm_ex_f32(const tagrec_graph* g, const float* W, int K, const float* X, const float* post,
                             const float* self, const float* B, float b_scale, float* Y, float* Yn, float* inv,
                             const uint8_t* row_mask, const uint8_t* in_flags, const unsigned* in_count, int D,
                             void* stream);
int tagrec_route_score_rows_f32(const tagrec_graph* g, const float* H, const float* T, float* logits, int K,
                                int accumulate, const uint8_t* row_mask, int D, void* stream);
int tagrec_slice_scale_f32(const float* X, const float* scale, float* Y, int64_t n_rows, int D, int K, void* stream);
int tagrec_slice_norm_fwd_f32(const float* X, float* Y, float* inv, int64_t n_rows, int D, int K, int apply_tanh,
                              void* stream);
int tagrec_slice_norm_bwd_f32(const float* X_raw, const float* inv, const float* dZ, float* dX, int64_t n_rows, int D,
                              int K, void* stream);
int tagrec_row_softmax_fwd_f32(const tagrec_graph* g, const float* logits, float* a, void* stream);
int tagrec_row_softmax_bwd_f32(const tagrec_graph* g, const float* a, const float* da, float* dlogits, void* stream);

/* ---- evaluation: sigmoid(U_b I^T) -> mask train positives -> top-K, fused (lightgcn.py:84-89, basic_test.py:36-50) --
 * U / I: propagated user / item tables, row-major [*, D].  users: int64 [n_users] ids to score.  train_ptr int64
 * [n_user_total + 1] / train_items int32: each user's train items, sorted (the rows basic_test.py:47 overwrites with
 * -1024; here they are simply never admitted).  top_idx int64 [n_users, K]: item ids by descending sigmoid score,
 * ties to the lower id, -1 if fewer than K items remain; top_val [n_users, K] (may be NULL): the scores.
 * D in {16,32,64,128,192,256,384,512}, K <= 64. */
int tagrec_eval_topk_f32(const float* U, const float* I, int64_t n_item, int D, const int64_t* users,
                         int64_t n_users, const int64_t* train_ptr, const int32_t* train_items, int K,
                         int64_t* top_idx, float* top_val, void* stream);

/* ---- negative sampler (train_data/utils.py:19-28, 31-40) -------------------------------------------------------
 * For each of n_rows positive rows with left id left[e] (a user, or a (user, tag) pair id): one uniform draw in
 * [0, n_right), re-drawn while it is in the left id's sorted positive list cols[rowptr[l] .. rowptr[l+1]).
 * Draw t of row e is a pure function of (seed, e, t): reproducible whatever the launch geometry. */
int tagrec_sample_negative_i64(const int64_t* left, int64_t n_rows, const int64_t* rowptr, const int32_t* cols,
                               int64_t n_left, int64_t n_right, uint64_t seed, int64_t* neg, void* stream);

/* ---- TransTag phase of TGCN (tgcn.py:251-261, loss.py:35-41, 27-32) on the EGO tables -------------------------
 * quad: int64 [B,4] = (user, tag, pos_item, neg_item).  fwd: loss_out[0] = mean relu(margin + ||u+t-p|| - ||u+t-n||),
 * loss_out[1] = l2reg_loss(u, t, p, n) (unweighted); dist [B,2] keeps the two distances; partials: 2*ceil(B/4) floats.
 * bwd: float-atomic scatter-add of g[0]*d loss + g[1]*d reg into caller-zeroed dEu / dEi / dEt (g = two upstream
 * gradients on the device, NULL = (1,1)). */
int tagrec_transtag_fwd_f32(const float* Eu, const float* Ei, const float* Et, int64_t ld, int D,
                            const int64_t* quad, int64_t B, float margin, float* dist, float* partials,
                            float* loss_out, void* stream);
int tagrec_transtag_bwd_f32(const float* Eu, const float* Ei, const float* Et, int64_t ld, int D,
                            const int64_t* quad, int64_t B, float margin, const float* dist, const float* g,
                            float* dEu, float* dEi, float* dEt, void* stream);

/* ---- Adam (torch.optim.Adam defaults as used at com.py:14,25,69), one fused pass -----------------
 *   m += (1-b1)(g-m);  v = b2 v + (1-b2) g^2;  p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps) */
int tagrec_adam_f32(float* p, const float* g, float* m, float* v, int64_t n,
                    float lr, float b1, float b2, float eps, int64_t step, void* stream);
/* The same update for n_tensors (small) parameter tensors in one launch per 64 of them: host arrays of device pointers and
 * element counts (< 2^31 each), one step count for all. */
int tagrec_adam_multi_f32(int n_tensors, float* const* p, const float* const* g, float* const* m, float* const* v,
                          const int64_t* n, float lr, float b1, float b2, float eps, int64_t step, void* stream);
/* The same update with the step counter (int64[1]) and the two step-dependent factors (float[2], scratch) in DEVICE
 * memory: the call advances *step_dev itself, so a HIP graph that captured it replays as successive steps. */
int tagrec_adam_graph_f32(float* p, const float* g, float* m, float* v, int64_t n,
                          float lr, float b1, float b2, float eps, int64_t* step_dev, float* coef_dev, void* stream);

/* The last backward hop with the optimizer folded in (training/basic_train.py:19-25: backward() then opt.step()):
 * row r of A G_in + b_scale B is the gradient of parameter row r; torch.optim.Adam's update (defaults; the arithmetic of
 * tagrec_adam_f32 at step `step`, counted from 1) is applied to p / m / v [n_rows, D] in the epilogue and no gradient
 * tensor is written.  in_flags / in_count / b_flags as in tagrec_spmm_axpy_sparse_f32.  G_in must not alias p. */
int tagrec_spmm_axpy_adam_f32(const tagrec_graph* g, const float* G_in, const uint8_t* in_flags, const unsigned* in_count,
                              const float* B, float b_scale, const uint8_t* b_flags, float* p, float* m, float* v,
                              float lr, float b1, float b2, float eps, int64_t step, int D, void* stream);
/* The same with the optimizer's step counter (int64) and its two step-dependent factors ([lr / (1 - b1^t), sqrt(1 - b2^t)])
 * in DEVICE memory: tagrec_adam_advance moves them one step forward on the stream (a one-thread kernel); the product reads
 * the factors there.  Nothing step-dependent is baked into the launch, so a captured HIP graph of the whole training step
 * (loss, backward, optimizer) replays correctly (train.GraphedStep with Adam(capturable=True).fuse_into(model)). */
int tagrec_adam_advance(int64_t* step_dev, float* coef_dev, float lr, float b1, float b2, void* stream);
int tagrec_spmm_axpy_adam_graph_f32(const tagrec_graph* g, const float* G_in, const uint8_t* in_flags,
                                    const unsigned* in_count, const float* B, float b_scale, const uint8_t* b_flags,
                                    float* p, float* m, float* v, float lr, float b1, float b2, float eps,
                                    int64_t* step_dev, float* coef_dev, int D, void* stream);

/* out = srcs[0] + ... + srcs[n_srcs - 1] (1 <= n_srcs <= 8, n floats each, summed left to right) in one pass: the
 * gradient of a table that several consumers read (what autograd's pairwise accumulation does in 3 passes per extra
 * gradient).  `srcs` is a HOST array of device pointers; `out` may alias a source. */
int tagrec_sum_n_f32(float* out, const float* const* srcs, int n_srcs, int64_t n, void* stream);

/* ---- NGCF layer, dense half (ngcf.py:77-86), exact-fp32 MFMA ------------------------------------------
 * Given N = A @ X (tagrec_spmm_f32) and W1p = W1 + b1, W2p = W2 + b2 (row-major [Din, Dout]; the
 * reference adds the 1 x Dout bias to the weight matrix, ngcf.py:78,82):
 *   fwd : Xp = LeakyReLU_0.2((N + X) W1p) + LeakyReLU_0.2((N * X) W2p);  inv_norm = 1/max(||Xp||, 1e-12);
 *         Z  = Xp / max(||Xp||, 1e-12) written with row stride ldz (one slot of the concat output)
 *   bwd : from dXp (gradient w.r.t. Xp): dP1, dP2 (pre-activation gradients), dN = dA1 + dA2 * X,
 *         dXd = dA1 + dA2 * N with dAi = dPi Wip^T.  The caller finishes dX = A^T dN + dXd
 *         (tagrec_spmm_axpy_f32).
 *   wgrad: dW1p = (N + X)^T dP1, dW2p = (N * X)^T dP2 (deterministic two-stage reduction; workspace of
 *         tagrec_ngcf_wgrad_workspace(Din, Dout) floats).  db = column sums of dWp (the bias broadcast).
 * Din, Dout in {16, 32, 64, 128} (128 -> 128 runs the backward kernels one matrix per launch: both LDS copies of both
 * matrices would not fit).
 * The *_rows_* forms serve the restricted training step (a layer whose output is needed on some rows only):
 *   row_mask (one byte per row, may be NULL = every row): rows whose byte is 0 are neither read nor written -- their
 *     slots in the outputs keep whatever they held, so every consumer must be handed the same mask (the weight gradient
 *     takes it too and counts such rows as zero);
 *   Z may be NULL (fwd): the normalised slot is not written (the step forms it on the batch rows from Xp and inv_norm);
 *   dz_flags (bwd, one byte per row, may be NULL): rows whose byte is 0 have dZ == 0; Xp, inv_norm and dZ are not read
 *     there (the concat gradient of a BPR step lives on the <= 3 B batch rows). */
int64_t tagrec_ngcf_wgrad_workspace(int Din, int Dout);
int tagrec_ngcf_dense_fwd_f32(const float* N, const float* X, const float* W1p, const float* W2p,
                              int64_t n_rows, int Din, int Dout, float* Xp, float* inv_norm, float* Z,
                              int64_t ldz, void* stream);
int tagrec_ngcf_dense_bwd_f32(const float* dXp, const float* N, const float* X, const float* W1p,
                              const float* W2p, int64_t n_rows, int Din, int Dout, float* dN, float* dXd,
                              float* dP1, float* dP2, void* stream);
/* bwd with the normalize-backward folded in: the gradient w.r.t. Xp is G (may be NULL; what the next layer sent
 * back) + normalize-backward(Xp, inv_norm, dZ), dZ = this layer's slot of the concat gradient, row stride ldz. */
int tagrec_ngcf_dense_bwd_norm_f32(const float* G, const float* Xp, const float* inv_norm, const float* dZ, int64_t ldz,
                                   const float* N, const float* X, const float* W1p, const float* W2p,
                                   int64_t n_rows, int Din, int Dout, float* dN, float* dXd, float* dP1,
                                   float* dP2, void* stream);
int tagrec_ngcf_wgrad_f32(const float* N, const float* X, const float* dP1, const float* dP2, int64_t n_rows,
                          int Din, int Dout, float* dW1p, float* dW2p, float* workspace,
                          int64_t workspace_floats, void* stream);
int tagrec_ngcf_dense_fwd_rows_f32(const float* N, const float* X, const float* W1p, const float* W2p,
                                   int64_t n_rows, int Din, int Dout, float* Xp, float* inv_norm, float* Z,
                                   int64_t ldz, const uint8_t* row_mask, void* stream);
int tagrec_ngcf_dense_bwd_rows_f32(const float* G, const float* Xp, const float* inv_norm, const float* dZ, int64_t ldz,
                                   const uint8_t* dz_flags, const float* N, const float* X, const float* W1p,
                                   const float* W2p, int64_t n_rows, int Din, int Dout, float* dN, float* dXd,
                                   float* dP1, float* dP2, const uint8_t* row_mask, void* stream);
int tagrec_ngcf_wgrad_rows_f32(const float* N, const float* X, const float* dP1, const float* dP2, int64_t n_rows,
                               int Din, int Dout, float* dW1p, float* dW2p, float* workspace,
                               int64_t workspace_floats, const uint8_t* row_mask, void* stream);

/* ---- TGCN neighbour-level attention (tgcn.py:20-37) over fixed-width neighbour tables (tgcn.py:194-202) ---
 * The caller forms the dense pieces with plain GEMMs:  P = ev W1[:D] + b  [n, A];  Q = ej W2  [m, A];
 * WT = cat(0, ew) W1[D:]  [n_wt, A] (row 0 = pad).  idx / widx: int32 [n, k], neighbour id + 1 and weight
 * index, 0 = pad (a zero row that still takes part in the softmax, as in the reference).
 *   fwd: attn[n,k] = softmax_k( relu(P[v] + WT[widx] + Q[idx-1]) . v );  out[v] = sum_k attn * Ej[idx-1]
 *   bwd: from dOut [n, D]: dP [n, A] (written), dWT [n_wt, A] and dv [A] (written; deterministic fold of
 *        per-block partials; workspace of tagrec_tgcn_attn_workspace(n_wt, A) floats), and EITHER
 *        dh = NULL: dQ [m, A] and dEj [m, D] by float-atomic scatter-add into caller-zeroed buffers, OR
 *        dh [n, k, A] (written): the per-(node, neighbour) pre-activation gradients; the caller then forms
 *        dQ = S dh and dEj = G dOut as pull products over the inverted neighbour table with tagrec_spmm_f32
 *        (S: unit values, columns = flat (node, slot) positions; G: values = attn, columns = source nodes).
 * D in {16,32,64,128,256}, A in {4,8,16,32,64}, k <= 64. */
int64_t tagrec_tgcn_attn_workspace(int n_wt, int A);
int tagrec_tgcn_attn_fwd_f32(const float* P, const float* Q, const float* WT, const float* v, const float* Ej,
                             const int32_t* idx, const int32_t* widx, int64_t n, int k, int D, int A,
                             float* attn, float* out, void* stream);
int tagrec_tgcn_attn_bwd_f32(const float* P, const float* Q, const float* WT, const float* v, const float* Ej,
                             const int32_t* idx, const int32_t* widx, const float* attn, const float* dOut,
                             int64_t n, int k, int D, int A, int n_wt, float* dP, float* dQ, float* dEj,
                             float* dh, float* dWT, float* dv, float* workspace, int64_t workspace_floats, void* stream);

/*   bwd, pull form without the dh round trip (three launches per relation over the relation's table inverted by
 *   destination: graph handle `inv` with rows = destinations, colidx = source rows, vals = attention weights):
 *     tagrec_attn_pull_da_f32 : dEj[j] = sum_p attn[p] dOut[p / k] (+ B[j]) and, on the way, da[p] = dOut[p / k] . Ej[j]
 *                               for every listed pair (the wave that walks row j holds both operands);
 *     tagrec_tgcn_attn_bwd_ds_f32 : from da: dP, dWT, dv as in attn_bwd, and comp [n k, 2] = (ds, relu bits of the A
 *                               pre-activations as an int32) per pair -- 8 bytes instead of dh's 4 A; no D-wide row is read;
 *                               w_major: a hint, the most frequent value of widx (its dWT row is summed in registers instead
 *                               of LDS atomics; any value, or -1, gives the same result);
 *     tagrec_attn_pull_dq_f32 : dQ[j] = v (.) sum_p ds[p] bits[p] (+ B[j]), over a handle whose colidx = PAIR ids.
 *   tagrec_attn_keys_i32: key[i] = idx[i] - 1, pads -> n_dst: the sort keys of the inversion.  A in {16, 32} (pull_dq). */
int tagrec_attn_pull_da_f32(const tagrec_graph* inv, const int32_t* pair, const float* dOut, const float* Ej,
                            const float* B, float* dEj, float* da, int D, void* stream);
/*   tagrec_attn_invert_fill: from the sort's permutation (int64 pair ids in destination order): pair (int32), src = pair / k,
 *   val = attn[pair] -- the colidx / vals of the inverted table that attn_pull_da walks (`inv`: rows = destinations,
 *   colidx = src, vals = val; `pair` rides along for the da store). */
int tagrec_attn_invert_fill(const int64_t* order, const float* attn, int k, int64_t n, int32_t* pair, int32_t* src, float* val,
                            void* stream);
int tagrec_tgcn_attn_bwd_ds_f32(const float* P, const float* Q, const float* WT, const float* v, const int32_t* idx,
                                const int32_t* widx, const float* attn, const float* da, int64_t n, int k, int A,
                                int n_wt, int w_major, float* dP, float* comp, float* dWT, float* dv, float* workspace,
                                int64_t workspace_floats, void* stream);
int tagrec_attn_pull_dq_f32(const tagrec_graph* inv, const float* comp, const float* v, int A, const float* B, float* dQ,
                            void* stream);
int tagrec_attn_keys_i32(const int32_t* idx, int64_t n, int32_t n_dst, int32_t* key, void* stream);
/*   tagrec_inv_filter_i32 : a step's inverted table WITHOUT a sort.  The neighbour tables are static (tgcn.py:194-202), so each
 *   relation's pair list sorted by destination is built once: perm_sorted (pair id v k + n of a non-pad slot) and dest_sorted
 *   (its destination id), n_entries of them.  A step computes a row subset: pos_src[v + 1] = 1 + position of source row v in
 *   it (0 = not computed; int32 [n_src + 1]); the entries of those rows are compacted IN ORDER into skey (destination,
 *   renumbered through pos_dst[d + 1] - 1 when pos_dst != NULL), pair (= position k + slot), src (= position) and val
 *   (= attn[pair]); skey[total .. capacity) is set to n_dst ("no entry").  capacity >= the number of listed entries (rows k
 *   is always enough).  workspace: tagrec_inv_filter_workspace(n_entries) int32, 8-byte aligned; workspace[1] = the total on
 *   return.  One pass over the static list (ticketed blocks, chained scan of their counts); k <= 64. */
int64_t tagrec_inv_filter_workspace(int64_t n_entries);
int tagrec_inv_filter_i32(const int32_t* perm_sorted, const int32_t* dest_sorted, int64_t n_entries, int k,
                          const int32_t* pos_src, const int32_t* pos_dst, const float* attn, int32_t n_dst,
                          int64_t capacity, int32_t* skey, int32_t* pair, int32_t* src, float* val,
                          int32_t* workspace, int64_t workspace_ints, void* stream);
/*   tagrec_nbr_gather_i32 : the neighbour tables of a ROW SUBSET in compact numbering (tgcn.py:194-202 restricted to the rows a
 *   mini-batch needs): out_idx[i, :] = pos[idx[rows[i], :]] (pos NULL: ids unchanged; pos[0] = 0 keeps the pad),
 *   out_widx[i, :] = widx[rows[i], :]. */
int tagrec_nbr_gather_i32(const int32_t* idx, const int32_t* widx, const int64_t* rows, const int32_t* pos, int64_t n_rows,
                          int k, int32_t* out_idx, int32_t* out_widx, void* stream);
/*     tagrec_attn_seg_dq_f32 : the same dQ from the pair list SORTED by destination (dest_sorted = the sorted keys, entries
 *                               with dest >= n_dst are the pads at the tail), as a balanced segmented sum: 256 entries per
 *                               wavefront, a finished segment ADDED to dQ by float atomics (the caller zeroes dQ; two relations
 *                               of one neighbour type add into the same buffer).  A in {4, 8, 16, 32}. */
int tagrec_attn_seg_dq_f32(const int32_t* dest_sorted, const int32_t* pair_sorted, int64_t n_entries, int32_t n_dst,
                           const float* comp, const float* v, int A, float* dQ, void* stream);

/* ---- the row plan of a restricted TGCN step (tgcn.py:236-249 reads the top layer at the batch rows only; the tables of
 *      tgcn.py:194-202 say which rows of the layer below those depend on) ---------------------------------------------
 * Node types 0, 1, 2 (user, item, tag) of sizes3[] nodes share one flag buffer: segment t starts at byte
 * tagrec_plan_segment_result(sizes3, t) (t = 3: the total = tagrec_plan_flags_workspace(sizes3) bytes) and holds n_t + 1
 * slots, slot 0 = the pad id of the neighbour tables, slot v + 1 = row v.
 *   plan_mark    : clears flags and counts, sets every row of a type with all3[t] != 0, then runs the descriptors
 *                  i < n_desc (at most 12): idx[i] == NULL -> flag rows[i][j * stride[i]] (j < n_rows[i]) of type
 *                  dst_type[i]; else flag idx[i][row, 0..k[i]) of type dst_type[i] for row = rows[i][j * stride[i]]
 *                  (rows[i] == NULL: row = j) < n_src[i].  Ids out of range are skipped and counted in counts[3].
 *   plan_compact : per type, the flagged rows in ascending order -> rows_out[row_base_t ..) (row_base = 0, n_0, n_0 + n_1;
 *                  capacity sum of sizes3), their number -> counts[t] (device int64 [4]), and the position map pos_out (int32,
 *                  the flag buffer's layout): pos_out[slot] = 1 + position of the row in its list, 0 if not flagged.
 *                  workspace: tagrec_plan_scan_workspace(sizes3) int32.
 *   plan_lookup  : out[i * out_stride] = pos[rows[i * stride] + 1] - 1 (a row id -> its position in a compact table). */
int64_t tagrec_plan_flags_workspace(const int64_t* sizes3);
int64_t tagrec_plan_segment_result(const int64_t* sizes3, int type);
int64_t tagrec_plan_scan_workspace(const int64_t* sizes3);
int tagrec_plan_mark_u8(int n_desc, const int32_t* const* idx, const int64_t* const* rows, const int64_t* stride,
                        const int64_t* n_rows, const int64_t* n_src, const int* k, const int* dst_type, const int* all3,
                        const int64_t* sizes3, uint8_t* flags, int64_t* counts, void* stream);
int tagrec_plan_compact_i64(const uint8_t* flags, const int64_t* sizes3, int64_t* rows_out, int32_t* pos_out, int64_t* counts,
                            int32_t* workspace, int64_t workspace_ints, void* stream);
int tagrec_plan_lookup_i64(const int32_t* pos, const int64_t* rows, int64_t stride, int64_t n, int64_t* out, int64_t out_stride,
                           void* stream);

/* ---- TGCN type-level attention + bit/vector convolutions + fusion layer, fused (tgcn.py:78-106) ------------
 * One node type per call.  T0/T1/T2 [n, D]: the (user-side, item-side, tag-side) vectors of each node, in
 * that order.  U [D, A], q [A], p [A]; wb [C, 3] (Conv2d(1,C,(3,1)) weight); w1 [V, D], w2 [V, 2, D],
 * w3 [V, 3, D] (Conv2d(1,V,(j,D)) weights); Wf [C*D + 6V, Dout], bf [Dout].
 *   fwd: out [n, Dout] = relu(y Wf + bf) with y the convolution features of e_j = softmax_j(relu(t_j U + q).p) t_j;
 *        bw_out [n, 3] = the softmax weights (kept for the backward pass).
 * Built for A = 32, C = 32, V = 8 (the reference's defaults) and D, Dout in {16, 32, 64, 128}. */
int tagrec_tgcn_fuse_fwd_f32(const float* T0, const float* T1, const float* T2, int64_t n, int D, int Dout,
                             int A, int C, int V, const float* U, const float* q, const float* p,
                             const float* wb, const float* w1, const float* w2, const float* w3,
                             const float* Wf, const float* bf, float* bw_out, float* out, void* stream);

/*   fwd with MESSAGE DROPOUT of the output (F.dropout on eu / ei / et, tgcn.py:217-219) in the epilogue: the n rows are up to
 *        three segments [0, lo1), [lo1, lo2), [lo2, n) -- the node types of a layer, merged into one launch -- each with its own
 *        seed (host array seeds3) and, when the launch computes a row subset, its node ids (rowsK, device int64; NULL = rows
 *        0 .. in order).  out = mask(seed, node id, column) relu(..) / (1 - p) with the counter-based mask of
 *        tagrec_dropout_rows_f32.  The backward entry points are used unchanged with out' = this output and dOut' =
 *        dOut / (1 - p): [out' > 0] = mask [out > 0]. */
int tagrec_tgcn_fuse_fwd_drop_f32(const float* T0, const float* T1, const float* T2, int64_t n, int D, int Dout,
                                  int A, int C, int V, const float* U, const float* q, const float* p,
                                  const float* wb, const float* w1, const float* w2, const float* w3,
                                  const float* Wf, const float* bf, float drop_p, const uint64_t* seeds3,
                                  const int64_t* rows0, const int64_t* rows1, const int64_t* rows2, int64_t lo1,
                                  int64_t lo2, float* bw_out, float* out, void* stream);

/*   bwd (data part): from dOut [n, Dout] and the forward's `out` (ReLU mask): dT0/dT1/dT2 [n, D]; yvec [n, 6V]
 *        (post-ReLU vector features) and dfeat [n, 6V] (their pre-activation gradients), dS [n, 3A] (type-attention
 *        pre-activation gradients) for the weight gradients; small [3C + 2A] = dwb | dq | dp
 *        (deterministic fold of per-block partials; workspace of tagrec_tgcn_fuse_bwd_workspace(Dout) floats). */
int64_t tagrec_tgcn_fuse_bwd_workspace(int Dout);
int tagrec_tgcn_fuse_bwd_f32(const float* T0, const float* T1, const float* T2, int64_t n, int D, int Dout,
                             int A, int C, int V, const float* U, const float* q, const float* p,
                             const float* wb, const float* w1, const float* w2, const float* w3,
                             const float* Wf, const float* out, const float* dOut, float* dT0, float* dT1,
                             float* dT2, float* yvec, float* dfeat, float* dS, float* small,
                             float* workspace, int64_t workspace_floats, void* stream);

/*   wf: every weight gradient that is a product over the NODE axis, in one launch.  result (of
 *        tagrec_tgcn_fuse_wf_result(D, Dout) floats) = [ dWf | G | dU ]:
 *          dWf [C*D + 6V, Dout] = y^T (dOut * [out > 0]), y re-formed on the fly from T0..T2, bw [n, 3], wb, yvec;
 *          G   [6V, 3, D]       = dfeat^T e_j  (the caller folds G into the Conv2d(1,V,(j,D)) weight gradients);
 *          dU  [D, A]           = sum_j t_j^T dS_j.
 *        workspace of tagrec_tgcn_fuse_wf_workspace(D, Dout) floats; deterministic fold of per-node-group partials. */
int64_t tagrec_tgcn_fuse_wf_result(int D, int Dout);
int64_t tagrec_tgcn_fuse_wf_workspace(int D, int Dout);
int tagrec_tgcn_fuse_wf_f32(const float* T0, const float* T1, const float* T2, const float* bw, const float* yvec,
                            const float* wb, const float* out, const float* dOut, const float* dfeat, const float* dS,
                            int64_t n, int D, int Dout, int C, int V, float* result, float* workspace,
                            int64_t workspace_floats, void* stream);

/* ---- tall-skinny dense products of the TGCN step (tgcn.py:20-37 after the split of the attention's matmuls) ----
 * exact-fp32 MFMA, HBM-bound: a node table times a small matrix, and the reverse products.  K, NO in {16, 32, 64, 128}.
 *   tall_mm : Y[n, NO] (+)= [X1 | X2][sel][n, K] W[K, NO] + bias.
 *             X2 != NULL: the tall operand is two tensors of K/2 columns each; sel (int64 [n], may be NULL) gathers its rows.
 *             W(k, c) = W1[k * w_sk + c * w_sc] (strides in floats: a transposed matrix is a stride pair); w_split = 1: W2
 *             holds the rows k >= K/2, w_split = 2: W2 holds the columns c >= NO/2 (same strides).
 *             Y2 != NULL: columns [NO/2, NO) go to Y2, both outputs with row stride NO/2 (b1 / b2: their biases, may be NULL).
 *             accumulate != 0: the product is added to what Y holds.
 *   tall_wgrad : dW[KI, NO] (+)= X^T [dY1 | dY2] and db1 / db2 (+)= column sums of dY1 / dY2 (any output may be NULL),
 *             contraction over the n rows; per-wave partials in `workspace` (tagrec_tall_wgrad_workspace(KI, NO) floats) folded
 *             in a fixed order.
 *   small_mm : C[M, N] (+)= A B with A(i, k) = A[i * sam + k * sak], B(k, j) = B[k * sbk + j * sbn] (look-up-table sized).
 *   row_add_at : dst[pos[i], :] += src[i, :] for DISTINCT positions (plain read-modify-write). */
int tagrec_tall_mm_f32(const float* X1, const float* X2, const int64_t* sel, int64_t n, int K, int NO, const float* W1,
                       const float* W2, int64_t w_sk, int64_t w_sc, int w_split, const float* b1, const float* b2,
                       float* Y1, float* Y2, int accumulate, void* stream);
/*   tall_mm_adam : the gradient G = G_in + X W (X [n, K], W(k, c) = W[k * w_sk + c * w_sc], G_in [n, NO] or NULL) of the
 *             embedding table p [n, NO] is not stored: torch.optim.Adam's update of p, exp_avg m and exp_avg_sq v (the arithmetic
 *             of tagrec_adam_f32, step = the number of this update) is applied in the product's epilogue -- the last term of
 *             a TGCN table's gradient is dQ W_2^T (tgcn.py:20-37), so the optimizer never reads a gradient tensor
 *             (training/basic_train.py:21,25). */
int tagrec_tall_mm_adam_f32(const float* X, int64_t n, int K, int NO, const float* W, int64_t w_sk, int64_t w_sc,
                            const float* G_in, float* p, float* m, float* v, float lr, float b1, float b2, float eps,
                            int64_t step, void* stream);
int64_t tagrec_tall_wgrad_workspace(int KI, int NO);
int tagrec_tall_wgrad_f32(const float* X, const float* dY1, const float* dY2, int64_t n, int KI, int NO, float* dW,
                          float* db1, float* db2, int acc_w, int acc_b, float* workspace, int64_t workspace_floats, void* stream);
int tagrec_small_mm_f32(const float* A, const float* B, float* C, int M, int N, int K, int64_t sam, int64_t sak,
                        int64_t sbk, int64_t sbn, int accumulate, void* stream);
int tagrec_row_add_at_f32(float* dst, const int64_t* pos, const float* src, int64_t n_rows, int D, void* stream);
/*   masked_colsum : result[D] = column sums of dOut (.) [out > 0] -- the bias gradient of a ReLU layer (dbf of the fusion
 *   layer, tgcn.py:104-106); per-block partials in `workspace` (tagrec_masked_colsum_workspace(D) floats), fixed-order fold. */
int64_t tagrec_masked_colsum_workspace(int D);
int tagrec_masked_colsum_f32(const float* dOut, const float* out, int64_t n_rows, int D, float* result, float* workspace,
                             int64_t workspace_floats, void* stream);

/* ---- bandwidth probes (measured ceilings of the box next to the 8 TB/s specification) ----------------
 * a = b + s * c over n floats (stream triad; 12 bytes per element; c == NULL: copy a = b, 8 bytes per element; non_temporal
 * picks the load / store flavour), and a random whole-row gather with the access shape
 * of the SpMM's neighbour gather (64 rows of D floats per wavefront and index chunk, nothing else): sums of the gathered
 * rows land in out[tagrec_probe_gather_out_floats()], which exists only to keep the loads alive.  Indices are NOT
 * range-checked on the device: the caller guarantees 0 <= idx[i] < n_rows. */
int tagrec_probe_triad_f32(float* a, const float* b, const float* c, float s, int64_t n, int non_temporal, void* stream);
/* read-only stream over n floats; out holds tagrec_probe_gather_out_floats() floats (per-thread sums that keep the loads alive) */
int tagrec_probe_read_f32(const float* b, int64_t n, float* out, void* stream);
int64_t tagrec_probe_gather_out_floats(void);
int tagrec_probe_gather_rows_f32(const float* table, int64_t n_rows, int D, const int32_t* idx, int64_t n_idx, float* out,
                                 void* stream);
/* One wavefront spins for spin_us and returns out2 = {shader cycles, 100 MHz ticks}: launched on another stream before a
 * kernel, it measures the clock that kernel runs at (the peaks of the roofline are quoted at 2.4 GHz). */
int tagrec_probe_clock(int64_t spin_us, int64_t* out2, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TAGREC_H */
