/*
 * rri_hip.h -- C ABI of librri_hip.so: the MI355X (gfx950) implementation of the
 * rank-one residue iteration (RRI) inner loop of maksimt/rri_nmf.
 *
 * The reference has no FFI: its seams are the Python callable
 *     nmf(X, k, ...) -> dict                       src/rri_nmf/nmf.py:98-108, :551-560
 * and the per-topic helpers it calls through **locals():
 *     _compute_update_T                            nmf.py:633-715
 *     _compute_update_W                            nmf.py:718-747
 *     qf_min                                       optimization.py:12-88
 *     euclidean_proj_simplex / proj_mat_to_simplex matrixops.py:5-100
 *     _project_and_check_reset_t / _check_reset_W  nmf.py:751-816
 *     TrueObjComputer.true_objective               nmf.py:71-94
 * A maintainer of the reference would bind exactly the entry points below with
 * ctypes (INTEGRATION.md shows the stub) and call them from nmf()'s sweep loop.
 *
 * Conventions
 *   - plain C, no torch / C++ types; every function returns an rri_status (0 = ok),
 *     never throws; rri_last_error() gives the text of the last failure on a handle.
 *   - host matrices are row-major, caller-owned and never written unless the
 *     argument is an explicit output; `ld` arguments are row strides in ELEMENTS.
 *   - one opaque handle per nmf() call; calls on one handle are serialised by the
 *     caller; all device work of a handle runs on ONE HIP stream (its own, or the
 *     caller's when `stream` is non-NULL, e.g. torch's current stream).
 *   - notation follows the reference: X n*d, W n*k ("documents x topics"),
 *     T k*d ("topics x features"; north_star's H).  W[:,t] is Ho's u_t, T[t,:] is v_t^T.
 */
#ifndef RRI_HIP_H
#define RRI_HIP_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RRI_ABI_VERSION 1
/* the Gram part of the reduce buffer travels as this many slice sums (see rri_topic_reduce_local) */
#define RRI_GRAM_SLICES 8
/* largest rank k a handle takes */
#define RRI_MAX_K 1024

typedef struct rri_ctx rri_ctx;   /* opaque: one nmf() call (or one row shard of it) on one GPU */
typedef struct rri_comm rri_comm; /* opaque: the communicator of a row-sharded run, one per process */
#define RRI_COMM_ID_BYTES 128     /* sizeof(ncclUniqueId) */

typedef int32_t rri_status;
enum {
    RRI_OK = 0,
    RRI_PAUSED = 1,             /* a sweep stopped at a rare branch; see rri_pending_event */
    RRI_ERR_INVALID = -1,       /* bad argument / shape / state                              */
    RRI_ERR_HIP = -2,           /* a HIP runtime call failed                                */
    RRI_ERR_UNSUPPORTED = -3,   /* option not available on the device path                  */
    RRI_ERR_UNBOUNDED = -4,     /* qf_min: minimum objective is unbounded (optimization.py:66-67,76-77) */
    RRI_ERR_W_COL_ZERO = -5,    /* assert sum(W[:,t]) > 0 failed (nmf.py:476)               */
    RRI_ERR_NOT_IMPLEMENTED = -6, /* qf_min: c<=0 with s not in {None,1.0} (optimization.py:72-73) */
    RRI_ERR_COMM = -7            /* a collective (RCCL, or the caller's transport) failed                   */
};

enum { RRI_F32 = 0, RRI_F64 = 1 };   /* storage type of X, mask, residual in HBM (arithmetic is float64) */
/* rri_create's `weighted`: the flavour of the handle */
enum { RRI_UNWEIGHTED = 0, RRI_WEIGHTED_DENSE = 1, RRI_WEIGHTED_SPARSE = 2, RRI_UNWEIGHTED_RESIDUAL = 3 };
enum { RRI_RESET_NONE = 0, RRI_RESET_MAX_RESID_DOCUMENT = 1, RRI_RESET_RANDOM = 2 };
enum { RRI_EVENT_NONE = 0, RRI_EVENT_RESET_T = 1, RRI_EVENT_RESET_W = 2 };

/* The options of nmf() that reach the inner loop (nmf.py:98-108).  "has_*" = the
 * Python value is not None. */
typedef struct rri_params {
    int32_t fix_W;                 /* nmf.py:460 */
    int32_t fix_T;                 /* nmf.py:417 */
    int32_t project_T_each_iter;   /* s = t_row_sum in the T-row qf_min, nmf.py:442-447 */
    int32_t has_t_row_sum;
    int32_t has_w_row_sum;         /* scalar w_row_sum: the `ub` of the W-column qf_min, nmf.py:469 */
    int32_t reset_method;          /* RRI_RESET_*; nmf.py:104 */
    int32_t resets_left;           /* nmf.py:105,192-193: budget still available */
    int32_t reserved0;
    double t_row_sum;
    double w_row_sum;
    double reg_w_l1, reg_w_l2, reg_t_l1, reg_t_l2; /* nmf.py:106 */
    double eps_div;                /* np.spacing(10), nmf.py:52 */
} rri_params;

typedef struct rri_event {
    int32_t kind;          /* RRI_EVENT_* */
    int32_t topic;         /* t: the row of T / column of W to reset */
    int32_t sweep;         /* sweep index (inside the interrupted call) of the step that detected it */
    int32_t resume_topic;  /* topic of that step: a RESET_T resumes at the W half of `topic`, a RESET_W at
                              the T half of `resume_topic` */
} rri_event;

/* ---- lifetime ------------------------------------------------------------------ */
uint32_t   rri_abi_version(void);
/* dtype: RRI_F32 / RRI_F64.  weighted: RRI_UNWEIGHTED; RRI_WEIGHTED_DENSE reserves the mask and masked-residual
 * buffers of the elementwise-weighted flavour (WRRI, nmf.py:687-701,735-746) as dense n x d arrays;
 * RRI_WEIGHTED_SPARSE keeps that flavour on a 0/1 observation pattern only (rri_upload_observed_csr).
 * RRI_UNWEIGHTED_RESIDUAL is the unweighted flavour on the EXPLICIT residual R = X - W T (kept in HBM beside X): every
 * topic step is one rank-one residual update pass R <- R -+ v u^T fused with the residual products R t / R^T w of
 * nmf.py:670-676,728-734 (SURVEY 8a "explicit-residual variant"; k >= 2, both halves free), 2 n d bytes per step
 * where the default Gram-form schedule (RRI_UNWEIGHTED) reads X once.  Same results to rounding.
 * device: HIP device ordinal.  stream: hipStream_t to run on, or NULL for an own stream. */
rri_status rri_create(rri_ctx** out, int64_t n, int64_t d, int32_t k, int32_t dtype,
                      int32_t weighted, int32_t device, void* stream);
rri_status rri_destroy(rri_ctx* ctx);
const char* rri_last_error(const rri_ctx* ctx);   /* NULL ctx: error of the last failed rri_create */

/* ---- data: replaces the numpy arrays nmf() holds (nmf.py:272,351,867-868) --------- */
/* host_dtype: RRI_F32 / RRI_F64 of the HOST buffer; converted to the handle's dtype. */
rri_status rri_upload_X(rri_ctx* ctx, const void* host, int64_t ld, int32_t host_dtype);
rri_status rri_upload_mask(rri_ctx* ctx, const void* host, int64_t ld, int32_t host_dtype); /* W_mat */
/* Zero-copy alternative: X (and mask) already in device memory in the handle's dtype,
 * row-major with row stride ld (a multiple of 16 bytes, base 16-byte aligned).  The calls synchronise the device
 * once, so memory just produced on another stream is complete (rri_bind_mask_device reads the mask at once to
 * bit-pack it); later changes to the memory are the caller's to order against the handle's stream. */
rri_status rri_bind_X_device(rri_ctx* ctx, const void* dev, int64_t ld);
rri_status rri_bind_mask_device(rri_ctx* ctx, const void* dev, int64_t ld);
/* Ingestion without host densification (the reference densifies with .toarray(), sklearn_interface.py:78-102):
 * X from host CSR arrays (indptr n+1 int64, indices int32 column ids, data of data_dtype) scattered into the
 * zero-filled dense device X; and the observation mask W_mat = [X != 0] of such a matrix, built bit-packed. */
rri_status rri_upload_X_csr(rri_ctx* ctx, const int64_t* indptr, const int32_t* indices, const void* data,
                            int64_t nnz, int32_t data_dtype);
rri_status rri_upload_mask_csr_pattern(rri_ctx* ctx, const int64_t* indptr, const int32_t* indices,
                                       const void* data, int64_t nnz, int32_t data_dtype);
/* RRI_WEIGHTED_SPARSE handles: W_mat is the 0/1 pattern of a CSR matrix and X is given on that pattern only
 * (values[p] = X[r, indices[p]], explicit zeros allowed; X is taken as 0 elsewhere, where W_mat = 0 hides it from
 * every sum of nmf.py:687-746 and the reset search of nmf.py:770-773 sees max(0 - WT, 0) = 0).  The recommender
 * case of sklearn_interface.py:78-102 without any dense n x d array: the residual is kept on the pattern, as a
 * CSR and a CSC copy.  Replaces rri_upload_X + rri_upload_mask for such a handle. */
rri_status rri_upload_observed_csr(rri_ctx* ctx, const int64_t* indptr, const int32_t* indices,
                                   const void* values, int64_t nnz, int32_t data_dtype);
rri_status rri_set_W(rri_ctx* ctx, const void* host, int64_t ld, int32_t host_dtype);  /* n*k */
rri_status rri_set_T(rri_ctx* ctx, const void* host, int64_t ld, int32_t host_dtype);  /* k*d */
rri_status rri_get_W(rri_ctx* ctx, void* host, int64_t ld, int32_t host_dtype);
rri_status rri_get_T(rri_ctx* ctx, void* host, int64_t ld, int32_t host_dtype);
rri_status rri_set_params(rri_ctx* ctx, const rri_params* p);

/* ---- the hot path: the topic loop of nmf.py:415-476 ----------------------------- */
/* Runs n_sweeps Gauss-Seidel sweeps (each = k topic steps: T-row update, then W-column
 * update).  Returns RRI_OK, RRI_PAUSED (a reset condition of nmf.py:762-783 / :796-816
 * was met: query rri_pending_event, resolve it with rri_apply_reset_* and call
 * rri_resume), or a negative error mirroring the reference's exceptions.
 * sweeps_done (may be NULL) receives the number of COMPLETED sweeps of this call.
 * Schedules chosen by the library, same results to summation order and the same events, statuses and resume positions:
 * launch-bound sizes run as one persistent launch (rri_onchip_info); with fix_T set (the fold-in of nmf.py:417 / 460-476,
 * sklearn_interface.py:327-333) an unweighted handle takes X T^T and T T^T once per T and runs the W half of all k topics of
 * a sweep as ONE launch over the rows of W, the column checks of nmf.py:471-476 in topic order after it (k n + k^2 doubles
 * of device memory more, allocated at the first such sweep; RRI_WSWEEP=0 in the environment of rri_create: topic by topic);
 * dense weighted handles whose W_mat is 0 / 1 and below 12 % set keep a second bit-packed copy of it (n d / 8 bytes);
 * handles of 10^8 elements or more time their pass once, before the first sweep, under the two ways of dealing its tiles to
 * the XCDs and keep the faster (the read-only pass over X: ~5 ms; the read-modify-write pass over a stored residual --
 * RRI_UNWEIGHTED_RESIDUAL, dense weighted --: ~12 ms, the residual rewritten with its own values meanwhile; RRI_ROT_CAL=0: not). */
rri_status rri_sweep(rri_ctx* ctx, int32_t n_sweeps, int32_t* sweeps_done);
rri_status rri_resume(rri_ctx* ctx, int32_t* sweeps_done);
rri_status rri_pending_event(rri_ctx* ctx, rri_event* ev);
/* 'max_resid_document' (nmf.py:770-776, :804-810), entirely on the device. */
rri_status rri_apply_reset_max_resid(rri_ctx* ctx, int32_t t, int64_t* row_chosen);
/* 'random' (nmf.py:778-783, :811-816): the caller draws the vectors with numpy's
 * global RNG (as the reference does) and hands them over. */
rri_status rri_apply_reset_vectors(rri_ctx* ctx, int32_t t, const double* T_row, const double* W_col);
/* skip the pending reset (budget exhausted, nmf.py:765-768 / :797-800) */
rri_status rri_skip_reset(rri_ctx* ctx);

/* Finer-grained steps (tests, host drivers): one half of one topic step. */
rri_status rri_update_T_row(rri_ctx* ctx, int32_t t);   /* nmf.py:417-458 */
rri_status rri_update_W_col(rri_ctx* ctx, int32_t t);   /* nmf.py:460-476 */

/* ---- the explicit residual (RRI_UNWEIGHTED_RESIDUAL handles) ---------------------- */
/* R = X - W T for the handle's current factors (the products of nmf.py:670-676 / 728-734 are taken on it). */
rri_status rri_residual_rebuild(rri_ctx* ctx);
/* The outer-product residual update as an operation of its own:
 *     R <- R - a b^T [- a2 b2^T]      (a, a2: n doubles; b, b2: d doubles; a2 = b2 = NULL for one term)
 * fused with y = R_new trow (n, nmf.py:729) and z = R_new^T wcol (d, nmf.py:672) of the updated residual, all host
 * vectors, float64 arithmetic, R rounded to the handle's storage type when stored.  y_out / z_out may be NULL.
 * The stored R then no longer belongs to the handle's (W, T): the next sweep rebuilds it. */
rri_status rri_residual_update(rri_ctx* ctx, const double* a, const double* b, const double* a2, const double* b2,
                               const double* trow, const double* wcol, double* y_out, double* z_out);
rri_status rri_get_residual(rri_ctx* ctx, void* host, int64_t ld, int32_t host_dtype);   /* n*d, as stored */

/* ---- around the loop ------------------------------------------------------------ */
/* Row-wise simplex projection of W (proj_mat_to_simplex, matrixops.py:72-100; final
 * projection nmf.py:519-529).  s_vec == NULL: every row to the scalar s; else n doubles. */
rri_status rri_project_W_rows(rri_ctx* ctx, double s, const double* s_vec);
/* true_objective (nmf.py:71-94) with the handle's regularisers and mask; float64 accumulation.  Right after rri_sweep on an
 * unweighted handle no pass over X is made (1/2 ||X||^2 - sum_t <w_t, X t_t> + 1/2 <W^T W, T T^T>, the cross terms left by the
 * sweep); after a sweep of the register-resident kernel nothing is launched at all -- the kernel left the value with the state
 * the host reads anyway (RRI_ONCHIP_OBJ=0: the Gram kernels instead). */
rri_status rri_objective(rri_ctx* ctx, double* out);
/* argmax over topics of every row of W (harden_distributions, matrixops.py:203-209). */
rri_status rri_argmax_rows(rri_ctx* ctx, int32_t* out_host);
/* X*T^T clipped reconstruction error on listed entries: RMSE of NMF_RS_Estimator.score /
 * RMSE_val (sklearn_interface.py:85-91,172-182).  idx = (i,j) pairs, vals = ratings.
 * On a handle with a communicator attached the call is collective: (i, j) are this rank's LOCAL rows (count may be 0),
 * the result is sqrt(sum of squared errors over all ranks / number of entries over all ranks), equal on every rank -- the
 * early-stop decision of nmf.py:381-407 is then the same everywhere.  A rank whose own arguments are bad (an index out of range,
 * a NULL list) still takes part in the collective and raises a flag in it: EVERY rank returns RRI_ERR_INVALID (the failing one
 * names its entry, the others say "on another rank"), nobody is left waiting, and the ranks stay in step for the next call. */
rri_status rri_masked_rmse(rri_ctx* ctx, const int64_t* ij, const double* vals, int64_t count,
                           double clip_lo, double clip_hi, double* out);
/* device-side copy of (W,T) for the early-stop rollback of nmf.py:360-363,393-407 */
rri_status rri_snapshot(rri_ctx* ctx);
rri_status rri_rollback(rri_ctx* ctx);

/* Products with the resident X for the initialisation (randomized SVD behind NNDSVD, initialization.py:105;
 * SURVEY 8f rank 2): out = X B (B: d x m host row-major, out: n x m) and out = X^T Q (Q: n x m, out: d x m),
 * float64 arithmetic on the stored X.  Blocking; not part of the sweep path.  On a pattern-only handle X is the
 * matrix of the observed values (= W_mat .* X, what initialization.py:104 factorises for the weighted flavour)
 * and m <= 64 per call. */
rri_status rri_X_times(rri_ctx* ctx, const double* B, int32_t m, double* out);
rri_status rri_Xt_times(rri_ctx* ctx, const double* Q, int32_t m, double* out);
/* The whole range finder of that randomized SVD in one call, its panels resident on the device (sklearn.utils.extmath.
 * randomized_svd behind initialization.py:105: n_iter rounds of Q <- normalise(A Q), Q <- normalise(A^T Q); Q <- orth(A Q);
 * B = Q^T A), with A = X (transpose = 0; Q0: d x m) or A = X^T (transpose = 1, scikit-learn's choice when n < d; Q0: n x m).
 * Every normalisation is (shifted) Cholesky-QR, three passes, instead of the host's LU / QR: another basis of the same range, so the SVD built
 * on Q and B (the small SVD of B stays with the caller) is scikit-learn's up to rounding.  Q0, Q_out (rows of A x m) and
 * B_out (m x columns of A) are host arrays, row-major; 1 <= m <= 64.  Dense handles only. */
rri_status rri_range_finder(rri_ctx* ctx, const double* Q0, int32_t m, int32_t n_iter, int32_t transpose, double* Q_out,
                            double* B_out);

/* Preprocessing of the resident dense X in place (unweighted handles):
 *   rri_column_positive_counts  df[j] = #{i : X[i,j] > 0}, the document frequencies of tfidf (matrixops.py:169)
 *   rri_scale_X                 X[i,j] <- (X[i,j] * col_scale[j]) / (sum_j X[i,j] * col_scale[j] + spacing(1)) when
 *                               normalize_rows != 0 (rows summing to < 1e-10 become 1/d: normalize, matrixops.py:139-147),
 *                               else X[i,j] * col_scale[j] (X * idf, matrixops.py:172); col_scale == NULL: all ones.
 * float64 arithmetic, rounded to the handle's storage type when stored. */
rri_status rri_column_positive_counts(rri_ctx* ctx, double* df_out);
rri_status rri_scale_X(rri_ctx* ctx, const double* col_scale, int32_t normalize_rows);

/* ---- row-sharded multi-GPU inside the library (one process per GPU; SURVEY 8b "sharded by row block internally",
 *      8e).  The reference has one call for the whole X (nmf.py:98-108); here every rank creates a handle for its row
 *      block [row_offset, row_offset + n) of the n_global-row problem (W rows alike, T replicated), attaches the
 *      process's communicator, and then calls the SAME entry points -- rri_sweep, rri_resume, rri_update_*,
 *      rri_objective, rri_apply_reset_* -- collectively, with the same arguments on every rank.  The one cross-row
 *      reduction of a topic step ([w_t^T X | w_t^T W | ||w_t||^2 | sum W[:,t-1]], or [numerator | denominator] of the
 *      weighted flavour) is all-reduced by RCCL over xGMI on the handle's stream between two kernels: a sweep is ONE
 *      call with no host work per topic.  Every decision (reset events, the assert of nmf.py:476, unbounded cases)
 *      is taken from all-reduced or replicated values, so all ranks return the same status.
 *      All flavours and flags of rri_sweep: fixed halves, k = 1, weighted dense and pattern-only, and the explicit-residual
 *      schedule (RRI_UNWEIGHTED_RESIDUAL: every rank keeps its rows of R; the sums it sends are the same message). */
/* rank 0 makes the id (ncclGetUniqueId) and hands it to the other ranks by any channel the host program has */
rri_status rri_comm_unique_id(uint8_t* id_out /* RRI_COMM_ID_BYTES */);
/* RCCL communicator of `world` ranks; this process is `rank` and drives HIP device `device`.  Collective. */
rri_status rri_comm_create(rri_comm** out, const uint8_t* id, int32_t rank, int32_t world, int32_t device);
/* The same protocol over the CALLER's transport (tests with several ranks on one GPU; hosts without RCCL): the
 * library drains the stream, hands `count` doubles in host memory to the function and copies the result back.
 * Each function returns 0 on success.  allreduce: sum in place over the ranks; allgather: recv = world x count;
 * broadcast: from `root` in place. */
typedef int32_t (*rri_allreduce_fn)(void* user, double* buf, int64_t count);
typedef int32_t (*rri_allgather_fn)(void* user, const double* send, int64_t count, double* recv);
typedef int32_t (*rri_broadcast_fn)(void* user, double* buf, int64_t count, int32_t root);
rri_status rri_comm_create_host(rri_comm** out, int32_t rank, int32_t world, rri_allreduce_fn allreduce,
                                rri_allgather_fn allgather, rri_broadcast_fn broadcast, void* user);
rri_status rri_comm_destroy(rri_comm* comm);     /* after every handle it was attached to is destroyed */
/* comm == NULL detaches.  row_offset: global index of the handle's first row. */
rri_status rri_attach_comm(rri_ctx* ctx, rri_comm* comm, int64_t row_offset, int64_t n_global);
/* small host-side collectives for the host driver (the vectors of a 'random' reset drawn on rank 0, nmf.py:778-783;
 * stop decisions; the d x m panels of a row-sharded randomized SVD): no-ops on a handle without a communicator. */
rri_status rri_comm_broadcast(rri_ctx* ctx, double* host, int64_t count, int32_t root);
rri_status rri_comm_allreduce_sum(rri_ctx* ctx, double* host, int64_t count);
rri_status rri_comm_stats(rri_ctx* ctx, int32_t* rank, int32_t* world, int64_t* allreduce_calls);

/* ---- row-sharded stepping with the collective in the CALLER's hands (the protocol the calls above run inside
 *      rri_sweep; kept for hosts that own their collectives and for the Gaussian mechanism, whose noise enters at the
 *      same point) ---- */
/* A topic step splits at the one cross-row reduction it needs.  rri_topic_reduce_local
 * leaves this rank's partial sums [w_t^T X (LD) | RRI_GRAM_SLICES x (w_t^T W (k), ||w_t||^2, sum W[:,t-1])]
 * (LD = d rounded up to the 16-byte row stride; a Gram entry is the sum of its slices)
 * in the reduce buffer; the caller all-reduces (sum) that buffer over the ranks (RCCL over
 * xGMI via torch.distributed) and calls rri_topic_finish, which is rank-local.
 * Weighted handles (WRRI, nmf.py:687-701): the buffer is [numerator w_t^T Rt (LD) | denominator
 * (w_t^2)^T M (LD) | sum of W[:,t-1], its negative-denominator flag], LD = d rounded up to the
 * 16-byte row stride; same protocol (one all-reduce of [numerator | denominator] per topic). */
rri_status rri_reduce_buffer(rri_ctx* ctx, void** dev_ptr, int64_t* n_elems); /* dtype = handle's */
rri_status rri_bind_reduce_buffer(rri_ctx* ctx, void* dev_ptr, int64_t n_elems);
/* (T must be free; W may be fixed -- the step is then the T row alone, the kept column taking its scale, nmf.py:450-452 -- and
 * k may be 1: the reference's loop, nmf.py:417-456, has no limit there.) */
rri_status rri_topic_reduce_local(rri_ctx* ctx, int32_t t);
/* Host access to the first `count` doubles of the reduce buffer between rri_topic_reduce_local and
 * rri_topic_finish: what a caller needs to perturb the T-row sums -- the Gaussian mechanism of nmf.py:422-435
 * adds its noise to wR and nw there.  Blocking. */
rri_status rri_reduce_read(rri_ctx* ctx, double* out, int64_t count);
rri_status rri_reduce_write(rri_ctx* ctx, const double* in, int64_t count);
rri_status rri_topic_finish(rri_ctx* ctx, int32_t t);
/* only the W-column half of topic t (a sharded run resuming after a T-row reset) */
rri_status rri_topic_finish_w(rri_ctx* ctx, int32_t t);
/* Pieces of the 'max_resid_document' reset for a sharded run (nmf.py:771-776): the largest
 * sum_j max(X - W T, 0)_ij^2 over THIS rank's rows and its local row index; and the reset row
 * max(X[i,:] - W[i,:] T, 0) of a local row (d doubles to the host).  The caller picks the global winner. */
rri_status rri_resid_row_argmax(rri_ctx* ctx, double* value, int64_t* local_row);
rri_status rri_reset_row(rri_ctx* ctx, int64_t local_row, double* row_out_host);
/* status word of the interrupted / failed step after a stream sync (same codes as rri_sweep) */
rri_status rri_poll(rri_ctx* ctx);
/* rank-local pieces of the objective: out[0] = 0.5*sum (Wm.)(X-WT)^2 over local rows,
 * out[1] = sum W^2, out[2] = sum |W| (local rows); T terms are replicated. */
rri_status rri_objective_parts(rri_ctx* ctx, double out[3]);

/* ---- measurement --------------------------------------------------------------- */
/* HIP-event timing of the streaming kernels on the handle's stream.  kernel_id:
 * 0 = fused X pass (row dots + column sums), 1 = W-column update, 2 = T-row update chain,
 * 3 = rank-one residual update (explicit-residual / WRRI flavour). */
/* on = 0: off; on = N > 0: bracket every N-th launch of each kernel id with an event pair (N = 1: all).
 * An event record costs a bubble of a few microseconds on the stream, so sample when the timed region is
 * also the throughput measurement. */
rri_status rri_timing_enable(rri_ctx* ctx, int32_t on);
rri_status rri_timing_read(rri_ctx* ctx, int32_t kernel_id, int64_t* launches, double* total_ms);
rri_status rri_synchronize(rri_ctx* ctx);
/* Launch-bound sizes: when X fits the registers of the chip -- 40 MB: about 10000 x 1024 or 5000 x 2048 in fp32, half the rows in
 * float64 -- and the configuration is the unweighted one with both halves free (plain with d <= 2048, or the topic-model flags
 * with T rows projected at every step and d <= 1024), k in 2..64, one device, rri_sweep / rri_resume run as ONE persistent
 * launch with X resident on chip and two (topic model: three) exchanges between workgroups per topic step
 * (rri_onchip_kernels.hpp) instead of three or four launches per topic step; the whole launch is then timed as kernel_id 0.  *eligible: would the next
 * rri_sweep take that path; *launches: how many it has taken on this handle.  RRI_ONCHIP=0 (environment, read by
 * rri_create) switches it off.  Either pointer may be NULL. */
rri_status rri_onchip_info(rri_ctx* ctx, int32_t* eligible, int64_t* launches);
/* The exchanges of that launch poll a bounded number of times.  When its workgroups cannot all run at once (a device shared
 * with another process, CUs masked away) the launch gives up, and the call does what the reference's sweep does under any
 * scheduling (nmf.py:415-476): it completes -- W and T are put back to what they were before the launch and the same steps
 * run on the launch-per-phase schedule, which the handle then keeps for a while: 2 s after its first fallback, twice as long after
 * every further one up to 64 s (and every handle of the process stays off the path for RRI_ONCHIP_BACKOFF_MS, default 2000).
 * *eligible of rri_onchip_info and RRI_ERR_UNSUPPORTED of rri_sweep_until therefore depend on the clock after a fallback.
 * *fallbacks: how often that happened on this handle. */
rri_status rri_onchip_fallbacks(rri_ctx* ctx, int64_t* fallbacks);
/* diagnostics: the XCD (XCC_ID) each of `count` workgroups of a launch on the handle's stream lands on */
rri_status rri_debug_xcc(rri_ctx* ctx, int32_t* out, int32_t count);
/* The sweep / objective / stop-rule loop of nmf.py:377-516 for launch-bound sizes, in one call: up to n_sweeps sweeps of the
 * register-resident kernel with the objective of every sweep kept (true_objective, nmf.py:71-94, 488-490) and the rule of
 * nmf.py:510 / optimization.py:284-291 applied ON THE DEVICE after every sweep: the run ends after the first sweep s with
 * |o_s - o_(s-1)| <= stop_scale, where o_(-1) = obj_prev and the caller passes stop_scale = eps_stop |o_0 - o_1| of its run
 * (a negative stop_scale never stops).  *sweeps_done sweeps have run; obj_hist (room for n_sweeps doubles) holds their
 * objectives, NaN where the kernel left none -- the sweep an event interrupted (the call ends with that sweep once the event
 * is resolved and rri_resume has finished it) or a sweep run on the launch-per-phase schedule after a fallback (the call then
 * ends after one sweep): take rri_objective there.  RRI_ERR_UNSUPPORTED when the handle does not take the persistent
 * path (rri_onchip_info) or n_sweeps > 512: call rri_sweep / rri_objective sweep by sweep. */
rri_status rri_sweep_until(rri_ctx* ctx, int32_t n_sweeps, double obj_prev, double stop_scale, double* obj_hist,
                           int32_t* sweeps_done);
/* Stand-alone kernels for roofline measurement (bench.py): R <- R - a b^T fused with the
 * next residual products, on a scratch copy of X, with the handle's own W[:,0], T[0,:] as factors. */
rri_status rri_bench_rank1_update(rri_ctx* ctx, int32_t reps, double* avg_ms);
rri_status rri_bench_stream_copy(rri_ctx* ctx, int32_t reps, double* avg_ms);

#ifdef __cplusplus
}
#endif
#endif /* RRI_HIP_H */
