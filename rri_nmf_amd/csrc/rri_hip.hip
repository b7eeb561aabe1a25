// rri_hip.hip -- librri_hip.so: host side of the C ABI declared in include/rri_hip.h.
//
// Build (see __graft_entry__.build / rri_nmf_amd/build.py):
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -shared -fPIC -Iinclude rri_nmf_amd/csrc/rri_hip.hip
//
// Data layout in HBM (one handle = one nmf() call = one row shard of one GPU).  X (and the mask /
// masked residual) are stored in the handle's dtype (fp32 or fp64); everything else is float64:
//   X     n x LD   row-major, LD = d rounded up to a 16-byte multiple of X's type, pad columns zero
//   Wt    k x n    k-major (W transposed): column t of W is the contiguous row Wt[t,:], so the
//                  streaming pass stages the active column into LDS with coalesced loads
//   T     k x LD   row-major, pad columns zero
//   Ypart npanels x n      per-column-panel partial row dots of the pass
//   Zpart nrb x LD         per-row-block partial column sums of the pass
//   Gpart nwb x (k+2)      per-block partial Gram row / norm / column sum (float64)
//   red   LD + 8 (k+2)     reduced [w^T X | 8 slice sums of (w^T W, ||w||^2, sum W[:,t-1])]  (all-reduce payload)
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <rccl/rccl.h>   // types and prototypes only: the symbols are resolved at run time (rccl_api below), so the
                         // library has no link-time dependency on RCCL and loads on a box without it

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <mutex>
#include <cstdlib>
#include <cstring>
#include <string>
#include <utility>
#include <vector>

#include "rri_hip.h"
#include "rri_kernels.hpp"
#include "rri_wrri_kernels.hpp"
#include "rri_sparse_kernels.hpp"
#include "rri_onchip_kernels.hpp"

using namespace rri;

namespace {

thread_local std::string g_create_error;   // text of the last failed rri_create of this thread

// temporary device buffer that is released on every return path
struct DevTmp {
    void* p = nullptr;
    ~DevTmp() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes); }
};

struct Cursor {
    int sweep, topic, phase;  // phase 0 = T-row half, 1 = W-column half
};

struct TimedLaunch {
    hipEvent_t a, b;
};

}  // namespace

// ---- communicator of a row-sharded run (one process per GPU) ---------------------------------------------------
// RCCL entry points, looked up in the process (a host program that already carries an RCCL -- PyTorch-ROCm ships its
// own, bound to its own HIP runtime -- must be the one used: two HIP runtimes in one process do not share streams) and
// only then in librccl.so.1.  RRI_RCCL_LIB names another file.
namespace {
struct RcclApi {
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclBroadcast) Broadcast = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclCommAbort) CommAbort = nullptr;     // optional: what a rank that fails between two matched collectives calls
    bool ok = false;
    std::string err;
};
RcclApi load_rccl() {
    RcclApi a;
    void* h = nullptr;
    auto resolve = [&](void* from) {
        a.GetUniqueId = (decltype(a.GetUniqueId))dlsym(from, "ncclGetUniqueId");
        a.CommInitRank = (decltype(a.CommInitRank))dlsym(from, "ncclCommInitRank");
        a.CommDestroy = (decltype(a.CommDestroy))dlsym(from, "ncclCommDestroy");
        a.AllReduce = (decltype(a.AllReduce))dlsym(from, "ncclAllReduce");
        a.AllGather = (decltype(a.AllGather))dlsym(from, "ncclAllGather");
        a.Broadcast = (decltype(a.Broadcast))dlsym(from, "ncclBroadcast");
        a.GetErrorString = (decltype(a.GetErrorString))dlsym(from, "ncclGetErrorString");
        a.CommAbort = (decltype(a.CommAbort))dlsym(from, "ncclCommAbort");
        return a.GetUniqueId && a.CommInitRank && a.CommDestroy && a.AllReduce && a.AllGather && a.Broadcast &&
               a.GetErrorString;
    };
    if (const char* e = getenv("RRI_RCCL_LIB")) {
        h = dlopen(e, RTLD_NOW | RTLD_GLOBAL);
        if (h && resolve(h)) { a.ok = true; return a; }
        a.err = std::string("RRI_RCCL_LIB=") + e + " could not be used";
        return a;
    }
    if (resolve(RTLD_DEFAULT)) { a.ok = true; return a; }
    for (const char* name : {"librccl.so.1", "librccl.so"}) {
        h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (h && resolve(h)) { a.ok = true; return a; }
    }
    a.err = "RCCL not found (no ncclAllReduce in the process, no librccl.so.1)";
    return a;
}
RcclApi& rccl_api() {
    static RcclApi api = load_rccl();
    return api;
}
}  // namespace

struct rri_comm {
    int rank = 0, world = 1, device = 0;
    ncclComm_t nccl = nullptr;                 // RCCL transport (xGMI), or
    rri_allreduce_fn h_allreduce = nullptr;    // host-callback transport (tests: several ranks on one GPU)
    rri_allgather_fn h_allgather = nullptr;
    rri_broadcast_fn h_broadcast = nullptr;
    void* user = nullptr;
    std::vector<double> hbuf, hbuf2;           // host staging of the callback transport
    long n_allreduce = 0;
    bool aborted = false;                      // a rank left a collective sequence half way: the communicator is unusable
};

struct rri_ctx {
    i64 n = 0, d = 0, LD = 0;
    int k = 0, dtype = RRI_F32, weighted = 0, device = 0;
    size_t es = 4;  // element size of X / mask / residual; all other buffers are float64
    int VN = 4, PW = 1024;
    hipStream_t stream = nullptr;
    bool own_stream = false;

    void *X = nullptr, *M = nullptr, *E = nullptr;
    i64 ldx = 0, ldm = 0;
    bool own_X = false, own_M = false;
    double *W = nullptr, *T = nullptr, *Wprev = nullptr, *Tprev = nullptr;
    double *Ypart = nullptr, *Zpart = nullptr, *red = nullptr, *xraw = nullptr, *Ttpart = nullptr;
    unsigned* Mbits = nullptr;   // bit-packed 0/1 mask (weighted flavour), ldb words per row; NULL = fp mask in M
    i64 ldb = 0;
    double* Qt = nullptr;     // X T^T (k x n), valid while T is fixed
    bool q_valid = false;
    // tile -> XCD rotation of the passes over X (read only) and over the stored residual (read-modify-write): 0 or 1, the faster
    // of the two for THIS handle's buffers (calibrate_rot); RRI_PASS_ROT forces one for all
    int rot_x = 0, rot_r = 0;
    bool rot_done = false;
    bool gfull_valid = false;   // Gfull = T T^T of the current T (k_wsweep_rows)
    double *Gfull = nullptr, *Wsweep0 = nullptr, *wsum_part = nullptr, *wsums = nullptr;   // whole-sweep W half with T fixed: lazily allocated
    double *Y2part = nullptr, *Z2part = nullptr, *dtv = nullptr, *dwv = nullptr, *wold = nullptr, *zeros = nullptr;  // weighted
    i64 ldw = 0;     // row stride of the k-major W (>= n)
    int nsplit = 4;  // column slices of k_tgram
    bool own_red = false;
    i64 red_elems = 0;
    // weighted flavour on a CSR observation pattern (rri_upload_observed_csr): no dense n x d array at all
    bool sparse = false;
    i64 nnz = 0;
    int sp_max_row = 0, kp = 8;
    i64* sp_rowptr = nullptr;   // canonical CSR of the pattern with X on it: residual rebuild, objective, resets
    int* sp_col = nullptr;
    void *sp_x = nullptr, *sp_e = nullptr;
    double* sp_Tt = nullptr;    // T transposed, d x kp
    // the two blocked copies of the residual (rri_sparse_kernels.hpp): [0] rows as segments, cut into column
    // blocks; [1] columns as segments, cut into row blocks
    struct SpCopy {
        int nblk = 1, bw = 1, lps = 64, nwork = 0;
        i64 nseg = 0, gdim = 0, count = 0;   // count: entries incl. the padding of every segment to a multiple of 4
        i64* segptr = nullptr;          // [nblk][nseg + 1]
        unsigned short* idx = nullptr;  // offset inside the block
        void* val = nullptr;
        int* perm = nullptr;            // position in the canonical CSR
        SpWork* work = nullptr;
    } sp[2];
    // dense weighted, one read-modify-write pass per topic step: the column sums a pass leaves for the next T row lack the
    // rank-one term of the W update that follows it; k_wmcorr takes that term from the mask alone (Cpart: its partials)
    double* Cpart = nullptr;
    double* N2part = nullptr;   // [cpart_rows x LD] nw = (w^2)^T M as row-block partials, where k_wmcorr_cols takes it (nw_from_mask)
    unsigned* Mcols = nullptr;  // the packed 0/1 mask once more with the rows in the bits (k_wmcorr_cols), built on first use
    bool nw_mask = false;       // this T-row step's nw was taken by k_wmcorr_cols (N2part), not by the pass (Z2part)
    bool mcols_tried = false;   // ... or found not worth it (dense mask, no memory)
    double mask_density = 1.0;
    int cpart_rows = 0;         // rows allocated in Cpart
    bool wcorr = false;         // the T-row step being enqueued subtracts T[wcorr_topic,:] .* sum_b Cpart[b]
    int wcorr_topic = 0, wcorr_nrb = 0;
    bool resid_fresh = false;   // weighted: E was rebuilt and no half step has run since
    bool dt_pending = false;    // weighted: dtv holds a T-row change that E does not contain yet
    // explicit-residual schedule of the unweighted flavour (RRI_UNWEIGHTED_RESIDUAL): R = X - W T lives in E and takes
    // the two rank-one terms of a topic step inside the pass of the next one
    bool explicit_resid = false;
    bool dw_pending = false;    // dwv holds a W-column change (of topic dw_topic) that R does not contain yet
    int dw_topic = 0;
    double* told = nullptr;     // T[t,:] before its last update (dt = T[t,:] - told while dt_pending)
    double *Gpart = nullptr, *tpart = nullptr, *rowobj = nullptr, *rowpos = nullptr, *normpart = nullptr;
    double *dtmp = nullptr;  // small double scratch (device): [0] sum, ...
    double* objbuf = nullptr;   // W^T W | T T^T | cross terms of the objective assembled after a sweep
    i64* itmp = nullptr;     // small i64 scratch (device)
    i64* tpart_idx = nullptr;
    double *resetT = nullptr, *resetW = nullptr;  // staging for 'random' reset vectors
    DevState* st = nullptr;

    int npanels = 1, rpb = 1, nrb = 1, nwb = 1, ntb = 1;
    int gpart_n = 1;   // rows of Gpart its last writer left (k_wcol: nwb, k_wcol_resid: nwb256, fused pass: nrb)
    int ttpart_n = 0;  // partial vectors T T[t]^T in Ttpart (k_tgram: nsplit)
    int xy_rows = 0;   // blocks per topic in XYpart (stride xy_stride)
    int xy_stride = 1;
    int ntb32 = 1;     // 32-column blocks of k_trow_small
    int tpart_n = 1;   // entries the last T-row step left in tpart (128- or 32-column blocks)

    rri_params prm{};
    bool have_params = false, have_X = false, have_W = false, have_T = false, have_M = false;


    bool carry_valid = false;
    int carry_topic = -1;
    // <w_t, X t_t> of every topic's W half (per k_wcol block): with ||X||^2 and the two Gram matrices it gives the
    // objective without another pass over X.  xy_run = topics 0 .. xy_run-1 of the current sweep have run their
    // W half in order since the last change from outside; xy_valid: all k have, and no T row changed since.
    double* XYpart = nullptr;
    int xy_run = -1;
    bool xy_valid = false, x_sq_valid = false;
    // the persistent sweep left the objective of the sweep that has just ended (minus 1/2 ||X||^2) in DevState.obj_track: pending
    // while the launch is in flight, valid exactly as long as xy_valid is
    bool obj_track_pending = false, obj_track_valid = false;
    double obj_track_value = 0.0;
    // rri_sweep_until: sweeps with the objective history kept and the stop rule applied on the device (persistent sweep only)
    struct { bool active = false; double prev = 0.0, scale = 0.0; int n = 0; double* out = nullptr; } until;
    double *objhist = nullptr, *objdec = nullptr;
    double x_sq = 0.0;
    bool pending_wcheck = false;
    int pending_wcheck_topic = -1;

    // interrupted run
    bool paused = false;
    int run_total = 0;
    Cursor resume_at{0, 0, 0};
    bool resume_done = false;
    rri_event pending{RRI_EVENT_NONE, -1, 0, 0};

    bool resid_valid = false;  // masked residual E is in sync with (W,T) (weighted flavour)
    bool skip_row_finish = false;  // the resumed W half must not re-run the T-row checks (stale partial sums)
    int nwb256 = 1;

    // row-sharded run: this handle holds rows [row_offset, row_offset + n) of an n_global-row problem and every
    // cross-row sum of the schedule is all-reduced over `comm` on the handle's stream (rri_attach_comm)
    rri_comm* comm = nullptr;
    i64 row_offset = 0, n_global = 0;
    double* ctail = nullptr;   // 8 doubles (device): small collectives (column verdicts, objective parts)
    double* cand = nullptr;    // 2 * world doubles (device): candidates of the max-residual reset
    rri_status comm_status = RRI_OK;   // first failure of a collective inside an enqueued sequence

    // register-resident sweeps (rri_onchip_kernels.hpp): per-workgroup partial arrays and the grid barrier's counter
    int n_cu = 0;
    double *mkZ = nullptr, *mkG = nullptr, *mkP = nullptr, *mkX = nullptr, *mkT = nullptr, *objE = nullptr;
    unsigned* mkbar = nullptr;
    long onchip_launches = 0;
    // a persistent launch whose workgroups could not synchronise (HALT_ERR_GRID_SYNC: the device was shared, not every
    // workgroup resident in time) is undone and its range of steps run on the launch-per-phase schedule instead:
    // W, T as they were before the launch, the host flags the launch-per-phase schedule would have started from
    double *Wsafe = nullptr, *Tsafe = nullptr;
    bool onchip_in_flight = false;      // the sequence just enqueued was a persistent launch (run_and_collect reads it)
    bool onchip_off = false;            // after a fallback the handle stays on the launch-per-phase schedule ...
    long long onchip_off_until = 0;     // ... until this time (steady clock, ns): 2 s after the first fallback, doubling up to 64 s
    bool onchip_saved_skip = false;
    long onchip_fallbacks = 0;

    int timing = 0;            // 0 off, N > 0: time every N-th launch of each kernel id
    long timing_seq[4] = {0, 0, 0, 0};
    std::vector<TimedLaunch> timed[4];
    std::vector<hipEvent_t> event_pool;

    std::string err;
};

namespace {

rri_status fail(rri_ctx* c, rri_status code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) c->err = buf;
    else g_create_error = buf;
    return code;
}

#define HIPCHK(c, call)                                                                          \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess)                                                                    \
            return fail((c), RRI_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
                        __FILE__, __LINE__);                                                     \
    } while (0)

#define CHECK_CTX(c) \
    if (!(c)) return RRI_ERR_INVALID

i64 round_up(i64 a, i64 b) { return (a + b - 1) / b * b; }

// ---- collectives of a row-sharded handle: all on the handle's stream --------------------------------------------
// RCCL: enqueued like a kernel (no host synchronisation).  Host-callback transport: the stream is drained, the
// buffer goes through host memory and the caller's function (tests with several ranks on one GPU).
void comm_fail(rri_ctx* c, rri_status code, const char* what, const char* detail) {
    if (c->comm_status == RRI_OK) {
        c->comm_status = code;
        c->err = std::string("collective failed: ") + what + ": " + (detail ? detail : "?");
    }
}
// A rank that cannot go on BETWEEN two collectives its peers will still enter (a device error after the all-gather of a
// reset and before its broadcast) must not simply return: the peers would block inside RCCL with nothing to interrupt them.
// It aborts the communicator (ncclCommAbort: pending and later collectives of every rank end with an error) and reports.
// The host-callback transport has no such call; its collectives are host functions of the caller, who owns their time-outs.
rri_status comm_abort(rri_ctx* c, const char* what, hipError_t e) {
    rri_comm* m = c->comm;
    if (m && m->nccl && !m->aborted && rccl_api().CommAbort) {
        (void)rccl_api().CommAbort(m->nccl);
        m->nccl = nullptr;
    }
    if (m) m->aborted = true;
    c->comm_status = RRI_OK;
    return fail(c, RRI_ERR_COMM, "%s failed between two collectives (%s): the communicator was aborted so that the other ranks do not wait", what, hipGetErrorString(e));
}
void comm_allreduce(rri_ctx* c, double* dev, i64 count) {
    rri_comm* m = c->comm;
    if (m && m->aborted) { comm_fail(c, RRI_ERR_COMM, "all-reduce", "the communicator was aborted"); return; }
    if (!m || c->comm_status != RRI_OK) return;
    m->n_allreduce += 1;
    if (m->nccl) {
        ncclResult_t r = rccl_api().AllReduce(dev, dev, (size_t)count, ncclDouble, ncclSum, m->nccl, c->stream);
        if (r != ncclSuccess) comm_fail(c, RRI_ERR_COMM, "ncclAllReduce", rccl_api().GetErrorString(r));
        return;
    }
    m->hbuf.resize((size_t)count);
    hipError_t e = hipMemcpyAsync(m->hbuf.data(), dev, (size_t)count * 8, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) { comm_fail(c, RRI_ERR_HIP, "staging", hipGetErrorString(e)); return; }
    if (m->h_allreduce(m->user, m->hbuf.data(), count) != 0) { comm_fail(c, RRI_ERR_COMM, "all-reduce callback", "non-zero return"); return; }
    e = hipMemcpyAsync(dev, m->hbuf.data(), (size_t)count * 8, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);   // hbuf is reused by the next collective
    if (e != hipSuccess) comm_fail(c, RRI_ERR_HIP, "staging", hipGetErrorString(e));
}
void comm_allgather(rri_ctx* c, const double* dev_send, i64 count, double* dev_recv) {
    rri_comm* m = c->comm;
    if (m && m->aborted) { comm_fail(c, RRI_ERR_COMM, "all-gather", "the communicator was aborted"); return; }
    if (!m || c->comm_status != RRI_OK) return;
    if (m->nccl) {
        ncclResult_t r = rccl_api().AllGather(dev_send, dev_recv, (size_t)count, ncclDouble, m->nccl, c->stream);
        if (r != ncclSuccess) comm_fail(c, RRI_ERR_COMM, "ncclAllGather", rccl_api().GetErrorString(r));
        return;
    }
    m->hbuf.resize((size_t)count);
    m->hbuf2.resize((size_t)count * m->world);
    hipError_t e = hipMemcpyAsync(m->hbuf.data(), dev_send, (size_t)count * 8, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) { comm_fail(c, RRI_ERR_HIP, "staging", hipGetErrorString(e)); return; }
    if (m->h_allgather(m->user, m->hbuf.data(), count, m->hbuf2.data()) != 0) { comm_fail(c, RRI_ERR_COMM, "all-gather callback", "non-zero return"); return; }
    e = hipMemcpyAsync(dev_recv, m->hbuf2.data(), (size_t)count * m->world * 8, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) comm_fail(c, RRI_ERR_HIP, "staging", hipGetErrorString(e));
}
void comm_broadcast(rri_ctx* c, double* dev, i64 count, int root) {
    rri_comm* m = c->comm;
    if (m && m->aborted) { comm_fail(c, RRI_ERR_COMM, "broadcast", "the communicator was aborted"); return; }
    if (!m || c->comm_status != RRI_OK) return;
    if (m->nccl) {
        ncclResult_t r = rccl_api().Broadcast(dev, dev, (size_t)count, ncclDouble, root, m->nccl, c->stream);
        if (r != ncclSuccess) comm_fail(c, RRI_ERR_COMM, "ncclBroadcast", rccl_api().GetErrorString(r));
        return;
    }
    m->hbuf.resize((size_t)count);
    hipError_t e = hipMemcpyAsync(m->hbuf.data(), dev, (size_t)count * 8, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) { comm_fail(c, RRI_ERR_HIP, "staging", hipGetErrorString(e)); return; }
    if (m->h_broadcast(m->user, m->hbuf.data(), count, root) != 0) { comm_fail(c, RRI_ERR_COMM, "broadcast callback", "non-zero return"); return; }
    e = hipMemcpyAsync(dev, m->hbuf.data(), (size_t)count * 8, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) comm_fail(c, RRI_ERR_HIP, "staging", hipGetErrorString(e));
}
// `count` host doubles summed over the ranks in place (objective parts, decisions of the host driver)
rri_status comm_allreduce_host(rri_ctx* c, double* host, i64 count) {
    if (!c->comm) return RRI_OK;
    if (count > 8) return fail(c, RRI_ERR_INVALID, "host all-reduce takes at most 8 values");
    HIPCHK(c, hipMemcpyAsync(c->ctail, host, (size_t)count * 8, hipMemcpyHostToDevice, c->stream));
    comm_allreduce(c, c->ctail, count);
    HIPCHK(c, hipMemcpyAsync(host, c->ctail, (size_t)count * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->comm_status != RRI_OK) { rri_status r = c->comm_status; c->comm_status = RRI_OK; return r; }
    return RRI_OK;
}

KParams kparams(const rri_ctx* c) {
    KParams p{};
    const rri_params& q = c->prm;
    p.fix_W = q.fix_W; p.fix_T = q.fix_T; p.project_T = q.project_T_each_iter;
    p.has_trs = q.has_t_row_sum; p.has_wrs = q.has_w_row_sum;
    p.reset_method = q.reset_method; p.resets_left = q.resets_left;
    p.t_row_sum = q.t_row_sum; p.w_row_sum = q.w_row_sum;
    p.reg_w_l1 = q.reg_w_l1; p.reg_w_l2 = q.reg_w_l2; p.reg_t_l1 = q.reg_t_l1; p.reg_t_l2 = q.reg_t_l2;
    p.eps = q.eps_div;
    return p;
}

// Does this handle step through its explicit residual?  Only with both halves free: with T (or W) fixed one half of every
// step is missing, and the Gram form is then the cheaper schedule by far (T fixed: X T^T once, no pass over the matrix per
// topic at all; W fixed: one read pass per topic instead of a read-modify-write), so a handle of the explicit-residual
// schedule takes it for such calls -- fold-in (sklearn_interface.py:327-333, nmf.py:417,460) works on either kind of handle.
// The stored residual is stale afterwards and rebuilt when next needed.
bool resid_sched(const rri_ctx* c) { return c->explicit_resid && !c->prm.fix_T && !c->prm.fix_W && c->k >= 2; }

bool no_regs(const rri_ctx* c) {
    const rri_params& q = c->prm;
    return std::abs(q.reg_w_l1) + std::abs(q.reg_w_l2) + std::abs(q.reg_t_l1) + std::abs(q.reg_t_l2) == 0.0;
}

// ---- timing ------------------------------------------------------------------------------
hipEvent_t get_event(rri_ctx* c) {
    if (!c->event_pool.empty()) {
        hipEvent_t e = c->event_pool.back();
        c->event_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}
struct TimedScope {
    rri_ctx* c;
    int id;
    TimedLaunch tl{nullptr, nullptr};
    bool on;
    TimedScope(rri_ctx* c_, int id_)
        : c(c_), id(id_),
          on(c_->timing > 0 && (c_->timing_seq[id_]++ % c_->timing) == 0 && c_->timed[id_].size() < 400000) {
        if (on) {
            tl.a = get_event(c);
            tl.b = get_event(c);
            (void)hipEventRecord(tl.a, c->stream);
        }
    }
    ~TimedScope() {
        if (on) {
            (void)hipEventRecord(tl.b, c->stream);
            c->timed[id].push_back(tl);
        }
    }
};

// ---- typed launch helpers ------------------------------------------------------------------
// geometry of k_pass, fixed per process (env RRI_PASS_UNROLL / RRI_PASS_NT)
int g_pass_unroll = 8, g_pass_nt = -1, g_pass_rs = 1;   // RS: LDS row sums (needs unroll 8).  nt: -1 = per handle (non-temporal
                                                        // loads where X cannot stay in the caches, plain loads where it can)
int g_pass_dma_sub = 0;       // RRI_PASS_DMA_SUB: row blocks per workgroup of the ring kernel (0: ~1024 workgroups)
int g_pass_dma = 0;           // RRI_PASS_DMA=1: the read-only pass through the LDS-DMA ring (k_pass_dma).  OFF by default: the same bits as
                              // k_pass, but between +3 % and -9 % in time by box, process and geometry (profiles/r04_pass_dma_ab.log)
int g_pass_unroll_upd = 16;   // rows in flight of the read-modify-write passes (RRI_PASS_UNROLL set: follows it)
int g_wpass_uc = 8;      // RRI_WPASS_UC: rows in flight of the writing weighted pass with a bit-packed mask: 8 (one mask word per
                         // chunk; +1.3 % at C5 over 4, 16 falls to one wave per SIMD: profiles/r02_weighted_pass_variants.log) or 4
int g_wpass_ud = 4;      // RRI_WPASS_UD: rows in flight of the one-pass weighted step.  4 (DPP row sums, 138 VGPRs, 3 waves per SIMD): 1.458 ms at
                         // BASELINE config 5 against 1.553 ms for 8 (LDS row sums, 240 VGPRs, 2 waves per SIMD) and 1.618 ms for 8 with
                         // DPP row sums, engines made alternately in one process (profiles/r04_wpass_one_variants.log)
int g_wpass_occ4 = 1;    // RRI_WPASS_OCC4=0: the one-pass step at the compiler's own register count (130: 3 waves per SIMD)
int g_wnw_mask = 1;      // RRI_WNW_MASK=0: the one-pass step keeps taking nw = (w^2)^T M in the read-modify-write pass on a sparse 0/1 mask too
int g_wmcorr_wgs = 16;   // RRI_WMCORR_WGS: workgroups per CU of k_wmcorr_cols (4: 60 us, 8: 48, 16: 46, 32: 44 at BASELINE config 5)
int g_wmcorr_cols = 1;   // RRI_WMCORR_COLS=0: the mask-only correction always walks every bit (k_wmcorr), also on a sparse mask
int g_wmcorr_skip = 1;   // RRI_WMCORR_SKIP=0: k_wmcorr does not test the row factors for zero
int g_wpass_one = 1;     // RRI_WPASS_ONE=0: the dense weighted flavour in two passes per topic step (read; read-modify-write), as rounds 1-3
int g_wpass_il = -1;     // RRI_WPASS_IL: 1 / 0 = interleaved / contiguous row chunks in every weighted pass; default: the writing ones
int g_side_jobs = 1;    // RRI_SIDE_JOBS=0: every small job as a launch of its own
int g_onchip = 1;       // RRI_ONCHIP=0: never the register-resident persistent sweep (rri_onchip_kernels.hpp)
int g_onchip_obj = 1;   // RRI_ONCHIP_OBJ=0: the persistent sweep does not leave the objective of its last sweep (rri_objective takes the Gram kernels)
int g_resid_mfma = 1;   // RRI_RESID_MFMA=0: the residual on the vector ALU for every k
int g_resid_split = 0;   // RRI_RESID_SPLIT=n: column ranges per row block of the residual rebuild (0: chosen from the grid)
int g_wsweep = 1;        // RRI_WSWEEP=0: runs with T fixed take the launch-per-topic W half (k_tgram, k_wcol, k_check_wcol per topic)
int g_trow_small = 1;    // RRI_TROW_SMALL=0: k_reduce + k_trow_numer as two launches at every size
int g_pass_interleave = -1;  // RRI_PASS_IL: 1 / 0 = interleaved / contiguous row chunks per workgroup of k_pass; default:
                             // interleaved up to 1024 workgroups (+2 % at 20000 x 5000; -1 % at C3, where it stays off)
int g_pass_rot = -1;     // RRI_PASS_ROT=0..7: rotate the tiles of the passes inside every group of 8 workgroups (another XCD per tile) by this much
                         // for every handle; unset: each handle's own calibrated 0 or 1 (calibrate_rot)
int g_rot_cal = 1;       // RRI_ROT_CAL=0: no calibration, rotation 0
int g_rmw_shop = 1;      // RRI_RMW_SHOP=n (diagnostics): places a large residual buffer is tried in (calibrate_rot); 1: the first allocation is kept
int g_obj_direct = 0;   // RRI_OBJ_DIRECT=1: the objective always through the residual (k_resid)
// kernels that touch X / mask / residual depend on the storage type SX; the rest is float64
template <typename SX>
struct LaunchX {
    typedef SX Elem;
    // row-dot slots of the 4 waves, the active W column, (UPD: one or two arrays of rank-one row factors,) the row-sum tiles
    static size_t pass_shmem(const rri_ctx* c, int upd) {
        return ((5 + upd) * (size_t)c->rpb + 4 * 8 * 72) * sizeof(double);
    }
    // the rank-one terms a pass folds into the residual before it takes its products (UPD = 1: a, b; UPD = 2: also a2 and
    // b2 - b2sub)
    struct Upd {
        const double *a = nullptr, *b = nullptr, *a2 = nullptr, *b2 = nullptr, *b2sub = nullptr;
    };
    // Non-temporal loads keep a streamed X from washing W, T and the partial sums out of the caches -- where X cannot
    // stay there anyway.  A matrix that fits the 256 MB Infinity Cache is re-read from it by every pass: plain loads are
    // 4 % faster at 10000 x 1000 (2020 against 1937 sweeps/s, profiles/r02_c2_variants.log).
    static bool pass_nt(const rri_ctx* c) {
        if (g_pass_nt >= 0) return g_pass_nt != 0;
        return (double)c->n * (double)c->LD * (double)c->es > 192.0e6;
    }
    template <bool DO_Y, bool DO_Z, int UPD, int U, bool NT, bool RS>
    static void pass_k(rri_ctx* c, void* Xp, i64 ldp, const double* trow, const double* wc, const Upd& u, const TgramJob& job) {
        const int ncols = (int)std::min<i64>(ldp, c->LD);
        typedef typename std::conditional<(UPD > 0), SX, const SX>::type XT;
        hipLaunchKernelGGL((k_pass<SX, DO_Y, DO_Z, UPD, U, NT, RS>), dim3(c->npanels * c->nrb + job.nblocks), dim3(256),
                           pass_shmem(c, UPD), c->stream, (XT*)Xp, ldp, (int)c->n, ncols, trow, wc, c->Ypart,
                           c->Zpart, c->LD, c->rpb, c->npanels, u.a, u.b, u.a2, u.b2, u.b2sub, (const DevState*)c->st, job,
                           // interleaved row chunks: the workgroups running at one time walk ONE window of the matrix, as a
                           // linear stream does.  Read-only pass: +2 % up to 1024 workgroups, nothing at C3.  Read-modify-write
                           // (UPD): +4-9 % at C3 (profiles/r02_residual_schedule_geometry.log) -- reads and writes of a window
                           // stay in the DRAM pages that are open
                           ((g_pass_interleave == 1 || (g_pass_interleave < 0 && (c->npanels * c->nrb <= 1024 || UPD > 0))) ? c->nrb : 0) |
                               ((g_pass_rot >= 0 ? g_pass_rot : (UPD > 0 ? c->rot_r : c->rot_x)) << 27));
    }
    // the read-only pass through the LDS-DMA ring (k_pass_dma), opt-in (RRI_PASS_DMA=1): where the ring and the row-dot slots fit
    // the LDS and X's rows are 16-byte aligned
    // A workgroup of the ring kernel walks `sub` consecutive row blocks of the handle's geometry as ONE stream (the ring stays full
    // across them) and still leaves one row of column sums per row block: Ypart / Zpart -- and every sum in them -- are exactly
    // what k_pass leaves.  (The hope was ~1024 long workgroups: measured 0.663 against 0.661 ms for the register kernel with two
    // blocks per workgroup, 0.72 with three, 0.671 with four -- profiles/r04_pass_dma_ab.log.)  The row-dot slots of the sub blocks
    // must fit the LDS next to the 64 KiB ring.
    static int pass_dma_sub(const rri_ctx* c, bool interleaved) {
        if (interleaved) return 1;
        if (g_pass_dma_sub > 0) return std::max(1, std::min(g_pass_dma_sub, (int)(2200 / std::max(c->rpb, 1))));
        const int want = (int)std::max<i64>(1, ((i64)c->npanels * c->nrb + 512) / 1024);
        return std::max(1, std::min(want, 2200 / std::max(c->rpb, 1)));
    }
    static size_t pass_dma_shmem(const rri_ctx* c, int sub) { return (size_t)4 * PASS_DMA_SLOTS * 1024 + 5 * (size_t)sub * c->rpb * sizeof(double); }
    static bool pass_dma(const rri_ctx* c, i64 ldp) {
        if (g_pass_dma == 0 || g_pass_unroll != 8 || !g_pass_rs) return false;
        if (pass_dma_shmem(c, 1) > 150 * 1024 || ldp % c->VN != 0 || c->rpb % PASS_DMA_CHUNK != 0) return false;
        return g_pass_dma == 1;
    }
    template <bool DO_Y, bool DO_Z, bool NT>
    static void pass_dma_k(rri_ctx* c, const void* Xp, i64 ldp, const double* trow, const double* wc, const TgramJob& job) {
        static bool attr_set[64] = {};   // per instantiation and device
        const int dv = c->device & 63;
        if (!attr_set[dv]) {
            // (the kernel also has 320 bytes of static LDS for its side job: dynamic + static must stay within the CU's 160 KiB)
            (void)hipFuncSetAttribute((const void*)k_pass_dma<SX, DO_Y, DO_Z, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);
            attr_set[dv] = true;
        }
        const int ncols = (int)std::min<i64>(ldp, c->LD);
        const bool il = g_pass_interleave == 1 || (g_pass_interleave < 0 && c->npanels * c->nrb <= 1024);
        const int sub = pass_dma_sub(c, il);
        const int ngroups = (c->nrb + sub - 1) / sub;
        hipLaunchKernelGGL((k_pass_dma<SX, DO_Y, DO_Z, NT>), dim3(c->npanels * ngroups + job.nblocks), dim3(256), pass_dma_shmem(c, sub),
                           c->stream, (const SX*)Xp, ldp, (int)c->n, ncols, trow, wc, c->Ypart, c->Zpart, c->LD, c->rpb,
                           c->npanels, (const DevState*)c->st, job, il ? c->nrb : 0, sub, c->nrb);
    }
    template <bool DO_Y, bool DO_Z, int UPD>
    static void pass_cfg(rri_ctx* c, void* Xp, i64 ldp, const double* trow, const double* wc, const Upd& u = Upd{},
                         const TgramJob& job = TgramJob{}) {
        if constexpr (UPD == 0) {
            if (pass_dma(c, ldp) && ((uintptr_t)Xp) % 16 == 0) {
                if (pass_nt(c)) pass_dma_k<DO_Y, DO_Z, true>(c, Xp, ldp, trow, wc, job);
                else pass_dma_k<DO_Y, DO_Z, false>(c, Xp, ldp, trow, wc, job);
                return;
            }
        }
        if (UPD > 0 && g_pass_unroll_upd == 16) {
            // the read-modify-write variants: 16 rows in flight per wave, row dots by DPP wave sums -- 0.665 against 0.639
            // of 8 TB/s for the 8-row LDS row-sum variant at C3 (profiles/r02_residual_schedule_geometry.log)
            if (pass_nt(c)) pass_k<DO_Y, DO_Z, UPD, 16, true, false>(c, Xp, ldp, trow, wc, u, job);
            else pass_k<DO_Y, DO_Z, UPD, 16, false, false>(c, Xp, ldp, trow, wc, u, job);
            return;
        }
        if (g_pass_unroll == 8 && g_pass_rs && DO_Y) {
            if (pass_nt(c)) pass_k<DO_Y, DO_Z, UPD, 8, true, true>(c, Xp, ldp, trow, wc, u, job);
            else pass_k<DO_Y, DO_Z, UPD, 8, false, true>(c, Xp, ldp, trow, wc, u, job);
            return;
        }
        const int key = g_pass_unroll * 2 + (pass_nt(c) ? 1 : 0);
        switch (key) {
#define RRI_CASE(U_)                                                                              \
    case U_ * 2 + 0: pass_k<DO_Y, DO_Z, UPD, U_, false, false>(c, Xp, ldp, trow, wc, u, job); break;   \
    case U_ * 2 + 1: pass_k<DO_Y, DO_Z, UPD, U_, true, false>(c, Xp, ldp, trow, wc, u, job); break;
            RRI_CASE(4)
            RRI_CASE(8)
            RRI_CASE(16)
#undef RRI_CASE
            default: pass_k<DO_Y, DO_Z, UPD, 8, true, false>(c, Xp, ldp, trow, wc, u, job);
        }
    }
    // row dots against T[t,:] (DO_Y) and column sums against W[:,tz] (DO_Z); `job`: the Gram row of T[t,:] rides along
    template <bool DO_Y, bool DO_Z>
    static void pass(rri_ctx* c, int t, int tz, const TgramJob& job = TgramJob{}) {
        TimedScope ts(c, 0);
        pass_cfg<DO_Y, DO_Z, 0>(c, c->X, c->ldx, c->T + (i64)t * c->LD, c->W + (i64)tz * c->ldw, Upd{}, job);
    }
    // explicit-residual schedule: the same products over the stored residual R (c->E, stride LD)
    static void rpass_colsums(rri_ctx* c, int tz) {
        TimedScope ts(c, 0);
        pass_cfg<false, true, 0>(c, c->E, c->LD, nullptr, c->W + (i64)tz * c->ldw);
    }
    // R <- R - a b^T [- a2 (b2 - b2sub)^T] fused with the row dots (against trow) and column sums (against wc) of
    // the new R: the rank-one residual update north_star names
    static void rank_update(rri_ctx* c, void* R, i64 ldr, const Upd& u, const double* trow, const double* wc,
                            const TgramJob& job = TgramJob{}) {
        TimedScope ts(c, 3);
        if (u.a2) pass_cfg<true, true, 2>(c, R, ldr, trow, wc, u, job);
        else pass_cfg<true, true, 1>(c, R, ldr, trow, wc, u, job);
    }
    template <bool DO_Y, bool DO_Z, bool UPD2, bool WRITE, bool MBITS, int U, bool RS>
    static void wpass_k(rri_ctx* c, const double* trow, const double* wc, const double* a1, const double* b1,
                        const double* a2, const double* b2) {
        if constexpr (DO_Z && MBITS) {      // a sparse 0/1 mask: nw comes from k_wmcorr_cols, the pass leaves Z2part alone
            if (nw_from_mask(c)) { wpass_k2<DO_Y, DO_Z, UPD2, WRITE, MBITS, U, RS, false>(c, trow, wc, a1, b1, a2, b2); return; }
        }
        wpass_k2<DO_Y, DO_Z, UPD2, WRITE, MBITS, U, RS, DO_Z>(c, trow, wc, a1, b1, a2, b2);
    }
    template <bool DO_Y, bool DO_Z, bool UPD2, bool WRITE, bool MBITS, int U, bool RS, bool DO_Z2>
    static void wpass_k2(rri_ctx* c, const double* trow, const double* wc, const double* a1, const double* b1,
                         const double* a2, const double* b2) {
        const int ncols = (int)std::min<i64>(c->ldx, c->LD);
        if constexpr (DO_Y && WRITE && U == 4) {
            if (g_wpass_occ4 && (DO_Z2 || !DO_Z)) {     // the one-pass step, 4 rows in flight: the build for four waves per SIMD
                hipLaunchKernelGGL((k_wpass_occ4<SX, DO_Y, DO_Z, UPD2, WRITE, U, true, MBITS, RS, DO_Z2>), dim3(c->npanels * c->nrb),
                                   dim3(256), (11 * (size_t)c->rpb + 4 * 8 * 72) * sizeof(double), c->stream, (SX*)c->E,
                                   (const SX*)c->M, c->LD, c->ldm, (const unsigned*)c->Mbits, c->ldb, (int)c->n, ncols, trow,
                                   wc, a1, b1, a2, b2, c->Ypart, c->Y2part, c->Zpart, c->Z2part, c->LD, c->rpb, c->npanels,
                                   (const DevState*)c->st, ((g_wpass_il == 1 || g_wpass_il < 0) ? c->nrb : 0) | ((g_pass_rot >= 0 ? g_pass_rot : c->rot_r) << 27));
                return;
            }
        }
        hipLaunchKernelGGL((k_wpass<SX, DO_Y, DO_Z, UPD2, WRITE, U, true, MBITS, RS, DO_Z2>), dim3(c->npanels * c->nrb),
                           dim3(256), (11 * (size_t)c->rpb + 4 * 8 * 72) * sizeof(double), c->stream, (SX*)c->E,
                           (const SX*)c->M, c->LD, c->ldm, (const unsigned*)c->Mbits, c->ldb, (int)c->n, ncols, trow,
                           wc, a1, b1, a2, b2, c->Ypart, c->Y2part, c->Zpart, c->Z2part, c->LD, c->rpb, c->npanels,
                           (const DevState*)c->st,
                           // interleaved row chunks for the passes that write E back (read-modify-write), as for k_pass<UPD>
                           ((g_wpass_il == 1 || (g_wpass_il < 0 && (WRITE || c->npanels * c->nrb <= 1024))) ? c->nrb : 0) |
                               ((g_pass_rot >= 0 ? g_pass_rot : c->rot_r) << 27));
    }
    template <bool DO_Y, bool DO_Z, bool UPD2, bool WRITE>
    static void wpass(rri_ctx* c, const double* trow, const double* wc, const double* a1, const double* b1,
                      const double* a2, const double* b2) {
        TimedScope ts(c, 3);
        if (c->sparse) { sp_wpass<DO_Y, DO_Z, UPD2, WRITE>(c, trow, wc, a1, b1, a2, b2); return; }
        // rows in flight: 8 for the passes that take row products (they use the LDS row sums); the writing pass (two
        // rank-one corrections, two sets of accumulators): 4 with an fp mask array, 8 with a bit-packed mask
        constexpr int U = DO_Y ? 8 : 4;
        const bool rs = DO_Y && g_pass_rs;
        if constexpr (DO_Y && WRITE) {      // the one-pass step: RRI_WPASS_UD=4 -> 4 rows in flight, DPP row sums (fewer registers)
            if (g_wpass_ud == 4) {
                if (c->Mbits) wpass_k<DO_Y, DO_Z, UPD2, WRITE, true, 4, false>(c, trow, wc, a1, b1, a2, b2);
                else wpass_k<DO_Y, DO_Z, UPD2, WRITE, false, 4, false>(c, trow, wc, a1, b1, a2, b2);
                return;
            }
        }
        if (c->Mbits) {
            if (rs) wpass_k<DO_Y, DO_Z, UPD2, WRITE, true, 8, DO_Y>(c, trow, wc, a1, b1, a2, b2);
            else if (!DO_Y && g_wpass_uc == 8) wpass_k<DO_Y, DO_Z, UPD2, WRITE, true, 8, false>(c, trow, wc, a1, b1, a2, b2);
            else wpass_k<DO_Y, DO_Z, UPD2, WRITE, true, U, false>(c, trow, wc, a1, b1, a2, b2);
        } else {
            if (rs) wpass_k<DO_Y, DO_Z, UPD2, WRITE, false, 8, DO_Y>(c, trow, wc, a1, b1, a2, b2);
            else wpass_k<DO_Y, DO_Z, UPD2, WRITE, false, U, false>(c, trow, wc, a1, b1, a2, b2);
        }
    }
    // c = M^T (wn .* dw) as row-block partials in Cpart (k_wmcorr): the correction of the column sums a one-pass topic step
    // leaves behind.  Geometry: ~4 workgroups per CU, row blocks of a multiple of 64 rows, at most 4096 (32 KiB of LDS).
    // the column-major copy of the packed mask, on first use (one-off: a kernel, a count, one synchronisation)
    static bool mask_cols(rri_ctx* c) {
        if (!c->Mbits || !g_wmcorr_cols) return false;
        if (!c->mcols_tried) {
            c->mcols_tried = true;
            const i64 ng = (c->n + 31) / 32;
            unsigned long long* cnt = nullptr;
            if (hipMalloc((void**)&c->Mcols, (size_t)ng * c->LD * sizeof(unsigned)) != hipSuccess) { c->Mcols = nullptr; (void)hipGetLastError(); return false; }
            if (hipMalloc((void**)&cnt, sizeof(unsigned long long)) != hipSuccess) { (void)hipFree(c->Mcols); c->Mcols = nullptr; (void)hipGetLastError(); return false; }
            (void)hipMemsetAsync(cnt, 0, sizeof(unsigned long long), c->stream);
            hipLaunchKernelGGL(k_mask_cols_from_bits, dim3(4096), dim3(256), 0, c->stream, (const unsigned*)c->Mbits, c->ldb, c->n, c->Mcols,
                               c->LD, cnt);
            unsigned long long h = 0;
            hipError_t e = hipMemcpyAsync(&h, cnt, sizeof h, hipMemcpyDeviceToHost, c->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
            (void)hipFree(cnt);
            if (e != hipSuccess) { (void)hipFree(c->Mcols); c->Mcols = nullptr; return false; }
            c->mask_density = (double)h / ((double)c->n * (double)c->d);
            if (c->mask_density > 0.12) { (void)hipFree(c->Mcols); c->Mcols = nullptr; }     // the dense-bit kernel is the cheaper one there
        }
        return c->Mcols != nullptr;
    }
    // the second column sum of a T-row step, nw = (w^2)^T M, from the mask-only kernel instead of the pass: dense handles on the
    // one-pass schedule whose mask is 0/1 and sparse enough for the column-major copy
    static bool nw_from_mask(rri_ctx* c) { return !c->sparse && g_wpass_one && g_wnw_mask && mask_cols(c); }
    // dw == NULL: nothing pending, nw alone (nw_from_mask handles)
    static void wmcorr(rri_ctx* c, const double* wn, const double* dw) {
        TimedScope ts(c, 2);
        if (mask_cols(c)) {     // a sparse 0/1 mask: the set bits only
            const bool nw = nw_from_mask(c);
            const int npg = (int)((c->LD + 255) / 256);
            i64 nrb = std::min<i64>(256, std::max<i64>(1, (g_wmcorr_wgs * (i64)std::max(c->n_cu, 1) + npg - 1) / npg));
            i64 rpb = std::min<i64>(2048, round_up((c->n + nrb - 1) / nrb, 32));
            nrb = (c->n + rpb - 1) / rpb;            // <= cpart_rows (256, or n / 2048 where that is more: rri_create)
            c->wcorr_nrb = (int)nrb;
            const dim3 grid((unsigned)(npg * nrb));
            const size_t sh = (size_t)rpb * sizeof(double) * ((dw && nw) ? 2 : 1);
#define RRI_WMC(HD, NW_) hipLaunchKernelGGL((k_wmcorr_cols<HD, NW_>), grid, dim3(256), sh, c->stream, (const unsigned*)c->Mcols, c->LD, \
                                            (int)c->n, (int)c->LD, wn, dw, c->Cpart, c->N2part, c->LD, (int)rpb, npg, (const DevState*)c->st)
            if (dw && nw) RRI_WMC(true, true);
            else if (dw) RRI_WMC(true, false);
            else if (nw) RRI_WMC(false, true);
#undef RRI_WMC
            return;
        }
        if (!dw) return;
        const bool bits = c->Mbits != nullptr;
        const int npg = bits ? (int)((c->ldb + 255) / 256) : c->npanels;
        i64 nrb = std::min<i64>(256, std::max<i64>(1, (4 * (i64)std::max(c->n_cu, 1) + npg - 1) / npg));
        i64 rpb = std::min<i64>(4096, round_up((c->n + nrb - 1) / nrb, 64));
        nrb = (c->n + rpb - 1) / rpb;
        c->wcorr_nrb = (int)std::min<i64>(nrb, c->cpart_rows);     // (cpart_rows covers every n: see rri_create)
        const int ncols = (int)std::min<i64>(bits ? c->ldb * 4 : c->ldm, c->LD);
        if (bits && g_wmcorr_skip)
            hipLaunchKernelGGL((k_wmcorr<SX, true, true>), dim3((unsigned)(npg * nrb)), dim3(256), (size_t)rpb * sizeof(double), c->stream,
                               (const SX*)nullptr, (i64)0, (const unsigned*)c->Mbits, c->ldb, (int)c->n, (int)(c->ldb * 4), wn, dw,
                               c->Cpart, c->LD, (int)rpb, npg, (const DevState*)c->st);
        else if (bits)
            hipLaunchKernelGGL((k_wmcorr<SX, true, false>), dim3((unsigned)(npg * nrb)), dim3(256), (size_t)rpb * sizeof(double), c->stream,
                               (const SX*)nullptr, (i64)0, (const unsigned*)c->Mbits, c->ldb, (int)c->n, (int)(c->ldb * 4), wn, dw,
                               c->Cpart, c->LD, (int)rpb, npg, (const DevState*)c->st);
        else
            hipLaunchKernelGGL((k_wmcorr<SX, false, false>), dim3((unsigned)(npg * nrb)), dim3(256), (size_t)rpb * sizeof(double), c->stream,
                               (const SX*)c->M, c->ldm, (const unsigned*)nullptr, (i64)0, (int)c->n, ncols, wn, dw, c->Cpart,
                               c->LD, (int)rpb, npg, (const DevState*)c->st);
    }
    // ---- sparse pattern (rri_sparse_kernels.hpp) ----------------------------------------------------
    template <bool DO_S, bool UPD2, bool WRITE, int LPS>
    static void sp_blk_k(rri_ctx* c, const rri_ctx::SpCopy& cp, const double* B1, const double* B2, const double* V,
                         const double* A1, const double* A2, double* S1, double* S2, i64 lds) {
        const size_t sh = sp_lds_bytes<SX>(cp.bw);
        static bool attr_set[64] = {};   // per instantiation and device (the attribute belongs to the device's code object)
        const int dv = c->device & 63;
        if (!attr_set[dv]) {
            (void)hipFuncSetAttribute((const void*)k_sp_blk<SX, DO_S, UPD2, WRITE, LPS>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, SP_BLOCK_BYTES + 64);
            attr_set[dv] = true;
        }
        if (cp.nwork < 1) return;
        hipLaunchKernelGGL((k_sp_blk<SX, DO_S, UPD2, WRITE, LPS>), dim3(cp.nwork), dim3(1024), sh, c->stream,
                           (const SpWork*)cp.work, (const i64*)cp.segptr, cp.nseg, (const unsigned short*)cp.idx,
                           (SX*)cp.val, cp.bw, cp.gdim, B1, B2, V, A1, A2, S1, S2, lds, (const DevState*)c->st);
    }
    template <bool DO_S, bool UPD2, bool WRITE>
    static void sp_blk(rri_ctx* c, int which, const double* B1, const double* B2, const double* V, const double* A1,
                       const double* A2, double* S1, double* S2, i64 lds) {
        const rri_ctx::SpCopy& cp = c->sp[which];
        switch (cp.lps) {
            case 8: sp_blk_k<DO_S, UPD2, WRITE, 8>(c, cp, B1, B2, V, A1, A2, S1, S2, lds); break;
            case 16: sp_blk_k<DO_S, UPD2, WRITE, 16>(c, cp, B1, B2, V, A1, A2, S1, S2, lds); break;
            case 32: sp_blk_k<DO_S, UPD2, WRITE, 32>(c, cp, B1, B2, V, A1, A2, S1, S2, lds); break;
            default: sp_blk_k<DO_S, UPD2, WRITE, 64>(c, cp, B1, B2, V, A1, A2, S1, S2, lds); break;
        }
    }
    // the same operation as wpass on the two copies of the pattern residual: row products from the row copy
    // (one Ypart "panel" per column block), column sums from the column copy (one Zpart row per row block)
    template <bool DO_Y, bool DO_Z, bool UPD2, bool WRITE>
    static void sp_wpass(rri_ctx* c, const double* trow, const double* wc, const double* a1, const double* b1,
                         const double* a2, const double* b2) {
        if (DO_Y || WRITE)   // segments = rows (factors a1, a2), gathered = columns (b1, b2, trow)
            sp_blk<DO_Y, UPD2, WRITE>(c, 0, b1, b2, trow, a1, a2, c->Ypart, c->Y2part, c->n);
        if (DO_Z || WRITE)   // segments = columns (b1, b2), gathered = rows (a1, a2, wc)
            sp_blk<DO_Z, UPD2, WRITE>(c, 1, a1, a2, wc, b1, b2, c->Zpart, c->Z2part, c->LD);
    }
    // out (nseg x m, row-major, device) = the X on the pattern (which = 0) or its transpose (1) times B (gdim x m)
    static void sp_spmm(rri_ctx* c, int which, const double* B, int m, double* part, double* out) {
        const rri_ctx::SpCopy& cp = c->sp[which];
        const i64 waves = (i64)cp.nblk * cp.nseg;
        hipLaunchKernelGGL((k_sp_spmm<SX>), dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, c->stream,
                           (const i64*)cp.segptr, cp.nseg, cp.nblk, (const unsigned short*)cp.idx, (const int*)cp.perm,
                           (const SX*)c->sp_x, cp.bw, B, m, part);
        const i64 count = cp.nseg * m;
        hipLaunchKernelGGL(k_sp_sum_blocks, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, c->stream,
                           (const double*)part, count, cp.nblk, out);
    }
    static void sp_resid(rri_ctx* c, bool write_e, double* rowobj, double* rowpos) {
        const i64 total = (i64)c->k * c->d;
        hipLaunchKernelGGL((k_convert2d<double, double, true>), dim3((unsigned)std::min<i64>(4096, (total + 255) / 256)),
                           dim3(256), 0, c->stream, (const double*)c->T, c->LD, c->sp_Tt, (i64)c->kp, (i64)c->k, c->d);
        hipLaunchKernelGGL((k_sp_resid<SX>), dim3((unsigned)((c->n + 3) / 4)), dim3(256), 4 * (size_t)c->kp * sizeof(double),
                           c->stream, (const i64*)c->sp_rowptr, (const int*)c->sp_col, (const SX*)c->sp_x, c->n,
                           (const double*)c->W, c->ldw, (const double*)c->sp_Tt, c->k, c->kp,
                           write_e ? (SX*)c->sp_e : (SX*)nullptr, rowobj, rowpos);
        if (write_e && c->nnz > 0)
            for (int w = 0; w < 2; ++w)
                hipLaunchKernelGGL((k_sp_permute<SX>), dim3(2048), dim3(256), 0, c->stream, (const SX*)c->sp_e,
                                   (const int*)c->sp[w].perm, c->sp[w].count, (SX*)c->sp[w].val);
    }
    // 0/1 masks are bit-packed (32 columns per word): the mask then costs 1/32 of its fp32 bytes per pass
    static rri_status pack_mask_if_binary(rri_ctx* c) {
        if (c->Mbits) { (void)hipFree(c->Mbits); c->Mbits = nullptr; }
        if (c->Mcols) { (void)hipFree(c->Mcols); c->Mcols = nullptr; }
        c->mcols_tried = false;
        if (const char* e = getenv("RRI_MASK_BITS")) if (atoi(e) == 0) return RRI_OK;
        hipError_t err = hipMemsetAsync(c->itmp, 0, sizeof(i64), c->stream);
        if (err != hipSuccess) return RRI_ERR_HIP;
        hipLaunchKernelGGL((k_mask_nonbinary<SX>), dim3(2048), dim3(256), 0, c->stream, (const SX*)c->M, c->ldm,
                           c->n, c->d, (int*)c->itmp);
        int bad = 1;
        err = hipMemcpyAsync(&bad, c->itmp, sizeof(int), hipMemcpyDeviceToHost, c->stream);
        if (err == hipSuccess) err = hipStreamSynchronize(c->stream);
        if (err != hipSuccess) return RRI_ERR_HIP;
        if (bad) return RRI_OK;
        c->ldb = (c->LD + 3) / 4;   // one word per 8 rows x 4 columns
        err = hipMalloc((void**)&c->Mbits, (size_t)((c->n + 7) / 8) * c->ldb * sizeof(unsigned));
        if (err != hipSuccess) { c->Mbits = nullptr; return RRI_ERR_HIP; }
        hipLaunchKernelGGL((k_mask_pack<SX>), dim3(4096), dim3(256), 0, c->stream, (const SX*)c->M, c->ldm, c->n,
                           c->d, c->Mbits, c->ldb);
        err = hipStreamSynchronize(c->stream);
        return err == hipSuccess ? RRI_OK : RRI_ERR_HIP;
    }
    template <int NT>
    static void xtt_mfma_k(rri_ctx* c, const double* Tm, int m, double* out) {
        constexpr int VN = XVec<SX>::N;
        const size_t sh = 2 * 64 * (size_t)(16 * NT + 1) * sizeof(double) + 4 * 16 * (size_t)(64 + VN) * sizeof(SX);
        static bool attr_set[64] = {};
        if (!attr_set[c->device & 63]) {
            (void)hipFuncSetAttribute((const void*)k_xtt_mfma<SX, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            attr_set[c->device & 63] = true;
        }
        hipLaunchKernelGGL((k_xtt_mfma<SX, NT>), dim3((unsigned)((c->n + 63) / 64)), dim3(256), sh, c->stream,
                           (const SX*)c->X, c->ldx, Tm, c->LD, (int)c->n, (int)c->d, m, out, c->ldw);
    }
    static void xtt_any(rri_ctx* c, const double* Tm, int m, double* out) {   // out (m x n) = (X Tm^T)^T
        if (g_resid_mfma) {   // the product on the matrix cores, up to 64 rows of Tm per launch
            for (int l0 = 0; l0 < m; l0 += 64) {
                const int mm = std::min(64, m - l0);
                const double* Tp = Tm + (i64)l0 * c->LD;
                double* op = out + (i64)l0 * c->ldw;
                if (mm <= 16) xtt_mfma_k<1>(c, Tp, mm, op);
                else if (mm <= 32) xtt_mfma_k<2>(c, Tp, mm, op);
                else if (mm <= 48) xtt_mfma_k<3>(c, Tp, mm, op);
                else xtt_mfma_k<4>(c, Tp, mm, op);
            }
            return;
        }
        hipLaunchKernelGGL((k_xtt<SX>), dim3((unsigned)((c->n + 63) / 64)), dim3(256), 0, c->stream, (const SX*)c->X,
                           c->ldx, Tm, c->LD, (int)c->n, (int)c->d, m, out, c->ldw);
    }
    static void xtt(rri_ctx* c) { xtt_any(c, c->T, c->k, c->Qt); }
    // column sums against NV = 8 row-vectors at once (Qt: 8 x n, stride ldw): out rows <- X^T q_v
    static void colsums8(rri_ctx* c, const double* Qt, int nv, double* zmulti, double* out_rows) {
        constexpr int NV = 8;
        const int ncols = (int)std::min<i64>(c->ldx, c->LD);
        hipLaunchKernelGGL((k_colsums<SX, NV>), dim3(c->npanels * c->nrb), dim3(256), (size_t)NV * c->rpb * sizeof(double),
                           c->stream, (const SX*)c->X, c->ldx, (int)c->n, ncols, Qt, c->ldw, nv, zmulti, c->LD, c->rpb,
                           c->npanels, c->nrb);
        const int nb = (int)((c->LD + 31) / 32);
        for (int v = 0; v < nv; ++v)
            hipLaunchKernelGGL(k_reduce, dim3(nb), dim3(1024), 0, c->stream,
                               (const double*)zmulti + (i64)v * c->nrb * c->LD, c->LD, c->nrb, (const double*)nullptr, 0,
                               c->k, out_rows + (i64)v * c->LD, (const DevState*)c->st);
    }
    static bool resid_w_resident(const rri_ctx* c) { return c->k <= 256; }   // 128 KiB of W tile at most
    static size_t resid_shmem(const rri_ctx* c) {
        return ((size_t)(resid_w_resident(c) ? c->k : 32) * 64 + 32 * 64) * sizeof(double) + 64 * 17 * sizeof(double);
    }
    static void resid(rri_ctx* c, bool masked, bool write_e, double* rowobj, double* rowpos) {
        if (c->sparse) { sp_resid(c, write_e, rowobj, rowpos); return; }   // outside the pattern nothing contributes
        const unsigned nb = (unsigned)((c->n + 63) / 64);
        if (c->k <= 64 && g_resid_mfma) {   // the k-panel product on the matrix cores
            const int ks = c->k <= 16 ? 4 : c->k <= 32 ? 8 : c->k <= 48 ? 12 : c->k <= 52 ? 13 : 16;
            const size_t shm = 2 * (size_t)(4 * ks) * RESID_TS * sizeof(double);
            const bool sums = rowobj || rowpos;      // a plain rebuild wants neither: its epilogue is convert, subtract, store
            // column ranges per row block (the rebuild without row sums): ~12 rounds of the chip's 2 workgroups per CU or more, so
            // that the last, partly filled round costs a twelfth and not a quarter (RRI_RESID_SPLIT=1: one range)
            int nsplit = 1;
            if (!sums && write_e) {
                const i64 per_round = 2 * (i64)std::max(c->n_cu, 1);
                nsplit = (int)std::min<i64>((c->d + 63) / 64, std::max<i64>(1, (12 * per_round + nb - 1) / nb));
                if (g_resid_split > 0) nsplit = (int)std::min<i64>((c->d + 63) / 64, g_resid_split);
            }
            const int dchunk = (int)round_up((c->d + nsplit - 1) / nsplit, 64);
            const unsigned ny = (unsigned)((c->d + dchunk - 1) / dchunk);
#define RRI_RESID_M2(MK, WE, KS_, SM)                                                                                \
    hipLaunchKernelGGL((k_resid_mfma<SX, MK, WE, KS_, 4, SM>), dim3(nb, (SM) ? 1u : ny), dim3(256), shm, c->stream, (const SX*)c->X, c->ldx, \
                       (const SX*)c->M, c->ldm, (const unsigned*)c->Mbits, c->ldb, (const double*)c->W, c->ldw,      \
                       (const double*)c->T, c->LD, (int)c->n, (int)c->d, c->k, rowobj, rowpos, (SX*)c->E, c->LD, (SM) ? (int)round_up(c->d, 64) : dchunk)
// a residual written without its row sums (the per-sweep rebuild of the explicit-residual schedule) skips them; !(WE)
// keeps the first branch from instantiating a kernel that no WRITE_E = false caller can reach
#define RRI_RESID_M(MK, WE, KS_)                           \
    do {                                                   \
        if (WE && !sums) RRI_RESID_M2(MK, WE, KS_, !(WE)); \
        else RRI_RESID_M2(MK, WE, KS_, true);              \
    } while (0)
#define RRI_RESID_K(MK, WE)                          \
    switch (ks) {                                    \
        case 4: RRI_RESID_M(MK, WE, 4); break;       \
        case 8: RRI_RESID_M(MK, WE, 8); break;       \
        case 12: RRI_RESID_M(MK, WE, 12); break;     \
        case 13: RRI_RESID_M(MK, WE, 13); break;     \
        default: RRI_RESID_M(MK, WE, 16); break;     \
    }
            if (masked && write_e) RRI_RESID_K(true, true)
            else if (masked) RRI_RESID_K(true, false)
            else if (write_e) RRI_RESID_K(false, true)
            else RRI_RESID_K(false, false)
#undef RRI_RESID_K
#undef RRI_RESID_M
#undef RRI_RESID_M2
            return;
        }
        const size_t sh = resid_shmem(c);
#define RRI_RESID(MK, WE)                                                                                       \
    hipLaunchKernelGGL((k_resid<SX, MK, WE>), dim3(nb), dim3(256), sh, c->stream, (const SX*)c->X, c->ldx,      \
                       (const SX*)c->M, c->ldm, (const unsigned*)c->Mbits, c->ldb, (const double*)c->W, c->ldw,   \
                       (const double*)c->T, c->LD, (int)c->n, (int)c->d, c->k, rowobj, rowpos, (SX*)c->E, c->LD,       \
                       resid_w_resident(c) ? 1 : 0)
        if (masked && write_e) RRI_RESID(true, true);
        else if (masked) RRI_RESID(true, false);
        else if (write_e) RRI_RESID(false, true);
        else RRI_RESID(false, false);
#undef RRI_RESID
    }
    static void reset_row(rri_ctx* c) {
        if (c->sparse) {
            (void)hipMemsetAsync(c->xraw, 0, (size_t)c->LD * sizeof(double), c->stream);
            hipLaunchKernelGGL((k_sp_reset_row<SX>), dim3((unsigned)std::max(1, (c->sp_max_row + 255) / 256)), dim3(256), 0,
                               c->stream, (const i64*)c->sp_rowptr, (const int*)c->sp_col, (const SX*)c->sp_x,
                               (const double*)c->W, c->ldw, (const double*)c->T, c->LD, c->k, (const i64*)c->itmp,
                               c->xraw);
            return;
        }
        hipLaunchKernelGGL((k_reset_row<SX>), dim3((unsigned)((c->d + 255) / 256)), dim3(256), 0, c->stream,
                           (const SX*)c->X, c->ldx, (const double*)c->W, c->ldw, (const double*)c->T, c->LD,
                           (int)c->d, c->k, (const i64*)c->itmp, c->xraw);
    }
    static hipError_t set_attrs() {
        hipError_t e = hipSuccess;
        const void* fns[] = {(const void*)k_resid<SX, true, true>, (const void*)k_resid<SX, true, false>,
                             (const void*)k_resid<SX, false, true>, (const void*)k_resid<SX, false, false>};
        for (const void* f : fns) {
            e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return e;
        }
        return e;
    }
};

#define DISPATCH(c, expr)                       \
    do {                                        \
        if ((c)->dtype == RRI_F32) {            \
            typedef LaunchX<float> L;           \
            expr;                               \
        } else {                                \
            typedef LaunchX<double> L;          \
            expr;                               \
        }                                       \
    } while (0)

struct LK {  // float64-only kernels
    // rows of Gpart a column update leaves: one per 64-row tile (k_wcol) or per 256-row block (k_wcol_resid)
    static int gpart_rows(const rri_ctx* c) { return c->gpart_n; }
    template <bool UPDATE>
    static void wcol_resid(rri_ctx* c, int t, int tn, int sweep) {
        TimedScope ts(c, 1);
        hipLaunchKernelGGL((k_wcol_resid<UPDATE>), dim3(c->nwb256), dim3(256), 0, c->stream, c->W, c->ldw, (int)c->n,
                           c->k, t, tn, (const double*)c->Ypart, c->npanels, (const double*)c->Ttpart, c->nsplit,
                           c->dwv, c->Gpart, sweep, kparams(c), c->st);
        c->gpart_n = c->nwb256;
    }
    static size_t wcol_shmem(const rri_ctx* c) { return (size_t)(2 * c->k + 2 + 256 + 64) * sizeof(double); }
    template <bool UPDATE, bool CARRY>
    static void wcol_src(rri_ctx* c, int t, int tn, int sweep, const double* ypart, int nslices) {
        TimedScope ts(c, 1);
        hipLaunchKernelGGL((k_wcol<UPDATE, CARRY>), dim3(c->nwb), dim3(256), wcol_shmem(c), c->stream, c->W, c->ldw,
                           (int)c->n, c->k, t, tn, ypart, nslices, (const double*)c->Ttpart, c->ttpart_n, c->Gpart,
                           c->XYpart + (i64)t * c->xy_stride, sweep, kparams(c), c->st);
        c->gpart_n = c->nwb;
        if (UPDATE) note_xy(c, t, c->nwb * WCOL_TILES);
    }
    // the cross terms <w_t, X t_t> of topic t were left in XYpart as `rows` block partials: the objective after the
    // sweep needs all k topics in order, written with the same block count
    static void note_xy(rri_ctx* c, int t, int rows) {
        if (t == 0) { c->xy_run = 1; c->xy_rows = rows; }
        else c->xy_run = (c->xy_run == t && c->xy_rows == rows) ? t + 1 : -1;
        c->xy_valid = c->xy_run == c->k;
        c->obj_track_valid = false;
    }
    template <bool UPDATE, bool CARRY>
    static void wcol(rri_ctx* c, int t, int tn, int sweep) {
        wcol_src<UPDATE, CARRY>(c, t, tn, sweep, (const double*)c->Ypart, c->npanels);
    }
    static void reduce(rri_ctx* c) {
        const int nb = (int)((c->LD + 31) / 32) + GRAM_SLICES;
        hipLaunchKernelGGL(k_reduce, dim3(nb), dim3(1024), 0, c->stream, (const double*)c->Zpart, c->LD, c->nrb,
                           (const double*)c->Gpart, gpart_rows(c), c->k, c->red, (const DevState*)c->st);
    }
    // no simplex projection configured: k_trow_numer stores the row itself and k_tgram finishes the checks
    static bool light(const rri_ctx* c) { return !(c->prm.project_T_each_iter && c->prm.has_t_row_sum); }
    static void trow(rri_ctx* c, int t, int check_prev, int tprev, int sweep, bool force_final) {
        hipLaunchKernelGGL(k_trow_numer, dim3(c->ntb), dim3(128), 0, c->stream, c->T, c->LD, (int)c->d, c->k, t,
                           (const double*)c->red, c->LD, c->xraw, c->tpart, c->tpart_idx, check_prev, tprev, sweep,
                           kparams(c), c->st, resid_sched(c) ? 1 : 0, resid_sched(c) ? c->told : (double*)nullptr);
        c->tpart_n = c->ntb;
        trow_final_if_needed(c, t, sweep, force_final);
    }
    // launch-bound sizes: k_reduce and k_trow_numer as one launch (every workgroup reduces the Gram partials itself)
    static bool small(const rri_ctx* c) {
        return g_trow_small && (double)gpart_rows(c) * (c->k + 2) * c->ntb32 <= 4.0e6;
    }
    static void trow_small(rri_ctx* c, int t, int check_prev, int tprev, int sweep, bool force_final) {
        hipLaunchKernelGGL(k_trow_small, dim3(c->ntb32), dim3(1024), 0, c->stream, c->T, c->LD, (int)c->d, c->k, t,
                           (const double*)c->Zpart, c->nrb, (const double*)c->Gpart, gpart_rows(c), c->red, c->LD, c->xraw,
                           c->tpart, c->tpart_idx, check_prev, tprev, sweep, kparams(c), c->st, resid_sched(c) ? 1 : 0,
                           resid_sched(c) ? c->told : (double*)nullptr);
        c->tpart_n = c->ntb32;
        trow_final_if_needed(c, t, sweep, force_final);
    }
    static void trow_final_if_needed(rri_ctx* c, int t, int sweep, bool force_final) {
        if (!light(c) || force_final)
            hipLaunchKernelGGL(k_trow_final, dim3(1), dim3(1024), 0, c->stream, c->T, c->LD, (int)c->d, t, c->xraw,
                               (const double*)c->tpart, (const i64*)c->tpart_idx, c->tpart_n, sweep, kparams(c), c->st);
    }
    static void check_prev_only(rri_ctx* c, int tprev, int sweep, int pos) {
        hipLaunchKernelGGL(k_check_red, dim3(1), dim3(64), 0, c->stream, (const double*)c->red, c->LD, c->k, tprev,
                           sweep, pos, kparams(c), c->st);
    }
    static void tgram(rri_ctx* c, int t, int finish, int sweep) {
        c->ttpart_n = c->nsplit;
        hipLaunchKernelGGL(k_tgram, dim3(c->k, c->nsplit), dim3(256), 0, c->stream, (const double*)c->T, c->LD,
                           (int)c->d, c->k, t, c->Ttpart, (const double*)c->tpart, c->tpart_n, finish, sweep,
                           kparams(c), c->st);
    }
    static void scale_wcol(rri_ctx* c, int t) {
        hipLaunchKernelGGL(k_scale_wcol, dim3((unsigned)((c->n + 255) / 256)), dim3(256), 0, c->stream, c->W,
                           c->ldw, (int)c->n, t, (const DevState*)c->st);
    }
    static void check_wcol(rri_ctx* c, int tprev, int sweep, int pos) {
        hipLaunchKernelGGL(k_check_wcol, dim3(1), dim3(256), 0, c->stream, (const double*)c->Gpart, gpart_rows(c), c->k,
                           tprev, sweep, pos, kparams(c), c->st);
    }
    static void proj_rows(rri_ctx* c, double s, const double* svec) {
        hipLaunchKernelGGL(k_proj_rows, dim3((unsigned)((c->n + 255) / 256)), dim3(256), 0, c->stream, c->W, c->ldw, (int)c->n, c->k, s,
                           svec);
    }
    static void norms(rri_ctx* c, const double* A, i64 rows, i64 cols, i64 ld) {
        hipLaunchKernelGGL(k_norms, dim3(256), dim3(256), 0, c->stream, A, rows, cols, ld, c->normpart);
    }
    static void reset_commit(rri_ctx* c, int t) {
        const i64 m = std::max(c->n, c->d);
        hipLaunchKernelGGL(k_reset_commit, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, c->stream, c->W, c->ldw,
                           c->T, c->LD, (int)c->n, (int)c->d, t, (const i64*)c->itmp, (const double*)c->xraw);
    }
    static void set_row_col(rri_ctx* c, int t, const double* trow, const double* wcolv) {
        const i64 m = std::max(c->n, c->d);
        hipLaunchKernelGGL(k_set_row_col, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, c->stream, c->W, c->ldw,
                           c->T, c->LD, (int)c->n, (int)c->d, t, trow, wcolv);
    }
    static void argmax_rows(rri_ctx* c, int* out) {
        hipLaunchKernelGGL(k_argmax_rows, dim3((unsigned)((c->n + 255) / 256)), dim3(256), 0, c->stream,
                           (const double*)c->W, c->ldw, (int)c->n, c->k, out);
    }
    static void masked_sqerr(rri_ctx* c, const i64* ij, const double* vals, i64 count, double lo, double hi) {
        hipLaunchKernelGGL(k_masked_sqerr, dim3(256), dim3(256), 0, c->stream, (const double*)c->W, c->ldw,
                           (const double*)c->T, c->LD, c->k, ij, vals, count, lo, hi, c->normpart);
    }
};

// ---- upload / download with conversion -------------------------------------------------------
template <typename Src, typename Dst, bool TR>
void launch_convert(rri_ctx* c, const void* src, i64 lds_, void* dst, i64 ldd, i64 rows, i64 cols) {
    const i64 total = rows * cols;
    const unsigned nb = (unsigned)std::min<i64>(4096, (total + 255) / 256);
    hipLaunchKernelGGL((k_convert2d<Src, Dst, TR>), dim3(nb ? nb : 1), dim3(256), 0, c->stream, (const Src*)src,
                       lds_, (Dst*)dst, ldd, rows, cols);
}

// host (rows x cols, stride ld, host_dtype) -> device (stride ldd, dev_dtype).
// transpose: the device image is cols x rows (dst[c][r] = host[r][c]).
rri_status to_device(rri_ctx* c, const void* host, i64 ld, int host_dtype, void* dev, i64 ldd, i64 rows,
                     i64 cols, int dev_dtype, bool transpose = false) {
    if (!host || ld < cols) return fail(c, RRI_ERR_INVALID, "bad host matrix (ld=%lld < cols=%lld)", ld, cols);
    if (host_dtype != RRI_F32 && host_dtype != RRI_F64) return fail(c, RRI_ERR_INVALID, "bad host dtype");
    const size_t hs = host_dtype == RRI_F32 ? 4 : 8;
    const size_t ds = dev_dtype == RRI_F32 ? 4 : 8;
    if (host_dtype == dev_dtype && !transpose) {
        HIPCHK(c, hipMemcpy2DAsync(dev, ldd * ds, host, ld * hs, cols * hs, rows, hipMemcpyHostToDevice,
                                   c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        return RRI_OK;
    }
    void* tmp = nullptr;
    HIPCHK(c, hipMalloc(&tmp, (size_t)rows * cols * hs));
    hipError_t e = hipMemcpy2DAsync(tmp, cols * hs, host, ld * hs, cols * hs, rows, hipMemcpyHostToDevice,
                                    c->stream);
    if (e == hipSuccess) {
        const bool hf = host_dtype == RRI_F32, df = dev_dtype == RRI_F32;
#define RRI_CONV(SRC, DST)                                                                  \
    do {                                                                                    \
        if (transpose) launch_convert<SRC, DST, true>(c, tmp, cols, dev, ldd, rows, cols);  \
        else launch_convert<SRC, DST, false>(c, tmp, cols, dev, ldd, rows, cols);           \
    } while (0)
        if (hf && df) RRI_CONV(float, float);
        else if (hf) RRI_CONV(float, double);
        else if (df) RRI_CONV(double, float);
        else RRI_CONV(double, double);
#undef RRI_CONV
        e = hipStreamSynchronize(c->stream);
    }
    (void)hipFree(tmp);
    if (e != hipSuccess) return fail(c, RRI_ERR_HIP, "upload failed: %s", hipGetErrorString(e));
    return RRI_OK;
}

// device (dev_dtype, stride ldd) -> host rows x cols.  transpose: the device image is cols x rows.
rri_status to_host(rri_ctx* c, const void* dev, i64 ldd, void* host, i64 ld, int host_dtype, i64 rows, i64 cols,
                   int dev_dtype, bool transpose = false) {
    if (!host || ld < cols) return fail(c, RRI_ERR_INVALID, "bad host matrix (ld=%lld < cols=%lld)", ld, cols);
    if (host_dtype != RRI_F32 && host_dtype != RRI_F64) return fail(c, RRI_ERR_INVALID, "bad host dtype");
    const size_t hs = host_dtype == RRI_F32 ? 4 : 8;
    const size_t ds = dev_dtype == RRI_F32 ? 4 : 8;
    if (host_dtype == dev_dtype && !transpose) {
        HIPCHK(c, hipMemcpy2DAsync(host, ld * hs, dev, ldd * ds, cols * hs, rows, hipMemcpyDeviceToHost,
                                   c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        return RRI_OK;
    }
    void* tmp = nullptr;
    HIPCHK(c, hipMalloc(&tmp, (size_t)rows * cols * hs));
    {
        const bool hf = host_dtype == RRI_F32, df = dev_dtype == RRI_F32;
        // source is the device image; for a transposed image its shape is cols x rows
#define RRI_CONV(SRC, DST)                                                                     \
    do {                                                                                       \
        if (transpose) launch_convert<SRC, DST, true>(c, dev, ldd, tmp, cols, cols, rows);     \
        else launch_convert<SRC, DST, false>(c, dev, ldd, tmp, cols, rows, cols);              \
    } while (0)
        if (df && hf) RRI_CONV(float, float);
        else if (df) RRI_CONV(float, double);
        else if (hf) RRI_CONV(double, float);
        else RRI_CONV(double, double);
#undef RRI_CONV
    }
    hipError_t e = hipMemcpy2DAsync(host, ld * hs, tmp, cols * hs, cols * hs, rows, hipMemcpyDeviceToHost,
                                    c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(tmp);
    if (e != hipSuccess) return fail(c, RRI_ERR_HIP, "download failed: %s", hipGetErrorString(e));
    return RRI_OK;
}

void invalidate(rri_ctx* c) {
    c->carry_valid = false;
    c->carry_topic = -1;
    c->resid_valid = false;
    c->xy_run = -1;
    c->xy_valid = false;
    c->obj_track_valid = false;
}

// ---- the topic-step scheduler ------------------------------------------------------------------
// The column verdict of _check_reset_W / the assert of nmf.py:471-476 from the partial sums in Gpart, taken NOW
// (where no T-row step follows that would carry it).  Row-sharded: the column sum is global, so the local share is
// all-reduced first (2 doubles) and every rank reaches the same verdict.
void wcheck_now(rri_ctx* c, int tprev, int sweep, int pos) {
    if (!c->comm) {
        if (c->weighted)
            hipLaunchKernelGGL(k_wcheck_wcol, dim3(1), dim3(256), 0, c->stream, (const double*)c->Gpart, c->nwb256, c->k,
                               tprev, sweep, pos, kparams(c), c->st, (double*)nullptr);
        else LK::check_wcol(c, tprev, sweep, pos);
        return;
    }
    if (c->weighted)
        hipLaunchKernelGGL(k_wcheck_wcol, dim3(1), dim3(256), 0, c->stream, (const double*)c->Gpart, c->nwb256, c->k,
                           tprev, sweep, pos, kparams(c), c->st, c->ctail);
    else
        hipLaunchKernelGGL(k_colsum_tail, dim3(1), dim3(256), 0, c->stream, (const double*)c->Gpart, LK::gpart_rows(c),
                           c->k, c->ctail, (const DevState*)c->st);
    comm_allreduce(c, c->ctail, 2);
    hipLaunchKernelGGL(k_wcheck_tail, dim3(1), dim3(64), 0, c->stream, (const double*)c->ctail, tprev, sweep, pos,
                       kparams(c), c->st);
}

// carry := Zpart/Gpart hold the partial sums of topic `carry_topic`.
void enqueue_prologue(rri_ctx* c, int t, int sweep) {
    // the pending W-column check reads Gpart, which the prologue overwrites: resolve it first
    if (c->pending_wcheck) {
        wcheck_now(c, c->pending_wcheck_topic, sweep, t);
        c->pending_wcheck = false;
    }
    LK::wcol<false, true>(c, t, t, sweep);                             // Gram row of w_t
    DISPATCH(c, (L::template pass<false, true>(c, t, t)));            // w_t^T X
    c->carry_valid = true;
    c->carry_topic = t;
}

void enqueue_T_half(rri_ctx* c, int sweep, int t, bool standalone) {
    if (!c->carry_valid || c->carry_topic != t) enqueue_prologue(c, t, sweep);
    {
        TimedScope ts(c, 2);
        const int chk = c->pending_wcheck ? 1 : 0;
        if (LK::small(c) && !c->comm) {
            LK::trow_small(c, t, chk, c->pending_wcheck_topic, sweep, standalone);
        } else {
            // the one cross-row reduction of a topic step: [w_t^T X | slices of (w_t^T W, ||w_t||^2, sum W[:,t-1])];
            // row-sharded, the ranks all-reduce it here, on the stream, between the two kernels
            LK::reduce(c);
            comm_allreduce(c, c->red, c->LD + (i64)GRAM_SLICES * (c->k + 2));
            LK::trow(c, t, chk, c->pending_wcheck_topic, sweep, standalone);
        }
        c->pending_wcheck = false;
        if (c->prm.fix_W && no_regs(c)) LK::scale_wcol(c, t);
    }
    c->carry_valid = false;
    c->resid_valid = false;
    c->q_valid = false; c->gfull_valid = false;   // T changed
    c->xy_valid = false;  // a T row changed: complete again after the W half of topic k-1
    c->obj_track_valid = false;
}

void enqueue_W_half(rri_ctx* c, int sweep, int t) {
    const int k = c->k;
    const bool carry_next = (k > 1) && !c->prm.fix_T;
    const int tn = (t + 1) % k;
    // T T[t,:]^T for k_wcol; the T-row checks ride along only when a T half of this topic just ran and left its sums.
    // Where a pass follows, the job joins its grid (no launch of its own); with T fixed there is no pass.
    const int finish = (LK::light(c) && !c->prm.fix_T && !c->skip_row_finish) ? 1 : 0;
    c->skip_row_finish = false;
    TgramJob job{};
    if (c->prm.fix_T || g_side_jobs == 0) {
        TimedScope ts(c, 2);
        LK::tgram(c, t, finish, sweep);
    } else {
        job = TgramJob{(const double*)c->T, c->LD, (int)c->d, c->k, t, c->Ttpart, (const double*)c->tpart, c->tpart_n,
                       c->nsplit, finish, sweep, kparams(c), c->st, c->k * c->nsplit};
        c->ttpart_n = c->nsplit;
    }
    if (carry_next) {
        DISPATCH(c, (L::template pass<true, true>(c, t, tn, job)));
        LK::wcol<true, true>(c, t, tn, sweep);
        c->carry_valid = true;
        c->carry_topic = tn;
        c->pending_wcheck = true;
        c->pending_wcheck_topic = t;
    } else {
        if (c->prm.fix_T) {
            // T is fixed: X T^T is computed once (k_xtt) and reused by every topic and every sweep
            if (!c->q_valid) {
                DISPATCH(c, L::xtt(c));
                c->q_valid = true;
            }
            LK::wcol_src<true, false>(c, t, tn, sweep, c->Qt + (i64)t * c->ldw, 1);
        } else {
            DISPATCH(c, (L::template pass<true, false>(c, t, tn, job)));
            LK::wcol<true, false>(c, t, tn, sweep);
        }
        // position of the NEXT step, where a resumed run continues
        int ns = sweep, np = t + 1;
        if (np == k) { np = 0; ns = sweep + 1; }
        wcheck_now(c, t, ns, np);
        c->carry_valid = false;
    }
    c->resid_valid = false;
}

// ---- explicit-residual schedule (RRI_UNWEIGHTED_RESIDUAL; SURVEY 8a "explicit-residual variant") --------------
// R = X - W T is kept in HBM (c->E) and every topic step is ONE read-modify-write pass over it:
//     pass of step t   R <- R - dw_{t-1} t_{t-1}^T - w_t dt_t^T   (the two rank-one terms pending since the W half of
//                      step t-1 and the T half of step t), fused with  y = R t_t  and  z = R^T w_{t+1}  of the new R
//     W half           numer_W = y + w_t ||t_t||^2 ;  dw_t = w_t' - w_t stays pending
//     T half (t+1)     numer_T = z - t_t <dw_t, w_{t+1}> + t_{t+1} ||w_{t+1}||^2   (column t+1 of W is untouched by
//                      step t, so z needs only the rank-one correction for the term that is still pending)
// 2 n d s bytes per topic step.  R is rebuilt from X, W, T (the k-panel GEMM k_resid_mfma) once per sweep, so the
// rounding of the stored residual never accumulates over more than k updates.  carry := Zpart / Gpart hold z and
// the correction coefficients for topic `carry_topic`.
void r_refresh(rri_ctx* c) {
    DISPATCH(c, L::resid(c, false, true, nullptr, nullptr));   // R = X - W T
    c->resid_valid = true;
    c->resid_fresh = true;
    c->dw_pending = false;     // the rebuilt R contains the current W and T
    c->dt_pending = false;
}

void enqueue_rT_half(rri_ctx* c, int sweep, int t, bool standalone) {
    if (!c->resid_valid) r_refresh(c);
    if (!c->carry_valid || c->carry_topic != t) {
        if (c->pending_wcheck) {            // reads Gpart, which the prologue overwrites
            wcheck_now(c, c->pending_wcheck_topic, sweep, t);
            c->pending_wcheck = false;
        }
        if (c->dw_pending) r_refresh(c);    // no pass to fold the pending column change into: rebuild instead
        LK::wcol_resid<false>(c, t, t, sweep);             // ||w_t||^2
        DISPATCH(c, L::rpass_colsums(c, t));               // R^T w_t
        c->carry_valid = true;
        c->carry_topic = t;
    }
    {
        TimedScope ts(c, 2);
        const int chk = c->pending_wcheck ? 1 : 0;
        if (LK::small(c) && !c->comm) {
            LK::trow_small(c, t, chk, c->pending_wcheck_topic, sweep, standalone);
        } else {
            // row-sharded: every rank holds its rows of R; the column sums R^T w_t, <dw, w_t>, ||w_t||^2 and the column
            // sum of the last update are sums over rows, all-reduced in one message as in the Gram form
            LK::reduce(c);
            comm_allreduce(c, c->red, c->LD + (i64)GRAM_SLICES * (c->k + 2));
            LK::trow(c, t, chk, c->pending_wcheck_topic, sweep, standalone);
        }
        c->pending_wcheck = false;
    }
    c->carry_valid = false;
    c->resid_fresh = false;
    c->dt_pending = true;     // told holds the previous row: R lacks w_t (T[t,:] - told)^T
    c->xy_valid = false;
    c->obj_track_valid = false;
}

void enqueue_rW_half(rri_ctx* c, int sweep, int t) {
    const int k = c->k;
    const int tn = (t + 1) % k;
    if (!c->resid_valid) r_refresh(c);
    const int finish = (LK::light(c) && !c->skip_row_finish) ? 1 : 0;
    c->skip_row_finish = false;
    TgramJob job{};
    if (g_side_jobs == 0) {
        TimedScope ts(c, 2);
        LK::tgram(c, t, finish, sweep);
    } else {
        job = TgramJob{(const double*)c->T, c->LD, (int)c->d, c->k, t, c->Ttpart, (const double*)c->tpart, c->tpart_n,
                       c->nsplit, finish, sweep, kparams(c), c->st, c->k * c->nsplit};
        c->ttpart_n = c->nsplit;
    }
    const double* trow = c->T + (i64)t * c->LD;
    DISPATCH(c, {
        typename L::Upd u;
        u.a = c->dw_pending ? c->dwv : c->zeros;
        u.b = c->T + (i64)(c->dw_pending ? c->dw_topic : t) * c->LD;
        u.a2 = c->W + (i64)t * c->ldw;               // the column BEFORE its update below
        u.b2 = trow;
        u.b2sub = c->dt_pending ? c->told : trow;    // no T half since the last pass: dt = 0
        L::rank_update(c, c->E, c->LD, u, trow, c->W + (i64)tn * c->ldw, job);
    });
    LK::wcol_resid<true>(c, t, tn, sweep);
    c->dw_pending = true;
    c->dw_topic = t;
    c->dt_pending = false;
    c->resid_fresh = false;
    c->carry_valid = true;
    c->carry_topic = tn;
    c->pending_wcheck = true;
    c->pending_wcheck_topic = t;
}

// ---- weighted flavour (rri_wrri_kernels.hpp) ---------------------------------------------------------
// carry := Zpart / Z2part hold the column sums (a, nw) of topic `carry_topic` over the CURRENT E.
void w_refresh(rri_ctx* c) {
    DISPATCH(c, L::resid(c, true, true, nullptr, nullptr));   // E = M .* (X - W T)
    c->resid_valid = true;
    c->resid_fresh = true;
    c->dt_pending = false;     // the rebuilt E contains the current T
    c->dw_pending = false;     // ... and W (pattern-only handles: the row copy's pending column change)
    c->carry_valid = false;
}

// take_check: the pending column verdict of the last W update rides in this launch (one device, rri_sweep's own loop)
void w_reduce(rri_ctx* c, bool take_check = false, int sweep = 0, int pos = 0) {
    const int nb = (int)((c->LD + 63) / 64);
    const int chk = (take_check && c->pending_wcheck) ? 1 : 0;
    const bool nwm = c->nw_mask;     // (set by enqueue_wT_sums: the T-row step's nw lies in N2part, k_wmcorr_cols' row blocks)
    hipLaunchKernelGGL(k_wreduce, dim3(nb), dim3(1024), 0, c->stream, (const double*)c->Zpart,
                       (const double*)(nwm ? c->N2part : c->Z2part), c->LD, c->nrb, nwm ? c->wcorr_nrb : c->nrb, c->wcorr ? (const double*)c->Cpart : (const double*)nullptr, c->wcorr_nrb,
                       (const double*)(c->T + (i64)c->wcorr_topic * c->LD), c->red, (const double*)c->Gpart, c->nwb256, c->k, chk,
                       c->pending_wcheck_topic, sweep, pos, kparams(c), c->st);
    if (chk) c->pending_wcheck = false;
}

// few row blocks of partial column sums, one device: the column verdict, both reductions and the closed form of the T row are
// ONE launch (k_wtrow_small)
bool wtrow_small(const rri_ctx* c) { return !c->comm && c->nrb <= 64; }

// column sums (a, nw) of topic t over the current E into red[0 .. 2 LD)
// `fused`: the caller goes straight on to enqueue_wT_solve(..., fused) -- rri_sweep does; the split stepping of
// rri_topic_reduce_local / rri_topic_finish, where the host reads (and may rewrite) red in between, does not
void enqueue_wT_sums(rri_ctx* c, int t, bool fused = false, bool take_check = false, int sweep = 0) {
    const double* wt_t = c->W + (i64)t * c->ldw;
    if (!c->carry_valid || c->carry_topic != t)
        DISPATCH(c, (L::template wpass<false, true, false, false>(c, nullptr, wt_t, c->zeros, c->zeros, nullptr, nullptr)));
    // dense handles: E -- and with it these sums, carried or just taken -- lacks the rank-one term of the last W update
    // (dw T[dw_topic,:]^T under the mask; the next pass folds it in).  Its share of the sums comes from the mask alone.
    c->wcorr = !c->sparse && c->dw_pending;
    c->nw_mask = false;
    DISPATCH(c, c->nw_mask = L::nw_from_mask(c));      // sparse 0/1 mask: nw from the same mask-only launch (the passes skipped it)
    if (c->wcorr || c->nw_mask) {
        if (c->wcorr) c->wcorr_topic = c->dw_topic;
        DISPATCH(c, L::wmcorr(c, wt_t, c->wcorr ? c->dwv : nullptr));
    }
    if (fused) return;       // reduced inside k_wtrow_small
    TimedScope ts(c, 2);
    w_reduce(c, take_check, sweep, t);
}

// T row from red (local sums, or all-reduced ones on the row-sharded path)
void enqueue_wT_solve(rri_ctx* c, int sweep, int t, bool fused = false) {
    const double* wt_t = c->W + (i64)t * c->ldw;
    {
        TimedScope ts(c, 2);
        const int nb_small = (int)((c->LD + 127) / 128);
        if (fused) {
            const int nb = nb_small;
            hipLaunchKernelGGL(k_wtrow_small, dim3(nb), dim3(128), 0, c->stream, (const double*)c->T, c->LD, (int)c->d, t,
                               (const double*)c->Zpart, (const double*)(c->nw_mask ? c->N2part : c->Z2part), c->LD, c->nrb,
                               c->nw_mask ? c->wcorr_nrb : c->nrb, c->wcorr ? (const double*)c->Cpart : (const double*)nullptr, c->wcorr_nrb,
                               (const double*)(c->T + (i64)c->wcorr_topic * c->LD), (const double*)c->Gpart,
                               c->nwb256, c->k, c->pending_wcheck ? 1 : 0, c->pending_wcheck_topic, sweep, c->red, c->xraw,
                               c->tpart, c->tpart_idx, kparams(c), c->st);
            c->pending_wcheck = false;
        } else
        hipLaunchKernelGGL(k_wtrow, dim3(c->ntb), dim3(128), 0, c->stream, (const double*)c->T, c->LD, (int)c->d, t,
                           (const double*)c->red, c->LD, c->xraw, c->tpart, c->tpart_idx, kparams(c),
                           (const DevState*)c->st);
        const int scale_w = (c->prm.fix_W && no_regs(c)) ? 1 : 0;
        hipLaunchKernelGGL(k_wtrow_final, dim3(1), dim3(1024), 0, c->stream, c->T, c->LD, (int)c->d, t, c->xraw,
                           (const double*)c->tpart, (const i64*)c->tpart_idx, fused ? nb_small : c->ntb, c->dtv, scale_w, sweep,
                           kparams(c), c->st);
    }
    c->carry_valid = false;
    c->resid_fresh = false;
    c->dt_pending = true;
    if (c->prm.fix_W && c->dw_pending) {   // a column change still pending on the row copy and no W half to fold it: rebuild
        c->resid_valid = false;
        c->dt_pending = false;
        if (no_regs(c)) LK::scale_wcol(c, t);
    } else if (c->prm.fix_W) {   // no W half follows: fold dt into E now, then rescale the kept column (nmf.py:450-452)
        DISPATCH(c, (L::template wpass<false, false, false, true>(c, nullptr, nullptr, wt_t, c->dtv, nullptr, nullptr)));
        c->dt_pending = false;
        if (no_regs(c)) LK::scale_wcol(c, t);
    }
}

void enqueue_wT_half(rri_ctx* c, int sweep, int t) {
    const bool fused = wtrow_small(c);
    enqueue_wT_sums(c, t, fused, /*take_check=*/!c->comm, sweep);
    if (c->comm) {
        // row-sharded: red = [numerator | denominator | sum of the last updated column, its negative-denominator flag]
        // is all-reduced; the pending column verdict is taken from the reduced tail (option A)
        double* tail = c->red + 2 * c->LD;
        if (c->pending_wcheck)
            hipLaunchKernelGGL(k_wcheck_wcol, dim3(1), dim3(256), 0, c->stream, (const double*)c->Gpart, c->nwb256, c->k,
                               c->pending_wcheck_topic, sweep, t, kparams(c), c->st, tail);
        else
            (void)hipMemsetAsync(tail, 0, 2 * sizeof(double), c->stream);
        comm_allreduce(c, c->red, 2 * c->LD + 2);
        if (c->pending_wcheck)
            hipLaunchKernelGGL(k_wcheck_tail, dim3(1), dim3(64), 0, c->stream, (const double*)tail, c->pending_wcheck_topic,
                               sweep, t, kparams(c), c->st);
        c->pending_wcheck = false;
    }
    enqueue_wT_solve(c, sweep, t, fused);
}

void enqueue_wW_half(rri_ctx* c, int sweep, int t, bool defer_check = false) {
    const int k = c->k;
    const int tn = (t + 1) % k;
    const double* trow = c->T + (i64)t * c->LD;
    // pending T-row correction: none when T is fixed, or when E was rebuilt after the row changed (reset + resume)
    const double* b1 = (c->prm.fix_T || !c->dt_pending) ? c->zeros : c->dtv;
    c->dt_pending = false;
    c->resid_fresh = false;
    const bool sp_merged = c->sparse;
    if (sp_merged) {
        // Pattern-only handles keep two copies of the residual, and each copy serves ONE kind of sum: rows -> row
        // products, columns -> column sums.  So the row copy need not be current between its own passes: the W-column
        // change of a topic step (dw t^T) is folded into it by the row pass of the NEXT step, together with that step's
        // T-row change -- ONE read-modify-write pass per copy and topic step (20 B per observed entry) where the
        // schedule shared with the dense flavour takes a read pass and a read-modify-write pass over the row copy
        // (26 B).  The copies then differ by storage roundings only (each term is still applied exactly once to each).
        const double* a2 = c->dw_pending ? c->dwv : c->zeros;          // dw of the previous step: k_wwcol below overwrites it
        const double* b2 = c->T + (i64)(c->dw_pending ? c->dw_topic : t) * c->LD;
        TimedScope ts(c, 3);
        DISPATCH(c, (L::template sp_blk<true, true, true>(c, 0, b1, b2, trow, c->W + (i64)t * c->ldw, a2, c->Ypart, c->Y2part, c->n)));
    } else if (g_wpass_one) {
        // Dense handles: ONE read-modify-write pass per topic step (rri_wrri_kernels.hpp, "one read-modify-write pass").  It folds
        // the W-column change of the step before (still pending) and this step's T-row change into E, writes E, and takes the
        // row products of this W update AND the column sums of the next T row; what those lack -- the term the update below
        // leaves pending -- the next T-row step takes from the mask alone (enqueue_wT_sums).  Nothing pending (T fixed, first
        // step after a rebuild): the pass only reads.
        const bool pend_w = c->dw_pending;
        const double* a2 = pend_w ? c->dwv : c->zeros;                 // k_wwcol below overwrites dwv after the pass has read it
        const double* b2 = c->T + (i64)(pend_w ? c->dw_topic : t) * c->LD;
        const double* wt_t = c->W + (i64)t * c->ldw;
        const double* wnx = c->W + (i64)tn * c->ldw;
        const bool cn = (k > 1) && !c->prm.fix_T;
        if (!pend_w && b1 == c->zeros) {
            if (cn) DISPATCH(c, (L::template wpass<true, true, false, false>(c, trow, wnx, c->zeros, c->zeros, nullptr, nullptr)));
            else DISPATCH(c, (L::template wpass<true, false, false, false>(c, trow, nullptr, c->zeros, c->zeros, nullptr, nullptr)));
        } else if (cn) DISPATCH(c, (L::template wpass<true, true, true, true>(c, trow, wnx, wt_t, b1, a2, b2)));
        else DISPATCH(c, (L::template wpass<true, false, true, true>(c, trow, nullptr, wt_t, b1, a2, b2)));
    } else {
        DISPATCH(c, (L::template wpass<true, false, false, false>(c, trow, nullptr, c->W + (i64)t * c->ldw, b1, nullptr, nullptr)));
    }
    {
        TimedScope ts(c, 1);
        hipLaunchKernelGGL(k_wwcol, dim3(c->nwb256), dim3(256), 0, c->stream, c->W, c->ldw, (int)c->n, k, t,
                           (const double*)c->Ypart, (const double*)c->Y2part, c->npanels, c->wold, c->dwv, c->Gpart,
                           kparams(c), (const DevState*)c->st);
    }
    const bool carry_next = (k > 1) && !c->prm.fix_T;
    const double* wn = c->W + (i64)tn * c->ldw;
    if (sp_merged) {
        TimedScope ts(c, 3);          // the column copy: both terms of this step, the column sums of the next topic
        if (carry_next) DISPATCH(c, (L::template sp_blk<true, true, true>(c, 1, c->wold, c->dwv, wn, b1, trow, c->Zpart, c->Z2part, c->LD)));
        else DISPATCH(c, (L::template sp_blk<false, true, true>(c, 1, c->wold, c->dwv, wn, b1, trow, c->Zpart, c->Z2part, c->LD)));
        c->dw_pending = true;
        c->dw_topic = t;
    } else if (g_wpass_one) {
        c->dw_pending = true;     // dwv x T[t,:] under the mask: folded into E by the pass of the next step
        c->dw_topic = t;
    } else if (carry_next) DISPATCH(c, (L::template wpass<false, true, true, true>(c, nullptr, wn, c->wold, b1, c->dwv, trow)));
    else DISPATCH(c, (L::template wpass<false, false, true, true>(c, nullptr, nullptr, c->wold, b1, c->dwv, trow)));
    int ns = sweep, np = t + 1;
    if (np == k) { np = 0; ns = sweep + 1; }
    if (defer_check) {   // row-sharded: the verdict needs the global column sum; it rides on the next topic's all-reduce
        c->pending_wcheck = true;
        c->pending_wcheck_topic = t;
    } else {
        wcheck_now(c, t, ns, np);
    }
    c->carry_valid = carry_next;
    c->carry_topic = tn;
}

// ---- which XCD gets which tile: the two speeds of the read-modify-write passes -------------------------------------------------
// The pass over a handle's stored residual runs at one of two speeds -- 1.35 against 1.50-1.57 ms for the rank-one update of
// BASELINE config 3, 1.39-1.41 against 1.55 ms for the weighted one-pass step -- and rounds 2-4 could only report which one a
// process had caught.  What decides it is which XCD is dealt which tile: the workgroups of a launch go round-robin over the 8
// XCDs, and rotating the tiles by ONE inside every group of 8 workgroups flips the speed -- every odd rotation fast and every
// even one slow, or the other way round in another process (tools/xcc_mode_probe.py, profiles/r04_xcc_mode_probe.log: the
// buffer's placement against the memory side's interleave, one would think; nothing a process can read).  So a handle that
// keeps a residual TIMES both before its first sweep: per rotation three null updates (a = 0: the residual is rewritten with
// its own values, bit for bit), the last two timed -- ~10 ms once per handle -- and keeps the faster.  The rotation changes
// which workgroup computes a tile and nothing in any sum: the results are the same bits whichever wins.
// (The read-only pass over X differs by ~1 % between the two: calibrated the same way, on X, for handles of the Gram form.)
hipError_t big_malloc(void** p, size_t bytes);
void calibrate_rot(rri_ctx* c) {
    if (c->rot_done) return;
    c->rot_done = true;
    if (g_pass_rot >= 0 || !g_rot_cal) return;
    const bool resid = resid_sched(c);
    const bool wdense = c->weighted && !c->sparse;
    const bool plain = !c->weighted && !resid && !c->sparse && !c->prm.fix_T;      // the Gram form: the read-only pass over X
    if (!(resid || wdense || plain) || (double)c->n * (double)c->d < 1.0e8 || c->npanels * c->nrb < 64) return;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (hipEventCreate(&e0) != hipSuccess) { (void)hipGetLastError(); return; }
    if (hipEventCreate(&e1) != hipSuccess) { (void)hipGetLastError(); (void)hipEventDestroy(e0); return; }
    if (plain) {
        // the read-only pass moves by ~1 % with the rotation (0.643 against 0.650 ms at BASELINE config 3, the same way in every
        // round of three processes): the same calibration on X, 8 passes once per handle
        float best = -1.0f;
        int best_rot = 0;
        for (int rot = -1; rot < 2; ++rot) {                 // -1: two warm passes
            c->rot_x = std::max(rot, 0);
            for (int rep = 0; rep < (rot < 0 ? 2 : 3); ++rep) {
                if (rot >= 0 && rep == 1) (void)hipEventRecord(e0, c->stream);
                DISPATCH(c, (L::template pass<true, true>(c, 0, 0)));
            }
            if (rot < 0) continue;
            (void)hipEventRecord(e1, c->stream);
            float ms = 0.0f;
            if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess) { best = -1.0f; break; }
            if (getenv("RRI_ROT_DEBUG")) fprintf(stderr, "rri: tile rotation %d: %.4f ms per read-only pass\n", rot, ms / 2.0f);
            if (best < 0.0f || ms < best) { best = ms; best_rot = rot; }
        }
        c->rot_x = best > 0.0f ? best_rot : 0;
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        c->carry_valid = false;
        c->carry_topic = -1;
        return;
    }
    auto refresh = [&]() {
        if (resid) r_refresh(c);
        else w_refresh(c);
    };
    if (!c->resid_valid) refresh();
    // one buffer: the better of the rotations 0 and 1 (or 0 .. RRI_ROT_CAL-1), by null updates; < 0: a launch or a wait failed
    auto time_buffer = [&](int* rot_out) -> float {
        float best = -1.0f;
        const int nrot = g_rot_cal > 1 ? std::min(g_rot_cal, 8) : 2;
        for (int rot = 0; rot < nrot; ++rot) {
            c->rot_r = rot;
            for (int rep = 0; rep < 3; ++rep) {
                if (rep == 1) (void)hipEventRecord(e0, c->stream);
                if (resid) {
                    DISPATCH(c, {
                        typename L::Upd u;
                        u.a = c->zeros; u.b = c->zeros; u.a2 = c->zeros; u.b2 = c->zeros; u.b2sub = c->zeros;
                        L::rank_update(c, c->E, c->LD, u, c->T, c->W);
                    });
                } else {
                    DISPATCH(c, (L::template wpass<true, true, true, true>(c, c->T, c->W, c->zeros, c->zeros, c->zeros, c->zeros)));
                }
            }
            (void)hipEventRecord(e1, c->stream);
            float ms = 0.0f;
            if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess) return -1.0f;
            if (getenv("RRI_ROT_DEBUG")) fprintf(stderr, "rri: tile rotation %d: %.4f ms per null update\n", rot, ms / 2.0f);
            if (best < 0.0f || ms < best) { best = ms; *rot_out = rot; }
        }
        return best;
    };
    // The level under the parity belongs to the BUFFER, not to the process (six residuals alive in one process: one at 1.34 ms,
    // one at 1.42, four at 1.50-1.53 whatever the rotation; physically contiguous memory: always 1.54 --
    // profiles/r04_rmw_buffer_probe.log).  RRI_RMW_SHOP=n (diagnostics, default 1 = off) tries a large residual in up to n
    // places -- a new allocation while the ones before it are still held, the residual rebuilt into it, timed -- until one runs
    // 7 % faster than the slowest seen, keeps the best and frees the others: on a box whose buffers are slow it found none in
    // 40 places (10 processes), which is why it is not the default.
    const size_t ebytes = (size_t)c->n * (size_t)c->LD * (c->dtype == RRI_F32 ? 4 : 8);
    const int places = ebytes >= ((size_t)1 << 30) ? std::max(1, g_rmw_shop) : 1;
    std::vector<void*> held;
    void* best_e = c->E;
    int best_rot = 0, rot = 0;
    c->rot_r = 0;
    for (int rep = 0; rep < 2; ++rep) {        // (the first passes of a process run ~5 % slow: not the rotation's doing)
        if (resid) {
            DISPATCH(c, {
                typename L::Upd u;
                u.a = c->zeros; u.b = c->zeros; u.a2 = c->zeros; u.b2 = c->zeros; u.b2sub = c->zeros;
                L::rank_update(c, c->E, c->LD, u, c->T, c->W);
            });
        } else {
            DISPATCH(c, (L::template wpass<true, true, true, true>(c, c->T, c->W, c->zeros, c->zeros, c->zeros, c->zeros)));
        }
    }
    float best_ms = time_buffer(&rot), worst_ms = best_ms;
    best_rot = rot;
    for (int a = 1; a < places && best_ms > 0.0f; ++a) {
        size_t fr = 0, tot = 0;
        if (hipMemGetInfo(&fr, &tot) != hipSuccess || fr < 2 * ebytes + ((size_t)4 << 30)) break;
        void* p = nullptr;
        if (big_malloc(&p, ebytes) != hipSuccess) { (void)hipGetLastError(); break; }
        if (c->LD != c->d) (void)hipMemsetAsync(p, 0, ebytes, c->stream);      // pad columns stay zero
        held.push_back(c->E);
        c->E = p;
        refresh();
        const float ms = time_buffer(&rot);
        if (getenv("RRI_ROT_DEBUG")) fprintf(stderr, "rri: residual buffer %d: %.4f ms (the first %.4f)\n", a, ms / 2.0f, worst_ms / 2.0f);
        if (ms < 0.0f) break;
        worst_ms = std::max(worst_ms, ms);
        if (ms < best_ms) { best_ms = ms; best_rot = rot; best_e = c->E; }
        if (best_ms < 0.93f * worst_ms) break;
    }
    held.push_back(c->E);
    for (void* p : held)
        if (p != best_e) (void)hipFree(p);
    c->E = best_e;                 // (its content: the residual of the current W, T -- the null updates rewrote it with itself)
    c->rot_r = best_ms > 0.0f ? best_rot : 0;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    c->carry_valid = false;          // the null updates left their own row products and column sums in the scratch arrays
    c->carry_topic = -1;
}

// ---- T fixed: the W half of all topics of a sweep as one launch (k_wsweep_rows) ---------------------------------------
bool wsweep_ok(const rri_ctx* c) {
    return g_wsweep && c->prm.fix_T && !c->prm.fix_W && !c->weighted && !c->sparse && c->k >= 1 &&
           wsweep_lds_bytes(c->k) <= 150 * 1024;
}
// topics [t0, k) of sweep `sweep`; false: a buffer could not be had (the caller takes the launch-per-topic schedule)
bool enqueue_wsweep(rri_ctx* c, int sweep, int t0) {
    const int k = c->k;
    const size_t f8 = sizeof(double);
    if (!c->Gfull && hipMalloc((void**)&c->Gfull, (size_t)k * k * f8) != hipSuccess) { c->Gfull = nullptr; (void)hipGetLastError(); return false; }
    if (!c->Wsweep0 && hipMalloc((void**)&c->Wsweep0, (size_t)k * c->ldw * f8) != hipSuccess) { c->Wsweep0 = nullptr; (void)hipGetLastError(); return false; }
    if (!c->wsum_part && hipMalloc((void**)&c->wsum_part, (size_t)k * c->nwb * f8) != hipSuccess) { c->wsum_part = nullptr; (void)hipGetLastError(); return false; }
    if (!c->wsums && hipMalloc((void**)&c->wsums, (size_t)k * f8) != hipSuccess) { c->wsums = nullptr; (void)hipGetLastError(); return false; }
    static bool attr_set[64] = {};
    const int dv = c->device & 63;
    if (!attr_set[dv]) {
        if (hipFuncSetAttribute((const void*)k_wsweep_rows, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024) != hipSuccess) { (void)hipGetLastError(); return false; }
        attr_set[dv] = true;
    }
    if (c->pending_wcheck) {     // (a column check left by steps before T was fixed)
        wcheck_now(c, c->pending_wcheck_topic, sweep, t0);
        c->pending_wcheck = false;
    }
    if (!c->q_valid) {           // X T^T: once per T, reused by every topic and every sweep
        DISPATCH(c, L::xtt(c));
        c->q_valid = true;
    }
    if (!c->gfull_valid) {
        hipLaunchKernelGGL(k_gram, dim3(k, k), dim3(256), 0, c->stream, (const double*)c->T, c->LD, c->d, k, c->Gfull);
        c->gfull_valid = true;
    }
    {
        TimedScope ts(c, 1);
        hipLaunchKernelGGL(k_wsweep_rows, dim3(c->nwb), dim3(64), wsweep_lds_bytes(k), c->stream, c->W, c->Wsweep0, c->ldw, (int)c->n, k,
                           t0, (const double*)c->Qt, (const double*)c->Gfull, c->wsum_part, c->nwb, c->XYpart, (i64)c->xy_stride,
                           kparams(c), (const DevState*)c->st);
        if (!c->comm)
            hipLaunchKernelGGL(k_wsweep_verdict, dim3(1), dim3(1024), 0, c->stream, (const double*)c->wsum_part, c->nwb, c->wsums,
                               (const double*)c->Gfull, k, t0, sweep, 3, kparams(c), c->st);
        else {      // row-sharded: the column sums of all k topics in ONE all-reduce (the launch-per-topic schedule takes k)
            hipLaunchKernelGGL(k_wsweep_verdict, dim3(1), dim3(1024), 0, c->stream, (const double*)c->wsum_part, c->nwb, c->wsums,
                               (const double*)c->Gfull, k, t0, sweep, 1, kparams(c), c->st);
            comm_allreduce(c, c->wsums, k);
            hipLaunchKernelGGL(k_wsweep_verdict, dim3(1), dim3(1024), 0, c->stream, (const double*)c->wsum_part, c->nwb, c->wsums,
                               (const double*)c->Gfull, k, t0, sweep, 2, kparams(c), c->st);
        }
        hipLaunchKernelGGL(k_wsweep_repair, dim3((unsigned)((c->n + 255) / 256)), dim3(256), 0, c->stream, c->W, (const double*)c->Wsweep0,
                           c->ldw, (int)c->n, k, (const DevState*)c->st);
    }
    for (int t = t0; t < k; ++t) LK::note_xy(c, t, c->nwb * WCOL_TILES);
    c->skip_row_finish = false;
    c->carry_valid = false;
    c->resid_valid = false;
    return true;
}

// sweeps [cur .. s_end) of the current call
void enqueue_range(rri_ctx* c, Cursor cur, int s_end) {
    const int k = c->k;
    if (cur.sweep < s_end) calibrate_rot(c);
    if (c->weighted) {
        for (int s = cur.sweep; s < s_end; ++s) {
            const int t0 = (s == cur.sweep) ? cur.topic : 0;
            const int sa = s;
            for (int t = t0; t < k; ++t) {
                const int ph = (s == cur.sweep && t == cur.topic) ? cur.phase : 0;
                // once per sweep (and after resets), unless rri_objective has just rebuilt it from the same W, T
                if (!c->resid_valid || (t == 0 && ph == 0 && !c->resid_fresh)) w_refresh(c);
                if (!c->prm.fix_T && ph == 0) enqueue_wT_half(c, sa, t);
                // the column verdict rides on the next T-row step where that step can take it: row-sharded (the all-reduce), or one
                // device with few row blocks (k_wtrow_small)
                // (or k_wreduce: every T-row step of this loop can take it now)
                if (!c->prm.fix_W) enqueue_wW_half(c, sa, t, !c->prm.fix_T);
            }
        }
        return;
    }
    if (resid_sched(c)) {
        for (int s = cur.sweep; s < s_end; ++s) {
            const int t0 = (s == cur.sweep) ? cur.topic : 0;
            const int sa = s;
            for (int t = t0; t < k; ++t) {
                const int ph = (s == cur.sweep && t == cur.topic) ? cur.phase : 0;
                // rebuilt once per sweep (and after anything changed W or T from outside), unless rri_objective has
                // just stored it for the same W, T
                if (!c->resid_valid || (t == 0 && ph == 0 && !c->resid_fresh)) r_refresh(c);
                if (ph == 0) enqueue_rT_half(c, sa, t, false);
                enqueue_rW_half(c, sa, t);
            }
        }
        return;
    }
    for (int s = cur.sweep; s < s_end; ++s) {
        const int t0 = (s == cur.sweep) ? cur.topic : 0;
        const int sa = s;
        if (wsweep_ok(c) && enqueue_wsweep(c, sa, t0)) continue;      // T fixed: the W half of every topic in one launch
        for (int t = t0; t < k; ++t) {
            const int ph = (s == cur.sweep && t == cur.topic) ? cur.phase : 0;
            if (!c->prm.fix_T && ph == 0) enqueue_T_half(c, sa, t, c->prm.fix_W != 0);
            if (!c->prm.fix_W) enqueue_W_half(c, sa, t);
        }
    }
}

// ---- register-resident persistent sweeps (rri_onchip_kernels.hpp) ---------------------------------------------
struct OnchipGeom { int CG, RG, rows_wg, rpw, NA, kS, G; size_t shmem; };
constexpr int ONCHIP_UNTIL_CAP = 512;   // sweeps per launch of rri_sweep_until (a slot row of 256 shares each)
constexpr int ONCHIP_MAX_RPW = 20;    // rows per wave held in registers (float4 each): 32 spills at 256 VGPRs
// beyond ONCHIP_SMALL_K topics the k-term dots keep 8 terms per lane and the registers take fewer resident rows WITHOUT a spill
// (round 4, compiler's resource report: plain 18 rows / 253 VGPRs, with the projection 14 rows / 249; 20 rows spilled 10 / 16 -- and
// 68 in round 3's build: the allocation moves with every edit, the report of `python -m rri_nmf_amd.build --report` is the record)
constexpr int ONCHIP_MAX_RPW_K64 = 18, ONCHIP_MAX_RPW_K64_PROJ = 14;
int onchip_rpw_cap(const rri_ctx* c, bool proj) {
    const int cap = c->k > ONCHIP_SMALL_K ? (proj ? ONCHIP_MAX_RPW_K64_PROJ : ONCHIP_MAX_RPW_K64) : ONCHIP_MAX_RPW;
    return c->dtype == RRI_F32 ? cap : cap / 2;       // float64 X: 8 registers per row and lane
}
bool onchip_geometry(const rri_ctx* c, OnchipGeom* g) {
    const bool proj = !LK::light(c);                   // the projection stage stages the whole T row per worker: d <= 1024
    if (c->LD > (proj ? 1024 : 2048) || c->n_cu < 1) return false;
    g->G = std::min(c->n_cu, 256);                     // the workers take 16 partials per lane group: G <= 16 ONCHIP_PG
    g->CG = c->LD <= 256 ? 1 : c->LD <= 512 ? 2 : c->LD <= 1024 ? 4 : 8;
    g->RG = ONCHIP_WAVES / g->CG;
    g->rows_wg = (int)((c->n + g->G - 1) / g->G);
    g->rpw = (g->rows_wg + g->RG - 1) / g->RG;
    g->NA = (int)((c->LD + ONCHIP_CWA - 1) / ONCHIP_CWA);      // workgroups that also own a column slice of T
    g->kS = c->k | 1;                                  // odd row stride of the LDS copy of W: no bank conflicts down a column
    if (g->rpw > onchip_rpw_cap(c, proj) || g->NA > 64 || g->NA > g->G || (i64)g->rows_wg * g->kS > 6144) return false;
    const size_t doubles = (size_t)g->rows_wg * g->kS + (size_t)c->k * ONCHIP_CWA + (c->k + 2) + (c->k + 1) +
                           (size_t)ONCHIP_PG * ONCHIP_CWA + (size_t)g->CG * g->rows_wg + 2 * (size_t)g->rows_wg +
                           (size_t)ONCHIP_WAVES * 256 + (size_t)ONCHIP_WAVES * 8 * 72 + 1024 + 40;
    g->shmem = doubles * sizeof(double);
    return g->shmem <= 150 * 1024;
}
// A persistent launch that gave up means the device is shared with somebody whose grids collide with ours: every handle of the
// process keeps off the persistent path until this time (steady clock, ns), so that a process that makes a handle per nmf()
// call does not walk into the same wait again and again (tools/onchip_two_processes.py).
std::atomic<long long> g_onchip_backoff_until{0};
long long steady_now_ns() { return (long long)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
// what the persistent kernel covers: the unweighted flavour, either storage type, both halves free (with or without the per-step
// simplex projection of T), 2 <= k <= ONCHIP_MAX_K, on one device
bool onchip_ok(const rri_ctx* c) {
    OnchipGeom g;
    // a handle that fell back tries the persistent path again once its own back-off has run out (a burst on another stream or
    // process must not cost a long-lived handle the launch-bound speed-up for good); eligibility therefore depends on the clock
    if (c->onchip_off && steady_now_ns() < c->onchip_off_until) return false;
    return g_onchip && steady_now_ns() >= g_onchip_backoff_until.load() && !c->weighted && !c->explicit_resid && !c->comm && !c->sparse && c->k >= 2 &&
           c->k <= ONCHIP_MAX_K && !c->prm.fix_W && !c->prm.fix_T && c->ldx % c->VN == 0 && ((uintptr_t)c->X) % 16 == 0 &&
           onchip_geometry(c, &g);
}
// Two persistent grids on one device (two handles on two streams) must not each hold a part of the CUs while waiting for
// the rest: inside a process EVERY persistent launch -- whatever its instantiation -- waits for the one before it on the same
// device (one mutex, one event per device, both at file scope: round 2 kept them inside the launch template, one set per
// instantiation, so an fp32 and a float64 handle were not ordered against each other).  Across processes nothing orders
// two grids; there the bounded polls end the wait and the call falls back (run_and_collect).
// hipLaunchCooperativeKernel is NOT used: it gives no stronger residency than a plain launch of a grid checked against the
// occupancy query (same admission, same queue), costs 15-19 us per launch, and a process that has used it dies in the HIP
// runtime's own exit handler under rocprofv3 ("the exit-time fault"; RRI_ONCHIP_COOP=1 keeps it reachable for
// tools/exit_probe).
std::mutex g_onchip_mu;
hipEvent_t g_onchip_last[64] = {};
int g_onchip_coop = 0;   // RRI_ONCHIP_COOP=1 (diagnostics): hipLaunchCooperativeKernel instead of the plain launch
hipError_t onchip_ordered_launch(rri_ctx* c, const void* fn, int grid, size_t shmem, const OnchipArgs& a) {
    std::lock_guard<std::mutex> lock(g_onchip_mu);
    const int dv = c->device & 63;
    if (!g_onchip_last[dv] && hipEventCreateWithFlags(&g_onchip_last[dv], hipEventDisableTiming) != hipSuccess) g_onchip_last[dv] = nullptr;
    if (g_onchip_last[dv]) (void)hipStreamWaitEvent(c->stream, g_onchip_last[dv], 0);
    OnchipArgs copy = a;
    void* args[] = {(void*)&copy};
    hipError_t le;
    if (g_onchip_coop) le = hipLaunchCooperativeKernel(fn, dim3(grid), dim3(ONCHIP_THREADS), args, (unsigned)shmem, c->stream);
    else le = hipLaunchKernel(fn, dim3(grid), dim3(ONCHIP_THREADS), args, shmem, c->stream);
    if (le == hipSuccess) le = hipGetLastError();
    if (le == hipSuccess && g_onchip_last[dv]) (void)hipEventRecord(g_onchip_last[dv], c->stream);
    return le;
}
template <typename SX, int RPW, bool DBG = false, bool PROJ = false, int KT = 3>
hipError_t onchip_launch(rri_ctx* c, const OnchipGeom& g, const OnchipArgs& a) {
    static bool attr_set[64] = {};
    const int dv = c->device & 63;
    const void* fn = (const void*)k_onchip_sweeps<SX, RPW, DBG, PROJ, KT>;
    if (!attr_set[dv]) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024);
        if (e != hipSuccess) { if (getenv("RRI_ONCHIP_DEBUG")) fprintf(stderr, "rri: hipFuncSetAttribute\n"); return e; }
        attr_set[dv] = true;
    }
    int per_cu = 0;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_onchip_sweeps<SX, RPW, DBG, PROJ, KT>, ONCHIP_THREADS, g.shmem);
    if (getenv("RRI_ONCHIP_DEBUG")) fprintf(stderr, "rri: occupancy %d per CU (%s), %d CUs\n", per_cu, hipGetErrorString(e), c->n_cu);
    if (e != hipSuccess) return e;
    if ((i64)per_cu * c->n_cu < g.G) return hipErrorCooperativeLaunchTooLarge;     // the hand-overs need every workgroup resident
    return onchip_ordered_launch(c, fn, g.G, g.shmem, a);
}
// sweeps [cur .. run_total) in one launch; false: not launched (the caller takes the launch-per-phase schedule)
bool enqueue_onchip(rri_ctx* c, Cursor cur) {
    OnchipGeom g;
    if (!onchip_geometry(c, &g)) return false;
    const int k = c->k;
    if (!c->mkZ) {
        if (hipMalloc((void**)&c->mkZ, (size_t)2 * g.G * c->LD * 8) != hipSuccess) { c->mkZ = nullptr; return false; }
        if (hipMalloc((void**)&c->mkG, (size_t)2 * g.G * (k + 2) * 8) != hipSuccess) return false;
        if (hipMalloc((void**)&c->mkP, (size_t)2 * 64 * (k + 1) * 8) != hipSuccess) return false;
        if (hipMalloc((void**)&c->mkbar, (size_t)(128 + g.G) * sizeof(unsigned)) != hipSuccess) return false;
        if (hipMalloc((void**)&c->mkX, (size_t)2 * c->LD * 8) != hipSuccess) return false;
        if (hipMalloc((void**)&c->mkT, (size_t)2 * c->LD * 8) != hipSuccess) return false;
        if (hipMalloc((void**)&c->objE, (size_t)ONCHIP_UNTIL_CAP * 256 * 8) != hipSuccess) return false;
        if (hipMalloc((void**)&c->objhist, (size_t)ONCHIP_UNTIL_CAP * 8) != hipSuccess) return false;
        if (hipMalloc((void**)&c->objdec, (size_t)ONCHIP_UNTIL_CAP * 8) != hipSuccess) return false;
        (void)hipMemsetAsync(c->mkZ, 0, (size_t)2 * g.G * c->LD * 8, c->stream);
        (void)hipMemsetAsync(c->mkG, 0, (size_t)2 * g.G * (k + 2) * 8, c->stream);
        (void)hipMemsetAsync(c->mkP, 0, (size_t)2 * 64 * (k + 1) * 8, c->stream);
    }
    if (!c->mkG || !c->mkP || !c->mkbar || !c->mkX || !c->mkT || !c->objE || !c->objhist || !c->objdec) return false;
    if (!c->Wsafe && hipMalloc((void**)&c->Wsafe, (size_t)k * c->ldw * 8) != hipSuccess) { c->Wsafe = nullptr; return false; }
    if (!c->Tsafe && hipMalloc((void**)&c->Tsafe, (size_t)k * c->LD * 8) != hipSuccess) { c->Tsafe = nullptr; return false; }
    (void)hipMemsetAsync(c->mkbar, 0, (size_t)(128 + g.G) * sizeof(unsigned), c->stream);
    OnchipArgs a{};
    a.X = c->X; a.ldx = c->ldx; a.n = (int)c->n; a.d = (int)c->d; a.LD = (int)c->LD; a.k = k;
    a.Wt = c->W; a.ldw = c->ldw; a.T = c->T; a.ldt = c->LD;
    a.mkZ = c->mkZ; a.mkG = c->mkG; a.mkP = c->mkP; a.mkX = c->mkX; a.mkT = c->mkT; a.objE = c->objE; a.xyp = c->XYpart; a.xy_stride = c->xy_stride; a.bar = c->mkbar;
    a.G = g.G; a.NA = g.NA; a.rows_wg = g.rows_wg; a.CG = g.CG; a.RG = g.RG; a.kS = g.kS;
    a.s0 = cur.sweep; a.t0 = cur.topic; a.ph0 = cur.phase; a.s_end = c->run_total;
    a.skip_row_finish = c->skip_row_finish ? 1 : 0;
    a.spin_limit = 2000000u;            // polls of ~1 us: a grid that stands still for seconds gives up (HALT_ERR_GRID_SYNC)
    a.entry_spin_limit = 40000u;        // the hand-over at kernel entry: a grid that is not resident as a whole shows within ~40 ms
    if (const char* e = getenv("RRI_ONCHIP_SPIN_LIMIT")) a.spin_limit = a.entry_spin_limit = (unsigned)std::max(0, atoi(e));   // tests: 0 = give up at once
    if (const char* e = getenv("RRI_ONCHIP_ENTRY_SPIN_LIMIT")) a.entry_spin_limit = (unsigned)std::max(0, atoi(e));
    a.jitter = 0u;
    if (const char* e = getenv("RRI_ONCHIP_JITTER")) a.jitter = (unsigned)strtoul(e, nullptr, 10);     // tests: seeded sleeps before every exchange store and first poll
    a.fail_step = -1;
    if (const char* e = getenv("RRI_ONCHIP_FAIL_STEP")) a.fail_step = atoi(e);       // tests: give up inside the run, at this topic step of the launch
    // the last sweep of the launch runs from its topic 0: its objective can be left behind (see eacc in the kernel)
    a.track = (g_onchip_obj && (c->run_total - 1 > cur.sweep || (cur.topic == 0 && cur.phase == 0))) ? 1 : 0;
    if (c->until.active && cur.topic == 0 && cur.phase == 0 && c->run_total - cur.sweep <= ONCHIP_UNTIL_CAP && c->x_sq_valid) {
        a.track = 2;
        a.objhist = c->objhist; a.dec = c->objdec;
        a.obj_prev = c->until.prev; a.stop_scale = c->until.scale; a.half_xsq = 0.5 * c->x_sq;
        // "not written" = NaN: the sweeps of the call whose objective the kernel did not leave are told by that
        (void)hipMemsetAsync(c->objhist, 0xFF, (size_t)(c->run_total - cur.sweep) * 8, c->stream);
    }
    a.nap_eighths = 5;
    if (const char* e = getenv("RRI_ONCHIP_NAP_EIGHTHS")) a.nap_eighths = std::min(7, std::max(0, atoi(e)));      // diagnostics
    a.p = kparams(c); a.st = c->st;
    a.dbg = nullptr;
    if (getenv("RRI_ONCHIP_TIMING")) {           // diagnostics: per-section ticks of the last launch, printed at the next one
        static long long* dbg = nullptr;
        if (!dbg && hipMalloc((void**)&dbg, 32 * sizeof(long long)) != hipSuccess) dbg = nullptr;
        if (dbg) {
            long long h[32];
            if (c->onchip_launches > 0 && hipMemcpy(h, dbg, sizeof h, hipMemcpyDeviceToHost) == hipSuccess) {
                static const char* names[14] = {"carries arrive", "A rest", "next carry + workers arrive", "mark absent", "row dots", "W update", "carry_post", "to next step",
                                                "closed form", "-", "slices arrive", "-", "project", "-"};
                for (int w = 0; w < 2; ++w) {
                    fprintf(stderr, "rri on-chip sections, workgroup %s (us total):", w == 0 ? "0 (worker)" : "G-1");
                    for (int i = 0; i < 13; ++i) fprintf(stderr, " %s %.1f;", names[i], h[16 * w + i] * 0.01);
                    fprintf(stderr, "\n");
                }
            }
            (void)hipMemsetAsync(dbg, 0, 32 * sizeof(long long), c->stream);
            a.dbg = dbg;
        }
    }
    // what a launch that gives up is rolled back to (0.9 MB at 10000 x 1000, k = 20: two copies of a few microseconds)
    (void)hipMemcpyAsync(c->Wsafe, c->W, (size_t)k * c->ldw * 8, hipMemcpyDeviceToDevice, c->stream);
    (void)hipMemcpyAsync(c->Tsafe, c->T, (size_t)k * c->LD * 8, hipMemcpyDeviceToDevice, c->stream);
    c->onchip_saved_skip = c->skip_row_finish;
    hipError_t e;
    {
        TimedLaunch tl{nullptr, nullptr};
        const bool timed = c->timing > 0 && c->timed[0].size() < 400000;
        if (timed) { tl.a = get_event(c); tl.b = get_event(c); (void)hipEventRecord(tl.a, c->stream); }
        const bool proj = !LK::light(c);          // project_T_each_iter with a t_row_sum: the topic-model instantiation
        if (c->k > ONCHIP_SMALL_K) {              // k-term dots of 8 terms per lane (k <= 64); no diagnostics build
            if (c->dtype == RRI_F64) {
                if (proj) e = g.rpw <= 4 ? onchip_launch<double, 4, false, true, 8>(c, g, a) : onchip_launch<double, ONCHIP_MAX_RPW_K64_PROJ / 2, false, true, 8>(c, g, a);
                else e = g.rpw <= 4 ? onchip_launch<double, 4, false, false, 8>(c, g, a) : onchip_launch<double, ONCHIP_MAX_RPW_K64 / 2, false, false, 8>(c, g, a);
            } else if (proj) e = g.rpw <= 8 ? onchip_launch<float, 8, false, true, 8>(c, g, a) : onchip_launch<float, ONCHIP_MAX_RPW_K64_PROJ, false, true, 8>(c, g, a);
            else e = g.rpw <= 8 ? onchip_launch<float, 8, false, false, 8>(c, g, a) : onchip_launch<float, ONCHIP_MAX_RPW_K64, false, false, 8>(c, g, a);
        } else if (c->dtype == RRI_F64) {         // 8 registers per row and lane: half the rows of the fp32 instantiations
            if (proj) e = g.rpw <= 4 ? onchip_launch<double, 4, false, true>(c, g, a) : onchip_launch<double, ONCHIP_MAX_RPW / 2, false, true>(c, g, a);
            else e = g.rpw <= 4 ? onchip_launch<double, 4>(c, g, a) : onchip_launch<double, ONCHIP_MAX_RPW / 2>(c, g, a);
        } else if (proj && a.dbg) e = g.rpw <= 8 ? onchip_launch<float, 8, true, true>(c, g, a) : onchip_launch<float, ONCHIP_MAX_RPW, true, true>(c, g, a);
        else if (proj) e = g.rpw <= 8 ? onchip_launch<float, 8, false, true>(c, g, a) : onchip_launch<float, ONCHIP_MAX_RPW, false, true>(c, g, a);
        else if (a.dbg) e = g.rpw <= 8 ? onchip_launch<float, 8, true>(c, g, a) : onchip_launch<float, ONCHIP_MAX_RPW, true>(c, g, a);
        else if (g.rpw <= 8) e = onchip_launch<float, 8>(c, g, a);
        else e = onchip_launch<float, ONCHIP_MAX_RPW>(c, g, a);
        if (timed) { (void)hipEventRecord(tl.b, c->stream); c->timed[0].push_back(tl); }
    }
    if (e != hipSuccess) {
        if (getenv("RRI_ONCHIP_DEBUG")) fprintf(stderr, "rri: on-chip sweep not launched (%s): rows/wg %d, rows/wave %d, LDS %zu B\n", hipGetErrorString(e), g.rows_wg, g.rpw, g.shmem);
        (void)hipGetLastError();
        return false;
    }
    c->onchip_launches += 1;
    c->onchip_in_flight = true;
    if (const char* path = getenv("RRI_ONCHIP_LOG")) {      // tests: which configurations took this path (one line per launch)
        if (FILE* f = fopen(path, "a")) {
            fprintf(f, "%lld %lld %d %s %s sweeps %d..%d\n", (long long)c->n, (long long)c->d, k, c->dtype == RRI_F32 ? "f32" : "f64",
                    LK::light(c) ? "plain" : "simplex", cur.sweep, c->run_total);
            fclose(f);
        }
    }
    // what the launch-per-phase schedule would find after these sweeps: no carried sums, nothing pending (the kernel
    // ran the last column check itself); the objective's cross terms are complete when the last sweep ran from topic 0
    const bool whole_last = c->run_total - 1 > cur.sweep || (cur.topic == 0 && cur.phase == 0);
    c->carry_valid = false; c->carry_topic = -1;
    c->pending_wcheck = false;
    c->resid_valid = false; c->q_valid = false; c->gfull_valid = false;
    c->skip_row_finish = false;
    c->xy_run = whole_last ? k : -1;
    c->xy_rows = g.G;
    c->xy_valid = whole_last;
    c->obj_track_valid = false;
    c->obj_track_pending = a.track != 0;
    return true;
}

void enqueue_final_check(rri_ctx* c, int sweep_arg) {
    if (c->pending_wcheck) {  // last column of the call: report it in this call
        wcheck_now(c, c->pending_wcheck_topic, sweep_arg, 0);
        c->pending_wcheck = false;
    }
}

void enqueue_from(rri_ctx* c, Cursor cur) {
    if (cur.sweep < c->run_total && onchip_ok(c) && enqueue_onchip(c, cur)) return;
    // rri_sweep_until without the persistent kernel: nobody applies the stop rule between the sweeps, so the call ends after
    // the sweep it is in and the caller decides
    if (c->until.active && cur.sweep < c->run_total) c->run_total = cur.sweep + 1;
    enqueue_range(c, cur, c->run_total);
    enqueue_final_check(c, c->run_total);
}

rri_status read_state(rri_ctx* c, DevState* out) {
    HIPCHK(c, hipMemcpyAsync(out, c->st, sizeof(DevState), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->comm_status != RRI_OK) {   // a collective of the sequence just run failed: its text is in c->err
        const rri_status r = c->comm_status;
        c->comm_status = RRI_OK;
        invalidate(c);
        c->pending_wcheck = false;
        return r;
    }
    return RRI_OK;
}

rri_status status_from_halt(rri_ctx* c, const DevState& s, int32_t* sweeps_done) {
    if (s.halt == 0) {
        c->paused = false;
        if (sweeps_done) *sweeps_done = c->run_total;
        return RRI_OK;
    }
    invalidate(c);
    c->pending_wcheck = false;
    if (s.halt == HALT_EVENT_STOP) {      // rri_sweep_until: the stop rule held after sweep halt_sweep - 1; nothing of the next one is stored
        c->paused = false;
        if (sweeps_done) *sweeps_done = s.halt_sweep;
        return RRI_OK;
    }
    if (s.halt > 0) {
        // rri_sweep_until: an event ends the call with the sweep it interrupts (position 0: the column check of the sweep before
        // -- nothing of sweep halt_sweep has begun); that sweep's objective is the caller's to take after the event is resolved
        if (c->until.active) c->run_total = std::min(c->run_total, (s.halt == HALT_EVENT_RESET_W && s.halt_pos == 0) ? s.halt_sweep : s.halt_sweep + 1);
        c->paused = true;
        c->pending.kind = s.halt == HALT_EVENT_RESET_T ? RRI_EVENT_RESET_T : RRI_EVENT_RESET_W;
        c->pending.topic = s.halt_topic;
        c->pending.sweep = s.halt_sweep;
        c->pending.resume_topic = s.halt_pos;
        if (s.halt == HALT_EVENT_RESET_T) c->resume_at = Cursor{s.halt_sweep, s.halt_topic, 1};
        else c->resume_at = Cursor{s.halt_sweep, s.halt_pos, 0};
        if (sweeps_done) *sweeps_done = s.halt_sweep;
        return RRI_PAUSED;
    }
    c->paused = false;
    if (sweeps_done) *sweeps_done = s.halt_sweep;
    switch (s.halt) {
        case HALT_ERR_UNBOUNDED:
            return fail(c, RRI_ERR_UNBOUNDED, "Minimum objective is unbounded (topic %d)", s.halt_topic);
        case HALT_ERR_W_COL_ZERO:
            return fail(c, RRI_ERR_W_COL_ZERO, "W[:, t] sums to 0 (topic %d)", s.halt_topic);
        case HALT_ERR_NOT_IMPLEMENTED:
            return fail(c, RRI_ERR_NOT_IMPLEMENTED, "s=%g is not yet implemented", c->prm.t_row_sum);
        case HALT_ERR_GRID_SYNC:
            return fail(c, RRI_ERR_HIP, "the persistent sweep's workgroups could not synchronise (not all resident); RRI_ONCHIP=0 selects the launch-per-phase schedule");
        default:
            return fail(c, RRI_ERR_INVALID, "unknown device status %d", s.halt);
    }
}

rri_status clear_halt(rri_ctx* c) {
    HIPCHK(c, hipMemsetAsync(c->st, 0, 32, c->stream));  // halt, halt_topic, halt_sweep, halt_pos, tmode, proj_iters
    return RRI_OK;
}

// The big read-modify-write buffers (the residual R / E).  RRI_MALLOC_CONTIGUOUS=1 (diagnostics, tools/rmw_place.py) asks the
// runtime for physically contiguous memory (hipExtMallocWithFlags, hipDeviceMallocContiguous).
int g_malloc_contig = 0;
hipError_t big_malloc(void** p, size_t bytes) {
    if (g_malloc_contig) {
        hipError_t e = hipExtMallocWithFlags(p, bytes, hipDeviceMallocContiguous);
        if (e == hipSuccess) return e;
        (void)hipGetLastError();
        if (getenv("RRI_ONCHIP_DEBUG")) fprintf(stderr, "rri: contiguous allocation of %zu bytes refused (%s); plain hipMalloc\n", bytes, hipGetErrorString(e));
    }
    return hipMalloc(p, bytes);
}

rri_status ready(rri_ctx* c) {
    if (!c->have_X || !c->have_W || !c->have_T || !c->have_params)
        return fail(c, RRI_ERR_INVALID, "X, W, T and params must be set before stepping");
    if (c->weighted && !c->have_M) return fail(c, RRI_ERR_INVALID, "weighted handle without a mask");
    return RRI_OK;
}

}  // namespace

// ====================================================================================================
extern "C" {

uint32_t rri_abi_version(void) { return RRI_ABI_VERSION; }

const char* rri_last_error(const rri_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

rri_status rri_create(rri_ctx** out, int64_t n, int64_t d, int32_t k, int32_t dtype, int32_t weighted,
                      int32_t device, void* stream) {
    if (!out) return RRI_ERR_INVALID;
    *out = nullptr;
    if (n < 1 || d < 1 || k < 1) return fail(nullptr, RRI_ERR_INVALID, "need n,d,k >= 1 (got %lld,%lld,%d)", n, d, k);
    if (dtype != RRI_F32 && dtype != RRI_F64) return fail(nullptr, RRI_ERR_INVALID, "dtype must be RRI_F32/RRI_F64");
    if (n > 2000000000LL || d > 2000000000LL) return fail(nullptr, RRI_ERR_INVALID, "n, d must fit int32");
    if (k > RRI_MAX_K) return fail(nullptr, RRI_ERR_UNSUPPORTED, "k=%d > %d not supported on the device path", k, RRI_MAX_K);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return fail(nullptr, RRI_ERR_HIP, "no HIP device available (librri_hip needs an MI355X)");
    if (device < 0 || device >= ndev) return fail(nullptr, RRI_ERR_INVALID, "device %d out of range [0,%d)", device, ndev);
    if (weighted < 0 || weighted > 3)
        return fail(nullptr, RRI_ERR_INVALID, "weighted must be RRI_UNWEIGHTED, RRI_WEIGHTED_DENSE, RRI_WEIGHTED_SPARSE or RRI_UNWEIGHTED_RESIDUAL");
    rri_ctx* c = new rri_ctx();
    const bool explicit_resid = weighted == RRI_UNWEIGHTED_RESIDUAL;
    if (explicit_resid) weighted = RRI_UNWEIGHTED;   // the same flavour of the algorithm, another schedule of its passes
    c->explicit_resid = explicit_resid;
    c->n = n; c->d = d; c->k = k; c->dtype = dtype; c->weighted = weighted; c->device = device;
    c->sparse = weighted == RRI_WEIGHTED_SPARSE;
    c->kp = (int)round_up(k, 8);
    c->es = dtype == RRI_F32 ? 4 : 8;
    c->VN = (int)(16 / c->es);
    // the switches are per process and read again by every rri_create: an unset variable means the default, not "what
    // the last handle was created with"
    g_pass_unroll = 8; g_pass_unroll_upd = 16; g_pass_nt = -1; g_pass_rs = 1; g_obj_direct = 0; g_pass_interleave = -1;
    g_wsweep = 1;
    g_resid_split = 0;
    if (const char* e = getenv("RRI_RESID_SPLIT")) g_resid_split = std::max(0, atoi(e));
    if (const char* e = getenv("RRI_WSWEEP")) g_wsweep = atoi(e) != 0;
    g_trow_small = 1; g_resid_mfma = 1; g_side_jobs = 1; g_onchip = 1; g_onchip_obj = 1; g_wpass_il = -1;
    g_wpass_uc = 8;
    g_wpass_one = 1;
    g_wpass_occ4 = 1;
    if (const char* e = getenv("RRI_WPASS_OCC4")) g_wpass_occ4 = atoi(e) != 0;
    g_wmcorr_cols = 1;
    g_wmcorr_wgs = 16;
    g_wnw_mask = 1;
    if (const char* e = getenv("RRI_WNW_MASK")) g_wnw_mask = atoi(e) != 0;
    if (const char* e = getenv("RRI_WMCORR_WGS")) g_wmcorr_wgs = std::min(64, std::max(1, atoi(e)));
    if (const char* e = getenv("RRI_WMCORR_COLS")) g_wmcorr_cols = atoi(e) != 0;
    g_wmcorr_skip = 1;
    if (const char* e = getenv("RRI_WMCORR_SKIP")) g_wmcorr_skip = atoi(e) != 0;
    g_wpass_ud = 4;
    if (const char* e = getenv("RRI_WPASS_UD")) g_wpass_ud = atoi(e) == 8 ? 8 : 4;
    if (const char* e = getenv("RRI_WPASS_ONE")) g_wpass_one = atoi(e) != 0;
    if (const char* e = getenv("RRI_PASS_UNROLL")) { int v = atoi(e); if (v == 4 || v == 8 || v == 16) { g_pass_unroll = v; g_pass_unroll_upd = v; } }
    if (const char* e = getenv("RRI_PASS_NT")) g_pass_nt = atoi(e) != 0 ? 1 : 0;
    g_pass_dma = 0;
    if (const char* e = getenv("RRI_PASS_DMA")) g_pass_dma = atoi(e) != 0 ? 1 : 0;
    g_pass_dma_sub = 0;
    if (const char* e = getenv("RRI_PASS_DMA_SUB")) g_pass_dma_sub = std::max(0, atoi(e));
    if (const char* e = getenv("RRI_PASS_RS")) g_pass_rs = atoi(e) != 0;
    if (const char* e = getenv("RRI_OBJ_DIRECT")) g_obj_direct = atoi(e) != 0;
    if (const char* e = getenv("RRI_PASS_IL")) g_pass_interleave = atoi(e) != 0 ? 1 : 0;
    if (const char* e = getenv("RRI_TROW_SMALL")) g_trow_small = atoi(e) != 0;
    if (const char* e = getenv("RRI_RESID_MFMA")) g_resid_mfma = atoi(e) != 0;
    if (const char* e = getenv("RRI_SIDE_JOBS")) g_side_jobs = atoi(e) != 0;
    if (const char* e = getenv("RRI_ONCHIP")) g_onchip = atoi(e) != 0;
    if (const char* e = getenv("RRI_ONCHIP_OBJ")) g_onchip_obj = atoi(e) != 0;
    g_onchip_coop = 0;
    g_pass_rot = -1;
    if (const char* e = getenv("RRI_PASS_ROT")) g_pass_rot = atoi(e) & 7;
    g_rot_cal = 1;
    g_rmw_shop = 1;
    if (const char* e = getenv("RRI_RMW_SHOP")) g_rmw_shop = std::min(12, std::max(1, atoi(e)));
    if (const char* e = getenv("RRI_ROT_CAL")) g_rot_cal = std::max(0, atoi(e));      // 0 off, 1 rotations {0, 1}, n > 1: rotations 0 .. n-1
    g_malloc_contig = 0;
    if (const char* e = getenv("RRI_MALLOC_CONTIGUOUS")) g_malloc_contig = atoi(e) != 0;
    if (const char* e = getenv("RRI_ONCHIP_COOP")) g_onchip_coop = atoi(e) != 0;
    if (const char* e = getenv("RRI_WPASS_IL")) g_wpass_il = atoi(e) != 0 ? 1 : 0;
    if (const char* e = getenv("RRI_WPASS_UC")) g_wpass_uc = atoi(e) == 4 ? 4 : 8;
    c->PW = 64 * c->VN * 4;   // columns per workgroup: 4 waves x (64 lanes x 16 B)
    c->LD = round_up(d, c->VN);
#define CR(call)                                                                                   \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess) {                                                                    \
            fail(nullptr, RRI_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_));             \
            rri_destroy(c);                                                                        \
            return RRI_ERR_HIP;                                                                    \
        }                                                                                          \
    } while (0)
    CR(hipSetDevice(device));
    CR(hipDeviceGetAttribute(&c->n_cu, hipDeviceAttributeMultiprocessorCount, device));
    if (stream) c->stream = (hipStream_t)stream;
    else { CR(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)); c->own_stream = true; }

    // geometry of the streaming pass
    c->npanels = (int)((c->LD + c->PW - 1) / c->PW);
    // Workgroups: a multiple of 512 (2 per CU: with 525 on 256 CUs some CUs get three and the pass waits for them),
    // as many as possible up to 2048 while each still walks ~192 rows or more -- with 49 rows each (20000 x 5000 at
    // 2048 workgroups) ramp-up and tail cost 13 % of the pass (profiles/r01_pass_workgroups_mid_size.log).
    // LDS per workgroup = (5 rows-doubles plain | 11 weighted) * rpb + 4 row-sum tiles (18 KiB): kept under 40 KiB so
    // that 4 workgroups (16 waves) fit a CU's 160 KiB -- with 62 KiB the weighted passes ran at 2 workgroups per CU
    // and 20 % slower.  (The explicit update kernel takes a sixth array and may run at 3 per CU.)
    const i64 rpb_cap = ((40 * 1024 - 4 * 8 * 72 * 8) / ((weighted ? 11 : explicit_resid ? 7 : 5) * 8)) / 16 * 16;
    int rpb_min = 32;
    if (const char* e = getenv("RRI_PASS_MIN_ROWS")) rpb_min = std::max(4, atoi(e));
    i64 rpb = 0;
    if (const char* e = getenv("RRI_PASS_WGS")) {
        const int nrb_t = std::max(1, std::max(1, atoi(e)) / c->npanels);
        rpb = (n + nrb_t - 1) / nrb_t;
    } else {
        // handles whose passes write a residual back (explicit-residual, dense weighted): the read-modify-write pass
        // likes ~8192 workgroups of >= 96 rows (+3 % at C3 for the residual schedule, +6 % for the weighted one)
        // (round 4) the one-pass weighted step: ~16384 workgroups of >= 48 rows -- 1.40 against 1.50 - 1.55 ms per pass at BASELINE
        // config 5, engines made alternately in one process; 24576: the same, 32768: 1.44; the partial column sums grow with the
        // row blocks, +17 us per launch of the T-row chain (profiles/r04_wpass_one_variants.log)
        const bool rmw = explicit_resid || weighted == RRI_WEIGHTED_DENSE;
        // (round 4, late) the read-only pass: at most 1024 -- which at BASELINE config 3 means the LDS cap below decides, 560 rows per
        // workgroup and 1790 workgroups instead of 496 rows and 2020: 0.647-0.653 against 0.662-0.664 ms in four processes of five,
        // equal in the fifth (N-way in one process, tools/env_ab.py; the row count is a stride between concurrent streams and
        // the pass is sensitive to it: 544 rows, between the two, 0.695 ms -- profiles/r04_pass_rows_per_workgroup.log)
        const int total_max = weighted == RRI_WEIGHTED_DENSE ? 16384 : rmw ? 8192 : 1024;
        const i64 rows_min = weighted == RRI_WEIGHTED_DENSE ? 48 : rmw ? 96 : 192;
        for (int total = total_max; total >= 512 && rpb == 0; total -= 512) {
            const int nrb_t = std::max(1, total / c->npanels);
            const i64 r = (n + nrb_t - 1) / nrb_t;
            if (r >= rows_min || total == 512) rpb = r;
        }
    }
    rpb = std::max<i64>(rpb, rpb_min);
    rpb = std::min<i64>(round_up(rpb, 16), rpb_cap);
    c->rpb = (int)rpb;
    c->nrb = (int)((n + rpb - 1) / rpb);
    if (c->sparse) {
        // no dense pass: the row copy is cut into column blocks (= Ypart panels), the column copy into row blocks
        // (= Zpart rows); block widths so that three factor tables of a block fit SP_BLOCK_BYTES of LDS
        i64 block_bytes = SP_BLOCK_BYTES;
        if (const char* e = getenv("RRI_SP_BLOCK_KB")) block_bytes = std::min<i64>(SP_BLOCK_BYTES, std::max(8, atoi(e)) * 1024LL);
        const i64 cap = block_bytes / (3 * (dtype == RRI_F32 ? 4 : 8));
        for (int w = 0; w < 2; ++w) {
            rri_ctx::SpCopy& cp = c->sp[w];
            cp.gdim = w == 0 ? d : n;
            cp.nseg = w == 0 ? n : d;
            cp.nblk = (int)((cp.gdim + cap - 1) / cap);
            cp.bw = (int)round_up((cp.gdim + cp.nblk - 1) / cp.nblk, 64);
        }
        c->npanels = c->sp[0].nblk;
        c->nrb = c->sp[1].nblk;
    }
    c->nwb = (int)((n + 64 * WCOL_TILES - 1) / (64 * WCOL_TILES));   // k_wcol blocks = rows of Gpart
    c->nwb256 = (int)((n + 255) / 256);
    c->ntb = (int)((d + 127) / 128);
    c->ntb32 = (int)((c->LD + 31) / 32);
    c->tpart_n = c->ntb;
    c->ldw = n;
    c->nsplit = (int)std::max<i64>(1, std::min<i64>(8, d / 2048));
    c->red_elems = round_up(std::max<i64>(c->LD + (i64)GRAM_SLICES * (k + 2), weighted ? 2 * c->LD + 2 : 0), 4);

    const size_t f8 = sizeof(double);
    const size_t es_x = c->es;
    CR(hipMalloc((void**)&c->W, (size_t)k * c->ldw * f8));
    CR(hipMalloc((void**)&c->T, (size_t)k * c->LD * f8));
    CR(hipMemsetAsync(c->T, 0, (size_t)k * c->LD * f8, c->stream));
    CR(hipMalloc((void**)&c->Ypart, (size_t)c->npanels * n * f8));
    CR(hipMemsetAsync(c->Ypart, 0, (size_t)c->npanels * n * f8, c->stream));
    CR(hipMalloc((void**)&c->Zpart, (size_t)c->nrb * c->LD * f8));
    CR(hipMemsetAsync(c->Zpart, 0, (size_t)c->nrb * c->LD * f8, c->stream));
    const size_t grows = (size_t)std::max(c->nwb, c->nrb);      // k_wcol leaves a row per 64-row tile, the fused pass one per row block
    c->gpart_n = c->nwb;
    c->xy_stride = (int)std::max<size_t>(std::max<size_t>(grows, (size_t)c->nwb * WCOL_TILES), (size_t)std::max(c->n_cu, 1));   // the on-chip sweep leaves one per CU
    CR(hipMalloc((void**)&c->Gpart, grows * (k + 2) * sizeof(double)));
    CR(hipMemsetAsync(c->Gpart, 0, grows * (k + 2) * sizeof(double), c->stream));
    CR(hipMalloc((void**)&c->XYpart, (size_t)k * c->xy_stride * f8));
    CR(hipMemsetAsync(c->XYpart, 0, (size_t)k * c->xy_stride * f8, c->stream));
    CR(hipMalloc((void**)&c->red, (size_t)c->red_elems * f8));
    CR(hipMemsetAsync(c->red, 0, (size_t)c->red_elems * f8, c->stream));
    c->own_red = true;
    CR(hipMalloc((void**)&c->xraw, (size_t)c->LD * f8));
    CR(hipMemsetAsync(c->xraw, 0, (size_t)c->LD * f8, c->stream));
    const size_t ttn = (size_t)std::max(c->nsplit, c->ntb32);   // k_tgram: nsplit column slices; k_trow_small: one per 32 columns
    c->ttpart_n = c->nsplit;
    CR(hipMalloc((void**)&c->Ttpart, ttn * k * f8));
    CR(hipMemsetAsync(c->Ttpart, 0, ttn * k * f8, c->stream));
    CR(hipMalloc((void**)&c->Qt, (size_t)k * c->ldw * f8));
    const size_t ntp = (size_t)std::max(c->ntb, c->ntb32);
    CR(hipMalloc((void**)&c->tpart, ntp * sizeof(double)));
    CR(hipMemsetAsync(c->tpart, 0, ntp * sizeof(double), c->stream));
    CR(hipMalloc((void**)&c->tpart_idx, ntp * sizeof(i64)));
    CR(hipMalloc((void**)&c->normpart, 256 * 3 * sizeof(double)));
    CR(hipMalloc((void**)&c->objbuf, (size_t)(2 * k * k + k) * sizeof(double)));
    CR(hipMalloc((void**)&c->dtmp, 16 * sizeof(double)));
    CR(hipMalloc((void**)&c->itmp, 16 * sizeof(i64)));
    if (weighted) {
        const i64 zn = std::max<i64>(c->LD, n);
        if (!c->sparse) {
            CR(big_malloc(&c->E, (size_t)n * c->LD * es_x));
            // k_resid writes the d real columns only; the passes stream all LD: the pad columns must hold zeros
            // (recycled memory there once held NaN patterns, which fmax(numer, 0) turned into zero rows of W)
            if (c->LD != d) CR(hipMemsetAsync(c->E, 0, (size_t)n * c->LD * es_x, c->stream));
        }
        CR(hipMalloc((void**)&c->Y2part, (size_t)c->npanels * n * f8));
        CR(hipMemsetAsync(c->Y2part, 0, (size_t)c->npanels * n * f8, c->stream));
        CR(hipMalloc((void**)&c->Z2part, (size_t)c->nrb * c->LD * f8));
        CR(hipMemsetAsync(c->Z2part, 0, (size_t)c->nrb * c->LD * f8, c->stream));
        if (!c->sparse) {
            c->cpart_rows = (int)std::max<i64>(256, (n + 2047) / 2048);
            CR(hipMalloc((void**)&c->Cpart, (size_t)c->cpart_rows * c->LD * f8));
            CR(hipMemsetAsync(c->Cpart, 0, (size_t)c->cpart_rows * c->LD * f8, c->stream));
            CR(hipMalloc((void**)&c->N2part, (size_t)c->cpart_rows * c->LD * f8));
            CR(hipMemsetAsync(c->N2part, 0, (size_t)c->cpart_rows * c->LD * f8, c->stream));
        }
        CR(hipMalloc((void**)&c->dtv, (size_t)c->LD * f8));
        CR(hipMalloc((void**)&c->dwv, (size_t)n * f8));
        CR(hipMalloc((void**)&c->wold, (size_t)n * f8));
        CR(hipMalloc((void**)&c->zeros, (size_t)zn * f8));
        CR(hipMemsetAsync(c->zeros, 0, (size_t)zn * f8, c->stream));
        CR(hipMemsetAsync(c->dtv, 0, (size_t)c->LD * f8, c->stream));
        if (c->sparse) {
            CR(hipMalloc((void**)&c->sp_Tt, (size_t)d * c->kp * f8));
            CR(hipMemsetAsync(c->sp_Tt, 0, (size_t)d * c->kp * f8, c->stream));
        }
    }
    if (explicit_resid) {
        const i64 zn = std::max<i64>(c->LD, n);
        CR(big_malloc(&c->E, (size_t)n * c->LD * es_x));
        if (c->LD != d) CR(hipMemsetAsync(c->E, 0, (size_t)n * c->LD * es_x, c->stream));   // pad columns stay zero
        CR(hipMalloc((void**)&c->dwv, (size_t)n * f8));
        CR(hipMemsetAsync(c->dwv, 0, (size_t)n * f8, c->stream));
        CR(hipMalloc((void**)&c->told, (size_t)c->LD * f8));
        CR(hipMemsetAsync(c->told, 0, (size_t)c->LD * f8, c->stream));
        CR(hipMalloc((void**)&c->zeros, (size_t)zn * f8));
        CR(hipMemsetAsync(c->zeros, 0, (size_t)zn * f8, c->stream));
    }
    CR(hipMalloc((void**)&c->st, sizeof(DevState)));
    CR(hipMemsetAsync(c->st, 0, sizeof(DevState), c->stream));
    // opt in to large dynamic LDS where a kernel needs it
    if (dtype == RRI_F32) CR(LaunchX<float>::set_attrs());
    else CR(LaunchX<double>::set_attrs());
    CR(hipStreamSynchronize(c->stream));
#undef CR
    *out = c;
    return RRI_OK;
}

rri_status rri_destroy(rri_ctx* c) {
    if (!c) return RRI_OK;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->own_X) (void)hipFree(c->X);
    if (c->own_M) (void)hipFree(c->M);
    void* bufs[] = {c->E, (void*)c->W, (void*)c->T, (void*)c->Wprev, (void*)c->Tprev,                     (void*)c->Ypart, (void*)c->Zpart, (void*)c->xraw, (void*)c->Ttpart, (void*)c->Gpart,
                    (void*)c->tpart, (void*)c->tpart_idx, (void*)c->rowobj, (void*)c->rowpos, (void*)c->normpart,
                    (void*)c->dtmp, (void*)c->itmp, (void*)c->resetT, (void*)c->resetW, (void*)c->st, (void*)c->Y2part,
                    (void*)c->Z2part, (void*)c->Mbits, (void*)c->Qt, (void*)c->dtv, (void*)c->dwv, (void*)c->wold, (void*)c->zeros,
                    (void*)c->XYpart, (void*)c->objbuf, (void*)c->told, (void*)c->Cpart, (void*)c->N2part, (void*)c->Mcols, (void*)c->Gfull, (void*)c->Wsweep0, (void*)c->wsum_part, (void*)c->wsums, (void*)c->ctail, (void*)c->cand, (void*)c->mkZ, (void*)c->mkG, (void*)c->mkP, (void*)c->mkX, (void*)c->mkT, (void*)c->objE, (void*)c->objhist, (void*)c->objdec, (void*)c->mkbar, (void*)c->Wsafe, (void*)c->Tsafe, (void*)c->sp_rowptr, (void*)c->sp_col, c->sp_x, c->sp_e, (void*)c->sp_Tt,
                    (void*)c->sp[0].segptr, (void*)c->sp[0].idx, c->sp[0].val, (void*)c->sp[0].perm, (void*)c->sp[0].work,
                    (void*)c->sp[1].segptr, (void*)c->sp[1].idx, c->sp[1].val, (void*)c->sp[1].perm, (void*)c->sp[1].work};
    for (void* b : bufs)
        if (b) (void)hipFree(b);
    if (c->own_red && c->red) (void)hipFree(c->red);
    for (int i = 0; i < 4; ++i)
        for (auto& tl : c->timed[i]) { (void)hipEventDestroy(tl.a); (void)hipEventDestroy(tl.b); }
    for (auto e : c->event_pool) (void)hipEventDestroy(e);
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return RRI_OK;
}

// ---- data ------------------------------------------------------------------------------------------
rri_status rri_upload_X(rri_ctx* c, const void* host, int64_t ld, int32_t host_dtype) {
    CHECK_CTX(c);
    if (c->sparse) return fail(c, RRI_ERR_UNSUPPORTED, "a sparse-pattern handle takes its data through rri_upload_observed_csr");
    HIPCHK(c, hipSetDevice(c->device));
    if (c->X && !c->own_X) c->X = nullptr;
    if (!c->X) {
        HIPCHK(c, hipMalloc(&c->X, (size_t)c->n * c->LD * c->es));
        c->own_X = true;
        if (c->LD != c->d) HIPCHK(c, hipMemsetAsync(c->X, 0, (size_t)c->n * c->LD * c->es, c->stream));
    }
    c->ldx = c->LD;
    rri_status s = to_device(c, host, ld, host_dtype, c->X, c->ldx, c->n, c->d, c->dtype);
    if (s == RRI_OK) { c->have_X = true; invalidate(c); c->q_valid = false; c->gfull_valid = false; c->x_sq_valid = false; }
    return s;
}

rri_status rri_upload_mask(rri_ctx* c, const void* host, int64_t ld, int32_t host_dtype) {
    CHECK_CTX(c);
    if (c->sparse) return fail(c, RRI_ERR_UNSUPPORTED, "a sparse-pattern handle takes its data through rri_upload_observed_csr");
    if (!c->weighted) return fail(c, RRI_ERR_INVALID, "handle was not created with weighted=1");
    HIPCHK(c, hipSetDevice(c->device));
    if (c->M && !c->own_M) c->M = nullptr;
    if (!c->M) {
        HIPCHK(c, hipMalloc(&c->M, (size_t)c->n * c->LD * c->es));
        c->own_M = true;
        if (c->LD != c->d) HIPCHK(c, hipMemsetAsync(c->M, 0, (size_t)c->n * c->LD * c->es, c->stream));
    }
    c->ldm = c->LD;
    rri_status s = to_device(c, host, ld, host_dtype, c->M, c->ldm, c->n, c->d, c->dtype);
    if (s == RRI_OK) {
        c->have_M = true;
        invalidate(c);
        DISPATCH(c, s = L::pack_mask_if_binary(c));
        if (s != RRI_OK) return fail(c, s, "packing the mask failed");
        if (c->Mbits && c->own_M) { (void)hipFree(c->M); c->M = nullptr; c->own_M = false; }   // bits replace it
    }
    return s;
}

namespace {
struct CsrDev {   // device copies of the host CSR arrays of one call
    i64* indptr = nullptr;
    int* indices = nullptr;
    void* data = nullptr;
    ~CsrDev() { (void)hipFree(indptr); (void)hipFree(indices); (void)hipFree(data); }
};
rri_status csr_to_device(rri_ctx* c, const int64_t* indptr, const int32_t* indices, const void* data, int64_t nnz,
                         int32_t data_dtype, CsrDev& out) {
    if (!indptr || (nnz > 0 && (!indices || !data)) || nnz < 0) return fail(c, RRI_ERR_INVALID, "bad CSR arrays");
    if (data_dtype != RRI_F32 && data_dtype != RRI_F64) return fail(c, RRI_ERR_INVALID, "bad CSR data dtype");
    if (indptr[0] != 0 || indptr[c->n] != nnz) return fail(c, RRI_ERR_INVALID, "indptr does not span nnz");
    for (i64 r = 0; r < c->n; ++r)
        if (indptr[r + 1] < indptr[r]) return fail(c, RRI_ERR_INVALID, "indptr not monotone at row %lld", r);
    for (i64 p = 0; p < nnz; ++p)
        if (indices[p] < 0 || indices[p] >= c->d) return fail(c, RRI_ERR_INVALID, "column index out of range at %lld", p);
    const size_t ds = data_dtype == RRI_F32 ? 4 : 8;
    HIPCHK(c, hipMalloc((void**)&out.indptr, (size_t)(c->n + 1) * sizeof(i64)));
    HIPCHK(c, hipMalloc((void**)&out.indices, (size_t)std::max<i64>(nnz, 1) * sizeof(int)));
    HIPCHK(c, hipMalloc(&out.data, (size_t)std::max<i64>(nnz, 1) * ds));
    HIPCHK(c, hipMemcpyAsync(out.indptr, indptr, (size_t)(c->n + 1) * sizeof(i64), hipMemcpyHostToDevice, c->stream));
    if (nnz > 0) {
        HIPCHK(c, hipMemcpyAsync(out.indices, indices, (size_t)nnz * sizeof(int), hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(out.data, data, (size_t)nnz * ds, hipMemcpyHostToDevice, c->stream));
    }
    return RRI_OK;
}
}  // namespace

rri_status rri_upload_X_csr(rri_ctx* c, const int64_t* indptr, const int32_t* indices, const void* data,
                            int64_t nnz, int32_t data_dtype) {
    CHECK_CTX(c);
    if (c->sparse) return fail(c, RRI_ERR_UNSUPPORTED, "a sparse-pattern handle takes its data through rri_upload_observed_csr");
    HIPCHK(c, hipSetDevice(c->device));
    CsrDev dv;
    rri_status s = csr_to_device(c, indptr, indices, data, nnz, data_dtype, dv);
    if (s != RRI_OK) return s;
    if (c->X && !c->own_X) c->X = nullptr;
    if (!c->X) {
        HIPCHK(c, hipMalloc(&c->X, (size_t)c->n * c->LD * c->es));
        c->own_X = true;
    }
    c->ldx = c->LD;
    HIPCHK(c, hipMemsetAsync(c->X, 0, (size_t)c->n * c->LD * c->es, c->stream));
    const unsigned nb = (unsigned)((c->n + 3) / 4);
    if (c->dtype == RRI_F32) {
        if (data_dtype == RRI_F32) hipLaunchKernelGGL((k_csr_scatter<float, float>), dim3(nb), dim3(256), 0, c->stream, dv.indptr, dv.indices, (const float*)dv.data, c->n, (float*)c->X, c->ldx);
        else hipLaunchKernelGGL((k_csr_scatter<double, float>), dim3(nb), dim3(256), 0, c->stream, dv.indptr, dv.indices, (const double*)dv.data, c->n, (float*)c->X, c->ldx);
    } else {
        if (data_dtype == RRI_F32) hipLaunchKernelGGL((k_csr_scatter<float, double>), dim3(nb), dim3(256), 0, c->stream, dv.indptr, dv.indices, (const float*)dv.data, c->n, (double*)c->X, c->ldx);
        else hipLaunchKernelGGL((k_csr_scatter<double, double>), dim3(nb), dim3(256), 0, c->stream, dv.indptr, dv.indices, (const double*)dv.data, c->n, (double*)c->X, c->ldx);
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->have_X = true;
    invalidate(c);
    c->q_valid = false; c->gfull_valid = false;
    c->x_sq_valid = false;
    return RRI_OK;
}

rri_status rri_upload_mask_csr_pattern(rri_ctx* c, const int64_t* indptr, const int32_t* indices, const void* data,
                                       int64_t nnz, int32_t data_dtype) {
    CHECK_CTX(c);
    if (c->sparse) return fail(c, RRI_ERR_UNSUPPORTED, "a sparse-pattern handle takes its data through rri_upload_observed_csr");
    if (!c->weighted) return fail(c, RRI_ERR_INVALID, "handle was not created with weighted=1");
    HIPCHK(c, hipSetDevice(c->device));
    CsrDev dv;
    rri_status s = csr_to_device(c, indptr, indices, data, nnz, data_dtype, dv);
    if (s != RRI_OK) return s;
    if (c->M && c->own_M) (void)hipFree(c->M);
    c->M = nullptr;
    c->own_M = false;
    c->ldm = c->LD;
    if (c->Mbits) { (void)hipFree(c->Mbits); c->Mbits = nullptr; }
    if (c->Mcols) { (void)hipFree(c->Mcols); c->Mcols = nullptr; }
    c->mcols_tried = false;
    c->ldb = (c->LD + 3) / 4;
    const size_t words = (size_t)((c->n + 7) / 8) * c->ldb;
    HIPCHK(c, hipMalloc((void**)&c->Mbits, words * sizeof(unsigned)));
    HIPCHK(c, hipMemsetAsync(c->Mbits, 0, words * sizeof(unsigned), c->stream));
    const unsigned nb = (unsigned)((c->n + 3) / 4);
    if (data_dtype == RRI_F32) hipLaunchKernelGGL((k_csr_pattern_bits<float>), dim3(nb), dim3(256), 0, c->stream, dv.indptr, dv.indices, (const float*)dv.data, c->n, c->Mbits, c->ldb);
    else hipLaunchKernelGGL((k_csr_pattern_bits<double>), dim3(nb), dim3(256), 0, c->stream, dv.indptr, dv.indices, (const double*)dv.data, c->n, c->Mbits, c->ldb);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->have_M = true;
    invalidate(c);
    return RRI_OK;
}

rri_status rri_upload_observed_csr(rri_ctx* c, const int64_t* indptr, const int32_t* indices, const void* values,
                                   int64_t nnz, int32_t data_dtype) {
    CHECK_CTX(c);
    if (!c->sparse) return fail(c, RRI_ERR_INVALID, "handle was not created with weighted=RRI_WEIGHTED_SPARSE");
    if (nnz >= 2147483647LL) return fail(c, RRI_ERR_UNSUPPORTED, "more than 2^31-1 observed entries");
    HIPCHK(c, hipSetDevice(c->device));
    CsrDev dv;   // validates the arrays; its device copies of indptr / indices become the CSR copy
    rri_status s = csr_to_device(c, indptr, indices, values, nnz, data_dtype, dv);
    if (s != RRI_OK) return s;
    for (i64 r = 0; r < c->n; ++r)
        for (i64 p = indptr[r] + 1; p < indptr[r + 1]; ++p)
            if (indices[p] <= indices[p - 1])
                return fail(c, RRI_ERR_INVALID, "column indices of row %lld are not strictly increasing", r);
    i64 longest_row = 0;
    for (i64 r = 0; r < c->n; ++r) longest_row = std::max<i64>(longest_row, (i64)(indptr[r + 1] - indptr[r]));
    void* old[] = {(void*)c->sp_rowptr, (void*)c->sp_col, c->sp_x, c->sp_e};
    for (void* b : old)
        if (b) (void)hipFree(b);
    c->sp_rowptr = dv.indptr; dv.indptr = nullptr;
    c->sp_col = dv.indices; dv.indices = nullptr;
    c->sp_x = nullptr; c->sp_e = nullptr;
    const size_t cnt = (size_t)std::max<i64>(nnz, 1);
    HIPCHK(c, hipMalloc(&c->sp_x, cnt * c->es));
    HIPCHK(c, hipMalloc(&c->sp_e, cnt * c->es));
    if (nnz > 0) {   // values -> storage type (dv.data holds them in the caller's type)
        const bool hf = data_dtype == RRI_F32, df = c->dtype == RRI_F32;
        if (hf && df) launch_convert<float, float, false>(c, dv.data, nnz, c->sp_x, nnz, 1, nnz);
        else if (hf) launch_convert<float, double, false>(c, dv.data, nnz, c->sp_x, nnz, 1, nnz);
        else if (df) launch_convert<double, float, false>(c, dv.data, nnz, c->sp_x, nnz, 1, nnz);
        else launch_convert<double, double, false>(c, dv.data, nnz, c->sp_x, nnz, 1, nnz);
    }
    // the two blocked copies: counting sort on the host, stable, so offsets ascend inside a segment
    int target_items = std::max(1, c->n_cu);
    if (const char* e = getenv("RRI_SP_ITEMS")) target_items = std::max(1, atoi(e));
    for (int w = 0; w < 2; ++w) {
        rri_ctx::SpCopy& cp = c->sp[w];
        void* oldc[] = {(void*)cp.segptr, (void*)cp.idx, cp.val, (void*)cp.perm, (void*)cp.work};
        for (void* b : oldc)
            if (b) (void)hipFree(b);
        cp.segptr = nullptr; cp.idx = nullptr; cp.val = nullptr; cp.perm = nullptr; cp.work = nullptr;
        const i64 nseg = cp.nseg, stride = nseg + 1;
        std::vector<i64> sp((size_t)cp.nblk * stride, 0);
        // count: entry (r, j) lives in block (gather index / bw), segment (the other index)
        for (i64 r = 0; r < c->n; ++r)
            for (i64 p = indptr[r]; p < indptr[r + 1]; ++p) {
                const i64 j = indices[p];
                const i64 g = w == 0 ? j : r, sgm = w == 0 ? r : j;
                sp[(size_t)((g / cp.bw) * stride + sgm + 1)] += 1;
            }
        i64 run = 0;   // exclusive prefix over (block, segment); every block row keeps nseg + 1 pointers.
        // Segments are padded to multiples of 4 entries (k_sp_blk moves quads).
        for (int b = 0; b < cp.nblk; ++b) {
            i64* row = sp.data() + (size_t)b * stride;
            row[0] = run;
            for (i64 q = 1; q <= nseg; ++q) {
                run += (row[q] + 3) / 4 * 4;
                row[q] = run;
            }
        }
        cp.count = run;
        const size_t cntp = (size_t)std::max<i64>(run, 4);
        std::vector<unsigned short> bidx(cntp, (unsigned short)cp.bw);      // pads: the zero slot of the factor tables
        std::vector<int> perm(cntp, -1);
        {
            std::vector<i64> fill((size_t)cp.nblk * nseg);
            for (int b = 0; b < cp.nblk; ++b)
                for (i64 q = 0; q < nseg; ++q) fill[(size_t)b * nseg + q] = sp[(size_t)b * stride + q];
            for (i64 r = 0; r < c->n; ++r)
                for (i64 p = indptr[r]; p < indptr[r + 1]; ++p) {
                    const i64 j = indices[p];
                    const i64 g = w == 0 ? j : r, sgm = w == 0 ? r : j;
                    const i64 b = g / cp.bw;
                    const i64 q = fill[(size_t)(b * nseg + sgm)]++;
                    bidx[(size_t)q] = (unsigned short)(g - b * cp.bw);
                    perm[(size_t)q] = (int)p;
                }
        }
        // Work items: runs of segments of one block, at most `target_items` in all and of equal entry count -- ONE round of
        // workgroups (a 1024-thread workgroup with the block's tables per CU).  Round 2 cut "about 3 x 256" items and got
        // 774-780: three rounds of the 256 CUs and a fourth for the last few, a quarter of every launch with the chip idle
        // (profiles/r03_sp_blk_probe.log: 138 -> 115 us per pass).  Every (block, segment) belongs to exactly one item --
        // also the empty ones, whose sums the consumers still read.
        std::vector<SpWork> work;
        {
            std::vector<i64> eb((size_t)cp.nblk);
            i64 total = 0;
            for (int b = 0; b < cp.nblk; ++b) {
                const i64* row = sp.data() + (size_t)b * stride;
                eb[(size_t)b] = row[nseg] - row[0];
                total += eb[(size_t)b];
            }
            const i64 spare = std::max<i64>(0, (i64)target_items - cp.nblk);      // every block needs one item; the rest by share
            for (int b = 0; b < cp.nblk; ++b) {
                const i64* row = sp.data() + (size_t)b * stride;
                i64 items_b = 1 + (total > 0 ? spare * eb[(size_t)b] / total : 0);
                items_b = std::max<i64>(1, std::min<i64>(items_b, eb[(size_t)b] / 4096));      // no items of a few entries
                i64 s0 = 0;
                for (i64 j = 1; j <= items_b && s0 < nseg; ++j) {
                    i64 s1 = nseg;
                    if (j < items_b) {
                        const i64 want = row[0] + eb[(size_t)b] * j / items_b;     // first segment boundary at or past the j-th share
                        s1 = std::lower_bound(row + s0 + 1, row + nseg, want) - row;
                        s1 = std::min<i64>(std::max<i64>(s1, s0 + 1), nseg);
                    }
                    work.push_back(SpWork{b, (int)s0, (int)s1, 0});
                    s0 = s1;
                }
                if (s0 < nseg) work.push_back(SpWork{b, (int)s0, (int)nseg, 0});
            }
        }
        cp.nwork = (int)work.size();
        // lanes per segment: 4 quads of 4 entries per lane and iteration
        const i64 avg = nnz / std::max<i64>(1, (i64)cp.nblk * nseg);
        cp.lps = avg >= 768 ? 64 : avg >= 384 ? 32 : avg >= 192 ? 16 : 8;
        if (const char* e = getenv(w == 0 ? "RRI_SP_LANES_ROW" : "RRI_SP_LANES_COL")) {
            const int v = atoi(e);
            if (v == 8 || v == 16 || v == 32 || v == 64) cp.lps = v;
        }
        HIPCHK(c, hipMalloc((void**)&cp.segptr, sp.size() * sizeof(i64)));
        HIPCHK(c, hipMalloc((void**)&cp.idx, cntp * sizeof(unsigned short)));
        HIPCHK(c, hipMalloc(&cp.val, cntp * c->es));
        HIPCHK(c, hipMemsetAsync(cp.val, 0, cntp * c->es, c->stream));
        HIPCHK(c, hipMalloc((void**)&cp.perm, cntp * sizeof(int)));
        HIPCHK(c, hipMalloc((void**)&cp.work, std::max<size_t>(1, work.size()) * sizeof(SpWork)));
        HIPCHK(c, hipMemcpyAsync(cp.segptr, sp.data(), sp.size() * sizeof(i64), hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(cp.idx, bidx.data(), cntp * sizeof(unsigned short), hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(cp.perm, perm.data(), cntp * sizeof(int), hipMemcpyHostToDevice, c->stream));
        if (!work.empty())
            HIPCHK(c, hipMemcpyAsync(cp.work, work.data(), work.size() * sizeof(SpWork), hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));   // the host vectors go out of scope
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->nnz = nnz;
    c->sp_max_row = (int)longest_row;
    c->have_X = true;
    c->have_M = true;
    invalidate(c);
    return RRI_OK;
}

rri_status rri_bind_X_device(rri_ctx* c, const void* dev, int64_t ld) {
    CHECK_CTX(c);
    if (c->sparse) return fail(c, RRI_ERR_UNSUPPORTED, "a sparse-pattern handle takes its data through rri_upload_observed_csr");
    if (!dev || ld < c->d || (ld * (i64)c->es) % 16 || ((uintptr_t)dev) % 16)
        return fail(c, RRI_ERR_INVALID, "device X must be 16-byte aligned with a 16-byte-multiple row stride >= d");
    if (c->d % c->VN) return fail(c, RRI_ERR_INVALID, "binding device X needs d %% %d == 0 (no pad columns)", c->VN);
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipDeviceSynchronize());   // the memory may have been produced on another stream a moment ago
    if (c->X && c->own_X) (void)hipFree(c->X);
    c->X = const_cast<void*>(dev);
    c->own_X = false;
    c->ldx = ld;
    c->have_X = true;
    invalidate(c);
    c->q_valid = false; c->gfull_valid = false;
    c->x_sq_valid = false;
    return RRI_OK;
}

rri_status rri_bind_mask_device(rri_ctx* c, const void* dev, int64_t ld) {
    CHECK_CTX(c);
    if (c->sparse) return fail(c, RRI_ERR_UNSUPPORTED, "a sparse-pattern handle takes its data through rri_upload_observed_csr");
    if (!c->weighted) return fail(c, RRI_ERR_INVALID, "handle was not created with weighted=1");
    if (!dev || ld < c->d || (ld * (i64)c->es) % 16 || ((uintptr_t)dev) % 16)
        return fail(c, RRI_ERR_INVALID, "device mask must be 16-byte aligned with a 16-byte-multiple row stride >= d");
    if (c->d % c->VN) return fail(c, RRI_ERR_INVALID, "binding a device mask needs d %% %d == 0", c->VN);
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipDeviceSynchronize());   // the mask is read at once (bit-packing): it must be complete
    if (c->M && c->own_M) (void)hipFree(c->M);
    c->M = const_cast<void*>(dev);
    c->own_M = false;
    c->ldm = ld;
    c->have_M = true;
    invalidate(c);
    rri_status ps = RRI_OK;
    DISPATCH(c, ps = L::pack_mask_if_binary(c));
    if (ps != RRI_OK) return fail(c, ps, "packing the mask failed");
    return RRI_OK;
}

rri_status rri_set_W(rri_ctx* c, const void* host, int64_t ld, int32_t host_dtype) {
    CHECK_CTX(c);
    HIPCHK(c, hipSetDevice(c->device));
    rri_status s = to_device(c, host, ld, host_dtype, c->W, c->ldw, c->n, c->k, RRI_F64, true);
    if (s == RRI_OK) { c->have_W = true; invalidate(c); c->pending_wcheck = false; }
    return s;
}
rri_status rri_set_T(rri_ctx* c, const void* host, int64_t ld, int32_t host_dtype) {
    CHECK_CTX(c);
    HIPCHK(c, hipSetDevice(c->device));
    rri_status s = to_device(c, host, ld, host_dtype, c->T, c->LD, c->k, c->d, RRI_F64);
    if (s == RRI_OK) { c->have_T = true; invalidate(c); c->q_valid = false; c->gfull_valid = false; }
    return s;
}
rri_status rri_get_W(rri_ctx* c, void* host, int64_t ld, int32_t host_dtype) {
    CHECK_CTX(c);
    HIPCHK(c, hipSetDevice(c->device));
    return to_host(c, c->W, c->ldw, host, ld, host_dtype, c->n, c->k, RRI_F64, true);
}
rri_status rri_get_T(rri_ctx* c, void* host, int64_t ld, int32_t host_dtype) {
    CHECK_CTX(c);
    HIPCHK(c, hipSetDevice(c->device));
    return to_host(c, c->T, c->LD, host, ld, host_dtype, c->k, c->d, RRI_F64);
}

rri_status rri_set_params(rri_ctx* c, const rri_params* p) {
    CHECK_CTX(c);
    if (!p) return fail(c, RRI_ERR_INVALID, "params is NULL");
    if (p->has_t_row_sum && !(p->t_row_sum > 0)) return fail(c, RRI_ERR_INVALID, "t_row_sum must be > 0");
    if (p->has_w_row_sum && !(p->w_row_sum > 0)) return fail(c, RRI_ERR_INVALID, "w_row_sum must be > 0");
    if (p->reset_method < 0 || p->reset_method > 2) return fail(c, RRI_ERR_INVALID, "bad reset_method");
    if (p->fix_W && p->fix_T) return fail(c, RRI_ERR_INVALID, "fix_W and fix_T together leave nothing to update");
    const bool form_before = c->have_params && resid_sched(c);
    // the objective a persistent sweep left behind has the penalties of THAT launch folded in
    if (c->have_params && (c->prm.reg_w_l1 != p->reg_w_l1 || c->prm.reg_w_l2 != p->reg_w_l2 || c->prm.reg_t_l1 != p->reg_t_l1 ||
                           c->prm.reg_t_l2 != p->reg_t_l2))
        c->obj_track_valid = false;
    c->prm = *p;
    c->have_params = true;
    if (c->explicit_resid && form_before != resid_sched(c)) {
        // the call that follows steps in the other form: the sums carried between calls belong to the form that left them,
        // and a residual the Gram form does not maintain is stale
        invalidate(c);
        c->dw_pending = false;
        c->dt_pending = false;
    }
    return RRI_OK;
}

// ---- the hot path ------------------------------------------------------------------------------------
static rri_status run_and_collect(rri_ctx* c, Cursor from, int32_t* sweeps_done) {
    HIPCHK(c, hipSetDevice(c->device));
    c->onchip_in_flight = false;
    c->obj_track_pending = false;
    enqueue_from(c, from);
    hipError_t le = hipGetLastError();
    if (le != hipSuccess) return fail(c, RRI_ERR_HIP, "kernel launch failed: %s", hipGetErrorString(le));
    DevState s;
    rri_status r = read_state(c, &s);
    if (r != RRI_OK) return r;
    if (c->onchip_in_flight && s.halt == HALT_ERR_GRID_SYNC) {
        // The persistent launch gave up: its workgroups did not all run at the same time within the bound of the polls (a
        // device shared with another process, CUs masked away).  The reference's sweep cannot fail for scheduling reasons
        // (nmf.py:415-476), so neither may this one: W and T go back to what they were before the launch, and the same
        // range of steps runs on the launch-per-phase schedule, which needs no co-residency.  The handle stays there.
        c->onchip_in_flight = false;
        c->onchip_fallbacks += 1;
        c->onchip_off = true;
        c->onchip_off_until = steady_now_ns() + (2000000000LL << std::min<long>(c->onchip_fallbacks - 1, 5));
        long long backoff_ms = 2000;                                        // the whole process: 2 s off the persistent path
        if (const char* e = getenv("RRI_ONCHIP_BACKOFF_MS")) backoff_ms = std::max(0, atoi(e));      // tests: 0
        g_onchip_backoff_until.store(steady_now_ns() + backoff_ms * 1000000LL);
        if (getenv("RRI_ONCHIP_DEBUG")) fprintf(stderr, "rri: the persistent sweep gave up; sweeps %d.. rerun launch by launch\n", from.sweep);
        HIPCHK(c, hipMemcpyAsync(c->W, c->Wsafe, (size_t)c->k * c->ldw * 8, hipMemcpyDeviceToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(c->T, c->Tsafe, (size_t)c->k * c->LD * 8, hipMemcpyDeviceToDevice, c->stream));
        // A launch that gave up IN the run (not at its entry) has left more than W and T behind: objective slots of the sweeps it
        // finished (rri_sweep_until's history: "NaN = no value from the kernel" must hold for the sweeps rerun below) and
        // DevState.obj_track.  The history goes back to "not written"; obj_track is never read for this call (onchip_in_flight is
        // off, invalidate() drops obj_track_valid), XYpart is rewritten by the rerun's own W halves.
        if (c->objhist && c->until.active)
            HIPCHK(c, hipMemsetAsync(c->objhist, 0xFF, (size_t)std::min(std::max(c->until.n, 0), ONCHIP_UNTIL_CAP) * 8, c->stream));
        c->obj_track_pending = false;
        invalidate(c);
        c->pending_wcheck = false;
        c->skip_row_finish = c->onchip_saved_skip;
        r = clear_halt(c);
        if (r != RRI_OK) return r;
        if (c->until.active) c->run_total = std::min(c->run_total, from.sweep + 1);     // no stop rule on this schedule: one sweep, then the caller
        enqueue_range(c, from, c->run_total);
        enqueue_final_check(c, c->run_total);
        le = hipGetLastError();
        if (le != hipSuccess) return fail(c, RRI_ERR_HIP, "kernel launch failed: %s", hipGetErrorString(le));
        r = read_state(c, &s);
        if (r != RRI_OK) return r;
    }
    if (c->onchip_in_flight && c->obj_track_pending && s.halt == 0 && c->xy_valid) {
        c->obj_track_value = s.obj_track;       // came with the state read above: no kernel, no second trip for rri_objective
        c->obj_track_valid = true;
    }
    c->obj_track_pending = false;
    c->onchip_in_flight = false;
    return status_from_halt(c, s, sweeps_done);
}

rri_status rri_sweep(rri_ctx* c, int32_t n_sweeps, int32_t* sweeps_done) {
    CHECK_CTX(c);
    rri_status r = ready(c);
    if (r != RRI_OK) return r;
    if (c->paused) return fail(c, RRI_ERR_INVALID, "a paused run is pending: resolve the event and call rri_resume");
    if (n_sweeps < 0) return fail(c, RRI_ERR_INVALID, "n_sweeps < 0");
    c->until.active = false;
    c->run_total = n_sweeps;
    r = clear_halt(c);
    if (r != RRI_OK) return r;
    return run_and_collect(c, Cursor{0, 0, 0}, sweeps_done);
}

static rri_status ensure_x_sq(rri_ctx* c);
// the objective history of a finished rri_sweep_until call: what the kernel left (NaN where it left nothing)
static rri_status until_finish(rri_ctx* c, rri_status r, const int32_t* sweeps_done) {
    if (!c->until.active || r == RRI_PAUSED) return r;
    c->until.active = false;
    if (r != RRI_OK || !c->until.out) return r;
    const int done = sweeps_done ? std::min(std::max(*sweeps_done, 0), c->until.n) : 0;
    for (int i = 0; i < c->until.n; ++i) c->until.out[i] = std::nan("");
    if (done > 0 && c->objhist && c->onchip_launches > 0) {
        hipError_t e = hipMemcpyAsync(c->until.out, c->objhist, (size_t)done * 8, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) return fail(c, RRI_ERR_HIP, "objective history: %s", hipGetErrorString(e));
    }
    return r;
}

rri_status rri_sweep_until(rri_ctx* c, int32_t n_sweeps, double obj_prev, double stop_scale, double* obj_hist, int32_t* sweeps_done) {
    CHECK_CTX(c);
    rri_status r = ready(c);
    if (r != RRI_OK) return r;
    if (c->paused) return fail(c, RRI_ERR_INVALID, "a paused run is pending: resolve the event and call rri_resume");
    if (n_sweeps < 1 || !obj_hist || !sweeps_done) return fail(c, RRI_ERR_INVALID, "n_sweeps < 1, or no place for the history / the count");
    HIPCHK(c, hipSetDevice(c->device));
    if (!onchip_ok(c) || n_sweeps > ONCHIP_UNTIL_CAP || !g_onchip_obj)
        return fail(c, RRI_ERR_UNSUPPORTED, "rri_sweep_until needs the register-resident sweep (rri_onchip_info) and at most %d sweeps", ONCHIP_UNTIL_CAP);
    r = ensure_x_sq(c);
    if (r != RRI_OK) return r;
    // the slots of the history the kernel writes must read "nothing" if no persistent launch of this call wrote them
    if (c->objhist) HIPCHK(c, hipMemsetAsync(c->objhist, 0xFF, (size_t)n_sweeps * 8, c->stream));
    c->until.active = true; c->until.prev = obj_prev; c->until.scale = stop_scale; c->until.n = n_sweeps; c->until.out = obj_hist;
    c->run_total = n_sweeps;
    r = clear_halt(c);
    if (r != RRI_OK) { c->until.active = false; return r; }
    r = run_and_collect(c, Cursor{0, 0, 0}, sweeps_done);
    if (r != RRI_OK && r != RRI_PAUSED) c->until.active = false;
    return until_finish(c, r, sweeps_done);
}

rri_status rri_resume(rri_ctx* c, int32_t* sweeps_done) {
    CHECK_CTX(c);
    if (!c->paused) return fail(c, RRI_ERR_INVALID, "nothing to resume");
    if (c->pending.kind != RRI_EVENT_NONE) return fail(c, RRI_ERR_INVALID, "pending event not resolved");
    c->paused = false;
    rri_status r = clear_halt(c);
    if (r != RRI_OK) return r;
    int32_t done_local = 0;
    r = run_and_collect(c, c->resume_at, sweeps_done ? sweeps_done : &done_local);
    if (r != RRI_OK && r != RRI_PAUSED) c->until.active = false;
    return until_finish(c, r, sweeps_done ? sweeps_done : &done_local);
}

rri_status rri_pending_event(rri_ctx* c, rri_event* ev) {
    CHECK_CTX(c);
    if (!ev) return fail(c, RRI_ERR_INVALID, "ev is NULL");
    *ev = c->paused ? c->pending : rri_event{RRI_EVENT_NONE, -1, 0, 0};
    return RRI_OK;
}

static void event_resolved(rri_ctx* c) {
    c->q_valid = false; c->gfull_valid = false;   // a reset rewrites T[t,:] even when T is otherwise fixed (nmf.py:808,814)
    if (c->pending.kind == RRI_EVENT_RESET_T) c->skip_row_finish = true;
    c->pending.kind = RRI_EVENT_NONE;
    if (c->prm.resets_left > 0) c->prm.resets_left -= 1;
    invalidate(c);
}

rri_status rri_apply_reset_max_resid(rri_ctx* c, int32_t t, int64_t* row_chosen) {
    CHECK_CTX(c);
    if (t < 0 || t >= c->k) return fail(c, RRI_ERR_INVALID, "topic out of range");
    HIPCHK(c, hipSetDevice(c->device));
    if (!c->rowpos) HIPCHK(c, hipMalloc((void**)&c->rowpos, (size_t)c->n * sizeof(double)));
    HIPCHK(c, clear_halt(c) == RRI_OK ? hipSuccess : hipErrorUnknown);
    DISPATCH(c, L::resid(c, false, false, nullptr, c->rowpos));
    hipLaunchKernelGGL(k_vec_sum_argmax, dim3(1), dim3(1024), 0, c->stream, (const double*)c->rowpos, c->n,
                       (double*)nullptr, c->itmp);
    if (c->comm) {
        // row-sharded (nmf.py:771-776 over all rows): every rank offers its largest row-residual norm, the global
        // winner -- lowest global row index on ties, as np.argmax -- broadcasts max(X[mi,:] - W[mi,:] T, 0); T[t,:]
        // becomes that row on every rank, W[:,t] the unit vector of the winning row
        rri_comm* m = c->comm;
        hipLaunchKernelGGL(k_pack_candidate, dim3(1), dim3(1), 0, c->stream, (const double*)c->rowpos,
                           (const i64*)c->itmp, c->row_offset, c->ctail);
        comm_allgather(c, c->ctail, 2, c->cand);
        std::vector<double> cand((size_t)2 * m->world);
        // the peers go on to the broadcast below: a failure here must not just return (comm_abort)
        hipError_t he = hipMemcpyAsync(cand.data(), c->cand, cand.size() * 8, hipMemcpyDeviceToHost, c->stream);
        if (he == hipSuccess) he = hipStreamSynchronize(c->stream);
        if (he != hipSuccess) return comm_abort(c, "reading the reset candidates", he);
        int win = 0;
        for (int r = 1; r < m->world; ++r)
            if (cand[2 * r] > cand[2 * win] || (cand[2 * r] == cand[2 * win] && cand[2 * r + 1] < cand[2 * win + 1])) win = r;
        if (m->rank == win) DISPATCH(c, L::reset_row(c));          // xraw = the reset row (d doubles)
        comm_broadcast(c, c->xraw, c->LD, win);
        if (m->rank != win) {
            const i64 none = -1;
            HIPCHK(c, hipMemcpyAsync(c->itmp, &none, sizeof(i64), hipMemcpyHostToDevice, c->stream));
        }
        LK::reset_commit(c, t);
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (c->comm_status != RRI_OK) { const rri_status r = c->comm_status; c->comm_status = RRI_OK; return r; }
        if (row_chosen) *row_chosen = (int64_t)cand[2 * win + 1];
        if (c->paused && c->pending.kind != RRI_EVENT_NONE) event_resolved(c);
        else invalidate(c);
        return RRI_OK;
    }
    DISPATCH(c, L::reset_row(c));
    LK::reset_commit(c, t);
    i64 mi = -1;
    HIPCHK(c, hipMemcpyAsync(&mi, c->itmp, sizeof(i64), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (row_chosen) *row_chosen = mi;
    if (c->paused && c->pending.kind != RRI_EVENT_NONE) event_resolved(c);
    else invalidate(c);
    return RRI_OK;
}

rri_status rri_apply_reset_vectors(rri_ctx* c, int32_t t, const double* T_row, const double* W_col) {
    CHECK_CTX(c);
    if (t < 0 || t >= c->k) return fail(c, RRI_ERR_INVALID, "topic out of range");
    HIPCHK(c, hipSetDevice(c->device));
    if (!c->resetT) HIPCHK(c, hipMalloc((void**)&c->resetT, (size_t)c->d * sizeof(double)));
    if (!c->resetW) HIPCHK(c, hipMalloc((void**)&c->resetW, (size_t)c->n * sizeof(double)));
    if (T_row) HIPCHK(c, hipMemcpyAsync(c->resetT, T_row, (size_t)c->d * sizeof(double), hipMemcpyHostToDevice, c->stream));
    if (W_col) HIPCHK(c, hipMemcpyAsync(c->resetW, W_col, (size_t)c->n * sizeof(double), hipMemcpyHostToDevice, c->stream));
    LK::set_row_col(c, t, T_row ? c->resetT : nullptr, W_col ? c->resetW : nullptr);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->paused && c->pending.kind != RRI_EVENT_NONE) event_resolved(c);
    else invalidate(c);
    HIPCHK(c, clear_halt(c) == RRI_OK ? hipSuccess : hipErrorUnknown);
    return RRI_OK;
}

rri_status rri_skip_reset(rri_ctx* c) {
    CHECK_CTX(c);
    if (!c->paused) return fail(c, RRI_ERR_INVALID, "no pending event");
    c->pending.kind = RRI_EVENT_NONE;
    c->prm.resets_left = 0;
    rri_status r = clear_halt(c);
    if (r != RRI_OK) return r;
    return RRI_OK;
}

rri_status rri_update_T_row(rri_ctx* c, int32_t t) {
    CHECK_CTX(c);
    rri_status r = ready(c);
    if (r != RRI_OK) return r;
    if (c->weighted) return fail(c, RRI_ERR_UNSUPPORTED, "half steps are not exposed for the weighted flavour");
    if (t < 0 || t >= c->k) return fail(c, RRI_ERR_INVALID, "topic out of range");
    HIPCHK(c, hipSetDevice(c->device));
    c->run_total = 1;
    r = clear_halt(c);
    if (r != RRI_OK) return r;
    if (resid_sched(c)) enqueue_rT_half(c, 0, t, true);
    else enqueue_T_half(c, 0, t, true);
    DevState s;
    r = read_state(c, &s);
    if (r != RRI_OK) return r;
    return status_from_halt(c, s, nullptr);
}

rri_status rri_update_W_col(rri_ctx* c, int32_t t) {
    CHECK_CTX(c);
    rri_status r = ready(c);
    if (r != RRI_OK) return r;
    if (c->weighted) return fail(c, RRI_ERR_UNSUPPORTED, "half steps are not exposed for the weighted flavour");
    if (t < 0 || t >= c->k) return fail(c, RRI_ERR_INVALID, "topic out of range");
    HIPCHK(c, hipSetDevice(c->device));
    c->run_total = 1;
    r = clear_halt(c);
    if (r != RRI_OK) return r;
    c->skip_row_finish = true;   // a lone W half: the T-row checks belong to rri_update_T_row
    if (resid_sched(c)) enqueue_rW_half(c, 0, t);
    else enqueue_W_half(c, 0, t);
    if (c->pending_wcheck) {
        wcheck_now(c, c->pending_wcheck_topic, 1, 0);
        c->pending_wcheck = false;
    }
    DevState s;
    r = read_state(c, &s);
    if (r != RRI_OK) return r;
    return status_from_halt(c, s, nullptr);
}

// ---- around the loop ------------------------------------------------------------------------------------
rri_status rri_project_W_rows(rri_ctx* c, double s, const double* s_vec) {
    CHECK_CTX(c);
    if (!c->have_W) return fail(c, RRI_ERR_INVALID, "W not set");
    if (!s_vec && !(s > 0)) return fail(c, RRI_ERR_INVALID, "Radius s must be strictly positive");
    HIPCHK(c, hipSetDevice(c->device));
    DevTmp dv;
    double* dvec = nullptr;
    if (s_vec) {
        HIPCHK(c, dv.alloc((size_t)c->n * sizeof(double)));
        dvec = (double*)dv.p;
        HIPCHK(c, hipMemcpyAsync(dvec, s_vec, (size_t)c->n * sizeof(double), hipMemcpyHostToDevice, c->stream));
    }
    LK::proj_rows(c, s, dvec);
    hipError_t e = hipStreamSynchronize(c->stream);
    invalidate(c);
    if (e != hipSuccess) return fail(c, RRI_ERR_HIP, "projection failed: %s", hipGetErrorString(e));
    return RRI_OK;
}

static rri_status norms_of(rri_ctx* c, const double* A, i64 rows, i64 cols, i64 ld, double out[3]) {
    LK::norms(c, A, rows, cols, ld);
    double h[256 * 3];
    HIPCHK(c, hipMemcpyAsync(h, c->normpart, sizeof h, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    out[0] = out[1] = out[2] = 0.0;
    for (int b = 0; b < 256; ++b) { out[0] += h[3 * b]; out[1] += h[3 * b + 1]; out[2] += h[3 * b + 2]; }
    return RRI_OK;
}

// out = {data term, ||W||^2, ||W||_1}; tn (optional) = {., ||T||^2, ||T||_1}: everything rri_objective needs, with
// ONE synchronisation on the path taken after a complete sweep
// ||X||^2, once per X (the constant of the objective's Gram form)
static rri_status ensure_x_sq(rri_ctx* c) {
    if (c->x_sq_valid) return RRI_OK;
    DISPATCH(c, hipLaunchKernelGGL((k_sqsum<typename L::Elem>), dim3(256), dim3(256), 0, c->stream,
                                   (const typename L::Elem*)c->X, c->ldx, c->n, c->d, c->normpart));
    double h[256];
    HIPCHK(c, hipMemcpyAsync(h, c->normpart, sizeof h, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->x_sq = 0.0;
    for (int b = 0; b < 256; ++b) c->x_sq += h[b];
    c->x_sq_valid = true;
    return RRI_OK;
}

static rri_status objective_terms(rri_ctx* c, double out[3], double* tn) {
    CHECK_CTX(c);
    if (!c->have_X || !c->have_W || !c->have_T) return fail(c, RRI_ERR_INVALID, "X, W, T must be set");
    if (c->weighted && !c->have_M) return fail(c, RRI_ERR_INVALID, "weighted handle without a mask");
    HIPCHK(c, hipSetDevice(c->device));
    if (!c->rowobj) HIPCHK(c, hipMalloc((void**)&c->rowobj, (size_t)c->n * sizeof(double)));
    if (c->weighted) {
        // the objective needs M .* (X - W T) -- which is the maintained residual: store it while it is being
        // computed, and the sweep that follows skips its own rebuild (nmf() asks for the objective after
        // every sweep, nmf.py:488-490)
        DISPATCH(c, L::resid(c, true, true, c->rowobj, nullptr));
        c->resid_valid = true;
        c->resid_fresh = true;
        c->dt_pending = false;
        c->dw_pending = false;
        c->carry_valid = false;
    } else if (c->xy_valid && !g_obj_direct) {
        // 1/2 ||X - W T||^2 = 1/2 ||X||^2 - sum_t <w_t, X t_t> + 1/2 <W^T W, T T^T>: the cross terms were left by
        // the W halves of the sweep that has just ended (k_wcol), ||X||^2 is taken once per X -- no pass over X.
        // The terms are of the size of ||X||^2: the result carries an absolute error of a few ulp of that
        // (relative 1e-11 at a residual of 0.5 %), far below what the stop rule of nmf.py:510 resolves.
        const int k = c->k;
        rri_status rx = ensure_x_sq(c);
        if (rx != RRI_OK) return rx;
        double* gw = c->objbuf;
        double* gt = gw + k * k;
        double* xy = gt + k * k;
        hipLaunchKernelGGL(k_gram, dim3(k, k), dim3(256), 0, c->stream, (const double*)c->W, c->ldw, c->n, k, gw);
        hipLaunchKernelGGL(k_gram, dim3(k, k), dim3(256), 0, c->stream, (const double*)c->T, c->LD, c->d, k, gt);
        hipLaunchKernelGGL(k_rows_sum, dim3(k), dim3(256), 0, c->stream, (const double*)c->XYpart, c->xy_rows, c->xy_stride, xy);
        // ||.||^2 are the traces of the Gram matrices; the 1-norms are only needed with an l1 penalty
        const bool need_l1 = c->prm.reg_w_l1 != 0.0 || c->prm.reg_t_l1 != 0.0 || !tn;
        double hw[256 * 3], ht[256 * 3];
        if (need_l1) {
            LK::norms(c, c->W, c->k, c->n, c->ldw);
            HIPCHK(c, hipMemcpyAsync(hw, c->normpart, sizeof hw, hipMemcpyDeviceToHost, c->stream));
            if (tn) {
                LK::norms(c, c->T, c->k, c->d, c->LD);
                HIPCHK(c, hipMemcpyAsync(ht, c->normpart, sizeof ht, hipMemcpyDeviceToHost, c->stream));
            }
        }
        std::vector<double> h((size_t)(2 * k * k + k));
        HIPCHK(c, hipMemcpyAsync(h.data(), gw, h.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        double cross = 0.0, quad = 0.0, w2 = 0.0, t2 = 0.0;
        for (int t = 0; t < k; ++t) cross += h[(size_t)2 * k * k + t];
        for (int a = 0; a < k * k; ++a) quad += h[(size_t)a] * h[(size_t)k * k + a];
        for (int t = 0; t < k; ++t) { w2 += h[(size_t)t * k + t]; t2 += h[(size_t)k * k + (size_t)t * k + t]; }
        double w1 = 0.0, t1 = 0.0;
        if (need_l1) {
            for (int b = 0; b < 256; ++b) w1 += hw[3 * b + 2];
            if (tn) for (int b = 0; b < 256; ++b) t1 += ht[3 * b + 2];
        }
        out[0] = 0.5 * c->x_sq - cross + 0.5 * quad;
        out[1] = w2;
        out[2] = w1;
        if (tn) { tn[0] = 0.0; tn[1] = t2; tn[2] = t1; }
        return RRI_OK;
    } else if (c->explicit_resid) {
        // the objective is 1/2 ||R||^2 of the residual this schedule keeps: store it while it is computed, and the
        // sweep that follows skips its own rebuild
        DISPATCH(c, L::resid(c, false, true, c->rowobj, nullptr));
        c->resid_valid = true;
        c->resid_fresh = true;
        c->dw_pending = false;
        c->dt_pending = false;
    } else {
        DISPATCH(c, L::resid(c, false, false, c->rowobj, nullptr));
    }
    hipLaunchKernelGGL(k_vec_sum_argmax, dim3(1), dim3(1024), 0, c->stream, (const double*)c->rowobj, c->n,
                       c->dtmp, (i64*)nullptr);
    double base = 0.0;
    HIPCHK(c, hipMemcpyAsync(&base, c->dtmp, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    double nw[3];
    rri_status r = norms_of(c, c->W, c->k, c->n, c->ldw, nw);
    if (r != RRI_OK) return r;
    out[0] = 0.5 * base;
    out[1] = nw[1];
    out[2] = nw[2];
    if (tn) return norms_of(c, c->T, c->k, c->d, c->LD, tn);
    return RRI_OK;
}

rri_status rri_objective_parts(rri_ctx* c, double out[3]) { return objective_terms(c, out, nullptr); }

rri_status rri_objective(rri_ctx* c, double* out) {
    CHECK_CTX(c);
    if (!out) return fail(c, RRI_ERR_INVALID, "out is NULL");
    if (c->obj_track_valid && c->xy_valid && !c->weighted && !c->comm && !g_obj_direct) {
        // the persistent sweep that has just ended left its objective (all terms but the constant): nothing to launch
        if (!c->have_X || !c->have_W || !c->have_T) return fail(c, RRI_ERR_INVALID, "X, W, T must be set");
        HIPCHK(c, hipSetDevice(c->device));
        rri_status rx = ensure_x_sq(c);
        if (rx != RRI_OK) return rx;
        *out = 0.5 * c->x_sq + c->obj_track_value;
        return RRI_OK;
    }
    double parts[3], nt[3];
    rri_status r = objective_terms(c, parts, nt);
    if (r != RRI_OK) return r;
    r = comm_allreduce_host(c, parts, 3);    // row-sharded: the terms over rows are sums over the ranks; T is replicated
    if (r != RRI_OK) return r;
    const rri_params& q = c->prm;
    // base + wr2 + tr2 + tr1 + wr1 (nmf.py:83-91)
    *out = parts[0] + 0.5 * q.reg_w_l2 * parts[1] + 0.5 * q.reg_t_l2 * nt[1] + q.reg_t_l1 * nt[2] +
           q.reg_w_l1 * parts[2];
    return RRI_OK;
}

rri_status rri_argmax_rows(rri_ctx* c, int32_t* out_host) {
    CHECK_CTX(c);
    if (!out_host) return fail(c, RRI_ERR_INVALID, "out is NULL");
    HIPCHK(c, hipSetDevice(c->device));
    DevTmp dv;
    HIPCHK(c, dv.alloc((size_t)c->n * sizeof(int)));
    int* dev = (int*)dv.p;
    LK::argmax_rows(c, dev);
    hipError_t e = hipMemcpyAsync(out_host, dev, (size_t)c->n * sizeof(int), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) return fail(c, RRI_ERR_HIP, "argmax failed: %s", hipGetErrorString(e));
    return RRI_OK;
}

rri_status rri_masked_rmse(rri_ctx* c, const int64_t* ij, const double* vals, int64_t count, double lo, double hi,
                           double* out) {
    CHECK_CTX(c);
    // row-sharded (a communicator attached): collective; (i, j) are LOCAL rows, a rank may hold no entry at all, and the
    // score is sqrt(sum over the ranks of the squared errors / sum of the counts): the same number on every rank, so
    // the early-stop decision of nmf.py:381-407 is the same on every rank
    // With a communicator this call is a collective: a rank that found its own arguments bad must not leave before the
    // all-reduce its peers are about to enter (they would block inside it).  It contributes nothing, raises an error flag that
    // travels with the sums, and EVERY rank returns the error afterwards -- the ranks stay in step.
    const char* bad = nullptr;
    i64 bad_entry = -1;
    if (!out || count < 0 || (count > 0 && (!ij || !vals)) || (count < 1 && !c->comm)) bad = "bad entry list";
    if (!bad)
        for (i64 e = 0; e < count && !bad; ++e)
            if (ij[2 * e] < 0 || ij[2 * e] >= c->n || ij[2 * e + 1] < 0 || ij[2 * e + 1] >= c->d) { bad = "entry out of range"; bad_entry = e; }
    if (bad && !c->comm) return bad_entry >= 0 ? fail(c, RRI_ERR_INVALID, "entry %lld out of range", bad_entry) : fail(c, RRI_ERR_INVALID, "%s", bad);
    HIPCHK(c, hipSetDevice(c->device));
    double s = 0.0;
    hipError_t e = hipSuccess;
    if (count > 0 && !bad) {
        i64* dij = nullptr;
        double* dv = nullptr;
        e = hipMalloc((void**)&dij, (size_t)count * 2 * sizeof(i64));
        if (e == hipSuccess) e = hipMalloc((void**)&dv, (size_t)count * sizeof(double));
        if (e == hipSuccess) e = hipMemcpyAsync(dij, ij, (size_t)count * 2 * sizeof(i64), hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(dv, vals, (size_t)count * sizeof(double), hipMemcpyHostToDevice, c->stream);
        double h[256];
        if (e == hipSuccess) {
            LK::masked_sqerr(c, dij, dv, count, lo, hi);
            e = hipMemcpyAsync(h, c->normpart, sizeof h, hipMemcpyDeviceToHost, c->stream);
        }
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (dij) (void)hipFree(dij);
        if (dv) (void)hipFree(dv);
        if (e == hipSuccess)
            for (int b = 0; b < 256; ++b) s += h[b];
    }
    // [squared errors, count, ranks whose arguments were bad, ranks whose device work failed]
    double tot[4] = {bad ? 0.0 : s, bad ? 0.0 : (double)count, bad ? 1.0 : 0.0, e != hipSuccess ? 1.0 : 0.0};
    if (c->comm) {
        // a rank whose validation or device work failed still takes part in the collective (its peers would wait for it
        // otherwise); the failure is reported on every rank afterwards
        const rri_status r = comm_allreduce_host(c, tot, 4);
        if (r != RRI_OK) return r;
    }
    if (bad) return bad_entry >= 0 ? fail(c, RRI_ERR_INVALID, "entry %lld out of range", bad_entry) : fail(c, RRI_ERR_INVALID, "%s", bad);
    if (e != hipSuccess) return fail(c, RRI_ERR_HIP, "masked rmse failed: %s", hipGetErrorString(e));
    if (tot[2] > 0.0) return fail(c, RRI_ERR_INVALID, "bad entry list on another rank (%d of them)", (int)tot[2]);
    if (tot[3] > 0.0) return fail(c, RRI_ERR_HIP, "masked rmse failed on another rank");
    if (!(tot[1] > 0.0)) return fail(c, RRI_ERR_INVALID, "no entry on any rank");
    *out = std::sqrt(tot[0] / tot[1]);
    return RRI_OK;
}

rri_status rri_snapshot(rri_ctx* c) {
    CHECK_CTX(c);
    HIPCHK(c, hipSetDevice(c->device));
    if (!c->Wprev) HIPCHK(c, hipMalloc((void**)&c->Wprev, (size_t)c->k * c->ldw * 8));
    if (!c->Tprev) HIPCHK(c, hipMalloc((void**)&c->Tprev, (size_t)c->k * c->LD * 8));
    HIPCHK(c, hipMemcpyAsync(c->Wprev, c->W, (size_t)c->k * c->ldw * 8, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->Tprev, c->T, (size_t)c->k * c->LD * 8, hipMemcpyDeviceToDevice, c->stream));
    return RRI_OK;
}

rri_status rri_rollback(rri_ctx* c) {
    CHECK_CTX(c);
    if (!c->Wprev || !c->Tprev) return fail(c, RRI_ERR_INVALID, "no snapshot taken");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(c->W, c->Wprev, (size_t)c->k * c->ldw * 8, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->T, c->Tprev, (size_t)c->k * c->LD * 8, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    invalidate(c);
    c->q_valid = false; c->gfull_valid = false;
    c->pending_wcheck = false;
    return RRI_OK;
}

// ---- the explicit residual (RRI_UNWEIGHTED_RESIDUAL handles) --------------------------------------------------
rri_status rri_residual_rebuild(rri_ctx* c) {
    CHECK_CTX(c);
    if (!c->explicit_resid) return fail(c, RRI_ERR_INVALID, "handle was not created with RRI_UNWEIGHTED_RESIDUAL");
    if (!c->have_X || !c->have_W || !c->have_T) return fail(c, RRI_ERR_INVALID, "X, W, T must be set");
    HIPCHK(c, hipSetDevice(c->device));
    r_refresh(c);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return RRI_OK;
}

rri_status rri_get_residual(rri_ctx* c, void* host, int64_t ld, int32_t host_dtype) {
    CHECK_CTX(c);
    if (!c->explicit_resid) return fail(c, RRI_ERR_INVALID, "handle was not created with RRI_UNWEIGHTED_RESIDUAL");
    HIPCHK(c, hipSetDevice(c->device));
    return to_host(c, c->E, c->LD, host, ld, host_dtype, c->n, c->d, c->dtype);
}

rri_status rri_residual_update(rri_ctx* c, const double* a, const double* b, const double* a2, const double* b2,
                               const double* trow, const double* wcol, double* y_out, double* z_out) {
    CHECK_CTX(c);
    if (!c->explicit_resid) return fail(c, RRI_ERR_INVALID, "handle was not created with RRI_UNWEIGHTED_RESIDUAL");
    if (!a || !b || !trow || !wcol || ((a2 == nullptr) != (b2 == nullptr)))
        return fail(c, RRI_ERR_INVALID, "a, b, trow, wcol are required; a2 and b2 come together");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, clear_halt(c) == RRI_OK ? hipSuccess : hipErrorUnknown);
    // row vectors (n) and column vectors (padded to LD with zeros) on the device
    DevTmp rows, cols;
    HIPCHK(c, rows.alloc((size_t)3 * c->n * sizeof(double)));
    HIPCHK(c, cols.alloc((size_t)3 * c->LD * sizeof(double)));
    double* dr = (double*)rows.p;
    double* dc = (double*)cols.p;
    HIPCHK(c, hipMemsetAsync(dc, 0, (size_t)3 * c->LD * sizeof(double), c->stream));
    const double* hr[3] = {a, a2, wcol};
    const double* hc[3] = {b, b2, trow};
    for (int q = 0; q < 3; ++q) {
        if (hr[q]) HIPCHK(c, hipMemcpyAsync(dr + (i64)q * c->n, hr[q], (size_t)c->n * sizeof(double), hipMemcpyHostToDevice, c->stream));
        if (hc[q]) HIPCHK(c, hipMemcpyAsync(dc + (i64)q * c->LD, hc[q], (size_t)c->d * sizeof(double), hipMemcpyHostToDevice, c->stream));
    }
    DISPATCH(c, {
        typename L::Upd u;
        u.a = dr;
        u.b = dc;
        if (a2) { u.a2 = dr + c->n; u.b2 = dc + c->LD; u.b2sub = c->zeros; }
        L::rank_update(c, c->E, c->LD, u, dc + 2 * c->LD, dr + 2 * c->n);
    });
    // z: the row-block partials in a fixed order (k_reduce); y: the column-panel partials, added here in panel order
    const int nb = (int)((c->LD + 31) / 32);
    hipLaunchKernelGGL(k_reduce, dim3(nb), dim3(1024), 0, c->stream, (const double*)c->Zpart, c->LD, c->nrb,
                       (const double*)nullptr, 0, c->k, c->red, (const DevState*)c->st);
    if (z_out) HIPCHK(c, hipMemcpyAsync(z_out, c->red, (size_t)c->d * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    std::vector<double> yp;
    if (y_out) {
        yp.resize((size_t)c->npanels * c->n);
        HIPCHK(c, hipMemcpyAsync(yp.data(), c->Ypart, yp.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (y_out)
        for (i64 i = 0; i < c->n; ++i) {
            double sacc = 0.0;
            for (int pgi = 0; pgi < c->npanels; ++pgi) sacc += yp[(size_t)pgi * c->n + i];
            y_out[i] = sacc;
        }
    invalidate(c);            // Zpart / red were used, and R no longer equals X - W T for the handle's factors
    c->pending_wcheck = false;
    return RRI_OK;
}

// ---- products with X for the initialisation -------------------------------------------------------------------
// out (nseg x m) = X B or X^T B on a pattern-only handle; B: gdim x m host row-major
static rri_status sparse_times(rri_ctx* c, int which, const double* B, int32_t m, double* out) {
    if (!c->have_X) return fail(c, RRI_ERR_INVALID, "X not set");
    if (!B || !out || m < 1 || m > 64) return fail(c, RRI_ERR_INVALID, "bad operand (1 <= m <= 64 columns on a pattern-only handle)");
    HIPCHK(c, hipSetDevice(c->device));
    const rri_ctx::SpCopy& cp = c->sp[which];
    // B is padded with zero rows up to nblk * bw so that a block's offsets always land inside it
    const i64 brows = (i64)cp.nblk * cp.bw;
    DevTmp bd, part, res;
    HIPCHK(c, bd.alloc((size_t)brows * m * sizeof(double)));
    HIPCHK(c, part.alloc((size_t)cp.nblk * cp.nseg * m * sizeof(double)));
    HIPCHK(c, res.alloc((size_t)cp.nseg * m * sizeof(double)));
    HIPCHK(c, hipMemsetAsync(bd.p, 0, (size_t)brows * m * sizeof(double), c->stream));
    HIPCHK(c, hipMemcpyAsync(bd.p, B, (size_t)cp.gdim * m * sizeof(double), hipMemcpyHostToDevice, c->stream));
    DISPATCH(c, L::sp_spmm(c, which, (const double*)bd.p, m, (double*)part.p, (double*)res.p));
    HIPCHK(c, hipMemcpyAsync(out, res.p, (size_t)cp.nseg * m * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return RRI_OK;
}

rri_status rri_X_times(rri_ctx* c, const double* B, int32_t m, double* out) {
    CHECK_CTX(c);
    if (c->sparse) return sparse_times(c, 0, B, m, out);
    if (!c->have_X) return fail(c, RRI_ERR_INVALID, "X not set");
    if (!B || !out || m < 1) return fail(c, RRI_ERR_INVALID, "bad operand");
    HIPCHK(c, hipSetDevice(c->device));
    DevTmp tm, outm;
    HIPCHK(c, tm.alloc((size_t)m * c->LD * sizeof(double)));      // B^T as an m x LD "T-like" operand
    HIPCHK(c, outm.alloc((size_t)m * c->ldw * sizeof(double)));   // (X B)^T, m x n
    HIPCHK(c, hipMemsetAsync(tm.p, 0, (size_t)m * c->LD * sizeof(double), c->stream));
    rri_status s = to_device(c, B, m, RRI_F64, tm.p, c->LD, c->d, m, RRI_F64, true);
    if (s != RRI_OK) return s;
    HIPCHK(c, clear_halt(c) == RRI_OK ? hipSuccess : hipErrorUnknown);
    DISPATCH(c, L::xtt_any(c, (const double*)tm.p, m, (double*)outm.p));
    return to_host(c, outm.p, c->ldw, out, m, RRI_F64, c->n, m, RRI_F64, true);
}

rri_status rri_Xt_times(rri_ctx* c, const double* Q, int32_t m, double* out) {
    CHECK_CTX(c);
    if (c->sparse) return sparse_times(c, 1, Q, m, out);
    if (!c->have_X) return fail(c, RRI_ERR_INVALID, "X not set");
    if (!Q || !out || m < 1) return fail(c, RRI_ERR_INVALID, "bad operand");
    HIPCHK(c, hipSetDevice(c->device));
    DevTmp qm, outm;
    HIPCHK(c, qm.alloc((size_t)m * c->ldw * sizeof(double)));     // Q^T, m x n: every column contiguous
    HIPCHK(c, outm.alloc((size_t)m * c->LD * sizeof(double)));    // (X^T Q)^T, m x LD
    rri_status s = to_device(c, Q, m, RRI_F64, qm.p, c->ldw, c->n, m, RRI_F64, true);
    if (s != RRI_OK) return s;
    HIPCHK(c, clear_halt(c) == RRI_OK ? hipSuccess : hipErrorUnknown);
    const bool tm_on = c->timing != 0;
    const int tsave = c->timing;
    c->timing = 0;
    DevTmp zm;      // partial column sums of 8 vectors at a time: X is read once per 8 vectors
    HIPCHK(c, zm.alloc((size_t)8 * c->nrb * c->LD * sizeof(double)));
    for (int l = 0; l < m; l += 8)
        DISPATCH(c, L::colsums8(c, (const double*)qm.p + (i64)l * c->ldw, std::min(8, m - l), (double*)zm.p,
                                (double*)outm.p + (i64)l * c->LD));
    c->timing = tsave;
    (void)tm_on;
    invalidate(c);   // Zpart / red were used as scratch
    return to_host(c, outm.p, c->LD, out, m, RRI_F64, c->d, m, RRI_F64, true);
}

// ---- the range finder of the randomized SVD, on the device ---------------------------------------------------------------
// sklearn.utils.extmath.randomized_svd (what initialization.py:105 calls) is: Q <- normalise(A Q), Q <- normalise(A^T Q), n_iter
// times; Q <- qr(A Q); B = Q^T A; small SVD of B.  All of its work is the products with X (rri_X_times / rri_Xt_times), and
// between them the (n or d) x m panels were normalised by LAPACK on the host: 0.7 of the 1.1 s of the start at 100000 x 10000
// (DESIGN 8).  Here the panels never leave the device: every normalisation is Cholesky-QR (G = Y^T Y by k_gram, its m x m
// Cholesky factor on the host -- 60 x 60 --, Y <- Y L^-T by k_lsolve_rows; shifted in its first of three passes) -- another basis of the same range than LU / QR give,
// so U, S, V of the SVD that follows are scikit-learn's up to rounding (the row-sharded start has done the same since round 2).
namespace {
// Lower Cholesky factor of the symmetric m x m G + shift_rel trace(G) I (row-major) in place.  A pivot that falls below 1e-14 of
// its diagonal entry -- a panel that is rank-deficient to working precision: a direction the rounding of G has already lost -- is
// held at that floor, so the factor stays finite and the direction comes out as normalised noise, as a QR would leave it.
bool host_cholesky(std::vector<double>& G, int m, double shift_rel) {
    double tr = 0.0;
    for (int i = 0; i < m; ++i) tr += G[(size_t)i * m + i];
    std::vector<double> d0((size_t)m);
    for (int i = 0; i < m; ++i) { d0[(size_t)i] = G[(size_t)i * m + i] + shift_rel * tr; G[(size_t)i * m + i] = d0[(size_t)i]; }
    for (int j = 0; j < m; ++j) {
        double dj = G[(size_t)j * m + j];
        for (int q = 0; q < j; ++q) dj -= G[(size_t)j * m + q] * G[(size_t)j * m + q];
        const double floor_j = d0[(size_t)j] > 0.0 ? 1e-14 * d0[(size_t)j] : 1e-300;
        if (!(dj > floor_j)) dj = floor_j;
        if (!std::isfinite(dj)) return false;
        const double ljj = std::sqrt(dj);
        G[(size_t)j * m + j] = ljj;
        for (int i = j + 1; i < m; ++i) {
            double v = G[(size_t)i * m + j];
            for (int q = 0; q < j; ++q) v -= G[(size_t)i * m + q] * G[(size_t)j * m + q];
            G[(size_t)i * m + j] = v / ljj;
            if (!std::isfinite(G[(size_t)i * m + j])) return false;
        }
        for (int i = 0; i < j; ++i) G[(size_t)i * m + j] = 0.0;
    }
    return true;
}
// The rows of At (m x len, stride ld, device) made orthonormal by shifted Cholesky-QR, three passes (Fukaya et al.: the first
// pass factorises G + s I, s = 1e-9 trace(G), which a panel of any condition number survives and which leaves it conditioned
// like 1e4; the next two are plain Cholesky-QR2).  A Gaussian test matrix times a matrix with one dominant direction -- rows
// normalised to sum 1, every document close to the mean -- is conditioned like 1e8 and worse: plain Cholesky-QR2 broke there.
rri_status cholqr2_rows(rri_ctx* c, double* At, i64 ld, i64 len, int m, double* Gdev) {
    std::vector<double> G((size_t)m * m);
    for (int round = 0; round < 3; ++round) {
        hipLaunchKernelGGL(k_gram, dim3(m, m), dim3(256), 0, c->stream, (const double*)At, ld, len, m, Gdev);
        HIPCHK(c, hipMemcpyAsync(G.data(), Gdev, G.size() * 8, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (!host_cholesky(G, m, round == 0 ? 1e-9 : 0.0)) return fail(c, RRI_ERR_INVALID, "range finder: the panel holds a non-finite value");
        HIPCHK(c, hipMemcpyAsync(Gdev, G.data(), G.size() * 8, hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(k_lsolve_rows, dim3((unsigned)((len + 255) / 256)), dim3(256), 0, c->stream, At, ld, len, m, (const double*)Gdev);
    }
    return RRI_OK;
}
}  // namespace

rri_status rri_range_finder(rri_ctx* c, const double* Q0, int32_t m, int32_t n_iter, int32_t transpose, double* Q_out,
                            double* B_out) {
    CHECK_CTX(c);
    if (c->sparse) return fail(c, RRI_ERR_UNSUPPORTED, "a pattern-only handle takes the products one by one (rri_X_times / rri_Xt_times)");
    if (!c->have_X) return fail(c, RRI_ERR_INVALID, "X not set");
    if (!Q0 || !Q_out || !B_out || m < 1 || m > 64 || n_iter < 0) return fail(c, RRI_ERR_INVALID, "bad operand (1 <= m <= 64)");
    HIPCHK(c, hipSetDevice(c->device));
    // panels, transposed as the product kernels take them: Pd (m x LD) lives on the column side of X, Pn (m x ldw) on the row side
    DevTmp pd, pn, zm, gd;
    HIPCHK(c, pd.alloc((size_t)m * c->LD * sizeof(double)));
    HIPCHK(c, pn.alloc((size_t)m * c->ldw * sizeof(double)));
    HIPCHK(c, zm.alloc((size_t)8 * c->nrb * c->LD * sizeof(double)));
    HIPCHK(c, gd.alloc((size_t)64 * 64 * sizeof(double)));
    double *Pd = (double*)pd.p, *Pn = (double*)pn.p;
    HIPCHK(c, hipMemsetAsync(Pd, 0, (size_t)m * c->LD * sizeof(double), c->stream));
    HIPCHK(c, clear_halt(c) == RRI_OK ? hipSuccess : hipErrorUnknown);
    const int tsave = c->timing;
    c->timing = 0;
    auto X_times_dev = [&]() { DISPATCH(c, L::xtt_any(c, (const double*)Pd, m, Pn)); };                 // Pn = (X Pd^T)^T
    auto Xt_times_dev = [&]() {                                                                          // Pd = (X^T Pn^T)^T
        for (int l = 0; l < m; l += 8)
            DISPATCH(c, L::colsums8(c, (const double*)Pn + (i64)l * c->ldw, std::min(8, m - l), (double*)zm.p, Pd + (i64)l * c->LD));
    };
    rri_status s;
    // A = X (transpose == 0: Q0 is d x m) or A = X^T (Q0 is n x m), as scikit-learn transposes when n < d
    if (!transpose) s = to_device(c, Q0, m, RRI_F64, Pd, c->LD, c->d, m, RRI_F64, true);
    else s = to_device(c, Q0, m, RRI_F64, Pn, c->ldw, c->n, m, RRI_F64, true);
    for (int it = 0; it < n_iter && s == RRI_OK; ++it) {
        if (!transpose) {
            X_times_dev();  s = cholqr2_rows(c, Pn, c->ldw, c->n, m, (double*)gd.p);
            if (s == RRI_OK) { Xt_times_dev(); s = cholqr2_rows(c, Pd, c->LD, c->d, m, (double*)gd.p); }
        } else {
            Xt_times_dev(); s = cholqr2_rows(c, Pd, c->LD, c->d, m, (double*)gd.p);
            if (s == RRI_OK) { X_times_dev(); s = cholqr2_rows(c, Pn, c->ldw, c->n, m, (double*)gd.p); }
        }
    }
    if (s == RRI_OK) {
        if (!transpose) {       // Q = orth(X Q) (n x m), B = Q^T X (m x d)
            X_times_dev();  s = cholqr2_rows(c, Pn, c->ldw, c->n, m, (double*)gd.p);
            if (s == RRI_OK) Xt_times_dev();
        } else {                // Q = orth(X^T Q) (d x m), B = Q^T X^T (m x n)
            Xt_times_dev(); s = cholqr2_rows(c, Pd, c->LD, c->d, m, (double*)gd.p);
            if (s == RRI_OK) X_times_dev();
        }
    }
    c->timing = tsave;
    invalidate(c);   // Zpart / red were used as scratch
    if (s != RRI_OK) return s;
    if (!transpose) {
        s = to_host(c, Pn, c->ldw, Q_out, m, RRI_F64, c->n, m, RRI_F64, true);
        if (s == RRI_OK) s = to_host(c, Pd, c->LD, B_out, c->d, RRI_F64, m, c->d, RRI_F64, false);
    } else {
        s = to_host(c, Pd, c->LD, Q_out, m, RRI_F64, c->d, m, RRI_F64, true);
        if (s == RRI_OK) s = to_host(c, Pn, c->ldw, B_out, c->n, RRI_F64, m, c->n, RRI_F64, false);
    }
    return s;
}

// ---- preprocessing of the resident X ---------------------------------------------------------------------------
rri_status rri_column_positive_counts(rri_ctx* c, double* df_out) {
    CHECK_CTX(c);
    if (!df_out) return fail(c, RRI_ERR_INVALID, "df_out is NULL");
    if (c->weighted || !c->have_X) return fail(c, RRI_ERR_INVALID, "needs an unweighted handle with a dense X");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, clear_halt(c) == RRI_OK ? hipSuccess : hipErrorUnknown);
    const int ncols = (int)std::min<i64>(c->ldx, c->LD);
    HIPCHK(c, hipMemsetAsync(c->Zpart, 0, (size_t)c->nrb * c->LD * sizeof(double), c->stream));
    DISPATCH(c, hipLaunchKernelGGL((k_col_count<typename L::Elem>), dim3(c->npanels * c->nrb), dim3(256), 0, c->stream,
                                   (const typename L::Elem*)c->X, c->ldx, (int)c->n, ncols, c->Zpart, c->LD, c->rpb,
                                   c->npanels));
    LK::reduce(c);
    HIPCHK(c, hipMemcpyAsync(df_out, c->red, (size_t)c->d * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    invalidate(c);   // Zpart / red were used as scratch
    return RRI_OK;
}

rri_status rri_scale_X(rri_ctx* c, const double* col_scale, int32_t normalize_rows) {
    CHECK_CTX(c);
    if (c->weighted || !c->have_X) return fail(c, RRI_ERR_INVALID, "needs an unweighted handle with a dense X");
    if (!c->own_X) return fail(c, RRI_ERR_INVALID, "X is bound caller memory: it is not rewritten in place");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, clear_halt(c) == RRI_OK ? hipSuccess : hipErrorUnknown);
    DevTmp sd, inv;
    double* sdev = nullptr;
    HIPCHK(c, sd.alloc((size_t)c->LD * sizeof(double)));
    sdev = (double*)sd.p;
    if (col_scale) {
        HIPCHK(c, hipMemsetAsync(sdev, 0, (size_t)c->LD * sizeof(double), c->stream));
        HIPCHK(c, hipMemcpyAsync(sdev, col_scale, (size_t)c->d * sizeof(double), hipMemcpyHostToDevice, c->stream));
    } else {
        std::vector<double> ones((size_t)c->LD, 0.0);
        std::fill(ones.begin(), ones.begin() + c->d, 1.0);
        HIPCHK(c, hipMemcpyAsync(sdev, ones.data(), (size_t)c->LD * sizeof(double), hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    double* invdev = nullptr;
    if (normalize_rows) {
        HIPCHK(c, inv.alloc((size_t)c->n * sizeof(double)));
        invdev = (double*)inv.p;
        // row sums of X * col_scale = the row dots of the streaming pass against col_scale
        const int tsave = c->timing;
        c->timing = 0;
        DISPATCH(c, (L::template pass_cfg<true, false, 0>(c, c->X, c->ldx, sdev, c->W)));
        c->timing = tsave;
        hipLaunchKernelGGL(k_row_inverse, dim3((unsigned)((c->n + 255) / 256)), dim3(256), 0, c->stream,
                           (const double*)c->Ypart, c->npanels, (int)c->n, invdev);
    }
    DISPATCH(c, hipLaunchKernelGGL((k_scale2d<typename L::Elem>), dim3(8192), dim3(256), 0, c->stream,
                                   (typename L::Elem*)c->X, c->ldx, c->n, (int)c->d, col_scale ? (const double*)sdev : nullptr,
                                   (const double*)invdev));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    invalidate(c);
    c->q_valid = false; c->gfull_valid = false;
    c->x_sq_valid = false;
    return RRI_OK;
}

// ---- row-sharded multi-GPU ---------------------------------------------------------------------------------
rri_status rri_reduce_buffer(rri_ctx* c, void** dev_ptr, int64_t* n_elems) {
    CHECK_CTX(c);
    if (dev_ptr) *dev_ptr = (void*)c->red;
    if (n_elems) *n_elems = c->red_elems;
    return RRI_OK;
}

rri_status rri_bind_reduce_buffer(rri_ctx* c, void* dev_ptr, int64_t n_elems) {
    CHECK_CTX(c);
    if (!dev_ptr || n_elems < c->red_elems || ((uintptr_t)dev_ptr) % 16)
        return fail(c, RRI_ERR_INVALID, "reduce buffer needs >= %lld elements, 16-byte aligned", c->red_elems);
    if (c->own_red && c->red) (void)hipFree(c->red);
    c->red = (double*)dev_ptr;
    c->own_red = false;
    invalidate(c);
    return RRI_OK;
}

rri_status rri_topic_reduce_local(rri_ctx* c, int32_t t) {
    CHECK_CTX(c);
    rri_status r = ready(c);
    if (r != RRI_OK) return r;
    if (c->prm.fix_T) return fail(c, RRI_ERR_UNSUPPORTED, "split stepping is the T-row step taken apart: T must be free");
    if (c->explicit_resid) return fail(c, RRI_ERR_UNSUPPORTED, "row-sharded stepping runs the Gram-form schedule only");
    if (c->comm) return fail(c, RRI_ERR_INVALID, "a communicator is attached: rri_sweep does the collectives itself");
    if (t < 0 || t >= c->k) return fail(c, RRI_ERR_INVALID, "topic out of range");
    HIPCHK(c, hipSetDevice(c->device));
    if (c->weighted) {
        // red = [a (LD) | nw (LD) | sum and negative-denominator flag of the last updated column]
        if (!c->resid_valid || (t == 0 && !c->resid_fresh)) w_refresh(c);     // once per sweep, as rri_sweep
        enqueue_wT_sums(c, t);
        if (c->pending_wcheck)
            hipLaunchKernelGGL(k_wcheck_wcol, dim3(1), dim3(256), 0, c->stream, (const double*)c->Gpart, c->nwb256, c->k,
                               c->pending_wcheck_topic, 0, t, kparams(c), c->st, c->red + 2 * c->LD);
        else
            HIPCHK(c, hipMemsetAsync(c->red + 2 * c->LD, 0, 2 * sizeof(double), c->stream));
        return RRI_OK;
    }
    if (!c->carry_valid || c->carry_topic != t) {
        // a local column check would see only this rank's rows: keep it pending for the reduced buffer
        const bool pend = c->pending_wcheck;
        const int pt = c->pending_wcheck_topic;
        c->pending_wcheck = false;
        enqueue_prologue(c, t, 0);
        c->pending_wcheck = pend;
        c->pending_wcheck_topic = pt;
        if (pend) return fail(c, RRI_ERR_INVALID, "carry lost while a sharded column check was pending");
    }
    LK::reduce(c);
    return RRI_OK;
}

rri_status rri_reduce_read(rri_ctx* c, double* out, int64_t count) {
    CHECK_CTX(c);
    if (!out || count < 0 || count > c->red_elems) return fail(c, RRI_ERR_INVALID, "bad reduce-buffer range");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(out, c->red, (size_t)count * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return RRI_OK;
}

rri_status rri_reduce_write(rri_ctx* c, const double* in, int64_t count) {
    CHECK_CTX(c);
    if (!in || count < 0 || count > c->red_elems) return fail(c, RRI_ERR_INVALID, "bad reduce-buffer range");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(c->red, in, (size_t)count * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return RRI_OK;
}

rri_status rri_topic_finish(rri_ctx* c, int32_t t) {
    CHECK_CTX(c);
    HIPCHK(c, hipSetDevice(c->device));
    if (c->weighted) {
        if (t >= c->k) return fail(c, RRI_ERR_INVALID, "topic out of range");
        if (c->pending_wcheck)
            hipLaunchKernelGGL(k_wcheck_tail, dim3(1), dim3(64), 0, c->stream, (const double*)(c->red + 2 * c->LD),
                               c->pending_wcheck_topic, 0, t < 0 ? 0 : t, kparams(c), c->st);
        c->pending_wcheck = false;
        if (t < 0) return RRI_OK;
        enqueue_wT_solve(c, 0, t);
        if (!c->prm.fix_W) enqueue_wW_half(c, 0, t, true);      // W fixed: the T row alone (nmf.py:417-458 without :460-476)
        return RRI_OK;
    }
    if (t < 0) {  // only the pending column check, against the (all-reduced) buffer
        if (c->pending_wcheck) {
            LK::check_prev_only(c, c->pending_wcheck_topic, 0, 0);
            c->pending_wcheck = false;
        }
        return RRI_OK;
    }
    if (t >= c->k) return fail(c, RRI_ERR_INVALID, "topic out of range");
    const int chk = c->pending_wcheck ? 1 : 0;
    // W fixed: no W half follows that would finish the row checks (they ride with the Gram row of T there), and the kept
    // column takes the row's scale (nmf.py:450-452) -- as enqueue_T_half does
    LK::trow(c, t, chk, c->pending_wcheck_topic, 0, c->prm.fix_W != 0);
    c->pending_wcheck = false;
    c->carry_valid = false;
    c->q_valid = false; c->gfull_valid = false;
    c->xy_valid = false;
    c->obj_track_valid = false;
    if (c->prm.fix_W) {
        if (no_regs(c)) LK::scale_wcol(c, t);
        return RRI_OK;
    }
    enqueue_W_half(c, 0, t);
    return RRI_OK;
}

rri_status rri_topic_finish_w(rri_ctx* c, int32_t t) {
    CHECK_CTX(c);
    if (t < 0 || t >= c->k) return fail(c, RRI_ERR_INVALID, "topic out of range");
    HIPCHK(c, hipSetDevice(c->device));
    if (c->prm.fix_W) return RRI_OK;   // W fixed: the step has no W half (the reset has rewritten row and column, nmf.py:770-783)
    if (c->weighted) {           // after a T-row reset: E is rebuilt from the new row and column
        if (!c->resid_valid) w_refresh(c);
        enqueue_wW_half(c, 0, t, true);
        return RRI_OK;
    }
    c->skip_row_finish = true;   // the T-row sums of this topic predate the reset
    enqueue_W_half(c, 0, t);
    return RRI_OK;
}

rri_status rri_resid_row_argmax(rri_ctx* c, double* value, int64_t* local_row) {
    CHECK_CTX(c);
    if (!value || !local_row) return fail(c, RRI_ERR_INVALID, "NULL output");
    HIPCHK(c, hipSetDevice(c->device));
    if (!c->rowpos) HIPCHK(c, hipMalloc((void**)&c->rowpos, (size_t)c->n * sizeof(double)));
    DISPATCH(c, L::resid(c, false, false, nullptr, c->rowpos));
    hipLaunchKernelGGL(k_vec_sum_argmax, dim3(1), dim3(1024), 0, c->stream, (const double*)c->rowpos, c->n,
                       (double*)nullptr, c->itmp);
    i64 mi = -1;
    HIPCHK(c, hipMemcpyAsync(&mi, c->itmp, sizeof(i64), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (mi < 0 || mi >= c->n) return fail(c, RRI_ERR_INVALID, "arg-max out of range");
    HIPCHK(c, hipMemcpyAsync(value, c->rowpos + mi, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    *local_row = mi;
    return RRI_OK;
}

rri_status rri_reset_row(rri_ctx* c, int64_t local_row, double* row_out_host) {
    CHECK_CTX(c);
    if (!row_out_host || local_row < 0 || local_row >= c->n) return fail(c, RRI_ERR_INVALID, "bad row");
    HIPCHK(c, hipSetDevice(c->device));
    const i64 mi = local_row;
    HIPCHK(c, hipMemcpyAsync(c->itmp, &mi, sizeof(i64), hipMemcpyHostToDevice, c->stream));
    DISPATCH(c, L::reset_row(c));
    HIPCHK(c, hipMemcpyAsync(row_out_host, c->xraw, (size_t)c->d * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return RRI_OK;
}

rri_status rri_poll(rri_ctx* c) {
    CHECK_CTX(c);
    HIPCHK(c, hipSetDevice(c->device));
    DevState s;
    rri_status r = read_state(c, &s);
    if (r != RRI_OK) return r;
    c->run_total = 0;
    r = status_from_halt(c, s, nullptr);
    if (s.halt < 0) (void)clear_halt(c);   // an event stays latched until rri_apply_reset_* / rri_skip_reset
    return r;
}

// ---- the communicator (one per process and group of ranks) ------------------------------------------------------
rri_status rri_comm_unique_id(uint8_t* id_out) {
    if (!id_out) return RRI_ERR_INVALID;
    RcclApi& a = rccl_api();
    if (!a.ok) return fail(nullptr, RRI_ERR_COMM, "%s", a.err.c_str());
    static_assert(sizeof(ncclUniqueId) == RRI_COMM_ID_BYTES, "RRI_COMM_ID_BYTES must be sizeof(ncclUniqueId)");
    ncclUniqueId id;
    ncclResult_t r = a.GetUniqueId(&id);
    if (r != ncclSuccess) return fail(nullptr, RRI_ERR_COMM, "ncclGetUniqueId: %s", a.GetErrorString(r));
    memcpy(id_out, &id, sizeof id);
    return RRI_OK;
}

rri_status rri_comm_create(rri_comm** out, const uint8_t* id, int32_t rank, int32_t world, int32_t device) {
    if (!out) return RRI_ERR_INVALID;
    *out = nullptr;
    if (!id || world < 1 || rank < 0 || rank >= world) return fail(nullptr, RRI_ERR_INVALID, "bad rank / world / id");
    RcclApi& a = rccl_api();
    if (!a.ok) return fail(nullptr, RRI_ERR_COMM, "%s", a.err.c_str());
    if (hipSetDevice(device) != hipSuccess) return fail(nullptr, RRI_ERR_HIP, "hipSetDevice(%d) failed", device);
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof uid);
    ncclComm_t comm = nullptr;
    ncclResult_t r = a.CommInitRank(&comm, world, uid, rank);
    if (r != ncclSuccess) return fail(nullptr, RRI_ERR_COMM, "ncclCommInitRank(rank %d of %d): %s", rank, world, a.GetErrorString(r));
    rri_comm* m = new rri_comm();
    m->rank = rank; m->world = world; m->device = device; m->nccl = comm;
    *out = m;
    return RRI_OK;
}

rri_status rri_comm_create_host(rri_comm** out, int32_t rank, int32_t world, rri_allreduce_fn allreduce,
                                rri_allgather_fn allgather, rri_broadcast_fn broadcast, void* user) {
    if (!out) return RRI_ERR_INVALID;
    *out = nullptr;
    if (world < 1 || rank < 0 || rank >= world || !allreduce || !allgather || !broadcast)
        return fail(nullptr, RRI_ERR_INVALID, "bad rank / world / callbacks");
    rri_comm* m = new rri_comm();
    m->rank = rank; m->world = world;
    m->h_allreduce = allreduce; m->h_allgather = allgather; m->h_broadcast = broadcast; m->user = user;
    *out = m;
    return RRI_OK;
}

rri_status rri_comm_destroy(rri_comm* m) {
    if (!m) return RRI_OK;
    if (m->nccl) {
        (void)hipSetDevice(m->device);
        (void)rccl_api().CommDestroy(m->nccl);
    }
    delete m;
    return RRI_OK;
}

rri_status rri_attach_comm(rri_ctx* c, rri_comm* comm, int64_t row_offset, int64_t n_global) {
    CHECK_CTX(c);
    if (!comm) {              // detach
        c->comm = nullptr;
        c->row_offset = 0;
        c->n_global = 0;
        invalidate(c);
        return RRI_OK;
    }
    if (row_offset < 0 || n_global < row_offset + c->n) return fail(c, RRI_ERR_INVALID, "row block [%lld, %lld) outside 0..%lld", (long long)row_offset, (long long)(row_offset + c->n), (long long)n_global);
    if (comm->nccl && comm->device != c->device) return fail(c, RRI_ERR_INVALID, "communicator lives on device %d, handle on %d", comm->device, c->device);
    HIPCHK(c, hipSetDevice(c->device));
    if (!c->ctail) {
        HIPCHK(c, hipMalloc((void**)&c->ctail, 8 * sizeof(double)));
        HIPCHK(c, hipMemsetAsync(c->ctail, 0, 8 * sizeof(double), c->stream));
    }
    if (c->cand) { (void)hipFree(c->cand); c->cand = nullptr; }
    HIPCHK(c, hipMalloc((void**)&c->cand, (size_t)2 * comm->world * sizeof(double)));
    c->comm = comm;
    c->row_offset = row_offset;
    c->n_global = n_global;
    c->comm_status = RRI_OK;
    invalidate(c);
    return RRI_OK;
}

rri_status rri_comm_broadcast(rri_ctx* c, double* host, int64_t count, int32_t root) {
    CHECK_CTX(c);
    if (!c->comm) return RRI_OK;                       // one rank: nothing to do
    if (!host || count < 1 || root < 0 || root >= c->comm->world) return fail(c, RRI_ERR_INVALID, "bad broadcast arguments");
    HIPCHK(c, hipSetDevice(c->device));
    DevTmp buf;
    HIPCHK(c, buf.alloc((size_t)count * sizeof(double)));
    HIPCHK(c, hipMemcpyAsync(buf.p, host, (size_t)count * 8, hipMemcpyHostToDevice, c->stream));
    comm_broadcast(c, (double*)buf.p, count, root);
    HIPCHK(c, hipMemcpyAsync(host, buf.p, (size_t)count * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->comm_status != RRI_OK) { const rri_status r = c->comm_status; c->comm_status = RRI_OK; return r; }
    return RRI_OK;
}

rri_status rri_comm_allreduce_sum(rri_ctx* c, double* host, int64_t count) {
    CHECK_CTX(c);
    if (!host || count < 1) return fail(c, RRI_ERR_INVALID, "bad all-reduce arguments");
    HIPCHK(c, hipSetDevice(c->device));
    if (count <= 8 || !c->comm) return comm_allreduce_host(c, host, count);
    DevTmp buf;                                        // larger host arrays (the d x m panels of a row-sharded start)
    HIPCHK(c, buf.alloc((size_t)count * sizeof(double)));
    HIPCHK(c, hipMemcpyAsync(buf.p, host, (size_t)count * 8, hipMemcpyHostToDevice, c->stream));
    comm_allreduce(c, (double*)buf.p, count);
    HIPCHK(c, hipMemcpyAsync(host, buf.p, (size_t)count * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->comm_status != RRI_OK) { const rri_status r = c->comm_status; c->comm_status = RRI_OK; return r; }
    return RRI_OK;
}

rri_status rri_comm_stats(rri_ctx* c, int32_t* rank, int32_t* world, int64_t* allreduce_calls) {
    CHECK_CTX(c);
    if (rank) *rank = c->comm ? c->comm->rank : 0;
    if (world) *world = c->comm ? c->comm->world : 1;
    if (allreduce_calls) *allreduce_calls = c->comm ? c->comm->n_allreduce : 0;
    return RRI_OK;
}

// ---- measurement ---------------------------------------------------------------------------------------------
rri_status rri_onchip_info(rri_ctx* c, int32_t* eligible, int64_t* launches) {
    CHECK_CTX(c);
    if (eligible) *eligible = (c->have_X && c->have_params && onchip_ok(c)) ? 1 : 0;
    if (launches) *launches = c->onchip_launches;
    return RRI_OK;
}

rri_status rri_debug_xcc(rri_ctx* c, int32_t* out, int32_t count) {
    CHECK_CTX(c);
    if (!out || count < 1 || count > 4096) return fail(c, RRI_ERR_INVALID, "bad count");
    HIPCHK(c, hipSetDevice(c->device));
    DevTmp dv;
    HIPCHK(c, dv.alloc((size_t)count * sizeof(int)));
    hipLaunchKernelGGL(k_xcc_probe, dim3((unsigned)count), dim3(64), 0, c->stream, (int*)dv.p);
    HIPCHK(c, hipMemcpyAsync(out, dv.p, (size_t)count * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return RRI_OK;
}

rri_status rri_onchip_fallbacks(rri_ctx* c, int64_t* fallbacks) {
    CHECK_CTX(c);
    if (fallbacks) *fallbacks = c->onchip_fallbacks;
    return RRI_OK;
}

rri_status rri_timing_enable(rri_ctx* c, int32_t on) {
    CHECK_CTX(c);
    c->timing = on < 0 ? 0 : on;
    for (int i = 0; i < 4; ++i) c->timing_seq[i] = 0;
    return RRI_OK;
}

rri_status rri_timing_read(rri_ctx* c, int32_t id, int64_t* launches, double* total_ms) {
    CHECK_CTX(c);
    if (id < 0 || id > 3) return fail(c, RRI_ERR_INVALID, "kernel_id out of range");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    double tot = 0.0;
    for (auto& tl : c->timed[id]) {
        float ms = 0.f;
        HIPCHK(c, hipEventElapsedTime(&ms, tl.a, tl.b));
        tot += ms;
        c->event_pool.push_back(tl.a);
        c->event_pool.push_back(tl.b);
    }
    if (launches) *launches = (int64_t)c->timed[id].size();
    if (total_ms) *total_ms = tot;
    c->timed[id].clear();
    return RRI_OK;
}

rri_status rri_synchronize(rri_ctx* c) {
    CHECK_CTX(c);
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return RRI_OK;
}

rri_status rri_bench_stream_copy(rri_ctx* c, int32_t reps, double* avg_ms) {
    CHECK_CTX(c);
    if (!c->have_X || reps < 1) return fail(c, RRI_ERR_INVALID, "X must be set and reps >= 1");
    HIPCHK(c, hipSetDevice(c->device));
    const size_t bytes = (size_t)c->n * c->ldx * c->es;
    void* dst = nullptr;
    HIPCHK(c, hipMalloc(&dst, bytes));
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    const i64 nvec = (i64)(bytes / 16);
    hipLaunchKernelGGL(k_stream_copy, dim3(256 * 8), dim3(256), 0, c->stream, (const float4*)c->X, (float4*)dst, nvec);
    (void)hipEventRecord(a, c->stream);
    for (int r = 0; r < reps; ++r)
        hipLaunchKernelGGL(k_stream_copy, dim3(256 * 8), dim3(256), 0, c->stream, (const float4*)c->X, (float4*)dst, nvec);
    (void)hipEventRecord(b, c->stream);
    hipError_t e = hipStreamSynchronize(c->stream);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, a, b);
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    (void)hipFree(dst);
    if (e != hipSuccess) return fail(c, RRI_ERR_HIP, "stream copy failed: %s", hipGetErrorString(e));
    if (avg_ms) *avg_ms = ms / reps;
    return RRI_OK;
}

rri_status rri_bench_rank1_update(rri_ctx* c, int32_t reps, double* avg_ms) {
    CHECK_CTX(c);
    if (!c->have_X || !c->have_W || !c->have_T || reps < 1 || c->weighted)
        return fail(c, RRI_ERR_INVALID, "an unweighted handle with X, W, T set and reps >= 1");
    HIPCHK(c, hipSetDevice(c->device));
    // scratch residual R = copy of X; every repetition folds the rank-one term w_0 t_0^T of the handle's own factors
    // into it (non-trivial row and column factors; R stays finite: it moves by reps * w_0 t_0^T) and takes the row
    // dots against T[0,:] and the column sums against W[:,0] of the result -- the work of rri_residual_update
    const size_t bytes = (size_t)c->n * c->ldx * c->es;
    void* R = nullptr;
    HIPCHK(c, hipMalloc(&R, bytes));
    (void)hipMemcpyAsync(R, c->X, bytes, hipMemcpyDeviceToDevice, c->stream);
    (void)hipMemsetAsync(c->st, 0, 16, c->stream);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const int tm = c->timing;
    c->timing = 0;
    auto once = [&]() {
        DISPATCH(c, {
            typename L::Upd u;
            u.a = c->W;
            u.b = c->T;
            L::rank_update(c, R, c->ldx, u, c->T, c->W);
        });
    };
    once();
    (void)hipEventRecord(e0, c->stream);
    for (int r = 0; r < reps; ++r) once();
    (void)hipEventRecord(e1, c->stream);
    hipError_t e = hipStreamSynchronize(c->stream);
    c->timing = tm;
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipFree(R);
    invalidate(c);
    if (e != hipSuccess) return fail(c, RRI_ERR_HIP, "rank-one bench failed: %s", hipGetErrorString(e));
    if (avg_ms) *avg_ms = ms / reps;
    return RRI_OK;
}

}  // extern "C"
