/* The C ABI used from plain C (no Python, no torch):
 *  1. 3 sweeps on a small planted problem: the objective must fall, the factors stay non-negative and finite.
 *     Prints "ok <objective before> <objective after> <checksum W> <checksum T>".
 *  2. the same problem ROW-SHARDED over two handles in this one process (two threads, one GPU), joined by the library's
 *     host-callback communicator (rri_comm_create_host; the callbacks add the two threads' buffers behind a barrier):
 *     rri_sweep on each handle, collectives inside.  Prints "sharded <rel. distance of W> <of T> from the one-handle run".
 *  3. the rank-one residual update as an operation (RRI_UNWEIGHTED_RESIDUAL): R = X - W T, then R <- R - a b^T, against
 *     the same arithmetic in C.  Prints "residual <largest difference in fp32 ulps of the largest entry>".
 * Build: gcc -std=c99 -pthread -I include tests/c/abi_smoke.c -o abi_smoke -L rri_nmf_amd/lib -lrri_hip -lm */
#define _POSIX_C_SOURCE 200809L
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "rri_hip.h"

/* ---- a two-rank transport inside one process: the three functions rri_comm_create_host wants ----------------- */
#define SLOT 65536
typedef struct { double slot[2][SLOT]; pthread_barrier_t bar; } shared_t;
typedef struct { int rank; shared_t* sh; } rank_t;
static int32_t cb_allreduce(void* user, double* buf, int64_t count) {
    rank_t* r = (rank_t*)user;
    if (count > SLOT) return 1;
    memcpy(r->sh->slot[r->rank], buf, (size_t)count * sizeof(double));
    pthread_barrier_wait(&r->sh->bar);
    for (int64_t i = 0; i < count; ++i) buf[i] = r->sh->slot[0][i] + r->sh->slot[1][i];   /* same order on both ranks */
    pthread_barrier_wait(&r->sh->bar);
    return 0;
}
static int32_t cb_allgather(void* user, const double* send, int64_t count, double* recv) {
    rank_t* r = (rank_t*)user;
    if (count > SLOT) return 1;
    memcpy(r->sh->slot[r->rank], send, (size_t)count * sizeof(double));
    pthread_barrier_wait(&r->sh->bar);
    for (int q = 0; q < 2; ++q) memcpy(recv + q * count, r->sh->slot[q], (size_t)count * sizeof(double));
    pthread_barrier_wait(&r->sh->bar);
    return 0;
}
static int32_t cb_broadcast(void* user, double* buf, int64_t count, int32_t root) {
    rank_t* r = (rank_t*)user;
    if (count > SLOT) return 1;
    if (r->rank == root) memcpy(r->sh->slot[root], buf, (size_t)count * sizeof(double));
    pthread_barrier_wait(&r->sh->bar);
    memcpy(buf, r->sh->slot[root], (size_t)count * sizeof(double));
    pthread_barrier_wait(&r->sh->bar);
    return 0;
}
typedef struct {
    rank_t me; int64_t lo, hi, n, d; int32_t k; const float* X; const double* W0; const double* T0; rri_params p;
    double* Wout; double* Tout; int status;
} job_t;
static void* shard_main(void* arg) {
    job_t* j = (job_t*)arg;
    rri_ctx* h = NULL;
    rri_comm* comm = NULL;
    const int64_t nl = j->hi - j->lo;
    j->status = 1;
    if (rri_comm_create_host(&comm, j->me.rank, 2, cb_allreduce, cb_allgather, cb_broadcast, &j->me) != RRI_OK) return NULL;
    if (rri_create(&h, nl, j->d, j->k, RRI_F32, RRI_UNWEIGHTED, 0, NULL) != RRI_OK) return NULL;
    if (rri_upload_X(h, j->X + j->lo * j->d, j->d, RRI_F32) != RRI_OK) return NULL;
    if (rri_set_W(h, j->W0 + j->lo * j->k, j->k, RRI_F64) != RRI_OK || rri_set_T(h, j->T0, j->d, RRI_F64) != RRI_OK) return NULL;
    if (rri_set_params(h, &j->p) != RRI_OK) return NULL;
    if (rri_attach_comm(h, comm, j->lo, j->n) != RRI_OK) return NULL;
    int32_t done = 0;
    if (rri_sweep(h, 3, &done) != RRI_OK || done != 3) { fprintf(stderr, "sharded rri_sweep: %s\n", rri_last_error(h)); return NULL; }
    if (rri_get_W(h, j->Wout + j->lo * j->k, j->k, RRI_F64) != RRI_OK || rri_get_T(h, j->Tout, j->d, RRI_F64) != RRI_OK) return NULL;
    rri_destroy(h);
    rri_comm_destroy(comm);
    j->status = 0;
    return NULL;
}

#define CHECK(call)                                                                    \
    do {                                                                               \
        rri_status s_ = (call);                                                        \
        if (s_ != RRI_OK) {                                                            \
            fprintf(stderr, "%s -> %d: %s\n", #call, (int)s_, rri_last_error(h));      \
            return 1;                                                                  \
        }                                                                              \
    } while (0)

static double lcg(unsigned long long* s) {   /* deterministic U(0,1) */
    *s = *s * 6364136223846793005ULL + 1442695040888963407ULL;
    return (double)((*s >> 11) & ((1ULL << 53) - 1)) / (double)(1ULL << 53);
}

int main(void) {
    const int64_t n = 700, d = 333;
    const int32_t k = 6;
    unsigned long long seed = 42;
    double* Ws = malloc(sizeof(double) * n * k), *Ts = malloc(sizeof(double) * k * d);
    float* X = malloc(sizeof(float) * n * d);
    double* W = malloc(sizeof(double) * n * k), *T = malloc(sizeof(double) * k * d);
    for (int64_t i = 0; i < n * k; ++i) Ws[i] = lcg(&seed) < 0.3 ? lcg(&seed) : 0.0;
    for (int64_t i = 0; i < k * d; ++i) Ts[i] = lcg(&seed) < 0.3 ? lcg(&seed) : 0.0;
    double mean = 0.0;
    for (int64_t i = 0; i < n; ++i)
        for (int64_t j = 0; j < d; ++j) {
            double v = 0.01 * lcg(&seed);
            for (int l = 0; l < k; ++l) v += Ws[i * k + l] * Ts[l * d + j];
            X[i * d + j] = (float)v;
            mean += v;
        }
    mean /= (double)(n * d);
    const double a = sqrt(mean / k);
    for (int64_t i = 0; i < n * k; ++i) W[i] = a * lcg(&seed);
    for (int64_t i = 0; i < k * d; ++i) T[i] = a * lcg(&seed);
    double* W0 = malloc(sizeof(double) * n * k), *T0 = malloc(sizeof(double) * k * d);   /* the start, for parts 2 and 3 */
    memcpy(W0, W, sizeof(double) * n * k);
    memcpy(T0, T, sizeof(double) * k * d);

    rri_ctx* h = NULL;
    if (rri_create(&h, n, d, k, RRI_F32, RRI_UNWEIGHTED, 0, NULL) != RRI_OK) {
        fprintf(stderr, "rri_create: %s\n", rri_last_error(NULL));
        return 1;
    }
    CHECK(rri_upload_X(h, X, d, RRI_F32));
    CHECK(rri_set_W(h, W, k, RRI_F64));
    CHECK(rri_set_T(h, T, d, RRI_F64));
    rri_params p = {0};
    p.reset_method = RRI_RESET_MAX_RESID_DOCUMENT;
    p.resets_left = 23;
    p.eps_div = 1.7763568394002505e-15;
    CHECK(rri_set_params(h, &p));
    double o0 = 0.0, o1 = 0.0;
    CHECK(rri_objective(h, &o0));
    int32_t done = 0;
    rri_status st = rri_sweep(h, 3, &done);
    while (st == RRI_PAUSED) {   /* a reset condition: resolve it as nmf.py:770-776 does, then go on */
        rri_event ev;
        CHECK(rri_pending_event(h, &ev));
        CHECK(rri_apply_reset_max_resid(h, ev.topic, NULL));
        st = rri_resume(h, &done);
    }
    if (st != RRI_OK || done != 3) { fprintf(stderr, "rri_sweep -> %d (%d sweeps): %s\n", (int)st, (int)done, rri_last_error(h)); return 1; }
    CHECK(rri_objective(h, &o1));
    CHECK(rri_get_W(h, W, k, RRI_F64));
    CHECK(rri_get_T(h, T, d, RRI_F64));
    double cw = 0.0, ct = 0.0;
    for (int64_t i = 0; i < n * k; ++i) { if (!(W[i] >= 0.0) || !isfinite(W[i])) { fprintf(stderr, "bad W\n"); return 1; } cw += W[i]; }
    for (int64_t i = 0; i < k * d; ++i) { if (!(T[i] >= 0.0) || !isfinite(T[i])) { fprintf(stderr, "bad T\n"); return 1; } ct += T[i]; }
    CHECK(rri_destroy(h));
    if (!(o1 < o0)) { fprintf(stderr, "objective did not fall: %g -> %g\n", o0, o1); return 1; }
    printf("ok %.10e %.10e %.10e %.10e\n", o0, o1, cw, ct);

    /* ---- 2. two row blocks (420 + 280 rows), two handles, two threads, the collective inside rri_sweep ---------- */
    {
        shared_t* sh = malloc(sizeof(shared_t));
        pthread_barrier_init(&sh->bar, NULL, 2);
        double* W2 = malloc(sizeof(double) * n * k), *T2 = malloc(sizeof(double) * 2 * k * d);
        job_t jobs[2];
        pthread_t th[2];
        for (int r = 0; r < 2; ++r) {
            jobs[r].me.rank = r; jobs[r].me.sh = sh;
            jobs[r].lo = r == 0 ? 0 : 420; jobs[r].hi = r == 0 ? 420 : n; jobs[r].n = n; jobs[r].d = d; jobs[r].k = k;
            jobs[r].X = X; jobs[r].W0 = W0; jobs[r].T0 = T0; jobs[r].p = p;
            jobs[r].Wout = W2; jobs[r].Tout = T2 + (int64_t)r * k * d;
            pthread_create(&th[r], NULL, shard_main, &jobs[r]);
        }
        for (int r = 0; r < 2; ++r) pthread_join(th[r], NULL);
        if (jobs[0].status || jobs[1].status) { fprintf(stderr, "sharded run failed\n"); return 1; }
        double dw = 0.0, nw = 0.0, dt = 0.0, nt = 0.0;
        for (int64_t i = 0; i < n * k; ++i) { dw += (W2[i] - W[i]) * (W2[i] - W[i]); nw += W[i] * W[i]; }
        for (int64_t i = 0; i < k * d; ++i) {
            if (T2[i] != T2[k * d + i]) { fprintf(stderr, "T differs between the ranks\n"); return 1; }   /* replicated, bit for bit */
            dt += (T2[i] - T[i]) * (T2[i] - T[i]); nt += T[i] * T[i];
        }
        printf("sharded %.3e %.3e\n", sqrt(dw / nw), sqrt(dt / nt));
        pthread_barrier_destroy(&sh->bar);
        free(sh); free(W2); free(T2);
    }

    /* ---- 3. the rank-one residual update as an operation -------------------------------------------------------- */
    {
        CHECK(rri_create(&h, n, d, k, RRI_F32, RRI_UNWEIGHTED_RESIDUAL, 0, NULL));
        CHECK(rri_upload_X(h, X, d, RRI_F32));
        CHECK(rri_set_W(h, W0, k, RRI_F64));
        CHECK(rri_set_T(h, T0, d, RRI_F64));
        CHECK(rri_set_params(h, &p));
        CHECK(rri_residual_rebuild(h));
        float* R0 = malloc(sizeof(float) * n * d), *R1 = malloc(sizeof(float) * n * d);
        double* av = malloc(sizeof(double) * n), *bv = malloc(sizeof(double) * d), *y = malloc(sizeof(double) * n), *z = malloc(sizeof(double) * d);
        for (int64_t i = 0; i < n; ++i) av[i] = lcg(&seed) - 0.3;
        for (int64_t jx = 0; jx < d; ++jx) bv[jx] = lcg(&seed) - 0.3;
        CHECK(rri_get_residual(h, R0, d, RRI_F32));
        CHECK(rri_residual_update(h, av, bv, NULL, NULL, bv, av, y, z));
        CHECK(rri_get_residual(h, R1, d, RRI_F32));
        double worst = 0.0, big = 0.0, ydiff = 0.0, ynorm = 0.0;
        for (int64_t i = 0; i < n; ++i) {
            double yi = 0.0;
            for (int64_t jx = 0; jx < d; ++jx) {
                const double want = (double)(float)((double)R0[i * d + jx] - av[i] * bv[jx]);
                const double diff = fabs((double)R1[i * d + jx] - want);
                if (diff > worst) worst = diff;
                if (fabs(want) > big) big = fabs(want);
                yi += (double)R1[i * d + jx] * bv[jx];
            }
            ydiff += (yi - y[i]) * (yi - y[i]); ynorm += yi * yi;
        }
        CHECK(rri_destroy(h));
        printf("residual %.3f %.3e\n", worst / (big * 1.1920928955078125e-7), sqrt(ydiff / ynorm));
        free(R0); free(R1); free(av); free(bv); free(y); free(z);
    }
    free(Ws); free(Ts); free(X); free(W); free(T); free(W0); free(T0);
    return 0;
}
