/* The C ABI used from plain C (no Python, no torch): 3 sweeps on a small planted problem, the objective must fall and
 * the factors must stay non-negative and finite.  Prints "ok <objective before> <objective after> <checksum W> <checksum T>".
 * Build: gcc -std=c99 -I include tests/c/abi_smoke.c -o abi_smoke -L rri_nmf_amd/lib -lrri_hip -lm */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include "rri_hip.h"

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
    free(Ws); free(Ts); free(X); free(W); free(T);
    return 0;
}
