/* sfm_oracle.c -- CPU oracle in C (float64, OpenMP).  TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * The same restatement as oracle/sfm_oracle.py, loop-form, for sizes the NumPy oracle is too slow for and as
 * the `cpu_baseline` (kind "port") that bench.py times on the GPU node's host cores.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  It is itself gated against the golden
 * vectors of the reference (tests/test_oracle_golden.py::test_c_oracle_matches_golden).
 *
 * Citations are to the upstream repository felixlutz/carla-social-force-model.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct { double lam, A, gamma, n, n_prime, epsilon, thr; } OIx;

typedef struct {
    int use_ped_radius;
    double max_speed_factor, tau, dt;
    int enabled[5];          /* acceleration, pedestrian, border, static, dynamic */
    OIx ped, stat, dyn;
    double border_a, border_b;
} OParams;

typedef struct { int K; const int32_t* off; const double* pts; const double* ctr; const double* extra; } OGeo;
/* borders: ctr = K x {cx, cy}, extra = K x {section_length};  obstacles: ctr = K x {cx, cy}, extra = K x {vx, vy} or NULL */

/* stateutils.normalize (stateutils.py:78-92) for one vector of k components */
static double unit(const double* a, int k, double* out) {
    double s = 0.0;
    for (int c = 0; c < k; ++c) s += a[c] * a[c];
    const double n = sqrt(s), div = (n == 0.0) ? 1.0 : n;
    for (int c = 0; c < k; ++c) out[c] = a[c] / div;
    return n;
}

/* forces.py:85-115 / :241-270.  e: unit direction (k comps), dist, dv = v_self - v_other.  Adds into F[0..k). */
static double moussaid(const OIx* p, const double* e, double dist, const double* dv, int k, double* F, double theta_tol,
                       double* mag, double* plain) {
    double D[3] = {0, 0, 0}, t[3] = {0, 0, 0};
    for (int c = 0; c < k; ++c) D[c] = p->lam * dv[c] + e[c];
    const double Dn = unit(D, k, t);
    double raw = atan2(e[1], e[0]) - atan2(t[1], t[0]);           /* stateutils.py:104-109 */
    double ang = raw;
    if (ang > M_PI) ang -= 2.0 * M_PI;                            /* :111-112 */
    if (ang < -M_PI) ang += 2.0 * M_PI;
    const double B = p->gamma * Dn;
    const double theta = ang + B * (-p->epsilon);
    const double a = -1.0 * dist / B;
    const double fv = -1.0 * p->A * exp(a - (p->n_prime * B * theta) * (p->n_prime * B * theta));
    const double sg = (theta > 0.0) - (theta < 0.0);
    const double ft = -1.0 * p->A * (isnan(theta) ? NAN : sg) * exp(a - (p->n * B * theta) * (p->n * B * theta));
    for (int c = 0; c < k; ++c) F[c] += fv * t[c];
    F[0] += ft * (-t[1]);
    F[1] += ft * t[0];
    if (!isnan(fv) && !isnan(ft)) {   /* magnitude x (1 + first-order fp32 angle-noise amplification / 1e-5), see sfm_oracle.py */
        const double res = 2.384185791015625e-07 / 1e-5, at = fabs(theta);
        const double xv = fabs(a) + (p->n_prime * B * theta) * (p->n_prime * B * theta);
        const double xt = fabs(a) + (p->n * B * theta) * (p->n * B * theta);
        *mag += fabs(fv) * (1.0 + (2.0 * (p->n_prime * B) * (p->n_prime * B) * at + xv) * res)
              + fabs(ft) * (1.0 + (2.0 * (p->n * B) * (p->n * B) * at + xt) * res);
        *plain += fabs(fv) + fabs(ft);                  /* the same sum without the conditioning weights */
    }
    if (theta_tol > 0.0 && (fabs(theta) < theta_tol || fabs(fabs(raw) - M_PI) < theta_tol) && !isnan(ft)) return 2.0 * fabs(ft);
    return 0.0;
}

/* first nearest sampled point: np.argmin over np.linalg.norm (forces.py:154,228) */
static int nearest(const double* pts, int o0, int o1, double x, double y) {
    int best = o0;
    double bd = INFINITY;
    for (int p = o0; p < o1; ++p) {
        const double dx = x - pts[2 * p], dy = y - pts[2 * p + 1];
        const double d = sqrt(dx * dx + dy * dy);
        if (d < bd) { bd = d; best = p; }
    }
    return best;
}

/* One tick for rows [i0, i1).  loc/vel/wp: N x 3.  forces: 6 x n x 3 (5 forces + total), vel_out: n x 3,
 * expo: n (sign/wrap exposure of the Moussaid terms), absum: n (sum of the magnitudes of all force terms, the
 * scale of the fp32 rounding error, each Moussaid term weighted by its fp32 conditioning), plain_absum: n (the same sum
 * unweighted: absum / plain_absum - 1 is the mean conditioning weight a parity test leaned on); any may be NULL. */
int oracle_tick(int N, int i0, int i1, const double* loc, const double* vel, const double* wp, const double* tspeed,
                const double* radius, const uint8_t* crossing, const OParams* P, const OGeo* borders,
                const OGeo* statics, const OGeo* dynamics, double* forces, double* vel_out, double* expo,
                double* absum, double theta_tol, int nthreads, double* plain_absum) {
    const int n = i1 - i0;
    if (n <= 0) return 0;
    memset(forces, 0, sizeof(double) * 18 * (size_t)n);
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel for schedule(dynamic, 16)
    for (int i = i0; i < i1; ++i) {
        const int r = i - i0;
        double* Fa = forces + ((size_t)0 * n + r) * 3;
        double* Fp = forces + ((size_t)1 * n + r) * 3;
        double* Fb = forces + ((size_t)2 * n + r) * 3;
        double* Fs = forces + ((size_t)3 * n + r) * 3;
        double* Fd = forces + ((size_t)4 * n + r) * 3;
        double* Ft = forces + ((size_t)5 * n + r) * 3;
        const double* xi = loc + 3 * (size_t)i;
        const double* vi = vel + 3 * (size_t)i;
        double ex = 0.0, mg = 0.0, pl = 0.0;
        if (P->enabled[0]) {                                    /* forces.py:46-53, stateutils.py:7-15 */
            double to[2] = {wp[3 * (size_t)i] - xi[0], wp[3 * (size_t)i + 1] - xi[1]}, e[2];
            unit(to, 2, e);
            Fa[0] = 1.0 / P->tau * (tspeed[i] * e[0] - vi[0]);
            Fa[1] = 1.0 / P->tau * (tspeed[i] * e[1] - vi[1]);
            Fa[2] = 1.0 / P->tau * (tspeed[i] * 0.0 - vi[2]);
            mg += (fabs(tspeed[i]) + sqrt(vi[0] * vi[0] + vi[1] * vi[1] + vi[2] * vi[2])) / P->tau;
            pl += (fabs(tspeed[i]) + sqrt(vi[0] * vi[0] + vi[1] * vi[1] + vi[2] * vi[2])) / P->tau;
        }
        if (P->enabled[1]) {                                    /* forces.py:74-117 */
            for (int j = 0; j < N; ++j) {
                if (j == i) continue;                            /* stateutils.py:41-49 */
                const double* xj = loc + 3 * (size_t)j;
                const double* vj = vel + 3 * (size_t)j;
                double diff[3] = {xj[0] - xi[0], xj[1] - xi[1], xj[2] - xi[2]}, e[3];
                double dist = unit(diff, 3, e);
                if (P->use_ped_radius) dist -= radius[i] + radius[j];
                double dv[3] = {vi[0] - vj[0], vi[1] - vj[1], vi[2] - vj[2]};
                ex += moussaid(&P->ped, e, dist, dv, 3, Fp, theta_tol, &mg, &pl);
            }
        }
        if (P->enabled[2] && borders && borders->K > 0) {       /* forces.py:138-179 */
            for (int k = 0; k < borders->K; ++k) {
                const double dx = xi[0] - borders->ctr[2 * k], dy = xi[1] - borders->ctr[2 * k + 1];
                if (!(sqrt(dx * dx + dy * dy) < borders->extra[k])) continue;
                const int o0 = borders->off[k], o1 = borders->off[k + 1];
                if (o1 <= o0) continue;
                const int b = nearest(borders->pts, o0, o1, xi[0], xi[1]);
                double d[2] = {xi[0] - borders->pts[2 * b], xi[1] - borders->pts[2 * b + 1]}, e[2];
                double dist = unit(d, 2, e);
                if (P->use_ped_radius) dist -= radius[i];
                const double mag = P->border_a * exp(-1.0 * dist / P->border_b);
                Fb[0] += e[0] * mag;
                Fb[1] += e[1] * mag;
                if (!(crossing && crossing[i])) { mg += fabs(mag); pl += fabs(mag); }
            }
            if (crossing && crossing[i]) { Fb[0] *= 0.0; Fb[1] *= 0.0; }
        }
        for (int which = 0; which < 2; ++which) {               /* forces.py:208-283 */
            const OGeo* g = which ? dynamics : statics;
            const OIx* ix = which ? &P->dyn : &P->stat;
            double* F = which ? Fd : Fs;
            if (!P->enabled[3 + which] || !g || g->K <= 0) continue;
            for (int k = 0; k < g->K; ++k) {
                const double dx = xi[0] - g->ctr[2 * k], dy = xi[1] - g->ctr[2 * k + 1];
                if (!(sqrt(dx * dx + dy * dy) < ix->thr)) continue;
                const int o0 = g->off[k], o1 = g->off[k + 1];
                if (o1 <= o0) continue;
                const int b = nearest(g->pts, o0, o1, xi[0], xi[1]);
                double d[2] = {g->pts[2 * b] - xi[0], g->pts[2 * b + 1] - xi[1]}, e[2];
                double dist = unit(d, 2, e);
                if (P->use_ped_radius) dist -= radius[i];
                double dv[2] = {vi[0] - (g->extra ? g->extra[2 * k] : 0.0), vi[1] - (g->extra ? g->extra[2 * k + 1] : 0.0)};
                double f2[3] = {0, 0, 0};
                ex += moussaid(ix, e, dist, dv, 2, f2, theta_tol, &mg, &pl);
                F[0] += f2[0];
                F[1] += f2[1];
            }
        }
        for (int c = 0; c < 3; ++c) Ft[c] = (((Fa[c] + Fp[c]) + Fb[c]) + Fs[c]) + Fd[c];
        /* pedestrian_simulation.py:117-124 + stateutils.py:18-23 */
        double v[3], s = 0.0;
        for (int c = 0; c < 3; ++c) { v[c] = vi[c] + P->dt * Ft[c]; s += v[c] * v[c]; }
        s = sqrt(s);
        if (s == 0.0) s = 1.0;
        const double fac = fmin(1.0, tspeed[i] * P->max_speed_factor / s);
        for (int c = 0; c < 3; ++c) vel_out[3 * (size_t)r + c] = v[c] * fac;
        if (expo) expo[r] = ex;
        if (absum) absum[r] = mg;
        if (plain_absum) plain_absum[r] = pl;
    }
    return 0;
}

int oracle_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
