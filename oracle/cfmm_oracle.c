/* TEST INFRASTRUCTURE -- CPU oracle in plain C (fp64, optional OpenMP).
 *
 * PARITY UNPINNED: the reference (/root/reference: arbitrage.py, liquidation.py, two-asset.py)
 * is four cvxpy scripts with no tests, no expected outputs and un-pinned third-party solvers
 * (cvxpy + ECOS/Clarabel) that are not installed and not installable in this container.  This
 * file restates the *model* of those scripts in dual-decomposition form; it is pinned against
 * oracle/primal_scipy.py (the scripts' primal model through SciPy SLSQP, no shared algorithm)
 * and against the survey-derived known answers in tests/golden/.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * The product (cfmm-routing-code_amd/) never links, imports or calls it.
 *
 * What is restated (reference file:line -> function here)
 *   per-pool constraint  phi_i(R_i + gamma_i D_i - L_i) >= phi_i(R_i), D_i, L_i >= 0
 *       weighted geo-mean, n assets   arbitrage.py:65, liquidation.py:65, two-asset.py:74 -> pool_geomean_n
 *       Uniswap v2 (equal weights, 2) arbitrage.py:68-70                                  -> pool_geomean2 (wa = 1/2)
 *       constant sum + x >= 0         arbitrage.py:73-74                                  -> pool_sum2
 *       Curve-style x+y-alpha/(xy)    (not in the reference; BASELINE config 5)           -> pool_curve2
 *   net trade psi = sum_i A_i (L_i - D_i)   arbitrage.py:54   -> the scatter-adds in oracle_eval
 *   utility + its constraints               arbitrage.py:57,77; liquidation.py:57,77-80;
 *                                           two-asset.py:66,86 -> unified (c, h, ctype), see oracle_step
 *   prob.solve()                            arbitrage.py:81-82 -> oracle_solve
 *
 * Dual decomposition:  g(nu) = (nu - c)'h + sum_i arb_i(A_i' nu),  minimised over the box the
 * utility allows;  grad = psi(nu) + h.  Iterated in log-prices with a projected L-BFGS whose
 * initial metric is the static diagonal D_j = sum_{pools with j} nu_j R_j (1 - w_j).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define MAXK 16

/* ---------------------------------------------------------------- per-pool subproblems */

/* weighted geo-mean, 2 assets; wa in (0,1), wb = 1 - wa.  y = L - D per leg. */
static inline void pool_geomean2(double Ra, double Rb, double g, double wa, double pa, double pb,
                                 double *ya, double *yb)
{
    double wb = 1.0 - wa;
    double va = wb * pa * Ra, vb = wa * pb * Rb;     /* a->b active iff g*wa*pb*Rb > wb*pa*Ra */
    *ya = 0.0; *yb = 0.0;
    if (g * vb > va) {                               /* tender a, receive b */
        if (wa == 0.5) {
            double x = sqrt(g * (pb / pa) * Ra * Rb);
            *ya = -(x - Ra) / g;  *yb = Rb - Ra * Rb / x;
        } else {
            double eta = wa / wb;
            double x = exp((log(g * eta * (pb / pa) * Rb) + eta * log(Ra)) / (eta + 1.0));
            *ya = -(x - Ra) / g;  *yb = Rb * (1.0 - pow(Ra / x, eta));
        }
    } else if (g * va > vb) {                        /* tender b, receive a */
        if (wa == 0.5) {
            double x = sqrt(g * (pa / pb) * Ra * Rb);
            *yb = -(x - Rb) / g;  *ya = Ra - Ra * Rb / x;
        } else {
            double eta = wb / wa;
            double x = exp((log(g * eta * (pa / pb) * Ra) + eta * log(Rb)) / (eta + 1.0));
            *yb = -(x - Rb) / g;  *ya = Ra * (1.0 - pow(Rb / x, eta));
        }
    }
}

/* weighted geo-mean, k assets: exact root of the piecewise-linear
 *   F(t) = sum_j w_j f(t - a_j),  a_j = log(R_j p_j / w_j),  f(u) = u | 0 | u + lg
 * by evaluating F at its 2k breakpoints and interpolating on the bracketing piece. */
static inline double gm_F(int k, const double *w, const double *a, double lg, double t)
{
    double s = 0.0;
    for (int j = 0; j < k; ++j) {
        double u = t - a[j];
        s += w[j] * (u < 0.0 ? u : (u > -lg ? u + lg : 0.0));
    }
    return s;
}

static inline void pool_geomean_n(int k, const double *R, const double *w, double g, const double *p,
                                  double *y)
{
    double a[MAXK], lg = log(g);
    for (int j = 0; j < k; ++j) a[j] = log(R[j] * p[j] / w[j]);
    /* tL = largest breakpoint with F <= 0, tR = smallest breakpoint with F >= 0 */
    double tL = -DBL_MAX, fL = 0.0, tR = DBL_MAX, fR = 0.0;
    for (int j = 0; j < 2 * k; ++j) {
        double t = (j < k) ? a[j] : a[j - k] - lg;
        double f = gm_F(k, w, a, lg, t);
        if (f <= 0.0 && t > tL) { tL = t; fL = f; }
        if (f >= 0.0 && t < tR) { tR = t; fR = f; }
    }
    double t;
    if (fL == 0.0) t = tL;
    else if (fR == 0.0) t = tR;
    else t = tL - fL * (tR - tL) / (fR - fL);
    double mu = exp(t);
    for (int j = 0; j < k; ++j) {
        double hi = mu * w[j] / p[j], lo = g * hi;
        double x = R[j] < lo ? lo : (R[j] > hi ? hi : R[j]);
        y[j] = (x < R[j]) ? (R[j] - x) : (R[j] - x) / g;
    }
}

/* constant sum, 2 assets: bang-bang.  `state`: 0 = evaluate normally; 1 = tied on its kink
 * (contributes nothing; the caller's recovery assigns the fill fraction). */
static inline void pool_sum2(double Ra, double Rb, double g, double pa, double pb, int tied,
                             double *ya, double *yb)
{
    *ya = 0.0; *yb = 0.0;
    if (tied) return;
    if (g * pb > pa)      { *ya = -Rb / g; *yb = Rb; }      /* tender a, drain b */
    else if (g * pa > pb) { *yb = -Ra / g; *ya = Ra; }      /* tender b, drain a */
}

/* Curve-style: phi = x + y - alpha/(xy).  Safeguarded Newton on x (tendered reserve). */
static inline double curve_y(double x, double C, double al)
{
    double b = C - x;
    return 0.5 * (b + sqrt(b * b + 4.0 * al / x));
}
static inline int curve_dir(double Rin, double Rout, double g, double al, double C,
                            double pin, double pout, double *yin, double *yout)
{
    double rho = pin / (g * pout);
    double m0 = (1.0 + al / (Rin * Rin * Rout)) / (1.0 + al / (Rin * Rout * Rout));
    if (!(m0 > rho)) return 0;
    /* h(x) = m(x, y(x)) - rho is decreasing; bracket [lo, hi] with h(lo) > 0 >= h(hi) */
    double lo = Rin, hi = Rin * 2.0;
    for (int it = 0; it < 200; ++it) {
        double yy = curve_y(hi, C, al);
        double hh = (1.0 + al / (hi * hi * yy)) / (1.0 + al / (hi * yy * yy)) - rho;
        if (hh <= 0.0) break;
        lo = hi; hi *= 2.0;
    }
    double x = lo;
    for (int it = 0; it < 100; ++it) {
        double yy = curve_y(x, C, al);
        double fx = 1.0 + al / (x * x * yy), fy = 1.0 + al / (x * yy * yy);
        double hx = fx / fy - rho;
        if (hx > 0.0) lo = x; else hi = x;
        /* derivative along the curve: dy/dx = -fx/fy */
        double yp = -fx / fy;
        double dfx = -2.0 * al / (x * x * x * yy) - al / (x * x * yy * yy) * yp;
        double dfy = -al / (x * x * yy * yy) - 2.0 * al / (x * yy * yy * yy) * yp;
        double dh = (dfx * fy - fx * dfy) / (fy * fy);
        double xn = x - hx / dh;
        if (!(xn > lo && xn < hi)) xn = 0.5 * (lo + hi);
        if (fabs(xn - x) <= 4e-16 * x) { x = xn; break; }
        x = xn;
    }
    *yin = -(x - Rin) / g;
    *yout = Rout - curve_y(x, C, al);
    return 1;
}
static inline void pool_curve2(double Ra, double Rb, double g, double al, double pa, double pb,
                               double *ya, double *yb)
{
    double C = Ra + Rb - al / (Ra * Rb);
    *ya = 0.0; *yb = 0.0;
    if (curve_dir(Ra, Rb, g, al, C, pa, pb, ya, yb)) return;
    curve_dir(Rb, Ra, g, al, C, pb, pa, yb, ya);
}
/* d(y_k)/d(log p_k) at the no-trade point, used only for the diagonal metric */
static inline void curve_diag(double Ra, double Rb, double al, double pa, double pb, double *da, double *db)
{
    /* price response of a stableswap pool near balance: |dx/dlog m| = m / |dm/dx| */
    double x = Ra, yy = Rb;
    double fx = 1.0 + al / (x * x * yy), fy = 1.0 + al / (x * yy * yy);
    double yp = -fx / fy;
    double dfx = -2.0 * al / (x * x * x * yy) - al / (x * x * yy * yy) * yp;
    double dfy = -al / (x * x * yy * yy) - 2.0 * al / (x * yy * yy * yy) * yp;
    double dm = (dfx * fy - fx * dfy) / (fy * fy);
    double dxdl = (fx / fy) / fabs(dm);          /* units of token a per unit log-price */
    *da = pa * dxdl;
    *db = pb * dxdl * (fx / fy);
}

/* ---------------------------------------------------------------- problem container */

enum { K_CP2 = 0, K_W2 = 1, K_SUM2 = 2, K_CURVE2 = 3, K_POW2 = 4 };

/* power sum  x^(1-t) + y^(1-t)  (not in the reference; the generic bucket's tenant, include/cfmm.h CFMM_POOL_POW2): the
 * CLOSED FORM -- (y/x)^t = pin / (gamma pout) on the level set -- which the library's exact path does not use (it runs
 * a generic root search on the forward exchange function): the two derivations are independent. */
static inline int pow2_dir(double Rin, double Rout, double g, double t, double pin, double pout, double *yin, double *yout)
{
    if (!(g * pout * pow(Rout / Rin, t) > pin)) return 0;
    /* x^q = K / (1 + rho^(q/t)), y = x rho^(1/t), in logarithms of x / R_in and y / R_out: forming R - x directly loses the
     * trade (and the profit, a difference of the two legs' values) of a pool with very unequal reserves to cancellation */
    const double q = 1.0 - t, lrho = log(pin / (g * pout));
    const double a = pow(Rout / Rin, q), b = exp(lrho * q / t);
    const double lx = log1p((a - b) / (1.0 + b)) / q;      /* the price condition fixes the tender ... */
    const double z = expm1(q * lx) / a;                    /* ... and the level set what is received: (x^q - R_in^q) / R_out^q */
    const double ly = log1p(-z) / q;                       /* (y = x rho^(1/t) in logarithms keeps five digits of a 1e-12 change) */
    *yin = -Rin * expm1(lx) / g; *yout = -Rout * expm1(ly);
    return 1;
}
static inline void pool_pow2(double Ra, double Rb, double g, double t, double pa, double pb, double *ya, double *yb)
{
    *ya = 0.0; *yb = 0.0;
    if (pow2_dir(Ra, Rb, g, t, pa, pb, ya, yb)) return;
    pow2_dir(Rb, Ra, g, t, pb, pa, yb, ya);
}
/* its share of the diagonal metric: nu_in L'(0) / |L''(0)| per direction with L' = (y/x)^t, L'' = -t L' (L'/y + 1/x) at
 * gamma = 1 (the general rule of csrc/phi2.hpp: generic_diag) */
static inline void pow2_diag(double Ra, double Rb, double t, double pa, double pb, double *da, double *db)
{
    const double la = pow(Rb / Ra, t), lb = pow(Ra / Rb, t);
    *da = pa * la / (t * la * (la / Rb + 1.0 / Ra));
    *db = pb * lb / (t * lb * (lb / Ra + 1.0 / Rb));
}

typedef struct {
    int kind; int64_t m;
    const double *Ra, *Rb, *fee, *param; const int32_t *ia, *ib;
    const int32_t *tied;                          /* SUM2 only, may be NULL */
} bucket2_t;

typedef struct {
    int k; int64_t m;
    const int32_t *idx; const double *R, *w, *fee;   /* slot-major: [k][m] */
} bucketn_t;

typedef struct {
    int n;
    int nb2, nbn;
    bucket2_t b2[16];
    bucketn_t bn[32];
    /* utility */
    double *c, *h; int32_t *ctype;
    /* ties: token j -> group grp[j], log-offset off[j] */
    int ng; int32_t *grp; double *off; double *glo, *ghi;
    /* L-BFGS state */
    int M, hist, head;
    double *S, *Y, *rho;
    double *s, *s_t, *Gs, *Gs_t, *d, *Ds, *nu, *psi, *diag, *q, *r, *alpha;
    double f, t_step, f_t;
    int iter, evals, status, first;
    double gap, infeas, pg;
    double primal; int general;   /* primal value f - gap term of the accepted point; the utility has table entries (ctype >= 3) */
    int nthreads;
} oracle_t;

oracle_t *oracle_create(int n)
{
    oracle_t *o = (oracle_t *)calloc(1, sizeof(oracle_t));
    o->n = n; o->ng = n; o->M = 8;
    o->c = calloc(n, 8); o->h = calloc(n, 8); o->ctype = calloc(n, 4);
    o->grp = calloc(n, 4); o->off = calloc(n, 8); o->glo = calloc(n, 8); o->ghi = calloc(n, 8);
    for (int j = 0; j < n; ++j) { o->grp[j] = j; }
    o->S = calloc((size_t)o->M * n, 8); o->Y = calloc((size_t)o->M * n, 8); o->rho = calloc(o->M, 8);
    o->alpha = calloc(o->M, 8);
    double **v[] = { &o->s, &o->s_t, &o->Gs, &o->Gs_t, &o->d, &o->Ds, &o->nu, &o->psi, &o->diag, &o->q, &o->r };
    for (unsigned i = 0; i < sizeof(v) / sizeof(v[0]); ++i) *v[i] = calloc(n, 8);
    o->nthreads = 1;
    return o;
}
void oracle_destroy(oracle_t *o)
{
    if (!o) return;
    free(o->c); free(o->h); free(o->ctype); free(o->grp); free(o->off); free(o->glo); free(o->ghi);
    free(o->S); free(o->Y); free(o->rho); free(o->alpha);
    free(o->s); free(o->s_t); free(o->Gs); free(o->Gs_t); free(o->d); free(o->Ds); free(o->nu);
    free(o->psi); free(o->diag); free(o->q); free(o->r);
    free(o);
}
void oracle_set_threads(oracle_t *o, int t) { o->nthreads = t < 1 ? 1 : t; }

/* the oracle borrows the caller's arrays (they must outlive it) */
int oracle_add_pools2(oracle_t *o, int kind, int64_t m, const double *Ra, const double *Rb,
                      const double *fee, const double *param, const int32_t *ia, const int32_t *ib,
                      const int32_t *tied)
{
    if (o->nb2 >= 16) return -1;
    bucket2_t b = { kind, m, Ra, Rb, fee, param, ia, ib, tied };
    o->b2[o->nb2++] = b;
    return 0;
}
int oracle_add_poolsN(oracle_t *o, int k, int64_t m, const int32_t *idx, const double *R,
                      const double *w, const double *fee)
{
    if (o->nbn >= 32 || k > MAXK) return -1;
    bucketn_t b = { k, m, idx, R, w, fee };
    o->bn[o->nbn++] = b;
    return 0;
}
void oracle_clear_pools(oracle_t *o) { o->nb2 = 0; o->nbn = 0; }

/* unified utility:  maximise c'psi  s.t.  psi_k + h_k >= 0 (ctype 0) | = 0 (1) | free (2)
 *   arbitrage   (arbitrage.py:57,77)      c = market value, h = 0, all ctype 0
 *   liquidation (liquidation.py:57,77-80) c = e_t, h = assets (h_t = 0), ctype 1, ctype[t] = 2
 *   swap        (two-asset.py:66,86)      c = e_t, h = assets, all ctype 0
 * dual box: ctype 0 -> nu >= c;  ctype 1 -> nu free (> 0);  ctype 2 -> nu = c. */
void oracle_set_utility(oracle_t *o, const double *c, const double *h, const int32_t *ctype)
{
    memcpy(o->c, c, 8 * o->n); memcpy(o->h, h, 8 * o->n); memcpy(o->ctype, ctype, 4 * o->n);
    o->general = 0;
    for (int j = 0; j < o->n; ++j) if (ctype[j] >= 3) o->general = 1;
}
/* ties (kinks of constant-sum pools): log nu_j = s[grp[j]] + off[j] */
void oracle_set_ties(oracle_t *o, int ng, const int32_t *grp, const double *off)
{
    o->ng = ng; memcpy(o->grp, grp, 4 * o->n); memcpy(o->off, off, 8 * o->n);
}

/* ---------------------------------------------------------------- one dual evaluation
 * psi[n] = sum_i A_i (L_i - D_i),  returns sum_i arb_i;  diag (optional): static metric. */
double oracle_eval(oracle_t *o, const double *nu, double *psi, double *diag)
{
    const int n = o->n;
    double ftot = 0.0;
    memset(psi, 0, 8 * n);
    if (diag) memset(diag, 0, 8 * n);
#ifdef _OPENMP
#pragma omp parallel num_threads(o->nthreads) reduction(+ : ftot)
#endif
    {
        double *lp = calloc(n, 8), *ld = diag ? calloc(n, 8) : NULL;
        for (int b = 0; b < o->nb2; ++b) {
            const bucket2_t *B = &o->b2[b];
#ifdef _OPENMP
#pragma omp for schedule(static) nowait
#endif
            for (int64_t i = 0; i < B->m; ++i) {
                int ia = B->ia[i], ib = B->ib[i];
                double pa = nu[ia], pb = nu[ib], ya, yb;
                double Ra = B->Ra[i], Rb = B->Rb[i], g = B->fee[i];
                switch (B->kind) {
                case K_CP2:    pool_geomean2(Ra, Rb, g, 0.5, pa, pb, &ya, &yb); break;
                case K_W2:     pool_geomean2(Ra, Rb, g, B->param[i], pa, pb, &ya, &yb); break;
                case K_SUM2:   pool_sum2(Ra, Rb, g, pa, pb, B->tied ? B->tied[i] : 0, &ya, &yb); break;
                case K_POW2:   pool_pow2(Ra, Rb, g, B->param[i], pa, pb, &ya, &yb); break;
                default:       pool_curve2(Ra, Rb, g, B->param[i], pa, pb, &ya, &yb); break;
                }
                lp[ia] += ya; lp[ib] += yb;
                ftot += pa * ya + pb * yb;
                if (ld) {
                    if (B->kind == K_CP2) { ld[ia] += 0.5 * pa * Ra; ld[ib] += 0.5 * pb * Rb; }
                    else if (B->kind == K_W2) { double wa = B->param[i]; ld[ia] += (1.0 - wa) * pa * Ra; ld[ib] += wa * pb * Rb; }
                    else if (B->kind == K_CURVE2) { double da, db; curve_diag(Ra, Rb, B->param[i], pa, pb, &da, &db); ld[ia] += da; ld[ib] += db; }
                    else if (B->kind == K_POW2) { double da, db; pow2_diag(Ra, Rb, B->param[i], pa, pb, &da, &db); ld[ia] += da; ld[ib] += db; }
                }
            }
        }
        for (int b = 0; b < o->nbn; ++b) {
            const bucketn_t *B = &o->bn[b];
            const int k = B->k; const int64_t m = B->m;
#ifdef _OPENMP
#pragma omp for schedule(static) nowait
#endif
            for (int64_t i = 0; i < m; ++i) {
                double R[MAXK], w[MAXK], p[MAXK], y[MAXK];
                for (int j = 0; j < k; ++j) { R[j] = B->R[j * m + i]; w[j] = B->w[j * m + i]; p[j] = nu[B->idx[j * m + i]]; }
                pool_geomean_n(k, R, w, B->fee[i], p, y);
                for (int j = 0; j < k; ++j) {
                    int t = B->idx[j * m + i];
                    lp[t] += y[j]; ftot += p[j] * y[j];
                    if (ld) ld[t] += (1.0 - w[j]) * p[j] * R[j];
                }
            }
        }
#ifdef _OPENMP
#pragma omp critical
#endif
        {
            for (int j = 0; j < n; ++j) psi[j] += lp[j];
            if (ld) for (int j = 0; j < n; ++j) diag[j] += ld[j];
        }
        free(lp); free(ld);
    }
    return ftot;
}

/* per-pool trades at prices nu (the read-back of deltas[i].value / lambdas[i].value,
 * two-asset.py:94,98): y = L - D;  D = max(-y, 0), L = max(y, 0). */
void oracle_trades2(oracle_t *o, int b, const double *nu, double *ya, double *yb)
{
    const bucket2_t *B = &o->b2[b];
    for (int64_t i = 0; i < B->m; ++i) {
        double pa = nu[B->ia[i]], pb = nu[B->ib[i]];
        switch (B->kind) {
        case K_CP2:    pool_geomean2(B->Ra[i], B->Rb[i], B->fee[i], 0.5, pa, pb, &ya[i], &yb[i]); break;
        case K_W2:     pool_geomean2(B->Ra[i], B->Rb[i], B->fee[i], B->param[i], pa, pb, &ya[i], &yb[i]); break;
        case K_SUM2:   pool_sum2(B->Ra[i], B->Rb[i], B->fee[i], pa, pb, B->tied ? B->tied[i] : 0, &ya[i], &yb[i]); break;
        case K_POW2:   pool_pow2(B->Ra[i], B->Rb[i], B->fee[i], B->param[i], pa, pb, &ya[i], &yb[i]); break;
        default:       pool_curve2(B->Ra[i], B->Rb[i], B->fee[i], B->param[i], pa, pb, &ya[i], &yb[i]); break;
        }
    }
}
void oracle_tradesN(oracle_t *o, int b, const double *nu, double *y /* [k][m] */)
{
    const bucketn_t *B = &o->bn[b];
    const int k = B->k; const int64_t m = B->m;
    for (int64_t i = 0; i < m; ++i) {
        double R[MAXK], w[MAXK], p[MAXK], yy[MAXK];
        for (int j = 0; j < k; ++j) { R[j] = B->R[j * m + i]; w[j] = B->w[j * m + i]; p[j] = nu[B->idx[j * m + i]]; }
        pool_geomean_n(k, R, w, B->fee[i], p, yy);
        for (int j = 0; j < k; ++j) y[j * m + i] = yy[j];
    }
}

/* ---------------------------------------------------------------- outer iteration
 * Projected L-BFGS on the group variables s (log nu_j = s[grp[j]] + off[j]).
 * One call consumes one evaluation (f_pools, psi at the trial point) and produces the next
 * trial point.  The device update kernel mirrors this function statement by statement. */

typedef struct {
    double tol_gap, tol_infeas, armijo, max_step;
    int max_evals, memory, pg_rule, pad;
} oracle_opts_t;

typedef struct {
    int evals, iters, status;      /* status: 0 running, 1 converged, 2 stalled, 3 max evals */
    double dual_value, primal_value, gap, infeas;
    double seconds, pg;
} oracle_stats_t;

static double dotn(int n, const double *a, const double *b) { double s = 0; for (int i = 0; i < n; ++i) s += a[i] * b[i]; return s; }

static void set_bounds(oracle_t *o)
{
    for (int r = 0; r < o->ng; ++r) { o->glo[r] = -INFINITY; o->ghi[r] = INFINITY; }
    for (int j = 0; j < o->n; ++j) {
        int r = o->grp[j];
        double lo = -INFINITY, hi = INFINITY;
        if (o->ctype[j] == 0) lo = (o->c[j] > 0) ? log(o->c[j]) : -INFINITY;
        else if (o->ctype[j] == 2) lo = hi = log(o->c[j]);
        lo -= o->off[j]; hi -= o->off[j];
        if (lo > o->glo[r]) o->glo[r] = lo;
        if (hi < o->ghi[r]) o->ghi[r] = hi;
    }
}

/* start: nu0 given per token; group variable = mean of (log nu0 - off) over members, clamped */
void oracle_start(oracle_t *o, const double *nu0, int memory)
{
    const int n = o->n, ng = o->ng;
    set_bounds(o);
    if (memory > 0 && memory != o->M) {
        o->M = memory;
        o->S = realloc(o->S, (size_t)o->M * n * 8); o->Y = realloc(o->Y, (size_t)o->M * n * 8);
        o->rho = realloc(o->rho, o->M * 8); o->alpha = realloc(o->alpha, o->M * 8);
    }
    int *cnt = calloc(ng, sizeof(int));
    for (int r = 0; r < ng; ++r) o->s_t[r] = 0.0;
    for (int j = 0; j < n; ++j) { o->s_t[o->grp[j]] += log(nu0[j]) - o->off[j]; cnt[o->grp[j]]++; }
    for (int r = 0; r < ng; ++r) {
        o->s_t[r] /= (cnt[r] ? cnt[r] : 1);
        if (o->s_t[r] < o->glo[r]) o->s_t[r] = o->glo[r];
        if (o->s_t[r] > o->ghi[r]) o->s_t[r] = o->ghi[r];
    }
    free(cnt);
    for (int j = 0; j < n; ++j) o->nu[j] = exp(o->s_t[o->grp[j]] + o->off[j]);
    o->hist = 0; o->head = 0; o->iter = 0; o->evals = 0; o->status = 0; o->first = 1; o->t_step = 1.0;
}

/* consume (f_pools, psi) evaluated at o->nu; diag is read on the first call only */
int oracle_step(oracle_t *o, double f_pools, const double *psi, const double *diag, const oracle_opts_t *opt)
{
    const int n = o->n, ng = o->ng, M = o->M;
    double f_t = f_pools, gapv = 0.0, viol = 0.0, scale = 0.0;
    double *ucurv = calloc(n, 8);              /* the utility table's share of the diagonal metric (first call) */
    for (int r = 0; r < ng; ++r) o->Gs_t[r] = 0.0;
    for (int j = 0; j < n; ++j) {
        double rj = psi[j] + o->h[j], v, a;
        if (o->ctype[j] >= 3) {
            /* the utility table (separable concave utilities beyond the reference's linear-plus-box: not in the reference, whose
             * objectives are linear, arbitrage.py:78): conjugate ubar(nu) = sup_P (u(P) - nu P), its maximiser P*, and the
             * Fenchel-Young gap term ubar + nu psi - u(psi).  ctype 3: u = c log(P + h); 4: u = c P - P^2 / (2 h). */
            const double c = o->c[j], h = o->h[j], nu = o->nu[j], ps = psi[j];
            double pstar, ubar, uval, curv;
            v = 0.0;
            if (o->ctype[j] == 3) {
                pstar = c / nu - h; ubar = c * log(c / nu) - c + nu * h;
                v = fmax(-(ps + h), 0.0); uval = c * log(fmax(ps + h, 1e-300)); curv = nu * h;
            } else {
                pstar = h * (c - nu); ubar = 0.5 * h * (c - nu) * (c - nu);
                uval = c * ps - 0.5 * ps * ps / h; curv = h * nu * (2.0 * nu - c);
            }
            rj = ps - pstar;
            f_t += ubar;
            gapv += ubar + nu * ps - uval;
            a = fmax(fabs(ps), fabs(pstar));
            if (o->first) ucurv[j] = fmax(curv, 0.0);
        } else {
            f_t += (o->nu[j] - o->c[j]) * o->h[j];
            gapv += (o->nu[j] - o->c[j]) * rj;
            v = (o->ctype[j] == 0) ? fmax(-rj, 0.0) : (o->ctype[j] == 1 ? fabs(rj) : 0.0);
            a = fmax(fabs(psi[j]), fabs(o->h[j]));
        }
        o->Gs_t[o->grp[j]] += o->nu[j] * rj;
        if (v > viol) viol = v;
        if (a > scale) scale = a;
    }
    o->evals++;
    if (o->first) {
        for (int r = 0; r < ng; ++r) o->Ds[r] = 0.0;
        for (int j = 0; j < n; ++j) o->Ds[o->grp[j]] += diag[j] + ucurv[j];
    }
    free(ucurv);
    int accept = o->first;
    if (!o->first) {
        /* Armijo on f; once the decrease is below the rounding noise of f (a sum over all pools),
         * fall back to the derivative form (Hager-Zhang "approximate Wolfe"): the slope along the
         * step at the trial point must not have turned strongly positive. */
        double dec = 0.0, dec_t = 0.0;
        for (int r = 0; r < ng; ++r) { double ds = o->s_t[r] - o->s[r]; dec += o->Gs[r] * ds; dec_t += o->Gs_t[r] * ds; }
        accept = (f_t == f_t) && ((f_t <= o->f + opt->armijo * dec) ||
                                  (f_t <= o->f + 1e-11 * fmax(1.0, fabs(o->f)) && dec_t <= 0.8 * fabs(dec)));
    }
    if (!accept) {
        o->t_step *= 0.5;
        if (o->t_step < 1e-9) { o->status = 2; return o->status; }
    } else {
        if (!o->first) {
            double *sv = o->S + (size_t)o->head * n, *yv = o->Y + (size_t)o->head * n;
            double sy = 0, ss = 0, yy = 0;
            for (int r = 0; r < ng; ++r) { sv[r] = o->s_t[r] - o->s[r]; yv[r] = o->Gs_t[r] - o->Gs[r]; sy += sv[r] * yv[r]; ss += sv[r] * sv[r]; yy += yv[r] * yv[r]; }
            if (sy > 1e-12 * sqrt(ss) * sqrt(yy)) { o->rho[o->head] = 1.0 / sy; o->head = (o->head + 1) % M; if (o->hist < M) o->hist++; }
            o->iter++;
        }
        for (int r = 0; r < ng; ++r) { o->s[r] = o->s_t[r]; o->Gs[r] = o->Gs_t[r]; }
        memcpy(o->psi, psi, 8 * n);
        o->f = f_t; o->first = 0;
        o->gap = fabs(gapv) / fmax(1.0, fabs(f_t));
        o->primal = f_t - gapv;
        o->infeas = viol / fmax(scale, 1e-300);
        {   /* value of the projected reduced gradient: sum_r |P(Gs)_r| / max(1,|f|) (>= gap) */
            double pg = 0.0;
            for (int r = 0; r < ng; ++r) {
                double G = o->Gs[r], v = G;
                if (o->glo[r] == o->ghi[r]) v = 0.0;
                else if (o->s[r] <= o->glo[r] + 1e-14) v = fmin(G, 0.0);
                else if (o->s[r] >= o->ghi[r] - 1e-14) v = fmax(G, 0.0);
                pg += fabs(v);
            }
            o->pg = pg / fmax(1.0, fabs(f_t));
        }
        if (opt->pg_rule ? (o->pg <= opt->tol_gap) : (o->gap <= opt->tol_gap && o->infeas <= opt->tol_infeas)) { o->status = 1; return o->status; }
        /* new direction */
        double *q = o->q, *rr = o->r;
        for (int r = 0; r < ng; ++r) {
            int act = (o->s[r] <= o->glo[r] + 1e-14 && o->Gs[r] > 0.0) || (o->s[r] >= o->ghi[r] - 1e-14 && o->Gs[r] < 0.0) || (o->glo[r] == o->ghi[r]);
            q[r] = act ? 0.0 : o->Gs[r];
        }
        double gp_sq = 0.0; for (int r = 0; r < ng; ++r) gp_sq += q[r] * q[r];
        for (int k = 0; k < o->hist; ++k) {
            int i = (o->head - 1 - k + M) % M;
            o->alpha[i] = o->rho[i] * dotn(ng, o->S + (size_t)i * n, q);
            const double *yv = o->Y + (size_t)i * n;
            for (int r = 0; r < ng; ++r) q[r] -= o->alpha[i] * yv[r];
        }
        for (int r = 0; r < ng; ++r) { double H = o->Ds[r] + fmax(o->Gs[r], 0.0); rr[r] = H > 0.0 ? q[r] / H : 0.0; }
        for (int k = o->hist - 1; k >= 0; --k) {
            int i = (o->head - 1 - k + M) % M;
            double beta = o->rho[i] * dotn(ng, o->Y + (size_t)i * n, rr);
            const double *sv = o->S + (size_t)i * n;
            for (int r = 0; r < ng; ++r) rr[r] += sv[r] * (o->alpha[i] - beta);
        }
        double dg = 0.0, dmax = 0.0;
        for (int r = 0; r < ng; ++r) {
            int act = (o->s[r] <= o->glo[r] + 1e-14 && o->Gs[r] > 0.0) || (o->s[r] >= o->ghi[r] - 1e-14 && o->Gs[r] < 0.0) || (o->glo[r] == o->ghi[r]);
            o->d[r] = act ? 0.0 : -rr[r];
            dg += o->d[r] * o->Gs[r];
            if (fabs(o->d[r]) > dmax) dmax = fabs(o->d[r]);
        }
        if (!(dg < 0.0) && gp_sq > 0.0) {          /* not a descent direction: restart from the metric */
            o->hist = 0; dmax = 0.0;
            for (int r = 0; r < ng; ++r) {
                int act = (o->s[r] <= o->glo[r] + 1e-14 && o->Gs[r] > 0.0) || (o->s[r] >= o->ghi[r] - 1e-14 && o->Gs[r] < 0.0) || (o->glo[r] == o->ghi[r]);
                double H = o->Ds[r] + fmax(o->Gs[r], 0.0);
                o->d[r] = (act || !(H > 0.0)) ? 0.0 : -o->Gs[r] / H;
                if (fabs(o->d[r]) > dmax) dmax = fabs(o->d[r]);
            }
        }
        o->t_step = (dmax > opt->max_step) ? opt->max_step / dmax : 1.0;
    }
    for (int r = 0; r < ng; ++r) {
        double v = o->s[r] + o->t_step * o->d[r];
        if (v < o->glo[r]) v = o->glo[r];
        if (v > o->ghi[r]) v = o->ghi[r];
        o->s_t[r] = v;
    }
    for (int j = 0; j < n; ++j) o->nu[j] = exp(o->s_t[o->grp[j]] + o->off[j]);
    if (o->evals >= opt->max_evals) o->status = 3;
    return o->status;
}

static double now_s(void)
{
#ifdef _OPENMP
    return omp_get_wtime();
#else
    return (double)clock() / CLOCKS_PER_SEC;
#endif
}

/* read-out for callers that drive oracle_start / oracle_eval / oracle_step themselves (the
 * pool-sharded loop of tests/test_distributed.py): trial prices, accepted prices, statistics */
/* psi at the accepted point (after the all-reduce when the caller runs the pool-sharded loop) */
void oracle_get_psi(oracle_t *o, double *psi) { memcpy(psi, o->psi, 8 * o->n); }

void oracle_get(oracle_t *o, double *nu_trial, double *nu_acc, oracle_stats_t *st)
{
    const int n = o->n;
    if (nu_trial) memcpy(nu_trial, o->nu, 8 * n);
    if (nu_acc) for (int j = 0; j < n; ++j) nu_acc[j] = exp(o->s[o->grp[j]] + o->off[j]);
    if (st) {
        st->evals = o->evals; st->iters = o->iter; st->status = o->status;
        st->dual_value = o->f; st->gap = o->gap; st->infeas = o->infeas; st->pg = o->pg; st->seconds = 0.0;
        double pv = 0.0;
        for (int j = 0; j < n; ++j) pv += o->c[j] * o->psi[j];
        st->primal_value = o->general ? o->primal : pv;      /* (table utilities: U(psi) = g - the Fenchel-Young gap terms) */
    }
}

/* prob.solve() (arbitrage.py:82).  nu0 = start prices; on return nu = accepted prices. */
int oracle_solve(oracle_t *o, const double *nu0, const oracle_opts_t *opt, oracle_stats_t *st, double *nu_out, double *psi_out)
{
    const int n = o->n;
    double *psi = calloc(n, 8);
    double t0 = now_s();
    oracle_start(o, nu0, opt->memory);
    while (o->status == 0) {
        double f = oracle_eval(o, o->nu, psi, o->first ? o->diag : NULL);
        oracle_step(o, f, psi, o->diag, opt);
    }
    st->seconds = now_s() - t0;
    st->evals = o->evals; st->iters = o->iter; st->status = o->status;
    st->dual_value = o->f; st->gap = o->gap; st->infeas = o->infeas; st->pg = o->pg;
    double pv = 0.0;
    for (int j = 0; j < n; ++j) { nu_out[j] = exp(o->s[o->grp[j]] + o->off[j]); psi_out[j] = o->psi[j]; pv += o->c[j] * o->psi[j]; }
    st->primal_value = o->general ? o->primal : pv;
    free(psi);
    return o->status;
}
