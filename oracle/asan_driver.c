/* TEST INFRASTRUCTURE: the C oracle under AddressSanitizer + UndefinedBehaviorSanitizer ("ASAN on the CPU twin").
 * A seeded random network with every pool kind (constant product, weighted, constant sum with tied pools, stableswap, power sum,
 * k-asset geo-mean 3..8) is evaluated, its tenders materialised and solved to its certificates for the three utilities of
 * arbitrage.py:57,77 / liquidation.py:57,77-80 / two-asset.py:66,86, single- and multi-threaded.  Built and run by
 * tests/test_oracle.py::test_c_oracle_is_clean_under_asan_and_ubsan (gcc -fsanitize=address,undefined): exit code 0 and the line
 * "asan driver ok" mean no report.  */
#include <stdio.h>
#include "cfmm_oracle.c"

static unsigned long long rng_state = 88172645463325252ull;
static double urand(void) { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return (double)(rng_state >> 11) * (1.0 / 9007199254740992.0); }
static double nrand(void) { double u = urand(), v = urand(); if (u < 1e-300) u = 1e-300; return sqrt(-2.0 * log(u)) * cos(6.283185307179586 * v); }

int main(void)
{
    enum { N = 24, M2 = 300, MN = 40 };
    double price[N];
    for (int j = 0; j < N; ++j) price[j] = exp(0.5 * nrand());
    int fails = 0;
    for (int variant = 0; variant < 4; ++variant) {             /* threads 1 | 4  x  with | without the constant-sum bucket */
        const int threads = (variant & 1) ? 4 : 1, with_sum = variant < 2;
        oracle_t *o = oracle_create(N);
        oracle_set_threads(o, threads);
        /* five two-asset buckets (kinds 0..4) */
        double *Ra[5], *Rb[5], *fee[5], *prm[5]; int32_t *ia[5], *ib[5], *tied = calloc(M2, 4);
        for (int k = 0; k < 5; ++k) {
            Ra[k] = malloc(8 * M2); Rb[k] = malloc(8 * M2); fee[k] = malloc(8 * M2); prm[k] = malloc(8 * M2);
            ia[k] = malloc(4 * M2); ib[k] = malloc(4 * M2);
            for (int i = 0; i < M2; ++i) {
                int a = (int)(urand() * N) % N, b = (a + 1 + (int)(urand() * (N - 1)) % (N - 1)) % N;
                if (k == 2 || k == 3) b = (a / 4) * 4 + (a % 4 + 1 + (int)(urand() * 3) % 3) % 4;      /* near-peg partners */
                const double L = exp(3.0 + nrand());
                double pa = price[a], pb = price[b];
                if (k == 2 || k == 3) pb = pa * exp(0.002 * nrand());
                ia[k][i] = a; ib[k][i] = b; fee[k][i] = 0.997 + 0.002 * urand();
                Ra[k][i] = L / pa * exp(0.02 * nrand()); Rb[k][i] = L / pb;
                prm[k][i] = 0.0;
                if (k == 1) { const double w = 0.2 + 0.6 * urand(); prm[k][i] = w; Rb[k][i] = L * (1.0 - w) / w / pb; }
                if (k == 3) prm[k][i] = Ra[k][i] * Ra[k][i] * Rb[k][i] * (0.1 + urand());
                if (k == 4) { const double t = 0.2 + 0.6 * urand(); prm[k][i] = t; Rb[k][i] = Ra[k][i] * pow(pa / pb, 1.0 / t); }
            }
            if (k == 2) for (int i = 0; i < M2; i += 7) tied[i] = 1;
            if (k == 2 && !with_sum) continue;      /* (first order alone does not settle on constant-sum kinks: the host's active-set loop does) */
            if (oracle_add_pools2(o, k, M2, Ra[k], Rb[k], fee[k], (k == 1 || k >= 3) ? prm[k] : NULL, ia[k], ib[k], k == 2 ? tied : NULL)) ++fails;
        }
        /* k-asset geo-mean buckets, slot-major [k][m] */
        int32_t *idx[9]; double *R[9], *w[9], *fn[9];
        for (int k = 3; k <= 8; ++k) {
            idx[k] = malloc(4 * k * MN); R[k] = malloc(8 * k * MN); w[k] = malloc(8 * k * MN); fn[k] = malloc(8 * MN);
            for (int i = 0; i < MN; ++i) {
                int start = (int)(urand() * N) % N; double ws = 0.0;
                for (int j = 0; j < k; ++j) { idx[k][j * MN + i] = (start + 2 * j + (j > 3)) % N; w[k][j * MN + i] = 1.0 + (int)(urand() * 4); ws += w[k][j * MN + i]; }
                for (int j = 0; j < k; ++j) {                       /* distinct tokens: repair collisions */
                    for (int q = 0; q < j; ++q) if (idx[k][q * MN + i] == idx[k][j * MN + i]) idx[k][j * MN + i] = (idx[k][j * MN + i] + 1) % N, q = -1;
                }
                const double L = exp(3.0 + nrand());
                for (int j = 0; j < k; ++j) { w[k][j * MN + i] /= ws; R[k][j * MN + i] = L * w[k][j * MN + i] / price[idx[k][j * MN + i]] * exp(0.02 * nrand()); }
                fn[k][i] = 0.997;
            }
            if (oracle_add_poolsN(o, k, MN, idx[k], R[k], w[k], fn[k])) ++fails;
        }
        double c[N], h[N], nu0[N], nu[N], psi[N], diag[N]; int32_t ct[N];
        for (int util = 0; util < 3; ++util) {
            for (int j = 0; j < N; ++j) { c[j] = util == 0 ? price[j] : 0.0; h[j] = 0.0; ct[j] = util == 1 ? 1 : 0; nu0[j] = price[j]; }
            if (util) { c[5] = 1.0; h[2] = 3.0 / price[2]; h[9] = 2.0 / price[9]; if (util == 1) ct[5] = 2; for (int j = 0; j < N; ++j) nu0[j] = price[j] / price[5]; }
            oracle_set_utility(o, c, h, ct);
            const double f = oracle_eval(o, nu0, psi, diag);
            if (!(f == f)) ++fails;
            for (int b = 0; b < 5; ++b) { double *ya = malloc(8 * M2), *yb = malloc(8 * M2); oracle_trades2(o, b, nu0, ya, yb); free(ya); free(yb); }
            oracle_opts_t opt = { 1e-7, 1e-7, 1e-4, 2.0, 8000, util == 2 ? 3 : 8, 0, 0 };
            oracle_stats_t st;
            oracle_solve(o, nu0, &opt, &st, nu, psi);
            printf("threads %d constant-sum %d utility %d: status %d evals %d gap %.2e infeas %.2e value %.10g\n", threads, with_sum, util, st.status, st.evals, st.gap, st.infeas, st.primal_value);
            if (st.status != 1 && st.status != 2 && st.status != 3) ++fails;
            if (!with_sum && st.status != 1) ++fails;          /* smooth pools only: the certificates must be reached */
        }
        oracle_destroy(o);
        for (int k = 0; k < 5; ++k) { free(Ra[k]); free(Rb[k]); free(fee[k]); free(prm[k]); free(ia[k]); free(ib[k]); }
        for (int k = 3; k <= 8; ++k) { free(idx[k]); free(R[k]); free(w[k]); free(fn[k]); }
        free(tied);
    }
    if (fails) { printf("asan driver: %d failures\n", fails); return 1; }
    printf("asan driver ok\n");
    return 0;
}
