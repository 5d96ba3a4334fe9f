/* cfmm.h -- C-ABI of libcfmm_hip.so: the MI355X-native replacement for the cvxpy call pair
 *
 *        prob = cp.Problem(obj, cons); prob.solve()
 *
 * at /root/reference/arbitrage.py:81-82, liquidation.py:84-85 and two-asset.py:90-91, plus the
 * result read-back at arbitrage.py:84, liquidation.py:87, two-asset.py:94-100.
 *
 * The reference has no FFI / plugin interface of its own (it is four Python scripts that call
 * cvxpy); this header is the boundary a maintainer binds with ctypes (INTEGRATION.md shows the
 * stub).  Conventions: every entry point returns 0 on success or a negative CFMM_E* code;
 * cfmm_last_error() gives the message; no exceptions, no callbacks; the caller owns every host
 * buffer (the library copies to HBM at upload and back at get); one ctx = one GPU = one host
 * thread at a time.  All floating point is fp64 (NumPy's default in arbitrage.py:14-20), all
 * indices int32.
 *
 * Model (the reference's, arbitrage.py:51-78):
 *      maximise   U(psi)
 *      subject to psi = sum_i A_i (Lambda_i - Delta_i)                      arbitrage.py:54
 *                 phi_i(R_i + gamma_i Delta_i - Lambda_i) >= phi_i(R_i)      arbitrage.py:60,63-74
 *                 Delta_i, Lambda_i >= 0                                     arbitrage.py:51-52
 * solved in its dual-decomposition form: for prices nu every pool is an independent
 * closed-form / Newton subproblem (the HIP kernels), psi(nu) is their scatter-sum, and nu is
 * driven by an on-device projected quasi-Newton iteration in log-prices.
 */
#ifndef CFMM_H
#define CFMM_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct cfmm_ctx cfmm_ctx;

enum {
    CFMM_OK = 0,
    CFMM_E_ARG = -1,        /* bad argument                                   */
    CFMM_E_HIP = -2,        /* a HIP runtime call failed                      */
    CFMM_E_STATE = -3,      /* call order (e.g. solve before upload)          */
    CFMM_E_LIMIT = -4,      /* size beyond what this build supports           */
    CFMM_E_RCCL = -5,       /* an RCCL call failed                            */
    CFMM_E_NUMERIC = -6,    /* non-finite value met during the iteration      */
    CFMM_E_UNSUPPORTED = -7 /* the requested method cannot take this problem  */
};

#define CFMM_AUTO_NEWTON_MIN_STABLE 4096
/* outer iteration of cfmm_solve */
enum {
    CFMM_METHOD_AUTO = 0,       /* second order when the network holds >= CFMM_AUTO_NEWTON_MIN_STABLE stableswap pools
                                   (the near-linear case the first-order iteration crawls on at scale), else first
                                   order; a first-order run that ends without its certificates is handed on         */
    CFMM_METHOD_LBFGS = 1,      /* projected L-BFGS in log-prices, fully on-device: ONE launch per outer iteration
                                   (iter_kernel, enqueued eagerly with run-ahead on a pinned progress word; no hipGraph
                                   on this default path -- graph replay serves the two-launch iteration only: price
                                   ties, > 2048 tokens, memory > 3)                                                 */
    CFMM_METHOD_NEWTON = 2      /* barrier-smoothed dual Newton: dense n x n Hessian, blocked Cholesky on-device    */
};

/* two-asset pool families: one SoA bucket each (`param` = per-pool 4th column) */
enum {
    CFMM_POOL_CP2 = 0,      /* constant product sqrt(xy)          arbitrage.py:68-70; param = NULL  */
    CFMM_POOL_W2 = 1,       /* weighted geo-mean x^wa y^(1-wa)    arbitrage.py:65 (2 assets); param = wa */
    CFMM_POOL_SUM2 = 2,     /* constant sum x+y, x,y >= 0         arbitrage.py:73-74; param = NULL  */
    CFMM_POOL_CURVE2 = 3,   /* x + y - alpha/(xy) (StableSwap at fixed D)  not in reference; param = alpha */
    CFMM_POOL_POW2 = 4,     /* power sum x^(1-t) + y^(1-t) (YieldSpace's curve; t -> 0: constant sum, t -> 1: towards constant
                               product)  not in reference; param = t in [0.001, 0.999].  The first tenant of the GENERIC
                               bucket: it has no code of its own beyond one table entry (csrc/phi2.hpp: Phi2<4>) -- the exact
                               pool solution, the diagonal metric, the smoothed solution and the tenders all come from the
                               generic root search on that entry's forward exchange function, on both outer iterations   */
    CFMM_POOL_KINDS2 = 5
};
#define CFMM_MAX_POOL_SIZE 8    /* n-asset geo-mean pools: 3..8 assets, one bucket per size */

/* K-asset trading functions OTHER than the weighted geometric mean: the K-asset table (csrc/phik.hpp: one PhiK<KIND> struct per
 * function -- SURVEY 8(f) rank 4; "a pool is whatever constraint line is written", arbitrage.py:63-74), 3..8 assets, one
 * bucket per (kind, size).  Evaluated as wave-tiles (leg per lane, LDS psi tile) in one launch behind the main evaluation; the
 * stableswap entry enters the second-order path with its exact generalised Hessian block, the constant-sum entry with the path's own
 * log barrier on its sign constraints (csrc/phik.hpp: sum_smooth_k -- one scalar root per pool, strictly feasible tenders for every weight,
 * closed-form Hessian). */
enum {
    CFMM_POOLK_STABLE = 0,  /* n-asset stableswap  sum x - alpha / prod x  (the paper's concave form; 2 assets: CFMM_POOL_CURVE2);
                               param = alpha.  Solved by the table's generic two-level search (no closed form)              */
    CFMM_POOLK_SUM = 1,     /* n-asset constant sum  sum x, x >= 0  (arbitrage.py:73-74 with more than two tokens); param = NULL.
                               First order: the exact LP vertex on the device; an optimum ON one of its kinks (a leg drained partly, two
                               tokens tied for cheapest) needs the caller's active-set loop (cfmm_set_ties + cfmm_set_pool_flagsG:
                               cfmm/problem.py) -- or CFMM_METHOD_NEWTON, which needs none (the barrier-smoothed pool fills partly by itself) */
    CFMM_POOLK_KINDS = 2
};

/* token constraint types of the unified utility  max c'psi :  psi_k + h_k (>=, =, free) 0 */
enum { CFMM_GE = 0, CFMM_EQ = 1, CFMM_FREE = 2,
       /* the utility table: separable concave utilities beyond the reference's linear-plus-box (not in the
        * reference, whose objectives are linear: arbitrage.py:78, liquidation.py:80, two-asset.py:87).  The token's c and h carry
        * the entry's two parameters; its price has no bound.  Both outer iterations take them (the first-order one in its generic
        * two-launch form); no price ties, no batching.
        *   CFMM_ULOG   u(Psi) = c log(Psi + h),        c > 0, h >= 0
        *   CFMM_UQUAD  u(Psi) = c Psi - Psi^2 / (2 h),  c >= 0, h > 0   (cfmm_set_utility requires c >= 0 of every entry kind) */
       CFMM_ULOG = 3, CFMM_UQUAD = 4 };

typedef struct {
    double tol_gap;         /* stop when |(nu-c)'(psi+h)| / max(1,|g|)      <= tol_gap     (1e-6) */
    double tol_infeas;      /*      and  max violation / max(|psi|,|h|)     <= tol_infeas  (1e-6) */
    double armijo;          /* sufficient-decrease constant (1e-4)                                 */
    double max_step;        /* cap on one step in log-price (2.0)                                  */
    int32_t max_evals;      /* cap on dual evaluations (2000); a hand-over to the second-order method starts a fresh count */
    int32_t memory;         /* L-BFGS pairs kept, 1..8; 0 = auto (8 up to 32 tokens, else 3: no more evaluations than 4..6 on average and the cheapest update)                                        */
    int32_t iters_per_graph;/* two-launch iteration only (price ties, > 2048 tokens, memory > 3): outer iterations
                               captured per hipGraph replay (4) -- also the chunk length of the pool-sharded RCCL path; the
                               default one-launch-per-iteration path (run-ahead on a progress word) ignores it */
    int32_t pg_rule;        /* 1: stop on the projected-gradient value <= tol_gap instead (used when
                               constant-sum pools are tied: psi then lacks their fill)           */
    int32_t method;         /* CFMM_METHOD_*  (0 = auto)                                                            */
    int32_t max_newton;     /* cap on second-order steps (200)                                                      */
    double barrier_shrink;  /* factor applied to the barrier weight once a step lands near the central path (0.1)   */
} cfmm_opts;

typedef struct {
    int32_t evals;          /* dual evaluations done = passes of every pool through its kernel     */
    int32_t iters;          /* accepted quasi-Newton steps                                         */
    int32_t status;         /* 1 converged, 2 stalled (line search), 3 max_evals, <0 error         */
    int32_t n_ranks;
    double dual_value;      /* g(nu)            upper bound                                        */
    double primal_value;    /* c'psi(nu)        the reference's prob.value (arbitrage.py:84)       */
    double gap, infeas;     /* the two certificates above                                          */
    double wall_seconds;    /* host clock around the outer loop (upload / read-back excluded)      */
    double device_seconds;  /* HIP events around the same region                                   */
    double pg;              /* sum |projected reduced gradient| / max(1,|g|)                       */
    int64_t pool_subproblems;   /* evals * pools on this rank                                      */
    double barrier_mu;      /* final barrier weight of a second-order solve (0 after a first-order one): psi and the
                               tenders read back are then those of the smoothed, strictly feasible primal point     */
    int32_t newton_steps;   /* second-order steps taken (Hessian assemblies + Cholesky factorisations)             */
    int32_t method;         /* CFMM_METHOD_LBFGS or CFMM_METHOD_NEWTON: what produced the result                    */
} cfmm_stats;

/* lifetime ------------------------------------------------------------------------------ */
int cfmm_create(int device, int n_tokens, cfmm_ctx **out);
int cfmm_destroy(cfmm_ctx *ctx);
/* A second context on the same GPU that SHARES the pool columns already resident in HBM (reference-counted,
 * no copy) but has its own stream, utility, prices and solver state: several solves over one pool set --
 * the parameter sweep of two-asset.py:34-100, or independent utilities -- can then be in flight at once
 * (one per host thread), one solve's single-workgroup nu update overlapping another's evaluation kernels.
 * Pools cannot be re-uploaded while clones exist. */
int cfmm_clone(cfmm_ctx *src, cfmm_ctx **out);
const char *cfmm_last_error(cfmm_ctx *ctx);       /* ctx may be NULL: last error of cfmm_create */
const char *cfmm_backend(cfmm_ctx *ctx);          /* "hip:gfx950"                               */
void cfmm_default_opts(cfmm_opts *o);

/* problem object: replaces local_indices / reserves / fees (arbitrage.py:6-28) and the dense
 * A_i matrices (arbitrage.py:42-48: never materialised -- `ia/ib/idx` ARE A_i) ------------- */
int cfmm_upload_pools2(cfmm_ctx *ctx, int kind, int64_t m, const double *Ra, const double *Rb,
                       const double *fee, const double *param, const int32_t *ia, const int32_t *ib);
/* k-asset weighted geo-mean pools (arbitrage.py:65), slot-major: x[j*m + i] = slot j of pool i;
 * weights normalised to sum 1 per pool */
int cfmm_upload_poolsN(cfmm_ctx *ctx, int k, int64_t m, const int32_t *idx, const double *R,
                       const double *w, const double *fee);
/* pools of the K-asset table (CFMM_POOLK_*), k = 3..8 assets (2 is accepted too: the cross-check of the table's generic search
 * against the two-asset buckets' closed forms, tests/test_gpu_table.py): idx, R slot-major [k][m] like cfmm_upload_poolsN, fee[m],
 * param[m] (alpha for CFMM_POOLK_STABLE, NULL for CFMM_POOLK_SUM).  Replaces a constraint line such as
 * `cp.sum(new_reserves) - alpha * cp.inv_prod(new_reserves) >= ...` / `cp.sum(new_reserves) >= cp.sum(reserves)` over k > 2
 * tokens (arbitrage.py:63-74 in the reference's style; the reference itself ships the two-token forms) */
int cfmm_upload_poolsG(cfmm_ctx *ctx, int kind, int k, int64_t m, const int32_t *idx, const double *R,
                       const double *fee, const double *param);
/* constant-sum pools sitting on their kink are `tied` (flag 1): they are skipped by the
 * kernels and their fill fraction is assigned by the host's primal recovery */
int cfmm_set_pool_flags(cfmm_ctx *ctx, int kind, const int32_t *flags /* [m] or NULL */);
/* the same for the K-asset table's constant-sum bucket of k tokens, per LEG (slot-major [k][m] like the bucket's columns, or NULL):
 * a flagged leg j of a pool sits on the kink gamma nu_j = nu_cheapest (arbitrage.py:73-74 over more than two tokens: that token is
 * partially drained at the optimum); it is left out of the evaluation and the tenders, the caller ties the two prices
 * (cfmm_set_ties) and adds the partial fill theta R_j itself */
int cfmm_set_pool_flagsG(cfmm_ctx *ctx, int k, const int32_t *flags);

/* utility: replaces obj + the psi constraints (arbitrage.py:57,77; liquidation.py:57,77-80;
 * two-asset.py:66,86).  ctype NULL = all CFMM_GE, h NULL = 0. */
int cfmm_set_utility(cfmm_ctx *ctx, const double *c, const double *h, const int32_t *ctype);
/* price ties: log nu_j = s[grp[j]] + off[j]; NULL, NULL = none */
int cfmm_set_ties(cfmm_ctx *ctx, int n_groups, const int32_t *grp, const double *off);

/* one dual evaluation = every pool's subproblem once + the reduction:  psi(nu), sum_i arb_i,
 * optionally the diagonal metric.  This is the unit BASELINE.json's metric counts. */
int cfmm_eval_dual(cfmm_ctx *ctx, const double *nu, double *arb_sum, double *psi, double *diag /* or NULL */);

/* the barrier-smoothed evaluation behind CFMM_METHOD_NEWTON: every direction of every two-asset pool solves
 *      max_{D > 0}  nu_out L(D) - nu_in D + mu log D        (constant sum: + mu log(R_out/gamma - D))
 * (k-asset geo-mean pools enter unsmoothed, with their exact solution and generalised Hessian);
 * value = sum of those optima, trade = sum nu'(L - D), psi[n] = sum A_i (L - D), and -- if H is not NULL -- the
 * n x n Hessian of `value` in log-prices minus its diag(nu * psi) term, column-major, LOWER triangle only. */
int cfmm_eval_smooth(cfmm_ctx *ctx, const double *nu, double mu, double *value, double *trade, double *psi, double *H);

/* test hook: the dense Cholesky solve of the second-order method on a caller-supplied SPD system (A: n x n
 * column-major, lower triangle read; n must equal the context's token count); *info != 0 flags a non-positive pivot */
int cfmm_debug_cholesky(cfmm_ctx *ctx, int n, const double *A, const double *b, double *x, int32_t *info);
/* test hook: x = A^-1 b for a NEW right-hand side through the factor and inverse factor the last cfmm_debug_cholesky left
 * (the two matrix-vector products of the second-order iteration's chord steps) */
int cfmm_debug_cholesky_apply(cfmm_ctx *ctx, int n, const double *b, double *x);

/* Reproducible mode.  By default psi is scatter-added with fp64 atomics, whose order -- lanes, waves, workgroups -- varies
 * from run to run: psi, and with it the path of a solve, is reproducible to rounding only.  With `on` != 0 every pool's
 * contribution is converted exactly to a 96-bit fixed-point integer and accumulated with integer atomics (associative:
 * any order gives the same bits), all-reduced as integers when pool-sharded, and converted back in a fixed order:
 * psi, the iterates and the evaluation count are then BITWISE identical from run to run and for any number of pool
 * shards / GPUs.  Cost, measured (DESIGN.md "Reproducible mode"): the evaluation kernel +20..30 % (three LDS atomics per leg
 * instead of one: 20.1 against 16.6 us at 1e6 mixed pools), the outer iteration +24..30 % (one small extra launch per
 * evaluation).  First-order path and cfmm_eval_dual; needs <= ~2600 tokens; fees >= 1e-3.  Also: CFMM_DETERMINISTIC=1. */
int cfmm_set_deterministic(cfmm_ctx *ctx, int on);
/* test hook of the reproducible mode: one dual evaluation returning the RAW integer limbs of psi ([3][n], limb-major, value =
 * (limb2 2^64 + limb1 2^32 + limb0) / 2^F, every limb a wrapped signed 64-bit sum) with the fixed-point exponent F derived from
 * (ref_reserve, ref_fee) instead of this context's own largest reserve / smallest fee -- so that the limbs of the S shards
 * of a network, added as integers on the host, can be compared bit for bit with the unsharded network's. */
int cfmm_debug_eval_limbs(cfmm_ctx *ctx, const double *nu, double ref_reserve, double ref_fee, uint64_t *limbs);

/* prob.solve(): nu0 = start prices.  NULL: continue from cfmm_set_nu / the previous solution -- after a second-order
 * solve that includes (a multiple of) its final barrier weight: the warm start of a parametric sweep (two-asset.py:34-100).
 * Constant-sum pools (arbitrage.py:12,20,28,72-74) can end PARTIALLY filled -- a kink of the dual that prices alone do not resolve.
 * With CFMM_METHOD_AUTO on a network of the reference's size (what cfmm_solve_sweep serves; linear-box utility, one GPU, no ties set by
 * the caller) the library runs the active-set loop over such kinks itself (round 6): stats.primal_value, cfmm_get_psi / _solution and
 * cfmm_get_trades2 return the point WITH the fills (arbitrage.py's pool 4: 38.6 %), stats.method = CFMM_METHOD_LBFGS.  Elsewhere AUTO
 * falls back on the second-order method, which needs no active set; an explicit CFMM_METHOD_LBFGS leaves the kinks to the caller
 * (cfmm_set_ties, cfmm_set_pool_flags: what cfmm/problem.py does for networks of any size). */
int cfmm_solve(cfmm_ctx *ctx, const double *nu0, const cfmm_opts *opts, cfmm_stats *out);

/* `nb` prob.solve() calls over the SAME pools in lock-step: the loop body of the parametric sweep, two-asset.py:34-100
 * (the pools, reserves and fees of lines 7-32 stay, only the utility of line 66 / 86 changes with t), or independent
 * baskets over one pool set.  ctxs[0] is the context the pools were uploaded to, ctxs[1..] its cfmm_clone()s, each with
 * its own cfmm_set_utility; nu0[b] (or nu0 itself) may be NULL = continue from that context's prices.  Every outer
 * iteration reads every pool column ONCE and solves each pool at all nb price vectors (first-order method; no price
 * ties, no stableswap pools, not pool-sharded); out[b] are the statistics of solve b (wall / device seconds: of the
 * whole batch).  nb <= cfmm_batch_capacity(n_tokens) (8 up to ~1100 tokens; bounded by the LDS tile beyond).
 * Afterwards every context is read back as after cfmm_solve (cfmm_get_solution, cfmm_get_trades*). */
int cfmm_solve_batch(cfmm_ctx *const *ctxs, int nb, const double *const *nu0, const cfmm_opts *opts, cfmm_stats *out);
int cfmm_batch_capacity(int n_tokens);

/* The reference's OWN sweep as one call: two-asset.py:34-100 solves the same 5-pool network (constant-sum pool included,
 * two-asset.py:7-32) under 50 utilities (current_assets = [t, 0, 0], :41-45; psi + current_assets >= 0, :86) and reads value,
 * psi and every pool's tenders at each (:93-100).  B utilities over ONE tiny network -- what one workgroup evaluates: <= 64 tokens,
 * <= 64 wave-tiles, no stableswap / generic / K-asset table pools, one GPU -- are solved in lock-step: a round is ONE launch with
 * one workgroup per unfinished point (the whole first-order solve of a point inside its workgroup), and the active-set loop over the
 * kinks of the constant-sum pools (arbitrage.py:73-74: tie the two prices of a pool found on its kink, re-solve, recover the fill
 * fraction, release ties whose fill leaves (0, 1)) runs inside the library between the rounds.
 *   c, h, ctype, nu0        [B][n]: utility (cfmm_set_utility's arrays; h / ctype may be NULL = 0 / CFMM_GE) and start prices per point
 *   m_sum, sum_*            the constant-sum bucket's columns as uploaded (the host half of the loop reads them); m_sum = 0: none
 *   kink_tol, max_rounds    <= 0: the defaults (1e-3, 6)
 *   nu, psi                 [B][n]: accepted prices and the DEVICE's net trade there (tied pools excluded)
 *   theta, tsgn             [B][m_sum]: fill fraction in (0, 1) and kink direction (+1: tender the first token, drain the second; -1:
 *                           the reverse) of every pool that ended tied on its kink, NaN / 0 elsewhere: psi_total = psi + sum theta d
 *   trades                  NULL, or [B][T]: per point, for every non-empty bucket in the order two-asset kinds 0.., then sizes 3..8,
 *                           delta [k][m] then lambda [k][m] (tied pools: zeros -- their tenders are theta x their full fill)
 *   out, rounds             [B] statistics (evals / iters summed over the rounds; wall / device seconds: of the whole sweep), rounds per point.
 *                           primal_value / dual_value / infeas of a point with tied pools are those of the point WITHOUT the tied pools' fills
 *                           (psi likewise): the caller adds theta x the full fill of each -- the objective is then c'psi */
int cfmm_solve_sweep(cfmm_ctx *ctx, int B, const double *c, const double *h, const int32_t *ctype, const double *nu0,
                     int64_t m_sum, const int32_t *sum_ia, const int32_t *sum_ib, const double *sum_fee, const double *sum_Ra, const double *sum_Rb,
                     const cfmm_opts *opts, double kink_tol, int max_rounds,
                     double *nu, double *psi, double *theta, int32_t *tsgn, double *trades, cfmm_stats *out, int32_t *rounds);

/* read-back (arbitrage.py:84 prob.value is stats.primal_value; psi.value; deltas/lambdas.value) */
int cfmm_get_nu(cfmm_ctx *ctx, double *nu);
int cfmm_set_nu(cfmm_ctx *ctx, const double *nu);
int cfmm_get_psi(cfmm_ctx *ctx, double *psi);
int cfmm_get_solution(cfmm_ctx *ctx, double *nu, double *psi);     /* both with one synchronisation; either may be NULL */
/* tenders at the accepted prices; slot-major [2][m] / [k][m]; either pointer may be NULL */
int cfmm_get_trades2(cfmm_ctx *ctx, int kind, double *delta, double *lambda);
int cfmm_get_tradesN(cfmm_ctx *ctx, int k, double *delta, double *lambda);
int cfmm_get_tradesG(cfmm_ctx *ctx, int kind, int k, double *delta, double *lambda);      /* K-asset table buckets, [k][m] */

/* pool-sharding over the GPUs of a node: one process per GPU, one RCCL all-reduce of
 * [psi | sum arb] per dual evaluation.  `uid` is the 128-byte ncclUniqueId made by rank 0. */
int cfmm_comm_unique_id(void *uid128);
int cfmm_comm_init(cfmm_ctx *ctx, int n_ranks, int rank, const void *uid128);

/* One-shot all-reduce over the xGMI mesh for the iteration's 8-50 KB messages (csrc/oneshot.hpp): every rank stores its
 * vector straight into a mailbox in each peer's HBM, one hop instead of a ring's 2 (R - 1); reduced in rank order, so every
 * rank holds the same bits.  RCCL (cfmm_comm_init) stays the default and carries whatever does not fit a mailbox.
 *   cfmm_oneshot_export: allocates this rank's mailbox, returns its 64-byte hipIpcMemHandle_t;
 *   cfmm_oneshot_import: all ranks' handles ([n_ranks][64], own entry ignored) -> the collectives above use the mailboxes;
 *   cfmm_oneshot_attach / cfmm_oneshot_mailbox: the same with raw device pointers, for ranks living in ONE process
 *     (several contexts on one GPU: how the exchange is tested where a second GPU is not available). */
int cfmm_oneshot_export(cfmm_ctx *ctx, void *handle64);
int cfmm_oneshot_import(cfmm_ctx *ctx, int n_ranks, int rank, const void *handles);
int cfmm_oneshot_attach(cfmm_ctx *ctx, int n_ranks, int rank, void *const *mailboxes);
/* with mailboxes attached AND an RCCL communicator: route the collectives through the one-shot exchange (1) or back through
 * RCCL (0).  Must be called with the same value on every rank (the one-shot epochs advance only while it is on).  Used by
 * cfmm.distributed's start-up check, which compares the two paths on the real peers before trusting the one-shot one. */
int cfmm_oneshot_enable(cfmm_ctx *ctx, int on);
void *cfmm_oneshot_mailbox(cfmm_ctx *ctx);

/* measurement hooks (bench.py): time `reps` back-to-back launches of the fused evaluation kernel
 * with HIP events on the library's stream, over every bucket (kind = CFMM_TIME_ALL: exactly the
 * launch one dual evaluation makes) or restricted to one bucket (kind = CFMM_POOL_*, or -k for
 * the k-asset bucket); returns the average seconds per launch in *sec_per_launch. */
#define CFMM_TIME_ALL 100
#define CFMM_TIME_TABLE 200      /* the K-asset table's own launch alone (every table bucket: table_eval_kernel) */
int cfmm_time_eval_kernel(cfmm_ctx *ctx, int kind, int reps, double *sec_per_launch);
/* the two pool-sharded pieces of an outer iteration, timed the same way (bench.py's per-iteration split): `reps`
 * back-to-back launches of the accumulator-slice fold, and `reps` back-to-back RCCL all-reduces of [psi | sum arb]
 * (n + 1 doubles) -- the latter only on a context with a communicator, and then EVERY rank must make this call */
int cfmm_time_collective(cfmm_ctx *ctx, int reps, double *fold_sec, double *allreduce_sec);
/* measurement: what it costs to hand `np` doubles from ONE workgroup per XCD to the other workgroups of that XCD through its L2 (store,
   acknowledge, flag, spin, load into LDS) -- the price of running iter_kernel's update once per XCD instead of once per workgroup
   (DESIGN (d) "tried and rejected").  out7: median, max over `reps` launches of the slowest follower's [data ready -> data in LDS] in us;
   median [ready -> flag seen]; workgroups whose XCC id is not blockIdx % 8; followers that failed; XCDs without a publisher; the XCC
   ids of workgroups 0 .. 7 as eight decimal digits.
   No reference counterpart (arbitrage.py:82 is one cvxpy call). */
int cfmm_time_xcd_handoff(cfmm_ctx *ctx, int np, int reps, double *out7);
/* the kernels of one second-order step (bench.py --config C5), each timed over `reps` launches at the current prices and
 * barrier weight mu: out4 = seconds per {smoothed evaluation with Hessian assembly, smoothed evaluation alone, dense
 * factorisation (all its launches), back substitution} */
/* NOTE: the hook overwrites the Hessian, the pin mask and the warm starts of the smoothed per-direction solves; a following
 * cfmm_solve(nu0 = NULL) therefore starts its barrier path afresh instead of continuing the previous one */
int cfmm_time_newton_kernels(cfmm_ctx *ctx, double mu, int reps, double *out4);
/* Shader-clock probe (bench.py: roofline.effective_clock_ghz_live).  MI355X clocks to its power budget, so a launch duration alone
 * cannot tell a slower binary from a slower clock.  cfmm_clock_probe_start puts ONE sleeping wave on a stream of its own that, every
 * `period_us`, stores {shader cycles (s_memtime), ticks of the constant 100 MHz counter (s_memrealtime)} into mapped pinned memory
 * until cfmm_clock_probe_stop, `max_ms`, or 8192 samples -- while whatever the caller enqueues on the library's stream runs beside
 * it.  cfmm_clock_probe_read copies the samples so far (no synchronisation; out[2 i] cycles, out[2 i + 1] ticks, oldest first);
 * the clock over an interval = (delta cycles) / (delta ticks x 10 ns).  Replaces nothing in the reference (a measurement hook). */
int cfmm_clock_probe_start(cfmm_ctx *ctx, double period_us, double max_ms);
int cfmm_clock_probe_read(cfmm_ctx *ctx, int64_t *out, int cap, int *count);
int cfmm_clock_probe_stop(cfmm_ctx *ctx, int64_t *out, int cap, int *count);
/* the chain of dependent v_fma_f64 the probe runs in front of its first sample: out3 = {shader cycles, 100 MHz ticks, links} -- cycles per
 * link is a pipeline constant, which shows that the first counter counts shader cycles */
int cfmm_clock_probe_chain(cfmm_ctx *ctx, int64_t *out3);
/* checks the cross-lane primitives of the update kernels (DPP / v_permlane*_swap / ds_swizzle butterflies and the
 * 64-value reduce-scatter) against exact integer sums on this device; 0 = pass */
int cfmm_selftest(cfmm_ctx *ctx);
/* kernel-tuning hook: 32 {shader-cycle, 100 MHz wall-clock} stamp pairs written by the last
 * evaluation / update kernels of a build made with -DCFMM_PHASE_TIMERS (zeros otherwise), then a
 * per-wave tile log and per-block start/end clocks of the last evaluation; `out` holds
 * 64 + 8 * 4096 + 2048 int64 */
int cfmm_debug_timers(cfmm_ctx *ctx, int64_t *out);
#ifdef CFMM_SMOOTH_HIST        /* tuning builds only (tools/build_variants.sh): iteration histogram of the smoothed per-direction solves */
int cfmm_debug_smooth_hist(cfmm_ctx *ctx, uint64_t *out128, int reset);
int cfmm_debug_smooth_samples(cfmm_ctx *ctx, double *out768);
#endif
int64_t cfmm_pool_count(cfmm_ctx *ctx);
/* measurement hook: bytes of pool columns one dual evaluation loads AS STORED (behind the first evaluation / solve: large
 * two-asset buckets then carry a compact mirror of their ids and fee, 21 B per constant-product pool instead of the 32 B
 * SURVEY 8(d) counts) */
int64_t cfmm_eval_bytes(cfmm_ctx *ctx);
void *cfmm_stream(cfmm_ctx *ctx);                 /* the hipStream_t the library launches on */

#ifdef __cplusplus
}
#endif
#endif
