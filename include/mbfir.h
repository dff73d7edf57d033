/*
 * mbfir.h -- C ABI of the MI355X-native convex FIR / SLR beta-polynomial designer.
 *
 * Drop-in boundary for the four convex designers of
 * shanghong/Multiband-RF-pulse-Design.  Each mbfir_*_solve() replaces the body
 * of one reference function from "create optimisation arrays" to "return taps"
 * -- i.e. problem assembly, the external solver call (CVX / linprog / quadprog)
 * and the tap extraction -- and is what a MEX gateway (matlab/mbfir_mex.c) or a
 * ctypes binding (the Python host mirror in this repository) binds to:
 *
 *   mbfir_ap_solve         <- [h,status] = fir_ap_cvx(n,f,a,d,obj,Peak,dbg)   reference fir_ap_cvx.m:1,44-202
 *   mbfir_qp_solve         <- [h,status] = fir_qp_cvx(n,f,a,d,k,obj,dbg)      reference fir_qp_cvx.m:1,34-209
 *   mbfir_linprog_solve    <- [h,status] = fir_linprog(n,f,a,d,h0,dbg)        reference ss/fir_linprog.m:2,46-271
 *   mbfir_qprog_phs_solve  <- [h,status] = fir_qprog_phs(n,f,ac,dc,x0,dbg)    reference ss/fir_qprog_phs.m:1,49-394
 *
 * Conventions (from the reference's MEX precedent rf_tools/mex5/b2a.c:31-68,
 * abrx.c:35-62: plain double arrays, separate real/imaginary planes):
 *   - all arrays are caller-owned, contiguous double; complex data is passed as
 *     separate re/im arrays; `f` has 2*nband entries in [-1,1], `a` 2*nband, `d` nband;
 *   - taps are written to h_re/h_im (n entries each, caller-allocated);
 *   - nothing throws across the ABI, nothing calls exit(); the context is
 *     reusable across calls (bisection wrappers call ~10 times in a row) and is
 *     not thread-safe (distinct contexts are).
 *
 * Return codes of the solve functions:
 *    0  MBFIR_SOLVED        status 'Solved'   (taps written)
 *    1  MBFIR_INFEASIBLE    status 'Failed'   (primal/dual infeasibility certificate found)
 *    2  MBFIR_NUMERICAL     status 'Failed'   (iteration limit / numerical breakdown)
 *    3  MBFIR_EARLY_FAIL    status 'Failed'   (reference returns Failed before solving,
 *                                              ss/fir_linprog.m:66-75, ss/fir_qprog_phs.m:193-202)
 *   <0  usage / HIP error; mbfir_last_error(ctx) holds the message
 *       -1 MBFIR_E_ARG (the reference's error() cases), -2 MBFIR_E_HIP, -3 MBFIR_E_NODEVICE
 */
#ifndef MBFIR_H
#define MBFIR_H

#ifdef __cplusplus
extern "C" {
#endif

#define MBFIR_SOLVED 0
#define MBFIR_INFEASIBLE 1
#define MBFIR_NUMERICAL 2
#define MBFIR_EARLY_FAIL 3
#define MBFIR_E_ARG (-1)
#define MBFIR_E_HIP (-2)
#define MBFIR_E_NODEVICE (-3)

typedef struct mbfir_ctx mbfir_ctx;

/* Solver options.  Zero-initialise (or call mbfir_default_opts) for the defaults. */
typedef struct mbfir_opts {
    int grid_m;        /* number of linspace samples of the frequency grid; 0 = the reference's
                          rule (2*n*15 ap, 10*n qp, 15*n|30*n linprog, 30*n qprog_phs)        */
    int max_iter;      /* 0 -> 200 */
    double feastol;    /* 0 -> 1e-8  relative primal/dual residual                              */
    double abstol;     /* 0 -> 1e-10 absolute gap                                               */
    double reltol;     /* 0 -> 1e-8  relative gap                                               */
    int refine;        /* -1 -> 2  iterative-refinement sweeps per KKT solve; values above 8 are clamped to 8 */
    int verbose;       /* 1: one line per IPM iteration on stderr                               */
    int shard_rank;    /* frequency-row sharding (multi-GPU): this process's rank ...           */
    int shard_size;    /* ... out of shard_size (0 or 1 = not sharded)                          */
    int dense_trig;    /* 0: use the lattice structure of the trig columns / frequency grid when it
                          is there (no trig matrix, moments instead of the dense Gram products);
                          1: always materialise the trig matrix and use the dense MFMA Gram kernel */
    int ddkkt;         /* extended-precision (double-double) KKT solve for nearly active cones whose NT weights
                          exceed 1e6 x the typical weight (fir_qp_cvx's error cones):
                          0 = automatic (on for mbfir_qp_solve, off for the other designers), 1 = on, -1 = off */
    int lanes;         /* mbfir_solve_batch: designs of one shape that advance in LOCK STEP on one context (one stream,
                          one launch per phase, the design index a grid dimension).  0 = automatic (as many as the
                          shape allows while every context still gets a unit), 1 = never, k > 1 = k per unit   */
} mbfir_opts;

/* Per-solve report. */
typedef struct mbfir_info {
    int status;        /* same as the return code */
    int iters;         /* IPM iterations */
    int n_unknowns;    /* N  (columns of the conic program) */
    int n_rows;        /* R  (rows of G) */
    int n_freq;        /* M_f (distinct frequency rows of the trig matrix) */
    int n_lp, n_q3, n_big;
    double pcost, dcost, gap, relgap, pres, dres;
    double ms_assemble, ms_solve, ms_post, ms_total;   /* host wall-clock */
    double ms_gram;      /* device time (HIP events on the solver stream), summed over the builds, of the
                            normal-matrix products: dense mode = the k_gram launches alone; lattice mode =
                            moment kernels + fold + H assembly                                          */
    double ms_chol;      /* device time of the factorisation launches (Cholesky + triangular inverse), summed */
    double gram_flop;    /* algorithmic flop of ONE build: dense nw * Mf * Nt * (Nt+1); lattice: the
                            moment recurrences, (3 D - 1) * Mf * (4 + 4 nw)                              */
    int gram_launches;   /* k_gram launches behind ms_gram (= builds * nw; builds = iterations + 1); 0 in
                            lattice mode                                                                */
    int lattice;         /* 1 if the solve ran in lattice (matrix-free) mode                             */
    double chol_flop;    /* flop of one factorisation + triangular inverse: 2/3 np^3                      */
    int chol_launches;   /* factorisation launches behind ms_chol: ONE k_chol_dag launch per build for lock-step units and
                            from np = 4096 on (round 3); builds * (np/64 + 1) k_chol_step launches for one or two smaller designs */
    int builds;          /* normal-matrix builds (= iterations + 1; + 1 more when the head of the next iteration --
                            scaling, normal matrix, factorisation -- went to the stream before the host had seen the
                            final iterate: that last build is never used, its pivot counter never read; ms_gram,
                            ms_chol and chol_launches count it too)                                               */
    int dd_iters;        /* iterations that ran the extended-precision KKT solve (opts.ddkkt)              */
    int dd_kmax;         /* largest number of strong eigen-directions it carried                          */
    int collectives;     /* all-reduces a row-sharded solve issued (0 otherwise)                                   */
    int lanes;           /* designs that shared this design's lock-step batch (ms_solve, ms_gram, ms_chol are then
                            those of the whole batch)                                                      */
    int dd_form;         /* form of the extended-precision solve that ran: 0 capacitance (saddle-point) form in plain double on
                            the matrix cores, 1 double-double factorisation of the whole matrix, -1 none            */
    double ms_cap;       /* capacitance form: device time of its three matrix-core products (Yt = U M', Zt = Yt M,
                            S = Yt Yt' + X^-1), summed over the builds                                     */
    double cap_flop;     /* ... and their flop, summed over the builds                                     */
    double collective_bytes;   /* bytes the all-reduces of a row-sharded solve carried on this rank (dense path: the PACKED lower
                                  triangle of the normal matrix per build, N (N + 1) / 2 doubles rounded up to 64 x 64 tiles)    */
    int correctors;      /* centrality-corrector solves (one per iteration for programs with orthant rows; 0 with
                            MBFIR_CORRECTOR=0) ...                                                                     */
    int correctors_taken;/* ... and the iterations that took the corrected direction (its step was 1 % longer)         */
    int gv_passes;       /* passes over the frequency rows the solve launched, all iterations: row responses G v ...    */
    int gtv_passes;      /* ... and transposed products G'v (a two-vector pass counts once; a lock-step unit's count)    */
} mbfir_info;

/* All-reduce hook for row-sharded solves (one process per GPU).  `buf` is a DEVICE pointer to
 * `count` doubles on the context's stream-ordered memory; op 0 = sum, 1 = max.  The hook must
 * return after the reduction is complete and visible to the device (0 = ok).  The Python host
 * wires this to torch.distributed (RCCL over xGMI).  What goes through it per IPM iteration: the
 * trigonometric moments of the normal matrix (~100 KB; dense_trig: the np x np normal matrix itself),
 * every G'v and preconditioner application (N doubles each) and a few scalars -- all ranks make the
 * same sequence of calls. */
typedef int (*mbfir_allreduce_fn)(void* buf, long count, int op, void* user);

/* RCCL communicator for row-sharded solves (one process per GPU): rank 0 calls mbfir_comm_unique_id and
 * hands the 128 bytes to the other ranks (the host language's own channel: torch.distributed, MPI, a file), every rank
 * then calls mbfir_comm_init with its rank.  With a communicator the per-iteration reductions are ncclAllReduce calls
 * (ncclDouble, sum / max) enqueued on the solver's stream: no host synchronisation and no callback per collective.
 * Without one, mbfir_set_allreduce's hook is used (the CPU/gloo rehearsal path).  RCCL is bound at run time.      */
int  mbfir_comm_unique_id(mbfir_ctx* ctx, char* id128);
int  mbfir_comm_init(mbfir_ctx* ctx, int nranks, int rank, const char* id128);
void mbfir_comm_destroy(mbfir_ctx* ctx);
/* test hook: all-reduce the host array v (n doubles; op 0 sum, 1 max) through the context's communicator, on its stream */
int  mbfir_test_comm_allreduce(mbfir_ctx* ctx, double* v, long n, int op);

mbfir_ctx*  mbfir_create(int device_id);
void        mbfir_destroy(mbfir_ctx* ctx);
const char* mbfir_last_error(mbfir_ctx* ctx);
void        mbfir_default_opts(mbfir_opts* opts);
void        mbfir_set_allreduce(mbfir_ctx* ctx, mbfir_allreduce_fn fn, void* user);
const char* mbfir_version(void);

int mbfir_ap_solve(mbfir_ctx* ctx, int n, int nband, const double* f, const double* a,
                   const double* d, double obj, double peak, const mbfir_opts* opts,
                   double* h_re, double* h_im, mbfir_info* info);

int mbfir_qp_solve(mbfir_ctx* ctx, int n, int nband, const double* f, const double* a,
                   const double* d, double kquad, const double* obj, int nobj,
                   const mbfir_opts* opts, double* h_re, double* h_im, mbfir_info* info);

int mbfir_linprog_solve(mbfir_ctx* ctx, int n, int nband, const double* f, const double* a,
                        const double* d, const mbfir_opts* opts,
                        double* h_re, double* h_im, mbfir_info* info);

int mbfir_qprog_phs_solve(mbfir_ctx* ctx, int n, int nband, const double* f,
                          const double* ac_re, const double* ac_im,
                          const double* dc_re, const double* dc_im, const mbfir_opts* opts,
                          double* h_re, double* h_im, mbfir_info* info);

/* Batch of independent designs -- the shape of the reference's outer loops (min-order / min-
 * duration bisections probe several n, fir_ap_cvx.m callers such as bSSFP_pulse_sb_mb.m:56-99
 * and dzbeta_min_order; parameter sweeps over obj / Peak).  The jobs are spread over `nctx`
 * contexts (all on one device, or one per device), one host thread per context; each context
 * has its own HIP stream, so the latency-bound phases of different designs overlap on the GPU.
 * `which`: 0 fir_ap_cvx (params = obj, Peak), 1 fir_qp_cvx (params = k, obj[0], obj[1], nobj),
 * 2 fir_linprog, 3 fir_qprog_phs (a: 2*nband and d: nband complex values, re/im interleaved).
 * Every job gets its own rc / info, as from the single-design entry points; returns 0 or the
 * most negative rc of the batch. */
typedef struct mbfir_job {
    int which, n, nband, rc;
    const double *f, *a, *d;
    double params[4];
    double *h_re, *h_im;          /* n doubles each, caller-allocated */
    mbfir_info info;
    double *z;                    /* optional: receives the conic solution (as mbfir_last_solution), up to z_cap doubles */
    int z_cap;
    char err[128];                /* message of the context that ran the job when rc < 0 */
} mbfir_job;

int mbfir_solve_batch(mbfir_ctx* const* ctxs, int nctx, mbfir_job* jobs, int njobs,
                      const mbfir_opts* opts);

/* Conic solution z = [x ; y] / tau of the last solve on this context (n_unknowns doubles of
 * mbfir_info; returns the count copied, or <0).  For fir_ap_cvx x is the autocorrelation
 * [r(0), Re r(1..n-1), Im r(1..n-1)] (fir_ap_cvx.m:185-186), for the others [Re h ; Im h] or the
 * half filter.  Lets callers check feasibility / optimality of what the solver returned. */
int mbfir_last_solution(mbfir_ctx* ctx, double* z, int capacity);

/* ---- introspection / test hooks (host only unless stated) --------------------------------
 * mbfir_assemble(): run the product's problem assembly for designer `which`
 *   (0 ap, 1 qp, 2 linprog, 3 qprog_phs) WITHOUT touching the GPU and return an opaque
 *   program; the accessors below expose its structured rows so tests can expand them to the
 *   dense (c,G,h) and compare with the oracle.  p1..: designer scalars (ap: obj,peak;
 *   qp: kquad,obj0,obj1,nobj).  For qprog_phs `a`,`d` hold interleaved re,im pairs.      */
typedef struct mbfir_program mbfir_program;
int  mbfir_assemble(int which, int n, int nband, const double* f, const double* a, const double* d,
                    const double* params, int grid_m, mbfir_program** out, char* err, int errlen);
void mbfir_program_free(mbfir_program* p);
/* the rows process `rank` of `size` keeps in a row-sharded solve (see mbfir_opts.shard_*): the frequencies are dealt out by
 * folded +w / -w PAIRS (pair q goes to rank q % size: both partners on one rank, one lattice recurrence serves them) with their
 * rows and cones; rows without a frequency (identity rows, spike / per-tap cones, the big cone) are REPLICATED on every rank and
 * counted once in the sums over the rows (mbfir_program_rep) */
int  mbfir_program_shard(const mbfir_program* p, int rank, int size, mbfir_program** out);
/* dims[0..9] = Nt, Ne, R, l, nq3, big, Mf, quad(0/1), nnz_id, reserved */
void mbfir_program_dims(const mbfir_program* p, int* dims);
/* w[Mf]; col_kind[Nt] (0 cos,1 sin); col_tau[Nt]; col_scale[Nt]; pcol[Nt]; psign[Nt]; c[N]     */
void mbfir_program_trig(const mbfir_program* p, double* w, int* col_kind, double* col_tau,
                        double* col_scale, int* pcol, double* psign, double* c);
/* per row: freq (or -1), col (or -1), alpha, beta, ey[3], h                                     */
/* rep[r] = 1 for the rows a row-sharded solve holds on EVERY rank (rows / cones without a frequency, the big cone): R ints */
void mbfir_program_replicated(const mbfir_program* p, int* rep);
void mbfir_program_rows(const mbfir_program* p, int* freq, int* col, double* alpha, double* beta,
                        double* ey, double* h);

/* ---- Inverse SLR: what dzrf_mb.m:239-240 does with the designed beta polynomial ----
 * n complex taps in, host arrays, all caller-allocated with n doubles each.
 *  mbfir_b2a  : minimum-phase alpha of beta, `a = b2a(b)`            (rf_tools/b2a.m:15-32, mag2mp.m:21-31)
 *  mbfir_ab2rf: RF pulse of (alpha, beta), `rf = ab2rf(a, b)`, n <= 2048   (rf_tools/ab2rf.m:14-29)
 *  mbfir_b2rf : `rf = b2rf(b)` = ab2rf(b2a(b), b) without the round trip   (rf_tools/mex5/b2rf.c)
 * rf is in radians per sample, as the reference returns it (dzrf_mb.m:244 rescales it to Gauss). */
int mbfir_b2a(mbfir_ctx* ctx, int n, const double* b_re, const double* b_im, double* a_re, double* a_im);
int mbfir_ab2rf(mbfir_ctx* ctx, int n, const double* a_re, const double* a_im, const double* b_re,
                const double* b_im, double* rf_re, double* rf_im);
int mbfir_b2rf(mbfir_ctx* ctx, int n, const double* b_re, const double* b_im, double* rf_re, double* rf_im);

/* ---- Forward simulation over off-resonance ----------------------------------------------
 * Cayley-Klein parameters (a, b) of the rotation an n-sample pulse produces at nx positions x
 * (rf in radians per sample; g: n per-sample gradient / time weights, NULL = 2 pi / n each, so that x counts
 * cycles over the pulse = frequency x duration).
 *  mode 0: `[a b] = abrm(rf, g, x)` (rf_tools/abrm.m:24-62, the .m twin of the MEX abrx that abr.m:26-30 calls
 *          and that sim_rf_spectral.m's blochC run agrees with for T1, T2 >> pulse length): one rotation about
 *          (Re rf, Im rf, x g) per sample;
 *  mode 1: the hard-pulse model that ab2rf inverts exactly (precession, then the hard pulse) -- closes the loop
 *          b -> mbfir_b2rf -> mbfir_abr -> B(w) to rounding.
 * abr.m's convention is b = -conj(b) of mode 0; mxy = 2 conj(a) b, mz = 1 - 2 |b|^2 (abr.m:11-14). */
int mbfir_abr(mbfir_ctx* ctx, int n, const double* rf_re, const double* rf_im, const double* g, int nx,
              const double* x, int mode, double* a_re, double* a_im, double* b_re, double* b_im);

/* Bloch-equation simulation with relaxation on the device: replaces the MEX bloch_simulation/blochC.c / blochH.c
 * (mexFunction :514-933 -> blochsimfz :422-512 -> blochsim :283-418, calcrotmat :171-236) that sim_rf_spectral.m:63-78
 * runs on the designed pulse.  b1 in Gauss (re / im planes, ntime samples), gx/gy/gz in G/cm (each may be NULL = 0),
 * tsteps = the ntime interval lengths in s, t1/t2 in s, df in Hz (nfreq), dx/dy/dz in cm (npos; each may be NULL = 0),
 * mode bit 0: steady state, bit 1: record every sample (the reference's `mode`), gamma in rad/s/G (6726.1 for C-13 =
 * blochC.c:5, 26754 for H-1 = blochH.c:6).  mx/my/mz: nfreq * npos * (mode & 2 ? ntime : 1) doubles, block (f, p) at
 * (f * npos + p) * ntout; on entry the first entry of every block holds the initial magnetisation (the gateway's
 * :826-866 convention; [0 0 1] for equilibrium), on exit the result.  One thread per (frequency, position).      */
int mbfir_bloch(mbfir_ctx* ctx, int ntime, const double* b1_re, const double* b1_im, const double* gx, const double* gy,
                const double* gz, const double* tsteps, double t1, double t2, int nfreq, const double* df, int npos,
                const double* dx, const double* dy, const double* dz, int mode, double gamma, double* mx, double* my, double* mz);

/* Device kernel test hooks (need a GPU; host arrays in, host arrays out):
 *  mbfir_test_gram: T = A' diag(dk) A for nw weight vectors; A is m x nt row-major,
 *     d is nw x m, out is nw x nt x nt (full symmetric).
 *  mbfir_test_chol: M = inv(chol(H)) for SPD H (n x n); out_l = L, out_m = L^-1 (row-major, lower).
 *  mbfir_test_specfact: x(2n-1) -> taps (n) through the device fmp2/mag2mp.
 *  mbfir_test_mfma_peak: measured fp64 MFMA rate, TFLOP/s (bench.py roofline peak).            */
int mbfir_test_gram(mbfir_ctx* ctx, int m, int nt, int nw, const double* A, const double* d, double* out);
int mbfir_test_chol(mbfir_ctx* ctx, int n, const double* H, double* out_l, double* out_m);
/*  mbfir_test_chol_lanes: the same for `nlanes` matrices factorised TOGETHER, the way a lock-step batch does it (lane index a
 *  grid dimension / the single-launch form); H, out_l, out_m hold nlanes consecutive n x n blocks; mask (nlanes ints or NULL):
 *  lanes with mask 0 are switched off (their outputs stay untouched).  form: -1 = the default of chol_inv_launch for this lane
 *  count, otherwise the value of MBFIR_CHOL_SPLIT to use. */
int mbfir_test_chol_lanes(mbfir_ctx* ctx, int n, int nlanes, int form, const int* mask, const double* H, double* out_l, double* out_m);
int mbfir_test_specfact(mbfir_ctx* ctx, int n, const double* x, double* h_re, double* h_im);
/*  mbfir_test_fold: the host-side analysis of a frequency grid w[m] for the lattice kernels (no GPU, no context): pairs
 *  +w / -w (fold != 0), cuts the folded list into equally spaced runs.  out[6]: lattice usable, folded entries, pairs,
 *  runs, longest run, self-check failures (must be 0).  (Own addition; nothing in the reference corresponds.) */
int mbfir_test_fold(const double* w, int m, int fold, long* out);
/*  mbfir_test_ddsolve: x = (H + U' diag(X) U)^-1 b through the double-double kernels of the extended-precision
 *     KKT solve; H n x n, U k x n (row-major), b and x as (hi, lo) pairs of nrhs x n arrays, nrhs <= 2;
 *     nfix receives the number of replaced pivots; Lh / Ll (optional, n x n) the Cholesky factor.            */
int mbfir_test_ddsolve(mbfir_ctx* ctx, int n, int k, const double* H, const double* U, const double* X, int nrhs,
                       const double* bh, const double* bl, double* xh, double* xl, int* nfix, double* Lh, double* Ll);
int mbfir_test_mfma_peak(mbfir_ctx* ctx, double* tflops_f64_mfma, double* tflops_f64_valu);
/* Device time (ms per call, HIP events, averaged over reps) of the Cholesky+inverse phase on a random SPD
 * n x n matrix, and of the Gram launches on a random m x nt matrix -- kernel tuning aid. */
int mbfir_test_time_kernels(mbfir_ctx* ctx, int n, int m, int nt, int reps, double* ms_chol, double* ms_gram);

#ifdef __cplusplus
}
#endif
#endif
