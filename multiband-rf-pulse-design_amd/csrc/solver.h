// Host-visible interface of the device IPM (solver.hip).
#pragma once
#include "program.h"
#include <stdexcept>
#include <string>
#include <vector>

namespace mbfir {

// solve_lanes was handed programs the lock-step path does not take as ONE unit (too many lanes, different designers or
// orders): the caller may run the designs one by one.  Every other exception of the solver is an internal / device error
// and is reported as such, never retried.
struct ShapeError : std::runtime_error {
    explicit ShapeError(const std::string& s) : std::runtime_error(s) {}
};
// a lock-step unit did not get its device memory (every lane is sized to the unit's maxima): recoverable -- the caller takes
// the unit's designs through the single-design path -- unlike an internal or device error, which is reported
struct ResourceError : std::runtime_error {
    explicit ResourceError(const std::string& s) : std::runtime_error(s) {}
};

enum { ST_OPTIMAL = 0, ST_PRIMAL_INFEASIBLE = 1, ST_DUAL_INFEASIBLE = 2, ST_MAXIT = 3, ST_NUMERICAL = 4,
       ST_OPTIMAL_INACCURATE = 5 };

struct SolveOpts {
    int max_iter = 200;
    double feastol = 1e-8, abstol = 1e-10, reltol = 1e-8;
    int refine = 2;
    int verbose = 0;
    int shard_rank = 0, shard_size = 1;     // frequency-row sharding (one process per GPU)
    bool dense_trig = false; // keep the materialised trig matrix and the dense MFMA Gram even when the lattice structure is there
    double ddkkt_theta = 0; // > 0: extended-precision (double-double) KKT solve for every NT weight above theta x the typical weight
    int dd_form = 0;        // the extended-precision solve: 0 its capacitance (saddle-point) form in plain double on the matrix cores
                            // (capkkt.hip), 1 the double-double factorisation of the whole normal matrix (ddlin.hip)
    bool timing = true;     // HIP-event timing of the k_gram launches and the Cholesky phase (events read at the end)
};

struct SolveInfo {
    int status = 0, iters = 0, n_unknowns = 0, n_rows = 0, n_freq = 0;
    double pcost = 0, dcost = 0, gap = 0, relgap = 0, pres = 0, dres = 0;
    double ms_assemble = 0, ms_solve = 0, ms_gram = 0, ms_chol = 0, gram_flop = 0;
    int h_builds = 0;       // number of (Gram, Cholesky) builds = iterations + 1 (initial point)
    int chol_launches = 0;  // k_chol_step launches timed in ms_chol
    double chol_flop = 0;   // factorisation + triangular inverse, per build
    int dd_iters = 0, dd_kmax = 0;   // iterations that ran the extended-precision solve; largest strong set
    double ms_cap = 0, cap_flop = 0; // capacitance form: device time and flop of the three matrix-core products (Yt, Zt, S), summed over the builds
    int dd_form = 0;                 // which form of that solve ran (0 capacitance / double, 1 double-double)
    int collectives = 0;    // all-reduces the (row-sharded) solve issued
    double collective_bytes = 0;   // ... and the bytes they carried (per rank)
    int lanes = 1;          // designs that shared the lock-step batch (ms_* are those of the whole batch)
    int lattice = 0;        // 1: lattice (matrix-free) mode; gram_flop then counts the moment recurrences
    int correctors = 0, correctors_taken = 0;   // centrality-corrector solves of this design, and how many of the corrected directions were taken
    int gv_passes = 0, gtv_passes = 0;          // row-response passes G v and transposed passes G'v the solve launched (all iterations; a lock-step unit's count)
};

class Solver {
public:
    explicit Solver(int device);
    ~Solver();
    Solver(const Solver&) = delete;
    Solver& operator=(const Solver&) = delete;
    // Solve the conic program; xout = x / tau (N entries).  Returns ST_*; throws HipError.
    int solve(const TrigProgram& P, const SolveOpts& o, std::vector<double>& xout, SolveInfo& info);
    // Lock-step batch: the programs (all of one shape, see shape_key) advance together through one stream, one
    // launch per phase with the design index as a grid dimension; designs that finish are masked out.
    void solve_lanes(const std::vector<const TrigProgram*>& Ps, const SolveOpts& o, std::vector<std::vector<double>>& xouts,
                     std::vector<SolveInfo>& infos);
    // equal keys = may share a lock-step unit: the program's CLASS (designer, order, unknowns, cones, lattice extent); grids,
    // row counts and chunk lists may differ within a unit on the lattice path without a big cone (heterogeneous units), elsewhere
    // the key also holds the exact shape
    static std::vector<long> shape_key(const TrigProgram& P, const SolveOpts& o);
    static int max_lanes(const TrigProgram& P, const SolveOpts& o);
    // host analysis of a frequency grid (no GPU needed): out[0] lattice ok, [1] folded entries, [2] pairs, [3] runs,
    // [4] longest run, [5] entries that fail the self-check (every frequency exactly once, |w| within 1 ulp of the entry's)
    static void test_fold(const double* w, int Mf, int fold, long* out);
    // fir_ap_cvx tap extraction on the device from the solution left by the last solve() / lane of solve_lanes().
    void specfact_last(int n, double* h_re, double* h_im, int lane = 0);
    void set_solution(const std::vector<double>& x);
    // Inverse SLR on the device (slr.hip).  b: n complex taps.  a_in null: a = b2a(b) (b2a.m:15-32), else a = a_in.
    // a_out (optional) receives a; rf (optional) receives ab2rf(a, b) (ab2rf.m:14-29).
    void slr(int n, const double* b_re, const double* b_im, const double* a_in_re, const double* a_in_im,
             double* a_re, double* a_im, double* rf_re, double* rf_im);
    // Forward simulation (slr.hip k_abr): a, b over nx positions; g null = 2 pi / n per sample; mode 0 abrm.m, 1 hard pulse.
    void abr(int n, const double* rf_re, const double* rf_im, const double* g, int nx, const double* x, int mode,
             double* a_re, double* a_im, double* b_re, double* b_im);
    // Bloch simulation with relaxation (slr.hip k_bloch; blochC.c:422-512).  m*: in = initial magnetisation at the first
    // sample of every (frequency, position) block, out = the result; nfreq * npos * (mode & 2 ? ntime : 1) doubles each.
    void bloch(int ntime, const double* b1_re, const double* b1_im, const double* gx, const double* gy, const double* gz,
               const double* tsteps, double t1, double t2, int nfreq, const double* df, int npos, const double* dx,
               const double* dy, const double* dz, int mode, double gamma, double* mx, double* my, double* mz);
    // kernel test hooks
    void test_gram(int m, int nt, int nw, const double* A, const double* d, double* out);
    void test_chol(int n, const double* H, double* out_l, double* out_m);
    void test_chol_lanes(int n, int nlanes, int form, const int* mask, const double* H, double* out_l, double* out_m);
    void test_specfact(int n, const double* x, double* h_re, double* h_im);
    void test_ddsolve(int n, int k, const double* H, const double* U, const double* X, int nrhs, const double* bh,
                      const double* bl, double* xh, double* xl, int* nfix, double* Lh_out = nullptr, double* Ll_out = nullptr);
    void test_mfma_peak(double* tf_mfma, double* tf_valu);
    void test_time_kernels(int n, int m, int nt, int reps, double* ms_chol, double* ms_gram);
    void* stream() const;
    // RCCL communicator for row-sharded solves (one process per GPU): rank 0 makes the id, everybody joins
    void comm_unique_id(char* id128);
    void comm_init(int nranks, int rank, const char* id128);
    void comm_destroy();
    void test_comm_allreduce(double* v, long n, int op);
    void set_allreduce(int (*fn)(void*, long, int, void*), void* user);

private:
    struct Impl;
    Impl* impl;
};

}  // namespace mbfir
