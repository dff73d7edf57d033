// C ABI of include/mbfir.h: context management and the four designer entry points.
// Each entry point = host assembly (assemble.cpp) -> device IPM (solver.hip) -> tap extraction.
#include "../../include/mbfir.h"
#include "program.h"
#include "solver.h"
#include <chrono>
#include <cstdio>
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <condition_variable>
#include <map>
#include <mutex>
#include <memory>
#include <system_error>
#include <thread>

using namespace mbfir;

struct mbfir_ctx {
    std::unique_ptr<Solver> solver;
    std::vector<double> last_x;
    std::string err;
    mbfir_allreduce_fn allreduce = nullptr;
    void* allreduce_user = nullptr;
    bool has_comm = false;
};

namespace {

thread_local std::string g_create_error;

double now_ms() {
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

constexpr double DDKKT_THETA = 1e6;       // oracle/designers.py uses the same value

// which: the designer (DES_*): the extended-precision KKT solve is on by default for fir_qp_cvx only
SolveOpts to_opts(const mbfir_opts* o, int which) {
    SolveOpts s;
    s.ddkkt_theta = which == DES_QP ? DDKKT_THETA : 0.0;
    if (!o) return s;
    if (o->ddkkt > 0) s.ddkkt_theta = DDKKT_THETA;
    if (o->ddkkt < 0) s.ddkkt_theta = 0.0;
    if (o->max_iter > 0) s.max_iter = o->max_iter;
    if (o->feastol > 0) s.feastol = o->feastol;
    if (o->abstol > 0) s.abstol = o->abstol;
    if (o->reltol > 0) s.reltol = o->reltol;
    if (o->refine >= 0) s.refine = o->refine > 8 ? 8 : o->refine;   // MAX_SWEEPS of solver.hip: the norm slots hold 9 values
    s.verbose = o->verbose;
    s.shard_rank = o->shard_rank;
    s.shard_size = o->shard_size;
    s.dense_trig = o->dense_trig != 0;
    return s;
}

int status_to_rc(int st) {
    switch (st) {
        case ST_OPTIMAL:
        case ST_OPTIMAL_INACCURATE: return MBFIR_SOLVED;      // reference accepts 'Inaccurate/Solved' (fir_ap_cvx.m:176)
        case ST_PRIMAL_INFEASIBLE:
        case ST_DUAL_INFEASIBLE: return MBFIR_INFEASIBLE;
        default: return MBFIR_NUMERICAL;
    }
}

void fill_info(mbfir_info* info, const TrigProgram& P, const SolveInfo& si, int rc, double t0, double t_asm,
               double t_solved, double t_end) {
    if (!info) return;
    std::memset(info, 0, sizeof(*info));
    info->status = rc; info->iters = si.iters; info->n_unknowns = P.N();
    info->n_rows = si.n_rows; info->n_freq = si.n_freq;      // of this process's row shard
    info->n_lp = P.l; info->n_q3 = P.nq3; info->n_big = P.big;
    info->pcost = si.pcost; info->dcost = si.dcost; info->gap = si.gap; info->relgap = si.relgap;
    info->pres = si.pres; info->dres = si.dres;
    info->ms_assemble = (t_asm - t0) + si.ms_assemble; info->ms_solve = si.ms_solve;
    info->ms_post = t_end - t_solved; info->ms_total = t_end - t0;
    info->ms_gram = si.ms_gram; info->ms_chol = si.ms_chol; info->gram_flop = si.gram_flop;
    info->gram_launches = si.lattice ? 0 : si.h_builds * (P.quad ? 3 : 1);
    info->lattice = si.lattice;
    info->chol_launches = si.chol_launches; info->chol_flop = si.chol_flop; info->builds = si.h_builds;
    info->dd_iters = si.dd_iters; info->dd_kmax = si.dd_kmax; info->lanes = si.lanes; info->collectives = si.collectives;
    info->collective_bytes = si.collective_bytes;
    info->ms_cap = si.ms_cap; info->cap_flop = si.cap_flop; info->dd_form = si.dd_iters > 0 ? si.dd_form : -1;
    info->correctors = si.correctors; info->correctors_taken = si.correctors_taken; info->gv_passes = si.gv_passes; info->gtv_passes = si.gtv_passes;
}

// Solution vector -> taps, per designer.  lane: which design of the solver's last lock-step batch (fir_ap_cvx runs
// the spectral factorisation on the device from the solution left there).
void taps_from_solution(Solver& solver, const TrigProgram& P, const std::vector<double>& x, int lane, double* h_re, double* h_im) {
    const int n = P.n;
    switch (P.which) {
        case DES_AP:
            solver.specfact_last(n, h_re, h_im, lane);           // fir_ap_cvx.m:185-186,202
            break;
        case DES_LINPROG: {
            // fill_h, ss/fir_linprog.m:274-296: Hermitian extension of the half filter
            const int nh = P.nhalf;
            for (int i = 0; i < n; ++i) { h_re[i] = 0; h_im[i] = 0; }
            if (P.real_filter) {
                if (P.odd_filter) for (int k = 0; k < nh; ++k) { h_re[nh - 1 + k] = x[k]; h_re[nh - 1 - k] = x[k]; }
                else for (int k = 0; k < nh; ++k) { h_re[nh + k] = x[k]; h_re[nh - 1 - k] = x[k]; }
            } else if (P.odd_filter) {
                for (int k = 0; k < nh; ++k) {
                    double re = x[k], im = k == 0 ? 0.0 : x[nh + k - 1];
                    h_re[nh - 1 + k] = re; h_im[nh - 1 + k] = im;
                    h_re[nh - 1 - k] = re; h_im[nh - 1 - k] = -im;
                }
            } else {
                for (int k = 0; k < nh; ++k) {
                    double re = x[k], im = x[nh + k];
                    h_re[nh + k] = re; h_im[nh + k] = im;
                    h_re[nh - 1 - k] = re; h_im[nh - 1 - k] = -im;
                }
            }
            break;
        }
        default:                                                 // fir_qp_cvx.m:209, ss/fir_qprog_phs.m:389
            for (int i = 0; i < n; ++i) { h_re[i] = x[i]; h_im[i] = x[n + i]; }
    }
}

// Common driver: `asm_rc` is the assembly result, `post` maps the solution vector to taps.
template <class Post>
int run(mbfir_ctx* ctx, int asm_rc, const std::string& asm_err, TrigProgram& P, const mbfir_opts* opts,
        mbfir_info* info, double t0, Post post) {
    if (!ctx) return MBFIR_E_ARG;
    if (info) std::memset(info, 0, sizeof(*info));
    if (asm_rc != 0) {
        ctx->err = asm_err;
        if (info) info->status = asm_rc;
        return asm_rc;
    }
    if (opts && opts->shard_size > 1 && !ctx->allreduce && !ctx->has_comm) {
        ctx->err = "row-sharded solve requested without a communicator (mbfir_comm_init) or an all-reduce hook";
        return MBFIR_E_ARG;
    }
    try {
        double t_asm = now_ms();
        SolveInfo si;
        std::vector<double> x;
        SolveOpts so = to_opts(opts, P.which);
        int st = ctx->solver->solve(P, so, x, si);
        // A numerical failure (never an infeasibility certificate) -- or a solve that only met the reduced tolerances
        // ('Inaccurate/Solved') -- is a precision limit of the double-precision normal equations: near cond(H) ~ 1e14 the
        // Cholesky factor starts replacing pivots, a few iterations earlier on the lattice path, whose moments carry
        // 1e-14 of recurrence noise, than on the dense one.  The remedy is precision, not the
        // path: retry once with the extended-precision KKT solve (same path), and only if that does not give a full-
        // accuracy solve either once more on the dense path.  The best of the attempts is returned.
        bool ran_retry = false;
        auto rank_of = [](int status) { return status == ST_OPTIMAL ? 2 : status == ST_OPTIMAL_INACCURATE ? 1 : 0; };
        auto retry = [&](bool dense) {
            const SolveInfo first = si;
            const std::vector<double> x_first = x;
            const int st_first = st;
            so.ddkkt_theta = DDKKT_THETA;
            so.dense_trig = so.dense_trig || dense;
            st = ctx->solver->solve(P, so, x, si);
            const bool keep_first = rank_of(st_first) > rank_of(st);
            const SolveInfo second = si;
            if (keep_first) { si = first; x = x_first; st = st_first; }
            const SolveInfo& other = keep_first ? second : first;
            si.ms_assemble += other.ms_assemble; si.ms_solve += other.ms_solve; si.ms_chol += other.ms_chol;
            si.ms_gram += other.ms_gram; si.h_builds += other.h_builds; si.chol_launches += other.chol_launches;
            si.iters += other.iters; si.dd_iters += other.dd_iters; si.ms_cap += other.ms_cap; si.cap_flop += other.cap_flop;
            ran_retry = true;
            return !keep_first;
        };
        auto wants_retry = [&]() { return status_to_rc(st) == MBFIR_NUMERICAL || st == ST_OPTIMAL_INACCURATE; };
        if (wants_retry() && so.shard_size <= 1 && !(so.ddkkt_theta > 0) && !(opts && opts->ddkkt < 0)) retry(false);
        // the extended-precision retry does not exist row-sharded (its double-double factorisation works on the whole normal
        // matrix of ONE context): the verdict stands, with the reason on record for the caller
        if (wants_retry() && so.shard_size > 1)
            ctx->err = st == ST_OPTIMAL_INACCURATE
                           ? "row-sharded solve met the reduced tolerances only ('Inaccurate/Solved'); the extended-precision retry runs unsharded only"
                           : "row-sharded solve ended 'numerical'; the extended-precision retry (opts.ddkkt) runs unsharded only -- solve this design on one GPU";
        // the capacitance form of the extended-precision solve works in plain double on a k x k matrix that loses rank when the
        // strong directions become dependent; the double-double factorisation of the whole matrix does not care: one more try
        if (wants_retry() && so.shard_size <= 1 && so.ddkkt_theta > 0 && so.dd_form == 0 && !std::getenv("MBFIR_DDFORM")) { so.dd_form = 1; retry(false); }
        if (wants_retry() && si.lattice && !so.dense_trig && so.shard_size <= 1 && (double)P.Mf * P.N() <= 6e8) retry(true);
        // fir_ap_cvx extracts its taps on the device from the solution the LAST solve left there; after a retry the
        // winner may be an earlier attempt: put the returned solution there
        if (ran_retry && status_to_rc(st) == MBFIR_SOLVED) ctx->solver->set_solution(x);
        ctx->last_x = x;
        double t_solved = now_ms();
        int rc = status_to_rc(st);
        if (rc == MBFIR_SOLVED) post(x);
        fill_info(info, P, si, rc, t0, t_asm, t_solved, now_ms());
        return rc;
    } catch (const std::exception& e) {
        ctx->err = e.what();
        if (info) info->status = MBFIR_E_HIP;
        return MBFIR_E_HIP;
    }
}

}  // namespace

// ---- fault diagnostics (MBFIR_FAULT_MAPS=<file>) ---------------------------------------------------
// Round 3 recorded a SIGSEGV below hipLaunchKernel under `rocprofv3 --kernel-trace` whose log held bare PCs and no module map,
// so the frames between this library and the faulting copy could not be attributed.  With MBFIR_FAULT_MAPS set, the first
// mbfir_create installs a SIGSEGV / SIGBUS handler that writes the fault address and /proc/self/maps -- taken AT THE FAULT,
// so that queue rings, kernarg pools and thread arenas mapped since start-up are in it -- to that file (async-signal-safe
// calls only) and then hands the signal to whoever was installed before (the profiler's stack printer, the default action).
#include <csignal>
#include <fcntl.h>
#include <unistd.h>
namespace {
char g_maps_path[512];
struct sigaction g_old_segv, g_old_bus;
void put_hex(int fd, unsigned long v) {
    char b[19] = "0x0000000000000000";
    for (int i = 0; i < 16; ++i) b[17 - i] = "0123456789abcdef"[(v >> (4 * i)) & 15];
    (void)!write(fd, b, 18);
}
void fault_handler(int sig, siginfo_t* si, void* uc) {
    const int out = open(g_maps_path, O_WRONLY | O_CREAT | O_APPEND, 0644);
    if (out >= 0) {
        const char h1[] = "\n==== mbfir fault: signal ";
        (void)!write(out, h1, sizeof(h1) - 1);
        put_hex(out, (unsigned long)sig);
        const char h2[] = " at address ";
        (void)!write(out, h2, sizeof(h2) - 1);
        put_hex(out, (unsigned long)si->si_addr);
        const char h3[] = " ; /proc/self/maps at the fault:\n";
        (void)!write(out, h3, sizeof(h3) - 1);
        const int in = open("/proc/self/maps", O_RDONLY);
        if (in >= 0) {
            char buf[4096];
            for (;;) { const ssize_t n = read(in, buf, sizeof(buf)); if (n <= 0) break; (void)!write(out, buf, size_t(n)); }
            close(in);
        }
        close(out);
    }
    struct sigaction* old = sig == SIGBUS ? &g_old_bus : &g_old_segv;
    sigaction(sig, old, nullptr);                          // back to the previous handler: returning re-executes the faulting access
    (void)uc;
}
void install_fault_maps() {
    static std::once_flag once;
    std::call_once(once, [] {
        const char* pth = std::getenv("MBFIR_FAULT_MAPS");
        if (!pth || !*pth) return;
        std::snprintf(g_maps_path, sizeof(g_maps_path), "%s", pth);
        struct sigaction sa;
        std::memset(&sa, 0, sizeof(sa));
        sa.sa_sigaction = fault_handler;
        sa.sa_flags = SA_SIGINFO | SA_ONSTACK;
        sigemptyset(&sa.sa_mask);
        sigaction(SIGSEGV, &sa, &g_old_segv);
        sigaction(SIGBUS, &sa, &g_old_bus);
    });
}
}  // namespace

extern "C" {

const char* mbfir_version(void) { return "mbfir 0.1 (gfx950)"; }

void mbfir_default_opts(mbfir_opts* o) {
    if (!o) return;
    std::memset(o, 0, sizeof(*o));
    o->refine = -1;
}

mbfir_ctx* mbfir_create(int device_id) {
    try {
        install_fault_maps();                                  // (a profiler's handlers are installed at load: this one chains to them)
        mbfir_ctx* c = new mbfir_ctx();
        c->solver.reset(new Solver(device_id));
        return c;
    } catch (const std::exception& e) {
        g_create_error = e.what();
        return nullptr;
    }
}

void mbfir_destroy(mbfir_ctx* ctx) { delete ctx; }

const char* mbfir_last_error(mbfir_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

void mbfir_set_allreduce(mbfir_ctx* ctx, mbfir_allreduce_fn fn, void* user) {
    if (!ctx) return;
    ctx->allreduce = fn;
    ctx->allreduce_user = user;
    ctx->solver->set_allreduce(fn, user);
}

int mbfir_comm_unique_id(mbfir_ctx* ctx, char* id128) {
    if (!ctx || !id128) return MBFIR_E_ARG;
    try { ctx->solver->comm_unique_id(id128); return 0; }
    catch (const std::exception& e) { ctx->err = e.what(); return MBFIR_E_HIP; }
}
int mbfir_comm_init(mbfir_ctx* ctx, int nranks, int rank, const char* id128) {
    if (!ctx || !id128 || nranks < 1 || rank < 0 || rank >= nranks) return MBFIR_E_ARG;
    try { ctx->solver->comm_init(nranks, rank, id128); ctx->has_comm = true; return 0; }
    catch (const std::exception& e) { ctx->err = e.what(); return MBFIR_E_HIP; }
}
int mbfir_test_comm_allreduce(mbfir_ctx* ctx, double* v, long n, int op) {
    if (!ctx || !v || n < 0) return MBFIR_E_ARG;
    try { ctx->solver->test_comm_allreduce(v, n, op); return 0; }
    catch (const std::exception& e) { ctx->err = e.what(); return MBFIR_E_HIP; }
}
void mbfir_comm_destroy(mbfir_ctx* ctx) {
    if (!ctx) return;
    ctx->solver->comm_destroy();
    ctx->has_comm = false;
}

int mbfir_last_solution(mbfir_ctx* ctx, double* z, int capacity) {
    if (!ctx || !z) return MBFIR_E_ARG;
    int n = int(ctx->last_x.size());
    if (capacity < n) n = capacity;
    std::memcpy(z, ctx->last_x.data(), sizeof(double) * n);
    return n;
}

int mbfir_ap_solve(mbfir_ctx* ctx, int n, int nband, const double* f, const double* a, const double* d,
                   double obj, double peak, const mbfir_opts* opts, double* h_re, double* h_im,
                   mbfir_info* info) {
    double t0 = now_ms();
    TrigProgram P;
    std::string e;
    int rc = assemble_ap(n, nband, f, a, d, obj, peak, opts ? opts->grid_m : 0, P, e);
    return run(ctx, rc, e, P, opts, info, t0, [&](const std::vector<double>& x) { taps_from_solution(*ctx->solver, P, x, 0, h_re, h_im); });
}

int mbfir_qp_solve(mbfir_ctx* ctx, int n, int nband, const double* f, const double* a, const double* d,
                   double kquad, const double* obj, int nobj, const mbfir_opts* opts, double* h_re,
                   double* h_im, mbfir_info* info) {
    double t0 = now_ms();
    TrigProgram P;
    std::string e;
    int rc = assemble_qp(n, nband, f, a, d, kquad, obj, nobj, opts ? opts->grid_m : 0, P, e);
    return run(ctx, rc, e, P, opts, info, t0, [&](const std::vector<double>& x) { taps_from_solution(*ctx->solver, P, x, 0, h_re, h_im); });
}

int mbfir_linprog_solve(mbfir_ctx* ctx, int n, int nband, const double* f, const double* a, const double* d,
                        const mbfir_opts* opts, double* h_re, double* h_im, mbfir_info* info) {
    double t0 = now_ms();
    TrigProgram P;
    std::string e;
    int rc = assemble_linprog(n, nband, f, a, d, opts ? opts->grid_m : 0, P, e);
    return run(ctx, rc, e, P, opts, info, t0, [&](const std::vector<double>& x) { taps_from_solution(*ctx->solver, P, x, 0, h_re, h_im); });
}

int mbfir_qprog_phs_solve(mbfir_ctx* ctx, int n, int nband, const double* f, const double* ac_re,
                          const double* ac_im, const double* dc_re, const double* dc_im,
                          const mbfir_opts* opts, double* h_re, double* h_im, mbfir_info* info) {
    double t0 = now_ms();
    TrigProgram P;
    std::string e;
    int rc = assemble_qprog_phs(n, nband, f, ac_re, ac_im, dc_re, dc_im, opts ? opts->grid_m : 0, P, e);
    return run(ctx, rc, e, P, opts, info, t0, [&](const std::vector<double>& x) { taps_from_solution(*ctx->solver, P, x, 0, h_re, h_im); });
}

// ---- batch of independent designs --------------------------------------------------------------
// The jobs are assembled on the host, grouped by shape (Solver::shape_key: same dimensions, same lattice
// structure) and cut into units of up to `lanes` designs.  A unit of one design runs as a single solve; a unit of
// several runs in LOCK STEP on one context: one stream, one launch per phase with the design index as a grid
// dimension, finished designs masked out (Solver::solve_lanes).  One host thread per context pulls units from a
// shared counter, so units also overlap across the contexts' streams.  The dependency-chain phases of one design
// (Cholesky panels, reductions) leave most of the chip idle; the other lanes' blocks fill it.
static int one_job(mbfir_ctx* ctx, mbfir_job& J, const mbfir_opts* opts) {
    switch (J.which) {
        case DES_AP:
            return mbfir_ap_solve(ctx, J.n, J.nband, J.f, J.a, J.d, J.params[0], J.params[1], opts, J.h_re, J.h_im, &J.info);
        case DES_QP:
            return mbfir_qp_solve(ctx, J.n, J.nband, J.f, J.a, J.d, J.params[0], J.params + 1, int(J.params[3]), opts, J.h_re, J.h_im, &J.info);
        case DES_LINPROG:
            return mbfir_linprog_solve(ctx, J.n, J.nband, J.f, J.a, J.d, opts, J.h_re, J.h_im, &J.info);
        case DES_QPROG_PHS: {
            std::vector<double> are(2 * J.nband), aim(2 * J.nband), dre(J.nband), dim(J.nband);
            for (int i = 0; i < 2 * J.nband; ++i) { are[i] = J.a[2 * i]; aim[i] = J.a[2 * i + 1]; }
            for (int i = 0; i < J.nband; ++i) { dre[i] = J.d[2 * i]; dim[i] = J.d[2 * i + 1]; }
            return mbfir_qprog_phs_solve(ctx, J.n, J.nband, J.f, are.data(), aim.data(), dre.data(), dim.data(), opts, J.h_re, J.h_im, &J.info);
        }
        default: return MBFIR_E_ARG;
    }
}
static int assemble_job(const mbfir_job& J, int grid_m, TrigProgram& P, std::string& e) {
    switch (J.which) {
        case DES_AP: return assemble_ap(J.n, J.nband, J.f, J.a, J.d, J.params[0], J.params[1], grid_m, P, e);
        case DES_QP: return assemble_qp(J.n, J.nband, J.f, J.a, J.d, J.params[0], J.params + 1, int(J.params[3]), grid_m, P, e);
        case DES_LINPROG: return assemble_linprog(J.n, J.nband, J.f, J.a, J.d, grid_m, P, e);
        case DES_QPROG_PHS: {
            std::vector<double> are(2 * J.nband), aim(2 * J.nband), dre(J.nband), dim(J.nband);
            for (int i = 0; i < 2 * J.nband; ++i) { are[i] = J.a[2 * i]; aim[i] = J.a[2 * i + 1]; }
            for (int i = 0; i < J.nband; ++i) { dre[i] = J.d[2 * i]; dim[i] = J.d[2 * i + 1]; }
            return assemble_qprog_phs(J.n, J.nband, J.f, are.data(), aim.data(), dre.data(), dim.data(), grid_m, P, e);
        }
        default: e = "unknown designer"; return MBFIR_E_ARG;
    }
}

// Assembled programs are a few MB of vectors each; giving them back to the allocator is 0.1-0.2 ms a piece (munmap).  The
// workers do that for their unit while other units still run; the programs of the LAST unit to finish -- the one every
// caller waits for -- are parked here and freed by the next batch's coordinator while its units run (or at unload).
static std::mutex g_parked_mu;
static std::vector<TrigProgram> g_parked;

int mbfir_solve_batch(mbfir_ctx* const* ctxs, int nctx, mbfir_job* jobs, int njobs, const mbfir_opts* opts) {
    if (!ctxs || nctx < 1 || (!jobs && njobs > 0) || njobs < 0) return MBFIR_E_ARG;
    for (int c = 0; c < nctx; ++c)
        if (!ctxs[c]) return MBFIR_E_ARG;
    const double t0 = now_ms();
    const bool sharded = opts && opts->shard_size > 1;
    int lanes_cap = opts && opts->lanes != 0 ? opts->lanes : 0;              // 0 = automatic, 1 = never lock-step
    if (const char* ev = std::getenv("MBFIR_LANES")) lanes_cap = std::atoi(ev);
    // ---- host assembly, in parallel; units start as soon as their designs are assembled -------------------
    // Round 2 assembled all jobs, then grouped, then started the units: 8 ms of a 265 ms headline batch with the GPU idle,
    // and the unit that decides the makespan (the tight-ripple designs at the head of a sweep) started last of all.  Now the
    // assembly threads take the jobs in order, and a coordinator (the calling thread) hands a unit to the waiting contexts
    // the moment its designs are there.  The units are formed SPECULATIVELY from neighbours in the job list on the
    // assumption that the batch has one shape (a sweep): the first design that differs from job 0, or fails to assemble,
    // ends the speculation, and what has not been handed out yet is grouped by shape as before.
    std::vector<TrigProgram> progs(njobs);
    std::vector<int> arc(njobs, 0);
    std::vector<std::vector<long>> keys(njobs);
    std::mutex mu;
    std::condition_variable cv_asm, cv_units;
    std::vector<char> assembled(njobs, 0);
    std::vector<std::vector<int>> units;
    size_t units_taken = 0, units_done = 0;
    bool units_closed = false;
    const bool trace = std::getenv("MBFIR_TRACE_BATCH") != nullptr;
    std::atomic<int> next_asm(0);
    auto asm_work = [&]() {
        for (;;) {
            const int q = next_asm.fetch_add(1);
            if (q >= njobs) break;
            std::string e;
            jobs[q].err[0] = 0;
            try {
                arc[q] = sharded || lanes_cap == 1 ? 0 : assemble_job(jobs[q], opts ? opts->grid_m : 0, progs[q], e);
                if (arc[q] == 0 && !(sharded || lanes_cap == 1)) {
                    // the shape key (CSR maps, lattice analysis: a sort of the grid) here, not in the serial grouping loop:
                    // 64 headline designs cost 40 ms there with the GPU idle; the structures stay with the program
                    const SolveOpts so = to_opts(opts, progs[q].which);
                    keys[q] = Solver::shape_key(progs[q], so);
                    keys[q].push_back(Solver::max_lanes(progs[q], so));
                }
            } catch (const std::exception& ex) {               // (host memory: the job is reported, the coordinator is not left waiting)
                arc[q] = MBFIR_E_HIP; e = ex.what();
            }
            if (arc[q] != 0) std::snprintf(jobs[q].err, sizeof(jobs[q].err), "%s", e.c_str());
            { std::lock_guard<std::mutex> lk(mu); assembled[q] = 1; }
            cv_asm.notify_all();
        }
    };
    auto wait_assembled = [&](int lo, int hi) {
        std::unique_lock<std::mutex> lk(mu);
        cv_asm.wait(lk, [&] { for (int q = lo; q < hi; ++q) if (!assembled[q]) return false; return true; });
    };
    auto push_unit = [&](std::vector<int> u) {
        { std::lock_guard<std::mutex> lk(mu); units.push_back(std::move(u)); }
        cv_units.notify_one();
    };
    int order = 0;
    if (const char* ev = std::getenv("MBFIR_UNIT_ORDER")) order = std::atoi(ev);
    // ---- units: designs of one shape, up to `lanes` of them ------------------------------------------
    auto coordinate = [&]() {
        if (sharded || lanes_cap == 1) {
            wait_assembled(0, njobs);
            for (int q = 0; q < njobs; ++q) push_unit({q});
            return;
        }
        // enough units to occupy every context, no more lanes per unit than the shape allows; an explicit request never
        // exceeds what the shape allows
        auto lanes_per_unit = [&](int cap, int G) {
            int per = std::min(cap, std::max(1, (G + nctx - 1) / nctx));
            if (lanes_cap > 1) per = std::min(std::min(lanes_cap, cap), G);
            return std::max(1, per);
        };
        int handed = 0;                                            // jobs [0, handed) are in units already
        if (njobs > 0 && order != 1) {
            wait_assembled(0, 1);
            if (arc[0] == 0) {
                const int per = lanes_per_unit(int(keys[0].back()), njobs);
                while (handed < njobs) {
                    const int hi = std::min(njobs, handed + per);
                    wait_assembled(handed, hi);
                    bool same = true;
                    for (int q = handed; q < hi && same; ++q) same = arc[q] == 0 && keys[q] == keys[0];
                    if (!same) break;
                    std::vector<int> u;
                    for (int q = handed; q < hi; ++q) u.push_back(q);
                    push_unit(std::move(u));
                    if (trace && handed == 0) std::fprintf(stderr, "[batch] first unit handed out at %.2f ms\n", now_ms() - t0);
                    handed = hi;
                }
            }
        }
        wait_assembled(handed, njobs);
        if (trace) std::fprintf(stderr, "[batch] %d jobs assembled at %.2f ms\n", njobs, now_ms() - t0);
        std::map<std::vector<long>, std::vector<int>> groups;
        for (int q = handed; q < njobs; ++q) {
            if (arc[q] != 0) { push_unit({q}); continue; }                  // the single-design path reports the assembly error
            groups[keys[q]].push_back(q);
        }
        for (auto& g : groups) {
            const int G = int(g.second.size()), per = lanes_per_unit(int(g.first.back()), G);
            // Which designs share a unit: neighbours in the job list (a sweep's neighbours tend to need similar
            // iteration counts, so the lanes of a unit finish together) or, MBFIR_UNIT_ORDER=1, dealt round-robin over
            // the units (every unit gets its share of the slow designs: at the end of the batch all streams still
            // have live lanes instead of one stream running the slow unit alone)
            const int nunits = (G + per - 1) / per;
            if (order == 1 && nunits > 1) {
                std::vector<std::vector<int>> us(nunits);
                for (int i = 0; i < G; ++i) us[i % nunits].push_back(g.second[i]);
                for (auto& u : us) push_unit(u);
            } else {
                for (int i = 0; i < G; i += per) push_unit(std::vector<int>(g.second.begin() + i, g.second.begin() + std::min(G, i + per)));
            }
        }
    };
    auto work = [&](int c) {
        mbfir_ctx* ctx = ctxs[c];
        for (;;) {
            std::vector<int> U;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_units.wait(lk, [&] { return units_taken < units.size() || units_closed; });
                if (units_taken >= units.size()) break;
                U = units[units_taken++];
            }
            auto finish_job = [&](int q) {
                mbfir_job& J = jobs[q];
                if (J.rc < 0 && !J.err[0]) std::snprintf(J.err, sizeof(J.err), "%s", ctx->err.c_str());
                if (J.z && J.z_cap > 0 && J.rc >= 0) {
                    const std::vector<double>& x = ctx->last_x;
                    std::memcpy(J.z, x.data(), sizeof(double) * std::min<size_t>(x.size(), size_t(J.z_cap)));
                }
            };
            auto release_programs = [&]() {
                bool last;
                { std::lock_guard<std::mutex> lk(mu); ++units_done; last = units_closed && units_done == units.size(); }
                if (!last) { for (int q : U) progs[q] = TrigProgram(); return; }
                std::lock_guard<std::mutex> lk(g_parked_mu);
                for (int q : U) g_parked.push_back(std::move(progs[q]));
            };
            if (U.size() == 1) {
                jobs[U[0]].rc = one_job(ctx, jobs[U[0]], opts);
                finish_job(U[0]);
                release_programs();
                continue;
            }
            std::vector<char> done(U.size(), 0);         // designs of the unit whose result is already out (a later error must not overwrite it)
            try {
                const double t_asm = now_ms();
                std::vector<const TrigProgram*> Ps;
                for (int q : U) Ps.push_back(&progs[q]);
                std::vector<std::vector<double>> xs;
                std::vector<SolveInfo> sis;
                const SolveOpts so = to_opts(opts, progs[U[0]].which);
                ctx->solver->solve_lanes(Ps, so, xs, sis);
                const double t_solved = now_ms();
                std::vector<int> redo;
                for (size_t b = 0; b < U.size(); ++b) {
                    mbfir_job& J = jobs[U[b]];
                    const int rc = status_to_rc(sis[b].status);
                    if (rc == MBFIR_NUMERICAL || sis[b].status == ST_OPTIMAL_INACCURATE) { redo.push_back(U[b]); continue; }   // single path: retries in extended precision
                    if (rc == MBFIR_SOLVED) taps_from_solution(*ctx->solver, progs[U[b]], xs[b], int(b), J.h_re, J.h_im);
                    fill_info(&J.info, progs[U[b]], sis[b], rc, t0, t_asm, t_solved, now_ms());
                    J.rc = rc;
                    ctx->last_x = xs[b];
                    finish_job(U[b]);
                    done[b] = 1;
                }
                for (size_t b = 0; b < U.size(); ++b) if (!done[b]) done[b] = 2;      // (the redo list: their single solves report for themselves)
                for (int q : redo) { jobs[q].rc = one_job(ctx, jobs[q], opts); finish_job(q); }
            } catch (const ShapeError& e) {
                // the lock-step path does not take these programs as one unit: every design of it goes through the
                // single-design path, which reports its own verdict or error
                ctx->err = e.what();
                for (size_t b = 0; b < U.size(); ++b) if (!done[b]) { const int q = U[b]; jobs[q].rc = one_job(ctx, jobs[q], opts); finish_job(q); }
            } catch (const ResourceError& e) {
                // the unit did not get its device memory (lanes sized to the unit's maxima: ADVICE r4): recoverable -- one design
                // at a time needs a lane's worth
                ctx->err = e.what();
                for (size_t b = 0; b < U.size(); ++b) if (!done[b]) { const int q = U[b]; jobs[q].rc = one_job(ctx, jobs[q], opts); finish_job(q); }
            } catch (const std::exception& e) {
                // an internal or device error (a lost in-launch hand-off of the factorisation, a HIP error):
                // reported for every design of the unit that has no result yet, NOT retried -- the single-design path takes other
                // kernels and would hide a defect of the lock-step ones behind a slow success
                ctx->err = e.what();
                for (size_t b = 0; b < U.size(); ++b) {
                    if (done[b] == 1) continue;
                    const int q = U[b];
                    std::memset(&jobs[q].info, 0, sizeof(jobs[q].info));
                    jobs[q].info.status = MBFIR_E_HIP; jobs[q].info.lanes = int(U.size());
                    jobs[q].rc = MBFIR_E_HIP;
                    std::snprintf(jobs[q].err, sizeof(jobs[q].err), "lock-step unit of %d: %s", int(U.size()), e.what());
                }
            }
            release_programs();
        }
    };
    // a worker / assembler that throws (bad_alloc while parking programs, ...) must not take the process down
    // (std::terminate from a thread's entry function): the first message is kept and the call returns MBFIR_E_HIP
    std::string thread_err;
    auto guarded = [&](auto&& body) {
        try { body(); }
        catch (const std::exception& e) { std::lock_guard<std::mutex> lk(mu); if (thread_err.empty()) thread_err = e.what(); }
        catch (...) { std::lock_guard<std::mutex> lk(mu); if (thread_err.empty()) thread_err = "unknown exception in a batch thread"; }
    };
    std::vector<std::thread> th;
    const int nasm = std::max(1, std::min(njobs, std::min(16, int(std::thread::hardware_concurrency()))));
    const int nwork = std::max(1, std::min(nctx, njobs));
    // Thread creation can fail (EAGAIN: up to 16 + nctx threads per call): what was started is joined and the rest of the
    // work is done by fewer threads -- in the limit by the calling thread alone (assembly inline, then the units one context
    // after the other) -- instead of destroying joinable threads, which ends the process.
    int started_asm = 0, started_work = 0;
    try {
        for (int c = 0; c < nasm; ++c) { th.emplace_back([&] { guarded(asm_work); }); ++started_asm; }
        for (int c = 0; c < nwork; ++c) { th.emplace_back([&, c] { guarded([&] { work(c); }); }); ++started_work; }
    } catch (const std::system_error&) {}
    if (started_asm == 0) guarded(asm_work);
    std::string coord_err;
    try { coordinate(); } catch (const std::exception& e) { coord_err = e.what(); }
    { std::lock_guard<std::mutex> lk(mu); units_closed = true; }
    cv_units.notify_all();
    {
        std::vector<TrigProgram> old;
        { std::lock_guard<std::mutex> lk(g_parked_mu); old.swap(g_parked); }
    }                                                                // (the previous batch's last programs, freed while the units run)
    if (started_work == 0) guarded([&] { work(0); });
    for (auto& t : th) t.join();
    if (coord_err.empty() && !thread_err.empty()) coord_err = thread_err;
    if (trace) std::fprintf(stderr, "[batch] %zu units, all done at %.2f ms\n", units.size(), now_ms() - t0);
    if (!coord_err.empty()) {
        for (int q = 0; q < njobs; ++q) {                  // (host memory exhausted while forming units: nothing of the batch is reported)
            jobs[q].rc = MBFIR_E_HIP;
            std::snprintf(jobs[q].err, sizeof(jobs[q].err), "%s", coord_err.c_str());
        }
        return MBFIR_E_HIP;
    }
    int worst = 0;
    for (int q = 0; q < njobs; ++q)
        if (jobs[q].rc < worst) worst = jobs[q].rc;
    return worst;                                         // 0, or the most negative error code of the batch
}

// ---- device kernel test hooks ------------------------------------------------------------------
#define MBFIR_TRY(ctx, stmt)                   \
    if (!ctx) return MBFIR_E_ARG;              \
    try { stmt; return 0; }                    \
    catch (const std::exception& e) { ctx->err = e.what(); return MBFIR_E_HIP; }

int mbfir_test_gram(mbfir_ctx* ctx, int m, int nt, int nw, const double* A, const double* d, double* out) {
    MBFIR_TRY(ctx, ctx->solver->test_gram(m, nt, nw, A, d, out));
}
int mbfir_test_chol(mbfir_ctx* ctx, int n, const double* H, double* out_l, double* out_m) {
    MBFIR_TRY(ctx, ctx->solver->test_chol(n, H, out_l, out_m));
}
int mbfir_test_chol_lanes(mbfir_ctx* ctx, int n, int nlanes, int form, const int* mask, const double* H, double* out_l, double* out_m) {
    MBFIR_TRY(ctx, ctx->solver->test_chol_lanes(n, nlanes, form, mask, H, out_l, out_m));
}
int mbfir_test_ddsolve(mbfir_ctx* ctx, int n, int k, const double* H, const double* U, const double* X, int nrhs,
                       const double* bh, const double* bl, double* xh, double* xl, int* nfix, double* Lh, double* Ll) {
    if (!ctx || n < 1 || k < 0 || nrhs < 1 || nrhs > 2 || !H || !bh || !bl || !xh || !xl || !nfix) return MBFIR_E_ARG;
    MBFIR_TRY(ctx, ctx->solver->test_ddsolve(n, k, H, U, X, nrhs, bh, bl, xh, xl, nfix, Lh, Ll));
}
int mbfir_test_fold(const double* w, int m, int fold, long* out) {
    if (!w || m < 1 || !out) return MBFIR_E_ARG;
    try { Solver::test_fold(w, m, fold, out); } catch (const std::exception&) { return MBFIR_E_HIP; }
    return 0;
}
int mbfir_test_specfact(mbfir_ctx* ctx, int n, const double* x, double* h_re, double* h_im) {
    MBFIR_TRY(ctx, ctx->solver->test_specfact(n, x, h_re, h_im));
}
int mbfir_b2a(mbfir_ctx* ctx, int n, const double* b_re, const double* b_im, double* a_re, double* a_im) {
    if (!ctx || n < 1 || !b_re || !b_im || !a_re || !a_im) return MBFIR_E_ARG;
    MBFIR_TRY(ctx, ctx->solver->slr(n, b_re, b_im, nullptr, nullptr, a_re, a_im, nullptr, nullptr));
}
int mbfir_bloch(mbfir_ctx* ctx, int ntime, const double* b1_re, const double* b1_im, const double* gx, const double* gy,
                const double* gz, const double* tsteps, double t1, double t2, int nfreq, const double* df, int npos,
                const double* dx, const double* dy, const double* dz, int mode, double gamma, double* mx, double* my, double* mz) {
    if (!ctx || ntime < 1 || nfreq < 1 || npos < 1 || !b1_re || !b1_im || !tsteps || !df || !mx || !my || !mz || mode < 0 || mode > 3 ||
        !(t1 > 0) || !(t2 > 0))
        return MBFIR_E_ARG;
    MBFIR_TRY(ctx, ctx->solver->bloch(ntime, b1_re, b1_im, gx, gy, gz, tsteps, t1, t2, nfreq, df, npos, dx, dy, dz, mode, gamma, mx, my, mz));
}
int mbfir_ab2rf(mbfir_ctx* ctx, int n, const double* a_re, const double* a_im, const double* b_re, const double* b_im,
                double* rf_re, double* rf_im) {
    if (!ctx || n < 1 || n > 2048 || !a_re || !a_im || !b_re || !b_im || !rf_re || !rf_im) return MBFIR_E_ARG;
    MBFIR_TRY(ctx, ctx->solver->slr(n, b_re, b_im, a_re, a_im, nullptr, nullptr, rf_re, rf_im));
}
int mbfir_b2rf(mbfir_ctx* ctx, int n, const double* b_re, const double* b_im, double* rf_re, double* rf_im) {
    if (!ctx || n < 1 || n > 2048 || !b_re || !b_im || !rf_re || !rf_im) return MBFIR_E_ARG;
    MBFIR_TRY(ctx, ctx->solver->slr(n, b_re, b_im, nullptr, nullptr, nullptr, nullptr, rf_re, rf_im));
}
int mbfir_abr(mbfir_ctx* ctx, int n, const double* rf_re, const double* rf_im, const double* g, int nx, const double* x,
              int mode, double* a_re, double* a_im, double* b_re, double* b_im) {
    if (!ctx || n < 1 || nx < 1 || !rf_re || !rf_im || !x || !a_re || !a_im || !b_re || !b_im || (mode != 0 && mode != 1))
        return MBFIR_E_ARG;
    MBFIR_TRY(ctx, ctx->solver->abr(n, rf_re, rf_im, g, nx, x, mode, a_re, a_im, b_re, b_im));
}
int mbfir_test_mfma_peak(mbfir_ctx* ctx, double* tf_mfma, double* tf_valu) {
    MBFIR_TRY(ctx, ctx->solver->test_mfma_peak(tf_mfma, tf_valu));
}

int mbfir_test_time_kernels(mbfir_ctx* ctx, int n, int m, int nt, int reps, double* ms_chol, double* ms_gram) {
    MBFIR_TRY(ctx, ctx->solver->test_time_kernels(n, m, nt, reps, ms_chol, ms_gram));
}

}  // extern "C"
