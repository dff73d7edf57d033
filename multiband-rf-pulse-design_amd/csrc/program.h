// Structured conic program consumed by the HIP interior-point solver.
//
//   minimise c'z   s.t.  G z + s = h,  s in  R_+^l  x  Q_3^nq3  x  Q_big
//   z = [x (Nt unknowns multiplied by trig rows) ; y (Ne <= 3 scalar unknowns)]
//
// G is never stored densely.  Every row is one of
//   trig row      g = alpha * r1(w_i) + beta * r2(w_i)         (+ ey on the y block)
//   identity row  g = alpha * e_col                            (+ ey on the y block)
// where r1(w)[j] = scale_j * cos|sin(w * tau_j) is one row of the materialised
// frequency matrix A1 (Mf x Nt, built once per design on the device) and
// r2 = P r1 is its quadrature partner, a signed permutation of r1
// (r2[j] = psign[j] * r1[pcol[j]]), so only A1 is ever read.
#pragma once
#include <memory>
#include <vector>
#include <string>

namespace mbfir {

enum Designer { DES_AP = 0, DES_QP = 1, DES_LINPROG = 2, DES_QPROG_PHS = 3 };

struct TrigProgram {
    int which = 0;
    int n = 0;                       // taps
    int Nt = 0, Ne = 0;              // trig unknowns, extra scalar unknowns
    int Mf = 0;                      // frequency rows of A1
    bool quad = false;               // any beta != 0 (needs r2)
    std::vector<double> w;           // Mf
    std::vector<int> col_kind;       // Nt : 0 = cos, 1 = sin
    std::vector<double> col_tau;     // Nt
    std::vector<double> col_scale;   // Nt
    std::vector<int> pcol;           // Nt
    std::vector<double> psign;       // Nt
    int R = 0, l = 0, nq3 = 0, big = 0;
    std::vector<int> freq;           // R : frequency index or -1
    std::vector<int> col;            // R : identity column or -1
    std::vector<double> alpha, beta; // R
    std::vector<double> ey;          // R*3
    std::vector<double> h;           // R
    std::vector<double> c;           // Nt+Ne
    // post-processing
    int nhalf = 0;                   // linprog
    bool real_filter = false, odd_filter = false;
    int N() const { return Nt + Ne; }
    // host-side structures the solver derives from the arrays above (CSR maps, lattice analysis): computed once by
    // Solver::shape_key -- which mbfir_solve_batch calls from its parallel assembly threads -- and reused by the solve
    mutable std::shared_ptr<void> prep;
    void add_row(int fr, int cl, double al, double be, double e0, double e1, double e2, double hh) {
        freq.push_back(fr); col.push_back(cl); alpha.push_back(al); beta.push_back(be);
        ey.push_back(e0); ey.push_back(e1); ey.push_back(e2); h.push_back(hh);
    }
};

// Row sharding for one process per GPU: rank r keeps the frequencies i with
// i % size == r (interleaved, so every rank gets the same band/transition mix) together with every
// row / cone attached to them; rows and cones with no frequency (identity rows, spike / per-tap
// cones, the big cone) are REPLICATED on every rank (x and y are replicated, so every rank computes
// them identically and can factorise the whole normal matrix itself); sums over the rows count them
// on rank 0 only.  The union of the shards' frequency rows is the original program's, none appears twice.
TrigProgram shard_program(const TrigProgram& Q, int rank, int size);
std::vector<int> replicated_rows(const TrigProgram& Q);     // 1: the row is one of those a row-sharded solve holds on every rank

// Return 0 ok, 3 early-fail (reference returns 'Failed' before solving), -1 argument error.
int assemble_ap(int n, int nband, const double* f, const double* a, const double* d,
                double obj, double peak, int grid_m, TrigProgram& P, std::string& err);
int assemble_qp(int n, int nband, const double* f, const double* a, const double* d,
                double kquad, const double* obj, int nobj, int grid_m, TrigProgram& P, std::string& err);
int assemble_linprog(int n, int nband, const double* f, const double* a, const double* d,
                     int grid_m, TrigProgram& P, std::string& err);
int assemble_qprog_phs(int n, int nband, const double* f, const double* ac_re, const double* ac_im,
                       const double* dc_re, const double* dc_im, int grid_m, TrigProgram& P,
                       std::string& err);

}  // namespace mbfir
