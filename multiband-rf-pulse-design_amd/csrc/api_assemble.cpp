// Host-only introspection entry points of include/mbfir.h (no HIP calls):
// expose the structured program produced by assemble.cpp so that the CPU test
// suite can expand it and compare with the oracle's dense (c, G, h).
#include "../../include/mbfir.h"
#include "program.h"
#include <cstring>

using namespace mbfir;

struct mbfir_program { TrigProgram P; };

extern "C" {

int mbfir_assemble(int which, int n, int nband, const double* f, const double* a, const double* d,
                   const double* params, int grid_m, mbfir_program** out, char* err, int errlen) {
    std::string e;
    mbfir_program* p = new mbfir_program();
    int rc = -1;
    switch (which) {
        case DES_AP: rc = assemble_ap(n, nband, f, a, d, params[0], params[1], grid_m, p->P, e); break;
        case DES_QP: rc = assemble_qp(n, nband, f, a, d, params[0], params + 1, int(params[3]), grid_m, p->P, e); break;
        case DES_LINPROG: rc = assemble_linprog(n, nband, f, a, d, grid_m, p->P, e); break;
        case DES_QPROG_PHS: {
            std::vector<double> are(2 * nband), aim(2 * nband), dre(nband), dim(nband);
            for (int i = 0; i < 2 * nband; ++i) { are[i] = a[2 * i]; aim[i] = a[2 * i + 1]; }
            for (int i = 0; i < nband; ++i) { dre[i] = d[2 * i]; dim[i] = d[2 * i + 1]; }
            rc = assemble_qprog_phs(n, nband, f, are.data(), aim.data(), dre.data(), dim.data(), grid_m, p->P, e);
            break;
        }
        default: e = "unknown designer";
    }
    if (err && errlen > 0) { std::strncpy(err, e.c_str(), errlen - 1); err[errlen - 1] = 0; }
    if (rc != 0) { delete p; *out = nullptr; return rc; }
    *out = p;
    return 0;
}

void mbfir_program_free(mbfir_program* p) { delete p; }

int mbfir_program_shard(const mbfir_program* p, int rank, int size, mbfir_program** out) {
    if (!p || !out || size < 1 || rank < 0 || rank >= size) return -1;
    mbfir_program* q = new mbfir_program();
    q->P = shard_program(p->P, rank, size);
    *out = q;
    return 0;
}

void mbfir_program_dims(const mbfir_program* p, int* dims) {
    const TrigProgram& P = p->P;
    int nnz_id = 0;
    for (int c : P.col) nnz_id += c >= 0;
    dims[0] = P.Nt; dims[1] = P.Ne; dims[2] = P.R; dims[3] = P.l; dims[4] = P.nq3; dims[5] = P.big;
    dims[6] = P.Mf; dims[7] = P.quad ? 1 : 0; dims[8] = nnz_id; dims[9] = 0;
}

void mbfir_program_trig(const mbfir_program* p, double* w, int* col_kind, double* col_tau,
                        double* col_scale, int* pcol, double* psign, double* c) {
    const TrigProgram& P = p->P;
    std::memcpy(w, P.w.data(), sizeof(double) * P.Mf);
    std::memcpy(col_kind, P.col_kind.data(), sizeof(int) * P.Nt);
    std::memcpy(col_tau, P.col_tau.data(), sizeof(double) * P.Nt);
    std::memcpy(col_scale, P.col_scale.data(), sizeof(double) * P.Nt);
    std::memcpy(pcol, P.pcol.data(), sizeof(int) * P.Nt);
    std::memcpy(psign, P.psign.data(), sizeof(double) * P.Nt);
    std::memcpy(c, P.c.data(), sizeof(double) * P.N());
}

void mbfir_program_rows(const mbfir_program* p, int* freq, int* col, double* alpha, double* beta,
                        double* ey, double* h) {
    const TrigProgram& P = p->P;
    std::memcpy(freq, P.freq.data(), sizeof(int) * P.R);
    std::memcpy(col, P.col.data(), sizeof(int) * P.R);
    std::memcpy(alpha, P.alpha.data(), sizeof(double) * P.R);
    std::memcpy(beta, P.beta.data(), sizeof(double) * P.R);
    std::memcpy(ey, P.ey.data(), sizeof(double) * 3 * P.R);
    std::memcpy(h, P.h.data(), sizeof(double) * P.R);
}

void mbfir_program_replicated(const mbfir_program* p, int* rep) {
    const std::vector<int> r = replicated_rows(p->P);
    std::memcpy(rep, r.data(), sizeof(int) * r.size());
}

}  // extern "C"
