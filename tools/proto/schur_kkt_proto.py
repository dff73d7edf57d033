import sys, warnings, time, os
sys.path.insert(0, '/root/repo'); warnings.filterwarnings("ignore")
import numpy as np
import importlib.util
sp = importlib.util.spec_from_file_location("spec", "/root/repo/multiband-rf-pulse-design_amd/spec.py"); spec = importlib.util.module_from_spec(sp); sp.loader.exec_module(spec)
from oracle import assemble
THETA = float(os.environ.get("THETA", "1e8"))
NREF = int(os.environ.get("NREF", "3"))
src = open('/root/repo/oracle/conic_ipm.py').read()
src = src.replace('''    def kkt_solve(Wm, H, cf, bx, bz):''', '''    def kkt_solve(Wm, H, cf, bx, bz):
        if Wm is None or STRONG.get("id") is not Wm or (STRONG.get('plain') is not None and STRONG['plain'][0] is Wm):
            return kkt_solve_ne(Wm, H, cf, bx, bz)
        st = STRONG
        U, lamp, Ep, M, C_M, winv2_w, sc = st["U"], st["lamp"], st["Ep"], st["M"], st["C_M"], st["winv2_w"], st["cones"]
        vec = bx.ndim == 1
        BX = bx[:, None] if vec else bx
        BZ = bz[:, None] if vec else bz
        def Hw_solve(b): return M.T @ (M @ b)
        def ebz(V):                                    # e+ . V_i for the strong cones
            Vq = V[cone.o3:cone.ob].reshape(cone.nq3, 3, -1)[sc]
            return np.einsum("ka,kan->kn", Ep, Vq)
        def solve_block(r1, r2):                       # [Hw U'; U -1/lam][dx; zeta] = [r1; r2]
            y = Hw_solve(r1)
            zeta = C_M.T @ (C_M @ (U @ y - r2))
            dx = Hw_solve(r1 - U.T @ zeta)
            return dx, zeta
        def spread(zeta):                              # strong part of dz in row coordinates
            o = np.zeros_like(BZ)
            oq = o[cone.o3:cone.ob].reshape(cone.nq3, 3, -1)
            oq[sc] = Ep[:, :, None] * zeta[:, None, :]
            return o
        wbz = winv2_w(BZ)
        b2 = ebz(BZ)
        r1 = BX + G.T @ wbz
        dx, zeta = solve_block(r1, b2)
        norms = []
        for _ in range(NREF + 1):
            Gdx = G @ dx
            dzw = winv2_w(Gdx) - wbz
            rho1 = BX - G.T @ dzw - U.T @ zeta
            rho2 = b2 - (U @ dx - zeta / lamp[:, None])
            norms.append(float(np.max(np.sqrt(np.sum(rho1 * rho1, axis=0)))))
            r2n = float(np.abs(rho2).max()) / max(1.0, float(np.abs(b2).max()))
            if _ == NREF or (norms[-1] <= REFTOL * nrm_c and r2n <= 1e-14): break
            ddx, dzeta = solve_block(rho1, rho2)
            dx = dx + ddx; zeta = zeta + dzeta
        if not (norms[-1] <= max(REFTOL * nrm_c, 1e-3 * norms[0]) and r2n <= 1e-10):
            FALLBACK[0] += 1
            if STRONG.get("plain") is None or STRONG["plain"][0] is not Wm:
                STRONG["plain"] = (Wm,) + tuple(factor_plain(Wm))
            nsweep[0] = MAX_SWEEPS
            return kkt_solve_ne(Wm, STRONG["plain"][1], STRONG["plain"][2], bx, bz)
        dz = dzw + spread(zeta)
        if os.environ.get("CHECK"):
            # reference: dense LU of the augmented system [0 G'; G -W^2]
            W2 = np.zeros((R, R))
            W2[np.arange(cone.l), np.arange(cone.l)] = 1.0 / Wm.dl
            Jm = np.diag([1.0, -1.0, -1.0])
            for q in range(cone.nq3):
                o = cone.o3 + 3 * q; wb = Wm.wb3[q]
                W2[o:o + 3, o:o + 3] = Wm.eta3[q] ** 2 * (2 * np.outer(wb, wb) - Jm)
            if cone.big:
                wb = Wm.wbb; Jb = -np.eye(cone.big); Jb[0, 0] = 1.0
                W2[cone.ob:, cone.ob:] = Wm.etab ** 2 * (2 * np.outer(wb, wb) - Jb)
            K = np.block([[np.zeros((N, N)), G.T], [G, -W2]])
            sol = np.linalg.solve(K, np.concatenate([BX, BZ], 0))
            for _r in range(2):
                rr = np.concatenate([BX, BZ], 0) - K @ sol
                sol = sol + np.linalg.solve(K, rr)
            dxr, dzr = sol[:N], sol[N:]
            dxb, dzb, _g = kkt_solve_ne(Wm, *factor_plain(Wm)[:2], bx, bz)
            STRONG["id"] = Wm
            dxb = dxb[:, None] if vec else dxb; dzb = dzb[:, None] if vec else dzb
            rel = lambda a_, b_: np.abs(a_ - b_).max(axis=0) / np.abs(b_).max(axis=0)
            print("      CHECK strong: dx", rel(dx, dxr), "dz", rel(dz, dzr), "| plain: dx", rel(dxb, dxr), "dz", rel(dzb, dzr))
        sweep_log.append([norms[0], norms[-1]])
        STRONG["res"] = (norms[0], norms[-1], float(np.abs(rho2).max()))
        return (dx[:, 0], dz[:, 0], Gdx[:, 0]) if vec else (dx, dz, Gdx)

    def kkt_solve_ne(Wm, H, cf, bx, bz):''')
src = src.replace('''    def factor(Wm):
        H = G.T @ (Wm.inv2(G) if Wm is not None else G)''', '''    ELIG = (np.count_nonzero(G[cone.o3:cone.ob].reshape(cone.nq3, 3, -1)[:, 1, :], axis=1) > 2) if cone.nq3 else None
    def factor_plain(Wm):
        STRONG["id"] = None
        NSTRONG[0] = 0
        H = G.T @ Wm.inv2(G)
        H = 0.5 * (H + H.T)
        L, nfix = chol_piv(H)
        chol_fixes[0] += nfix
        return H, np.linalg.inv(L)
    def factor(Wm):
        STRONG["id"] = None
        NSTRONG[0] = 0
        if Wm is not None and cone.nq3:
            w0 = Wm.wb3[:, 0]; n1 = np.sqrt(np.maximum(w0 * w0 - 1.0, 0.0))
            e2 = 1.0 / Wm.eta3 ** 2
            kap = (w0 + n1) ** 2
            lam_all = np.concatenate([Wm.dl, e2])      # LP weights and the middle eigenvalue of every Q3 cone
            ref = 2.0 ** np.median(np.floor(np.log2(lam_all)))
            strong = (e2 * kap >= THETA * ref) & (n1 > 0) & ELIG
            if strong.any():
                sc = np.nonzero(strong)[0]
                what = -Wm.wb3[sc, 1:] / n1[sc, None]                  # direction of the vector part of J wbar
                Ep = np.concatenate([np.ones((len(sc), 1)), what], 1) / np.sqrt(2.0)
                Em = np.concatenate([np.ones((len(sc), 1)), -what], 1) / np.sqrt(2.0)
                Eo = np.concatenate([np.zeros((len(sc), 1)), -what[:, 1:2], what[:, 0:1]], 1)
                lamp = e2[sc] * kap[sc]; lamm = e2[sc] / kap[sc]; lamo = e2[sc]
                def winv2_w(V):
                    vec = V.ndim == 1
                    if vec: V = V[:, None]
                    o = Wm.inv2(V)
                    Vq = V[cone.o3:cone.ob].reshape(cone.nq3, 3, -1)[sc]
                    am = np.einsum("ka,kan->kn", Em, Vq); ao = np.einsum("ka,kan->kn", Eo, Vq)
                    oq = o[cone.o3:cone.ob].reshape(cone.nq3, 3, -1)
                    oq[sc] = (lamm[:, None] * am)[:, None, :] * Em[:, :, None] + (lamo[:, None] * ao)[:, None, :] * Eo[:, :, None]
                    return o[:, 0] if vec else o
                Gq = G[cone.o3:cone.ob].reshape(cone.nq3, 3, -1)[sc]
                U = np.einsum("ka,kan->kn", Ep, Gq)
                Hw = G.T @ winv2_w(G)
                Hw = 0.5 * (Hw + Hw.T)
                L, nfix = chol_piv(Hw)
                chol_fixes[0] += nfix
                M = np.linalg.inv(L)
                Y = M @ U.T
                Cm = np.diag(1.0 / lamp) + Y.T @ Y
                Lc, nfc = chol_piv(0.5 * (Cm + Cm.T))
                if nfc and os.environ.get("NOFC"):
                    FALLBACK[0] += 1
                    nsweep[0] = MAX_SWEEPS
                    return factor_plain(Wm)
                STRONG.update(id=Wm, cones=sc, U=U, lamp=lamp, Ep=Ep, M=M, C_M=np.linalg.inv(Lc), winv2_w=winv2_w, nfc=nfc)
                NSTRONG[0] = len(sc)
                return Hw, M
        H = G.T @ (Wm.inv2(G) if Wm is not None else G)''')
src = src.replace("def solve(c, G, h, l,", "import os\nSTRONG = {}\nFALLBACK = [0]\nNSTRONG = [0]\nTHETA = %g\nNREF = %d\ndef solve(c, G, h, l," % (THETA, NREF))
src = src.replace("        nsweep[0] = next_sweeps(sweep_log, nsweep[0], REFTOL * nrm_c)", "        if os.environ.get('VERB') and (it % int(os.environ['VERB']) == 0): print('   it', it, 'strong', NSTRONG[0], 'fixes', chol_fixes[0], 'gap %.2e pres %.1e dres %.1e' % (relgap, pres, dres), 'res', [('%.0e' % nl[0], '%.0e' % nl[-1]) for nl in sweep_log], 'alpha %.3f' % alpha, 'rho2 %.1e nfc %d' % (STRONG.get('res', (0,0,0))[2], STRONG.get('nfc', -1)), flush=True)\n        nsweep[0] = next_sweeps(sweep_log, nsweep[0], REFTOL * nrm_c) if not NSTRONG[0] else nsweep[0]")
mod = type(sys)("ipms3"); mod.__dict__["__name__"] = "oracle.conic_ipm_s3"
exec(compile(src, "conic_ipm_s3", "exec"), mod.__dict__)
if __name__ == "__main__":
    n, m = int(sys.argv[1]), int(sys.argv[2])
    prob = os.environ.get("PROB", "h1qp")
    if prob == "h1qp":
        f, a, d = spec.spec_h1_dualband(n)
        P = assemble.assemble_fir_qp_cvx(n, f, a, d, 120.0, 1e6, m)
    elif prob == "h1ap":
        f, a, d = spec.spec_h1_dualband(n)
        P = assemble.assemble_fir_ap_cvx(n, f, a, d, 0.1, 1e-3, m)
    else:
        f, a, d = spec.spec_c13_bssfp(n)
        P = assemble.assemble_fir_ap_cvx(n, f, a, d, 0.1, 1e-3, m)
    t0 = time.time()
    r = mod.solve(P["c"], P["G"], P["h"], P["l"], P["nq3"], P["big"])
    print("fallbacks", mod.FALLBACK[0]); print("theta %g nref %d: status %d iters %d pcost %.10e relgap %.1e pres %.1e dres %.1e fixes %d time %.1f" % (THETA, NREF, r["status"], r["iters"], r["pcost"], r["relgap"], r["pres"], r["dres"], r["chol_fixes"], time.time() - t0))
