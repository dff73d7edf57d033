"""Is the factorisation of [S 0; 0 I] bit-identical to that of S in its leading block?  (the capacitance matrix of a lane padded to its
unit's largest strong set).  S small-diagonal + low-rank, as the capacitance matrix is."""
import os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); warnings.filterwarnings("ignore")
import numpy as np, mbfir
rng = np.random.default_rng(5)
for n, pad in ((128, 192), (128, 256), (64, 128), (192, 256), (512, 576)):
    Y = rng.standard_normal((n, 300)) * np.exp(rng.uniform(-6, 2, (n, 1)))
    S = Y @ Y.T + np.diag(np.exp(rng.uniform(-20, -5, n)))
    Sp = np.eye(pad); Sp[:n, :n] = S
    L1, M1 = mbfir.test_chol(S)
    L2, M2 = mbfir.test_chol(Sp)
    Ls, Ms = mbfir.test_chol_lanes(np.stack([Sp, Sp, Sp]))
    print(n, pad, "L leading equal:", np.array_equal(np.tril(L1), np.tril(L2[:n, :n])), "M leading equal:", np.array_equal(np.tril(M1), np.tril(M2[:n, :n])),
          "| lanes form: L", np.array_equal(np.tril(L1), np.tril(Ls[0][:n, :n])), "M", np.array_equal(np.tril(M1), np.tril(Ms[0][:n, :n])),
          "| pad block identity:", np.array_equal(np.tril(M2[n:, n:]), np.eye(pad - n)), "cross zero:", not np.any(M2[n:, :n]))
