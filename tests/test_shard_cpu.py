"""CPU tests of the multi-GPU (row-sharded) path: the shard partition of the structured program and the
torch.distributed all-reduce hook, exercised with world_size 2 on the gloo backend."""
import ctypes
import os
import socket
import sys

import numpy as np
import pytest
from conftest import CASES, ROOT, WHICH

import mbfir


def _params(fn, args):
    if fn == "fir_ap_cvx":
        return args[4:6]
    if fn == "fir_qp_cvx":
        obj = np.atleast_1d(args[5]).astype(float)
        return [args[4]] + list(obj) + [0.0] * (2 - len(obj)) + [len(obj)]
    return [0.0]


@pytest.mark.parametrize("name", ["ap_c13_58", "qp_modelB25", "lin_cplx31", "qphs22"])
@pytest.mark.parametrize("size", [2, 3, 8])
def test_shards_partition_the_rows(name, size):
    fn, args = CASES[name]
    rc, full = mbfir.assemble_dense(WHICH[fn], args[0], args[1], args[2], args[3], _params(fn, args))
    assert rc == 0
    N = full["G"].shape[1]
    HtH = np.zeros((N, N))
    gth = np.zeros(N)
    rows = 0
    cones = 0
    freqs = 0
    nrep = int(full["rep"].sum())
    rep_cones = int(full["rep"][:full["l"]].sum()) + int(full["rep"][full["l"]:full["l"] + 3 * full["nq3"]].sum()) // 3 + (1 if full["big"] else 0)
    for r in range(size):
        rc, S = mbfir.assemble_dense(WHICH[fn], args[0], args[1], args[2], args[3], _params(fn, args), shard=(r, size))
        assert rc == 0
        assert np.array_equal(S["c"], full["c"])               # x, y (and c) are replicated
        # rows / cones WITHOUT a frequency (identity rows, spike / per-tap cones, the big cone) are replicated on every rank --
        # the same rows, in the same order -- so that every rank can assemble and factorise the whole normal matrix (round 4);
        # sums over the rows count them once (on rank 0)
        own = np.ones(S["G"].shape[0]) if r == 0 else 1.0 - S["rep"]
        assert int(S["rep"].sum()) == nrep and S["big"] == full["big"]
        assert np.array_equal(S["G"][S["rep"] == 1], full["G"][full["rep"] == 1]) and np.array_equal(S["h"][S["rep"] == 1], full["h"][full["rep"] == 1])
        HtH += S["G"].T @ (own[:, None] * S["G"])
        gth += S["G"].T @ (own * S["h"])
        rows += int(own.sum())
        cones += S["l"] + S["nq3"] + (1 if S["big"] else 0) - (rep_cones if r > 0 else 0)
        freqs += S["Mf"]
        assert abs(S["Mf"] - full["Mf"] / size) <= 2                                # balanced: folded +w / -w PAIRS are dealt in turn
        # both partners of a folded pair sit on the same rank (one lattice recurrence serves them): every negative frequency of
        # the shard finds its mirror image in the shard whenever the full grid holds it
        # (duplicated grid points -- a band edge on a linspace sample, the two ends -pi / +pi -- leave a few without a partner)
        wf, ws = np.sort(full["w"]), np.sort(S["w"])
        lonely = sum(1 for w in ws[ws < 0] if np.abs(wf + w).min() <= 4e-15 * np.pi and np.abs(ws + w).min() > 4e-15 * np.pi)
        assert lonely <= 4, lonely
    assert rows == full["G"].shape[0] and freqs == full["Mf"]
    assert cones == full["l"] + full["nq3"] + (1 if full["big"] else 0)
    # the sum over shards of G'G and G'h (replicated rows counted once) is the full one: what the per-iteration all-reduce relies on
    assert np.abs(HtH - full["G"].T @ full["G"]).max() <= 1e-10 * np.abs(HtH).max()
    assert np.abs(gth - full["G"].T @ full["h"]).max() <= 1e-10 * max(1.0, np.abs(gth).max())


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, size, port, name, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    import mbfir as mb
    from conftest import CASES as CS, WHICH as WH
    dist.init_process_group("gloo", rank=rank, world_size=size)

    def wrap(ptr, count):          # host memory stands in for the solver's device buffer
        arr = np.ctypeslib.as_array((ctypes.c_double * count).from_address(ptr))
        return torch.from_numpy(arr)

    hook = mb.make_torch_allreduce(wrap=wrap)
    fn, args = CS[name]
    rc, S = mb.assemble_dense(WH[fn], args[0], args[1], args[2], args[3], _params(fn, args), shard=(rank, size))
    d = 1.0 / (1.0 + np.arange(S["G"].shape[0]) % 7)             # some positive row weights
    if rank > 0:
        d = d * (1.0 - S["rep"])                                 # replicated rows count on rank 0 only
    H = np.ascontiguousarray(S["G"].T @ (d[:, None] * S["G"]))   # this rank's share of G' D G
    g = np.ascontiguousarray(S["G"].T @ (d * S["h"]))
    m = np.array([float(rank), -float(rank)])
    assert hook(H.ctypes.data, H.size, 0) == 0                   # sum, in place
    assert hook(g.ctypes.data, g.size, 0) == 0
    assert hook(m.ctypes.data, m.size, 1) == 0                   # max
    q.put((rank, H, g, m))
    dist.barrier()
    dist.destroy_process_group()


def test_allreduce_hook_world_size_2_gloo():
    """Two processes, gloo: each holds one row shard, the hook all-reduces its Gram share, its gradient
    share and a max -- the three reductions a row-sharded IPM iteration performs."""
    import torch.multiprocessing as mp
    name, size = "ap_c13_58", 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, size, port, name, q)) for r in range(size)]
    for p in procs:
        p.start()
    out = [q.get(timeout=120) for _ in range(size)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    fn, args = CASES[name]
    rc, full = mbfir.assemble_dense(WHICH[fn], args[0], args[1], args[2], args[3], _params(fn, args))
    # the weights depend on the local row index, so rebuild the expected sum shard by shard
    Hexp = np.zeros_like(out[0][1])
    gexp = np.zeros_like(out[0][2])
    for r in range(size):
        rc, S = mbfir.assemble_dense(WHICH[fn], args[0], args[1], args[2], args[3], _params(fn, args), shard=(r, size))
        d = 1.0 / (1.0 + np.arange(S["G"].shape[0]) % 7)
        if r > 0:
            d = d * (1.0 - S["rep"])
        Hexp += S["G"].T @ (d[:, None] * S["G"])
        gexp += S["G"].T @ (d * S["h"])
    for rank, H, g, m in out:
        assert np.abs(H - Hexp).max() <= 1e-12 * np.abs(Hexp).max()
        assert np.abs(g - gexp).max() <= 1e-12 * max(1.0, np.abs(gexp).max())
        assert list(m) == [1.0, 0.0]
    assert np.array_equal(out[0][1], out[1][1])        # bit-identical on both ranks: replicated state stays in step
