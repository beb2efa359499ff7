"""The chunked (nested-dissection) order of the banded Cholesky solve, checked on the CPU: the plan the library builds on the
host (csrc/chol_nd.hip, `mqs_sba_solve_plan_dump`) is replayed with numpy, block operation by block operation as the kernels
apply it -- eager updates inside a front, deferred ("lazy") sums for the tiles two fronts share, the three substitution
passes -- and has to reproduce numpy's solve.  The replay also checks what makes the GPU result reproducible: within one
launch no tile is written twice, and nothing a launch reads as panel input is written by it.

The GPU kernels themselves are compared with the same numpy solve in tests/test_ba_files.py (`-m gpu`)."""
import ctypes

import numpy as np
import pytest


NB = 32


def _plan(lib, n, hb, parts):
    need = lib.mqs_sba_solve_plan_dump(n, hb, parts, None, 0)
    if need == 0:
        return None
    buf = np.zeros(need, dtype=np.int32)
    got = lib.mqs_sba_solve_plan_dump(n, hb, parts, buf.ctypes.data_as(ctypes.c_void_p), need)
    assert got == need
    h = buf[:8]
    nst, ndesc, nlazy, ncontrib, nvec, ncols, nparts = (int(v) for v in h[:7])
    o = 8
    stages = buf[o:o + 8 * nst].reshape(nst, 8); o += 8 * nst
    descs = buf[o:o + 32 * ndesc].reshape(ndesc, 32); o += 32 * ndesc
    lazy = buf[o:o + 8 * nlazy].reshape(nlazy, 8); o += 8 * nlazy
    contrib = buf[o:o + ncontrib]; o += ncontrib
    vecs = buf[o:o + 4 * nvec].reshape(nvec, 4); o += 4 * nvec
    cols = buf[o:o + ncols]; o += ncols
    assert o == need
    return dict(stages=stages, descs=descs, lazy=lazy, contrib=contrib, vecs=vecs, cols=cols, parts=nparts)


def _banded_spd(n, hb, seed):
    rng = np.random.default_rng(seed)
    S = np.zeros((n, n))
    for d in range(1, hb + 1):
        v = rng.standard_normal(n - d)
        S[np.arange(n - d), np.arange(d, n)] = v
    S = S + S.T
    S[np.arange(n), np.arange(n)] = np.abs(S).sum(axis=1) + 1.0 + rng.random(n)
    return S


def _replay(S, b, plan):
    n = S.shape[0]
    nblk = (n + NB - 1) // NB
    N = nblk * NB
    A = np.eye(N)
    A[:n, :n] = S
    x = np.zeros(N)
    x[:n] = b
    sl = lambda k: slice(k * NB, (k + 1) * NB)
    Linv = {}
    factored = []

    def factor(k):
        T = A[sl(k), sl(k)]
        L = np.linalg.cholesky(np.tril(T) + np.tril(T, -1).T)
        A[sl(k), sl(k)] = L                      # the inverse lives in the upper triangle on the GPU; kept aside here
        Linv[k] = np.linalg.inv(L)
        factored.append(k)

    for st in plan["stages"]:
        nf, nsteps, doff, loff, nlazy, voff, nvec, maxlen = (int(v) for v in st)
        # ---- lazy launch ----
        written = set()
        for T in plan["lazy"][loff:loff + nlazy]:
            x1, x2, start, count, fac = (int(v) for v in T[:5])
            acc = np.zeros((NB, NB))
            for k in plan["contrib"][start:start + count]:
                assert int(k) in Linv                                   # a finished block column
                acc += A[sl(x1), sl(int(k))] @ A[sl(x2), sl(int(k))].T
            assert (x1, x2) not in written
            written.add((x1, x2))
            A[sl(x1), sl(x2)] -= acc
            if x1 != x2:
                A[sl(x2), sl(x1)] = A[sl(x1), sl(x2)].T
            if fac:
                assert x1 == x2
                factor(x1)
        # ---- steps ----
        for t in range(nsteps):
            written, panel_in = set(), set()
            todo = []
            for f in range(nf):
                d = plan["descs"][doff + t * nf + f]
                k, cnt, eager, tiles = (int(v) for v in d[:4])
                if k < 0 or cnt == 0:
                    continue
                assert k in Linv, "pivot block not factored before its step"
                blk = [int(v) for v in d[4:4 + cnt]]
                ecols = max(eager, 1)
                assert tiles == sum(cnt - bj for bj in range(ecols))
                X = [A[sl(k), sl(X_)].T @ Linv[k].T for X_ in blk]      # panel input read from the mirror
                for X_ in blk:
                    panel_in.add((k, X_))
                todo.append((k, blk, eager, X))
            for k, blk, eager, X in todo:
                for bi, X_ in enumerate(blk):
                    assert (X_, k) not in written
                    written.add((X_, k))
                    A[sl(X_), sl(k)] = X[bi]
                for bj in range(eager):
                    for bi in range(bj, len(blk)):
                        key = (blk[bi], blk[bj])
                        assert key not in written and (key[1], key[0]) not in written
                        written.add(key)
                        if bi != bj:
                            written.add((key[1], key[0]))
                        A[sl(blk[bi]), sl(blk[bj])] -= X[bi] @ X[bj].T
                        if bi != bj:
                            A[sl(blk[bj]), sl(blk[bi])] = A[sl(blk[bi]), sl(blk[bj])].T
                if eager > 0:
                    assert blk[0] == k + 1
                    factor(blk[0])
            assert not (written & panel_in), "a launch writes a tile it also reads as panel input"
    assert sorted(factored) == list(range(nblk))
    # ---- forward ----
    for st in plan["stages"]:
        nf, nsteps, doff, loff, nlazy, voff, nvec, maxlen = (int(v) for v in st)
        for V in plan["vecs"][voff:voff + nvec]:
            X_, start, count = int(V[0]), int(V[1]), int(V[2])
            s = np.zeros(NB)
            for k in plan["cols"][start:start + count]:
                s += A[sl(X_), sl(int(k))] @ x[sl(int(k))]
            x[sl(X_)] -= s
        for f in range(nf):
            for t in range(nsteps):
                d = plan["descs"][doff + t * nf + f]
                k, cnt, eager = int(d[0]), int(d[1]), int(d[2])
                if k < 0:
                    break
                x[sl(k)] = Linv[k] @ x[sl(k)]
                for u in range(eager):
                    X_ = int(d[4 + u])
                    x[sl(X_)] -= A[sl(X_), sl(k)] @ x[sl(k)]
    # ---- backward ----
    for st in plan["stages"][::-1]:
        nf, nsteps, doff, loff, nlazy, voff, nvec, maxlen = (int(v) for v in st)
        for f in range(nf):
            ks = [plan["descs"][doff + t * nf + f] for t in range(nsteps)]
            for d in ks[::-1]:
                k, cnt = int(d[0]), int(d[1])
                if k < 0:
                    continue
                tsum = np.zeros(NB)
                for X_ in d[4:4 + cnt]:
                    tsum += A[sl(int(X_)), sl(k)].T @ x[sl(int(X_))]
                x[sl(k)] = Linv[k].T @ (x[sl(k)] - tsum)
    return x[:n]


SHAPES = [(5286, 101, 0), (5286, 101, 2), (5286, 101, 16), (1200, 35, 0), (32 * 40 + 7, 64, 4), (900, 17, 0), (2048, 200, 2),
          (3000, 1, 0)]


@pytest.mark.parametrize("n,hb,parts", SHAPES)
def test_plan_replay_solves_the_system(mqs, n, hb, parts):
    lib = mqs._lib.lib()
    plan = _plan(lib, n, hb, parts)
    assert plan is not None, "the cut should apply to this shape"
    S = _banded_spd(n, hb, seed=n + hb)
    b = np.random.default_rng(1).standard_normal(n)
    x = _replay(S, b, plan)
    ref = np.linalg.solve(S, b)
    assert np.abs(x - ref).max() <= 1e-11 * max(1.0, np.abs(ref).max())
    # the cut shortens the chain of dependent launches
    steps = sum(int(s[1]) for s in plan["stages"])
    assert steps < (n + NB - 1) // NB
    if parts > 1:
        assert plan["parts"] <= parts


def _edge_shapes():
    rng = np.random.default_rng(2024)
    shapes = [(32 * 20, 31, 0), (32 * 20, 32, 0), (32 * 20 + 1, 33, 0), (32 * 20 - 1, 1, 4), (32 * 41 + 31, 64, 0), (32 * 41 + 1, 65, 8),
              (1116, 35, 0), (1116, 101, 0)]
    for _ in range(10):
        hb = int(rng.integers(1, 130))
        w = (hb - 1) // 32 + 1
        nblk = int(rng.integers(3 * w + 2, 60))
        shapes.append((32 * nblk - int(rng.integers(0, 32)), hb, int(rng.choice([0, 0, 2, 4, 8, 16]))))
    return shapes


@pytest.mark.parametrize("n,hb,parts", _edge_shapes())
def test_plan_replay_on_block_boundaries_and_random_shapes(mqs, n, hb, parts):
    """Half bandwidths on and around multiples of the block size, sizes one off a multiple of 32, random shapes: whenever the
    library decides to cut, the replayed plan solves the system; when it declines (0 ints), that is the natural order."""
    lib = mqs._lib.lib()
    plan = _plan(lib, n, hb, parts)
    if plan is None:
        return
    S = _banded_spd(n, hb, seed=3 * n + hb)
    b = np.random.default_rng(n).standard_normal(n)
    x = _replay(S, b, plan)
    ref = np.linalg.solve(S, b)
    assert np.abs(x - ref).max() <= 1e-11 * max(1.0, np.abs(ref).max())


def test_auto_cut_of_the_kt2_shape(mqs):
    """881 poses, landmarks seen by 17 consecutive poses: 166 dependent block columns become 16 chunks of <= 7 + 4 separator
    levels of 4: 23 steps (and 8 chunks, 18 + 3 x 4, when asked for)."""
    lib = mqs._lib.lib()
    plan = _plan(lib, 5286, 101, 0)
    assert plan["parts"] == 16 and len(plan["stages"]) == 5
    assert [int(s[0]) for s in plan["stages"]] == [16, 8, 4, 2, 1]
    assert sum(int(s[1]) for s in plan["stages"]) == 23
    # every block column's structure stays within 3 separator widths
    assert int(plan["descs"][:, 1].max()) <= 12
    plan8 = _plan(lib, 5286, 101, 8)
    assert plan8["parts"] == 8 and [int(s[0]) for s in plan8["stages"]] == [8, 4, 2, 1]
    assert sum(int(s[1]) for s in plan8["stages"]) <= 31


@pytest.mark.parametrize("n,hb,parts", [(300, 101, 0), (5286, 101, 1), (5286, 0, 0), (5286, 2000, 0), (64, 5, 0)])
def test_no_cut_where_it_does_not_pay(mqs, n, hb, parts):
    lib = mqs._lib.lib()
    assert _plan(lib, n, hb, parts) is None
