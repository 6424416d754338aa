"""Host side of the level-scheduled tile Cholesky (mpsfm_amd/csrc/chol_plan.hip), checked without a device: the camera
order, the symbolic factorisation and the launch tables are interpreted with NumPy — every item does on 32 x 32 tiles what
its workgroup does on the GPU, launches in order, items of one launch in ANY order (they run concurrently there) — and the
result is compared with a dense Cholesky solve."""

import ctypes as C

import numpy as np
import pytest

from mpsfm_amd import capi

T = 32


def _plan(adj, depth=-2, pinv_max_tiles=64, inv_rows=2):
    L = capi.lib()
    L.mpsfm_debug_plan_create.restype = C.c_void_p
    L.mpsfm_debug_plan_create.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32]
    L.mpsfm_debug_plan_get.restype = C.c_int64
    L.mpsfm_debug_plan_get.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64]
    L.mpsfm_debug_plan_destroy.argtypes = [C.c_void_p]
    a = np.ascontiguousarray(adj, dtype=np.uint8)
    h = L.mpsfm_debug_plan_create(a.ctypes.data, a.shape[0], depth, pinv_max_tiles, inv_rows)
    assert h
    out = {}
    names = ["header", "slot_of_nat", "struct_start", "struct_rows", "parent", "level", "items", "launch_start", "srcs", "rows",
             "asm_tiles", "back_cols", "back_start", "col_of_slot"]
    for what, name in enumerate(names):
        n = L.mpsfm_debug_plan_get(h, what, None, 0)
        buf = np.zeros(max(n, 1), np.int32)
        L.mpsfm_debug_plan_get(h, what, buf.ctypes.data, n)
        out[name] = buf[:n]
    L.mpsfm_debug_plan_destroy(h)
    hd = out["header"]
    out.update(dict(zip(["ncv", "nslots", "n", "nt", "nlevels", "nd_depth", "use_pinv", "n_items", "products", "roles"], map(int, hd))))
    it = out["items"].reshape(-1, 4).astype(np.int64)
    out["items"] = [dict(type=int(a & 0xffff), ti=int((a >> 16) & 0xffff), tk=int(b & 0xffff), nsrc=int((b >> 16) & 0xffff), src=int(c), aux=int(d))
                    for a, b, c, d in it]
    return out


def ring_graph(n, reach, extra=0, seed=0):
    """Cameras on a closed orbit, every camera sharing landmarks with the `reach` next ones, plus a few random long links."""
    rng = np.random.default_rng(seed)
    adj = np.zeros((n, n), np.uint8)
    for i in range(n):
        for d in range(1, reach + 1):
            adj[i, (i + d) % n] = adj[(i + d) % n, i] = 1
    for _ in range(extra):
        a, b = rng.integers(0, n, 2)
        if a != b:
            adj[a, b] = adj[b, a] = 1
    return adj


def reduced_system(adj, P, seed):
    """SPD matrix with the block pattern of the graph, every camera's 6 x 6 blocks at the columns the plan gives its slot
    (padding columns between segments: identity rows), and a rhs."""
    rng = np.random.default_rng(seed)
    ncv, ns, n = P["ncv"], P["nslots"], P["n"]
    slot = P["col_of_slot"][P["slot_of_nat"]] // 6 if False else None
    col = P["col_of_slot"][P["slot_of_nat"]]  # first column of every camera (caller's order)
    S = np.zeros((n, n))
    for i in range(ncv):
        for j in range(i, ncv):
            if i == j or adj[i, j]:
                B = rng.standard_normal((6, 6)) * (1.0 if i == j else 0.3)
                a, b = int(col[i]), int(col[j])
                S[a:a + 6, b:b + 6] += B
                if i != j:
                    S[b:b + 6, a:a + 6] += B.T
    S = 0.5 * (S + S.T)
    real = np.zeros(n, bool)
    for i in range(ncv):
        real[col[i]:col[i] + 6] = True
    S[np.diag_indices(n)] = np.where(real, np.abs(S).sum(1) + 1.0, 1.0)
    rhs = np.where(real, rng.standard_normal(n), 0.0)
    return S, rhs


def interpret(P, S, rhs, rng):
    nt, n = P["nt"], P["n"]
    N = nt * T
    A = np.zeros((N + T, N))
    A[:n, :n] = S
    for r in range(n, N):
        A[r, r] = 1.0
    A[N, :n] = rhs  # row 0 of the rhs tile row
    live = set(int(x) for x in P["asm_tiles"])
    lt = lambda ti, tj: ti * (ti + 1) // 2 + tj
    tile = lambda ti, tj: A[ti * T:(ti + 1) * T, tj * T:(tj + 1) * T]
    # tiles outside the plan must be structurally zero in S
    for ti in range(nt):
        for tj in range(ti):
            if lt(ti, tj) not in live:
                assert not tile(ti, tj).any(), f"tile ({ti},{tj}) of S is nonzero but not in the plan"
    Linv = {}
    Pinv = {}
    srcs, rows = P["srcs"], P["rows"]
    done_col = np.full(nt, -1)
    for l in range(P["nlevels"]):
        items = P["items"][P["launch_start"][l]:P["launch_start"][l + 1]]
        # reads see the state before the launch (what another workgroup of the same launch writes may not be there yet);
        # two items of a launch must never write the same tile
        A0 = A.copy()
        t0 = lambda ti, tj: A0[ti * T:(ti + 1) * T, tj * T:(tj + 1) * T]
        written = set()
        reads = []  # (item index, key) of everything an item reads that it does not own
        Pinv0 = {k: v.copy() for k, v in Pinv.items()}
        for q in rng.permutation(len(items)):
            it = items[q]
            if it["type"] == 2:  # inverse role: P(i,k) += L(i,j) X(j,k)
                j, k = it["ti"], it["tk"]
                assert done_col[j] >= 0 and done_col[j] < l
                X = Linv[j] if k == j else -Linv[j] @ Pinv0.get((j, k), np.zeros((T, T)))
                reads.append((q, ("Linv", j)))
                if k != j:
                    reads.append((q, ("P", j, k)))
                for i in rows[it["aux"]:it["aux"] + it["nsrc"]]:
                    reads.append((q, ("A", int(i), j)))
                    key = ("P", int(i), k)
                    assert key not in written
                    written.add(key)
                    Pinv[(int(i), k)] = Pinv0.get((int(i), k), np.zeros((T, T))) + t0(int(i), j) @ X
                continue
            ti, tk = it["ti"], it["tk"]
            src = srcs[it["src"]:it["src"] + it["nsrc"]]
            if it["type"] == 1:  # trailing tile
                assert lt(ti, tk) in live and ("A", ti, tk) not in written
                written.add(("A", ti, tk))
                acc = t0(ti, tk).copy()
                for c in src:
                    reads += [(q, ("A", ti, int(c))), (q, ("A", tk, int(c)))]
                    assert done_col[c] >= 0 and done_col[c] < l
                    acc -= t0(ti, int(c)) @ t0(tk, int(c)).T
                tile(ti, tk)[:] = acc
                continue
            # panel tile of column tk
            D = t0(tk, tk).copy()
            X = t0(ti, tk).copy() if ti != tk else None
            reads.append((q, ("A", tk, tk)))
            for e in src:
                c, xf = int(e) & 0xffff, bool(int(e) >> 16)
                reads.append((q, ("A", tk, c)))
                if X is not None and xf:
                    reads.append((q, ("A", ti, c)))
                assert done_col[c] >= 0 and done_col[c] < l
                D -= t0(tk, c) @ t0(tk, c).T
                if X is not None and xf:
                    X -= t0(ti, c) @ t0(tk, c).T
            Lkk = np.linalg.cholesky(D)
            if ti == tk:
                Linv[tk] = np.linalg.inv(Lkk)
                written.add(("Linv", tk))
                done_col[tk] = l
            else:
                assert ("A", ti, tk) not in written
                written.add(("A", ti, tk))
                tile(ti, tk)[:] = np.linalg.solve(Lkk, X.T).T
        # no item reads what ANOTHER item of the same launch writes (on the GPU it might see either value)
        for q, key in reads:
            assert key not in written, f"launch {l}: item {q} reads {key}, which another item of the launch writes"
    assert (done_col >= 0).all()
    z = A[N, :N].copy()  # forward-substituted rhs
    w = np.concatenate([Linv[i].T @ z[i * T:(i + 1) * T] for i in range(nt)])
    if P["use_pinv"]:
        y = w.copy()
        for (i, k), Pt in Pinv.items():
            y[k * T:(k + 1) * T] -= Pt.T @ w[i * T:(i + 1) * T]
    else:
        y = np.zeros(N)
        ss, sr = P["struct_start"], P["struct_rows"]
        lev = P["level"]
        for b in range(len(P["back_start"]) - 1):
            cols = P["back_cols"][P["back_start"][b]:P["back_start"][b + 1]]
            for j in cols:
                v = z[j * T:(j + 1) * T].copy()
                for i in sr[ss[j]:ss[j + 1]]:
                    if i < nt:
                        assert lev[i] > lev[j]
                        v -= tile(int(i), int(j)).T @ y[i * T:(i + 1) * T]
                y[j * T:(j + 1) * T] = Linv[int(j)].T @ v
    return y[:n]


@pytest.mark.parametrize("n,reach,extra,depth,pinv", [
    (70, 5, 0, -1, True), (70, 5, 0, 0, True), (70, 5, 0, 1, True), (120, 6, 0, 2, True), (120, 6, 3, -2, True),
    (150, 4, 0, 2, False), (260, 5, 2, 3, False), (20, 19, 0, -2, True), (7, 2, 0, -2, True),
])
def test_plan_solves_the_reduced_system(n, reach, extra, depth, pinv):
    adj = ring_graph(n, reach, extra, seed=n)
    P = _plan(adj, depth=depth, pinv_max_tiles=64 if pinv else 0)
    assert P["ncv"] == n and P["nslots"] == n and P["n"] >= 6 * n
    slot = P["slot_of_nat"]
    assert sorted(slot.tolist()) == list(range(n))  # a permutation
    c = np.sort(P["col_of_slot"])
    assert c[0] == 0 and (np.diff(c) >= 6).all() and c[-1] + 6 == P["n"]  # six columns each, no overlap
    assert P["use_pinv"] == int(pinv and P["nt"] <= 64)
    S, rhs = reduced_system(adj, P, seed=1)
    y = interpret(P, S, rhs, np.random.default_rng(5))
    ref = np.linalg.solve(S, rhs)
    assert np.abs(y - ref).max() <= 1e-9 * max(1.0, np.abs(ref).max())


def test_two_separate_scenes_are_two_chains():
    a = ring_graph(60, 4)
    adj = np.zeros((120, 120), np.uint8)
    adj[:60, :60] = a
    adj[60:, 60:] = a
    P = _plan(adj, depth=0)
    # every component is a multiple of 16 slots but the last; the two chains advance in the same launches
    assert P["nlevels"] <= (P["nt"] + 1) // 2 + 2
    S, rhs = reduced_system(adj, P, seed=2)
    y = interpret(P, S, rhs, np.random.default_rng(1))
    assert np.abs(y - np.linalg.solve(S, rhs)).max() < 1e-9


def test_dissection_shortens_the_chain_of_an_orbit():
    adj = ring_graph(199, 14)  # the C3 camera graph in shape
    ident = _plan(adj, depth=-1)
    auto = _plan(adj, depth=-2)
    assert ident["nlevels"] == ident["nt"] == 38
    assert auto["nd_depth"] >= 1 and auto["nlevels"] <= 24
    assert auto["products"] <= 2 * ident["products"]


def _graph(kind, n, seed=0):
    rng = np.random.default_rng(seed)
    adj = np.zeros((n, n), np.uint8)
    if kind == "path":
        for i in range(n - 1):
            adj[i, i + 1] = 1
    elif kind == "star":
        adj[0, 1:] = 1
    elif kind == "complete":
        adj[:] = 1
    elif kind == "grid":  # cameras on a lattice (aerial blocks): neighbours in both directions
        w = int(np.sqrt(n))
        for i in range(n):
            for d in (1, w, w + 1, w - 1):
                j = i + d
                if j < n and not (d == 1 and j % w == 0):
                    adj[i, j] = 1
    elif kind == "random":
        m = rng.random((n, n)) < 6.0 / n
        adj[m] = 1
    elif kind == "isolated":  # a sequence, a clique and cameras that share nothing with anyone
        for i in range(40):
            for d in (1, 2, 3):
                if i + d < 40:
                    adj[i, i + d] = 1
        adj[50:62, 50:62] = 1
    adj = np.maximum(adj, adj.T)
    np.fill_diagonal(adj, 0)
    return adj


@pytest.mark.parametrize("kind,n", [("path", 90), ("star", 70), ("complete", 40), ("grid", 144), ("random", 100), ("isolated", 75)])
@pytest.mark.parametrize("depth", [-2, 3])
def test_plan_on_other_camera_graphs(kind, n, depth):
    adj = _graph(kind, n, seed=n)
    P = _plan(adj, depth=depth)
    S, rhs = reduced_system(adj, P, seed=3)
    y = interpret(P, S, rhs, np.random.default_rng(9))
    ref = np.linalg.solve(S, rhs)
    assert np.abs(y - ref).max() <= 1e-9 * max(1.0, np.abs(ref).max())
    if kind == "path" and depth == -2:
        assert P["nlevels"] < P["nt"]  # a chain dissects into chains
    if kind == "complete":
        assert P["nlevels"] == P["nt"]  # nothing to gain: one dense chain, whatever the order
