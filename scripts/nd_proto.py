"""Prototype of the nested-dissection camera order + tile-level symbolic factorisation + level schedule
(counts launches and tile products; no arithmetic).  python scripts/nd_proto.py C3 [depth]"""
import sys
import numpy as np
sys.path.insert(0, '.')
from mpsfm_amd.synthetic import make_config


def camera_graph(prob):
    nc = prob.n_cams
    var = np.ones(nc, bool)
    var[np.asarray(prob.pose_const, bool)] = False
    slot0 = -np.ones(nc, int)
    slot0[var] = np.arange(var.sum())
    ncv = int(var.sum())
    cam = np.concatenate([prob.obs_cam, prob.dobs_cam]) if prob.n_dobs else np.asarray(prob.obs_cam)
    pt = np.concatenate([prob.obs_pt, prob.dobs_pt]) if prob.n_dobs else np.asarray(prob.obs_pt)
    o = np.argsort(pt, kind='stable')
    cam, pt = cam[o], pt[o]
    adj = np.zeros((ncv, ncv), bool)
    st = np.flatnonzero(np.r_[True, pt[1:] != pt[:-1], True])
    for a, b in zip(st[:-1], st[1:]):
        s = np.unique(slot0[cam[a:b]])
        s = s[s >= 0]
        adj[np.ix_(s, s)] = True
    np.fill_diagonal(adj, False)
    return adj


def bfs_levels(adj, nodes, start):
    inset = np.zeros(adj.shape[0], bool); inset[nodes] = True
    lev = -np.ones(adj.shape[0], int)
    lev[start] = 0
    frontier = [start]; levels = [[start]]
    while frontier:
        nxt = []
        for u in frontier:
            for v in np.flatnonzero(adj[u] & inset & (lev < 0)):
                lev[v] = len(levels); nxt.append(v)
        if nxt: levels.append(nxt)
        frontier = nxt
    return levels


def components(adj, nodes):
    left = set(nodes); comps = []
    while left:
        s = min(left)
        lv = bfs_levels(adj, np.array(sorted(left)), s)
        c = [v for l in lv for v in l]
        comps.append(c); left -= set(c)
    return comps


def pseudo_peripheral(adj, nodes):
    deg = adj[np.ix_(nodes, nodes)].sum(1)
    s = nodes[int(np.argmin(deg))]
    best = -1
    for _ in range(8):
        lv = bfs_levels(adj, nodes, s)
        if len(lv) <= best: break
        best = len(lv)
        last = lv[-1]
        d = [adj[v][nodes].sum() for v in last]
        s = last[int(np.argmin(d))]
    return s


def rcm(adj, nodes):
    nodes = np.array(nodes)
    out = []
    for comp in components(adj, nodes):
        comp = np.array(comp)
        s = pseudo_peripheral(adj, comp)
        lv = bfs_levels(adj, comp, s)
        # Cuthill-McKee: within a level order by parent order then degree
        order = [v for l in lv for v in l]
        out += order[::-1]
    return out


class Seg:
    def __init__(self, cams, children=()):
        self.cams = list(cams); self.children = list(children)


def nd(adj, nodes, depth, min_leaf):
    nodes = np.array(nodes)
    comps = components(adj, nodes)
    if len(comps) > 1:
        return Seg([], [nd(adj, c, depth, min_leaf) for c in comps])
    if depth == 0 or len(nodes) < min_leaf:
        return Seg(rcm(adj, nodes))
    s = pseudo_peripheral(adj, nodes)
    lv = bfs_levels(adj, nodes, s)
    if len(lv) < 5:
        return Seg(rcm(adj, nodes))
    sizes = np.array([len(l) for l in lv]); cum = np.cumsum(sizes)
    tot = cum[-1]
    best, bm = None, None
    for m in range(1, len(lv) - 1):
        a, b = cum[m - 1], tot - cum[m]
        cost = max(a, b) + sizes[m]
        if best is None or cost < best: best, bm = cost, m
    sep = lv[bm]
    A = [v for l in lv[:bm] for v in l]; B = [v for l in lv[bm + 1:] for v in l]
    return Seg(rcm(adj, sep), [nd(adj, A, depth - 1, min_leaf), nd(adj, B, depth - 1, min_leaf)])


def flatten(seg, adj, out, segs, is_root=True, move_up_max=4):
    """Post-order; every non-root segment is made a multiple of 16 slots: <= move_up_max cameras move up into the parent,
    otherwise dummy slots (-1) pad it.  Returns the cameras moved up."""
    moved = []
    for c in seg.children:
        moved += flatten(c, adj, out, segs, False, move_up_max)
    cams = moved + seg.cams
    up = []
    if not is_root and cams:
        r = len(cams) % 16
        if 0 < r <= move_up_max and len(cams) > 16:
            up = cams[-r:]; cams = cams[:-r]
        elif r:
            cams = cams + [-1] * (16 - r)
    if cams:
        segs.append((len(out), len(out) + len(cams)))
        out += cams
    elif not is_root and not cams:
        pass
    return up


def symbolic(adj, order):
    ns = len(order)
    n = 6 * ns; nt = (n + 31) // 32
    pos = {c: i for i, c in enumerate(order) if c >= 0}
    pat = np.zeros((nt + 1, nt + 1), bool)
    tiles_of = lambda s: range((6 * s) // 32, (6 * s + 5) // 32 + 1)
    for c, i in pos.items():
        for d in np.flatnonzero(adj[c]):
            j = pos[d]
            for a in tiles_of(i):
                for b in tiles_of(j):
                    if a >= b: pat[a, b] = True
        for a in tiles_of(i):
            for b in tiles_of(i):
                if a >= b: pat[a, b] = True
    pat[nt, :nt] = True  # rhs row
    struct = [None] * nt; parent = -np.ones(nt, int)
    children = [[] for _ in range(nt)]
    for j in range(nt):
        s = set(np.flatnonzero(pat[j + 1:, j]) + j + 1)
        for c in children[j]:
            s |= (struct[c] - {j})
        struct[j] = s
        ps = [i for i in s if i < nt]
        if ps:
            parent[j] = min(ps); children[parent[j]].append(j)
    level = np.zeros(nt, int)
    for j in range(nt):
        level[j] = 1 + max([level[c] for c in children[j]], default=-1)
    return nt, struct, parent, level


def report(name, depth):
    prob, _ = make_config(name)
    adj = camera_graph(prob)
    ncv = adj.shape[0]
    for label, order in [("identity", list(range(ncv)))] + [("nd%d" % d, None) for d in depth]:
        if order is None:
            d = int(label[2:])
            tree = nd(adj, np.arange(ncv), d, 48)
            order, segs = [], []
            flatten(tree, adj, order, segs)
        nt, struct, parent, level = symbolic(adj, order)
        nl = level.max() + 1
        tiles = sum(len(s) + 1 for s in struct)
        trail = sum(len(s) * (len(s) + 1) // 2 for s in struct)
        per = [sum(len(struct[j]) * (len(struct[j]) + 1) // 2 for j in range(nt) if level[j] == l) for l in range(nl)]
        print("%-9s slots %4d (dummies %3d) nt %3d  levels %3d  L tiles %5d of %5d  tile products %6d  max per launch %d" % (
            label, len(order), sum(1 for c in order if c < 0), nt, nl, tiles, nt * (nt + 1) // 2, trail, max(per)))


if __name__ == "__main__":
    name = sys.argv[1] if len(sys.argv) > 1 else "C3"
    depth = [int(x) for x in sys.argv[2:]] or [1, 2, 3]
    report(name, depth)
