"""Development tool (CPU, numpy / scipy): preconditioners for the north-star normal equations.

Dumps the block-sparse system H x = g of one Gauss-Newton iteration of oracle/solve6_oracle.c on a synthetic frame
and counts PCG iterations to a relative residual for: 6x6 block-Jacobi (the round-2 product), additive Schwarz on
node + regularisation-neighbour patches, and two-level variants (block-Jacobi + a coarse space of rigid motions of
aggregates of nodes).  usage: python tools/pcg6_experiments.py [C2] [frame] [gn]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

import oracle as O
from dynfu_amd import synth


def dump_system(name, frame, gn, warm_iter=64, **kw):
    cfg = synth.CONFIGS[name]
    c = synth.canonical(cfg)
    intr = synth.intrinsics(cfg)
    depth = synth.depth_frame(cfg, frame)
    P, Nm = O.points_normals(depth, *intr)
    D, k = cfg["D"], cfg["k"]
    cap = D * 64
    row_ptr, cols = np.zeros(D + 1, np.int32), np.zeros(cap, np.int32)
    blk, g = np.zeros((cap, 6, 6)), np.zeros(6 * D)
    nblk = C.c_long(0)
    L = O.lib6()
    L.orc6_set_dump.argtypes = [C.c_int] + [C.c_void_p] * 4 + [C.c_long, C.c_void_p]
    L.orc6_set_dump(gn, row_ptr.ctypes.data, cols.ctypes.data, blk.ctypes.data, g.ctypes.data, cap, C.addressof(nblk))
    params = dict(num_iter=1, gn_iter=gn + 1, linear_iter=warm_iter, lambda_=200.0, threads=8)
    params.update(kw)
    dq, st = O.solve6(c["node_pos"], c["node_dq"], c["node_w"], k, c["verts"], c["normals"], P, Nm, intr, **params)
    nb = nblk.value
    H = sp.bsr_matrix((blk[:nb], cols[:nb], row_ptr), shape=(6 * D, 6 * D))
    return H.tocsr(), g, c, st, dq


def pcg(H, g, Minv, tol, maxit=2000, hist=None):
    x = np.zeros_like(g)
    r = g.copy()
    z = Minv(r)
    p = z.copy()
    rz = rz0 = r @ z
    for it in range(maxit):
        q = H @ p
        a = rz / (p @ q)
        x += a * p
        r -= a * q
        z = Minv(r)
        rzn = r @ z
        if hist is not None:
            hist.append(np.sqrt(rzn / rz0))
        if rzn <= tol * tol * rz0:
            return x, it + 1
        p = z + (rzn / rz) * p
        rz = rzn
    return x, maxit


def block_jacobi(H, D):
    Hb = H.tobsr((6, 6))
    inv = np.zeros((D, 6, 6))
    for n in range(D):
        for e in range(Hb.indptr[n], Hb.indptr[n + 1]):
            if Hb.indices[e] == n:
                inv[n] = np.linalg.inv(Hb.data[e])
    return lambda r: np.einsum("nij,nj->ni", inv, r.reshape(D, 6)).reshape(-1)


def schwarz_patches(H, D, reg, weighted=False):
    """additive Schwarz: patch n = node n + its regularisation neighbours"""
    Hc = H.tocsr()
    sols = []
    count = np.zeros(D)
    for n in range(D):
        nodes = [n] + [m for m in reg[n] if m >= 0]
        dof = np.concatenate([np.arange(6 * m, 6 * m + 6) for m in nodes])
        A = Hc[dof][:, dof].toarray()
        sols.append((dof, np.linalg.inv(A)))
        count[nodes] += 1

    def apply(r):
        z = np.zeros_like(r)
        for dof, Ai in sols:
            z[dof] += Ai @ r[dof]
        if weighted:
            z = (z.reshape(D, 6) / count[:, None]).reshape(-1)
        return z
    return apply


def aggregates(pos, size):
    """contiguous runs of `size` nodes in a space-filling (Morton) order of their positions"""
    D = len(pos)
    q = ((pos - pos.min(0)) / (np.ptp(pos, axis=0).max() + 1e-9) * 1023).astype(np.int64)

    def spread(v):
        o = np.zeros_like(v)
        for b in range(10):
            o |= ((v >> b) & 1) << (3 * b)
        return o
    code = spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)
    order = np.argsort(code, kind="stable")
    agg = np.zeros(D, np.int64)
    agg[order] = np.arange(D) // size
    return agg


def prolongation(pos_now, agg, rigid=True):
    """coarse unknown of aggregate A: a rigid twist (omega, v) about the aggregate's centroid; node i's twist (about its own
    position) is (omega, v + omega x (g_i - c_A)).  rigid=False: piecewise constant twists."""
    D = len(pos_now)
    na = agg.max() + 1
    cent = np.zeros((na, 3))
    np.add.at(cent, agg, pos_now)
    cent /= np.bincount(agg)[:, None]
    rows, cols, vals = [], [], []
    for i in range(D):
        a = agg[i]
        d = pos_now[i] - cent[a]
        B = np.eye(6)
        if rigid:
            # v_i = v + omega x d = v - [d]x omega
            B[3:, :3] = -np.array([[0, -d[2], d[1]], [d[2], 0, -d[0]], [-d[1], d[0], 0]])
        for r in range(6):
            for cc in range(6):
                if B[r, cc] != 0:
                    rows.append(6 * i + r), cols.append(6 * a + cc), vals.append(B[r, cc])
    return sp.csr_matrix((vals, (rows, cols)), shape=(6 * D, 6 * na))


def two_level(H, D, Pm, smoother, mode="additive"):
    Ac = (Pm.T @ H @ Pm).toarray()
    Ac += 1e-12 * np.trace(Ac) / len(Ac) * np.eye(len(Ac))
    Aci = np.linalg.inv(Ac)
    if mode == "additive":
        return lambda r: smoother(r) + Pm @ (Aci @ (Pm.T @ r))
    # symmetric multiplicative (one pre- and one post-smoothing step)
    def apply(r):
        z = smoother(r)
        r1 = r - H @ z
        z = z + Pm @ (Aci @ (Pm.T @ r1))
        r2 = r - H @ z
        return z + smoother(r2)
    return apply


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "C2"
    frame = int(sys.argv[2]) if len(sys.argv) > 2 else 7
    gn = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    H, g, c, st, dq = dump_system(name, frame, gn)
    cfg = synth.CONFIGS[name]
    D, k = cfg["D"], cfg["k"]
    print("system %s frame %d gn %d: %d unknowns, %d blocks (%.1f per row)" % (name, frame, gn, 6 * D, H.nnz // 36, H.nnz / 36 / D))
    _, _, reg = O.graph6(c["node_pos"], c["node_w"], k, c["verts"][:16])
    bj = block_jacobi(H, D)
    pos = c["node_pos"].astype(np.float64)
    results = {}

    def run(label, M):
        for tol in (1e-1, 1e-2, 1e-3, 1e-6):
            h = []
            x, it = pcg(H, g, M, tol, hist=h)
            results.setdefault(label, []).append(it)
        print("%-44s iterations to 1e-1 / 1e-2 / 1e-3 / 1e-6: %s" % (label, results[label]), flush=True)

    run("block-Jacobi 6x6", bj)
    run("Schwarz node+reg patches", schwarz_patches(H, D, reg))
    for size in (8, 16, 32, 64):
        agg = aggregates(pos, size)
        for rigid in (False, True):
            Pm = prolongation(pos, agg, rigid)
            run("BJ + coarse(%d nodes, %s) additive" % (size, "rigid" if rigid else "const"), two_level(H, D, Pm, bj))
        run("BJ + coarse(%d nodes, rigid) multiplicative" % size, two_level(H, D, prolongation(pos, agg, True), bj, "mult"))


if __name__ == "__main__":
    main()
