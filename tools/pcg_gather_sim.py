"""dev (CPU): LDS bank conflicts of the register-resident PCG's gather of p, simulated on the structure of a configuration's
normal matrix (32 lanes per pass, 32 banks of 4 bytes: the model tools/microbench_lds_gather.hip confirms — `perm32` 3.0 clk
per wave instruction, random 7.0).  Rows by length as the kernel assigns them; the order of a row's entries, the numbering of
the columns and a greedy per-half-wave slot assignment as variants.  Writes the slot -> column tables of some variants for
    tools/microbench_lds_gather <tables.bin> name...
Round 5 (C2): hash order 3 424 clocks per sweep, columns sorted 1 487, greedy 1 420, ideal 1 048 — and on the GPU the sorted
order (the order-stable variant assembles it) shortens the iteration by 7 % only: with the conflicts gone the loop waits for
its VALU work and the LDS latency instead (DESIGN_NOTES A.7).
usage: python tools/pcg_gather_sim.py [C2] [tables.bin]"""
import numpy as np, sys
from scipy.spatial import cKDTree
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from dynfu_amd import synth
cfg = synth.CONFIGS[sys.argv[1] if len(sys.argv)>1 else "C2"]
c = synth.canonical(cfg)
V, P = c["verts"], c["node_pos"]
D, k = len(P), cfg["k"]
tree = cKDTree(P)
_, idx = tree.query(V, k=k)
_, nidx = tree.query(P, k=k+1)
rows = [set() for _ in range(D)]
import itertools
pairs = set()
for cols in (idx, nidx[:, :]):
    a = np.asarray(cols)
    for i in range(a.shape[1]):
        for j in range(a.shape[1]):
            if i != j:
                pairs.update(zip(a[:, i].tolist(), a[:, j].tolist()))
for a, b in pairs:
    rows[a].add(b)
for a in range(D): rows[a].add(a)
cnt = np.array([len(r) for r in rows])
print("D", D, "nnz", cnt.sum(), "mean", cnt.mean(), "max", cnt.max(), "min", cnt.min())
rows = np.array([sorted(r) for r in rows], dtype=object)
rng = np.random.default_rng(0)
D = len(rows); NT = 1024; E = 32
def cost_of(slots, pos):
    """slots: [lane][q] -> column or -1 for one wave (64 lanes); cost in clocks: per half-wave max multiplicity of distinct addresses per bank"""
    tot = 0
    nq = max(len(s) for s in slots)
    for q in range(nq):
        for h in range(2):
            addr = set()
            for l in range(32 * h, 32 * h + 32):
                s = slots[l]
                addr.add(pos[s[q]] if q < len(s) and s[q] >= 0 else pos[0])
            banks = np.bincount([a & 31 for a in addr], minlength=32)
            tot += banks.max()
    return tot, nq
def layout(order_rows, entry_order, pos):
    """order_rows: rank->row (length D).  thread t: row A = rank t, row B = rank D-1-t.  Returns total clocks for the WG's matvec + ideal"""
    tot = 0; ideal = 0
    for w in range(NT // 64):
        A = [entry_order(order_rows[w * 64 + l], w * 64 + l) for l in range(64)]
        B = [entry_order(order_rows[D - 1 - (w * 64 + l)], w * 64 + l) for l in range(64)]
        ca, na = cost_of(A, pos); cb, nb = cost_of(B, pos)
        tot += ca + cb; ideal += 2 * (na + nb)
    return tot, ideal
cnt = np.array([len(r) for r in rows])
by_len = np.argsort(-cnt, kind="stable")
ident = np.arange(D)
def rnd(r, t):
    e = list(rows[r]); rng.shuffle(e); return e
print("current (length-sorted rows, random entry order, natural columns): clocks %d ideal %d" % layout(by_len, rnd, ident))
def sorted_cols(r, t): return sorted(rows[r])
print("sorted columns: %d %d" % layout(by_len, sorted_cols, ident))
# Morton order of nodes
def morton(P):
    q = ((P - P.min(0)) / (P.max(0) - P.min(0) + 1e-9) * 1023).astype(np.int64)
    def spread(v):
        v = (v | (v << 16)) & 0x030000FF; v = (v | (v << 8)) & 0x0300F00F; v = (v | (v << 4)) & 0x030C30C3; v = (v | (v << 2)) & 0x09249249; return v
    return spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)
mo = np.argsort(morton(P)); pos_m = np.empty(D, np.int64); pos_m[mo] = np.arange(D)
print("morton rows + morton positions, random entries: %d %d" % layout(mo, rnd, pos_m))
def sorted_pos(r, t): return sorted(rows[r], key=lambda c: pos_m[c])
print("morton rows + positions, entries sorted by position: %d %d" % layout(mo, sorted_pos, pos_m))
def sorted_rel(r, t): return sorted(rows[r], key=lambda c: (pos_m[c] - pos_m[r]))
print("length rows, morton positions, sorted by pos: %d %d" % layout(by_len, sorted_pos, pos_m))
# greedy matching per half-wave, slot by slot, priority to banks of highest remaining degree
def greedy_wave(rowlist, pos):
    out = [[] for _ in rowlist]
    for h in range(0, len(rowlist), 32):
        lanes = list(range(h, min(h + 32, len(rowlist))))
        rem = {l: list(rowlist[l]) for l in lanes}
        nq = max(len(rem[l]) for l in lanes)
        for q in range(nq):
            deg = np.zeros(32, int)
            for l in lanes:
                for c in rem[l]: deg[pos[c] & 31] += 1
            taken = {}
            # lanes with fewest options first
            for l in sorted(lanes, key=lambda l: len(rem[l])):
                if not rem[l]: out[l].append(-1); continue
                best = None
                for c in rem[l]:
                    b = pos[c] & 31
                    load = taken.get(b, 0)
                    key = (load, -deg[b])
                    if best is None or key < best[0]: best = (key, c)
                c = best[1]; rem[l].remove(c); out[l].append(c); taken[pos[c] & 31] = taken.get(pos[c] & 31, 0) + 1
    return out
def layout_g(order_rows, pos):
    tot = 0; ideal = 0
    for w in range(NT // 64):
        A = greedy_wave([sorted(rows[order_rows[w * 64 + l]]) for l in range(64)], pos)
        B = greedy_wave([sorted(rows[order_rows[D - 1 - (w * 64 + l)]]) for l in range(64)], pos)
        ca, na = cost_of(A, pos); cb, nb = cost_of(B, pos)
        tot += ca + cb; ideal += 2 * (na + nb)
    return tot, ideal
print("greedy per half-wave (length rows, natural cols): %d %d" % layout_g(by_len, ident))
print("greedy per half-wave (morton rows + positions): %d %d" % layout_g(mo, pos_m))

# ---- tables for the microbenchmark: [variant][q][t] uint16
def table(order_rows, fn):
    T = np.zeros((E, NT), np.uint16)
    for w in range(NT // 64):
        A = fn([sorted(rows[order_rows[w * 64 + l]]) for l in range(64)])
        B = fn([sorted(rows[order_rows[D - 1 - (w * 64 + l)]]) for l in range(64)])
        for l in range(64):
            t = w * 64 + l
            for q, c in enumerate(A[l]): T[q, t] = max(c, 0)
            for q, c in enumerate(B[l]): T[E - 1 - q, t] = max(c, 0)
    return T
def f_rand(rl):
    out = []
    for r in rl:
        e = list(r); rng.shuffle(e); out.append(e)
    return out
tabs = [table(by_len, f_rand), table(by_len, lambda rl: rl), table(by_len, lambda rl: greedy_wave(rl, ident))]
# synthetic: conflict-free per half-wave of 32 lanes (32 banks), random otherwise; conflict-free per 64 lanes in 64 banks
T32 = np.zeros((E, NT), np.uint16); T64 = np.zeros((E, NT), np.uint16); T16 = np.zeros((E, NT), np.uint16)
for q in range(E):
    for h in range(NT // 32):
        T32[q, 32 * h:32 * h + 32] = rng.permutation(32) + 32 * rng.integers(0, 64, 32)
    for h in range(NT // 64):
        T64[q, 64 * h:64 * h + 64] = rng.permutation(64) + 64 * rng.integers(0, 32, 64)
    for h in range(NT // 16):
        T16[q, 16 * h:16 * h + 16] = rng.permutation(32)[:16] + 32 * rng.integers(0, 64, 16)
tabs += [T32, T64, T16]
if len(sys.argv) > 2:
    np.stack(tabs).tofile(sys.argv[2])
    print("tables: matrix-random-order matrix-sorted matrix-greedy32 perm32 perm64 perm16of32 ->", sys.argv[2])
