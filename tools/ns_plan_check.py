"""dev (GPU; a -DDFA_S6_DEBUG=<first node> build): the north-star plan and assembly against sums made here in fp64.
  1. the pair lists of s6_pattern: slot 0 holds every row with its own neighbour, every upper slot exactly the pairs whose
     neighbour is the slot's column, ascending; the 256 work units of a node tile its lists (every record once);
  2. the moments S the assembly sums (dumped by the debug build for 256 nodes from <first node>) against sum h_a h_j l l^T
     over the same records of the vertex records in global memory, and -J^T r in l coordinates;
  3. the blocks it writes against M_a S M_b^T with the M it staged, their mirrors, and M^-1 of the diagonal block.
usage: DFA_EXTRA_CXXFLAGS=-DDFA_S6_DEBUG=1024 python dynfu_amd/build.py --force; BASE=1024 CFG=C3 FR=7 python tools/ns_plan_check.py
(LAM: lambda of the solve, default 0 so that the blocks are the data term's alone; KK: k override)"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import dynfu_amd as A
from dynfu_amd import synth
L = A.load()
hip = C.CDLL("libamdhip64.so.7") if False else None
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
def fetch(ptr, n, dtype):
    t = torch.empty(n, dtype=dtype, device="cuda")
    # device-to-device copy through hipMemcpy of the runtime torch loaded
    import ctypes
    rt = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
    rt.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    assert rt.hipMemcpy(t.data_ptr(), ptr, t.numel() * t.element_size(), 3) == 0
    return t.cpu().numpy()
name, frame = os.environ.get("CFG", "T0"), int(os.environ.get("FR", 4))
cfg = dict(synth.CONFIGS[name], k=int(os.environ.get("KK", synth.CONFIGS[name]["k"]))); c = synth.canonical(cfg); intr = synth.intrinsics(cfg)
depth = synth.depth_frame(cfg, frame); k = cfg["k"]
P, Nm = A.compute_points_normals(dev(depth), *intr)
s = A.Solver6(cfg["D"], len(c["verts"]), k)
keep = [dev(c["node_pos"]), dev(c["node_dq"]), dev(c["node_w"]), dev(c["verts"]), dev(c["normals"])]
s.set_problem(*keep)
s.solve(P, Nm, *intr, A.Solve6Params(num_iter=1, gn_iter=1, linear_iter=2, lambda_=float(os.environ.get("LAM", "0"))))
torch.cuda.synchronize()
out = (C.c_void_p * 20)()
L.dfa_dev_solver6_ptrs.argtypes = [C.c_void_p, C.c_void_p]
assert L.dfa_dev_solver6_ptrs(s._h, out) == 0
cap, D, N = int(out[15] or 0), int(out[16] or 0), int(out[17] or 0)
print("cap", cap, "D", D, "N", N)
utab = fetch(out[0], D * 256, torch.int32).view(np.uint32).reshape(D, 256)
pptr = fetch(out[1], D * (cap + 1), torch.int32).reshape(D, cap + 1)
plist = fetch(out[2], N * k * k, torch.int32).view(np.uint32)
bcnt = fetch(out[3], D, torch.int32); bfu = fetch(out[4], D, torch.int32)
nptr = fetch(out[5], D + 1, torch.int32); nlist = fetch(out[6], N * k, torch.int32).view(np.uint32)
idx = fetch(out[10], N * k, torch.int32).reshape(N, k)
bcols = fetch(out[7], D * cap, torch.int32).reshape(D, cap)
bad = 0
BASE = int(os.environ.get('BASE', 0))
DD = min(D, 256)
for a in range(BASE, BASE + DD):
    cnt, fu = bcnt[a], bfu[a]
    beg, ln = nptr[a], nptr[a + 1] - nptr[a]
    rows = nlist[beg:beg + ln]
    # slot 0 list
    l0 = plist[pptr[a, 0]:pptr[a, 1]]
    exp0 = (np.arange(ln, dtype=np.uint32) << 8) | ((rows % k) << 4) | (rows % k)
    if len(l0) != ln or not np.array_equal(l0, exp0):
        bad += 1
        if bad < 4: print("node", a, "slot-0 list wrong: len", len(l0), "rows", ln, l0[:6], exp0[:6])
    # every record of every list covered exactly once by the units
    seen = {}
    for u in range(256):
        info = int(utab[a, u]); q = info & 63; ph = (info >> 6) & 1023; n = (info >> 16) & 1023
        if q >= cnt or not (q == 0 or q >= fu) or n == 0: continue
        for at in range(pptr[a, q] + ph, pptr[a, q + 1], n):
            seen[at] = seen.get(at, 0) + 1
    want = set(range(pptr[a, 0], pptr[a, 1]))
    for q in range(fu, cnt): want |= set(range(pptr[a, q], pptr[a, q + 1]))
    if set(seen) != want or any(v != 1 for v in seen.values()):
        bad += 1
        if bad < 8: print("node", a, "units do not tile the lists: seen", len(seen), "want", len(want), "cnt", cnt, "fu", fu)
    # upper lists: exactly the pairs whose neighbour is the slot's column, ascending
    vv = rows.astype(np.int64) // k
    nb = idx[vv]                      # (ln, k)
    for q in range(fu, cnt):
        rr, jj = np.nonzero(nb == bcols[a, q])
        exp = (rr.astype(np.uint32) << 8) | ((rows[rr] % k).astype(np.uint32) << 4) | jj.astype(np.uint32)
        got = plist[pptr[a, q]:pptr[a, q + 1]]
        if len(exp) != len(got) or not np.array_equal(exp, got):
            bad += 1
            if bad < 12: print("node", a, "slot", q, "list differs: len", len(got), "expected", len(exp))
print("bad", bad)
# ---- moments against a CPU sum
KT = 4 if k <= 4 else 8  # the kernels' template K: a record is l[8], f[KT], (weight, weight x residual, 0, 0)
rec = fetch(out[11], N * (12 + KT), torch.float32).reshape(N, 12 + KT)
rmeta = rec[:, 8 + KT:]
mom = np.zeros(256 * (48 * 36 + 8), np.float32)
L.dfa_dev_s6_moments.argtypes = [C.c_void_p]
assert L.dfa_dev_s6_moments(mom.ctypes.data) == 0
mom = mom.reshape(256, 48 * 36 + 8)
def sym(i, j):
    if i > j: i, j = j, i
    t = i >> 1
    if i % 2 == 0: return 2 * [0, 4, 7, 9][t] + (j - i)
    if j == i: return 32 + t
    return 20 + 2 * [0, 3, 5][t] + (j - i - 1)
for a in (BASE, BASE + 1, BASE + 17, BASE + 200):
    cnt, fu = bcnt[a], bfu[a]
    beg, ln = nptr[a], nptr[a + 1] - nptr[a]
    rows = nlist[beg:beg + ln]
    for q in [0] + list(range(fu, min(cnt, fu + 2))):
        S = np.zeros((8, 8)); g8 = np.zeros(8)
        for pr in plist[pptr[a, q]:pptr[a, q + 1]]:
            r, j = int(pr) >> 8, int(pr) & 15
            en = int(rows[r]); v, oj = en // k, en % k
            l = rec[v, :8].astype(np.float64); h = rec[v, 8:8 + k].astype(np.float64)  # h_j = sqrt(rho) f_j
            S += h[oj] * h[j] * np.outer(l, l)
            if q == 0: g8 += -float(rmeta[v, 0]) * h[oj] * l                           # rmeta[v, 0] = sqrt(rho) r
        got = np.array([[mom[a - BASE, q * 36 + sym(i, j)] for j in range(8)] for i in range(8)])
        err = np.abs(got - S).max() / max(np.abs(S).max(), 1e-30)
        print("node %d slot %d: |S| %.3e  rel err %.2e" % (a, q, np.abs(S).max(), err), end="")
        if q == 0:
            gg = mom[a - BASE, 48 * 36:48 * 36 + 8]
            print("   g8 rel err %.2e" % (np.abs(gg - g8).max() / max(np.abs(g8).max(), 1e-30)), end="")
        print()
# ---- blocks against M S M^T
mm = np.zeros(256 * 48 * 48, np.float32)
L.dfa_dev_s6_m.argtypes = [C.c_void_p]
assert L.dfa_dev_s6_m(mm.ctypes.data) == 0
mm = mm.reshape(256, 48, 6, 8)
bvals = fetch(out[8], D * cap * 36, torch.float32).reshape(D, cap, 6, 6)
gdev = fetch(out[9], D * 6, torch.float32).reshape(D, 6)
rslot = fetch(out[14], D * cap, torch.uint8).reshape(D, cap)
for a in (BASE, BASE + 1, BASE + 17, BASE + 40):
    cnt, fu = bcnt[a], bfu[a]
    for q in [0] + list(range(fu, min(cnt, fu + 3))):
        S = np.array([[mom[a - BASE, q * 36 + sym(i, j)] for j in range(8)] for i in range(8)], np.float64)
        H = mm[a - BASE, 0].astype(np.float64) @ S @ mm[a - BASE, q].astype(np.float64).T
        if q == 0: H += 1e-4 * 0 * np.eye(6)
        got = bvals[a, q]
        print("node %d slot %d (col %d): |H| %.3e  err %.2e" % (a, q, bcols[a, q], np.abs(H).max(), np.abs(got - H).max() / max(np.abs(H).max(), 1e-30)), end="")
        if q == 0:
            g = mm[a - BASE, 0].astype(np.float64) @ mom[a - BASE, 48 * 36:48 * 36 + 8]
            print("   g err %.2e" % (np.abs(gdev[a] - g).max() / max(np.abs(g).max(), 1e-30)), end="")
        else:
            b, rs = bcols[a, q], rslot[a, q]
            print("   mirror err %.2e (rslot %d)" % (np.abs(bvals[b, rs] - H.T).max() / max(np.abs(H).max(), 1e-30), rs), end="")
        print()
# ---- every node, every block
worst = 0
for a in range(BASE, BASE + DD):
    cnt, fu = bcnt[a], bfu[a]
    beg, ln = nptr[a], nptr[a + 1] - nptr[a]
    rows = nlist[beg:beg + ln]
    for q in [0] + list(range(fu, cnt)):
        prs = plist[pptr[a, q]:pptr[a, q + 1]].astype(np.int64)
        r, j = prs >> 8, prs & 15
        en = rows[r].astype(np.int64); v, oj = en // k, en % k
        l = rec[v, :8].astype(np.float64)
        f = rec[v, 8:8 + k].astype(np.float64)
        cf = f[np.arange(len(v)), oj] * f[np.arange(len(v)), j]
        S = (l * cf[:, None]).T @ l
        got = np.array([[mom[a - BASE, q * 36 + sym(i, jj)] for jj in range(8)] for i in range(8)])
        err = np.abs(got - S).max() / max(np.abs(S).max(), 1e-30)
        H = mm[a - BASE, 0].astype(np.float64) @ S @ mm[a - BASE, q].astype(np.float64).T
        if q == 0: H += 1e-4 * np.eye(6) * 0
        herr = np.abs(bvals[a, q] - H).max() / max(np.abs(H).max(), 1e-30)
        b, rs = bcols[a, q], rslot[a, q]
        merr = 0.0 if q == 0 else np.abs(bvals[b, rs] - H.T).max() / max(np.abs(H).max(), 1e-30)
        if np.abs(S).max() < 1e-12: err = merr = herr = 0.0
        if max(err, merr, herr if q else 0) > 1e-5:
            print("node", a, "rows", ln, "slot", q, "records", len(prs), "S err %.2e H err %.2e mirror err %.2e" % (err, herr, merr))
        worst = max(worst, err, merr)
print("worst", worst)

minv = fetch(out[18], D * 36, torch.float32).reshape(D, 6, 6)
w = 0
for a in range(BASE, BASE + DD):
    Hd = bvals[a, 0].astype(np.float64)
    e = np.abs(minv[a] @ Hd - np.eye(6)).max()
    w = max(w, e)
    if e > 1e-3 and a < 12: print("node", a, "minv H - I:", e, "cond", np.linalg.cond(Hd))
print("worst |minv H - I|", w)
