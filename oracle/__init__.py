"""ctypes front-end of the CPU oracle (oracle/liboracle.so) and of oracle/_ref.

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg — never from the product package ``dynfu_amd``.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_REF = None


def build(force=False):
    """Compile liboracle.so (and _ref/ when the reference checkout is present)."""
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".c", ".h", ".inc", ".cpp"))]
    stale = force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs)
    ref_so = os.path.join(_HERE, "_ref", "libref_knn.so")
    ref_mc = os.path.join(_HERE, "_ref", "mc_tables.bin")
    have_ref = os.path.exists("/root/reference/include/nanoflann/nanoflann.hpp")
    if stale or (have_ref and not (os.path.exists(ref_so) and os.path.exists(ref_mc))):
        subprocess.check_call(["make", "-C", _HERE, "-s", "all"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _declare(_LIB)
    return _LIB


def ref_lib():
    """The reference's own nanoflann k-NN (oracle/_ref/libref_knn.so); None if not built."""
    global _REF
    if _REF is None:
        p = os.path.join(_HERE, "_ref", "libref_knn.so")
        if not os.path.exists(p):
            build()
        if not os.path.exists(p):
            return None
        _REF = C.CDLL(p)
        _REF.ref_knn.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        _REF.ref_knn.restype = None
    return _REF


def ref_mc_tables():
    """The reference's marching-cubes case tables (oracle/_ref/mc_tables.bin, extracted from the
    reference checkout at build time); None if not built.  Returns (tri 256x16, num_verts 256) int32."""
    p = os.path.join(_HERE, "_ref", "mc_tables.bin")
    if not os.path.exists(p):
        build()
    if not os.path.exists(p):
        return None
    a = np.fromfile(p, dtype="<i4")
    return np.ascontiguousarray(a[:4096].reshape(256, 16)), np.ascontiguousarray(a[4096:])


class SolveParams(C.Structure):
    _fields_ = [
        ("num_iter", C.c_int),
        ("nonlinear_iter", C.c_int),
        ("linear_iter", C.c_int),
        ("tukey_offset", C.c_float),
        ("psi_data", C.c_float),
        ("lambda_", C.c_float),
        ("psi_reg", C.c_float),
        ("pcg_tol", C.c_float),
        ("gn_tol", C.c_float),
        ("use_double", C.c_int),
        ("threads", C.c_int),
    ]


class SolveStats(C.Structure):
    _fields_ = [
        ("initial_cost", C.c_double),
        ("final_cost", C.c_double),
        ("grad_first", C.c_double),
        ("gn_iters", C.c_int),
        ("pcg_iters", C.c_int),
    ]


def _declare(L):
    vp, i, f = C.c_void_p, C.c_int, C.c_float
    L.orc_float_to_half.argtypes = [f]
    L.orc_float_to_half.restype = C.c_uint16
    L.orc_half_to_float.argtypes = [C.c_uint16]
    L.orc_half_to_float.restype = f
    L.orc_compute_dists.argtypes = [vp, i, vp, i, i, i, f, f, f, f]
    L.orc_compute_dists.restype = None
    L.orc_tsdf_clear.argtypes = [vp, i, i, i]
    L.orc_tsdf_clear.restype = None
    L.orc_tsdf_integrate.argtypes = [vp, i, i, i, vp, i, i, i, vp, f, i, vp, f, f, f, f, i]
    L.orc_tsdf_integrate.restype = C.c_long
    L.orc_tsdf_integrate_slab.argtypes = [vp, i, i, i, vp, i, i, i, i, vp, f, i, vp, f, f, f, f, i]
    L.orc_tsdf_integrate_slab.restype = C.c_long
    L.orc_tsdf_raycast_points.argtypes = [vp, i, i, i, vp, f, vp, vp, f, f, f, f, f, f, vp, i, vp, i, i, i, i]
    L.orc_tsdf_raycast_points.restype = None
    L.orc_tsdf_vertex_normals.argtypes = [vp, i, i, i, vp, f, vp, i, vp]
    L.orc_tsdf_vertex_normals.restype = None
    L.orc_correspond_projective.argtypes = [vp, vp, i, vp, i, vp, i, i, i, f, f, f, f, f, f, vp, vp, vp]
    L.orc_correspond_projective.restype = None
    L.orc_tsdf_raycast_depth.argtypes = [vp, i, i, i, vp, f, vp, vp, f, f, f, f, f, f, vp, i, vp, i, i, i, i]
    L.orc_tsdf_raycast_depth.restype = None
    L.orc_dq_from_euler.argtypes = [f, f, f, f, f, f, vp]
    L.orc_dq_from_quat_trans.argtypes = [vp, vp, vp]
    L.orc_dq_from_rodrigues.argtypes = [vp, vp, vp]
    for n in ("add", "sub", "mul"):
        getattr(L, "orc_dq_" + n).argtypes = [vp, vp, vp]
    L.orc_dq_scale.argtypes = [vp, f, vp]
    L.orc_dq_normalize.argtypes = [vp, vp]
    L.orc_dq_get_translation.argtypes = [vp, vp]
    L.orc_dq_transform_vertex.argtypes = [vp, vp, vp]
    L.orc_dq_get_rodrigues.argtypes = [vp, vp]
    for n in ("roll", "pitch", "yaw"):
        getattr(L, "orc_dq_" + n).argtypes = [vp]
        getattr(L, "orc_dq_" + n).restype = f
    L.orc_marching_cubes.argtypes = [vp, i, i, i, vp, vp, vp, vp, C.c_long, vp]
    L.orc_marching_cubes.restype = C.c_long
    L.orc_knn.argtypes = [vp, i, vp, i, i, vp, i]
    L.orc_transformation_weight.argtypes = [vp, f, vp]
    L.orc_transformation_weight.restype = f
    L.orc_calc_dqb.argtypes = [vp, vp, vp, i, i, vp, vp]
    L.orc_warp_to_live.argtypes = [vp, vp, vp, i, i, vp, vp, i, vp, vp, i]
    L.orc_solve_ref.argtypes = [vp, vp, vp, i, i, vp, vp, i, C.POINTER(SolveParams), vp, vp, C.POINTER(SolveStats)]
    L.orc_tukey_weights.argtypes = [vp, vp, vp, i, i, vp, vp, i, f, f, vp, i]
    L.orc_huber_weights.argtypes = [vp, vp, vp, i, i, f, vp]


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


# ------------------------------------------------------------------ half / TSDF ----
def float_to_half(x):
    return int(lib().orc_float_to_half(float(np.float32(x))))


def half_to_float(h):
    return float(lib().orc_half_to_float(int(h)))


def compute_dists(depth, fx, fy, cx, cy):
    depth = np.ascontiguousarray(depth, dtype=np.uint16)
    rows, cols = depth.shape
    out = np.empty_like(depth)
    lib().orc_compute_dists(_p(depth), depth.strides[0], _p(out), out.strides[0], cols, rows, fx, fy, cx, cy)
    return out


def tsdf_integrate(vol, dists, voxel_size, trunc, max_weight, vol2cam, fx, fy, cx, cy, threads=1):
    """vol: uint32 (Z,Y,X) contiguous, updated in place. Returns #updated voxels."""
    assert vol.dtype == np.uint32 and vol.flags.c_contiguous
    Z, Y, X = vol.shape
    dists = np.ascontiguousarray(dists, dtype=np.uint16)
    rows, cols = dists.shape
    vs, a = _f32(voxel_size), _f32(vol2cam).reshape(-1)
    return int(lib().orc_tsdf_integrate(_p(dists), dists.strides[0], cols, rows, _p(vol), X, Y, Z, _p(vs), trunc,
                                        max_weight, _p(a), fx, fy, cx, cy, threads))


def tsdf_integrate_slab(slab, z0, dists, voxel_size, trunc, max_weight, vol2cam, fx, fy, cx, cy, threads=1):
    """slab: uint32 (z1 - z0, Y, X) = slices [z0, z1) of a volume, updated in place exactly as the full sweep would."""
    assert slab.dtype == np.uint32 and slab.flags.c_contiguous
    n, Y, X = slab.shape
    dists = np.ascontiguousarray(dists, dtype=np.uint16)
    rows, cols = dists.shape
    vs, a = _f32(voxel_size), _f32(vol2cam).reshape(-1)
    return int(lib().orc_tsdf_integrate_slab(_p(dists), dists.strides[0], cols, rows, _p(slab), X, Y, z0, z0 + n, _p(vs),
                                             trunc, max_weight, _p(a), fx, fy, cx, cy, threads))


def tsdf_raycast_points(vol, voxel_size, trunc, cam2vol, Rinv, fx, fy, cx, cy, step_factor, delta_factor, cols,
                        rows, threads=1):
    Z, Y, X = vol.shape
    pts = np.empty((rows, cols, 4), np.float32)
    nrm = np.empty((rows, cols, 4), np.float32)
    vs, a, ri = _f32(voxel_size), _f32(cam2vol).reshape(-1), _f32(Rinv).reshape(-1)
    lib().orc_tsdf_raycast_points(_p(vol), X, Y, Z, _p(vs), trunc, _p(a), _p(ri), fx, fy, cx, cy, step_factor,
                                  delta_factor, _p(pts), pts.strides[0], _p(nrm), nrm.strides[0], cols, rows, threads)
    return pts, nrm


def correspond_projective(vertices, normals, vmap, nmap, fx, fy, cx, cy, dist_thresh, min_cosine):
    """projective association (find_coresp's gates); vmap / nmap: rows x cols x 4 float32; returns (v, n or None, pixel)"""
    v = np.ascontiguousarray(vertices, np.float32)
    nr = None if normals is None else np.ascontiguousarray(normals, np.float32)
    vm = np.ascontiguousarray(vmap, np.float32)
    nm = None if nmap is None else np.ascontiguousarray(nmap, np.float32)
    rows, cols = vm.shape[:2]
    out_v = np.empty((len(v), 3), np.float32)
    out_n = None if nm is None else np.empty((len(v), 3), np.float32)
    pix = np.empty(len(v), np.int32)
    lib().orc_correspond_projective(_p(v), None if nr is None else _p(nr), len(v), _p(vm), vm.strides[0],
                                    None if nm is None else _p(nm), 0 if nm is None else nm.strides[0], cols, rows, fx, fy,
                                    cx, cy, dist_thresh, min_cosine, _p(out_v), None if out_n is None else _p(out_n), _p(pix))
    return out_v, out_n, pix


def tsdf_vertex_normals(vol, voxel_size, delta_factor, points):
    """normals (n x 4) of `points` (n x 4, volume metric frame) from the TSDF gradient: compute_normal of the raycaster"""
    Z, Y, X = vol.shape
    pts = np.ascontiguousarray(points, np.float32)
    out = np.empty_like(pts)
    lib().orc_tsdf_vertex_normals(_p(vol), X, Y, Z, _p(_f32(voxel_size)), delta_factor, _p(pts), len(pts), _p(out))
    return out


def tsdf_raycast_depth(vol, voxel_size, trunc, cam2vol, Rinv, fx, fy, cx, cy, step_factor, delta_factor, cols, rows,
                       threads=1):
    Z, Y, X = vol.shape
    dep = np.empty((rows, cols), np.uint16)
    nrm = np.empty((rows, cols, 4), np.float32)
    vs, a, ri = _f32(voxel_size), _f32(cam2vol).reshape(-1), _f32(Rinv).reshape(-1)
    lib().orc_tsdf_raycast_depth(_p(vol), X, Y, Z, _p(vs), trunc, _p(a), _p(ri), fx, fy, cx, cy, step_factor,
                                 delta_factor, _p(dep), dep.strides[0], _p(nrm), nrm.strides[0], cols, rows, threads)
    return dep, nrm


# ------------------------------------------------------------------ dual quaternion -
def dq_from_euler(yaw, pitch, roll, x, y, z):
    out = np.empty(8, np.float32)
    lib().orc_dq_from_euler(yaw, pitch, roll, x, y, z, _p(out))
    return out


def dq_from_rodrigues(rod, t):
    out = np.empty(8, np.float32)
    lib().orc_dq_from_rodrigues(_p(_f32(rod)), _p(_f32(t)), _p(out))
    return out


def _dq_bin(name, a, b):
    out = np.empty(8, np.float32)
    getattr(lib(), "orc_dq_" + name)(_p(_f32(a)), _p(_f32(b)), _p(out))
    return out


def dq_add(a, b):
    return _dq_bin("add", a, b)


def dq_sub(a, b):
    return _dq_bin("sub", a, b)


def dq_mul(a, b):
    return _dq_bin("mul", a, b)


def dq_scale(a, s):
    out = np.empty(8, np.float32)
    lib().orc_dq_scale(_p(_f32(a)), s, _p(out))
    return out


def dq_normalize(a):
    out = np.empty(8, np.float32)
    lib().orc_dq_normalize(_p(_f32(a)), _p(out))
    return out


def dq_transform_vertex(a, v):
    out = np.empty(3, np.float32)
    lib().orc_dq_transform_vertex(_p(_f32(a)), _p(_f32(v)), _p(out))
    return out


def dq_get_translation(a):
    out = np.empty(3, np.float32)
    lib().orc_dq_get_translation(_p(_f32(a)), _p(out))
    return out


def dq_get_rodrigues(a):
    out = np.empty(3, np.float32)
    lib().orc_dq_get_rodrigues(_p(_f32(a)), _p(out))
    return out


def dq_roll(a):
    return float(lib().orc_dq_roll(_p(_f32(a))))


def dq_pitch(a):
    return float(lib().orc_dq_pitch(_p(_f32(a))))


def dq_yaw(a):
    return float(lib().orc_dq_yaw(_p(_f32(a))))


# ------------------------------------------------------------------ warp field -----
def marching_cubes(vol, cell_size, tri_table, num_verts_table, max_vertices=None):
    """orc_marching_cubes: (points float32 (n,4), total vertices, occupied voxels).  vol: uint32 (Z,Y,X)."""
    vol = np.ascontiguousarray(vol, np.uint32)
    Z, Y, X = vol.shape
    tri = np.ascontiguousarray(tri_table, np.int32)
    nv = np.ascontiguousarray(num_verts_table, np.int32)
    cs = _f32(cell_size)
    occ = C.c_long(0)
    if max_vertices is None:  # count first
        max_vertices = lib().orc_marching_cubes(_p(vol), X, Y, Z, _p(cs), _p(tri), _p(nv), None, 0, C.byref(occ))
    out = np.zeros((max(max_vertices, 1), 4), np.float32)
    total = lib().orc_marching_cubes(_p(vol), X, Y, Z, _p(cs), _p(tri), _p(nv), _p(out), max_vertices, C.byref(occ))
    return out[:min(total, max_vertices)], total, occ.value


def knn(nodes, query, k, threads=1):
    nodes, query = _f32(nodes), _f32(query)
    idx = np.empty((len(query), k), np.int32)
    lib().orc_knn(_p(nodes), len(nodes), _p(query), len(query), k, _p(idx), threads)
    return idx


def correspond(canon_v, canon_n, live_v, threads=1):
    """DynFusion::findCorrespondingFrame (src/dynfu/dyn_fusion.cpp:212-242): nearest canonical vertex
    of every live vertex (exact 1-NN, orc_knn) and the gathered vertex / normal clouds."""
    canon_v = _f32(canon_v)
    idx = knn(canon_v, live_v, 1, threads)[:, 0]
    out_n = _f32(canon_n)[idx] if canon_n is not None else None
    return canon_v[idx], out_n, idx


def ref_knn(nodes, query, k):
    """k-NN by the reference's vendored nanoflann. Returns (idx, dist_sqr)."""
    R = ref_lib()
    if R is None:
        return None
    nodes, query = _f32(nodes), _f32(query)
    idx = np.empty((len(query), k), np.int32)
    d = np.empty((len(query), k), np.float32)
    R.ref_knn(_p(nodes), len(nodes), _p(query), len(query), k, _p(idx), _p(d))
    return idx, d


def transformation_weight(g, dg_w, v):
    return float(lib().orc_transformation_weight(_p(_f32(g)), dg_w, _p(_f32(v))))


def calc_dqb(node_pos, node_dq, node_w, k, p):
    node_pos, node_dq, node_w = _f32(node_pos), _f32(node_dq), _f32(node_w)
    out = np.empty(8, np.float32)
    lib().orc_calc_dqb(_p(node_pos), _p(node_dq), _p(node_w), len(node_pos), k, _p(_f32(p)), _p(out))
    return out


def warp_to_live(node_pos, node_dq, node_w, k, verts, normals=None, threads=1):
    node_pos, node_dq, node_w, verts = _f32(node_pos), _f32(node_dq), _f32(node_w), _f32(verts)
    ov = np.empty_like(verts)
    on = None
    if normals is not None:
        normals = _f32(normals)
        on = np.empty_like(normals)
    lib().orc_warp_to_live(_p(node_pos), _p(node_dq), _p(node_w), len(node_pos), k, _p(verts), _p(normals),
                           len(verts), _p(ov), _p(on), threads)
    return ov, on


def tukey_weights(node_pos, node_dq, node_w, k, canon, live, tukey_offset, psi_data, threads=1):
    node_pos, node_dq, node_w, canon, live = map(_f32, (node_pos, node_dq, node_w, canon, live))
    out = np.empty(len(canon), np.float32)
    lib().orc_tukey_weights(_p(node_pos), _p(node_dq), _p(node_w), len(node_pos), k, _p(canon), _p(live), len(canon),
                            tukey_offset, psi_data, _p(out), threads)
    return out


def huber_weights(node_pos, node_dq, node_w, k, psi_reg):
    node_pos, node_dq, node_w = map(_f32, (node_pos, node_dq, node_w))
    out = np.empty(len(node_pos), np.float32)
    lib().orc_huber_weights(_p(node_pos), _p(node_dq), _p(node_w), len(node_pos), k, psi_reg, _p(out))
    return out


def solve_ref(node_pos, node_dq, node_w, k, canon, live, num_iter=1, nonlinear_iter=1, linear_iter=256,
              tukey_offset=4.652, psi_data=0.01, lambda_=0.0, psi_reg=1e-4, pcg_tol=0.0, gn_tol=0.0, use_double=True,
              threads=1):
    """Returns (translations Dx3, node_dq_out Dx8, stats dict)."""
    node_pos, node_dq, node_w, canon, live = map(_f32, (node_pos, node_dq, node_w, canon, live))
    D, N = len(node_pos), len(canon)
    prm = SolveParams(num_iter, nonlinear_iter, linear_iter, tukey_offset, psi_data, lambda_, psi_reg, pcg_tol, gn_tol,
                      1 if use_double else 0, threads)
    st = SolveStats()
    t = np.zeros((D, 3), np.float32)
    dq_out = np.zeros((D, 8), np.float32)
    lib().orc_solve_ref(_p(node_pos), _p(node_dq), _p(node_w), D, k, _p(canon), _p(live), N, C.byref(prm), _p(t),
                        _p(dq_out), C.byref(st))
    return t, dq_out, dict(initial_cost=st.initial_cost, final_cost=st.final_cost, gn_iters=st.gn_iters,
                           pcg_iters=st.pcg_iters)


# ------------------------------------------------------------- north-star solve (6-DoF), solve6_oracle.c
class Solve6Params(C.Structure):
    _fields_ = [("num_iter", C.c_int), ("gn_iter", C.c_int), ("linear_iter", C.c_int), ("tukey_offset", C.c_float),
                ("psi_data", C.c_float), ("lambda_", C.c_float), ("psi_reg", C.c_float), ("dist_thresh", C.c_float),
                ("cos_thresh", C.c_float), ("damping", C.c_float), ("pcg_tol", C.c_float), ("pcg_tol_first", C.c_float),
                ("pcg_tol_decay", C.c_float), ("pcg_tol_adapt", C.c_float), ("threads", C.c_int), ("gn_tol", C.c_float),
                ("reuse_matrix", C.c_int)]


ORC6_HIST = 32


class Solve6Stats(C.Structure):
    _fields_ = [("initial_cost", C.c_double), ("final_cost", C.c_double), ("gn_iters", C.c_int), ("pcg_iters", C.c_int),
                ("valid_first", C.c_long), ("valid_last", C.c_long), ("cost_hist", C.c_double * ORC6_HIST),
                ("pcg_rel_hist", C.c_double * ORC6_HIST), ("pcg_it_hist", C.c_int * ORC6_HIST), ("pcg_tol_hist", C.c_double * ORC6_HIST),
                ("valid_hist", C.c_long * ORC6_HIST), ("stop_hist", C.c_int * ORC6_HIST), ("gn_solves", C.c_int),
                ("gn_rejected", C.c_int), ("gn_converged", C.c_int), ("hist_n", C.c_int)]


SOLVE6_DEFAULTS = dict(num_iter=2, gn_iter=3, linear_iter=100, tukey_offset=4.652, psi_data=0.01, lambda_=200.0,
                       psi_reg=1e-4, dist_thresh=0.1, cos_thresh=0.5, damping=1e-4, pcg_tol=1e-6, pcg_tol_first=0.0,
                       pcg_tol_decay=1.0, pcg_tol_adapt=0.0, threads=1, gn_tol=0.0, reuse_matrix=0)


def solve6_params(**kw):
    d = dict(SOLVE6_DEFAULTS)
    d.update(kw)
    return Solve6Params(*[d[n] for n, _ in Solve6Params._fields_])


def _declare6(L):
    vp, i, f = C.c_void_p, C.c_int, C.c_float
    L.orc6_points_normals.argtypes = [vp, i, i, i, f, f, f, f, vp, i, vp, i]
    L.orc6_graph.argtypes = [vp, vp, i, i, vp, i, vp, vp, vp, i]
    L.orc6_warp.argtypes = [vp, i, vp, vp, vp, vp, i, vp, vp]
    L.orc6_data_jacobian.argtypes = [vp, vp, vp, vp, i, vp, vp, vp]
    L.orc6_apply_twist.argtypes = [vp, vp, vp, vp]
    sig = [vp, vp, vp, i, i, vp, vp, i, vp, i, vp, i, i, i, f, f, f, f, C.POINTER(Solve6Params)]
    L.orc6_solve.argtypes = sig + [vp, C.POINTER(Solve6Stats)]
    L.orc6_cost.argtypes = sig + [vp]
    L.orc6_cost.restype = C.c_double
    L._six = True


def lib6():
    L = lib()
    if not getattr(L, "_six", False):
        _declare6(L)
    return L


def points_normals(depth, fx, fy, cx, cy):
    """computePointNormals (imgproc.cu:187-215): (points, normals) float32 (H, W, 4), NaN where undefined"""
    depth = np.ascontiguousarray(depth, np.uint16)
    H, W = depth.shape
    P, Nm = np.empty((H, W, 4), np.float32), np.empty((H, W, 4), np.float32)
    lib6().orc6_points_normals(_p(depth), W * 2, W, H, fx, fy, cx, cy, _p(P), W * 16, _p(Nm), W * 16)
    return P, Nm


def graph6(node_pos, node_w, k, canon, threads=1):
    node_pos, node_w, canon = map(_f32, (node_pos, node_w, canon))
    N, D = len(canon), len(node_pos)
    idx, wn = np.empty((N, k), np.int32), np.empty((N, k), np.float32)
    reg = np.empty((D, k), np.int32)
    lib6().orc6_graph(_p(node_pos), _p(node_w), D, k, _p(canon), N, _p(idx), _p(wn), _p(reg), threads)
    return idx, wn, reg


def warp6(node_dq, idx, wn, canon, canon_n=None):
    node_dq, canon = _f32(node_dq), _f32(canon)
    idx, wn = np.ascontiguousarray(idx, np.int32), _f32(wn)
    out_p = np.empty_like(canon)
    cn = _f32(canon_n) if canon_n is not None else None
    out_n = np.empty_like(canon) if cn is not None else None
    lib6().orc6_warp(_p(node_dq), idx.shape[1], _p(idx), _p(wn), _p(canon), _p(cn), len(canon), _p(out_p), _p(out_n))
    return out_p, out_n


def data_jacobian6(node_pos, node_dq, idx_v, wn_v, c):
    """(J (k, 6, 3) = d p / d twist_j, p (3,)) of one vertex"""
    node_pos, node_dq = _f32(node_pos), _f32(node_dq)
    idx_v, wn_v, c = np.ascontiguousarray(idx_v, np.int32), _f32(wn_v), _f32(c)
    k = len(idx_v)
    J, p = np.zeros((k, 6, 3), np.float64), np.zeros(3, np.float64)
    lib6().orc6_data_jacobian(_p(node_pos), _p(node_dq), _p(idx_v), _p(wn_v), k, _p(c), _p(J), _p(p))
    return J, p


def apply_twist6(node_pos_i, dq, twist):
    out = np.zeros(8, np.float32)
    tw = np.ascontiguousarray(twist, np.float64)
    lib6().orc6_apply_twist(_p(_f32(node_pos_i)), _p(_f32(dq)), _p(tw), _p(out))
    return out


def _solve6_args(node_pos, node_dq, node_w, k, canon, canon_n, vmap, nmap, intr, prm):
    node_pos, node_dq, node_w, canon, vmap, nmap = map(_f32, (node_pos, node_dq, node_w, canon, vmap, nmap))
    cn = _f32(canon_n) if canon_n is not None else None
    H, W = vmap.shape[:2]
    fx, fy, cx, cy = intr
    keep = (node_pos, node_dq, node_w, canon, cn, vmap, nmap)
    args = [_p(node_pos), _p(node_dq), _p(node_w), len(node_pos), k, _p(canon), _p(cn), len(canon), _p(vmap), W * 16,
            _p(nmap), W * 16, W, H, fx, fy, cx, cy, C.byref(prm)]
    return keep, args


def solve6(node_pos, node_dq, node_w, k, canon, canon_n, vmap, nmap, intr, **params):
    """North-star solve.  Returns (node_dq_out D x 8, stats dict)."""
    prm = solve6_params(**params)
    keep, args = _solve6_args(node_pos, node_dq, node_w, k, canon, canon_n, vmap, nmap, intr, prm)
    out = np.zeros((len(keep[0]), 8), np.float32)
    st = Solve6Stats()
    lib6().orc6_solve(*args, _p(out), C.byref(st))
    d = {n: getattr(st, n) for n, _ in Solve6Stats._fields_}
    n = st.hist_n  # slots of the Gauss-Newton loop (gn_tol > 0: + the closing check), skipped ones included
    for name in ("cost_hist", "pcg_rel_hist", "pcg_it_hist", "pcg_tol_hist", "valid_hist", "stop_hist"):
        d[name] = list(d[name])[:n]
    return out, d


def cost6(node_pos, node_dq, node_w, k, canon, canon_n, vmap, nmap, intr, **params):
    prm = solve6_params(**params)
    keep, args = _solve6_args(node_pos, node_dq, node_w, k, canon, canon_n, vmap, nmap, intr, prm)
    nv = C.c_long(0)
    c = lib6().orc6_cost(*args, C.byref(nv))
    return c, nv.value


# --------------------------------------------------------------- depth pre-processing, img_oracle.c
def _declare_img(L):
    vp, i, f = C.c_void_p, C.c_int, C.c_float
    L.orc_exp_neg.argtypes, L.orc_exp_neg.restype = [f], f
    L.orc_bilateral.argtypes = [vp, i, vp, i, i, i, i, f, f]
    L.orc_truncate_depth.argtypes = [vp, i, i, i, f]
    L.orc_depth_pyr.argtypes = [vp, i, i, i, vp, i, f]
    L.orc_normals_mask_depth.argtypes = [vp, i, i, i, f, f, f, f, vp, i]
    L.orc_resize_depth_normals.argtypes = [vp, i, vp, i, i, i, vp, i, vp, i]
    L.orc_resize_points_normals.argtypes = [vp, i, vp, i, i, i, vp, i, vp, i]
    L._img = True


def _libimg():
    L = lib()
    if not getattr(L, "_img", False):
        _declare_img(L)
    return L


def bilateral(depth, ksz, sigma_spatial, sigma_depth):
    depth = np.ascontiguousarray(depth, np.uint16)
    H, W = depth.shape
    out = np.zeros_like(depth)
    _libimg().orc_bilateral(_p(depth), W * 2, _p(out), W * 2, W, H, ksz, sigma_spatial, sigma_depth)
    return out


def truncate_depth(depth, max_dist):
    out = np.ascontiguousarray(depth, np.uint16).copy()
    H, W = out.shape
    _libimg().orc_truncate_depth(_p(out), W * 2, W, H, max_dist)
    return out


def depth_pyr(depth, sigma_depth):
    depth = np.ascontiguousarray(depth, np.uint16)
    H, W = depth.shape
    out = np.zeros((H // 2, W // 2), np.uint16)
    if out.size:
        _libimg().orc_depth_pyr(_p(depth), W * 2, W, H, _p(out), (W // 2) * 2, sigma_depth)
    return out


def normals_mask_depth(depth, fx, fy, cx, cy):
    """returns (masked depth copy, normals (H, W, 4))"""
    d = np.ascontiguousarray(depth, np.uint16).copy()
    H, W = d.shape
    n = np.zeros((H, W, 4), np.float32)
    _libimg().orc_normals_mask_depth(_p(d), W * 2, W, H, fx, fy, cx, cy, _p(n), W * 16)
    return d, n


def resize_depth_normals(depth, normals):
    depth, normals = np.ascontiguousarray(depth, np.uint16), _f32(normals)
    H, W = depth.shape
    d, n = np.zeros((H // 2, W // 2), np.uint16), np.zeros((H // 2, W // 2, 4), np.float32)
    if d.size:
        _libimg().orc_resize_depth_normals(_p(depth), W * 2, _p(normals), W * 16, W, H, _p(d), (W // 2) * 2, _p(n), (W // 2) * 16)
    return d, n


def resize_points_normals(points, normals):
    points, normals = _f32(points), _f32(normals)
    H, W = points.shape[:2]
    v, n = np.zeros((H // 2, W // 2, 4), np.float32), np.zeros((H // 2, W // 2, 4), np.float32)
    if v.size:
        _libimg().orc_resize_points_normals(_p(points), W * 16, _p(normals), W * 16, W, H, _p(v), (W // 2) * 16, _p(n), (W // 2) * 16)
    return v, n


# ------------------------------------------------------------------ node insertion (warp_oracle.c)
def unsupported_flags(node_pos, node_w, k, verts, threads=1):
    verts = _f32(verts)
    D = 0 if node_pos is None else len(node_pos)
    flags = np.zeros(len(verts), np.uint8)
    L = lib()
    L.orc_unsupported_flags.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    L.orc_unsupported_flags(_p(_f32(node_pos)) if D else None, _p(_f32(node_w)) if D else None, D, k, _p(verts),
                            len(verts), _p(flags), threads)
    return flags


def voxel_grid(points, leaf):
    points = _f32(points).reshape(-1, 3)
    out = np.zeros((max(len(points), 1), 3), np.float32)
    L = lib()
    L.orc_voxel_grid.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_void_p]
    L.orc_voxel_grid.restype = C.c_int
    m = L.orc_voxel_grid(_p(points), len(points), leaf, _p(out))
    return out[:m].copy()


# ------------------------------------------------------------------------- rigid ICP (icp_oracle.c)
def icp_sums(curr, ncurr, prev, nprev, aff12, intr, dist_thres=0.1, angle_thres=0.3490658503988659):
    """27 sums + number of matched pixels of one linearisation; curr/prev: uint16 depth (H, W) or float32 (H, W, 4)"""
    depth_variant = curr.dtype == np.uint16
    curr, prev = np.ascontiguousarray(curr), np.ascontiguousarray(prev)
    ncurr, nprev, aff = _f32(ncurr), _f32(nprev), _f32(aff12)
    H, W = curr.shape[:2]
    step = W * (2 if depth_variant else 16)
    sums = np.zeros(27, np.float64)
    m = C.c_long(0)
    L = lib()
    L.orc_icp_sums.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                               C.c_int, C.c_int, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float,
                               C.c_float, C.c_void_p, C.c_void_p]
    fx, fy, cx, cy = intr
    L.orc_icp_sums(1 if depth_variant else 0, _p(curr), step, _p(ncurr), W * 16, _p(prev), step, _p(nprev), W * 16, W, H,
                   _p(aff), fx, fy, cx, cy, dist_thres, angle_thres, _p(sums), C.byref(m))
    return sums, m.value
