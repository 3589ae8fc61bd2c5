"""ctypes binding of include/dynfu_amd.h.  Tensors are torch CUDA tensors (device memory +
stream plumbing only); every function launches on torch's current stream."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

# every symbol include/dynfu_amd.h declares (tests/test_capi_symbols.py checks the .so exports them)
SYMBOLS = [
    "dfa_last_error", "dfa_version", "dfa_abi_version", "dfa_abi_struct_size", "dfa_compute_dists", "dfa_tsdf_clear", "dfa_tsdf_integrate",
    "dfa_tsdf_clear_integrate", "dfa_tsdf_raycast_points", "dfa_tsdf_raycast_depth", "dfa_tsdf_raycast_tally", "dfa_tsdf_vertex_normals", "dfa_correspond_projective", "dfa_knn", "dfa_warp_to_live",
    "dfa_calc_dqb", "dfa_unsupported_vertices", "dfa_icp_sums", "dfa_repack_points", "dfa_compact_points", "dfa_transform_points", "dfa_warp_to_live_graph",
    "dfa_correspond", "dfa_marching_cubes", "dfa_mc_default_tables",
    "dfa_tsdf_occupancy_bytes", "dfa_tsdf_clear_occ", "dfa_tsdf_integrate_occ", "dfa_tsdf_clear_integrate_occ", "dfa_tsdf_clear_integrate_known_occ", "dfa_marching_cubes_occ",
    "dfa_depth_bilateral_filter", "dfa_depth_truncate", "dfa_depth_build_pyramid", "dfa_compute_normals_mask_depth",
    "dfa_resize_depth_normals", "dfa_resize_points_normals",
    "dfa_compute_points_normals", "dfa_solver6_create", "dfa_solver6_destroy", "dfa_solver6_set_problem",
    "dfa_solver6_solve", "dfa_solver6_set_node_transforms", "dfa_solver6_node_dq", "dfa_solver6_warp", "dfa_solver6_get_stats",
    "dfa_solver6_enable_timing", "dfa_solver6_get_timing",
    "dfa_solver_create", "dfa_solver_destroy", "dfa_solver_set_problem", "dfa_solver_solve", "dfa_solver_set_deterministic", "dfa_solver_matrix_entries", "dfa_solver_matrix_row_lengths", "dfa_solver_gradient",
    "dfa_solver_translations", "dfa_solver_node_dq", "dfa_solver_tukey_weights", "dfa_solver_huber_weights",
    "dfa_solver_data_graph", "dfa_solver_reg_graph", "dfa_solver_get_stats", "dfa_solver_enable_timing",
    "dfa_solver_get_timing", "dfa_solver_warp_to_live", "dfa_solver_set_overlap_callback", "dfa_solver_team_pcg_info",
]


_OVERLAP_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_int)  # dfa_overlap_fn


class Solve6Params(C.Structure):
    """dfa_solve6_params (north-star solve)"""
    _fields_ = [("num_iter", C.c_int), ("gn_iter", C.c_int), ("linear_iter", C.c_int), ("tukey_offset", C.c_float),
                ("psi_data", C.c_float), ("lambda_", C.c_float), ("psi_reg", C.c_float), ("dist_thresh", C.c_float),
                ("cos_thresh", C.c_float), ("damping", C.c_float), ("pcg_tol", C.c_float), ("pcg_tol_first", C.c_float),
                ("pcg_tol_decay", C.c_float), ("pcg_tol_adapt", C.c_float), ("adaptive_launch", C.c_int), ("gn_tol", C.c_float)]

    DEFAULTS = dict(num_iter=2, gn_iter=3, linear_iter=100, tukey_offset=4.652, psi_data=0.01, lambda_=200.0,
                    psi_reg=1e-4, dist_thresh=0.1, cos_thresh=0.5, damping=1e-4, pcg_tol=1e-6, pcg_tol_first=0.0,
                    pcg_tol_decay=1.0, pcg_tol_adapt=0.0, adaptive_launch=0, gn_tol=0.0)

    def __init__(self, **kw):
        d = dict(self.DEFAULTS)
        d.update(kw)
        super().__init__(*[d[n] for n, _ in self._fields_])


SOLVE6_HIST = 32  # DFA_SOLVE6_HIST
ABI_VERSION = 6   # DFA_ABI_VERSION


class _Solve6Stats(C.Structure):
    _fields_ = [("initial_cost", C.c_double), ("final_cost", C.c_double), ("gn_iters", C.c_int), ("pcg_iters", C.c_int),
                ("gn_solves", C.c_int), ("gn_rejected", C.c_int), ("gn_converged", C.c_int), ("hist_n", C.c_int),
                ("valid_first", C.c_longlong), ("valid_last", C.c_longlong), ("max_row_blocks", C.c_int),
                ("overflow", C.c_int), ("pcg_short", C.c_int), ("pcg_launches", C.c_int), ("cost_hist", C.c_double * SOLVE6_HIST), ("pcg_rel_hist", C.c_float * SOLVE6_HIST),
                ("pcg_it_hist", C.c_int * SOLVE6_HIST), ("pcg_tol_hist", C.c_float * SOLVE6_HIST),
                ("valid_hist", C.c_longlong * SOLVE6_HIST), ("stop_hist", C.c_int * SOLVE6_HIST)]


class _Solve6Timing(C.Structure):
    _fields_ = [("linearise_ms", C.c_float), ("assemble_ms", C.c_float), ("pcg_ms", C.c_float),
                ("gn_iterations", C.c_int), ("matrix_blocks", C.c_longlong)]


class DynfuAmdError(RuntimeError):
    pass


class SolveParams(C.Structure):
    """dfa_solve_params"""
    _fields_ = [("num_iter", C.c_int), ("nonlinear_iter", C.c_int), ("linear_iter", C.c_int),
                ("tukey_offset", C.c_float), ("psi_data", C.c_float), ("lambda_", C.c_float), ("psi_reg", C.c_float),
                ("pcg_tol", C.c_float), ("gn_tol", C.c_float)]


class _SolveStats(C.Structure):
    _fields_ = [("initial_cost", C.c_double), ("final_cost", C.c_double), ("gn_iters", C.c_int),
                ("pcg_iters", C.c_int), ("max_row_nnz", C.c_int), ("gn_noop", C.c_int)]


class _SolveTiming(C.Structure):
    _fields_ = [("pcg_ms", C.c_float), ("assemble_ms", C.c_float), ("pcg_launches", C.c_int),
                ("assemble_launches", C.c_int), ("matrix_nnz", C.c_longlong), ("pcg_iters", C.c_longlong),
                ("solves", C.c_int), ("reserved", C.c_int)]


def lib_path():
    # DFA_LIB_PATH: a library built with other compile-time constants (A/B measurements, tools/ab_variant.sh)
    return os.environ.get("DFA_LIB_PATH") or os.path.join(_HERE, "libdynfu_amd.so")


def dev_lib_path():
    """the development flavour (-DDFA_DEV_AB: environment A/B switches and the non-default kernel variants)"""
    return os.path.join(_HERE, "libdynfu_amd_dev.so")


class use_library:
    """`with use_library(dev_lib_path()):` — every call of this module goes to another build of the library inside the
    block (tests that compare kernel variants, A/B scripts); the product library is what everything else uses."""

    def __init__(self, path):
        self.path = path

    def __enter__(self):
        global _LIB
        self.saved = _LIB
        try:
            if self.path not in _LOADED:
                _LIB = None
                load(self.path)
            _LIB = _LOADED[self.path]
        except BaseException:
            _LIB = self.saved  # (the other build is missing: the module keeps the library it had)
            raise
        return _LIB

    def __exit__(self, *exc):
        global _LIB
        _LIB = self.saved
        return False


_LOADED = {}


def load(path=None):
    """Loads libdynfu_amd.so; raises (never falls back) when it has not been built."""
    global _LIB
    if _LIB is not None and path is None:
        return _LIB
    # torch first: its bundled libamdhip64 (SONAME libamdhip64.so.7) must be the HIP runtime this
    # library binds to, so that torch's streams and device pointers are valid inside it
    import torch  # noqa: F401
    p = path or lib_path()
    if not os.path.exists(p):
        raise DynfuAmdError("%s not found: build it with `python dynfu_amd/build.py` "
                            "(__graft_entry__.build()). There is no CPU fallback." % p)
    L = C.CDLL(p)
    vp, i, f = C.c_void_p, C.c_int, C.c_float
    L.dfa_last_error.restype = C.c_char_p
    L.dfa_version.restype = C.c_char_p
    # ABI guard: the ctypes mirrors above are kept in step with include/dynfu_amd.h by hand
    L.dfa_abi_struct_size.restype = C.c_size_t
    L.dfa_abi_struct_size.argtypes = [i]
    if L.dfa_abi_version() != ABI_VERSION:
        raise DynfuAmdError("%s implements ABI version %d, this binding expects %d" % (p, L.dfa_abi_version(), ABI_VERSION))
    for sid, mirror in enumerate((SolveParams, _SolveStats, _SolveTiming, Solve6Params, _Solve6Stats, _Solve6Timing)):
        if L.dfa_abi_struct_size(sid) != C.sizeof(mirror):
            raise DynfuAmdError("struct %d: the library assumes %d bytes, the ctypes mirror %s has %d"
                                % (sid, L.dfa_abi_struct_size(sid), mirror.__name__, C.sizeof(mirror)))
    L.dfa_compute_dists.argtypes = [vp, i, vp, i, i, i, f, f, f, f, vp]
    L.dfa_tsdf_clear.argtypes = [vp, i, i, i, vp]
    integ = [vp, i, i, i, vp, i, i, i, vp, f, i, vp, f, f, f, f, vp]
    L.dfa_tsdf_integrate.argtypes = integ
    L.dfa_tsdf_clear_integrate.argtypes = integ
    L.dfa_tsdf_integrate_occ.argtypes = integ[:-1] + [vp, vp]
    L.dfa_tsdf_clear_integrate_occ.argtypes = integ[:-1] + [vp, vp]
    L.dfa_tsdf_clear_integrate_known_occ.argtypes = integ[:-1] + [vp, vp]
    L.dfa_tsdf_clear_occ.argtypes = [vp, i, i, i, vp, vp]
    L.dfa_tsdf_occupancy_bytes.argtypes = [i, i, i]
    L.dfa_tsdf_occupancy_bytes.restype = C.c_size_t
    ray = [vp, i, i, i, vp, f, vp, vp, f, f, f, f, f, f, vp, i, vp, i, i, i, vp]
    L.dfa_tsdf_raycast_points.argtypes = ray
    L.dfa_tsdf_raycast_depth.argtypes = ray
    L.dfa_tsdf_raycast_tally.argtypes = [vp, i, i, i, vp, f, vp, vp, f, f, f, f, f, f, i, i, vp, vp, vp]
    L.dfa_tsdf_vertex_normals.argtypes = [vp, i, i, i, vp, f, vp, i, vp, vp]
    L.dfa_correspond_projective.argtypes = [vp, vp, i, vp, i, vp, i, i, i, f, f, f, f, f, f, vp, vp, vp, vp]
    L.dfa_knn.argtypes = [vp, vp, i, vp, i, i, vp, vp, vp]
    L.dfa_warp_to_live.argtypes = [vp, vp, vp, i, i, vp, vp, i, vp, vp, vp]
    L.dfa_depth_bilateral_filter.argtypes = [vp, i, vp, i, i, i, i, f, f, vp]
    L.dfa_depth_truncate.argtypes = [vp, i, i, i, f, vp]
    L.dfa_depth_build_pyramid.argtypes = [vp, i, i, i, vp, i, f, vp]
    L.dfa_compute_normals_mask_depth.argtypes = [vp, i, i, i, f, f, f, f, vp, i, vp]
    L.dfa_resize_depth_normals.argtypes = [vp, i, vp, i, i, i, vp, i, vp, i, vp]
    L.dfa_resize_points_normals.argtypes = [vp, i, vp, i, i, i, vp, i, vp, i, vp]
    L.dfa_compute_points_normals.argtypes = [vp, i, i, i, f, f, f, f, vp, i, vp, i, vp]
    L.dfa_solver6_create.argtypes = [i, i, i, C.POINTER(vp)]
    L.dfa_solver6_destroy.argtypes = [vp]
    L.dfa_solver6_destroy.restype = None
    L.dfa_solver6_set_problem.argtypes = [vp, vp, vp, vp, i, vp, vp, i, vp]
    L.dfa_solver6_set_node_transforms.argtypes = [vp, vp]
    L.dfa_solver6_solve.argtypes = [vp, vp, i, vp, i, i, i, f, f, f, f, C.POINTER(Solve6Params), vp]
    L.dfa_solver6_node_dq.argtypes = [vp]
    L.dfa_solver6_node_dq.restype = vp
    L.dfa_solver6_warp.argtypes = [vp, vp, vp, vp]
    L.dfa_solver6_get_stats.argtypes = [vp, C.POINTER(_Solve6Stats), vp]
    L.dfa_solver6_enable_timing.argtypes = [vp, i]
    L.dfa_solver6_get_timing.argtypes = [vp, C.POINTER(_Solve6Timing), vp]
    L.dfa_marching_cubes.argtypes = [vp, i, i, i, vp, vp, vp, vp, i, vp, vp]
    L.dfa_marching_cubes_occ.argtypes = [vp, vp, i, i, i, vp, vp, vp, vp, i, vp, vp]
    L.dfa_mc_default_tables.argtypes = [vp, vp]
    L.dfa_icp_sums.argtypes = [i, vp, i, vp, i, vp, i, vp, i, i, i, vp, f, f, f, f, f, f, vp, vp, vp]
    L.dfa_calc_dqb.argtypes = [vp, vp, vp, i, i, vp, i, vp, vp]
    L.dfa_unsupported_vertices.argtypes = [vp, vp, i, i, vp, i, vp, vp]
    L.dfa_correspond.argtypes = [vp, vp, i, vp, i, vp, vp, vp, vp]
    L.dfa_repack_points.argtypes = [vp, i, vp, i, i, C.c_float, vp]
    L.dfa_compact_points.argtypes = [vp, vp, i, vp, vp, vp, vp]
    L.dfa_warp_to_live_graph.argtypes = [vp, vp, vp, i, i, vp, vp, vp, i, vp, vp, vp]
    L.dfa_transform_points.argtypes = [vp, i, C.POINTER(C.c_float), i, vp, vp]
    L.dfa_solver_create.argtypes = [i, i, i, C.POINTER(vp)]
    L.dfa_solver_destroy.argtypes = [vp]
    L.dfa_solver_destroy.restype = None
    L.dfa_solver_set_problem.argtypes = [vp, vp, vp, vp, i, vp, vp, vp, vp, i, vp]
    L.dfa_solver_solve.argtypes = [vp, C.POINTER(SolveParams), vp]
    L.dfa_solver_set_deterministic.argtypes = [vp, i]
    for n in ("translations", "node_dq", "tukey_weights", "huber_weights", "data_graph", "reg_graph", "matrix_entries",
              "matrix_row_lengths", "gradient"):
        fn = getattr(L, "dfa_solver_" + n)
        fn.argtypes = [vp]
        fn.restype = vp
    L.dfa_solver_warp_to_live.argtypes = [vp, vp, vp, vp, vp]
    L.dfa_solver_get_stats.argtypes = [vp, C.POINTER(_SolveStats), vp]
    L.dfa_solver_enable_timing.argtypes = [vp, i]
    L.dfa_solver_set_overlap_callback.argtypes = [vp, _OVERLAP_FN, vp]
    L.dfa_solver_get_timing.argtypes = [vp, C.POINTER(_SolveTiming), vp]
    L.dfa_solver_team_pcg_info.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    runtimes = set()
    with open("/proc/self/maps") as maps:
        for line in maps:
            if "libamdhip64" in line:
                runtimes.add(line.split()[-1])
    if len(runtimes) > 1:
        raise DynfuAmdError("two HIP runtimes are mapped in this process: %s" % sorted(runtimes))
    _LIB = L
    _LOADED[p] = L
    return L


def version():
    return load().dfa_version().decode()


def _check(rc):
    if rc != 0:
        raise DynfuAmdError("dynfu_amd error %d: %s" % (rc, load().dfa_last_error().decode()))


def _torch():
    import torch
    if not torch.cuda.is_available():
        raise DynfuAmdError("no GPU visible: dynfu_amd has no CPU path")
    return torch


def _stream():
    return C.c_void_p(_torch().cuda.current_stream().cuda_stream)


def _dev(t, dtype=None, name="tensor"):
    torch = _torch()
    if t is None:
        return None
    if not (isinstance(t, torch.Tensor) and t.is_cuda):
        raise DynfuAmdError("%s must be a CUDA tensor" % name)
    if dtype is not None and t.dtype != dtype:
        raise DynfuAmdError("%s must have dtype %s, got %s" % (name, dtype, t.dtype))
    return C.c_void_p(t.data_ptr())


def _farr(vals, n):
    a = (C.c_float * n)(*[float(v) for v in vals])
    return a


def _aff12(m):
    """accepts 12 floats (R row-major then t) or a (3x4)/(4x4) array [R|t]"""
    import numpy as np
    m = np.asarray(m, dtype=np.float32)
    if m.size == 12 and m.ndim == 1:
        return _farr(m, 12)
    m = m.reshape(-1, 4)[:3]
    return _farr(list(m[:, :3].reshape(-1)) + list(m[:, 3]), 12)


# ------------------------------------------------------------------------------- TSDF seam
def compute_dists(depth, dists, fx, fy, cx, cy):
    torch = _torch()
    rows, cols = depth.shape
    _check(load().dfa_compute_dists(_dev(depth, torch.uint16, "depth"), depth.stride(0) * 2,
                                    _dev(dists, torch.uint16, "dists"), dists.stride(0) * 2, cols, rows, fx, fy, cx, cy,
                                    _stream()))


def _vol_dims(vol):
    torch = _torch()
    if vol.dtype not in (torch.uint32, torch.int32) or vol.dim() != 3 or not vol.is_contiguous():
        raise DynfuAmdError("volume must be a contiguous (Z, Y, X) 32-bit tensor")
    Z, Y, X = vol.shape
    return X, Y, Z


def tsdf_occupancy(vol):
    """a fresh occupancy map for `vol` (dfa_tsdf_occupancy_bytes; uint8 CUDA tensor (ceil(Z/8), ceil(Y/2), ceil(X/32)))"""
    torch = _torch()
    X, Y, Z = _vol_dims(vol)
    n = load().dfa_tsdf_occupancy_bytes(X, Y, Z)
    shape = ((Z + 7) // 8, (Y + 1) // 2, (X + 31) // 32)
    assert n == shape[0] * shape[1] * shape[2]
    return torch.zeros(shape, dtype=torch.uint8, device=vol.device)


def tsdf_clear(vol, occupancy=None):
    X, Y, Z = _vol_dims(vol)
    if occupancy is None:
        _check(load().dfa_tsdf_clear(_dev(vol), X, Y, Z, _stream()))
    else:
        _check(load().dfa_tsdf_clear_occ(_dev(vol), X, Y, Z, _dev(occupancy, _torch().uint8, "occupancy"), _stream()))


def _integrate(fn, vol, dists, voxel_size, trunc, max_weight, vol2cam, fx, fy, cx, cy, occupancy=None):
    torch = _torch()
    X, Y, Z = _vol_dims(vol)
    rows, cols = dists.shape
    occ = () if occupancy is None else (_dev(occupancy, torch.uint8, "occupancy"),)
    _check(fn(_dev(dists, torch.uint16, "dists"), dists.stride(0) * 2, cols, rows, _dev(vol), X, Y, Z,
              _farr(voxel_size, 3), trunc, max_weight, _aff12(vol2cam), fx, fy, cx, cy, *occ, _stream()))


def tsdf_integrate(vol, dists, voxel_size, trunc, max_weight, vol2cam, fx, fy, cx, cy, occupancy=None):
    L = load()
    _integrate(L.dfa_tsdf_integrate if occupancy is None else L.dfa_tsdf_integrate_occ, vol, dists, voxel_size, trunc, max_weight,
               vol2cam, fx, fy, cx, cy, occupancy)


def tsdf_clear_integrate(vol, dists, voxel_size, trunc, max_weight, vol2cam, fx, fy, cx, cy, occupancy=None, occupancy_known=False):
    """occupancy_known: the map describes the volume as it is now (left by tsdf_clear(vol, occupancy) or an occupancy-keeping
    sweep, nothing else has written the volume since): boxes of zeros that stay zeros are not stored again
    (dfa_tsdf_clear_integrate_known_occ)."""
    L = load()
    fn = L.dfa_tsdf_clear_integrate if occupancy is None else (L.dfa_tsdf_clear_integrate_known_occ if occupancy_known else L.dfa_tsdf_clear_integrate_occ)
    _integrate(fn, vol, dists, voxel_size, trunc, max_weight, vol2cam, fx, fy, cx, cy, occupancy)


def tsdf_raycast_points(vol, voxel_size, trunc, cam2vol, Rinv, fx, fy, cx, cy, step_factor, delta_factor, points,
                        normals):
    torch = _torch()
    X, Y, Z = _vol_dims(vol)
    rows, cols = points.shape[:2]
    _check(load().dfa_tsdf_raycast_points(_dev(vol), X, Y, Z, _farr(voxel_size, 3), trunc, _aff12(cam2vol),
                                          _farr(list(map(float, _flat(Rinv))), 9), fx, fy, cx, cy, step_factor,
                                          delta_factor, _dev(points, torch.float32, "points"), points.stride(0) * 4,
                                          _dev(normals, torch.float32, "normals"), normals.stride(0) * 4, cols, rows,
                                          _stream()))


def tsdf_raycast_depth(vol, voxel_size, trunc, cam2vol, Rinv, fx, fy, cx, cy, step_factor, delta_factor, depth,
                       normals):
    torch = _torch()
    X, Y, Z = _vol_dims(vol)
    rows, cols = depth.shape[:2]
    _check(load().dfa_tsdf_raycast_depth(_dev(vol), X, Y, Z, _farr(voxel_size, 3), trunc, _aff12(cam2vol),
                                         _farr(list(map(float, _flat(Rinv))), 9), fx, fy, cx, cy, step_factor,
                                         delta_factor, _dev(depth, torch.uint16, "depth"), depth.stride(0) * 2,
                                         _dev(normals, torch.float32, "normals"), normals.stride(0) * 4, cols, rows,
                                         _stream()))


def tsdf_raycast_tally(vol, voxel_size, trunc, cam2vol, Rinv, fx, fy, cx, cy, step_factor, delta_factor, cols, rows,
                       unique=True):
    """work of one raycast (measurement): dict(rays_entered, march_fetches, hits, trilinear_fetches, unique_voxels)"""
    torch = _torch()
    X, Y, Z = _vol_dims(vol)
    counts = torch.zeros(4, dtype=torch.int64, device=vol.device)
    bits = torch.zeros((X * Y * Z + 31) // 32, dtype=torch.int32, device=vol.device) if unique else None
    _check(load().dfa_tsdf_raycast_tally(_dev(vol), X, Y, Z, _farr(voxel_size, 3), trunc, _aff12(cam2vol),
                                         _farr(list(map(float, _flat(Rinv))), 9), fx, fy, cx, cy, step_factor, delta_factor,
                                         cols, rows, counts.data_ptr(), bits.data_ptr() if unique else None, _stream()))
    c = counts.cpu().tolist()
    out = dict(rays_entered=c[0], march_fetches=c[1], hits=c[2], trilinear_fetches=c[3], unique_voxels=None)
    if unique:
        n, chunk = 0, 1 << 24
        for o in range(0, bits.numel(), chunk):  # population count in pieces (no 8x blow-up of a 128 MiB bitmap)
            b = bits[o:o + chunk]
            b = (b & 0x55555555) + ((b >> 1) & 0x55555555)
            b = (b & 0x33333333) + ((b >> 2) & 0x33333333)
            b = (b + (b >> 4)) & 0x0F0F0F0F
            n += int(((b * 0x01010101) >> 24 & 0xFF).sum())
        out["unique_voxels"] = n
        # distinct 64-byte (16 voxels) and 128-byte (32 voxels) lines of the volume those voxels lie in: what a cache that
        # never fetched a line twice would have to read
        out["unique_lines_64B"] = int((bits.view(torch.int16) != 0).sum())
        out["unique_lines_128B"] = int((bits != 0).sum())
    return out


def tsdf_vertex_normals(vol, voxel_size, delta_factor, points):
    """normals of surface points (n x 4 float32 CUDA tensor, the volume's metric frame — what marching_cubes returns)
    from the TSDF gradient; returns an n x 4 tensor (xyz, 0), NaN where the gradient stencil leaves the volume"""
    torch = _torch()
    X, Y, Z = _vol_dims(vol)
    n = int(points.shape[0])
    normals = torch.empty((max(n, 1), 4), dtype=torch.float32, device=vol.device)
    _check(load().dfa_tsdf_vertex_normals(_dev(vol), X, Y, Z, _farr(voxel_size, 3), float(delta_factor),
                                          _dev(points, torch.float32, "points") if n else None, n,
                                          _dev(normals) if n else None, _stream()))
    return normals[:n]


# ----------------------------------------------------------------------- marching-cubes seam
def mc_default_tables():
    """(tri_table 256x16, num_verts_table 256) int32 numpy arrays of dfa_mc_default_tables."""
    import numpy as np
    tri, nv = np.zeros((256, 16), np.int32), np.zeros(256, np.int32)
    _check(load().dfa_mc_default_tables(tri.ctypes.data_as(C.c_void_p), nv.ctypes.data_as(C.c_void_p)))
    return tri, nv


def marching_cubes(vol, cell_size, tri_table, num_verts_table, max_vertices, occupancy=None):
    """cuda::MarchingCubes::run.  tri_table / num_verts_table: int32 CUDA tensors (256x16, 256).
    Returns (points (max_vertices, 4) float32 CUDA tensor, total int32 CUDA tensor of 1 element);
    only the first min(total, max_vertices) points are written.  occupancy: the volume's occupancy map (tsdf_occupancy,
    kept by the *_occ sweeps) — the same output without reading the empty part of the volume."""
    torch = _torch()
    X, Y, Z = _vol_dims(vol)
    pts = torch.empty((max(max_vertices, 1), 4), dtype=torch.float32, device=vol.device)
    total = torch.zeros((1,), dtype=torch.int32, device=vol.device)
    tail = (_farr(cell_size, 3), _dev(tri_table, torch.int32, "tri_table"), _dev(num_verts_table, torch.int32, "num_verts_table"),
            _dev(pts) if max_vertices > 0 else None, max_vertices, _dev(total), _stream())
    if occupancy is None:
        _check(load().dfa_marching_cubes(_dev(vol), X, Y, Z, *tail))
    else:
        _check(load().dfa_marching_cubes_occ(_dev(vol), _dev(occupancy, torch.uint8, "occupancy"), X, Y, Z, *tail))
    return pts, total


def _flat(m):
    import numpy as np
    return np.asarray(m, dtype=np.float32).reshape(-1)


def correspond_projective(vertices, normals, vmap, nmap, fx, fy, cx, cy, dist_thresh, min_cosine):
    """projective association of n vertices (n x 3, camera frame) into the live maps (rows x cols x 4 float32 CUDA
    tensors, nmap / normals may be None).  Returns (live vertices n x 3, live normals n x 3 or None, pixel index n)."""
    torch = _torch()
    n = int(vertices.shape[0])
    rows, cols = vmap.shape[:2]
    out_v = torch.empty((n, 3), dtype=torch.float32, device=vmap.device)
    out_n = torch.empty((n, 3), dtype=torch.float32, device=vmap.device) if nmap is not None else None
    pix = torch.empty((n,), dtype=torch.int32, device=vmap.device)
    _check(load().dfa_correspond_projective(_dev(vertices, torch.float32, "vertices") if n else None,
                                            _dev(normals, torch.float32, "normals"), n, _dev(vmap, torch.float32, "vmap"),
                                            vmap.stride(0) * 4, _dev(nmap, torch.float32, "nmap"),
                                            nmap.stride(0) * 4 if nmap is not None else 0, cols, rows, fx, fy, cx, cy,
                                            dist_thresh, min_cosine, _dev(out_v), _dev(out_n), _dev(pix), _stream()))
    return out_v, out_n, pix


# -------------------------------------------------------------------------- warp-field seam
def knn(node_pos, node_w, query, k, want_weights=True):
    torch = _torch()
    D, n = node_pos.shape[0], query.shape[0]
    idx = torch.empty((n, k), dtype=torch.int32, device=query.device)
    w = torch.empty((n, k), dtype=torch.float32, device=query.device) if want_weights else None
    _check(load().dfa_knn(_dev(node_pos, torch.float32, "node_pos"), _dev(node_w, torch.float32, "node_w"), D,
                          _dev(query, torch.float32, "query"), n, k, _dev(idx), _dev(w), _stream()))
    return idx, w


def warp_to_live(node_pos, node_dq, node_w, k, verts, normals=None):
    torch = _torch()
    out_v = torch.empty_like(verts)
    out_n = torch.empty_like(normals) if normals is not None else None
    _check(load().dfa_warp_to_live(_dev(node_pos, torch.float32, "node_pos"), _dev(node_dq, torch.float32, "node_dq"),
                                   _dev(node_w, torch.float32, "node_w"), node_pos.shape[0], k,
                                   _dev(verts, torch.float32, "verts"), _dev(normals, torch.float32, "normals"),
                                   verts.shape[0], _dev(out_v), _dev(out_n), _stream()))
    return out_v, out_n


def warp_to_live_graph(node_pos, node_dq, node_w, idx, verts, normals=None):
    """dfa_warp_to_live_graph: the warp of dfa_warp_to_live with the k-NN indices given (idx (N, k) int32)"""
    torch = _torch()
    N, k = idx.shape
    out_v = torch.empty_like(verts)
    out_n = torch.empty_like(verts) if normals is not None else None
    f32 = torch.float32
    _check(load().dfa_warp_to_live_graph(_dev(node_pos, f32, "node_pos"), _dev(node_dq, f32, "node_dq"), _dev(node_w, f32, "node_w"),
                                         node_pos.shape[0], k, _dev(idx, torch.int32, "idx"), _dev(verts, f32, "verts"),
                                         _dev(normals, f32, "normals"), N, out_v.data_ptr(),
                                         out_n.data_ptr() if out_n is not None else None, _stream()))
    return out_v, out_n


def calc_dqb(node_pos, node_dq, node_w, k, points):
    """Warpfield::calcDQB at n points -> (n, 8) dual quaternions"""
    torch = _torch()
    out = torch.empty((points.shape[0], 8), dtype=torch.float32, device=points.device)
    _check(load().dfa_calc_dqb(_dev(node_pos, torch.float32, "node_pos"), _dev(node_dq, torch.float32, "node_dq"),
                               _dev(node_w, torch.float32, "node_w"), node_pos.shape[0], k,
                               _dev(points, torch.float32, "points"), points.shape[0], _dev(out), _stream()))
    return out


def unsupported_vertices(node_pos, node_w, k, verts):
    """Warpfield::getUnsupportedVertices -> uint8 flags (N,)"""
    torch = _torch()
    flags = torch.empty((verts.shape[0],), dtype=torch.uint8, device=verts.device)
    D = 0 if node_pos is None else node_pos.shape[0]
    _check(load().dfa_unsupported_vertices(_dev(node_pos, torch.float32, "node_pos") if D else None,
                                           _dev(node_w, torch.float32, "node_w") if D else None, D, k,
                                           _dev(verts, torch.float32, "verts"), verts.shape[0], _dev(flags), _stream()))
    return flags


def repack_points(src, dst_stride, pad=1.0):
    """(n, s) float32 -> (n, dst_stride): xyz of every point, remaining floats = pad (dfa_repack_points)"""
    torch = _torch()
    n, s = src.shape
    dst = torch.empty((n, dst_stride), dtype=torch.float32, device=src.device)
    _check(load().dfa_repack_points(_dev(src, torch.float32, "src"), s, dst.data_ptr(), dst_stride, n, pad, _stream()))
    return dst


def transform_points(points, aff12, with_translation=True):
    """R p (+ t) of (n, 3) points (dfa_transform_points); aff12 = R row-major then t"""
    torch = _torch()
    out = torch.empty_like(points)
    _check(load().dfa_transform_points(_dev(points, torch.float32, "points"), points.shape[0], _farr(aff12, 12),
                                       1 if with_translation else 0, out.data_ptr(), _stream()))
    return out


def compact_points(points, flags, want_index=True):
    """the points with a non-zero flag, in index order (dfa_compact_points) -> (points (m, 3), index (m,) | None)"""
    torch = _torch()
    N = flags.shape[0]
    out = torch.empty((max(N, 1), 3), dtype=torch.float32, device=flags.device)
    idx = torch.empty((max(N, 1),), dtype=torch.int32, device=flags.device) if want_index else None
    count = torch.zeros((1,), dtype=torch.int32, device=flags.device)
    _check(load().dfa_compact_points(_dev(points, torch.float32, "points") if N else None,
                                     _dev(flags, torch.uint8, "flags") if N else None, N, out.data_ptr(),
                                     idx.data_ptr() if want_index else None, count.data_ptr(), _stream()))
    m = int(count.item())
    return out[:m], (idx[:m] if want_index else None)


def correspond(canon_v, canon_n, live_v, want_index=True):
    """DynFusion::findCorrespondingFrame: (vertices, normals, index) of the nearest canonical vertex
    of every live vertex."""
    torch = _torch()
    n_live = live_v.shape[0]
    out_v = torch.empty((n_live, 3), dtype=torch.float32, device=live_v.device)
    out_n = torch.empty_like(out_v) if canon_n is not None else None
    idx = torch.empty((n_live,), dtype=torch.int32, device=live_v.device) if want_index else None
    _check(load().dfa_correspond(_dev(canon_v, torch.float32, "canon_v"), _dev(canon_n, torch.float32, "canon_n"),
                                 canon_v.shape[0], _dev(live_v, torch.float32, "live_v"), n_live, _dev(out_v),
                                 _dev(out_n), _dev(idx), _stream()))
    return out_v, out_n, idx


# ------------------------------------------------------------------------------ solver seam
class Solver:
    """dfa_solver plan.  Mirrors CombinedSolver's life cycle: set_problem == initializeProblemInstance,
    solve == solveAll; results stay on the device."""

    def __init__(self, max_D, max_N, k):
        _torch()
        self._h = C.c_void_p()
        self._L = load()  # the library that creates the plan serves it for life (use_library may swap the module's)
        self.k, self.max_D, self.max_N = k, max_D, max_N
        _check(self._L.dfa_solver_create(max_D, max_N, k, C.byref(self._h)))
        self._keep = None
        self.D = self.N = 0

    def close(self):
        if getattr(self, "_h", None) and self._h.value and getattr(self, "_L", None) is not None:
            self._L.dfa_solver_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close

    def set_problem(self, node_pos, node_dq, node_w, canon, live, canon_normals=None, live_normals=None):
        torch = _torch()
        f32 = torch.float32
        self.D, self.N = node_pos.shape[0], canon.shape[0]
        self._keep = (node_pos, node_dq, node_w, canon, live, canon_normals, live_normals)  # borrowed by the plan
        _check(self._L.dfa_solver_set_problem(self._h, _dev(node_pos, f32, "node_pos"), _dev(node_dq, f32, "node_dq"),
                                             _dev(node_w, f32, "node_w"), self.D, _dev(canon, f32, "canon"),
                                             _dev(canon_normals, f32), _dev(live, f32, "live"),
                                             _dev(live_normals, f32), self.N, _stream()))

    def set_deterministic(self, on=True):
        """order-stable variant (dfa_solver_set_deterministic); applies from the next set_problem"""
        _check(self._L.dfa_solver_set_deterministic(self._h, 1 if on else 0))

    def solve(self, params):
        _check(self._L.dfa_solver_solve(self._h, C.byref(params), _stream()))
        err, self._overlap_error = getattr(self, "_overlap_error", None), None
        if err is not None:
            raise err

    def _view(self, name, shape, dtype):
        """Copy of a plan-owned device array as a torch tensor (zero-copy view, then clone)."""
        torch = _torch()
        n = 1
        for d in shape:
            n *= d
        if n == 0:
            return torch.empty(shape, dtype=dtype, device="cuda")
        ptr = getattr(self._L, "dfa_solver_" + name)(self._h)
        typestr = {torch.float32: "<f4", torch.int32: "<i4"}[dtype]

        class _Holder:
            __cuda_array_interface__ = dict(shape=tuple(shape), typestr=typestr, data=(int(ptr), False), version=2)

        return torch.as_tensor(_Holder(), device="cuda").clone()

    def matrix(self):
        """(entries (256, D, 2) float32 — value, column bits —, row lengths (D,), gradient (D, 3)) of the last iteration"""
        torch = _torch()
        return (self._view("matrix_entries", (256, self.D, 2), torch.float32), self._view("matrix_row_lengths", (self.D,), torch.int32),
                self._view("gradient", (self.D, 3), torch.float32))

    def translations(self):
        return self._view("translations", (self.D, 3), _torch().float32)

    def node_dq(self):
        return self._view("node_dq", (self.D, 8), _torch().float32)

    def tukey_weights(self):
        return self._view("tukey_weights", (self.N,), _torch().float32)

    def huber_weights(self):
        return self._view("huber_weights", (self.D,), _torch().float32)

    def data_graph(self):
        return self._view("data_graph", (self.N, self.k), _torch().int32)

    def reg_graph(self):
        return self._view("reg_graph", (self.D, self.k), _torch().int32)

    def warp_to_live(self, normals=None):
        """canonical vertices of the problem warped by the solved transforms (plan graph, no second search)"""
        torch = _torch()
        out_v = torch.empty((self.N, 3), dtype=torch.float32, device="cuda")
        out_n = torch.empty_like(out_v) if normals is not None else None
        _check(self._L.dfa_solver_warp_to_live(self._h, _dev(normals, torch.float32, "normals"), _dev(out_v), _dev(out_n),
                                              _stream()))
        return out_v, out_n

    def enable_timing(self, on=True):
        """True / 1 starts a new measurement, 2 resumes a paused one, False / 0 pauses"""
        _check(self._L.dfa_solver_enable_timing(self._h, int(on)))

    def set_overlap_callback(self, fn):
        """fn(gn_iteration) (or None) is called by every following solve() right after the assembly launch of each
        Gauss-Newton iteration has been enqueued, before that iteration's PCG launch (dfa_solver_set_overlap_callback;
        -1 when the solve runs no iteration): the place to enqueue independent chip-wide work on another stream, behind
        an event recorded on the current one."""
        self._overlap_error = None
        if fn is None:
            self._overlap_cb = _OVERLAP_FN()
        else:
            def trampoline(_user, _stream, gn_iteration):
                try:
                    fn(gn_iteration)
                except BaseException as e:  # an exception cannot cross the C frame: re-raised by solve()
                    self._overlap_error = e
            self._overlap_cb = _OVERLAP_FN(trampoline)  # kept alive with the plan
        _check(self._L.dfa_solver_set_overlap_callback(self._h, self._overlap_cb, None))

    def timing(self):
        t = _SolveTiming()
        _check(self._L.dfa_solver_get_timing(self._h, C.byref(t), _stream()))
        return dict(pcg_ms=t.pcg_ms, assemble_ms=t.assemble_ms, pcg_launches=t.pcg_launches,
                    assemble_launches=t.assemble_launches, matrix_nnz=t.matrix_nnz, pcg_iters=t.pcg_iters, solves=t.solves)

    def team_pcg_info(self):
        """dfa_solver_team_pcg_info: {launches, aborts, disabled} of the team PCG (plans of 2 049 .. ~9 300 nodes)"""
        a, b, c = C.c_int(), C.c_int(), C.c_int()
        _check(self._L.dfa_solver_team_pcg_info(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return dict(launches=a.value, aborts=b.value, disabled=bool(c.value))

    def stats(self):
        st = _SolveStats()
        _check(self._L.dfa_solver_get_stats(self._h, C.byref(st), _stream()))
        return dict(initial_cost=st.initial_cost, final_cost=st.final_cost, gn_iters=st.gn_iters,
                    pcg_iters=st.pcg_iters, max_row_nnz=st.max_row_nnz, gn_noop=st.gn_noop)


# ------------------------------------------------------------------------ north-star solver seam
def compute_points_normals(depth, fx, fy, cx, cy):
    """kfusion::cuda::computePointNormals: (points, normals) float32 (H, W, 4) CUDA tensors"""
    torch = _torch()
    rows, cols = depth.shape
    pts = torch.empty((rows, cols, 4), dtype=torch.float32, device=depth.device)
    nrm = torch.empty_like(pts)
    _check(load().dfa_compute_points_normals(_dev(depth, torch.uint16, "depth"), depth.stride(0) * 2, cols, rows, fx, fy,
                                             cx, cy, _dev(pts), cols * 16, _dev(nrm), cols * 16, _stream()))
    return pts, nrm


class Solver6:
    """dfa_solver6 plan: the north-star 6-DoF solve."""

    def __init__(self, max_D, max_N, k):
        _torch()
        self._h = C.c_void_p()
        self._L = load()  # the library that creates the plan serves it for life (use_library may swap the module's)
        self.k, self.max_D, self.max_N = k, max_D, max_N
        _check(self._L.dfa_solver6_create(max_D, max_N, k, C.byref(self._h)))
        self._keep = None
        self.D = self.N = 0

    def close(self):
        if getattr(self, "_h", None) and self._h.value and getattr(self, "_L", None) is not None:
            self._L.dfa_solver6_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close

    def set_problem(self, node_pos, node_dq, node_w, canon, canon_normals=None):
        f32 = _torch().float32
        self.D, self.N = node_pos.shape[0], canon.shape[0]
        self._keep = (node_pos, node_dq, node_w, canon, canon_normals)  # borrowed by the plan
        _check(self._L.dfa_solver6_set_problem(self._h, _dev(node_pos, f32, "node_pos"), _dev(node_dq, f32, "node_dq"),
                                              _dev(node_w, f32, "node_w"), self.D, _dev(canon, f32, "canon"),
                                              _dev(canon_normals, f32, "canon_normals"), self.N, _stream()))

    def set_node_transforms(self, node_dq):
        """the same graphs, new starting transforms (dfa_solver6_set_node_transforms)"""
        self._keep = self._keep[:1] + (node_dq,) + self._keep[2:]
        _check(self._L.dfa_solver6_set_node_transforms(self._h, _dev(node_dq, _torch().float32, "node_dq")))

    def solve(self, vmap, nmap, fx, fy, cx, cy, params):
        f32 = _torch().float32
        rows, cols = vmap.shape[:2]
        _check(self._L.dfa_solver6_solve(self._h, _dev(vmap, f32, "vmap"), vmap.stride(0) * 4, _dev(nmap, f32, "nmap"),
                                        nmap.stride(0) * 4, cols, rows, fx, fy, cx, cy, C.byref(params), _stream()))

    def node_dq(self):
        torch = _torch()
        ptr = self._L.dfa_solver6_node_dq(self._h)

        class _Holder:
            __cuda_array_interface__ = dict(shape=(self.D, 8), typestr="<f4", data=(int(ptr), False), version=2)

        return torch.as_tensor(_Holder(), device="cuda").clone()

    def warp(self, want_normals=True):
        torch = _torch()
        out_v = torch.empty((self.N, 3), dtype=torch.float32, device="cuda")
        out_n = torch.empty_like(out_v) if want_normals and self._keep[4] is not None else None
        _check(self._L.dfa_solver6_warp(self._h, _dev(out_v), _dev(out_n), _stream()))
        return out_v, out_n

    def enable_timing(self, on=True):
        _check(self._L.dfa_solver6_enable_timing(self._h, 1 if on else 0))

    def timing(self):
        t = _Solve6Timing()
        _check(self._L.dfa_solver6_get_timing(self._h, C.byref(t), _stream()))
        return {n: getattr(t, n) for n, _ in _Solve6Timing._fields_}

    def stats(self):
        st = _Solve6Stats()
        _check(self._L.dfa_solver6_get_stats(self._h, C.byref(st), _stream()))
        d = {n: getattr(st, n) for n, _ in _Solve6Stats._fields_}
        n = st.hist_n  # slots of the Gauss-Newton loop (gn_tol > 0: + the closing check), skipped ones included
        for name in ("cost_hist", "pcg_rel_hist", "pcg_it_hist", "pcg_tol_hist", "valid_hist", "stop_hist"):
            d[name] = list(d[name])[:n]
        return d


# ------------------------------------------------------------------- depth pre-processing seam
def depth_bilateral_filter(depth, kernel_size, sigma_spatial, sigma_depth):
    torch = _torch()
    rows, cols = depth.shape
    out = torch.empty_like(depth)
    _check(load().dfa_depth_bilateral_filter(_dev(depth, torch.uint16, "depth"), depth.stride(0) * 2, _dev(out),
                                             out.stride(0) * 2, cols, rows, kernel_size, sigma_spatial, sigma_depth,
                                             _stream()))
    return out


def depth_truncate(depth, max_dist):
    torch = _torch()
    rows, cols = depth.shape
    _check(load().dfa_depth_truncate(_dev(depth, torch.uint16, "depth"), depth.stride(0) * 2, cols, rows, max_dist, _stream()))


def depth_build_pyramid(depth, sigma_depth):
    torch = _torch()
    rows, cols = depth.shape
    out = torch.zeros((rows // 2, cols // 2), dtype=torch.uint16, device=depth.device)
    _check(load().dfa_depth_build_pyramid(_dev(depth, torch.uint16, "depth"), depth.stride(0) * 2, cols, rows, _dev(out),
                                          max(out.stride(0), 1) * 2, sigma_depth, _stream()))
    return out


def compute_normals_mask_depth(depth, fx, fy, cx, cy):
    """in place on depth; returns the float4 normal map"""
    torch = _torch()
    rows, cols = depth.shape
    nrm = torch.empty((rows, cols, 4), dtype=torch.float32, device=depth.device)
    _check(load().dfa_compute_normals_mask_depth(_dev(depth, torch.uint16, "depth"), depth.stride(0) * 2, cols, rows, fx,
                                                 fy, cx, cy, _dev(nrm), cols * 16, _stream()))
    return nrm


def resize_depth_normals(depth, normals):
    torch = _torch()
    rows, cols = depth.shape
    d = torch.zeros((rows // 2, cols // 2), dtype=torch.uint16, device=depth.device)
    n = torch.zeros((rows // 2, cols // 2, 4), dtype=torch.float32, device=depth.device)
    _check(load().dfa_resize_depth_normals(_dev(depth, torch.uint16, "depth"), depth.stride(0) * 2,
                                           _dev(normals, torch.float32, "normals"), normals.stride(0) * 4, cols, rows,
                                           _dev(d), max(d.stride(0), 1) * 2, _dev(n), max(n.stride(0), 4) * 4, _stream()))
    return d, n


def resize_points_normals(points, normals):
    torch = _torch()
    rows, cols = points.shape[:2]
    v = torch.zeros((rows // 2, cols // 2, 4), dtype=torch.float32, device=points.device)
    n = torch.zeros_like(v)
    _check(load().dfa_resize_points_normals(_dev(points, torch.float32, "points"), points.stride(0) * 4,
                                            _dev(normals, torch.float32, "normals"), normals.stride(0) * 4, cols, rows,
                                            _dev(v), max(v.stride(0), 4) * 4, _dev(n), max(n.stride(0), 4) * 4, _stream()))
    return v, n


# --------------------------------------------------------------------------------- rigid-ICP seam
def icp_sums(curr, ncurr, prev, nprev, aff12, fx, fy, cx, cy, dist_thres=0.1, angle_thres=0.3490658503988659):
    """one linearisation of the rigid projective ICP -> (27 sums CUDA float tensor, matched pixels CUDA int tensor)"""
    torch = _torch()
    depth_variant = curr.dtype == torch.uint16
    rows, cols = curr.shape[:2]
    esz = 2 if depth_variant else 4
    sums = torch.zeros(27, dtype=torch.float32, device=curr.device)
    matched = torch.zeros(1, dtype=torch.int32, device=curr.device)
    _check(load().dfa_icp_sums(1 if depth_variant else 0, _dev(curr), curr.stride(0) * esz, _dev(ncurr, torch.float32, "ncurr"),
                               ncurr.stride(0) * 4, _dev(prev), prev.stride(0) * esz, _dev(nprev, torch.float32, "nprev"),
                               nprev.stride(0) * 4, cols, rows, _aff12(aff12), fx, fy, cx, cy, dist_thres, angle_thres,
                               _dev(sums), _dev(matched), _stream()))
    return sums, matched
