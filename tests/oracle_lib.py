"""ctypes access to the CPU checkers under oracle/ (TEST INFRASTRUCTURE).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
Two libraries export the same C interface (oracle/oracle_api.h):

* ``oracle/liboracle.so``           -- the restatement (oracle/pwpp_oracle.cpp)
* ``oracle/_ref/libpwpp_ref*.so``   -- the reference's own patchworkpp.cpp compiled unmodified
                                       against oracle/eigen_shim (built only where
                                       /root/reference exists; the .so travels to the GPU box)
"""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")

ARITH_EIGEN_F32 = 0   # float sums in storage order (plainest reading of Eigen)
ARITH_FXP = 1         # the product's fixed-point contract (restatement only)
ARITH_EXACT_F64 = 2   # reference-neutral arbiter: double sums of the unquantised floats
ARITH_F32_PACKET4 = 3  # float sums, four partial sums (a SIMD reduction order)
ARITH_FXP21 = 4       # contract v3 of rounds 3-5 (2^-21 m grid), restatement only: a witness
REF_LIBS = {ARITH_EIGEN_F32: "libpwpp_ref.so", ARITH_EXACT_F64: "libpwpp_ref_exact.so",
            ARITH_F32_PACKET4: "libpwpp_ref_pk4.so"}

DEC_NAMES = {1: "not_upright", 2: "far_ground", 3: "heading", 4: "ground", 5: "tgr_reject", 6: "tgr_revert"}


class Params(ctypes.Structure):
    """Mirror of pwo_params (oracle/oracle_api.h) = patchwork::Params (reference patchworkpp.h:42-112)."""

    _fields_ = (
        [(n, ctypes.c_int32) for n in
         "verbose enable_RNR enable_RVPF enable_TGR num_iter num_lpr num_min_pts num_zones "
         "num_rings_of_interest".split()]
        + [(n, ctypes.c_double) for n in
           "RNR_ver_angle_thr RNR_intensity_thr sensor_height th_seeds th_dist th_seeds_v th_dist_v "
           "max_range min_range uprightness_thr adaptive_seed_selection_margin".split()]
        + [("num_sectors_each_zone", ctypes.c_int32 * 4), ("num_rings_each_zone", ctypes.c_int32 * 4),
           ("max_flatness_storage", ctypes.c_int32), ("max_elevation_storage", ctypes.c_int32),
           ("elevation_thr", ctypes.c_double * 4), ("flatness_thr", ctypes.c_double * 4)]
    )


class PatchRecord(ctypes.Structure):
    _fields_ = [("bin", ctypes.c_int32), ("concentric_idx", ctypes.c_int32), ("n_points", ctypes.c_int32),
                ("n_ground", ctypes.c_int32), ("n_nonground", ctypes.c_int32), ("decision", ctypes.c_int32),
                ("mean", ctypes.c_float * 3), ("normal", ctypes.c_float * 3), ("sv", ctypes.c_float * 3),
                ("pad_", ctypes.c_float), ("d", ctypes.c_double)]


RECORD_DTYPE = np.dtype([("bin", "<i4"), ("concentric_idx", "<i4"), ("n_points", "<i4"), ("n_ground", "<i4"),
                         ("n_nonground", "<i4"), ("decision", "<i4"), ("mean", "<f4", 3), ("normal", "<f4", 3),
                         ("sv", "<f4", 3), ("pad_", "<f4"), ("d", "<f8")])
assert RECORD_DTYPE.itemsize == ctypes.sizeof(PatchRecord)


def build(quiet=True):
    """(Re)build liboracle.so and, where /root/reference exists, oracle/_ref/*.so."""
    subprocess.run(["make", "-C", ORACLE_DIR, "all"], check=True,
                   stdout=subprocess.DEVNULL if quiet else None)


def _vp(a):
    return a.ctypes.data_as(ctypes.c_void_p)


class _Lib:
    def __init__(self, path, has_ext):
        self.path = path
        self.lib = L = ctypes.CDLL(path)
        self.has_ext = has_ext
        L.pwo_create.restype = ctypes.c_void_p
        L.pwo_create.argtypes = [ctypes.POINTER(Params), ctypes.c_int]
        L.pwo_destroy.argtypes = [ctypes.c_void_p]
        L.pwo_estimate_ground.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
        for name in ("pwo_num_ground", "pwo_num_nonground", "pwo_num_patches"):
            getattr(L, name).argtypes = [ctypes.c_void_p]
        for name in ("pwo_get_ground_indices", "pwo_get_nonground_indices", "pwo_get_ground", "pwo_get_nonground",
                     "pwo_get_centers", "pwo_get_normals"):
            getattr(L, name).argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        L.pwo_get_height.restype = ctypes.c_double
        L.pwo_get_height.argtypes = [ctypes.c_void_p]
        L.pwo_get_time_taken.restype = ctypes.c_double
        L.pwo_get_time_taken.argtypes = [ctypes.c_void_p]
        L.pwo_get_thresholds.argtypes = [ctypes.c_void_p] + [ctypes.c_void_p] * 3
        L.pwo_get_history_len.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
        L.pwo_get_history.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        L.pwo_bench.restype = ctypes.c_double
        L.pwo_bench.argtypes = [ctypes.POINTER(Params), ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        if has_ext:
            L.pwo_ext_num_records.argtypes = [ctypes.c_void_p]
            L.pwo_ext_get_records.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
            L.pwo_ext_set_state.argtypes = [ctypes.c_void_p, ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p]
            L.pwo_ext_jacobi.argtypes = [ctypes.c_void_p] * 3
            L.pwo_ext_fxp_geometry.argtypes = [ctypes.c_void_p] * 5
            L.pwo_ext_quantise.restype = ctypes.c_longlong
            L.pwo_ext_quantise.argtypes = [ctypes.c_float, ctypes.c_double, ctypes.c_int]
            L.pwo_ext_quantise_z.restype = ctypes.c_longlong
            L.pwo_ext_quantise_z.argtypes = [ctypes.c_float, ctypes.c_double, ctypes.c_int, ctypes.c_double]
            L.pwo_ext_z_origin.restype = ctypes.c_double
            L.pwo_ext_z_origin.argtypes = [ctypes.c_double]

    def default_params(self):
        p = Params()
        self.lib.pwo_default_params(ctypes.byref(p))
        return p

    def supports(self, arith):
        return bool(self.lib.pwo_arith_supported(arith))


_cache = {}


def restatement():
    """oracle/liboracle.so (built on demand)."""
    if "o" not in _cache:
        path = os.path.join(ORACLE_DIR, "liboracle.so")
        if not os.path.exists(path):
            build()
        _cache["o"] = _Lib(path, True)
    return _cache["o"]


def reference(arith=ARITH_EIGEN_F32):
    """oracle/_ref/libpwpp_ref*.so of that flavour, or None when it does not exist (not built, or the
    flavour is the fixed-point contract, which only the restatement implements)."""
    key = ("r", arith)
    if key not in _cache:
        name = REF_LIBS.get(arith)
        path = os.path.join(ORACLE_DIR, "_ref", name) if name else None
        _cache[key] = _Lib(path, False) if path and os.path.exists(path) else None
    return _cache[key]


class Result:
    """Everything one estimateGround() call produced."""

    def __init__(self):
        self.ground_idx = self.nonground_idx = None
        self.ground = self.nonground = self.centers = self.normals = None
        self.height = self.time_taken = None
        self.sensor_height = None
        self.elevation_thr = self.flatness_thr = None
        self.hist_elev = self.hist_flat = None
        self.records = None


class Estimator:
    """One stateful object (reference: one PatchWorkpp instance) behind either library."""

    def __init__(self, lib, params=None, arith=ARITH_FXP):
        self._l = lib
        self.params = params if params is not None else lib.default_params()
        self._h = ctypes.c_void_p(lib.lib.pwo_create(ctypes.byref(self.params), arith))
        if not self._h:
            raise RuntimeError("%s does not support arith=%d" % (lib.path, arith))

    def close(self):
        if self._h:
            self._l.lib.pwo_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def fxp_geometry(self):
        """(shift, z half-range in metres, per-bin origin x, per-bin origin y) of the fixed-point contract."""
        nb = sum(self.params.num_rings_each_zone[k] * self.params.num_sectors_each_zone[k] for k in range(4))
        sh, zr = ctypes.c_int(), ctypes.c_double()
        ox, oy = np.zeros(nb, np.float32), np.zeros(nb, np.float32)
        self._l.lib.pwo_ext_fxp_geometry(self._h, ctypes.byref(sh), ctypes.byref(zr), _vp(ox), _vp(oy))
        return sh.value, zr.value, ox, oy

    def set_state(self, sensor_height, elevation_thr, flatness_thr):
        e = np.ascontiguousarray(elevation_thr, np.float64)
        f = np.ascontiguousarray(flatness_thr, np.float64)
        self._l.lib.pwo_ext_set_state(self._h, float(sensor_height), _vp(e), _vp(f))

    def run(self, pts):
        pts = np.ascontiguousarray(pts, dtype=np.float32)
        assert pts.ndim == 2 and pts.shape[1] in (3, 4)
        L, h = self._l.lib, self._h
        L.pwo_estimate_ground(h, _vp(pts), pts.shape[0], pts.shape[1])
        r = Result()
        ng, nn, npat = L.pwo_num_ground(h), L.pwo_num_nonground(h), L.pwo_num_patches(h)
        r.ground_idx = np.zeros(ng, np.int32)
        r.nonground_idx = np.zeros(nn, np.int32)
        r.ground = np.zeros((ng, 3), np.float32)
        r.nonground = np.zeros((nn, 3), np.float32)
        r.centers = np.zeros((npat, 3), np.float32)
        r.normals = np.zeros((npat, 3), np.float32)
        L.pwo_get_ground_indices(h, _vp(r.ground_idx))
        L.pwo_get_nonground_indices(h, _vp(r.nonground_idx))
        L.pwo_get_ground(h, _vp(r.ground))
        L.pwo_get_nonground(h, _vp(r.nonground))
        L.pwo_get_centers(h, _vp(r.centers))
        L.pwo_get_normals(h, _vp(r.normals))
        r.height = L.pwo_get_height(h)
        r.time_taken = L.pwo_get_time_taken(h)
        sh = np.zeros(1)
        r.elevation_thr = np.zeros(4)
        r.flatness_thr = np.zeros(4)
        L.pwo_get_thresholds(h, _vp(sh), _vp(r.elevation_thr), _vp(r.flatness_thr))
        r.sensor_height = float(sh[0])
        r.hist_elev, r.hist_flat = [], []
        for which, dst in ((0, r.hist_elev), (1, r.hist_flat)):
            for ring in range(4):
                k = L.pwo_get_history_len(h, which, ring)
                a = np.zeros(k)
                if k:
                    L.pwo_get_history(h, which, ring, _vp(a))
                dst.append(a)
        if self._l.has_ext:
            k = L.pwo_ext_num_records(h)
            r.records = np.zeros(k, RECORD_DTYPE)
            if k:
                L.pwo_ext_get_records(h, _vp(r.records))
        return r


def cpu_bench(lib, frames, total, threads, params=None, arith=ARITH_EIGEN_F32):
    """Frame-parallel CPU timing: returns (wall_seconds, sum_of_call_seconds)."""
    params = params if params is not None else lib.default_params()
    frames = [np.ascontiguousarray(f, np.float32) for f in frames]
    cols = frames[0].shape[1]
    ptrs = (ctypes.c_void_p * len(frames))(*[f.ctypes.data for f in frames])
    ns = (ctypes.c_int * len(frames))(*[f.shape[0] for f in frames])
    call = ctypes.c_double(0.0)
    wall = lib.lib.pwo_bench(ctypes.byref(params), arith, ptrs, ns, cols, len(frames), total, threads,
                             ctypes.byref(call))
    if wall < 0:
        raise RuntimeError("pwo_bench refused (arith=%d threads=%d)" % (arith, threads))
    return wall, call.value


def jacobi(cov):
    """The oracle's float 3x3 Jacobi SVD: returns (U row-major 3x3, sv)."""
    cov = np.ascontiguousarray(cov, np.float32).reshape(9)
    u = np.zeros(9, np.float32)
    sv = np.zeros(3, np.float32)
    restatement().lib.pwo_ext_jacobi(_vp(cov), _vp(u), _vp(sv))
    return u.reshape(3, 3), sv


if __name__ == "__main__":
    # Worker of bench.py's cpu_baseline: one single-threaded process of the frame-parallel CPU harness.
    #   oracle_lib.py --bench-worker FRAMES.npz COUNT T_GO KIND ARITH
    # loads the library and the frames, waits for the wall-clock mark T_GO, runs COUNT frames (fresh object each) on ONE thread
    # and prints "t_begin t_end frames".
    import sys
    import time
    if len(sys.argv) >= 7 and sys.argv[1] == "--bench-worker":
        _z = np.load(sys.argv[2])
        _frames = [_z[k] for k in _z.files]
        _count, _t_go, _kind, _arith = int(sys.argv[3]), float(sys.argv[4]), sys.argv[5], int(sys.argv[6])
        _lib = reference(_arith) if _kind == "reference" else restatement()
        cpu_bench(_lib, _frames, len(_frames), 1, arith=_arith)  # (pages, caches)
        while time.time() < _t_go:
            time.sleep(0.002)
        _b = time.time()
        cpu_bench(_lib, _frames, _count, 1, arith=_arith)
        print(_b, time.time(), _count)
