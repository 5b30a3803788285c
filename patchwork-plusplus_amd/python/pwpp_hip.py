"""ctypes binding of the C-ABI in include/pwpp.h (libpwpp_hip.so).

Thin host-side plumbing for tests, bench.py and Python callers that want the batch API;
the drop-in Python surface of the reference (module ``pypatchworkpp``) is the pybind11
module built from pybinding.cpp next to this file.  No CPU fallback: if the shared
library is missing or there is no GPU, constructing a :class:`Handle` raises.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "lib", "libpwpp_hip.so")

LAYOUT_ROW_MAJOR, LAYOUT_COL_MAJOR = 0, 1
MEM_HOST, MEM_DEVICE = 0, 1
MODE_FRESH, MODE_STREAMS = 0, 1
MEM_HOST_PINNED = 2
NUM_KERNELS = 11

DEC_NAMES = {1: "not_upright", 2: "far_ground", 3: "heading", 4: "ground", 5: "tgr_reject", 6: "tgr_revert"}


class Params(ctypes.Structure):
    """pwpp_params = patchwork::Params (reference patchworkpp.h:42-112)."""

    _fields_ = (
        [(n, ctypes.c_int32) for n in
         "verbose enable_RNR enable_RVPF enable_TGR num_iter num_lpr num_min_pts num_zones "
         "num_rings_of_interest pad0_".split()]
        + [(n, ctypes.c_double) for n in
           "RNR_ver_angle_thr RNR_intensity_thr sensor_height th_seeds th_dist th_seeds_v th_dist_v "
           "max_range min_range uprightness_thr adaptive_seed_selection_margin intensity_thr".split()]
        + [("num_sectors_each_zone", ctypes.c_int32 * 4), ("num_rings_each_zone", ctypes.c_int32 * 4),
           ("max_flatness_storage", ctypes.c_int32), ("max_elevation_storage", ctypes.c_int32),
           ("elevation_thr", ctypes.c_double * 4), ("flatness_thr", ctypes.c_double * 4)]
    )


class State(ctypes.Structure):
    _fields_ = [("sensor_height", ctypes.c_double), ("elevation_thr", ctypes.c_double * 4),
                ("flatness_thr", ctypes.c_double * 4), ("elevation_len", ctypes.c_int32 * 4),
                ("flatness_len", ctypes.c_int32 * 4)]


RECORD_DTYPE = np.dtype([("bin", "<i4"), ("concentric_idx", "<i4"), ("n_points", "<i4"), ("n_ground", "<i4"),
                         ("n_nonground", "<i4"), ("decision", "<i4"), ("mean", "<f4", 3), ("normal", "<f4", 3),
                         ("sv", "<f4", 3), ("rounds", "<i4"), ("d", "<f8")])


class DeviceView(ctypes.Structure):
    _fields_ = [("indices", ctypes.c_void_p), ("frame_base", ctypes.POINTER(ctypes.c_int64)),
                ("counts", ctypes.POINTER(ctypes.c_int32)), ("frames", ctypes.c_int32), ("pad_", ctypes.c_int32)]


class PwppError(RuntimeError):
    pass


_lib = None


def load():
    """Load libpwpp_hip.so (raises if it has not been built -- there is no fallback)."""
    global _lib
    if _lib is None:
        path = os.environ.get("PWPP_LIB_PATH", LIB_PATH)  # A/B runs of two builds (tools/ab_bench.sh)
        if not os.path.exists(path):
            raise PwppError("%s not built; run __graft_entry__.build() or make -C patchwork-plusplus_amd" % path)
        L = ctypes.CDLL(path)
        L.pwpp_last_error.restype = ctypes.c_char_p
        L.pwpp_kernel_name.restype = ctypes.c_char_p
        L.pwpp_get_height.restype = ctypes.c_double
        L.pwpp_get_time_us.restype = ctypes.c_double
        vp, ci = ctypes.c_void_p, ctypes.c_int
        L.pwpp_create.argtypes = [ctypes.POINTER(Params), ci, ctypes.POINTER(vp)]
        L.pwpp_destroy.argtypes = [vp]
        L.pwpp_estimate_ground.argtypes = [vp, vp, ci, ci, ci]
        L.pwpp_estimate_ground_batch.argtypes = [vp, vp, vp, ci, ci, ci, ci, ci]
        L.pwpp_estimate_ground_fields.argtypes = [vp, vp, ci, ci, ci, ci, ci, ci]
        L.pwpp_synchronize.argtypes = [vp]
        L.pwpp_set_num_streams.argtypes = [vp, ci]
        L.pwpp_get_counts.argtypes = [vp, ci, vp, vp, vp]
        for name in ("pwpp_get_ground_indices", "pwpp_get_nonground_indices", "pwpp_get_ground_xyz",
                     "pwpp_get_nonground_xyz", "pwpp_get_centers", "pwpp_get_normals"):
            getattr(L, name).argtypes = [vp, ci, vp]
        L.pwpp_get_patch_records.argtypes = [vp, ci, vp, ci]
        L.pwpp_get_height.argtypes = [vp]
        L.pwpp_get_time_us.argtypes = [vp]
        L.pwpp_get_state.argtypes = [vp, ci, ctypes.POINTER(State)]
        L.pwpp_get_history.argtypes = [vp, ci, ci, ci, vp, ci]
        L.pwpp_set_state.argtypes = [vp, ci, ctypes.POINTER(State)]
        L.pwpp_set_history.argtypes = [vp, ci, ci, ci, vp, ci]
        L.pwpp_get_bin_boxes.argtypes = [vp, vp, ci]
        L.pwpp_get_fxp_geometry.argtypes = [vp, vp, vp, ci]
        L.pwpp_get_fixed_up_frames.argtypes = [vp]
        L.pwpp_get_fixed_up_frames.restype = ctypes.c_int64
        L.pwpp_get_clamped_frames.argtypes = [vp]
        L.pwpp_get_clamped_frames.restype = ctypes.c_int64
        L.pwpp_get_plane_state.argtypes = [vp, ci, vp]
        L.pwpp_set_plane_state.argtypes = [vp, ci, vp]
        L.pwpp_get_device_view.argtypes = [vp, ctypes.POINTER(DeviceView)]
        L.pwpp_set_profiling.argtypes = [vp, ci]
        L.pwpp_host_alloc.argtypes = [ctypes.POINTER(vp), ctypes.c_uint64]
        L.pwpp_host_free.argtypes = [vp]
        L.pwpp_get_all_indices.argtypes = [vp, vp, vp, vp]
        L.pwpp_get_kernel_profile.argtypes = [vp, vp, vp]
        L.pwpp_reset_kernel_profile.argtypes = [vp]
        L.pwpp_get_fxp_shift.argtypes = [vp]
        L.pwpp_get_fxp_origins.argtypes = [vp, vp, ci]
        L.pwpp_set_option.argtypes = [vp, ctypes.c_char_p, ctypes.c_char_p]
        L.pwpp_trim_workspace.argtypes = [vp]
        L.pwpp_get_workspace_bytes.argtypes = [vp]
        L.pwpp_get_workspace_bytes.restype = ctypes.c_int64
        L.pwpp_get_one_pass_stats.argtypes = [vp, vp, vp]
        L.pwpp_get_redo_stats.argtypes = [vp, vp, vp]
        L.pwpp_pipe_create.argtypes = [vp, ci, ci, ctypes.POINTER(vp)]
        L.pwpp_pipe_submit.argtypes = [vp, vp, vp, ci, ci, ci, ci, ci, ctypes.POINTER(vp)]
        L.pwpp_pipe_set_num_streams.argtypes = [vp, ci]
        L.pwpp_get_arena_stats.argtypes = [vp, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64)]
        L.pwpp_pipe_drain.argtypes = [vp]
        L.pwpp_pipe_handle.argtypes = [vp, ci]
        L.pwpp_pipe_handle.restype = vp
        L.pwpp_pipe_destroy.argtypes = [vp]
        L.pwpp_set_output_order.argtypes = [vp, ci]
        L.pwpp_set_overlap.argtypes = [vp, ci]
        L.pwpp_kernel_name.argtypes = [ci]
        _lib = L
    return _lib


def pinned_empty(shape, dtype=np.float32):
    """A page-locked numpy array (pwpp_host_alloc); keep it alive while the GPU may touch it."""
    L = load()
    n = int(np.prod(shape)) * np.dtype(dtype).itemsize
    p = ctypes.c_void_p()
    if L.pwpp_host_alloc(ctypes.byref(p), n) < 0:
        raise PwppError(L.pwpp_last_error().decode())
    buf = (ctypes.c_char * max(n, 1)).from_address(p.value)
    arr = np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)
    _pinned_keep[arr.ctypes.data] = (p, buf)
    return arr


def pinned_free(arr):
    ent = _pinned_keep.pop(arr.ctypes.data, None)
    if ent:
        load().pwpp_host_free(ent[0])


_pinned_keep = {}


def default_params():
    p = Params()
    load().pwpp_params_default(ctypes.byref(p))
    return p


def _vp(a):
    return a.ctypes.data_as(ctypes.c_void_p)


class Handle:
    """One pwpp_handle: one device, one HIP stream, the workspace and the adaptive state."""

    def __init__(self, params=None, device=0):
        self._L = load()
        self.params = params if params is not None else default_params()
        h = ctypes.c_void_p()
        self._h = None
        self._check(self._L.pwpp_create(ctypes.byref(self.params), device, ctypes.byref(h)))
        self._h = h
        self._keep = None

    def _check(self, rc):
        if rc < 0:
            raise PwppError("pwpp error %d: %s" % (rc, self._L.pwpp_last_error().decode()))
        return rc

    def close(self):
        if self._h:
            if not getattr(self, "_borrowed", False):  # (a view of a Pipe's handle: the pipe destroys it)
                self._L.pwpp_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- the hot path -----------------------------------------------------------------
    def estimate_ground(self, pts):
        """Reference estimateGround(): one host frame, stateful (stream 0)."""
        pts = np.asarray(pts, dtype=np.float32)
        if pts.ndim != 2:
            raise PwppError("expected a 2-D array")
        if pts.flags["C_CONTIGUOUS"]:
            layout = LAYOUT_ROW_MAJOR
        elif pts.flags["F_CONTIGUOUS"]:
            layout = LAYOUT_COL_MAJOR
        else:
            pts = np.ascontiguousarray(pts)
            layout = LAYOUT_ROW_MAJOR
        self._check(self._L.pwpp_estimate_ground(self._h, _vp(pts), pts.shape[0], pts.shape[1], layout))

    def estimate_ground_batch(self, frames, mode=MODE_FRESH):
        """Host frames (list of (n,cols) float32 arrays, all C-contiguous or all Fortran-contiguous =
        Eigen::MatrixXf storage), synchronous."""
        col_major = all(f.dtype == np.float32 and f.flags["F_CONTIGUOUS"] and not f.flags["C_CONTIGUOUS"] for f in frames)
        if not col_major:
            frames = [np.ascontiguousarray(f, dtype=np.float32) for f in frames]
        cols = frames[0].shape[1]
        ptrs = (ctypes.c_void_p * len(frames))(*[f.ctypes.data for f in frames])
        ns = (ctypes.c_int32 * len(frames))(*[f.shape[0] for f in frames])
        self._check(self._L.pwpp_estimate_ground_batch(self._h, ptrs, ns, len(frames), cols,
                                                       LAYOUT_COL_MAJOR if col_major else LAYOUT_ROW_MAJOR, MEM_HOST, mode))

    def submit_pinned_batch(self, frames, mode=MODE_FRESH):
        """Frames in page-locked host memory (pinned_empty): copies and launches are only enqueued;
        the arrays must stay untouched until synchronize() / a getter."""
        cols = frames[0].shape[1]
        ptrs = (ctypes.c_void_p * len(frames))(*[f.ctypes.data for f in frames])
        ns = (ctypes.c_int32 * len(frames))(*[f.shape[0] for f in frames])
        self._keep = (ptrs, ns, frames)
        self._check(self._L.pwpp_estimate_ground_batch(self._h, ptrs, ns, len(frames), cols, LAYOUT_ROW_MAJOR,
                                                       MEM_HOST_PINNED, mode))

    def estimate_ground_batch_device(self, ptrs, ns, cols=4, layout=LAYOUT_ROW_MAJOR, mode=MODE_FRESH):
        """Device-resident frames: ptrs = device addresses (ints), asynchronous until synchronize()."""
        k = len(ptrs)
        cp = (ctypes.c_void_p * k)(*ptrs)
        cn = (ctypes.c_int32 * k)(*ns)
        self._keep = (cp, cn)
        self._check(self._L.pwpp_estimate_ground_batch(self._h, cp, cn, k, cols, layout, MEM_DEVICE, mode))

    def make_device_batch(self, ptrs, ns):
        """Pre-built ctypes argument arrays for repeated launches of the same batch."""
        k = len(ptrs)
        return (ctypes.c_void_p * k)(*ptrs), (ctypes.c_int32 * k)(*ns), k

    def launch_device_batch(self, batch, cols=4, layout=LAYOUT_ROW_MAJOR, mode=MODE_FRESH):
        cp, cn, k = batch
        self._check(self._L.pwpp_estimate_ground_batch(self._h, cp, cn, k, cols, layout, MEM_DEVICE, mode))

    def synchronize(self):
        self._check(self._L.pwpp_synchronize(self._h))

    def set_num_streams(self, n):
        self._check(self._L.pwpp_set_num_streams(self._h, n))

    # ---- results ----------------------------------------------------------------------
    def counts(self, frame=0):
        a, b, c = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
        self._check(self._L.pwpp_get_counts(self._h, frame, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)))
        return a.value, b.value, c.value

    def _get(self, fn, frame, shape, dtype):
        out = np.zeros(shape, dtype)
        self._check(fn(self._h, frame, _vp(out) if out.size else None))
        return out

    def ground_indices(self, frame=0):
        return self._get(self._L.pwpp_get_ground_indices, frame, self.counts(frame)[0], np.int32)

    def nonground_indices(self, frame=0):
        return self._get(self._L.pwpp_get_nonground_indices, frame, self.counts(frame)[1], np.int32)

    def ground(self, frame=0):
        return self._get(self._L.pwpp_get_ground_xyz, frame, (self.counts(frame)[0], 3), np.float32)

    def nonground(self, frame=0):
        return self._get(self._L.pwpp_get_nonground_xyz, frame, (self.counts(frame)[1], 3), np.float32)

    def centers(self, frame=0):
        return self._get(self._L.pwpp_get_centers, frame, (self.counts(frame)[2], 3), np.float32)

    def normals(self, frame=0):
        return self._get(self._L.pwpp_get_normals, frame, (self.counts(frame)[2], 3), np.float32)

    def patch_records(self, frame=0):
        n = self.counts(frame)[2]
        out = np.zeros(max(n, 1), RECORD_DTYPE)
        k = self._check(self._L.pwpp_get_patch_records(self._h, frame, _vp(out), len(out)))
        return out[:k]

    def height(self):
        return self._L.pwpp_get_height(self._h)

    def time_us(self):
        return self._L.pwpp_get_time_us(self._h)

    def state(self, index=0):
        s = State()
        self._check(self._L.pwpp_get_state(self._h, index, ctypes.byref(s)))
        return s

    def set_state(self, stream, sensor_height, elevation_thr, flatness_thr):
        s = State()
        s.sensor_height = sensor_height
        for k in range(4):
            s.elevation_thr[k] = elevation_thr[k]
            s.flatness_thr[k] = flatness_thr[k]
        self._check(self._L.pwpp_set_state(self._h, stream, ctypes.byref(s)))

    def estimate_ground_fields(self, data, n, point_step, off_x, off_y, off_z, off_intensity=-1):
        """One frame handed over as a sensor_msgs/PointCloud2 data blob (bytes-like / uint8 array)."""
        buf = np.frombuffer(data, np.uint8) if not isinstance(data, np.ndarray) else data
        self._keep = buf
        self._check(self._L.pwpp_estimate_ground_fields(self._h, _vp(buf), n, point_step, off_x, off_y, off_z, off_intensity))

    def set_history(self, stream, which, ring, values):
        v = np.ascontiguousarray(values, np.float64)
        self._check(self._L.pwpp_set_history(self._h, stream, which, ring, _vp(v) if v.size else None, int(v.size)))

    def checkpoint(self, stream=0):
        """Everything a stream carries from frame to frame (pwpp_get_state + the eight histories + the plane members)."""
        st = self.state(stream)
        return dict(sensor_height=st.sensor_height, elevation_thr=list(st.elevation_thr), flatness_thr=list(st.flatness_thr),
                    hist=[[self.history(stream, w, r) for r in range(4)] for w in range(2)], plane=self.plane_state(stream))

    def restore(self, ck, stream=0):
        self.set_state(stream, ck["sensor_height"], ck["elevation_thr"], ck["flatness_thr"])
        for w in range(2):
            for r in range(4):
                self.set_history(stream, w, r, ck["hist"][w][r])
        if "plane" in ck:
            self.set_plane_state(stream, ck["plane"])

    def clamped_frames(self):
        """Frames in which a patch's final ground set spanned more than z0 +- ZR vertically (pwpp_get_clamped_frames)."""
        return int(self._L.pwpp_get_clamped_frames(self._h))

    def fixed_up_frames(self):
        """Frames finished by the serial fix-up kernel so far (pwpp_get_fixed_up_frames)."""
        return int(self._L.pwpp_get_fixed_up_frames(self._h))

    def plane_state(self, index=0):
        """{mean[3], normal[3], singular values[3], d} of the plane the stream's last frame fitted last (pwpp_get_plane_state)."""
        out = np.zeros(10, np.float32)
        self._check(self._L.pwpp_get_plane_state(self._h, index, _vp(out)))
        return out

    def set_plane_state(self, stream, values):
        v = np.ascontiguousarray(values, np.float32)
        assert v.shape == (10,)
        self._check(self._L.pwpp_set_plane_state(self._h, stream, _vp(v)))

    def history(self, index, which, ring):
        n = self._check(self._L.pwpp_get_history(self._h, index, which, ring, None, 0))
        out = np.zeros(n, np.float64)
        if n:
            self._check(self._L.pwpp_get_history(self._h, index, which, ring, _vp(out), n))
        return out

    def device_view(self):
        v = DeviceView()
        self._check(self._L.pwpp_get_device_view(self._h, ctypes.byref(v)))
        return v

    def all_indices(self, out=None):
        """Every frame's ground + non-ground list in one device-to-host copy.

        Returns (indices, frame_base, counts): frame f's ground list is
        indices[frame_base[f] : frame_base[f] + counts[f, 0]], its non-ground list follows."""
        v = self.device_view()
        base = np.zeros(v.frames + 1, np.int64)
        counts = np.zeros((v.frames, 8), np.int32)
        total = int(np.ctypeslib.as_array(v.frame_base, shape=(v.frames + 1,))[-1])
        if out is None:
            out = np.empty(max(total, 1), np.int32)
        self._check(self._L.pwpp_get_all_indices(self._h, _vp(out), _vp(base), _vp(counts)))
        return out[:total], base, counts

    def all_counts(self):
        """(frames, 8) int32 array of per-frame counters of the last call."""
        v = self.device_view()
        return np.ctypeslib.as_array(v.counts, shape=(v.frames, 8)).copy()

    # ---- measurement --------------------------------------------------------------------
    def set_profiling(self, on):
        self._check(self._L.pwpp_set_profiling(self._h, 1 if on else 0))

    def reset_kernel_profile(self):
        self._check(self._L.pwpp_reset_kernel_profile(self._h))

    def kernel_profile(self):
        ms = np.zeros(NUM_KERNELS, np.float64)
        cnt = np.zeros(NUM_KERNELS, np.int64)
        self._check(self._L.pwpp_get_kernel_profile(self._h, _vp(ms), _vp(cnt)))
        return {self._L.pwpp_kernel_name(k).decode(): (float(ms[k]), int(cnt[k])) for k in range(NUM_KERNELS)}

    def fxp_shift(self):
        return self._L.pwpp_get_fxp_shift(self._h)

    def fxp_origins(self):
        """(B, 2) float32: the origin of every bin's fixed-point plane-fit sums."""
        nb = self._L.pwpp_get_fxp_origins(self._h, None, 0)  # size query
        self._check(nb)
        out = np.zeros((nb, 2), np.float32)
        b = self._L.pwpp_get_fxp_origins(self._h, _vp(out), out.shape[0])
        self._check(b)
        return out[:b].copy()

    def set_option(self, name, value):
        """Tuning / test switches (pwpp_set_option): fit_plan, fit_concurrent, one_pass, one_pass_min_frames,
        one_pass_scale, overlap_mode, overlap_ranges, fit_streams, bin_block, hi_split, hi_split_zones, debug_flags.
        None of them changes a result."""
        self._check(self._L.pwpp_set_option(self._h, name.encode(), str(value).encode()))

    def workspace_bytes(self):
        return int(self._L.pwpp_get_workspace_bytes(self._h))

    def trim_workspace(self):
        """Free the per-batch workspaces (results of the last call are gone afterwards)."""
        self._check(self._L.pwpp_trim_workspace(self._h))

    def set_output_order(self, reference):
        """True: the points of a patch come out in the reference's order (z-sorted bins); False: scatter order."""
        self._check(self._L.pwpp_set_output_order(self._h, 1 if reference else 0))

    def set_overlap(self, on):
        """True: batches of 128+ frames run as two frame ranges on the handle's two streams (same results)."""
        self._check(self._L.pwpp_set_overlap(self._h, 1 if on else 0))

    def arena_stats(self):
        """(frames whose overgrown parts were moved into the overflow arena on the device, slots per frame, arena slots per frame)."""
        a, b, c = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
        self._check(self._L.pwpp_get_arena_stats(self._h, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)))
        return int(a.value), int(b.value), int(c.value)

    def redo_stats(self):
        """(frames that went through one-pass binning, frames redone on the two-pass path after a segment overflow)"""
        a, b = ctypes.c_int64(0), ctypes.c_int64(0)
        self._check(self._L.pwpp_get_redo_stats(self._h, ctypes.byref(a), ctypes.byref(b)))
        return int(a.value), int(b.value)

    def one_pass_stats(self):
        """(batches launched with one-pass binning, batches redone on the two-pass path after an overflow)"""
        a, b = ctypes.c_int64(0), ctypes.c_int64(0)
        self._check(self._L.pwpp_get_one_pass_stats(self._h, ctypes.byref(a), ctypes.byref(b)))
        return int(a.value), int(b.value)


class Pipe:
    """Batches of independent frames in flight (pwpp_pipe_*): `depth` handles, each batch goes to the next one in turn.
    submit_device_batch returns a Handle VIEW of the handle that holds the batch: synchronize() it, then read the results with
    the usual getters -- before that handle comes round again, `depth` submits later."""

    def __init__(self, params=None, device=0, depth=2):
        self._L = load()
        self.params = params if params is not None else default_params()
        p = ctypes.c_void_p()
        self._p = None
        rc = self._L.pwpp_pipe_create(ctypes.byref(self.params), device, depth, ctypes.byref(p))
        if rc < 0:
            raise PwppError("pwpp error %d: %s" % (rc, self._L.pwpp_last_error().decode()))
        self._p = p
        self.depth = depth
        self._views = {}

    def _view(self, raw):
        key = raw.value if hasattr(raw, "value") else int(raw)
        if key not in self._views:
            v = Handle.__new__(Handle)  # a view: the pipe owns the handle
            v._L, v.params, v._h, v._keep, v._borrowed = self._L, self.params, ctypes.c_void_p(key), None, True
            v._pipe = self  #  a view keeps its pipe alive: the pipe owns the handle the view points at
            self._views[key] = v
        return self._views[key]

    def handle(self, index):
        raw = self._L.pwpp_pipe_handle(self._p, index)
        return self._view(ctypes.c_void_p(raw)) if raw else None

    def set_num_streams(self, streams_per_handle):
        """MODE_STREAMS through the pipe: every handle owns one GROUP of `streams_per_handle` streams; submit k must carry the next
        frames of group k mod depth."""
        rc = self._L.pwpp_pipe_set_num_streams(self._p, streams_per_handle)
        if rc < 0:
            raise PwppError("pwpp error %d: %s" % (rc, self._L.pwpp_last_error().decode()))

    def submit_device_batch(self, batch, cols=4, layout=LAYOUT_ROW_MAJOR, mode=MODE_FRESH):
        cp, cn, k = batch
        holder = ctypes.c_void_p()
        rc = self._L.pwpp_pipe_submit(self._p, cp, cn, k, cols, layout, MEM_DEVICE, mode, ctypes.byref(holder))
        if rc < 0:
            raise PwppError("pwpp error %d: %s" % (rc, self._L.pwpp_last_error().decode()))
        return self._view(holder)

    def drain(self):
        rc = self._L.pwpp_pipe_drain(self._p)
        if rc < 0:
            raise PwppError("pwpp error %d: %s" % (rc, self._L.pwpp_last_error().decode()))

    def workspace_bytes(self):
        return sum(self.handle(i).workspace_bytes() for i in range(self.depth))

    def close(self):
        if self._p is not None:
            for v in self._views.values():  #  the views' handles die with the pipe: a getter on one must fail, not crash
                v._h = None
                v._pipe = None
            self._L.pwpp_pipe_destroy(self._p)
            self._p = None
            self._views = {}

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
