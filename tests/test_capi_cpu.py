"""No-GPU checks of the product library: it loads, exports every symbol include/pwpp.h
declares, mirrors the reference's parameter defaults, validates arguments, and FAILS LOUDLY
without a GPU instead of falling back to a CPU path."""
import ctypes
import os

import numpy as np
import re
import subprocess

import pytest

import oracle_lib as ol
import pwpp_hip

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _has_gpu():
    return pwpp_hip.load().pwpp_device_count() > 0


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(pwpp_hip.LIB_PATH):
        subprocess.run(["make", "-C", os.path.join(ROOT, "patchwork-plusplus_amd"), "lib/libpwpp_hip.so"], check=True,
                       stdout=subprocess.DEVNULL)
    return pwpp_hip.load()


def test_exports_every_declared_symbol(lib):
    hdr = open(os.path.join(ROOT, "include", "pwpp.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = set(re.findall(r"\b(pwpp_[a-z_0-9]+)\s*\(", hdr))
    assert len(names) >= 25
    for n in sorted(names):
        assert hasattr(lib, n), "libpwpp_hip.so does not export %s" % n
    # ... and nothing else: the library is built with -fvisibility=hidden and a version script (VERDICT r02 item 8:
    # pwpp_launch_*, pwpp_debug_read and a kernel stub used to be visible)
    out = subprocess.run(["nm", "-D", "--defined-only", pwpp_hip.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = {ln.split()[-1] for ln in out.splitlines() if ln.strip()}
    assert exported == names, (sorted(exported - names), sorted(names - exported))


def test_params_default_mirror_reference(lib, oracle_built):
    p = pwpp_hip.default_params()
    o = oracle_built.restatement().default_params()
    for name, _ in ol.Params._fields_:
        a, b = getattr(p, name), getattr(o, name)
        if hasattr(a, "__len__"):
            assert list(a) == list(b), name
        else:
            assert a == b, name
    assert ctypes.sizeof(pwpp_hip.State) == 104


def test_create_rejects_bad_params_before_touching_the_gpu(lib):
    def rc_for(**kw):
        p = pwpp_hip.default_params()
        for k, v in kw.items():
            setattr(p, k, v)
        h = ctypes.c_void_p()
        rc = lib.pwpp_create(ctypes.byref(p), 0, ctypes.byref(h))
        if rc == 0:
            lib.pwpp_destroy(h)
        return rc
    assert rc_for(num_zones=3) == -5
    assert rc_for(num_iter=0) == -5
    assert rc_for(num_lpr=1000) == -5
    assert rc_for(num_rings_of_interest=5) == -1
    assert rc_for(max_range=1.0) == -1
    assert b"four zones" in (rc_for(num_zones=5) and lib.pwpp_last_error())


def test_no_cpu_fallback(lib):
    if _has_gpu():
        pytest.skip("a GPU is present")
    with pytest.raises(pwpp_hip.PwppError) as e:
        pwpp_hip.Handle()
    assert "no CPU path" in str(e.value)
    import pypatchworkpp
    with pytest.raises(RuntimeError):
        pypatchworkpp.patchworkpp(pypatchworkpp.Parameters())


def test_pipe_arguments_without_a_gpu(lib):
    """pwpp_pipe_* (batches in flight): argument errors come back before anything touches a device; a valid pipe needs a GPU like
    a handle does -- no CPU fallback behind the pipe either."""
    with pytest.raises(pwpp_hip.PwppError, match="depth"):
        pwpp_hip.Pipe(depth=0)
    with pytest.raises(pwpp_hip.PwppError, match="depth"):
        pwpp_hip.Pipe(depth=5)
    assert lib.pwpp_pipe_handle(None, 0) is None and lib.pwpp_pipe_destroy(None) == 0
    if not _has_gpu():
        with pytest.raises(pwpp_hip.PwppError) as e:
            pwpp_hip.Pipe(depth=2)
        assert "no CPU path" in str(e.value)


def test_pybind_module_surface():
    """Same names as the reference module (python/patchworkpp/pybinding.cpp:9-57)."""
    import pypatchworkpp as m
    assert m.__version__ == "0.0.1"
    p = m.Parameters()
    for f in ("verbose enable_RNR enable_RVPF enable_TGR num_iter num_lpr num_min_pts num_zones "
              "num_rings_of_interest RNR_ver_angle_thr RNR_intensity_thr sensor_height th_seeds th_dist "
              "th_seeds_v th_dist_v max_range min_range uprightness_thr adaptive_seed_selection_margin "
              "intensity_thr num_sectors_each_zone num_rings_each_zone max_flatness_storage "
              "max_elevation_storage elevation_thr flatness_thr").split():
        assert hasattr(p, f), f
    assert p.num_sectors_each_zone == [16, 32, 54, 32] and p.num_rings_each_zone == [2, 4, 4, 4]
    assert p.sensor_height == 1.723 and p.num_min_pts == 10 and p.enable_RNR is True
    p.th_dist = 0.2
    assert p.th_dist == 0.2
    for meth in ("getHeight getTimeTaken getGround getNonground getCenters getGroundIndices "
                 "getNongroundIndices getNormals estimateGround").split():
        assert hasattr(m.patchworkpp, meth), meth


def test_product_never_touches_the_oracle():
    """The judge's rule: nothing in the product may link, import or execute oracle/."""
    pkg = os.path.join(ROOT, "patchwork-plusplus_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".hip", ".cpp", ".h", ".py", "Makefile")):
                txt = open(os.path.join(dirpath, fn), errors="ignore").read()
                assert "oracle_lib" not in txt and "liboracle" not in txt and "pwo_" not in txt, fn
    out = subprocess.run(["ldd", pwpp_hip.LIB_PATH], capture_output=True, text=True).stdout
    assert "oracle" not in out


def test_cpp_demo_builds_against_the_class_mirror():
    """The demo program compiles and links against the C++ header + libpwpp_hip.so (no GPU needed)."""
    exe = os.path.join(ROOT, "patchwork-plusplus_amd", "examples", "demo_sequential")
    subprocess.run(["make", "-C", os.path.join(ROOT, "patchwork-plusplus_amd"), "examples/demo_sequential"], check=True,
                   stdout=subprocess.DEVNULL)
    assert os.path.exists(exe)
    if not _has_gpu():
        r = subprocess.run([exe], capture_output=True, text=True)
        assert r.returncode == 1 and "no CPU path" in r.stdout


def test_c_abi_from_plain_c99():
    """include/pwpp.h is C, not C++ in disguise: a C99 program (gcc -std=c99 -pedantic -Werror) builds against
    it and links with the library alone; without a GPU it reports that there is no device and exits (no CPU
    path), with one it is exercised by the GPU suite."""
    exe = os.path.join(ROOT, "patchwork-plusplus_amd", "examples", "capi_demo")
    subprocess.run(["make", "-C", os.path.join(ROOT, "patchwork-plusplus_amd"), "examples/capi_demo"], check=True,
                   stdout=subprocess.DEVNULL)
    assert os.path.exists(exe)
    if _has_gpu():
        return
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 2 and "no CPU path" in r.stderr


def _numpy_bins(pts, p):
    """pc2czm (reference patchworkpp.cpp:578-622) in numpy doubles: the bin of every point, -1 outside (min_range, max_range]."""
    import numpy as np
    x, y = pts[:, 0].astype(np.float64), pts[:, 1].astype(np.float64)
    r = np.sqrt(x * x + y * y)
    th = np.arctan2(y, x)
    th = np.where(th > 0, th, th + 2 * np.pi)
    mn, mx = p.min_range, p.max_range
    mr = [mn, (7 * mn + mx) / 8, (3 * mn + mx) / 4, (mn + mx) / 2, mx]
    rings, sect = list(p.num_rings_each_zone), list(p.num_sectors_each_zone)
    base = np.cumsum([0] + [a * b for a, b in zip(rings, sect)])
    ok = (r > mn) & (r <= mx)
    k = np.digitize(r, mr[1:4])
    code = np.full(len(r), -1)
    for z in range(4):
        m = ok & (k == z)
        ring = np.minimum(((r[m] - mr[z]) / ((mr[z + 1] - mr[z]) / rings[z])).astype(int), rings[z] - 1)
        sec = np.minimum((th[m] / (2 * np.pi / sect[z])).astype(int), sect[z] - 1)
        code[m] = base[z] + ring * sect[z] + sec
    return code


def test_bin_boxes_contain_every_point_the_reference_bins_there(lib):
    """The fit kernels skip the high part of a bin when no point of the box [bin box] x [split height, inf) can lie below
    the plane (pwpp_fit.hip, stage_needs_hi): that is only exact if every point the reference bins into b lies inside
    box b.  Host logic, no GPU: random clouds plus points on every boundary the binning has (ring and sector edges nudged
    by a few float ulps, the axes, the diagonals, min / max range), for CZM shapes from one sector per ring to 128."""
    import numpy as np
    lib.pwpp_get_bin_boxes.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    rng = np.random.default_rng(3)
    shapes = [((16, 32, 54, 32), (2, 4, 4, 4), 2.7, 80.0), ((36, 36, 36, 36), (2, 4, 4, 4), 2.7, 80.0),
              ((1, 1, 1, 1), (1, 1, 1, 1), 2.7, 80.0), ((2, 3, 5, 7), (1, 2, 3, 1), 0.3, 200.0),
              ((128, 128, 128, 128), (4, 4, 4, 4), 1.0, 50.0), ((4, 8, 16, 4), (5, 1, 2, 3), 5.0, 20.0)]
    for sect, rings, mn, mx in shapes:
        p = pwpp_hip.default_params()
        p.min_range, p.max_range = mn, mx
        for k in range(4):
            p.num_sectors_each_zone[k] = sect[k]
            p.num_rings_each_zone[k] = rings[k]
        nb = lib.pwpp_get_bin_boxes(ctypes.byref(p), None, 0)
        assert nb == sum(a * b for a, b in zip(sect, rings))
        boxes = np.zeros((nb, 4), np.float32)
        assert lib.pwpp_get_bin_boxes(ctypes.byref(p), boxes.ctypes.data_as(ctypes.c_void_p), nb) == nb
        mr = [mn, (7 * mn + mx) / 8, (3 * mn + mx) / 4, (mn + mx) / 2, mx]
        radii, angles = [mn, mx], [0.0]
        for z in range(4):
            radii += [mr[z] + i * (mr[z + 1] - mr[z]) / rings[z] for i in range(rings[z] + 1)]
            angles += [i * 2 * np.pi / sect[z] for i in range(sect[z] + 1)]
        angles += [q * np.pi / 4 for q in range(9)]
        pts = []
        for r in radii:
            for dr in (-3e-7, -1e-7, 0.0, 1e-7, 3e-7):
                rr = r * (1 + dr)
                a = rng.uniform(0, 2 * np.pi, 40)
                pts.append(np.stack([rr * np.cos(a), rr * np.sin(a)], 1))
                aa = np.array(angles)
                pts.append(np.stack([rr * np.cos(aa), rr * np.sin(aa)], 1))
        for a in angles:
            for da in (-3e-7, -1e-7, 0.0, 1e-7, 3e-7):
                rr = rng.uniform(mn, mx, 25)
                pts.append(np.stack([rr * np.cos(a + da), rr * np.sin(a + da)], 1))
        rr, a = rng.uniform(0.5 * mn, 1.05 * mx, 20000), rng.uniform(0, 2 * np.pi, 20000)
        pts.append(np.stack([rr * np.cos(a), rr * np.sin(a)], 1))
        for ux, uy in ((1, 0), (-1, 0), (0, 1), (0, -1), (1, 1), (-1, 1), (1, -1), (-1, -1)):  # exact axes and diagonals
            rr = rng.uniform(mn, mx, 50).astype(np.float32)
            pts.append(np.stack([rr * ux / np.hypot(ux, uy), rr * uy / np.hypot(ux, uy)], 1))
        xy = np.concatenate(pts).astype(np.float32)
        code = _numpy_bins(xy, p)
        m = code >= 0
        b = boxes[code[m]]
        assert (xy[m, 0] >= b[:, 0]).all() and (xy[m, 0] <= b[:, 1]).all() and (xy[m, 1] >= b[:, 2]).all() and (xy[m, 1] <= b[:, 3]).all()
        # ... and the boxes are not absurdly generous: each is inside the circle of radius max_range (+ its margin)
        assert np.abs(boxes).max() <= mx * 1.0002 + 0.002
        # (a bin of a quarter turn or less is a proper sector: its box is smaller than the full disc's)
        if min(sect) >= 4:
            assert ((boxes[:, 1] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 2])).max() < (2 * mx) ** 2 * 0.5


def test_fixed_point_geometry_matches_the_restatement(lib, oracle_built):
    """Shift and per-bin origins of the plane-fit sums are computed on the host in pwpp_create and,
    independently, by the CPU restatement: they must agree for every CZM shape and range -- host logic, no GPU."""
    import numpy as np
    lib.pwpp_get_fxp_geometry.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    ora = oracle_built.restatement()
    rng = np.random.default_rng(11)
    shapes = [((16, 32, 54, 32), (2, 4, 4, 4), 2.7, 80.0), ((1, 1, 1, 1), (1, 1, 1, 1), 2.7, 80.0), ((3, 3, 3, 3), (2, 2, 2, 2), 2.7, 80.0),
              ((128, 128, 128, 128), (4, 4, 4, 4), 1.0, 50.0), ((16, 32, 54, 32), (2, 4, 4, 4), 2.7, 500.0), ((4, 4, 4, 4), (1, 1, 1, 1), 0.3, 8000.0)]
    for _ in range(20):
        shapes.append((tuple(int(v) for v in rng.choice([1, 2, 3, 4, 5, 8, 16, 36, 54, 64], 4)), tuple(int(v) for v in rng.integers(1, 7, 4)),
                       float(rng.uniform(0.1, 5.0)), float(rng.uniform(10.0, 3000.0))))
    for sect, rings, mn, mx in shapes:
        p = pwpp_hip.default_params()
        p.min_range, p.max_range = mn, mx
        op = ora.default_params()
        op.min_range, op.max_range = mn, mx
        for k in range(4):
            p.num_sectors_each_zone[k] = op.num_sectors_each_zone[k] = sect[k]
            p.num_rings_each_zone[k] = op.num_rings_each_zone[k] = rings[k]
        nb = sum(a * b for a, b in zip(sect, rings))
        shift = ctypes.c_int(-1)
        xy = np.zeros((nb, 2), np.float32)
        assert lib.pwpp_get_fxp_geometry(ctypes.byref(p), ctypes.byref(shift), xy.ctypes.data_as(ctypes.c_void_p), nb) == nb
        sh, zr, ox, oy = ol.Estimator(ora, op, arith=ol.ARITH_FXP).fxp_geometry()
        assert shift.value == sh and zr == 2.0 ** (35 - sh), (sect, rings, mn, mx)  # contract v4: |Q| <= 2^35 on a 2^-sh m grid
        assert np.array_equal(xy[:, 0], ox) and np.array_equal(xy[:, 1], oy), (sect, rings, mn, mx)


def test_plane_distance_is_monotone_in_every_coordinate():
    """The exact skip of a bin's high part (pwpp_fit.hip, stage_needs_hi) evaluates the reference's distance expression
    (patchworkpp.cpp:551-554: three float products, two float adds, one double add -- no FMA contraction) at ONE corner of
    the box [xmin, xmax] x [ymin, ymax] x [zs, inf) and concludes that every point of the box is at least as far.  That holds
    because every correctly rounded operation is monotone in each operand: checked here in the same float32 / float64
    arithmetic on millions of (plane, box, point) triples, including planes of wildly different scales, tiny boxes (points
    a few ulps apart), huge coordinates and zero components."""
    import numpy as np
    rng = np.random.default_rng(2)

    def dist(nx, ny, nz, d, x, y, z):
        f = np.float32
        return ((f(nx) * f(x) + f(ny) * f(y)).astype(f) + f(nz) * f(z)).astype(f).astype(np.float64) + np.float64(d)

    with np.errstate(over="ignore", invalid="ignore"):
        for scale, width in ((1.0, 10.0), (1.0, 1e-5), (100.0, 50.0), (1e-3, 1.0), (1e6, 1e3), (1.0, 0.0)):
            n = 400000
            nrm = rng.normal(size=(n, 3)).astype(np.float32)
            nrm[:, 2] = np.abs(nrm[:, 2])                       # the reference flips the normal to n_z >= 0 (:68)
            nrm[rng.random(n) < 0.1, 0] = 0.0
            nrm[rng.random(n) < 0.1, 2] = 0.0
            d = (rng.normal(size=n) * scale).astype(np.float32).astype(np.float64)
            lo = (rng.normal(size=(n, 3)) * scale).astype(np.float32)
            hi = (lo.astype(np.float64) + np.abs(rng.normal(size=(n, 3))) * width).astype(np.float32)
            hi = np.maximum(hi, lo)
            t = rng.random((n, 3)).astype(np.float32)
            pt = np.clip((lo + (hi - lo) * t).astype(np.float32), lo, hi)
            pt[:, 2] = np.maximum(pt[:, 2], lo[:, 2])          # z only bounded below (box open to +inf)
            pt[rng.random(n) < 0.2, 2] += np.float32(abs(scale) * 3)
            cx = np.where(nrm[:, 0] >= 0, lo[:, 0], hi[:, 0])
            cy = np.where(nrm[:, 1] >= 0, lo[:, 1], hi[:, 1])
            corner = dist(nrm[:, 0], nrm[:, 1], nrm[:, 2], d, cx, cy, lo[:, 2])
            point = dist(nrm[:, 0], nrm[:, 1], nrm[:, 2], d, pt[:, 0], pt[:, 1], pt[:, 2])
            ok = np.isfinite(corner) & np.isfinite(point)
            assert (point[ok] >= corner[ok]).all(), (scale, width)


def test_float_threshold_of_the_plane_tests_is_exact():
    """The streamed passes compare the float part s of the reference's distance expression with ONE float threshold T instead
    of evaluating  double(s) + d < thr  per point (pwpp_common.hpp, plane_test_threshold).  Restated here step for step
    (start at the real-number boundary, walk the ordered float keys, binary search as the backstop) and checked against
    the definition -- T is the smallest float that fails -- on thresholds and offsets of every scale, the ends of the float
    range, infinities and NaN, and by brute force over every float around T."""
    rng = np.random.default_rng(11)

    def key(f):
        b = np.float32(f).view(np.uint32)
        return int(~b & 0xffffffff) if b & 0x80000000 else int(b | 0x80000000)

    def unkey(k):
        k = np.uint32(k)
        return (np.uint32(k & 0x7fffffff) if k & 0x80000000 else np.uint32(~k)).view(np.float32)

    KMIN, KMAX = 0x007fffff, 0xff800000
    assert unkey(KMIN) == -np.inf and unkey(KMAX) == np.inf and key(-0.0) + 1 == key(0.0)

    with np.errstate(invalid="ignore", over="ignore"):  # (inf - inf, inf + -inf: NaN on purpose)
        def passes(k, d, thr):
            return bool(np.float64(unkey(k)) + d < thr)

        def threshold(d, thr):
            if not passes(KMIN, d, thr):
                return KMIN
            t = np.float32(thr - d)
            k = key(t) if t == t else KMAX
            k = min(max(k, KMIN), KMAX)
            for _ in range(3):
                if k < KMAX and passes(k, d, thr):
                    k += 1
            for _ in range(3):
                if k > KMIN and not passes(k - 1, d, thr):
                    k -= 1
            if passes(k, d, thr) or not passes(k - 1, d, thr):
                lo, hi = KMIN, KMAX
                while hi - lo > 1:
                    mid = lo + ((hi - lo) >> 1)
                    if passes(mid, d, thr):
                        lo = mid
                    else:
                        hi = mid
                k = hi
            return k

        cases = [(0.0, 0.125), (1.723, 0.125), (-1.723, 0.125), (0.0, -1.6), (1e-30, 1e-30), (3e38, 1.0), (-3e38, 1.0), (0.0, 3.5e38),
                 (0.0, -3.5e38), (np.inf, 1.0), (-np.inf, 1.0), (np.nan, 1.0), (1.0, np.nan), (1.0, np.inf), (1.0, -np.inf), (0.0, 0.0),
                 (1e-320, 1e-320), (2.0 ** 60, 2.0 ** 60 + 4096.0)]
        for _ in range(3000):
            s = 10.0 ** rng.uniform(-12, 12)
            cases.append((float(rng.normal() * s), float(rng.normal() * s * 10.0 ** rng.uniform(-3, 3))))
        for d, thr in cases:
            d, thr = np.float64(d), np.float64(thr)
            k = threshold(d, thr)
            T = unkey(k)
            if k == KMIN:
                assert not passes(KMIN, d, thr) or T == -np.inf
            else:
                assert not passes(k, d, thr) and passes(k - 1, d, thr), (d, thr, T)
            # the test the kernels run (s < T) against the reference's expression, on the floats around T and a few others
            for kk in list(range(max(KMIN, k - 6), min(KMAX, k + 6) + 1)) + [KMIN, KMAX, key(0.0), key(-0.0), key(1.0), key(-1.0)]:
                sv = unkey(kk)
                assert bool(sv < T) == passes(kk, d, thr), (d, thr, sv, T)
            assert not (np.float32(np.nan) < T)


def test_field_blobs_must_be_float_aligned(lib):
    """a PointCloud2 blob is read in place as float32 -- a data pointer off a 4-byte boundary is refused before
    anything touches it (no handle, no GPU needed for the check)."""
    buf = (ctypes.c_uint8 * 256)()
    base = ctypes.addressof(buf)
    base += (-base) % 16
    n = (ctypes.c_int32 * 1)(4)
    for off, want in ((1, -1), (2, -1), (3, -1)):
        ptrs = (ctypes.c_void_p * 1)(base + off)
        rc = lib.pwpp_estimate_ground_fields_batch(None, ptrs, n, 1, 16, 0, 4, 8, -1, 0, 0)
        assert rc == want and b"aligned" in lib.pwpp_last_error()
    ptrs = (ctypes.c_void_p * 1)(base)  # aligned: the call gets as far as the missing handle
    assert lib.pwpp_estimate_ground_fields_batch(None, ptrs, n, 1, 16, 0, 4, 8, -1, 0, 0) == -1
    assert b"aligned" not in lib.pwpp_last_error()


def test_ros_node_core_compiles_without_ros():
    """The ROS 2 node's logic (ros/include/patchworkpp_ros/segmentation_core.hpp) is plain C++17 over the class mirror: it must
    compile where there is no ROS, and its parameter mapping must name exactly the parameters the reference's node declares
    (ros/src/GroundSegmentationServer.cpp:28-44 there).  The rclcpp glue (ros/src/ground_segmentation_server.cpp) is checked
    for the topic and parameter names only -- it cannot be compiled here."""
    import re
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg = os.path.join(root, "patchwork-plusplus_amd")
    subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Werror", "-I", os.path.join(pkg, "ros", "include"), "-I", os.path.join(pkg, "include"),
                    "-I", os.path.join(root, "include"), os.path.join(pkg, "examples", "ros_core_demo.cpp")], check=True)
    core = open(os.path.join(pkg, "ros", "include", "patchworkpp_ros", "segmentation_core.hpp")).read()
    names = set(re.findall(r'get_(?:double|int|bool)\("([a-z_]+)"', core))
    assert names == {"sensor_height", "num_iter", "num_lpr", "num_min_pts", "th_seeds", "th_dist", "th_seeds_v", "th_dist_v", "max_range",
                     "min_range", "uprightness_thr", "verbose"}
    glue = open(os.path.join(pkg, "ros", "src", "ground_segmentation_server.cpp")).read()
    for topic in ('"pointcloud_topic"', '"/patchworkpp/cloud"', '"/patchworkpp/ground"', '"/patchworkpp/nonground"', '"base_frame"', '"patchworkpp_node"'):
        assert topic in glue
    launch = open(os.path.join(pkg, "ros", "launch", "patchworkpp.launch.py")).read()
    for k, v in (("sensor_height", "1.88"), ("num_min_pts", "0"), ("th_seeds", "0.3"), ("uprightness_thr", "0.101"), ("min_range", "1.0")):
        assert re.search(r'"%s": %s[,}\s]' % (k, re.escape(v)), launch), k
