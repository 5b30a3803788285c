#!/usr/bin/env python3
"""bench.py -- frames/s of the MI355X-native Patchwork++ estimateGround() hot path.

Contract (driver): ``python bench.py --gpus N --steps K --warmup W`` prints ONE JSON line on
rank 0.  For N > 1 it is launched under ``torch.distributed.run`` with one rank per GPU.

* A *step* is one pass of the hot path (RNR -> CZM binning -> per-patch plane fitting ->
  GLE/TGR -> index lists) over one batch of ``--frames`` (default 1024) frames that are already
  resident in HBM as 1024 distinct device buffers (2 GB, far beyond the 256 MiB Infinity
  Cache): BASELINE.json configs[2], "Batch of 1024 replayed KITTI frames on 1 MI355X".  Every
  frame is processed with fresh state (= a fresh reference object per frame).  The timed region
  keeps ``--in-flight`` (default 2) batches enqueued, one library handle each -- the way a caller
  with a stream of batches drives the C-ABI (double buffering); ``synchronous`` in the line is the
  one-batch-at-a-time step rounds 1-4 timed.
* Frames: the six KITTI sample frames of the reference (tests/golden/kitti_*.bin.xz,
  byte-identical to /root/reference/data/*.bin) replayed round-robin; synthetic 64-beam
  frames (pwpp_synth.make_cloud) if the fixtures are missing.
* N GPUs: every rank processes its own batch (weak scaling, frames are independent -- no
  data-path collective); RCCL is used only to agree on the slowest rank's time.
* ``roofline``: HBM roofline of the dominant kernel.  achieved = algorithmic bytes of the
  batch (16 N + 4 N + 24 P per frame) / that kernel's mean duration, measured
  with HIP events on the library's own stream inside the timed region.
* ``cpu_baseline``: the reference's own patchworkpp.cpp (oracle/_ref, "reference") or the
  restatement ("port"), frame-parallel on the host cores, on a bounded sample.
* ``latency``: BASELINE.json configs[1], one frame end to end on one GPU (launch-bound).
"""
import argparse
import json
import lzma
import os
import sys
import time

import numpy as np

# RCCL / device-memory sharing between the ranks of one node needs dmabuf IPC on this driver (the image exports this
# already; set here before the HIP runtime starts in case a launcher scrubs the environment)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "patchwork-plusplus_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); 6290 GB/s measured-achievable


def load_source_frames(kind):
    if kind == "dense":
        import pwpp_synth
        return [pwpp_synth.make_dense_cloud(1000 + k) for k in range(4)], "synthetic-128beam-500k"
    frames = []
    gold = os.path.join(ROOT, "tests", "golden")
    for k in range(6):
        p = os.path.join(gold, "kitti_%06d.bin.xz" % k)
        if not os.path.exists(p):
            frames = []
            break
        with lzma.open(p, "rb") as f:
            frames.append(np.frombuffer(f.read(), np.float32).reshape(-1, 4).copy())
    if frames:
        return frames, "kitti-sample-x6-replayed"
    import pwpp_synth
    return [pwpp_synth.make_cloud(1000 + k) for k in range(6)], "synthetic-64beam"


def cpu_limits():
    """What the host lets this process use: hardware threads, affinity mask, cgroup CPU quota (the frame-parallel baseline
    stops scaling at ~15 workers on the GPU boxes of this pool -- this says whether a quota is why)."""
    out = {"os_cpu_count": os.cpu_count()}
    try:
        out["affinity"] = len(os.sched_getaffinity(0))
    except Exception:
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
        try:
            out[os.path.basename(path)] = open(path).read().strip()
        except Exception:
            pass
    try:
        out["loadavg"] = open("/proc/loadavg").read().split()[:3]
    except Exception:
        pass
    return out


def quota_cores():
    """CPUs' worth of time the cgroup grants this container (cpu.max = quota period), or None: the GPU boxes of this pool show 256
    hardware threads and grant 16 -- which is where the frame-parallel baseline stops scaling."""
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else float(q) / float(p)
    except Exception:
        return None


def cpu_baseline(src, budget_s=4.0):
    """Reference CPU path on the host cores, bounded sample (rank 0, N=1 only).

    VERDICT r04 item 6: the frame-parallel harness as PROCESSES -- one single-threaded worker process per core group
    (tests/oracle_lib.py run as a script: its own heap, its own copy of the reference build), started together on a wall-clock
    mark -- so that glibc's allocator is not shared between the frames in flight.  `value` is the best of {1, 32, 64, 128, all}
    workers; the per-count table stays in the line, and so does the threads-in-one-process number of rounds 1-4."""
    import subprocess
    import tempfile
    import oracle_lib as ol
    lib, kind, arith = ol.reference(ol.ARITH_EIGEN_F32), "reference", ol.ARITH_EIGEN_F32
    if lib is None:
        lib, kind = ol.restatement(), "port"
    cores = os.cpu_count() or 1
    w1, _ = ol.cpu_bench(lib, src, len(src), 1, arith=arith)  # one pass, one thread
    per_frame = w1 / len(src)
    fps1 = 1.0 / per_frame
    per_worker = max(len(src), int(budget_s / max(per_frame, 1e-6)))
    per_worker -= per_worker % len(src)
    table = []
    scratch = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
    with tempfile.TemporaryDirectory(dir=scratch) as tmp:
        path = os.path.join(tmp, "frames.npz")
        np.savez(path, *src)
        for workers in sorted(set(w for w in (1, 32, 64, 128, cores) if w <= cores)):
            t_go = time.time() + 3.0 + 0.02 * workers  # every worker has loaded its library and the frames by then (late ones say so)
            # (bounded sample: beyond 32 workers the per-worker share shrinks, so that a machine whose rate stops growing there --
            # the GPU boxes of this pool: 256 hardware threads, ~15 cores' worth of throughput -- does not spend minutes here)
            mine = per_worker if workers <= 32 else max(len(src), per_worker * 32 // workers // len(src) * len(src))
            cmd = [sys.executable, os.path.join(ROOT, "tests", "oracle_lib.py"), "--bench-worker", path, str(mine), repr(t_go),
                   kind, str(arith)]
            env = dict(os.environ, OMP_NUM_THREADS="1")
            procs = [subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, env=env) for _ in range(workers)]
            begins, ends, frames = [], [], 0
            for pr in procs:
                try:
                    out, _ = pr.communicate(timeout=180)
                    b, e, n = out.split()[-3:]
                    begins.append(float(b)), ends.append(float(e))
                    frames += int(n)
                except Exception:
                    pr.kill()
            if frames:
                wall = max(ends) - min(begins)
                table.append({"workers": workers, "frames": frames, "wall_s": wall, "frames_per_s": frames / wall,
                              "scaling": frames / wall / fps1 / workers, "late_starters": sum(1 for b in begins if b > t_go + 0.05)})
    # rounds 1-4: all frames in ONE process on `cores` threads (oracle/ref_capi.cpp pwo_bench) -- allocator-bound
    total = int(max(cores * 2, min(2048, 3.0 / max(per_frame, 1e-6) * cores)))
    total -= total % cores
    wall_t, _ = ol.cpu_bench(lib, src, total, cores, arith=arith)
    threads_row = {"threads": cores, "frames_per_s": total / wall_t, "scaling": total / wall_t / fps1 / cores}
    best = max(table, key=lambda r: r["frames_per_s"]) if table else {"workers": cores, "frames_per_s": total / wall_t, "scaling": threads_row["scaling"], "frames": total, "wall_s": wall_t}
    return {
        "value": best["frames_per_s"], "unit": "frames/s", "cores": best["workers"], "kind": kind,
        "host_threads": cores, "cpu_limits": cpu_limits(), "single_thread_fps": fps1,
        "scaling": best["scaling"],  # value / single-thread rate / workers: 1.0 = linear
        "by_workers": table, "threads_in_one_process": threads_row,
        "cpu_quota_cores": quota_cores(),
        "sample": "%d frames (%d distinct source frames replayed, fresh reference object per frame) in %.1f s wall on %d single-threaded "
                  "worker processes -- the best of the worker counts in `by_workers`; g++ -O3 build of the reference's own patchworkpp.cpp "
                  "+ Eigen stand-in (oracle/_ref); the reference has no OpenMP in this path, frame-parallelism across processes is the harness's"
                  % (best["frames"], len(src), best["wall_s"], best["workers"]),
    }


def ingest_leg(pwpp_hip, src, gpu_index, chunk=256, chunks=8):
    """Host-resident frames end to end: every frame is copied H2D (2 MB) and its index lists D2H (0.5 MB)."""
    bufs, outs = [], []
    for _ in range(2):
        rows = sum(src[i % len(src)].shape[0] for i in range(chunk))
        slab, at, fr = pwpp_hip.pinned_empty((rows, 4)), 0, []
        for i in range(chunk):
            a = src[i % len(src)]
            slab[at:at + a.shape[0]] = a
            fr.append(slab[at:at + a.shape[0]])
            at += a.shape[0]
        bufs.append(fr)
        outs.append(pwpp_hip.pinned_empty((rows,), np.int32))
    mb_in = sum(f.nbytes for f in bufs[0]) / 1e6
    H = [pwpp_hip.Handle(device=gpu_index), pwpp_hip.Handle(device=gpu_index)]
    for k in range(2):  # warm-up: allocations
        H[k].submit_pinned_batch(bufs[k])
        H[k].all_indices(outs[k])
    t0 = time.perf_counter()
    ground = 0
    for k in range(chunks):
        H[k % 2].submit_pinned_batch(bufs[k % 2])
        if k > 0:
            _, _, counts = H[(k - 1) % 2].all_indices(outs[(k - 1) % 2])
            ground += int(counts[:, 0].sum())
    _, _, counts = H[(chunks - 1) % 2].all_indices(outs[(chunks - 1) % 2])
    ground += int(counts[:, 0].sum())
    dt = time.perf_counter() - t0
    for hh in H:
        hh.close()
    return {"frames_per_s": chunks * chunk / dt, "h2d_GBps": chunks * mb_in / 1e3 / dt, "chunk_frames": chunk, "chunks": chunks,
            "ground_points": ground,
            "what": "page-locked host slabs -> H2D -> pipeline -> D2H of all index lists, two handles double-buffered (PWPP_MEM_HOST_PINNED)"}


def estimate_memory_gb(args):
    """What one rank's GPU must hold (checked before RCCL is initialised): the input buffers of every batch in flight and the workspaces
    of the handles -- per point 16 B of input and ~26 B of workspace per handle (bin-ordered planes at ~1.5 slots per point, index
    lists, codes), frames of ~125 k points (64-beam) or ~480 k (dense); the extra legs of a single-GPU run (1024 dense frames on two
    handles) are the larger item there."""
    pts = 480e3 if args.workload == "dense" else 125e3
    handles = 1 + max(1, min(4, args.in_flight))
    if args.workload == "streams":
        return (6 * args.frames * pts * 16 + args.frames * pts * 30) / 1e9 + 1.0
    main = args.frames * pts * (16.0 * max(1, min(4, args.in_flight)) + 30.0 * handles)
    extras = 0.0
    if args.gpus == 1 and not args.skip_extras and args.workload == "kitti":
        # (measured, round 6: 1024 dense frames from 64 clouds hold 48 GB per handle -- the segments follow every part's largest count over
        # all clouds -- 1024 varied 64-beam frames 15 GB; both legs keep two handles)
        extras = max(args.dense_frames * 480e3 * (16.0 + 2 * 100.0), args.distinct_frames * 125e3 * (2 * 16.0 + 2 * 125.0 + 2 * 36.0))
    return (main + extras) / 1e9 + 2.0


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: re-run this command line as N ranks under torch.distributed.run
    (static rendezvous on 127.0.0.1 and a free port: the container's hostname may not resolve).  Returns the exit code."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, PWPP_BENCH_SELF_SPAWNED="1")
    return subprocess.call(cmd, env=env)


def mask_sha256(n, ground_idx):
    import hashlib
    m = np.zeros(n, np.uint8)
    m[np.asarray(ground_idx)] = 1
    return hashlib.sha256(np.packbits(m).tobytes()).hexdigest()


def hash_spot_check(h, kind, frame_of, ns):
    """Ground masks of a few frames of a synthetic leg against tests/golden/frame_hashes.json (written by tests/golden/make_frame_hashes.py
    from the CPU restatement of the contract; data only -- nothing under oracle/ runs here).  frame_of: golden frame number -> batch index."""
    path = os.path.join(ROOT, "tests", "golden", "frame_hashes.json")
    if not os.path.exists(path):
        return None
    gold = json.load(open(path))[kind]
    checked, bad = [], []
    for key, g in gold.items():
        i = frame_of(int(key))
        if i is None:
            continue
        checked.append(int(key))
        if ns[i] != g["points"] or mask_sha256(ns[i], h.ground_indices(i)) != g["sha256"]:
            bad.append(int(key))
    return {"frames_checked": checked, "mismatches": bad, "ok": len(checked) > 0 and not bad,
            "against": "tests/golden/frame_hashes.json: SHA-256 of the ground mask per frame, from the CPU restatement of the contract"}


def two_in_flight(pwpp_hip, torch, h0, batch0, make_second, frames, n, gpu_index, params=None):
    """frames/s of the headline's schedule on (h0, batch0): a second handle joins (make_second(handle) -> its batch), two batches in
    flight, every handle on one stream.  Returns (frames/s, frames the second -- cold -- handle had to redo)."""
    h1 = pwpp_hip.Handle(params, device=gpu_index) if params is not None else pwpp_hip.Handle(device=gpu_index)
    hs, bs = [h0, h1], [batch0, make_second(h1)]
    for hh in hs:
        hh.set_overlap(False)

    def run(m):
        for k in range(m):
            if k >= 2:
                hs[k % 2].synchronize()
            hs[k % 2].launch_device_batch(bs[k % 2], cols=4, mode=pwpp_hip.MODE_FRESH)
        for hh in hs:
            hh.synchronize()

    run(6)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(n)
    torch.cuda.synchronize()
    rate = frames * n / (time.perf_counter() - t0)
    redone = h1.redo_stats()[1]
    h1.close()
    h0.set_overlap(True)
    return rate, redone


def dense_leg(pwpp_hip, torch, dev, gpu_index, frames=1024, steps=5, distinct=64):
    """BASELINE.json configs[4] on this GPU, outside the timed region: dense synthetic 128-beam ~480k-point frames, 36-sector CZM --
    `frames` device buffers filled from `distinct` different clouds (pwpp_synth.dense_frame(0..distinct-1), generated by worker
    processes), the headline's schedule, a spot check of the ground masks against committed hashes."""
    import pwpp_synth
    t0 = time.perf_counter()
    distinct = max(1, min(distinct, frames))
    src = pwpp_synth.dense_frames(distinct)
    gen_s = time.perf_counter() - t0
    ns = [src[i % distinct].shape[0] for i in range(frames)]
    offs = np.concatenate([[0], np.cumsum(ns)]).astype(np.int64)
    big = torch.empty((int(offs[-1]), 4), dtype=torch.float32, device=dev)
    for k in range(distinct):
        t = torch.from_numpy(src[k]).to(dev)
        for i in range(k, frames, distinct):
            big[offs[i]:offs[i + 1]].copy_(t)
        del t
    torch.cuda.synchronize()
    params = pwpp_hip.default_params()
    for k in range(4):
        params.num_sectors_each_zone[k] = 36
    h = pwpp_hip.Handle(params, device=gpu_index)
    batch = h.make_device_batch([big.data_ptr() + int(offs[i]) * 16 for i in range(frames)], ns)

    def step():
        h.launch_device_batch(batch, cols=4, mode=pwpp_hip.MODE_FRESH)
        h.synchronize()

    step()
    redo_cold = h.redo_stats()[1]
    counts = h.all_counts()
    for i in range(frames):
        assert counts[i, 0] + counts[i, 1] + counts[i, 5] == ns[i], "dense leg: partition property violated in frame %d" % i
    parity = hash_spot_check(h, "dense", lambda k: k if k < distinct and k < frames else None, ns)
    assert parity is None or parity["ok"] or os.environ.get("PWPP_BENCH_NO_SELFCHECK"), "dense leg: ground masks differ from the committed hashes: %r" % parity
    for _ in range(3):  # (warm-up after the host-side check, which leaves the GPU idle)
        step()
    r0 = h.redo_stats()[1]
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    dt = (time.perf_counter() - t0) / steps
    b_alg = float(sum(20 * ns[i] + 24 * int(counts[i, 2]) for i in range(frames)))
    ws = h.workspace_bytes() / 1e9
    dptrs = [big.data_ptr() + int(offs[i]) * 16 for i in range(frames)]
    fps2, redo_second = two_in_flight(pwpp_hip, torch, h, batch, lambda hh: hh.make_device_batch(dptrs, ns), frames, 2 * steps, gpu_index, params)
    redone_steady = h.redo_stats()[1] - r0
    h.close()
    dt_sync, dt = dt, frames / fps2
    return {"schedule": "as the headline: two batches in flight, one handle each", "synchronous": {"frames_per_s": frames / dt_sync, "ms_per_step": 1000.0 * dt_sync},
            "workload": "configs[4] on one GPU: %d dense synthetic 128-beam frames (~%d points each) from %d distinct clouds, 36-sector CZM, device-resident "
                        "(%.2f GB), fresh state" % (frames, int(np.mean(ns)), distinct, offs[-1] * 16 / 1e9),
            "frames": frames, "distinct_clouds": distinct, "steps": steps, "frames_per_s": frames / dt, "ms_per_step": 1000.0 * dt,
            "algorithmic_bytes_per_step": b_alg, "pipeline_achieved_GBps": b_alg / dt / 1e9, "pipeline_frac": b_alg / dt / 1e9 / HBM_PEAK_GBS,
            "redone_frames": {"first_batch_cold_handle": redo_cold, "steady_state": redone_steady, "second_cold_handle": redo_second},
            "parity": parity, "workspace_gb": ws, "generated_in_s": gen_s}


def distinct_leg(pwpp_hip, torch, dev, gpu_index, frames=1024, steps=10):
    """the headline on NON-replayed data.  `frames` distinct synthetic 64-beam frames (pwpp_synth.varied_frame:
    110-130 k points, terrain / clutter / sensor height all varied) in distinct device buffers, through a COLD handle: its first
    batch (table of bin segments from a 32-frame probe), a batch of frames the handle has never seen, then the steady state.
    An overflowing frame is redone alone (pwpp_get_redo_stats counts frames).  Outside the timed region."""
    import pwpp_synth
    t0 = time.perf_counter()
    src = pwpp_synth.varied_frames(frames)
    gen_s = time.perf_counter() - t0
    ns = [a.shape[0] for a in src]
    offs = np.concatenate([[0], np.cumsum(ns)]).astype(np.int64)
    big = torch.from_numpy(np.concatenate(src, axis=0)).to(dev)
    torch.cuda.synchronize()
    ptrs = [big.data_ptr() + int(offs[i]) * 16 for i in range(frames)]
    h = pwpp_hip.Handle(device=gpu_index)
    half = frames // 2

    def run(batch):
        r0 = h.redo_stats()[1]
        t1 = time.perf_counter()
        h.launch_device_batch(batch, cols=4, mode=pwpp_hip.MODE_FRESH)
        h.synchronize()
        return 1000.0 * (time.perf_counter() - t1), h.redo_stats()[1] - r0

    first = h.make_device_batch(ptrs[:half], ns[:half])
    unseen = h.make_device_batch(ptrs[half:], ns[half:])
    whole = h.make_device_batch(ptrs, ns)
    ms_first, redo_first = run(first)
    ms_unseen, redo_unseen = run(unseen)
    ms_whole, redo_whole = run(whole)
    counts = h.all_counts()
    for i in range(frames):
        assert counts[i, 0] + counts[i, 1] + counts[i, 5] == ns[i], "distinct leg: partition property violated in frame %d" % i
    parity = hash_spot_check(h, "varied", lambda k: k if k < frames else None, ns)  # (not only the partition property)
    assert parity is None or parity["ok"] or os.environ.get("PWPP_BENCH_NO_SELFCHECK"), "distinct leg: ground masks differ from the committed hashes: %r" % parity
    for _ in range(3):
        run(whole)
    r0 = h.redo_stats()[1]
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(steps):
        h.launch_device_batch(whole, cols=4, mode=pwpp_hip.MODE_FRESH)
        h.synchronize()
    dt = (time.perf_counter() - t1) / steps
    b_alg = float(sum(20 * ns[i] + 24 * int(counts[i, 2]) for i in range(frames)))

    fps_in_flight, redone_second = two_in_flight(pwpp_hip, torch, h, whole, lambda hh: hh.make_device_batch(ptrs, ns), frames, 2 * steps, gpu_index)
    h.set_profiling(True)   # per-kernel times of the single-stream schedule on these frames
    h.reset_kernel_profile()
    for _ in range(3):
        h.launch_device_batch(whole, cols=4, mode=pwpp_hip.MODE_FRESH)
        h.synchronize()
    prof = h.kernel_profile()
    h.set_profiling(False)
    # CONTROL: the same kind of data REPLAYED -- the first six of these frames round-robin over `frames` distinct buffers, the way
    # the headline replays the six KITTI frames.  distinct / control isolates what replaying buys; control / headline is the data.
    rep_ns = [ns[i % 6] for i in range(frames)]
    rep_offs = np.concatenate([[0], np.cumsum(rep_ns)]).astype(np.int64)
    rep = torch.empty((int(rep_offs[-1]), 4), dtype=torch.float32, device=dev)
    for i in range(frames):
        rep[rep_offs[i]:rep_offs[i + 1]].copy_(big[offs[i % 6]:offs[i % 6 + 1]])
    torch.cuda.synchronize()
    hc = pwpp_hip.Handle(device=gpu_index)
    cb = hc.make_device_batch([rep.data_ptr() + int(rep_offs[i]) * 16 for i in range(frames)], rep_ns)
    for _ in range(4):
        hc.launch_device_batch(cb, cols=4, mode=pwpp_hip.MODE_FRESH)
        hc.synchronize()
    t1 = time.perf_counter()
    for _ in range(steps):
        hc.launch_device_batch(cb, cols=4, mode=pwpp_hip.MODE_FRESH)
        hc.synchronize()
    dtc = (time.perf_counter() - t1) / steps
    rep_ptrs = [rep.data_ptr() + int(rep_offs[i]) * 16 for i in range(frames)]
    control_in_flight, _ = two_in_flight(pwpp_hip, torch, hc, cb, lambda hh: hh.make_device_batch(rep_ptrs, rep_ns), frames, 2 * steps, gpu_index)
    control = {"frames_per_s": control_in_flight, "frames_per_s_synchronous": frames / dtc, "ms_per_step": 1000.0 * dtc, "workspace_gb": hc.workspace_bytes() / 1e9, "points_per_frame": int(np.mean(rep_ns)),
               "what": "six of the same synthetic frames replayed over %d distinct buffers (how the headline treats the six KITTI frames)" % frames}
    hc.close()
    del rep
    out = {"workload": "%d DISTINCT synthetic 64-beam frames (pwpp_synth.varied_frame(0..%d): %d-%d points, mean %d), device-resident in "
                       "distinct buffers (%.2f GB), fresh state per frame, cold handle" % (frames, frames - 1, min(ns), max(ns), int(np.mean(ns)), offs[-1] * 16 / 1e9),
           "frames": frames, "steps": steps, "frames_per_s": fps_in_flight, "ms_per_step": 1000.0 * frames / fps_in_flight,
           "schedule": "as the headline: two batches in flight, one handle each (a second, cold handle joins: its redone frames = %d)" % redone_second,
           "synchronous": {"frames_per_s": frames / dt, "ms_per_step": 1000.0 * dt},
           "redone_frames_steady": h.redo_stats()[1] - r0, "parity": parity,
           "frames_with_parts_moved_into_the_arena": {"all_calls_of_this_handle": h.arena_stats()[0], "frames_processed": h.redo_stats()[0]},
           "slots_per_point": h.arena_stats()[1] / float(np.mean(ns)),
           "first_batch": {"frames": half, "ms": ms_first, "redone_frames": redo_first, "what": "cold handle: allocations, 32-frame histogram probe, first launch"},
           "unseen_batch": {"frames": frames - half, "ms": ms_unseen, "redone_frames": redo_unseen,
                            "what": "frames this handle has never seen, segments sized from the first batch's counts"},
           "first_whole_batch": {"frames": frames, "ms": ms_whole, "redone_frames": redo_whole, "what": "first call of this size (workspace grows)"},
           "algorithmic_bytes_per_step": b_alg, "pipeline_frac": b_alg * fps_in_flight / frames / 1e9 / HBM_PEAK_GBS,
           "workspace_gb": h.workspace_bytes() / 1e9, "generated_in_s": gen_s,
           "replayed_control": control, "vs_replayed_control": fps_in_flight / control["frames_per_s"],
           "kernel_ms": {k: v[0] / max(v[1], 1) for k, v in prof.items()}}
    h.close()
    return out


def streams_leg(pwpp_hip, torch, dev, gpu_index, src_dev, ns_src, counts=(1, 64, 256, 1024), steps=30):
    """SURVEY 8f-f1 in the driver's line: S long-lived stateful streams stepped in lock-step -- the reference's
    real use (one PatchWorkpp object per sensor, demo_sequential.cpp:54-67), device-resident frames; stream s sees the source frames
    in the order s, s+1, ...  The single stream is also reported by its GPU time per frame in steady state (A-GLE histories full)."""
    out = {"what": "PWPP_MODE_STREAMS: S stateful streams in lock-step, one frame per stream and step; outside the timed region.  gpu_us_median = "
                   "HIP events on the handle's main stream, first kernel -> index lists written; with up to 64 streams the update of the streams' "
                   "adaptive thresholds (K5's second launch, option split_k5) runs on the handle's second stream under K6 and the host's turn-around "
                   "and is joined before the next call (tools/stream_latency.py: 108 -> 100 us for one stream)", "by_streams": []}
    K = len(src_dev)
    for S in counts:
        h = pwpp_hip.Handle(device=gpu_index)
        h.set_num_streams(S)
        batches = [h.make_device_batch([src_dev[(s + t) % K].data_ptr() for s in range(S)], [ns_src[(s + t) % K] for s in range(S)]) for t in range(K)]
        warm = 150 if S == 1 else 12   # (one stream: until the 1000-entry histories of the near rings are full)
        for t in range(warm):
            h.launch_device_batch(batches[t % K], cols=4, mode=pwpp_hip.MODE_STREAMS)
            h.synchronize()
        gpu = []
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n_steps = steps * (4 if S == 1 else 1)
        for t in range(n_steps):
            h.launch_device_batch(batches[t % K], cols=4, mode=pwpp_hip.MODE_STREAMS)
            h.synchronize()
            gpu.append(h.time_us())
        dt = (time.perf_counter() - t0) / n_steps
        row = {"streams": S, "ms_per_lockstep": 1000.0 * dt, "frames_per_s": S / dt, "gpu_us_median": sorted(gpu)[len(gpu) // 2]}
        if S == 1:
            row["history_entries"] = [int(len(h.history(0, 0, r))) for r in range(4)]
            row["gpu_us_min_max"] = [min(gpu), max(gpu)]
        out["by_streams"].append(row)
        h.close()
    # 1024 streams as TWO groups of 512 through the library's pipe in PWPP_MODE_STREAMS (round 6: pwpp_pipe_submit takes the mode;
    # handle g of the pipe owns group g's streams, submit k carries group k mod 2 -- a stream's frames stay in order on its handle,
    # and the lock-step of one group runs under the other's)
    S = 512
    pipe = pwpp_hip.Pipe(device=gpu_index, depth=2)
    pipe.set_num_streams(S)
    bs = [[pipe.handle(g).make_device_batch([src_dev[(g * S + s + t) % K].data_ptr() for s in range(S)], [ns_src[(g * S + s + t) % K] for s in range(S)])
           for t in range(K)] for g in range(2)]

    def run(n):
        for k in range(n):
            pipe.submit_device_batch(bs[k % 2][(k // 2) % K], cols=4, mode=pwpp_hip.MODE_STREAMS)
        pipe.drain()

    run(24)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 2 * steps
    run(n)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    lock = [r for r in out["by_streams"] if r["streams"] == 2 * S]
    out["two_groups_in_flight"] = {"streams": 2 * S, "groups": 2, "frames_per_s": S * n / dt, "ms_per_lockstep_of_a_group": 1000.0 * dt / n,
                                   "vs_lockstep_of_all": (S * n / dt) / lock[0]["frames_per_s"] if lock else None,
                                   "what": "pwpp_pipe_* in PWPP_MODE_STREAMS, depth 2: two groups of 512 streams, one pipe handle each, alternating -- "
                                           "against ONE handle stepping all 1024 streams in lock-step (by_streams)"}
    pipe.close()
    return out


def streams_workload(args, pwpp_hip, pwpp_dist, torch, dev, gpu_index, backend, world, rank):
    """`--workload streams` (SURVEY 8f-f1 at N >= 1): S = --frames long-lived stateful streams PER GPU, stepped in
    lock-step -- a step is one frame for each of the rank's streams (PWPP_MODE_STREAMS: adaptive thresholds, histories and sensor
    height carried from frame to frame per stream, as one PatchWorkpp object per sensor does).  Global stream g = rank + world * s
    lives on rank g mod world and sees the six source frames in the order g, g + 1, ...; its frames sit in buffers of their own
    (6 S distinct buffers per rank).  No data-path collective; the ranks agree on the slowest rank's time as in the headline."""
    src, data_name = load_source_frames("kitti")
    K, S = len(src), args.frames
    src_dev = [torch.from_numpy(a).to(dev) for a in src]
    ns_t, ptr_t, keep = [], [], []
    for t in range(K):
        pick = [((rank + world * s) + t) % K for s in range(S)]
        ns = [src[j].shape[0] for j in pick]
        offs = np.concatenate([[0], np.cumsum(ns)]).astype(np.int64)
        big = torch.empty((int(offs[-1]), 4), dtype=torch.float32, device=dev)
        for s in range(S):
            big[offs[s]:offs[s + 1]].copy_(src_dev[pick[s]])
        keep.append(big)
        ns_t.append(ns)
        ptr_t.append([big.data_ptr() + int(offs[s]) * 16 for s in range(S)])
    torch.cuda.synchronize()
    h = pwpp_hip.Handle(device=gpu_index)
    h.set_num_streams(S)
    batches = [h.make_device_batch(ptr_t[t], ns_t[t]) for t in range(K)]
    clock = [0]

    def step():
        t = clock[0] % K
        h.launch_device_batch(batches[t], cols=4, mode=pwpp_hip.MODE_STREAMS)
        h.synchronize()
        clock[0] += 1
        return t

    t = step()
    counts = h.all_counts()
    for s in range(S):
        assert counts[s, 0] + counts[s, 1] + counts[s, 5] == ns_t[t][s], "partition property violated in stream %d" % s
    for _ in range(max(args.warmup, 1)):
        step()
    pwpp_dist.barrier(dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    pwpp_dist.barrier(dev)
    my_elapsed = time.perf_counter() - t0
    elapsed, total = pwpp_dist.aggregate(my_elapsed, S * args.steps, dev if backend == "nccl" else None)
    per_gpu = pwpp_dist.gather_values(S * args.steps / my_elapsed, dev if backend == "nccl" else None)
    heights = pwpp_dist.gather_values(h.state(0).sensor_height, dev if backend == "nccl" else None)
    dist_info = pwpp_dist.describe(backend, dev)
    dist_info["launcher"] = "bench.py spawned its own ranks (torch.distributed.run)" if os.environ.get("PWPP_BENCH_SELF_SPAWNED") else \
        ("torch.distributed.run" if world > 1 else "single process")
    dist_info["workspace_gb_per_rank"] = [v / 1e9 for v in pwpp_dist.gather_values(float(h.workspace_bytes()), dev if backend == "nccl" else None)]
    if rank == 0:
        print(json.dumps({
            "metric": "frames/sec, 64-beam ~120k-pt cloud, estimateGround() hot path, STATEFUL streams (SURVEY 8f-f1)",
            "value": total / elapsed, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1000.0 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 points; f64 binning/thresholds; int64/int128 fixed-point plane-fit sums", "data": data_name,
            "config": {"workload": "f1: %d stateful streams per GPU in lock-step (one frame per stream and step), stream g on rank g mod %d, "
                                   "device-resident frames in %d distinct buffers per rank" % (S, world, K * S),
                       "streams_per_gpu": S, "parallelism": "streams sharded, dp%d" % world},
            "per_gpu": [{"rank": r, "frames_per_s": v} for r, v in enumerate(per_gpu)],
            "sensor_height_of_each_ranks_first_stream": heights, "dist": dist_info}))
    h.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--frames", type=int, default=1024, help="frames per batch per GPU")
    ap.add_argument("--workload", default="kitti", choices=["kitti", "dense", "streams"],
                    help="kitti: configs[2] (the headline); dense: configs[4]; streams: SURVEY 8f-f1 -- --frames stateful streams per GPU in lock-step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile-events", action="store_true")
    ap.add_argument("--no-overlap", action="store_true",
                    help="single-stream schedule (pwpp_set_overlap(0)); default is the library's default: batches of 128+ frames "
                         "as two frame ranges on two streams")
    ap.add_argument("--overlap", action="store_true", help="(default; kept for older command lines)")
    ap.add_argument("--profile-steps", type=int, default=7, help="steps of the separate single-stream pass that measures per-kernel times")
    ap.add_argument("--dense-frames", type=int, default=1024, help="frames of the configs[4] leg (outside the timed region, N = 1 only)")
    ap.add_argument("--dense-distinct", type=int, default=64, help="distinct clouds the configs[4] leg's buffers are filled from")
    ap.add_argument("--in-flight", type=int, default=2, help="batches kept enqueued in the timed region, one handle each (1 = synchronous steps on one handle)")
    ap.add_argument("--distinct-frames", type=int, default=1024, help="frames of the non-replayed leg (outside the timed region, N = 1 only; 0 = skip)")
    ap.add_argument("--skip-latency", action="store_true")
    ap.add_argument("--skip-extras", action="store_true", help="no parity_check / reference_order / ingest legs (all of them run outside the timed region)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        # Started as a plain process (`python bench.py --gpus N`, the way the driver starts --gpus 1): become the launcher --
        # N ranks of this script under torch.distributed.run on this node, one per GPU; rank 0's JSON line passes through.
        raise SystemExit(spawn_ranks(args.gpus))

    if os.environ.get("PWPP_BENCH_ECHO_RANK"):  # (tests/test_dist_cpu.py: which ranks were started, and by whom)
        print("bench rank %s of %s%s" % (os.environ.get("RANK", "0"), os.environ.get("WORLD_SIZE", "1"),
                                         " (self-spawned)" if os.environ.get("PWPP_BENCH_SELF_SPAWNED") else ""), file=sys.stderr, flush=True)
        if int(os.environ.get("WORLD_SIZE", "1")) > 1:  # every rank has said so before any goes on (without a GPU the first rank to
            import torch.distributed as dist             # fail makes the launcher stop the others -- possibly before their line)
            dist.init_process_group("gloo")
            dist.barrier()
            dist.destroy_process_group()

    import torch

    import pwpp_dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # Test hooks for the N > 1 code path where only one GPU can be leased (tests/test_gpu_parity.py::
    # test_bench_two_ranks_on_one_gpu): PWPP_BENCH_SHARE_DEVICE=1 puts every rank on GPU 0, PWPP_BENCH_BACKEND=gloo
    # replaces RCCL (two ranks cannot share a GPU under RCCL).  The driver's runs use neither.
    gpu_index = 0 if os.environ.get("PWPP_BENCH_SHARE_DEVICE") else local_rank
    backend = os.environ.get("PWPP_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(gpu_index)
    dev = torch.device("cuda", gpu_index)
    # Before any collective is set up: does this rank's GPU have the memory the workload will take?  An
    # out-of-memory inside rank k of 8, minutes into a run, is the dullest way for the first multi-GPU run to die.
    need_gb = estimate_memory_gb(args)
    free_b, total_b = torch.cuda.mem_get_info(gpu_index)
    if free_b / 1e9 < need_gb:
        raise SystemExit("bench.py: rank %s: GPU %d has %.1f GB free of %.1f GB, this workload needs about %.1f GB (inputs + %d workspaces): "
                         "free the device or lower --frames" % (os.environ.get("RANK", "0"), gpu_index, free_b / 1e9, total_b / 1e9, need_gb,
                                                              1 + max(1, min(4, args.in_flight))))
    world, rank, local_rank = pwpp_dist.init(backend, dev)  # "nccl" = RCCL over xGMI; no-op for a single process
    if args.gpus != world:
        raise SystemExit("bench.py --gpus %d was started as %d process(es): launch it with torch.distributed.run --nproc-per-node %d"
                         % (args.gpus, world, args.gpus))

    import pwpp_hip

    if args.workload == "streams":
        streams_workload(args, pwpp_hip, pwpp_dist, torch, dev, gpu_index, backend, world, rank)
        pwpp_dist.finalize()
        return

    src, data_name = load_source_frames(args.workload)
    F = args.frames
    # F distinct device buffers carved from one allocation; each frame starts 16-byte aligned
    which = pwpp_dist.shard_sources(len(src), F, rank)
    ns = [src[j].shape[0] for j in which]
    offs = np.concatenate([[0], np.cumsum(ns)]).astype(np.int64)
    big = torch.empty((int(offs[-1]), 4), dtype=torch.float32, device=dev)
    src_dev = [torch.from_numpy(s).to(dev) for s in src]
    for i in range(F):
        big[offs[i]:offs[i + 1]].copy_(src_dev[which[i]])
    torch.cuda.synchronize()
    ptrs = [big.data_ptr() + int(offs[i]) * 16 for i in range(F)]

    params = pwpp_hip.default_params()
    if args.workload == "dense":  # BASELINE.json configs[4]: 36-sector CZM
        for k in range(4):
            params.num_sectors_each_zone[k] = 36
    h = pwpp_hip.Handle(params, device=gpu_index)
    h.set_overlap(not args.no_overlap)
    batch = h.make_device_batch(ptrs, ns)

    def step():
        h.launch_device_batch(batch, cols=4, mode=pwpp_hip.MODE_FRESH)
        h.synchronize()

    held = {}  # which of the D batches each pipe handle processed last

    def run_steps(n):
        """n steps with up to D batches in flight; returns with every batch landed."""
        if D == 1:
            for _ in range(n):
                step()
            return
        for k in range(n):
            holder = pipe.submit_device_batch(batches[k % D], cols=4)  # (waits for the batch this handle launched D submits ago)
            held[holder._h.value] = k % D
        pipe.drain()

    step()  # (one step for the checks below; the W warm-up steps run right in front of the timed region, after the host-side checks)
    # self-check: replays of the same source frame must produce identical counts, and
    # ground + non-ground must partition the frame (no reference data needed on the GPU box)
    counts = h.all_counts()
    n_patches = counts[:, 2]
    for i in range(F if not os.environ.get("PWPP_BENCH_NO_SELFCHECK") else 0):  # (perf-only experiments with deliberately wrong kernels)
        assert counts[i, 0] + counts[i, 1] + counts[i, 5] == ns[i], "partition property violated in frame %d" % i
    by_src = {}
    for i in range(F):
        by_src.setdefault(which[i], set()).add(tuple(int(v) for v in counts[i, :3]))
    assert os.environ.get("PWPP_BENCH_NO_SELFCHECK") or all(len(v) == 1 for v in by_src.values()), "replayed frames disagree: %r" % by_src

    # Batches IN FLIGHT (round 5): the timed region keeps D = --in-flight batches enqueued -- D handles, each with its own input
    # buffers, workspace and stream, every one a full batch of F frames; step k goes to handle k mod D, which first waits for the
    # batch it launched D steps earlier (its results are complete and readable until then).  The ramp-up of one batch (binning,
    # nothing to overlap with) then runs under the ramp-down of the one before (last fits, index lists): 2.46 against 2.63 ms per
    # batch for the synchronous step with the in-handle overlap schedule (profiles/r05_pipelined_batches.txt).  Each handle runs
    # the plain single-stream schedule (the in-handle overlap schedule on top is slower: 2.83 ms).  D = 1: rounds 1-4's step.
    D = max(1, min(4, args.in_flight)) if F >= 128 and not args.no_overlap else 1
    batches, inputs = [batch], [big]
    for d in range(1, D):
        which_d = pwpp_dist.shard_sources(len(src), F, rank + d)  # (the same frames in another rotation: other buffers, other addresses)
        ns_d = [src[j].shape[0] for j in which_d]
        offs_d = np.concatenate([[0], np.cumsum(ns_d)]).astype(np.int64)
        big_d = torch.empty((int(offs_d[-1]), 4), dtype=torch.float32, device=dev)
        for i in range(F):
            big_d[offs_d[i]:offs_d[i + 1]].copy_(src_dev[which_d[i]])
        inputs.append(big_d)
        batches.append(h.make_device_batch([big_d.data_ptr() + int(offs_d[i]) * 16 for i in range(F)], ns_d))
    # the library's own pipe (pwpp_pipe_*: D handles, single-stream schedule each); H = [the first handle of the legs, the pipe's handles]
    pipe = pwpp_hip.Pipe(params, device=gpu_index, depth=D) if D > 1 else None
    H = [h] + ([pipe.handle(i) for i in range(D)] if pipe else [])
    torch.cuda.synchronize()
    selfcheck = not os.environ.get("PWPP_BENCH_NO_SELFCHECK")
    # oracle anchor (outside the timed region): the ground masks and plane normals of batch frames 0-5 and
    # of one frame of the second frame range against the committed goldens, which tests/golden/make_golden.py generated
    # from the reference's own patchworkpp.cpp (oracle/_ref, eigen-f32 build).  Nothing under oracle/ is touched here.
    parity = None
    gpath = os.path.join(ROOT, "tests", "golden", "kitti_golden.npz")
    if args.workload == "kitti" and data_name.startswith("kitti") and os.path.exists(gpath) and not args.skip_extras:
        gold = np.load(gpath)
        picks = sorted(set(i for i in list(range(6)) + [F - 1 - (F // 4)] if 0 <= i < F))
        iou_min, dn_max = 1.0, 0.0
        for i in picks:
            k = which[i]
            mask = np.zeros(ns[i], np.uint8)
            mask[h.ground_indices(i)] = 1
            want = np.unpackbits(gold["f32/fresh/%d/ground_mask" % k])[:ns[i]]
            inter, union = int((mask & want).sum()), int((mask | want).sum())
            iou_min = min(iou_min, inter / union if union else 1.0)
            dn_max = max(dn_max, float(np.abs(h.normals(i) - gold["f32/fresh/%d/normals" % k]).max()))
        parity = {"frames": len(picks), "batch_frames": picks, "iou": iou_min, "max_dnormal": dn_max,
                  "scope": "%d of the %d batch frames (the six distinct sources + one frame of the second frame range): ground masks and "
                           "plane normals only; the bit-for-bit check of all %d frames (index lists, planes, state) is "
                           "tests/test_gpu_parity.py::test_full_size_batch_properties in the -m gpu suite" % (len(picks), F, F),
                  "against": "tests/golden/kitti_golden.npz = the reference's patchworkpp.cpp (oracle/_ref, float sums), fresh state per frame"}
        assert not selfcheck or (iou_min == 1.0 and dn_max < 1e-4), "parity anchor failed: %r" % parity

    # The timed region runs the library's default schedule (overlap mode for batches of 128+ frames) with no
    # profiling events in it; the per-kernel times and the roofline line come from a SEPARATE single-stream pass
    # after the timed region (HIP events around every launch would serialise the two frame ranges).
    run_steps(max(args.warmup, 1))  # W untimed warm-up steps (the checks above leave the GPU idle for a while: warm up AFTER them)
    pwpp_dist.barrier(dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run_steps(args.steps)
    torch.cuda.synchronize()
    pwpp_dist.barrier(dev)
    elapsed = time.perf_counter() - t0
    my_elapsed = elapsed
    elapsed, total_frames = pwpp_dist.aggregate(elapsed, F * args.steps, dev if backend == "nccl" else None)  # MAX time, SUM frames over ranks
    if D > 1:  # every handle's last batch is complete and equal to the first handle's first one, frame for frame of the same source
        for d in range(D):
            if H[1 + d]._h.value not in held:
                continue
            cd, wd = H[1 + d].all_counts(), pwpp_dist.shard_sources(len(src), F, rank + held[H[1 + d]._h.value])
            first_of = {}
            for i in range(F):
                first_of.setdefault(which[i], i)
            for i in range(0, F, 37):
                assert not selfcheck or tuple(cd[i, :3]) == tuple(counts[first_of[wd[i]], :3]), "handle %d frame %d differs from handle 0" % (d, i)
    # the synchronous step of rounds 1-4 (launch, wait; the library's in-handle overlap schedule), for comparison across rounds
    sync_leg = None
    if D > 1:
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(10):
            step()
        torch.cuda.synchronize()
        dts = (time.perf_counter() - t1) / 10
        sync_leg = {"ms_per_step": 1000.0 * dts, "frames_per_s": F / dts, "steps": 10,
                    "what": "ONE batch at a time on one handle (launch, wait), library default schedule (two frame ranges over three streams): "
                            "the step rounds 1-4 reported as `value`; also the latency of one 1024-frame batch"}
    # The same schedule with the plane-fit sums on rounds 3-5's 2^-21 m grid (option exact_moments = 0): what contract v4 -- sums exact
    # on the reference's own floats, the default -- costs.  Outside the timed region.
    exact_off = None
    if D > 1 and not args.skip_extras:
        p0 = pwpp_hip.Pipe(params, device=gpu_index, depth=D)
        for d in range(D):
            p0.handle(d).set_option("exact_moments", "0")
        for k in range(6):
            p0.submit_device_batch(batches[k % D], cols=4)
        p0.drain()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        n0 = 40
        for k in range(n0):
            p0.submit_device_batch(batches[k % D], cols=4)
        p0.drain()
        torch.cuda.synchronize()
        dt0 = (time.perf_counter() - t1) / n0
        exact_off = {"ms_per_step": 1000.0 * dt0, "frames_per_s": F / dt0, "steps": n0,
                     "what": "pwpp_set_option(exact_moments, 0): integer moments on a 2^-21 m grid (contract v3 of rounds 3-5) -- faster, and off the "
                             "reference by 1-31 indices on 0.2 % of varied frames (profiles/r06_parity_statistics_10k.json); the headline runs the default, exact_moments = 1"}
        p0.close()
    prof, prof_all = {}, {}
    if not args.no_profile_events:  # kernel times of the single-stream schedule, outside the timed region
        # Two untimed steps first, then the MEDIAN of the steps' own values per kernel: the first launches of a kernel with scratch
        # memory on a stream that has not run it yet can take several milliseconds (queue-side scratch set-up) -- twice in five
        # runs of this round the small-patch fit kernel read 1.3-1.6 instead of 0.7 ms as a mean over five steps and became the
        # "dominant kernel" of the roofline.
        h.set_profiling(True)
        for _ in range(2):
            step()
        per_step = {}
        for _ in range(max(args.profile_steps, 1)):
            h.reset_kernel_profile()
            step()
            for name, (ms, launches) in h.kernel_profile().items():
                per_step.setdefault(name, []).append(ms / max(launches, 1))
        for name, vals in per_step.items():
            vals = sorted(vals)
            prof[name] = (vals[len(vals) // 2], 1)
            prof_all[name] = vals
    h.set_profiling(False)

    # single-frame latency (configs[1]): EVERY distinct source frame on its own, device-resident, fresh state (VERDICT r04 item 4:
    # frame 0 alone was the best case) -- median of 30 calls per source, GPU time between HIP events on the library's stream
    lat_rows = []
    seen_src = []
    for i in range(F):
        if which[i] not in seen_src:
            seen_src.append(which[i])
            one = h.make_device_batch(ptrs[i:i + 1], ns[i:i + 1])
            lat, lat_gpu = [], []
            for _ in range(0 if args.skip_latency else 35):
                t1 = time.perf_counter()
                h.launch_device_batch(one, cols=4, mode=pwpp_hip.MODE_FRESH)
                h.synchronize()
                lat.append(time.perf_counter() - t1)
                lat_gpu.append(h.time_us())
            if lat:
                lat, lat_gpu = sorted(lat[5:]), sorted(lat_gpu[5:])
                lat_rows.append({"source": int(which[i]), "points": int(ns[i]), "gpu_us": lat_gpu[len(lat_gpu) // 2], "wall_ms": 1000.0 * lat[len(lat) // 2]})
        if len(seen_src) == len(src):
            break
    lat_rows.sort(key=lambda r: r["source"])

    per_gpu = pwpp_dist.gather_values(F * args.steps / my_elapsed, dev if backend == "nccl" else None)  # every rank's own frames/s
    per_gpu_redone = pwpp_dist.gather_values(float(sum(hh.redo_stats()[1] for hh in H)), dev if backend == "nccl" else None)
    per_gpu_ws = pwpp_dist.gather_values(float(sum(hh.workspace_bytes() for hh in H)) / 1e9, dev if backend == "nccl" else None)
    dist_info = pwpp_dist.describe(backend, dev)  # backend, world size, RCCL version, every rank's GPU (all-gather)
    dist_info["launcher"] = "bench.py spawned its own ranks (torch.distributed.run)" if os.environ.get("PWPP_BENCH_SELF_SPAWNED") else \
        ("torch.distributed.run" if world > 1 else "single process")
    dist_info["workspace_gb_per_rank"] = [v / 1e9 for v in pwpp_dist.gather_values(float(sum(hh.workspace_bytes() for hh in H)), dev if backend == "nccl" else None)]

    # reference-order output mode: the same batch with every sub-list in the reference's z-sorted order
    ref_order = None
    if not args.skip_extras:
        h.set_output_order(True)
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        ref_order = {"ms_per_step": 1000.0 * (time.perf_counter() - t1) / 5, "steps": 5,
                     "what": "pwpp_set_output_order(PWPP_ORDER_REFERENCE): k_order_sublists after k_emit, one stream"}
        h.set_output_order(False)

    # ingest (PCIe-inclusive, never `value`): chunks of 256 frames from page-locked host slabs, two handles
    # double-buffered (H2D + pipeline of one chunk under the D2H of the other's index lists), rank 0 at N = 1 only
    ingest = None
    if world == 1 and args.workload == "kitti" and not args.skip_extras:
        try:
            ingest = ingest_leg(pwpp_hip, src, gpu_index)
        except Exception as e:
            ingest = {"frames_per_s": None, "error": str(e)}

    dense = None
    if world == 1 and args.workload == "kitti" and not args.skip_extras:
        try:
            dense = dense_leg(pwpp_hip, torch, dev, gpu_index, frames=args.dense_frames, distinct=args.dense_distinct)
        except Exception as e:
            dense = {"frames_per_s": None, "error": str(e)}

    distinct = None
    if world == 1 and args.workload == "kitti" and not args.skip_extras and args.distinct_frames > 0:
        try:
            distinct = distinct_leg(pwpp_hip, torch, dev, gpu_index, frames=args.distinct_frames)
        except Exception as e:
            distinct = {"frames_per_s": None, "error": str(e)}

    streams = None
    if world == 1 and args.workload == "kitti" and not args.skip_extras:
        try:
            streams = streams_leg(pwpp_hip, torch, dev, gpu_index, src_dev, [a.shape[0] for a in src])
        except Exception as e:
            streams = {"by_streams": [], "error": str(e)}

    if rank == 0:
        fps = total_frames / elapsed
        b_alg = float(sum(20 * ns[i] + 24 * int(n_patches[i]) for i in range(F)))  # bytes per batch (one GPU)
        out = {
            "metric": "frames/sec, 64-beam ~120k-pt cloud, estimateGround() hot path (ground-idx IoU==1.0 vs CPU ref enforced by tests)",
            "value": fps, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1000.0 * elapsed / args.steps, "ms_per_frame": 1000.0 * elapsed / (args.steps * F),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 points; f64 binning/thresholds; int64/int128 fixed-point plane-fit sums",
            "data": data_name,
            "config": {"workload": "configs[2]: batch of %d replayed 64-beam frames per GPU, device-resident in %d distinct "
                                   "buffers (%.2f GB), fresh state per frame, default 4-zone CZM"
                                   % (F, F, offs[-1] * 16 / 1e9) if args.workload == "kitti" else
                                   "configs[4]: %d dense synthetic 128-beam ~500k-pt frames per GPU, 36-sector CZM" % F,
                       "frames_per_gpu": F, "points_per_frame": int(np.mean(ns)), "parallelism": "frames sharded, dp%d" % world,
                       "batches_in_flight": D,
                       "schedule": ("one stream" if args.no_overlap or F < 128 else
                                    ("%d batches in flight through the library's pipe (pwpp_pipe_*: one handle, workspace and stream per batch in flight; every step = one "
                                     "whole batch of %d frames; a handle's results stay readable until its next launch); each handle: one stream" % (D, F) if D > 1 else
                                     "library default: two frame ranges, binning and lists on the main stream, each range's plane fits on its own"))
                                   + "; kernel_ms / roofline.kernel_ms: median over a separate single-stream pass of %d steps (after 2 untimed ones) outside the timed region" % args.profile_steps},
            "binning": {"one_pass_batches": h.one_pass_stats()[0], "redone_two_pass": h.one_pass_stats()[1],
                        "one_pass_frames": h.redo_stats()[0], "redone_frames": h.redo_stats()[1],
                        "frames_with_parts_moved_into_the_arena": sum(hh.arena_stats()[0] for hh in H),
                        "slots_per_point": H[-1].arena_stats()[1] / float(np.mean(ns)), "arena_slots_per_frame": H[-1].arena_stats()[2],
                        "workspace_gb": h.workspace_bytes() / 1e9, "input_gb": float(offs[-1]) * 16 / 1e9,
                        "workspace_gb_all_handles": sum(hh.workspace_bytes() for hh in H) / 1e9},
        }
        if lat_rows:
            g = sorted(r["gpu_us"] for r in lat_rows)
            wl = sorted(r["wall_ms"] for r in lat_rows)
            out["latency"] = {"workload": "configs[1]: single frame, device-resident, fresh state -- each of the %d distinct source frames on its own" % len(lat_rows),
                              "gpu_us": g[len(g) // 2], "gpu_us_min": g[0], "gpu_us_median": g[len(g) // 2], "gpu_us_max": g[-1],
                              "ms_per_frame_wall": wl[len(wl) // 2], "ms_per_frame_wall_max": wl[-1], "by_source": lat_rows,
                              "what": "median of 30 calls per source frame; gpu_us = between HIP events on the library's stream, first kernel to last"}
        if sync_leg is not None:
            out["synchronous"] = sync_leg
        out["per_gpu"] = [{"rank": r, "frames_per_s": v, "redone_frames": int(per_gpu_redone[r]), "workspace_gb": per_gpu_ws[r]} for r, v in enumerate(per_gpu)]
        out["dist"] = dist_info
        out["selfcheck"] = bool(selfcheck)
        if exact_off is not None:
            out["exact_moments_off"] = exact_off
        if parity is not None:
            out["parity_check"] = parity
        if ref_order is not None:
            out["reference_order"] = ref_order
        if ingest is not None:
            out["ingest"] = ingest
        if dense is not None:
            out["dense"] = dense
        if distinct is not None:
            out["distinct"] = distinct
            if distinct.get("frames_per_s"):
                distinct["vs_value"] = distinct["frames_per_s"] / (fps / world)
        if streams is not None:
            out["streams"] = streams
        if prof:
            dom = max(prof, key=lambda k: prof[k][0])
            dom_ms = prof[dom][0] / max(prof[dom][1], 1)
            traffic = None
            tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
            if os.path.exists(tpath):
                try:
                    traffic = json.load(open(tpath)).get(dom)
                except Exception:
                    traffic = None
            ach = b_alg / (dom_ms * 1e-3) / 1e9
            out["roofline"] = {"bound": "hbm", "kernel": dom, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
                               "traffic_source": "profiles/hbm_traffic.json (its own `source` key names the script and round that wrote it): "
                                                 "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over the same workload, per launch over the "
                                                 "whole batch, 2 x FETCH_SIZE + WRITE_SIZE; a constant read from a tracked file, not measured inside this run",
                               "algorithmic_bytes_per_launch": b_alg, "kernel_ms": dom_ms,
                               "pipeline_achieved": b_alg * args.steps / elapsed / 1e9,
                               "pipeline_frac": b_alg * args.steps / elapsed / 1e9 / HBM_PEAK_GBS,
                               # fps x B_alg against the 8 TB/s peak (above) and against what a read
                               # stream reaches on this chip (6.29 TB/s in MI355X_MICROARCH.md; 6.3-6.45 measured: profiles/r03_read_bw.txt)
                               "pipeline_frac_of_achievable": b_alg * args.steps / elapsed / 1e9 / 6290.0}
            out["kernel_ms"] = {k: v[0] / max(v[1], 1) for k, v in prof.items()}
            out["kernel_ms_min_max"] = {k: [v[0], v[-1]] for k, v in prof_all.items() if v and v[-1] > 0.02}
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(src[:6])
            except Exception as e:  # the checker libs are optional at bench time
                out["cpu_baseline"] = {"value": None, "unit": "frames/s", "cores": 0, "kind": "port", "sample": "unavailable: %s" % e}
        print(json.dumps(out))
    pwpp_dist.finalize()


if __name__ == "__main__":
    main()
