#!/usr/bin/env python3
"""Benchmark of the path tracer's hot path (one Scene::Pathtrace frame) on N MI355X GPUs.

A step is one progressive frame: `spp` new samples for every pixel of the image, traced and shaded
by the wavefront HIP kernels and added to the accumulation buffer.  The default workload is the
reference's Cornell box (area light, Lambertian global illumination) at 1024x1024 with the
reference's 10 samples per pixel per frame, SAH BVH traversal.

With N > 1 (one process per GPU, launched by torch.distributed.run) every rank renders its own
`spp` samples of a frame that adds N*spp samples per pixel (sample split, weak scaling) and the
per-frame accumulation buffers are summed onto rank 0 with one RCCL reduce.

Prints one JSON line on rank 0: metric / value (closest-hit queries per second, whole job), spp_per_s,
ms_per_step, `roofline` for the dominant kernel, per-stage milliseconds, a short kd-tree run, a `sponza`
sub-record (the large-tree regime: 210 k-triangle stand-in at 3840x2160, both accelerators) and `cpu_baseline`
(the CPU oracle timed on this host, rank 0 at N = 1 only).

Which ceiling `roofline` is quoted against follows from where the walk's operands come from.  A scene whose
whole tree is staged in LDS (PT_OPT_TREE_IN_LDS: the Cornell boxes) has no global-memory path in the walk:
its node and triangle operands cost instruction issue, not bandwidth, so the bound is the FP32 vector peak and
`achieved` = algorithmic FLOPs (27 per slab test + 75 per triangle test, both counted by the kernel itself in
an untimed pass) per launch / the launch time measured live with HIP events.  For a tree that leaves the CU the
same FP32 fraction is reported beside the measured byte rates at L2 and HBM, and `bound` names the ceiling the
kernel is closest to.  HBM bytes per launch (`traffic`) come from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE
passes run by this program on the same workload before it opens the GPU itself (`traffic_source` says so), or,
when the profiler cannot be run, from the committed passes under profiles/.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E, spec
# ceilings for random 64-byte fetches, measured with tools/node_fetch_probe.hip (profiles/r02_node_fetch_probe.txt)
L1_PEAK_ACCESSES_PER_US = 820000.0   # vector L1 tag accesses, chip-wide
L2_PEAK_REQUESTS_PER_US = 130000.0   # L2 requests, chip-wide (= 8.3 TB/s of 64-byte nodes)
L2_PEAK_GBS = 34500.0       # aggregate over the eight XCDs' L2 caches, measured figure of the hardware guide
FP32_PEAK_TFLOPS = 157.3    # 256 CUs x 4 SIMDs x 64 FMA lanes per 2 cycles x 2 FLOP x 2.4 GHz
FLOP_PER_BOX_TEST = 27.0    # 6 subtractions + 6 divisions + 3 swaps + 4 min/max + 2 sign tests + entry compare, as counted for the reference's slab test
FLOP_PER_TRI_TEST = 75.0    # two cross products, four dot products, three divisions, the barycentric and range tests

WORKLOADS = {
    # name: (obj, needs probe)
    "cornellbox": ("cornellbox.obj", False),
    "cornellbox-glass": ("cornellbox-glass.obj", True),
    "cornellbox-metal": ("cornellbox-metal.obj", True),
    "cornellbox-alpha": ("cornellbox-alpha.obj", True),
    "cornellball-glass": ("cornellball-glass.obj", False),
    "teapot": ("teapot.obj", True),
    "teapot-glass": ("teapot-glass.obj", True),
    "testobj": ("testObj.obj", True),
    "bunny": ("bunny.obj", True),
    # procedural stand-in for the undistributed sponza.obj (tools/make_sponza.py), probe-lit
    "sponza": (None, True),
}
# per-workload defaults (width, height, spp per frame and GPU); the command line overrides them
DEFAULT_SHAPE = {"sponza": (3840, 2160, 4)}


def sponza_path():
    """Generates the stand-in courtyard once (seeded, ~210 k triangles) and returns its .obj path."""
    from make_sponza import build, write
    folder = os.path.join(ROOT, "gpurun_out", "sponza")
    path = os.path.join(folder, "sponza_standin.obj")
    if not os.path.exists(path):
        tmp = os.path.join(ROOT, "gpurun_out", "sponza.%d.tmp" % os.getpid())
        os.makedirs(tmp, exist_ok=True)
        write(build(seed=7, detail=1.0), os.path.join(tmp, "sponza_standin.obj"))
        try:
            os.replace(tmp, folder)
        except OSError:  # another rank was first
            import shutil
            shutil.rmtree(tmp, ignore_errors=True)
    return path


def workload_obj(args):
    from common import scene_path
    if args.obj:
        return args.obj
    obj = WORKLOADS[args.workload][0]
    return sponza_path() if obj is None else scene_path(obj)


def workload_camera(args):
    if args.workload == "sponza" and not args.obj:
        from make_sponza import CAMERA
        return CAMERA
    return None


def is_default_workload(args, world):
    return (args.workload == "cornellbox" and not args.obj and args.accel == "bvh" and
            (args.width, args.height, args.spp) == (1024, 1024, 10) and world == 1)


def committed_counters(args, world):
    """Counter figures of trace_kernel from the committed rocprofv3 --pmc passes (profiles/r03_pmc_*.json, else r02:
    tools/gpu_pmc_all.sh on the same workload, one frame in flight).  Only attached to the workloads those
    passes ran; None otherwise."""
    tag = None
    if is_default_workload(args, world):
        tag = "cornellbox_bvh"
    elif args.workload == "sponza" and not args.obj and (args.width, args.height) == (3840, 2160) and args.spp >= 2:
        # (a batch holds two samples per pixel at this size whatever the frame's sample count, and every rank of a
        # sample split runs the same batches, so the per-launch rates of the committed passes apply)
        tag = "sponza_%s" % args.accel
    if tag is None:
        return None
    out = {}
    for kind in ("valu", "cache"):
        for rnd in ("r03", "r02"):  # the newest passes that were committed
            name = "profiles/%s_pmc_%s_%s.json" % (rnd, kind, tag)
            path = os.path.join(ROOT, name)
            if os.path.exists(path):
                k = json.load(open(path))["kernels"].get("trace_kernel")
                if k:
                    out[kind] = k
                    out[kind + "_source"] = name
                    break
    return out or None


def live_hbm_traffic(args, world):
    """HBM bytes per trace_kernel launch measured now: two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; a pass
    cannot hold both) over three frames of the default workload in child processes, run BEFORE this process opens
    the GPU.  Returns (bytes per launch, launches seen) or None when the profiler is absent, this program is itself
    being profiled, or a pass fails; never raises."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if not is_default_workload(args, world) or args.no_pmc or shutil.which("rocprofv3") is None:
        return None
    if any(k.startswith(("ROCPROFILER_", "ROCPROF_", "ROCP_")) for k in os.environ):
        return None  # already under a profiler: a nested one would fight over the counters
    tmp = tempfile.mkdtemp(prefix="pt_bench_pmc_")
    env = dict(os.environ, TMPDIR=tmp)
    tot = {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, counter)
            cmd = ["timeout", "-s", "KILL", "90", "rocprofv3", "--pmc", counter, "--output-format", "csv", "-d", d, "--",
                   sys.executable, os.path.join(ROOT, "tools", "prof_frame.py"), "cornellbox.obj", "1024", "3", "bvh"]
            r = subprocess.run(cmd, cwd=tmp, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=120)
            if r.returncode != 0:
                return None
            kib, ids = 0.0, set()
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if row["Counter_Name"] == counter and "trace_kernel" in row["Kernel_Name"]:
                        kib += float(row["Counter_Value"])
                        ids.add(row["Dispatch_Id"])
            if not ids:
                return None
            tot[counter] = (kib * 1024.0, len(ids))
        # gfx950 tallies the 128-byte requests of a wide coalesced read as 64 bytes: the read side is doubled
        launches = tot["FETCH_SIZE"][1]
        return (tot["FETCH_SIZE"][0] * 2.0 / launches + tot["WRITE_SIZE"][0] / tot["WRITE_SIZE"][1], launches)
    except Exception:
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def bytes_per_ray(n_box, n_tri, hit_frac, textured=False):
    """Algorithmic bytes one closest-hit query moves: 24 B per slab test (6 floats), 36 B per
    triangle test (9 floats), 48 B of normals + material id for a winning hit, 160 B of wavefront
    state (ray 32 B + hit record 16 B + path state 32 B, each written once and read once)."""
    return 24.0 * n_box + 36.0 * n_tri + 48.0 * hit_frac + 160.0


def make_scene(pta, args, device, accel):
    needs_probe = WORKLOADS[args.workload][1]
    s = pta.Scene(device)
    path = workload_obj(args)
    if not s.load(path):
        raise SystemExit("cannot load %s" % path)
    if args.obj:
        if args.hdr:
            s.load_env(args.hdr)
    elif needs_probe:
        s.load_env(probe_path(args))
    cam = workload_camera(args)
    if cam:
        s.set_camera(**cam)
    s.pre_calc(accel)
    if args.lds_nodes is not None:
        s.set_option(pta.OPT_LDS_NODES, args.lds_nodes)
    if args.traversal is not None:
        s.set_option(pta.OPT_TRAVERSAL, args.traversal)
    if args.tail_level is not None:
        s.set_option(pta.OPT_TAIL_LEVEL, args.tail_level)
    if args.blocks_per_cu is not None:
        s.set_option(pta.OPT_BLOCKS_PER_CU, args.blocks_per_cu)
    if getattr(args, "max_batch", None) is not None:
        s.set_option(pta.OPT_MAX_BATCH, args.max_batch)
    if args.streams is not None:
        s.set_option(pta.OPT_STREAMS, args.streams)
    s.set_resolution(args.width, args.height)
    return s


def probe_path(args):
    """512x512 synthetic probe (the reference's uffizi probe is not distributed), written once."""
    from make_probe import probe_pixels, write_hdr
    path = os.path.join(ROOT, "gpurun_out", "bench_probe_512.hdr")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    if not os.path.exists(path):
        tmp = path + ".%d.tmp" % os.getpid()
        write_hdr(tmp, probe_pixels(512, 1))
        os.replace(tmp, path)
    return path


def cpu_baseline(args, budget_ref=10.0, frames_ref=5, budget_core=6.0, frames_core=3, most_samples=12_000_000):
    """The CPU restatement of the reference's multithreaded loop (column strips, one thread per strip), timed on
    this host on a bounded sample of the same workload: once with one thread per core this process may run on, and
    once under the reference's own rule of 3 x hardware_concurrency threads (src/cs488.h:98)."""
    import orc
    from PIL import Image
    from common import read_hdr_independent
    needs_probe = WORKLOADS[args.workload][1]
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count() or 1
    # a GPU box hands a job a share of its CPUs through OMP_NUM_THREADS-style limits rather than through the
    # affinity mask; never start more worker threads than that share
    share = int(os.environ.get("OMP_NUM_THREADS", "0") or 0)
    cores = max(1, min(usable, share) if share > 0 else min(usable, 16))
    o = orc.Oracle()
    if o.load_obj(workload_obj(args)) <= 0:
        raise SystemExit("oracle cannot load the workload")
    for i, m in enumerate(o.materials()):
        if m.textured:
            o.set_texture(i, np.asarray(Image.open(m.tex_file.decode()).convert("RGB"), np.uint8))
    if needs_probe and not args.obj:
        o.set_env(read_hdr_independent(probe_path(args)))
    cam = workload_camera(args)
    if cam:
        o.set_camera(**cam)
    o.build(orc.BVH if args.accel == "bvh" else orc.KD)
    # bounded sample: the same view and spp on an image shrunk until one frame is <= 12 M samples
    w, h, shrink = args.width, args.height, 1
    while (w // shrink) * (h // shrink) * args.spp > most_samples:
        shrink *= 2
    w, h = w // shrink, h // shrink

    def timed(threads, budget_s, most_frames):
        o.reset(w, h)
        o.render(spp=1, frames=1, threads=threads)  # warm-up
        o.reset(w, h)
        frames, t0, rays = 0, time.time(), 0
        while True:
            _, _, cnt = o.render(spp=args.spp, frames=1, threads=threads)
            rays += int(cnt[0])
            frames += 1
            dt = time.time() - t0
            if dt > budget_s or frames >= most_frames:
                return rays / dt * 1e-6, frames, dt

    hw = os.cpu_count() or 1
    # the reference's own rule first: numThreads = 3 x hardware_concurrency (src/cs488.h:98); a strip narrower than one
    # column holds no pixel (src/cs488.h:1885-1890), so no more threads than columns
    ref_threads = min(3 * hw, w)
    mrays, frames, dt = timed(ref_threads, budget_ref, frames_ref)
    core_mrays, core_frames, _ = timed(cores, budget_core, frames_core) if frames_core > 0 else (None, 0, 0.0)
    return {
        "s_per_frame": dt / frames * float(args.width * args.height) / float(w * h),
        "value": mrays, "unit": "Mrays/s", "cores": cores, "threads": ref_threads, "kind": "port",
        "spp_per_s": frames * args.spp / dt * (w * h) / float(args.width * args.height),
        "host": {"hardware_concurrency": hw, "cpus_this_process_may_use": usable, "thread_share": share or None},
        "one_thread_per_core": {"threads": cores, "value": core_mrays, "frames": core_frames} if core_frames else None,
        "note": "threads = 3 x hardware_concurrency as the reference sets numThreads (src/cs488.h:98), over column strips; the oracle gives every "
                "strip its own PCG32 stream where the reference's threads share one state.  `cores` is the number of CPUs this process may use.",
        "sample": "%d frame(s) of the same view at %dx%d (1/%d of the workload's resolution per axis) x %d spp; spp_per_s is scaled to the "
                  "workload's pixel count" % (frames, w, h, shrink, args.spp),
    }


def fp32_tflops(n_box, n_tri, rays, seconds):
    return (FLOP_PER_BOX_TEST * n_box + FLOP_PER_TRI_TEST * n_tri) * rays / seconds * 1e-12 if seconds > 0 else 0.0


def sponza_record(pta, args, device, stream_handle):
    """The large-tree regime beside the default workload: the 210 k-triangle stand-in for sponza.obj (which the
    reference does not distribute) at 3840x2160, 4 spp per frame, both accelerators.  For each: throughput over
    a few frames, slab and triangle tests per ray from a counted pass, the FP32 fraction of trace_kernel from
    live events (one frame in flight), and the committed cache-counter figures of the same workload."""
    import copy
    out = {"workload": "sponza stand-in (tools/make_sponza.py, seed 7, detail 1.0) 3840x2160, 4 spp/frame, synthetic probe"}
    sub = copy.copy(args)
    sub.workload, sub.obj, sub.hdr = "sponza", None, None
    sub.width, sub.height, sub.spp = DEFAULT_SHAPE["sponza"]
    sub.lds_nodes = sub.traversal = sub.tail_level = sub.blocks_per_cu = sub.streams = sub.max_batch = None
    for name, accel in (("bvh", pta.BVH), ("kd", pta.KD)):
        sub.accel = name
        s = make_scene(pta, sub, device, accel)
        s.set_stream(stream_handle)
        s.set_option(pta.OPT_COUNT_TESTS, 1)
        s.reset_counters()
        s.pathtrace(sub.spp)
        s.synchronize()
        c = s.counters()
        n_box = c[pta.CNT_BOX_TESTS] / max(c[pta.CNT_RAYS], 1)
        n_tri = c[pta.CNT_TRI_TESTS] / max(c[pta.CNT_RAYS], 1)
        s.set_option(pta.OPT_COUNT_TESTS, 0)
        # trace_kernel alone: one frame in flight, events around its launches only
        # (the same kernel as with several frames in flight: traversal bit 3 keeps the workgroups on their own regions,
        # where a frame rendered alone would otherwise draw from the shared pools)
        lanes = s.get_option(pta.OPT_STREAMS)
        form = s.get_option(pta.OPT_TRAVERSAL)
        s.set_option(pta.OPT_STREAMS, 1)
        s.set_option(pta.OPT_TRAVERSAL, form | 8)
        s.pathtrace(sub.spp)
        s.synchronize()
        s.reset_counters()
        s.set_option(pta.OPT_PROFILE, 2)
        for _ in range(2):
            s.pathtrace(sub.spp)
        s.synchronize()
        ms, launches = s.timers()
        c = s.counters()
        rays_kernel = float(c[pta.CNT_RAYS] - c[pta.CNT_TAIL_RAYS])
        alone_s = ms[pta.TIMER_TRACE] * 1e-3
        s.set_option(pta.OPT_PROFILE, 0)
        s.set_option(pta.OPT_STREAMS, lanes)
        s.set_option(pta.OPT_TRAVERSAL, form)
        # throughput as the product runs it
        for _ in range(2):
            s.pathtrace(sub.spp)
        s.synchronize()
        s.reset_counters()
        n = 6 if name == "bvh" else 4
        t0 = time.perf_counter()
        for _ in range(n):
            s.pathtrace(sub.spp)
        s.synchronize()
        dt = time.perf_counter() - t0
        c = s.counters()
        tf = fp32_tflops(n_box, n_tri, rays_kernel, alone_s)
        rec = {"Mrays/s": c[pta.CNT_RAYS] / dt * 1e-6, "spp_per_s": n * sub.spp / dt, "ms_per_step": dt / n * 1e3, "steps": n,
               "box_tests_per_ray": n_box, "tri_tests_per_ray": n_tri, "tree_in_lds": bool(s.get_option(pta.OPT_TREE_IN_LDS)),
               "lds_nodes": s.get_option(pta.OPT_LDS_NODES),
               "trace_kernel_alone": {"avg_launch_ms": ms[pta.TIMER_TRACE] / max(launches[pta.TIMER_TRACE], 1),
                                      "rays_per_launch": rays_kernel / max(launches[pta.TIMER_TRACE], 1),
                                      "fp32_TFLOPs": tf, "fp32_frac": tf / FP32_PEAK_TFLOPS}}
        cc = committed_counters(sub, 1)
        if cc:
            rec["committed_counters"] = summarise_counters(cc)
        out[name] = rec
        s.close()
    if not args.no_cpu_baseline:
        # CPU leg of the same view (oracle, SAH BVH; its kd builder is the reference's quadratic one and does not finish
        # on 210 k triangles): a bounded sample, the image shrunk until a frame holds at most 6 M samples
        sub.accel = "bvh"
        out["cpu_baseline"] = cpu_baseline(sub, budget_ref=8.0, frames_ref=2, budget_core=0.0, frames_core=0, most_samples=6_000_000)
        out["bvh"]["gpu_over_cpu"] = out["bvh"]["Mrays/s"] / out["cpu_baseline"]["value"]
    return out


def config1_record(pta, args, device, stream_handle):
    """The reference's own published case (README.md:63-65: teapot.obj, SAH BVH, 512x384, 10 spp per frame, 0.5 s per
    frame on the author's machine): seconds per frame of the GPU path and of the CPU oracle on this host.  The teapot
    has no emitter, so both are lit by the synthetic probe (the reference's probe is not distributed)."""
    import copy
    sub = copy.copy(args)
    sub.workload, sub.obj, sub.hdr, sub.accel = "teapot", None, None, "bvh"
    sub.width, sub.height, sub.spp = 512, 384, 10
    sub.lds_nodes = sub.traversal = sub.tail_level = sub.blocks_per_cu = sub.streams = sub.max_batch = None
    s = make_scene(pta, sub, device, pta.BVH)
    s.set_stream(stream_handle)
    for _ in range(20):
        s.pathtrace(sub.spp)
    s.synchronize()
    s.reset_counters()
    n = 200
    t0 = time.perf_counter()
    for _ in range(n):
        s.pathtrace(sub.spp)
    s.synchronize()
    dt = time.perf_counter() - t0
    rays = s.counters()[pta.CNT_RAYS]
    s.close()
    out = {"workload": "teapot.obj + synthetic probe 512x384, 10 spp/frame, SAH BVH (BASELINE config 1)",
           "reference_published_s_per_frame": 0.5,
           "gpu": {"s_per_frame": dt / n, "Mrays/s": rays / dt * 1e-6, "spp_per_s": n * sub.spp / dt, "steps": n}}
    if not args.no_cpu_baseline:
        cpu = cpu_baseline(sub, budget_ref=3.0, frames_ref=8, budget_core=2.0, frames_core=4)
        out["cpu_baseline"] = cpu
        out["gpu_over_cpu"] = out["gpu"]["Mrays/s"] / cpu["value"]
    return out


def summarise_counters(cc):
    """The figures of committed_counters() that the bench line carries."""
    out = {}
    v, c = cc.get("valu"), cc.get("cache")
    if v:
        out["valu_issue_busy"] = v.get("valu_issue_busy")
        out["salu_issue_busy"] = v.get("salu_issue_busy")
        out["valu_lane_occupancy"] = v.get("valu_lane_occupancy")
        out["valu_source"] = cc["valu_source"]
    if c:
        for k in ("l1_read_hit_rate", "l1_accesses_per_us", "l1_frac_of_peak", "l2_hit_rate", "l2_requests_per_us", "l2_request_frac_of_peak", "l2_GBs_upper_bound", "l2_frac_of_peak_upper_bound", "hbm_bytes_per_launch", "hbm_GBs",
                  "hbm_frac_of_peak", "avg_launch_s"):
            if k in c:
                out[k] = c[k]
        out["cache_source"] = cc["cache_source"]
    return out


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(n):
    """`bench.py --gpus N` started plainly (no launcher, WORLD_SIZE unset): this process becomes the parent of N
    copies of itself, one per GPU, with the environment torch.distributed.run would give them (RANK, LOCAL_RANK,
    WORLD_SIZE, MASTER_ADDR = 127.0.0.1, a free MASTER_PORT).  The parent never opens the GPU, never imports torch
    and prints nothing: rank 0 writes the one JSON line to the standard output they share.  Returns the exit code
    (non-zero as soon as one rank fails; the others are then ended by their own pids)."""
    import subprocess
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), PT_BENCH_SELF_LAUNCHED="1")
        procs.append(subprocess.Popen([sys.executable, "-u", os.path.abspath(__file__)] + sys.argv[1:], env=env))
    code = 0
    live = list(procs)
    deadline = time.time() + float(os.environ.get("PT_BENCH_LAUNCH_TIMEOUT", "3000"))
    while live:
        time.sleep(0.05)
        if time.time() > deadline:  # a rank that never returns must not hold the caller for ever
            for q in live:
                q.kill()
            return 124
        for p in list(live):
            rc = p.poll()
            if rc is None:
                continue
            live.remove(p)
            if rc != 0 and code == 0:
                code = rc if rc > 0 else 1
                for q in live:  # one rank down: the collective of the others would wait for ever
                    q.terminate()
    return code


class stdout_to_stderr:
    """RCCL and gloo announce themselves on standard output when a communicator is made; this program's standard output
    is one JSON line, so descriptor 1 points at descriptor 2 while the group is set up."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)


def dry_launch(args, world, rank):
    """--dry-launch: what the launcher half promises, without a GPU: every rank joins a gloo group and reports itself;
    rank 0 prints how many did."""
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    seen = torch.zeros(world, dtype=torch.int64)
    seen[rank] = 1 + int(os.environ.get("LOCAL_RANK", "0"))
    with stdout_to_stderr():
        dist.init_process_group("gloo", rank=rank, world_size=world)
        dist.all_reduce(seen)
    if rank == 0:
        print(json.dumps({"dry_launch": True, "n_gpus": world, "ranks_reported": int((seen > 0).sum()),
                          "local_ranks": [int(v) - 1 for v in seen], "gpus_asked": args.gpus,
                          "self_launched": os.environ.get("PT_BENCH_SELF_LAUNCHED") == "1", "backend": "gloo"}))
    dist.barrier()
    dist.destroy_process_group()


def reduce_cadence(frame_ms, n_floats, world):
    """Frames a rank adds into one per-frame buffer before the buffer is reduced (--reduce-every 0 = this rule).
    A reduce of B bytes onto one rank over xGMI is priced at 0.1 ms + B / 50 GB/s (ring over the 153 GB/s links at about a
    third of a link's rate, the small-message end of RCCL's curve) and may cost 3 % of the frames it covers."""
    if world <= 1:
        return 1
    reduce_ms = 0.1 + n_floats * 4.0 / 50e6
    k = int(np.ceil(reduce_ms / (0.03 * max(frame_ms, 1e-3))))
    return max(1, min(k, 16))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None,
                    help="timed frames (default 200, 12 for the sponza workload: with a few dozen 2-ms frames the empty pipeline "
                         "at both ends of the timed region costs 2-3 %%)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed frames before (default 10, 2 for the sponza workload)")
    ap.add_argument("--workload", default="cornellbox", choices=sorted(WORKLOADS))
    ap.add_argument("--obj", default=None, help="render this .obj instead of a named workload")
    ap.add_argument("--hdr", default=None)
    ap.add_argument("--accel", default="bvh", choices=["bvh", "kd"])
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--spp", type=int, default=None, help="samples per pixel per frame and per GPU")
    ap.add_argument("--lds-nodes", type=int, default=None, help="tree nodes staged in LDS by the trace kernel (PT_OPT_LDS_NODES)")
    ap.add_argument("--blocks-per-cu", type=int, default=None, help="PT_OPT_BLOCKS_PER_CU")
    ap.add_argument("--streams", type=int, default=None, help="PT_OPT_STREAMS: batches (frames) in flight, 1..4")
    ap.add_argument("--tail-level", type=int, default=None, help="PT_OPT_TAIL_LEVEL (0 = one launch per level)")
    ap.add_argument("--max-batch", type=int, default=None, help="PT_OPT_MAX_BATCH: most camera samples of one wavefront batch")
    ap.add_argument("--traversal", type=lambda v: int(v, 0), default=None, help="PT_OPT_TRAVERSAL bits")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kd", action="store_true", help="skip the short kd-tree comparison run")
    ap.add_argument("--strong", action="store_true",
                    help="strong scaling: --spp is the frame's TOTAL samples per pixel, split over the ranks by sample_share "
                         "(the analogue of the reference's column strips, src/cs488.h:1885-1890); default is weak scaling, --spp per GPU")
    ap.add_argument("--force-collective", action="store_true",
                    help="at N = 1 too: initialise the process group and run every frame through the sample-split entry, the "
                         "external per-frame buffer and the reduce (a one-rank communicator), so that the path N > 1 takes executes")
    ap.add_argument("--no-sponza", action="store_true", help="skip the sponza stand-in sub-record")
    ap.add_argument("--no-pmc", action="store_true", help="do not run the rocprofv3 --pmc child passes for live HBM traffic")
    ap.add_argument("--reduce-every", type=int, default=0,
                    help="N > 1: frames added into one per-frame buffer before it is reduced onto rank 0 (0 = chosen from the "
                         "measured frame time so that the reduce costs about 3 %%: 1 for frames of tens of ms, several for 2-ms frames)")
    ap.add_argument("--dry-launch", action="store_true",
                    help="start the ranks, let every one report itself over gloo, print the count and exit: no GPU is touched")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend; gloo only rehearses the multi-process path on a box with one GPU")
    args = ap.parse_args()
    dw, dh, dspp = DEFAULT_SHAPE.get(args.workload, (1024, 1024, 10))
    args.width = args.width or dw
    args.height = args.height or dh
    args.spp = args.spp or dspp
    if args.steps is None:
        args.steps = 12 if args.workload == "sponza" else 200
    if args.warmup is None:
        args.warmup = 2 if args.workload == "sponza" else 10

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started without a launcher: start the ranks here, before anything touches the GPU (or imports torch)
        sys.exit(launch_ranks(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.dry_launch:
        dry_launch(args, world, rank)
        return
    # counter passes in child processes first, while this process has not touched the GPU yet
    live_traffic = live_hbm_traffic(args, world) if rank == 0 else None

    import torch
    import torch.distributed as dist
    import path_tracer_amd as pta

    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the renderer has no CPU path")
    if args.backend == "nccl" and world > torch.cuda.device_count():
        raise SystemExit("--gpus %d with the RCCL backend needs %d GPUs and this node shows %d (--backend gloo rehearses the "
                         "multi-process path on fewer)" % (world, world, torch.cuda.device_count()))
    local_rank %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    collective = world > 1 or args.force_collective
    if collective:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        with stdout_to_stderr():  # until the first collective has run
            if args.backend == "nccl":
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
            else:
                dist.init_process_group("gloo", rank=rank, world_size=world)
            dist.barrier()
            if args.backend == "nccl":
                torch.cuda.synchronize()

    accel = pta.BVH if args.accel == "bvh" else pta.KD
    scene = make_scene(pta, args, local_rank, accel)
    stream = torch.cuda.current_stream()
    scene.set_stream(stream.cuda_stream)
    n_pix = args.width * args.height

    reducer = None
    frame_spp = args.spp if args.strong else args.spp * world  # samples per pixel a frame adds, all ranks together
    if collective:
        from path_tracer_amd.distributed import FrameReducer, sample_share
        deltas = [torch.zeros(n_pix * 3, dtype=torch.float32, device="cuda") for _ in range(2)]
        master = torch.zeros(n_pix * 3, dtype=torch.float32, device="cuda") if rank == 0 else None
        reducer = FrameReducer(dist, deltas, master, every=max(args.reduce_every, 1))
        first, count = sample_share(frame_spp, world, rank)
        # Paths that fork (translucent surfaces) can outgrow a batch's ray queues; such a batch adds nothing until
        # it is rendered again at the next synchronisation, when its frame's buffer may long be in a reduce.  So
        # every wait for the renderer below goes through FrameReducer.settle, which hands the wait a buffer of its
        # own and reduces that too: no frame waits, no sample is lost.

    def step():
        if not collective:
            scene.pathtrace(args.spp)
        else:
            # frame k renders into buffer k % 2 while the reduce of frame k-1 is still in flight
            scene.use_external_accum(reducer.begin_frame().data_ptr())
            scene.pathtrace_split(frame_spp, first, count)
            reducer.reduce(frame_spp)

    def settle():
        if reducer is not None:
            reducer.settle(scene)
        else:
            scene.synchronize()

    def fence():
        settle()
        if reducer is not None:
            reducer.finish()
        torch.cuda.synchronize()
        if collective:
            dist.barrier()
        torch.cuda.synchronize()

    # untimed counted pass: slab and triangle tests per ray for the roofline's byte count
    scene.set_option(pta.OPT_COUNT_TESTS, 1)
    scene.reset_counters()
    step()
    settle()
    c = scene.counters()
    n_box = c[pta.CNT_BOX_TESTS] / max(c[pta.CNT_RAYS], 1)
    n_tri = c[pta.CNT_TRI_TESTS] / max(c[pta.CNT_RAYS], 1)
    hit_frac = c[pta.CNT_HITS] / max(c[pta.CNT_RAYS], 1)
    scene.set_option(pta.OPT_COUNT_TESTS, 0)

    fence()
    w0 = time.perf_counter()
    for _ in range(args.warmup):
        step()
    fence()
    if reducer is not None and args.reduce_every <= 0 and args.warmup > 0:
        # frames per reduce from the frame time just seen (the slowest rank's, so that every rank decides alike)
        t = torch.tensor([(time.perf_counter() - w0) / args.warmup * 1e3], dtype=torch.float64,
                         device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        reducer.every = reduce_cadence(float(t.item()), n_pix * 3, world)
    # per-stage milliseconds from a few untimed frames with events around every kernel, one frame in
    # flight at a time (with more, a kernel's span includes the time it shares the chip) ...
    # (traversal bit 3: the kernel form of the timed region, not the pooled one a lone frame would get)
    streams = scene.get_option(pta.OPT_STREAMS)
    form = scene.get_option(pta.OPT_TRAVERSAL)
    tail_setting = scene.get_option(pta.OPT_TAIL_LEVEL)
    tail_used = scene.get_option(pta.OPT_TAIL_LEVEL_USED)  # where the warm-up frames handed over to the fused launch
    scene.set_option(pta.OPT_STREAMS, 1)
    scene.set_option(pta.OPT_TRAVERSAL, form | 8)
    # one frame at a time the library would hand some scenes over earlier (a small kd-tree after the camera rays): the
    # stage pass times the launches of the timed region, so it keeps that region's hand-over level
    scene.set_option(pta.OPT_TAIL_LEVEL, tail_used if tail_used > 0 else tail_setting)
    scene.reset_counters()
    scene.set_option(pta.OPT_PROFILE, 1)
    stage_frames = min(args.steps, 5)
    for _ in range(stage_frames):
        step()
    fence()
    stage_ms, stage_launches = scene.timers()
    cs = scene.counters()
    stage_rays_kernel = float(cs[pta.CNT_RAYS] - cs[pta.CNT_TAIL_RAYS])
    stage_trace_launch_ms = stage_ms[pta.TIMER_TRACE] / max(stage_launches[pta.TIMER_TRACE], 1)
    stage_trace_s = stage_ms[pta.TIMER_TRACE] * 1e-3
    stage_ms = [m / stage_frames for m in stage_ms]
    scene.set_option(pta.OPT_PROFILE, 0)
    scene.set_option(pta.OPT_STREAMS, streams)
    scene.set_option(pta.OPT_TRAVERSAL, form)
    scene.set_option(pta.OPT_TAIL_LEVEL, tail_setting)
    # ... and the timed region with events around the launches of the dominant kernel only
    scene.reset_counters()
    scene.set_option(pta.OPT_PROFILE, 2)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    scene.set_option(pta.OPT_PROFILE, 0)
    c = scene.counters()
    ms, launches = scene.timers()
    rays_local = float(c[pta.CNT_RAYS])
    rays_tail = float(c[pta.CNT_TAIL_RAYS])     # traced by the fused launch of the deep levels, timed separately
    rays_kernel = rays_local - rays_tail        # traced by trace_kernel launches
    max_queue = c[pta.CNT_MAX_QUEUE]

    if world > 1:
        dev = "cuda" if args.backend == "nccl" else "cpu"
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        per_rank = torch.zeros(world, dtype=torch.float64, device=dev)
        per_rank[rank] = rays_local
        dist.all_reduce(per_rank, op=dist.ReduceOp.SUM)
        rays_total = float(per_rank.sum().item())
        rays_per_rank = [float(v) for v in per_rank.tolist()]
    else:
        rays_total = rays_local
        rays_per_rank = [rays_local]

    out = None
    if rank == 0:
        b_ray = bytes_per_ray(n_box, n_tri, hit_frac)
        trace_s = ms[pta.TIMER_TRACE] * 1e-3
        n_launch = max(launches[pta.TIMER_TRACE], 1)
        avg_launch_s = trace_s / n_launch
        in_lds = bool(scene.get_option(pta.OPT_TREE_IN_LDS))
        tflops = fp32_tflops(n_box, n_tri, rays_kernel, trace_s)
        tflops_alone = fp32_tflops(n_box, n_tri, stage_rays_kernel, stage_trace_s)
        cc = committed_counters(args, world)
        counters = summarise_counters(cc) if cc else None
        # HBM bytes per launch: this run's own counter passes, else the committed ones of the same workload
        traffic, traffic_source = None, None
        if live_traffic:
            traffic = live_traffic[0]
            traffic_source = ("rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE child passes of this run (tools/prof_frame.py: 3 frames of the same "
                              "workload, %d trace_kernel launches, one frame in flight; read side doubled for gfx950)" % live_traffic[1])
        elif counters and "hbm_bytes_per_launch" in counters:
            traffic, traffic_source = counters["hbm_bytes_per_launch"], counters["cache_source"] + " (committed; not measured in this run)"
        # Fractions of every ceiling that applies.  A tree staged in LDS has no memory path in the walk: its operands
        # cost vector instructions, so its bound is the FP32 peak (the HBM figure beside it is the ray and hit queues).
        # A tree that leaves the CU is quoted against the ceiling it is closest to: FP32, vector L1 (tag accesses), L2
        # (requests: the walk's fetches are random 64-byte reads, whose measured ceilings are far below the streaming
        # byte rates) or HBM, the cache figures from the counter passes of the same workload.
        fracs = {"valu": tflops / FP32_PEAK_TFLOPS}
        hbm_gbs = traffic / stage_trace_launch_ms * 1e-6 if (traffic and stage_trace_launch_ms > 0) else None
        if hbm_gbs is not None:
            fracs["hbm"] = hbm_gbs / HBM_PEAK_GBS
        if not in_lds and counters:
            if "l2_request_frac_of_peak" in counters:
                fracs["l2"] = counters["l2_request_frac_of_peak"]
            if "l1_frac_of_peak" in counters:
                fracs["l1"] = counters["l1_frac_of_peak"]
        bound = "valu" if in_lds else max(fracs, key=lambda k: fracs[k])
        if bound == "valu":
            roof = {"bound": "valu", "achieved": tflops, "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": fracs["valu"]}
        elif bound == "hbm":
            roof = {"bound": "hbm", "achieved": hbm_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": fracs["hbm"]}
        elif bound == "l1":
            roof = {"bound": "l1", "achieved": counters["l1_accesses_per_us"] * 1e-3, "peak": L1_PEAK_ACCESSES_PER_US * 1e-3, "unit": "G accesses/s", "frac": fracs["l1"]}
        else:
            roof = {"bound": "l2", "achieved": counters["l2_requests_per_us"] * 1e-3, "peak": L2_PEAK_REQUESTS_PER_US * 1e-3, "unit": "G requests/s", "frac": fracs["l2"]}
        roof.update({
            "traffic": traffic, "traffic_unit": "HBM bytes per launch", "traffic_source": traffic_source,
            "kernel": "trace_kernel (closest hit)", "tree_in_lds": in_lds,
            "fractions": fracs,
            "fp32": {"achieved": tflops, "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tflops / FP32_PEAK_TFLOPS,
                     "flop_per_ray": FLOP_PER_BOX_TEST * n_box + FLOP_PER_TRI_TEST * n_tri,
                     # the walk may not fuse a multiply with an add (the reference's bits), so one FLOP per lane and issue
                     # slot is the most its instruction stream can reach: half the peak above
                     "frac_of_unfused_peak": tflops / (0.5 * FP32_PEAK_TFLOPS),
                     "definition": "algorithmic: 27 FLOP per slab test + 75 per triangle test, tests counted by the kernel in an untimed pass"},
            "hbm": {"achieved": hbm_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm_gbs / HBM_PEAK_GBS if hbm_gbs is not None else None,
                    "definition": "measured HBM bytes per launch / launch time with one frame in flight (the counter passes run that way)"},
            "committed_counters": counters,
            "box_tests_per_ray": n_box, "tri_tests_per_ray": n_tri, "hit_fraction": hit_frac,
            "launches": launches[pta.TIMER_TRACE], "avg_launch_ms": avg_launch_s * 1e3,
            # sum of the trace launches' durations per frame: above ms_per_step when frames overlap
            "kernel_ms_per_step": trace_s * 1e3 / args.steps,
            "rays_per_launch": rays_kernel / n_launch,
            "rays_in_fused_tail_fraction": rays_tail / max(rays_local, 1.0),
            # the survey's per-ray byte formula prices node and triangle operands as memory traffic; kept as a named
            # figure only, because those operands come from LDS or cache and not from HBM
            "algorithmic_operand_bytes": {"per_ray": b_ray, "per_launch": rays_kernel * b_ray / n_launch,
                                          "rate_GBs": rays_kernel * b_ray / trace_s * 1e-9 if trace_s > 0 else 0.0},
            "note": ("%d frames are in flight on streams of their own in the timed region, so a launch's duration includes the time it "
                     "shares the chip with the other frames' kernels; `alone` is the same kernel with one frame in flight "
                     "(%d untimed frames)" % (streams, stage_frames)) if streams > 1 else None,
            "alone": {"achieved": tflops_alone, "frac": tflops_alone / FP32_PEAK_TFLOPS, "unit": "TFLOP/s", "avg_launch_ms": stage_trace_launch_ms},
        })
        out = {
            "metric": "Mrays/s", "value": rays_total / elapsed * 1e-6, "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong" if args.strong else "weak",
            "vs_baseline": None, "dtype": "f32", "data": ("synthetic scene + synthetic probe" if args.workload == "sponza" else
                     "reference scene file + synthetic probe" if WORKLOADS[args.workload][1] else "reference scene file"),
            "config": {
                "workload": "%s %dx%d, %d spp/frame%s, SAH %s, progressive" % (
                    args.obj or WORKLOADS[args.workload][0] or "sponza stand-in (tools/make_sponza.py)", args.width, args.height,
                    frame_spp if args.strong else args.spp, "" if args.strong else "/GPU", args.accel.upper()),
                "frames_in_flight": streams, "fused_tail_from_level": int(tail_used), "lane_streams_replaced": int(c[pta.CNT_STREAM_SWAPS]),
                "lds_nodes": int(scene.get_option(pta.OPT_LDS_NODES)),
                "parallelism": ("sample-split x%d + %s reduce of the accumulation buffer every %d frame(s)" % (
                    world, "RCCL" if args.backend == "nccl" else "gloo (host-staged rehearsal)", reducer.every)) if collective else "single GPU",
                "reduce_every": reducer.every if reducer is not None else None,
                "reduces_in_timed_region_and_before": reducer.reduces if reducer is not None else None,
                "collective_ranks": dist.get_world_size() if collective else None,
                "collective_backend": dist.get_backend() if collective else None,
                "self_launched": os.environ.get("PT_BENCH_SELF_LAUNCHED") == "1",
            },
            "per_rank_Mrays/s": [v / elapsed * 1e-6 for v in rays_per_rank],
            "spp_per_s": frame_spp * args.steps / elapsed,
            "rays_per_sample": rays_total / (float(n_pix) * frame_spp * args.steps),
            "roofline": roof,
            "stage_ms_per_step": {
                "raygen": stage_ms[pta.TIMER_RAYGEN], "trace": stage_ms[pta.TIMER_TRACE],
                "shade": stage_ms[pta.TIMER_SHADE], "accumulate": stage_ms[pta.TIMER_ACCUM],
                "fused_tail": stage_ms[pta.TIMER_TAIL],
                "note": "%d untimed frames, one in flight at a time, events on every kernel (the timed region overlaps frames, so its stages do not add up to ms_per_step)" % stage_frames,
            },
            "max_queue": max_queue,
        }

    if world == 1 and rank == 0 and not collective:
        if not args.no_kd and args.accel == "bvh":
            scene.close()  # its streams go first: a process's streams share a few hardware queues
            scene = None
            kd = make_scene(pta, args, local_rank, pta.KD)
            kd.set_stream(stream.cuda_stream)
            for _ in range(4):
                kd.pathtrace(args.spp)
            kd.synchronize()
            kd.reset_counters()
            k0 = time.perf_counter()
            n_kd = max(3, args.steps // 3)
            for _ in range(n_kd):
                kd.pathtrace(args.spp)
            kd.synchronize()
            k1 = time.perf_counter() - k0
            out["kd_tree"] = {"Mrays/s": kd.counters()[pta.CNT_RAYS] / k1 * 1e-6, "spp_per_s": n_kd * args.spp / k1, "steps": n_kd}
            kd.close()
        if not args.no_sponza and is_default_workload(args, world):
            if scene is not None:
                scene.close()
                scene = None
            out["sponza"] = sponza_record(pta, args, local_rank, stream.cuda_stream)
            out["config1"] = config1_record(pta, args, local_rank, stream.cuda_stream)
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args)
            out["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]

    if rank == 0:
        print(json.dumps(out))
    if collective:
        dist.barrier()
        dist.destroy_process_group()
    if scene is not None:
        scene.close()


if __name__ == "__main__":
    main()
