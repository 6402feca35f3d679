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
ms_per_step, `roofline` for the dominant kernel (algorithmic bytes per launch over the launch time measured
with HIP events, committed counter figures for HBM traffic and VALU use), per-stage milliseconds, a short
kd-tree run, and `cpu_baseline` (the CPU oracle timed on this host, rank 0 at N = 1 only).
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

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E, spec

WORKLOADS = {
    # name: (obj, needs probe)
    "cornellbox": ("cornellbox.obj", False),
    "cornellbox-glass": ("cornellbox-glass.obj", True),
    "cornellbox-metal": ("cornellbox-metal.obj", True),
    "cornellbox-alpha": ("cornellbox-alpha.obj", True),
    "cornellball-glass": ("cornellball-glass.obj", False),
    "teapot": ("teapot.obj", True),
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


def measured_traffic(args, world):
    """HBM bytes per trace_kernel launch from the committed rocprofv3 --pmc passes (FETCH_SIZE and
    WRITE_SIZE collected in separate runs of the same workload, read side doubled as gfx950 needs);
    counters cannot be read inside this process, so the number is only attached to the workload
    those passes ran (the default one) and is null otherwise."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    is_default = (args.workload == "cornellbox" and not args.obj and args.accel == "bvh" and
                  (args.width, args.height, args.spp) == (1024, 1024, 10) and world == 1)
    if not is_default or not os.path.exists(path):
        return None, None
    k = json.load(open(path))["kernels"].get("trace_kernel")
    return (k["hbm_bytes_per_launch"], "profiles/r01_pmc_traffic.json") if k else (None, None)


def measured_valu(args, world):
    """VALU side of trace_kernel from the committed SQ counter passes (tools/pmc_valu.py): share of
    the chip's VALU issue slots in use and lanes active per VALU instruction.  Same restriction as
    measured_traffic: only for the workload the passes ran."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_valu.json")
    is_default = (args.workload == "cornellbox" and not args.obj and args.accel == "bvh" and
                  (args.width, args.height, args.spp) == (1024, 1024, 10) and world == 1)
    if not is_default or not os.path.exists(path):
        return None
    k = json.load(open(path))["kernels"].get("trace_kernel")
    if not k:
        return None
    return {"issue_busy": k.get("valu_issue_busy"), "salu_issue_busy": k.get("salu_issue_busy"), "lane_occupancy": k.get("valu_lane_occupancy"),
            "wave_instructions_per_launch": k["per_launch"].get("SQ_INSTS_VALU"), "source": "profiles/r01_pmc_valu.json"}


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


def cpu_baseline(args):
    """The CPU restatement of the reference's multithreaded loop (column strips, one thread per
    strip), timed on this host on a bounded sample of the same workload."""
    import orc
    from PIL import Image
    from common import read_hdr_independent
    needs_probe = WORKLOADS[args.workload][1]
    cores = min(16, os.cpu_count() or 1)
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
    while (w // shrink) * (h // shrink) * args.spp > 12_000_000:
        shrink *= 2
    w, h = w // shrink, h // shrink
    o.reset(w, h)
    o.render(spp=1, frames=1, threads=cores)  # warm-up
    o.reset(w, h)
    frames, t0, rays = 0, time.time(), 0
    while True:
        _, _, cnt = o.render(spp=args.spp, frames=1, threads=cores)
        rays += int(cnt[0])
        frames += 1
        dt = time.time() - t0
        if dt > 10.0 or frames >= 5:
            break
    return {
        "value": rays / dt * 1e-6, "unit": "Mrays/s", "cores": cores, "kind": "port",
        "spp_per_s": frames * args.spp / dt * (w * h) / float(args.width * args.height),
        "sample": "%d frame(s) of the same view at %dx%d (1/%d of the workload's resolution per axis) x %d spp, %d threads over "
                  "column strips; spp_per_s is scaled to the workload's pixel count" % (frames, w, h, shrink, args.spp, cores),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=3)
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
    ap.add_argument("--traversal", type=lambda v: int(v, 0), default=None, help="PT_OPT_TRAVERSAL bits")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kd", action="store_true", help="skip the short kd-tree comparison run")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend; gloo only rehearses the multi-process path on a box with one GPU")
    args = ap.parse_args()
    dw, dh, dspp = DEFAULT_SHAPE.get(args.workload, (1024, 1024, 10))
    args.width = args.width or dw
    args.height = args.height or dh
    args.spp = args.spp or dspp

    import torch
    import torch.distributed as dist
    import path_tracer_amd as pta

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the renderer has no CPU path")
    local_rank %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    accel = pta.BVH if args.accel == "bvh" else pta.KD
    scene = make_scene(pta, args, local_rank, accel)
    stream = torch.cuda.current_stream()
    scene.set_stream(stream.cuda_stream)
    n_pix = args.width * args.height

    reducer = None
    if world > 1:
        from path_tracer_amd.distributed import FrameReducer, sample_share
        deltas = [torch.zeros(n_pix * 3, dtype=torch.float32, device="cuda") for _ in range(2)]
        master = torch.zeros(n_pix * 3, dtype=torch.float32, device="cuda") if rank == 0 else None
        reducer = FrameReducer(dist, deltas, master)
        first, count = sample_share(args.spp * world, world, rank)

    def step():
        if world == 1:
            scene.pathtrace(args.spp)
        else:
            # frame k renders into buffer k % 2 while the reduce of frame k-1 is still in flight
            scene.use_external_accum(reducer.begin_frame().data_ptr())
            scene.pathtrace_split(args.spp * world, first, count)
            reducer.reduce(args.spp * world)

    def fence():
        if reducer is not None:
            reducer.finish()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # untimed counted pass: slab and triangle tests per ray for the roofline's byte count
    scene.set_option(pta.OPT_COUNT_TESTS, 1)
    scene.reset_counters()
    step()
    scene.synchronize()
    c = scene.counters()
    n_box = c[pta.CNT_BOX_TESTS] / max(c[pta.CNT_RAYS], 1)
    n_tri = c[pta.CNT_TRI_TESTS] / max(c[pta.CNT_RAYS], 1)
    hit_frac = c[pta.CNT_HITS] / max(c[pta.CNT_RAYS], 1)
    scene.set_option(pta.OPT_COUNT_TESTS, 0)

    for _ in range(args.warmup):
        step()
    fence()
    # per-stage milliseconds from a few untimed frames with events around every kernel, one frame in
    # flight at a time (with more, a kernel's span includes the time it shares the chip) ...
    streams = scene.get_option(pta.OPT_STREAMS)
    scene.set_option(pta.OPT_STREAMS, 1)
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
        r = torch.tensor([rays_local], dtype=torch.float64, device=dev)
        dist.all_reduce(r, op=dist.ReduceOp.SUM)
        rays_total = float(r.item())
    else:
        rays_total = rays_local

    out = None
    if rank == 0:
        b_ray = bytes_per_ray(n_box, n_tri, hit_frac)
        trace_s = ms[pta.TIMER_TRACE] * 1e-3
        traffic, traffic_source = measured_traffic(args, world)
        achieved = rays_kernel * b_ray / trace_s * 1e-9 if trace_s > 0 else 0.0
        achieved_alone = stage_rays_kernel * b_ray / stage_trace_s * 1e-9 if stage_trace_s > 0 else 0.0
        out = {
            "metric": "Mrays/s", "value": rays_total / elapsed * 1e-6, "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": ("synthetic scene + synthetic probe" if args.workload == "sponza" else
                     "reference scene file + synthetic probe" if WORKLOADS[args.workload][1] else "reference scene file"),
            "config": {
                "workload": "%s %dx%d, %d spp/frame/GPU, SAH %s, progressive" % (
                    args.obj or WORKLOADS[args.workload][0] or "sponza stand-in (tools/make_sponza.py)", args.width, args.height,
                    args.spp, args.accel.upper()),
                "frames_in_flight": streams, "lane_streams_replaced": int(c[pta.CNT_STREAM_SWAPS]),
                "parallelism": ("sample-split x%d + %s reduce of the accumulation buffer" % (
                    world, "RCCL" if args.backend == "nccl" else "gloo (host-staged rehearsal)")) if world > 1 else "single GPU",
            },
            "spp_per_s": args.spp * world * args.steps / elapsed,
            "rays_per_sample": rays_local / (float(n_pix) * args.spp * args.steps),
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_unit": "bytes per launch",
                "traffic_source": traffic_source,
                "algorithmic_bytes_per_launch": rays_kernel * b_ray / max(launches[pta.TIMER_TRACE], 1),
                "kernel": "trace_kernel (closest hit)", "bytes_per_ray": b_ray,
                "box_tests_per_ray": n_box, "tri_tests_per_ray": n_tri, "hit_fraction": hit_frac,
                "launches": launches[pta.TIMER_TRACE], "avg_launch_ms": ms[pta.TIMER_TRACE] / max(launches[pta.TIMER_TRACE], 1),
                "rays_per_launch": rays_kernel / max(launches[pta.TIMER_TRACE], 1),
                "rays_in_fused_tail_fraction": rays_tail / max(rays_local, 1.0),
                "valu": measured_valu(args, world),
                "note": ("%d frames are in flight on streams of their own in the timed region, so a launch's duration includes the time it "
                         "shares the chip with the other frames' kernels; `alone` is the same kernel with one frame in flight "
                         "(%d untimed frames)" % (streams, stage_frames)) if streams > 1 else None,
                "alone": {"achieved": achieved_alone, "frac": achieved_alone / HBM_PEAK_GBS, "avg_launch_ms": stage_trace_launch_ms},
            },
            "stage_ms_per_step": {
                "raygen": stage_ms[pta.TIMER_RAYGEN], "trace": stage_ms[pta.TIMER_TRACE],
                "shade": stage_ms[pta.TIMER_SHADE], "accumulate": stage_ms[pta.TIMER_ACCUM],
                "fused_tail": stage_ms[pta.TIMER_TAIL],
                "note": "%d untimed frames, one in flight at a time, events on every kernel (the timed region overlaps frames, so its stages do not add up to ms_per_step)" % stage_frames,
            },
            "max_queue": max_queue,
        }

    if world == 1 and rank == 0:
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
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args)
            out["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]

    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if scene is not None:
        scene.close()


if __name__ == "__main__":
    main()
