#!/usr/bin/env python3
"""GPU experiment: time the trace stage under every traversal mode and check each result-exact
mode against the CPU oracle on a ray batch.  mode bits: 0 = reciprocal-form slab test, 2 = refill only when the whole wave is idle,
4..7 = k of the node phase's early-leave rule."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")):  # run as: python tests/exp_trace_modes.py
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import orc  # noqa: E402
import path_tracer_amd as pta  # noqa: E402
from common import load_gpu_scene, load_oracle_scene, ray_batch  # noqa: E402

scenes = sys.argv[1].split(",") if len(sys.argv) > 1 else ["cornellbox.obj"]
res = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
for obj in scenes:
    for accel in (pta.BVH, pta.KD):
        o = load_oracle_scene(obj, accel, textures=False)
        rays = ray_batch(o.get_tris()[0], n_random=20000, n_vertex=2000)
        ref = o.trace(rays)
        s = load_gpu_scene(obj, accel, env="probe_64.hdr" if obj.startswith("teapot") else None)
        s.set_resolution(res, res)
        for mode in (0x00, 0x01, 0x20, 0x21, 0x31, 0x41, 0x25):
            s.set_option(pta.OPT_TRAVERSAL, mode)
            got = s.intersect(rays)
            same_tri = int((got["tri"] != ref["tri"]).sum())
            same_t = int((got["t"].view(np.uint32) != ref["t"].view(np.uint32)).sum())
            s.reset()
            for _ in range(2):
                s.pathtrace(10)
            s.synchronize()
            s.reset_counters()
            s.set_option(pta.OPT_PROFILE, 1)
            t0 = time.perf_counter()
            for _ in range(8):
                s.pathtrace(10)
            s.synchronize()
            dt = (time.perf_counter() - t0) / 8
            ms, n = s.timers()
            c = s.counters()
            s.set_option(pta.OPT_PROFILE, 0)
            print("%-22s %s mode %02x: tri diffs %5d  t diffs %5d | frame %.2f ms trace %.2f shade %.2f | %.0f Mrays/s" % (
                obj, "kd " if accel else "bvh", mode, same_tri, same_t, dt * 1e3, ms[1] / 8, ms[2] / 8, c[0] / 8 / dt * 1e-6), flush=True)
