#!/usr/bin/env python3
"""How much would two batches in flight on two streams gain?  Renders the same frames with one
context, then with two contexts (each has its own stream and queues) fed alternately from one host
thread, optionally with fewer resident workgroups per CU so that the other stream's kernels fit
beside the walk.  usage: two_stream_probe.py [scene.obj] [resolution] [frames] [accel]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import path_tracer_amd as pta  # noqa: E402
from common import scene_path  # noqa: E402

obj = sys.argv[1] if len(sys.argv) > 1 else "cornellbox.obj"
res = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 60
accel = pta.KD if len(sys.argv) > 4 and sys.argv[4] == "kd" else pta.BVH


def make(per_cu):
    s = pta.Scene(0)
    assert s.load(scene_path(obj))
    s.pre_calc(accel)
    s.set_resolution(res, res)
    s.set_option(pta.OPT_PROFILE, 0)
    if per_cu:
        s.set_option(pta.OPT_BLOCKS_PER_CU, per_cu)
    return s


def run(ctxs, spp):
    for _ in range(4):
        for s in ctxs:
            s.pathtrace(spp)
    for s in ctxs:
        s.synchronize()
    t0 = time.perf_counter()
    for _ in range(frames):
        for s in ctxs:
            s.pathtrace(spp)
    for s in ctxs:
        s.synchronize()
    return (time.perf_counter() - t0) / frames * 1e3


for per_cu in (0, 4):
    for n in (1, 2, 3, 4):
        ctxs = [make(per_cu) for _ in range(n)]
        t = run(ctxs, 10)
        print("blocks/CU %d: %d contexts x 10 spp: %.3f ms per 10 spp" % (per_cu, n, t / n), flush=True)
        if n > 1 and 10 % n == 0:
            t = run(ctxs, 10 // n)
            print("blocks/CU %d: %d contexts x %d spp: %.3f ms per 10 spp" % (per_cu, n, 10 // n, t), flush=True)
        del ctxs
