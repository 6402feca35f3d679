#!/usr/bin/env python3
"""Milliseconds per frame as a function of how many frames are timed, per number of frames in flight.
usage: lane_probe.py [scene.obj] [bvh|kd] [resolution]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import path_tracer_amd as pta  # noqa: E402
from common import scene_path  # noqa: E402

obj = sys.argv[1] if len(sys.argv) > 1 else "cornellbox.obj"
accel = pta.KD if len(sys.argv) > 2 and sys.argv[2] == "kd" else pta.BVH
res = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
for lanes in (1, 2, 3):
    s = pta.Scene(0)
    assert s.load(scene_path(obj))
    s.pre_calc(accel)
    s.set_option(pta.OPT_STREAMS, lanes)
    s.set_resolution(res, res)
    for _ in range(2):
        s.pathtrace(10)
    s.synchronize()
    line = []
    for frames in (2, 4, 6, 6, 10, 20, 40):
        t0 = time.perf_counter()
        for _ in range(frames):
            s.pathtrace(10)
        s.synchronize()
        line.append("%d: %.3f" % (frames, (time.perf_counter() - t0) / frames * 1e3))
    print("lanes %d  ms/frame by frames timed  %s" % (lanes, "  ".join(line)), flush=True)
    s.close()
