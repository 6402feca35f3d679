#!/usr/bin/env python3
"""Wall-clock milliseconds per progressive frame with and without per-kernel event timing (PT_OPT_PROFILE).
usage: frame_time.py [scene.obj] [resolution] [frames]"""
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
s = pta.Scene(0)
assert s.load(scene_path(obj))
s.pre_calc(pta.BVH)
s.set_resolution(res, res)
for profile in (0, 1, 0, 1):
    s.set_option(pta.OPT_PROFILE, profile)
    for _ in range(5):
        s.pathtrace(10)
    s.synchronize()
    t0 = time.perf_counter()
    for _ in range(frames):
        s.pathtrace(10)
    s.synchronize()
    print("profile %d: %.3f ms per frame" % (profile, (time.perf_counter() - t0) / frames * 1e3), flush=True)
