#!/usr/bin/env python3
"""Share of lanes at work in the two phases of the walk, from the counting instantiation:
lane node steps / (64 x wave node steps) and triangle tests / (64 x wave triangle rounds).
usage: lane_use.py [scene.obj ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import path_tracer_amd as pta  # noqa: E402
from common import scene_path  # noqa: E402

for obj in (sys.argv[1:] or ["cornellbox.obj"]):
    for accel in (pta.BVH, pta.KD):
        s = pta.Scene(0)
        assert s.load(scene_path(obj))
        if obj.startswith("teapot"):
            s.load_env(scene_path("probe_64.hdr"))
        s.pre_calc(accel)
        s.set_option(pta.OPT_COUNT_TESTS, 1)
        s.set_resolution(1024, 1024)
        s.pathtrace(10)
        s.synchronize()
        c = s.counters()
        rays = c[pta.CNT_RAYS]
        steps, rounds = c[pta.CNT_WAVE_STEPS], c[pta.CNT_WAVE_ROUNDS]
        print("%-22s %s: %.1f node steps/ray at %.0f %% of lanes, %.1f triangle tests/ray at %.0f %% of lanes; per ray %.3f wave steps + %.3f wave rounds"
              % (obj, "kd " if accel == pta.KD else "bvh", c[pta.CNT_BOX_TESTS] / 2 / rays, 100.0 * c[pta.CNT_BOX_TESTS] / 2 / (64.0 * steps),
                 c[pta.CNT_TRI_TESTS] / rays, 100.0 * c[pta.CNT_TRI_TESTS] / (64.0 * rounds), steps / rays, rounds / rays), flush=True)
        s.close()
