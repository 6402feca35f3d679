#!/usr/bin/env python3
"""Contexts created one after another in one process (all kept alive): does each still overlap its two lanes?
A process's streams share a few hardware queues; the library probes its lanes and replaces a stream that runs in
order with another one.  usage: lane_probe2.py [null]   (null: the contexts use the null stream as theirs)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import path_tracer_amd as pta  # noqa: E402
from common import scene_path  # noqa: E402

null_stream = len(sys.argv) > 1 and sys.argv[1] == "null"
keep = []
for k in range(6):
    s = pta.Scene(0)
    assert s.load(scene_path("cornellbox.obj"))
    s.pre_calc(pta.BVH)
    if null_stream:
        s.set_stream(0)
    s.set_resolution(1024, 1024)
    for _ in range(4):
        s.pathtrace(10)
    s.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        s.pathtrace(10)
    s.synchronize()
    print("context %d (%s stream): %.3f ms/frame, lane streams replaced %d" % (
        k, "null" if null_stream else "own", (time.perf_counter() - t0) / 20 * 1e3, s.counters()[pta.CNT_STREAM_SWAPS]), flush=True)
    keep.append(s)
