#!/usr/bin/env python3
"""Small workload for rocprofv3 passes: a few progressive frames of one scene, no PyTorch.
One frame in flight at a time unless PT_PROF_STREAMS says otherwise: the passes are read per kernel.  The trace kernel
is the form that runs with two frames in flight (workgroups on their own regions, traversal bit 3) unless a mode is given.
PT_PROF_TAIL_LEVEL sets PT_OPT_TAIL_LEVEL (0: one trace launch per bounce level, no fused launch).
usage: prof_frame.py [scene.obj] [resolution] [frames] [accel bvh|kd] [traversal mode]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import path_tracer_amd as pta  # noqa: E402
from common import scene_path  # noqa: E402

obj = sys.argv[1] if len(sys.argv) > 1 else "cornellbox.obj"
res = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 3
accel = pta.KD if (len(sys.argv) > 4 and sys.argv[4] == "kd") else pta.BVH
s = pta.Scene(0)
spp = 10
if obj == "sponza":  # the 3840x2160, 4 spp/frame courtyard workload of bench.py
    import bench
    from make_sponza import CAMERA
    assert s.load(bench.sponza_path())
    s.load_env(scene_path("probe_64.hdr"))
    s.set_camera(**CAMERA)
    spp = 4
else:
    assert s.load(scene_path(obj))
    if obj.startswith("teapot"):
        s.load_env(scene_path("probe_64.hdr"))
s.pre_calc(accel)
s.set_option(pta.OPT_STREAMS, int(os.environ.get("PT_PROF_STREAMS", "1")))
s.set_option(pta.OPT_TRAVERSAL, int(sys.argv[5], 0) if len(sys.argv) > 5 else s.get_option(pta.OPT_TRAVERSAL) | 8)
if "PT_PROF_TAIL_LEVEL" in os.environ:
    s.set_option(pta.OPT_TAIL_LEVEL, int(os.environ["PT_PROF_TAIL_LEVEL"]))
s.set_resolution(*((3840, 2160) if obj == "sponza" else (res, res)))
for _ in range(frames):
    s.pathtrace(spp)
s.synchronize()
c = s.counters()
print("rays", c[0], "samples", c[3])
