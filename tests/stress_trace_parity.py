#!/usr/bin/env python3
"""GPU stress run, not collected by pytest: closest-hit parity against the CPU oracle on ray batches a hundred
times the size of the ones in test_gpu_trace_parity.py, every scene, both accelerators, both ray sources of the
trace kernel (per-workgroup regions and the pooled source of launches that run alone), several seeds.
usage: stress_trace_parity.py [rays per batch, default 400000] [seeds, default 2] [traversal bits to try, default 0,8;
2 = two rays per lane]"""
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
import numpy as np  # noqa: E402
import orc  # noqa: E402
import path_tracer_amd as pta  # noqa: E402
from common import ray_batch, scene_path  # noqa: E402
from test_gpu_trace_parity import check_batch  # noqa: E402

n_rays = int(sys.argv[1]) if len(sys.argv) > 1 else 400000
seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 2
modes = [int(m, 0) for m in sys.argv[3].split(",")] if len(sys.argv) > 3 else [0, 8]
SCENES = ["cornellbox.obj", "cornellbox-glass.obj", "cornellbox-alpha.obj", "cornellball-glass.obj", "cube.obj", "textured.obj",
          "smallsphere.obj", "teapot.obj", "testObj.obj", "bunny.obj"]
# the 210 k-triangle stand-in for sponza.obj (tools/make_sponza.py), written next to this run's temporary files
import tempfile  # noqa: E402
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tools"))
from make_sponza import build, write  # noqa: E402
tmp = tempfile.mkdtemp(prefix="pt_stress_")
standin = os.path.join(tmp, "sponza_standin.obj")
write(build(seed=7, detail=1.0), standin)
SCENES.append(standin)
total = 0
for obj in SCENES:
    for accel in (pta.BVH, pta.KD):
        t0 = time.time()
        path = obj if os.path.isabs(obj) else scene_path(obj)
        s = pta.Scene(0)
        assert s.load(path), obj
        s.pre_calc(accel)
        o = orc.Oracle()
        assert o.load_obj(path) == s.num_triangles()
        if accel == pta.KD:
            o.import_kd(s.get_tree())  # held node for node equal to the oracle's own builder by the CPU tests
        else:
            o.build(orc.BVH)
        pos = o.get_tris()[0]
        hits = 0
        for seed in range(seeds):
            rays = ray_batch(pos, n_random=n_rays * 3 // 4, n_vertex=n_rays // 12, seed=100 + seed)
            for mode in modes:
                s.set_option(pta.OPT_TRAVERSAL, s.get_option(pta.OPT_TRAVERSAL) & ~10 | mode)
                hits += check_batch(o, s, rays)
            total += len(rays)
        print("%-22s %-3s ok: %d rays x %d settings, %d hits, %.1f s" % (os.path.basename(obj), "kd" if accel == pta.KD else "bvh", seeds * len(rays), len(modes), hits,
                                                                      time.time() - t0), flush=True)
        s.close()
print("all bit-identical:", total, "rays")
