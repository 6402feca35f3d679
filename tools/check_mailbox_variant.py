import sys, os, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import orc, path_tracer_amd as pta
from common import load_gpu_scene, load_oracle_scene, ray_batch
for obj in ("cornellbox.obj", "cornellbox-glass.obj"):
    o = load_oracle_scene(obj, orc.KD, textures=False)
    rays = ray_batch(o.get_tris()[0], n_random=200000, n_vertex=2000, seed=3)
    ref = o.trace(rays)
    s = load_gpu_scene(obj, pta.KD)
    got = s.intersect(rays)
    ok = np.array_equal(got["tri"], ref["tri"]) and np.array_equal(got["t"].view(np.uint32), ref["t"].view(np.uint32)) and np.array_equal(got["bary"].view(np.uint32), ref["bary"].view(np.uint32))
    s.set_option(pta.OPT_COUNT_TESTS, 1); s.reset_counters(); s.intersect(rays); c = s.counters()
    print(obj, "parity", ok, "tri tests/ray %.2f" % (c[pta.CNT_TRI_TESTS] / c[pta.CNT_RAYS]))
    assert ok
