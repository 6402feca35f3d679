"""The GPU renderer at 4096 samples per pixel against committed converged renders of the CPU oracle
(tests/golden/oracle/converged.npz, two seeds per scene, written by tests/golden/make_converged_images.py): every
scene of the benchmark configurations -- Lambertian with area light, glass, GGX metal, translucent continuation,
textured teapot variants and testObj under the probe.  At this sample count the image noise is 1/32 of a single
sample's, so the gates are tight: the RMSE against the two oracle images (root of the mean of the two squares: a single
render's outlier pixel -- the probe's sun through glass -- sits in one of the two references and in the floor alike)
within 1.25 x the RMSE between the two, mean radiance within half a percent.  BVH and kd-tree give the same image bit
for bit (same hits), so each scene is rendered with one of them."""
import os

import numpy as np
import pytest

import orc
from common import GOLDEN, load_gpu_scene, rmse

pytestmark = pytest.mark.gpu

REF = np.load(os.path.join(GOLDEN, "oracle", "converged.npz"))
SCENES = [
    ("cornellbox", "cornellbox.obj", orc.BVH, None),
    ("cornellbox", "cornellbox.obj", orc.KD, None),
    ("cornellbox-glass", "cornellbox-glass.obj", orc.BVH, "probe_64.hdr"),
    ("cornellbox-metal", "cornellbox-metal.obj", orc.KD, "probe_64.hdr"),
    ("cornellbox-alpha", "cornellbox-alpha.obj", orc.BVH, "probe_64.hdr"),
    ("cornellbox-demo", "cornellbox-demo.obj", orc.BVH, None),
    ("cornellball-glass", "cornellball-glass.obj", orc.KD, None),
    ("teapot", "teapot.obj", orc.BVH, "probe_64.hdr"),
    ("teapot-glass", "teapot-glass.obj", orc.KD, "probe_64.hdr"),
    ("teapot-metal", "teapot-metal.obj", orc.BVH, "probe_64.hdr"),
    ("testObj", "testObj.obj", orc.BVH, "probe_64.hdr"),
    ("testObj-glass", "testObj-glass.obj", orc.BVH, "probe_64.hdr"),
]


@pytest.mark.parametrize("key,obj,accel,env", SCENES, ids=["%s-%s" % (s[0], "kd" if s[2] == orc.KD else "bvh") for s in SCENES])
def test_converged_image_against_the_committed_oracle_renders(gpu_device, key, obj, accel, env):
    a, b = REF[key + "/a"], REF[key + "/b"]
    h, w = a.shape[:2]
    s = load_gpu_scene(obj, accel, env=env)
    s.set_resolution(w, h)
    for _ in range(64):
        s.pathtrace(64)
    g = s.frame_buffer()
    assert s.total_samples() == 4096 and s.counters()[4] == 0  # no sample dropped as NaN
    s.close()
    floor = rmse(a, b)
    both = float(np.sqrt(0.5 * (rmse(g, a) ** 2 + rmse(g, b) ** 2)))
    assert both <= 1.25 * floor, (rmse(g, a), rmse(g, b), floor)
    ref_mean = 0.5 * (float(a.mean()) + float(b.mean()))
    tol = max(0.005, 3.0 * abs(float(a.mean()) - float(b.mean())) / ref_mean)
    assert abs(float(g.mean()) - ref_mean) <= tol * ref_mean, (float(g.mean()), ref_mean, tol)
