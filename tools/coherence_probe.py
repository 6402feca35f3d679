#!/usr/bin/env python3
"""GPU experiment: how much faster does the trace kernel walk first-bounce rays when a workgroup's
rays are grouped by direction octant (and by the triangle they leave, or by the cell of a grid their origin lies in)?  Builds the level-1 ray
population of a view on the host (cosine-distributed directions about the flat normal of the
triangle each primary ray hits), then times pt_trace_batch's kernel for several orderings.
usage: coherence_probe.py [scene.obj | sponza] [resolution] [bvh|kd]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import path_tracer_amd as pta  # noqa: E402
from common import load_gpu_scene, scene_path  # noqa: E402

obj = sys.argv[1] if len(sys.argv) > 1 else "cornellbox.obj"
res = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
accel = pta.KD if (len(sys.argv) > 3 and sys.argv[3] == "kd") else pta.BVH
spp = 4
if obj == "sponza":  # the stand-in courtyard of bench.py, from its camera
    import bench
    from make_sponza import CAMERA
    s = pta.Scene(0)
    assert s.load(bench.sponza_path())
    s.set_camera(**CAMERA)
    s.pre_calc(accel)
else:
    s = load_gpu_scene(obj, accel)
s.set_resolution(res, res)
cam = s.camera_rays()
h = s.intersect(cam)
pos = s.get_mesh()[0].reshape(-1, 3, 3)
ok = h["tri"] >= 0
tri = h["tri"][ok]
o = cam[ok, :3] + h["t"][ok, None] * cam[ok, 3:]
n = np.cross(pos[tri, 1] - pos[tri, 0], pos[tri, 2] - pos[tri, 0])
n /= np.linalg.norm(n, axis=1, keepdims=True)
n[np.einsum("ij,ij->i", n, cam[ok, 3:]) > 0] *= -1
rng = np.random.default_rng(5)
o = np.repeat(o, spp, axis=0); n = np.repeat(n, spp, axis=0); tri = np.repeat(tri, spp)
u1, u2 = rng.random(len(o)), rng.random(len(o))
r, th = np.sqrt(u1), 2 * np.pi * u2
up = np.where(np.abs(n[:, 2:3]) < 0.999, np.array([[0, 0, 1.0]]), np.array([[1.0, 0, 0]]))
tg = np.cross(up, n); tg /= np.linalg.norm(tg, axis=1, keepdims=True)
bt = np.cross(n, tg)
d = tg * (r * np.cos(th))[:, None] + bt * (r * np.sin(th))[:, None] + n * np.sqrt(1 - u1)[:, None]
rays = np.concatenate([o + 1e-4 * n, d], axis=1).astype(np.float32)
N = len(rays)
octant = (d[:, 0] > 0).astype(np.int64) | ((d[:, 1] > 0).astype(np.int64) << 1) | ((d[:, 2] > 0).astype(np.int64) << 2)
block = np.arange(N) // 5120
# cell of a 32^3 grid over the origins' bounds, cells numbered along a Morton curve
cell = np.clip(((rays[:, :3] - rays[:, :3].min(0)) / np.maximum(np.ptp(rays[:, :3], axis=0), 1e-9) * 32).astype(np.int64), 0, 31)
morton = np.zeros(N, np.int64)
for bit in range(5):
    for axis in range(3):
        morton |= ((cell[:, axis] >> bit) & 1) << (3 * bit + axis)


def timed(order, label):
    rr = rays[order]
    s.intersect(rr[:1024])
    s.reset_counters()
    s.set_option(pta.OPT_PROFILE, 1)
    for _ in range(3):
        s.intersect(rr)
    ms, nl = s.timers()
    s.set_option(pta.OPT_PROFILE, 0)
    print("%-44s %8.1f us per launch, %.0f Mrays/s" % (label, ms[pta.TIMER_TRACE] / nl[pta.TIMER_TRACE] * 1e3, N / (ms[pta.TIMER_TRACE] / nl[pta.TIMER_TRACE]) * 1e-3), flush=True)


print(obj, "kd" if accel == pta.KD else "bvh", N, "rays")
timed(np.arange(N), "queue order (pixel order)")
timed(np.lexsort((octant, block)), "octant within 5120-ray regions")
timed(np.lexsort((tri, octant, block)), "octant, triangle within regions")
timed(np.lexsort((octant, tri, block)), "triangle, octant within regions")
timed(np.argsort(octant, kind="stable"), "octant, whole batch")
timed(np.lexsort((morton, block)), "origin cell (32^3, Morton) within regions")
timed(np.lexsort((octant, morton, block)), "origin cell, octant within regions")
timed(np.lexsort((octant, morton)), "origin cell, octant, whole batch")
timed(rng.permutation(N), "random order")

