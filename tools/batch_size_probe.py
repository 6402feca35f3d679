#!/usr/bin/env python3
"""GPU experiment: closest-hit throughput of pt_trace_batch on first-bounce rays as a function of the batch size, i.e. of
the number of rays a lane walks per launch.  Separates the per-launch costs (staging the tree, filling and draining the
waves) from the per-ray cost.  usage: batch_size_probe.py [scene.obj] [bvh|kd]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import path_tracer_amd as pta  # noqa: E402
from common import load_gpu_scene  # noqa: E402

obj = sys.argv[1] if len(sys.argv) > 1 else "cornellbox.obj"
accel = pta.KD if (len(sys.argv) > 2 and sys.argv[2] == "kd") else pta.BVH
s = load_gpu_scene(obj, accel)
s.set_resolution(1024, 1024)
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
spp = 16
o = np.repeat(o, spp, axis=0); n = np.repeat(n, spp, axis=0)
u1, u2 = rng.random(len(o)), rng.random(len(o))
r, th = np.sqrt(u1), 2 * np.pi * u2
up = np.where(np.abs(n[:, 2:3]) < 0.999, np.array([[0, 0, 1.0]]), np.array([[1.0, 0, 0]]))
tg = np.cross(up, n); tg /= np.linalg.norm(tg, axis=1, keepdims=True)
bt = np.cross(n, tg)
d = tg * (r * np.cos(th))[:, None] + bt * (r * np.sin(th))[:, None] + n * np.sqrt(1 - u1)[:, None]
rays = np.concatenate([o + 1e-4 * n, d], axis=1).astype(np.float32)
rays = rays[rng.permutation(len(rays))]  # any prefix is a fair sample of the population
print(obj, "kd" if accel == pta.KD else "bvh", len(rays), "first-bounce rays")
for size in (1 << 17, 1 << 18, 1 << 19, 1 << 20, 1 << 21, 1 << 22, 1 << 23):
    if size > len(rays):
        break
    rr = rays[:size]
    s.intersect(rr[:4096])
    s.reset_counters()
    s.set_option(pta.OPT_PROFILE, 1)
    for _ in range(3):
        s.intersect(rr)
    ms, nl = s.timers()
    s.set_option(pta.OPT_PROFILE, 0)
    us = ms[pta.TIMER_TRACE] / nl[pta.TIMER_TRACE] * 1e3
    print("%9d rays: %8.1f us per launch, %6.0f Mrays/s, %5.1f rays per lane of the grid" % (size, us, size / us, size / (min((size + 255) // 256, 2048) * 256.0)), flush=True)
