#!/usr/bin/env python3
"""GPU experiment (needs a library built with `make EXTRA=-DPT_TIMELINE`): when do the workgroups of one trace_kernel
launch start and end, and when does each of their waves run out of rays?  Shows how much of a launch is spent with
the chip partly empty.  usage: wg_timeline.py [scene.obj] [bvh|kd] [rays]"""
import ctypes
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
want = int(sys.argv[3]) if len(sys.argv) > 3 else 5800000
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
spp = max(1, (want + len(o) - 1) // len(o))
o = np.repeat(o, spp, axis=0); n = np.repeat(n, spp, axis=0)
u1, u2 = rng.random(len(o)), rng.random(len(o))
r, th = np.sqrt(u1), 2 * np.pi * u2
up = np.where(np.abs(n[:, 2:3]) < 0.999, np.array([[0, 0, 1.0]]), np.array([[1.0, 0, 0]]))
tg = np.cross(up, n); tg /= np.linalg.norm(tg, axis=1, keepdims=True)
bt = np.cross(n, tg)
d = tg * (r * np.cos(th))[:, None] + bt * (r * np.sin(th))[:, None] + n * np.sqrt(1 - u1)[:, None]
rays = np.concatenate([o + 1e-4 * n, d], axis=1).astype(np.float32)
rays = rays[rng.permutation(len(rays))][:want]
print(obj, "kd" if accel == pta.KD else "bvh", len(rays), "first-bounce rays", flush=True)

L = pta.lib()
for opt in sys.argv[4:]:
    k, v = opt.split("=")
    s.set_option(getattr(pta, k), int(v, 0))
for _ in range(2):
    s.intersect(rays)
s.reset_counters()
s.set_option(pta.OPT_PROFILE, 1)
s.intersect(rays)
ms, nl = s.timers()
s.set_option(pta.OPT_PROFILE, 0)
print("launch by events: %.1f us" % (ms[pta.TIMER_TRACE] / nl[pta.TIMER_TRACE] * 1e3))
tl = np.zeros(4096 * 8, np.uint64)
assert L.pt_debug_timeline(tl.ctypes.data_as(ctypes.c_void_p)) == 0
tl = tl.reshape(4096, 8)
if os.environ.get("TL_DUMP"):
    np.save(os.environ["TL_DUMP"], tl)
tl = tl[tl[:, 6] > 0]
t0 = tl[:, 0].astype(np.int64)
t1 = tl[:, 1].astype(np.int64)
base = t0.min()
lo32 = (t0 & 0xFFFFFFFF)
wave_end = ((tl[:, 2:6].astype(np.int64) - lo32[:, None]) & 0xFFFFFFFF) + t0[:, None]  # 32-bit stamps -> absolute
us = 0.01  # 100 MHz
start = (t0 - base) * us
end = (t1 - base) * us
wend = (wave_end - base) * us
total = end.max()
print("workgroups %d, rays per workgroup %d..%d, launch span %.1f us" % (len(tl), tl[:, 6].min(), tl[:, 6].max(), total))
pc = [0, 5, 25, 50, 75, 95, 100]
print("start  us percentiles", pc, np.percentile(start, pc).round(1))
print("end    us percentiles", pc, np.percentile(end, pc).round(1))
print("wave end us percentiles", pc, np.percentile(wend.ravel(), pc).round(1))
print("workgroup run time us ", pc, np.percentile(end - start, pc).round(1))
print("spread of the four wave ends inside a workgroup, us", pc, np.percentile(wend.max(1) - wend.min(1), pc).round(1))
# waves running over time
bins = np.linspace(0, total, 41)
running = [(np.sum((start[:, None] <= t) & (wend > t))) for t in bins[:-1]]
peak = max(running)
print("waves running at 40 instants (of %d at the peak):" % peak)
print(" ".join("%d" % (100 * x // peak) for x in running))
area = np.sum(wend - start[:, None])
print("wave-time used / (peak waves x span) = %.3f" % (area / (peak * total)))
# per XCD and CU
xcc = (tl[:, 7] >> np.uint64(32)).astype(np.int64) & 0xF
hw = tl[:, 7].astype(np.int64) & 0xFFFFFFFF
cu = (hw >> 8) & 0xF
se = (hw >> 13) & 0x7
for x in sorted(set(xcc.tolist())):
    m = xcc == x
    print("xcd %d: %4d workgroups, end median %.1f max %.1f" % (x, m.sum(), np.median(end[m]), end[m].max()))
key = xcc * 1000 + se * 100 + cu
per = {}
for k_, e in zip(key.tolist(), end.tolist()):
    per.setdefault(k_, []).append(e)
counts = np.array([len(v) for v in per.values()])
print("compute units seen %d, workgroups per CU min %d max %d" % (len(per), counts.min(), counts.max()))
s.close()
