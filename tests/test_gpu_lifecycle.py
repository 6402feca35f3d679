"""Contexts and groups come and go without leaking device memory or host threads: a serving process creates and
destroys scenes for as long as it lives."""
import ctypes
import os

import numpy as np
import pytest

import path_tracer_amd as pta
from common import load_gpu_scene, ray_batch

pytestmark = pytest.mark.gpu


def _free_device_bytes():
    hip = ctypes.CDLL("libamdhip64.so")
    free, total = ctypes.c_size_t(), ctypes.c_size_t()
    assert hip.hipMemGetInfo(ctypes.byref(free), ctypes.byref(total)) == 0
    return free.value


def _host_threads():
    for line in open("/proc/self/status"):
        if line.startswith("Threads:"):
            return int(line.split()[1])
    return 0


def _use(obj, accel, group):
    s = load_gpu_scene(obj, accel, env="probe_64.hdr" if obj.startswith("teapot") else None)  # the teapot has no light of its own
    s.set_resolution(256, 192)
    g = pta.Group([s]) if group else None
    for _ in range(3):
        (g or s).pathtrace(4)
    (g or s).synchronize()
    img = s.accumulation_buffer()
    assert np.isfinite(img).all() and img.max() > 0
    rays = ray_batch(s.get_mesh()[0], n_random=3000, n_vertex=100, seed=1)
    hits = s.intersect(rays)
    assert (hits["tri"] >= 0).any()
    if g is not None:
        g.close()
    s.close()


def test_contexts_and_groups_release_what_they_took(gpu_device):
    cases = [("cornellbox.obj", pta.BVH, False), ("teapot.obj", pta.KD, False), ("cornellbox-glass.obj", pta.KD, True),
             ("teapot.obj", pta.BVH, True)]
    for c in cases:  # first uses: code objects, the runtime's own pools and the library's static array stay
        _use(*c)
    free0, threads0 = _free_device_bytes(), _host_threads()
    for k in range(24):
        _use(*cases[k % len(cases)])
    free1, threads1 = _free_device_bytes(), _host_threads()
    assert free0 - free1 < (16 << 20), "device memory went down by %.1f MiB over 24 contexts" % ((free0 - free1) / 2**20)
    assert threads1 <= threads0 + 1, (threads0, threads1)
