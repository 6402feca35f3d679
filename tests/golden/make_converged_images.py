#!/usr/bin/env python3
"""Writes tests/golden/oracle/converged.npz: for every scene of the benchmark configurations two renders of the CPU
oracle at 64 x 48 pixels and 4096 samples per pixel, seeds A and B (eight threads: one PCG32 stream per column strip,
so the images are reproducible).  tests/test_gpu_converged.py holds the GPU renderer's 4096-sample image against both:
at that sample count the noise is 1/32 of a single sample's, so a bias of half a percent in any surface model shows.

Run from the repository root after a deliberate change to the oracle:  python tests/golden/make_converged_images.py
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

import orc  # noqa: E402
from common import load_oracle_scene  # noqa: E402

OUT = os.path.join(HERE, "oracle", "converged.npz")
W, H, SPP, FRAMES, THREADS = 64, 48, 64, 64, 8
SEEDS = (orc.DEFAULT_SEED, 0x1234567887654321 | 1)
# (key, scene, accelerator, probe)
SCENES = [
    ("cornellbox", "cornellbox.obj", orc.BVH, None),
    ("cornellbox-glass", "cornellbox-glass.obj", orc.BVH, "probe_64.hdr"),
    ("cornellbox-metal", "cornellbox-metal.obj", orc.BVH, "probe_64.hdr"),
    ("cornellbox-alpha", "cornellbox-alpha.obj", orc.BVH, "probe_64.hdr"),
    ("cornellbox-demo", "cornellbox-demo.obj", orc.BVH, None),
    ("cornellball-glass", "cornellball-glass.obj", orc.BVH, None),
    ("teapot", "teapot.obj", orc.BVH, "probe_64.hdr"),
    ("teapot-glass", "teapot-glass.obj", orc.BVH, "probe_64.hdr"),
    ("teapot-metal", "teapot-metal.obj", orc.BVH, "probe_64.hdr"),
    ("testObj", "testObj.obj", orc.BVH, "probe_64.hdr"),
    ("testObj-glass", "testObj-glass.obj", orc.BVH, "probe_64.hdr"),
]


def render(obj, accel, env, seed):
    o = load_oracle_scene(obj, accel, env=env)
    o.reset(W, H)
    o.render(spp=SPP, frames=FRAMES, threads=THREADS, state=seed)
    img = o.image()
    o.close()
    return img


if __name__ == "__main__":
    out = {}
    for key, obj, accel, env in SCENES:
        for tag, seed in zip("ab", SEEDS):
            out["%s/%s" % (key, tag)] = render(obj, accel, env, seed)
        a, b = out[key + "/a"], out[key + "/b"]
        print("%-18s mean %.5f / %.5f  rmse(a,b) %.5f" % (key, a.mean(), b.mean(), float(np.sqrt(np.mean((a - b) ** 2)))), flush=True)
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")
