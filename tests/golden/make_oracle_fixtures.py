#!/usr/bin/env python3
"""Freezes the CPU oracle's outputs: writes tests/golden/oracle/fixtures.json from oracle/pt_oracle.cpp as it is now.

Every GPU parity target in the suite is recomputed from the oracle when a test runs, so an edit to the oracle would
silently move the bar; tests/test_oracle_frozen.py recomputes what is recorded here and compares.  Recorded, per
scene and accelerator: SHA-256 of the closest-hit results (triangle index, t bits, beta and gamma bits) for the
suite's standard ray batch (common.ray_batch) and for the 512 x 384 pixel-centre camera rays, and of the tree
(children, leaf ranges, box bits, leaf contents); the first 1024 outputs of the PCG32 restatement (hash, and the first
eight values in clear); the hash of a single-thread 64 x 48 frame of 8 samples per pixel, which pins the whole
shading path including the order of its random draws.  The frame hashes go through the C library's sinf / cosf / powf /
acosf and hold for the C library of this image (the GPU box runs the same image).

Run from the repository root after a deliberate change to the oracle:  python tests/golden/make_oracle_fixtures.py
"""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

import orc  # noqa: E402
from common import load_oracle_scene, ray_batch  # noqa: E402

OUT = os.path.join(HERE, "oracle", "fixtures.json")

# (scene, accelerators, environment for the frame or None, render a frame?)
SCENES = [
    ("cornellbox.obj", (orc.BVH, orc.KD), None, True),
    ("cornellbox-glass.obj", (orc.BVH,), "probe_64.hdr", True),
    ("cornellbox-metal.obj", (orc.BVH,), "probe_64.hdr", True),
    ("cornellbox-alpha.obj", (orc.BVH,), "probe_64.hdr", True),
    ("cornellball-glass.obj", (orc.BVH, orc.KD), None, False),
    ("textured.obj", (orc.BVH,), None, False),
    ("teapot.obj", (orc.BVH, orc.KD), "probe_64.hdr", True),
    ("testObj.obj", (orc.BVH,), None, False),
]


def sha(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        a = np.ascontiguousarray(a)
        h.update(str(a.dtype).encode() + str(a.shape).encode())
        h.update(a.tobytes())
    return h.hexdigest()


def hits_hash(res):
    return sha(res["tri"], res["t"].view(np.uint32), res["bary"].view(np.uint32))


def scene_record(obj, accel, env, frame):
    o = load_oracle_scene(obj, accel, env=env)
    rec = {}
    tree = o.kd_dump() if accel == orc.KD else o.bvh_dump()
    keys = [k for k in ("is_leaf", "left", "right", "first", "count") if k in tree]
    rec["tree_nodes"] = int(len(tree["left"]))
    rec["tree"] = sha(*[tree[k] for k in keys], np.asarray(tree["box"], np.float32).view(np.uint32), tree["refs"])
    rays = ray_batch(o.get_tris()[0])
    res = o.trace(rays)
    rec["ray_batch"] = {"n": int(len(rays)), "hits": int((res["tri"] >= 0).sum()), "sha256": hits_hash(res)}
    cam, _ = o.camera_rays(512, 384, jitter=False)
    res = o.trace(cam)
    rec["camera_512x384"] = {"hits": int((res["tri"] >= 0).sum()), "sha256": hits_hash(res)}
    if frame:
        o.reset(64, 48)
        total, state, cnt = o.render(spp=8, frames=1, threads=1, state=orc.DEFAULT_SEED)
        img = o.image()
        rec["frame_64x48x8"] = {"sha256": sha(img.view(np.uint32)), "rays": int(cnt[0]), "rng_state_after": "%016x" % state,
                                "mean": float(np.float64(img.mean()))}
    o.close()
    return rec


def compute():
    out = {"scenes": {}}
    u32, f32, _ = orc.pcg32(1024)
    out["pcg32_1024"] = {"u32_sha256": sha(u32), "f32_sha256": sha(f32.view(np.uint32)), "first_u32": [int(x) for x in u32[:8]]}
    for obj, accels, env, frame in SCENES:
        for accel in accels:
            out["scenes"]["%s/%s" % (obj, "kd" if accel == orc.KD else "bvh")] = scene_record(obj, accel, env, frame)
    return out


if __name__ == "__main__":
    data = compute()
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    with open(OUT, "w") as f:
        json.dump(data, f, indent=1, sort_keys=True)
    print("wrote %s: %d scene records" % (OUT, len(data["scenes"])))
