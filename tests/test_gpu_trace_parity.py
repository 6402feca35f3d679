"""Closest-hit parity of the HIP trace kernels against the CPU oracle on identical ray batches.

Bar: hit/miss flag equal, triangle index equal (or a member of the oracle's exact-tie set when
several triangles return the identical distance), hit distance equal to 1e-5 relative; in
practice the distances are required to be bit-identical here and the looser bound is only the
documented ceiling.
"""
import numpy as np
import pytest

import orc
from common import load_gpu_scene, load_oracle_scene, ray_batch

pytestmark = pytest.mark.gpu

CASES = [
    ("cornellbox.obj", orc.BVH), ("cornellbox.obj", orc.KD),
    ("cornellbox-glass.obj", orc.BVH),
    ("cornellball-glass.obj", orc.BVH), ("cornellball-glass.obj", orc.KD),
    ("cube.obj", orc.BVH), ("cube.obj", orc.KD),
    ("textured.obj", orc.BVH), ("smallsphere.obj", orc.KD),
    ("teapot.obj", orc.BVH), ("teapot.obj", orc.KD),
]


def check_batch(o, s, rays):
    ref = o.trace(rays)
    got = s.intersect(rays)
    assert np.array_equal(ref["tri"] >= 0, got["tri"] >= 0), "hit/miss flags differ"
    hit = ref["tri"] >= 0
    rel = np.abs(got["t"][hit] - ref["t"][hit]) / np.maximum(np.abs(ref["t"][hit]), 1e-30)
    assert rel.size == 0 or rel.max() <= 1e-5
    assert np.array_equal(got["t"][hit].view(np.uint32), ref["t"][hit].view(np.uint32)), "t not bit-identical"
    assert np.all(got["t"][~hit] == np.float32(3.4028234663852886e38))
    diff = np.nonzero(got["tri"] != ref["tri"])[0]
    for i in diff:  # must be an exact tie of the oracle's winner
        ties = o.tie_set(rays[i], ref["t"][i])
        assert got["tri"][i] in ties, (i, got["tri"][i], ref["tri"][i], ties)
    assert len(diff) == 0, "winner differs inside a tie set: traversal order is not the oracle's"
    assert np.array_equal(got["bary"][hit].view(np.uint32), ref["bary"][hit].view(np.uint32))
    return int(hit.sum())


@pytest.mark.parametrize("obj,accel", CASES)
def test_ray_batch_parity(gpu_device, obj, accel):
    o = load_oracle_scene(obj, accel, textures=False)
    s = load_gpu_scene(obj, accel)
    pos = o.get_tris()[0]
    rays = ray_batch(pos)
    nh = check_batch(o, s, rays)
    assert nh > 100


@pytest.mark.parametrize("accel", [orc.BVH, orc.KD])
def test_camera_rays_parity(gpu_device, accel):
    """All 512x384 pixel-centre camera rays of the default camera."""
    o = load_oracle_scene("cornellbox.obj", accel)
    s = load_gpu_scene("cornellbox.obj", accel)
    s.set_resolution(512, 384)
    rays_ref, _ = o.camera_rays(512, 384, jitter=False)
    rays = s.camera_rays()
    assert np.array_equal(rays.view(np.uint32), rays_ref.view(np.uint32))
    check_batch(o, s, rays)


def test_single_triangle_and_empty_batch(gpu_device):
    import path_tracer_amd as pta
    s = pta.Scene(gpu_device)
    s.create_single_triangle()
    s.pre_calc(pta.BVH)
    o = orc.Oracle()
    o.single_triangle()
    o.build(orc.BVH)
    rays = np.array([[0, 0, 1.5, 0, 0, -1], [0, 0, 1.5, 0, 0, 1], [0.49, -0.49, 1, 0, 0, -1], [2, 2, 1, 0, 0, -1]], np.float32)
    check_batch(o, s, rays)
    out = s.intersect(np.zeros((0, 6), np.float32))
    assert out["tri"].shape == (0,)


def test_nan_and_degenerate_rays(gpu_device):
    """Rays with NaN, infinite and all-zero directions take the same branches as on the CPU."""
    o = load_oracle_scene("cornellbox.obj", orc.BVH)
    s = load_gpu_scene("cornellbox.obj", orc.BVH)
    rays = np.array([
        [0, 0, 1.5, np.nan, 0, -1], [0, 0, 1.5, 0, 0, 0], [0, 0, 1.5, np.inf, 0, -1],
        [np.nan, 0, 1.5, 0, 0, -1], [0, 0, 0, 0, 1, 0], [0, 0, 0, 0, -1, 0], [0, 0, 0, 1, 0, 0],
        [0.1, 0.2, 0.0, 1e-30, 1e-30, -1], [0, 0, 1e30, 0, 0, -1],
    ], np.float32)
    ref = o.trace(rays)
    got = s.intersect(rays)
    assert np.array_equal(ref["tri"], got["tri"])
    assert np.array_equal(ref["t"].view(np.uint32), got["t"].view(np.uint32))
