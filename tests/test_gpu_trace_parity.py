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


def bounce_rays(s, rays, hits, rng, spp=2):
    """The next-level ray population of a batch: from every hit point (interpolated from the
    barycentrics, offset along the flat normal) a few cosine-distributed directions, plus the
    straight-through continuation an alpha fork would trace."""
    pos = s.get_mesh()[0].reshape(-1, 3, 3)
    ok = hits["tri"] >= 0
    tri = hits["tri"][ok]
    b, g = hits["bary"][ok, 0:1], hits["bary"][ok, 1:2]
    p = (1 - b - g) * pos[tri, 0] + b * pos[tri, 1] + g * pos[tri, 2]
    n = np.cross(pos[tri, 1] - pos[tri, 0], pos[tri, 2] - pos[tri, 0])
    n /= np.maximum(np.linalg.norm(n, axis=1, keepdims=True), 1e-30)
    n[np.einsum("ij,ij->i", n, rays[ok, 3:]) > 0] *= -1
    p = np.repeat(p, spp, axis=0); nn = np.repeat(n, spp, axis=0)
    u1, u2 = rng.random(len(p)), rng.random(len(p))
    r, th = np.sqrt(u1), 2 * np.pi * u2
    up = np.where(np.abs(nn[:, 2:3]) < 0.999, np.array([[0, 0, 1.0]]), np.array([[1.0, 0, 0]]))
    tg = np.cross(up, nn); tg /= np.linalg.norm(tg, axis=1, keepdims=True)
    bt = np.cross(nn, tg)
    d = tg * (r * np.cos(th))[:, None] + bt * (r * np.sin(th))[:, None] + nn * np.sqrt(1 - u1)[:, None]
    bounce = np.concatenate([p + 1e-4 * nn, d], axis=1)
    through = np.concatenate([(1 - b - g) * pos[tri, 0] + b * pos[tri, 1] + g * pos[tri, 2] - 1e-4 * n, rays[ok, 3:]], axis=1)
    return np.concatenate([bounce, through]).astype(np.float32)


@pytest.mark.parametrize("obj,accel", [("cornellbox.obj", orc.BVH), ("cornellbox.obj", orc.KD), ("cornellball-glass.obj", orc.BVH),
                                        ("teapot.obj", orc.BVH), ("teapot.obj", orc.KD)])
def test_bounce_populations_parity(gpu_device, obj, accel):
    """Three generations of secondary rays as the renderer produces them (origins on surfaces, offset by
    epsilon, incoherent directions): several hundred thousand rays per scene, every one bit-identical."""
    o = load_oracle_scene(obj, accel, textures=False)
    s = load_gpu_scene(obj, accel)
    s.set_resolution(256, 192)
    rng = np.random.default_rng(17)
    rays = s.camera_rays()
    total = 0
    for _ in range(3):
        hits = s.intersect(rays)
        check_batch(o, s, rays)
        total += len(rays)
        rays = bounce_rays(s, rays, hits, rng)
        if len(rays) > 150000:
            rays = rays[rng.permutation(len(rays))[:150000]]
        if len(rays) == 0:
            break
    assert total > 50000
