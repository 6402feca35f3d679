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
def test_host_camera_model_and_pixel_centre_rays(gpu_device, accel):
    """All 512x384 pixel-centre camera rays of the default camera: the rays come from the library's HOST camera
    model (pt_camera_rays, the reference's eyeRay form) and are compared with the oracle's, then traced on the
    GPU.  The ray-generation KERNEL is checked in tests/test_gpu_raygen.py."""
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


def tie_heavy_mesh(seed=23):
    """Geometry built to produce exact ties and degenerate cases: a grid of coplanar triangles sharing every
    edge and vertex, each cell's triangles stored two or three times under different indices, a second
    coplanar layer shifted by half a cell (overlapping, same plane), zero-area and repeated-vertex
    triangles, a few slivers, and one far outlier that stretches the root box."""
    rng = np.random.default_rng(seed)
    tris = []
    n = 14
    xs = np.linspace(-1, 1, n + 1, dtype=np.float32)
    for i in range(n):
        for j in range(n):
            a = [xs[i], xs[j], 0.0]; b = [xs[i + 1], xs[j], 0.0]; c = [xs[i + 1], xs[j + 1], 0.0]; d = [xs[i], xs[j + 1], 0.0]
            for _ in range(int(rng.integers(1, 4))):          # duplicates: same triangles, different ids
                tris += [[a, b, c], [a, c, d]]
            if (i + j) % 3 == 0:                               # overlapping coplanar layer, shifted
                h = float(xs[1] - xs[0]) / 2
                tris.append([[a[0] + h, a[1] + h, 0.0], [b[0] + h, b[1] + h, 0.0], [c[0] + h, c[1] + h, 0.0]])
    for _ in range(40):                                        # zero area: repeated vertex / collinear
        p = (rng.random(3) * 2 - 1).tolist(); q = (rng.random(3) * 2 - 1).tolist()
        tris.append([p, p, q])
        tris.append([p, q, [(p[k] + q[k]) / 2 for k in range(3)]])
    for _ in range(40):                                        # slivers
        p = rng.random(3) * 2 - 1
        tris.append([p.tolist(), (p + [1.5, 1e-6, 0]).tolist(), (p + [0.7, 0, 1e-6]).tolist()])
    tris.append([[50, 50, 50], [51, 50, 50], [50, 51, 50]])
    # a second, tilted wall of shared-edge triangles behind the first
    for i in range(n):
        for j in range(n):
            z = lambda x, y: -0.5 + 0.25 * x
            a = [xs[i], xs[j]]; b = [xs[i + 1], xs[j]]; c = [xs[i + 1], xs[j + 1]]; d = [xs[i], xs[j + 1]]
            tris += [[a + [z(*a)], b + [z(*b)], c + [z(*c)]], [a + [z(*a)], c + [z(*c)], d + [z(*d)]]]
    pos = np.asarray(tris, np.float32).reshape(-1, 9)
    nrm = np.tile(np.array([0, 0, 1], np.float32), (pos.shape[0], 3)).reshape(-1, 9)
    return pos, nrm


@pytest.mark.parametrize("accel", [orc.BVH, orc.KD])
def test_exact_ties_resolve_to_the_oracle_winner(gpu_device, accel):
    """With duplicated and overlapping coplanar triangles most rays see several triangles at the identical
    distance; the reference keeps the first one its traversal visits (strict <, src/cs488.h:1677-1681,
    1350-1354).  The winner's index must be the oracle's for every ray, which pins the visiting order."""
    import path_tracer_amd as pta
    pos, nrm = tie_heavy_mesh()
    o = orc.Oracle()
    o.set_tris(pos, nrm)
    o.build(accel)
    s = pta.Scene(0)
    s.set_mesh(pos, nrm)
    s.pre_calc(accel)
    rng = np.random.default_rng(5)
    # rays through grid vertices, edge midpoints and cell interiors of the z = 0 layer, from both sides
    n = 14
    xs = np.linspace(-1, 1, 2 * n + 1)
    gx, gy = np.meshgrid(xs, xs)
    targets = np.stack([gx.ravel(), gy.ravel(), np.zeros(gx.size)], 1)
    targets = np.concatenate([targets, np.stack([rng.random(3000) * 2 - 1, rng.random(3000) * 2 - 1, np.zeros(3000)], 1)])
    rays = []
    for eye in ([0.1, -0.2, 2.0], [0.0, 0.0, 3.0], [0.3, 0.4, -2.5], [-1.7, 0.2, 1.1]):
        d = targets - np.asarray(eye)
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        rays.append(np.concatenate([np.broadcast_to(np.asarray(eye, np.float64), d.shape), d], 1))
    # axis-parallel rays straight down the grid lines
    down = np.concatenate([targets[:, :2], np.full((len(targets), 1), 1.5), np.zeros((len(targets), 2)), -np.ones((len(targets), 1))], 1)
    rays = np.concatenate(rays + [down]).astype(np.float32)
    ref = o.trace(rays)
    got = s.intersect(rays)
    assert np.array_equal(got["tri"], ref["tri"])
    assert np.array_equal(got["t"].view(np.uint32), ref["t"].view(np.uint32))
    hit = ref["tri"] >= 0
    assert np.array_equal(got["bary"][hit].view(np.uint32), ref["bary"][hit].view(np.uint32))
    # the test is only meaningful if ties are common: count rays whose winner shares its distance
    tied = sum(1 for i in rng.choice(np.nonzero(hit)[0], 300, replace=False) if len(o.tie_set(rays[i], ref["t"][i])) > 1)
    assert tied > 100, tied
    check_batch(o, s, ray_batch(pos, n_random=4000, n_vertex=800, seed=9))
