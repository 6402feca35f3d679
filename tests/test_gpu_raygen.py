"""raygen_kernel against Scene::generateRay (src/cs488.h:1825-1839), bit for bit.

The kernel's random numbers are counter-based (the reference draws from one global PCG32 stream), so the
test restates the generator in numpy -- squares32(counter, key) with key = mix(seed, global sample index)
and counter = pixel << 32 | path id << 8 | dimension -- and then evaluates the reference's expressions in
binary32 with numpy, one rounding per operation, in the reference's order: r1 is drawn before r2, the
jitter (r * 2 - 1) spans two pixels around the pixel's corner, U is scaled by aspect * film, the direction
is normalize(pixelPos - eye) with the dot product summed left to right from zero.
"""
import ctypes

import numpy as np
import pytest

import path_tracer_amd as pta
from common import load_gpu_scene

pytestmark = pytest.mark.gpu

U64 = np.uint64
f32 = np.float32
_libm = ctypes.CDLL("libm.so.6")
_libm.tanf.restype = ctypes.c_float
_libm.tanf.argtypes = [ctypes.c_float]
tanf = _libm.tanf


def mix64(z):
    z = z + U64(0x9E3779B97F4A7C15)
    z = (z ^ (z >> U64(30))) * U64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> U64(27))) * U64(0x94D049BB133111EB)
    return z ^ (z >> U64(31))


def squares32(ctr, key):
    x = ctr * key
    y = x
    z = y + key
    rot = lambda v: (v >> U64(32)) | (v << U64(32))
    x = rot(x * x + y)
    x = rot(x * x + z)
    x = rot(x * x + y)
    return ((x * x + z) >> U64(32)).astype(np.uint32)


def uniform(pixel, sample, dim, seed):
    """The kernel's draw number `dim` of camera path 1 of (pixel, global sample index): float(double(u32) / 2^32),
    the conversion of the reference's PCG32::rand (src/cs488.h:140-142)."""
    with np.errstate(over="ignore"):
        key = mix64(U64(seed) ^ mix64(U64(sample))) | U64(1)
        ctr = (pixel.astype(U64) << U64(32)) | U64(1 << 8) | U64(dim)
        u = squares32(ctr, key)
    return (u.astype(np.float64) / 4294967296.0).astype(f32)


def normalize(v):
    # src/linalg.h:501-504: dot = ((0 + x*x) + y*y) + z*z, then every component divided by its square root
    d = (f32(0) + v[..., 0] * v[..., 0]) + v[..., 1] * v[..., 1]
    d = d + v[..., 2] * v[..., 2]
    return v / np.sqrt(d)[..., None]


def cross(a, b):
    return np.stack([a[..., 1] * b[..., 2] - a[..., 2] * b[..., 1], a[..., 2] * b[..., 0] - a[..., 0] * b[..., 2],
                     a[..., 0] * b[..., 1] - a[..., 1] * b[..., 0]], -1)


def generate_rays(w, h, sample, seed, eye, lookat, up, fov_deg, film):
    eye, lookat, up = (np.asarray(v, f32) for v in (eye, lookat, up))
    view = normalize(lookat - eye)                      # src/cs488.h:2284
    wd = normalize(-view)                               # :1827
    ud = normalize(cross(normalize(up), wd))            # :1828 (globalUp is a unit vector in the reference)
    vd = cross(wd, ud)                                  # :1829
    aspect = f32(f32(w) / f32(h))                       # :52
    half_angle = f32(fov_deg) * (f32(3.14159265358979) / f32(180.0)) * f32(0.5)
    dist = f32(film) / (f32(2.0) * f32(tanf(float(half_angle))))  # :79, with the C library's tanf like the host code
    pixel = np.arange(w * h, dtype=np.int64)
    px, py = (pixel % w).astype(f32), (pixel // w).astype(f32)
    r1 = uniform(pixel, sample, 0, seed)                # drawn first (:1833)
    r2 = uniform(pixel, sample, 1, seed)                # drawn second (:1834)
    U = (px + (r1 * f32(2) - f32(1))) / f32(w) - f32(0.5)
    V = (py + (r2 * f32(2) - f32(1))) / f32(h) - f32(0.5)
    a = (aspect * f32(film)) * U                        # :1836
    b = f32(film) * V
    pix = ((eye[None, :] + a[:, None] * ud[None, :]) + b[:, None] * vd[None, :]) - dist * wd[None, :]
    d = normalize(pix - eye[None, :])
    return np.concatenate([np.broadcast_to(eye, d.shape), d], 1).astype(f32), r1, r2


CAMERAS = [
    dict(),  # the reference's default camera (src/cs488.h:74-87)
    dict(eye=(-2.0, 0.05, 0.15), lookat=(1.0, -0.1, -0.05), up=(0.0, 1.0, 0.0), fov=60.0, film=0.05),
]


@pytest.mark.parametrize("cam", CAMERAS)
@pytest.mark.parametrize("shape", [(512, 384), (97, 61), (1024, 1024)])
def test_raygen_kernel_is_generate_ray(gpu_device, shape, cam):
    w, h = shape
    s = load_gpu_scene("cornellbox.obj", pta.BVH)
    full = dict(eye=(0, 0, 1.5), lookat=(0, 0, 0), up=(0, 1, 0), fov=45.0, film=0.032)
    full.update(cam)
    s.set_camera(**full)
    s.set_resolution(w, h)
    seed = s.get_option(pta.OPT_SEED)
    for sample in (0, 7, 12345):
        got = s.camera_samples(sample)
        ref, r1, r2 = generate_rays(w, h, sample, seed, full["eye"], full["lookat"], full["up"], full["fov"], full["film"])
        assert np.array_equal(got[:, :3].view(np.uint32), ref[:, :3].view(np.uint32)), "origins"
        same = np.all(got[:, 3:].view(np.uint32) == ref[:, 3:].view(np.uint32), axis=1)
        assert same.all(), (int((~same).sum()), got[~same][:3], ref[~same][:3])
        # the jitter really spans (-1, 1) pixels and the two draws differ
        assert r1.min() < 0.01 and r1.max() > 0.99 and not np.array_equal(r1, r2)
    # a different seed gives different rays, the same seed the same ones
    again = s.camera_samples(7)
    s.set_option(pta.OPT_SEED, 12345)
    other = s.camera_samples(7)
    s.set_option(pta.OPT_SEED, seed)
    assert np.array_equal(again, s.camera_samples(7)) and not np.array_equal(again, other)
    s.close()
