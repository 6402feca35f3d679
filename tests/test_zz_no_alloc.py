"""pt_trace_batch_static -- the closest-hit entry that allocates nothing (its staging is a file-scope
device array) -- against pt_trace_batch AND against the CPU oracle on the same rays: hit flag, triangle
index and the bits of t, beta and gamma equal, on two calls in a row, the second with a smaller input
(the array keeps the first call's contents)."""
import numpy as np
import pytest

import path_tracer_amd as pta
from common import DeviceArray, load_gpu_scene, load_oracle_scene, ray_batch

pytestmark = pytest.mark.gpu


def _device_call(s, rays, with_bary=True):
    n = rays.shape[0]
    d_rays = DeviceArray(rays)
    # poisoned outputs: every element must be written by the call
    d_tri = DeviceArray(np.full(n, -77, np.int32))
    d_t = DeviceArray(np.full(n, -5.0, np.float32))
    d_bary = DeviceArray(np.full((n, 2), -5.0, np.float32))
    s.intersect_static(d_rays.ptr.value, d_tri.ptr.value, d_t.ptr.value, d_bary.ptr.value if with_bary else None, n)
    s.synchronize()
    out = dict(tri=d_tri.download(), t=d_t.download(), bary=d_bary.download())
    for a in (d_rays, d_tri, d_t, d_bary):
        a.free()
    return out


def _same(a, b):
    assert np.array_equal(a["tri"], b["tri"])
    assert np.array_equal(a["t"].view(np.uint32), b["t"].view(np.uint32))
    assert np.array_equal(a["bary"].view(np.uint32), b["bary"].view(np.uint32))


@pytest.mark.parametrize("obj,accel", [("cornellbox.obj", pta.BVH), ("teapot.obj", pta.KD)])
def test_static_entry_equals_the_allocating_one(obj, accel):
    s = load_gpu_scene(obj, accel)
    o = load_oracle_scene(obj, accel, textures=False)  # the checker itself, not only the entry point it mirrors
    pos = s.get_mesh()[0]
    big = ray_batch(pos, n_random=40000, n_vertex=2000, seed=11)
    small = ray_batch(pos, n_random=700, n_vertex=40, seed=12)
    assert small.shape[0] < big.shape[0] <= pta.TRACE_STATIC_MAX_RAYS
    for rays in (big, small):  # the second call reuses the array the first one filled
        got = _device_call(s, rays)
        _same(got, s.intersect(rays))
        _same(got, o.trace(rays))
    # without the optional barycentric output: that array stays untouched
    got = _device_call(s, small, with_bary=False)
    ref = s.intersect(small)
    assert np.array_equal(got["tri"], ref["tri"]) and np.array_equal(got["t"].view(np.uint32), ref["t"].view(np.uint32))
    assert (got["bary"] == -5.0).all()
    s.close()


def test_static_entry_refuses_more_rays_than_its_array_holds():
    s = load_gpu_scene("cornellbox.obj", pta.BVH)
    n = pta.TRACE_STATIC_MAX_RAYS + 1
    d = DeviceArray(np.zeros(16, np.float32))  # never read: the size check comes first
    with pytest.raises(pta.PtError) as e:
        s.intersect_static(d.ptr.value, d.ptr.value, d.ptr.value, None, n)
    assert e.value.code == pta.ERR_INVALID_ARG and "131072" in str(e.value)
    d.free()
    s.close()
