"""Image parity of the HIP wavefront renderer against the CPU oracle at matched sample counts.

The two renderers draw different random numbers (the oracle follows the reference's single
PCG32 stream, the GPU uses a counter-based generator), so agreement is statistical: the RMSE
between two independent N-sample renders of the same estimator is sqrt(2) times the one-render
noise.  The bound used is 1.25 x the oracle-vs-oracle RMSE measured here with two different
seeds, and the mean luminance must agree to 1 % (2 % for the scenes whose image is dominated by
a few very bright texels).
"""
import os

import numpy as np
import pytest

import orc
from common import load_gpu_scene, load_oracle_scene, rmse

pytestmark = pytest.mark.gpu

W, H = 128, 96
SPP, FRAMES = 16, 8  # 128 samples per pixel


def render_pair(obj, accel, env, threads=8):
    o = load_oracle_scene(obj, accel, env=env)
    o.reset(W, H)
    o.render(spp=SPP, frames=FRAMES, threads=threads, state=orc.DEFAULT_SEED)
    a = o.image()
    o.reset(W, H)
    o.render(spp=SPP, frames=FRAMES, threads=threads, state=0x1234567887654321 | 1)
    b = o.image()
    s = load_gpu_scene(obj, accel, env=env)
    s.set_resolution(W, H)
    for _ in range(FRAMES):
        s.pathtrace(SPP)
    g = s.frame_buffer()
    assert s.total_samples() == SPP * FRAMES
    return a, b, g, s


@pytest.mark.parametrize("obj,accel,env,mean_tol", [
    ("cornellbox.obj", orc.BVH, None, 0.01),
    ("cornellbox.obj", orc.KD, None, 0.01),
    ("cornellbox-glass.obj", orc.BVH, "probe_64.hdr", 0.02),
    ("cornellbox-metal.obj", orc.BVH, "probe_64.hdr", 0.02),
    ("cornellbox-alpha.obj", orc.BVH, "probe_64.hdr", 0.02),
    ("cornellbox-demo.obj", orc.KD, None, 0.02),
    ("cornellball-glass.obj", orc.BVH, None, 0.02),
])
def test_image_rmse(gpu_device, obj, accel, env, mean_tol):
    a, b, g, s = render_pair(obj, accel, env)
    floor = rmse(a, b)
    assert floor > 0
    assert rmse(g, a) <= 1.25 * floor, (rmse(g, a), floor)
    assert rmse(g, b) <= 1.25 * floor, (rmse(g, b), floor)
    ref_mean = 0.5 * (a.mean() + b.mean())
    assert abs(g.mean() - ref_mean) <= mean_tol * ref_mean, (g.mean(), ref_mean)
    assert np.isfinite(g).all()


def test_teapot_textured_ibl(gpu_device):
    """Textured Lambertian teapot lit only by the probe (the teapot scene has no emitter)."""
    a, b, g, s = render_pair("teapot.obj", orc.BVH, "probe_64.hdr")
    floor = rmse(a, b)
    assert rmse(g, a) <= 1.25 * floor
    assert abs(g.mean() - a.mean()) <= 0.02 * a.mean()


def test_progressive_accumulation_and_reset(gpu_device):
    s = load_gpu_scene("cornellbox.obj", orc.BVH)
    s.set_resolution(64, 48)
    s.pathtrace(4)
    acc1 = s.accumulation_buffer()
    f1 = s.frame_buffer()
    assert np.allclose(f1, acc1 / 4.0, rtol=1e-6, atol=0)
    s.pathtrace(4)
    acc2 = s.accumulation_buffer()
    assert s.total_samples() == 8
    assert np.all(acc2 >= acc1 - 1e-6)
    assert np.allclose(s.frame_buffer(), acc2 / 8.0, rtol=1e-6, atol=0)
    s.reset()
    assert s.total_samples() == 0
    assert not s.accumulation_buffer().any() and not s.frame_buffer().any()


def test_batching_and_split_are_invisible(gpu_device):
    """The image depends on (seed, pixel, global sample index) only: tiny batches and a two-way
    sample split give the same accumulation buffer as one big batch, up to float summation order
    of the split halves."""
    import path_tracer_amd as pta
    s = load_gpu_scene("cornellbox-alpha.obj", orc.BVH)
    s.set_resolution(64, 48)
    s.pathtrace(6)
    whole = s.accumulation_buffer()
    t = load_gpu_scene("cornellbox-alpha.obj", orc.BVH)
    t.set_option(pta.OPT_MAX_BATCH, 1000)  # less than one sample of the image: pixel-range batches
    t.set_resolution(64, 48)
    t.pathtrace(6)
    small = t.accumulation_buffer()
    # forked paths add to a sample with float atomics, whose order is not fixed
    assert np.allclose(whole, small, rtol=2e-5, atol=1e-6)
    a = load_gpu_scene("cornellbox-alpha.obj", orc.BVH)
    b = load_gpu_scene("cornellbox-alpha.obj", orc.BVH)
    for x, (first, count) in ((a, (0, 3)), (b, (3, 3))):
        x.set_resolution(64, 48)
        x.pathtrace_split(6, first, count)
    both = a.accumulation_buffer() + b.accumulation_buffer()
    assert np.allclose(whole, both, rtol=2e-5, atol=1e-6)
    assert a.total_samples() == 6


def test_full_size_frame_properties(gpu_device):
    """The benchmark's frame (cornellbox, 1024 x 1024, 10 spp) is too large for the oracle to repeat in a
    test, so it is checked through properties that do not depend on size: the same samples in smaller
    batches and as a two-way sample split give the same accumulation buffer (bit for bit: no two paths of
    this scene share a sample), the ray count matches, frame = accumulation / total, and the mean radiance
    agrees with an oracle render of the same view at 192 x 192."""
    import path_tracer_amd as pta
    W = H = 1024
    s = load_gpu_scene("cornellbox.obj", orc.BVH)
    s.set_resolution(W, H)
    s.pathtrace(10)
    whole = s.accumulation_buffer()
    c_whole = s.counters()
    assert c_whole[pta.CNT_SAMPLES] == W * H * 10 and c_whole[pta.CNT_NAN_SAMPLES] == 0
    assert np.allclose(s.frame_buffer(), whole / 10.0, rtol=1e-6, atol=0)

    t = load_gpu_scene("cornellbox.obj", orc.BVH)
    t.set_option(pta.OPT_MAX_BATCH, 3 << 20)   # three samples of the image per batch
    t.set_resolution(W, H)
    t.pathtrace(10)
    assert np.array_equal(t.accumulation_buffer(), whole)
    assert t.counters()[pta.CNT_RAYS] == c_whole[pta.CNT_RAYS]

    a = load_gpu_scene("cornellbox.obj", orc.BVH)
    b = load_gpu_scene("cornellbox.obj", orc.BVH)
    for x, (first, count) in ((a, (0, 5)), (b, (5, 5))):
        x.set_resolution(W, H)
        x.pathtrace_split(10, first, count)
    # each half is summed in sample order from zero, so only the final addition differs in rounding
    assert np.allclose(a.accumulation_buffer() + b.accumulation_buffer(), whole, rtol=1e-6, atol=1e-7)

    o = load_oracle_scene("cornellbox.obj", orc.BVH)
    o.reset(192, 192)
    o.render(spp=40, frames=1, threads=8)
    ref = o.image()
    assert abs(whole.mean() / 10.0 - ref.mean()) < 0.01 * ref.mean()


@pytest.mark.timeout(900)
def test_sample_split_into_external_buffers_with_frames_in_flight(gpu_device):
    """The multi-GPU frame loop on one GPU (tests/check_external_accum.py: two contexts play ranks 0 and 1
    of a sample split into caller-owned torch buffers, with two frames in flight each); run as a child
    process because PyTorch has to open the GPU before this library does in a process that uses both."""
    import subprocess
    import sys
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "check_external_accum.py")
    r = subprocess.run([sys.executable, script], capture_output=True, text=True, timeout=850)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "external accumulation ok" in r.stdout


@pytest.mark.parametrize("obj,shot,roughness", [
    ("cornellbox.obj", "cornellbox_15000spp.png", None),
    ("cornellbox.obj", "cornellbox_19720spp.png", None),
    ("cornellbox-glass.obj", "cornellbox-glass_7200spp.png", None),
    ("cornellbox-metal.obj", "cornellbox_roughness_0.1.png", 0.1),
    ("cornellbox-metal.obj", "cornellbox_roughness_0.3.png", 0.3),
    ("cornellbox-metal.obj", "cornellbox_roughness_0.5.png", 0.5),
    ("cornellbox-alpha.obj", "cornellbox-alpha_purple.png", None),   # alpha_blending/alpha_purple.png: the file as shipped
    ("cornellbox-demo.obj", "cornellbox-demo_best.png", None),       # project_demo/best.png: metal wall, glass box, all shading kinds
])
def test_gpu_render_against_the_reference_screenshots(gpu_device, obj, shot, roughness):
    """The GPU path against the reference's own output, without the oracle in between: 2000 spp at the
    screenshots' 512 x 384 (project_demo/path_tracing, area_lights, microfacets with the metal's Ro set to the
    value in the file name), quantised like Image::save.  The image mean within 0.8 % (measured: 0.03 %, glass
    0.4 %) and the 16 x 16 block means within 0.003 RMS (measured 0.0007-0.0013); per pixel the RMSE is about
    0.01-0.017, the screenshots' own noise and 8-bit steps."""
    from PIL import Image
    from common import GOLDEN
    import path_tracer_amd as pta
    ref = np.asarray(Image.open(os.path.join(GOLDEN, "screenshots", shot)).convert("RGB"), np.float32)[::-1] / 255.0
    s = load_gpu_scene(obj, orc.BVH)
    if roughness is not None:
        mats = s.materials()
        metals = [m for m in mats if m.type == pta.METAL]
        assert len(metals) == 1
        metals[0].roughness = roughness
        s.set_materials(mats)
        s.pre_calc(pta.BVH)
    s.set_resolution(512, 384)
    for _ in range(200):
        s.pathtrace(10)
    mine = np.floor(np.clip(s.frame_buffer(), 0, 1) * 255.0) / 255.0

    def blocks(img):
        return img.reshape(384 // 16, 16, 512 // 16, 16, 3).mean((1, 3))

    assert abs(mine.mean() - ref.mean()) < 0.008 * ref.mean(), (mine.mean(), ref.mean())
    block_err = float(np.sqrt(np.mean((blocks(mine) - blocks(ref)) ** 2)))
    assert block_err < (0.004 if "alpha" in shot else 0.003), block_err   # the alpha screenshot is the noisiest
    assert float(np.sqrt(np.mean((mine - ref) ** 2))) < 0.03


def test_small_and_odd_shapes(gpu_device):
    """Sizes and settings at the edges: one pixel, one sample, a prime-sized image, maxLevel 0 and 62, a
    resolution change on a live context, a batch limit below one run of 64 samples, and a view that sees
    nothing.  Every result must be finite, deterministic and independent of the batch limit."""
    import path_tracer_amd as pta
    s = load_gpu_scene("cornellbox.obj", orc.BVH)
    for w, h, spp in ((1, 1, 1), (1, 1, 7), (3, 2, 1), (61, 37, 3), (64, 1, 2)):
        s.set_resolution(w, h)
        s.pathtrace(spp)
        a = s.accumulation_buffer()
        assert a.shape == (h, w, 3) and np.isfinite(a).all() and s.total_samples() == spp
        t = load_gpu_scene("cornellbox.obj", orc.BVH)
        t.set_option(pta.OPT_MAX_BATCH, 5)
        t.set_resolution(w, h)
        t.pathtrace(spp)
        assert np.array_equal(t.accumulation_buffer(), a), (w, h, spp)
        assert np.allclose(s.frame_buffer(), a / spp, rtol=1e-6)
    for level in (0, 1, 62):
        s.set_option(pta.OPT_MAX_LEVEL, level)
        s.set_resolution(32, 24)
        s.pathtrace(2)
        img = s.accumulation_buffer()
        assert np.isfinite(img).all()
        if level == 0:
            # only the light itself is visible: a hit at level 0 adds Ke, its bounce ends at level 1 > maxLevel
            assert img.max() > 0 and (img == 0).mean() > 0.5
    s.set_option(pta.OPT_MAX_LEVEL, 15)
    # looking away from the scene: every camera ray misses, no environment -> black, and nothing is traced twice
    s.set_camera(eye=(0, 0, 5), lookat=(0, 0, 10))
    s.set_resolution(16, 16)
    s.reset_counters()
    s.pathtrace(4)
    assert not s.accumulation_buffer().any()
    c = s.counters()
    assert c[pta.CNT_RAYS] == 16 * 16 * 4 and c[pta.CNT_HITS] == 0


def test_translucent_scenes_render_the_same_bits_every_time(gpu_device):
    """Where paths fork (Op < 1, src/cs488.h:2038-2044) several paths add to one camera sample.  Their sums are kept in
    fixed point, so the order in which they arrive cannot show: two renders give the same bits, as the reference does
    with one thread, and so does the same frame cut into smaller batches."""
    import path_tracer_amd as pta

    def render(max_batch=None, streams=None):
        s = load_gpu_scene("cornellbox-alpha.obj", orc.KD, env="probe_64.hdr")
        if max_batch:
            s.set_option(pta.OPT_MAX_BATCH, max_batch)
        if streams:
            s.set_option(pta.OPT_STREAMS, streams)
        s.set_resolution(160, 120)
        for _ in range(3):
            s.pathtrace(6)
        img, acc, c = s.frame_buffer(), s.accumulation_buffer(), s.counters()
        s.close()
        return img, acc, c

    a_img, a_acc, a_c = render()
    b_img, b_acc, b_c = render()
    assert np.array_equal(a_img, b_img) and np.array_equal(a_acc, b_acc)
    assert a_c[pta.CNT_RAYS] == b_c[pta.CNT_RAYS] and a_c[pta.CNT_NAN_SAMPLES] == 0
    c_img, c_acc, c_c = render(max_batch=160 * 120 * 2, streams=1)
    assert np.array_equal(a_acc, c_acc) and c_c[pta.CNT_RAYS] == a_c[pta.CNT_RAYS]
    assert np.isfinite(a_img).all() and a_img.mean() > 0.05
