"""GPU tests of the trace-kernel settings (all must give bit-identical hits), the exactness of the
reciprocal-form division they rely on, the procedural courtyard scene, and the CS488 command line."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

import orc
import path_tracer_amd as pta
from common import ROOT, load_gpu_scene, load_oracle_scene, ray_batch, scene_path

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("obj,accel", [("cornellbox.obj", pta.BVH), ("cornellball-glass.obj", pta.KD), ("teapot.obj", pta.BVH)])
def test_every_traversal_setting_gives_the_oracle_hits(gpu_device, obj, accel):
    o = load_oracle_scene(obj, accel, textures=False)
    rays = ray_batch(o.get_tris()[0], n_random=6000, n_vertex=600, seed=11)
    # add rays outside the range the reciprocal form is trusted for: they must take the division form
    extra = np.array([[0, 0, 1.5, 0, 0, -1], [0, 0, 1.5, 1e-12, 0.3, -1], [1e-20, 0, 1.5, 0.1, 0.2, -1],
                      [0, 0, 1.5, 1e12, 3.0, -1e12], [0.1, 0.1, 2e15, 0.01, 0.02, -1]], np.float32)
    rays = np.concatenate([rays, extra])
    ref = o.trace(rays)
    s = load_gpu_scene(obj, accel)
    for mode in (0x00, 0x01, 0x04, 0x05, 0x08, 0x09, 0x21, 0x41, 0x49, 0xF1, 0x2041, 0x0841, 0x8001, 0x8009, 0x8841, 0x03, 0x0b, 0x0843, 0x4040):
        s.set_option(pta.OPT_TRAVERSAL, mode)
        got = s.intersect(rays)
        assert np.array_equal(got["tri"], ref["tri"]), hex(mode)
        assert np.array_equal(got["t"].view(np.uint32), ref["t"].view(np.uint32)), hex(mode)
        assert np.array_equal(got["bary"].view(np.uint32), ref["bary"].view(np.uint32)), hex(mode)
    assert s.get_option(pta.OPT_TRAVERSAL) == 0x4040


@pytest.mark.parametrize("accel", [pta.BVH, pta.KD])
def test_pooled_and_private_ray_sources_on_ragged_batches(gpu_device, accel):
    """pt_trace_batch lets the workgroups draw chunks of 64 rays from shared pools; with bit 3 of the traversal
    setting every workgroup keeps to its own region.  Batch sizes around the chunk, the wave, the workgroup and the
    region count (one region per 256 rays up to the resident grid, then longer regions with a ragged last one)."""
    o = load_oracle_scene("teapot.obj", accel, textures=False)
    rays = ray_batch(o.get_tris()[0], n_random=600000, n_vertex=600, seed=17)
    ref = o.trace(rays)
    s = load_gpu_scene("teapot.obj", accel)
    for mode in (0x01, 0x09, 0x03):  # 0x03: two rays per lane, private regions
        s.set_option(pta.OPT_TRAVERSAL, mode)
        for n in (1, 63, 64, 65, 255, 257, 4097, 70001, 524288 + 77, len(rays)):
            got = s.intersect(rays[:n])
            assert np.array_equal(got["tri"], ref["tri"][:n]), (hex(mode), n)
            assert np.array_equal(got["t"].view(np.uint32), ref["t"][:n].view(np.uint32)), (hex(mode), n)
            assert np.array_equal(got["bary"].view(np.uint32), ref["bary"][:n].view(np.uint32)), (hex(mode), n)
    s.close()


def test_traversal_settings_give_the_same_image(gpu_device):
    imgs = []
    for mode, streams in ((0x00, 2), (0x41, 2), (0x2041, 2), (0x01, 1), (0x09, 1)):  # one stream: pooled unless bit 3
        s = load_gpu_scene("cornellbox.obj", pta.BVH)
        s.set_option(pta.OPT_STREAMS, streams)
        s.set_option(pta.OPT_TRAVERSAL, mode)
        s.set_resolution(96, 72)
        s.pathtrace(8)
        imgs.append(s.accumulation_buffer())
    for im in imgs[1:]:
        assert np.array_equal(imgs[0], im)  # same hits, same random numbers, no forks in this scene


def test_walk_forms_of_a_tree_in_memory_give_the_same_image(gpu_device):
    """teapot.obj is walked in memory: 64-byte nodes, 192-byte nodes (bit 15) and two rays per lane (bit 1) are the
    same walk per ray, so a frame's accumulation buffer is the same to the bit."""
    imgs = []
    for mode in (0x01, 0x8001, 0x03, 0x2003):
        s = load_gpu_scene("teapot.obj", pta.BVH)
        assert not s.get_option(pta.OPT_TREE_IN_LDS)
        s.set_option(pta.OPT_TRAVERSAL, mode)
        s.set_resolution(96, 72)
        s.pathtrace(4)
        imgs.append(s.accumulation_buffer())
        s.close()
    for im in imgs[1:]:
        assert np.array_equal(imgs[0], im)


def test_reciprocal_division_is_exact_on_a_slice(gpu_device):
    """The exhaustive run (all 2^46 significand pairs, no mismatch) is recorded in
    profiles/r01_verify_recip_division.log; here 2^37 pairs from both ends and the middle."""
    exe = os.path.join(ROOT, "tools", "verify_recip")
    assert os.path.exists(exe), "build it with __graft_entry__.build()"
    for first in ("0", "511", "1022"):
        out = subprocess.run([exe, first, "2"], capture_output=True, text=True, timeout=120)
        assert out.returncode == 0, out.stdout + out.stderr
        assert re.search(r"TOTAL 137438953472 pairs, 0 mismatches", out.stdout), out.stdout
    bad = subprocess.run([exe, "300", "1", "1"], capture_output=True, text=True, timeout=120)
    assert bad.returncode == 1  # the uncorrected product is caught by the same checker


def test_refined_hardware_reciprocal_is_correctly_rounded(gpu_device):
    """exact_reciprocal (v_rcp_f32 + one Newton step) against the IEEE quotient 1/d for every binary32 d with
    2^-60 <= |d| <= 2^60: the whole sweep takes seconds; the raw v_rcp_f32 must be caught by the same checker."""
    exe = os.path.join(ROOT, "tools", "verify_fast_rcp")
    assert os.path.exists(exe), "build it with __graft_entry__.build()"
    out = subprocess.run([exe, "0"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and re.search(r"2030043136 values checked, 0 mismatches", out.stdout), out.stdout + out.stderr
    raw = subprocess.run([exe, "1"], capture_output=True, text=True, timeout=120)
    assert re.search(r"raw v_rcp_f32: 2030043136 values checked, [1-9]\d* mismatches", raw.stdout), raw.stdout


def test_courtyard_stand_in_scene(gpu_device, tmp_path):
    """Small version of the sponza stand-in: every material kind, texture, probe; loader and hits
    against the oracle, image against the oracle."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from make_sponza import CAMERA, build, write
    from PIL import Image
    obj = str(tmp_path / "court.obj")
    n = write(build(seed=7, detail=0.2), obj)
    assert 5000 < n < 20000
    o = orc.Oracle()
    assert o.load_obj(obj) == n
    for i, m in enumerate(o.materials()):
        if m.textured:
            o.set_texture(i, np.asarray(Image.open(m.tex_file.decode()).convert("RGB"), np.uint8))
    from common import read_hdr_independent
    o.set_env(read_hdr_independent(scene_path("probe_64.hdr")))
    o.set_camera(**CAMERA)
    o.build(orc.BVH)
    s = pta.Scene(gpu_device)
    assert s.load(obj) and s.num_triangles() == n
    types = sorted(m.type for m in s.materials())
    assert types.count(pta.METAL) == 1 and types.count(pta.GLASS) == 1
    s.load_env(scene_path("probe_64.hdr"))
    s.set_camera(**CAMERA)
    s.pre_calc(pta.BVH)
    W, H = 96, 54
    s.set_resolution(W, H)
    rays = s.camera_rays()
    ref, got = o.trace(rays), s.intersect(rays)
    assert np.array_equal(ref["tri"], got["tri"]) and np.array_equal(ref["t"].view(np.uint32), got["t"].view(np.uint32))
    for _ in range(4):
        s.pathtrace(16)
    g = s.frame_buffer()
    o.reset(W, H)
    o.render(spp=16, frames=4, threads=8)
    a = o.image()
    o.reset(W, H)
    o.render(spp=16, frames=4, threads=8, state=0x5DEECE66D | 1)
    b = o.image()
    floor = float(np.sqrt(np.mean((a - b) ** 2)))
    assert float(np.sqrt(np.mean((g - a) ** 2))) <= 1.25 * floor
    assert abs(g.mean() - a.mean()) <= 0.03 * a.mean()


def run_cli(*args):
    exe = os.path.join(ROOT, "path_tracer_amd", "CS488")
    return subprocess.run([exe] + list(args), capture_output=True, text=True, timeout=120)


def test_cli_stdout_protocol(gpu_device, tmp_path):
    obj = scene_path("cornellbox.obj")
    out_png = str(tmp_path / "frame.png")
    r = run_cli(obj, "--frames", "3", "--width", "64", "--height", "48", "--out", out_png)
    assert r.returncode == 0, r.stderr
    lines = r.stdout.splitlines()
    assert lines[0] == 'Loading "%s"...' % obj
    assert lines[1] == 'Loaded "%s" with 36 triangles.' % obj
    assert lines[2] == "Building KdTree..." and lines[3] == "Done."   # kd-tree is the reference's default
    body = lines[4:10]
    assert [body[0], body[2], body[4]] == ["10", "20", "30"]            # total samples, then seconds, per frame
    for sec in (body[1], body[3], body[5]):
        assert float(sec) > 0
    assert lines[10] == 'Saved "%s".' % out_png
    from PIL import Image
    img = np.asarray(Image.open(out_png))
    assert img.shape == (48, 64, 3) and img.max() == 255 and img[0].max() < 5   # light saturates; top border is black
    r = run_cli(obj, scene_path("probe_64.hdr"), "--accel", "bvh", "--frames", "1", "--spp", "2", "--width", "32", "--height", "24",
                "--no-progressive", "--stats")
    assert r.returncode == 0
    assert 'Loaded "%s".' % scene_path("probe_64.hdr") in r.stdout and "Building BVH..." in r.stdout
    assert r.stdout.splitlines()[-2] == "2" and "Mrays/s" in r.stderr and "tests/ray" not in r.stderr
    r = run_cli(obj, "--accel", "bvh", "--frames", "1", "--spp", "2", "--width", "32", "--height", "24", "--count-tests")
    assert r.returncode == 0 and "Mrays/s" in r.stderr and "box tests/ray" in r.stderr  # the counting kernels, on request


def test_cli_frames_overlap_unless_asked_to_drain(gpu_device, tmp_path):
    """Without a reader of the pixels the frame loop only waits for the frame a few before (pt_wait_frames); with
    --drain-frames it waits for every frame as the reference's join does.  Same lines, same image."""
    obj = scene_path("cornellbox.obj")
    outs = []
    for extra in ([], ["--drain-frames"]):
        out = str(tmp_path / ("frame%d.pfm" % len(outs)))
        r = run_cli(obj, "--accel", "bvh", "--frames", "9", "--spp", "3", "--width", "80", "--height", "60", "--out", out, "--trace-frames", *extra)
        assert r.returncode == 0, r.stderr
        body = r.stdout.splitlines()[4:22]
        assert body[0::2] == [str(3 * k) for k in range(1, 10)] and all(float(x) > 0 for x in body[1::2])
        outs.append(open(out, "rb").read())
        # a frame's two lines are printed after the frame has ended, as the reference prints them after its join
        # (src/cs488.h:1893-1896, 2322): the pairs printed never outnumber the frames that are through
        trace = [l.split() for l in r.stderr.splitlines() if l.startswith("frame-lines")]
        assert len(trace) == 9 and [int(t[1]) for t in trace] == list(range(1, 10))
        for t in trace:
            printed, queued, unfinished = int(t[1]), int(t[3]), int(t[5])
            assert printed <= queued - unfinished
        if extra:  # the join: every frame's lines right after the frame
            assert all(int(t[3]) == int(t[1]) and int(t[5]) == 0 for t in trace)
    assert outs[0] == outs[1]


@pytest.mark.gpu
def test_frames_in_flight_counts_down_without_waiting(gpu_device):
    s = load_gpu_scene("cornellbox.obj", pta.BVH)
    s.set_resolution(512, 384)
    assert s.frames_in_flight() == 0
    for _ in range(6):
        s.pathtrace(10)
    seen = [s.frames_in_flight()]
    assert 0 <= seen[0] <= 6
    s.wait_frames(2)
    seen.append(s.frames_in_flight())
    assert seen[1] <= 2
    s.synchronize()
    assert s.frames_in_flight() == 0
    s.close()


@pytest.mark.gpu
def test_wait_frames(gpu_device):
    """pt_wait_frames(keep) returns once at most `keep` queued frames are unfinished; waiting for all of them leaves
    the image a frame-by-frame run gives."""
    def render(pace):
        s = load_gpu_scene("cornellbox.obj", pta.BVH)
        s.set_resolution(96, 72)
        for _ in range(7):
            s.pathtrace(4)
            if pace == "each":
                s.synchronize()
            elif pace == "keep2":
                s.wait_frames(2)
        s.wait_frames(0)
        s.wait_frames(5)   # nothing left to wait for
        img = s.frame_buffer()
        with pytest.raises(Exception):
            s.wait_frames(-1)
        s.close()
        return img
    a, b, c = render("each"), render("keep2"), render("none")
    assert np.array_equal(a, b) and np.array_equal(a, c)


def test_cli_fallbacks(gpu_device, tmp_path):
    r = run_cli("--frames", "1", "--width", "16", "--height", "12")
    assert r.returncode == 0
    assert r.stdout.startswith("Specify .obj file in the command line arguments. Example: CS488.exe cornellbox.obj\n"
                               "Making a single triangle instead.\nBuilding KdTree...\nDone.\n10\n")
    missing = str(tmp_path / "none.obj")
    r = run_cli(missing, "--frames", "1", "--width", "16", "--height", "12")
    assert r.returncode == 0
    assert r.stdout.startswith('Loading "%s"...\nCannot open "%s" for reading\nInvalid .obj file.\nMaking a single triangle instead.\n'
                               % (missing, missing))
    r = run_cli(scene_path("cornellbox.obj"), "--accel", "octree")
    assert r.returncode == 2


@pytest.mark.gpu
def test_overflowing_batches_are_rendered_again(gpu_device):
    """Every surface translucent: a path forks at each hit (src/cs488.h:2036-2045), so a workgroup's
    population outgrows a 2x queue.  Flagged batches add nothing and are rendered again in halves at
    the next synchronisation; the image must equal the one rendered with queues that never overflow
    (up to the order of the float atomics on a sample shared by forked paths), and both must agree
    with the oracle statistically."""
    def render(factor, max_batch):
        s = load_gpu_scene("cornellbox.obj", pta.BVH)
        mats = s.materials()
        for m in mats:
            m.opacity = 0.5
        s.set_materials(mats)
        s.pre_calc(pta.BVH)
        s.set_option(pta.OPT_QUEUE_FACTOR, factor)
        s.set_option(pta.OPT_MAX_BATCH, max_batch)
        s.set_resolution(64, 48)
        for _ in range(2):
            s.pathtrace(8)
        img = s.accumulation_buffer()
        c = s.counters()
        return img, c

    small, c_small = render(2, 64 * 48 * 8)
    big, c_big = render(64, 64 * 48 * 8)
    assert c_big[pta.CNT_REPAIRED] == 0
    assert c_small[pta.CNT_REPAIRED] > 0
    assert c_small[pta.CNT_SAMPLES] == c_big[pta.CNT_SAMPLES] == 64 * 48 * 16
    assert np.allclose(small, big, rtol=2e-4, atol=1e-5)
    # pixel batches (one sample per pixel per batch) take the other way of halving
    pix, c_pix = render(2, 1024)
    assert c_pix[pta.CNT_REPAIRED] > 0
    assert np.allclose(pix, big, rtol=2e-4, atol=1e-5)

    o = load_oracle_scene("cornellbox.obj", orc.BVH)
    for i in range(o.num_materials()):
        o.set_material(i, opacity=0.5)
    o.build(orc.BVH)
    o.reset(64, 48)
    o.render(spp=16, frames=1, threads=8)
    ref = o.image()
    assert abs(big.mean() / 16 - ref.mean()) < 0.03 * ref.mean()


@pytest.mark.gpu
@pytest.mark.parametrize("obj,accel", [("cornellbox.obj", pta.BVH), ("cornellbox-alpha.obj", pta.KD), ("teapot.obj", pta.BVH)])
def test_fused_tail_gives_the_per_level_image(gpu_device, obj, accel):
    """The fused launch of the deep levels runs the same device code on the same queue regions as
    the per-level launches: same ray count, same image (bit for bit where no two paths share a
    sample; forked paths add to one sample with float atomics, whose order is free)."""
    def render(tail):
        s = load_gpu_scene(obj, accel, env="probe_64.hdr" if obj.startswith("teapot") else None)
        s.set_option(pta.OPT_TAIL_LEVEL, tail)
        s.set_resolution(96, 72)
        s.pathtrace(16)
        return s.accumulation_buffer(), s.counters()

    ref, c_ref = render(0)
    assert c_ref[pta.CNT_TAIL_RAYS] == 0
    for tail in (6, 2, 1):
        img, c = render(tail)
        assert c[pta.CNT_RAYS] == c_ref[pta.CNT_RAYS]
        assert c[pta.CNT_HITS] == c_ref[pta.CNT_HITS]
        assert 0 < c[pta.CNT_TAIL_RAYS] < c[pta.CNT_RAYS] or obj.startswith("teapot")
        if "alpha" in obj:
            assert np.allclose(img, ref, rtol=1e-5, atol=1e-6)
        else:
            assert np.array_equal(img, ref)


@pytest.mark.gpu
def test_tail_level_follows_the_measured_rays_per_level(gpu_device):
    """With PT_OPT_TAIL_LEVEL left alone the first batch uses the rule for the scene's kind (level 2 for a
    frame this small); once that batch's rays per level have been read back, a level that holds few rays
    hands over earlier (level 1 here), so later frames trace more of their rays in the fused launch --
    and the image stays the per-level one bit for bit."""
    def render(tail):
        s = load_gpu_scene("cornellbox.obj", pta.BVH)
        if tail is not None:
            s.set_option(pta.OPT_TAIL_LEVEL, tail)
        s.set_resolution(96, 72)
        per_frame = []
        for _ in range(3):
            before = s.counters()[pta.CNT_TAIL_RAYS]
            s.pathtrace(16)
            s.synchronize()
            per_frame.append(s.counters()[pta.CNT_TAIL_RAYS] - before)
        return s.accumulation_buffer(), per_frame, s.counters()[pta.CNT_RAYS]

    ref, none, rays_ref = render(0)
    assert none == [0, 0, 0]
    img, tail_rays, rays = render(None)
    assert rays == rays_ref and np.array_equal(img, ref)
    assert 0 < tail_rays[0] < tail_rays[1] and tail_rays[2] > tail_rays[0]
    _, fixed, _ = render(1)
    assert abs(tail_rays[1] - fixed[1]) < 0.1 * fixed[1]  # level 1: every ray after the camera rays' first bounce


@pytest.mark.gpu
@pytest.mark.parametrize("obj,accel", [("cornellbox.obj", pta.BVH), ("cornellbox.obj", pta.KD), ("teapot.obj", pta.BVH)])
def test_frames_in_flight_do_not_change_the_image(gpu_device, obj, accel):
    """PT_OPT_STREAMS: consecutive batches run on streams of their own, but the kernels that touch
    the image stay on the context's stream in program order, so frames, counters and the image after
    every frame are bit for bit those of one batch at a time -- also with many small batches per
    frame, with a restart in between and with an external accumulation buffer."""
    def render(lanes, max_batch):
        s = load_gpu_scene(obj, accel, env="probe_64.hdr" if obj.startswith("teapot") else None)
        s.set_option(pta.OPT_STREAMS, lanes)
        assert s.get_option(pta.OPT_STREAMS) == lanes
        s.set_option(pta.OPT_MAX_BATCH, max_batch)
        s.set_resolution(96, 72)
        s.pathtrace(3)
        s.reset()                      # a restart with batches still in flight (src/cs488.h:2288-2290)
        frames = []
        for _ in range(5):
            s.pathtrace(4)
            frames.append(s.frame_buffer())
        return frames, s.accumulation_buffer(), s.counters()

    for max_batch in (1 << 24, 96 * 72, 1000):
        ref_frames, ref_acc, ref_c = render(1, max_batch)
        for lanes in (2, 3, 4):
            frames, acc, c = render(lanes, max_batch)
            assert np.array_equal(acc, ref_acc)
            for a, b in zip(frames, ref_frames):
                assert np.array_equal(a, b)
            assert list(c[:8]) == list(ref_c[:8])   # rays, tests, samples, hits ...; where the fused launch takes over may differ
    s = load_gpu_scene(obj, accel)
    with pytest.raises(Exception):
        s.set_option(pta.OPT_STREAMS, 5)
    with pytest.raises(Exception):
        s.set_option(pta.OPT_STREAMS, 0)


@pytest.mark.gpu
def test_batches_in_flight_are_chosen_per_scene(gpu_device):
    """Unless the caller sets PT_OPT_STREAMS it is chosen when a frame is queued: two batches in flight for large
    frames of a scene walked in LDS whose surfaces are all diffuse, three otherwise; a context whose frames go into a
    collective (an external per-frame buffer) keeps two; a value the caller set is left alone."""
    for obj, w, h, want in (("cornellbox.obj", 1024, 1024, 2), ("cornellbox.obj", 512, 384, 3), ("cornellbox-alpha.obj", 1024, 1024, 3),
                            ("cornellbox-glass.obj", 1024, 1024, 3), ("teapot.obj", 1024, 1024, 3)):
        s = load_gpu_scene(obj, pta.BVH)
        s.set_resolution(w, h)
        s.pathtrace(10)
        assert s.get_option(pta.OPT_STREAMS) == want, (obj, w, h)
        s.close()
    s = pta.Scene(gpu_device)
    s.set_option(pta.OPT_STREAMS, 4)
    assert s.load(scene_path("cornellbox.obj"))
    s.pre_calc(pta.BVH)
    assert s.get_option(pta.OPT_STREAMS) == 4
    s.close()
    from common import DeviceArray
    s = load_gpu_scene("teapot.obj", pta.BVH, env="probe_64.hdr")
    s.set_resolution(64, 48)
    s.pathtrace(2)
    own = s.accumulation_buffer()
    buf = DeviceArray(np.zeros(64 * 48 * 3, np.float32))
    s.use_external_accum(buf.ptr.value)
    assert s.get_option(pta.OPT_STREAMS) == 2
    s.reset()
    s.pathtrace(2)
    s.synchronize()
    assert np.array_equal(buf.download().reshape(own.shape), own)
    s.use_external_accum(0)
    s.close()
    buf.free()


@pytest.mark.gpu
def test_cli_scripted_input(gpu_device, tmp_path):
    """Scripted events stand in for the window's callbacks (src/cs488.h:240-349): a translation keeps
    accumulating (the reference restarts only when view direction or right vector change, :2266-2293),
    a drag restarts, I / F write files named with the reference's generator, ESCAPE ends the loop."""
    script = tmp_path / "events.txt"
    script.write_text("# frame event\n1 key W\n2 key I\n3 drag 40 -10\n4 key F\n6 key F\n6 save %s\n8 key ESCAPE\n" % (tmp_path / "shot.pfm"))
    r = subprocess.run([os.path.join(ROOT, "path_tracer_amd", "CS488"), scene_path("cornellbox.obj"), "--accel", "bvh", "--frames", "12",
                        "--spp", "2", "--width", "48", "--height", "36", "--script", str(script), "--sequence", str(tmp_path / "seq_")],
                       capture_output=True, text=True, timeout=120, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr
    lines = r.stdout.splitlines()
    i0 = lines.index("Done.") + 1
    body = [ln for ln in lines[i0:] if not ln.startswith(("Saved", "Saving", "(Recording"))]
    totals = body[0::2]
    # frames 0..7 render (ESCAPE at 8); W at frame 1 keeps the count running, the drag at frame 3 restarts it
    assert totals == ["2", "4", "6", "2", "4", "6", "8", "10"]
    assert "(Recording started)" in r.stdout and "(Recording done)" in r.stdout
    names = sorted(os.listdir(tmp_path))
    stills = [n for n in names if re.fullmatch(r"output\d+\.png", n)]
    recorded = [n for n in names if re.fullmatch(r"output\d+\.gif", n)]
    assert len(stills) == 1 and len(recorded) == 1
    from PIL import Image
    gif = Image.open(tmp_path / recorded[0])
    assert gif.size == (48, 36) and gif.n_frames == 2      # frames 4 and 5 were recorded
    assert gif.info["duration"] == 10                       # globalGIFdelay = 1 hundredth (src/cs488.h:107)
    gif.seek(1)                                             # the frame rendered as seq_0005 (same clamp and truncation)
    want = np.asarray(Image.open(tmp_path / "seq_0005.png").convert("RGB")).astype(int)
    got = np.asarray(gif.convert("RGB")).astype(int)
    assert np.abs(got - want).max() <= 51                   # one step of the 6 x 7 x 6 palette
    assert abs(got.mean() - want.mean()) < 2.0              # the ordered dither keeps the mean
    assert len([n for n in names if n.startswith("seq_")]) == 8 and "shot.pfm" in names
    # the generator is the reference's PCG32 (src/cs488.h:130-143): the I key at frame 2 comes after two per-frame draws
    _, vals, _ = orc.pcg32(3)
    assert stills[0] == "output%d.png" % int(1000.0 * float(vals[2]))


def test_an_lds_stage_that_cannot_launch_is_refused(gpu_device):
    """PT_OPT_LDS_NODES beyond what a workgroup's LDS holds beside its traversal stacks is refused when it is set (every
    trace launch would fail otherwise), and the setting in force stays."""
    o = load_oracle_scene("teapot.obj", pta.BVH, textures=False)
    rays = ray_batch(o.get_tris()[0], n_random=3000, n_vertex=100, seed=5)
    ref = o.trace(rays)
    s = load_gpu_scene("teapot.obj", pta.BVH)
    s.set_option(pta.OPT_LDS_NODES, 256)
    assert s.get_option(pta.OPT_LDS_NODES) == 256
    with pytest.raises(pta.PtError) as e:
        s.set_option(pta.OPT_LDS_NODES, 2048)
    assert e.value.code == pta.ERR_INVALID_ARG
    assert s.get_option(pta.OPT_LDS_NODES) == 256
    got = s.intersect(rays)
    assert np.array_equal(got["tri"], ref["tri"]) and np.array_equal(got["t"].view(np.uint32), ref["t"].view(np.uint32))
    s.close()
