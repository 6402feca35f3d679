"""The multi-GPU path of the C++ host (pt_group_*: sample split over the GPUs of one process, one RCCL ncclReduce per
frame onto the first member) on the one GPU a test box has: a group of ONE member goes through the same code --
per-frame buffer, ncclCommInitAll, ncclReduce on the member's stream, fold into the accumulation, resolve -- and must
give the single-context image.  (Scaling over several GPUs cannot be measured here; the split itself is covered by the
sample-split tests of test_gpu_image_parity.py and, over processes, by test_distributed_gloo.py.)"""
import os
import subprocess

import numpy as np
import pytest

import path_tracer_amd as pta
from common import load_gpu_scene, scene_path

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_group_rejects_what_cannot_render():
    import ctypes
    L = pta.lib()
    g = ctypes.c_void_p()
    assert L.pt_group_create(0, None, ctypes.byref(g)) == pta.ERR_INVALID_ARG
    h = ctypes.c_void_p()
    assert L.pt_create(pta.DEVICE_NONE, ctypes.byref(h)) == pta.PT_OK
    arr = (ctypes.c_void_p * 1)(h)
    assert L.pt_group_create(1, arr, ctypes.byref(g)) == pta.ERR_NOT_READY and not g.value
    assert L.pt_group_render(None, 1) == pta.ERR_INVALID_ARG
    L.pt_destroy(h)


@pytest.mark.gpu
@pytest.mark.parametrize("obj,accel,env", [("cornellbox.obj", pta.BVH, None), ("cornellbox-glass.obj", pta.KD, "probe_64.hdr")])
def test_group_of_one_is_the_single_context_image(gpu_device, obj, accel, env):
    W, H, SPP = 160, 120, 6
    plain = load_gpu_scene(obj, accel, env=env)
    plain.set_resolution(W, H)
    member = load_gpu_scene(obj, accel, env=env)
    member.set_resolution(W, H)
    group = pta.Group([member])
    plain.pathtrace(SPP)
    group.pathtrace(SPP)
    group.synchronize()
    # first frame: the per-frame buffer is summed from zero exactly like the accumulation buffer of the plain context
    assert np.array_equal(member.frame_buffer(), plain.frame_buffer())
    assert np.array_equal(member.accumulation_buffer(), plain.accumulation_buffer())
    for _ in range(3):
        plain.pathtrace(SPP)
        group.pathtrace(SPP)
    # later frames add a frame's sum to the accumulation in one addition instead of sample by sample
    assert member.total_samples() == plain.total_samples() == 4 * SPP
    assert np.allclose(member.frame_buffer(), plain.frame_buffer(), rtol=2e-6, atol=1e-7)
    assert member.counters()[pta.CNT_RAYS] == plain.counters()[pta.CNT_RAYS]
    with pytest.raises(pta.PtError):  # a member is driven through the group only
        member.pathtrace(SPP)
    group.reset()
    assert member.total_samples() == 0 and not member.frame_buffer().any() and not member.accumulation_buffer().any()
    group.pathtrace(SPP)
    plain.reset()
    plain.pathtrace(SPP)
    assert np.array_equal(member.frame_buffer(), plain.frame_buffer())
    group.close()
    # out of the group the context renders by itself again
    member.reset()
    member.pathtrace(SPP)
    assert np.array_equal(member.frame_buffer(), plain.frame_buffer())
    member.close()
    plain.close()


@pytest.mark.gpu
def test_frame_reducer_settles_overflowed_batches_into_a_buffer_of_their_own(gpu_device):
    """The one-process-per-GPU form of the same hazard: frames go into the exchange without a wait, so a batch that
    overflowed is rendered again only at FrameReducer.settle -- into a cleared buffer that is reduced like a frame.
    The accumulation is short of those batches before, complete after."""
    import torch
    import torch.distributed as dist
    from path_tracer_amd.distributed import FrameReducer

    def scene(factor):
        s = load_gpu_scene("cornellbox.obj", pta.BVH)
        mats = s.materials()
        for m in mats:
            m.opacity = 0.5
        s.set_materials(mats)
        s.pre_calc(pta.BVH)
        s.set_option(pta.OPT_QUEUE_FACTOR, factor)
        s.set_option(pta.OPT_MAX_BATCH, 64 * 48 * 8)
        s.set_resolution(64, 48)
        return s

    ref = scene(64)
    for _ in range(2):
        ref.pathtrace(8)
    want = ref.accumulation_buffer().reshape(-1)
    s = scene(2)
    s.set_stream(torch.cuda.current_stream().cuda_stream)
    n = 64 * 48 * 3
    deltas = [torch.zeros(n, dtype=torch.float32, device="cuda") for _ in range(2)]
    master = torch.zeros(n, dtype=torch.float32, device="cuda")
    red = FrameReducer(dist, deltas, master)  # no process group: one rank, the collective call is skipped
    for _ in range(2):
        s.use_external_accum(red.begin_frame().data_ptr())
        s.pathtrace_split(8, 0, 8)
        red.reduce(8)
    red.finish()
    torch.cuda.synchronize()
    short = master.cpu().numpy().copy()
    red.settle(s)
    red.finish()
    torch.cuda.synchronize()
    c = s.counters()
    assert c[pta.CNT_REPAIRED] > 0 and c[pta.CNT_SAMPLES] == 64 * 48 * 16
    got = master.cpu().numpy()
    assert short.sum() < 0.9 * want.sum()  # whole batches were missing
    assert np.allclose(got, want, rtol=2e-4, atol=1e-5)
    assert red.total_samples == 16
    s.use_external_accum(0)
    s.close()
    ref.close()


@pytest.mark.gpu
@pytest.mark.parametrize("wait", ["group", "download"])
def test_group_completes_overflowed_batches_at_the_next_wait(gpu_device, wait):
    """Translucent surfaces everywhere and 2x queues: batches overflow, add nothing at first and are rendered again
    at a wait -- by then their frame's buffer has been through the reduce, so the wait (pt_group_synchronize, or the
    root's own pt_download) clears the per-frame buffers, lets the repairs land there and reduces once more."""
    def scene(factor):
        s = load_gpu_scene("cornellbox.obj", pta.BVH)
        mats = s.materials()
        for m in mats:
            m.opacity = 0.5
        s.set_materials(mats)
        s.pre_calc(pta.BVH)
        s.set_option(pta.OPT_QUEUE_FACTOR, factor)
        s.set_option(pta.OPT_MAX_BATCH, 64 * 48 * 8)
        s.set_resolution(64, 48)
        return s

    ref = scene(64)
    for _ in range(2):
        ref.pathtrace(8)
    member = scene(2)
    group = pta.Group([member])
    for _ in range(2):
        group.pathtrace(8)
    if wait == "group":
        group.synchronize()
    got = member.accumulation_buffer()
    c = member.counters()
    assert c[pta.CNT_REPAIRED] > 0 and c[pta.CNT_SAMPLES] == 64 * 48 * 16
    assert np.allclose(got, ref.accumulation_buffer(), rtol=2e-4, atol=1e-5)
    assert member.total_samples() == 16
    group.close()
    member.close()
    ref.close()


@pytest.mark.gpu
def test_cli_gpus_option_keeps_the_protocol_and_the_image(gpu_device, tmp_path):
    exe = os.path.join(ROOT, "path_tracer_amd", "CS488")
    outs = []
    for extra in ([], ["--gpus", "1"], ["--gpus", "1", "--reduce-every", "2"]):  # three frames at two per reduce: the last by the final wait
        pfm = str(tmp_path / ("f%d.pfm" % len(outs)))
        r = subprocess.run([exe, scene_path("cornellbox.obj"), "--accel", "bvh", "--frames", "3", "--spp", "4", "--width", "96", "--height", "72",
                            "--out", pfm, "--stats"] + extra, capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr
        lines = r.stdout.splitlines()
        assert lines[0].startswith("Loading") and "Building BVH..." in lines and "Done." in lines
        k = lines.index("Done.")
        assert [lines[k + 1 + 2 * i] for i in range(3)] == ["4", "8", "12"]   # totalSamples, then seconds, per frame
        assert all(float(lines[k + 2 + 2 * i]) > 0 for i in range(3))
        assert "Mrays/s" in r.stderr
        raw = open(pfm, "rb").read()
        outs.append(np.frombuffer(raw[-96 * 72 * 12:], np.float32))
    assert np.allclose(outs[0], outs[1], rtol=2e-6, atol=1e-7)
    assert np.allclose(outs[0], outs[2], rtol=2e-6, atol=1e-7)


@pytest.mark.gpu
def test_group_wait_frames(gpu_device):
    """pt_group_wait_frames(keep) paces a frame loop that reads no pixels; the image is the one a loop that waits
    for every frame gives."""
    def render(pace):
        s = load_gpu_scene("cornellbox.obj", pta.BVH)
        s.set_resolution(96, 72)
        g = pta.Group([s])
        for _ in range(7):
            g.pathtrace(4)
            if pace == "each":
                g.synchronize()
            elif pace == "keep2":
                g.wait_frames(2)
        g.wait_frames(0)
        g.wait_frames(3)  # nothing left to wait for
        with pytest.raises(Exception):
            g.wait_frames(-1)
        img = s.frame_buffer()
        g.close()
        s.close()
        return img
    a, b, c = render("each"), render("keep2"), render("none")
    assert np.array_equal(a, b) and np.array_equal(a, c)


@pytest.mark.gpu
def test_group_cadence_reduces_every_kth_frame_and_gives_the_same_image(gpu_device):
    """pt_group_set_cadence(k): k frames are added into one per-frame buffer before it is reduced; the wait reduces the
    frames since the last one.  Same samples, same sum up to the order of the float additions."""
    def render(cadence, frames):
        s = load_gpu_scene("cornellbox.obj", pta.BVH)
        s.set_resolution(96, 72)
        g = pta.Group([s])
        if cadence > 1:
            g.set_cadence(cadence)
        for _ in range(frames):
            g.pathtrace(4)
        g.synchronize()
        img, acc, total = s.frame_buffer(), s.accumulation_buffer(), s.total_samples()
        stats = g.stats()
        g.close()
        s.close()
        return img, acc, total, stats
    a, acc_a, total_a, st_a = render(1, 7)
    b, acc_b, total_b, st_b = render(3, 7)
    assert total_a == total_b == 28
    assert st_a[:2] == (7, 7) and st_b[:2] == (7, 3)  # 3 + 3 frames by the cadence, the seventh by the wait
    assert st_a[2] == 0                                # a group of one queues its launches on the caller's thread
    assert np.allclose(a, b, rtol=2e-6, atol=1e-7) and np.allclose(acc_a, acc_b, rtol=2e-6, atol=1e-6)
    assert a.mean() > 0.01
    with pytest.raises(pta.PtError):
        s = load_gpu_scene("cornellbox.obj", pta.BVH)
        s.set_resolution(32, 24)
        g = pta.Group([s])
        try:
            g.set_cadence(0)
        finally:
            g.close()
            s.close()


def _gpus():
    import torch
    return torch.cuda.device_count()


@pytest.mark.gpu
@pytest.mark.parametrize("obj,env,cadence", [("cornellbox.obj", None, 1), ("cornellbox-alpha.obj", "probe_64.hdr", 1), ("cornellbox.obj", None, 2)])
def test_group_of_two_gpus_gives_the_single_gpu_image(gpu_device, obj, env, cadence):
    """Two members on two GPUs (runs only where the node shows at least two): every member's launches queued by its own
    host thread, ncclCommInitAll over both devices, one grouped ncclReduce per frame on the members' exchange streams.
    The image on members[0] is the single-GPU image up to the order of the float additions, for a scene whose paths fork
    (exact fixed-point radiance per sample, so the split changes only the order of the per-sample adds) and one whose
    paths do not."""
    if _gpus() < 2:
        pytest.skip("needs two GPUs in one node")
    W, H, SPP = 160, 120, 6
    plain = load_gpu_scene(obj, pta.BVH, env=env)
    plain.set_resolution(W, H)
    members = [load_gpu_scene(obj, pta.BVH, env=env, device=d) for d in (0, 1)]
    for m in members:
        m.set_resolution(W, H)
    group = pta.Group(members)
    if cadence > 1:
        group.set_cadence(cadence)
    for _ in range(5):
        plain.pathtrace(SPP)
        group.pathtrace(SPP)
    group.synchronize()
    assert group.stats()[2] == 2
    assert members[0].total_samples() == plain.total_samples() == 5 * SPP
    assert np.allclose(members[0].frame_buffer(), plain.frame_buffer(), rtol=1e-5, atol=1e-6)
    rays = sum(m.counters()[pta.CNT_RAYS] for m in members)
    assert rays == plain.counters()[pta.CNT_RAYS]          # the same paths, split by sample index
    assert min(m.counters()[pta.CNT_RAYS] for m in members) > 0.3 * rays
    group.close()
    for m in members:
        m.close()
    plain.close()
