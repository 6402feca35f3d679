"""The oracle's outputs against the committed record of them (tests/golden/oracle/fixtures.json, written by
tests/golden/make_oracle_fixtures.py): closest hits on the suite's ray batches and on the 512 x 384 pixel-centre camera
rays, tree structure, PCG32 stream and a single-thread frame per scene.  An edit to oracle/pt_oracle.cpp that changes
any result fails here, so the bar the GPU path is held to cannot move unnoticed."""
import json
import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_oracle_fixtures as mk  # noqa: E402

RECORD = json.load(open(mk.OUT))


def pcg32_fast_python(n, state=0xcafef00dd15ea5e5):
    """PCG32::pcg32_fast (src/cs488.h:133-139) in Python integers: an independent statement of the same generator."""
    out = []
    for _ in range(n):
        x = state
        count = x >> 61
        state = (x * 6364136223846793005) & 0xFFFFFFFFFFFFFFFF
        x ^= x >> 22
        out.append((x >> (22 + count)) & 0xFFFFFFFF)
    return out


def test_pcg32_stream_is_the_recorded_one():
    u32, f32, _ = mk.orc.pcg32(1024)
    assert [int(x) for x in u32] == pcg32_fast_python(1024)
    assert [float(x) for x in f32[:64]] == [float(mk.np.float32(v / 4294967296.0)) for v in pcg32_fast_python(64)]  # :140-142
    assert [int(x) for x in u32[:8]] == RECORD["pcg32_1024"]["first_u32"]
    assert mk.sha(u32) == RECORD["pcg32_1024"]["u32_sha256"]
    assert mk.sha(f32.view("uint32")) == RECORD["pcg32_1024"]["f32_sha256"]


@pytest.mark.parametrize("obj,accels,env,frame", mk.SCENES, ids=[s[0] for s in mk.SCENES])
def test_oracle_outputs_are_the_recorded_ones(obj, accels, env, frame):
    for accel in accels:
        key = "%s/%s" % (obj, "kd" if accel == mk.orc.KD else "bvh")
        assert mk.scene_record(obj, accel, env, frame) == RECORD["scenes"][key], key


def test_every_recorded_scene_is_checked():
    keys = {"%s/%s" % (obj, "kd" if a == mk.orc.KD else "bvh") for obj, accels, _, _ in mk.SCENES for a in accels}
    assert keys == set(RECORD["scenes"])


def test_converged_renders_cover_every_benchmark_scene():
    """tests/golden/oracle/converged.npz (two 4096-sample oracle renders per scene, the references of
    tests/test_gpu_converged.py) holds exactly the scenes its generator lists, as finite 48 x 64 images; one of them is
    recomputed at a sixteenth of the samples and must agree with the stored pair within the noise of that."""
    import make_converged_images as cv
    ref = mk.np.load(cv.OUT)
    assert set(ref.files) == {"%s/%s" % (key, tag) for key, _, _, _ in cv.SCENES for tag in "ab"}
    for name in ref.files:
        assert ref[name].shape == (cv.H, cv.W, 3) and mk.np.isfinite(ref[name]).all() and ref[name].mean() > 0.05, name
    o = cv.load_oracle_scene("cornellbox.obj", mk.orc.BVH)
    o.reset(cv.W, cv.H)
    o.render(spp=64, frames=4, threads=cv.THREADS, state=cv.SEEDS[0])
    quick = o.image()
    a, b = ref["cornellbox/a"], ref["cornellbox/b"]
    assert abs(quick.mean() - a.mean()) < 0.01 * a.mean()
    floor = float(mk.np.sqrt(mk.np.mean((a - b) ** 2)))
    assert float(mk.np.sqrt(mk.np.mean((quick - a) ** 2))) < 4.0 * floor  # 1/16 of the samples: about 2.9 x the pair's RMSE
