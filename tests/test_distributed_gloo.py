"""world_size-2 test of the sample-split exchange step on CPU tensors (gloo): two ranks render
disjoint sample ranges of the same frames with the CPU oracle standing in for the GPU renderer,
the per-frame accumulation buffers are reduced onto rank 0, and the result must equal a
single-process accumulation of the same per-rank images."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _render_share(rank, frame, first, count, w, h):
    """Accumulation of samples [first, first+count) of one frame; seeded by (frame, sample)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orc
    from common import load_oracle_scene
    o = load_oracle_scene("cornellbox.obj", orc.BVH)
    acc = np.zeros((h, w, 3), np.float32)
    for s in range(first, first + count):
        o.reset(w, h)
        o.render(spp=1, frames=1, threads=1, state=(0x9E3779B97F4A7C15 * (1 + frame * 64 + s)) % (1 << 64) | 1)
        acc += o.image("accum")
    return acc


def _worker(rank, world, port, spp_frame, frames, w, h, out_path, n_buffers, every=1):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    from path_tracer_amd.distributed import FrameReducer, sample_share
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    deltas = [torch.zeros(h * w * 3, dtype=torch.float32) for _ in range(n_buffers)]
    master = torch.zeros(h * w * 3, dtype=torch.float32) if rank == 0 else None
    red = FrameReducer(dist, deltas, master, every=every)
    first, count = sample_share(spp_frame, world, rank)
    for f in range(frames):
        buf = red.begin_frame()
        buf += torch.from_numpy(_render_share(rank, f, first, count, w, h).reshape(-1))
        red.reduce(spp_frame)
    red.finish()
    assert red.reduces == (frames + every - 1) // every and red.total_samples == spp_frame * frames
    if rank == 0:
        np.save(out_path, red.frame().numpy().reshape(h, w, 3))
    dist.barrier()
    dist.destroy_process_group()


def test_sample_share_partitions():
    sys.path.insert(0, ROOT)
    from path_tracer_amd.distributed import sample_share
    for spp in (0, 1, 7, 10, 4096):
        for world in (1, 2, 3, 8):
            got = [sample_share(spp, world, r) for r in range(world)]
            assert got[0][0] == 0 and sum(c for _, c in got) == spp
            for (f0, c0), (f1, _) in zip(got, got[1:]):
                assert f1 == f0 + c0
            assert max(c for _, c in got) - min(c for _, c in got) <= 1
    assert sample_share(4096, 8, 3) == (1536, 512)
    with pytest.raises(ValueError):
        sample_share(8, 2, 2)


@pytest.mark.parametrize("n_buffers,every", [(1, 1), (2, 1), (2, 2)])
def test_two_rank_reduce_matches_single_process(tmp_path, n_buffers, every):
    """n_buffers = 2 is the overlapped form bench.py uses (reduce of frame k in flight while frame
    k+1 is produced); 3 frames so that a buffer is reused while its reduce is pending.  every = 2 is the
    cadence for short frames: two frames go into one buffer before it is reduced, and the odd third
    frame is reduced by finish()."""
    import torch.multiprocessing as mp
    world, spp_frame, frames, w, h = 2, 3, 3, 32, 24
    out = str(tmp_path / "frame.npy")
    mp.spawn(_worker, args=(world, _free_port(), spp_frame, frames, w, h, out, n_buffers, every), nprocs=world, join=True)
    got = np.load(out)
    sys.path.insert(0, ROOT)
    from path_tracer_amd.distributed import sample_share
    want = np.zeros((h, w, 3), np.float32)
    for f in range(frames):
        parts = [_render_share(r, f, *sample_share(spp_frame, world, r), w, h) for r in range(world)]
        want += parts[0] + parts[1]
    want /= float(spp_frame * frames)
    assert np.allclose(got, want, rtol=1e-6, atol=1e-7)
    assert got.mean() > 0.01


def test_settle_reduces_what_a_wait_renders_again():
    """FrameReducer.settle hands the renderer's wait a cleared buffer and sends it through the exchange: what the wait
    adds (batches rendered again after a queue overflow) reaches the accumulation, the frames already reduced are
    not touched, and the sample count does not move.  A stand-in renderer, one rank, CPU tensors."""
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    from path_tracer_amd.distributed import FrameReducer

    deltas = [torch.zeros(12) for _ in range(2)]
    master = torch.zeros(12)
    red = FrameReducer(dist, deltas, master)

    class Renderer:
        target = None
        late = 0.0  # what an overflowed batch still owes

        def use_external_accum(self, ptr):
            self.target = next(d for d in deltas if d.data_ptr() == ptr)

        def render(self, value, overflowed):
            if overflowed:
                self.late += value
            else:
                self.target += value

        def synchronize(self):
            self.target += self.late
            self.late = 0.0

    r = Renderer()
    for frame, overflowed in enumerate((False, True, False)):
        r.use_external_accum(red.begin_frame().data_ptr())
        r.render(float(frame + 1), overflowed)
        red.reduce(4)
    red.finish()
    assert torch.all(master == 4.0) and red.total_samples == 12  # frame 2's batch is still owed
    red.settle(r)
    assert torch.all(red.frame() == 6.0 / 12) and red.total_samples == 12
    red.settle(r)  # nothing owed: a cleared buffer goes through, nothing changes
    assert torch.all(red.frame() == 6.0 / 12)
