#!/usr/bin/env python3
"""The multi-GPU frame loop on one GPU: two contexts play ranks 0 and 1 of a sample split, each renders
its share of every frame into one of two caller-owned device buffers in turn (torch tensors, the context's
stream is torch's), a FrameReducer folds each buffer into that rank's total when its turn comes again.
With two frames in flight per context the accumulate of frame k must land in the buffer that was current
when frame k was issued, after its clear and before it is added to the total: the sum of the two ranks'
totals must equal one context rendering all samples.  Run by test_gpu_image_parity.py as a child process
(PyTorch opens the GPU first here)."""
import os
import sys

import numpy as np
import torch

torch.cuda.init()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch.distributed as dist  # noqa: E402
import orc  # noqa: E402
import path_tracer_amd as pta  # noqa: E402
from common import load_gpu_scene  # noqa: E402
from path_tracer_amd.distributed import FrameReducer, sample_share  # noqa: E402

W, H, spp, frames = 160, 120, 4, 7
ref = load_gpu_scene("cornellbox.obj", orc.BVH)
ref.set_resolution(W, H)
for _ in range(frames):
    ref.pathtrace(2 * spp)
want = ref.accumulation_buffer().reshape(-1)

stream = torch.cuda.current_stream()
ranks = []
for r in range(2):
    s = load_gpu_scene("cornellbox.obj", orc.BVH)
    s.set_stream(stream.cuda_stream)
    s.set_resolution(W, H)
    deltas = [torch.full((W * H * 3,), 7.0, dtype=torch.float32, device="cuda") for _ in range(2)]
    master = torch.zeros(W * H * 3, dtype=torch.float32, device="cuda")
    ranks.append((s, FrameReducer(dist, deltas, master), sample_share(2 * spp, 2, r)))
for _ in range(frames):
    for s, red, (first, count) in ranks:
        s.use_external_accum(red.begin_frame().data_ptr())
        s.pathtrace_split(2 * spp, first, count)
        red.reduce(2 * spp)
total = None
for s, red, _ in ranks:
    red.finish()
    s.synchronize()
    total = red.master.clone() if total is None else total + red.master
    assert red.total_samples == frames * 2 * spp and s.total_samples() == frames * 2 * spp
torch.cuda.synchronize()
got = total.cpu().numpy()
# every frame's halves are summed from zero and added, so only the rounding of those additions differs
assert np.allclose(got, want, rtol=2e-6, atol=1e-6), float(np.abs(got - want).max())
for s, _, _ in ranks:
    s.use_external_accum(0)
    s.close()
print("external accumulation ok")
