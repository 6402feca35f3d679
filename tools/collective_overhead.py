#!/usr/bin/env python3
"""GPU experiment: what does the per-frame exchange step of the one-process-per-GPU path cost on ONE rank, where the
collective has nothing to move?  Times the cornellbox frame (1024 x 1024, 10 spp, BVH) with the pieces of
bench.py's collective step added one at a time.  usage: collective_overhead.py [frames]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import path_tracer_amd as pta  # noqa: E402
from common import load_gpu_scene  # noqa: E402
from path_tracer_amd.distributed import FrameReducer  # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 200
torch.cuda.set_device(0)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
n = 1024 * 1024 * 3
deltas = [torch.zeros(n, dtype=torch.float32, device="cuda") for _ in range(2)]
master = torch.zeros(n, dtype=torch.float32, device="cuda")


def scene(lanes=None):
    s = load_gpu_scene("cornellbox.obj", pta.BVH)
    s.set_resolution(1024, 1024)
    s.set_stream(torch.cuda.current_stream().cuda_stream)
    if lanes:
        s.set_option(pta.OPT_STREAMS, lanes)
    return s


def timed(name, s, step):
    for _ in range(10):
        step(s)
    s.synchronize(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(frames):
        step(s)
    s.synchronize(); torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / frames * 1e3
    print("%-58s %.3f ms/frame  (lanes %d)" % (name, ms, s.get_option(pta.OPT_STREAMS)), flush=True)
    s.close()


k = [0]


def plain(s):
    s.pathtrace(10)


def split_only(s):
    s.pathtrace_split(10, 0, 10)


def ext_one_buffer(s):
    s.use_external_accum(deltas[0].data_ptr())
    s.pathtrace_split(10, 0, 10)


def ext_two_buffers(s):
    s.use_external_accum(deltas[k[0] & 1].data_ptr()); k[0] += 1
    s.pathtrace_split(10, 0, 10)


def ext_zero(s):
    d = deltas[k[0] & 1]; k[0] += 1
    d.zero_()
    s.use_external_accum(d.data_ptr())
    s.pathtrace_split(10, 0, 10)


def ext_zero_add(s):
    d = deltas[k[0] & 1]; k[0] += 1
    master.add_(d)
    d.zero_()
    s.use_external_accum(d.data_ptr())
    s.pathtrace_split(10, 0, 10)


def ext_reduce_sync(s):
    d = deltas[k[0] & 1]; k[0] += 1
    d.zero_()
    s.use_external_accum(d.data_ptr())
    s.pathtrace_split(10, 0, 10)
    dist.reduce(d, dst=0)


pending = [None, None]


def ext_reduce_async(s):
    i = k[0] & 1; k[0] += 1
    d = deltas[i]
    if pending[i] is not None:
        pending[i].wait()
    d.zero_()
    s.use_external_accum(d.data_ptr())
    s.pathtrace_split(10, 0, 10)
    pending[i] = dist.reduce(d, dst=0, async_op=True)


timed("pathtrace", scene(), plain)
timed("pathtrace, lanes 2 set by hand", scene(2), plain)
timed("pathtrace_split", scene(), split_only)
timed("external buffer (one)", scene(), ext_one_buffer)
timed("external buffers (two, alternating)", scene(), ext_two_buffers)
timed("... + zero_ per frame", scene(), ext_zero)
timed("... + master.add_ per frame", scene(), ext_zero_add)
timed("... zero_ + dist.reduce (blocking call form)", scene(), ext_reduce_sync)
timed("... zero_ + dist.reduce(async_op) waited two frames later", scene(), ext_reduce_async)
s = scene()
red = FrameReducer(dist, deltas, master)


def full(s_):
    s_.use_external_accum(red.begin_frame().data_ptr())
    s_.pathtrace_split(10, 0, 10)
    red.reduce(10)


timed("FrameReducer step of bench.py", s, full)
red.finish()
dist.destroy_process_group()
