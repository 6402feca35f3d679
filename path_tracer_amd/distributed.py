"""Sample-split rendering over several GPUs: one process per GPU (torch.distributed, backend
"nccl" = RCCL on ROCm; "gloo" in the CPU tests), every rank renders its own share of a frame's
samples for ALL pixels and the per-frame accumulation buffers are summed onto rank 0 with one
reduce.  The reference has no counterpart: its only parallelism is the column-strip thread pool
of Scene::Pathtrace (src/cs488.h:1885-1895).
"""


def sample_share(spp_frame, world, rank):
    """(first, count) of the sample indices rank `rank` renders out of a frame's spp_frame
    samples: contiguous ranges, the first spp_frame % world ranks take one extra sample."""
    if spp_frame < 0 or world <= 0 or not (0 <= rank < world):
        raise ValueError("bad sample split: spp_frame=%r world=%r rank=%r" % (spp_frame, world, rank))
    base, extra = divmod(spp_frame, world)
    first = rank * base + min(rank, extra)
    return first, base + (1 if rank < extra else 0)


class FrameReducer:
    """Per-frame exchange step with the reduce overlapped with the next frame's rendering.

    `deltas` are two (or more) per-frame accumulation buffers of this rank; frame k is rendered
    into deltas[k % len(deltas)] (begin_frame() clears it and returns it, the caller points the
    renderer at it), reduce() starts the sum of that buffer over all ranks onto the root without
    waiting for it, and the buffer is only waited for when its turn comes again (or in finish()).
    On the root every completed sum is added to `master`, the progressive accumulation.  With one
    buffer the exchange is synchronous.

    `every` > 1 is the cadence for short frames: that many consecutive frames are rendered into
    one buffer (begin_frame() hands out the same, uncleared buffer; reduce() only counts) and the
    buffer is reduced after the last of them, so a 12.6 MB reduce is paid once per `every` 2-ms
    frames.  The sum and the sample count are the same; the root's image is complete after
    finish() (or frame()), and between reduces it lags by fewer than `every` frames.
    """

    def __init__(self, dist, deltas, master=None, root=0, group=None, every=1):
        self.dist, self.master, self.root, self.group = dist, master, root, group
        self.every = max(1, int(every))
        self.in_buffer = 0   # frames rendered into the current buffer and not reduced yet
        self.reduces = 0     # collectives issued
        self.deltas = list(deltas) if isinstance(deltas, (list, tuple)) else [deltas]
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.total_samples = 0
        self.frame_index = 0
        self.pending = [None] * len(self.deltas)  # per buffer: (work handle or None, needs add)
        if self.rank == root and master is None:
            raise ValueError("the root rank needs a master accumulation buffer")

    def _retire(self, slot):
        """Waits for the reduce issued on buffer `slot` (if any) and folds it into master."""
        p = self.pending[slot]
        if p is None:
            return
        work = p
        if work is not True:
            work.wait()
        if self.rank == self.root:
            self.master.add_(self.deltas[slot])
        self.pending[slot] = None

    def begin_frame(self):
        slot = self.frame_index % len(self.deltas)
        if self.in_buffer == 0:
            self._retire(slot)
            self.deltas[slot].zero_()
        return self.deltas[slot]

    def reduce(self, spp_frame):
        """Ends a frame of spp_frame new samples per pixel; starts the collective when the buffer has its `every` frames."""
        self.total_samples += spp_frame
        self.in_buffer += 1
        if self.in_buffer >= self.every:
            self.flush()

    def flush(self):
        """Starts the reduce of the current buffer if frames have gone into it since the last one."""
        if self.in_buffer == 0:
            return
        self.in_buffer = 0
        self.reduces += 1
        slot = self.frame_index % len(self.deltas)
        buf = self.deltas[slot]
        work = True
        if self.dist.is_initialized():  # also with one rank: the collective call itself runs (RCCL has nothing to move then)
            if buf.is_cuda and self.dist.get_backend(self.group) == "gloo":
                # rehearsal path (gloo reduces host tensors only): stage through host memory
                host = buf.cpu()
                self.dist.reduce(host, dst=self.root, op=self.dist.ReduceOp.SUM, group=self.group)
                buf.copy_(host)
            else:
                work = self.dist.reduce(buf, dst=self.root, op=self.dist.ReduceOp.SUM, group=self.group, async_op=True)
        self.pending[slot] = work
        self.frame_index += 1

    def settle(self, scene, spp_frame=0):
        """Waits for `scene` (a path_tracer_amd.Scene rendering into this reducer's buffers) without losing samples:
        a batch whose ray queues overflowed adds nothing until a wait renders it again, and by then the buffer of its
        frame may already be in a reduce.  So the wait is given a cleared buffer of its own, whatever it renders again
        lands there, and that buffer goes through the exchange like a frame (of `spp_frame` new samples per pixel,
        normally none).  Every rank has to call this at the same point of its frame loop; call it before finish(),
        frame() or anything else that needs all samples in.  Until then the accumulation is short of the batches
        that overflowed, nothing else."""
        self.flush()
        scene.use_external_accum(self.begin_frame().data_ptr())
        scene.synchronize()
        self.total_samples += spp_frame
        self.in_buffer = 1
        self.flush()

    def finish(self):
        self.flush()
        for slot in range(len(self.deltas)):
            self._retire(slot)

    def frame(self):
        """accumulation / total samples on the root (src/cs488.h:1860), after all sums have landed."""
        self.finish()
        if self.rank != self.root:
            return None
        return self.master / float(max(self.total_samples, 1))
