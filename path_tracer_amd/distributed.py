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
    """Per-frame exchange step.  `delta` is this rank's accumulation of ONE frame (the renderer
    writes into it; cleared before every frame), `master` the progressive accumulation that only
    the root keeps.  reduce() sums all ranks' deltas onto the root and adds the sum to master."""

    def __init__(self, dist, delta, master=None, root=0, group=None):
        self.dist, self.delta, self.master, self.root, self.group = dist, delta, master, root, group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.total_samples = 0
        if self.rank == root and master is None:
            raise ValueError("the root rank needs a master accumulation buffer")

    def begin_frame(self):
        self.delta.zero_()

    def reduce(self, spp_frame):
        if self.dist.is_initialized() and self.dist.get_world_size(self.group) > 1:
            if self.delta.is_cuda and self.dist.get_backend(self.group) == "gloo":
                # rehearsal path (gloo reduces host tensors only): stage through host memory
                host = self.delta.cpu()
                self.dist.reduce(host, dst=self.root, op=self.dist.ReduceOp.SUM, group=self.group)
                self.delta.copy_(host)
            else:
                self.dist.reduce(self.delta, dst=self.root, op=self.dist.ReduceOp.SUM, group=self.group)
        self.total_samples += spp_frame
        if self.rank == self.root:
            self.master.add_(self.delta)

    def frame(self):
        """accumulation / total samples on the root (src/cs488.h:1860)."""
        if self.rank != self.root:
            return None
        return self.master / float(max(self.total_samples, 1))
