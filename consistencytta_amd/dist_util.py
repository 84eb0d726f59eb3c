"""One-process-per-GPU plumbing for the replicated (clip-sharded) inference path and the
timing protocol of bench.py.  Backend "nccl" is RCCL on ROCm; "gloo" is used by the CPU tests.
Generation shards by clip with NO data-path collective; the only collectives are
the timing barrier, the MAX-reduce of the elapsed time and the opt-in 2-scalar (max, min)
reduce (`decode_to_waveform(world_extrema=True)`) that makes the vocoder's batch-global centring
match a single-process run of the unsharded batch."""
import os

import torch
import torch.distributed as dist


def env_world():
    return (int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")),
            int(os.environ.get("LOCAL_RANK", "0")))


def init(backend, device=None):
    world, rank, _ = env_world()
    if world > 1 and not dist.is_initialized():
        kw = {"device_id": device} if (backend == "nccl" and device is not None) else {}
        dist.init_process_group(backend, **kw)
    return world, rank


def barrier(device=None):
    if device is not None and device.type == "cuda":
        torch.cuda.synchronize(device)
    if dist.is_initialized():
        dist.barrier()
    if device is not None and device.type == "cuda":
        torch.cuda.synchronize(device)


def max_over_ranks(seconds, device):
    if not dist.is_initialized():
        return float(seconds)
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def count_ranks(device):
    """An all-reduce(SUM) of ones: the number of ranks the communicator really joined (1 without a process group).  bench.py
    prints it as `rccl_ranks` so that a scaling record shows what the collective library saw, not what the flags said."""
    if not dist.is_initialized():
        return 1
    t = torch.ones(1, dtype=torch.float32, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(round(float(t.item())))


def all_agree(ok, device):
    """True iff `ok` holds on EVERY rank (one MAX all-reduce of an error flag).  Every rank must call it at the same point:
    it is how the ranks decide TOGETHER whether to enter a sequence of collectives that a one-sided failure (a hipGraph
    capture, a parity assert) would otherwise leave half-entered."""
    return max_over_ranks(0.0 if ok else 1.0, device) == 0.0


def global_wav_extrema_(max_min):
    """In place: `max_min` = [max, min] of this rank's waveforms (a 2-element float tensor, device or host) becomes the pair
    over ALL ranks.  vocoder_infer centres with batch-global extrema (hifigan/utilities.py:85), so a clip-sharded batch
    needs these two scalars to reproduce the single-process result exactly
    (`AutoencoderKL.decode_to_waveform(..., world_extrema=True)`).  One MAX all-reduce of (max, -min)."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        max_min[1].neg_()
        dist.all_reduce(max_min, op=dist.ReduceOp.MAX)
        max_min[1].neg_()
    return max_min


def allreduce_sum_(flat, bucket_elems=64 << 20):
    """In-place SUM all-reduce of a flat gradient buffer in large buckets (default 256 MiB of fp32):
    xGMI rings are per-link bound, so few big collectives beat many small ones; the buckets are issued
    asynchronously back to back and waited for together.  The 1/world factor is folded into the
    optimizer kernel (`FusedAdamW.step(grad_scale=1/world)`), saving a pass over the gradients.
    No-op when torch.distributed is not initialised (single process)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return 1
    works = []
    n = flat.numel()
    for off in range(0, n, bucket_elems):
        works.append(dist.all_reduce(flat[off:min(n, off + bucket_elems)], op=dist.ReduceOp.SUM, async_op=True))
    for w in works:
        w.wait()
    return dist.get_world_size()


class GradientBuckets:
    """Overlaps the data-parallel gradient all-reduce with the backward pass: the engine finishes the U-Net's
    blocks in reverse order (out head, up blocks, mid, down blocks, conv_in + embeddings) and reports each one;
    `ready(block)` immediately issues the asynchronous SUM all-reduce of that block's contiguous slice of the flat
    gradient buffer (RCCL runs it on its own stream behind the kernels already enqueued), `wait()` joins them
    before the optimizer.  Blocks are merged until a bucket holds at least `min_elems` elements so that the small
    level-0 blocks do not become many tiny collectives on the per-link-bound xGMI rings."""

    def __init__(self, flat_grad, ranges, min_elems=16 << 20, compress=None, twin=None):
        """`compress=torch.bfloat16` sends every bucket as bf16 (half the xGMI bytes: 1.12 GB instead of 2.24 GB per
        step) and adds the reduced values back into the fp32 buffer; default fp32 = DDP's exact sum.
        `twin`: the persistent low-precision staging buffer of the same length as `flat_grad`, owned by whoever owns the
        gradient buffer (`FusedAdamW.grad_twin(dtype)`: allocated once, freed with the optimizer); without one this
        object allocates its own and it lives exactly as long as the object (no module-level cache)."""
        self.flat, self.ranges, self.min_elems = flat_grad, dict(ranges), int(min_elems)
        self.works, self.pending = [], []
        if compress is not None and compress not in (torch.bfloat16, torch.float16):
            raise ValueError("gradient all-reduce dtype must be None (fp32), torch.bfloat16 or torch.float16")
        self.compress = compress
        if twin is not None and (twin.numel() != flat_grad.numel() or twin.dtype != compress or twin.device != flat_grad.device):
            raise ValueError("GradientBuckets: the twin buffer must match the gradient buffer's length and device and the compress dtype")
        self._twin = twin
        # CTTA_FORCE_COLLECTIVES=1 keeps the block-wise path on with a single rank (tests / profiling of the overlap)
        force = os.environ.get("CTTA_FORCE_COLLECTIVES", "0") == "1"
        self.enabled = dist.is_initialized() and (dist.get_world_size() > 1 or force)
        self.world = dist.get_world_size() if self.enabled else 1

    def _flush(self):
        # merge adjacent ranges, one collective per contiguous run
        for lo, hi in _merge(self.pending):
            if self.compress is None:
                self.works.append((dist.all_reduce(self.flat[lo:hi], op=dist.ReduceOp.SUM, async_op=True), None, lo, hi))
            else:   # a persistent low-precision twin of the flat buffer (allocated once per gradient buffer and dtype,
                # not one temporary per bucket and step)
                if self._twin is None:
                    self._twin = torch.empty(self.flat.numel(), dtype=self.compress, device=self.flat.device)
                tmp = self._twin[lo:hi]
                tmp.copy_(self.flat[lo:hi])
                self.works.append((dist.all_reduce(tmp, op=dist.ReduceOp.SUM, async_op=True), tmp, lo, hi))
        self.pending = []

    def ready(self, block):
        if not self.enabled or block not in self.ranges:
            return
        self.pending.append(self.ranges[block])
        if sum(hi - lo for lo, hi in self.pending) >= self.min_elems:
            self._flush()

    def wait(self):
        """Joins all collectives; returns the world size (the 1/world factor goes into the optimizer kernel)."""
        if self.enabled:
            self._flush()
            for w, tmp, lo, hi in self.works:
                w.wait()
                if tmp is not None:
                    self.flat[lo:hi].copy_(tmp)
        self.works = []
        return self.world


def _merge(ranges):
    out = []
    for lo, hi in sorted(ranges):
        if out and out[-1][1] == lo:
            out[-1] = (out[-1][0], hi)
        else:
            out.append((lo, hi))
    return out


class AnyRankFlag:
    """`flag` (a 0-dim / 1-element bool-like device tensor) OR-ed over all ranks, asynchronously: issued before the
    backward pass, read after it.  Used for the NaN-loss skip of the training step (tools/train_utils.py:167-172):
    the SUM all-reduce spreads one rank's NaN gradients to every rank, so every rank must skip the update together
    or the replicas diverge."""

    def __init__(self, flag):
        self.t = flag.reshape(1).to(torch.float32)
        self.work = None
        if dist.is_initialized() and dist.get_world_size() > 1:
            self.work = dist.all_reduce(self.t, op=dist.ReduceOp.MAX, async_op=True)

    def result(self):
        if self.work is not None:
            self.work.wait()
        return bool(self.t.item() > 0)


def broadcast_(flat, src=0):
    """Rank `src`'s parameters to every rank (what DDP does at wrap time)."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(flat, src=src)


def finish():
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
