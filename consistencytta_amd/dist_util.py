"""One-process-per-GPU plumbing for the replicated (clip-sharded) inference path and the
timing protocol of bench.py.  Backend "nccl" is RCCL on ROCm; "gloo" is used by the CPU tests.
Generation shards by clip with NO data-path collective; the only collectives are
the timing barrier, the MAX-reduce of the elapsed time and the optional 2-scalar (max, min)
reduce that makes the vocoder's batch-global centring match a single-process run."""
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


def shard_clips(n_clips, world, rank):
    """Round-robin clip ownership: clip i belongs to rank i % world."""
    return list(range(rank, n_clips, world))


def global_wav_extrema(local_max, local_min, device):
    """(max, min) over ALL ranks' waveforms: vocoder_infer centres with batch-global extrema
    (hifigan/utilities.py:85), so a sharded batch needs these two scalars to match exactly."""
    t = torch.tensor([local_max, -local_min], dtype=torch.float32, device=device)
    if dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t[0]), -float(t[1])


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


def broadcast_(flat, src=0):
    """Rank `src`'s parameters to every rank (what DDP does at wrap time)."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(flat, src=src)


def finish():
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
