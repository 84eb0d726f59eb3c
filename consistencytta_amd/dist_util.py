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


def finish():
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
