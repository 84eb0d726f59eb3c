"""world_size-2 gloo test of the one-process-per-GPU plumbing used by bench.py (timing barrier,
MAX-reduce of elapsed time, clip sharding, global waveform extrema)."""
import os
import socket

import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world), RANK=str(rank),
                      LOCAL_RANK=str(rank))
    from consistencytta_amd import dist_util as du
    dev = torch.device("cpu")
    w, r = du.init("gloo")
    assert (w, r) == (world, rank)
    du.barrier(dev)
    t = du.max_over_ranks(1.0 + rank, dev)              # slowest rank defines the step time
    wav = torch.linspace(-0.5 - 0.1 * rank, 0.3 + 0.2 * rank, 11)
    mx, mn = du.global_wav_extrema_(torch.stack([wav.max(), wav.min()])).tolist()
    # data-parallel distillation step: SUM all-reduce of the flat gradient in several buckets, 1/world folded
    # into the optimizer; rank-0 parameters broadcast at start-up
    grad = torch.arange(11, dtype=torch.float32) * (rank + 1)
    w_ = du.allreduce_sum_(grad, bucket_elems=4)
    params = torch.full((5,), float(rank + 7))
    du.broadcast_(params)
    # block-wise buckets as the backward pass reports them (last block first); small blocks merge
    g2 = torch.arange(11, dtype=torch.float32) * (rank + 1)
    buckets = du.GradientBuckets(g2, {0: (0, 3), 1: (3, 7), 2: (7, 11)}, min_elems=5)
    assert buckets.enabled
    for blk in (2, 1, 0):
        buckets.ready(blk)
    assert buckets.wait() == 2 and g2.tolist() == [3.0 * i for i in range(11)]
    # bf16-compressed buckets: half the bytes on the wire, result added back into the fp32 buffer
    g3 = torch.arange(11, dtype=torch.float32) * (rank + 1)
    b3 = du.GradientBuckets(g3, {0: (0, 3), 1: (3, 7), 2: (7, 11)}, min_elems=5, compress=torch.bfloat16)
    for blk in (2, 1, 0):
        b3.ready(blk)
    assert b3.wait() == 2 and g3.tolist() == [3.0 * i for i in range(11)] and g3.dtype == torch.float32
    # NaN-loss skip: one rank's flag reaches every rank (all skip the update together)
    flag_any = du.AnyRankFlag(torch.tensor(rank == 1)).result()
    flag_none = du.AnyRankFlag(torch.tensor(False)).result()
    assert flag_any is True and flag_none is False
    q.put((rank, t, None, mx, mn, w_, grad.tolist(), params.tolist()))
    du.finish()


def test_two_rank_protocol():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[1] for r in res] == [2.0, 2.0]
    for r in res:
        assert abs(r[3] - 0.5) < 1e-6 and abs(r[4] + 0.6) < 1e-6  # extrema agree on all ranks
        assert r[5] == 2 and r[6] == [3.0 * i for i in range(11)]   # (1 + 2) * i summed over both ranks, all buckets
        assert r[7] == [7.0] * 5                                    # everyone holds rank 0's parameters


def test_allreduce_is_identity_without_a_process_group():
    from consistencytta_amd import dist_util as du
    g = torch.arange(6, dtype=torch.float32)
    assert du.allreduce_sum_(g) == 1 and g.tolist() == [0.0, 1.0, 2.0, 3.0, 4.0, 5.0]
