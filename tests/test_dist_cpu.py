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
    assert du.count_ranks(dev) == world                         # bench.py's `rccl_ranks`: an all-reduce of ones
    assert du.all_agree(True, dev) is True and du.all_agree(rank == 0, dev) is False   # one failing rank -> nobody proceeds
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


def _load_bench():
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec_ = importlib.util.spec_from_file_location("bench_under_test", os.path.join(root, "bench.py"))
    mod = importlib.util.module_from_spec(spec_)
    spec_.loader.exec_module(mod)
    return mod, root


def test_bench_builds_the_rank_launcher_command_line():
    """`python bench.py --gpus N` starts its own ranks: N processes on this node, rendezvous on 127.0.0.1, the script's
    arguments passed through unchanged (the reference's launcher is `accelerate launch`, train.sh:29)."""
    import sys
    bench, root = _load_bench()
    cmd = bench.launcher_command(["--gpus", "8", "--steps", "20", "--warmup", "3"], 8, 29517)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "8"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29517"
    i = cmd.index(os.path.join(root, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "8", "--steps", "20", "--warmup", "3"]
    assert not any("exec" in c for c in cmd)


def test_bench_parent_launches_a_child_and_never_touches_the_gpu(tmp_path):
    """The parent of a self-launched run must not import torch (so it cannot initialise HIP), must start the ranks as a
    CHILD process and hand back the child's exit code.  Dry run: the command line; real run on this GPU-less container:
    both ranks refuse to run without a GPU and the parent returns their non-zero code instead of hanging."""
    import json
    import subprocess
    import sys
    _, root = _load_bench()
    probe = ("import sys, runpy; sys.argv = ['bench.py', '--gpus', '2', '--steps', '1', '--warmup', '0']\n"
             "try:\n    runpy.run_path(%r, run_name='__main__')\nexcept SystemExit as e:\n    code = e.code\n"
             "print('TORCH_IMPORTED', 'torch' in sys.modules, 'RC', code)\n" % os.path.join(root, "bench.py"))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["CTTA_BENCH_LAUNCH_DRYRUN"] = "1"
    r = subprocess.run([sys.executable, "-c", probe], capture_output=True, text=True, env=env, timeout=120)
    assert "TORCH_IMPORTED False RC 0" in r.stdout, r.stdout + r.stderr
    cmd = json.loads(r.stdout.splitlines()[0])["launcher_command"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "2"
    env.pop("CTTA_BENCH_LAUNCH_DRYRUN")
    env["CTTA_BENCH_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, "-c", probe], capture_output=True, text=True, env=env, timeout=600)
    assert "TORCH_IMPORTED False" in r.stdout, r.stdout + r.stderr
    assert "RC 0" not in r.stdout                       # the ranks' failure is the parent's exit code
    assert "bench.py needs a GPU" in r.stderr           # ... and it came from real child ranks


def test_eight_ranks_with_late_banners_keep_the_json_line_last(tmp_path, capsys):
    """`python bench.py --gpus 8` relays its child's stdout; the ONE JSON line must be the last line the parent prints even when
    eight ranks flush C-stdio banners (RCCL prints its version / library path that way) AFTER rank 0 printed it.  A stub script
    in place of bench.py, started through the real launcher path (`self_launch` -> `torch.distributed.run`, 8 processes on
    127.0.0.1): every rank leaves an unflushed C `printf` banner behind, rank 0 prints the line in the middle."""
    import argparse
    import json
    bench, root = _load_bench()
    stub = tmp_path / "stub_bench.py"
    stub.write_text(
        "import ctypes, json, os, sys, time\n"
        "rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])\n"
        "libc = ctypes.CDLL(None)\n"
        "os.write(1, ('rank %d of %d started\\n' % (rank, world)).encode())      # one write per line: ranks share the pipe\n"
        "libc.printf(b'RCCL version 2.x banner of rank %d (flushed at exit)\\n', rank)\n"
        "time.sleep(0.2 * (world - rank))\n"
        "if rank == 0:\n"
        "    os.write(1, (json.dumps({'metric': 'stub', 'value': 1.0, 'n_gpus': world, 'rccl_ranks': world}) + '\\n').encode())\n"
        "time.sleep(0.5)\n")
    env_keys = ("WORLD_SIZE", "RANK", "LOCAL_RANK", "CTTA_BENCH_LAUNCH_DRYRUN")
    saved = {k: os.environ.pop(k) for k in env_keys if k in os.environ}
    try:
        rc = bench.self_launch(argparse.Namespace(gpus=8), ["--gpus", "8"], script=str(stub))
    finally:
        os.environ.update(saved)
    out = [l for l in capsys.readouterr().out.splitlines() if l.strip()]
    assert rc == 0, out
    assert sum(1 for l in out if l.startswith("rank ") and l.endswith("started")) == 8, out
    assert sum(1 for l in out if "banner of rank" in l) == 8
    last = json.loads(out[-1])
    assert last["metric"] == "stub" and last["n_gpus"] == 8 and last["rccl_ranks"] == 8
    assert sum(1 for l in out if l.startswith('{"metric"')) == 1
