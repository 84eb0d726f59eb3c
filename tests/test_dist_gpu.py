"""Two data-parallel ranks on ONE GPU (collectives through gloo): the N > 1 `train_step` path on the HIP engines
(tools/train_utils.py:152-183 under accelerate's DDP).  Each rank is a fresh child process."""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _launch(mode, out_dir):
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        os.makedirs(out_dir, exist_ok=True)
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "dist_gpu_worker.py"), mode, str(out_dir)],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    logs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append(out.decode(errors="replace")[-3000:])
    for p, log in zip(procs, logs):
        assert p.returncode == 0, log
    return [torch.load(os.path.join(out_dir, "rank%d.pt" % r)) for r in range(2)]


def _single_process_reference():
    import dist_gpu_worker as W
    dev = torch.device("cuda:0")
    m = W.build(dev, seed_shift=0)          # rank 0's weights: what both ranks hold after the broadcast
    m.train()
    P, z0, draws = W.global_batch()
    Ps, z, kw = W.shard(P, z0, draws, 0, W.GLOBAL_B, dev)
    return W.run_step(m, Ps, z, kw)


def test_two_rank_train_step_equals_one_process_on_the_global_batch(tmp_path):
    r0, r1 = _launch("plain", tmp_path)
    loss, grad, after, before, _ = _single_process_reference()
    # rank 0's weights reached rank 1, both ranks ended with identical parameters and shadows
    assert torch.equal(r0["before"], r1["before"]) and torch.equal(r0["before"], before.cpu())
    assert torch.equal(r0["after"], r1["after"]) and torch.equal(r0["target_after"], r1["target_after"])
    assert torch.equal(r0["grad"], r1["grad"])
    # mean of the two shard losses = loss of the global batch; averaged gradients = its gradient
    assert abs(0.5 * (r0["loss"] + r1["loss"]) - loss) <= 2e-3 * abs(loss)
    g = grad.cpu()
    rel = float((r0["grad"] - g).norm() / g.norm())
    print("2 ranks x B=2 vs 1 process x B=4: averaged gradient rel_l2 %.3e" % rel)
    assert rel <= 1e-5           # a sample's result does not depend on its batch mates: only the summation order differs
    d_ref, d_two = after.cpu() - before.cpu(), r0["after"] - r0["before"]
    upd = float((d_ref - d_two).norm() / d_ref.norm())
    print("parameter update rel diff %.3e" % upd)
    assert upd <= 1e-3


def test_nan_on_one_rank_skips_the_update_on_every_rank(tmp_path):
    r0, r1 = _launch("nan", tmp_path)
    assert r1["loss"] != r1["loss"] and r0["loss"] == r0["loss"]        # only rank 1 saw the NaN
    for r in (r0, r1):
        assert r["grad"] is None and r["step_count"] == 0              # AdamW never ran
        assert torch.equal(r["after"], r["before"])
    assert torch.equal(r0["after"], r1["after"])                        # replicas still bit-identical


def test_bf16_gradient_allreduce_option(tmp_path):
    r0, r1 = _launch("bf16", tmp_path)
    _, grad, _, _, _ = _single_process_reference()
    assert torch.equal(r0["grad"], r1["grad"]) and torch.equal(r0["after"], r1["after"])
    g = grad.cpu()
    rel = float((r0["grad"] - g).norm() / g.norm())
    print("bf16-compressed all-reduce: averaged gradient rel_l2 %.3e" % rel)
    assert rel <= 2.5e-2


def test_two_rank_segmented_graph_step_equals_the_eager_two_rank_step(tmp_path):
    """the data-parallel step as segmented hipGraph replays (forward + loss + out head | one graph per
    backward block, the bucket all-reduce issued between two replays) against the eager 2-rank `train_step` on the same
    shards and draws: per-rank loss bit-identical, the averaged gradient equal up to the LayerNorm-atomics round-off
    (the run-to-run noise floor of the eager step itself), parameters bit-identical ACROSS ranks after the update."""
    e0, e1 = _launch("plain", tmp_path / "eager")
    g0, g1 = _launch("graph", tmp_path / "graph")
    assert g0["loss"] == e0["loss"] and g1["loss"] == e1["loss"]
    assert torch.equal(g0["grad"], g1["grad"]) and torch.equal(g0["after"], g1["after"])
    assert torch.equal(g0["target_after"], g1["target_after"]) and torch.equal(g0["before"], e0["before"])
    rel = float((g0["grad"] - e0["grad"]).norm() / e0["grad"].norm())
    print("segmented replay vs eager, 2 ranks: averaged gradient rel diff %.3e" % rel)
    assert rel <= 1e-7
    d_e, d_g = e0["after"] - e0["before"], g0["after"] - g0["before"]
    assert float((d_e - d_g).norm() / d_e.norm()) <= 1e-3
    assert g0["step_count"] == 1


def test_sharded_vocoder_centring_with_world_extrema_equals_the_unsharded_batch(tmp_path):
    """`decode_to_waveform(world_extrema=True)` -- two ranks hold two clips each, exchange (max, -min)
    (dist_util.global_wav_extrema_) and centre with the WHOLE batch's pair: the int16 shards equal a single process on the
    four clips bit for bit (hifigan/utilities.py:85 centres with batch-global extrema).  The default (per-shard centring,
    no data-path collective) is the documented deviation and differs on the rank that does not hold the loudest clip."""
    import numpy as np
    import dist_gpu_worker as W
    r0, r1 = _launch("wav", tmp_path)
    v = W.tiny_vocoder(torch.device("cuda:0"))
    ref = v.decode_to_waveform(W.wav_batch().to("cuda:0"))
    assert np.array_equal(torch.cat([r0["world"], r1["world"]]).numpy(), ref)
    assert not np.array_equal(torch.cat([r0["local"], r1["local"]]).numpy(), ref)
