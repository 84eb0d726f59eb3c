"""Backward pass of the student U-Net: every parameter gradient the HIP engine
produces is compared with torch autograd over the CPU oracle's forward (fp32), on the tiny
config of the golden cases.

Tolerance: activations and activation-gradients travel in bf16 (fp32 accumulation), the oracle is
fp32 end to end.  Per parameter tensor the relative L2 error must stay below GRAD_REL_L2; the
error over the whole concatenated gradient below GRAD_REL_L2_ALL.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import cases  # noqa: E402
from consistencytta_amd import modules, spec  # noqa: E402
from gpu_util import DEV, bf16_round  # noqa: E402
from oracle import nets as onets  # noqa: E402

GRAD_REL_L2 = 8e-2
GRAD_REL_L2_ALL = 4e-2


def _oracle_grads(cfg, sd, x, ts, gs, enc, mask, dout, guided=True):
    params = {k: v.clone().requires_grad_(k != "guidance_proj.weight") for k, v in sd.items()}
    out = onets.unet_forward(cfg, params, x, ts, gs if guided else None, enc, mask)
    (out * dout).sum().backward()
    return out.detach(), {k: p.grad for k, p in params.items() if p.requires_grad}


def _compare(grads_hip, grads_ref):
    num = den = 0.0
    worst = ("", 0.0)
    rows = []
    for k, r in grads_ref.items():
        g = grads_hip[k].detach().float().cpu()
        e = float((g - r).norm())
        n = float(r.norm())
        num += e * e
        den += n * n
        rel = e / max(n, 1e-30)
        rows.append((rel, k, n))
        if rel > worst[1]:
            worst = (k, rel)
    rows.sort(reverse=True)
    for rel, k, n in rows[:12]:
        print("  %-70s rel_l2 %.3e  |ref| %.3e" % (k, rel, n))
    total = (num / den) ** 0.5
    print("all parameters: rel_l2 %.3e ; worst %s %.3e" % (total, worst[0], worst[1]))
    return total, worst


@pytest.mark.parametrize("B,H,W,L", [(2, 32, 8, 7), (3, 16, 8, 5)])
def test_unet_parameter_gradients_match_autograd(B, H, W, L):
    cfg = cases.TINY_UNET
    sd = cases.unet_weights(cfg, True, 1)
    x, ts, gs, enc, mask = cases.unet_inputs(cfg, B, H, W, L, "train_tiny")
    dout = bf16_round(cases.t(spec.det_uniform("train.dout", (B, cfg["out_channels"], H, W), 21))) * 0.01
    ref_out, ref = _oracle_grads(cfg, sd, x, ts, gs, enc, mask, dout)

    m = modules.UNet2DConditionGuidedModel.from_config(cfg)
    m.load_state_dict(sd)
    m = m.to(DEV)
    out = m.forward_train(x.to(DEV), ts.to(DEV), gs.to(DEV), enc.to(DEV), mask.to(DEV))
    rel_out = float((out.cpu() - ref_out).norm() / ref_out.norm())
    assert rel_out < 2.5e-2, rel_out
    m.backward(dout.to(DEV))
    torch.cuda.synchronize()
    grads = {k: p.grad for k, p in m.named_parameters() if p.requires_grad}
    assert set(grads) == set(ref)
    assert m.get_parameter("guidance_proj.weight").grad is None       # embeddings.py:229 requires_grad=False
    total, worst = _compare(grads, ref)
    assert np.isfinite(total) and total <= GRAD_REL_L2_ALL
    assert worst[1] <= GRAD_REL_L2, worst

    # .grad semantics: a second forward/backward accumulates
    m.forward_train(x.to(DEV), ts.to(DEV), gs.to(DEV), enc.to(DEV), mask.to(DEV))
    m.backward(dout.to(DEV))
    torch.cuda.synchronize()
    k0 = "down_blocks.1.resnets.0.conv1.weight"
    assert float((grads[k0].cpu() - 2 * ref[k0]).norm() / (2 * ref[k0]).norm()) <= GRAD_REL_L2
    # and the plain forward still works on the training handle
    with torch.no_grad():
        out2 = m(x.to(DEV), ts.to(DEV), guidance=gs.to(DEV), encoder_hidden_states=enc.to(DEV),
                 encoder_attention_mask=mask.to(DEV)).sample
    assert float((out2 - out).abs().max()) == 0.0


def test_backward_requires_a_training_forward():
    cfg = cases.TINY_UNET
    m = modules.UNet2DConditionGuidedModel.from_config(cfg).init_deterministic(1).to(DEV)
    x, ts, gs, enc, mask = cases.unet_inputs(cfg, 1, 16, 8, 4, "train_err")
    m.enable_training = True
    m(x.to(DEV), ts.to(DEV), guidance=gs.to(DEV), encoder_hidden_states=enc.to(DEV))
    from consistencytta_amd import _native as N
    with pytest.raises(N.CttaError, match="no training forward"):
        m.backward(torch.zeros(1, cfg["out_channels"], 16, 8))
