"""CPU restatement of the task-level arithmetic of `models.AudioLCM` (test oracle):
`_query_teacher` (models/audio_distilled_model.py:286-322), `forward` in training and
validation mode (models/audio_consistency_model.py:239-427) and `inference` (:429-548),
on top of oracle.nets / oracle.heun.  Random draws are passed in (the reference draws them from
the global RNG, :284-286,312,326,478)."""
import torch

from . import heun
from .nets import unet_forward


class Nets:
    """cfg + the four state dicts of an AudioLCM (teacher is the unguided U-Net)."""

    def __init__(self, cfg, teacher, student, target, ema):
        self.cfg, self.teacher, self.student, self.target, self.ema = cfg, teacher, student, target, ema


def query_teacher(n, z_scaled, t, embeds_cf, mask_cf, w):
    """CFG teacher query: batch doubled, (1-w)*uncond + w*cond with per-sample w."""
    t = t if torch.is_tensor(t) else torch.tensor([t], dtype=torch.float64)
    t2 = torch.cat([t.reshape(-1)] * 2) if t.numel() != 1 else t.reshape(-1)
    pred = unet_forward(n.cfg, n.teacher, torch.cat([z_scaled] * 2), t2 if t2.numel() > 1 else float(t2[0]), None,
                        embeds_cf, mask_cf)
    u, c = pred.chunk(2)
    return heun.cfg_combine(u, c, w if torch.is_tensor(w) else torch.full((z_scaled.shape[0],), float(w)))


def _tables(num_steps=18):
    ts, sig = heun.set_timesteps(num_steps)
    return torch.from_numpy(ts), torch.from_numpy(sig)


def _teacher_two_queries(n, P, z0, noise, inds, w, ts, sig):
    """Shared front half of forward(): noising, 2 teacher queries and the Heun update (:307-351)."""
    t_np1, t_n = ts[inds], ts[inds + 2]
    s_np1, s_n = sig[inds], sig[inds + 1]            # sigma at t_{n+1} and the next sigma (= sigma at t_n)
    z_noisy = heun.add_noise(z0, noise, s_np1)
    z_gauss = noise * sig.max()
    last = (t_np1 == ts.max()).reshape(-1, 1, 1, 1)
    z_np1 = torch.where(last, z_gauss, z_noisy)
    z_np1_scaled = heun.scale_model_input(z_np1, s_np1)
    v1 = query_teacher(n, z_np1_scaled, t_np1, P["embeds_cf"], P["mask_cf"], w)
    zhat, d, dt = heun.step_first(v1, z_np1, s_np1, s_n)
    zhat_scaled = heun.scale_model_input(zhat, s_n)
    v2 = query_teacher(n, zhat_scaled, t_n, P["embeds_cf"], P["mask_cf"], w)
    zhat = heun.step_second(v2, zhat, s_n, z_np1, d, dt)
    return z_np1_scaled, t_np1, zhat, heun.scale_model_input(zhat, s_n), t_n, s_np1


def distill_loss(n, P, z0, noise, time_inds, w, snr_gamma=5.0):
    """Training-mode forward(): time_inds are the even indices t_{n+1} = timesteps[time_inds]."""
    ts, sig = _tables()
    z_np1_scaled, t_np1, zhat, zhat_scaled, t_n, s_np1 = _teacher_two_queries(n, P, z0, noise, time_inds, w, ts, sig)
    target = unet_forward(n.cfg, n.target, zhat_scaled, t_n, w, P["embeds"], P["mask"])
    target = torch.where((t_n == 0).reshape(-1, 1, 1, 1), z0, target)
    pred = unet_forward(n.cfg, n.student, z_np1_scaled, t_np1, w, P["embeds"], P["mask"])
    return heun.snr_mse_loss(pred, target, s_np1, snr_gamma)


def validation_losses(n, P, z0, noise, validation_mode, w, snr_gamma=5.0, run_teacher=True):
    """validation-mode forward() (:354-405): (loss_w_gt, loss_w_teacher, loss_consis, loss_teacher)."""
    ts, sig = _tables()
    B = z0.shape[0]
    ti = len(ts) - 1 - int(validation_mode * 2)
    inds = torch.full((B,), ti, dtype=torch.int64)
    z_np1_scaled, t_np1, zhat, zhat_scaled, t_n, s_np1 = _teacher_two_queries(n, P, z0, noise, inds, w, ts, sig)
    from_np1 = unet_forward(n.cfg, n.target, z_np1_scaled, t_np1, w, P["embeds"], P["mask"])
    from_n = unet_forward(n.cfg, n.target, zhat_scaled, t_n, w, P["embeds"], P["mask"])
    if run_teacher:   # the teacher continues from t_n to 0 with the full Heun schedule
        first = True
        stored = None
        for j in range(ti + 2, len(ts)):
            s = sig[j:j + 1].expand(B) if first else sig[j:j + 1].expand(B)
            tj = ts[j:j + 1].expand(B)
            if first:
                sj, sj1 = sig[j].expand(B), sig[j + 1].expand(B)
                v = query_teacher(n, heun.scale_model_input(zhat, sj), tj, P["embeds_cf"], P["mask_cf"], w)
                nxt, d, dt = heun.step_first(v, zhat, sj, sj1)
                stored = (zhat, d, dt, sj1)
                zhat = nxt
            else:
                x0, d, dt, sj1 = stored
                v = query_teacher(n, heun.scale_model_input(zhat, sj1), tj, P["embeds_cf"], P["mask_cf"], w)
                zhat = heun.step_second(v, zhat, sj1, x0, d, dt)
            first = not first
    mse = torch.nn.functional.mse_loss
    loss_consis = heun.snr_mse_loss(from_np1, from_n, s_np1, snr_gamma)
    return mse(from_np1, z0), mse(from_np1, zhat), loss_consis, mse(zhat, z0)


def inference_student(n, sd, P, noise, w_in, w_post, num_steps, renoise=None):
    """inference() student part: 1 query at t=999, then (num_steps-1) re-noise + query rounds."""
    use_cf = w_post > 1.0
    enc, mask = (P["embeds_cf"], P["mask_cf"]) if use_cf else (P["embeds"], P["mask"])
    ts, sig = _tables(18)

    def calc(z, t, sigma):
        zin = torch.cat([z] * 2) if use_cf else z
        zin = heun.scale_model_input(zin, sigma.expand(zin.shape[0]))
        zh = unet_forward(n.cfg, sd, zin, float(t), float(w_in), enc, mask)
        if use_cf:
            u, c = zh.chunk(2)
            zh = (1 - w_post) * u + w_post * c
        return zh

    z = calc(noise * sig.max(), ts[0], sig[0:1])
    ts2, sig2 = _tables(num_steps)
    k = 0
    for j in range(1, len(ts2), 2):
        s = sig2[j + 1:j + 2]   # index_for_timestep takes the LAST match of the duplicated timestep
        zn = heun.add_noise(z, renoise[k] if isinstance(renoise, (list, tuple)) else renoise, s.expand(z.shape[0]))
        z = calc(zn, ts2[j], s)
        k += 1
    return z


def inference_teacher(n, P, noise, w, num_teacher_steps):
    ts, sig = _tables(num_teacher_steps)
    B = noise.shape[0]
    z = noise * sig.max()
    first, stored = True, None
    for j in range(len(ts)):
        tj = float(ts[j])
        if first:
            sj, sj1 = sig[j].expand(B), sig[j + 1].expand(B)
            v = query_teacher(n, heun.scale_model_input(z, sj), tj, P["embeds_cf"], P["mask_cf"], w)
            nxt, d, dt = heun.step_first(v, z, sj, sj1)
            stored = (z, d, dt, sj1)
            z = nxt
            # the very first timestep appears once: the scheduler stays in 2nd-order state afterwards
        else:
            x0, d, dt, sj1 = stored
            v = query_teacher(n, heun.scale_model_input(z, sj1), tj, P["embeds_cf"], P["mask_cf"], w)
            z = heun.step_second(v, z, sj1, x0, d, dt)
        first = not first
    return z


# ----------------------------------------------------------------------------------- stage 1 (AudioGDM)
def gdm_loss(n, P, z0, noise, time_inds, w, snr_gamma=5.0):
    """AudioGDM.forward (models/audio_guided_model.py:87-164): DDPM noising at t = timesteps[time_inds], one CFG teacher
    query, the guided student's prediction, MSE weighted by min(snr, gamma) / (snr + 1) per instance."""
    from . import ddim
    ac = ddim.alphas_cumprod()
    ts = ddim.ddpm_timesteps()
    t_n = ts[time_inds]
    z_noisy = ddim.add_noise(z0, noise, t_n, ac)
    last = (t_n == ts.max()).reshape(-1, 1, 1, 1)
    z_n = torch.where(last, noise * 1.0, z_noisy)                 # init_noise_sigma = 1; scale_model_input = identity
    teacher = query_teacher(n, z_n, t_n, P["embeds_cf"], P["mask_cf"], w)
    student = unet_forward(n.cfg, n.student, z_n, t_n, w, P["embeds"], P["mask"])
    inst = ((student - teacher) ** 2).mean(dim=(1, 2, 3))
    return (inst * ddim.gdm_loss_weights(t_n, snr_gamma, ac)).mean()


def gdm_inference(n, sd, P, noise, w_in, num_steps):
    """AudioGDM.inference (:166-244) with guidance_scale_post = 1: num_steps DDIM steps of the guided student."""
    from . import ddim
    ac = ddim.alphas_cumprod()
    z = noise * 1.0
    B = noise.shape[0]
    for t in ddim.ddim_timesteps(num_steps):
        v = unet_forward(n.cfg, sd, z, float(t), float(w_in), P["embeds"], P["mask"])
        z = ddim.ddim_step(v, torch.full((B,), int(t)), z, num_steps, ac)
    return z


def mel_loss_instances(dd, sd, pred, target, scale_factor, mse_weight=.7, mel_weight=.3):
    """MelLoss.forward (tools/losses.py:47-64), reduction='instance': 0.3 * mse(decode(pred), decode(target)) +
    0.7 * mse(pred, target), each a mean over everything but the batch axis.  Differentiable in `pred`."""
    from .nets import vae_decode
    mel_p = vae_decode(dd, sd, pred.float(), scale_factor)
    mel_t = vae_decode(dd, sd, target.float(), scale_factor)
    l_mel = ((mel_p - mel_t) ** 2).reshape(pred.shape[0], -1).mean(dim=1) * mel_weight
    l_lat = ((pred.float() - target.float()) ** 2).reshape(pred.shape[0], -1).mean(dim=1) * mse_weight
    return l_mel + l_lat
