"""CPU restatement of the waveform-domain training losses (TEST INFRASTRUCTURE ONLY: imported by tests/ alone).

`tools/losses.py:146-169` STFTLoss.stft -- torch.stft(x.double(), fft_size, shift_size, win_length, hann_window,
return_complex=True), magnitude = sqrt(clamp(re^2 + im^2, 1e-8)), transposed to (B, frames, bins), float32;
`:72-92` SpectralConvergengeLoss -- ||y - x||_F / ||y||_F per instance; `:95-120` LogSTFTMagnitudeLoss -- mean |log y - log x|
per instance; `:187-256` MultiResolutionSTFTLoss -- factor_mse * mse(latents) + factor_mag * mean_r(mag) + factor_sc *
mean_r(sc) over the resolutions (1024, 120, 600), (2048, 240, 1200), (512, 50, 240).  Pinned by being the reference's own
torch calls (torch.stft is the reference's STFT); the product path computes the magnitudes on csrc/stft_loss.hip."""
import torch

RESOLUTIONS = ((1024, 120, 600), (2048, 240, 1200), (512, 50, 240))


def stft_magnitude(x, fft_size, shift_size, win_length):
    spec = torch.stft(x.double(), fft_size, shift_size, win_length, torch.hann_window(win_length).double().to(x.device),
                      return_complex=True)
    return torch.clamp(spec.real ** 2 + spec.imag ** 2, min=1e-8).sqrt().transpose(2, 1).float()


def multi_resolution_stft_terms(x_wav, y_wav, resolutions=RESOLUTIONS):
    """(sc, mag): per-instance spectral-convergence and log-magnitude terms averaged over the resolutions."""
    sc = mag = 0.
    for fft, hop, win in resolutions:
        xm, ym = stft_magnitude(x_wav, fft, hop, win), stft_magnitude(y_wav, fft, hop, win)
        B = xm.shape[0]
        sc = sc + (ym - xm).reshape(B, -1).norm(dim=1) / ym.reshape(B, -1).norm(dim=1)
        mag = mag + (torch.log(ym) - torch.log(xm)).abs().reshape(B, -1).mean(dim=1)
    return sc / len(resolutions), mag / len(resolutions)


def multi_resolution_stft_loss(pred_latent, target_latent, pred_wav, target_wav, factor_sc=0.2, factor_mag=0.2, factor_mse=1.,
                               sr=16000):
    B = pred_latent.shape[0]
    mse = ((pred_latent.float() - target_latent.float()) ** 2).reshape(B, -1).mean(dim=1)
    sc, mag = multi_resolution_stft_terms(pred_wav[:, :sr * 10], target_wav[:, :sr * 10])
    return factor_mse * mse + factor_mag * mag + factor_sc * sc
