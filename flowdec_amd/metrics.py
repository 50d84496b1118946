"""Parity metrics between two waveforms: the reference's own evaluation formulas, used here
to express build-vs-reference differences in the units the paper reports.  Host-side NumPy / torch-CPU, not on the hot path.

* si_sxr      -- SI-SDR / SI-SIR / SI-SAR (flowdec/eval/metrics.py:256-270, components :554-563); pinned by
                 tests/golden/g14_metrics.npz (reference code run on seeded signals).
* logspec_mse -- mean squared error of 10*log10 power spectrograms, 32 ms symmetric-Hann window / 8 ms hop
                 (eval/metrics.py:333-372).  The reference computes the spectrogram with torchaudio, which is not in
                 this image: restated with torch.stft, parity unpinned.
"""
import numpy as np
import torch


def _flat(a) -> np.ndarray:
    if isinstance(a, torch.Tensor):
        a = a.detach().cpu().numpy()
    return np.asarray(a, dtype=np.float64).reshape(-1) if np.asarray(a).dtype == np.float64 else np.asarray(a).reshape(-1)


def si_sxr_components(s_hat, s, n):
    s_target = (np.dot(s_hat, s) / np.linalg.norm(s) ** 2) * s
    e_noise = (np.dot(s_hat, n) / np.linalg.norm(n) ** 2) * n
    return s_target, e_noise, s_hat - s_target - e_noise


def si_sxr(x_hat, x, y):
    """-> (si_sdr, si_sir, si_sar) in dB; x = reference signal, y = degraded input (noise n = y - x, or y + x when that has
    less power: the reference's global-phase-flip guard)."""
    x_hat, x, y = _flat(x_hat), _flat(x), _flat(y)
    n = y - x
    if np.linalg.norm(y + x) < np.linalg.norm(y - x):
        n = y + x
    s_target, e_noise, e_art = si_sxr_components(x_hat, x, n)
    p = np.linalg.norm(s_target) ** 2
    return (10 * np.log10(p / np.linalg.norm(e_noise + e_art) ** 2), 10 * np.log10(p / np.linalg.norm(e_noise) ** 2),
            10 * np.log10(p / np.linalg.norm(e_art) ** 2))


def si_sdr(x_hat, x) -> float:
    """Plain scale-invariant SDR of x_hat against x (no noise decomposition), for build-vs-reference comparisons."""
    x_hat, x = _flat(x_hat).astype(np.float64), _flat(x).astype(np.float64)
    s_target = (np.dot(x_hat, x) / np.dot(x, x)) * x
    return float(10 * np.log10(np.dot(s_target, s_target) / max(np.dot(x_hat - s_target, x_hat - s_target), 1e-300)))


def logspec_mse(x_hat, x, sr: int = 48000, win_dur: float = 32e-3, hop_dur: float = 8e-3, eps: float = 1e-8) -> float:
    n_fft, hop = int(win_dur * sr), int(hop_dur * sr)
    win = torch.signal.windows.hann(n_fft)

    def logspec(a):
        a = torch.as_tensor(np.asarray(_flat(a), dtype=np.float32))
        S = torch.stft(a, n_fft, hop_length=hop, win_length=n_fft, window=win, center=True, pad_mode="reflect", return_complex=True)
        return 10 * torch.log10(torch.clamp(S.abs() ** 2, min=eps))
    return float(torch.mean(torch.square(logspec(x) - logspec(x_hat))))
