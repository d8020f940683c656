"""Log-mel frontend: waveform -> Whisper input_features on the GPU.

Replaces WhisperFeatureExtractor._torch_extract_fbank_features
(transformers/models/whisper/feature_extraction_whisper.py:135-168; mel bank :95-103 ->
transformers/audio_utils.py:638-730 with norm="slaney", mel_scale="slaney"), which the oracle runs on the CPU.
The host side only builds constant tables (fp64 -> fp32) and owns workspace tensors; the arithmetic is
csrc/logmel.hip.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib
from .ops import _stream

N_FFT, HOP, N_BINS = 400, 160, 201
SAMPLING_RATE = 16000
N_SAMPLES = 30 * SAMPLING_RATE  # one 30 s window


def _hz_to_mel_slaney(f):
    f = np.asarray(f, dtype=np.float64)
    min_log_hertz, min_log_mel, logstep = 1000.0, 15.0, 27.0 / np.log(6.4)
    mels = 3.0 * f / 200.0
    log_region = f >= min_log_hertz
    with np.errstate(divide="ignore", invalid="ignore"):
        mels = np.where(log_region, min_log_mel + np.log(np.maximum(f, 1e-30) / min_log_hertz) * logstep, mels)
    return mels


def _mel_to_hz_slaney(m):
    m = np.asarray(m, dtype=np.float64)
    min_log_hertz, min_log_mel, logstep = 1000.0, 15.0, np.log(6.4) / 27.0
    f = 200.0 * m / 3.0
    return np.where(m >= min_log_mel, min_log_hertz * np.exp(logstep * (m - min_log_mel)), f)


def mel_filter_bank(n_mels: int = 128, n_bins: int = N_BINS, fmin=0.0, fmax=8000.0, sr=SAMPLING_RATE) -> np.ndarray:
    """Slaney-scale, slaney-normalised triangular bank, [n_bins, n_mels] float64 (audio_utils.py:638-730)."""
    mel_pts = np.linspace(_hz_to_mel_slaney(fmin), _hz_to_mel_slaney(fmax), n_mels + 2)
    filter_freqs = _mel_to_hz_slaney(mel_pts)
    fft_freqs = np.linspace(0, sr // 2, n_bins)
    fdiff = np.diff(filter_freqs)
    slopes = filter_freqs[None, :] - fft_freqs[:, None]
    down = -slopes[:, :-2] / fdiff[:-1]
    up = slopes[:, 2:] / fdiff[1:]
    fb = np.maximum(0.0, np.minimum(down, up))
    enorm = 2.0 / (filter_freqs[2: n_mels + 2] - filter_freqs[:n_mels])
    return fb * enorm[None, :]


def dft_basis(nbins_pad: int = 256):
    """hann(400, periodic)-folded real-DFT basis: cosb/sinb [400, nbins_pad] fp32 (bins >= 201 are zero)."""
    n = np.arange(N_FFT, dtype=np.float64)
    k = np.arange(N_BINS, dtype=np.float64)
    win = 0.5 - 0.5 * np.cos(2.0 * np.pi * n / N_FFT)
    ang = 2.0 * np.pi * np.outer(n, k) / N_FFT
    cosb = np.zeros((N_FFT, nbins_pad), np.float32)
    sinb = np.zeros((N_FFT, nbins_pad), np.float32)
    cosb[:, :N_BINS] = (win[:, None] * np.cos(ang)).astype(np.float32)
    sinb[:, :N_BINS] = (-win[:, None] * np.sin(ang)).astype(np.float32)
    return cosb, sinb


class LogMelFrontend:
    """GPU log-mel: ``frontend(wav[W, n_samples] fp32 cuda) -> input_features [W, n_mels, n_samples/160]``."""

    def __init__(self, device, n_mels: int = 128, nbins_pad: int = 256):
        cosb, sinb = dft_basis(nbins_pad)
        self.n_mels = n_mels
        self.nbins_pad = nbins_pad
        self.cosb = torch.from_numpy(cosb).to(device)
        self.sinb = torch.from_numpy(sinb).to(device)
        self.melT = torch.from_numpy(mel_filter_bank(n_mels).astype(np.float32)).contiguous().to(device)  # [201, n_mels]

    def __call__(self, wav: torch.Tensor, out_dtype=torch.float32) -> torch.Tensor:
        if not wav.is_cuda or wav.dtype != torch.float32:
            raise _lib.AfkError("LogMelFrontend: waveform must be a float32 HIP tensor [W, n_samples]")
        from . import custom_ops  # noqa: F401  (registers torch.ops.afk.logmel)

        return torch.ops.afk.logmel(wav.contiguous(), self.cosb, self.sinb, self.melT, self.n_mels, self.nbins_pad, out_dtype == torch.bfloat16)
