"""On-device audio side of the AF3 processor: window split -> log-mel on the GPU -> frame masks and
<sound>-token counts, fed straight to the encoder.

Mirrors AudioFlamingo3Processor._process_audio (transformers/models/audioflamingo3/processing_audioflamingo3.py:122-160):
30 s windows (:127-148), at most max_audio_len/30 of them (:130-137), every chunk zero-padded to 480 000 samples
(feature_extraction_whisper.py:300-307), frame mask = sample mask[::160] (:332-341), tokens per sample =
((sum_frames - 1)//2 + 1 - 2)//2 + 1 (:122-125,157-160).  The tokenizer / chat-template string work stays with the reference
processor; `expand_sound_tokens` does the id-level equivalent of "<sound>" -> "<sound>" x N (:169-171).
The log-mel arithmetic runs in csrc/logmel.hip (frontend.LogMelFrontend); the oracle computes it on the CPU.
"""
from __future__ import annotations

from typing import List, Sequence

import numpy as np
import torch

from .frontend import HOP, N_SAMPLES, LogMelFrontend


class AudioPreprocessor:
    def __init__(self, device, n_mels: int = 128, max_audio_len: int = 600, chunk_seconds: int = 30, sampling_rate: int = 16000):
        self.device = torch.device(device)
        self.window = chunk_seconds * sampling_rate
        assert self.window == N_SAMPLES
        self.max_windows = max_audio_len // chunk_seconds
        self.frontend = LogMelFrontend(self.device, n_mels)

    def __call__(self, audio: Sequence[np.ndarray], out_dtype=torch.bfloat16):
        per_sample: List[int] = []
        chunks: List[np.ndarray] = []
        for a in audio:
            a = np.asarray(a, dtype=np.float32).reshape(-1)
            n_win = min(max(1, (a.shape[0] + self.window - 1) // self.window), self.max_windows)   # :128-137
            per_sample.append(n_win)
            cap = min(a.shape[0], n_win * self.window)
            for i in range(n_win):
                chunks.append(a[i * self.window: min((i + 1) * self.window, cap)])                  # :139-143
        W = len(chunks)
        host = torch.zeros((W, self.window), dtype=torch.float32).pin_memory() if torch.cuda.is_available() else torch.zeros((W, self.window))
        lens = torch.empty(W, dtype=torch.int64)
        for i, c in enumerate(chunks):
            host[i, : c.shape[0]] = torch.from_numpy(np.ascontiguousarray(c))
            lens[i] = c.shape[0]
        wav = host.to(self.device, non_blocking=True)
        feats = self.frontend(wav, out_dtype=out_dtype)                                              # [W, n_mels, 3000]
        T = self.window // HOP
        frames = (lens + HOP - 1) // HOP                                                             # mask[::160] keeps i*160 < len
        mask = (torch.arange(T)[None, :] < frames[:, None]).to(torch.int32).to(self.device)
        per = torch.split(frames, per_sample)
        total = torch.stack([p.sum() for p in per])
        n_tok = ((total - 1) // 2 + 1 - 2) // 2 + 1                                                  # :122-125
        return {"input_features": feats, "input_features_mask": mask, "num_audio_tokens": n_tok, "windows_per_sample": per_sample}


def expand_sound_tokens(input_ids: Sequence[int], audio_token_id: int, n_tokens: int) -> List[int]:
    """id-level form of the processor's "<sound>" -> "<sound>" x N expansion (:169-171): the single placeholder is repeated"""
    out: List[int] = []
    done = False
    for t in input_ids:
        if t == audio_token_id and not done:
            out.extend([audio_token_id] * int(n_tokens))
            done = True
        else:
            out.append(int(t))
    return out
