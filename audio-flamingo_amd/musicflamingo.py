"""Music Flamingo on the MI355X path: the AF3 training / generate path plus the rotary TIME embedding that the
reference applies to the encoder output before the projector (transformers/models/musicflamingo/modeling_musicflamingo.py, "MF" below:
MusicFlamingoRotaryEmbedding :47-126, apply_rotary_time_emb :187-204, timestamps :331-372, call site :231-235).

Same state_dict keys as the reference (the rotary module only has non-persistent buffers), same config class
(`transformers.MusicFlamingoConfig`: rope_parameters = {rope_theta 1200, partial_rotary_factor 0.2}, audio_frame_step 0.01).
The rotation itself is the HIP kernel `afk_rotary_time` (forward and transposed-rotation backward); the angle tables are a few
thousand fp32 values per window and are built with torch index arithmetic on the device, exactly in the reference's order of
operations (fp32).  The `<sound_bos>/<sound_eos>` markers of Music Flamingo are inserted by the processor (string work, out of scope).
"""
from __future__ import annotations

import math

import torch

from . import functional as F_
from ._lib import AfkError
from .modeling import AudioFlamingo3ForConditionalGeneration


class MusicFlamingoForConditionalGeneration(AudioFlamingo3ForConditionalGeneration):
    def __init__(self, config, device="cuda", init_seed=None):
        super().__init__(config, device=device, init_seed=init_seed)
        rp = config.rope_parameters
        if rp.get("rope_type", "default") != "default":
            raise AfkError("MusicFlamingo: only the default rope_type is implemented")
        self.mf_theta = float(rp["rope_theta"])
        self.mf_max_len = float(config.max_position_embeddings)          # = rope_theta (configuration_musicflamingo.py:94)
        self.mf_frame_step = float(config.audio_frame_step)
        head_dim = getattr(config, "head_dim", None) or config.audio_config.hidden_size
        dim = int(head_dim * rp.get("partial_rotary_factor", 1.0))
        dev = self.device_
        self.mf_inv_freq = 1.0 / (self.mf_theta ** (torch.arange(0, dim, 2, dtype=torch.float32, device=dev) / dim))     # MF:88-96
        pos = torch.arange(int(self.mf_max_len), dtype=torch.float32, device=dev) / self.mf_max_len * (2 * math.pi)
        self.mf_pos_angles = torch.repeat_interleave(pos.unsqueeze(-1) * self.mf_inv_freq, 2, dim=-1)                       # MF:121-126

    def _audio_timestamps(self, input_ids, post_lengths, T3):
        """seconds [W, T3] of every encoder output row: (index of the window INSIDE its sample) * window length + 40 ms per row - what the
        reference derives from the <sound> runs of input_ids (MF:331-372).  Derivation used here (sync-free, no run extraction): windows are
        consumed in placeholder order, so window w owns the placeholder ranks [r_w, r_w + post_w) with r_w the exclusive prefix sum of the
        per-window token counts; the flat position of rank r is where the running placeholder count first reaches r + 1.  A window opens a
        new sample exactly when the token in front of its first placeholder is not a placeholder of the same row; its index inside the
        sample is the distance to the latest such opening window (a running maximum)."""
        dev = self.device_
        S = input_ids.shape[1]
        flat = input_ids.reshape(-1)
        snd = flat == self.audio_token_id
        running = torch.cumsum(snd.to(torch.int64), 0)                      # placeholders seen up to and including each position
        W = post_lengths.shape[0]
        first_rank = torch.cumsum(post_lengths, 0) - post_lengths           # exclusive prefix sum
        pos = torch.searchsorted(running, first_rank + 1).clamp_max(flat.numel() - 1)
        continues = (pos % S != 0) & snd[(pos - 1).clamp_min(0)]           # the same <sound> run carries on: same sample as window w - 1
        w = torch.arange(W, device=dev)
        opener = torch.cummax(torch.where(continues, torch.zeros_like(w), w), 0).values
        step = self.mf_frame_step * 4                                       # conv2 stride 2, avg-pool 2: 40 ms per encoder output row
        rows = torch.arange(T3, device=dev, dtype=torch.float32) * step
        return (w - opener).unsqueeze(1) * T3 * step + rows

    # MF:97-118
    def _tables(self, ts, T3):
        inv = self.mf_inv_freq
        wpos = torch.round(ts[:, 0] / (self.mf_frame_step * 4 * T3)) / self.mf_max_len
        wfreq = torch.repeat_interleave(wpos.unsqueeze(-1) * inv, 2, dim=-1)[:, None, :]
        tfreq = self.mf_pos_angles[:T3][None, :, :]
        wfreq, tfreq = torch.broadcast_tensors(wfreq, tfreq)
        freqs = torch.cat((wfreq, tfreq), dim=-1) * (-ts * 2 * math.pi).unsqueeze(-1)
        return freqs.cos().contiguous(), freqs.sin().contiguous()

    def _post_encoder(self, x, W, T3, n_tok, input_ids):
        if input_ids is None:
            raise AfkError("MusicFlamingo.get_audio_features needs input_ids (the <sound> runs give every window its time offset)")
        post = n_tok if n_tok is not None else torch.full((W,), T3, device=self.device_, dtype=torch.long)
        ts = self._audio_timestamps(input_ids.to(self.device_), post, T3)
        cos, sin = self._tables(ts, T3)
        R = cos.shape[-1]
        return F_.RotaryTimeFn.apply(x, cos.reshape(W * T3, R), sin.reshape(W * T3, R), self.arena)
