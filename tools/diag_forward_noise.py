#!/usr/bin/env python
"""Where does forward rounding noise enter?  Depth-reduced full-width AF3: projector output and every decoder hidden state of (a) the reference in
bf16 and (b) this repo's model, each against the reference in fp32 (same state_dict, same batch as tools/parity_fulldepth.py).  Diagnostic only."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from tools.parity_fulldepth import _reference_features, _rel, BF, restore_rope_buffers

def main(enc=2, dec=2, B=4):
    from transformers import AudioFlamingo3ForConditionalGeneration as Ref
    dev = torch.device("cuda", 0)
    cfg = bench.af3_7b_config(enc, dec)
    waves, ids, labels = bench.synthetic_batch(B, 0, dev, 1)
    feats_ref, fmask = (t.to(dev) for t in _reference_features(waves.cpu().numpy()))
    torch.manual_seed(0)
    with torch.device(dev):
        ref = Ref(cfg)
    g = torch.Generator(device=dev).manual_seed(4)
    with torch.no_grad():
        for k, p in ref.named_parameters():
            if k.endswith(".bias"):
                p.copy_(0.02 * torch.randn(p.shape, device=dev, generator=g))
            elif "norm" in k.split(".")[-2] and k.endswith(".weight"):
                p.copy_(1 + 0.05 * torch.randn(p.shape, device=dev, generator=g))
    ref.to(BF)
    sd = {k: v.detach().clone() for k, v in ref.state_dict().items()}

    def ref_run(dtype, feats):
        restore_rope_buffers(ref.to(dtype))
        acts = {}
        hooks = []
        at = ref.model.audio_tower
        def hk(name):
            def f(mod, inp, out):
                acts[name] = (out[0] if isinstance(out, tuple) else out).detach().float()
            return f
        hooks.append(at.conv1.register_forward_hook(hk("enc.conv1")))
        hooks.append(at.conv2.register_forward_hook(hk("enc.conv2")))
        for i, l in enumerate(at.layers):
            hooks.append(l.register_forward_hook(hk(f"enc.layer{i}")))
            hooks.append(l.self_attn.register_forward_hook(hk(f"enc.layer{i}.attn")))
        hooks.append(at.register_forward_hook(lambda m, i, o: acts.__setitem__("enc.out", (o.last_hidden_state if hasattr(o, "last_hidden_state") else o[0]).detach().float())))
        hooks.append(ref.model.multi_modal_projector.register_forward_hook(hk("projector")))
        for i, l in enumerate(ref.model.language_model.layers):
            hooks.append(l.self_attn.register_forward_hook(hk(f"dec.layer{i}.attn")))
            hooks.append(l.mlp.register_forward_hook(hk(f"dec.layer{i}.mlp")))
        with torch.no_grad():
            out = ref(input_ids=ids, input_features=feats.to(dtype), input_features_mask=fmask, labels=labels, output_hidden_states=True)
        for h in hooks:
            h.remove()
        for i, h in enumerate(out.hidden_states):
            acts[f"dec.hidden{i}"] = h.detach().float()
        acts["logits"] = out.logits.detach().float()
        return acts

    a32 = ref_run(torch.float32, feats_ref)
    a16 = ref_run(BF, feats_ref)
    from audio_flamingo_amd.frontend import LogMelFrontend
    from audio_flamingo_amd.modeling import AudioFlamingo3ForConditionalGeneration as Mine
    fe = LogMelFrontend(dev)
    my_feats = fe(waves, out_dtype=BF)
    a16m = ref_run(BF, my_feats.float())       # the reference in bf16 on OUR features: isolates the frontend difference
    del ref
    m = Mine(cfg, device=dev, init_seed=0)
    m.load_state_dict(sd)
    mine = {}
    with torch.no_grad():
        for name, f in (("ours", my_feats), ("ours_on_ref_feats", feats_ref.to(BF))):
            out = m(input_ids=ids, input_features=f, labels=labels, return_logits=True, output_hidden_states=True)
            d = {f"dec.hidden{i}": h.float() for i, h in enumerate(out.hidden_states)}
            d["logits"] = out.logits.float()
            d["projector"] = out.audio_hidden_states.float()
            mine[name] = d
    print(f"{'stage':24s} {'ref bf16':>10s} {'ref bf16/our feats':>20s} {'ours':>10s} {'ours/ref feats':>16s}")
    for k in a32:
        row = [_rel(a16[k], a32[k]), _rel(a16m[k], a32[k])]
        for name in ("ours", "ours_on_ref_feats"):
            if k in mine[name]:
                x = mine[name][k]
                row.append(_rel(x.reshape(a32[k].shape) if x.numel() == a32[k].numel() else x, a32[k]) if x.numel() == a32[k].numel() else float("nan"))
            else:
                row.append(float("nan"))
        print(f"{k:24s} {row[0]:10.5f} {row[1]:20.5f} {row[2]:10.5f} {row[3]:16.5f}")

if __name__ == "__main__":
    main(*(int(x) for x in sys.argv[1:]))
