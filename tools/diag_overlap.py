import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_model_gpu import _model, G
from audio_flamingo_amd.arena import FusedAdamW
from audio_flamingo_amd.dp import BackwardOverlap
dev = torch.device("cuda")
g = torch.load(os.path.join(G, "tiny64_caseB.pt"))
kw = dict(input_ids=g["ids"].to(dev), input_features=g["feats"].to(dev), input_features_mask=g["fmask"].to(dev), labels=g["labels"].to(dev))
def run(mode):
    m = _model(dev)
    opt = FusedAdamW(m.arena, lr=1e-3, weight_decay=0.01)
    if mode in ("wgrad", "both"): m.arena.enable_wgrad_stream(True)
    ov = BackwardOverlap(m.arena, opt) if mode in ("overlap", "both") else None
    m.zero_grad()
    if ov: ov.begin_step()
    loss = m(**kw).loss
    loss.backward()
    if ov: ov.finish()
    else:
        m.arena.join_streams()
    torch.cuda.synchronize()
    return m, float(loss)
ref, l0 = run("serial")
for mode in ("serial", "wgrad", "overlap", "both"):
    for rep in range(2):
        m, l = run(mode)
        bad = []
        for b0, b1 in zip(ref.arena.order, m.arena.order):
            if not torch.equal(b0.grad, b1.grad):
                d = (b0.grad.float() - b1.grad.float()).abs().max().item()
                bad.append((b0.key.replace("model.", "")[-45:], round(d, 6), round(b0.grad.float().abs().max().item(), 6)))
        print(mode, rep, "loss equal", l == l0, "n_bad", len(bad), bad[:6], flush=True)
