"""Diagnostic: per-phase host time (no sync) and per-phase GPU time (sync at phase ends) of one training step."""
import os, sys, time, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import os
os.environ.setdefault("TD_ALLOW_RANDOM_TEXT_ENCODER", "1")
import tubedetr_amd
from tubedetr_amd.models import build_model
from tubedetr_amd.util.misc import NestedTensor
from bench import make_batch, BatchTokenizer, WORKLOADS

T, res, k, L = WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "cfg3"]
dev = torch.device("cuda:0")
torch.manual_seed(42)
model, criterion, wd = build_model(tubedetr_amd.default_args(stride=k, compute_dtype=torch.bfloat16, video_max_len_train=200))
model.to(dev).train()
tok = BatchTokenizer(); model.transformer.tokenizer = tok
b = make_batch(T, res, k, L, 1, dev); tok.batch = b

def phases(sync):
    out = {}
    def mark(name, t0):
        if sync: torch.cuda.synchronize()
        out[name] = out.get(name, 0) + (time.perf_counter() - t0) * 1e3
        return time.perf_counter()
    model.zero_grad(set_to_none=True)
    t0 = time.perf_counter()
    samples = NestedTensor(b["frames"], b["frames_mask"]); fast = NestedTensor(b["frames_fast"], b["fast_mask"])
    # encode split: backbone slow / fast / rest
    feats, pos = model.backbone(samples); t0 = mark("backbone_slow_fwd", t0)
    with torch.no_grad(): ff, _ = model.backbone(fast)
    t0 = mark("backbone_fast_fwd", t0)
    cache = model(samples, b["durations"], ["x"], encode_and_save=True, samples_fast=fast); t0 = mark("encode_total(again)", t0)
    out_ = model(samples, b["durations"], ["x"], encode_and_save=False, memory_cache=cache); t0 = mark("decode_fwd", t0)
    tmask = torch.ones(1, T, dtype=torch.bool, device=dev)
    targets = [{"boxes": bx[None]} for bx in b["target_boxes"]]
    ld = criterion(out_, targets, b["inter_idx"], tmask); loss = sum(ld[k_] * wd[k_] for k_ in ld); t0 = mark("criterion", t0)
    loss.backward(); t0 = mark("backward", t0)
    return out

for _ in range(2): phases(True)
torch.cuda.synchronize()
g = phases(True)
torch.cuda.synchronize()
h = phases(False)
torch.cuda.synchronize()
print("phase                      gpu+host(sync) ms    host-only(no sync) ms")
for k_ in g: print(f"{k_:26s} {g[k_]:10.2f} {h[k_]:18.2f}")
print("sum", sum(g.values()), sum(h.values()))
