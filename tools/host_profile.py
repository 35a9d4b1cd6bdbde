"""cProfile of the host side of the training step (where does the Python time go?)."""
import os, sys, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import os
os.environ.setdefault("TD_ALLOW_RANDOM_TEXT_ENCODER", "1")
import tubedetr_amd
from tubedetr_amd.models import build_model
from tubedetr_amd.harness import forward_step
from tubedetr_amd.functional import invalidate_prepared
from bench import make_batch, BatchTokenizer, WORKLOADS
T, res, k, L = WORKLOADS["cfg3"]
dev = torch.device("cuda:0")
model, criterion, wd = build_model(tubedetr_amd.default_args(stride=k, compute_dtype=torch.bfloat16))
model.to(dev).train(); model.slow_frames_are_strided_fast = True
tok = BatchTokenizer(); model.transformer.tokenizer = tok
b = make_batch(T, res, k, L, 1, dev); tok.batch = b
def step():
    invalidate_prepared(); model.zero_grad(set_to_none=True)
    loss, *_ = forward_step(model, criterion, wd, b); loss.backward()
for _ in range(3): step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(5): step()
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); ps = pstats.Stats(pr, stream=s).sort_stats("tottime"); ps.print_stats(38); print(s.getvalue()[:7000])
