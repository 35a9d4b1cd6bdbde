"""Where do the step's copies / adds / fills come from?  One eager training step under torch.profiler with Python stacks;
prints, per aten op of interest, the innermost tubedetr_amd frame, launch count and device time.  usage: copy_sources.py [clips]"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("TD_ALLOW_RANDOM_TEXT_ENCODER", "1")
import torch
import tubedetr_amd
from tubedetr_amd.models import build_model
from tubedetr_amd.harness import forward_step
from bench import make_batch, BatchTokenizer, WORKLOADS
from torch.profiler import profile, ProfilerActivity

T, res, k, L = WORKLOADS["cfg3"]
clips = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device("cuda:0")
model, criterion, wd = build_model(tubedetr_amd.default_args(stride=k, compute_dtype=torch.bfloat16))
model.to(dev).train()
tok = BatchTokenizer(); model.transformer.tokenizer = tok
b = make_batch(T, res, k, L, 1, dev, clips=clips); tok.batch = b
def step():
    model.zero_grad(set_to_none=True); loss, *_ = forward_step(model, criterion, wd, b); loss.backward()
for _ in range(2): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step(); torch.cuda.synchronize()
WANT = ("aten::copy_", "aten::add", "aten::add_", "aten::cat", "aten::index", "aten::zeros", "aten::zero_", "aten::fill_", "aten::contiguous", "aten::clone",
        "aten::index_put_", "aten::_index_put_impl_", "aten::sum", "aten::mul", "aten::to", "aten::_to_copy", "aten::select_backward", "aten::slice_backward")
agg = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.events():
    if ev.name not in WANT or ev.device_time_total <= 0:
        continue
    frame = next((f for f in ev.stack if "tubedetr_amd" in f or "bench.py" in f), ev.stack[0] if ev.stack else "?")
    a = agg[(ev.name, frame.strip()[-90:])]
    a[0] += 1; a[1] += ev.device_time_total
print(f"{'op':24s} {'n':>4s} {'dev us':>9s}  innermost project frame")
for (name, frame), (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print(f"{name:24s} {n:4d} {us:9.1f}  {frame}")
