"""Which tensors do the step's library (aten) copies / adds / fills touch?  One eager training step under torch.profiler with shapes;
prints, per (op, input shapes), launch count and device time.  usage: op_shapes.py [clips]"""
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
clips = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = torch.device("cuda:0")
model, criterion, wd = build_model(tubedetr_amd.default_args(stride=k, compute_dtype=torch.bfloat16))
model.to(dev).train()
tok = BatchTokenizer(); model.transformer.tokenizer = tok
b = make_batch(T, res, k, L, 1, dev, clips=clips); tok.batch = b
def step():
    model.zero_grad(set_to_none=True); loss, *_ = forward_step(model, criterion, wd, b); loss.backward()
for _ in range(2): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step(); torch.cuda.synchronize()
WANT = ("aten::copy_", "aten::add", "aten::add_", "aten::cat", "aten::index", "aten::zero_", "aten::fill_", "aten::clone", "aten::_to_copy",
        "aten::index_put_", "aten::_index_put_impl_", "aten::sum", "aten::mul", "aten::select_backward", "aten::slice_backward", "aten::embedding_dense_backward")
agg = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.events():
    if ev.name not in WANT or ev.device_time_total <= 0:
        continue
    a = agg[(ev.name, str(ev.input_shapes)[:110])]
    a[0] += 1; a[1] += ev.device_time_total
print(f"{'op':28s} {'n':>4s} {'dev us':>9s}  input shapes")
for (name, shp), (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:60]:
    print(f"{name:28s} {n:4d} {us:9.1f}  {shp}")
