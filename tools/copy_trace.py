"""Which Python lines launch the step's non-library kernels (torch adds, copies, fills)?  One eager 16-clip step under
torch.profiler with stacks; groups aten ops by (op, first frame inside this repository).  usage: python tools/copy_trace.py [clips]"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("TD_ALLOW_RANDOM_TEXT_ENCODER", "1")
import torch
import tubedetr_amd
from tubedetr_amd import functional as Fk
from tubedetr_amd.models import build_model
from tubedetr_amd.harness import forward_step
from bench import make_batch, BatchTokenizer, WORKLOADS
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
T, res, k, L = WORKLOADS["cfg3"]
dev = torch.device("cuda:0")
model, criterion, wd = build_model(tubedetr_amd.default_args(stride=k, compute_dtype=torch.bfloat16))
model.to(dev).train()
tok = BatchTokenizer(); model.transformer.tokenizer = tok
b = make_batch(T, res, k, L, B, dev); tok.batch = b
params = [p for p in model.parameters() if p.requires_grad]
def step():
    Fk.invalidate_prepared()
    for p in params: p.grad = None
    loss, *_ = forward_step(model, criterion, wd, b)
    loss.backward()
for _ in range(2): step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
acc = collections.defaultdict(lambda: [0, 0.0])
WANT = ("aten::copy_", "aten::add", "aten::add_", "aten::fill_", "aten::zero_", "aten::cat", "aten::index", "aten::index_put_", "aten::clone", "aten::mul", "aten::sum", "aten::_to_copy", "aten::masked_fill_", "aten::where", "aten::embedding", "aten::embedding_dense_backward", "aten::cumsum", "aten::ne", "aten::arange", "aten::bitwise_not", "aten::div", "aten::sigmoid", "aten::sigmoid_backward")
for e in prof.events():
    if e.name not in WANT or e.device_time_total <= 0: continue
    where = "?"
    for fr in (e.stack or []):
        if "/tubedetr_amd/" in fr or "bench.py" in fr or "/autograd/" in fr:
            where = fr.strip().split("/")[-1][:90]
            if "/tubedetr_amd/" in fr or "bench.py" in fr: break
    a = acc[(e.name, where)]
    a[0] += 1; a[1] += e.device_time_total
tot = sum(v[1] for v in acc.values())
print(f"device time of the listed aten ops: {tot/1e3:.2f} ms per step")
for (n, w), (c, d) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:45]:
    print(f"{d/1e3:7.3f} ms {c:5d} x  {n:28s} {w}")
