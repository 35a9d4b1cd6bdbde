"""Host time of each custom autograd Function (forward and backward) per training step."""
import os, sys, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import os
os.environ.setdefault("TD_ALLOW_RANDOM_TEXT_ENCODER", "1")
import tubedetr_amd
from tubedetr_amd import functional as Fk
from tubedetr_amd.models import backbone as bb
from tubedetr_amd.models import build_model
from tubedetr_amd.harness import forward_step
from bench import make_batch, BatchTokenizer, WORKLOADS
acc = collections.defaultdict(lambda: [0, 0.0])
def wrap(cls, name):
    f = getattr(cls, name)
    def g(*a, **k):
        t = time.perf_counter(); r = f(*a, **k); d = time.perf_counter() - t
        e = acc[f"{cls.__name__}.{name}"]; e[0] += 1; e[1] += d; return r
    setattr(cls, name, staticmethod(g))
for cls in (Fk.LinearFn, Fk.FFNFn, Fk.AddLayerNormFn, Fk.AddFn, Fk.MHAFn, Fk.DropoutFn, Fk.CastFn, bb.ResNetTrunkFn):
    wrap(cls, "forward"); wrap(cls, "backward")
T, res, k, L = WORKLOADS["cfg3"]
dev = torch.device("cuda:0")
model, criterion, wd = build_model(tubedetr_amd.default_args(stride=k, compute_dtype=torch.bfloat16))
model.to(dev).train(); model.slow_frames_are_strided_fast = True
tok = BatchTokenizer(); model.transformer.tokenizer = tok
b = make_batch(T, res, k, L, 1, dev); tok.batch = b
params = [p for p in model.parameters() if p.requires_grad]
tt = collections.defaultdict(float)
def step(rec):
    Fk.invalidate_prepared()
    for p in params: p.grad = None
    t0 = time.perf_counter(); loss, *_ = forward_step(model, criterion, wd, b); t1 = time.perf_counter(); loss.backward(); t2 = time.perf_counter()
    if rec: tt["fwd"] += t1 - t0; tt["bwd"] += t2 - t1
for _ in range(3): step(False)
torch.cuda.synchronize(); acc.clear()
N = 5
for _ in range(N): step(True)
torch.cuda.synchronize()
print(f"host fwd {tt['fwd']/N*1e3:.2f} ms  bwd {tt['bwd']/N*1e3:.2f} ms per step")
for k_, (c, d) in sorted(acc.items(), key=lambda kv: -kv[1][1]): print(f"{k_:28s} {c/N:6.1f} calls/step {d/N*1e3:7.2f} ms/step  {d/c*1e6:7.1f} us/call")
