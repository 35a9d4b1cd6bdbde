"""Per-launch table of the MFMA kernels in one training step (in-library HIP-event timing): shape, ms, TFLOP/s, GB/s.
usage: conv_table.py [clips per step, default 4]   (family ids: include/tubedetr_hip.h TD_PROF_*)"""
import os, sys, csv, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import os
os.environ.setdefault("TD_ALLOW_RANDOM_TEXT_ENCODER", "1")
import tubedetr_amd
from tubedetr_amd import _hip
from tubedetr_amd.models import build_model
from tubedetr_amd.harness import forward_step
from bench import make_batch, BatchTokenizer, WORKLOADS
T, res, k, L = WORKLOADS["cfg3"]
dev = torch.device("cuda:0")
model, criterion, wd = build_model(tubedetr_amd.default_args(stride=k, compute_dtype=torch.bfloat16))
model.to(dev).train()
tok = BatchTokenizer(); model.transformer.tokenizer = tok
clips = int(sys.argv[1]) if len(sys.argv) > 1 else 4
b = make_batch(T, res, k, L, 1, dev, clips=clips); tok.batch = b
for _ in range(2):
    model.zero_grad(set_to_none=True); loss, *_ = forward_step(model, criterion, wd, b); loss.backward()
torch.cuda.synchronize()
Lb = _hip.lib(); Lb.td_prof_enable(1)
model.zero_grad(set_to_none=True); loss, *_ = forward_step(model, criterion, wd, b); loss.backward()
torch.cuda.synchronize()
Lb.td_prof_dump(b"/tmp/prof_dump.csv"); Lb.td_prof_enable(0)
rows = list(csv.DictReader(open("/tmp/prof_dump.csv")))
agg = collections.OrderedDict()
for r in rows:
    key = (r["family"], r["M"], r["N"], r["K"], r["R"], r["stride"], r["mode"])
    a = agg.setdefault(key, [0, 0.0]); a[0] += 1; a[1] += float(r["ms"])
tot = sum(v[1] for v in agg.values())
print(f"total MFMA-kernel ms/step {tot:.2f}")
print("fam      M     N     K  R st md/sp  cnt   ms_tot  us_each   TF/s   GB/s(min)")
for key, (cnt, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(os.environ.get('ROWS', '70'))]:
    fam, M, N, K, R, st, md = (int(x) for x in key)
    fl = 2.0 * M * N * K * cnt
    if fam == 2: byts = (M * N + M * K / (R * R if R > 1 else 1)) * 2 * cnt + N * K * 4 * cnt
    else: byts = (M * K / (R * R if R > 1 else 1) + N * K + M * N) * 2 * cnt
    print(f"{fam:3d} {M:7d} {N:5d} {K:5d} {R:2d} {st:2d} {md:4d} {cnt:5d} {ms:8.3f} {ms/cnt*1e3:8.1f} {fl/ms/1e9:7.1f} {byts/ms/1e6:8.0f}")
