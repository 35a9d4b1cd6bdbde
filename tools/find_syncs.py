import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, traceback
import os
os.environ.setdefault("TD_ALLOW_RANDOM_TEXT_ENCODER", "1")
import tubedetr_amd
from tubedetr_amd.models import build_model
from tubedetr_amd.harness import forward_step
from bench import make_batch, BatchTokenizer
dev = torch.device("cuda:0")
model, criterion, wd = build_model(tubedetr_amd.default_args(stride=4, compute_dtype=torch.bfloat16))
model.to(dev).train()
tok = BatchTokenizer(); model.transformer.tokenizer = tok
b = make_batch(8, 96, 4, 6, 1, dev); tok.batch = b
for _ in range(2):
    loss, *_ = forward_step(model, criterion, wd, b); loss.backward()
torch.cuda.synchronize()
torch.cuda.set_sync_debug_mode("warn")
warnings.simplefilter("always")
import warnings as w
def showwarning(message, category, filename, lineno, file=None, line=None):
    st = [f for f in traceback.extract_stack() if "site-packages/torch" not in f.filename][-6:-1]
    print("SYNC:", str(message)[:80], " <- ", " | ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in st))
w.showwarning = showwarning
loss, *_ = forward_step(model, criterion, wd, b)
print("---- backward")
loss.backward()
torch.cuda.set_sync_debug_mode("default")
