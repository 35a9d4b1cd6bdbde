"""Find reads of uninitialised device memory: run one small training step under the electric-fence allocator with
TD_EFENCE_POISON=1 (fresh memory = 0xFF bytes = NaN) and report (a) the first tensor-level op (tubedetr_amd.ops.*) whose
output is non-finite although all of its tensor inputs were finite, (b) every parameter whose gradient is non-finite.
Usage: python tests/efence/trace_nan.py [bf16|fp32] [train|eval] [T res k L] [fast0|fast1]"""
import os
import sys

os.environ.setdefault("TD_EFENCE_POISON", "1")
os.environ.setdefault("TD_ALLOW_RANDOM_TEXT_ENCODER", "1")
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import install  # noqa: E402

if os.environ.get("TD_EFENCE_OFF") != "1":
    install.install()
import torch  # noqa: E402


def tensors_of(obj, out):
    if torch.is_tensor(obj):
        out.append(obj)
    elif isinstance(obj, (list, tuple)):
        for o in obj:
            tensors_of(o, out)
    elif isinstance(obj, dict):
        for o in obj.values():
            tensors_of(o, out)


def finite(t):
    return (not t.is_floating_point()) or bool(torch.isfinite(t).all())


def main():
    a = sys.argv[1:]
    dt = torch.bfloat16 if (a[0:1] or ["bf16"])[0] == "bf16" else torch.float32
    train = (a[1:2] or ["train"])[0] == "train"
    T, res, k, L = (int(x) for x in (a[2:6] or [8, 96, 4, 6]))
    fast = (a[6:7] or ["fast1"])[0] == "fast1"
    import tubedetr_amd
    from oracle.tubedetr_oracle import OracleConfig
    from oracle.weights import fill_state, state_spec, synthetic_batch
    from tubedetr_amd import ops
    from tubedetr_amd.harness import FixedTokenizer, batch_to, forward_step
    from tubedetr_amd.models import build_model

    first = []

    OUT_ARGS = {"conv_gemm_raw": (2,), "mha_bwd": (8, 9, 10), "conv1x1s_dgrad_scatter": (), "linear_wgrad_batch": ()}

    def wrap(name, fn):
        def w(*args, **kw):
            ins = []
            skip = OUT_ARGS.get(name, ())
            tensors_of(([x for i_, x in enumerate(args) if i_ not in skip], {k_: v for k_, v in kw.items() if k_ not in ("out", "dbias")}), ins)
            torch.cuda.synchronize()
            ok_in = all(finite(t) for t in ins if t.is_cuda and t.numel())
            r = fn(*args, **kw)
            torch.cuda.synchronize()
            outs = []
            tensors_of(r, outs)
            tensors_of([v for kk, v in kw.items() if kk in ("out", "dbias")], outs)
            bad = [tuple(t.shape) for t in outs if t.is_cuda and not finite(t)]
            if bad and len(first) < 8:
                first.append(name)
                tag = "NAN-ORIGIN" if ok_in else "nan-propagated"
                print(f"[{tag}] {name}: outputs non-finite {bad}; input shapes {[tuple(t.shape) for t in ins]} finite {[finite(t) for t in ins]}", flush=True)
            return r
        return w

    for name in dir(ops):
        fn = getattr(ops, name)
        if callable(fn) and not name.startswith("_") and getattr(fn, "__module__", "") == ops.__name__ and name not in ("set_dropout_counter", "zeros_f32", "vec_of", "pad_to", "conv_out"):
            setattr(ops, name, wrap(name, fn))

    cfg = OracleConfig(stride=k, fast=fast)
    batch = synthetic_batch(T=T, res=res, k=k, L=L, seed=31)
    sd = fill_state(state_spec(cfg), 9)
    dev = torch.device("cuda:0")
    model, criterion, weight_dict = build_model(tubedetr_amd.default_args(stride=k, fast=fast, compute_dtype=dt))
    model.load_state_dict(sd, strict=True)
    model.to(dev).train(train)
    model.transformer.tokenizer = FixedTokenizer(batch["input_ids"], batch["attention_mask"])
    for it in range(2):
        for p in model.parameters():
            p.grad = None
        loss, ld, out, cache = forward_step(model, criterion, weight_dict, batch_to(batch, dev))
        torch.cuda.synchronize()
        print(f"iter {it}: loss {loss.item()}  finite cache: { {k_: finite(v) for k_, v in cache.items() if torch.is_tensor(v)} }", flush=True)
        loss.backward()
        torch.cuda.synchronize()
        bad = [n for n, p in model.named_parameters() if p.grad is not None and not bool(torch.isfinite(p.grad).all())]
        print(f"iter {it}: {len(bad)} parameters with non-finite gradients: {bad[:40]}", flush=True)
    print("protected:", install.protected() if os.environ.get("TD_EFENCE_OFF") != "1" else None)


if __name__ == "__main__":
    main()
