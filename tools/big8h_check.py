"""conv_gemm_big8h_kernel (3x3 / stride 1: a channel chunk's rows staged once for all nine taps) against conv_gemm_big8_kernel (one
activation piece set per tap): the results must be BIT-IDENTICAL (same K order, same zero padding), forward and input gradient, with
bias / ReLU / mask / residual epilogues, ragged last tiles, two column tiles (512 output channels), images whose rows do not divide the
tile.  Also prints the time of both.  usage: big8h_check.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tubedetr_amd import ops

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)


def timeit(fn, reps=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(True), torch.cuda.Event(True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


bad = 0
for frames, H, W, C in [(800, 22, 22, 256), (333, 22, 22, 256), (1601, 11, 11, 512), (400, 22, 22, 256), (640, 14, 14, 256), (350, 20, 24, 256)]:
    x = torch.randn(frames, H, W, C, device=dev, generator=g).bfloat16()
    wf = (torch.randn(C, 9 * C, device=dev, generator=g) * 0.02).bfloat16()
    bias = torch.randn(C, device=dev, generator=g)
    res = torch.randn(frames, H, W, C, device=dev, generator=g).bfloat16()
    act = torch.randn(frames, H, W, C, device=dev, generator=g).relu().bfloat16()
    cases = {
        "fwd bias+relu": lambda o: ops.conv_fwd(x, wf, bias, 3, 3, 1, 1, relu=True, out=o),
        "fwd bias+res+relu": lambda o: ops.conv_fwd(x, wf, bias, 3, 3, 1, 1, residual=res, relu=True, out=o),
        "dgrad mask": lambda o: ops.conv_dgrad(x, wf, (H, W), 3, 3, 1, 1, mask_src=act, out=o),
        "dgrad res+mask": lambda o: ops.conv_dgrad(x, wf, (H, W), 3, 3, 1, 1, residual=res, mask_src=act, out=o),
    }
    for name, fn in cases.items():
        outs, us = {}, {}
        for halo in ("1", "0"):
            os.environ["TD_CONV_BIG_HALO"] = halo
            o = torch.empty(frames, H, W, C, device=dev, dtype=torch.bfloat16)
            o.fill_(float("nan"))
            fn(o)
            torch.cuda.synchronize()
            outs[halo] = o
            us[halo] = timeit(lambda: fn(o))
        same = torch.equal(outs["1"].view(torch.int16), outs["0"].view(torch.int16))
        nan = torch.isnan(outs["1"].float()).any().item()
        d = (outs["1"].float() - outs["0"].float()).abs().max().item()
        bad += (not same) or nan
        print(f"N={frames} {H}x{W} C={C} {name:18s}: bit-identical={same} nan={nan} max|diff|={d:.3e}  halo {us['1']:7.1f} us  per-tap {us['0']:7.1f} us", flush=True)
os.environ.pop("TD_CONV_BIG_HALO", None)
print("FAILED" if bad else "ALL BIT-IDENTICAL")
