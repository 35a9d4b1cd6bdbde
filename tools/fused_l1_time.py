"""Times the fused stem / fused layer1 bottleneck kernels at the benchmark's shape (1 000 frames of res 352) next to the
launch sequences they replace.  python tools/fused_l1_time.py [frames]"""
import math
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tubedetr_amd import _hip, ops  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
dev = torch.device("cuda:0")
dt = torch.bfloat16
g = torch.Generator(device=dev).manual_seed(0)


def timeit(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return ts[len(ts) // 2]


def prep(w, b):
    wf, _, bo, _ = ops.weight_prep(w, dt, bias=b, need_dgrad=False)
    return wf, bo


H = W = 88
for Cin in (256, 64):
    x = (torch.randn((N, H, W, Cin), generator=g, device=dev)).relu().to(dt)
    w1, b1 = prep(torch.randn(64, Cin, 1, 1, generator=g, device=dev) / math.sqrt(Cin), torch.randn(64, generator=g, device=dev))
    w2, b2 = prep(torch.randn(64, 64, 3, 3, generator=g, device=dev) / 24, torch.randn(64, generator=g, device=dev))
    w3, b3 = prep(torch.randn(256, 64, 1, 1, generator=g, device=dev) / 8, torch.randn(256, generator=g, device=dev))
    wd, bd = (prep(torch.randn(256, 64, 1, 1, generator=g, device=dev) / 8, torch.randn(256, generator=g, device=dev)) if Cin == 64 else (None, None))
    out = torch.empty((N, H, W, 256), dtype=dt, device=dev)

    def fused():
        _hip.check(_hip.lib().td_bottleneck_fused(x.data_ptr(), out.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), w3.data_ptr(),
                                                  b3.data_ptr(), _hip.ptr(wd), _hip.ptr(bd), N, H, W, Cin, _hip.TD_BF16, _hip.stream_ptr()), "td_bottleneck_fused")

    def separate():
        h1 = ops.conv_fwd(x, w1, b1, 1, 1, 1, 0, relu=True)
        h2 = ops.conv_fwd(h1, w2, b2, 3, 3, 1, 1, relu=True)
        idn = x if wd is None else ops.conv_fwd(x, wd, bd, 1, 1, 1, 0)
        return ops.conv_fwd(h2, w3, b3, 1, 1, 1, 0, residual=idn, relu=True)

    tf, ts = timeit(fused), timeit(separate)
    gb = N * H * W * (Cin + 256) * 2 / 1e9
    print(f"bottleneck Cin={Cin}: fused {tf:.3f} ms ({gb / tf:.2f} TB/s of in+out bytes), layer by layer {ts:.3f} ms", flush=True)
    del x, out

# stem
Hs = Ws = 352
x4 = torch.randn((N, Hs, Ws, 4), generator=g, device=dev).to(dt)
x4[..., 3] = 0
w = torch.randn(64, 3, 7, 7, generator=g, device=dev) * 0.1
wf8, _, b_out, _ = ops.weight_prep(w, dt, bias=torch.randn(64, generator=g, device=dev), need_dgrad=False, cpad=8)
wp = torch.empty((64, 224), dtype=dt, device=dev)
_hip.check(_hip.lib().td_stem_pair_weights(wf8.data_ptr(), wp.data_ptr(), 64, _hip.TD_BF16, _hip.stream_ptr()), "pairs")
y = torch.empty((N, 88, 88, 64), dtype=dt, device=dev)
c = torch.empty((N, 176, 176, 64), dtype=dt, device=dev)


def stem_fused():
    _hip.check(_hip.lib().td_stem_pool(x4.data_ptr(), wp.data_ptr(), b_out.data_ptr(), y.data_ptr(), N, Hs, Ws, _hip.TD_BF16, _hip.stream_ptr()), "td_stem_pool")


def stem_sep():
    ops.conv_gemm_raw(x4.view(N, Hs, Ws // 2, 8), wp, c, ops._desc(N, Hs, Ws // 2, 8, 176, 176, 7, 4, 2, 3, 0, 64, 64, stride_w=1, pad_w=2), ops._epi(b_out, None, None, True))
    return ops.maxpool3x3s2(c)


print(f"stem: fused {timeit(stem_fused):.3f} ms, conv + pool {timeit(stem_sep):.3f} ms", flush=True)
