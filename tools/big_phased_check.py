"""Race screen + A/B identity of the phased 256 x 256 main loop (conv_gemm_big8_kernel) against the lock-step one
(conv_gemm_big_kernel): both accumulate every output element in the same order (K tiles ascending, k-step 0 then 1), so the
results must be BIT-identical, run after run.  The knob is read once per process: the script re-runs itself per arm.
The same harness screens the re-pipelined persistent pointwise kernel (pw_resident2_kernel vs pw_resident_kernel, knob
TD_PW_PERSIST_V2): same arithmetic, bit-identical results required.
usage: big_phased_check.py [knob]     (driver: runs both arms of TD_CONV_BIG_PHASED (default) or of the named knob, compares)
       big_phased_check.py arm <out>  (one arm, knobs taken from the environment)"""
import os, subprocess, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

CASES = [  # (kind, frames, H, C, Nout)
    ("fwd3x3", 125, 22, 256, 256), ("fwd3x3", 1000, 22, 256, 256), ("dgrad3x3", 200, 22, 256, 256), ("fwd3x3", 500, 11, 512, 512),
    ("pw", 125, 22, 1024, 256), ("pw", 500, 11, 2048, 512), ("pw_res", 301, 22, 512, 256), ("fwd3x3", 37, 22, 256, 256),
    # 128 output channels: the three-stage phased instance (layer2's 3x3 forward / input gradient, its conv1, an odd K-tile count)
    ("fwd3x3", 200, 44, 128, 128), ("dgrad3x3", 100, 44, 128, 128), ("pw", 200, 44, 512, 128), ("pw_res", 150, 44, 832, 128), ("fwd3x3", 23, 44, 128, 128),
    # persistent pointwise instance (K <= 256): conv3 + residual of layer3 / layer1 / layer2, a conv1 input gradient (residual + mask), no operands, dropout
    ("pw_res", 800, 22, 256, 1024), ("pw_res", 100, 88, 64, 256), ("pw_res", 77, 44, 128, 512), ("pw_res_mask", 200, 22, 256, 1024),
    ("pw_res_mask", 50, 44, 128, 512), ("pw", 333, 22, 192, 256), ("pw_mask", 200, 22, 256, 1024), ("pw_drop", 100, 22, 256, 2048)]


def arm(out):
    from tubedetr_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(7)
    res = {}
    for ci, (kind, N, H, C, Nn) in enumerate(CASES):
        if kind in ("fwd3x3", "dgrad3x3"):
            x = torch.randn(N, H, H, C, device=dev, generator=g).relu().bfloat16()
            w = (torch.randn(Nn, 9 * C, device=dev, generator=g) * 0.02).bfloat16()
            b = torch.randn(Nn, device=dev, generator=g)
            if kind == "fwd3x3":
                run = lambda: ops.conv_fwd(x, w, b, 3, 3, 1, 1, relu=True)
            else:
                act = torch.randn(N, H, H, C, device=dev, generator=g).relu().bfloat16()
                run = lambda: ops.conv_dgrad(x, w, (H, H), 3, 3, 1, 1, mask_src=act)
        else:
            x = torch.randn(N * H * H, C, device=dev, generator=g).bfloat16()
            w = (torch.randn(Nn, C, device=dev, generator=g) * 0.02).bfloat16()
            b = torch.randn(Nn, device=dev, generator=g)
            r = torch.randn(N * H * H, Nn, device=dev, generator=g).bfloat16() if "res" in kind else None
            mk = torch.randn(N * H * H, Nn, device=dev, generator=g).relu().bfloat16() if "mask" in kind else None
            if kind == "pw_drop":
                run = lambda: ops.linear_fwd(x, w, b, relu=True, dropout_p=0.1, seed=1234)
            else:
                run = lambda: ops.linear_fwd(x, w, b, residual=r, relu=mk is None, mask_src=mk)
        first = run().clone()
        bad = 0
        for _ in range(12):  # race screen: every repeat bit-identical to the first
            bad += int(not torch.equal(run(), first))
        res[ci] = first.cpu()
        print(f"case {ci} {kind} N={N} H={H} C={C} N={Nn}: repeats differing from the first: {bad}", flush=True)
        assert bad == 0
    torch.save(res, out)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "arm":
        arm(sys.argv[2])
        sys.exit(0)
    knob = sys.argv[1] if len(sys.argv) > 1 else "TD_CONV_BIG_PHASED"
    outs = []
    for v in ("0", "1"):
        o = f"/tmp/big_phased_{v}.pt"
        env = dict(os.environ, **{knob: v})
        subprocess.run([sys.executable, os.path.abspath(__file__), "arm", o], check=True, env=env)
        outs.append(torch.load(o))
    ok = True
    for ci in outs[0]:
        a, b = outs[0][ci], outs[1][ci]
        same = torch.equal(a, b)
        md = (a.float() - b.float()).abs().max().item()
        print(f"case {ci} {CASES[ci]}: {knob}=1 == {knob}=0 bitwise: {same} (max abs diff {md:.3e})")
        ok &= same
    print("IDENTICAL" if ok else "MISMATCH")
    sys.exit(0 if ok else 1)
