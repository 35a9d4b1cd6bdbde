"""The N>1 code path with the REAL model on a real MI355X.

(a) 1-rank RCCL process group, the exact sequence bench.py runs at N>1: the step captured in a HIP graph ends with the
    gather of all gradients into the flat exchange buffer, the all-reduce runs outside the graph, `.grad` is the
    zero-copy view - the result must equal the gradients of the plain eager step.
(b) world size 2 on ONE GPU (two processes, `gloo` moving the device buffer through the host - RCCL refuses two ranks on
    one device): each rank back-propagates its own clip through the real model, the flat reducer averages; rank 0 checks
    every parameter against the average of the two single-process gradients it computes itself.
"""
import os
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _small_model(dtype, seed=0):
    import tubedetr_amd
    from tubedetr_amd.models import build_model

    torch.manual_seed(seed)
    model, criterion, weight_dict = build_model(tubedetr_amd.default_args(stride=2, compute_dtype=dtype))
    # reference init leaves fast_residual at zero (inert branch): randomise it so every parameter gets a real gradient
    with torch.no_grad():
        for p in model.transformer.fast_residual.parameters():
            p.normal_(0, 0.05)
    return model.to(torch.device("cuda:0")).eval(), criterion, weight_dict


def _clip(seed):
    from oracle.weights import synthetic_batch
    from tubedetr_amd.harness import batch_to

    return batch_to(synthetic_batch(T=6, res=64, k=2, L=5, seed=seed), torch.device("cuda:0"))


def _step(model, criterion, weight_dict, batch):
    from tubedetr_amd.functional import invalidate_prepared
    from tubedetr_amd.harness import forward_step

    invalidate_prepared()
    loss, _, _, _ = forward_step(model, criterion, weight_dict, batch)
    loss.backward()
    return loss


def test_graph_captured_gather_allreduce_attach_equals_plain_step():
    import torch.distributed as dist

    from tubedetr_amd.distributed import FlatGradAllReducer, sync_num_boxes
    from tubedetr_amd.harness import FixedTokenizer

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1")
    dev = torch.device("cuda:0")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    os.environ["TD_TEXT_STREAM"] = "0"
    try:
        model, criterion, weight_dict = _small_model(torch.bfloat16)
        batch = _clip(4)
        model.transformer.tokenizer = FixedTokenizer(batch["input_ids"], batch["attention_mask"])
        params = [p for p in model.parameters() if p.requires_grad]
        names = [n for n, p in model.named_parameters() if p.requires_grad]
        criterion.external_num_boxes = torch.ones(1, dtype=torch.float32, device=dev)
        sync_num_boxes(batch["target_boxes"].shape[0], criterion.external_num_boxes)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                for p in params:
                    p.grad = None
                _step(model, criterion, weight_dict, batch)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        plain = [None if p.grad is None else p.grad.detach().float().clone() for p in params]
        for p in params:
            p.grad = None
        reducer = FlatGradAllReducer(params)
        reducer.always_communicate = True  # issue the RCCL call although the group has one rank
        graph = torch.cuda.CUDAGraph()
        # thread_local: the RCCL watchdog thread may query its events while this thread captures (global mode would fail the capture or the watchdog)
        with torch.cuda.graph(graph, capture_error_mode="thread_local"):
            _step(model, criterion, weight_dict, batch)
            reducer.gather()
        reducer.attach()
        for _ in range(3):
            graph.replay()
            reducer.all_reduce()
        torch.cuda.synchronize()
        checked = 0
        for i, (n, p, ref) in enumerate(zip(names, params, plain)):
            if ref is None:
                assert p.grad is None, n  # RoBERTa's pooler: unused everywhere -> stays None (find_unused_parameters semantics)
                continue
            assert p.grad is not None and p.grad.data_ptr() == reducer.views[i].data_ptr(), n
            scale = ref.abs().max().clamp_min(1e-4)
            err = ((p.grad.float() - ref).abs().max() / scale).item()
            assert err < 2e-2, (n, err)
            checked += 1
        assert checked > 300
    except BaseException:
        import traceback

        traceback.print_exc()  # (a device error makes destroy_process_group abort the process: show the cause first)
        raise
    finally:
        os.environ.pop("TD_TEXT_STREAM", None)
        try:
            torch.cuda.synchronize()
        except Exception:
            pass
        dist.destroy_process_group()


def _rank_main(rank, world, port, out_path):
    import torch.distributed as dist

    from tubedetr_amd.distributed import FlatGradAllReducer, sync_num_boxes
    from tubedetr_amd.harness import FixedTokenizer

    os.environ.setdefault("TD_ALLOW_RANDOM_TEXT_ENCODER", "1")
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        dev = torch.device("cuda:0")
        model, criterion, weight_dict = _small_model(torch.float32, seed=0)  # same seed: replicas start identical (DDP broadcasts rank 0's)
        params = [p for p in model.parameters() if p.requires_grad]
        criterion.external_num_boxes = torch.ones(1, dtype=torch.float32, device=dev)
        clips = [_clip(1000 * r + 3) for r in range(world)]  # bench.py's per-rank seeds
        sync_num_boxes(clips[rank]["target_boxes"].shape[0], criterion.external_num_boxes)
        model.transformer.tokenizer = FixedTokenizer(clips[rank]["input_ids"], clips[rank]["attention_mask"])
        _step(model, criterion, weight_dict, clips[rank])
        reducer = FlatGradAllReducer(params)
        reducer.reduce(attach=True)
        torch.cuda.synchronize()
        got = [None if p.grad is None else p.grad.detach().float().clone() for p in params]
        if rank == 0:
            want = None
            for r in range(world):
                for p in params:
                    p.grad = None
                model.transformer.tokenizer = FixedTokenizer(clips[r]["input_ids"], clips[r]["attention_mask"])
                _step(model, criterion, weight_dict, clips[r])
                g = [None if p.grad is None else p.grad.detach().float().clone() for p in params]
                want = g if want is None else [a if b is None else a + b for a, b in zip(want, g)]
            worst, n = 0.0, 0
            for a, b in zip(got, want):
                assert (a is None) == (b is None)
                if a is None:
                    continue
                ref = b / world
                worst = max(worst, ((a - ref).abs().max() / (1e-4 * ref.abs().max() + 1e-6)).item())  # fp32 atomics re-order sums
                n += 1
            with open(out_path, "w") as f:
                f.write(f"{worst} {n}")
    finally:
        dist.destroy_process_group()


def test_world_size_2_real_model_one_gpu(tmp_path):
    import torch.multiprocessing as mp

    out = str(tmp_path / "ws2.txt")
    mp.spawn(_rank_main, args=(2, _free_port(), out), nprocs=2, join=True)
    worst, n = open(out).read().split()
    assert int(n) > 300 and float(worst) < 1.0, (worst, n)  # every |got - mean| <= 1e-4 * max|mean| + 1e-6


@pytest.mark.parametrize("text_stream", ["0", "1"])  # "1": RoBERTa on a forked branch inside the first graph (bench.py's N > 1 default)
def test_split_backward_two_graphs_with_overlapped_exchange_equals_plain_step(text_stream):
    """The N>1 execution bench.py uses: the step is cut at the trunk boundary and captured as TWO HIP graphs (forward +
    first backward stage + gather of the early gradients | trunk backward + gather of the trunk's gradients); the
    all-reduce of the early 0.57 GB is launched between the two replays and overlaps the trunk backward.  With a 1-rank
    RCCL group the result must equal the gradients of a plain single-backward step."""
    import torch.distributed as dist

    from tubedetr_amd.distributed import FlatGradAllReducer, sync_num_boxes
    from tubedetr_amd.harness import FixedTokenizer, backward_in_stages, forward_step, set_split_backward
    from tubedetr_amd.functional import invalidate_prepared

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1")
    dev = torch.device("cuda:0")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    os.environ["TD_TEXT_STREAM"] = text_stream
    try:
        model, criterion, weight_dict = _small_model(torch.bfloat16)
        batch = _clip(4)
        model.transformer.tokenizer = FixedTokenizer(batch["input_ids"], batch["attention_mask"])
        params = [p for p in model.parameters() if p.requires_grad]
        names = [n for n, p in model.named_parameters() if p.requires_grad]
        criterion.external_num_boxes = torch.ones(1, dtype=torch.float32, device=dev)
        sync_num_boxes(batch["target_boxes"].shape[0], criterion.external_num_boxes)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                for p in params:
                    p.grad = None
                _step(model, criterion, weight_dict, batch)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        plain = [None if p.grad is None else p.grad.detach().float().clone() for p in params]
        # staged, eager
        set_split_backward(model, True)
        late = [p for n, p in model.named_parameters() if n.startswith("backbone.") and p.requires_grad]
        reducer = FlatGradAllReducer(params, late=late)
        reducer.always_communicate = True
        assert [r[2] for r in reducer.runs] == [False, True, False] and sum(r[1] - r[0] for r in reducer.runs if r[2]) == sum(p.numel() for p in late)
        for p in params:
            p.grad = None
        invalidate_prepared()
        loss, _, _, _ = forward_step(model, criterion, weight_dict, batch)
        backward_in_stages(model, loss, after_first_stage=lambda: reducer.launch(early=True))
        reducer.launch(early=False)
        reducer.finish(attach=True)
        torch.cuda.synchronize()

        def check(tag):
            n_ok = 0
            for n, p, ref in zip(names, params, plain):
                if ref is None:
                    assert p.grad is None, (tag, n)
                    continue
                scale = ref.abs().max().clamp_min(1e-4)
                err = ((p.grad.float() - ref).abs().max() / scale).item()
                assert err < 2e-2, (tag, n, err)
                n_ok += 1
            assert n_ok > 300

        check("eager")
        # staged, two graphs sharing one memory pool
        for p in params:
            p.grad = None
        reducer2 = FlatGradAllReducer(params, late=late)
        reducer2.always_communicate = True
        with torch.cuda.stream(side):  # job tables / caches of the split path exist before capture
            invalidate_prepared()
            l_, _, _, _ = forward_step(model, criterion, weight_dict, batch)
            backward_in_stages(model, l_)
            for p in params:
                p.grad = None
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g1, g2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        from tubedetr_amd import ops as _ops

        _ops.reset_capture_arena()
        with torch.cuda.graph(g1, capture_error_mode="thread_local"):
            invalidate_prepared()
            l_, _, _, _ = forward_step(model, criterion, weight_dict, batch)
            l_.backward()
            reducer2.gather_stage(early=True)
        _ops.reset_capture_arena()  # (two captures back to back: the second graph gets zero-fill nodes of its own)
        with torch.cuda.graph(g2, pool=g1.pool(), capture_error_mode="thread_local"):
            model.backbone[0].body.backward_trunk()
            reducer2.gather_stage(early=False)
        for _ in range(3):
            g1.replay()
            reducer2.exchange_stage(early=True)
            g2.replay()
            reducer2.exchange_stage(early=False)
            reducer2.finish(attach=True)
        torch.cuda.synchronize()
        check("two graphs")
        # the trunk itself stage by stage (td_resnet_bwd only_stage: layer4 | layer3 | layer2, each stage's weight gradients in a launch of their
        # own): its gradients leave in three pieces while the remaining stages are still computed - eagerly, then as 1 + 3 graphs
        from tubedetr_amd.harness import trunk_stage_groups

        groups = trunk_stage_groups(model)
        assert len(groups) == 3 and sum(len(g_) for g_ in groups) == len(late)
        reducer3 = FlatGradAllReducer(params, late_groups=groups)
        reducer3.always_communicate = True
        assert [r[2] for r in reducer3.runs if r[2]] == [3, 2, 1]  # parameter order layer2, layer3, layer4 = late stages 3, 2, 1
        for p in params:
            p.grad = None
        invalidate_prepared()
        loss, _, _, _ = forward_step(model, criterion, weight_dict, batch)
        seen = []
        backward_in_stages(model, loss, after_first_stage=lambda: reducer3.launch(early=True),
                           after_trunk_stage=lambda k_, ws_: (seen.append((k_, len(ws_))), reducer3.launch(stage=k_)))
        reducer3.finish(attach=True)
        torch.cuda.synchronize()
        assert [k_ for k_, _ in seen] == [1, 2, 3] and [n_ for _, n_ in seen] == [len(g_) for g_ in groups]
        check("trunk in three pieces, eager")
        for p in params:
            p.grad = None
        reducer4 = FlatGradAllReducer(params, late_groups=groups)
        reducer4.always_communicate = True
        body = model.backbone[0].body
        g1 = torch.cuda.CUDAGraph()
        _ops.reset_capture_arena()
        with torch.cuda.graph(g1, capture_error_mode="thread_local"):
            invalidate_prepared()
            l_, _, _, _ = forward_step(model, criterion, weight_dict, batch)
            l_.backward()
            reducer4.gather_stage(early=True)
        it = body.backward_trunk_iter()
        gs = []
        for _ in range(3):
            g_ = torch.cuda.CUDAGraph()
            _ops.reset_capture_arena()
            with torch.cuda.graph(g_, pool=g1.pool(), capture_error_mode="thread_local"):
                st_, _w = next(it)
                reducer4.gather_stage(stage=4 - st_)
            gs.append((g_, 4 - st_))
        assert next(it, None) is None
        for _ in range(3):
            g1.replay()
            reducer4.exchange_stage(early=True)
            for g_, k_ in gs:
                g_.replay()
                reducer4.exchange_stage(stage=k_)
            reducer4.finish(attach=True)
        torch.cuda.synchronize()
        check("four graphs")
    except BaseException:
        import traceback

        traceback.print_exc()  # (a device error makes destroy_process_group abort the process: show the cause first)
        raise
    finally:
        os.environ.pop("TD_TEXT_STREAM", None)
        try:
            torch.cuda.synchronize()
        except Exception:
            pass
        dist.destroy_process_group()


def test_reduce_scatter_all_gather_exchange_on_rccl():
    """collective="rs_ag" through RCCL itself (a 1-rank group: the averaged buffer equals the input): the asynchronous
    reduce-scatter / all-gather pair of the staged exchange, an odd buffer length (the remainder goes through an all-reduce),
    fp32 and bf16 wire.  Bit-equality with the single all-reduce at world size 2 is the gloo test's (tests/test_distributed_cpu.py)."""
    import torch.distributed as dist

    from tubedetr_amd.distributed import FlatGradAllReducer

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1")
    dev = torch.device("cuda:0")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        torch.manual_seed(0)
        m = torch.nn.Sequential(torch.nn.Linear(33, 65), torch.nn.Linear(65, 17), torch.nn.Linear(17, 3, bias=False)).to(dev)
        assert sum(p.numel() for p in m.parameters()) % 2 == 1
        for wire in (torch.float32, torch.bfloat16):
            g = torch.Generator(device=dev).manual_seed(5)
            for p in m.parameters():
                p.grad = torch.randn(p.shape, generator=g, device=dev)
            want = torch.cat([p.grad.flatten() for p in m.parameters()])
            if wire == torch.bfloat16:
                want = want.to(torch.bfloat16).float()
            red = FlatGradAllReducer(m.parameters(), wire, late=list(m[1].parameters()), collective="rs_ag")
            red.always_communicate = True
            red.launch(early=True)
            red.launch(early=False)
            red.finish(attach=True)
            torch.cuda.synchronize()
            assert torch.equal(red.flat, want)
            red2 = FlatGradAllReducer(m.parameters(), wire, collective="rs_ag")
            red2.always_communicate = True
            for p, v in zip(m.parameters(), torch.split(want, [p.numel() for p in m.parameters()])):
                p.grad = v.view_as(p).clone()
            red2.reduce(attach=False)
            torch.cuda.synchronize()
            assert torch.equal(red2.flat, want)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("collective", ["all_reduce", "rs_ag"])
def test_bench_main_at_world_size_2_on_one_gpu(tmp_path, collective):
    """BASELINE config 4's ENTRY POINT, rehearsed on the one-GPU box: `python bench.py --gpus 2 --oversubscribe --backend gloo` runs
    bench.main() end to end at world == 2 - self-launch under torch.distributed.run, broadcast of rank 0's weights, the step cut at the
    trunk boundary and captured as two HIP graphs per rank, the early exchange between the two replays and the late one behind them,
    MAX-over-ranks timing, ONE JSON line with n_gpus = 2 (main.py:372-376, util/dist.py:210-247).  Rank 0's averaged gradients of the
    last step must equal the mean of the two ranks' single-process gradients, computed here with the same weights and clips."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    dump = str(tmp_path / "grads.pt")
    steps, warmup, B = 2, 1, 2
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--oversubscribe", "--backend", "gloo", "--workload", "cfg1",
           "--clips-per-gpu", str(B), "--steps", str(steps), "--warmup", str(warmup), "--roofline-steps", "0", "--cpu-frames", "0",
           "--eval-dropout-off", "--grad-collective", collective, "--dump-grads", dump]
    env = dict(os.environ, TD_ALLOW_RANDOM_TEXT_ENCODER="1")
    env.pop("WORLD_SIZE", None)
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == steps and out["config"]["global_batch"] == 2 * B and out["config"]["parallelism"] == "dp2"
    assert out["execution"].startswith("2 hip_graphs"), out["execution"]
    assert out["gradient_exchange"].startswith(f"staged flat {collective}")
    assert out["process_group"]["backend"] == "gloo" and out["process_group"]["oversubscribed"]
    assert out["value"] > 0 and abs(out["value"] - 2 * B * steps / (out["ms_per_step"] * steps * 1e-3)) < 0.02 * out["value"]

    # the expected average, in this process: same init (seed 42 = rank 0's, broadcast to rank 1), same clips (seeds 1000 * rank + step)
    sys.path.insert(0, root)
    import bench as bench_mod
    import tubedetr_amd
    from tubedetr_amd.functional import invalidate_prepared
    from tubedetr_amd.harness import forward_step
    from tubedetr_amd.models import build_model

    got = torch.load(dump)
    assert got["world"] == 2
    T, res, k, L = bench_mod.WORKLOADS["cfg1"]
    dev = torch.device("cuda:0")
    torch.manual_seed(42)
    model, criterion, weight_dict = build_model(tubedetr_amd.default_args(stride=k, fast=True, no_tsa=False, compute_dtype=torch.bfloat16, video_max_len_train=max(200, T)))
    model.to(dev).eval()
    tok = bench_mod.BatchTokenizer()
    model.transformer.tokenizer = tok
    params = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
    assert [n for n, _ in params] == got["names"] and [p.numel() for _, p in params] == got["numels"]
    want = None
    for seed in got["batch_seeds"]:
        batch = bench_mod.make_batch(T, res, k, L, seed, dev, B, "u8")
        tok.batch = batch
        for _, p in params:
            p.grad = None
        invalidate_prepared()
        loss, _, _, _ = forward_step(model, criterion, weight_dict, batch)
        loss.backward()
        g = [None if p.grad is None else p.grad.detach().float().clone() for _, p in params]
        want = g if want is None else [a if b is None else (b if a is None else a + b) for a, b in zip(want, g)]
    flat, off, checked = got["flat"].to(dev), 0, 0
    for (n, p), w, used in zip(params, want, got["used"]):
        a = flat[off : off + p.numel()].view_as(p)
        off += p.numel()
        if w is None:
            assert not used, n  # RoBERTa's pooler: no rank used it
            continue
        ref = w / 2
        scale = ref.abs().max().clamp_min(1e-4)
        err = ((a - ref).abs().max() / scale).item()
        assert err < 2e-2, (n, err)  # bf16 kernels whose weight gradients accumulate with fp32 atomics: same bound as the 1-rank tests
        checked += 1
    assert checked > 300
