"""World-size-2 `gloo` tests (CPU) of the N>1 path: the data-parallel protocol bench.py / main.py use around the model
(two forward calls through the DDP wrapper before one backward, find_unused_parameters, gradient averaging), the
criterion's cross-rank num_boxes normalisation (no host sync), per-rank clip sharding and the max-over-ranks timing."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(fn, world=2):
    port = _free_port()
    mp.spawn(_entry, args=(world, port, fn), nprocs=world, join=True)


def _entry(rank, world, port, fn):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        fn(rank, world)
    finally:
        dist.destroy_process_group()


class TwoPhase(torch.nn.Module):
    """Stand-in with TubeDETR's calling convention: encode -> cache, decode(cache) -> outputs; one parameter (like
    RoBERTa's pooler) never reaches the loss."""

    def __init__(self):
        super().__init__()
        torch.manual_seed(0)
        self.enc = torch.nn.Linear(8, 8)
        self.dec = torch.nn.Linear(8, 4)
        self.pooler = torch.nn.Linear(8, 8)

    def forward(self, x, encode_and_save=True, memory_cache=None):
        if encode_and_save:
            return {"mem": torch.relu(self.enc(x))}
        return {"pred": self.dec(memory_cache["mem"])}


def _ddp_two_calls(rank, world):
    model = TwoPhase()
    ddp = torch.nn.parallel.DistributedDataParallel(model, find_unused_parameters=True)
    g = torch.Generator().manual_seed(1000 * rank)  # per-rank clip, like bench.py's 1000*rank+step seeds
    x = torch.randn(5, 8, generator=g)
    cache = ddp(x, encode_and_save=True)
    out = ddp(x, encode_and_save=False, memory_cache=cache)
    out["pred"].pow(2).sum().backward()
    assert model.pooler.weight.grad is None or model.pooler.weight.grad.abs().sum() == 0
    # reference: average over ranks of the single-process gradients
    ref = TwoPhase()
    grads = []
    for r in range(world):
        ref.zero_grad()
        xr = torch.randn(5, 8, generator=torch.Generator().manual_seed(1000 * r))
        ref(xr, False, ref(xr))["pred"].pow(2).sum().backward()
        grads.append([p.grad.clone() for p in (ref.enc.weight, ref.dec.weight)])
    for got, a, b in zip((model.enc.weight.grad, model.dec.weight.grad), grads[0], grads[1]):
        assert torch.allclose(got, (a + b) / world, atol=1e-6)


def _criterion_num_boxes(rank, world):
    from tubedetr_amd.models.tubedetr import SetCriterion

    crit = SetCriterion(["boxes", "sted", "guided_attn"], sigma=1)
    T = 4 + 2 * rank  # ranks hold different numbers of annotated frames
    g = torch.Generator().manual_seed(7 + rank)
    boxes = torch.rand(T, 4, generator=g) * 0.3 + 0.3
    out = {"pred_boxes": torch.rand(T, 4, generator=g) * 0.3 + 0.3, "pred_sted": torch.randn(1, T, 2, generator=g),
           "weights": torch.softmax(torch.randn(1, T, T, generator=g), -1)}
    targets = [{"boxes": b[None]} for b in boxes]
    tm = torch.ones(1, T, dtype=torch.bool)
    ld = crit(out, targets, [[0, T - 1]], tm)
    num_boxes = (4 + 6) / world  # all-reduced sum / world size (tubedetr.py:407-413)
    expect = (out["pred_boxes"] - boxes).abs().sum() / num_boxes
    assert torch.allclose(ld["loss_bbox"], expect, atol=1e-6)
    assert all(torch.isfinite(v) for v in ld.values())


def _timing_max(rank, world):
    t = torch.tensor([1.0 + rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert t.item() == float(world)  # bench.py reports the slowest rank's time


def _flat_grad_allreduce(rank, world):
    """tubedetr_amd.distributed.FlatGradAllReducer == averaging every gradient over the ranks; parameters without a
    gradient (RoBERTa's pooler in the real model) are exchanged as zeros and keep grad None; bf16 wire stays close."""
    from tubedetr_amd.distributed import FlatGradAllReducer, sync_num_boxes

    for wire in (torch.float32, torch.bfloat16):
        model = TwoPhase()
        x = torch.randn(5, 8, generator=torch.Generator().manual_seed(1000 * rank))
        model(x, False, model(x))["pred"].pow(2).sum().backward()
        assert model.pooler.weight.grad is None
        red = FlatGradAllReducer(model.parameters(), wire)
        assert red.numel == sum(p.numel() for p in model.parameters())
        red.reduce()
        ref = TwoPhase()
        want = None
        for r in range(world):
            ref.zero_grad()
            xr = torch.randn(5, 8, generator=torch.Generator().manual_seed(1000 * r))
            ref(xr, False, ref(xr))["pred"].pow(2).sum().backward()
            g = [ref.enc.weight.grad.clone(), ref.dec.bias.grad.clone()]
            want = g if want is None else [a + b for a, b in zip(want, g)]
        tol = 1e-6 if wire == torch.float32 else 2e-2
        for got, w in zip((model.enc.weight.grad, model.dec.bias.grad), want):
            assert torch.allclose(got, w / world, atol=tol * max(1.0, w.abs().max().item())), wire
        assert model.pooler.weight.grad is None
        # zero-copy hand-over: same values, .grad re-pointed at the flat exchange buffer, grad-less parameters stay None
        model2 = TwoPhase()
        model2(x, False, model2(x))["pred"].pow(2).sum().backward()
        red2 = FlatGradAllReducer(model2.parameters(), wire)
        red2.gather()
        red2.all_reduce()
        red2.attach()
        assert torch.equal(model2.enc.weight.grad, model.enc.weight.grad) and model2.pooler.weight.grad is None
        assert model2.enc.weight.grad.data_ptr() == red2.views[[id(p) for p in red2.params].index(id(model2.enc.weight))].data_ptr()
        red2.gather()  # gradients already living in the flat buffer are left alone
        assert torch.equal(model2.enc.weight.grad, model.enc.weight.grad)
    nb = sync_num_boxes(4 + 2 * rank, torch.zeros(1))
    assert nb.item() == (4 + 6) / world
    assert sync_num_boxes(0, torch.zeros(1)).item() == 1.0


def _flat_rank_dependent_usage(rank, world):
    """A parameter used on rank 0 only (DDP find_unused_parameters semantics): every rank ends up with the averaged
    gradient, both with the zero-copy attach and with scatter; the globally unused one stays None everywhere."""
    from tubedetr_amd.distributed import FlatGradAllReducer

    class M(torch.nn.Module):
        def __init__(self):
            super().__init__()
            torch.manual_seed(0)
            self.a, self.b, self.never = torch.nn.Linear(4, 4), torch.nn.Linear(4, 4), torch.nn.Linear(4, 4)

    x = torch.ones(2, 4)
    for mode in ("attach", "scatter"):
        m = M()
        loss = m.a(x).sum() + (m.b(x).pow(2).sum() if rank == 0 else 0.0)
        loss.backward()
        assert (m.b.weight.grad is None) == (rank != 0)
        red = FlatGradAllReducer(m.parameters())
        red.reduce(attach=(mode == "attach"))
        ref = M()
        ref.b(x).pow(2).sum().backward()
        assert m.b.weight.grad is not None and torch.allclose(m.b.weight.grad, ref.b.weight.grad / world, atol=1e-6), mode
        assert m.never.weight.grad is None and m.a.weight.grad is not None


def _flat_staged_overlap(rank, world):
    """Staged exchange (launch(early) while the last backward stage still runs, launch(late), finish) == one flat
    all-reduce; the early collectives are started before the late parameters even have a gradient."""
    from tubedetr_amd.distributed import FlatGradAllReducer

    class M(torch.nn.Module):
        def __init__(self):
            super().__init__()
            torch.manual_seed(0)
            self.head, self.trunk, self.tail, self.never = (torch.nn.Linear(4, 4) for _ in range(4))

    for wire in (torch.float32, torch.bfloat16):
        m = M()
        x = torch.randn(3, 4, generator=torch.Generator().manual_seed(rank))
        feat = m.trunk(x)
        leaf = feat.detach().requires_grad_()
        (m.head(leaf).sum() + m.tail(leaf).pow(2).sum()).backward()  # stage 1: stops at the trunk boundary
        red = FlatGradAllReducer(m.parameters(), wire, late=list(m.trunk.parameters()))
        assert [r[2] for r in red.runs] == [False, True, False]
        assert m.trunk.weight.grad is None
        red.launch(early=True)
        feat.backward(leaf.grad)  # stage 2
        red.launch(early=False)
        red.finish(attach=True)
        ref = M()
        want = None
        for r in range(world):
            ref.zero_grad()
            xr = torch.randn(3, 4, generator=torch.Generator().manual_seed(r))
            f = ref.trunk(xr)
            (ref.head(f).sum() + ref.tail(f).pow(2).sum()).backward()
            g = [p.grad.clone() for p in (ref.head.weight, ref.trunk.weight, ref.tail.bias)]
            want = g if want is None else [a + b for a, b in zip(want, g)]
        tol = 1e-6 if wire == torch.float32 else 2e-2
        for got, w in zip((m.head.weight.grad, m.trunk.weight.grad, m.tail.bias.grad), want):
            assert torch.allclose(got, w / world, atol=tol * max(1.0, w.abs().max().item())), wire
        assert m.never.weight.grad is None


def _flat_trunk_in_three_pieces(rank, world):
    """late_groups: the trunk's gradients leaving stage by stage (launch(stage=1), (2), (3) as ResNetBody.backward_trunk(after_stage=...) hands
    them over) give BIT FOR BIT the averaged buffer of the one late exchange (launch(early=False)) - both collectives, fp32 and bf16 wire - and
    the pieces can be started before the later stages have a gradient at all."""
    from tubedetr_amd.distributed import FlatGradAllReducer

    class M(torch.nn.Module):
        def __init__(self):
            super().__init__()
            torch.manual_seed(0)
            self.head = torch.nn.Linear(5, 3)
            self.l4, self.l3, self.l2 = torch.nn.Linear(7, 5), torch.nn.Linear(6, 7), torch.nn.Linear(4, 6)  # odd sizes: runs that do not divide by the world

    def grads(m):
        g = torch.Generator().manual_seed(100 + rank)
        for p in m.parameters():
            p.grad = torch.randn(p.shape, generator=g)

    for wire in (torch.float32, torch.bfloat16):
        for coll in ("all_reduce", "rs_ag"):
            m1, m2 = M(), M()
            grads(m1)
            one = FlatGradAllReducer(m1.parameters(), wire, late=list(m1.l4.parameters()) + list(m1.l3.parameters()) + list(m1.l2.parameters()), collective=coll)
            one.launch(early=True)
            one.launch(early=False)
            one.finish(attach=True)
            three = FlatGradAllReducer(m2.parameters(), wire, late_groups=[list(m2.l4.parameters()), list(m2.l3.parameters()), list(m2.l2.parameters())], collective=coll)
            assert three.n_late_stages == 3 and [r[2] for r in three.runs] == [0, 1, 2, 3]
            g = torch.Generator().manual_seed(100 + rank)
            vals = [torch.randn(p.shape, generator=g) for p in m2.parameters()]
            by = dict(zip([id(p) for p in m2.parameters()], vals))
            for p in m2.head.parameters():
                p.grad = by[id(p)]
            three.launch(early=True)
            for k_, mod in enumerate((m2.l4, m2.l3, m2.l2)):
                assert all(p.grad is None for later in (m2.l4, m2.l3, m2.l2)[k_:] for p in later.parameters())  # not yet computed
                for p in mod.parameters():
                    p.grad = by[id(p)]
                three.launch(stage=k_ + 1)
            three.finish(attach=True)
            assert torch.equal(one.flat, three.flat), (wire, coll)
            for a, b in zip(m1.parameters(), m2.parameters()):
                assert torch.equal(a.grad, b.grad)


def _eval_collectives(rank, world):
    """util/dist.py:34-122 of the reference: all_gather of picklable per-rank results of DIFFERENT sizes (the evaluators'
    prediction dicts) and reduce_dict of the loss dict (sum / average, one collective, key order independent of the rank)."""
    from tubedetr_amd.util.dist import all_gather, get_world_size, reduce_dict

    assert get_world_size() == world
    mine = {"rank": rank, "preds": {f"video_{rank}_{i}": {"sted": [i, i + rank + 1], "boxes": torch.arange(4.0 * (i + 1)).view(-1, 4) + rank} for i in range(3 + 5 * rank)},
            "blob": "x" * (17 + 1000 * rank)}
    got = all_gather(mine)
    assert len(got) == world and [g["rank"] for g in got] == list(range(world))
    for r, g in enumerate(got):
        assert len(g["preds"]) == 3 + 5 * r and len(g["blob"]) == 17 + 1000 * r
        assert g["preds"][f"video_{r}_2"]["sted"] == [2, 3 + r]
        assert torch.equal(g["preds"][f"video_{r}_1"]["boxes"], torch.arange(8.0).view(-1, 4) + r)
    # ranks insert the keys in different orders; values differ per rank
    keys = ["loss_bbox", "loss_giou", "loss_sted", "loss_guided_attn_3"]
    order = keys if rank == 0 else keys[::-1]
    d = {k: torch.tensor(float(keys.index(k) + 1) * (rank + 1)) for k in order}
    avg = reduce_dict(d, average=True)
    tot = reduce_dict(d, average=False)
    ranks_sum = sum(r + 1 for r in range(world))
    for k in keys:
        assert abs(tot[k].item() - (keys.index(k) + 1) * ranks_sum) < 1e-6
        assert abs(avg[k].item() - (keys.index(k) + 1) * ranks_sum / world) < 1e-6
    assert all(torch.equal(d[k], torch.tensor(float(keys.index(k) + 1) * (rank + 1))) for k in keys)  # inputs untouched


def _flat_rs_ag(rank, world):
    """collective="rs_ag" (reduce-scatter + all-gather on the flat buffer, the remainder that does not divide by the world size
    through a small all-reduce) gives bit for bit what the single all-reduce gives - plain and staged, fp32 and bf16 wire, a total
    length that is odd."""
    from tubedetr_amd.distributed import FlatGradAllReducer

    class M(torch.nn.Module):
        def __init__(self):
            super().__init__()
            torch.manual_seed(0)
            self.head, self.trunk, self.tail = torch.nn.Linear(5, 3), torch.nn.Linear(7, 5), torch.nn.Linear(5, 3, bias=False)

    def grads(m):
        g = torch.Generator().manual_seed(77 + rank)
        for p_ in m.parameters():
            p_.grad = torch.randn(p_.shape, generator=g)

    for wire in (torch.float32, torch.bfloat16):
        res = {}
        for coll in ("all_reduce", "rs_ag"):
            m = M()
            assert sum(p_.numel() for p_ in m.parameters()) % 2 == 1
            grads(m)
            red = FlatGradAllReducer(m.parameters(), wire, collective=coll)
            red.reduce(attach=True)
            m2 = M()
            grads(m2)
            red2 = FlatGradAllReducer(m2.parameters(), wire, late=list(m2.trunk.parameters()), collective=coll)
            red2.launch(early=True)
            red2.launch(early=False)
            red2.finish(attach=True)
            res[coll] = (red.flat.clone(), red2.flat.clone())
        for a, b in zip(res["all_reduce"], res["rs_ag"]):
            assert torch.equal(a, b), (wire, (a - b).abs().max())
        assert torch.equal(res["rs_ag"][0], res["rs_ag"][1])  # staged == plain


@pytest.mark.parametrize("fn", [_ddp_two_calls, _criterion_num_boxes, _timing_max, _flat_grad_allreduce, _flat_rank_dependent_usage, _flat_staged_overlap, _flat_trunk_in_three_pieces,
                                _flat_rs_ag, _eval_collectives])
def test_world_size_2_gloo(fn):
    _run(fn)


def test_eval_collectives_single_process():
    from tubedetr_amd.util.dist import all_gather, reduce_dict

    d = {"a": torch.tensor(1.0)}
    assert all_gather({"x": 1}) == [{"x": 1}] and reduce_dict(d) is d  # world size 1: identity, like util/dist.py:44-45,108-109


def test_bench_batches_are_sharded_by_rank():
    import bench

    a = bench.make_batch(4, 32, 2, 5, 1000 * 0 + 3, torch.device("cpu"))
    b = bench.make_batch(4, 32, 2, 5, 1000 * 1 + 3, torch.device("cpu"))
    assert tuple(a["frames"].shape) == (2, 3, 32, 32) and a["frames_fast"].shape == (4, 3, 32, 32)
    assert torch.equal(a["frames"].materialize(), a["frames_fast"][::2])  # slow = every k-th fast frame (vidstg.py:250-251), as an index list
    c = bench.make_batch(4, 32, 2, 5, 3, torch.device("cpu"), clips=2)
    assert torch.equal(c["frames"].materialize(), torch.cat([c["frames_fast"][0:4:2], c["frames_fast"][4:8:2]])) and c["durations"] == [4, 4]
    assert not torch.equal(a["frames_fast"], b["frames_fast"])       # different clip per rank
    assert a["durations"] == [4] and a["inter_idx"] == [[0, 3]] and a["target_boxes"].shape == (4, 4)


def test_bench_self_launches_its_ranks():
    """`python bench.py --gpus N` without torchrun starts its own ranks (one per GPU, rendezvous on 127.0.0.1) and keeps the one-line
    contract; on a box without N devices every rank says so instead of dying in an assert (util/dist.py:210-247 is the reference's side)."""
    import json
    import subprocess
    import sys

    import bench

    argv = bench.launcher_argv(4, ["--gpus", "4", "--steps", "7", "--dry-launch", "--warmup", "2"], 29777)
    assert argv[:3] == [sys.executable, "-m", "torch.distributed.run"] and "--nproc-per-node=4" in argv and "--nnodes=1" in argv
    assert argv[argv.index("--master-addr") + 1] == "127.0.0.1" and argv[argv.index("--master-port") + 1] == "29777"
    i = argv.index(os.path.abspath(bench.__file__))
    assert argv[i + 1:] == ["--gpus", "4", "--steps", "7", "--warmup", "2"]  # the ranks get the caller's flags, minus the launcher's own
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, bench.__file__, "--gpus", "2", "--steps", "3", "--dry-launch"], capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode == 0
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and "--nproc-per-node=2" in json.loads(lines[0])["launch"]
    if not torch.cuda.is_available():  # no device here: the launched ranks must explain themselves and the exit code must propagate
        r = subprocess.run([sys.executable, bench.__file__, "--gpus", "2", "--steps", "1", "--warmup", "0"], capture_output=True, text=True, env=env, timeout=300)
        assert r.returncode != 0 and "needs 2 GPUs on this node" in r.stderr and r.stdout.strip() == ""


def test_bench_rehearsal_flags_reach_the_ranks_and_fail_cleanly_without_a_device():
    """--oversubscribe / --backend gloo (the one-GPU rehearsal of BASELINE config 4's entry point, tests/test_distributed_gpu.py) are
    ordinary flags: the self-launcher hands them to the ranks; with NO device at all the ranks still say what is missing; RCCL with
    more ranks than devices is refused before any process group exists."""
    import subprocess
    import sys

    import bench

    argv = bench.launcher_argv(2, ["--gpus", "2", "--oversubscribe", "--backend", "gloo", "--dump-grads", "/tmp/x.pt"], 29778)
    i = argv.index(os.path.abspath(bench.__file__))
    assert argv[i + 1:] == ["--gpus", "2", "--oversubscribe", "--backend", "gloo", "--dump-grads", "/tmp/x.pt"]
    if not torch.cuda.is_available():
        env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
        r = subprocess.run([sys.executable, bench.__file__, "--gpus", "2", "--oversubscribe", "--backend", "gloo", "--steps", "1", "--warmup", "0"],
                           capture_output=True, text=True, env=env, timeout=300)
        assert r.returncode != 0 and "has no device" in r.stderr and r.stdout.strip() == ""


def test_set_deterministic_switches_the_library_and_torch():
    """The library's flag is one atomic (td_set_deterministic / td_get_deterministic: no environment access on the launch path); torch's
    own setting is restored to what it WAS, not forced off, and importing the package touches neither."""
    import tubedetr_amd

    assert not torch.are_deterministic_algorithms_enabled()  # importing the package changed nothing
    try:
        tubedetr_amd.set_deterministic(True)
        assert tubedetr_amd.is_deterministic() and torch.are_deterministic_algorithms_enabled()
    finally:
        tubedetr_amd.set_deterministic(False)
    assert not tubedetr_amd.is_deterministic() and not torch.are_deterministic_algorithms_enabled()
    # a setting the user made is given back
    torch.use_deterministic_algorithms(True, warn_only=False)
    try:
        tubedetr_amd.set_deterministic(True)
        tubedetr_amd.set_deterministic(False)
        assert torch.are_deterministic_algorithms_enabled() and not torch.is_deterministic_algorithms_warn_only_enabled()
    finally:
        torch.use_deterministic_algorithms(False)
