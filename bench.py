#!/usr/bin/env python3
"""bench.py - training clips/sec (forward + backward) of the TubeDETR hot path on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N>1 it is launched under
torch.distributed.run with one rank per GPU (RCCL).  A "step" = one pass of the hot path over one synthetic batch
per GPU: the two model calls of engine.py:67-80 (video-text encoder, then space-time decoder), the criterion, and
the backward pass; at N>1 plus the gradient exchange.  Rank 0 prints ONE JSON line.

Workload at N=1: BASELINE.json configs[2] (the config the metric is quoted on): T=100 frames, stride k=4, res=352,
L=30 text tokens, bf16 MFMA kernels with fp32 accumulation, random-init weights, train mode (dropout active),
weights re-prepared every step (as after an optimizer step), all 125 trunk-forward frames of every clip executed
(`--dedupe` skips the 25 slow frames inside the fast pass).  `--clips-per-gpu B` videos per GPU per step (the
reference's --batch_size, main.py:63; default 16: 288 GB of HBM hold the 111 GB this batch needs, and every latency-bound
launch of the step - the 100-row-per-clip decoder, RoBERTa on 30 tokens per clip, the 12 100-row trunk backward - then does B
times the work; measured in round 3: 88.9 / 95.1 clips/s at B = 8 / 16 on one box, profiles/README.md): `value` counts
clips, not steps.  Up to B = 8 the slow and the fast frames of a step share ONE trunk pass (1 000 frames); beyond that a pass
that keeps activations for backward would exceed the kernels' 32-bit tensor addressing (1 083 bf16 frames of res 352): the slow
frames (kept for backward) and the no-grad fast frames (1 600, ONE td_resnet_fwd call whose layer1 launches go out in two frame
groups) run as two passes.  Inputs are generated on the device before the timed region.

Execution: the step is captured once in HIP graph(s) and replayed (`--no-graph`: eager launches; GPU-bound too from B = 4 on).
  N = 1 : one graph, RoBERTa on a forked branch (its 120-row GEMMs overlap the trunk; `--no-text-stream`: linear graph).
  N > 1 : the step is cut at the ResNet trunk boundary (harness.backward_in_stages) and captured as TWO graphs (RoBERTa
          still on a forked branch of the first); the
          all-reduce of the gradients that are final after the first one (heads, decoder, encoder, RoBERTa, input_proj:
          0.57 of 0.74 GB) is started between the two replays and overlaps the trunk backward, the trunk's own 0.17 GB
          follow (tubedetr_amd/distributed.py).  `--no-overlap`: one graph + one flat all-reduce after it; `--ddp`: torch
          DistributedDataParallel like main.py:372-376.
At N=1 the measurement runs in a child process; if it dies the parent re-measures (linear graph, then eager launches), so
a bench line is always produced; `attempts` in the JSON records what happened.

Extra legs (rank 0, after the timed region, not part of `value`):
  roofline     : `--roofline-steps` more identical steps, launched eagerly, with HIP events recorded on the launch
                 stream around every launch of the MFMA kernel families; per family achieved = algorithmic FLOPs or
                 algorithmic HBM bytes / summed event-bracketed duration (nothing subtracted: the bracketing itself adds a
                 few us per launch, reported as `event_pair_overhead_us`, so `frac` errs low against rocprofv3), bound = the
                 roof it sits closer to.  `traffic` / `mfma_util` are NOT measured by this process: they are the
                 rocprofv3 PMC results committed under profiles/ (counters cannot be read from inside the timed process).
                 `roofline` is the family with the largest share of the step, the others follow in `other_mfma_kernels`.
  cpu_baseline : the CPU oracle (oracle/, a port of the reference algorithm) timed on the host cores: fwd+bwd of a T=`--cpu-frames` clip of the
                 same resolution (default 100 = the benchmarked clip, ~18 s per pass), one short warm-up + median of 3 passes.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("TD_ALLOW_RANDOM_TEXT_ENCODER", "1")  # synthetic benchmark: random-init roberta-base geometry (no files offline)

WORKLOADS = {
    # name: (T, res, k, L)
    "cfg3": (100, 352, 4, 30),   # headline: res=352 k=4 T=100 L=30
    "cfg2": (64, 224, 2, 20),
    "cfg1": (8, 224, 5, 20),
}
PEAK_HBM_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s
PEAK_BF16_TFLOPS = 2500.0  # dense bf16 MFMA peak, MI355X_MICROARCH.md
# algorithmic TFLOP per clip fwd+bwd
ALGO_TFLOP_PER_CLIP = {"cfg3": 6.847, "cfg2": 2.538, "cfg1": 0.233}
DEFAULT_CLIPS_PER_GPU = 16
PMC_TRAFFIC = "r06_pmc_traffic.json"
PMC_MFMA = "r06_pmc_mfma.json"


def make_batch(T, res, k, L, seed, device, clips=1, frames="u8"):
    """SURVEY.md 8d synthetic clips, generated directly in HBM: slow = video[::k], fast = all frames; `clips` videos of equal
    duration per batch (video-major frame order, like util/misc.py's collate).  frames="u8" (default): decoded uint8 pixels, what
    the device-side input path takes (tubedetr_amd/data.py: ImageNet normalisation, NHWC and the bf16 cast happen in the trunk's
    input kernel); frames="fp32": host-normalised fp32 frames ~ N(0,1), the reference's collate format (util/misc.py:106-178)."""
    g = torch.Generator(device=device).manual_seed(seed)
    if frames == "u8":
        video = torch.randint(0, 256, (clips * T, 3, res, res), generator=g, device=device, dtype=torch.uint8)
    else:
        video = torch.randn(clips * T, 3, res, res, generator=g, device=device)
    ids = torch.randint(3, 50000, (clips, L), generator=torch.Generator().manual_seed(seed))  # token ids start on the host, like a tokenizer's output
    ids[:, 0], ids[:, -1] = 0, 2
    cxcy = torch.rand(clips * T, 2, generator=g, device=device) * 0.6 + 0.2
    wh = torch.rand(clips * T, 2, generator=g, device=device) * 0.3 + 0.1
    from tubedetr_amd.util.misc import FrameSources

    n_slow = math.ceil(T / k)
    slow_idx = (torch.arange(clips)[:, None] * T + torch.arange(0, T, k)[None, :]).reshape(-1).to(torch.int32).to(device)
    return {
        "frames": FrameSources([(video, slow_idx)], None, [tuple(int(c_ * T + j) for c_ in range(clips) for j in range(0, T, k))]),  # slow clip = video[::k] of every video: an index list over the same pixels (no copy) + its host copy (what data.ClipPipeline hands over)
        "frames_mask": torch.zeros((clips * n_slow, res, res), dtype=torch.bool, device=device),
        "frames_fast": video,
        "fast_mask": torch.zeros((clips * T, res, res), dtype=torch.bool, device=device),
        "durations": [T] * clips,
        "input_ids": ids,
        "attention_mask": torch.ones(clips, L, dtype=torch.long),
        "target_boxes": torch.cat([cxcy, wh], 1),
        "inter_idx": [[0, T - 1]] * clips,
    }


class BatchTokenizer:
    """Feeds the current synthetic batch's token ids to the model (no tokenizer files offline)."""

    def __init__(self):
        self.batch = None

    def batch_encode_plus(self, text, padding="longest", return_tensors="pt"):
        from transformers import BatchEncoding

        be = BatchEncoding({"input_ids": self.batch["input_ids"].clone(), "attention_mask": self.batch["attention_mask"].clone()})
        be._encodings = [None] * len(text)
        be._td_no_padding = True  # synthetic captions have no padding (mask is all ones by construction)
        return be


def cpu_model_name() -> str:
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.lower().startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def cpu_baseline(T_sample, res, k, L, T_full):
    """Reference algorithm on the host cores (oracle port), fwd+bwd of a T_sample-frame clip: one warm-up pass on a SHORT clip (thread
    pool, allocator), then the median of THREE measured passes.  Default T_sample = T_full (100): a measured pass of the benchmarked
    clip, nothing scaled; a shorter sample is scaled and labelled so."""
    from oracle.tubedetr_oracle import OracleConfig, train_step
    from oracle.weights import fill_state, state_spec, synthetic_batch

    cores = min(os.cpu_count() or 1, 32)  # more threads than this only slows the small-batch CPU convolutions down
    torch.set_num_threads(cores)
    cfg = OracleConfig(stride=k)
    sd = fill_state(state_spec(cfg), 1, requires_grad=True)
    times = []
    for it in range(4):  # pass 0 = warm-up on 2 k frames, passes 1-3 are measured
        batch = synthetic_batch(T=(2 * k if it == 0 else T_sample), res=res, k=k, L=L, seed=5)
        for v in sd.values():
            v.grad = None
        t0 = time.time()
        loss, _, _, _ = train_step(sd, cfg, batch)
        loss.backward()
        dt = time.time() - t0
        if it > 0:
            times.append(dt)
    times.sort()
    med = times[len(times) // 2]
    per_clip = med * (T_full / T_sample)
    scaled = T_sample != T_full
    return {"value": 1.0 / per_clip, "unit": "clips/s", "cores": cores, "cpu": cpu_model_name(), "host_logical_cpus": os.cpu_count(),
            "kind": "port", "kind_detail": (f"port, scaled from T={T_sample} to T={T_full} (not a measured T={T_full} pass)" if scaled else
                                            f"port, measured passes of the T={T_full} clip (nothing scaled)"),
            "sample": f"fwd+bwd of a T={T_sample} clip (k={k}, res={res}, L={L}) by the CPU oracle, median {med:.1f}s of {len(times)} passes "
                      f"({', '.join(f'{t_:.1f}' for t_ in times)}) after one warm-up pass on a {2 * k}-frame clip"
                      + (f", scaled x{T_full}/{T_sample} to T={T_full}" if scaled else "")}


def launcher_argv(n_gpus, bench_args, port):
    """Command line that starts `n_gpus` ranks of this script on this node: one process per GPU under torch.distributed.run, rendezvous on
    127.0.0.1 (the container hostname may not resolve).  Same shape as the driver's own N > 1 command."""
    rest = [x for x in bench_args if x != "--dry-launch"]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + rest


def free_port():
    import socket

    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        return s_.getsockname()[1]


def self_launch(n_gpus, bench_args, dry=False):
    import subprocess

    argv = launcher_argv(n_gpus, bench_args, free_port())
    if dry:
        print(json.dumps({"launch": argv}), flush=True)
        return 0
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs on this host driver
    r = subprocess.run(argv, stdout=subprocess.PIPE, text=True, env=env)
    line = next((ln for ln in reversed(r.stdout.splitlines()) if ln.startswith("{")), None)
    if line:
        print(line, flush=True)
    elif r.returncode == 0:
        print("[bench] the ranks exited cleanly without a JSON line", file=sys.stderr, flush=True)
        return 1
    return r.returncode


def _trace(msg):
    if os.environ.get("TD_BENCH_TRACE"):
        print(f"[bench] {msg}", file=sys.stderr, flush=True)


def _load_profile_json(name):
    try:
        return json.load(open(os.path.join(ROOT, "profiles", name)))
    except Exception:
        return {}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="cfg3", choices=list(WORKLOADS))
    ap.add_argument("--clips-per-gpu", type=int, default=DEFAULT_CLIPS_PER_GPU, help="videos per GPU per step (the reference's --batch_size, main.py:63)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--roofline-steps", type=int, default=1)
    ap.add_argument("--cpu-frames", type=int, default=100, help="frames of the CPU-baseline sample clip (0 = skip); 100 = the benchmarked clip itself: ~18 s per pass, 3 passes")
    ap.add_argument("--keep-prepared-weights", action="store_true", help="diagnostic: reuse prepared bf16 weights across steps")
    ap.add_argument("--dedupe", action="store_true",
                    help="do not recompute the slow frames inside the fast pass (exact, slow = video[::k]); off by default so the timed step "
                         "executes the same work as the reference's")
    ap.add_argument("--dedupe-steps", type=int, default=5, help="N=1: eager steps timed with the dead work skipped, reported as value_dedupe beside the headline (0 = skip)")
    ap.add_argument("--graph", dest="graph", action="store_true", default=True, help="capture the step in HIP graph(s)")
    ap.add_argument("--no-graph", dest="graph", action="store_false")
    ap.add_argument("--force-ddp", action="store_true", help="diagnostic: run the N>1 code path (process group + gradient exchange) with one rank")
    ap.add_argument("--ddp", action="store_true", help="N>1: use torch DistributedDataParallel like main.py:372-376 instead of the flat exchange")
    ap.add_argument("--no-overlap", action="store_true", help="N>1: one flat all-reduce after the whole backward instead of the staged, overlapped exchange")
    ap.add_argument("--trunk-pieces", type=int, default=1, choices=[1, 3],
                    help="N>1, staged exchange: 3 = the trunk's backward is issued stage by stage (layer4 | layer3 | layer2), each stage's weight gradients in a launch "
                         "of their own, and leave in three pieces under the remaining backward instead of one piece behind it (3 or 4 HIP graphs instead of 2)")
    ap.add_argument("--grad-wire-dtype", default="fp32", choices=["fp32", "bf16"], help="dtype of the gradient all-reduce on the wire")
    ap.add_argument("--grad-collective", default="all_reduce", choices=["all_reduce", "rs_ag"],
                    help="N>1: how the flat gradient buffer is averaged: one all-reduce per stage, or reduce-scatter + all-gather on the flat buffer (same result)")
    ap.add_argument("--frames", default="u8", choices=["u8", "fp32"],
                    help="input frames: u8 = decoded uint8 pixels normalised on the device (the input path of tubedetr_amd/data.py, default); "
                         "fp32 = host-normalised fp32 frames, the reference's collate format")
    ap.add_argument("--no-fast", action="store_true")
    ap.add_argument("--no-tsa", action="store_true")
    ap.add_argument("--eval-dropout-off", action="store_true", help="diagnostic only: run in eval mode")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="N>1: process-group backend.  nccl = RCCL over xGMI (one rank per GPU, the measured configuration); gloo = the collectives "
                         "staged through the host - with --oversubscribe the way to run the whole N>1 control flow on a box with fewer GPUs than ranks")
    ap.add_argument("--oversubscribe", action="store_true",
                    help="rehearsal: rank r uses device r %% (visible devices) instead of failing when there are fewer GPUs than ranks "
                         "(RCCL refuses two ranks on one device: use --backend gloo); the JSON line is labelled, it is not a scaling measurement")
    ap.add_argument("--dump-grads", default=None, help="rank 0 writes the averaged flat gradient buffer of the LAST timed step (and its layout) to this file (tests)")
    ap.add_argument("--child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--dry-launch", action="store_true", help="--gpus N > 1 without a launcher: print the launcher command line as JSON and exit")
    ap.add_argument("--text-stream", action="store_true", help="(default at N=1; accepted for older command lines)")
    ap.add_argument("--no-text-stream", action="store_true",
                    help="graph mode: capture RoBERTa on the main stream (a linear graph) instead of its own stream (a forked graph branch: "
                         "its latency-bound 120-row GEMMs then overlap the trunk, ~2 ms per step at 4 clips; the default at N=1)")
    a = ap.parse_args()

    if os.environ.get("TD_EFENCE") == "1":  # diagnostic: electric-fence device allocator (tests/efence/), eager launches only
        sys.path.insert(0, os.path.join(ROOT, "tests", "efence"))
        import install as efence_install

        efence_install.install()
        a.graph, a.child = False, True
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ and not a.child:
        # `python bench.py --gpus N` without an external launcher: start the N ranks ourselves (what util/dist.py:210-247 gets from
        # torch.distributed.run's environment), pass rank 0's single JSON line through and propagate the exit code
        sys.exit(self_launch(a.gpus, sys.argv[1:], dry=a.dry_launch))
    if world == 1 and not a.child and not a.force_ddp and a.graph and os.environ.get("TD_BENCH_ISOLATE", "1") != "0":
        # Single-GPU run: the measurement happens in a child process; should it die, the parent re-measures with eager
        # launches instead of losing the bench line.
        import subprocess

        argv = [sys.executable, os.path.abspath(__file__)] + sys.argv[1:] + ["--child"]
        attempts = []
        for extra in ([], ["--no-text-stream"], ["--no-graph"], ["--no-graph", "--clips-per-gpu", "4"]):  # forked graph, linear graph, eager launches, half the batch
            r = subprocess.run(argv + extra, stdout=subprocess.PIPE, text=True)
            line = next((ln for ln in reversed(r.stdout.splitlines()) if ln.startswith("{")), None)
            attempts.append({"args": extra, "returncode": r.returncode})
            if r.returncode == 0 and line:
                out = json.loads(line)
                out["attempts"] = attempts
                print(json.dumps(out), flush=True)
                return
            print(f"[bench] child {extra} failed with exit code {r.returncode}; retrying", file=sys.stderr, flush=True)
        raise SystemExit("bench: every attempt failed")
    # The process's stdout carries exactly ONE line, the JSON: native libraries write there too (RCCL prints a version banner
    # through C stdio when the first communicator is created), so file descriptor 1 is pointed at stderr for the whole run
    # and the JSON line is written to the saved descriptor at the end.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    a.text_stream = a.graph and not a.no_text_stream and not a.ddp
    if a.graph and not a.text_stream:
        os.environ.setdefault("TD_TEXT_STREAM", "0")  # single-stream capture: RoBERTa stays on the main stream
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"bench: --gpus {a.gpus} but WORLD_SIZE={world}: launch as `python bench.py --gpus {a.gpus}` (self-launching) or under "
                         f"torch.distributed.run --nproc-per-node {a.gpus}")
    n_dev = torch.cuda.device_count()
    if local_rank >= n_dev and not (a.oversubscribe and n_dev > 0):
        raise SystemExit(f"bench: rank {rank} (local rank {local_rank}) has no device: --gpus {a.gpus} needs {a.gpus} GPUs on this node, {n_dev} visible "
                         f"(--oversubscribe --backend gloo rehearses the N>1 path on fewer devices)")
    if a.oversubscribe and a.backend == "nccl" and world > n_dev:
        raise SystemExit("bench: --oversubscribe with more ranks than devices needs --backend gloo (RCCL refuses two ranks on one device)")
    dev_index = local_rank % n_dev if a.oversubscribe else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1 or a.force_ddp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29512")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if a.backend == "nccl":
            torch.distributed.init_process_group("nccl", device_id=dev)  # (util/dist.py:210-247: the launcher's environment)
        else:
            torch.distributed.init_process_group("gloo")

    import tubedetr_amd
    from tubedetr_amd import _hip
    from tubedetr_amd import ops as ops_
    from tubedetr_amd.functional import invalidate_prepared, set_wgrad_deferral
    from tubedetr_amd.harness import backward_in_stages, forward_step, set_split_backward
    from tubedetr_amd.models import build_model

    T, res, k, L = WORKLOADS[a.workload]
    B = max(1, a.clips_per_gpu)
    cdt = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    torch.manual_seed(42 + rank)  # main.py:358
    args = tubedetr_amd.default_args(stride=k, fast=not a.no_fast, no_tsa=a.no_tsa, compute_dtype=cdt, video_max_len_train=max(200, T))
    model, criterion, weight_dict = build_model(args)
    model.to(dev)
    # The headline line executes the reference's full work: every one of the 125 trunk-forward frames of a clip, although the model can PROVE from
    # its inputs that 25 of them are computed twice (None = prove it and skip them, the product's default; --dedupe selects that for the timed steps,
    # and `value_dedupe` in the JSON line is that mode timed in the same run either way).
    model.slow_frames_are_strided_fast = None if a.dedupe else False
    model.train(not a.eval_dropout_off)
    tok = BatchTokenizer()
    model.transformer.tokenizer = tok
    net = model
    distributed = world > 1 or a.force_ddp
    staged = distributed and not a.ddp and not a.no_overlap
    reducer, pieces = None, 1
    if distributed and a.ddp:
        set_wgrad_deferral(False)  # DDP's reducer hooks read every gradient the moment autograd produces it
        net = torch.nn.parallel.DistributedDataParallel(model, device_ids=[local_rank], find_unused_parameters=True)  # main.py:372-376
    elif distributed:
        # replicas start from rank 0's weights (what DDP's constructor does), then exchange gradients through one flat
        # buffer (tubedetr_amd/distributed.py); the forward / backward of the step itself holds no collective
        from tubedetr_amd.distributed import FlatGradAllReducer, broadcast_, sync_num_boxes

        for t_ in list(model.parameters()) + list(model.buffers()):
            broadcast_(t_.data, 0)
        late = [p_ for n_, p_ in model.named_parameters() if n_.startswith("backbone.") and p_.requires_grad] if staged else None
        pieces = a.trunk_pieces if staged else 1
        if pieces > 1:
            from tubedetr_amd.harness import trunk_stage_groups

            reducer = FlatGradAllReducer(model.parameters(), torch.bfloat16 if a.grad_wire_dtype == "bf16" else torch.float32, late_groups=trunk_stage_groups(model), collective=a.grad_collective)
            assert reducer.n_late_stages == 3 and sum(reducer.is_late) == len(late)
        else:
            reducer = FlatGradAllReducer(model.parameters(), torch.bfloat16 if a.grad_wire_dtype == "bf16" else torch.float32, late=late, collective=a.grad_collective)
        criterion.external_num_boxes = torch.ones(1, dtype=torch.float32, device=dev)
        reducer.always_communicate = a.force_ddp  # exercise the RCCL call in the 1-rank diagnostic
        set_split_backward(model, staged)

    n_batches = a.warmup + a.steps + a.roofline_steps
    batches = [make_batch(T, res, k, L, 1000 * rank + s, dev, B, a.frames) for s in range(min(n_batches, 4))]
    params = [p_ for p_ in model.parameters() if p_.requires_grad]

    def eager_step(i):
        b = batches[i % len(batches)]
        tok.batch = b
        if not a.keep_prepared_weights:
            invalidate_prepared()  # as after an optimizer step: weights are re-cast / re-folded inside the timed step
        for p_ in params:  # = optimizer.zero_grad(set_to_none=True) without re-walking the module tree
            p_.grad = None
        if reducer is not None:
            sync_num_boxes(b["target_boxes"].shape[0], criterion.external_num_boxes)
        loss, _, _, _ = forward_step(net, criterion, weight_dict, b)
        if staged and pieces > 1:
            backward_in_stages(model, loss, after_first_stage=lambda: reducer.launch(early=True), after_trunk_stage=lambda k_, ws_: reducer.launch(stage=k_))
            reducer.finish(attach=True)
        elif staged:
            backward_in_stages(model, loss, after_first_stage=lambda: reducer.launch(early=True))  # exchange overlaps the trunk backward
            reducer.launch(early=False)
            reducer.finish(attach=True)
        else:
            loss.backward()
            if reducer is not None:
                reducer.reduce(attach=True)  # gather, ONE all-reduce, .grad re-pointed at the flat buffer (no copy back)
        return loss

    step = eager_step

    def fence():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    # ---- HIP graph(s): the ~1500 launches of a step are captured once and replayed, so the host only copies the next
    # batch into the static input buffers and bumps the dropout step counter ----
    execution = "eager"
    if a.graph and not (distributed and a.ddp):
        try:
            static = {k_: (v.clone() if torch.is_tensor(v) else v) for k_, v in batches[0].items()}
            static["frames"] = type(batches[0]["frames"])([(static["frames_fast"], batches[0]["frames"].parts[0][1])], None, batches[0]["frames"].index_host)  # slow clip: index list over the static video
            for k_ in ("input_ids", "attention_mask"):
                static[k_] = static[k_].to(dev)
            counter = torch.zeros(1, dtype=torch.int32, device=dev)
            ops_.set_dropout_counter(counter)

            def body1():
                tok.batch = static
                invalidate_prepared()
                l_, _, _, _ = forward_step(net, criterion, weight_dict, static)
                l_.backward()  # staged: stops at the trunk boundary
                if reducer is not None:
                    if staged:
                        reducer.gather_stage(early=True)
                    else:
                        reducer.gather()  # no collective: part of the captured step
                return l_

            def body2():
                model.backbone[0].body.backward_trunk()
                reducer.gather_stage(early=False)

            def body2_pieces():  # generator: one trunk stage + the gather of its gradients per next()
                for st_, _ in model.backbone[0].body.backward_trunk_iter():
                    reducer.gather_stage(stage=4 - st_)
                    yield 4 - st_

            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(3):
                    for p_ in params:
                        p_.grad = None
                    body1()
                    if staged and pieces > 1:
                        for _ in body2_pieces():
                            pass
                    elif staged:
                        body2()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            for p_ in params:
                p_.grad = None
            _trace("eager warm-up on the capture side stream done")
            # with a process group alive its watchdog thread polls HIP events: only THIS thread's calls are held to capture rules
            cap_mode = "thread_local" if distributed else "global"
            graph = torch.cuda.CUDAGraph()
            ops_.reset_capture_arena()
            with torch.cuda.graph(graph, capture_error_mode=cap_mode):
                static_loss = body1()
            graph2, graphs2 = None, []
            if staged and pieces > 1:
                it2 = body2_pieces()
                for _ in range(reducer.n_late_stages):
                    g2 = torch.cuda.CUDAGraph()
                    ops_.reset_capture_arena()
                    with torch.cuda.graph(g2, pool=graph.pool(), capture_error_mode=cap_mode):
                        k2 = next(it2)
                    graphs2.append((g2, k2))
                assert next(it2, None) is None
            elif staged:
                graph2 = torch.cuda.CUDAGraph()
                ops_.reset_capture_arena()
                with torch.cuda.graph(graph2, pool=graph.pool(), capture_error_mode=cap_mode):
                    body2()
            torch.cuda.synchronize()
            _trace("capture done")
            if reducer is not None:
                reducer.attach()  # from now on .grad of every parameter is its (averaged) slice of the flat buffer

            def step(i):  # noqa: F811
                b_ = batches[i % len(batches)]
                for k_, v in b_.items():
                    if torch.is_tensor(v):
                        static[k_].copy_(v, non_blocking=True)
                counter.add_(1)
                if reducer is not None:
                    sync_num_boxes(b_["target_boxes"].shape[0], criterion.external_num_boxes)
                graph.replay()
                if staged and graphs2:
                    reducer.exchange_stage(early=True)
                    for g2_, k2_ in graphs2:                # layer4 | layer3 | layer2: each stage's gradients leave while the next stage runs
                        g2_.replay()
                        reducer.exchange_stage(stage=k2_)
                    reducer.finish(attach=None)
                elif staged:
                    reducer.exchange_stage(early=True)   # 0.57 GB, overlaps the second graph (trunk backward)
                    graph2.replay()
                    reducer.exchange_stage(early=False)  # the trunk's 0.17 GB
                    reducer.finish(attach=None)
                elif reducer is not None:
                    reducer.all_reduce()  # the only collective of the step, outside the graph
                return static_loss

            execution = ("hip_graph (text encoder on a forked branch)" if a.text_stream else "hip_graph (linear)") if not staged else \
                (f"{1 + max(1, len(graphs2))} hip_graphs (cut at the trunk boundary" + (" and between the trunk's stages" if graphs2 else "") + ", exchange overlapped" + ("; text encoder on a forked branch of the first)" if a.text_stream else ")"))
        except Exception as exc:  # capture not possible: measure the eager path
            ops_.set_dropout_counter(None)
            torch.cuda.synchronize()
            step = eager_step
            execution = f"eager (graph capture failed: {type(exc).__name__}: {str(exc)[:120]})"

    if os.environ.get("TD_BENCH_MEMMAP"):  # fault triage: where every allocator segment / block lives before the replays start
        torch.cuda.synchronize()
        snap = [{"address": s_["address"], "total_size": s_["total_size"], "stream": s_["stream"], "segment_type": s_["segment_type"],
                 "blocks": [(b_["address"] if "address" in b_ else None, b_["size"], b_["state"]) for b_ in s_["blocks"]]} for s_ in torch.cuda.memory_snapshot()]
        json.dump(snap, open(os.environ["TD_BENCH_MEMMAP"], "w"))
        open(os.environ["TD_BENCH_MEMMAP"] + ".maps", "w").write(open("/proc/self/maps").read())  # every mapping of the process (host, pinned, device apertures)
        from tubedetr_amd.ops import job_tables as jt_
        json.dump({"job_tables_host": [(s_[0].data_ptr(), s_[0].numel()) for s_ in jt_.slots + jt_.retired],
                   "job_tables_dev": [(s_[1].data_ptr(), s_[1].numel()) for s_ in jt_.slots + jt_.retired]}, open(os.environ["TD_BENCH_MEMMAP"] + ".ptrs", "w"))
    host_idle_ms = None
    for i in range(a.warmup):
        last_warm = i == a.warmup - 1 and i > 0
        if last_warm:
            torch.cuda.synchronize()  # the last warm-up step is enqueued on an IDLE device: how long the host needs by itself
            th0 = time.perf_counter()
        step(i)
        if last_warm:
            host_idle_ms = (time.perf_counter() - th0) * 1e3
        if os.environ.get("TD_BENCH_TRACE"):
            torch.cuda.synchronize()
            _trace(f"warm-up step {i} done")
    fence()
    _trace("timed region starts")
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 1)]  # one event per step boundary (no sync): the spread of the K steps
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(a.steps):
        loss = step(a.warmup + i)
        marks[i + 1].record()
    host_elapsed = time.perf_counter() - t0  # host-side enqueue time of the K steps (before waiting for the GPU)
    fence()
    elapsed = time.perf_counter() - t0
    step_ms = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(a.steps))
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev if a.backend == "nccl" else "cpu")
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        elapsed = tt.item()
    if a.dump_grads and rank == 0 and reducer is not None:
        torch.cuda.synchronize()
        last = (a.warmup + a.steps - 1) % len(batches)
        torch.save({"flat": reducer.flat.detach().cpu(), "names": [n_ for n_, p_ in model.named_parameters() if p_.requires_grad],
                    "numels": [p_.numel() for p_ in reducer.params], "batch_seeds": [1000 * r_ + last for r_ in range(world)],
                    "used": reducer._global_used, "world": world}, a.dump_grads)
    assert math.isfinite(loss.item()), "non-finite loss"

    # ---- the same workload with the dead work skipped (`value_dedupe`): the model proves from its inputs that the slow clip is every k-th frame of
    # the fast frames' own buffer and does not push those pixels through the trunk twice (100 / 125 of the reference's trunk-forward FLOPs; SURVEY
    # 8a' "dead work the build may skip").  Timed here, in the same run, as a short EAGER loop (the step is GPU-bound eagerly as well at this batch):
    # the headline `value` above stays the reference's full work.
    dedupe_rec = None
    if world == 1 and not a.dedupe and a.dedupe_steps > 0 and not a.no_fast:
        ops_.set_dropout_counter(None)
        model.slow_frames_are_strided_fast = None
        try:
            for i in range(3):  # (the mode allocates differently shaped trunk workspaces: let the allocator settle)
                eager_step(a.warmup + a.steps + i)
            torch.cuda.synchronize()
            dm = [torch.cuda.Event(enable_timing=True) for _ in range(a.dedupe_steps + 1)]
            dm[0].record()
            for i in range(a.dedupe_steps):
                eager_step(a.warmup + a.steps + 3 + i)
                dm[i + 1].record()
            torch.cuda.synchronize()
            d_all = sorted(dm[i].elapsed_time(dm[i + 1]) for i in range(a.dedupe_steps))
            d_ms = d_all[len(d_all) // 2]  # median step: one allocator stall must not stand for the mode
            dedupe_rec = {"value_dedupe": round(B * 1e3 / d_ms, 3), "ms_per_step_dedupe": round(d_ms, 2), "ms_per_step_dedupe_min_max": [round(d_all[0], 2), round(d_all[-1], 2)], "steps": a.dedupe_steps, "execution": "eager (median step)",
                          "trunk_forward_frames_per_clip": {"reference": T + math.ceil(T / k), "executed": T},
                          "note": "slow frames proven to be fast[::k] of the same buffer (FrameSources aliasing + host index list) and not recomputed; "
                                  "not the headline: `value` executes every frame of the reference algorithm"}
        finally:
            model.slow_frames_are_strided_fast = False
    roofline, cpu = None, None
    if a.roofline_steps > 0 and rank != 0 and distributed:
        # N > 1: the eager step holds collectives (num_boxes, the gradient exchange), so every rank runs the roofline steps that
        # rank 0 times - a rank that skipped them would leave rank 0's collectives unmatched
        ops_.set_dropout_counter(None)
        for i in range(a.roofline_steps):
            eager_step(a.warmup + a.steps + i)
        torch.cuda.synchronize()
    if rank == 0:
        if a.roofline_steps > 0:
            L_ = _hip.lib()
            ops_.set_dropout_counter(None)
            L_.td_prof_enable(1)
            for i in range(a.roofline_steps):
                eager_step(a.warmup + a.steps + i)  # event-timed launches are issued eagerly (not from the graph)
            torch.cuda.synchronize()
            code = _hip.TD_BF16 if cdt == torch.bfloat16 else _hip.TD_F32
            peak = PEAK_BF16_TFLOPS if cdt == torch.bfloat16 else 157.3
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            from kernel_families import FAMILY_OF_PROF_ID, prof_key

            fams = {fam: prof_key(fam, fp32=cdt != torch.bfloat16) for fam in FAMILY_OF_PROF_ID}
            # family keys = the name stems of the kernels that form them (tools/kernel_families.py, shared with the PMC aggregation): 0 / 1 / 3 the
            # 128x128 / 128x64 / 64x128 tile instances (all pipeline depths, pointwise / tap-uniform / two-source); 2 wide-tile + 128x128 batched weight
            # gradients; 4 the persistent 1x1; 5 / 6 the 256-row tile kernels on spatial (MFMA-bound) and on pointwise K >= 512 layers; 7 the LDS-resident
            # fused stem / layer1 blocks; 8 the decoder's time-aligned cross-attention core (bf16: MFMA tiles, a few MFMAs per kilobyte of memory rows -
            # bound = HBM); 9 the per-layer weight gradients.  `kernel` in the JSON = the exact kernel names of the family in the committed PMC pass.
            pmc, mfma = _load_profile_json(PMC_TRAFFIC), _load_profile_json(PMC_MFMA)
            # an (event, event) pair around nothing: what the bracketing itself adds to every launch (reported, NOT subtracted)
            cal = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(200)]
            for e0_, e1_ in cal:
                e0_.record()
                e1_.record()
            torch.cuda.synchronize()
            ev_over_ms = sorted(e0_.elapsed_time(e1_) for e0_, e1_ in cal)[len(cal) // 2]
            per = []
            for fam, kname in fams.items():
                n, ms, fl, by = C.c_longlong(), C.c_double(), C.c_double(), C.c_double()
                _hip.check(L_.td_prof_collect(fam, code, C.byref(n), C.byref(ms), C.byref(fl)), "td_prof_collect")
                _hip.check(L_.td_prof_collect_bytes(fam, code, C.byref(by)), "td_prof_collect_bytes")
                if n.value:
                    tfl = fl.value / (ms.value * 1e-3) / 1e12
                    gbs = by.value / (ms.value * 1e-3) / 1e9
                    hbm_bound = gbs / PEAK_HBM_GBS > tfl / peak  # the roof this family sits closer to
                    t_ = pmc.get(kname)
                    traffic = (t_["fetch_bytes_per_launch"] + t_["write_bytes_per_launch"]) if (t_ and t_.get("fetch_bytes_per_launch") and t_.get("write_bytes_per_launch")) else None
                    traffic_why = None if traffic else "no committed PMC pass names this kernel family"
                    if traffic and pmc.get("_steps"):
                        # The committed counters describe THIS run only if the family is launched as often per step here as in the PMC pass
                        # (a kernel change that re-routes layers without re-taking profiles/ would otherwise report stale bytes).  The
                        # batched weight gradients are exempt from equality - an accumulating phase is a launch of its own - and are
                        # re-scaled to this run's launch count instead.
                        pmc_per_step, here_per_step = t_["dispatches"] / pmc["_steps"], n.value / a.roofline_steps
                        if fam in (2, 9) or abs(pmc_per_step - here_per_step) < 0.5:
                            traffic = round(traffic * pmc_per_step / here_per_step)
                        else:
                            traffic, traffic_why = None, (f"stale: profiles/{PMC_TRAFFIC} saw {pmc_per_step:g} launches of this family per step, this run {here_per_step:g} "
                                                          f"- re-take the PMC passes (tools/final_profile.sh)")
                    mu = mfma.get(kname, {}).get("mfma_util") if isinstance(mfma.get(kname), dict) else None
                    if mu is not None and traffic is None and traffic_why and traffic_why.startswith("stale"):
                        mu = None  # the same passes: stale with the traffic figure
                    exact = (t_ or {}).get("kernels") or (mfma.get(kname, {}).get("kernels") if isinstance(mfma.get(kname), dict) else None)
                    rec = {"bound": "hbm" if hbm_bound else "mfma", "kernel": " + ".join(exact) if exact else kname, "family": kname,
                           "achieved": round(gbs if hbm_bound else tfl, 2), "peak": PEAK_HBM_GBS if hbm_bound else peak,
                           "unit": "GB/s" if hbm_bound else "TFLOP/s", "frac": round((gbs / PEAK_HBM_GBS) if hbm_bound else (tfl / peak), 4),
                           "traffic": traffic, "traffic_null_reason": traffic_why,
                           "traffic_source": (f"STATIC, not measured by this run: HBM bytes per launch from the committed rocprofv3 PMC passes of the same command "
                                              f"(FETCH_SIZE x2 + WRITE_SIZE, profiles/{PMC_TRAFFIC})") if traffic else None,
                           "mfma_util": mu, "mfma_util_source": (f"STATIC: SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs) over the family's launches, profiles/{PMC_MFMA}") if mu is not None else None,
                           "algorithmic_bytes_per_launch": round(by.value / n.value), "achieved_gbs": round(gbs, 1), "achieved_tflops": round(tfl, 2),
                           "launches_per_step": n.value // a.roofline_steps, "event_pair_overhead_us": round(ev_over_ms * 1e3, 2),
                           "avg_launch_us": round(ms.value * 1e3 / n.value, 2), "kernel_ms_per_step": round(ms.value / a.roofline_steps, 3),
                           "algorithmic_gflop_per_step": round(fl.value / a.roofline_steps / 1e9, 1)}
                    per.append(rec)
            L_.td_prof_enable(0)
            if per:
                per.sort(key=lambda r: -r["kernel_ms_per_step"])
                roofline = dict(per[0])  # the dominant kernel (largest share of the step)
                roofline["other_mfma_kernels"] = per[1:]
                # north_star's decoder-attention group from per-dispatch PMC rows (profiles/README.md), the coarse family view as fallback
                grp = (mfma.get("decoder_attention_group_precise") or mfma.get("decoder_attention_group")) if isinstance(mfma, dict) else None
                if grp:
                    roofline["decoder_attention_group"] = grp  # north_star's sub-target, static from profiles/
        if world == 1 and a.cpu_frames > 0:
            try:
                cpu = cpu_baseline(max(a.cpu_frames, k), res, k, L, T)
            except Exception as e:  # the baseline must never take the GPU number down with it
                cpu = {"value": None, "unit": "clips/s", "cores": os.cpu_count(), "kind": "port", "sample": f"failed: {e!r}"}

    if os.environ.get("TD_EFENCE") == "1":
        print(f"[bench] electric fence: every allocation fenced = {efence_install.protected()}", file=sys.stderr, flush=True)
    if rank == 0:
        clips = world * a.steps * B
        value = clips / elapsed
        step_tflop = ALGO_TFLOP_PER_CLIP.get(a.workload)
        out = {
            "metric": ("training clips/sec (fwd+bwd) at T=100 k=4 res=352, 1/2/4/8 MI355X" if a.workload == "cfg3" else "training clips/sec (fwd+bwd)"), "value": round(value, 3), "unit": "clips/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": round(elapsed / a.steps * 1e3, 2),
            "ms_per_step_min": round(step_ms[0], 2), "ms_per_step_median": round(step_ms[len(step_ms) // 2], 2), "ms_per_step_max": round(step_ms[-1], 2),
            "clips_per_step_per_gpu": B, "host_enqueue_ms_per_step": round(host_elapsed / a.steps * 1e3, 2),
            "host_enqueue_ms_idle_device": None if host_idle_ms is None else round(host_idle_ms, 2),
            "host_enqueue_note": "host_enqueue_ms_per_step is measured with the K steps enqueued back to back (the launch call of a replay returns only when the device "
                                 "queue has room for its ~1 500 packets, i.e. about one step behind the device); host_enqueue_ms_idle_device is one step enqueued on an idle device",
            "peak_hbm_reserved_gb": round(torch.cuda.max_memory_reserved(dev) / 2**30, 1), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
            "execution": execution,
            "gradient_exchange": (None if not distributed else ("torch DDP (find_unused_parameters)" if a.ddp else
                                  (f"staged flat {reducer.collective} overlapped with the trunk backward, {a.grad_wire_dtype} on the wire" if staged else f"flat {reducer.collective} after backward, {a.grad_wire_dtype} on the wire"))),
            "process_group": (None if not distributed else {"backend": a.backend, "ranks": world, "devices_visible": n_dev,
                                                            "oversubscribed": bool(a.oversubscribe and world > n_dev),
                                                            "note": ("REHEARSAL of the N>1 control flow: several ranks share one device and the collectives go through the host - "
                                                                     "not a scaling measurement") if (a.oversubscribe and world > n_dev) or a.backend != "nccl" else None}),
            "config": {"workload": f"{a.workload}: T={T} k={k} res={res} L={L}, {B} clip(s)/GPU/step, fast={not a.no_fast}, tsa={not a.no_tsa}, train-mode dropout={not a.eval_dropout_off}, "
                                   f"frames={'uint8 pixels, normalised on the device' if a.frames == 'u8' else 'host-normalised fp32'}",
                       "global_batch": world * B, "parallelism": f"dp{world}", "weights": "random init (reference scheme), seed 42+rank"},
            "flops_note": ("slow frames not recomputed in the fast pass (identical pixels): executed trunk-forward work is 100/125 of the "
                           "reference algorithm's; roofline fractions use executed FLOPs, step_frac_of_mfma_peak the reference algorithm's 6.847 TFLOP") if (a.dedupe and not a.no_fast) else None,
            "step_frac_of_mfma_peak": round(step_tflop * value / world / PEAK_BF16_TFLOPS, 4) if (step_tflop and a.dtype == "bf16") else None,
            "dedupe": dedupe_rec, "value_dedupe": None if dedupe_rec is None else dedupe_rec["value_dedupe"],
            "roofline": roofline, "cpu_baseline": cpu,
        }
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if world > 1 or a.force_ddp:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
