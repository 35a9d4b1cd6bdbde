#!/usr/bin/env python3
"""bench.py - training clips/sec (forward + backward) of the TubeDETR hot path on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N>1 it is launched under
torch.distributed.run with one rank per GPU (RCCL).  A "step" = one pass of the hot path over one synthetic clip
per GPU: the two model calls of engine.py:67-80 (video-text encoder, then space-time decoder), the criterion, and
the backward pass; at N>1 followed by ONE flat all-reduce of the gradients (tubedetr_amd/distributed.py; `--ddp`
uses torch DistributedDataParallel like main.py:372-376 instead).  Rank 0 prints ONE JSON line.

Workload at N=1: BASELINE.json configs[2] (the config the metric is quoted on): T=100 frames, stride k=4,
res=352, L=30 text tokens, 1 clip per GPU, bf16 MFMA kernels with fp32 accumulation, random-init weights,
train mode (dropout active), weights re-prepared every step (as after an optimizer step), all 125 trunk-forward
frames executed (`--dedupe` skips the 25 slow frames inside the fast pass).  Inputs are generated on the device before
the timed region.

Execution: the whole step is captured once in a single-stream HIP graph and replayed (`--no-graph`: eager launches,
host-bound).  At N=1 the measurement runs in a child process; if that process dies (a GPU memory fault was seen
intermittently with the forked two-stream graph, `--text-stream`), the parent re-measures with eager launches, so a
bench line is always produced; `attempts` in the JSON records what happened.

Extra legs (rank 0, after the timed region, not part of `value`):
  roofline     : `--roofline-steps` more identical steps, launched eagerly, with HIP events recorded on the launch
                 stream around every launch of the MFMA kernel families; per family achieved = algorithmic FLOPs or
                 algorithmic HBM bytes / summed duration, bound = the roof it sits closer to, traffic = PMC-measured
                 HBM bytes per launch (profiles/r01_pmc_traffic.json).  `roofline` is the family with the largest
                 share of the step, the others follow in `other_mfma_kernels`.
  cpu_baseline : the CPU oracle (oracle/, a port of the reference algorithm) timed on the host cores on a bounded
                 sample (a T=`--cpu-frames` clip of the same resolution, fwd+bwd) and scaled to T=100.
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
# algorithmic GFLOP per clip fwd+bwd
ALGO_TFLOP_PER_CLIP = {"cfg3": 6.847, "cfg2": 2.538, "cfg1": 0.233}


def make_batch(T, res, k, L, seed, device, clips=1):
    """SURVEY.md 8d synthetic clips, generated directly in HBM: video ~ N(0,1), slow = video[::k], fast = all frames; `clips`
    videos of equal duration per batch (video-major frame order, like util/misc.py's collate)."""
    g = torch.Generator(device=device).manual_seed(seed)
    video = torch.randn(clips * T, 3, res, res, generator=g, device=device)
    ids = torch.randint(3, 50000, (clips, L), generator=torch.Generator().manual_seed(seed))  # token ids start on the host, like a tokenizer's output
    ids[:, 0], ids[:, -1] = 0, 2
    cxcy = torch.rand(clips * T, 2, generator=g, device=device) * 0.6 + 0.2
    wh = torch.rand(clips * T, 2, generator=g, device=device) * 0.3 + 0.1
    n_slow = math.ceil(T / k)
    slow = video.view(clips, T, 3, res, res)[:, ::k].reshape(clips * n_slow, 3, res, res).contiguous()
    return {
        "frames": slow,
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


def cpu_baseline(T_sample, res, k, L, T_full):
    """Reference algorithm on the host cores (oracle port), fwd+bwd of a T_sample-frame clip, scaled to T_full frames."""
    from oracle.tubedetr_oracle import OracleConfig, train_step
    from oracle.weights import fill_state, state_spec, synthetic_batch

    cores = min(os.cpu_count() or 1, 32)  # more threads than this only slows the small-batch CPU convolutions down
    torch.set_num_threads(cores)
    cfg = OracleConfig(stride=k)
    sd = fill_state(state_spec(cfg), 1, requires_grad=True)
    batch = synthetic_batch(T=T_sample, res=res, k=k, L=L, seed=5)
    best, spent = None, 0.0
    for _ in range(2):  # second pass = steady state; skipped when the first one already used the time budget
        for v in sd.values():
            v.grad = None
        t0 = time.time()
        loss, _, _, _ = train_step(sd, cfg, batch)
        loss.backward()
        dt = time.time() - t0
        best = dt if best is None else min(best, dt)
        spent += dt
        if spent > 20.0:
            break
    per_clip = best * (T_full / T_sample)
    return {"value": 1.0 / per_clip, "unit": "clips/s", "cores": cores, "kind": "port",
            "sample": f"fwd+bwd of a T={T_sample} clip (k={k}, res={res}, L={L}) by the CPU oracle in {best:.1f}s, scaled x{T_full}/{T_sample} to T={T_full}"}


def _trace(msg):
    if os.environ.get("TD_BENCH_TRACE"):
        print(f"[bench] {msg}", file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="cfg3", choices=list(WORKLOADS))
    ap.add_argument("--clips-per-gpu", type=int, default=1, help="videos per GPU per step (the reference's --batch_size, main.py:63)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--roofline-steps", type=int, default=2)
    ap.add_argument("--cpu-frames", type=int, default=48, help="frames of the CPU-baseline sample clip (0 = skip)")
    ap.add_argument("--keep-prepared-weights", action="store_true", help="diagnostic: reuse prepared bf16 weights across steps")
    ap.add_argument("--dedupe", action="store_true",
                    help="do not recompute the slow frames inside the fast pass (exact, slow = video[::k]); off by default so the timed step "
                         "executes the same work as the reference's")
    ap.add_argument("--no-dedupe", action="store_true", help="(default behaviour; kept for compatibility)")
    ap.add_argument("--graph", dest="graph", action="store_true", default=True, help="capture the step in a HIP graph (N=1 only)")
    ap.add_argument("--no-graph", dest="graph", action="store_false")
    ap.add_argument("--force-ddp", action="store_true", help="diagnostic: run the N>1 code path (process group + gradient exchange) with one rank")
    ap.add_argument("--ddp", action="store_true", help="N>1: use torch DistributedDataParallel like main.py:372-376 instead of the flat all-reduce")
    ap.add_argument("--grad-wire-dtype", default="fp32", choices=["fp32", "bf16"], help="dtype of the gradient all-reduce on the wire")
    ap.add_argument("--no-fast", action="store_true")
    ap.add_argument("--no-tsa", action="store_true")
    ap.add_argument("--eval-dropout-off", action="store_true", help="diagnostic only: run in eval mode")
    ap.add_argument("--child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--text-stream", action="store_true",
                    help="graph mode: keep RoBERTa on its own stream (a forked graph branch: ~0.5 ms less GPU time per step but a 24 ms "
                         "hipGraphLaunch, and the only configuration in which a replay ever hit a GPU memory fault)")
    a = ap.parse_args()

    if os.environ.get("TD_EFENCE") == "1":  # diagnostic: electric-fence device allocator (tests/efence/), eager launches only
        sys.path.insert(0, os.path.join(ROOT, "tests", "efence"))
        import install as efence_install

        efence_install.install()
        a.graph, a.child = False, True
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1 and not a.child and not a.force_ddp and a.graph and os.environ.get("TD_BENCH_ISOLATE", "1") != "0":
        # Single-GPU run: the measurement happens in a child process.  A GPU memory fault during a graph replay (seen
        # intermittently on fresh boxes with the forked two-stream graph) kills the process that owns the HIP context;
        # the parent then re-measures with eager launches, which never faulted, instead of losing the bench line.
        import subprocess

        argv = [sys.executable, os.path.abspath(__file__)] + sys.argv[1:] + ["--child"]
        attempts = []
        for extra in ([], [], ["--no-graph"]):
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
    if a.graph and not a.text_stream:
        os.environ.setdefault("TD_TEXT_STREAM", "0")  # single-stream capture: a linear graph launches in ~5 ms of host time
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {a.gpus}"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1 or a.force_ddp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29512")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        torch.distributed.init_process_group("nccl", device_id=dev)

    import tubedetr_amd
    from tubedetr_amd import _hip
    from tubedetr_amd import ops as ops_
    from tubedetr_amd.harness import forward_step
    from tubedetr_amd.models import build_model

    T, res, k, L = WORKLOADS[a.workload]
    cdt = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    torch.manual_seed(42 + rank)  # main.py:358
    args = tubedetr_amd.default_args(stride=k, fast=not a.no_fast, no_tsa=a.no_tsa, compute_dtype=cdt, video_max_len_train=max(200, T))
    model, criterion, weight_dict = build_model(args)
    model.to(dev)
    model.slow_frames_are_strided_fast = bool(a.dedupe and not a.no_dedupe)  # legal because the synthetic clip has slow = video[::k]
    model.train(not a.eval_dropout_off)
    tok = BatchTokenizer()
    model.transformer.tokenizer = tok
    net = model
    distributed = world > 1 or a.force_ddp
    reducer = None
    if distributed and a.ddp:
        net = torch.nn.parallel.DistributedDataParallel(model, device_ids=[local_rank], find_unused_parameters=True)  # main.py:372-376
    elif distributed:
        # replicas start from rank 0's weights (what DDP's constructor does), then exchange gradients with ONE flat
        # all-reduce per step (tubedetr_amd/distributed.py); the step itself holds no collective
        from tubedetr_amd.distributed import FlatGradAllReducer, sync_num_boxes

        for t_ in list(model.parameters()) + list(model.buffers()):
            torch.distributed.broadcast(t_.data, 0)
        reducer = FlatGradAllReducer(model.parameters(), torch.bfloat16 if a.grad_wire_dtype == "bf16" else torch.float32)
        criterion.external_num_boxes = torch.ones(1, dtype=torch.float32, device=dev)
        reducer.always_communicate = a.force_ddp  # exercise the RCCL call in the 1-rank diagnostic

    n_batches = a.warmup + a.steps + a.roofline_steps
    batches = [make_batch(T, res, k, L, 1000 * rank + s, dev, a.clips_per_gpu) for s in range(min(n_batches, 4))]

    from tubedetr_amd.functional import invalidate_prepared

    params = [p_ for p_ in model.parameters() if p_.requires_grad]

    def step(i):
        return eager_step(i)

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
        loss.backward()
        if reducer is not None:
            reducer.reduce(attach=True)  # gather, ONE all-reduce, .grad re-pointed at the flat buffer (no copy back)
        return loss

    def fence():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    # ---- optional whole-step HIP graph (single GPU): the ~1500 launches of a step are captured once and replayed, so
    # the host only copies the next clip into the static input buffers and bumps the dropout step counter ----
    execution = "eager"
    if a.graph and not (distributed and a.ddp):
        try:
            static = {k_: (v.clone() if torch.is_tensor(v) else v) for k_, v in batches[0].items()}
            for k_ in ("input_ids", "attention_mask"):
                static[k_] = static[k_].to(dev)
            counter = torch.zeros(1, dtype=torch.int32, device=dev)
            ops_.set_dropout_counter(counter)

            def body():
                tok.batch = static
                invalidate_prepared()
                l_, _, _, _ = forward_step(net, criterion, weight_dict, static)
                l_.backward()
                if reducer is not None:
                    reducer.gather()  # no collective: part of the captured step
                return l_

            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(3):
                    for p_ in params:
                        p_.grad = None
                    body()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            for p_ in params:
                p_.grad = None
            _trace("eager warm-up on the capture side stream done")
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                static_loss = body()
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
                graph.replay()  # ends with the gather of the gradients into the flat exchange buffer
                if reducer is not None:
                    reducer.all_reduce()  # the only collective of the step, outside the graph
                return static_loss

            execution = "hip_graph"
        except Exception as exc:  # capture not possible: measure the eager path
            ops_.set_dropout_counter(None)
            torch.cuda.synchronize()
            execution = f"eager (graph capture failed: {type(exc).__name__}: {str(exc)[:120]})"

    if os.environ.get("TD_BENCH_MEMMAP"):  # fault triage: where every allocator segment / block lives before the replays start
        torch.cuda.synchronize()
        snap = [{"address": s_["address"], "total_size": s_["total_size"], "stream": s_["stream"], "segment_type": s_["segment_type"],
                 "blocks": [(b_["address"] if "address" in b_ else None, b_["size"], b_["state"]) for b_ in s_["blocks"]]} for s_ in torch.cuda.memory_snapshot()]
        json.dump(snap, open(os.environ["TD_BENCH_MEMMAP"], "w"))
    for i in range(a.warmup):
        step(i)
        if os.environ.get("TD_BENCH_TRACE"):
            torch.cuda.synchronize()
            _trace(f"warm-up step {i} done")
    fence()
    _trace("timed region starts")
    t0 = time.perf_counter()
    for i in range(a.steps):
        loss = step(a.warmup + i)
    host_elapsed = time.perf_counter() - t0  # host-side enqueue time of the K steps (before waiting for the GPU)
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        elapsed = tt.item()
    assert math.isfinite(loss.item()), "non-finite loss"

    roofline, cpu = None, None
    if rank == 0:
        if a.roofline_steps > 0:
            L_ = _hip.lib()
            ops_.set_dropout_counter(None)
            L_.td_prof_enable(1)
            for i in range(a.roofline_steps):
                eager_step(a.warmup + a.steps + i)  # event-timed launches are issued eagerly (not from the graph)
            torch.cuda.synchronize()
            code = _hip.TD_BF16 if cdt == torch.bfloat16 else _hip.TD_F32
            tname = "unsigned short" if cdt == torch.bfloat16 else "float"
            peak = PEAK_BF16_TFLOPS if cdt == torch.bfloat16 else 157.3
            fams = {0: f"td::conv_gemm_kernel<{tname}, 128, 128, *, *>", 3: f"td::conv_gemm_kernel<{tname}, 64, 128, *, *>",
                    1: f"td::conv_gemm_kernel<{tname}, 128, 64, *, *>", 2: f"td::conv_wgrad_kernel|conv_wgrad_batch_kernel<{tname}>", 4: "td::pw_resident_kernel<*>"}  # * = all pipeline depths, pointwise and generic instances
            # PMC-measured HBM traffic per launch of the same command (tools/pmc_traffic.py, committed under profiles/):
            # counters cannot be read from inside the process being timed
            pmc = {}
            try:
                pmc = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_pmc_traffic.json")))
            except Exception:
                pass
            # an (event, event) pair around nothing: what the bracketing itself adds to every launch; subtracted below so
            # that the averages can be compared with rocprofv3's kernel durations (profiles/)
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
                    ms.value = max(ms.value - n.value * ev_over_ms, 0.5 * ms.value)
                    tfl = fl.value / (ms.value * 1e-3) / 1e12
                    gbs = by.value / (ms.value * 1e-3) / 1e9
                    hbm_bound = gbs / PEAK_HBM_GBS > tfl / peak  # the roof this family sits closer to
                    t_ = pmc.get(kname if fam != 2 else "td::conv_wgrad_batch_kernel<%s>" % tname)
                    traffic = (t_["fetch_bytes_per_launch"] + t_["write_bytes_per_launch"]) if (t_ and t_.get("fetch_bytes_per_launch") and t_.get("write_bytes_per_launch")) else None
                    rec = {"bound": "hbm" if hbm_bound else "mfma", "kernel": kname,
                           "achieved": round(gbs if hbm_bound else tfl, 2), "peak": PEAK_HBM_GBS if hbm_bound else peak,
                           "unit": "GB/s" if hbm_bound else "TFLOP/s", "frac": round((gbs / PEAK_HBM_GBS) if hbm_bound else (tfl / peak), 4),
                           "traffic": traffic if fam != 2 else None,
                           "traffic_note": ("HBM bytes per launch, rocprofv3 PMC FETCH_SIZE x2 + WRITE_SIZE averaged over the family's launches "
                                            "(profiles/r01_pmc_traffic.json)") if (traffic and fam != 2) else None,
                           "algorithmic_bytes_per_launch": round(by.value / n.value), "achieved_gbs": round(gbs, 1), "achieved_tflops": round(tfl, 2),
                           "launches_per_step": n.value // a.roofline_steps, "event_pair_overhead_us_subtracted": round(ev_over_ms * 1e3, 2),
                           "avg_launch_us": round(ms.value * 1e3 / n.value, 2), "kernel_ms_per_step": round(ms.value / a.roofline_steps, 3),
                           "algorithmic_gflop_per_step": round(fl.value / a.roofline_steps / 1e9, 1)}
                    per.append(rec)
            L_.td_prof_enable(0)
            if per:
                per.sort(key=lambda r: -r["kernel_ms_per_step"])
                roofline = dict(per[0])  # the dominant kernel (largest share of the step)
                roofline["other_mfma_kernels"] = per[1:]
        if world == 1 and a.cpu_frames > 0:
            try:
                cpu = cpu_baseline(max(a.cpu_frames, k), res, k, L, T)
            except Exception as e:  # the baseline must never take the GPU number down with it
                cpu = {"value": None, "unit": "clips/s", "cores": os.cpu_count(), "kind": "port", "sample": f"failed: {e!r}"}

    if os.environ.get("TD_EFENCE") == "1":
        print(f"[bench] electric fence: every allocation fenced = {efence_install.protected()}", file=sys.stderr, flush=True)
    if rank == 0:
        clips = world * a.steps * a.clips_per_gpu
        value = clips / elapsed
        step_tflop = ALGO_TFLOP_PER_CLIP.get(a.workload)
        out = {
            "metric": ("training clips/sec (fwd+bwd) at T=100 k=4 res=352, 1/2/4/8 MI355X" if a.workload == "cfg3" else "training clips/sec (fwd+bwd)"), "value": round(value, 3), "unit": "clips/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": round(elapsed / a.steps * 1e3, 2), "host_enqueue_ms_per_step": round(host_elapsed / a.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
            "execution": execution, "gradient_exchange": (None if not distributed else ("torch DDP (find_unused_parameters)" if a.ddp else f"flat all-reduce, {a.grad_wire_dtype} on the wire")),
            "config": {"workload": f"{a.workload}: T={T} k={k} res={res} L={L}, {a.clips_per_gpu} clip(s)/GPU/step, fast={not a.no_fast}, tsa={not a.no_tsa}, train-mode dropout={not a.eval_dropout_off}",
                       "global_batch": world * a.clips_per_gpu, "parallelism": f"dp{world}", "weights": "random init (reference scheme), seed 42+rank"},
            "flops_note": ("slow frames not recomputed in the fast pass (identical pixels): executed trunk-forward work is 100/125 of the "
                           "reference algorithm's; roofline fractions use executed FLOPs, step_frac_of_mfma_peak the reference algorithm's 6.847 TFLOP") if (model.slow_frames_are_strided_fast and not a.no_fast) else None,
            "step_frac_of_mfma_peak": round(step_tflop * value / world / PEAK_BF16_TFLOPS, 4) if (step_tflop and a.dtype == "bf16") else None,
            "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(out), flush=True)
    if world > 1 or a.force_ddp:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
