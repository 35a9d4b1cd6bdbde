"""Host-side collectives of the evaluation path (util/dist.py:34-128 of the reference): gathering picklable per-rank
results and averaging the loss dict for logging.  Pure torch.distributed (RCCL on the GPUs, gloo in the CPU tests); the
gradient exchange of the training step lives in tubedetr_amd/distributed.py."""
from __future__ import annotations

import io
import pickle
from typing import Any, Dict, List

import torch
import torch.distributed as dist


def get_world_size() -> int:
    return dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1


def all_gather(data: Any, group=None) -> List[Any]:
    """Gather an arbitrary picklable object from every rank (util/dist.py:34-95): sizes first, then the padded byte
    buffers.  One all_gather_object-free implementation so it works on RCCL (device buffers) and gloo alike."""
    world = get_world_size()
    if world == 1:
        return [data]
    backend = dist.get_backend(group)
    dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    payload = pickle.dumps(data)
    local = torch.frombuffer(bytearray(payload), dtype=torch.uint8).to(dev)
    sizes = [torch.zeros(1, dtype=torch.long, device=dev) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([local.numel()], dtype=torch.long, device=dev), group=group)
    sizes = [int(s.item()) for s in sizes]
    mx = max(sizes)
    if local.numel() != mx:
        local = torch.cat([local, torch.zeros(mx - local.numel(), dtype=torch.uint8, device=dev)])
    bufs = [torch.empty(mx, dtype=torch.uint8, device=dev) for _ in range(world)]
    dist.all_gather(bufs, local, group=group)
    return [pickle.load(io.BytesIO(b[:n].cpu().numpy().tobytes())) for b, n in zip(bufs, sizes)]


def reduce_dict(input_dict: Dict[str, torch.Tensor], average: bool = True) -> Dict[str, torch.Tensor]:
    """All-reduce the values of a dict of scalar tensors in ONE collective (util/dist.py:98-128), sorted by key so every
    rank stacks them in the same order."""
    world = get_world_size()
    if world < 2:
        return input_dict
    with torch.no_grad():
        names = sorted(input_dict)
        vals = torch.stack([input_dict[k].detach().float() for k in names])
        dist.all_reduce(vals)
        if average:
            vals = vals / world
        return {k: v for k, v in zip(names, vals)}
