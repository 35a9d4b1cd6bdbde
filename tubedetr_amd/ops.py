"""Tensor-level wrappers over the C ABI (no autograd here; see functional.py).

Every function launches hand-written gfx950 kernels on the current HIP stream.  Activations are
NHWC / row-major in the compute dtype (torch.float32 = exact-fp32 parity mode, torch.bfloat16 =
throughput mode).  There is no CPU path: CPU tensors raise.
"""
from __future__ import annotations

import os

import ctypes as C
from typing import Optional, Tuple

import torch

from . import _hip
from ._hip import ConvDesc, Epilogue, check, dtype_code, ptr, stream_ptr

Tensor = torch.Tensor


class _ZeroArena:
    """Pre-zeroed fp32 arena for the accumulate-into outputs of the backward kernels (weight / bias / LayerNorm
    gradients are produced with atomics): one fill per chunk instead of one torch.zeros launch per gradient
    (~200 per step).  Slices are handed out once and never reused; a chunk lives as long as any slice of it does."""

    def __init__(self, chunk: int):
        self.chunk = chunk  # floats
        self.buf = None
        self.off = 0

    def take(self, shape, device) -> torch.Tensor:
        n = 1
        for d_ in shape:
            n *= int(d_)
        n_al = (n + 63) // 64 * 64  # keep 256-byte alignment of every slice
        if self.buf is None or self.buf.device != device or self.off + n_al > self.buf.numel():
            self.buf = torch.zeros(max(self.chunk, n_al), dtype=torch.float32, device=device)
            self.off = 0
        out = self.buf[self.off : self.off + n].view(shape)
        self.off += n_al
        return out


_ARENAS: dict = {}  # one arena per launch stream: a chunk returns to the allocator pool of the stream that allocated it
_CAPTURE_ARENAS: dict = {}  # stream -> arena of the stream capture in progress (its chunks belong to that graph's private pool)


def reset_capture_arena() -> None:
    """Call before starting a new stream capture that directly follows another one (two graphs captured back to back): slices of
    a chunk whose fill node belongs to the previous graph must not be handed to the next.  (A capture that follows eager
    launches needs nothing: the first eager zeros_f32 call drops the arenas.)"""
    _CAPTURE_ARENAS.clear()


_USE_ARENA = __import__("os").environ.get("TD_ZERO_ARENA", "1") != "0"


def zeros_f32(shape, device) -> torch.Tensor:
    if not _USE_ARENA:
        return torch.zeros(shape, dtype=torch.float32, device=device)
    shape = tuple(shape) if not isinstance(shape, int) else (shape,)
    if device.type == "cuda" and torch.cuda.is_current_stream_capturing():
        # inside a HIP-graph capture: an arena of its own, used for this capture only (sharing one arena between a capture's
        # private pool and eager steps faulted on ROCm 7.0) - every chunk is ONE captured fill node, re-zeroed at every replay,
        # instead of one fill node per LayerNorm / bias gradient (~75 per step)
        # (one arena per capturing stream: a chunk's fill node is ordered only with the work of the stream it was captured on)
        sp = stream_ptr()
        if sp not in _CAPTURE_ARENAS:
            _CAPTURE_ARENAS[sp] = _ZeroArena(256 * 1024)
        return _CAPTURE_ARENAS[sp].take(shape, device)
    if _CAPTURE_ARENAS:
        _CAPTURE_ARENAS.clear()  # a capture has ended: its arenas are never touched again
    key = (str(device), stream_ptr() if device.type == "cuda" else 0)
    arena = _ARENAS.get(key)
    if arena is None:
        arena = _ARENAS[key] = _ZeroArena(32 * 1024 * 1024)
    return arena.take(shape, device)


def vec_of(dt: torch.dtype) -> int:
    return 8 if dt == torch.bfloat16 else 4


def pad_to(n: int, m: int) -> int:
    return (n + m - 1) // m * m


def conv_out(h: int, k: int, stride: int, pad: int) -> int:
    return (h + 2 * pad - k) // stride + 1


def _desc(N, Hs, Ws, Cs, Ho, Wo, R, S, stride, pad, mode, Nc, ldc, out_sp=1, out_H=0, out_W=0, stride_w=None, pad_w=None) -> ConvDesc:
    if stride_w is None and pad_w is None:
        return ConvDesc(N, Hs, Ws, Cs, Ho, Wo, R, S, stride, pad, mode, Nc, ldc, out_sp, out_H, out_W)
    return ConvDesc(N, Hs, Ws, Cs, Ho, Wo, R, S, stride, pad, mode, Nc, ldc, out_sp, out_H, out_W, 1, stride if stride_w is None else stride_w,
                    pad if pad_w is None else pad_w)  # (stride, pad) vertical, (stride_w, pad_w) horizontal: forward geometry only


_DROPOUT_COUNTER = [None]  # device uint32/int32 tensor or None


def set_dropout_counter(counter: Optional[torch.Tensor]) -> None:
    """Device-resident step counter handed to every dropout-capable launch from now on (None = plain seeds).  A step
    captured in a HIP graph re-keys its dropout masks from it at every replay: increment it between replays.  This is
    host-side launch context only - the C ABI takes the pointer per call and keeps no state."""
    if counter is not None:
        assert counter.is_cuda and counter.numel() == 1 and counter.element_size() == 4
    _DROPOUT_COUNTER[0] = counter


def _ctr():
    c = _DROPOUT_COUNTER[0]
    return c.data_ptr() if c is not None else None


def _epi(bias=None, residual=None, mask_src=None, relu=False, sigmoid=False, dropout_p=0.0, seed=0, alpha=1.0) -> Epilogue:
    return Epilogue(ptr(bias), ptr(residual), ptr(mask_src), int(relu), int(sigmoid), float(dropout_p), int(seed) & 0xFFFFFFFF, float(alpha),
                    _ctr() if dropout_p > 0 else None)


def conv_gemm_raw(src: Tensor, w: Tensor, out: Tensor, desc: ConvDesc, epi: Optional[Epilogue]):
    check(_hip.lib().td_conv_gemm(ptr(src), ptr(w), ptr(out), C.byref(desc), C.byref(epi) if epi is not None else None,
                                  dtype_code(src.dtype), stream_ptr()), "td_conv_gemm")


def conv_fwd(x: Tensor, w_fwd: Tensor, bias: Optional[Tensor], R: int, S: int, stride: int, pad: int, *,
             residual: Optional[Tensor] = None, relu: bool = False, out: Optional[Tensor] = None) -> Tensor:
    """x [N,H,W,C] (NHWC), w_fwd [Co, R*S*C]  ->  y [N,Ho,Wo,Co] = relu(conv(x) + bias + residual)."""
    N, H, W, Cs = x.shape
    Co = w_fwd.shape[0]
    assert w_fwd.shape[1] == R * S * Cs and w_fwd.dtype == x.dtype and x.is_contiguous() and w_fwd.is_contiguous()
    Ho, Wo = conv_out(H, R, stride, pad), conv_out(W, S, stride, pad)
    y = out if out is not None else torch.empty((N, Ho, Wo, Co), dtype=x.dtype, device=x.device)
    conv_gemm_raw(x, w_fwd, y, _desc(N, H, W, Cs, Ho, Wo, R, S, stride, pad, 0, Co, Co), _epi(bias, residual, None, relu))
    return y


def conv_dgrad(g: Tensor, w_dgrad: Tensor, in_hw: Tuple[int, int], R: int, S: int, stride: int, pad: int, *,
               residual: Optional[Tensor] = None, mask_src: Optional[Tensor] = None, out: Optional[Tensor] = None) -> Tensor:
    """g [N,Ho,Wo,Co] = grad of the conv output, w_dgrad [Ci, R*S*Co]  ->  dx [N,H,W,Ci] (+residual, masked)."""
    N, Hg, Wg, Co = g.shape
    H, W = in_hw
    Ci = w_dgrad.shape[0]
    assert w_dgrad.shape[1] == R * S * Co and g.is_contiguous()
    dx = out if out is not None else torch.empty((N, H, W, Ci), dtype=g.dtype, device=g.device)
    conv_gemm_raw(g, w_dgrad, dx, _desc(N, Hg, Wg, Co, H, W, R, S, stride, pad, 1, Ci, Ci), _epi(None, residual, mask_src))
    return dx


def conv1x1s_dgrad_scatter(g: Tensor, w_dgrad: Tensor, dx: Tensor, stride: int, *, mask_src: Optional[Tensor] = None):
    """Input gradient of a strided 1x1 conv, accumulated in place: dx[n, ho*s, wo*s, :] = (dx + g @ W) (masked).
    Positions not hit by the stride keep dx but still need the mask: callers pass mask_src only when dx was
    already masked or apply the mask themselves."""
    N, Hg, Wg, Co = g.shape
    _, H, W, Ci = dx.shape
    d = _desc(N, Hg, Wg, Co, Hg, Wg, 1, 1, 1, 0, 0, Ci, Ci, stride, H, W)
    conv_gemm_raw(g, w_dgrad, dx, d, _epi(None, dx, mask_src))
    return dx


def conv_wgrad(g: Tensor, x: Tensor, R: int, S: int, stride: int, pad: int, *, out: Optional[Tensor] = None, splits: int = 0) -> Tensor:
    """dw_k [Co, R*S*C] fp32 (+)= sum_m g[m][co] * im2col(x)[m][k].  `out` (fp32) is accumulated into."""
    N, H, W, Cs = x.shape
    _, Ho, Wo, Co = g.shape
    dw = out if out is not None else zeros_f32((Co, R * S * Cs), x.device)
    d = _desc(N, H, W, Cs, Ho, Wo, R, S, stride, pad, 0, Co, Co)
    check(_hip.lib().td_conv_wgrad(ptr(g), ptr(x), ptr(dw), C.byref(d), Co, dtype_code(x.dtype), splits, stream_ptr()), "td_conv_wgrad")
    return dw


class _JobTables:
    """Caller-side staging of the job tables of td_conv_wgrad_batch / td_resnet_bwd (the C ABI allocates nothing): a
    ring of (page-locked host, device) buffer pairs.  An eager call recycles the pair used eight calls ago after its
    event has passed; a call made inside a stream capture takes a pair out of the ring for good (the captured copy node
    re-reads the host buffer at every replay), so the pairs must exist before the capture starts - any eager step
    creates them."""

    RING = 12

    def __init__(self):
        self.slots, self.retired, self.cap, self.next = [], [], 0, 0

    def _grow(self, nbytes, device):
        self.cap = max(self.cap, (nbytes + 4095) // 4096 * 4096, 65536)
        self.slots = [s for s in self.slots if s[0].numel() >= self.cap]
        while len(self.slots) < self.RING:
            self.slots.append([torch.empty(self.cap, dtype=torch.uint8, pin_memory=True),
                               torch.empty(self.cap, dtype=torch.uint8, device=device), None])

    def take(self, nbytes: int, device):
        """-> (host tensor, device tensor, done) ; call done() after the launch has been enqueued."""
        capturing = torch.cuda.is_current_stream_capturing()
        if capturing:
            ok = [s for s in self.slots if s[0].numel() >= nbytes and s[1].device == device]
            if not ok:
                raise RuntimeError("td job tables must exist before a stream capture: run one training step eagerly first")
            slot = ok[0]
            self.slots.remove(slot)
            self.retired.append(slot)  # alive (and untouched) for as long as the graph may replay
            return slot[0], slot[1], (lambda: None)
        if len(self.slots) < self.RING or self.cap < nbytes:
            self._grow(nbytes, device)
        slot = self.slots[self.next % len(self.slots)]
        self.next += 1
        if slot[2] is not None:
            slot[2].synchronize()  # the launch that last used this pair (RING calls ago) must have consumed it

        def done():
            ev = torch.cuda.Event()
            ev.record()
            slot[2] = ev

        return slot[0], slot[1], done


job_tables = _JobTables()


def conv_wgrad_batch(jobs) -> list:
    """jobs: iterable of (g NHWC [N,Ho,Wo,Co], x NHWC [N,H,W,C], R, S, stride, pad, scale [Co] fp32 | None, ci_real).
    One launch (two when pointwise and spatial kernels are mixed); returns dW per job in the parameter layout
    [Co, ci_real, R, S] fp32 with `scale` folded in."""
    jobs = list(jobs)
    arr = (_hip.WgradJob * len(jobs))()
    outs = []
    for a, (g, x, R, S, stride, pad, scale, ci_real) in zip(arr, jobs):
        N, H, W, Cs = x.shape
        _, Ho, Wo, Co = g.shape
        assert g.is_contiguous() and x.is_contiguous() and g.dtype == x.dtype
        dw = torch.empty((Co, ci_real, R, S), dtype=torch.float32, device=x.device)
        outs.append(dw)
        a.g, a.src, a.dW, a.scale = g.data_ptr(), x.data_ptr(), dw.data_ptr(), (scale.data_ptr() if scale is not None else None)
        a.d = _desc(N, H, W, Cs, Ho, Wo, R, S, stride, pad, 0, Co, Co)
        a.ldg, a.ci_real, a.dbias = Co, ci_real, None
    nbytes = _hip.lib().td_conv_wgrad_batch_table_bytes(len(jobs))
    th, td_, done = job_tables.take(nbytes, jobs[0][0].device)
    check(_hip.lib().td_conv_wgrad_batch(arr, len(jobs), dtype_code(jobs[0][0].dtype), th.data_ptr(), td_.data_ptr(), nbytes, stream_ptr()),
          "td_conv_wgrad_batch")
    done()
    return outs


def linear_wgrad_batch(jobs) -> None:
    """ONE launch for the weight (and bias) gradients of many linear layers: jobs = list of (g [M,N], x [M,K], dW data
    pointer of an [N,K] fp32 buffer, dbias data pointer of an [N] fp32 buffer or None).  Every output element is written
    (no zero-initialisation needed).  g and x must stay alive until the stream has passed the launch."""
    if not jobs:
        return
    # one launch per (dtype, device): the table carries ONE element size (a mixed queue - an fp32 head next to bf16 layers,
    # TD_HIP_ROBERTA=0 - would otherwise be read with the wrong one)
    key0 = (jobs[0][0].dtype, jobs[0][0].device)
    if any((j[0].dtype, j[0].device) != key0 for j in jobs):
        groups: dict = {}
        for j in jobs:
            groups.setdefault((j[0].dtype, j[0].device), []).append(j)
        for sub in groups.values():
            linear_wgrad_batch(sub)
        return
    arr = (_hip.WgradJob * len(jobs))()
    for a, job in zip(arr, jobs):
        g, x, dw_ptr, db_ptr = job[:4]
        M, K = x.shape
        Nn = g.shape[1]
        assert g.shape[0] == M and g.stride(1) == 1 and x.is_contiguous() and g.dtype == x.dtype  # g: rows of stride ldg (a column block is fine)
        a.g, a.src, a.dW, a.scale, a.dbias = g.data_ptr(), x.data_ptr(), dw_ptr, None, db_ptr
        a.d = _desc(1, M, 1, K, M, 1, 1, 1, 1, 0, 0, Nn, Nn)
        a.ldg, a.ci_real = g.stride(0), K
        a.accumulate = int(len(job) > 4 and bool(job[4]))  # second operand stream of a two-source layer: dW += g^T x, behind the overwriting jobs
    nbytes = _hip.lib().td_conv_wgrad_batch_table_bytes(len(jobs))
    dev = jobs[0][0].device
    th, td_, done = job_tables.take(nbytes, dev)
    check(_hip.lib().td_conv_wgrad_batch(arr, len(jobs), dtype_code(jobs[0][0].dtype), th.data_ptr(), td_.data_ptr(), nbytes, stream_ptr()),
          "td_conv_wgrad_batch(linears)")
    done()


def linear_fwd(x: Tensor, w: Tensor, bias: Optional[Tensor] = None, *, residual=None, relu=False, sigmoid=False,
               mask_src=None, dropout_p=0.0, seed=0, alpha=1.0, out: Optional[Tensor] = None) -> Tensor:
    """x [M,K], w [N,K] (K contiguous)  ->  [M,N] = epilogue(alpha * x @ w^T + bias + residual)."""
    M, K = x.shape
    Nn = w.shape[0]
    assert w.shape[1] == K and w.dtype == x.dtype and x.is_contiguous() and w.is_contiguous()
    y = out if out is not None else torch.empty((M, Nn), dtype=x.dtype, device=x.device)
    conv_gemm_raw(x, w, y, _desc(1, M, 1, K, M, 1, 1, 1, 1, 0, 0, Nn, Nn), _epi(bias, residual, mask_src, relu, sigmoid, dropout_p, seed, alpha))
    return y


def pw_chain2(y2: Tensor, w3: Tensor, b3: Tensor, residual: Tensor, w1: Tensor, b1: Tensor, *, out: Optional[Tensor] = None,
              h1: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
    """td_pw_chain2: out [M,4P] = relu(y2 [M,P] @ w3^T + b3 + residual), h1 [M,P] = relu(out @ w1^T + b1) in one launch (bf16, P = 256):
    conv3 + identity of one layer3 bottleneck and conv1 of the next; bit-identical to the two linear_fwd calls."""
    M, P = y2.shape
    assert y2.dtype == torch.bfloat16 and w3.shape == (4 * P, P) and w1.shape == (P, 4 * P) and residual.shape == (M, 4 * P)
    assert y2.is_contiguous() and w3.is_contiguous() and w1.is_contiguous() and residual.is_contiguous()
    assert b3.dtype == torch.float32 and b1.dtype == torch.float32 and b3.numel() == 4 * P and b1.numel() == P
    out = out if out is not None else torch.empty((M, 4 * P), dtype=y2.dtype, device=y2.device)
    h1 = h1 if h1 is not None else torch.empty((M, P), dtype=y2.dtype, device=y2.device)
    check(_hip.lib().td_pw_chain2(ptr(y2), ptr(w3), ptr(b3), ptr(residual), ptr(out), ptr(w1), ptr(b1), ptr(h1), M, P, dtype_code(y2.dtype),
                                  stream_ptr()), "td_pw_chain2")
    return out, h1


def _rows2d(t: Tensor):
    assert t.dim() == 2 and t.stride(1) == 1, "row-major 2-D operand expected (unit column stride)"
    return t


def linear_ex(a1: Tensor, w: Tensor, bias: Optional[Tensor] = None, *, a2: Optional[Tensor] = None, a1_map: Optional[Tensor] = None,
              a2_map: Optional[Tensor] = None, w_shared: bool = False, out: Optional[Tensor] = None, out_map: Optional[Tensor] = None,
              out_rows: Optional[int] = None, residual: Optional[Tensor] = None, res_map: Optional[Tensor] = None, relu=False, mask_src=None,
              dropout_p=0.0, seed=0, alpha=1.0) -> Tensor:
    """td_linear_ex: out[out_map[m]] = epilogue([a1[a1_map[m]] | a2[a2_map[m]]] @ w^T + bias + residual[res_map[m]]).  a1 / a2 / out /
    residual are row-major 2-D (row stride = stride(0)); maps are int32 device vectors of length M (None = identity);
    w_shared: w is [N, K1] and multiplies both sources (= (a1 + a2) @ w^T)."""
    _rows2d(a1)
    K1 = a1.shape[1]
    K2 = _rows2d(a2).shape[1] if a2 is not None else 0
    M = int(a1_map.numel()) if a1_map is not None else a1.shape[0]
    Nn = w.shape[0]
    assert w.is_contiguous() and w.dtype == a1.dtype and w.shape[1] == (K1 if (w_shared or a2 is None) else K1 + K2)
    for m_ in (a1_map, a2_map, out_map, res_map):
        assert m_ is None or (m_.dtype == torch.int32 and m_.is_contiguous() and m_.numel() == M)
    if out is None:
        out = torch.empty((out_rows if out_rows is not None else M, Nn), dtype=a1.dtype, device=a1.device)
    _rows2d(out)
    x = _hip.LinearExDesc()
    x.M, x.N, x.K1, x.K2 = M, Nn, K1, K2
    x.lda1, x.lda2, x.ldc = a1.stride(0), (a2.stride(0) if a2 is not None else 0), out.stride(0)
    x.ldr = _rows2d(residual).stride(0) if residual is not None else 0
    x.w_shared = int(bool(w_shared and a2 is not None))
    x.rows1, x.rows2 = a1.shape[0], (a2.shape[0] if a2 is not None else 0)
    x.a1_map, x.a2_map, x.out_map, x.res_map = ptr(a1_map), ptr(a2_map), ptr(out_map), ptr(res_map)
    epi = _epi(bias, residual, mask_src, relu, False, dropout_p, seed, alpha)
    check(_hip.lib().td_linear_ex(ptr(a1), ptr(a2), ptr(w), ptr(out), C.byref(x), C.byref(epi), dtype_code(a1.dtype), stream_ptr()), "td_linear_ex")
    return out


def rows_copy(src: Tensor, src_map: Optional[Tensor], dst: Tensor, dst_map: Optional[Tensor], n_rows: int, add: Optional[Tensor] = None) -> Tensor:
    """dst[dst_map[i]] = src[src_map[i]] (+ add[i]) for i < n_rows; all row-major 2-D with the same column count."""
    _rows2d(src), _rows2d(dst)
    cols = src.shape[1]
    assert dst.shape[1] == cols and dst.dtype == src.dtype and (add is None or (_rows2d(add).shape[1] == cols and add.dtype == src.dtype and add.shape[0] >= n_rows))
    for m_ in (src_map, dst_map):
        assert m_ is None or (m_.dtype == torch.int32 and m_.is_contiguous() and m_.numel() == n_rows)
    check(_hip.lib().td_rows_copy(ptr(src), ptr(src_map), ptr(add), ptr(dst), ptr(dst_map), n_rows, cols, src.stride(0), add.stride(0) if add is not None else 0,
                                  dst.stride(0), dtype_code(src.dtype), stream_ptr()), "td_rows_copy")
    return dst


def rows_segment_sum(inp: Tensor, idx: Tensor, seg_ptr: Tensor, out: Tensor, out_map: Optional[Tensor] = None) -> Tensor:
    """out[out_map[r]] = sum of inp[idx[j]] over j in [seg_ptr[r], seg_ptr[r+1]) (fp32 accumulation)."""
    _rows2d(inp), _rows2d(out)
    n_out = seg_ptr.numel() - 1
    assert idx.dtype == torch.int32 and seg_ptr.dtype == torch.int32 and out.dtype == inp.dtype and out.shape[1] == inp.shape[1]
    assert out_map is None or (out_map.dtype == torch.int32 and out_map.numel() == n_out)
    check(_hip.lib().td_rows_segment_sum(ptr(inp), ptr(idx), ptr(seg_ptr), ptr(out), ptr(out_map), n_out, inp.shape[1], inp.stride(0), out.stride(0),
                                         dtype_code(inp.dtype), stream_ptr()), "td_rows_segment_sum")
    return out


def linear_wgrad(g: Tensor, x: Tensor, *, out: Optional[Tensor] = None, splits: int = 0, dbias: Optional[Tensor] = None) -> Tensor:
    """dW [N,K] fp32 (+)= g^T @ x   with g [M,N], x [M,K]; optionally dbias [N] fp32 += column sums of g (same launch)."""
    M, K = x.shape
    Nn = g.shape[1]
    dw = out if out is not None else zeros_f32((Nn, K), x.device)
    d = _desc(1, M, 1, K, M, 1, 1, 1, 1, 0, 0, Nn, Nn)
    if dbias is not None:
        assert dbias.dtype == torch.float32 and dbias.numel() == Nn and dbias.is_contiguous()
    assert g.stride(1) == 1 and x.is_contiguous()  # g may be a column block of a wider row-major buffer (row stride = ldg)
    check(_hip.lib().td_conv_wgrad_bias(ptr(g), ptr(x), ptr(dw), ptr(dbias), C.byref(d), g.stride(0), dtype_code(x.dtype), splits, stream_ptr()),
          "td_conv_wgrad")
    return dw


def weight_prep(W: Tensor, dtype: torch.dtype, *, bn=None, bias: Optional[Tensor] = None, need_dgrad: bool = True, cpad: Optional[int] = None):
    """W fp32 [Co,Ci,R,S] (or [Co,Ci]) -> (w_fwd [Co,R*S*Cpad], w_dgrad [Ci,R*S*Co] | None, bias_out [Co] fp32, scale [Co] fp32).
    bn = (weight, bias, running_mean, running_var) of a FrozenBatchNorm2d to fold, or None."""
    if W.dim() == 2:
        Co, Ci = W.shape
        R = S = 1
    else:
        Co, Ci, R, S = W.shape
    cp = cpad if cpad is not None else pad_to(Ci, vec_of(dtype))
    dev = W.device
    wf = torch.empty((Co, R * S * cp), dtype=dtype, device=dev)
    wd = torch.empty((Ci, R * S * Co), dtype=dtype, device=dev) if need_dgrad else None
    b_out = torch.empty(Co, dtype=torch.float32, device=dev) if (bn is not None or bias is not None) else None
    sc = torch.empty(Co, dtype=torch.float32, device=dev) if bn is not None else None
    bw, bb, brm, brv = bn if bn is not None else (None, None, None, None)
    check(_hip.lib().td_weight_prep(ptr(W), ptr(bw), ptr(bb), ptr(brm), ptr(brv), ptr(bias), Co, Ci, R, S, cp, ptr(wf), ptr(wd),
                                    ptr(b_out), ptr(sc), dtype_code(dtype), stream_ptr()), "td_weight_prep")
    return wf, wd, b_out, sc


def wgrad_finalize(dw_k: Tensor, scale: Optional[Tensor], shape, cpad: int, out: Optional[Tensor] = None, accumulate: bool = False) -> Tensor:
    Co, Ci, R, S = shape
    dW = out if out is not None else torch.empty(shape, dtype=torch.float32, device=dw_k.device)
    check(_hip.lib().td_wgrad_finalize(ptr(dw_k), ptr(scale), ptr(dW), Co, Ci, R, S, cpad, int(accumulate), stream_ptr()), "td_wgrad_finalize")
    return dW


def nchw_to_nhwc(x: Tensor, dtype: torch.dtype, cpad: int) -> Tensor:
    N, Cc, H, W = x.shape
    assert x.dtype == torch.float32 and x.is_contiguous()
    y = torch.empty((N, H, W, cpad), dtype=dtype, device=x.device)
    check(_hip.lib().td_nchw_to_nhwc(ptr(x), ptr(y), N, Cc, H, W, cpad, dtype_code(dtype), stream_ptr()), "td_nchw_to_nhwc")
    return y


def nhwc_to_nchw(x: Tensor) -> Tensor:
    N, H, W, Cc = x.shape
    y = torch.empty((N, Cc, H, W), dtype=torch.float32, device=x.device)
    check(_hip.lib().td_nhwc_to_nchw(ptr(x), ptr(y), N, Cc, H, W, dtype_code(x.dtype), stream_ptr()), "td_nhwc_to_nchw")
    return y


def cast(x: Tensor, dtype: torch.dtype) -> Tensor:
    if x.dtype == dtype:
        return x
    y = torch.empty(x.shape, dtype=dtype, device=x.device)
    check(_hip.lib().td_cast(ptr(x.contiguous()), ptr(y), x.numel(), dtype_code(x.dtype), dtype_code(dtype), stream_ptr()), "td_cast")
    return y


def maxpool3x3s2(x: Tensor) -> Tensor:
    N, H, W, Cc = x.shape
    y = torch.empty((N, conv_out(H, 3, 2, 1), conv_out(W, 3, 2, 1), Cc), dtype=x.dtype, device=x.device)
    check(_hip.lib().td_maxpool3x3s2(ptr(x), ptr(y), N, H, W, Cc, dtype_code(x.dtype), stream_ptr()), "td_maxpool3x3s2")
    return y


def add_layernorm_fwd(x: Tensor, r: Optional[Tensor], gamma: Tensor, beta: Tensor, eps: float, save: bool = True):
    rows, cols = x.shape
    y = torch.empty_like(x)
    s = torch.empty_like(x) if (save and r is not None) else None
    mean = torch.empty(rows, dtype=torch.float32, device=x.device) if save else None
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device) if save else None
    check(_hip.lib().td_add_layernorm_fwd(ptr(x), ptr(r), ptr(gamma), ptr(beta), ptr(y), ptr(s), ptr(mean), ptr(rstd), rows, cols, eps,
                                          dtype_code(x.dtype), stream_ptr()), "td_add_layernorm_fwd")
    return y, (s if r is not None else x), mean, rstd


def add_layernorm_bwd(dy: Tensor, s: Tensor, mean: Tensor, rstd: Tensor, gamma: Tensor, extra: Optional[Tensor] = None):
    rows, cols = dy.shape
    ds = torch.empty_like(dy)
    dgamma = zeros_f32(cols, dy.device)
    dbeta = zeros_f32(cols, dy.device)
    check(_hip.lib().td_add_layernorm_bwd(ptr(dy), ptr(s), ptr(mean), ptr(rstd), ptr(gamma), ptr(extra), ptr(ds), ptr(dgamma), ptr(dbeta),
                                          rows, cols, dtype_code(dy.dtype), stream_ptr()), "td_add_layernorm_bwd")
    return ds, dgamma, dbeta


def colsum(g: Tensor, out: Optional[Tensor] = None) -> Tensor:
    rows, cols = g.shape
    o = out if out is not None else zeros_f32(cols, g.device)
    check(_hip.lib().td_colsum(ptr(g), ptr(o), rows, cols, cols, dtype_code(g.dtype), stream_ptr()), "td_colsum")
    return o


def add(a: Tensor, b: Optional[Tensor]) -> Tensor:
    y = torch.empty_like(a)
    check(_hip.lib().td_add(ptr(a), ptr(b), ptr(y), a.numel(), dtype_code(a.dtype), stream_ptr()), "td_add")
    return y


def relu_bwd(dy: Tensor, y: Tensor, scale: float = 1.0) -> Tensor:
    g = torch.empty_like(dy)
    check(_hip.lib().td_relu_bwd(ptr(dy), ptr(y), ptr(g), dy.numel(), scale, dtype_code(dy.dtype), stream_ptr()), "td_relu_bwd")
    return g


def pos_sine(mask: Tensor, npf: int, dtype: torch.dtype, temperature: float = 10000.0, rows: Optional[int] = None) -> Tensor:
    """mask (N,h,w) bool/uint8 -> pos [N, rows, 2*npf]; rows >= h*w (default h*w), the rows behind the h*w tokens are zeros (the
    positional operand of the text tokens that follow the visual ones in the encoder's sequence)."""
    N, h, w = mask.shape
    m8 = mask.to(torch.uint8).contiguous()
    rows = h * w if rows is None else int(rows)
    pos = torch.empty((N, rows, 2 * npf), dtype=dtype, device=mask.device)
    check(_hip.lib().td_pos_sine(ptr(m8), ptr(pos), N, h, w, npf, temperature, rows, dtype_code(dtype), stream_ptr()), "td_pos_sine")
    return pos


def mha_fwd(q: Tensor, k: Tensor, v: Tensor, key_pad: Optional[Tensor], H: int, scale: float, *, need_wavg: bool = False,
            dropout_p: float = 0.0, seed: int = 0):
    """q [B,Lq,>=E] / k,v [B,Lk,>=E] views with unit stride on the last dim and row stride = stride(1)."""
    B, Lq, E = q.shape
    Lk = k.shape[1]
    hd = E // H
    for t_ in (q, k, v):
        assert t_.stride(2) == 1 and t_.stride(0) == t_.shape[1] * t_.stride(1)
    out = torch.empty((B, Lq, E), dtype=q.dtype, device=q.device)
    probs = torch.empty((B, H, Lq, Lk), dtype=torch.float32, device=q.device)
    wavg = torch.empty((B, Lq, Lk), dtype=torch.float32, device=q.device) if need_wavg else None
    kp = key_pad.to(torch.uint8).contiguous() if key_pad is not None else None
    check(_hip.lib().td_mha_fwd(ptr(q), ptr(k), ptr(v), ptr(kp), ptr(out), ptr(probs), ptr(wavg), B, H, Lq, Lk, hd, q.stride(1), k.stride(1),
                                v.stride(1), E, scale, dropout_p, seed & 0xFFFFFFFF, _ctr() if dropout_p > 0 else None, dtype_code(q.dtype), stream_ptr()),
          "td_mha_fwd")
    return out, probs, wavg


def mha_lean_ok(q: Tensor, k: Tensor, v: Tensor, H: int) -> bool:
    """Can td_mha_lean_* take these projected rows?  (bf16, head dim 32, Lk <= 256, Lq <= 448, 16-byte aligned rows)"""
    if os.environ.get("TD_MHA_LEAN", "1") == "0":
        return False
    E = q.shape[2]
    return (q.dtype == torch.bfloat16 and E // H == 32 and k.shape[1] <= 256 and q.shape[1] <= 448
            and all(t_.stride(1) % 8 == 0 and t_.data_ptr() % 16 == 0 for t_ in (q, k, v)))


def mha_lean_fwd(q: Tensor, k: Tensor, v: Tensor, key_pad: Optional[Tensor], H: int, scale: float, *, dropout_p: float = 0.0, seed: int = 0):
    """Attention core without the weights: -> (out [B,Lq,E], stats [B*H*Lq, 4] fp32, key_pad uint8 or None) - what
    mha_lean_bwd needs besides q, k, v and the output."""
    B, Lq, E = q.shape
    Lk = k.shape[1]
    for t_ in (q, k, v):
        assert t_.stride(2) == 1 and t_.stride(0) == t_.shape[1] * t_.stride(1)
    out = torch.empty((B, Lq, E), dtype=q.dtype, device=q.device)
    stats = torch.empty((B * H * Lq, 4), dtype=torch.float32, device=q.device)
    kp = key_pad.to(torch.uint8).contiguous() if key_pad is not None else None
    check(_hip.lib().td_mha_lean_fwd(ptr(q), ptr(k), ptr(v), ptr(kp), ptr(out), ptr(stats), B, H, Lq, Lk, E // H, q.stride(1), k.stride(1), v.stride(1), E,
                                     scale, dropout_p, seed & 0xFFFFFFFF, _ctr() if dropout_p > 0 else None, dtype_code(q.dtype), stream_ptr()),
          "td_mha_lean_fwd")
    return out, stats, kp


def mha_lean_bwd(q: Tensor, k: Tensor, v: Tensor, kp: Optional[Tensor], out: Tensor, dout: Tensor, stats: Tensor, H: int, scale: float,
                 dq: Tensor, dk: Tensor, dv: Tensor, *, dropout_p: float = 0.0, seed: int = 0):
    B, Lq, E = q.shape
    Lk = k.shape[1]
    assert dout.is_contiguous() and out.is_contiguous()
    for a, b in ((q, dq), (k, dk), (v, dv)):
        assert a.stride() == b.stride() and a.shape == b.shape
    check(_hip.lib().td_mha_lean_bwd(ptr(q), ptr(k), ptr(v), ptr(kp), ptr(out), ptr(dout), ptr(stats), ptr(dq), ptr(dk), ptr(dv), B, H, Lq, Lk, E // H,
                                     q.stride(1), k.stride(1), v.stride(1), E, scale, dropout_p, seed & 0xFFFFFFFF, _ctr() if dropout_p > 0 else None,
                                     dtype_code(q.dtype), stream_ptr()), "td_mha_lean_bwd")
    return dq, dk, dv


def head_blocks_expand(W: Tensor, bias: Optional[Tensor], alpha: float, H: int, dtype: torch.dtype):
    """One [E,E] slice of a packed in_proj weight (fp32, [out][in]) -> (w_n [E, H*E + nb], w_t [H*E + nb, E]) in ``dtype``: row j of
    w_n holds W[j] * alpha in column block head(j) and, with a bias (nb = H), bias[j] in column H*E + head(j)."""
    E = W.shape[0]
    assert W.shape == (E, E) and W.dtype == torch.float32 and W.is_contiguous() and (bias is None or (bias.dtype == torch.float32 and bias.is_contiguous()))
    ld = H * E + (H if bias is not None else 0)
    w_n = torch.empty((E, ld), dtype=dtype, device=W.device)
    w_t = torch.empty((ld, E), dtype=dtype, device=W.device)
    check(_hip.lib().td_head_blocks_expand(ptr(W), ptr(bias), float(alpha), ptr(w_n), ptr(w_t), E, H, dtype_code(dtype), stream_ptr()), "td_head_blocks_expand")
    return w_n, w_t


def head_blocks_extract(G: Tensor, alpha: float, dW: Tensor, db: Optional[Tensor], H: int) -> None:
    """dW [E,E] fp32 (a row slice of a packed gradient is fine) and optionally db [E] from the dense gradient G [E, H*E (+ H)] of w_n."""
    E = dW.shape[0]
    assert G.dtype == torch.float32 and G.is_contiguous() and G.shape == (E, H * E + (H if db is not None else 0))
    assert dW.shape == (E, E) and dW.dtype == torch.float32 and dW.is_contiguous() and (db is None or (db.is_contiguous() and db.dtype == torch.float32))
    check(_hip.lib().td_head_blocks_extract(ptr(G), float(alpha), ptr(dW), ptr(db), E, H, stream_ptr()), "td_head_blocks_extract")


def cross_q1_fwd(u: Tensor, mem: Tensor, pos: Optional[Tensor], key_pad: Optional[Tensor], F: int, S: int, H: int, *, need_wavg: bool = True,
                 dropout_p: float = 0.0, seed: int = 0):
    """Frame core of the time-aligned cross-attention with the projections on the query side (csrc/cross_attn.hip):
    u [F, H*E] (scaled W_k,h^T q), mem / pos [F*S, E]  ->  probs [F,H,S] fp32, wavg [F,1,S] fp32 | None, zext [F, H*E + H]."""
    E = mem.shape[1]
    assert u.shape == (F, H * E) and mem.shape == (F * S, E) and u.is_contiguous() and mem.is_contiguous() and u.dtype == mem.dtype
    assert pos is None or (pos.shape == mem.shape and pos.is_contiguous() and pos.dtype == mem.dtype)
    kp = key_pad.to(torch.uint8).contiguous() if key_pad is not None else None
    assert kp is None or kp.numel() == F * S
    ldz = H * E + H
    probs = torch.empty((F, H, S), dtype=torch.float32, device=u.device)
    wavg = torch.empty((F, 1, S), dtype=torch.float32, device=u.device) if need_wavg else None  # nn.MultiheadAttention's [B, Lq, Lk]
    zext = torch.empty((F, ldz), dtype=u.dtype, device=u.device)
    check(_hip.lib().td_cross_q1_fwd(ptr(u), ptr(mem), ptr(pos), ptr(kp), ptr(probs), ptr(wavg), ptr(zext), F, S, H, E, ldz, dropout_p, seed & 0xFFFFFFFF,
                                     _ctr() if dropout_p > 0 else None, dtype_code(u.dtype), stream_ptr()), "td_cross_q1_fwd")
    return probs, wavg, zext


def cross_q1_bwd(u: Tensor, mem: Tensor, pos: Optional[Tensor], probs: Tensor, d_zext: Tensor, dwavg: Optional[Tensor], d_mem: Optional[Tensor], accumulate: bool,
                 F: int, S: int, H: int, *, dropout_p: float = 0.0, seed: int = 0) -> Tensor:
    """-> d_u [F, H*E]; d_mem [F*S, E] fp32 is overwritten (accumulate=False) or added to; None: the memory needs no gradient."""
    E = mem.shape[1]
    assert d_zext.shape == (F, H * E + H) and d_zext.is_contiguous() and d_zext.dtype == u.dtype and probs.is_contiguous()
    assert d_mem is None or (d_mem.shape == (F * S, E) and d_mem.dtype == torch.float32 and d_mem.is_contiguous())
    assert dwavg is None or (dwavg.dtype == torch.float32 and dwavg.is_contiguous() and dwavg.numel() == F * S)
    d_u = torch.empty((F, H * E), dtype=u.dtype, device=u.device)
    check(_hip.lib().td_cross_q1_bwd(ptr(u), ptr(mem), ptr(pos), ptr(probs), ptr(d_zext), ptr(dwavg), ptr(d_u), ptr(d_mem), int(bool(accumulate)), F, S, H, E,
                                     H * E + H, dropout_p, seed & 0xFFFFFFFF, _ctr() if dropout_p > 0 else None, dtype_code(u.dtype), stream_ptr()),
          "td_cross_q1_bwd")
    return d_u


def cross_q1_bwd_coef(u: Tensor, mem: Tensor, pos: Optional[Tensor], probs: Tensor, d_zext: Tensor, dwavg: Optional[Tensor], coef: Tensor, coef_col: int,
                      F: int, S: int, H: int, *, dropout_p: float = 0.0, seed: int = 0) -> Tensor:
    """cross_q1_bwd for the deferred d(memory) of the bf16 mode: -> d_u [F, H*E]; instead of touching a [F*S, E] gradient the layer leaves
    its sixteen coefficients per memory row in columns coef_col .. coef_col + 15 of ``coef`` [F*S, KP] bf16 (cross_q1_dmem forms the
    gradient of all layers from them in one pass)."""
    E = mem.shape[1]
    assert u.dtype == torch.bfloat16 and d_zext.shape == (F, H * E + H) and d_zext.is_contiguous() and d_zext.dtype == u.dtype and probs.is_contiguous()
    assert coef.dtype == torch.bfloat16 and coef.is_contiguous() and coef.shape[0] == F * S and coef.shape[1] % 32 == 0 and coef_col % 16 == 0 and coef_col + 16 <= coef.shape[1]
    assert dwavg is None or (dwavg.dtype == torch.float32 and dwavg.is_contiguous() and dwavg.numel() == F * S)
    d_u = torch.empty((F, H * E), dtype=u.dtype, device=u.device)
    check(_hip.lib().td_cross_q1_bwd_coef(ptr(u), ptr(mem), ptr(pos), ptr(probs), ptr(d_zext), ptr(dwavg), ptr(d_u), ptr(coef), coef.shape[1], coef_col, F, S, H, E,
                                          H * E + H, dropout_p, seed & 0xFFFFFFFF, _ctr() if dropout_p > 0 else None, dtype_code(u.dtype), stream_ptr()),
          "td_cross_q1_bwd_coef")
    return d_u


def cross_q1_dmem(coef: Tensor, layers, F: int, S: int, H: int, E: int) -> Tensor:
    """d(memory) [F*S, E] bf16 of all the time-aligned cross-attention layers from their coefficient rows (cross_q1_bwd_coef) and
    ``layers`` = [(u_l, d_zext_l) | None per layer]: one MFMA product per frame (csrc/cross_attn.hip cross_q1_dmem_kernel)."""
    import ctypes as C

    n = len(layers)
    assert coef.dtype == torch.bfloat16 and coef.is_contiguous() and coef.shape == (F * S, coef.shape[1]) and coef.shape[1] >= 16 * n
    us = (C.c_void_p * n)(*[(ptr(l[0]) if l is not None else None) for l in layers])
    dzs = (C.c_void_p * n)(*[(ptr(l[1]) if l is not None else None) for l in layers])
    for l in layers:
        assert l is None or (l[0].dtype == torch.bfloat16 and l[0].shape == (F, H * E) and l[0].is_contiguous() and l[1].shape == (F, H * E + H) and l[1].is_contiguous())
    d_mem = torch.empty((F * S, E), dtype=torch.bfloat16, device=coef.device)
    check(_hip.lib().td_cross_q1_dmem(ptr(coef), coef.shape[1], us, dzs, n, ptr(d_mem), F, S, H, E, H * E + H, dtype_code(coef.dtype), stream_ptr()), "td_cross_q1_dmem")
    return d_mem


def mha_bwd(q: Tensor, k: Tensor, v: Tensor, dout: Tensor, probs: Tensor, dwavg: Optional[Tensor], H: int, scale: float,
            dq: Tensor, dk: Tensor, dv: Tensor, *, dropout_p: float = 0.0, seed: int = 0):
    """dq/dk/dv must be allocated by the caller with exactly the strides of q/k/v (e.g. views of a packed buffer)."""
    B, Lq, E = q.shape
    Lk = k.shape[1]
    hd = E // H
    ws = torch.empty_like(probs)
    assert dout.is_contiguous()
    for a, b in ((q, dq), (k, dk), (v, dv)):
        assert a.stride() == b.stride() and a.shape == b.shape
    check(_hip.lib().td_mha_bwd(ptr(q), ptr(k), ptr(v), ptr(dout), ptr(probs), ptr(dwavg), ptr(dq), ptr(dk), ptr(dv), ptr(ws), B, H, Lq, Lk, hd,
                                q.stride(1), k.stride(1), v.stride(1), E, scale, dropout_p, seed & 0xFFFFFFFF, _ctr() if dropout_p > 0 else None,
                                dtype_code(q.dtype), stream_ptr()), "td_mha_bwd")
    return dq, dk, dv


def gelu_fwd(x: Tensor) -> Tensor:
    y = torch.empty_like(x)
    check(_hip.lib().td_gelu_fwd(ptr(x), ptr(y), x.numel(), dtype_code(x.dtype), stream_ptr()), "td_gelu_fwd")
    return y


def gelu_bwd(dy: Tensor, x: Tensor) -> Tensor:
    dx = torch.empty_like(dy)
    check(_hip.lib().td_gelu_bwd(ptr(dy), ptr(x), ptr(dx), dy.numel(), dtype_code(dy.dtype), stream_ptr()), "td_gelu_bwd")
    return dx


def dropout(x: Tensor, p: float, seed: int) -> Tensor:
    y = torch.empty_like(x)
    check(_hip.lib().td_dropout(ptr(x), ptr(y), x.numel(), p, seed & 0xFFFFFFFF, _ctr() if p > 0 else None, dtype_code(x.dtype), stream_ptr()),
          "td_dropout")
    return y
