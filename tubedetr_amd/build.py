"""Build libtubedetr_hip.so (gfx950) in-tree with hipcc.  `python -m tubedetr_amd.build`."""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libtubedetr_hip.so")
SOURCES = ["api.cpp", "gemm_conv.hip", "prep.hip", "elementwise.hip", "attention.hip", "resnet_exec.hip", "optim.hip", "criterion.hip", "stem.hip", "bottleneck.hip", "cross_attn.hip", "chain.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-Wno-unused-result"]


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


# The kernels with hand-counted s_waitcnt sequences (conv_gemm_big8*, pw_resident2, pw_chain2: inline-asm loads / LDS-DMA whose completion the
# compiler does not track) were validated - bit for bit against their uncounted forms, tests/test_bench_shapes_gpu.py - with THIS compiler.  A
# different one may schedule its own scalar / vector loads into those sequences: the build goes on, loudly; re-run the GPU tests before trusting it.
HIPCC_VALIDATED = "roc-7.2.0"


def _check_compiler(hipcc: str) -> None:
    try:
        v = subprocess.run([hipcc, "--version"], capture_output=True, text=True, timeout=60).stdout
    except Exception as e:  # noqa: BLE001
        v = repr(e)
    if HIPCC_VALIDATED not in v:
        print(f"[tubedetr_amd.build] WARNING: hipcc is not the validated {HIPCC_VALIDATED} toolchain ({v.splitlines()[0] if v else 'unknown'}): "
              "run `pytest tests -m gpu` (test_persistent_256_row_kernel_equals_one_tile_per_workgroup, test_chained_conv3_conv1_pair_at_bench_shape) before use",
              file=sys.stderr, flush=True)


def _stale(out: str, deps) -> bool:
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def build_lib(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    hipcc = _hipcc()
    _check_compiler(hipcc)
    headers = [os.path.join(CSRC, "td_common.h"), os.path.join(os.path.dirname(HERE), "include", "tubedetr_hip.h")]
    objs, jobs = [], []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(LIBDIR, os.path.splitext(s)[0] + ".o")
        objs.append(obj)
        if force or _stale(obj, [src] + headers):
            cmd = [hipcc] + FLAGS + (["-x", "hip"] if s.endswith(".cpp") else []) + ["-c", src, "-o", obj]
            jobs.append(cmd)

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + r.stdout + r.stderr)

    with ThreadPoolExecutor(max_workers=8) as ex:
        list(ex.map(run, jobs))
    if force or jobs or _stale(LIB, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    return LIB


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv))
