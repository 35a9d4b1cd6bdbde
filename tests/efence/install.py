"""Install the electric-fence device allocator (tests/efence/efence.cpp) as torch's CUDA allocator.  Must run before
the first device allocation of the process.  Test infrastructure only."""
import ctypes
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libtd_efence.so")


def build(force: bool = False) -> str:
    src = os.path.join(HERE, "efence.cpp")
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(src):
        hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
        subprocess.run([hipcc, "-O2", "-fPIC", "-shared", "-std=c++17", "-x", "hip", "--offload-arch=gfx950", src, "-o", LIB], check=True)
    return LIB


def install():
    import torch

    build()
    alloc = torch.cuda.memory.CUDAPluggableAllocator(LIB, "td_efence_malloc", "td_efence_free")
    torch.cuda.memory.change_current_allocator(alloc)
    return ctypes.CDLL(LIB)


def protected() -> bool:
    """True when every allocation so far was fenced (the VMM API worked on this box)."""
    lib = ctypes.CDLL(LIB)
    lib.td_efence_protected.restype = ctypes.c_int
    return bool(lib.td_efence_protected())
