"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads and exports every symbol that
include/tubedetr_hip.h declares (no kernel is launched here - there is no GPU in the build container)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "tubedetr_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(td_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported():
    from tubedetr_amd import _hip
    from tubedetr_amd.build import build_lib

    build_lib(verbose=False)
    lib = ctypes.CDLL(_hip.lib_path())
    declared = _declared()
    assert len(declared) >= 15
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/tubedetr_hip.h but not exported"
    assert sorted(_hip.EXPORTS) == declared, "python binding table and header disagree"
    assert _hip.lib().td_abi_version() == _hip.EXPECTED_ABI


def test_binding_argument_counts_and_struct_layouts_match_the_header():
    """Every prototype of include/tubedetr_hip.h has as many parameters as the ctypes signature that calls it, and every
    struct as many fields as its ctypes mirror (a changed signature must not survive in the binding of an old checkout)."""
    from tubedetr_amd import _hip

    src = open(os.path.join(ROOT, "include", "tubedetr_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    flat = " ".join(src.split())
    protos = dict()
    for m in re.finditer(r"\b(?:int|size_t|const char\*)\s+(td_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", flat):
        args = m.group(2).strip()
        protos[m.group(1)] = 0 if args in ("", "void") else len(args.split(","))
    sigs = dict(_hip._SIGS)
    sigs.update(_hip._SIZE_SIGS)
    checked = 0
    for name, argtypes in sigs.items():
        assert name in protos, name
        assert protos[name] == len(argtypes), f"{name}: header has {protos[name]} parameters, the binding passes {len(argtypes)}"
        checked += 1
    assert checked >= 40
    structs = {m.group(2): m.group(1) for m in re.finditer(r"typedef struct \w+ \{(.*?)\} (td_\w+);", flat)}

    def n_fields(body):
        n = 0
        for decl in body.split(";"):
            decl = decl.strip()
            if decl:
                n += decl.count(",") + 1
        return n

    mirrors = {"td_conv_desc": _hip.ConvDesc, "td_epilogue": _hip.Epilogue, "td_wgrad_job": _hip.WgradJob, "td_frame_source": _hip.FrameSource,
               "td_prep_item": _hip.PrepItem, "td_optim_segment": _hip.OptimSegment, "td_linear_ex_desc": _hip.LinearExDesc}
    for cname, cls in mirrors.items():
        assert cname in structs, cname
        assert n_fields(structs[cname]) == len(cls._fields_), f"{cname}: {n_fields(structs[cname])} fields in the header, {len(cls._fields_)} in the binding"


def test_errors_are_reported_not_swallowed():
    """Invalid arguments return an error code + message (no launch is attempted)."""
    from tubedetr_amd import _hip

    L = _hip.lib()
    d = _hip.ConvDesc(1, 4, 4, 3, 4, 4, 1, 1, 1, 0, 0, 8, 8, 1, 0, 0)  # C=3 is not a multiple of the vector width
    rc = L.td_conv_gemm(ctypes.c_void_p(16), ctypes.c_void_p(16), ctypes.c_void_p(16), ctypes.byref(d), None, _hip.TD_BF16, None)
    assert rc != 0 and b"multiple" in L.td_last_error()
    rc = L.td_mha_fwd(*( [ctypes.c_void_p(16)] * 3 + [None, ctypes.c_void_p(16), ctypes.c_void_p(16), None] + [1, 8, 4, 4, 48, 512, 512, 512, 512, 0.1, 0.0, 0, None, 0, None]))
    assert rc != 0 and b"head dim" in L.td_last_error()


def test_product_has_no_oracle_or_cpu_fallback():
    """The product package must never import the oracle (parity would be void)."""
    pkg = os.path.join(ROOT, "tubedetr_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                txt = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt, f
