"""ctypes binding of libvllm_hip.so.  The header ``include/vllm_hip.h`` is the single source of truth: the
prototypes below are parsed from it, so a symbol that is declared but not exported fails at load time."""
import ctypes
import os
import re
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
HEADER = os.path.join(_ROOT, "include", "vllm_hip.h")

_lib = None

_CTYPES = {
    "int": ctypes.c_int, "long": ctypes.c_long, "int64_t": ctypes.c_int64, "float": ctypes.c_float,
    "double": ctypes.c_double, "vllm_stream_t": ctypes.c_void_p, "size_t": ctypes.c_size_t,
}


def lib_path() -> str:
    return os.path.join(_HERE, "_build", "libvllm_hip.so")


def build(force: bool = False, jobs: int = 8) -> str:
    """Compile every HIP source for gfx950 (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-s", "-C", os.path.join(_HERE, "csrc"), f"-j{jobs}"]
    if force:
        cmd.append("-B")
    subprocess.check_call(cmd)
    return lib_path()


def parse_header(path: str = HEADER):
    """-> {name: (restype, [argtypes])} for every ``vllm_*`` function declared in the header."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    src = re.sub(r"//[^\n]*", " ", src)
    protos = {}
    for m in re.finditer(r"\b(int|void|const\s+char\s*\*)\s*(vllm_\w+)\s*\(([^)]*)\)\s*;", src):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        restype = ctypes.c_char_p if "char" in ret else (None if ret == "void" else ctypes.c_int)
        argtypes = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                if "*" in a:
                    argtypes.append(ctypes.c_void_p)
                else:
                    toks = [t for t in a.replace("const", " ").split() if t]
                    argtypes.append(_CTYPES[toks[0]])
        protos[name] = (restype, argtypes)
    return protos


def lib():
    """Load (once) and return the shared library.  Fails loudly when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} is missing: the HIP extension has not been built. Run `python -c \"import __graft_entry__ as g; "
            "g.build()\"` (or `make -C visionllm_amd/csrc`). There is no CPU fallback.")
    L = ctypes.CDLL(path)
    for name, (restype, argtypes) in parse_header().items():
        fn = getattr(L, name)  # AttributeError if a declared symbol is not exported
        fn.restype = restype
        fn.argtypes = argtypes
    if L.vllm_abi_version() != 1:
        raise RuntimeError("libvllm_hip.so ABI version mismatch")
    _lib = L
    return L


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = lib().vllm_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"{what or 'libvllm_hip'} failed ({rc}): {msg}")


def ptr(t):
    """Device pointer of a torch tensor (0 for None / empty)."""
    if t is None:
        return None
    return ctypes.c_void_p(t.data_ptr())


def current_stream(device=None):
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)
