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
    """In-tree build; VLLM_HIP_LIB names another build of the same ABI (A/B timing of two builds in one session)."""
    return os.environ.get("VLLM_HIP_LIB") or os.path.join(_HERE, "_build", "libvllm_hip.so")


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
    for m in re.finditer(r"\b(int|long|void|const\s+char\s*\*)\s*(vllm_\w+)\s*\(([^)]*)\)\s*;", src):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        restype = (ctypes.c_char_p if "char" in ret else None if ret == "void"
                   else ctypes.c_long if ret == "long" else ctypes.c_int)
        argtypes = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                if "char" in a and "*" in a:
                    argtypes.append(ctypes.c_char_p)
                elif "*" in a:
                    argtypes.append(ctypes.c_void_p)
                else:
                    toks = [t for t in a.replace("const", " ").split() if t]
                    argtypes.append(_CTYPES[toks[0]])
        protos[name] = (restype, argtypes)
    return protos


def header_abi_version(path: str = HEADER) -> int:
    """VLLM_ABI_VERSION of include/vllm_hip.h (the mirrors below are written against THIS header)."""
    m = re.search(r"^#define\s+VLLM_ABI_VERSION\s+(\d+)", open(path).read(), flags=re.M)
    if not m:
        raise RuntimeError("include/vllm_hip.h: VLLM_ABI_VERSION not found")
    return int(m.group(1))


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
    # Load torch FIRST: the extension must bind to the HIP runtime (libamdhip64) that PyTorch-ROCm ships and has
    # initialised.  Loading ours first pulls in /opt/rocm's copy and the two runtimes then disagree about devices
    # ("no ROCm-capable device is detected" at the first launch).
    import torch  # noqa: F401
    L = ctypes.CDLL(path)
    for name, (restype, argtypes) in parse_header().items():
        fn = getattr(L, name)  # AttributeError if a declared symbol is not exported
        fn.restype = restype
        fn.argtypes = argtypes
    want = header_abi_version()
    if L.vllm_abi_version() != want:
        raise RuntimeError(f"libvllm_hip.so ABI version {L.vllm_abi_version()} != header's VLLM_ABI_VERSION {want}: rebuild "
                           "(make -C visionllm_amd/csrc)")
    _lib = L
    return L


def set_option(name: str, value: int) -> int:
    """Process-wide tuning / test knob (see include/vllm_hip.h: vllm_set_option)."""
    return lib().vllm_set_option(name.encode(), int(value))


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


# ---- ctypes mirrors of the descriptor structs of include/vllm_hip.h (sizes are verified against the library) ----
_P = ctypes.c_void_p


class VllmVitLayer(ctypes.Structure):
    _fields_ = [(n, _P) for n in ("norm1_w", "norm1_b", "qkv_w", "qkv_b", "q_norm_w", "k_norm_w", "proj_w", "proj_b",
                                  "ls1", "norm2_w", "norm2_b", "fc1_w", "fc1_b", "fc2_w", "fc2_b", "ls2",
                                  "qkv_w_ln", "qkv_colsum", "qkv_bias_ln", "fc1_w_ln", "fc1_colsum", "fc1_bias_ln")]


class VllmVitDesc(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int) for n in ("arch", "num_layers", "hidden", "heads", "inter", "patch", "image",
                                            "kpad", "act", "pixel_is_f32")] + \
               [("eps", ctypes.c_float)] + \
               [(n, _P) for n in ("patch_w", "patch_b", "cls", "pos", "pre_ln_w", "pre_ln_b")] + \
               [("layers", ctypes.POINTER(VllmVitLayer))]


class VllmBridgeDesc(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int) for n in ("kind", "depth", "in_features", "out_features", "pixel_shuffle",
                                            "skip_cls")] + \
               [("ln_eps", ctypes.c_float), ("ln_w", _P), ("ln_b", _P), ("w", _P * 4), ("b", _P * 4)]


class VllmMsdaLayerDesc(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in ("d_model", "n_heads", "n_levels", "n_points", "ref_dim",
                                              "use_4d_normalizer", "geometry", "reserved0")] + \
               [(n, _P) for n in ("value_proj_w", "value_proj_b", "sampling_offsets_w", "sampling_offsets_b",
                                  "attention_weights_w", "attention_weights_b", "output_proj_w", "output_proj_b")]


def check_struct_layouts():
    L = lib()
    assert ctypes.sizeof(VllmMsdaLayerDesc) == L.vllm_msda_layer_desc_sizeof(), "VllmMsdaLayerDesc layout mismatch"
    assert ctypes.sizeof(VllmVitDesc) == L.vllm_vit_desc_sizeof(), "VllmVitDesc layout mismatch"
    assert ctypes.sizeof(VllmVitLayer) == L.vllm_vit_layer_sizeof(), "VllmVitLayer layout mismatch"
    assert ctypes.sizeof(VllmBridgeDesc) == L.vllm_bridge_desc_sizeof(), "VllmBridgeDesc layout mismatch"


EPI_BIAS, EPI_GELU, EPI_QUICK_GELU, EPI_RESIDUAL, EPI_EMBED, EPI_F32 = 0, 1, 2, 3, 4, 5
ARCH_INTERNVIT, ARCH_CLIP = 0, 1
BRIDGE_LINEAR, BRIDGE_MLP_GELU, BRIDGE_INTERNVL_MLP = 0, 1, 2

_workspaces = {}


def workspace(device, nbytes, slot=0):
    """Grow-only per-device scratch buffer (torch owns the memory; the library never allocates).  ``slot`` names
    independent buffers for work that runs concurrently on different streams."""
    import torch
    key = (str(device), slot)
    buf = _workspaces.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(int(nbytes) + 256, dtype=torch.uint8, device=device)
        _workspaces[key] = buf
    return buf
