"""ORACLE -- test infrastructure only (CPU restatement of the reference's hot path).

Nothing under ``visionllm_amd/`` may import this package.  Allowed importers: ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py``.
"""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))


def build(force: bool = False) -> str:
    """Compile the C restatement (gcc) into oracle/_build/ and return the .so path."""
    so = os.path.join(_HERE, "_build", "libmsda_oracle.so")
    stale = force
    for name in ("msda", "dcnv3"):
        lib, src = os.path.join(_HERE, "_build", f"lib{name}_oracle.so"), os.path.join(_HERE, f"{name}_oracle.c")
        stale = stale or not os.path.exists(lib) or os.path.getmtime(lib) < os.path.getmtime(src)
    if stale:
        subprocess.check_call(["make", "-s", "-C", _HERE] + (["-B"] if force else []))
    return so
