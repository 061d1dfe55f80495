"""Module-name shim: ``import MultiScaleDeformableAttention as MSDA`` (the pybind module built from
VisionLLMv2/visionllmv2/model/unipose/ops/src/vision.cpp:13-16) resolves to the HIP implementation when
``visionllm_amd/compat`` is on ``sys.path``.  Exposes exactly the two functions of that module."""
from visionllm_amd.ms_deform_attn import ms_deform_attn_backward, ms_deform_attn_forward  # noqa: F401
