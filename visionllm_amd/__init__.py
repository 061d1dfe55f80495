"""visionllm_amd -- MI355X (gfx950) native image -> visual-token hot path of VisionLLMv2.

Host-side mirror of the reference's operator/module interfaces for that path only:

* ``ms_deform_attn``  -- ``ms_deform_attn_forward/backward``, ``MSDeformAttnFunction``, ``MSDeformAttn``,
  ``MultiScaleDeformableAttention`` (mmcv) and ``GroundingDinoMultiscaleDeformableAttention`` (HF) mirrors.
* ``intern_vit`` / ``clip_vit`` -- drop-in ``vis_encoder`` modules (same parameter names as the reference).
* ``bridge``          -- ``pixel_shuffle`` + ``vl_bridge`` projector.
* ``dist``            -- data-parallel sharding of images and the RCCL all-gather of visual tokens.

All compute goes through ``libvllm_hip.so`` (hand-written HIP, C ABI in ``include/vllm_hip.h``).  There is NO
CPU fallback: calling an op without the built extension, or with CPU tensors, raises.
"""
from ._lib import build, lib, lib_path  # noqa: F401

__version__ = "0.1.0"
