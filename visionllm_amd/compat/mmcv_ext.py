"""``mmcv._ext`` alias for the two MSDA entry points mmcv binds (mmcv/ops/csrc/pytorch/pybind.cpp:788-799; loaded by
``ext_loader.load_ext('_ext', ['ms_deform_attn_backward', 'ms_deform_attn_forward'])``,
mmcv/mmcv/ops/multi_scale_deform_attn.py:19-20, mmcv/mmcv/utils/ext_loader.py:12-16).

    import visionllm_amd.compat.mmcv_ext as e; e.install()      # before ``import mmcv.ops``

registers this module as ``mmcv._ext`` when no compiled ``mmcv._ext`` is importable, so mmcv's own
``MultiScaleDeformableAttnFunction`` (forward returns the output; backward fills three caller-allocated gradients,
multi_scale_deform_attn.py:54-94) runs on the HIP kernels unchanged.  Other mmcv ops are not provided: asking the
loader for them fails with mmcv's own assertion ("<fun> miss in module _ext")."""
import sys

from visionllm_amd.ms_deform_attn import ms_deform_attn_backward_ as _backward_into
from visionllm_amd.ms_deform_attn import ms_deform_attn_forward as _forward


def ms_deform_attn_forward(value, value_spatial_shapes, value_level_start_index, sampling_locations, attention_weights,
                           im2col_step=64):
    return _forward(value, value_spatial_shapes, value_level_start_index, sampling_locations, attention_weights, im2col_step)


def ms_deform_attn_backward(value, value_spatial_shapes, value_level_start_index, sampling_locations, attention_weights,
                            grad_output, grad_value, grad_sampling_loc, grad_attn_weight, im2col_step=64):
    _backward_into(value, value_spatial_shapes, value_level_start_index, sampling_locations, attention_weights, grad_output,
                   grad_value, grad_sampling_loc, grad_attn_weight, im2col_step)


def install(force: bool = False):
    """Register this module as ``mmcv._ext`` (returns the module that ends up registered)."""
    if not force and "mmcv._ext" in sys.modules:
        return sys.modules["mmcv._ext"]
    sys.modules["mmcv._ext"] = sys.modules[__name__]
    return sys.modules[__name__]
