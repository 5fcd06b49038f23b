"""Stand-in for the reference's compiled CUDA extension `MultiScaleDeformableAttention` (OPS/src/vision.cpp:14-17 exports exactly
these two functions; OPS/functions/ms_deform_attn_func.py:21-29 imports the module by this name and OPS/make.sh builds it with
nvcc).  Put this directory on `sys.path` (or copy the file next to the reference's `ops/`) and the reference's own
`MSDeformAttnFunction` / `MSDeformAttn` run -- forward and backward -- on libaxvs.so's HIP kernels:

    import sys, axial_vs_amd, os
    sys.path.insert(0, os.path.join(os.path.dirname(axial_vs_amd.__file__), "compat"))

Signatures as in OPS/src/ms_deform_attn.h:24-67:
    ms_deform_attn_forward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step) -> output
    ms_deform_attn_backward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_output, im2col_step)
        -> (grad_value, grad_sampling_loc, grad_attn_weight)
"""
from axial_vs_amd.msda import ms_deform_attn_backward, ms_deform_attn_forward  # noqa: F401

__all__ = ["ms_deform_attn_forward", "ms_deform_attn_backward"]
