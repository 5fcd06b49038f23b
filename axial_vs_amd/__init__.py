"""axial_vs_amd -- MI355X-native axial-trajectory attention (Axial-VS / MaXTron hot path).

Python host mirroring the reference nn.Module surface; all compute runs in hand-written HIP kernels
behind the C-ABI of libaxvs.so (include/axvs.h).
"""
from .modules import (AxialTrajectoryAttention5D, PositionEmbeddingSine3D, TemporalAxialTrajectoryAttentionLayer,
                      GraphedForward, TemporalEncoder, TemporalTrajectoryAttentionLayer, TrajectoryAttention,
                      TubeLinkTemporalEncoder, disable_range_check, enable_range_check, invalidate_pack,
                      range_check_report, set_default_dtype, check_status, set_handoff_policy)

from .cross_clip import CrossClipTrackingModule, TubeLinkCrossClipHead
from .pixel_decoder import (MSDeformAttnPixelDecoder, MSDeformAttnTransformerEncoder, MSDeformAttnTransformerEncoderOnly,
                            PositionEmbeddingSine, WithinClipTrackingModule)
from .matching import linear_sum_assignment, match_clips, match_from_embds
from .tube_link import MultiScaleDeformableAxialTrajectoryAttention
from .msda import MSDeformAttn, MSDeformAttnFunction, MSDeformAttnTransformerEncoderLayer, ms_deform_attn_backward, ms_deform_attn_forward

__all__ = ["MultiScaleDeformableAxialTrajectoryAttention", "linear_sum_assignment", "match_from_embds", "match_clips", "WithinClipTrackingModule", "MSDeformAttnPixelDecoder", "MSDeformAttnTransformerEncoder", "MSDeformAttnTransformerEncoderOnly",
           "PositionEmbeddingSine", "CrossClipTrackingModule", "TubeLinkCrossClipHead", "MSDeformAttn", "MSDeformAttnTransformerEncoderLayer", "ms_deform_attn_forward", "ms_deform_attn_backward", "MSDeformAttnFunction", "TrajectoryAttention", "TemporalAxialTrajectoryAttentionLayer", "TemporalTrajectoryAttentionLayer",
           "TemporalEncoder", "TubeLinkTemporalEncoder", "PositionEmbeddingSine3D", "AxialTrajectoryAttention5D",
           "set_default_dtype", "GraphedForward", "invalidate_pack", "enable_range_check", "range_check_report", "disable_range_check", "check_status", "set_handoff_policy"]
