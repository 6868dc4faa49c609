"""pctrans_amd -- MI355X (gfx950) native implementation of PCTrans' Mask2Former-style decoder hot path.

Layout mirrors the reference's `connectomics/model/maskformer_block/` so that its classes can be swapped in by
import path:

    pctrans_amd.MultiScaleDeformableAttention        <- the pybind extension module of ops/src/vision.cpp
    pctrans_amd.pixel_decoder.ops.functions          <- ops/functions/ms_deform_attn_func.py
    pctrans_amd.pixel_decoder.ops.modules            <- ops/modules/ms_deform_attn.py
    pctrans_amd.csrc/                                <- hand-written HIP kernels + the C ABI (include/pctrans_hip.h)

The compute path is libpctrans_hip.so; there is no CPU or eager-PyTorch fallback for device tensors.
"""
__version__ = "0.1.0"
