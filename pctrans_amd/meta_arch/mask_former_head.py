"""MaskFormerHead: pixel decoder -> transformer decoder.

API mirror of meta_arch/mask_former_head.py of the reference: ctor kwargs (:44-58), `from_config` keys (:88-115),
`forward(features, targets=None, mask=None, criterion=None) -> (predictions, mask_features)` (:117-154), sub-module
names `pixel_decoder` / `predictor` (state-dict prefix `sem_seg_head.` in the meta-arch), and the v1 -> v2 state-dict
key migration (:22-42).  detectron2's registries are replaced by the two name tables below.
"""
import logging
from typing import Dict

from torch import nn

from ..layers import ShapeSpec
from ..pixel_decoder.msdeformattn import MSDeformAttnPixelDecoder
from ..transformer_decoder.mask2former_transformer_decoder import MultiScaleMaskedTransformerDecoder

PIXEL_DECODERS = {"MSDeformAttnPixelDecoder": MSDeformAttnPixelDecoder}
TRANSFORMER_DECODERS = {"MultiScaleMaskedTransformerDecoder": MultiScaleMaskedTransformerDecoder}


def build_pixel_decoder(cfg, input_shape):
    """pixel_decoder/fpn.py:21-33 of the reference: look the class up by cfg name; it must have forward_features."""
    name = cfg.MODEL.SEM_SEG_HEAD.PIXEL_DECODER_NAME
    cls = PIXEL_DECODERS[name]
    model = cls(**cls.from_config(cfg, input_shape))
    if not callable(getattr(model, "forward_features", None)):
        raise ValueError(f"Only SEM_SEG_HEADS with forward_features method can be used as pixel decoder. "
                         f"Please implement forward_features for {name} to only return mask features.")
    return model


def build_transformer_decoder(cfg, in_channels, mask_classification=True):
    """transformer_decoder/maskformer_transformer_decoder.py:21-27 of the reference."""
    cls = TRANSFORMER_DECODERS[cfg.MODEL.MASK_FORMER.TRANSFORMER_DECODER_NAME]
    return cls(**cls.from_config(cfg, in_channels, mask_classification))


class MaskFormerHead(nn.Module):
    _version = 2

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                              error_msgs):
        version = local_metadata.get("version", None)
        if version is None or version < 2:
            moved = False
            for k in list(state_dict.keys()):
                if "sem_seg_head" in k and not k.startswith(prefix + "predictor"):
                    state_dict[k.replace(prefix, prefix + "pixel_decoder.")] = state_dict.pop(k)
                    moved = True
            if moved:
                logging.getLogger(__name__).warning(
                    f"Weight format of {self.__class__.__name__} have changed! Applying automatic conversion now ...")
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                                      error_msgs)

    def __init__(self, input_shape: Dict[str, ShapeSpec], *, num_classes: int, pixel_decoder: nn.Module,
                 loss_weight: float = 1.0, ignore_value: int = -1, transformer_predictor: nn.Module,
                 transformer_in_feature: str, attn_mask_threshold: float = 0.5):
        super().__init__()
        self.in_features = [k for k, _ in sorted(input_shape.items(), key=lambda kv: kv[1].stride)]
        self.ignore_value = ignore_value
        self.common_stride = 4
        self.loss_weight = loss_weight
        self.pixel_decoder = pixel_decoder
        self.predictor = transformer_predictor
        self.transformer_in_feature = transformer_in_feature
        self.num_classes = num_classes
        self.attn_mask_threshold = attn_mask_threshold

    @classmethod
    def from_config(cls, cfg, input_shape: Dict[str, ShapeSpec]):
        head, mf = cfg.MODEL.SEM_SEG_HEAD, cfg.MODEL.MASK_FORMER
        if mf.TRANSFORMER_IN_FEATURE in ("transformer_encoder", "multi_scale_pixel_decoder"):
            in_channels = head.CONVS_DIM
        elif mf.TRANSFORMER_IN_FEATURE == "pixel_embedding":
            in_channels = head.MASK_DIM
        else:
            in_channels = input_shape[mf.TRANSFORMER_IN_FEATURE].channels
        return dict(
            input_shape={k: v for k, v in input_shape.items() if k in head.IN_FEATURES},
            ignore_value=head.IGNORE_VALUE, num_classes=head.NUM_CLASSES,
            pixel_decoder=build_pixel_decoder(cfg, input_shape), loss_weight=head.LOSS_WEIGHT,
            transformer_in_feature=mf.TRANSFORMER_IN_FEATURE,
            transformer_predictor=build_transformer_decoder(cfg, in_channels, mask_classification=True),
            attn_mask_threshold=head.ATTENTION_MASK_THRESHOLD)

    def forward(self, features, targets=None, mask=None, criterion=None):
        return self.layers(features, targets, mask, criterion)

    def layers(self, features, targets=None, mask=None, criterion=None):
        mask_features, transformer_encoder_features, multi_scale_features = \
            self.pixel_decoder.forward_features(features)
        if self.transformer_in_feature == "multi_scale_pixel_decoder":
            predictions = self.predictor(multi_scale_features, targets, mask_features, mask,
                                         self.attn_mask_threshold, criterion)
        elif self.transformer_in_feature == "transformer_encoder":
            assert transformer_encoder_features is not None, "Please use the TransformerEncoderPixelDecoder."
            predictions = self.predictor(transformer_encoder_features, mask_features, mask)
        elif self.transformer_in_feature == "pixel_embedding":
            predictions = self.predictor(mask_features, mask_features, mask)
        else:
            predictions = self.predictor(features[self.transformer_in_feature], mask_features, mask)
        return predictions, mask_features
