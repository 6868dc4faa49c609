"""The handful of config keys the hot path reads, as a plain attribute tree (the reference uses detectron2's CfgNode:
connectomics/config/defaults.py + maskfoermer_config.py:6-211; merged with configs/*/.yaml).  Values below are the
shipped CVPPP-PCTrans settings (configs/CVPPP/CVPPP-PCTrans.yaml:8-51 over CVPPP-PCTrans-Base.yaml)."""
from types import SimpleNamespace as NS

RESNET_SHAPES = {
    # detectron2 build_resnet_backbone output_shape(): res2..res5 channels / strides
    50: {"res2": (256, 4), "res3": (512, 8), "res4": (1024, 16), "res5": (2048, 32)},
    18: {"res2": (64, 4), "res3": (128, 8), "res4": (256, 16), "res5": (512, 32)},
}


def get_cfg(num_queries=100, enc_in_features=("res3", "res4", "res5"), norm="SyncBN", sem_norm="SyncBN",
            dec_layers=10, enc_layers=6, train_num_points=12544, dataset="CVPPP"):
    head = NS(NAME="MaskFormerHead", IGNORE_VALUE=0, NUM_CLASSES=2, LOSS_WEIGHT=1.0, CONVS_DIM=128, MASK_DIM=16,
              NORM=norm, PIXEL_DECODER_NAME="MSDeformAttnPixelDecoder",
              IN_FEATURES=["res2", "res3", "res4", "res5"],
              DEFORMABLE_TRANSFORMER_ENCODER_IN_FEATURES=list(enc_in_features), COMMON_STRIDE=4,
              TRANSFORMER_ENC_LAYERS=enc_layers, ATTENTION_MASK_THRESHOLD=0.5)
    mf = NS(SEMANTIC_LOSS_ON=True, SEMANTIC_NORM=sem_norm, TRANSFORMER_DECODER_NAME="MultiScaleMaskedTransformerDecoder",
            TRANSFORMER_IN_FEATURE="multi_scale_pixel_decoder", HIDDEN_DIM=128, NUM_OBJECT_QUERIES=num_queries,
            NHEADS=8, DROPOUT=0.0, DIM_FEEDFORWARD=1024, PRE_NORM=False, ENFORCE_INPUT_PROJ=False,
            DEC_LAYERS=dec_layers, POSITION_POINTS_NUM=1, REL_COORD=True,
            # loss / test keys (configs/CVPPP/CVPPP-PCTrans.yaml:25-51, config/maskfoermer_config.py)
            DEEP_SUPERVISION=True, NO_OBJECT_WEIGHT=0.1, CLASS_WEIGHT=2.0, MASK_WEIGHT=5.0, DICE_WEIGHT=5.0,
            SEM_WEIGHT=5.0, EMB_WEIGHT=1.0, REID_WEIGHT_QUERY=1.0, REID_WEIGHT_MASK=1.0, REF_POINTS_WEIGHT=5.0,
            SIZE_DIVISIBILITY=32, TRAIN_NUM_POINTS=train_num_points, OVERSAMPLE_RATIO=3.0,
            IMPORTANCE_SAMPLE_RATIO=0.75,
            TEST=NS(SEMANTIC_ON=False, INSTANCE_ON=True, PANOPTIC_ON=False, OVERLAP_THRESHOLD=0.8,
                    OBJECT_MASK_THRESHOLD=0.8))
    return NS(MODEL=NS(SEM_SEG_HEAD=head, MASK_FORMER=mf), DATASET=NS(DATA_TYPE=dataset))


def resnet_output_shape(depth=50):
    from .layers import ShapeSpec
    return {k: ShapeSpec(channels=c, stride=s) for k, (c, s) in RESNET_SHAPES[depth].items()}
