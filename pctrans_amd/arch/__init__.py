from .maskformer import MaskFormer  # noqa: F401
