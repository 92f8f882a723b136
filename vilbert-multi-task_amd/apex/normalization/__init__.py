from .fused_layer_norm import FusedLayerNorm  # noqa: F401
