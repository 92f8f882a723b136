from vilbert.vilbert import BertLayerNorm as FusedLayerNorm  # noqa: F401  (same (hidden, eps=...) constructor)
