"""Import-name shim: the reference scripts do ``from apex.parallel import DistributedDataParallel as DDP``
(train_concap.py:507-512, train_tasks.py:490-497) and its model tries
``from apex.normalization.fused_layer_norm import FusedLayerNorm`` (vilbert.py:297-298). NVIDIA apex does
not exist on ROCm images; these names resolve to the MI355X-native implementations instead."""
