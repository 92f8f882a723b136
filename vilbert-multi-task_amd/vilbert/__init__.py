"""MI355X-native ``vilbert`` package: import-compatible with the reference's ``vilbert`` for the model
path (``from vilbert.vilbert import BertConfig, BertForMultiModalPreTraining, VILBertForVLTasks``)."""

import os as _os

# A training step keeps up to four HIP streams busy (text | image encoder streams, each with a weight-gradient side
# stream - autograd_ops.py), plus RCCL's own in data-parallel runs. HIP maps streams round-robin onto
# GPU_MAX_HW_QUEUES (default 4) hardware queues; streams that share a queue serialise and the overlap silently
# disappears (measured, B = 256 step: 2,375 samples/s with 4 queues, 2,422-2,429 with 6-8; B = 64: 1,887 -> 1,913).
# The variable is read when the HIP runtime initialises, i.e. at the first device call - importing this package early
# (the training scripts do) is in time. (A whole-step HIP graph replays faster with 4 queues: vilbert/graphed.py.)
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
