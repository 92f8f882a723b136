"""MI355X-native ``vilbert`` package: import-compatible with the reference's ``vilbert`` for the model
path (``from vilbert.vilbert import BertConfig, BertForMultiModalPreTraining, VILBertForVLTasks``)."""

import os as _os

# The two-stream encoder (text || image HIP streams) needs its streams on DIFFERENT hardware queues. HIP maps streams
# round-robin onto GPU_MAX_HW_QUEUES (default 4) queues; once RCCL has created its own streams the side stream ends up
# sharing a queue with the main stream and the overlap silently disappears (measured: DistributedDataParallel step
# 117.0 ms with 4 queues, 111.5 ms with 8; plain step 110.5 ms). The variable is read when the HIP runtime initialises,
# i.e. at the first device call - importing this package early (the training scripts do) is in time.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
