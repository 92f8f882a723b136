"""MI355X-native ``vilbert`` package: import-compatible with the reference's ``vilbert`` for the model
path (``from vilbert.vilbert import BertConfig, BertForMultiModalPreTraining, VILBertForVLTasks``)."""

# (GPU_MAX_HW_QUEUES is raised by vilbert/distributed.py, i.e. only for data-parallel runs: RCCL's streams otherwise push
# the encoder's side stream onto the main stream's hardware queue. Single-GPU runs keep HIP's default of 4 queues - with
# 8, a whole-step HIP graph with its two-stream fork replays 20 % slower: 1,386 vs 1,730 samples/s at batch 64.)
