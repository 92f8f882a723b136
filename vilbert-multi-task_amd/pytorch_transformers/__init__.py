"""Import-name shim for the one thing the reference's training scripts take from pytorch-transformers:
``from pytorch_transformers.optimization import AdamW, WarmupLinearSchedule, WarmupConstantSchedule``
(train_concap.py:27, train_tasks.py:26-30) - resolved to the MI355X-native optimizer (vilbert/optim.py)."""
