"""Import-name shim: the reference's `--baseline` switch (train_tasks.py:181-183, eval_tasks.py:136-138) imports
``from pytorch_transformers.modeling_bert import BertConfig`` for its single-stream baseline model
(vilbert/basebert.py - out of scope here). The name resolves to `transformers.BertConfig` when installed."""
from transformers import BertConfig  # noqa: F401
