"""Import-name shim: ``from pytorch_transformers.tokenization_bert import BertTokenizer`` (train_concap.py:26,
task_utils.py:18 and every dataset module). Tokenisation is data side, not rebuilt: the name resolves to the
`transformers` package's BertTokenizer (the successor of pytorch-transformers 1.0.0, same vocabulary files and
`from_pretrained(name, do_lower_case=...)` signature) when that package is installed."""
try:
    from transformers import BertTokenizer  # noqa: F401
except Exception as _e:                      # pragma: no cover - transformers is present in the ROCm image
    _err = _e

    class BertTokenizer(object):
        @classmethod
        def from_pretrained(cls, *args, **kwargs):
            raise ImportError("neither pytorch_transformers nor transformers is installed: %r" % (_err,))
