"""Import-name shim (see tokenization_bert.py): ``from pytorch_transformers.tokenization_roberta import RobertaTokenizer``."""
try:
    from transformers import RobertaTokenizer  # noqa: F401
except Exception as _e:                         # pragma: no cover
    _err = _e

    class RobertaTokenizer(object):
        @classmethod
        def from_pretrained(cls, *args, **kwargs):
            raise ImportError("neither pytorch_transformers nor transformers is installed: %r" % (_err,))
