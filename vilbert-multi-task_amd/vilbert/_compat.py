"""Import-name placeholders that let the reference's UNMODIFIED scripts import on an image that lacks their data /
logging dependencies. Nothing here is on the hot path and nothing here re-implements the reference.

`train_concap.py:24-31`, `train_tasks.py:14-47`, `vilbert/task_utils.py:16-20`, `vilbert/datasets/*.py` and
`vilbert/utils.py:19-28` import at module scope: tensorboardX, easydict, tensorpack, lmdb, msgpack_numpy, h5py,
jsonlines / json_lines, boto3, botocore, torch._six. `install()` registers a placeholder for each of those that is NOT
importable here (an installed package always wins):

* functional where the real thing is a few lines (`easydict.EasyDict`, `jsonlines.open`, `torch._six.inf`,
  a `tensorboardX.SummaryWriter` that forwards to `torch.utils.tensorboard` when TensorBoard is installed and otherwise
  drops the scalars);
* for the storage libraries (lmdb, h5py, tensorpack, msgpack_numpy, boto3) a module whose every attribute raises
  `ImportError` naming the missing package at the moment it is USED - importing the dataset modules works, opening an
  LMDB without lmdb installed fails loudly.
"""
import importlib
import importlib.machinery
import importlib.util
import json
import math
import sys
import types


def _missing(name):
    if name in sys.modules:
        return sys.modules[name] is None
    try:
        return importlib.util.find_spec(name) is None
    except (ImportError, ValueError, AttributeError):
        return True


class _Unavailable(types.ModuleType):
    """Stands in for a package that is not installed: import succeeds, any use raises."""

    def __init__(self, name):
        super().__init__(name)
        self.__path__ = []              # a package, so that `import pkg.sub` is looked up in sys.modules
        self.__placeholder__ = True
        # other libraries probe with importlib.util.find_spec(name), which raises on a module without a spec
        self.__spec__ = importlib.machinery.ModuleSpec(name, None)

    def __getattr__(self, attr):
        if attr.startswith("__"):
            raise AttributeError(attr)
        name = self.__name__

        def _raise(*a, **k):
            raise ImportError("%s.%s was called but the package '%s' is not installed in this image (vilbert._compat "
                              "only provides the import name)" % (name, attr, name.split(".")[0]))
        _raise.__name__ = attr
        return _raise


class EasyDict(dict):
    """Attribute-access dict, recursive like easydict.EasyDict (train_tasks.py:18,213: `edict(yaml.safe_load(f))`)."""

    def __init__(self, d=None, **kwargs):
        super().__init__()
        for k, v in dict(d or {}, **kwargs).items():
            self[k] = v

    @classmethod
    def _wrap(cls, v):
        if isinstance(v, dict) and not isinstance(v, cls):
            return cls(v)
        if isinstance(v, (list, tuple)):
            return type(v)(cls._wrap(x) for x in v)
        return v

    def __setitem__(self, k, v):
        super().__setitem__(k, self._wrap(v))

    __setattr__ = __setitem__

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __delattr__(self, k):
        try:
            del self[k]
        except KeyError:
            raise AttributeError(k)

    def update(self, d=None, **kwargs):
        for k, v in dict(d or {}, **kwargs).items():
            self[k] = v


class _JsonLinesReader(object):
    def __init__(self, fp):
        self._fp = fp

    def __iter__(self):
        for line in self._fp:
            line = line.strip()
            if line:
                yield json.loads(line)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self._fp.close()

    def close(self):
        self._fp.close()


def _jsonlines_open(path, mode="r", **kwargs):
    if "r" not in mode:
        raise ImportError("jsonlines is not installed; vilbert._compat only reads")
    return _JsonLinesReader(open(path, "r", encoding="utf-8"))


class _NullSummaryWriter(object):
    """tensorboardX.SummaryWriter's surface as vilbert/utils.py:169,208-217 uses it; scalars are dropped."""

    def __init__(self, *args, **kwargs):
        pass

    def add_scalar(self, *args, **kwargs):
        pass

    def add_scalars(self, *args, **kwargs):
        pass

    def flush(self):
        pass

    def close(self):
        pass


def _summary_writer():
    try:
        from torch.utils.tensorboard import SummaryWriter   # needs the tensorboard package
        return SummaryWriter
    except Exception:
        return _NullSummaryWriter


def _module(name, **attrs):
    m = types.ModuleType(name)
    m.__placeholder__ = True
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    for k, v in attrs.items():
        setattr(m, k, v)
    return m


def install():
    """Register the placeholders (idempotent). Returns the list of names that were filled in."""
    made = []

    def put(name, factory):
        if _missing(name):
            sys.modules[name] = factory()
            made.append(name)

    put("easydict", lambda: _module("easydict", EasyDict=EasyDict))
    put("jsonlines", lambda: _module("jsonlines", open=_jsonlines_open, Reader=_JsonLinesReader))
    put("json_lines", lambda: _module("json_lines", reader=_JsonLinesReader, open=_jsonlines_open))
    put("tensorboardX", lambda: _module("tensorboardX", SummaryWriter=_summary_writer()))
    if "torch._six" not in sys.modules and _missing("torch._six"):
        sys.modules["torch._six"] = _module("torch._six", inf=math.inf, string_classes=(str, bytes))
        made.append("torch._six")
    for name in ("lmdb", "h5py", "msgpack_numpy", "boto3", "tensorpack", "cv2"):
        put(name, lambda name=name: _Unavailable(name))
    if getattr(sys.modules.get("msgpack_numpy"), "__placeholder__", False):
        # concept_cap_dataset.py:27 calls msgpack_numpy.patch() at import; without the package there is nothing to patch
        # (decoding a numpy-carrying record later fails in msgpack itself)
        sys.modules["msgpack_numpy"].patch = lambda: None
    if getattr(sys.modules.get("tensorpack"), "__placeholder__", False):
        put("tensorpack.dataflow", lambda: _Unavailable("tensorpack.dataflow"))
        # `class BertPreprocessBatch(td.RNGDataFlow)` (concept_cap_dataset.py) subclasses at import time
        td = sys.modules["tensorpack.dataflow"]
        td.RNGDataFlow = type("RNGDataFlow", (object,), {"reset_state": lambda self: None})
        td.DataFlow = type("DataFlow", (object,), {})
        td.ProxyDataFlow = type("ProxyDataFlow", (object,), {})
        sys.modules["tensorpack"].dataflow = td
    # refer_expression_dataset.py:16 `from tools.refer.refer import REFER`: an un-vendored git submodule of the
    # reference (tools/refer -> lichengunc/refer); absent unless the checkout was cloned recursively
    if _missing("tools.refer.refer"):
        m = _Unavailable("tools.refer.refer")
        m.REFER = m.__getattr__("REFER")
        sys.modules["tools.refer.refer"] = m
        made.append("tools.refer.refer")
    if _missing("botocore"):
        sys.modules["botocore"] = _module("botocore")
        sys.modules["botocore.exceptions"] = _module("botocore.exceptions", ClientError=type("ClientError", (Exception,), {}))
        sys.modules["botocore"].exceptions = sys.modules["botocore.exceptions"]
        made.append("botocore")
    return made
