"""Import the REAL reference model (read-only tree at /root/reference) under an alias.

TEST INFRASTRUCTURE ONLY. Works only where /root/reference exists (the build container);
`available()` is False on the GPU box, where the committed fixtures in tests/golden/ and the
restatement in oracle/vilbert_oracle.py take over.

Why an alias: the product package is itself importable as ``vilbert`` (drop-in for the
reference's ``from vilbert.vilbert import ...``), so the reference package is loaded as
``vilbert_reference`` via importlib; its only intra-package import is the relative
``from .utils import PreTrainedModel`` (reference vilbert/vilbert.py:23), which survives the rename.

Why stubs: reference vilbert/utils.py:19-28 imports boto3, botocore.exceptions, tensorboardX and
torch._six at module scope; none is used by the model path.
"""
import importlib.util
import math
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("VILBERT_REFERENCE_ROOT", "/root/reference")
_ALIAS = "vilbert_reference"


def available():
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "vilbert", "vilbert.py"))


def _install_stubs():
    def stub(name, **attrs):
        if name in sys.modules:
            return sys.modules[name]
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m

    stub("boto3")
    stub("botocore")
    stub("botocore.exceptions", ClientError=Exception)
    stub("tensorboardX", SummaryWriter=object)
    stub("torch._six", inf=math.inf)


def load():
    """Return the reference's vilbert.vilbert module (imported as vilbert_reference.vilbert)."""
    if not available():
        raise RuntimeError("reference tree not present at %s" % REFERENCE_ROOT)
    name = _ALIAS + ".vilbert"
    if name in sys.modules:
        return sys.modules[name]
    _install_stubs()
    pkg_dir = os.path.join(REFERENCE_ROOT, "vilbert")
    spec = importlib.util.spec_from_file_location(
        _ALIAS, os.path.join(pkg_dir, "__init__.py"), submodule_search_locations=[pkg_dir]
    )
    pkg = importlib.util.module_from_spec(spec)
    sys.modules[_ALIAS] = pkg
    spec.loader.exec_module(pkg)
    # The reference tries `from apex.normalization.fused_layer_norm import FusedLayerNorm` first
    # (vilbert.py:297-298). The product ships an `apex` import shim that resolves to the HIP LayerNorm;
    # the ORACLE must use the reference's own pure-Python BertLayerNorm (:304-317), so apex is made
    # unimportable while the reference module is executed.
    saved = {k: sys.modules.pop(k) for k in list(sys.modules) if k == "apex" or k.startswith("apex.")}
    sys.modules["apex"] = None
    try:
        mod = importlib.import_module(name)
    finally:
        del sys.modules["apex"]
        sys.modules.update(saved)
    return mod


def config_path(name):
    return os.path.join(REFERENCE_ROOT, "config", name)
