"""The reference's OWN `vilbert/utils.py`, executed under this module name (nothing of it is restated here).

`vilbert.utils` of this package provides `PreTrainedModel` (the model path); every other name of the reference's
utils.py - `tbLogger` (:151-482), `MultiTaskStopOnPlateau` (:39-148), `cached_path`, the S3 helpers - is forwarded to
this module by `vilbert.utils.__getattr__`. It is a real submodule (not an anonymous importlib load) so that the objects
the training scripts pickle into their checkpoints (`torch.save({"tb_logger": tbLogger, "task_stop_controller": ...})`,
train_tasks.py:623-636) can be unpickled by a later process. Importable only when a reference checkout is attached
(vilbert/__init__.py: attach_reference / $VILBERT_REFERENCE_ROOT).
"""
import os as _os

import vilbert as _pkg
from . import _compat as _compat

_dir = _pkg.REFERENCE_PACKAGE_DIR or _pkg.attach_reference()
if not _dir or not _os.path.isfile(_os.path.join(_dir, "utils.py")):
    raise ImportError("no reference checkout attached (set VILBERT_REFERENCE_ROOT): vilbert.utils only provides "
                      "PreTrainedModel on its own; tbLogger / MultiTaskStopOnPlateau / cached_path live in the reference")
_compat.install()                                # boto3 / botocore / tensorboardX / torch._six import names
__file__ = _os.path.join(_dir, "utils.py")
# The source runs with __name__ = "vilbert.utils" - the name it has in the reference - so that its classes record
# `__module__ = "vilbert.utils"`: train_tasks.py:623-636 pickles `tbLogger` / `MultiTaskStopOnPlateau` objects into the
# resume checkpoint, and pickle stores the class as (module, name). "vilbert.utils.tbLogger" resolves in BOTH code bases
# (here through vilbert.utils.__getattr__ to this very object), so resume checkpoints are interchangeable with upstream.
_own_name, __name__ = __name__, "vilbert.utils"
try:
    with open(__file__, "r", encoding="utf-8") as _f:
        exec(compile(_f.read(), __file__, "exec"), globals())
finally:
    __name__ = _own_name
