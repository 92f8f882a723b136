#!/usr/bin/env python
"""Run one of the reference's scripts, UNMODIFIED, on the MI355X-native model:

    cd /path/to/vilbert-multi-task                       # the scripts use relative config/ paths
    python /path/to/vilbert-multi-task_amd/run_reference.py train_concap.py --config_file config/... [its own flags]
    python -m torch.distributed.run --nproc-per-node 8 /path/to/vilbert-multi-task_amd/run_reference.py train_tasks.py ...

Why a launcher: `python train_concap.py` puts the script's own directory in front of PYTHONPATH, so
`from vilbert.vilbert import ...` (train_concap.py:31) would find the reference's pure-PyTorch model again. Started
through this file the import order is: this directory (`vilbert`, `apex`, `pytorch_transformers` of this repository)
first, the reference checkout last; `vilbert.datasets`, `vilbert.task_utils`, `vilbert.optimization`, `vilbert.basebert`
and the logging helpers of `vilbert.utils` fall through to the reference's files (vilbert/__init__.py), and the
third-party names the image lacks get import placeholders (vilbert/_compat.py). The script itself runs through runpy
as `__main__` with its own argv.
"""
import os
import runpy
import sys


def main(argv):
    if not argv or argv[0] in ("-h", "--help"):
        print(__doc__)
        return 2
    script = os.path.abspath(argv[0])
    if not os.path.isfile(script):
        raise SystemExit("run_reference: no such script: %s" % argv[0])
    here = os.path.dirname(os.path.abspath(__file__))
    root = os.environ.setdefault("VILBERT_REFERENCE_ROOT", os.path.dirname(script))
    sys.path[:] = [here] + [p for p in sys.path if os.path.abspath(p or os.getcwd()) not in (here, os.path.abspath(root))]
    sys.path.append(os.path.abspath(root))                # `tools.refer`, `evaluation`: the reference's own top-level dirs
    import vilbert
    if vilbert.attach_reference(root) is None:
        raise SystemExit("run_reference: %s holds no vilbert/ package" % root)
    from vilbert import _compat
    _compat.install()
    sys.argv = [script] + list(argv[1:])
    runpy.run_path(script, run_name="__main__")
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
