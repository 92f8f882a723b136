"""Generate tests/golden/*.npz by running the REAL reference model (build container only).

    python tests/golden/make_golden.py [case ...]      (no arguments = every case)

Imports /root/reference/vilbert/vilbert.py through oracle/ref_loader.py (alias + import stubs), loads
the seeded state dict from oracle/synth.py, runs the reference forward on CPU in eval mode and stores
its outputs (fp32). Nothing under /root/reference is copied; only output tensors are saved.
"""
import sys

import numpy as np
import torch

import cases
from oracle import ref_loader


def main():
    ref = ref_loader.load()
    torch.set_grad_enabled(False)
    only = sys.argv[1:]
    for case, c in cases.CASES.items():
        if only and case not in only:
            continue
        cfg, sd, x = cases.case_inputs(case)
        rc = ref.BertConfig.from_dict(cfg)
        model = ref.VILBertForVLTasks(rc, num_labels=1) if c["kind"] == "vltasks" \
            else ref.BertForMultiModalPreTraining(rc)
        res = model.load_state_dict(sd, strict=True)
        assert not res.missing_keys and not res.unexpected_keys
        model.eval()
        out = model(*cases.forward_args(case, x))
        names = cases.output_names(case)
        blob = {n: cases.sample(case, n, out[i]).contiguous().numpy().astype(np.float32)
                for i, n in enumerate(names)}
        np.savez_compressed(cases.path(case), **blob)
        print(case, {k: v.shape for k, v in blob.items()})


if __name__ == "__main__":
    main()
