"""Generates tests/golden/concap_batch.npz: a small raw Conceptual-Captions batch and the outputs of the
REFERENCE's own finishing code on it (loader __iter__ + the objective-1 label edit), see
tests/ref_loader_source.py. Run in the build container:  python tests/golden/make_concap_golden.py"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import batch_oracle as bo  # noqa: E402
import ref_loader_source as rls  # noqa: E402


def main():
    raw = bo.make_raw_batch(6, tokens=9, regions=7, feat_dim=16, n_classes=5, vocab=50, seed=11)
    tup = tuple(raw[n].copy() for n in bo.RAW_FIELDS) + (np.arange(6),)
    out = rls.reference_loader_iter([tup])[0]
    ref = {n: out[i].numpy() for i, n in enumerate(bo.OUT_FIELDS)}
    il1, lm1 = rls.reference_objective1_edit(torch.tensor(raw["image_label"]), torch.tensor(raw["lm_label_ids"]),
                                             torch.tensor(raw["is_next"]))
    save = {"raw_" + k: v for k, v in raw.items()}
    save.update({"ref_" + k: v for k, v in ref.items()})
    save.update(ref_obj1_image_label=il1.numpy(), ref_obj1_lm_label_ids=lm1.numpy())
    np.savez_compressed(os.path.join(HERE, "concap_batch.npz"), **save)
    print("wrote concap_batch.npz", {k: v.shape for k, v in save.items() if k.startswith("ref_")})


if __name__ == "__main__":
    main()
