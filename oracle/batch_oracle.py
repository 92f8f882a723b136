"""TEST INFRASTRUCTURE - CPU restatement of the batch finishing the reference does per pre-training step.
Never imported by the product path (vilbert-multi-task_amd/); only tests/ use it.

Follows /root/reference/vilbert/datasets/concept_cap_dataset.py:241-282 (ConceptCapLoaderTrain.__iter__:
global mean-region feature, [0,0,1,1,1] box, mask column) and /root/reference/train_concap.py:535-540
(objective 1: labels of aligned pairs are dropped). Pinned: tests/test_input_pipeline.py executes the
reference's own ``__iter__`` source (extracted from the reference file with ``ast``; the module itself
cannot be imported - tensorpack / lmdb are absent) and the reference's label-edit lines beside this
restatement, and tests/golden/concap_batch.npz holds outputs produced that way.
"""
import numpy as np

RAW_FIELDS = ("input_ids", "input_mask", "segment_ids", "lm_label_ids", "is_next", "image_feat", "image_loc",
              "image_target", "image_label", "image_mask", "masked_label")


def make_raw_batch(batch, tokens=36, regions=36, feat_dim=2048, n_classes=1601, vocab=30522, seed=0):
    """A raw worker batch with the loader's conventions (concept_cap_dataset.py:430-520,608-670): ragged
    numbers of boxes, padded regions zero, ~15 % masked tokens / regions, overlap-masked regions."""
    g = np.random.RandomState(seed)
    n_box = g.randint(min(10, regions), regions + 1, size=batch)
    box_ok = np.arange(regions)[None, :] < n_box[:, None]
    feat = (g.rand(batch, regions, feat_dim).astype(np.float32) * 2.0) * box_ok[:, :, None]
    loc = (g.rand(batch, regions, 5).astype(np.float32)) * box_ok[:, :, None]
    image_mask = box_ok.astype(np.int64)
    image_label = np.where((g.rand(batch, regions) < 0.15) & box_ok, 1, -1).astype(np.int64)
    masked_label = ((image_label == 1) | ((g.rand(batch, regions) < 0.1) & box_ok)).astype(np.int64)
    masked_label[0, :] = 1                         # a sample whose count of unmasked regions is 0 (-> divisor 1)
    feat = feat * (g.rand(batch, regions) > 0.1 * (image_label == 1))[:, :, None].astype(np.float32)
    n_tok = g.randint(min(8, tokens), tokens + 1, size=batch)
    tok_ok = np.arange(tokens)[None, :] < n_tok[:, None]
    ids = g.randint(0, vocab, size=(batch, tokens)).astype(np.int64) * tok_ok
    lm = np.where((g.rand(batch, tokens) < 0.15) & tok_ok, ids, -1).astype(np.int64)
    lm[1, 2] = 0                                   # a genuine label 0: the objective-1 edit turns it into -1
    target = g.rand(batch, regions, n_classes).astype(np.float32)
    target /= target.sum(-1, keepdims=True)
    return dict(input_ids=ids, input_mask=tok_ok.astype(np.int64), segment_ids=np.zeros((batch, tokens), np.int64),
                lm_label_ids=lm, is_next=g.randint(0, 2, size=batch).astype(np.int64), image_feat=feat,
                image_loc=loc, image_target=target, image_label=image_label, image_mask=image_mask,
                masked_label=masked_label)


def finish_batch(raw, objective=0):
    """Returns the dict of the ten arrays the training loop hands to the model."""
    feat, masked_label = raw["image_feat"], raw["masked_label"]
    batch_size = feat.shape[0]
    # concept_cap_dataset.py:249-256
    sum_count = np.sum(masked_label == 0, axis=1, keepdims=True)
    sum_count[sum_count == 0] = 1
    g_feat = np.sum(feat, axis=1) / sum_count                   # float32 / int64 -> float64
    image_feat = np.array(np.concatenate([g_feat[:, None, :], feat], axis=1), dtype=np.float32)
    # :258-265
    g_loc = np.repeat(np.array([[0, 0, 1, 1, 1]], dtype=np.float32), batch_size, axis=0)
    image_loc = np.array(np.concatenate([g_loc[:, None, :], raw["image_loc"]], axis=1), dtype=np.float32)
    # :266-267
    image_mask = np.concatenate([np.ones((batch_size, 1), dtype=raw["image_mask"].dtype), raw["image_mask"]], axis=1)
    image_label, lm = raw["image_label"].copy(), raw["lm_label_ids"].copy()
    if objective == 1:
        # train_concap.py:535-540
        keep = (raw["is_next"] == 0).astype(np.int64)[:, None]
        image_label = image_label * keep
        image_label[image_label == 0] = -1
        lm = lm * keep
        lm[lm == 0] = -1
    return dict(input_ids=raw["input_ids"], input_mask=raw["input_mask"], segment_ids=raw["segment_ids"],
                lm_label_ids=lm, is_next=raw["is_next"], image_feat=image_feat, image_loc=image_loc,
                image_target=raw["image_target"], image_label=image_label, image_mask=image_mask)


OUT_FIELDS = ("input_ids", "input_mask", "segment_ids", "lm_label_ids", "is_next", "image_feat", "image_loc",
              "image_target", "image_label", "image_mask")
