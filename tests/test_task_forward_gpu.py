"""GPU side of row g3 / f4: what the reference's multi-task trainer does with the model - `ForwardModelsTrain`
(/root/reference/vilbert/task_utils.py:167-374: per-process batch reshaping, task token, model call, per-type loss) -
through the HIP `VILBertForVLTasks`, against the same arithmetic through the CPU oracle.

The arithmetic is oracle/task_forward_oracle.py, which tests/test_reference_scripts.py pins to the REAL function in the
build container; here both sides go through it, only the model differs. Shapes are vilbert_tasks.yml's
(max_seq_length / max_region_num of each task), the tasks are those of BASELINE.json configs[3] ("tasks 1-2-4-7-8") plus
NLVR2's paired-image layout. Loss, score and every parameter gradient (dropout forced off) are compared."""
import pytest
import torch

import helpers
from oracle import synth, task_forward_oracle as tf, vilbert_oracle as vo

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
# task, batch, tokens, regions (vilbert_tasks.yml max_seq_length / max_region_num), task_specific_tokens
CASES = [("TASK1", 6, 23, 101, True),      # VQA, 3129 soft labels
         ("TASK2", 4, 26, 101, False),     # GenomeQA
         ("TASK4", 4, 20, 200, True),      # Visual7w pointing: 200 regions, vision_logit[:, 101:] gathered by choice ids
         ("TASK7", 3, 30, 101, True),      # retrieval COCO: 4 (image, caption) pairs per sample -> 12 model rows
         ("TASK8", 2, 30, 101, False),     # retrieval Flickr30k
         ("TASK12", 4, 40, 101, True)]     # NLVR2: two images per statement, pooled pairs -> vil_binary_prediction


@pytest.mark.parametrize("task_id,batch,n_tok,n_reg,task_tokens", CASES, ids=[c[0] for c in CASES])
def test_forward_models_train_through_the_hip_model_matches_the_oracle(task_id, batch, n_tok, n_reg, task_tokens):
    import vilbert.vilbert as V
    from vilbert.vilbert import BertConfig, VILBertForVLTasks

    cfg = synth.load_config("bert_base_2layer_2conect.json")
    cfg.update(v_target_size=1601, task_specific_tokens=task_tokens)
    num_labels = 3129
    sd = synth.make_state_dict(cfg, "vltasks", seed=21)
    assert sd["vil_prediction.logit_fc.3.weight"].shape[0] == num_labels
    data = tf.make_task_batch(task_id, batch, n_tok, n_reg, num_labels=num_labels, seed=31)

    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items() if k != "cls.predictions.decoder.weight"}
    leaves["cls.predictions.decoder.weight"] = leaves["bert.embeddings.word_embeddings.weight"]

    def oracle_model(*a):
        return vo.vltasks_forward(leaves, cfg, *a[:7], task_ids=a[7] if task_tokens else None)
    want_loss, want_score = tf.forward_train(task_id, data, oracle_model)
    want_loss.backward()

    orig = V._drop_p
    V._drop_p = lambda m: 0.0
    try:
        net = VILBertForVLTasks(BertConfig.from_dict(cfg), num_labels=num_labels)
        net.load_state_dict(sd)
        net = net.to(DEV).train()
        got_loss, got_score = tf.forward_train(task_id, helpers.to_device(data, DEV), net)
        got_loss.backward()
        torch.cuda.synchronize()
    finally:
        V._drop_p = orig

    assert abs(got_loss.item() - want_loss.item()) <= 1e-4 * max(1.0, abs(want_loss.item())), (got_loss.item(), want_loss.item())
    assert abs(float(got_score) - float(want_score)) <= 1e-6
    gmax = max(v.grad.abs().max().item() for v in leaves.values() if v.grad is not None)
    seen = 0
    for name, p in net.named_parameters():
        ref = leaves[name].grad
        if ref is None or ref.abs().max().item() == 0.0:
            assert p.grad is None or p.grad.abs().max().item() <= 2e-7 * gmax, name
            continue
        err = (p.grad.cpu().double() - ref.double()).abs().max().item()
        # (+ 5e-7: a gradient that is zero in exact arithmetic - vil_logit.bias under the retrieval cross-entropy, whose
        # softmax gradients sum to 0 over the options - is pure fp32 rounding noise of an O(1) sum on both sides)
        bound = 2e-4 * ref.abs().max().item() + 2e-7 * gmax + 5e-7
        assert err <= bound, "%s %s: grad err %.3e > %.3e" % (task_id, name, err, bound)
        seen += 1
    assert seen > 100
