"""TEST INFRASTRUCTURE - CPU/GPU-agnostic restatement of how the reference's multi-task trainer calls the model.
Never imported by the product path.

Follows /root/reference/vilbert/task_utils.py:167-374 (`ForwardModelsTrain`): the per-`process` reshaping of a dataset
tuple into the model's eight arguments (:243-309), the task token (:311), the model call (:312-322) and the per-`type`
loss / score (:325-372), with the loss modules of `LoadLosses` (:28-31, 377-390). Only torch ops; `model` is any callable
with `VILBertForVLTasks.forward`'s signature (the HIP model on a GPU, or `oracle.vilbert_oracle.vltasks_forward` bound
to a state dict), so the same function produces both sides of a parity test.

Pinned: tests/test_reference_scripts.py runs the REAL `ForwardModelsTrain` (imported from the reference through the
package's `__path__` fall-through) beside this restatement on the same batches with the same stand-in model, for every
(type, process) pair of the tasks BASELINE.json configs[3] names (TASK1, 2, 4, 7, 8) plus NLVR2 / SNLI-VE / VCR-style.
"""
import torch
import torch.nn as nn

LOSSES = {"BCEWithLogitLoss": nn.BCEWithLogitsLoss(reduction="mean"), "CrossEntropyLoss": nn.CrossEntropyLoss()}

# (type, loss, process) of vilbert_tasks.yml for the tasks of BASELINE configs[3] and the other head types
TASKS = {
    "TASK1": ("VL-classifier", "BCEWithLogitLoss", "normal"),          # VQA
    "TASK2": ("VL-classifier", "BCEWithLogitLoss", "normal"),          # GenomeQA
    "TASK4": ("V-logit-mc", "BCEWithLogitLoss", "normal"),             # Visual7w pointing
    "TASK7": ("VL-logit", "CrossEntropyLoss", "retrieval"),            # retrieval COCO
    "TASK8": ("VL-logit", "CrossEntropyLoss", "retrieval"),            # retrieval Flickr30k
    "TASK12": ("VL-binary-classifier", "BCEWithLogitLoss", "nlvr"),    # NLVR2
    "TASK13": ("VL-tri-classifier", "BCEWithLogitLoss", "normal"),     # SNLI-VE
    "TASK15": ("VL-classifier-GQA", "BCEWithLogitLoss", "normal"),     # GQA
    "TASK9": ("V-logit", "BCEWithLogitLoss", "normal"),                # refcoco
    "TASK5": ("VL-logit", "CrossEntropyLoss", "expand"),               # VCR Q->A
}


def score_with_logits(logits, labels):
    """compute_score_with_logits, task_utils.py (one-hot of the arg-max times the soft labels)."""
    one_hot = torch.zeros_like(labels).scatter_(1, logits.argmax(1, keepdim=True), 1.0)
    return one_hot * labels


def model_arguments(task_id, process, batch):
    """Dataset tuple -> (question, features, spatials, segment_ids, input_mask, image_mask, co_attention_mask,
    task_tokens), target, extras. task_utils.py:188-311."""
    if task_id in ("TASK4", "TASK17"):
        features, spatials, image_mask, question, target, input_mask, segment_ids, mc_ids, co_mask, _qid = batch
    else:
        features, spatials, image_mask, question, target, input_mask, segment_ids, co_mask, _qid = batch
        mc_ids = None
    batch_size, num_options = features.size(0), None
    if process == "expand":                                    # :243-269 one image, several text options
        n_box, num_options = features.size(1), question.size(1)
        features = features.unsqueeze(1).expand(batch_size, num_options, n_box, 2048).contiguous().view(-1, n_box, 2048)
        spatials = spatials.unsqueeze(1).expand(batch_size, num_options, n_box, 5).contiguous().view(-1, n_box, 5)
        image_mask = image_mask.unsqueeze(1).expand(batch_size, num_options, n_box).contiguous().view(-1, n_box)
        question, input_mask, segment_ids = (t.view(-1, t.size(2)) for t in (question, input_mask, segment_ids))
        co_mask = co_mask.view(-1, co_mask.size(2), co_mask.size(3))
    elif process == "retrieval":                               # :271-283 several (image, caption) pairs per sample
        num_options = question.size(1)
        features, spatials = features.view(-1, features.size(2), features.size(3)), spatials.view(-1, spatials.size(2), spatials.size(3))
        image_mask = image_mask.view(-1, image_mask.size(2))
        question, input_mask, segment_ids = (t.view(-1, t.size(2)) for t in (question, input_mask, segment_ids))
        co_mask = co_mask.view(-1, co_mask.size(2), co_mask.size(3))
    elif process == "nlvr":                                    # :285-309 two images per statement
        num_options = question.size(1)
        features = features.view(batch_size * 2, features.size(1) // 2, features.size(2))
        spatials = spatials.view(batch_size * 2, spatials.size(1) // 2, spatials.size(2))
        image_mask = image_mask.view(batch_size * 2, image_mask.size(1) // 2)
        question = question.repeat(1, 2).view(batch_size * 2, -1)
        input_mask = input_mask.repeat(1, 2).view(batch_size * 2, -1)
        segment_ids = segment_ids.repeat(1, 2).view(batch_size * 2, -1)
        co_mask = co_mask.view(batch_size * 2, co_mask.size(1) // 2, co_mask.size(2))
    elif process != "normal":
        raise NotImplementedError(process)                     # "dialog" (VisDial) is not restated
    task_tokens = torch.full((question.size(0), 1), int(task_id[4:]), dtype=question.dtype, device=question.device)
    return ((question, features, spatials, segment_ids, input_mask, image_mask, co_mask, task_tokens), target,
            dict(batch_size=batch_size, num_options=num_options, multiple_choice_ids=mc_ids))


def forward_train(task_id, batch, model, tasks=TASKS):
    """(loss, batch_score) of one training batch of `task_id`. task_utils.py:312-374."""
    kind, loss_name, process = tasks[task_id]
    criterion = LOSSES[loss_name]
    args, target, ex = model_arguments(task_id, process, batch)
    (vil_prediction, vil_prediction_gqa, vil_logit, vil_binary_prediction, vil_tri_prediction, _vision_prediction,
     vision_logit, _ling_prediction, _ling_logit) = tuple(model(*args))[:9]
    n = float(ex["batch_size"])
    if kind in ("VL-classifier", "VL-classifier-GQA"):
        pred = vil_prediction if kind == "VL-classifier" else vil_prediction_gqa
        loss = criterion(pred, target).mean() * target.size(1)
        score = score_with_logits(pred, target).sum() / n
    elif kind == "VL-logit":
        logits = vil_logit.view(ex["batch_size"], ex["num_options"])
        loss = criterion(logits, target)
        score = float((logits.argmax(1) == target).sum()) / n
    elif kind == "V-logit":
        loss = criterion(vision_logit, target).mean() * target.size(1)
        picked = target.squeeze(2).gather(1, vision_logit.argmax(1).view(-1, 1))
        score = float(torch.sum(picked > 0.5)) / ex["batch_size"]
    elif kind == "V-logit-mc":
        choice = vision_logit[:, 101:].squeeze(2).gather(1, ex["multiple_choice_ids"]).unsqueeze(2)
        loss = criterion(choice, target).mean() * target.size(1)
        score = float((choice.argmax(1) == target.argmax(1)).sum()) / n
    elif kind == "VL-binary-classifier":
        loss = criterion(vil_binary_prediction, target).mean()
        score = score_with_logits(vil_binary_prediction, target).sum() / n
    elif kind == "VL-tri-classifier":
        loss = criterion(vil_tri_prediction, target).mean()
        score = score_with_logits(vil_tri_prediction, target).sum() / n
    else:
        raise NotImplementedError(kind)
    return loss, score


def make_task_batch(task_id, batch, n_tok, n_reg, num_labels=3129, options=4, seed=0, tasks=TASKS, feat_scale=1.0):
    """A synthetic dataset tuple in the layout the reference's datasets return for `task_id`
    (vqa_dataset.py / retreival_dataset.py / visual7w_pointing_dataset.py / nlvr2_dataset.py / vcr_dataset.py
    `__getitem__`, stacked by the default collate)."""
    kind, _loss, process = tasks[task_id]
    g = torch.Generator().manual_seed(seed)
    lead_i = {"normal": (batch,), "expand": (batch,), "retrieval": (batch, options), "nlvr": (batch,)}[process]
    lead_t = {"normal": (batch,), "expand": (batch, options), "retrieval": (batch, options), "nlvr": (batch,)}[process]
    reg = n_reg * 2 if process == "nlvr" else n_reg
    features = torch.rand(*lead_i, reg, 2048, generator=g) * feat_scale
    spatials = torch.rand(*lead_i, reg, 5, generator=g)
    image_mask = (torch.rand(*lead_i, reg, generator=g) < 0.85).long()
    image_mask[..., 0] = 1
    question = torch.randint(1, 30000, (*lead_t, n_tok), generator=g)
    input_mask = (torch.rand(*lead_t, n_tok, generator=g) < 0.8).long()
    input_mask[..., 0] = 1
    segment_ids = torch.zeros(*lead_t, n_tok, dtype=torch.long)
    co_mask = torch.zeros(*lead_t, reg, n_tok)
    qid = torch.arange(batch)
    if kind in ("VL-classifier", "VL-classifier-GQA"):
        width = num_labels if kind == "VL-classifier" else 1533
        target = torch.zeros(batch, width)
        idx = torch.randint(0, width, (batch, 3), generator=g)
        target.scatter_(1, idx, torch.rand(batch, 3, generator=g))
    elif kind == "VL-logit":
        target = torch.randint(0, options, (batch,), generator=g) if process == "expand" else torch.zeros(batch, dtype=torch.long)
    elif kind == "V-logit":
        target = (torch.rand(batch, reg, 1, generator=g) < 0.1).float()
    elif kind == "V-logit-mc":
        target = torch.zeros(batch, 4, 1)
        target[torch.arange(batch), torch.randint(0, 4, (batch,), generator=g), 0] = 1.0
        mc = torch.stack([torch.randperm(reg - 101, generator=g)[:4] for _ in range(batch)])
        return (features, spatials, image_mask, question, target, input_mask, segment_ids, mc, co_mask, qid)
    elif kind == "VL-binary-classifier":
        target = torch.zeros(batch, 2)
        target[torch.arange(batch), torch.randint(0, 2, (batch,), generator=g)] = 1.0
    elif kind == "VL-tri-classifier":
        target = torch.zeros(batch, 3)
        target[torch.arange(batch), torch.randint(0, 3, (batch,), generator=g)] = 1.0
    return (features, spatials, image_mask, question, target, input_mask, segment_ids, co_mask, qid)
