"""bench.py host-side pieces that can be checked without a GPU: the algorithmic FLOP formula the roofline
uses (15.247 GFLOP BertModel forward, 17.202 incl. VLTasks heads, 17.171 incl. the
pre-training heads, 32.993 for bert_large) and the synthetic batch conventions."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from vilbert.vilbert import BertConfig  # noqa: E402


def _cfg(name):
    return BertConfig.from_json_file(os.path.join(ROOT, "vilbert-multi-task_amd", "config", name)).to_dict()


def test_flop_formula_matches_the_survey_figures():
    base = _cfg("bert_base_6layer_6conect.json")
    bert, vl = bench.model_flops_per_sample(base, 36, 36, "vltasks")
    assert bert / 1e9 == pytest.approx(15.247, abs=2e-3)
    assert vl / 1e9 == pytest.approx(17.202, abs=2e-3)
    _, pre = bench.model_flops_per_sample(base, 36, 36, "pretraining")
    assert pre / 1e9 == pytest.approx(17.171, abs=2e-3)
    _, pre37 = bench.model_flops_per_sample(base, 36, 37, "pretraining")
    assert 3 * pre37 / 1e9 == pytest.approx(52.0, abs=0.05)           # fwd+bwd at the loader's R = 37
    large, _ = bench.model_flops_per_sample(_cfg("bert_large_6layer_6conect.json"), 36, 36, "vltasks")
    assert large / 1e9 == pytest.approx(32.993, abs=2e-3)


def test_synthetic_batch_follows_the_loader_conventions():
    cfg = _cfg("bert_base_6layer_6conect.json")
    x = bench.synthetic_batch(cfg, 4, 36, 37, 7, True)
    assert x["image_feat"].shape == (4, 37, 2048) and x["image_loc"].shape == (4, 37, 5)
    assert x["image_target"].shape == (4, 36, 1601) and x["image_label"].shape == (4, 36)
    assert torch.allclose(x["image_target"].sum(-1), torch.ones(4, 36), atol=1e-5)
    assert set(x["image_label"].unique().tolist()) <= {-1, 1} and (x["image_label"][:, 0] == 1).all()
    assert ((x["masked_lm_labels"] == -1) | (x["masked_lm_labels"] == x["input_ids"])).all()
    assert (x["image_loc"][:, 0] == torch.tensor([0.0, 0.0, 1.0, 1.0, 1.0])).all()
    y = bench.synthetic_batch(cfg, 4, 36, 37, 7, True)
    assert all(torch.equal(x[k], y[k]) for k in x)                      # seeded


def test_bench_starts_its_own_ranks_when_not_under_a_launcher():
    """`python bench.py --gpus 2` without RANK in the environment re-executes itself through torch.distributed.run
    (one process per GPU, rendezvous on 127.0.0.1). Without GPUs every rank stops at its device check - which proves
    that the ranks were started with the right environment."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env=env)
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        assert r.returncode == 0, r.stderr[-2000:]
        return
    assert r.returncode != 0
    text = r.stdout + r.stderr
    assert "GPU(s) visible" in text, text[-2000:]
