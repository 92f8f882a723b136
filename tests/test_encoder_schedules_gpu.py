"""HIP model against the oracle on the encoder schedule of every shipped two-stream config (widths shrunk, layer counts
and connection ids kept - see tests/test_encoder_schedules.py, which pins the oracle on the same cases to the real
reference). Bar: 1e-4 + 1e-4 |want| on all nine outputs of VILBertForVLTasks."""
import pytest
import torch

import helpers
from helpers import cases
from oracle import synth, vilbert_oracle as vo
from test_encoder_schedules import SHIPPED, VL_ARGS, shrunk

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("cfgname", SHIPPED)
def test_hip_model_follows_the_schedule(cfgname):
    from vilbert.vilbert import BertConfig, VILBertForVLTasks
    cfg = shrunk(cfgname)
    sd = synth.make_state_dict(cfg, "vltasks", seed=31)
    m = VILBertForVLTasks(BertConfig.from_dict(cfg), num_labels=1)
    res = m.load_state_dict(sd, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    m = m.eval().to(DEV)
    x = synth.make_inputs(cfg, 3, 9, 7, seed=31, ragged=True)
    args = tuple(x[n] for n in VL_ARGS)
    with torch.no_grad():
        got = m(*helpers.to_device(args, DEV))
        want = vo.vltasks_forward(sd, cfg, *args)
    for n, g, w in zip(cases.VL_NAMES, got, want):
        helpers.assert_close(g, w, "%s/%s" % (cfgname, n))
