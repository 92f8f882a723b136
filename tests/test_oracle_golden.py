"""Pins oracle/vilbert_oracle.py against the committed golden vectors (outputs of the REAL reference,
tests/golden/make_golden.py). Runs everywhere - this is what pins the oracle on the GPU box, where
/root/reference does not exist."""
import pytest
import torch

import helpers
from helpers import cases
from oracle import vilbert_oracle as vo


@pytest.mark.parametrize("case", list(cases.CASES))
def test_oracle_reproduces_reference_outputs(case):
    c = cases.CASES[case]
    cfg, sd, x = cases.case_inputs(case)
    args = cases.forward_args(case, x)
    with torch.no_grad():
        out = vo.vltasks_forward(sd, cfg, *args) if c["kind"] == "vltasks" else vo.pretraining_forward(sd, cfg, *args)
    gold = helpers.load_golden(case)
    names = cases.output_names(case)
    assert set(gold) == set(names)
    for i, n in enumerate(names):
        # the restatement runs the same fp32 torch CPU ops as the reference: agreement is ~1e-6
        helpers.assert_close(cases.sample(case, n, out[i]), gold[n], n, atol=2e-5, rtol=2e-5)
