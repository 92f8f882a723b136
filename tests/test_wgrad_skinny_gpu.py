"""Weight gradient of a linear layer with at most 8 input features (the 5-wide region-location projection of the
image embeddings, reference vilbert.py:385-386: image_location_embeddings = nn.Linear(5, v_hidden_size)).

As a tiled GEMM this is a [v_hidden, 5] output reduced over batch x regions rows; csrc/gemm.hip runs it as one streaming
pass over the gradient (wgrad_skinny_kernel) with an ordered reduce over row slabs. Checked against float64, in both
reduction modes (ordered through the workspace = default, fp32 atomics), accumulating into an existing gradient, and
bit-identical from run to run in the ordered mode."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _rand(*shape, seed=0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed))


@pytest.fixture(params=[True, False], ids=["ordered", "atomics"])
def reduction(request):
    from vilbert import _native
    wanted = _native._DET["wanted"]
    _native.set_deterministic(request.param)
    yield request.param
    _native.set_deterministic(wanted)


# rows (batch x regions), out-features, in-features: the model's shape at batch 256 / 64, ragged row counts, every K <= 8,
# an out-feature count that is not a multiple of the 256 columns of a block
@pytest.mark.parametrize("M,N,K", [(9472, 1024, 5), (2368, 1024, 5), (9473, 1024, 5), (257, 1024, 1), (4000, 1024, 8),
                                   (3001, 332, 3), (9472, 2048, 5)])
@pytest.mark.parametrize("bias", [True, False])
def test_matches_float64(reduction, M, N, K, bias):
    from vilbert import ops
    x, dy = _rand(M, K, seed=1).to(DEV), _rand(M, N, seed=2).to(DEV)
    (dw,), (db,) = ops.linear_bwd_weight(dy, x, 1, N, [bias])
    want_w = dy.double().t() @ x.double()
    assert dw.shape == (N, K)
    err = (dw.double() - want_w).abs().max().item() / want_w.abs().max().item()
    assert err < 2e-6, err
    if bias:
        want_b = dy.double().sum(0)
        assert (db.double() - want_b).abs().max().item() / want_b.abs().max().item() < 2e-6
    else:
        assert db is None
    if reduction:
        (dw2,), (db2,) = ops.linear_bwd_weight(dy, x, 1, N, [bias])
        assert torch.equal(dw, dw2) and (not bias or torch.equal(db, db2))


def test_adds_into_an_existing_gradient(reduction):
    """Gradient-arena slices hold earlier contributions (gradient accumulation): the kernel adds."""
    from vilbert import ops
    M, N, K = 9472, 1024, 5
    x, dy = _rand(M, K, seed=3).to(DEV), _rand(M, N, seed=4).to(DEV)
    w0, b0 = _rand(N, K, seed=5).to(DEV), _rand(N, seed=6).to(DEV)
    dw, db = w0.clone(), b0.clone()
    ops.linear_bwd_weight(dy, x, 1, N, [True], dw_out=[dw], db_out=[db])
    want_w = w0.double() + dy.double().t() @ x.double()
    want_b = b0.double() + dy.double().sum(0)
    assert (dw.double() - want_w).abs().max().item() / want_w.abs().max().item() < 2e-6
    assert (db.double() - want_b).abs().max().item() / want_b.abs().max().item() < 2e-6


def test_strided_rows():
    """dY may be a column slice of a wider buffer, X a row-strided view: leading dimensions are honoured."""
    from vilbert import ops
    M, N, K = 2368, 1024, 5
    wide = _rand(M, N + 256, seed=7).to(DEV)
    xw = _rand(M, 8, seed=8).to(DEV)
    dy, x = wide[:, 128:128 + N], xw[:, :K]
    (dw,), (db,) = ops.linear_bwd_weight(dy, x, 1, N, [True])
    want_w = dy.double().t() @ x.double()
    assert (dw.double() - want_w).abs().max().item() / want_w.abs().max().item() < 2e-6
    assert (db.double() - dy.double().sum(0)).abs().max().item() < 1e-3
