"""The fused per-variable embedding MLP (csrc/duett_embed_train.hip: Linear(2, 64) -> ReLU -> BatchNormLastDim -> Linear(64, 24), hidden
activations recomputed instead of stored) against float64 autograd of the same layers (duett/duett.py:11-39) and against the grouped-layer
kernels it replaces."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
EPS, MOM = 1e-5, 0.1


def _params(G, KIN, C, E, seed):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s, scale=1.0: torch.randn(*s, generator=g) * scale
    return dict(W0=r(G, C, KIN, scale=0.7), b0=r(G, C, scale=0.3), bw=1 + r(G, C, scale=0.2), bb=r(G, C, scale=0.2), W1=r(G, E, C, scale=0.2),
                b1=r(G, E, scale=0.1), rm=r(G, C, scale=0.1), rv=1 + r(G, C, scale=0.1).abs())


def _ref(x, P, dout, batch_stats):
    """float64 autograd; returns out, grads, (running_mean, running_var) after the call"""
    x = x.double().requires_grad_(True)
    p = {k: v.double().requires_grad_(k in ("W0", "b0", "bw", "bb", "W1", "b1")) for k, v in P.items()}
    a = torch.relu(torch.einsum("grk,gck->grc", x, p["W0"]) + p["b0"][:, None])
    R = x.shape[1]
    if batch_stats:
        mu, var = a.mean(1), a.var(1, unbiased=False)
        rm = (1 - MOM) * p["rm"] + MOM * mu.detach()
        rv = (1 - MOM) * p["rv"] + MOM * a.var(1, unbiased=True).detach()
    else:
        mu, var, rm, rv = p["rm"], p["rv"], p["rm"], p["rv"]
    hb = (a - mu[:, None]) / torch.sqrt(var[:, None] + EPS) * p["bw"][:, None] + p["bb"][:, None]
    out = torch.einsum("grc,gec->gre", hb, p["W1"]) + p["b1"][:, None]
    out.backward(dout.double())
    grads = {"x": x.grad, **{k: p[k].grad for k in ("W0", "b0", "bw", "bb", "W1", "b1")}}
    return out.detach(), grads, (rm.detach(), rv.detach())


def _run(x, P, dout, batch_stats, fused):
    from multimodal_edema_prediction_amd import duett_train as DT
    xd = x.to(DEV).requires_grad_(True)
    p = {k: v.clone().to(DEV).requires_grad_(k in ("W0", "b0", "bw", "bb", "W1", "b1")) for k, v in P.items()}
    prev = DT._FUSED_GMLP
    DT._FUSED_GMLP = fused
    try:
        out = DT._mlp_bn(xd, p["W0"], p["b0"], p["bw"], p["bb"], p["rm"], p["rv"], p["W1"], p["b1"], 0, batch_stats)
        out.backward(dout.to(DEV))
    finally:
        DT._FUSED_GMLP = prev
    grads = {"x": xd.grad, **{k: p[k].grad for k in ("W0", "b0", "bw", "bb", "W1", "b1")}}
    return out.detach(), grads, (p["rm"].detach(), p["rv"].detach())


def _close(a, b, tol, what):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    err = float((a - b).abs().max())
    assert err <= tol * max(float(b.abs().max()), 1e-6), (what, err, float(b.abs().max()))


@pytest.mark.parametrize("G,R", [(5, 300), (3, 1500), (2, 7)])
@pytest.mark.parametrize("batch_stats", [True, False])
def test_fused_embedding_mlp_against_float64_autograd(G, R, batch_stats):
    torch.manual_seed(G * 1000 + R)
    KIN, C, E = 2, 64, 24
    x = torch.randn(G, R, KIN)
    x[..., 1] = torch.randint(0, 6, (G, R)).float() * 0.3            # the count-embedding column takes few distinct values
    dout = torch.randn(G, R, E)
    P = _params(G, KIN, C, E, seed=R)
    o_ref, g_ref, (rm_ref, rv_ref) = _ref(x, P, dout, batch_stats)
    o, g, (rm, rv) = _run(x, P, dout, batch_stats, fused=True)
    _close(o, o_ref, 2e-5, "out")
    for k in g_ref:
        _close(g[k], g_ref[k], 2e-4, "grad " + k)
    _close(rm, rm_ref, 1e-5, "running_mean")
    _close(rv, rv_ref, 1e-5, "running_var")
    o2, g2, _ = _run(x, P, dout, batch_stats, fused=True)                # fixed-order sums: the same bits again
    assert torch.equal(o, o2) and all(torch.equal(g[k], g2[k]) for k in g)
    o_old, g_old, (rm_old, rv_old) = _run(x, P, dout, batch_stats, fused=False)
    _close(o, o_old, 1e-5, "out vs grouped layers")
    for k in g_old:
        _close(g[k], g_old[k], 1e-4, "grad vs grouped layers " + k)
    _close(rm, rm_old, 1e-6, "running_mean vs grouped layers")
    _close(rv, rv_old, 1e-6, "running_var vs grouped layers")


def test_unsupported_widths_take_the_grouped_layers():
    from multimodal_edema_prediction_amd.abi import lib
    assert lib().medp_gmlp_supported(2, 64, 24) == 1
    assert lib().medp_gmlp_supported(8, 128, 24) == 0 and lib().medp_gmlp_supported(2, 64, 32) == 0
    torch.manual_seed(0)
    G, R, KIN, C, E = 1, 64, 8, 128, 24                                  # the tab encoder's shape
    x, dout, P = torch.randn(G, R, KIN), torch.randn(G, R, E), _params(G, KIN, C, E, seed=1)
    o_ref, g_ref, _ = _ref(x, P, dout, True)
    o, g, _ = _run(x, P, dout, True, fused=True)
    _close(o, o_ref, 2e-5, "out")
    for k in g_ref:
        _close(g[k], g_ref[k], 2e-4, "grad " + k)
