"""device-side batch assembly — `medp_feats_to_input` (reference duett/duett.py:159-187) and
`medp_ssl_mask_batch` (the masking half of `pretrain_prep_batch`, :189-237) against the CPU oracle (oracle/duett_ref.py, itself
pinned by the reference's fixtures in tests/test_oracle_golden.py / test_oracle_cfg1.py).  Byte work: BIT-EXACT with the
augmentation off, for every way a batch can arrive (host tensors, separate device tensors, rows of one stacked buffer), ragged
and over-length series included; the augmentation (library RNG, not torch's) is checked through its statistics and invariants."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

V, DS, MAXLEN = 16, 8, 32


def _series(n, T, seed):
    g = torch.Generator().manual_seed(seed)
    obs = torch.rand(T, V, generator=g) < 0.3
    vals = torch.randn(T, V, generator=g) * obs
    cnt = obs.float() * torch.randint(1, 6, (T, V), generator=g).float()
    return torch.cat((vals, cnt), 1), torch.randn(DS, generator=g), torch.arange(1, T + 1, dtype=torch.float32) / 24.0


def _batch(lens):
    items = [_series(i, T, 100 + i) for i, T in enumerate(lens)]
    return tuple(i[0] for i in items), tuple(i[1] for i in items), tuple(i[2] for i in items)


def _model(**kw):
    from multimodal_edema_prediction_amd.main_architecture_duett import load_duett_backbone
    torch.manual_seed(0)
    return load_duett_backbone("synthetic", d_static_num=DS, d_time_series_num=V, n_timesteps=MAXLEN, freeze=True, **kw).cuda()


@pytest.mark.parametrize("lens", [[32, 32, 32, 32], [32, 40, 7, 1, 33, 20], [5, 9, 3], [50, 64]])
@pytest.mark.parametrize("where", ["host", "device", "stacked"])
def test_feats_to_input_bit_exact(lens, where):
    from oracle import duett_ref
    m = _model()
    x = _batch(lens)
    want = duett_ref.feats_to_input((x[0], x[1], list(x[2])), max_len=MAXLEN)
    if where == "device":
        xin = tuple(tuple(t.cuda() for t in part) for part in x)
    elif where == "stacked":
        if len(set(lens)) != 1:
            pytest.skip("a stacked buffer holds equally long series")
        st = [torch.stack(part).cuda() for part in x]
        xin = tuple(tuple(s[i] for i in range(len(lens))) for s in st)
    else:
        xin = x
    before = [t.clone() for t in xin[0]]
    got = m.feats_to_input((xin[0], xin[1], xin[2]), len(lens))
    assert got[3] == want[3] == [min(n, MAXLEN) for n in lens]
    for g, w, name in zip(got[:3], want[:3], ("xs_static", "xs_ts", "xs_times")):
        assert g.shape == w.shape and g.dtype == torch.float32, name
        assert torch.equal(g.cpu(), w), name
    for a, b in zip(before, xin[0]):
        assert torch.equal(a, b)                    # the caller's tensors are left alone


def test_feats_to_input_augmentation_statistics():
    B, T = 64, 32
    m = _model(aug_noise=0.25, aug_mask=0.2)
    m.train()                                        # augmentation is a train()-mode behaviour (duett.py:169)
    x = _batch([T] * B)
    clean = _model().feats_to_input(x, B)
    s0, a0, t0, _ = (m.feats_to_input(x, B))
    s1, a1, _, _ = (m.feats_to_input(x, B))
    assert not torch.equal(a0, a1)                   # a fresh draw per call
    a, c = a0.cpu(), clean[1].cpu()
    dropped = a[:, :, -1] == 1
    assert set(a[:, :, -1].unique().tolist()) <= {0.0, 1.0}
    assert abs(float(dropped.float().mean()) - 0.2) < 0.03
    assert float(a[dropped][:, :-1].abs().max()) == 0.0                     # dropped timestep: row := 0, mask column := 1
    keep = ~dropped
    cnt = c[:, :, V:2 * V]
    assert torch.equal(a[:, :, V:2 * V][keep], cnt[keep])                    # counts are never perturbed
    d = (a[:, :, :V] - c[:, :, :V])[keep]
    ck = cnt[keep]
    assert float(d[ck == 0].abs().max()) == 0.0                              # noise is scaled by the count: none where unobserved
    z = d[ck > 0] / (0.25 * ck[ck > 0])
    assert z.numel() > 5000 and abs(float(z.mean())) < 0.05 and abs(float(z.std()) - 1.0) < 0.05
    assert abs(float(torch.mean(z ** 4)) - 3.0) < 0.4                       # normal, not uniform
    ds = (s0.cpu() - clean[0].cpu()) / 0.25
    assert abs(float(ds.std()) - 1.0) < 0.15 and torch.equal(t0, clean[2])
    m.eval()
    e = m.feats_to_input(x, B)
    assert torch.equal(e[1], clean[1]) and torch.equal(e[0], clean[0])       # eval(): no augmentation


@pytest.mark.parametrize("dropout,events", [(0.5, True), (0.0, True), (0.5, False)])
def test_ssl_mask_batch_bit_exact(dropout, events):
    from multimodal_edema_prediction_amd.duett import Model
    from oracle import duett_ref
    lens = [32, 32, 20, 32, 9, 32, 32, 2]
    x = _batch(lens)
    torch.manual_seed(0)
    m = Model(DS, V, 1, masked_transform_timesteps=MAXLEN, max_len=MAXLEN, pretrain=True, pretrain_dropout=dropout,
              predict_events=events, seed=7).cuda()
    got = m.pretrain_prep_batch(x, len(lens))
    want = duett_ref.pretrain_prep_batch((x[0], x[1], list(x[2])), np.random.default_rng(7), V, MAXLEN, pretrain_dropout=dropout,
                                         predict_events=events)
    (gs, gc, gt, gn), (ws, wc, wt, wn) = got[0], want[0]
    assert gn == wn and torch.equal(gs.cpu(), ws) and torch.equal(gt.cpu(), wt)
    # bit-exact, signed zeros included (x * False keeps the sign of x)
    assert torch.equal(gc.cpu().view(torch.int32), wc.view(torch.int32))
    assert torch.equal(got[1].cpu(), want[1]) and torch.equal(got[2].cpu(), want[2])
    if events:
        assert torch.equal(got[3].cpu(), want[3]) and torch.equal(got[4].cpu(), want[4])
    else:
        assert got[3] == [] and got[4] == []
