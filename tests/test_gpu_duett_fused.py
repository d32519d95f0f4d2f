"""The fused DuETT front end (csrc/duett.hip): `medp_duett_embed_fwd` — psi built directly in the event view with the event
embedding added and the first ScaleNorm applied, and the time embedding — and `medp_duett_swap_add_norm` (axis swap + positional
add + the next encoder's ScaleNorm in one pass), against the CPU oracle (oracle/duett_ref.py, pinned by the reference's fixtures)
and against the separate launches they replace (bit-identical ScaleNorm)."""
import ctypes
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

from multimodal_edema_prediction_amd import functional as Fn  # noqa: E402
from multimodal_edema_prediction_amd.abi import check, lib, ptr, stream  # noqa: E402

DEV = "cuda"


@pytest.mark.parametrize("B,T,V", [(8, 32, 16), (5, 96, 48)])
def test_embed_stage_against_oracle(B, T, V):
    from multimodal_edema_prediction_amd.cohort import CohortCfg, make_batch
    from multimodal_edema_prediction_amd.main_architecture_duett import load_duett_backbone
    from oracle import duett_ref
    DS, E = 8, 24
    torch.manual_seed(0)
    m = load_duett_backbone("synthetic", d_static_num=DS, d_time_series_num=V, n_timesteps=T, freeze=True)
    with torch.no_grad():                                   # non-trivial BatchNorm statistics and norm gains
        for n, b in m.named_buffers():
            if n.endswith("running_mean"):
                b.copy_(0.1 * torch.randn_like(b))
            elif n.endswith("running_var"):
                b.copy_(0.5 + torch.rand_like(b))
        m.event_transformers[0].layers[0][0][0].g.fill_(1.3)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    m = m.to(DEV)
    batch = make_batch(CohortCfg(n_timesteps=T, n_vars=V, d_static=DS, image_size=32), 7, B, mode="student")
    x = (batch["x_ts"], batch["x_static"], list(batch["bin_ends"]))
    xin = duett_ref.feats_to_input(x, max_len=T)
    _, inter = duett_ref.encode(sd, duett_ref.DuettCfg(d_static_num=DS, d_time_series_num=V, n_timesteps=T), xin, return_intermediates=True)
    psi0, temb_ref = inter["psi0"], inter["time_emb"]                                   # [B,T+1,V+1,E], [B,T+1,(V+1)E]
    xe_ref = psi0.transpose(1, 2).flatten(2) + sd["full_event_embedding.weight"]          # model :80
    g = sd["event_transformers.0.layers.0.0.0.g"]
    h_ref = torch.nn.functional.normalize(xe_ref, dim=-1) * xe_ref.shape[-1] ** 0.5 * g

    xs_static, xs_ts, xs_times, _ = m.feats_to_input(x, B)
    w = m._prepare()[0]
    T1, V1 = T + 1, V + 1
    xe = torch.empty((B, V1, T1 * E), device=DEV)
    h = torch.empty((B, V1, T1 * E), device=DEV, dtype=torch.bfloat16)
    temb = torch.empty((B, T1, V1 * E), device=DEV)
    p0 = torch.empty((B, T1, V1, E), device=DEV)
    tab = torch.empty((B, E), device=DEV)
    check(lib().medp_duett_embed_fwd(ctypes.byref(w), ptr(xs_static), ptr(xs_ts), ptr(xs_times), B, T, ptr(xe), ptr(h), ptr(temb), ptr(p0),
                                     ptr(tab), 3, stream()), "duett_embed_fwd")
    assert float((p0.cpu() - psi0).abs().max()) < 2e-5
    assert float((xe.cpu() - xe_ref).abs().max()) < 2e-5
    assert float((temb.cpu() - temb_ref).abs().max()) < 2e-5
    assert float((h.float().cpu() - h_ref).abs().max()) < 8e-3 * float(h_ref.abs().max())            # bf16 output
    assert torch.equal(h, Fn.scalenorm(xe, m.event_transformers[0].layers[0][0][0].g))               # == the separate launch, bit for bit


@pytest.mark.parametrize("B,T,V", [(8, 32, 16), (6, 96, 48), (2, 140, 20)])
def test_embed_stage_mfma_form_is_bit_identical_to_the_valu_form(B, T, V, request):
    """The psi / time embeddings on the matrix cores in exact fp32 (`v_mfma_f32_16x16x4_f32`: a k-ordered fmaf chain) against the
    VALU kernels they replace: every output element bit for bit — psi0, the event-view rows, their bf16 ScaleNorm, the time
    embedding — on batches with masked time steps, masked events (SSL), over-length rows (T + 1 > 128: several passes of the cell
    loop) and the REP row.  So the fp32 kernel mode needs no second form (medp_dbg_embed_mfma switches at run time)."""
    from multimodal_edema_prediction_amd.main_architecture_duett import load_duett_backbone
    DS, E = 8, 24
    torch.manual_seed(3)
    m = load_duett_backbone("synthetic", d_static_num=DS, d_time_series_num=V, n_timesteps=T, freeze=True)
    with torch.no_grad():
        for n, b in m.named_buffers():
            if n.endswith("running_mean"):
                b.copy_(0.1 * torch.randn_like(b))
            elif n.endswith("running_var"):
                b.copy_(0.5 + torch.rand_like(b))
    m = m.to(DEV)
    g = torch.Generator().manual_seed(11)
    obs = torch.rand(B, T, V, generator=g) < 0.3
    xs_ts = torch.zeros(B, T, 2 * V + 1)
    xs_ts[:, :, :V] = torch.randn(B, T, V, generator=g) * obs
    xs_ts[:, :, V:2 * V] = obs.float() * torch.randint(1, 20, (B, T, V), generator=g).float()      # counts above the table: clipped
    xs_ts[:, :, V:2 * V][torch.rand(B, T, V, generator=g) < 0.05] = -1.0                            # masked events (SSL)
    xs_ts[:, :, 2 * V] = (torch.rand(B, T, generator=g) < 0.15).float()                            # masked time steps
    xs_ts[0, :, 2 * V] = 1.0                                                                         # a sample with every step masked
    xs_static, xs_times = torch.randn(B, DS, generator=g), torch.rand(B, T, generator=g) * 4
    xs_ts, xs_static, xs_times = xs_ts.to(DEV), xs_static.to(DEV), xs_times.to(DEV)
    w = m._prepare()[0]
    T1, V1 = T + 1, V + 1
    L = lib()
    prev = L.medp_dbg_embed_mfma(1)
    request.addfinalizer(lambda: L.medp_dbg_embed_mfma(prev))

    def run(on):
        L.medp_dbg_embed_mfma(on)
        xe = torch.full((B, V1, T1 * E), float("nan"), device=DEV)
        h = torch.zeros((B, V1, T1 * E), device=DEV, dtype=torch.bfloat16)
        temb = torch.full((B, T1, V1 * E), float("nan"), device=DEV)
        p0 = torch.full((B, T1, V1, E), float("nan"), device=DEV)
        tab = torch.empty((B, E), device=DEV)
        check(L.medp_duett_embed_fwd(ctypes.byref(w), ptr(xs_static), ptr(xs_ts), ptr(xs_times), B, T, ptr(xe), ptr(h), ptr(temb), ptr(p0),
                                     ptr(tab), 3, stream()), "duett_embed_fwd")
        torch.cuda.synchronize()
        return xe, h, temb, p0

    a, b_ = run(1), run(0)
    for name, x, y in zip(("xe", "h", "temb", "psi0"), a, b_):
        assert not torch.isnan(x.float()).any(), name
        assert torch.equal(x, y), f"{name}: {int((x != y).sum())} elements differ, max {float((x.float() - y.float()).abs().max()):.3e}"


@pytest.mark.parametrize("B,A1,A2", [(3, 33, 17), (4, 97, 49), (2, 49, 97), (1, 257, 97)])
@pytest.mark.parametrize("pending_norm,batched_add", [(False, False), (True, True)])
def test_swap_add_norm(B, A1, A2, pending_norm, batched_add):
    E = 24
    g0 = torch.Generator().manual_seed(B * 1000 + A1)
    x = torch.randn(B, A1, A2, E, generator=g0)
    rn = (0.5 + torch.rand(B * A1, generator=g0)) if pending_norm else None
    gp, gn = torch.tensor([1.2]), torch.tensor([0.8])
    add = torch.randn((B, A2, A1, E) if batched_add else (A2, A1, E), generator=g0)
    sc = (rn.view(B, A1, 1, 1) * ((A2 * E) ** 0.5 * gp)) if pending_norm else 1.0
    x_ref = (x * sc).transpose(1, 2) + add                                             # [B, A2, A1, E]
    xo = torch.empty((B, A2, A1 * E), device=DEV)
    ho = torch.empty((B, A2, A1 * E), device=DEV, dtype=torch.bfloat16)
    xd, rd, gpd, ad, gnd = x.to(DEV), (rn.to(DEV) if pending_norm else None), gp.to(DEV), add.to(DEV), gn.to(DEV)   # kept alive past the launch
    check(lib().medp_duett_swap_add_norm(ptr(xd), ptr(rd), ptr(gpd), ptr(ad), A2 * A1 * E if batched_add else 0, ptr(gnd), 1e-12, ptr(xo),
                                         ptr(ho), B, A1, A2, E, stream()), "swap_add_norm")
    ref = x_ref.reshape(B, A2, A1 * E)
    assert float((xo.cpu() - ref).abs().max()) <= 2e-6 * float(ref.abs().max()) + 1e-6
    assert torch.equal(ho, Fn.scalenorm(xo, gnd))                               # the fused norm == medp_scalenorm_fwd of its x


@pytest.mark.parametrize("B,N,H,dh", [(3, 17, 2, 12), (2, 33, 2, 12), (4, 49, 2, 12), (5, 97, 2, 12), (2, 257, 2, 12), (2, 100, 3, 16), (1, 16, 1, 4)])
def test_attn_dh16_mfma_forward(B, N, H, dh):
    """DuETT's attention on the matrix cores (medp_attn_dh16_fwd) against an fp64 softmax(QK^T/sqrt(dh))V on the same bf16-rounded
    operands (tolerance: bf16 probabilities and output, 8e-3 of the output range) and against the fp32 VALU kernel it replaces."""
    D = H * dh
    g0 = torch.Generator().manual_seed(N)
    qkv = torch.randn(B, N, 3 * D, generator=g0) * 1.5
    qd = qkv.to(DEV)
    o = torch.full((B, N, D), float("nan"), device=DEV, dtype=torch.bfloat16)
    rc = lib().medp_attn_dh16_fwd(ptr(qd), 3 * D, ptr(o), D, B, N, H, dh, dh ** -0.5, stream())
    check(rc, "attn_dh16_fwd")
    r = qkv.to(torch.bfloat16).double().view(B, N, 3, H, dh)
    q, k, v = r[:, :, 0].transpose(1, 2), r[:, :, 1].transpose(1, 2), r[:, :, 2].transpose(1, 2)
    want = (torch.softmax(q @ k.transpose(-1, -2) * dh ** -0.5, -1) @ v).transpose(1, 2).reshape(B, N, D)
    got = o.float().cpu().double()
    assert torch.isfinite(got).all()
    assert float((got - want).abs().max()) < 8e-3 * max(1.0, float(want.abs().max()))
    old = Fn.attn_small_fwd(qd[..., :D], qd[..., D:2 * D], qd[..., 2 * D:], B, N, N, H, dh, dh ** -0.5, q_batch_stride=N * 3 * D,
                            kv_batch_stride=N * 3 * D, out_dtype=torch.bfloat16)
    assert float((old.float() - o.float()).abs().max()) < 3e-2 * max(1.0, float(want.abs().max()))
    assert lib().medp_attn_dh16_fwd(ptr(qd), 3 * D, ptr(o), D, B, N, H, 20, 0.2, stream()) == -2          # not built: caller falls back
