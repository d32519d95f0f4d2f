"""CPU: the module-protocol boundary pinned against tests/golden/signatures.json, which
tests/golden/make_golden_signatures.py wrote from the reference's OWN classes (`inspect.signature`, `named_parameters()`,
`duett_kd_collate`'s table; /root/reference/models/main_architecture_duett.py:994-1010,1075-1079,1205-1222,546-555,
training_duett/data_processing.py:394-411, training_duett/trainer.py:88-102).

A product constructor / forward must accept every reference call: same parameter names in the same order, same kinds, same
defaults.  The only difference allowed is KEYWORD-ONLY extras whose name starts with `_` or is `config` (private switches of
this build, all defaulted)."""
import inspect
import json
import os

import pytest
import torch

from multimodal_edema_prediction_amd import cohort, duett, losses_duett, trajectory
from multimodal_edema_prediction_amd import main_architecture_duett as M

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "signatures.json")))

PRODUCT = {
    "DuettFeatureExtractor": M.DuettFeatureExtractor, "CXREncoder": M.CXREncoder,
    "PatchDualPathologyPerceiver": M.PatchDualPathologyPerceiver, "_PerceiverBlock": M._PerceiverBlock,
    "TeacherModel": M.TeacherModel, "StudentModel": M.StudentModel, "LocalTrajectoryEncoder": trajectory.LocalTrajectoryEncoder,
    "Model": duett.Model, "StudentKDLoss": losses_duett.StudentKDLoss, "PathologyMultiLabelLoss": losses_duett.PathologyMultiLabelLoss,
    "DualPathologyLoss": losses_duett.DualPathologyLoss, "VanillaKLKD": losses_duett.VanillaKLKD,
}
CASES = [(c, m) for c, ms in sorted(GOLD["classes"].items()) for m in sorted(ms)]


def _compare(ref_rows, fn, where):
    params = [(n, p) for n, p in inspect.signature(fn).parameters.items() if n != "self"]
    extras = [(n, p) for n, p in params if p.kind is inspect.Parameter.KEYWORD_ONLY and (n.startswith("_") or n == "config")]
    ours = [(n, p) for n, p in params if (n, p) not in extras]
    for n, p in extras:
        assert p.default is not inspect.Parameter.empty, f"{where}: private extra `{n}` has no default"
    assert [n for n, _ in ours] == [r[0] for r in ref_rows], f"{where}: parameter names / order differ from the reference"
    for (n, p), (rn, rkind, rdef) in zip(ours, ref_rows):
        assert p.kind.name == rkind, f"{where}: `{n}` is {p.kind.name}, reference {rkind}"
        if rdef is None:
            assert p.default is inspect.Parameter.empty, f"{where}: `{n}` has a default, the reference's is required"
        else:
            assert p.default is not inspect.Parameter.empty and repr(p.default) == rdef, \
                f"{where}: default of `{n}` is {p.default!r}, reference {rdef}"


@pytest.mark.parametrize("cname,method", CASES)
def test_signature_matches_reference(cname, method):
    cls = PRODUCT[cname]
    _compare(GOLD["classes"][cname][method], getattr(cls, method), f"{cname}.{method}")


def test_function_signatures_match_reference():
    _compare(GOLD["functions"]["load_duett_backbone"], M.load_duett_backbone, "load_duett_backbone")
    # cohort.collate is this build's duett_kd_collate (same two positional parameters; `mode` defaults to "teacher" here)
    assert [n for n in inspect.signature(cohort.collate).parameters] == [r[0] if r[0] != "batch" else "items"
                                                                          for r in GOLD["functions"]["duett_kd_collate"]]


def _build():
    T, V, DS, K = 32, 16, 8, 7
    torch.manual_seed(0)
    backbone = M.load_duett_backbone("synthetic", d_static_num=DS, d_time_series_num=V, n_timesteps=T, freeze=False)
    cxr = M.CXREncoder("synthetic", freeze=True)
    per = M.PatchDualPathologyPerceiver(K, backbone.d_representation)
    teacher = M.TeacherModel(backbone, cxr, per, cxr_return_patches=True, d_img=768, use_aux_cxr=False, patch_dual_pathology_mode=True)
    student = M.StudentModel(M.load_duett_backbone("synthetic", d_static_num=DS, d_time_series_num=V, n_timesteps=T, freeze=False))
    traj = trajectory.LocalTrajectoryEncoder(n_vars=V, n_timesteps=24)
    return {"TeacherModel": teacher, "StudentModel": student, "PatchDualPathologyPerceiver": per,
            "DuettFeatureExtractor": backbone, "LocalTrajectoryEncoder": traj}


@pytest.fixture(scope="module")
def built():
    return _build()


def _xt_internal(name: str) -> bool:
    """Names owned by x_transformers (third party, not installed: parity unpinned at that boundary)."""
    return "event_transformers." in name or "time_transformers." in name


@pytest.mark.parametrize("label", sorted(GOLD["parameters"]))
@pytest.mark.parametrize("what", ["parameters", "buffers"])
def test_parameter_and_buffer_names_match_reference(built, label, what):
    mod = built[label]
    it = mod.named_parameters() if what == "parameters" else mod.named_buffers()
    ours = {n: list(t.shape) for n, t in it if not n.startswith("cxr.backbone.")}
    ref = {r[0]: r[1] for r in GOLD[what][label]}
    ours_pinned = {n: s for n, s in ours.items() if not _xt_internal(n)}
    ref_pinned = {n: s for n, s in ref.items() if not _xt_internal(n)}
    if what == "buffers":            # workspaces / caches of this build are non-persistent and never named like a reference buffer
        ours_pinned = {n: s for n, s in ours_pinned.items() if n in ref_pinned or not n.split(".")[-1].startswith("_")}
    assert sorted(ours_pinned) == sorted(ref_pinned), \
        f"{label} {what}: only here {sorted(set(ours_pinned) - set(ref_pinned))[:6]}, only in the reference {sorted(set(ref_pinned) - set(ours_pinned))[:6]}"
    for n, s in ref_pinned.items():
        assert ours_pinned[n] == s, f"{label}.{n}: shape {ours_pinned[n]}, reference {s}"
    # the unpinned encoder: the same COUNT of tensors and the same total size per encoder (layout-free check)
    ours_xt = sorted(tuple(s) for n, s in ours.items() if _xt_internal(n))
    ref_xt = sorted(tuple(s) for n, s in ref.items() if _xt_internal(n))
    assert ours_xt == ref_xt, f"{label} {what}: encoder-internal tensor shapes differ from the restated x_transformers layout"


def test_requires_grad_pattern_and_name_contract(built):
    """trainer.py:88-102 groups parameters by name: prefixes `duett.` / `cxr.`, substring `correction_head`, suffixes `.beta`,
    `_queries`; engine.py:14-20 and evaluator.py:249-253 reach sub-modules by attribute."""
    t = built["TeacherModel"]
    ref = {r[0]: r[2] for r in GOLD["parameters"]["TeacherModel"]}
    names = [n for n, _ in t.named_parameters()]
    for pat, pred in (("duett.", str.startswith), ("cxr.", str.startswith), ("correction_head", str.__contains__),
                      (".beta", str.endswith), ("_queries", str.endswith)):
        ours = sorted(n for n in names if pred(n, pat) and not n.startswith("cxr.backbone.") and not _xt_internal(n))
        theirs = sorted(n for n in ref if pred(n, pat) and not _xt_internal(n))
        assert ours == theirs, f"name pattern `{pat}` selects different parameters"
    for cname, attrs in GOLD["attrs"].items():
        obj = {"TeacherModel": t, "PatchDualPathologyPerceiver": t.perceiver, "DuettFeatureExtractor": t.duett, "CXREncoder": t.cxr}[cname]
        for a in attrs:
            assert hasattr(obj, a), f"{cname}.{a} missing"
    assert any(isinstance(m, torch.nn.Dropout) for m in t.perceiver.correction_head.modules())       # trainer.py:205-209


def test_collate_table_matches_duett_kd_collate():
    ccfg = cohort.CohortCfg(n_timesteps=32, n_vars=16, d_static=8, image_size=224, seed=1234)
    for mode, ref in GOLD["collate"].items():
        items = [cohort.make_item(ccfg, i, with_image=(mode == "teacher")) for i in range(3)]
        out = cohort.collate(items, mode)
        assert sorted(out) == sorted(ref), f"collate({mode}) keys"
        for k, r in ref.items():
            v = out[k]
            if r["type"] == "tensor":
                assert isinstance(v, torch.Tensor) and str(v.dtype) == r["dtype"] and list(v.shape) == r["shape"], k
            else:
                assert type(v).__name__ == r["type"] and len(v) == r["len"], k
                assert str(v[0].dtype) == r["elem_dtype"] and list(v[0].shape) == r["elem_shape"], k


def test_output_key_tables_are_the_ones_the_gpu_tests_check():
    """The dict keys themselves are asserted on the GPU (test_gpu_model::test_teacher_forward_dict); here: the fixture lists
    exactly the keys engine.py / evaluator.py read (engine.py:145-157,183-186,284)."""
    keys = set(GOLD["outputs"]["TeacherModel.forward"])
    assert {"main_logit", "img_logits", "ts_logits", "fusion_logits", "ts_correction", "scaled_correction"} == keys
    assert set(GOLD["outputs"]["TeacherModel.forward(return_attn=True)"]) - keys == {"img_tokens", "ts_tokens", "fusion_tokens", "img_attn", "ts_attn"}
    assert GOLD["outputs"]["DualPathologyLoss.forward"] == sorted(["total", "img_total", "ts_total", "fus_total", "img_per", "ts_per", "fus_per"])
    assert GOLD["outputs"]["StudentKDLoss.forward"] == ["bce", "kd", "total"]
