"""The synthetic linear-probe checkpoint of the dual teacher (what cxr_linear_training.ipynb :827-845 saves: label list + the
Linear(768, C) classifier), rebuilt identically wherever the tests run (tests/golden/make_golden_dual.py used the same content)."""
from helpers import synth_tensor

PRETRAINED_LABELS = ["label_opacity", "label_edema", "label_fracture", "label_cardiomegaly", "label_consolidation",
                     "label_effusion", "label_pneumothorax", "label_pneumonia", "label_atelectasis"]


def cxr_head_state():
    return {"num_classes": len(PRETRAINED_LABELS), "label_cols": list(PRETRAINED_LABELS),
            "classifier_state_dict": {"1.weight": synth_tensor("cxr_head.1.weight", (len(PRETRAINED_LABELS), 768), seed=6),
                                      "1.bias": synth_tensor("cxr_head.1.bias", (len(PRETRAINED_LABELS),), seed=6)}}
