"""MI355X-native hot path of lastdancewithyou/multimodal_edema_prediction.

Host-side mirror of the reference's model/loss interface over a C-ABI HIP library
(`csrc/` → `libmedp_hip.so`, declared in `include/medp_hip.h`).  See DESIGN.md.
"""
import os as _os


__version__ = "0.1.0"
