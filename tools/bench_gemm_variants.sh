#!/bin/bash
# A/B of the GEMM variants at the ViT shapes (each variant in its own process; shapes interleaved inside)
for v in 3 4; do echo "== MEDP_GEMM_VARIANT=$v"; MEDP_GEMM_VARIANT=$v python tools/bench_kernels.py --gemm-only; done
