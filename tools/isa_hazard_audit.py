#!/usr/bin/env python3
"""Dependency-distance audit of a gfx950 ISA listing (hipcc --cuda-device-only -S).

For every VGPR read it finds the nearest earlier writer in the same basic block and records the number of wait states
between them (every instruction = 1, `s_nop N` = N + 1), then prints the MINIMUM distance per (producer class ->
consumer class) pair, and the waits that precede every `s_barrier`.  Used to compare the packed-f32 softmax build of
attention_dh64.hip (git 1d227fd) with the scalar one (DESIGN.md "Bit stability"): which software-managed hazards of the
CDNA3/4 ISA (trans-op forwarding, MFMA result -> VALU, VALU -> permlane-swap, LDS-DMA -> barrier) sit at their minimum.

    python tools/isa_hazard_audit.py a.s [b.s]
"""
from __future__ import annotations

import re
import sys
from collections import defaultdict

REG = re.compile(r"\b([va])(\d+)\b|\b([va])\[(\d+):(\d+)\]")
TRANS = ("v_exp_", "v_log_", "v_rcp_", "v_rsq_", "v_sqrt_", "v_sin_", "v_cos_")


def regs(tok: str):
    out = []
    for m in REG.finditer(tok):
        if m.group(1):
            out.append((m.group(1), int(m.group(2))))
        else:
            out += [(m.group(3), i) for i in range(int(m.group(4)), int(m.group(5)) + 1)]
    return out


def klass(mn: str) -> str:
    if mn.startswith("v_mfma") or mn.startswith("v_smfma"):
        return "mfma"
    if mn.startswith(TRANS):
        return "trans"
    if mn.startswith("v_pk_") and mn.endswith("_f32"):
        return "pk_f32"
    if mn.startswith("v_permlane"):
        return "permlane_swap"
    if mn.startswith("v_cvt_pk_bf16"):
        return "cvt_pk"
    if mn.startswith("ds_read") or mn.startswith("ds_load"):
        return "ds_read"
    if mn.startswith(("global_load", "buffer_load", "flat_load")):
        return "vmem_load"
    if mn.startswith("v_"):
        return "valu"
    return "other"


def audit(path: str):
    mins: dict = {}
    barrier_waits = defaultdict(int)
    last_writer: dict = {}
    pos = 0
    last_wait = None
    since_dma = False
    for line in open(path):
        s = line.split(";")[0].strip()
        if not s or s.startswith((".", "//")):
            continue
        if s.endswith(":"):                      # label: new basic block
            last_writer.clear()
            continue
        parts = s.split(None, 1)
        mn = parts[0]
        ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
        if mn == "s_nop":
            pos += int(ops[0], 0) + 1
            continue
        pos += 1
        if mn == "s_waitcnt":
            last_wait = parts[1] if len(parts) > 1 else ""
            continue
        if "load_lds" in mn:
            since_dma = True
        if mn == "s_barrier":
            barrier_waits[(last_wait or "none", "after LDS-DMA" if since_dma else "no DMA pending")] += 1
            since_dma = False
            continue
        k = klass(mn)
        if k == "other":
            continue
        is_store = mn.startswith(("global_store", "buffer_store", "ds_write", "ds_store", "flat_store")) or "load_lds" in mn
        dst = [] if is_store else (regs(ops[0]) if ops else [])
        srcs = [r for o in (ops if is_store else ops[1:]) for r in regs(o)]
        if mn.startswith("v_permlane"):           # swap: both operands are read and written
            srcs = [r for o in ops for r in regs(o)]
            dst = srcs
        for r in srcs:
            w = last_writer.get(r)
            if w is not None:
                key = (w[1], k)
                d = pos - w[0] - 1
                if key not in mins or d < mins[key][0]:
                    mins[key] = (d, w[2], s)
        for r in dst:
            last_writer[r] = (pos, k, s)
    return mins, barrier_waits


def main():
    res = [audit(p) for p in sys.argv[1:]]
    keys = sorted({k for m, _ in res for k in m})
    print(f"{'producer -> consumer':34s}" + "".join(f"{p[-28:]:>30s}" for p in sys.argv[1:]))
    for k in keys:
        row = f"{k[0] + ' -> ' + k[1]:34s}"
        for m, _ in res:
            row += f"{(str(m[k][0]) if k in m else '-'):>30s}"
        print(row)
    for p, (m, bw) in zip(sys.argv[1:], res):
        print(f"\n{p}: waits preceding s_barrier")
        for (w, d), n in sorted(bw.items()):
            print(f"   {n:3d} x  [{d}]  last s_waitcnt = {w}")
        for k in (("trans", "pk_f32"), ("mfma", "pk_f32"), ("pk_f32", "mfma"), ("trans", "cvt_pk"), ("valu", "permlane_swap"),
                  ("permlane_swap", "valu")):
            if k in m:
                print(f"   closest {k[0]} -> {k[1]} (distance {m[k][0]}):\n      {m[k][1]}\n      {m[k][2]}")


if __name__ == "__main__":
    main()
