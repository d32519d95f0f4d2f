"""Build libmedp_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
from __future__ import annotations

import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmedp_hip.so")


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.cpp")))


def headers():
    return glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(ROOT, "include", "*.h"))


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in sources() + headers())


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs, procs = [], []
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    for src in sources():
        obj = os.path.join(HERE, "build", os.path.basename(src) + ".o")
        objs.append(obj)
        if not force and os.path.exists(obj) and all(os.path.getmtime(obj) > os.path.getmtime(d) for d in [src] + headers()):
            continue
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-x", "hip", "-I", CSRC,
               "-I", os.path.join(ROOT, "include"), "-Wno-unused-result", "-Wno-unused-value", "-c", src, "-o", obj]
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out}")
        if verbose and out.strip():
            print(out)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}")
    if verbose:
        print(f"built {LIB}")
    return LIB


def build_variant_lib(tag: str, defines: list) -> str:
    """libmedp_hip_<tag>.so: the library with gemm_bf16_v7.hip compiled with extra -D switches (timing-only ablation builds of the
    persistent GEMM, tools/ablate_gemm_v7.py: MEDP_V7_ABLATE_MFMA, MEDP_V7_ABLATE_LOADS).  Loaded only through MEDP_HIP_LIB."""
    build()
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    src = os.path.join(CSRC, "gemm_bf16_v7.hip")
    obj = os.path.join(HERE, "build", f"gemm_bf16_v7.{tag}.o")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-x", "hip", "-I", CSRC, "-I", os.path.join(ROOT, "include"),
           "-Wno-unused-result", "-Wno-unused-value"] + [f"-D{d}" for d in defines] + ["-c", src, "-o", obj]
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    objs = [os.path.join(HERE, "build", os.path.basename(s) + ".o") for s in sources() if not s.endswith("gemm_bf16_v7.hip")] + [obj]
    out = os.path.join(HERE, f"libmedp_hip_{tag}.so")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs, check=True)
    return out


def build_trace_lib() -> str:
    """libmedp_hip_trace.so: the same library with gemm_bf16_v7.hip compiled -DMEDP_V7_PHASE_TRACE (in-kernel phase clocks,
    tools/trace_gemm_v7.py --phases).  A profiling aid, loaded only when MEDP_HIP_LIB points at it."""
    build()
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    src = os.path.join(CSRC, "gemm_bf16_v7.hip")
    obj = os.path.join(HERE, "build", "gemm_bf16_v7.trace.o")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-x", "hip", "-I", CSRC, "-I", os.path.join(ROOT, "include"),
           "-Wno-unused-result", "-Wno-unused-value", "-DMEDP_V7_PHASE_TRACE", "-c", src, "-o", obj]
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    objs = [os.path.join(HERE, "build", os.path.basename(s) + ".o") for s in sources() if not s.endswith("gemm_bf16_v7.hip")] + [obj]
    out = os.path.join(HERE, "libmedp_hip_trace.so")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs, check=True)
    return out


if __name__ == "__main__":
    if "--trace" in sys.argv:
        print(build_trace_lib())
        sys.exit(0)
    if "--ablate" in sys.argv:
        print(build_variant_lib("nomfma", ["MEDP_V7_ABLATE_MFMA"]))
        print(build_variant_lib("noloads", ["MEDP_V7_ABLATE_LOADS"]))
        sys.exit(0)

    build(force="--force" in sys.argv)
