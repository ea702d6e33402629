"""Builds libbez_sim.so (HIP, gfx950 only) in-tree: python -m bez_isaacgym_amd.build"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "bez_sim.hip")
DEPS = [SRC] + [os.path.join(HERE, "csrc", f) for f in ("bez_kernels.h", "bez_spatial.h", "bez_model_gen.h")] + \
       [os.path.join(HERE, "..", "include", "bez_sim.h")]
OUT = os.path.join(HERE, "lib", "libbez_sim.so")


def lib_path():
    return OUT


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(d) > t for d in DEPS)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    # -fno-slp-vectorize: hipcc's SLP pass packs adjacent scalar f32 ops into v_pk_* and pays for it in v_mov shuffles
    # (23.1k -> 16.4k VALU instructions in the fused kernel, 33.2 -> 29.1 us per step on MI355X)
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fno-slp-vectorize"] + \
        os.environ.get("BEZ_HIPCC_FLAGS", "").split() + ["-o", OUT, SRC]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
    print(OUT)
