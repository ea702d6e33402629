"""Builds libbez_sim.so (HIP, gfx950 only) in-tree: python -m bez_isaacgym_amd.build"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "bez_sim.hip")
OUT = os.path.join(HERE, "lib", "libbez_sim.so")
STAMP = OUT + ".buildinfo"  # hash of every source + the compiler flags the .so was built from


def _deps():
    import glob
    return sorted(glob.glob(os.path.join(HERE, "csrc", "*.h")) + glob.glob(os.path.join(HERE, "csrc", "*.inc")) + glob.glob(os.path.join(HERE, "csrc", "*.hip")) +
                  [os.path.join(HERE, "..", "include", "bez_sim.h")])


def _flags():
    return ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize"] + os.environ.get("BEZ_HIPCC_FLAGS", "").split()


def _sources():
    import glob
    return sorted(glob.glob(os.path.join(HERE, "csrc", "*.hip")))


def source_hash():
    import hashlib
    h = hashlib.sha256(" ".join(_flags()).encode())
    for d in _deps():
        h.update(os.path.basename(d).encode())
        with open(d, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def lib_path():
    """The in-tree library -- or, for same-box A/B measurements of library variants (tools/ppo_ab.sh, tools/ab_bench.py), the file BEZ_SIM_LIB
    names: the tree is never overwritten by an experiment (round-4 advisor finding)."""
    return os.environ.get("BEZ_SIM_LIB") or OUT


def needs_build():
    if not os.path.exists(OUT) or not os.path.exists(STAMP):
        return True
    with open(STAMP) as f:
        return f.read().strip() != source_hash()


def build(force=False, verbose=False):
    """Compile + link under an inter-process lock: every rank of a torchrun job imports the package and may find a stale tree at
    the same moment; only the first one builds, the others wait and then see the fresh stamp.  Objects go to a per-process
    directory and the finished library / stamp are moved into place atomically, so a reader never maps a half-written file."""
    import fcntl
    import shutil
    import tempfile
    if os.environ.get("BEZ_SIM_LIB"):
        return os.environ["BEZ_SIM_LIB"]   # an explicitly named variant is used as it is
    if not force and not needs_build():
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    with open(os.path.join(os.path.dirname(OUT), ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not needs_build():  # another process built it while this one waited
                return OUT
            hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
            # -fno-slp-vectorize: hipcc's SLP pass packs adjacent scalar f32 ops into v_pk_* and pays for it in v_mov shuffles
            # (23.1k -> 16.4k VALU instructions in the fused kernel, 33.2 -> 29.1 us per step on MI355X)
            # one translation unit per .hip file, compiled side by side, then linked into the one shared library
            objdir = tempfile.mkdtemp(prefix="obj.%d." % os.getpid(), dir=os.path.dirname(OUT))
            try:
                procs, objs = [], []
                for src in _sources():
                    obj = os.path.join(objdir, os.path.basename(src)[:-4] + ".o")
                    cmd = [hipcc] + _flags() + ["-c", "-o", obj, src]
                    if verbose:
                        print(" ".join(cmd))
                    procs.append((cmd, subprocess.Popen(cmd)))
                    objs.append(obj)
                for cmd, p in procs:
                    if p.wait() != 0:
                        raise subprocess.CalledProcessError(p.returncode, cmd)
                tmp_so = os.path.join(objdir, "libbez_sim.so")
                cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp_so] + objs
                if verbose:
                    print(" ".join(cmd))
                subprocess.run(cmd, check=True)
                h = source_hash()
                if os.path.exists(STAMP):
                    os.remove(STAMP)  # never a fresh stamp beside an old library
                os.replace(tmp_so, OUT)
                tmp_stamp = os.path.join(objdir, "stamp")
                with open(tmp_stamp, "w") as f:
                    f.write(h + "\n")
                os.replace(tmp_stamp, STAMP)
            finally:
                shutil.rmtree(objdir, ignore_errors=True)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
    print(OUT)
