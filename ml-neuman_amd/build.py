"""Build libneuman_hip.so in-tree for gfx950 (hipcc cross-compiles without a GPU).

    python ml-neuman_amd/build.py            # incremental
    python ml-neuman_amd/build.py --force

One object per .hip file (compiled in parallel), linked into ml-neuman_amd/lib/libneuman_hip.so.
-ffp-contract=off: the reference's elementwise torch ops round a*b and +c separately; fused
multiply-adds appear only where the source spells fmaf().
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "lib", "libneuman_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -fno-slp-vectorize: -O3 otherwise packs adjacent scalar f32 multiplies / adds into v_pk_*_f32, which cost ~20 cycles extra per
# instruction beside MFMAs on gfx950 (MI355X_MICROARCH.md) -- the MLP epilogues are written with scalar f32 on purpose
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-slp-vectorize", "-Wall", "-Wno-unused-function"]


def _newer(target, deps):
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(d) <= t for d in deps)


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    srcs = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(HERE, "..", "include", "neuman_hip.h"))
    jobs = []
    for s in srcs:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJ, s[:-4] + ".o")
        if force or not _newer(obj, [src] + hdrs):
            jobs.append((src, obj))

    def cc(job):
        src, obj = job
        cmd = [HIPCC] + FLAGS + ["-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        return job, r

    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        for (src, obj), r in ex.map(cc, jobs):
            if verbose and (r.stderr.strip() or r.returncode):
                sys.stderr.write(r.stderr)
            if r.returncode:
                raise RuntimeError(f"hipcc failed on {src}")
    objs = [os.path.join(OBJ, s[:-4] + ".o") for s in srcs]
    if jobs or force or not os.path.exists(LIB):
        r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs, capture_output=True, text=True)
        if r.returncode:
            sys.stderr.write(r.stderr)
            raise RuntimeError("link failed")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
