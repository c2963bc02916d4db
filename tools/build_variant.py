"""Build an experimental libneuman_hip variant into ml-neuman_amd/lib/exp/ (git-ignored) with extra -D flags on ONE source file
(default csrc/mlp.hip):
    python tools/build_variant.py NAME [--src mlp_phase.hip] -DSOME_PROBE ..."""
import os, subprocess, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(ROOT, "ml-neuman_amd"))
import build as B
name, flags = sys.argv[1], sys.argv[2:]
src = "mlp.hip"
if "--src" in flags:
    i = flags.index("--src")
    src = flags[i + 1]
    del flags[i:i + 2]
out = os.path.join(ROOT, "ml-neuman_amd", "lib", "exp")
os.makedirs(out, exist_ok=True)
B.build(verbose=False)                                        # the other objects
obj = os.path.join(B.OBJ, f"{src[:-4]}_{name}.o")
subprocess.run([B.HIPCC] + B.FLAGS + flags + ["-c", os.path.join(B.CSRC, src), "-o", obj], check=True)
objs = [os.path.join(B.OBJ, s[:-4] + ".o") for s in sorted(os.listdir(B.CSRC)) if s.endswith(".hip") and s != src] + [obj]
subprocess.run([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(out, f"libneuman_hip_{name}.so")] + objs, check=True)
print(os.path.join(out, f"libneuman_hip_{name}.so"))
