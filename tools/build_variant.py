"""Build an experimental libneuman_hip variant into ml-neuman_amd/lib/exp/ (git-ignored) with extra -D flags:
    python tools/build_variant.py NAME -DI8R_EXP_NO_E ..."""
import os, subprocess, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(ROOT, "ml-neuman_amd"))
import build as B
name, flags = sys.argv[1], sys.argv[2:]
out = os.path.join(ROOT, "ml-neuman_amd", "lib", "exp")
os.makedirs(out, exist_ok=True)
obj = os.path.join(B.OBJ, f"mlp_{name}.o")
subprocess.run([B.HIPCC] + B.FLAGS + flags + ["-c", os.path.join(B.CSRC, "mlp.hip"), "-o", obj], check=True)
objs = [os.path.join(B.OBJ, s[:-4] + ".o") for s in sorted(os.listdir(B.CSRC)) if s.endswith(".hip") and s != "mlp.hip"] + [obj]
subprocess.run([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(out, f"libneuman_hip_{name}.so")] + objs, check=True)
print(os.path.join(out, f"libneuman_hip_{name}.so"))
