"""one-off: the LDS state the second tile of workgroup 0 starts from, generated stream vs plain-HIP build"""
import os, subprocess, sys, json
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import i8t_debug as D
if __name__ == "__main__":
    if len(sys.argv) > 1:
        np.savez(sys.argv[1], r0=D.run('posenc', 131072, [99], 0)["state99"], r1=D.run('posenc', 131072, [99], 1)["state99"])
        sys.exit(0)
    m0 = D.run('posenc', 131072, [99], 0)["state99"]
    m1 = D.run('posenc', 131072, [99], 1)["state99"]
    subprocess.run([sys.executable, __file__, "/tmp/ref2.npz"], check=True, env=dict(os.environ, NEUMAN_HIP_LIB=D.REF))
    r = np.load("/tmp/ref2.npz")
    # which part of the encodings differs: word index of a wave's 24 KB -> region, chunk, hi / lo half
    for w in range(1):
        am = m1[64 * w:64 * w + 64, :96].reshape(-1)
        bm = r["r1"][64 * w:64 * w + 64, :96].reshape(-1)
        bad = am != bm
        reg = {"posA": (0, 2048), "posB": (2048, 4096), "dirA": (4096, 5120), "dirB": (5120, 6144)}
        out = {}
        for k, (lo, hi) in reg.items():
            sub = bad[lo:hi].reshape(-1, 2, 32, 4)            # [chunk][hi|lo][row][dword]
            out[k] = {"frac": float(sub.mean()), "per_chunk": [round(float(x), 2) for x in sub.mean((1, 2, 3))], "hi_lo": [round(float(x), 2) for x in sub.mean((0, 2, 3))],
                      "per_dword": [round(float(x), 2) for x in sub.mean((0, 1, 2))], "rows_bad": int(sub.any((0, 1, 3)).sum())}
        print(json.dumps(out))
        k = int(np.nonzero(bad)[0][0])
        print("first differing word", k, hex(int(am[k]) & 0xffffffff), hex(int(bm[k]) & 0xffffffff), "next", [hex(int(x) & 0xffffffff) for x in am[k:k + 4]], [hex(int(x) & 0xffffffff) for x in bm[k:k + 4]])
    for name, a, b in (("tile0", m0, r["r0"]), ("tile256", m1, r["r1"])):
        print(json.dumps({"which": name, "pe_equal": bool((a[:, :96] == b[:, :96]).all()), "pe_frac": float((a[:, :96] == b[:, :96]).mean()),
                          "ring_equal": bool((a[:, 96:104] == b[:, 96:104]).all()), "ring_frac": float((a[:, 96:104] == b[:, 96:104]).mean()),
                          "bias_equal": bool((a[:, 104:114] == b[:, 104:114]).all()), "off_slot_mine": [int(a[0, 114]), int(a[0, 115])], "off_slot_ref": [int(b[0, 114]), int(b[0, 115])],
                          "vars_mine_lane0": [int(x) for x in a[0, 116:130]], "vars_ref_lane0": [int(x) for x in b[0, 116:130]], "vars_equal": bool((a[:, 116:130] == b[:, 116:130]).all()),
                          "vars_mine_lane70": [int(x) for x in a[70, 116:130]], "vars_ref_lane70": [int(x) for x in b[70, 116:130]],
                          "pe_per_wave": [float((a[64 * w:64 * w + 64, :96] == b[64 * w:64 * w + 64, :96]).mean()) for w in range(4)]}))
