"""CPU emulation (float64 numpy, no device): would a backward-data chain in the RENDERING kernels' cheaper arithmetic -- dZ quantised per row to 16-bit
fixed point, weights to 16-bit per column, exact integer products (what nerf_mlp_i8s_kernel does to activations, 1.55 ns per evaluation against the
4.2 of the split-bf16 chain) -- hold the gates of tests/test_hip_train16.py::test_backward_net16_against_float64 (bias gradients = column sums of dZ
within 5e-5 of each layer's largest; the default chain measures 2e-5)?  Same net (synthetic.make_joiner(1)), same batch sizes, same d_raw scale as the
test.  Output of the committed run: profiles/r06_train_experiments.md section 3.

    python tools/train_int16_emulation.py
"""
import sys, numpy as np, torch
sys.path.insert(0, __import__('os').path.join(__import__('os').path.dirname(__import__('os').path.abspath(__file__)), '..', 'ml-neuman_amd'))
from neuman_hip import synthetic
torch.manual_seed(0)
net = synthetic.make_joiner(1)
P = [p.detach().double().numpy() for p in net.nerf.ordered_params()]
W = P[0:16:2]; B = P[1:16:2]
Wv, Wf, wa, Wr = P[16], P[18], P[20][0], P[22]
def pe(x, nf):
    out = [x]
    for k in range(nf):
        out += [np.sin(x * 2.0**k), np.cos(x * 2.0**k)]
    return np.concatenate(out, -1)
print(net.pos_pe.N_freqs, net.pos_pe.max_freq, net.dir_pe.N_freqs, [w.shape for w in W])
def run(n, seed, mode):
    rng = np.random.default_rng(seed)
    pts = rng.random((n, 3)) * 2 - 1
    dirs = rng.standard_normal((n, 3)); dirs /= np.linalg.norm(dirs, axis=-1, keepdims=True)
    e = pe(pts, net.pos_pe.N_freqs); ed = pe(dirs, net.dir_pe.N_freqs)
    h = e; masks = []
    for i in range(8):
        if i == 5: h = np.concatenate([e, h], -1)
        z = h @ W[i].T + B[i]; masks.append(z > 0); h = np.maximum(z, 0)
    feat = h @ Wf.T + P[19]
    hv = np.maximum(np.concatenate([feat, ed], -1) @ Wv.T + P[17], 0)
    d_raw = rng.standard_normal((n, 4)) * 2e-5
    d_hv = (d_raw[:, :3] @ Wr) * (hv > 0)
    d_feat = d_hv @ Wv[:, :256]
    def q_rows(x, bits):
        if bits is None: return x
        if bits == 'fp16': return x.astype(np.float16).astype(np.float64) if False else (np.float64(1) * (x * 2**14 / np.abs(x).max()).astype(np.float16).astype(np.float64) * np.abs(x).max() / 2**14)
        m = np.abs(x).max(-1, keepdims=True) / (2**(bits-1) - 129); m[m == 0] = 1
        return np.rint(x / m) * m
    def q_w(w, bits):
        if bits is None or bits == 'fp16': return w
        m = np.abs(w).max(0, keepdims=True) / (2**(bits-1) - 129)   # per input feature of the transposed product (rows of W^T)
        return np.rint(w / m) * m
    errs = []
    d = (d_feat @ Wf + d_raw[:, 3:4] * wa[None]) * masks[7]
    dq = (q_rows(d_feat, mode) @ q_w(Wf, mode) + d_raw[:, 3:4] * wa[None]) * masks[7]
    for i in range(7, -1, -1):
        errs.append((np.abs(dq.sum(0) - d.sum(0)).max() / np.abs(d.sum(0)).max(), np.abs(dq - d).max() / np.abs(d).max()))
        if i:
            d = (d @ W[i][:, -256:]) * masks[i - 1]
            dq = (q_rows(dq, mode) @ q_w(W[i][:, -256:], mode)) * masks[i - 1]
    return errs
for mode in (16, 'fp16'):
    for n in (1000, 4224):
        er = run(n, n + 1, mode)
        print(mode, n, 'colsum rel err per layer:', ' '.join(f'{a:.1e}' for a, b in er), '| elem:', ' '.join(f'{b:.1e}' for a, b in er))
