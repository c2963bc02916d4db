"""The instruction-stream emitter of the hand-scheduled gfx950 kernels (tools/gen_f16t.py -> csrc/mlp_f16t_body.h): every instruction goes through
Asm.op, which keeps the books a hand-written stream needs -- the in-order LDS return queue (an `s_waitcnt lgkmcnt(n)` with the exact count before the
first use of a ds_read's destination), MFMA result hazards (wait states before a non-accumulate reader / writer of an MFMA destination), VALU -> MFMA /
permlane operand hazards -- and pads with s_nop where the schedule does not already cover them.  tests/test_generated_streams.py re-checks the LDS part
on the emitted text with an independent pass.

(Round 4 built a second stream on it, the 512-register two-sub-tile form of the i8x3 shading kernel; it equalled nerf_mlp_i8s_kernel's time and was
removed in round 5 -- profiles/r04_i8t_kernel.md keeps its write-up, the git history its generator.)"""
import os

import numpy as np

A_W = 128                                                  # first AccVGPR of the weight buffers (the no-weight-read probe skips loads into them)
PROBE_NO_VALU = os.environ.get("I8T_NO_VALU") == "1"      # timing probes of a stream without its VALU work / weight reads / MFMAs (garbage results)
PROBE_NO_WREAD = os.environ.get("I8T_NO_WREAD") == "1"
PROBE_NO_MFMA = os.environ.get("I8T_NO_MFMA") == "1"


def f32hex(x):
    return "0x%08x" % int(np.float32(x).view(np.uint32))


# ---- registers ------------------------------------------------------------------------------------------------------------------------------
def vr(i, n=1):
    return f"v{i}" if n == 1 else f"v[{i}:{i + n - 1}]"


def ar(i, n=1):
    return f"a{i}" if n == 1 else f"a[{i}:{i + n - 1}]"


def regs_of(tok):
    """'v[4:7]' -> ['v4','v5','v6','v7'];  's[2:3]', 'a12', 'vcc', '|v3|' ..."""
    tok = tok.strip().strip('|').lstrip('-')
    if tok in ("vcc", "exec", "m0", "scc"):
        return [tok]
    if len(tok) > 1 and tok[0] in "vas" and (tok[1].isdigit() or tok[1] == '['):
        if tok[1] == '[':
            lo, hi = tok[2:-1].split(':')
            return [f"{tok[0]}{k}" for k in range(int(lo), int(hi) + 1)]
        return [tok]
    return []


# ---- the emitter ----------------------------------------------------------------------------------------------------------------------------
class Asm:
    MFMA_D_STATES = int(os.environ.get("I8T_MFMA_D", "16"))          # MFMA result -> any reader / writer that is not the accumulate chain (8-pass XDL needs 12; margin)
    VALU_MFMA_STATES = 2        # VALU-written register -> MFMA operand
    VALU_PERM_STATES = 2        # VALU-written register -> v_permlane32_swap
    TRANS_STATES = 1

    def __init__(self):
        self.lines = []
        self.n = 0                       # issued instructions (wait states) so far
        self.lds = []                    # outstanding LDS reads, oldest first: sets of destination registers
        self.w_mfma = {}                 # reg -> state index of the last MFMA writing it
        self.w_valu = {}                 # reg -> state index of the last VALU write
        self.w_trans = {}
        self.stats = {'mfma': 0, 'valu': 0, 'ds': 0, 'nop_states': 0, 'waits': 0, 'salu': 0, 'vmem': 0}

    def raw(self, text):
        self.lines.append(text)

    def comment(self, text):
        self.lines.append("; " + text)

    def barrier_state(self):
        """a label / call boundary: forget nothing, but make every pending hazard safe on every path"""
        self.nop(self.MFMA_D_STATES)
        self.lines.append("s_waitcnt lgkmcnt(0)")
        self.lds = []

    def nop(self, states):
        while states > 0:
            k = min(states, 8)
            self.lines.append(f"s_nop {k - 1}")
            self.n += k
            self.stats['nop_states'] += k
            states -= k

    def wait_lds(self, keep):
        if len(self.lds) > keep:
            self.lines.append(f"s_waitcnt lgkmcnt({keep})")
            self.stats['waits'] += 1
            self.lds = self.lds[len(self.lds) - keep:] if keep else []

    def _need_lds(self, regs):
        regs = set(regs)
        last = -1
        for i, d in enumerate(self.lds):
            if d & regs:
                last = i
        if last >= 0:
            self.wait_lds(len(self.lds) - 1 - last)

    def op(self, kind, text, dst=(), src=(), chain=None):
        """kind: mfma | valu | trans | perm | ds_read | ds_write | vmem | salu | other.  dst / src: operand tokens.
        chain: for an MFMA the token of its C operand when it equals D (the accumulate chain: no wait states)."""
        d = [r for t in dst for r in regs_of(t)]
        s = [r for t in src for r in regs_of(t)]
        self._need_lds(d + s)
        need = 0
        touched = d + s
        for r in touched:
            if r in self.w_mfma:
                if kind == 'mfma' and chain is not None and r in regs_of(chain) and r in d:
                    continue
                need = max(need, self.w_mfma[r] + self.MFMA_D_STATES + 1 - self.n)
        if kind == 'mfma':
            for r in s:
                if r in self.w_valu:
                    need = max(need, self.w_valu[r] + self.VALU_MFMA_STATES + 1 - self.n)
        if kind == 'perm':
            for r in s + d:
                if r in self.w_valu:
                    need = max(need, self.w_valu[r] + self.VALU_PERM_STATES + 1 - self.n)
        if kind in ('valu', 'perm', 'mfma', 'vmem', 'ds_write', 'ds_read'):
            for r in s:
                if r in self.w_trans:
                    need = max(need, self.w_trans[r] + self.TRANS_STATES + 1 - self.n)
        if need > 0:
            self.nop(need)
        self.lines.append(text)
        if kind == 'mfma':
            for r in d:
                self.w_mfma[r] = self.n
                self.w_valu.pop(r, None)
            self.stats['mfma'] += 1
        elif kind in ('valu', 'trans', 'perm'):
            for r in d:
                self.w_valu[r] = self.n
                self.w_mfma.pop(r, None)
                if kind == 'trans':
                    self.w_trans[r] = self.n
                else:
                    self.w_trans.pop(r, None)
            self.stats['valu'] += 1
        elif kind == 'ds_read':
            self.lds.append(set(d))
            for r in d:
                self.w_mfma.pop(r, None)
                self.w_valu.pop(r, None)
            self.stats['ds'] += 1
        elif kind == 'salu':
            self.stats['salu'] += 1
        elif kind == 'vmem':
            self.stats['vmem'] += 1
        self.n += 1

    # -- shorthands
    def valu(self, text, dst, src):
        if PROBE_NO_VALU and not text.startswith(("v_add_u32", "v_mov_b32", "v_cmp")):
            return
        self.op('valu', text, dst, src)

    def salu(self, text):
        self.op('salu', text)

    def ds_read128(self, dst, addr, off):
        assert 0 <= off < 65536 and off % 16 == 0, off
        if PROBE_NO_WREAD and dst.startswith(f"a[{A_W}") or PROBE_NO_WREAD and dst.startswith(f"a[{A_W + 4}") or PROBE_NO_WREAD and dst.startswith(f"a[{A_W + 8}") \
                or PROBE_NO_WREAD and dst.startswith(f"a[{A_W + 12}"):
            return
        self.op('ds_read', f"ds_read_b128 {dst}, {addr} offset:{off}", [dst], [addr])

    def mfma_i8(self, acc, a, b, first):
        if PROBE_NO_MFMA:
            return
        c = "0" if first else acc
        self.op('mfma', f"v_mfma_i32_32x32x32_i8 {acc}, {a}, {b}, {c}", [acc], [a, b] + ([] if first else [acc]), chain=None if first else acc)

    def mfma_bf(self, acc, a, b):
        if PROBE_NO_MFMA:
            return
        self.op('mfma', f"v_mfma_f32_32x32x16_bf16 {acc}, {a}, {b}, {acc}", [acc], [a, b, acc], chain=acc)
