#!/usr/bin/env python3
"""Generator of ml-neuman_amd/csrc/mlp_i8t_body.h: stages 0 .. 10 of nerf_mlp_i8t_kernel (csrc/mlp_i8t.hip) as ONE hand-allocated gfx950
instruction stream -- 4 waves per CU with all 512 registers each, two 32-sample sub-tiles (A, B) per wave.

    python tools/gen_i8t.py            -> writes the header;  --check: regenerate and compare with the file in the tree (tests/test_mlp_pack.py)

Why a generator: hipcc cannot allocate this register file (its plain-HIP form of the same kernel, -DNM_I8T_HIP, spills 193 registers and
shuttles thousands of values between the two halves of the file), and the placement of every non-MFMA instruction into the gaps between
MFMAs is the point of the kernel.  The arithmetic is nerf_mlp_i8s_kernel's, instruction for instruction (csrc/mlp_i8as.h); the plain-HIP
form stays in the tree as the stage-by-stage reference (nm_mlp_forward_i8t_debug).

Register file of a lane (the compiler keeps v0..v3):
    a0..a127     the resident inputs of the running stage: X[A].h[0..7], X[A].l[0..7], X[B].h[0..7], X[B].l[0..7] (4 registers per k-step)
    a128..a143   two weight-fragment buffers (hi, lo) read from the LDS ring one k-step ahead
    a144..a255   7 parked output blocks (the stage's 16 output blocks of 16 registers do not fit the VALU half of the file)
    v4..v11      inputs (ring read base, bias base, encoding base, copy offset, the two output addresses)
    v12..v29     scalars of the wave's two sub-tiles (row scales, running maxima, sigma ...) and temporaries
    v30..v45     16 temporaries: the bias values of the block being dequantised / encoding operands / unparked values
    v46..v109    four int32 accumulator chains: A.cross, A.hihi, B.cross, B.hihi
    v110..v253   9 output blocks
emit() keeps the books the assembler does not: the LDS return queue (s_waitcnt lgkmcnt counts), and the software-managed wait states of
gfx950 (MFMA result -> any other reader / writer, VALU-written MFMA operands, v_permlane32_swap operands, transcendental results).
"""
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
OUT = os.environ.get("I8T_OUT") or os.path.join(ROOT, "ml-neuman_amd", "csrc", "mlp_i8t_body.h")

# ---- layout constants (csrc/mlp_layout.h, csrc/mlp_i8t.hip) ------------------------------------------------------------------------------
STEP = 2048


def stage_shape8(s):
    return {0: (8, 0, 4), 5: (8, 8, 4), 8: (9, 8, 0), 9: (4, 8, 2), 10: (1, 4, 0)}.get(s, (8, 8, 0))


def stage_b_off(s):
    return sum(stage_shape8(i)[0] * 32 for i in range(s))


K_BIAS_FLOATS = stage_b_off(11)
K_WEIGHT_BYTES8 = sum(n * (a + b) * STEP for n, a, b in (stage_shape8(i) for i in range(11)))
POS_B = 8192            # sub-tile B's position encodings behind A's
DIR_A = 16384           # direction encodings behind both position blocks
DIR_B = DIR_A + 4096
SLOT_BYTES = 16384
C_INV = float(np.float32(32639.0) / np.float32(32767.0))
C_SCALE = float(np.float32(1.0) / np.float32(32639.0))


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


V_RDBASE, V_BIAS, V_PE, V_CP0, V_OUTA, V_OUTB = 4, 5, 6, 7, 8, 10
V_RD, V_CP1 = 12, 13
V_SX, V_SXIN, V_M, V_SIG, V_INV = (16, 17), (18, 19), (20, 21), (22, 23), (24, 25)
V_BIASST, V_T0, V_T1, V_T2 = 26, 27, 28, 29
V_VB = 30                                      # 16 temporaries
V_ACC = {('A', 'c'): 46, ('A', 'h'): 62, ('B', 'c'): 78, ('B', 'h'): 94}
V_F = 110                                      # 9 blocks of 16
V_DBGOFF = 254
A_X = {('A', 'h'): 0, ('A', 'l'): 32, ('B', 'h'): 64, ('B', 'l'): 96}
A_W = 128
A_PARK = 144
S_IMG, S_RING0, S_OFF, S_SLOT, S_USIG, S_UR, S_UG, S_UB, S_SIGSC, S_DBGST, S_DBG, S_KAPPA = 36, 38, 39, 40, 41, 42, 43, 44, 45, 46, 50, 49
S_SEL_LO, S_SEL_HI, S_C128, S_REFILL, S_P, S_TMP, S_RET, S_RET2, S_ST, S_SAVE, S_KOFF, S_RDOFF = 52, 53, 54, 55, 56, 58, 60, 62, 64, 66, 68, 59
H = ('A', 'B')
PROBE_NO_VALU = os.environ.get("I8T_NO_VALU") == "1"      # timing probes: the stream without its VALU work / without its weight reads (garbage results)
PROBE_NO_WREAD = os.environ.get("I8T_NO_WREAD") == "1"
PROBE_NO_MFMA = os.environ.get("I8T_NO_MFMA") == "1"
PER_GAP = int(os.environ.get("I8T_PER_GAP", "5"))          # non-MFMA instructions placed behind every MFMA of an i8 block


class Slot:
    """where an output block of 16 registers lives"""

    def __init__(self, kind, base):
        self.kind, self.base = kind, base

    def r(self, i, n=1):
        return vr(self.base + i, n) if self.kind == 'v' else ar(self.base + i, n)


VSLOT = [Slot('v', V_F + 16 * i) for i in range(9)]
PSLOT = [Slot('a', A_PARK + 16 * i) for i in range(7)]
# homes of the 16 output blocks of a 256-wide stage: blocks 0..2 of both sub-tiles and A's block 3 are parked
HOME = {}
_p = 0
for b in range(8):
    for h in H:
        if b < 3 or (b == 3 and h == 'A'):
            HOME[(h, b)] = PSLOT[_p]
            _p += 1
_v = 0
for b in range(3, 8):
    for h in H:
        if (h, b) not in HOME:
            HOME[(h, b)] = VSLOT[_v]
            _v += 1
LAND = {'A': HOME[('A', 7)], 'B': HOME[('B', 7)]}        # where a block that will be parked is dequantised first
HOME9 = {(h, b): VSLOT[2 * b + (h == 'B')] for b in range(4) for h in H}      # stage 9: four blocks per sub-tile, all in the VALU half


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


class Gen:
    def __init__(self):
        self.A = Asm()
        self.blk = 0                     # flat ring block index within the tile (static)

    # ---------------------------------------------------------------------------------------------------------------------------------------
    # the weight ring
    # ---------------------------------------------------------------------------------------------------------------------------------------
    @staticmethod
    def block_steps(i):
        i %= 83
        return 4 if i < 8 else (8 if i < 82 else 4)

    def ring_start(self):
        """top of ring block self.blk: its read pointer.  The block was handed over inside its predecessor (ring_handover): its data is
        visible to every wave, s[S_RDOFF] is its slot's offset."""
        A = self.A
        A.comment(f"---- ring block {self.blk}")
        A.valu(f"v_add_u32 {vr(V_RD)}, s{S_RDOFF}, {vr(V_RDBASE)}", [vr(V_RD)], [vr(V_RDBASE)])

    def ring_handover(self):
        """inside block i = self.blk, two k-steps before its end: hand-over of block i + 1.  Every wave waits for its own pieces of block
        i + 1 (issued one block ago: nothing younger is in flight) and meets the others: the block's data is visible from here on, and
        everybody is at least this far into block i, i.e. done with block i - 1, whose slot takes the pieces of block i + 2 -> their copy plan.
        s[S_SLOT] = slot of the block handed over next."""
        A = self.A
        i = self.blk
        A.raw("s_waitcnt vmcnt(0)")
        A.raw("s_barrier")
        A.salu(f"s_lshl_b32 s{S_RDOFF}, s{S_SLOT}, 14")                        # block i + 1 reads here
        A.salu(f"s_add_u32 s{S_TMP}, s{S_SLOT}, 1")
        A.salu(f"s_cmp_eq_u32 s{S_SLOT}, 2")
        A.salu(f"s_cselect_b32 s{S_SLOT}, 0, s{S_TMP}")                        # slot of block i + 2 = the one block i - 1 gave up
        A.salu(f"s_lshl_b32 s{S_TMP}, s{S_SLOT}, 14")
        A.salu(f"s_add_u32 s{S_REFILL}, s{S_RING0}, s{S_TMP}")
        A.salu(f"s_add_u32 s{S_P}, s{S_IMG}, s{S_OFF}")
        A.salu(f"s_addc_u32 s{S_P + 1}, s{S_IMG + 1}, 0")
        n2 = self.block_steps(i + 2)
        return self.copy_fillers([('piece', j, n2) for j in range(2 * n2 // 4)])

    def copy_piece(self, j, n2, last):
        A = self.A
        A.salu(f"s_add_u32 m0, s{S_REFILL}, {j * 4096}")
        A.raw("s_nop 0")
        A.n += 1
        A.op('vmem', f"global_load_lds_dwordx4 {vr(V_CP0 if j == 0 else V_CP1 + j - 1)}, s[{S_P}:{S_P + 1}]", [], [vr(V_CP0 if j == 0 else V_CP1 + j - 1)])
        if last:
            A.salu(f"s_add_u32 s{S_OFF}, s{S_OFF}, {n2 * STEP}")
            A.salu(f"s_cmp_eq_u32 s{S_OFF}, {K_WEIGHT_BYTES8}")
            A.salu(f"s_cselect_b32 s{S_OFF}, 0, s{S_OFF}")

    def copy_fillers(self, plan):
        """-> list of closures, one per piece"""
        out = []
        for k, (_, j, n2) in enumerate(plan):
            out.append(lambda j=j, n2=n2, last=(k == len(plan) - 1): self.copy_piece(j, n2, last))
        return out

    # ---------------------------------------------------------------------------------------------------------------------------------------
    # pieces of arithmetic (each returns a list of closures = filler instructions in program order)
    # ---------------------------------------------------------------------------------------------------------------------------------------
    def bias_reads(self, bias_addr, off_bytes):
        """the 16 bias values of a block for this lane (dequant16 / bias16: vf4 at bias_blk + 8 q) -> V_VB[0..15]"""
        return [lambda q=q: self.A.ds_read128(vr(V_VB + 4 * q, 4), bias_addr, off_bytes + 32 * q) for q in range(4)]

    def combine(self, h, slot):
        """t = 256 * hihi + cross, exact -> the 16 registers of `slot` (a VALU block)"""
        assert slot.kind == 'v'
        ac, ah = V_ACC[(h, 'c')], V_ACC[(h, 'h')]
        return [lambda r=r: self.A.valu(f"v_lshl_add_u32 {slot.r(r)}, {vr(ah + r)}, 8, {vr(ac + r)}", [slot.r(r)], [vr(ah + r), vr(ac + r)]) for r in range(16)]

    def dequant(self, h, slot, relu, with_max=True, regs=range(16)):
        """f = fma(float(t), sxin, bias) in place; running row maximum (max of f under ReLU, of |f| otherwise).  Emitted by kind -- all the
        conversions, then the multiply-adds, then the maxima -- so that no instruction follows the one it depends on."""
        hi = H.index(h)
        out = []
        regs = list(regs)
        for r in regs:
            out.append(lambda r=r: self.A.valu(f"v_cvt_f32_i32 {slot.r(r)}, {slot.r(r)}", [slot.r(r)], [slot.r(r)]))
        for r in regs:
            out.append(lambda r=r: self.A.valu(f"v_fma_f32 {slot.r(r)}, {slot.r(r)}, {vr(V_SXIN[hi])}, {vr(V_VB + r)}", [slot.r(r)],
                                               [slot.r(r), vr(V_SXIN[hi]), vr(V_VB + r)]))
        if with_max:
            for r in regs:
                if r & 1:
                    out.append(self.max2(hi, slot.r(r - 1), slot.r(r), relu))
        return out

    def max2(self, hi, x, y, relu):
        m = vr(V_M[hi])
        if relu:
            return lambda: self.A.valu(f"v_max3_f32 {m}, {m}, {x}, {y}", [m], [m, x, y])
        return lambda: self.A.valu(f"v_max3_f32 {m}, {m}, |{x}|, |{y}|", [m], [m, x, y])

    def park(self, src, dst):
        assert src.kind == 'v' and dst.kind == 'a'
        return [lambda r=r: self.A.valu(f"v_accvgpr_write_b32 {dst.r(r)}, {src.r(r)}", [dst.r(r)], [src.r(r)]) for r in range(16)]

    def unpark(self, src, dst_base):
        return [lambda r=r: self.A.valu(f"v_accvgpr_read_b32 {vr(dst_base + r)}, {src.r(r)}", [vr(dst_base + r)], [src.r(r)]) for r in range(16)]

    def row_scale(self, hi):
        """M = row maximum over both lane halves; inv = M > 0 ? C_INV * rcp(M) : 0; sx = M > 0 ? M * C_SCALE : 1   (mlp_i8as.h row_max, inv_of, scale_of)"""
        A = self.A
        m, inv, sx = vr(V_M[hi]), vr(V_INV[hi]), vr(V_SX[hi])
        t0, t1 = vr(V_T0), vr(V_T1)
        A.valu(f"v_mov_b32 {t0}, {m}", [t0], [m])
        A.valu(f"v_mov_b32 {t1}, {m}", [t1], [m])
        A.op('perm', f"v_permlane32_swap_b32 {t0}, {t1}", [t0, t1], [t0, t1])
        A.valu(f"v_max_f32 {m}, {t0}, {t1}", [m], [t0, t1])
        A.op('trans', f"v_rcp_f32 {t0}, {m}", [t0], [m])
        A.valu(f"v_cmp_lt_f32 vcc, 0, {m}", ['vcc'], [m])
        A.valu(f"v_mul_f32 {t0}, {f32hex(C_INV)}, {t0}", [t0], [t0])
        A.valu(f"v_mul_f32 {t1}, {f32hex(C_SCALE)}, {m}", [t1], [m])
        A.valu(f"v_cndmask_b32 {inv}, 0, {t0}, vcc", [inv], [t0, 'vcc'])
        A.valu(f"v_cndmask_b32 {sx}, 1.0, {t1}, vcc", [sx], [t1, 'vcc'])

    def quant(self, h, b, base, relu):
        """16 outputs in v[base ..] -> X[h].h[b], X[h].l[b]   (mlp_i8as.h quant16; the block's registers are consumed).  By kind again: the
        16 multiplies, the 8 packed conversions, the 8 limb offsets, the 8 byte shuffles (in place), the 8 moves into the input file."""
        A = self.A
        hi = H.index(h)
        inv = vr(V_INV[hi])
        for r in range(16):
            A.valu(f"v_mul_f32 {vr(base + r)}, {vr(base + r)}, {inv}" + (" clamp" if relu else ""), [vr(base + r)], [vr(base + r), inv])
        for i in range(8):
            A.valu(f"v_cvt_pknorm_i16_f32 {vr(base + 2 * i)}, {vr(base + 2 * i)}, {vr(base + 2 * i + 1)}", [vr(base + 2 * i)], [vr(base + 2 * i), vr(base + 2 * i + 1)])
        for i in range(8):
            A.valu(f"v_pk_add_i16 {vr(base + 2 * i + 1)}, {vr(base + 2 * i)}, s{S_C128}", [vr(base + 2 * i + 1)], [vr(base + 2 * i)])
        xh, xl = A_X[(h, 'h')] + 4 * b, A_X[(h, 'l')] + 4 * b
        # P[i] sits in register 2 i, Y[i] = P[i] + 128 in 2 i + 1;  lo[k] = perm(P[2k+1], P[2k]) -> register 4 k, hi[k] = perm(Y[2k+1], Y[2k]) -> 4 k + 1
        for k in range(4):
            A.valu(f"v_perm_b32 {vr(base + 4 * k)}, {vr(base + 4 * k + 2)}, {vr(base + 4 * k)}, s{S_SEL_LO}", [vr(base + 4 * k)], [vr(base + 4 * k + 2), vr(base + 4 * k)])
            A.valu(f"v_perm_b32 {vr(base + 4 * k + 1)}, {vr(base + 4 * k + 3)}, {vr(base + 4 * k + 1)}, s{S_SEL_HI}", [vr(base + 4 * k + 1)],
                   [vr(base + 4 * k + 3), vr(base + 4 * k + 1)])
        for k in range(4):
            A.valu(f"v_accvgpr_write_b32 {ar(xl + k)}, {vr(base + 4 * k)}", [ar(xl + k)], [vr(base + 4 * k)])
            A.valu(f"v_accvgpr_write_b32 {ar(xh + k)}, {vr(base + 4 * k + 1)}", [ar(xh + k)], [vr(base + 4 * k + 1)])

    # ---------------------------------------------------------------------------------------------------------------------------------------
    # blocks
    # ---------------------------------------------------------------------------------------------------------------------------------------
    def i8_block(self, nsteps, fillers, per_gap=None, pre=(), urgent=None, prefetched=False, prefetch_next=False):
        """one ring block of `nsteps` limb k-steps for both sub-tiles: four accumulator chains (A.cross, A.hihi, B.cross, B.hihi), every
        weight fragment read from the ring once for six MFMAs; `fillers` (closures) go into the gaps between MFMAs.  pre: instructions that
        must precede the first MFMA; urgent[s]: instructions that must be issued before the MFMAs of step s + 1 (the requantisation of the
        previous stage's block s + 1 = this stage's input of step s + 1): spread over step s.  The hand-over of the NEXT ring block happens
        two steps before the end, the copies of the block after it ride in the last two steps; prefetched: the weights of step 0 were read by
        the previous block (prefetch_next, after its hand-over)."""
        A = self.A
        per_gap = per_gap or PER_GAP
        self.ring_start()
        fill_iter = iter(list(fillers))
        urgent = urgent or {}
        wb = lambda s, part: ar(A_W + 8 * (s & 1) + 4 * part, 4)                 # noqa: E731
        rd = vr(V_RD)
        if not prefetched:
            A.ds_read128(wb(0, 0), rd, 0)
            A.ds_read128(wb(0, 1), rd, 1024)
        for f in pre:
            f()
        copies = []
        for s in range(nsteps):
            gap_extra = []
            if s == nsteps - 2:
                copies = self.ring_handover()
                if prefetch_next:
                    gap_extra.append(lambda: A.valu(f"v_add_u32 {vr(V_T2)}, s{S_RDOFF}, {vr(V_RDBASE)}", [vr(V_T2)], [vr(V_RDBASE)]))
            if s + 1 < nsteps:
                gap_extra.append(lambda s=s: A.ds_read128(wb(s + 1, 0), rd, (s + 1) * STEP))
                gap_extra.append(lambda s=s: A.ds_read128(wb(s + 1, 1), rd, (s + 1) * STEP + 1024))
            elif prefetch_next:                       # the last step: step 0 of the next block (its buffer was free after the previous step's MFMAs)
                gap_extra.append(lambda: A.ds_read128(wb(0, 0), vr(V_T2), 0))
                gap_extra.append(lambda: A.ds_read128(wb(0, 1), vr(V_T2), 1024))
            if s >= nsteps - 2:                       # this block's copies: half in each of the last two steps
                k = (len(copies) + 1) // 2 if s == nsteps - 2 else len(copies)
                gap_extra += copies[:k]
                copies = copies[k:]
            urg = list(urgent.get(s, ()))
            per_urg = (len(urg) + 5) // 6
            first = s == 0
            seq = []
            for h in H:
                ac, ah = vr(V_ACC[(h, 'c')], 16), vr(V_ACC[(h, 'h')], 16)
                xl, xh = ar(A_X[(h, 'l')] + 4 * s, 4), ar(A_X[(h, 'h')] + 4 * s, 4)
                seq.append((ac, wb(s, 0), xl, first))
                seq.append((ah, wb(s, 0), xh, first))
                seq.append((ac, wb(s, 1), xh, False))
            for k, (acc, a, b, fst) in enumerate(seq):
                A.mfma_i8(acc, a, b, fst)
                budget = per_gap
                while gap_extra and budget > 0:
                    gap_extra.pop(0)()
                    budget -= 1
                for _ in range(per_urg):
                    if urg:
                        urg.pop(0)()
                        budget -= 1
                while budget > 0:
                    f = next(fill_iter, None)
                    if f is None:
                        break
                    f()
                    budget -= 1
            for f in gap_extra + urg:
                f()
        for f in fill_iter:                       # whatever did not fit under the MFMAs
            f()
        self.blk += 1

    def enc_block(self, chains, pe_off, steps_per_chain, fillers=()):
        """one ring block of split-bf16 steps over the encodings: chains = [(sub-tile, slot, first weight step in the block)], every chain
        runs `steps_per_chain` steps t = 0 .. with the operands of chunk pair t of the sub-tile's encoding rows; the chains are interleaved
        step by step (a chain's three MFMAs per step depend on each other).  The next ring block is handed over before the last step."""
        A = self.A
        self.ring_start()
        rd, pe = vr(V_RD), vr(V_PE)
        fill = list(fillers)
        wsteps = sorted(set(ws for _, _, ws in chains))
        assert len(wsteps) * steps_per_chain <= 8
        for t in range(steps_per_chain):
            if t == steps_per_chain - 1:
                fill = self.ring_handover() + fill
            fill_iter = iter(fill)
            # operands of this step: x hi / lo of A and of B -> V_VB[0..15]
            for hi, h in enumerate(H):
                base = pe_off[h] + t * 2048
                A.ds_read128(vr(V_VB + 8 * hi, 4), pe, base)
                A.ds_read128(vr(V_VB + 8 * hi + 4, 4), pe, base + 512)
            for wi, ws in enumerate(wsteps):                                   # weights of every chain group at this step (<= 2 groups in flight)
                A.ds_read128(ar(A_W + 8 * (wi & 1), 4), rd, (ws + t) * STEP)
                A.ds_read128(ar(A_W + 8 * (wi & 1) + 4, 4), rd, (ws + t) * STEP + 1024)
                grp = [c for c in chains if c[2] == ws]
                for part in range(3):
                    for h, slot, _ in grp:
                        hi = H.index(h)
                        xh, xl = vr(V_VB + 8 * hi, 4), vr(V_VB + 8 * hi + 4, 4)
                        whi, wlo = ar(A_W + 8 * (wi & 1), 4), ar(A_W + 8 * (wi & 1) + 4, 4)
                        a, b = [(whi, xl), (wlo, xh), (whi, xh)][part]
                        A.mfma_bf(slot.r(0, 16), a, b)
                        for _ in range(3):
                            f = next(fill_iter, None)
                            if f is not None:
                                f()
            fill = list(fill_iter)
        for f in fill:
            f()
        self.blk += 1

    # ---------------------------------------------------------------------------------------------------------------------------------------
    # stage tails
    # ---------------------------------------------------------------------------------------------------------------------------------------
    def collect(self, fn):
        """run an emitting function with the emitter in recording mode -> the list of closures that replay its instructions one by one"""
        rec = []
        A = self.A
        real_op = A.op
        A.op = lambda *a, **k: rec.append(lambda a=a, k=k: real_op(*a, **k))
        try:
            fn()
        finally:
            A.op = real_op
        return rec

    def finish(self, homes, nblk, relu, max_now):
        """end of a stage: the row maximum (recomputed over all outputs when max_now: stages whose encoding part came last) and the row
        scales.  The requantisation itself is DEFERRED: -> plan[b] = the instructions that turn output block b of both sub-tiles into
        X[.].h[b], X[.].l[b]; the next stage issues plan[0] before its first MFMA and plan[s + 1] under the MFMAs of its first block's step s
        (its input of step s + 1), so that only a sixteenth of the requantisation is exposed."""
        A = self.A
        for hi, h in enumerate(H):
            if max_now:
                A.valu(f"v_mov_b32 {vr(V_M[hi])}, 0", [vr(V_M[hi])], [])
                for b in range(nblk):
                    slot = homes[(h, b)]
                    if slot.kind == 'a':
                        for f in self.unpark(slot, V_VB):
                            f()
                        src = lambda r: vr(V_VB + r)            # noqa: E731
                    else:
                        src = lambda r, slot=slot: slot.r(r)    # noqa: E731
                    for r in range(1, 16, 2):
                        self.max2(hi, src(r - 1), src(r), relu)()
            self.row_scale(hi)
        plan = []
        for b in range(nblk):
            def one(b=b):
                for h in H:
                    slot = homes[(h, b)]
                    if slot.kind == 'a':
                        for f in self.unpark(slot, V_VB):
                            f()
                        base = V_VB
                    else:
                        base = slot.base
                    self.quant(h, b, base, relu)
            plan.append(self.collect(one))
        return plan

    def quant_plan(self, homes, nblk, relu):
        """the deferred requantisation of a PREVIOUS stage as seen from the next one (same instructions as finish() returns)"""
        plan = []
        for b in range(nblk):
            def one(b=b):
                for h in H:
                    slot = homes[(h, b)]
                    if slot.kind == 'a':
                        for f in self.unpark(slot, V_VB):
                            f()
                        base = V_VB
                    else:
                        base = slot.base
                    self.quant(h, b, base, relu)
            plan.append(self.collect(one))
        return plan

    @staticmethod
    def first_block_args(plan):
        return dict(pre=plan[0], urgent={s: plan[s + 1] for s in range(len(plan) - 1)})

    def sxin(self, kappa_off_reg):
        """sxin = sx * (256 * kappa[st]) for both sub-tiles; kappa read from LDS at kappa base + kappa_off_reg (a VGPR holding the address)"""
        A = self.A
        A.op('ds_read', f"ds_read_b32 {vr(V_T2)}, {kappa_off_reg}", [vr(V_T2)], [kappa_off_reg])
        A.valu(f"v_mul_f32 {vr(V_T2)}, 0x43800000, {vr(V_T2)}", [vr(V_T2)], [vr(V_T2)])
        for hi in range(2):
            A.valu(f"v_mul_f32 {vr(V_SXIN[hi])}, {vr(V_SX[hi])}, {vr(V_T2)}", [vr(V_SXIN[hi])], [vr(V_SX[hi]), vr(V_T2)])
            A.valu(f"v_mov_b32 {vr(V_M[hi])}, 0", [vr(V_M[hi])], [])

    def kappa_addr(self, st_sgpr=None, st=None):
        """V_T0 <- LDS address of kappa[st]"""
        A = self.A
        if st is not None:
            A.valu(f"v_mov_b32 {vr(V_T0)}, s{S_KAPPA}", [vr(V_T0)], [])
            A.valu(f"v_add_u32 {vr(V_T0)}, {4 * st}, {vr(V_T0)}", [vr(V_T0)], [vr(V_T0)])
        else:
            A.salu(f"s_lshl_b32 s{S_TMP}, s{st_sgpr}, 2")
            A.salu(f"s_add_u32 s{S_TMP}, s{S_TMP}, s{S_KAPPA}")
            A.valu(f"v_mov_b32 {vr(V_T0)}, s{S_TMP}", [vr(V_T0)], [])
        return vr(V_T0)

    # ---------------------------------------------------------------------------------------------------------------------------------------
    # stages
    # ---------------------------------------------------------------------------------------------------------------------------------------
    def hidden_blocks(self, bias_addr, bias_off, relu, nblk=8, homes=HOME, extra_first=(), first_args=None, after_first=None):
        """the i8 blocks of a 256-wide (or, stage 9, 128-wide) stage.  The dequantisation of block b - 1 -- and the move of a block that
        does not live in the VALU half of the file to its parking place -- ride under the MFMAs of block b; the last block's follows its
        combine."""
        pending = list(extra_first)              # fillers that ride under the next block
        for b in range(nblk):
            self.i8_block(8, pending, prefetched=b > 0, prefetch_next=b < nblk - 1, **((first_args or {}) if b == 0 else {}))
            if b == 0 and after_first is not None:
                after_first()
            pending = []
            land = {}
            for h in H:
                home = homes[(h, b)]
                land[h] = home if home.kind == 'v' else LAND[h]
                for f in self.combine(h, land[h]):
                    f()
            pending += self.bias_reads(bias_addr, bias_off + 128 * b)
            for h in H:
                pending += self.dequant(h, land[h], relu)
                if homes[(h, b)].kind == 'a':
                    pending += self.park(land[h], homes[(h, b)])
        return pending

    def stage0(self):
        A = self.A
        A.comment("==== stage 0: encodings only (split bf16), ReLU")
        for b in range(8):
            # bias16: the accumulators start at the biases, read straight into the blocks (either half of the file)
            for h in H:
                slot = HOME[(h, b)]
                for q in range(4):
                    A.ds_read128(slot.r(4 * q, 4), vr(V_BIAS), (stage_b_off(0) + 32 * b) * 4 + 32 * q)
            self.enc_block([(h, HOME[(h, b)], 0) for h in H], {'A': 0, 'B': POS_B}, 4)
        self.finish(HOME, 8, True, True)

    def hidden_stage_body(self):
        """stages 1..7 as a subroutine: s[S_ST] = stage, V_BIASST = bias base of the stage; stage 5 runs its four encoding blocks"""
        A = self.A
        A.raw(".Li8t_hidden:")
        A.barrier_state()
        ka = self.kappa_addr(st_sgpr=S_ST)
        self.sxin(ka)
        start_blk = self.blk

        def dump_prev():                              # X of the previous stage is complete after this stage's first block
            A.salu(f"s_sub_u32 s{S_TMP}, s{S_ST}, 1")
            A.salu(f"s_cmp_eq_u32 s{S_DBGST}, s{S_TMP}")
            A.raw(".Li8t_hidden_dump_go:")
            A.raw("s_cbranch_scc0 .Li8t_hidden_nodump")
            A.barrier_state()
            A.raw(f"s_call_b64 s[{S_RET2}:{S_RET2 + 1}], .Li8t_dump")
            A.raw(".Li8t_hidden_nodump:")
            A.barrier_state()
        pending = self.hidden_blocks(vr(V_BIASST), 0, True, first_args=self.first_block_args(self.quant_plan(HOME, 8, True)), after_first=dump_prev)
        for f in pending:
            f()
        A.salu(f"s_cmp_eq_u32 s{S_ST}, 5")
        A.raw("s_cbranch_scc0 .Li8t_hidden_tail")
        A.barrier_state()
        blk_after_i8 = self.blk
        for u in range(4):
            self.enc_block([(h, HOME[(h, 2 * u + k)], 4 * k) for k in range(2) for h in H], {'A': 0, 'B': POS_B}, 4)
        # the row maximum over everything, encodings included
        for hi, h in enumerate(H):
            A.valu(f"v_mov_b32 {vr(V_M[hi])}, 0", [vr(V_M[hi])], [])
            for b in range(8):
                slot = HOME[(h, b)]
                if slot.kind == 'a':
                    for f in self.unpark(slot, V_VB):
                        f()
                    src = lambda r: vr(V_VB + r)            # noqa: E731
                else:
                    src = lambda r, slot=slot: slot.r(r)    # noqa: E731
                for r in range(1, 16, 2):
                    self.max2(hi, src(r - 1), src(r), True)()
        A.raw(".Li8t_hidden_tail:")
        A.barrier_state()
        self.finish(HOME, 8, True, False)
        A.barrier_state()
        A.raw(f"s_setpc_b64 s[{S_RET}:{S_RET + 1}]")
        self.blk = start_blk                     # (the callers account for the ring blocks)
        return blk_after_i8 - start_blk

    def call_hidden(self, st):
        A = self.A
        A.comment(f"==== stage {st}")
        A.salu(f"s_mov_b32 s{S_ST}, {st}")
        A.valu(f"v_add_u32 {vr(V_BIASST)}, {256 * st * 4}, {vr(V_BIAS)}", [vr(V_BIASST)], [vr(V_BIAS)])
        A.barrier_state()
        A.raw(f"s_call_b64 s[{S_RET}:{S_RET + 1}], .Li8t_hidden")
        self.blk += 8 + (4 if st == 5 else 0)

    def stage8(self):
        A = self.A
        A.comment("==== stage 8: alpha (row 0 of its block, first in the stream) + feature (linear, 256)")
        self.sxin(self.kappa_addr(st=8))
        boff = stage_b_off(8) * 4
        # alpha block; stage 7's requantisation rides under it
        self.i8_block(8, [], **self.first_block_args(self.quant_plan(HOME, 8, True)))
        self.dump_check(7)
        for h in H:
            for f in self.combine(h, LAND[h])[:1]:
                f()
        A.op('ds_read', f"ds_read_b32 {vr(V_VB)}, {vr(V_BIAS)} offset:{boff + 256 * 4}", [vr(V_VB)], [vr(V_BIAS)])
        for hi, h in enumerate(H):
            for f in self.dequant(h, LAND[h], False, with_max=False, regs=[0]):
                f()
            A.valu(f"v_mul_f32 {vr(V_SIG[hi])}, {LAND[h].r(0)}, s{S_USIG}", [vr(V_SIG[hi])], [LAND[h].r(0)])
        pending = self.hidden_blocks(vr(V_BIAS), boff, False)
        for f in pending:
            f()
        self.finish(HOME, 8, False, False)

    def stage9(self):
        A = self.A
        A.comment("==== stage 9: views layer, K = feature(256) ++ d_pe(32), N = 128, ReLU")
        self.sxin(self.kappa_addr(st=9))
        pending = self.hidden_blocks(vr(V_BIAS), stage_b_off(9) * 4, True, nblk=4, homes=HOME9, first_args=self.first_block_args(self.quant_plan(HOME, 8, False)),
                                     after_first=lambda: self.dump_check(8))
        for f in pending:
            f()
        self.enc_block([(h, HOME9[(h, b)], 2 * b) for b in range(4) for h in H], {'A': DIR_A, 'B': DIR_B}, 2)
        self.finish(HOME9, 4, True, True)

    def stage10(self):
        A = self.A
        A.comment("==== stage 10: rgb (rows 0..2 of one block), K = 128; the sample's record")
        self.sxin(self.kappa_addr(st=10))
        self.i8_block(4, [], **self.first_block_args(self.quant_plan(HOME9, 4, True)))
        self.dump_check(9)
        boff = stage_b_off(10) * 4
        A.ds_read128(vr(V_VB, 4), vr(V_BIAS), boff)
        for hi, h in enumerate(H):
            slot = VSLOT[hi]
            for f in self.combine(h, slot)[:3]:
                f()
            for f in self.dequant(h, slot, False, with_max=False, regs=[0, 1, 2]):
                f()
            A.valu(f"v_mul_f32 {slot.r(0)}, {slot.r(0)}, s{S_UR}", [slot.r(0)], [slot.r(0)])
            A.valu(f"v_mul_f32 {slot.r(1)}, {slot.r(1)}, s{S_UG}", [slot.r(1)], [slot.r(1)])
            A.valu(f"v_mul_f32 {slot.r(2)}, {slot.r(2)}, s{S_UB}", [slot.r(2)], [slot.r(2)])
            A.valu(f"v_mul_f32 {slot.r(3)}, {vr(V_SIG[hi])}, s{S_SIGSC}", [slot.r(3)], [vr(V_SIG[hi])])
            out = V_OUTA if h == 'A' else V_OUTB
            A.valu(f"v_cmp_ne_u64 vcc, 0, {vr(out, 2)}", ['vcc'], [vr(out, 2)])
            A.raw("s_nop 1")
            A.n += 2
            A.raw(f"s_and_saveexec_b64 s[{S_SAVE}:{S_SAVE + 1}], vcc")
            A.op('vmem', f"global_store_dwordx4 {vr(out, 2)}, {slot.r(0, 4)}, off", [], [vr(out, 2), slot.r(0, 4)])
            A.raw("s_nop 1")
            A.n += 2
            A.raw(f"s_mov_b64 exec, s[{S_SAVE}:{S_SAVE + 1}]")

    def dump_check(self, st):
        A = self.A
        A.salu(f"s_cmp_eq_u32 s{S_DBGST}, {st}")
        A.raw(f"s_cbranch_scc0 .Li8t_nodump{st}")
        A.barrier_state()
        A.raw(f"s_call_b64 s[{S_RET2}:{S_RET2 + 1}], .Li8t_dump")
        A.raw(f".Li8t_nodump{st}:")
        A.barrier_state()

    def dump_body(self):
        """X (a0..a127) | sx A | sx B of this lane -> dbg + lane offset   (the tile's first workgroup only: the base is 0 elsewhere)"""
        A = self.A
        A.raw(".Li8t_dump:")
        A.barrier_state()
        A.salu(f"s_cmp_eq_u64 s[{S_DBG}:{S_DBG + 1}], 0")
        A.raw(".Li8t_dump_go:")
        A.raw("s_cbranch_scc1 .Li8t_dump_end")
        for k in range(128):
            A.valu(f"v_accvgpr_read_b32 {vr(V_T0)}, {ar(k)}", [vr(V_T0)], [ar(k)])
            A.raw("s_nop 1")
            A.n += 2
            A.op('vmem', f"global_store_dword {vr(V_DBGOFF)}, {vr(V_T0)}, s[{S_DBG}:{S_DBG + 1}] offset:{4 * k}", [], [vr(V_DBGOFF), vr(V_T0)])
            A.raw("s_nop 1")
            A.n += 2
        for hi in range(2):
            A.op('vmem', f"global_store_dword {vr(V_DBGOFF)}, {vr(V_SX[hi])}, s[{S_DBG}:{S_DBG + 1}] offset:{4 * (128 + hi)}", [], [vr(V_DBGOFF), vr(V_SX[hi])])
        A.raw("s_nop 1")
        A.raw(".Li8t_dump_end:")
        A.barrier_state()
        A.raw(f"s_setpc_b64 s[{S_RET2}:{S_RET2 + 1}]")

    # ---------------------------------------------------------------------------------------------------------------------------------------
    def build(self):
        A = self.A
        A.comment("constants")
        A.salu(f"s_mov_b32 s{S_KOFF}, m0")
        A.salu(f"s_mov_b32 s{S_SEL_LO}, 0x06040200")
        A.salu(f"s_mov_b32 s{S_SEL_HI}, 0x07050301")
        A.salu(f"s_mov_b32 s{S_C128}, 0x00800080")
        for j in range(1, 4):
            A.valu(f"v_add_u32 {vr(V_CP1 + j - 1)}, {j * 4096}, {vr(V_CP0)}", [vr(V_CP1 + j - 1)], [vr(V_CP0)])
        # the tile's first block was handed over by the previous tile's last (or by the kernel's start-up): its slot precedes s[S_SLOT]
        A.salu(f"s_sub_u32 s{S_TMP}, s{S_SLOT}, 1")
        A.salu(f"s_cmp_eq_u32 s{S_SLOT}, 0")
        A.salu(f"s_cselect_b32 s{S_TMP}, 2, s{S_TMP}")
        A.salu(f"s_lshl_b32 s{S_RDOFF}, s{S_TMP}, 14")
        self.stage0()
        for st in range(1, 8):
            self.call_hidden(st)
        assert self.blk == 68, self.blk
        self.stage8()
        self.stage9()
        self.stage10()
        assert self.blk == 83, self.blk
        A.raw("s_branch .Li8t_end")
        self.blk = 8
        self.hidden_stage_body()
        self.dump_body()
        A.raw(".Li8t_end:")
        A.salu(f"s_mov_b32 m0, s{S_KOFF}")
        A.nop(Asm.MFMA_D_STATES)
        return A


def wrapper(A):
    used_v, used_a, used_s = set(), set(), set()
    import re
    for ln in A.lines:
        if ln.startswith(';'):
            continue
        for m in re.finditer(r"\b([vas])\[(\d+):(\d+)\]|\b([vas])(\d+)\b", ln):
            if m.group(1):
                k, lo, hi = m.group(1), int(m.group(2)), int(m.group(3))
                rs = range(lo, hi + 1)
            else:
                k, rs = m.group(4), [int(m.group(5))]
            {'v': used_v, 'a': used_a, 's': used_s}[k].update(rs)
    pinned_v = {V_RDBASE, V_BIAS, V_PE, V_CP0, V_OUTA, V_OUTA + 1, V_OUTB, V_OUTB + 1, V_DBGOFF}
    pinned_s = {S_IMG, S_IMG + 1, S_RING0, S_OFF, S_SLOT, S_USIG, S_UR, S_UG, S_UB, S_SIGSC, S_DBGST, S_DBG, S_DBG + 1, S_KAPPA}
    assert min(used_v) >= 4, sorted(used_v)[:8]
    clob = [f'"v{i}"' for i in sorted(used_v - pinned_v)] + [f'"a{i}"' for i in sorted(used_a)] + [f'"s{i}"' for i in sorted(used_s - pinned_s)]
    clob += ['"vcc"', '"scc"', '"memory"']
    body = "\n".join('        "' + ln.replace('\\', '\\\\').replace('"', '\\"') + '\\n\\t"' for ln in A.lines)
    st = A.stats
    return f'''// GENERATED by tools/gen_i8t.py -- do not edit.  Stages 0 .. 10 of nerf_mlp_i8t_kernel as one hand-allocated instruction stream.
// {len(A.lines)} lines: {st['mfma']} MFMA, {st['valu']} VALU, {st['ds']} LDS reads, {st['vmem']} VMEM, {st['salu']} SALU, {st['nop_states']} padded wait states,
// {st['waits']} lgkmcnt waits (static counts of the emitted text; the hidden-stage subroutine runs seven times per tile).
#pragma once

__device__ __forceinline__ void stages_asm(const Args8t& A, const MlpArgs& a, RingT& R, const uint4* pw, unsigned bias_lds, const float* kappa, int g, int s,
                                           int tid, int64_t tile, int64_t row0, float u_sigma, float u_r, float u_g, float u_b) {{
    // (the generated stream addresses everything through these: see the register map at the top of tools/gen_i8t.py)
    register unsigned v_rdbase asm("v{V_RDBASE}") = (unsigned)(uintptr_t)R.rd;
    register unsigned v_bias asm("v{V_BIAS}") = bias_lds;
    register unsigned v_pe asm("v{V_PE}") = (unsigned)(uintptr_t)pw + (unsigned)(g * 1024 + s * 16);
    register unsigned v_cp0 asm("v{V_CP0}") = (unsigned)((uintptr_t)R.src - (uintptr_t)A.image8);       // lane * 16 + wave * 1024
    float4* rec = reinterpret_cast<float4*>(a.out);
    const int64_t ia = row0 + s, ib = row0 + 32 + s;
    // (branch-free on purpose: hipcc places the spill of a value that lives across the statement below at the top of the join block of a
    //  divergent region, BEFORE the exec mask is restored -- half the lanes then reload garbage in the next tile.  The record index is
    //  computed for a clamped row by every lane, the predicate is a select.)
    const int64_t ca = ia < a.n ? ia : a.n - 1, cb = ib < a.n ? ib : a.n - 1;
    const unsigned long long qa = (unsigned long long)(uintptr_t)(rec + sample_record(a, ca)), qb = (unsigned long long)(uintptr_t)(rec + sample_record(a, cb));
    const unsigned long long pa = (g == 0 && ia < a.n) ? qa : 0ull;
    const unsigned long long pb = (g == 0 && ib < a.n) ? qb : 0ull;
    register unsigned v_outa0 asm("v{V_OUTA}") = (unsigned)pa;
    register unsigned v_outa1 asm("v{V_OUTA + 1}") = (unsigned)(pa >> 32);
    register unsigned v_outb0 asm("v{V_OUTB}") = (unsigned)pb;
    register unsigned v_outb1 asm("v{V_OUTB + 1}") = (unsigned)(pb >> 32);
    register unsigned v_dbgoff asm("v{V_DBGOFF}") = (unsigned)tid * 520u;
    register unsigned s_img0 asm("s{S_IMG}") = (unsigned)(uintptr_t)A.image8;
    register unsigned s_img1 asm("s{S_IMG + 1}") = (unsigned)((unsigned long long)(uintptr_t)A.image8 >> 32);
    register unsigned s_ring0 asm("s{S_RING0}") = __builtin_amdgcn_readfirstlane(R.lds0);
    register unsigned s_off asm("s{S_OFF}") = __builtin_amdgcn_readfirstlane((unsigned)R.off);
    register unsigned s_slot asm("s{S_SLOT}") = __builtin_amdgcn_readfirstlane((unsigned)R.slot);
    register float s_usig asm("s{S_USIG}") = u_sigma;
    register float s_ur asm("s{S_UR}") = u_r;
    register float s_ug asm("s{S_UG}") = u_g;
    register float s_ub asm("s{S_UB}") = u_b;
    register float s_sigsc asm("s{S_SIGSC}") = a.sigma_scale;
    register int s_dbgst asm("s{S_DBGST}") = (A.dbg && tile == A.dbg_tile && blockIdx.x == 0) ? A.dbg_stage : -1;
    register unsigned s_dbg0 asm("s{S_DBG}") = (unsigned)(uintptr_t)A.dbg;
    register unsigned s_dbg1 asm("s{S_DBG + 1}") = (unsigned)((unsigned long long)(uintptr_t)A.dbg >> 32);
    register unsigned s_kappa asm("s{S_KAPPA}") = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)kappa);
#ifdef NM_I8T_SYNC
    __syncthreads();
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#endif
    asm volatile(
#ifdef NM_I8T_EMPTY
        "s_nop 0\\n\\t"
#else
{body}
#endif
        : "+s"(s_off), "+s"(s_slot)
        : "v"(v_rdbase), "v"(v_bias), "v"(v_pe), "v"(v_cp0), "v"(v_outa0), "v"(v_outa1), "v"(v_outb0), "v"(v_outb1), "v"(v_dbgoff), "s"(s_img0), "s"(s_img1), "s"(s_ring0), "s"(s_usig), "s"(s_ur), "s"(s_ug),
          "s"(s_ub), "s"(s_sigsc), "s"(s_dbgst), "s"(s_dbg0), "s"(s_dbg1), "s"(s_kappa)
#ifdef NM_I8T_FEWCLOB
        : {", ".join(c for c in clob if not (c.startswith('"a') or (c.startswith('"v') and c[2:-1].isdigit() and int(c[2:-1]) >= 128)))});
#else
        : {", ".join(clob)});
#endif
#ifdef NM_I8T_SYNC
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
#endif
    R.off = (int)s_off;
    R.slot = (int)s_slot;
}}
'''


def main():
    g = Gen()
    A = g.build()
    text = wrapper(A)
    if "--check" in sys.argv:
        cur = open(OUT).read() if os.path.exists(OUT) else ""
        print("up to date" if cur == text else "STALE")
        sys.exit(0 if cur == text else 1)
    with open(OUT, "w") as f:
        f.write(text)
    print(OUT, len(A.lines), "lines", A.stats)


if __name__ == "__main__":
    main()
