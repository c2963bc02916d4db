#!/usr/bin/env python3
"""Generator of ml-neuman_amd/csrc/mlp_f16t_body.h: the density-only NM_PREC_FP16X3 network (stages 0..7 and the alpha row: what the sampling
pass of a two-pass render evaluates, csrc/mlp.hip nerf_mlp_kernel<fp16x3> with sigma_only) as ONE hand-allocated gfx950 instruction stream for
nerf_sigma_f16t_kernel (csrc/mlp_f16t.hip): activation-stationary, 4 waves per CU with all 512 registers each, 32 samples per wave.

    python tools/gen_f16t.py         (uses the emitter tools/asm_emit.py: LDS return queue and wait-state bookkeeping)

The arithmetic is nerf_mlp_kernel's, MFMA for MFMA: per output block and k-step the products wh.xl, wl.xh, wh.xh on v_mfma_f32_32x32x16_f16
in that order into an accumulator that starts at the bias, then x 2^-k, clamp, split into two fp16 parts (v_cvt_pk_f16_f32, back-conversion,
exact difference, v_cvt_pk_f16_f32) -- so sigma is bit-identical.  What changes:

  * a wave owns 32 samples and ALL features: the accumulator layout of the 32x32x16 MFMA gives a lane, for its sample, exactly the k-slots the
    next stage's B operand wants from it (mlp_layout.h slot_feature: the weight columns were permuted for it in round 1), so the split parts of
    output block b ARE the B fragments of k-steps 2b, 2b+1 of the next stage, in place.  Activations never touch LDS; no barrier between stages;
  * the inputs of a stage (16 k-steps x (hi, lo) x 4 registers = 128) ping-pong between the two halves of the register file: even stages read
    a0..a127 and write v62..v189, odd stages the other way round -- the next stage's inputs are written while the current one still reads its own;
  * output blocks are computed in PAIRS (two accumulator chains interleaved MFMA by MFMA); the split of pair p rides in the gaps between the
    MFMAs of pair p + 1 (of the next stage's first pair for the last one: its inputs are k-steps 12..15, first needed by that pair's second half);
  * the weights come through a 3 x 32 KB LDS ring filled by LDS-DMA (one copy per CU and 128 samples, handed over inside the predecessor unit).

Register file of a lane (the compiler keeps v0..v3):
    a0..a127     inputs of even stages (2, 4, 6, alpha): k-step t = a[8t .. 8t+3] (hi), a[8t+4 .. 8t+7] (lo)
    a128..a159   two weight buffers: block A hi / lo, block B hi / lo of one k-step each
    v4..v9       inputs (ring read base, bias base, encoding base, copy offset, output address);  v10..v31 scalars, temporaries
    v62..v189    inputs of odd stages (1, 3, 5, 7), same layout
    v190..v253   four accumulator blocks (two pairs)
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from asm_emit import Asm, ar, f32hex, vr  # noqa: E402

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
OUT = os.environ.get("F16T_OUT") or os.path.join(ROOT, "ml-neuman_amd", "csrc", "mlp_f16t_body.h")
STEP = 2048
K_STAGES = 11
PER_GAP = int(os.environ.get("F16T_PER_GAP", "4"))
PROBE = os.environ.get("F16T_PROBE", "")                  # timing probes (garbage results): novalu | nowread | nomfma | nocopy | nobarrier


def stage_b_off(s):                                        # fp16 image: stage_shape(s).nblk * 32 floats per stage
    nblk = {0: 8, 5: 8, 8: 9, 9: 4, 10: 1}
    return sum(nblk.get(i, 8) * 32 for i in range(s))


K_BIAS_FLOATS = stage_b_off(11)
SLOT_SHIFT = 15                                            # 32 KB ring slots
# Which 1 KB parts of a pair k-step (block A hi, A lo, block B hi, B lo) do NOT go through the LDS ring but straight from the stream (L2 / the CU's
# vector cache) into registers, D k-steps ahead: 0: none; 1: B lo; 2: B hi and B lo.  An experiment (profiles/r04_f16t_kernel.md): the launch
# takes the same time with 0, 1 and 2 -- it is the energy of moving a fragment, not the pipe it moves through, that costs -- so 0 is what is built.
NDIR = int(os.environ.get("F16T_NDIR", "0"))
D = 16                                                     # direct parts in flight: k-steps ahead (divides every stage's first pair k-step)
PARTS = [(0, 0), (0, 1), (1, 0), (1, 1)]
DIRECT = PARTS[4 - NDIR:]
LDS_PARTS = PARTS[:4 - NDIR]
NL = len(LDS_PARTS)
PAIR_STEPS = 480                                           # pair k-steps of a tile (the alpha row's 16 are single-block steps, all LDS)
K_LDS_BYTES = (PAIR_STEPS * NL + 16 * 2) * 1024
K_DIR_BYTES = PAIR_STEPS * NDIR * 1024
assert K_LDS_BYTES + K_DIR_BYTES == 976 * STEP

V_RDBASE, V_BIAS, V_PE, V_CP0, V_OUT = 4, 5, 6, 7, 8
V_RD, V_T2, V_BIASST = 10, 11, 12
V_SC = (13, 14)                                            # 2^-k of the stage being split, by stage parity
V_TMP = 15                                                 # one scratch register
V_XPE = 16                                                 # 8: encoding operands hi (4), lo (4)
V_EP = 24                                                  # 8 temporaries of the split
V_XO = 62
V_ACC = 190
V_DBGOFF = 254
A_XE = 0
A_W = 128                                                  # two buffers of the LDS-read parts of a k-step
A_DIR = A_W + 8 * NL                                       # D entries of the direct parts (the last ones spill into v32..v47)
V_L16 = 32 + 16                                            # lane * 16
S_IMG, S_RING0, S_OFF, S_SLOT, S_OS, S_SIGSC, S_DBGST, S_TAB, S_DBG = 36, 38, 39, 40, 41, 42, 43, 44, 46
S_REFILL, S_P, S_TMP, S_RDOFF, S_RET, S_RET2, S_ST, S_SAVE, S_M0, S_CLAMP = 52, 54, 56, 57, 58, 60, 62, 64, 66, 67
S_BB, S_BP = 48, 68                                        # the direct stream: its base (input), the running pointer


def xreg(par, t, part):
    """B fragment of k-step t: part 0 = hi, 1 = lo; even-stage inputs live in the accumulator half, odd-stage inputs in the VALU half"""
    base = 8 * t + 4 * part
    return ar(A_XE + base, 4) if par == 0 else vr(V_XO + base, 4)


def xreg1(par, t, part, i):
    base = 8 * t + 4 * part + i
    return ar(A_XE + base) if par == 0 else vr(V_XO + base)


def lbuf(s, idx):
    return ar(A_W + 4 * NL * (s & 1) + 4 * idx, 4)


def dbuf(e, k):
    g = e * NDIR + k
    nag = (256 - A_DIR) // 4
    if g < nag:
        return ar(A_DIR + 4 * g, 4)
    assert 32 + 4 * (g - nag) + 3 < V_L16
    return vr(32 + 4 * (g - nag), 4)


class GenF:
    def __init__(self):
        self.A = Asm()
        self.unit = 0                                     # ring unit index within the tile (static)
        self.bn = 0                                       # pair k-step index within the tile (static; inside a subroutine: its first caller's)
        self.in_sub = False
        self.unit_loads = 0
        self.units = []                                   # (k-steps, LDS parts per k-step)
        for st in range(8):
            for _ in range(4):
                if st == 0:
                    self.units.append((4, NL))
                else:
                    if st == 5:
                        self.units.append((4, NL))
                    self.units += [(8, NL), (8, NL)]
        self.units += [(8, 2), (8, 2)]
        assert len(self.units) == 66 and sum(n * k for n, k in self.units) * 1024 == K_LDS_BYTES

    def ubytes(self, i):
        n, k = self.units[i % len(self.units)]
        return n * k * 1024

    # ---- ring (hand-over inside the predecessor, copies behind it) ------------------------------------------------
    def ring_start(self):
        A = self.A
        A.comment(f"---- ring unit {self.unit}")
        A.valu(f"v_add_u32 {vr(V_RD)}, s{S_RDOFF}, {vr(V_RDBASE)}", [vr(V_RD)], [vr(V_RDBASE)])

    def ring_handover(self):
        A = self.A
        i = self.unit
        # the pieces of unit i + 1 were issued inside unit i - 1; everything this unit has issued since (direct loads only) is younger
        A.raw(f"s_waitcnt vmcnt({self.unit_loads})")
        if PROBE != "nobarrier":
            A.raw("s_barrier")
        A.salu(f"s_lshl_b32 s{S_RDOFF}, s{S_SLOT}, {SLOT_SHIFT}")
        A.salu(f"s_add_u32 s{S_TMP}, s{S_SLOT}, 1")
        A.salu(f"s_cmp_eq_u32 s{S_SLOT}, 2")
        A.salu(f"s_cselect_b32 s{S_SLOT}, 0, s{S_TMP}")
        A.salu(f"s_lshl_b32 s{S_TMP}, s{S_SLOT}, {SLOT_SHIFT}")
        A.salu(f"s_add_u32 s{S_REFILL}, s{S_RING0}, s{S_TMP}")
        nbytes = self.ubytes(i + 2)
        assert nbytes % 4096 == 0
        pieces = nbytes // 4096
        out = []
        for j in range(pieces):
            out.append(lambda j=j, last=(j == pieces - 1), nbytes=nbytes: self.copy_piece(j, nbytes, last))
        return out

    def copy_piece(self, j, nbytes, last):
        A = self.A
        A.salu(f"s_add_u32 s{S_TMP}, s{S_OFF}, {j * 4096}")
        A.salu(f"s_add_u32 s{S_P}, s{S_IMG}, s{S_TMP}")
        A.salu(f"s_addc_u32 s{S_P + 1}, s{S_IMG + 1}, 0")
        A.salu(f"s_add_u32 m0, s{S_REFILL}, {j * 4096}")
        A.raw("s_nop 0")
        A.n += 1
        if PROBE != "nocopy":
            A.op('vmem', f"global_load_lds_dwordx4 {vr(V_CP0)}, s[{S_P}:{S_P + 1}]", [], [vr(V_CP0)])
        if last:
            A.salu(f"s_add_u32 s{S_OFF}, s{S_OFF}, {nbytes}")
            A.salu(f"s_cmp_eq_u32 s{S_OFF}, {K_LDS_BYTES}")
            A.salu(f"s_cselect_b32 s{S_OFF}, 0, s{S_OFF}")

    def direct_loads(self, e):
        """the direct parts of a pair k-step -> entry e, and the pointer moves on"""
        A = self.A
        for k in range(NDIR):
            d = dbuf(e, k)
            A.op('vmem', f"global_load_dwordx4 {d}, {vr(V_L16)}, s[{S_BP}:{S_BP + 1}]" + (f" offset:{1024 * k}" if k else ""), [], [vr(V_L16)])
            self.unit_loads += 1
        if NDIR:
            A.salu(f"s_add_u32 s{S_BP}, s{S_BP}, {1024 * NDIR}")
            A.salu(f"s_addc_u32 s{S_BP + 1}, s{S_BP + 1}, 0")

    # ---- pieces ----------------------------------------------------------------------------------------------------------------------------
    @staticmethod
    def acc(slot, n=16, i=0):
        return vr(V_ACC + 16 * slot + i, n)

    def bias_init(self, slot, bias_addr, off_bytes):
        """the accumulators of a block start at its biases (bias_prefetch + init_bias: float4 at bias_blk + 8 q + 4 g), read straight into them"""
        for q in range(4):
            self.A.ds_read128(self.acc(slot, 4, 4 * q), bias_addr, off_bytes + 32 * q)

    def split(self, slot, b, par_out, sc):
        """convert_act<RELU, fp16x3> of one block: its 16 outputs -> the (hi, lo) fragments of k-steps 2b, 2b+1 of the next stage's inputs
        (split8<true, true>: x scale, clamp to [0, 65504], RNE to fp16 pairs, back, exact difference, RNE).  -> closures, by kind"""
        A = self.A
        out = []
        if PROBE == "novalu":
            return out
        f = lambda i: self.acc(slot, 1, i)                                      # noqa: E731
        for i in range(16):
            out.append(lambda i=i: A.valu(f"v_mul_f32 {f(i)}, {f(i)}, {vr(sc)}", [f(i)], [f(i), vr(sc)]))
        for i in range(16):
            out.append(lambda i=i: A.valu(f"v_med3_f32 {f(i)}, {f(i)}, 0, s{S_CLAMP}", [f(i)], [f(i)]))
        for qp in range(2):
            t = 2 * b + qp
            for p in range(4):
                x, y = f(8 * qp + 2 * p), f(8 * qp + 2 * p + 1)
                hdst = xreg1(par_out, t, 0, p)
                ldst = xreg1(par_out, t, 1, p)
                th, tl = vr(V_EP + 2 * (p & 1)), vr(V_EP + 2 * (p & 1) + 1)     # back-converted halves
                hv = vr(V_EP + 4 + (p & 1))                                     # the packed hi pair (also the source of the back-conversion)

                if par_out == 1:
                    hv = hdst                                                   # (a VALU-half destination: converted in place)

                def one(x=x, y=y, hdst=hdst, ldst=ldst, th=th, tl=tl, hv=hv):
                    A.valu(f"v_cvt_pk_f16_f32 {hv}, {x}, {y}", [hv], [x, y])
                    A.valu(f"v_cvt_f32_f16_e32 {th}, {hv}", [th], [hv])
                    A.valu(f"v_cvt_f32_f16_sdwa {tl}, {hv} dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1", [tl], [hv])
                    A.valu(f"v_sub_f32 {x}, {x}, {th}", [x], [x, th])
                    A.valu(f"v_sub_f32 {y}, {y}, {tl}", [y], [y, tl])
                    if par_out == 0:
                        A.valu(f"v_accvgpr_write_b32 {hdst}, {hv}", [hdst], [hv])
                        A.valu(f"v_cvt_pk_f16_f32 {hv}, {x}, {y}", [hv], [x, y])
                        A.valu(f"v_accvgpr_write_b32 {ldst}, {hv}", [ldst], [hv])
                    else:
                        A.valu(f"v_cvt_pk_f16_f32 {ldst}, {x}, {y}", [ldst], [x, y])
                out += self.collect(one)
        return out

    def collect(self, fn):
        rec = []
        A = self.A
        real_op = A.op
        A.op = lambda *a, **k: rec.append(lambda a=a, k=k: real_op(*a, **k))
        try:
            fn()
        finally:
            A.op = real_op
        return rec

    def pair_unit(self, nsteps, kind, par_in, t0, slots, fillers, pe_t0=0):
        """one ring unit: `nsteps` k-steps of a pair of output blocks (slots = two accumulator blocks; one block: the alpha row).  kind 'h':
        the B fragments are the resident inputs of parity par_in, k-steps t0 ..; kind 'pe': the wave's encoding rows (LDS), chunk pairs pe_t0 ..
        fillers ride behind the MFMAs; leftovers are flushed at the end of the unit."""
        A = self.A
        self.ring_start()
        self.unit_loads = 0
        fill_iter = iter(list(fillers))
        nb = len(slots)
        rd = vr(V_RD)
        lparts = LDS_PARTS if nb == 2 else PARTS[:2]
        assert (nsteps, len(lparts)) == self.units[self.unit % len(self.units)]

        def wsrc(s, blk, part):
            if (blk, part) in lparts:
                return lbuf(s, lparts.index((blk, part)))
            return dbuf(self.bn % D, DIRECT.index((blk, part)))

        def wreads(s, addr):
            out = []
            for idx in range(len(lparts)):
                if PROBE != "nowread":
                    out.append(lambda idx=idx: A.ds_read128(lbuf(s, idx), addr, (s * len(lparts) + idx) * 1024))
            return out
        for f in wreads(0, rd):
            f()
        copies = []
        for s in range(nsteps):
            gap_extra = []
            if s == nsteps - 2:
                copies = self.ring_handover()
            if s + 1 < nsteps:
                gap_extra += wreads(s + 1, rd)
            if s >= nsteps - 2:
                k = (len(copies) + 1) // 2 if s == nsteps - 2 else len(copies)
                gap_extra += copies[:k]
                copies = copies[k:]
            if kind == 'pe':
                A.ds_read128(vr(V_XPE, 4), vr(V_PE), (pe_t0 + s) * 2048)
                A.ds_read128(vr(V_XPE + 4, 4), vr(V_PE), (pe_t0 + s) * 2048 + 512)
                xh, xl = vr(V_XPE, 4), vr(V_XPE + 4, 4)
            else:
                xh, xl = xreg(par_in, t0 + s, 0), xreg(par_in, t0 + s, 1)
            seq = []
            for part in range(3):
                for blk in range(nb):
                    wp, b = [(0, xl), (1, xh), (0, xh)][part]
                    seq.append((self.acc(slots[blk]), wsrc(s, blk, wp), b, nb == 2 and (blk, wp) in DIRECT))
            waited = False
            for acc, a, b, direct in seq:
                if direct and not waited:                                             # this k-step's direct parts: everything younger may stay in flight
                    younger = D - 1 if self.in_sub else min(D - 1, PAIR_STEPS - 1 - self.bn)
                    A.raw(f"s_waitcnt vmcnt({NDIR * younger})")
                    waited = True
                if PROBE != "nomfma":
                    A.op('mfma', f"v_mfma_f32_32x32x16_f16 {acc}, {a}, {b}, {acc}", [acc], [a, b, acc], chain=acc)
                budget = PER_GAP
                while gap_extra and budget > 0:
                    gap_extra.pop(0)()
                    budget -= 1
                while budget > 0:
                    f = next(fill_iter, None)
                    if f is None:
                        break
                    f()
                    budget -= 1
            if nb == 2:
                if NDIR and (self.in_sub or self.bn + D < PAIR_STEPS):                # the entry just consumed takes the parts of k-step bn + D
                    self.direct_loads(self.bn % D)
                self.bn += 1
            for f in gap_extra:
                f()
            if PROBE == "nomfma":
                A.wait_lds(0)
        for f in fill_iter:
            f()
        self.unit += 1

    # ---- stages ----------------------------------------------------------------------------------------------------------------------------
    def load_scale(self, st_const=None):
        """V_SC[st & 1] <- f16tab[st] = 2^-k of stage st (the per-stage factors follow the bias table in LDS)"""
        A = self.A
        if st_const is not None:
            A.valu(f"v_mov_b32 {vr(V_TMP)}, s{S_TAB}", [vr(V_TMP)], [])
            A.op('ds_read', f"ds_read_b32 {vr(V_SC[st_const & 1])}, {vr(V_TMP)} offset:{4 * st_const}", [vr(V_SC[st_const & 1])], [vr(V_TMP)])

    def stage_pairs(self, st, pending, bias_addr, bias_off, dump_after_first=None):
        """the four block pairs of stage st (0..7).  pending: the split of the previous stage's last pair, riding under this stage's first unit."""
        par_in, par_out = st & 1, (st + 1) & 1
        sc = V_SC[st & 1]
        for p in range(4):
            slots = (2 * (p & 1), 2 * (p & 1) + 1)
            for k, slot in enumerate(slots):
                self.bias_init(slot, bias_addr, bias_off + 128 * (2 * p + k))
            first = True
            if st == 0:
                self.pair_unit(4, 'pe', None, 0, slots, pending)
                pending = []
                if p == 0 and dump_after_first:
                    dump_after_first()
            else:
                if st == 5:
                    self.pair_unit(4, 'pe', None, 0, slots, pending)
                    pending = []
                    first = False
                    if p == 0 and dump_after_first:
                        dump_after_first()
                self.pair_unit(8, 'h', par_in, 0, slots, pending)
                if first and p == 0 and dump_after_first:
                    dump_after_first()
                self.pair_unit(8, 'h', par_in, 8, slots, [])
            pending = []
            for k, slot in enumerate(slots):
                pending += self.split(slot, 2 * p + k, par_out, sc)
        return pending

    def last_pair_split(self, st):
        """the split of stage st's fourth pair as the NEXT stage's code sees it (slots 2, 3 -> k-steps 12..15 of the inputs of parity (st + 1) & 1)"""
        out = []
        for k, slot in enumerate((2, 3)):
            out += self.split(slot, 6 + k, (st + 1) & 1, V_SC[st & 1])
        return out

    def dump_check(self, prev_st_sgpr=None, prev_st=None, par=0):
        """X of the previous stage is complete after this stage's first unit: dump it when asked for"""
        A = self.A
        tag = f"{self.unit}_{par}"
        if prev_st is not None:
            A.salu(f"s_cmp_eq_u32 s{S_DBGST}, {prev_st}")
        else:
            A.salu(f"s_sub_u32 s{S_TMP}, s{prev_st_sgpr}, 1")
            A.salu(f"s_cmp_eq_u32 s{S_DBGST}, s{S_TMP}")
        A.raw(f"s_cbranch_scc0 .Lf16t_nodump{tag}")
        A.barrier_state()
        A.raw(f"s_call_b64 s[{S_RET2}:{S_RET2 + 1}], .Lf16t_dump{par}")
        A.raw(f".Lf16t_nodump{tag}:")
        A.barrier_state()

    def dump_body(self, par):
        A = self.A
        A.raw(f".Lf16t_dump{par}:")
        A.barrier_state()
        for k in range(128):
            if par == 0:
                A.valu(f"v_accvgpr_read_b32 {vr(V_TMP)}, {ar(A_XE + k)}", [vr(V_TMP)], [ar(A_XE + k)])
                src = vr(V_TMP)
            else:
                src = vr(V_XO + k)
            A.raw("s_nop 1")
            A.n += 2
            A.op('vmem', f"global_store_dword {vr(V_DBGOFF)}, {src}, s[{S_DBG}:{S_DBG + 1}] offset:{4 * k}", [], [vr(V_DBGOFF), src])
            A.raw("s_nop 1")
            A.n += 2
        A.barrier_state()
        A.raw(f"s_setpc_b64 s[{S_RET2}:{S_RET2 + 1}]")

    def generic_body(self, par):
        """stages 1, 3 (par 1) / 2, 6 (par 0) as a subroutine: s[S_ST] = stage, V_BIASST its bias base, V_SC[par] its scale"""
        A = self.A
        A.raw(f".Lf16t_stage{par}:")
        A.barrier_state()
        start, bstart = self.unit, self.bn
        st = 1 if par else 2
        self.bn = 16 + 64 * (st - 1)
        self.in_sub = True
        self.stage_pairs(st, self.last_pair_split(st - 1), vr(V_BIASST), 0, dump_after_first=lambda: self.dump_check(prev_st_sgpr=S_ST, par=par))
        A.barrier_state()
        A.raw(f"s_setpc_b64 s[{S_RET}:{S_RET + 1}]")
        self.unit, self.bn = start, bstart
        self.in_sub = False

    def call_generic(self, st):
        A = self.A
        A.comment(f"==== stage {st}")
        A.salu(f"s_mov_b32 s{S_ST}, {st}")
        A.valu(f"v_add_u32 {vr(V_BIASST)}, {256 * st * 4}, {vr(V_BIAS)}", [vr(V_BIASST)], [vr(V_BIAS)])
        self.load_scale(st)
        A.barrier_state()
        A.raw(f"s_call_b64 s[{S_RET}:{S_RET + 1}], .Lf16t_stage{st & 1}")
        self.unit += 8
        self.bn += 64
        assert self.bn % D == 0

    def inline_stage(self, st):
        """stages 4 and 7: the units copied behind their last two hand-overs (stage 5's encoding unit, the alpha row's) are not the 16-step
        units the subroutines' copies assume"""
        A = self.A
        A.comment(f"==== stage {st} (inline)")
        self.load_scale(st)
        self.stage_pairs(st, self.last_pair_split(st - 1), vr(V_BIAS), stage_b_off(st) * 4, dump_after_first=lambda: self.dump_check(prev_st=st - 1, par=st & 1))

    def build(self):
        A = self.A
        A.comment("constants")
        A.salu(f"s_mov_b32 s{S_M0}, m0")
        A.salu(f"s_mov_b32 s{S_CLAMP}, {f32hex(65504.0)}")
        A.salu(f"s_sub_u32 s{S_TMP}, s{S_SLOT}, 1")
        A.salu(f"s_cmp_eq_u32 s{S_SLOT}, 0")
        A.salu(f"s_cselect_b32 s{S_TMP}, 2, s{S_TMP}")
        A.salu(f"s_lshl_b32 s{S_RDOFF}, s{S_TMP}, {SLOT_SHIFT}")
        if NDIR:
            A.comment("the direct parts of the first D pair k-steps")
            A.salu(f"s_mov_b64 s[{S_BP}:{S_BP + 1}], s[{S_BB}:{S_BB + 1}]")
            A.valu(f"v_and_b32 {vr(V_L16)}, 1023, {vr(V_CP0)}", [vr(V_L16)], [vr(V_CP0)])
            for e in range(D):
                self.direct_loads(e)
        A.comment("==== stage 0: encodings -> 256, ReLU")
        self.load_scale(0)
        self.stage_pairs(0, [], vr(V_BIAS), stage_b_off(0) * 4)
        for st in (1, 2, 3):
            self.call_generic(st)
        self.inline_stage(4)
        A.comment("==== stage 5: skip layer, its four encoding steps first")
        self.load_scale(5)
        self.stage_pairs(5, self.last_pair_split(4), vr(V_BIAS), stage_b_off(5) * 4, dump_after_first=lambda: self.dump_check(prev_st=4, par=1))
        self.call_generic(6)
        self.inline_stage(7)
        A.comment("==== the alpha row of stage 8 -> sigma")
        assert self.unit == 64 and self.bn == PAIR_STEPS, (self.unit, self.bn)
        self.bias_init(0, vr(V_BIAS), (stage_b_off(8) + 256) * 4)
        self.pair_unit(8, 'h', 0, 0, (0,), self.last_pair_split(7))
        self.dump_check(prev_st=7, par=0)
        self.pair_unit(8, 'h', 0, 8, (0,), [])
        assert self.unit == 66
        o = self.acc(1, 4)                                                           # the record: (0, 0, 0, sigma)
        A.valu(f"v_mul_f32 {self.acc(1, 1, 3)}, {self.acc(0, 1, 0)}, s{S_OS}", [self.acc(1, 1, 3)], [self.acc(0, 1, 0)])
        A.valu(f"v_mul_f32 {self.acc(1, 1, 3)}, {self.acc(1, 1, 3)}, s{S_SIGSC}", [self.acc(1, 1, 3)], [self.acc(1, 1, 3)])
        for i in range(3):
            A.valu(f"v_mov_b32 {self.acc(1, 1, i)}, 0", [self.acc(1, 1, i)], [])
        A.valu(f"v_cmp_ne_u64 vcc, 0, {vr(V_OUT, 2)}", ['vcc'], [vr(V_OUT, 2)])
        A.raw("s_nop 1")
        A.n += 2
        A.raw(f"s_and_saveexec_b64 s[{S_SAVE}:{S_SAVE + 1}], vcc")
        A.op('vmem', f"global_store_dwordx4 {vr(V_OUT, 2)}, {o}, off", [], [vr(V_OUT, 2), o])
        A.raw("s_nop 1")
        A.n += 2
        A.raw(f"s_mov_b64 exec, s[{S_SAVE}:{S_SAVE + 1}]")
        A.raw("s_branch .Lf16t_end")
        self.unit = 4
        self.generic_body(1)
        self.unit = 12
        self.generic_body(0)
        self.dump_body(0)
        self.dump_body(1)
        A.raw(".Lf16t_end:")
        A.salu(f"s_mov_b32 m0, s{S_M0}")
        A.nop(Asm.MFMA_D_STATES)
        return A


def wrapper(A):
    import re
    used_v, used_a, used_s = set(), set(), set()
    for ln in A.lines:
        if ln.startswith(';'):
            continue
        for m in re.finditer(r"\b([vas])\[(\d+):(\d+)\]|\b([vas])(\d+)\b", ln):
            if m.group(1):
                k, rs = m.group(1), range(int(m.group(2)), int(m.group(3)) + 1)
            else:
                k, rs = m.group(4), [int(m.group(5))]
            {'v': used_v, 'a': used_a, 's': used_s}[k].update(rs)
    pinned_v = {V_RDBASE, V_BIAS, V_PE, V_CP0, V_OUT, V_OUT + 1, V_DBGOFF}
    pinned_s = {S_IMG, S_IMG + 1, S_RING0, S_OFF, S_SLOT, S_OS, S_SIGSC, S_DBGST, S_TAB, S_DBG, S_DBG + 1, S_BB, S_BB + 1}
    assert min(used_v) >= 4
    clob = [f'"v{i}"' for i in sorted(used_v - pinned_v)] + [f'"a{i}"' for i in sorted(used_a)] + [f'"s{i}"' for i in sorted(used_s - pinned_s)]
    clob += ['"vcc"', '"scc"', '"memory"']
    body = "\n".join('        "' + ln.replace('\\', '\\\\').replace('"', '\\"') + '\\n\\t"' for ln in A.lines)
    st = A.stats
    return f'''// GENERATED by tools/gen_f16t.py -- do not edit.  The density-only fp16x3 network of nerf_sigma_f16t_kernel as one hand-allocated stream.
// {len(A.lines)} lines: {st['mfma']} MFMA, {st['valu']} VALU, {st['ds']} LDS reads, {st['vmem']} VMEM, {st['salu']} SALU, {st['nop_states']} padded wait states
// (static counts of the text; the two generic-stage subroutines run three times each per tile).
#pragma once
constexpr int kF16tNdir = {NDIR};                  // 1 KB parts of a pair k-step that bypass the LDS ring (mlp_host.hip sigma_stream_table must cut the stream for it)
constexpr int kF16tLdsBytes = {K_LDS_BYTES};       // the ring's part of the stream; the direct parts follow
constexpr int kF16tUnit0Pieces = {4 * NL // 4};    // 1 KB pieces per wave of a stage-0 ring unit (4 k-steps)

__device__ __forceinline__ void sigma_stages_asm(const ArgsF& A, const MlpArgs& a, RingF& R, const uint4* pw, unsigned bias_lds, unsigned tab_lds, int g, int s,
                                                 int tid, int64_t tile, int64_t row0, float os) {{
    register unsigned v_rdbase asm("v{V_RDBASE}") = (unsigned)(uintptr_t)R.rd;
    register unsigned v_bias asm("v{V_BIAS}") = bias_lds;
    register unsigned v_pe asm("v{V_PE}") = (unsigned)(uintptr_t)pw + (unsigned)(g * 1024 + s * 16);
    register unsigned v_cp0 asm("v{V_CP0}") = (unsigned)((uintptr_t)R.src - (uintptr_t)A.stream);       // lane * 16 + wave * 1024
    float4* rec = reinterpret_cast<float4*>(a.out);
    // (branch-free on purpose: see profiles/r04_i8t_kernel.md -- hipcc places a spill in front of the exec restore of a divergent region before the statement)
    const int64_t i0 = row0 + s, ci = i0 < a.n ? i0 : a.n - 1;
    const unsigned long long q = (unsigned long long)(uintptr_t)(rec + sample_record(a, ci));
    const unsigned long long po = (g == 0 && i0 < a.n) ? q : 0ull;
    register unsigned v_out0 asm("v{V_OUT}") = (unsigned)po;
    register unsigned v_out1 asm("v{V_OUT + 1}") = (unsigned)(po >> 32);
    register unsigned v_dbgoff asm("v{V_DBGOFF}") = (unsigned)tid * 512u;
    register unsigned s_img0 asm("s{S_IMG}") = (unsigned)(uintptr_t)A.stream;
    register unsigned s_img1 asm("s{S_IMG + 1}") = (unsigned)((unsigned long long)(uintptr_t)A.stream >> 32);
    register unsigned s_ring0 asm("s{S_RING0}") = __builtin_amdgcn_readfirstlane(R.lds0);
    register unsigned s_off asm("s{S_OFF}") = __builtin_amdgcn_readfirstlane((unsigned)R.off);
    register unsigned s_slot asm("s{S_SLOT}") = __builtin_amdgcn_readfirstlane((unsigned)R.slot);
    register float s_os asm("s{S_OS}") = os;
    register float s_sigsc asm("s{S_SIGSC}") = a.sigma_scale;
    register int s_dbgst asm("s{S_DBGST}") = (A.dbg && tile == A.dbg_tile && blockIdx.x == 0) ? A.dbg_stage : -1;
    register unsigned s_tab asm("s{S_TAB}") = __builtin_amdgcn_readfirstlane(tab_lds);
    register unsigned s_dbg0 asm("s{S_DBG}") = (unsigned)(uintptr_t)A.dbg;
    register unsigned s_dbg1 asm("s{S_DBG + 1}") = (unsigned)((unsigned long long)(uintptr_t)A.dbg >> 32);
    const unsigned long long bb = (unsigned long long)(uintptr_t)A.stream + (unsigned long long)kF16tLdsBytes;
    register unsigned s_bb0 asm("s{S_BB}") = (unsigned)bb;
    register unsigned s_bb1 asm("s{S_BB + 1}") = (unsigned)(bb >> 32);
    asm volatile(
{body}
        : "+s"(s_off), "+s"(s_slot)
        : "v"(v_rdbase), "v"(v_bias), "v"(v_pe), "v"(v_cp0), "v"(v_out0), "v"(v_out1), "v"(v_dbgoff), "s"(s_img0), "s"(s_img1), "s"(s_ring0), "s"(s_os),
          "s"(s_sigsc), "s"(s_dbgst), "s"(s_tab), "s"(s_dbg0), "s"(s_dbg1), "s"(s_bb0), "s"(s_bb1)
        : {", ".join(clob)});
    R.off = (int)s_off;
    R.slot = (int)s_slot;
}}
'''


def main():
    g = GenF()
    A = g.build()
    text = wrapper(A)
    with open(OUT, "w") as f:
        f.write(text)
    print(OUT, len(A.lines), "lines", A.stats)


if __name__ == "__main__":
    main()
