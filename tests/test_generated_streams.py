"""The hand-scheduled kernel's instruction stream (csrc/mlp_f16t_body.h) is a generated file: what is committed is
what the committed generator emits, and the emitter's bookkeeping holds on them (every LDS read is waited for before its first use; MFMA
results are not touched inside the hazard window)."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
CSRC = os.path.join(ROOT, "ml-neuman_amd", "csrc")


@pytest.mark.parametrize("gen,env,header", [("gen_f16t.py", "F16T_OUT", "mlp_f16t_body.h")])
def test_committed_stream_is_the_generators_output(tmp_path, gen, env, header):
    out = tmp_path / header
    e = {k: v for k, v in os.environ.items() if not k.startswith(("I8T_", "F16T_", "NM_I8T", "NM_F16T"))}
    e[env] = str(out)
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", gen)], check=True, env=e, timeout=600, stdout=subprocess.DEVNULL)
    assert out.read_text() == open(os.path.join(CSRC, header)).read()


@pytest.mark.parametrize("header", ["mlp_f16t_body.h"])
def test_stream_text_is_self_consistent(header):
    """an independent pass over the text: registers written by ds_read_b128 are not read before an s_waitcnt lgkmcnt that covers them
    (in-order return: the wait's count must be <= the number of LDS reads issued after the one in question)"""
    lines = [m.group(1) for m in re.finditer(r'^\s*"(.*?)\\n\\t"$', open(os.path.join(CSRC, header)).read(), re.M)]
    assert len(lines) > 5000

    def regs(tok):
        tok = tok.strip().strip(',')
        m = re.fullmatch(r'([va])\[(\d+):(\d+)\]', tok)
        if m:
            return {f"{m.group(1)}{i}" for i in range(int(m.group(2)), int(m.group(3)) + 1)}
        return {tok} if re.fullmatch(r'[va]\d+', tok) else set()
    pending = []                                             # destination register sets of outstanding LDS reads, oldest first
    checked = 0
    at_label = False
    for ln in lines:
        if ln.startswith(';'):
            continue
        if ln.endswith(':'):
            at_label = True
            continue
        if at_label and pending:
            assert ln.startswith(("s_nop", "s_cbranch")) or ln == "s_waitcnt lgkmcnt(0)", f"label reached with LDS reads in flight, then: {ln}"
        if not ln.startswith("s_nop"):
            at_label = False
        op, _, rest = ln.partition(' ')
        if op == "s_waitcnt":
            m = re.search(r'lgkmcnt\((\d+)\)', rest)
            if m:
                keep = int(m.group(1))
                pending = pending[len(pending) - keep:] if keep else []
            continue
        if op in ("s_call_b64", "s_setpc_b64", "s_branch"):
            assert not pending, f"control transfer with LDS reads in flight: {ln}"
            continue
        if op.startswith("s_cbranch"):                       # (forward skips over a call: the join label is followed by lgkmcnt(0), checked below)
            continue
        toks = [t for t in re.split(r'[ ,]+', rest) if t]
        used = set().union(*[regs(t) for t in toks]) if toks else set()
        for d in pending:
            assert not (d & used), f"{ln}: uses {sorted(d & used)[:4]} before the wait that covers its LDS read"
        if op in ("ds_read_b128", "ds_read_b32"):
            pending.append(regs(toks[0]))
            checked += 1
    assert checked > 500
