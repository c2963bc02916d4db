"""Which source lines do an iteration's launches come from?  A TorchDispatchMode that counts every ATen operation of ONE call of `fn` by the
innermost frame of this repository on the Python stack; operations run by autograd's backward pass are counted against the line that built the
node (anomaly mode records that traceback).  An ATen operation is not a kernel launch one-to-one (views launch nothing, a `cat` may launch
two), so read the output as where the launches are, not as their exact number; `rocprofv3 --kernel-trace --stats` has the number.
    NEUMAN_LAUNCH_SOURCES=1 python tools/human_step_bench.py 2048 5      (prints the table after the timed iterations)"""
import collections
import os
import sys

import torch
from torch.utils._python_dispatch import TorchDispatchMode

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
_VIEWS = ('view', 'reshape', 'expand', 'slice', 'select', 'unsqueeze', 'squeeze', 'permute', 'transpose', 't.', 'detach', 'alias', 'as_strided', 'unbind', 'split',
          'empty', '_unsafe_view', 'lift_fresh', 'is_', 'size', 'stride', 'numel', 'dim', 'sym_', '_local_scalar_dense', 'narrow', 'unfold', 'chunk', 'result_type',
          'can_cast', 'item', 'set_', 'resize_', 'record_stream', '_has_compatible_shallow_copy_type')


def _launches(name):
    short = name.split('.')[1] if '.' in name else name
    return not any(short.startswith(v) for v in _VIEWS)


def _repo_frame():
    try:
        f = sys._getframe(2)
    except ValueError:                                                   # (autograd's own thread: no Python frames above the dispatcher's)
        return None
    while f is not None:
        fn = f.f_code.co_filename
        if fn.startswith(ROOT) and not fn.endswith('launch_sources.py'):
            return f"{os.path.relpath(fn, ROOT)}:{f.f_lineno} {f.f_code.co_name}"
        f = f.f_back
    return None


def _node_frame(node):
    tb = node.metadata.get('traceback_') if node is not None else None
    if not tb:
        return None
    for line in reversed(tb):                                            # 'File "...", line N, in fn' entries, innermost last
        line = line.strip()
        if line.startswith('File "' + ROOT) and 'launch_sources.py' not in line:
            path, rest = line[6:].split('", line ', 1)
            no, fn = rest.split(', in ', 1)
            return f"{os.path.relpath(path, ROOT)}:{no} {fn.splitlines()[0]}"
    return None


class Counter(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.by_line = collections.Counter()
        self.ops = collections.defaultdict(collections.Counter)

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if _launches(name):
            node = torch._C._current_autograd_node()
            where = _repo_frame()
            if node is not None:
                where = f"backward of {_node_frame(node) or type(node).__name__}" if where is None or 'backward' not in where else f"backward in {where}"
            where = where or "(no repository frame)"
            self.by_line[where] += 1
            self.ops[where][name.replace('aten.', '').replace('.default', '')] += 1
        return func(*args, **(kwargs or {}))


def count(fn, top=45, out=sys.stderr):
    c = Counter()
    with torch.autograd.set_detect_anomaly(True, check_nan=False), c:
        fn()
    total = sum(c.by_line.values())
    print(f"[launch sources] {total} launching ATen operations in one call", file=out)
    for where, n in c.by_line.most_common(top):
        ops = " ".join(f"{k}x{v}" for k, v in c.ops[where].most_common(6))
        print(f"  {n:4d}  {where}   [{ops}]", file=out)
    return c
