"""The oracle is test infrastructure: nothing shipped (ml-neuman_amd/, bench.py's GPU leg) may import it."""
import ast
import os
import re

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def oracle_imports(tree):
    """(node, enclosing function name or None) for every import of the oracle package."""
    found = []

    def visit(node, fn):
        for child in ast.iter_child_nodes(node):
            f = child.name if isinstance(child, (ast.FunctionDef, ast.AsyncFunctionDef)) else fn
            if isinstance(child, ast.Import) and any(a.name.split('.')[0] == 'oracle' for a in child.names):
                found.append((child, fn))
            if isinstance(child, ast.ImportFrom) and (child.module or '').split('.')[0] == 'oracle':
                found.append((child, fn))
            visit(child, f)
    visit(tree, None)
    return found


def test_package_never_imports_oracle():
    pkg = os.path.join(ROOT, "ml-neuman_amd")
    offenders = []
    for d, _, files in os.walk(pkg):
        for f in files:
            path = os.path.join(d, f)
            if f.endswith(".py"):
                if oracle_imports(ast.parse(open(path).read())):
                    offenders.append(path)
            elif f.endswith((".hip", ".h", ".cpp")):
                if re.search(r'#include\s+"[^"]*oracle', open(path, errors="replace").read()):
                    offenders.append(path)
    assert not offenders, offenders


def test_bench_uses_oracle_only_for_cpu_baseline():
    tree = ast.parse(open(os.path.join(ROOT, "bench.py")).read())
    imps = oracle_imports(tree)
    assert imps, "bench.py must time the CPU baseline with the oracle"
    assert all(fn == "cpu_baseline" for _, fn in imps), [(ast.dump(n), fn) for n, fn in imps]


def test_graft_entry_uses_oracle_only_in_smoke():
    tree = ast.parse(open(os.path.join(ROOT, "__graft_entry__.py")).read())
    assert all(fn == "smoke" for _, fn in oracle_imports(tree))
