"""CPU: every python source of the repo parses (tools/ scripts run only on the GPU box — a syntax error there costs a gpurun call),
and no product module imports the oracle (it is test infrastructure: tests/, __graft_entry__.smoke() and bench.py's baselines only)."""
import ast
import glob
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _sources():
    out = []
    for pat in ("*.py", "tools/*.py", "tests/*.py", "tests/golden/*.py", "oracle/*.py", "diffusion-by-maxentirl_amd/**/*.py"):
        out += glob.glob(os.path.join(ROOT, pat), recursive=True)
    return sorted(set(out))


def test_every_python_source_parses():
    files = _sources()
    assert len(files) > 60
    for f in files:
        with open(f) as fh:
            src = fh.read()
        tree = ast.parse(src, filename=f)
        # "string" (expr) with nothing between them is a call of a str: the bug class that broke tools/refresh_profiles.py in round 6
        for node in ast.walk(tree):
            if isinstance(node, ast.Call) and isinstance(node.func, (ast.Constant, ast.JoinedStr)):
                raise AssertionError(f"{f}:{node.lineno}: a string literal is called (missing operator or comma?)")


def test_product_package_never_imports_the_oracle():
    for f in glob.glob(os.path.join(ROOT, "diffusion-by-maxentirl_amd", "**", "*.py"), recursive=True):
        tree = ast.parse(open(f).read(), filename=f)
        for node in ast.walk(tree):
            names = []
            if isinstance(node, ast.Import):
                names = [a.name for a in node.names]
            elif isinstance(node, ast.ImportFrom) and node.module:
                names = [node.module]
            assert not any(n == "oracle" or n.startswith("oracle.") for n in names), f"{f}:{node.lineno} imports the oracle"
