"""The oracle is test infrastructure: nothing under scratchpad_amd/ (or bench.py's timed GPU path)
may import or call it, and the product has no torch/CPU fallback for its ops."""
import ast
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "scratchpad_amd")


def _py_files(d):
    for base, _, files in os.walk(d):
        for f in files:
            if f.endswith(".py"):
                yield os.path.join(base, f)


def test_product_never_imports_oracle_or_tests():
    for path in _py_files(PKG):
        tree = ast.parse(open(path).read(), path)
        for node in ast.walk(tree):
            names = []
            if isinstance(node, ast.Import):
                names = [a.name for a in node.names]
            elif isinstance(node, ast.ImportFrom):
                names = [node.module or ""]
            for n in names:
                assert not n.split(".")[0] in ("oracle", "tests"), f"{path} imports {n}"
        assert "/root/reference" not in open(path).read(), f"{path} mentions the reference tree path"


def test_native_sources_do_not_reference_oracle():
    for base, _, files in os.walk(os.path.join(PKG, "csrc")):
        for f in files:
            assert "oracle" not in open(os.path.join(base, f)).read()


def test_bench_uses_oracle_only_in_cpu_baseline():
    src = open(os.path.join(ROOT, "bench.py")).read()
    tree = ast.parse(src)
    for node in ast.walk(tree):
        if isinstance(node, ast.FunctionDef):
            body = ast.get_source_segment(src, node)
            if re.search(r"\boracle\b", body) and node.name not in ("cpu_baseline", "cpu_baseline_cfg1"):
                # docstring mentions are fine; imports are not
                assert not re.search(r"^\s*(from|import)\s+oracle", body, flags=re.M), node.name
    top_level_imports = [n for n in tree.body if isinstance(n, (ast.Import, ast.ImportFrom))]
    for n in top_level_imports:
        mod = n.module if isinstance(n, ast.ImportFrom) else n.names[0].name
        assert not (mod or "").startswith("oracle")


def test_ops_have_no_native_fallback_methods():
    from scratchpad_amd.custom_op import CustomOp
    from scratchpad_amd.layers import RMSNorm, RotaryEmbedding, SiluAndMul
    for cls in (CustomOp, RMSNorm, SiluAndMul, RotaryEmbedding):
        assert not hasattr(cls, "forward_native"), cls
    op = SiluAndMul()
    assert op._forward_method == op.forward_hip
