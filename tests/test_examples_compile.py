"""CPU check: the example drivers and tools at least compile (their GPU runs are in test_gpu_examples.py)."""
import glob
import os
import py_compile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_examples_and_tools_compile(tmp_path):
    files = sorted(glob.glob(os.path.join(ROOT, "examples", "*.py")) + glob.glob(os.path.join(ROOT, "tools", "*.py")))
    assert len(files) >= 8
    for k, f in enumerate(files):
        py_compile.compile(f, cfile=str(tmp_path / f"{k}.pyc"), doraise=True)
