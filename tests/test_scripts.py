"""CPU-only: every script of the repository at least byte-compiles, and the ones that take --help print it without
touching a GPU (argument parsing happens before any device work)."""
import glob
import os
import py_compile
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCRIPTS = sorted(glob.glob(os.path.join(ROOT, "tools", "*.py")) + glob.glob(os.path.join(ROOT, "examples", "*.py")) +
                 [os.path.join(ROOT, "bench.py"), os.path.join(ROOT, "__graft_entry__.py")])


@pytest.mark.parametrize("path", SCRIPTS, ids=[os.path.relpath(p, ROOT) for p in SCRIPTS])
def test_script_compiles(path, tmp_path):
    py_compile.compile(path, cfile=str(tmp_path / "x.pyc"), doraise=True)


@pytest.mark.parametrize("rel", ["bench.py", "tools/run_config4.py", "tools/run_config5.py", "tools/multi_sched_trace.py",
                                 "examples/nqueens.py", "examples/employee_scheduling.py"])
def test_help_needs_no_gpu(rel):
    r = subprocess.run([sys.executable, os.path.join(ROOT, rel), "--help"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "usage" in r.stdout.lower(), r.stderr[-500:]
