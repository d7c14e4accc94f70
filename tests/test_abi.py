"""CPU-only: the C-ABI library builds, loads and exports every symbol that
include/ls_amd.h declares; no compute calls (no GPU here)."""
import ctypes as C
import os
import re

import pytest

import ls_amd
from ls_amd import _lib


def _declared_functions():
    with open(_lib.HEADER_PATH) as f:
        text = f.read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ls_[a-z0-9_]+)\s*\(", text)))


def test_library_is_built():
    assert os.path.exists(_lib.SO_PATH), "run `python constraint-solver_amd/build.py`"


def test_every_declared_symbol_is_exported_and_bound():
    lib = _lib.load()
    names = _declared_functions()
    assert len(names) >= 20
    for name in names:
        assert hasattr(lib, name), f"{name} declared in ls_amd.h but not exported"
        assert name in _lib.SIGNATURES, f"{name} has no ctypes signature"
    assert sorted(_lib.SIGNATURES) == names


def test_abi_version():
    assert _lib.load().ls_abi_version() == 1


def test_struct_layouts_match_header():
    assert C.sizeof(_lib.NqResult) == 32
    assert C.sizeof(_lib.SchedMove) == 12
    assert C.sizeof(_lib.SchedResult) == 32


def test_no_cpu_fallback_without_device():
    if ls_amd.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(ls_amd.LsError) as ei:
        ls_amd.Context(0)
    assert ei.value.code == _lib.LS_ERR_NO_DEVICE
    assert "no CPU fallback" in str(ei.value)


def test_product_never_references_oracle():
    root = os.path.dirname(os.path.dirname(os.path.abspath(_lib.__file__)))
    for dirpath, _, files in os.walk(root):
        for fn in files:
            if fn.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                with open(os.path.join(dirpath, fn)) as f:
                    src = f.read()
                assert "import oracle" not in src and "liboracle" not in src and "orc_" not in src, fn


def test_rust_sys_bindings_cover_the_header():
    # bindings/rust is source-only (no Rust toolchain here) but must at least name every C function
    root = os.path.dirname(os.path.dirname(os.path.abspath(_lib.HEADER_PATH)))
    with open(os.path.join(root, "bindings", "rust", "local-search-amd", "src", "sys.rs")) as f:
        rs = f.read()
    assert sorted(set(re.findall(r"pub fn (ls_[a-z0-9_]+)\(", rs))) == _declared_functions()


def test_rust_safe_wrapper_and_adaptors_cover_the_drop_in_path():
    # bindings/rust is source only; keep it honest mechanically: the safe wrapper must wrap every entry point the two
    # plug-in adaptors call, every wrapped C function must exist in the header, and the adaptors must implement the
    # reference's own traits (MoveProposer / SolutionScoreCalculator: local-search/src/local_search.rs:58-90) with the
    # reference's associated-type names, so that `local-search/` needs no edit.
    root = os.path.dirname(os.path.dirname(os.path.abspath(_lib.HEADER_PATH)))
    rs = os.path.join(root, "bindings", "rust")
    with open(os.path.join(rs, "local-search-amd", "src", "lib.rs")) as f:
        lib_rs = f.read()
    wrapped_c = sorted(set(re.findall(r"sys::(ls_[a-z0-9_]+)\(", lib_rs)))
    assert set(wrapped_c) <= set(_declared_functions())
    for need in ("ls_ctx_create", "ls_ctx_destroy", "ls_last_error", "ls_nqueens_col_scores", "ls_nqueens_eval_batch_next",
                 "ls_nqueens_eval", "ls_sched_instance_upload", "ls_sched_score", "ls_sched_eval", "ls_sched_moves_upload",
                 "ls_sched_eval_resident", "ls_sched_eval_exhaustive"):
        assert need in wrapped_c, need
    safe_fns = set(re.findall(r"pub fn ([a-z_0-9]+)\(", lib_rs))
    for ex, traits in (("nqueens", ("NQueensSolution", "NQueensScore")), ("employee-scheduling", ("ScheduleSolution", "ScheduleScore"))):
        with open(os.path.join(rs, "examples", ex, "gpu.rs")) as f:
            gpu = f.read()
        called = set(re.findall(r"\.ctx\s*\.\s*([a-z_0-9]+)\(", gpu)) | set(re.findall(r"\.ctx\n\s*\.([a-z_0-9]+)\(", gpu))
        assert called and called <= safe_fns, (ex, called - safe_fns)
        assert "impl MoveProposer for" in gpu and "impl SolutionScoreCalculator for" in gpu
        assert "type R = rand_chacha::ChaCha20Rng;" in gpu and f"type Solution = {traits[0]};" in gpu
        assert f"type _Solution = {traits[0]};" in gpu and f"type _Score = {traits[1]};" in gpu
        assert "Box<dyn Iterator<Item = Self::Solution>>" in gpu
        with open(os.path.join(rs, "examples", ex, ex + ".patch")) as f:
            patch = f.read()
        touched = re.findall(r"^\+\+\+ b/(\S+)", patch, flags=re.M)
        assert touched and not any(t.startswith("local-search/") for t in touched), touched   # zero diff in local-search/


def _build_c_example(tmp_path, name="nqueens_c"):
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(_lib.HEADER_PATH)))
    exe = str(tmp_path / name)
    libdir = os.path.dirname(_lib.SO_PATH)
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-O2", "-I", os.path.dirname(_lib.HEADER_PATH),
                    os.path.join(root, "examples", name + ".c"), "-o", exe, "-L", libdir, "-lls_amd",
                    f"-Wl,-rpath,{libdir}"], check=True, capture_output=True, text=True, timeout=120)
    return exe


def test_header_is_plain_c_and_c_example_links(tmp_path):
    # the boundary is a C ABI: the header must compile as strict C99 and a C program must link against the library
    import subprocess
    src = tmp_path / "hdr.c"
    src.write_text('#include "ls_amd.h"\nint main(void) { return (int)sizeof(ls_nq_result) - 32; }\n')
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.dirname(_lib.HEADER_PATH),
                    "-c", str(src), "-o", str(tmp_path / "hdr.o")], check=True, capture_output=True, timeout=60)
    _build_c_example(tmp_path, "employee_scheduling_c")
    exe = _build_c_example(tmp_path)
    if ls_amd.device_count() == 0:                 # and, like every entry point, it fails loudly without a GPU
        r = subprocess.run([exe, "8", "42"], capture_output=True, text=True, timeout=60)
        assert r.returncode == 1 and "no CPU fallback" in r.stderr
