"""Host logic, no GPU: the lockstep batcher of csrc/host/plugins.hpp (ls_multi_solver's launch merger) compiled against
a fake CPU C ABI (tests/cpp/batcher_fault_test.cpp) and run under AddressSanitizer and ThreadSanitizer.  Covers the
failure path ADVICE r1 flagged: a failing restart must not leave parked requests of unwound stacks behind."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "batcher_fault_test.cpp")


@pytest.mark.parametrize("sanitizer", ["address,undefined", "thread"])
def test_batcher_failure_paths_under_sanitizers(tmp_path, sanitizer):
    exe = str(tmp_path / "batcher_fault_test")
    subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-mavx2", f"-fsanitize={sanitizer}", "-fno-omit-frame-pointer", SRC, "-o", exe,
                    "-lpthread"], check=True, capture_output=True, text=True, timeout=300)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1", TSAN_OPTIONS="halt_on_error=1")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), (r.stdout[-2000:], r.stderr[-4000:])


def test_weighted_sampler_histogram_selection_equals_selection_sort(tmp_path):
    # csrc/host/rng.hpp: index_sample_weighted (histogram top-k) vs index_sample_weighted_reference (selection + sort)
    src = os.path.join(ROOT, "tests", "cpp", "sample_weighted_test.cpp")
    exe = str(tmp_path / "sample_weighted_test")
    subprocess.run(["g++", "-std=c++17", "-O2", "-g", "-mavx2", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", src, "-o", exe],
                   check=True, capture_output=True, text=True, timeout=300)
    # the bracket / filter loops of the pre-filter have an AVX-512 form and a scalar form: both
    for extra in ({}, {"LS_HOST_NO_AVX512": "1"}):
        r = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=dict(os.environ, **extra))
        assert r.returncode == 0 and r.stdout.strip().endswith("OK"), (extra, r.stdout[-2000:], r.stderr[-4000:])


def test_fibre_pool(tmp_path):
    # csrc/host/fibers.hpp: ls_multi_solver's fibre pool (hand-written context switch, guard-paged stacks, idle workers take
    # suspended fibres from busy ones) -- plain and under UBSan (ASan / TSan would need their fibre annotations)
    src = os.path.join(ROOT, "tests", "cpp", "fiber_pool_test.cpp")
    for flags in (["-O2"], ["-O1", "-fsanitize=undefined", "-fno-omit-frame-pointer"]):
        exe = str(tmp_path / ("fiber_pool_test" + str(len(flags))))
        subprocess.run(["g++", "-std=c++17", "-g", *flags, src, "-o", exe, "-lpthread"], check=True, capture_output=True, text=True, timeout=300)
        for _ in range(3):
            r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
            assert r.returncode == 0 and r.stdout.strip().endswith("OK"), (flags, r.stdout[-2000:], r.stderr[-4000:])


@pytest.mark.parametrize("sanitizer", ["address,undefined", "thread"])
def test_server_host_state_machine_under_sanitizers(tmp_path, sanitizer):
    # csrc/srv_common.hpp (SrvHost: post / wait / relaunch / give-up / pause of the resident servers) against a fake HIP runtime
    # and a fake device thread that answers, drains and goes silent (tests/cpp/srv_host_test.cpp): the 16-bit tag wrap, the
    # no-answer error, SRV_RC_GIVE_UP, a failing launch, QUIT as a whole multi-line message
    src = os.path.join(ROOT, "tests", "cpp", "srv_host_test.cpp")
    exe = str(tmp_path / "srv_host_test")
    subprocess.run(["g++", "-std=c++17", "-O1", "-g", f"-fsanitize={sanitizer}", "-fno-omit-frame-pointer", "-Wno-tsan",
                    "-I", os.path.join(ROOT, "tests", "cpp", "fake_hip"), src, "-o", exe, "-lpthread"],
                   check=True, capture_output=True, text=True, timeout=300)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1", TSAN_OPTIONS="halt_on_error=1")
    for _ in range(2):
        r = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0 and r.stdout.strip().endswith("OK"), (r.stdout[-2000:], r.stderr[-4000:])
