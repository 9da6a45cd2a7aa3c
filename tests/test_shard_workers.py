"""CPU-only: the worker pool behind rsp_mcsc_column_sums (rcppsparse_amd/csrc/shard_workers.hpp -- the code that
librcppsparse_hip.so ships, pure host C++): the threads that stay parked on a futex between the calls of the
single-process multi-GPU handle (the reference has no threads on this path, src/example.cpp:26-32 runs on
the R main thread -- which is exactly why these must never lose a wake-up or leave a thread behind).  Built plain, under
ThreadSanitizer and under the address / undefined-behaviour sanitizers (the GPU pool offers no sanitizers), each with
workers that park at once (RSP_MCSC_SPIN_US=0) and with the default spin window."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "c", "shard_workers_selftest.cpp")


@pytest.mark.parametrize("sanitize", ["plain", "tsan", "asan-ubsan"])
def test_worker_pool_never_loses_a_wake_up_or_a_thread(tmp_path, sanitize):
    exe = str(tmp_path / "shard_workers_selftest")
    flags = {"plain": ["-O2"],
             "tsan": ["-O1", "-g", "-fsanitize=thread"],
             "asan-ubsan": ["-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-fno-sanitize-recover=all"]}[sanitize]
    subprocess.run(["g++", "-std=c++17", "-Wall", "-Wextra", "-pthread", *flags, SRC, "-o", exe], check=True)
    rounds = "6000" if sanitize == "plain" else "1500"
    for spin in ("0", "50"):
        r = subprocess.run([exe, rounds], capture_output=True, text=True, timeout=600,
                           env=dict(os.environ, RSP_MCSC_SPIN_US=spin, ASAN_OPTIONS="detect_leaks=1", TSAN_OPTIONS="halt_on_error=1"))
        assert r.returncode == 0, (spin, r.stdout[-2000:] + r.stderr[-3000:])
        assert "shard workers selftest ok" in r.stdout
