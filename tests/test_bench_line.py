"""The bench line's assembly, without a GPU: bench.assemble_line is pure, so the shape of the line at N = 1 and at
N = 8 (eight entries in config.shards, the c5 records of also_sharded, cpu_baseline on rank 0) is checked here, and
above all its SIZE: the driver keeps the last 8 KiB of stdout, and a line that does not fit loses its head (round 4:
`read_ceiling`, `latency` and `pipelined` never reached the driver's record)."""
import json
import math

import numpy as np
import pytest

import bench


def _args(*flags):
    return bench.parse_args(list(flags))


def _head(world, nnz=1_000_000_000, ncol=1_000_000, rehearsal=False, small=False):
    bounds = [k * ncol // world for k in range(world + 1)]
    xb = [k * nnz // world for k in range(world + 1)]
    use_comm = world > 1
    shards = [{"rank": r, "device": 0 if rehearsal else r, "c0": bounds[r], "c1": bounds[r + 1], "x0": xb[r], "x1": xb[r + 1],
               "kernel_ms": 0.15512345678 + 1e-3 * r, "gather_ms": 0.0212345678 if use_comm else None} for r in range(world)]
    return {"workload": "c3", "nrow": 10_000_000, "ncol": ncol, "nnz": nnz, "shape": "uniform", "partition": "nnz", "world": world,
            "p": np.zeros(2, dtype=np.int32), "elapsed": 0.0246512345678 / world, "regions_ms": [0.021, 0.0209, 0.0215] if small else None,
            "kernel_ms": 1.2203585505485535 / world, "kernel_ms_median": 1.2205619812011719 / world,
            "kernel_ms_min": 1.2061229944229126 / world, "kernel_ms_max_over_ranks": 1.23 / world,
            "kernel_timing": "per_call_after" if use_comm else "per_call",
            "gather_ms": 0.02123 if use_comm else None, "gather_ms_min": 0.019 if use_comm else None,
            "gather_ms_max": 0.031 if use_comm else None, "gather_ms_max_over_ranks": 0.033 if use_comm else None,
            "algo_bytes": 8 * (nnz // world) + 12 * (ncol // world) + 4, "imbalance": 1.0000123456, "ncopies": 1,
            "shards": shards,
            "parity": {"max_abs_err_over_l1": 4.6e-16, "tolerance": 1e-12, "max_rel_err_where_ref_ge_1e-3_l1": 3.1e-13,
                       "columns_checked": "all", "ncol": ncol, "columns_out_of_tolerance": 0,
                       "empty_columns_exactly_plus_zero": True},
            "lat_med": 1.2612345678, "lat_min": 1.2512345678, "lat_med_max": 1.2712345678,
            "pipelined": {"value": 8.2e11, "ms_per_step": 1.219, "compute_streams": 2, "output_buffers": 4},
            "planned_shards": None if world == 1 else {"value": 4.1e12, "ms_per_step": 0.244, "forms_by_rank": ["columns"] * world,
                                                       "plan_ms_rank0": 1.93, "parity_err": 4.4e-16, "bad_columns": 0},
            "direct_gather": None if world == 1 else {"value": 3.9e12, "ms_per_step": 0.256, "parity_err": 4.4e-16, "bad_columns": 0},
            "host_gather": None if world == 1 else {"value": 3.6e12, "ms_per_step": 0.278, "parity_err": 4.4e-16, "bad_columns": 0},
            "peer_probe": None if world == 1 or rehearsal else "passed",
            "plan": None, "gather_name": "rsp_comm_gatherv (C ABI, RCCL)" if use_comm else None, "fell_back": False,
            "x0_for_ceiling": None}


def _also_records():
    recs = []
    for spec in bench.ALSO_AUTO:
        planned = ":" in spec
        recs.append(bench.sig({"workload": spec, "form": "lean" if planned else "general",
                               "launches": 1 if planned else 2, "plan_ms": 1.6123456 if planned else None,
                               "plan_by": ("device" if spec.endswith("device") else "host") if planned else None,
                               "early_general_calls": 3 if spec.endswith("device") else None, "steps": 200, "x_copies": 5,
                               "ms_per_call": 0.0157123456, "regions_ms": [0.0157123, 0.0158123, 0.0156123],
                               "kernel_ms": 0.0156123456, "host_stall_suspected": False,
                               "algo_bytes": 92_000_004, "frac": 0.7366123456, "parity_err": 0.0,
                               "bad_columns": 0, "traffic": 92_200_000.0, "traffic_in_run": spec in bench.ALSO_TRAFFIC_NOW}))
    return recs


CPU = {"value": 2.56207e9, "unit": "nnz/s", "cores": 1, "kind": "port",
       "sample": "first 1000000 columns (1000000000 nnz) of the same matrix, 10 reps, median", "best": 2.577e9,
       "host_cpus": 256, "all_cores_value": 3.1e10, "all_cores": 16}


def test_default_n1_line_fits_the_drivers_tail_and_carries_the_flat_scalars():
    extras = {"traffic": {"bytes": 8032627189.3, "in_run": True, "file": None, "read_bytes": 8020620928.0,
                          "write_bytes": 12148224.0, "seconds": 3.44},
              "read_ceiling": {"GBps": 6941.2, "ms_per_launch": 1.1525, "reps": 5},
              "also": _also_records(), "also_seconds": 17.1, "cpu_baseline": CPU,
              "mcsc": {"mcsc8_kernels_us": 46.12345, "mcsc8_serial_call_us": 63.12345, "mcsc8_serial_overhead_us": 17.0123,
                       "mcsc8_serial_last_enqueue_us": 24.51234, "mcsc8_workers_call_us": 53.12345,
                       "mcsc8_workers_overhead_us": 7.012345, "mcsc8_workers_last_enqueue_us": 11.71234}}
    line = bench.assemble_line(_args(), _head(1), extras)
    text = json.dumps(line)
    assert len(text) < bench.LINE_BYTES_MAX - 700, len(text)          # (headroom: real numbers print longer than round ones)
    assert line["roofline"]["mcsc8_workers_overhead_us"] == pytest.approx(7.01235, rel=1e-5) and line["roofline"]["mcsc8_serial_call_us"] > 60
    roof = line["roofline"]
    # what the driver's record keeps: scalars of `roofline` -- every figure of the run has to be one
    for k in ("read_ceiling_GBps", "frac_of_ceiling", "traffic_over_algorithmic", "traffic_measured_in_run"):
        assert k in roof and not isinstance(roof[k], (dict, list)), k
    for spec in bench.ALSO_AUTO:
        k = "also_" + bench.key_of(spec)
        for f in ("_frac", "_kernel_ms", "_ms_per_call", "_parity_err"):
            assert isinstance(roof[k + f], float), k + f
    assert roof["also_c2_traffic_x"] == pytest.approx(92_200_000.0 / 92_000_004, rel=1e-5)
    assert "also_c2_planned_traffic_recorded_x" in roof and "also_c2_planned_traffic_x" not in roof
    assert roof["frac_of_ceiling"] == pytest.approx(roof["achieved"] / 6941.2, rel=1e-5)
    assert roof["traffic_over_algorithmic"] == pytest.approx(8032627189.3 / roof["algorithmic_bytes_per_launch"], rel=1e-5)
    assert all(not isinstance(v, (dict, list)) for v in roof.values())
    assert "also" in line and "also" not in roof                        # the records travel once
    assert line["cpu_baseline"]["cores"] == 1 and line["config"]["parallelism"] == "single"
    assert line["value"] == 1_000_000_000 * 20 / _head(1)["elapsed"]    # the contract's own keys stay exact
    assert "notes" not in line
    assert line["config"]["host_stall_suspected"] is False


@pytest.mark.parametrize("world", [2, 4, 6, 8])
def test_n_rank_line_fits_and_lists_every_shard(world):
    sharded = {"c5_nnz": {"value": 5.1e12, "ms_per_step": 0.196, "imbalance": 1.0071, "kernel_ms_max": 0.158,
                          "gather_ms_max": 0.031, "kernel_ms_by_rank": [0.155] * world, "nnz_by_rank": [125_000_000] * world,
                          "parity_err": 4.5e-16, "bad_columns": 0},
               "c5_cols": {"value": 3.0e12, "ms_per_step": 0.33, "imbalance": 1.93, "kernel_ms_max": 0.29,
                           "gather_ms_max": 0.03, "kernel_ms_by_rank": [0.29] * world, "nnz_by_rank": [241_000_000] * world,
                           "parity_err": 4.5e-16, "bad_columns": 0}}
    refusal = ["ncclCommInitRank failed: error 5 (invalid usage)"] * world
    line = bench.assemble_line(_args("--gpus", str(world), "--rendezvous", "gloo"), _head(world, rehearsal=True),
                               {"also_sharded": sharded, "cpu_baseline": CPU}, devices=1, rehearsal=True, comm_rehearsal=refusal)
    text = json.dumps(line)
    assert len(text) < bench.LINE_BYTES_MAX, len(text)
    cfg = line["config"]
    assert cfg["parallelism"] == "rehearsal" and cfg["devices"] == 1 and cfg["comm_refused_on_ranks"] == world
    assert [s["rank"] for s in cfg["shards"]] == list(range(world))
    assert sum(s["nnz"] for s in cfg["shards"]) == 1_000_000_000 and cfg["shards"][-1]["c1"] == 1_000_000
    roof = line["roofline"]
    for k in ("also_c5_nnz_value", "also_c5_nnz_imbalance", "also_c5_cols_value", "also_c5_cols_imbalance",
              "also_c5_nnz_parity_err", "also_c5_cols_ms_per_step"):
        assert isinstance(roof[k], float), k
    assert line["also_sharded"]["c5_cols"]["kernel_ms_by_rank"] == [0.29] * world
    assert line["cpu_baseline"]["kind"] == "port"                       # rank 0 at every N
    assert line["n_gpus"] == world and line["scaling"] == "strong"
    threads = {"threads_value": 4.4e12, "threads_ms_per_step": 0.2271234, "threads_launch": "workers", "threads_devices_distinct": True,
               "threads_parity_err": 4.6e-16, "threads_seconds": 31.2, "threads_d2h_pinned_ms": 0.2271234,
               "threads_d2h_pageable_ms": 0.2561234, "threads_stores_pinned_ms": 0.2391234, "threads_none_pinned_ms": 0.1791234,
               "threads_rccl_pinned_ms": 0.2931234, "threads_last_enqueue_us": 11.71234, "threads_serial_d2h_pinned_ms": 0.2461234,
               "threads_call_minus_slowest_kernel_us": 74.31234, "threads_kernel_ms_max_over_shards": 0.1521234}
    real = bench.assemble_line(_args("--gpus", str(world)), _head(world),
                               {"cpu_baseline": CPU, "also_sharded": sharded, "threads": threads,
                                "rccl": {"version": 22606, "library": "/usr/local/lib/python3.10/dist-packages/torch/lib/librccl.so"}},
                               devices=world)
    assert real["config"]["parallelism"] == "ranges+rccl" and "comm_refused_on_ranks" not in real["config"]
    # round 6: everything the driver's first real N > 1 run can tell, as flat scalars, and still under the 8 KiB tail
    assert len(json.dumps(real)) < bench.LINE_BYTES_MAX, len(json.dumps(real))
    assert real["config"]["rccl_version"] == 22606 and real["config"]["rccl_library"].endswith("librccl.so")
    assert real["config"]["peer_probe"] == "passed"
    rr = real["roofline"]
    for k in ("threads_value", "threads_d2h_pinned_ms", "threads_rccl_pinned_ms", "threads_last_enqueue_us",
              "direct_gather_value", "host_gather_value", "planned_shards_value", "pipelined_value", "host_gather_ms_per_step"):
        assert isinstance(rr[k], float), k
    assert real["host_gather"]["value"] == 3.6e12 and rr["threads_launch"] == "workers"
    assert all(not isinstance(v, (dict, list)) for v in rr.values())
    failed = bench.assemble_line(_args("--gpus", str(world)), _head(world), {"threads": {"threads_error": "exit 1: boom"}}, devices=world)
    assert failed["roofline"]["threads_error"] == "exit 1: boom" and "threads_value" not in failed["roofline"]


def test_small_headline_reports_its_regions_and_flags_a_host_stall():
    H = _head(1, nnz=10_000_000, small=True)
    H["kernel_timing"] = "region"
    H["kernel_ms"] = 0.0156
    H["elapsed"] = 20 * 0.1996e-3                                      # round 4's bad sample: 0.1996 ms per call, 0.0156 ms of kernels
    line = bench.assemble_line(_args("--workload", "c2"), H, {})
    assert line["config"]["host_stall_suspected"] is True and line["config"]["regions_ms"] == [0.021, 0.0209, 0.0215]
    H["elapsed"] = 20 * 0.0158e-3
    assert bench.assemble_line(_args("--workload", "c2"), H, {})["config"]["host_stall_suspected"] is False


def test_median_region_and_rounding_helpers():
    assert bench.median_region([(3.0, 1.0), (1.0, 2.0), (2.0, 3.0)]) == (2.0, 3.0)
    assert bench.sig(1.23456789e-5) == 1.23457e-5 and bench.sig(7) == 7 and bench.sig(True) is True
    assert bench.sig({"a": [0.123456789, {"b": 1e9 / 3}]}) == {"a": [0.123457, {"b": 333333000.0}]}
    assert bench.key_of("c2:planned-device") == "c2_planned_device"
    assert math.isclose(bench.sig(2.0 / 3.0), 0.666667)


def test_verbose_line_carries_the_glossary():
    line = bench.assemble_line(_args("--verbose"), _head(1), {})
    assert set(line["notes"]) == set(bench.GLOSSARY)


def test_peer_probe_runner_turns_every_failure_into_a_verdict(tmp_path):
    """bench.run_peer_probe starts throw-away processes before kernels are allowed to store across devices (direct_gather on a
    real multi-GPU node).  Whatever happens to the child -- it dies on a signal as a process does whose queue faulted, it exits
    non-zero, it hangs, it prints nothing -- the runner returns a 'failed: ...' verdict and never raises; a child that says
    'passed' / 'same device' / 'no peer access' is believed."""
    import sys
    py = sys.executable
    assert bench.run_peer_probe(0, 1, command=[py, "-c", "print('passed')"]) == "passed"
    assert bench.run_peer_probe(0, 0, command=[py, "-c", "print('noise'); print('same device')"]) == "same device"
    assert bench.run_peer_probe(0, 1, command=[py, "-c", "print('no peer access')"]) == "no peer access"
    v = bench.run_peer_probe(0, 1, command=[py, "-c", "import os, signal; os.kill(os.getpid(), signal.SIGSEGV)"])
    assert v.startswith("failed: the probe exited with")
    v = bench.run_peer_probe(0, 1, command=[py, "-c", "import sys; sys.stderr.write('Memory access fault by GPU node-2'); sys.exit(134)"])
    assert v.startswith("failed: the probe exited with 134") and "Memory access fault" in v
    v = bench.run_peer_probe(0, 1, timeout_s=1, command=[py, "-c", "import time; time.sleep(30)"])
    assert v == "failed: the probe did not finish in 1 s"
    assert bench.run_peer_probe(0, 1, command=[py, "-c", "pass"]) == "failed: the probe printed no verdict"
    assert bench.run_peer_probe(0, 1, command=[str(tmp_path / "no_such_program")]).startswith("failed: the probe could not be started")
    # and the line says what was seen instead of measuring
    H = _head(2)
    H["direct_gather"] = {"value": None, "note": "not measured: peer probe: no peer access"}
    H["peer_probe"] = "no peer access"
    line = bench.assemble_line(_args("--gpus", "2"), H, {}, devices=2)
    assert line["config"]["peer_probe"] == "no peer access" and "direct_gather_value" not in line["roofline"]
    assert line["direct_gather"]["note"].endswith("no peer access")
