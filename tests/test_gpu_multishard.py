"""-m gpu tests of the N > 1 path on the ONE GPU this box has:

  * bench.py itself, run as a child process with --force-comm, so the communicator created
    through the C ABI, the in-order kernel -> gatherv call, the latency loop and the pipelined
    figure are under the driver's own test run;
  * BASELINE configs 4 and 5 at FULL size: the 8 nnz-balanced column-range shards of the
    1e9-nnz uniform and Zipf matrices, generated in HBM at their offsets exactly as bench.py
    does, summed one after the other through rsp_column_sums_device and reassembled.

  * bench.py with N > 1 ranks and real HIP compute: `--rendezvous gloo`
    lets the ranks share this box's one GPU, so the rank != 0 branches of bench.py, the shard
    offsets x0 > 0, the gathered whole-matrix parity and the max-over-ranks statistics all run.

The 8-rank RCCL exchange itself needs 8 GPUs (the driver's node); its layout and driver are
covered by tests/test_sharded_gloo.py, and a one-rank communicator runs here.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle
from rcppsparse_amd import capi, sharded, synth

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RTOL = 1e-12


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("gpu-marked tests need a GPU: the HIP path has no CPU fallback")
    capi.load()
    yield torch
    capi.set_tuning(0)


def _run_bench(*flags, timeout=900):
    # (bench.py itself puts HSA_ENABLE_IPC_MODE_LEGACY=0 into its environment before the HIP runtime starts)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]          # the contract: ONE JSON line
    d = json.loads(lines[0])
    d["_line_bytes"] = len(lines[0])                  # (the driver keeps the last 8 KiB of stdout)
    return d


def test_bench_child_process_runs_the_comm_path_at_shard_size(torch_cuda):
    """`bench.py --gpus 1 --force-comm --workload c4shard`: one C4 shard (1.25e8 nnz), the
    communicator of the C ABI, the gatherv inside every call."""
    torch_cuda.cuda.empty_cache()
    d = _run_bench("--gpus", "1", "--force-comm", "--workload", "c4shard", "--steps", "8", "--warmup", "2",
                   "--no-cpu-baseline")
    assert d["n_gpus"] == 1 and d["steps"] == 8 and d["unit"] == "nnz/s"
    assert d["config"]["gather"] == "rsp_comm_gatherv (C ABI, RCCL)"
    assert d["config"]["gather_fell_back_to_torch_distributed"] is False
    par = d["parity"]
    assert par["columns_checked"] == "all" and par["ncol"] == 125_000
    assert par["columns_out_of_tolerance"] == 0 and par["max_abs_err_over_l1"] <= RTOL
    assert par["empty_columns_exactly_plus_zero"] is True
    # both protocols are reported, and the like-for-like one is the metric
    assert d["latency_ms_per_call"] > 0 and d["latency"]["calls"] >= 1
    assert d["pipelined"]["value"] > 0 and d["pipelined"]["compute_streams"] == 2
    assert d["value"] == pytest.approx(125_000_000 * 8 / (d["ms_per_step"] * 8e-3), rel=1e-9)
    # a call cannot be faster than its kernels; a single call pays the host round trip on top
    roof = d["roofline"]
    assert roof["kernel_ms"] <= d["ms_per_step"] * 1.06        # (the split comes from a second pass with events around every call)
    assert d["latency_ms_per_call"] >= roof["kernel_ms_min"]
    # the mean gather time lies between the individual gather timings
    assert roof["gather_ms_min"] <= roof["gather_ms"] <= roof["gather_ms_max"]
    assert roof["kernel_timing"] == "per_call_after"                   # a call with a gather: the timed region carries no events
    assert 0.3 < roof["frac"] < 1.0


def test_bench_child_process_default_protocol_small(torch_cuda):
    """The N = 1 protocol of the bench line (no communicator), on a small Zipf matrix, with the
    CPU baseline leg: every key of the contract is there and parity covers every column."""
    torch_cuda.cuda.empty_cache()
    d = _run_bench("--workload", "tiny", "--steps", "6", "--warmup", "2")
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline",
                "latency_ms_per_call", "pipelined", "parity"):
        assert key in d, key
    assert d["config"]["gather"] is None and d["dtype"] == "f64" and d["vs_baseline"] is None
    assert d["parity"]["columns_checked"] == "all" and d["parity"]["columns_out_of_tolerance"] == 0
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] == 1
    assert set(d["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"}


def test_bench_line_measures_its_own_ceiling_traffic_and_more_workloads(torch_cuda):
    """The N = 1 line carries, measured in the same run on the same device and as FLAT SCALARS of `roofline` (what the driver's
    record keeps): the read ceiling (a read-only kernel with the column sums' access shape over the same x), the HBM traffic
    from two rocprofv3 counter passes run as child processes (also for the plan-free `also` record c2), and per `also`
    workload frac / kernel ms / ms per call (median of three regions) / parity -- here a C4 shard as the headline (a small
    call: three regions of its own) and C2 three ways.  The whole line fits the driver's 8 KiB tail."""
    torch_cuda.cuda.empty_cache()
    d = _run_bench("--workload", "c4shard", "--steps", "10", "--warmup", "3", "--no-cpu-baseline", "--traffic-pass", "on",
                   "--also", "c2,c2:planned,c2:planned-device")
    assert d["_line_bytes"] < 8000
    roof = d["roofline"]
    assert all(not isinstance(v, (dict, list)) for v in roof.values())
    assert 3000 < roof["read_ceiling_GBps"] < 8000 and roof["read_ceiling_reps"] == 5
    assert 0.5 < roof["frac_of_ceiling"] < 1.05
    assert roof["frac_of_ceiling"] == pytest.approx(roof["achieved"] / roof["read_ceiling_GBps"], rel=1e-4)
    assert roof["traffic_measured_in_run"] is True, roof.get("traffic_not_measured")
    assert 0.98 < roof["traffic_over_algorithmic"] < 1.10                                # 1 GB of x: past every cache
    assert roof["traffic"] == pytest.approx(roof["traffic_over_algorithmic"] * roof["algorithmic_bytes_per_launch"], rel=1e-4)
    assert roof["traffic_read_bytes"] > 100 * roof["traffic_write_bytes"]
    assert roof["kernel_timing"] == "region" and len(d["config"]["regions_ms"]) == 3     # a small call without a gather
    assert d["ms_per_step"] == pytest.approx(sorted(d["config"]["regions_ms"])[1], rel=1e-4)   # `value` is the median region
    assert d["config"]["host_stall_suspected"] is False
    recs = {r["workload"]: r for r in d["also"]}
    assert set(recs) == {"c2", "c2:planned", "c2:planned-device"} and "also" not in roof
    for spec, r in recs.items():
        assert r["bad_columns"] == 0 and r["x_copies"] >= 5 and 0.2 < r["frac"] < 1.0
        assert len(r["regions_ms"]) == 3 and r["ms_per_call"] == pytest.approx(sorted(r["regions_ms"])[1], rel=1e-4)
        assert r["host_stall_suspected"] is False and r["ms_per_call"] < 1.5 * r["kernel_ms"]
        k = "also_" + spec.replace(":", "_").replace("-", "_")
        assert roof[k + "_frac"] == r["frac"] and roof[k + "_kernel_ms"] == r["kernel_ms"]
        assert roof[k + "_ms_per_call"] == r["ms_per_call"] and roof[k + "_parity_err"] == r["parity_err"]
    # the plan-free entry plans for itself (round 5): BASELINE config 2 through rsp_column_sums_device settles on the lean
    # form by its own device-side inspection -- one launch, the reference's bits -- without the caller asking for a plan
    assert recs["c2"]["form"] == "lean" and recs["c2"]["launches"] == 1 and recs["c2"]["planned_by"] == "entry"
    assert recs["c2"]["parity_err"] == 0.0 and recs["c2"]["frac"] > 0.55
    assert recs["c2"]["traffic_in_run"] is True and 0.98 < roof["also_c2_traffic_x"] < 1.15   # re-measured now, not a constant
    assert recs["c2"]["traffic_kernels"] == "lean"
    for k in ("c2:planned", "c2:planned-device"):
        assert recs[k]["form"] == "lean" and recs[k]["launches"] == 1 and recs[k]["planned_by"] == "caller"
        assert recs[k]["parity_err"] == 0.0                                               # the reference's bits
        assert recs[k]["kernel_ms"] < 1.25 * recs["c2"]["kernel_ms"]                      # (the entry's own plan also checks p[]: 4 B per column more)
        assert recs[k]["traffic_in_run"] is False
    # the headline here, one C4 shard, settles on the columns form the same way
    assert roof["kernel"].startswith("colsums_columns_kernel") and d["config"]["shards"][0]["form"] == "columns"
    assert roof["traffic_kernels"] == "columns"
    dev = recs["c2:planned-device"]
    assert dev["plan_by"] == "device" and dev["plan_ms"] < 1.0                            # device time of the inspection kernels
    assert dev["early_general_calls"] >= 0 and recs["c2:planned"]["plan_by"] == "host"


@pytest.mark.parametrize("workload,world,nnz,ncol", [("c4shard", 2, 125_000_000, 125_000),
                                                     ("tiny", 3, 4_000_000, 40_000),
                                                     ("c3", 4, 1_000_000_000, 1_000_000)])
def test_bench_n_ranks_share_the_gpu_with_real_hip_compute(torch_cuda, workload, world, nnz, ncol):
    """`bench.py --gpus N --rendezvous gloo`: bench.py starts the N ranks ITSELF as fresh child processes with the
    environment torch.distributed.run would give them (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*: torchrun's own agent
    would hold a seventh context on a pool that allows six; nothing here re-executes a process that has touched the GPU), every
    rank generates ITS column range of the matrix in HBM at shard.x0 and sums it through
    rsp_column_sums_device, the slices are gathered to rank 0 (as host copies over gloo: RCCL
    refuses two ranks on one device) and rank 0 checks EVERY column of the gathered result against
    the oracle.  What this pins: the rank != 0 control flow of bench.py, the partition, the
    displacements, the max-over-ranks statistics and the N > 1 shape of the JSON line.
    The c3 case is the DEFAULT line of the driver's multi-GPU run (BASELINE config 4) at FOUR ranks -- this pool ends a
    call when a seventh process opens the card; this test process holds a context too, and one slot is left free for
    whatever harness runs the suite (five ranks ran under pytest during the round, six run outside it:
    profiles/r05_rehearsal_lines.jsonl; the 8-rank layout, partition and line are covered without a GPU in
    tests/test_bench_line.py and tests/test_sharded_gloo.py) -- with everything such a line carries: planned_shards,
    direct_gather with four mappers, the Zipf matrix (BASELINE config 5) by the same protocol under both partitions
    (`also_sharded`), and the CPU loop timed on rank 0."""
    torch_cuda.cuda.empty_cache()
    d = _run_bench("--gpus", str(world), "--rendezvous", "gloo", "--try-comm", "--workload", workload,
                   "--steps", "8", "--warmup", "2", "--latency-calls", "5", timeout=1500)
    assert d["_line_bytes"] < 8000
    assert d["n_gpus"] == world and d["steps"] == 8 and d["scaling"] == "strong"
    cfg = d["config"]
    assert cfg["rendezvous"] == "gloo" and cfg["parallelism"] == "rehearsal"
    assert cfg["gather"] == sharded.HostStagedGather.name and cfg["gather_fell_back_to_torch_distributed"] is False
    shards = cfg["shards"]
    assert [s["rank"] for s in shards] == list(range(world))
    assert shards[0]["x0"] == 0 and shards[0]["c0"] == 0
    assert shards[-1]["x1"] == nnz and shards[-1]["c1"] == ncol
    for a, b in zip(shards, shards[1:]):
        assert b["x0"] == a["x1"] > 0 and b["c0"] == a["c1"] > 0      # rank r > 0 starts inside x
    assert all(s["kernel_ms"] > 0 for s in shards)                     # every rank timed its own launches
    if workload == "c4shard":                                          # every rank's plan-free entry settled on the columns form
        assert [s["form"] for s in shards] == ["columns"] * world
    elif workload == "c3":
        # (a quarter of C3 is 2.5e8 entries +- a column: right AT the size up to which columns of ~1000 entries take the
        # columns form, kColumnsTwoWavesMaxNnz -- either side of it is a valid choice, rank by rank)
        assert all(s["form"] in ("columns", "general") for s in shards)
    else:
        assert all(s["form"] in ("general", "lean", "columns") for s in shards)
    per = [s["nnz"] for s in shards]
    assert cfg["shard_imbalance_max_over_mean"] == pytest.approx(max(per) / (sum(per) / world), rel=1e-5)
    assert cfg["shard_imbalance_max_over_mean"] < 1.05
    par = d["parity"]
    assert par["columns_checked"] == "all" and par["ncol"] == ncol
    assert par["columns_out_of_tolerance"] == 0 and par["max_abs_err_over_l1"] <= RTOL
    assert par["empty_columns_exactly_plus_zero"] is True
    assert d["latency_ms_per_call"] > 0 and d["latency"]["calls"] == 5
    assert d["latency"]["ms_median_max_over_ranks"] >= d["latency"]["ms_min"]
    assert d["pipelined"]["value"] > 0
    assert d["value"] == pytest.approx(nnz * 8 / (d["ms_per_step"] * 8e-3), rel=1e-9)
    roof = d["roofline"]
    assert roof["kernel_ms_max_over_ranks"] >= max(s["kernel_ms"] for s in shards) * (1 - 1e-4)
    assert roof["algorithmic_bytes_per_launch"] == 8 * shards[0]["nnz"] + 4 * (shards[0]["c1"] + 1) + 8 * shards[0]["c1"]
    cb = d["cpu_baseline"]                                               # rank 0, at every N, in the same run
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["value"] > 1e8
    # round 4: the two further figures of an N > 1 line, each with its own whole-matrix parity, neither feeding `value`
    ps = d["planned_shards"]
    assert ps["value"] > 0 and len(ps["forms_by_rank"]) == world
    assert ps["bad_columns"] == 0 and ps["parity_err"] <= RTOL
    if workload == "c4shard":
        assert ps["forms_by_rank"] == ["columns"] * world            # columns of ~1000 entries in shards of <= 2.5e8
    elif workload == "c3":
        assert all(f in ("columns", "general kernels", "general") for f in ps["forms_by_rank"])
    dg = d["direct_gather"]
    assert dg["value"] is not None, dg                                # the ranks mapped rank 0's buffer (hipIpc) ...
    assert dg["bad_columns"] == 0 and dg["parity_err"] <= RTOL        # ... and wrote into it
    # round 6: SURVEY section 5's other comparator -- every rank's copy engine writes its slice into ONE page-locked vector in
    # shared memory; whole-matrix parity of what the root then reads from HOST memory; and the flat scalars of all of them
    hg = d["host_gather"]
    assert hg["value"] is not None, hg
    assert hg["bad_columns"] == 0 and hg["parity_err"] <= RTOL
    for name in ("planned_shards", "direct_gather", "host_gather", "pipelined"):
        assert roof[name + "_value"] == d[name]["value"] and roof[name + "_ms_per_step"] == d[name]["ms_per_step"]
    assert cfg["peer_probe"] is None                                  # (the rehearsal shares one device: nothing to probe)
    # --try-comm: the C-ABI communicator's multi-rank bootstrap ran between the rank processes (unique id
    # from rank 0, rsp_comm_init everywhere) and RCCL refused the shared device on EVERY rank, cleanly
    assert cfg["comm_refused_on_ranks"] == world and "comm_rehearsal_unexpected" not in cfg, cfg
    if workload == "c3":
        # BASELINE config 5 inside the default N > 1 line: the Zipf matrix by the protocol of `value`, nnz-balanced and
        # with the naive equal-column-count partition (SURVEY 8e's comparator), every column against the oracle
        a = d["also_sharded"]
        assert set(a) == {"c5_nnz", "c5_cols"}
        for k, rec in a.items():
            assert rec["value"] > 0 and rec["bad_columns"] == 0 and rec["parity_err"] <= RTOL
            assert len(rec["kernel_ms_by_rank"]) == world and sum(rec["nnz_by_rank"]) == nnz
            assert rec["imbalance"] == pytest.approx(max(rec["nnz_by_rank"]) / (nnz / world), rel=1e-5)
            for f in ("value", "ms_per_step", "imbalance", "kernel_ms_max", "gather_ms_max", "parity_err"):
                assert roof[f"also_{k}_{f}"] == rec[f]                # the flat scalars the driver's record keeps
        # no column is split: the nnz-balanced cut is at most one column (<= nrow = 1 % of the entries) over the mean, and
        # never worse than the equal-column-count cut of the same (randomly permuted) Zipf matrix
        assert a["c5_nnz"]["imbalance"] <= 1.0 + world * 0.0101 and a["c5_nnz"]["imbalance"] <= a["c5_cols"]["imbalance"]
    else:
        assert "also_sharded" not in d


@pytest.mark.parametrize("shape", ["uniform", "zipf"])
def test_c4_c5_eight_shards_at_full_size_on_one_gpu(torch_cuda, shape):
    """BASELINE configs 4 (uniform) and 5 (Zipf): 1e7 x 1e6, nnz 1e9, cut into 8 nnz-balanced
    column ranges.  Each shard's x is generated in HBM at shard.x0 (as every rank of bench.py
    does), summed through rsp_column_sums_device with its rebased offsets, and the slices are
    reassembled at the gather layout's displacements.  Checks: the oracle on both edges of every
    shard, identical bits on a second pass, the imbalance figure, and the reassembled result
    against ONE launch over the whole matrix (different chunking, same tolerance)."""
    torch = torch_cuda
    nrow, ncol, nnz, G = 10_000_000, 1_000_000, 1_000_000_000, 8
    if torch.cuda.get_device_properties(0).total_memory < 24 * 2**30:
        pytest.skip("needs >= 24 GB of HBM")
    torch.cuda.empty_cache()
    counts_col = (synth.uniform_counts(ncol, nnz, seed=42, nrow=nrow) if shape == "uniform"
                  else synth.zipf_counts(ncol, nnz, seed=42, nrow=nrow))
    p = synth.offsets_from_counts(counts_col)
    shards = [sharded.make_shard(p, r, G) for r in range(G)]
    counts, displs = sharded.gather_layout(shards[0].bounds)
    assert int(counts.sum()) == ncol and shards[0].x0 == 0 and shards[-1].x1 == nnz
    imb = sharded.imbalance(p, shards[0].bounds)
    assert imb < (1.0001 if shape == "uniform" else 1.02), imb
    # no column is split, and the shards tile x exactly
    for a, b in zip(shards, shards[1:]):
        assert a.c1 == b.c0 and a.x1 == b.x0

    full = torch.empty(ncol, dtype=torch.float64, device="cuda")       # what rank 0 would receive
    for sh in shards:
        xt = torch.empty(sh.nnz, dtype=torch.float64, device="cuda")
        capi.gen_values_device(xt, seed=42, first_idx=sh.x0, kind=0)
        pt = torch.from_numpy(sh.p_local).cuda()
        ws = capi.alloc_workspace(sh.ncol, sh.nnz)
        out = torch.empty(sh.ncol, dtype=torch.float64, device="cuda")
        capi.column_sums_device(xt, pt, out, ws)
        again = torch.empty_like(out)
        capi.column_sums_device(xt, pt, again, ws)
        assert torch.equal(out, again), sh.rank                           # run-to-run bits
        d0 = int(displs[sh.rank])
        full[d0:d0 + sh.ncol].copy_(out)
        got = out.cpu().numpy()
        # the oracle on both edges of the shard (the columns next to the cuts)
        for a, b in ((0, min(300, sh.ncol)), (max(0, sh.ncol - 300), sh.ncol)):
            lo, hi = int(sh.p_local[a]), int(sh.p_local[b])
            xs = oracle.gen_values(hi - lo, 42, sh.x0 + lo, 0)
            pl = (sh.p_local[a:b + 1] - lo).astype(np.int32)
            ref = oracle.column_sums(xs, pl)
            scale = oracle.column_abs_sums(xs, pl)
            assert np.all(np.abs(got[a:b] - ref) <= RTOL * scale), (shape, sh.rank, a)
        empty = np.diff(sh.p_local) == 0
        assert np.all(got[empty] == 0.0) and not np.any(np.signbit(got[empty]))
        del xt, out, again, ws, pt

    # one launch over the whole matrix: every column of the reassembled result agrees with it
    xt = torch.empty(nnz, dtype=torch.float64, device="cuda")
    capi.gen_values_device(xt, seed=42, kind=0)
    pt = torch.from_numpy(p).cuda()
    whole = capi.column_sums_device(xt, pt)
    l1 = capi.column_reduce_device(xt, pt, capi.OP_SUM_ABS)
    assert bool(torch.all((whole - full).abs() <= 2 * RTOL * l1))
    assert [int(d) for d in displs] == [sh.c0 for sh in shards]     # slices land at their columns


@pytest.mark.parametrize("shape", ["uniform", "zipf"])
def test_c4_c5_multi_device_handle_at_full_size(torch_cuda, shape):
    """BASELINE configs 4 and 5 through the form a multi-GPU HANDLE actually runs: the 1e9-entry
    matrices uploaded once as 8 resident shards (MultiDeviceCSC, rsp_mcsc_*; all on this box's one device), every shard
    inspected at upload.  The uniform matrix's shards (1.25e8 entries in columns of ~1000) must have taken the COLUMNS
    form -- two wavefronts per column, one launch -- which until now was only tested on small shapes; the Zipf shards
    keep the general kernels.  Checks: the shards tile the columns, the oracle on both edges of every shard, every
    column against the plan-free device entry, identical bits on a second call."""
    torch = torch_cuda
    nrow, ncol, nnz, G = 10_000_000, 1_000_000, 1_000_000_000, 8
    if torch.cuda.get_device_properties(0).total_memory < 40 * 2**30:
        pytest.skip("needs >= 40 GB of HBM")
    try:
        free_host = os.sysconf("SC_AVPHYS_PAGES") * os.sysconf("SC_PAGE_SIZE")
    except (ValueError, OSError):
        free_host = 0
    if free_host < 14 * 2**30:
        pytest.skip("needs >= 14 GB of free host memory for x")
    torch.cuda.empty_cache()
    counts_col = (synth.uniform_counts(ncol, nnz, seed=42, nrow=nrow) if shape == "uniform"
                  else synth.zipf_counts(ncol, nnz, seed=42, nrow=nrow))
    p = synth.offsets_from_counts(counts_col)
    x = oracle.gen_values_threads(nnz, 42, 0, 0, max(1, min(16, len(os.sched_getaffinity(0)))))
    h = capi.MultiDeviceCSC(x, p, (nrow, ncol), devices=[0] * G)
    try:
        assert h.dims() == (nrow, ncol, G)
        info = [h.shard_info(k) for k in range(G)]
        assert info[0]["c0"] == 0 and info[-1]["c1"] == ncol
        assert all(a["c1"] == b["c0"] for a, b in zip(info, info[1:]))
        assert sum(s["nnz"] for s in info) == nnz
        assert [s["c0"] for s in info] + [ncol] == [int(b) for b in capi.partition_columns(p, G)]
        if shape == "uniform":
            assert [s["form"] for s in info] == ["columns"] * G, info
        else:
            assert all(s["form"] in ("general kernels", "snapped") for s in info), info
        got = h.column_sums()
        assert got.tobytes() == h.column_sums().tobytes()                       # run-to-run bits
        for s in info:                                                          # the oracle next to every cut
            for a, b in ((s["c0"], min(s["c0"] + 300, s["c1"])), (max(s["c0"], s["c1"] - 300), s["c1"])):
                lo, hi = int(p[a]), int(p[b])
                pl = (p[a:b + 1] - lo).astype(np.int32)
                ref = oracle.column_sums(x[lo:hi], pl)
                scale = oracle.column_abs_sums(x[lo:hi], pl)
                assert np.all(np.abs(got[a:b] - ref) <= RTOL * scale), (shape, s, a)
        empty = np.diff(p) == 0
        assert np.all(got[empty] == 0.0) and not np.any(np.signbit(got[empty]))
        means = h.column_means()
        assert means.tobytes() == (got / nrow).tobytes()
    finally:
        h.close()
    # every column against one plan-free launch over the whole matrix
    xt = torch.empty(nnz, dtype=torch.float64, device="cuda")
    capi.gen_values_device(xt, seed=42, kind=0)
    pt = torch.from_numpy(p).cuda()
    whole = capi.column_sums_device(xt, pt)
    l1 = capi.column_reduce_device(xt, pt, capi.OP_SUM_ABS)
    assert bool(torch.all((whole - torch.from_numpy(got).cuda()).abs() <= 2 * RTOL * l1))


@pytest.mark.parametrize("world,workload,nrow,nnz", [(1, "tiny", 200_000, 4_000_000), (2, "tiny", 200_000, 4_000_000),
                                                     (3, "m10_3e7", 3_000_000, 30_000_000)])
def test_bench_rowsums_over_column_range_shards(torch_cuda, world, workload, nrow, nnz):
    """`bench.py --op rowsums` (missing 5): every rank sums the rows of ITS column range
    (rsp_row_sums_device on its x / i slices), the partial vectors of nrow doubles are reduced in rank order to rank 0
    (here over gloo: the ranks share the GPU; on the driver's node rsp_comm_reduce_rows over RCCL) and rank 0 checks every
    row against the oracle's scatter loop over the whole matrix."""
    torch_cuda.cuda.empty_cache()
    flags = ["--op", "rowsums", "--workload", workload, "--steps", "4", "--warmup", "1"]
    if world > 1:
        flags += ["--gpus", str(world), "--rendezvous", "gloo"]
    d = _run_bench(*flags)
    assert d["metric"].startswith("rowSums nnz/s") and d["config"]["op"] == "rowsums"
    assert d["n_gpus"] == world and d["steps"] == 4 and d["unit"] == "nnz/s" and d["dtype"] == "f64"
    assert d["value"] == pytest.approx(nnz * 4 / (d["ms_per_step"] * 4e-3), rel=1e-9)
    par = d["parity"]
    assert par["rows_checked"] == "all" and par["nrow"] == nrow and par["rows_out_of_tolerance"] == 0
    assert par["max_abs_err_over_l1"] <= RTOL
    shards = d["config"]["shards"]
    assert len(shards) == world and sum(s["nnz"] for s in shards) == nnz and all(s["kernel_ms"] > 0 for s in shards)
    roof = d["roofline"]
    assert roof["algorithmic_bytes_per_launch"] == 12 * shards[0]["nnz"] + 8 * nrow and 0 < roof["frac"] < 1
    if world > 1:
        assert d["config"]["reduce"] == sharded.GlooReduceRows.name and d["config"]["parallelism"] == "rehearsal"
        assert roof["reduce_ms"] > 0 and roof["reduce_bytes_per_rank"] == 8 * nrow
    else:
        assert d["config"]["reduce"] is None and roof["reduce_ms"] is None


def test_direct_write_gather_between_two_processes_sharing_the_gpu(torch_cuda, tmp_path):
    """The pieces of the direct-write gather on their own (rsp_shared_result_* + rsp_host_barrier_*): a child process maps
    this process's result buffer through its IPC handle and its column-sum kernels write their slice straight into it; a
    barrier between the two processes orders the parent's read behind the child's kernels."""
    torch = torch_cuda
    ncol, split = 30_000, 17_123
    counts = synth.uniform_counts(ncol, 2_000_000, seed=7, nrow=None)
    p = synth.offsets_from_counts(counts)
    x = synth.gen_values(int(p[-1]), seed=7, kind=0)
    shared = capi.SharedResult(ncol)
    name = f"/rsp_test_{os.getpid()}"
    child = tmp_path / "child.py"
    child.write_text(f"""
import os, sys
sys.path.insert(0, {ROOT!r})
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import numpy as np, torch
from rcppsparse_amd import capi, synth
capi.load()
handle = bytes.fromhex(sys.argv[1])
bar = capi.HostBarrier({name!r}, 2, 1)
view = capi.SharedResult({ncol}, handle=handle)
p = synth.offsets_from_counts(synth.uniform_counts({ncol}, 2_000_000, seed=7, nrow=None))
x = synth.gen_values(int(p[-1]), seed=7, kind=0)
c0 = {split}
pl = (p[c0:] - p[c0]).astype(np.int32)
xt, pt = torch.from_numpy(x[p[c0]:]).cuda(), torch.from_numpy(pl).cuda()
class Out:
    def data_ptr(self): return view.ptr + 8 * c0
ws = capi.alloc_workspace(len(pl) - 1, xt.numel())
capi.prepared_column_sums(xt, pt, Out(), ws)()
torch.cuda.synchronize()
bar.wait()          # the parent may read now
bar.wait()          # ... and has read
view.close(); bar.close()
""")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    proc = subprocess.Popen([sys.executable, str(child), shared.handle.hex()], env=env, stdout=subprocess.PIPE,
                            stderr=subprocess.PIPE, text=True)
    try:
        bar = capi.HostBarrier(name, 2, 0)
        # this process sums the first columns into the same buffer meanwhile
        pl = p[:split + 1].copy()
        xt, pt = torch.from_numpy(x[:p[split]]).cuda(), torch.from_numpy(pl).cuda()

        class Out:
            def data_ptr(self):
                return shared.ptr
        ws = capi.alloc_workspace(split, xt.numel())
        capi.prepared_column_sums(xt, pt, Out(), ws)()
        torch.cuda.synchronize()
        bar.wait(timeout=120.0)
        got = shared.read()
        bar.wait(timeout=120.0)
        bar.close()
    finally:
        out, err = proc.communicate(timeout=120)
    assert proc.returncode == 0, err[-2000:]
    ref = oracle.column_sums(x, p)
    assert np.all(np.abs(got - ref) <= RTOL * oracle.column_abs_sums(x, p))
    shared.close()


# ------------------------------------------------------------------ rowSums over column-range shards
def test_row_sums_eight_shards_reduce_in_rank_order(torch_cuda):
    """Multi-GPU rowSums (the collective is a reduce of f64[nrow]) rehearsed on one GPU at
    1e8 entries: 1e6 x 1e5 cut into 8 nnz-balanced column ranges; every shard's partial row sums come from
    rsp_row_sums_device on its x / i slices (as a rank would compute them), the 8 vectors are added in rank
    order by the kernel rsp_comm_reduce_rows uses (rsp_add_partials_device).  Against the oracle's scatter loop
    over the WHOLE matrix with 1e-12 * sum|x| per row; bit-identical to adding the partial vectors one after
    the other; and a one-rank communicator runs rsp_comm_reduce_rows itself (add kernel + gatherv, no peers)."""
    torch = torch_cuda
    nrow, ncol, nnz, G = 1_000_000, 100_000, 100_000_000, 8
    p = synth.offsets_from_counts(synth.uniform_counts(ncol, nnz, seed=42, nrow=nrow))
    pt = torch.from_numpy(p).cuda()
    xt = torch.empty(nnz, dtype=torch.float64, device="cuda")
    it = torch.empty(nnz, dtype=torch.int32, device="cuda")
    capi.gen_values_device(xt, 42, 0, 0)
    capi.gen_row_indices_device(it, pt, nrow, 42)
    shards = [sharded.make_shard(p, r, G) for r in range(G)]
    parts = torch.empty((G, nrow), dtype=torch.float64, device="cuda")
    for sh in shards:
        capi.row_sums_device(xt[sh.x0:sh.x1], it[sh.x0:sh.x1], nrow, parts[sh.rank])
    total = capi.add_partials_device(parts)
    # the same sum, one vector after the other (torch adds elementwise with one rounding per add)
    seq = parts[0].clone()
    for k in range(1, G):
        seq += parts[k]
    assert torch.equal(total, seq + 0.0)
    got = total.cpu().numpy()
    x = oracle.gen_values(nnz, 42, 0, 0)
    i = oracle.gen_row_indices(p, nrow, 42)
    ref = oracle.row_sums(x, i, p, nrow)
    scale = np.bincount(i, weights=np.abs(x), minlength=nrow)
    assert np.all(np.abs(got - ref) <= RTOL * scale), float(np.max(np.abs(got - ref) / np.maximum(scale, 1e-300)))
    means = capi.add_partials_device(parts, ncol_for_means=ncol).cpu().numpy()
    assert means.tobytes() == (got / ncol).tobytes()
    # the collective itself with one rank: no peers, so its add step and its gatherv to the root's result
    comm = capi.Comm(capi.comm_unique_id(), 1, 0, 0)
    try:
        res = torch.full((nrow,), -1.0, dtype=torch.float64, device="cuda")
        comm.reduce_rows(parts[3], res, root=0)
        torch.cuda.synchronize()
        assert torch.equal(res, parts[3] + 0.0)
    finally:
        comm.close()


def test_bench_child_process_planned_c2(torch_cuda):
    """`bench.py --workload c2 --planned`: BASELINE config 2 through the inspector-executor plan.  The plan takes the
    lean form (every column of C2 is short), its cost is reported beside the calls (`config.planned.plan_ms`), the
    kernel of the line is the one-launch lean kernel, and because a lean call adds every column in the reference's
    order the whole-matrix parity figure is exactly zero."""
    torch_cuda.cuda.empty_cache()
    d = _run_bench("--workload", "c2", "--planned", "--steps", "50", "--warmup", "5", "--no-cpu-baseline",
                   "--latency-calls", "3")
    pl = d["config"]["planned"]
    assert pl["form"] == "lean" and pl["snapped"] is True and pl["plan_ms"] > 0 and pl["chunks"] > 10_000
    assert d["roofline"]["kernel"].startswith("colsums_lean_kernel")
    assert d["roofline"]["kernel_timing"] == "region" and len(d["config"]["regions_ms"]) == 3
    par = d["parity"]
    assert par["columns_checked"] == "all" and par["ncol"] == 1_000_000 and par["columns_out_of_tolerance"] == 0
    assert par["max_abs_err_over_l1"] == 0.0
    assert d["config"]["x_copies_rotated"] >= 5                      # every call reads from HBM, not the Infinity Cache
    assert 0.3 < d["roofline"]["frac"] < 1.0


def test_bench_child_process_planned_vignette(torch_cuda):
    """`bench.py --workload vignette --planned`: the shape of the reference's own benchmark (100000 x 1000, 1000 columns
    of ~1e4 entries, Documentation.Rmd:425) through the plan's columns form -- one workgroup per column, one launch, no
    records -- with every column of the result checked against the oracle."""
    torch_cuda.cuda.empty_cache()
    d = _run_bench("--workload", "vignette", "--planned", "--steps", "50", "--warmup", "5", "--no-cpu-baseline",
                   "--latency-calls", "3")
    pl = d["config"]["planned"]
    assert pl["form"] == "columns" and pl["snapped"] is True and pl["chunks"] == 1000
    assert d["roofline"]["kernel"].startswith("colsums_columns_kernel")
    par = d["parity"]
    assert par["columns_checked"] == "all" and par["ncol"] == 1000 and par["columns_out_of_tolerance"] == 0
    assert par["max_abs_err_over_l1"] <= 1e-12
    assert 0.3 < d["roofline"]["frac"] < 1.0


def test_peer_probe_child_on_one_device_says_not_applicable(torch_cuda):
    """bench.py's probe before direct_gather between different devices, on the one device this box
    has: the child process starts, sees owner == writer and reports 'same device' -- the branch a 1-GPU box can reach.  The
    writer grandchild is exercised too, against a buffer this process exports (two processes, one device: the IPC path that
    test_direct_write_gather_between_two_processes_sharing_the_gpu covers end to end)."""
    import bench
    assert bench.run_peer_probe(0, 0) == "same device"
    assert capi.device_can_access_peer(0, 0) is True
    shared = capi.SharedResult(bench.PROBE_N)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--peer-probe-writer", shared.handle.hex(), "0"], cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert np.array_equal(shared.read(), np.arange(1, bench.PROBE_N + 1, dtype=np.float64))
    shared.close()


@pytest.mark.parametrize("gpus,workload", [(1, "tinyu"), (4, "tinyu"), (8, "c4shard")])
def test_bench_parallelism_threads_line(torch_cuda, gpus, workload):
    """`bench.py --parallelism threads`: ONE process drives N shards through the resident multi-GPU handle (rsp_mcsc_*) -- the
    multi-GPU path an R session reaches.  Same line, whole-matrix parity, every gather / launch
    combination as a flat scalar; on this box every shard shares the one device (config.devices_distinct false)."""
    torch_cuda.cuda.empty_cache()
    d = _run_bench("--parallelism", "threads", "--gpus", str(gpus), "--workload", workload, "--steps", "6", "--warmup", "2")
    assert d["_line_bytes"] < 8000
    cfg, roof = d["config"], d["roofline"]
    assert d["n_gpus"] == gpus and cfg["parallelism"] == "threads" and cfg["devices_distinct"] is (gpus == 1)
    assert cfg["launch"] == ("workers" if gpus >= 3 else "serial") and cfg["gather"] == ("blit" if gpus == 1 else "d2h")
    assert len(cfg["shards"]) == gpus and cfg["shards"][0]["c0"] == 0 and all(s["device"] == 0 for s in cfg["shards"])
    assert cfg["rccl_version"] > 20000 and "librccl" in cfg["rccl_library"]
    par = d["parity"]
    assert par["columns_out_of_tolerance"] == 0 and par["max_abs_err_over_l1"] <= RTOL and par["columns_checked"] == "all"
    assert len(cfg["regions_ms"]) == 3 and d["ms_per_step"] == pytest.approx(sorted(cfg["regions_ms"])[1], rel=1e-5)
    nnz = sum(s["nnz"] for s in cfg["shards"])
    assert d["value"] == pytest.approx(nnz / (d["ms_per_step"] * 1e-3), rel=1e-9)
    g0 = cfg["gather"]
    for k in ("threads_d2h_pinned_ms", "threads_blit_pinned_ms", f"threads_{g0}_pageable_ms", "threads_stores_pinned_ms",
              "threads_none_pinned_ms", "threads_last_enqueue_us", "call_minus_slowest_kernel_us"):
        assert isinstance(roof[k], float), k
    assert not any(k.endswith("_bits_differ") for k in roof)          # every gather returned the bits of `value`'s run
    if gpus == 1:
        assert roof["threads_rccl_pinned_ms"] > 0                     # ncclCommInitAll over one device, one D2H
    else:
        assert "threads_rccl_pinned_ms" not in roof                   # (RCCL refuses two ranks on one device: not tried)
        assert roof[f"threads_serial_{g0}_pinned_ms" if cfg["launch"] == "workers" else f"threads_workers_{g0}_pinned_ms"] > 0
    if workload == "c4shard":
        assert all(s["form"] == "columns" for s in cfg["shards"]) and 0.2 < roof["frac"] < 1.0


@pytest.mark.parametrize("gpus", [1, 3])
def test_bench_rowsums_through_the_single_process_handle(torch_cuda, gpus):
    """`bench.py --op rowsums --parallelism threads`: Matrix::rowSums (reference RcppSparse.h:138-144) through rsp_mcsc_row_sums --
    every shard's partial vector added in shard order ON THE DEVICES (round 6), the result in a host vector; every row against
    the oracle's scatter loop over the whole matrix."""
    torch_cuda.cuda.empty_cache()
    d = _run_bench("--op", "rowsums", "--parallelism", "threads", "--gpus", str(gpus), "--workload", "tinyu", "--steps", "4",
                   "--warmup", "1")
    assert d["metric"].startswith("rowSums") and d["n_gpus"] == gpus and d["config"]["parallelism"] == "threads"
    assert d["config"]["op"] == "rowsums" and len(d["config"]["shards"]) == gpus and len(d["config"]["regions_ms"]) == 3
    par = d["parity"]
    assert par["rows_checked"] == "all" and par["rows_out_of_tolerance"] == 0 and par["max_abs_err_over_l1"] <= RTOL
    assert d["value"] == pytest.approx(4_000_000 / (d["ms_per_step"] * 1e-3), rel=1e-9)
