#!/usr/bin/env python3
"""bench.py -- columnSums over a 1e9-nnz CSC dgCMatrix on N MI355X GPUs.

Contract (one JSON line from rank 0):
    python bench.py --gpus N --steps K --warmup W
    N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
               --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one columnSums call over the whole matrix: the hot path of reference
src/example.cpp:26-32, here `rsp_column_sums_device` (hand-written HIP, called
through the C ABI of include/rcppsparse_hip.h) on inputs already resident in HBM.
Workload (BASELINE.json configs[2], the one the metric is quoted on): 1e7 x 1e6,
nnz = 1e9, uniform ("rsparsematrix-like") -- synthetic, seeded, generated in HBM.
With N > 1 the SAME matrix is split into N nnz-balanced contiguous column ranges
(strong scaling: total work fixed), each rank sums its range, and the per-rank
slices are gathered to rank 0 with an RCCL gatherv over xGMI inside every step.

THE SAME PROTOCOL AT EVERY N (so per-N values can be divided by each other):
`value` = nnz of the whole matrix x K / wall time of K calls issued back to back, each call
in order on ONE stream per rank: column-sum kernels, then (N > 1) the gatherv of that call
-- the synchronous-call semantics of the reference function; call k+1 starts when call k,
its gather included, is done.  Max over ranks, barrier + synchronize on both sides.

The line is COMPACT (under 8000 bytes at every N, tests/test_bench_line.py): numbers and short
codes only.  `python bench.py --explain` prints what every key means (the prose that earlier
rounds carried inside the line); `--verbose` puts that glossary into the line as `notes`.
Everything a reader of the driver's record needs sits as FLAT SCALARS inside `roofline`
(the record keeps the scalars of `roofline`, `config` and `cpu_baseline`, other keys by name only).
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# Multi-process GPU work on this pool's hosts: the kernel driver offers dmabuf IPC only, and without
# this setting RCCL's exchange of device buffers between the ranks' processes fails with
# "hipIpcGetMemHandle: invalid argument".  It has to be in the environment before the HIP runtime
# starts, i.e. before torch is imported.  An explicit setting wins.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
# Nothing this file measures may come from the Rcpp layer's host loop (columnsums_impl.hpp answers on the CPU when
# a machine has no GPU or the matrix is tiny): a GPU is required in this process and in every child it starts.
os.environ["RCPPSPARSE_REQUIRE_GPU"] = "1"

HBM_PEAK_GBPS = 8000.0     # MI355X spec peak (MI355X_MICROARCH.md: 8.0 TB/s)
SEED = 42
LINE_BYTES_MAX = 8000      # the driver keeps the last 8 KiB of stdout

WORKLOADS = {
    # name: (nrow, ncol, nnz, shape)
    "c2": (1_000_000, 1_000_000, 10_000_000, "uniform"),
    "c3": (10_000_000, 1_000_000, 1_000_000_000, "uniform"),
    "c5": (10_000_000, 1_000_000, 1_000_000_000, "zipf"),
    "c5desc": (10_000_000, 1_000_000, 1_000_000_000, "zipf-descending"),   # worst-case column order
    # the shape of the reference's own vignette benchmark: rsparsematrix(100000, 1000, 0.1)
    # (vignettes/Documentation.Rmd:425), 1e7 nnz in 1000 columns of ~1e4
    "vignette": (100_000, 1_000, 10_000_000, "uniform"),
    # experiments (not BASELINE configs): one shard of C4, and a single 1e9-long column
    # (no column ends inside the stream: the ceiling of the streaming fast path)
    "c4shard": (10_000_000, 125_000, 125_000_000, "uniform"),
    "stream": (2_000_000_000, 1, 1_000_000_000, "uniform"),
    "fewcols": (10_000_000, 10_000, 1_000_000_000, "uniform"),   # 1e4 columns of 1e5 entries (row-restricted sums: profiles/r04_masked.md)
    # column-length regimes between C2 (10 per column) and C3 (1000 per column), all 1e9 nnz
    "m30": (10_000_000, 33_000_000, 1_000_000_000, "uniform"),
    "m100": (10_000_000, 10_000_000, 1_000_000_000, "uniform"),
    "m300": (10_000_000, 3_300_000, 1_000_000_000, "uniform"),
    "m10": (10_000_000, 100_000_000, 1_000_000_000, "uniform"),
    # ~10 per column at sizes between C2 and 1e9 (the lean planned form: profiles/r03_lean_sizes.jsonl)
    "m10_3e7": (3_000_000, 3_000_000, 30_000_000, "uniform"),
    "m10_1e8": (10_000_000, 10_000_000, 100_000_000, "uniform"),
    # small shapes for the -m gpu tests that run this file as a child process
    "tiny": (200_000, 40_000, 4_000_000, "zipf"),
    "tinyu": (200_000, 40_000, 4_000_000, "uniform"),
}

# What the keys of the line mean (`--explain`; `--verbose` adds it to the line as `notes`).
GLOSSARY = {
    "value": "nnz of the whole matrix x K / wall time of K calls issued back to back; each call in order on ONE stream per "
             "rank: column-sum kernels, then (N > 1) that call's gatherv; max over ranks; barrier + synchronize on both sides. "
             "Small single-GPU workloads (< 2e8 nnz, no gather): THREE such regions, `value` is the median one, "
             "`config.regions_ms` lists ms per call of all three",
    "config.parallelism": "single | ranges+rccl (N nnz-balanced contiguous column ranges + RCCL gatherv to rank 0) | "
                          "rehearsal (--rendezvous gloo: ranks share the box's devices, slices travel as host copies over gloo "
                          "because RCCL refuses two ranks on one device; a rehearsal of the N > 1 control flow with real HIP "
                          "compute, never a multi-GPU number)",
    "config.shards": "what every rank owned and measured, rank order: columns [c0, c1), entries, mean kernel / gather ms",
    "config.host_stall_suspected": "ms_per_step > 1.5 x roofline.kernel_ms on a call without a gather: the host, not the device, "
                                   "set the pace of the region",
    "latency_ms_per_call": "one call at a time: barrier, then host launch -> kernels -> gatherv -> stream synchronize on "
                           "rank 0 (the gathered result is complete and the host has seen it); median",
    "pipelined": "calls overlapped across steps (launches alternate over the compute streams, the gather of call k runs on "
                 "its own stream beside the kernel of call k+1); NOT the protocol of `value`",
    "planned_shards": "N > 1: the protocol of `value` with every rank's launches going through an inspector-executor plan "
                      "of its own shard (inspected once, outside the timed region); NOT `value`",
    "direct_gather": "N > 1: every rank's kernels store into rank 0's result buffer (mapped with hipIpcOpenMemHandle) at the "
                     "rank's displacement; per call: launch, wait for the own stream, cross a shared-memory barrier; "
                     "NOT the protocol of `value`",
    "host_gather": "N > 1: every rank copies its slice (hipMemcpyAsync behind its kernels) over its own host link into ONE page-locked "
                   "vector in POSIX shared memory that all rank processes map; per call: launch, the copy, wait for the own stream, "
                   "cross a shared-memory barrier -- the root then holds the whole vector in HOST memory; NOT the protocol of `value`",
    "config.peer_probe": "what the throw-away child processes saw before direct_gather was allowed between different devices: "
                         "passed | same device | no peer access | failed: ...",
    "roofline.achieved": "algorithmic bytes of one launch (8 B/nnz + 12 B/column, SURVEY.md 8d; i[] is never read) of rank 0's "
                         "shard / roofline.kernel_ms",
    "roofline.kernel_ms": "mean device time of the kernels of one call, HIP events on the launch stream.  kernel_timing = "
                          "per_call: an event pair around the kernels of each timed call; region: ONE pair around each timed "
                          "region / K (small calls: a pair per call would idle the queue); per_call_after: N > 1, the timed "
                          "region carries no events, the K calls are issued once more right after it with events around "
                          "kernels and gather",
    "roofline.traffic": "HBM bytes of one call: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes, FETCH_SIZE "
                        "doubled (MI355X_MICROARCH.md), main kernel + fix-up.  traffic_measured_in_run true: two child passes "
                        "of this workload on this device after the timed regions of THIS run; false: the figure of the "
                        "committed passes under profiles/ (roofline.traffic_file)",
    "roofline.read_ceiling_GBps": "a read-only kernel with the access shape of the column-sum kernel (same chunk grid, same "
                                  "1 KiB nt loads, same register pipeline; no p[], no stores) over the SAME x on the SAME "
                                  "device in the SAME run; frac_of_ceiling = achieved / read_ceiling_GBps",
    "roofline.also_*": "more single-GPU workloads measured after the headline one, outside its timed region, by the same "
                       "protocol, each with whole-matrix parity; never part of `value`.  <w>_frac = achieved / 8000 GB/s, "
                       "<w>_kernel_ms, <w>_ms_per_call (median of three regions), <w>_parity_err = max |gpu - ref| / sum|x| "
                       "over all columns, <w>_traffic_x = HBM bytes / algorithmic bytes measured in this run, "
                       "<w>_traffic_recorded_x = the same from the committed passes under profiles/",
    "config.auto_plan": "1 (default): the plan-free entry plans for itself; 0: RSP_AUTO_PLAN=0 in the environment -- general kernels only",
    "config.shards[].form": "the form rank r's calls took: rsp_column_sums_device plans for itself (include/rcppsparse_hip.h) -- "
                            "general kernels on the first calls, then lean / columns where the device-side inspection of p[] "
                            "selects them; the timed regions start after every rank has settled (--planned: the caller's plan)",
    "also": "the same records whole: form = general | snapped | lean | columns; planned_by = entry (the plan-free entry's own "
            "plan) | caller (rsp_column_sums_plan_*); launches per call; plan_ms (plan_by = host: "
            "rsp_column_sums_plan_create on a host copy of p[]; device: rsp_column_sums_plan_create_device, device time of "
            "the inspection kernels); early_general_calls = calls answered by the general kernels before a device-made plan "
            "was known",
    "also_sharded": "N > 1 (default c3 line): the protocol of `value` run again, outside its timed region, on c5 (Zipf "
                    "nnz/column) with the nnz-balanced partition and with the naive equal-column-count partition "
                    "(SURVEY.md 8e's comparator); per-rank kernel ms, imbalance = max / mean entries per shard, "
                    "whole-matrix parity",
    "parity": "EVERY column of the gathered result against the oracle (1-thread C restatement of the reference loop), after "
              "the timed region: |gpu - ref| <= 1e-12 x sum|x| per column; empty columns exactly +0.0",
    "config.parallelism=threads": "ONE process, N shards through rsp_mcsc_column_sums (the resident multi-GPU handle: per shard a stream, "
                                  "an output, a plan; per handle one page-locked result vector and parked worker threads; nothing is "
                                  "created per call).  `value` = K synchronous calls back to back, the handle's default gather (config.gather: "
                                  "blit = a copy kernel per slice with a device per shard, d2h = the runtime's copy command where "
                                  "shards share a device; either way every slice over its own host link), result left in the page-locked vector; THREE such regions, the "
                                  "median one is `value` (config.regions_ms lists all three).  roofline.threads_<gather>_"
                                  "<dest>_ms: ms per call of the other combinations -- gather d2h | blit | stores | rccl (ncclCommInitAll, "
                                  "grouped send / recv to shard 0's device, one D2H; needs a device per shard) | none (slices stay on "
                                  "the devices: launch + wait only); dest pinned (the handle's vector) | pageable (a malloc'ed vector: "
                                  "what an R NumericVector is).  threads_last_enqueue_us: host clock from a call's entry to the return "
                                  "of the last shard's launch; config.devices_distinct false = a rehearsal on fewer cards than shards",
    "roofline.mcsc8_*": "default N = 1 line: what the HOST adds to the single-process multi-GPU call (rsp_mcsc_column_sums: the path an R "
                        "session reaches) -- eight shards of 131072 entries on this device (their kernels take microseconds: what is "
                        "left of a call IS the host); <launch>_call_us = median wall time of a call, kernels_us = the shards' kernel "
                        "times added up (each timed alone), <launch>_overhead_us = the difference, <launch>_last_enqueue_us = host clock "
                        "until the last shard's launch was issued; launch serial | workers",
    "roofline.threads_*": "--parallelism ranks at N > 1: the figures of a `--parallelism threads` CHILD run over the same devices after "
                          "the ranks have finished (threads_value = its `value`); threads_error instead if the child failed",
    "cpu_baseline": "the oracle (kind port: restatement of reference src/example.cpp:26-32), 1 thread, on this box's host on "
                    "a bounded prefix of the same matrix, rank 0, at every N; all_cores_value: the same loop under an OpenMP "
                    "parallel-for over the columns (NOT the reference's behaviour)",
}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--nnz", type=int, default=0, help="override nnz (experiments)")
    ap.add_argument("--kind", type=int, default=0, help="0 signed two-decimal, 1 U(0,1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--chunk-rows", type=int, default=0)
    ap.add_argument("--partition", default="nnz", choices=["nnz", "cols"],
                    help="N>1: nnz-balanced column ranges (default) or the naive equal-column-count split")
    ap.add_argument("--gather-buffers", type=int, default=4,
                    help="pipelined figure: per-shard output buffers in the kernel/gather pipeline")
    ap.add_argument("--compute-streams", type=int, default=2,
                    help="pipelined figure: streams the column-sum launches alternate over")
    ap.add_argument("--latency-calls", type=int, default=25,
                    help="calls timed one at a time for latency_ms_per_call")
    ap.add_argument("--no-pipelined", action="store_true", help="skip the separate pipelined figure")
    ap.add_argument("--rendezvous", default="nccl", choices=["nccl", "gloo"],
                    help="nccl (default): one rank per GPU, torch.distributed over RCCL, the gatherv on the "
                         "C-ABI RCCL communicator.  gloo: REHEARSAL of the N > 1 control flow on a box with "
                         "fewer GPUs than ranks -- ranks share devices (rank r on device r %% device_count), "
                         "rendezvous and barriers over gloo, and because RCCL refuses two ranks on one device "
                         "the slices travel as host copies over gloo.  Its numbers are not multi-GPU numbers.")
    ap.add_argument("--planned", action="store_true",
                    help="inspector-executor form: the offsets are inspected once (rsp_column_sums_plan_create, "
                         "reported as plan_ms, outside every timed region) and every call is "
                         "rsp_column_sums_planned_device.  The headline C3 `value` stays plan-free.")
    ap.add_argument("--no-lean", action="store_true", help="--planned: keep the plan out of the lean form (A/B)")
    ap.add_argument("--try-comm", action="store_true",
                    help="--rendezvous gloo only: also take the C-ABI communicator through its multi-rank "
                         "bootstrap (unique id from rank 0, rsp_comm_init on every rank).  With ranks sharing a "
                         "device RCCL must refuse it; config.comm_refused_on_ranks counts the refusals")
    ap.add_argument("--op", default="colsums", choices=["colsums", "rowsums"],
                    help="colsums (default): the headline path.  rowsums: Matrix::rowSums (reference RcppSparse.h:138-144) over "
                         "the same column-range shards -- every rank sums the rows of ITS columns (rsp_row_sums_device) and "
                         "the partial vectors of nrow doubles are reduced in rank order to rank 0.  Its line carries metric "
                         "'rowSums nnz/s ...': a next-row figure, never the headline")
    ap.add_argument("--no-planned-shards", action="store_true", help="N > 1: skip the separate planned_shards figure")
    ap.add_argument("--direct-gather", default="auto", choices=["auto", "on", "off"],
                    help="N > 1: the separate direct_gather figure (the ranks' kernels store into rank 0's result buffer, mapped "
                         "with hipIpc).  auto = in the --rendezvous gloo rehearsal always; between DIFFERENT devices only after a "
                         "PROBE has passed: rank 0 starts a child process (and that a second one) which does exactly this between "
                         "rank 0's and rank 1's device -- hipDeviceCanAccessPeer, an IPC-mapped buffer, one kernel storing through "
                         "the mapping from the other process, the values read back -- so that a mapping the driver does not honour "
                         "takes a throw-away process down, not the job (config.peer_probe says what it saw)")
    ap.add_argument("--host-gather", default="auto", choices=["auto", "on", "off"],
                    help="N > 1: the separate host_gather figure (SURVEY.md section 5's other comparator: every rank copies its slice "
                         "over its own host link into ONE page-locked vector in shared memory; per call: kernels, the copy, a wait "
                         "for the own stream, a host barrier).  auto = on")
    ap.add_argument("--peer-probe", nargs=2, type=int, metavar=("OWNER", "WRITER"), help=argparse.SUPPRESS)
    ap.add_argument("--peer-probe-writer", nargs=2, metavar=("HANDLE_HEX", "WRITER"), help=argparse.SUPPRESS)
    ap.add_argument("--also", default="auto",
                    help="N = 1: more single-GPU workloads measured after the headline one, OUTSIDE its timed region "
                         "(never part of `value`).  auto = every other single-GPU BASELINE configuration and its planned form "
                         "when the headline workload is c3, nothing otherwise; or a comma-separated list of workload[:planned]")
    ap.add_argument("--no-also", action="store_true", help="skip the `also` records (profiling runs)")
    ap.add_argument("--also-sharded", default="auto",
                    help="N > 1: more sharded workloads by the protocol of `value`, after the headline one.  auto = c5:nnz,c5:cols "
                         "when the headline is the default c3 line; or a comma-separated list of workload:partition; none = skip")
    ap.add_argument("--traffic-pass", default="auto", choices=["auto", "on", "off"],
                    help="roofline.traffic measured IN THIS RUN: two child runs of this file under rocprofv3 (--pmc FETCH_SIZE, "
                         "--pmc WRITE_SIZE, separate passes as MI355X_MICROARCH.md prescribes) on the same device after the "
                         "timed regions.  auto = for the default c3 line at N = 1 (headline, and the plan-free c2 / c5 `also` "
                         "records); off = the figure of the committed passes (profiles/*traffic*.json), labelled as such")
    ap.add_argument("--traffic-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--ceiling-reps", type=int, default=5,
                    help="launches of the read-only kernel timed for roofline.read_ceiling_GBps (0 = skip)")
    ap.add_argument("--force-comm", action="store_true",
                    help="N=1 only: still create the RCCL communicator and run the gatherv in every "
                         "call (rehearsal of the N>1 code path on a 1-GPU box)")
    ap.add_argument("--parallelism", default="ranks", choices=["ranks", "threads"],
                    help="ranks (default): one process per GPU, the RCCL gatherv between them.  threads: ONE process drives all "
                         "N shards through the resident multi-GPU handle (rsp_mcsc_*: what an R session -- one process -- reaches; "
                         "shard k on device k %% device_count, so a 1-GPU box rehearses it with every shard on its one card); "
                         "no torchrun, same JSON line with config.parallelism 'threads'")
    ap.add_argument("--threads-figure", default="auto", choices=["auto", "on", "off"],
                    help="N > 1, --parallelism ranks: after the ranks have finished, rank 0 runs `--parallelism threads` over the "
                         "same devices as a CHILD process (a timeout around it; its failure costs the line nothing) and adds its "
                         "figures to the line as roofline.threads_*.  auto = on a real multi-GPU run (rendezvous nccl)")
    ap.add_argument("--verbose", action="store_true", help="add the glossary of the line's keys to the line (`notes`)")
    ap.add_argument("--explain", action="store_true", help="print the glossary of the line's keys and exit")
    return ap.parse_args(argv)


def relaunch_under_torchrun(args) -> int:
    """`python bench.py --gpus N` typed by hand: start the N ranks as a child."""
    port = str(29500 + os.getpid() % 2000)
    if args.rendezvous == "gloo":
        # the rehearsal on a box with fewer GPUs than ranks: the ranks are started directly, with the environment
        # torch.distributed.run would give them (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*).  torchrun's agent process
        # itself holds the GPU open on this image, and this pool allows six processes per card: without the agent a
        # one-GPU box takes six ranks instead of five (five under pytest, whose own process holds a context too).
        procs = []
        for r in range(args.gpus):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                       MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
        rc = 0
        while procs:
            for pr in list(procs):
                code = pr.poll()
                if code is None:
                    continue
                procs.remove(pr)
                if code != 0 and rc == 0:       # one rank failed: the others would wait for it in a collective forever
                    rc = code
                    for other in procs:
                        other.terminate()       # (exactly the processes started above)
            time.sleep(0.05)
        return rc
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def build_offsets(name, nnz_override=0):
    from rcppsparse_amd import synth
    nrow, ncol, nnz, shape = WORKLOADS[name]
    if nnz_override:
        nnz = nnz_override
    if shape == "uniform":
        counts = synth.uniform_counts(ncol, nnz, SEED, nrow)
    elif shape == "zipf-descending":
        counts = synth.zipf_counts(ncol, nnz, SEED, nrow, order="descending")
    else:
        counts = synth.zipf_counts(ncol, nnz, SEED, nrow)
    return nrow, ncol, nnz, shape, synth.offsets_from_counts(counts)


def usable_cores():
    """Host cores this process may really use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return n


def sig(v, digits=6):
    """Nested numbers of the line carry six significant digits (the contract's own keys stay exact)."""
    if isinstance(v, float):
        return float(f"{v:.{digits}g}")
    if isinstance(v, dict):
        return {k: sig(x, digits) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [sig(x, digits) for x in v]
    return v


def cpu_baseline(p, kind, target_nnz=1_000_000_000, reps=10):
    """Oracle (kind "port": restatement of reference src/example.cpp:26-32), 1 thread (the
    reference path has no OpenMP), on the host of this box: the whole matrix when the host
    has room for x (8 B/nnz), else the first columns holding what fits (~10 s of CPU work)."""
    import numpy as np
    import oracle
    try:
        free = os.sysconf("SC_AVPHYS_PAGES") * os.sysconf("SC_PAGE_SIZE")
        target_nnz = min(target_nnz, max(10_000_000, int(free * 0.4) // 8))
    except (ValueError, OSError):
        target_nnz = min(target_nnz, 200_000_000)
    ncol_s = int(np.searchsorted(p, target_nnz, side="right")) - 1
    ncol_s = max(1, min(ncol_s, len(p) - 1))
    nnz_s = int(p[ncol_s])
    x = oracle.gen_values(nnz_s, SEED, 0, kind)
    ps = np.ascontiguousarray(p[:ncol_s + 1])
    oracle.column_sums(x, ps)            # warm-up
    times = []
    for _ in range(reps):
        t0 = time.perf_counter()
        oracle.column_sums(x, ps)
        times.append(time.perf_counter() - t0)
    times.sort()
    med = times[len(times) // 2]
    out = {
        "value": nnz_s / med, "unit": "nnz/s", "cores": 1, "kind": "port",
        "sample": f"first {ncol_s} columns ({nnz_s} nnz) of the same matrix, {reps} reps, median",
        "best": nnz_s / times[0], "host_cpus": os.cpu_count(),
    }
    # optional second figure: the same per-column loop under an OpenMP parallel-for over
    # the columns on all host cores this process may use.  Not the reference's behaviour (its path
    # is single-threaded): reported for scale only.
    try:
        nthreads = usable_cores()
        if nthreads > 1:
            del x
            xt = oracle.gen_values_threads(nnz_s, SEED, 0, kind, nthreads)   # first touch by the same threads
            oracle.column_sums_threads(xt, ps, nthreads)
            tt = []
            for _ in range(reps):
                t0 = time.perf_counter()
                oracle.column_sums_threads(xt, ps, nthreads)
                tt.append(time.perf_counter() - t0)
            tt.sort()
            out["all_cores_value"] = nnz_s / tt[len(tt) // 2]
            out["all_cores"] = nthreads
    except Exception as e:   # never let the optional figure break the bench line
        out["all_cores_error"] = str(e)[:120]
    return sig(out)


def parity_whole_matrix(got, p, kind, seed=SEED, first_idx=0, slab_nnz=60_000_000):
    """EVERY column of `got` against the oracle (outside the timed region).  The matrix is walked
    in slabs of whole columns; each slab's x is regenerated on the host by the oracle's counter-based
    generator (bit-identical to the device generator), summed by the oracle's per-column loop
    (the threaded variant runs the same sequential loop per column: identical bits) and compared
    in the grading form |gpu - ref| <= 1e-12 * sum|x| per column."""
    import numpy as np
    import oracle
    ncol = len(p) - 1
    p64 = np.asarray(p, dtype=np.int64)
    nthreads = max(1, usable_cores())
    worst, worst_rel, nbad, empties_exact = 0.0, 0.0, 0, True
    c0 = 0
    while c0 < ncol:
        c1 = int(np.searchsorted(p64, p64[c0] + slab_nnz, side="right")) - 1
        c1 = min(ncol, max(c1, c0 + 1))
        lo, hi = int(p64[c0]), int(p64[c1])
        pl = (p64[c0:c1 + 1] - lo).astype(np.int32)
        if hi > lo:
            xv = oracle.gen_values_threads(hi - lo, seed, first_idx + lo, kind, nthreads)
            ref = oracle.column_sums_threads(xv, pl, nthreads)
            scale = oracle.column_abs_sums(xv, pl)
        else:
            ref = np.zeros(c1 - c0)
            scale = np.zeros(c1 - c0)
        g = got[c0:c1]
        err = np.abs(g - ref)
        nbad += int(np.count_nonzero(~(err <= 1e-12 * scale)))
        nz = scale > 0
        if nz.any():
            worst = max(worst, float(np.max(err[nz] / scale[nz])))
        well = nz & (np.abs(ref) >= 1e-3 * scale)          # columns without heavy cancellation
        if well.any():
            worst_rel = max(worst_rel, float(np.max(err[well] / np.abs(ref[well]))))
        empty = pl[1:] == pl[:-1]
        if empty.any():
            empties_exact = empties_exact and bool(np.all(g[empty] == 0.0) and not np.any(np.signbit(g[empty])))
        c0 = c1
    return {"max_abs_err_over_l1": worst, "tolerance": 1e-12,
            "max_rel_err_where_ref_ge_1e-3_l1": worst_rel,
            "columns_checked": "all", "ncol": ncol, "columns_out_of_tolerance": nbad,
            "empty_columns_exactly_plus_zero": empties_exact}


def parity_ok(par):
    return par["columns_out_of_tolerance"] == 0 and par["empty_columns_exactly_plus_zero"]


def traffic_from_profiles(workload):
    """(HBM bytes per launch, file) from the latest committed PMC passes (profiles/*traffic*.json)."""
    import glob
    best, src = None, None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*traffic*.json"))):
        try:
            d = json.load(open(f))
            if d.get("workload") == workload and d.get("hbm_bytes_per_launch") is not None:
                best, src = d.get("hbm_bytes_per_launch"), os.path.relpath(f, ROOT)
        except Exception:
            pass
    return best, src


def traffic_child(args):
    """The program the counter passes are pointed at: the workload resident in HBM and a few plan-free calls, nothing else."""
    import torch
    from rcppsparse_amd import capi
    capi.load()
    torch.cuda.set_device(0)
    nrow, ncol, nnz, shape, p = build_offsets(args.workload, args.nnz)
    x = torch.empty(nnz, dtype=torch.float64, device="cuda")
    capi.gen_values_device(x, SEED, 0, args.kind)
    pt = torch.from_numpy(p).cuda()
    out = torch.empty(ncol, dtype=torch.float64, device="cuda")
    ws = capi.alloc_workspace(ncol, nnz, "cuda")
    run = capi.prepared_column_sums(x, pt, out, ws)
    run()
    capi.column_sums_device_settle(pt, nnz)     # the entry plans for itself: count the form it settles on
    for _ in range(max(1, args.steps)):
        run()
    torch.cuda.synchronize()


def traffic_measured_now(workload, kind):
    """(HBM bytes per call, detail) from two counter passes run NOW as child processes on this device, or (None, why not).
    rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (never with a trace option); KiB -> bytes; FETCH_SIZE
    doubled (gfx950 tallies the 128-byte requests of a 16 B/lane stream at 64 B: MI355X_MICROARCH.md); main kernel + fix-up."""
    import csv
    import glob
    import shutil
    import signal
    import tempfile
    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(prof):
        return None, "rocprofv3 not found on this box"
    means = {}
    with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, counter)
            cmd = [prof, "--pmc", counter, "--output-format", "csv", "-d", out, "--", sys.executable, os.path.abspath(__file__),
                   "--traffic-child", "--workload", workload, "--steps", "3", "--kind", str(kind)]
            env = dict(os.environ, TMPDIR="/tmp")
            try:
                pr = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE,
                                      start_new_session=True)
                try:
                    _, err = pr.communicate(timeout=150)
                except subprocess.TimeoutExpired:
                    os.killpg(pr.pid, signal.SIGKILL)      # (the group this call started, nothing else)
                    pr.communicate()
                    return None, f"the {counter} pass did not finish in 150 s"
                if pr.returncode != 0:
                    return None, f"the {counter} pass exited with {pr.returncode}: {err.decode(errors='replace')[-200:]}"
            except OSError as e:
                return None, f"the {counter} pass could not be started: {e}"
            rows = []      # (dispatch id, kernel family, KiB) of every column-sum launch of the child, in dispatch order
            for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    if r["Counter_Name"] != counter:
                        continue
                    for k in COLSUM_KERNELS:
                        if k in r["Kernel_Name"]:
                            rows.append((int(r.get("Dispatch_Id", len(rows))), k, float(r["Counter_Value"])))
            if not rows:
                return None, f"the {counter} pass reported no column-sum launch"
            rows.sort()
            # the child's last calls all take the form the entry has settled on: the last launch names it
            last = rows[-1][1]
            fams = ("colsums_chunks_kernel", "colsums_fixup_kernel") if last in ("colsums_chunks_kernel", "colsums_fixup_kernel") else (last,)
            means[counter] = 0.0
            for fam in fams:      # KiB per call: mean over the last three launches of each kernel of the call
                vals = [v for _, k, v in rows if k == fam][-3:]
                means[counter] += sum(vals) / len(vals)
            family = "+".join(f.replace("colsums_", "").replace("_kernel", "") for f in fams)
    rd, wr = 2 * 1024 * means["FETCH_SIZE"], 1024 * means["WRITE_SIZE"]
    return rd + wr, {"read_bytes": rd, "write_bytes": wr, "kernels": family}


COLSUM_KERNELS = ("colsums_chunks_kernel", "colsums_fixup_kernel", "colsums_lean_kernel", "colsums_columns_kernel")
ALSO_AUTO = ("c2", "c2:planned", "c2:planned-device", "c5", "c4shard", "c4shard:planned", "vignette:planned")
ALSO_TRAFFIC_NOW = ("c2", "c5")          # plan-free records whose HBM traffic the default line re-measures in the run
ALSO_SHARDED_AUTO = ("c5:nnz", "c5:cols")
PLAN_FORMS = {3: "columns", 2: "lean", 1: "snapped", 0: "general"}
N_REGIONS_SMALL = 3


def key_of(spec):
    return spec.replace(":", "_").replace("-", "_")


def small_call_regions(torch, stream, launch_k, steps, fence, nregions=N_REGIONS_SMALL):
    """Small calls: `nregions` timed regions of `steps` calls back to back, ONE HIP event pair around each (an event pair per
    call would idle the queue).  Returns [(wall seconds, device ms per call)] in the order measured."""
    out = []
    for _ in range(nregions):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        fence()
        t0 = time.perf_counter()
        e0.record(stream)
        for k in range(steps):
            launch_k(k)
        e1.record(stream)
        fence()
        out.append((time.perf_counter() - t0, e0.elapsed_time(e1) / steps))
    return out


def median_region(regions):
    """The region with the median wall time (an odd number of regions)."""
    order = sorted(range(len(regions)), key=lambda k: regions[k][0])
    return regions[order[len(order) // 2]]


def also_record(torch, capi, spec, args, dev, dev_index, stream, traffic_now=False):
    """One more single-GPU workload, measured like the headline one (inputs resident in HBM, rotated copies of x
    where the workload would fit the Infinity Cache, K calls back to back on one stream, HIP events on that
    stream, every column against the oracle) and reported as a compact record."""
    name, _, mode = spec.partition(":")
    planned = mode in ("planned", "planned-device")
    nrow, ncol, nnz, shape, p = build_offsets(name, 0)
    small = nnz < 200_000_000
    steps = args.steps * 10 if small else args.steps      # a 20 us call wants more than 20 of them in the region
    ncopies = max(1, -(-400_000_000 // max(1, 8 * nnz)))
    xs = []
    for k in range(ncopies):
        xk = torch.empty(nnz, dtype=torch.float64, device=dev)
        capi.gen_values_device(xk, SEED + k, 0, args.kind)
        xs.append(xk)
    pt = torch.from_numpy(p).to(dev)
    out = torch.empty(ncol, dtype=torch.float64, device=dev)
    ws = capi.alloc_workspace(ncol, nnz, dev)
    early = None
    if mode == "planned-device":
        # p[] is never shown to the host: the inspection runs as kernels on the launch stream, nothing waits for it, and
        # the calls issued meanwhile are answered by the general kernels (counted here) until the host has seen the result
        capi.ColumnSumsPlan(pt, nnz=nnz, stream=stream).wait().close()   # (first use loads the inspector's kernels: not what plan_ms is about)
        torch.cuda.synchronize()
        plan = capi.ColumnSumsPlan(pt, nnz=nnz, stream=stream)
        launches = [plan.prepared(xk, pt, out, ws, stream=stream) for xk in xs]
        early = 0
        while not plan.ready() and early < 10_000:
            launches[early % ncopies]()
            early += 1
    else:
        plan = capi.ColumnSumsPlan(p, nnz=nnz, device=dev_index) if planned else None
        launches = [(plan.prepared(xk, pt, out, ws, stream=stream) if plan is not None else
                     capi.prepared_column_sums(xk, pt, out, ws, stream=stream)) for xk in xs]
    for k in range(max(args.warmup, ncopies)):
        launches[k % ncopies]()
    auto_form = None
    if plan is None:
        # the plan-free entry plans for itself (include/rcppsparse_hip.h): wait until it has settled on a form, then warm that
        auto_form = capi.column_sums_device_settle(pt, nnz, stream=stream)
        for k in range(max(args.warmup, ncopies)):
            launches[k % ncopies]()
    mk = lambda: torch.cuda.Event(enable_timing=True)   # noqa: E731
    torch.cuda.synchronize()
    regions_ms = None
    if small:
        regions = small_call_regions(torch, stream, lambda k: launches[k % ncopies](), steps, torch.cuda.synchronize)
        wall, kernel_ms = median_region(regions)
        regions_ms = [w / steps * 1e3 for w, _ in regions]
    else:          # an event pair around the kernels of every call
        evs = [(mk(), mk()) for _ in range(steps)]
        t0 = time.perf_counter()
        for k in range(steps):
            evs[k][0].record(stream)
            launches[k % ncopies]()
            evs[k][1].record(stream)
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        kernel_ms = sum(a.elapsed_time(b) for a, b in evs) / steps
    launches[0]()          # copy 0 (seed SEED) for the parity check
    torch.cuda.synchronize()
    par = parity_whole_matrix(out.cpu().numpy(), p, args.kind)
    if not parity_ok(par):
        raise SystemExit(f"parity check failed on also:{spec}: {json.dumps(par)}")
    algo = 8 * nnz + 4 * (ncol + 1) + 8 * ncol
    achieved = algo / (kernel_ms * 1e-3) / 1e9
    one_launch = (plan is not None and plan.snapped) or auto_form in ("lean", "columns")
    ms_per_call = wall / steps * 1e3
    rec = {"workload": spec,
           "form": (auto_form if auto_form != "unknown" else "general") if plan is None else PLAN_FORMS[plan.form],
           "planned_by": "entry" if plan is None else "caller",
           "launches": 1 if one_launch else 2,
           "plan_ms": None if plan is None else plan.inspect_ms,
           "plan_by": None if plan is None else ("device" if plan.device_made else "host"),
           "early_general_calls": early,
           "steps": steps, "x_copies": ncopies,
           "ms_per_call": ms_per_call, "regions_ms": regions_ms, "kernel_ms": kernel_ms,
           "host_stall_suspected": bool(ms_per_call > 1.5 * kernel_ms),
           "algo_bytes": algo, "frac": achieved / HBM_PEAK_GBPS,
           "parity_err": par["max_abs_err_over_l1"], "bad_columns": par["columns_out_of_tolerance"]}
    if plan is not None:
        plan.close()
    del xs, launches, out, ws, pt
    torch.cuda.empty_cache()
    measured = None
    if traffic_now and plan is None:
        measured, _how = traffic_measured_now(name, args.kind)
    if measured is not None:
        rec["traffic"], rec["traffic_in_run"], rec["traffic_kernels"] = measured, True, _how["kernels"]
    else:
        # (committed passes: <workload>_traffic.json is the plan-free command's, <workload>planned_traffic.json the --planned one's)
        rec["traffic"], rec["traffic_in_run"] = traffic_from_profiles(name + ("planned" if plan is not None and plan.snapped else ""))[0], False
    return sig(rec)


def timed_steps(torch, dist, world, stat_dev, fence, steps, step_fn):
    """K calls of step_fn back to back between two fences; the MAX over ranks of the wall time."""
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        step_fn()
    fence()
    el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=stat_dev)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    return float(el[0])


def planned_shards_figure(ctx):
    """N > 1, separate key: every rank inspects ITS shard's offsets once (a ColumnSumsPlan on p_local: at C4 shard
    size the columns form, one launch per call instead of main kernel + fix-up) and the same protocol as `value` runs
    again -- planned launch, then that call's gather, in order on one stream.  Never feeds `value`."""
    torch, dist, capi, sharded = ctx["torch"], ctx["dist"], ctx["capi"], ctx["sharded"]
    args, shard, world, rank = ctx["args"], ctx["shard"], ctx["world"], ctx["rank"]
    plan, err = None, None
    try:
        plan = capi.ColumnSumsPlan(shard.p_local, nnz=shard.nnz, device=ctx["dev_index"])
    except Exception as e:            # (a rank that cannot plan must not leave the others alone in the collectives below)
        err = str(e)
    ok = torch.tensor([0.0 if plan is None else 1.0], device=ctx["stat_dev"])
    if world > 1:
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if float(ok[0]) == 0.0:
        if plan is not None:
            plan.close()
        return {"value": None, "note": f"not measured: a rank could not make its plan ({err})"[:160]} if rank == 0 else None
    xs, pt, out, ws, s_main = ctx["xs"], ctx["pt"], ctx["out_main"], ctx["ws_main"], ctx["s_main"]
    launches = [plan.prepared(xk, pt, out, ws, stream=s_main) for xk in xs]
    n = [0]

    def compute(_shard):
        launches[n[0] % len(launches)]()
        n[0] += 1
        return out
    driver = sharded.ShardedColumnSums(shard, compute, ctx["new_gather"](s_main))
    recv = ctx["recv"]
    for _ in range(args.warmup):
        driver.step(recv)
    elapsed = timed_steps(torch, dist, world, ctx["stat_dev"], ctx["fence"], args.steps, lambda: driver.step(recv))
    n[0] = 0
    driver.step(recv)
    ctx["fence"]()
    forms = [None] * world
    mine = PLAN_FORMS[plan.form]
    if world > 1:
        dist.all_gather_object(forms, mine)
    else:
        forms = [mine]
    fig = None
    if rank == 0:
        full = (recv if ctx["use_comm"] else out).cpu().numpy()
        par = parity_whole_matrix(full, ctx["p"], args.kind)
        fig = sig({"value": ctx["nnz"] * args.steps / elapsed, "ms_per_step": elapsed / args.steps * 1e3,
                   "forms_by_rank": forms, "plan_ms_rank0": plan.inspect_ms,
                   "parity_err": par["max_abs_err_over_l1"], "bad_columns": par["columns_out_of_tolerance"]})
    plan.close()
    return fig


def direct_gather_figure(ctx):
    """N > 1, separate key: the direct-write comparator of the gatherv.  Rank 0 exports its result
    buffer (rsp_shared_result_alloc: hipIpcGetMemHandle), the other rank processes map it, and every rank's kernels take
    `mapped + displacement` as their output pointer: the exchange is the kernels' own result stores over xGMI plus one
    fence per call (every rank waits for its stream, then the ranks cross a shared-memory barrier).  Never `value`."""
    torch, dist, capi = ctx["torch"], ctx["dist"], ctx["capi"]
    args, world, rank = ctx["args"], ctx["world"], ctx["rank"]
    ncol, displs, s_main = ctx["ncol"], ctx["displs"], ctx["s_main"]
    note = None
    try:
        box = [None, None]
        if rank == 0:
            shared = capi.SharedResult(ncol)
            box = [shared.handle, f"/rsp_bench_{os.getpid()}"]
        if world > 1:
            dist.broadcast_object_list(box, src=0)
        if rank != 0:
            shared = capi.SharedResult(ncol, handle=box[0])
        barrier = capi.HostBarrier(box[1], world, rank)
    except Exception as e:                      # e.g. IPC not offered by this host: say so, measure nothing
        note = f"not measured: {e}"[:160]
        ok = torch.tensor([0.0], device=ctx["stat_dev"])
    else:
        ok = torch.tensor([1.0], device=ctx["stat_dev"])
    if world > 1:
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if float(ok[0]) == 0.0:
        notes = [None] * world
        if world > 1:
            dist.all_gather_object(notes, note)
        return {"value": None, "note": ([n for n in notes if n] or [note])[0]} if rank == 0 else None

    class Out:                                   # what the prepared launcher needs of an output tensor
        def __init__(self, ptr):
            self._p = ptr

        def data_ptr(self):
            return self._p
    out = Out(shared.ptr + 8 * int(displs[rank]))
    launches = [capi.prepared_column_sums(xk, ctx["pt"], out, ctx["ws_main"], stream=s_main) for xk in ctx["xs"]]
    n = [0]

    def step():
        launches[n[0] % len(launches)]()
        n[0] += 1
        s_main.synchronize()                      # this rank's slice is complete in the root's memory ...
        barrier.wait()                            # ... and after the barrier every rank's is
    for _ in range(args.warmup):
        step()
    elapsed = timed_steps(torch, dist, world, ctx["stat_dev"], ctx["fence"], args.steps, step)
    n[0] = 0
    step()
    fig = None
    if rank == 0:
        par = parity_whole_matrix(shared.read(stream=s_main), ctx["p"], args.kind)
        fig = sig({"value": ctx["nnz"] * args.steps / elapsed, "ms_per_step": elapsed / args.steps * 1e3,
                   "parity_err": par["max_abs_err_over_l1"], "bad_columns": par["columns_out_of_tolerance"]})
    barrier.wait()
    barrier.close()
    if rank != 0:
        shared.close()
    if world > 1:
        dist.barrier()
    if rank == 0:
        shared.close()
    return fig


PROBE_N = 4096


def peer_probe_writer(args):
    """Grandchild of the probe: a SECOND process on the writer's device maps the owner's buffer (hipIpcOpenMemHandle) and lets
    the column-sum kernels store PROBE_N results through the mapping -- exactly what a rank does in direct_gather."""
    import numpy as np
    import torch
    from rcppsparse_amd import capi
    handle, writer = bytes.fromhex(args.peer_probe_writer[0]), int(args.peer_probe_writer[1])
    capi.load()
    torch.cuda.set_device(writer)
    shared = capi.SharedResult(PROBE_N, handle=handle)

    class Out:
        def data_ptr(self):
            return shared.ptr
    x = torch.arange(1, PROBE_N + 1, dtype=torch.float64, device=f"cuda:{writer}")
    pt = torch.arange(0, PROBE_N + 1, dtype=torch.int32, device=f"cuda:{writer}")      # one entry per column: sums = 1 .. n
    ws = capi.alloc_workspace(PROBE_N, PROBE_N, f"cuda:{writer}")
    capi.prepared_column_sums(x, pt, Out(), ws)()
    torch.cuda.synchronize()
    shared.close()
    del np
    return 0


def peer_probe(args):
    """The probe's child process: owns a small IPC-exported buffer on OWNER's device, has a second process store into it from
    WRITER's device, reads it back.  Exit code 0 and ONE line on stdout: passed | same device | no peer access | failed: why."""
    import numpy as np
    import torch
    from rcppsparse_amd import capi
    owner, writer = args.peer_probe
    if owner == writer:
        print("same device")
        return 0
    capi.load()
    try:
        if not (capi.device_can_access_peer(writer, owner) and capi.device_can_access_peer(owner, writer)):
            print("no peer access")
            return 0
        torch.cuda.set_device(owner)
        shared = capi.SharedResult(PROBE_N)
        env = dict(os.environ)
        pr = subprocess.run([sys.executable, os.path.abspath(__file__), "--peer-probe-writer", shared.handle.hex(), str(writer)],
                            env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=90)
        if pr.returncode != 0:
            print(f"failed: the writing process exited with {pr.returncode}: {pr.stderr.decode(errors='replace')[-120:]}".replace("\n", " "))
            return 0
        got = shared.read()
        shared.close()
        if np.array_equal(got, np.arange(1, PROBE_N + 1, dtype=np.float64)):
            print("passed")
        else:
            print(f"failed: {int(np.count_nonzero(got != np.arange(1, PROBE_N + 1)))} of {PROBE_N} values did not arrive")
    except subprocess.TimeoutExpired:
        print("failed: the writing process did not finish in 90 s")
    except Exception as e:   # noqa: BLE001
        print(f"failed: {e}"[:160].replace("\n", " "))
    return 0


def run_peer_probe(owner, writer, timeout_s=150, command=None):
    """Starts the probe as a child process of its own session and returns its one-line verdict.  A child that dies (a fault
    ends the process that owns the queue), hangs or prints nothing is a failed probe -- never an exception here."""
    import signal
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK", "ROLE_NAME",
                        "MASTER_ADDR", "MASTER_PORT", "GROUP_WORLD_SIZE", "ROLE_WORLD_SIZE") and not k.startswith("TORCHELASTIC_")}
    cmd = command or [sys.executable, os.path.abspath(__file__), "--peer-probe", str(owner), str(writer)]
    try:
        pr = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, start_new_session=True)
        try:
            out, err = pr.communicate(timeout=timeout_s)
        except subprocess.TimeoutExpired:
            os.killpg(pr.pid, signal.SIGKILL)      # (the group this call started, nothing else)
            pr.communicate()
            return f"failed: the probe did not finish in {timeout_s} s"
    except OSError as e:
        return f"failed: the probe could not be started: {e}"[:160]
    if pr.returncode != 0:
        return f"failed: the probe exited with {pr.returncode}: {err.decode(errors='replace')[-100:]}".replace("\n", " ")[:200]
    lines = [ln.strip() for ln in out.decode(errors="replace").splitlines() if ln.strip()]
    verdicts = [ln for ln in lines if ln.startswith(("passed", "same device", "no peer access", "failed"))]
    return verdicts[-1][:200] if verdicts else "failed: the probe printed no verdict"


def host_gather_figure(ctx):
    """N > 1, separate key: SURVEY.md section 5's other comparator -- per-GPU D2H into disjoint slices of ONE pinned buffer.
    The buffer lives in POSIX shared memory, every rank maps and page-locks it; per call a rank launches its kernels, copies
    its slice behind them (its own host link), waits for its stream and crosses a host barrier.  Never `value`."""
    torch, dist, capi = ctx["torch"], ctx["dist"], ctx["capi"]
    args, world, rank = ctx["args"], ctx["world"], ctx["rank"]
    ncol, displs, s_main, shard = ctx["ncol"], ctx["displs"], ctx["s_main"], ctx["shard"]
    note, vec, barrier = None, None, None
    try:
        box = [None]
        if rank == 0:
            box = [f"/rsp_bench_host_{os.getpid()}"]
        if world > 1:
            dist.broadcast_object_list(box, src=0)
        if rank == 0:
            vec = capi.SharedHostVector(box[0], ncol, create=True)
        if world > 1:
            dist.barrier()
        if rank != 0:
            vec = capi.SharedHostVector(box[0], ncol, create=False)
        barrier = capi.HostBarrier(box[0] + "_b", world, rank)
        ok = torch.tensor([1.0], device=ctx["stat_dev"])
    except Exception as e:   # noqa: BLE001
        note = f"not measured: {e}"[:160]
        ok = torch.tensor([0.0], device=ctx["stat_dev"])
    if world > 1:
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if float(ok[0]) == 0.0:
        notes = [None] * world
        if world > 1:
            dist.all_gather_object(notes, note)
        if vec is not None:
            vec.close()
        return {"value": None, "note": ([n for n in notes if n] or [note])[0]} if rank == 0 else None
    out = torch.empty(max(shard.ncol, 1), dtype=torch.float64, device=ctx["dev"])[:shard.ncol]
    launches = [capi.prepared_column_sums(xk, ctx["pt"], out, ctx["ws_main"], stream=s_main) for xk in ctx["xs"]]
    n = [0]

    def step():
        launches[n[0] % len(launches)]()
        n[0] += 1
        vec.copy_from_device(out, int(displs[rank]), stream=s_main)
        s_main.synchronize()                      # this rank's slice is in the shared vector ...
        barrier.wait()                            # ... and after the barrier every rank's is
    for _ in range(args.warmup):
        step()
    elapsed = timed_steps(torch, dist, world, ctx["stat_dev"], ctx["fence"], args.steps, step)
    n[0] = 0
    step()
    fig = None
    if rank == 0:
        par = parity_whole_matrix(vec.array.copy(), ctx["p"], args.kind)
        fig = sig({"value": ctx["nnz"] * args.steps / elapsed, "ms_per_step": elapsed / args.steps * 1e3,
                   "parity_err": par["max_abs_err_over_l1"], "bad_columns": par["columns_out_of_tolerance"]})
    barrier.wait()
    barrier.close()
    if rank != 0:
        vec.close()
    if world > 1:
        dist.barrier()
    if rank == 0:
        vec.close()
    return fig


def rows_parity(got, p, nrow, ncol, kind):
    """EVERY row of `got` against the oracle's scatter loop over the whole matrix (reference RcppSparse.h:138-144), column slab
    after column slab in storage order: (max |gpu - ref| / sum|x| over the rows, rows out of the 1e-12 tolerance)."""
    import numpy as np
    import oracle
    ref, scale = np.zeros(nrow), np.zeros(nrow)
    p64 = np.asarray(p, dtype=np.int64)
    c0 = 0
    while c0 < ncol:
        c1 = int(np.searchsorted(p64, p64[c0] + 40_000_000, side="right")) - 1
        c1 = min(ncol, max(c1, c0 + 1))
        lo, hi = int(p64[c0]), int(p64[c1])
        if hi > lo:
            xv = oracle.gen_values_threads(hi - lo, SEED, lo, kind, max(1, usable_cores()))
            iv = oracle.gen_row_indices(p, nrow, SEED, c0, c1)
            oracle.row_sums_accumulate(xv, iv, ref, scale)
        c0 = c1
    err = np.abs(got - ref)
    nbad = int(np.count_nonzero(~(err <= 1e-12 * scale)))
    nz = scale > 0
    worst = float(np.max(err[nz] / scale[nz])) if nz.any() else 0.0
    return worst, nbad


def main_threads_rowsums(args):
    """--op rowsums --parallelism threads: Matrix::rowSums through the single-process multi-GPU handle (rsp_mcsc_row_sums):
    every shard sums the rows of its own columns, the partial vectors are added in shard order ON THE DEVICES (slices
    exchanged device to device, one add kernel per device, reduced slices written into a page-locked host vector) and the
    call returns when the nrow sums are in the caller's host vector.  Three regions of K synchronous calls, the median counts."""
    import numpy as np
    import torch
    from rcppsparse_amd import capi, sharded
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    capi.load()
    G = max(1, args.gpus)
    ndev = torch.cuda.device_count()
    devices = [k % ndev for k in range(G)]
    nrow, ncol, nnz, shape, p = build_offsets(args.workload, args.nnz)
    shards = [sharded.make_shard(p, k, G, balance=args.partition) for k in range(G)]
    # the rows of the WHOLE matrix (the generator hashes the global element index), made once on device 0, dealt out
    pg = torch.from_numpy(np.ascontiguousarray(p)).to("cuda:0")
    i_all = torch.empty(nnz, dtype=torch.int32, device="cuda:0")
    with torch.cuda.device(0):
        capi.gen_row_indices_device(i_all, pg, nrow, SEED)
        torch.cuda.synchronize()
    xs, ps, its = [], [], []
    for k, sh in enumerate(shards):
        with torch.cuda.device(devices[k]):
            d = f"cuda:{devices[k]}"
            xt = torch.empty(max(sh.nnz, 2), dtype=torch.float64, device=d)[:sh.nnz]
            if sh.nnz:
                capi.gen_values_device(xt, SEED, sh.x0, args.kind)
            xs.append(xt)
            ps.append(torch.from_numpy(sh.p_local).to(d))
            its.append(i_all[sh.x0:sh.x1].to(d, copy=True))
            torch.cuda.synchronize()
    del i_all, pg
    torch.cuda.empty_cache()
    h = capi.MultiDeviceCSC.wrap_device(xs, ps, nrow, i_ts=its)
    h.row_sums()                                                    # (builds every shard's row form and the reduce's buffers)
    out = np.empty(nrow, dtype=np.float64)
    regions = []
    for _ in range(args.warmup):
        h.row_sums_into(out)
    for _ in range(N_REGIONS_SMALL):
        t0 = time.perf_counter()
        for _ in range(args.steps):
            h.row_sums_into(out)
        regions.append((time.perf_counter() - t0) / args.steps * 1e3)
    ms = sorted(regions)[len(regions) // 2]
    worst, nbad = rows_parity(out, p, nrow, ncol, args.kind)
    if nbad:
        raise SystemExit(f"rowSums parity check failed (threads): {nbad} rows out of tolerance (worst {worst:.3e})")
    forms = [h.shard_info(k)["form"] for k in range(G)]
    h.close()
    line = {"metric": "rowSums nnz/s (next row f1: Matrix::rowSums over column-range shards, shard-ordered reduce)",
            "value": nnz / (ms * 1e-3), "unit": "nnz/s", "n_gpus": G, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": sig({"workload": f"{args.workload}: {nrow}x{ncol} CSC dgCMatrix, nnz={nnz}, {shape} nnz/column, rows ascending "
                                       f"and distinct per column (stratified), values kind {args.kind}, seed {SEED}",
                           "op": "rowsums", "parallelism": "threads", "devices": ndev, "devices_distinct": len(set(devices)) == G,
                           "reduce": "on the devices: peer copies of row slices + rows_add_partials_kernel, shard order",
                           "result": "a pageable host vector of nrow doubles", "regions_ms": regions,
                           "shards": [{"shard": k, "device": devices[k], "c0": sh.c0, "c1": sh.c1, "nnz": sh.nnz, "column_form": forms[k]}
                                      for k, sh in enumerate(shards)]}),
            "roofline": sig({"bound": "hbm", "achieved": (12 * nnz + 8 * nrow) / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBPS * len(set(devices)),
                             "unit": "GB/s", "frac": (12 * nnz + 8 * nrow) / (ms * 1e-3) / 1e9 / (HBM_PEAK_GBPS * len(set(devices))),
                             "traffic": None, "kernel": "the whole call (every shard's accumulate pass, the reduce, the result's way to the host)",
                             "algorithmic_bytes_per_launch": 12 * nnz + 8 * nrow}),
            "parity": {"max_abs_err_over_l1": worst, "tolerance": 1e-12, "rows_checked": "all", "nrow": nrow, "rows_out_of_tolerance": nbad},
            "cpu_baseline": None}
    print(json.dumps(line), flush=True)


def main_rowsums(args):
    """bench.py --op rowsums: Matrix::rowSums (reference RcppSparse.h:138-144) over column-range shards.  A step = every rank
    sums the rows of its own columns (rsp_row_sums_device: nrow doubles) and the partial vectors are reduced IN RANK ORDER to
    rank 0 (rsp_comm_reduce_rows: all-to-all of row slices over xGMI, one add kernel, gatherv).  Same timing protocol as the
    headline path; parity = every row against the oracle's scatter loop over the whole matrix."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import numpy as np
    import torch
    import torch.distributed as dist
    from rcppsparse_amd import capi, sharded
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    rehearsal = args.rendezvous == "gloo"
    dev_index = local_rank % torch.cuda.device_count() if rehearsal else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    stat_dev = torch.device("cpu") if rehearsal else dev
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo") if rehearsal else dist.init_process_group("nccl", device_id=dev)
    capi.load()
    nrow, ncol, nnz, shape, p = build_offsets(args.workload, args.nnz)
    shard = sharded.make_shard(p, rank, world, balance=args.partition)
    # this rank's slices of x and i, generated in HBM (the row generator hashes the GLOBAL element index, so the rows are
    # those of the whole matrix: they are made from the global offsets and the shard's part is kept)
    x = torch.empty(shard.nnz, dtype=torch.float64, device=dev)
    capi.gen_values_device(x, SEED, shard.x0, args.kind)
    pg = torch.from_numpy(np.ascontiguousarray(p)).to(dev)
    i_all = torch.empty(nnz, dtype=torch.int32, device=dev)
    capi.gen_row_indices_device(i_all, pg, nrow, SEED)
    i_loc = i_all[shard.x0:shard.x1].clone()
    del i_all, pg
    torch.cuda.empty_cache()
    partial = torch.empty(nrow, dtype=torch.float64, device=dev)
    result = torch.empty(nrow, dtype=torch.float64, device=dev) if rank == 0 else None
    nbytes = int(capi.load().rsp_row_sums_workspace_bytes(nrow, shard.nnz))
    if nbytes == 0:
        raise SystemExit("rsp_row_sums_workspace_bytes failed")
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    s_main = torch.cuda.Stream()
    torch.cuda.set_stream(s_main)
    comm = None
    if world > 1 and not rehearsal:
        uid = torch.zeros(capi.UNIQUE_ID_BYTES, dtype=torch.uint8, device=dev)
        if rank == 0:
            uid.copy_(torch.frombuffer(bytearray(capi.comm_unique_id()), dtype=torch.uint8))
        dist.broadcast(uid, 0)
        comm = capi.Comm(bytes(uid.cpu().numpy().tobytes()), world, rank, local_rank)
        reduce = sharded.RcclReduceRows(comm, nrow, dev, 0, stream=s_main)
    elif world > 1:
        reduce = sharded.GlooReduceRows(dist, rank, world, 0)
    else:
        reduce = None

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def compute(_shard):
        return capi.row_sums_device(x, i_loc, nrow, out_t=(partial if reduce is not None else result), workspace=ws,
                                    stream=s_main)
    mk = lambda: torch.cuda.Event(enable_timing=True)   # noqa: E731
    evs = []

    def step(record=False):
        e = (mk(), mk(), mk()) if record else None
        if e:
            e[0].record(s_main)
        part = compute(shard)
        if e:
            e[1].record(s_main)
        if reduce is not None:
            reduce(part, result, 0)
        if e:
            e[2].record(s_main)
            evs.append(e)
    for _ in range(args.warmup):
        step()
    elapsed = timed_steps(torch, dist, world, stat_dev, fence, args.steps, step)
    for _ in range(args.steps):                  # the kernel / reduce split, outside the timed region
        step(record=True)
    fence()
    kernel_ms = sum(e[0].elapsed_time(e[1]) for e in evs) / len(evs)
    reduce_ms = sum(e[1].elapsed_time(e[2]) for e in evs) / len(evs)
    mine = torch.tensor([shard.c0, shard.c1, shard.nnz, kernel_ms, reduce_ms], dtype=torch.float64, device=stat_dev)
    per_rank = [torch.zeros_like(mine) for _ in range(world)]
    if world > 1:
        dist.all_gather(per_rank, mine)
    else:
        per_rank = [mine]
    out = None
    if rank == 0:
        got = result.cpu().numpy()
        worst, nbad = rows_parity(got, p, nrow, ncol, args.kind)
        if nbad:
            raise SystemExit(f"rowSums parity check failed: {nbad} rows out of tolerance (worst {worst:.3e})")
        # one rank's launch: reads its x and i (12 B per entry), writes nrow sums
        algo = 12 * shard.nnz + 8 * nrow
        achieved = algo / (kernel_ms * 1e-3) / 1e9
        out = {
            "metric": "rowSums nnz/s (next row f1: Matrix::rowSums over column-range shards, rank-ordered reduce)",
            "value": nnz * args.steps / elapsed, "unit": "nnz/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {nrow}x{ncol} CSC dgCMatrix, nnz={nnz}, {shape} nnz/column, rows "
                                   f"ascending and distinct per column (stratified), values kind {args.kind}, seed {SEED}",
                       "op": "rowsums",
                       "parallelism": "single" if world == 1 else ("rehearsal" if rehearsal else "ranges+rccl"),
                       "devices": torch.cuda.device_count(),
                       "reduce": None if reduce is None else reduce.name,
                       "rendezvous": args.rendezvous if world > 1 else None,
                       "shards": sig([{"rank": r, "c0": int(t[0]), "c1": int(t[1]), "nnz": int(t[2]), "kernel_ms": float(t[3]),
                                       "reduce_ms": float(t[4]) if world > 1 else None} for r, t in enumerate(per_rank)])},
            "roofline": sig({"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                             "frac": achieved / HBM_PEAK_GBPS, "traffic": None,
                             "kernel": "rsp_row_sums_device (histogram + partition + accumulate + combine) of rank 0's shard",
                             "kernel_ms": kernel_ms, "reduce_ms": reduce_ms if world > 1 else None,
                             "reduce_bytes_per_rank": None if world == 1 else 8 * nrow,
                             "algorithmic_bytes_per_launch": algo, "kernel_timing": "per_call_after"}),
            "parity": {"max_abs_err_over_l1": worst, "tolerance": 1e-12, "rows_checked": "all", "nrow": nrow,
                       "rows_out_of_tolerance": nbad},
        }
    if comm is not None:
        torch.cuda.synchronize()
        comm.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if out is not None:
        print(json.dumps(out), flush=True)


def main_threads(args):
    """--parallelism threads: ONE process, the resident multi-GPU handle (rsp_mcsc_*, csrc/multigpu.cpp) over N column-range
    shards.  Same line, same metric; a call is synchronous (it returns when the whole vector is in host memory)."""
    import numpy as np
    import torch
    from rcppsparse_amd import capi, sharded

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    capi.load()
    capi.set_tuning(args.chunk_rows)
    G = max(1, args.gpus)
    ndev = torch.cuda.device_count()
    devices = [k % ndev for k in range(G)]
    distinct = len(set(devices)) == G
    nrow, ncol, nnz, shape, p = build_offsets(args.workload, args.nnz)
    shards = [sharded.make_shard(p, k, G, balance=args.partition) for k in range(G)]
    xs, ps = [], []
    for k, sh in enumerate(shards):
        with torch.cuda.device(devices[k]):
            xt = torch.empty(max(sh.nnz, 2), dtype=torch.float64, device=f"cuda:{devices[k]}")[:sh.nnz]
            if sh.nnz:
                capi.gen_values_device(xt, SEED, sh.x0, args.kind)
            xs.append(xt)
            ps.append(torch.from_numpy(sh.p_local).to(f"cuda:{devices[k]}"))
            torch.cuda.synchronize()
    h = capi.MultiDeviceCSC.wrap_device(xs, ps, nrow)
    forms = [h.shard_info(k)["form"] for k in range(G)]
    kernel_ms = [h.shard_kernel_ms(k, reps=max(3, args.steps)) if shards[k].nnz else 0.0 for k in range(G)]
    pinned, pageable = h.result_buffer(), np.empty(ncol, dtype=np.float64)
    launch = h.config()["launch"]

    def run(gather, dest, steps, warmup, nregions=N_REGIONS_SMALL):
        """ms per call (median of `nregions` regions of `steps` synchronous calls back to back), ms per call of every region,
        median host time until the last shard's launch had been issued."""
        h.set_gather(gather)
        out = pinned if dest == "pinned" else pageable
        for _ in range(warmup):
            h.column_sums(out=out)
        regions, enq = [], []
        for _ in range(nregions):
            per_call = []
            t0 = time.perf_counter()
            for _ in range(steps):
                t1 = time.perf_counter()
                h.column_sums(out=out)
                per_call.append(time.perf_counter() - t1)
                enq.append(max(h.last_call_stamps()["enqueued_us"]))
            regions.append((time.perf_counter() - t0) / steps * 1e3)
            if os.environ.get("RSP_BENCH_DEBUG"):
                print(f"[threads] {gather}/{dest}: " + " ".join(f"{t * 1e6:.0f}" for t in per_call), file=sys.stderr, flush=True)
        return sorted(regions)[len(regions) // 2], regions, sorted(enq)[len(enq) // 2]

    # `value`: the handle's default gather (blit with a device per shard, d2h where shards share one), result in the
    # page-locked vector; three regions, the median one counts (one call in a few
    # hundred stalls for milliseconds somewhere below the library -- the host runtime or the box's CPU quota --, and K = 20
    # calls of ~0.2 ms cannot average that away: the same protocol as the small single-GPU workloads)
    gather0 = h.config()["gather"]
    ms_value, regions_ms, last_enq = run(gather0, "pinned", args.steps, args.warmup)
    got = np.array(pinned, copy=True)
    parity = parity_whole_matrix(got, p, args.kind)
    if not parity_ok(parity):
        raise SystemExit(f"parity check failed (threads, {args.workload}): {json.dumps(parity)}")
    figures = {f"{gather0}_pinned_ms": ms_value, "last_enqueue_us": last_enq}
    combos = [("blit" if gather0 == "d2h" else "d2h", "pinned"), (gather0, "pageable"), ("stores", "pinned"), ("none", "pinned")]
    if distinct:
        combos.append(("rccl", "pinned"))
    for gather, dest in combos:
        try:
            ms, _, _ = run(gather, dest, args.steps, args.warmup)
            figures[f"{gather}_{dest}_ms"] = ms
            if gather != "none" and np.asarray(pinned if dest == "pinned" else pageable).tobytes() != got.tobytes():
                figures[f"{gather}_{dest}_bits_differ"] = True
        except capi.RspError as e:
            figures[f"{gather}_{dest}_error"] = str(e)[:120]
    rccl = {}
    try:
        rccl = capi.rccl_info()
    except Exception as e:   # noqa: BLE001
        rccl = {"error": str(e)[:80]}
    h.set_gather(gather0)
    # the other launch mode by the protocol of `value` (the default comes from the shard count)
    other = "serial" if launch == "workers" else "workers"
    if G > 1:
        h.set_launch(other)
        figures[f"{other}_{gather0}_pinned_ms"], _, figures[f"{other}_last_enqueue_us"] = run(gather0, "pinned", args.steps, args.warmup)
        h.set_launch(launch)
    h.close()

    algo = 8 * shards[0].nnz + 4 * (shards[0].ncol + 1) + 8 * shards[0].ncol
    achieved = algo / (kernel_ms[0] * 1e-3) / 1e9 if kernel_ms[0] > 0 else 0.0
    roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
            "traffic": None, "kernel": "the shards' planned launches (config.shards[].form), shard 0 timed alone with HIP events",
            "kernel_ms": kernel_ms[0], "kernel_ms_max_over_shards": max(kernel_ms), "algorithmic_bytes_per_launch": algo,
            "call_minus_slowest_kernel_us": ms_value * 1e3 - max(kernel_ms) * 1e3}
    for k, v in figures.items():
        roof["threads_" + k] = v
    cfg = {"workload": f"{args.workload}: {nrow}x{ncol} CSC dgCMatrix, nnz={nnz}, {shape} nnz/column, values kind {args.kind}, seed {SEED}",
           "parallelism": "threads", "devices": ndev, "devices_distinct": distinct, "launch": launch, "gather": gather0,
           "result": "page-locked host vector", "partition": args.partition, "regions_ms": regions_ms,
           "shard_imbalance_max_over_mean": sharded.imbalance(p, shards[0].bounds),
           "rccl_version": rccl.get("version"), "rccl_library": rccl.get("library"),
           "shards": [{"shard": k, "device": devices[k], "c0": sh.c0, "c1": sh.c1, "nnz": sh.nnz, "kernel_ms": kernel_ms[k],
                       "form": forms[k]} for k, sh in enumerate(shards)]}
    line = {"metric": "columnSums nnz/s + achieved HBM GB/s vs roofline, 1e9-nnz CSC at 1/2/4/8 GPUs",
            "value": nnz / (ms_value * 1e-3), "unit": "nnz/s", "n_gpus": G, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_value, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic", "config": sig(cfg), "roofline": sig(roof), "parity": sig(parity)}
    line["cpu_baseline"] = None if args.no_cpu_baseline else cpu_baseline(p, args.kind)
    if args.verbose:
        line["notes"] = GLOSSARY
    print(json.dumps(line), flush=True)


THREADS_CHILD_KEYS = ("threads_d2h_pinned_ms", "threads_blit_pinned_ms", "threads_d2h_pageable_ms", "threads_blit_pageable_ms",
                      "threads_stores_pinned_ms", "threads_none_pinned_ms", "threads_rccl_pinned_ms", "threads_last_enqueue_us",
                      "threads_serial_d2h_pinned_ms", "threads_workers_d2h_pinned_ms", "threads_serial_blit_pinned_ms",
                      "threads_workers_blit_pinned_ms", "threads_rccl_pinned_error", "call_minus_slowest_kernel_us",
                      "kernel_ms_max_over_shards")


def threads_child_figures(args, timeout_s=300):
    """`--parallelism threads` over the same devices as a CHILD process (the ranks of this job have finished their work; a
    child cannot take this process down with it): {threads_*: ...} for the line, or {threads_error: why}."""
    import signal
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK", "ROLE_NAME",
                        "MASTER_ADDR", "MASTER_PORT", "GROUP_WORLD_SIZE", "ROLE_WORLD_SIZE") and not k.startswith("TORCHELASTIC_")}
    cmd = [sys.executable, os.path.abspath(__file__), "--parallelism", "threads", "--gpus", str(args.gpus), "--steps", str(args.steps),
           "--warmup", str(args.warmup), "--workload", args.workload, "--kind", str(args.kind), "--partition", args.partition,
           "--no-cpu-baseline"] + (["--nnz", str(args.nnz)] if args.nnz else [])
    t0 = time.perf_counter()
    try:
        pr = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, start_new_session=True)
        try:
            out, err = pr.communicate(timeout=timeout_s)
        except subprocess.TimeoutExpired:
            os.killpg(pr.pid, signal.SIGKILL)      # (the group this call started, nothing else)
            pr.communicate()
            return {"threads_error": f"the child did not finish in {timeout_s} s"}
    except OSError as e:
        return {"threads_error": f"the child could not be started: {e}"[:160]}
    if pr.returncode != 0:
        return {"threads_error": f"exit {pr.returncode}: {err.decode(errors='replace')[-160:]}"}
    try:
        line = json.loads([ln for ln in out.decode().splitlines() if ln.startswith("{")][-1])
    except (IndexError, ValueError):
        return {"threads_error": "the child printed no line"}
    res = {"threads_value": line["value"], "threads_ms_per_step": line["ms_per_step"],
           "threads_launch": line["config"].get("launch"), "threads_gather": line["config"].get("gather"), "threads_devices_distinct": line["config"].get("devices_distinct"),
           "threads_parity_err": line["parity"]["max_abs_err_over_l1"], "threads_seconds": time.perf_counter() - t0}
    for k in THREADS_CHILD_KEYS:
        if k in line["roofline"]:
            res[k if k.startswith("threads_") else "threads_" + k] = line["roofline"][k]
    return res


def mcsc_overhead_figure(torch, capi, synth, shards=8, shard_nnz=131072, shard_ncol=1024, calls=200):
    """What the HOST adds to the single-process multi-GPU call (rsp_mcsc_column_sums, the path an R session reaches): `shards`
    tiny shards resident on this device -- their kernels take a few microseconds, what is left of a call's wall time IS the
    host (launches, events, wake-ups, waits).  Median wall time of `calls` calls minus the sum of the shards' kernel times
    (each timed alone), in both launch modes; result left in the handle's page-locked vector, gather as the handle chooses."""
    xs, ps = [], []
    for k in range(shards):
        counts = synth.uniform_counts(shard_ncol, shard_nnz, seed=SEED + k, nrow=1_000_000)
        xt = torch.empty(shard_nnz, dtype=torch.float64, device="cuda")
        capi.gen_values_device(xt, SEED, k * shard_nnz, 0)
        xs.append(xt)
        ps.append(torch.from_numpy(synth.offsets_from_counts(counts)).cuda())
    torch.cuda.synchronize()
    h = capi.MultiDeviceCSC.wrap_device(xs, ps, 1_000_000)
    out = h.result_buffer()
    kernels_us = sum(h.shard_kernel_ms(k, reps=20) for k in range(shards)) * 1e3
    fig = {"mcsc8_kernels_us": kernels_us}
    for launch in ("serial", "workers"):
        h.set_launch(launch)
        for _ in range(10):
            h.column_sums(out=out)
        wall, enq = [], []
        for _ in range(calls):
            h.column_sums(out=out)
            st = h.last_call_stamps()
            wall.append(st["call_us"])
            enq.append(max(st["enqueued_us"]))
        wall.sort()
        enq.sort()
        fig[f"mcsc8_{launch}_call_us"] = wall[len(wall) // 2]
        fig[f"mcsc8_{launch}_overhead_us"] = wall[len(wall) // 2] - kernels_us
        fig[f"mcsc8_{launch}_last_enqueue_us"] = enq[len(enq) // 2]
    h.close()
    return fig


def make_communicator(env, counts, displs, recv, shard_ncol):
    """The C-ABI communicator (rsp_comm_*), checked with a trial gatherv of a known pattern.  If it
    cannot be created or delivers wrong data on any rank, EVERY rank switches to the same gatherv
    through torch.distributed (plumbing only, never a compute fallback); config.gather says which."""
    torch, dist, capi = env["torch"], env["dist"], env["capi"]
    rank, world, dev = env["rank"], env["world"], env["dev"]
    uid = torch.zeros(capi.UNIQUE_ID_BYTES, dtype=torch.uint8, device=dev)
    if rank == 0:
        uid.copy_(torch.frombuffer(bytearray(capi.comm_unique_id()), dtype=torch.uint8))
    if world > 1:
        dist.broadcast(uid, 0)

    def all_ok(ok):
        if world == 1:
            return bool(ok)
        flag = torch.tensor([int(ok)], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return bool(int(flag.item()))

    comm = None
    try:
        comm = capi.Comm(bytes(uid.cpu().numpy().tobytes()), world, rank, env["local_rank"])
        ok = True
    except Exception as e:
        print(f"[rank {rank}] rsp_comm_init failed: {e}", file=sys.stderr, flush=True)
        ok = False
    if all_ok(ok):
        ok = True
        try:   # rank r sends r + 1 everywhere in its slice
            probe = torch.full((shard_ncol,), float(rank + 1), dtype=torch.float64, device=dev)
            if recv is not None:
                recv.zero_()
            comm.gatherv(probe, recv, counts, displs, 0, stream=torch.cuda.current_stream())
            torch.cuda.synchronize()
            if rank == 0:
                want = torch.repeat_interleave(
                    torch.arange(1, world + 1, dtype=torch.float64, device=dev),
                    torch.as_tensor([int(c) for c in counts], device=dev))
                ok = bool(torch.equal(recv, want))
        except Exception as e:
            print(f"[rank {rank}] trial gatherv failed: {e}", file=sys.stderr, flush=True)
            ok = False
        if all_ok(ok):
            return comm, False
    if comm is not None:
        comm.close()
    if world == 1:
        raise SystemExit("--force-comm: the C-ABI communicator failed (see stderr)")
    return None, True


def rehearse_comm_init(torch, dist, capi, rank, world, dev_index):
    """The multi-rank bootstrap of the C-ABI communicator between ranks that share a device: the unique
    id travels from rank 0 (here over gloo), every rank calls rsp_comm_init, RCCL's bootstrap connects
    the processes and then has to refuse the duplicate device.  Returns what every rank saw."""
    uid = torch.zeros(capi.UNIQUE_ID_BYTES, dtype=torch.uint8)
    if rank == 0:
        uid.copy_(torch.frombuffer(bytearray(capi.comm_unique_id()), dtype=torch.uint8))
    dist.broadcast(uid, 0)
    try:
        c = capi.Comm(bytes(uid.numpy().tobytes()), world, rank, dev_index)
        c.close()
        seen = "created (unexpected: the ranks share a device)"
    except capi.RspError as e:
        seen = str(e)
    everyone = [None] * world
    dist.all_gather_object(everyone, seen)
    return everyone


def run_sharded_workload(env, name, partition, full, nnz_override=0):
    """One workload through the protocol of `value` on this job's ranks: generate this rank's column range in HBM, K calls back
    to back (kernels, then the call's gatherv, in order on one stream), the kernel / gather split, every column of the gathered
    result against the oracle.  full = the headline workload: also latency, the pipelined figure, planned_shards, direct_gather.
    Returns the measurements on rank 0 (plain numbers), None elsewhere."""
    torch, dist, capi, sharded, args = env["torch"], env["dist"], env["capi"], env["sharded"], env["args"]
    rank, world, dev, dev_index, stat_dev = env["rank"], env["world"], env["dev"], env["dev_index"], env["stat_dev"]
    rehearsal, use_comm = env["rehearsal"], env["use_comm"]

    nrow, ncol, nnz, shape, p = build_offsets(name, nnz_override)
    shard = sharded.make_shard(p, rank, world, balance=partition)
    counts, displs = sharded.gather_layout(shard.bounds)

    # inputs resident in HBM before anything is timed.  A workload that would fit in the 256 MiB
    # Infinity Cache (C2: 80 MB) is timed over a rotation of distinct copies of x (> 400 MB in
    # total, different seeds) so that every step reads from HBM.
    ncopies = max(1, -(-400_000_000 // max(1, 8 * shard.nnz)))
    xs = []
    for k in range(ncopies):
        xk = torch.empty(shard.nnz, dtype=torch.float64, device=dev)
        capi.gen_values_device(xk, SEED + k, shard.x0, args.kind)
        xs.append(xk)
    pt = torch.from_numpy(shard.p_local).to(dev)
    recv = torch.empty(ncol, dtype=torch.float64, device=dev) if (use_comm and rank == 0) else None
    if use_comm and not rehearsal and "comm" not in env:       # once per job, with the first workload's layout for its trial
        env["comm"], env["fell_back"] = make_communicator(env, counts, displs, recv, shard.ncol)
    comm, fell_back = env.get("comm"), env.get("fell_back", False)

    def new_out():
        return torch.empty(max(shard.ncol, 1), dtype=torch.float64, device=dev)[:shard.ncol]

    def new_gather(stream):
        if not use_comm:
            return None
        if comm is not None:
            return sharded.RcclGather(comm, counts, displs, 0, stream=stream)
        if rehearsal:
            return sharded.HostStagedGather(dist, rank, world, counts, displs, 0, stream=stream)
        return sharded.TorchGather(dist, rank, world, counts, displs, 0, stream=stream)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    s_main = env["s_main"]
    # ------------------------------------------------------------------ the timed protocol
    # one call = kernels, then this call's gather, in order on s_main (rcppsparse_amd/sharded.py)
    # rank 0 sums straight into its slice of the gathered result, so its own part of the gatherv is
    # no copy at all (rsp_comm_gatherv skips a slice that is already in place)
    out_main = (recv[int(displs[0]):int(displs[0]) + shard.ncol] if (recv is not None and comm is not None)
                else new_out())
    ws_main = capi.alloc_workspace(shard.ncol, shard.nnz, dev)
    plan = None
    if args.planned and full:
        capi.set_lean(not args.no_lean)
        plan = capi.ColumnSumsPlan(shard.p_local, nnz=shard.nnz, device=dev_index)

    def prepare(xk, out, ws, stream):
        if plan is not None:
            return plan.prepared(xk, pt, out, ws, stream=stream)
        return capi.prepared_column_sums(xk, pt, out, ws, stream=stream)

    launch_main = [prepare(xk, out_main, ws_main, s_main) for xk in xs]
    calls = [0]

    def compute(_shard):
        launch_main[calls[0] % ncopies]()
        calls[0] += 1
        return out_main

    driver = sharded.ShardedColumnSums(shard, compute, new_gather(s_main))
    for _ in range(args.warmup):
        driver.step(recv)
    form_code = {"general": 0, "lean": 2, "columns": 3}
    if plan is None:
        # rsp_column_sums_device plans for itself: every rank waits until ITS entry has settled on a form (outside every timed
        # region: the inspection is ~23 us of device time behind the first call), then warms that form.  (The same number of
        # steps on every rank, whatever its shard holds: a step with a gather is a collective.)
        # (rsp_column_sums_device_settle: from here on every call of this rank takes ONE form -- bit-stable run to run)
        my_form = form_code.get(capi.column_sums_device_settle(pt, shard.nnz, stream=s_main), 0) if shard.nnz > 0 else 0
        for _ in range(args.warmup):
            driver.step(recv)
    else:
        my_form = 0 if plan is None else plan.form
    # Timing events are queue packets of their own: a pair around a call leaves the queue idle for a few
    # microseconds (rocprofv3 kernel trace of C2 with events on every 4th call: 16 us of gaps per 4 calls,
    # profiles/r03_c2.md).  Harmless around a 1.2 ms call, a fifth of a 20 us one.  So:
    #   * large single-GPU workloads (C3): an event pair around the kernels of EVERY timed call;
    #   * small calls without a gather (C2, a C4 shard alone): THREE timed regions, ONE event pair around each,
    #     kernel time per launch = the region's device time / K; the median region is the one reported;
    #   * calls with a gather (N > 1): the timed region carries no events at all (`value` is the clean
    #     back-to-back rate); the kernel / gather split comes from the same K calls issued once more right
    #     after it, with events around every call.
    whole_region = (not use_comm) and shard.nnz < 200_000_000
    per_call_events = not use_comm and not whole_region
    mk = lambda: torch.cuda.Event(enable_timing=True)   # noqa: E731

    def run_calls(with_events):
        evs = [(mk(), mk(), mk()) for _ in range(args.steps)] if with_events else None
        fence()
        t_begin = time.perf_counter()
        for k in range(args.steps):
            if evs is None:
                driver.step(recv)
            else:
                e = evs[k]
                e[0].record(s_main)
                driver.step(recv, on_computed=lambda e=e: e[1].record(s_main))
                e[2].record(s_main)
        fence()
        return time.perf_counter() - t_begin, evs

    regions_ms = None
    if whole_region:
        regions = small_call_regions(torch, s_main, lambda k: driver.step(recv), args.steps, fence)
        elapsed, region_kernel_ms = median_region(regions)
        regions_ms = [w / args.steps * 1e3 for w, _ in regions]
        ktimes = [region_kernel_ms]
        evs = None
    else:
        elapsed, evs = run_calls(per_call_events)
        if use_comm:
            _, evs = run_calls(True)          # the split, outside the timed region
        ktimes = sorted(e[0].elapsed_time(e[1]) for e in evs)
    kernel_ms = sum(ktimes) / len(ktimes)
    gtimes = sorted(e[1].elapsed_time(e[2]) for e in evs) if use_comm else [0.0]
    gather_ms = sum(gtimes) / len(gtimes)

    lat_med = lat_min = 0.0
    pipe = None
    if full:
        # -------------------------------------------------------------- latency of one call
        lat = []
        for _ in range(max(1, args.latency_calls)):
            fence()
            t1 = time.perf_counter()
            driver.step(recv)
            s_main.synchronize()          # rank 0: every slice has arrived; other ranks: their send is done
            lat.append((time.perf_counter() - t1) * 1e3)
        lat.sort()
        lat_med, lat_min = lat[len(lat) // 2], lat[0]

        # -------------------------------------------------------------- pipelined figure (separate key)
        if not args.no_pipelined:
            ncs = max(1, args.compute_streams)
            nbuf = max(ncs, args.gather_buffers - args.gather_buffers % ncs)
            s_computes = [torch.cuda.Stream() for _ in range(ncs)]
            s_comm = torch.cuda.Stream() if use_comm else None
            wss = [capi.alloc_workspace(shard.ncol, shard.nnz, dev) for _ in range(ncs)]
            outs = [new_out() for _ in range(nbuf)]
            prepared = [[[prepare(xk, o, wss[q], s_computes[q]) for xk in xs]
                         for o in outs] for q in range(ncs)]
            launches = [[(lambda n, f=prepared[q][k]: f[n % ncopies]()) for k in range(nbuf)] for q in range(ncs)]
            g = new_gather(s_comm)
            gathers = [(None if g is None else (lambda o=o: g(o, recv))) for o in outs]
            pl = sharded.PipelinedColumnSums(torch, launches, gathers, s_computes, s_comm, nbuf)
            for _ in range(args.warmup):
                pl.step()
            pipe_elapsed = timed_steps(torch, dist, world, stat_dev, fence, args.steps, pl.step)
            pipe = sig({"value": nnz * args.steps / pipe_elapsed, "ms_per_step": pipe_elapsed / args.steps * 1e3,
                        "compute_streams": ncs, "output_buffers": nbuf})
            del prepared, launches, outs, wss

    # ------------------------------------------------------------------ N > 1: two more separate figures
    planned_shards = direct_gather = host_gather = None
    if full and world > 1:
        ctx = {"torch": torch, "dist": dist, "capi": capi, "sharded": sharded, "args": args, "rank": rank, "world": world,
               "dev": dev, "dev_index": dev_index, "stat_dev": stat_dev, "shard": shard, "counts": counts, "displs": displs,
               "xs": xs, "pt": pt, "s_main": s_main, "fence": fence, "nnz": nnz, "ncol": ncol, "p": p, "recv": recv,
               "use_comm": use_comm, "out_main": out_main, "ws_main": ws_main, "new_gather": new_gather}
        if plan is None and not args.no_planned_shards:
            planned_shards = planned_shards_figure(ctx)
        peer_probe_verdict = None
        want_direct = args.direct_gather == "on" or (args.direct_gather == "auto" and rehearsal)
        if args.direct_gather == "auto" and not rehearsal:
            # between different devices: only after throw-away processes have done the same thing and lived (rank 0 asks,
            # everybody hears the answer)
            box = [None]
            gathered = [None] * world
            dist.all_gather_object(gathered, dev_index)
            if rank == 0:
                box = [run_peer_probe(gathered[0], gathered[1])]
            dist.broadcast_object_list(box, src=0)
            peer_probe_verdict = box[0]
            want_direct = peer_probe_verdict == "passed"
        if want_direct:
            direct_gather = direct_gather_figure(ctx)
        elif peer_probe_verdict is not None and rank == 0:
            direct_gather = {"value": None, "note": f"not measured: peer probe: {peer_probe_verdict}"[:160]}
        env["peer_probe"] = peer_probe_verdict
        if args.host_gather != "off":
            host_gather = host_gather_figure(ctx)

    stats = torch.tensor([elapsed, kernel_ms, gather_ms, lat_med], dtype=torch.float64, device=stat_dev)
    # what every rank owned and measured (rank order), so the line shows the whole partition
    mine = torch.tensor([shard.c0, shard.c1, shard.x0, shard.x1, kernel_ms, gather_ms, dev_index, my_form],
                        dtype=torch.float64, device=stat_dev)
    per_rank = [torch.zeros_like(mine) for _ in range(world)]
    if world > 1:
        dist.all_reduce(stats, op=dist.ReduceOp.MAX)
        dist.all_gather(per_rank, mine)
    else:
        per_rank = [mine]
    elapsed, kernel_ms_max, gather_ms_max, lat_med_max = (float(v) for v in stats)
    per_rank = [[float(v) for v in t] for t in per_rank]

    # ------------------------------------------------------------------ parity: every column
    # one more call on copy 0 of x (seed SEED), outside every timed region
    calls[0] = 0
    driver.step(recv)
    fence()
    H = None
    if rank == 0:
        full_result = (recv if use_comm else out_main).cpu().numpy()
        parity = parity_whole_matrix(full_result, p, args.kind)
        if not parity_ok(parity):
            raise SystemExit(f"parity check failed ({name}, partition {partition}): {json.dumps(parity)}")
        gname = None if not use_comm else (sharded.HostStagedGather.name if rehearsal else
                                           sharded.TorchGather.name if fell_back else sharded.RcclGather.name)
        H = {"workload": name, "nrow": nrow, "ncol": ncol, "nnz": nnz, "shape": shape, "partition": partition, "world": world,
             "p": p, "elapsed": elapsed, "regions_ms": regions_ms,
             "kernel_ms": kernel_ms, "kernel_ms_median": ktimes[len(ktimes) // 2], "kernel_ms_min": ktimes[0],
             "kernel_ms_max_over_ranks": kernel_ms_max,
             "kernel_timing": "region" if whole_region else ("per_call_after" if use_comm else "per_call"),
             "gather_ms": gather_ms if use_comm else None, "gather_ms_min": gtimes[0] if use_comm else None,
             "gather_ms_max": gtimes[-1] if use_comm else None, "gather_ms_max_over_ranks": gather_ms_max if use_comm else None,
             # the launch rank 0 timed processed its own shard
             "algo_bytes": 8 * shard.nnz + 4 * (shard.ncol + 1) + 8 * shard.ncol,
             "imbalance": sharded.imbalance(p, shard.bounds), "ncopies": ncopies,
             "shards": [{"rank": r, "device": int(t[6]), "c0": int(t[0]), "c1": int(t[1]), "x0": int(t[2]), "x1": int(t[3]),
                         "kernel_ms": t[4], "gather_ms": t[5] if use_comm else None, "form": PLAN_FORMS[int(t[7])]}
                        for r, t in enumerate(per_rank)],
             "form": PLAN_FORMS[int(my_form)], "auto_plan": capi.debug_get("auto_plan"),
             "parity": parity, "lat_med": lat_med, "lat_min": lat_min, "lat_med_max": lat_med_max,
             "pipelined": pipe, "planned_shards": planned_shards, "direct_gather": direct_gather, "host_gather": host_gather,
             "peer_probe": env.get("peer_probe"),
             "plan": (None if plan is None else
                      {"form": PLAN_FORMS[plan.form], "snapped": plan.snapped, "plan_ms": plan.inspect_ms, "chunks": plan.nchunks,
                       "entries_per_chunk": plan.chunk_elems, "max_skip": plan.max_skip,
                       "kernel": ("colsums_columns_kernel" if plan.columns else "colsums_lean_kernel" if plan.lean else
                                  "colsums_chunks_kernel<PLANNED>" if plan.snapped else "colsums_chunks_kernel+fixup")}),
             "gather_name": gname, "fell_back": bool(fell_back), "x0_for_ceiling": xs[0] if full else None}
    if plan is not None:
        plan.close()
    if not full:
        del xs, launch_main, out_main, ws_main, pt, recv
        torch.cuda.empty_cache()
    return H


def sharded_summary(H, steps):
    """The flat scalars of one `also_sharded` workload (N > 1): the protocol of `value` on another matrix / partition."""
    v = H["nnz"] * steps / H["elapsed"]
    return sig({"value": v, "ms_per_step": H["elapsed"] / steps * 1e3, "imbalance": H["imbalance"],
                "kernel_ms_max": H["kernel_ms_max_over_ranks"], "gather_ms_max": H["gather_ms_max_over_ranks"],
                "kernel_ms_by_rank": [s["kernel_ms"] for s in H["shards"]],
                "nnz_by_rank": [s["x1"] - s["x0"] for s in H["shards"]],
                "parity_err": H["parity"]["max_abs_err_over_l1"], "bad_columns": H["parity"]["columns_out_of_tolerance"]})


def assemble_line(args, H, extras, devices=1, rehearsal=False, comm_rehearsal=None):
    """The JSON line from the measurements (pure: tests/test_bench_line.py builds one for 8 ranks without a GPU).
    H: what run_sharded_workload returned on rank 0 for the headline workload; extras: traffic, read ceiling, `also`
    records, `also_sharded` summaries and the CPU baseline, each optional."""
    world, steps = H["world"], args.steps
    use_comm = H["gather_name"] is not None
    achieved = H["algo_bytes"] / (H["kernel_ms"] * 1e-3) / 1e9
    ms_per_step = H["elapsed"] / steps * 1e3
    plan = H["plan"]
    roof = {
        "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
        "traffic": None, "traffic_over_algorithmic": None, "traffic_measured_in_run": False, "traffic_file": None,
        "kernel": ({"lean": "colsums_lean_kernel (the entry's own plan)", "columns": "colsums_columns_kernel (the entry's own plan)"}
                   .get(H.get("form"), "colsums_chunks_kernel+fixup") if plan is None else plan["kernel"]),
        "kernel_ms": H["kernel_ms"], "kernel_ms_median": H["kernel_ms_median"], "kernel_ms_min": H["kernel_ms_min"],
        "kernel_timing": H["kernel_timing"], "kernel_ms_max_over_ranks": H["kernel_ms_max_over_ranks"],
        "gather_ms": H["gather_ms"], "gather_ms_min": H["gather_ms_min"], "gather_ms_max": H["gather_ms_max"],
        "gather_ms_max_over_ranks": H["gather_ms_max_over_ranks"],
        "algorithmic_bytes_per_launch": H["algo_bytes"],
    }
    tr = extras.get("traffic")
    if tr is not None:
        roof["traffic"] = tr["bytes"]
        roof["traffic_over_algorithmic"] = None if tr["bytes"] is None else tr["bytes"] / H["algo_bytes"]
        roof["traffic_measured_in_run"] = bool(tr.get("in_run"))
        roof["traffic_file"] = tr.get("file")
        for k in ("read_bytes", "write_bytes", "seconds", "not_measured", "kernels"):
            if tr.get(k) is not None:
                roof["traffic_" + k] = tr[k]
    rc = extras.get("read_ceiling")
    if rc is not None:
        roof["read_ceiling_GBps"] = rc["GBps"]
        roof["read_ceiling_ms"] = rc["ms_per_launch"]
        roof["read_ceiling_reps"] = rc["reps"]
        roof["frac_of_ceiling"] = achieved / rc["GBps"]
    also = extras.get("also")
    if also:
        for r in also:
            k = "also_" + key_of(r["workload"])
            roof[k + "_frac"] = r["frac"]
            roof[k + "_kernel_ms"] = r["kernel_ms"]
            roof[k + "_ms_per_call"] = r["ms_per_call"]
            roof[k + "_parity_err"] = r["parity_err"]
            if r.get("traffic") is not None:
                roof[k + ("_traffic_x" if r.get("traffic_in_run") else "_traffic_recorded_x")] = r["traffic"] / r["algo_bytes"]
    also_sharded = extras.get("also_sharded")
    if also_sharded:
        for name, s in also_sharded.items():
            k = "also_" + key_of(name)
            if "value" in s:
                for f in ("value", "ms_per_step", "imbalance", "kernel_ms_max", "gather_ms_max", "parity_err"):
                    roof[f"{k}_{f}"] = s[f]
    th = extras.get("threads")
    if th:
        roof.update(th)
    if extras.get("mcsc"):
        roof.update(extras["mcsc"])
    for name in ("planned_shards", "direct_gather", "host_gather", "pipelined"):
        fig = H.get(name)
        if fig and fig.get("value") is not None:
            roof[name + "_value"] = fig["value"]
            roof[name + "_ms_per_step"] = fig["ms_per_step"]
    rccl = extras.get("rccl") or {}
    cfg = {
        "workload": f"{H['workload']}: {H['nrow']}x{H['ncol']} CSC dgCMatrix, nnz={H['nnz']}, {H['shape']} nnz/column, "
                    f"values kind {args.kind}, seed {SEED}",
        "parallelism": "single" if world == 1 else ("rehearsal" if rehearsal else "ranges+rccl"),
        "devices": devices,
        "rendezvous": args.rendezvous if world > 1 else None,
        "partition": H["partition"],
        "shard_imbalance_max_over_mean": H["imbalance"],
        "chunk_rows": args.chunk_rows,
        "auto_plan": H.get("auto_plan", 1),
        "x_copies_rotated": H["ncopies"],
        "gather": H["gather_name"],
        "gather_fell_back_to_torch_distributed": H["fell_back"],
        "regions_ms": H["regions_ms"],
        "host_stall_suspected": bool(not use_comm and ms_per_step > 1.5 * H["kernel_ms"]),
        "planned": None if plan is None else {k: v for k, v in plan.items() if k != "kernel"},
        "shards": [{"rank": s["rank"], "device": s["device"], "c0": s["c0"], "c1": s["c1"], "x0": s["x0"], "x1": s["x1"],
                    "nnz": s["x1"] - s["x0"], "kernel_ms": s["kernel_ms"], "gather_ms": s["gather_ms"],
                    "form": s.get("form", "general")} for s in H["shards"]],
    }
    if rccl:                                   # (N > 1 or --force-comm: which RCCL really ran)
        cfg["rccl_version"], cfg["rccl_library"] = rccl.get("version"), rccl.get("library")
    if world > 1:
        cfg["peer_probe"] = H.get("peer_probe")
    if comm_rehearsal is not None:
        refused = [("ncclCommInitRank" in s and "error 5" in s) for s in comm_rehearsal]
        cfg["comm_refused_on_ranks"] = int(sum(refused))
        odd = [s for s, r in zip(comm_rehearsal, refused) if not r]
        if odd:
            cfg["comm_rehearsal_unexpected"] = odd[0][:160]
    line = {
        "metric": "columnSums nnz/s + achieved HBM GB/s vs roofline, 1e9-nnz CSC at 1/2/4/8 GPUs",
        "value": H["nnz"] * steps / H["elapsed"], "unit": "nnz/s", "n_gpus": world, "steps": steps,
        "warmup": args.warmup, "ms_per_step": ms_per_step,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": sig(cfg),
        "latency_ms_per_call": H["lat_med"],
        "latency": sig({"ms_median": H["lat_med"], "ms_min": H["lat_min"], "ms_median_max_over_ranks": H["lat_med_max"],
                        "calls": max(1, args.latency_calls)}),
        "pipelined": H["pipelined"],
        "planned_shards": H["planned_shards"],
        "direct_gather": H["direct_gather"],
        "roofline": sig(roof),
        "parity": sig(H["parity"]),
    }
    if world > 1:
        line["host_gather"] = H.get("host_gather")
    if also:
        line["also"] = also
        line["also_seconds"] = sig(extras.get("also_seconds"))
    if also_sharded:
        line["also_sharded"] = also_sharded
    if "cpu_baseline" in extras:
        line["cpu_baseline"] = extras["cpu_baseline"]
    if args.verbose:
        line["notes"] = GLOSSARY
    return line


def main(argv=None):
    args = parse_args(argv)
    if args.explain:
        for k, v in GLOSSARY.items():
            print(f"{k}\n    {v}")
        return
    if args.peer_probe_writer:
        sys.exit(peer_probe_writer(args))
    if args.peer_probe:
        sys.exit(peer_probe(args))
    if args.parallelism == "threads":
        return main_threads_rowsums(args) if args.op == "rowsums" else main_threads(args)
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(relaunch_under_torchrun(args))

    if args.traffic_child:
        return traffic_child(args)
    if args.op == "rowsums":
        return main_rowsums(args)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    import torch
    import torch.distributed as dist
    from rcppsparse_amd import capi, sharded

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    rehearsal = args.rendezvous == "gloo"
    if rehearsal and args.force_comm:
        raise SystemExit("--force-comm needs an RCCL communicator; --rendezvous gloo never creates one")
    dev_index = local_rank % torch.cuda.device_count() if rehearsal else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    # small host-visible statistics travel on the rendezvous backend: device tensors over RCCL,
    # host tensors over gloo
    stat_dev = torch.device("cpu") if rehearsal else dev
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    capi.load()
    capi.set_tuning(args.chunk_rows)

    comm_rehearsal = None
    if rehearsal and args.try_comm and world > 1:
        comm_rehearsal = rehearse_comm_init(torch, dist, capi, rank, world, dev_index)
    # Streams come from torch's pool (non-blocking): nothing here runs on the legacy null stream,
    # which would implicitly synchronise with any blocking stream a library creates.
    torch.cuda.synchronize()
    s_main = torch.cuda.Stream()
    torch.cuda.set_stream(s_main)
    env = {"torch": torch, "dist": dist, "capi": capi, "sharded": sharded, "args": args, "rank": rank, "world": world,
           "local_rank": local_rank, "dev": dev, "dev_index": dev_index, "stat_dev": stat_dev, "rehearsal": rehearsal,
           "use_comm": world > 1 or args.force_comm, "s_main": s_main}

    H = run_sharded_workload(env, args.workload, args.partition, full=True, nnz_override=args.nnz)
    extras = {}
    default_c3 = args.workload == "c3" and not args.planned and not args.nnz
    if rank == 0 and world == 1:
        traffic_key = args.workload + ("planned" if H["plan"] is not None and H["plan"]["snapped"] else "")
        recorded, src = traffic_from_profiles(traffic_key)
        extras["traffic"] = {"bytes": recorded, "in_run": False, "file": src}
        want_pass = args.traffic_pass == "on" or (args.traffic_pass == "auto" and default_c3)
        if want_pass:
            # counters cannot be read inside a timed run: two short child runs of the same workload under rocprofv3, now,
            # on this device (their launches are the plan-free call's: the kernels `roofline` is about)
            t_pass = time.perf_counter()
            measured, how = traffic_measured_now(args.workload, args.kind)
            if measured is not None:
                extras["traffic"] = dict(how, bytes=measured, in_run=True, file=None, seconds=time.perf_counter() - t_pass)
            else:
                extras["traffic"]["not_measured"] = str(how)[:160]
        if args.ceiling_reps > 0:
            # the same x (copy 0), the same device, the same run; after every timed region
            x0 = H["x0_for_ceiling"]
            ms = capi.read_ceiling_device(x0, reps=args.ceiling_reps)
            extras["read_ceiling"] = {"GBps": 8 * x0.numel() / (ms * 1e-3) / 1e9, "ms_per_launch": ms,
                                      "reps": args.ceiling_reps}
        if not args.no_also:
            specs = (ALSO_AUTO if default_c3 else ()) if args.also == "auto" else \
                tuple(t for t in args.also.split(",") if t)
            if specs:
                H["x0_for_ceiling"] = None
                t_also = time.perf_counter()
                extras["also"] = [also_record(torch, capi, spec, args, dev, dev_index, s_main,
                                              traffic_now=want_pass and spec in ALSO_TRAFFIC_NOW) for spec in specs]
                extras["also_seconds"] = time.perf_counter() - t_also
            if default_c3:
                # (round 6) the host's share of the single-process multi-GPU call, measured by this very run: eight tiny shards
                try:
                    from rcppsparse_amd import synth
                    extras["mcsc"] = mcsc_overhead_figure(torch, capi, synth)
                except Exception as e:   # noqa: BLE001  (an extra figure never costs the line)
                    extras["mcsc"] = {"mcsc8_error": str(e)[:100]}
    if world > 1:
        specs = (ALSO_SHARDED_AUTO if default_c3 and args.partition == "nnz" else ()) if args.also_sharded == "auto" else \
            tuple(t for t in args.also_sharded.split(",") if t and t != "none")
        if specs:
            if H is not None:
                H["x0_for_ceiling"] = None
            torch.cuda.empty_cache()
            summaries = {}
            for spec in specs:
                name, _, part = spec.partition(":")
                Hs = run_sharded_workload(env, name, part or "nnz", full=False)
                if Hs is not None:
                    summaries[key_of(spec)] = sharded_summary(Hs, args.steps)
            if rank == 0:
                extras["also_sharded"] = summaries
    if rank == 0 and not args.no_cpu_baseline:
        # rank 0 at EVERY N (the other ranks wait at the barrier below): the reference loop on this box's host cores, same run
        extras["cpu_baseline"] = cpu_baseline(H["p"], args.kind)
    elif rank == 0:
        extras["cpu_baseline"] = None
    comm = env.get("comm")
    if comm is not None:
        torch.cuda.synchronize()
        comm.close()
    if rank == 0 and (world > 1 or args.force_comm):
        try:
            extras["rccl"] = capi.rccl_info()
        except Exception as e:   # noqa: BLE001
            extras["rccl"] = {"error": str(e)[:80]}
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        want_threads = args.threads_figure == "on" or (args.threads_figure == "auto" and world > 1 and not rehearsal
                                                       and args.workload == "c3" and not args.planned)
        if want_threads:
            # the ranks are done (the process group is gone); what an R session -- ONE process -- gets out of the same devices
            H["x0_for_ceiling"] = None
            torch.cuda.empty_cache()
            extras["threads"] = threads_child_figures(args)
        line = assemble_line(args, H, extras, devices=torch.cuda.device_count(), rehearsal=rehearsal,
                             comm_rehearsal=comm_rehearsal)
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
