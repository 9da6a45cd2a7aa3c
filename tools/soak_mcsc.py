#!/usr/bin/env python3
"""Time-boxed soak of the single-process multi-GPU handle (rsp_mcsc_*, round 6): random matrices (short, long, Zipf, runs of
empty columns, more shards than columns with entries), 1..8 shards on this box's one device, every launch mode x gather in
random order on the SAME handle, sums and means, the page-locked destination and a pageable one, handles created and closed
all the time (worker threads come and go).  Every result against the oracle (1e-12 * column 1-norm) and, bit for bit, against
the first result of the same handle: every combination must return the bits of the per-shard device calls.

    python tools/soak_mcsc.py [seconds] [seed]
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["RCPPSPARSE_REQUIRE_GPU"] = "1"

import numpy as np   # noqa: E402
import torch         # noqa: E402,F401

import oracle        # noqa: E402
from rcppsparse_amd import capi, synth   # noqa: E402


def random_matrix(rng):
    fam = int(rng.integers(0, 5))
    if fam == 0:      # short columns (lean)
        ncol = int(rng.integers(2_000, 300_000))
        counts = np.minimum(rng.poisson(rng.uniform(2, 30), size=ncol), 64).astype(np.int64)
    elif fam == 1:    # long similar columns (columns form)
        ncol = int(rng.integers(8, 600))
        counts = rng.integers(2_500, 9_000, size=ncol).astype(np.int64)
    elif fam == 2:    # Zipf (general kernels)
        ncol = int(rng.integers(500, 40_000))
        counts = synth.zipf_counts(ncol, int(rng.integers(200_000, 3_000_000)), seed=int(rng.integers(1, 1 << 30)), nrow=500_000)
    elif fam == 3:    # runs of empty columns
        ncol = int(rng.integers(1_000, 100_000))
        counts = np.where(rng.random(ncol) < rng.uniform(0.02, 0.5), rng.integers(1, 400, size=ncol), 0).astype(np.int64)
    else:             # tiny: fewer columns with entries than shards
        ncol = int(rng.integers(1, 12))
        counts = np.where(rng.random(ncol) < 0.4, rng.integers(1, 50, size=ncol), 0).astype(np.int64)
    p = synth.offsets_from_counts(counts)
    return p, synth.gen_values(int(p[-1]), seed=int(rng.integers(1, 1 << 30)), kind=int(rng.integers(0, 2))), fam


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    capi.load()
    t0 = last_note = time.time()
    handles = calls = row_handles = 0
    combos = {}
    while time.time() - t0 < seconds:
        p, x, fam = random_matrix(rng)
        ncol, nrow = len(p) - 1, 500_000
        G = int(rng.choice([1, 2, 3, 4, 8]))
        ref, scale = oracle.column_sums(x, p), oracle.column_abs_sums(x, p)
        h = capi.MultiDeviceCSC(x, p, (nrow, ncol), devices=[0] * G)
        handles += 1
        first = None
        pageable = np.empty(ncol)
        for _ in range(int(rng.integers(2, 9))):
            launch = str(rng.choice(["serial", "workers"]))
            gather = str(rng.choice(["d2h", "blit", "stores"] + (["rccl"] if G == 1 else [])))
            h.set_launch(launch)
            h.set_gather(gather)
            means = rng.random() < 0.25
            if means:
                got = h.column_means()
                want_bits = None if first is None else (first / nrow).tobytes()
                check = got * nrow
            else:
                dest = pageable if rng.random() < 0.5 else h.result_buffer()
                got = np.array(h.column_sums(out=dest), copy=True)
                if first is None:
                    first = got
                want_bits = first.tobytes()
                check = got
            calls += 1
            combos[f"{launch}/{gather}"] = combos.get(f"{launch}/{gather}", 0) + 1
            bad = ~(np.abs(check - ref) <= 1e-12 * scale * (1 if not means else 4))
            if bad.any() or (want_bits is not None and got.tobytes() != want_bits):
                c = int(np.flatnonzero(bad)[0]) if bad.any() else -1
                print(json.dumps({"FAILED": True, "family": fam, "shards": G, "launch": launch, "gather": gather, "means": bool(means),
                                  "column": c, "bits_differ": bool(want_bits is not None and got.tobytes() != want_bits)}))
                sys.exit(1)
        h.close()
        # every fourth matrix also by row (round 6: the shards' partial vectors are added on the devices): against the scatter
        # loop's sums, bit-stable, means = sums / ncol, and the SAME bits as the shards' own row sums added in shard order
        if handles % 4 == 0 and len(x) > 0:
            nr = int(rng.choice([7, 20_000, 70_000, 300_000]))
            ii = np.sort(rng.integers(0, nr, size=len(x)).astype(np.int32))   # (any rows: the regrouping forms do not need them ascending per column)
            hr = capi.MultiDeviceCSC(x, p, (nr, ncol), devices=[0] * G, i=ii)
            rs, rs2, rm = hr.row_sums(), hr.row_sums(), hr.row_means()
            hr.close()
            rref = np.bincount(ii, weights=x, minlength=nr)
            rscale = np.bincount(ii, weights=np.abs(x), minlength=nr)
            bounds = capi.partition_columns(p, G)
            blocked = None
            for k in range(G):
                c0, c1 = int(bounds[k]), int(bounds[k + 1])
                if c1 == c0:
                    continue
                hk = capi.DeviceCSC(x[p[c0]:p[c1]], capi.rebase_offsets(p, c0, c1), (nr, c1 - c0), i=ii[p[c0]:p[c1]])
                part = hk.row_sums()
                hk.close()
                blocked = part if blocked is None else blocked + part
            rows_ok = (np.all(np.abs(rs - rref) <= 1e-12 * rscale) and rs.tobytes() == rs2.tobytes()
                       and rm.tobytes() == (rs / ncol).tobytes() and (blocked is None or rs.tobytes() == (blocked + 0.0).tobytes()))
            row_handles += 1
            if not rows_ok:
                print(json.dumps({"FAILED": "row sums", "family": fam, "shards": G, "nrow": nr}))
                sys.exit(1)
        if time.time() - last_note > 60:
            last_note = time.time()
            print(f"[soak_mcsc] {int(last_note - t0)} s: {handles} handles, {calls} calls", file=sys.stderr, flush=True)
    threads = len(os.listdir(f"/proc/{os.getpid()}/task"))
    print(json.dumps({"seconds": round(time.time() - t0, 1), "seed": seed, "handles": handles, "calls_checked": calls, "row_sum_handles_checked": row_handles,
                      "launch_gather_combinations": combos, "threads_alive_at_the_end": threads, "mismatches": 0}))


if __name__ == "__main__":
    main()
