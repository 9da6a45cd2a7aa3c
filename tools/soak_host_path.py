#!/usr/bin/env python3
"""Soak of the host-side entries (round 5): rsp_column_sums_host (one stream and grow-only buffers per device kept between
calls, rsp_release_cached, the cap RSP_ONE_SHOT_KEEP_MB -- set small here, so that calls on both sides of it alternate),
rsp_csc_upload (p[] inspected on the device from 65536 columns on, on host threads below) with column sums / means behind
the handle, and rsp_column_sums_host_multi on one device listed twice.  Sizes from nothing to ~2e7 entries and 1 to ~2e6
columns in random order -- big after small after big --, random column structure; every result against the oracle within
1e-12 of the column's 1-norm (a handle's lean form: the reference's bits).  Half of the time a SECOND THREAD makes one-shot
calls of its own at the same time (the entry is documented as safe from several threads: a mutex around the kept buffers).

    python tools/soak_host_path.py [seconds] [seed]          (on the GPU box)"""
import json
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["RCPPSPARSE_REQUIRE_GPU"] = "1"
os.environ.setdefault("RSP_ONE_SHOT_KEEP_MB", "48")

import numpy as np   # noqa: E402

import oracle        # noqa: E402
from rcppsparse_amd import capi, synth   # noqa: E402


def random_matrix(rng):
    shape = int(rng.integers(0, 7))
    if shape == 0:      # tiny (the R examples' sizes)
        ncol, mean = int(rng.integers(1, 40)), float(rng.uniform(0, 6))
    elif shape == 1:    # many short columns (lean form behind a handle)
        ncol, mean = int(rng.choice([65_535, 65_536, 65_537, 200_000, 1_000_000, 2_000_000])), float(rng.uniform(0.5, 12))
    elif shape == 2:    # few long columns (columns form)
        ncol, mean = int(rng.integers(1, 900)), float(rng.uniform(2_000, 40_000))
    elif shape == 3:    # around the threshold of the device inspection, any lengths
        ncol, mean = int(rng.integers(60_000, 72_000)), float(rng.uniform(0, 150))
    elif shape == 4:    # medium
        ncol, mean = int(rng.integers(100, 50_000)), float(rng.uniform(0, 400))
    elif shape == 5:    # empty matrix / empty columns only
        ncol, mean = int(rng.integers(1, 100_000)), 0.0
    else:               # one giant column among short ones
        ncol, mean = int(rng.integers(2, 300_000)), float(rng.uniform(0, 8))
    mean = min(mean, 2.0e7 / ncol)
    fam = int(rng.integers(0, 3))
    if mean == 0.0:
        counts = np.zeros(ncol, dtype=np.int64)
    elif fam == 0:
        counts = rng.poisson(mean, ncol).astype(np.int64)
    elif fam == 1:
        counts = rng.integers(0, int(2 * mean) + 2, ncol).astype(np.int64)
    else:
        counts = np.full(ncol, int(mean), dtype=np.int64)
        counts[rng.random(ncol) < 0.3] = 0
    if shape == 6:
        counts[int(rng.integers(0, ncol))] += int(rng.integers(100_000, 3_000_000))
    p = synth.offsets_from_counts(counts)
    nnz = int(p[-1])
    x = synth.gen_values(nnz, seed=int(rng.integers(1, 1 << 30)), kind=int(rng.integers(0, 2)))
    return x, p, ncol, nnz


def check(got, x, p, what, exact=False):
    ref = oracle.column_sums(x, p)
    scale = oracle.column_abs_sums(x, p)
    if got.shape != ref.shape or not np.all(np.abs(got - ref) <= 1e-12 * scale):
        bad = np.flatnonzero(~(np.abs(got - ref) <= 1e-12 * scale))[:3] if got.shape == ref.shape else []
        raise AssertionError(f"{what}: shape {got.shape} / {ref.shape}, first bad columns {list(bad)}")
    if exact and got.tobytes() != ref.tobytes():
        raise AssertionError(f"{what}: the lean form must return the reference's bits")
    empty = scale == 0
    if np.any(got[empty] != 0.0) or np.any(np.signbit(got[empty])):
        raise AssertionError(f"{what}: an empty column must be +0.0")


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    capi.load()
    stats = {"one_shot": 0, "one_shot_threaded": 0, "handle": 0, "multi": 0, "released": 0, "forms": {}}
    errors = []
    stop = threading.Event()
    busy = threading.Event()

    def second_thread():
        r2 = np.random.default_rng(seed + 999)
        while not stop.is_set():
            if not busy.wait(0.05):
                continue
            try:
                x, p, ncol, nnz = random_matrix(r2)
                if nnz > 4_000_000:
                    continue
                check(capi.column_sums_host(x, p), x, p, f"second thread one-shot {ncol} x {nnz}")
                stats["one_shot_threaded"] += 1
            except Exception as e:   # noqa: BLE001
                errors.append(repr(e))
                return

    th = threading.Thread(target=second_thread, daemon=True)
    th.start()
    t0 = last = time.time()
    step = 0
    try:
        while time.time() - t0 < seconds and not errors:
            x, p, ncol, nnz = random_matrix(rng)
            if rng.random() < 0.5:
                busy.set()
            else:
                busy.clear()
            what = int(rng.integers(0, 10))
            tag = f"step {step}: {ncol} columns, {nnz} entries"
            if what < 6:
                check(capi.column_sums_host(x, p), x, p, tag + " one-shot")
                stats["one_shot"] += 1
            elif what < 9:
                h = capi.DeviceCSC(x, p, (max(1, ncol), ncol))
                form = h.column_form()
                stats["forms"][form] = stats["forms"].get(form, 0) + 1
                check(h.column_sums(), x, p, tag + f" handle ({form})", exact=(form == "lean"))
                nrow = int(rng.integers(1, 1_000_000))
                h2 = capi.DeviceCSC(x, p, (nrow, ncol))
                means = h2.column_means()
                ref = oracle.column_sums(x, p) / nrow
                scale = oracle.column_abs_sums(x, p) / nrow
                if not np.all(np.abs(means - ref) <= 2e-12 * scale + 0.0):
                    raise AssertionError(tag + " handle means")
                h.close()
                h2.close()
                stats["handle"] += 1
            else:
                if nnz <= 6_000_000:
                    check(capi.column_sums_host_multi(x, p, devices=[0, 0]), x, p, tag + " host_multi on device 0 twice")
                    stats["multi"] += 1
            if rng.random() < 0.05:
                capi.release_cached()
                stats["released"] += 1
            step += 1
            if time.time() - last > 60:
                last = time.time()
                print(f"[soak_host_path] {int(last - t0)} s: {step} steps", file=sys.stderr, flush=True)
    finally:
        stop.set()
        busy.set()
        th.join(timeout=30)
    if errors:
        print(json.dumps({"FAILED": errors[:3], "steps": step}))
        sys.exit(1)
    print(json.dumps({"seconds": round(time.time() - t0, 1), "seed": seed, "steps": step, **stats, "mismatches": 0,
                      "keep_mb": os.environ["RSP_ONE_SHOT_KEEP_MB"]}))


if __name__ == "__main__":
    main()
