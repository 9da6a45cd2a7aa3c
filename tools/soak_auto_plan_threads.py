#!/usr/bin/env python3
"""The plan-free device entry from SEVERAL THREADS at once (rsp_column_sums_device plans for itself, capi.hip auto_enqueue;
the header's thread-safety rules: the entry may be called from any number of threads, each on a stream of its own).
Three matrices of different shapes (short columns: lean form; few long columns: columns form; mixed lengths: the
general kernels) live once in HBM and are SHARED -- the threads' calls meet on the same keys of the library's plan cache --
while every thread has a stream, an output vector and a workspace of its own.  Every call's result is compared with the
oracle.  The main thread calls rsp_release_cached every half second under the running calls (images and entries go away
and come back) and, every few seconds, stops the threads, writes new offsets into one matrix IN PLACE and lets them go on.

    python tools/soak_auto_plan_threads.py [seconds] [threads] [seed]          (on the GPU box)"""
import json
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["RCPPSPARSE_REQUIRE_GPU"] = "1"

import numpy as np   # noqa: E402
import torch         # noqa: E402

import oracle        # noqa: E402
from rcppsparse_amd import capi, synth   # noqa: E402


def offsets(rng, ncol, nnz, family):
    if family == "short":
        c = rng.poisson(nnz / ncol, ncol).astype(np.int64)
        c = np.minimum(c, 60)
    elif family == "long":
        m = nnz // ncol
        c = rng.integers(int(0.8 * m), int(1.2 * m), ncol).astype(np.int64)
    else:
        c = rng.multinomial(nnz, rng.dirichlet(np.full(ncol, 0.5))).astype(np.int64)
    diff = nnz - int(c.sum())          # fit to exactly nnz entries (the buffers keep their sizes)
    k = 0
    while diff != 0:
        j = k % ncol
        if diff > 0 and (family != "short" or c[j] < 60):
            c[j] += 1
            diff -= 1
        elif diff < 0 and c[j] > 0:
            c[j] -= 1
            diff += 1
        k += 1
    return synth.offsets_from_counts(c)


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    nthreads = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    seed = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    rng = np.random.default_rng(seed)
    capi.load()
    capi.set_auto_plan(True)
    mats = []
    for ncol, nnz, family in ((150_000, 1_500_000, "short"), (700, 2_800_000, "long"), (40_000, 2_000_000, "mixed")):
        x = synth.gen_values(nnz, seed=seed + len(mats), kind=0)
        p = offsets(rng, ncol, nnz, family)
        mats.append({"ncol": ncol, "nnz": nnz, "family": family, "x": x, "p": p,
                     "xt": torch.from_numpy(x).cuda(), "pt": torch.from_numpy(p).cuda(),
                     "ref": oracle.column_sums(x, p), "scale": oracle.column_abs_sums(x, p)})
    go = threading.Event()
    go.set()
    stop = threading.Event()
    idle = [threading.Event() for _ in range(nthreads)]
    errors, calls, forms = [], [0] * nthreads, [{} for _ in range(nthreads)]

    def worker(t):
        r = np.random.default_rng(seed * 100 + t)
        stream = torch.cuda.Stream()
        outs = [torch.empty(m["ncol"], dtype=torch.float64, device="cuda") for m in mats]
        wss = [capi.alloc_workspace(m["ncol"], m["nnz"]) for m in mats]
        try:
            while not stop.is_set():
                if not go.is_set():
                    idle[t].set()
                    go.wait()
                    idle[t].clear()
                k = int(r.integers(0, len(mats)))
                m = mats[k]
                burst = int(r.integers(1, 6))
                with torch.cuda.stream(stream):
                    outs[k].fill_(-7.0)
                    for _ in range(burst):
                        capi.column_sums_device(m["xt"], m["pt"], outs[k], wss[k], stream=stream)
                stream.synchronize()
                got = outs[k].cpu().numpy()
                if not np.all(np.abs(got - m["ref"]) <= 1e-12 * m["scale"]):
                    c = int(np.flatnonzero(~(np.abs(got - m["ref"]) <= 1e-12 * m["scale"]))[0])
                    errors.append(f"thread {t}, matrix {m['family']}: column {c} got {got[c]!r} ref {m['ref'][c]!r}")
                    return
                calls[t] += burst
                f = capi.column_sums_device_form(m["pt"], m["nnz"])
                forms[t][f] = forms[t].get(f, 0) + 1
        except Exception as e:   # noqa: BLE001
            errors.append(f"thread {t}: {e!r}")
        finally:
            idle[t].set()

    threads = [threading.Thread(target=worker, args=(t,), daemon=True) for t in range(nthreads)]
    for th in threads:
        th.start()
    t0 = last = last_mut = time.time()
    released = mutated = 0
    while time.time() - t0 < seconds and not errors:
        time.sleep(0.5)
        capi.release_cached()
        released += 1
        if time.time() - last_mut > 4.0:
            go.clear()                                   # the threads finish their calls and wait
            for ev in idle:
                ev.wait(timeout=60)
            torch.cuda.synchronize()
            m = mats[int(rng.integers(0, len(mats)))]
            m["p"] = offsets(rng, m["ncol"], m["nnz"], m["family"])
            m["pt"].copy_(torch.from_numpy(m["p"]))     # same address, same sizes, new offsets
            torch.cuda.synchronize()
            m["ref"], m["scale"] = oracle.column_sums(m["x"], m["p"]), oracle.column_abs_sums(m["x"], m["p"])
            mutated += 1
            last_mut = time.time()
            go.set()
        if time.time() - last > 60:
            last = time.time()
            print(f"[soak_auto_plan_threads] {int(last - t0)} s: {sum(calls)} calls", file=sys.stderr, flush=True)
    stop.set()
    go.set()
    for th in threads:
        th.join(timeout=60)
    if errors:
        print(json.dumps({"FAILED": errors[:3], "calls": sum(calls)}))
        sys.exit(1)
    merged = {}
    for f in forms:
        for k, v in f.items():
            merged[k] = merged.get(k, 0) + v
    print(json.dumps({"seconds": round(time.time() - t0, 1), "threads": nthreads, "seed": seed, "calls_checked": sum(calls),
                      "per_thread": calls, "form_after_burst": merged, "release_cached_under_load": released,
                      "offsets_rewritten_in_place": mutated, "mismatches": 0}))


if __name__ == "__main__":
    main()
