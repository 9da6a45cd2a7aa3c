#!/usr/bin/env python3
"""Soak of the device-side inspector (rsp_column_sums_plan_create_device, csrc/inspect_device.hip): random offset arrays
-- short columns, long columns, runs of empty columns, columns on chunk edges, Zipf lengths, single giants -- each planned
twice, from a host copy (inspect.hpp) and from the offsets in HBM; the two plans must agree in form, sizes, max_skip and,
BIT FOR BIT, in the image the device holds (snapped records / lean headers + 16-bit offsets), and the sums of both must
pass the parity bar against the oracle (the lean form: the reference's bits).
    python tools/soak_device_plan.py [seconds] [seed]          (on the GPU box)"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["RCPPSPARSE_REQUIRE_GPU"] = "1"

import numpy as np   # noqa: E402
import torch         # noqa: E402

import oracle        # noqa: E402
from rcppsparse_amd import capi, synth   # noqa: E402


def random_counts(rng):
    parts = []
    for _ in range(int(rng.integers(1, 5))):
        n = int(rng.integers(1, 60_000))
        kind = int(rng.integers(0, 9))
        if kind == 0:
            parts.append(rng.poisson(rng.integers(1, 40), n))
        elif kind == 1:
            parts.append(np.zeros(n, dtype=np.int64))
        elif kind == 2:
            parts.append(rng.integers(0, 65, n))
        elif kind == 3:
            parts.append(rng.integers(60, 70, max(1, n // 4)))
        elif kind == 4:
            parts.append(rng.integers(400, 5_000, max(1, n // 200)))
        elif kind == 5:
            parts.append(np.where(rng.random(n) < 0.5, 0, rng.integers(1, 20, n)))
        elif kind == 6:
            parts.append(np.array([int(rng.integers(1, 3_000_000))]))
        elif kind == 7:
            parts.append(rng.integers(2_048, 6_000, max(128, n // 100)))
        else:
            parts.append(synth.zipf_counts(max(2, n // 10), int(rng.integers(10_000, 2_000_000)), seed=int(rng.integers(1, 1 << 30)), nrow=1_000_000))
    return np.concatenate(parts).astype(np.int64)


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    capi.load()
    t0 = time.time()
    n = 0
    forms = {}
    while time.time() - t0 < seconds:
        counts = random_counts(rng)
        if counts.sum() == 0 or counts.sum() > 400_000_000:
            continue
        capi.set_tuning(int(rng.choice([0, 0, 0, 1, 3, 17])))
        capi.set_lean(int(rng.choice([1, 1, 2, 0])))
        p = synth.offsets_from_counts(counts)
        nnz = int(p[-1])
        x = synth.gen_values(nnz, seed=int(rng.integers(1, 1 << 30)), kind=0)
        xt, pt = torch.from_numpy(x).cuda(), torch.from_numpy(p).cuda()
        host = capi.ColumnSumsPlan(p, nnz=nnz)
        dev = capi.ColumnSumsPlan(pt, nnz=nnz).wait()
        tag = (len(counts), nnz, seed, n)
        same = (dev.form, dev.nchunks, dev.chunk_elems, dev.max_skip) == (host.form, host.nchunks, host.chunk_elems, host.max_skip)
        if not same:
            # the one allowed difference: a chunk denser than the device-made image has room for keeps it out of the lean form
            assert host.lean and not dev.lean, (tag, host.form, dev.form, host.max_skip, dev.max_skip)
        else:
            for what in (0, 1):
                a, b = host.image(what), dev.image(what)
                assert a.tobytes() == b.tobytes(), (tag, what, a.size, b.size)
        gd = dev.column_sums(xt, pt).cpu().numpy()
        gh = host.column_sums(xt, pt).cpu().numpy()
        ref = oracle.column_sums(x, p)
        scale = oracle.column_abs_sums(x, p)
        assert np.all(np.abs(gd - ref) <= 1e-12 * scale) and np.all(np.abs(gh - ref) <= 1e-12 * scale), tag
        if dev.lean:
            assert gd.tobytes() == ref.tobytes(), tag
        forms[dev.form] = forms.get(dev.form, 0) + 1
        host.close()
        dev.close()
        n += 1
    capi.set_tuning(0)
    capi.set_lean(1)
    print(json.dumps({"cases": n, "seconds": time.time() - t0, "seed": seed,
                      "device_made_forms": {{3: "columns", 2: "lean", 1: "snapped", 0: "general"}[k]: v for k, v in forms.items()}}))


if __name__ == "__main__":
    main()
