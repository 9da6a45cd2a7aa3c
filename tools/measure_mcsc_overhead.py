#!/usr/bin/env python3
"""Host overhead of the single-process multi-GPU column-sum call (rsp_mcsc_column_sums) -- VERDICT round 5, next 1.

k = 1, 2, 4, 8 shards of BASELINE config 4's size (1.25e8 entries in 125000 columns each: 1e9 / 8), all resident on
THIS box's one device (rsp_mcsc_wrap_device: generated in HBM, nothing uploaded).  On one card the k kernels share the
device, so the device time of a call is the SUM of the shards' kernel times ("serialized kernel time", each shard timed
alone with HIP events); what is left of the call's wall time is what the host adds:

    overhead_us = median wall time of a call - sum of the shards' kernel times

per launch mode (serial | workers) and gather (none = the slices stay on the devices: launch + wait cost alone | d2h |
stores), with the result copied into a pageable vector (what an R
NumericVector is) and, `nocopy`, left in the handle's page-locked vector.  On a node with k devices the kernels overlap
instead, and what then matters is when the LAST shard's launch has been issued: `last_enqueue_us` (host clock from the
call's entry to the return of the slowest shard's enqueue), reported from the same calls.
`hot` = calls back to back (the workers are still spinning), `cold` = 2 ms of sleep between calls (they have parked).

    python tools/measure_mcsc_overhead.py [--shards 1,2,4,8] [--calls 40] [--out profiles/r06_mcsc_overhead.json]
"""
import argparse
import json
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["RCPPSPARSE_REQUIRE_GPU"] = "1"

import numpy as np  # noqa: E402


def med(v):
    return float(statistics.median(v))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shards", default="1,2,4,8")
    ap.add_argument("--calls", type=int, default=40)
    ap.add_argument("--shard-nnz", type=int, default=125_000_000)
    ap.add_argument("--shard-ncol", type=int, default=125_000)
    ap.add_argument("--out", default="")
    args = ap.parse_args()

    import torch
    from rcppsparse_amd import capi, synth
    capi.load()
    torch.cuda.set_device(0)
    nrow = 10_000_000
    rec = {"what": "rsp_mcsc_column_sums: wall time of a call minus the serialized kernel time, shards of C4's size on one device",
           "device": torch.cuda.get_device_name(0), "shard_nnz": args.shard_nnz, "shard_ncol": args.shard_ncol,
           "host_cores": len(os.sched_getaffinity(0)), "calls": args.calls, "runs": []}
    for G in [int(t) for t in args.shards.split(",")]:
        xs, ps = [], []
        for k in range(G):
            counts = synth.uniform_counts(args.shard_ncol, args.shard_nnz, seed=42 + k, nrow=nrow)
            xt = torch.empty(args.shard_nnz, dtype=torch.float64, device="cuda")
            capi.gen_values_device(xt, seed=42, first_idx=k * args.shard_nnz, kind=0)
            xs.append(xt)
            ps.append(torch.from_numpy(synth.offsets_from_counts(counts)).cuda())
        torch.cuda.synchronize()
        h = capi.MultiDeviceCSC.wrap_device(xs, ps, nrow)
        ncol = h.ncol
        forms = [h.shard_info(k)["form"] for k in range(G)]
        kernel_us = [h.shard_kernel_ms(k, reps=20) * 1e3 for k in range(G)]
        serialized = sum(kernel_us)
        pageable = np.empty(ncol, dtype=np.float64)
        want = None
        for launch in (("serial",) if G == 1 else ("serial", "workers")):
            for gather in ("none", "d2h", "blit", "stores"):
                h.set_launch(launch)
                h.set_gather(gather)
                for dest in (("nocopy",) if gather == "none" else ("pageable", "nocopy")):
                    out = pageable if dest == "pageable" else h.result_buffer()
                    for pace in ("hot", "cold"):
                        for _ in range(5):
                            h.column_sums(out=out)
                        wall, last_enq, last_done, copy_tail = [], [], [], []
                        for _ in range(args.calls):
                            if pace == "cold":
                                time.sleep(0.002)
                            h.column_sums(out=out)
                            st = h.last_call_stamps()
                            wall.append(st["call_us"])
                            last_enq.append(max(st["enqueued_us"]))
                            last_done.append(max(st["done_us"]))
                            copy_tail.append(st["call_us"] - max(st["done_us"]))
                        got = np.array(out, copy=True)
                        if gather != "none":
                            if want is None:
                                want = got
                            assert got.tobytes() == want.tobytes(), (G, launch, gather, dest)
                        rec["runs"].append({
                            "shards": G, "launch": launch, "gather": gather, "dest": dest, "pace": pace,
                            "wall_us_median": round(med(wall), 1), "wall_us_min": round(min(wall), 1),
                            "serialized_kernel_us": round(serialized, 1),
                            "overhead_us": round(med(wall) - serialized, 1),
                            "last_enqueue_us": round(med(last_enq), 1),
                            "after_last_stream_us": round(med(copy_tail), 1),
                        })
                        print(json.dumps(rec["runs"][-1]), flush=True)
        rec.setdefault("shard_forms", {})[str(G)] = forms
        rec.setdefault("shard_kernel_us", {})[str(G)] = [round(v, 1) for v in kernel_us]
        h.close()
        del xs, ps
        torch.cuda.empty_cache()
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        with open(args.out, "w") as f:
            json.dump(rec, f, indent=1)
    print("SUMMARY " + json.dumps({f"{r['shards']}/{r['launch']}/{r['gather']}/{r['dest']}/{r['pace']}": r["overhead_us"]
                                   for r in rec["runs"]}))


if __name__ == "__main__":
    main()
