#!/usr/bin/env python3
"""Soak of the plan-free device entry's own planning (rsp_column_sums_device, capi.hip auto_enqueue; round 5).

A few device buffers of fixed sizes ("slots": ncol, nnz, x, p, workspace) are reused for the whole run -- the situation the
design has to survive: the SAME addresses and sizes, new offsets.  Every step picks a slot, writes a new random p[] of
that slot's (ncol, nnz) into the device buffer IN PLACE (short columns / arbitrary lengths / one giant column / runs of
empty columns / long similar columns), sometimes new values too, and calls the entry one to four times; EVERY call's
result is compared with the oracle (lean form: bit for bit).  Between steps the library is in every state there is: plan
unknown, known, stale, re-inspecting, given up (after four stale rounds a key stays on the general kernels;
rsp_release_cached every 60 steps starts it over).
    python tools/soak_auto_plan.py [seconds] [seed]          (on the GPU box)"""
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


def fit(counts, nnz, rng, cap=None):
    """counts (any non-negative integers) -> the same shape with sum exactly nnz (and no entry above cap)"""
    counts = np.asarray(counts, dtype=np.int64)
    if cap is not None:
        counts = np.minimum(counts, cap)
    diff = nnz - int(counts.sum())
    n = len(counts)
    while diff != 0:
        k = rng.integers(0, n, size=min(abs(diff), n))
        if diff > 0:
            room = np.ones(len(k), dtype=bool) if cap is None else counts[k] < cap
            np.add.at(counts, k[room], 1)
        else:
            k = k[counts[k] > 0]
            np.add.at(counts, np.unique(k), -1)
        diff = nnz - int(counts.sum())
    return counts


def random_offsets(rng, ncol, nnz):
    mean = nnz / ncol
    fam = int(rng.integers(0, 6))
    if fam == 0 and mean <= 40:           # every column short: the lean form's shape
        c = fit(rng.poisson(mean, ncol), nnz, rng, cap=64)
        name = "short"
    elif fam == 1:                        # arbitrary lengths
        c = fit(rng.multinomial(nnz, rng.dirichlet(np.full(ncol, rng.uniform(0.2, 3.0)))), nnz, rng)
        name = "dirichlet"
    elif fam == 2:                        # one giant column
        c = np.zeros(ncol, dtype=np.int64)
        c[int(rng.integers(0, ncol))] = int(nnz * rng.uniform(0.2, 0.9))
        c = fit(c + rng.multinomial(nnz - int(c.sum()), np.full(ncol, 1.0 / ncol)), nnz, rng)
        name = "giant"
    elif fam == 3:                        # runs of empty columns
        live = rng.random(ncol) < rng.uniform(0.05, 0.6)
        live[int(rng.integers(0, ncol))] = True
        c = np.zeros(ncol, dtype=np.int64)
        c[live] = rng.multinomial(nnz, np.full(int(live.sum()), 1.0 / live.sum()))
        name = "empty-runs"
    elif fam == 4 and mean >= 2048:       # long similar columns: the columns form's shape
        c = fit(rng.integers(int(0.7 * mean), int(1.3 * mean), ncol), nnz, rng)
        name = "long-similar"
    else:                                 # uniform
        c = rng.multinomial(nnz, np.full(ncol, 1.0 / ncol)).astype(np.int64)
        name = "uniform"
    p = synth.offsets_from_counts(c)
    assert p[0] == 0 and int(p[-1]) == nnz and len(p) == ncol + 1
    return p, name


def rewrites(n_rewrites, calls_between=70, deep_queue=False):
    """p[] rewritten IN PLACE every `calls_between` calls (32 clean planned calls forgive a stale
    round, so the key is re-planned every time): before round 6 every rewrite left one plan image behind for good.  Free
    HBM (hipMemGetInfo) must stay within ONE image of where it stood after the first plan, the sums right, the key planned.
    The caller waits for its update before it goes on (what a synchronous caller like R does).  deep_queue: it does not --
    the host then runs hundreds of calls ahead of the device, every new plan is stale before its statistics arrive, and
    after four such rounds the key rightly stays on the general kernels; memory must stay bounded all the same."""
    capi.load()
    capi.release_cached()
    ncol, mean = 1_000_000, 10                     # BASELINE config 2's shape: a lean image of ~2.7 MB
    rng = np.random.default_rng(7)
    pa = synth.offsets_from_counts(np.minimum(rng.poisson(mean, size=ncol), 64).astype(np.int64))
    nnz = int(pa[-1])
    pb = pa.copy()
    inner = np.flatnonzero((pa[1:-1] > pa[:-2]) & (pa[1:-1] < pa[2:]))[::5] + 1
    pb[inner] -= 1
    x = synth.gen_values(nnz, seed=7, kind=0)
    xt, pt = torch.from_numpy(x).cuda(), torch.from_numpy(pa).cuda()
    qa, qb = torch.from_numpy(pa).cuda(), torch.from_numpy(pb).cuda()
    out = torch.empty(ncol, dtype=torch.float64, device="cuda")
    ws = capi.alloc_workspace(ncol, nnz)
    refs = {0: (oracle.column_sums(x, pa), oracle.column_abs_sums(x, pa)), 1: (oracle.column_sums(x, pb), oracle.column_abs_sums(x, pb))}
    torch.cuda.synchronize()
    free_before_plan = torch.cuda.mem_get_info()[0]
    capi.column_sums_device(xt, pt, out, ws)
    assert capi.column_sums_device_settle(pt, nnz) == "lean"
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    image = free_before_plan - free0
    low = free0
    t0 = last_note = time.time()
    for r in range(n_rewrites):
        which = (r + 1) % 2
        pt.copy_(qb if which else qa)
        if not deep_queue:
            torch.cuda.synchronize()
        for _ in range(calls_between):
            capi.column_sums_device(xt, pt, out, ws)
        if r % 250 == 249 or r == n_rewrites - 1:
            torch.cuda.synchronize()
            got = out.cpu().numpy()
            ref, scale = refs[which]
            assert np.all(np.abs(got - ref) <= 1e-12 * scale), r
            free = torch.cuda.mem_get_info()[0]
            low = min(low, free)
            if free0 - free > max(image, 1 << 20) + (1 << 21):
                print(json.dumps({"FAILED": "HBM use grows", "rewrite": r, "free_at_start": free0, "free_now": free, "image_bytes": image}))
                sys.exit(1)
        if time.time() - last_note > 60:
            last_note = time.time()
            print(f"[soak_auto_plan] rewrite {r} of {n_rewrites}", file=sys.stderr, flush=True)
    form = capi.column_sums_device_settle(pt, nnz)
    torch.cuda.synchronize()
    capi.column_sums_device(xt, pt, out, ws)          # (a call collects what has been retired)
    print(json.dumps({"rewrites": n_rewrites, "deep_queue": deep_queue, "plans_made": capi.debug_get("auto_plans_made"),
                      "plans_freed": capi.debug_get("auto_plans_freed"), "plans_from_recycled_allocations": capi.debug_get("auto_plans_recycled"), "plans_retired_now": capi.debug_get("auto_plans_retired"), "calls_between": calls_between, "calls": n_rewrites * calls_between,
                      "seconds": round(time.time() - t0, 1), "image_bytes": int(image), "free_at_start": int(free0),
                      "lowest_free_seen": int(low), "most_extra_bytes_held": int(free0 - low), "form_at_the_end": form,
                      "mismatches": 0}))
    assert deep_queue or form == "lean"


def main():
    if len(sys.argv) > 1 and sys.argv[1] in ("rewrites", "rewrites-deep"):
        return rewrites(int(sys.argv[2]) if len(sys.argv) > 2 else 10_000, deep_queue=sys.argv[1] == "rewrites-deep")
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    capi.load()
    capi.set_auto_plan(True)
    slots = []
    for ncol, nnz in ((150_000, 1_500_000), (400_000, 3_000_000), (60_000, 2_200_000), (700, 2_800_000), (300, 1_300_000),
                      (1_000_000, 1_200_000)):
        x = synth.gen_values(nnz, seed=seed + len(slots), kind=0)
        slots.append({"ncol": ncol, "nnz": nnz, "x": x, "xt": torch.from_numpy(x).cuda(),
                      "pt": torch.zeros(ncol + 1, dtype=torch.int32, device="cuda"),
                      "out": torch.empty(ncol, dtype=torch.float64, device="cuda"), "ws": capi.alloc_workspace(ncol, nnz),
                      "p": None})
    t0 = last_note = time.time()
    steps = calls = 0
    forms, fams, exact_calls, worst = {}, {}, 0, 0.0
    while time.time() - t0 < seconds:
        s = slots[int(rng.integers(0, len(slots)))]
        if s["p"] is None or rng.random() < 0.35:
            s["p"], fam = random_offsets(rng, s["ncol"], s["nnz"])
            fams[fam] = fams.get(fam, 0) + 1
            s["pt"].copy_(torch.from_numpy(s["p"]))               # in place: same address, same sizes, new offsets
            s["ref"] = None
        if rng.random() < 0.15:
            s["x"] = synth.gen_values(s["nnz"], seed=int(rng.integers(1, 1 << 30)), kind=int(rng.integers(0, 2)))
            s["xt"].copy_(torch.from_numpy(s["x"]))
            s["ref"] = None
        if s["ref"] is None:
            s["ref"] = (oracle.column_sums(s["x"], s["p"]), oracle.column_abs_sums(s["x"], s["p"]))
        ref, scale = s["ref"]
        for k in range(int(rng.integers(1, 5))):
            if rng.random() < 0.3:
                capi.column_sums_device_settle(s["pt"], s["nnz"])    # sometimes make the plan now and let its result arrive first
            s["out"].fill_(-3.0)
            capi.column_sums_device(s["xt"], s["pt"], s["out"], s["ws"])
            form = capi.column_sums_device_form(s["pt"], s["nnz"])
            got = s["out"].cpu().numpy()
            calls += 1
            err = np.abs(got - ref)
            bad = ~(err <= 1e-12 * scale)
            if bad.any():
                c = int(np.flatnonzero(bad)[0])
                print(json.dumps({"FAILED": True, "step": steps, "call": k, "slot": [s["ncol"], s["nnz"]], "form_after": form,
                                  "column": c, "got": float(got[c]), "ref": float(ref[c]), "len": int(s["p"][c + 1] - s["p"][c])}))
                sys.exit(1)
            nz = scale > 0
            if nz.any():
                worst = max(worst, float(np.max(err[nz] / scale[nz])))
            if got.tobytes() == ref.tobytes():
                exact_calls += 1
            forms[form] = forms.get(form, 0) + 1
        steps += 1
        if steps % 60 == 0:
            capi.release_cached()
        if time.time() - last_note > 60:            # (a run that writes nothing for minutes is taken to be hung)
            last_note = time.time()
            print(f"[soak_auto_plan] {int(last_note - t0)} s: {steps} steps, {calls} calls checked", file=sys.stderr, flush=True)
    print(json.dumps({"seconds": round(time.time() - t0, 1), "seed": seed, "steps": steps, "calls_checked": calls,
                      "calls_with_the_references_bits": exact_calls, "form_after_call": forms, "offset_families": fams,
                      "worst_err_over_l1": worst, "slots": [[s["ncol"], s["nnz"]] for s in slots], "mismatches": 0}))


if __name__ == "__main__":
    main()
