#!/usr/bin/env python3
"""Form-edge non-regression sweep.

The library picks one of several forms of a call from hard thresholds measured on one device class.  This tool goes
to every threshold that has a forcing knob, builds a matrix just on the chosen form's side of it (and, where the
alternative can run there too, just on the other side), times the CHOSEN form and its NEIGHBOUR on the same matrix and
reports chosen / neighbour.  A ratio above 1.10 means the threshold sends that matrix to the slower form on this device:
exit code 1 (and `ok: false` in the JSON line of that edge).

    python tools/edge_sweep.py [--quick] [--only lean,columns,...] [--out profiles/r04_form_edges.json]   (on the GPU box)

Edges covered (knob that forces the neighbour):
  lean | snapped            every column <= 64 entries                          rsp_debug_set("lean", 0)
  snapped | general         no column reaches > 512 past a chunk edge            the plan-free entry
  columns | general         min column >= 2048 (4 wavefronts) / >= 512 up to     the plan-free entry; rsp_debug_set("columns_form", 2)
                            2.5e8 entries (2 wavefronts); max <= 4 x mean         forces the columns form on the far side
  taper | no taper          calls of more than 12288 body chunks                 rsp_debug_set("taper_*")
  slices | L2 probes        row-restricted sums, > 2^20 rows: >= 16384 columns,   rsp_debug_set("row_slices", 0 / 2)
                            >= 32 entries per column and slice
  segments | regrouped      a handle's row sums: >= 128 entries per column and    rsp_debug_set("row_segments", 0 / 2)
                            row block (a trade between first and repeated calls:
                            limit 1.25 here)
  tall | exact              crossprod: <= 512 columns of >= 4096 entries          rsp_set_crossprod_exact
Not covered (no forcing knob; their thresholds are compile-time constants, measured in profiles/r02_* / r03_*): the
L1 / LDS bitmap sizes of the row-restricted sums, direct / partition / coarse / two-level row sums (block counts), the
short-call pipeline, lean rows per chunk (RSP_LEAN_ROWS is read once per process).
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["RCPPSPARSE_REQUIRE_GPU"] = "1"

import numpy as np   # noqa: E402
import torch         # noqa: E402

from rcppsparse_amd import capi, synth   # noqa: E402

LIMIT = 1.10
FORMS = {3: "columns", 2: "lean", 1: "snapped", 0: "general"}


def offsets(counts):
    return synth.offsets_from_counts(np.asarray(counts, dtype=np.int64))


def timed(fn, copies, reps):
    """ms per call: `reps` calls back to back rotating over `copies` (inputs of small calls must come from HBM, not
    from the Infinity Cache), one event pair around the lot; best of 3 such regions."""
    for k in range(min(reps, 3 * copies)):
        fn(k % copies)
    best = float("inf")
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a.record()
        for k in range(reps):
            fn(k % copies)
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / reps)
    return best


class ColumnSums:
    """x copies + offsets of one matrix on the device; callables for the plan-free entry and for a plan."""

    def __init__(self, p, seed=11):
        self.p = p
        self.ncol, self.nnz = len(p) - 1, int(p[-1])
        self.copies = max(1, -(-400_000_000 // max(1, 8 * self.nnz)))
        self.reps = 200 if self.nnz < 50_000_000 else (40 if self.nnz < 400_000_000 else 12)
        self.xs = []
        for k in range(self.copies):
            x = torch.empty(self.nnz, dtype=torch.float64, device="cuda")
            capi.gen_values_device(x, seed + k, 0, 0)
            self.xs.append(x)
        self.pt = torch.from_numpy(p).cuda()
        self.out = torch.empty(self.ncol, dtype=torch.float64, device="cuda")
        self.ws = capi.alloc_workspace(self.ncol, self.nnz)

    def general_ms(self):
        self.ws = capi.alloc_workspace(self.ncol, self.nnz)      # (the chunking knobs change what the carries need)
        runs = [capi.prepared_column_sums(x, self.pt, self.out, self.ws) for x in self.xs]
        return timed(lambda k: runs[k](), self.copies, self.reps)

    def planned_ms(self):
        plan = capi.ColumnSumsPlan(self.p, nnz=self.nnz)
        runs = [plan.prepared(x, self.pt, self.out, self.ws) for x in self.xs]
        ms = timed(lambda k: runs[k](), self.copies, self.reps)
        form = FORMS[plan.form]
        plan.close()
        return ms, form

    def check(self, a_fn, b_fn):
        """the two forms agree within the documented tolerance on copy 0"""
        a_fn()
        a = self.out.clone()
        b_fn()
        l1 = capi.column_reduce_device(self.xs[0], self.pt, capi.OP_SUM_ABS)
        assert bool(torch.all((a - self.out).abs() <= 2e-12 * l1)), "forms disagree"


def edge(name, side, shape, chosen, chosen_ms, other, other_ms, limit=None, note=None):
    ratio = chosen_ms / other_ms
    rec = {"edge": name, "side": side, "shape": shape, "chosen": chosen, "chosen_ms": chosen_ms, "neighbour": other,
           "neighbour_ms": other_ms, "chosen_over_neighbour": ratio, "ok": ratio <= (limit or LIMIT)}
    if limit:
        rec["limit"] = limit
    if note:
        rec["note"] = note
    print(json.dumps(rec), flush=True)
    return rec


def sweep_lean(quick):
    out = []
    rng = np.random.default_rng(1)
    shapes = [("1e6 columns of ~10 (C2)", rng.poisson(10, 1_000_000)),
              ("2e5 columns of exactly 64", np.full(200_000, 64)),
              ("3e5 columns of 40..64", rng.integers(40, 65, 300_000))]
    if not quick:
        shapes += [("1e7 columns of ~10", rng.poisson(10, 10_000_000)), ("3.3e6 columns of ~30", rng.poisson(30, 3_300_000).clip(0, 64))]
        # where the lean form stops paying: mean lengths up to the 64-entry limit at 1e7 and 1e8 entries
        for mean in (20, 30, 36, 40, 44, 48, 56):
            for total in (10_000_000, 100_000_000):
                shapes.append((f"{total // mean} columns of {mean - 4}..{mean + 4}", rng.integers(mean - 4, mean + 5, total // mean)))
    for label, counts in shapes:
        m = ColumnSums(offsets(np.minimum(counts, 64)))
        capi.set_lean(True)
        auto_ms, form = m.planned_ms()
        capi.set_lean(False)
        snap_ms, form2 = m.planned_ms()
        capi.set_lean(True)
        if form == "lean":
            out.append(edge("lean | snapped", "lean side (all columns <= 64)", label, "lean", auto_ms, form2, snap_ms))
            out.append(edge("lean | general (no plan)", "lean side", label, "lean", auto_ms, "general", m.general_ms()))
        else:     # past the mean-length limit of the lean form: the neighbour is the lean form, forced
            capi.set_lean(2)
            lean_ms, f3 = m.planned_ms()
            capi.set_lean(True)
            assert f3 == "lean", f3
            out.append(edge("lean | snapped", "snapped side (all columns <= 64, mean above the lean form's limit)", label, form,
                            auto_ms, "lean (forced)", lean_ms))
        del m
        torch.cuda.empty_cache()
    return out


def sweep_snapped(quick):
    out = []
    rng = np.random.default_rng(2)
    shapes = [("1e5 columns of 70..200", rng.integers(70, 201, 100_000)), ("4e4 columns of 300..500", rng.integers(300, 501, 40_000))]
    if not quick:
        shapes.append(("1e6 columns of 70..200", rng.integers(70, 201, 1_000_000)))
    capi.set_columns_form(0)
    for label, counts in shapes:
        m = ColumnSums(offsets(counts))
        ms, form = m.planned_ms()
        g = m.general_ms()
        if form == "snapped":
            out.append(edge("snapped | general", "snapped side (max skip <= 512)", label, "snapped", ms, "general", g))
        del m
        torch.cuda.empty_cache()
    capi.set_columns_form(1)
    return out


def sweep_columns(quick):
    """chosen columns just above each threshold; just below it the general kernels are chosen and the columns form is the
    forced neighbour."""
    out = []
    rng = np.random.default_rng(3)
    cases = [
        ("min length 2048", "columns side", rng.integers(2048, 2300, 1_000), True),
        ("min length 2048", "general side (columns of 1900..2047, 3e8 entries: beyond the two-wavefront size limit)",
         rng.integers(1900, 2048, 150_000), False),
        ("min length 512 (two wavefronts)", "columns side", rng.integers(512, 700, 100_000), True),
        ("min length 512 (two wavefronts)", "general side (columns of 400..511)", rng.integers(400, 512, 100_000), False),
        ("vignette shape", "columns side", rng.integers(9_000, 11_000, 1_000), True),
        ("max <= 4 x mean", "general side (one column of 6 x the mean among 2000 of ~5000)",
         np.concatenate([rng.integers(4_500, 5_500, 2_000), [30_000]]), False),
    ]
    for nc in (8, 32, 96, 127):        # fewer than 128 columns: the form is taken while no column is longer than 32768 entries
        for ln in (3_000, 22_000, 36_000, 46_000, 56_000, 70_000, 300_000):
            cases.append((f"{nc} columns of ~{ln}", "?", rng.integers(int(ln * 0.95), int(ln * 1.05), nc), None))
    if not quick:
        cases += [
            ("two wavefronts up to 2.5e8 entries", "columns side (2.4e8 entries)", rng.integers(900, 1_100, 240_000), True),
            ("two wavefronts up to 2.5e8 entries", "general side (2.6e8 entries)", rng.integers(900, 1_100, 260_000), False),
            ("two wavefronts up to 2.5e8 entries", "general side (C3: 1e9 entries)", synth.uniform_counts(1_000_000, 1_000_000_000, 42, 10_000_000), False),
        ]
    for label, side, counts, expect_columns in cases:
        m = ColumnSums(offsets(counts))
        capi.set_columns_form(1)
        ms, form = m.planned_ms()
        if expect_columns is None:
            expect_columns = form == "columns"
            side = "columns side" if expect_columns else "general side"
        assert (form == "columns") == expect_columns, (label, side, form)
        if expect_columns:
            out.append(edge("columns | general: " + label, side, f"{m.ncol} columns, {m.nnz} entries", "columns", ms,
                            "general", m.general_ms()))
        else:
            capi.set_columns_form(2)
            forced_ms, f2 = m.planned_ms()
            capi.set_columns_form(1)
            assert f2 == "columns", f2
            chosen_ms = ms if form != "general" else m.general_ms()
            out.append(edge("columns | general: " + label, side, f"{m.ncol} columns, {m.nnz} entries", form, chosen_ms,
                            "columns (forced)", forced_ms))
        del m
        torch.cuda.empty_cache()
    return out


def sweep_taper(quick):
    out = []
    sizes = [300_000_000, 500_000_000] if not quick else [300_000_000]
    for nnz in sizes:
        ncol = nnz // 1000
        m = ColumnSums(offsets(synth.uniform_counts(ncol, nnz, 5, 10_000_000)))
        plan = capi.plan_describe(nnz)
        tapered = plan["tail_elems"] != plan["body_elems"]
        capi.set_taper(-1, -1)
        auto_ms = m.general_ms()
        if tapered:
            capi.set_taper(0, 0)
        else:
            capi.set_taper(100, 64)
        other_ms = m.general_ms()
        capi.set_taper(-1, -1)
        out.append(edge("taper | no taper", f"{'tapered' if tapered else 'untapered'} side ({plan['nchunks']} chunks)",
                        f"{ncol} columns, {nnz} entries", "taper" if tapered else "no taper", auto_ms,
                        "no taper" if tapered else "taper 10 % x 64 rows", other_ms))
        del m
        torch.cuda.empty_cache()
    return out


def masked_ms(nrow, ncol, nnz, reps=5):
    p = offsets(synth.uniform_counts(ncol, nnz, 42, nrow))
    pt = torch.from_numpy(p).cuda()
    xt = torch.empty(nnz, dtype=torch.float64, device="cuda")
    it = torch.empty(nnz, dtype=torch.int32, device="cuda")
    capi.gen_values_device(xt, 42, 0, 0)
    capi.gen_row_indices_device(it, pt, nrow, 42)
    out = torch.empty(ncol, dtype=torch.float64, device="cuda")
    ws = torch.empty(capi.in_rows_workspace_bytes(nrow, ncol, nnz), dtype=torch.uint8, device="cuda")
    bits = torch.from_numpy(np.random.default_rng(0).integers(0, 2**32, size=(nrow + 31) // 32, dtype=np.uint32)).cuda()
    res = {}
    for setting, key in ((1, "auto"), (0, "L2"), (2, "slices")):
        capi.set_row_slices(setting)
        if key == "auto":
            res["form"] = capi.in_rows_form(nrow, ncol, nnz)
        res[key] = timed(lambda k: capi.column_sums_in_rows_device(xt, it, pt, nrow, bits, False, out, ws), 1, reps)
    capi.set_row_slices(1)
    return res


def sweep_slices(quick):
    out = []
    cases = [("16384 columns", 10_000_000, 16_384, 1_000_000_000), ("16384 columns", 10_000_000, 16_000, 1_000_000_000),
             ("32 entries per column and slice", 10_000_000, 3_000_000, 1_000_000_000),
             ("32 entries per column and slice", 10_000_000, 3_400_000, 1_000_000_000)]
    if quick:
        cases = [("32 entries per column and slice", 4_000_000, 600_000, 100_000_000), ("32 entries per column and slice", 4_000_000, 900_000, 100_000_000)]
    else:
        cases += [("C3 shape", 10_000_000, 1_000_000, 1_000_000_000), ("few long columns (1e4)", 10_000_000, 10_000, 1_000_000_000)]
    for label, nrow, ncol, nnz in cases:
        r = masked_ms(nrow, ncol, nnz)
        chosen = r["form"]
        out.append(edge("slices | L2 probes: " + label, f"{chosen} chosen", f"{nrow} rows, {ncol} columns, {nnz} entries "
                        f"({nnz / ncol / -(-nrow // (1 << 20)):.0f} per column and slice)", chosen, r["auto"],
                        "L2" if chosen == "slices" else "slices (forced)", r["L2"] if chosen == "slices" else r["slices"]))
        torch.cuda.empty_cache()
    return out


def sweep_segments(quick):
    """a handle's row sums, repeated calls (the form's table / regrouped copy already built)"""
    out = []
    import oracle
    cases = [("128 per column and block", 60_000, 1_000, 600), ("128 per column and block", 60_000, 1_000, 400)]
    if not quick:
        cases += [("128 per column and block", 1_000_000, 30_000, 8_000), ("128 per column and block", 1_000_000, 30_000, 7_000)]
        cases += [("shorter pieces", 60_000, 4_000, 256), ("shorter pieces", 60_000, 4_000, 160), ("shorter pieces", 1_000_000, 30_000, 4_000),
                  ("shorter pieces", 1_000_000, 30_000, 2_500), ("shorter pieces", 1_000_000, 60_000, 1_500)]
    for label, nrow, ncol, per_col in cases:
        p = offsets(np.full(ncol, per_col))
        nnz = int(p[-1])
        x = synth.gen_values(nnz, 5, kind=0)
        i = oracle.gen_row_indices(p, nrow, 5)
        res = {}
        import time
        for setting, key in ((1, "auto"), (0, "regrouped"), (2, "segments")):
            capi.set_row_segments(setting)
            h = capi.DeviceCSC(x, p, (nrow, ncol), i=i)
            # what a handle is for: the first call (builds the table / the regrouped copy) and a few repeated ones
            t0 = time.perf_counter()
            for _ in range(5):
                h.row_sums()
            res[key] = (time.perf_counter() - t0) / 5 * 1e3
            if key == "auto":
                res["form"] = h.row_form()
            h.close()
        capi.set_row_segments(1)
        chosen = res["form"]
        blocks = -(-nrow // 16384)
        out.append(edge("segments | regrouped copy", f"{chosen} chosen", f"{nrow} rows ({blocks} blocks), {ncol} columns of {per_col} "
                        f"({per_col / blocks:.0f} per column and block); mean of the first five handle calls (the first builds the form) incl. the copy of the result",
                        chosen, res["auto"], "regrouped" if chosen == "segments" else "segments (forced)",
                        res["regrouped"] if chosen == "segments" else res["segments"], limit=1.25,
                        note="a trade, not a speed edge: the segments form's first call is 3-4 x cheaper (no regrouped copy is "
                             "built or kept), its repeated calls up to 25 % dearer; host wall time of ~0.1-1 ms calls, noisy: "
                             "this edge fails the sweep only beyond 1.25"))
    return out


def sweep_crossprod(quick):
    out = []
    import oracle
    cases = [(1_000_000, 256, 4_096), (1_000_000, 256, 10_000), (1_000_000, 256, 12_000), (1_000_000, 256, 14_000), (1_000_000, 256, 20_000), (1_000_000, 256, 40_000),
             (250_000, 256, 4_096), (250_000, 224, 6_000), (4_000_000, 256, 40_000),
             (1_000_000, 192, 4_096), (1_000_000, 192, 8_000), (1_000_000, 192, 12_000), (1_000_000, 160, 9_000), (250_000, 192, 4_096),
             (1_000_000, 128, 6_000), (1_000_000, 100, 5_000), (250_000, 128, 4_096), (4_000_000, 128, 20_000),
             (1_000_000, 512, 4_096), (1_000_000, 512, 12_000), (1_000_000, 512, 20_000), (1_000_000, 384, 8_000), (1_000_000, 384, 14_000),
             (250_000, 400, 4_096), (4_000_000, 320, 30_000), (100_000, 512, 4_096),
             # (widths between the tile counts: the 24 / 32-tile kernels per real tile count, round 5)
             (1_000_000, 272, 8_000), (1_000_000, 272, 14_000), (1_000_000, 300, 10_000), (1_000_000, 420, 12_000), (1_000_000, 420, 20_000), (250_000, 288, 5_000),
             (1_000_000, 128, 4_096), (1_000_000, 128, 10_000), (1_000_000, 64, 4_096), (100_000, 256, 4_096),
             (4_000_000, 200, 8_000), (4_000_000, 200, 60_000)] if not quick else [(300_000, 64, 4_096)]
    for nrow, ncol, per_col in cases:
        p = offsets(np.full(ncol, per_col))
        nnz = int(p[-1])
        pt = torch.from_numpy(p).cuda()
        xt = torch.from_numpy(synth.gen_values(nnz, 5, kind=0)).cuda()
        it = torch.from_numpy(oracle.gen_row_indices(p, nrow, 5)).cuda()
        res = {}
        chosen = capi.crossprod_form(nrow, ncol, nnz)
        for key, exact, force in (("exact", True, 0), ("tall", False, 1)):
            capi.set_crossprod_exact(exact)
            os.environ["RSP_CROSSPROD_TALL_ALWAYS"] = "1" if force else "0"   # (read at every plan: the cost model aside)
            nbytes = int(capi.load().rsp_crossprod_workspace_bytes(nrow, ncol, nnz))
            ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
            o = torch.empty((ncol, ncol), dtype=torch.float64, device="cuda")
            res[key] = timed(lambda k: capi.crossprod_device(xt, it, pt, nrow, o, ws), 1, 3)
        os.environ["RSP_CROSSPROD_TALL_ALWAYS"] = "0"
        capi.set_crossprod_exact(False)
        other = "exact" if chosen == "tall" else "tall"
        out.append(edge("crossprod tall | exact", f"{chosen} chosen", f"{nrow} rows, {ncol} columns of {per_col} "
                        f"({100.0 * per_col / nrow:.2f} % dense)", chosen, res[chosen], other + (" (forced)" if other == "tall" else ""), res[other]))
    return out


SWEEPS = {"lean": sweep_lean, "snapped": sweep_snapped, "columns": sweep_columns, "taper": sweep_taper, "slices": sweep_slices,
          "segments": sweep_segments, "crossprod": sweep_crossprod}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true", help="small shapes only (the -m gpu smoke)")
    ap.add_argument("--only", default="")
    ap.add_argument("--limit", type=float, default=1.10, help="chosen / neighbour above this fails an edge")
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    capi.load()
    capi.set_auto_plan(False)      # ("the plan-free entry" below means the general kernels: since round 5 it would otherwise plan for itself)
    torch.cuda.set_device(0)
    global LIMIT
    LIMIT = args.limit
    names = [n for n in args.only.split(",") if n] or list(SWEEPS)
    recs = []
    for n in names:
        recs += SWEEPS[n](args.quick)
    bad = [r for r in recs if not r["ok"]]
    props = torch.cuda.get_device_properties(0)
    summary = {"device": props.name, "device_uuid": str(getattr(props, "uuid", "")), "compute_units": props.multi_processor_count,
               "limit": LIMIT, "edges": recs,
               "worst": max(r["chosen_over_neighbour"] for r in recs) if recs else None,
               "edges_where_the_chosen_form_is_more_than_10_percent_slower": [f'{r["edge"]} [{r["side"]}]' for r in bad]}
    if args.out:
        json.dump(summary, open(args.out, "w"), indent=1)
    print(json.dumps({k: v for k, v in summary.items() if k != "edges"}), flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
