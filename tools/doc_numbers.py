#!/usr/bin/env python3
"""Regenerates the measured-numbers tables of DESIGN.md (section 5) and BASELINE.md (section 4) from
the committed evidence of a round, so that the prose never drifts from the files:

    python tools/doc_numbers.py r02

Sources, all under profiles/:
  <round>_bench_lines.jsonl           bench.py JSON lines (one per workload), no profiler attached
  <round>_<tag>_kernel_stats.csv      rocprofv3 --kernel-trace --stats of the same bench command
  <round>_<tag>_traffic.json          HBM bytes per launch from the FETCH_SIZE / WRITE_SIZE passes
The table is also written to profiles/<round>_numbers.md; in the two documents it replaces whatever
stands between the markers <!-- numbers:begin --> and <!-- numbers:end -->."""
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")


def kernel_avgs(path):
    """(main kernel us, fix-up us or None) of the profiled calls: the column-sum kernel launched most often (the plan-free
    entry plans for itself since round 5: a command's first calls run the general kernels, the rest the form it settles on)"""
    main = fix = None
    calls, name = 0, ""
    if not os.path.exists(path):
        return None, None
    for r in csv.DictReader(open(path)):
        if ("colsums_chunks_kernel" in r["Name"] or "colsums_lean_kernel" in r["Name"] or "colsums_columns_kernel" in r["Name"]) \
                and int(r["Calls"]) > calls:
            main, calls, name = float(r["AverageNs"]) / 1e3, int(r["Calls"]), r["Name"]
        if "colsums_fixup_kernel" in r["Name"]:
            fix = float(r["AverageNs"]) / 1e3
    return main, (fix if "colsums_chunks_kernel" in name else None)


def driver_record(rnd):
    """(round tag, parsed bench line) of the driver's own N = 1 run: BENCH_<rnd>.json if the driver has run this
    round's code already, else the latest earlier one (a record of THAT round's code)."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "BENCH_r*.json"))):
        tag = os.path.basename(f)[6:-5]
        if tag <= rnd:
            try:
                parsed = json.load(open(f)).get("parsed")
            except Exception:
                parsed = None
            if parsed:
                best = (tag, parsed)
    return best


def row_of(d, label, m=None, f=None, tr=None, own_traffic=False):
    r = d["roofline"]
    pipe = d.get("pipelined") or {}
    gather = f" (+ gatherv {r['gather_ms'] * 1e3:.1f} us)" if r.get("gather_ms") else ""
    prof = "-" if m is None else (f"{m:.1f} + {f:.1f}" if f is not None else f"{m:.1f} (one launch)")
    # (the driver's record keeps the contract keys, `config`, `roofline` and `cpu_baseline` of the line; the rest only by name)
    lat = f"{d['latency_ms_per_call']:.4f}" if "latency_ms_per_call" in d else "-"
    ovl = f"{pipe['ms_per_step']:.4f}" if "ms_per_step" in pipe else "-"
    par = f"{d['parity']['max_abs_err_over_l1']:.1e}" if "parity" in d else "checked in the run (exit 0)"
    if tr is None and own_traffic and r.get("traffic"):
        tr = r["traffic"] / r["algorithmic_bytes_per_launch"]
    return (f"| {label} | {d['ms_per_step']:.4f} | {r['kernel_ms']:.4f}{gather} | {d['value']:.3e} | {r['achieved']:.0f} | "
            f"{100 * r['frac']:.1f} % | {lat} | {ovl} | "
            f"{prof} | {'-' if tr is None else f'{tr:.3f}'} | {par} |")


def main():
    rnd = sys.argv[1] if len(sys.argv) > 1 else "r03"
    lines = [json.loads(l) for l in open(os.path.join(P, f"{rnd}_bench_lines.jsonl")) if l.strip()]
    out = ["| workload (`bench.py --workload`) | ms per call (wall, K calls back to back) | kernels per call, HIP events | "
           "nnz/s | algorithmic GB/s | of 8 TB/s | one call alone, launch to host (ms) | calls overlapped (ms per call) | "
           "rocprofv3 main + fix-up (us) | HBM bytes / algorithmic | parity: max err / column 1-norm (all columns) |",
           "|---|---|---|---|---|---|---|---|---|---|---|"]
    drv = driver_record(rnd)
    if drv and drv[0] == rnd:     # the driver's own run of this round's code is the headline
        out.append(row_of(drv[1], f"c3 -- the driver's run (`BENCH_{rnd}.json`)", own_traffic=True))
    for d in lines:
        tag = d["config"]["workload"].split(":")[0]
        form = (d["config"].get("planned") or {}).get("form")
        ptag = tag + ("planned" if form in ("lean", "snapped", "columns") else "")
        auto = d["config"]["shards"][0].get("form") if form is None else None
        label = (tag + (f" (the entry's own plan: {auto} form)" if auto in ("lean", "columns") else "")) if form is None else f"{tag} `--planned` ({form} form)"
        if form is None and d["config"].get("auto_plan") == 0:
            label = f"{tag} `RSP_AUTO_PLAN=0` (general kernels)"
            ptag = tag + "general"
        if form == "snapped" and any((x["config"].get("planned") or {}).get("form") == "lean" and
                                     x["config"]["workload"].split(":")[0] == tag for x in lines):
            ptag = None               # (the committed planned profile of this workload is the lean run's)
            label = f"{tag} `--planned --no-lean` (snapped form)"
        m = f = tr = None
        if ptag:
            m, f = kernel_avgs(os.path.join(P, f"{rnd}_{ptag}_kernel_stats.csv"))
            tf = os.path.join(P, f"{rnd}_{ptag}_traffic.json")
            if os.path.exists(tf):
                tr = json.load(open(tf))["hbm_bytes_per_launch"] / d["roofline"]["algorithmic_bytes_per_launch"]
        out.append(row_of(d, label, m, f, tr))
    if drv and drv[0] != rnd:
        p = drv[1]
        out.append("")
        out.append(f"The driver's latest own record is `BENCH_{drv[0]}.json` (round {drv[0][1:]}'s code, a device of its choosing): "
                   f"c3 {p['ms_per_step']:.4f} ms per call, kernels {p['roofline']['kernel_ms']:.4f} ms = "
                   f"{100 * p['roofline']['frac']:.1f} % of 8 TB/s.  The rows above are the builder's runs of THIS round's code on "
                   f"the devices `gpurun` handed out (c3 on them: 1.19-1.24 ms, 81-85 %, device to device).")
    # what the default line measures beside the headline figure (round 4): the read-only ceiling, its own traffic, more workloads
    # (round 5: flat scalars in `roofline` -- what the driver's record keeps; the `also` list itself travels in the line only)
    full = next((d for d in ([drv[1]] if drv and drv[0] == rnd else []) + lines if (d["roofline"].get("read_ceiling_GBps"))), None)
    if full:
        r = full["roofline"]
        src = "the driver's run" if (drv and drv[0] == rnd and full is drv[1]) else "the builder's default run (`bench.py`, no flags)"
        out.append("")
        out.append(f"In the same run as the c3 line ({src}): a read-only kernel with the column sums' access shape over the same 8 GB of x "
                   f"reaches **{r['read_ceiling_GBps']:.0f} GB/s** ({100 * r['read_ceiling_GBps'] / 8000:.1f} % of the 8 TB/s spec peak) -- the c3 call is at "
                   f"**{100 * r['frac_of_ceiling']:.1f} % of that measured ceiling**"
                   + (f"; HBM traffic of one call from two rocprofv3 counter passes run by the bench itself: {r['traffic'] / 1e9:.4f} GB = "
                      f"{r['traffic_over_algorithmic']:.4f} x the algorithmic bytes" if r.get("traffic_measured_in_run") else "") + ".")
        names = sorted({k[5:-5] for k in r if k.startswith("also_") and k.endswith("_frac")}, key=lambda n: list(r).index(f"also_{n}_frac"))
        recs = {a["workload"].replace(":", "_").replace("-", "_"): a for a in (full.get("also") or [])}
        if names:
            out.append("")
            out.append("| `also` workload of that line (flat scalars `roofline.also_<w>_*`) | form | planned by | ms per call (median of 3 regions) | kernels, HIP events (ms) | of 8 TB/s | HBM bytes / algorithmic | parity: max err / column 1-norm |")
            out.append("|---|---|---|---|---|---|---|---|")
            for n in names:
                a = recs.get(n, {})
                tx = r.get(f"also_{n}_traffic_x")
                trx = r.get(f"also_{n}_traffic_recorded_x")
                traffic = f"{tx:.3f} (this run)" if tx is not None else (f"{trx:.3f} (profiles/)" if trx is not None else "-")
                by = {"entry": "the entry itself", "caller": "the caller's plan"}.get(a.get("planned_by"), "-")
                if a.get("form") == "general":
                    by = "-"
                if a.get("plan_by") == "device":
                    by += " (made on the device)"
                out.append(f"| {n} | {a.get('form', '-')} | {by} | {r[f'also_{n}_ms_per_call']:.4f} | {r[f'also_{n}_kernel_ms']:.4f} | "
                           f"{100 * r[f'also_{n}_frac']:.1f} % | {traffic} | {r[f'also_{n}_parity_err']:.1e} |")
    cpu = [d for d in lines if d.get("cpu_baseline")]
    if cpu:
        c = cpu[0]["cpu_baseline"]
        out.append("")
        out.append(f"CPU baseline in the same run (oracle, {c['cores']} thread, {c['sample'].split(',')[0]}): "
                   f"{c['value']:.3e} nnz/s" + (f"; the same loop under an OpenMP parallel-for over the columns on "
                                                 f"{c['all_cores']} cores (not what the reference does): "
                                                 f"{c['all_cores_value']:.3e} nnz/s" if "all_cores_value" in c else "") + ".")
    table = "\n".join(out) + "\n"
    open(os.path.join(P, f"{rnd}_numbers.md"), "w").write(
        f"# Measured numbers of round {rnd[1:]} (generated by tools/doc_numbers.py; do not edit)\n\n" + table)
    for doc in ("DESIGN.md", "BASELINE.md"):
        path = os.path.join(ROOT, doc)
        s = open(path).read()
        pat = re.compile(r"(<!-- numbers:begin -->\n).*?(<!-- numbers:end -->)", re.S)
        if not pat.search(s):
            print(f"{doc}: no numbers markers, left alone")
            continue
        open(path, "w").write(pat.sub(lambda m: m.group(1) + table + m.group(2), s))
        print(f"{doc}: table refreshed")


if __name__ == "__main__":
    main()
