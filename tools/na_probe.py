#!/usr/bin/env python3
"""NA / NaN payloads through every column-sum form: which bits come back.

R's NA_real_ is the signalling NaN 0x7FF00000000007A2 (low word 1954); the reference's plain `+=`
(src/example.cpp:30) returns it quieted with the payload kept (0x7FF80000000007A2), which R still
prints as NA.  This prints, per form, the hex bits of the sums of a few probe columns; the x86 column
is the same sequential add done by the host CPU of the box (python floats: addsd)."""
import json
import os
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["RCPPSPARSE_REQUIRE_GPU"] = "1"

NA = struct.unpack("<d", struct.pack("<Q", 0x7FF00000000007A2))[0]
NAN = struct.unpack("<d", struct.pack("<Q", 0x7FF8000000000000))[0]
NAN2 = struct.unpack("<d", struct.pack("<Q", 0xFFF8000000000123))[0]    # another payload, sign set
INF = float("inf")


def bits(v):
    return "%016x" % struct.unpack("<Q", struct.pack("<d", float(v)))[0]


def columns():
    rng = np.random.default_rng(5)
    fin = lambda n: list(np.round(rng.normal(size=n), 2))   # noqa: E731
    cols = {
        "na_alone": [NA],
        "short_na_mid": [1.5, NA, 2.5],
        "short_na_first": [NA, 1.5, 2.5],
        "short_na_last": [1.5, 2.5, NA],
        "nan_then_na": [NAN, NA],
        "na_then_nan": [NA, NAN],
        "nan2_then_na": [NAN2, 1.0, NA],
        "na_then_inf": [NA, INF],
        "inf_then_na": [INF, NA],
        "inf_minus_inf_then_na": [INF, -INF, NA],
        "na_then_inf_minus_inf": [NA, INF, -INF],
        "len40_na_at_0": [NA] + fin(39),
        "len40_na_at_20": fin(20) + [NA] + fin(19),
        "len40_na_at_39": fin(39) + [NA],
        "len1000_na_at_500": fin(500) + [NA] + fin(499),
        "len1000_nan_at_100_na_at_900": fin(100) + [NAN] + fin(799) + [NA] + fin(99),
        "len1000_na_at_100_nan_at_900": fin(100) + [NA] + fin(799) + [NAN] + fin(99),
        "len100000_na_at_77777": fin(77777) + [NA] + fin(22222),
        "finite": fin(10),
    }
    return cols


def x86(col):
    acc = 0.0
    for v in col:
        acc = acc + v
    return acc


def main():
    import torch
    from rcppsparse_amd import capi
    capi.load()
    cols = columns()
    names = list(cols)
    out = {"x86_python": {n: bits(x86(cols[n])) for n in names}}

    def matrix(sel):
        x = np.array([v for n in sel for v in cols[n]], dtype=np.float64)
        p = np.concatenate([[0], np.cumsum([len(cols[n]) for n in sel])]).astype(np.int32)
        return x, p

    def record(tag, sel, got):
        out[tag] = {n: bits(g) for n, g in zip(sel, got)}

    x, p = matrix(names)
    assert bits(x[0]) == "7ff00000000007a2"                      # numpy kept the signalling bit pattern
    record("host_one_shot", names, capi.column_sums_host(x, p))
    xt, pt = torch.from_numpy(x).cuda(), torch.from_numpy(p).cuda()
    record("device_general", names, capi.column_sums_device(xt, pt).cpu().numpy())
    h = capi.DeviceCSC(x, p, (200000, len(names)))
    out["handle_form"] = h.column_form()
    record("handle", names, h.column_sums())
    h.close()
    capi.set_lean(0)
    plan = capi.ColumnSumsPlan(p, nnz=len(x))
    out["planned_no_lean_form"] = plan.form
    record("planned_no_lean", names, plan.column_sums(xt, pt).cpu().numpy())
    plan.close()
    # lean: every column <= 64 entries
    short = [n for n in names if len(cols[n]) <= 64]
    xs, ps = matrix(short)
    capi.set_lean(2)
    plan = capi.ColumnSumsPlan(ps, nnz=len(xs))
    out["lean_form"] = plan.form
    record("planned_lean", short, plan.column_sums(torch.from_numpy(xs).cuda(), torch.from_numpy(ps).cuda()).cpu().numpy())
    plan.close()
    capi.set_lean(1)
    # columns form, forced: one workgroup per column
    capi.set_columns_form(2)
    plan = capi.ColumnSumsPlan(p, nnz=len(x))
    out["columns_form"] = plan.form
    record("planned_columns", names, plan.column_sums(xt, pt).cpu().numpy())
    plan.close()
    capi.set_columns_form(1)
    # the same columns far apart in a big matrix (the streaming path of the main kernel: long chunks)
    big = 3_000_000
    rng = np.random.default_rng(6)
    filler = np.round(rng.normal(size=big), 2)
    xb = np.concatenate([filler, x, filler])
    pb = np.concatenate([[0], [big], big + p[1:], [2 * big + len(x)]]).astype(np.int32)
    got = capi.column_sums_device(torch.from_numpy(xb).cuda(), torch.from_numpy(pb).cuda()).cpu().numpy()
    record("device_general_inside_6e6", names, got[1:-1])
    # means: sums / nrow
    got = capi.column_sums_device(xt, pt, nrow_for_means=7).cpu().numpy()
    record("device_means_nrow7", names, got)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
