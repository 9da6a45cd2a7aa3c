#!/usr/bin/env python3
"""Where does a wavefront of the main kernel spend its time?  Uses the diagnostic build
(make -C rcppsparse_amd/csrc stamps): lane 0 of the first 32768 chunks stamps the 100 MHz
constant clock at kernel entry / after the column search / after the p-window fill / after
the first 4 rows / after the first 8 rows / after the last row / at exit.  Prints medians in
microseconds.  The stamped build is slower than the product; read the shares, not the total."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from bench import build_offsets, SEED


NST = 65536   # kStampChunks of the diagnostic build


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
    taper = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else None   # "permille,rows[,chunk_rows]"
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    L = ctypes.CDLL(os.path.join(here, "rcppsparse_amd", "librcppsparse_hip_stamps.so"))
    vp, i32, i64 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64
    L.rsp_column_sums_workspace_bytes.argtypes = [i32, i64]
    L.rsp_column_sums_workspace_bytes.restype = ctypes.c_size_t
    L.rsp_column_sums_device.argtypes = [vp, vp, i32, i64, vp, vp, ctypes.c_size_t, vp]
    L.rsp_gen_values_device.argtypes = [vp, i64, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_int, vp]
    L.rsp_debug_read_stamps.argtypes = [vp, ctypes.c_int]
    stream = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else None   # "waves,rows"
    if taper:
        L.rsp_debug_set(b"taper_permille", taper[0])
        L.rsp_debug_set(b"taper_rows", taper[1])
        if len(taper) > 2:
            L.rsp_debug_set(b"chunk_rows", taper[2])
    if stream and len(stream) > 2:
        L.rsp_debug_set(b"experiment", stream[2])     # third field of an (otherwise unused) "a,b,experiment" argument
    nrow, ncol, nnz, shape, p = build_offsets(wl, 0)
    pt = torch.from_numpy(p).cuda()
    xt = torch.empty(nnz, dtype=torch.float64, device="cuda")
    L.rsp_gen_values_device(xt.data_ptr(), nnz, SEED, 0, 0, None)
    out = torch.empty(ncol, dtype=torch.float64, device="cuda")
    ws = torch.empty(L.rsp_column_sums_workspace_bytes(ncol, nnz), dtype=torch.uint8, device="cuda")
    for _ in range(5):
        assert L.rsp_column_sums_device(xt.data_ptr(), pt.data_ptr(), ncol, nnz, out.data_ptr(), ws.data_ptr(),
                                        ws.numel(), None) == 0
    torch.cuda.synchronize()
    st = np.zeros(NST * 8, dtype=np.uint64)
    assert L.rsp_debug_read_stamps(st.ctypes.data, st.size) == 0
    st = st.reshape(NST, 8).astype(np.int64)
    widx = np.nonzero(st[:, 6] > 0)[0]
    st = st[widx]
    t0 = st[:, 0].min()
    us = lambda a: float(np.median(a)) / 100.0
    names = ["entry->search done", "search->window filled", "window->first 4 rows", "4 rows->8 rows",
             "8 rows->last row", "last row->exit"]
    print(f"workload {wl}: {len(st)} stamped waves; entry skew (median, max) = "
          f"{us(st[:, 0] - t0):.2f}, {float((st[:, 0] - t0).max()) / 100:.2f} us")
    for k, nm in enumerate(names):
        print(f"  {nm:26s} {us(st[:, k + 1] - st[:, k]):8.2f} us   (p90 {float(np.percentile(st[:, k + 1] - st[:, k], 90)) / 100:.2f})")
    print(f"  wave lifetime              {us(st[:, 6] - st[:, 0]):8.2f} us;  last exit - first entry = "
          f"{float(st[:, 6].max() - t0) / 100:.2f} us")
    # resident waves over time (10 us bins): ramp-up, plateau, tail
    span = int(st[:, 6].max() - t0)
    edges = np.arange(0, span + 1000, 1000)
    starts = np.histogram(st[:, 0] - t0, bins=edges)[0]
    ends = np.histogram(st[:, 6] - t0, bins=edges)[0]
    resident = np.cumsum(starts) - np.cumsum(ends)
    step = max(1, len(resident) // 24)
    print("  resident stamped waves every", step * 10, "us:", " ".join(str(int(v)) for v in resident[::step]))
    # delivered bandwidth over time: a wave consumes its first 8 rows between stamps 2 and 4 and the
    # rest of its chunk, at a steady pace, between stamps 4 and 5 (rows of 1 KiB)
    nbins = 40
    width = span / nbins
    rows = np.zeros(nbins)
    total_rows = (nnz + 127) // 128
    plan = (ctypes.c_int32 * 4)()
    L.rsp_plan_describe.argtypes = [i64, ctypes.POINTER(ctypes.c_int32)]
    assert L.rsp_plan_describe(nnz, plan) == 0       # the chunking in force: body chunks, then the tapered tail
    per_wave = np.where(widx < plan[1], plan[0] / 128.0, plan[2] / 128.0)
    for a, b, share in ((2, 4, None), (4, 5, None)):
        ta, tb = (st[:, a] - t0).astype(np.float64), (st[:, b] - t0).astype(np.float64)
        r = np.minimum(8.0, per_wave) if a == 2 else np.maximum(per_wave - 8.0, 0.0)
        dur = np.maximum(tb - ta, 1.0)
        for k in range(nbins):
            lo, hi = k * width, (k + 1) * width
            ov = np.clip(np.minimum(tb, hi) - np.maximum(ta, lo), 0.0, None)
            rows[k] += float(np.sum(r * ov / dur))
    gbps = rows * 1024.0 / (width / 100.0 * 1e-6) / 1e9
    print(f"  consumed GB/s per {width / 100:.1f} us bin:", " ".join(f"{v:.0f}" for v in gbps))


if __name__ == "__main__":
    main()
