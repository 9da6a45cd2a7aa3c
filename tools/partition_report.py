#!/usr/bin/env python3
"""Shard balance of the column-range partitioner on the BASELINE shapes (CPU only, pure integer):
max / mean nnz per shard for the nnz-balanced split (rsp_partition_columns) and for the naive
equal-column-count split, at G = 2, 4, 8."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from bench import build_offsets
from rcppsparse_amd import capi, sharded


def main():
    for wl in ("c3", "c5", "c5desc"):
        nrow, ncol, nnz, shape, p = build_offsets(wl, 0)
        row = {"workload": wl, "shape": shape, "longest_column": int(np.diff(p).max())}
        for G in (2, 4, 8):
            b_nnz = capi.partition_columns(p, G)
            b_col = np.array([(k * ncol) // G for k in range(G + 1)], dtype=np.int32)
            row[f"G{G}"] = {"nnz_balanced": round(sharded.imbalance(p, b_nnz), 4),
                            "equal_columns": round(sharded.imbalance(p, b_col), 4)}
        print(json.dumps(row))


if __name__ == "__main__":
    main()
