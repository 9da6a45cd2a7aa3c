#!/bin/bash
# Run ON THE GPU BOX: the slice-major row-restricted sums with pieces of 64 and of 128 entries (RSP_SLICE_PIECE) against
# the general kernel (bitmap probed in L2), on matrices with few entries per column and slice.  One JSON line per shape
# and piece size: auto / L2 / slices (forced) in ms.    bash tools/slice_piece_sweep.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
for spec in "10000000,4000000,1000000000" "10000000,3000000,1000000000" "30000000,1000000,1000000000" "10000000,2000000,1000000000" \
            "60000000,1000000,1000000000" "10000000,8000000,1000000000" "10000000,1000000,1000000000" "4000000,2000000,200000000"; do
  for piece in 64 128; do
    RSP_SLICE_PIECE=$piece timeout -k 10 200 python3 - "$spec" $piece <<'PY'
import json, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tools"))
import edge_sweep
from rcppsparse_amd import capi
capi.load()
nrow, ncol, nnz = (int(v) for v in sys.argv[1].split(","))
r = edge_sweep.masked_ms(nrow, ncol, nnz, reps=3)
print(json.dumps({"nrow": nrow, "ncol": ncol, "nnz": nnz, "per_column_and_slice": nnz / ncol / -(-nrow // (1 << 20)),
                  "piece": int(sys.argv[2]), "auto_form": r["form"], "auto_ms": r["auto"], "L2_ms": r["L2"], "slices_ms": r["slices"]}), flush=True)
PY
  done
done
