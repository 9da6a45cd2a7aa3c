#!/bin/bash
# Run ON THE GPU BOX: the tall crossprod at 1e6 rows x ncol columns (half the rows stored per column) for widths on
# both sides of the tile counts the panel-table kernel is laid out for -- one JSON line per width into the file given.
#   bash tools/crossprod_widths.sh gpurun_out/xp_widths.jsonl [ncol ...]
out=$1; shift
widths=${@:-"256 257 272 288 304 320 336 352 368 384 385 400 416 432 448 464 480 496 512"}
: > $out
for n in $widths; do
  timeout -k 10 120 python3 ${GRAFT_REPO_ROOT:-/root/repo}/tools/run_crossprod_tall.py $n 5 | grep '^{' >> $out || { echo "width $n failed: stopping"; exit 1; }
done
cat $out
