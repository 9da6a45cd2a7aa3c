#!/bin/bash
# Run ON THE GPU BOX: the shapes of profiles/rNN_row_segments.json (tools/check_row_segments.py)
for spec in 3e4,1e5,5e8 6e4,1e5,5e8 1e5,1e3,1e7 1e6,3e4,5e8 4e5,2e4,2e8; do
  timeout -k 10 400 python3 tools/check_row_segments.py $spec 5 || echo "FAILED $spec"
done
