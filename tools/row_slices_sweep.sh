for spec in 2e6,1e6,1e9,uniform 4e6,1e6,1e9,uniform 3e7,1e6,1e9,uniform 6e7,1e6,1e9,uniform 1e7,1e5,1e9,uniform 1e7,4e6,1e9,uniform 1e7,1e6,1e9,zipf 1e7,1e6,2e8,uniform; do
  timeout -k 10 200 python tools/check_row_slices.py $spec 5 || echo "FAILED $spec"
done
