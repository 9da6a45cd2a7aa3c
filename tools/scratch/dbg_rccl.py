import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import oracle
from rcppsparse_amd import capi, synth
capi.load()
counts = synth.zipf_counts(5000, 600_000, seed=5, nrow=100_000); counts[::17]=0
p = synth.offsets_from_counts(counts); x = synth.gen_values(int(p[-1]), seed=5, kind=0)
ref = oracle.column_sums(x,p)
h = capi.MultiDeviceCSC(x, p, (100000, 5000), devices=[0])
print(h.shard_info(0))
a = h.column_sums(); a2 = h.column_sums()
h.set_gather("stores"); s = h.column_sums()
h.set_gather("rccl"); b = h.column_sums(); b2 = h.column_sums()
h.set_gather("d2h"); c = h.column_sums()
for name, v in (("a",a),("a2",a2),("stores",s),("rccl",b),("rccl2",b2),("d2h again",c)):
    d = np.flatnonzero(v != ref)
    print(name, "cols != oracle bits:", d.size, d[:8], "vs a:", np.flatnonzero(v != a)[:8], np.max(np.abs(v-ref)))
h.close()
