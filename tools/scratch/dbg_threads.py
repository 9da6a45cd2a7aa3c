import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from rcppsparse_amd import capi, synth
capi.load()
G=int(sys.argv[1]); nnz=125_000_000*8//G; ncol=1_000_000//G
xs=[];ps=[]
for k in range(G):
    counts = synth.uniform_counts(ncol, nnz, seed=42+k, nrow=10_000_000)
    xt = torch.empty(nnz, dtype=torch.float64, device="cuda"); capi.gen_values_device(xt, seed=42, first_idx=k*nnz, kind=0)
    xs.append(xt); ps.append(torch.from_numpy(synth.offsets_from_counts(counts)).cuda())
torch.cuda.synchronize()
h = capi.MultiDeviceCSC.wrap_device(xs, ps, 10_000_000)
pinned = h.result_buffer()
for gather in ("d2h","stores","none","d2h","none","rccl" if G==1 else "none"):
    h.set_gather(gather)
    ts=[]
    for _ in range(24):
        t0=time.perf_counter(); h.column_sums(out=pinned); ts.append((time.perf_counter()-t0)*1e6)
    print(G, gather, [round(t) for t in ts])
h.close()
