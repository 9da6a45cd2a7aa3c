import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from rcppsparse_amd import capi, synth
capi.load()
G=8; nnz=125_000_000; ncol=125_000
xs=[];ps=[]
for k in range(G):
    counts = synth.uniform_counts(ncol, nnz, seed=42+k, nrow=10_000_000)
    xt = torch.empty(nnz, dtype=torch.float64, device="cuda"); capi.gen_values_device(xt, seed=42, first_idx=k*nnz, kind=0)
    xs.append(xt); ps.append(torch.from_numpy(synth.offsets_from_counts(counts)).cuda())
torch.cuda.synchronize()
print("cpu.max", open("/sys/fs/cgroup/cpu.max").read().strip() if os.path.exists("/sys/fs/cgroup/cpu.max") else None, "affinity", len(os.sched_getaffinity(0)))
h = capi.MultiDeviceCSC.wrap_device(xs, ps, 10_000_000)
pinned = h.result_buffer()
N=int(sys.argv[1]) if len(sys.argv)>1 else 1500
for launch in ("serial","workers"):
  for gather in ("none","d2h"):
    h.set_launch(launch); h.set_gather(gather)
    ts=[]
    for _ in range(N):
        h.column_sums(out=pinned); ts.append(h.last_call_stamps()["call_us"])
    ts=np.array(ts); med=np.median(ts)
    bad=np.flatnonzero(ts>1.5*med)
    print(launch, gather, "median", round(med), "max", round(ts.max()), "n>1.5med", bad.size, "at", bad[:10], [round(t) for t in ts[bad[:10]]])
st=open("/sys/fs/cgroup/cpu.stat").read() if os.path.exists("/sys/fs/cgroup/cpu.stat") else ""
print(st.replace("\n"," | "))
h.close()
