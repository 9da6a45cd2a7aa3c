#!/usr/bin/env python3
"""Which HIP runtime calls wait for the device?  (round 6: the plan-free entries promise never to.)

~20 ms of kernels are queued on a side stream; then ONE runtime call is timed on the host.  A call that returns in
microseconds while the queued work still needs its 20 ms did not wait; a call that takes the 20 ms itself drained the device.
Result on ROCm 7 / MI355X (profiles/r06_runtime_waits.json): hipFree, hipHostFree and hipFreeAsync (of a hipMalloc'ed block)
WAIT; hipMalloc, hipHostMalloc, hipEventCreate, hipEventDestroy do not -- which is why dead plans' allocations are recycled,
not freed, inside a call (csrc/capi.hip auto_plan_free) and why the second sighting of a key may allocate.

    python tools/probe_runtime_waits.py > profiles/r06_runtime_waits.json
"""
import ctypes
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

hip = ctypes.CDLL("libamdhip64.so", mode=ctypes.RTLD_GLOBAL)
big = torch.empty(1 << 28, dtype=torch.float32, device="cuda")
side = torch.cuda.Stream()


def queue_work(n=60):
    with torch.cuda.stream(side):
        for _ in range(n):
            big.add_(1.0)


queue_work(2)
torch.cuda.synchronize()
out = {"what": "host time of one runtime call made while ~20 ms of kernels are queued on another stream", "calls": {}}
for what in ("hipMalloc", "hipHostMalloc", "hipEventCreate", "hipFree", "hipHostFree", "hipFreeAsync", "hipEventDestroy"):
    p = ctypes.c_void_p()
    if what in ("hipFree", "hipFreeAsync"):
        assert hip.hipMalloc(ctypes.byref(p), 12 << 20) == 0
    elif what == "hipHostFree":
        assert hip.hipHostMalloc(ctypes.byref(p), 4096, 0) == 0
    elif what == "hipEventDestroy":
        assert hip.hipEventCreateWithFlags(ctypes.byref(p), 2) == 0
    torch.cuda.synchronize()
    queue_work()
    t0 = time.perf_counter()
    if what == "hipMalloc":
        rc = hip.hipMalloc(ctypes.byref(p), 12 << 20)
    elif what == "hipHostMalloc":
        rc = hip.hipHostMalloc(ctypes.byref(p), 4096, 0)
    elif what == "hipEventCreate":
        rc = hip.hipEventCreate(ctypes.byref(p))
    elif what == "hipFreeAsync":
        rc = hip.hipFreeAsync(p, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    else:
        rc = getattr(hip, what)(p)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    out["calls"][what] = {"rc": rc, "call_ms": round((t1 - t0) * 1e3, 3), "queued_work_still_needed_ms": round((t2 - t1) * 1e3, 3),
                          "waited_for_the_device": (t1 - t0) > 5 * max(t2 - t1, 1e-4)}
print(json.dumps(out, indent=1))
