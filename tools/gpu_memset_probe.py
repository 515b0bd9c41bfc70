"""Is hipMemset() of device memory asynchronous to the host on this runtime?  (Round 6: gsr_api.cpp cleared a new control block with hipMemset -- the
null stream -- and launched the first kernel that counts in it on the caller's non-blocking stream.)  Times the call and the synchronize behind it on
a 4 GiB buffer, and checks whether a kernel on a non-blocking stream can overtake the clear.  Prints one JSON line."""
import ctypes as C
import json
import time

import torch

hip = C.CDLL("libamdhip64.so")
hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
n = 4 << 30
buf = torch.empty(n, dtype=torch.uint8, device="cuda:0")
buf.fill_(1)
torch.cuda.synchronize()
out = {}
for rep in range(3):
    t0 = time.perf_counter()
    rc = hip.hipMemset(buf.data_ptr(), 0, n)
    t1 = time.perf_counter()
    hip.hipDeviceSynchronize()
    t2 = time.perf_counter()
    out[f"rep{rep}"] = {"rc": rc, "call_us": round((t1 - t0) * 1e6, 1), "sync_after_us": round((t2 - t1) * 1e6, 1)}
# overtaking: clear (null stream), then at once a tiny fill of the LAST byte on a non-blocking stream; if the clear is still running it overwrites the 1
overtaken = 0
st = torch.cuda.Stream()
for rep in range(20):
    buf.fill_(1)
    torch.cuda.synchronize()
    hip.hipMemset(buf.data_ptr(), 0, n)
    with torch.cuda.stream(st):
        buf[-1:].fill_(7)
    torch.cuda.synchronize()
    overtaken += int(buf[-1].item() != 7)
out["non_blocking_stream_work_overtaken_by_the_clear"] = f"{overtaken} of 20"
out["hipMemset_returns_before_the_clear_is_done"] = bool(out["rep2"]["sync_after_us"] > 10 * max(out["rep2"]["call_us"], 1.0))
print(json.dumps(out))
