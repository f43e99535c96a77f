"""Development (GPU box): does a hipGraph captured through torch carry an EXTERNAL event record node (torch.cuda.Event(external=True)), can the host
wait for it while the rest of the graph is still running, and does a kernel's store into pinned host memory arrive before that event completes?"""
import ctypes, os, sys, time, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from findnpropagate_amd import lib as L
dev = torch.device("cuda", 0)
lib = L.load()
src = torch.zeros(1, dtype=torch.int32, device=dev)
pin = torch.zeros(16, dtype=torch.int32, pin_memory=True)
x = torch.zeros(1 << 26, device=dev)
arr = (ctypes.c_void_p * 1)(src.data_ptr())
hip = ctypes.CDLL("libamdhip64.so")
hev = ctypes.c_void_p()
assert hip.hipEventCreateWithFlags(ctypes.byref(hev), 0x2) == 0      # hipEventDisableTiming
class _Ev:
    def record(self):
        rc = hip.hipEventRecordWithFlags(hev, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream), 0x1)   # hipEventRecordExternal
        assert rc == 0, rc
    def synchronize(self):
        assert hip.hipEventSynchronize(hev) == 0
ev = _Ev()
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    src.add_(1); torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        src.add_(1)
        L.check(lib.fnp_gather_counts(ctypes.cast(arr, ctypes.c_void_p), 1, 0, ctypes.c_void_p(pin.data_ptr()), L.stream()), "gather")
        ev.record()
        for _ in range(40): x.add_(1.0)
    out = []
    for i in range(6):
        torch.cuda.synchronize()
        t0 = time.perf_counter(); g.replay(); t1 = time.perf_counter()
        ev.synchronize(); t2 = time.perf_counter(); seen = int(pin[0]); torch.cuda.synchronize(); t3 = time.perf_counter()
        out.append({"replay_us": round((t1 - t0) * 1e6), "event_us": round((t2 - t0) * 1e6), "all_us": round((t3 - t0) * 1e6), "pinned_value_at_event": seen, "device_value": int(src.item())})
print(json.dumps(out))
