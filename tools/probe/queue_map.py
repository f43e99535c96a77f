"""Development (GPU box): which stream-to-hardware-queue arrangements let a pipeline's index chains run under the other batch's convolutions.
torch's pool streams are sorted into hardware-queue classes by the spin test of concurrent_streams; pipelines are then built on chosen classes."""
import os, sys, time, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from findnpropagate_amd import sparse as S, synthetic as syn
from findnpropagate_amd.backbones_3d import VoxelResBackBone8x
from findnpropagate_amd.backbones_3d import spconv_backbone as SB
dev = torch.device("cuda", 0); B = int(sys.argv[1]); ctx = sys.argv[2] if len(sys.argv) > 2 else "fresh"
grid = np.round((np.array(syn.POINT_CLOUD_RANGE[3:]) - np.array(syn.POINT_CLOUD_RANGE[:3])) / np.array(syn.VOXEL_SIZE)).astype(int)
net = syn.init_backbone_weights(VoxelResBackBone8x({"USE_BIAS": False}, 5, grid), 0).to(dev).eval()
pts, off = syn.make_batch(list(range(B))); pts, off = torch.from_numpy(pts).to(dev), torch.from_numpy(off).to(dev)
cfg = S.make_voxel_cfg(syn.VOXEL_SIZE, syn.POINT_CLOUD_RANGE, 5, 10, 160000)
if ctx == "bench":     # what bench.py did before it built its pipeline: two-branch graphs replayed one at a time
    with torch.no_grad():
        for _ in range(5): net.forward_points_graphed(pts, off, B, cfg, probe=True)
    torch.cuda.synchronize()

def aliased(a, b, ticks=200000):
    """is b starved while a runs CHIP-FILLING kernels back to back (12 matrix products of ~0.5 ms)?"""
    if a == b:
        return True
    torch.cuda.synchronize()
    with torch.cuda.stream(a):
        for _ in range(12): torch.mm(aliased.A, aliased.A, out=aliased.C)
    ev = torch.cuda.Event(); t0 = time.perf_counter()
    with torch.cuda.stream(b): aliased.word.fill_(1)
    ev.record(b); ev.synchronize(); w = time.perf_counter() - t0
    torch.cuda.synchronize(); tot = time.perf_counter() - t0
    aliased.log.append((round(w * 1e3, 2), round(tot * 1e3, 2)))
    return w > 0.5 * tot
aliased.log = []
aliased.A = torch.randn(6144, 6144, device=dev, dtype=torch.bfloat16); aliased.C = torch.empty_like(aliased.A)
for _ in range(3): torch.mm(aliased.A, aliased.A, out=aliased.C)
torch.cuda.synchronize()
aliased.word = torch.zeros(1, dtype=torch.int32, device=dev); aliased.word2 = torch.zeros(1, dtype=torch.int32, device=dev); aliased.helper = torch.cuda.Stream(dev)
t0 = time.perf_counter(); torch.cuda._sleep(200000); torch.cuda.synchronize(); spin_ms = (time.perf_counter() - t0) * 1e3
caller = torch.cuda.current_stream(dev)
classes = [[caller]]
pool = [torch.cuda.Stream(dev) for _ in range(14)]
for s in pool:
    for c in classes:
        if aliased(c[0], s):
            c.append(s); break
    else:
        classes.append([s])
out = {"test_ms": aliased.log[:40], "scenes": B, "context": ctx, "spin_ms": round(spin_ms, 2), "classes": [len(c) for c in classes], "class_of_pool": [next(i for i, c in enumerate(classes) if s_ in c) for s_ in pool], "caller_class_size": len(classes[0])}
def q(i, j=0):      # j-th stream of class i (class 0 = the caller's; its member 0 is the caller's stream itself)
    c = classes[i]
    return c[min(j + (1 if i == 0 else 0), len(c) - 1)]
K = 30 if B > 8 else 200
arr = {"all distinct, none with the caller": (1, 2, 3), "conv with the caller": (1, 2, 0), "slot 0 with the caller": (0, 1, 2),
       "both slots in one queue": (1, 1, 2), "slot 0 with conv": (1, 2, 1), "slot 1 with conv": (1, 2, 2), "all three in one queue": (1, 1, 1)}
res = {}
P = [x for c in classes[1:] for x in c]     # pool streams outside the caller's class
if len(P) < 5:
    print(json.dumps(out)); sys.exit(0)
explicit = {"slot 0 IS the caller's stream": [caller, P[0], P[1]], "slot 1 IS the caller's stream": [P[0], caller, P[1]],
            "conv IS the caller's stream": [P[0], P[1], caller], "three pool streams (none the caller's)": [P[2], P[3], P[4]]}
with torch.no_grad():
    for name, st in explicit.items():
        st = st if B >= 8 else st[:2]
        pipe = SB.PointsPipeline(net, B, cfg, depth=2, capacity=(pts.shape[0] + 65535) // 65536 * 65536, streams=st)
        for _ in pipe.map([(pts, off)] * 6): pass
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in pipe.map([(pts, off)] * K): pass
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
        res[name] = round(B / dt)
        del pipe
    for name, (a, b, c) in ({} if len(sys.argv) > 3 else arr).items():
        if max(a, b, c) >= len(classes):
            res[name] = None; continue
        serial = B >= 8
        st = [q(a, 0), q(b, 1 if b == a else 0)] + ([q(c, 2 if (c == a and c == b) else 1 if c in (a, b) else 0)] if serial else [])
        if len(set(id(x) for x in st)) < len(st):
            res[name] = "not enough streams in that class"; continue
        pipe = SB.PointsPipeline(net, B, cfg, depth=2, capacity=(pts.shape[0] + 65535) // 65536 * 65536, streams=st)
        for _ in pipe.map([(pts, off)] * 6): pass
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in pipe.map([(pts, off)] * K): pass
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
        res[name] = round(B / dt)
        del pipe
out["scenes_per_s"] = res
print(json.dumps(out))
