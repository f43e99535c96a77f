"""Development (GPU box): host time of the pieces of a training FORWARD (perf_counter around the Python entry points; the GPU runs
asynchronously, the strided layers' .item() waits are reported apart)."""
import os, sys, time, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from findnpropagate_amd import sparse as S, synthetic as syn
from findnpropagate_amd.backbones_3d import VoxelResBackBone8x
from findnpropagate_amd.spconv import conv as C, norm as N
dev = torch.device("cuda", 0); B = 16
grid = np.round((np.array(syn.POINT_CLOUD_RANGE[3:]) - np.array(syn.POINT_CLOUD_RANGE[:3])) / np.array(syn.VOXEL_SIZE)).astype(int)
net = syn.init_backbone_weights(VoxelResBackBone8x({"USE_BIAS": False, "FNP_DTYPE": "bf16"}, 5, grid), 0).to(dev)
pts, off = syn.make_batch(list(range(B)))
cfg = S.make_voxel_cfg(syn.VOXEL_SIZE, syn.POINT_CLOUD_RANGE, 5, 10, 160000)
vox = S.voxelize(torch.from_numpy(pts).to(dev), torch.from_numpy(off).to(dev), B, cfg)
n = int(vox["n"].item())
bd = lambda: {"voxel_features": vox["mean"][:n], "voxel_coords": vox["coords"][:n].float(), "batch_size": B}
acc = collections.defaultdict(lambda: [0, 0.0])
def wrap(obj, name, label):
    f = getattr(obj, name)
    def g(*a, **k):
        t = time.perf_counter(); r = f(*a, **k); d = time.perf_counter() - t
        acc[label][0] += 1; acc[label][1] += d
        return r
    setattr(obj, name, g)
wrap(S, "pack_weight_train", "pack_weight_train"); wrap(S, "conv_forward", "S.conv_forward"); wrap(S, "rulebook_subm", "rulebook_subm")
wrap(S, "rulebook_strided", "rulebook_strided"); wrap(N, "bn_act", "bn_act")
wrap(C.SparseConvolution, "forward", "SparseConvolution.forward"); wrap(C.SparseConvolution, "prefetch", "prefetch")
wrap(torch.cuda.Event, "synchronize", "Event.synchronize")
orig_item = torch.Tensor.item
def item(self):
    t = time.perf_counter(); r = orig_item(self); acc["item()"][0] += 1; acc["item()"][1] += time.perf_counter() - t; return r
torch.Tensor.item = item
net.train()
for _ in range(3):
    out = net(bd())
torch.cuda.synchronize(); acc.clear()
R = 5
t0 = time.perf_counter()
for _ in range(R):
    out = net(bd())
t1 = time.perf_counter(); torch.cuda.synchronize()
print("forward host ms %.2f" % ((t1 - t0) / R * 1e3))
for k, (c, d) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print("%-28s %5.1f calls/fwd %8.1f us/fwd %7.1f us/call" % (k, c / R, d / R * 1e6, d / c * 1e6))

# phases of the full step, a device synchronisation after each (wall time of the phase alone) and the host's enqueue time
opt = torch.optim.SGD(net.parameters(), lr=1e-4)
def loss_of(out):
    return sum((t.features.float() ** 2).mean() for t in list(out["multi_scale_3d_features"].values()) + [out["encoded_spconv_tensor"]])
for _ in range(2):
    opt.zero_grad(set_to_none=True); loss_of(net(bd())).backward(); opt.step()
torch.cuda.synchronize()
ph = collections.defaultdict(float)
for _ in range(R):
    opt.zero_grad(set_to_none=True)
    t = time.perf_counter(); out = net(bd()); l = loss_of(out); h = time.perf_counter(); torch.cuda.synchronize(); e = time.perf_counter()
    ph["forward+loss host"] += h - t; ph["forward+loss wall"] += e - t
    t = time.perf_counter(); l.backward(); h = time.perf_counter(); torch.cuda.synchronize(); e = time.perf_counter()
    ph["backward host"] += h - t; ph["backward wall"] += e - t
    t = time.perf_counter(); opt.step(); h = time.perf_counter(); torch.cuda.synchronize(); e = time.perf_counter()
    ph["optimizer host"] += h - t; ph["optimizer wall"] += e - t
for k, v in ph.items():
    print("%-22s %.2f ms" % (k, v / R * 1e3))
