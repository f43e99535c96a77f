import os, sys, time, tempfile, shutil, json, collections
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from findnpropagate_amd import extract as E, synthetic as syn
from findnpropagate_amd.dense_heads import FrustumProposerOG
from findnpropagate_amd.detectors import Detector3DTemplate
dev = torch.device("cuda", 0)
PARAMS = {'lq': 0.0, 'uq': 0.25, 'cq': 1.0, 'iou_w': 1.0, 'nms_normal': 1.0, 'dst_w': 0.0, 'dns_w': 1.0,
          'min_cam_iou': 0.3, 'score_thr': 0.45, 'nms_2d': 0.4, 'nms_3d': 0.0, 'clamp_bottom': 1, 'num_sizes': 1}
head = FrustumProposerOG(model_cfg={"PARAMS": PARAMS, "PREDS_PATH": "PreprocessedGLIP", "BOX_FORMAT": "xyxy"}, image_detector=lambda bd: bd["dets"]).eval()
acc = collections.defaultdict(float)
def wrap(obj, name, tag):
    f = getattr(obj, name)
    def g(*a, **k):
        t = time.perf_counter(); r = f(*a, **k); acc[tag] += time.perf_counter() - t; return r
    setattr(obj, name, g)
with tempfile.TemporaryDirectory() as warm:
    E.extract_pseudo_labels(syn.SeekerScenes(8, 8, dev), head, warm, dev, write="own")
N = 512
data = syn.SeekerScenes(N, 8, dev)
wrap(head, "launch", "launch (all)"); wrap(head, "enumerate_frustums", "  enumerate_frustums"); _md = FrustumProposerOG._matrices_device
def _mdw(bd, d):
    t = time.perf_counter(); r = _md(bd, d); acc["  _matrices_device"] += time.perf_counter() - t; return r
FrustumProposerOG._matrices_device = staticmethod(_mdw)
wrap(head, "_params", "  _params")
wrap(E, "_records_from_launch", "records_from_launch"); wrap(E, "all_gather_records", "all_gather_records"); wrap(E, "frame_path", "frame_path")
_rc = Detector3DTemplate.recall_counter_vector
def _rcw(*a, **k):
    t = time.perf_counter(); r = _rc(*a, **k); acc["recall_counter_vector"] += time.perf_counter() - t; return r
Detector3DTemplate.recall_counter_vector = staticmethod(_rcw)
wrap(E._Writer, "submit", "writer.submit"); wrap(E, "collate_scenes", "collate")
wrap(type(data), "__getitem__", "dataset[i]")
if len(sys.argv) > 1 and sys.argv[1] == "nowrite":
    E.save_frame = lambda *a, **k: None
out = tempfile.mkdtemp(prefix="fnp_x_")
rec = {}
torch.cuda.synchronize(); t0 = time.perf_counter()
E.extract_pseudo_labels(data, head, out, dev, write="own", recall=rec)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
shutil.rmtree(out, ignore_errors=True)
print(json.dumps({"us_per_scene_total": round(1e6 * dt / N, 1), "sections_us_per_scene": {k: round(1e6 * v / N, 1) for k, v in acc.items()}}))
