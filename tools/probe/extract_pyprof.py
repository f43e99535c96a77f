"""Development (GPU box): where the host's time goes in the pseudo-label extraction loop (cProfile of the main thread over 512 scenes)."""
import os, sys, cProfile, pstats, io, tempfile, time, shutil
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from findnpropagate_amd import extract as E, synthetic as syn
from findnpropagate_amd.dense_heads import FrustumProposerOG
dev = torch.device("cuda", 0)
PARAMS = {'lq': 0.0, 'uq': 0.25, 'cq': 1.0, 'iou_w': 1.0, 'nms_normal': 1.0, 'dst_w': 0.0, 'dns_w': 1.0,
          'min_cam_iou': 0.3, 'score_thr': 0.45, 'nms_2d': 0.4, 'nms_3d': 0.0, 'clamp_bottom': 1, 'num_sizes': 1}
head = FrustumProposerOG(model_cfg={"PARAMS": PARAMS, "PREDS_PATH": "PreprocessedGLIP", "BOX_FORMAT": "xyxy"}, image_detector=lambda bd: bd["dets"]).eval()
with tempfile.TemporaryDirectory() as warm:
    E.extract_pseudo_labels(syn.SeekerScenes(8, 8, dev), head, warm, dev, write="own")
data = syn.SeekerScenes(512, 8, dev)
for mode in ("time", "profile"):
    out = tempfile.mkdtemp(prefix="fnp_x_")
    rec = {}
    torch.cuda.synchronize(); t0 = time.perf_counter()
    if mode == "profile":
        pr = cProfile.Profile(); pr.enable()
    E.extract_pseudo_labels(data, head, out, dev, write="own", recall=rec)
    if mode == "profile":
        pr.disable()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(mode, "ms per scene %.3f" % (1e3 * dt / 512))
    shutil.rmtree(out, ignore_errors=True)
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(40); print(s.getvalue()[:9000])
