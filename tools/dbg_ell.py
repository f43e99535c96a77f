import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from findnpropagate_amd import sparse as S
import test_gpu_ell as T
cuda = torch.device("cuda", 0)
rng = np.random.default_rng(1234)
for cin, cout, dt in ((5, 16, torch.float32), (16, 16, torch.bfloat16)):
    d_idx, n_dev, n, grid, x, w, sc, sh = T._conv_inputs(rng, cuda, cin, cout, dt, n=2000, blob=False)
    table = S.rulebook_subm(d_idx, n_dev, grid, 3)
    ell = S.rulebook_subm_ell(d_idx, n_dev, grid, pool_records=n)
    a = S.conv_forward(x, w, table, n_dev)[:n].float()
    b = S.conv_forward_ell(x, w, ell, n_dev)[:n].float()
    d = (a - b).abs()
    print(cin, cout, "max diff", float(d.max()), "scale", float(a.abs().max()), "rows differing", int((d.max(1).values > 0).sum()), "of", n)
    bad = torch.nonzero(d.max(1).values > 1e-3)[:3, 0].tolist()
    nb = table.nbr[:, :n]
    for r in bad:
        print(" row", r, "valid k", torch.nonzero(nb[:, r] >= 0)[:, 0].tolist(), "a", a[r, :4].tolist(), "b", b[r, :4].tolist())
    # per-column pattern
    print(" cols with diff:", torch.nonzero(d.max(0).values > 1e-3)[:, 0].tolist())
