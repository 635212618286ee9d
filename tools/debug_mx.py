import os, sys
import numpy as np, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fastspeech2_amd import _lib
from tests import ops_binding as ops
C, N, k = 384, 1024, 9
rs = np.random.RandomState(0)
w = torch.from_numpy(rs.uniform(-1, 1, size=(N, C, k)).astype(np.float32) / np.float32(np.sqrt(C * k))).cuda()
b = torch.from_numpy(rs.uniform(-0.5, 0.5, size=(N,)).astype(np.float32)).cuda()
for bm in (256, 64):
    _lib.set_option("FS2_BM", bm)
    for R in (517, 4096, 30208):
        for dist in ("normal", "uniform"):
            x = torch.from_numpy((rs.normal(size=(R, C)) if dist == "normal" else rs.uniform(-1, 1, size=(R, C))).astype(np.float32)).cuda()
            ref = F.conv1d(x.t().unsqueeze(0), w, b, padding=4)[0].t()
            for prec in ("mix_mx", "bf16x3"):
                y, _ = ops.conv_gemm(x, w, b, None, False, None, 1e-5, 0, None, None, precision=prec)
                nan = torch.isnan(y)
                err = float((y - ref)[~nan].abs().max()) if (~nan).any() else float("nan")
                rows = nan.any(1).nonzero().flatten()
                print("BM=%d R=%d %s %s: nan %d of %d, rows with nan %d (first %s), max err on the rest %.2e, absmax x %.2f" % (
                    bm, R, dist, prec, int(nan.sum()), nan.numel(), rows.numel(), rows[:3].tolist(), err, float(x.abs().max())))
