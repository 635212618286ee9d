"""Run the production conv kernel through the C ABI (fs2_op_conv_gemm) at the decoder FFN size; meant to be run under
`rocprofv3 --kernel-trace`, whose trace gives the duration of every launch in order."""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fastspeech2_amd import _lib
from tests import ops_binding as ops

R, C, N, k = 30208, 384, 1024, 9
rs = np.random.RandomState(0)
x = torch.from_numpy(rs.normal(size=(R, C)).astype(np.float32)).cuda()
w = torch.from_numpy(rs.uniform(-1, 1, size=(N, C, k)).astype(np.float32) / np.float32(np.sqrt(C * k))).cuda()
b = torch.zeros(N).cuda()
_lib.set_option("FS2_BM", 256)
cases = [("bias+relu", x, w, b, 1), ("nobias", x, w, None, 1), ("noact", x, w, b, 0), ("nobias_noact", x, w, None, 0),
         ("x=0", torch.zeros_like(x), w, b, 1), ("w=0", x, torch.zeros_like(w), b, 1), ("x=small", x * 1e-3, w, b, 1)]
for name, xx, ww, bb, act in cases:
    y, _ = ops.conv_gemm(xx, ww, bb, None, False, None, 1e-5, act, None, None, precision="mix_mx")
    print(name, float(y.abs().mean()))
