#!/usr/bin/env python
"""One K19 shape a few times (for rocprofv3 --pmc passes):  python tools/run_one_conv3x3.py I O H [iters]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import mlagg_unet_amd  # noqa: E402,F401
from mlagg_unet_amd import _lib  # noqa: E402

I, O, H = (int(v) for v in sys.argv[1:4])
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 5
lib = _lib.lib()
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
B = 10
x = torch.randn(B, I, H, H, device=dev)
w = torch.randn(O, I, 3, 3, device=dev) * (9 * I) ** -0.5
y = torch.empty(B, O, H, H, device=dev)
ws = torch.empty(lib.mlagg_conv3x3_workspace_bytes(O, I), device=dev, dtype=torch.uint8)
for _ in range(iters):
    _lib.check(lib.mlagg_conv3x3_fwd(x.data_ptr(), I * H * H, w.data_ptr(), 0, None, y.data_ptr(), O * H * H, ws.data_ptr(), B, O, I, H, H, st), "f")
torch.cuda.synchronize()
print("ok", float(y.abs().mean()))
