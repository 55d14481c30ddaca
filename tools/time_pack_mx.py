import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from centerfusiondetect3d_amd import ops
dev = torch.device("cuda:0")
for B in (1, 6, 8, 16):
    x = torch.relu(torch.randn(B, 112, 200, 64, device=dev))
    for _ in range(3): ops.pack_feat_mx(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.pack_feat_mx(x)
    e1.record(); torch.cuda.synchronize()
    print(B, e0.elapsed_time(e1) / 20 * 1e3, "us")
