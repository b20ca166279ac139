import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bias_gan_amd
from bias_gan_amd import _lib as L
N, H, W, cin, cout, k = 8, 72, 48, int(os.environ.get("CIN", 728)), int(os.environ.get("COUT", 728)), int(os.environ.get("K", 1))
pad = (k - 1) // 2
x = torch.randn(N, H, W, cin, device="cuda").bfloat16()
w = torch.zeros(cout, k, k, (cin + 63) // 64 * 64, device="cuda", dtype=torch.bfloat16)
w[..., :cin] = (torch.randn(cout, k, k, cin, device="cuda") * 0.05).bfloat16()
y = torch.empty(N, H, W, cout, device="cuda", dtype=torch.bfloat16)
desc = L.ConvDesc(L.BF16, N, H, W, cin, H, W, cout, k, k, 1, pad, 1, cin, cout)
for _ in range(int(os.environ.get("REPS", 10))):
    L.call("bg_conv2d_fwd", desc, x.data_ptr(), w.data_ptr(), None, y.data_ptr())
torch.cuda.synchronize()
