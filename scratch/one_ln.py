import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from esmdiff_amd.engine import layernorm_bf16
x = torch.randn(25800, 1536, device="cuda")
w = torch.randn(1536, device="cuda"); b = torch.randn(1536, device="cuda")
for _ in range(5):
    y = layernorm_bf16(x, w, b)
torch.cuda.synchronize()
print("done")
