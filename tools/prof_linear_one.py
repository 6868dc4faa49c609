import sys, torch
sys.path.insert(0, '/root/repo')
from pctrans_amd import fused_ops
rows = 64 * 21760
x = torch.randn(rows, 128, device="cuda")
lin = torch.nn.Linear(128, 1024).cuda()
with torch.no_grad():
    for _ in range(3):
        fused_ops.linear_k128(x, lin.weight, lin.bias, relu=True)
torch.cuda.synchronize()
