import os, sys, cProfile, pstats, io, torch
sys.path.insert(0, '/root/repo/tools')
from bench_train_full import build
model, step = build()
for _ in range(3): step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(2): step()
torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); ps = pstats.Stats(pr, stream=s).sort_stats('cumulative'); ps.print_stats(45)
out = s.getvalue().splitlines()
for l in out:
    if 'pctrans_amd' in l or 'tottime' in l or 'scipy' in l: print(l[:170])
