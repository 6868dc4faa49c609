import sys, torch
sys.path.insert(0, '/root/repo')
import bench
class A: batch, image, queries, levels, dtype = 64, 512, 100, 4, "bf16"
args = A(); dev = torch.device("cuda", 0)
head, shapes = bench.build_head(args, dev)
feats = bench.synth_features(shapes, args.batch, args.image, dev, 1)
def fwd():
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        return head(feats)
for _ in range(3): fwd()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    fwd(); torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    if e.key in ("aten::copy_", "aten::contiguous", "aten::clone", "aten::to", "aten::_to_copy", "aten::cat", "aten::add", "aten::mul", "aten::permute") and e.device_time_total > 50:
        rows.append((e.device_time_total, e.count, e.key, str(e.input_shapes)[:110]))
rows.sort(reverse=True)
for r in rows[:28]:
    print("%8.0f us  x%-4d %-16s %s" % r)
