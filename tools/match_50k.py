import sys, torch
sys.path.insert(0, '/root/repo')
from vulkansift_amd import api, multigpu
api.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
a = torch.from_numpy(api.gen_synthetic_descriptors(1, n)).cuda(); b = torch.from_numpy(api.gen_synthetic_descriptors(2, n)).cuda()
for _ in range(3):
    multigpu.hip_match_fn(a, 0, b)
torch.cuda.synchronize()
