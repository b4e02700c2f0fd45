import sys, numpy as np, torch
sys.path.insert(0,'/root/repo')
from vulkansift_amd import api as vk, multigpu
from oracle import oracle
vk.load()
a = vk.gen_synthetic_descriptors(51, 33000)
b = vk.gen_synthetic_descriptors(52, 1000)
b[1] = b[0]; b[700] = b[3]; b[999] = b[130]; a[7] = b[3]; a[8] = b[130]
rec = multigpu.hip_match_fn(torch.from_numpy(a).cuda(), 100, torch.from_numpy(b).cuda())
torch.cuda.synchronize()
got = multigpu.records_to_struct(rec.cpu().numpy())
ref = oracle.match_2nn(a, b)
bad = np.flatnonzero((got["idx_b1"] != ref["idx_b1"]) | (got["idx_b2"] != ref["idx_b2"]))
print(len(bad), bad[:20])
for r in bad[:8]:
    print(r, got[r], ref[r])
