import sys, numpy as np, os
sys.path.insert(0,'/root/repo')
from vulkansift_amd import api as vk
if os.environ.get("VKLIB"): vk.LIB_PATH = os.environ["VKLIB"]
from oracle import oracle as O
vk.lib().vksift_setLogLevel(1)
imgs = [vk.gen_synthetic_image(7000 + i, 256, 192) for i in range(3)]
cfg = vk.default_config(sift_buffer_count=3, input_image_max_size=256*192)
with vk.Instance(cfg, batch_capacity=3) as inst:
    inst.detectFeaturesBatch(imgs, 0)
    feats = [inst.downloadFeatures(i) for i in range(3)]
for i in range(3):
    ref,_ = O.detect(O.default_config(math_mode=1), imgs[i])
    f = feats[i]
    print(os.environ.get("TAG"), i, len(f), len(ref), [int((f["octave_idx"]==o).sum()) for o in range(-1,4)], [int((ref["octave_idx"]==o).sum()) for o in range(-1,4)])
