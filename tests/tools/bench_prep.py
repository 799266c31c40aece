"""sample preparation: GPU (efgh_amd.data) vs the Pillow/numpy path the reference uses, one RELLIS-sized sample"""
import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
import torch
from PIL import Image
from efgh_amd.data import prepare as P
from oracle import prep_oracle as PO

rng = np.random.default_rng(0)
img = rng.integers(0, 256, (1200, 1920, 3), dtype=np.uint8)
n = 220000
pcd = np.empty((n, 4), np.float32); pcd[:, :2] = rng.uniform(-60, 60, (n, 2)); pcd[:, 2] = rng.uniform(-3, 3, n); pcd[:, 3] = .5
gts = P.preproc_gt(0.1, -0.05, 0.3, 0.2, 0.1, -0.3, 0.15)
raw = (900, 1600)
img_d = torch.from_numpy(img).cuda(); pcd_d = torch.from_numpy(pcd).cuda()
for _ in range(3):
    P.preproc_img(img_d, gts, raw, True); P.preproc_pcd(pcd_d, gts, 131072, flip_xy=True)
torch.cuda.synchronize()
t0 = time.perf_counter(); K = 20
for _ in range(K):
    P.preproc_img(img_d, gts, raw, True); P.preproc_pcd(pcd_d, gts, 131072, flip_xy=True)
torch.cuda.synchronize()
gpu = (time.perf_counter() - t0) / K * 1e3
t0 = time.perf_counter()
for _ in range(K):
    P.preproc_img(img, gts, raw, True); P.preproc_pcd(pcd, gts, 131072, flip_xy=True)
torch.cuda.synchronize()
gpu_h2d = (time.perf_counter() - t0) / K * 1e3


def cpu_once():      # what loader_utils.py does, with Pillow itself
    deg = PO.rot_deg_of(gts['rand_init_c'])
    raw_img = np.array(Image.fromarray(img).resize((raw[1], raw[0])))
    rot = PO.crop_image(np.array(Image.fromarray(img).rotate(deg, expand=True)), raw)
    small = np.array(Image.fromarray(rot).resize((raw[1] // 2, raw[0] // 2)))
    PO.zero_pad_image(small, (raw[0] // 2, raw[1] // 2)); PO.image_valid_mask(rot, raw)
    idx = np.random.choice(range(150000), size=131072, replace=False)
    PO.preproc_pcd(pcd * np.array([-1, -1, 1, 1], np.float32), gts, 131072, sampled_indices=idx[idx < 150000])
    return raw_img


t0 = time.perf_counter()
for _ in range(3):
    cpu_once()
cpu = (time.perf_counter() - t0) / 3 * 1e3
print('one RELLIS sample (1200x1920 JPEG-decoded frame -> 900x1600 views + 450x800 input, 220k -> 131072 points): '
      'GPU %.2f ms (inputs resident), %.2f ms incl. the H2D copies; Pillow/numpy on one host core %.1f ms' % (gpu, gpu_h2d, cpu))
