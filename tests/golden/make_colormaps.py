"""Writes efgh_amd/common/colormaps.npz: matplotlib's 256-entry `plasma` (image_draw's default, numpy_utils.py:8) and `jet`
(eval_image_draw's, :181) look-up tables as the uint8 triples the reference ends up with (255 * lut -> astype uint8).
Run in the build container (matplotlib 3.10.8 here); the product reads the .npz and never imports matplotlib."""
import os

import matplotlib
matplotlib.use('Agg')
import matplotlib.pyplot as plt
import numpy as np

out = {}
for name in ('plasma', 'jet'):
    cmap = getattr(plt.cm, name)
    x = (np.arange(256) + 0.5) / 256.0                      # one value inside every bin
    out[name] = (255 * cmap(x)[:, :3]).astype('uint8')
    assert cmap.N == 256
dst = os.path.join(os.path.dirname(__file__), '..', '..', 'efgh_amd', 'common', 'colormaps.npz')
np.savez_compressed(dst, **out)
print(dst, {k: v.shape for k, v in out.items()}, 'matplotlib', matplotlib.__version__)
