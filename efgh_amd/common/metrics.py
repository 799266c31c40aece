"""Device-side mirror of the reference's `Err` meter (common/helper.py:128-207): same constructor, `flush`, `update(gt, pred)`
and `.dict` (`rot_mean`, `rot_std`, `trs_mean`, `trs_std`), but the per-step errors are computed by `efgh_pose_errors` and kept in
HBM; nothing is copied to the host until `.dict` is read (the reference pays a `.cpu()` of both poses every step)."""
import numpy as np
import torch

from .. import _C
from .._C import c_int32, ptr


class Err(object):
    def __init__(self, dataset, capacity=1 << 16):
        self.dataset = dataset
        self.mode = 1 if dataset == 'KITTI_RAW' else 0     # helper.py:146-149
        self.capacity = capacity
        self._buf = None
        self._n = 0

    def flush(self, keys=None):
        self._n = 0

    def update(self, gt, pred):
        g, p = gt['sensor2_T_sensor1'], pred['sensor2_T_sensor1']
        _C.require_cuda(p)
        g = g.to(device=p.device, dtype=torch.float32)[:1].contiguous()          # sample 0 only, as helper.py:143-144
        p = p.detach().to(torch.float32)[:1].contiguous()
        if self._buf is None:
            self._buf = torch.empty((2, self.capacity), dtype=torch.float32, device=p.device)
        if self._n >= self.capacity:
            nb = torch.empty((2, self.capacity * 2), dtype=torch.float32, device=p.device)
            nb[:, :self.capacity] = self._buf
            self._buf, self.capacity = nb, self.capacity * 2
        _C.check(_C.lib().efgh_pose_errors(ptr(g), ptr(p), c_int32(1), c_int32(self.mode),
                                           _C.c_void_p(self._buf.data_ptr() + 4 * self._n),
                                           _C.c_void_p(self._buf.data_ptr() + 4 * (self.capacity + self._n)),
                                           _C.stream_ptr()))
        self._n += 1

    @property
    def error_dict(self):
        if self._n == 0:
            return {}
        h = self._buf[:, :self._n].cpu().numpy()
        return {'rot': list(h[0]), 'trs': list(h[1])}

    @property
    def dict(self):
        """running mean / population std over the history (np.mean / np.std of the lists, helper.py:158-159)"""
        if self._n == 0:
            return {}
        h = self._buf[:, :self._n].double().cpu().numpy()
        return {'rot_mean': float(np.mean(h[0])), 'rot_std': float(np.std(h[0])),
                'trs_mean': float(np.mean(h[1])), 'trs_std': float(np.std(h[1]))}
