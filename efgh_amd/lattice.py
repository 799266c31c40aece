"""Device-side permutohedral lattice pyramid (replaces GenerateData.__call__,
nets/generate_data.py:117-193, which the reference runs on the CPU inside forward)."""
import math

import numpy as np
import torch

from . import _C

EXPECTED_STD = 4 * math.sqrt(2 / 3)            # generate_data.py:19


class LatticeLevel:
    """one pyramid level; in the batched form the arrays cover all samples (sample-major) and
    seg_in / seg hold the per-sample offsets of input points / vertices (host lists, len B+1)"""
    __slots__ = ('n_in', 'H', 'bary', 'off', 'nbr', 'emg', 'pts_next', 'cap', 'H_dev', 'seg_in', 'seg', 'sid')

    def sample(self, b):
        """per-sample view with LOCAL indices, exactly the reference's per-sample arrays"""
        p0, p1, h0, h1 = self.seg_in[b], self.seg_in[b + 1], self.seg[b], self.seg[b + 1]
        out = LatticeLevel()
        out.n_in, out.H = p1 - p0, h1 - h0
        out.bary = self.bary[:, p0:p1]
        out.emg = self.emg[:, p0:p1] if self.emg.shape[0] == 4 and self.emg.dim() == 2 and self.emg.shape[1] == self.n_in \
            else self.emg[p0:p1, :4].t()
        out.off = self.off[:, p0:p1] - h0
        nb = self.nbr[h0:h1]
        out.nbr = torch.where(nb >= 0, nb - h0, nb)
        out.pts_next = self.pts_next[:, h0:h1]
        return out


def build_pyramid(pc, scales, feat_bufs=None, sync=True):
    """pc: (3,N) fp32 CUDA tensor (one sample).  Returns a list of LatticeLevel.

    feat_bufs: optional list of per-level callables / None.  When given, feat_bufs[l](n_in)
    returns a [n_in][C] fp32 buffer whose channels 0..3 receive el_minus_gr directly."""
    _C.require_cuda(pc)
    L = _C.lib()
    dev = pc.device
    assert pc.dim() == 2 and pc.size(0) == 3 and pc.dtype == torch.float32
    pts, cstride, n = pc.contiguous(), pc.size(1), pc.size(1)
    out = []
    st = _C.stream_ptr()
    for l, s in enumerate(scales):
        s = float(s)
        cap = 4 * n
        hcap = L.efgh_lattice_hash_capacity(n)
        lv = LatticeLevel()
        lv.n_in, lv.cap = n, cap
        lv.bary = torch.empty((4, n), dtype=torch.float32, device=dev)
        if feat_bufs is not None and feat_bufs[l] is not None:
            fb = feat_bufs[l](n)
            lv.emg = fb
            emg_ptr, emg_ps, emg_rs = fb, fb.stride(0), 1
        else:
            lv.emg = torch.empty((4, n), dtype=torch.float32, device=dev)
            emg_ptr, emg_ps, emg_rs = lv.emg, 1, n
        lv.off = torch.empty((4, n), dtype=torch.int32, device=dev)
        vkeys = torch.empty((cap, 4), dtype=torch.int32, device=dev)
        lv.pts_next = torch.empty((3, cap), dtype=torch.float32, device=dev)
        minmax = torch.empty(8, dtype=torch.int32, device=dev)
        hkeys = torch.empty(hcap, dtype=torch.int64, device=dev)
        hvals = torch.empty(hcap, dtype=torch.int32, device=dev)
        lv.H_dev = torch.empty(1, dtype=torch.int32, device=dev)
        ws = torch.empty(L.efgh_lattice_workspace_bytes(n), dtype=torch.uint8, device=dev)
        _C.check(L.efgh_lattice_build(
            _C.ptr(pts), _C.c_int64(cstride), _C.c_int32(n), _C.c_float(np.float32(s)),
            _C.c_float(np.float32(EXPECTED_STD * s)), _C.ptr(lv.bary), _C.ptr(emg_ptr),
            _C.c_int64(emg_ps), _C.c_int64(emg_rs), _C.ptr(lv.off), _C.ptr(vkeys), _C.ptr(lv.pts_next),
            _C.ptr(minmax), _C.ptr(hkeys), _C.ptr(hvals), _C.c_int64(hcap), _C.ptr(lv.H_dev),
            _C.ptr(ws), st))
        H = int(lv.H_dev.item())          # one host sync per level (sizes the next level)
        lv.H = H
        lv.nbr = torch.empty((H, 16), dtype=torch.int32, device=dev)
        _C.check(L.efgh_lattice_neighbors(_C.ptr(vkeys), _C.ptr(minmax), _C.ptr(hkeys), _C.ptr(hvals),
                                          _C.c_int64(hcap), _C.ptr(lv.H_dev), _C.c_int32(H),
                                          _C.ptr(lv.nbr), st))
        out.append(lv)
        pts, cstride, n = lv.pts_next, cap, H
    return out


def build_pyramid_batched(pc, scales, feat_bufs=None):
    """pc: (B,3,N) fp32 CUDA tensor.  One launch sequence and ONE host read-back per level for the whole
    batch; every sample keeps its own lattice (own key ranges, own vertex numbering)."""
    _C.require_cuda(pc)
    L = _C.lib()
    dev = pc.device
    B, _, N = pc.shape
    assert pc.dtype == torch.float32 and pc.size(1) == 3
    pts = pc.permute(1, 0, 2).reshape(3, B * N).contiguous()
    cstride, n = B * N, B * N
    sid = torch.arange(B, dtype=torch.int32, device=dev).repeat_interleave(N).contiguous() if B > 1 else None
    seg_in = [b * N for b in range(B + 1)]
    out = []
    st = _C.stream_ptr()
    for l, s in enumerate(scales):
        s = float(s)
        cap = 4 * n
        hcap = L.efgh_lattice_hash_capacity(n)
        lv = LatticeLevel()
        lv.n_in, lv.cap, lv.seg_in, lv.sid = n, cap, seg_in, sid
        lv.bary = torch.empty((4, n), dtype=torch.float32, device=dev)
        if feat_bufs is not None and feat_bufs[l] is not None:
            fb = feat_bufs[l](n)
            lv.emg = fb
            emg_ptr, emg_ps, emg_rs = fb, fb.stride(0), 1
        else:
            lv.emg = torch.empty((4, n), dtype=torch.float32, device=dev)
            emg_ptr, emg_ps, emg_rs = lv.emg, 1, n
        lv.off = torch.empty((4, n), dtype=torch.int32, device=dev)
        vkeys = torch.empty((cap, 4), dtype=torch.int32, device=dev)
        lv.pts_next = torch.empty((3, cap), dtype=torch.float32, device=dev)
        minmax = torch.empty(8 * B, dtype=torch.int32, device=dev)
        hkeys = torch.empty(hcap, dtype=torch.int64, device=dev)
        hvals = torch.empty(hcap, dtype=torch.int32, device=dev)
        info = torch.empty(1 + B, dtype=torch.int32, device=dev)            # [H_total, seg_first[0..B)]
        vsid = torch.empty(cap, dtype=torch.int32, device=dev) if B > 1 else None
        ws = torch.empty(L.efgh_lattice_workspace_bytes(n), dtype=torch.uint8, device=dev)
        _C.check(L.efgh_lattice_build_batched(
            _C.ptr(pts), _C.c_int64(cstride), _C.c_int32(n), _C.c_float(np.float32(s)),
            _C.c_float(np.float32(EXPECTED_STD * s)), _C.ptr(lv.bary), _C.ptr(emg_ptr),
            _C.c_int64(emg_ps), _C.c_int64(emg_rs), _C.ptr(lv.off), _C.ptr(vkeys), _C.ptr(lv.pts_next),
            _C.ptr(minmax), _C.ptr(hkeys), _C.ptr(hvals), _C.c_int64(hcap), _C.ptr(info),
            _C.ptr(ws), _C.ptr(sid), _C.c_int32(B), _C.ptr(vsid),
            _C.c_void_p(info.data_ptr() + 4), st))
        host = info.cpu().tolist()           # the one host sync of this level (sizes the next level)
        H = host[0]
        seg = ([0] if B == 1 else host[1:]) + [H]
        lv.H, lv.seg = H, seg
        lv.H_dev = info[:1]
        lv.nbr = torch.empty((H, 16), dtype=torch.int32, device=dev)
        _C.check(L.efgh_lattice_neighbors_batched(_C.ptr(vkeys), _C.ptr(minmax), _C.ptr(hkeys), _C.ptr(hvals),
                                                  _C.c_int64(hcap), _C.ptr(lv.H_dev), _C.c_int32(H), _C.ptr(lv.nbr),
                                                  _C.ptr(vsid), _C.c_int32(B), st))
        out.append(lv)
        pts, cstride, n, seg_in = lv.pts_next, cap, H, seg
        sid = vsid[:H] if B > 1 else None
    return out
