"""Tensor-level wrappers over the C-ABI (include/efgh_hip.h).  torch is used for device memory and
the stream only; every arithmetic op below runs in libefgh_hip.so.  No CPU fallback."""
import ctypes
import os as _os
import threading

import torch

from . import _C
from ._C import c_float, c_int32, c_int64, ptr

ACT_NONE, ACT_RELU, ACT_LEAKY = 0, 1, 2


def _L():
    return _C.lib()


def _st():
    return _C.stream_ptr()


def ceil4(n):
    return (n + 3) // 4 * 4


# ----------------------------------------------------------------------------------------------
# weight packing (cached on the parameter's version counter)
# ----------------------------------------------------------------------------------------------
def _cached(owner, key, versions, fn):
    """memoise `fn()` on the owning tensor / module itself (never on id(): ids are recycled)"""
    store = owner.__dict__.setdefault('_efgh_cache', {})
    ent = store.get(key)
    if ent is not None and ent[0] == versions:
        return ent[1]
    val = fn()
    store[key] = (versions, val)
    return val


class Epoch:
    """content epoch of a set of weights that something rewrites through raw pointers, outside torch's version counters
    (train.FusedAdam on a train.FlatParams): every parameter of the set carries a reference (`_efgh_epoch`), the optimizer bumps
    `n` once per step.  The packed layouts that have a persistent buffer are registered HERE, per owner - two models driven from
    two Python threads never repack (or invalidate) each other's weights."""
    __slots__ = ('n', 'jobs', 'tables', 'repacked_at', '__weakref__')

    def __init__(self):
        self.n = 0
        self.jobs = []           # [(weakref to the weight, key)]: every packed layout of the owner's weights with a persistent buffer
        self.tables = {}         # device index -> (signature, jobs tensor, prefix tensor, total)
        self.repacked_at = {}    # device index -> n of the last batched repack enqueued by repack_stale


HOLDER_GEN = [0]                 # bumped whenever parameters change owner (train.FlatParams): caches of "the owners of this model" key on it
GLOBAL_EPOCH = Epoch()           # weights without an owner (no FlatParams: stock torch optimizers bump the version counters)
_LOCK = threading.RLock()        # guards the Python-side job tables (the kernels themselves are ordered by their streams)


def epoch_of(t):
    return getattr(t, '_efgh_epoch', None) or GLOBAL_EPOCH


def bump_epoch(holder=None):
    """the weights of `holder` (an Epoch; None: every weight without an owner) were rewritten in place"""
    (holder or GLOBAL_EPOCH).n += 1


class _ThreadState(threading.local):
    """per-thread switches (two Python threads may drive two models on one GPU; autograd's device thread is a third):
    train_step - inside a training step (forward: set by EFGHBackbone.forward; backward: GemmLayerFn.backward restores the value
    its forward saw); w2v_wanted - set by GemmLayerFn.forward around its launches when the weight gradient will be asked for"""

    def __init__(self):
        self.train_step = False
        self.w2v_wanted = False
        self.nbt_pending = None        # see bn_tick


TLS = _ThreadState()


def bn_tick(bn):
    """`bn.num_batches_tracked += 1` of a training forward (nn.BatchNorm2d.forward does it per layer: 93 one-element launches per
    EFGHNet forward).  Inside a train.FlatParams step the layer's slot is counted on the host and added once per step; inside an
    EFGHBackbone forward (any optimizer: the reference's own loop) the counters are collected and bumped by ONE multi-tensor add
    when that forward returns (nbt_flush); a sub-network called on its own bumps its counter at once, as before."""
    slot = getattr(bn, '_efgh_nbt', None)          # (train.FlatParams, index)
    if slot is not None and slot[0].collect_ticks:
        slot[0].tick(slot[1])
        return
    t = bn.num_batches_tracked
    if t is None:
        return
    pend = TLS.nbt_pending
    if pend is None or not torch.is_tensor(t):
        bn.num_batches_tracked += 1                # (layers._NbtPair forwards `+=` to its parts, which come back here)
    else:
        ent = pend.get(id(t))
        if ent is None:
            pend[id(t)] = [t, 1]
        else:
            ent[1] += 1


def nbt_collect():
    """start collecting bn_tick() calls of this thread; -> True when the caller owns the collection (and has to nbt_flush())"""
    if TLS.nbt_pending is not None:
        return False
    TLS.nbt_pending = {}
    return True


def nbt_flush():
    pend, TLS.nbt_pending = TLS.nbt_pending, None
    if not pend:
        return
    by_count = {}
    for t, k in pend.values():
        by_count.setdefault((k, t.device, t.dtype), []).append(t)
    with torch.no_grad():
        for (k, _, _), ts in by_count.items():
            torch._foreach_add_(ts, k)


def _ver(*tensors):
    # _efgh_gen: content generation of a buffer this module rewrites in place through raw pointers (packed weights)
    return (tuple(epoch_of(t).n for t in tensors),) + tuple((t._version, t.data_ptr(), getattr(t, '_efgh_gen', 0)) for t in tensors)


BATCH_PACK = True        # every stale packed layout of an owner in ONE launch (tests flip it to compare with per-layout packing)
PACK_TILED = True        # ... through LDS tiles, coalesced on both sides (False: the flat one-lane-per-element kernel)


class _PackJob(ctypes.Structure):
    """mirror of efgh_pack_job (include/efgh_hip.h)"""
    _fields_ = [('W', ctypes.c_void_p), ('Wp', ctypes.c_void_p), ('N', c_int32), ('T', c_int32), ('C', c_int32), ('Np', c_int32),
                ('Cp', c_int32), ('pad', c_int32), ('sn', c_int64), ('sc', c_int64), ('st', c_int64), ('taps', c_int32 * 16)]


def _pack_one(w, buf, prm):
    N, T, C, Np, Cp, sn, sc, st, taps = prm
    tp = (ctypes.c_int32 * 16)(*([int(t) for t in taps] + [0] * (16 - len(taps)))) if taps is not None else None
    _C.check(_L().efgh_pack_weight_padded(ptr(w), ptr(buf), c_int32(N), c_int32(T), c_int32(C), c_int32(Np), c_int32(Cp),
                                          c_int64(sn), c_int64(sc), c_int64(st), tp, _st()))
    buf._efgh_gen = getattr(buf, '_efgh_gen', 0) + 1


def _same_storage(a, b):
    """two _ver() keys of the same tensors that differ at most in the optimizer epoch and the version counters"""
    return len(a) == len(b) and all(x[1:] == y[1:] for x, y in zip(a[1:], b[1:]))


def _repack_all(device, holder):
    """ONE launch re-packs every registered layout of every live weight of `holder` on `device` whose cache entry is stale only
    because the optimizer stepped (train.FusedAdam bumps the owner's Epoch): 279 launches of 5 us per training step otherwise"""
    with _LOCK:
        live, jobs = [], []
        for ref, key in holder.jobs:
            w = ref()
            if w is None or epoch_of(w) is not holder:      # (gone, or re-homed in a FlatParams built after its first forward)
                continue
            live.append((ref, key))
            ent = w.__dict__.get('_efgh_cache', {}).get(key)
            if ent is None or not w.is_cuda or w.device != device:
                continue
            cur = _ver(w)
            # stale, and the buffers are the ones the job table was built for: the optimizer epoch moved (FusedAdam) and / or the
            # version counter did (a stock torch optimizer's in-place update).  A re-allocated weight (another data_ptr) or a
            # re-generated one falls through to pack_weight's own path
            if ent[0] != cur and _same_storage(ent[0], cur):
                jobs.append((w, key, ent[1], cur))
        holder.jobs[:] = live
        if not jobs:
            return
        sig = (PACK_TILED,) + tuple((w.data_ptr(), buf.data_ptr(), key) for w, key, buf, _ in jobs)
        tab = holder.tables.get(device.index)
        if tab is None or tab[0] != sig:
            arr = (_PackJob * len(jobs))()
            prefix, total = [], 0
            for i, (w, key, buf, _) in enumerate(jobs):
                N, T, C, Np, Cp, sn, sc, st, taps = buf._efgh_pack
                j = arr[i]
                j.W, j.Wp, j.N, j.T, j.C, j.Np, j.Cp, j.sn, j.sc, j.st = w.data_ptr(), buf.data_ptr(), N, T, C, Np, Cp, sn, sc, st
                for q in range(16):
                    j.taps[q] = int(taps[q]) if (taps is not None and q < len(taps)) else (q if q < T else 0)
                prefix.append(total)
                total += (Np * T * Cp) if not PACK_TILED else ((Np + 7) // 8) * ((Cp + 31) // 32)
            prefix.append(total)
            raw = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(device)
            pre = torch.tensor(prefix, dtype=torch.int64).to(device)
            tab = holder.tables[device.index] = (sig, raw, pre, total)
        if PACK_TILED:      # (prefix over LDS tiles of 8 rows x 32 channels)
            _C.check(_L().efgh_pack_weight_batched_tiled(ptr(tab[1]), ptr(tab[2]), c_int32(len(jobs)), c_int64(tab[3]), _st()))
        else:
            _C.check(_L().efgh_pack_weight_batched(ptr(tab[1]), ptr(tab[2]), c_int32(len(jobs)), c_int64(tab[3]), _st()))
        for w, key, buf, cur in jobs:
            buf._efgh_gen = getattr(buf, '_efgh_gen', 0) + 1
            w.__dict__['_efgh_cache'][key] = (cur, buf)
        _rewino_all(device, holder, [buf for _, _, buf, _ in jobs])


class _WinoJob(ctypes.Structure):
    """mirror of efgh_wino_pack_job (include/efgh_hip.h)"""
    _fields_ = [('Wp', ctypes.c_void_p), ('U', ctypes.c_void_p), ('N', c_int32), ('C', c_int32), ('kind', c_int32), ('pad', c_int32),
                ('first_block', c_int64)]


BATCH_WINO = True        # the Winograd-domain weights of the re-packed layouts in one launch as well (tests flip it)


def _rewino_all(device, holder, bufs):
    """the Winograd-domain images (wino_weight / wino2d_weight) of the packed layouts `bufs`, which were just rewritten in place,
    recomputed IN PLACE by one launch (48 + 48 launches and as many allocations per training step otherwise); called under _LOCK"""
    if not BATCH_WINO:
        return
    derived = []
    for buf in bufs:
        store = buf.__dict__.get('_efgh_cache')
        if not store:
            continue
        for kind, key in ((0, ('wino',)), (1, ('wino2d',))):
            ent = store.get(key)
            if ent is not None and ent[1].device == device:
                derived.append((buf, key, kind, ent[1]))
    if not derived:
        return
    sig = tuple((buf.data_ptr(), U.data_ptr(), kind) for buf, _, kind, U in derived)
    tab = holder.tables.get(('wino', device.index))
    if tab is None or tab[0] != sig:
        arr = (_WinoJob * len(derived))()
        nb = 0
        for i, (buf, _, kind, U) in enumerate(derived):
            N, C = (U.shape[2], U.shape[0] // 3 * 16) if kind == 0 else (U.shape[1], U.shape[2])
            j = arr[i]
            j.Wp, j.U, j.N, j.C, j.kind, j.first_block = buf.data_ptr(), U.data_ptr(), N, C, kind, nb
            nb += ((3 * C * N if kind == 0 else N * C) + 255) // 256
        raw = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(device)
        tab = holder.tables[('wino', device.index)] = (sig, raw, nb)
    _C.check(_L().efgh_wino_pack_batched(ptr(tab[1]), c_int32(len(derived)), c_int64(tab[2]), _st()))
    for buf, key, _, U in derived:
        buf.__dict__['_efgh_cache'][key] = (_ver(buf), U)


def repack_stale(device, holder=None):
    """Enqueue the batched in-place repack of every packed weight of `holder` (an Epoch; None: the weights without an owner) that
    went stale with the last optimizer step on the CURRENT stream.  Callers that fork work onto several streams
    (nets.EFGHBackbone.forward) call this BEFORE the fork: the repack rewrites persistent buffers that layers on every branch
    read, so it has to be ordered before all of them - left to the first pack_weight() of the step it would run on whichever
    branch stream happens to be enqueued first, unordered with the others."""
    holder = holder or GLOBAL_EPOCH
    if not BATCH_PACK or device is None or device.type != 'cuda':
        return
    if holder.repacked_at.get(device.index) == holder.n:
        if holder is not GLOBAL_EPOCH or not _sentinel_stale(holder, device):
            return
    holder.repacked_at[device.index] = holder.n
    _repack_all(device, holder)


_SENTINELS = {}          # id(holder) -> (len(jobs), [job indices of a few TRAINABLE weights, spread over the list])


def _sentinel_stale(holder, device):
    """weights without an owner are updated by stock torch optimizers, which move version counters, not the holder's epoch: a few
    registered layouts stand for all (an optimizer step rewrites every TRAINABLE parameter) - if one of their keys is stale, the
    whole set is re-packed by the one batched launch instead of layer by layer (145 launches per step of the reference's own
    loop).  The sentinels are chosen among weights with requires_grad (with `grad_false_keys` freezing whole sub-networks, main.py:
    162-183, a frozen weight never goes stale and must not stand for the rest); the choice is cached per job-list length."""
    jobs = holder.jobs
    ent = _SENTINELS.get(id(holder))
    if ent is None or ent[0] != len(jobs) or any(i >= len(jobs) or jobs[i][0]() is None or jobs[i][0]().requires_grad != ent[2] for i in ent[1]):
        train = [i for i, (ref, _) in enumerate(jobs) if ref() is not None and ref().requires_grad]
        if not train:                 # (nothing trainable registered: plain tensors updated in place by the caller - any of them may stand)
            train = [i for i, (ref, _) in enumerate(jobs) if ref() is not None]
        pick = sorted({train[0], train[len(train) // 3], train[2 * len(train) // 3], train[-1]}) if train else []
        ent = _SENTINELS[id(holder)] = (len(jobs), pick, bool(pick) and jobs[pick[0]][0]().requires_grad)
    for i in ent[1]:
        ref, key = jobs[i]
        w = ref()
        if w is None or not w.is_cuda or w.device != device or epoch_of(w) is not holder:
            continue
        c = w.__dict__.get('_efgh_cache', {}).get(key)
        if c is not None and c[0] != _ver(w):
            return True
    return False


def pack_weight(w, N, T, C, sn, sc, st, taps=None, Np=None, Cp=None, key=None):
    """Wp[n][t][c] = w.flat[n*sn + c*sc + taps[t]*st]; optionally zero-padded to (Np, T, Cp).  With a `key` the layout gets a
    persistent buffer on the weight that is re-packed IN PLACE when the weight changes (derived caches see `_efgh_gen` move)."""
    import weakref
    Np = Np or N
    Cp = Cp or C
    prm = (N, T, C, Np, Cp, sn, sc, st, None if taps is None else tuple(int(t) for t in taps))
    wd = w.detach()
    _C.require_cuda(wd)
    if key is None or not wd.is_contiguous():
        out = torch.empty((Np, T, Cp), dtype=torch.float32, device=wd.device)
        _pack_one(wd.contiguous(), out, prm)
        return out
    key = tuple(key)
    store = w.__dict__.setdefault('_efgh_cache', {})
    ent = store.get(key)
    cur = _ver(w)
    if ent is not None and ent[0] == cur:
        return ent[1]
    if ent is None or getattr(ent[1], '_efgh_pack', None) != prm:
        buf = torch.empty((Np, T, Cp), dtype=torch.float32, device=wd.device)
        buf._efgh_pack = prm
        _pack_one(wd, buf, prm)
        store[key] = (cur, buf)
        buf._efgh_holder = epoch_of(w)
        with _LOCK:
            jl = buf._efgh_holder.jobs
            if len(jl) >= 1024 and len(jl) % 256 == 0:      # weights that come and go (per-step temporaries): drop the dead ones
                jl[:] = [j for j in jl if j[0]() is not None]
            jl.append((weakref.ref(w), key))
        return buf
    if BATCH_PACK and _same_storage(ent[0], cur):     # the optimizer stepped: everything else is stale the same way
        _repack_all(wd.device, epoch_of(w))
        ent = store[key]
        if ent[0] == cur:
            return ent[1]
    _pack_one(wd, ent[1], prm)
    store[key] = (cur, ent[1])
    if getattr(ent[1], '_efgh_holder', None) is not epoch_of(w):      # the weight changed owner: register the layout there
        ent[1]._efgh_holder = epoch_of(w)
        with _LOCK:
            ent[1]._efgh_holder.jobs.append((weakref.ref(w), key))
    return ent[1]


def pad_vec(v, Np, fill=0.0):
    if v is None or v.numel() == Np:
        return v
    v = v.detach()
    if not v.is_cuda or v.dtype != torch.float32 or not v.is_contiguous():
        out = torch.full((Np,), fill, dtype=torch.float32, device=v.device)
        out[:v.numel()] = v
        return out
    out = torch.empty((Np,), dtype=torch.float32, device=v.device)
    _C.check(_L().efgh_pad_vec(ptr(v), c_int32(v.numel()), ptr(out), c_int32(Np), c_float(fill), _st()))
    return out


# ----------------------------------------------------------------------------------------------
# gather-GEMM
# ----------------------------------------------------------------------------------------------
def gemm_grid_m(M, N):
    return _L().efgh_gather_gemm_grid_m(c_int64(M), c_int32(N))


USE_WINO = True  # Winograd F(4,3) kernel for the "same" 3x3 convolutions


def wino_eligible(mode, C, N, geom, T=None):
    """the layers efgh_wino_conv3x3 serves (mirror of efgh_wino_supported): 3x3, stride 1, pad 1, C%16 == N%64 == 0"""
    if not USE_WINO or mode != 1 or geom is None or C % 16 or N % 64:
        return False
    (B, Hin, Win, Hv, Wv, sh, sw, dh, dw, Ho, Wo, osh, osw, oh0, ow0) = geom
    return (len(dh) == 9 and (sh, sw, osh, osw, oh0, ow0) == (1, 1, 1, 1, 0, 0) and Hv == Hin == Ho and Wv == Win == Wo
            and list(dh) == [t // 3 - 1 for t in range(9)] and list(dw) == [t % 3 - 1 for t in range(9)])


USE_WINO2D = True  # F(4x4,3x3) as transform / batched GEMM / transform (wino2d.hip)
# channel threshold of the 2-D form.  Rounds 2-4: 256 (forward / data gradient), 128 only for the weight gradient - the 128-channel
# layers were break-even on k_gather_gemm<0>'s planes.  Round 5: with the LDS-DMA staged planes (planes.hip, +8-16 %) the 128-channel
# layers win too - training step 269.8 -> 263.9 ms (+10 GB of kept transformed inputs: 96 GB peak at batch 8), eval forward 41.0 ->
# 39.7 ms per batch of 4 (gpurun_out/r5_ac_*, r5_ad_*; alternating runs on one box).  The full GPU suite is green at 128.
WINO2D_MIN_C = 128           # forward / data gradient
WINO2D_MIN_C_WGRAD = 128     # weight gradient
WINO2D_MIN_C_TRAIN = 128     # forward / data gradient inside a training step (kept separate for A/B runs)


def _pow2(v):
    return v > 0 and (v & (v - 1)) == 0


def wino2d_eligible(mode, C, N, geom, wgrad=False):
    """the layers the 2-D Winograd path serves (mirror of efgh_wino2d_supported + the channel threshold below which the
    transform passes cost more than the saved MFMAs, measured with tools/bench_wino.py): 3x3, stride 1, pad 1, C and N powers
    of two, >= 256 channels (forward / data gradient; >= 128 inside a training step) or >= 128 (weight gradient)"""
    if not USE_WINO2D or not wino_eligible(mode, C, N, geom):
        return False
    lim = WINO2D_MIN_C_WGRAD if wgrad else (min(WINO2D_MIN_C, WINO2D_MIN_C_TRAIN) if TLS.train_step else WINO2D_MIN_C)
    if geom[1] < 8 or geom[2] < 8:          # maps of fewer than 2 x 2 tiles: mostly padding (and nothing to gain)
        return False
    return min(C, N) >= lim and _pow2(C // 4) and _pow2(N // 4) and C <= 512 and N <= 512


def wino2d_weight(Wp, N, C):
    """U = G w G^T of a packed [N][9][C] weight (efgh_wino2d_pack), cached on the packed tensor"""
    def make():
        U = torch.empty((36, N, C), dtype=torch.float32, device=Wp.device)
        _C.check(_L().efgh_wino2d_pack(ptr(Wp), ptr(U), c_int32(N), c_int32(C), _st()))
        return U
    return _cached(Wp, ('wino2d',), _ver(Wp), make)


USE_C4 = True      # dedicated MFMA kernels for the 4-channel input layers (c4conv.hip)


def c4_eligible(mode, C, N, geom, wgrad=False):
    """the layers efgh_c4_conv3x3 / efgh_c4_wgrad serve (mirror of efgh_c4_supported): 3x3, pad 1, stride 1 or 2, C == 4"""
    if not USE_C4 or mode != 1 or geom is None or C != 4:
        return False
    if (N % 16 or not 32 <= N <= 128) if wgrad else N not in (32, 64, 128):
        return False
    (B, Hin, Win, Hv, Wv, sh, sw, dh, dw, Ho, Wo, osh, osw, oh0, ow0) = geom
    return (len(dh) == 9 and sh == sw and sh in (1, 2) and (osh, osw, oh0, ow0) == (1, 1, 0, 0) and Hv == Ho and Wv == Wo
            and list(dh) == [t // 3 - 1 for t in range(9)] and list(dw) == [t % 3 - 1 for t in range(9)])


USE_SMALLC = True      # dedicated kernels for the 16- / 32-channel 3x3 layers (smallc.hip)


SC_MIN_PIXELS_32 = 60000        # per image, so that the choice of kernel (and with it the rounding) does not depend on the batch size


def sc_eligible(mode, C, N, geom, wgrad=False):
    """the layers efgh_sc_conv3x3 / efgh_sc_wgrad serve (mirror of efgh_sc_supported): stride 1, "same" size; 3x3 / pad 1 with C and N in
    {16, 32}, or 1x1 with (C, N) = (64, 32) / (32, 64).  The 32 -> 32 forward / data-gradient launch keeps 144 weight registers per
    lane, which only pays on large maps (tools/bench_smallc.py: 0.092 vs 0.075 ms at 8 x 94 x 322 pixels, 0.244 vs 0.270 at 8 x 190 x 637)"""
    if not USE_SMALLC or mode != 1 or geom is None:
        return False
    (B, Hin, Win, Hv, Wv, sh, sw, dh, dw, Ho, Wo, osh, osw, oh0, ow0) = geom
    if not ((sh, sw, osh, osw, oh0, ow0) == (1, 1, 1, 1, 0, 0) and Hv == Hin == Ho and Wv == Win == Wo):
        return False
    if len(dh) == 1:
        return (C, N) in ((64, 32), (32, 64)) and dh[0] == 0 and dw[0] == 0
    if C == 32 and N == 32 and not wgrad and Hin * Win < SC_MIN_PIXELS_32:
        return False
    return (C in (16, 32) and N in (16, 32) and len(dh) == 9
            and list(dh) == [t // 3 - 1 for t in range(9)] and list(dw) == [t % 3 - 1 for t in range(9)])


def _sc_aligned(d, lda, ldo, residual, ldr, stats):
    """the small-channel kernels move 16-byte vectors on A, out and the residual: a channel slice at an offset that is not a
    multiple of four falls through to the generic kernel (efgh_sc_conv3x3 would refuse it)"""
    ok = (lda % 4 == 0 and ldo % 4 == 0 and d.A % 16 == 0 and d.out % 16 == 0
          and (residual is None or (ldr % 4 == 0 and d.residual % 16 == 0)))
    if not ok and stats is not None:
        # (stats_rows() sized the statistics buffer for the small-channel launch: a silent switch of kernels would misread it)
        raise _C.EfghError('16- / 32-channel layer with BatchNorm statistics on a channel slice that is not 16-byte aligned')
    return ok


def stats_rows(mode, C, N, geom, M):
    """rows of the per-tile BatchNorm statistics buffer the GEMM launch for this layer writes"""
    if sc_eligible(mode, C, N, geom):
        return _L().efgh_sc_stats_rows(c_int32(geom[0]), c_int32(geom[1]), c_int32(geom[2]))
    if c4_eligible(mode, C, N, geom):
        return _L().efgh_c4_stats_rows(c_int32(geom[0]), c_int32(geom[9]), c_int32(geom[10]))
    if wino2d_eligible(mode, C, N, geom):
        return _L().efgh_wino2d_stats_rows(c_int32(geom[0]), c_int32(geom[1]), c_int32(geom[2]), c_int32(N))
    if wino_eligible(mode, C, N, geom):
        return _L().efgh_wino_grid_m(c_int32(geom[0]), c_int32(geom[1]), c_int32(geom[2]))
    return gemm_grid_m(M, N)


def wino_weight(Wp, N, C):
    """U = G w of a packed [N][9][C] weight (efgh_wino_pack), cached on the packed tensor"""
    def make():
        U = torch.empty((3 * C // 16, 6, N, 16), dtype=torch.float32, device=Wp.device)
        _C.check(_L().efgh_wino_pack(ptr(Wp), ptr(U), c_int32(N), c_int32(C), _st()))
        return U
    return _cached(Wp, ('wino',), _ver(Wp), make)


USE_THIN = True

def thin_eligible(mode, C, N, T):
    """shapes served by the VALU "thin" kernels (thin.hip) instead of the MFMA tile"""
    if not USE_THIN or mode != 1 or N % 4 or C % 4:
        return False
    # C == 4 with many outputs (the 3->64 input convs) is faster on the MFMA tile (fwd 1.2x, wgrad 2x, fused BN
    # statistics: tools/bench_thin.py); the VALU form is kept for the narrow ones (4->3 range conv)
    if C == 4 and N <= 16 and T in (1, 2, 4, 9) and T * N * 16 + 4 * 64 * 36 * 4 <= 64 * 1024:
        return True
    return N == 4 and T * C * 16 <= 60 * 1024


def conv_bytes(M, N, T, C):
    """algorithmic bytes of one contraction launch: the input rows and the output rows once, the weights once"""
    return 4.0 * (float(M) * (C + N) + float(N) * T * C)


RIDGE_FLOP_PER_BYTE = 30.0      # 157 TFLOP/s fp32 MFMA over the ~5 TB/s a streaming pass reaches


def hbm_bound(M, N, T, C):
    """a contraction whose arithmetic intensity lies below the ridge of the fp32 MFMA / HBM rooflines is bound by its bytes (the
    1x1 layers, the narrow heads, the point branch's 32-channel layers): bench.py prices it against the HBM roofline
    (`roofline_hbm_convs`), not in the MFMA families"""
    return 2.0 * M * N * T * C / conv_bytes(M, N, T, C) < RIDGE_FLOP_PER_BYTE


PROFILE_DED = None      # bench.py: every launch served by a dedicated small-channel kernel (thin.hip / c4conv.hip / smallc.hip), with its bytes
TRACE_THIN = None       # tests set this to a list: efgh_thin_supported's answer per thin launch
PROFILE_WINO = None     # launches served by the Winograd kernel (else they are listed in PROFILE)
PROFILE = None          # bench.py sets this to a list: (start_event, end_event, algorithmic_flops) per launch


KSPLIT_MAX_ROWS = 16384


def _blur_gemm_ksplit(A, lda, C, Wp, N, M, out, ldo, table, bias, act, slope, a_off, out_off, flops, alias_mask=False):
    """BCL blur on a level with few vertices (M <= 16 k rows: 9-75 workgroups each walking K = 15*C serially, 6-23 TFLOP/s): the
    15 neighbour taps are split over 5 problems of ONE batched launch (3 taps each, their own columns of the neighbour table and
    their own slice of the packed weight), the five partial planes are added - with bias and activation - by efgh_fold_planes."""
    S, Ts = 5, 3
    dev = out.device

    def regroup():
        return Wp.view(N, S, Ts * C).permute(1, 0, 2).contiguous()             # [S][N][Ts*C]
    Wg = _cached(Wp, ('ksplit', S), _ver(Wp), regroup)
    part = _scratch(S * M * N, dev)
    gather_gemm(A, lda, C, Ts, Wg, N, M, part, N, mode=2, table=table, a_off=a_off, batch=(S, 0, N * Ts * C, M * N, Ts),
                flops=flops if flops is not None else 2.0 * M * N * 15 * C, alias_mask=alias_mask)
    b = None
    if bias is not None:
        b = bias if bias.numel() == N else None
        assert b is not None
    _C.check(_L().efgh_fold_planes(ptr(part), c_int32(S), c_int64(M), c_int32(N), ptr(b), c_int32(act), c_float(slope),
                                   _C.c_void_p(out.data_ptr() + 4 * out_off), c_int64(ldo), _st()))


def gather_gemm(A, lda, C, T, Wp, N, M, out, ldo, mode=0, geom=None, table=None, bias=None, scale=None,
                shift=None, residual=None, ldr=0, act=ACT_NONE, slope=0.0, stats=None, a_off=0, out_off=0,
                res_off=0, M_dev=None, flops=None, batch=None, alias_mask=False, bn_bwd=None, pool=False, lazy=None, pre_v=None):
    """See efgh_gemm_desc.  A/out/residual may be addressed with an element offset (channel slices).
    pool: `out` is the 2x2 max-pooled map [B][Ho/2][Wo/2][ldo] (only for launches pool_fusable() accepts: the 2-D Winograd path).
    bn_bwd: a BnSrc (nets/fn.py) - the BatchNorm layer that produced the tensor whose GRADIENT this launch writes; when the
    kernel serving the launch supports it (Winograd F(4,3)) the column sums of that layer's BatchNorm backward are taken in the
    epilogue and the [rows][2][N] partials are RETURNED (else None: the layer runs its own reduction pass).
    lazy: a LazyAct - A is the RAW output of a train-mode BatchNorm layer and act(A*scale + shift) is applied by the consumer (only the
    2-D Winograd input transform can: the caller asks lazy_capable() first).  pre_v: the input transform of the launch, already
    made (wino2d_bwd_transforms: the data gradient of a layer whose draw is never stored); A may then be None."""
    if (lazy is not None or pre_v is not None) and not (M_dev is None and batch is None and lazy_capable(mode, C, N, geom)):
        raise _C.EfghError('gather_gemm(lazy= / pre_v=) on a launch the 2-D Winograd path does not serve (ask lazy_capable() first)')
    if (KSPLIT_MAX_ROWS and mode == 2 and M <= KSPLIT_MAX_ROWS and T == 15 and N % 4 == 0 and T * C >= 1024 and batch is None
            and scale is None and shift is None and residual is None and stats is None and M_dev is None
            and (bias is None or bias.numel() == N)):
        return _blur_gemm_ksplit(A, lda, C, Wp, N, M, out, ldo, table, bias, act, slope, a_off, out_off, flops, alias_mask)
    if PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    bn_stats = None
    d = _C.GemmDesc()
    es = 4
    d.A = (A.data_ptr() + a_off * es) if A is not None else 0
    d.lda, d.C, d.T, d.mode = lda, C, T, mode
    if geom is not None:
        (d.B, d.Hin, d.Win, d.Hv, d.Wv, d.sh, d.sw, dh, dw, d.Ho, d.Wo, d.osh, d.osw, d.oh0, d.ow0) = geom
        for i, (a, b) in enumerate(zip(dh, dw)):
            d.dh[i], d.dw[i] = a, b
    d.table = 0 if table is None else table.data_ptr()
    d.W, d.N, d.M = Wp.data_ptr(), N, M
    d.M_dev = 0 if M_dev is None else M_dev.data_ptr()
    d.bias = 0 if bias is None else bias.data_ptr()
    d.scale = 0 if scale is None else scale.data_ptr()
    d.shift = 0 if shift is None else shift.data_ptr()
    d.residual = 0 if residual is None else residual.data_ptr() + res_off * es
    d.ldr, d.act, d.slope = ldr, act, slope
    d.out = out.data_ptr() + out_off * es
    d.ldo = ldo
    d.stats = 0 if stats is None else stats.data_ptr()
    if batch is not None:
        d.nbatch, d.batch_stride_a, d.batch_stride_w, d.batch_stride_out = batch[:4]
        d.batch_stride_table = batch[4] if len(batch) > 4 else 0
    d.table_alias_mask = 1 if (alias_mask and mode == 2) else 0
    thin = stats is None and M_dev is None and thin_eligible(mode, C, N, T)
    wino = sc = False
    if thin:
        if TRACE_THIN is not None:
            TRACE_THIN.append(int(_L().efgh_thin_supported(ctypes.byref(d))))      # (tests: which form serves the launch)
        _C.check(_L().efgh_thin_gemm(ctypes.byref(d), _st()))
    elif M_dev is None and batch is None and c4_eligible(mode, C, N, geom):
        assert lda % 4 == 0
        thin = True             # (for the profile lists: an HBM-bound launch, not part of the MFMA GEMM family)
        if pool:
            _C.check(_L().efgh_c4_conv3x3_pooled(ctypes.byref(d), _st()))
            pool = False
        else:
            _C.check(_L().efgh_c4_conv3x3(ctypes.byref(d), _st()))
    elif M_dev is None and batch is None and sc_eligible(mode, C, N, geom) and _sc_aligned(d, lda, ldo, residual, ldr, stats):
        sc = True
        _C.check(_L().efgh_sc_conv3x3(ctypes.byref(d), _st()))
    elif M_dev is None and batch is None and wino2d_eligible(mode, C, N, geom):
        wino = '2d'
        if (bn_bwd is not None and stats is None and BN_BWD_FUSED_2D and out_off == 0 and not pool and bn_bwd.fits(M, N)
                and bn_bwd.y is None and bn_bwd.raw.stride(-2) % 4 == 0):
            # the BatchNorm-backward sums of the layer whose activation gradient this launch writes, in its output transform: that
            # layer's reduction pass (a read of dy and of raw) is not run (efgh_gemm_desc.stats_mode 1)
            rows = _L().efgh_wino2d_stats_rows(c_int32(geom[0]), c_int32(geom[1]), c_int32(geom[2]), c_int32(N))
            bn_stats = torch.empty((rows, 2, N), dtype=torch.float32, device=out.device)
            d.stats, d.stats_mode = bn_stats.data_ptr(), 1
            d.bn_raw, d.bn_ldraw = bn_bwd.raw.data_ptr(), bn_bwd.raw.stride(-2)
            d.bn_pscale, d.bn_pshift = bn_bwd.psc.data_ptr(), bn_bwd.psh.data_ptr()
            d.bn_mean, d.bn_invstd = bn_bwd.mean.data_ptr(), bn_bwd.invstd.data_ptr()
            d.bn_act, d.bn_slope = bn_bwd.act, bn_bwd.slope
        _wino2d_forward(d, A, a_off, lda, C, N, geom, Wp, pool=pool, lazy=lazy, pre_v=pre_v)
        pool = False
    elif M_dev is None and batch is None and wino_eligible(mode, C, N, geom) and pool == 'h':
        wino = True
        pool = False
        _C.check(_L().efgh_wino_conv3x3_hpool(ctypes.byref(d), ptr(wino_weight(Wp, N, C)), _st()))
    elif M_dev is None and batch is None and wino_eligible(mode, C, N, geom):
        wino = True
        if bn_bwd is not None and stats is None and BN_BWD_FUSED and out_off == 0 and bn_bwd.fits(M, N):
            rows = _L().efgh_wino_grid_m(c_int32(geom[0]), c_int32(geom[1]), c_int32(geom[2]))
            bn_stats = torch.empty((rows, 2, N), dtype=torch.float32, device=out.device)
            d.stats, d.stats_mode = bn_stats.data_ptr(), 1
            d.bn_raw, d.bn_ldraw = bn_bwd.raw.data_ptr(), bn_bwd.raw.stride(-2)
            if bn_bwd.y is not None:
                d.bn_y, d.bn_ldy = bn_bwd.y.data_ptr(), bn_bwd.y.stride(-2)
            else:
                d.bn_pscale, d.bn_pshift = bn_bwd.psc.data_ptr(), bn_bwd.psh.data_ptr()
            d.bn_mean, d.bn_invstd = bn_bwd.mean.data_ptr(), bn_bwd.invstd.data_ptr()
            d.bn_act, d.bn_slope = bn_bwd.act, bn_bwd.slope
        _C.check(_L().efgh_wino_conv3x3(ctypes.byref(d), ptr(wino_weight(Wp, N, C)), _st()))
    else:
        _C.check(_L().efgh_gather_gemm(ctypes.byref(d), _st()))
    if pool:
        raise _C.EfghError('gather_gemm(pool=True) on a launch no pooling epilogue serves (ask pool_fusable() first)')
    if PROFILE is not None and (thin or (not wino and batch is None and hbm_bound(M, N, T, C))):
        e1.record()
        if PROFILE_THIN is not None:
            PROFILE_THIN.append((e0, e1, conv_bytes(M, N, T, C), (mode, M, N, T, C)))
        if PROFILE_DED is not None and (thin or sc):      # every launch a dedicated kernel (thin / 4-channel / small-channel) served
            PROFILE_DED.append((e0, e1, conv_bytes(M, N, T, C), (mode, M, N, T, C)))
    elif PROFILE is not None:
        e1.record()
        if PROFILE_DED is not None and sc:
            PROFILE_DED.append((e0, e1, conv_bytes(M, N, T, C), (mode, M, N, T, C)))
        rec = (e0, e1, float(flops) if flops is not None else 2.0 * M * N * T * C, (mode, M, N, T, C))
        if wino == '2d' and PROFILE_WINO2D is not None:
            PROFILE_WINO2D.append(rec)
        else:
            (PROFILE_WINO if (wino and PROFILE_WINO is not None) else PROFILE).append(rec)
    return bn_stats


GEMM_DMA = True                 # eligible efgh_gather_gemm launches on the LDS-DMA staged instances (a process-wide switch of the library)


def apply_switches():
    """push the switches that live inside the library (bench.py --set, tools): call after changing GEMM_DMA"""
    _L().efgh_gather_gemm_set_dma(c_int32(1 if GEMM_DMA else 0))


PLANE_DMA = True                # the 36 planes of a 2-D Winograd layer on the LDS-DMA staged kernels (planes.hip); False: k_gather_gemm / k_gather_wgrad
PLANE_DMA_NBUF = 0              # ring slots (0: the library's default; 2 or 3 for A/B runs, tools/bench_planes.py)
PROFILE_WINO2D = None           # bench.py: whole 2-D Winograd layers (three launches), direct-form FLOPs
PROFILE_WINO2D_GEMM = None      # bench.py: their batched GEMM launches alone, EXECUTED FLOPs (2*36*T*C*N)


def _batched_plain_gemm(A3, W3, out3, rows, C, N):
    """out3[a] = A3[a] @ W3[a]^T for the 36 alpha planes: k_gather_gemm, mode 0, one launch"""
    g = _C.GemmDesc()
    g.A, g.lda, g.C, g.T, g.mode = A3.data_ptr(), 36 * C, C, 1, 0          # activations are tile-major [rows][36][C]
    g.W, g.N, g.M = W3.data_ptr(), N, rows
    g.out, g.ldo = out3.data_ptr(), 36 * N
    g.nbatch, g.batch_stride_a, g.batch_stride_w, g.batch_stride_out = 36, C, N * C, N
    if PROFILE_WINO2D_GEMM is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    if PLANE_DMA and _L().efgh_plane_gemm_supported(ctypes.byref(g)):
        _C.check(_L().efgh_plane_gemm(ctypes.byref(g), c_int32(PLANE_DMA_NBUF), _st()))      # LDS-DMA staged (planes.hip)
    else:
        _C.check(_L().efgh_gather_gemm(ctypes.byref(g), _st()))
    if PROFILE_WINO2D_GEMM is not None:
        e1.record()
        PROFILE_WINO2D_GEMM.append((e0, e1, 2.0 * 36 * rows * C * N, (0, rows, N, 36, C)))


W2V_CACHE = {}          # training: B^T x B of a layer's input, kept from the forward for the layer's weight gradient
W2V_KEEP = True


def w2v_clear():
    """drop the transformed inputs of THIS thread's forwards that no backward came for (nets.EFGHBackbone.forward calls this at
    the start of every train-mode forward, train.Trainer at the start of every step: an entry never outlives the step that made
    it, whatever loop drives it).  Entries are tagged with the thread that made them - the backward that pops them runs on
    autograd's device thread - so two threads driving two models do not drop each other's."""
    me = threading.get_ident()
    with _LOCK:
        for k in [k for k, v in W2V_CACHE.items() if v[2] == me]:
            del W2V_CACHE[k]


POOL_FUSED = True               # inference: MaxPool2d(2,2) folded into the producing layer (4-channel input layers, 2-D Winograd output transform)
POOL_HALF = True                # ... and its horizontal half into the 1-D Winograd epilogue (k_wino43<.., HPOOL> + efgh_maxpool_v2)


def pool_fusable(mode, C, N, geom, residual=None, stats=None):
    """can gather_gemm(pool=...) serve this launch?  True: the whole 2x2 window in the producer (4-channel input layers, 2-D Winograd
    path); 'h': the horizontal half (1-D Winograd path; `out` is [B][Ho][Wo/2][ldo], maxpool_v2 follows); False: no"""
    if not (POOL_FUSED and residual is None and stats is None and geom is not None and geom[1] >= 2 and geom[2] >= 2):
        return False
    if thin_eligible(mode, C, N, len(geom[7])):
        return False
    if c4_eligible(mode, C, N, geom):                    # (gather_gemm's order of preference)
        return geom[5] == 1 and geom[6] == 1
    if sc_eligible(mode, C, N, geom):
        return False
    if wino2d_eligible(mode, C, N, geom):
        return True
    if POOL_HALF and wino_eligible(mode, C, N, geom) and geom[2] >= 2:
        return 'h'              # 1-D Winograd: the horizontal half in its epilogue, maxpool_v2 finishes the window
    return False


class LazyAct:
    """a train-mode BatchNorm layer's normalise + activate pass that was NOT run: the tensor carrying this (`_efgh_lazy`) is the layer's
    RAW output, and its single consumer - a 2-D Winograd layer - applies act(raw*scale + shift) inside its input transform
    (efgh_wino2d_input_act): the activation is never written or re-read"""
    __slots__ = ('scale', 'shift', 'act', 'slope')

    def __init__(self, scale, shift, act, slope):
        self.scale, self.shift, self.act, self.slope = scale, shift, act, slope


LAZY_ACT = True                 # producers defer, 2-D Winograd consumers apply (nets/fn.py, nets/layers.py); False: every activation is materialised
W2_BWD_FUSED = True             # BatchNorm backward 'apply' inside the gradient-side transforms of the 2-D Winograd layers (efgh_wino2d_bwd_transforms)
W2_BWD_FUSED_POOL = True        # ... also for the layers whose activation goes straight into MaxPool2d(2,2) (efgh_wino2d_bwd_transforms_pooled)
LAZY_HITS = [0, 0]              # (tests: deferred activations consumed, fused backward transforms run)


def lazy_capable(mode, C, N, geom):
    """does gather_gemm serve this launch on the 2-D Winograd path (whose input transform can apply a pending BatchNorm + activation)?
    (>= 128 channels on both sides: none of the kernels gather_gemm prefers - thin, 4-channel, small-channel - overlaps)"""
    return geom is not None and wino2d_eligible(mode, C, N, geom) and min(C, N) >= 64


def wino2d_input(A, a_off, lda, C, B, H, W, V, lazy=None):
    if lazy is None:
        _C.check(_L().efgh_wino2d_input(_C.c_void_p(A.data_ptr() + 4 * a_off), c_int64(lda), c_int32(C), c_int32(B), c_int32(H),
                                        c_int32(W), ptr(V), _st()))
    else:
        LAZY_HITS[0] += 1
        _C.check(_L().efgh_wino2d_input_act(_C.c_void_p(A.data_ptr() + 4 * a_off), c_int64(lda), c_int32(C), c_int32(B), c_int32(H),
                                            c_int32(W), ptr(lazy.scale), ptr(lazy.shift), c_int32(lazy.act), c_float(lazy.slope),
                                            ptr(V), _st()))


def wino2d_bwd_transforms(dy, lddy, raw, ldraw, ybits, psc, psh, mean, invstd, coef, m1, m2, N, B, H, W, act, slope, want_dres):
    """-> (Vd, Gy, dres | None): see efgh_wino2d_bwd_transforms (include/efgh_hip.h)"""
    T2 = _L().efgh_wino2d_tiles(c_int32(B), c_int32(H), c_int32(W))
    dev = raw.device
    Vd = torch.empty((T2, 36, N), dtype=torch.float32, device=dev)
    Gy = torch.empty((T2, 36, N), dtype=torch.float32, device=dev)
    dres = torch.empty((B, H, W, N), dtype=torch.float32, device=dev) if want_dres else None
    LAZY_HITS[1] += 1
    _C.check(_L().efgh_wino2d_bwd_transforms(ptr(dy), c_int64(lddy), ptr(raw), c_int64(ldraw), ptr(ybits), ptr(psc), ptr(psh), ptr(mean),
                                             ptr(invstd), ptr(coef), ptr(m1), ptr(m2), c_int32(N), c_int32(B), c_int32(H), c_int32(W),
                                             c_int32(act), c_float(slope), ptr(Vd), ptr(Gy), ptr(dres), c_int64(N), _st()))
    return Vd, Gy, dres


def _wino2d_forward(d, A, a_off, lda, C, N, geom, Wp, pool=False, lazy=None, pre_v=None):
    """input transform -> 36 batched GEMMs -> output transform with the layer's epilogue (descriptor d)"""
    B, H, W = geom[0], geom[1], geom[2]
    T = _L().efgh_wino2d_tiles(c_int32(B), c_int32(H), c_int32(W))
    dev = A.device if A is not None else pre_v.device
    Mb = torch.empty((T, 36, N), dtype=torch.float32, device=dev)
    if pre_v is not None:
        assert tuple(pre_v.shape) == (T, 36, C)
        V = pre_v
    else:
        V = torch.empty((T, 36, C), dtype=torch.float32, device=dev)
        wino2d_input(A, a_off, lda, C, B, H, W, V, lazy)
    if pre_v is None and W2V_KEEP and TLS.w2v_wanted and a_off == 0:
        # 2.25x the activation, ~10 GB over the eligible layers at batch 8 (of 288 GB): saves one transform pass per layer and step
        with _LOCK:
            if len(W2V_CACHE) > 512:
                W2V_CACHE.clear()
            W2V_CACHE[(A.data_ptr(), lda, C, B, H, W)] = (V, A._version, threading.get_ident())
    _batched_plain_gemm(V, wino2d_weight(Wp, N, C), Mb, T, C, N)
    if pool:
        _C.check(_L().efgh_wino2d_output_pooled(ptr(Mb), ctypes.byref(d), _st()))
    else:
        _C.check(_L().efgh_wino2d_output(ptr(Mb), ctypes.byref(d), _st()))


# ----------------------------------------------------------------------------------------------
# BatchNorm / elementwise
# ----------------------------------------------------------------------------------------------
def bn_finalize(stats, G, C, count, gamma, beta, rmean, rvar, momentum, eps, save=False):
    dev = stats.device
    scale = torch.empty(C, dtype=torch.float32, device=dev)
    shift = torch.empty(C, dtype=torch.float32, device=dev)
    sm = torch.empty(C, dtype=torch.float32, device=dev) if save else None
    si = torch.empty(C, dtype=torch.float32, device=dev) if save else None
    _C.check(_L().efgh_bn_finalize(ptr(stats), c_int32(G), c_int32(C), ctypes.c_double(count), ptr(gamma), ptr(beta),
                                   ptr(rmean), ptr(rvar), c_float(momentum), c_float(eps), ptr(scale), ptr(shift),
                                   ptr(sm), ptr(si), _st()))
    return scale, shift, sm, si


def col_stats(x, M, C, ld, x_off=0):
    G = _L().efgh_col_stats_groups(c_int64(M))
    stats = torch.empty((G, 2, C), dtype=torch.float32, device=x.device)
    _C.check(_L().efgh_col_stats(_C.c_void_p(x.data_ptr() + 4 * x_off), c_int64(M), c_int32(C), c_int64(ld),
                                 ptr(stats), _st()))
    return stats, G


def scale_shift_act(x, ldx, scale, shift, y, ldy, M, C, act=ACT_NONE, slope=0.0, res=None, ldr=0, x_off=0,
                    y_off=0, res_off=0, bits=None):
    """bits: an int32 tensor of M*C/32 words that receives the sign bits of y (efgh_scale_shift_act_bits; C % 32 == 0)"""
    if bits is not None:
        _C.check(_L().efgh_scale_shift_act_bits(
            _C.c_void_p(x.data_ptr() + 4 * x_off), c_int64(ldx), ptr(scale), ptr(shift),
            _C.c_void_p(0 if res is None else res.data_ptr() + 4 * res_off), c_int64(ldr),
            _C.c_void_p(y.data_ptr() + 4 * y_off), c_int64(ldy), ptr(bits), c_int64(M), c_int32(C), c_int32(act), c_float(slope),
            _st()))
        return
    _C.check(_L().efgh_scale_shift_act(
        _C.c_void_p(x.data_ptr() + 4 * x_off), c_int64(ldx), ptr(scale), ptr(shift),
        _C.c_void_p(0 if res is None else res.data_ptr() + 4 * res_off), c_int64(ldr),
        _C.c_void_p(y.data_ptr() + 4 * y_off), c_int64(ldy), c_int64(M), c_int32(C), c_int32(act), c_float(slope),
        _st()))


def maxpool2(x):
    B, H, W, C = x.shape
    y = torch.empty((B, H // 2, W // 2, C), dtype=torch.float32, device=x.device)
    _C.check(_L().efgh_maxpool2(ptr(x), ptr(y), c_int32(B), c_int32(H), c_int32(W), c_int32(C), _st()))
    return y


def maxpool_v2(x):
    """the vertical half of MaxPool2d(2,2): [B][H][W][C] -> [B][H/2][W][C]"""
    B, H, W, C = x.shape
    y = torch.empty((B, H // 2, W, C), dtype=torch.float32, device=x.device)
    _C.check(_L().efgh_maxpool_v2(ptr(x), ptr(y), c_int32(B), c_int32(H), c_int32(W), c_int32(C), _st()))
    return y


def maxpool2_affine(raw, scale, shift, act, slope=0.0):
    """max pool of act(raw*scale + shift) without materialising the activation; raw [B][H][W][C]"""
    B, H, W, C = raw.shape
    y = torch.empty((B, H // 2, W // 2, C), dtype=torch.float32, device=raw.device)
    _C.check(_L().efgh_maxpool2_affine(ptr(raw), ptr(scale), ptr(shift), c_int32(act), c_float(slope), ptr(y), c_int32(B),
                                       c_int32(H), c_int32(W), c_int32(C), _st()))
    return y


def maxpool2_bwd_affine(raw, scale, shift, act, slope, dy):
    B, H, W, C = raw.shape
    dx = torch.zeros_like(raw) if (H % 2 or W % 2) else torch.empty_like(raw)
    _C.check(_L().efgh_maxpool2_bwd_affine(ptr(raw), ptr(scale), ptr(shift), c_int32(act), c_float(slope), ptr(dy), ptr(dx),
                                           c_int32(B), c_int32(H), c_int32(W), c_int32(C), _st()))
    return dx


def nchw_to_nhwc(x, Cd=None):
    _C.require_cuda(x)
    x = x.contiguous()
    B, Cs = x.shape[0], x.shape[1]
    sp = tuple(x.shape[2:])
    HW = 1
    for s in sp:
        HW *= s
    Cd = Cd or Cs
    y = torch.empty((B,) + sp + (Cd,), dtype=torch.float32, device=x.device)
    _C.check(_L().efgh_nchw_to_nhwc(ptr(x), ptr(y), c_int32(B), c_int32(Cs), c_int64(HW), c_int32(Cd), _st()))
    return y


def nhwc_to_nchw(x, Cs=None):
    B, ld = x.shape[0], x.shape[-1]
    sp = tuple(x.shape[1:-1])
    HW = 1
    for s in sp:
        HW *= s
    Cs = Cs or ld
    y = torch.empty((B, Cs) + sp, dtype=torch.float32, device=x.device)
    _C.check(_L().efgh_nhwc_to_nchw(ptr(x), c_int64(ld), ptr(y), c_int32(B), c_int32(Cs), c_int64(HW), _st()))
    return y


SEGMENT_TWO_STAGE = True        # the per-sample reductions of E / H / G's heads spread over ~1024 workgroups (False: one per 256 columns)


def _segment_ws(C, nseg, device):
    return _scratch((_L().efgh_segment_workspace(c_int32(C), c_int32(nseg)) + 3) // 4, device)


def segment_colmax(x, ld, C, seg, nseg, want_arg=False):
    y = torch.empty((nseg, C), dtype=torch.float32, device=x.device)
    arg = torch.empty((nseg, C), dtype=torch.int32, device=x.device) if want_arg else None
    rows = x.numel() // ld
    if SEGMENT_TWO_STAGE and rows >= 2048:
        _C.check(_L().efgh_segment_colmax_ws(ptr(x), c_int64(ld), c_int32(C), ptr(seg), c_int32(nseg), c_int64(rows), ptr(y), ptr(arg),
                                             ptr(_segment_ws(C, nseg, x.device)), _st()))
    else:
        _C.check(_L().efgh_segment_colmax(ptr(x), c_int64(ld), c_int32(C), ptr(seg), c_int32(nseg), ptr(y), ptr(arg), _st()))
    return y, arg


def segment_colmean(x, ld, C, rows_per_seg, nseg):
    y = torch.empty((nseg, C), dtype=torch.float32, device=x.device)
    if SEGMENT_TWO_STAGE and rows_per_seg >= 512:
        _C.check(_L().efgh_segment_colmean_ws(ptr(x), c_int64(ld), c_int32(C), c_int32(rows_per_seg), c_int32(nseg), ptr(y),
                                              ptr(_segment_ws(C, nseg, x.device)), _st()))
    else:
        _C.check(_L().efgh_segment_colmean(ptr(x), c_int64(ld), c_int32(C), c_int32(rows_per_seg), c_int32(nseg), ptr(y), _st()))
    return y


def heads_to_nchw(x):
    """x (B,H,W,4): channel 0 depth, channels 1-2 mask logits -> g_depth (B,1,H,W), g_mask = softmax (B,2,H,W)"""
    B, H, W, ld = x.shape
    assert ld == 4 and x.is_contiguous()
    depth = torch.empty((B, 1, H, W), dtype=torch.float32, device=x.device)
    mask = torch.empty((B, 2, H, W), dtype=torch.float32, device=x.device)
    _C.check(_L().efgh_heads_to_nchw(ptr(x), ptr(depth), ptr(mask), c_int32(B), c_int64(H * W), _st()))
    return depth, mask


def heads_bwd(mask, dmask, ddepth):
    B, _, H, W = mask.shape
    dx = torch.empty((B, H, W, 4), dtype=torch.float32, device=mask.device)
    _C.check(_L().efgh_heads_bwd(ptr(mask), ptr(dmask) if dmask is not None else None, ptr(ddepth) if ddepth is not None else None,
                                 c_int32(B), c_int64(H * W), ptr(dx), _st()))
    return dx


def softmax2_to_nchw(x):
    B, H, W, ld = x.shape
    y = torch.empty((B, 2, H, W), dtype=torch.float32, device=x.device)
    _C.check(_L().efgh_softmax2_to_nchw(ptr(x), c_int64(ld), ptr(y), c_int32(B), c_int64(H * W), _st()))
    return y


# ----------------------------------------------------------------------------------------------
# BCL splat
# ----------------------------------------------------------------------------------------------
SPLAT_LANES = 0          # 0 = the kernel picks the lane mapping from the mean list length
PROFILE_BCL = None      # bench.py: (start_event, end_event, algorithmic_bytes, what) per BCL index/splat launch


def _bcl_prof(what, nbytes):
    """context manager: event-time the launches inside it (bench.py only)"""
    class _P:
        def __enter__(self_):
            if PROFILE_BCL is not None:
                self_.e0, self_.e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                self_.e0.record()

        def __exit__(self_, *exc):
            if PROFILE_BCL is not None:
                self_.e1.record()
                PROFILE_BCL.append((self_.e0, self_.e1, float(nbytes), what))
    return _P()


def splat_fwd(lv, feat, Cf, use_emg=True, normalize=True):
    """BCL splat of one lattice level (efgh_amd.lattice.LatticeLevel): rows [el_minus_gr (4, from the lattice) | feat[:, :Cf]]
    of the level's n_in points -> splat [H][4 + Cf] (use_emg=False: feat rows only, [H][Cf]), wsum [H].
    normalize=False: args['bcn_use_norm'] = False of the reference (bilateralNN.py:196: no density normalisation)"""
    _C.require_cuda(feat)
    n, H = lv.n_in, lv.H
    assert feat.shape[0] >= n and feat.stride(1) == 1 and feat.stride(0) % 4 == 0 and Cf % 4 == 0
    C = Cf + (4 if use_emg else 0)
    splat = torch.empty((H, C), dtype=torch.float32, device=feat.device)
    wsum = torch.empty((H,), dtype=torch.float32, device=feat.device)
    with _bcl_prof('splat', float(n) * (4 * C + 48) + float(H) * (4 * C + 4)):                 # SURVEY 8d bytes
        _C.check(_L().efgh_splat_gather(ptr(lv.emg_pm if use_emg else None), ptr(feat), c_int64(feat.stride(0)), c_int32(Cf),
                                        ptr(lv.bary_pm), ptr(lv.list), ptr(lv.vseg), c_int32(H), c_int32(max(1, 4 * n // max(H, 1))), c_int32(SPLAT_LANES), c_int32(1 if normalize else 0), ptr(splat),
                                        ptr(wsum), _st()))
    return splat, wsum


def splat_bwd(lv, gsplat, wsum, Cf, gfeat, use_emg=True, normalize=True):
    """gradient of splat_fwd w.r.t. feat[:, :Cf] (el_minus_gr carries none) -> gfeat [n][ld]"""
    C = gsplat.shape[1]
    with _bcl_prof('splat bwd', float(lv.n_in) * (4 * C + 48) + float(lv.H) * (4 * C + 4)):
        _C.check(_L().efgh_splat_bwd(ptr(gsplat), c_int32(C), c_int32(4 if use_emg else 0), ptr(wsum), c_int32(Cf), ptr(lv.bary_pm),
                                     ptr(lv.off_pm), c_int32(lv.n_in), ptr(gfeat), c_int64(gfeat.stride(0)), c_int32(1 if normalize else 0), _st()))


def neighbor_gather_adjoint(lv, src, C):
    """adjoint of the blur's neighbour gather through the level's own table: src [H][15*C] -> [H][C]"""
    from .lattice import ALIAS_CAP, INFO_ALIAS
    dst = torch.empty((lv.H, C), dtype=torch.float32, device=src.device)
    with _bcl_prof('gather bwd', float(lv.H) * (4 * C + 15 * 8 + 4 * C)):
        _C.check(_L().efgh_table_gather_transposed(ptr(src), ptr(lv.nbr), c_int32(lv.H), c_int32(C), ptr(lv.alist),
                                                   _C.c_void_p(lv.info.data_ptr() + 4 * INFO_ALIAS), c_int32(ALIAS_CAP), ptr(dst),
                                                   _st()))
    return dst


# BatchNorm-backward column sums in the epilogue of the Winograd dgrad that produces dy (saves the reduction pass's read of dy).
# OPT-IN: measured on a batch-8 training step it LOSES 6 ms (313.8 vs 307.8 ms, same box, alternating runs): the epilogue's 64
# extra 4-byte loads per thread of the producer's raw output sit at the end of the kernel, behind the MFMA loop, and cost
# k_wino43<true> more than the 1-GB read the reduction pass no longer does.
BN_BWD_FUSED = False
BN_BWD_FUSED_2D = True     # ... in the OUTPUT TRANSFORM of a 2-D Winograd data gradient (an HBM-bound pass that has the gradient tile in registers anyway)


def bwd_finalize_f32(stats, C, count):
    """fold of the [rows][2][C] partials of a stats_mode-1 launch -> (sum_dpre, sum_dpre_xhat) float, (mean, mean) double"""
    dev = stats.device
    s1 = torch.empty(C, dtype=torch.float32, device=dev)
    s2 = torch.empty(C, dtype=torch.float32, device=dev)
    m1 = torch.empty(C, dtype=torch.float64, device=dev)
    m2 = torch.empty(C, dtype=torch.float64, device=dev)
    _C.check(_L().efgh_bwd_finalize_f32(ptr(stats), c_int32(stats.shape[0]), c_int32(C), ctypes.c_double(count), ptr(s1), ptr(s2),
                                        ptr(m1), ptr(m2), _st()))
    return s1, s2, m1, m2


BLUR_DGRAD_FUSED = True
# residual BatchNorm layers keep the activation mask of their backward as sign bits (1/32 of re-reading the activation in both passes)
BN_MASK_BITS = True


def blur_dgrad(lv, draw, C0, w, C):
    """data gradient of the BCL blur (15-neighbour gather + Conv2d(C, C0, (15,1)), bilateralNN.py:240-246) w.r.t. the splatted rows:
    draw [H][C0] -> [H][C].  The neighbour relation of the lattice is symmetric except for the aliased hits the build marks, so
    the adjoint of gather + convolution is the SAME gather-GEMM on the gradient with tap-mirrored weights
    (dx[h] = sum_t W_{inv t}^T draw[nbr[h][t]], inv t = 15 - t): no [H][15 C] intermediate is written and read back
    (538 MB at level 0 of a batch of 8).  The handful of aliased hits is added by efgh_blur_dgrad_alias in a fixed order."""
    from .lattice import ALIAS_CAP, INFO_ALIAS
    H = lv.H
    inv = [0] + [15 - t for t in range(1, 15)]
    Wd = pack_weight(w, C, 15, C0, 15, C * 15, 1, inv, key=('blur0_d',))
    dx = torch.empty((H, C), dtype=torch.float32, device=draw.device)
    gather_gemm(draw, draw.stride(0), C0, 15, Wd, C, H, dx, C, mode=2, table=lv.nbr, alias_mask=True,
                flops=2.0 * H * 15 * C * C0)
    if getattr(lv, 'n_alias', 1) == 0:          # (known on the host since the pyramid's read-back)
        return dx
    wd = w.detach()
    assert wd.is_contiguous() and wd.numel() == C0 * C * 15
    _C.check(_L().efgh_blur_dgrad_alias(ptr(draw), c_int64(draw.stride(0)), c_int32(C0), ptr(wd), c_int32(C), ptr(lv.alist),
                                        _C.c_void_p(lv.info.data_ptr() + 4 * INFO_ALIAS), c_int32(ALIAS_CAP), ptr(dx), c_int64(C),
                                        _st()))
    return dx


# ----------------------------------------------------------------------------------------------
# rasterisers, rotate
# ----------------------------------------------------------------------------------------------
def range_image(pc, e_l, H, W, fov_up, fov_down):
    """pc (B,3,N), e_l (B,4,4) -> img [B][H][W][4], pix (B,N)"""
    _C.require_cuda(pc, e_l)
    pc, e_l = pc.contiguous(), e_l.detach().contiguous()
    B, _, N = pc.shape
    dev = pc.device
    pix = torch.empty((B, N), dtype=torch.int32, device=dev)
    vals = torch.empty((B, N, 4), dtype=torch.float32, device=dev)
    win = torch.empty((B, H * W), dtype=torch.int32, device=dev)
    img = torch.empty((B, H, W, 4), dtype=torch.float32, device=dev)
    _C.check(_L().efgh_range_image(ptr(pc), ptr(e_l), c_int32(B), c_int32(N), c_int32(H), c_int32(W),
                                   ctypes.c_double(fov_up), ctypes.c_double(fov_down), ptr(pix), ptr(vals),
                                   ptr(win), ptr(img), _st()))
    return img, pix


def depth_image(pc, cam_T_velo, H, W):
    _C.require_cuda(pc, cam_T_velo)
    pc, P = pc.contiguous(), cam_T_velo.detach().contiguous()
    B, _, N = pc.shape
    dev = pc.device
    pix = torch.empty((B, N), dtype=torch.int32, device=dev)
    vals = torch.empty((B, N, 4), dtype=torch.float32, device=dev)
    win = torch.empty((B, H * W), dtype=torch.int32, device=dev)
    img = torch.empty((B, H, W, 4), dtype=torch.float32, device=dev)
    _C.check(_L().efgh_depth_image(ptr(pc), ptr(P), c_int32(B), c_int32(N), c_int32(H), c_int32(W), ptr(pix),
                                   ptr(vals), ptr(win), ptr(img), _st()))
    return img, pix


def rotate_nearest_u8(img, rot_deg, want_nchw=True, want_nhwc4=True):
    _C.require_cuda(img, rot_deg)
    img = img.detach().contiguous()
    B, _, H, W = img.shape
    o1 = torch.empty((B, 3, H, W), dtype=torch.float32, device=img.device) if want_nchw else None
    o2 = torch.empty((B, H, W, 4), dtype=torch.float32, device=img.device) if want_nhwc4 else None
    _C.check(_L().efgh_rotate_nearest_u8(ptr(img), ptr(rot_deg.detach().contiguous().float()), c_int32(B), c_int32(H),
                                         c_int32(W), ptr(o1), ptr(o2), _st()))
    return o1, o2


# ----------------------------------------------------------------------------------------------
# F correlation head
# ----------------------------------------------------------------------------------------------
def minmax(x):
    """x [B][...] -> (B,2) per-sample (min,max)"""
    B = x.shape[0]
    n = x.numel() // B
    G = _L().efgh_minmax_groups(c_int64(n))
    part = torch.empty((B, G, 2), dtype=torch.float32, device=x.device)
    mm = torch.empty((B, 2), dtype=torch.float32, device=x.device)
    _C.check(_L().efgh_minmax(ptr(x), c_int32(B), c_int64(n), ptr(part), ptr(mm), _st()))
    return mm


USE_MFMA_CORR = True


def corr_head(cam, rng, want_logit=False, want_aux=False):
    """cam [B][h][wc][16], rng [B][h][wr][16] -> f_score (B, wr + 2*(wr//8) - wc + 1)"""
    B, h, wc, C = cam.shape
    wr = rng.shape[2]
    assert C == 16 and rng.shape[1] == h
    off = int(wr / 8)
    wp = wr + 2 * off
    nj = wp - wc + 1
    dev = cam.device
    cam_mm, rng_mm = minmax(cam), minmax(rng)
    score = torch.empty((B, nj), dtype=torch.float32, device=dev)
    logit = torch.empty((B, nj), dtype=torch.float32, device=dev) if want_logit else None
    segw = (wc + 31) // 32
    nseg = ceil4((wc + segw - 1) // segw)
    # training (want_aux) takes the MFMA path whenever its backward does (corr1d_bwd): rp then has pitch wpitch, not wp
    mfma = USE_MFMA_CORR and (not want_aux or h * 16 >= 128) and h * segw * 16 < 65536 and nseg <= 32
    wpitch = wp + segw if mfma else wp
    rp = torch.empty((B, h, wpitch, C), dtype=torch.float32, device=dev)
    _C.check(_L().efgh_corr_pad(ptr(rng), ptr(rng_mm), c_int32(B), c_int32(h), c_int32(wr), c_int32(C),
                                c_int32(off), c_int32(wpitch), ptr(rp), _st()))
    if mfma:
        # MFMA formulation: camera-row segments become the N dimension, groups of image rows the batch (split-K)
        nsplit = max(d for d in range(1, 17) if h % d == 0)
        T = h // nsplit
        nseg_real = (wc + segw - 1) // segw
        Wc = torch.empty((B, nsplit, nseg, T, segw * 16), dtype=torch.float32, device=dev)
        _C.check(_L().efgh_corr_pack_cam(ptr(cam), ptr(cam_mm), c_int32(B), c_int32(h), c_int32(wc), c_int32(segw),
                                         c_int32(nseg), c_int32(nsplit), ptr(Wc), _st()))
        P = torch.empty((B, nsplit, wp, nseg), dtype=torch.float32, device=dev)
        geom = (1, T, wpitch, 1, wp, 1, 1, [], [], 1, wp, 1, 1, 0, 0)
        gather_gemm(rp, 16, segw * 16, T, Wc, nseg, wp, P, nseg, mode=3, geom=geom, flops=2.0 * B * nj * h * wc * 16,
                    batch=(B * nsplit, T * wpitch * 16, nseg * T * segw * 16, wp * nseg))
        _C.check(_L().efgh_corr_fold(ptr(P), c_int32(B), c_int32(nsplit), c_int64(wp), c_int32(nseg),
                                     c_int32(nseg_real), c_int32(segw), c_int32(nj), ptr(logit), ptr(score), _st()))
        if want_aux:
            return score, logit, rp, cam_mm, rng_mm
        return score, logit
    part = torch.empty((B, h, nj), dtype=torch.float32, device=dev)
    _C.check(_L().efgh_corr1d(ptr(rp), ptr(cam), ptr(cam_mm), c_int32(B), c_int32(h), c_int32(wc), c_int32(wp),
                              ptr(part), ptr(logit), ptr(score), _st()))
    if want_aux:
        return score, logit, rp, cam_mm, rng_mm
    return score, logit


# ----------------------------------------------------------------------------------------------
# backward
# ----------------------------------------------------------------------------------------------
PROFILE_WGRAD = None
PROFILE_WINO_WGRAD = None
PROFILE_THIN = None
USE_WINO_WGRAD = True


_SCRATCH = {}
WGRAD_SIDE = _os.environ.get('EFGH_WGRAD_STREAM', '1') != '0'


def wgrad_stream(device):
    """the stream weight gradients are launched on when they are written straight into a flat gradient buffer (nets/fn.py):
    train.Trainer joins it before the optimizer.  Together with the backbone's two side streams and the current stream that makes
    four: HIP multiplexes streams onto 4 hardware queues, and a fifth stream in the process makes two of them share a queue -
    measured: +12 ms per training step from merely having created it."""
    from .nets import efghbackbone as bb
    return bb._side_stream(device, 2)


def reserve_comm_queue():
    """data-parallel training (train.Trainer with world > 1): the collective library brings its own stream, and HIP multiplexes
    streams onto four hardware queues - a fifth stream makes two of them share a queue (measured: +12 ms per step from merely
    having created it).  The weight gradients therefore go back onto the stream of their layer's backward, which leaves the
    current stream + two branch streams + the communication stream.  EFGH_WGRAD_STREAM=1 in the environment keeps the fourth
    compute stream regardless."""
    global WGRAD_SIDE
    if _os.environ.get('EFGH_WGRAD_STREAM') is None:
        WGRAD_SIDE = False


def side_streams():
    """every side stream this package has created (branch streams of the backbone + the weight-gradient stream)"""
    from .nets import efghbackbone as bb
    return list(bb._SIDE.values())
DETERMINISTIC = False     # (kept for callers that set it; a no-op since round 4: see gather_wgrad)


def _scratch(nfloats, device):
    """transient fp32 workspace (row-chunk partials of a weight gradient), grown on demand and shared by all launches of a
    stream: a partial plane set is written and folded by consecutive launches of ONE C-ABI call on the current stream"""
    nfloats = int(nfloats)
    if nfloats <= 0:
        return None
    key = (device.index, _C.stream_ptr().value, threading.get_ident())      # (autograd's device thread is its own)
    t = _SCRATCH.get(key)
    if t is None or t.numel() < nfloats:
        t = _SCRATCH[key] = torch.empty(max(nfloats, 1 << 22), dtype=torch.float32, device=device)
    return t


FOLD_UNPACK = True       # a split weight gradient's final fold writes the reference layout itself (no packed plane, no unpack launch)
FOLD_UNPACK_HITS = [0]   # (tests: how many weight gradients took that route)


def wgrad_lazy_capable(mode, C, N, geom):
    """does gather_wgrad serve this launch on the 2-D Winograd path?"""
    return bool(USE_WINO_WGRAD and geom is not None and wino2d_eligible(mode, C, N, geom, wgrad=True) and min(C, N) >= 64)


def gather_wgrad(A, lda, C, T, N, M, G, ldg, dWp, mode=0, geom=None, table=None, unpack=None, lazy=None, pre_gy=None):
    """unpack: (dW, N_real, T, C_real, Cp, sn, sc, st, taps, accumulate) - where the gradient belongs in the reference layout
    (ops.unpack_weight's arguments).  -> True when the launch left it there itself (its final fold did the unpack: dWp is then
    NOT written), False when dWp holds the packed gradient and the caller has to unpack it.
    lazy: A is a raw BatchNorm output with a pending activation (LazyAct; 2-D Winograd path only: the kept forward transform already has
    it applied, a re-transform applies it again).  pre_gy: the gradient-side transform, already made (wino2d_bwd_transforms); G may be None"""
    if (lazy is not None or pre_gy is not None) and not wgrad_lazy_capable(mode, C, N, geom):
        raise _C.EfghError('gather_wgrad(lazy= / pre_gy=) on a launch the 2-D Winograd weight gradient does not serve')
    od = None                # efgh_wgrad_out_desc: the entry points that can leave the reference layout themselves take it explicitly
    if unpack is not None and FOLD_UNPACK and unpack[4] % 4 == 0 and unpack[2] <= 16:
        dW_, n_, t_, c_, cp_, sn_, sc_, st_, taps_, acc_ = unpack
        od = _C.WgradOutDesc()
        od.W, od.N, od.T, od.C, od.Cp, od.sn, od.sc, od.st = dW_.data_ptr(), n_, t_, c_, cp_, sn_, sc_, st_
        for i_, t__ in enumerate(taps_):
            od.taps[i_] = int(t__)
        od.accumulate = 1 if acc_ else 0
    odp = None if od is None else ctypes.byref(od)
    done = False
    if PROFILE_WGRAD is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    d = _C.GemmDesc()
    d.A = A.data_ptr()
    d.lda, d.C, d.T, d.mode = lda, C, T, mode
    if geom is not None:
        (d.B, d.Hin, d.Win, d.Hv, d.Wv, d.sh, d.sw, dh, dw, d.Ho, d.Wo, d.osh, d.osw, d.oh0, d.ow0) = geom
        for i, (a, b) in enumerate(zip(dh, dw)):
            d.dh[i], d.dw[i] = a, b
    d.table = 0 if table is None else table.data_ptr()
    d.N, d.M = N, M
    # (round 4: no weight-gradient kernel combines partial sums with atomics any more - the thin and the stride-2 4-channel kernels
    # leave per-workgroup / per-wave partial planes that are folded in a fixed order, like every other one: all 353 gradients of a
    # training step are bit-reproducible run to run by default, and EFGH_DETERMINISTIC has nothing left to switch)
    nq = N // 4
    thin = thin_eligible(mode, C, N, T) and (N == 4 or (T in (1, 2, 4, 9) and nq & (nq - 1) == 0))
    wino = sc = False
    if (C == 4 and N == 4 and T == 9 and mode == 1 and lda % 4 == 0 and ldg % 4 == 0 and dWp.data_ptr() % 16 == 0
            and _L().efgh_c4n4_supported(ctypes.byref(d))):
        # the 1- / 2-channel 3x3 convolutions behind G's transposed heads: column-walking stencil, per-workgroup partial planes folded
        # in a fixed order (also under EFGH_DETERMINISTIC: no atomics)
        thin = True             # (profile lists: an HBM-bound launch)
        done = _C.check_wrote(_L().efgh_c4n4_wgrad(ctypes.byref(d), ptr(G), c_int64(ldg), ptr(dWp),
                                                   ptr(_scratch(_L().efgh_c4n4_wgrad_workspace(ctypes.byref(d)), dWp.device)), odp, _st()))
    elif thin:
        done = _C.check_wrote(_L().efgh_thin_wgrad(ctypes.byref(d), ptr(G), c_int64(ldg), ptr(dWp),
                                                   ptr(_scratch(_L().efgh_thin_wgrad_workspace(ctypes.byref(d)), dWp.device)), odp, _st()))
    elif (USE_SMALLC and C == 4 and N in (32, 64) and lda % 4 == 0 and ldg % 4 == 0 and d.A % 16 == 0 and G.data_ptr() % 16 == 0
          and geom is not None and geom[5] == 1 and sc_eligible(mode, 16, 16, geom, wgrad=True)):
        # 4-channel input layers at stride 1: the small-channel weight-gradient kernel (G staged by 16-byte loads, per-wave partial
        # planes folded in a fixed order) instead of k_c4_wgrad (4-byte G loads, fp32 atomics)
        thin = True             # (profile lists: an HBM-bound launch)
        done = _C.check_wrote(_L().efgh_sc_wgrad(ctypes.byref(d), ptr(G), c_int64(ldg), ptr(dWp),
                                                 ptr(_scratch(_L().efgh_sc_wgrad_workspace(ctypes.byref(d)), dWp.device)), odp, _st()))
    elif c4_eligible(mode, C, N, geom, wgrad=True):
        thin = True             # (profile lists, as above)
        done = _C.check_wrote(_L().efgh_c4_wgrad(ctypes.byref(d), ptr(G), c_int64(ldg), ptr(dWp),
                                                 ptr(_scratch(_L().efgh_c4_wgrad_workspace(ctypes.byref(d)), dWp.device)), odp, _st()))
    elif sc_eligible(mode, C, N, geom, wgrad=True) and lda % 4 == 0 and ldg % 4 == 0 and d.A % 16 == 0 and G.data_ptr() % 16 == 0:
        sc = True
        done = _C.check_wrote(_L().efgh_sc_wgrad(ctypes.byref(d), ptr(G), c_int64(ldg), ptr(dWp),
                                                 ptr(_scratch(_L().efgh_sc_wgrad_workspace(ctypes.byref(d)), dWp.device)), odp, _st()))
    elif USE_WINO_WGRAD and wino2d_eligible(mode, C, N, geom, wgrad=True):
        wino = '2d'
        B, H, W = geom[0], geom[1], geom[2]
        T2 = _L().efgh_wino2d_tiles(c_int32(B), c_int32(H), c_int32(W))
        dev = dWp.device
        with _LOCK:
            kept = W2V_CACHE.pop((A.data_ptr(), lda, C, B, H, W), None)
        if kept is not None:
            kept[0].record_stream(torch.cuda.current_stream())          # (made on the forward's stream, maybe read on another)
        S = torch.empty((36, N, C), dtype=torch.float32, device=dev)
        if kept is not None and kept[1] == A._version and kept[0].shape == (T2, 36, C):
            V = kept[0]                                                            # B^T x B from the forward pass
        else:
            V = torch.empty((T2, 36, C), dtype=torch.float32, device=dev)
            wino2d_input(A, 0, lda, C, B, H, W, V, lazy)
        if pre_gy is not None:
            assert tuple(pre_gy.shape) == (T2, 36, N)
            Gy = pre_gy
        else:
            Gy = torch.empty((T2, 36, N), dtype=torch.float32, device=dev)
            _C.check(_L().efgh_wino2d_dy(ptr(G), c_int64(ldg), c_int32(N), c_int32(B), c_int32(H), c_int32(W), ptr(Gy), _st()))
        g = _C.GemmDesc()
        g.A, g.lda, g.C, g.T, g.mode, g.N, g.M = V.data_ptr(), 36 * C, C, 1, 0, N, T2
        g.nbatch, g.batch_stride_a = 36, C
        if PROFILE_WINO2D_GEMM is not None:
            f0, f1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            f0.record()
        if PLANE_DMA and _L().efgh_plane_wgrad_supported(ctypes.byref(g), c_int64(36 * N)):
            _C.check(_L().efgh_plane_wgrad_batched(ctypes.byref(g), ptr(Gy), c_int64(36 * N), c_int64(N), ptr(S), c_int64(N * C),
                                                   ptr(_scratch(_L().efgh_plane_wgrad_workspace(ctypes.byref(g)), dev)),
                                                   c_int32(PLANE_DMA_NBUF), _st()))
        else:
            _C.check(_L().efgh_gather_wgrad_batched(ctypes.byref(g), ptr(Gy), c_int64(36 * N), c_int64(N), ptr(S), c_int64(N * C),
                                                    ptr(_scratch(_L().efgh_gather_wgrad_workspace(ctypes.byref(g)), dev)), _st()))
        if PROFILE_WINO2D_GEMM is not None:
            f1.record()
            PROFILE_WINO2D_GEMM.append((f0, f1, 2.0 * 36 * T2 * C * N, (0, T2, N, 36, C)))
        done = _C.check_wrote(_L().efgh_wino2d_wfinish(ptr(S), ptr(dWp), c_int32(N), c_int32(C), odp, _st()))      # (only the finish kernel may write the reference layout)
    elif USE_WINO_WGRAD and C % 64 == 0 and wino_eligible(mode, C, N, geom):
        wino = True
        S = _scratch(_L().efgh_wino_wgrad_workspace(ctypes.byref(d)), dWp.device)    # per-tile-range partials
        done = _C.check_wrote(_L().efgh_wino_wgrad(ctypes.byref(d), ptr(G), c_int64(ldg), ptr(S), ptr(dWp), odp, _st()))
    else:
        done = _C.check_wrote(_L().efgh_gather_wgrad(ctypes.byref(d), ptr(G), c_int64(ldg), ptr(dWp),
                                                     ptr(_scratch(_L().efgh_gather_wgrad_workspace(ctypes.byref(d)), dWp.device)), odp, _st()))
    if done:
        FOLD_UNPACK_HITS[0] += 1
    if PROFILE_WGRAD is not None and (thin or (not wino and hbm_bound(M, N, T, C))):
        e1.record()
        if PROFILE_THIN is not None:
            PROFILE_THIN.append((e0, e1, conv_bytes(M, N, T, C), (mode, M, N, T, C)))
        if PROFILE_DED is not None and (thin or sc):
            PROFILE_DED.append((e0, e1, conv_bytes(M, N, T, C), (mode, M, N, T, C)))
    elif PROFILE_WGRAD is not None:
        e1.record()
        if PROFILE_DED is not None and sc:
            PROFILE_DED.append((e0, e1, conv_bytes(M, N, T, C), (mode, M, N, T, C)))
        rec = (e0, e1, 2.0 * M * N * T * C, (mode, M, N, T, C))
        if wino == '2d' and PROFILE_WINO2D is not None:
            PROFILE_WINO2D.append(rec)
        else:
            (PROFILE_WINO_WGRAD if (wino and PROFILE_WINO_WGRAD is not None) else PROFILE_WGRAD).append(rec)
    return done


def unpack_weight(Wp, W, N, T, C, Cp, sn, sc, st, taps, accumulate=False):
    tp = (ctypes.c_int32 * 16)(*([int(t) for t in taps] + [0] * (16 - len(taps))))
    _C.check(_L().efgh_unpack_weight(ptr(Wp), ptr(W), c_int32(N), c_int32(T), c_int32(C), c_int32(Cp), c_int64(sn),
                                     c_int64(sc), c_int64(st), tp, c_int32(1 if accumulate else 0), _st()))


def table_scatter_add(src, table, M, T, C, dst):
    """dst[table[m][t]][c] += src[m][t*C+c] (atomic form; the BCL uses neighbor_gather_adjoint)"""
    _C.check(_L().efgh_table_scatter_add(ptr(src), ptr(table), c_int64(M), c_int32(T), c_int32(C), ptr(dst), _st()))


def bwd_groups(M):
    return _L().efgh_bwd_groups(c_int64(M))


def act_bn_bwd_reduce(dy, lddy, y, ldy, raw, ldraw, mean, invstd, M, C, act, slope, part, s1, s2, m1, m2,
                      pscale=None, pshift=None):
    _C.check(_L().efgh_act_bn_bwd_reduce(ptr(dy), c_int64(lddy), ptr(y), c_int64(ldy), ptr(raw), c_int64(ldraw),
                                         ptr(mean), ptr(invstd), ptr(pscale), ptr(pshift), c_int64(M), c_int32(C),
                                         c_int32(act),
                                         c_float(slope), ptr(part), ptr(s1), ptr(s2), ptr(m1), ptr(m2), _st()))


POOL_REDUCE_FROM_Y = True       # ReLU layers: the pooled BatchNorm-backward sums from the pooled gradient and the pooled activation only


def pool_bn_bwd(dy_pool, raw, mean, invstd, coef, pscale, pshift, act, slope, s1=None, s2=None, transforms=False, y_pool=None):
    """BatchNorm backward of a conv+BN+act layer fused with MaxPool2d(2,2), from the pooled gradient; raw [B][H][W][C].
    -> (draw [B][H][W][C], dbeta [C], dgamma [C]); transforms=True (a 2-D Winograd layer, round 6): draw is not stored - the apply pass
    runs inside the layer's two gradient-side transforms (efgh_wino2d_bwd_transforms_pooled) -> ((Vd, Gy), dbeta, dgamma)"""
    B, H, W, C = raw.shape
    dev = raw.device
    s1 = s1 if s1 is not None else torch.empty(C, dtype=torch.float32, device=dev)
    s2 = s2 if s2 is not None else torch.empty(C, dtype=torch.float32, device=dev)
    m1 = torch.empty(C, dtype=torch.float64, device=dev)
    m2 = torch.empty(C, dtype=torch.float64, device=dev)
    if (POOL_REDUCE_FROM_Y and y_pool is not None and act == ACT_RELU and y_pool.is_contiguous()
            and tuple(y_pool.shape) == (B, H // 2, W // 2, C)):
        # (y_pool: the layer's own output - the next layer's backward keeps it alive; the full-resolution raw map is not read)
        G = _L().efgh_bwd_groups(c_int64(B * (H // 2) * (W // 2)))
        part = torch.empty((G, 2, C), dtype=torch.float64, device=dev)
        _C.check(_L().efgh_pool_bn_bwd_reduce_pooled(ptr(dy_pool), ptr(y_pool), ptr(raw), ptr(mean), ptr(invstd), ptr(pscale), ptr(pshift),
                                                     c_int32(B), c_int32(H), c_int32(W), c_int32(C), ptr(part), ptr(s1), ptr(s2), ptr(m1),
                                                     ptr(m2), _st()))
    else:
        G = _L().efgh_pool_bwd_groups(c_int32(B), c_int32(H), c_int32(W))
        part = torch.empty((G, 2, C), dtype=torch.float64, device=dev)
        _C.check(_L().efgh_pool_bn_bwd_reduce(ptr(dy_pool), ptr(raw), ptr(mean), ptr(invstd), ptr(pscale), ptr(pshift), c_int32(B),
                                              c_int32(H), c_int32(W), c_int32(C), c_int32(act), c_float(slope), ptr(part), ptr(s1),
                                              ptr(s2), ptr(m1), ptr(m2), _st()))
    if transforms:
        T2 = _L().efgh_wino2d_tiles(c_int32(B), c_int32(H), c_int32(W))
        Vd = torch.empty((T2, 36, C), dtype=torch.float32, device=dev)
        Gy = torch.empty((T2, 36, C), dtype=torch.float32, device=dev)
        LAZY_HITS[1] += 1
        _C.check(_L().efgh_wino2d_bwd_transforms_pooled(ptr(dy_pool), c_int64(C), ptr(raw), c_int64(C), ptr(pscale), ptr(pshift), ptr(mean),
                                                        ptr(invstd), ptr(coef), ptr(m1), ptr(m2), c_int32(C), c_int32(B), c_int32(H),
                                                        c_int32(W), c_int32(act), c_float(slope), ptr(Vd), ptr(Gy), _st()))
        return (Vd, Gy), s1, s2
    draw = torch.empty_like(raw)
    _C.check(_L().efgh_pool_bn_bwd_apply(ptr(dy_pool), ptr(raw), ptr(mean), ptr(invstd), ptr(coef), ptr(m1), ptr(m2), ptr(pscale),
                                         ptr(pshift), c_int32(B), c_int32(H), c_int32(W), c_int32(C), c_int32(act),
                                         c_float(slope), ptr(draw), _st()))
    return draw, s1, s2


def act_bn_bwd_apply(dy, lddy, y, ldy, raw, ldraw, mean, invstd, coef, m1, m2, M, C, act, slope, draw, lddraw,
                     dres, lddres, pscale=None, pshift=None):
    _C.check(_L().efgh_act_bn_bwd_apply(ptr(dy), c_int64(lddy), ptr(y), c_int64(ldy), ptr(raw), c_int64(ldraw),
                                        ptr(mean), ptr(invstd), ptr(coef), ptr(m1), ptr(m2), ptr(pscale), ptr(pshift),
                                        c_int64(M), c_int32(C),
                                        c_int32(act), c_float(slope), ptr(draw), c_int64(lddraw), ptr(dres),
                                        c_int64(lddres), _st()))


def col_sum(x, M, C):
    stats, G = col_stats(x, M, C, x.shape[-1])
    return stats[:, 0, :].sum(0)


def maxpool2_bwd(x, dy, dx):
    B, H, W, C = x.shape
    _C.check(_L().efgh_maxpool2_bwd(ptr(x), ptr(dy), ptr(dx), c_int32(B), c_int32(H), c_int32(W), c_int32(C), _st()))


def segment_colmax_bwd(dy, arg, nseg, C, dx, ld):
    _C.check(_L().efgh_segment_colmax_bwd(ptr(dy), ptr(arg), c_int32(nseg), c_int32(C), ptr(dx), c_int64(ld), _st()))


def segment_colmean_bwd(dy, P, nseg, C, dx, ld):
    _C.check(_L().efgh_segment_colmean_bwd(ptr(dy), c_int32(P), c_int32(nseg), c_int32(C), ptr(dx), c_int64(ld),
                                           _st()))


def softmax2_bwd(y, dy, dx):
    B, _, H, W = y.shape
    _C.check(_L().efgh_softmax2_bwd(ptr(y), ptr(dy), c_int32(B), c_int64(H * W), ptr(dx), c_int64(dx.shape[-1]),
                                    _st()))


def raster_bwd(pix, gimg, B, N, HW):
    gv = torch.empty((B, N, 4), dtype=torch.float32, device=gimg.device)
    _C.check(_L().efgh_raster_bwd(ptr(pix), ptr(gimg), c_int32(B), c_int32(N), c_int64(HW), ptr(gv), _st()))
    return gv


def raster_pose_bwd(pix, gimg, pc, e_l, B, N, HW, mode):
    """gradient w.r.t. the pose argument of the range (mode 0, -> (B,16)) / depth (mode 1, -> (B,12)) rasteriser"""
    _C.require_cuda(pix, gimg, pc)
    pc = pc.contiguous()
    part = torch.empty((B, 64, 16), dtype=torch.float64, device=gimg.device)
    out = torch.empty((B, 16 if mode == 0 else 12), dtype=torch.float32, device=gimg.device)
    _C.check(_L().efgh_raster_pose_bwd(ptr(pix), ptr(gimg), ptr(pc), ptr(None if e_l is None else e_l.contiguous()), c_int32(B),
                                       c_int32(N), c_int64(HW), c_int32(mode), ptr(part), ptr(out), _st()))
    return out


def corr1d_bwd_mfma(rp, cam, cam_mm, dl, B, h, wc, wp, rp_pitch=None):
    """the two correlation gradients as Toeplitz GEMMs on the MFMA kernel (see efgh_corr_planes); rp [B][h][rp_pitch][16]"""
    dev = cam.device
    nj = wp - wc + 1
    rp_pitch = rp_pitch or wp
    wpP, wcP = ceil4(wp), ceil4(wc)
    f = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
    rpT, camT = f(B, h * 16, wpP), f(B, h * 16, wcP)
    _C.check(_L().efgh_corr_planes(ptr(rp), None, c_int32(B), c_int32(h), c_int32(wp), c_int32(rp_pitch), c_int32(wpP),
                                   ptr(rpT), _st()))
    _C.check(_L().efgh_corr_planes(ptr(cam), ptr(cam_mm), c_int32(B), c_int32(h), c_int32(wc), c_int32(wc), c_int32(wcP),
                                   ptr(camT), _st()))
    T, TT = f(B, wc, wpP), f(B, wp, wcP)
    _C.check(_L().efgh_corr_toeplitz(ptr(dl), c_int32(B), c_int32(nj), c_int32(wc), c_int32(wp), c_int32(wpP), c_int32(0),
                                     ptr(T), _st()))
    _C.check(_L().efgh_corr_toeplitz(ptr(dl), c_int32(B), c_int32(nj), c_int32(wp), c_int32(wc), c_int32(wcP), c_int32(1),
                                     ptr(TT), _st()))
    dcamT, drpT = f(B, h * 16, wc), f(B, h * 16, wp)
    M = h * 16
    gather_gemm(rpT, wpP, wpP, 1, T, wc, M, dcamT, wc, mode=0, flops=2.0 * B * h * wc * 16 * nj,
                batch=(B, M * wpP, wc * wpP, M * wc))
    gather_gemm(camT, wcP, wcP, 1, TT, wp, M, drpT, wp, mode=0, flops=2.0 * B * h * wc * 16 * nj,
                batch=(B, M * wcP, wp * wcP, M * wp))
    dcam, drp = f(B, h, wc, 16), f(B, h, wp, 16)
    _C.check(_L().efgh_corr_unplanes(ptr(dcamT), c_int32(B), c_int32(h), c_int32(wc), c_int32(wc), ptr(dcam), _st()))
    _C.check(_L().efgh_corr_unplanes(ptr(drpT), c_int32(B), c_int32(h), c_int32(wp), c_int32(wp), ptr(drp), _st()))
    return dcam, drp


def corr1d_bwd(rp, cam, cam_mm, dl, B, h, wc, wp):
    """rp [B][h][pitch >= wp][16] (the MFMA forward pads the rows); returns (dcam_n [B][h][wc][16], drp [B][h][wp][16])"""
    pitch = rp.shape[2]
    if (USE_MFMA_CORR and h * 16 >= 128) or pitch != wp:
        return corr1d_bwd_mfma(rp, cam, cam_mm, dl, B, h, wc, wp, rp_pitch=pitch)
    dcam = torch.empty_like(cam)
    drp = torch.empty_like(rp)
    _C.check(_L().efgh_corr1d_bwd(ptr(rp), ptr(cam), ptr(cam_mm), ptr(dl), c_int32(B), c_int32(h), c_int32(wc),
                                  c_int32(wp), ptr(dcam), ptr(drp), _st()))
    return dcam, drp


def norm_bwd(x, dxn, mm):
    """gradient of x / (max(x) - min(x)) per sample given d/d(x_n); x, dxn [B][...] contiguous"""
    B = x.shape[0]
    n = x.numel() // B
    G = _L().efgh_minmax_groups(c_int64(n))
    part = torch.empty((B, G, 3), dtype=torch.float32, device=x.device)
    dx = torch.empty_like(x)
    _C.check(_L().efgh_norm_bwd(ptr(x), ptr(dxn), ptr(mm), c_int32(B), c_int64(n), ptr(part), ptr(dx), _st()))
    return dx


def corr_unpad(drp, B, h, w, C, off):
    dx = torch.empty((B, h, w, C), dtype=torch.float32, device=drp.device)
    _C.check(_L().efgh_corr_unpad(ptr(drp), c_int32(B), c_int32(h), c_int32(w), c_int32(C), c_int32(off), ptr(dx),
                                  _st()))
    return dx


def convt_col2im(Y, B, Hin, Win, Ho, Wo, O, pad, scale, shift, act, slope, out):
    _C.check(_L().efgh_convt_col2im(ptr(Y), c_int64(Y.shape[-1]), c_int32(B), c_int32(Hin), c_int32(Win), c_int32(Ho),
                                    c_int32(Wo), c_int32(O), c_int32(pad), ptr(scale), ptr(shift), c_int32(act),
                                    c_float(slope), ptr(out), c_int64(out.shape[-1]), _st()))


def convt_im2col(G, B, Hin, Win, Ho, Wo, O, pad, Ycol):
    _C.check(_L().efgh_convt_im2col(ptr(G), c_int64(G.shape[-1]), c_int32(B), c_int32(Hin), c_int32(Win), c_int32(Ho),
                                    c_int32(Wo), c_int32(O), c_int32(pad), ptr(Ycol), c_int64(Ycol.shape[-1]), _st()))


# ----------------------------------------------------------------------------------------------
# G image losses (losses/loss_utils.py:186-199)
# ----------------------------------------------------------------------------------------------
def gimg_loss_fwd(pred_depth, pred_mask, gdep4, img_mask):
    """-> (out3 = [l_depth, l_mask, sum(valid)] device tensor, gt_depth (B,1,H,W), gt_mask (B,1,H,W))"""
    B, _, H, W = pred_depth.shape
    dev = pred_depth.device
    gt_depth = torch.empty((B, 1, H, W), dtype=torch.float32, device=dev)
    gt_mask = torch.empty((B, 1, H, W), dtype=torch.float32, device=dev)
    G = _L().efgh_gimg_loss_groups(c_int64(B * H * W))
    part = torch.empty((G, 3), dtype=torch.float64, device=dev)
    out3 = torch.empty(3, dtype=torch.float32, device=dev)
    _C.check(_L().efgh_gimg_loss_fwd(ptr(pred_depth), ptr(pred_mask), c_int64(pred_mask.stride(0)), ptr(gdep4), ptr(img_mask),
                                     c_int32(B), c_int64(H * W), ptr(gt_depth), ptr(gt_mask), ptr(part), ptr(out3), _st()))
    return out3, gt_depth, gt_mask


def gimg_loss_bwd(pred_depth, pred_mask, gt_depth, img_mask, out3, g_depth, g_mask):
    B, _, H, W = pred_depth.shape
    d_depth = torch.empty_like(pred_depth)
    d_mask = torch.zeros_like(pred_mask)                   # channel 1 receives no gradient from this loss
    _C.check(_L().efgh_gimg_loss_bwd(ptr(pred_depth), ptr(pred_mask), c_int64(pred_mask.stride(0)), ptr(gt_depth), ptr(img_mask),
                                     c_int32(B), c_int64(H * W), ptr(out3), ptr(g_depth), ptr(g_mask), ptr(d_depth), ptr(d_mask),
                                     c_int64(d_mask.stride(0)), _st()))
    return d_depth, d_mask
