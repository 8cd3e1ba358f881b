"""Tensor-level wrappers over the C ABI (one function per fsraft_* entry point).

These allocate outputs with torch (device memory + current stream are the only things
torch provides here) and call straight into libfsraft.so.  No autograd in this file;
see core/corr.py, core/update.py and core/raft.py for the autograd.Function wrappers.
"""
import ctypes
import math
import os
import threading

import torch
import torch.optim.optimizer as _optimizer_module

from . import _lib as L


def _lib():
    return L.load()


def parameters_updated(params):
    """Tell the weight-pack caches (keyed on `Parameter._version`) that `params` changed by a route that does not bump
    the version counter.  `torch.optim.*(fused=True)` is such a route: `_fused_adamw_` updates the parameters in place and
    leaves `_version` where it was (checked on torch 2.10), so a GEMM-ready pack made before the step would be reused
    after it.  The post-step hook below calls this for every fused optimizer; hipGraph replays need it by hand."""
    ps = [p for p in params if isinstance(p, torch.Tensor)]
    if ps:
        torch._C._autograd._unsafe_set_version_counter(ps, [p._version + 1 for p in ps])


def _after_optimizer_step(opt, args, kwargs):
    parameters_updated([p for g in opt.param_groups if g.get("fused") for p in g["params"]])


_optimizer_module.register_optimizer_step_post_hook(_after_optimizer_step)


class KernelTimer:
    """Optional per-launch timing with events recorded on the launching stream (torch's current
    stream is the stream every fsraft kernel is enqueued on).  bench.py installs one for the
    timed region; when `ops.TIMER is None` (the default) nothing is recorded."""

    def __init__(self, detail=False):
        self.rec = {}          # family -> list of (start_event, end_event, flops, bytes)
        self.detail = detail   # also keep one entry per (family, shape tag): scripts/layer_times.py

    def begin(self):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def end(self, family, e0, flops=0.0, nbytes=0.0, flops_done=None, tag=None):
        """flops: the ALGORITHMIC count of the launch; flops_done (optional, a callable evaluated in summary(), i.e. after the
        device sync): what the launch really multiplied where that is less -- the volume-backward GEMMs visit only the k-tiles
        the step's lookups reached."""
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        self.rec.setdefault(family, []).append((e0, e1, flops, nbytes, flops_done))
        if self.detail and tag:
            self.rec.setdefault(family + "|" + tag, []).append((e0, e1, flops, nbytes, flops_done))

    def summary(self):
        """{family: dict(launches, ms_total, ms_avg, flops, bytes, flops_done)} -- call after a device sync."""
        out = {}
        for fam, rows in self.rec.items():
            if "|" in fam and not self.detail:
                continue
            ms = sum(r[0].elapsed_time(r[1]) for r in rows)
            out[fam] = dict(launches=len(rows), ms_total=ms, ms_avg=ms / len(rows),
                            flops=sum(r[2] for r in rows), bytes=sum(r[3] for r in rows),
                            flops_done=sum((r[4]() if r[4] is not None else r[2]) for r in rows))
        return out


TIMER = None


class _ZeroPool:
    """Small zero-initialised buffers (atomic-accumulation targets, padded channels) carved out of 32 MB chunks that are
    zero-filled ONCE: a training step used to issue ~285 separate fill kernels of a few KB each.  A chunk lives as long
    as any buffer carved from it; the pool itself only holds the chunk it is currently carving.

    Under hipGraph capture the same buffers are used again by every replay, so "once" has to mean once per replay: a
    capture never carves from a chunk that was filled outside it (or inside another capture) -- chunks are keyed on the
    capture's id, so its first request opens a new chunk, whose fill is recorded in the graph ahead of every use of the
    buffers carved from it."""
    CHUNK = 32 << 20
    SMALL = 64 << 10            # 1-element anchors and other <= 256-byte requests: their own chunk, so that an anchor that
                                # lives for a whole step does not pin 32 MB

    def __init__(self):
        self.cur = {}
        self.epoch = 0                  # bumped by new_epoch(); chunks belong to one (epoch, capture id)

    @staticmethod
    def _capture_id(device):
        """0 outside a capture, else the id of the hipGraph capture the current stream is in (fsraft_stream_capture_id):
        two captures back to back differ in it even when no eager take() ran in between."""
        if not torch.cuda.is_current_stream_capturing():
            return 0
        cid = ctypes.c_ulonglong(0)
        L.check(_lib().fsraft_stream_capture_id(L.stream(), ctypes.byref(cid)), "stream_capture_id")
        return int(cid.value) or 1

    def take(self, shape, device):
        n = 1
        for v in shape:
            n *= int(v)
        nb = (n * 4 + 255) // 256 * 256
        if nb > self.CHUNK // 4 or n == 0 or torch.device(device).type != "cuda":
            return torch.zeros(shape, device=device, dtype=torch.float32)
        small = nb <= 256
        size = self.SMALL if small else self.CHUNK
        tag = (self.epoch, self._capture_id(device))
        # (per host thread as well: the carve below is a read-modify-write of the entry -- ADVICE r5)
        key = (torch.device(device).index, torch.cuda.current_stream(device).cuda_stream, small, threading.get_ident())
        ent = self.cur.get(key)
        if ent is None or ent[1] + nb > size or ent[2] != tag:
            ent = [torch.zeros(size // 4, device=device, dtype=torch.float32), 0, tag]
            self.cur[key] = ent
        o = ent[1] // 4
        ent[1] += nb
        return ent[0][o:o + n].view(shape)

    def new_epoch(self):
        self.epoch += 1


_ZEROS = _ZeroPool()


def new_zero_epoch():
    """Force the next zero-pool request to open a fresh chunk (captures are told apart by their id; this is for callers
    that want buffers of one phase not to share a chunk with the next)."""
    _ZEROS.new_epoch()


def zeros(*shape, device):
    """torch.zeros(shape, fp32) from the zero pool (see _ZeroPool)."""
    if len(shape) == 1 and isinstance(shape[0], (tuple, list, torch.Size)):
        shape = tuple(shape[0])
    return _ZEROS.take(shape, device)


# ------------------------------------------------------------------ amax words (split arithmetic, csrc/split_arith.hpp)
# Every tensor the matrix pipe reads is split into fp16 pieces of x * scale, the scale a power of two derived ON THE DEVICE
# from the tensor's "amax word": one 4-byte word holding the bit pattern of max |x| (or an upper bound).  A word is a
# 1-element tensor of the zero pool.  Producers inside the library raise the word of their destination (Dst.amax,
# fsraft_conv_desc.dst_amax, ...); for everything else `ensure_amax` runs fsraft_amax_jobs over the tensor right before its
# consumer -- always correct, one more small launch.  A buffer carries its word as the attribute `_fs_amax` ONLY when every
# kernel that writes it raises the word (views find their base's word): `tracked()` is for code that knows that.
def new_amax(device):
    return zeros(1, device=device)


def tracked(t, word=None):
    """Mark buffer t as carrying an amax word that all of its writers raise; returns t.  (The word is noted on t and on the
    tensor t is a view of: views made from t later find it through their `_base`, which is that root.)"""
    w = new_amax(t.device) if word is None else word
    t._fs_amax = w
    if t._base is not None and t._base.numel() == t.numel():      # (a re-shaped whole -- never a slice of a shared pool chunk)
        t._base._fs_amax = w
    return t


def amax_of(t):
    """The amax word a tensor (or the buffer it is a view of) carries, or None."""
    w = getattr(t, "_fs_amax", None)
    if w is None and t._base is not None:
        w = getattr(t._base, "_fs_amax", None)
    return w


_AMAX_ONE = {}


def amax_one(device):
    """A word holding the bit pattern of 1.0: for tensors bounded by 1 (gates, softmax rows)."""
    k = torch.device(device).index or 0
    w = _AMAX_ONE.get(k)
    if w is None:
        w = _AMAX_ONE[k] = torch.ones(1, device=device, dtype=torch.float32)
    return w


_AMAX_TRACE = None


def amax_jobs(jobs):
    """jobs: (data_ptr, rows, C, ld, word tensor); word = max(word, max |x|) for each, 32 jobs per launch."""
    n = len(jobs)
    if not n:
        return
    if os.environ.get("FSRAFT_AMAX_TRACE"):           # debugging aid: who still needs a pass of its own (printed at exit)
        global _AMAX_TRACE
        import atexit
        import collections
        import traceback
        if _AMAX_TRACE is None:
            _AMAX_TRACE = collections.Counter()
            atexit.register(lambda: [print(f"amax pass x{c}: {k}", flush=True) for k, c in _AMAX_TRACE.most_common(40)])
        fr = traceback.extract_stack(limit=6)[:-1]
        _AMAX_TRACE[" < ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in reversed(fr))] += 1
    ptrs = (ctypes.c_void_p * n)(*[j[0] for j in jobs])
    rows = (ctypes.c_int64 * n)(*[j[1] for j in jobs])
    cs = (ctypes.c_int64 * n)(*[j[2] for j in jobs])
    lds = (ctypes.c_int64 * n)(*[j[3] for j in jobs])
    words = (ctypes.c_void_p * n)(*[j[4].data_ptr() for j in jobs])
    L.check(_lib().fsraft_amax_jobs(ctypes.cast(ptrs, L._PP), rows, cs, lds, ctypes.cast(words, L._PP), n, L.stream()), "amax_jobs")


def amax_tensor(t):
    """Fresh word holding max |t| of a contiguous fp32 tensor."""
    w = new_amax(t.device)
    amax_jobs([(t.data_ptr(), 1, t.numel(), t.numel(), w)])
    return w


def amax_scaled(word, factor):
    """Fresh word = factor * word (a bound derived from another tensor's: fsraft_amax_scaled)."""
    out = new_amax(word.device)
    L.check(_lib().fsraft_amax_scaled(L.ptr(word), float(factor), L.ptr(out), L.stream()), "amax_scaled")
    return out


def amax_of_tensors(ts):
    """One word bounding max |x| over the contiguous fp32 tensors ts: the word they share if they all are views of one tracked
    buffer, else a fresh word computed now (one launch)."""
    ws = [amax_of(t) for t in ts]
    if ws[0] is not None and all(w is ws[0] for w in ws):
        return ws[0]
    w = new_amax(ts[0].device)
    amax_jobs([(t.data_ptr(), 1, t.numel(), t.numel(), w) for t in ts])
    return w


def _wptr(t):
    """Device address of the amax word a record tensor was split with (None in exact-fp32 mode / no word: scale 1)."""
    w = amax_of(t) if t is not None else None
    return L.ptr(w)


def ensure_amax(views):
    """Give every V of `views` an amax word: the one its buffer carries, else a fresh one computed now (one batched launch
    for all that need it).  Returns the words' device addresses (0 in exact-fp32 mode: the kernels do not scale)."""
    if exact_mode():
        return [None] * len(views)
    jobs = []
    if AMAX_AUDIT and not torch.cuda.is_current_stream_capturing():
        # every word a buffer CARRIES is checked against the truth (synchronises): a writer that forgot to raise it shows up
        # here, not as an overflow three layers later.  tests/ run the whole GPU suite once in this mode.
        for v in views:
            if v.amax is not None:
                true = float(v.t[..., v.off:v.off + v.C].abs().max())
                word = float(v.amax.item())
                if not (true == 0.0 or 2.0 * word > true) or word != word:
                    raise RuntimeError(f"amax audit: word {word:g} does not bound max |x| = {true:g} (shape {tuple(v.t.shape)}, "
                                       f"channels [{v.off}, {v.off + v.C}))")
    for v in views:
        if v.amax is None:
            v.amax = new_amax(v.t.device)
            rows = v.t.numel() // v.ld
            jobs.append((v.ptr, rows, v.C, v.ld, v.amax))
    amax_jobs(jobs)
    return [v.amax.data_ptr() for v in views]


AMAX_AUDIT = bool(os.environ.get("FSRAFT_AMAX_AUDIT"))     # (module attribute: tests set it)
_NO_WORDS = int(os.environ.get("FSRAFT_NO_WORDS", "0"))   # measurement only: bit 0 = convolutions read no words (scale 1), bit 1 = raise none


def pyramid_sizes(H, W, num_levels=4):
    out = []
    for _ in range(num_levels):
        out.append((H, W))
        H, W = H // 2, W // 2
    return out


def _planar2_strides(t):
    """(bs, cs, ps) for a [B,2,H,W] tensor whose H,W dims are jointly contiguous-strided
    (covers NCHW-contiguous and channels-last)."""
    B, C, H, W = t.shape
    sb, sc, sh, sw = t.stride()
    if sh != W * sw:
        raise RuntimeError("2-channel tensor must have a single pixel stride (got strides %s)" % (t.stride(),))
    return sb, sc, sw


# ------------------------------------------------------------------ correlation volume
def corr_build(fmap1, fmap2, num_levels=4):
    L.require_cuda_f32(fmap1, fmap2)
    fmap1 = fmap1.contiguous()
    fmap2 = fmap2.contiguous()
    B, C, H, W = fmap1.shape
    sizes = pyramid_sizes(H, W, num_levels)
    if sizes[-1][0] < 1 or sizes[-1][1] < 1:
        raise RuntimeError(f"feature map {H}x{W} too small for {num_levels} pyramid levels")
    levels = [torch.empty(B * H * W, 1, h, w, device=fmap1.device, dtype=torch.float32) for h, w in sizes]
    pp, keep = L.ptr_array(levels)
    t = TIMER
    e0 = t.begin() if t else None
    a1, a2 = (None, None) if exact_mode() else (amax_tensor(fmap1), amax_tensor(fmap2))
    L.check(_lib().fsraft_corr_build(L.ptr(fmap1), L.ptr(fmap2), pp, num_levels, B, C, H, W, L.ptr(a1), L.ptr(a2), L.stream()), "corr_build")
    if t:
        N = H * W
        t.end("corr_build", e0, 2.0 * B * N * N * C, 4.0 * B * (2 * N * C + N * sum(h * w for h, w in sizes)))
    return levels


class VolLayout:
    """Tiled-row layout of the volume pyramid (csrc/corr_layout.hpp, fsraft_vol_layout)."""
    __slots__ = ("nlev", "H", "W", "P", "h", "w", "th", "tw", "off")
    _cache = {}

    def __init__(self, H, W, num_levels=4):
        buf = (ctypes.c_int * 24)()
        L.check(_lib().fsraft_vol_layout(H, W, num_levels, buf), "vol_layout")
        v = list(buf)
        self.nlev, self.H, self.W, self.P = v[:4]
        self.h, self.w, self.th, self.tw, self.off = v[4:8], v[8:12], v[12:16], v[16:20], v[20:24]

    @classmethod
    def get(cls, H, W, num_levels=4):
        k = (H, W, num_levels)
        if k not in cls._cache:
            cls._cache[k] = cls(H, W, num_levels)
        return cls._cache[k]

    def level_view(self, vol, l):
        """Level l of a tiled volume [rows, P] as the reference's row-major [rows, 1, h_l, w_l] tensor (a copy: API edge)."""
        rows = vol.shape[0]
        n = self.th[l] * self.tw[l] * 16
        t = vol[:, self.off[l]:self.off[l] + n].reshape(rows, self.th[l], self.tw[l], 4, 4).permute(0, 1, 3, 2, 4)
        return t.reshape(rows, 1, self.th[l] * 4, self.tw[l] * 4)[:, :, :self.h[l], :self.w[l]].contiguous()


BUILD_REC = True       # volume build on the record GEMM core when the arithmetic mode is split (fp16x3) and C % 32 == 0


def fmap_records(fmap):
    """[B,C,H,W] fp32 -> [B,H*W,C] records (pixel-major, the operand format of the record GEMM core).  A channels_last
    feature map (what the channels-last encoders hand over) already is pixel-major: no transpose."""
    B, C, H, W = fmap.shape
    if fmap.is_contiguous(memory_format=torch.channels_last) and not fmap.is_contiguous():
        return to_records(fmap.permute(0, 2, 3, 1).reshape(B, H * W, C))
    return to_records(nchw_to_nhwc(fmap).view(B, H * W, C))


def corr_build_tiled(fmap1, fmap2, num_levels=4, recs=None):
    """-> (vol [B*H*W, P], VolLayout): all-pairs volume + pyramid in the tiled-row layout.  recs: optional (f1r, f2r) from
    fmap_records (reused by the backward pass)."""
    L.require_cuda_f32(fmap1, fmap2)
    fmap1 = fmap1.contiguous()
    fmap2 = fmap2.contiguous()
    B, C, H, W = fmap1.shape
    lay = VolLayout.get(H, W, num_levels)
    vol = torch.empty(B * H * W, lay.P, device=fmap1.device, dtype=torch.float32)
    t = TIMER
    e0 = t.begin() if t else None
    if recs is not None:
        L.check(_lib().fsraft_corr_build_rec(L.ptr(recs[0]), L.ptr(recs[1]), L.ptr(vol), num_levels, B, C, H, W, _wptr(recs[0]),
                                             _wptr(recs[1]), L.stream()), "corr_build_rec")
    else:
        a1, a2 = (None, None) if exact_mode() else (amax_tensor(fmap1), amax_tensor(fmap2))
        L.check(_lib().fsraft_corr_build_tiled(L.ptr(fmap1), L.ptr(fmap2), L.ptr(vol), num_levels, B, C, H, W, L.ptr(a1), L.ptr(a2),
                                               L.stream()), "corr_build_tiled")
    if t:
        N = H * W
        t.end("corr_build", e0, 2.0 * B * N * N * C, 4.0 * B * (2 * N * C + N * sum(h * w for h, w in zip(lay.h, lay.w))))
    return vol, lay


def corr_lookup_tiled_fwd(vol, lay, coords, radius, is_flow=False, out=None):
    """-> [B,H,W,L*(2r+1)^2] channels-last (into `out` if given).  is_flow: `coords` holds the flow, the query position is pixel
    grid + flow."""
    L.require_cuda_f32(vol, coords)
    B, _, H, W = coords.shape
    bs, cs, ps = _planar2_strides(coords)
    ch = lay.nlev * (2 * radius + 1) ** 2
    if out is None:
        out = tracked(torch.empty(B, H, W, ch, device=coords.device, dtype=torch.float32))     # (the lookup raises the word)
    elif tuple(out.shape) != (B, H, W, ch) or not out.is_contiguous() or out.dtype != torch.float32:
        raise ValueError(f"lookup out= must be a contiguous fp32 [{B},{H},{W},{ch}] tensor")
    t = TIMER
    e0 = t.begin() if t else None
    L.check(_lib().fsraft_corr_lookup_tiled_fwd(L.ptr(vol), lay.nlev, L.ptr(coords), bs, cs, ps, L.ptr(out), B, H, W, radius,
                                                int(is_flow), _wptr(out), L.stream()), "corr_lookup_tiled_fwd")
    if t:
        t.end("corr_lookup_fwd", e0, 0.0, 4.0 * B * H * W * (lay.nlev * (2 * radius + 2) ** 2 + 2 + ch))
    return out


def corr_dvol_build(douts, coords, lay, B, radius, records=False, is_flow=False, q0=0, nq=0, wmask=None, out=None):
    """Gradient volume [B*H*W, P] of all stashed lookups (douts[t]: [B,H,W,CH] channels-last, coords[t]: [B,2,H,W]);
    records=True: rows of [32 hi | 32 lo] fp16 records (the operand format of gemm_rec_nt / gemm_rec_tn).
    q0 / nq: build only queries [q0, q0 + nq) (into rows 0 .. nq-1).  wmask (KTileLists.wmask): write only the records the two
    list GEMMs of corr_build_bwd_tiled(ktiles=...) read -- the rest of the returned tensor is then UNINITIALISED.
    out: optional preallocated [rows, P] tensor."""
    H, W = lay.H, lay.W
    rows = nq if nq else B * H * W - q0
    dvol = torch.empty(rows, lay.P, device=douts[0].device, dtype=torch.float32) if out is None else out
    if wmask is not None and not (records and len(douts) <= 16 and DVOL_BOX):
        wmask = None                      # (only the bounding-box kernel honours the mask; the row kernel writes whole rows)
    # scratch for the work list of queries whose lookups spread beyond the bounding-box kernel's box (1 + rows unsigned)
    qlist = torch.empty(rows + 1, device=douts[0].device, dtype=torch.int32) if len(douts) <= 16 else None
    word = None
    if records:
        # the rows leave as fp16 pieces of dV * scale: a cell collects at most one unit of bilinear weight per lookup, so
        # |dV| <= (number of lookups) x max |dout| -- the word the two GEMMs reading the rows are given as well
        word = amax_scaled(amax_of_tensors(douts), len(douts))
        dvol._fs_amax = word
    t = TIMER
    e0 = t.begin() if t else None
    for g0 in range(0, len(douts), 16):
        ds, cs_ = douts[g0:g0 + 16], coords[g0:g0 + 16]
        L.require_cuda_f32(*ds, *cs_)
        n = len(ds)
        a_d = (ctypes.c_void_p * n)(*[d.data_ptr() for d in ds])
        a_c = (ctypes.c_void_p * n)(*[c.data_ptr() for c in cs_])
        st = []
        for c in cs_:
            st += list(_planar2_strides(c))
        a_s = (ctypes.c_int64 * (3 * n))(*st)
        L.check(_lib().fsraft_corr_dvol_build(ctypes.cast(a_d, L._PP), ctypes.cast(a_c, L._PP), a_s, n, L.ptr(dvol), lay.nlev, B, H,
                                              W, radius, int(g0 > 0), int(records), int(is_flow), q0, rows, L.ptr(qlist), L.ptr(wmask),
                                              L.ptr(word), L.stream()),
                "corr_dvol_build")
    if t:   # per lookup read dOut + read-modify-write the window taps; plus the zero fill of the dense gradient
        nl = lay.nlev
        t.end("corr_lookup_bwd", e0, 0.0, 4.0 * rows * (len(douts) * (nl * (2 * radius + 1) ** 2 + 2 + 2 * nl * (2 * radius + 2) ** 2)
                                                         + sum(h * w for h, w in zip(lay.h, lay.w))))
    return dvol


# Route constants.  Each was an environment switch while its A/B was open (profiles/README.md holds the same-box numbers); they are
# plain module attributes now -- tests and scripts/ that compare routes set them directly.
BWD_KSKIP = True       # volume-backward GEMMs over the k-tiles the lookups reached only
CHUNK_KSKIP = False    # ... also in the chunked backward of AlternateCorrBlock (measured slower)
NT_LIST_KSPLIT = 1     # k-slices of the listed dF1 GEMM
TN_LIST_KSPLIT = 2     # k-slices of the listed d2cat GEMM (one slice with plain stores measured slower: 270 vs 210 us)
DVOL_WMASK = True      # ... and the gradient volume written only where they read
DVOL_BOX = True        # gradient volume on the wave-per-query kernel (scripts that call fsraft_set_dvol_box(0) clear this too)


class KTileLists:
    """k-tile lists of the two volume-backward GEMMs (fsraft_corr_bwd_ktiles) for one step's lookups."""
    __slots__ = ("nt_list", "nt_count", "nt_stride", "tn_list", "tn_count", "tn_stride", "wmask")


def corr_bwd_ktiles(coords, lay, B, radius, is_flow=False, q0=0, nq=0):
    """coords: the step's lookup coordinates ([B,2,H,W] each, at most 16) -> KTileLists, or None where the lists do not apply
    (more than 16 lookups, rows beyond the kernels' bitmaps): the caller then contracts densely.  nq > 0: the lists of the chunk
    of queries [q0, q0 + nq) of one image (one list set, as for B = 1 and H*W = nq)."""
    n = len(coords)
    Bfull = B
    H, W, P = lay.H, lay.W, lay.P
    if n < 1 or n > 16 or P // 32 > 2048 or -(-H * W // 32) > 2048:        # (the GEMMs hold a list of <= 2048 k-tiles in LDS)
        return None
    L.require_cuda_f32(*coords)
    dev = coords[0].device
    HW = H * W
    if nq:
        if q0 // HW != (q0 + nq - 1) // HW:
            return None
        B, HW = 1, nq
    k = KTileLists()
    ntiles, mtiles, ktq = -(-HW // 128), -(-P // 256), -(-HW // 32)
    k.nt_stride, k.tn_stride = P // 32, ktq
    k.nt_list = torch.empty(B * ntiles * k.nt_stride, device=dev, dtype=torch.int32)
    k.nt_count = torch.empty(B * ntiles, device=dev, dtype=torch.int32)
    k.tn_list = torch.empty(B * mtiles * k.tn_stride, device=dev, dtype=torch.int32)
    k.tn_count = torch.empty(B * mtiles, device=dev, dtype=torch.int32)
    bits = torch.empty(B * ktq * (-(-mtiles // 32)), device=dev, dtype=torch.int32)
    k.wmask = torch.empty(B * ktq * (-(-(P // 32) // 32)), device=dev, dtype=torch.int32)     # records per 32-query block the GEMMs read
    a_c = (ctypes.c_void_p * n)(*[c.data_ptr() for c in coords])
    st = []
    for c in coords:
        st += list(_planar2_strides(c))
    a_s = (ctypes.c_int64 * (3 * n))(*st)
    tm = TIMER
    e0 = tm.begin() if tm else None
    L.check(_lib().fsraft_corr_bwd_ktiles(ctypes.cast(a_c, L._PP), a_s, n, lay.nlev, Bfull, H, W, radius, int(is_flow), q0, nq, L.ptr(k.nt_list),
                                          L.ptr(k.nt_count), k.nt_stride, L.ptr(bits), L.ptr(k.tn_list), L.ptr(k.tn_count), k.tn_stride,
                                          L.ptr(k.wmask), L.stream()), "corr_bwd_ktiles")
    if tm:      # (its time belongs to the volume backward it shortens; no algorithmic bytes of its own)
        tm.end("corr_build_bwd", e0, 0.0, 0.0)
    if os.environ.get("FSRAFT_KTILE_STATS"):      # debugging aid (synchronises): how much of the contraction the lists keep
        print(f"k-tiles kept: NT {k.nt_count.sum().item() / (B * ntiles * (P // 32)):.3f}  TN {k.tn_count.sum().item() / (B * mtiles * ktq):.3f}  "
              f"records written {sum(bin(v & 0xffffffff).count('1') for v in k.wmask.tolist()) / (B * ktq * (P // 32)):.3f}", flush=True)
    return k


F2CAT_REC = True       # pooled target-side operand as records in one pass (False: two kernels)
F2CAT_REC_MAX_PLANE = 12288                                   # csrc/corr_tiled.hip: F2C_MAX_PLANE


def f2cat_records(fmap2, lay):
    """[B,C,H,W] -> records of f2cat [B,C,P] (level-l cell = mean of fmap2 over its 2^l x 2^l pixels, zero in pad cells): the
    A operand of dF1 = s * f2cat . dV^T.  One pass (plane pooled in LDS) when a plane fits, else f2cat + to_records."""
    B, C, H, W = fmap2.shape
    fmap2 = fmap2.contiguous()
    if F2CAT_REC and H * W <= F2CAT_REC_MAX_PLANE:
        f2r = torch.empty(B, C, lay.P, device=fmap2.device, dtype=torch.float32)
        f2r._fs_amax = amax_tensor(fmap2)       # (bounds the pooled levels too: they are means)
        L.check(_lib().fsraft_corr_f2cat_rec(L.ptr(fmap2), L.ptr(f2r), lay.nlev, B, C, H, W, L.ptr(f2r._fs_amax), L.stream()), "corr_f2cat_rec")
        return f2r
    f2cat = torch.empty(B, C, lay.P, device=fmap2.device, dtype=torch.float32)
    L.check(_lib().fsraft_corr_f2cat(L.ptr(fmap2), L.ptr(f2cat), lay.nlev, B, C, H, W, L.stream()), "corr_f2cat")
    return to_records(f2cat)


def corr_bwd_chunked(fmap1, fmap2, douts, coords, lay, radius, is_flow=False, chunk=2048, f1r=None):
    """Backward of volume + lookups WITHOUT an O(N^2) buffer (AlternateCorrBlock's contract, pytorch/core/corr.py:63-91): the
    gradient volume exists for `chunk` queries at a time (chunk x P floats, records), and every chunk feeds the same two
    record GEMMs as the dense path: dF1[:, chunk] = s * f2cat . dV^T and d2cat += s * dV^T . f1[chunk].  (dfmap1, dfmap2) NCHW."""
    B, C, H, W = fmap1.shape
    N, P = H * W, lay.P
    s = 1.0 / math.sqrt(C)
    tm = TIMER
    e0 = tm.begin() if tm else None
    f2r = f2cat_records(fmap2, lay)
    if f1r is None:
        f1r = fmap_records(fmap1)                                   # [B, N, C] records
    Cr = f1r.shape[-1]                                              # record pitch of a pixel's channel vector (ceil32(C) floats)
    d1 = torch.empty(B, C, N, device=fmap1.device, dtype=torch.float32)
    d2cat = torch.zeros(B, P, C, device=fmap1.device, dtype=torch.float32)
    lib = _lib()
    for b in range(B):
        for i0 in range(0, N, chunk):
            n = min(chunk, N - i0)
            # (CHUNK_KSKIP, off: as in the dense path the two GEMMs can visit only the k-tiles the chunk's lookups reached -- measured
            #  at 1 x 47x156 in 2048-query chunks: the gradient rows 0.263 -> 0.227 ms, but the per-chunk pre-pass and the short GEMMs'
            #  fixed costs take more than the skipped k-tiles give back, 0.784 -> 0.889 ms)
            kt = corr_bwd_ktiles(coords, lay, B, radius, is_flow, q0=b * N + i0, nq=n) if (BWD_KSKIP and CHUNK_KSKIP) else None
            dV = corr_dvol_build(douts, coords, lay, B, radius, records=True, is_flow=is_flow, q0=b * N + i0, nq=n,
                                 wmask=kt.wmask if (kt is not None and DVOL_WMASK) else None)
            a_nt = (ctypes.c_void_p(f2r.data_ptr() + b * C * P * 4), P, 0, L.ptr(dV), P, 0,
                    ctypes.c_void_p(d1.data_ptr() + (b * C * N + i0) * 4), N, 0, 1, C, n, P, s)
            a_tn = (L.ptr(dV), P, 0, ctypes.c_void_p(f1r.data_ptr() + (b * N + i0) * Cr * 4), Cr, 0,
                    ctypes.c_void_p(d2cat.data_ptr() + b * P * C * 4), C, 0, 1, P, C, n, s)
            if kt is not None:
                L.check(lib.fsraft_gemm_rec_nt_list(*a_nt, 2, 0, L.ptr(kt.nt_list), L.ptr(kt.nt_count), kt.nt_stride, 1, _wptr(f2r), _wptr(dV), L.stream()), "gemm_rec_nt_list")
                L.check(lib.fsraft_gemm_rec_tn_list(*a_tn, 3, 1, L.ptr(kt.tn_list), L.ptr(kt.tn_count), kt.tn_stride, 0, _wptr(dV), _wptr(f1r), L.stream()), "gemm_rec_tn_list")
            else:
                # d1[b][:, i0:i0+n] = s * f2cat[b] [C x P] . dV [n x P]^T
                L.check(lib.fsraft_gemm_rec_nt(*a_nt, 8, 0, _wptr(f2r), _wptr(dV), L.stream()), "gemm_rec_nt")
                # d2cat[b] [P x C] += s * dV [n x P]^T . f1[b][i0:i0+n] [n x C]
                L.check(lib.fsraft_gemm_rec_tn(*a_tn, 3, 1, _wptr(dV), _wptr(f1r), L.stream()), "gemm_rec_tn")
    d2 = torch.empty(B, H, W, C, device=fmap1.device, dtype=torch.float32)
    L.check(lib.fsraft_corr_dfmap2(L.ptr(d2cat), L.ptr(d2), lay.nlev, B, C, H, W, L.stream()), "corr_dfmap2")
    out = d1.view(B, C, H, W), nhwc_to_nchw(d2, C)
    if tm:   # alt-corr backward of all lookups of the step: compulsory bytes = per lookup dOut + coords, the feature maps and
        #      their gradients; FLOPs = the window dot products of alt_cuda_corr.backward (correlation_kernel.cu:122-256)
        T = len(douts)
        K = lay.nlev * (2 * radius + 1) ** 2
        tm.end("altcorr_bwd", e0, 4.0 * T * B * N * lay.nlev * (2 * radius + 2) ** 2 * C, 4.0 * B * N * (T * (K + 2) + 4 * C))
    return out


ALT_MFMA = True      # AlternateCorrBlock lookups on the matrix pipe (fsraft_altcorr_mfma_fwd) when the records are supplied
ALT_DISPATCH = os.environ.get("FSRAFT_ALT_DISPATCH", "1") != "0"   # alt-corr lookups pick their kernel per launch by flow regime (0: always the matrix-pipe kernel)


def altcorr_fused_fwd(f1_cl, f2_levels, coords, radius, is_flow=False, recs=None, out=None, regime=None):
    """f1_cl [B,H,W,C], f2_levels[l] [B,H>>l,W>>l,C] channels-last -> [B,H,W,L*(2r+1)^2] (scaled by 1/sqrt(C)).
    recs = (f1r, [f2r per level]): the same maps as records -> the tile GEMM kernel (split arithmetic); without them, or with
    the exact-fp32 arithmetic selected, the fp32 dot-product kernels.  regime: 8 zeroed int32 on the device -> the launch
    picks between the two by the spread of the flow (fsraft.h, fsraft_altcorr_mfma_fwd)."""
    L.require_cuda_f32(f1_cl, coords, *f2_levels)
    B, H, W, C = f1_cl.shape
    bs, cs, ps = _planar2_strides(coords)
    nl = len(f2_levels)
    ch = nl * (2 * radius + 1) ** 2
    if out is None:
        out = torch.empty(B, H, W, ch, device=f1_cl.device, dtype=torch.float32)
    elif tuple(out.shape) != (B, H, W, ch) or not out.is_contiguous() or out.dtype != torch.float32:
        raise ValueError(f"lookup out= must be a contiguous fp32 [{B},{H},{W},{ch}] tensor")
    pp, keep = L.ptr_array(f2_levels)
    t = TIMER
    e0 = t.begin() if t else None
    if recs is not None and ALT_MFMA and SPLIT_VOLUME_BWD and C % 32 == 0 and C <= 256:
        pr, keep2 = L.ptr_array(recs[1])
        pw, keep3 = L.ptr_array([amax_of(r) for r in recs[1]])
        L.check(_lib().fsraft_altcorr_mfma_fwd(L.ptr(recs[0]), pr, L.ptr(f1_cl), pp, nl, L.ptr(coords), bs, cs, ps, int(is_flow), L.ptr(out),
                                               B, H, W, C, radius, _wptr(recs[0]), pw, L.ptr(regime) if regime is not None else None,
                                               L.stream()), "altcorr_mfma_fwd")
    else:
        L.check(_lib().fsraft_altcorr_fused_fwd(L.ptr(f1_cl), pp, nl, L.ptr(coords), bs, cs, ps, int(is_flow), L.ptr(out), B, H, W, C,
                                                radius, L.stream()), "altcorr_fused_fwd")
    if amax_of(out) is not None:          # (a tracked destination -- a MotionBatch slot: these kernels raise no word themselves)
        amax_jobs([(out.data_ptr(), 1, out.numel(), out.numel(), amax_of(out))])
    if t:   # compulsory bytes of the alt path per iteration = fmap1 + the pooled fmap2 pyramid + coords + out
        t.end("altcorr_fwd", e0, 2.0 * B * H * W * nl * (2 * radius + 2) ** 2 * C,
              4.0 * B * (H * W * C + sum(f.shape[1] * f.shape[2] for f in f2_levels) * C + H * W * (2 + out.shape[-1])))
    return out


def corr_build_bwd_tiled(fmap1, fmap2, dvol, lay, records=False, f1r=None, ktiles=None):
    """(dfmap1, dfmap2) NCHW from the gradient volume in the tiled-row layout: two GEMMs that contract over whole rows
    (K = P resp. M = P), the pooling chain folded into the pooled operand f2cat and the un-pool of the feature gradient.
    records=True: dvol holds records; both GEMMs run on the LDS-DMA record core (gemm_rec.hpp)."""
    B, C, H, W = fmap1.shape
    N, P = H * W, lay.P
    s = 1.0 / math.sqrt(C)
    tm = TIMER
    e0 = tm.begin() if tm else None
    dV = dvol.view(B, N, P)
    if records:
        f2r = f2cat_records(fmap2, lay)
        if f1r is None:
            f1r = fmap_records(fmap1)
        if ktiles is not None:      # only the k-tiles the step's lookups reached (the rest of the gradient rows is zero records)
            Cr = f1r.shape[-1]
            d1 = torch.empty(B, C, N, device=fmap1.device, dtype=torch.float32)
            d2cat = torch.empty(B, P, C, device=fmap1.device, dtype=torch.float32)
            lib = _lib()
            e1 = tm.begin() if tm else None
            L.check(lib.fsraft_gemm_rec_nt_list(L.ptr(f2r), P, C * P * 4, L.ptr(dV), P, N * P * 4, L.ptr(d1), N, C * N, B, C, N, P, s, NT_LIST_KSPLIT, 0,
                                                L.ptr(ktiles.nt_list), L.ptr(ktiles.nt_count), ktiles.nt_stride, 1, _wptr(f2r), _wptr(dvol),
                                                L.stream()), "gemm_rec_nt_list")
            if tm:      # (algorithmic FLOPs: the dense contraction; flops_done: the listed (128-query tile, 32-cell record) pairs)
                cnt = ktiles.nt_count
                tm.end("gemm_f32", e1, 2.0 * B * C * N * P, 4.0 * B * (C * P + N * P + C * N),
                       flops_done=lambda cnt=cnt: 2.0 * C * 128 * 32 * float(cnt.sum().item()))
                e1 = tm.begin()
            L.check(lib.fsraft_gemm_rec_tn_list(L.ptr(dV), P, N * P * 4, L.ptr(f1r), Cr, N * Cr * 4, L.ptr(d2cat), C, P * C, B, P, C, N, s,
                                                TN_LIST_KSPLIT, 0, L.ptr(ktiles.tn_list), L.ptr(ktiles.tn_count), ktiles.tn_stride, 0,
                                                _wptr(dvol), _wptr(f1r), L.stream()), "gemm_rec_tn_list")
            if tm:      # (flops_done: the listed (256-cell tile, 32-query block) pairs)
                cnt = ktiles.tn_count
                tm.end("gemm_f32", e1, 2.0 * B * P * C * N, 4.0 * B * (N * P + N * C + P * C),
                       flops_done=lambda cnt=cnt: 2.0 * 256 * C * 32 * float(cnt.sum().item()))
        else:
            d1 = gemm_rec_nt(f2r, dV, s, b_amax=amax_of(dvol))                              # [B,C,N] = s * f2cat . dV^T
            d2cat = gemm_rec_tn(dV, f1r, P, C, s, ksplit=2, a_amax=amax_of(dvol))           # [B,P,C] = s * dV^T . f1^T
    else:
        f2cat = torch.empty(B, C, P, device=fmap1.device, dtype=torch.float32)
        L.check(_lib().fsraft_corr_f2cat(L.ptr(fmap2.contiguous()), L.ptr(f2cat), lay.nlev, B, C, H, W, L.stream()), "corr_f2cat")
        f1t = nchw_to_nhwc(fmap1).view(B, N, C)
        d1 = gemm(f2cat, dV, True, s)
        if C % 4 == 0 and SPLIT_VOLUME_BWD:
            d2cat = gemm_tn_split(dV, f1t, s)                     # (both operands k-major)
        else:
            d2cat = gemm(dV.transpose(1, 2).contiguous(), f1t.transpose(1, 2).contiguous(), True, s)
    d2 = torch.empty(B, H, W, C, device=fmap1.device, dtype=torch.float32)
    L.check(_lib().fsraft_corr_dfmap2(L.ptr(d2cat), L.ptr(d2), lay.nlev, B, C, H, W, L.stream()), "corr_dfmap2")
    out = d1.view(B, C, H, W), nhwc_to_nchw(d2, C)
    if tm:   # SURVEY.md 8d, build backward: read dV + read the feature maps + write their gradients (the zero fill of dV is
        #      counted with the lookups' backward); FLOPs of the two contractions over the level-0 volume
        tm.end("corr_build_bwd", e0, 4.0 * B * N * N * C, 4.0 * B * (N * sum(h * w for h, w in zip(lay.h, lay.w)) + 4 * N * C))
    return out


def corr_pool_pyramid(level0, num_levels, same=False):
    """level0 [rows, 1, h, w] (or [rows, h, w]) -> [level0, 2x2 averages, ...] with floor sizes (fsraft_corr_pool_pyramid), or,
    same=True, TensorFlow's avg_pool2d(level0, 2^l, 2^l, 'SAME') levels with ceil sizes (fsraft_corr_pool_pyramid_same)."""
    L.require_cuda_f32(level0)
    level0 = level0.contiguous()
    rows, h, w = level0.shape[0], level0.shape[-2], level0.shape[-1]
    if same:
        levels = [level0.view(rows, 1, h, w)] + [torch.empty(rows, 1, -(-h // (1 << l)), -(-w // (1 << l)), device=level0.device,
                                                             dtype=torch.float32) for l in range(1, num_levels)]
        pp, keep = L.ptr_array(levels)
        L.check(_lib().fsraft_corr_pool_pyramid_same(pp, num_levels, rows, h, w, L.stream()), "corr_pool_pyramid_same")
        return levels
    sizes = pyramid_sizes(h, w, num_levels)
    if sizes[-1][0] < 1 or sizes[-1][1] < 1:
        raise RuntimeError(f"volume {h}x{w} too small for {num_levels} pyramid levels")
    levels = [level0.view(rows, 1, h, w)] + [torch.empty(rows, 1, a, b, device=level0.device, dtype=torch.float32)
                                              for a, b in sizes[1:]]
    pp, keep = L.ptr_array(levels)
    L.check(_lib().fsraft_corr_pool_pyramid(pp, num_levels, rows, h, w, L.stream()), "corr_pool_pyramid")
    return levels


def space_to_depth2(x, inverse=False):
    """x [B,H,W,C] -> [B,H/2,W/2,4C] with channel (sy*2+sx)*C + c = pixel (2y+sy, 2x+sx) (inverse: [B,h,w,4C] -> [B,2h,2w,C])."""
    L.require_cuda_f32(x)
    x = x.contiguous()
    if inverse:
        B, h, w, C4 = x.shape
        out = torch.empty(B, 2 * h, 2 * w, C4 // 4, device=x.device, dtype=torch.float32)
        L.check(_lib().fsraft_space_to_depth2(L.ptr(x), L.ptr(out), B, 2 * h, 2 * w, C4 // 4, 1, L.stream()), "space_to_depth2")
    else:
        B, H, W, C = x.shape
        out = torch.empty(B, H // 2, W // 2, 4 * C, device=x.device, dtype=torch.float32)
        L.check(_lib().fsraft_space_to_depth2(L.ptr(x), L.ptr(out), B, H, W, C, 0, L.stream()), "space_to_depth2")
    return out


def transpose_batched(x):
    """x [B, M, N] contiguous -> [B, N, M] contiguous through the tiled layout kernel (4-6 TB/s)."""
    L.require_cuda_f32(x)
    B, M, N = x.shape
    out = torch.empty(B, N, M, device=x.device, dtype=torch.float32)
    L.check(_lib().fsraft_nhwc_to_nchw(L.ptr(x.contiguous()), L.ptr(out), B, N, M, N, 0, 0, L.stream()), "transpose_batched")
    return out


def corr_unpool_bwd_(dlevels, B, H, W):
    pp, keep = L.ptr_array(dlevels)
    L.check(_lib().fsraft_corr_unpool_bwd(pp, len(dlevels), B, H, W, L.stream()), "corr_unpool_bwd")


def corr_lookup_fwd(levels, coords, radius, nhwc=False, same=False):
    """same=True: the pyramid has TensorFlow 'SAME' (ceil) level sizes."""
    L.require_cuda_f32(coords, *levels)
    B, _, H, W = coords.shape
    bs, cs, ps = _planar2_strides(coords)
    ch = len(levels) * (2 * radius + 1) ** 2
    if nhwc:
        out = torch.empty(B, H, W, ch, device=coords.device, dtype=torch.float32)
    else:
        out = torch.empty(B, ch, H, W, device=coords.device, dtype=torch.float32)
    pp, keep = L.ptr_array(levels)
    t = TIMER
    e0 = t.begin() if t else None
    fn = _lib().fsraft_corr_lookup_fwd_same if same else _lib().fsraft_corr_lookup_fwd
    L.check(fn(pp, len(levels), L.ptr(coords), bs, cs, ps, L.ptr(out), int(nhwc), B, H, W, radius, L.stream()), "corr_lookup_fwd")
    if t:   # per query: L*(2r+2)^2 window floats + 2 coords in, L*(2r+1)^2 out  
        t.end("corr_lookup_fwd", e0, 0.0, 4.0 * B * H * W * (len(levels) * (2 * radius + 2) ** 2 + 2 + ch))
    return out


def corr_lookup_bwd_(dlevels, coords, dout, radius, nhwc=False):
    L.require_cuda_f32(coords, dout, *dlevels)
    B, _, H, W = coords.shape
    bs, cs, ps = _planar2_strides(coords)
    dout = dout.contiguous()
    pp, keep = L.ptr_array(dlevels)
    t = TIMER
    e0 = t.begin() if t else None
    L.check(_lib().fsraft_corr_lookup_bwd(pp, len(dlevels), L.ptr(coords), bs, cs, ps, L.ptr(dout), int(nhwc), B, H, W,
                                          radius, L.stream()), "corr_lookup_bwd")
    if t:   # read dout, read-modify-write the window taps
        nl = len(dlevels)
        t.end("corr_lookup_bwd", e0, 0.0,
              4.0 * B * H * W * (nl * (2 * radius + 1) ** 2 + 2 + 2 * nl * (2 * radius + 2) ** 2))


def gemm(A, Bm, trans_b, alpha=1.0, out=None, accumulate=False):
    """Batched C[b] = alpha * A[b] @ (Bm[b]^T if trans_b else Bm[b]); A [b,M,K], Bm [b,N,K] or [b,K,N]."""
    L.require_cuda_f32(A, Bm)
    A = A.contiguous()
    Bm = Bm.contiguous()
    b, M, K = A.shape
    N = Bm.shape[1] if trans_b else Bm.shape[2]
    if out is None:
        out = torch.empty(b, M, N, device=A.device, dtype=torch.float32)
    t = TIMER
    e0 = t.begin() if t else None
    split = trans_b and not exact_mode()       # (the kernel splits when trans_b and aligned; unused words cost two small launches)
    L.check(_lib().fsraft_gemm_f32(L.ptr(A), K, M * K, L.ptr(Bm), Bm.shape[2], Bm.shape[1] * Bm.shape[2], L.ptr(out),
                                   N, M * N, b, M, N, K, int(trans_b), float(alpha), int(accumulate),
                                   L.ptr(amax_tensor(A)) if split else None, L.ptr(amax_tensor(Bm)) if split else None, L.stream()),
            "gemm_f32")
    if t:
        t.end("gemm_f32", e0, 2.0 * b * M * N * K, 4.0 * b * (M * K + N * K + M * N))
    return out


def rec_pitch(C):
    """Row pitch (floats) of an activation record tensor with C channels: whole records, and an odd number of 128-byte lines
    per row -- with an even count (128 channels = 512 B, 256 = 1 KB) the 256 rows of a k-tile hit every 4th / 8th line,
    i.e. a fraction of the L2 channels, and every workgroup does so at the same time."""
    r = (C + 31) // 32
    return 32 * (r | 1)


def to_records(x, pad=False, amax=None):
    """x [..., K] fp32 contiguous -> [..., ceil32(K)] "records" (same dtype container: every 32 floats of a row are replaced
    by [32 hi | 32 lo] fp16 pieces of x * scale; the padded tail of the last record is zero).  pad=True: row pitch rec_pitch(K).
    amax: the word of x (default: the one x carries, else computed here); the result carries it as `_fs_amax`."""
    L.require_cuda_f32(x)
    x = x.contiguous()
    K = x.shape[-1]
    rows = x.numel() // K
    ldr = rec_pitch(K) if pad else (K + 31) // 32 * 32
    out = torch.empty(*x.shape[:-1], ldr, device=x.device, dtype=torch.float32)
    word = amax if amax is not None else amax_of(x)
    if word is None:
        word = amax_tensor(x)
    elif AMAX_AUDIT and not torch.cuda.is_current_stream_capturing():
        true = float(x.abs().max())
        if not (true == 0.0 or 2.0 * float(word.item()) > true):
            raise RuntimeError(f"amax audit (to_records): word {float(word.item()):g} does not bound max |x| = {true:g}")
    out._fs_amax = word
    L.check(_lib().fsraft_to_records(L.ptr(x), K, L.ptr(out), ldr, rows, K, L.ptr(word), L.stream()), "to_records")
    return out


def gemm_rec_nt(Ar, Br, alpha=1.0, ksplit=1, out=None, accumulate=False, a_amax=None, b_amax=None):
    """C[b] = alpha * A[b] @ B[b]^T on record operands Ar [b,M,K], Br [b,N,K] (outputs of to_records / record producers)."""
    b, M, K = Ar.shape
    N = Br.shape[1]
    if out is None:
        out = torch.empty(b, M, N, device=Ar.device, dtype=torch.float32)
    t = TIMER
    e0 = t.begin() if t else None
    L.check(_lib().fsraft_gemm_rec_nt(L.ptr(Ar), K, M * K * 4, L.ptr(Br), K, N * K * 4, L.ptr(out), N, M * N, b, M, N, K, float(alpha),
                                      int(ksplit), int(accumulate), L.ptr(a_amax) if a_amax is not None else _wptr(Ar),
                                      L.ptr(b_amax) if b_amax is not None else _wptr(Br), L.stream()), "gemm_rec_nt")
    if t:
        t.end("gemm_f32", e0, 2.0 * b * M * N * K, 4.0 * b * (M * K + N * K + M * N))
    return out


def gemm_rec_nt_raw(A, lda, sA, Bm, ldb, sB, C, ldc, sC, batch, M, N, K, alpha=1.0, ksplit=1, accumulate=False, a_amax=None, b_amax=None):
    """fsraft_gemm_rec_nt on explicit (device address, pitch in floats, batch stride in floats) triples."""
    t = TIMER
    e0 = t.begin() if t else None
    L.check(_lib().fsraft_gemm_rec_nt(ctypes.c_void_p(A), lda, sA * 4, ctypes.c_void_p(Bm), ldb, sB * 4, ctypes.c_void_p(C), ldc, sC,
                                      batch, M, N, K, float(alpha), int(ksplit), int(accumulate), L.ptr(a_amax), L.ptr(b_amax), L.stream()), "gemm_rec_nt")
    if t:
        t.end("gemm_f32", e0, 2.0 * batch * M * N * K, 4.0 * batch * (M * K + N * K + M * N))


def gemm_rec_tn_raw(A, lda, sA, Bm, ldb, sB, C, ldc, sC, batch, M, N, K, alpha=1.0, ksplit=1, accumulate=False, a_amax=None, b_amax=None):
    """fsraft_gemm_rec_tn on explicit (device address, pitch in floats, batch stride in floats) triples."""
    t = TIMER
    e0 = t.begin() if t else None
    L.check(_lib().fsraft_gemm_rec_tn(ctypes.c_void_p(A), lda, sA * 4, ctypes.c_void_p(Bm), ldb, sB * 4, ctypes.c_void_p(C), ldc, sC,
                                      batch, M, N, K, float(alpha), int(ksplit), int(accumulate), L.ptr(a_amax), L.ptr(b_amax), L.stream()), "gemm_rec_tn")
    if t:
        t.end("gemm_f32", e0, 2.0 * batch * M * N * K, 4.0 * batch * (M * K + N * K + M * N))


def gemm_rec_tn(Ar, Br, M, N, alpha=1.0, ksplit=1, out=None, accumulate=False, a_amax=None, b_amax=None):
    """C[b] = alpha * A[b]^T @ B[b] on k-major record operands Ar [b,K,lda] (records along m), Br [b,K,ldb] (records along n)."""
    b, K, lda = Ar.shape
    ldb = Br.shape[2]
    if out is None:
        out = torch.empty(b, M, N, device=Ar.device, dtype=torch.float32)
    t = TIMER
    e0 = t.begin() if t else None
    L.check(_lib().fsraft_gemm_rec_tn(L.ptr(Ar), lda, K * lda * 4, L.ptr(Br), ldb, K * ldb * 4, L.ptr(out), N, M * N, b, M, N, K,
                                      float(alpha), int(ksplit), int(accumulate), L.ptr(a_amax) if a_amax is not None else _wptr(Ar),
                                      L.ptr(b_amax) if b_amax is not None else _wptr(Br), L.stream()), "gemm_rec_tn")
    if t:
        t.end("gemm_f32", e0, 2.0 * b * M * N * K, 4.0 * b * (M * K + N * K + M * N))
    return out


def _amax_region(ptr, ld, stride, batch, rows, cols):
    """Word of the [batch][rows][cols] fp32 region at device address ptr (row pitch ld, batch stride `stride`, in floats)."""
    w = new_amax(torch.device("cuda", torch.cuda.current_device()))
    if stride == rows * ld or batch == 1:
        amax_jobs([(ptr, batch * rows, cols, ld, w)])
    else:
        amax_jobs([(ptr + 4 * b * stride, rows, cols, ld, w) for b in range(batch)])
    return w


def gemm_raw(A, lda, sA, Bm, ldb, sB, C, ldc, sC, batch, M, N, K, trans_b, alpha=1.0, accumulate=False, a_amax=None, b_amax=None):
    """fsraft_gemm_f32 on explicit (pointer, pitch, batch stride) triples; A/Bm/C are ints (device addresses).  a_amax / b_amax:
    the operands' words for the split kernel (trans_b); computed here when not given."""
    if trans_b and not exact_mode():
        a_amax = a_amax if a_amax is not None else _amax_region(A, lda, sA, batch, M, K)
        b_amax = b_amax if b_amax is not None else _amax_region(Bm, ldb, sB, batch, N, K)
    t = TIMER
    e0 = t.begin() if t else None
    L.check(_lib().fsraft_gemm_f32(ctypes.c_void_p(A), lda, sA, ctypes.c_void_p(Bm), ldb, sB, ctypes.c_void_p(C), ldc, sC,
                                   batch, M, N, K, int(trans_b), float(alpha), int(accumulate), L.ptr(a_amax), L.ptr(b_amax), L.stream()), "gemm_f32")
    if t:
        t.end("gemm_f32", e0, 2.0 * batch * M * N * K, 4.0 * batch * (M * K + N * K + M * N))


def gemm_tn_raw(A, lda, sA, Bm, ldb, sB, C, ldc, sC, batch, M, N, K, alpha=1.0, accumulate=False, a_amax=None, b_amax=None):
    """C[b][m][n] (+)= alpha * sum_k A[b][k][m] Bm[b][k][n] on the split core (M, N, pitches % 4 == 0); a_amax / b_amax: the
    operands' words, computed here when not given."""
    a_amax = a_amax if a_amax is not None else _amax_region(A, lda, sA, batch, K, M)
    b_amax = b_amax if b_amax is not None else _amax_region(Bm, ldb, sB, batch, K, N)
    t = TIMER
    e0 = t.begin() if t else None
    L.check(_lib().fsraft_gemm_tn_split(ctypes.c_void_p(A), lda, sA, ctypes.c_void_p(Bm), ldb, sB, ctypes.c_void_p(C), ldc,
                                        sC, batch, M, N, K, float(alpha), int(accumulate), L.ptr(a_amax), L.ptr(b_amax), L.stream()), "gemm_tn_split")
    if t:
        t.end("gemm_f32", e0, 2.0 * batch * M * N * K, 4.0 * batch * (M * K + N * K + M * N))


# ------------------------------------------------------------------ GMA (attention / aggregate)
def softmax_rows_(S):
    """In-place softmax over the last dimension of a contiguous tensor (gma.py:74)."""
    L.require_cuda_f32(S)
    n = S.shape[-1]
    L.check(_lib().fsraft_softmax_rows(L.ptr(S), S.numel() // n, n, L.stream()), "softmax_rows")
    return S


def softmax_rows_bwd_(A, dA):
    """dA <- A * (dA - sum(dA * A, -1)) in place."""
    L.require_cuda_f32(A, dA)
    n = A.shape[-1]
    L.check(_lib().fsraft_softmax_rows_bwd(L.ptr(A), L.ptr(dA), A.numel() // n, n, L.stream()), "softmax_rows_bwd")
    return dA


def softmax_rows_rec_(S):
    """In-place softmax over the last dimension (n % 32 == 0, n <= 16352) whose result overwrites the logits as RECORDS (the
    operand form of gemm_rec_nt / gemm_rec_tn, see to_records): the tensor keeps its shape and dtype, its bytes are records."""
    L.require_cuda_f32(S)
    n = S.shape[-1]
    L.check(_lib().fsraft_softmax_rows_rec(L.ptr(S), S.numel() // n, n, L.stream()), "softmax_rows_rec")
    S._fs_amax = amax_one(S.device)         # probabilities: the records are split with the scale of a word holding 1.0
    return S


def softmax_rows_bwd_rec_(Ar, dA):
    """Ar: records of softmax_rows_rec_; dA: fp32 gradient, overwritten with the RECORDS of A * (dA - sum(dA * A, -1))."""
    L.require_cuda_f32(Ar, dA)
    n = Ar.shape[-1]
    word = amax_scaled(amax_tensor(dA), 2.0)          # |A (dA - <dA, A>)| <= 2 max |dA|
    L.check(_lib().fsraft_softmax_rows_bwd_rec(L.ptr(Ar), L.ptr(dA), Ar.numel() // n, n, L.ptr(word), L.stream()), "softmax_rows_bwd_rec")
    dA._fs_amax = word
    return dA


def gma_mix_fwd(x, y, gamma, dst):
    """dst = x + gamma * y over V channel slices (gma.py:113); gamma: 1-element device tensor."""
    M = x.t.numel() // x.ld
    L.check(_lib().fsraft_gma_mix_fwd(ctypes.c_void_p(x.ptr), x.ld, ctypes.c_void_p(y.ptr), y.ld, L.ptr(gamma),
                                      ctypes.c_void_p(dst.ptr), dst.ld, M, x.C, L.ptr(dst.amax), L.stream()), "gma_mix_fwd")


def gma_mix_bwd(d, y, gamma, dx, dy, dgamma):
    """dx += d; dy = gamma * d; dgamma += sum(d * y)."""
    M = d.t.numel() // d.ld
    L.check(_lib().fsraft_gma_mix_bwd(ctypes.c_void_p(d.ptr), d.ld, ctypes.c_void_p(y.ptr), y.ld, L.ptr(gamma),
                                      ctypes.c_void_p(dx.ptr), dx.ld, ctypes.c_void_p(dy.ptr), dy.ld, L.ptr(dgamma), M, d.C,
                                      L.ptr(dx.amax), L.ptr(dy.amax), L.stream()), "gma_mix_bwd")


SPLIT_VOLUME_BWD = True     # dF2 through the k-major split (fp16x3) GEMM (needs H*W % 4 == 0)


def gemm_tn_split(A, Bm, alpha=1.0):
    """C[b] = alpha * A[b]^T @ Bm[b] with A [b,K,M], Bm [b,K,N] (both k-major), split (fp16x3) core."""
    L.require_cuda_f32(A, Bm)
    b, K, M = A.shape
    N = Bm.shape[2]
    out = torch.empty(b, M, N, device=A.device, dtype=torch.float32)
    t = TIMER
    e0 = t.begin() if t else None
    L.check(_lib().fsraft_gemm_tn_split(L.ptr(A), M, K * M, L.ptr(Bm), N, K * N, L.ptr(out), N, M * N, b, M, N, K,
                                        float(alpha), 0, L.ptr(amax_tensor(A)), L.ptr(amax_tensor(Bm)), L.stream()), "gemm_tn_split")
    if t:
        t.end("gemm_f32", e0, 2.0 * b * M * N * K, 4.0 * b * (M * K + N * K + M * N))
    return out


def corr_build_bwd(fmap1, fmap2, dlevels):
    """dlevels: accumulated dL/dV_l (modified in place).  Returns (dfmap1, dfmap2) as NCHW."""
    B, C, H, W = fmap1.shape
    N = H * W
    corr_unpool_bwd_(dlevels, B, H, W)
    dV = dlevels[0].view(B, N, N)
    s = 1.0 / math.sqrt(C)
    f1 = fmap1.contiguous().view(B, C, N)
    f2 = fmap2.contiguous().view(B, C, N)
    d1 = gemm(f2, dV, True, s)     # [B,C,N]: sum_j f2[c][j] dV[i][j]
    if SPLIT_VOLUME_BWD and N % 4 == 0 and C % 4 == 0:
        # dF2^T[j][c] = sum_i dV[i][j] f1^T[i][c]: both operands k-major -> transposed-read split GEMM
        f1t = nchw_to_nhwc(fmap1).view(B, N, C)
        d2t = gemm_tn_split(dV, f1t, s)                       # [B, N(j), C]
        d2 = nhwc_to_nchw(d2t.view(B, H, W, C), C)
        return d1.view(B, C, H, W), d2
    d2 = gemm(f1, dV, False, s)    # [B,C,N]: sum_i f1[c][i] dV[i][j]
    return d1.view(B, C, H, W), d2.view(B, C, H, W)


# ------------------------------------------------------------------ alternate (on-the-fly) correlation
def _altcorr_sets(coords, B, H1, W1):
    """N of coords [B,N,H1,W1,2] (correlation.cpp:23-33: any number of coordinate sets per query pixel)."""
    if coords.dim() != 5 or coords.shape[0] != B or tuple(coords.shape[2:]) != (H1, W1, 2) or coords.shape[1] < 1:
        raise RuntimeError(f"coords must be [B,N,H1,W1,2] with B={B}, H1={H1}, W1={W1}, got {tuple(coords.shape)}")
    return int(coords.shape[1])


def altcorr_fwd(fmap1, fmap2, coords, radius):
    L.require_cuda_f32(fmap1, fmap2, coords)
    for t, n in ((fmap1, "fmap1"), (fmap2, "fmap2"), (coords, "coords")):
        if not t.is_contiguous():
            raise RuntimeError(f"{n} must be contiguous")       # correlation.cpp:19-21
    B, H1, W1, C = fmap1.shape
    _, H2, W2, _ = fmap2.shape
    N = _altcorr_sets(coords, B, H1, W1)
    rd = 2 * radius + 1
    corr = torch.empty(B, N, rd * rd, H1, W1, device=fmap1.device, dtype=torch.float32)
    L.check(_lib().fsraft_altcorr_fwd(L.ptr(fmap1), L.ptr(fmap2), L.ptr(coords), L.ptr(corr), B, N, H1, W1, H2, W2, C,
                                      radius, L.stream()), "altcorr_fwd")
    return corr


def altcorr_bwd(fmap1, fmap2, coords, corr_grad, radius):
    L.require_cuda_f32(fmap1, fmap2, coords, corr_grad)
    for t, n in ((fmap1, "fmap1"), (fmap2, "fmap2"), (coords, "coords"), (corr_grad, "corr_grad")):
        if not t.is_contiguous():
            raise RuntimeError(f"{n} must be contiguous")
    B, H1, W1, C = fmap1.shape
    _, H2, W2, _ = fmap2.shape
    N = _altcorr_sets(coords, B, H1, W1)
    rd = 2 * radius + 1
    if tuple(corr_grad.shape) != (B, N, rd * rd, H1, W1):
        raise RuntimeError(f"corr_grad must be [B,N,(2r+1)^2,H1,W1] = {(B, N, rd * rd, H1, W1)}, got {tuple(corr_grad.shape)}")
    g1 = torch.empty_like(fmap1)
    g2 = torch.zeros_like(fmap2)
    L.check(_lib().fsraft_altcorr_bwd(L.ptr(fmap1), L.ptr(fmap2), L.ptr(coords), L.ptr(corr_grad), L.ptr(g1),
                                      L.ptr(g2), B, N, H1, W1, H2, W2, C, radius, L.stream()), "altcorr_bwd")
    return g1, g2, torch.zeros_like(coords)


# ------------------------------------------------------------------ upsamplers
def upsample_fwd(flow, mask_nhwc):
    L.require_cuda_f32(flow, mask_nhwc)
    N, _, H, W = flow.shape
    bs, cs, ps = _planar2_strides(flow)
    up = torch.empty(N, 2, 8 * H, 8 * W, device=flow.device, dtype=torch.float32)
    t = TIMER
    e0 = t.begin() if t else None
    L.check(_lib().fsraft_upsample_fwd(L.ptr(flow), bs, cs, ps, L.ptr(mask_nhwc), L.ptr(up), N, H, W, L.stream()),
            "upsample_fwd")
    if t:
        t.end("upsample_fwd", e0, 0.0, 4.0 * N * H * W * (576 + 128 + 2))
    return up


def upsample_bwd(flow, mask_nhwc, dup):
    L.require_cuda_f32(flow, mask_nhwc, dup)
    N, _, H, W = flow.shape
    bs, cs, ps = _planar2_strides(flow)
    dup = dup.contiguous()
    dmask = tracked(torch.empty_like(mask_nhwc))          # (its only writer, below, raises the word: the mask head's backward reads it)
    dflow = torch.empty(N, 2, H, W, device=flow.device, dtype=torch.float32)
    scratch = torch.empty(N * H * W * 18, device=flow.device, dtype=torch.float32)
    L.check(_lib().fsraft_upsample_bwd(L.ptr(flow), bs, cs, ps, L.ptr(mask_nhwc), L.ptr(dup), L.ptr(dmask),
                                       L.ptr(dflow), L.ptr(scratch), N, H, W, _wptr(dmask), L.stream()), "upsample_bwd")
    return dflow, dmask


def upflow8_fwd(flow):
    L.require_cuda_f32(flow)
    flow = flow.contiguous()
    N, C, H, W = flow.shape
    up = torch.empty(N, C, 8 * H, 8 * W, device=flow.device, dtype=torch.float32)
    L.check(_lib().fsraft_upflow8_fwd(L.ptr(flow), L.ptr(up), N, C, H, W, L.stream()), "upflow8_fwd")
    return up


def upflow8_bwd(dup, H, W):
    dup = dup.contiguous()
    N, C = dup.shape[:2]
    dflow = torch.empty(N, C, H, W, device=dup.device, dtype=torch.float32)
    L.check(_lib().fsraft_upflow8_bwd(L.ptr(dup), L.ptr(dflow), N, C, H, W, L.stream()), "upflow8_bwd")
    return dflow


# ------------------------------------------------------------------ layout helpers
def nchw_to_nhwc(src, dst=None, coff=0, accumulate=False):
    """src [B,C,H,W] contiguous -> dst[..., coff:coff+C] of a [B,H,W,ld] buffer."""
    L.require_cuda_f32(src)
    src = src.contiguous()
    B, C, H, W = src.shape
    if dst is None:
        # (padding channels must read as zero; a multiple-of-4 channel count has none and needs no fill)
        dst = (torch.empty if C % 4 == 0 else torch.zeros)(B, H, W, (C + 3) // 4 * 4, device=src.device, dtype=torch.float32)
    ld = dst.shape[-1]
    L.check(_lib().fsraft_nchw_to_nhwc(L.ptr(src), L.ptr(dst), B, C, H * W, ld, coff, int(accumulate), L.stream()),
            "nchw_to_nhwc")
    return dst


def nhwc_to_nchw(src, C=None, coff=0, dst=None, accumulate=False):
    """src [B,H,W,ld] -> [B,C,H,W] contiguous taking channels coff:coff+C."""
    L.require_cuda_f32(src)
    B, H, W, ld = src.shape
    C = ld if C is None else C
    if dst is None:
        dst = torch.empty(B, C, H, W, device=src.device, dtype=torch.float32)
    L.check(_lib().fsraft_nhwc_to_nchw(L.ptr(src), L.ptr(dst), B, C, H * W, ld, coff, int(accumulate), L.stream()),
            "nhwc_to_nchw")
    return dst


def im2col7(flow, cols):
    B, _, H, W = flow.shape
    bs, cs, ps = _planar2_strides(flow)
    L.check(_lib().fsraft_im2col7(L.ptr(flow), bs, cs, ps, L.ptr(cols), cols.shape[-1], B, H, W, _wptr(cols), L.stream()), "im2col7")
    return cols


def col2im7(dcols, dflow, accumulate):
    B, _, H, W = dflow.shape
    L.check(_lib().fsraft_col2im7(L.ptr(dcols), dcols.shape[-1], L.ptr(dflow), B, H, W, int(accumulate), L.stream()),
            "col2im7")


def flow_to_nhwc(flow, dst, coff):
    B, _, H, W = flow.shape
    bs, cs, ps = _planar2_strides(flow)
    L.check(_lib().fsraft_flow_to_nhwc(L.ptr(flow), bs, cs, ps, L.ptr(dst), dst.shape[-1], coff, B, H * W, _wptr(dst), L.stream()),
            "flow_to_nhwc")


def nhwc_to_flow(src, coff, dflow, accumulate):
    B, _, H, W = dflow.shape
    L.check(_lib().fsraft_nhwc_to_flow(L.ptr(src), src.shape[-1], coff, L.ptr(dflow), B, H * W, int(accumulate),
                                       L.stream()), "nhwc_to_flow")


def relu_bwd_(g, y, C):
    M = g.numel() // g.shape[-1]
    L.check(_lib().fsraft_relu_bwd(L.ptr(g), g.shape[-1], L.ptr(y), y.shape[-1], M, C, L.stream()), "relu_bwd")


def gru_bwd1(dhn, z, q, h, dzr, dq, dh, hid, dzr_sum=None, dq_sum=None, dhn2=None):
    """dhn2: optional second summand of the incoming hidden-state gradient (same shape as dhn)."""
    M = dhn.numel() // hid
    L.check(_lib().fsraft_gru_bwd1(L.ptr(dhn), L.ptr(dhn2), L.ptr(z), L.ptr(q), L.ptr(h), L.ptr(dzr), dzr.shape[-1], L.ptr(dq),
                                   L.ptr(dh), L.ptr(dzr_sum), L.ptr(dq_sum), M, hid, _wptr(dzr), _wptr(dq), _wptr(dh), L.stream()), "gru_bwd1")


def gru_bwd2(drh, r, h, dzr, dh, hid, dzr_sum=None):
    M = drh.numel() // hid
    L.check(_lib().fsraft_gru_bwd2(L.ptr(drh), L.ptr(r), L.ptr(h), L.ptr(dzr), dzr.shape[-1], L.ptr(dh), L.ptr(dzr_sum), M, hid,
                                   _wptr(dzr), _wptr(dh), L.stream()), "gru_bwd2")


def col_sum_(x, C, out, scale=1.0):
    M = x.numel() // x.shape[-1]
    L.check(_lib().fsraft_col_sum(L.ptr(x), x.shape[-1], M, C, L.ptr(out), float(scale), L.stream()), "col_sum")


def sum_n_(tensors, out, accumulate=False):
    """out.flatten()[:n] = (accumulate ? same : 0) + sum of the tensors (n = their common element count, added in list order); out
    may be larger (sample-major buffers: the leading samples are written)."""
    n = tensors[0].numel()
    for t in tensors:
        L.require_cuda_f32(t)
        if t.numel() != n or not t.is_contiguous():
            raise ValueError("sum_n_: contiguous tensors of one size")
    if out.numel() < n or not out.is_contiguous():
        raise ValueError("sum_n_: out too small / not contiguous")
    pp, keep = L.ptr_array(tensors)
    L.check(_lib().fsraft_sum_n(pp, len(tensors), L.ptr(out), n, int(accumulate), L.stream()), "sum_n")
    return out


def axpby_(x, y, a=1.0, b=1.0):
    L.check(_lib().fsraft_axpby(L.ptr(x), L.ptr(y), float(a), float(b), x.numel(), L.stream()), "axpby")


# ------------------------------------------------------------------ convolutions
def conv_ktot(srcC, KH, KW):
    return _lib().fsraft_conv_ktot(L.int_array(srcC), len(srcC), KH, KW)


def pack_weight(w, srcC, mode):
    """w: [Cout,Cin,KH,KW] contiguous.  mode 0 -> [Cout,Ktot]; mode 1 -> [Cin,Ktot'] (data gradient);
    modes 10 / 11: same matrices with every 32-k run stored as [32 hi | 32 lo] fp16 (split (fp16x3) core)."""
    L.require_cuda_f32(w)
    w = w.contiguous()
    Cout, Cin, KH, KW = w.shape
    if mode % 10 == 0:
        rows, kt = Cout, conv_ktot(srcC, KH, KW)
    else:
        rows, kt = Cin, conv_ktot([Cout], KH, KW)
    wpk = torch.empty(rows, kt, device=w.device, dtype=torch.float32)
    word = amax_tensor(w) if mode >= 10 else None       # split packs hold fp16 pieces of w * scale(amax)
    L.check(_lib().fsraft_pack_conv_weight(L.ptr(w), L.ptr(wpk), Cout, Cin, KH, KW, L.int_array(srcC), len(srcC), mode,
                                           0, L.ptr(word), L.stream()), "pack_conv_weight")
    if word is not None:
        wpk._fs_amax = word
    return wpk


class PackPlan:
    """Jobs for fsraft_pack_conv_weights: every packed matrix of a module in ceil(n / 16) launches, carved out of one
    allocation, read in place from the parameters (fused layers, channel selections and the space-to-depth rewrite of a
    stride-2 weight are address arithmetic of the kernel, not torch.cat / slicing / scatter launches)."""

    def __init__(self, device):
        self.device = device
        self.jobs = []          # (PackJob, offset in floats, shape)
        self.total = 0
        self.keep = []
        self.words = {}         # split packs: tuple of the pieces' data_ptrs -> amax word shared by every pack of those weights
        self.amax_jobs = []

    def _job(self, ws, cin_full, kh, kw, srcC, srcOff, mode, flags=0, scale=1.0, accumulate=False):
        j = L.PackJob()
        for i, w in enumerate(ws):
            L.require_cuda_f32(w)
            assert w.is_contiguous()
            j.w[i] = w.data_ptr()
            j.rows[i] = w.shape[0]
        j.npiece = len(ws)
        j.cin_full, j.kh, j.kw = cin_full, kh, kw
        for i, (c, o) in enumerate(zip(srcC, srcOff)):
            j.srcC[i], j.srcOff[i] = c, o
        j.nsrc = len(srcC)
        j.mode, j.flags, j.scale, j.accumulate = mode, flags, scale, int(accumulate)
        self.keep.append(ws)
        return j

    def pack(self, ws, srcC, mode, srcOff=None, kh=None, kw=None, cin_full=None, frag=False, s2d=False):
        """Queue one packed matrix (modes 0 / 1 / 10 / 11); returns a handle for result()."""
        w0 = ws[0]
        cin_full = w0.shape[1] if cin_full is None else cin_full
        kh = (2 if s2d else w0.shape[2]) if kh is None else kh
        kw = (2 if s2d else w0.shape[3]) if kw is None else kw
        srcOff = [sum(srcC[:i]) for i in range(len(srcC))] if srcOff is None else srcOff
        cout = sum(w.shape[0] for w in ws)
        if mode % 10 == 0:
            rows, kt = cout, conv_ktot(srcC, kh, kw)
        else:
            rows, kt = sum(srcC), conv_ktot([cout], kh, kw)
        if frag:
            rows = (rows + 31) // 32 * 32
        j = self._job(ws, cin_full, kh, kw, srcC, srcOff, mode, (1 if frag else 0) | (2 if s2d else 0))
        if mode >= 10:
            # fp16 pieces of w * scale: the scale comes from the largest magnitude of the pieces' WHOLE tensors (an upper
            # bound for the channel ranges a job uses), one word per set of pieces, computed in run() ahead of the packs
            key = tuple(w.data_ptr() for w in ws)
            word = self.words.get(key)
            if word is None:
                word = self.words[key] = new_amax(self.device)
                self.amax_jobs += [(w.data_ptr(), 1, w.numel(), w.numel(), word) for w in ws]
            j.amax = word.data_ptr()
            j._word = word
        self.jobs.append((j, self.total, (rows, kt)))
        self.total += (rows * kt + 63) // 64 * 64
        return len(self.jobs) - 1

    def bias(self, bs):
        j = self._job(bs, 1, 1, 1, [1], [0], 3)
        n = sum(b.shape[0] for b in bs)
        self.jobs.append((j, self.total, (n,)))
        self.total += (n + 63) // 64 * 64
        return len(self.jobs) - 1

    def run(self):
        """Launch; returns the list of packed tensors in queue order (views of one buffer)."""
        buf = torch.empty(max(self.total, 1), device=self.device, dtype=torch.float32)
        arr = (L.PackJob * max(len(self.jobs), 1))()
        outs = []
        for i, (j, off, shape) in enumerate(self.jobs):
            n = 1
            for v in shape:
                n *= v
            t = buf[off:off + n].view(shape)
            j.wpk = t.data_ptr()
            arr[i] = j
            if getattr(j, "_word", None) is not None:
                t._fs_amax = j._word
            outs.append(t)
        amax_jobs(self.amax_jobs)
        if self.jobs:
            L.check(_lib().fsraft_pack_conv_weights(arr, len(self.jobs), L.stream()), "pack_conv_weights")
        self.keep = []
        return outs


def unpack_weight_grads(items, device):
    """items: (dwpk, grad_tensors, srcC, srcOff, cin_full, kh, kw, scale, s2d) per packed gradient; grad_tensors are the
    parameter-shaped pieces (stacked along the output channels) the gradient is written into, in place, by one launch per 16
    items -- the reverse of PackPlan."""
    if not items:
        return
    arr = (L.PackJob * len(items))()
    plan = PackPlan(device)
    for i, (dwpk, gs, srcC, srcOff, cin_full, kh, kw, scale, s2d) in enumerate(items):
        j = plan._job(gs, cin_full, kh, kw, srcC, srcOff, 2, 2 if s2d else 0, scale)
        j.wpk = dwpk.data_ptr()
        arr[i] = j
    L.check(_lib().fsraft_pack_conv_weights(arr, len(items), L.stream()), "unpack_conv_weights")


def exact_mode():
    """True while the exact-fp32 convolution kernels are selected (fsraft_set_arithmetic(0)); in split (fp16x3) mode only
    layers with <= 32 outputs still read the fp32 packs."""
    return _lib().fsraft_get_tuning(3) == 0


def set_arithmetic(split):
    """fsraft_set_arithmetic: True / 1 = every GEMM-shaped kernel on fp16x3 products of scaled operands (the default), False / 0 = exact fp32
    MFMA.  Also picks the matching host-side routes (record operands for the volume backward and the GMA GEMMs)."""
    global SPLIT_VOLUME_BWD
    L.check(_lib().fsraft_set_arithmetic(1 if split else 0), "set_arithmetic")
    SPLIT_VOLUME_BWD = bool(split)


def pack_pair(w, srcC, dgrad=False):
    """(fp32 pack, split pack) of a weight for the forward (modes 0 / 10) or data-gradient (1 / 11) GEMM.  The fp32 pack is
    only built when a kernel will read it; otherwise the split pack stands in for it (same shape, never dereferenced as fp32)."""
    m = 1 if dgrad else 0
    split = pack_weight(w, srcC, 10 + m)
    rows = w.shape[1] if dgrad else w.shape[0]
    return (pack_weight(w, srcC, m) if (exact_mode() or rows <= 32) else split), split


def fragment_order(wps):
    """Split pack [N, Ktot] (modes 10 / 11) -> fragment order for fsraft_conv_desc.wpk_frag: [k-tile][32-row block]
    [hi k0-15, hi k16-31, lo k0-15, lo k16-31][lane = 32 * (k half) + row][4 dwords], rows zero-padded to a multiple of 32."""
    N, ktot = wps.shape
    kt, nb = ktot // 32, (N + 31) // 32
    w = wps.contiguous().view(torch.int32)
    if nb * 32 != N:
        w = torch.cat([w, w.new_zeros(nb * 32 - N, ktot)])
    w = w.view(nb, 32, kt, 2, 2, 2, 4).permute(2, 0, 3, 4, 5, 1, 6).contiguous()      # kt, nb, hi/lo, s, k half, row, dword
    out = w.view(torch.float32).view(kt, nb, 4, 64, 4)
    if amax_of(wps) is not None:
        out._fs_amax = amax_of(wps)
    return out


def unpack_weight_grad(dwpk, shape, srcC, out=None, accumulate=False):
    Cout, Cin, KH, KW = shape
    if out is None:
        out = torch.empty(shape, device=dwpk.device, dtype=torch.float32)
        accumulate = False
    L.check(_lib().fsraft_pack_conv_weight(L.ptr(out), L.ptr(dwpk), Cout, Cin, KH, KW, L.int_array(srcC), len(srcC), 2,
                                           int(accumulate), None, L.stream()), "unpack_conv_weight")
    return out


class V:
    """Channels [off, off+C) of a channels-last buffer [B,H,W,ld] (pitch = the buffer's ld)."""
    __slots__ = ("t", "off", "C", "ld", "amax")

    def __init__(self, t, C=None, off=0, amax=None):
        self.t, self.off, self.ld = t, off, t.shape[-1]
        self.C = (t.shape[-1] - off) if C is None else C
        self.amax = amax if amax is not None else amax_of(t)      # amax word (1-element tensor) or None: computed on demand
        assert off % 4 == 0 and self.ld % 4 == 0, "channel slices must stay 16-byte aligned"

    @property
    def ptr(self):
        return self.t.data_ptr() + 4 * self.off


class Dst:
    """Output channel range starting at GEMM column n0 -> strided destination."""
    __slots__ = ("t", "off", "bs", "ps", "cs", "n0", "acc", "mask", "amax")

    def __init__(self, t, off, bs, ps, cs, n0=0, acc=False):
        self.t, self.off, self.bs, self.ps, self.cs, self.n0, self.acc = t, off, bs, ps, cs, n0, acc
        self.mask = None            # optional V: ReLU-backward mask applied to this range by the epilogue
        self.amax = amax_of(t)      # the buffer's amax word, raised by the epilogue (None: the buffer carries none)

    def masked(self, v):
        """Zero the written values where the forward activation v (a V over the same channels) is <= 0."""
        self.mask = v
        return self

    @staticmethod
    def nhwc(buf, coff=0, n0=0, acc=False):
        B, H, W, ld = buf.shape
        return Dst(buf, coff, H * W * ld, ld, 1, n0, acc)

    @staticmethod
    def nchw(buf, n0=0, acc=False):
        B, C, H, W = buf.shape
        return Dst(buf, 0, C * H * W, 1, H * W, n0, acc)


_CONV_WS = {}                 # (device index, stream, host thread) -> scratch tensor
CONV_WS_FLOATS = 24 << 20     # 96 MB: three slices of the largest small-M layer (8832 pixels x 512 outputs)
CONV_WS_MAX_PIXELS = 16384    # the split-K route only exists for grids that leave CUs idle


def _conv_workspace(device, pixels):
    """The split-K scratch buffer of a small convolution enqueued on `device`'s current stream, or None (the C ABI allocates
    nothing; the buffer travels in the call's descriptor, fsraft_conv_desc.ws).  One buffer per (device, stream): convolutions
    issued on two streams (core/l2l.py runs the supervisor's encoders beside the student's iterations) or by two host threads
    (one per device, the reference's nn.DataParallel caller) never share slabs, and nothing is registered process-wide.
    ADVICE r5: the host thread is part of the key as well -- two threads enqueueing on ONE stream of one device (legal, if unusual)
    would otherwise interleave their partial-tile and finish launches over one buffer."""
    if pixels > CONV_WS_MAX_PIXELS:
        return None
    key = (torch.device(device).index or 0, torch.cuda.current_stream(device).cuda_stream, threading.get_ident())
    ws = _CONV_WS.get(key)
    if ws is None:
        ws = _CONV_WS.setdefault(key, torch.empty(CONV_WS_FLOATS, device=device, dtype=torch.float32))
    return ws


def conv_forward(srcs, wpk, bias, B, H, W, KH, KW, N, dsts, relu=False, alpha=1.0, epi=0, h=None, z=None,
                 aux1=None, aux2=None, hid=0, wpk_split=None, pre=None, wpk_frag=None, pad=None, stats=None):
    """srcs: list of V (concatenated along channels).  dsts: list of Dst.
    GRU epilogues (epi 2: z|r, epi 3: q) take h, z, aux buffers as [B,H,W,ld] tensors.
    stats: a zeroed [2, B * 8, N] tensor -> the kernel adds the per-image column sums of its result and of their squares to it
    where it can (fsraft_conv_forward_stats); returns True if it did (False / None otherwise)."""
    d = L.ConvDesc()
    for i, v in enumerate(srcs):
        d.src[i] = v.ptr; d.srcC[i] = v.C; d.srcld[i] = v.ld
    d.nsrc = len(srcs)
    d.wpk = wpk.data_ptr()
    d.wpk_split = wpk_split.data_ptr() if wpk_split is not None else None
    d.wpk_frag = wpk_frag.data_ptr() if wpk_frag is not None else None
    if pad is not None:                       # (rows above, columns left of) the output pixel; default: centred taps
        d.pad_h1, d.pad_w1 = pad[0] + 1, pad[1] + 1
    d.bias = bias.data_ptr() if bias is not None else None
    d.B, d.H, d.W, d.KH, d.KW, d.N = B, H, W, KH, KW, N
    for i, ds in enumerate(dsts):
        d.dst[i] = ds.t.data_ptr() + 4 * ds.off
        d.dst_bs[i], d.dst_ps[i], d.dst_cs[i] = ds.bs, ds.ps, ds.cs
        d.dst_n0[i] = ds.n0; d.dst_acc[i] = int(ds.acc)
        if ds.mask is not None:
            d.rmask[i] = ds.mask.ptr; d.ldmask[i] = ds.mask.ld; d.maskc[i] = ds.mask.C
    d.ndst = len(dsts)
    d.relu = int(relu); d.alpha = float(alpha); d.epi = epi
    if h is not None:
        d.h = h.data_ptr(); d.ldh = h.shape[-1]
    if z is not None:
        d.z = z.data_ptr(); d.ldz = z.shape[-1]
    if aux1 is not None:
        d.aux1 = aux1.data_ptr(); d.ld1 = aux1.shape[-1]
    if aux2 is not None:
        d.aux2 = aux2.data_ptr(); d.ld2 = aux2.shape[-1]
    d.hid = hid
    if pre is not None:
        d.pre = pre.data_ptr(); d.ldpre = pre.shape[-1]
    ws = _conv_workspace(dsts[0].t.device, B * H * W)
    if ws is not None:
        d.ws = ws.data_ptr(); d.ws_floats = ws.numel()
    split = (wpk_split is not None or wpk_frag is not None) and not exact_mode()
    if _NO_WORDS & 1:        # (measurement only, scripts/conv_micro.py: the kernels then run with scale 1 / raise nothing)
        split = False
    if split:
        # split arithmetic: scales from the amax words of the sources and of the weights
        wsrc = wpk_split if wpk_split is not None else wpk_frag
        ww = amax_of(wsrc)
        if ww is None:
            raise RuntimeError("conv_forward: the split weight pack carries no amax word (pack it with ops.pack_weight / PackPlan)")
        d.w_amax = ww.data_ptr()
        for i, a in enumerate(ensure_amax(srcs)):
            d.src_amax[i] = a
    for i, ds in enumerate(dsts):
        if ds.amax is not None and not (_NO_WORDS & 2):
            d.dst_amax[i] = ds.amax.data_ptr()
    if epi == 2 and aux1 is not None and amax_of(aux1) is not None:       # r*h
        d.dst_amax[1] = amax_of(aux1).data_ptr()
    t = TIMER
    e0 = t.begin() if t else None
    carried = None
    if stats is not None:
        done = ctypes.c_int(0)
        L.check(_lib().fsraft_conv_forward_stats(ctypes.byref(d), L.ptr(stats[0]), L.ptr(stats[1]), stats.shape[1] // B, ctypes.byref(done),
                                                 L.stream()), "conv_forward_stats")
        carried = bool(done.value)
    else:
        L.check(_lib().fsraft_conv_forward(ctypes.byref(d), L.stream()), "conv_forward")
    if t:
        cin = sum(v.C for v in srcs)
        t.end("conv_igemm", e0, 2.0 * B * H * W * N * cin * KH * KW, 4.0 * B * H * W * (cin + N),
              tag=f"{B}x{H}x{W} {KH}x{KW} {'+'.join(str(v.C) for v in srcs)}->{N} epi{epi}{' relu' if relu else ''}{' masked' if any(ds.mask is not None for ds in dsts) else ''}")
    return carried


def conv_wgrad(dy, srcs, dwpk, B, H, W, KH, KW, dbias=None):
    """dwpk [Cout,Ktot] += dy^T im2col(srcs);  dbias [Cout] += column sums of dy (optional).
    dy: V over the (already act'-scaled) output gradient."""
    arr = (ctypes.c_void_p * len(srcs))(*[v.ptr for v in srcs])
    pp = ctypes.cast(arr, L._PP)
    am = ensure_amax([dy] + list(srcs))
    a_am = (ctypes.c_void_p * len(srcs))(*am[1:])
    e0w = TIMER.begin() if TIMER else None
    L.check(_lib().fsraft_conv_wgrad(ctypes.c_void_p(dy.ptr), dy.ld, dy.C, pp, L.int_array([v.C for v in srcs]),
                                     L.int_array([v.ld for v in srcs]), len(srcs), L.ptr(dwpk), L.ptr(dbias), B, H, W,
                                     KH, KW, ctypes.c_void_p(am[0]), ctypes.cast(a_am, L._PP), L.stream()), "conv_wgrad")
    if TIMER:
        cin = sum(v.C for v in srcs)
        TIMER.end("conv_wgrad", e0w, 2.0 * B * H * W * dy.C * cin * KH * KW, 4.0 * B * H * W * (cin + dy.C),
                  tag=f"{B}x{H}x{W} {KH}x{KW} {'+'.join(str(v.C) for v in srcs)}->{dy.C} x1")


def conv_wgrad_multi(dys, srcs, dwpk, B, H, W, KH, KW, dbias=None):
    """One launch for several (dy, srcs) pairs of identical shape: dwpk += sum_t dy_t^T im2col(srcs_t).
    dys: list of V; srcs: list (same length) of lists of V."""
    n, nsrc = len(dys), len(srcs[0])
    dy0 = dys[0]
    a_dy = (ctypes.c_void_p * n)(*[v.ptr for v in dys])
    a_src = (ctypes.c_void_p * (n * nsrc))(*[v.ptr for sl in srcs for v in sl])
    am = ensure_amax(list(dys) + [v for sl in srcs for v in sl])
    m_dy = (ctypes.c_void_p * n)(*am[:n])
    m_src = (ctypes.c_void_p * (n * nsrc))(*am[n:])
    e0w = TIMER.begin() if TIMER else None
    L.check(_lib().fsraft_conv_wgrad_multi(ctypes.cast(a_dy, L._PP), n, dy0.ld, dy0.C, ctypes.cast(a_src, L._PP),
                                           L.int_array([v.C for v in srcs[0]]), L.int_array([v.ld for v in srcs[0]]), nsrc,
                                           L.ptr(dwpk), L.ptr(dbias), B, H, W, KH, KW, ctypes.cast(m_dy, L._PP),
                                           ctypes.cast(m_src, L._PP), L.stream()), "conv_wgrad_multi")
    if TIMER:
        cin = sum(v.C for v in srcs[0])
        TIMER.end("conv_wgrad", e0w, 2.0 * n * B * H * W * dy0.C * cin * KH * KW, 4.0 * n * B * H * W * (cin + dy0.C),
                  tag=f"{B}x{H}x{W} {KH}x{KW} {'+'.join(str(v.C) for v in srcs[0])}->{dy0.C} x{n}")


def stem_fwd(x, w, bias=None):
    """The encoders' 7x7 stride-2 stem (pytorch/core/extractor.py:135, :212) on csrc/stem.hip: x [B,3,H,W] contiguous fp32,
    w [N,3,7,7] (N = 32 or 64) -> [B,Ho,Wo,N] channels-last."""
    L.require_cuda_f32(x, w)
    B, C, H, W = x.shape
    N = w.shape[0]
    if C != 3 or tuple(w.shape[1:]) != (3, 7, 7) or not x.is_contiguous():
        raise RuntimeError("stem_fwd: x [B,3,H,W] contiguous and w [N,3,7,7] expected")
    out = torch.empty(B, (H - 1) // 2 + 1, (W - 1) // 2 + 1, N, device=x.device, dtype=torch.float32)
    t = TIMER
    e0 = t.begin() if t else None
    xa = amax_of(x)
    if xa is None:
        xa = x._fs_amax = amax_tensor(x)        # (kept on the image tensor: the weight gradient reads it again)
    L.check(_lib().fsraft_stem7x7s2_fwd(L.ptr(x), L.ptr(w.contiguous()), L.ptr(bias), L.ptr(out), B, H, W, N, L.ptr(xa), L.stream()), "stem7x7s2_fwd")
    if t:
        t.end("stem", e0, 2.0 * out.numel() * 147, 4.0 * (x.numel() + out.numel()))
    return out


def stem_wgrad(x, dy):
    """dW [N,3,7,7] of stem_fwd from dy [B,Ho,Wo,N] channels-last (contiguous)."""
    L.require_cuda_f32(x, dy)
    B, C, H, W = x.shape
    N = dy.shape[-1]
    if not (x.is_contiguous() and dy.is_contiguous()) or tuple(dy.shape[:3]) != (B, (H - 1) // 2 + 1, (W - 1) // 2 + 1):
        raise RuntimeError("stem_wgrad: contiguous x [B,3,H,W] and dy [B,Ho,Wo,N] expected")
    lib = _lib()
    scratch = torch.empty(lib.fsraft_stem_slots() * 64 * 192, device=x.device, dtype=torch.float32)
    dw = torch.empty(N, 3, 7, 7, device=x.device, dtype=torch.float32)
    t = TIMER
    e0 = t.begin() if t else None
    xa = amax_of(x) if amax_of(x) is not None else amax_tensor(x)
    da = amax_of(dy) if amax_of(dy) is not None else amax_tensor(dy)
    L.check(lib.fsraft_stem7x7s2_wgrad(L.ptr(x), L.ptr(dy), L.ptr(dw), L.ptr(scratch), B, H, W, N, L.ptr(xa), L.ptr(da), L.stream()), "stem7x7s2_wgrad")
    if t:
        t.end("stem", e0, 2.0 * dy.numel() * 147, 4.0 * (x.numel() + dy.numel()))
    return dw


def conv_small_fwd(x, w_oihw, bias, out_nchw):
    """3x3 convolution to 2 output channels as per-pixel dot products; x: V (channels-last), out: [B,2,H,W]."""
    B, N, H, W = out_nchw.shape
    t = TIMER
    e0 = t.begin() if t else None
    L.check(_lib().fsraft_conv_small_fwd(ctypes.c_void_p(x.ptr), x.ld, x.C, L.ptr(w_oihw), L.ptr(bias), L.ptr(out_nchw),
                                         N * H * W, H * W, 1, N, B, H, W, 3, 3, L.stream()), "conv_small_fwd")
    if t:
        t.end("conv_small", e0, 2.0 * B * H * W * N * x.C * 9, 4.0 * B * H * W * (x.C + N))


def conv_small_wgrad(dys, xs, dwpk, dbias, B, H, W):
    """dwpk (packed [2][9*ceil32(C)]) += sum over the (dy, x) pairs; dys/xs: lists of V."""
    n = len(dys)
    a_dy = (ctypes.c_void_p * n)(*[v.ptr for v in dys])
    a_x = (ctypes.c_void_p * n)(*[v.ptr for v in xs])
    L.check(_lib().fsraft_conv_small_wgrad(ctypes.cast(a_dy, L._PP), ctypes.cast(a_x, L._PP), n, dys[0].ld, xs[0].ld, xs[0].C,
                                           L.ptr(dwpk), L.ptr(dbias), dys[0].C, B, H, W, 3, 3, L.stream()), "conv_small_wgrad")


def conv_small_dgrad(dy, w_oihw, dst, mask, B, H, W):
    """Data gradient of conv_small_fwd: dy V (channels-last, two gradients per pixel), w_oihw [2,C,3,3], dst V (channels-last
    slice that receives dx, c < C), mask V or None (dx = 0 where mask <= 0: the ReLU in front of the convolution)."""
    C = dst.C
    t = TIMER
    e0 = t.begin() if t else None
    L.check(_lib().fsraft_conv_small_dgrad(ctypes.c_void_p(dy.ptr), dy.ld, L.ptr(w_oihw), ctypes.c_void_p(dst.ptr), dst.ld,
                                           ctypes.c_void_p(mask.ptr) if mask is not None else None, mask.ld if mask is not None else 0,
                                           C, 2, B, H, W, 3, 3, L.ptr(dst.amax), L.stream()), "conv_small_dgrad")
    if t:
        t.end("conv_small", e0, 2.0 * B * H * W * 2 * C * 9, 4.0 * B * H * W * (C * (2 if mask is not None else 1) + 2))


def col_sum_v(v, out, scale=1.0):
    M = v.t.numel() // v.ld
    L.check(_lib().fsraft_col_sum(ctypes.c_void_p(v.ptr), v.ld, M, v.C, L.ptr(out), float(scale), L.stream()), "col_sum")
