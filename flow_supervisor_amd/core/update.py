"""RAFT update blocks on hand-written HIP kernels (rows a6-a8 of SURVEY.md section 8).

Drop-in for pytorch/core/update.py of the reference: the public classes
``BasicUpdateBlock`` / ``SmallUpdateBlock`` (and the sub-module containers
``BasicMotionEncoder``, ``SmallMotionEncoder``, ``SepConvGRU``, ``ConvGRU``, ``FlowHead``)
keep the reference's constructor arguments, ``forward`` signatures, return values and
state_dict keys/shapes (update.py:6-136), so reference checkpoints load unchanged.

What differs is the execution: the sub-modules only own parameters.  A forward call runs
one ``autograd.Function`` whose forward and backward are explicit sequences of libfsraft
kernels (implicit-GEMM convolutions on fp32 MFMA with fused bias / ReLU / GRU-gate
epilogues, weight-gradient GEMMs, and a handful of elementwise kernels) over channels-last
buffers.  torch.cat never materialises: convolutions read up to three source tensors and
write into channel slices.  There is no eager/PyTorch fallback: without libfsraft.so, or
on CPU tensors, forward raises.
"""
import os
import weakref

import torch
import torch.nn as nn

from .. import _lib as L
from .. import ops
from ..ops import Dst, V
from . import streams


# --------------------------------------------------------------------------- containers
class FlowHead(nn.Module):
    """Parameter container, update.py:6-14 (conv1 3x3 in->hidden, conv2 3x3 hidden->2)."""

    def __init__(self, input_dim=128, hidden_dim=256):
        super().__init__()
        self.conv1 = nn.Conv2d(input_dim, hidden_dim, 3, padding=1)
        self.conv2 = nn.Conv2d(hidden_dim, 2, 3, padding=1)


class ConvGRU(nn.Module):
    """Parameter container, update.py:16-31 (3x3 gates)."""

    def __init__(self, hidden_dim=128, input_dim=192 + 128):
        super().__init__()
        for n in ("convz", "convr", "convq"):
            setattr(self, n, nn.Conv2d(hidden_dim + input_dim, hidden_dim, 3, padding=1))


class SepConvGRU(nn.Module):
    """Parameter container, update.py:33-60 ((1,5) pass then (5,1) pass)."""

    def __init__(self, hidden_dim=128, input_dim=192 + 128):
        super().__init__()
        for n in ("convz1", "convr1", "convq1"):
            setattr(self, n, nn.Conv2d(hidden_dim + input_dim, hidden_dim, (1, 5), padding=(0, 2)))
        for n in ("convz2", "convr2", "convq2"):
            setattr(self, n, nn.Conv2d(hidden_dim + input_dim, hidden_dim, (5, 1), padding=(2, 0)))


class SmallMotionEncoder(nn.Module):
    """Parameter container, update.py:62-77."""

    def __init__(self, args):
        super().__init__()
        cor_planes = args.corr_levels * (2 * args.corr_radius + 1) ** 2
        self.convc1 = nn.Conv2d(cor_planes, 96, 1, padding=0)
        self.convf1 = nn.Conv2d(2, 64, 7, padding=3)
        self.convf2 = nn.Conv2d(64, 32, 3, padding=1)
        self.conv = nn.Conv2d(128, 80, 3, padding=1)


class BasicMotionEncoder(nn.Module):
    """Parameter container, update.py:79-97."""

    def __init__(self, args):
        super().__init__()
        cor_planes = args.corr_levels * (2 * args.corr_radius + 1) ** 2
        self.convc1 = nn.Conv2d(cor_planes, 256, 1, padding=0)
        self.convc2 = nn.Conv2d(256, 192, 3, padding=1)
        self.convf1 = nn.Conv2d(2, 128, 7, padding=3)
        self.convf2 = nn.Conv2d(128, 64, 3, padding=1)
        self.conv = nn.Conv2d(64 + 192, 128 - 2, 3, padding=1)


def _pad4(c):
    return (c + 3) // 4 * 4


# measured: +2 % at best, and it makes per-kernel event timing meaningless (kernels of the two streams
# overlap), so weight gradients stay on the main stream unless asked for
_WGRAD_SIDE_STREAM = False
FLOW_BRANCH_STREAM = True     # the motion encoder's flow branch on a second stream beside its correlation branch (forward): +0.6 % config 3, +1.5 % at one pair
# one weight-gradient launch per layer per step (all iterations' operands stashed) instead of one per iteration
_DEFER_WGRAD = True


# --------------------------------------------------------------------------- layer table
class _Layer:
    """One GEMM of the block: which parameters form its weight, its taps and its inputs."""

    def __init__(self, key, wnames, kh, kw, src_c, view_as=None, bias=True, cin_sel=None):
        self.key = key              # short id
        self.cin_sel = cin_sel      # [(start, stop), ...] input-channel ranges of the OIHW weight this GEMM uses (None: all)
        self.bias = bias            # nn.Conv2d(bias=False) layers have no bias parameter
        self.wnames = wnames        # parameter prefixes concatenated along Cout (e.g. convz1+convr1)
        self.kh, self.kw = kh, kw
        self.src_c = src_c          # channel counts of the concatenated inputs
        self.view_as = view_as      # (Cin, kh, kw) to reinterpret the OIHW weight (7x7 conv as 1x1 over im2col)


class _Engine:
    """Owns the packed-weight cache and runs the kernel sequences of one update block."""

    def __init__(self, module, small, gma=False):
        self.m = module
        self.small = small
        self.gma = gma              # GMAUpdateBlock: Aggregate between the motion encoder and the GRU
        if small:
            self.corr_c, self.c1, self.c2, self.f1, self.f2, self.cv = module.cor_planes, 96, 0, 64, 32, 80
            self.hid, self.inp_c, self.head_c, self.has_mask = 96, 64, 128, False
            passes = [("", 3, 3)]
        else:
            self.corr_c, self.c1, self.c2, self.f1, self.f2, self.cv = module.cor_planes, 256, 192, 128, 64, 126
            self.hid, self.inp_c, self.head_c, self.has_mask = 128, 128, 256, True
            passes = [("1", 1, 5), ("2", 5, 1)]
        self.passes = passes
        self.mot_c = self.cv + 2                                   # motion features = [conv out | flow]
        cor_out = self.c2 if self.c2 else self.c1                  # channels the corr branch contributes to cor_flo
        self.cf_c = cor_out + self.f2
        hid, inp_c, mot = self.hid, self.inp_c, self.mot_c
        self.x_c = mot * (2 if gma else 1)                         # [motion | motion_global] share one buffer
        mot = self.x_c
        ls = [_Layer("c1", ["encoder.convc1"], 1, 1, [self.corr_c])]
        if self.c2:
            ls.append(_Layer("c2", ["encoder.convc2"], 3, 3, [self.c1]))
        ls += [
            _Layer("f1", ["encoder.convf1"], 1, 1, [98], view_as=(98, 1, 1)),
            _Layer("f2", ["encoder.convf2"], 3, 3, [self.f1]),
            _Layer("cv", ["encoder.conv"], 3, 3, [self.cf_c]),
        ]
        if gma:
            ls.append(_Layer("av", ["aggregator.to_v"], 1, 1, [self.mot_c], bias=False))
        # GRU gates: conv(cat(h, inp, motion)) = conv_hm(cat(h, motion)) + conv_i(inp) + b.  The context features inp do
        # not change over the iterations of a step, so conv_i(inp) + b is evaluated once per step ("zi*", "qi*") and
        # enters every iteration's epilogue as a per-pixel addend; its backward runs once on the summed gate gradients.
        hm = [(0, hid), (hid + inp_c, hid + inp_c + mot)]
        ii = [(hid, hid + inp_c)]
        for sfx, kh, kw in passes:
            zr, qq = ["gru.convz" + sfx, "gru.convr" + sfx], ["gru.convq" + sfx]
            ls.append(_Layer("zr" + sfx, zr, kh, kw, [hid, mot], bias=False, cin_sel=hm))
            ls.append(_Layer("q" + sfx, qq, kh, kw, [hid, mot], bias=False, cin_sel=hm))
            ls.append(_Layer("zi" + sfx, zr, kh, kw, [inp_c], cin_sel=ii))
            ls.append(_Layer("qi" + sfx, qq, kh, kw, [inp_c], cin_sel=ii))
        if self.has_mask:
            ls.append(_Layer("hd", ["flow_head.conv1", "mask.0"], 3, 3, [hid]))
            ls.append(_Layer("m2", ["mask.2"], 1, 1, [self.head_c]))
        else:
            ls.append(_Layer("hd", ["flow_head.conv1"], 3, 3, [hid]))
        ls.append(_Layer("fh2", ["flow_head.conv2"], 3, 3, [self.head_c]))
        self.layers = {l.key: l for l in ls}
        self.order = [l.key for l in ls]
        self.pnames = []                                            # flat parameter order fed to the Function
        self.ctx_keys = [k + sfx for sfx, _, _ in passes for k in ("zi", "qi")]
        for l in ls:
            for w in l.wnames:
                for n in [w + ".weight"] + ([w + ".bias"] if l.bias else []):
                    if n not in self.pnames:
                        self.pnames.append(n)
        self.extra = ["aggregator.gamma"] if gma else []            # non-conv parameters
        self.pnames += self.extra
        self._cache_key = None
        self._cache = None
        self._pstate = None

    # ---- parameter-gradient accumulation across the calls of one step -----------------
    def param_state(self, params):
        """(state, anchor) shared by every forward call made with the same parameter values.
        Weight/bias gradients of all those calls accumulate in ONE packed arena and are unpacked
        once, when autograd reaches the anchor's producer (_ParamFn) -- i.e. after the last of
        the 12 update-block backwards of a RAFT step -- instead of 26 tensors x 12 iterations."""
        if not (torch.is_grad_enabled() and any(p.requires_grad for p in params)):
            return None, None
        key = tuple((p.data_ptr(), p._version) for p in params)
        st = self._pstate
        if st is None or st.key != key or st.consumed:
            st = _ParamState(key)
            st.anchor = _ParamFn.apply(self, st, *params)
            self._pstate = st
        return st, st.anchor

    def _grad_arena(self, st, P, dev):
        if st.arena is None:
            sizes = [(k, P[k][0].numel(), P[k][3] if self.layers[k].bias else 0) for k in self.order]
            total = sum((a + 3) // 4 * 4 + (b + 3) // 4 * 4 for _, a, b in sizes) + 4 * len(self.extra)
            st.arena = torch.zeros(total, device=dev, dtype=torch.float32)
            st.dW, st.dB, o = {}, {}, 0
            for k, a, b in sizes:
                st.dW[k] = st.arena[o:o + a].view_as(P[k][0]); o += (a + 3) // 4 * 4
                st.dB[k] = st.arena[o:o + b] if b else None; o += (b + 3) // 4 * 4
            for n in self.extra:
                st.dB[n] = st.arena[o:o + 1]; o += 4
        return st.dW, st.dB

    def _side_stream(self, dev):
        s = self.__dict__.get("_side")
        if s is None or s.device != dev:
            s = torch.cuda.Stream(device=dev)
            self.__dict__["_side"] = s
        return s

    def unpack_param_grads(self, st, P, params):
        """packed arena -> list of gradients in self.pnames order (fused layers split back)."""
        if _WGRAD_SIDE_STREAM and st.arena is not None:
            torch.cuda.current_stream(st.arena.device).wait_stream(self._side_stream(st.arena.device))
        st.keep = None
        if st.pending:
            dW, dB = self._grad_arena(st, P, params[0].device)
            for (k, B, H, W), lst in st.pending.items():
                l = self.layers[k]
                if k == "fh2" and self.head_c % 4 == 0 and self.head_c <= 512:
                    ops.conv_small_wgrad([dy for dy, _ in lst], [srcs[0] for _, srcs in lst], dW[k], dB[k], B, H, W)
                    continue
                ops.conv_wgrad_multi([dy for dy, _ in lst], [srcs for _, srcs in lst], dW[k], B, H, W, l.kh, l.kw,
                                     dbias=dB[k])
            st.pending = {}
        if self.has_mask and st.dW is not None:
            st.dB["m2"].mul_(0.25)          # (the weight gradient's 0.25 is the scale of its unpack job)
        byname = dict(zip(self.pnames, params))
        # parameter-shaped gradients carved out of one buffer and filled by ONE batched un-pack launch: fused layers split
        # back, and the two GEMMs of a GRU convolution ((h, motion) part, context part) write their channel ranges of the same
        # tensor -- together they cover it, so nothing is zero-filled
        wn = [n for n in self.pnames if n.endswith(".weight")]
        flat = torch.empty(sum((byname[n].numel() + 3) // 4 * 4 for n in wn), device=params[0].device, dtype=torch.float32)
        grads, o = {}, 0
        for n in wn:
            grads[n] = flat[o:o + byname[n].numel()].view(byname[n].shape)
            o += (byname[n].numel() + 3) // 4 * 4
        items = []
        for k in self.order:
            l = self.layers[k]
            gs = [grads[w + ".weight"] for w in l.wnames]
            cin_full, kh, kw = l.view_as if l.view_as is not None else tuple(gs[0].shape[1:])
            off = [a for a, _ in l.cin_sel] if l.cin_sel is not None else [sum(l.src_c[:i]) for i in range(len(l.src_c))]
            items.append((st.dW[k], gs, l.src_c, off, cin_full, kh, kw, 0.25 if k == "m2" else 1.0, False))
            if l.bias:
                o = 0
                for wname in l.wnames:
                    n = byname[wname + ".weight"].shape[0]
                    grads[wname + ".bias"] = st.dB[k][o:o + n]
                    o += n
        ops.unpack_weight_grads(items, params[0].device)
        for n in self.extra:
            grads[n] = st.dB[n].reshape(byname[n].shape)
        st.arena = st.dW = st.dB = None
        return [grads[n] for n in self.pnames]

    # ---- parameters -------------------------------------------------------------
    def params(self):
        sd = dict(self.m.named_parameters())
        return [sd[n] for n in self.pnames]

    def _packed(self, params):
        """{key: (wpk_fwd, wpk_dgrad, bias, out_channels, oihw_shape)}; repacked when any parameter changed."""
        key = (ops.exact_mode(),) + tuple((p.data_ptr(), p._version) for p in params)      # (the mode decides which packs exist)
        if key == self._cache_key:
            return self._cache
        byname = dict(zip(self.pnames, params))
        out = {}
        exact = ops.exact_mode()
        with torch.no_grad():
            for n in self.extra:
                out[n] = byname[n].detach().float().reshape(-1)
            # the 2-output flow-head convolution runs on the dot-product kernels straight from the OIHW weight
            out["fh2.raw"] = (byname["flow_head.conv2.weight"].detach().contiguous().float(),
                              byname["flow_head.conv2.bias"].detach().contiguous().float())
            # every packed matrix of the block in one batched launch, read in place from the parameters (fused layers and the
            # (h, motion) / context split of the GRU weights are address arithmetic of fsraft_pack_conv_weights)
            plan = ops.PackPlan(params[0].device)
            handles = {}
            for k in self.order:
                l = self.layers[k]
                ws = [byname[w + ".weight"].detach().contiguous().float() for w in l.wnames]
                cin_full, kh, kw = l.view_as if l.view_as is not None else tuple(ws[0].shape[1:])
                off = [a for a, _ in l.cin_sel] if l.cin_sel is not None else None
                cout, cin = sum(w.shape[0] for w in ws), sum(l.src_c)
                kw_ = dict(srcOff=off, kh=kh, kw=kw, cin_full=cin_full)
                fws = plan.pack(ws, l.src_c, 10, **kw_)
                fw = plan.pack(ws, l.src_c, 0, **kw_) if (exact or cout <= 32) else fws
                dgs = plan.pack(ws, l.src_c, 11, **kw_)
                dg = plan.pack(ws, l.src_c, 1, **kw_) if (exact or cin <= 32) else dgs
                b = None
                if l.bias:
                    bs = [byname[w_ + ".bias"].detach().contiguous().float() for w_ in l.wnames]
                    b = bs[0] if len(bs) == 1 else plan.bias(bs)
                handles[k] = (fw, dg, b, cout, (cout, cin, kh, kw), fws, dgs)
            packed = plan.run()
            for k, (fw, dg, b, cout, shape, fws, dgs) in handles.items():
                out[k] = (packed[fw], packed[dg], packed[b] if isinstance(b, int) else b, cout, shape, packed[fws], packed[dgs])
        self._cache_key, self._cache = key, out
        return out

    # ---- forward ------------------------------------------------------------------
    def context(self, inp, params):
        """Once per step: {"zi*"/"qi*": conv_i(inp) + bias as [B,H,W,N]} for the GRU gate convolutions."""
        L.require_cuda_f32(inp)
        B, H, W, _ = inp.shape
        P = self._packed(params)
        out = {}
        vinp = V(inp, self.inp_c)           # (one view for the four convolutions: its amax word is worked out once)
        for k in self.ctx_keys:
            l = self.layers[k]
            buf = torch.empty(B, H, W, P[k][3], device=inp.device, dtype=torch.float32)
            ops.conv_forward([vinp], P[k][0], P[k][2], B, H, W, l.kh, l.kw, P[k][3], [Dst.nhwc(buf)],
                             wpk_split=P[k][5])
            out[k] = buf
        return out

    def context_backward(self, cst, P, st):
        """Backward of context(): weight / bias gradients of the inp part into the arena, returns dL/d inp."""
        inp = cst.inp
        B, H, W, _ = inp.shape
        dW, dB = self._grad_arena(st, P, inp.device)
        # the first data gradient overwrites, the others accumulate: only padding channels (if any) need a zero fill
        pad = _pad4(self.inp_c) != self.inp_c
        dinp = (ops.zeros if pad else (lambda *sh, device: torch.empty(*sh, device=device, dtype=torch.float32)))(
            B, H, W, _pad4(self.inp_c), device=inp.device)
        acc = pad
        vinp = V(inp, self.inp_c)
        for k in self.ctx_keys:
            g = cst.dsum.get(k)
            parts = cst.parts.get(k)
            if parts:           # deferred sums: the kept per-iteration gradients (first grad_samples samples of the batch) in one pass
                g = g if g is not None else ops.zeros((B,) + tuple(parts[0].shape[1:]), device=inp.device) if parts[0].shape[0] != B \
                    else torch.empty((B,) + tuple(parts[0].shape[1:]), device=inp.device, dtype=torch.float32)
                ops.sum_n_(parts, g, accumulate=cst.dsum.get(k) is not None)
            if g is None:
                continue
            l = self.layers[k]
            n = P[k][3]
            vg = V(g, n)
            ops.conv_wgrad(vg, [vinp], dW[k], B, H, W, l.kh, l.kw, dbias=dB[k])
            ops.conv_forward([vg], P[k][1], None, B, H, W, l.kh, l.kw, self.inp_c, [Dst.nhwc(dinp, 0, 0, acc)],
                             wpk_split=P[k][6])
            acc = True
        if not acc:
            dinp.zero_()
        cst.dsum = {}
        cst.parts = {}
        return dinp

    def forward(self, net, ctxb, corr, flow, params, save, attn=None, attn_t=None, need_mask=True, head_out=None, mslot=None, hlast_out=None):
        """net/corr: channels-last [B,H,W,C]; ctxb: context() of the context features; flow: [B,2,H,W] (any pixel stride).
        Returns (net_out [B,H,W,hid], mask [B,H,W,576] or None, delta [B,2,H,W]) and, if `save`,
        a dict of the intermediates backward needs.  need_mask=False (inference, every iteration but the last: the
        reference computes the mask each time and drops it, raft.py:134-139) skips the 256 -> 576 mask convolution."""
        L.require_cuda_f32(net, corr, flow)
        B, H, W, _ = net.shape
        dev = net.device
        P = self._packed(params)

        def buf(c, zero=False, track=True):
            # track: the buffer carries an amax word (ops.tracked) that every kernel writing it below raises -- the convolution
            # epilogues, im2col7, flow_to_nhwc -- so that the convolutions reading it find their scale without a pass of their own
            ld = _pad4(c)
            t = ops.zeros(B, H, W, ld, device=dev) if (zero or ld != c) else torch.empty(B, H, W, ld, device=dev, dtype=torch.float32)
            return ops.tracked(t) if track else t

        def conv(k, srcs, dsts, relu=False, alpha=1.0, **kw):
            l = self.layers[k]
            wpk, _, bias, n = P[k][:4]
            ops.conv_forward(srcs, wpk, bias, B, H, W, l.kh, l.kw, n, dsts, relu=relu, alpha=alpha,
                             wpk_split=P[k][5], **kw)

        hid = self.hid
        if mslot is not None:
            # slots of a MotionBatch: the motion encoder's activations of all iterations back to back, so that its backward can
            # run once per step over T x B x H x W pixels
            mb, t = mslot
            if corr.data_ptr() != mb.corr[t].data_ptr():
                mb.corr[t].copy_(corr)
                if ops.amax_of(mb.corr) is not None:          # (a copy does not raise the slot buffer's word: do it here)
                    ops.amax_jobs([(mb.corr[t].data_ptr(), 1, mb.corr[t].numel(), mb.corr[t].numel(), ops.amax_of(mb.corr))])
            corr = mb.corr[t]
            cor1 = mb.cor1[t] if self.c2 else None
            corflo, cols, flo1, motion = mb.corflo[t], mb.cols[t], mb.flo1[t], mb.motion[t]
        else:
            cor1 = buf(self.c1) if self.c2 else None
            corflo = buf(self.cf_c)
            cols = ops.tracked(torch.empty(B, H, W, _pad4(98), device=dev, dtype=torch.float32))      # (im2col7 writes the two pad columns itself)
            flo1 = buf(self.f1)
            motion = buf(self.x_c)           # GMA: channels [mot_c, 2 mot_c) hold motion_global
        cor_out = self.c2 if self.c2 else self.c1

        def flow_branch():
            ops.im2col7(flow, cols)
            conv("f1", [V(cols, 98)], [Dst.nhwc(flo1)], relu=True)
            conv("f2", [V(flo1, self.f1)], [Dst.nhwc(corflo, cor_out)], relu=True)

        # the motion encoder's two branches (update.py:83-87: convc1/convc2 on the correlation features, convf1/convf2 on the
        # flow) meet in `conv`: the flow branch on a second stream beside the correlation branch (core/streams.py; every buffer
        # was allocated above, on the caller's stream, and the two write disjoint channels of corflo)
        fb_side = FLOW_BRANCH_STREAM and streams.OVERLAP and dev.type == "cuda"
        if fb_side:
            with torch.cuda.stream(streams.fork(dev, 1)):
                flow_branch()
        if self.c2:
            conv("c1", [V(corr, self.corr_c)], [Dst.nhwc(cor1)], relu=True)
            conv("c2", [V(cor1, self.c1)], [Dst.nhwc(corflo)], relu=True)
        else:
            conv("c1", [V(corr, self.corr_c)], [Dst.nhwc(corflo)], relu=True)
        if fb_side:
            streams.join(dev, which=1)
        else:
            flow_branch()
        conv("cv", [V(corflo, self.cf_c)], [Dst.nhwc(motion)], relu=True)
        ops.flow_to_nhwc(flow, motion, self.cv)
        v = agg = None
        if self.gma:
            # Aggregate (gma.py:102-115, heads = 1): v = to_v(motion); motion_global = motion + gamma * (attn @ v)
            N, mc = H * W, self.mot_c
            if attn is None or attn.numel() != B * N * N:
                raise RuntimeError("GMA update block needs the attention map [B,1,H*W,H*W] (single head)")
            L.require_cuda_f32(attn)
            attn = attn.contiguous()
            v, agg = buf(mc), buf(mc, track=False)
            conv("av", [V(motion, mc, 0)], [Dst.nhwc(v)])
            if attn_t is not None:
                # attn @ v on the record GEMM core: the attention map was split to records once per pair (attn_t), v is
                # transposed ([mc][N], 3.6 MB) so that both operands are rows of records along the contraction index
                Nr = attn_t.shape[-1]
                vt = ops.to_records(ops.transpose_batched(v.view(B, N, mc)), amax=ops.amax_of(v))          # [B, mc, Nr]
                ops.gemm_rec_nt_raw(attn_t.data_ptr(), Nr, N * Nr, vt.data_ptr(), Nr, mc * Nr, agg.data_ptr(), mc, N * mc, B, N, mc, Nr,
                                    ksplit=2, a_amax=ops.amax_of(attn_t), b_amax=ops.amax_of(vt))
            else:
                ops.gemm_raw(attn.data_ptr(), N, N * N, v.data_ptr(), mc, N * mc, agg.data_ptr(), mc, N * mc, B, N, mc, N, False)
            ops.gma_mix_fwd(V(motion, mc, 0), V(agg), P["aggregator.gamma"], V(motion, mc, mc))

        h = net
        gates = []
        for pi, (sfx, _, _) in enumerate(self.passes):
            z, r, rh, q = buf(hid), buf(hid), buf(hid), buf(hid)
            # (the block's output state: into its HeadBatch slot when there is one -- the batched head backward reads all of them)
            hn = hlast_out if (hlast_out is not None and pi == len(self.passes) - 1) else buf(hid)
            xs = [V(motion, self.x_c)]
            conv("zr" + sfx, [V(h, hid)] + xs, [Dst.nhwc(z)], epi=2, h=h, aux1=rh, aux2=r, hid=hid, pre=ctxb["zi" + sfx])
            conv("q" + sfx, [V(rh, hid)] + xs, [Dst.nhwc(hn)], epi=3, h=h, z=z, aux1=q, pre=ctxb["qi" + sfx])
            gates.append((h, z, r, rh, q))
            h = hn
        nhead = self.head_c * (2 if self.has_mask else 1)
        # head_out: a slot of a HeadBatch -- the mask convolution (and the upsampler) of all iterations then run as one launch
        # after the loop, on the slots
        head = buf(nhead) if head_out is None else head_out
        conv("hd", [V(h, hid)], [Dst.nhwc(head)], relu=True)
        delta = torch.empty(B, 2, H, W, device=dev, dtype=torch.float32)
        if self.head_c % 4 == 0 and self.head_c <= 512:
            ops.conv_small_fwd(V(head, self.head_c), P["fh2.raw"][0], P["fh2.raw"][1], delta)
        else:
            conv("fh2", [V(head, self.head_c)], [Dst.nchw(delta)])
        mask = None
        if self.has_mask and (need_mask or save) and head_out is None:
            mask = buf(576)
            conv("m2", [V(head, self.head_c, self.head_c)], [Dst.nhwc(mask)], alpha=0.25)
        saved = None
        if save:
            saved = dict(B=B, H=H, W=W, corr=corr, cor1=cor1, corflo=corflo, cols=cols, flo1=flo1,
                         motion=motion, gates=gates, hlast=h, head=head, attn=attn, v=v, agg=agg, attn_r=attn_t)
        return h, mask, delta, saved

    # ---- backward -----------------------------------------------------------------
    def backward(self, S, P, st, dnet_out, dmask, ddelta, need_input_grads=True, ast=None, cst=None, need_dflow=True, motion_only=None, heads_only=False):
        """Accumulates parameter gradients into the packed arena of `st` and the gate gradients into `cst.dsum`
        (the context part's backward runs once per step, context_backward); returns (dnet, dcorr, dflow)."""
        B, H, W = S["B"], S["H"], S["W"]
        dev = S["corr"].device
        lib = L.load()
        # grad_samples (forward_cl): the caller vouches that only the first k samples of the batch receive gradient from its
        # loss (the flow-supervisor step batches a labelled and an unlabelled sample; the supervisor's predictions of the
        # unlabelled one carry none).  Every buffer is sample-major, so the whole backward simply runs on B = k: the kernels read
        # the first k samples of the saved activations and of the incoming gradients; what goes back to autograd (dnet, dcorr)
        # and the context part's running sums are full-size with zeros behind sample k.
        Bf = B
        gs = S.get("gs")
        if gs is not None and 0 < gs < Bf and (not self.gma or ast is None) and not need_dflow and motion_only is None and not heads_only:
            B = gs
            dnet_out = dnet_out[:B] if dnet_out is not None else None
            ddelta = ddelta[:B] if ddelta is not None else None
            dmask = dmask[:B] if dmask is not None else None
        M = B * H * W
        hid = self.hid

        def buf(c, zero=False, track=True, word=None):
            # (track: as in forward -- here the writers are the data-gradient epilogues, gru_bwd1 / gru_bwd2 and conv_small_dgrad;
            #  word: share another buffer's amax word)
            ld = _pad4(c)
            t = ops.zeros(B, H, W, ld, device=dev) if (zero or ld != c) else torch.empty(B, H, W, ld, device=dev, dtype=torch.float32)
            return ops.tracked(t, word) if track else t

        dW, dB = self._grad_arena(st, P, dev)
        if st.keep is None:
            st.keep = []

        def relu_bwd(g, y):
            L.check(lib.fsraft_relu_bwd(L.c_void_p(g.ptr), g.ld, L.c_void_p(y.ptr), y.ld, M, g.C, L.stream()), "relu_bwd")

        # Weight gradients run on a side stream, concurrently with the data-gradient chain on the
        # main stream: the two only share read-only inputs, and co-scheduling them fills the CUs a
        # single 220..880-workgroup GEMM leaves idle in its last wave.  Everything the side stream
        # reads is kept alive in st.keep until the arena is unpacked (which joins the streams).
        side = self._side_stream(dev) if _WGRAD_SIDE_STREAM else None
        main = torch.cuda.current_stream(dev)
        keep = st.keep if st.keep is not None else []

        def wgrad(k, dy, srcs):
            l = self.layers[k]
            if st.pending is not None:
                # deferred: dW = sum over the iterations of a step is one launch per layer, issued when the arena is
                # unpacked (the operands stay alive, and are not written again, until then)
                st.pending.setdefault((k, B, H, W), []).append((dy, srcs))
                return
            if side is None:
                ops.conv_wgrad(dy, srcs, dW[k], B, H, W, l.kh, l.kw, dbias=dB[k])
                return
            ev = torch.cuda.Event()
            ev.record(main)
            side.wait_event(ev)
            keep.append((dy.t, [v.t for v in srcs]))
            with torch.cuda.stream(side):
                ops.conv_wgrad(dy, srcs, dW[k], B, H, W, l.kh, l.kw, dbias=dB[k])

        def dgrad(k, dy, dsts, alpha=1.0):
            l = self.layers[k]
            n_in = sum(l.src_c)
            ops.conv_forward([dy], P[k][1], None, B, H, W, l.kh, l.kw, n_in, dsts, alpha=alpha, wpk_split=P[k][6])

        dh = dh_full = None
        mbs = S.get("mb")             # (MotionBatch, slot): the motion encoder's backward of all iterations runs once, after slot 0's
        if motion_only is not None:
            dmotion, motion = motion_only, S["motion"]
        if motion_only is None:
            # ---- heads
            hb = S.get("hb")
            dh_more = None
            if hb is not None and hb[0].heads_done:
                # the whole head backward of this iteration (flow head, mask head, the 3x3 convolution under both) already ran, for
                # all iterations at once, inside the batched backward of the mask head + upsampler (_MaskUpFn): it depends on the
                # loss alone, not on the recurrence.  What is left is to add the hidden-state gradient arriving from iteration t + 1.
                dh = hb[0].dh_heads[hb[1]][:B]
                if ddelta is not None and ddelta.data_ptr() != hb[0].dflow[hb[1]].data_ptr():
                    raise RuntimeError("HeadBatch: an iteration's delta_flow received a gradient from somewhere other than its prediction")
                dh_more = dnet_out.contiguous() if dnet_out is not None else None      # (added inside the first gate-gradient kernel)
            else:
                head = S["head"]
                if heads_only:
                    dhead = S["dhead"]
                elif hb is not None:
                    # the mask half of dhead was written by the batched backward of the mask head (HeadBatch / _MaskUpFn), which
                    # autograd runs before this node: this iteration's delta gradient comes out of it
                    dhead = hb[0].dhead_slot(hb[1])[:B]
                else:
                    dhead = buf(self.head_c * (2 if self.has_mask else 1))
                if self.has_mask and hb is None and not heads_only:
                    if dmask is not None:
                        # y = 0.25*(Wx+b)  =>  everything upstream sees 0.25*dmask: the data gradient applies the factor in its
                        # epilogue, the weight / bias gradients once per step when the arena is unpacked
                        g = dmask.contiguous()
                        wgrad("m2", V(g, 576), [V(head, self.head_c, self.head_c)])
                        dgrad("m2", V(g, 576), [Dst.nhwc(dhead, self.head_c).masked(V(head, self.head_c, self.head_c))], alpha=0.25)
                    else:
                        dhead[..., self.head_c:].zero_()
                dd = ops.zeros(B, H, W, 4, device=dev)
                if ddelta is not None:
                    ops.flow_to_nhwc(ddelta, dd, 0)
                    wgrad("fh2", V(dd, 2), [V(head, self.head_c)])
                    if FH2_DGRAD_SMALL and "fh2.raw" in P and self.head_c % 4 == 0 and self.head_c <= 256:
                        # 18 multiply-adds per element: a streaming kernel, not an implicit GEMM with K padded from 18 to 288
                        ops.conv_small_dgrad(V(dd, 2), P["fh2.raw"][0], V(dhead, self.head_c, 0), V(head, self.head_c, 0), B, H, W)
                    else:
                        dgrad("fh2", V(dd, 2), [Dst.nhwc(dhead, 0).masked(V(head, self.head_c, 0))])
                else:
                    dhead[..., : self.head_c].zero_()
                # (no separate ReLU-backward pass: the two data-gradient epilogues above wrote dhead already masked by head > 0)
                hlast = S["hlast"]
                wgrad("hd", V(dhead), [V(hlast, hid)])
                dh = S["dh_out"] if heads_only else buf(hid, track=False)       # (filled by a copy below; read by gru_bwd1 only)
                if dnet_out is not None:
                    dh.copy_(dnet_out)
                    dgrad("hd", V(dhead), [Dst.nhwc(dh, 0, 0, True)])
                else:
                    dgrad("hd", V(dhead), [Dst.nhwc(dh)])
                if heads_only:
                    return None, None, None

            # ---- GRU passes, last to first
            motion = S["motion"]
            # every GRU data gradient adds its motion part; the first one (the q convolution of the last pass covers all x_c
            # channels) overwrites instead, so the buffer needs no zero fill -- only its padding channels, if any, do
            dmotion = mbs[0].dmotion[mbs[1]][:B] if mbs is not None else buf(self.x_c)
            dm_first = [True]

            def dm_acc():
                first, dm_first[0] = dm_first[0], False
                return not first

            def ctx_sum(k, like):
                """Running sum of the gate gradients over the iterations of the step (filled by the gru_bwd kernels) -- or, deferred:
                the iteration's buffer is only noted and context_backward adds the kept buffers up in one pass."""
                if cst is None:
                    return None
                if CTX_SUM_DEFERRED:
                    cst.parts.setdefault(k, []).append(like)
                    return None
                if k not in cst.dsum:
                    cst.dsum[k] = ops.zeros((Bf,) + tuple(like.shape[1:]), device=like.device)
                return cst.dsum[k]

            first_pass = self.passes[0][0]
            for (sfx, _, _), (h, z, r, rh, q) in reversed(list(zip(self.passes, S["gates"]))):
                dzr = buf(2 * hid)
                dq = buf(hid, word=ops.amax_of(dzr))       # (one word for the two gate gradients: gru_bwd1 then raises it once)
                if B != Bf and sfx == first_pass:          # the buffer that goes back to autograd as dnet: full size, zeros behind sample B
                    dh_full = ops.zeros(Bf, H, W, _pad4(hid), device=dev)
                    dhp = dh_full[:B]
                else:
                    dhp = buf(hid, track=False)            # (never read by a convolution)
                zsum, qsum = ctx_sum("zi" + sfx, dzr), ctx_sum("qi" + sfx, dq)
                ops.gru_bwd1(dh, z, q, h, dzr, dq, dhp, hid, zsum, qsum, dhn2=dh_more)
                dh_more = None
                xs = [V(motion, self.x_c)]
                wgrad("q" + sfx, V(dq, hid), [V(rh, hid)] + xs)
                drh = buf(hid)
                dgrad("q" + sfx, V(dq, hid), [Dst.nhwc(drh, 0, 0), Dst.nhwc(dmotion, 0, hid, dm_acc())])
                ops.gru_bwd2(drh, r, h, dzr, dhp, hid, zsum)
                wgrad("zr" + sfx, V(dzr, 2 * hid), [V(h, hid)] + xs)
                dm = Dst.nhwc(dmotion, 0, hid, True)
                if sfx == first_pass and not self.gma:
                    # last accumulation into dmotion: its epilogue applies the motion encoder's ReLU backward to the conv
                    # channels [0, cv) (the two flow channels behind them pass through)
                    dm = dm.masked(V(motion, self.cv))
                dgrad("zr" + sfx, V(dzr, 2 * hid), [Dst.nhwc(dhp, 0, 0, True), dm])
                dh = dhp

            # ---- Aggregate (GMA): motion_global = motion + gamma * (attn @ to_v(motion))
            if self.gma:
                N, mc = H * W, self.mot_c
                attn, v, agg = S["attn"], S["v"], S["agg"]
                dagg, dv = buf(mc), buf(mc, track=False)        # (dagg: raised by gma_mix_bwd; dv comes out of a GEMM that raises nothing)
                ops.gma_mix_bwd(V(dmotion, mc, mc), V(agg), P["aggregator.gamma"], V(dmotion, mc, 0), V(dagg),
                                dB["aggregator.gamma"])
                attn_r = S.get("attn_r")
                if attn_r is not None:   # dv = attn^T dagg: both operands k-major records -> transposed-read record GEMM
                    Nr = attn_r.shape[-1]
                    dr = ops.to_records(dagg.view(B, N, mc), amax=ops.amax_of(dagg))
                    ops.gemm_rec_tn_raw(attn_r.data_ptr(), Nr, N * Nr, dr.data_ptr(), dr.shape[-1], N * dr.shape[-1], dv.data_ptr(), mc,
                                        N * mc, B, N, mc, N, ksplit=2, a_amax=ops.amax_of(attn_r), b_amax=ops.amax_of(dr))
                elif N % 4 == 0:     # (exact-fp32 test mode) both operands k-major -> transposed-read split GEMM
                    ops.gemm_tn_raw(attn.data_ptr(), N, N * N, dagg.data_ptr(), mc, N * mc, dv.data_ptr(), mc, N * mc, B, N, mc, N)
                else:
                    at = attn.view(B, N, N).transpose(1, 2).contiguous()
                    ops.gemm_raw(at.data_ptr(), N, N * N, dagg.data_ptr(), mc, N * mc, dv.data_ptr(), mc, N * mc, B, N, mc, N, False)
                if ast is not None:  # dattn = sum_t dagg_t v_t^T is formed once per step from the stashed factors
                    ast.stash.append((dagg.view(B, N, mc), v.view(B, N, mc)))
                wgrad("av", V(dv, mc), [V(motion, mc, 0)])
                dgrad("av", V(dv, mc), [Dst.nhwc(dmotion, 0, 0, True)])

        if mbs is not None:
            # deferred: this iteration's dmotion sits in its slot; the motion encoder's backward (four data gradients, five weight
            # gradients) runs for all slots at once when slot 0 -- the first iteration, the last to run backward -- has handed in
            # its own.  The correlation gradient returned here is the slot's view of the batch's output, filled by then.
            mb, t = mbs
            mb.parked.add(t)
            if t == 0:
                mb.run(self, P, st)
            return (dh_full if B != Bf else dh), mb.dcorr[t], None

        # ---- motion encoder
        # (the loops detach the flow that enters an iteration, raft.py:123: its gradient -- the pass-through channels of the
        #  motion features plus the 7x7 convolution's data gradient -- is then three launches per iteration nobody reads)
        dflow = torch.empty(B, 2, H, W, device=dev, dtype=torch.float32) if need_dflow else None
        if need_dflow:
            ops.nhwc_to_flow(dmotion, self.cv, dflow, False)
        if self.gma:      # (GMA adds to dmotion after the GRU, so the mask cannot ride on a GRU epilogue)
            relu_bwd(V(dmotion, self.cv), V(motion, self.cv))
        corflo = S["corflo"]
        wgrad("cv", V(dmotion, self.cv), [V(corflo, self.cf_c)])
        dcorflo = buf(self.cf_c)
        dgrad("cv", V(dmotion, self.cv), [Dst.nhwc(dcorflo).masked(V(corflo, self.cf_c))])
        cor_out = self.c2 if self.c2 else self.c1
        flo1 = S["flo1"]
        wgrad("f2", V(dcorflo, self.f2, cor_out), [V(flo1, self.f1)])
        dflo1 = buf(self.f1)
        dgrad("f2", V(dcorflo, self.f2, cor_out), [Dst.nhwc(dflo1).masked(V(flo1, self.f1))])
        cols = S["cols"]
        wgrad("f1", V(dflo1, self.f1), [V(cols, 98)])
        if need_dflow:
            dcols = torch.empty(B, H, W, _pad4(98), device=dev, dtype=torch.float32)     # (col2im7 reads the 98 channels only)
            dgrad("f1", V(dflo1, self.f1), [Dst.nhwc(dcols)])
            ops.col2im7(dcols, dflow, True)
        corr = S["corr"]
        dcorr = dcorr_full = None
        if need_input_grads:
            if motion_only is not None:
                dcorr_full = S["dcorr_out"]
            else:
                dcorr_full = ops.zeros(Bf, H, W, _pad4(self.corr_c), device=dev) if B != Bf else buf(self.corr_c)
            dcorr = dcorr_full[:B]
        if self.c2:
            cor1 = S["cor1"]
            wgrad("c2", V(dcorflo, self.c2, 0), [V(cor1, self.c1)])
            dcor1 = buf(self.c1)
            dgrad("c2", V(dcorflo, self.c2, 0), [Dst.nhwc(dcor1).masked(V(cor1, self.c1))])
            wgrad("c1", V(dcor1, self.c1), [V(corr, self.corr_c)])
            if need_input_grads:
                dgrad("c1", V(dcor1, self.c1), [Dst.nhwc(dcorr)])
        else:
            wgrad("c1", V(dcorflo, self.c1, 0), [V(corr, self.corr_c)])
            if need_input_grads:
                dgrad("c1", V(dcorflo, self.c1, 0), [Dst.nhwc(dcorr)])

        return (dh_full if B != Bf else dh), dcorr_full, dflow


class _CtxState:
    """Per-step state of the context convolution: its outputs (forward) and the summed gate gradients (backward)."""
    __slots__ = ("key", "inp", "bufs", "dsum", "parts", "anchor", "consumed", "zero", "keep")

    def __init__(self, key, inp, bufs):
        self.key, self.inp, self.bufs, self.dsum, self.anchor, self.consumed, self.zero = key, inp, bufs, {}, None, False, None
        self.keep = None        # the tracked input tensor itself, for as long as the anchor stands for it (_forward_nchw reuses it by identity)
        self.parts = {}         # CTX_SUM_DEFERRED: per key the iterations' gate-gradient buffers, summed once in context_backward


class _CtxFn(torch.autograd.Function):
    """(param anchor, inp) -> 1-element anchor standing for conv_i(inp) + b of all GRU gate convolutions."""

    @staticmethod
    def forward(ctx, engine, st, cst, params, anchor, inp):
        ctx.engine, ctx.st, ctx.cst = engine, st, cst
        ctx.P = engine._packed(params)
        ctx.set_materialize_grads(False)
        return ops.zeros(1, device=inp.device)

    @staticmethod
    def backward(ctx, g):
        cst = ctx.cst
        cst.consumed = True
        dinp = ctx.engine.context_backward(cst, ctx.P, ctx.st)
        cst.bufs = None
        # (state -> anchor -> grad_fn -> ctx -> state is a reference cycle: cut it here, or the context features and whatever else
        #  the state holds wait for the cyclic collector -- tens of MB per step that only show up as a creeping peak)
        cst.anchor = cst.keep = None
        ctx.cst = ctx.st = ctx.P = ctx.engine = None
        return None, None, None, None, None, dinp


class _ParamState:
    __slots__ = ("key", "anchor", "arena", "dW", "dB", "consumed", "zero", "keep", "pending")

    def __init__(self, key):
        self.key, self.anchor, self.arena, self.dW, self.dB, self.consumed, self.zero = key, None, None, None, None, False, None
        self.keep = None
        self.pending = {} if (key is not None and _DEFER_WGRAD) else None     # frozen-parameter states compute nothing


class _ParamFn(torch.autograd.Function):
    """params -> 1-element anchor.  Its backward delivers the accumulated parameter gradients."""

    @staticmethod
    def forward(ctx, engine, st, *params):
        ctx.engine, ctx.st, ctx.params = engine, st, params
        ctx.P = engine._packed(params)
        ctx.set_materialize_grads(False)
        return ops.zeros(1, device=params[0].device)

    @staticmethod
    def backward(ctx, g):
        st = ctx.st
        st.consumed = True
        if st.arena is None:
            out = (None, None) + tuple(torch.zeros_like(p) for p in ctx.params)
        else:
            out = (None, None) + tuple(ctx.engine.unpack_param_grads(st, ctx.P, ctx.params))
        st.anchor = None                     # (cut the state -> anchor -> grad_fn -> ctx -> state cycle)
        ctx.st = ctx.P = ctx.params = ctx.engine = None
        return out


class _AttnState:
    """Per-step accumulator of dL/d attention (GMA).  dattn = sum_t dagg_t v_t^T over the iterations of a step is
    ONE GEMM with K = T*128 over the stashed (dagg_t, v_t) factors (14 MB each at 4 x 55 x 128) instead of T
    read-modify-write passes over the 0.8 GB map."""
    __slots__ = ("key", "anchor", "stash", "consumed", "zero")

    def __init__(self, key):
        self.key, self.anchor, self.stash, self.consumed, self.zero = key, None, [], False, None


class _AttnFn(torch.autograd.Function):
    """attention -> 1-element anchor; backward hands the accumulated gradient to the attention's producer."""

    @staticmethod
    def forward(ctx, ast, attn):
        ctx.ast, ctx.shape = ast, attn.shape
        ctx.set_materialize_grads(False)
        return ops.zeros(1, device=attn.device)

    @staticmethod
    def backward(ctx, g):
        ast = ctx.ast
        ast.consumed = True
        stash, ast.stash = ast.stash, []
        ast.anchor = None                    # (cut the state -> anchor -> grad_fn -> ctx -> state cycle)
        ctx.ast = None
        if not stash:
            return None, None
        D = torch.cat([d for d, _ in stash], 2) if len(stash) > 1 else stash[0][0].contiguous()
        Vc = torch.cat([v for _, v in stash], 2) if len(stash) > 1 else stash[0][1].contiguous()
        B, N, K = D.shape
        dattn = torch.empty(ctx.shape, device=D.device, dtype=torch.float32)
        dattn._fs_owned = True           # (gma._AttentionFn.backward may turn it into dS in place: nobody else holds it)
        if ops.SPLIT_VOLUME_BWD:         # record GEMM core (K = T * 128 is a multiple of 32)
            Dr, Vr = ops.to_records(D), ops.to_records(Vc)
            ops.gemm_rec_nt_raw(Dr.data_ptr(), Dr.shape[-1], N * Dr.shape[-1], Vr.data_ptr(), Vr.shape[-1], N * Vr.shape[-1],
                                dattn.data_ptr(), N, N * N, B, N, N, Dr.shape[-1], a_amax=ops.amax_of(Dr), b_amax=ops.amax_of(Vr))
        else:
            ops.gemm_raw(D.data_ptr(), K, N * K, Vc.data_ptr(), K, N * K, dattn.data_ptr(), N, N * N, B, N, N, K, True)
        return None, dattn


class _UpdateFn(torch.autograd.Function):
    """(anchor; net, inp, corr: channels-last; flow: NCHW) -> (net', mask channels-last or empty, delta NCHW)."""

    @staticmethod
    def forward(ctx, engine, st, params, anchor, net, cst, canchor, corr, flow, ast=None, attn=None, aanchor=None, attn_t=None, hb=None,
                grad_samples=None, mb=None):
        need = any(ctx.needs_input_grad)      # (grad mode is off inside Function.forward; this is the reliable signal)
        slot = hb.next_slot() if hb is not None else None
        if mb is not None and (not need or ctx.needs_input_grad[8]):
            # (the caller's lookup already wrote into the batch's next slot: dropping the batch here would leave the saved
            #  correlation features in a slot the next iteration overwrites)
            raise RuntimeError("motion_batch needs a recorded call whose flow input carries no gradient")
        mt = mb.next_slot() if mb is not None else None
        h, mask, delta, saved = engine.forward(net, cst.bufs, corr, flow, params, save=need, attn=attn, attn_t=attn_t,
                                               head_out=None if slot is None else hb.head[slot],
                                               mslot=None if mt is None else (mb, mt),
                                               hlast_out=None if (slot is None or hb.hlast is None) else hb.hlast[slot])
        if saved is not None and mt is not None:
            saved["mb"] = (mb, mt)
        if saved is not None and slot is not None:
            saved["hb"] = (hb, slot)
        if saved is not None and grad_samples is not None:
            saved["gs"] = int(grad_samples)
        ctx.engine, ctx.st, ctx.saved = engine, st, saved
        ctx.ast, ctx.cst = ast, cst
        ctx.P = engine._packed(params) if need else None
        ctx.has_mask = mask is not None
        if mask is None:
            mask = torch.empty(0, device=net.device)
            ctx.mark_non_differentiable(mask)
        return h, mask, delta

    @staticmethod
    def backward(ctx, dh, dmask, ddelta):
        eng = ctx.engine
        S, ctx.saved = ctx.saved, None
        if S is None:
            raise RuntimeError("update block backward ran twice on the same graph (retain_graph is not supported)")
        dmask = dmask if ctx.has_mask else None
        dh = dh.contiguous() if dh is not None else None
        cst = ctx.cst
        dnet, dcorr, dflow = eng.backward(S, ctx.P, ctx.st, dh, dmask, ddelta, need_input_grads=ctx.needs_input_grad[7], ast=ctx.ast,
                                          cst=cst if cst.anchor is not None else None, need_dflow=ctx.needs_input_grad[8])
        # the three anchors get no gradient tensor: autograd still runs their producers (_ParamFn, _CtxFn, _AttnFn) once
        # every consumer is done -- that ordering is all they are for -- and skips 3 x 12 one-element accumulation kernels
        return (None, None, None, None, dnet, None, None, dcorr, dflow, None, None, None, None, None, None, None)


# (module attributes, not environment switches: the tests that compare the per-step batches with the per-iteration route set them)
FH2_DGRAD_SMALL = True    # data gradient of the flow head's 256 -> 2 convolution on csrc/conv_small.hip (False: implicit GEMM)
HEAD_BATCH = True
MOTION_BATCH = True
CTX_SUM_DEFERRED = True   # gate-gradient sums of a step in one pass (False: running sums in gru_bwd1/2)
HEADS_BWD_BATCH = True


class HeadBatch:
    """The mask head and the convex upsampler of ALL iterations of a step as one launch each (and their backward likewise).
    Neither feeds the recurrence -- an iteration hands on the hidden state and the flow; the mask only shapes that iteration's
    full-resolution prediction (raft.py:134-139) -- so the loop writes each iteration's head activations into a slot of one
    buffer and `finish(flows)` runs the 1x1 mask convolution over T x B x H x W pixels and the upsampler over T x B images:
    4 x T launches per step become 4, and at one or two pairs per GPU they fill the chip instead of a fifth of it.
    Same values as T separate calls (the kernels are per-pixel / per-image); gradients reach the iterations through the
    flows (delta_t) and through `dhead_slot` (the mask half of each iteration's head gradient)."""

    def __init__(self, eng, params, st, anchor, T, B, H, W, device):
        self.eng, self.params, self.st, self.anchor = eng, params, st, anchor
        self.T, self.B, self.H, self.W = T, B, H, W
        self.head = ops.tracked(torch.empty(T, B, H, W, 2 * eng.head_c, device=device, dtype=torch.float32))
        self.hlast = ops.tracked(torch.empty(T, B, H, W, _pad4(eng.hid), device=device, dtype=torch.float32)) if _pad4(eng.hid) == eng.hid else None
        self.dhead = None
        self.dh_heads = self.dflow = None
        self.heads_done = False
        self.n = 0

    @staticmethod
    def fits(T, B, H, W):
        return T * B * H * W * 576 * 4 < 0x7fffffff          # (buffer-addressed kernels: 32-bit byte offsets)

    def next_slot(self):
        if self.n >= self.T:
            raise RuntimeError("HeadBatch: more update-block calls than slots")
        self.n += 1
        return self.n - 1

    def dhead_slot(self, t):
        if self.dhead is None:      # (no prediction of this batch reached the loss: the mask head gets no gradient)
            self.dhead = torch.zeros_like(self.head)
        return self.dhead[t]

    def finish(self, flows):
        """flows: the T flow fields [B,2,H,W] after each iteration -> the T upsampled predictions [B,2,8H,8W]."""
        if len(flows) != self.n:
            raise RuntimeError(f"HeadBatch: {self.n} update-block calls but {len(flows)} flows")
        return list(_MaskUpFn.apply(self, self.anchor, *flows))


class MotionBatch:
    """The motion encoder's backward of ALL iterations of a step as one launch per layer.  In backward only the GRU chain is
    sequential: the gradient of an iteration's motion features leaves it (towards the correlation lookup, whose own backward
    is deferred to the volume's build node anyway, and into weight gradients, which are deferred to the end of the step), so
    nothing waits for it.  The forward writes each iteration's motion-encoder activations (and the lookup writes its output)
    into slots of [T,B,H,W,C] buffers; an iteration's backward parks its dmotion in a slot and returns its slot of `dcorr`;
    when slot 0 -- the first iteration, the last to run backward -- has parked its own, `run` executes the four data
    gradients and queues the five weight gradients over T x B x H x W pixels.  4 x T launches become 4, on grids that fill the
    chip at one pair per GPU (per layer -10..-35 % at four pairs, -50..-75 % at one or two: round 3, docs/history)."""

    def __init__(self, eng, T, B, H, W, device, zero=False):
        def e(c, track=True):
            # (one amax word per slot buffer, raised by every kernel that writes a slot: the step's largest magnitude)
            t = torch.empty(T, B, H, W, _pad4(c), device=device, dtype=torch.float32)
            return ops.tracked(t) if track else t
        self.T, self.B, self.H, self.W, self.n = T, B, H, W, 0
        self.corr = e(eng.corr_c)
        self.cor1 = e(eng.c1) if eng.c2 else None
        self.corflo, self.cols, self.flo1, self.motion = e(eng.cf_c), e(98), e(eng.f1), e(eng.x_c)
        # zero: the iterations run their backward on the first k samples only (grad_samples) and park dmotion for those; the
        # batch then multiplies zeros for the others
        self.dmotion = ops.tracked(torch.zeros(T, B, H, W, _pad4(eng.x_c), device=device, dtype=torch.float32)) if zero else e(eng.x_c)
        self.dcorr = e(eng.corr_c)          # (tracked: the gradient volume takes its bound from this word)
        self.parked = set()             # slots whose iteration has handed in its dmotion

    @staticmethod
    def fits(eng, T, B, H, W):
        cs = (eng.corr_c, eng.c1, eng.cf_c, eng.f1, eng.x_c)
        return all(c % 4 == 0 for c in cs) and T * B * H * W * max(cs) * 4 < 0x7fffffff      # no pad channels; 32-bit byte offsets

    def next_slot(self):
        if self.n >= self.T:
            raise RuntimeError("MotionBatch: more update-block calls than slots")
        self.n += 1
        return self.n - 1

    def run(self, eng, P, st):
        n, B, H, W = self.n, self.B, self.H, self.W
        for t in range(n):              # an iteration whose backward never ran (a loss on a subset of the predictions and no
            if t not in self.parked:    # HeadBatch) handed in nothing: its slot is uninitialised memory, its gradient is zero
                self.dmotion[t].zero_()
        self.parked = set()

        def v(t):
            return None if t is None else t[:n].view(n * B, H, W, t.shape[-1])
        S = dict(B=n * B, H=H, W=W, corr=v(self.corr), cor1=v(self.cor1), corflo=v(self.corflo), cols=v(self.cols), flo1=v(self.flo1),
                 motion=v(self.motion), dcorr_out=v(self.dcorr))
        eng.backward(S, P, st, None, None, None, need_input_grads=True, need_dflow=False, motion_only=v(self.dmotion))


def _one_tensor(ts):
    """The tensors of `ts` as ONE tensor [len(ts) * n0, ...] without a copy when they are equal-shaped contiguous pieces lying
    back to back in one storage (the loss kernel hands out its gradients that way); None otherwise."""
    g0 = ts[0]
    if g0 is None or not g0.is_contiguous():
        return None
    nb = g0.numel() * g0.element_size()
    base = g0.untyped_storage().data_ptr()
    for i, g in enumerate(ts):
        if (g is None or g.shape != g0.shape or g.dtype != g0.dtype or not g.is_contiguous() or g.untyped_storage().data_ptr() != base
                or g.data_ptr() != g0.data_ptr() + i * nb):
            return None
    shape = (len(ts) * g0.shape[0],) + tuple(g0.shape[1:])
    st, acc = [], 1
    for d in reversed(shape):
        st.append(acc)
        acc *= d
    return torch.as_strided(g0, shape, tuple(reversed(st)))


class _MaskUpFn(torch.autograd.Function):
    """(HeadBatch, parameter anchor, flow_0 .. flow_{T-1}) -> T upsampled predictions: mask = 0.25 * conv1x1(head_mask) over
    all slots, then upsample_flow (raft.py:72-83) over T x B images."""

    @staticmethod
    def forward(ctx, hb, anchor, *flows):
        eng = hb.eng
        T, B, H, W = hb.n, hb.B, hb.H, hb.W
        P = eng._packed(hb.params)
        head = hb.head[:T].view(T * B, H, W, 2 * eng.head_c)
        fl = torch.stack([f.float() for f in flows]).view(T * B, 2, H, W)
        mask = torch.empty(T * B, H, W, 576, device=head.device, dtype=torch.float32)
        l = eng.layers["m2"]
        ops.conv_forward([V(head, eng.head_c, eng.head_c)], P["m2"][0], P["m2"][2], T * B, H, W, l.kh, l.kw, 576, [Dst.nhwc(mask)],
                         alpha=0.25, wpk_split=P["m2"][5])
        up = ops.upsample_fwd(fl, mask)
        ctx.hb, ctx.P = hb, P
        ctx.save_for_backward(fl, mask)
        ctx.set_materialize_grads(False)
        return tuple(up[t * B:(t + 1) * B] for t in range(T))

    @staticmethod
    def backward(ctx, *gs):
        hb, P = ctx.hb, ctx.P
        eng = hb.eng
        fl, mask = ctx.saved_tensors
        T, B, H, W = hb.n, hb.B, hb.H, hb.W
        if all(g is None for g in gs):
            return (None, None) + (None,) * T
        dup = _one_tensor(gs)
        if dup is None:
            z = next(g for g in gs if g is not None)
            dup = torch.cat([g if g is not None else torch.zeros_like(z) for g in gs])
        dflow, dmask = ops.upsample_bwd(fl, mask, dup)
        ctx.hb = None
        head = hb.head[:T].view(T * B, H, W, 2 * eng.head_c)
        hb.dhead = ops.tracked(torch.empty_like(hb.head))      # (writers: the mask head's data gradient, conv_small_dgrad)
        dhead = hb.dhead[:T].view(T * B, H, W, 2 * eng.head_c)
        hc = eng.head_c
        st = hb.st
        # y = 0.25 * (W x + b): the data gradient applies the factor in its epilogue, the weight / bias gradients once per step
        # when the arena is unpacked (unpack_param_grads)
        gv, xv = V(dmask, 576), V(head, hc, hc)
        if st is not None and st.key is not None:
            dW, dB = eng._grad_arena(st, P, head.device)
            if st.pending is not None:
                st.pending.setdefault(("m2", T * B, H, W), []).append((gv, [xv]))
            else:
                ops.conv_wgrad(gv, [xv], dW["m2"], T * B, H, W, 1, 1, dbias=dB["m2"])
        ops.conv_forward([gv], P["m2"][1], None, T * B, H, W, 1, 1, hc, [Dst.nhwc(dhead, hc).masked(xv)], alpha=0.25,
                         wpk_split=P["m2"][6])
        dfl = dflow.view(T, B, 2, H, W)
        if HEADS_BWD_BATCH and hb.hlast is not None and st is not None and st.key is not None:
            # the rest of the heads' backward depends on nothing but what this node just produced: the flow head (from dflow), then
            # the 3x3 convolution under both heads -- for all iterations at once; each iteration's backward starts from its slot
            # of dh_heads
            hb.dflow = dfl
            hb.dh_heads = torch.empty_like(hb.hlast)
            Sb = dict(B=T * B, H=H, W=W, corr=head, head=head, hlast=hb.hlast[:T].view(T * B, H, W, hb.hlast.shape[-1]), dhead=dhead,
                      dh_out=hb.dh_heads[:T].view(T * B, H, W, hb.hlast.shape[-1]))
            eng.backward(Sb, P, st, None, None, dflow, need_input_grads=False, need_dflow=False, heads_only=True)
            hb.heads_done = True
        return (None, None) + tuple(dfl[t] for t in range(T))


class _ToCL(torch.autograd.Function):
    """NCHW-contiguous -> channels-last buffer [B,H,W,C] via the fsraft transpose kernel."""

    @staticmethod
    def forward(ctx, x):
        return ops.nchw_to_nhwc(x)

    @staticmethod
    def backward(ctx, g):
        return ops.nhwc_to_nchw(g.contiguous())


class _FromCL(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return ops.nhwc_to_nchw(x)

    @staticmethod
    def backward(ctx, g):
        return ops.nchw_to_nhwc(g)


TWINS = True      # False: every NCHW tensor crossing the reference-shaped API is converted again by its consumer (rounds 1-5)
NCHW_VIEWS = True # `net`, `up_mask` and `corr` leave the reference-shaped entry points as NCHW-SHAPED VIEWS of the channels-last tensors (as_nchw)


def to_channels_last(x):
    """[B,C,H,W] -> [B,H,W,C] (C must be a multiple of 4 for the GEMM kernels).
    An NCHW tensor that one of this package's reference-shaped entry points returned (CorrBlock.__call__, BasicUpdateBlock.forward)
    carries its channels-last original as `_fs_cl` (from_channels_last): the reference's loop hands `corr`, `net` and `up_mask`
    straight from one swapped block to the next (pytorch/core/raft.py:125-137), and the consumer then continues from the twin --
    same autograd graph, one node earlier -- instead of transposing the copy back.  The twin is dropped
    as soon as the NCHW tensor was written in place (`_version`)."""
    if x.shape[1] % 4 != 0:
        raise RuntimeError("channel count must be a multiple of 4")
    if x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last) and not x.is_contiguous():
        return x.permute(0, 2, 3, 1)          # already [B][H][W][C] in memory (the channels_last context encoder's output): a view
    tw = getattr(x, "_fs_cl", None) if TWINS else None
    if tw is not None and tw[1] == x._version and tw[0].shape[0] == x.shape[0] and tw[0].shape[-1] >= x.shape[1] \
            and tuple(tw[0].shape[1:3]) == tuple(x.shape[2:]) and tw[0].requires_grad == x.requires_grad:
        return tw[0]
    return _ToCL.apply(x)


def as_nchw(x):
    """[B,H,W,C] -> [B,C,H,W] for the tensors the reference's loop hands from one swapped block to the next (`corr`:
    raft.py:125-127, `net`: raft.py:129, `up_mask`: raft.py:137): a permuted VIEW -- shape and values of the reference's tensor,
    torch.channels_last memory format, no copy; `to_channels_last` of it is the original again, and autograd's permutes cost
    nothing.  The one thing the reference's own code does to such a tensor besides passing it on -- `mask.view(N, 1, 9, 8, 8, H, W)`
    in upsample_flow, raft.py:75 -- splits the channel dimension only, which a channels-last view allows
    (test_reference_upsample_flow_takes_the_mask_view)."""
    if NCHW_VIEWS and TWINS:
        return x.permute(0, 3, 1, 2)
    return from_channels_last(x)


def from_channels_last(x):
    """[B,H,W,C] -> NCHW contiguous (what the reference's callers receive); the result remembers x (see to_channels_last)."""
    y = _FromCL.apply(x)
    if TWINS:
        y._fs_cl = (x, y._version)
    return y


class _UpdateBlockBase(nn.Module):
    small = False
    gma = False

    def _engine(self):
        e = self.__dict__.get("_eng")
        if e is None:
            e = _Engine(self, self.small, self.gma)
            self.__dict__["_eng"] = e
        return e

    def _attn_state(self, attention):
        """(state, anchor) shared by all calls of one step that pass the same attention tensor."""
        if attention is None or not (torch.is_grad_enabled() and attention.requires_grad):
            return None, None
        ast = self.__dict__.get("_ast")
        if ast is None or ast.key() is not attention or ast.consumed:
            ast = _AttnState(weakref.ref(attention))      # identity, not address: freed storage gets reused
            ast.anchor = _AttnFn.apply(ast, attention)
            self.__dict__["_ast"] = ast
        return ast, ast.anchor

    def _attn_transposed(self, attention):
        """The attention map as records ([B, N, ceil32(N)], gemm_rec.hpp), split once per attention tensor (one pair / one
        step) and reused by every iteration's `attn @ v` (rows of records along the contraction index) and, in backward, by
        `attn^T @ dagg` (the same tensor read k-major).  (Round 1 kept a transposed fp32 copy here instead.)"""
        if attention is None or not self._engine().gma or not ops.SPLIT_VOLUME_BWD:
            return None         # (SPLIT_VOLUME_BWD off = exact-fp32 test mode: keep the exact NN GEMM)
        from .gma import is_records
        if is_records(attention):       # the softmax wrote records over its logits (gma.ATTN_RECORDS): this IS the one copy
            t = attention.detach().view(attention.shape[0], attention.shape[-1], attention.shape[-1])
            t._fs_amax = ops.amax_one(t.device)         # (probabilities: split with the scale of a word holding 1.0)
            return t
        c = self.__dict__.get("_attn_t")
        if c is None or c[0]() is not attention or c[1] != attention._version:
            with torch.no_grad():
                B, N = attention.shape[0], attention.shape[-1]
                t = ops.to_records(attention.detach().reshape(B, N, N).contiguous().float())
            c = (weakref.ref(attention), attention._version, t)
            self.__dict__["_attn_t"] = c
        return c[2]

    def _ctx_state(self, eng, st, params, anchor, inp, track):
        """Context convolution of `inp`, shared by every call of a step that passes the same inp tensor (and the
        same parameter values).  With `track`, its backward is threaded through a 1-element anchor."""
        pkey = tuple((p.data_ptr(), p._version) for p in params)
        cst = self.__dict__.get("_cst")
        stale = (cst is None or cst.key[0]() is not inp or cst.key[1] != inp._version or cst.key[2] != pkey
                 or cst.consumed or cst.bufs is None or (cst.anchor is not None) != track
                 or (track and cst.key[3] is not st))
        if stale:
            with torch.no_grad():
                bufs = eng.context(inp.detach(), params)
            cst = _CtxState((weakref.ref(inp), inp._version, pkey, st), inp.detach(), bufs)
            if track:
                cst.anchor = _CtxFn.apply(eng, st, cst, params, anchor, inp)
                cst.keep = inp
            self.__dict__["_cst"] = cst
        return cst

    @L.on_tensor_device
    def forward_cl(self, net, inp, corr, flow, attention=None, need_mask=True, head_batch=None, grad_samples=None, motion_batch=None):
        """Channels-last entry used by our RAFT loop: no layout conversion at all.
        net/inp/corr: [B,H,W,C]; flow: [B,2,H,W]; attention (GMA only): [B,1,N,N].
        Returns (net', mask_cl or None, delta).  need_mask=False: the caller will not upsample this iteration's flow
        (test_mode, every iteration but the last); honoured when no gradient is being recorded."""
        eng = self._engine()
        params = tuple(eng.params())
        st, anchor = eng.param_state(params)
        ast, aanchor = self._attn_state(attention)
        if anchor is None:
            if torch.is_grad_enabled() and (ast is not None or any(t.requires_grad for t in (net, inp, corr, flow))):
                st = self.__dict__.get("_frozen_st")       # inputs need grads, params frozen: one state per step
                if st is None or st.consumed:
                    st = _ParamState(None)
                    st.zero = ops.zeros(1, device=net.device)
                    self.__dict__["_frozen_st"] = st
                anchor = st.zero
            else:
                cst = self._ctx_state(eng, None, params, None, inp, False)
                h, mask, delta, _ = eng.forward(net, cst.bufs, corr, flow, params, save=False, attn=attention,
                                                attn_t=self._attn_transposed(attention), need_mask=need_mask)
                return h, mask, delta
        track = torch.is_grad_enabled() and (inp.requires_grad or any(p.requires_grad for p in params))
        cst = self._ctx_state(eng, st, params, anchor, inp, track)
        attn = attention.detach() if attention is not None else None
        attn_r = self._attn_transposed(attention)
        if attn_r is None and attention is not None:
            from .gma import is_records
            if is_records(attention):       # (made under the split arithmetic, used under the exact one: there is no dense copy to fall back to)
                raise RuntimeError("the attention map holds records but the record GEMMs are switched off (exact arithmetic)")
        h, mask, delta = _UpdateFn.apply(eng, st, params, anchor, net, cst, cst.anchor, corr, flow, ast, attn, aanchor,
                                         attn_r, head_batch, grad_samples, motion_batch)
        return h, (mask if (eng.has_mask and head_batch is None) else None), delta

    def motion_batch(self, iters, net, grad_samples=None):
        """A MotionBatch for `iters` calls of forward_cl(..., motion_batch=...) on states shaped like `net`, or None where it does
        not apply (see head_batch).  The caller has the lookup write into `mb.corr[mb.n]` (CorrBlock(..., out=)).
        grad_samples: what the calls will pass as grad_samples (the batch then starts from zeroed slots)."""
        eng = self._engine()
        if not (MOTION_BATCH and torch.is_grad_enabled() and net.is_cuda):
            return None
        params = tuple(eng.params())
        st, anchor = eng.param_state(params)
        B, H, W, _ = net.shape
        if anchor is None or st.pending is None or not MotionBatch.fits(eng, iters, B, H, W):
            return None
        return MotionBatch(eng, iters, B, H, W, net.device, zero=grad_samples is not None)

    def head_batch(self, iters, net):
        """A HeadBatch for `iters` calls of forward_cl(..., head_batch=...) on states shaped like `net` ([B,H,W,hid]), or None
        where it does not apply (no mask head, no gradient being recorded, frozen parameters, switched off, too large)."""
        eng = self._engine()
        if not (HEAD_BATCH and eng.has_mask and torch.is_grad_enabled() and net.is_cuda):
            return None
        params = tuple(eng.params())
        st, anchor = eng.param_state(params)
        B, H, W, _ = net.shape
        if anchor is None or not HeadBatch.fits(iters, B, H, W):
            return None
        return HeadBatch(eng, params, st, anchor, iters, B, H, W, net.device)

    def _forward_nchw(self, net, inp, corr, flow, attention=None):
        # The channels-last copy of `inp` is reused while the caller passes the SAME tensor object at the same version
        # (identity through a weak reference: a freed tensor's address is handed out again by the caching allocator, so
        # an address-keyed cache would serve the previous pair's context features to the next pair).
        # When `inp` carries a gradient, the cache holds the channels-last tensor WEAKLY: the step's autograd graph keeps it alive
        # (_CtxFn's input) for exactly as long as it can be reused, and the module does not pin the previous step's graph.  Round 5
        # converted such an `inp` anew on every call, and with a new object per call the once-per-step context convolutions
        # (_ctx_state: shared by identity of their input) ran, forward and backward, once per ITERATION in the reference-shaped loop.
        cache = self.__dict__.get("_inp_cache")
        tracked = inp.requires_grad and torch.is_grad_enabled()
        inp_cl = None
        if cache is not None and cache[0]() is inp and cache[1] == inp._version and cache[3] == tracked:
            inp_cl = cache[2]() if tracked else cache[2]
        if inp_cl is None:
            inp_cl = to_channels_last(inp)
            self.__dict__["_inp_cache"] = (weakref.ref(inp), inp._version, weakref.ref(inp_cl) if tracked else inp_cl, tracked)
        h, mask, delta = self.forward_cl(to_channels_last(net), inp_cl, to_channels_last(corr), flow, attention)
        return as_nchw(h), (as_nchw(mask) if mask is not None else None), delta


class SmallUpdateBlock(_UpdateBlockBase):
    """update.py:99-112.  forward(net, inp, corr, flow) -> (net, None, delta_flow)."""
    small = True

    def __init__(self, args, hidden_dim=96):
        super().__init__()
        self.cor_planes = args.corr_levels * (2 * args.corr_radius + 1) ** 2
        self.encoder = SmallMotionEncoder(args)
        self.gru = ConvGRU(hidden_dim=hidden_dim, input_dim=82 + 64)
        self.flow_head = FlowHead(hidden_dim, hidden_dim=128)
        if hidden_dim != 96:
            raise NotImplementedError("the HIP small update block is built for hidden_dim=96 (the only value RAFT uses)")

    @L.on_tensor_device
    def forward(self, net, inp, corr, flow):
        return self._forward_nchw(net, inp, corr, flow)


class GMAUpdateBlock(_UpdateBlockBase):
    """gma_update.py:112-139.  forward(net, inp, corr, flow, attention) -> (net, mask, delta_flow).
    Same kernels as BasicUpdateBlock with a 512-channel GRU input [h | inp | motion | motion_global];
    the Aggregate step (to_v 1x1 conv, attn @ v, gamma mix) runs between the motion encoder and the GRU."""
    gma = True

    def __init__(self, args, hidden_dim=128):
        super().__init__()
        from .gma import Aggregate
        self.args = args
        self.cor_planes = args.corr_levels * (2 * args.corr_radius + 1) ** 2
        self.encoder = BasicMotionEncoder(args)
        self.gru = SepConvGRU(hidden_dim=hidden_dim, input_dim=128 + hidden_dim + hidden_dim)
        self.flow_head = FlowHead(hidden_dim, hidden_dim=256)
        self.mask = nn.Sequential(
            nn.Conv2d(128, 256, 3, padding=1),
            nn.ReLU(inplace=True),
            nn.Conv2d(256, 64 * 9, 1, padding=0))
        self.aggregator = Aggregate(args=self.args, dim=128, dim_head=128, heads=self.args.num_heads)
        if hidden_dim != 128:
            raise NotImplementedError("the HIP update block is built for hidden_dim=128 (RAFT-GMA's value)")
        if self.args.num_heads != 1:
            raise NotImplementedError("the fused Aggregate step is built for num_heads=1 (train_gma.py:354 default)")

    @L.on_tensor_device
    def forward(self, net, inp, corr, flow, attention):
        return self._forward_nchw(net, inp, corr, flow, attention)


class BasicUpdateBlock(_UpdateBlockBase):
    """update.py:114-136.  forward(net, inp, corr, flow, upsample=True) -> (net, mask, delta_flow)."""

    def __init__(self, args, hidden_dim=128, input_dim=128):
        super().__init__()
        self.args = args
        self.cor_planes = args.corr_levels * (2 * args.corr_radius + 1) ** 2
        self.encoder = BasicMotionEncoder(args)
        self.gru = SepConvGRU(hidden_dim=hidden_dim, input_dim=128 + hidden_dim)
        self.flow_head = FlowHead(hidden_dim, hidden_dim=256)
        self.mask = nn.Sequential(
            nn.Conv2d(128, 256, 3, padding=1),
            nn.ReLU(inplace=True),
            nn.Conv2d(256, 64 * 9, 1, padding=0))
        if hidden_dim != 128 or input_dim != 128:
            raise NotImplementedError("the HIP update block is built for hidden_dim=input_dim=128 (RAFT's values)")

    @L.on_tensor_device
    def forward(self, net, inp, corr, flow, upsample=True):
        return self._forward_nchw(net, inp, corr, flow)
