"""GMA attention and aggregation (pytorch/core/gma.py) on libfsraft kernels.

``Attention`` runs once per image pair: q,k = to_qk(context) (1x1 implicit-GEMM conv), sim = scale * q k^T
(split (fp16x3) NT GEMM), row softmax in place over the [B,1,N,N] map.  ``Aggregate`` runs every iteration inside
``GMAUpdateBlock`` (core/update.py) as v = to_v(motion), attn @ v, motion + gamma * out; the stand-alone
``Aggregate.forward`` here uses the same kernels for callers outside the update block.

Single-head, content-only attention (the reference's default: train_gma.py:354, no --position_* flag) is the HIP
path.  The optional relative-position terms (gma.py:6-31, 63-69) and multi-head maps are composed from torch
tensor ops on the GPU around the same softmax; they are not part of the measured path.
"""
import torch
from torch import nn

from .. import _lib as L
from .. import ops
from ..ops import Dst, V
from .update import from_channels_last, to_channels_last


class RelPosEmb(nn.Module):
    """gma.py:6-31: per-axis relative position embeddings; q [b,heads,h,w,d] -> scores [b,heads,h,w,h,w]."""

    def __init__(self, max_pos_size, dim_head):
        super().__init__()
        self.rel_height = nn.Embedding(2 * max_pos_size - 1, dim_head)
        self.rel_width = nn.Embedding(2 * max_pos_size - 1, dim_head)
        deltas = torch.arange(max_pos_size).view(1, -1) - torch.arange(max_pos_size).view(-1, 1)
        self.register_buffer("rel_ind", deltas + max_pos_size - 1)

    def forward(self, q):
        b, heads, h, w, c = q.shape
        he = self.rel_height(self.rel_ind[:h, :h].reshape(-1)).view(h, h, c)      # [x, u, d]
        we = self.rel_width(self.rel_ind[:w, :w].reshape(-1)).view(w, w, c)       # [y, v, d]
        hs = torch.einsum("bhxyd,xud->bhxyu", q, he)[..., None]                   # broadcast over v
        ws = torch.einsum("bhxyd,yvd->bhxyv", q, we)[..., None, :]                # broadcast over u
        return hs + ws


def _conv1x1(x_cl, w, cin, cout, B, H, W, mode):
    wpk = ops.pack_weight(w, [cin], mode)
    wsp = ops.pack_weight(w, [cin], 10 + mode)
    n = cin if mode == 1 else cout
    out = torch.empty(B, H, W, n, device=x_cl.device, dtype=torch.float32)
    ops.conv_forward([V(x_cl, cout if mode == 1 else cin)], wpk, None, B, H, W, 1, 1, n, [Dst.nhwc(out)], wpk_split=wsp)
    return out


def _conv1x1_wgrad(dy_cl, x_cl, cin, cout, B, H, W):
    dwpk = torch.zeros(cout, ops.conv_ktot([cin], 1, 1), device=x_cl.device, dtype=torch.float32)
    ops.conv_wgrad(V(dy_cl, cout), [V(x_cl, cin)], dwpk, B, H, W, 1, 1)
    return ops.unpack_weight_grad(dwpk, (cout, cin, 1, 1), [cin])


def _attn_t_times(attn, x, B, N, C):
    """attn^T @ x for attn [B,N,N], x [B,N,C] contiguous."""
    out = torch.empty(B, N, C, device=x.device, dtype=torch.float32)
    if N % 4 == 0:
        ops.gemm_tn_raw(attn.data_ptr(), N, N * N, x.data_ptr(), C, N * C, out.data_ptr(), C, N * C, B, N, C, N)
    else:
        at = ops.transpose_batched(attn.view(B, N, N))
        ops.gemm_raw(at.data_ptr(), N, N * N, x.data_ptr(), C, N * C, out.data_ptr(), C, N * C, B, N, C, N, False)
    return out


# The attention map of one pair is 198 MB in fp32 at 55 x 128 (N = 7040), 0.8 GB for a batch of four, and its only readers on
# the training path are record GEMMs (attn @ v, attn^T @ dagg of every iteration) and the softmax backward.  ATTN_RECORDS keeps
# ONE copy of it, as records, written by the softmax itself over the logits (N % 32 == 0: a record row is as long as the fp32
# row).  The tensor handed from Attention.forward_cl to the update block keeps shape [B,1,N,N] and dtype float32 -- autograd only
# needs those -- but its BYTES are records; `is_records` tells the consumers (update._UpdateBlockBase._attn_transposed).  The
# reference API (Attention.forward) always returns the dense map.
ATTN_RECORDS = True


def _records_ok(D, N):
    """Shapes the record softmax pair covers (LDS holds a row of the map and of its gradient, 8 bytes per element, within the
    64 KB a launch gets without an opt-in: N <= 8160), split arithmetic."""
    return ATTN_RECORDS and ops.SPLIT_VOLUME_BWD and D % 32 == 0 and N % 32 == 0 and 32 <= N <= 8160


def mark_records(t, like=None):
    """Flag `t` as holding records (or carry the flag of `like` over to a detached alias)."""
    if like is None or is_records(like):
        t._fs_attn_records = True
    return t


def is_records(t):
    return bool(getattr(t, "_fs_attn_records", False))


class _AttentionFn(torch.autograd.Function):
    """context [B,H,W,C] channels-last, to_qk weight [2D,C,1,1] -> softmax(scale q k^T) as [B,1,N,N] (records=True: as
    records in that shape, see ATTN_RECORDS)."""

    @staticmethod
    def forward(ctx, x_cl, w, scale, records=False):
        L.require_cuda_f32(x_cl, w)
        B, H, W, C = x_cl.shape
        D, N = w.shape[0] // 2, H * W
        w = w.detach().contiguous().float()
        qk = _conv1x1(x_cl, w, C, 2 * D, B, H, W, 0)
        attn = torch.empty(B, 1, N, N, device=x_cl.device, dtype=torch.float32)
        if ops.SPLIT_VOLUME_BWD and D % 32 == 0:          # record GEMM core: q and k are record slices of one [N][2D] tensor
            qkr = ops.to_records(qk.view(B, N, 2 * D))
            ops.gemm_rec_nt_raw(qkr.data_ptr(), 2 * D, N * 2 * D, qkr.data_ptr() + 4 * D, 2 * D, N * 2 * D, attn.data_ptr(), N, N * N,
                                B, N, N, D, scale, a_amax=ops.amax_of(qkr), b_amax=ops.amax_of(qkr))
        else:
            ops.gemm_raw(qk.data_ptr(), 2 * D, N * 2 * D, qk.data_ptr() + 4 * D, 2 * D, N * 2 * D, attn.data_ptr(), N, N * N,
                         B, N, N, D, True, scale)
        records = bool(records) and _records_ok(D, N)
        if records:
            ops.softmax_rows_rec_(attn)
        else:
            ops.softmax_rows_(attn)
        ctx.save_for_backward(x_cl, w, qk, attn)
        ctx.scale, ctx.records = scale, records
        return attn

    @staticmethod
    def backward(ctx, dA):
        x_cl, w, qk, attn = ctx.saved_tensors
        B, H, W, C = x_cl.shape
        D, N, scale = w.shape[0] // 2, H * W, ctx.scale
        # (the gradient tensor is overwritten: _AttnFn.backward allocates it for us and says so; anything else is copied first)
        own = bool(getattr(dA, "_fs_owned", False)) and dA.is_contiguous()
        dA = dA if own else dA.contiguous().clone()
        dS = ops.softmax_rows_bwd_rec_(attn, dA) if ctx.records else ops.softmax_rows_bwd_(attn, dA)
        ds_word = ops.amax_of(dS)             # records: the word softmax_rows_bwd_rec_ split them with
        del dA
        dqk = torch.empty_like(qk)
        # dq = scale dS k ; dk = scale dS^T q
        if ops.SPLIT_VOLUME_BWD and D % 32 == 0:
            # record GEMM core: dS split to records once; dq = dS . (k^T)^T with k^T [D][N] (rows of records along j),
            # dk = dS^T . q with both operands read k-major (records along the output index)
            dSr = dS.view(B, N, N) if ctx.records else ops.to_records(dS.view(B, N, N))
            ds_word = ds_word if ctx.records else ops.amax_of(dSr)
            del dS
            Nr = dSr.shape[-1]
            qkr = ops.to_records(qk.view(B, N, 2 * D))                                   # q = records 0..D/32-1 of a row, k the rest
            kt = ops.to_records(ops.transpose_batched(qk.view(B, N, 2 * D)[:, :, D:].contiguous()))    # [B, D, Nr]
            ops.gemm_rec_nt_raw(dSr.data_ptr(), Nr, N * Nr, kt.data_ptr(), Nr, D * Nr, dqk.data_ptr(), 2 * D, N * 2 * D, B, N, D, Nr,
                                scale, ksplit=2, a_amax=ds_word, b_amax=ops.amax_of(kt))
            ops.gemm_rec_tn_raw(dSr.data_ptr(), Nr, N * Nr, qkr.data_ptr(), 2 * D, N * 2 * D, dqk.data_ptr() + 4 * D, 2 * D, N * 2 * D,
                                B, N, D, N, scale, ksplit=2, a_amax=ds_word, b_amax=ops.amax_of(qkr))
            dw = _conv1x1_wgrad(dqk, x_cl, C, 2 * D, B, H, W) if ctx.needs_input_grad[1] else None
            dx = _conv1x1(dqk, w, C, 2 * D, B, H, W, 1) if ctx.needs_input_grad[0] else None
            return dx, dw, None, None
        dSt = ops.transpose_batched(dS.view(B, N, N))        # tiled transpose: 4-6 TB/s, the strided copy reaches 1.5-2
        if N % 4 == 0 and D % 4 == 0:      # k-major operands on the transposed-read split (fp16x3) GEMM
            ops.gemm_tn_raw(dSt.data_ptr(), N, N * N, qk.data_ptr() + 4 * D, 2 * D, N * 2 * D, dqk.data_ptr(), 2 * D,
                            N * 2 * D, B, N, D, N, scale)
            ops.gemm_tn_raw(dS.data_ptr(), N, N * N, qk.data_ptr(), 2 * D, N * 2 * D, dqk.data_ptr() + 4 * D, 2 * D,
                            N * 2 * D, B, N, D, N, scale)
        else:
            ops.gemm_raw(dS.data_ptr(), N, N * N, qk.data_ptr() + 4 * D, 2 * D, N * 2 * D, dqk.data_ptr(), 2 * D,
                         N * 2 * D, B, N, D, N, False, scale)
            ops.gemm_raw(dSt.data_ptr(), N, N * N, qk.data_ptr(), 2 * D, N * 2 * D, dqk.data_ptr() + 4 * D, 2 * D,
                         N * 2 * D, B, N, D, N, False, scale)
        dw = _conv1x1_wgrad(dqk, x_cl, C, 2 * D, B, H, W) if ctx.needs_input_grad[1] else None
        dx = _conv1x1(dqk, w, C, 2 * D, B, H, W, 1) if ctx.needs_input_grad[0] else None
        return dx, dw, None, None


class Attention(nn.Module):
    """gma.py:34-76.  forward(fmap [B,dim,H,W]) -> attention [B,heads,N,N]."""

    def __init__(self, *, args, dim, max_pos_size=100, heads=4, dim_head=128):
        super().__init__()
        self.args = args
        self.heads = heads
        self.scale = dim_head ** -0.5
        inner_dim = heads * dim_head
        self.to_qk = nn.Conv2d(dim, inner_dim * 2, 1, bias=False)
        self.pos_emb = RelPosEmb(max_pos_size, dim_head)

    def _positional(self):
        return bool(getattr(self.args, "position_only", False) or getattr(self.args, "position_and_content", False))

    @L.on_tensor_device
    def forward(self, fmap):
        if self.heads == 1 and not self._positional() and fmap.shape[1] % 4 == 0:
            return self.forward_cl(to_channels_last(fmap.float()))
        return self._forward_general(fmap)

    @L.on_tensor_device
    def forward_cl(self, fmap_cl, records=False):
        """Channels-last entry (content-only, single head): [B,H,W,dim] -> [B,1,N,N].  records=True (the update block's
        forward_cl is the consumer): the map may come back as records in that shape (ATTN_RECORDS above; `is_records`)."""
        if self.heads != 1 or self._positional():
            return self._forward_general(from_channels_last(fmap_cl))
        _, H, W, _ = fmap_cl.shape
        if records and _records_ok(self.to_qk.weight.shape[0] // 2, H * W):
            return mark_records(_AttentionFn.apply(fmap_cl, self.to_qk.weight, self.scale, True))
        return _AttentionFn.apply(fmap_cl, self.to_qk.weight, self.scale)

    def _forward_general(self, fmap):
        heads, (b, c, h, w) = self.heads, fmap.shape
        q, k = torch.einsum("oc,bchw->bohw", self.to_qk.weight[:, :, 0, 0], fmap).chunk(2, dim=1)
        q = self.scale * q.reshape(b, heads, -1, h, w).permute(0, 1, 3, 4, 2)
        k = k.reshape(b, heads, -1, h, w).permute(0, 1, 3, 4, 2)
        if getattr(self.args, "position_only", False):
            sim = self.pos_emb(q)
        else:
            sim = torch.einsum("bhxyd,bhuvd->bhxyuv", q, k)
            if getattr(self.args, "position_and_content", False):
                sim = sim + self.pos_emb(q)
        sim = sim.reshape(b, heads, h * w, h * w).contiguous()
        return sim.softmax(dim=-1)


class _AggregateFn(torch.autograd.Function):
    """(attn [B,1,N,N], fmap channels-last [B,H,W,C], to_v weight [C,C,1,1], gamma [1]) -> fmap + gamma (attn @ v)."""

    @staticmethod
    def forward(ctx, attn, x_cl, w, gamma):
        if is_records(attn):
            raise TypeError("Aggregate got an attention map that holds records (Attention.forward_cl(..., records=True) is for "
                            "the update block's forward_cl); Attention.forward returns the dense map")
        L.require_cuda_f32(attn, x_cl, w, gamma)
        B, H, W, C = x_cl.shape
        N = H * W
        attn = attn.contiguous()
        w = w.detach().contiguous().float()
        v = _conv1x1(x_cl, w, C, C, B, H, W, 0)
        agg = torch.empty_like(v)
        ops.gemm_raw(attn.data_ptr(), N, N * N, v.data_ptr(), C, N * C, agg.data_ptr(), C, N * C, B, N, C, N, False)
        out = torch.empty_like(v)
        g = gamma.detach().float().reshape(-1)
        ops.gma_mix_fwd(V(x_cl, C), V(agg), g, V(out))
        ctx.save_for_backward(attn, x_cl, w, g, v, agg)
        return out

    @staticmethod
    def backward(ctx, dout):
        attn, x_cl, w, g, v, agg = ctx.saved_tensors
        B, H, W, C = x_cl.shape
        N = H * W
        dout = dout.contiguous()
        dx = torch.zeros_like(x_cl)
        dagg = torch.empty_like(agg)
        dgamma = torch.zeros(1, device=dout.device)
        ops.gma_mix_bwd(V(dout, C), V(agg), g, V(dx, C), V(dagg), dgamma)
        dv = _attn_t_times(attn, dagg, B, N, C).view(B, H, W, C)
        dattn = None
        if ctx.needs_input_grad[0]:
            dattn = torch.empty_like(attn)
            ops.gemm_raw(dagg.data_ptr(), C, N * C, v.data_ptr(), C, N * C, dattn.data_ptr(), N, N * N, B, N, N, C, True)
        dw = _conv1x1_wgrad(dv, x_cl, C, C, B, H, W) if ctx.needs_input_grad[2] else None
        if ctx.needs_input_grad[1]:
            ops.conv_forward([V(dv, C)], ops.pack_weight(w, [C], 1), None, B, H, W, 1, 1, C, [Dst.nhwc(dx, 0, 0, True)],
                             wpk_split=ops.pack_weight(w, [C], 11))
        return dattn, dx, dw, dgamma


class Aggregate(nn.Module):
    """gma.py:79-115.  forward(attn [B,heads,N,N], fmap [B,dim,H,W]) -> fmap + gamma * project(attn @ to_v(fmap))."""

    def __init__(self, args, dim, heads=4, dim_head=128):
        super().__init__()
        self.args = args
        self.heads = heads
        self.scale = dim_head ** -0.5
        inner_dim = heads * dim_head
        self.to_v = nn.Conv2d(dim, inner_dim, 1, bias=False)
        self.gamma = nn.Parameter(torch.zeros(1))
        self.project = nn.Conv2d(inner_dim, dim, 1, bias=False) if dim != inner_dim else None

    @L.on_tensor_device
    def forward(self, attn, fmap):
        if self.heads == 1 and self.project is None and fmap.shape[1] % 4 == 0:
            return from_channels_last(_AggregateFn.apply(attn, to_channels_last(fmap.float()), self.to_v.weight, self.gamma))
        heads, (b, c, h, w) = self.heads, fmap.shape
        v = torch.einsum("oc,bchw->bohw", self.to_v.weight[:, :, 0, 0], fmap).reshape(b, heads, -1, h * w)
        out = torch.einsum("bhij,bhdj->bhdi", attn, v).reshape(b, -1, h, w)
        if self.project is not None:
            out = torch.einsum("oc,bchw->bohw", self.project.weight[:, :, 0, 0], out)
        return fmap + self.gamma * out
