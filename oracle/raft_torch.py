"""CPU oracle for the RAFT hot path (TEST INFRASTRUCTURE, NOT PRODUCT CODE).

This file restates, in plain fp32 torch ops on the CPU, the algorithm of the
reference's PyTorch RAFT path.  It exists to *check* the HIP kernels: only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import it.  Nothing under ``flow_supervisor_amd/`` imports it, and the
product path raises if its HIP library is missing instead of falling back here.

Parity status: PINNED.  Every function below is checked in
``tests/test_oracle_vs_golden.py`` against fixtures in ``tests/golden/*.npz``
that were produced by importing the reference itself
(``/root/reference/pytorch/core``) with ``tests/golden/make_golden.py``.
The reference ships no tests or golden vectors of its own.

Everything is written functionally over a ``state_dict``-style mapping
``{key: tensor}`` using the reference's parameter names, so a reference
checkpoint can be fed straight in.  Citations are to files under
``/root/reference/pytorch``.
"""
import math

import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------
# helpers: core/utils/utils.py
# --------------------------------------------------------------------------
def coords_grid(batch, ht, wd, device="cpu"):
    """[B,2,ht,wd] float grid, channel 0 = x (column), channel 1 = y (row).
    core/utils/utils.py:74-77."""
    ys = torch.arange(ht, dtype=torch.float32, device=device).view(1, 1, ht, 1).expand(batch, 1, ht, wd)
    xs = torch.arange(wd, dtype=torch.float32, device=device).view(1, 1, 1, wd).expand(batch, 1, ht, wd)
    return torch.cat([xs, ys], dim=1).contiguous()


def bilinear_sample_zero(img, x, y):
    """Bilinear sample of img[M,h,w] at pixel coordinates x,y [M,K]; taps outside
    the image contribute zero.  Equivalent to grid_sample(align_corners=True,
    padding_mode='zeros') as used by core/utils/utils.py:57-71, written as an
    explicit 4-tap gather so it does not depend on grid_sample itself."""
    M, h, w = img.shape
    x0 = torch.floor(x)
    y0 = torch.floor(y)
    fx = x - x0
    fy = y - y0
    x0 = x0.long()
    y0 = y0.long()
    flat = img.reshape(M, h * w)
    out = torch.zeros_like(x)
    for dy_, wy in ((0, 1.0 - fy), (1, fy)):
        for dx_, wx in ((0, 1.0 - fx), (1, fx)):
            xi = x0 + dx_
            yi = y0 + dy_
            ok = (xi >= 0) & (xi < w) & (yi >= 0) & (yi < h)
            idx = (yi.clamp(0, h - 1) * w + xi.clamp(0, w - 1))
            v = torch.gather(flat, 1, idx)
            out = out + torch.where(ok, v * wy * wx, torch.zeros_like(v))
    return out


def upflow8(flow):
    """8x bilinear (align_corners=True) upsample times 8.  core/utils/utils.py:80-82."""
    h, w = flow.shape[-2:]
    return 8.0 * F.interpolate(flow, size=(8 * h, 8 * w), mode="bilinear", align_corners=True)


def input_pad_amounts(ht, wd, mode="sintel"):
    """[left,right,top,bottom] replicate-pad amounts.  core/utils/utils.py:9-16."""
    ph = (((ht // 8) + 1) * 8 - ht) % 8
    pw = (((wd // 8) + 1) * 8 - wd) % 8
    if mode == "sintel":
        return [pw // 2, pw - pw // 2, ph // 2, ph - ph // 2]
    return [pw // 2, pw - pw // 2, 0, ph]


# --------------------------------------------------------------------------
# a1/a2: all-pairs volume + pyramid.  core/corr.py:13-27, 52-60
# --------------------------------------------------------------------------
def corr_volume(fmap1, fmap2):
    """V[b,i,j] = <f1[b,:,i], f2[b,:,j]> / sqrt(C) -> [B,H,W,1,H,W].  core/corr.py:52-60."""
    B, C, H, W = fmap1.shape
    a = fmap1.reshape(B, C, H * W).transpose(1, 2)
    b = fmap2.reshape(B, C, H * W)
    v = torch.bmm(a, b) / math.sqrt(float(C))
    return v.reshape(B, H, W, 1, H, W)


def corr_pyramid(fmap1, fmap2, num_levels=4):
    """List of [B*H*W,1,h_l,w_l]; level l+1 = 2x2 mean (floor) of level l.  core/corr.py:19-27."""
    B, C, H, W = fmap1.shape
    lvl = corr_volume(fmap1, fmap2).reshape(B * H * W, 1, H, W)
    pyr = [lvl]
    for _ in range(num_levels - 1):
        lvl = F.avg_pool2d(lvl, 2, stride=2)
        pyr.append(lvl)
    return pyr


# --------------------------------------------------------------------------
# a3: radius-r pyramid lookup.  core/corr.py:29-50
# --------------------------------------------------------------------------
def corr_lookup(pyramid, coords, radius):
    """coords [B,2,H,W] (x,y) -> [B, L*(2r+1)^2, H, W].

    Output channel = l*(2r+1)^2 + i*(2r+1) + j samples level l at
    (x/2^l + (i-r), y/2^l + (j-r)): the *x* offset is the slow index.  This is
    what core/corr.py:37-46 produces: `delta = stack(meshgrid(dy,dx))` puts the
    first-axis value into the last-dim slot 0, which bilinear_sampler reads as x."""
    B, _, H, W = coords.shape
    r = radius
    n = 2 * r + 1
    xy = coords.permute(0, 2, 3, 1).reshape(B * H * W, 2)
    off = torch.arange(-r, r + 1, dtype=torch.float32, device=coords.device)
    ox = off.view(n, 1).expand(n, n).reshape(1, n * n)   # slow index -> x offset
    oy = off.view(1, n).expand(n, n).reshape(1, n * n)   # fast index -> y offset
    outs = []
    for l, lvl in enumerate(pyramid):
        x = xy[:, 0:1] / (2 ** l) + ox
        y = xy[:, 1:2] / (2 ** l) + oy
        s = bilinear_sample_zero(lvl[:, 0], x, y)         # [BHW, n*n]
        outs.append(s.view(B, H, W, n * n))
    out = torch.cat(outs, dim=-1)
    return out.permute(0, 3, 1, 2).contiguous()


# --------------------------------------------------------------------------
# a4/a5: on-the-fly ("alternate") lookup.  core/corr.py:63-91 +
# alt_cuda_corr/correlation_kernel.cu:18-119 (forward), :122-256 (backward)
# --------------------------------------------------------------------------
def alt_corr_level(fmap1_nhwc, fmap2_nhwc, coords, radius):
    """One call of alt_cuda_corr.forward restated with dense torch ops.

    fmap1 [B,H1,W1,C], fmap2 [B,H2,W2,C], coords [B,1,H1,W1,2] (x,y) ->
    [B,1,(2r+1)^2,H1,W1], UNSCALED (caller divides by sqrt(C), corr.py:91).
    For every integer offset (iy,ix) in [0,2r+1]^2 the kernel dots f1 with the
    fmap2 pixel (floor(y)-r+iy, floor(x)-r+ix) (zero outside, .cu:80-83) and
    splats the dot into the <=4 neighbouring outputs with weights
    dy*dx, dy*(1-dx), (1-dy)*dx, (1-dy)*(1-dx) (.cu:92-114); output channel
    index is iy + (2r+1)*ix (.cu:92-95)."""
    B, H1, W1, C = fmap1_nhwc.shape
    _, H2, W2, _ = fmap2_nhwc.shape
    r = radius
    rd = 2 * r + 1
    x = coords[:, 0, :, :, 0]
    y = coords[:, 0, :, :, 1]
    fx0 = torch.floor(x)
    fy0 = torch.floor(y)
    dx = (x - fx0).unsqueeze(1)
    dy = (y - fy0).unsqueeze(1)
    x0 = fx0.long()
    y0 = fy0.long()
    f2flat = fmap2_nhwc.reshape(B, H2 * W2, C)
    dots = torch.zeros(B, rd + 1, rd + 1, H1, W1, dtype=fmap1_nhwc.dtype, device=fmap1_nhwc.device)
    for iy in range(rd + 1):
        for ix in range(rd + 1):
            h2 = y0 - r + iy
            w2 = x0 - r + ix
            ok = (h2 >= 0) & (h2 < H2) & (w2 >= 0) & (w2 < W2)
            idx = (h2.clamp(0, H2 - 1) * W2 + w2.clamp(0, W2 - 1)).reshape(B, H1 * W1, 1).expand(B, H1 * W1, C)
            g = torch.gather(f2flat, 1, idx).reshape(B, H1, W1, C)
            s = (g * fmap1_nhwc).sum(-1)
            dots[:, iy, ix] = torch.where(ok, s, torch.zeros_like(s))
    # out[iy_o, ix_o] = (1-dy)(1-dx) d[iy_o,ix_o] + (1-dy)dx d[iy_o,ix_o+1]
    #                 + dy(1-dx) d[iy_o+1,ix_o] + dy dx d[iy_o+1,ix_o+1]
    d00 = dots[:, :rd, :rd]
    d01 = dots[:, :rd, 1:]
    d10 = dots[:, 1:, :rd]
    d11 = dots[:, 1:, 1:]
    dyb = dy.unsqueeze(1)
    dxb = dx.unsqueeze(1)
    out = (1 - dyb) * (1 - dxb) * d00 + (1 - dyb) * dxb * d01 + dyb * (1 - dxb) * d10 + dyb * dxb * d11
    # [B, iy, ix, H, W] -> channel = iy + rd*ix  (ix slow)
    out = out.permute(0, 2, 1, 3, 4).reshape(B, 1, rd * rd, H1, W1)
    return out


def alt_corr_lookup(fmap1, fmap2, coords, num_levels=4, radius=4):
    """AlternateCorrBlock.__call__ (core/corr.py:74-91): fmap2 is avg-pooled per
    level, fmap1 stays at level 0, coords are divided by 2^l."""
    B, C, H, W = fmap1.shape
    f1 = fmap1.permute(0, 2, 3, 1).contiguous()
    c = coords.permute(0, 2, 3, 1)
    outs = []
    f2 = fmap2
    for l in range(num_levels):
        if l > 0:
            f2 = F.avg_pool2d(f2, 2, stride=2)
        f2n = f2.permute(0, 2, 3, 1).contiguous()
        ci = (c / 2 ** l).reshape(B, 1, H, W, 2).contiguous()
        outs.append(alt_corr_level(f1, f2n, ci, radius).squeeze(1))
    out = torch.stack(outs, dim=1).reshape(B, -1, H, W)
    return out / math.sqrt(float(C))


# --------------------------------------------------------------------------
# a6-a8: update block.  core/update.py
# --------------------------------------------------------------------------
def _conv(sd, name, x, pad):
    return F.conv2d(x, sd[name + ".weight"], sd[name + ".bias"], padding=pad)


def basic_motion_encoder(sd, p, flow, corr):
    """core/update.py:79-97."""
    cor = F.relu(_conv(sd, p + "convc1", corr, 0))
    cor = F.relu(_conv(sd, p + "convc2", cor, 1))
    flo = F.relu(_conv(sd, p + "convf1", flow, 3))
    flo = F.relu(_conv(sd, p + "convf2", flo, 1))
    out = F.relu(_conv(sd, p + "conv", torch.cat([cor, flo], 1), 1))
    return torch.cat([out, flow], 1)


def small_motion_encoder(sd, p, flow, corr):
    """core/update.py:62-77."""
    cor = F.relu(_conv(sd, p + "convc1", corr, 0))
    flo = F.relu(_conv(sd, p + "convf1", flow, 3))
    flo = F.relu(_conv(sd, p + "convf2", flo, 1))
    out = F.relu(_conv(sd, p + "conv", torch.cat([cor, flo], 1), 1))
    return torch.cat([out, flow], 1)


def _gru_pass(sd, p, sfx, h, x, pad):
    hx = torch.cat([h, x], 1)
    z = torch.sigmoid(_conv(sd, p + "convz" + sfx, hx, pad))
    r = torch.sigmoid(_conv(sd, p + "convr" + sfx, hx, pad))
    q = torch.tanh(_conv(sd, p + "convq" + sfx, torch.cat([r * h, x], 1), pad))
    return (1 - z) * h + z * q


def sep_conv_gru(sd, p, h, x):
    """core/update.py:33-60: (1,5) pass then (5,1) pass."""
    h = _gru_pass(sd, p, "1", h, x, (0, 2))
    h = _gru_pass(sd, p, "2", h, x, (2, 0))
    return h


def conv_gru(sd, p, h, x):
    """core/update.py:16-31."""
    return _gru_pass(sd, p, "", h, x, 1)


def flow_head(sd, p, h):
    """core/update.py:6-14."""
    return _conv(sd, p + "conv2", F.relu(_conv(sd, p + "conv1", h, 1)), 1)


def basic_update_block(sd, p, net, inp, corr, flow):
    """core/update.py:127-136 -> (net, mask, delta_flow)."""
    mf = basic_motion_encoder(sd, p + "encoder.", flow, corr)
    net = sep_conv_gru(sd, p + "gru.", net, torch.cat([inp, mf], 1))
    delta = flow_head(sd, p + "flow_head.", net)
    m = _conv(sd, p + "mask.2", F.relu(_conv(sd, p + "mask.0", net, 1)), 0)
    return net, 0.25 * m, delta


def small_update_block(sd, p, net, inp, corr, flow):
    """core/update.py:106-112 -> (net, None, delta_flow)."""
    mf = small_motion_encoder(sd, p + "encoder.", flow, corr)
    net = conv_gru(sd, p + "gru.", net, torch.cat([inp, mf], 1))
    delta = flow_head(sd, p + "flow_head.", net)
    return net, None, delta


# --------------------------------------------------------------------------
# a11: GMA attention / aggregate / update block (single head, content only).
# core/gma.py:54-76, 102-115; core/gma_update.py:127-139
# --------------------------------------------------------------------------
def gma_attention(sd, p, fmap, dim_head=128):
    """softmax over all N positions of (scale*q) . k with q,k = chunk(to_qk(fmap)) -> [B,1,N,N]  (gma.py:54-76)."""
    B, C, H, W = fmap.shape
    qk = F.conv2d(fmap, sd[p + "to_qk.weight"])
    q, k = qk[:, :dim_head].reshape(B, dim_head, H * W), qk[:, dim_head:].reshape(B, dim_head, H * W)
    sim = torch.bmm((dim_head ** -0.5 * q).transpose(1, 2), k)            # [B, N(xy), N(uv)]
    return torch.softmax(sim, dim=-1).unsqueeze(1)


def gma_aggregate(sd, p, attn, fmap):
    """fmap + gamma * (attn @ to_v(fmap))  (gma.py:102-115, heads=1 so no projection)."""
    B, C, H, W = fmap.shape
    v = F.conv2d(fmap, sd[p + "to_v.weight"]).reshape(B, C, H * W)        # [B, d, N(j)]
    out = torch.bmm(attn[:, 0], v.transpose(1, 2))                        # [B, N(i), d]
    return fmap + sd[p + "gamma"] * out.transpose(1, 2).reshape(B, C, H, W)


def gma_update_block(sd, p, net, inp, corr, flow, attn):
    """core/gma_update.py:127-139 -> (net, mask, delta_flow)."""
    mf = basic_motion_encoder(sd, p + "encoder.", flow, corr)
    mfg = gma_aggregate(sd, p + "aggregator.", attn, mf)
    net = sep_conv_gru(sd, p + "gru.", net, torch.cat([inp, mf, mfg], 1))
    delta = flow_head(sd, p + "flow_head.", net)
    m = _conv(sd, p + "mask.2", F.relu(_conv(sd, p + "mask.0", net, 1)), 0)
    return net, 0.25 * m, delta


# --------------------------------------------------------------------------
# a9: convex 8x upsampler.  core/raft.py:72-83
# --------------------------------------------------------------------------
def upsample_flow(flow, mask):
    """flow [N,2,H,W], mask [N,576,H,W] -> [N,2,8H,8W].
    mask channel = k*64 + sy*8 + sx, k = ky*3+kx over the zero-padded 3x3
    neighbourhood of 8*flow; softmax over k."""
    N, _, H, W = flow.shape
    m = torch.softmax(mask.reshape(N, 9, 8, 8, H, W), dim=1)
    fp = F.pad(8.0 * flow, (1, 1, 1, 1))
    out = torch.zeros(N, 2, 8, 8, H, W, dtype=flow.dtype, device=flow.device)
    for ky in range(3):
        for kx in range(3):
            nb = fp[:, :, ky:ky + H, kx:kx + W]                      # [N,2,H,W]
            out = out + m[:, ky * 3 + kx].unsqueeze(1) * nb.reshape(N, 2, 1, 1, H, W)
    # [N,2,sy,sx,H,W] -> [N,2,H,sy,W,sx]
    return out.permute(0, 1, 4, 2, 5, 3).reshape(N, 2, 8 * H, 8 * W)


# --------------------------------------------------------------------------
# encoders (callers of the path; kept as framework convs in the product too).
# core/extractor.py
# --------------------------------------------------------------------------
def _norm(sd, name, x, kind):
    if kind == "instance":
        return F.instance_norm(x)
    if kind == "batch":   # frozen statistics (freeze_bn, core/raft.py:58-61)
        return F.batch_norm(x, sd[name + ".running_mean"], sd[name + ".running_var"],
                            sd[name + ".weight"], sd[name + ".bias"], training=False)
    return x


def _res_block(sd, p, x, kind, stride):
    """core/extractor.py:6-56."""
    y = F.relu(_norm(sd, p + "norm1", F.conv2d(x, sd[p + "conv1.weight"], sd[p + "conv1.bias"], stride=stride, padding=1), kind))
    y = F.relu(_norm(sd, p + "norm2", F.conv2d(y, sd[p + "conv2.weight"], sd[p + "conv2.bias"], padding=1), kind))
    if stride != 1:
        x = F.conv2d(x, sd[p + "downsample.0.weight"], sd[p + "downsample.0.bias"], stride=stride)
        x = _norm(sd, p + "norm3", x, kind)   # downsample.1 IS norm3 (shared module, extractor.py:44-46)
    return F.relu(x + y)


def _bottleneck(sd, p, x, kind, stride):
    """core/extractor.py:60-116."""
    y = F.relu(_norm(sd, p + "norm1", F.conv2d(x, sd[p + "conv1.weight"], sd[p + "conv1.bias"]), kind))
    y = F.relu(_norm(sd, p + "norm2", F.conv2d(y, sd[p + "conv2.weight"], sd[p + "conv2.bias"], stride=stride, padding=1), kind))
    y = F.relu(_norm(sd, p + "norm3", F.conv2d(y, sd[p + "conv3.weight"], sd[p + "conv3.bias"]), kind))
    if stride != 1:
        x = F.conv2d(x, sd[p + "downsample.0.weight"], sd[p + "downsample.0.bias"], stride=stride)
        x = _norm(sd, p + "norm4", x, kind)   # downsample.1 IS norm4 (extractor.py:102-104)
    return F.relu(x + y)


def encoder(sd, p, x, kind, small):
    """BasicEncoder (core/extractor.py:118-192) / SmallEncoder (:195-267), eval mode."""
    blk = _bottleneck if small else _res_block
    x = F.conv2d(x, sd[p + "conv1.weight"], sd[p + "conv1.bias"], stride=2, padding=3)
    x = F.relu(_norm(sd, p + "norm1", x, kind))
    for li, stride in ((1, 1), (2, 2), (3, 2)):
        x = blk(sd, f"{p}layer{li}.0.", x, kind, stride)
        x = blk(sd, f"{p}layer{li}.1.", x, kind, 1)
    return F.conv2d(x, sd[p + "conv2.weight"], sd[p + "conv2.bias"])


# --------------------------------------------------------------------------
# the model loop.  core/raft.py:86-144
# --------------------------------------------------------------------------
def raft_forward(sd, image1, image2, iters=12, small=False, alternate_corr=False,
                 flow_init=None, test_mode=False, gma=False):
    """RAFT.forward (gma=True: RAFTGMA.forward, core/gma_network.py:72-129) with frozen BN and dropout 0."""
    hdim, cdim, radius = (96, 64, 3) if small else (128, 128, 4)
    image1 = (2 * (image1 / 255.0) - 1.0).contiguous()
    image2 = (2 * (image2 / 255.0) - 1.0).contiguous()
    B = image1.shape[0]
    fm = encoder(sd, "fnet.", torch.cat([image1, image2], 0), "instance", small)
    fmap1, fmap2 = fm[:B].float(), fm[B:].float()
    if not alternate_corr:
        pyr = corr_pyramid(fmap1, fmap2, 4)
    cnet = encoder(sd, "cnet.", image1, "none" if small else "batch", small)
    net = torch.tanh(cnet[:, :hdim])
    inp = torch.relu(cnet[:, hdim:hdim + cdim])
    attn = gma_attention(sd, "att.", inp) if gma else None
    H8, W8 = image1.shape[2] // 8, image1.shape[3] // 8
    coords0 = coords_grid(B, H8, W8, image1.device)
    coords1 = coords_grid(B, H8, W8, image1.device)
    if flow_init is not None:
        coords1 = coords1 + flow_init
    preds = []
    flow_up = None
    for _ in range(iters):
        coords1 = coords1.detach()
        if alternate_corr:
            corr = alt_corr_lookup(fmap1, fmap2, coords1, 4, radius)
        else:
            corr = corr_lookup(pyr, coords1, radius)
        flow = coords1 - coords0
        if gma:
            net, up_mask, delta = gma_update_block(sd, "update_block.", net, inp, corr, flow, attn)
        elif small:
            net, up_mask, delta = small_update_block(sd, "update_block.", net, inp, corr, flow)
        else:
            net, up_mask, delta = basic_update_block(sd, "update_block.", net, inp, corr, flow)
        coords1 = coords1 + delta
        if up_mask is None:
            flow_up = upflow8(coords1 - coords0)
        else:
            flow_up = upsample_flow(coords1 - coords0, up_mask)
        preds.append(flow_up)
    if test_mode:
        return coords1 - coords0, flow_up
    return preds


def sequence_loss_zero_gt(preds, gamma=0.8):
    """The bench objective: sum_i gamma^(n-1-i) * mean(sqrt(pred_i^2 + 1e-6)),
    the Charbonnier form of pytorch/train.py:60-96 against gt = 0 with every pixel valid."""
    n = len(preds)
    loss = 0.0
    for i, p in enumerate(preds):
        loss = loss + (gamma ** (n - i - 1)) * torch.sqrt(p * p + 1e-6).mean()
    return loss


def sequence_loss(flow_preds, flow_gt, valid, gamma=0.8, gamma2=1.0, max_flow=400.0):
    """pytorch/train.py:60-96 restated.  PARITY UNPINNED: pytorch/train.py cannot be imported in the build container (it
    needs cv2), so no golden vector exists for this function; the formula is short enough to read against the source."""
    nm = len(flow_preds)
    n = nm // 2
    mag = torch.sum(flow_gt ** 2, dim=1).sqrt()
    mask = (valid >= 0.5) & (mag < max_flow)
    loss = 0.0
    for i in range(nm):
        w = gamma ** (n - i - 1) if i < n else gamma2 ** (n - (i - n) - 1)
        diff = flow_preds[i] - flow_gt
        loss = loss + w * (mask[:, None] * (diff ** 2 + 0.001 ** 2) ** 0.5).mean()
    e = torch.sum((flow_preds[n - 1] - flow_gt) ** 2, dim=1).sqrt().view(-1)[(valid > 0.5).view(-1)]
    return loss, {"epe": e.mean().item(), "1px": (e < 1).float().mean().item(), "3px": (e < 3).float().mean().item(),
                  "5px": (e < 5).float().mean().item()}


def epe(a, b):
    """Mean end-point error, raft/metric.py:23-31."""
    return torch.sqrt(((a - b) ** 2).sum(dim=1)).mean()
