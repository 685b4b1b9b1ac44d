"""Per-primitive autograd functions over the gfx950 kernels, for *standalone* use of the module
classes (``AttentionModule(128)(feats, attn)`` etc.).  The network itself does not come through
here: ``NeuralModuleNetwork`` batches every primitive of a step into grouped launches
(``probnmn.runtime.engine``).  Same kernels, same arithmetic; one launch per primitive, n items
(the batch dimension) sharing one weight.

Tensors are the reference's NCHW; maps are converted to the kernels' NHWC once per call (free when
the tensor is already ``channels_last``, which is what these functions return).
"""
from typing import Optional

import numpy as np
import torch

from probnmn import _hip

C = _hip.CHANNELS
SHAPES = ((14, 14), (28, 28))  # feature-map sizes the conv kernels are built for


def _need(t: torch.Tensor, channels: int, what: str, like=None):
    """Validate a map argument; returns its (H, W).  ``like``: (H, W) it has to agree with."""
    if t.device.type != "cuda":
        raise _hip.HipLibraryError("%s is on %s: the HIP path needs a ROCm device (no CPU fallback)" % (what, t.device))
    if t.dim() != 4 or t.size(1) != channels or tuple(t.shape[2:]) not in SHAPES:
        raise NotImplementedError(
            "%s has shape %s; the gfx950 kernels are built for (n, %d, 14, 14) and (n, %d, 28, 28)"
            % (what, tuple(t.shape), channels, channels))
    hw = tuple(t.shape[2:])
    if like is not None and hw != tuple(like):
        raise ValueError("%s is %dx%d but the other operand is %dx%d" % ((what,) + hw + tuple(like)))
    return hw


def _nhwc(t: torch.Tensor) -> torch.Tensor:
    """(n,c,H,W) logical -> contiguous [n][H][W][c] storage."""
    return t.detach().float().permute(0, 2, 3, 1).contiguous()


def _nchw_view(buf: torch.Tensor, n: int, c: int, hw) -> torch.Tensor:
    return buf.view(n, hw[0], hw[1], c).permute(0, 3, 1, 2)


def _wcl(w: torch.Tensor) -> torch.Tensor:
    """[Cout,Cin,KH,KW] logical -> contiguous [Cout][KH*KW][Cin]."""
    return w.detach().float().permute(0, 2, 3, 1).contiguous()


def _launch(name: str, rec: np.ndarray, dev: torch.device, *args):
    buf = _hip.to_device(rec, dev)
    _hip.check(getattr(_hip.lib(), name)(buf.data_ptr(), len(rec), *args, _hip.stream_ptr(dev)), name)
    return buf


def _conv(dev, n, hw, inp, weight, bias, out, *, in2=None, mask=None, gate=None, dilation=1, cin_chunks=1, ntaps=9,
          relu=1, accumulate=False):
    H, W = hw
    HW = H * W
    rec = np.zeros(n, _hip.CONV_ITEM)
    step = HW * C * 4
    e = np.arange(n, dtype=np.int64)
    rec["in"] = inp.data_ptr() + e * step
    if in2 is not None:
        rec["in2"] = in2.data_ptr() + e * step
    if mask is not None:
        rec["mask"] = mask.data_ptr() + e * HW * 4
    if gate is not None:
        rec["gate"] = gate.data_ptr() + e * step
    rec["weight"] = weight.data_ptr()
    if bias is not None:
        rec["bias"] = bias.data_ptr()
    rec["out"] = out.data_ptr() + e * step
    rec["dilation"] = dilation
    rec["flags"] = 1 if accumulate else 0
    _launch("pnmn_conv_nhwc", rec, dev, H, W, cin_chunks, ntaps, C, C, 1, relu)


def _transpose(dev, w_cl: torch.Tensor, cout: int, cin: int, ntaps: int) -> torch.Tensor:
    wt = torch.empty(cin * ntaps * cout, dtype=torch.float32, device=dev)
    rec = np.zeros(1, _hip.WTRANS_ITEM)
    rec[0]["src"], rec[0]["dst"] = w_cl.data_ptr(), wt.data_ptr()
    rec[0]["cout"], rec[0]["cin"], rec[0]["ntaps"] = cout, cin, ntaps
    _launch("pnmn_transpose_weights", rec, dev)
    return wt


def _wgrad(dev, n, hw, x, dy, gate, dw, db, *, x2=None, xmask=None, dilation=1, ntaps=9, cin_blocks=1):
    H, W = hw
    HW = H * W
    items = np.zeros(n, _hip.WGRAD_ITEM)
    step = HW * C * 4
    e = np.arange(n, dtype=np.int64)
    items["x"] = x.data_ptr() + e * step
    if x2 is not None:
        items["x2"] = x2.data_ptr() + e * step
    if xmask is not None:
        items["xmask"] = xmask.data_ptr() + e * HW * 4
    items["dy"] = dy.data_ptr() + e * step
    items["gate"] = gate.data_ptr() + e * step
    items["dilation"] = dilation
    jobs = np.zeros(1, _hip.WGRAD_JOB)
    jobs[0]["dw"], jobs[0]["dbias"] = dw.data_ptr(), db.data_ptr()
    jobs[0]["item_begin"], jobs[0]["item_end"] = 0, n
    ibuf, jbuf = _hip.to_device(items, dev), _hip.to_device(jobs, dev)
    _hip.check(_hip.lib().pnmn_conv_wgrad(ibuf.data_ptr(), jbuf.data_ptr(), 1, H, W, ntaps, cin_blocks, 1, C, C,
                                          _hip.stream_ptr(dev)), "pnmn_conv_wgrad")


class _Conv3x3Relu(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, mask, dilation):
        hw = _need(x, C, "conv input")
        HW = hw[0] * hw[1]
        dev, n = x.device, x.size(0)
        xh, w = _nhwc(x), _wcl(weight)
        b = bias.detach().float().contiguous()
        m = None if mask is None else mask.detach().float().reshape(n, HW).contiguous()
        y = torch.empty(n, HW, C, dtype=torch.float32, device=dev)
        _conv(dev, n, hw, xh, w, b, y, mask=m, dilation=dilation)
        ctx.save_for_backward(xh, w, m if m is not None else torch.empty(0, device=dev), y)
        ctx.dilation, ctx.has_mask, ctx.hw = dilation, m is not None, hw
        return _nchw_view(y, n, C, hw)

    @staticmethod
    def backward(ctx, dy):
        xh, w, m, y = ctx.saved_tensors
        dev, n = xh.device, xh.size(0)
        hw = ctx.hw
        H, W = hw
        HW = H * W
        m = m if ctx.has_mask else None
        dyh = _nhwc(dy)
        wt = _transpose(dev, w, C, C, 9)
        dxm = torch.empty_like(xh)
        _conv(dev, n, hw, dyh, wt, None, dxm, gate=y, dilation=ctx.dilation, relu=0)
        dmask = None
        if m is not None:
            dx = torch.zeros_like(xh)
            dm = torch.zeros(n, HW, dtype=torch.float32, device=dev)
            rec = np.zeros(n, _hip.MASKBWD_ITEM)
            e = np.arange(n, dtype=np.int64)
            rec["dx"] = dxm.data_ptr() + e * HW * C * 4
            rec["feats"] = xh.data_ptr() + e * HW * C * 4
            rec["attn"] = m.data_ptr() + e * HW * 4
            rec["dfeats"] = dx.data_ptr() + e * HW * C * 4
            rec["dattn"] = dm.data_ptr() + e * HW * 4
            _launch("pnmn_mask_bwd", rec, dev, HW)
            dmask = dm.view(n, 1, H, W)
        else:
            dx = dxm
        dw = torch.zeros(C, 9, C, dtype=torch.float32, device=dev)
        db = torch.zeros(C, dtype=torch.float32, device=dev)
        _wgrad(dev, n, hw, xh, dyh, y, dw, db, xmask=m, dilation=ctx.dilation)
        return _nchw_view(dx, n, C, hw), dw.view(C, 3, 3, C).permute(0, 3, 1, 2), db, dmask, None


class _ProjectionRelu(torch.autograd.Function):
    @staticmethod
    def forward(ctx, in1, in2, weight, bias):
        hw = _need(in1, C, "projection input 1")
        _need(in2, C, "projection input 2", like=hw)
        dev, n = in1.device, in1.size(0)
        a, b2, w = _nhwc(in1), _nhwc(in2), _wcl(weight)
        b = bias.detach().float().contiguous()
        y = torch.empty(n, hw[0] * hw[1], C, dtype=torch.float32, device=dev)
        _conv(dev, n, hw, a, w, b, y, in2=b2, cin_chunks=2, ntaps=1)
        ctx.save_for_backward(a, b2, w, y)
        ctx.hw = hw
        return _nchw_view(y, n, C, hw)

    @staticmethod
    def backward(ctx, dy):
        a, b2, w, y = ctx.saved_tensors
        dev, n = a.device, a.size(0)
        hw = ctx.hw
        dyh = _nhwc(dy)
        wt = _transpose(dev, w, C, 2 * C, 1)  # [256][1][128]
        da, db2 = torch.empty_like(a), torch.empty_like(b2)
        _conv(dev, n, hw, dyh, wt[: C * C], None, da, gate=y, ntaps=1, relu=0)
        _conv(dev, n, hw, dyh, wt[C * C:], None, db2, gate=y, ntaps=1, relu=0)
        dw = torch.zeros(C, 1, 2 * C, dtype=torch.float32, device=dev)
        db = torch.zeros(C, dtype=torch.float32, device=dev)
        _wgrad(dev, n, hw, a, dyh, y, dw, db, x2=b2, ntaps=1, cin_blocks=2)
        return _nchw_view(da, n, C, hw), _nchw_view(db2, n, C, hw), dw.view(C, 2 * C, 1, 1), db


class _Dot1Sigmoid(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        hw = ctx.hw = _need(x, C, "head input")
        H, W = hw
        HW = H * W
        dev, n = x.device, x.size(0)
        xh = _nhwc(x)
        w = weight.detach().float().reshape(C).contiguous()
        b = bias.detach().float().reshape(1).contiguous()
        out = torch.empty(n, HW, dtype=torch.float32, device=dev)
        rec = ctx.rec = np.zeros(n, _hip.DOT1_ITEM)
        e = np.arange(n, dtype=np.int64)
        rec["in"], rec["w"], rec["b"] = xh.data_ptr() + e * HW * C * 4, w.data_ptr(), b.data_ptr()
        rec["out"] = out.data_ptr() + e * HW * 4
        _launch("pnmn_dot1_sigmoid_fwd", rec, dev, HW)
        ctx.save_for_backward(xh, w, b, out)
        return out.view(n, 1, H, W)

    @staticmethod
    def backward(ctx, dout):
        xh, w, b, out = ctx.saved_tensors
        dev, n = xh.device, xh.size(0)
        hw = ctx.hw
        HW = hw[0] * hw[1]
        do = dout.detach().float().reshape(n, HW).contiguous()
        din = torch.empty_like(xh)
        dw = torch.zeros(C, dtype=torch.float32, device=dev)
        db = torch.zeros(1, dtype=torch.float32, device=dev)
        rec = ctx.rec.copy()
        e = np.arange(n, dtype=np.int64)
        rec["dout"], rec["din"] = do.data_ptr() + e * HW * 4, din.data_ptr() + e * HW * C * 4
        rec["dw"], rec["db"] = dw.data_ptr(), db.data_ptr()
        _launch("pnmn_dot1_sigmoid_bwd", rec, dev, HW)
        return _nchw_view(din, n, C, hw), dw.view(1, C, 1, 1), db


class _Same(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feats, attn, weight, bias):
        hw = ctx.hw = _need(feats, C, "SameModule features")
        _need(attn, 1, "SameModule attention", like=hw)
        H, W = hw
        HW = H * W
        dev, n = feats.device, feats.size(0)
        fh = _nhwc(feats)
        a = attn.detach().float().reshape(n, HW).contiguous()
        w = weight.detach().float().reshape(C + 1).contiguous()
        b = bias.detach().float().reshape(1).contiguous()
        out = torch.empty(n, HW, dtype=torch.float32, device=dev)
        rec = ctx.rec = np.zeros(n, _hip.SAME_ITEM)
        e = np.arange(n, dtype=np.int64)
        rec["feats"], rec["attn"] = fh.data_ptr() + e * HW * C * 4, a.data_ptr() + e * HW * 4
        rec["w"], rec["b"], rec["out"] = w.data_ptr(), b.data_ptr(), out.data_ptr() + e * HW * 4
        _launch("pnmn_same_fwd", rec, dev, HW)
        ctx.save_for_backward(fh, a, w, b, out)
        return out.view(n, 1, H, W)

    @staticmethod
    def backward(ctx, dout):
        fh, a, w, b, out = ctx.saved_tensors
        dev, n = fh.device, fh.size(0)
        hw = ctx.hw
        H, W = hw
        HW = H * W
        do = dout.detach().float().reshape(n, HW).contiguous()
        dfe = torch.zeros_like(fh)
        dat = torch.zeros_like(a)
        dw = torch.zeros(C + 1, dtype=torch.float32, device=dev)
        db = torch.zeros(1, dtype=torch.float32, device=dev)
        rec = ctx.rec.copy()
        e = np.arange(n, dtype=np.int64)
        rec["dout"], rec["dfeats"] = do.data_ptr() + e * HW * 4, dfe.data_ptr() + e * HW * C * 4
        rec["dattn"], rec["dw"], rec["db"] = dat.data_ptr() + e * HW * 4, dw.data_ptr(), db.data_ptr()
        _launch("pnmn_same_bwd", rec, dev, HW)
        return _nchw_view(dfe, n, C, hw), dat.view(n, 1, H, W), dw.view(1, C + 1, 1, 1), db


class _MinMax(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, is_max):
        hw = None
        for t, what in ((a, "first operand"), (b, "second operand")):
            if t.dim() != 4 or t.size(1) not in (1, C):
                raise NotImplementedError("And/Or %s has %s channels; the kernels take 1 or 128" % (what, t.size(1)))
            hw = _need(t, t.size(1), "And/Or " + what, like=hw)
        H, W = hw
        HW = H * W
        if a.size(0) != b.size(0):
            raise NotImplementedError("And/Or operands must have the same batch size")
        dev, n = a.device, a.size(0)
        ac, bc = a.size(1), b.size(1)
        oc = max(ac, bc)
        ah = _nhwc(a) if ac == C else a.detach().float().reshape(n, HW).contiguous()
        bh = _nhwc(b) if bc == C else b.detach().float().reshape(n, HW).contiguous()
        out = torch.empty(n, HW * oc, dtype=torch.float32, device=dev)
        rec = ctx.rec = np.zeros(n, _hip.MINMAX_ITEM)
        e = np.arange(n, dtype=np.int64)
        rec["a"], rec["b"] = ah.data_ptr() + e * HW * ac * 4, bh.data_ptr() + e * HW * bc * 4
        rec["out"] = out.data_ptr() + e * HW * oc * 4
        rec["a_channels"], rec["b_channels"], rec["is_max"] = ac, bc, int(is_max)
        _launch("pnmn_minmax_fwd", rec, dev, HW, C)
        ctx.save_for_backward(ah, bh)
        ctx.dims = (n, ac, bc, oc, hw)
        return _nchw_view(out, n, oc, hw) if oc == C else out.view(n, 1, H, W)

    @staticmethod
    def backward(ctx, dout):
        ah, bh = ctx.saved_tensors
        n, ac, bc, oc, hw = ctx.dims
        H, W = hw
        HW = H * W
        dev = ah.device
        do = _nhwc(dout) if oc == C else dout.detach().float().reshape(n, HW).contiguous()
        da, db = torch.zeros_like(ah), torch.zeros_like(bh)
        rec = ctx.rec.copy()
        e = np.arange(n, dtype=np.int64)
        rec["dout"] = do.data_ptr() + e * HW * oc * 4
        rec["da"], rec["db"] = da.data_ptr() + e * HW * ac * 4, db.data_ptr() + e * HW * bc * 4
        _launch("pnmn_minmax_bwd", rec, dev, HW, C)
        ga = _nchw_view(da, n, C, hw) if ac == C else da.view(n, 1, H, W)
        gb = _nchw_view(db, n, C, hw) if bc == C else db.view(n, 1, H, W)
        return ga, gb, None


# ---- public functional surface -------------------------------------------------------------------
def conv3x3_relu(x, weight, bias, mask: Optional[torch.Tensor] = None, dilation: int = 1):
    """relu(conv2d(x * mask, weight, bias, padding=dilation, dilation=dilation)), 128 -> 128 channels."""
    if tuple(weight.shape) != (C, C, 3, 3):
        raise NotImplementedError("conv3x3 weight must be (128,128,3,3), got %s" % (tuple(weight.shape),))
    if mask is not None:
        _need(mask, 1, "attention", like=tuple(x.shape[2:]))
    return _Conv3x3Relu.apply(x, weight, bias, mask, dilation)


def projection_relu(in1, in2, weight, bias):
    """relu(conv1x1(cat([in1, in2], 1))), 256 -> 128 channels."""
    if tuple(weight.shape) != (C, 2 * C, 1, 1):
        raise NotImplementedError("projection weight must be (128,256,1,1), got %s" % (tuple(weight.shape),))
    return _ProjectionRelu.apply(in1, in2, weight, bias)


def dot1_sigmoid(x, weight, bias):
    """sigmoid(conv1x1(x)), 128 -> 1 channel."""
    return _Dot1Sigmoid.apply(x, weight, bias)


def same(feats, attn, weight, bias):
    """SameModule arithmetic (argmax gather, correlate, 129 -> 1 conv, sigmoid)."""
    return _Same.apply(feats, attn, weight, bias)


def minmax(a, b, is_max: bool):
    """torch.max / torch.min of two maps with 1 <-> 128 channel broadcast."""
    return _MinMax.apply(a, b, bool(is_max))
