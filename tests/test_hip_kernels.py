"""Every C-ABI kernel of libprobnmn_hip.so against plain PyTorch CPU fp32 (the torch ops the
reference calls at that site).  Tolerance: fp32 accumulation-order differences only --
rtol 2e-4 / atol 2e-4 on O(1) values (K up to 9216 products per output)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

H = W = 14
HW = H * W
C = 128
RT, AT = 2e-4, 2e-4


@pytest.fixture(autouse=True, params=[14, 28], ids=["14x14", "28x28"])
def map_size(request):
    """Every test below runs at both feature-map sizes the kernels are built for (BASELINE configs
    1-4: 14x14; config 5: 28x28, where a conv / weight-gradient workgroup covers a 7-row band)."""
    global H, W, HW
    H = W = request.param
    HW = H * W
    yield request.param
    H = W = 14
    HW = H * W


@pytest.fixture(scope="module")
def hip():
    from probnmn import _hip

    _hip.lib()
    assert torch.cuda.is_available(), "gpu tests need a MI355X"
    return _hip


def dev():
    return torch.device("cuda:0")


def nhwc(t):  # (n,C,H,W) cpu -> (n,HW,C) cuda contiguous
    return t.permute(0, 2, 3, 1).reshape(t.size(0), -1, t.size(1)).contiguous().to(dev())


def from_nhwc(t, n, c):  # (n,HW,c) cuda -> (n,c,H,W) cpu
    return t.reshape(n, H, W, c).permute(0, 3, 1, 2).contiguous().cpu()


def wcl(w):  # (cout,cin,kh,kw) -> [cout][taps][cin] cuda
    return w.permute(0, 2, 3, 1).reshape(w.size(0), -1, w.size(1)).contiguous().to(dev())


def ptr(t, off=0):
    return t.data_ptr() + 4 * off


def gen(seed):
    g = torch.Generator().manual_seed(seed)
    return g


def run(hip, name, recs, *args):
    d = dev()
    buf = hip.to_device(recs, d)
    code = getattr(hip.lib(), name)(buf.data_ptr(), len(recs), *args, hip.stream_ptr(d))
    hip.check(code, name)
    torch.cuda.synchronize()
    return buf


@pytest.mark.parametrize("dilation", [1, 2, 4, 8])
def test_conv3x3_masked_relu(hip, dilation):
    g = gen(dilation)
    n = 3
    x = torch.relu(torch.randn(n, C, H, W, generator=g))
    m = torch.sigmoid(torch.randn(n, 1, H, W, generator=g))
    ws = [torch.randn(C, C, 3, 3, generator=g) * (2.0 / (9 * C)) ** 0.5 for _ in range(n)]
    bs = [torch.randn(C, generator=g) * 0.1 for _ in range(n)]
    ref = torch.cat(
        [F.relu(F.conv2d(x[i : i + 1] * m[i : i + 1], ws[i], bs[i], padding=dilation, dilation=dilation)) for i in range(n)]
    )
    xd, md = nhwc(x), m.reshape(n, HW).to(dev())
    wd = [wcl(w) for w in ws]
    bd = [b.to(dev()) for b in bs]
    out = torch.full((n, HW, C), float("nan"), device=dev())
    recs = np.zeros(n, hip.CONV_ITEM)
    for i in range(n):
        recs[i]["in"] = ptr(xd[i])
        recs[i]["mask"] = ptr(md[i]) if i != 1 else 0
        recs[i]["weight"] = ptr(wd[i])
        recs[i]["bias"] = ptr(bd[i])
        recs[i]["out"] = ptr(out[i])
        recs[i]["dilation"] = dilation
    ref[1] = F.relu(F.conv2d(x[1:2], ws[1], bs[1], padding=dilation, dilation=dilation))[0]  # no mask
    run(hip, "pnmn_conv_nhwc", recs, H, W, 1, 9, C, C, 1, 1)
    torch.testing.assert_close(from_nhwc(out, n, C), ref, rtol=RT, atol=AT)


@pytest.mark.parametrize("n", [5, 17, 41, 129, 300, 520, 777])
def test_conv_many_items_every_launch_shape(hip, n):
    """Item counts around and beyond one round of 256 workgroups: the library cuts such launches into
    whole rounds with one K-split plus a remainder with a larger one, and maps the splits of an item to
    one XCD -- every item must still get exactly its own result."""
    g = gen(n)
    x = torch.relu(torch.randn(n, C, H, W, generator=g))
    m = torch.sigmoid(torch.randn(n, 1, H, W, generator=g))
    w = torch.randn(4, C, C, 3, 3, generator=g) * (2.0 / (9 * C)) ** 0.5
    b = torch.randn(4, C, generator=g) * 0.1
    xd, md = nhwc(x), m.reshape(n, HW).to(dev())
    wd = [wcl(w[k]) for k in range(4)]
    bd = [b[k].to(dev()) for k in range(4)]
    out = torch.full((n, HW, C), float("nan"), device=dev())
    recs = np.zeros(n, hip.CONV_ITEM)
    for i in range(n):
        recs[i]["in"], recs[i]["mask"] = ptr(xd[i]), ptr(md[i])
        recs[i]["weight"], recs[i]["bias"], recs[i]["out"] = ptr(wd[i % 4]), ptr(bd[i % 4]), ptr(out[i])
        recs[i]["dilation"] = 1 + (i % 2)
    run(hip, "pnmn_conv_nhwc", recs, H, W, 1, 9, C, C, 1, 1)
    xm = (x * m).to(dev())
    got = out.reshape(n, H, W, C).permute(0, 3, 1, 2)
    for k in range(4):
        for d in (1, 2):
            rows = [i for i in range(n) if i % 4 == k and 1 + (i % 2) == d]
            if rows:
                ref = F.relu(F.conv2d(xm[rows], w[k].to(dev()), b[k].to(dev()), padding=d, dilation=d))
                torch.testing.assert_close(got[rows], ref, rtol=RT, atol=AT)


def test_conv_transpose_detecting_identity(hip):
    """A = delta weights: out channel n copies in channel (n*7+3)%128 from tap (n%9): catches
    row/col or tap-order mix-ups that random data with loose tolerance could hide."""
    g = gen(3)
    x = torch.randn(1, C, H, W, generator=g)
    w = torch.zeros(C, C, 3, 3)
    for n in range(C):
        w[n, (n * 7 + 3) % C, (n % 9) // 3, (n % 9) % 3] = 1.0
    ref = F.conv2d(x, w, None, padding=1)
    xd, wd = nhwc(x), wcl(w)
    out = torch.empty(1, HW, C, device=dev())
    recs = np.zeros(1, hip.CONV_ITEM)
    recs[0]["in"], recs[0]["weight"], recs[0]["out"], recs[0]["dilation"] = ptr(xd), ptr(wd), ptr(out), 1
    run(hip, "pnmn_conv_nhwc", recs, H, W, 1, 9, C, C, 1, 0)
    assert torch.equal(from_nhwc(out, 1, C), ref)


def test_conv_stem_chunks_and_two_sources(hip):
    g = gen(11)
    n = 2
    # stem-like: 1024 input channels = 8 chunks, pixel stride 1024
    x = torch.relu(torch.randn(n, 1024, H, W, generator=g))
    w = torch.randn(C, 1024, 3, 3, generator=g) * (2.0 / (9 * 1024)) ** 0.5
    b = torch.randn(C, generator=g) * 0.1
    ref = F.relu(F.conv2d(x, w, b, padding=1))
    xd, wd, bd = nhwc(x), wcl(w), b.to(dev())
    out = torch.empty(n, HW, C, device=dev())
    recs = np.zeros(n, hip.CONV_ITEM)
    for i in range(n):
        recs[i]["in"], recs[i]["weight"], recs[i]["bias"], recs[i]["out"] = ptr(xd[i]), ptr(wd), ptr(bd), ptr(out[i])
        recs[i]["dilation"] = 1
    run(hip, "pnmn_conv_nhwc", recs, H, W, 8, 9, 1024, C, 1, 1)
    torch.testing.assert_close(from_nhwc(out, n, C), ref, rtol=RT, atol=AT)

    # projection: 1x1 over cat(in1, in2)
    a = torch.relu(torch.randn(n, C, H, W, generator=g))
    c = torch.relu(torch.randn(n, C, H, W, generator=g))
    wp = torch.randn(C, 2 * C, 1, 1, generator=g) * (1.0 / (2 * C)) ** 0.5
    bp = torch.randn(C, generator=g) * 0.1
    refp = F.relu(F.conv2d(torch.cat([a, c], 1), wp, bp))
    ad, cd, wpd, bpd = nhwc(a), nhwc(c), wcl(wp), bp.to(dev())
    outp = torch.empty(n, HW, C, device=dev())
    recs = np.zeros(n, hip.CONV_ITEM)
    for i in range(n):
        recs[i]["in"], recs[i]["in2"] = ptr(ad[i]), ptr(cd[i])
        recs[i]["weight"], recs[i]["bias"], recs[i]["out"] = ptr(wpd), ptr(bpd), ptr(outp[i])
    run(hip, "pnmn_conv_nhwc", recs, H, W, 2, 1, C, C, 1, 1)
    torch.testing.assert_close(from_nhwc(outp, n, C), refp, rtol=RT, atol=AT)

    # classifier conv1x1 128 -> 1024 (8 output blocks, pixel stride 1024)
    wc = torch.randn(1024, C, 1, 1, generator=g) * (1.0 / C) ** 0.5
    bc = torch.randn(1024, generator=g) * 0.1
    refc = F.relu(F.conv2d(a, wc, bc))
    wcd, bcd = wcl(wc), bc.to(dev())
    outc = torch.empty(n, HW, 1024, device=dev())
    recs = np.zeros(n, hip.CONV_ITEM)
    for i in range(n):
        recs[i]["in"], recs[i]["weight"], recs[i]["bias"], recs[i]["out"] = ptr(ad[i]), ptr(wcd), ptr(bcd), ptr(outc[i])
    run(hip, "pnmn_conv_nhwc", recs, H, W, 1, 1, C, 1024, 8, 1)
    torch.testing.assert_close(from_nhwc(outc, n, 1024), refc, rtol=RT, atol=AT)


@pytest.mark.parametrize("n", [3, 41, 300, 520])
def test_conv1x1_many_items_every_launch_shape(hip, n):
    """1x1 convolutions over two sources (the ComparisonModule projection) and into four output blocks (the
    classifier's 128 -> 512) at item counts that take every split of the launch planner; delta weights make the
    comparison exact (a tap / channel / block mix-up cannot hide in a tolerance)."""
    g = gen(n)
    a = torch.relu(torch.randn(n, C, H, W, generator=g))
    c = torch.relu(torch.randn(n, C, H, W, generator=g))
    m = torch.sigmoid(torch.randn(n, 1, H, W, generator=g))
    wp = torch.randn(3, C, 2 * C, 1, 1, generator=g) * (1.0 / (2 * C)) ** 0.5
    bp = torch.randn(3, C, generator=g) * 0.1
    ad, cd, md = nhwc(a), nhwc(c), m.reshape(n, HW).to(dev())
    wpd = [wcl(wp[k]) for k in range(3)]
    bpd = [bp[k].to(dev()) for k in range(3)]
    outp = torch.full((n, HW, C), float("nan"), device=dev())
    recs = np.zeros(n, hip.CONV_ITEM)
    for i in range(n):
        recs[i]["in"], recs[i]["in2"] = ptr(ad[i]), ptr(cd[i])
        recs[i]["weight"], recs[i]["bias"], recs[i]["out"] = ptr(wpd[i % 3]), ptr(bpd[i % 3]), ptr(outp[i])
        if i % 2:
            recs[i]["mask"] = ptr(md[i])
    run(hip, "pnmn_conv_nhwc", recs, H, W, 2, 1, C, C, 1, 1)
    got = from_nhwc(outp, n, C)
    for k in range(3):
        for masked in (0, 1):
            rows = [i for i in range(n) if i % 3 == k and i % 2 == masked]
            if rows:
                x = torch.cat([a[rows], c[rows]], 1) * (m[rows] if masked else 1.0)
                ref = F.relu(F.conv2d(x.to(dev()), wp[k].to(dev()), bp[k].to(dev()))).cpu()
                torch.testing.assert_close(got[rows], ref, rtol=RT, atol=AT)

    # 128 -> 512 with a permutation matrix: out channel o copies in channel (o * 37 + 5) % 128, exactly
    wc = torch.zeros(4 * C, C, 1, 1)
    for o in range(4 * C):
        wc[o, (o * 37 + 5) % C, 0, 0] = 1.0
    wcd = wcl(wc)
    outc = torch.full((n, HW, 4 * C), float("nan"), device=dev())
    recs = np.zeros(n, hip.CONV_ITEM)
    for i in range(n):
        recs[i]["in"], recs[i]["weight"], recs[i]["out"] = ptr(ad[i]), ptr(wcd), ptr(outc[i])
    run(hip, "pnmn_conv_nhwc", recs, H, W, 1, 1, C, 4 * C, 4, 0)
    assert torch.equal(from_nhwc(outc, n, 4 * C).cpu(), F.conv2d(a, wc))


@pytest.mark.parametrize("dilation", [1, 2, 4, 8])
def test_conv_dgrad_and_wgrad_match_autograd(hip, dilation):
    g = gen(20 + dilation)
    n = 5
    x = torch.relu(torch.randn(n, C, H, W, generator=g))
    m = torch.sigmoid(torch.randn(n, 1, H, W, generator=g))
    w = (torch.randn(C, C, 3, 3, generator=g) * (2.0 / (9 * C)) ** 0.5).requires_grad_(True)
    b = (torch.randn(C, generator=g) * 0.1).requires_grad_(True)
    xin = (x * m).requires_grad_(True)
    y = F.relu(F.conv2d(xin, w, b, padding=dilation, dilation=dilation))
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)

    # dgrad = conv of the gated dy with the transposed, tap-reversed weight
    wd = wcl(w.detach())
    wt = torch.empty(C, 9, C, device=dev())
    t = np.zeros(1, hip.WTRANS_ITEM)
    t[0]["src"], t[0]["dst"], t[0]["cout"], t[0]["cin"], t[0]["ntaps"] = ptr(wd), ptr(wt), C, C, 9
    run(hip, "pnmn_transpose_weights", t)
    dyd, yd = nhwc(dy), nhwc(y.detach())
    dx = torch.full((n, HW, C), float("nan"), device=dev())
    recs = np.zeros(n, hip.CONV_ITEM)
    for i in range(n):
        recs[i]["in"], recs[i]["gate"], recs[i]["weight"], recs[i]["out"] = ptr(dyd[i]), ptr(yd[i]), ptr(wt), ptr(dx[i])
        recs[i]["dilation"] = dilation
    run(hip, "pnmn_conv_nhwc", recs, H, W, 1, 9, C, C, 1, 0)
    torch.testing.assert_close(from_nhwc(dx, n, C), xin.grad, rtol=RT, atol=AT)
    # accumulate flag
    recs["flags"] = hip.CONV_ACCUMULATE
    run(hip, "pnmn_conv_nhwc", recs, H, W, 1, 9, C, C, 1, 0)
    torch.testing.assert_close(from_nhwc(dx, n, C), 2 * xin.grad, rtol=RT, atol=2 * AT)

    # wgrad: two jobs (3 + 2 items) into the same dw / dbias
    xd, md = nhwc(x), m.reshape(n, HW).to(dev())
    items = np.zeros(n, hip.WGRAD_ITEM)
    for i in range(n):
        items[i]["x"], items[i]["xmask"], items[i]["dy"], items[i]["gate"] = ptr(xd[i]), ptr(md[i]), ptr(dyd[i]), ptr(yd[i])
        items[i]["dilation"] = dilation
    dw = torch.zeros(C, 9, C, device=dev())
    db = torch.zeros(C, device=dev())
    jobs = np.zeros(2, hip.WGRAD_JOB)
    jobs["dw"], jobs["dbias"] = ptr(dw), ptr(db)
    jobs[0]["item_begin"], jobs[0]["item_end"] = 0, 3
    jobs[1]["item_begin"], jobs[1]["item_end"] = 3, 5
    ibuf = hip.to_device(items, dev())
    jbuf = hip.to_device(jobs, dev())
    hip.check(hip.lib().pnmn_conv_wgrad(ibuf.data_ptr(), jbuf.data_ptr(), 2, H, W, 9, 1, 1, C, C, hip.stream_ptr(dev())), "wgrad")
    torch.cuda.synchronize()
    ref_dw = w.grad.permute(0, 2, 3, 1).reshape(C, 9, C)
    torch.testing.assert_close(dw.cpu(), ref_dw, rtol=RT, atol=5 * AT)
    torch.testing.assert_close(db.cpu(), b.grad, rtol=RT, atol=5 * AT)
    # the same launch with a CU budget (pnmn_conv_wgrad_cus: 3 workgroups walk the 2 jobs x 2 slabs; the banded 28x28
    # kernel ignores the budget): same sums
    dw2, db2 = torch.zeros_like(dw), torch.zeros_like(db)
    jobs["dw"], jobs["dbias"] = ptr(dw2), ptr(db2)
    jbuf2 = hip.to_device(jobs, dev())
    hip.check(hip.lib().pnmn_conv_wgrad_cus(ibuf.data_ptr(), jbuf2.data_ptr(), 2, H, W, 9, 1, 1, C, C, 3 if H == 14 else 0,
                                            hip.stream_ptr(dev())), "wgrad (budget)")
    torch.cuda.synchronize()
    torch.testing.assert_close(dw2, dw, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(db2, db, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("sole", [0, 1])
@pytest.mark.parametrize("dilation", [1, 2])
def test_conv_dgrad_fused_mask_backward(hip, sole, dilation):
    """Data-gradient of a conv whose forward input was feats * attn, with the mask backward fused into the
    epilogue (PNMN_CONV_MASKBWD; atomics, or plain read-modify-write with PNMN_CONV_MB_SOLE)."""
    g = gen(300 + sole + 2 * dilation)
    n = 3
    feats = torch.relu(torch.randn(n, C, H, W, generator=g)).requires_grad_(True)
    attn = torch.sigmoid(torch.randn(n, 1, H, W, generator=g))
    attn[1] = 1.0  # item 1: the all-ones attention `scene` produces (mb_attn == NULL)
    attn.requires_grad_(True)
    w = torch.randn(C, C, 3, 3, generator=g) * (2.0 / (9 * C)) ** 0.5
    b = torch.randn(C, generator=g) * 0.1
    y = F.relu(F.conv2d(feats * attn, w, b, padding=dilation, dilation=dilation))
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    wd = wcl(w)
    wt = torch.empty(C, 9, C, device=dev())
    t = np.zeros(1, hip.WTRANS_ITEM)
    t[0]["src"], t[0]["dst"], t[0]["cout"], t[0]["cin"], t[0]["ntaps"] = ptr(wd), ptr(wt), C, C, 9
    run(hip, "pnmn_transpose_weights", t)
    dyd, yd, fd = nhwc(dy), nhwc(y.detach()), nhwc(feats.detach())
    ad = attn.detach().reshape(n, HW).to(dev())
    prior = torch.randn(n, HW, C, generator=g)  # dFEAT already holds other consumers' gradients
    dfe = prior.clone().to(dev())
    dat = torch.zeros(n, HW, device=dev())
    recs = np.zeros(n, hip.CONV_ITEM)
    for i in range(n):
        recs[i]["in"], recs[i]["gate"], recs[i]["weight"] = ptr(dyd[i]), ptr(yd[i]), ptr(wt)
        recs[i]["dilation"] = dilation
        recs[i]["flags"] = hip.CONV_MASKBWD | (8 if sole else 0)
        recs[i]["mb_feats"], recs[i]["mb_dfeats"] = ptr(fd[i]), ptr(dfe[i])
        if i != 1:
            recs[i]["mb_attn"], recs[i]["mb_dattn"] = ptr(ad[i]), ptr(dat[i])
    run(hip, "pnmn_conv_nhwc", recs, H, W, 1, 9, C, C, 1, 0)
    torch.testing.assert_close(from_nhwc(dfe, n, C) - from_nhwc(prior.to(dev()), n, C), feats.grad, rtol=RT, atol=AT)
    got = dat.cpu().reshape(n, 1, H, W)
    keep = [0, 2]
    torch.testing.assert_close(got[keep], attn.grad[keep], rtol=RT, atol=20 * AT)  # sums of 128 products of O(1) terms
    assert float(got[1].abs().max()) == 0.0


@pytest.mark.parametrize("n", [3, 37])
def test_conv_dgrad_dattn_epilogue_and_deferred_feature_gradient(hip, n):
    """The default backward of `feats * attn` in front of a masked conv: the data gradient stores dx to its own map and
    fuses only d(attention) (PNMN_CONV_DATTN); pnmn_feat_grad_gather then adds, per example, every masked conv's
    dx * attn into d(feats) in one pass.  Two masked convs share example 0's d(feats) map."""
    g = gen(900 + n)
    owner = [0, 0] + list(range(1, n - 1))  # conv i reads / adds into example owner[i]
    n_ex = max(owner) + 1
    feats = torch.relu(torch.randn(n_ex, C, H, W, generator=g)).requires_grad_(True)
    attn = torch.sigmoid(torch.randn(n, 1, H, W, generator=g))
    attn[2] = 1.0  # the all-ones attention `scene` produces (attn == NULL: no d(attention), plain dx)
    attn.requires_grad_(True)
    w = torch.randn(C, C, 3, 3, generator=g) * (2.0 / (9 * C)) ** 0.5
    y = F.relu(F.conv2d(feats[owner] * attn, w, None, padding=1))
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    wt = torch.empty(C, 9, C, device=dev())
    t = np.zeros(1, hip.WTRANS_ITEM)
    wd = wcl(w)
    t[0]["src"], t[0]["dst"], t[0]["cout"], t[0]["cin"], t[0]["ntaps"] = ptr(wd), ptr(wt), C, C, 9
    run(hip, "pnmn_transpose_weights", t)
    dyd, yd, fd = nhwc(dy), nhwc(y.detach()), nhwc(feats.detach())
    ad = attn.detach().reshape(n, HW).to(dev())
    dx = torch.full((n, HW, C), float("nan"), device=dev())  # every element must be written
    dat = torch.zeros(n, HW, device=dev())
    prior = torch.randn(n_ex, HW, C, generator=g)  # d(feats) already holds other consumers' gradients
    gfeat = prior.clone().to(dev())
    recs = np.zeros(n, hip.CONV_ITEM)
    gather = np.zeros(n, hip.MASKBWD_ITEM)
    for i in range(n):
        recs[i]["in"], recs[i]["gate"], recs[i]["weight"], recs[i]["out"] = ptr(dyd[i]), ptr(yd[i]), ptr(wt), ptr(dx[i])
        recs[i]["dilation"] = 1
        gather[i]["dx"], gather[i]["dfeats"] = ptr(dx[i]), ptr(gfeat[owner[i]])
        if i != 2:
            recs[i]["flags"] = hip.CONV_DATTN
            recs[i]["mb_feats"], recs[i]["mb_attn"], recs[i]["mb_dattn"] = ptr(fd[owner[i]]), ptr(ad[i]), ptr(dat[i])
            gather[i]["attn"] = ptr(ad[i])
    run(hip, "pnmn_conv_nhwc", recs, H, W, 1, 9, C, C, 1, 0)
    got = dat.cpu().reshape(n, 1, H, W)
    keep = [i for i in range(n) if i != 2]
    torch.testing.assert_close(got[keep], attn.grad[keep], rtol=RT, atol=20 * AT)
    assert float(got[2].abs().max()) == 0.0 and not bool(torch.isnan(dx).any())
    order = np.argsort(gather["dfeats"], kind="stable")
    buf = hip.to_device(gather[order], dev())
    hip.check(hip.lib().pnmn_feat_grad_gather(buf.data_ptr(), gfeat.data_ptr(), n, n_ex, HW, hip.stream_ptr(dev())), "gather")
    torch.cuda.synchronize()
    torch.testing.assert_close(from_nhwc(gfeat, n_ex, C) - from_nhwc(prior.to(dev()), n_ex, C), feats.grad, rtol=RT, atol=2 * AT)


def test_wgrad_1x1_two_sources_and_wide_output(hip):
    g = gen(31)
    n = 4
    a = torch.relu(torch.randn(n, C, H, W, generator=g))
    c = torch.relu(torch.randn(n, C, H, W, generator=g))
    wp = (torch.randn(C, 2 * C, 1, 1, generator=g) * 0.05).requires_grad_(True)
    y = F.relu(F.conv2d(torch.cat([a, c], 1), wp))
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    ad, cd, dyd, yd = nhwc(a), nhwc(c), nhwc(dy), nhwc(y.detach())
    items = np.zeros(n, hip.WGRAD_ITEM)
    for i in range(n):
        items[i]["x"], items[i]["x2"], items[i]["dy"], items[i]["gate"] = ptr(ad[i]), ptr(cd[i]), ptr(dyd[i]), ptr(yd[i])
    dw = torch.zeros(C, 1, 2 * C, device=dev())
    jobs = np.zeros(1, hip.WGRAD_JOB)
    jobs[0]["dw"], jobs[0]["item_begin"], jobs[0]["item_end"] = ptr(dw), 0, n
    ibuf, jbuf = hip.to_device(items, dev()), hip.to_device(jobs, dev())
    hip.check(hip.lib().pnmn_conv_wgrad(ibuf.data_ptr(), jbuf.data_ptr(), 1, H, W, 1, 2, 1, C, C, hip.stream_ptr(dev())), "wgrad")
    torch.cuda.synchronize()
    torch.testing.assert_close(dw.cpu().reshape(C, 2 * C), wp.grad.reshape(C, 2 * C), rtol=RT, atol=5 * AT)

    # 128 -> 1024 (classifier conv): 8 output blocks, dy pixel stride 1024, no gate
    wc = (torch.randn(1024, C, 1, 1, generator=g) * 0.05).requires_grad_(True)
    bc = torch.zeros(1024, requires_grad=True)
    y = F.conv2d(a, wc, bc)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    dyd = nhwc(dy)
    items = np.zeros(n, hip.WGRAD_ITEM)
    for i in range(n):
        items[i]["x"], items[i]["dy"] = ptr(ad[i]), ptr(dyd[i])
    dw = torch.zeros(1024, 1, C, device=dev())
    db = torch.zeros(1024, device=dev())
    jobs = np.zeros(1, hip.WGRAD_JOB)
    jobs[0]["dw"], jobs[0]["dbias"], jobs[0]["item_begin"], jobs[0]["item_end"] = ptr(dw), ptr(db), 0, n
    ibuf, jbuf = hip.to_device(items, dev()), hip.to_device(jobs, dev())
    hip.check(hip.lib().pnmn_conv_wgrad(ibuf.data_ptr(), jbuf.data_ptr(), 1, H, W, 1, 1, 8, C, 1024, hip.stream_ptr(dev())), "wgrad")
    torch.cuda.synchronize()
    torch.testing.assert_close(dw.cpu().reshape(1024, C), wc.grad.reshape(1024, C), rtol=RT, atol=5 * AT)
    torch.testing.assert_close(db.cpu(), bc.grad, rtol=RT, atol=5 * AT)

    # the GEMM kernel's work split (round 5): three jobs of 5 + 1 + 3 items, the last one adding into ANOTHER weight, with a
    # ReLU gate and an attention mask -- two workgroups per output block whose stage ranges begin and end inside items and
    # cross the job with the other weight (a flush in the middle of a range)
    n = 9
    a = torch.relu(torch.randn(n, C, H, W, generator=g))
    m = torch.sigmoid(torch.randn(n, 1, H, W, generator=g))
    w1 = (torch.randn(512, C, 1, 1, generator=g) * 0.05).requires_grad_(True)
    w2 = (torch.randn(512, C, 1, 1, generator=g) * 0.05).requires_grad_(True)
    b1, b2 = torch.zeros(512, requires_grad=True), torch.zeros(512, requires_grad=True)
    xin = a * m
    y = torch.cat((F.relu(F.conv2d(xin[:6], w1, b1)), F.relu(F.conv2d(xin[6:], w2, b2))), 0)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    ad, md, dyd, yd = nhwc(a), m.reshape(n, HW).to(dev()), nhwc(dy), nhwc(y.detach())
    items = np.zeros(n, hip.WGRAD_ITEM)
    for i in range(n):
        items[i]["x"], items[i]["xmask"], items[i]["dy"], items[i]["gate"] = ptr(ad[i]), ptr(md[i]), ptr(dyd[i]), ptr(yd[i])
    dws = [torch.zeros(512, 1, C, device=dev()) for _ in range(2)]
    dbs = [torch.zeros(512, device=dev()) for _ in range(2)]
    jobs = np.zeros(3, hip.WGRAD_JOB)
    for j, (lo, hi, k) in enumerate(((0, 5, 0), (5, 6, 0), (6, 9, 1))):
        jobs[j]["dw"], jobs[j]["dbias"], jobs[j]["item_begin"], jobs[j]["item_end"] = ptr(dws[k]), ptr(dbs[k]), lo, hi
    ibuf, jbuf = hip.to_device(items, dev()), hip.to_device(jobs, dev())
    hip.check(hip.lib().pnmn_conv_wgrad(ibuf.data_ptr(), jbuf.data_ptr(), 3, H, W, 1, 1, 4, C, 512, hip.stream_ptr(dev())), "wgrad")
    torch.cuda.synchronize()
    for k, (wk, bk) in enumerate(((w1, b1), (w2, b2))):
        torch.testing.assert_close(dws[k].cpu().reshape(512, C), wk.grad.reshape(512, C), rtol=RT, atol=5 * AT)
        torch.testing.assert_close(dbs[k].cpu(), bk.grad, rtol=RT, atol=5 * AT)


def test_dot1_sigmoid_fwd_bwd(hip):
    g = gen(40)
    n = 3
    x = torch.relu(torch.randn(n, C, H, W, generator=g)).requires_grad_(True)
    w = (torch.randn(1, C, 1, 1, generator=g) * 0.2).requires_grad_(True)
    b = torch.randn(1, generator=g).requires_grad_(True)
    y = torch.sigmoid(F.conv2d(x, w, b))
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    xd, wd, bd = nhwc(x.detach()), w.detach().reshape(C).to(dev()), b.detach().to(dev())
    out = torch.empty(n, HW, device=dev())
    din = torch.empty(n, HW, C, device=dev())
    dw = torch.zeros(C, device=dev())
    db = torch.zeros(1, device=dev())
    dyd = dy.reshape(n, HW).to(dev())
    recs = np.zeros(n, hip.DOT1_ITEM)
    for i in range(n):
        recs[i]["in"], recs[i]["w"], recs[i]["b"], recs[i]["out"] = ptr(xd[i]), ptr(wd), ptr(bd), ptr(out[i])
        recs[i]["dout"], recs[i]["din"], recs[i]["dw"], recs[i]["db"] = ptr(dyd[i]), ptr(din[i]), ptr(dw), ptr(db)
    run(hip, "pnmn_dot1_sigmoid_fwd", recs, HW)
    torch.testing.assert_close(out.cpu().reshape(n, 1, H, W), y.detach(), rtol=1e-5, atol=1e-6)
    run(hip, "pnmn_dot1_sigmoid_bwd", recs, HW)
    torch.testing.assert_close(from_nhwc(din, n, C), x.grad, rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(dw.cpu(), w.grad.reshape(C), rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(db.cpu(), b.grad, rtol=1e-4, atol=1e-5)


def test_same_fwd_bwd(hip):
    from oracle import nmn_oracle

    g = gen(50)
    n = 3
    feats = torch.relu(torch.randn(n, C, H, W, generator=g)).requires_grad_(True)
    attn = torch.sigmoid(torch.randn(n, 1, H, W, generator=g) * 2).requires_grad_(True)
    with torch.no_grad():
        attn[1, 0, 3, 5] = attn[1, 0, 9, 2] = 0.999  # tie: first maximum must win
    sd = {"s.conv.weight": (torch.randn(1, C + 1, 1, 1, generator=g) * 0.1).requires_grad_(True),
          "s.conv.bias": torch.randn(1, generator=g).requires_grad_(True)}
    ys = [nmn_oracle.same_module(sd, "s", feats[i : i + 1], attn[i : i + 1]) for i in range(n)]
    y = torch.cat(ys)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    fd, ad = nhwc(feats.detach()), attn.detach().reshape(n, HW).to(dev())
    wd, bd = sd["s.conv.weight"].detach().reshape(C + 1).to(dev()), sd["s.conv.bias"].detach().to(dev())
    out = torch.empty(n, HW, device=dev())
    dfe = torch.zeros(n, HW, C, device=dev())
    dat = torch.zeros(n, HW, device=dev())
    dw = torch.zeros(C + 1, device=dev())
    db = torch.zeros(1, device=dev())
    dyd = dy.reshape(n, HW).to(dev())
    recs = np.zeros(n, hip.SAME_ITEM)
    for i in range(n):
        recs[i]["feats"], recs[i]["attn"], recs[i]["w"], recs[i]["b"], recs[i]["out"] = ptr(fd[i]), ptr(ad[i]), ptr(wd), ptr(bd), ptr(out[i])
        recs[i]["dout"], recs[i]["dfeats"], recs[i]["dattn"], recs[i]["dw"], recs[i]["db"] = ptr(dyd[i]), ptr(dfe[i]), ptr(dat[i]), ptr(dw), ptr(db)
    run(hip, "pnmn_same_fwd", recs, HW)
    torch.testing.assert_close(out.cpu().reshape(n, 1, H, W), y.detach(), rtol=1e-5, atol=1e-6)
    run(hip, "pnmn_same_bwd", recs, HW)
    torch.testing.assert_close(from_nhwc(dfe, n, C), feats.grad, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(dat.cpu().reshape(n, 1, H, W), attn.grad, rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(dw.cpu(), sd["s.conv.weight"].grad.reshape(C + 1), rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(db.cpu(), sd["s.conv.bias"].grad, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("ac,bc", [(1, 1), (1, C), (C, 1), (C, C)])
@pytest.mark.parametrize("is_max", [0, 1])
def test_minmax_fwd_bwd(hip, ac, bc, is_max):
    g = gen(60 + ac + 2 * bc + is_max)
    a = torch.rand(1, ac, H, W, generator=g)
    b = torch.rand(1, bc, H, W, generator=g)
    a[0, 0, 0, :4] = 1.0
    b[0, 0, 0, :4] = 1.0  # ties -> gradient halves
    a.requires_grad_(True)
    b.requires_grad_(True)
    y = torch.max(a, b) if is_max else torch.min(a, b)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    oc = max(ac, bc)

    def lay(t, ch):
        return nhwc(t.detach())[0] if ch == C else t.detach().reshape(HW).to(dev())

    adv, bdv, dyd = lay(a, ac), lay(b, bc), lay(dy, oc)
    out = torch.empty(HW * oc, device=dev())
    da = torch.zeros(HW * ac, device=dev())
    dbb = torch.zeros(HW * bc, device=dev())
    recs = np.zeros(1, hip.MINMAX_ITEM)
    recs[0]["a"], recs[0]["b"], recs[0]["out"], recs[0]["dout"] = ptr(adv), ptr(bdv), ptr(out), ptr(dyd)
    recs[0]["da"], recs[0]["db"] = ptr(da), ptr(dbb)
    recs[0]["a_channels"], recs[0]["b_channels"], recs[0]["is_max"] = ac, bc, is_max
    run(hip, "pnmn_minmax_fwd", recs, HW, C)
    run(hip, "pnmn_minmax_bwd", recs, HW, C)

    def back(t, ch):
        return from_nhwc(t.reshape(1, HW, C), 1, C) if ch == C else t.cpu().reshape(1, 1, H, W)

    assert torch.equal(back(out, oc), y.detach())
    torch.testing.assert_close(back(da, ac), a.grad, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(back(dbb, bc), b.grad, rtol=1e-5, atol=1e-5)


def test_mask_bwd_and_accumulate(hip):
    g = gen(70)
    feats = torch.randn(1, C, H, W, generator=g).requires_grad_(True)
    attn = torch.rand(1, 1, H, W, generator=g).requires_grad_(True)
    x = feats * attn.repeat(1, C, 1, 1)
    dx = torch.randn(x.shape, generator=g)
    x.backward(dx)
    dxd, fd, ad = nhwc(dx)[0], nhwc(feats.detach())[0], attn.detach().reshape(HW).to(dev())
    dfe = torch.zeros(HW, C, device=dev())
    dat = torch.zeros(HW, device=dev())
    recs = np.zeros(2, hip.MASKBWD_ITEM)
    recs[0]["dx"], recs[0]["feats"], recs[0]["attn"], recs[0]["dfeats"], recs[0]["dattn"] = ptr(dxd), ptr(fd), ptr(ad), ptr(dfe), ptr(dat)
    dfe2 = torch.zeros(HW, C, device=dev())
    recs[1]["dx"], recs[1]["dfeats"] = ptr(dxd), ptr(dfe2)  # attn == NULL: all-ones attention
    run(hip, "pnmn_mask_bwd", recs, HW)
    torch.testing.assert_close(from_nhwc(dfe.reshape(1, HW, C), 1, C), feats.grad, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(dat.cpu().reshape(1, 1, H, W), attn.grad, rtol=1e-4, atol=1e-5)
    assert torch.equal(dfe2, dxd)
    ax = np.zeros(1, hip.AXPY_ITEM)
    ax[0]["src"], ax[0]["dst"], ax[0]["n"] = ptr(dxd), ptr(dfe2), HW * C
    run(hip, "pnmn_accumulate", ax)
    assert torch.equal(dfe2, 2 * dxd)


def test_layout_roundtrip(hip):
    g = gen(80)
    x = torch.randn(3, 1024, H, W, generator=g)
    xd = x.to(dev())
    y = torch.empty(3, HW, 1024, device=dev())
    hip.check(hip.lib().pnmn_nchw_to_nhwc(xd.data_ptr(), y.data_ptr(), 3, 1024, HW, hip.stream_ptr(dev())), "nchw_to_nhwc")
    assert torch.equal(y.cpu(), x.permute(0, 2, 3, 1).reshape(3, HW, 1024))
    z = torch.empty_like(xd)
    hip.check(hip.lib().pnmn_nhwc_to_nchw(y.data_ptr(), z.data_ptr(), 3, 1024, HW, hip.stream_ptr(dev())), "nhwc_to_nchw")
    assert torch.equal(z.cpu(), x)


def test_maxpool_flatten_fwd_bwd(hip):
    g = gen(90)
    n, CC = 3, 1024
    pre = torch.randn(n, CC, H, W, generator=g).requires_grad_(True)
    act = F.relu(pre)
    y = F.max_pool2d(act, 2, 2).reshape(n, -1)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    actd = nhwc(act.detach())
    out = torch.empty(n, CC * (H // 2) * (W // 2), device=dev())
    hip.check(hip.lib().pnmn_maxpool2_flatten_fwd(actd.data_ptr(), out.data_ptr(), n, H, W, CC, hip.stream_ptr(dev())), "maxpool fwd")
    assert torch.equal(out.cpu(), y.detach())
    din = torch.empty(n, HW, CC, device=dev())
    dyd = dy.to(dev())
    hip.check(hip.lib().pnmn_maxpool2_flatten_bwd(actd.data_ptr(), dyd.data_ptr(), din.data_ptr(), n, H, W, CC, hip.stream_ptr(dev())), "maxpool bwd")
    assert torch.equal(from_nhwc(din, n, CC), pre.grad)


def test_answer_loss(hip, map_size):
    if map_size != 14:
        pytest.skip("independent of the map size")
    g = gen(100)
    n, A = 37, 28
    logits = (torch.randn(n, A, generator=g) * 3).requires_grad_(True)
    answers = torch.randint(0, A, (n,), generator=g)
    valid = (torch.rand(n, generator=g) > 0.3).int()
    loss_ref = F.cross_entropy(logits, answers, reduction="none").clone()
    loss_ref[valid == 0] = 3.33
    (loss_ref.mean()).backward()
    ld, ad, vd = logits.detach().to(dev()), answers.to(dev()), valid.to(dev())
    pred = torch.empty(n, dtype=torch.long, device=dev())
    loss = torch.empty(n, device=dev())
    dl = torch.empty(n, A, device=dev())
    hip.check(hip.lib().pnmn_answer_loss(ld.data_ptr(), ad.data_ptr(), vd.data_ptr(), pred.data_ptr(), loss.data_ptr(), dl.data_ptr(), n, A, 28, 1.0 / n, hip.stream_ptr(dev())), "loss")
    torch.cuda.synchronize()
    torch.testing.assert_close(loss.cpu(), loss_ref.detach(), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(dl.cpu(), logits.grad, rtol=1e-4, atol=1e-7)
    pref = logits.detach().argmax(1)
    pref[valid == 0] = 28
    assert torch.equal(pred.cpu(), pref)
    # without answers: loss = -max logprob
    hip.check(hip.lib().pnmn_answer_loss(ld.data_ptr(), None, vd.data_ptr(), pred.data_ptr(), loss.data_ptr(), None, n, A, 28, 1.0, hip.stream_ptr(dev())), "loss")
    torch.cuda.synchronize()
    ref2 = -F.log_softmax(logits.detach(), -1).max(1).values
    ref2[valid == 0] = 3.33
    torch.testing.assert_close(loss.cpu(), ref2, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("V,T", [(44, 27), (100, 46), (130, 5)])
def test_sequence_nll_fwd_bwd(hip, map_size, V, T):
    """pnmn_seq_nll_{fwd,bwd} against log_softmax / gather / masked mean, incl. strided views (logits[:, :-1],
    targets[:, 1:]) and a mask taken from a different token matrix (sampled vs trimmed programs)."""
    if map_size != 14:
        pytest.skip("independent of the map size")
    g = gen(400 + V)
    B = 9
    full = (torch.randn(B, T + 1, V, generator=g) * 3).requires_grad_(True)
    toks = torch.randint(1, V, (B, T + 1), generator=g)
    masks = toks.clone()
    lens = torch.randint(0, T + 1, (B,), generator=g)
    lens[0], lens[1] = T + 1, 0  # a full row and an all-padding row (loss 0, no gradient)
    for b in range(B):
        masks[b, int(lens[b]):] = 0
    logits, tk, mk = full[:, :-1], toks[:, 1:], masks[:, 1:]
    w = (mk != 0).float()
    nll = -F.log_softmax(logits, dim=-1).gather(2, tk.unsqueeze(-1)).squeeze(-1) * w
    ref = nll.sum(1) / (w.sum(1) + 1e-13)
    dl = torch.randn(B, generator=g)
    ref.backward(dl)
    fd = full.detach().to(dev())
    ld, td, md = fd[:, :-1], toks.to(dev())[:, 1:], masks.to(dev())[:, 1:]
    loss = torch.empty(B, device=dev())
    lse = torch.empty(B, T, device=dev())
    st = hip.stream_ptr(dev())
    hip.check(hip.lib().pnmn_seq_nll_fwd(ld.data_ptr(), ld.stride(0), td.data_ptr(), td.stride(0), md.data_ptr(), md.stride(0),
                                         0, loss.data_ptr(), lse.data_ptr(), B, T, V, 1e-13, st), "seq_nll_fwd")
    dlog = torch.zeros(B, T, V, device=dev())
    dld = dl.to(dev())
    hip.check(hip.lib().pnmn_seq_nll_bwd(ld.data_ptr(), ld.stride(0), td.data_ptr(), td.stride(0), md.data_ptr(), md.stride(0),
                                         0, lse.data_ptr(), dld.data_ptr(), dlog.data_ptr(), T * V, B, T, V, 1e-13, st), "seq_nll_bwd")
    torch.cuda.synchronize()
    torch.testing.assert_close(loss.cpu(), ref.detach(), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(dlog.cpu(), full.grad[:, :-1], rtol=1e-4, atol=1e-6)
    assert float(loss[1]) == 0.0 and float(dlog[1].abs().max()) == 0.0


def test_clamp_adam_matches_torch(hip, map_size):
    if map_size != 14:
        pytest.skip("independent of the map size")
    g = gen(110)
    p = torch.randn(100003, generator=g)
    ref = p.clone().requires_grad_(True)
    opt = torch.optim.Adam([ref], lr=1e-3, weight_decay=0.01)
    pd = p.to(dev())
    m = torch.zeros_like(pd)
    v = torch.zeros_like(pd)
    for step in range(1, 4):
        grad = torch.randn(p.shape, generator=g) * 4
        ref.grad = grad.clamp(-5, 5)
        opt.step()
        gd = grad.to(dev())
        recs = np.zeros(1, hip.ADAM_ITEM)
        recs[0]["param"], recs[0]["grad"], recs[0]["exp_avg"], recs[0]["exp_avg_sq"], recs[0]["n"] = ptr(pd), ptr(gd), ptr(m), ptr(v), p.numel()
        recs[0]["bc1"], recs[0]["bc2_sqrt"] = 1.0 - 0.9 ** step, (1.0 - 0.999 ** step) ** 0.5  # (per item: ABI 9)
        run(hip, "pnmn_clamp_adam", recs, 1e-3, 0.9, 0.999, 1e-8, 0.01, 5.0)
    torch.testing.assert_close(pd.cpu(), ref.detach(), rtol=1e-5, atol=1e-6)
