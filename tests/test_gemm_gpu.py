"""pnmn_gemm / pnmn_colsum (csrc/gemm.hip) through the C ABI against torch in fp64: every operand layout, ragged shapes
(vocabulary-sized N / K that are no multiple of 4), bias, accumulation, split-K (deterministic: two runs are bit-equal),
the shifted "previous state" operand, and several problems in one launch.  Tolerance: 2e-5 of the largest entry of the
result plus K-scaled fp32 round-off (measured ~1e-6)."""
import numpy as np
import pytest
import torch

from probnmn import _hip

pytestmark = pytest.mark.gpu


def _desc(a, b, c, M, N, K, flags=0, bias=None, split=1, ws=None, shift_t=0, h0=None):
    d = np.zeros(1, _hip.GEMM_DESC)
    d["a"], d["b"], d["c"] = a.data_ptr(), b.data_ptr(), c.data_ptr()
    d["lda"], d["ldb"], d["ldc"] = a.stride(0), b.stride(0), c.stride(0)
    d["M"], d["N"], d["K"], d["flags"], d["split_k"], d["shift_t"] = M, N, K, flags, split, shift_t
    if bias is not None:
        d["bias"] = bias.data_ptr()
    if ws is not None:
        d["workspace"] = ws.data_ptr()
    if h0 is not None:
        d["shift_h0"], d["ld_h0"] = h0.data_ptr(), h0.stride(0)
    return d


def _run(descs):
    dev = torch.device("cuda:0")
    rec = np.concatenate(descs)
    _hip.check(_hip.lib().pnmn_gemm(rec.ctypes.data, len(rec), _hip.stream_ptr(dev)), "gemm")
    torch.cuda.synchronize()


def _ws(M, N, split):
    n = int(_hip.lib().pnmn_gemm_workspace_bytes(M, N, split))
    return torch.zeros(max(n, 4), dtype=torch.uint8, device="cuda:0")


def _close(got, want, K):
    scale = float(want.abs().max()) + 1e-30
    err = float((got.double() - want).abs().max()) / scale
    assert err < 2e-5 + 2e-7 * K ** 0.5, err


@pytest.mark.parametrize("ta,tb", [(0, 0), (0, 1), (1, 0), (1, 1)])
@pytest.mark.parametrize("M,N,K", [(300, 200, 256), (128, 128, 32), (517, 93, 260), (44, 256, 1000), (1000, 44, 47), (1, 1, 1)])
def test_layouts_and_ragged_shapes(ta, tb, M, N, K):
    g = torch.Generator(device="cuda:0").manual_seed(M * 7 + N * 3 + K + ta * 2 + tb)
    A = torch.randn((K, M) if ta else (M, K), device="cuda:0", generator=g)
    B = torch.randn((N, K) if tb else (K, N), device="cuda:0", generator=g)
    bias = torch.randn(N, device="cuda:0", generator=g)
    C = torch.full((M, N), float("nan"), device="cuda:0")
    _run([_desc(A, B, C, M, N, K, flags=ta * _hip.GEMM_A_T + tb * _hip.GEMM_B_T, bias=bias)])
    want = (A.double().t() if ta else A.double()) @ (B.double().t() if tb else B.double()) + bias.double()
    _close(C, want, K)


def test_strided_operands_and_accumulate():
    g = torch.Generator(device="cuda:0").manual_seed(5)
    big_a = torch.randn(700, 300, device="cuda:0", generator=g)
    big_b = torch.randn(1024, 512, device="cuda:0", generator=g)
    big_c = torch.randn(700, 2048, device="cuda:0", generator=g)
    A, B, C = big_a[:, 20:276], big_b[:, 256:], big_c[:, 1024:]  # [700,256], weight [1024,256] as a column block, out [700,1024]
    before = C.clone()
    _run([_desc(A, B, C, 700, 1024, 256, flags=_hip.GEMM_B_T | _hip.GEMM_ACC)])
    _close(C, before.double() + A.double() @ B.double().t(), 256)
    assert torch.equal(big_c[:, :1024], big_c[:, :1024])  # (untouched columns stay finite)


@pytest.mark.parametrize("split", [2, 7, 16, 64])
def test_split_k_weight_gradient_is_deterministic(split):
    g = torch.Generator(device="cuda:0").manual_seed(split)
    K, M, N = 128 * 47, 1024, 256
    dy = torch.randn(K, M, device="cuda:0", generator=g)
    x = torch.randn(K, N, device="cuda:0", generator=g)
    ws = _ws(M, N, split)
    outs = []
    for _ in range(2):
        C = torch.empty(M, N, device="cuda:0")
        _run([_desc(dy, x, C, M, N, K, flags=_hip.GEMM_A_T, split=split, ws=ws)])
        outs.append(C)
    assert torch.equal(outs[0], outs[1])
    _close(outs[0], dy.double().t() @ x.double(), K)


def test_shifted_previous_state_operand():
    g = torch.Generator(device="cuda:0").manual_seed(11)
    B_, T, H = 37, 13, 256
    hs = torch.randn(B_, T, H, device="cuda:0", generator=g)
    h0 = torch.randn(B_, H, device="cuda:0", generator=g)
    dg = torch.randn(B_ * T, 1024, device="cuda:0", generator=g)
    for init in (h0, None):
        first = h0 if init is not None else torch.zeros_like(h0)
        hprev = torch.cat((first.unsqueeze(1), hs[:, :-1]), 1).reshape(B_ * T, H)
        split = 4
        C = torch.empty(1024, H, device="cuda:0")
        ws = _ws(1024, H, split)
        _run([_desc(dg, hs.view(B_ * T, H), C, 1024, H, B_ * T, flags=_hip.GEMM_A_T, split=split, ws=ws, shift_t=T, h0=init)])
        _close(C, dg.double().t() @ hprev.double(), B_ * T)


def test_eight_problems_in_one_launch():
    g = torch.Generator(device="cuda:0").manual_seed(3)
    descs, checks = [], []
    for k in range(8):
        M, N, K = 64 + 50 * k, 300 - 30 * k, 96 + 32 * k
        A = torch.randn(M, K, device="cuda:0", generator=g)
        B = torch.randn(N, K, device="cuda:0", generator=g)
        C = torch.empty(M, N, device="cuda:0")
        split = 1 + (k % 3)
        ws = _ws(M, N, split)  # (kept alive with the operands: the record only holds its address)
        descs.append(_desc(A, B, C, M, N, K, flags=_hip.GEMM_B_T, split=split, ws=ws))
        checks.append((C, A, B, K, ws))
    _run(descs)
    for C, A, B, K, _ in checks:
        _close(C, A.double() @ B.double().t(), K)


@pytest.mark.parametrize("R,C", [(5888, 1024), (3, 44), (0, 256), (47104, 1024)])
def test_colsum(R, C):
    g = torch.Generator(device="cuda:0").manual_seed(R + C)
    x = torch.randn(max(R, 1), C, device="cuda:0", generator=g)[:R]
    ws = torch.zeros(int(_hip.lib().pnmn_colsum_workspace_bytes(R, C)), dtype=torch.uint8, device="cuda:0")
    out, out2 = torch.ones(C, device="cuda:0"), torch.empty(C, device="cuda:0")
    for acc in (0, 1):
        prev = out.clone()
        _hip.check(_hip.lib().pnmn_colsum(x.data_ptr(), C, R, C, out.data_ptr(), out2.data_ptr(), acc, ws.data_ptr(),
                                          _hip.stream_ptr(x.device)), "colsum")
        torch.cuda.synchronize()
        want = x.double().sum(0) + (prev.double() if acc else 0)
        assert float((out.double() - want).abs().max()) < 1e-4 * (1 + float(want.abs().max()))
        assert torch.equal(out, out2)


def test_rate_report(capsys):
    """Not an assertion on speed (boxes differ): prints TFLOP/s of the shapes the seq2seq plan launches."""
    shapes = [("xp2 b1024", 47104, 1024, 256, 0, 1, 1), ("xp2 b128", 5888, 1024, 256, 0, 1, 1), ("dx b1024", 47104, 256, 1024, 0, 0, 1),
              ("logits b1024", 47104, 93, 256, 0, 1, 1), ("dx b128", 5888, 256, 1024, 0, 0, 3),
              ("fc1 fwd 64", 64, 1024, 50176, 0, 1, 48), ("fc1 dx 64", 64, 50176, 1024, 0, 0, 1), ("fc1 dw 64", 1024, 50176, 64, 1, 0, 1),
              ("fc1 fwd 512", 512, 1024, 50176, 0, 1, 12), ("fc1 dx 512", 512, 50176, 1024, 0, 0, 1), ("fc1 dw 512", 1024, 50176, 512, 1, 0, 1)] \
        + [("wgrad b1024", 1024, 256, 47104, 1, 0, s) for s in (4, 16, 32)] + [("wgrad b128", 1024, 256, 5888, 1, 0, s) for s in (1, 2, 4, 8, 16, 32)]
    for name, M, N, K, ta, tb, split in shapes:
        A = torch.randn((K, M) if ta else (M, K), device="cuda:0")
        B = torch.randn((N, K) if tb else (K, N), device="cuda:0")
        C = torch.empty(M, N, device="cuda:0")
        ws = _ws(M, N, split)
        d = _desc(A, B, C, M, N, K, flags=ta * _hip.GEMM_A_T + tb * _hip.GEMM_B_T, split=split, ws=ws)
        st = _hip.stream_ptr(A.device)
        for _ in range(3):
            _hip.lib().pnmn_gemm(d.ctypes.data, 1, st)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            _hip.lib().pnmn_gemm(d.ctypes.data, 1, st)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        ref = (A.t() if ta else A) @ (B.t() if tb else B)
        e0.record()
        for _ in range(20):
            ref = (A.t() if ta else A) @ (B.t() if tb else B)
        e1.record()
        torch.cuda.synchronize()
        ms_t = e0.elapsed_time(e1) / 20
        with capsys.disabled():
            print("gemm %-14s M %6d N %5d K %6d split %2d: %7.1f us = %6.1f TFLOP/s (torch %7.1f us)"
                  % (name, M, N, K, split, ms * 1e3, 2.0 * M * N * K / ms / 1e9, ms_t * 1e3))


@pytest.mark.parametrize("M,N,K,split,tb", [(1024, 256, 128 * 37 + 5, 9, 0), (44, 256, 3000, 1, 0), (100, 256, 6000, 5, 0), (300, 130, 257, 1, 1),
                                            (1024, 512, 4096, 4, 0)])
def test_column_sums_of_the_transposed_operand(M, N, K, split, tb):
    """pnmn_gemm_desc.colsum: the bias gradient beside a weight gradient dy^T x (sum over the rows of dy), split or not,
    ragged widths (lda = vocabulary size: the scalar loader), two outputs; twice the same bits."""
    g = torch.Generator(device="cuda:0").manual_seed(M + N + K)
    dy = torch.randn(K, M, device="cuda:0", generator=g)
    x = torch.randn((N, K) if tb else (K, N), device="cuda:0", generator=g)
    ws = _ws(M, N, split)
    outs = []
    for _ in range(2):
        C = torch.empty(M, N, device="cuda:0")
        s1 = torch.full((M + 3,), float("nan"), device="cuda:0")
        s2 = torch.full((M + 3,), float("nan"), device="cuda:0")
        d = _desc(dy, x, C, M, N, K, flags=_hip.GEMM_A_T + tb * _hip.GEMM_B_T, split=split, ws=ws)
        d["colsum"], d["colsum2"] = s1.data_ptr(), s2.data_ptr()
        _run([d])
        outs.append((C, s1, s2))
    C, s1, s2 = outs[0]
    _close(C, dy.double().t() @ (x.double().t() if tb else x.double()), K)
    _close(s1[:M], dy.double().sum(0), K)
    assert torch.equal(s1[:M], s2[:M]) and bool(torch.isnan(s1[M:]).all()) and bool(torch.isnan(s2[M:]).all())
    assert torch.equal(outs[1][1][:M], s1[:M]) and torch.equal(outs[1][0], C)


def test_column_sums_need_the_transposed_layout():
    a = torch.zeros(128, 64, device="cuda:0")
    b = torch.zeros(64, 128, device="cuda:0")
    c = torch.zeros(128, 128, device="cuda:0")
    d = _desc(a, b, c, 128, 128, 64)
    d["colsum"] = c.data_ptr()
    assert _hip.lib().pnmn_gemm(d.ctypes.data, 1, _hip.stream_ptr(torch.device("cuda:0"))) == _hip.ESHAPE


@pytest.mark.parametrize("ta,tb", [(0, 0), (0, 1), (1, 0), (1, 1)])
@pytest.mark.parametrize("M,N,K", [(256, 384, 70), (128, 128, 33), (384, 256, 4 * 32 + 31)])
def test_interior_tiles_with_a_partial_last_k_tile(ta, tb, M, N, K):
    """Whole 128 x 128 tiles (the direct-to-LDS loop) followed by a last k tile that is not full (the register loop)."""
    g = torch.Generator(device="cuda:0").manual_seed(M + 3 * N + 5 * K + ta + 2 * tb)
    ldk = K + (4 - K % 4) % 4 + 4  # (16-byte aligned rows, wider than K)
    A = torch.randn((K, M) if ta else (M, ldk), device="cuda:0", generator=g)
    B = torch.randn((N, ldk) if tb else (K, N), device="cuda:0", generator=g)
    Av, Bv = (A if ta else A[:, :K]), (B[:, :K] if tb else B)
    C = torch.full((M, N), float("nan"), device="cuda:0")
    _run([_desc(Av, Bv, C, M, N, K, flags=ta * _hip.GEMM_A_T + tb * _hip.GEMM_B_T)])
    _close(C, (Av.double().t() if ta else Av.double()) @ (Bv.double().t() if tb else Bv.double()), K)


def test_operands_that_are_not_16_byte_aligned_take_the_register_loader():
    g = torch.Generator(device="cuda:0").manual_seed(21)
    M, N, K = 256, 256, 128
    buf_a = torch.randn(M * K + 1, device="cuda:0", generator=g)
    buf_b = torch.randn(N * K + 3, device="cuda:0", generator=g)
    A, B = buf_a[1:].view(M, K), buf_b[3:].view(N, K)  # (4 and 12 bytes off a 16-byte boundary)
    C = torch.empty(M, N, device="cuda:0")
    _run([_desc(A, B, C, M, N, K, flags=_hip.GEMM_B_T)])
    _close(C, A.double() @ B.double().t(), K)


@pytest.mark.parametrize("T,rows,split", [(13, 64, 1), (7, 96, 3), (46, 32, 4)])
def test_shifted_operand_with_interior_tiles(T, rows, split):
    """dy^T direct to LDS beside the shifted state operand through registers (whole k tiles), then a partial last k tile; the
    hidden-state rows of example b at t = 0 come from h0[b]."""
    g = torch.Generator(device="cuda:0").manual_seed(T * rows)
    H = 256
    hs = torch.randn(rows, T, H, device="cuda:0", generator=g)
    h0 = torch.randn(rows, 2 * H, device="cuda:0", generator=g)[:, H:]  # (row stride 512: ld_h0)
    dg = torch.randn(rows * T, 1024, device="cuda:0", generator=g)
    for init in (h0, None):
        first = h0 if init is not None else torch.zeros(rows, H, device="cuda:0")
        hprev = torch.cat((first.unsqueeze(1), hs[:, :-1]), 1).reshape(rows * T, H)
        C = torch.empty(1024, H, device="cuda:0")
        ws = _ws(1024, H, split)
        _run([_desc(dg, hs.view(rows * T, H), C, 1024, H, rows * T, flags=_hip.GEMM_A_T, split=split, ws=ws, shift_t=T, h0=init)])
        _close(C, dg.double().t() @ hprev.double(), rows * T)
