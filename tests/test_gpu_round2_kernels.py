"""Round-2 kernels against plain fp64 / fp32 torch references of the same op (each through the C ABI):
the ReLU bit mask of the planes GEMMs, the N = 256 q + r head split, the split-K forward-form fp32 GEMM of the volume
decoder's input gradients, the fused LeakyReLU-backward + bias-gradient pass, deferred (batched) slab reductions."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    from hosnerf_amd import _lib
    _lib.require_gpu()
    return torch.device("cuda")


def relu_bits_reference(act: torch.Tensor, rows_pad: int, ld: int) -> torch.Tensor:
    """include/hosrender.h (hos_linearp_fwd, relu_bits) restated with torch ops: per 32-row x 64-column block 64 dwords,
    dword l + 32 h, bit 31 - (16 y + r) <-> row (r&3) + 8 (r>>2) + 4 h, column 32 y + l."""
    M, N = act.shape
    full = torch.zeros(rows_pad, (ld + 63) // 64 * 64, dtype=torch.bool, device=act.device)
    full[:M, :N] = act > 0
    Mr, Kc = full.shape
    m = full.view(Mr // 32, 32, Kc // 64, 2, 32).permute(0, 2, 1, 3, 4)
    r = torch.arange(16, device=act.device)
    rowidx = ((r & 3) + 8 * (r >> 2))[None, :] + 4 * torch.arange(2, device=act.device)[:, None]
    mm = m[:, :, rowidx].permute(0, 1, 2, 5, 4, 3)
    w = (2 ** (31 - (16 * torch.arange(2, device=act.device)[:, None] + r[None, :]))).to(torch.int64)
    v = (mm.to(torch.int64) * w).sum((-1, -2))
    return torch.where(v >= 2 ** 31, v - 2 ** 32, v).to(torch.int32).reshape(Mr // 32, Kc // 64, 64)


@pytest.mark.parametrize("M,N,K", [(1024, 1024, 256), (1000, 96, 64), (300, 256, 128), (4096, 160, 576), (64, 32, 32)])
def test_relu_bit_mask(dev, M, N, K):
    """FWD writes one bit per element (acc + bias > 0) in accumulator layout; DGRAD masked by those bits == DGRAD masked by the
    fp16 planes of the activation == fp64.  Ragged M and N, both tile widths, rows / columns of partial blocks."""
    from hosnerf_amd import ops
    g = torch.Generator(device="cuda").manual_seed(M + N)
    X = torch.randn(M, K, device=dev, generator=g)
    W = torch.randn(N, K, device=dev, generator=g) / K ** 0.5
    b = torch.randn(N, device=dev, generator=g) * 0.3
    X16, _ = ops.split_planes2(X, wantb=False)
    W16, _ = ops.split_planes(W, dtype=torch.float16)
    ld = (N + 31) // 32 * 32
    Y = ops.Planes.empty(M, ld, torch.float16, dev, N, relu_bits=True)
    Y.bits.fill_(-1)                                                       # rows / blocks the kernel must not need stay garbage
    ops.linearp_fwd(X16, K, W16, b, M, N, True, Y, None)
    act = Y.float()                                                         # what the kernel stored: relu(acc + bias) as hi + lo
    ref_bits = relu_bits_reference(act, (M + 31) // 32 * 32, ld)
    # whole blocks below M: every dword; the last partial row block: only the bits of rows < M
    nfull = M // 32
    assert int((Y.bits[:nfull] != ref_bits[:nfull]).sum()) == 0
    assert float((torch.relu(X.double() @ W.double().T + b.double()) - act.double()).abs().max()) < 2e-5
    # next layer's data gradient: dX[M, N] = (dZ[M, N2] @ W2[N2, N]) masked by act > 0
    N2 = 64
    W2 = torch.randn(N2, N, device=dev, generator=g) / N ** 0.5
    dY = torch.randn(M, N2, device=dev, generator=g)
    _, WTb = ops.split_planes(W2, dtype=torch.bfloat16, transposed=True, row_major=False)
    _, dZ = ops.split_planes2(dY, want16=False)
    d_bits = ops.Planes.empty(M, ld, torch.bfloat16, dev, N)
    d_planes = ops.Planes.empty(M, ld, torch.bfloat16, dev, N)
    ops.linearp_dgrad(dZ, WTb, dZ.ld, M, N, mask=Y, dX=d_bits)
    Yplain = ops.Planes(Y.t, Y.rows, Y.cols)                               # the same activation without its bits: planes mask
    ops.linearp_dgrad(dZ, WTb, dZ.ld, M, N, mask=Yplain, dX=d_planes)
    # the two masks differ only where 0 < acc + bias underflows the fp16 hi part (|x| < 3e-8): none at these scales
    assert float((d_bits.float() - d_planes.float()).abs().max()) == 0.0
    ref = (dY.double() @ W2.double()) * (act > 0)
    assert float((d_bits.float().double() - ref).abs().max()) < 2e-4 * float(ref.abs().max()) + 1e-5


@pytest.mark.parametrize("M,N,K", [(1500, 257, 1024), (700, 300, 64), (260, 384, 128)])
def test_head_split_256q_plus_r(dev, M, N, K):
    """N = 256 q + r (r <= 128) with an fp32 epilogue runs as a 256-wide + a 128-wide launch: same result as fp64, incl. the
    NeRF head's density column (aux) on either side of the split."""
    from hosnerf_amd import ops
    g = torch.Generator(device="cuda").manual_seed(N)
    X = torch.randn(M, K, device=dev, generator=g)
    W = torch.randn(N, K, device=dev, generator=g) / K ** 0.5
    b = torch.randn(N, device=dev, generator=g)
    X16, _ = ops.split_planes2(X, wantb=False)
    W16, _ = ops.split_planes(W, dtype=torch.float16)
    ref = X.double() @ W.double().T + b.double()
    C = torch.full((M, (N + 3) // 4 * 4), 7.0, device=dev)
    ops.linearp_fwd(X16, K, W16, b, M, N, False, None, None, C=C, epilogue=ops.EPI_RELU)
    assert float((C[:, :N].double() - torch.relu(ref)).abs().max()) < 2e-5
    for aux_col in (N - 1, 3):
        C.fill_(7.0)
        dens = torch.empty(M, device=dev)
        ops.linearp_fwd(X16, K, W16, b, M, N, False, None, None, C=C, epilogue=ops.EPI_NERF_HEAD, aux=dens, aux_col=aux_col, p0=-1.0)
        keep = [c for c in range(N) if c != aux_col]
        assert float((C[:, keep].double() - ref[:, keep]).abs().max()) < 2e-5
        assert float((dens.double() - torch.nn.functional.softplus(ref[:, aux_col] - 1.0)).abs().max()) < 2e-5
        assert float(C[:, aux_col].min()) == 7.0 and float(C[:, aux_col].max()) == 7.0        # the density column is not stored in C


@pytest.mark.parametrize("M,N,K", [(1, 1024, 32768), (8, 512, 32768), (64, 512, 16384), (512, 256, 16384), (4096, 256, 1728), (5, 40, 64)])
def test_linear_fwd_splitk(dev, M, N, K):
    """C = A . W^T with the reduction split over workgroups (the decoder's input gradients) == fp64; stale contents of C are
    overwritten; exact-fp32 MFMA accuracy."""
    from hosnerf_amd._lib import call, ptr
    g = torch.Generator(device="cuda").manual_seed(K + M)
    A = torch.randn(M, K, device=dev, generator=g)
    W = torch.randn(N, K, device=dev, generator=g) / K ** 0.5
    C = torch.full((M, N), 3.0, device=dev)
    call("hos_linear_fwd_splitk", ptr(A), A.stride(0), ptr(W), W.stride(0), ptr(C), C.stride(0), M, N, K)
    ref = A.double() @ W.double().T
    assert float((C.double() - ref).abs().max()) < 3e-5 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("R,C,leaky", [(32768, 27, False), (4096, 256, True), (8, 512, True), (512, 100, True), (3, 5, False)])
def test_deconv3d_dpre(dev, R, C, leaky):
    """LeakyReLU(0.2) backward + bias gradient in one pass == torch.where + column sum (accumulating into db)."""
    from hosnerf_amd._lib import call, ptr
    g = torch.Generator(device="cuda").manual_seed(R + C)
    go = torch.randn(R, C, device=dev, generator=g)
    out = torch.randn(R, C, device=dev, generator=g)
    out[::3] = 0.0                                                          # out == 0 takes the negative slope (x > 0 is false)
    db = torch.full((C,), 0.5, device=dev)
    dpre = torch.full((R, C), 9.0, device=dev)
    call("hos_deconv3d_dpre", ptr(go), ptr(out), R, C, 0.2, int(leaky), ptr(dpre) if leaky else None, ptr(db))
    ref = torch.where(out > 0, go, 0.2 * go) if leaky else go
    if leaky:
        assert torch.equal(dpre, ref)
    else:
        assert float(dpre.min()) == 9.0                                     # not written
    assert float((db.double() - (0.5 + ref.double().sum(0))).abs().max()) < 1e-3 * max(1.0, float(ref.double().sum(0).abs().max()))


def test_deferred_slab_reductions(dev):
    """ops.deferred_bwd_reduce(): the slab reductions of several thin-layer backward launches run as one batched launch at the
    exit; gradients equal the immediate form's (and fp64), more jobs than one batch holds, mixed layer shapes."""
    from hosnerf_amd import ops
    g = torch.Generator(device="cuda").manual_seed(5)
    M = 40000
    layers = []
    for i in range(20):                                                     # > 16 jobs: two batched launches
        N, K = (128, 128) if i % 3 else (256, 256)
        dY = torch.randn(M, N, device=dev, generator=g) * 1e-2
        X = torch.relu(torch.randn(M, K, device=dev, generator=g))
        W = torch.randn(N, K, device=dev, generator=g) / K ** 0.5
        layers.append((N, K, dY, X, W))

    def run(deferred):
        outs = []
        ctx = ops.deferred_bwd_reduce() if deferred else None
        if ctx:
            ctx.__enter__()
        for N, K, dY, X, W in layers:
            dW = torch.zeros(N, K, device=dev); db = torch.zeros(N, device=dev)
            if N == 128:
                dX = torch.empty(M, K, device=dev)
                ops.linear_bwd_fused(dY, X, W, dW, db, N, K, dX, True)
            else:
                ops.linear_wgrad(dY, X, dW, db, N, K)
            outs.append((dW, db))
        if ctx:
            ctx.__exit__(None, None, None)
        torch.cuda.synchronize()
        return outs

    a, b = run(True), run(False)
    for (dWa, dba), (dWb, dbb), (N, K, dY, X, W) in zip(a, b, layers):
        ref = dY.double().T @ X.double()
        tol = 3e-4 * float(ref.abs().max())
        assert float((dWa.double() - ref).abs().max()) < tol and float((dWb.double() - ref).abs().max()) < tol
        assert float((dWa - dWb).abs().max()) < 1e-5 * float(ref.abs().max()) + 1e-7       # same partials, fp32 atomics in another order
        assert float((dba.double() - dY.double().sum(0)).abs().max()) < 2e-3


@pytest.mark.parametrize("M,K,softplus", [(131072, 256, True), (65536, 1024, True), (1000, 256, True), (777, 1024, False), (3, 64, True)])
def test_planes_rowdot_head(dev, M, K, softplus):
    """One-column head as a row dot over fp16 planes == fp64 on the SAME planes values (the weight row is not split), with the
    device-scalar bias and the density bias; and it agrees with the N = 1 planes GEMM it replaces."""
    from hosnerf_amd import ops
    g = torch.Generator(device="cuda").manual_seed(M + K)
    X = torch.randn(M, K, device=dev, generator=g)
    w = torch.randn(K, device=dev, generator=g) / K ** 0.5
    b = torch.randn(1, device=dev, generator=g)
    X16, _ = ops.split_planes2(X, wantb=False)
    out = torch.full((M,), 9.0, device=dev)
    ops.planes_rowdot(X16, K, w, b, out, p0=-1.0, softplus=softplus)
    z = X16.float().double() @ w.double() + b.double() - 1.0
    ref = torch.nn.functional.softplus(z) if softplus else z
    assert float((out.double() - ref).abs().max()) < 3e-6
    if softplus:
        W16, _ = ops.split_planes(w.view(1, K).contiguous(), dtype=torch.float16)
        dens = torch.empty(M, device=dev)
        ops.linearp_fwd(X16, K, W16, b, M, 1, False, None, None, epilogue=ops.EPI_DENSITY, aux=dens, p0=-1.0)
        assert float((out - dens).abs().max()) < 1e-5


@pytest.mark.parametrize("P", [1, 1000, 70001])
def test_zero_padded_operand_rows(dev, P):
    """hos_slice_pad / hos_rgbsigma_grad write WHOLE [P, 32] rows (values in the leading columns, zeros behind) into uninitialised
    storage; rows past a device-side count stay untouched."""
    from hosnerf_amd import ops
    g = torch.Generator().manual_seed(P)
    src = torch.randn(P, 3, generator=g).to(dev)
    out = torch.full((P, 32), float("nan"), device=dev)
    ops.slice_pad(src, 0, 3, out)
    assert torch.equal(out[:, :3], src) and float(out[:, 3:].abs().max()) == 0
    live = max(1, P // 2)
    out = torch.full((P, 32), 7.0, device=dev)
    ops.slice_pad(src, 0, 3, out, rows_dev=torch.tensor([live], dtype=torch.int32, device=dev))
    assert torch.equal(out[:live, :3], src[:live]) and float(out[:live, 3:].abs().max()) == 0 and torch.all(out[live:] == 7.0)
    y = torch.rand(P, 4, generator=g).to(dev)
    y[:, 3] = torch.where(torch.rand(P, generator=g).to(dev) < 0.5, torch.zeros(P, device=dev), y[:, 3])
    gy = torch.randn(P, 4, generator=g).to(dev)
    dz = torch.full((P, 32), float("nan"), device=dev)
    ops.rgbsigma_grad(gy, y, dz)
    want = torch.cat([gy[:, :3] * y[:, :3] * (1 - y[:, :3]), torch.where(y[:, 3:] > 0, gy[:, 3:], torch.zeros_like(gy[:, 3:]))], -1)
    assert float((dz[:, :4] - want).abs().max()) < 1e-6 and float(dz[:, 4:].abs().max()) == 0
