"""Times the pieces of the motion-weight volume decoder (ConvTranspose3d stack, hos_deconv.hip + exact-fp32 GEMMs) one by one.
usage: python scripts/bench_decoder.py   (on the GPU box)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hosnerf_amd import ops
from hosnerf_amd._lib import call, ptr

dev = torch.device("cuda:0")


def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


chans = [(1024, 512), (512, 512), (512, 256), (256, 256), (256, 27)]      # network_util.py:31-44 at volume_size 32, 26 bones
D = 1
print(f"{'layer':28s} {'fwd gemm':>9s} {'torch mm':>9s} {'col2im':>8s} | {'dpre':>7s} {'im2col':>7s} {'dx':>7s} {'mm':>7s} {'dW':>7s} {'mm':>7s}")
for n in range(5):
    Cin, Cout = chans[n]
    M = D ** 3
    x = torch.randn(M, Cin, device=dev)
    W = torch.randn(Cin, Cout * 64, device=dev) * 0.02
    bias = torch.zeros(Cout, device=dev)
    ycol = torch.empty(M, Cout * 64, device=dev)
    out = torch.empty(8 * M, Cout, device=dev)
    g = torch.randn(8 * M, Cout, device=dev)
    dpre = torch.empty_like(g); db = torch.zeros(Cout, device=dev)
    dycol = torch.empty(M, Cout * 64, device=dev)
    dx = torch.empty(M, Cin, device=dev)
    gW = torch.zeros(Cin, Cout * 64, device=dev)
    with ops.gemm_mode(ops.GEMM_FP32):
        t_f = timeit(lambda: ops.linear_dgrad(x, W, Cin, Cout * 64, ycol))
        t_mm = timeit(lambda: torch.mm(x, W, out=ycol))
        t_c = timeit(lambda: call("hos_deconv3d_col2im", ptr(ycol), ptr(bias), D, Cout, 0.2, 1, ptr(out)))
        t_dp = timeit(lambda: call("hos_deconv3d_dpre", ptr(g), ptr(out), 8 * M, Cout, 0.2, 1, ptr(dpre), ptr(db)))
        t_im = timeit(lambda: call("hos_deconv3d_im2col", ptr(dpre), D, Cout, ptr(dycol)))
        t_dx = timeit(lambda: call("hos_linear_fwd_splitk", ptr(dycol), dycol.stride(0), ptr(W), W.stride(0), ptr(dx), dx.stride(0), M, Cin, Cout * 64))
        t_dxmm = timeit(lambda: torch.mm(dycol, W.t(), out=dx))
        if M <= 64:
            t_dw = timeit(lambda: call("hos_outer_accum", ptr(x), x.stride(0), ptr(dycol), dycol.stride(0), ptr(gW), gW.stride(0), M, Cin, Cout * 64))
        else:
            t_dw = timeit(lambda: ops.linear_wgrad(x, dycol, gW, None, Cin, Cout * 64))
        t_dwmm = timeit(lambda: gW.addmm_(x.t(), dycol))
        t_dwg = timeit(lambda: ops.linear_wgrad(x, dycol, gW, None, Cin, Cout * 64)) if M <= 64 else t_dw      # the tiled GEMM for few rows too
    print(f"L{n} M={M:5d} Cin={Cin:4d} N={Cout*64:6d} {t_f:9.1f} {t_mm:9.1f} {t_c:8.1f} | {t_dp:7.1f} {t_im:7.1f} {t_dx:7.1f} {t_dxmm:7.1f} {t_dw:7.1f} {t_dwmm:7.1f}  (tiled wgrad {t_dwg:6.1f})")
    D *= 2

# the compact first layer as the network runs it: ONE voxel, 8 live taps: [1, 1024] x [1024, 4096]
x = torch.randn(1, 1024, device=dev); wc = torch.randn(1024, 4096, device=dev) * 0.02; gwc = torch.zeros_like(wc); b = torch.zeros(512, device=dev)
dy = torch.randn(1, 4096, device=dev)
t_f = timeit(lambda: ops.deconv3d_first(x, wc, gwc, b, True))
t_o = timeit(lambda: call("hos_outer_accum", ptr(x), x.stride(0), ptr(dy), dy.stride(0), ptr(gwc), gwc.stride(0), 1, 1024, 4096))
print(f"compact first layer: forward (gemv + bias + leaky) {t_f:.1f} us, weight gradient (outer) {t_o:.1f} us")
