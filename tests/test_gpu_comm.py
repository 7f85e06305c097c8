"""libhoscomm.so on the MI355X: the RCCL collectives behind include/hoscomm.h, eagerly and INSIDE a captured hipGraph (what
torch.distributed's collectives cannot do), on a one-rank communicator -- this pool's boxes have one GPU; the N > 1 arithmetic of the
exchange is covered by tests/test_dist_cpu.py (gloo, world 2) and tests/test_gpu_dist.py."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_one_rank_communicator_eager_and_in_a_graph():
    from hosnerf_amd.comm import HosComm
    dev = torch.device("cuda")
    c = HosComm(0, 1)
    try:
        assert c.size() == 1
        x = torch.randn(1 << 20, device=dev)
        ref = x.clone()
        c.all_reduce(x)                                   # average over one rank: identity
        c.all_reduce(x, average=False)
        torch.cuda.synchronize()
        assert torch.equal(x, ref)
        c.all_reduce_spans(x, [(0, 1024), (4096, 0), (8192, 100000)])
        g = c.all_gather(x[:4096].view(64, 64))
        torch.cuda.synchronize()
        assert torch.equal(x, ref) and g.shape == (1, 64, 64) and torch.equal(g[0], ref[:4096].view(64, 64))
        # inside a captured step: a kernel, the collective, another kernel -- replayed three times
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        buf = torch.zeros(4096, device=dev)
        with torch.cuda.stream(s):
            for _ in range(2):
                buf.add_(1.0)
                c.all_reduce(buf)
                buf.mul_(2.0)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        buf.zero_()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            buf.add_(1.0)
            c.all_reduce(buf)
            buf.mul_(2.0)
        for _ in range(3):
            graph.replay()
        torch.cuda.synchronize()
        assert float(buf[0]) == 14.0 and float(buf[-1]) == 14.0      # ((0+1)*2+1)*2+1)*2
    finally:
        c.close()
