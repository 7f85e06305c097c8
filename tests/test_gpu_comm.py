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


def test_flat_gradient_exchange_through_hoscomm_inside_a_captured_step(tmp_path):
    """`train.allreduce_flat_grad` on its HosComm path (round 5: `train.use_hoscomm`): a stage-1 forward + backward + the gradient
    exchange through libhoscomm + clip + Adam captured as ONE graph and replayed -- with torch.distributed's collectives the exchange
    has to stay outside the graph.  One rank, so the sum is the identity: the replayed step must equal the same step without a
    communicator; the MAX-reduce of the range-guard word and the volume-gradient exchange go through the same object."""
    import json
    import os
    from hosnerf_amd import synth, train
    from hosnerf_amd.comm import HosComm
    from hosnerf_amd.mipnerf360 import MipNeRF360
    from hosnerf_amd.train import FusedAdam, allreduce_flat_grad, stage1_loss
    dev = torch.device("cuda")
    d = str(tmp_path)
    json.dump({"f0": {"time": 0.4}}, open(os.path.join(d, "transitions_times.json"), "w"))
    b = {k: v.to(dev) for k, v in synth.stage1_batch(64, seed=1).items()}
    b["times"] = 0.5                                    # a host scalar: the state is selected on the host, no device read under capture
    jit = [torch.rand(64, device=dev) for _ in range(3)]          # one stratified offset per ray and level

    def build():
        m = MipNeRF360(d, opaque_background=True)
        m.load_state_dict(synth.background_state_dict(777, 2), strict=False)
        m = m.to(dev)
        return m, FusedAdam(m, lr=1e-3, max_grad_norm=0.001)

    def step(m, opt, hos):
        opt.zero_grad()
        rend, hist = m(b, 0.5, True, True, 0.1, 1e6, jitters=jit)
        stage1_loss(rend[-1]["rgb"], b["target"], hist)[0].backward()
        world = allreduce_flat_grad(m, hos=hos)
        opt.step(dynamic=True, reduced=True)
        return world

    m0, o0 = build()
    o0.set_step_hyper(1e-3)
    assert step(m0, o0, None) == 1
    torch.cuda.synchronize()
    c = HosComm(0, 1)
    prev = train.use_hoscomm(c)
    try:
        m1, o1 = build()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):                      # warm-up on a side stream (allocations, lazy initialisation)
            o1.set_step_hyper(1e-3)
            step(m1, o1, None)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        m1, o1 = build()
        graph = torch.cuda.CUDAGraph()
        o1.set_step_hyper(1e-3)
        with torch.cuda.graph(graph):
            assert step(m1, o1, None) == 1              # hos=None: the communicator registered with use_hoscomm
        p_before = m1.flat_param.clone()
        graph.replay()
        torch.cuda.synchronize()
        moved = float((m1.flat_param - p_before).abs().max())
        # (not bit-identical: a few bias gradients are summed with fp32 atomics, whose order differs between two launches)
        assert moved > 1e-4 and float((m1.flat_param - m0.flat_param).abs().max()) < 1e-3 * moved
        flag = torch.tensor([0, 3], dtype=torch.int32, device=dev)
        c.all_reduce_max_u32(flag)
        torch.cuda.synchronize()
        assert flag.tolist() == [0, 3]
    finally:
        train.use_hoscomm(prev)
        c.close()
