"""Worker of tests/test_gpu_dist.py::test_two_rank_step_equals_gradient_averaging -- launched with torch.distributed.run, two
ranks sharing the one GPU (gloo carries the collectives).  Each rank takes the data-parallel stage-2 step of bench.py on ITS
item (different poses / rays per rank, like the reference's per-rank dataloader); rank 0 saves the parameters after STEPS steps.
HOS_SHARD_DECODER=1: the volume decoder's first three layers are sharded over the two ranks (bench._maybe_shard_decoder)."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.environ.get("HOS_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

RAYS, STEPS = 512, 2


def seed_for(rank, step):
    return 4242 + 1000 * rank + step


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    w = bench.Stage2(dev, rank, world, RAYS)
    extra = {}
    if os.environ.get("HOS_SHARD_DECODER") == "1":
        # the sharded forward against a replicated network with the same weights (every rank takes part in the collectives)
        with torch.no_grad():
            vol = w.net._motion_weight_volume(w.batch["motion_weights_priors"])
            comm, layers = w.net.decoder_shard, w.net._shard_layers
            w.net.decoder_shard, w.net._shard_layers = None, ()
            ref = w.net._motion_weight_volume(w.batch["motion_weights_priors"])
            w.net.decoder_shard, w.net._shard_layers = comm, layers
        extra["volume_rel_err"] = float((vol - ref).abs().max() / ref.abs().max())
    for i in range(STEPS):
        torch.manual_seed(seed_for(rank, i))          # the stratified jitter of this rank's step
        w.host_prepare(i)
        w.eager_step(i)
        if i == 0:                                    # the clip norm of the first step: sum of squares of the SUMMED gradient, all shards
            torch.cuda.synchronize()
            extra["sumsq_step0"] = float(w.opt.clip._partials.double().sum()) if getattr(w.opt.clip, "_partials", None) is not None else None
    torch.cuda.synchronize()
    if os.environ.get("HOS_SHARD_DECODER") == "1":
        assert w.net.decoder_shard is not None and len(w.net.decoder_shard_spans()) == 3
        w.net.gather_decoder_shards()                   # collective: rank 0's flat buffer is complete again before it is saved
        torch.cuda.synchronize()
    if rank == 0:
        torch.save({"param": w.net.store.param.detach().cpu(), "rays_local": w.rays_local, **extra}, os.environ["HOS_DP_OUT"])
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
