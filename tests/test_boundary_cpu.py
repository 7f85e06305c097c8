"""CPU checks of the drop-in boundary: the C-ABI library loads and exports every declared symbol, the Python
mirror exposes the reference's state_dict keys/shapes, and the product never touches the oracle or the CPU."""
import json
import os
import re
import tempfile

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _basedir():
    d = tempfile.mkdtemp(prefix="hos_basedir_")
    with open(os.path.join(d, "transitions_times.json"), "w") as f:
        json.dump({"f0": {"time": 0.4}}, f)
    return d


def test_library_exports_every_declared_symbol():
    from hosnerf_amd import _lib
    header = open(os.path.join(ROOT, "include", "hosrender.h")).read()
    declared = set(re.findall(r"\b(hos_[a-z0-9_]+)\s*\(", header))
    declared.discard("hos_stream_t")
    assert len(declared) >= 25
    lib = _lib.load()                      # loads without a GPU: no compute is called
    for name in sorted(declared):
        assert hasattr(lib, name), f"libhosrender.so does not export {name}"
    assert declared == set(_lib.PROTOTYPES), declared ^ set(_lib.PROTOTYPES)
    assert lib.hos_version() >= 100
    assert lib.hos_error_string(-3) == b"unsupported shape"


def test_argument_validation_without_gpu():
    from hosnerf_amd import _lib
    lib = _lib.load()
    assert lib.hos_linear_fwd(0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0.0, 0.0, 0, 0) == -1       # HOS_E_ARG
    assert lib.hos_resample(0, 0, 64, 8, 64, 0.0, 1.0, 0, 10.0, 0.0, 0, 0, 0.0, 0.1, 1e6, 0, 0, 0, 0) == -1
    # the round-3 entry points of the folded thin MLPs: null pointers -> HOS_E_ARG, never a launch
    assert lib.hos_mlp_chain_pack_fold(0, 0, 0, 0, 75, 36, 0, 0, 0, 0) == -1
    assert lib.hos_mlp_chain_unfold_grad(0, 0, 0, 75, 36, 0, 128, 0, 0) == -1
    assert lib.hos_canonical_fold_pack(0, 128, 0, 0, 384, 0, 0, 256, 63, 64, 256, 0, 0, 0, 0, 0) == -1
    assert lib.hos_canonical_fold_unfold(0, 0, 0, 0, 0, 128, 0, 384, 0, 256, 63, 64, 256, 0, 0, 0, 0, 0, 0) == -1
    assert lib.hos_mlp_chain128_fwd(0, 0, 0, 64, 0, 0, 0, 0, 128, 0, 1024, 0, 0) == -1          # E == NULL is legal (folded form), PE is not
    # shape errors are reported before anything is dereferenced: more feature columns than the 64-wide folded first layer holds
    import ctypes
    buf = (ctypes.c_float * 4)()
    ptrs = (ctypes.c_void_p * 7)(*[ctypes.addressof(buf)] * 7)
    ld = (ctypes.c_int * 7)(*[256] * 7)
    a = ctypes.addressof(buf)
    assert lib.hos_mlp_chain_pack_fold(ctypes.addressof(ptrs), ctypes.addressof(ld), ctypes.addressof(ptrs), a, 75, 65, a, a, a, 0) == -3   # HOS_E_SHAPE


def test_no_cpu_fallback():
    from hosnerf_amd import _lib, ops
    with pytest.raises(_lib.HosLibraryError):
        ops.alpha_weights(torch.ones(2, 4), torch.ones(2, 5), torch.ones(2, 3), True)
    with pytest.raises(_lib.HosLibraryError):
        ops.volumetric_rendering(torch.ones(2, 4, 3), torch.ones(2, 4), 1.0)


def test_state_dict_surface_matches_reference():
    from hosnerf_amd.human_nerf import Network, default_cfg
    from hosnerf_amd.mipnerf360 import MipNeRF360
    ref = json.load(open(os.path.join(ROOT, "tests", "golden", "state_dict_keys.json")))
    d = _basedir()
    bk = {k: list(v.shape) for k, v in MipNeRF360(d, opaque_background=True).state_dict().items()}
    assert bk == ref["state_mipnerf360"]
    hu = {k: list(v.shape) for k, v in Network(default_cfg(d)).state_dict().items()}
    assert hu == ref["human_network"]


def test_flat_store_views_and_padding():
    from hosnerf_amd import synth
    from hosnerf_amd.mipnerf360 import MipNeRF360
    m = MipNeRF360(_basedir(), opaque_background=True)
    sd = synth.background_state_dict(777, 2)
    m.load_state_dict(sd, strict=False)
    for k, v in m.state_dict().items():
        if k in sd:
            assert torch.equal(v, sd[k]) and v.is_contiguous()
    lo, hi = m.flat_param.data_ptr(), m.flat_param.data_ptr() + 4 * m.flat_param.numel()
    for n, p in m.named_parameters():
        assert lo <= p.data_ptr() < hi and lo - lo <= p.grad.data_ptr() - m.flat_grad.data_ptr() < 4 * m.flat_grad.numel(), n
    used = sum(p.numel() for p in m.parameters())
    assert used == 9498630
    assert float(m.flat_param.abs().sum()) == pytest.approx(float(sum(p.detach().abs().sum() for p in m.parameters())), rel=1e-6)
    m.flat_grad.fill_(1.0)
    m.zero_grad()
    assert float(m.flat_grad.abs().max()) == 0


def test_lr_schedule_and_state_selection():
    from hosnerf_amd.mipnerf360 import select_state
    from hosnerf_amd.train import stage1_lr
    assert abs(stage1_lr(0, 500000) - 2e-3 * 0.01) < 1e-12
    assert abs(stage1_lr(500000, 500000) - 2e-5) < 1e-12
    assert stage1_lr(256, 500000) < stage1_lr(512, 500000)
    import numpy as np
    tt = np.array([0.2, 0.4, 0.6], dtype=np.float32)
    assert [select_state(t, tt) for t in (0.1, 0.2, 0.41, 0.6, 0.7)] == [0, 1, 2, 2, 3]


def test_select_model_registry(tmp_path):
    """utils/select_option.py::select_model names: the three stages resolve to modules that expose the
    reference's attribute names; unknown names raise."""
    import json
    from hosnerf_amd.select_option import select_model
    (tmp_path / "transitions_times.json").write_text(json.dumps({"f0": {"time": 0.4}}))
    s1 = select_model("state_mipnerf360", str(tmp_path))
    assert type(s1.model).__name__ == "MipNeRF360" and hasattr(s1, "training_step") and hasattr(s1, "configure_optimizers")
    assert len(list(s1.parameters())) > 0 and abs(s1.learning_rate(0) - 2e-3 * 0.01) < 1e-9
    with pytest.raises(ValueError):
        select_model("nope", str(tmp_path))
    import importlib
    mod = importlib.import_module("hosnerf_amd.plugins.network_amd")
    assert mod.Network.__name__ == "Network"


def test_checkpoint_round_trip_and_stage3_warm_start(tmp_path):
    """S3/run.py:206-212: the stage-3 module is warm-started from a stage-2 checkpoint (`human.*` keys) and a stage-1
    checkpoint (`model.*` keys) with `load_state_dict(ckpt['state_dict'], strict=False)`.  The Lightning-style files are
    built from the reference's key names (tests/golden/state_dict_keys.json)."""
    import json
    import torch
    from hosnerf_amd import synth
    from hosnerf_amd.select_option import load_checkpoint, save_checkpoint, select_model
    (tmp_path / "transitions_times.json").write_text(json.dumps({"f0": {"time": 0.4}}))
    ref = json.load(open(os.path.join(ROOT, "tests", "golden", "state_dict_keys.json")))
    bsd, hsd = synth.background_state_dict(5, 2), synth.human_state_dict(6, 2)
    assert set(bsd) <= set(ref["state_mipnerf360"]) and set(hsd) == set(ref["human_network"])
    s1 = {"state_dict": {"model." + k: v for k, v in bsd.items()}, "global_step": 7}
    s2 = {"state_dict": {"human." + k: v for k, v in hsd.items()}, "global_step": 9}
    torch.save(s1, tmp_path / "bkgd.ckpt")
    torch.save(s2, tmp_path / "human.ckpt")
    lit = select_model("hosnerf", str(tmp_path))
    keys = set(lit.state_dict())
    assert keys == {"model." + k for k in ref["state_mipnerf360"]} | {"human." + k for k in ref["human_network"]}
    flat_ptr = lit.human.flat_param.data_ptr()
    missing, unexpected = load_checkpoint(lit, str(tmp_path / "human.ckpt"))
    assert not unexpected and all(k.startswith("model.") for k in missing)
    missing, unexpected = load_checkpoint(lit, str(tmp_path / "bkgd.ckpt"))
    assert not unexpected and all(k.startswith("human.") or k not in s1["state_dict"] for k in missing)
    got = lit.state_dict()
    assert all(torch.equal(got["human." + k], v) for k, v in hsd.items())
    assert all(torch.equal(got["model." + k], v) for k, v in bsd.items())
    assert lit.human.flat_param.data_ptr() == flat_ptr, "loading copies into the flat store, it does not re-allocate it"
    assert lit.net.model is lit.model and lit.net.human is lit.human
    # our own checkpoint has the same surface and reloads bit-exactly into a fresh module
    save_checkpoint(lit, str(tmp_path / "last.ckpt"), global_step=11)
    ck = torch.load(tmp_path / "last.ckpt", weights_only=False)
    assert set(ck["state_dict"]) == keys and ck["global_step"] == 11
    lit2 = select_model("hosnerf", str(tmp_path))
    missing, unexpected = load_checkpoint(lit2, str(tmp_path / "last.ckpt"), strict=True)
    assert not missing and not unexpected
    assert torch.equal(lit2.human.flat_param, lit.human.flat_param) and torch.equal(lit2.model.flat_param, lit.model.flat_param)


def test_gemm_mode_is_a_process_default_plus_a_per_thread_override():
    """the arithmetic mode must not be process-global state behind a context manager.  `hos_set_gemm_mode` is the
    process default, `hos_set_thread_gemm_mode` / `ops.gemm_mode` an override of the CALLING thread only (no kernel is launched)."""
    import threading
    from hosnerf_amd import _lib, ops
    lib = _lib.load()
    ops.set_gemm_mode(ops.GEMM_PLANES)
    assert ops.get_gemm_mode() == ops.GEMM_PLANES and lib.hos_get_gemm_mode() == ops.GEMM_BF16X3
    seen = {}

    def other():
        seen["before"] = (ops.get_gemm_mode(), lib.hos_get_gemm_mode())
        with ops.gemm_mode(ops.GEMM_BF16X3):
            seen["inside_other"] = (ops.get_gemm_mode(), lib.hos_get_gemm_mode())
            ready.set()
            go.wait(5)
        seen["after"] = (ops.get_gemm_mode(), lib.hos_get_gemm_mode())

    ready, go = threading.Event(), threading.Event()
    with ops.gemm_mode(ops.GEMM_FP32):
        assert ops.get_gemm_mode() == ops.GEMM_FP32 and lib.hos_get_gemm_mode() == ops.GEMM_FP32
        t = threading.Thread(target=other)
        t.start()
        assert ready.wait(5)
        assert ops.get_gemm_mode() == ops.GEMM_FP32 and lib.hos_get_gemm_mode() == ops.GEMM_FP32      # the other thread's switch is invisible here
        with ops.gemm_mode(ops.GEMM_BF16X3):                                                              # nesting restores the enclosing override
            assert lib.hos_get_gemm_mode() == ops.GEMM_BF16X3
        assert lib.hos_get_gemm_mode() == ops.GEMM_FP32
        go.set()
        t.join()
    assert seen["before"] == (ops.GEMM_PLANES, ops.GEMM_BF16X3)           # this thread's fp32 override was invisible there
    assert seen["inside_other"] == (ops.GEMM_BF16X3, ops.GEMM_BF16X3) and seen["after"] == (ops.GEMM_PLANES, ops.GEMM_BF16X3)
    assert ops.get_gemm_mode() == ops.GEMM_PLANES and lib.hos_get_gemm_mode() == ops.GEMM_BF16X3
    assert lib.hos_set_thread_gemm_mode(7) != 0


def test_first_deconv_layer_compact_copy_round_trips_through_state_dict():
    """The first ConvTranspose3d of the volume decoder meets one input voxel: only the 2x2x2 centre taps of its 4x4x4 kernel are
    live.  The network keeps them in a compact copy (the reference-shaped parameter is inactive during training) and
    synchronises on load / save: a state dict goes in and comes out unchanged, and a change of the live copy shows in
    `state_dict()` at exactly the live taps."""
    from hosnerf_amd import synth
    from hosnerf_amd.human_nerf import Network, default_cfg
    net = Network(default_cfg(_basedir()))
    sd = synth.human_state_dict(777, 2)
    net.load_state_dict(sd, strict=True)
    k = "mweight_vol_decoder.decoder.block_conv.0.weight"
    out = net.state_dict()
    assert all(torch.equal(out[n], sd[n]) for n in sd)
    full = sd[k].reshape(1024, 512, 64)
    comp = net._w0c.view(net.store.param).view(1024, 8, 512)
    taps = Network._LIVE_TAPS
    assert taps == [21, 22, 25, 26, 37, 38, 41, 42]
    for j, t in enumerate(taps):
        assert torch.equal(comp[:, j, :], full[:, :, t])
    comp[:, 3, :] += 1.0
    out2 = net.state_dict()[k].reshape(1024, 512, 64)
    assert torch.equal(out2[:, :, taps[3]], full[:, :, taps[3]] + 1.0)
    dead = [t for t in range(64) if t not in taps]
    assert torch.equal(out2[:, :, dead], full[:, :, dead])
    # optimiser / norm / zeroing skip the reference-shaped parameter
    (off, n), = net.store.inactive
    net.flat_grad.fill_(1.0)
    net.zero_grad()
    assert float(net.flat_grad[off:off + n].min()) == 1.0 and float(net.flat_grad[:off].abs().max()) == 0 and float(net.flat_grad[off + n:].abs().max()) == 0


def test_comm_library_exports_every_declared_symbol():
    """libhoscomm.so (include/hoscomm.h, SURVEY 8(b).6: `hos_allreduce_*` over RCCL): loads without a GPU, exports every declared
    entry point, and rejects null arguments without touching RCCL."""
    from hosnerf_amd import comm
    header = open(os.path.join(ROOT, "include", "hoscomm.h")).read()
    declared = set(re.findall(r"\b(hos_[a-z0-9_]+)\s*\(", header))
    assert declared == set(comm.PROTOTYPES), declared ^ set(comm.PROTOTYPES)
    lib = comm.load()
    for name in sorted(declared):
        assert hasattr(lib, name), f"libhoscomm.so does not export {name}"
    assert lib.hos_allreduce_avg_f32(0, 0, 16, 0) == -1 and lib.hos_comm_init(0, 2, 0, 0) == -1
    assert lib.hos_allgather_f32(0, 0, 0, 4, 0) == -1 and lib.hos_allreduce_avg_f32_spans(0, 0, 0, 1, 0) == -1


def test_group_backward_entry_points_validate_without_gpu():
    """hos_mlp_chain_bwd* (round 4): configuration tables and argument checks, no launch."""
    from hosnerf_amd import _lib
    lib = _lib.load()
    assert [lib.hos_mlp_chain_bwd_steps(c) for c in range(4)] == [2, 3, 3, 3] and lib.hos_mlp_chain_bwd_steps(4) == -1
    # LDS images: (hi, lo) planes of [N_][K_ + 16] bf16, rounded up to whole 8 KB copy rounds
    assert lib.hos_mlp_chain_bwd_image_bytes(0, 0) == 24576 and lib.hos_mlp_chain_bwd_image_bytes(0, 1) == 73728
    assert lib.hos_mlp_chain_bwd_image_bytes(1, 0) == 40960 and lib.hos_mlp_chain_bwd_image_bytes(2, 5) == 0
    # slabs: 256 workgroups x sum over steps of (N_ * K_ + N_) floats from 16 384 rows on
    assert lib.hos_mlp_chain_bwd_ws_floats(0, 262144) == 256 * ((32 * 128 + 32) + (128 * 128 + 128))
    assert lib.hos_mlp_chain_bwd_ws_floats(2, 64) == 1 * (2 * (128 * 128 + 128) + (128 * 64 + 128))
    assert lib.hos_mlp_chain_bwd_ws_floats(9, 64) == 0
    assert lib.hos_mlp_chain_bwd_pack(0, 0, 0, 0, 0, 0, 0, 0, 0) == -1 and lib.hos_mlp_chain_bwd_pack(9, 0, 0, 0, 0, 0, 0, 0, 0) == -1
    assert lib.hos_mlp_chain_bwd(0, 0, 32, 1024, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0) == -1
    assert lib.hos_mlp_chain_bwd(7, 0, 32, 1024, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0) == -1


def test_lpips_entry_points_validate_without_gpu():
    """hos_lpips.hip (round 4): null pointers / odd sizes are rejected before any launch; the Python module refuses to run unloaded."""
    from hosnerf_amd import _lib
    from hosnerf_amd.lpips import LPIPS, VGG16_CFG, TAP_AFTER_CONV
    lib = _lib.load()
    assert lib.hos_lpips_prep(0, 16, 0, 0) == -1 and lib.hos_im2col3x3(0, 1, 4, 4, 3, 0, 32, 0) == -1
    assert lib.hos_col2im3x3(0, 32, 1, 4, 4, 3, 0, 0, 0) == -1 and lib.hos_maxpool2x2_fwd(0, 1, 4, 4, 3, 0, 0) == -1
    assert lib.hos_lpips_head_fwd(0, 0, 1, 4, 64, 0.25, 0, 0) == -1 and lib.hos_lpips_head_bwd(0, 0, 1, 4, 64, 0.25, 0, 0, 0, 0) == -1
    assert lib.hos_unpack_patches_fwd(0, 0, 0, 1.0, 4, 0, 0) == -1 and lib.hos_unpack_patches_bwd(0, 0, 4, 1.0, 1.0, 1.0, 0, 0) == -1
    assert lib.hos_lpips_finish(0, 2, 0, 0) == -1
    assert sum(1 for v in VGG16_CFG if v != "M") == 13 and TAP_AFTER_CONV == (1, 3, 6, 9, 12)
    assert not LPIPS().ready()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_sharded_decoder_spans_partition_the_layers(world):
    """`Network.shard_decoder` (round 5) on the host: over the ranks the owned rows of every sharded layer are disjoint,
    contiguous and cover the layer; a rank's active spans + its inactive spans are the whole flat buffer; the rows a rank does not
    own are exactly what it marks inactive; sharding twice raises.  No kernel runs: this is the bookkeeping the optimiser spans,
    the gradient-norm completion and `gather_decoder_shards` rely on."""
    from hosnerf_amd.human_nerf import Network, default_cfg

    class Comm:                     # what shard_decoder reads of a train.ShardComm
        def __init__(self, rank):
            self.rank, self.world = rank, world

    owned, layers = [], None
    net = Network(default_cfg(_basedir()))
    before = list(net.store.inactive)
    for rank in range(world):
        net.store.inactive[:] = before                     # (one module, re-sharded as every rank in turn: construction is the slow part)
        net.decoder_shard, net._shard_layers = None, ()
        net.shard_decoder(Comm(rank))
        spans = net.decoder_shard_spans()
        assert len(spans) == 3                                                 # the first three transposed convolutions
        rows = [net._shard_rows(n) for n in range(3)]                          # (offset, floats per input-channel row, Cin)
        layers = layers or rows
        assert rows == layers
        for (off, n), (loff, row, cin) in zip(spans, rows):
            cs = cin // world
            assert n == cs * row and off == loff + rank * cs * row            # a contiguous range of input-channel rows
        owned.append(spans)
        # active + inactive = the whole buffer, no overlap
        act = net.store.active_spans()
        ina = sorted(net.store.inactive)
        covered = sorted(act + ina)
        pos = 0
        for off, n in covered:
            assert off >= pos
            pos = max(pos, off + n)
        assert pos == net.store.size and sum(n for _, n in act) + sum(n for _, n in _merge(ina)) == net.store.size
        # the new inactive spans are the other ranks' rows of the three layers
        new = sum(n for _, n in _merge(sorted(set(ina) - set(before))))
        assert new == sum(row * cin for _, row, cin in rows) - sum(n for _, n in spans)
        with pytest.raises(RuntimeError):
            net.shard_decoder(Comm(rank))
    for k, (loff, row, cin) in enumerate(layers):                             # over the ranks: disjoint cover of each layer
        pieces = sorted(owned[r][k] for r in range(world))
        assert pieces[0][0] == loff and pieces[-1][0] + pieces[-1][1] == loff + row * cin
        for (o0, n0), (o1, _) in zip(pieces, pieces[1:]):
            assert o0 + n0 == o1


def _merge(spans):
    out = []
    for off, n in sorted(spans):
        if out and off <= out[-1][0] + out[-1][1]:
            out[-1] = (out[-1][0], max(out[-1][1], off + n - out[-1][0]))
        else:
            out.append((off, n))
    return out


def test_lazy_spans_cover_exactly_the_parameters_that_can_be_without_a_gradient():
    """Round 6 (`hos_adam_lazy_prepare`): the spans the optimiser treats like torch's "grad is None" parameters are exactly the state
    embeddings (one span each) and the pose decoder (one span: its parameters start together at the kick-in iteration), every span is
    float4-aligned, and the optimiser's learning-rate ranges are cut at them without losing or duplicating a float."""
    from hosnerf_amd.human_nerf import Network, default_cfg
    from hosnerf_amd.mipnerf360 import MipNeRF360
    from hosnerf_amd.train import FusedAdam, _split_at_lazy, human_lr_ranges
    d = _basedir()
    for mod, lazy_names in ((MipNeRF360(d, opaque_background=True), ("stateembeds",)), (Network(default_cfg(d)), ("stateembeds", "pose_decoder"))):
        spans = sorted(mod.lazy_param_spans())
        base = mod.flat_param.data_ptr()
        inside = lambda off, n: any(lo <= off and off + n <= lo + ln for lo, ln in spans)
        for name, p in mod.named_parameters():
            off = (p.data_ptr() - base) // 4
            lazy = any(k in name for k in lazy_names)
            assert inside(off, p.numel()) == lazy, name
            if "stateembeds" in name:
                assert (off, p.numel()) in spans, name                  # an embedding IS a span (its own step count)
        assert all(lo % 4 == 0 and ln % 4 == 0 for lo, ln in spans)
        assert all(a[0] + a[1] <= b[0] for a, b in zip(spans, spans[1:]))
        opt = FusedAdam(mod, lr=1e-3, lr_ranges=human_lr_ranges(mod) if isinstance(mod, Network) else None)
        assert opt.lazy_spans == spans and opt.lazy_state.shape == (len(spans), 8)
        got = sorted((off, n) for off, n, _ in opt.lr_ranges)
        assert all(a[0] + a[1] <= b[0] for a, b in zip(got, got[1:]))
        want = sum(n for _, n in mod.store.active_spans())
        assert sum(n for _, n in got) == want                           # nothing lost, nothing twice
        # every lazy span lies in exactly one lazy range, as row (k - first) of it; adjacent embeddings share ONE range (rows)
        covered = []
        for (off, n, _), k in zip(opt.lr_ranges, opt._range_lazy):
            if k is not None:
                first, rows, rl = k
                assert n == rows * rl
                covered += [(off + i * rl, rl) for i in range(rows)]
                assert spans[first:first + rows] == covered[-rows:]
        assert covered == spans
        assert len(opt.lr_ranges) <= 8                                  # whatever the number of states
    # the cutter itself: a run of adjacent equal spans stays one range with a row length
    r, idx = _split_at_lazy([(0, 100, 1.0), (100, 60, 0.1)], [(8, 4), (12, 4), (96, 4), (120, 40)])
    assert r == [(0, 8, 1.0), (8, 8, 1.0), (16, 80, 1.0), (96, 4, 1.0), (100, 20, 0.1), (120, 40, 0.1)]
    assert idx == [None, (0, 2, 4), None, (2, 1, 4), None, (3, 1, 40)]
    r, idx = _split_at_lazy([(0, 16, 1.0), (16, 16, 0.1)], [(8, 4), (12, 4), (16, 4)])       # a run cut by a learning-rate boundary
    assert r == [(0, 8, 1.0), (8, 8, 1.0), (16, 4, 0.1), (20, 12, 0.1)] and idx == [None, (0, 2, 4), (2, 1, 4), None]
    with pytest.raises(ValueError):
        _split_at_lazy([(0, 100, 1.0), (100, 60, 0.1)], [(98, 4)])      # a span may not straddle two learning rates
