"""CPU (-m "not gpu"): the multi-GPU path of the parallel denoise loop.
  * shard_units / group_units invariants (every (window, CFG half) unit exactly once, contiguous, balanced),
  * world_size-2 gloo run of ParallelDenoiser with kernel test doubles: both ranks end bit-identical and equal to the
    single-process run and (within fp16 rounding of the exchanged predictions) to the oracle loop."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import fake_ops


def test_shard_units_invariants():
    from musev_amd.pipelines.parallel_denoise import Unit, group_units, shard_units
    for n_win in (1, 2, 3, 6, 12):
        for halves in (1, 2):
            for world in (1, 2, 3, 4, 8):
                shards = shard_units(n_win, halves, world)
                flat = [u for s in shards for u in s]
                assert flat == [Unit(w, h) for w in range(n_win) for h in range(halves)]
                sizes = [len(s) for s in shards]
                assert max(sizes) - min(sizes) <= 1
                for s in shards:
                    for wdw, hs in group_units(s):
                        assert hs == sorted(hs) and len(hs) <= halves
    # config 4: 12 windows x 2 halves over 8 ranks -> 3 units per rank (ideal 8x)
    assert [len(s) for s in shard_units(12, 2, 8)] == [3] * 8


def _patch(monkeypatch_like):
    from musev_amd import ops
    for name in ("window_gather", "window_scatter_add", "window_units_reduce", "cfg_ddim_step", "cfg_affine_step"):
        monkeypatch_like(ops, name, getattr(fake_ops, name))


def _run_loop(group=None, scheduler=None, frames=20, schedule="uniform", window=8, overlap=2, n_cond=1, **loop_kw):
    from musev_amd import ops
    from musev_amd.pipelines.parallel_denoise import ParallelDenoiser
    saved = {n: getattr(ops, n) for n in ("window_gather", "window_scatter_add", "window_units_reduce", "cfg_ddim_step", "cfg_affine_step")}
    try:
        _patch(setattr)
        ParallelDenoiser._device_check = False
        g = torch.Generator().manual_seed(0)
        lat = torch.randn(1, 4, frames, 4, 4, generator=g)
        cond = torch.randn(1, 4, n_cond, 4, 4, generator=g)
        prompt = torch.randn(2, 7, 16, generator=g)
        den = ParallelDenoiser(fake_ops.FakeUNet(), scheduler=scheduler, context_frames=window, context_overlap=overlap,
                               context_schedule=schedule)
        return den(lat, prompt, num_inference_steps=5, guidance_scale=3.5, condition_latents=cond, group=group, **loop_kw), (lat, cond, prompt)
    finally:
        ParallelDenoiser._device_check = True
        for n, f in saved.items():
            setattr(ops, n, f)



def _plain(v):
    """tensors -> numpy before they go through a multiprocessing Manager: numpy arrays are pickled BY VALUE, torch tensors by shared-
    memory file descriptor -- and a descriptor whose owner (the worker) has exited by the time the parent asks for it raises EOFError"""
    if torch.is_tensor(v):
        return ("__tensor__", v.detach().cpu().numpy())
    if isinstance(v, (list, tuple)):
        return type(v)(_plain(e) for e in v)
    return v


def _unplain(v):
    if isinstance(v, tuple) and len(v) == 2 and isinstance(v[0], str) and v[0] == "__tensor__":
        return torch.from_numpy(v[1])
    if isinstance(v, (list, tuple)):
        return type(v)(_unplain(e) for e in v)
    return v


def _worker(rank, world, port, ret, kw=None):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    out, _ = _run_loop(group=dist.group.WORLD, **(kw or {}))
    ret[rank] = _plain(out.clone())
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_multi_rank_gloo_matches_single_process_and_oracle(world):
    """world 2: 8 units -> 4 + 4; world 3: 3 + 3 + 2 (the last rank's unused exchange slot must not be accumulated)"""
    single, (lat, cond, prompt) = _run_loop(None)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    ret = {k: _unplain(v) for k, v in dict(ret).items()}
    for r in range(1, world):
        assert torch.equal(ret[0], ret[r]), "replicated latents diverged between ranks"
    # the 2-rank run exchanges fp16 predictions exactly like the 1-rank run consumes them -> identical results
    assert torch.equal(ret[0], single)
    from oracle import pipeline as opipe
    fake = fake_ops.FakeUNet()
    want = opipe.denoise_loop(fake.nchw, lat, prompt, num_inference_steps=5, guidance_scale=3.5, condition_latents=cond,
                              context_frames=8, context_overlap=2, motion_speed=8.0)
    assert (single - want).abs().max().item() < 5e-3  # fp16 window inputs / predictions vs the fp32 oracle loop


def test_condition_frames_at_head_and_tail_follow_the_reference():
    """vision_condition_latent_index = [0, -1] (the CLI's condition_images_index): the loop against the oracle
    loop, whose index logic is pinned to the reference's own functions (tests/golden/reference_condition_index.json).  One window:
    slot 0 = condition frame 0, slot 1 = zeros, the tail condition frame overwritten by the last generated frame, the UNet told that
    slots 0 and n_cond + T - 1 are condition frames, the final re-insert at positions 0 and n_cond + T - 1.  More than one window:
    IndexError, as the reference's index_copy_ raises."""
    from oracle import pipeline as opipe
    for vis, n_cond in (([0, -1], 2), ([-1], 1), ([0, 3], 2), (None, 2)):
        got, (lat, cond, prompt) = _run_loop(None, frames=8, n_cond=n_cond, vision_condition_latent_index=vis)
        fake = fake_ops.FakeUNet()
        want = opipe.denoise_loop(fake.nchw, lat, prompt, num_inference_steps=5, guidance_scale=3.5, condition_latents=cond,
                                  context_frames=8, context_overlap=2, motion_speed=8.0, vision_condition_latent_index=vis)
        assert got.shape == want.shape == (1, 4, n_cond + 8, 4, 4)
        assert (got - want).abs().max().item() < 5e-3, vis
        v_res, l_res = opipe.condition_indices(n_cond, 8, vis)
        assert torch.equal(got[:, :, v_res], cond), "the condition frames come back at their positions, untouched"
    with pytest.raises(IndexError):
        _run_loop(None, frames=20, n_cond=2, vision_condition_latent_index=[0, -1])
    with pytest.raises(IndexError):  # the oracle (= torch's index_copy_ as in the reference) refuses the same call
        lat = torch.randn(1, 4, 20, 4, 4)
        opipe.denoise_loop(fake_ops.FakeUNet().nchw, lat, torch.randn(2, 7, 16), num_inference_steps=5, guidance_scale=3.5,
                           condition_latents=torch.randn(1, 4, 2, 4, 4), context_frames=8, context_overlap=2, vision_condition_latent_index=[0, -1])


def test_euler_loop_matches_oracle_loop():
    """the loop with the Euler-discrete scheduler (scale_model_input on the gathered latents, fused affine step) against
    the oracle loop restating pipeline_controlnet.py:1846-2147 + scheduling_euler_discrete.py, kernel test doubles on CPU"""
    from musev_amd.schedulers import EulerDiscreteScheduler
    from oracle import pipeline as opipe
    got, (lat, cond, prompt) = _run_loop(None, scheduler=EulerDiscreteScheduler())
    fake = fake_ops.FakeUNet()
    want = opipe.denoise_loop(fake.nchw, lat, prompt, num_inference_steps=5, guidance_scale=3.5, condition_latents=cond,
                              context_frames=8, context_overlap=2, motion_speed=8.0, scheduler="euler")
    assert got.shape == want.shape
    # Euler latents live in sigma-scaled space (sigma_max = 14.6): compare relative to that scale
    assert (got - want).abs().max().item() < 5e-3 * 14.6


@pytest.mark.parametrize("method", ["linear", "two_stage", "fix_two_stage"])
def test_guidance_schedule_in_the_loop(method):
    """guidance_scale -> guidance_scale_end over the steps (pipeline_controlnet.py:1718-1723, consumed at :2103)"""
    from oracle import pipeline as opipe
    got, (lat, cond, prompt) = _run_loop(None, guidance_scale_end=1.5, guidance_scale_method=method)
    fake = fake_ops.FakeUNet()
    want = opipe.denoise_loop(fake.nchw, lat, prompt, num_inference_steps=5, guidance_scale=3.5, condition_latents=cond,
                              context_frames=8, context_overlap=2, motion_speed=8.0, guidance_scale_end=1.5,
                              guidance_scale_method=method)
    const = opipe.denoise_loop(fake.nchw, lat, prompt, num_inference_steps=5, guidance_scale=3.5, condition_latents=cond,
                               context_frames=8, context_overlap=2, motion_speed=8.0)
    assert (got - want).abs().max().item() < 5e-3
    assert (want - const).abs().max().item() > 1e-2, "the schedule must matter for the check to mean anything"


# ---- windows of unequal length: `uniform_v2`, the CLI default (scripts/inference/text2video.py:499-505) ---------------------
def _spawn(world, kw):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ret = mp.Manager().dict()
    mp.spawn(_worker, args=(world, port, ret, kw), nprocs=world, join=True)
    ret = {k: _unplain(v) for k, v in dict(ret).items()}
    return ret


@pytest.mark.parametrize("frames,window,overlap", [(24, 12, 4), (22, 8, 2), (48, 12, 4)])
def test_uniform_v2_short_last_window_matches_oracle(frames, window, overlap):
    """T = 24 / 48 with window 12, overlap 4 end in an 8-frame window (VERDICT r1 'missing' 2); the reference runs every
    window as its own UNet call of whatever length (pipeline_controlnet.py:1900-1946)"""
    from musev_amd.pipelines.context import prepare_global_context
    from oracle import pipeline as opipe
    wins = [c[0] for c in prepare_global_context("uniform_v2", 5, frames, window, 1, overlap, 1)]
    assert len({len(w) for w in wins}) > 1, "the case must contain windows of unequal length"
    kw = dict(frames=frames, schedule="uniform_v2", window=window, overlap=overlap)
    got, (lat, cond, prompt) = _run_loop(None, **kw)
    fake = fake_ops.FakeUNet()
    want = opipe.denoise_loop(fake.nchw, lat, prompt, num_inference_steps=5, guidance_scale=3.5, condition_latents=cond,
                              context_frames=window, context_overlap=overlap, context_schedule="uniform_v2", motion_speed=8.0)
    assert got.shape == want.shape
    assert (got - want).abs().max().item() < 5e-3


def test_uniform_v2_multi_rank_gloo():
    """3 ranks over 3 windows x 2 halves with a short last window: slots sized by the longest window, the short unit's tail
    of its exchange slot is never accumulated; ranks bit-identical and equal to the single-process run"""
    kw = dict(frames=24, schedule="uniform_v2", window=12, overlap=4)
    single, _ = _run_loop(None, **kw)
    for world in (2, 3):
        ret = _spawn(world, kw)
        for r in range(1, world):
            assert torch.equal(ret[0], ret[r])
        assert torch.equal(ret[0], single)


def test_config4_world8_gloo_wraparound_window():
    """BASELINE config 4's schedule -- 96 frames, window 12, overlap 4, `uniform`: 12 windows incl. the wrap-around window
    [88..95, 0..3] -- as 24 units over 8 ranks (3 per rank) under gloo; tiny latents, kernel test doubles"""
    from musev_amd.pipelines.context import prepare_global_context
    wins = [c[0] for c in prepare_global_context("uniform", 5, 96, 12, 1, 4, 1)]
    assert len(wins) == 12 and wins[-1] == list(range(88, 96)) + [0, 1, 2, 3]
    kw = dict(frames=96, window=12, overlap=4)
    single, (lat, cond, prompt) = _run_loop(None, **kw)
    ret = _spawn(8, kw)
    for r in range(1, 8):
        assert torch.equal(ret[0], ret[r]), "replicated latents diverged between ranks"
    assert torch.equal(ret[0], single)
    from oracle import pipeline as opipe
    want = opipe.denoise_loop(fake_ops.FakeUNet().nchw, lat, prompt, num_inference_steps=5, guidance_scale=3.5,
                              condition_latents=cond, context_frames=12, context_overlap=4, motion_speed=8.0)
    assert (single - want).abs().max().item() < 5e-3


def test_window_visiting_a_frame_twice_is_refused():
    """`uniform` with context_stride > 1 wraps strided windows modulo T; T = 20, window 12, hop 2 visits frames twice
    (ADVICE r1): refused loudly instead of racing in the scatter-add"""
    with pytest.raises(NotImplementedError):
        from musev_amd import ops
        from musev_amd.pipelines.parallel_denoise import ParallelDenoiser
        ParallelDenoiser._device_check = False
        try:
            den = ParallelDenoiser(fake_ops.FakeUNet(), context_frames=12, context_overlap=4, context_stride=2)
            g = torch.Generator().manual_seed(0)
            den(torch.randn(1, 4, 20, 4, 4, generator=g), torch.randn(2, 7, 16, generator=g), num_inference_steps=2, guidance_scale=3.5)
        finally:
            ParallelDenoiser._device_check = True


# ---- side models on one rank + one broadcast -----------------------------------------------------------------------
def _side_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from musev_amd.pipelines.conditioning import get_referencenet_emb_sharded
    calls = []

    def fake_refnet(sample, timestep, encoder_hidden_states, num_frames, return_ndim):
        calls.append(1)
        g = torch.Generator().manual_seed(5)
        down = [torch.randn(2, c, 1, s_, s_, generator=g).half() for c, s_ in ((320, 8), (320, 8), (640, 4), (1280, 2))]
        return down, torch.randn(2, 1280, 1, 2, 2, generator=g).half(), None

    lat = torch.zeros(2, 4, 8, 8)
    tok = torch.zeros(2, 4, 768)
    down, mid, sa = get_referencenet_emb_sharded(fake_refnet if rank == 1 else None, lat if rank == 1 else None, 1, tok, None,
                                                 group=dist.group.WORLD, src=1, device=torch.device("cpu"))
    ret[rank] = _plain(([d.clone() for d in down], mid.clone(), len(calls)))
    dist.destroy_process_group()


def test_side_model_outputs_are_computed_on_one_rank_and_broadcast():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ret = mp.Manager().dict()
    mp.spawn(_side_worker, args=(3, port, ret), nprocs=3, join=True)
    ret = {k: _unplain(v) for k, v in dict(ret).items()}
    assert [ret[r][2] for r in range(3)] == [0, 1, 0], "the side model must run on the source rank only"
    for r in (0, 2):
        assert all(torch.equal(a, b) for a, b in zip(ret[r][0], ret[1][0])) and torch.equal(ret[r][1], ret[1][1])
    assert [tuple(d.shape) for d in ret[0][0]] == [(2, 320, 1, 8, 8), (2, 320, 1, 8, 8), (2, 640, 1, 4, 4), (2, 1280, 1, 2, 2)]
