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
    for name in ("window_gather", "window_scatter_add", "cfg_ddim_step", "cfg_affine_step"):
        monkeypatch_like(ops, name, getattr(fake_ops, name))


def _run_loop(group=None, scheduler=None, **loop_kw):
    from musev_amd import ops
    from musev_amd.pipelines.parallel_denoise import ParallelDenoiser
    saved = {n: getattr(ops, n) for n in ("window_gather", "window_scatter_add", "cfg_ddim_step", "cfg_affine_step")}
    try:
        _patch(setattr)
        ParallelDenoiser._device_check = False
        g = torch.Generator().manual_seed(0)
        lat = torch.randn(1, 4, 20, 4, 4, generator=g)
        cond = torch.randn(1, 4, 1, 4, 4, generator=g)
        prompt = torch.randn(2, 7, 16, generator=g)
        den = ParallelDenoiser(fake_ops.FakeUNet(), scheduler=scheduler, context_frames=8, context_overlap=2)
        return den(lat, prompt, num_inference_steps=5, guidance_scale=3.5, condition_latents=cond, group=group, **loop_kw), (lat, cond, prompt)
    finally:
        ParallelDenoiser._device_check = True
        for n, f in saved.items():
            setattr(ops, n, f)


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    out, _ = _run_loop(group=dist.group.WORLD)
    ret[rank] = out.clone()
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
    for r in range(1, world):
        assert torch.equal(ret[0], ret[r]), "replicated latents diverged between ranks"
    # the 2-rank run exchanges fp16 predictions exactly like the 1-rank run consumes them -> identical results
    assert torch.equal(ret[0], single)
    from oracle import pipeline as opipe
    fake = fake_ops.FakeUNet()
    want = opipe.denoise_loop(fake.nchw, lat, prompt, num_inference_steps=5, guidance_scale=3.5, condition_latents=cond,
                              context_frames=8, context_overlap=2, motion_speed=8.0)
    assert (single - want).abs().max().item() < 5e-3  # fp16 window inputs / predictions vs the fp32 oracle loop


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
