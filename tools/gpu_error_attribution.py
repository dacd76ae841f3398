"""Where does the HIP forward's ~5e-3 |delta eps|max come from?  (run on the GPU box.)

    python tools/gpu_error_attribution.py [--case musev_cfg2_loop] [--out gpurun_out/r04_attribution.json]

Two measurements at BASELINE config-2 size (512x512, 12 + 1 frames, CFG batch 2, full-width `musev`, the loop tests' noise-predictor
weights, the loop's first UNet input):

1. ROUNDING CLASSES IN THE ORACLE.  The fp32 oracle (oracle/unet3d.py, here evaluated by torch on the GPU in fp32) is re-run with ONE
   class of values rounded to fp16 at a time through oracle.unet3d.HOOK -- weights, inputs, projection / convolution outputs,
   GroupNorm / LayerNorm outputs, the residual stream on the network's identity path ("stream_outer"), the residual adds inside the
   transformer blocks ("stream_inner"), attention probabilities (round-to-nearest and round-toward-zero) ... -- then with all of
   them (an fp16 evaluation with fp32 accumulation: what any fp16 implementation does), and with all but one.  |delta eps|max of each
   run against the unrounded oracle says which roundings the output is sensitive to, independent of any kernel.
2. THE HIP FORWARD against the same oracle output, at the block boundaries both sides expose (UNet3DConditionModel._collect /
   unet3d_forward(collect=)) and at the output, for the four {column statistics, LayerNorm fold} combinations.

The oracle is used as the checker (test infrastructure); nothing here is on the product path."""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402


def rtz16(x: torch.Tensor) -> torch.Tensor:
    """round toward zero to fp16 precision (v_cvt_pkrtz_f16_f32), for values in fp16's normal range"""
    y = (x.float().contiguous().view(torch.int32) & ~0x1FFF).view(torch.float32)
    return torch.where(x.abs() < 6.2e-5, x.half().float(), y)


def rtn16(x: torch.Tensor) -> torch.Tensor:
    return x.half().float()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--case", default="musev_cfg2_loop")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r04_attribution.json"))
    ap.add_argument("--small", action="store_true", help="2-level net at 16x16 latents (smoke run of the tool itself)")
    ap.add_argument("--no-hip", action="store_true")
    ap.add_argument("--only", default=None, help="substring: run only the oracle rounding runs whose name contains it (and 'ALL (')")
    args = ap.parse_args()
    from golden_cases import LOOP_CASES_AT_SIZE, loop_case_inputs, loop_case_state_dict, loop_case_unet_kwargs
    from oracle import pipeline as opipe
    from oracle import unet3d
    dev = torch.device("cuda", 0) if torch.cuda.is_available() else torch.device("cpu")
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    case = dict(LOOP_CASES_AT_SIZE[args.case])
    if args.small:
        case.update(arch=dict(block_out_channels=(320, 640), layers_per_block=1, down_block_types=("CrossAttnDownBlock3D", "DownBlock3D"),
                              up_block_types=("UpBlock3D", "CrossAttnUpBlock3D")), h=16, w=16, T=5)
    cfg, sd = loop_case_state_dict(case)
    latents, cond, prompt = loop_case_inputs(case)
    side = loop_case_unet_kwargs(case, cfg)
    sched = opipe.DDIMOracle()
    sched.set_timesteps(case["num_inference_steps"])
    t0 = int(sched.timesteps[0])
    n_cond = cond.shape[2]
    x = torch.cat([cond, latents], dim=2).repeat(2, 1, 1, 1, 1)  # [uncond | cond] halves see the same latents (pipeline_controlnet.py:1908)
    T = x.shape[2]
    kw = dict(sample_index=torch.arange(n_cond, T), vision_conditon_frames_sample_index=torch.arange(n_cond), sample_frame_rate=8, **side)

    def to_dev(v):
        if torch.is_tensor(v):
            return v.to(dev)
        if isinstance(v, (list, tuple)):
            return [to_dev(u) for u in v]
        return v

    sd_dev = {k: v.to(dev) for k, v in sd.items()}
    sd_16 = {k: rtn16(v) for k, v in sd_dev.items()}
    xd, pd_ = x.to(dev), prompt.to(dev)
    kwd = {k: to_dev(v) for k, v in kw.items()}

    def oracle(classes=(), p_round=rtn16, weights16=False, inputs16=False, collect=None, carry_channels=None):
        """carry_channels: widths of the residual stream whose identity path keeps the unrounded sum (a two-fp16 carry) while the
        layers read its fp16 rounding ("stream_read")"""
        cl = set(classes)

        def hook(kind, v):
            if kind == "attn_p":
                return p_round(v) if "attn_p" in cl else v
            if kind == "stream_outer" and carry_channels is not None and (v.shape[1] in carry_channels or v.shape[-1] in carry_channels):
                return v
            return rtn16(v) if kind in cl else v

        unet3d.HOOK = hook if cl else None
        try:
            with torch.no_grad():
                out = unet3d.unet3d_forward(sd_16 if weights16 else sd_dev, cfg, rtn16(xd) if inputs16 else xd, t0,
                                            rtn16(pd_) if inputs16 else pd_, collect=collect, **kwd)
        finally:
            unet3d.HOOK = None
        return out.float()

    def stats(a, ref):
        d = (a - ref).abs()
        return {"max": d.max().item(), "rms": d.pow(2).mean().sqrt().item(), "p999": d.flatten().kthvalue(int(0.999 * d.numel())).values.item()}

    t_start = time.time()
    ref_taps = {}
    ref = oracle(collect=ref_taps)
    report = {"case": args.case, "small": bool(args.small), "timestep": t0, "eps_absmax": ref.abs().max().item(), "eps_rms": ref.pow(2).mean().sqrt().item(),
              "device": str(dev), "classes": {}, "hip": {}}
    print(f"oracle fp32 on {dev}: |eps|max {report['eps_absmax']:.3f} rms {report['eps_rms']:.3f}  ({time.time() - t_start:.0f} s)", flush=True)
    ALL = ["emb", "conv_in", "gemm", "gn", "gn_out", "ln", "stream_outer", "stream_read", "stream_inner", "attn_q", "attn_p", "attn_o"]
    runs = [("weights", dict(weights16=True)), ("inputs", dict(inputs16=True))]
    runs += [(c, dict(classes=[c])) for c in ALL]
    runs += [("attn_p_rtz", dict(classes=["attn_p"], p_round=rtz16)),
             ("ALL (fp16 storage everywhere, fp32 accumulate)", dict(classes=ALL, weights16=True, inputs16=True)),
             ("ALL, P toward zero", dict(classes=ALL, weights16=True, inputs16=True, p_round=rtz16)),
             ("ALL but ln (every LayerNorm folded)", dict(classes=[c for c in ALL if c != "ln"], weights16=True, inputs16=True)),
             ("ALL but stream_outer", dict(classes=[c for c in ALL if c != "stream_outer"], weights16=True, inputs16=True)),
             ("ALL but stream_outer, stream_inner", dict(classes=[c for c in ALL if not c.startswith("stream")], weights16=True, inputs16=True)),
             ("ALL but stream_outer, conv_in, gn_out", dict(classes=[c for c in ALL if c not in ("stream_outer", "conv_in", "gn_out")], weights16=True, inputs16=True)),
             ("ALL, two-fp16 carry on the outer stream (layers read hi)", dict(classes=ALL, weights16=True, inputs16=True, carry_channels=(320, 640, 1280))),
             ("ALL, two-fp16 carry at level 0 only (C = 320)", dict(classes=ALL, weights16=True, inputs16=True, carry_channels=(320,))),
             ("ALL but weights", dict(classes=ALL, inputs16=True)),
             ("ALL but gemm", dict(classes=[c for c in ALL if c != "gemm"], weights16=True, inputs16=True)),
             ("ALL but gn", dict(classes=[c for c in ALL if c != "gn"], weights16=True, inputs16=True))]
    for name, k in runs:
        if args.only and args.only not in name and not name.startswith("ALL ("):
            continue
        st = stats(oracle(**k), ref)
        report["classes"][name] = st
        print(f"  rounded: {name:55s} |d eps|max {st['max']:.3e}  p99.9 {st['p999']:.3e}  rms {st['rms']:.3e}", flush=True)
    if not args.no_hip and dev.type == "cuda":
        from musev_amd import ops
        from musev_amd.models.unet_loader import load_unet_by_name
        model = load_unet_by_name(case["flavour"], sd_unet_model=sd, dtype=torch.float16, **case["arch"]).to(dev)
        for carry, cs, fold in ((True, True, True), (False, True, True), (True, False, True), (True, True, False), (False, False, False)):
            ops.CARRY, ops.COLSTATS, ops.LN_FOLD = carry, cs, fold
            ops._ln_fold_cache.clear()
            taps = {}
            model._collect = taps
            with torch.no_grad():
                got = model(xd, torch.tensor(t0, device=dev), encoder_hidden_states=pd_, return_dict=False, **kwd)[0].float()
            model._collect = None
            torch.cuda.synchronize()
            entry = {"out": stats(got, ref), "taps": {}}
            for name, v in taps.items():
                if name in ref_taps:
                    r = ref_taps[name].float().cpu()
                    d = (v - r).abs()
                    # the carrier features (calibrate_as_denoiser: channels 0..7 of the level-0 stream) next to the rest
                    entry["taps"][name] = {"max": d.max().item(), "rms": d.pow(2).mean().sqrt().item(), "ref_absmax": r.abs().max().item(),
                                           "max_ch0_7": d[:, :8].max().item() if d.shape[1] >= 8 else None}
            report["hip"][f"carry={int(carry)} colstats={int(cs)} ln_fold={int(fold)}"] = entry
            print(f"  HIP carry={int(carry)} colstats={int(cs)} ln_fold={int(fold)}: |d eps|max {entry['out']['max']:.3e} p99.9 {entry['out']['p999']:.3e} rms {entry['out']['rms']:.3e}", flush=True)
            if cs and fold:
                print(f"    taps, carry={int(carry)}:")
                for name, e in entry["taps"].items():
                    print(f"      tap {name:22s} |d|max {e['max']:.3e} (channels 0-7: {e['max_ch0_7']:.3e})  rms {e['rms']:.3e}  |ref|max {e['ref_absmax']:.2f}", flush=True)
        ops.CARRY, ops.COLSTATS, ops.LN_FOLD = True, True, True
        ops._ln_fold_cache.clear()
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(report, f, indent=1)
    print(f"wrote {args.out}  ({time.time() - t_start:.0f} s)")


if __name__ == "__main__":
    main()
