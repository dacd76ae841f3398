#!/usr/bin/env python3
"""Offline study of the GEMM tile choice on UNSEEN problem sizes, from the committed tuner measurements (no GPU needed).

    python tools/tile_choice_study.py [--out profiles/<tag>_tile_choice_study.json]

Every tuner file (tools/gpu_gemm_tune.py -> profiles/*_gemm_tune.json) holds, per distinct problem of one forward, the measured
time of EVERY catalogue configuration (and the split factors tried), so any selection policy can be priced exactly on a
resolution it never saw: train on the files of one resolution, pick for the problems of another, add up `launches x ms[pick]`.
Policies:
  rules      the rule chooser of csrc/gemm.hip:choose_config (the tuner's "-2/0" row: measured, not modelled)
  best       the per-problem optimum of the test file itself (what a fresh tune would give)
  inherit    what the product does for a size that is not in its table: the entry of the same (mode, N, K, epilogue) with the
             NEAREST M within a factor of 3 lends its tile; its split factor only within a factor of 1.26, else the split rule
             (the product's only fallback until round 4)
  keyed      VERDICT r3 item 9's proposal: no exact shapes -- the key is (mode, epilogue, K-tile bucket, grid-fill bucket of the
             128 x 160 grid against the CU count); each key votes for the configuration with the least total time over the
             training problems that fall on it; unseen keys fall back to the rules
  hybrid     what csrc/gemm.hip:choose_config does since round 4: the nearest same-layer entry lends its tile only when it sits in
             the SAME key bucket as the problem (same K class, same grid fill: a 256 x 320 tile tuned on a full grid is not handed
             to a grid of 104 blocks), else the key's vote, else the nearest entry anyway, else the rules
Sizes: a 2.25x (512^2 -> 768^2) and a 0.44x change, both further than the reference README's 512 x 320 (0.625x) or an 8-frame
window (0.69x).  The keyed table of the product (gemm_tuned.h:kGemmKeyed, tools/gpu_gemm_tune.py --merge) is trained on ALL files."""
from __future__ import annotations

import argparse
import json
import math
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CUS = 256


def load(tag):
    with open(os.path.join(ROOT, "profiles", f"{tag}_gemm_tune.json")) as f:
        return json.load(f)


def block_dims(cfg_desc):
    return cfg_desc[0], cfg_desc[1]


def splitk_rule(blocks, nk):  # csrc/gemm.hip:splitk_rule
    s = (2 * CUS) // max(blocks, 1)
    s = min(s, nk // 8, 8)
    return max(s, 1)


def splitk_clamp(p, nsplit):  # csrc/gemm.hip:splitk_clamp
    nk = (p["K"] + 63) // 64
    if p["geglu"] or nsplit < 1:
        return 1
    while nsplit > 1 and (nsplit * p["M"] * p["N"] * 4 > (64 << 20) or nk // nsplit < 4):
        nsplit >>= 1
    return nsplit


def effective_split(p, cfg, configs, want):
    nk = (p["K"] + 63) // 64
    bm, bn = block_dims(configs[cfg])
    blocks = -(-p["M"] // bm) * -(-p["N"] // bn)
    ns = want
    if ns < 1:
        ns = splitk_rule(blocks, nk) if (blocks <= CUS and nk >= 16) else 1
    ns = splitk_clamp(p, ns)
    per = -(-nk // ns)
    ns = -(-nk // per)
    return 1 if p["ln"] else ns


def applies(p, cfg, configs):
    """csrc/gemm.hip:gemm_cfg_applies -- the GEGLU gate pairs 16-column tiles: even TN only (the 128 / 256-column tiles of the
    catalogue).  The tuner's forced run of any other tile on a GEGLU problem silently measured the rules."""
    return not p["geglu"] or configs[cfg][1] % 128 == 0


def price(p, cfg, ns):
    """measured time of (cfg, ns) on problem p; the tuner tried splits 1, 2, 4, 8 on a subset of tiles: an untried split is priced
    at the nearest tried one of that tile (None if the tile itself was not measured, e.g. an odd TN under GEGLU)"""
    ms = p["ms"]
    if f"{cfg}/{ns}" in ms:
        return ms[f"{cfg}/{ns}"]
    tried = sorted((abs(math.log2(int(k.split("/")[1])) - math.log2(max(ns, 1))), v) for k, v in ms.items() if k.split("/")[0] == str(cfg))
    return tried[0][1] if tried else None


def same_layer(a, b):
    return a["mode"] == b["mode"] and a["N"] == b["N"] and a["K"] == b["K"] and a["geglu"] == b["geglu"]


def nearest_entry(p, train):
    return _nearest(p, train)[0]


def _nearest(p, train):
    best, best_d = None, 1e30
    for e in train:
        if not same_layer(e, p) or (e["ln"] and not p["ln"]):
            continue
        r = p["M"] / e["M"]
        d = r if r > 1 else 1 / r
        if d > 3.0:
            continue
        if bool(e["ln"]) != bool(p["ln"]):
            d *= 1.0001
        if d < best_d:
            best, best_d = e, d
    return best, best_d


def pick_inherit(p, train, configs):
    best, best_d = _nearest(p, train)
    if best is None or best["picked"] is None or best["picked"][0] < 0 or not applies(p, best["picked"][0], configs):
        return None  # (no entry / the rules measured best there / a tile the epilogue cannot run: the lookup skips it)
    cfg, ns = best["picked"]
    return cfg, effective_split(p, cfg, configs, ns if best_d <= 1.26 else 0)


K_EDGES = (5, 10, 20, 60)          # K tiles of 64: level-0 projections | 640-wide | 1280-wide | conv K | long conv K
FILL_EDGES_X2 = (1, 2, 4, 8, 32)   # 2 x (128 x 160 grid / CUs): half a round | one | two | four | sixteen | more


def key_of(p):
    """csrc/gemm.hip:gemm_key -- (mode, geglu, ln, K bucket, grid-fill bucket); integer arithmetic as there"""
    nk = (p["K"] + 63) // 64
    kb = sum(nk > e for e in K_EDGES)
    grid = -(-p["M"] // 128) * -(-p["N"] // 160)
    fb = sum(2 * grid > e * CUS for e in FILL_EDGES_X2)
    return (p["mode"], p["geglu"], p["ln"], kb, fb)


TWO_BLOCK_CFGS = (0, 1, 2, 3, 13, 14, 15, 16)   # as tools/gpu_gemm_tune.py: two LDS stages, <= 80 KB, two blocks per CU


def train_keyed(train, configs, prefer_two_block=False):
    votes = {}
    for e in train:
        k = key_of(e)
        for c in range(len(configs)):
            t = price(e, c, effective_split(e, c, configs, 0)) if applies(e, c, configs) else None
            if t is None:
                continue
            votes.setdefault(k, {}).setdefault(c, [0.0, 0])
            votes[k][c][0] += t * e["launches"]
            votes[k][c][1] += 1
    out = {}
    for k, per in votes.items():
        n = max(v[1] for v in per.values())
        full = {c: v[0] for c, v in per.items() if v[1] == n}  # tiles measured on every problem of the key
        best = min(full, key=full.get)
        if prefer_two_block:  # the table's objective (two streams share the CUs): the fastest two-block tile within 3 % of the fastest
            two = {c: v for c, v in full.items() if c in TWO_BLOCK_CFGS and v <= 1.03 * full[best]}
            if two:
                best = min(two, key=two.get)
        out[k] = best
    return out


def evaluate(test, train, configs):
    keyed = train_keyed(train, configs)
    tot = {"rules": 0.0, "best": 0.0, "inherit": 0.0, "keyed": 0.0, "hybrid": 0.0}
    miss = {"inherit": 0, "keyed": 0, "hybrid": 0}
    worst = []
    for p in test["problems"]:
        n = p["launches"]
        rules = p["ms"]["-2/0"]
        tot["rules"] += n * rules
        tot["best"] += n * p["best_ms"]
        ch = pick_inherit(p, train, configs)
        t = price(p, *ch) if ch else None
        if t is None:
            miss["inherit"] += 1
            t = rules
        tot["inherit"] += n * t
        t_inherit = None if ch is None else t
        worst.append((n * (t - p["best_ms"]), f'{p["mode"]} M{p["M"]} N{p["N"]} K{p["K"]} g{p["geglu"]} ln{p["ln"]}', ch, p["picked"] or "rules", round(t / p["best_ms"], 3)))
        k = key_of(p)
        t = None
        if k in keyed:
            c = keyed[k]
            t = price(p, c, effective_split(p, c, configs, 0))
        if t is None:
            miss["keyed"] += 1
            t = rules
        tot["keyed"] += n * t
        # hybrid (the product): same-bucket neighbour -> key vote -> any neighbour within 3x -> rules
        e = nearest_entry(p, train)
        th = t_inherit if (e is not None and key_of(e) == k) else None
        if th is None and k in keyed:
            th = price(p, keyed[k], effective_split(p, keyed[k], configs, 0))
        if th is None:
            th = t_inherit
        if th is None:
            miss["hybrid"] += 1
            th = rules
        tot["hybrid"] += n * th
    worst.sort(reverse=True)
    return tot, miss, worst[:5]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    files = {t: load(t) for t in ("r03g_musev512", "r03g_refnet512", "r03g_refnet768", "r04t_musev512")}
    configs = files["r03g_musev512"]["configs"]
    assert all(f["configs"] == configs for f in files.values())
    legs = [("r03g_refnet768", ["r03g_musev512", "r03g_refnet512"], "512^2 tables -> 768^2 problems (M x 2.25)"),
            ("r03g_refnet512", ["r03g_refnet768"], "768^2 table -> 512^2 problems (M x 0.44)"),
            ("r03g_musev512", ["r03g_refnet768"], "768^2 refnet table -> 512^2 musev problems (M x 0.44, other flavour)"),
            ("r04t_musev512", ["r03g_refnet768"], "768^2 table of the round-3 build -> 512^2 problems on the round-4 build")]
    report = []
    for test_tag, train_tags, what in legs:
        train = [p for t in train_tags for p in files[t]["problems"]]
        tot, miss, worst = evaluate(files[test_tag], train, configs)
        row = {"test": test_tag, "train": train_tags, "what": what, "ms_per_forward": {k: round(v, 3) for k, v in tot.items()},
               "over_best": {k: round(v / tot["best"], 4) for k, v in tot.items()}, "fell_back_to_rules": miss,
               "largest_inherit_losses": [dict(ms_lost=round(w[0], 4), problem=w[1], inherited=w[2], best=w[3], ratio=w[4]) for w in worst]}
        report.append(row)
        print(f"{what}\n   " + "  ".join(f"{k} {tot[k]:.2f} ms ({tot[k] / tot['best']:.3f}x)" for k in ("best", "hybrid", "inherit", "keyed", "rules")) +
              f"   [rules fallbacks: hybrid {miss['hybrid']}, inherit {miss['inherit']}, keyed {miss['keyed']} of {len(files[test_tag]['problems'])}]")
        for w in worst[:3]:
            print(f"      -{w[0]:.3f} ms  {w[1]}  inherited {w[2]} best {w[3]}  x{w[4]}")
    if args.out:
        with open(args.out, "w") as f:
            json.dump({"tool": "tools/tile_choice_study.py", "cus": CUS, "legs": report}, f, indent=1)


if __name__ == "__main__":
    main()
