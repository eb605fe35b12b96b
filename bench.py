#!/usr/bin/env python
"""bench.py -- cost-volumes/sec (SGA + LGA2 forward+backward) at 240x624x192 on MI355X.

One STEP = one cost volume as GA-Net feeds the ops at crop 240x624, max_disp 192
(SURVEY.md 8d):
    SgaFunction  fwd+bwd on x [1,32,65,80,208] with four guidance tensors [1,32,5,80,208]
  + Lga2Function fwd+bwd (radius 2, two chained passes) on x [1,193,240,624], f [1,75,240,624]
fp32, synthetic inputs resident in HBM before the timed region, torch.manual_seed(123).

    python bench.py [--gpus N] [--steps K] [--warmup W]
N > 1 is launched by the driver as torch.distributed.run, one rank per GPU (without a launcher this file re-executes
itself that way); ranks run independent cost volumes (the ops have no cross-sample term, so there is no data-path
collective; the barrier / max-over-ranks timing protocol runs over gloo on the host): weak scaling,
value = N*K / max-over-ranks time.  After the timed region the ranks run the one collective a data-parallel caller of this
path has -- the 26.3 MB fp32 gradient all-reduce of GANet-deep -- over RCCL and report it as `rccl` (never part of
`value`).  Rank 0 prints ONE JSON line.

Extra objects in the line:
  roofline      ONE kernel under the name rocprofv3 prints for it -- the launch with the largest share of the step -- against its
                binding bound (HBM bandwidth for the SGA kernels, fp32 FMA rate for the LGA kernels), from HIP-event
                timings of every kernel in place on the launch stream after the timed region; `kernels` lists all 16 launches,
                `families` the six kernel families with both fractions, `dominant_family` / `worst_family` name the family with
                the largest share and the one furthest below its bound; the measured traffic (PMC) beside the algorithmic bytes
  value         the contract's protocol: W warm-up replays + K timed ones straight after the graph capture; `value_settled` = the
                same K steps after 0.1 s more of replays (the clock ramp after the capture is worth ~1.5 %), `value_1s` sustained
  cpu_baseline  the CPU checker (oracle/_ref = the reference's own kernel bodies when the prebuilt
                .so is present, else the C restatement) timed on this box's host cores on ONE
                cost volume of the same workload
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from ganet_amd import dist as gdist  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
SGA_SHAPE = (1, 32, 65, 80, 208)
LGA_SHAPE = (1, 193, 240, 624)
RADIUS = 2
SETTLE_S = 0.1                 # extra replays before the SECOND measurement (`value_settled`): see main()

# algorithmic bytes (SURVEY.md 8d / BASELINE.md 2): inputs read once + outputs written once
_V = 4 * 1 * 32 * 65 * 80 * 208
_G = 4 * 1 * 32 * 5 * 80 * 208
_VL = 4 * 1 * 193 * 240 * 624
_F = 4 * 1 * 75 * 240 * 624
ALG_BYTES = {
    "sga_fwd": 2 * _V + 4 * _G,
    "sga_bwd": 3 * _V + 8 * _G,
    "lga2_fwd": 3 * _VL + _F,
    "lga2_bwd": 4 * _VL + 2 * _F,
}
UNIT_BYTES = sum(ALG_BYTES.values())     # 1,764,106,240 B per cost volume


def make_inputs(device):
    torch.manual_seed(123)
    x = torch.randn(SGA_SHAPE, device=device, requires_grad=True)
    gshape = SGA_SHAPE[:2] + (5,) + SGA_SHAPE[3:]
    gs = [F.normalize(torch.randn(gshape, device=device), p=1, dim=2).requires_grad_() for _ in range(4)]
    go = torch.randn(SGA_SHAPE, device=device)
    xl = torch.randn(LGA_SHAPE, device=device, requires_grad=True)
    f = F.normalize(torch.randn((1, 75) + LGA_SHAPE[2:], device=device), p=1, dim=1).requires_grad_()
    gy = torch.randn(LGA_SHAPE, device=device)
    return x, gs, go, xl, f, gy


def one_step(inp):
    from ganet_amd.functions.GANet import Lga2Function, SgaFunction
    x, gs, go, xl, f, gy = inp
    out = SgaFunction.apply(x, *gs)
    g1 = torch.autograd.grad(out, [x] + gs, go)
    y = Lga2Function.apply(xl, f, RADIUS)
    g2 = torch.autograd.grad(y, [xl, f], gy)
    return g1, g2


def one_step_two_streams(inp, side):
    """Same work as one_step, the SGA half on the current stream and the LGA half on `side`: the two halves of a
    step have no data dependency in this workload (reported separately, never as `value`)."""
    from ganet_amd.functions.GANet import Lga2Function, SgaFunction
    x, gs, go, xl, f, gy = inp
    cur = torch.cuda.current_stream()
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        y = Lga2Function.apply(xl, f, RADIUS)
        g2 = torch.autograd.grad(y, [xl, f], gy)
    out = SgaFunction.apply(x, *gs)
    g1 = torch.autograd.grad(out, [x] + gs, go)
    cur.wait_stream(side)
    return g1, g2


def stage_timings(inp, iters=5, only=None):
    """Device time (ms) of EVERY kernel of a step, measured in place: the step's 16 launches issued through the C ABI in the
    order the autograd Functions issue them, a HIP event between consecutive launches on the launch stream, `iters` passes.
    (A kernel repeated on its own sees its inputs cached from its previous run -- with the non-temporal result stores a scan
    repeated alone runs 30 % faster than inside the step, an LGA pass ~12 % -- and a "whole call minus its scans" difference
    adds the biases of both sides; the interval between two events holds one kernel and one launch gap, which is what the
    kernel trace of the same command shows as well: profiles/*kernel_stats_in_step.csv.)"""
    from ganet_amd import _native
    lib = _native.lib()
    x, gs, go, xl, f, gy = [t.detach() if torch.is_tensor(t) else [u.detach() for u in t] for t in inp]
    N, C, D, H, W = x.shape
    st = torch.cuda.current_stream().cuda_stream
    A = torch.empty((4,) + tuple(x.shape), device=x.device)
    out = torch.empty_like(x)
    mask = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
    kp = torch.empty((4, N, C, H, W), dtype=torch.int16, device=x.device)
    G = torch.empty_like(A)
    gx = torch.empty_like(x)
    gw = [torch.empty_like(g) for g in gs]
    B, DL, HL, WL = xl.shape
    y, gxl = torch.empty_like(xl), torch.empty_like(xl)
    gf = torch.empty_like(f)
    # LGA2 as Lga2Function runs it: the private intermediate and its gradient pair-interleaved (functions/GANet.py: _LgaChain)
    tp = torch.empty(B * ((DL + 1) // 2) * HL * WL * 2, device=xl.device)
    gtp = torch.empty_like(tp)
    edge = torch.empty((B, 3, HL, WL), device=xl.device)      # the filters' edge sums: written by the first pass, read by the data-backward launches
    p = lambda t: t.data_ptr()      # noqa: E731
    names = ["down", "up", "right", "left"]
    gp = [p(g) for g in gs]
    calls = []
    for d in range(4):
        calls.append((f"sga_scan_fwd_{names[d]}", lambda d=d: lib.call("ganet_sga_scan_forward_ws", p(x), gp[d], p(A), N, C, D, H, W, d, st)))
    calls.append(("sga_merge_argmax", lambda: lib.call("ganet_sga_merge", p(A), p(out), p(mask), p(kp), N, C, D, H, W, st)))
    for d in range(4):
        calls.append((f"sga_bwd_scan_{names[d]}", lambda d=d: lib.call("ganet_sga_backward_scan_ws", gp[d], p(mask), p(kp),
                                                                      p(go), p(G), N, C, D, H, W, d, st)))
    calls.append(("sga_bwd_point", lambda: lib.call("ganet_sga_backward_point", p(x), *gp, p(A), p(G), p(gx), *[p(t) for t in gw],
                                                    N, C, D, H, W, st)))
    calls += [
        ("lga_fwd_apply_1", lambda: lib.call("ganet_lga_apply_paired_edges", p(xl), p(f), p(tp), p(edge), B, DL, HL, WL, RADIUS, 0, 0, 1, st)),
        ("lga_fwd_apply_2", lambda: lib.call("ganet_lga_apply_paired", p(tp), p(f), p(y), B, DL, HL, WL, RADIUS, 0, 1, 0, st)),
        ("lga_bwd_filter_grad_2", lambda: lib.call("ganet_lga_filter_grad_paired", p(tp), p(gy), p(gf), B, DL, HL, WL, RADIUS, 0, 1, 0, st)),
        ("lga_bwd_data_2", lambda: lib.call("ganet_lga_apply_paired_edges", p(gy), p(f), p(gtp), p(edge), B, DL, HL, WL, RADIUS, 1, 0, 1, st)),
        ("lga_bwd_filter_grad_1", lambda: lib.call("ganet_lga_filter_grad_paired", p(xl), p(gtp), p(gf), B, DL, HL, WL, RADIUS, 1, 0, 1, st)),
        ("lga_bwd_data_1", lambda: lib.call("ganet_lga_apply_paired_edges", p(gtp), p(f), p(gxl), p(edge), B, DL, HL, WL, RADIUS, 1, 1, 0, st)),
    ]
    if only is not None:          # (development A/B scripts: one op's kernels, e.g. with a library build that lacks the newer entries)
        calls = [c for c in calls if c[0].startswith(only)]
    for _ in range(2):
        for _, fn in calls:
            fn()
    torch.cuda.synchronize()
    acc = [0.0] * len(calls)
    for _ in range(iters):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(len(calls) + 1)]
        ev[0].record()
        for i, (_, fn) in enumerate(calls):
            fn()
            ev[i + 1].record()
        ev[-1].synchronize()
        for i in range(len(calls)):
            acc[i] += ev[i].elapsed_time(ev[i + 1])
    res = {name: a / iters for (name, _), a in zip(calls, acc)}
    res["step_sum_of_kernels"] = sum(res.values())
    if only is not None and only != "lga":
        return res
    # per PASS, as earlier rounds reported the LGA2 chain (mean of its two passes)
    res["lga_fwd_pass"] = (res["lga_fwd_apply_1"] + res["lga_fwd_apply_2"]) / 2
    res["lga_bwd_pass"] = sum(v for k, v in res.items() if k.startswith(("lga_bwd_filter_grad", "lga_bwd_data"))) / 2
    return res


# which kernels make up one launch of each family (names as in profiles/traffic_pmc.json)
_FAMILY_KERNELS = {
    "sga_scan_fwd": [["sga_col_fwd<5, true, true>"], ["sga_col_fwd<5, false, true>"],
                     ["sga_row_fwd<2, 32, 4, 1, false, false, 64, 9>"], ["sga_row_fwd<2, 32, 4, 1, true, false, 64, 9>"]],
    "sga_merge_argmax": [["sga_merge_px4"]],
    "sga_bwd_scan": [["sga_col_bwdg<5, false, true, true>"], ["sga_col_bwdg<5, true, true, true>"],
                     ["sga_row_bwdg<2, 32, 4, 1, true, false, 64, 9>"], ["sga_row_bwdg<2, 32, 4, 1, false, false, 64, 9>"]],
    "sga_bwd_point": [["sga_bwd_point<4, false, true>|sga_bwd_point<4, false>|sga_bwd_point<4, false, false>"]],   # (a|b: first name found)
    # (workgroup-ring kernels = the default, GANET_LGA_WAVE=2; the one-wave names as alternatives: whichever the traffic file holds)
    "lga_apply+filter_grad (bwd pass)": [["lga_filter_grad_pair_xp<5>|lga_filter_grad_pp_xp<2, 3, 0>",
                                          "lga_apply_pp_wxo<2, true, false>|lga_apply_pp_xo<2, true, false>"],
                                         ["lga_filter_grad_pp_gypx<2, 3, 0>",
                                          "lga_apply_pp_wpi<2, true, false>|lga_apply_pp_pi<2, true, false>"]],
    "lga_apply (fwd pass)": [["lga_apply_pp_wxo<2, false, false>|lga_apply_pp_xo<2, false, false>"],
                             ["lga_apply_pp_wpi<2, false, false>|lga_apply_pp_pi<2, false, false>"]],
}


FP32_PEAK_TFLOPS = 157.3        # MI355X_MICROARCH.md: fp32 vector = fp32 matrix peak
# what this chip was MEASURED to reach with the access patterns / instruction streams these kernels can use (micro-benchmarks
# under scripts/ubench, results in profiles/): the spec peaks above stay the denominators of every `frac`
ACHIEVABLE = {"hbm_copy_GBs": 6300.0, "hbm_4in_1out_marching_GBs": 5000.0, "hbm_linear_fill_GBs": 4400.0,
              "fp32_pk_fma_TFLOPs": 134.0,
              "source": "profiles/r3b_mall_bw.txt, scripts/ubench/stream_patterns.hip, profiles/r2a_ubench_valu_rate.txt"}
# north_star: "(for LGA) MFMA utilisation reported against gfx950 peak".  The LGA contraction is [D x 25] . [25 x 3] per PIXEL with both
# operands private to the pixel: an MFMA tile would carry N = 3 useful columns of 16 / 32, and the one fp32 shape that fits,
# v_mfma_f32_4x4x1_16B_f32, was measured at 77 - 90 TFLOP/s alone against 134 for v_pk_fma_f32 (mixing the two is slower than
# either): the kernels use the packed VALU and no MFMA instruction, so MFMA utilisation is 0 by construction.
MFMA_NOTE = {"used": False, "mfma_utilisation": 0.0, "fp32_mfma_peak_TFLOPs": FP32_PEAK_TFLOPS,
             "f32_4x4x1_measured_TFLOPs": [77.0, 90.0], "pk_fma_measured_TFLOPs": 134.0,
             "why": "per-pixel [D x 25].[25 x 3]: N = 3 of an MFMA tile's 16/32 columns; packed fp32 VALU is the faster unit.  Only "
                    "v_mfma_f32_4x4x1_16B_f32 (16 independent 4x4 blocks: N = 3 of 4) was timed; the 16x16x4 / 32x32x2 f32 shapes reach "
                    "155 TFLOP/s per MI355X_MICROARCH.md but at N = 3 of 16 / 32 columns deliver <= 19 % / <= 9 % of that as useful "
                    "flops (<= 29 TFLOP/s against the 47 the packed VALU kernels run at), so they were not built",
             "source": "profiles/r2a_ubench_valu_rate.txt, DESIGN.md section 3.3"}
_LGA_PASS_FLOPS = 2.0 * 75 * 193 * 240 * 624      # 75 FMAs per output element, one apply or one filter-gradient pass
_FAMILY_FLOPS = {"lga_apply (fwd pass)": _LGA_PASS_FLOPS, "lga_apply+filter_grad (bwd pass)": 2 * _LGA_PASS_FLOPS}


# Every launch of the step under the name rocprofv3 prints for it (template instantiation, namespace and argument list dropped;
# a|b: whichever the traffic file holds -- the workgroup-ring kernels are the default, the one-wave names the fallback), with the
# API-level bytes it is charged with (the family rows' split, see roofline_from_stages; an LGA backward pass's 2 V_L + F goes half to
# its filter gradient and half to its data-backward) and, for the LGA kernels, its flops (their binding bound).
_STAGE_KERNELS = {
    "sga_scan_fwd_down": "sga_col_fwd<5, true, true>", "sga_scan_fwd_up": "sga_col_fwd<5, false, true>",
    "sga_scan_fwd_right": "sga_row_fwd<2, 32, 4, 1, false, false, 64, 9>", "sga_scan_fwd_left": "sga_row_fwd<2, 32, 4, 1, true, false, 64, 9>",
    "sga_merge_argmax": "sga_merge_px4",
    "sga_bwd_scan_down": "sga_col_bwdg<5, false, true, true>", "sga_bwd_scan_up": "sga_col_bwdg<5, true, true, true>",
    "sga_bwd_scan_right": "sga_row_bwdg<2, 32, 4, 1, true, false, 64, 9>", "sga_bwd_scan_left": "sga_row_bwdg<2, 32, 4, 1, false, false, 64, 9>",
    "sga_bwd_point": "sga_bwd_point<4, false, true>|sga_bwd_point<4, false>|sga_bwd_point<4, false, false>",
    "lga_fwd_apply_1": "lga_apply_pp_wxo<2, false, false>|lga_apply_pp_xo<2, false, false>",
    "lga_fwd_apply_2": "lga_apply_pp_wpi<2, false, false>|lga_apply_pp_pi<2, false, false>",
    "lga_bwd_filter_grad_2": "lga_filter_grad_pair_xp<5>|lga_filter_grad_pp_xp<2, 3, 0>",
    "lga_bwd_data_2": "lga_apply_pp_wxo<2, true, false>|lga_apply_pp_xo<2, true, false>",
    "lga_bwd_filter_grad_1": "lga_filter_grad_pp_gypx<2, 3, 0>",
    "lga_bwd_data_1": "lga_apply_pp_wpi<2, true, false>|lga_apply_pp_pi<2, true, false>",
}


def _stage_alg_bytes(stage):
    if stage.startswith("sga_scan_fwd_"):
        return _V / 4 + _G
    if stage == "sga_merge_argmax":
        return _V
    if stage.startswith("sga_bwd_scan_"):
        return _V / 4
    if stage == "sga_bwd_point":
        return 2 * _V + 8 * _G
    if stage.startswith("lga_fwd_apply"):
        return ALG_BYTES["lga2_fwd"] / 2
    return ALG_BYTES["lga2_bwd"] / 4


def kernel_table(stages, kern):
    """One row per launch of the step: rocprof kernel name, its measured time in place, share of the step, binding bound and
    fraction of that bound's peak (HBM for the SGA kernels, fp32 for the LGA kernels), measured traffic where the PMC file has it."""
    total = sum(stages[k] for k in _STAGE_KERNELS if k in stages)
    rows = []
    for stage, spec in _STAGE_KERNELS.items():
        if stage not in stages:
            continue
        ms = stages[stage]
        name = next((n for n in spec.split("|") if kern and n in kern), spec.split("|")[0])
        b = _stage_alg_bytes(stage)
        hbm = b / (ms * 1e-3) / 1e9
        row = {"stage": stage, "kernel": name, "ms": round(ms, 4), "share": round(ms / total, 4), "bound": "hbm",
               "alg_bytes": int(b), "hbm_GBs": round(hbm, 1), "hbm_frac": round(hbm / HBM_PEAK_GBS, 4), "frac": round(hbm / HBM_PEAK_GBS, 4),
               "traffic": (kern[name]["read_bytes"] + kern[name]["write_bytes"]) if kern and name in kern else None}
        if stage.startswith("lga_"):
            tf = _LGA_PASS_FLOPS / (ms * 1e-3) / 1e12
            row.update({"bound": "fp32", "fp32_tflops": round(tf, 1), "frac": round(tf / FP32_PEAK_TFLOPS, 4)})
        rows.append(row)
    return rows


def csrc_tree_hash():
    """sha256[:12] over the kernel sources (ganet_amd/csrc/*.hip|*.h|*.inc, names and contents, sorted): identifies the tree a
    traffic file was measured on (scripts/pmc_traffic.py stores it; bench warns when the file describes other kernels)."""
    import hashlib
    d = os.path.join(ROOT, "ganet_amd", "csrc")
    h = hashlib.sha256()
    for fn in sorted(os.listdir(d)):
        if fn.endswith((".hip", ".h", ".inc")):
            h.update(fn.encode())
            with open(os.path.join(d, fn), "rb") as f:
                h.update(f.read())
    return h.hexdigest()[:12]


def pmc_traffic_tree():
    """the csrc hash profiles/traffic_pmc.json was generated on (None: an older file / no file)"""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic_pmc.json")) as f:
            return json.load(f).get("csrc_tree")
    except (OSError, ValueError):
        return None


def pmc_traffic():
    """Per-kernel traffic (bytes per dispatch at the L2 <-> fabric boundary) from the committed rocprofv3 --pmc
    passes (profiles/traffic_pmc.json, made by `scripts/gpu_session.sh <tag> pmc` + scripts/pmc_traffic.py on an MI355X with
    this workload).  Counters cannot be collected inside the timed run, hence the file; None when it is absent."""
    fn = os.path.join(ROOT, "profiles", "traffic_pmc.json")
    try:
        with open(fn) as f:
            return json.load(f)["kernels"]
    except (OSError, ValueError, KeyError):
        return None


def _family_traffic(name, kern):
    """Mean traffic of one launch of the family (a launch = the kernels of one inner list)."""
    if kern is None:
        return None
    tot, n = 0, 0
    for launch in _FAMILY_KERNELS[name]:
        try:
            for spec in launch:
                k = next(n for n in spec.split("|") if n in kern)
                tot += kern[k]["read_bytes"] + kern[k]["write_bytes"]
        except StopIteration:
            return None
        n += 1
    return int(tot / n)


def roofline_from_stages(stages):
    """HBM roofline per kernel family and for the dominant one (largest time share of a step).
    `achieved` = the family's algorithmic bytes per launch / its average launch duration.  The op-level
    figures of SURVEY 8d are split by which kernel touches the API-level tensor: SGA forward 2V+4G =
    4 scans x (V/4 + G) [x, guidance] + merge V [out]; SGA backward 3V+8G = 4 adjoint scans x V/4 [gradOut] +
    per-pixel kernel 2V+8G [x, guidance in; gradInput, guidance grads out]; an LGA pass is half of its op.
    `traffic` = measured bytes per launch from the PMC passes (pmc_traffic())."""
    fam = {
        "sga_scan_fwd": ([k for k in stages if k.startswith("sga_scan_fwd_")], _V / 4 + _G, 1),
        "sga_merge_argmax": (["sga_merge_argmax"], _V, 1),
        "sga_bwd_scan": ([k for k in stages if k.startswith("sga_bwd_scan_")], _V / 4, 1),
        "sga_bwd_point": (["sga_bwd_point"], 2 * _V + 8 * _G, 1),
        # an LGA "launch" of a family = one PASS: forward one apply kernel, backward filter gradient + data-backward
        "lga_apply+filter_grad (bwd pass)": (["lga_bwd_pass"], ALG_BYTES["lga2_bwd"] / 2, 2),
        "lga_apply (fwd pass)": (["lga_fwd_pass"], ALG_BYTES["lga2_fwd"] / 2, 2),
    }
    kern = pmc_traffic()
    table, best, best_t = [], None, -1.0
    unit_traffic = 0
    for name, (keys, bytes_per_launch, mult) in fam.items():
        avg_ms = sum(stages[k] for k in keys) / len(keys)
        step_ms = sum(stages[k] for k in keys) * mult
        achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9
        traffic = _family_traffic(name, kern)
        row = {"kernel": name, "launches_per_step": len(keys) * mult, "avg_launch_ms": round(avg_ms, 4),
               "step_ms": round(step_ms, 4), "bound": "hbm", "frac": round(achieved / HBM_PEAK_GBS, 4),
               "alg_bytes_per_launch": int(bytes_per_launch), "hbm_GBs": round(achieved, 1),
               "hbm_frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic}
        if traffic is not None:
            # what the launch actually moves per second at the L2 <-> fabric boundary (measured bytes / measured time): a family
            # far below its algorithmic roofline may still sit near the fabric's rate because of what it re-reads
            row["traffic_GBs"] = round(traffic / (avg_ms * 1e-3) / 1e9, 1)
        if traffic is not None and unit_traffic is not None:
            unit_traffic += traffic * len(keys) * mult
        else:
            unit_traffic = None
        if name in _FAMILY_FLOPS:
            # the LGA families' binding bound is arithmetic (fp32 VALU = fp32 MFMA rate, 157.3 TFLOP/s; SURVEY 8d: 27.6 flop/B
            # lies above the ridge of 19.7): `frac` is the fp32 fraction, the HBM fraction stays alongside
            tf = _FAMILY_FLOPS[name] / (avg_ms * 1e-3) / 1e12
            row["bound"] = "fp32"
            row["fp32_tflops"] = round(tf, 1)
            row["fp32_frac"] = round(tf / FP32_PEAK_TFLOPS, 4)
            row["frac"] = row["fp32_frac"]
        table.append(row)
        if step_ms > best_t:
            best, best_t = row, step_ms
    # The headline object is ONE kernel under its rocprof name: the launch with the largest share of the step, against its own
    # binding bound (a family of two kernels summed must not be the headline).  The family table stays, and
    # `dominant_family` / `worst_family` name the family with the largest share and the one furthest below its bound.
    kernels = kernel_table(stages, kern)
    top = max(kernels, key=lambda r: r["ms"])
    fp32 = top["bound"] == "fp32"
    worst = min(table, key=lambda r: r["frac"])
    out = {"bound": "fp32 (VALU)" if fp32 else "hbm", "kernel": top["kernel"], "stage": top["stage"], "share_of_step": top["share"],
           "achieved": top["fp32_tflops"] if fp32 else top["hbm_GBs"], "peak": FP32_PEAK_TFLOPS if fp32 else HBM_PEAK_GBS,
           "unit": "TFLOP/s" if fp32 else "GB/s", "frac": top["frac"],
           "alg_flops_per_launch": int(_LGA_PASS_FLOPS) if fp32 else None,
           "alg_bytes_per_launch": top["alg_bytes"], "hbm_GBs": top["hbm_GBs"], "hbm_frac": top["hbm_frac"],
           "avg_launch_ms": top["ms"], "traffic": top["traffic"],
           "traffic_source": "profiles/traffic_pmc.json (rocprofv3 --pmc TCC_EA0 request counters x request size, bytes per launch "
                             "at the L2 <-> fabric boundary: Infinity-Cache hits included, so an upper bound of DRAM traffic)" if top["traffic"] else None,
           "largest_kernel": top["kernel"],
           "dominant_family": {"kernel": best["kernel"], "step_ms": best["step_ms"], "bound": best["bound"], "frac": best["frac"]},
           "worst_family": {"kernel": worst["kernel"], "bound": worst["bound"], "frac": worst["frac"]},
           "kernels": kernels, "families": table}
    if unit_traffic is not None:
        out["unit_traffic_bytes"] = int(unit_traffic)
        out["unit_traffic_ratio"] = round(unit_traffic / UNIT_BYTES, 3)
    out["timing"] = ("every kernel of the step timed in place (HIP events between consecutive launches of one in-order pass over "
                     "the step's 16 launches): stage_ms; no family is a difference of other measurements")
    out["mfma"] = MFMA_NOTE
    out["achievable"] = ACHIEVABLE
    # which kernels the traffic figures describe: the file is made in a profiler session of its own (counters cannot be collected
    # inside the timed run), so it names the tree it was measured on and a mismatch is said out loud
    out["traffic_tree"] = pmc_traffic_tree()
    out["csrc_tree"] = csrc_tree_hash()
    out["traffic_tree_matches"] = out["traffic_tree"] == out["csrc_tree"]
    if kern is not None and not out["traffic_tree_matches"]:
        print(f"[bench] WARNING: profiles/traffic_pmc.json was measured on csrc tree {out['traffic_tree']}, this tree is "
              f"{out['csrc_tree']}: roofline.traffic / unit_traffic_ratio may describe other kernels "
              "(regenerate: scripts/gpu_session.sh <tag> pmc)", file=sys.stderr, flush=True)
    return out


def cpu_baseline():
    """One cost volume (SGA fwd+bwd + LGA2 fwd+bwd) on the host cores through the CPU checker."""
    import numpy as np
    from oracle import oracle
    kind = "reference" if oracle.have("reference") else "port"
    ora = oracle.Oracle(kind)
    rng = np.random.default_rng(123)
    x = rng.standard_normal(SGA_SHAPE).astype(np.float32)
    gshape = SGA_SHAPE[:2] + (5,) + SGA_SHAPE[3:]
    gs = []
    for _ in range(4):
        g = rng.standard_normal(gshape).astype(np.float32)
        gs.append((g / np.abs(g).sum(2, keepdims=True)).astype(np.float32))
    go = rng.standard_normal(SGA_SHAPE).astype(np.float32)
    xl = rng.standard_normal(LGA_SHAPE).astype(np.float32)
    f = rng.standard_normal((1, 75) + LGA_SHAPE[2:]).astype(np.float32)
    f = (f / np.abs(f).sum(1, keepdims=True)).astype(np.float32)
    gy = rng.standard_normal(LGA_SHAPE).astype(np.float32)
    t0 = time.time()
    out, tmp, mask = ora.sga_forward(x, *gs)
    ora.sga_backward(x, *gs, tmp, mask, go)
    y, ins = ora.lga_chain_forward(xl, f, RADIUS, 2)
    ora.lga_chain_backward(ins, f, gy, RADIUS)
    dt = time.time() - t0
    return {"value": round(1.0 / dt, 4), "unit": "cost-volumes/sec", "cores": int(ora.threads), "kind": kind,
            "sample": "1 cost volume (SGA fwd+bwd + LGA2 fwd+bwd, same shapes), %.1f s wall" % dt}


def _free_port():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: re-execute this command line under torch.distributed.run,
    one rank per GPU, rendezvous on 127.0.0.1 (what the driver's own launch line does)."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC for the RCCL probe after the timed region (task notes: the host driver supports no legacy IPC)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def stub_main(args):
    """--stub-step: the launch / barrier / max-over-ranks / one-JSON-line protocol on CPU (gloo) with a placeholder
    step -- exercises the N > 1 plumbing where there is no GPU (tests/test_dist_gloo.py).  Not a measurement."""
    sys.stdout.flush()
    saved_fd = os.dup(1)
    os.dup2(2, 1)                       # gloo announces its connections on stdout: keep stdout for the one JSON line
    try:
        ctx = gdist.init(args.gpus, backend="gloo")
        gdist.barrier(ctx)
    finally:
        sys.stdout.flush()
        os.dup2(saved_fd, 1)
        os.close(saved_fd)
    a = torch.randn(64, 64)

    def step():
        return (a @ a).sum().item()

    for _ in range(args.warmup):
        step()
    elapsed = gdist.timed_region(ctx, lambda: [step() for _ in range(args.steps)])
    line = {"metric": "stub (protocol test only)", "value": round(ctx.world_size * args.steps / elapsed, 2),
            "unit": "steps/sec", "n_gpus": ctx.world_size, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "stub"}
    if ctx.rank == 0:
        print("[bench] measured: " + json.dumps(line), file=sys.stderr, flush=True)
    # the protocol test's stand-in for the RCCL probe of the real run: same code path (child processes, own rendezvous), gloo
    delay = float(os.environ.get("GANET_BENCH_TEST_DELAY_RANK0", "0"))
    if delay and ctx.rank == 0:
        time.sleep(delay)
    probe = gdist.allreduce_probe_isolated(ctx, "gloo", 0, timeout_s=float(os.environ.get("GANET_BENCH_PROBE_TIMEOUT", "150")),
                                           nelem=1 << 16, iters=2)
    if probe is not None:
        line["rccl"] = probe
    gdist.finish(ctx)
    if ctx.rank == 0:
        print(json.dumps(line))


def sustained_rate(ctx, run_block, block_steps, min_seconds=1.0, min_blocks=10):
    """value_1s: the same step replayed in blocks for >= min_seconds; median block time, max over ranks.  The K-step
    timed region is a few tens of ms; this is the same quantity from a window long enough for clocks to settle."""
    times = []
    t_all = time.perf_counter()
    while len(times) < min_blocks or time.perf_counter() - t_all < min_seconds:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run_block()
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
        if len(times) >= 400:
            break
    times.sort()
    med = gdist.max_over_ranks(ctx, times[len(times) // 2])
    return ctx.world_size * block_steps / med, len(times)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="launch every step eagerly instead of replaying a hipGraph")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-overlap", dest="overlap", action="store_false",
                    help="skip the extra measurement of the step with its SGA and LGA halves on two streams "
                         "(`two_stream_ms_per_step`: an extra field, never `value`)")
    ap.add_argument("--stub-step", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))
    if args.stub_step:
        return stub_main(args)

    # The op benchmark has NO data-path collective (ranks own independent cost volumes): the only cross-rank traffic is the
    # timing protocol (two barriers and one max-reduction of a scalar).  That runs over gloo on the host, so no collective
    # ever shares a stream with -- or is captured into -- the timed hipGraph; RCCL over xGMI is what the training harness
    # (harness/train.py: DistributedDataParallel, backend "nccl") uses for its gradient all-reduce.
    # (gloo announces its connections on stdout: keep stdout for the one JSON line)
    sys.stdout.flush()
    saved_fd = os.dup(1)
    os.dup2(2, 1)
    try:
        ctx = gdist.init(args.gpus, backend="gloo")
        gdist.barrier(ctx)
    finally:
        sys.stdout.flush()
        os.dup2(saved_fd, 1)
        os.close(saved_fd)
    # (GANET_BENCH_DEVICE: development only -- several ranks on ONE GPU, to exercise the N > 1 path on a single-GPU box)
    device = torch.device("cuda", int(os.environ.get("GANET_BENCH_DEVICE", ctx.local_rank)))
    torch.cuda.set_device(device)
    from ganet_amd import _native
    assert not _native.lib().is_simulator

    inp = make_inputs(device)
    # One step = ~16 kernel launches + a dozen tensor allocations issued from Python.  The GPU side
    # is ~2.5 ms, so an idle host runs ahead, but on a contended host the eager loop becomes
    # host-bound (observed 10.7 ms/step on a busy box with identical kernel times).  The step is
    # therefore captured ONCE into a hipGraph (same kernels, same buffers from the graph's private
    # pool) and replayed K times; --no-graph times the eager loop instead.
    launch_mode = "eager"
    graph = None
    if not args.no_graph:
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(max(args.warmup, 3)):
                    one_step(inp)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            # (thread_local: with N > 1 ranks the RCCL watchdog thread polls events while this thread captures)
            with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                keep = one_step(inp)          # outputs stay alive inside the graph's pool
            launch_mode = "hipGraph replay"
        except Exception as e:            # capture unsupported: fall back to eager launches
            print(f"[bench] graph capture failed ({type(e).__name__}: {e}); timing eager launches", file=sys.stderr)
            graph = None
            torch.cuda.synchronize()
    if graph is not None:
        # The GPU sat idle while the graph was captured and needs ~30 ms of work to be back at its running clocks: the first
        # 20-step region after a capture reads 1.3 % slower than every later one, and so does one after 0.5 s of idling
        # (scripts/diag_bench_timing.py, profiles/r7k_diag_bench_timing.txt); W = 5 warm-up steps are 8 ms.  Rounds 4 - 5 therefore
        # replayed for SETTLE_S seconds first and printed THAT measurement as `value`; since round 6:
        # `value` IS the contract's protocol: W warm-up replays and K timed ones straight after the capture (`--warmup 5`
        # means what the driver typed).  The same K steps once the device is back at its running clocks (SETTLE_S seconds of
        # replays, then W + K again) are reported beside it as `value_settled`; `value_1s` is the sustained rate.
        for _ in range(args.warmup):
            graph.replay()
        elapsed = gdist.timed_region(ctx, lambda: [graph.replay() for _ in range(args.steps)],
                                     sync=torch.cuda.synchronize)
        t_settle = time.perf_counter()
        while time.perf_counter() - t_settle < SETTLE_S:
            for _ in range(10):
                graph.replay()
            torch.cuda.synchronize()
        for _ in range(args.warmup):
            graph.replay()
        elapsed_settled = gdist.timed_region(ctx, lambda: [graph.replay() for _ in range(args.steps)],
                                             sync=torch.cuda.synchronize)
        value_settled = ctx.world_size * args.steps / elapsed_settled
    else:
        for _ in range(args.warmup):
            one_step(inp)
        elapsed = gdist.timed_region(ctx, lambda: [one_step(inp) for _ in range(args.steps)],
                                     sync=torch.cuda.synchronize)
        value_settled = None
    value = ctx.world_size * args.steps / elapsed
    if graph is not None:
        value_1s, n_blocks = sustained_rate(ctx, lambda: [graph.replay() for _ in range(args.steps)], args.steps)
    else:
        value_1s, n_blocks = sustained_rate(ctx, lambda: [one_step(inp) for _ in range(args.steps)], args.steps)

    line = {
        "metric": "cost-volumes/sec (SGA+LGA fwd+bwd) at 240x624x192",
        "value": round(value, 2), "unit": "cost-volumes/sec", "n_gpus": ctx.world_size,
        "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / args.steps, 4),
        "value_1s": round(value_1s, 2), "value_1s_blocks": n_blocks,
        "value_settled": round(value_settled, 2) if value_settled is not None else None,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "configs[1]: SGA fwd+bwd [1,32,65,80,208] (4x guidance [1,32,5,80,208]) + "
                               "LGA2 r=2 fwd+bwd [1,193,240,624] (filters [1,75,240,624]), one sample per GPU",
                   "parallelism": "independent cost volumes per GPU, no data-path collective",
                   "launch": launch_mode,
                   "protocol": "value = W warm-up + K timed replays straight after the graph capture (the contract); value_settled = the same "
                               "after settle_s more of replays; value_1s = median block over >= 1 s",
                   "settle_s": SETTLE_S if graph is not None else 0.0},
        "unit_alg_bytes": UNIT_BYTES,
        "unit_hbm_frac": round(value / ctx.world_size * UNIT_BYTES / (HBM_PEAK_GBS * 1e9), 4),
    }
    if ctx.rank == 0:
        # the measured value is on record before anything else runs (stderr; the ONE stdout line comes last)
        print("[bench] measured: " + json.dumps(line), file=sys.stderr, flush=True)
    if ctx.world_size > 1:
        # The gradient all-reduce of a data-parallel caller (26.3 MB fp32 = GANet-deep's parameters) over RCCL / xGMI: AFTER the
        # timed region, BEFORE rank 0's extra measurements (ranks 1..N-1 never sit in a collective with a timeout while rank 0
        # is busy), on every rank, in child processes of their own (gdist.allreduce_probe_isolated: a hung or failed RCCL call
        # cannot take the benchmark's processes or its measured value with it).  Never part of `value`.
        delay = float(os.environ.get("GANET_BENCH_TEST_DELAY_RANK0", "0"))       # tests: rank 0 arrives late at the probe
        if delay and ctx.rank == 0:
            time.sleep(delay)
        line["rccl"] = gdist.allreduce_probe_isolated(ctx, os.environ.get("GANET_BENCH_COLLECTIVE", "nccl"), device.index)
    if ctx.rank == 0:
        if graph is not None:
            # for reference: the same step launched eagerly from Python (host-speed dependent)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                one_step(inp)
            torch.cuda.synchronize()
            line["eager_ms_per_step"] = round(1e3 * (time.perf_counter() - t0) / args.steps, 4)
        # the contract's extra objects first; a failure in one of them must not cost the line its measured value
        if not args.no_roofline:
            try:
                stages = stage_timings(inp)
                line["roofline"] = roofline_from_stages(stages)
                line["stage_ms"] = {k: round(v, 4) for k, v in stages.items()}
            except Exception as e:            # noqa: BLE001
                line["roofline"] = {"error": f"{type(e).__name__}: {e}"[:300]}
        if not args.no_cpu_baseline and ctx.world_size == 1:
            try:
                line["cpu_baseline"] = cpu_baseline()
            except Exception as e:            # noqa: BLE001
                line["cpu_baseline"] = {"error": f"{type(e).__name__}: {e}"[:300]}
        if args.overlap:
            # informational: SGA half and LGA half of the step on two streams (memory-bound scans beside VALU-bound LGA)
            try:
                side2 = torch.cuda.Stream()
                cap = torch.cuda.Stream()
                cap.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(cap):
                    for _ in range(3):
                        one_step_two_streams(inp, side2)
                torch.cuda.current_stream().wait_stream(cap)
                torch.cuda.synchronize()
                g2s = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g2s, capture_error_mode="thread_local"):
                    keep2 = one_step_two_streams(inp, side2)
                for _ in range(args.warmup):
                    g2s.replay()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(args.steps):
                    g2s.replay()
                torch.cuda.synchronize()
                line["two_stream_ms_per_step"] = round(1e3 * (time.perf_counter() - t0) / args.steps, 4)
            except Exception as e:
                line["two_stream_ms_per_step"] = f"failed: {type(e).__name__}: {e}"
    gdist.finish(ctx)
    if ctx.rank == 0:
        print(json.dumps(line))


if __name__ == "__main__":
    main()
