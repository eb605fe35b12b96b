"""Static instruction mix of the hot kernels from `hipcc -S` (VALU / packed FMA / DPP / SALU / LDS / vector memory /
waits / branches / scratch), written to stdout.  python scripts/isa_mix.py > profiles/<tag>_instruction_mix.txt"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ganet_amd import build

HOT = ["sga_col_fwdILi5ELb1ELb1", "sga_col_bwdgILi5ELb0ELb1", "sga_row_fwdILi5ELi32ELi4ELi1ELb0ELb1",
       "sga_row_bwdgILi5ELi32ELi4ELi1ELb0", "sga_bwd_pointILi4ELb0", "sga_merge_px4", "lga_apply_ppILi2", "lga_apply_pp_piILi2",
       "lga_apply_pp_poILi2", "lga_apply_pp_xoILi2", "lga_apply_pp_xILi2", "lga_filter_grad_ppILi2", "lga_filter_grad_pp_xpILi2", "lga_filter_grad_pp_gypILi2",
       "lga_filter_grad_pp_gypxILi2", "lga_filter_grad_pp_xILi2", "sga_row_fwdILi2ELi32ELi4ELi1ELb0ELb0ELi64", "sga_row_bwdgILi2ELi32ELi4ELi1ELb0ELb0ELi64",
       "lga_filter_gradILi2", "lga_applyILi2ELb0"]
print("static instruction counts per kernel (whole kernel body, all paths; gfx950, hipcc -O3; the row-forward kernels are\n"
      "compiled with -fno-slp-vectorize, see ganet_amd/build.py)\n")
print(f"{'kernel':58s} {'VALU':>6s} {'pk_fma':>6s} {'dpp':>5s} {'SALU':>6s} {'LDS':>5s} {'VMEM':>5s} {'waits':>5s} {'vmcnt0':>6s} {'branch':>6s} {'scratch':>7s} {'VGPR':>5s}")
for src, extra in build.SOURCES.items():
    with tempfile.NamedTemporaryFile(suffix=".s") as tf:
        subprocess.run(["hipcc"] + build.HIPCC_FLAGS + extra + ["-I", build.CSRC, "-S", "--cuda-device-only",
                        os.path.join(build.CSRC, src), "-o", tf.name], check=True, stderr=subprocess.DEVNULL)
        txt = open(tf.name).read().split("\n")
    cur, res, vg = None, {}, {}
    for l in txt:
        m = re.match(r"^(_ZN2ga\S+):", l)
        if m:
            cur = m.group(1); res[cur] = dict(valu=0, pk=0, dpp=0, salu=0, lds=0, vmem=0, wait=0, w0=0, br=0, scr=0)
            continue
        m = re.match(r"\s*\.vgpr_count:\s*(\d+)", l)
        t = l.strip()
        if cur and t and not t.startswith((";", ".")):
            op = t.split()[0]
            r = res[cur]
            if op.startswith("v_"):
                r["valu"] += 1
                if op.startswith("v_pk_fma"): r["pk"] += 1
                if "dpp" in t or "quad_perm" in t or "row_" in t: r["dpp"] += 1
            elif op.startswith("s_waitcnt"):
                r["wait"] += 1
                if "vmcnt(0)" in t: r["w0"] += 1
            elif op.startswith("s_cbranch") or op.startswith("s_branch"): r["br"] += 1
            elif op.startswith("s_"): r["salu"] += 1
            elif op.startswith("ds_"): r["lds"] += 1
            elif op.startswith(("global_", "buffer_", "flat_")): r["vmem"] += 1
            elif op.startswith("scratch_"): r["scr"] += 1
            if op == "s_endpgm": cur = None
    # VGPR counts from the metadata notes
    for m in re.finditer(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)", "\n".join(txt)):
        vg[m.group(1)] = int(m.group(2))
    for k, r in res.items():
        if any(h in k for h in HOT):
            name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip().split("(")[0].replace("void ga::", "")
            print(f"{name[:58]:58s} {r['valu']:6d} {r['pk']:6d} {r['dpp']:5d} {r['salu']:6d} {r['lds']:5d} {r['vmem']:5d} {r['wait']:5d} {r['w0']:6d} {r['br']:6d} {r['scr']:7d} {vg.get(k, 0):5d}")
