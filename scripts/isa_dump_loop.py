"""Dumps one loop of one kernel from a `hipcc -S` listing, with an op histogram (development aid for the hand-scheduled
kernels: what exactly is the steady body made of?).
python scripts/isa_dump_loop.py file.s <kernel-name-substring> [loop-index | all] [--hist-only]
Loops are listed longest first; index 0 = the longest."""
import re, subprocess, sys
from collections import Counter


def kernels(lines):
    i = 0
    while i < len(lines):
        m = re.match(r"^(_ZN2ga\S+):", lines[i])
        if m:
            end = next(j for j in range(i, len(lines)) if "s_endpgm" in lines[j])
            yield m.group(1), lines[i:end + 1]
            i = end
        i += 1


def loops_of(body):
    labels = {mm.group(1): k for k, bl in enumerate(body) for mm in [re.match(r"^(\.LBB\d+_\d+):", bl)] if mm}
    out = []
    for k, bl in enumerate(body):
        mm = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", bl)
        if mm and mm.group(1) in labels and labels[mm.group(1)] < k:
            out.append((labels[mm.group(1)], k))
    return sorted(out, key=lambda ab: ab[0] - ab[1])


def classify(op):
    if op.startswith("v_"):
        return "VALU"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op.startswith("s_"):
        return "SALU"
    if op.startswith("ds_"):
        return "LDS"
    if "load_lds" in op:
        return "DMA"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "VMEM"
    return "other"


def main():
    path, want = sys.argv[1], sys.argv[2]
    which = sys.argv[3] if len(sys.argv) > 3 and not sys.argv[3].startswith("--") else "0"
    hist_only = "--hist-only" in sys.argv
    lines = open(path).read().split("\n")
    for name, body in kernels(lines):
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.split("(")[0].replace("void ga::", "")
        if want not in dem:
            continue
        lp = loops_of(body)
        print(f"== {dem}: {len(lp)} loops")
        for n, (a, b) in enumerate(lp):
            if which != "all" and n != int(which):
                continue
            ins = [bl.strip() for bl in body[a:b + 1] if bl.strip() and not bl.strip().startswith((";", ".")) and not bl.strip().endswith(":")]
            cls = Counter(classify(t.split()[0]) for t in ins)
            ops = Counter(t.split()[0] for t in ins)
            print(f"-- loop {n} @{a}..{b}: {len(ins)} instr  " + "  ".join(f"{k}={v}" for k, v in sorted(cls.items())))
            print("   " + "  ".join(f"{k}:{v}" for k, v in ops.most_common(40)))
            if not hist_only:
                for t in ins:
                    print("      " + t)


if __name__ == "__main__":
    main()
