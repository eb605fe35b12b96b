"""Static check of the hand-counted-vmcnt kernels: inside the main loop of each LDS-DMA kernel the compiler must not have added
vector-memory operations of its own (register spills = scratch_load / scratch_store count in vmcnt and would silently break the
explicit s_waitcnt vmcnt(n) bookkeeping).  Also: the copy batches set M0 without saving it (lga_kernels.h: GA_M0_SAVE_ASM), which is
only sound while nothing else in the kernel reads or writes M0 -- every mention of m0 must be one of the batches' own
`s_mov_b32 m0, <sgpr>` / `s_add_u32 m0, m0, ...`.  python scripts/isa_loop_check.py [asm file]   (no GPU needed; exits 1 on a finding)"""
import os, re, subprocess, sys, tempfile
from collections import Counter
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# The kernels to check are DERIVED from the listing: every kernel of namespace ga whose body contains a global_load_lds_*
# instruction is one whose vector-memory counter the source counts by hand (a typed list went stale when
# kernels were deleted, and the safety of the un-clobbered M0 rests on this script seeing every such kernel).


def asm_text(path=None):
    if path:
        return open(path).read()
    from ganet_amd import build
    with tempfile.NamedTemporaryFile(suffix=".s") as tf:
        subprocess.run(["hipcc"] + build.HIPCC_FLAGS + ["-I", build.CSRC, "-S", "--cuda-device-only",
                        os.path.join(build.CSRC, "ganet_capi.hip"), "-o", tf.name], check=True, stderr=subprocess.DEVNULL)
        return open(tf.name).read()


def main():
    txt = asm_text(sys.argv[1] if len(sys.argv) > 1 else None)
    lines = txt.split("\n")
    bad = 0
    checked = 0
    for i, l in enumerate(lines):
        m = re.match(r"^(_ZN2ga\S+):", l)
        if not m:
            continue
        end = next(j for j in range(i, len(lines)) if "s_endpgm" in lines[j])
        body = lines[i:end]
        if not any("global_load_lds" in bl for bl in body):
            continue
        checked += 1
        foreign_m0 = [bl.strip() for bl in body if re.search(r"\bm0\b", bl.split(";")[0])
                      and not re.match(r"\s*(s_mov_b32 m0, s\d+|s_add_u32 m0, m0, )", bl)]
        if foreign_m0:
            nm = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.split("(")[0].replace("void ga::", "")
            print(f"{nm:40s} m0 touched outside the copy batches: {foreign_m0[:3]}   <-- UNSAFE")
            bad += 1
        labels = {mm.group(1): k for k, bl in enumerate(body) for mm in [re.match(r"^(\.LBB\d+_\d+):", bl)] if mm}
        loops = []
        for k, bl in enumerate(body):
            mm = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", bl)
            if mm and mm.group(1) in labels and labels[mm.group(1)] < k:
                loops.append((k - labels[mm.group(1)], labels[mm.group(1)], k))
        if not loops:
            continue
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.split("(")[0].replace("void ga::", "")
        seen = set()
        for _, a, b in sorted(loops, reverse=True):
            if any(a >= a0 and b <= b0 for a0, b0 in seen):
                continue                                   # nested inside a loop already reported
            ops = Counter(bl.split()[0] for bl in body[a:b + 1] if bl.strip() and not bl.strip().startswith((";", ".")))
            if ops.get("v_pk_fma_f32", 0) < 20:
                continue
            seen.add((a, b))
            scr = sum(v for k, v in ops.items() if k.startswith("scratch_"))
            valu = sum(v for k, v in ops.items() if k.startswith("v_"))
            print(f"{name:40s} loop@{a:5d}: {sum(ops.values()):4d} instr, {valu:4d} VALU ({ops.get('v_pk_fma_f32', 0)} pk_fma, {ops.get('v_pk_mul_f32', 0)} pk_mul)  "
                  f"{ops.get('ds_read_b64', 0) + ops.get('ds_read2_b32', 0):3d} ds_read_b64 / ds_read2_b32  {sum(v for k, v in ops.items() if 'load_lds' in k):3d} lds-dma  "
                  f"{ops.get('s_waitcnt', 0):3d} waits  scratch-in-loop {scr}" + ("   <-- UNSAFE" if scr else ""))
            bad += scr > 0
    print(f"{checked} kernels with LDS-DMA copies checked")
    if checked == 0:
        print("no LDS-DMA kernel found in the listing: the check did not see what it is there for   <-- UNSAFE")
        bad += 1
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
