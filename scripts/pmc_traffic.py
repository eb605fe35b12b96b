"""Per-kernel HBM-side traffic from the memory-side PMC passes (scripts/gpu_session.sh <tag> pmc):
   bytes = TCC_EA0_RDREQ_128B*128 + RDREQ_64B*64 + RDREQ_32B*32 (+ other reads * 64) and WRREQ_64B*64 + other writes * 32,
   mean per dispatch.  (Request counts x request size instead of FETCH_SIZE/WRITE_SIZE: MI355X_MICROARCH.md notes that
   FETCH_SIZE tallies 128-byte requests at 64 B on gfx950.)  Infinity-Cache hits are included: this is traffic at the
   L2 <-> fabric boundary, an upper bound of DRAM traffic.
   python scripts/pmc_traffic.py gpurun_out/<tag>/summary.txt profiles/traffic_pmc.json"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
txt = open(sys.argv[1]).read()
out = {}
for blk in txt.split("== ")[1:]:
    lines = blk.strip().split("\n")
    name = lines[0].strip().replace("ga::", "")
    d = {}
    for l in lines[1:]:
        p = l.split()
        d[p[0]] = float(p[1])
    if "TCC_EA0_RDREQ_sum" not in d:
        continue
    r128, r64, r32 = d.get("TCC_EA0_RDREQ_128B_sum", 0), d.get("TCC_EA0_RDREQ_64B_sum", 0), d.get("TCC_EA0_RDREQ_32B_sum", 0)
    rd = r128 * 128 + r64 * 64 + r32 * 32 + max(0.0, d["TCC_EA0_RDREQ_sum"] - r128 - r64 - r32) * 64
    w64 = d.get("TCC_EA0_WRREQ_64B_sum", 0)
    wr = w64 * 64 + max(0.0, d.get("TCC_EA0_WRREQ_sum", 0) - w64) * 32
    out[name] = {"read_bytes": int(rd), "write_bytes": int(wr), "duration_us_profiled": d.get("duration_us")}
import bench      # csrc_tree_hash(): the kernel sources these counters were measured on (bench.py warns when they differ from HEAD's)
json.dump({"source": sys.argv[1], "csrc_tree": bench.csrc_tree_hash(), "method": "TCC_EA0_RDREQ/WRREQ request counts x request size, mean per dispatch (rocprofv3 --pmc, own passes)",
           "kernels": out}, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out, indent=1))
