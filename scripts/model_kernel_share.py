"""Splits a rocprofv3 `--kernel-trace --stats` kernel_stats CSV of a full-model run (harness.infer / harness.train)
into who spent the device time: this library's guided-aggregation kernels (namespace ga::), MIOpen / rocBLAS convolution
and GEMM kernels, and the rest of PyTorch (elementwise, BatchNorm, interpolation, copies).
    python scripts/model_kernel_share.py <kernel_stats.csv> [passes]      -> JSON on stdout
`passes` = how many model passes the trace contains (to report ms per pass)."""
import csv
import json
import re
import sys


def classify(name):
    if "ga::" in name:
        return "guided aggregation (libganet_hip)"
    low = name.lower()
    if re.search(r"miopen|igemm|conv|gemm|cijk_|winograd|naive_conv|im2col|col2im|xdlops|batched_transpose|transpose", low):
        return "MIOpen / BLAS (convolutions)"
    if re.search(r"batch_norm|batchnorm|bn_", low):
        return "BatchNorm"
    if re.search(r"upsample|interpolat", low):
        return "interpolation"
    return "other PyTorch kernels"


def main():
    path = sys.argv[1]
    passes = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    groups, top = {}, []
    total = 0.0
    with open(path) as f:
        for row in csv.DictReader(f):
            ns = float(row["TotalDurationNs"])
            total += ns
            g = classify(row["Name"])
            groups[g] = groups.get(g, 0.0) + ns
            top.append((ns, int(row["Calls"]), row["Name"][:110]))
    top.sort(reverse=True)
    out = {"total_device_ms_per_pass": round(total / passes / 1e6, 3),
           "groups": {k: {"ms_per_pass": round(v / passes / 1e6, 3), "share": round(v / total, 4)}
                      for k, v in sorted(groups.items(), key=lambda kv: -kv[1])},
           "top_kernels": [{"ms_per_pass": round(ns / passes / 1e6, 3), "calls": c, "name": n} for ns, c, n in top[:12]]}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
