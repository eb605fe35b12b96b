"""Who spent the device time of a model pass: torch.profiler (roctracer) around the TIMED passes only -- MIOpen's find-mode
benchmarking and first-use compilation happen in the warm-up and stay out of the picture (a rocprofv3 trace of the whole
process is flooded by them) -- aggregated into: this library's guided-aggregation kernels (namespace ga::), MIOpen / BLAS
convolution kernels, BatchNorm, interpolation, and the rest of PyTorch."""
import re


def classify(name):
    if "ga::" in name:
        return "guided aggregation (libganet_hip)"
    low = name.lower()
    if re.search(r"batch_norm|batchnorm|bn_|bnfwd|bnbwd", low):
        return "BatchNorm"
    if re.search(r"miopen|igemm|conv|gemm|cijk_|winograd|im2col|col2im|xdlops|batched_transpose|transpose|ck::|kernel_grouped", low):
        return "MIOpen / BLAS (convolutions)"
    if re.search(r"batch_norm|batchnorm|bn_|bnfwd|bnbwd", low):
        return "BatchNorm"
    if re.search(r"upsample|interpolat", low):
        return "interpolation"
    return "other PyTorch kernels"


def profile_passes(fn, passes):
    """Runs fn() `passes` times under the profiler; returns the share table (per pass)."""
    import torch
    from torch.profiler import ProfilerActivity, profile
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(passes):
            fn()
        torch.cuda.synchronize()
    groups, kernels, total = {}, {}, 0.0
    for row in prof.key_averages():
        dt = 0.0
        for attr in ("self_device_time_total", "self_cuda_time_total"):        # (us; name differs between torch versions)
            v = getattr(row, attr, None)
            if v:
                dt = float(v)
                break
        if dt <= 0.0:
            continue
        g = classify(row.key)
        groups[g] = groups.get(g, 0.0) + dt
        kernels[row.key] = kernels.get(row.key, 0.0) + dt
        total += dt
    if total == 0.0:
        return {"error": "the profiler returned no device events"}
    top = sorted(kernels.items(), key=lambda kv: -kv[1])[:10]
    return {"device_ms_per_pass": round(total / passes / 1e3, 3),
            "groups": {k: {"ms_per_pass": round(v / passes / 1e3, 3), "share": round(v / total, 4)}
                       for k, v in sorted(groups.items(), key=lambda kv: -kv[1])},
            "top_kernels": [{"ms_per_pass": round(v / passes / 1e3, 3), "name": k[:100]} for k, v in top]}
