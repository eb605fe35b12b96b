"""Diagnostic (not a pytest file): DPP lane-pattern probe and a few parity cases on the
product library and -- if present -- the ds_bpermute debug build, to localise a failure
quickly when GPU minutes are scarce.  python tests/gpu_diag.py"""
import os
import sys
import traceback

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.dirname(HERE), HERE]
import torch  # noqa: E402

import parity_cases as pc  # noqa: E402
from ganet_amd._native import CApi  # noqa: E402
from oracle.oracle import Oracle  # noqa: E402
from test_gpu_parity import TorchDev  # noqa: E402

print("torch", torch.__version__, "device", torch.cuda.get_device_name(0))
dev = TorchDev()
ora = Oracle("port")
root = os.path.dirname(HERE)
for name in ["libganet_hip.so", "libganet_hip_nodpp.so"]:
    path = os.path.join(root, "ganet_amd", name)
    if not os.path.exists(path):
        continue
    print("=====", name)
    api = CApi(path)
    try:
        scratch = dev.zeros((512,), np.int32)
        host = np.zeros(512, np.int32)
        try:
            api.call("ganet_selftest_dpp", scratch.data_ptr(), host.ctypes.data, dev.stream)
            print("dpp selftest OK")
        except Exception as e:
            print("dpp selftest FAILED:", e)
            for p in range(8):
                print(" pattern", p, host[p * 64:p * 64 + 20].tolist())
        for shape in [(1, 2, 5, 4, 8), (1, 2, 33, 6, 12), (1, 1, 65, 3, 7)]:
            x, gs, go = pc.sga_inputs(shape, seed=1)
            for d in range(4):
                try:
                    pc.check_sga_scan(api, dev, ora, x, gs[d], d)
                    print("scan", shape, "dir", d, "OK")
                except AssertionError as e:
                    print("scan", shape, "dir", d, "FAIL", str(e)[:200])
            out, tmp, mask = ora.sga_forward(x, *gs)
            grads = ora.sga_backward(x, *gs, tmp, mask, go)
            want = {"out": out, "mask": mask.astype(np.uint8), "gx": grads[0]}
            for d in range(4):
                want[f"gw{d}"] = grads[1 + d]
            try:
                print("fwd+bwd", shape, pc.check_sga_forward_backward(api, dev, x, gs, go, want))
            except AssertionError as e:
                print("fwd+bwd", shape, "FAIL", str(e)[:300])
        rng = np.random.default_rng(0)
        xl = rng.standard_normal((1, 9, 12, 40)).astype(np.float32)
        f = pc.l1norm(rng.standard_normal((1, 75, 12, 40)), 1)
        gy = rng.standard_normal(xl.shape).astype(np.float32)
        y, ins = ora.lga_chain_forward(xl, f, 2, 2)
        gx, gf = ora.lga_chain_backward(ins, f, gy, 2)
        try:
            print("lga2", pc.check_lga_chain(api, dev, xl, f, gy, 2, 2, {"y": y, "gx": gx, "gf": gf}))
        except AssertionError as e:
            print("lga2 FAIL", str(e)[:300])
    except Exception:
        traceback.print_exc()
