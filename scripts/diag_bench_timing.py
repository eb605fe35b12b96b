"""Why does bench.py's 20-step region read 3 - 5 % slower than scripts/ab_step.py's 50-replay event timing of the same graph on
the same box?  Same capture as bench.py, then the K-step host-timed region repeated at several points of the process's life,
beside event timings.  python scripts/diag_bench_timing.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
dev = torch.device("cuda:0")
inp = bench.make_inputs(dev)
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(5):
        bench.one_step(inp)
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, capture_error_mode="thread_local"):
    keep = bench.one_step(inp)

def host_region(k):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): g.replay()
    torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t0) / k

def event_region(k):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(k): g.replay()
    e1.record(); e1.synchronize(); return e0.elapsed_time(e1) / k

for _ in range(5): g.replay()
print("right after capture + 5 warmup replays:  host 20 steps %.4f ms" % host_region(20), flush=True)
print("again:                                   host 20 steps %.4f" % host_region(20))
print("again:                                   host 20 steps %.4f" % host_region(20))
print("events, 50 replays:                      %.4f" % event_region(50))
print("host 200 steps:                          %.4f" % host_region(200))
t0 = time.time()
while time.time() - t0 < 1.0: host_region(20)
print("after 1 s of replays: host 20 steps      %.4f   events 50: %.4f" % (host_region(20), event_region(50)))
time.sleep(0.5)
print("after 0.5 s idle: host 20 steps          %.4f   then %.4f   then events 50: %.4f" % (host_region(20), host_region(20), event_region(50)))
# a second capture of the same step (its own pool)
g2 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g2, capture_error_mode="thread_local"):
    keep2 = bench.one_step(inp)
g_, g = g, g2
for _ in range(5): g.replay()
print("second graph (own pool): host 20 steps   %.4f   events 50: %.4f" % (host_region(20), event_region(50)))
g = g_
print("first graph again: host 20 steps         %.4f   events 50: %.4f" % (host_region(20), event_region(50)))
