import os, sys, time
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "board-game-simulator-python_amd")]
os.environ.setdefault("BGS_ROLLOUT_WPS", "2")
import torch
from simulator.batch import ConnectBatch
streams = [torch.cuda.Stream() for _ in range(2)]
batches = []
for s in streams:
    with torch.cuda.stream(s):
        batches.append(ConnectBatch(6, 7, 4, 1 << 20, use_torch=True))
for i in range(10):
    batches[i % 2].rollout(i, from_initial=True)
torch.cuda.synchronize()
K = 400
t0 = time.perf_counter()
for i in range(K):
    k = i % 2
    with torch.cuda.stream(streams[k]):
        batches[k].rollout(100 + i, from_initial=True)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"with-context: enqueue {1e6*(t1-t0)/K:.1f} us/step, total {1e6*(t2-t0)/K:.1f} us/step")
t0 = time.perf_counter()
for i in range(K):
    batches[i % 2].rollout(1000 + i, from_initial=True)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"bare calls  : enqueue {1e6*(t1-t0)/K:.1f} us/step, total {1e6*(t2-t0)/K:.1f} us/step")
