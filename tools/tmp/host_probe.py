import sys, os, time, json, collections
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from hanabizero_amd import device_replay, learner
T = collections.defaultdict(float); N = collections.defaultdict(int)
def wrap(obj, name, label=None):
    f = getattr(obj, name)
    def g(*a, **k):
        t0 = time.perf_counter()
        r = f(*a, **k)
        T[label or name] += time.perf_counter() - t0; N[label or name] += 1
        return r
    setattr(obj, name, g)
for n in ("sample", "assemble", "policy_re_inputs", "update_priorities", "windows"):
    wrap(device_replay.DeviceReplay, n)
wrap(device_replay, "policy_re_device")
wrap(learner.GraphedUpdate, "run", "graph_run")
wrap(learner, "adjust_lr")
wrap(learner.LearnerPipeline, "step", "step_total")
import loop_bench
sys.argv = ["loop_bench.py"] + sys.argv[1:]
loop_bench.main()
steps = N["step_total"]
print({k: round(1e3 * v / steps, 3) for k, v in T.items()}, steps)
