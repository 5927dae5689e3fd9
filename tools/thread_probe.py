import faulthandler, os, sys, time, threading
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import dpcr_agb_amd
from dpcr_agb_amd import synthetic
from dpcr_agb_amd.config import MODEL_OPTIONS, TRAINING_NFI, Opt
from dpcr_agb_amd.instance import MinkowskiBaselineModel
dpcr_agb_amd.limit_host_threads()
dev = torch.device("cuda:0")
ds = synthetic.SyntheticDataset(stat_seeds=range(10_000, 10_032))
model = MinkowskiBaselineModel(Opt(MODEL_OPTIONS["SENet14"]), "minkowski", ds).to(dev).train()
model.init_train_objects(TRAINING_NFI)
pool = [synthetic.make_sparse_batch(list(range(i * 32, (i + 1) * 32)), n_points=16000).to(dev) for i in range(2)]
def step(i):
    model.set_input(pool[i % 2], dev)
    model.optimize_parameters(epoch=0, batch_size=32, num_batches=133)
for i in range(10): step(i)
torch.cuda.synchronize()
def tids():
    out = {}
    for tid in os.listdir("/proc/self/task"):
        raw = open(f"/proc/self/task/{tid}/stat").read()
        f = raw[raw.rindex(")") + 2:].split()
        out[tid] = (raw[raw.index("(")+1:raw.rindex(")")], int(f[11]) + int(f[12]))
    return out
t0 = tids()
faulthandler.dump_traceback_later(0.25, repeat=False, file=sys.stderr)
for i in range(60): step(i)
torch.cuda.synchronize()
t1 = tids()
busy = sorted(((t1[t][1] - t0.get(t, ("", 0))[1], t1[t][0], t) for t in t1), reverse=True)[:5]
print("busy threads (ticks over 60 steps):", busy, "main tid", os.getpid(), "native ids:", {th.name: th.native_id for th in threading.enumerate()})
