import json, os, sys, torch
ROOT = os.getcwd()
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import train_eval
from train_eval import acceptance_data, _load_cpu_leg_module
cfg = json.load(open("tests/golden/r2_cpu_leg.json"))["config"]
dev = torch.device("cuda", 0)
data = acceptance_data(cfg, dev)
gen = _load_cpu_leg_module()
from dpcr_agb_amd.config import TRAINING_NFI
import random
def run(trial, **kw):
    train, val, train_h, val_mean = data
    model = gen.build_model(cfg, train_h, trial).to(dev)
    model.set_kernel_options(**kw)
    model.init_train_objects(TRAINING_NFI)
    nb = len(train)
    random.seed(gen.trial_seeds(trial)["drop_seed"])
    for epoch in range(cfg["epochs"]):
        model.train()
        for i in gen.shuffle_rng(trial, epoch).permutation(nb):
            model.set_input(train[i], dev)
            model.optimize_parameters(epoch, cfg["batch"], nb)
    model.calibrate_bn(train, dev, epochs=cfg["calibrate_passes"])
    return model.evaluate(val, dev, val_mean)["r2"]
variants = {"fp32 deterministic": dict(precision="fp32", deterministic_wgrad=True),
            "fp32 atomics": dict(precision="fp32", deterministic_wgrad=False),
            "fp32 atomics again": dict(precision="fp32", deterministic_wgrad=False),
            "fp32 no pair-compacted kernel": dict(precision="fp32", deterministic_wgrad=True, cmp_mode=0),
            "fp32 unfused tail": dict(precision="fp32", deterministic_wgrad=True, fused_tail=False),
            "bf16x3": dict(precision="bf16x3")}
for trial in (0, 1):
    for name, kw in variants.items():
        r2 = run(trial, **kw)
        print(json.dumps(dict(trial=trial, variant=name, r2=[round(float(v), 4) for v in r2])), flush=True)
