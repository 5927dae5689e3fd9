"""Target statistics of the reference's own data split: /root/reference/nfi-data/train_split.csv (BMag_ha, V_ha — the two
regression targets of the AGB configuration, conf/data/instance/NFI/*.yaml; 4 271 plots).  The only real data in the reference
tree; written to tests/golden/nfi_target_stats.json (numbers only).  Run in the build container:
    python tests/golden/make_nfi_target_stats.py"""
import json
import os

import pandas as pd

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SRC = "/root/reference/nfi-data"


def main():
    out = {}
    for split in ("train", "val", "test"):
        d = pd.read_csv(os.path.join(SRC, f"{split}_split.csv"))[["BMag_ha", "V_ha"]].astype("float64")
        out[split] = dict(rows=int(len(d)), mean=d.mean().tolist(), std=d.std(ddof=0).tolist(), min=d.min().tolist(),
                          max=d.max().tolist(), first_rows=d.head(5).values.tolist())
    out["targets"] = ["BMag_ha", "V_ha"]
    out["source"] = "nfi-data/{train,val,test}_split.csv of the reference tree (columns BMag_ha, V_ha)"
    with open(os.path.join(ROOT, "tests", "golden", "nfi_target_stats.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps({k: out[k]["rows"] for k in ("train", "val", "test")}))


if __name__ == "__main__":
    main()
