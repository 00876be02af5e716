#!/bin/bash
# Round 3, call AJ: NTN large-batch forms: the B = 1100 oracle case with the forms off / on, step time, kernel table
ulimit -c 0
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
python - <<'PY' 2>&1 | tail -12
import os, sys
sys.path.insert(0, "tests"); sys.path.insert(0, "oracle"); sys.path.insert(0, ".")
import numpy as np, torch
import hip_util, kge_oracle as ko
from pykg2vec_amd.trainer import Trainer
for big in ("0", "1"):
    os.environ["KGE_NTN_BIG"] = big
    rng = np.random.default_rng(42)
    E, R, B = 300, 11, 1100
    hp = dict(ent_hidden_size=96, rel_hidden_size=64, lmbda=1e-3, margin=1.0)
    P = ko.init_params("ntn", rng, tot_entity=E, tot_relation=R, ent_hidden_size=96, rel_hidden_size=64)
    pos = np.stack([rng.integers(E, size=B), rng.integers(R, size=B), rng.integers(E, size=B)], 1)
    nh = pos[:, 0].copy(); nr = pos[:, 1].copy(); nt = pos[:, 2].copy()
    flip = rng.random(B) > 0.5; rnd = rng.integers(E, size=B)
    nh = np.where(flip, nh, rnd); nt = np.where(flip, rnd, nt)
    batch = (pos[:, 0], pos[:, 1], pos[:, 2], nh, nr, nt)
    hp_run = dict(hp, neg_rate=1)
    loss_ref, G_ref, scores_ref, _ = ko.train_step_grads("ntn", P, batch, **hp_run)
    m = hip_util.model_from_params("ntn", P, hp, E, R, train=pos)
    cfg = hip_util.make_config(E, R, hp_run, pos, pos[:1], pos[:1])
    tr = Trainer(m, cfg); tr.build_model()
    b = [hip_util.dev(x) for x in batch]
    loss = tr.train_step_pairwise(*b)
    names = [n.split(".")[0] for n, _ in hip_util.table_parameters(m)]
    print("big", big, "loss", loss.item(), loss_ref, {nme: (float(np.abs(g.cpu().numpy() - G_ref[nme]).max()), float(np.abs(G_ref[nme]).max())) for nme, g in zip(names, tr.flat.grad_views)})
PY
echo "KGE_NTN_BIG=1 $(KGE_NTN_BIG=1 ONLY="mfma-batch NTN" timeout 300 python tools/config_perf.py 2>&1 | tail -1)" | tee $O/aj3_perf.log
KGE_NTN_BIG=1 ONLY="mfma-batch NTN" timeout 300 rocprofv3 --kernel-trace -d $O/aj_kt -o ntn -- python tools/config_perf.py > $O/aj_kt.log 2>&1
python tools/rocpd_summary.py $(find $O/aj_kt -name "*.db") $O/aj3_ntn_kernels.md > /dev/null 2>&1
rm -rf $O/aj_kt
awk -F'|' '{print substr($2,1,70), "|", $5, "|", $7}' $O/aj3_ntn_kernels.md | head -8
