# one case of a soak seed with its diagnosis.  usage: bash tools/runs/r4s2_case.sh <cases> <seed> <case>
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4s2_case
SOAK_ONLY=$3 timeout -k 10 300 python - $1 $2 <<'PY' 2>&1 | tee gpurun_out/r4s2_case/case.txt | tail -30
import os, sys, runpy
import numpy as np
np.set_printoptions(linewidth=200)
sys.argv = ["tools/soak_parity.py", sys.argv[1], sys.argv[2]]
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import test_gpu_parity as T
from oracle.crbm_oracle import synthetic_onehot
# the failing case by hand: which array holds the NaN, and where
K, M, A, ds, pool, Lf, B, n, L, k = 60, 40, 1, True, 1, 224, 5, 12, 777, 3
os.environ["CRBM_STATS"] = "split"
rng = np.random.default_rng(0)
for bshift, wscale in ((4.0, 1.0), (6.5, 1.4), (2.5, 0.4)):
    m, o = T.make_pair(K, M, ds=ds, batchsize=B, cd_k=k, Lf=Lf, bshift=bshift, wscale=wscale, pooling=pool, input_dims=A)
    D = synthetic_onehot(n, L, seed=39, A=A)
    m._trainingFct(D)
    o.train_step(D)
    for name, g, w in (("W", m.motifs.get_value(), o.W), ("b", m.bias.get_value(), o.b), ("c", m.c.get_value(), o.c)):
        print(bshift, wscale, name, "gpu nan", int(np.isnan(g).sum()), "oracle nan", int(np.isnan(np.asarray(w)).sum()), "max|d|", float(np.nanmax(np.abs(g - w))))
    h, hp = m.get_fantasy()
    print("   hidden on (gpu / oracle):", float(h.mean()), float(o.fantasy_h.mean()))
PY
