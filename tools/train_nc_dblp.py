"""DiffGraph_NC end to end on the committed DBLP edge lists, author features and labels: NCCoach (the training step
replayed as a HIP graph) for a few epochs, validation / test accuracy and F1 after every epoch, time per epoch.
The split is this script's own (seeded: 20 % of the authors train, 10 % validate, 70 % test) -- the reference's
index files are not part of its repository.
Usage: python tools/train_nc_dblp.py [epochs]"""
import os
import sys
import time
from types import SimpleNamespace

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_next  # noqa: E402
from diffgraph_amd.train_nc import NCCoach  # noqa: E402

epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev, NA, mats, adjs, features, _, labels = bench_next._nc_dblp_setup()
onehot = torch.zeros(NA, int(labels.max()) + 1)
onehot[torch.arange(NA), labels] = 1.0
perm = np.random.default_rng(0).permutation(NA)
n_trn, n_val = NA // 5, NA // 10
handler = SimpleNamespace(feature_list=features, labels=onehot, he_adjs=adjs, train_idx=[perm[:n_trn]],
                          val_idx=[perm[n_trn: n_trn + n_val]], test_idx=[perm[n_trn + n_val:]])
for use_graph in (True, False):
    torch.manual_seed(0)
    coach = NCCoach(handler, device=dev, use_graph=use_graph, seed=0)
    coach.prepareModel()
    print(f"--- {'step replayed as a HIP graph' if use_graph else 'step launched eagerly'}: {n_trn} training authors, "
          f"batch {coach.args.batch}, {-(-n_trn // coach.args.batch)} steps per epoch")
    for ep in range(epochs):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = coach.trainEpoch(0)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        val, test = coach.testEpoch(0)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"epoch {ep}: bceLoss {res['bceLoss']:.4f} diffLoss {res['diffLoss']:.5f} | val acc {float(val['acc']):.4f} "
              f"macro-F1 {val['macro']:.4f} | test acc {float(test['acc']):.4f} macro-F1 {test['macro']:.4f} micro-F1 "
              f"{test['micro']:.4f} | train {1e3 * (t1 - t0):.1f} ms, test {1e3 * (t2 - t1):.1f} ms")
