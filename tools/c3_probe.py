import sys, os, time, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import neuron_poker_amd as npa
eng = npa.Engine(0, kernel_times=True)
g = np.random.default_rng(65536)
hole, board = [], []
for i in range(65536):
    b = 3 if i % 2 == 0 else 4
    c = g.choice(52, 2 + b, replace=False)
    hole.append(c[:2]); board.append(list(c[2:]) + [255] * (5 - b))
q = npa.pack_queries(hole, board, 6, 20000)
for _ in range(2): eng.eval_batch(q, seed=1)
ks = []
for i in range(5):
    eng.eval_batch(q, seed=i); ks.append(eng.last_kernel_ms)
print(os.environ.get("MCQ_LIBRARY", "in-tree"), "configs[3] kernel ms", np.median(ks))
