import numpy as np, sys, time
sys.path.insert(0, ".")
import wepp_amd as w
g = w.generate_tree(21, 4_000_000)
reads = g.reads(52, 1_000_000, read_len=150, amplicon_len=400, amplicon_step=300, p_substitution=0.001, p_n=0.005)
mat = w.Mat(g.tree)
res = mat.place_batch(reads)
cls, pid = mat.last_plans(reads.n_reads)
print("classes", np.bincount(cls, minlength=7).tolist(), w.PLAN_NAMES if hasattr(w, "PLAN_NAMES") else "")
