import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
from test_lattice_gpu import _build
from test_gpu_parity import ambiguous
names = ["lane_groups", "lane_fwdx", "lane_bwd", "lane_pair", "lane_nstates", "lane_logw", "t_buckets", "t_tile_base", "t_b_arc", "t_b_rank",
         "t_b_src", "t_t_pos", "t_t_src", "t_a_off", "t_split_arcs", "pair_w"]
from carmel_amd.model import Corpus
w, c = ambiguous(1)
rng = np.random.default_rng(1)
c.weight[:] = rng.uniform(0.5, 3.0, c.n_pairs)
extra = Corpus.from_lists([([2, 3, 2], [3]), ([], []), ([2], []), ([3, 3, 3, 3, 3, 3, 3, 3], [2, 2])], np.array([1.0, 2.0, 0.5, 1.5]))
c = Corpus(np.concatenate([c.in_off, c.in_off[-1] + extra.in_off[1:]]), np.concatenate([c.in_sym, extra.in_sym]),
           np.concatenate([c.out_off, c.out_off[-1] + extra.out_off[1:]]), np.concatenate([c.out_sym, extra.out_sym]),
           np.concatenate([c.weight, extra.weight]))
a, b = _build(w, c, False), _build(w, c, True)
for k in range(16):
    if a["fp"][k] != b["fp"][k]:
        print("differs:", names[k])
print("done")
