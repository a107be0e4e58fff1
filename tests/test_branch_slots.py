"""The bound behind `ORC_MAX_SAVE` (csrc/dev_types.h, the planner in csrc/batch.cpp `build_device`): a branch point's
frame is kept in a slot while all of its subtrees but the last are walked; with the subtree that needs most slots walked
last, a joint tree of n joints needs at most floor(log2(n + 1)) slots, and no more than four up to the 32 joints the build accepts
(the complete binary tree of 31 joints needs exactly four).
The recurrence of the planner, restated, over random and worst-case trees."""
import math

import numpy as np


def slots_needed(children, k):
    need = sorted(slots_needed(children, c) for c in children[k])
    return max([v + (1 if i + 1 < len(need) else 0) for i, v in enumerate(need)], default=0)


def test_four_slots_hold_any_tree_of_32_joints():
    rng = np.random.default_rng(5)
    worst = {}
    for trial in range(4000):
        n = int(rng.integers(1, 33))
        shape = trial % 4
        parent = [-1]
        for k in range(1, n):
            if shape == 0:
                parent.append(int(rng.integers(0, k)))                       # uniform random recursive tree
            elif shape == 1:
                parent.append(int(rng.integers(max(0, k - 3), k)))           # long and thin
            elif shape == 2:
                parent.append((k - 1) // 2)                                  # the complete binary tree: the worst case
            else:
                parent.append(int(rng.integers(0, min(k, 4))))               # bushy near the root
        children = [[] for _ in range(n)]
        for k in range(1, n):
            children[parent[k]].append(k)
        need = slots_needed(children, 0)
        assert need <= 4, (n, parent)
        assert need <= math.floor(math.log2(n + 1)), (n, parent, need)
        worst[n] = max(worst.get(n, 0), need)
    assert worst[31] == 4 and worst[15] == 3 and worst[7] == 2          # complete binary trees reach the bound minus one
