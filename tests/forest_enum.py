"""Enumeration helpers for the sampler tests: derivations of a packed forest, a matcher of samples against a forest, and
the EXACT stationary distribution of one sweep of forest-em's Gibbs sampler on a corpus small enough to enumerate (built
from the definition of a block's proposal, forest-em.hpp:750-766 / forest.hpp:725-758 / gibbs.hpp:589-592, 769-792 -- not
from the oracle's code)."""
import itertools

import numpy as np

def children(nxt, i):
    j = i + 1
    while j < int(nxt[i]):
        yield j
        j = int(nxt[j])


def match(label, ref, nxt, i, sample, pos):
    """end positions at which node i can finish consuming `sample` from `pos` in pre-order (choose_random's visit order,
    forest.hpp:725-758: an OR node picks one child, an AND node records its rule and expands every child in order)"""
    if ref[i] >= 0:
        return match(label, ref, nxt, int(ref[i]), sample, pos)
    if label[i] == 0:
        out = set()
        for c in children(nxt, i):
            out |= match(label, ref, nxt, c, sample, pos)
        return out
    if pos >= len(sample) or sample[pos] != label[i]:
        return set()
    ends = {pos + 1}
    for c in children(nxt, i):
        nxt_ends = set()
        for e in ends:
            nxt_ends |= match(label, ref, nxt, c, sample, e)
        ends = nxt_ends
        if not ends:
            break
    return ends


# ---- the exact stationary distribution of a sweep on an enumerable corpus -------------------------------------------------
def match_bfs(label, ref, nxt, sample):
    """True iff `sample` lists, in BREADTH-FIRST order, the rules of a derivation of the forest: the walk of the several-lanes
    sampler -- a frontier of nodes; an AND entry records its rule and hands all of its children to the next frontier, an OR
    entry hands on ONE child (its choice), back-references resolve to their shared sub-forest.  The choice of an OR entry is
    checked as soon as it is made: a chosen AND child's rule must be the sample's entry at the position the child will be
    recorded at (the number of AND entries before it in the next frontier)."""
    def res(i):
        while ref[i] >= 0:
            i = int(ref[i])
        return i

    def kids(i):
        return [res(c) for c in children(nxt, i)]

    import sys
    sys.setrecursionlimit(max(20000, sys.getrecursionlimit()))
    sample = list(sample)

    def level(front, pos):
        """front: the nodes of this frontier; pos: sample position of the first AND entry of this frontier"""
        if not front:
            return pos == len(sample)
        n_and = sum(1 for v in front if label[v] != 0)
        if pos + n_and > len(sample) or any(sample[pos + k] != label[v] for k, v in enumerate(v for v in front if label[v] != 0)):
            return False
        nxt_pos = pos + n_and

        def fill(i, nf, n_and_next):
            if i == len(front):
                return level(nf, nxt_pos)
            v = front[i]
            if label[v] != 0:
                ks = kids(v)
                na = n_and_next
                for c in ks:  # the AND children are recorded next level in this order: check them now
                    if label[c] != 0:
                        if nxt_pos + na >= len(sample) or sample[nxt_pos + na] != label[c]:
                            return False
                        na += 1
                return fill(i + 1, nf + ks, na)
            for c in kids(v):
                if label[c] != 0 and (nxt_pos + n_and_next >= len(sample) or sample[nxt_pos + n_and_next] != label[c]):
                    continue
                if fill(i + 1, nf + [c], n_and_next + (1 if label[c] != 0 else 0)):
                    return True
            return False
        return fill(0, [], 0)
    return level([res(0)], 0)


TOY_FORESTS = """(OR (1 (OR 4 5)) (2 6) (3 (OR 4 6)))
(OR (1 5) (2 (OR 4 5 6)))
(OR (2 (OR (7 4) (8 5))) (3 6) (1 4))
(OR (3 (OR 5 6)) (1 (OR 7 8)) 2)
(OR (7 (OR 1 2)) (8 3) (7 4 5))
"""
TOY_NORM = "((1 2 3) (4 5 6) (7 8))"
# a second enumerable corpus for what the first does not have: an OR node with five children, an AND node with nine (a
# breadth-first frontier wider than the eight lanes a forest gets), a shared sub-forest expanded twice in one derivation
TOY2_FORESTS = """(OR (1 (OR 4 5 6 7 8)) (2 #1(OR 4 5) #1))
(OR (3 1 2 4 5 6 7 8 1 2) (3 4))
(OR (1 (OR 4 5)) (2 (OR 6 7 8)) 3)
"""
TOY2_NORM = "((1 2 3) (4 5 6 7 8))"


def derivations(label, ref, nxt, i):
    """all derivations below node i, each a tuple of rule ids in visit order"""
    if ref[i] >= 0:
        return derivations(label, ref, nxt, int(ref[i]))
    kids = list(children(nxt, i))
    if label[i] == 0:
        return [d for c in kids for d in derivations(label, ref, nxt, c)]
    out = [(int(label[i]),)]
    for c in kids:
        out = [a + b for a in out for b in derivations(label, ref, nxt, c)]
    return out


def group_priors(n_rules, group_off, group_rule, lw, alpha):
    """prior pseudo-count per rule as forest-em's sampler sets it up: the weights are normalised per group first
    (define_gibbs -> normalize), then prior = alpha * p0 * |group| (gibbs.hpp:589-592).  Returns (group_of, gsize, p0, prior);
    rules outside every group have group_of = -1 (fixed probability)."""
    gsize = np.diff(np.asarray(group_off).astype(np.int64))
    group_of = np.full(n_rules, -1, np.int64)
    group_of[np.asarray(group_rule)] = np.repeat(np.arange(len(gsize)), gsize)
    in_g = group_of >= 0
    z = np.bincount(group_of[in_g], weights=np.exp(np.asarray(lw))[in_g], minlength=len(gsize))
    p0 = np.zeros(n_rules)
    p0[in_g] = np.exp(np.asarray(lw))[in_g] / z[group_of[in_g]]
    prior = np.zeros(n_rules)
    prior[in_g] = alpha * p0[in_g] * gsize[group_of[in_g]]
    return group_of, gsize, p0, prior


def stationary(derivs, prior, group_of, parallel):
    """Stationary distribution over joint assignments (one derivation per forest) of one sweep of forest-em's sampler.
    A block's new derivation d is drawn with probability proportional to prod_{r in d} (n_r + prior_r) / (n_g(r) + prior_g),
    the counts n taken WITHOUT the block's own current sample and held fixed while the block is drawn (proposal_prob,
    forest-em.hpp:750-766; choose_random draws top-down by inside values, i.e. exactly proportionally to that product).
    exact sweep: blocks one after another, each seeing the new samples of the earlier ones; parallel sweep: every block
    against the previous sweep's samples of the others.  Every rule must belong to a group.  Returns (states, pi, use):
    use[b][j] = rule-use vector of forest b's derivation j."""
    F = len(derivs)
    dims = [len(d) for d in derivs]
    states = list(itertools.product(*[range(k) for k in dims]))  # C order: the last forest's index runs fastest
    n_rules = len(prior)
    use = [[np.bincount(np.asarray(d, np.int64), minlength=n_rules).astype(np.float64) for d in ds] for ds in derivs]
    in_g = group_of >= 0
    G = int(group_of.max()) + 1
    gprior = np.bincount(group_of[in_g], weights=prior[in_g], minlength=G)

    def block_dist(b, others_counts):
        gsum = np.bincount(group_of[in_g], weights=others_counts[in_g], minlength=G) + gprior
        p = np.ones(n_rules)
        p[in_g] = (others_counts + prior)[in_g] / gsum[group_of[in_g]]
        q = np.array([np.prod(p[list(d)]) for d in derivs[b]])
        return q / q.sum()

    n = len(states)
    strides = [int(np.prod(dims[b + 1:])) for b in range(F)]
    if parallel:
        K = np.empty((n, n))
        for k, s in enumerate(states):
            tot = sum(use[b][s[b]] for b in range(F))
            row = np.ones(1)
            for b in range(F):
                row = np.multiply.outer(row, block_dist(b, tot - use[b][s[b]])).ravel()
            K[k] = row
    else:
        K = np.eye(n)
        for b in range(F):
            Kb = np.zeros((n, n))
            for k, s in enumerate(states):
                tot = sum(use[c][s[c]] for c in range(F))
                q = block_dist(b, tot - use[b][s[b]])
                base = k - s[b] * strides[b]
                Kb[k, base + np.arange(dims[b]) * strides[b]] = q
            K = K @ Kb
    assert np.allclose(K.sum(1), 1.0)
    pi = np.full(n, 1.0 / n)
    for _ in range(20000):
        nxt = pi @ K
        if np.max(np.abs(nxt - pi)) < 1e-15:
            pi = nxt
            break
        pi = nxt
    return states, pi / pi.sum(), use


def expected_weights(states, pi, use, prior, group_of, gsize):
    """what the sampler's final rule weights converge to: (E[uses] + prior) / (that sum over the rule's norm group)
    (from_gibbs forest-em.hpp:736-742 over time-averaged counts, gibbs.hpp:626-638)"""
    F = len(use)
    en = sum(pi[k] * sum(use[b][s[b]] for b in range(F)) for k, s in enumerate(states))
    in_g = group_of >= 0
    tot = np.bincount(group_of[in_g], weights=(en + prior)[in_g], minlength=len(gsize))
    out = np.zeros(len(prior))
    out[in_g] = (en + prior)[in_g] / tot[group_of[in_g]]
    return out


def toy_setup(node_off, label, ref, nxt, n_rules, group_off, group_rule, lw, alpha):
    """derivations of every forest of a toy corpus + the exact stationary final weights of both sweeps:
    returns (derivs, {False: exact-chain weights, True: parallel-sweep weights}) indexed by rule id"""
    derivs = []
    for f in range(len(node_off) - 1):
        b, e = int(node_off[f]), int(node_off[f + 1])
        derivs.append(derivations(label[b:e], ref[b:e], nxt[b:e], 0))
    group_of, gsize, p0, prior = group_priors(n_rules, group_off, group_rule, lw, alpha)
    out = {}
    for par in (False, True):
        states, pi, use = stationary(derivs, prior, group_of, par)
        out[par] = expected_weights(states, pi, use, prior, group_of, gsize)
    return derivs, out
