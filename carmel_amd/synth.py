"""Synthetic transducers and corpora for the benchmark configurations (fixed seeds).

The generator is this build's own code.  Like the reference's WFST::generate (carmel/src/fst.cc:24-79) every pair
is read off an actual start->final path, so every pair has at least one derivation.
"""
import numpy as np

from .model import Corpus, Wfst

FIRST_SYM = 2  # 0 = *e*, 1 = *w* (carmel/src/fst.h:58-59,409)


def random_wfst(n_states, out_degree, n_sym=64, p_eps=0.1, seed=1):
    """n_states states, every non-final state has `out_degree` arcs: arc 0 goes to the single final state
    (n_states-1, no out-arcs), the rest to uniform non-final states.  Labels: *e* with probability p_eps per
    side, else uniform over n_sym-1 symbols.  Weights Dirichlet(1) per (state, input) group, i.e. a proper
    conditional model."""
    rng = np.random.default_rng(seed)
    F = n_states - 1
    ns = n_states - 1  # states with arcs
    src = np.repeat(np.arange(ns, dtype=np.uint32), out_degree)
    dst = rng.integers(0, F, size=(ns, out_degree), dtype=np.uint32)
    dst[:, 0] = F
    dst = dst.reshape(-1)

    def labels():
        lab = rng.integers(FIRST_SYM, FIRST_SYM + n_sym - 1, size=ns * out_degree, dtype=np.uint32)
        lab[rng.random(ns * out_degree) < p_eps] = 0
        return lab

    isym, osym = labels(), labels()
    g = rng.exponential(size=ns * out_degree)
    key = src.astype(np.uint64) * np.uint64(1 << 20) + isym.astype(np.uint64)
    _, inv = np.unique(key, return_inverse=True)
    sums = np.bincount(inv, weights=g)
    logw = np.log(g / sums[inv])
    return Wfst(n_states, F, src, dst, isym, osym, logw)


def random_walk_corpus(w, n_pairs, min_arcs=5, max_arcs=40, seed=1, out_degree=None):
    """pairs read off uniform random walks start->final of min_arcs..max_arcs arcs (last arc = arc 0 of the
    current state, which enters the final state)"""
    rng = np.random.default_rng(seed + 1000003)
    if out_degree is None:
        out_degree = int(np.searchsorted(w.src, 1))  # arcs of state 0
    L = rng.integers(min_arcs, max_arcs + 1, size=n_pairs)
    cur = np.zeros(n_pairs, dtype=np.int64)
    maxL = int(L.max())
    ins = np.zeros((n_pairs, maxL), dtype=np.uint32)
    outs = np.zeros((n_pairs, maxL), dtype=np.uint32)
    for step in range(maxL):
        active = step < L
        last = step == L - 1
        pick = rng.integers(1, out_degree, size=n_pairs)
        pick[last] = 0
        arc = cur * out_degree + pick
        arc[~active] = 0
        ins[:, step] = np.where(active, w.isym[arc], 0)
        outs[:, step] = np.where(active, w.osym[arc], 0)
        cur = np.where(active, w.dst[arc].astype(np.int64), cur)
    mi, mo = ins != 0, outs != 0
    in_off = np.concatenate([[0], np.cumsum(mi.sum(1))]).astype(np.uint64)
    out_off = np.concatenate([[0], np.cumsum(mo.sum(1))]).astype(np.uint64)
    return Corpus(in_off, ins[mi], out_off, outs[mo])


def clustered_wfst(n_states, out_degree, members=2, n_sym=64, p_eps=0.1, seed=1, n_in_sym=4):
    """A transducer whose derivation lattices are AMBIGUOUS at any size (random_wfst's are single paths: with uniform
    destinations over 10^6 states two partial derivations never meet again).  The states with arcs form clusters of
    `members` states; the cluster's `out_degree / members` moves each carry one label pair and one destination cluster,
    and every member of the source cluster has an arc for the move to EVERY member of the destination cluster -- an HMM
    whose hidden variable is the member: a pair read off a walk determines the cluster sequence, not the members, so its
    lattice is positions x members with members^2 arcs between neighbouring positions (every lattice state has `members`
    in-arcs; the tagging cascade of the tutorial has the same shape).  Move 0 enters the single final state (`members`
    parallel arcs per state, each a parameter of its own).  Arc order is state-major, moves in order, destination member
    innermost.  Input labels come from a SMALL alphabet (n_in_sym - 1 symbols), output labels from n_sym - 1: several moves
    of a cluster share an input symbol, so the conditional model P(out | in) spreads its mass over them and a pair's
    probability is below one (with distinct input labels every pair would have probability exactly 1 whatever the weights).
    Weights Dirichlet(1) per (state, input) group like random_wfst."""
    rng = np.random.default_rng(seed)
    M = int(members)
    assert out_degree % M == 0 and (n_states - 1) % M == 0
    moves = out_degree // M
    F = n_states - 1
    ncl = (n_states - 1) // M
    dcl = rng.integers(0, ncl, size=(ncl, moves), dtype=np.int64)

    def labels(k):
        lab = rng.integers(FIRST_SYM, FIRST_SYM + k - 1, size=(ncl, moves), dtype=np.uint32)
        lab[rng.random((ncl, moves)) < p_eps] = 0
        return lab

    li, lo = labels(n_in_sym), labels(n_sym)
    ns = ncl * M
    src = np.repeat(np.arange(ns, dtype=np.uint32), out_degree)
    cl = np.repeat(np.arange(ncl, dtype=np.int64), M * out_degree)              # cluster of every arc's source
    mv = np.tile(np.repeat(np.arange(moves, dtype=np.int64), M), ns)             # move of every arc
    dm = np.tile(np.arange(M, dtype=np.int64), ns * moves)                       # destination member
    dst = (dcl[cl, mv] * M + dm).astype(np.uint32)
    dst[mv == 0] = F
    isym, osym = li[cl, mv], lo[cl, mv]
    g = rng.exponential(size=ns * out_degree)
    key = src.astype(np.uint64) * np.uint64(1 << 20) + isym.astype(np.uint64)
    _, inv = np.unique(key, return_inverse=True)
    sums = np.bincount(inv, weights=g)
    return Wfst(n_states, F, src, dst, isym, osym, np.log(g / sums[inv]))


def clustered_walk_corpus(w, n_pairs, out_degree, members=2, min_arcs=5, max_arcs=40, seed=1):
    """pairs read off random walks over clustered_wfst's moves (the last move is move 0, into the final state); the walk
    starts at state 0, i.e. member 0 of cluster 0"""
    rng = np.random.default_rng(seed + 1000003)
    M = int(members)
    moves = out_degree // M
    L = rng.integers(min_arcs, max_arcs + 1, size=n_pairs)
    cur = np.zeros(n_pairs, dtype=np.int64)  # current state (member 0 of the current cluster)
    maxL = int(L.max())
    ins = np.zeros((n_pairs, maxL), dtype=np.uint32)
    outs = np.zeros((n_pairs, maxL), dtype=np.uint32)
    for step in range(maxL):
        active = step < L
        last = step == L - 1
        pick = rng.integers(1, moves, size=n_pairs) if moves > 1 else np.zeros(n_pairs, dtype=np.int64)
        pick[last] = 0
        arc = cur * out_degree + pick * M  # the move's arc to member 0 of the destination
        arc[~active] = 0
        ins[:, step] = np.where(active, w.isym[arc], 0)
        outs[:, step] = np.where(active, w.osym[arc], 0)
        cur = np.where(active & ~last, w.dst[arc].astype(np.int64), cur)
    mi, mo = ins != 0, outs != 0
    in_off = np.concatenate([[0], np.cumsum(mi.sum(1))]).astype(np.uint64)
    out_off = np.concatenate([[0], np.cumsum(mo.sum(1))]).astype(np.uint64)
    return Corpus(in_off, ins[mi], out_off, outs[mo])


MIX_REGIONS = (
    # (clusters, members, moves incl. the one into the final state, input alphabet, walk lengths, share of the pairs)
    (300000, 1, 10, 64, (5, 40), 0.90),    # single paths like config 4's: one lattice per lane, the tile sweep's class
    (100000, 3, 4, 4, (5, 40), 0.09),      # c4a's ambiguous lattices: positions x 3 states, windowed lane groups
    (12500, 8, 4, 4, (40, 600), 0.01),     # `long`'s: 64 arcs between positions, one lattice per wavefront
)


def mixed_wfst(regions=MIX_REGIONS, n_sym=64, p_eps=0.1, seed=9, entries=4):
    """One transducer whose corpus mixes the lattice classes (round-5 verdict: every benchmarked corpus was one class).  The
    start state (a cluster of its own) has `entries` moves into every REGION; a region is clustered_wfst's construction with its
    own cluster size -- 1: single paths (config 4), 3: c4a's ambiguous lattices, 8: `long`'s wide ones -- and walks stay inside
    their region.  States: the start, then the regions' clusters in order, then the final state; arcs state-major, moves in
    order, destination member innermost, move 0 into the final state (one arc per member, as clustered_wfst).  Returns
    (Wfst, layout) -- layout is what mixed_walk_corpus needs to follow moves."""
    rng = np.random.default_rng(seed)
    first_state = [1]
    for ncl, M, _mv, _k, _w, _p in regions:
        first_state.append(first_state[-1] + ncl * M)
    F = first_state[-1]
    n_states = F + 1
    srcs, dsts, isyms, osyms, gs = [], [], [], [], []

    def labels(shape, k):
        lab = rng.integers(FIRST_SYM, FIRST_SYM + k - 1, size=shape, dtype=np.uint32)
        lab[rng.random(shape) < p_eps] = 0
        return lab

    # the start state: move 0 into the final state, then `entries` moves into every region (to every member of the cluster entered)
    start_moves = [(F, 1)]
    for r, (ncl, M, _mv, _k, _w, _p) in enumerate(regions):
        for c in rng.integers(0, ncl, size=entries):
            start_moves.append((first_state[r] + int(c) * M, M))
    s_in = labels(len(start_moves), n_sym)
    s_out = rng.permutation(n_sym - 1)[:len(start_moves)].astype(np.uint32) + FIRST_SYM  # (distinct: the first move names the region)
    start_first_arc = []
    n0 = 0
    for m, (d0, M) in enumerate(start_moves):
        start_first_arc.append(n0)
        srcs.append(np.zeros(M, dtype=np.uint32))
        dsts.append((d0 + np.arange(M)).astype(np.uint32))
        isyms.append(np.full(M, s_in[m], dtype=np.uint32))
        osyms.append(np.full(M, s_out[m], dtype=np.uint32))
        n0 += M
    gs.append(rng.exponential(size=n0))
    arc_base = [n0]
    for r, (ncl, M, moves, k_in, _w, _p) in enumerate(regions):
        s0 = first_state[r]
        deg = moves * M
        ns = ncl * M
        dcl = rng.integers(0, ncl, size=(ncl, moves), dtype=np.int64)
        li, lo = labels((ncl, moves), k_in), labels((ncl, moves), n_sym)
        cl = np.repeat(np.arange(ncl, dtype=np.int64), M * deg)
        mv = np.tile(np.repeat(np.arange(moves, dtype=np.int64), M), ns)
        dm = np.tile(np.arange(M, dtype=np.int64), ns * moves)
        dst = (s0 + dcl[cl, mv] * M + dm).astype(np.uint32)
        dst[mv == 0] = F
        srcs.append(np.repeat(np.arange(s0, s0 + ns, dtype=np.uint32), deg))
        dsts.append(dst)
        isyms.append(li[cl, mv])
        osyms.append(lo[cl, mv])
        gs.append(rng.exponential(size=ns * deg))
        arc_base.append(arc_base[-1] + ns * deg)
    src, dst, isym, osym, g = (np.concatenate(x) for x in (srcs, dsts, isyms, osyms, gs))
    key = src.astype(np.uint64) * np.uint64(1 << 20) + isym.astype(np.uint64)
    _, inv = np.unique(key, return_inverse=True)
    sums = np.bincount(inv, weights=g)
    w = Wfst(n_states, F, src, dst, isym, osym, np.log(g / sums[inv]))
    layout = dict(regions=regions, first_state=first_state, arc_base=arc_base, start_moves=start_moves, start_first_arc=start_first_arc,
                  entries=entries)
    return w, layout


def mixed_walk_corpus(w, layout, n_pairs, seed=9):
    """pairs read off walks over mixed_wfst: the first move enters a region (drawn by the regions' shares of the pairs), the walk
    stays there for its length, the last move is move 0 into the final state.  Pairs of the regions are interleaved in corpus
    order (drawn independently), as a real corpus mixes short and long sentences."""
    rng = np.random.default_rng(seed + 1000003)
    regions = layout["regions"]
    shares = np.array([r[5] for r in regions], dtype=np.float64)
    reg = rng.choice(len(regions), size=n_pairs, p=shares / shares.sum())
    lo = np.array([r[4][0] for r in regions])[reg]
    hi = np.array([r[4][1] for r in regions])[reg]
    L = rng.integers(lo, hi + 1)  # moves inside the region (incl. the last one)
    M = np.array([r[1] for r in regions], dtype=np.int64)[reg]
    moves = np.array([r[2] for r in regions], dtype=np.int64)[reg]
    s0 = np.array(layout["first_state"][:-1], dtype=np.int64)[reg]
    a0 = np.array(layout["arc_base"][:-1], dtype=np.int64)[reg]
    ent = rng.integers(0, layout["entries"], size=n_pairs)
    sm = 1 + reg * layout["entries"] + ent  # the start state's move
    first_arc = np.array(layout["start_first_arc"], dtype=np.int64)[sm]
    maxL = int(L.max()) + 1
    ins = np.zeros((n_pairs, maxL), dtype=np.uint32)
    outs = np.zeros((n_pairs, maxL), dtype=np.uint32)
    ins[:, 0], outs[:, 0] = w.isym[first_arc], w.osym[first_arc]
    cur = w.dst[first_arc].astype(np.int64)  # member 0 of the cluster entered
    for step in range(maxL - 1):
        active = step < L
        last = step == L - 1
        pick = np.minimum(1 + (rng.random(n_pairs) * (moves - 1)).astype(np.int64), moves - 1)
        pick[last] = 0
        arc = a0 + (cur - s0) * (moves * M) + pick * M
        arc[~active] = 0
        ins[:, step + 1] = np.where(active, w.isym[arc], 0)
        outs[:, step + 1] = np.where(active, w.osym[arc], 0)
        cur = np.where(active & ~last, w.dst[arc].astype(np.int64), cur)
    mi, mo = ins != 0, outs != 0
    in_off = np.concatenate([[0], np.cumsum(mi.sum(1))]).astype(np.uint64)
    out_off = np.concatenate([[0], np.cumsum(mo.sum(1))]).astype(np.uint64)
    return Corpus(in_off, ins[mi], out_off, outs[mo]), reg


CONFIGS = {
    # name: (n_states, out_degree, n_pairs, seed)
    "toy": (200, 6, 300, 7),
    "c2": (100000, 20, 50000, 1),    # BASELINE.json configs[1]: 100k states / 2M arcs, 50k pairs
    "c4": (1000000, 10, 1000000, 3),  # configs[3]: 1M states / 10M arcs, 1M pairs
}

# the clustered (ambiguous) workloads next to them: name -> (n_states, out_degree, members, n_pairs, seed, min_arcs, max_arcs)
CLUSTERED = {
    # config 4's sizes with lattices that exercise the log-semiring sum: 10^6 states / 12*10^6 arcs / 10^6 pairs, three
    # members per cluster => a lattice is positions x 3 states, 9 arcs between neighbouring positions, 3 in-arcs per state
    "c4a": (999999 + 1, 12, 3, 1000000, 5, 5, 40),
    # few LONG lattices (320 .. 4800 states, 64 arcs per position): the large-lattice path, one lattice per wavefront
    "long": (8 * 12500 + 1, 32, 8, 5000, 6, 40, 600),
    "toya": (3 * 60 + 1, 12, 3, 300, 8, 3, 14),
}


def make_config(name, n_pairs=None, rank=0, walk=None):
    """(transducer, corpus) of a named synthetic workload; rank r > 0 draws another shard of the same size (other walks over
    the same transducer), walk = (min_arcs, max_arcs) overrides the walk lengths (experiments)"""
    if name in ("mix", "toymix"):
        # a corpus of all three lattice classes over one transducer (mixed_wfst): 500 000 pairs, 90 / 9 / 1 % by class (5 000 long
        # lattices, as many as `long` has: fewer leave the chip's SIMDs with a wavefront or two each -- 200 000 pairs: 16.5 % against 18 %)
        regions = MIX_REGIONS if name == "mix" else ((300, 1, 6, 16, (3, 12), 0.7), (90, 3, 4, 4, (3, 12), 0.2), (24, 8, 4, 4, (8, 30), 0.1))
        w, layout = mixed_wfst(regions, seed=9)
        c, _reg = mixed_walk_corpus(w, layout, n_pairs or (500000 if name == "mix" else 400), seed=9 + 7919 * rank)
        return w, c
    if name in CLUSTERED:
        n_states, deg, members, npairs, seed, lo, hi = CLUSTERED[name]
        if n_pairs is not None:
            npairs = n_pairs
        if walk:
            lo, hi = walk
        w = clustered_wfst(n_states, deg, members=members, seed=seed)
        c = clustered_walk_corpus(w, npairs, deg, members=members, min_arcs=lo, max_arcs=hi, seed=seed + 7919 * rank)
        return w, c
    n_states, deg, npairs, seed = CONFIGS[name]
    if n_pairs is not None:
        npairs = n_pairs
    lo, hi = walk or (5, 40)
    w = random_wfst(n_states, deg, seed=seed)
    c = random_walk_corpus(w, npairs, min_arcs=lo, max_arcs=hi, seed=seed + 7919 * rank, out_degree=deg)
    return w, c


def random_forests(n_forests, n_rules=500000, mean_nodes=55, p_backref=0.6, group_mean=8, seed=4):
    """packed AND/OR forests as preorder node arrays (label, ref, next), OR fan-out
    2-4, ~20 % of the children of AND nodes are back-references to earlier shared sub-forests, rule ids Zipf over
    n_rules parameters, normalisation groups of mean size group_mean.  Returns (node_off, label, ref, next,
    n_rules + 1, group_off, group_rule).  mean_nodes = 55 gives 50.6 nodes a forest: 100 000 forests are the 5 * 10^6 nodes
    BASELINE.json's configs[4] names (46, until round 5, gave 4.5 * 10^6)."""
    rng = np.random.default_rng(seed)
    zipf_p = 1.0 / np.arange(1, n_rules + 1)
    zipf_p /= zipf_p.sum()
    labels, refs, nexts, node_off = [], [], [], [0]
    rule_pool = rng.choice(n_rules, size=n_forests * mean_nodes * 4 + 4096, p=zipf_p) + 1
    rp = 0
    u = rng.random(n_forests * mean_nodes * 16 + 16384)
    up = 0
    for _ in range(n_forests):
        lab, ref, nxt = [], [], []
        shared = []
        budget = [int(mean_nodes * (0.5 + u[up % len(u)]))]
        up += 1

        def node(depth, kind):
            nonlocal rp, up
            i = len(lab)
            lab.append(0)
            ref.append(-1)
            nxt.append(0)
            budget[0] -= 1
            if kind == 0:  # OR: 2-4 AND children
                for _k in range(2 + int(u[up % len(u)] * 3)):
                    node(depth + 1, 1)
                up += 1
            else:
                lab[i] = int(rule_pool[rp % len(rule_pool)])
                rp += 1
                nk = 0 if (budget[0] <= 0 or depth > 8) else int(u[up % len(u)] * 3.2)
                up += 1
                for _k in range(nk):
                    r = u[up % len(u)]
                    up += 1
                    if shared and r < p_backref:
                        j = len(lab)
                        lab.append(0)
                        ref.append(shared[int(u[up % len(u)] * len(shared))])
                        up += 1
                        nxt.append(j + 1)
                    else:
                        c = len(lab)
                        node(depth + 1, 0 if r > 0.6 else 1)
                        if r > 0.7:
                            shared.append(c)
            nxt[i] = len(lab)
        node(0, 0)
        labels.extend(lab)
        refs.extend(ref)
        nexts.extend(nxt)
        node_off.append(len(labels))
    perm = rng.permutation(n_rules) + 1
    sizes = rng.poisson(group_mean - 1, size=n_rules // group_mean + 8) + 1
    goff = np.concatenate([[0], np.cumsum(sizes)])
    goff = goff[goff <= n_rules]
    if goff[-1] != n_rules:
        goff = np.append(goff, n_rules)
    return (np.asarray(node_off, np.uint64), np.asarray(labels, np.uint32), np.asarray(refs, np.int32),
            np.asarray(nexts, np.uint32), n_rules + 1, goff.astype(np.uint64), perm.astype(np.uint32))


def forests_to_text(node_off, label, ref, nxt, f0, f1):
    """forests [f0, f1) of the node arrays random_forests returns, in forest-em's text format (one forest per line:
    `(OR a b)`, `(rule child ...)`, `#k(...)` defines and `#k` references a shared sub-forest; forest.hpp:39-46)"""
    import sys
    sys.setrecursionlimit(max(10000, sys.getrecursionlimit()))
    lines = []
    for f in range(f0, f1):
        b, e = int(node_off[f]), int(node_off[f + 1])
        lab, rf, nx = label[b:e], ref[b:e], nxt[b:e]
        ids = {}
        for i in range(e - b):
            if rf[i] >= 0 and int(rf[i]) not in ids:
                ids[int(rf[i])] = len(ids) + 1

        def emit(i):
            if rf[i] >= 0:
                return "#%d" % ids[int(rf[i])]
            kids, j = [], i + 1
            while j < int(nx[i]):
                kids.append(emit(j))
                j = int(nx[j])
            head = "OR" if lab[i] == 0 else str(int(lab[i]))
            body = "(%s %s)" % (head, " ".join(kids)) if kids else head
            return ("#%d%s" % (ids[i], body)) if i in ids else body
        lines.append(emit(0))
    return "\n".join(lines) + "\n"


CIPHER_PLAIN = ["_"] + [chr(ord("A") + i) for i in range(26)]


def cipher_files(n_lines, min_len=30, max_len=80, seed=2):
    """SURVEY.md section 8d config 3, in the reference's own file formats (carmel/sample/decipher, tutorial cipher.*):
    a character bigram language model as a locked acceptor over *e*:"C" arcs (START + 27 states, the word separator
    "_" can stop: `_ -> END`), a 1-state 27x27 substitution channel "C":"c" with uniform weights, and a corpus of
    n_lines pairs (blank input line, min_len..max_len quoted cipher symbols).  Plain text is drawn from the bigram
    model itself and enciphered with a random permutation.  Returns (lm_text, channel_text, corpus_text)."""
    rng = np.random.default_rng(seed)
    n = len(CIPHER_PLAIN)
    big = rng.dirichlet(np.full(n, 0.3), size=n)       # P(next | prev), peaked like letter bigrams
    start = rng.dirichlet(np.full(n, 0.5))
    p_end = 0.1                                         # P(END | "_")
    lm = ["END"]
    for c in range(n):
        lm.append('(START (%s *e* "%s" %.15g!))' % (CIPHER_PLAIN[c], CIPHER_PLAIN[c], start[c]))
    for a in range(n):
        scale = (1.0 - p_end) if a == 0 else 1.0
        for c in range(n):
            lm.append('(%s (%s *e* "%s" %.15g!))' % (CIPHER_PLAIN[a], CIPHER_PLAIN[c], CIPHER_PLAIN[c], big[a, c] * scale))
    lm.append("(_ (END *e* *e* %.15g!))" % p_end)
    cipher_syms = [s.lower() if s != "_" else "_" for s in CIPHER_PLAIN]
    ch = ["0"]
    for a in range(n):
        for c in range(n):
            ch.append('(0 (0 "%s" "%s"))' % (CIPHER_PLAIN[a], cipher_syms[c]))
    perm = rng.permutation(n - 1) + 1                  # "_" maps to itself, letters are permuted
    key = np.concatenate([[0], perm])
    lens = rng.integers(min_len, max_len + 1, size=n_lines)
    cdf = np.cumsum(big, axis=1)
    # every line draws len uniforms (the last two go unused) and walks the bigram chain from the separator; the walks of
    # all lines advance together, one position per step (same draws, same text as a line-by-line loop)
    u_all = rng.random(int(lens.sum()))
    off = np.concatenate([[0], np.cumsum(lens)])[:-1]
    maxL = int(lens.max())
    sym = np.zeros((n_lines, maxL), dtype=np.int64)  # plain symbol per position; position 0 and len - 1 are the separator
    cur = np.zeros(n_lines, dtype=np.int64)
    for k in range(maxL - 2):
        act = np.nonzero(k < lens - 2)[0]
        if not len(act):
            break
        nxt = (cdf[cur[act]] < u_all[off[act] + k][:, None]).sum(axis=1)  # = searchsorted(cdf[cur], u), side "left"
        nxt = np.minimum(nxt, n - 1)
        cur[act] = nxt
        sym[act, k + 1] = nxt
    quoted = np.array(['"%s"' % cipher_syms[key[c]] for c in range(n)], dtype=object)
    lines = ["\n" + " ".join(quoted[sym[i, :lens[i]]]) for i in range(n_lines)]
    return "\n".join(lm) + "\n", "\n".join(ch) + "\n", "\n".join(lines) + "\n"
