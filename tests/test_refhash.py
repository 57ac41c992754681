"""CPU: the order in which carmel enumerates a CONDITIONAL transducer's normalisation groups -- a walk over a hash table per
state (fst.h:1362-1446 over State::index; graehl/shared/2hash.h) -- restated three times: tests/refhash_model.py (plain Python,
the referee here), oracle/refhash.hpp (through the oracle's --fem-norm text), csrc/host/refhash.hpp (the product: the GPU test
tests/test_cli_gpu.py::test_fem_norm_lists_the_groups_in_the_reference_order holds the front end against the oracle)."""
import re

import numpy as np
import pytest

import refhash_model as M


def test_model_table_life():
    """hand-checkable facts of 2hash.h: 4 buckets at least, a power of two >= the requested size; growth when the entry count
    reaches (unsigned)(0.9f * buckets), after which growAt = 2 * growAt + 1; every key exactly once in the walk"""
    t = M.RefHashTable(3)
    assert len(t.table) == 4 and t.grow_at == 3
    t = M.RefHashTable(8)
    assert len(t.table) == 8 and t.grow_at == 7
    for k in range(6):
        t.insert(k)
    assert len(t.table) == 8
    t.insert(6)  # the 7th entry: the table doubles before it goes in
    assert len(t.table) == 16 and t.grow_at == 15 and sorted(t.keys()) == list(range(7))
    assert M.RefHashTable(9).mask + 1 == 16 and M.RefHashTable(1000).mask + 1 == 1024
    # the hash: a * golden ratio, high half folded into the low half
    assert M.uint32_hash(1) == (2654435769 ^ (2654435769 >> 16)) and M.uint32_hash(0) == 0
    # a repeated key neither grows the table nor moves
    t = M.RefHashTable(4)
    for k in (5, 5, 5, 9, 5):
        t.insert(k)
    assert sorted(t.keys()) == [5, 9] and t.cnt == 2


def _fst_text(rng, n_states, n_sym, max_arcs):
    """a random transducer in carmel's text form whose states have 0 .. max_arcs arcs over input symbols s1 .. s<n_sym>; returns
    the text and, per state, the list of input-symbol ids of its arcs in file order (*e* = 0, *w* = 1, then first seen)"""
    ids, per_state, lines = {}, [], []
    names = ["q%d" % i for i in range(n_states)]
    lines.append(names[-1])
    for s in range(n_states):
        arcs = []
        for k in range(int(rng.integers(1, max_arcs + 1)) if s + 1 < n_states else int(rng.integers(0, 3))):
            sym = "s%d" % int(rng.integers(0, n_sym))
            d = names[s + 1] if (k == 0 and s + 1 < n_states) else names[int(rng.integers(0, n_states))]  # (every state on a path)
            lines.append('(%s (%s "%s" "%s" %.3f))' % (names[s], d, sym, sym, float(rng.uniform(0.1, 1.0))))
            arcs.append(sym)
        per_state.append(arcs)
    return "\n".join(lines) + "\n", per_state


@pytest.mark.parametrize("seed", range(8))
def test_oracle_conditional_group_order_is_the_models(oracle, seed):
    """the oracle's --fem-norm text (cascade.h:85-116 over NormGroupIter) on random transducers: line for line the groups the
    Python model of the hash walk gives -- states in order, a state's input symbols in bucket order, a symbol's arcs newest first"""
    rng = np.random.default_rng(100 + seed)
    n_states = int(rng.integers(2, 9))
    text, per_state = _fst_text(rng, n_states, n_sym=int(rng.integers(2, 40)), max_arcs=int(rng.choice([3, 7, 8, 9, 20, 70])))
    if not any(per_state):
        pytest.skip("no arcs drawn")
    oc = oracle.OracleCascade([text])
    oc.composed()
    got = oracle.fem_export(oc, oc.corpus(""), 1, "C", [0.0])
    # the transducer as the oracle holds it: arcs in arc-id order = state-major list order (arc ids of --fem-norm are 1-based)
    a = oc.composed().arrays()
    want, base = [], 1
    for s in range(int(a["n_states"])):
        arcs = [int(x) for x in a["isym"][a["src"] == s]]
        for grp in M.conditional_groups(arcs):
            want.append("(" + "".join(" %d" % (base + j) for j in grp) + " )")
        base += len(arcs)
    assert base - 1 == len(a["src"]) and len(want) >= 1
    assert got == "(\n" + "".join(w + "\n" for w in want) + ")\n"
    # the walk is not the first-seen order (the oracle's former numbering) once a state has a few symbols
    if seed == 0:
        first_seen = []
        for s in range(int(a["n_states"])):
            seen = []
            for x in a["isym"][a["src"] == s]:
                if int(x) not in seen:
                    seen.append(int(x))
            first_seen.append(seen)
        walk = [[int(a["isym"][a["src"] == s][g[0]]) for g in M.conditional_groups([int(x) for x in a["isym"][a["src"] == s]])]
                for s in range(int(a["n_states"]))]
        assert any(w != f for w, f in zip(walk, first_seen))


def test_oracle_joint_groups_include_states_without_arcs(oracle):
    """NormGroupIter under JOINT visits every state (fst.h:1439-1441): a state without arcs prints an empty group"""
    text = 'F\n(S (A "a" "x" 0.5))\n(S (F "b" "y" 0.5))\n(A (F "a" "x" 1))\n'
    oc = oracle.OracleCascade([text])
    oc.composed()
    assert oracle.fem_export(oc, oc.corpus(""), 1, "J", [0.0]) == "(\n( 1 2 )\n( 3 )\n( )\n)\n"  # S, A, then the arc-less F
