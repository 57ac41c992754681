"""Test helpers: the host-side lattice image as numpy arrays, and a plain numpy sweep over that image (used on
machines without a GPU to check the LAYOUT the kernels consume against the oracle's E-step)."""
import ctypes as C

import numpy as np

from carmel_amd._capi import lib, ptr

BUNDLE_DTYPE = np.dtype([("in_base", "<u8"), ("out_base", "<u8"), ("off_base", "<u8"), ("n_states", "<u4"),
                         ("n_levels", "<u4"), ("level_base", "<u4"), ("pair_base", "<u4"), ("n_pairs", "<u4"),
                         ("flags", "<u4"), ("n_arcs", "<u8"), ("pad", "<u8")])
assert BUNDLE_DTYPE.itemsize == 64


TRANS_TILE = 16384
TRANS_BUCKET_DTYPE = np.dtype([("item_base", "<u8"), ("n_items", "<u4"), ("arc_lo", "<u4"), ("n_arcs", "<u4"),
                               ("flags", "<u4")])
LANE_DTYPE = np.dtype([("stream_base", "<u8"), ("maxlen", "<u4"), ("n_lanes", "<u4"), ("pair_base", "<u4"),
                       ("max_states", "<u4"), ("window", "<u4"), ("spill_row", "<u4")])
assert LANE_DTYPE.itemsize == 32
LANE_LAST, LANE_VALID = 0x80000000, 0x40000000
WAVE_DTYPE = np.dtype([("fwd_base", "<u8"), ("bwd_base", "<u8"), ("n_states", "<u4"), ("n_levels", "<u4"),
                       ("level_base", "<u4"), ("pair", "<u4"), ("max_width", "<u4"), ("ring", "<u4"), ("logw", "<f8"),
                       ("n_arcs", "<u8"), ("spill_base", "<u8")])
assert WAVE_DTYPE.itemsize == 64
WAVE_VALID = 0x80000000


def host_lattices(w, c, prune=True, threads=2, small_pairs=0, small_states=0, lane_states=-1):
    h = C.c_void_p()
    rc = lib.carmel_hip_host_build(C.byref(h), w.n_states, w.final, w.n_arcs, ptr(w.src), ptr(w.dst), ptr(w.isym),
                                   ptr(w.osym), c.n_pairs, ptr(c.in_off), ptr(c.in_sym), ptr(c.out_off),
                                   ptr(c.out_sym), ptr(c.weight), int(prune), threads, small_pairs, small_states,
                                   lane_states)
    assert rc == 0
    dims = np.zeros(19, np.uint64)
    lib.carmel_hip_host_dims(h, ptr(dims))
    nb, noff, na, nlev, npair, ncls = (int(x) for x in dims[:6])
    out = dict(bundles=np.zeros(nb, BUNDLE_DTYPE), in_arcs=np.zeros((na, 2), np.uint32),
               out_arcs=np.zeros((na, 2), np.uint32), in_off=np.zeros(noff, np.uint32),
               out_off=np.zeros(noff, np.uint32), level_off=np.zeros(nlev, np.uint32),
               pair_start=np.zeros(npair, np.uint32), pair_final=np.zeros(npair, np.uint32),
               pair_id=np.zeros(npair, np.uint32), pair_logw=np.zeros(npair), classes=np.zeros((ncls, 5), np.uint32),
               has_deriv=np.zeros(c.n_pairs, np.uint8))
    lib.carmel_hip_host_export(h, ptr(out["bundles"]), ptr(out["in_arcs"]), ptr(out["out_arcs"]), ptr(out["in_off"]),
                               ptr(out["out_off"]), ptr(out["level_off"]), ptr(out["pair_start"]),
                               ptr(out["pair_final"]), ptr(out["pair_id"]), ptr(out["pair_logw"]),
                               ptr(out["classes"]), ptr(out["has_deriv"]))
    ng, nrec, nslot, nlc = (int(x) for x in dims[10:14])
    out.update(lane_groups=np.zeros(ng, LANE_DTYPE), lane_fwd=np.zeros((nrec, 2), np.uint32),
               lane_bwd=np.zeros((nrec, 2), np.uint32), lane_pair=np.zeros(nslot, np.uint32),
               lane_nstates=np.zeros(nslot, np.uint32), lane_logw=np.zeros(nslot),
               lane_classes=np.zeros((nlc, 3), np.uint32), total_states=int(dims[14]), total_arcs=int(dims[15]),
               explored_states=int(dims[8]), explored_arcs=int(dims[9]), last_pair=tuple(int(x) for x in dims[16:19]))
    lib.carmel_hip_host_export_lanes(h, ptr(out["lane_groups"]), ptr(out["lane_fwd"]), ptr(out["lane_bwd"]),
                                     ptr(out["lane_pair"]), ptr(out["lane_nstates"]), ptr(out["lane_logw"]),
                                     ptr(out["lane_classes"]))
    td = np.zeros(6, np.uint64)
    null = C.c_void_p(None)
    lib.carmel_hip_host_transpose(h, ptr(td), *([null] * 10))
    ni, nbk, nt, nsp, npost, narc = (int(x) for x in td)
    tr = dict(n_items=ni, n_post=npost, n_arcs=narc, buckets=np.zeros(nbk, TRANS_BUCKET_DTYPE),
              tile_base=np.zeros(nt + 1 if nbk else 0, np.uint64), b_arc=np.zeros(ni if nbk else 0, np.uint16),
              b_rank=np.zeros(ni if nbk else 0, np.uint16), b_src=np.zeros(ni if nbk else 0, np.uint32),
              t_pos=np.zeros(ni if nbk else 0, np.uint16), t_src=np.zeros(ni if nbk else 0, np.uint32),
              split_arcs=np.zeros(nsp, np.uint32), arc_off=np.zeros(narc + 1, np.uint64),
              slot_pos=np.zeros(ni, np.uint64))
    lib.carmel_hip_host_transpose(h, null, ptr(tr["buckets"]), ptr(tr["tile_base"]), ptr(tr["b_arc"]),
                                  ptr(tr["b_rank"]), ptr(tr["b_src"]), ptr(tr["t_pos"]), ptr(tr["t_src"]),
                                  ptr(tr["split_arcs"]), ptr(tr["arc_off"]), ptr(tr["slot_pos"]))
    ti = np.zeros(2, np.uint32)
    lib.carmel_hip_host_tile_sweep(h, ptr(ti), null)
    tr["tile"] = int(ti[0])
    tr["tile_group"] = np.zeros(int(ti[1]), np.uint32)
    if ti[1]:
        lib.carmel_hip_host_tile_sweep(h, null, ptr(tr["tile_group"]))
    out["transpose"] = tr
    wd = np.zeros(6, np.uint64)
    lib.carmel_hip_host_export_waves(h, ptr(wd), *([null] * 8))
    nw, nfw, nbw, nlv, nwc, wbase = (int(x) for x in wd)
    wv = dict(descs=np.zeros(nw, WAVE_DTYPE), fwd=np.zeros((nfw, 2), np.uint32), bwd=np.zeros(nbw, np.uint32),
              bwd_arc=np.zeros(nbw, np.uint32), level_off=np.zeros(nlv, np.uint32), frow=np.zeros(nlv, np.uint32),
              brow=np.zeros(nlv, np.uint32), classes=np.zeros((nwc, 4), np.uint32), slot_base=wbase)
    lib.carmel_hip_host_export_waves(h, null, ptr(wv["descs"]), ptr(wv["fwd"]), ptr(wv["bwd"]), ptr(wv["bwd_arc"]),
                                     ptr(wv["level_off"]), ptr(wv["frow"]), ptr(wv["brow"]), ptr(wv["classes"]))
    out["waves"] = wv
    lib.carmel_hip_host_free(h)
    out.update(n_kept=int(dims[6]), n_cyclic=int(dims[7]), explored_states=int(dims[8]), explored_arcs=int(dims[9]))
    return out


def _lse(xs):
    xs = np.asarray(xs, dtype=np.float64)
    if len(xs) == 0:
        return -np.inf
    m = xs.max()
    if m == -np.inf:
        return -np.inf
    return m + np.log(np.exp(xs - m).sum())


def _lwadd(a, b):
    if a == -np.inf:
        return b
    if b == -np.inf:
        return a
    d = a - b
    if d > 36:
        return a
    if d < -36:
        return b
    return (b + np.log1p(np.exp(d))) if d < 0 else (a + np.log1p(np.exp(-d)))


def alpha_s(col, s):
    return col[s]


def numpy_sweep(img, logw, n_pairs_total):
    """forward / backward / counts over the bundle image exactly as kernels.hip walks it.
    Returns (counts linear per WFST arc, per-pair ln prob)."""
    counts = np.zeros(len(logw))
    plp = np.full(n_pairs_total, -np.inf)
    # lane groups: the per-lane record streams exactly as sweep_lane_kernel consumes them
    for g in img.get("lane_groups", []):
        base, ml = int(g["stream_base"]), int(g["maxlen"])
        # windowed groups (LaneGroup::window): state s lives in ring row s mod window -- stale rows are POISONED here, so
        # an arc that reaches outside the ring shows up as a NaN -- and the forward values are parked in `spill`
        win = int(g["window"])
        for l in range(int(g["n_lanes"])):
            slot = int(g["pair_base"]) + l
            S = int(img["lane_nstates"][slot])
            rows = win if win else S
            if win:
                assert win & (win - 1) == 0 and int(g["max_states"]) == win
            wm = rows - 1 if win else 0xffffffff
            col = np.full(rows, np.nan)
            spill = np.full(S, np.nan)
            col[0] = spill[0] = 0.0
            d, terms, wcache = 1, [], {}
            for k in range(ml):
                x, arc = img["lane_fwd"][base + k * 64 + l]
                if not x & LANE_VALID:
                    continue
                src = int(x & 0x3ff)
                assert src < d and (not win or d - src < win)
                terms.append(col[src & wm] + logw[arc])
                bpos = (x >> 10) & 0xfffff  # the forward record points at the arc's backward position (wcache slot)
                assert bpos not in wcache
                wcache[bpos] = (arc, logw[arc])
                if x & LANE_LAST:
                    col[d & wm] = spill[d] = _lse(terms)
                    d, terms = d + 1, []
            assert d == S and not terms
            lp = col[(S - 1) & wm]
            plp[img["lane_pair"][slot]] = lp
            col[(S - 1) & wm] = img["lane_logw"][slot] - lp
            s, terms = S - 2, []
            for k in range(ml):
                x, arc = img["lane_bwd"][base + k * 64 + l]
                if not x & LANE_VALID:
                    continue
                dst = int(x & 0x3ff)
                assert dst > s and ((x >> 10) & 0x3ff) == s  # the backward record carries its source state
                assert wcache[k][0] == arc  # the forward pass left this arc's weight at exactly this position
                t = wcache[k][1] + col[dst & wm]
                counts[arc] += np.exp(spill[(x >> 10) & 0x3ff] + t)
                terms.append(t)
                if x & LANE_LAST:
                    col[s & wm] = _lse(terms)
                    s, terms = s - 1, []
            assert s == -1 and not terms
    # one-per-wavefront lattices: rows of 64 records per level, exactly as sweep_wave_kernel consumes them (WaveDesc)
    wv = img.get("waves")
    for d in (wv["descs"] if wv is not None else []):
        S, NL, lb = int(d["n_states"]), int(d["n_levels"]), int(d["level_base"])
        lvl, frow, brow = (wv[k][lb:lb + NL + 1].astype(np.int64) for k in ("level_off", "frow", "brow"))
        fb, bb = int(d["fwd_base"]), int(d["bwd_base"])
        assert lvl[0] == 0 and lvl[1] == 1 and lvl[NL] == S and lvl[NL] - lvl[NL - 1] == 1  # start and goal alone on their levels
        assert int(np.diff(lvl).max()) == int(d["max_width"])
        # ring form (WaveDesc::ring): state s lives in slot s mod ring -- stale slots are what they are, so an arc reaching
        # further back than the ring would read another state's value and the comparison with the oracle would fail; the
        # forward values are parked in `spill` for the posteriors
        ring = int(d["ring"])
        if ring:
            assert ring & (ring - 1) == 0 and int(d["max_width"]) <= 64 and ring < S

        class Ring(object):
            def __init__(self):
                self.a = np.full(ring if ring else S, np.nan)
                self.owner = np.full(ring if ring else S, -1)

            def __getitem__(self, s):
                k = s % ring if ring else s
                assert self.owner[k] == s, "ring slot holds another state's value"
                return self.a[k]

            def __setitem__(self, s, v):
                k = s % ring if ring else s
                self.a[k], self.owner[k] = v, s
        val = Ring()
        spill = np.full(S, np.nan)
        val[0] = spill[0] = 0.0
        n_valid = 0
        for l in range(1, NL):
            assert frow[l + 1] > frow[l]
            terms = {}
            for r in range(frow[l], frow[l + 1]):
                for x, y in wv["fwd"][fb + r * 64:fb + r * 64 + 64]:
                    if not x & WAVE_VALID:
                        continue
                    src, dr = int(x & 0xffff), int((x >> 16) & 0x3fff)
                    assert src < lvl[l] and dr < lvl[l + 1] - lvl[l], "in-arc source must lie in an earlier level"
                    arc = int(wv["bwd_arc"][bb + int(y)])  # the forward record points at the arc's backward position
                    assert arc != 0xffffffff and wv["bwd"][bb + int(y)] & WAVE_VALID
                    assert int(wv["bwd"][bb + int(y)] & 0xffff) == lvl[l] + dr  # ... whose record names the same destination
                    terms.setdefault(dr, []).append(val[src] + logw[arc])
                    n_valid += 1
            assert sorted(terms) == list(range(lvl[l + 1] - lvl[l]))  # every state of the level has an in-arc
            for dr, ts in terms.items():
                val[lvl[l] + dr] = spill[lvl[l] + dr] = _lse(ts)
        assert n_valid == int(d["n_arcs"])
        lp = val[S - 1]
        plp[int(d["pair"])] = lp
        val[S - 1] = float(d["logw"]) - lp
        n_valid = 0
        for k in range(1, NL):
            l = NL - 1 - k
            assert brow[k + 1] > brow[k]
            terms = {}
            for r in range(brow[k], brow[k + 1]):
                for j in range(64):
                    pos = bb + r * 64 + j
                    x = int(wv["bwd"][pos])
                    if not x & WAVE_VALID:
                        assert wv["bwd_arc"][pos] == 0xffffffff
                        continue
                    dst, sr = x & 0xffff, (x >> 16) & 0x3fff
                    assert dst >= lvl[l + 1] and sr < lvl[l + 1] - lvl[l], "out-arc destination must lie in a later level"
                    arc = int(wv["bwd_arc"][pos])
                    t = logw[arc] + val[dst]
                    counts[arc] += np.exp(spill[lvl[l] + sr] + t)
                    terms.setdefault(sr, []).append(t)
                    n_valid += 1
            assert sorted(terms) == list(range(lvl[l + 1] - lvl[l]))
            for sr, ts in terms.items():
                val[lvl[l] + sr] = _lse(ts)
        assert n_valid == int(d["n_arcs"])
    for b in img["bundles"]:
        ns = int(b["n_states"])
        ob, ib, ab = int(b["off_base"]), int(b["in_base"]), int(b["out_base"])
        ioff = img["in_off"][ob:ob + ns + 1]
        ooff = img["out_off"][ob:ob + ns + 1]
        ia = img["in_arcs"][ib:ib + int(b["n_arcs"])]
        oa = img["out_arcs"][ab:ab + int(b["n_arcs"])]
        lv = img["level_off"][int(b["level_base"]):int(b["level_base"]) + int(b["n_levels"]) + 1]
        pb, npb = int(b["pair_base"]), int(b["n_pairs"])
        alpha = np.full(ns, -np.inf)
        beta = np.full(ns, -np.inf)
        if b["flags"] & 1:  # cyclic: the reference's in-order scatter sweeps
            st, fin = int(img["pair_start"][pb]), int(img["pair_final"][pb])
            alpha[st] = 0.0
            for s in range(ns):
                for a in range(ioff[0] * 0 + ooff[s], ooff[s + 1]):
                    d, arc = oa[a]
                    alpha[d] = _lwadd(alpha[d], alpha[s] + logw[arc])
            prob = alpha[fin]
            plp[img["pair_id"][pb]] = prob
            beta[fin] = 0.0
            for s in range(ns - 1, -1, -1):
                for a in range(ioff[s], ioff[s + 1]):
                    u, arc = ia[a]
                    beta[u] = _lwadd(beta[u], beta[s] + logw[arc])
            for s in range(ns):
                for a in range(ooff[s], ooff[s + 1]):
                    d, arc = oa[a]
                    counts[arc] += np.exp(logw[arc] + alpha[s] + beta[d] + img["pair_logw"][pb] - prob)
            continue
        for p in range(npb):
            alpha[img["pair_start"][pb + p]] = 0.0
        for l in range(1, int(b["n_levels"])):
            for s in range(lv[l], lv[l + 1]):
                r = ia[ioff[s]:ioff[s + 1]]
                assert np.all(r[:, 0] < lv[l]), "in-arc source must lie in an earlier level"
                alpha[s] = _lse(alpha[r[:, 0]] + logw[r[:, 1]])
        for p in range(npb):
            f = img["pair_final"][pb + p]
            plp[img["pair_id"][pb + p]] = alpha[f]
            beta[f] = img["pair_logw"][pb + p] - alpha[f]
        for l in range(int(b["n_levels"]) - 1, -1, -1):
            for s in range(lv[l], lv[l + 1]):
                r = oa[ooff[s]:ooff[s + 1]]
                if len(r) == 0:
                    continue
                assert np.all(r[:, 0] >= lv[l + 1]), "out-arc destination must lie in a later level"
                t = logw[r[:, 1]] + beta[r[:, 0]]
                beta[s] = _lse(t)
                np.add.at(counts, r[:, 1], np.exp(alpha[s] + t))
    return counts, plp


def transpose_weights(tr, logw, n_wcache):
    """numpy model of trans_w_bucket_kernel + trans_w_tile_kernel: arc-order weights -> wcache (position order)"""
    x = np.zeros(tr["n_items"])
    for b in tr["buckets"]:
        lo, n = int(b["item_base"]), int(b["n_items"])
        lds = logw[int(b["arc_lo"]):int(b["arc_lo"]) + int(b["n_arcs"])]
        x[lo:lo + n] = lds[tr["b_arc"][lo:lo + n]]
    wc = np.zeros(n_wcache)
    for t in range(len(tr["tile_base"]) - 1):
        T = tr["tile"]
        p0 = t * T
        if p0 >= n_wcache:
            break
        lds = np.zeros(T)
        i0, i1 = int(tr["tile_base"][t]), int(tr["tile_base"][t + 1])
        lds[tr["t_pos"][i0:i1]] = x[tr["t_src"][i0:i1]]
        n = min(T, n_wcache - p0)
        wc[p0:p0 + n] = lds[:n]
    return wc


def transpose_counts(tr, post):
    """numpy model of trans_c_tile_kernel + trans_c_bucket_kernel: posteriors (position order) -> per-arc sums"""
    x = np.zeros(tr["n_items"])
    for t in range(len(tr["tile_base"]) - 1):
        p0 = t * tr["tile"]
        lds = post[p0:p0 + tr["tile"]]
        i0, i1 = int(tr["tile_base"][t]), int(tr["tile_base"][t + 1])
        x[i0:i1] = lds[tr["t_pos"][i0:i1]]
    counts = np.zeros(tr["n_arcs"])
    for b in tr["buckets"]:
        lo, n = int(b["item_base"]), int(b["n_items"])
        lds = np.zeros(n)
        lds[tr["b_rank"][lo:lo + n]] = x[tr["b_src"][lo:lo + n]]
        if b["flags"] & 2:  # TRANS_SINGLE: a hub arc's own bucket(s); & 1 = one of several pieces
            counts[int(b["arc_lo"])] += lds.sum()
        else:
            for a in range(int(b["arc_lo"]), int(b["arc_lo"]) + int(b["n_arcs"])):
                counts[a] = lds[int(tr["arc_off"][a]) - lo:int(tr["arc_off"][a + 1]) - lo].sum()
    return counts


def set_hip_option(key, value):
    """one of the library's switches (carmel_hip_set_option; the former environment variable CARMEL_HIP_<KEY>); None unsets it"""
    import carmel_amd
    carmel_amd.set_option(key, None if value is None else str(value))


class hip_env(object):
    """switches named like the former environment variables ({"CARMEL_HIP_FOREST_MULTI": "0"}) as library options for the length of
    a with-block"""

    def __init__(self, env):
        self.env = {("timing" if k == "CARMEL_TIMING" else k[len("CARMEL_HIP_"):].lower()): v for k, v in env.items()}

    def __enter__(self):
        import carmel_amd
        self.saved = {k: carmel_amd.get_option(k) for k in self.env}
        for k, v in self.env.items():
            carmel_amd.set_option(k, v)
        return self

    def __exit__(self, *exc):
        import carmel_amd
        for k, v in self.saved.items():
            carmel_amd.set_option(k, v)
        return False
