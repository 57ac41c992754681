"""ORACLE — TEST INFRASTRUCTURE ONLY.  ctypes binding of oracle/liboracle.so (the CPU restatement of the
reference's algorithm).  Imported only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liboracle.so")


def build(force=False):
    if force or not os.path.exists(LIB_PATH):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so", "oracle_carmel"])


build()
lib = C.CDLL(LIB_PATH)
vp = C.c_void_p
lib.orc_last_error.restype = C.c_char_p
lib.orc_wfst_from_arrays.restype = vp
lib.orc_wfst_from_arrays.argtypes = [C.c_uint32, C.c_uint32, C.c_uint64, vp, vp, vp, vp, vp, vp]
lib.orc_wfst_parse.restype = vp
lib.orc_wfst_parse.argtypes = [C.c_char_p, C.c_int]
lib.orc_wfst_free.argtypes = [vp]
lib.orc_wfst_dims.argtypes = [vp, vp, vp, vp]
lib.orc_wfst_export.argtypes = [vp] + [vp] * 6
lib.orc_wfst_set_logw.argtypes = [vp, vp]
lib.orc_wfst_reduce.argtypes = [vp]
lib.orc_wfst_normalize.argtypes = [vp, C.c_int, C.c_double]
lib.orc_wfst_write.restype = vp
lib.orc_wfst_write.argtypes = [vp, C.c_int, C.c_int, C.c_int]
lib.orc_free_str.argtypes = [vp]
lib.orc_wfst_alphabet_size.restype = C.c_uint32
lib.orc_wfst_alphabet_size.argtypes = [vp, C.c_int]
lib.orc_corpus_from_arrays.restype = vp
lib.orc_corpus_from_arrays.argtypes = [C.c_uint64, vp, vp, vp, vp, vp]
lib.orc_corpus_parse.restype = vp
lib.orc_corpus_parse.argtypes = [vp, C.c_char_p]
lib.orc_corpus_free.argtypes = [vp]
lib.orc_corpus_dims.argtypes = [vp, vp, vp, vp]
lib.orc_corpus_export.argtypes = [vp] + [vp] * 5
lib.orc_estimate.argtypes = [vp, vp, C.c_int, vp, vp, vp, vp, vp, C.c_int]
lib.orc_lattice.argtypes = [vp, vp, C.c_uint64, C.c_int] + [vp] * 8
lib.orc_train.argtypes = [vp, vp, C.c_int, C.c_double, C.c_int, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int,
                          C.c_int, vp, C.c_int, vp, vp]
lib.orc_train_cascade_text.argtypes = [C.c_int, vp, C.c_char_p, C.c_char_p, vp, C.c_int, C.c_double, C.c_double,
                                       C.c_int, C.c_int, C.c_int, vp, vp, vp, C.c_int, vp, vp]


lib.orc_bench_em.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, vp]


def bench_em(w, c, norm_group=0, iters=3, threads=1):
    """CPU baseline: cached lattices, `iters` x (estimate + maximize).  Returns dict(sec_per_iter, lattice_arcs,
    build_sec, ln_prob)"""
    out = np.zeros(4)
    _chk(lib.orc_bench_em(w.h, c.h, norm_group, iters, threads, _p(out)))
    return dict(sec_per_iter=out[0], lattice_arcs=out[1], build_sec=out[2], ln_prob=out[3])


def bench_em_fit(w, c, norm_group=0, iters=2, threads=1, check=False):
    """CPU baseline with the fixed (O(|WFST arcs|): clear counts + maximize) and the per-lattice-arc cost of an EM
    iteration separated (orc_bench_em_fit): E-step timed on a quarter and on all of the sample's cached lattices,
    maximize on its own; serial, and -- threads > 1 -- with OpenMP.  Returns dict(build_sec, arcs_quarter, arcs_all,
    serial=dict(estep_quarter, estep_all, maximize, ln_prob, fixed_sec, sec_per_arc), threaded=... or None)"""
    out = np.zeros(11)
    counts_ln = pair_lp = None
    if check:
        counts_ln, pair_lp = np.zeros(w.dims()[1]), np.full(len(c.arrays()["weight"]), np.nan)
    _chk(lib.orc_bench_em_fit_check(w.h, c.h, norm_group, iters, threads, _p(out), _p(counts_ln), _p(pair_lp)))

    def leg(o):
        eq, ea, mx, lp = (float(v) for v in out[o:o + 4])
        b = (ea - eq) / max(out[2] - out[1], 1.0)
        return dict(estep_quarter=eq, estep_all=ea, maximize=mx, ln_prob=lp, sec_per_arc=b,
                    fixed_sec=max(eq - b * out[1], 0.0) + mx)
    return dict(build_sec=float(out[0]), arcs_quarter=float(out[1]), arcs_all=float(out[2]), serial=leg(3),
                threaded=leg(7) if threads > 1 else None, counts_ln=counts_ln, pair_logprob=pair_lp)


def _p(a):
    return None if a is None else a.ctypes.data_as(vp)


def _chk(rc):
    if rc != 0:
        raise RuntimeError("oracle: " + lib.orc_last_error().decode())


TRACE_FIELDS = ("iter", "log2_prob", "log2_ppx_symbol", "log2_ppx_example", "new_best", "rel_ppx_ratio_ln",
                "last_change", "n_example")


class OracleWfst(object):
    def __init__(self, handle):
        self.h = handle

    @classmethod
    def from_arrays(cls, w):
        """w: carmel_amd.model.Wfst (or anything with the same array attributes)"""
        return cls(lib.orc_wfst_from_arrays(w.n_states, w.final, w.n_arcs, _p(w.src), _p(w.dst), _p(w.isym),
                                            _p(w.osym), _p(w.logw), _p(w.group)))

    @classmethod
    def parse(cls, text, always_named=True):
        h = lib.orc_wfst_parse(text.encode(), 1 if always_named else 0)
        if not h:
            raise ValueError("oracle: bad WFST text")
        return cls(h)

    def __del__(self):
        if getattr(self, "h", None):
            lib.orc_wfst_free(self.h)
            self.h = None

    def dims(self):
        a, b, c = C.c_uint32(), C.c_uint64(), C.c_uint32()
        lib.orc_wfst_dims(self.h, C.byref(a), C.byref(b), C.byref(c))
        return a.value, b.value, c.value

    def arrays(self):
        ns, na, fin = self.dims()
        src, dst, i, o, g = (np.zeros(na, np.uint32) for _ in range(5))
        lw = np.zeros(na)
        lib.orc_wfst_export(self.h, _p(src), _p(dst), _p(i), _p(o), _p(lw), _p(g))
        return dict(n_states=ns, final=fin, src=src, dst=dst, isym=i, osym=o, logw=lw, group=g)

    def set_logw(self, lw):
        lw = np.ascontiguousarray(lw, dtype=np.float64)
        lib.orc_wfst_set_logw(self.h, _p(lw))

    def reduce(self):
        lib.orc_wfst_reduce(self.h)

    def normalize(self, group=0, add_count=0.0):
        lib.orc_wfst_normalize(self.h, group, add_count)

    def write(self, full=False, onearc=False, wmode=0):
        s = lib.orc_wfst_write(self.h, int(full), int(onearc), wmode)
        txt = C.string_at(s).decode()
        lib.orc_free_str(s)
        return txt

    def alphabet_size(self, output=False):
        return lib.orc_wfst_alphabet_size(self.h, int(output))


class OracleCorpus(object):
    def __init__(self, handle):
        self.h = handle

    @classmethod
    def from_arrays(cls, c):
        return cls(lib.orc_corpus_from_arrays(c.n_pairs, _p(c.in_off), _p(c.in_sym), _p(c.out_off), _p(c.out_sym),
                                              _p(c.weight)))

    @classmethod
    def parse(cls, wfst, text):
        return cls(lib.orc_corpus_parse(wfst.h, text.encode()))

    def __del__(self):
        if getattr(self, "h", None):
            lib.orc_corpus_free(self.h)
            self.h = None

    def arrays(self):
        n, a, b = C.c_uint64(), C.c_uint64(), C.c_uint64()
        lib.orc_corpus_dims(self.h, C.byref(n), C.byref(a), C.byref(b))
        io, oo = np.zeros(n.value + 1, np.uint64), np.zeros(n.value + 1, np.uint64)
        isym, osym = np.zeros(a.value, np.uint32), np.zeros(b.value, np.uint32)
        wt = np.zeros(n.value)
        lib.orc_corpus_export(self.h, _p(io), _p(isym), _p(oo), _p(osym), _p(wt))
        return dict(in_off=io, in_sym=isym, out_off=oo, out_sym=osym, weight=wt)


def estimate(w, c, prune=True, n_threads=1):
    """one E-step; returns dict(counts_ln, pair_logprob, has_deriv, stats, sum_logprob, sum_weighted_logprob)"""
    ns, na, _ = w.dims()
    npairs = len(c.arrays()["weight"])
    counts = np.zeros(na)
    pl = np.zeros(npairs)
    hd = np.zeros(npairs, np.uint8)
    stats = np.zeros(4)
    sums = np.zeros(2)
    _chk(lib.orc_estimate(w.h, c.h, int(prune), _p(counts), _p(pl), _p(hd), _p(stats), _p(sums), n_threads))
    return dict(counts_ln=counts, pair_logprob=pl, has_deriv=hd.astype(bool), stats=stats, sum_logprob=sums[0],
                sum_weighted_logprob=sums[1])


lib.orc_estimate_matrix.argtypes = [vp, vp, vp, vp, vp]


def estimate_matrix(w, c):
    """one E-step of carmel --matrix-fb (oracle/matrix.hpp); dict(counts_ln, pair_logprob, sum_logprob,
    sum_weighted_logprob, eps_back_edges)"""
    ns, na, _ = w.dims()
    npairs = len(c.arrays()["weight"])
    counts = np.zeros(na)
    pl = np.zeros(npairs)
    sums = np.zeros(3)
    _chk(lib.orc_estimate_matrix(w.h, c.h, _p(counts), _p(pl), _p(sums)))
    return dict(counts_ln=counts, pair_logprob=pl, sum_logprob=sums[0], sum_weighted_logprob=sums[1], eps_back_edges=int(sums[2]))


def lattice(w, c, pair, prune=True):
    ns, na, fin, nb = C.c_uint32(), C.c_uint64(), C.c_uint32(), C.c_uint32()
    _chk(lib.orc_lattice(w.h, c.h, pair, int(prune), C.byref(ns), C.byref(na), C.byref(fin), None, None, None, None,
                         C.byref(nb)))
    if ns.value == 0:
        return None
    s, d, a = (np.zeros(na.value, np.uint32) for _ in range(3))
    order = np.zeros(ns.value, np.uint32)
    _chk(lib.orc_lattice(w.h, c.h, pair, int(prune), C.byref(ns), C.byref(na), C.byref(fin), _p(s), _p(d), _p(a),
                         _p(order), C.byref(nb)))
    return dict(n_states=ns.value, fin=fin.value, src=s, dst=d, arcid=a, reverse_order=order, n_back_edges=nb.value)


def train(w, c, norm_group=0, add_count=0.0, weight_is_prior_count=False, smooth_floor=0.0, converge_arc_delta=1e-4,
          converge_ppx_ratio=.999, max_iter=500, cache=False, prune=True, max_trace=1000, rate_growth=1.0):
    tr = np.zeros((max_trace, 8))
    n = C.c_int(0)
    best = C.c_double(0)
    lib.orc_set_rate_growth(C.c_double(rate_growth))
    _chk(lib.orc_train(w.h, c.h, norm_group, add_count, int(weight_is_prior_count), smooth_floor, converge_arc_delta,
                       converge_ppx_ratio, max_iter, int(cache), int(prune), _p(tr), max_trace, C.byref(n),
                       C.byref(best)))
    lib.orc_set_rate_growth(C.c_double(1.0))
    rows = [dict(zip(TRACE_FIELDS, tr[i])) for i in range(n.value)]
    return best.value, rows


def train_cascade_text(wfst_texts, corpus_text, normby=None, priors=None, max_iter=500, converge_arc_delta=1e-4,
                       converge_ppx_ratio=.999, cache=False, full=True, onearc=True, max_trace=1000):
    n = len(wfst_texts)
    arr = (C.c_char_p * n)(*[t.encode() for t in wfst_texts])
    tr = np.zeros((max_trace, 8))
    nt = C.c_int(0)
    out = vp()
    olen = C.c_uint64(0)
    dims = np.zeros(2, np.uint32)
    pri = None if priors is None else np.ascontiguousarray(priors, dtype=np.float64)
    _chk(lib.orc_train_cascade_text(n, arr, corpus_text.encode(), (normby or "").encode() or None, _p(pri), max_iter,
                                    converge_arc_delta, converge_ppx_ratio, int(cache), int(full), int(onearc),
                                    C.byref(out), C.byref(olen), _p(tr), max_trace, C.byref(nt), _p(dims)))
    raw = C.string_at(out, olen.value)
    lib.orc_free_str(out)
    texts = [t.decode() for t in raw.split(b"\0")[:n]]
    rows = [dict(zip(TRACE_FIELDS, tr[i])) for i in range(nt.value)]
    return rows, texts, (int(dims[0]), int(dims[1]))


lib.orc_cascade_compose_text.restype = vp
lib.orc_bench_em_fit.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, vp]
lib.orc_bench_em_fit_check.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, vp, vp, vp]
lib.orc_cascade_compose_text.argtypes = [C.c_int, vp]
lib.orc_cascade_compose_text_ex.restype = vp
lib.orc_cascade_compose_text_ex.argtypes = [C.c_int, vp, C.c_int, C.c_int]
lib.orc_cascade_free.argtypes = [vp]
lib.orc_cascade_composed.restype = vp
lib.orc_cascade_composed.argtypes = [vp]
lib.orc_cascade_dims.argtypes = [vp, vp]
lib.orc_cascade_export.argtypes = [vp] * 8
lib.orc_cascade_member_states.argtypes = [vp, vp]
lib.orc_cascade_corpus.restype = vp
lib.orc_cascade_corpus.argtypes = [vp, C.c_char_p]
lib.orc_cascade_write_member.restype = vp
lib.orc_cascade_write_member.argtypes = [vp, C.c_int, vp, C.c_int, C.c_int]


class OracleCascade(object):
    """composition + chain bookkeeping of `carmel --train-cascade a b ...` as flat arrays"""

    def __init__(self, wfst_texts, remember=True, dash_a=False):
        """remember=False: plain `carmel a b` (no chains); dash_a: carmel -a (2-state filter, compose.cc:219-313)"""
        n = len(wfst_texts)
        arr = (C.c_char_p * n)(*[t.encode() for t in wfst_texts])
        self.h = lib.orc_cascade_compose_text_ex(n, arr, int(remember), int(dash_a))
        if not self.h:
            raise RuntimeError("oracle: " + lib.orc_last_error().decode())
        dims = np.zeros(4, np.uint64)
        lib.orc_cascade_dims(self.h, _p(dims))
        self.n_members, self.n_params, self.n_chains, ne = (int(x) for x in dims)
        self.param_logw = np.zeros(self.n_params)
        self.param_group, self.param_member, self.param_src, self.param_in = (
            np.zeros(self.n_params, np.uint32) for _ in range(4))
        self.chain_off = np.zeros(self.n_chains + 1, np.uint64)
        self.chain_param = np.zeros(max(ne, 1), np.uint64)
        lib.orc_cascade_export(self.h, _p(self.param_logw), _p(self.param_group), _p(self.param_member),
                               _p(self.param_src), _p(self.param_in), _p(self.chain_off), _p(self.chain_param))

    def __del__(self):
        if getattr(self, "h", None):
            lib.orc_cascade_free(self.h)
            self.h = None

    def composed(self):
        return OracleWfst(lib.orc_cascade_composed(self.h))

    def corpus(self, text):
        return OracleCorpus(lib.orc_cascade_corpus(self.h, text.encode()))

    def write_member(self, m, param_logw, full=True, onearc=True):
        pl = np.ascontiguousarray(param_logw, dtype=np.float64)
        s = lib.orc_cascade_write_member(self.h, m, _p(pl), int(full), int(onearc))
        txt = C.string_at(s).decode()
        lib.orc_free_str(s)
        return txt

    @property
    def member_states(self):
        """state count of every member transducer (a JOINT member has one norm group per state)"""
        out = np.zeros(self.n_members, np.uint32)
        lib.orc_cascade_member_states(self.h, _p(out))
        return out

    def as_dict(self, member_norm, member_add_count=None):
        n = self.n_members
        return dict(param_logw=self.param_logw, param_group=self.param_group, param_member=self.param_member,
                    param_src=self.param_src, param_in=self.param_in,
                    member_norm=np.ascontiguousarray(member_norm, dtype=np.int32),
                    member_add_count=np.ascontiguousarray(member_add_count if member_add_count is not None
                                                          else np.zeros(n), dtype=np.float64),
                    chain_off=self.chain_off, chain_param=self.chain_param)


UNIFORM_FN = C.CFUNCTYPE(C.c_double, C.c_uint32, C.c_uint32, C.c_uint32)
lib.orc_set_native_uniform_seed.argtypes = [C.c_uint64]
lib.orc_native_uniform.restype = C.c_double
lib.orc_native_uniform.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32]


def _uniform_cb(uniform):
    """`uniform` is a Python callable u(iter, block, step), or an int: the seed of the oracle's own counter-based stream
    (orc_native_uniform, value for value the product's carmel_hip_gibbs_uniform(seed, ...)) -- no Python call per draw"""
    if isinstance(uniform, (int, np.integer)):
        lib.orc_set_native_uniform_seed(int(uniform))
        return C.cast(lib.orc_native_uniform, UNIFORM_FN)
    return UNIFORM_FN(uniform)
lib.orc_gibbs_run.argtypes = [vp, vp, C.c_char_p, vp, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_int,
                              UNIFORM_FN, vp, vp, vp, vp, vp, C.c_uint64, vp]


lib.orc_gibbs_power.argtypes = [C.c_double, C.c_double, C.c_uint32, C.c_uint32]
lib.orc_gibbs_power.restype = C.c_double


def gibbs_power(high_temp, low_temp, iters, sweep):
    """1 / temperature of one sweep (gibbs.hpp:838-839)"""
    return lib.orc_gibbs_power(high_temp, low_temp, iters, sweep)


lib.orc_fem_export.argtypes = [vp, vp, C.c_char_p, vp, C.c_int, C.c_char_p, C.c_ulong]
lib.orc_fem_export.restype = C.c_long


def fem_export(cascade, corpus, which, normby=None, priors=None):
    """carmel --fem-forest (which=0) / --fem-norm (1) / --fem-param (2) / --fem-alpha (3) as text"""
    pri = None if priors is None else np.ascontiguousarray(priors, dtype=np.float64)
    nb = (normby or "").encode() or None
    n = lib.orc_fem_export(cascade.h, corpus.h, nb, _p(pri), which, None, 0)
    if n < 0:
        raise RuntimeError("orc_fem_export failed")
    buf = C.create_string_buffer(n + 1)
    lib.orc_fem_export(cascade.h, corpus.h, nb, _p(pri), which, buf, n + 1)
    return buf.value.decode()


lib.orc_set_gibbs_prior_inference.argtypes = [C.c_double, C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_uint32, vp, C.c_uint32]
lib.orc_gibbs_last_prior_trace.argtypes = [vp, C.c_uint32, vp, C.c_uint32]
lib.orc_gibbs_last_prior_trace.restype = C.c_uint32


def gibbs_run(cascade, corpus, uniform, normby=None, priors=None, iters=10, burnin=0, uniform_p0=False,
              dirichlet_p0=False, final_counts=False, exclude_prior=False, max_samples=1 << 22, high_temp=1.0,
              low_temp=1.0, expectation=False, restarts=0, argmax_final=False, argmax_sum=False, init_em=0,
              em_p0=False, init_from_p0=False, prior_inference=None, include_self=False, random_start=False, state_trace=False):
    """carmel --crp on an OracleCascade; prior_inference = dict(stddev, global_, local, restart_fresh, start, end, groupby)
    turns on prior-scale inference (gibbs.hpp:525-553). `uniform(iter, block, step)` supplies every random01() draw.
    Returns dict(iter_logprob, iter_cheap_logprob, param_logw, samples=[per block list of member-arc indices])"""
    n = cascade.n_params
    ilp, icl = np.zeros((iters + 1) * (restarts + 1)), np.zeros((iters + 1) * (restarts + 1))
    plw = np.zeros(n)
    samp = np.zeros(max_samples, np.uint32)
    nb = C.c_uint32(0)
    n_pairs = len(corpus.arrays()["weight"])
    off = np.zeros(n_pairs + 2, np.uint64)
    pri = None if priors is None else np.ascontiguousarray(priors, dtype=np.float64)
    cb = _uniform_cb(uniform)
    lib.orc_set_gibbs_temps(C.c_double(high_temp), C.c_double(low_temp))
    lib.orc_set_gibbs_expectation(int(expectation))
    lib.orc_set_gibbs_self_start(int(include_self), int(random_start))
    lib.orc_set_gibbs_restarts(int(restarts), int(argmax_final), int(argmax_sum))
    lib.orc_set_gibbs_init_em(int(init_em), int(em_p0))
    lib.orc_set_gibbs_init_from_p0(int(init_from_p0))
    pi = dict(prior_inference or {})
    gb = np.ascontiguousarray(pi.get("groupby", []), dtype=np.int32)
    lib.orc_set_gibbs_prior_inference(C.c_double(pi.get("stddev", 0.0)), int(pi.get("global_", False)), int(pi.get("local", False)),
                                      int(pi.get("restart_fresh", False)), int(pi.get("start", 0)), int(pi.get("end", 0)),
                                      _p(gb) if len(gb) else None, len(gb))
    lib.orc_set_gibbs_state_trace(int(state_trace))
    _chk(lib.orc_gibbs_run(cascade.h, corpus.h, (normby or "").encode() or None, _p(pri), iters, burnin,
                           int(uniform_p0), int(dirichlet_p0), int(final_counts), int(exclude_prior), cb, _p(ilp),
                           _p(icl), _p(plw), _p(samp), _p(off), max_samples, C.byref(nb)))
    samples = [samp[int(off[b]):int(off[b + 1])].tolist() for b in range(nb.value)]
    after = np.zeros(len(ilp))
    lib.orc_gibbs_last_after(_p(after), len(after))
    lib.orc_set_gibbs_init_from_p0(0)
    ptrace, cum = np.zeros((len(ilp), 6)), np.zeros(4096)
    lib.orc_gibbs_last_prior_trace.restype = C.c_uint32
    ncum = lib.orc_gibbs_last_prior_trace(_p(ptrace), len(ilp), _p(cum), len(cum))
    lib.orc_set_gibbs_prior_inference(C.c_double(0.0), 0, 0, 0, 0, 0, None, 0)
    out = dict(iter_logprob=ilp, iter_cheap_logprob=icl, iter_after_logprob=after, param_logw=plw, samples=samples,
               best_run=lib.orc_gibbs_best_run(), prior_trace=ptrace, prior_cumulative=cum[:ncum])
    if state_trace:
        # what --print-counts-* / --print-norms-* show: after every sweep of every run, per parameter (member-arc order)
        # {count x, time-weighted sum s, tmax, prior}; the kept run's finalized {count, prob}; per parameter {define_param id,
        # norm id or -1, prior-scale group or 0}
        nsw = len(ilp)
        st, fin, ids = np.zeros(nsw * n * 4), np.zeros(n * 2), np.zeros(n * 3, np.int32)
        lib.orc_set_gibbs_state_trace(0)
        lib.orc_gibbs_last_state.restype = C.c_uint64
        lib.orc_gibbs_last_state(_p(st), C.c_uint64(len(st)), _p(fin), C.c_uint64(len(fin)), _p(ids), C.c_uint64(len(ids)))
        out.update(state=st.reshape(nsw, n, 4), final=fin.reshape(n, 2), ids=ids.reshape(n, 3))
    return out


lib.orc_forests_parse.restype = vp
lib.orc_forests_parse.argtypes = [C.c_char_p, C.c_char_p]
lib.orc_forests_free.argtypes = [vp]
lib.orc_forests_dims.argtypes = [vp, vp]
lib.orc_forests_export.argtypes = [vp] * 7
lib.orc_forests_set_weights.argtypes = [vp, vp]
lib.orc_forests_get_weights.argtypes = [vp, vp]
lib.orc_forests_estimate.restype = C.c_double
lib.orc_forests_estimate.argtypes = [vp, C.c_double, vp, vp]
lib.orc_forests_maximize.restype = C.c_double
lib.orc_forests_maximize.argtypes = [vp, C.c_double, C.c_int]
lib.orc_forests_init_rule_weights.argtypes = [vp, C.c_int]
lib.orc_forests_randomize.argtypes = [vp, vp]
lib.orc_forests_viterbi_line.restype = C.c_int
lib.orc_forests_viterbi_line.argtypes = [vp, C.c_uint64, C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_double)]
lib.orc_forests_gibbs.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_double, UNIFORM_FN, vp, vp, vp, vp,
                                  C.c_uint64]


class OracleForests(object):
    """forest-em inputs (-f forests, -n normgroups) parsed and restated by the oracle"""

    def __init__(self, forests_text, normgroups_text):
        self.h = lib.orc_forests_parse(forests_text.encode(), normgroups_text.encode())
        if not self.h:
            raise ValueError("oracle: " + lib.orc_last_error().decode())
        d = np.zeros(5, np.uint64)
        lib.orc_forests_dims(self.h, _p(d))
        self.n_forests, self.n_nodes, self.n_rules, self.n_groups, ne = (int(x) for x in d)
        self.node_off = np.zeros(self.n_forests + 1, np.uint64)
        self.label, self.next = np.zeros(self.n_nodes, np.uint32), np.zeros(self.n_nodes, np.uint32)
        self.ref = np.zeros(self.n_nodes, np.int32)
        self.group_off = np.zeros(self.n_groups + 1, np.uint64)
        self.group_rule = np.zeros(max(ne, 1), np.uint32)
        lib.orc_forests_export(self.h, _p(self.node_off), _p(self.label), _p(self.ref), _p(self.next),
                               _p(self.group_off), _p(self.group_rule))

    def __del__(self):
        if getattr(self, "h", None):
            lib.orc_forests_free(self.h)
            self.h = None

    def set_weights(self, lw):
        lw = np.ascontiguousarray(lw, dtype=np.float64)
        assert len(lw) == self.n_rules
        lib.orc_forests_set_weights(self.h, _p(lw))

    def weights(self):
        lw = np.zeros(self.n_rules)
        lib.orc_forests_get_weights(self.h, _p(lw))
        return lw

    def estimate(self, prior_count=0.0):
        c, pf = np.zeros(self.n_rules), np.zeros(self.n_forests)
        avg = lib.orc_forests_estimate(self.h, prior_count, _p(c), _p(pf))
        return avg, c, pf

    def maximize(self, add_k=0.0, zero_zerocounts=False):
        return lib.orc_forests_maximize(self.h, add_k, int(zero_zerocounts))

    def init_rule_weights(self, ones=False):
        """forest-em.hpp:297-318 without -I: uniform per norm group (rules of no group: weight 0), or all 1 (-u)"""
        lib.orc_forests_init_rule_weights(self.h, int(ones))

    def randomize(self, fraction):
        """forest-em.hpp:393-399: fraction[rule] per member, each group divided by its sum"""
        fr = np.ascontiguousarray(fraction, np.float64)
        lib.orc_forests_randomize(self.h, _p(fr))

    def viterbi_line(self, forest, mode=0):
        """the -v / --outviterbi-file line of one forest, and ln of its best derivation"""
        buf = C.create_string_buffer(1 << 16)
        best = C.c_double(0)
        n = lib.orc_forests_viterbi_line(self.h, forest, mode, buf, len(buf), C.byref(best))
        assert n >= 0
        return buf.value.decode(), best.value

    def gibbs(self, uniform, iters, burnin=0, alpha=0.1, uniform_p0=False, final_counts=False, max_samples=1 << 22,
              alphas=None, high_temp=1.0, low_temp=1.0, prior_inference=None, exclude_prior=False):
        lib.orc_set_gibbs_temps(C.c_double(high_temp), C.c_double(low_temp))
        lib.orc_forests_set_exclude_prior(int(exclude_prior))
        pi = dict(prior_inference or {})
        lib.orc_set_gibbs_prior_inference(C.c_double(pi.get("stddev", 0.0)), int(pi.get("global_", False)),
                                          int(pi.get("local", False)), 0, int(pi.get("start", 0)), int(pi.get("end", 0)), None, 0)
        al = None if alphas is None else np.ascontiguousarray(alphas, dtype=np.float64)
        lib.orc_forests_set_alphas(_p(al), 0 if al is None else len(al))
        ilp, icl = np.zeros(iters + 1), np.zeros(iters + 1)
        samp = np.zeros(max_samples, np.uint32)
        off = np.zeros(self.n_forests + 1, np.uint64)
        cb = _uniform_cb(uniform)
        _chk(lib.orc_forests_gibbs(self.h, iters, burnin, int(uniform_p0), int(final_counts), alpha, cb, _p(ilp), _p(icl),
                                   _p(samp), _p(off), max_samples))
        samples = [samp[int(off[b]):int(off[b + 1])].tolist() for b in range(self.n_forests)]
        ptrace, cum = np.zeros((iters + 1, 6)), np.zeros(1 << 16)
        ncum = lib.orc_gibbs_last_prior_trace(_p(ptrace), iters + 1, _p(cum), len(cum))
        lib.orc_set_gibbs_prior_inference(C.c_double(0.0), 0, 0, 0, 0, 0, None, 0)
        fin = np.zeros(2 * self.n_rules)
        lib.orc_forests_gibbs_last_final.restype = C.c_uint64
        lib.orc_forests_gibbs_last_final(_p(fin), C.c_uint64(len(fin)))
        # final: per rule {count as finalize_cumulative_counts left it, norm group or -1} (what forest-em --print-counts-* shows)
        return dict(iter_logprob=ilp, iter_cheap_logprob=icl, samples=samples, prior_trace=ptrace,
                    prior_cumulative=cum[:ncum], final=fin.reshape(self.n_rules, 2))


CLI = os.path.join(_HERE, "oracle_carmel")
