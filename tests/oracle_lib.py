"""ctypes binding of oracle/liboracle.so -- test infrastructure only (see oracle/epa_oracle.c).

Imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; never by the product.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
_LIBS = {}
# flags of the second build of the SAME source that bench.py times as `cpu_baseline` (the strict
# build stays the parity checker): full optimisation for the host CPU the bench runs on.  Built on
# that box (-march=native code must not travel between machines), so it is never shipped.
FAST_CFLAGS = ("-O3 -march=native -funroll-loops -ffp-contract=fast -fno-math-errno -std=gnu99 -fPIC -fopenmp "
               "-DORC_FAST_KERNELS")   # + the vector-shaped 4-state x 4-category loops (oracle/epa_oracle.c)


def build(force=False):
    so = os.path.join(ORACLE_DIR, "liboracle.so")
    src = os.path.join(ORACLE_DIR, "epa_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-B", "liboracle.so"],
                              stdout=subprocess.DEVNULL)
    return so


def build_fast():
    """oracle/_fast/liboracle_fast.so: always rebuilt for the CPU of the machine that calls it"""
    subprocess.check_call(["make", "-C", ORACLE_DIR, "-B", "fast", "FAST_CFLAGS=" + FAST_CFLAGS],
                          stdout=subprocess.DEVNULL)
    return os.path.join(ORACLE_DIR, "_fast", "liboracle_fast.so")


def lib(fast=False):
    if fast not in _LIBS:
        L = C.CDLL(build_fast() if fast else build())
        dp = C.POINTER(C.c_double)
        L.orc_create.restype = C.c_void_p
        L.orc_create.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p),
                                 C.c_size_t, C.c_int, dp, dp, C.c_int, dp, dp, C.c_double]
        L.orc_destroy.argtypes = [C.c_void_p]
        L.orc_num_branches.argtypes = [C.c_void_p]
        L.orc_num_tips.argtypes = [C.c_void_p]
        L.orc_tree_lnl.restype = C.c_double
        L.orc_tree_lnl.argtypes = [C.c_void_p, C.c_int]
        L.orc_gamma_rates.argtypes = [C.c_double, C.c_int, dp]
        L.orc_numbered_newick.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_int]
        L.orc_branch_info.argtypes = [C.c_void_p, C.c_int, dp, C.POINTER(C.c_int)]
        L.orc_branch_sides.argtypes = [C.c_void_p, C.c_int, dp, C.POINTER(C.c_uint32), dp,
                                       C.POINTER(C.c_uint32)]
        L.orc_branch_lookup.argtypes = [C.c_void_p, C.c_int, dp]
        L.orc_preplace.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_char_p), C.c_int, dp]
        L.orc_thorough.argtypes = [C.c_void_p, C.c_long, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                   C.POINTER(C.c_char_p), C.c_int, dp, dp, dp,
                                   C.POINTER(C.c_long)]
        L.orc_direct_default_lnl.restype = C.c_double
        L.orc_direct_default_lnl.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_int]
        L.orc_pendant_derivatives.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_double, dp, dp,
                                              dp]
        L.orc_model_evals.restype = dp
        L.orc_model_evals.argtypes = [C.c_void_p]
        L.orc_model_u.restype = dp
        L.orc_model_u.argtypes = [C.c_void_p]
        L.orc_model_uinv.restype = dp
        L.orc_model_uinv.argtypes = [C.c_void_p]
        L.orc_set_aa_x_quirk.argtypes = [C.c_void_p, C.c_int]
        L.orc_next_create_rate_scalers.argtypes = [C.c_int]
        L.orc_rate_scalers.argtypes = [C.c_void_p]
        L.orc_set_raxml_blo.argtypes = [C.c_void_p, C.c_int]
        L.orc_set_blo.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int]
        L.orc_max_threads.restype = C.c_int
        L.orc_score_at.argtypes = [C.c_void_p, C.c_long, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                   C.POINTER(C.c_char_p), C.c_int, dp, dp, dp, dp]
        L.orc_set_rounding_variant.argtypes = [C.c_void_p, C.c_uint64]
        L.orc_trace_pair.restype = C.c_long
        L.orc_trace_pair.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_int, dp, C.c_long, dp, dp, dp]
        L.orc_char_column.argtypes = [C.c_int, C.c_char, C.c_int]
        L.orc_char_mask.restype = C.c_uint32
        L.orc_char_mask.argtypes = [C.c_int, C.c_char]
        _LIBS[fast] = L
    return _LIBS[fast]


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _strs(lst):
    arr = (C.c_char_p * len(lst))()
    arr[:] = [s if isinstance(s, bytes) else s.encode() for s in lst]
    return arr


def gamma_rates(alpha, k=4):
    out = np.zeros(k)
    lib().orc_gamma_rates(alpha, k, _dp(out))
    return out


class Oracle:
    """The CPU restatement of the reference tree + placement evaluator."""

    def __init__(self, newick, labels, seqs, states, subst, freqs, rates, weights=None, pinv=0.0,
                 rate_scalers=False, fast=False):
        """rate_scalers: per-rate scaling (PLL_ATTRIB_RATE_SCALERS; the reference turns it on for
        trees with more than 2000 tips, src/io/file_io.cpp:211-214).  fast: the -O3 -march=native
        build (bench.py's timed CPU leg only; parity always uses the strict build)"""
        L = self.L = lib(fast)
        self.rate_scalers = bool(rate_scalers)
        L.orc_next_create_rate_scalers(int(self.rate_scalers))
        self.W = len(seqs[0])
        self.s = states
        self.c = len(rates)
        subst = np.ascontiguousarray(subst, dtype=np.float64)
        freqs = np.ascontiguousarray(freqs, dtype=np.float64)
        rates = np.ascontiguousarray(rates, dtype=np.float64)
        if weights is None:
            weights = np.full(self.c, 1.0 / self.c)
        weights = np.ascontiguousarray(weights, dtype=np.float64)
        self._keep = (_strs(labels), _strs(seqs))
        self.h = L.orc_create(newick.encode(), len(labels), self._keep[0], self._keep[1], self.W,
                              states, _dp(subst), _dp(freqs), self.c, _dp(rates), _dp(weights),
                              pinv)
        L.orc_next_create_rate_scalers(0)
        if not self.h:
            raise RuntimeError("oracle: orc_create failed (tree / MSA / model)")
        self.B = L.orc_num_branches(self.h)
        self.n_tips = L.orc_num_tips(self.h)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orc_destroy(self.h)
            self.h = None

    def tree_lnl(self, b):
        return self.L.orc_tree_lnl(self.h, b)

    def set_blo(self, min_branch=0.0, max_branch=0.0, default_branch=0.0, epsilon=0.0, newton_variant=-1):
        """optimiser constants / Newton variant (see orc_set_blo); 0 / -1 keep the current value"""
        self.L.orc_set_blo(self.h, min_branch, max_branch, default_branch, epsilon, newton_variant)

    def set_raxml_blo(self, on=True):
        """--raxml-blo: radius-1 local BLO (optimize.cpp:274-279) instead of the sliding rule"""
        self.L.orc_set_raxml_blo(self.h, int(on))

    def set_aa_x_quirk(self, on=True):
        """quirk D4: AA 'X' preplaced in the 'N' (asparagine) column (Lookup_Store.hpp:63-66)"""
        self.L.orc_set_aa_x_quirk(self.h, int(on))

    def numbered_newick(self, prec=10):
        buf = C.create_string_buffer(256 * (self.B + 4))
        n = self.L.orc_numbered_newick(self.h, prec, buf, len(buf))
        assert n > 0
        return buf.value.decode()

    def branch_info(self, b):
        o, t = C.c_double(), C.c_int()
        self.L.orc_branch_info(self.h, b, C.byref(o), C.byref(t))
        return o.value, bool(t.value)

    def branch_sides(self, b):
        n = self.W * self.c * self.s
        cp, cd = np.zeros(n), np.zeros(n)
        nsc = self.W * (self.c if self.rate_scalers else 1)
        sp, sd = np.zeros(nsc, np.uint32), np.zeros(nsc, np.uint32)
        u32 = C.POINTER(C.c_uint32)
        self.L.orc_branch_sides(self.h, b, _dp(cp), sp.ctypes.data_as(u32), _dp(cd),
                               sd.ctypes.data_as(u32))
        shp = (self.W, self.c, self.s)
        return cp.reshape(shp), sp, cd.reshape(shp), sd

    def branch_lookup(self, b):
        ncol = 16 if self.s == 4 else 24
        t = np.zeros((self.W, ncol))
        self.L.orc_branch_lookup(self.h, b, _dp(t))
        return t

    def eigen(self):
        L = self.L
        s = self.s
        ev = np.ctypeslib.as_array(L.orc_model_evals(self.h), (20,))[:s].copy()
        u = np.ctypeslib.as_array(L.orc_model_u(self.h), (400,))[:s * s].copy().reshape(s, s)
        ui = np.ctypeslib.as_array(L.orc_model_uinv(self.h), (400,))[:s * s].copy().reshape(s, s)
        return ev, u, ui

    def preplace(self, queries, premask=True):
        q = _strs(queries)
        out = np.zeros((len(queries), self.B))
        rc = self.L.orc_preplace(self.h, len(queries), q, int(premask), _dp(out))
        if rc:
            raise RuntimeError("oracle preplace error %d" % rc)
        return out

    def thorough(self, pair_branch, pair_seq, queries, premask=True):
        pb = np.ascontiguousarray(pair_branch, dtype=np.int32)
        ps = np.ascontiguousarray(pair_seq, dtype=np.int32)
        n = len(pb)
        lnl, pen, dis = np.zeros(n), np.zeros(n), np.zeros(n)
        stats = (C.c_long * 3)()
        q = _strs(queries)
        ip = C.POINTER(C.c_int)
        rc = self.L.orc_thorough(self.h, n, pb.ctypes.data_as(ip), ps.ctypes.data_as(ip), q,
                                int(premask), _dp(lnl), _dp(pen), _dp(dis), stats)
        if rc:
            raise RuntimeError("oracle thorough error %d" % rc)
        self.last_stats = {"rounds": stats[0], "newton_evals": stats[1], "reverts": stats[2]}
        return lnl, pen, dis

    def score_at(self, pair_branch, pair_seq, queries, pendant, distal, proximal=None, premask=True):
        """edge lnL of each (branch, query) pair at GIVEN lengths (no optimiser): orc_score_at.
        proximal None = the sliding rule's original - distal."""
        pb = np.ascontiguousarray(pair_branch, dtype=np.int32)
        ps = np.ascontiguousarray(pair_seq, dtype=np.int32)
        pe = np.ascontiguousarray(pendant, dtype=np.float64)
        di = np.ascontiguousarray(distal, dtype=np.float64)
        pr = None if proximal is None else np.ascontiguousarray(proximal, dtype=np.float64)
        out = np.zeros(len(pb))
        ip = C.POINTER(C.c_int)
        rc = self.L.orc_score_at(self.h, len(pb), pb.ctypes.data_as(ip), ps.ctypes.data_as(ip),
                                _strs(queries), int(premask), _dp(pe), _dp(di),
                                None if pr is None else _dp(pr), _dp(out))
        if rc:
            raise RuntimeError("oracle score_at error %d" % rc)
        return out

    def set_rounding_variant(self, seed=0):
        """0: the oracle as written; != 0: a faithfully rounded sibling (1-ulp moves of every exp of
        the derivative tables and every site likelihood of the score, odd seeds sum sites from the
        far end) -- see orc_model.rounding_variant"""
        self.L.orc_set_rounding_variant(self.h, int(seed))

    def trace_pair(self, b, query, premask=True, cap=4096):
        """(rows, lnl, pendant, distal) of one pair; rows [n][4] = {1|2, t, f, f'} per derivative
        evaluation of the pendant / distal solve, {3, new -lnL, old -lnL, reverted} per round"""
        rows = np.zeros((cap, 4))
        l, p, d = C.c_double(), C.c_double(), C.c_double()
        n = self.L.orc_trace_pair(self.h, int(b), query.encode(), int(premask), _dp(rows), cap,
                                 C.byref(l), C.byref(p), C.byref(d))
        return rows[:min(n, cap)], l.value, p.value, d.value

    def direct_default_lnl(self, b, query, premask=True):
        return self.L.orc_direct_default_lnl(self.h, b, query.encode(), int(premask))

    def pendant_derivatives(self, b, query, t):
        f, df, l = C.c_double(), C.c_double(), C.c_double()
        self.L.orc_pendant_derivatives(self.h, b, query.encode(), t, C.byref(f), C.byref(df),
                                      C.byref(l))
        return f.value, df.value, l.value
