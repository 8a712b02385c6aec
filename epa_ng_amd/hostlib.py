"""ctypes binding of libepa_host.so: the C++ host pipeline above the C-ABI (reference-tree
precompute, chunk loop, heuristics, LWR / filters, jplace).  No likelihood code in Python."""
import ctypes as C
import os

import numpy as np

from . import api

HERE = os.path.dirname(os.path.abspath(__file__))
HOST_SO = os.path.join(HERE, "libepa_host.so")
_LIB = None


def host_lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(HOST_SO):
            raise ImportError("libepa_host.so is missing: run build() of __graft_entry__.py")
        api.dev_lib()  # libepa_host links against libepa_dev (rpath $ORIGIN)
        L = C.CDLL(HOST_SO)
        dp = C.POINTER(C.c_double)
        L.epa_host_last_error.restype = C.c_char_p
        L.epa_host_ref_create.restype = C.c_void_p
        L.epa_host_ref_create.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_char_p),
                                          C.POINTER(C.c_char_p), C.c_char_p, C.c_int, dp, dp,
                                          C.c_int, dp, dp]
        L.epa_host_ref_create_ex.restype = C.c_void_p
        L.epa_host_ref_create_ex.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_char_p),
                                             C.POINTER(C.c_char_p), C.c_char_p, C.c_int, dp, dp,
                                             C.c_int, dp, dp, C.c_double]
        L.epa_host_ref_create_ex2.restype = C.c_void_p
        L.epa_host_ref_create_ex2.argtypes = L.epa_host_ref_create_ex.argtypes + [C.c_int]
        L.epa_host_ref_in_rtree.argtypes = [C.c_void_p, C.c_uint32, C.c_double,
                                            C.POINTER(C.c_uint32), C.POINTER(C.c_double)]
        L.epa_host_parse_model.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t]
        L.epa_host_ref_destroy.argtypes = [C.c_void_p]
        L.epa_host_configure_threads.restype = C.c_int
        u32p = C.POINTER(C.c_uint32)
        L.epa_host_ref_dims.argtypes = [C.c_void_p, u32p, u32p, u32p, u32p]
        L.epa_host_ref_tree_logl.restype = C.c_double
        L.epa_host_ref_tree_logl.argtypes = [C.c_void_p, C.c_uint32]
        L.epa_host_ref_numbered_newick.argtypes = [C.c_void_p, C.c_uint, C.c_char_p, C.c_size_t]
        L.epa_host_ref_model.argtypes = [C.c_void_p, dp, dp, dp, dp, dp, dp]
        L.epa_host_ref_subst.argtypes = [C.c_void_p, dp]
        L.epa_host_ref_model_string.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
        L.epa_host_ref_branch.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p),
                                          C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                          C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), dp]
        L.epa_host_ref_tipmap.restype = C.c_uint32
        L.epa_host_ref_tipmap.argtypes = [C.c_void_p, u32p, C.c_uint32]
        L.epa_host_dev_create.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        L.epa_host_dev_create_ex.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        L.epa_host_dev_create_flags.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_uint32,
                                                C.POINTER(C.c_void_p)]
        L.epa_host_dev_create_opts.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_double,
                                               C.POINTER(C.c_void_p)]
        L.epa_host_place_file.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_uint32, C.c_int,
                                          C.c_double, C.c_int, C.c_int, C.c_char_p,
                                          C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.epa_host_filter.argtypes = [dp, C.c_uint32, C.c_double, C.c_int, C.c_uint32, C.c_uint32,
                                      u32p, u32p]
        L.epa_host_heuristic.argtypes = [dp, C.c_uint32, C.c_uint32, C.c_int, C.c_double, u32p,
                                         u32p, C.c_uint64, C.POINTER(C.c_uint64)]
        _LIB = L
    return _LIB


def cli_exe():
    """path of the epa-ng-amd executable; (re)built from source first when the host toolchain is
    present, so a stale binary is never what gets tested"""
    import importlib.util
    import shutil
    if shutil.which("g++"):
        spec = importlib.util.spec_from_file_location("epa_build", os.path.join(HERE, "build.py"))
        b = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(b)
        b.build_host()
    exe = os.path.join(HERE, "epa-ng-amd")
    if not os.path.exists(exe):
        raise ImportError("epa-ng-amd is missing: run build() of __graft_entry__.py")
    return exe


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _strs(lst):
    arr = (C.c_char_p * len(lst))()
    arr[:] = [s if isinstance(s, bytes) else s.encode() for s in lst]
    return arr


def configure_threads():
    """caps this process's OpenMP threads at the CPUs it may really use (affinity mask and cgroup
    quota); returns the count.  Shared runtime: also governs the test oracle's OpenMP loops."""
    return host_lib().epa_host_configure_threads()


class Reference:
    """Reference tree + MSA + model with all directional CLVs precomputed on the host
    (mirror of the reference's `Tree`, src/tree/Tree.cpp:16-56)."""

    def __init__(self, newick, labels, seqs, model=None, states=None, subst=None, freqs=None,
                 rates=None, weights=None, pinv=0.0, preserve_rooting=True):
        L = host_lib()
        self._keep = (_strs(labels), _strs(seqs))
        if model is not None:
            self.h = L.epa_host_ref_create_ex2(newick.encode(), len(labels), self._keep[0],
                                               self._keep[1], model.encode(), 0, None, None, 0, None,
                                               None, 0.0, int(preserve_rooting))
        else:
            subst = np.ascontiguousarray(subst, np.float64)
            freqs = np.ascontiguousarray(freqs, np.float64)
            rates = np.ascontiguousarray(rates, np.float64)
            w = None if weights is None else np.ascontiguousarray(weights, np.float64)
            self.h = L.epa_host_ref_create_ex2(newick.encode(), len(labels), self._keep[0],
                                               self._keep[1], None, states, _dp(subst), _dp(freqs),
                                               len(rates), _dp(rates), None if w is None else _dp(w),
                                               float(pinv), int(preserve_rooting))
        if not self.h:
            raise RuntimeError(L.epa_host_last_error().decode())
        s, c, w_, b = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint32()
        L.epa_host_ref_dims(self.h, C.byref(s), C.byref(c), C.byref(w_), C.byref(b))
        self.s, self.c, self.W, self.B = s.value, c.value, w_.value, b.value

    def __del__(self):
        if getattr(self, "h", None):
            host_lib().epa_host_ref_destroy(self.h)
            self.h = None

    def tree_lnl(self, branch=0):
        return host_lib().epa_host_ref_tree_logl(self.h, branch)

    def in_rtree(self, branch, distal):
        """(edge, distal) on the unrooted working tree -> on the rooted input tree, or None when
        the input was unrooted / rooting is not preserved (rtree_mapper::in_rtree)."""
        ob, od = C.c_uint32(), C.c_double()
        rc = host_lib().epa_host_ref_in_rtree(self.h, branch, distal, C.byref(ob), C.byref(od))
        if rc == 1:
            return None
        if rc:
            raise RuntimeError(host_lib().epa_host_last_error().decode())
        return ob.value, od.value

    def numbered_newick(self, precision=10):
        buf = C.create_string_buffer(256 * (self.B + 4))
        n = host_lib().epa_host_ref_numbered_newick(self.h, precision, buf, len(buf))
        assert n > 0
        return buf.value.decode()

    def model(self):
        s, c = self.s, self.c
        ev, u, ui, f = np.zeros(s), np.zeros(s * s), np.zeros(s * s), np.zeros(s)
        r, w = np.zeros(c), np.zeros(c)
        host_lib().epa_host_ref_model(self.h, _dp(ev), _dp(u), _dp(ui), _dp(f), _dp(r), _dp(w))
        return {"eigenvals": ev, "u": u.reshape(s, s), "uinv": ui.reshape(s, s), "freqs": f,
                "rates": r, "weights": w}

    def subst(self):
        """exchangeabilities of the model in use (upper triangle, row-major)"""
        out = np.zeros(self.s * (self.s - 1) // 2)
        host_lib().epa_host_ref_subst(self.h, _dp(out))
        return out

    def model_string(self):
        buf = C.create_string_buffer(1 << 14)
        n = host_lib().epa_host_ref_model_string(self.h, buf, len(buf))
        assert n > 0
        return buf.value.decode()

    def tipmap(self):
        buf = np.zeros(256, np.uint32)
        n = host_lib().epa_host_ref_tipmap(self.h, buf.ctypes.data_as(C.POINTER(C.c_uint32)), 256)
        return buf[:n].copy()

    def branch(self, b):
        """-> dict of numpy views (prox_clv [W][c][s], prox_scaler, dist_clv|None,
        dist_tip|None, dist_scaler|None, length), tips expanded by the caller if wanted."""
        p = [C.c_void_p() for _ in range(5)]
        ln = C.c_double()
        host_lib().epa_host_ref_branch(self.h, b, *[C.byref(x) for x in p], C.byref(ln))
        shp = (self.W, self.c, self.s)

        def view(ptr, ctype, shape):
            if not ptr.value:
                return None
            n = int(np.prod(shape))
            return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ctype)), (n,)).reshape(shape)
        return {"prox_clv": view(p[0], C.c_double, shp), "prox_scaler": view(p[1], C.c_uint32, (self.W,)),
                "dist_clv": view(p[2], C.c_double, shp), "dist_tip": view(p[3], C.c_uint8, (self.W,)),
                "dist_scaler": view(p[4], C.c_uint32, (self.W,)), "length": ln.value}

    def evaluator(self, device=0, aa_x_as_n=False, device_precompute=True, rate_scalers=False,
                  raxml_blo=False, newton_variant=0, blo_min_branch=0.0, keep_eigenvalues=False):
        """reference -> GPU -> api.Evaluator (epa_ctx created by the C++ host).  By default the
        tree and the tip sequences are sent and all directional CLVs are computed on the device;
        device_precompute=False uploads the host-computed CLVs instead.  rate_scalers: per-rate
        numerical scaling (EPA_FLAG_RATE_SCALERS); raxml_blo: --raxml-blo (EPA_FLAG_RAXML_BLO);
        newton_variant: bit 0 EPA_FLAG_NEWTON_SLOW_BISECT, bit 1 EPA_FLAG_NEWTON_STRICT_DF;
        blo_min_branch: PLLMOD_OPT_MIN_BRANCH_LEN (0 = default 1e-4); keep_eigenvalues:
        EPA_FLAG_KEEP_EIGENVALUES (the stationary eigenvalue is not snapped to 0)."""
        h = C.c_void_p()
        flags = ((0x2 if rate_scalers else 0) | (0x4 if raxml_blo else 0) | (0x8 if newton_variant & 1 else 0) |
                 (0x10 if newton_variant & 2 else 0) | (0x20 if keep_eigenvalues else 0))
        rc = host_lib().epa_host_dev_create_opts(self.h, device, int(aa_x_as_n), int(device_precompute),
                                                 flags, float(blo_min_branch), C.byref(h))
        if rc:
            raise api.EpaError(rc, host_lib().epa_host_last_error().decode())
        ev = api.Evaluator.__new__(api.Evaluator)
        ev.h, ev.L, ev.B, ev.W, ev.s, ev.c = h, api.dev_lib(), self.B, self.W, self.s, self.c
        ev.last_stats = None
        return ev

    def place_file(self, query_file, outdir, chunk_size=5000, prescoring=True, threshold=0.99999,
                   premasking=True, device=0, invocation="epa_ng_amd"):
        nq, npairs = C.c_uint64(), C.c_uint64()
        rc = host_lib().epa_host_place_file(self.h, query_file.encode(), outdir.encode(), chunk_size,
                                            int(prescoring), threshold, int(premasking), device,
                                            invocation.encode(), C.byref(nq), C.byref(npairs))
        if rc:
            raise RuntimeError(host_lib().epa_host_last_error().decode())
        return nq.value, npairs.value


def filter_lwr(lwr, thresh, acc=False, mn=1, mx=0xffffffff):
    lw = np.ascontiguousarray(lwr, np.float64)
    out = np.zeros(len(lw), np.uint32)
    n = C.c_uint32()
    rc = host_lib().epa_host_filter(_dp(lw), len(lw), thresh, int(acc), mn, mx,
                                    out.ctypes.data_as(C.POINTER(C.c_uint32)), C.byref(n))
    if rc:
        raise RuntimeError(host_lib().epa_host_last_error().decode())
    return out[:n.value]


def parse_model(path):
    """model descriptor from a RAxML 8 info / RAxML-NG .bestModel / IQ-TREE report file"""
    buf = C.create_string_buffer(1 << 16)
    n = host_lib().epa_host_parse_model(str(path).encode(), buf, len(buf))
    if n < 0:
        raise RuntimeError(host_lib().epa_host_last_error().decode())
    return buf.value.decode()


def premask(ref_file, query_file):
    """column mask of the reference's premasking (1 = all-gap in the reference or in the query file)"""
    lib = host_lib()
    lib.epa_host_premask.restype = C.c_long
    buf = np.zeros(1 << 24, np.uint8)
    n = lib.epa_host_premask(str(ref_file).encode(), str(query_file).encode(),
                             buf.ctypes.data_as(C.POINTER(C.c_uint8)), C.c_size_t(buf.size))
    if n < 0:
        raise RuntimeError(lib.epa_host_last_error().decode())
    return buf[:n].copy()


def heuristic(lnl, mode="dynamic", thresh=0.99999):
    lnl = np.ascontiguousarray(lnl, np.float64)
    Q, B = lnl.shape
    cap = Q * B
    pb, ps = np.zeros(cap, np.uint32), np.zeros(cap, np.uint32)
    n = C.c_uint64()
    u32p = C.POINTER(C.c_uint32)
    rc = host_lib().epa_host_heuristic(_dp(lnl), Q, B, {"dynamic": 0, "fixed": 1, "baseball": 2}[mode],
                                       thresh, pb.ctypes.data_as(u32p), ps.ctypes.data_as(u32p), cap,
                                       C.byref(n))
    if rc:
        raise RuntimeError(host_lib().epa_host_last_error().decode())
    return pb[:n.value], ps[:n.value]
