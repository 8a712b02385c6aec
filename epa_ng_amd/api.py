"""Host-side Python surface of the MI355X placement evaluator.

Thin ctypes binding over the C-ABI of include/epa_dev.h (libepa_dev.so, hand-written HIP for
gfx950).  There is NO CPU fallback: if the shared library is missing or no GPU is visible the
calls raise.  Names mirror the reference's operators for this path:
    Evaluator.preplace   <-> place()           src/core/place.cpp:41-95
    Evaluator.thorough   <-> place_thorough()  src/core/place.cpp:97-171
    Evaluator.select     <-> apply_heuristic() src/core/heuristics.hpp:119-127 (dynamic)
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
DEV_SO = os.environ.get("EPA_DEV_SO", os.path.join(HERE, "libepa_dev.so"))

__all__ = ["EpaError", "dev_lib", "device_count", "encode_queries", "Evaluator", "PAIR_DTYPE",
           "RESULT_DTYPE", "ROW_DTYPE", "DEV_SO", "Packed4", "pack_codes_4bit", "unpack_codes_4bit", "Comm",
           "comm_unique_id", "mapped_rccl_path", "comm_set_library", "comm_library_path", "comm_set_default_timeout",
           "pci_id_str"]

PAIR_DTYPE = np.dtype([("branch_id", np.uint32), ("seq_id", np.uint32)])
RESULT_DTYPE = np.dtype([("lnl", np.float64), ("pendant_length", np.float64),
                         ("distal_length", np.float64)])

# one gathered row of the multi-GPU exchange (include/epa_dev.h: epa_row; seq_id is global)
ROW_DTYPE = np.dtype([("branch_id", np.uint32), ("seq_id", np.uint32), ("lnl", np.float64),
                      ("pendant_length", np.float64), ("distal_length", np.float64)])


class EpaError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("epa_dev error %d: %s" % (code, msg))
        self.code = code


class _RefDesc(C.Structure):
    _fields_ = [("states", C.c_uint32), ("rate_cats", C.c_uint32), ("sites", C.c_uint32),
                ("branches", C.c_uint32),
                ("eigenvals", C.c_void_p), ("eigenvecs_u", C.c_void_p),
                ("eigenvecs_uinv", C.c_void_p), ("freqs", C.c_void_p), ("rates", C.c_void_p),
                ("rate_weights", C.c_void_p), ("prop_invar", C.c_double),
                ("prox_clv", C.c_void_p), ("prox_scaler", C.c_void_p), ("dist_clv", C.c_void_p),
                ("dist_tipchars", C.c_void_p), ("dist_scaler", C.c_void_p),
                ("branch_length", C.c_void_p),
                ("tipmap", C.c_void_p), ("tipmap_size", C.c_uint32),
                ("blo_min_branch", C.c_double), ("blo_max_branch", C.c_double),
                ("blo_default_branch", C.c_double), ("blo_epsilon", C.c_double),
                ("pendant_default", C.c_double), ("blo_max_rounds", C.c_uint32),
                ("blo_max_newton", C.c_uint32), ("flags", C.c_uint32), ("aa_x_as_n", C.c_uint32),
                ("invariant_state", C.c_void_p)]


class _Stats(C.Structure):
    _fields_ = [("pairs", C.c_uint64), ("rounds", C.c_uint64), ("newton_evals", C.c_uint64),
                ("reverts", C.c_uint64)]


_LIB = None


def dev_lib():
    """Loads libepa_dev.so (fails loudly when the HIP extension has not been built)."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(DEV_SO):
            raise ImportError("libepa_dev.so is missing: run `python __graft_entry__.py` "
                              "(build()) -- the HIP extension is the only compute path")
        # When PyTorch shares the process (bench.py, tests: device buffers, torch.distributed) its
        # bundled HIP runtime must be the first one loaded, otherwise torch later finds "No HIP
        # GPUs": import it before dlopen()ing the extension.  torch is plumbing, never compute.
        try:
            import torch  # noqa: F401
            if torch.cuda.is_available():
                torch.cuda.init()
        except ImportError:
            pass
        L = C.CDLL(DEV_SO)
        L.epa_dev_device_count.restype = C.c_int
        L.epa_dev_create.argtypes = [C.POINTER(_RefDesc), C.c_int, C.POINTER(C.c_void_p)]
        L.epa_dev_destroy.argtypes = [C.c_void_p]
        L.epa_dev_last_error.restype = C.c_char_p
        L.epa_dev_last_error.argtypes = [C.c_void_p]
        L.epa_dev_set_stream.argtypes = [C.c_void_p, C.c_void_p]
        L.epa_dev_build_lookup.argtypes = [C.c_void_p]
        L.epa_dev_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
        L.epa_encode_queries.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32,
                                         C.POINTER(C.c_char_p), C.c_int, C.c_int, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32)]
        L.epa_encode_queries_compact.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32,
                                                 C.POINTER(C.c_char_p), C.c_int, C.c_int, C.c_uint32,
                                                 C.c_void_p, C.c_void_p, C.c_void_p,
                                                 C.POINTER(C.c_uint32)]
        L.epa_dev_set_query_layout.argtypes = [C.c_void_p, C.c_uint32]
        L.epa_dev_preplace.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32,
                                       C.c_void_p]
        L.epa_dev_thorough.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_uint32, C.c_void_p, C.POINTER(_Stats)]
        L.epa_dev_set_heuristic.argtypes = [C.c_void_p, C.c_int, C.c_double]
        L.epa_dev_set_query_packing.argtypes = [C.c_void_p, C.c_int]
        L.epa_pack_codes_4bit.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
        L.epa_unpack_codes_4bit.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
        L.epa_dev_select_candidates.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_double,
                                                C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
        L.epa_dev_place_chunk.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32,
                                          C.c_uint32, C.c_double, C.c_void_p, C.c_void_p,
                                          C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(_Stats)]
        L.epa_dev_chunk_stage.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
        L.epa_dev_chunk_launch.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_double, C.c_void_p,
                                           C.c_void_p, C.c_uint64, C.c_uint32]
        L.epa_dev_chunk_launch_begin.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_double, C.c_void_p,
                                           C.c_void_p, C.c_uint64, C.c_uint32]
        L.epa_dev_chunk_launch_end.argtypes = [C.c_void_p, C.c_int]
        L.epa_dev_chunk_launch_many_begin.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.c_int, C.c_uint32, C.c_double,
                                                      C.c_uint64, C.c_uint32]
        L.epa_dev_chunk_launch_many.argtypes = L.epa_dev_chunk_launch_many_begin.argtypes
        L.epa_dev_chunk_finish.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p),
                                           C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.POINTER(_Stats)]
        L.epa_dev_tree_logl.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_double)]
        L.epa_dev_place_all.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32,
                                        C.c_uint32, C.c_double, C.c_int, C.c_uint32, C.c_uint32,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.POINTER(_Stats)]
        L.epa_comm_get_unique_id.argtypes = [C.c_void_p]
        L.epa_comm_create.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_uint32, C.c_int,
                                      C.POINTER(C.c_void_p)]
        L.epa_comm_destroy.argtypes = [C.c_void_p]
        L.epa_comm_destroy.restype = None
        L.epa_dev_gather_results.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64,
                                             C.c_uint32, C.POINTER(C.c_uint64)]
        L.epa_dev_gather_slot.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_uint32, C.POINTER(C.c_uint64)]
        L.epa_comm_collect.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_void_p), C.POINTER(C.c_uint32),
                                       C.POINTER(C.c_uint64)]
        L.epa_comm_flush.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]
        L.epa_comm_carried_rows.argtypes = [C.c_void_p]
        L.epa_comm_carried_rows.restype = C.c_uint64
        L.epa_dev_gather_rows.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
        L.epa_dev_place_all_rows.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_double,
                                             C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p),
                                             C.POINTER(C.c_uint64), C.c_void_p]
        L.epa_comm_abort.argtypes = [C.c_void_p]
        L.epa_comm_abort.restype = None
        L.epa_comm_set_library.argtypes = [C.c_char_p]
        L.epa_comm_library_path.restype = C.c_char_p
        L.epa_comm_set_timeout.argtypes = [C.c_void_p, C.c_double]
        L.epa_comm_probe.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.POINTER(C.c_uint64)]
        L.epa_comm_set_self_send.argtypes = [C.c_void_p, C.c_int]
        L.epa_comm_device_rows.argtypes = [C.c_void_p, C.c_uint64, C.c_int]
        L.epa_comm_device_rows.restype = C.c_void_p
        L.epa_dev_mem_info.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.epa_dev_xcd_shares.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
        L.epa_dev_xcd_shares.restype = C.c_int
        L.epa_dev_last_kernel_ms.restype = C.c_double
        L.epa_dev_last_sclk_mhz.argtypes = [C.c_void_p]
        L.epa_dev_last_sclk_mhz.restype = C.c_double
        L.epa_dev_last_kernel_ms.argtypes = [C.c_void_p, C.c_char_p]
        _LIB = L
    return _LIB


def device_count():
    return dev_lib().epa_dev_device_count()


class Packed4:
    """code rows in the 4-bit wire format (epa_pack_codes_4bit): `data` is [Q][(stride + 1) // 2]
    (numpy on the host or a torch cuda tensor), `stride` the row length of the unpacked layout"""

    def __init__(self, data, stride):
        self.data, self.stride = data, int(stride)


def pack_codes_4bit(codes):
    """[Q][stride] nucleotide codes -> Packed4 (two codes per byte, earlier site in the high nibble)"""
    codes = np.ascontiguousarray(codes, np.uint8)
    Q, stride = codes.shape
    out = np.zeros((Q, (stride + 1) // 2), np.uint8)
    rc = dev_lib().epa_pack_codes_4bit(codes.ctypes.data, Q, stride, out.ctypes.data)
    if rc:
        raise EpaError(rc, "pack_codes_4bit: a code does not fit four bits")
    return Packed4(out, stride)


def unpack_codes_4bit(p):
    Q = p.data.shape[0]
    out = np.zeros((Q, p.stride), np.uint8)
    dev_lib().epa_unpack_codes_4bit(np.ascontiguousarray(p.data).ctypes.data, Q, p.stride, out.ctypes.data)
    return out


def _ptr(a):
    """numpy array -> host pointer; torch tensor / int -> raw (device) pointer"""
    if a is None:
        return None
    if isinstance(a, Packed4):
        return _ptr(a.data)
    if isinstance(a, np.ndarray):
        assert a.flags["C_CONTIGUOUS"]
        return a.ctypes.data
    if isinstance(a, int):
        return a
    return a.data_ptr()  # torch tensor


def encode_queries(states, seqs, premasking=True, aa_x_as_n=False, compact=False):
    """ASCII query rows -> (codes uint8, win_begin uint32 [Q], win_span uint32 [Q]).
    codes is [Q][W] (the aligned rows), or with compact=True [Q][stride] holding only each
    query's window (stride = longest window rounded up to 16; see epa_encode_queries_compact).
    The Evaluator methods recognise the layout by the row length."""
    Q = len(seqs)
    W = len(seqs[0])
    arr = (C.c_char_p * Q)()
    arr[:] = [s if isinstance(s, bytes) else s.encode() for s in seqs]
    for s in arr:
        if len(s) != W:
            raise EpaError(-4, "Query sequence length not same as reference alignment!")
    wb = np.zeros(Q, np.uint32)
    ws = np.zeros(Q, np.uint32)
    bad = C.c_uint32(0)
    L = dev_lib()

    def check(rc):
        if rc:
            raise EpaError(rc, "query %d: %s" % (bad.value, "char is invalid!" if rc == -6 else
                                                 "does not appear to have any non-gap sites!"))
    if not compact:
        codes = np.zeros((Q, W), np.uint8)
        check(L.epa_encode_queries(states, W, Q, arr, int(premasking), int(aa_x_as_n),
                                   codes.ctypes.data, wb.ctypes.data, ws.ctypes.data, C.byref(bad)))
        return codes, wb, ws
    check(L.epa_encode_queries_compact(states, W, Q, arr, int(premasking), int(aa_x_as_n), 0, None,
                                       wb.ctypes.data, ws.ctypes.data, C.byref(bad)))
    stride = (int(ws.max()) + 15) // 16 * 16
    if stride == W:       # keep the two layouts distinguishable by their row length
        stride += 16
    codes = np.zeros((Q, stride), np.uint8)
    check(L.epa_encode_queries_compact(states, W, Q, arr, int(premasking), int(aa_x_as_n), stride,
                                       codes.ctypes.data, wb.ctypes.data, ws.ctypes.data, C.byref(bad)))
    return codes, wb, ws


class Evaluator:
    """One epa_ctx (one GPU).  Reference-side inputs per branch, libpll layouts:
    prox_clv[b], dist_clv[b]: float64 [W][c][s] (dist_clv[b] may be None when dist_tip[b] is a
    uint8 [W] tip-code row), prox_scaler[b] / dist_scaler[b]: uint32 [W] or None."""

    def __init__(self, states, rates, weights, eigenvals, u, uinv, freqs, branch_length,
                 prox_clv, dist_clv, prox_scaler=None, dist_scaler=None, dist_tip=None,
                 tipmap=None, device=0, aa_x_as_n=False, pinv=0.0, invariant_state=None, flags=0):
        """flags: EPA_FLAG_* of include/epa_dev.h (0x2 = per-rate scalers: the scaler rows are then uint32 [W][c])"""
        L = dev_lib()
        B = len(branch_length)
        self.B, self.s, self.c = B, states, len(rates)
        W = None
        for a in list(prox_clv):
            W = a.shape[0]
            break
        self.W = W
        f64 = lambda a: np.ascontiguousarray(a, dtype=np.float64)  # noqa: E731
        keep = [f64(eigenvals), f64(u), f64(uinv), f64(freqs), f64(rates), f64(weights),
                f64(branch_length)]

        def ptr_array(items, dtype):
            arr = (C.c_void_p * B)()
            held = []
            for i, it in enumerate(items if items is not None else [None] * B):
                if it is None:
                    arr[i] = None
                else:
                    a = np.ascontiguousarray(it, dtype=dtype)
                    held.append(a)
                    arr[i] = a.ctypes.data
            return arr, held

        pc, h1 = ptr_array(prox_clv, np.float64)
        dc, h2 = ptr_array(dist_clv, np.float64)
        ps, h3 = ptr_array(prox_scaler, np.uint32)
        ds, h4 = ptr_array(dist_scaler, np.uint32)
        dt, h5 = ptr_array(dist_tip, np.uint8)
        tm = np.ascontiguousarray(tipmap, dtype=np.uint32) if tipmap is not None else None
        d = _RefDesc()
        d.states, d.rate_cats, d.sites, d.branches = states, self.c, W, B
        d.eigenvals, d.eigenvecs_u, d.eigenvecs_uinv = (keep[0].ctypes.data, keep[1].ctypes.data,
                                                       keep[2].ctypes.data)
        d.freqs, d.rates, d.rate_weights = keep[3].ctypes.data, keep[4].ctypes.data, keep[5].ctypes.data
        d.prop_invar = float(pinv)
        inv = None
        if invariant_state is not None:
            inv = np.ascontiguousarray(invariant_state, dtype=np.int8)
            d.invariant_state = inv.ctypes.data
        d.prox_clv = C.cast(pc, C.c_void_p)
        d.prox_scaler = C.cast(ps, C.c_void_p)
        d.dist_clv = C.cast(dc, C.c_void_p)
        d.dist_tipchars = C.cast(dt, C.c_void_p) if dist_tip is not None else None
        d.dist_scaler = C.cast(ds, C.c_void_p)
        d.branch_length = keep[6].ctypes.data
        if tm is not None:
            d.tipmap, d.tipmap_size = tm.ctypes.data, len(tm)
        d.aa_x_as_n = int(aa_x_as_n)
        d.flags = int(flags)
        h = C.c_void_p()
        rc = L.epa_dev_create(C.byref(d), device, C.byref(h))
        if rc:
            raise EpaError(rc, L.epa_dev_last_error(None).decode())
        self.h = h
        self.L = L
        self.last_stats = None

    def close(self):
        if getattr(self, "h", None):
            self.L.epa_dev_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def _check(self, rc):
        if rc:
            raise EpaError(rc, self.L.epa_dev_last_error(self.h).decode())

    def set_stream(self, stream_ptr):
        self._check(self.L.epa_dev_set_stream(self.h, stream_ptr))

    def set_option(self, key, value=1):
        """diagnostic switch of this context (include/epa_dev.h epa_dev_set_option)"""
        self._check(self.L.epa_dev_set_option(self.h, key.encode(), int(value)))

    def build_lookup(self):
        self._check(self.L.epa_dev_build_lookup(self.h))

    def _layout(self, codes):
        """tells the context whether `codes` rows are aligned rows (W bytes) or compact windows"""
        packed = isinstance(codes, Packed4)
        row = codes.stride if packed else (int(codes.shape[1]) if len(codes.shape) == 2 else self.W)
        self._check(self.L.epa_dev_set_query_layout(self.h, 0 if row == self.W else row))
        self._check(self.L.epa_dev_set_query_packing(self.h, 4 if packed else 8))

    def preplace(self, codes, win_begin, win_span, Q=None, out=None):
        """-> lnl [Q][B].  Inputs numpy (host) or torch cuda tensors (HBM-resident)."""
        Q = len(win_begin) if Q is None else Q
        if out is None:
            out = np.empty((Q, self.B), np.float64)
        self._layout(codes)
        self._check(self.L.epa_dev_preplace(self.h, _ptr(codes), _ptr(win_begin), _ptr(win_span),
                                            Q, _ptr(out)))
        return out

    def thorough(self, pairs, codes, win_begin, win_span, Q=None, n_pairs=None, out=None):
        """pairs: structured PAIR_DTYPE array (or device buffer) -> RESULT_DTYPE array."""
        Q = len(win_begin) if Q is None else Q
        n = len(pairs) if n_pairs is None else n_pairs
        if out is None:
            out = np.empty(n, RESULT_DTYPE)
        st = _Stats()
        self._layout(codes)
        self._check(self.L.epa_dev_thorough(self.h, _ptr(pairs), n, _ptr(codes), _ptr(win_begin),
                                            _ptr(win_span), Q, _ptr(out), C.byref(st)))
        self.last_stats = {"pairs": st.pairs, "rounds": st.rounds,
                           "newton_evals": st.newton_evals, "reverts": st.reverts}
        return out

    def set_heuristic(self, mode="dynamic", param=0.0):
        """selection rule of select() / place_chunk(): "dynamic" (the call's threshold), "fixed"
        (param = fraction of the branches) or "baseball" (epa_dev_set_heuristic)"""
        self._check(self.L.epa_dev_set_heuristic(self.h, {"dynamic": 0, "fixed": 1, "baseball": 2}[mode],
                                                 float(param)))

    def select(self, lnl, Q, threshold=0.99999, max_pairs=None, out=None):
        """dynamic heuristic -> branch-major sorted PAIR_DTYPE array"""
        if max_pairs is None:
            max_pairs = Q * self.B
        host_out = out is None
        if host_out:
            out = np.empty(max_pairs, PAIR_DTYPE)
        n = C.c_uint64(0)
        self._check(self.L.epa_dev_select_candidates(self.h, _ptr(lnl), Q, threshold, _ptr(out),
                                                     max_pairs, C.byref(n)))
        return out[:n.value] if host_out else n.value

    def place_chunk(self, codes, win_begin, win_span, Q=None, threshold=0.99999, max_span=0,
                    max_pairs=None, pairs_out=None, results_out=None):
        """preplace -> dynamic heuristic -> thorough, fused on device (epa_dev_place_chunk).
        Returns (pairs, results) numpy arrays, or n_pairs when device buffers are supplied."""
        Q = len(win_begin) if Q is None else Q
        if max_pairs is None:
            max_pairs = Q * 64
        host = pairs_out is None
        if host:
            pairs_out = np.empty(max_pairs, PAIR_DTYPE)
            results_out = np.empty(max_pairs, RESULT_DTYPE)
        n = C.c_uint64(0)
        st = _Stats()
        self._layout(codes)
        self._check(self.L.epa_dev_place_chunk(self.h, _ptr(codes), _ptr(win_begin), _ptr(win_span),
                                               Q, max_span, threshold, _ptr(pairs_out),
                                               _ptr(results_out), max_pairs, C.byref(n),
                                               C.byref(st)))
        self.last_stats = {"pairs": st.pairs, "rounds": st.rounds,
                           "newton_evals": st.newton_evals, "reverts": st.reverts}
        if host:
            return pairs_out[:n.value], results_out[:n.value]
        return n.value

    # ---- double-buffered chunk pipeline (epa_dev_chunk_stage / _launch / _finish): the upload of
    # chunk k+1 and the download of chunk k-1 overlap the kernels of chunk k
    def chunk_stage(self, slot, codes, win_begin, win_span):
        """HOST arrays (numpy; codes may be Packed4) -> pinned buffer -> async H2D; returns at once.
        torch cuda tensors (all three) are read in place: keep them untouched until chunk_finish"""
        self._layout(codes)
        data = codes.data if isinstance(codes, Packed4) else codes
        assert isinstance(data, np.ndarray) == isinstance(win_begin, np.ndarray) == isinstance(win_span, np.ndarray)
        self._check(self.L.epa_dev_chunk_stage(self.h, slot, _ptr(codes), _ptr(win_begin), _ptr(win_span),
                                               len(win_begin)))

    @staticmethod
    def _chunk_flags(keep_on_device, host_ordered):
        # EPA_CHUNK_NO_D2H | EPA_CHUNK_HOST_ORDERED (include/epa_dev.h): host_ordered = the caller touches staged-in-place
        # inputs / device-resident results only before launch / after finish, no stream-level ordering wanted
        return (1 if keep_on_device else 0) | (2 if host_ordered else 0)

    def chunk_launch(self, slot, threshold=0.99999, max_span=0, max_pairs=None, pairs_out=None,
                     results_out=None, keep_on_device=False, host_ordered=False):
        """preplace -> heuristic -> thorough of the staged chunk; blocks only until the candidate
        count is known.  pairs_out / results_out: optional device buffers (torch) of max_pairs rows"""
        assert max_pairs is not None
        self._check(self.L.epa_dev_chunk_launch(self.h, slot, max_span, threshold, _ptr(pairs_out),
                                                _ptr(results_out), max_pairs, self._chunk_flags(keep_on_device, host_ordered)))

    def chunk_launch_begin(self, slot, threshold=0.99999, max_span=0, max_pairs=None, pairs_out=None,
                           results_out=None, keep_on_device=False, host_ordered=False):
        """first half of chunk_launch: preplacement + candidate selection queued on the slot's own
        stream, returns without waiting"""
        assert max_pairs is not None
        self._check(self.L.epa_dev_chunk_launch_begin(self.h, slot, max_span, threshold, _ptr(pairs_out),
                                                      _ptr(results_out), max_pairs, self._chunk_flags(keep_on_device, host_ordered)))

    def chunk_launch_many_begin(self, slots, threshold=0.99999, max_span=0, max_pairs=None, keep_on_device=False,
                                host_ordered=False):
        """group launch of several STAGED slots (one chunk body over their concatenated queries); slots[0] leads:
        chunk_launch_end(slots[0]), then chunk_finish(slot) for every member"""
        assert max_pairs is not None
        arr = (C.c_int * len(slots))(*slots)
        self._check(self.L.epa_dev_chunk_launch_many_begin(self.h, arr, len(slots), max_span, threshold, max_pairs,
                                                           self._chunk_flags(keep_on_device, host_ordered)))

    def chunk_launch_many(self, slots, **kw):
        self.chunk_launch_many_begin(slots, **kw)
        self.chunk_launch_end(slots[0])

    def chunk_launch_end(self, slot):
        """second half: waits for the candidate count, queues the thorough kernels + result D2H"""
        self._check(self.L.epa_dev_chunk_launch_end(self.h, slot))

    def chunk_finish(self, slot, copy=True):
        """waits for the slot's results -> (pairs, results) numpy views of the slot's pinned buffer
        (copy=False: valid until the slot is staged again), or n_pairs when the launch kept the
        results on the device"""
        pp, pr, n, st = C.c_void_p(), C.c_void_p(), C.c_uint64(0), _Stats()
        rc = self.L.epa_dev_chunk_finish(self.h, slot, C.byref(pp), C.byref(pr), C.byref(n), C.byref(st))
        self.last_stats = {"pairs": st.pairs, "rounds": st.rounds,
                           "newton_evals": st.newton_evals, "reverts": st.reverts}
        self._check(rc)
        n = n.value
        import ctypes
        if n == 0:
            return np.zeros(0, PAIR_DTYPE), np.zeros(0, RESULT_DTYPE)
        pairs = np.ctypeslib.as_array(ctypes.cast(pp, C.POINTER(C.c_uint32)), (2 * n,)).view(PAIR_DTYPE)
        res = np.ctypeslib.as_array(ctypes.cast(pr, C.POINTER(C.c_double)), (3 * n,)).view(RESULT_DTYPE)
        return (pairs.copy(), res.copy()) if copy else (pairs, res)

    def chunk_finish_device(self, slot):
        """finish() of a launch with keep_on_device=True -> n_pairs (results are in the buffers
        handed to chunk_launch)"""
        n, st = C.c_uint64(0), _Stats()
        rc = self.L.epa_dev_chunk_finish(self.h, slot, None, None, C.byref(n), C.byref(st))
        self.last_stats = {"pairs": st.pairs, "rounds": st.rounds,
                           "newton_evals": st.newton_evals, "reverts": st.reverts}
        self._check(rc)
        return n.value

    def place_all(self, codes, win_begin, win_span, Q=None, min_lwr=0.01, acc=False, filter_min=1,
                  filter_max=7, max_span=0):
        """--no-heur on device: thorough placement on every branch, LWR over all of them, filter.
        Returns a list (one entry per query) of (branch_id, lnl, pendant, distal, lwr) arrays,
        best placement first."""
        Q = len(win_begin) if Q is None else Q
        pairs = np.zeros(Q * filter_max, PAIR_DTYPE)
        res = np.zeros(Q * filter_max, RESULT_DTYPE)
        lwr = np.zeros(Q * filter_max, np.float64)
        counts = np.zeros(Q, np.uint32)
        st = _Stats()
        self._layout(codes)
        self._check(self.L.epa_dev_place_all(self.h, _ptr(codes), _ptr(win_begin), _ptr(win_span), Q,
                                             max_span, min_lwr, int(acc), filter_min, filter_max,
                                             _ptr(pairs), _ptr(res), _ptr(lwr), _ptr(counts), C.byref(st)))
        self.last_stats = {"pairs": st.pairs, "rounds": st.rounds,
                           "newton_evals": st.newton_evals, "reverts": st.reverts}
        out = []
        for q in range(Q):
            sl = slice(q * filter_max, q * filter_max + int(counts[q]))
            assert np.all(pairs["seq_id"][sl] == q)
            out.append((pairs["branch_id"][sl].copy(), res["lnl"][sl].copy(), res["pendant_length"][sl].copy(),
                        res["distal_length"][sl].copy(), lwr[sl].copy()))
        return out

    def tree_logl(self, branch=0):
        """log-likelihood of the reference tree evaluated on the device at `branch`"""
        v = C.c_double(0.0)
        self._check(self.L.epa_dev_tree_logl(self.h, branch, C.byref(v)))
        return v.value

    def xcd_shares(self):
        """shares of a Newton launch's pair list the eight XCDs currently take (epa_dev_xcd_shares)"""
        out = (C.c_double * 8)()
        self._check(self.L.epa_dev_xcd_shares(self.h, out))
        return np.array(list(out))

    def sclk_mhz(self):
        """shader clock (MHz) the last single-class Newton launch ran at (epa_dev_last_sclk_mhz)"""
        return float(self.L.epa_dev_last_sclk_mhz(self.h))

    def kernel_ms(self, which):
        return self.L.epa_dev_last_kernel_ms(self.h, which.encode())


def comm_unique_id():
    """rank 0: the 128 bytes every rank passes to Comm() (hand them over out of band)"""
    L = dev_lib()
    buf = C.create_string_buffer(128)
    rc = L.epa_comm_get_unique_id(buf)
    if rc:
        raise EpaError(rc, (L.epa_dev_last_error(None) or b"").decode())
    return buf.raw


def mapped_rccl_path():
    """the librccl this process has ALREADY mapped (PyTorch's torch/lib/librccl.so once torch.distributed's nccl
    backend is up), or None -- what comm_set_library() should be given so that the product binds the same copy"""
    try:
        with open("/proc/self/maps") as f:
            for line in f:
                i = line.find("/")
                if i >= 0 and os.path.basename(line[i:].strip()).startswith("librccl.so"):
                    return line[i:].strip()
    except OSError:
        pass
    return None


def comm_set_library(path):
    """the transport library of every later Comm (before the first one): epa_comm_set_library"""
    L = dev_lib()
    rc = L.epa_comm_set_library(path.encode() if path else None)
    if rc:
        raise EpaError(rc, (L.epa_dev_last_error(None) or b"").decode())


def comm_library_path():
    """the file the product bound its ten RCCL entry points from ('' if none could be loaded)"""
    return (dev_lib().epa_comm_library_path() or b"").decode()


def comm_set_default_timeout(seconds):
    """process default of every communicator's host-side waits, incl. ncclCommInitRank inside Comm()"""
    dev_lib().epa_comm_set_timeout(None, float(seconds))


def pci_id_str(v):
    return "%04x:%02x:%02x" % (v >> 16, (v >> 8) & 0xff, v & 0xff)


class Comm:
    """The product library's RCCL gather of (pair, result) rows to rank 0 (include/epa_dev.h, epa_comm_*;
    epa_ng_amd/csrc/comm.hip).  One per process / Evaluator."""

    def __init__(self, ev, unique_id, rank, world, rows_cap, depth=2, self_send=False):
        self.ev, self.L, self.rank, self.world, self.rows_cap, self.depth = ev, ev.L, rank, world, rows_cap, depth
        h = C.c_void_p()
        ev._check(self.L.epa_comm_create(ev.h, unique_id, rank, world, rows_cap, depth, C.byref(h)))
        self.h = h
        if self_send:
            ev._check(self.L.epa_comm_set_self_send(self.h, 1))

    def set_timeout(self, seconds):
        self.ev._check(self.L.epa_comm_set_timeout(self.h, float(seconds)))

    def probe(self, timeout_s=60.0):
        """collective handshake (epa_comm_probe); rank 0 -> the PCI ids of every rank's device, others -> None"""
        ids = (C.c_uint64 * self.world)()
        self.ev._check(self.L.epa_comm_probe(self.ev.h, self.h, float(timeout_s), ids))
        return [pci_id_str(int(x)) for x in ids] if self.rank == 0 else None

    def close(self):
        if getattr(self, "h", None):
            self.L.epa_comm_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def post(self, pairs, results, n, seq_offset=0):
        """pairs / results: DEVICE buffers (torch tensors or raw pointers) of >= n rows -> ticket"""
        t = C.c_uint64(0)
        self.ev._check(self.L.epa_dev_gather_results(self.ev.h, self.h, _ptr(pairs), _ptr(results), n, seq_offset,
                                                     C.byref(t)))
        return t.value

    def post_slot(self, slot, seq_offset=0):
        """a chunk slot launched with keep_on_device=True, after chunk_launch_end -> ticket"""
        t = C.c_uint64(0)
        self.ev._check(self.L.epa_dev_gather_slot(self.ev.h, self.h, slot, seq_offset, C.byref(t)))
        return t.value

    def abort(self):
        """a failing rank's way out: ncclCommAbort + release without waiting for the peers"""
        if getattr(self, "h", None):
            self.L.epa_comm_abort(self.h)
            self.h = None

    def collect_counts(self, ticket):
        """rank 0: waits for gather `ticket`; -> per-rank valid row counts.  The rows stay in HBM
        (device_rows)."""
        cnt = (C.c_uint32 * self.world)()
        pend = (C.c_uint64 * self.world)()
        self.ev._check(self.L.epa_comm_collect(self.h, ticket, None, cnt, pend))
        self.last_pending = [int(x) for x in pend]
        return [int(x) for x in cnt]

    def device_rows(self, ticket, rank):
        """device address of rank's row block of gather `ticket` on rank 0 (None once the slot is reused)"""
        return self.L.epa_comm_device_rows(self.h, ticket, rank)

    def collect(self, ticket):
        """rank 0: list (one ROW_DTYPE array per rank) of gather `ticket`"""
        import ctypes
        ptrs = (C.c_void_p * self.world)()
        cnt = (C.c_uint32 * self.world)()
        pend = (C.c_uint64 * self.world)()
        self.ev._check(self.L.epa_comm_collect(self.h, ticket, ptrs, cnt, pend))
        self.last_pending = [int(x) for x in pend]
        out = []
        for r in range(self.world):
            if cnt[r] == 0:
                out.append(np.zeros(0, ROW_DTYPE))
                continue
            raw = np.ctypeslib.as_array(ctypes.cast(ptrs[r], C.POINTER(C.c_uint8)), (32 * cnt[r],))
            out.append(raw.view(ROW_DTYPE).copy())
        return out

    def flush(self, on_ticket=None):
        """collective: drains carried rows with extra gathers (rounds of <= depth, agreed on by an all-reduce
        each); on_ticket(t) is called for every extra gather as soon as it may be collected (rank 0 must
        collect inside it).  -> list of the extra tickets"""
        out = []
        while True:
            first, n = C.c_uint64(0), C.c_uint32(0)
            self.ev._check(self.L.epa_comm_flush(self.ev.h, self.h, C.byref(first), C.byref(n)))
            if n.value == 0:
                return out
            for t in range(first.value, first.value + n.value):
                out.append(t)
                if on_ticket is not None:
                    on_ticket(t)

    @property
    def carried_rows(self):
        return int(self.L.epa_comm_carried_rows(self.h))
