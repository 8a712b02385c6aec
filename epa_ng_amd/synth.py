"""Deterministic synthetic workloads for the benchmark / parity configs of BASELINE.json
(SURVEY.md section 8d): random-join reference tree, MSA simulated down that tree under the
model (so multi-round branch-length optimisation is exercised), and short reads cut from tip
sequences with substitutions.  Pure numpy; data generation only, no likelihood code."""
import math

import numpy as np

DNA = "ACGT"
AA = "ARNDCQEGHILKMFPSTWYV"

# model string pinned by the reference's test/src/parse_model.cpp:10-11
CFG2_SUBST = [0.787874, 1.821672, 1.294006, 0.698421, 3.034135, 1.0]
CFG2_FREQS = [0.256465, 0.222535, 0.308594, 0.212406]
CFG2_ALPHA = 0.478218


def rate_matrix(subst, freqs):
    s = len(freqs)
    f = np.asarray(freqs, float)
    R = np.zeros((s, s))
    R[np.triu_indices(s, 1)] = subst
    R = R + R.T
    Q = R * f[None, :]
    np.fill_diagonal(Q, 0.0)
    np.fill_diagonal(Q, -Q.sum(1))
    Q /= -(f * np.diag(Q)).sum()
    return Q


def pmatrix(Q, freqs, t):
    """exp(Qt) through the symmetrised eigen-decomposition (data generation only)."""
    sq = np.sqrt(np.asarray(freqs, float))
    A = Q * sq[:, None] / sq[None, :]
    A = 0.5 * (A + A.T)
    w, V = np.linalg.eigh(A)
    P = (V * np.exp(w * t)[None, :]) @ V.T
    P = P * sq[None, :] / sq[:, None]
    P = np.clip(P, 0.0, None)
    return P / P.sum(1, keepdims=True)


class Node:
    __slots__ = ("kids", "label", "length")

    def __init__(self, label=None):
        self.kids, self.label, self.length = [], label, 0.0


def random_tree(n_tips, seed, mean_bl=0.05, lo=1e-4, hi=1.0):
    """random-join topology; branch lengths Exp(mean) clamped to [lo, hi]; top trifurcation."""
    rng = np.random.RandomState(seed)
    pool = [Node("t%d" % i) for i in range(n_tips)]
    for nd in pool:
        nd.length = float(min(max(rng.exponential(mean_bl), lo), hi))
    while len(pool) > 3:
        i, j = sorted(rng.choice(len(pool), 2, replace=False))
        b = pool.pop(j)
        a = pool.pop(i)
        p = Node()
        p.kids = [a, b]
        p.length = float(min(max(rng.exponential(mean_bl), lo), hi))
        pool.append(p)
    root = Node()
    root.kids = pool
    return root


def newick(root):
    out = []

    def rec(n):
        if n.kids:
            out.append("(")
            for i, k in enumerate(n.kids):
                if i:
                    out.append(",")
                rec(k)
            out.append(")")
        else:
            out.append(n.label)
        if n is not root:
            out.append(":%r" % n.length)

    import sys
    sys.setrecursionlimit(max(10000, sys.getrecursionlimit()))
    rec(root)
    out.append(";")
    return "".join(out)


def simulate_msa(root, W, subst, freqs, cat_rates, seed):
    """-> (labels, sequences) for the tips, simulated down the tree."""
    rng = np.random.RandomState(seed)
    s = len(freqs)
    alphabet = DNA if s == 4 else AA
    Q = rate_matrix(subst, freqs)
    cats = rng.randint(0, len(cat_rates), W)
    root_states = rng.choice(s, W, p=np.asarray(freqs) / np.sum(freqs))
    labels, seqs = [], []
    stack = [(root, root_states)]
    while stack:
        node, st = stack.pop()
        if not node.kids:
            labels.append(node.label)
            seqs.append("".join(alphabet[i] for i in st))
            continue
        for k in node.kids:
            child = np.empty(W, np.int64)
            u = rng.random_sample(W)
            for c, r in enumerate(cat_rates):
                idx = np.nonzero(cats == c)[0]
                if not len(idx):
                    continue
                cdf = np.cumsum(pmatrix(Q, freqs, k.length * r), axis=1)
                child[idx] = (u[idx, None] > cdf[st[idx]]).sum(1).clip(0, s - 1)
            stack.append((k, child))
    order = np.argsort([int(l[1:]) for l in labels])
    return [labels[i] for i in order], [seqs[i] for i in order]


def make_reads(seqs, n_reads, read_len, sub_rate, seed, states=4):
    """reads = read_len consecutive columns of a random tip sequence, sub_rate random
    substitutions, every other column '-'."""
    rng = np.random.RandomState(seed)
    alphabet = DNA if states == 4 else AA
    W = len(seqs[0])
    tips = rng.randint(0, len(seqs), n_reads)
    starts = rng.randint(0, W - read_len + 1, n_reads)
    arr = np.frombuffer("".join(seqs).encode(), dtype=np.uint8).reshape(len(seqs), W)
    out = np.full((n_reads, W), ord("-"), np.uint8)
    alpha = np.frombuffer(alphabet.encode(), dtype=np.uint8)
    cols = starts[:, None] + np.arange(read_len)[None, :]
    frag = arr[tips[:, None], cols]
    mut = rng.random_sample(frag.shape) < sub_rate
    frag = np.where(mut, alpha[rng.randint(0, len(alpha), frag.shape)], frag)
    out[np.arange(n_reads)[:, None], cols] = frag
    return [row.tobytes().decode() for row in out], starts


def gamma_rates(alpha, k=4):
    """Yang-1994 mean discretisation (scipy; data generation only)."""
    from scipy.stats import gamma as g
    d, d1 = g(alpha, scale=1.0 / alpha), g(alpha + 1.0, scale=1.0 / alpha)
    cuts = [0.0] + [d.ppf(i / k) for i in range(1, k)] + [math.inf]
    return np.array([k * (d1.cdf(cuts[i + 1]) - d1.cdf(cuts[i])) for i in range(k)])


def dna_workload(n_tips=512, W=1500, n_reads=100000, read_len=150, seeds=(1, 2, 3)):
    """cfg2 of BASELINE.json."""
    root = random_tree(n_tips, seeds[0])
    rates = gamma_rates(CFG2_ALPHA)
    labels, seqs = simulate_msa(root, W, CFG2_SUBST, CFG2_FREQS, rates, seeds[1])
    reads, _ = make_reads(seqs, n_reads, read_len, 0.03, seeds[2])
    return {"newick": newick(root), "labels": labels, "seqs": seqs, "reads": reads,
            "states": 4, "subst": CFG2_SUBST, "freqs": CFG2_FREQS, "rates": rates,
            "weights": np.full(4, 0.25)}


def aa_model(seed=7):
    """deterministic pseudo-random PROTGTR exchangeabilities + frequencies (cfg3 stand-in: the
    named LG matrix lives in the pll-modules model database, which is not available)"""
    rng = np.random.RandomState(seed)
    subst = np.round(rng.gamma(1.0, 2.0, 190) + 0.01, 6)
    freqs = rng.dirichlet(np.full(20, 8.0))
    freqs = freqs / freqs.sum()
    return subst.tolist(), freqs.tolist()


def aa_workload(n_tips=2000, W=500, n_reads=50000, read_len=100, seeds=(11, 12, 13), alpha=0.563473):
    """cfg3 of BASELINE.json."""
    subst, freqs = aa_model()
    root = random_tree(n_tips, seeds[0])
    rates = gamma_rates(alpha)
    labels, seqs = simulate_msa(root, W, subst, freqs, rates, seeds[1])
    reads, _ = make_reads(seqs, n_reads, read_len, 0.03, seeds[2], states=20)
    return {"newick": newick(root), "labels": labels, "seqs": seqs, "reads": reads, "states": 20,
            "subst": subst, "freqs": freqs, "rates": rates, "weights": np.full(4, 0.25)}


# lookup-column order of the query codes (NT_MAP / AA_MAP of the reference's src/util/maps.hpp)
NT_COLS = "-TGKCYSBAWRDMHVN"
AA_COLS = "ACDEFGHIKLMNPQRSTVWY-XBZ"


def make_reads_compact(seqs, n_reads, read_len, sub_rate, seed, states=4):
    """The reads of make_reads(seqs, n_reads, read_len, sub_rate, seed) -- same random draws --
    straight in the compact wire layout of epa_encode_queries_compact: (codes uint8 [n][stride]
    holding only each read's window, win_begin uint32 [n], win_span uint32 [n]); no n x W ASCII
    rows are ever built (10^7 reads of a 1500-column alignment would be 15 GB of text)."""
    rng = np.random.RandomState(seed)
    alphabet = DNA if states == 4 else AA
    cols_of = NT_COLS if states == 4 else AA_COLS
    W = len(seqs[0])
    tips = rng.randint(0, len(seqs), n_reads)
    starts = rng.randint(0, W - read_len + 1, n_reads)
    arr = np.frombuffer("".join(seqs).encode(), dtype=np.uint8).reshape(len(seqs), W)
    alpha = np.frombuffer(alphabet.encode(), dtype=np.uint8)
    cols = starts[:, None] + np.arange(read_len)[None, :]
    frag = arr[tips[:, None], cols]
    mut = rng.random_sample(frag.shape) < sub_rate
    frag = np.where(mut, alpha[rng.randint(0, len(alpha), frag.shape)], frag)
    lut = np.full(256, 255, np.uint8)
    for i, ch in enumerate(cols_of):
        lut[ord(ch)] = i
    stride = (read_len + 15) // 16 * 16
    if stride == W:
        stride += 16
    codes = np.zeros((n_reads, stride), np.uint8)
    codes[:, :read_len] = lut[frag]
    assert codes.max() < 255
    # get_valid_range trims literal '-' only; the simulated tips hold none
    return codes, starts.astype(np.uint32), np.full(n_reads, read_len, np.uint32)


def compact_to_ascii(codes, win_begin, win_span, W, states=4):
    """compact code rows -> full-width ASCII query rows (for the CPU oracle, which reads text)"""
    cols_of = np.frombuffer((NT_COLS if states == 4 else AA_COLS).encode(), dtype=np.uint8)
    out = np.full((len(codes), W), ord("-"), np.uint8)
    for q in range(len(codes)):
        b, n = int(win_begin[q]), int(win_span[q])
        out[q, b:b + n] = cols_of[codes[q, :n]]
    return [row.tobytes().decode() for row in out]


def _full_code_rows(codes, win_begin, win_span, W):
    """compact 4-bit-code rows -> [Q][W] code rows (0 = '-'), vectorised"""
    Q, stride = codes.shape
    full = np.zeros((Q, W), np.uint8)
    cols = win_begin.astype(np.int64)[:, None] + np.arange(stride, dtype=np.int64)[None, :]
    ok = np.arange(stride)[None, :] < win_span.astype(np.int64)[:, None]
    rows = np.broadcast_to(np.arange(Q, dtype=np.int64)[:, None], cols.shape)
    full[rows[ok], cols[ok]] = codes[ok]
    return full


def _headers8(first, n):
    """n fixed-width labels q0000000 .. as an [n][8] byte matrix"""
    ids = np.arange(first, first + n, dtype=np.int64)
    h = np.empty((n, 8), np.uint8)
    h[:, 0] = ord("q")
    for k in range(7):
        h[:, 7 - k] = ord("0") + (ids // 10 ** k) % 10
    return h


def write_query_files(fasta_path, bfast_path, chunks, W):
    """bench.py's cli_e2e leg: the same nucleotide reads as an aligned FASTA file (one header line + one sequence
    line of W characters per read) and as the reference's binary fasta (src/io/Binary_Fasta.hpp:
    "BFAST\\0\\0" | u64 n | u64 mask_len + mask chars | n x (u64 id, u64 offset) | per read: u64 header length,
    header, u64 sites, ceil(sites / 2) bytes of 4-bit codes, earlier site in the high nibble).
    chunks: iterable of (codes [Q][stride] uint8, win_begin, win_span).  -> number of reads"""
    import struct
    chunks = list(chunks)
    n = sum(len(c[1]) for c in chunks)
    lut = np.frombuffer(NT_COLS.encode(), dtype=np.uint8)
    rec = 8 + 8 + 8 + (W + 1) // 2                      # bfast record: fixed size with 8-byte labels
    data0 = 7 + 8 + 8 + W + 16 * n
    first = 0
    with open(fasta_path, "wb") as ff, open(bfast_path, "wb") as fb:
        fb.write(b"BFAST\0\0" + struct.pack("<QQ", n, W) + b"0" * W)
        table = np.empty((n, 2), "<u8")
        table[:, 0] = np.arange(n)
        table[:, 1] = data0 + rec * np.arange(n, dtype=np.uint64)
        fb.write(table.tobytes())
        for codes, wb, ws in chunks:
            Q = len(wb)
            full = _full_code_rows(codes, wb, ws, W)
            hd = _headers8(first, Q)
            fa = np.empty((Q, 1 + 8 + 1 + W + 1), np.uint8)
            fa[:, 0] = ord(">")
            fa[:, 1:9] = hd
            fa[:, 9] = ord("\n")
            fa[:, 10:10 + W] = lut[full]
            fa[:, 10 + W] = ord("\n")
            ff.write(fa.tobytes())
            if W & 1:
                full = np.concatenate([full, np.zeros((Q, 1), np.uint8)], axis=1)
            bf = np.empty((Q, rec), np.uint8)
            bf[:, 0:8] = np.frombuffer(struct.pack("<Q", 8), np.uint8)
            bf[:, 8:16] = hd
            bf[:, 16:24] = np.frombuffer(struct.pack("<Q", W), np.uint8)
            bf[:, 24:] = (full[:, 0::2] << 4) | full[:, 1::2]
            fb.write(bf.tobytes())
            first += Q
    return n
