"""Independent full-matrix DP for extension alignment (CPU ORACLE second opinion).

Test infrastructure, NOT product code.  Written from the recurrence definitions
(SURVEY.md §8a), not from the row-streaming C code: whole H/E/F matrices, no band,
no beg/end trimming, no early break, no zdrop.  It therefore agrees with
ksw_extend2 only on inputs where none of those can bind — in practice: band
w >= max(qlen, tlen), zdrop = 0, and h0 large enough that every H stays > 0
(tests construct exactly such inputs).

Restates sw_pe_array_sw_extend.v K2/K4/K5/K7/K9 rows of SURVEY.md §8a.
"""
import numpy as np


def full_dp(query, target, mat, o_del, e_del, o_ins, e_ins, h0, variant=0, m=5):
    q = np.asarray(query, dtype=np.int64)
    t = np.asarray(target, dtype=np.int64)
    mat = np.asarray(mat, dtype=np.int64).reshape(m, m)
    ql, tl = len(q), len(t)
    oe_del, oe_ins = o_del + e_del, o_ins + e_ins
    H = np.zeros((tl, ql), dtype=np.int64)
    E = np.zeros((tl + 1, ql), dtype=np.int64)      # E[i][j] enters row i
    top = [max(h0 - oe_ins - j * e_ins, 0) for j in range(ql)]       # H(-1, j)
    left = [max(h0 - o_del - e_del * (i + 1), 0) for i in range(tl)]  # H(i, -1)
    for i in range(tl):
        f = 0
        for j in range(ql):
            if i == 0:
                diag = h0 if j == 0 else top[j - 1]
            else:
                diag = left[i - 1] if j == 0 else H[i - 1][j - 1]
            s = mat[t[i]][q[j]]
            if variant == 1:
                M = diag + s if diag != 0 else 0
            else:
                M = diag + s
            h = max(M, E[i][j], f)
            H[i][j] = h
            base = M if variant == 1 else h
            E[i + 1][j] = max(E[i][j] - e_del, max(base - oe_del, 0))
            f = max(f - e_ins, max(base - oe_ins, 0))
    # scalars with the tie-break rules of K5/K7
    best, max_i, max_j, max_off = h0, -1, -1, 0
    gscore, max_ie = -1, -1
    for i in range(tl):
        row = H[i]
        mrow = int(row.max()) if ql else 0
        mj = int(np.nonzero(row == mrow)[0][-1]) if ql else -1     # ties -> later j
        if ql and not (gscore > row[ql - 1]):                       # ties -> later i
            max_ie = i
        if ql:
            gscore = max(gscore, int(row[ql - 1]))
        if mrow > best:
            best, max_i, max_j = mrow, i, mj
            max_off = max(max_off, abs(mj - i))
    return dict(score=int(best), qle=max_j + 1, tle=max_i + 1, gtle=max_ie + 1,
                gscore=int(gscore), max_off=int(max_off), minH=int(H.min()) if H.size else 0)


NEG = -(1 << 40)


def global_dp(query, target, mat, o_del, e_del, o_ins, e_ins, m=5):
    """Independent unbanded global affine-gap alignment score (Gotoh; three full matrices), written from the textbook
    recurrence — the second opinion for oracle/ksw_global_ref.c when its band cannot bind.  Deletion = gap in the
    query (consumes target), insertion = gap in the target (consumes query)."""
    q = np.asarray(query, dtype=np.int64)
    t = np.asarray(target, dtype=np.int64)
    mat = np.asarray(mat, dtype=np.int64).reshape(m, m)
    ql, tl = len(q), len(t)
    H = np.full((tl + 1, ql + 1), NEG, dtype=np.int64)
    E = np.full((tl + 1, ql + 1), NEG, dtype=np.int64)     # ends with a deletion
    F = np.full((tl + 1, ql + 1), NEG, dtype=np.int64)     # ends with an insertion
    H[0][0] = 0
    for j in range(1, ql + 1):
        F[0][j] = H[0][j] = -(o_ins + e_ins * j)
    for i in range(1, tl + 1):
        E[i][0] = H[i][0] = -(o_del + e_del * i)
        for j in range(1, ql + 1):
            Mv = H[i - 1][j - 1] + mat[t[i - 1]][q[j - 1]]
            # bwa opens gaps from M only (ksw.c comment: "10M3I3D10M" is disallowed by global())
            E[i][j] = max(E[i - 1][j] - e_del, _M(H, E, F, i - 1, j, mat, t, q) - o_del - e_del)
            F[i][j] = max(F[i][j - 1] - e_ins, _M(H, E, F, i, j - 1, mat, t, q) - o_ins - e_ins)
            H[i][j] = max(Mv, E[i][j], F[i][j])
    return int(H[tl][ql])


def _M(H, E, F, i, j, mat, t, q):
    """M(i,j) = H(i-1,j-1) + s: the score of a path that ends with a (mis)match at (i,j); borders: the leading gap."""
    if i == 0 and j == 0:
        return 0
    if i == 0 or j == 0:
        return NEG          # a border cell ends with a gap, not with a match
    return H[i - 1][j - 1] + mat[t[i - 1]][q[j - 1]]


def local_dp(query, target, mat, o_del, e_del, o_ins, e_ins, m=5):
    """Textbook Gotoh LOCAL alignment (second opinion for ksw_align): whole matrices, row-vectorised in numpy.
    E (gap along the target, "deletion") and F (gap along the query, "insertion") may follow each other freely —
    bwa's striped code forbids an insertion directly followed by a deletion, so its score is <= this one and equal
    whenever no optimal alignment needs that.  Returns dict(score, ends=set of (te, qe) reaching the score)."""
    q = np.asarray(query, dtype=np.int64)
    t = np.asarray(target, dtype=np.int64)
    mat = np.asarray(mat, dtype=np.int64).reshape(m, m)
    ql, tl = len(q), len(t)
    oe_del, oe_ins = o_del + e_del, o_ins + e_ins
    Hp = np.zeros(ql + 1, dtype=np.int64)
    E = np.zeros(ql + 1, dtype=np.int64)
    best, ends = 0, set()
    for i in range(tl):
        s = mat[t[i]][q] if ql else np.zeros(0, dtype=np.int64)
        M = np.maximum(Hp[:-1] + s, 0)
        M = np.maximum(M, E[1:])
        H = np.zeros(ql + 1, dtype=np.int64)
        f = 0
        for j in range(ql):                                  # F runs along the query inside the row
            h = max(M[j], f)
            H[j + 1] = h
            f = max(f - e_ins, h - oe_ins, 0)
        E[1:] = np.maximum(np.maximum(E[1:] - e_del, H[1:] - oe_del), 0)
        mrow = int(H.max()) if ql else 0
        if mrow > best:
            best, ends = mrow, set()
        if mrow == best and best > 0:
            for j in np.nonzero(H[1:] == best)[0]:
                ends.add((i, int(j)))
        Hp = H
    return dict(score=int(best), ends=ends)
