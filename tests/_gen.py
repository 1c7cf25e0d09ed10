"""Random / adversarial seed generators for the tests (numpy only; deterministic by seed)."""
import numpy as np


def mutate(rng, ref, n, sub, indel):
    """Derive an n-base read from reference bases `ref` with substitutions and indels."""
    out = []
    rp = 0
    while len(out) < n:
        r = rng.random()
        if r < indel / 2:
            out.append(int(rng.integers(0, 4)))
        elif r < indel:
            rp += int(rng.integers(1, 12))
        else:
            b = int(ref[rp]) if rp < len(ref) else int(rng.integers(0, 4))
            rp += 1
            if rng.random() < sub:
                b = (b + 1 + int(rng.integers(0, 3))) & 3
            out.append(b)
    return np.array(out, dtype=np.uint8)


def random_seeds(rng, n, qmin=1, qmax=140, tfac=2.0, sub=0.03, indel=0.01, junk=0.2, nrate=0.0, h0max=60,
                 both_sides=True, tmin=0):
    """List of seed dicts for host.make_tasks()."""
    seeds = []
    for k in range(n):
        s = {"h0": int(rng.integers(1, h0max + 1))}
        for side in ("l", "r"):
            if side == "l" and (not both_sides or rng.random() < 0.2):
                continue
            if side == "r" and both_sides and rng.random() < 0.1 and "lq" in s:
                continue
            ql = int(rng.integers(qmin, qmax + 1))
            tl = max(tmin, int(rng.integers(0, int(ql * tfac) + 2)))
            t = rng.integers(0, 4, tl).astype(np.uint8)
            if rng.random() < junk:
                q = rng.integers(0, 4, ql).astype(np.uint8)
            else:
                q = mutate(rng, t, ql, sub, indel)
            if nrate > 0:
                q[rng.random(ql) < nrate] = 4
                t[rng.random(tl) < nrate] = 4
            s[side + "q"], s[side + "t"] = q, t
        s["init_score"] = -1 if rng.random() < 0.8 else int(rng.integers(-1, 50))
        s["tag"] = int(rng.integers(0, 2 ** 32))
        seeds.append(s)
    return seeds


def query_has_n(tasks, arena, side):
    """Per task: does the left (side 0) / right (side 1) query hold a base code >= 4?  (the binning's second key)"""
    import numpy as np
    pf, lf = ("lquery", "lqlen") if side == 0 else ("rquery", "rqlen")
    base = arena.ctypes.data
    a = np.asarray(arena).view(np.uint8).ravel()
    out = np.zeros(len(tasks), bool)
    for i, (p, l) in enumerate(zip(tasks[pf], tasks[lf])):
        if l:
            o = int(p) - base
            out[i] = bool((a[o:o + int(l)] >= 4).any())
    return out
