"""Seeded synthetic amplicon generator shared by tests and bench.py.

Template = left flank + 5.8S-end motif + ITS2 spacer + LSU-start motif + right flank, motifs
being the consensus (CONS column) of a random `3_*` / `4_*` profile of the taxon file in use.
Library = 2 % distinct templates, reads drawn Zipf(1.1) over templates, 0.3 % substitutions,
0.05 % N, 5 % reverse-complemented.  Returns ASCII bytes + offsets ready for
Engine.set_reads_buffer.
"""
import numpy as np

SEED = 20240405
_COMP = np.zeros(256, np.uint8)
for a, b in zip(b"ACGTN", b"TGCAN"):
    _COMP[a] = b


def consensus_motifs(hmm_text, prefix):
    """consensus strings of every profile whose NAME starts with prefix"""
    out = []
    for block in hmm_text.split("//\n"):
        if "NAME  " not in block:
            continue
        name = block.split("NAME  ")[1].split("\n")[0].strip()
        if not name.startswith(prefix):
            continue
        lines = block.split("\n")
        i = next(k for k, ln in enumerate(lines) if ln.startswith("HMM "))
        cons = []
        k = i + 5                       # header, COMPO, insert line, begin transitions
        while k < len(lines):
            tok = lines[k].split()
            if len(tok) >= 7 and tok[0].isdigit():
                cons.append(tok[6].upper())
                k += 3
            else:
                break
        out.append("".join(cons))
    return out


def make_reads(hmm_text, n_reads, config=2, left="3_", right="4_", seed=None, fixed_len=300,
               len_range=(300, 580), frac_templates=0.02, sub_rate=0.003, n_rate=0.0005, rc_rate=0.05):
    rng = np.random.default_rng(SEED + config if seed is None else seed)
    lm = [m for m in consensus_motifs(hmm_text, left) if len(m) == 45]
    rm = [m for m in consensus_motifs(hmm_text, right) if len(m) == 45]
    nt = max(1, int(n_reads * frac_templates))
    acgt = np.frombuffer(b"ACGT", np.uint8)
    if fixed_len:
        lens = np.full(nt, fixed_len, np.int64)
    else:
        lens = rng.integers(len_range[0], len_range[1] + 1, nt)
    lflank = rng.integers(60, 111, nt)
    rflank = np.full(nt, 60)
    Lmax = int(lens.max())
    tmpl = acgt[rng.integers(0, 4, (nt, Lmax))]
    li = rng.integers(0, len(lm), nt)
    ri = rng.integers(0, len(rm), nt)
    lmot = np.array([np.frombuffer(m.encode(), np.uint8) for m in lm])
    rmot = np.array([np.frombuffer(m.encode(), np.uint8) for m in rm])
    for t in range(nt):
        a = int(lflank[t])
        tmpl[t, a:a + 45] = lmot[li[t]]
        b = int(lens[t] - rflank[t] - 45)
        tmpl[t, b:b + 45] = rmot[ri[t]]
    # Zipf(1.1) over templates
    w = 1.0 / np.arange(1, nt + 1) ** 1.1
    ids = rng.choice(nt, size=n_reads, p=w / w.sum())
    rlen = lens[ids]
    offs = np.zeros(n_reads + 1, np.int64)
    np.cumsum(rlen, out=offs[1:])
    blob = np.empty(int(offs[-1]), np.uint8)
    if fixed_len:
        reads = tmpl[ids]                                   # [n, L]
        sub = rng.random(reads.shape) < sub_rate
        if sub.any():
            cur = reads[sub]
            idx = np.searchsorted(acgt, cur)
            reads[sub] = acgt[(idx + rng.integers(1, 4, cur.shape[0])) % 4]
        nm = rng.random(reads.shape) < n_rate
        reads[nm] = ord("N")
        rc = rng.random(n_reads) < rc_rate
        reads[rc] = _COMP[reads[rc][:, ::-1]]
        blob = reads.reshape(-1)
    else:
        rc = rng.random(n_reads) < rc_rate
        for i in range(n_reads):
            r = tmpl[ids[i], :rlen[i]].copy()
            sub = rng.random(r.shape[0]) < sub_rate
            if sub.any():
                idx = np.searchsorted(acgt, r[sub])
                r[sub] = acgt[(idx + rng.integers(1, 4, int(sub.sum()))) % 4]
            r[rng.random(r.shape[0]) < n_rate] = ord("N")
            if rc[i]:
                r = _COMP[r[::-1]]
            blob[offs[i]:offs[i + 1]] = r
    return blob.tobytes(), offs


def to_strings(blob, offs):
    return [blob[offs[i]:offs[i + 1]].decode() for i in range(len(offs) - 1)]
