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
_COMP.setflags(write=False)


_SYNTHC = False


def _synthc():
    """tests/_synthc.so (tests/csrc/synthc.c, built by __graft_entry__.build()) or None"""
    global _SYNTHC
    if _SYNTHC is False:
        import ctypes
        import os
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_synthc.so")
        _SYNTHC = None
        if os.path.exists(path) and not os.environ.get("SYNTH_NO_C"):
            L = ctypes.CDLL(path)
            vp, i64 = ctypes.c_void_p, ctypes.c_int64
            L.synth_ragged_gather.argtypes = [vp, i64, vp, vp, vp, i64, vp]
            L.synth_ragged_gather.restype = None
            L.synth_revcomp.argtypes = [vp, vp, vp, i64, vp]
            L.synth_revcomp.restype = None
            _SYNTHC = L
    return _SYNTHC


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
               len_range=(300, 580), frac_templates=0.02, sub_rate=0.003, n_rate=0.0005, rc_rate=0.05,
               template_seed=None, as_array=False):
    """fixed_len > 0: every read has that length (BASELINE configs[1]); fixed_len = 0: template lengths uniform in
    len_range (configs[2]: merged pairs, 300-580).  template_seed: the template library is drawn from its own generator,
    so that shards generated with different `seed`s share their templates (cross-shard duplicates).  as_array: return
    the bases as a numpy uint8 array instead of bytes (no 4-GB copy at configs[2] size)."""
    rng = np.random.default_rng(SEED + config if seed is None else seed)
    trng = rng if template_seed is None else np.random.default_rng(template_seed)
    lm = [m for m in consensus_motifs(hmm_text, left) if len(m) == 45]
    rm = [m for m in consensus_motifs(hmm_text, right) if len(m) == 45]
    nt = max(1, int(n_reads * frac_templates))
    acgt = np.frombuffer(b"ACGT", np.uint8)
    if fixed_len:
        lens = np.full(nt, fixed_len, np.int64)
    else:
        lens = trng.integers(len_range[0], len_range[1] + 1, nt)
    lflank = trng.integers(60, 111, nt)
    rflank = np.full(nt, 60)
    Lmax = int(lens.max())
    tmpl = acgt[trng.integers(0, 4, (nt, Lmax))]
    li = trng.integers(0, len(lm), nt)
    ri = trng.integers(0, len(rm), nt)
    lmot = np.array([np.frombuffer(m.encode(), np.uint8) for m in lm])
    rmot = np.array([np.frombuffer(m.encode(), np.uint8) for m in rm])
    for t in range(nt) if nt < 4096 else ():               # small libraries: the original loop (same draws, same result)
        a = int(lflank[t])
        tmpl[t, a:a + 45] = lmot[li[t]]
        b = int(lens[t] - rflank[t] - 45)
        tmpl[t, b:b + 45] = rmot[ri[t]]
    if nt >= 4096:
        cols = np.arange(45)
        rows = np.arange(nt)[:, None]
        tmpl[rows, lflank[:, None] + cols] = lmot[li]
        tmpl[rows, (lens - rflank - 45)[:, None] + cols] = rmot[ri]
    # Zipf(1.1) over templates
    w = 1.0 / np.arange(1, nt + 1) ** 1.1
    ids = rng.choice(nt, size=n_reads, p=w / w.sum())
    rlen = lens[ids]
    offs = np.zeros(n_reads + 1, np.int64)
    np.cumsum(rlen, out=offs[1:])
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
        # ragged lengths, in blocks of reads: the block's text is gathered from the templates, substitutions and N's are
        # drawn as a binomial count of positions over the block's bases (a position hit twice is hit once), then the flagged
        # reads are reverse-complemented.  The two byte-moving loops run in C when tests/_synthc.so is built (same result).
        blob = np.empty(int(offs[-1]), np.uint8)
        rc = (rng.random(n_reads) < rc_rate).astype(np.uint8)
        ids = np.ascontiguousarray(ids, np.int64)
        rlen = np.ascontiguousarray(rlen, np.int64)
        lib = _synthc()
        ar = np.arange(Lmax)
        BLK = 1 << 19
        for c0 in range(0, n_reads, BLK):
            c1 = min(n_reads, c0 + BLK)
            blk = blob[offs[c0]:offs[c1]]
            co = offs[c0:c1 + 1] - offs[c0]
            if lib is not None:
                lib.synth_ragged_gather(tmpl.ctypes.data, Lmax, ids[c0:c1].ctypes.data, rlen[c0:c1].ctypes.data,
                                        co.ctypes.data, c1 - c0, blk.ctypes.data)
            else:
                blk[:] = tmpl[ids[c0:c1]][ar[None, :] < rlen[c0:c1, None]]
            tot = int(co[-1])
            k = int(rng.binomial(tot, sub_rate)) if tot else 0
            if k:
                flat = rng.integers(0, tot, k)
                blk[flat] = acgt[(np.searchsorted(acgt, blk[flat]) + rng.integers(1, 4, k)) % 4]
            k = int(rng.binomial(tot, n_rate)) if tot else 0
            if k:
                blk[rng.integers(0, tot, k)] = ord("N")
            if lib is not None:
                lib.synth_revcomp(blk.ctypes.data, co.ctypes.data, rc[c0:c1].ctypes.data, c1 - c0, _COMP.ctypes.data)
            else:
                for i in np.nonzero(rc[c0:c1])[0]:
                    blk[co[i]:co[i + 1]] = _COMP[blk[co[i]:co[i + 1]][::-1]]
    return (blob if as_array else blob.tobytes()), offs


def to_strings(blob, offs):
    return [blob[offs[i]:offs[i + 1]].decode() for i in range(len(offs) - 1)]
