"""CPU tests for the paired-end merge oracle (oracle/orc_merge.c; SURVEY 8f row f2, vsearch --fastq_mergepairs restated).

PARITY UNPINNED: the reference's merged fixture (227 reads) was made by BBMerge, not vsearch -- measured here from its
qualities.  What these tests hold: (a) the C restatement equals an independent Python statement of the same procedure;
(b) anchors against the reference's data: 236 of the 250 fixture pairs merge (the reference's end-to-end test expects
235 trimmed reads from the vsearch merge, tests/test_main_pytest.py:252), and on the pairs both tools merge the
merged SEQUENCES agree.
"""
import gzip
import math
import os

import numpy as np

import orc

_COMP = str.maketrans("ACGTN", "TGCAN")


def _read_fastq(path):
    out = []
    with gzip.open(path, "rt") as f:
        while True:
            h = f.readline()
            if not h:
                break
            s = f.readline().strip()
            f.readline()
            q = f.readline().strip()
            out.append((h[1:].split()[0], s, q))
    return out


# ---------------------------------------------------------------- an independent statement of the procedure
def _q2p(c):
    x = c - 33
    return 0.75 if x < 2 else 10.0 ** (-x / 10.0)


def _qual(p):
    return 33 + max(0, min(41, int(round(-10.0 * math.log10(p)))))


def py_merge(f, fq, r, rq, maxdiffs=40, maxee=2.0, allow_stagger=False):
    fl, rl = len(f), len(r)
    rc, rcq = r[::-1].translate(_COMP), rq[::-1]
    pos = {}
    for i in range(fl - 4):
        if "N" not in f[i:i + 5]:
            pos.setdefault(f[i:i + 5], []).append(i)
    diag = {}
    for j in range(rl - 4):
        for i in pos.get(rc[j:j + 5], ()) if "N" not in rc[j:j + 5] else ():
            diag[i - j] = diag.get(i - j, 0) + 1
    cands = []
    for shift in sorted((d for d, c in diag.items() if c >= 4), reverse=True):
        a, b = max(0, shift), min(fl, shift + rl)
        score = high = drop = 0.0
        diffs = 0
        for p in range(b - 1, a - 1, -1):
            px, py = _q2p(ord(fq[p])), _q2p(ord(rcq[p - shift]))
            if f[p] == rc[p - shift]:
                score += math.log2((1.0 - px - py + px * py * 4.0 / 3.0) / 0.25)
            else:
                score += math.log2(((px + py) / 3.0 - px * py * 4.0 / 9.0) / 0.25)
                diffs += 1
            high = max(high, score)
            drop = max(drop, high - score)
        if drop >= 16.0:
            score = -1000.0
        cands.append((score, shift, diffs, b - a))
    if not cands:
        return "nokmers", None
    best = cands[0]
    for c in cands[1:]:
        if c[0] > best[0]:
            best = c
    if sum(c[0] >= 16.0 for c in cands) > 1:
        return "repeat", None
    score, shift, diffs, ov = best
    if score < 16.0:
        return "minscore", None
    if diffs > maxdiffs:
        return "maxdiffs", None
    if ov < 10:
        return "minovlen", None
    if shift < 0 and not allow_stagger:
        return "staggered", None
    a, b = max(0, shift), min(fl, shift + rl)
    seq, qual = list(f[:a]), list(fq[:a])
    for p in range(a, b):
        fs, rs, qa, qb = f[p], rc[p - shift], ord(fq[p]), ord(rcq[p - shift])
        px, py = _q2p(qa), _q2p(qb)
        if rs == "N":
            s, q = fs, qa
        elif fs == "N":
            s, q = rs, qb
        elif fs == rs:
            s, q = fs, _qual(px * py / 3.0 / (1.0 - px - py + 4.0 * px * py / 3.0))
        elif qa > qb:
            s, q = fs, _qual(px * (1.0 - py / 3.0) / (px + py - 4.0 * px * py / 3.0))
        else:
            s, q = rs, _qual(py * (1.0 - px / 3.0) / (px + py - 4.0 * px * py / 3.0))
        seq.append(s)
        qual.append(chr(q))
    if shift + rl >= fl:
        seq += list(rc[b - shift:])
        qual += list(rcq[b - shift:])
    if sum(_q2p(ord(c)) for c in qual) > maxee:
        return "maxee", None
    return "ok", ("".join(seq), "".join(qual))


# ---------------------------------------------------------------- tests
def test_merge_matches_the_python_statement_on_the_fixture_pairs(gold):
    r1 = _read_fastq(os.path.join(gold, "4774-1-MSITS3_R1.fastq.gz"))
    r2 = _read_fastq(os.path.join(gold, "4774-1-MSITS3_R2.fastq.gz"))
    assert len(r1) == len(r2) == 250
    reasons = {}
    for (h, f, fq), (_, r, rq) in zip(r1, r2):
        for stagger in (False, True):
            reason, s, q, score, shift = orc.merge_pair(f, fq, r, rq, allow_stagger=stagger)
            preason, pm = py_merge(f, fq, r, rq, allow_stagger=stagger)
            assert reason == preason, (h, reason, preason)
            if reason == "ok":
                assert (s, q) == pm, h
        reasons[h] = orc.merge_pair(f, fq, r, rq)
    # anchors against the reference's data
    merged = {h: v for h, v in reasons.items() if v[0] == "ok"}
    assert len(merged) == 236
    fixture = {h: (s, q) for h, s, q in _read_fastq(os.path.join(gold, "seq.fq.gz"))}
    both = [h for h in merged if h in fixture]
    assert len(both) == 226
    assert sum(merged[h][1] == fixture[h][0] for h in both) == 225
    # and the measurement that says the fixture is not a vsearch product: agreeing Q38 + Q<12 bases carry 38 + q/4 there,
    # where the Edgar-Flyvbjerg posterior is far above the 41 cap
    low = 0
    for (h, f, fq), (_, r, rq) in zip(r1, r2):
        if h not in fixture or h not in merged or merged[h][1] != fixture[h][0]:
            continue
        shift = merged[h][4]
        rcq = rq[::-1]
        rc = r[::-1].translate(_COMP)
        for p in range(max(0, shift), min(len(f), shift + len(r))):
            qa, qb = ord(fq[p]) - 33, ord(rcq[p - shift]) - 33
            if f[p] == rc[p - shift] and max(qa, qb) == 38 and 2 <= min(qa, qb) < 12:
                assert ord(fixture[h][1][p]) - 33 == 38 + min(qa, qb) // 4 and ord(merged[h][2][p]) - 33 == 41
                low += 1
    assert low > 500


def test_merge_semantics():
    rng = np.random.default_rng(2)
    frag = "".join(rng.choice(list("ACGT"), 400))
    rc = lambda s: s[::-1].translate(_COMP)
    f, r = frag[:250], rc(frag[150:])                    # 100-base overlap
    q = "I" * 250
    reason, s, mq, score, shift = orc.merge_pair(f, q, r, q)
    assert reason == "ok" and s == frag and shift == 150 and set(mq[150:250]) == {"J"} and mq[:150] == "I" * 150
    # a disagreement keeps the better base; ties go to the reverse read
    f2 = f[:200] + ("A" if f[200] != "A" else "C") + f[201:]
    fq2 = q[:200] + "#" + q[201:]
    assert orc.merge_pair(f2, fq2, r, q)[1] == frag
    reason, s, *_ = orc.merge_pair(f2, q, r, q)
    assert reason == "ok" and s == frag
    # N yields to the other read and keeps that read's quality
    fN = f[:210] + "N" + f[211:]
    reason, s, mq, *_ = orc.merge_pair(fN, q[:210] + "!" + q[211:], r, q)
    assert reason == "ok" and s == frag and mq[210] == "I"
    # too many expected errors
    assert orc.merge_pair(f, "+" * 250, r, "+" * 250)[0] == "maxee"
    # unrelated reads share no diagonal with four 5-mers
    other = "".join(rng.choice(list("ACGT"), 250))
    assert orc.merge_pair(f, q, other, q)[0] in ("nokmers", "minscore")
    # staggered pair: the reverse read's 3' end overhangs the forward read's start
    fs, rs = frag[100:300], rc(frag[50:250])
    assert orc.merge_pair(fs, "I" * 200, rs, "I" * 200)[0] == "staggered"
    reason, s, *_ = orc.merge_pair(fs, "I" * 200, rs, "I" * 200, allow_stagger=True)
    assert reason == "ok" and s == frag[100:300][:150] + "" and len(s) == 150
    # a tandem repeat offers several alignments
    rep = "ACGGTCATTG" * 30
    assert orc.merge_pair(rep[:200], "I" * 200, rc(rep[100:300]), "I" * 200)[0] == "repeat"
