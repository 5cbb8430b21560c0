#!/usr/bin/env python3
"""Generate tests/golden/* from the reference (run in the build container only).

/root/reference is a Python project; its glue classes can be imported here (with
stand-ins for the third-party modules that are not installed) to freeze expected
outputs.  Nothing of the reference's source travels: this script writes DATA only
(inputs and expected outputs) under tests/golden/.

Outputs
  fixture_reads.fa.gz        the 227 merged reads of tests/test_data/ex_tmpdir/seq.fq.gz (id, sequence)
  fixture_uc.txt             tests/test_data/ex_tmpdir/uc.txt verbatim (frozen vsearch output, data)
  fixture_rep.fa             tests/test_data/ex_tmpdir/rep.fa verbatim (frozen vsearch output, data)
  matchdict.json             Dedup(uc.txt).matchdict as parsed by the reference class (227 entries)
  fungi_its2_coords.tsv      226 golden (read, rep, start, stop, tlen) from t2_r1.fq/t2_r2.fq
  runtime_hmm_names.json     create_runtime_hmm(taxa, region) NAME lists for the taxon files present
  itsposition_cases.json     synthetic domtbl texts + the ddict/get_position the reference class returns
  xxh64_kat.json             XXH64 known answers from the `xxhash` library
  4774-1-MSITS3_R{1,2}.fastq.gz, seq.fq.gz, t2_r{1,2}.fq.gz
                             inputs and byte-compared outputs of the reference's trimming tests (data)
  T.hmm.gz, mini.hmm         ITSx profile DATA: Tracheophyta set (stand-in taxon for the bench),
                             and a 10-profile subset for fast tests
  all_its2.hmm.gz            the --taxa All --region ITS2 runtime profile set (814 profiles) as create_runtime_hmm writes it
"""
import gzip
import io
import json
import os
import random
import shutil
import sys
import tempfile
import types

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
TD = os.path.join(REF, "tests", "test_data")


def import_reference():
    for name in ("pyzstd", "Bio", "Bio.SeqIO", "Bio.Seq", "Bio.SeqRecord", "itsxpress._version"):
        m = types.ModuleType(name)
        sys.modules[name] = m
    sys.modules["Bio.Seq"].Seq = object
    sys.modules["Bio.SeqRecord"].SeqRecord = object
    sys.modules["Bio"].SeqIO = sys.modules["Bio.SeqIO"]
    sys.modules["itsxpress._version"].version = "0+golden"
    sys.modules["itsxpress._version"].__version__ = "0+golden"
    sys.path.insert(0, REF)
    import itsxpress.SeqSample as S
    import itsxpress.main as M
    import itsxpress.definitions as D
    return S, M, D


def read_fastq(path):
    op = gzip.open if path.endswith(".gz") else open
    recs = []
    with op(path, "rt") as f:
        while True:
            h = f.readline()
            if not h:
                break
            s = f.readline().rstrip("\n")
            f.readline()
            q = f.readline().rstrip("\n")
            recs.append((h[1:].split()[0], s, q))
    return recs


def main():
    os.makedirs(OUT, exist_ok=True)
    S, M, D = import_reference()

    # --- fixture reads and frozen vsearch outputs (data files of the reference's own tests)
    merged = read_fastq(os.path.join(TD, "ex_tmpdir", "seq.fq.gz"))
    with gzip.open(os.path.join(OUT, "fixture_reads.fa.gz"), "wt") as f:
        for rid, s, _ in merged:
            f.write(">%s\n%s\n" % (rid, s))
    shutil.copyfile(os.path.join(TD, "ex_tmpdir", "uc.txt"), os.path.join(OUT, "fixture_uc.txt"))
    shutil.copyfile(os.path.join(TD, "ex_tmpdir", "rep.fa"), os.path.join(OUT, "fixture_rep.fa"))
    # inputs and byte-compared outputs of the reference's trimming tests (test_main_pytest.py:378-397): data files
    for fn in ("4774-1-MSITS3_R1.fastq.gz", "4774-1-MSITS3_R2.fastq.gz"):
        shutil.copyfile(os.path.join(TD, fn), os.path.join(OUT, fn))
    shutil.copyfile(os.path.join(TD, "ex_tmpdir", "seq.fq.gz"), os.path.join(OUT, "seq.fq.gz"))
    for fn in ("t2_r1.fq", "t2_r2.fq"):
        with open(os.path.join(TD, fn), "rb") as f, gzip.GzipFile(os.path.join(OUT, fn + ".gz"), "wb", mtime=0) as g:
            g.write(f.read())

    # --- P1: matchdict as the reference's Dedup.parse builds it
    dd = S.Dedup(uc_file=os.path.join(TD, "ex_tmpdir", "uc.txt"), rep_file="", seq_file="")
    with open(os.path.join(OUT, "matchdict.json"), "w") as f:
        json.dump(dd.matchdict, f, indent=0, sort_keys=True)
    assert len(dd.matchdict) == 227

    # --- P5: golden trim coordinates from the byte-compared outputs t2_r1.fq / t2_r2.fq
    raw1 = {r[0]: r for r in read_fastq(os.path.join(TD, "4774-1-MSITS3_R1.fastq"))}
    raw2 = {r[0]: r for r in read_fastq(os.path.join(TD, "4774-1-MSITS3_R2.fastq"))}
    t1 = read_fastq(os.path.join(TD, "t2_r1.fq"))
    t2 = {r[0]: r for r in read_fastq(os.path.join(TD, "t2_r2.fq"))}
    mlen = {r[0]: len(r[1]) for r in merged}
    rows = []
    for rid, s, q in t1:
        r1 = raw1[rid]
        # start: unique offset at which the trimmed (seq, qual) sits inside raw R1
        cands = [i for i in range(len(r1[1]) - len(s) + 1) if r1[1][i:i + len(s)] == s and r1[2][i:i + len(s)] == q]
        assert len(cands) == 1, (rid, cands)
        start = cands[0]
        s2, q2 = t2[rid][1], t2[rid][2]
        r2 = raw2[rid]
        c2 = [i for i in range(len(r2[1]) - len(s2) + 1) if r2[1][i:i + len(s2)] == s2 and r2[2][i:i + len(s2)] == q2]
        assert len(c2) == 1, (rid, c2)
        r2start = c2[0]
        rep = dd.matchdict[rid]
        tlen = mlen[rep]
        stop = tlen - r2start
        # cross-check with the R1/R2 slice arithmetic of SeqSample.py:639-661
        r2end = tlen - start
        assert (len(s2) == r2end - r2start) or (r2end > len(r2[1])), rid
        rows.append((rid, rep, start, stop, tlen))
    assert len(rows) == 226
    byrep = {}
    for rid, rep, a, b, t in rows:
        assert byrep.setdefault(rep, (a, b, t)) == (a, b, t)
    tot = sum(len(dict((m[0], m[1]) for m in merged)[rid][a:b]) for rid, rep, a, b, t in rows)
    assert tot == 42637, tot
    with open(os.path.join(OUT, "fungi_its2_coords.tsv"), "w") as f:
        f.write("read\trep\tstart\tstop\ttlen\n")
        for r in rows:
            f.write("%s\t%s\t%d\t%d\t%d\n" % r)

    # --- P3: profile selection
    names = {}
    tmp = tempfile.mkdtemp()
    present = [t for t, fn in D.taxa_dict.items() if os.path.exists(os.path.join(D.ROOT_DIR, "ITSx_db", "HMMs", fn))]
    for taxa in present + ["All"]:
        for region in ("ITS2", "ITS1", "ALL"):
            p = M.create_runtime_hmm(taxa, region, tmp)
            with open(p) as f:
                names["%s|%s" % (taxa, region)] = [ln[6:].strip() for ln in f if ln.startswith("NAME  ")]
    with open(os.path.join(OUT, "runtime_hmm_names.json"), "w") as f:
        json.dump({"taxa_dict": D.taxa_dict, "names": names}, f)
    shutil.rmtree(tmp)

    # --- P4: ItsPosition semantics on synthetic domtbl rows
    rng = random.Random(7)
    cases = []
    tmp = tempfile.mkdtemp()
    prof_names = ["3_End_x_58S_a", "3_End_x_58S_b", "4_Start_x_LSU_a", "4_Start_x_LSU_b", "1_SSU_end", "2_58S_start"]
    for ci in range(12):
        lines = ["# header line"]
        seqs = ["s%d" % i for i in range(6)]
        for pn in prof_names:
            for sq in seqs:
                if rng.random() < 0.6:
                    ndom = rng.choice([1, 1, 1, 2])
                    for d in range(ndom):
                        sc = rng.choice([10.0, 12.5, 33.3, 33.3, 50.1, 52.2, 8.9])
                        fr = rng.randint(1, 200)
                        to = fr + rng.randint(20, 60)
                        tlen = 300 + int(sq[1:])
                        row = [sq, "-", str(tlen), pn, "-", "45", "1e-10", "%.1f" % (sc + 1), "0.1", str(d + 1),
                               str(ndom), "1e-9", "1e-9", "%.1f" % sc, "0.0", "1", "45", str(fr + 1), str(to - 1),
                               str(fr), str(to), "0.95", "-"]
                        lines.append(" ".join(row))
        text = "\n".join(lines) + "\n"
        p = os.path.join(tmp, "d%d.txt" % ci)
        with open(p, "w") as f:
            f.write(text)
        for region in ("ITS2", "ITS1", "ALL"):
            ip = S.ItsPosition(p, region)
            pos = {}
            for sq in seqs:
                try:
                    pos[sq] = list(ip.get_position(sq))
                except KeyError:
                    pos[sq] = "KeyError"
            cases.append({"domtbl": text, "region": region, "ddict": ip.ddict, "positions": pos})
    shutil.rmtree(tmp)
    with open(os.path.join(OUT, "itsposition_cases.json"), "w") as f:
        json.dump(cases, f)

    # --- XXH64 known answers
    import xxhash
    kat = []
    for n in [0, 1, 3, 4, 7, 8, 15, 16, 31, 32, 33, 63, 64, 75, 76, 100, 150, 255, 1000]:
        data = bytes(rng.getrandbits(8) for _ in range(n))
        for seed in (0, 1, 20240405):
            kat.append({"hex": data.hex(), "seed": seed, "h": xxhash.xxh64(data, seed=seed).intdigest()})
    with open(os.path.join(OUT, "xxh64_kat.json"), "w") as f:
        json.dump(kat, f)

    # --- profile DATA needed on the GPU box (no /root/reference there)
    hmmdir = os.path.join(D.ROOT_DIR, "ITSx_db", "HMMs")
    with open(os.path.join(hmmdir, "T.hmm"), "rb") as f, gzip.GzipFile(os.path.join(OUT, "T.hmm.gz"), "wb", mtime=0) as g:
        g.write(f.read())
    blocks = open(os.path.join(hmmdir, "T.hmm")).read().split("//\n")
    keep = []
    want = {"3_": 3, "4_": 3, "1_": 1, "2_": 1}
    for b in blocks:
        if "NAME  " not in b:
            continue
        nm = b.split("NAME  ")[1].split("\n")[0].strip()
        if want.get(nm[:2], 0) > 0:
            want[nm[:2]] -= 1
            keep.append(b + "//\n")
    g_blocks = open(os.path.join(hmmdir, "G.hmm")).read().split("//\n")
    for b in g_blocks:     # the two short (M=25, M=11) models exercise Q != 12
        if "LENG  25" in b or "LENG  11" in b:
            keep.append(b + "//\n")
    with open(os.path.join(OUT, "mini.hmm"), "w") as f:
        f.write("".join(keep))
    # f4: the orientation reference ships with the reference package (DATA): vsearch --orient --db (SeqSample.py:59)
    shutil.copyfile(os.path.join(D.ROOT_DIR, "universal_orient_ref_clean.fasta.gz"), os.path.join(OUT, "universal_orient_ref_clean.fasta.gz"))
    # --taxa All --region ITS2 (BASELINE configs[3]): the runtime file create_runtime_hmm writes, 814 profiles (F.hmm absent)
    tmp = tempfile.mkdtemp()
    with open(M.create_runtime_hmm("All", "ITS2", tmp), "rb") as f, gzip.GzipFile(os.path.join(OUT, "all_its2.hmm.gz"), "wb", mtime=0) as g:
        g.write(f.read())
    shutil.rmtree(tmp)
    print("goldens written to", OUT)


if __name__ == "__main__":
    main()
