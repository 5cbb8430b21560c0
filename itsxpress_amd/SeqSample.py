"""Host-side mirror of the reference's operator interface for the hot path.

Same class names, method names, argument meaning, attributes and error behaviour as
itsxpress/SeqSample.py (SeqSample:18-225, ItsPosition:368-498, Dedup:501-562), but where the
reference shells out to `vsearch` / `hmmsearch` and then re-parses their text output, these
methods drive the in-process HIP engine (itsxpress_amd.engine.Engine -> libitsx_hip.so).

File-compatible: deduplicate()/cluster()/_search() still create tempdir/uc.txt, rep.fa and
domtbl.txt (users pass --keeptemp; ItsPosition/Dedup of the *reference* can read them).
Array fast path: ItsPosition.from_engine / Dedup.from_engine / SeqSample.trim_coordinates
skip the text round trip.

Two switches, both read from the environment when the method runs (or set as attributes of the sample object):
  ITSXPRESS_GPUS=N   (`sobj.gpus`)  one sample spread over the N GPUs of a node: N worker processes behind the same methods
                     (itsxpress_amd/multi.py: exact global dereplication, summed domZ, files byte-identical to one GPU's);
  ITSXPRESS_ARRAYS=1 (`sobj.fast`)  the hot path hands ARRAYS to the consumers instead of text files: deduplicate / _search write
                     nothing, the search runs the lazy domain stage (pairs that cannot win ItsPosition's argmax stop after their
                     Forward score), and `uc_file` / `dom_file` are EngineTable tokens that this module's Dedup / ItsPosition
                     accept in place of paths -- the reference's own call sequence (main.py:534-554, 626-638) runs unchanged.

Either side of the path is mirrored too: read orientation (orient_reads), paired-end
merging (_merge_reads), the trimmed-FASTQ writers (Dedup.create_*), and many samples as one batch
(itsxpress_amd/batch.py).
"""
import logging
import os
from typing import Any, Dict, Optional, Tuple, Union

from .engine import Engine
from ._lib import EngineError
from .definitions import maxmismatches

logger = logging.getLogger(__name__)

_REGION_PREFIX = {"ITS2": ("3_", "4_"), "ITS1": ("1_", "2_"), "ALL": ("1_", "4_")}


class EngineTable(str):
    """What `uc_file` / `dom_file` hold in arrays mode: a path-like token (the file it stands for is NOT written) that carries the
    engine whose arrays replace the file.  This module's Dedup / ItsPosition take it where the reference's take a path."""

    def __new__(cls, path, engine, kind):
        self = super().__new__(cls, path)
        self.engine, self.kind = engine, kind
        return self


def _new_engine(gpus=None, path=None, fast=False):
    """one Engine, or -- ITSXPRESS_GPUS > 1 -- N workers behind the same interface (started before this process touches a GPU)"""
    from .multi import MultiEngine, gpus_from_env
    from .stream import StreamEngine, stream_wanted
    n = int(gpus) if gpus else gpus_from_env()
    if n > 1:
        return MultiEngine(n)
    # ITSXPRESS_STREAM=1 (or, in arrays mode, a large input file and no ITSXPRESS_STREAM=0): file-order chunks of one FASTQ, scored
    # while the rest of the file is still being inflated
    return StreamEngine() if stream_wanted(path, fast) else Engine()


def _fast_from_env():
    return os.environ.get("ITSXPRESS_ARRAYS", "").strip() not in ("", "0")


def _winners_from_env():
    return os.environ.get("ITSXPRESS_DOMTBL", "").strip().lower() == "winners"


class SeqSample:
    """Base class: dereplicate -> search, as the reference's SeqSample (SeqSample.py:18-225)."""

    def __init__(self, fastq: str, tempdir: str) -> None:
        self.tempdir: str = tempdir
        self.fastq: str = fastq
        self.uc_file: Optional[str] = None
        self.rep_file: Optional[str] = None
        self.dom_file: Optional[str] = None
        self.seq_file: Optional[str] = None
        self.r1: Optional[str] = None
        self.fastq2: Optional[str] = None
        self._engine: Optional[Engine] = None
        self.gpus: Optional[int] = None          # None: ITSXPRESS_GPUS (default 1)
        self.fast: Optional[bool] = None         # None: ITSXPRESS_ARRAYS (default off)
        self.domtbl: Optional[str] = None        # None: ITSXPRESS_DOMTBL ("winners": files as ever, domtbl.txt = the rows that can win)

    # -- engine plumbing -------------------------------------------------------------
    @property
    def engine(self) -> Engine:
        if self._engine is None:
            try:
                self._engine = _new_engine(getattr(self, "gpus", None), path=self.seq_file or self.fastq, fast=self._is_fast())
            except FileNotFoundError:
                logger.error("The HIP engine (libitsx_hip.so) was not found; build it first")
                raise
        return self._engine

    def _is_fast(self) -> bool:
        f = getattr(self, "fast", None)
        return _fast_from_env() if f is None else bool(f)

    def _load_reads(self) -> None:
        eng = self.engine
        if getattr(self, "_reads_loaded_from", None) != self.seq_file:
            eng.load_reads_file(self.seq_file)
            self._reads_loaded_from = self.seq_file

    # -- f4 --------------------------------------------------------------------------
    def orient_reads(self, threads: Union[int, str] = 1) -> None:
        """Replaces `vsearch --orient FASTQ --db universal_orient_ref_clean.fasta.gz --fastqout oriented.fq`
        (SeqSample.py:48-91): 12-mer counts on both strands on the GPU, then the oriented FASTQ; the sample's
        fastq / seq_file / r1 point at it afterwards, as in the reference."""
        try:
            from .definitions import ROOT_DIR
            from .trim import write_oriented_fastq
            orient_ref = os.path.join(ROOT_DIR, "universal_orient_ref_clean.fasta.gz")
            oriented_fastq = os.path.join(self.tempdir, "oriented.fq")
            os.makedirs(self.tempdir, exist_ok=True)
            self.engine.orient_load_db(orient_ref)
            strand, _, _ = self.engine.orient_file(self.fastq)
            write_oriented_fastq(self.fastq, oriented_fastq, strand)
            self.fastq = oriented_fastq
            self.seq_file = oriented_fastq
            self.r1 = oriented_fastq
            self._reads_loaded_from = None
        except EngineError as e:
            logging.exception("Could not orient reads with the HIP engine: %s", e)
            raise e
        except FileNotFoundError as f:
            logging.error("The HIP engine, the reads or the orientation reference were not found")
            raise f

    # -- a1 --------------------------------------------------------------------------
    def deduplicate(self, threads: Union[int, str] = 1) -> None:
        """Replaces `vsearch --fastx_uniques ... --strand both` (SeqSample.py:93-131)."""
        try:
            self.uc_file = os.path.join(self.tempdir, "uc.txt")
            self.rep_file = os.path.join(self.tempdir, "rep.fa")
            self._load_reads()
            # vsearch's --minseqlength default is 32 for the clustering / derep_* / usearch_global commands and 1 for the
            # others, --fastx_uniques among them: only empty reads vanish from uc.txt here
            n = self.engine.derep(strand_both=True, minseqlength=1)
            if self._is_fast():                     # arrays mode: Dedup reads the engine, nothing is written
                self.uc_file = EngineTable(self.uc_file, self.engine, "uc")
            else:
                self.engine.write_uc(self.uc_file)
                self.engine.write_rep_fasta(self.rep_file)
            if n is not None:                       # (a streaming engine defers the work until the search knows the profiles)
                logging.info("itsx_hip derep: %d reads -> %d unique sequences", self.engine.n_reads, n)
        except EngineError as e:
            logging.exception("Could not perform dereplication with the HIP engine: %s", e)
            raise e
        except FileNotFoundError as f:
            logging.error("The HIP engine or its input was not found")
            raise f

    # -- a2 --------------------------------------------------------------------------
    def cluster(self, threads: Union[int, str], cluster_id: float = 0.995) -> None:
        """Replaces `vsearch --cluster_size ... --id X --strand both` (SeqSample.py:133-176): greedy centroid
        clustering on the GPU (k_cluster.hip; the test oracle restates the procedure, parity unpinned).
        cluster_id == 1.0 is exact dereplication, as in main.py:534-537."""
        try:
            self.uc_file = os.path.join(self.tempdir, "uc.txt")
            self.rep_file = os.path.join(self.tempdir, "rep.fa")
            self._load_reads()
            cid = float(cluster_id)
            if cid < 1.0 and (getattr(self.engine, "world", 1) > 1 or getattr(self.engine, "deferred", False)):
                # greedy clustering is sequential by definition (a query sees every centroid before it): one GPU runs it, in one piece
                logging.warning("cluster_id < 1 does not shard over GPUs or chunks: this sample runs on one GPU (ITSXPRESS_GPUS / ITSXPRESS_STREAM ignored)")
                self._engine.close()
                self._engine = Engine()
                self._reads_loaded_from = None
                self._load_reads()
            self.engine.cluster(cid, strand_both=True)
            if self._is_fast():
                self.uc_file = EngineTable(self.uc_file, self.engine, "uc")
            else:
                self.engine.write_uc(self.uc_file)
                self.engine.write_rep_fasta(self.rep_file)
        except EngineError as e:
            logging.exception("Could not perform clustering with the HIP engine: %s", e)
            raise e
        except FileNotFoundError as f:
            logging.error("The HIP engine or its input was not found")
            raise f

    # -- a4 --------------------------------------------------------------------------
    def _search(self, hmmfile: str, threads: Union[int, str]) -> None:
        """Replaces `hmmsearch --domtblout ... -T 10 --F1 1e-6 --F2 1e-6 --F3 1e-6` (SeqSample.py:178-225)."""
        try:
            self.dom_file = os.path.join(self.tempdir, "domtbl.txt")
            eng = self.engine
            if not getattr(eng, "deferred", False) and eng.n_unique == 0 and self.rep_file and os.path.exists(self.rep_file) and eng.n_reads == 0:
                # _search called on a rep.fa produced elsewhere (the reference's tests do this)
                eng.load_reads_file(self.rep_file)
                eng.derep(strand_both=False, minseqlength=0)
            eng.load_profiles(path=hmmfile)
            fast = self._is_fast()
            dt = getattr(self, "domtbl", None)
            winners = (not fast) and ((dt or "").lower() == "winners" if dt is not None else _winners_from_env())
            # file-compatible: every domain row stays (domtbl.txt); arrays mode: the lazy domain stage, coordinates only;
            # ITSXPRESS_DOMTBL=winners: the files of the first with the search of the second -- domtbl.txt holds, per target and side,
            # the row ItsPosition.parse ends up with: the reference's own parser reads the same dictionary out of it as out of
            # hmmsearch's full table (SeqSample.py:400-461 keeps the first strictly greatest score)
            # (a streaming engine defers its work: asking it for a count would run the load stage apart from the search)
            nu = 0 if (fast or getattr(eng, "deferred", False)) else int(getattr(eng, "n_unique", 0) or 0)
            if not fast and not winners and nu > 1000000:
                logging.info("itsx_hip search: %d unique sequences in file-compatible mode -- every (sequence, profile) pair is evaluated and "
                             "domtbl.txt gets about %d rows (~%.0f GB); ITSXPRESS_DOMTBL=winners writes only the rows ItsPosition can end up with "
                             "(same files, same trimmed reads, the search several times faster), ITSXPRESS_ARRAYS=1 keeps the tables in the engine "
                             "altogether (INTEGRATION.md 3b)", nu, nu * 136, nu * 136 * 200 / 1e9)
            eng.set_rows_mode("lazy" if (fast or winners) else "full")
            eng.search(T=10.0, F1=1e-6, F2=1e-6, F3=1e-6)
            eng.finalize(domE=10.0)
            if fast:
                self.dom_file = EngineTable(self.dom_file, eng, "domtbl")
            else:
                eng.set_kept_rows(winners)
                eng.write_domtbl(self.dom_file)
        except EngineError as e:
            logging.exception("Could not perform ITS identification with the HIP engine: %s", e)
            raise e
        except FileNotFoundError as f:
            logging.error("The HIP engine or the HMM file was not found")
            raise f

    def plan_output(self, outfile: str, region: str, gzipped: bool = False, zstd_file: bool = False, trim_ccs: bool = False) -> None:
        """Optional, before deduplicate(): where the trimmed single-end reads will go.  A streaming engine (ITSXPRESS_STREAM=1 with
        ITSXPRESS_ARRAYS=1) then writes them WHILE it scores -- Dedup.create_trimmed_seqs, called later with the same arguments as in
        the reference (main.py:626-638), only waits for the last bytes; every other engine ignores the plan."""
        eng = self.engine
        if self._is_fast() and hasattr(eng, "plan_output"):
            left, right = _REGION_PREFIX[region]
            eng.plan_output(outfile, left, right, gzipped=gzipped, zstd_file=zstd_file, trim_ccs=trim_ccs)

    def plan_output_paired(self, outfile1: str, outfile2: str, region: str, gzipped: bool = False, zstd_file: bool = False) -> None:
        """Optional, before deduplicate(): where the two mates' trimmed files will go (Dedup.create_paired_trimmed_seqs' arguments,
        main.py:556-624).  A streaming engine then writes them while it scores; every other engine ignores the plan."""
        eng = self.engine
        # (only when the files that are merged are the files that are sliced: with reversed primers the reference merges fastq2 / fastq
        # but slices fastq / fastq2, SeqSample.py:244-264 -- that sample takes the staged writer)
        if self._is_fast() and hasattr(eng, "plan_output_paired") and getattr(self, "r1", None) == self.fastq:
            left, right = _REGION_PREFIX[region]
            eng.plan_output_paired(outfile1, outfile2, left, right, gzipped=gzipped, zstd_file=zstd_file)

    # -- array fast path (a5/a6/a7 composed) ----------------------------------------------
    def trim_coordinates(self, region: str):
        """Per-read (start, stop, tlen, in_ddict) arrays straight from the device; -1 = None."""
        left, right = _REGION_PREFIX[region]
        return self.engine.trim_coords(left, right)


class SeqSampleNotPaired(SeqSample):
    """Unpaired input (SeqSample.py:228-241)."""

    def __init__(self, fastq: str, tempdir: str) -> None:
        SeqSample.__init__(self, fastq, tempdir)
        self.seq_file = self.fastq
        self.r1 = self.fastq
        self.fastq2 = None


class SeqSamplePairedNotInterleaved(SeqSample):
    """Paired input in two files (SeqSample.py:244-264)."""

    def __init__(self, fastq: str, tempdir: str, fastq2: str, reversed_primers: bool = False) -> None:
        SeqSample.__init__(self, fastq, tempdir)
        if reversed_primers:
            self.r1 = fastq2
            self.fastq2 = fastq
        else:
            self.r1 = fastq
            self.fastq2 = fastq2

    # -- f2 --------------------------------------------------------------------------
    def _merge_reads(self, threads: Union[int, str], stagger: bool = False) -> None:
        """Replaces `vsearch --fastq_mergepairs R1 --reverse R2 --fastqout seq.fq --fastq_maxdiffs 40 --fastq_maxee 2
        --fastq_qmax 93 [--fastq_allowmergestagger]` (SeqSample.py:266-365; constants definitions.py:79-82) with the
        engine's merge kernel; writes tempdir/seq.fq and points seq_file at it, as the reference does."""
        try:
            if self.r1 is None or self.fastq2 is None:
                raise ValueError("Both r1 and fastq2 paths must be defined to merge reads.")
            os.makedirs(self.tempdir, exist_ok=True)
            seq_file = os.path.join(self.tempdir, "seq.fq")
            eng = self.engine
            if self._is_fast() and not hasattr(eng, "merge_pairs_load") and hasattr(eng, "_plain"):
                # arrays mode on a streaming engine: the merged reads live in ONE context (nothing of a merged sample is inflated
                # while it is scored: the merge needs both files whole) -- that context is the sample's engine from here on
                plain = eng._plain()
                eng._plain_eng = None
                eng.close()
                self._engine = eng = plain
            if self._is_fast() and hasattr(eng, "merge_pairs_load"):
                # arrays mode: the merged reads become the engine's read set where the merge kernel left them; no seq.fq, no second parse
                n, m = eng.merge_pairs_load(self.r1, self.fastq2, maxdiffs=maxmismatches, maxee=2.0, allow_stagger=bool(stagger))
                self._reads_loaded_from = seq_file
            else:
                n, m = eng.merge_pairs_files(self.r1, self.fastq2, seq_file, maxdiffs=maxmismatches, maxee=2.0,
                                             allow_stagger=bool(stagger))
            if n >= 0:
                logging.info("%d pairs, %d merged", n, m)
            else:
                logging.info("itsx_hip: the pairs are merged chunk by chunk while the files are inflated (streamed)")
            self.seq_file = seq_file
        except EngineError as e:
            logging.exception("Could not perform read merging with the HIP engine: %s", e)
            raise e
        except FileNotFoundError as f:
            logging.error("The HIP engine or its input was not found")
            raise f


class ItsPosition:
    """ITS boundary positions per representative (mirror of SeqSample.py:368-498).

    `ddict` has the reference's shape: {seq: {"left": {"score","to_pos","from_pos"},
    "right": {...}, "tlen": n}}.  The FIRST row with a strictly greater score wins."""

    def __init__(self, domtable: Optional[str], region: str) -> None:
        self.domtable = domtable
        self._ddict: Optional[Dict[str, Any]] = {}
        self._engine = None
        self.leftprefix, self.rightprefix = _REGION_PREFIX[region]
        if isinstance(domtable, EngineTable):       # arrays mode: the engine's per-representative coordinates ARE the table
            self._engine = domtable.engine
            self._ddict = None                      # built on demand (get_position / ddict); the writers use the arrays directly
        elif domtable is not None:
            self.parse()

    @property
    def ddict(self) -> Dict[str, Any]:
        if self._ddict is None:
            self._ddict = self._ddict_from_engine()
        return self._ddict

    @ddict.setter
    def ddict(self, value) -> None:
        self._ddict = value

    def _ddict_from_engine(self) -> Dict[str, Any]:
        """The reference's dict shape from the engine's arrays: one entry per representative that has a reported row; scores are
        not kept by the lazy / compacted search (None): get_position never reads them."""
        eng = self._engine
        start, stop, tlen, ind = eng.rep_coords(self.leftprefix, self.rightprefix)
        rep_of, _, uniq_of = eng.get_derep()
        names = eng.read_names()
        seed = {}
        for r in range(len(rep_of)):
            if rep_of[r] == r:
                seed[int(uniq_of[r])] = names[r]
        out: Dict[str, Any] = {}
        for u in range(len(ind)):
            if not ind[u]:
                continue
            e: Dict[str, Any] = {}
            if start[u] >= 0:
                e["left"] = {"score": None, "to_pos": int(start[u]), "from_pos": None}
            if stop[u] >= 0:
                e["right"] = {"score": None, "to_pos": None, "from_pos": int(stop[u]) + 1}
            if tlen[u] >= 0:
                e["tlen"] = int(tlen[u])
            out[seed[u]] = e
        return out

    def _score(self, sequence: str, stype: str, score: float, from_pos: int, to_pos: int, tlen: int) -> None:
        entry = self.ddict[sequence]
        if stype in entry:
            if score > entry[stype]["score"]:
                entry[stype].update(score=score, to_pos=to_pos, from_pos=from_pos)
        else:
            entry[stype] = {"score": score, "to_pos": to_pos, "from_pos": from_pos}
            entry["tlen"] = tlen

    def parse(self) -> None:
        try:
            with open(self.domtable, "r") as f:
                for line in f:
                    if line.startswith("#"):
                        continue
                    ll = line.split()
                    sequence, tlen, hmmprofile = ll[0], int(ll[2]), ll[3]
                    score, from_pos, to_pos = float(ll[13]), int(ll[19]), int(ll[20])
                    if sequence not in self.ddict:
                        self.ddict[sequence] = {}
                    if hmmprofile.startswith(self.leftprefix):
                        self._score(sequence, "left", score, from_pos, to_pos, tlen)
                    elif hmmprofile.startswith(self.rightprefix):
                        self._score(sequence, "right", score, from_pos, to_pos, tlen)
        except Exception as e:
            logging.error("Exception occurred when parsing HMMSearch results")
            raise e

    @classmethod
    def from_engine(cls, engine: Engine, region: str, names) -> "ItsPosition":
        """Build the same ddict from the engine's domain rows (no text round trip).
        `names[i]` is the label of unique representative i."""
        self = cls(None, region)
        prof = engine.profile_names()
        for d in engine.domains():
            if not d["dom_reported"]:
                continue
            seq = names[int(d["rep"])]
            if seq not in self.ddict:
                self.ddict[seq] = {}
            score = float("%.1f" % d["bitscore"])
            p = prof[int(d["prof"])]
            if p.startswith(self.leftprefix):
                self._score(seq, "left", score, int(d["ienv"]), int(d["jenv"]), int(d["tlen"]))
            elif p.startswith(self.rightprefix):
                self._score(seq, "right", score, int(d["ienv"]), int(d["jenv"]), int(d["tlen"]))
        return self

    def get_position(self, sequence: str) -> Tuple[Optional[int], Optional[int], Optional[int]]:
        try:
            entry = self.ddict[sequence]
        except KeyError:
            logging.debug("No ITS stop or start sites were identified for sequence {}, skipping.".format(sequence))
            raise KeyError
        start = int(entry["left"]["to_pos"]) if "left" in entry else None
        stop = int(entry["right"]["from_pos"]) - 1 if "right" in entry else None
        tlen = int(entry["tlen"]) if "tlen" in entry else None
        return (start, stop, tlen)


class Dedup:
    """read -> representative map (mirror of SeqSample.py:501-562; trimming/writing is out of scope)."""

    def __init__(self, uc_file: Optional[str], rep_file: str, seq_file: str,
                 fastq: Optional[str] = None, fastq2: Optional[str] = None) -> None:
        self.matchdict: Optional[Dict[str, str]] = None
        self.uc_file = uc_file
        self.rep_file = rep_file
        self.seq_file = seq_file
        self.fastq = fastq
        self.fastq2 = fastq2
        self._engine = None
        if isinstance(uc_file, EngineTable):        # arrays mode: the read -> representative map stays in the engine
            self._engine = uc_file.engine
        elif uc_file is not None:
            self.parse()

    def _matchdict_from_engine(self) -> Dict[str, str]:
        names = self._engine.read_names()
        rep_of, _, _ = self._engine.get_derep()
        return {names[i]: names[int(r)] for i, r in enumerate(rep_of) if r >= 0}

    def __getattribute__(self, name):
        # matchdict of an engine-backed Dedup is built the first time somebody asks for it (10 M entries are not free)
        if name == "matchdict":
            d = object.__getattribute__(self, "__dict__")
            if d.get("matchdict") is None and d.get("_engine") is not None:
                d["matchdict"] = object.__getattribute__(self, "_matchdict_from_engine")()
        return object.__getattribute__(self, name)

    def parse(self) -> None:
        try:
            with open(self.uc_file, "r") as f:
                self.matchdict = {}
                for line in f:
                    ll = line.split()
                    if ll[0] == "S":
                        self.matchdict[ll[8]] = ll[8]
                    elif ll[0] == "H":
                        self.matchdict[ll[8]] = ll[9]
        except Exception as e:
            logging.exception("Could not parse the '.uc' file.")
            raise e

    # -- f1: the consumers of the coordinates, native writers (SeqSample.py:713-790, 886-949) ------------
    def create_trimmed_seqs(self, outfile: str, gzipped: bool, zstd_file: bool, itspos: "ItsPosition",
                            wri_file: bool, tempdir: str = "", trim_ccs: bool = False) -> None:
        """Single-end: write seq_file's records trimmed to record[start:stop] (same filter as the reference)."""
        from .trim import coords_from_dicts, read_names, write_trimmed_fastq
        if not wri_file:
            return
        if self._engine is not None and getattr(itspos, "_engine", None) is self._engine:
            planned = getattr(self._engine, "output_planned", None)
            if planned is not None and planned(outfile, itspos.leftprefix, itspos.rightprefix, gzipped, zstd_file, trim_ccs):
                self._engine.finish_output()        # a streaming engine has written it while it scored (SeqSample.plan_output)
                return
            # arrays mode: per-read coordinates straight from the engine, in the order of seq_file's records (the engine read them from it)
            start, stop, _, _ = self._engine.trim_coords(itspos.leftprefix, itspos.rightprefix)
            if not os.path.exists(self.seq_file) and getattr(self._engine, "_last_merge", None) is not None:
                # a paired sample whose merged reads never left the engine (_merge_reads in arrays mode writes no seq.fq) and whose caller
                # wants the MERGED reads trimmed (main.py:596-624: --fastq2 without --outfile2): the merged records are written now
                logging.info("itsx_hip: writing the merged reads to %s for the single merged output", self.seq_file)
                self._engine.write_merged_fastq(self.seq_file)
            write_trimmed_fastq(self.seq_file, outfile, start, stop, gzipped=gzipped, trim_ccs=trim_ccs, zstd_file=zstd_file)
            return
        names = read_names(self.seq_file)           # the native writer's own record parser: names and records cannot disagree
        start, stop, _ = coords_from_dicts(names, self.matchdict, itspos)
        write_trimmed_fastq(self.seq_file, outfile, start, stop, gzipped=gzipped, trim_ccs=trim_ccs, zstd_file=zstd_file)

    def create_paired_trimmed_seqs(self, outfile1: str, outfile2: str, gzipped: bool, zstd_file: bool,
                                   itspos: "ItsPosition", wri_file: bool, trim_ccs: bool = False) -> None:
        """Paired: slice the ORIGINAL R1/R2 records with the merged read's coordinates."""
        from .trim import coords_from_dicts, write_trimmed_paired
        if self.fastq is None or self.fastq2 is None:
            raise ValueError("Both fastq and fastq2 paths must be defined to create paired trimmed sequences.")
        if not wri_file:
            return
        if self._engine is not None and getattr(itspos, "_engine", None) is self._engine:
            planned = getattr(self._engine, "output_planned_paired", None)
            if planned is not None and not trim_ccs and planned(outfile1, outfile2, itspos.leftprefix, itspos.rightprefix, gzipped, zstd_file):
                self._engine.finish_output()        # a streaming engine has written both while it scored (SeqSample.plan_output_paired)
                return
            start, stop, tlen, _ = self._engine.trim_coords(itspos.leftprefix, itspos.rightprefix)
            write_trimmed_paired(self.fastq, self.fastq2, outfile1, outfile2, self._engine.read_names_raw(), start, stop, tlen,
                                 gzipped=gzipped, trim_ccs=trim_ccs, zstd_file=zstd_file)
            return
        names = list(self.matchdict.keys())
        start, stop, tlen = coords_from_dicts(names, self.matchdict, itspos)
        write_trimmed_paired(self.fastq, self.fastq2, outfile1, outfile2, names, start, stop, tlen,
                             gzipped=gzipped, trim_ccs=trim_ccs, zstd_file=zstd_file)

    @classmethod
    def from_engine(cls, engine: Engine, names) -> "Dedup":
        """matchdict from the engine's arrays (names[i] = label of read i)."""
        self = cls(None, "", "")
        rep_of, _, _ = engine.get_derep()
        self.matchdict = {names[i]: names[int(r)] for i, r in enumerate(rep_of) if r >= 0}
        return self


def install():
    """Swap the engine into an importable reference package: after this call
    `itsxpress.SeqSample.SeqSample.{deduplicate,cluster,_search,orient_reads}` and the paired sample's
    `_merge_reads` run on the GPU while the CLI,
    ItsPosition, Dedup and the QIIME 2 plugin stay untouched."""
    import itsxpress.SeqSample as ref  # noqa: the reference package must be installed

    def _eng(obj):
        if getattr(obj, "_engine", None) is None:
            obj._engine = _new_engine(getattr(obj, "gpus", None))
        return obj._engine

    ref.SeqSample.engine = property(_eng)
    ref.SeqSample._load_reads = SeqSample._load_reads
    # The reference's own ItsPosition / Dedup stay in place here and open uc.txt / domtbl.txt by PATH, so the installed methods always
    # run file-compatible -- whatever ITSXPRESS_ARRAYS says (its EngineTable tokens are understood by this package's mirror classes only)
    if _fast_from_env():
        logging.getLogger(__name__).info("ITSXPRESS_ARRAYS is ignored by install(): the reference's ItsPosition / Dedup read uc.txt and domtbl.txt")
    ref.SeqSample._is_fast = lambda self: False
    ref.SeqSample.deduplicate = SeqSample.deduplicate
    ref.SeqSample.cluster = SeqSample.cluster
    ref.SeqSample._search = SeqSample._search
    ref.SeqSample.trim_coordinates = SeqSample.trim_coordinates
    # either side of the path: orientation of CCS reads and paired-end merging
    ref.SeqSample.orient_reads = SeqSample.orient_reads
    if hasattr(ref, "SeqSamplePairedNotInterleaved"):
        ref.SeqSamplePairedNotInterleaved._merge_reads = SeqSamplePairedNotInterleaved._merge_reads
