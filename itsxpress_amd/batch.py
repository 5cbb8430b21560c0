"""Per-sample batching for callers that hold MANY samples.

The QIIME 2 plugin walks its manifest and runs the whole path once per sample
(itsxpress/q2_itsxpress.py:273-333: `_set_fastqs_and_check` -> `sobj.deduplicate` -> `sobj._search` ->
`ItsPosition` / `Dedup` -> writer, inside `for sample in samples.itertuples()`).  Amplicon samples are small
(10^4..10^5 reads), so each of those runs leaves most of an MI355X idle and pays the launch and transfer
latencies again.  `SampleBatch` takes the same `SeqSample` objects and runs the two engine steps ONCE for all of
them -- one read set, one pass of every kernel -- while every sample keeps exactly the results of its own run:

* reads are dereplicated within their sample only (representative = first occurrence inside the sample),
* hmmsearch's domZ (reported targets per profile, the factor in the domain E-value threshold) is counted per
  (sample, profile),
* `uc.txt`, `rep.fa`, `domtbl.txt` are written per sample, byte-identical to the files a run of that sample
  alone writes (tests/test_gpu_batch.py), and each `SeqSample`'s `uc_file` / `rep_file` / `dom_file` point at
  them, so the reference's `ItsPosition(...)` / `Dedup(...)` constructors and writers downstream stay as they are.

Nothing here computes: the grouping, the counters and the writers live behind the C ABI
(`itsx_load_reads_files`, `itsx_set_samples`, `itsx_select_sample`; include/itsx_hip.h).
"""
import logging
import os
from typing import List, Optional, Sequence, Union

import numpy as np

from .engine import Engine
from ._lib import EngineError
from .SeqSample import _REGION_PREFIX, _winners_from_env


class SampleBatch:
    """Runs `deduplicate()` and `_search()` of many SeqSample objects as one engine pass.

    samples: objects with `seq_file` and `tempdir` (the mirror's or the reference's SeqSample*, after
    `_merge_reads` / `orient_reads` where those apply).  Method names and arguments follow SeqSample."""

    def __init__(self, samples: Sequence, engine: Optional[Engine] = None, subdirs: Optional[Sequence[str]] = None) -> None:
        self.samples = list(samples)
        if not self.samples:
            raise ValueError("SampleBatch needs at least one sample")
        for s in self.samples:
            if getattr(s, "seq_file", None) is None:
                raise ValueError("every sample needs its seq_file before batching (merge paired reads first)")
        self.engine = engine if engine is not None else Engine()
        # the plugin gives every sample the same tempdir and overwrites uc.txt / rep.fa / domtbl.txt per sample;
        # a batch holds them all at once, so each sample writes into its own sub-directory
        self.subdirs = list(subdirs) if subdirs is not None else ["sample_%04d" % i for i in range(len(self.samples))]
        self.counts: Optional[np.ndarray] = None       # reads per sample
        self.first: Optional[np.ndarray] = None        # index of each sample's first read in the batch

    def _dir(self, i: int) -> str:
        d = os.path.join(self.samples[i].tempdir, self.subdirs[i])
        os.makedirs(d, exist_ok=True)
        return d

    # -- a1 for all samples ------------------------------------------------------------------
    def deduplicate(self, threads: Union[int, str] = 1) -> None:
        """`vsearch --fastx_uniques ... --strand both` of every sample (SeqSample.py:93-131), one device pass."""
        try:
            eng = self.engine
            self.counts = np.asarray(eng.load_reads_files([s.seq_file for s in self.samples]), np.int64)
            self.first = np.concatenate([[0], np.cumsum(self.counts)[:-1]]).astype(np.int64)
            n = eng.derep(strand_both=True, minseqlength=1)      # SeqSample.deduplicate's value (SeqSample.py:96 passes no --minseqlength; the mirror uses 1): a batch and a solo run must agree on short reads
            for i, s in enumerate(self.samples):
                d = self._dir(i)
                s.uc_file = os.path.join(d, "uc.txt")
                s.rep_file = os.path.join(d, "rep.fa")
                eng.select_sample(i)
                eng.write_uc(s.uc_file)
                eng.write_rep_fasta(s.rep_file)
            eng.select_sample(-1)
            logging.info("itsx_hip batch derep: %d samples, %d reads -> %d unique sequences", len(self.samples), eng.n_reads, n)
        except EngineError as e:
            logging.exception("Could not perform dereplication with the HIP engine: %s", e)
            raise e
        except FileNotFoundError as f:
            logging.error("The HIP engine or its input was not found")
            raise f

    def cluster(self, threads: Union[int, str], cluster_id: float = 0.995) -> None:
        """cluster_id == 1.0 is dereplication (main.py:534-537).  Greedy clustering below 1.0 is sequential per sample
        and is not batched: run `sobj.cluster(...)` per sample for that."""
        if float(cluster_id) == 1.0:
            return self.deduplicate(threads=threads)
        raise EngineError(-5, "cluster_id < 1 is not batched across samples; call SeqSample.cluster per sample")

    # -- a4 for all samples ------------------------------------------------------------------
    def _search(self, hmmfile: str, threads: Union[int, str] = 1) -> None:
        """`hmmsearch --domtblout ... -T 10 --F1 1e-6 --F2 1e-6 --F3 1e-6` of every sample's rep.fa (SeqSample.py:178-225)."""
        try:
            eng = self.engine
            if self.counts is None:
                raise EngineError(-1, "deduplicate() the batch before _search()")
            eng.load_profiles(path=hmmfile)
            # ITSXPRESS_DOMTBL=winners (SeqSample._search): the lazy search, every sample's domtbl.txt = the rows its ItsPosition ends up with
            winners = _winners_from_env()
            if winners:
                eng.set_rows_mode("lazy")
            try:
                eng.search(T=10.0, F1=1e-6, F2=1e-6, F3=1e-6)
                eng.finalize(domE=10.0)
                eng.set_kept_rows(winners)
                for i, s in enumerate(self.samples):
                    s.dom_file = os.path.join(self._dir(i), "domtbl.txt")
                    eng.select_sample(i)
                    eng.write_domtbl(s.dom_file)
                eng.select_sample(-1)
            finally:
                if winners:
                    eng.set_kept_rows(False)
                    eng.set_rows_mode(None)
        except EngineError as e:
            logging.exception("Could not perform ITS identification with the HIP engine: %s", e)
            raise e
        except FileNotFoundError as f:
            logging.error("The HIP engine or the HMM file was not found")
            raise f

    # -- array fast path ----------------------------------------------------------------------
    def trim_coordinates(self, region: str) -> List[tuple]:
        """Per sample: (start, stop, tlen, in_ddict) arrays over that sample's reads in file order; -1 = None."""
        left, right = _REGION_PREFIX[region]
        a = self.engine.trim_coords(left, right)
        out = []
        for i in range(len(self.samples)):
            lo, hi = int(self.first[i]), int(self.first[i] + self.counts[i])
            out.append(tuple(x[lo:hi] for x in a))
        return out
