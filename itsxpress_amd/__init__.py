"""itsxpress_amd: MI355X (gfx950) engine for the ITSxpress hot path.

dereplicate reads -> score representatives against the ITSx profile HMMs -> per-read trim
coordinates, behind the reference's own SeqSample / ItsPosition / Dedup interface.
The compute lives in libitsx_hip.so (C ABI: include/itsx_hip.h); Python here is plumbing.
"""
from ._lib import EngineError, lib  # noqa: F401
from .engine import Engine, read_fastx  # noqa: F401
from .SeqSample import (Dedup, ItsPosition, SeqSample, SeqSampleNotPaired,  # noqa: F401
                        SeqSamplePairedNotInterleaved, install)
from .main import create_runtime_hmm  # noqa: F401
from .trim import write_trimmed_fastq, write_trimmed_paired  # noqa: F401

__version__ = "0.1.0"
