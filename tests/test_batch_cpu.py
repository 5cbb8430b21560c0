"""Host-side checks of the sample batch that need no GPU: argument validation happens before any context is made, and
the batch mirror exposes the reference's method names."""
import inspect

import pytest

from itsxpress_amd import SeqSample
from itsxpress_amd.batch import SampleBatch


def test_batch_rejects_bad_input_before_touching_the_engine():
    with pytest.raises(ValueError):
        SampleBatch([])
    s = SeqSample("reads.fq", "/tmp")          # the base class leaves seq_file unset (paired samples set it when merging)
    with pytest.raises(ValueError):
        SampleBatch([s])


def test_batch_mirrors_the_sample_interface():
    for name in ("deduplicate", "cluster", "_search"):
        a = inspect.signature(getattr(SampleBatch, name)).parameters
        b = inspect.signature(getattr(SeqSample, name)).parameters
        assert list(a)[:len(b)] == list(b), name       # same leading arguments as the reference-shaped SeqSample methods
    assert hasattr(SampleBatch, "trim_coordinates")
