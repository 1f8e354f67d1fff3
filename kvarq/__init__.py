"""The ``kvarq`` import names of the reference, bound to the MI355X engine: what a checkout of KvarQ gains when
``kvarq/engine.so`` (the C extension built from csrc/workhorse.c, setup.py:31-35) is replaced by this package's
``engine.py``.  Only the three modules the hot path touches live here -- ``kvarq.engine``, ``kvarq.fastq``
(the exception class the engine raises, the ``Fastq`` probe) and ``kvarq.log`` (the logger it reports through);
the rest of the reference's ``kvarq`` package (CLI, GUI, testsuites, genes) is out of scope and
stays the reference's own.  See INTEGRATION.md."""
from kvarq_amd import VERSION  # noqa: F401
