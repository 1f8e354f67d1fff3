"""``kvarq.fastq`` -- the names the engine and its callers resolve here (csrc/workhorse.c:1598-1600 looks up
``FastqFileFormatException`` at import; kvarq/analyse.py and kvarq/cli.py use ``Fastq``)."""
from kvarq_amd.fastq import FastqFileFormatException, Fastq, Q2A, ASCII      # noqa: F401
