"""``kvarq.fastq`` counterpart, as far as the engine needs it: the exception the
scan raises on malformed records (reference kvarq/fastq.py; resolved by the C
engine at import, csrc/workhorse.c:1598-1600) and the PHRED -> ASCII helper the
callers use to derive ``Amin`` (kvarq/fastq.py:41-42, 245-247)."""


class FastqFileFormatException(Exception):
    """raised when a record does not start with '@' or its 3rd line not with '+'"""


# quality characters in ASCII order, as kvarq/fastq.py:41-42 lists them
ASCII = ''.join(chr(c) for c in range(33, 127))

# vendor variants: name -> dQ (offset of Q=0 inside ASCII), kvarq/fastq.py:44-53
VARIANTS = {'Sanger': 0, 'Solexa': 31, 'Illumina 1.3+': 31, 'Illumina 1.5+': 31, 'Illumina 1.8+': 0}


def Q2A(Q, variant='Sanger'):
    """ASCII character of PHRED score ``Q`` (kvarq/fastq.py:245-247); Q=13 on
    Sanger/Illumina 1.8+ is '.', the product default ``Amin`` (kvarq/config.py:3)"""
    return ASCII[Q + VARIANTS[variant]]
