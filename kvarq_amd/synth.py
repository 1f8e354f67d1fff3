"""
Synthetic workload of SURVEY.md section 8(d): a random 4.41 Mbp genome, the
MTBC-shaped target table cut from it, and FastQ reads sampled from it.

Everything is derived from a counter-based generator (one splitmix64
finalisation per draw), so that this numpy implementation, the host C++ one and
the HIP kernel in ``csrc/synth.hip`` produce identical bytes for any record
range without sharing state.

Shape sources (reference, data only): genome size ``testsuites/MTBC/_util.py:9``;
27 phylo SNPs ``testsuites/MTBC/phylo.py:132-162``; 58 resistance SNPs and 4
regions ``testsuites/MTBC/resistance.py:136-201``; 43 spoligotyping spacers
``testsuites/MTBC/spoligo.py:76-118``; 25 bp flanks ``kvarq/config.py:9``;
both strands ``kvarq/analyse.py:352-354``.
"""
import numpy as np

SEED = 0x4B56415251            # "KVARQ"
GENOME_SIZE = 4411532
FLANK = 25

_M64 = (1 << 64) - 1
GOLD = 0x9E3779B97F4A7C15
C_REC = 0xD1B54A32D192ED03     # stride between records
C_GEN = 0xA0761D6478BD642F     # genome stream offset
P_ERR = 0.005                  # substitution probability per base
P_BADQ = 0.01                  # probability of a sub-threshold quality
THR_ERR = int(P_ERR * (1 << 32))
THR_BADQ = int(P_BADQ * (1 << 24))
Q_GOOD, Q_BAD = ord('I'), ord('#')


def mix64(x):
    """splitmix64 finaliser; x is a python int or a numpy uint64 array"""
    if isinstance(x, np.ndarray):
        x = x.astype(np.uint64)
        with np.errstate(over='ignore'):
            x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
            x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
            return x ^ (x >> np.uint64(31))
    x &= _M64
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & _M64
    return x ^ (x >> 31)


# --- table shape (positions are 1-based genome coordinates) -----------------

PHYLO_SNPS = [
    (3920109, 'T'), (3597682, 'T'), (1590555, 'T'), (1834177, 'C'), (3304966, 'A'), (2711722, 'G'),
    (301341, 'A'), (4266647, 'G'), (157129, 'T'), (3326554, 'A'), (2154724, 'C'), (648856, 'T'),
    (1377185, 'G'), (801959, 'T'), (2859147, 'T'), (2427828, 'C'), (378404, 'A'), (4269522, 'A'),
    (14806, 'C'), (1663221, 'G'), (497126, 'A'), (3480645, 'G'), (1427476, 'T'), (3624593, 'T'),
    (2112832, 'C'), (3587446, 'A'), (1849051, 'T'),
]

RESISTANCE_SNPS = [
    (2155276, 'C', 'T'), (1673432, 'T', 'A'), (1673432, 'T', 'C'), (1673425, 'C', 'T'),
    (3877949, 'T', 'C'), (3877949, 'T', 'G'), (3877960, 'A', 'G'), (3877960, 'A', 'C'),
    (764669, 'C', 'G'), (764670, 'C', 'G'), (764817, 'T', 'C'), (764817, 'T', 'G'), (764819, 'T', 'G'),
    (764822, 'G', 'A'), (764822, 'G', 'C'), (764840, 'A', 'G'), (764841, 'T', 'C'), (764918, 'G', 'C'),
    (765461, 'A', 'C'), (765462, 'A', 'G'), (765463, 'C', 'G'), (7606, 'C', 'A'), (7677, 'G', 'A'),
    (7678, 'C', 'G'), (6767, 'G', 'A'), (6768, 'G', 'A'), (781687, 'A', 'G'), (781822, 'A', 'C'),
    (781822, 'A', 'T'), (781822, 'A', 'G'), (1472337, 'C', 'A'), (1472337, 'C', 'G'), (1472337, 'C', 'T'),
    (1472358, 'C', 'A'), (1472358, 'C', 'G'), (1472358, 'C', 'T'), (1472359, 'A', 'C'), (1472359, 'A', 'G'),
    (1472359, 'A', 'T'), (1472362, 'C', 'A'), (1472362, 'C', 'G'), (1472362, 'C', 'T'), (1472752, 'A', 'C'),
    (1472752, 'A', 'G'), (1472752, 'A', 'T'), (1473246, 'A', 'C'), (1473246, 'A', 'G'), (1473246, 'A', 'T'),
    (1473247, 'C', 'A'), (1473247, 'C', 'G'), (1473247, 'C', 'T'), (4247429, 'A', 'G'), (4247431, 'G', 'A'),
    (4247431, 'G', 'T'), (4247431, 'G', 'C'), (4247429, 'A', 'C'), (4247730, 'G', 'C'), (4248003, 'A', 'G'),
]

RESISTANCE_REGIONS = [(2155167, 2155169), (761082, 761162), (7521, 7583), (2288681, 2289241)]

SPOLIGO_SPACERS = [
    'TGATCCAGAGCCGGCGACCCTCTAT', 'CAAAAGCTGTCGCCCAAGCATGAGG', 'TAGAAGGCGATCACTGGAAGCACGG',
    'CTGATGATTGGTCGGCGTATGACGT', 'TAATCCCGCACAAGTGGTCAGAAAA', 'GAAATTGAAGCCGGAAATGACGACG',
    'GCAGCCCCGAGTACTCGCTCTCCTC', 'CGGCGAGGCTGGGGGCGGTTTCACG', 'GCTGTCAGCACATGGGATTCCGAGT',
    'GGAAGTCAACTAGAGCGGGTGTCGA', 'CCAGGTTGCCGCCGCCGTTGCTCAC', 'ATCTCCCCGGGCGGGCAGCAGATAT',
    'GGGAGAGGGAATGGCAATGATGGTC', 'CCGAGCCGACCATCCGCATCACACC', 'CGAAATTCACTGCGCGTTATTCAAG',
    'GATTTACGACGCTGACGGGAACTCG', 'CGGAGTCATCCGCGCGGGCCGGCGC', 'CATCTGCAGCTCGCCCGGGTCCATG',
    'ACCAGGATCAGCGCCAAGCCAGTTA', 'TGATCTTCTCTCCTGGCGAGGTCAA', 'TCGACGATTGGGACATCGACATCGA',
    'TTGTCTCAATCGTGCCGTCTGCGGT', 'CGAGCTGGACCGCATCAGCGATGCT', 'CGAGCACGTCTCACCCAGCAGGCGG',
    'TGACAGGGTGCGGTGGTCGCTGATC', 'GCGCCGGATGATGGTGGTGCTGAAG', 'ATCCGCGGGAAGAGATCACGAATCC',
    'GTTGTGATCGCTAAACGCCGGGGCA', 'TGGTCGTGTCGTGGAGCCTGTATTT', 'GGCTGGAAAAGGGCGCGGGGCAACC',
    'ACTTGATCGACGCGAACCTGTCTGA', 'TGAACACGCCGATACCTATTTGGTC', 'TCAAGTGCGGCACCGCCGTCATGTC',
    'TTCGACGGTGTGGGCGAGGTGACTT', 'GTTGGAAGCGTTTCGAGCGTACGGA', 'GCTGCGGATGTGGTGCTGGATTTCG',
    'AAGGGGGACTGTGGACGAGTTCGCG', 'GCGCACAACGCATCCGCCATCCACG', 'CCACGCCGATTTACTGGCCATCGTC',
    'GGACCTGTATGAGGCACAGATGGCG', 'TACCTGATAGAAGCCGGAAAGCTCC', 'GTCGCGCTCGTCCATGTCCCACCAT',
    'CTCCCGCACCCGGTGCGATTCTGCG',
]

# spacers are planted into the synthetic genome at the direct-repeat locus,
# separated by the 36 bp direct repeat, so that spoligo templates get hits too
DR_LOCUS = 3119185
DIRECT_REPEAT = 'GTTTCCGTCCCCTCTCGGGGTTTTGGGTCTGACGAC'

# the seven barcode SNP positions shown in docs/tutorial.rst:246-252; the other
# 55 of the 62 "MTBC-SNP-barcodes" templates are drawn from the generator
BARCODE_KNOWN = [615938, 4404247, 3021283, 3216553, 2622402, 1491275, 3479545]

_COMPLEMENT = bytes.maketrans(b'ACGTN', b'TGCAN')


def revcomp(seq):
    """kvarq/genes.py:257-262 (Sequence.reverse): complement, reversed"""
    return seq.translate(_COMPLEMENT)[::-1]


def genome(seed=SEED, size=GENOME_SIZE):
    """uint8 array of ASCII bases: i.i.d. ACGT, then the planted loci"""
    idx = np.arange(size, dtype=np.uint64)
    with np.errstate(over='ignore'):
        r = mix64(np.uint64((seed + C_GEN) & _M64) + idx * np.uint64(GOLD))
    g = np.frombuffer(b'ACGT', dtype=np.uint8)[(r >> np.uint64(62)).astype(np.int64)].copy()
    for pos, orig, _ in RESISTANCE_SNPS:
        if pos <= size:
            g[pos - 1] = ord(orig)
    at = DR_LOCUS - 1
    for sp in SPOLIGO_SPACERS:
        blk = (DIRECT_REPEAT + sp).encode()
        if at + len(blk) <= size:
            g[at:at + len(blk)] = np.frombuffer(blk, dtype=np.uint8)
        at += len(blk)
    return g


def _snp_template(g, pos, base):
    s = bytearray(g[pos - 1 - FLANK:pos + FLANK].tobytes())
    s[FLANK] = ord(base)
    return bytes(s)


def table(g, name='MTBC', scale=1, seed=SEED):
    """
    list of plus-strand template byte strings.

    ``MTBC``: 27 + 58 SNPs x 51 bp, 4 regions (53/131/113/611 bp), 43 spacers x
    25 bp = 132 templates, 6318 bases.  ``MTBC+barcodes``: plus 62 SNP
    templates = 194 templates, 9480 bases.  ``scale`` > 1 appends (scale-1) x
    132 further SNP templates at generated positions (LDS-pressure sweep).
    """
    plus = [_snp_template(g, p, b) for p, b in PHYLO_SNPS]
    plus += [_snp_template(g, p, b) for p, _, b in RESISTANCE_SNPS]
    plus += [g[a - 1 - FLANK:b + FLANK].tobytes() for a, b in RESISTANCE_REGIONS]
    plus += [s.encode() for s in SPOLIGO_SPACERS]
    extra = 0
    if name == 'MTBC+barcodes':
        extra = 62
    elif name != 'MTBC':
        raise ValueError(name)
    extra += (scale - 1) * 132
    k = 0
    known = list(BARCODE_KNOWN) if name == 'MTBC+barcodes' else []
    while extra > 0:
        if known:
            pos = known.pop(0)
        else:
            pos = 1000 + mix64(seed + 0x5EED + k * GOLD) % (len(g) - 2000)
            k += 1
        ref = chr(g[pos - 1])
        base = 'ACGT'[('ACGT'.index(ref) + 1 + (pos % 3)) % 4]
        plus.append(_snp_template(g, pos, base))
        extra -= 1
    return plus


def both_strands(plus):
    """kvarq/analyse.py:352-354: plus strands followed by their reverse complements"""
    return list(plus) + [revcomp(s) for s in plus]


# --- reads -------------------------------------------------------------------

HEADER_FMT = '@SYN.%09d 1:N:0\n'


def record_bytes(L):
    return 2 * L + 25


def reads(g, first, n, L=150, seed=SEED):
    """
    records first .. first+n-1 as one uint8 array of n * (2L+25) bytes.

    draw(r, k) = mix64(seed + r*C_REC + k*GOLD):
      k = 0      -> start = lo32 % (G-L+1); strand = bit 63
      k = 1 + j  -> base j: substitution iff lo32 < THR_ERR, the replacement is
                    the (1 + bits 32..33 % 3)-th next letter of ACGT; quality j
                    is '#' iff bits 40..63 < THR_BADQ, else 'I'
    """
    G = len(g)
    rb = record_bytes(L)
    out = np.empty((n, rb), dtype=np.uint8)
    rec = np.arange(first, first + n, dtype=np.uint64)
    with np.errstate(over='ignore'):
        base_state = np.uint64(seed & _M64) + rec * np.uint64(C_REC)
        d0 = mix64(base_state)
        start = ((d0 & np.uint64(0xFFFFFFFF)) % np.uint64(G - L + 1)).astype(np.int64)
        minus = (d0 >> np.uint64(63)).astype(bool)
        j = np.arange(L, dtype=np.uint64)
        dj = mix64(base_state[:, None] + (j[None, :] + np.uint64(1)) * np.uint64(GOLD))
    lo = (dj & np.uint64(0xFFFFFFFF)).astype(np.int64)
    sub = lo < THR_ERR
    shift = (((dj >> np.uint64(32)) & np.uint64(3)).astype(np.int64) % 3) + 1
    badq = ((dj >> np.uint64(40)).astype(np.int64)) < THR_BADQ

    code_of = np.full(256, 0, dtype=np.int64)
    for i, c in enumerate(b'ACGT'):
        code_of[c] = i
    letters = np.frombuffer(b'ACGT', dtype=np.uint8)
    jj = np.arange(L, dtype=np.int64)
    # forward: genome[start+j]; reverse: complement(genome[start+L-1-j])
    gi = np.where(minus[:, None], start[:, None] + (L - 1 - jj)[None, :], start[:, None] + jj[None, :])
    codes = code_of[g[gi]]
    codes = np.where(minus[:, None], 3 - codes, codes)      # A<->T, C<->G in ACGT order
    codes = np.where(sub, (codes + shift) % 4, codes)
    hdr = np.frombuffer(b''.join((HEADER_FMT % r).encode() for r in range(first, first + n)), dtype=np.uint8)
    out[:, :21] = hdr.reshape(n, 21)
    out[:, 21:21 + L] = letters[codes]
    out[:, 21 + L] = 10
    out[:, 22 + L] = ord('+')
    out[:, 23 + L] = 10
    out[:, 24 + L:24 + 2 * L] = np.where(badq, Q_BAD, Q_GOOD)
    out[:, 24 + 2 * L] = 10
    return out.reshape(-1)
