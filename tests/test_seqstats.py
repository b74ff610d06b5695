"""`--stats` columns (depth/depth.go:191-200, :244-252): the device base-class counter
gd_seq_stats against the oracle's byte-by-byte restatement, and the CLI with --stats
against golden rows + the oracle's formatted columns.  The semantics live in an external
module (faidx) that no reference test pins: PARITY UNPINNED, the oracle restatement is the
contract (oracle/pyoracle.py::seq_stats)."""
import numpy as np
import pytest

from oracle import bamio, pyoracle as po
from tests import helpers as H


def random_seq(rng, n, odd=(0, 10, 0x7b, 0x60, 0xe3, 0xc7, 0xff, ord("R"), ord("y"))):
    """Bases with soft-masked (lower-case) stretches, N runs and a few odd bytes."""
    s = rng.choice(np.frombuffer(b"ACGT", np.uint8), size=n)
    i = 0
    while i < n:
        run = int(rng.integers(1, 400))
        kind = rng.random()
        if kind < 0.25:
            s[i:i + run] |= 0x20                      # lower case
        elif kind < 0.30:
            s[i:i + run] = ord("N") if rng.random() < 0.5 else ord("n")
        i += run
    for b in odd:
        if n:
            s[int(rng.integers(0, n))] = b
    return s.tobytes()


def test_oracle_known_answers():
    # hand-checked: ACGTacgtNNCG -> G/C at 1,2,5,6,10,11; CpG at (1,2),(5,6),(10,11); lower 4..7
    seq = b"ACGTacgtNNCG"
    assert po.seq_stats(seq, 0, 12) == (6, 3, 4)
    assert po.seq_stats(seq, 0, 2) == (1, 1, 0)       # the G that completes the CpG lies past the window
    assert po.seq_stats(seq, 11, 12) == (1, 0, 0)
    assert po.seq_stats(seq, 10, 11) == (1, 1, 0)
    assert po.seq_stats(seq, 11, 40) == (1, 0, 0)     # clipped to the contig; no look-ahead past its end
    assert po.seq_stats(seq, 12, 40) == (0, 0, 0)
    assert po.seq_stats(b"cG" * 3, 0, 6) == (6, 3, 3)  # case-insensitive pairing
    assert po.stats_columns(seq, 0, 12, po.STATS_WINDOW) == "\t0.5\t0.5\t0.333"
    assert po.stats_columns(seq, 20, 30, po.STATS_WINDOW) == "\t0\t0\t0"
    assert po.stats_columns(b"ACGT" * 100, 0, 300, po.STATS_WINDOW) == "\t0.5\t0.5\t0"


def test_both_contracts_on_windows_with_n():
    """The forks of the unpinned faidx.Stats contract, each on a window where it matters
    (include/goleft_depth_host.h GDH_STATS_*): N / soft-masked n in the window, a CpG-dense window, a CpG
    across a FASTA line break."""
    seq = b"ACGTacgtNNCG"                                    # 10 A/C/G/T, 2 N; 6 G/C; 3 CpG; 4 lower case
    assert po.seq_counts(seq, 0, 12) == (6, 3, 4, 10, 4)
    assert po.stats_columns(seq, 0, 12, po.STATS_FAIDX) == "\t0.6\t0.6\t0.4"
    assert po.stats_columns(seq, 0, 12, po.STATS_WINDOW) == "\t0.5\t0.5\t0.333"
    soft = b"ACGTnnnnacgtNN"                                 # soft-masked n: lower case, not a base
    assert po.seq_counts(soft, 0, 14) == (4, 2, 8, 8, 4)
    assert po.stats_columns(soft, 0, 14, po.STATS_DENOM_ACGT | po.STATS_MASKED_ACGT) == "\t0.5\t0.5\t0.5"
    assert po.stats_columns(soft, 0, 14, po.STATS_DENOM_ACGT) == "\t0.5\t0.5\t1"        # any lower case / 8 bases
    assert po.stats_columns(soft, 0, 14, po.STATS_MASKED_ACGT) == "\t0.286\t0.286\t0.286"
    assert po.stats_columns(soft, 0, 14, po.STATS_WINDOW) == "\t0.286\t0.286\t0.571"
    assert po.stats_columns(b"NNNNnnnn", 0, 8, po.STATS_FAIDX) == "\t0\t0\t0"             # no base at all
    assert po.stats_columns(b"NNNNnnnn", 0, 8, po.STATS_WINDOW) == "\t0\t0\t0.5"
    dense = b"CGCGCGNNNN"                                    # 2 cpg / 6 bases = 1 (the window's last C pairs with
    assert po.seq_counts(dense, 0, 5) == (5, 3, 0, 5, 0)     # the G past it): 6 / 5 clamps to 1
    assert po.stats_columns(dense, 0, 5, po.STATS_FAIDX) == "\t1\t1\t0"
    assert po.stats_columns(dense, 0, 5, po.STATS_FAIDX & ~po.STATS_CPG_CLAMP) == "\t1\t1.2\t0"
    wrapped = b"AAACGAAC" + b"GAAAAAAA"                       # 8 bases per line: the second CpG straddles the break
    assert po.seq_counts(wrapped, 0, 16) == (4, 2, 0, 16, 0)
    assert po.seq_counts(wrapped, 0, 16, line_bases=8) == (4, 1, 0, 16, 0)
    # ... and a C that is the last base of the WINDOW starts none either under that reading (its G is not among the
    # window's bytes); without it the base after the window counts
    assert po.seq_counts(wrapped, 0, 4, line_bases=8)[1] == 0 and po.seq_counts(wrapped, 0, 4)[1] == 1
    assert po.seq_counts(wrapped, 0, 5, line_bases=8)[1] == 1
    assert po.stats_columns(wrapped, 0, 16, po.STATS_FAIDX, line_bases=8) == "\t0.25\t0.125\t0"
    assert po.stats_columns(wrapped, 0, 16, po.STATS_FAIDX & ~po.STATS_CPG_RAW_LINES, line_bases=8) == "\t0.25\t0.25\t0"


def test_host_formatter_equals_oracle_under_every_contract():
    """gdh_format_stats (the product's formatter, fed integer counts) against oracle/pyoracle.py::stats_columns
    for all 16 contracts."""
    from goleft_amd import _hostlib as hl
    rng = np.random.default_rng(5)
    for n in (1, 7, 60, 333):
        seq = random_seq(rng, n)
        for lb in (0, 5, 60):
            for _ in range(12):
                s = int(rng.integers(0, n))
                e = s + int(rng.integers(0, n + 3))
                for contract in range(16):
                    raw = lb if contract & po.STATS_CPG_RAW_LINES else 0
                    got = hl.format_stats(contract, 1, s, e, *po.seq_counts(seq, s, e, raw))
                    assert got == po.stats_columns(seq, s, e, contract, lb), (n, lb, s, e, contract)
    assert hl.format_stats(po.STATS_FAIDX, 0, 0, 10, 5, 1, 2, 9, 2) == "\t0\t0\t0"    # chromosome not in the FASTA


@pytest.mark.gpu
@pytest.mark.parametrize("n", [0, 1, 3, 4, 5, 63, 64, 257, 1000, 4099, 200003])
def test_device_counts_equal_oracle(n):
    from goleft_amd.engine import DepthEngine
    rng = np.random.default_rng(100 + n)
    seq = random_seq(rng, n)
    wins = [(0, n), (0, 0), (n, n + 10), (n - 1, n + 5), (-5, 3), (1, 2), (2, 1), (3, 7), (0, 1)]
    for W in (1, 7, 250, 1000):                       # anchored windows like the callback's
        if n / W < 3000:
            wins += [(k, min(k + W, n)) for k in range(0, n, W)]
    for _ in range(300):                              # arbitrary alignment and length
        s = int(rng.integers(-3, n + 3))
        wins.append((s, s + int(rng.integers(0, 3000))))
    wins = [(max(s, 0), e) for s, e in wins]
    st = np.array([w[0] for w in wins], np.int64)
    en = np.array([w[1] for w in wins], np.int64)
    with DepthEngine(0) as eng:
        eng.seq_load(seq)
        gc, cpg, low = eng.seq_stats(st, en)
        ex = {lb: eng.seq_stats_ex(st, en, lb) for lb in (0, 1, 3, 4, 60, 61)}
        # a second contig replaces the first
        eng.seq_load(b"CGcg")
        g2 = eng.seq_stats(np.array([0], np.int64), np.array([4], np.int64))
    sel = range(len(wins)) if n <= 5000 else rng.choice(len(wins), 400, replace=False)
    for k in sel:
        assert (int(gc[k]), int(cpg[k]), int(low[k])) == po.seq_stats(seq, int(st[k]), int(en[k])), (n, wins[k])
        for lb, arrs in ex.items():
            assert tuple(int(a[k]) for a in arrs) == po.seq_counts(seq, int(st[k]), int(en[k]), lb), (n, wins[k], lb)
    assert [int(x[0]) for x in g2] == [4, 2, 2]


@pytest.mark.gpu
def test_seq_stats_needs_a_sequence():
    from goleft_amd.engine import DepthEngine
    with DepthEngine(0) as eng:
        with pytest.raises(RuntimeError):
            eng.seq_stats(np.array([0], np.int64), np.array([4], np.int64))


def write_fasta(path, contigs, seqs, width=60):
    fai = []
    with open(path, "wb") as f:
        for (name, length), seq in zip(contigs, seqs):
            assert len(seq) == length
            f.write(b">" + name.encode() + b" test\n")
            off = f.tell()
            for i in range(0, length, width):
                f.write(seq[i:i + width] + b"\n")
            fai.append("%s\t%d\t%d\t%d\t%d\n" % (name, length, off, width, width + 1))
    open(path + ".fai", "w").write("".join(fai))


@pytest.mark.gpu
@pytest.mark.parametrize("contract", [po.STATS_FAIDX, po.STATS_WINDOW, po.STATS_DENOM_ACGT | po.STATS_CPG_CLAMP])
@pytest.mark.parametrize("mode,W", [("wg", 1000), ("wg", 71), ("bed", 55)])
def test_cli_stats_columns(tmp_path, mode, W, contract):
    # depth/functional-test.sh passes --stats everywhere (:45,:56,:73 ...) without asserting the values
    from goleft_amd import depth
    contigs, reads, _ = H.load_golden_bam("t")
    bamio.write_bam(str(tmp_path / "t.bam"), contigs, reads, unplaced=3)
    rng = np.random.default_rng(7)
    # (line breaks cannot be bases of a FASTA record)
    seqs = [random_seq(rng, ln, odd=(0x7b, 0x60, 0xe3, ord("R"), ord("y"), ord("*"))) for _, ln in contigs]
    write_fasta(str(tmp_path / "t.fa"), contigs, seqs)
    beds = H.golden_beds()["t"]
    prefix = str(tmp_path / "o")
    args = ["-Q", "1", "--ordered", "--windowsize", str(W), "--stats", "--prefix", prefix,
            "--reference", str(tmp_path / "t.fa")]
    if mode == "bed":
        (tmp_path / "windows.bed").write_text("".join("%s\t%d\t%d\n" % tuple(r) for r in beds["regions"]))
        args += ["--bed", str(tmp_path / "windows.bed")]
    from goleft_amd import _hostlib as hl
    old = hl.get_stats_contract()
    assert old == po.STATS_FAIDX                           # the default (GOLEFT_STATS_CONTRACT unset)
    hl.set_stats_contract(contract)
    try:
        assert depth.Main(args + [str(tmp_path / "t.bam")]) == 0
    finally:
        hl.set_stats_contract(old)
    by_name = {c[0]: s for c, s in zip(contigs, seqs)}
    want = []
    for line in beds["%s_w%d" % (mode, W)]["depth"].splitlines():
        chrom, s, e, _ = line.split("\t")
        want.append(line + po.stats_columns(by_name[chrom], int(s), int(e), contract, 60) + "\n")
    got = open(prefix + ".depth.bed").read()
    assert got == "".join(want)
    assert open(prefix + ".callable.bed").read() == beds["%s_w%d" % (mode, W)]["callable"]
