#!/bin/bash
# ASan/UBSan run of the host BAM reader (host/bam_reader.cpp) over mutated inputs: a valid BGZF container
# (CRCs recomputed) around a damaged record stream, next to a damaged .bai.  CPU only.
#   tools/fuzz_bam_reader.sh [n_files=300]
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
D=$(mktemp -d /tmp/gd_fuzz_XXXX)
N=${1:-300}
g++ -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -std=c++17 -I $R/goleft_amd/csrc/host \
    -o $D/drv $R/tools/fuzz_bam_reader.cpp $R/goleft_amd/csrc/host/bam_reader.cpp -lz -pthread
python3 - "$R" "$D" "$N" <<'PY'
import sys
import numpy as np
root, d, n = sys.argv[1], sys.argv[2], int(sys.argv[3])
sys.path.insert(0, root)
from oracle import bamio
from tests import helpers as H
rng = np.random.default_rng(1)
contigs = [("f1", 30_000), ("f2", 9_000)]
reads = {0: H.random_reads(rng, 30_000, 300, max_len=80), 1: H.long_cigar_reads(rng, 9_000, [70_000, 3], max_step=2)}
p = d + "/v.bam"
bamio.write_bam(p, contigs, reads, unplaced=1, index=True)
raw0 = bamio.bgzf_decompress(open(p, "rb").read())
bai0 = open(p + ".bai", "rb").read()
hdr = 12 + int.from_bytes(raw0[4:8], "little") + sum(8 + len(nm) + 1 for nm, _ in contigs)
for i in range(n):
    raw, bai = bytearray(raw0), bytearray(bai0)
    for _ in range(int(rng.integers(1, 12))):
        raw[int(rng.integers(hdr if i % 3 else 0, len(raw)))] = int(rng.integers(0, 256))
        bai[int(rng.integers(4, len(bai)))] = int(rng.integers(0, 256))
    q = "%s/c%04d.bam" % (d, i)
    open(q, "wb").write(bamio.bgzf_compress(bytes(raw)))
    open(q + ".bai", "wb").write(bytes(bai))
PY
bad=0
for f in $D/c*.bam; do
  if ! timeout 60 $D/drv $f > $D/out.txt 2>&1; then echo "FAIL $f"; tail -5 $D/out.txt; bad=1; fi
done
[ $bad = 0 ] && echo "fuzz ok: $N mutated inputs, no sanitizer report, no hang" && rm -rf $D
exit $bad
