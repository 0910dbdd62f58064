#!/bin/bash
# End-to-end wall time of the drop-in CLI (parse + upload + kernels + FASTQ output): usage cli_e2e.sh READS [REPEATS]
set -e
N=${1:-1000000}; REP=${2:-1}
D=$(mktemp -d)
python - "$N" "$D" <<'PY'
import sys
sys.path.insert(0, ".")
from rattle_amd import synth
n = int(sys.argv[1]); d = sys.argv[2]
cat, qcat, off, tid, _ = synth.reads_packed(n, max(5, n // 200), 1, True, seed=20260929, exon=(50, 210))
with open(d + "/reads.fq", "wb") as f:
    for i in range(n):
        f.write(b"@r%d\n" % i); f.write(cat[int(off[i]):int(off[i + 1])].tobytes()); f.write(b"\n+\n"); f.write(qcat[int(off[i]):int(off[i + 1])].tobytes()); f.write(b"\n")
PY
ls -la $D/reads.fq
for rep in $(seq 1 $REP); do
s=$(date +%s%N); ./rattle_amd/csrc/rattle cluster -i $D/reads.fq -o $D -t 32 2>$D/err0.txt; (grep -E "rattle cli\]" $D/err0.txt || true); e=$(date +%s%N); echo "rattle cluster: $(( (e - s) / 1000000 )) ms"
s=$(date +%s%N); ./rattle_amd/csrc/rattle correct -i $D/reads.fq -c $D/clusters.out -o $D -t 32 2>$D/err.txt; grep -E "rattle( cli)?\]" $D/err.txt | grep -vE "poa class|stage: " | head -40 || true; e=$(date +%s%N); echo "rattle correct: $(( (e - s) / 1000000 )) ms"
done
ls -la $D | tail -5
rm -rf $D
