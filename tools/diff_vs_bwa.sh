#!/bin/bash
# tools/diff_vs_bwa.sh -- the differential tester that pins this repository's engine and oracle to REAL bwa, for whoever has
# a bwa checkout (the reference's submodule is absent from /root/reference and there is no network here: SURVEY.md 0.1).
#
#   BWADIR=/path/to/bwa [GPU=1] tools/diff_vs_bwa.sh REF.fa PAIRS.txt
#
# REF.fa: a FASTA; PAIRS.txt: one "READ1 READ2" per line.  Steps:
#   1. make -C $BWADIR (libbwa.a + the bwa binary); `bwa index REF.fa`; this repository's ema_index_build on a copy of REF.fa;
#      byte comparison of the .bwt/.sa/.pac/.ann/.amb files of the two (index interchangeability)
#   2. tools/bwa_dump.c built against real libbwa -> real.txt        (mem_align1_core / mem_matesw / mem_reg2aln per read)
#   3. tools/oracle_dump.py (the CPU oracle)          -> oracle.txt   diff vs real.txt
#   4. with GPU=1: the same bwa_dump.c built against libema_bwaabi.so (the engine behind the nine symbols) -> engine.txt, diffed too
# Exit code 0 = everything identical; 1 = a difference (the first lines are shown); 2 = cannot run.
set -u
here=$(cd "$(dirname "$0")/.." && pwd)
if [ -z "${BWADIR:-}" ]; then
	echo "BWADIR not set: no bwa checkout to compare with (parity stays unpinned; see DESIGN.md).  Usage: BWADIR=/path/to/bwa $0 REF.fa PAIRS.txt"
	exit 2
fi
if [ $# -lt 2 ] || [ ! -f "$BWADIR/bwamem.h" ]; then echo "usage: BWADIR=/path/to/bwa $0 REF.fa PAIRS.txt  (BWADIR must hold bwamem.h)"; exit 2; fi
ref=$1; pairs=$2
work=$(mktemp -d /tmp/ema_diff_XXXXXX)
make -C "$BWADIR" > "$work/make.log" 2>&1 || { echo "building bwa failed, see $work/make.log"; exit 2; }
mkdir -p "$work/real" "$work/ours"
cp "$ref" "$work/real/ref.fa"; cp "$ref" "$work/ours/ref.fa"
"$BWADIR/bwa" index "$work/real/ref.fa" > "$work/index.log" 2>&1 || { echo "bwa index failed"; exit 2; }
python3 -c "import sys; sys.path.insert(0, '$here'); from ema_amd import build_index; build_index('$work/ours/ref.fa')" || exit 2
rc=0
for e in bwt sa pac ann amb; do
	if cmp -s "$work/real/ref.fa.$e" "$work/ours/ref.fa.$e"; then echo "index .$e: identical to bwa index"; else echo "index .$e: DIFFERS from bwa index"; rc=1; fi
done
cc -O2 -DREAL_BWA -I"$BWADIR" -o "$work/dump_real" "$here/tools/bwa_dump.c" "$BWADIR/libbwa.a" -lz -lm -lpthread || { echo "cannot build bwa_dump against libbwa"; exit 2; }
"$work/dump_real" "$work/real/ref.fa" "$pairs" > "$work/real.txt" || exit 2
python3 "$here/tools/oracle_dump.py" "$work/ours/ref.fa" "$pairs" > "$work/oracle.txt" || exit 2
if diff -q "$work/real.txt" "$work/oracle.txt" > /dev/null; then echo "oracle: identical to real bwa on $(grep -c '^P' "$work/real.txt") pairs"; else echo "oracle: DIFFERS from real bwa"; diff "$work/real.txt" "$work/oracle.txt" | head -20; rc=1; fi
if [ "${GPU:-0}" = 1 ]; then
	cc -O2 -I"$here/include" -o "$work/dump_engine" "$here/tools/bwa_dump.c" -L"$here/ema_amd" -lema_bwaabi -Wl,-rpath,"$here/ema_amd" || exit 2
	"$work/dump_engine" "$work/ours/ref.fa" "$pairs" > "$work/engine.txt" || exit 2
	if diff -q "$work/real.txt" "$work/engine.txt" > /dev/null; then echo "engine: identical to real bwa"; else echo "engine: DIFFERS from real bwa"; diff "$work/real.txt" "$work/engine.txt" | head -20; rc=1; fi
fi
echo "outputs in $work"
exit $rc
