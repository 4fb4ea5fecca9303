#!/bin/bash
# Round 6: GPU tier (or a subset), then bench legs given as "NAME|ENV|ARGS" triples.
# usage: tools/r06_gpu.sh OUTNAME "PYTEST ARGS (or 'none')" "name|ENV=..|bench args" ...
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r06}
mkdir -p $OUT
shift
PYT="${1:-tests}"
shift
if [ "$PYT" != "none" ]; then
  ( time timeout 1500 python -m pytest $PYT -m gpu -x -q -p no:cacheprovider ) > $OUT/pytest.log 2>&1
  echo "pytest rc $?" >> $OUT/pytest.log
  tail -8 $OUT/pytest.log
fi
for leg in "$@"; do
  name="${leg%%|*}"; rest="${leg#*|}"; envs="${rest%%|*}"; args="${rest#*|}"
  ( time env $envs timeout 900 python bench.py $args ) > $OUT/bench_$name.json 2> $OUT/bench_$name.err
  echo "== $name rc $?"; tail -c 300 $OUT/bench_$name.err; head -c 1200 $OUT/bench_$name.json; echo
done
