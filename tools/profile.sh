#!/bin/bash
# Profile one bench.py configuration on the GPU box (run under gpurun from the repo root):
#   bash tools/profile.sh <tag> <counter-passes|none> -- <bench.py args...>
# Writes gpurun_out/<tag>/{stats,fetch,write,pmcN}/ ; summarise with tools/prof_summary.py and copy to profiles/.
# Counter passes run separately from the kernel trace and from each other (--pmc with --kernel-trace only), as
# MI355X_MICROARCH.md prescribes; "python3 bench.py" is what follows "--" (no env/bash hop under rocprofv3).
set -e
tag=$1; shift
passes=$1; shift
[ "$1" = "--" ] && shift
out=gpurun_out/$tag
mkdir -p $out
cd /tmp 2>/dev/null || true
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python3 bench.py "$@" --no-cpu-baseline > $out/bench.json 2> $out/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py "$@" --no-cpu-baseline > $out/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/fetch -- python3 bench.py "$@" --no-cpu-baseline > $out/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/write -- python3 bench.py "$@" --no-cpu-baseline > $out/write.log 2>&1
if [ "$passes" != "none" ]; then
  i=0
  IFS=';' read -ra P <<< "$passes"
  for p in "${P[@]}"; do
    i=$((i+1))
    rocprofv3 --pmc $p --kernel-trace --output-format csv -d $out/pmc$i -- python3 bench.py "$@" --no-cpu-baseline > $out/pmc$i.log 2>&1
  done
fi
echo "profile $tag done"
