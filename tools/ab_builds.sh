#!/bin/bash
# Same-box A/B of two builds of the library: alternates tools/kbench.py of the tree in $1 (an older checkout with
# its own package + built libofarn.so, e.g. ab/r02 = round 2's d963013) and of this tree, N times, one wave of 512 pairs at 1080p.
#   tools/ab_builds.sh ab/r02 3 > gpurun_out/r04_polyexp_ab.txt
other=${1:-ab/r02}; n=${2:-3}; batch=${3:-512}
root=$(cd "$(dirname "$0")/.." && pwd)
echo "# $(date -u +%FT%TZ) $(/opt/rocm/bin/rocminfo 2>/dev/null | grep -m1 'Marketing Name.*MI' | sed 's/ *Marketing Name: *//')"
echo "# alternating: $other (A) vs HEAD (B); python tools/kbench.py --batch $batch --reps 3"
for i in $(seq 1 "$n"); do
  echo "## alternation $i: A = $other"
  (cd "$root/$other" && python tools/kbench.py --batch "$batch" --reps 3 2>&1) || exit 1
  echo "## alternation $i: B = HEAD"
  (cd "$root" && python tools/kbench.py --batch "$batch" --reps 3 2>&1) || exit 1
done
