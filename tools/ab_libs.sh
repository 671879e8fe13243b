#!/bin/bash
# Same-box A/B of library variants built from THIS tree (python -m hackathonopticalflow_amd.build --out libofarn_x.so -DFLAG):
# alternates tools/kbench.py over the given "label=ENV1=v1,ENV2=v2" legs, N rounds.  Prints the polyexp / flow_iter / total lines.
#   tools/ab_libs.sh 2 512 "240=" "192=OFARN_LIB=hackathonopticalflow_amd/libofarn_pe192.so" ...
n=$1; batch=$2; shift 2
root=$(cd "$(dirname "$0")/.." && pwd); cd "$root"
for i in $(seq 1 "$n"); do
  for leg in "$@"; do
    label=${leg%%=*}; envs=${leg#*=}
    echo "## round $i: $label   [$envs]"
    ( IFS=','; for kv in $envs; do [ -n "$kv" ] && export "$kv"; done; python tools/kbench.py --batch "$batch" --reps 3 2>&1 | grep -E "polyexp +L[012]|flow_iter +L0|^total" ) || exit 1
  done
done
