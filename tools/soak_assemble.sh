#!/bin/bash
# Soak of the unlocked assembly rounds (pba_cons_round) on the GPU box: six configurations the tests do not hold, the CPU
# oracle beside every round.  Every line printed must say "checked_same": true.
# usage (through gpurun): bash tools/soak_assemble.sh [first seed]
set -e
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
s0=${1:-21}
for s in $(seq $s0 $((s0 + 5))); do
  timeout -k 10 200 python tools/bench_assemble.py --genome $((30000 + s*1000)) --reads $((150 + s*5)) --read-len $((1500 + (s%3)*900)) \
      --err 0.$((10 + s%8)) --start-len $((3000 + (s%4)*1500)) --rounds 6 --check-rounds 6 --trials $((16 + (s%3)*8)) --seed $s | tail -1
done
