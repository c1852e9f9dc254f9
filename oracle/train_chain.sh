#!/bin/bash
# ORACLE support - build-container only.  Runs the reference trainer (oracle/train_fixture.py = /root/reference/main.py,
# unmodified) in STAGES, each stage starting from the previous stage's checkpoint (-init_nw_weight), because main.py only
# saves at the very end of a run (main.py:248) and a CPU-hours run should not be all-or-nothing.
#   oracle/train_chain.sh <workdir> <threads> <stages> <init.pt|default> <main.py flags...>
# Every stage's flags, log and checkpoint stay in <workdir>; <workdir>/latest.pt is the newest checkpoint.
set -u
WD=$1; THREADS=$2; STAGES=$3; INIT=$4; shift 4
mkdir -p "$WD"; cd "$WD" || exit 1
echo "flags: $*" > flags.txt
for s in $(seq 1 "$STAGES"); do
  if [ -f latest.pt ]; then INIT=$PWD/latest.pt; fi
  before=$(ls tmp/torch_model_*.pt 2>/dev/null | wc -l)
  OMP_NUM_THREADS=$THREADS MKL_NUM_THREADS=$THREADS python -u /root/repo/oracle/train_fixture.py "$@" -init_nw_weight "$INIT" --no-cuda > "stage_$s.log" 2>&1
  new=$(ls -t tmp/torch_model_*.pt 2>/dev/null | head -1)
  after=$(ls tmp/torch_model_*.pt 2>/dev/null | wc -l)
  if [ "$after" -le "$before" ]; then echo "stage $s produced no checkpoint" >> chain.log; exit 1; fi
  cp "$new" "stage_$s.pt"; cp "$new" latest.pt
  echo "stage $s done: $new  $(grep -a 'with ber' "stage_$s.log" | tail -1)" >> chain.log
done
