#!/bin/bash
# disassemble one kernel of an object file: bash tools/kdis.sh <obj.o> <mangled-name substring> [out.s]
set -e
T=$(mktemp -d); cp "$1" $T/o.o; (cd $T && /opt/rocm/lib/llvm/bin/llvm-objdump --offloading o.o > /dev/null)
CO=$(ls $T/o.o.*gfx950* | head -1)
/opt/rocm/lib/llvm/bin/llvm-objdump -d --no-show-raw-insn $CO | awk -v pat="$2" '/^[0-9a-f]+ <.*>:$/{f = index($0, pat) > 0} f' > ${3:-/tmp/kdis.s}
wc -l ${3:-/tmp/kdis.s}
