#!/bin/bash
# A/B of the fused tail's variants on one device, interleaved.  Arguments: N, then variants as ENV=VALUE settings, e.g.
#   bash tools/ab_tail.sh 1000000 MQS_BA_TAIL_SOLVERS=0 MQS_BA_TAIL_SOLVERS=1 MQS_BA_TAIL_FORM=2
N=${1:-1000000}; shift
for i in 1 2; do for v in "$@"; do env $v python tools/ab_lin.py $N 4 3 300 2>/dev/null | tail -1 | sed "s/^{/{\"variant\": \"$v\", /"; done; done
