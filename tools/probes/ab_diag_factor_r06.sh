#!/bin/bash
# Round 6: the resident adjuster's diagonal-tile factor, four wavefronts with a barrier per pivot (the library) against one wavefront in panels of
# four pivots on the matrix pipe (tools/build_variant.sh diagmfma slam_ba.hip -DMQS_SLAM_BA_DIAG_MFMA=1 -> build/stamps6/), interleaved on one box.
for i in 1 2; do
for lib in multiple-quadrotor-slam_amd/libmqslam_hip.so build/stamps6/libmqslam_diagmfma.so; do
  MQS_LIB_PATH=$lib python tools/probes/icl_selection_stamps.py 200 3 0 2>/dev/null | python -c "
import sys, json
rows=[json.loads(l) for l in sys.stdin if l.startswith('{')]
tot=lambda k: sum(r['phase_us'].get(k,0) for r in rows)
print('$lib'.split('/')[-1], 'adjustments', len(rows), 'kernel_us', round(sum(r['kernel_us'] for r in rows)), 'cholesky', round(tot('cholesky')), 'backsolve', round(tot('backsolve')), 'trials', sum(r['trials'] for r in rows))"
  MQS_LIB_PATH=$lib python tools/probes/ba_groups_study.py 2>/dev/null | tail -1
done; done
