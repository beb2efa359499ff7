export MQS_ICL_FIXTURE=tests/golden/icl_nuim_traj3n/sequence_200.npz
for sig in 0.02 0.005; do for w in 6 10; do for s in 0 1; do
  MQS_BA_WINDOW_POINT_SIGMA=$sig python tools/run_icl_nuim.py 200 --ba --window $w --seed $s 2>/dev/null | tail -1
done; done; done
