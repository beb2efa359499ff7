export MQS_ICL_FIXTURE=tests/golden/icl_nuim_traj3n/sequence_200.npz
for s in 0 1 2 3 9 10; do python tools/run_icl_nuim.py 200 --screen 1.0 --seed $s 2>/dev/null | tail -1; done
for s in 0 1 2 3; do python tools/run_icl_nuim.py 200 --ba --screen 1.0 --seed $s 2>/dev/null | tail -1; done
