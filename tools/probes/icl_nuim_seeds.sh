for s in 0 1 2 3; do python tools/run_icl_nuim.py 80 --seed $s 2>/dev/null | tail -1; done
for s in 0 1 2 3; do python tools/run_icl_nuim.py 80 --ba --seed $s 2>/dev/null | tail -1; done
export MQS_ICL_FIXTURE=tests/golden/icl_nuim_traj3n/sequence_200.npz
for s in 1 2 3; do python tools/run_icl_nuim.py 200 --seed $s 2>/dev/null | tail -1; done
for s in 1 2 3; do python tools/run_icl_nuim.py 200 --ba --seed $s 2>/dev/null | tail -1; done
python tools/run_icl_nuim.py 200 --ba --reassociate --seed 0 2>/dev/null | tail -1
