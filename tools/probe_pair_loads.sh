R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pair; mkdir -p $O; rm -f $O/ab.txt; cd $R
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -q -m gpu --timeout 600 2>&1 | tail -2
run() { label=$1; shift
  env "$@" python bench.py $W 2> $O/err_$label.log | python -c "
import json,sys
l=json.loads(sys.stdin.readlines()[-1]); r=l.get('roofline') or {}
hk={k['kernel']:k['us'] for k in r.get('all_hash_kernels',[])}
print('$label', l['ms_per_step'], l['config']['ms_per_step_min'], l['config']['ms_per_step_max'], hk.get('hash_encode_fwd[prop_s128]'), hk.get('hash_encode_fwd[prop_s64]'))" | tee -a $O/ab.txt
}
C="--no-cpu-baseline --secondary= --full-model= --trained-steps 0"
W="$C --workload mixed16384_neuradar"
run pair A=1
run before NR_LIB_PATH=$R/neuradar_amd/csrc/libneuradar_hip_before.so
run pair2 A=1
run before2 NR_LIB_PATH=$R/neuradar_amd/csrc/libneuradar_hip_before.so
W="$C --workload cam4096_l16f2_w64"
run cam_pair A=1
run cam_before NR_LIB_PATH=$R/neuradar_amd/csrc/libneuradar_hip_before.so
