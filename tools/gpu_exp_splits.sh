# sensitivity of the small-batch training step to the split-K heuristic of the exact-fp32 GEMM (bring-up build)
export MONOLOCO_HIP_LIB=$PWD/monoloco_amd/lib/libmonoloco_hip_bringup.so
for cfg in "512 128" "256 128" "128 128" "1024 64" "1024 32" "2048 32" "2048 16"; do
set -- $cfg
echo -n "maxwg $1 mink $2: "; ML_GEMM_MAXWG=$1 ML_GEMM_MINK=$2 timeout 200 python tools/exp_train_small.py 2>&1 | tail -1
done
