export MONOLOCO_HIP_LIB=$GRAFT_REPO_ROOT/monoloco_amd/lib/libmonoloco_hip_bringup.so
for off in 0 10 22 45; do echo "cap 64 offset $off"; STREAM_OFFSET_US=$off ML_GRID_CAP=64 python tools/exp_streams.py 4 2>&1 | grep "streams 4\|calib"; done
echo "cap 128 offset 45"; STREAM_OFFSET_US=45 ML_GRID_CAP=128 python tools/exp_streams.py 4 2>&1 | grep "streams 2"
