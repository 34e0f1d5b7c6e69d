cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_configs.py -m gpu -x -q -k "packed_polyline or side_stream" 2>&1 | tail -3
for v in 3 4 3 4 3 4; do echo "PACK=$v"; TB_ENCODE_PACK=$v python tests/probes/gpu_encode_time.py 2>&1 | tail -2; done
