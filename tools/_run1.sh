cd $GRAFT_REPO_ROOT
TB_HIP_LIB=$PWD/trafficbots_amd/lib/libtrafficbots_hip_prof.so python tools/gpu_stage_profile.py 2>&1 | tail -60
