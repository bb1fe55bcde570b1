set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
LATTE_AMD_LIB=latte_amd/lib/liblatte_amd_dbg.so timeout 900 python tools/ladder_probe.py 1.5 both 0,128,256 2>&1 | grep -v amdgpu.ids > gpurun_out/r6_probe_ladder_lookahead.log
LATTE_AMD_LIB=latte_amd/lib/liblatte_amd_dbg.so timeout 900 python tools/ladder_probe.py 1.5 inmodel 0,128,256 2>&1 | grep -v amdgpu.ids >> gpurun_out/r6_probe_ladder_lookahead.log
cat gpurun_out/r6_probe_ladder_lookahead.log
