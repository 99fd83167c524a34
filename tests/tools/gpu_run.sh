# scratch script of the last `gpurun` call (see README.md in this directory)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out/r02_env_probe.txt
: > $O
run() { label="$1"; shift; env "$@" timeout 200 python tests/tools/env_probe.py "$label" 2>/dev/null | tail -1 >> $O; }
run default A=1
run HIP_FORCE_DEV_KERNARG=0 HIP_FORCE_DEV_KERNARG=0
run HIP_FORCE_DEV_KERNARG=1 HIP_FORCE_DEV_KERNARG=1
run GRAPH_PACKET_CAPTURE=0 DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run GRAPH_PACKET_CAPTURE=1 DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run AMD_OPT_FLUSH=0 AMD_OPT_FLUSH=0
run AMD_OPT_FLUSH=1 AMD_OPT_FLUSH=1
run ROC_SYSTEM_SCOPE_SIGNAL=0 ROC_SYSTEM_SCOPE_SIGNAL=0
run GPU_FLUSH_ON_EXECUTION=0 GPU_FLUSH_ON_EXECUTION=0
run GGML_HIP_GRAPH=0 GGML_HIP_GRAPH=0
cat $O
