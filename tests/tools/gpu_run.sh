# scratch script of the last `gpurun` call (see README.md in this directory)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I. -o /tmp/engine_probe tests/tools/engine_probe.hip 2>&1 | grep -E "error" 
for cfg in "layer 64 7 1" "layer 64 5 1" "layer 64 3 1" "ffn 64 7 1" "wo 64 7 1"; do
  echo "=== $cfg" >> gpurun_out/r02_engine4.txt
  timeout 90 /tmp/engine_probe $cfg >> gpurun_out/r02_engine4.txt 2>&1; echo "rc=$?" >> gpurun_out/r02_engine4.txt
done
cat gpurun_out/r02_engine4.txt
