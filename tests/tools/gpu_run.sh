cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
for x in 0 2; do
rm -rf /tmp/pf$x
GGML_HIP_MMQ_XCDN=$x timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pf$x -o bench -- python bench.py --mode prefill --steps 1 --warmup 1 > gpurun_out/pf$x.log 2>&1
echo "XCDN=$x"; python - <<PY
import sqlite3,glob
con=sqlite3.connect(glob.glob('/tmp/pf$x/*.db')[0])
q="select e.name, d.grid_size_x, d.grid_size_y, count(*), avg(e.counter_value), avg(e.duration)/1e3 from pmc_events e join counters_collection d on d.dispatch_id=e.dispatch_id where e.name like '%k_mmq_dma%' and e.counter_name='FETCH_SIZE' and d.counter_name='FETCH_SIZE' group by 1,2,3"
for r in con.execute(q): print(r[0][:24], (r[1],r[2]), r[3], "FETCH KiB %.0f  -> x2 = %.1f MB"%(r[4], 2*r[4]*1024/1e6), "us %.1f"%r[5])
PY
done
