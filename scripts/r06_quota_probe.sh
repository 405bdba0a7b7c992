cat /sys/fs/cgroup/cpu.max; grep -E "nr_throttled|throttled_usec|nr_periods" /sys/fs/cgroup/cpu.stat; nproc; cat /sys/fs/cgroup/cpuset.cpus.effective 2>/dev/null; cat /proc/loadavg; lscpu | grep -E "Model name|Socket|Core|Thread|NUMA node\(s\)"
python3 - <<'PY'
import os,time,multiprocessing as mp
def burn(_):
    t=time.perf_counter(); x=0
    while time.perf_counter()-t<1.0: x+=1
    return x
if __name__=="__main__":
    for n in (1,8,16,32,64,128,256):
        with mp.Pool(n) as p:
            r=p.map(burn, range(n))
            print(n, "procs: M iterations per proc-second", round(sum(r)/1e6/n,2), " sum", round(sum(r)/1e6,1), flush=True)
PY
grep -E "nr_throttled|throttled_usec|nr_periods" /sys/fs/cgroup/cpu.stat
