/* rt_cpus.h — how many CPUs this process can actually keep busy.
 *
 * std::thread::hardware_concurrency() is the machine's: on the MI355X boxes of this project 256 hardware threads (2 x EPYC 9575F) — of which the container may use SIXTEEN
 * (cgroup v2 cpu.max "1600000 100000": 16 CPU-seconds per second).  A phase that starts 256 workers there burns its period's quota in 6 ms and sleeps for the other 94:
 * the CPU oracle measured 16-19 CPUs busy whatever the thread count, a compute loop in 256 processes 1.6 M iterations per process-second against 15 M in 16
 * (scripts/r06_quota_probe.sh, scripts/r06_oracle_scaling.py, profiles/r06_cpu_baseline.txt).  Every default thread count of the host code, the builder and the oracle
 * is this number since the end of round 6: the smallest of the online CPUs, the affinity mask and the cgroup CPU quota (v2 cpu.max, v1 cfs_quota_us / cfs_period_us),
 * rounded up; RESTIR_CPUS overrides.  Header-only, C and C++ (include/ is the contract both the product and the oracle build against; neither links the other). */
#ifndef RT_CPUS_H
#define RT_CPUS_H
#ifndef _GNU_SOURCE
#define _GNU_SOURCE 1
#endif
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

static inline int rt_cpu_quota(void)   /* CPUs the cgroup quota allows (rounded up), 0 = no quota / unknown */
{
  FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r");
  if(f) {
    char q[64]; long long period = 0; int c = 0;
    if(fscanf(f, "%63s %lld", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0) { const long long quota = atoll(q); if(quota > 0) c = (int)((quota + period - 1) / period); }
    fclose(f);
    return c;
  }
  long long quota = -1, period = 0;
  f = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r");
  if(f) { if(fscanf(f, "%lld", &quota) != 1) quota = -1; fclose(f); }
  f = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r");
  if(f) { if(fscanf(f, "%lld", &period) != 1) period = 0; fclose(f); }
  return (quota > 0 && period > 0) ? (int)((quota + period - 1) / period) : 0;
}

static inline int rt_cpu_budget(void)
{
  const char* e = getenv("RESTIR_CPUS");
  if(e && atoi(e) > 0) return atoi(e);
  long n = sysconf(_SC_NPROCESSORS_ONLN);
  if(n < 1) n = 1;
#ifdef CPU_COUNT   /* (sched.h under _GNU_SOURCE — every C++ translation unit of g++ / hipcc; a strict C unit that included system headers first goes without the mask) */
  {
    cpu_set_t set;
    CPU_ZERO(&set);
    if(sched_getaffinity(0, sizeof(set), &set) == 0) { const int c = CPU_COUNT(&set); if(c > 0 && c < n) n = c; }
  }
#endif
  const int q = rt_cpu_quota();
  if(q > 0 && q < n) n = q;
  return (int)n;
}
#endif
