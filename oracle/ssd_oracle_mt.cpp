/*
 * ssd_oracle_mt.cpp — the CPU oracle on all host cores: the native runner behind bench.py's `cpu_baseline_all_cores`
 * ("all host cores, one frame per thread - the fair comparison for frame-sharded multi-GPU").
 *
 * TEST / BENCH INFRASTRUCTURE, like the rest of oracle/: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * call it; the product (lib/libssd_hip.so) neither links nor loads it.
 *
 * One std::thread per given CPU, pinned to it.  Each thread first COPIES the frame it will work on into memory it allocates and
 * touches itself (first touch: on its own NUMA node), then waits at a start line; from there every thread runs the single-threaded
 * oracle (ssdo_process_lean: the whole per-frame path, pointcloud.cpp:608-626 restated) `reps` times on its copy.  Frames are
 * independent (Pointcloud::process is stateless), so this is what eight host threads feeding eight GPUs would be replaced by.  glibc
 * gives every thread an allocation arena of its own, so the oracle's per-frame vectors do not meet in one lock.  The wall time is
 * taken from the first thread's start to the last thread's end.
 * A second leg measures what the host's memory delivers to the same threads - each one reading its private 9.4 MB frame again and
 * again - so that the line can say whether the cores or the memory bound the figure.
 */
#include "ssd_oracle.h"

#include <atomic>
#include <chrono>
#include <cstring>
#include <memory>
#include <thread>
#include <vector>

#include <pthread.h>
#include <sched.h>

namespace
{
using clk = std::chrono::steady_clock;

void pin_to(int cpu)
{
  if(cpu < 0)
    return;
  cpu_set_t set;
  CPU_ZERO(&set);
  CPU_SET(cpu, &set);
  pthread_setaffinity_np(pthread_self(), sizeof(set), &set);
}
} // namespace

extern "C" int ssdo_process_many(const ssdo_config *cfg, const ssdo_calibration *cal, const float *const *frames, int n_distinct,
                                 const int *cpus, int n_threads, int reps, double *wall_seconds, double *read_gb_per_s,
                                 long long *steps_total)
{
  if(!cfg || !cal || !frames || n_distinct < 1 || n_threads < 1 || reps < 1 || !wall_seconds)
    return -1;
  const size_t floats = size_t(cfg->width) * size_t(cfg->height) * 3;
  std::atomic<int> ready{ 0 }, go{ 0 }, failed{ 0 };
  std::atomic<long long> steps{ 0 };
  std::vector<clk::time_point> t0(n_threads), t1(n_threads), r0(n_threads), r1(n_threads);
  std::vector<double> sink(n_threads, 0.0);
  std::vector<std::thread> pool;
  pool.reserve(n_threads);
  for(int t = 0; t < n_threads; t++)
    pool.emplace_back([&, t]()
    {
      pin_to(cpus ? cpus[t] : -1);
      std::unique_ptr<float[]> mine(new float[floats]);
      std::memcpy(mine.get(), frames[t % n_distinct], floats * sizeof(float));        /* first touch on this thread's node */
      std::unique_ptr<double[]> out(new double[SSDO_MAX_STEPS * 9]);
      ready.fetch_add(1);
      while(go.load(std::memory_order_acquire) < 1)
        std::this_thread::yield();
      t0[t] = clk::now();
      long long n = 0;
      for(int r = 0; r < reps; r++)
      {
        int status = 0;
        const int k = ssdo_process_lean(cfg, cal, mine.get(), out.get(), &status);
        if(k < 0)
          failed.fetch_add(1);
        else
          n += k;
      }
      t1[t] = clk::now();
      steps.fetch_add(n);
      /* second leg: the memory's rate for these threads (a plain read of the private frame, summed so that it is not removed) */
      ready.fetch_add(1);
      while(go.load(std::memory_order_acquire) < 2)
        std::this_thread::yield();
      r0[t] = clk::now();
      double s = 0.0;
      for(int r = 0; r < 8; r++)
      {
        const float *p = mine.get();
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        for(size_t i = 0; i + 3 < floats; i += 4)
        {
          a0 += p[i]; a1 += p[i + 1]; a2 += p[i + 2]; a3 += p[i + 3];
        }
        s += double(a0) + a1 + a2 + a3;
      }
      r1[t] = clk::now();
      sink[t] = s;
    });
  while(ready.load() < n_threads)
    std::this_thread::yield();
  go.store(1, std::memory_order_release);
  while(ready.load() < 2 * n_threads)
    std::this_thread::yield();
  go.store(2, std::memory_order_release);
  for(auto &th : pool)
    th.join();
  clk::time_point a = t0[0], b = t1[0], ra = r0[0], rb = r1[0];
  for(int t = 1; t < n_threads; t++)
  {
    if(t0[t] < a) a = t0[t];
    if(t1[t] > b) b = t1[t];
    if(r0[t] < ra) ra = r0[t];
    if(r1[t] > rb) rb = r1[t];
  }
  *wall_seconds = std::chrono::duration<double>(b - a).count();
  if(read_gb_per_s)
  {
    const double rs = std::chrono::duration<double>(rb - ra).count();
    *read_gb_per_s = rs > 0.0 ? 8.0 * double(n_threads) * double(floats) * sizeof(float) / rs * 1e-9 : 0.0;
    double keep = 0.0;
    for(double v : sink)
      keep += v;
    if(keep == 12345.678)          /* the sums are used */
      *read_gb_per_s += 1e-30;
  }
  if(steps_total)
    *steps_total = steps.load();
  return failed.load() ? -2 : 0;
}
