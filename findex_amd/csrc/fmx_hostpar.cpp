// fmx_hostpar.cpp -- see fmx_hostpar.h.
#include "fmx_hostpar.h"

#include <sched.h>

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <vector>

namespace fmx {

static std::atomic<unsigned> g_threads_override{0};

static unsigned detect_threads() {
  unsigned n = std::max(1u, std::thread::hardware_concurrency());
  cpu_set_t set;
  if (sched_getaffinity(0, sizeof set, &set) == 0) n = std::max(1, CPU_COUNT(&set));
  // cgroup v2: "<quota> <period>" or "max <period>"; cgroup v1: two files
  if (FILE *f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
    char q[32] = {0};
    long period = 0;
    if (std::fscanf(f, "%31s %ld", q, &period) == 2 && q[0] != 'm' && period > 0) {
      const long quota = std::atol(q);
      if (quota > 0) n = std::min<unsigned>(n, (unsigned)std::max(1l, (quota + period - 1) / period));
    }
    std::fclose(f);
  } else if (FILE *fq = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
    long quota = -1, period = 0;
    if (std::fscanf(fq, "%ld", &quota) != 1) quota = -1;
    std::fclose(fq);
    if (FILE *fp = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
      if (std::fscanf(fp, "%ld", &period) != 1) period = 0;
      std::fclose(fp);
    }
    if (quota > 0 && period > 0) n = std::min<unsigned>(n, (unsigned)std::max(1l, (quota + period - 1) / period));
  }
  return std::min(n, 256u);
}

unsigned host_threads() {
  const unsigned o = g_threads_override.load(std::memory_order_relaxed);
  if (o) return o;
  static const unsigned detected = detect_threads();
  return detected;
}

void set_host_threads(unsigned n) { g_threads_override.store(std::min(n, 1024u), std::memory_order_relaxed); }

void parallel_for(size_t n, size_t grain, const std::function<void(size_t, size_t)> &body) {
  if (!n) return;
  if (!grain) grain = 1;
  const size_t chunks = (n + grain - 1) / grain;
  const unsigned nt = (unsigned)std::min<size_t>(host_threads(), chunks);
  if (nt <= 1) { body(0, n); return; }
  std::atomic<size_t> next{0};
  auto work = [&]() {
    for (;;) {
      const size_t c = next.fetch_add(1, std::memory_order_relaxed);
      if (c >= chunks) return;
      body(c * grain, std::min(n, (c + 1) * grain));
    }
  };
  std::vector<std::thread> th;
  th.reserve(nt - 1);
  for (unsigned t = 1; t < nt; t++) {
    try {
      th.emplace_back(work);
    } catch (...) {       // no more threads to be had (std::system_error): the ones that exist, and this one, take all chunks
      break;
    }
  }
  work();
  for (std::thread &t : th) t.join();
}

Worker::Worker() : th_([this] { run(); }) {}

Worker::~Worker() {
  {
    std::lock_guard<std::mutex> lk(mu_);
    stop_ = true;
  }
  cv_job_.notify_all();
  th_.join();
}

void Worker::submit(std::function<void()> job) {
  {
    std::lock_guard<std::mutex> lk(mu_);
    q_.push_back(std::move(job));
    busy_++;
  }
  cv_job_.notify_one();
}

void Worker::wait() {
  std::unique_lock<std::mutex> lk(mu_);
  cv_idle_.wait(lk, [this] { return busy_ == 0; });
}

void Worker::run() {
  for (;;) {
    std::function<void()> job;
    {
      std::unique_lock<std::mutex> lk(mu_);
      cv_job_.wait(lk, [this] { return stop_ || !q_.empty(); });
      if (q_.empty()) return;             // stop requested and nothing left
      job = std::move(q_.front());
      q_.pop_front();
    }
    job();
    {
      std::lock_guard<std::mutex> lk(mu_);
      busy_--;
    }
    cv_idle_.notify_all();
  }
}

}  // namespace fmx
