// fmx_hostpar.h -- the library's own host-side parallelism: how many cores it may use, a chunked parallel-for
// for the regex front-end (REParser.re2post + ReTree.apply are per-regex and independent), and a persistent
// worker thread that a handle keeps for the one-process-several-GPUs entry points.
#pragma once
#include <stddef.h>

#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <thread>

namespace fmx {

// Cores this process may really use: the affinity mask capped by the cgroup CPU quota (a container that sees
// 256 CPUs may own 16), overridable with fmx_config_set("threads", "N").
unsigned host_threads();
void set_host_threads(unsigned n);      // 0 = back to the detected value

// body(a, b) over [0, n) in chunks of `grain`, on up to host_threads() threads (the caller's included);
// chunks are handed out by an atomic counter.  Exceptions must not leave body.
void parallel_for(size_t n, size_t grain, const std::function<void(size_t, size_t)> &body);

// One thread that runs submitted jobs in order.  fmx_search_batch_multi / fmx_regex_batch_match_multi hand slice r
// to the worker of handle r instead of creating and joining a thread per slice and call.
class Worker {
 public:
  Worker();
  ~Worker();                            // finishes what was submitted, then joins
  Worker(const Worker &) = delete;
  Worker &operator=(const Worker &) = delete;
  void submit(std::function<void()> job);
  void wait();                          // until every submitted job has run

 private:
  void run();
  std::mutex mu_;
  std::condition_variable cv_job_, cv_idle_;
  std::deque<std::function<void()>> q_;
  size_t busy_ = 0;
  bool stop_ = false;
  std::thread th_;
};

}  // namespace fmx
