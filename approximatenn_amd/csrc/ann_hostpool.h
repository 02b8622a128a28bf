// ann_hostpool.h -- a small persistent pool of host threads for the byte work around the host-pointer ABI:
// copying caller buffers into pinned staging memory and hashing an index for the residency cache's strict mode.
//
// query() hands over pageable host pointers on every call (/root/reference/ann.h:61-62).  One core copies about
// 10 GB/s, so the 5 MB of a cfg3 batch cost ~0.5 ms on one thread -- a third of the whole GPU step -- and the 8 GB of
// a cfg3 index hash in ~1 s.  Spread over the pool both are memory-bound instead.  Threads are created on first use
// and park on a condition variable between jobs (wake-up ~10 us).
#ifndef APPROXNN_HIP_ANN_HOSTPOOL_H
#define APPROXNN_HIP_ANN_HOSTPOOL_H
#include <string.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

class HostPool {
 public:
  static HostPool &get() {
    static HostPool *p = new HostPool();  // never destroyed: worker threads may outlive static destructors otherwise
    return *p;
  }
  unsigned threads() const { return nthr_; }

  // fn(i) for i in [0, items), spread over the pool; returns when all are done.  Not re-entrant (one job at a time).
  void run(size_t items, const std::function<void(size_t)> &fn) {
    if (!items) return;
    if (items == 1 || nthr_ <= 1) {
      for (size_t i = 0; i < items; i++) fn(i);
      return;
    }
    std::lock_guard<std::mutex> one_job(job_mu_);
    start_workers();
    {
      std::lock_guard<std::mutex> lk(mu_);
      fn_ = &fn, items_ = items;
      next_.store(0);
      pending_ = (unsigned)workers_.size();
      generation_++;
    }
    cv_.notify_all();
    work();  // the caller takes part
    std::unique_lock<std::mutex> lk(mu_);
    done_cv_.wait(lk, [&] { return pending_ == 0; });
    fn_ = NULL;
  }

  // The same, asynchronously: begin() hands the items to the WORKERS only and returns at once; the caller goes on (e.g.
  // watches per-item completion flags the items set, and feeds finished pieces to the GPU) and calls end() to wait for
  // the job.  With a pool of one thread begin() runs the items inline.
  void begin(size_t items, const std::function<void(size_t)> &fn) {
    job_mu_.lock();
    async_inline_ = false;
    if (!items) return;
    if (nthr_ <= 1) {
      for (size_t i = 0; i < items; i++) fn(i);
      return;
    }
    start_workers();
    {
      std::lock_guard<std::mutex> lk(mu_);
      fn_ = &fn, items_ = items;
      next_.store(0);
      pending_ = (unsigned)workers_.size();
      generation_++;
    }
    async_inline_ = true;
    cv_.notify_all();
  }
  void end() {
    if (async_inline_) {
      work();  // help with whatever is left
      std::unique_lock<std::mutex> lk(mu_);
      done_cv_.wait(lk, [&] { return pending_ == 0; });
      fn_ = NULL;
    }
    job_mu_.unlock();
  }

  // memcpy spread over the pool in pieces of at least `grain` bytes
  void copy(void *dst, const void *src, size_t bytes, size_t grain = (size_t)1 << 18) {
    if (bytes <= 2 * grain || nthr_ <= 1) {
      memcpy(dst, src, bytes);
      return;
    }
    const size_t pieces = std::min<size_t>(nthr_, (bytes + grain - 1) / grain);
    const size_t per = ((bytes + pieces - 1) / pieces + 63) & ~(size_t)63;
    run(pieces, [&](size_t i) {
      const size_t a = i * per, b = std::min(bytes, a + per);
      if (a < b) memcpy((char *)dst + a, (const char *)src + a, b - a);
    });
  }

 private:
  HostPool() {
    unsigned hw = std::thread::hardware_concurrency();
    const char *e = getenv("ANN_HIP_HOST_THREADS");
    unsigned want = e && atoi(e) > 0 ? (unsigned)atoi(e) : std::min(hw ? hw : 1u, 16u);
    nthr_ = std::max(1u, std::min(want, 64u));
  }
  void start_workers() {
    if (!workers_.empty() || nthr_ <= 1) return;
    for (unsigned t = 0; t + 1 < nthr_; t++) workers_.emplace_back([this] { loop(); });
    for (auto &w : workers_) w.detach();
  }
  void work() {
    for (;;) {
      const size_t i = next_.fetch_add(1);
      if (i >= items_) break;
      (*fn_)(i);
    }
  }
  void loop() {
    unsigned long seen = 0;
    for (;;) {
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return generation_ != seen; });
        seen = generation_;
      }
      work();
      std::lock_guard<std::mutex> lk(mu_);
      if (--pending_ == 0) done_cv_.notify_one();
    }
  }
  unsigned nthr_ = 1;
  bool async_inline_ = false;
  std::vector<std::thread> workers_;
  std::mutex mu_, job_mu_;
  std::condition_variable cv_, done_cv_;
  const std::function<void(size_t)> *fn_ = NULL;
  size_t items_ = 0;
  std::atomic<size_t> next_{0};
  unsigned pending_ = 0;
  unsigned long generation_ = 0;
};
#endif
