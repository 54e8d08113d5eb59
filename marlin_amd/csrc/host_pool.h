// A small persistent pool of host threads shared by the prover (hiding parts of the commitments, prover.hip) and the fixed-base
// MSM's result combination (capi.hip: FbRun::finish).
#pragma once
#include <condition_variable>
#include <deque>
#include <functional>
#include <future>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>
#include "host_ff.h"

namespace mh {
// A few persistent host threads for the small host-side group operations that run beside a device batch (the hiding parts of
// the commitments).  std::async starts a thread per call: ~30 us each, four to eight of them in a row right before the round's
// MSM is launched -- 0.13 ms of idle GPU per commit round in the kernel trace (profiles/r03x_dispatch_gaps_*).
class HostPool {
 public:
  template <class F>
  auto submit(F f) -> std::future<decltype(f())> {
    auto task = std::make_shared<std::packaged_task<decltype(f())()>>(std::move(f));
    auto fut = task->get_future();
    {
      std::lock_guard<std::mutex> lk(mu_);
      if (workers_.empty()) for (int i = 0; i < 8; i++) workers_.emplace_back([this] { run(); });
      q_.emplace_back([task] { (*task)(); });
    }
    cv_.notify_one();
    return fut;
  }
  ~HostPool() {
    { std::lock_guard<std::mutex> lk(mu_); stop_ = true; }
    cv_.notify_all();
    for (auto& t : workers_) t.join();
  }
 private:
  void run() {
    for (;;) {
      std::function<void()> job;
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [this] { return stop_ || !q_.empty(); });
        if (q_.empty()) return;
        job = std::move(q_.front()); q_.pop_front();
      }
      job();
    }
  }
  std::mutex mu_; std::condition_variable cv_; std::deque<std::function<void()>> q_; std::vector<std::thread> workers_; bool stop_ = false;
};
inline HostPool& host_pool() { static HostPool p; return p; }
// The pool's tasks read locals of the submitting frame through pointers (blinding vectors, witness quotients), and a
// packaged_task's future does not block in its destructor the way std::async's did: every frame that submits registers its
// futures here, so that ANY way out of it -- an MH_TRY in between included -- first waits for the tasks still running.
struct WaitAll {
  std::vector<std::future<hostff::HG1>*> fs;
  WaitAll() = default;
  WaitAll(std::initializer_list<std::future<hostff::HG1>*> l) : fs(l) {}
  void add(std::vector<std::future<hostff::HG1>>& v) { for (auto& f : v) fs.push_back(&f); }
  ~WaitAll() { for (auto* f : fs) if (f->valid()) f->wait(); }
};

}  // namespace mh
