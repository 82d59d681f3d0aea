// fdoct_hostcopy.h -- the host side of fdoct_process's chunk pipeline when the caller's buffers are PAGEABLE (what cv::Mat
// owns: the buffers the patch of INTEGRATION.md 1 hands over).  The HIP runtime stages a pageable copy through its own
// bounce buffer on the calling thread, one direction at a time, so the three-stream pipeline of fdoct_capi.cpp degenerates
// to upload -> kernels -> download in sequence (round 5: 1.5e6 A-scans/s against 7.8e6 from pinned memory).  Here the
// library owns pinned staging slots and a few threads move a chunk between them and the caller's memory while the DMA
// engines and the kernels work on the neighbouring chunks.  Nothing here touches the device.
#pragma once
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

namespace fdoct_impl {

class HostCopyPool {
 public:
  // `threads` counts the calling thread: threads - 1 workers are started (none for threads <= 1: copies run inline).
  explicit HostCopyPool(int threads) {
    const int workers = threads > 1 ? threads - 1 : 0;
    try {
      for (int i = 0; i < workers; i++) th_.emplace_back([this, i] { worker(i + 1); });
    } catch (...) {  // thread creation refused (a process limit): work with the ones that started
    }
  }
  ~HostCopyPool() {
    {
      std::lock_guard<std::mutex> lk(m_);
      stop_ = true;
    }
    cv_job_.notify_all();
    for (auto& t : th_) t.join();
  }
  HostCopyPool(const HostCopyPool&) = delete;
  HostCopyPool& operator=(const HostCopyPool&) = delete;

  int threads() const { return (int)th_.size() + 1; }

  // rows x width bytes from (src, spitch) to (dst, dpitch); returns when every byte has been written.
  void copy2d(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t rows) {
    if (!rows || !width) return;
    Job j{static_cast<unsigned char*>(dst), static_cast<const unsigned char*>(src), dpitch, spitch, width, rows, 1};
    if (th_.empty() || rows * width < (size_t)1 << 20) {  // under a megabyte the wake-ups cost more than they bring
      run(j, 0);
      return;
    }
    j.parts = (int)th_.size() + 1;
    {
      std::lock_guard<std::mutex> lk(m_);
      job_ = j;
      pending_ = (int)th_.size();
      ++gen_;
    }
    cv_job_.notify_all();
    run(j, 0);
    std::unique_lock<std::mutex> lk(m_);
    cv_done_.wait(lk, [this] { return pending_ == 0; });
  }
  void copy(void* dst, const void* src, size_t bytes) {
    // one "row" per 64 KB so that the split is even whatever the size
    const size_t piece = (size_t)64 << 10;
    const size_t whole = bytes / piece;
    if (whole) copy2d(dst, piece, src, piece, piece, whole);
    if (bytes % piece) std::memcpy(static_cast<unsigned char*>(dst) + whole * piece, static_cast<const unsigned char*>(src) + whole * piece, bytes % piece);
  }

 private:
  struct Job {
    unsigned char* dst;
    const unsigned char* src;
    size_t dpitch, spitch, width, rows;
    int parts;
  };
  static void run(const Job& j, int part) {
    const size_t r0 = j.rows * (size_t)part / (size_t)j.parts, r1 = j.rows * (size_t)(part + 1) / (size_t)j.parts;
    if (r0 >= r1) return;
    if (j.dpitch == j.width && j.spitch == j.width) {
      std::memcpy(j.dst + r0 * j.width, j.src + r0 * j.width, (r1 - r0) * j.width);
      return;
    }
    for (size_t r = r0; r < r1; r++) std::memcpy(j.dst + r * j.dpitch, j.src + r * j.spitch, j.width);
  }
  void worker(int part) {
    uint64_t seen = 0;
    for (;;) {
      Job j;
      {
        std::unique_lock<std::mutex> lk(m_);
        cv_job_.wait(lk, [&] { return stop_ || gen_ != seen; });
        if (stop_) return;
        seen = gen_;
        j = job_;
      }
      run(j, part);
      {
        std::lock_guard<std::mutex> lk(m_);
        if (--pending_ == 0) cv_done_.notify_one();
      }
    }
  }
  std::vector<std::thread> th_;
  std::mutex m_;
  std::condition_variable cv_job_, cv_done_;
  Job job_{};
  uint64_t gen_ = 0;
  int pending_ = 0;
  bool stop_ = false;
};

}  // namespace fdoct_impl
