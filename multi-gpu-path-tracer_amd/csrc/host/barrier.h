// barrier.h -- reusable thread barrier with the reference's interface (barrier.h:19-70:
// Barrier(n), wait(), reset(n)); generation-counted condition variable.
#pragma once

#include <condition_variable>
#include <cstddef>
#include <mutex>
#include <stdexcept>

namespace ptamd {

class Barrier {
public:
    explicit Barrier(std::size_t num) : expected_(num)
    {
        if (num == 0) throw std::invalid_argument("Barrier thread count cannot be 0");
    }
    Barrier(const Barrier &) = delete;
    Barrier &operator=(const Barrier &) = delete;

    void wait()
    {
        std::unique_lock<std::mutex> lock(m_);
        const std::size_t gen = generation_;
        if (++arrived_ == expected_) {
            arrived_ = 0;
            ++generation_;
            cv_.notify_all();
            return;
        }
        cv_.wait(lock, [&] { return generation_ != gen; });
    }

    void reset(std::size_t new_thread_num)
    {
        std::unique_lock<std::mutex> lock(m_);
        expected_ = new_thread_num;
        arrived_ = 0;
        generation_ = 0;
    }

private:
    std::size_t expected_, arrived_ = 0, generation_ = 0;
    std::mutex m_;
    std::condition_variable cv_;
};

} // namespace ptamd
