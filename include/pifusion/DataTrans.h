// DataTrans.h -- the tracker->fusion queue of the reference (src/DataTrans.h:10-83) for C++
// users of this library: singleton per payload type, capacity 30, product() drops the OLDEST
// element when full, consumption() blocks until an element is available.
#ifndef PIFUSION_DATATRANS_H
#define PIFUSION_DATATRANS_H
#include <condition_variable>
#include <deque>
#include <mutex>

template <typename T>
class DataTrans {
public:
    static DataTrans& Instance() { static DataTrans inst; return inst; }
    void product(const T& v)
    {
        {
            std::lock_guard<std::mutex> l(mu_);
            while (q_.size() >= kMax) { q_.pop_front(); ++dropped_; }
            q_.push_back(v);
        }
        not_empty_.notify_one();
    }
    void consumption(T& v)
    {
        std::unique_lock<std::mutex> l(mu_);
        not_empty_.wait(l, [this] { return !q_.empty(); });
        v = q_.front();
        q_.pop_front();
    }
    size_t size() { std::lock_guard<std::mutex> l(mu_); return q_.size(); }
    size_t dropped() { std::lock_guard<std::mutex> l(mu_); return dropped_; }
private:
    DataTrans() {}
    static constexpr size_t kMax = 30;
    std::deque<T> q_;
    std::mutex mu_;
    std::condition_variable not_empty_;
    size_t dropped_ = 0;
};
#endif
