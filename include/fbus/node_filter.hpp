// node_filter.hpp -- ONE host object that owns the filters of a whole node: several devices, one process (header-only, over the C ABI).
//
// The reference's filter is one stateful object driven by one thread (C++/src/filter.cpp:190-250: FILTER::FilterThreadFunction).
// fbus::NodeFilter is its counterpart for a batch that spans the GPUs of a node: `total` independent filters (no cross-filter term in
// ImuUpdate.m:36-82 / MeasureUpdate.m:37-103) cut into contiguous, 64-aligned shards, shard k on devices[k] behind its own
// fbus::BatchedFilter and its own HOST THREAD, so that the launches of a step go out to all devices at once instead of one device
// after the other (a launch costs ~4-8 us of host time; tools/node_rate.cpp measures the rate per thread).  No communication while
// stepping.  The one collective of the path, the gather of the packed records:
//   gather(out)       every device receives all records (RCCL all-gather / grouped broadcasts over xGMI; needs one device per shard),
//   gather_to(k, out) device devices[k] receives all records by peer copies on the shards' streams (no communicator; also works
//                     when shards share a device -- how a one-GPU box rehearses the class).
// The kernel-family choice of every shard is keyed on the WHOLE job (fbus_ekf_set_policy_batch): the shard layout does not change
// the result by a bit.
//
//   fbus::NodeFilter<float> node(262144, {0, 1, 2, 3, 4, 5, 6, 7}, prm);
//   node.for_each_shard([&](int k, fbus::BatchedFilter<float>& f) { f.predict_dev(acc[k], gyr[k], dt[k]); });   // concurrently
//   node.sync();  node.gather_to(0, out_on_device_0);
#pragma once
#include "batched_filter.hpp"

#include <algorithm>
#include <condition_variable>
#include <exception>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

namespace fbus {

template <typename Real = float>
class NodeFilter {
public:
    static constexpr int TILE = 64;
    // [lo, hi) of the filters shard k owns: contiguous, tile (64) aligned, sizes differ by at most one tile (= ShardedFilter's rule)
    static void shard_range(long total, int k, int shards, long& lo, long& hi)
    {
        const long tiles = (total + TILE - 1) / TILE;
        lo = std::min(tiles * k / shards * TILE, total);
        hi = std::min(tiles * (k + 1) / shards * TILE, total);
    }

    NodeFilter(long total, const std::vector<int>& devices, const fbus_params& prm, int nstate = 18)
        : total_(total), devices_(devices)
    {
        const int n = int(devices.size());
        if (n < 1) throw Error(FBUS_ERR_INVALID, "NodeFilter: no devices");
        for (int k = 0; k < n; ++k) {
            long lo, hi;
            shard_range(total, k, n, lo, hi);
            if (hi <= lo) throw Error(FBUS_ERR_INVALID, "NodeFilter: empty shard (more devices than 64-filter tiles)");
            shards_.emplace_back(new BatchedFilter<Real>(int(hi - lo), prm, devices[k], nstate));
            check(k, fbus_ekf_set_policy_batch(shards_[k]->handle(), int(std::min<long>(total, 0x7fffffff))), "fbus_ekf_set_policy_batch");
        }
        size_t bpf = 0;
        check(0, fbus_ekf_records(shards_[0]->handle(), nullptr, &bpf, nullptr), "fbus_ekf_records");
        for (int k = 0; k < n; ++k) bytes_.push_back(size_t((hi(k) - lo(k) + TILE - 1) / TILE * TILE) * bpf);
        for (int k = 0; k < n; ++k) workers_.emplace_back(new Worker());
    }
    ~NodeFilter()
    {
        for (auto& w : workers_) w->stop();
    }
    NodeFilter(const NodeFilter&) = delete;
    NodeFilter& operator=(const NodeFilter&) = delete;

    int shards() const { return int(shards_.size()); }
    long total() const { return total_; }
    long lo(int k) const { long a, b; shard_range(total_, k, shards(), a, b); return a; }
    long hi(int k) const { long a, b; shard_range(total_, k, shards(), a, b); return b; }
    int device(int k) const { return devices_[k]; }
    BatchedFilter<Real>& shard(int k) { return *shards_[k]; }

    // fn(k, shard k) on every shard's own host thread, all at once; returns when all have returned (the first exception is rethrown)
    template <typename F>
    void for_each_shard(F&& fn)
    {
        const int n = shards();
        std::vector<std::exception_ptr> err(n);
        for (int k = 0; k < n; ++k)
            workers_[k]->post([&, k] { try { fn(k, *shards_[k]); } catch (...) { err[k] = std::current_exception(); } });
        for (int k = 0; k < n; ++k) workers_[k]->wait();
        for (int k = 0; k < n; ++k) if (err[k]) std::rethrow_exception(err[k]);
    }
    void sync() { for_each_shard([](int, BatchedFilter<Real>& f) { f.sync(); }); }

    // ---- the gather of the packed records ---------------------------------------------------------------------------------
    size_t gathered_bytes() const { size_t s = 0; for (size_t b : bytes_) s += b; return s; }
    size_t offset_of(int k) const { size_t s = 0; for (int j = 0; j < k; ++j) s += bytes_[j]; return s; }
    // every device receives all records: out[k] = gathered_bytes() of memory on devices[k].  RCCL (ncclCommInitAll at first use).
    void gather(const std::vector<void*>& out)
    {
        std::vector<fbus_ekf_t> hs;
        for (auto& s : shards_) hs.push_back(s->handle());
        if (!comm_) { check(0, fbus_ekf_comm_init_all(hs.data(), int(hs.size())), "fbus_ekf_comm_init_all"); comm_ = true; }
        check(0, fbus_ekf_gather_group(hs.data(), int(hs.size()), out.data(), bytes_.data()), "fbus_ekf_gather_group");
    }
    // devices[dst] receives all records (gathered_bytes() at `out`), shard k at offset_of(k): peer copies on the shards' streams,
    // complete after sync()
    void gather_to(int dst, void* out)
    {
        for (int k = 0; k < shards(); ++k)
            check(k, fbus_ekf_copy_records(shards_[k]->handle(), static_cast<char*>(out) + offset_of(k), devices_[dst]), "fbus_ekf_copy_records");
    }

private:
    // one host thread per shard, fed one job at a time
    struct Worker {
        std::mutex m;
        std::condition_variable cv;
        std::function<void()> job;
        bool busy = false, quit = false;
        std::thread t;
        Worker() : t([this] { run(); }) {}
        void run()
        {
            std::unique_lock<std::mutex> l(m);
            for (;;) {
                cv.wait(l, [this] { return busy || quit; });
                if (quit) return;
                l.unlock();
                job();
                l.lock();
                busy = false;
                cv.notify_all();
            }
        }
        void post(std::function<void()> j)
        {
            std::unique_lock<std::mutex> l(m);
            cv.wait(l, [this] { return !busy; });
            job = std::move(j);
            busy = true;
            cv.notify_all();
        }
        void wait()
        {
            std::unique_lock<std::mutex> l(m);
            cv.wait(l, [this] { return !busy; });
        }
        void stop()
        {
            { std::unique_lock<std::mutex> l(m); cv.wait(l, [this] { return !busy; }); quit = true; cv.notify_all(); }
            if (t.joinable()) t.join();
        }
    };
    void check(int k, int rc, const char* where) const
    {
        if (rc == FBUS_OK) return;
        const char* d = fbus_ekf_last_error(shards_[k]->handle());
        throw Error(rc, std::string(where) + " (shard " + std::to_string(k) + "): " + fbus_status_string(rc) + ((d && *d) ? std::string(" (") + d + ")" : std::string()));
    }
    long total_;
    std::vector<int> devices_;
    std::vector<std::unique_ptr<BatchedFilter<Real>>> shards_;
    std::vector<std::unique_ptr<Worker>> workers_;
    std::vector<size_t> bytes_;
    bool comm_ = false;
};

}  // namespace fbus
