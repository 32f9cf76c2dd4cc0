// sharded_filter.hpp -- one rank's share of a batch of filters on one GPU of a multi-GPU job (header-only, over the C ABI).
//
// The reference has no counterpart: it runs ONE filter on one CPU thread (C++/src/filter.cpp:190-250).  The filters of a batch
// are independent (no cross-filter term in ImuUpdate.m:36-82 / MeasureUpdate.m:37-103), so a job of `total` filters is cut
// into contiguous, 64-aligned ranges -- one process (rank) per GPU -- that step with NO communication; the one collective of the
// path is the gather of the packed records at the end, issued by the library on the handle's stream over RCCL
// (fbus_ekf_gather: ncclAllGather for equal shards, a grouped ncclBroadcast per rank for ragged ones; xGMI inside a node).
//
//   rank 0:  id = fbus::ShardedFilter<float>::unique_id();            -> hand the 128 bytes to every rank (MPI_Bcast, a file, ...)
//   rank r:  fbus::ShardedFilter<float> f(total, r, world, id, prm, /*device*/ r);
//            f.filter().predict_dev(...); f.filter().correct_dev(...);   // this rank's filters [f.lo(), f.hi())
//            f.gather(out_dev);                                       // every rank's records, rank k at f.offset_of(k)
#pragma once
#include "batched_filter.hpp"

#include <algorithm>
#include <array>
#include <vector>

namespace fbus {

template <typename Real = float>
class ShardedFilter {
public:
    using UniqueId = std::array<char, 128>;
    static constexpr int TILE = 64;

    // [lo, hi) of the filters `rank` owns: contiguous, tile (64) aligned, sizes differ by at most one tile
    static void shard_range(long total, int rank, int world, long& lo, long& hi)
    {
        const long tiles = (total + TILE - 1) / TILE;
        lo = std::min(tiles * rank / world * TILE, total);
        hi = std::min(tiles * (rank + 1) / world * TILE, total);
    }
    static UniqueId unique_id()
    {
        UniqueId id;
        if (int rc = fbus_ekf_comm_unique_id(id.data())) throw Error(rc, std::string("fbus_ekf_comm_unique_id: ") + fbus_status_string(rc));
        return id;
    }

    ShardedFilter(long total, int rank, int world, const UniqueId& id, const fbus_params& prm, int device, int nstate = 18)
        : total_(total), rank_(rank), world_(world), flt_(make(total, rank, world), prm, device, nstate)
    {
        // the kernel-family choice is keyed on the WHOLE job, not on this shard: every shard layout runs the same kernels and
        // produces the bits of the single-handle run (team and one-wave kernels agree to fp32 rounding only)
        check(fbus_ekf_set_policy_batch(flt_.handle(), int(std::min<long>(total, 0x7fffffff))), "fbus_ekf_set_policy_batch");
        check(fbus_ekf_comm_init(flt_.handle(), id.data(), rank, world), "fbus_ekf_comm_init");
        size_t bpf = 0;
        check(fbus_ekf_records(flt_.handle(), nullptr, &bpf, nullptr), "fbus_ekf_records");
        bytes_.resize(world);
        for (int k = 0; k < world; ++k) {
            long lo, hi;
            shard_range(total, k, world, lo, hi);
            // a rank's records are whole 64-filter tiles: (filters rounded up to a tile) x bytes per filter
            bytes_[k] = size_t((hi - lo + TILE - 1) / TILE * TILE) * bpf;
        }
    }

    BatchedFilter<Real>& filter() { return flt_; }
    long lo() const { long a, b; shard_range(total_, rank_, world_, a, b); return a; }
    long hi() const { long a, b; shard_range(total_, rank_, world_, a, b); return b; }
    size_t gathered_bytes() const { size_t s = 0; for (size_t b : bytes_) s += b; return s; }
    size_t offset_of(int rank) const { size_t s = 0; for (int k = 0; k < rank; ++k) s += bytes_[k]; return s; }

    // all ranks' packed records into out_dev (gathered_bytes() of device memory) on every rank; asynchronous on the
    // filter's stream until filter().sync()
    void gather(void* out_dev) { check(fbus_ekf_gather(flt_.handle(), out_dev, bytes_.data()), "fbus_ekf_gather"); }

private:
    static int make(long total, int rank, int world)
    {
        long lo, hi;
        shard_range(total, rank, world, lo, hi);
        if (hi <= lo) throw Error(FBUS_ERR_INVALID, "ShardedFilter: empty shard (more ranks than 64-filter tiles)");
        return int(hi - lo);
    }
    void check(int rc, const char* where) const
    {
        if (rc == FBUS_OK) return;
        const char* d = fbus_ekf_last_error(flt_.handle());
        throw Error(rc, std::string(where) + ": " + fbus_status_string(rc) + ((d && *d) ? std::string(" (") + d + ")" : std::string()));
    }
    long total_;
    int rank_, world_;
    BatchedFilter<Real> flt_;
    std::vector<size_t> bytes_;

};

}  // namespace fbus
