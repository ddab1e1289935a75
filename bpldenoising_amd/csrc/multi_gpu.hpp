// multi_gpu.hpp -- one handle over several MI355X devices, driven from ONE host thread.
//
// Why it exists: the reference's caller is a single Julia task (bilevel_learn,
// /root/reference/src/TRBox.jl:192-273; the only other task is the GR visualiser,
// /root/reference/src/BilevelVisualise.jl:283), so the drop-in form of "images sharded across the 8 GPUs of
// one node" is ONE handle that fans out inside the library:
//   * the O images are block-distributed over the shards (shard_range below == sharding.shard_range);
//     images are independent ROF problems sharing only alpha
//     (/root/reference/src/TVLearningFunctionVec.jl:57-65), so there is no halo and no data-path collective;
//   * every shard is an ordinary single-device handle (all kernels, graphs and workspaces of bpltv.hip) with
//     its own persistent worker thread, which owns the device context of that shard;
//   * per evaluation ONE RCCL collective over xGMI on the [cost, grad...] vector (the loss is a plain sum over
//     all entries, :20; both gradient wrappers sum per-image terms, :76-82, :168-173):
//     ncclAllReduce(sum, f64) in place on the shards' partial vectors, or -- params.deterministic, scalar /
//     patch parameters -- ncclAllGather of the per-image rows, added in global image order on the host so
//     that the totals are bitwise independent of the number of GPUs;
//   * u shards go device-to-host straight into the caller's slices.
// The communicator comes from ncclCommInitAll (one process, one rank per device).  RCCL cannot place two
// ranks on one device; a handle whose device list repeats a device (bpltv_create_sharded: rehearsal on a
// one-GPU box) performs the same reduction on the host instead and says so in stats.collective.
#pragma once
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <condition_variable>
#include <functional>
#include <mutex>
#include <set>
#include <thread>
#include <vector>

#include "tiling.hpp"   // shard_range

namespace bpltv {

// One persistent host thread per shard: it sets its device once and then runs the jobs the caller's thread
// hands it.  run_all() posts one job per worker and waits for all of them, so the library is quiescent when
// an ABI call returns (threading).
class ShardWorker {
public:
    explicit ShardWorker(int device) : device_(device), th_([this] { loop(); }) {}
    ~ShardWorker() {
        {
            std::lock_guard<std::mutex> lk(m_);
            quit_ = true;
        }
        cv_.notify_all();
        th_.join();
    }
    void post(std::function<int()> job) {
        {
            std::lock_guard<std::mutex> lk(m_);
            job_ = std::move(job);
            has_job_ = true;
            done_ = false;
        }
        cv_.notify_all();
    }
    int wait() {
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [this] { return done_; });
        return rc_;
    }

private:
    void loop() {
        (void)hipSetDevice(device_);
        for (;;) {
            std::function<int()> job;
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [this] { return has_job_ || quit_; });
                if (quit_ && !has_job_) return;
                job = std::move(job_);
                has_job_ = false;
            }
            const int rc = job();
            {
                std::lock_guard<std::mutex> lk(m_);
                rc_ = rc;
                done_ = true;
            }
            cv_.notify_all();
        }
    }
    int device_;
    std::mutex m_;
    std::condition_variable cv_;
    std::function<int()> job_;
    bool has_job_ = false, done_ = true, quit_ = false;
    int rc_ = 0;
    std::thread th_;   // last member: the thread starts after everything above is initialised
};

// pack the per-image pieces of the last evaluate into rows [cost_k, grad_k[0..P)], zero padded to `maxloc`
// rows -- the send buffer of the all-gather.  perimg[O], gred[P][O] (patch_sum_kernel's layout).
__global__ void pack_rows_kernel(const double* __restrict__ perimg, const double* __restrict__ gred, int O, int P,
                                 int maxloc, double* __restrict__ rows) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    const int W = 1 + P;
    if (e >= maxloc * W) return;
    const int k = e / W, c = e - k * W;
    double v = 0.0;
    if (k < O) v = (c == 0) ? perimg[k] : gred[(size_t)(c - 1) * O + k];
    rows[e] = v;
}

}  // namespace bpltv
