// Host launch rate of T concurrent threads, each driving its own handle through fbus::NodeFilter (one process, one host thread per
// shard): the evidence that a node's >= 6x scaling is not host-bound -- a step of 65 536 filters takes ~13 us on the device, so a host
// thread has to issue a launch in well under that.  Tiny shards (64 filters: the kernels take ~3 us) so that the host is what is timed.
// On a one-GPU box all shards sit on device 0.
//   g++ -std=c++14 -O2 -I include tools/node_rate.cpp -o tools/_build/node_rate -L fbus-ekf_amd/lib -lfbus_ekf -pthread
//   tools/_build/node_rate [threads ...]
#include <fbus/node_filter.hpp>
#include <chrono>
#include <cstdio>
#include <cstdlib>

int main(int argc, char** argv)
{
    std::vector<int> counts;
    for (int i = 1; i < argc; ++i) counts.push_back(std::atoi(argv[i]));
    if (counts.empty()) counts = { 1, 2, 4, 8 };
    const fbus_params prm = fbus::BatchedFilter<float>::defaults(FBUS_DIALECT_MATLAB);
    const int launches = 20000;
    for (int T : counts) {
        fbus::NodeFilter<float> node(64L * T, std::vector<int>(T, 0), prm);
        // inputs: any device memory will do (zero-initialised records of a helper handle: accel = gyro = 0, dt = 0)
        fbus::BatchedFilter<float> helper(4096, prm, 0);
        void* dev = nullptr;
        fbus_ekf_records(helper.handle(), &dev, nullptr, nullptr);
        const float* in = static_cast<const float*>(dev);
        std::vector<double> secs(T);
        node.for_each_shard([&](int k, fbus::BatchedFilter<float>& f) {
            f.reset_covariance();
            for (int i = 0; i < 200; ++i) f.predict_dev(in, in + 1024, in + 2048);
            f.sync();
            const auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < launches; ++i) f.predict_dev(in, in + 1024, in + 2048);
            const auto t1 = std::chrono::steady_clock::now();        // submission time only: what the host thread spends per launch
            f.sync();
            secs[k] = std::chrono::duration<double>(t1 - t0).count();
        });
        double worst = 0, sum = 0;
        for (double s : secs) { worst = std::max(worst, s); sum += s; }
        std::printf("%d thread(s): %.2f us per launch per thread (slowest thread %.2f us), %.3g launches/s in all\n", T,
                    sum / T / launches * 1e6, worst / launches * 1e6, T * launches / worst);
    }
    return 0;
}
