// fbus_ekf.hip -- kernels, launchers and the C ABI (include/fbus_ekf.h) of the
// MI355X-native batched error-state EKF.  gfx950 only; no CPU fallback.
//
// HBM layout of the filter records ("64-filter tiles of 16-byte chunks"): a record is
// NRECP elements of T = NCH chunks of 16 bytes.  Filters are grouped in tiles of 64
// (one wave); a tile is NCH consecutive 1 KiB pieces and piece c holds chunk c of the
// tile's 64 filters, lane-major.  Chunk c of filter b is therefore at byte offset
// ((b / 64) * NCH + c) * 1024 + (b % 64) * 16: a wave moves its tile with NCH fully
// coalesced 1 KiB buffer_load_dwordx4 / buffer_store_dwordx4 over one contiguous
// NCH KiB region, and a rank's records are one contiguous block for the RCCL gather.
#define FBUS_EKF_NO_ABI_CHECK      // this file DEFINES fbus_ekf_create: no macro here
#include "../../include/fbus_ekf.h"
#include "ekf_kernels.hpp"
#include "ekf_launch.hpp"

#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

using namespace fbus;

namespace {

// ---------------------------------------------------------------------------------
// host-side constants
// ---------------------------------------------------------------------------------
void rotmat_to_quat(const double R[9], double q[4])
{   // trace based, as Eigen's Quaterniond(Matrix3d) (filter.cpp:630, main.cpp marker load)
    double t = R[0] + R[4] + R[8];
    if (t > 0) {
        t = std::sqrt(t + 1.0);
        q[0] = 0.5 * t;
        t = 0.5 / t;
        q[1] = (R[7] - R[5]) * t; q[2] = (R[2] - R[6]) * t; q[3] = (R[3] - R[1]) * t;
    } else {
        int i = 0;
        if (R[4] > R[0]) i = 1;
        if (R[8] > R[4 * i]) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        t = std::sqrt(R[4 * i] - R[4 * j] - R[4 * k] + 1.0);
        q[1 + i] = 0.5 * t;
        t = 0.5 / t;
        q[0] = (R[3 * k + j] - R[3 * j + k]) * t;
        q[1 + j] = (R[3 * j + i] + R[3 * i + j]) * t;
        q[1 + k] = (R[3 * k + i] + R[3 * i + k]) * t;
    }
}

struct HostConst {
    double R_IL[9], P_IL[3], Q_IL[4], CL[16];
    std::vector<double> mk;             // n_markers x MK_STRIDE
    std::vector<double> mkc;            // FBUS_MAX_MARKERS x MKC_STRIDE: corner 0, x axis, y axis of every marker (pixel fold, double)
    std::vector<short> id2slot;
};

bool build_host_const(const fbus_params& prm, HostConst& hc, std::string& err)
{
    // T_IL = diag(-1,-1,1,1) * T_SC_left    FBUS_EKF.m:68 ; filter.hpp:67-70
    double T[16];
    std::memcpy(T, prm.T_SC_left, sizeof(T));
    for (int j = 0; j < 4; ++j) { T[j] = -T[j]; T[4 + j] = -T[4 + j]; }
    double t[3];
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) hc.R_IL[3 * i + j] = T[4 * i + j];
        t[i] = T[4 * i + 3];
    }
    for (int i = 0; i < 3; ++i)         // P_IL = -R_IL' t   MeasureUpdate.m:47 ; filter.cpp:631-632
        hc.P_IL[i] = -(hc.R_IL[i] * t[0] + hc.R_IL[3 + i] * t[1] + hc.R_IL[6 + i] * t[2]);
    rotmat_to_quat(hc.R_IL, hc.Q_IL);
    if (prm.n_markers < 0 || prm.n_markers > FBUS_MAX_MARKERS) { err = "n_markers out of range"; return false; }
    hc.id2slot.assign(FBUS_MAX_MARKER_ID + 1, (short)-1);
    hc.mk.assign((size_t)FBUS_MAX_MARKERS * MK_STRIDE, 0.0);        // always the full table: kernels copy it to LDS whole
    hc.mkc.assign((size_t)FBUS_MAX_MARKERS * MKC_STRIDE, 0.0);
    const double w = hc.Q_IL[0], x = hc.Q_IL[1], y = hc.Q_IL[2], z = hc.Q_IL[3];
    // Lq(Q_IL) * L2, L2 = diag(1,-1,-1,-1)   MeasureUpdate.m:39-44
    const double LL2[16] = { w,  x,  y,  z,
                             x, -w,  z, -y,
                             y, -z, -w,  x,
                             z,  y, -x, -w };
    std::memcpy(hc.CL, LL2, sizeof(LL2));
    for (int k = 0; k < prm.n_markers; ++k) {
        const int id = prm.marker_id[k];
        if (id < 0 || id > FBUS_MAX_MARKER_ID) { err = "marker id out of range"; return false; }
        hc.id2slot[id] = (short)k;
        double* m = &hc.mk[(size_t)k * MK_STRIDE];
        for (int i = 0; i < 3; ++i) m[i] = prm.marker_pos[k][i];
        double q[4];
        rotmat_to_quat(prm.marker_rot[k], q);
        for (int i = 0; i < 4; ++i) m[3 + i] = q[i];
        // c_m = 1/4 |Q_IL|^2 |Qm|^2: the isotropic information of the marker's four quaternion rows per unit weight and |q|^2
        // (PoseFold, ekf_device.hpp)
        m[7] = 0.25 * (w * w + x * x + y * y + z * z) * (q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
        // the marker frame of the corner / pixel rows in double: R_m = rotmat(quat(marker_rot)) as the other kernels (and the
        // oracle) build it -- quaternion_to_rotmat.m:22-33 -- corner k at P_m + R_m c_k, c_k in the marker's x-y plane
        double* c = &hc.mkc[(size_t)k * MKC_STRIDE];
        const double Rm0[3] = { q[0] * q[0] + q[1] * q[1] - q[2] * q[2] - q[3] * q[3], 2 * (q[1] * q[2] + q[0] * q[3]), 2 * (q[1] * q[3] - q[0] * q[2]) };
        const double Rm1[3] = { 2 * (q[1] * q[2] - q[0] * q[3]), q[0] * q[0] - q[1] * q[1] + q[2] * q[2] - q[3] * q[3], 2 * (q[2] * q[3] + q[0] * q[1]) };
        for (int i = 0; i < 3; ++i) { c[i] = prm.marker_pos[k][i]; c[3 + i] = Rm0[i]; c[6 + i] = Rm1[i]; }
    }
    return true;
}

}  // namespace

// ---------------------------------------------------------------------------------
// handle
// ---------------------------------------------------------------------------------
struct fbus_ekf {
    // ---- launch policy, derived from the device at create (fbus_ekf_create), environment overrides read there once ----------
    LaunchPolicy lp;                  // SIMD count, two-wave threshold, vector measurement loads (handed to the launchers)
    int cus = 256;                    // hipDeviceProp::multiProcessorCount (FBUS_FAKE_SIMDS / 4 overrides it)
    size_t l2_bytes = (size_t)4 << 20;        // hipDeviceProp::l2CacheSize (one XCD's L2)
    size_t mall_bytes = (size_t)256 << 20;    // memory-side Infinity Cache: not exposed by the runtime; 256 MiB (MI300X / MI355X), FBUS_MALL_MB
    int policy_batch = 0;             // fbus_ekf_set_policy_batch: the batch the kernel-FAMILY choice is keyed on (0 = this handle's)
    bool records_warm = false;        // the last kernel stored the records with the default cache policy (they sit in L2)
    int big_records_mb = 56;          // records larger than this run the predict with default-policy loads and stores (FBUS_BIG_RECORDS_MB)
    bool warm_after_correct = false;  // experiment knob FBUS_WARM_AFTER_CORRECT=1: the first predict behind a correct takes the "warm" load policy
    int predict_ld = 0;               // record-load policy of the per-call predict: 0 auto (see launch_predict_t), 1 always nt, 2 always default
    int predict_policy_force = -1;    // FBUS_PREDICT_POLICY=0|1|2 (sweeps): nt / nt, default loads + nt stores, default / default
    int B = 0, Bs = 0, device = 0, dtype = 32, N = 18;
    // RCCL communicator of the multi-GPU gather (fbus_ekf_comm_*): one rank per handle
    void* comm = nullptr;
    bool own_comm = false;
    int comm_rank = 0, comm_world = 1;
    // team kernels (several waves per 64-filter tile, ekf_team.hpp): 0 = chosen per launch from the wave count, 1 = never,
    // 2..4 = always with that many roles (fbus_ekf_set_team, FBUS_TEAM_PREDICT / FBUS_TEAM_CORRECT at create)
    int team_predict = 0, team_correct = 0;
    int team_frame = 0;               // FBUS_TEAM_FRAME: 0 = follows team_predict, 1 = never, 2 = always
    int meas_split = -1;              // FBUS_MEAS_SPLIT: -1 auto, 0 never, 2 / 4: always the divided-update pixel kernel with that many waves per tile
    bool no_frame_meas = false;       // FBUS_NO_FRAME_MEAS=1 (A/B runs): fbus_ekf_frame_meas_fused_dev always as predict_n + the per-call update
    fbus_params prm{};
    HostConst hc;
    hipStream_t own_stream = nullptr, stream = nullptr;
    hipEvent_t order_ev = nullptr;    // fbus_ekf_wait_stream / fbus_ekf_signal_stream
    void* recs = nullptr;
    bool own_recs = false;
    size_t rec_bytes = 0, bytes_per_filter = 0;
    void* d_mk = nullptr;
    double* d_mkc = nullptr;            // HostConst::mkc on the device
    short* d_id2slot = nullptr;
    unsigned char* d_applied = nullptr;
    void* d_ema_carry = nullptr;        // B x 6, previous EMA-filtered IMU sample
    bool ema_has_carry = false;
    // staging for the host-pointer entry points (grown on demand)
    void* stage[6] = { nullptr, nullptr, nullptr, nullptr, nullptr, nullptr };
    size_t stage_cap[6] = { 0, 0, 0, 0, 0, 0 };
    // asynchronous host-pointer entry points (fbus_ekf_predict_async / _correct_async): a ring of pinned staging slots + a copy stream
    struct AsyncSlot { void* host = nullptr; void* dev = nullptr; size_t cap = 0; hipEvent_t copied = nullptr, done = nullptr; bool busy = false; };
    static constexpr int ASYNC_SLOTS = 8;
    AsyncSlot aring[ASYNC_SLOTS];
    unsigned anext = 0;
    hipStream_t copy_stream = nullptr;
    int64_t async_calls = 0, async_waits = 0, async_direct = 0;      // calls, calls that had to wait for a slot, pieces DMA'd in place
    std::string err;
    // timing
    bool timing = false;
    struct EvPair { hipEvent_t a, b; int kind; int count; };
    bool timing_suspended = false;   // frame_dev brackets its run of predicts with ONE pair
    int timing_stride = 1;           // frame_dev: bracket every stride-th frame only
    bool capturing = false;          // between graph_begin and graph_end: no events, no host syncs
    std::vector<hipGraphExec_t> graphs;
    int64_t frame_count = 0;
    std::vector<EvPair> ev_pool;
    size_t ev_used = 0;
    double t_ms[FBUS_KERNEL_COUNT] = { 0, 0, 0, 0, 0, 0 };
    int64_t t_n[FBUS_KERNEL_COUNT] = { 0, 0, 0, 0, 0, 0 };
};

namespace {

// Every entry point runs with the handle's device current (a process may hold handles on several GPUs) and
// restores the caller's device on return.
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    explicit DeviceGuard(const fbus_ekf* h) { if (h) enter(h->device); }
    explicit DeviceGuard(int device) { enter(device); }
    void enter(int device)
    {
        if (hipGetDevice(&prev) == hipSuccess && prev != device) switched = (hipSetDevice(device) == hipSuccess);
    }
    ~DeviceGuard() { if (switched) (void)hipSetDevice(prev); }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};

int fail(fbus_ekf_t h, int code, const std::string& msg)
{
    if (h) h->err = msg;
    return code;
}

#define HIP_TRY(h, call)                                                                        \
    do {                                                                                        \
        hipError_t e_ = (call);                                                                 \
        if (e_ != hipSuccess)                                                                   \
            return fail((h), FBUS_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
    } while (0)

size_t esize(const fbus_ekf* h) { return h->dtype == 32 ? 4 : 8; }

template <typename T>
DevConst<T> make_dc(const fbus_ekf* h)
{
    DevConst<T> dc;
    for (int i = 0; i < 4; ++i) dc.qd[i] = (T)h->prm.q_diag[i];
    dc.r_pos = (T)h->prm.r_pos;
    dc.r_quat = (T)h->prm.r_quat;
    for (int i = 0; i < 9; ++i) dc.R_IL[i] = (T)h->hc.R_IL[i];
    for (int i = 0; i < 3; ++i) dc.P_IL[i] = (T)h->hc.P_IL[i];
    for (int i = 0; i < 4; ++i) dc.Q_IL[i] = (T)h->hc.Q_IL[i];
    for (int i = 0; i < 16; ++i) dc.CL[i] = (T)h->hc.CL[i];
    dc.switch_thres = (T)h->prm.switch_thres;
    dc.cov_form = h->prm.cov_form;
    dc.mk = (const T*)h->d_mk;
    dc.id2slot = h->d_id2slot;
    return dc;
}

int flush_events(fbus_ekf_t h)
{
    if (h->ev_used == 0) return FBUS_OK;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    for (size_t i = 0; i < h->ev_used; ++i) {
        float ms = 0.f;
        HIP_TRY(h, hipEventElapsedTime(&ms, h->ev_pool[i].a, h->ev_pool[i].b));
        h->t_ms[h->ev_pool[i].kind] += ms;
        h->t_n[h->ev_pool[i].kind] += h->ev_pool[i].count;
    }
    h->ev_used = 0;
    return FBUS_OK;
}

// returns the index of the event pair to close after the launch, or -1
int timing_begin(fbus_ekf_t h, int kind, int count = 1)
{
    if (!h->timing || h->timing_suspended || h->capturing) return -1;
    if (h->ev_used == h->ev_pool.size()) {
        if (h->ev_pool.size() >= 8192) {
            if (flush_events(h) != FBUS_OK) return -1;
        } else {
            fbus_ekf::EvPair p;
            if (hipEventCreate(&p.a) != hipSuccess || hipEventCreate(&p.b) != hipSuccess) return -1;
            h->ev_pool.push_back(p);
        }
    }
    const int i = (int)h->ev_used++;
    h->ev_pool[i].kind = kind;
    h->ev_pool[i].count = count;
    (void)hipEventRecord(h->ev_pool[i].a, h->stream);
    return i;
}

void timing_end(fbus_ekf_t h, int i)
{
    if (i >= 0) (void)hipEventRecord(h->ev_pool[i].b, h->stream);
}

// How many waves should share one 64-filter tile?  One wave per tile (the lane-per-filter kernels) fills the chip from
// 1024 tiles on; below that the SIMDs that would idle can take a share of every filter's work instead (ekf_team.hpp).
// Measured (rocprofv3 kernel trace, profiles/r03_team_kernels.txt), one-wave -> team:
//   predict    4096 filters 4.52 -> 4.00 us (3 roles), 16 384: 4.80 -> 4.52, 32 768: 6.4 -> 8.8 (the roles' overlapping loads cost
//              more than the shorter instruction streams save once the launch moves 47 MB)            => up to 256 tiles
//   predict_n  K = 8: 4096 filters 18.8 -> 10.3 us, 16 384: 20.7 -> 17.5, 32 768: 23.2 -> 20.9                 => up to 512 tiles
//   correct    4096 filters 6.4 -> 7.6 us, 16 384: 7.2 -> 9.2, 32 768: 10.0 -> 18: the one-wave kernel folds its markers under
//              the load latency and the team pays two exchanges and a redundant 6 x 6 solve per role       => never by default
// (round 4) The thresholds are fractions of the device's SIMD count (256 / 512 tiles = a quarter / half of MI355X's 1024 SIMDs: what
// was measured is "how much of the chip a one-wave launch leaves idle"), and the batch they are compared with is the POLICY batch:
// the handle's own unless fbus_ekf_set_policy_batch names the whole job -- team and one-wave kernels agree to fp32 rounding only,
// so a job cut into shards (fbus::ShardedFilter) keys the choice on the total and gets the same kernels whatever the shard layout.
int policy_tiles(const fbus_ekf* h) { return ((h->policy_batch > 0 ? h->policy_batch : h->B) + 63) / 64; }
int quarter_chip(const fbus_ekf* h) { return h->lp.simds / 4; }
int half_chip(const fbus_ekf* h) { return h->lp.simds / 2; }
int team_roles_predict(const fbus_ekf* h, int K)
{
    if (h->dtype != 32 || h->team_predict == 1) return 1;
    if (h->team_predict >= 2) return K > 1 ? 4 : (h->team_predict > 4 ? 4 : h->team_predict);
    const int tiles = policy_tiles(h);
    if (K > 1) return tiles <= half_chip(h) ? 4 : 1;
    return tiles <= quarter_chip(h) ? 3 : 1;
}
// correct from stereo corners (stacked mode) / from corner pixels (ekf_meas.hpp: the markers of a filter divided among the roles; these
// kernels are bound by the VALU work per marker).  fbus_ekf_set_team's correct_roles: 1 = never, 2 = two roles, 3..4 = four;
// 0 = four up to a quarter of the chip, two up to half.  Both record types.
int team_roles_pixels(const fbus_ekf* h, int M)
{
    if (M < 2 || h->team_correct == 1) return 1;
    if (h->team_correct >= 2) return h->team_correct >= 3 ? 4 : 2;
    const int tiles = policy_tiles(h);
    return tiles <= quarter_chip(h) ? 4 : (tiles <= half_chip(h) ? 2 : 1);
}
// (round 5) correct_pixels with the UPDATE divided between the waves of a tile as well (ekf_meas_split.hpp: a solver and an updater wave,
// every wave below 256 registers): 0 = not this launch (the one-wave-tail kernel with team_roles_pixels' fold roles), 2 = two waves per
// tile (from a quarter of the chip on, full-chip launches included: two waves per SIMD there), 4 = four (small launches).  fp32
// records and the port square to the camera only; fbus_ekf_set_team's correct_roles = 1 keeps the one-wave kernel.
int meas_split_roles(const fbus_ekf* h, int M)
{
    const double* n = h->prm.port_normal;
    if (h->dtype != 32 || M < 2 || h->team_correct == 1 || h->meas_split == 0) return 0;
    if (!(n[0] == 0.0 && n[1] == 0.0 && n[2] == 1.0)) return 0;
    if (h->meas_split > 0) return h->meas_split;
    if (h->team_correct >= 2) return h->team_correct >= 3 ? 4 : 2;
    const int tiles = policy_tiles(h);
    return tiles <= quarter_chip(h) ? 4 : (tiles <= half_chip(h) ? 2 : 0);
}
// fused frame / frame window (frames_team_kernel: the predict_n pipeline + the one-shot correct divided over the four roles).  Follows the predict
// setting (fbus_ekf_set_team: 1 = never, 2..4 = always); FBUS_TEAM_FRAME=1|2 overrides.  Two workgroups of four waves fit a CU
// (80 KiB of LDS, 250 registers), so the automatic choice ends at 512 tiles (profiles/logs/r03_team_frame.txt: +8 % / +12 % at
// 32 768 filters, 0.8x at 40 960).
bool team_frames(const fbus_ekf* h, int mode)
{
    if (h->dtype != 32 || h->prm.cov_form == FBUS_COV_JOSEPH) return false;
    if (mode != MODE_NEAREST && mode != MODE_STACKED) return false;
    if (h->team_frame == 1 || (h->team_frame == 0 && h->team_predict == 1)) return false;
    if (h->team_frame == 2 || h->team_predict >= 2) return true;
    return policy_tiles(h) <= half_chip(h);
}
// (round 4, measured and NOT kept: a batch of more than one wave per SIMD as launches of one round each.  The per-call kernels run
// 65 536 filters -- 52 MB of records, exactly one wave per SIMD -- at 7.7 TB/s because the records stay cache-resident from launch
// to launch; two such launches over the two halves of 131 072 filters do NOT run at twice 12.2 us (29.1 us against 28.0 us for the
// single launch, 60.7 against 55.8 at 262 144: tools/r4_by_batch.sh, profiles/r04_bench_by_batch.txt) -- what is lost past 65 536
// filters is the residency (56 MB, section 4.1 of DESIGN.md), not the launch shape, and beyond it the kernels stream at the
// 6.3-6.7 TB/s this part copies at.)

template <typename T, int N, int D>
int launch_predict_t(fbus_ekf_t h, int K, const void* accel, const void* gyro, const void* dt, int dt_per_filter)
{
    const int ev = timing_begin(h, K == 1 ? FBUS_KERNEL_PREDICT : FBUS_KERNEL_PREDICT_N);
    // the first predict after a kernel that stored the records with the default cache policy (correct, fused frame)
    // reads them with the default policy too; the others stream them non-temporally (see predict_kernel)
    // ... and larger batches run with the default policy on loads AND stores.  Measured on the final round-2 kernels
    // (profiles/logs/r02_policy_by_size.log, headline nt/nt -> default/default): 65 536 filters (52 MB of records, 1024 waves = one
    // round) 4.91e9 -> 4.39e9; 73 728: +1 %; 81 920: +2.5 %; 98 304: +3 %; 131 072: +7 %; 163 840: +11 %; 196 608: +13 %;
    // 262 144 (210 MB): 65.4 -> 55.7 us per launch.  The non-temporal stream only pays while the whole batch is one round of
    // waves whose records stay in the Infinity Cache between launches; the threshold sits just above that batch.
    // Round 3 sweep (profiles/logs/r03_policy_sweep.txt: fp32 N = 18 / N = 15 and fp64, 32 768 .. 1 048 576 filters, all three
    // policies forced through the environment knobs): the crossover between nt / nt and default / default sits at 52-60 MB of
    // records for 800-byte, 608-byte AND 1600-byte records alike (768 waves of fp64 records are on the default side, 1024 waves
    // of N = 15 records on the nt side): a cache-capacity effect, keyed on bytes, not on the wave count.  Beyond the 256 MiB
    // Infinity Cache the records stream from HBM and non-temporal STORES win again (524 288 filters = 419 MB: 133 -> 116 us
    // with default loads; 1 048 576 = 839 MB: 290 -> 275 us with nt loads as well -- nothing is left to hit).
    const size_t mall = h->mall_bytes;
    const bool big = h->rec_bytes > ((size_t)h->big_records_mb << 20);
    int policy = (big ? 2 : (h->records_warm ? 1 : 0));
    if (big && h->rec_bytes > mall) policy = (h->rec_bytes > 2 * mall) ? 0 : 1;
    if (h->predict_ld == 1) policy = 0;
    if (h->predict_ld == 2) policy = big ? 2 : 1;
    if (h->predict_policy_force >= 0) policy = h->predict_policy_force;
    h->records_warm = false;
    const int roles = team_roles_predict(h, K);
    if constexpr (sizeof(T) == 4) {
        if (roles > 1)
            launch_predict_team_k<T, N, D>(h->stream, (T*)h->recs, h->B, K, roles, policy, (const T*)accel, (const T*)gyro,
                                           (const T*)dt, dt_per_filter ? 1 : 0, make_dc<T>(h));
    }
    if (roles <= 1 || sizeof(T) != 4)
        launch_predict_k<T, N, D>(h->stream, (T*)h->recs, h->B, K, policy, (const T*)accel, (const T*)gyro, (const T*)dt,
                                  dt_per_filter ? 1 : 0, make_dc<T>(h), h->lp);
    timing_end(h, ev);
    HIP_TRY(h, hipGetLastError());
    return FBUS_OK;
}

template <typename T, int N, int D>
int launch_correct_t(fbus_ekf_t h, int M, const int32_t* ids, const void* pos, const void* quat, int mode,
                     const uint8_t* skip)
{
    const int ev = timing_begin(h, FBUS_KERNEL_CORRECT);
    h->records_warm = h->warm_after_correct;   // false: written through (sc1), the next predict streams them like any other
    launch_correct_k<T, N, D>(h->stream, (T*)h->recs, h->B, M, (const int*)ids, (const T*)pos, (const T*)quat, mode,
                                  h->prm.cov_form == FBUS_COV_JOSEPH, (const unsigned char*)skip, h->d_applied, make_dc<T>(h), h->lp);
    timing_end(h, ev);
    HIP_TRY(h, hipGetLastError());
    return FBUS_OK;
}

#define DISPATCH(h, FN, ...)                                                                     \
    do {                                                                                         \
        const int key_ = ((h)->dtype == 64 ? 4 : 0) | ((h)->N == 15 ? 2 : 0) | ((h)->prm.dialect == FBUS_DIALECT_CPP ? 1 : 0); \
        switch (key_) {                                                                          \
            case 0: return FN<float, 18, DIALECT_MATLAB>(__VA_ARGS__);                           \
            case 1: return FN<float, 18, DIALECT_CPP>(__VA_ARGS__);                              \
            case 2: return FN<float, 15, DIALECT_MATLAB>(__VA_ARGS__);                           \
            case 3: return FN<float, 15, DIALECT_CPP>(__VA_ARGS__);                              \
            case 4: return FN<double, 18, DIALECT_MATLAB>(__VA_ARGS__);                          \
            case 5: return FN<double, 18, DIALECT_CPP>(__VA_ARGS__);                             \
            case 6: return FN<double, 15, DIALECT_MATLAB>(__VA_ARGS__);                          \
            default: return FN<double, 15, DIALECT_CPP>(__VA_ARGS__);                            \
        }                                                                                        \
    } while (0)

int launch_predict(fbus_ekf_t h, int K, const void* a, const void* g, const void* dt, int per)
{
    DISPATCH(h, launch_predict_t, h, K, a, g, dt, per);
}

int launch_correct(fbus_ekf_t h, int M, const int32_t* ids, const void* pos, const void* quat, int mode,
                   const uint8_t* skip)
{
    DISPATCH(h, launch_correct_t, h, M, ids, pos, quat, mode, skip);
}

template <typename T, int N, int D>
int launch_frame_t(fbus_ekf_t h, int K, const void* accel, const void* gyro, const void* dt, int dt_per_filter, int M,
                   const int32_t* ids, const void* pos, const void* quat, int mode, const uint8_t* skip)
{
    const bool f64_fused = sizeof(T) == 8 && mode == MODE_STACKED && h->prm.cov_form != FBUS_COV_JOSEPH && K > 0 && K <= 255;
    if ((sizeof(T) == 8 && !f64_fused) || (h->prm.cov_form == FBUS_COV_JOSEPH && mode != MODE_STACKED)) {
        // no fused kernel for fp64 outside (stacked, simple) and none for the Joseph form with the reference mode's 7 row-by-row
        // updates (it spilled): those frames are one predict_n launch and one correct launch -- the same arithmetic
        int rc = FBUS_OK;
        if (K > 0) rc = launch_predict_t<T, N, D>(h, K, accel, gyro, dt, dt_per_filter);
        if (rc == FBUS_OK && M > 0) rc = launch_correct_t<T, N, D>(h, M, ids, pos, quat, mode, skip);
        return rc;
    }
    if constexpr (sizeof(T) == 8) {
        // (round 4) fp64, stacked, simple form: ONE launch per camera frame -- the parked K-step predict loop + the row-split passes
        // with the record resident in registers / LDS (frame2_kernel<double>)
        const int ev = timing_begin(h, FBUS_KERNEL_FRAME);
        h->records_warm = true;
        launch_frame_k<T, N, D>(h->stream, (T*)h->recs, h->B, K, (const T*)accel, (const T*)gyro, (const T*)dt,
                                dt_per_filter ? 1 : 0, M, (const int*)ids, (const T*)pos, (const T*)quat, mode, false,
                                (const unsigned char*)skip, h->d_applied, make_dc<T>(h), h->lp);
        timing_end(h, ev);
        HIP_TRY(h, hipGetLastError());
        return FBUS_OK;
    }
    if constexpr (sizeof(T) == 4) {
    const int ev = timing_begin(h, FBUS_KERNEL_FRAME);
    h->records_warm = true;
    if (team_frames(h, mode) && K <= 255) {
        const unsigned char kc1 = (unsigned char)K;
        launch_frames_team_k<T, N, D>(h->stream, (T*)h->recs, h->B, 1, &kc1, (const T*)accel, (const T*)gyro, (const T*)dt,
                                      dt_per_filter ? 1 : 0, M, (const int*)ids, (const T*)pos, (const T*)quat, mode,
                                      (const unsigned char*)skip, h->d_applied, make_dc<T>(h));
    } else
    launch_frame_k<T, N, D>(h->stream, (T*)h->recs, h->B, K, (const T*)accel, (const T*)gyro, (const T*)dt,
                            dt_per_filter ? 1 : 0, M, (const int*)ids, (const T*)pos, (const T*)quat, mode,
                            h->prm.cov_form == FBUS_COV_JOSEPH, (const unsigned char*)skip, h->d_applied, make_dc<T>(h), h->lp);
    timing_end(h, ev);
    HIP_TRY(h, hipGetLastError());
    }
    return FBUS_OK;
}

int launch_frame(fbus_ekf_t h, int K, const void* accel, const void* gyro, const void* dt, int per, int M,
                 const int32_t* ids, const void* pos, const void* quat, int mode, const uint8_t* skip)
{
    DISPATCH(h, launch_frame_t, h, K, accel, gyro, dt, per, M, ids, pos, quat, mode, skip);
}

template <typename T, int N, int D>
int launch_frames_t(fbus_ekf_t h, int F, const unsigned char* kc, const void* accel, const void* gyro, const void* dt,
                    int dt_per_filter, int M, const int32_t* ids, const void* pos, const void* quat, int mode, const uint8_t* skip)
{
    if constexpr (sizeof(T) == 4) {
        const int ev = timing_begin(h, FBUS_KERNEL_FRAME, F);
        h->records_warm = true;
        if (team_frames(h, mode))
            launch_frames_team_k<T, N, D>(h->stream, (T*)h->recs, h->B, F, kc, (const T*)accel, (const T*)gyro, (const T*)dt,
                                          dt_per_filter ? 1 : 0, M, (const int*)ids, (const T*)pos, (const T*)quat, mode,
                                          (const unsigned char*)skip, h->d_applied, make_dc<T>(h));
        else
        launch_frames_k<T, N, D>(h->stream, (T*)h->recs, h->B, F, kc, (const T*)accel, (const T*)gyro, (const T*)dt,
                                 dt_per_filter ? 1 : 0, M, (const int*)ids, (const T*)pos, (const T*)quat, mode,
                                 h->prm.cov_form == FBUS_COV_JOSEPH, (const unsigned char*)skip, h->d_applied, make_dc<T>(h));
        timing_end(h, ev);
        HIP_TRY(h, hipGetLastError());
    }
    return FBUS_OK;
}

int launch_frames(fbus_ekf_t h, int F, const unsigned char* kc, const void* a, const void* g, const void* dt, int per, int M,
                  const int32_t* ids, const void* pos, const void* quat, int mode, const uint8_t* skip)
{
    DISPATCH(h, launch_frames_t, h, F, kc, a, g, dt, per, M, ids, pos, quat, mode, skip);
}


template <typename T, int N>
int pack_t(fbus_ekf_t h, const void* nom, const void* rot, const void* P, const int32_t* prev)
{
    const int grid = (h->B + BLOCK - 1) / BLOCK;             // one wave per 64-filter tile
    hipLaunchKernelGGL((pack_kernel<T, N>), dim3(grid), dim3(BLOCK), 0, h->stream, (T*)h->recs, h->B,
                       (const T*)nom, (const T*)rot, (const T*)P, (const int*)prev);
    HIP_TRY(h, hipGetLastError());
    return FBUS_OK;
}

template <typename T, int N>
int unpack_t(fbus_ekf_t h, void* nom, void* rot, void* P, int32_t* prev)
{
    const int grid = (h->B + BLOCK - 1) / BLOCK;             // one wave per 64-filter tile
    hipLaunchKernelGGL((unpack_kernel<T, N>), dim3(grid), dim3(BLOCK), 0, h->stream, (const T*)h->recs, h->B,
                       (T*)nom, (T*)rot, (T*)P, (int*)prev);
    HIP_TRY(h, hipGetLastError());
    return FBUS_OK;
}

template <typename T, int N>
int reset_cov_t(fbus_ekf_t h)
{
    const int grid = (h->B + BLOCK - 1) / BLOCK;
    const double* d = h->prm.p0_diag;
    hipLaunchKernelGGL((reset_cov_kernel<T, N>), dim3(grid), dim3(BLOCK), 0, h->stream, (T*)h->recs, h->B,
                       (T)d[0], (T)d[1], (T)d[2], (T)d[3], (T)d[4], (T)d[5]);
    HIP_TRY(h, hipGetLastError());
    return FBUS_OK;
}

#define DISPATCH2(h, FN, ...)                                                        \
    do {                                                                             \
        if ((h)->dtype == 32) {                                                      \
            if ((h)->N == 18) return FN<float, 18>(__VA_ARGS__);                     \
            return FN<float, 15>(__VA_ARGS__);                                       \
        }                                                                            \
        if ((h)->N == 18) return FN<double, 18>(__VA_ARGS__);                        \
        return FN<double, 15>(__VA_ARGS__);                                          \
    } while (0)

int do_pack(fbus_ekf_t h, const void* n, const void* r, const void* P, const int32_t* pv) { DISPATCH2(h, pack_t, h, n, r, P, pv); }
int do_unpack(fbus_ekf_t h, void* n, void* r, void* P, int32_t* pv) { DISPATCH2(h, unpack_t, h, n, r, P, pv); }
int do_reset_cov(fbus_ekf_t h) { DISPATCH2(h, reset_cov_t, h); }

template <typename T>
VisConst<T> make_vc(const fbus_ekf* h)
{
    const fbus_params& p = h->prm;
    double RL[9], RR[9], PL[3], PR[3];
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) { RL[3 * i + j] = p.T_SC_left[4 * i + j]; RR[3 * i + j] = p.T_SC_right[4 * i + j]; }
        PL[i] = p.T_SC_left[4 * i + 3]; PR[i] = p.T_SC_right[4 * i + 3];
    }
    VisConst<T> vc;
    double Rrl[9], Rlr[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double a = 0, b = 0;
            for (int k = 0; k < 3; ++k) { a += RL[3 * i + k] * RR[3 * j + k]; b += RR[3 * i + k] * RL[3 * j + k]; }
            Rrl[3 * i + j] = a; Rlr[3 * i + j] = b;
        }
    for (int i = 0; i < 3; ++i) {
        double a = 0, b = 0;
        for (int k = 0; k < 3; ++k) { a += Rrl[3 * i + k] * PR[k]; b += Rlr[3 * i + k] * PR[k]; }
        vc.P_LR[i] = (T)(PL[i] - a);
        vc.t_LRn[i] = (T)(PL[i] - b);
    }
    for (int i = 0; i < 9; ++i) { vc.R_RL[i] = (T)Rrl[i]; vc.R_LRn[i] = (T)Rlr[i]; }
    {   // exact inverse of R_RL (adjugate / determinant), for the forward projection into the right camera
        const double* m = Rrl;
        const double det = m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
        const double inv[9] = { (m[4] * m[8] - m[5] * m[7]), (m[2] * m[7] - m[1] * m[8]), (m[1] * m[5] - m[2] * m[4]),
                                (m[5] * m[6] - m[3] * m[8]), (m[0] * m[8] - m[2] * m[6]), (m[2] * m[3] - m[0] * m[5]),
                                (m[3] * m[7] - m[4] * m[6]), (m[1] * m[6] - m[0] * m[7]), (m[0] * m[4] - m[1] * m[3]) };
        for (int i = 0; i < 9; ++i) vc.R_RL_inv[i] = (T)(inv[i] / det);
    }
    vc.alpha0 = (T)(p.n_air / p.n_glass);
    vc.alpha1 = (T)(p.n_glass / p.n_water);
    vc.sqrt_minus0 = p.n_air < p.n_glass;       // vision.cpp:513
    vc.sqrt_minus1 = p.n_glass > p.n_water;     // vision.cpp:532
    vc.d_air = (T)p.d_air; vc.d_glass = (T)p.d_glass;
    for (int i = 0; i < 3; ++i) vc.nrm[i] = (T)p.port_normal[i];
    {
        const double a = (p.n_air / p.n_glass) * (p.n_glass / p.n_water);
        for (int i = 0; i < 3; ++i) {
            vc.tri[2 * i] = (T)(a * Rrl[3 * i]); vc.tri[2 * i + 1] = (T)(a * Rrl[3 * i + 1]);
            vc.tri[6 + i] = (T)(Rrl[3 * i + 2] * (p.d_air + p.d_glass) + (double)vc.P_LR[i]);
        }
        vc.tri[9] = (T)(p.d_air / a); vc.tri[10] = (T)(p.d_glass * (p.n_air / p.n_glass) / a); vc.tri[11] = (T)a;
    }
    return vc;
}

// constants of the round-4 pixel fold (double whatever the record type)
MeasConst make_mc(const fbus_ekf* h)
{
    const fbus_params& p = h->prm;
    const VisConst<double> vc = make_vc<double>(h);
    MeasConst mc;
    for (int i = 0; i < 3; ++i) mc.P_IL[i] = h->hc.P_IL[i];
    // XL = F R_IL t_I (the triangulation's axis flip undone, vision.cpp:597-599); XR = R_RL^-1 (XL - P_LR) (vision.cpp:555-556)
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) mc.McL[3 * i + j] = (i < 2 ? -1.0 : 1.0) * h->hc.R_IL[3 * i + j];
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) {
            double a = 0;
            for (int k = 0; k < 3; ++k) a += vc.R_RL_inv[3 * i + k] * mc.McL[3 * k + j];
            mc.McR[3 * i + j] = a;
        }
        double b = 0;
        for (int k = 0; k < 3; ++k) b += vc.R_RL_inv[3 * i + k] * vc.P_LR[k];
        mc.tR[i] = -b;
    }
    for (int j = 0; j < 3; ++j) {
        mc.n[j] = p.port_normal[j];
        mc.nML[j] = mc.nMR[j] = 0;
    }
    for (int j = 0; j < 3; ++j)
        for (int i = 0; i < 3; ++i) { mc.nML[j] += mc.n[i] * mc.McL[3 * i + j]; mc.nMR[j] += mc.n[i] * mc.McR[3 * i + j]; }
    {   // R_IL' R_IL = McL' McL (F is orthogonal)
        const double* Mm = mc.McL;
        int o = 0;
        for (int i = 0; i < 3; ++i)
            for (int j = i; j < 3; ++j) mc.NI[o++] = Mm[i] * Mm[j] + Mm[3 + i] * Mm[3 + j] + Mm[6 + i] * Mm[6 + j];
    }
    {   // adj(McL) (cofactors transposed) and McL^-T = cof(McL) / det
        const double* m = mc.McL;
        const double adj[9] = { (m[4] * m[8] - m[5] * m[7]), (m[2] * m[7] - m[1] * m[8]), (m[1] * m[5] - m[2] * m[4]),
                                (m[5] * m[6] - m[3] * m[8]), (m[0] * m[8] - m[2] * m[6]), (m[2] * m[3] - m[0] * m[5]),
                                (m[3] * m[7] - m[4] * m[6]), (m[1] * m[6] - m[0] * m[7]), (m[0] * m[4] - m[1] * m[3]) };
        const double det = m[0] * adj[0] + m[1] * adj[3] + m[2] * adj[6];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) { mc.adjL[3 * i + j] = adj[3 * i + j]; mc.MiTL[3 * i + j] = adj[3 * j + i] / det; }
    }
    mc.a0 = p.n_air / p.n_glass;
    mc.a1 = p.n_air / p.n_water;
    mc.d_air = p.d_air; mc.d_glass = p.d_glass;
    mc.klim = 0.81 * mc.a1 * mc.a1 / (1.0 - mc.a1 * mc.a1);
    mc.st[0] = (float)mc.a1; mc.st[1] = (float)(mc.a1 * mc.a1); mc.st[2] = (float)(1.0 - mc.a1 * mc.a1); mc.st[3] = (float)(1.0 - mc.a0 * mc.a0);
    mc.st[4] = (float)mc.d_air; mc.st[5] = (float)(mc.d_glass * mc.a0); mc.st[6] = (float)((mc.d_air + mc.d_glass * mc.a0) / mc.a1); mc.st[7] = 0.f;
    mc.mkc = h->d_mkc;
    return mc;
}

template <typename T>
int launch_marker_pose_t(fbus_ekf_t h, int n, int geometry, const void* left, const void* right, void* pos,
                         void* quat, void* corners3d)
{
    const int grid = (n + 255) / 256;
    const int ev = timing_begin(h, FBUS_KERNEL_MARKER_POSE);
    hipLaunchKernelGGL((marker_pose_kernel<T>), dim3(grid), dim3(256), 0, h->stream, n, geometry, (const T*)left,
                       (const T*)right, (T*)pos, (T*)quat, (T*)corners3d, make_vc<double>(h));
    timing_end(h, ev);
    HIP_TRY(h, hipGetLastError());
    return FBUS_OK;
}

template <typename T, int N>
int init_gb_t(fbus_ekf_t h, int Tn, const void* accel, const void* gyro)
{
    hipLaunchKernelGGL((init_gravity_bias_kernel<T, N>), dim3((h->B + 255) / 256), dim3(256), 0, h->stream, (T*)h->recs,
                       h->B, Tn, (const T*)accel, (const T*)gyro);
    HIP_TRY(h, hipGetLastError());
    return FBUS_OK;
}
int do_init_gb(fbus_ekf_t h, int Tn, const void* a, const void* g) { DISPATCH2(h, init_gb_t, h, Tn, a, g); }

template <typename T, int N, int D>
int pose_init_t(fbus_ekf_t h, int M, const int32_t* ids, const void* pos, const void* quat, int what,
                const uint8_t* mask, void* out7)
{
    hipLaunchKernelGGL((pose_init_kernel<T, N, D>), dim3((h->B + 255) / 256), dim3(256), 0, h->stream, (T*)h->recs, h->B,
                       M, (const int*)ids, (const T*)pos, (const T*)quat, what, (T)h->prm.max_dist,
                       (const unsigned char*)mask, (T*)out7, h->d_applied, make_dc<T>(h));
    HIP_TRY(h, hipGetLastError());
    return FBUS_OK;
}
int do_pose_init(fbus_ekf_t h, int M, const int32_t* ids, const void* pos, const void* quat, int what,
                 const uint8_t* mask, void* out7)
{
    DISPATCH(h, pose_init_t, h, M, ids, pos, quat, what, mask, out7);
}

template <typename T, int N, int D>
int launch_correct_corners_t(fbus_ekf_t h, int M, const int32_t* ids, const void* left, const void* right, int geometry,
                             int mode, const uint8_t* skip)
{
    // the kernel fetches a slot's image points with 16-byte loads (a slot is 32 / 48 contiguous bytes): the arrays must start on a
    // 16-byte boundary -- any allocation does; a view offset by one to three elements does not and is refused, not read unaligned
    if (((reinterpret_cast<uintptr_t>(left) | reinterpret_cast<uintptr_t>(right)) & 15) != 0)
        return fail(h, FBUS_ERR_INVALID, "fbus_ekf_correct_corners: left / right must be 16-byte aligned device pointers");
    const int ev = timing_begin(h, FBUS_KERNEL_CORRECT_CORNERS);
    // triangulation and fold in double, non-cancelling update (ekf_meas.hpp); records written through (sc1) as correct_kernel's
    h->records_warm = h->warm_after_correct;
    const int roles = mode == MODE_STACKED ? team_roles_pixels(h, M) : 1;
    launch_corners2_k<T, N, D>(h->stream, (T*)h->recs, h->B, M, (const int*)ids, (const T*)left, (const T*)right, geometry, mode,
                               roles, h->prm.marker_size, h->prm.r_pos, h->prm.switch_thres, (const unsigned char*)skip,
                               h->d_applied, h->d_id2slot, make_mc(h), make_vc<double>(h), make_vc<T>(h));
    timing_end(h, ev);
    HIP_TRY(h, hipGetLastError());
    return FBUS_OK;
}

template <typename T, int N, int D>
int launch_correct_pixels_t(fbus_ekf_t h, int M, const int32_t* ids, const void* left, const void* right, const uint8_t* skip)
{
    if (((reinterpret_cast<uintptr_t>(left) | reinterpret_cast<uintptr_t>(right)) & 15) != 0)      // 16-byte loads, see correct_corners
        return fail(h, FBUS_ERR_INVALID, "fbus_ekf_correct_pixels: left / right must be 16-byte aligned device pointers");
    const int ev = timing_begin(h, FBUS_KERNEL_CORRECT_CORNERS);
    // double-precision fold + non-cancelling update (ekf_meas.hpp), both record types, either covariance form (the form is
    // symmetric by construction and subtracts nothing on the rows the measurement shrinks: what Joseph's form is chosen for)
    h->records_warm = h->warm_after_correct;          // written through (sc1), as correct_kernel's records
    const int split = meas_split_roles(h, M);
    if constexpr (sizeof(T) == 4) {
        if (split > 0)
            launch_pixels_split_k<T, N, D>(h->stream, (T*)h->recs, h->B, M, (const int*)ids, (const T*)left, (const T*)right, split,
                                           h->prm.marker_size, h->prm.r_pix, (const unsigned char*)skip, h->d_applied, h->d_id2slot, make_mc(h));
    }
    if (split == 0 || sizeof(T) != 4) {
        const int roles = team_roles_pixels(h, M);
        launch_pixels2_k<T, N, D>(h->stream, (T*)h->recs, h->B, M, (const int*)ids, (const T*)left, (const T*)right, roles,
                                  h->prm.marker_size, h->prm.r_pix, (const unsigned char*)skip, h->d_applied, h->d_id2slot, make_mc(h));
    }
    timing_end(h, ev);
    HIP_TRY(h, hipGetLastError());
    return FBUS_OK;
}

int launch_correct_pixels(fbus_ekf_t h, int M, const int32_t* ids, const void* left, const void* right, const uint8_t* skip)
{
    DISPATCH(h, launch_correct_pixels_t, h, M, ids, left, right, skip);
}

// One camera frame with the north star's MeasureUpdate: K predicts + correct_pixels (kind 0) / correct_corners (kind 1).
// ONE launch (frame_meas_kernel: record resident, covariance parked in LDS across the fold) where the per-call update would run one wave
// per tile anyway -- fp32 records, more than half a chip of tiles (or fbus_ekf_set_team(., 1)) -- and equal to the per-call
// sequence to fp32 rounding there (bit-equal: its update alone, K = 0, and a window to its frames); otherwise predict_n + the per-call update (whose team forms fill a small launch better than one resident wave
// per tile could; fp64 records: the resident fold + covariance do not fit 512 registers).
// fused: does this handle take the resident kernel (frame_meas_kernel) for M marker slots of this kind / mode?
template <typename T>
bool frame_meas_resident(const fbus_ekf* h, int kind, int M, int mode)
{
    const int roles = (kind == MEAS_CORNERS && mode != MODE_STACKED) ? 1 : team_roles_pixels(h, M);
    return sizeof(T) == 4 && M > 0 && roles == 1 && !h->no_frame_meas;
}
template <typename T, int N, int D>
int launch_frame_meas_t(fbus_ekf_t h, int F, const unsigned char* kc, const void* accel, const void* gyro, const void* dt, int dt_per_filter,
                        int kind, int M, const int32_t* ids, const void* left, const void* right, int geometry, int mode, const uint8_t* skip)
{
    // F = 1: one frame; F > 1: a window (the caller has checked that the resident kernel applies)
    const int K = kc[0];
    if (F == 1 && !frame_meas_resident<T>(h, kind, M, mode)) {
        int rc = FBUS_OK;
        if (K > 0) rc = launch_predict_t<T, N, D>(h, K, accel, gyro, dt, dt_per_filter);
        if (rc == FBUS_OK && M > 0)
            rc = kind == MEAS_PIXELS ? launch_correct_pixels_t<T, N, D>(h, M, ids, left, right, skip)
                                     : launch_correct_corners_t<T, N, D>(h, M, ids, left, right, geometry, mode, skip);
        return rc;
    }
    if constexpr (sizeof(T) == 4) {
        if (((reinterpret_cast<uintptr_t>(left) | reinterpret_cast<uintptr_t>(right)) & 15) != 0)
            return fail(h, FBUS_ERR_INVALID, "fbus_ekf_frame(s)_meas_fused_dev: left / right must be 16-byte aligned device pointers");
        const int ev = timing_begin(h, FBUS_KERNEL_FRAME, F);
        h->records_warm = h->warm_after_correct;      // written through (sc1), as the per-call updates: the next predict streams them
        const DevConst<T> dc = make_dc<T>(h);
        launch_frame_meas_k<T, N, D>(h->stream, (T*)h->recs, h->B, F, kc, (const T*)accel, (const T*)gyro, (const T*)dt, dt_per_filter ? 1 : 0,
                                     kind, M, (const int*)ids, (const T*)left, (const T*)right, geometry, mode, h->prm.marker_size,
                                     kind == MEAS_PIXELS ? h->prm.r_pix : h->prm.r_pos, h->prm.switch_thres, (const unsigned char*)skip,
                                     h->d_applied, h->d_id2slot, make_mc(h), make_vc<double>(h), make_vc<T>(h), dc.qd);
        timing_end(h, ev);
        HIP_TRY(h, hipGetLastError());
    }
    return FBUS_OK;
}
int launch_frame_meas(fbus_ekf_t h, int F, const unsigned char* kc, const void* accel, const void* gyro, const void* dt, int per, int kind,
                      int M, const int32_t* ids, const void* left, const void* right, int geometry, int mode, const uint8_t* skip)
{
    DISPATCH(h, launch_frame_meas_t, h, F, kc, accel, gyro, dt, per, kind, M, ids, left, right, geometry, mode, skip);
}
bool frame_meas_is_resident(const fbus_ekf* h, int kind, int M, int mode)
{
    return h->dtype == 32 ? frame_meas_resident<float>(h, kind, M, mode) : false;
}

int launch_correct_corners(fbus_ekf_t h, int M, const int32_t* ids, const void* left, const void* right, int geometry,
                           int mode, const uint8_t* skip)
{
    DISPATCH(h, launch_correct_corners_t, h, M, ids, left, right, geometry, mode, skip);
}

int ensure_stage(fbus_ekf_t h, int slot, size_t bytes)
{
    if (bytes <= h->stage_cap[slot]) return FBUS_OK;
    if (h->stage[slot]) {
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        HIP_TRY(h, hipFree(h->stage[slot]));
        h->stage[slot] = nullptr;
        h->stage_cap[slot] = 0;
    }
    HIP_TRY(h, hipMalloc(&h->stage[slot], bytes));
    h->stage_cap[slot] = bytes;
    return FBUS_OK;
}

// copies a host array into staging slot `slot`; returns the device pointer through out
int stage_in(fbus_ekf_t h, int slot, const void* host, size_t bytes, const void** out)
{
    *out = nullptr;
    if (!host || bytes == 0) return FBUS_OK;
    int rc = ensure_stage(h, slot, bytes);
    if (rc != FBUS_OK) return rc;
    HIP_TRY(h, hipMemcpyAsync(h->stage[slot], host, bytes, hipMemcpyHostToDevice, h->stream));
    *out = h->stage[slot];
    return FBUS_OK;
}

// ---- asynchronous host-pointer calls ---------------------------------------------------------------------------------
// The reference's caller hands one IMU sample at a time to a filter thread and returns at once (FILTER::SetImuData under a mutex,
// filter.cpp:24-55; BatchImuProcessing issues one predict per sample, :505-516).  The synchronous host-pointer entry points stage
// pageable memory and wait for the kernel (103 us per predict at 65 536 filters); these do not wait:
//   * every input array is taken BY VALUE at the call: pageable memory is copied into a pinned ring slot by the calling thread
//     (the caller's buffer is free again on return), pinned memory (hipHostMalloc / hipHostRegister / fbus_ekf_host_register) is
//     DMA'd in place (it must stay unchanged until fbus_ekf_async_inputs_consumed / fbus_ekf_sync);
//   * the H2D copy runs on a copy stream of the handle's, the kernel on the handle's stream behind an event: the copy of call i + 1
//     overlaps the kernel of call i;
//   * a slot is reused after ASYNC_SLOTS calls; the call then waits for THAT slot's kernel only (back-pressure, counted).
// Completion and device-side errors: fbus_ekf_sync, or any host-pointer result (fbus_ekf_get_state, fbus_ekf_get_applied).
struct AsyncPiece { const void* src; size_t bytes; const void** dev_out; };

bool host_ptr_is_pinned(const void* p)
{
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return false; }   // (an ordinary malloc'ed pointer)
    return at.type == hipMemoryTypeHost;
}

int async_begin(fbus_ekf_t h, AsyncPiece* pieces, int n, fbus_ekf::AsyncSlot** out)
{
    if (h->capturing) return fail(h, FBUS_ERR_INVALID, "the asynchronous host-pointer calls cannot be captured into a graph (use the _dev entry points)");
    if (!h->copy_stream) HIP_TRY(h, hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking));
    fbus_ekf::AsyncSlot& a = h->aring[h->anext++ % fbus_ekf::ASYNC_SLOTS];
    if (!a.copied) {
        HIP_TRY(h, hipEventCreateWithFlags(&a.copied, hipEventDisableTiming));
        HIP_TRY(h, hipEventCreateWithFlags(&a.done, hipEventDisableTiming));
    }
    ++h->async_calls;
    if (a.busy) {                                   // the kernel that read this slot ASYNC_SLOTS calls ago
        if (hipEventQuery(a.done) != hipSuccess) { ++h->async_waits; HIP_TRY(h, hipEventSynchronize(a.done)); }
        a.busy = false;
    }
    size_t total = 0;
    for (int i = 0; i < n; ++i) total += (pieces[i].bytes + 255) & ~(size_t)255;
    if (total > a.cap) {
        if (a.host) HIP_TRY(h, hipHostFree(a.host));
        if (a.dev) HIP_TRY(h, hipFree(a.dev));
        a.host = a.dev = nullptr; a.cap = 0;
        const size_t cap = (total + (total >> 2) + 65535) & ~(size_t)65535;
        HIP_TRY(h, hipHostMalloc(&a.host, cap, hipHostMallocDefault));
        HIP_TRY(h, hipMalloc(&a.dev, cap));
        a.cap = cap;
    }
    // staged pieces first (one contiguous range -> ONE copy), then the pinned ones in place
    size_t off = 0, staged_end = 0;
    bool direct[8] = { false, false, false, false, false, false, false, false };
    for (int i = 0; i < n; ++i) {
        *pieces[i].dev_out = nullptr;
        if (!pieces[i].src || pieces[i].bytes == 0) continue;
        direct[i] = pieces[i].bytes >= 4096 && host_ptr_is_pinned(pieces[i].src);
        if (direct[i]) continue;
        std::memcpy((char*)a.host + off, pieces[i].src, pieces[i].bytes);
        *pieces[i].dev_out = (char*)a.dev + off;
        off += (pieces[i].bytes + 255) & ~(size_t)255;
        staged_end = off;
    }
    if (staged_end) HIP_TRY(h, hipMemcpyAsync(a.dev, a.host, staged_end, hipMemcpyHostToDevice, h->copy_stream));
    for (int i = 0; i < n; ++i) {
        if (!direct[i]) continue;
        HIP_TRY(h, hipMemcpyAsync((char*)a.dev + off, pieces[i].src, pieces[i].bytes, hipMemcpyHostToDevice, h->copy_stream));
        *pieces[i].dev_out = (char*)a.dev + off;
        off += (pieces[i].bytes + 255) & ~(size_t)255;
        ++h->async_direct;
    }
    HIP_TRY(h, hipEventRecord(a.copied, h->copy_stream));
    HIP_TRY(h, hipStreamWaitEvent(h->stream, a.copied, 0));
    *out = &a;
    return FBUS_OK;
}

int async_end(fbus_ekf_t h, fbus_ekf::AsyncSlot* a, int rc)
{
    // (also behind a failed launch: the slot's memory must not be rewritten while the stream may still read it)
    HIP_TRY(h, hipEventRecord(a->done, h->stream));
    a->busy = true;
    return rc;
}

size_t record_elems(int dtype, int N)
{
    if (dtype == 32) return N == 18 ? Rec<float, 18>::NRECP : Rec<float, 15>::NRECP;
    return N == 18 ? Rec<double, 18>::NRECP : Rec<double, 15>::NRECP;
}

}  // namespace

// ---------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------
extern "C" {

const char* fbus_status_string(int s)
{
    switch (s) {
        case FBUS_OK: return "ok";
        case FBUS_ERR_INVALID: return "invalid argument";
        case FBUS_ERR_NO_DEVICE: return "no usable HIP device (this library has no CPU fallback)";
        case FBUS_ERR_HIP: return "HIP runtime error";
        case FBUS_ERR_UNSUPPORTED: return "unsupported dtype/nstate/mode";
        case FBUS_ERR_NOMEM: return "out of memory";
        case FBUS_ERR_ABI: return "caller and library were built against different versions of fbus_ekf.h";
        default: return "unknown status";
    }
}

int fbus_params_default(fbus_params* prm, int dialect)
{
    if (!prm || (dialect != FBUS_DIALECT_MATLAB && dialect != FBUS_DIALECT_CPP)) return FBUS_ERR_INVALID;
    std::memset(prm, 0, sizeof(*prm));
    prm->dialect = dialect;
    prm->cov_form = FBUS_COV_SIMPLE;
    // FBUS_EKF.m:36-39,103-106 ; paramconfig.yml:46-49 via filter.hpp:108-115
    prm->q_diag[0] = 1e-3; prm->q_diag[1] = 1e-4; prm->q_diag[2] = 1e-3; prm->q_diag[3] = 1e-4;
    if (dialect == FBUS_DIALECT_MATLAB) {
        prm->r_pos = 0.01; prm->r_quat = 0.01;                       // FBUS_EKF.m:32-33
        const double d[6] = { 1e-4, 0.1, 1e-4, 1e-3, 1e-3, 100.0 };  // FBUS_EKF.m:88-99
        std::memcpy(prm->p0_diag, d, sizeof(d));
    } else {
        prm->r_pos = 0.001; prm->r_quat = 0.001;                     // paramconfig.yml:53-54
        const double d[6] = { 1e-4, 1e-2, 1e-4, 1e-2, 1e-2, 100.0 }; // filter.hpp:29-34
        std::memcpy(prm->p0_diag, d, sizeof(d));
    }
    // camerainfo1.yml == matlab/config/camerainfo.yml, raw T_SC
    const double TL[16] = { -0.999862, 0.015685, -0.00548, 0.059967,
                            -0.015639, -0.999843, -0.00827, 0.000127837,
                            -0.005609, -0.008183, 0.999951, -0.002,
                            0, 0, 0, 1 };
    const double TR[16] = { -0.999826, 0.00929485, -0.0161445, -0.0601272,
                            -0.00937869, -0.999942, 0.00514829, 0.000124714,
                            -0.0160959, 0.00529897, 0.999857, -0.002,
                            0, 0, 0, 1 };
    std::memcpy(prm->T_SC_left, TL, sizeof(TL));
    std::memcpy(prm->T_SC_right, TR, sizeof(TR));
    // GetMarkerMap.m:1-63 == markersetup.yml
    const double I3[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 };
    const double RA[9] = { 1, 0, 0, 0, 0, -1, 0, 1, 0 };
    const double RB[9] = { 1, 0, 0, 0, -1, 0, 0, 0, -1 };
    const struct { int id; double pos[3]; const double* rot; } map[12] = {
        { 0, { 0, 0, 0 }, I3 },         { 1, { 0, 0.61, 0.285 }, RA },  { 2, { 0, 0.61, 1.185 }, RA },
        { 3, { 0, 0.61, 2.085 }, RA },  { 4, { 0, 0.61, 2.985 }, RA },  { 5, { 0, 0.265, 4.12 }, RB },
        { 6, { 0, -0.635, 4.12 }, RB }, { 7, { 0, -1.535, 4.12 }, RB }, { 8, { 0, -2.435, 4.12 }, RB },
        { 16, { 0, -2.7, 0 }, I3 },     { 17, { 0, -1.8, 0 }, I3 },     { 18, { 0, -0.9, 0 }, I3 } };
    prm->n_markers = 12;
    for (int k = 0; k < 12; ++k) {
        prm->marker_id[k] = map[k].id;
        std::memcpy(prm->marker_pos[k], map[k].pos, sizeof(double) * 3);
        std::memcpy(prm->marker_rot[k], map[k].rot, sizeof(double) * 9);
    }
    prm->switch_thres = 0.5;    // paramconfig.yml:57
    prm->max_dist = 2.0;        // paramconfig.yml:56
    prm->n_air = 1.00; prm->n_glass = 1.49; prm->n_water = 1.32;    // paramconfig.yml:31-42
    prm->d_air = 0.002; prm->d_glass = 0.02;
    prm->port_normal[0] = 0; prm->port_normal[1] = 0; prm->port_normal[2] = 1;
    prm->marker_size = 0.28;    // vision.hpp:114
    prm->r_pix = 1e-6;          // (1e-3)^2 in normalised image coordinates: ~0.4 px at the recordings' focal length
    return FBUS_OK;
}

int fbus_params_validate(const fbus_params* prm, char* msg, size_t msg_len)
{
    // everything fbus_ekf_create derives from the parameters on the HOST (no device needed): the camera constants, the marker
    // table and the id -> slot map.  Also what the CPU sanitizer build exercises (tests/test_sanitizers_cpu.py).
    if (msg && msg_len) msg[0] = 0;
    if (!prm) return FBUS_ERR_INVALID;
    std::string err;
    if (prm->dialect != FBUS_DIALECT_MATLAB && prm->dialect != FBUS_DIALECT_CPP) err = "dialect must be FBUS_DIALECT_MATLAB or FBUS_DIALECT_CPP";
    else if (prm->cov_form != FBUS_COV_SIMPLE && prm->cov_form != FBUS_COV_JOSEPH) err = "cov_form must be FBUS_COV_SIMPLE or FBUS_COV_JOSEPH";
    else if (!(prm->r_pos > 0) || !(prm->r_quat > 0)) err = "measurement noise must be positive";
    else {
        HostConst hc;
        (void)build_host_const(*prm, hc, err);
    }
    if (err.empty()) return FBUS_OK;
    if (msg && msg_len) { std::strncpy(msg, err.c_str(), msg_len - 1); msg[msg_len - 1] = 0; }
    return FBUS_ERR_INVALID;
}

int fbus_ekf_abi_version(void) { return FBUS_ABI_VERSION; }
size_t fbus_params_size(void) { return sizeof(fbus_params); }

int fbus_ekf_create_checked(fbus_ekf_t* out, const fbus_params* prm, size_t params_size, int abi_version, int batch, int device,
                            int dtype, int nstate)
{
    if (out) *out = nullptr;
    // checked BEFORE prm is read: a shorter struct must not be copied past its end
    if (abi_version != FBUS_ABI_VERSION || params_size != sizeof(fbus_params)) return FBUS_ERR_ABI;
    return fbus_ekf_create(out, prm, batch, device, dtype, nstate);
}

int fbus_ekf_create(fbus_ekf_t* out, const fbus_params* prm, int batch, int device, int dtype, int nstate)
{
    if (!out) return FBUS_ERR_INVALID;
    *out = nullptr;
    if (!prm || batch <= 0) return FBUS_ERR_INVALID;
    if ((dtype != 32 && dtype != 64) || (nstate != 15 && nstate != 18)) return FBUS_ERR_UNSUPPORTED;
    // the checks of fbus_params_validate (dialect, cov_form, positive noise -- a zero r_pos would turn the fold's weights into NaN --,
    // marker table): the header promises them here
    if (fbus_params_validate(prm, nullptr, 0) != FBUS_OK) return FBUS_ERR_INVALID;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return FBUS_ERR_NO_DEVICE;
    DeviceGuard guard_(device);
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess || cur != device) return FBUS_ERR_NO_DEVICE;

    fbus_ekf* h = new (std::nothrow) fbus_ekf();
    if (!h) return FBUS_ERR_NOMEM;
    h->B = batch;
    h->Bs = (batch + 63) / 64 * 64;
    {   // launch policy from the device: CU count -> SIMDs, L2 size; the memory-side cache is not exposed (256 MiB assumed)
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess) {
            if (prop.multiProcessorCount > 0) h->cus = prop.multiProcessorCount;
            if (prop.l2CacheSize > 0) h->l2_bytes = (size_t)prop.l2CacheSize;
        }
        int simds = h->cus * 4;
        bool fake = false;
        if (const char* e = std::getenv("FBUS_FAKE_SIMDS")) { const int v = std::atoi(e); if (v >= 4) { simds = v / 4 * 4; h->cus = simds / 4; fake = true; } }
        h->lp.simds = simds;
        h->lp.two_wave_min_b = simds * 64 + 1;
        if (const char* e = std::getenv("FBUS_TWO_WAVE_MIN_B")) h->lp.two_wave_min_b = std::atoi(e);
        h->lp.meas_vec = std::getenv("FBUS_NO_MEAS_VEC") == nullptr;
        // 256 MiB is MI355X's (and MI300X's) Infinity Cache whatever the CU count of the SKU: not scaled with the device; only the
        // test knob FBUS_FAKE_SIMDS (a pretended smaller chip) scales it down with the SIMD count, FBUS_MALL_MB sets it outright
        h->mall_bytes = (size_t)256 << 20;
        if (fake && simds < 1024) h->mall_bytes = ((size_t)256 << 20) / 1024 * (size_t)simds;
        if (const char* e = std::getenv("FBUS_MALL_MB")) { const long v = std::atol(e); if (v > 0) h->mall_bytes = (size_t)v << 20; }
        // records larger than this leave the one-round, cache-resident regime (measured crossover 52-60 MB on 1024 SIMDs, 4.1)
        // (a cache-capacity effect: the same 56 MB on any part with a 256 MiB Infinity Cache; scaled only for a pretended chip)
        h->big_records_mb = (fake && simds < 1024) ? (int)(56L * simds / 1024) : 56;
    }
    if (const char* e = std::getenv("FBUS_BIG_RECORDS_MB")) h->big_records_mb = std::atoi(e);
    if (const char* e = std::getenv("FBUS_WARM_AFTER_CORRECT")) h->warm_after_correct = std::atoi(e) != 0;
    if (const char* e = std::getenv("FBUS_PREDICT_POLICY")) { const int v = std::atoi(e); if (v >= 0 && v <= 2) h->predict_policy_force = v; }
    if (const char* e = std::getenv("FBUS_TEAM_PREDICT")) { const int v = std::atoi(e); if (v >= 0 && v <= 4) h->team_predict = v; }
    if (const char* e = std::getenv("FBUS_TEAM_CORRECT")) { const int v = std::atoi(e); if (v >= 0 && v <= 4) h->team_correct = v; }
    if (const char* e = std::getenv("FBUS_NO_FRAME_MEAS")) h->no_frame_meas = std::atoi(e) != 0;
    if (const char* e = std::getenv("FBUS_MEAS_SPLIT")) { const int v = std::atoi(e); if (v == 0 || v == 2 || v == 4) h->meas_split = v; }
    if (const char* e = std::getenv("FBUS_TEAM_FRAME")) { const int v = std::atoi(e); if (v >= 0 && v <= 2) h->team_frame = v; }
    if (const char* e = std::getenv("FBUS_PREDICT_LD"))          // experiment knob: nt | default | auto
        h->predict_ld = !std::strcmp(e, "nt") ? 1 : (!std::strcmp(e, "default") ? 2 : 0);
    h->device = device;
    h->dtype = dtype;
    h->N = nstate;
    h->prm = *prm;
    std::string err;
    if (!build_host_const(h->prm, h->hc, err)) { delete h; return FBUS_ERR_INVALID; }

    auto bail = [&](int code) { fbus_ekf_destroy(h); return code; };
    if (hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking) != hipSuccess) return bail(FBUS_ERR_HIP);
    h->stream = h->own_stream;
    h->bytes_per_filter = record_elems(dtype, nstate) * esize(h);
    h->rec_bytes = h->bytes_per_filter * (size_t)h->Bs;
    if (hipMalloc(&h->recs, h->rec_bytes) != hipSuccess) return bail(FBUS_ERR_NOMEM);
    h->own_recs = true;
    if (hipMemsetAsync(h->recs, 0, h->rec_bytes, h->stream) != hipSuccess) return bail(FBUS_ERR_HIP);
    if (hipMalloc((void**)&h->d_applied, (size_t)h->Bs) != hipSuccess) return bail(FBUS_ERR_NOMEM);
    if (hipMemsetAsync(h->d_applied, 0, (size_t)h->Bs, h->stream) != hipSuccess) return bail(FBUS_ERR_HIP);
    // marker table + id lookup
    const size_t nmk = h->hc.mk.size();
    if (hipMalloc(&h->d_mk, nmk * esize(h)) != hipSuccess) return bail(FBUS_ERR_NOMEM);
    if (dtype == 32) {
        std::vector<float> f(h->hc.mk.begin(), h->hc.mk.end());
        if (hipMemcpy(h->d_mk, f.data(), nmk * 4, hipMemcpyHostToDevice) != hipSuccess) return bail(FBUS_ERR_HIP);
    } else {
        if (hipMemcpy(h->d_mk, h->hc.mk.data(), nmk * 8, hipMemcpyHostToDevice) != hipSuccess) return bail(FBUS_ERR_HIP);
    }
    if (hipMalloc((void**)&h->d_mkc, h->hc.mkc.size() * sizeof(double)) != hipSuccess) return bail(FBUS_ERR_NOMEM);
    if (hipMemcpy(h->d_mkc, h->hc.mkc.data(), h->hc.mkc.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) return bail(FBUS_ERR_HIP);
    const size_t lut = h->hc.id2slot.size() * sizeof(short);
    if (hipMalloc((void**)&h->d_id2slot, lut) != hipSuccess) return bail(FBUS_ERR_NOMEM);
    if (hipMemcpy(h->d_id2slot, h->hc.id2slot.data(), lut, hipMemcpyHostToDevice) != hipSuccess) return bail(FBUS_ERR_HIP);
    if (hipStreamSynchronize(h->stream) != hipSuccess) return bail(FBUS_ERR_HIP);
    *out = h;
    return FBUS_OK;
}

int fbus_ekf_destroy(fbus_ekf_t h)
{
    DeviceGuard guard_(h);
    if (!h) return FBUS_OK;
    (void)hipStreamSynchronize(h->stream);
    for (auto& p : h->ev_pool) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
    for (auto g : h->graphs) if (g) (void)hipGraphExecDestroy(g);
    for (int i = 0; i < 6; ++i) if (h->stage[i]) (void)hipFree(h->stage[i]);
    if (h->copy_stream) (void)hipStreamSynchronize(h->copy_stream);
    for (auto& a : h->aring) {
        if (a.host) (void)hipHostFree(a.host);
        if (a.dev) (void)hipFree(a.dev);
        if (a.copied) (void)hipEventDestroy(a.copied);
        if (a.done) (void)hipEventDestroy(a.done);
    }
    if (h->copy_stream) (void)hipStreamDestroy(h->copy_stream);
    if (h->own_recs && h->recs) (void)hipFree(h->recs);
    if (h->d_applied) (void)hipFree(h->d_applied);
    if (h->d_ema_carry) (void)hipFree(h->d_ema_carry);
    if (h->d_mk) (void)hipFree(h->d_mk);
    if (h->d_mkc) (void)hipFree(h->d_mkc);
    if (h->d_id2slot) (void)hipFree(h->d_id2slot);
    if (h->order_ev) (void)hipEventDestroy(h->order_ev);
    (void)fbus_ekf_comm_destroy(h);
    if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
    delete h;
    return FBUS_OK;
}

int fbus_ekf_set_team(fbus_ekf_t h, int predict_roles, int correct_roles)
{
    if (!h || predict_roles < 0 || predict_roles > 4 || correct_roles < 0 || correct_roles > 4) return FBUS_ERR_INVALID;
    h->team_predict = predict_roles;
    h->team_correct = correct_roles;
    return FBUS_OK;
}

int fbus_ekf_set_policy_batch(fbus_ekf_t h, int total_filters)
{
    if (!h || total_filters < 0) return FBUS_ERR_INVALID;
    h->policy_batch = total_filters;
    h->lp.policy_b = total_filters;      // the two-wave (<= 256-register) kernel forms follow the job's size too (ekf_launch.hpp)
    return FBUS_OK;
}

int fbus_ekf_launch_info(fbus_ekf_t h, int what, int arg, int* value)
{
    if (!h || !value) return FBUS_ERR_INVALID;
    switch (what) {
        case FBUS_INFO_SIMDS: *value = h->lp.simds; break;
        case FBUS_INFO_ONE_ROUND_FILTERS: *value = h->lp.simds * 64; break;
        case FBUS_INFO_TWO_WAVE_MIN_B: *value = h->lp.two_wave_min_b; break;
        case FBUS_INFO_BIG_RECORDS_MB: *value = h->big_records_mb; break;
        case FBUS_INFO_MALL_MB: *value = (int)(h->mall_bytes >> 20); break;
        case FBUS_INFO_L2_KB: *value = (int)(h->l2_bytes >> 10); break;
        case FBUS_INFO_POLICY_BATCH: *value = h->policy_batch > 0 ? h->policy_batch : h->B; break;
        case FBUS_INFO_ROLES_PREDICT: *value = team_roles_predict(h, arg > 1 ? arg : 1); break;
        case FBUS_INFO_ROLES_MEAS: *value = team_roles_pixels(h, arg > 0 ? arg : 4); break;
        case FBUS_INFO_TEAM_FRAMES: *value = team_frames(h, MODE_STACKED) ? 1 : 0; break;
        case FBUS_INFO_MEAS_SPLIT: *value = meas_split_roles(h, arg > 0 ? arg : 4); break;
        default: return FBUS_ERR_INVALID;
    }
    return FBUS_OK;
}

int fbus_ekf_set_stream(fbus_ekf_t h, void* hip_stream)
{
    DeviceGuard guard_(h);
    if (!h) return FBUS_ERR_INVALID;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    h->stream = (hip_stream == FBUS_STREAM_OWN) ? h->own_stream : (hipStream_t)hip_stream;   // NULL = legacy default stream
    return FBUS_OK;
}

// cross-stream ordering without a host sync: one reusable event per handle
static int order_streams(fbus_ekf_t h, hipStream_t first, hipStream_t then)
{
    if (first == then) return FBUS_OK;
    if (!h->order_ev) HIP_TRY(h, hipEventCreateWithFlags(&h->order_ev, hipEventDisableTiming));
    HIP_TRY(h, hipEventRecord(h->order_ev, first));
    HIP_TRY(h, hipStreamWaitEvent(then, h->order_ev, 0));
    return FBUS_OK;
}

int fbus_ekf_wait_stream(fbus_ekf_t h, void* other_stream)
{
    DeviceGuard guard_(h);
    if (!h || other_stream == FBUS_STREAM_OWN) return FBUS_ERR_INVALID;
    return order_streams(h, (hipStream_t)other_stream, h->stream);
}

int fbus_ekf_signal_stream(fbus_ekf_t h, void* other_stream)
{
    DeviceGuard guard_(h);
    if (!h || other_stream == FBUS_STREAM_OWN) return FBUS_ERR_INVALID;
    return order_streams(h, h->stream, (hipStream_t)other_stream);
}

int fbus_ekf_sync(fbus_ekf_t h)
{
    DeviceGuard guard_(h);
    if (!h) return FBUS_ERR_INVALID;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return FBUS_OK;
}

const char* fbus_ekf_last_error(fbus_ekf_t h) { return h ? h->err.c_str() : "null handle"; }

int fbus_ekf_set_state_dev(fbus_ekf_t h, const void* nominal, const void* rot, const void* P, const int32_t* prev_id)
{
    DeviceGuard guard_(h);
    if (!h) return FBUS_ERR_INVALID;
    return do_pack(h, nominal, rot, P, prev_id);
}

int fbus_ekf_get_state_dev(fbus_ekf_t h, void* nominal, void* rot, void* P, int32_t* prev_id)
{
    DeviceGuard guard_(h);
    if (!h) return FBUS_ERR_INVALID;
    return do_unpack(h, nominal, rot, P, prev_id);
}

int fbus_ekf_set_state(fbus_ekf_t h, const void* nominal, const void* rot, const void* P, const int32_t* prev_id)
{
    DeviceGuard guard_(h);
    if (!h) return FBUS_ERR_INVALID;
    const size_t es = esize(h), B = (size_t)h->B, N = (size_t)h->N;
    const void *dn, *dr, *dP, *dp;
    int rc;
    if ((rc = stage_in(h, 0, nominal, B * 19 * es, &dn)) != FBUS_OK) return rc;
    if ((rc = stage_in(h, 1, rot, B * 9 * es, &dr)) != FBUS_OK) return rc;
    if ((rc = stage_in(h, 2, P, B * N * N * es, &dP)) != FBUS_OK) return rc;
    if ((rc = stage_in(h, 3, prev_id, B * 4, &dp)) != FBUS_OK) return rc;
    if ((rc = do_pack(h, dn, dr, dP, (const int32_t*)dp)) != FBUS_OK) return rc;
    HIP_TRY(h, hipStreamSynchronize(h->stream));    // host buffers may be reused by the caller
    return FBUS_OK;
}

int fbus_ekf_get_state(fbus_ekf_t h, void* nominal, void* rot, void* P, int32_t* prev_id)
{
    DeviceGuard guard_(h);
    if (!h) return FBUS_ERR_INVALID;
    const size_t es = esize(h), B = (size_t)h->B, N = (size_t)h->N;
    int rc;
    if (nominal && (rc = ensure_stage(h, 0, B * 19 * es)) != FBUS_OK) return rc;
    if (rot && (rc = ensure_stage(h, 1, B * 9 * es)) != FBUS_OK) return rc;
    if (P && (rc = ensure_stage(h, 2, B * N * N * es)) != FBUS_OK) return rc;
    if (prev_id && (rc = ensure_stage(h, 3, B * 4)) != FBUS_OK) return rc;
    if ((rc = do_unpack(h, nominal ? h->stage[0] : nullptr, rot ? h->stage[1] : nullptr, P ? h->stage[2] : nullptr,
                        prev_id ? (int32_t*)h->stage[3] : nullptr)) != FBUS_OK) return rc;
    if (nominal) HIP_TRY(h, hipMemcpyAsync(nominal, h->stage[0], B * 19 * es, hipMemcpyDeviceToHost, h->stream));
    if (rot) HIP_TRY(h, hipMemcpyAsync(rot, h->stage[1], B * 9 * es, hipMemcpyDeviceToHost, h->stream));
    if (P) HIP_TRY(h, hipMemcpyAsync(P, h->stage[2], B * N * N * es, hipMemcpyDeviceToHost, h->stream));
    if (prev_id) HIP_TRY(h, hipMemcpyAsync(prev_id, h->stage[3], B * 4, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return FBUS_OK;
}

int fbus_ekf_reset_cov(fbus_ekf_t h)
{
    DeviceGuard guard_(h);
    if (!h) return FBUS_ERR_INVALID;
    return do_reset_cov(h);
}

// ---------------------------------------------------------------------------------
// multi-GPU: the one collective of the path, a gather of the packed records over RCCL
// ---------------------------------------------------------------------------------
// RCCL is bound at first use (dlopen by soname): a process that already holds an RCCL -- PyTorch-ROCm loads its own copy of
// librccl.so.1 -- shares it, and the library still loads on a box without RCCL (the single-GPU path never touches it).
extern "C++" {
namespace {
struct Rccl {
    using UniqueId = struct { char internal[128]; };
    int (*GetUniqueId)(UniqueId*) = nullptr;
    int (*CommInitRank)(void**, int, UniqueId, int) = nullptr;
    int (*CommInitAll)(void**, int, const int*) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    int (*Broadcast)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    std::string err;
    bool ok = false;
};
Rccl& rccl()
{
    static Rccl r = [] {
        Rccl x;
        void* lib = nullptr;
        for (const char* name : { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" })
            if ((lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL))) break;
        if (!lib) {
            const char* why = dlerror();            // one call: dlerror() clears the message it returns
            x.err = std::string("librccl.so.1 not found: ") + (why ? why : "");
            return x;
        }
        auto sym = [&](const char* n) { void* p = dlsym(lib, n); if (!p && x.err.empty()) x.err = std::string("RCCL symbol missing: ") + n; return p; };
        x.GetUniqueId = (decltype(x.GetUniqueId))sym("ncclGetUniqueId");
        x.CommInitRank = (decltype(x.CommInitRank))sym("ncclCommInitRank");
        x.CommInitAll = (decltype(x.CommInitAll))sym("ncclCommInitAll");
        x.CommDestroy = (decltype(x.CommDestroy))sym("ncclCommDestroy");
        x.AllGather = (decltype(x.AllGather))sym("ncclAllGather");
        x.Broadcast = (decltype(x.Broadcast))sym("ncclBroadcast");
        x.GroupStart = (decltype(x.GroupStart))sym("ncclGroupStart");
        x.GroupEnd = (decltype(x.GroupEnd))sym("ncclGroupEnd");
        x.GetErrorString = (decltype(x.GetErrorString))sym("ncclGetErrorString");
        x.ok = x.err.empty();
        return x;
    }();
    return r;
}
constexpr int kNcclUint8 = 1;       // ncclDataType_t (rccl.h): ncclInt8 = 0, ncclUint8 = 1
int rccl_fail(fbus_ekf_t h, const char* what, int rc)
{
    Rccl& r = rccl();
    return fail(h, FBUS_ERR_HIP, std::string(what) + ": " + (r.GetErrorString ? r.GetErrorString(rc) : "RCCL error"));
}
}  // namespace
}  // extern "C++"

int fbus_ekf_comm_unique_id(void* id128)
{
    if (!id128) return FBUS_ERR_INVALID;
    Rccl& r = rccl();
    if (!r.ok) return FBUS_ERR_UNSUPPORTED;
    Rccl::UniqueId id;
    if (r.GetUniqueId(&id) != 0) return FBUS_ERR_HIP;
    std::memcpy(id128, &id, sizeof(id));
    return FBUS_OK;
}

int fbus_ekf_comm_init(fbus_ekf_t h, const void* id128, int rank, int world)
{
    DeviceGuard guard_(h);
    if (!h || !id128 || world < 1 || rank < 0 || rank >= world) return FBUS_ERR_INVALID;
    Rccl& r = rccl();
    if (!r.ok) return fail(h, FBUS_ERR_UNSUPPORTED, r.err);
    (void)fbus_ekf_comm_destroy(h);
    Rccl::UniqueId id;
    std::memcpy(&id, id128, sizeof(id));
    void* comm = nullptr;
    const int rc = r.CommInitRank(&comm, world, id, rank);         // binds to the current device = the handle's
    if (rc != 0) return rccl_fail(h, "ncclCommInitRank", rc);
    h->comm = comm; h->own_comm = true; h->comm_rank = rank; h->comm_world = world;
    return FBUS_OK;
}

int fbus_ekf_comm_init_all(fbus_ekf_t* handles, int n)
{
    // one process, n handles on n DIFFERENT devices (fbus::NodeFilter): ncclCommInitAll, rank k = handles[k]
    if (!handles || n < 1) return FBUS_ERR_INVALID;
    for (int k = 0; k < n; ++k) {
        if (!handles[k]) return FBUS_ERR_INVALID;
        for (int j = 0; j < k; ++j)
            if (handles[j]->device == handles[k]->device)
                return fail(handles[k], FBUS_ERR_INVALID, "fbus_ekf_comm_init_all: two handles on one device (RCCL wants one rank per device)");
    }
    Rccl& r = rccl();
    if (!r.ok) return fail(handles[0], FBUS_ERR_UNSUPPORTED, r.err);
    std::vector<int> devs(n);
    std::vector<void*> comms(n, nullptr);
    for (int k = 0; k < n; ++k) { devs[k] = handles[k]->device; (void)fbus_ekf_comm_destroy(handles[k]); }
    const int rc = r.CommInitAll(comms.data(), n, devs.data());
    if (rc != 0) return rccl_fail(handles[0], "ncclCommInitAll", rc);
    for (int k = 0; k < n; ++k) { handles[k]->comm = comms[k]; handles[k]->own_comm = true; handles[k]->comm_rank = k; handles[k]->comm_world = n; }
    return FBUS_OK;
}

int fbus_ekf_gather_group(fbus_ekf_t* handles, int n, void* const* out_dev, const size_t* bytes_of_rank)
{
    // the gather of ALL ranks of one process in one RCCL group (a single thread may not issue the ranks' collectives one by one)
    if (!handles || !out_dev || n < 1) return FBUS_ERR_INVALID;
    // everything fbus_ekf_gather would refuse is refused HERE, before the group is opened: a rank that fails inside an open group
    // leaves the ranks in front of it with an incomplete collective enqueued, and ncclGroupEnd would then hang their streams
    for (int k = 0; k < n; ++k) {
        fbus_ekf_t h = handles[k];
        if (!h) return FBUS_ERR_INVALID;
        if (!out_dev[k]) return fail(h, FBUS_ERR_INVALID, "fbus_ekf_gather_group: out_dev[k] is NULL");
        if (!h->comm) return fail(h, FBUS_ERR_INVALID, "fbus_ekf_gather_group: a handle without communicator (fbus_ekf_comm_init_all first)");
        if (h->comm_world != n || h->comm_rank != k)
            return fail(h, FBUS_ERR_INVALID, "fbus_ekf_gather_group: handles[k] is not rank k of an n-rank communicator");
        if (bytes_of_rank && bytes_of_rank[k] != h->rec_bytes)
            return fail(h, FBUS_ERR_INVALID, "fbus_ekf_gather_group: bytes_of_rank[k] is not handles[k]'s record size");
        if (!bytes_of_rank && h->rec_bytes != handles[0]->rec_bytes)
            return fail(h, FBUS_ERR_INVALID, "fbus_ekf_gather_group: ragged shards need bytes_of_rank");
    }
    Rccl& r = rccl();
    if (!r.ok) return fail(handles[0], FBUS_ERR_UNSUPPORTED, r.err);
    int rc = r.GroupStart();
    int first = FBUS_OK;
    for (int k = 0; k < n && rc == 0; ++k) {
        const int e = fbus_ekf_gather(handles[k], out_dev[k], bytes_of_rank);
        if (e != FBUS_OK && first == FBUS_OK) first = e;
    }
    const int rc2 = r.GroupEnd();
    if (first != FBUS_OK) return first;
    if (rc != 0 || rc2 != 0) return rccl_fail(handles[0], "ncclGroupStart / ncclGroupEnd", rc != 0 ? rc : rc2);
    return FBUS_OK;
}

int fbus_ekf_copy_records(fbus_ekf_t h, void* dst, int dst_device)
{
    // this handle's packed records -> dst on dst_device (any device of the process), on the handle's stream: the gather of a
    // single-process job whose consumer sits on ONE device (or whose shards share a device) without a communicator
    DeviceGuard guard_(h);
    if (!h || !dst || dst_device < 0) return FBUS_ERR_INVALID;
    if (dst_device == h->device) HIP_TRY(h, hipMemcpyAsync(dst, h->recs, h->rec_bytes, hipMemcpyDeviceToDevice, h->stream));
    else HIP_TRY(h, hipMemcpyPeerAsync(dst, dst_device, h->recs, h->device, h->rec_bytes, h->stream));
    return FBUS_OK;
}

int fbus_ekf_comm_attach(fbus_ekf_t h, void* nccl_comm, int rank, int world)
{
    if (!h || !nccl_comm || world < 1 || rank < 0 || rank >= world) return FBUS_ERR_INVALID;
    Rccl& r = rccl();
    if (!r.ok) return fail(h, FBUS_ERR_UNSUPPORTED, r.err);
    (void)fbus_ekf_comm_destroy(h);
    h->comm = nccl_comm; h->own_comm = false; h->comm_rank = rank; h->comm_world = world;
    return FBUS_OK;
}

int fbus_ekf_comm_destroy(fbus_ekf_t h)
{
    if (!h) return FBUS_ERR_INVALID;
    if (h->comm && h->own_comm) {
        DeviceGuard guard_(h);
        (void)hipStreamSynchronize(h->stream);
        (void)rccl().CommDestroy(h->comm);
    }
    h->comm = nullptr; h->own_comm = false; h->comm_rank = 0; h->comm_world = 1;
    return FBUS_OK;
}

int fbus_ekf_gather(fbus_ekf_t h, void* out_dev, const size_t* bytes_of_rank)
{
    DeviceGuard guard_(h);
    if (!h || !out_dev) return FBUS_ERR_INVALID;
    if (!h->comm) return fail(h, FBUS_ERR_INVALID, "fbus_ekf_gather: no communicator (fbus_ekf_comm_init / fbus_ekf_comm_attach first)");
    Rccl& r = rccl();
    const int W = h->comm_world;
    bool equal = true;
    if (bytes_of_rank) {
        if (bytes_of_rank[h->comm_rank] != h->rec_bytes) return fail(h, FBUS_ERR_INVALID, "fbus_ekf_gather: bytes_of_rank[rank] is not this handle's record size");
        for (int k = 0; k < W; ++k) equal = equal && bytes_of_rank[k] == h->rec_bytes;
    }
    if (equal) {
        // equal shards (weak scaling, or a total that divides evenly): one all-gather of the packed records
        const int rc = r.AllGather(h->recs, out_dev, h->rec_bytes, kNcclUint8, h->comm, h->stream);
        if (rc != 0) return rccl_fail(h, "ncclAllGather", rc);
        return FBUS_OK;
    }
    // ragged shards (a total batch cut into 64-aligned ranges): one grouped broadcast per rank = an all-gather-v, no padding
    int rc = r.GroupStart();
    size_t off = 0;
    for (int k = 0; k < W && rc == 0; ++k) {
        rc = r.Broadcast(h->recs, (char*)out_dev + off, bytes_of_rank[k], kNcclUint8, k, h->comm, h->stream);
        off += bytes_of_rank[k];
    }
    const int rc2 = r.GroupEnd();
    if (rc != 0 || rc2 != 0) return rccl_fail(h, "ncclBroadcast (grouped)", rc != 0 ? rc : rc2);
    return FBUS_OK;
}

int fbus_ekf_records(fbus_ekf_t h, void** dev_ptr, size_t* bytes_per_filter, size_t* total_bytes)
{
    DeviceGuard guard_(h);
    if (!h) return FBUS_ERR_INVALID;
    if (dev_ptr) *dev_ptr = h->recs;
    if (bytes_per_filter) *bytes_per_filter = h->bytes_per_filter;
    if (total_bytes) *total_bytes = h->rec_bytes;
    return FBUS_OK;
}

int fbus_ekf_attach_records(fbus_ekf_t h, void* dev_ptr, size_t total_bytes)
{
    DeviceGuard guard_(h);
    if (!h || !dev_ptr) return FBUS_ERR_INVALID;
    if (total_bytes != h->rec_bytes) return fail(h, FBUS_ERR_INVALID, "attach_records: size mismatch");
    if (((uintptr_t)dev_ptr & 15) != 0) return fail(h, FBUS_ERR_INVALID, "attach_records: pointer not 16-byte aligned");
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    HIP_TRY(h, hipMemcpy(dev_ptr, h->recs, h->rec_bytes, hipMemcpyDeviceToDevice));
    if (h->own_recs) HIP_TRY(h, hipFree(h->recs));
    h->recs = dev_ptr;
    h->own_recs = false;
    return FBUS_OK;
}

int fbus_ekf_predict_n_dev(fbus_ekf_t h, int K, const void* accel, const void* gyro, const void* dt, int dt_per_filter)
{
    DeviceGuard guard_(h);
    if (!h || !accel || !gyro || !dt || K < 1) return FBUS_ERR_INVALID;
    return launch_predict(h, K, accel, gyro, dt, dt_per_filter);
}

int fbus_ekf_predict_dev(fbus_ekf_t h, const void* accel, const void* gyro, const void* dt, int dt_per_filter)
{
    DeviceGuard guard_(h);
    return fbus_ekf_predict_n_dev(h, 1, accel, gyro, dt, dt_per_filter);
}

int fbus_ekf_predict_n(fbus_ekf_t h, int K, const void* accel, const void* gyro, const void* dt, int dt_per_filter)
{
    DeviceGuard guard_(h);
    if (!h || !accel || !gyro || !dt || K < 1) return FBUS_ERR_INVALID;
    const size_t es = esize(h), B = (size_t)h->B;
    const void *da, *dg, *dd;
    int rc;
    if ((rc = stage_in(h, 0, accel, (size_t)K * B * 3 * es, &da)) != FBUS_OK) return rc;
    if ((rc = stage_in(h, 1, gyro, (size_t)K * B * 3 * es, &dg)) != FBUS_OK) return rc;
    if ((rc = stage_in(h, 2, dt, (size_t)K * (dt_per_filter ? B : 1) * es, &dd)) != FBUS_OK) return rc;
    if ((rc = launch_predict(h, K, da, dg, dd, dt_per_filter)) != FBUS_OK) return rc;
    HIP_TRY(h, hipStreamSynchronize(h->stream));    // staging buffers are reused by the next host call
    return FBUS_OK;
}

int fbus_ekf_predict(fbus_ekf_t h, const void* accel, const void* gyro, const void* dt, int dt_per_filter)
{
    DeviceGuard guard_(h);
    return fbus_ekf_predict_n(h, 1, accel, gyro, dt, dt_per_filter);
}

int fbus_ekf_predict_n_async(fbus_ekf_t h, int K, const void* accel, const void* gyro, const void* dt, int dt_per_filter)
{
    DeviceGuard guard_(h);
    if (!h || !accel || !gyro || !dt || K < 1) return FBUS_ERR_INVALID;
    const size_t es = esize(h), B = (size_t)h->B;
    const void *da, *dg, *dd;
    AsyncPiece pc[3] = { { accel, (size_t)K * B * 3 * es, &da }, { gyro, (size_t)K * B * 3 * es, &dg },
                         { dt, (size_t)K * (dt_per_filter ? B : 1) * es, &dd } };
    fbus_ekf::AsyncSlot* a;
    int rc = async_begin(h, pc, 3, &a);
    if (rc != FBUS_OK) return rc;
    return async_end(h, a, launch_predict(h, K, da, dg, dd, dt_per_filter));
}

int fbus_ekf_predict_async(fbus_ekf_t h, const void* accel, const void* gyro, const void* dt, int dt_per_filter)
{
    return fbus_ekf_predict_n_async(h, 1, accel, gyro, dt, dt_per_filter);
}

int fbus_ekf_correct_async(fbus_ekf_t h, int M, const int32_t* ids, const void* pos, const void* quat, int mode, const uint8_t* skip)
{
    DeviceGuard guard_(h);
    if (!h || !ids || !pos || !quat || M < 1 || M > FBUS_MAX_VISIBLE) return FBUS_ERR_INVALID;
    if (mode != FBUS_MODE_NEAREST && mode != FBUS_MODE_STACKED) return FBUS_ERR_UNSUPPORTED;
    const size_t es = esize(h), B = (size_t)h->B;
    const void *di, *dp, *dq, *ds;
    AsyncPiece pc[4] = { { ids, B * M * 4, &di }, { pos, B * M * 3 * es, &dp }, { quat, B * M * 4 * es, &dq }, { skip, skip ? B : 0, &ds } };
    fbus_ekf::AsyncSlot* a;
    int rc = async_begin(h, pc, 4, &a);
    if (rc != FBUS_OK) return rc;
    return async_end(h, a, launch_correct(h, M, (const int32_t*)di, dp, dq, mode, (const uint8_t*)ds));
}

int fbus_ekf_correct_pixels_async(fbus_ekf_t h, int M, const int32_t* ids, const void* left, const void* right, const uint8_t* skip)
{
    DeviceGuard guard_(h);
    if (!h || !ids || !left || M < 1 || M > FBUS_MAX_VISIBLE) return FBUS_ERR_INVALID;
    if (!(h->prm.r_pix > 0)) return fail(h, FBUS_ERR_INVALID, "r_pix must be positive");
    const size_t es = esize(h), B = (size_t)h->B;
    const void *di, *dl, *dr, *ds;
    AsyncPiece pc[4] = { { ids, B * M * 4, &di }, { left, B * M * 8 * es, &dl }, { right, right ? B * M * 8 * es : 0, &dr }, { skip, skip ? B : 0, &ds } };
    fbus_ekf::AsyncSlot* a;
    int rc = async_begin(h, pc, 4, &a);
    if (rc != FBUS_OK) return rc;
    return async_end(h, a, launch_correct_pixels(h, M, (const int32_t*)di, dl, dr, (const uint8_t*)ds));
}

int fbus_ekf_async_inputs_consumed(fbus_ekf_t h)
{
    DeviceGuard guard_(h);
    if (!h) return FBUS_ERR_INVALID;
    if (h->copy_stream) HIP_TRY(h, hipStreamSynchronize(h->copy_stream));
    return FBUS_OK;
}

int fbus_ekf_async_stats(fbus_ekf_t h, int64_t* calls, int64_t* waits, int64_t* direct_pieces)
{
    if (!h) return FBUS_ERR_INVALID;
    if (calls) *calls = h->async_calls;
    if (waits) *waits = h->async_waits;
    if (direct_pieces) *direct_pieces = h->async_direct;
    return FBUS_OK;
}

int fbus_ekf_host_register(void* ptr, size_t bytes)
{
    if (!ptr || bytes == 0) return FBUS_ERR_INVALID;
    return hipHostRegister(ptr, bytes, hipHostRegisterDefault) == hipSuccess ? FBUS_OK : ((void)hipGetLastError(), FBUS_ERR_HIP);
}

int fbus_ekf_host_unregister(void* ptr)
{
    if (!ptr) return FBUS_ERR_INVALID;
    return hipHostUnregister(ptr) == hipSuccess ? FBUS_OK : ((void)hipGetLastError(), FBUS_ERR_HIP);
}

int fbus_ekf_correct_dev(fbus_ekf_t h, int M, const int32_t* ids, const void* pos, const void* quat, int mode,
                         const uint8_t* skip)
{
    DeviceGuard guard_(h);
    if (!h || !ids || !pos || !quat || M < 1 || M > FBUS_MAX_VISIBLE) return FBUS_ERR_INVALID;
    if (mode != FBUS_MODE_NEAREST && mode != FBUS_MODE_STACKED) return FBUS_ERR_UNSUPPORTED;
    return launch_correct(h, M, ids, pos, quat, mode, skip);
}

int fbus_ekf_correct(fbus_ekf_t h, int M, const int32_t* ids, const void* pos, const void* quat, int mode,
                     const uint8_t* skip)
{
    DeviceGuard guard_(h);
    if (!h || !ids || !pos || !quat || M < 1 || M > FBUS_MAX_VISIBLE) return FBUS_ERR_INVALID;
    if (mode != FBUS_MODE_NEAREST && mode != FBUS_MODE_STACKED) return FBUS_ERR_UNSUPPORTED;
    const size_t es = esize(h), B = (size_t)h->B;
    const void *di, *dp, *dq, *ds;
    int rc;
    if ((rc = stage_in(h, 0, ids, B * M * 4, &di)) != FBUS_OK) return rc;
    if ((rc = stage_in(h, 1, pos, B * M * 3 * es, &dp)) != FBUS_OK) return rc;
    if ((rc = stage_in(h, 2, quat, B * M * 4 * es, &dq)) != FBUS_OK) return rc;
    if ((rc = stage_in(h, 3, skip, B, &ds)) != FBUS_OK) return rc;
    if ((rc = launch_correct(h, M, (const int32_t*)di, dp, dq, mode, (const uint8_t*)ds)) != FBUS_OK) return rc;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return FBUS_OK;
}

int fbus_ekf_correct_corners_dev(fbus_ekf_t h, int M, const int32_t* ids, const void* left, const void* right,
                                 int geometry, int mode, const uint8_t* skip)
{
    DeviceGuard guard_(h);
    if (!h || !ids || !left || M < 1 || M > FBUS_MAX_VISIBLE) return FBUS_ERR_INVALID;
    if (geometry != FBUS_VIS_REFRACTIVE && geometry != FBUS_VIS_PINHOLE && geometry != FBUS_VIS_CORNERS3D)
        return FBUS_ERR_UNSUPPORTED;
    if (geometry != FBUS_VIS_CORNERS3D && !right) return FBUS_ERR_INVALID;
    if (mode != FBUS_MODE_NEAREST && mode != FBUS_MODE_STACKED) return FBUS_ERR_UNSUPPORTED;
    return launch_correct_corners(h, M, ids, left, right, geometry, mode, skip);
}

int fbus_ekf_correct_corners(fbus_ekf_t h, int M, const int32_t* ids, const void* left, const void* right,
                             int geometry, int mode, const uint8_t* skip)
{
    DeviceGuard guard_(h);
    if (!h || !ids || !left || M < 1 || M > FBUS_MAX_VISIBLE) return FBUS_ERR_INVALID;
    const size_t es = esize(h), B = (size_t)h->B;
    const size_t lw = geometry == FBUS_VIS_CORNERS3D ? 12 : 8;
    const void *di, *dl, *dr, *ds;
    int rc;
    if ((rc = stage_in(h, 0, ids, B * M * 4, &di)) != FBUS_OK) return rc;
    if ((rc = stage_in(h, 1, left, B * M * lw * es, &dl)) != FBUS_OK) return rc;
    if ((rc = stage_in(h, 2, geometry == FBUS_VIS_CORNERS3D ? nullptr : right, B * M * 8 * es, &dr)) != FBUS_OK) return rc;
    if ((rc = stage_in(h, 3, skip, B, &ds)) != FBUS_OK) return rc;
    if ((rc = fbus_ekf_correct_corners_dev(h, M, (const int32_t*)di, dl, dr, geometry, mode, (const uint8_t*)ds)) != FBUS_OK)
        return rc;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return FBUS_OK;
}

int fbus_ekf_correct_pixels_dev(fbus_ekf_t h, int M, const int32_t* ids, const void* left, const void* right,
                                const uint8_t* skip)
{
    DeviceGuard guard_(h);
    if (!h || !ids || !left || M < 1 || M > FBUS_MAX_VISIBLE) return FBUS_ERR_INVALID;
    if (!(h->prm.r_pix > 0)) return fail(h, FBUS_ERR_INVALID, "r_pix must be positive");
    return launch_correct_pixels(h, M, ids, left, right, skip);
}

int fbus_ekf_correct_pixels(fbus_ekf_t h, int M, const int32_t* ids, const void* left, const void* right,
                            const uint8_t* skip)
{
    DeviceGuard guard_(h);
    if (!h || !ids || !left || M < 1 || M > FBUS_MAX_VISIBLE) return FBUS_ERR_INVALID;
    const size_t es = esize(h), B = (size_t)h->B;
    const void *di, *dl, *dr, *ds;
    int rc;
    if ((rc = stage_in(h, 0, ids, B * M * sizeof(int32_t), &di)) != FBUS_OK) return rc;
    if ((rc = stage_in(h, 1, left, B * M * 8 * es, &dl)) != FBUS_OK) return rc;
    if ((rc = stage_in(h, 2, right, right ? B * M * 8 * es : 0, &dr)) != FBUS_OK) return rc;
    if ((rc = stage_in(h, 3, skip, skip ? B : 0, &ds)) != FBUS_OK) return rc;
    if ((rc = fbus_ekf_correct_pixels_dev(h, M, (const int32_t*)di, dl, dr, (const uint8_t*)ds)) != FBUS_OK) return rc;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return FBUS_OK;
}

int fbus_ekf_get_applied(fbus_ekf_t h, uint8_t* applied_host)
{
    DeviceGuard guard_(h);
    if (!h || !applied_host) return FBUS_ERR_INVALID;
    HIP_TRY(h, hipMemcpyAsync(applied_host, h->d_applied, (size_t)h->B, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return FBUS_OK;
}

int fbus_ekf_frame_fused_dev(fbus_ekf_t h, int K, const void* accel, const void* gyro, const void* dt, int dt_per_filter,
                             int M, const int32_t* ids, const void* pos, const void* quat, int mode, const uint8_t* skip)
{
    DeviceGuard guard_(h);
    if (!h || K < 0 || M < 0 || M > FBUS_MAX_VISIBLE) return FBUS_ERR_INVALID;
    if (K > 0 && (!accel || !gyro || !dt)) return FBUS_ERR_INVALID;
    if (M > 0 && (!ids || !pos || !quat)) return FBUS_ERR_INVALID;
    if (mode != FBUS_MODE_NEAREST && mode != FBUS_MODE_STACKED) return FBUS_ERR_UNSUPPORTED;
    return launch_frame(h, K, accel, gyro, dt, dt_per_filter, M, ids, pos, quat, mode, skip);
}

int fbus_ekf_frame_meas_fused_dev(fbus_ekf_t h, int K, const void* accel, const void* gyro, const void* dt, int dt_per_filter,
                                  int kind, int M, const int32_t* ids, const void* left, const void* right, int geometry, int mode,
                                  const uint8_t* skip)
{
    DeviceGuard guard_(h);
    // everything is validated before the first launch: a rejected call must not leave the state advanced by the K predicts
    if (!h || K < 0 || M < 0 || M > FBUS_MAX_VISIBLE) return FBUS_ERR_INVALID;
    if (kind != FBUS_MEAS_PIXELS && kind != FBUS_MEAS_CORNERS) return FBUS_ERR_UNSUPPORTED;
    if (K > 0 && (!accel || !gyro || !dt)) return FBUS_ERR_INVALID;
    if (M > 0 && (!ids || !left)) return FBUS_ERR_INVALID;
    if (kind == FBUS_MEAS_PIXELS) {
        if (!(h->prm.r_pix > 0)) return fail(h, FBUS_ERR_INVALID, "r_pix must be positive");
        geometry = FBUS_VIS_REFRACTIVE; mode = FBUS_MODE_STACKED;          // (not used by the pixel rows)
    } else {
        if (geometry != FBUS_VIS_REFRACTIVE && geometry != FBUS_VIS_PINHOLE && geometry != FBUS_VIS_CORNERS3D) return FBUS_ERR_UNSUPPORTED;
        if (M > 0 && geometry != FBUS_VIS_CORNERS3D && !right) return FBUS_ERR_INVALID;
        if (mode != FBUS_MODE_NEAREST && mode != FBUS_MODE_STACKED) return FBUS_ERR_UNSUPPORTED;
    }
    // (advisor, round 5) on every route, not only the resident one: the per-call updates behind the fall-back routes (fp64 records, team /
    // split forms, K > 255) refuse unaligned image points AFTER the predicts have run
    if (M > 0 && ((reinterpret_cast<uintptr_t>(left) | reinterpret_cast<uintptr_t>(right)) & 15) != 0)
        return fail(h, FBUS_ERR_INVALID, "fbus_ekf_frame_meas_fused_dev: left / right must be 16-byte aligned device pointers");
    if (K > 255) {          // (the resident kernel counts a frame's samples in a byte)
        int rc = launch_predict(h, K, accel, gyro, dt, dt_per_filter);
        if (rc == FBUS_OK && M > 0)
            rc = kind == FBUS_MEAS_PIXELS ? launch_correct_pixels(h, M, ids, left, right, skip)
                                          : launch_correct_corners(h, M, ids, left, right, geometry, mode, skip);
        return rc;
    }
    const unsigned char kc1 = (unsigned char)K;
    return launch_frame_meas(h, 1, &kc1, accel, gyro, dt, dt_per_filter, kind, M, ids, left, right, geometry, mode, skip);
}

int fbus_ekf_frames_meas_fused_dev(fbus_ekf_t h, int nframes, const int32_t* kcount, const void* accel, const void* gyro, const void* dt,
                                   int dt_per_filter, int kind, int M, const int32_t* ids, const void* left, const void* right, int geometry,
                                   int mode, const uint8_t* skip)
{
    DeviceGuard guard_(h);
    if (!h || nframes < 0 || nframes > FBUS_MAX_WINDOW_FRAMES || M < 0 || M > FBUS_MAX_VISIBLE) return FBUS_ERR_INVALID;
    if (kind != FBUS_MEAS_PIXELS && kind != FBUS_MEAS_CORNERS) return FBUS_ERR_UNSUPPORTED;
    if (nframes > 0 && !kcount) return FBUS_ERR_INVALID;
    unsigned char kc[FBUS_MAX_WINDOW_FRAMES];
    size_t total = 0;
    for (int f = 0; f < nframes; ++f) {
        if (kcount[f] < 0 || kcount[f] > 255) return FBUS_ERR_INVALID;
        kc[f] = (unsigned char)kcount[f];
        total += (size_t)kcount[f];
    }
    if (total > 0 && (!accel || !gyro || !dt)) return FBUS_ERR_INVALID;
    if (M > 0 && nframes > 0 && (!ids || !left)) return FBUS_ERR_INVALID;
    if (kind == FBUS_MEAS_PIXELS) {
        if (!(h->prm.r_pix > 0)) return fail(h, FBUS_ERR_INVALID, "r_pix must be positive");
        geometry = FBUS_VIS_REFRACTIVE; mode = FBUS_MODE_STACKED;
    } else {
        if (geometry != FBUS_VIS_REFRACTIVE && geometry != FBUS_VIS_PINHOLE && geometry != FBUS_VIS_CORNERS3D) return FBUS_ERR_UNSUPPORTED;
        if (M > 0 && nframes > 0 && geometry != FBUS_VIS_CORNERS3D && !right) return FBUS_ERR_INVALID;
        if (mode != FBUS_MODE_NEAREST && mode != FBUS_MODE_STACKED) return FBUS_ERR_UNSUPPORTED;
    }
    if (nframes == 0) return FBUS_OK;
    // every frame's arrays start a multiple of 16 bytes behind the first (B M x 32 / 48 bytes x element size): one check covers the window
    if (M > 0 && ((reinterpret_cast<uintptr_t>(left) | reinterpret_cast<uintptr_t>(right)) & 15) != 0)
        return fail(h, FBUS_ERR_INVALID, "fbus_ekf_frames_meas_fused_dev: left / right must be 16-byte aligned device pointers");
    // the resident window kernel where the frame form takes the resident kernel (fp32 records, one wave per tile); elsewhere frame by
    // frame through the frame entry point's routes -- the same arithmetic
    if (nframes > 1 && frame_meas_is_resident(h, kind, M, mode))
        return launch_frame_meas(h, nframes, kc, accel, gyro, dt, dt_per_filter, kind, M, ids, left, right, geometry, mode, skip);
    const size_t es = esize(h), B = (size_t)h->B;
    const size_t lw = (kind == FBUS_MEAS_CORNERS && geometry == FBUS_VIS_CORNERS3D) ? 12 : 8;
    size_t k0 = 0;
    for (int f = 0; f < nframes; ++f) {
        const int rc = launch_frame_meas(h, 1, kc + f, (const char*)accel + k0 * B * 3 * es, (const char*)gyro + k0 * B * 3 * es,
                                         (const char*)dt + k0 * (dt_per_filter ? B : 1) * es, dt_per_filter, kind, M,
                                         ids ? ids + (size_t)f * B * M : nullptr, left ? (const char*)left + (size_t)f * B * M * lw * es : nullptr,
                                         right ? (const char*)right + (size_t)f * B * M * 8 * es : nullptr, geometry, mode,
                                         skip ? skip + (size_t)f * B : nullptr);
        if (rc != FBUS_OK) return rc;
        k0 += kc[f];
    }
    return FBUS_OK;
}

int fbus_ekf_frames_fused_dev(fbus_ekf_t h, int nframes, const int32_t* kcount, const void* accel, const void* gyro,
                              const void* dt, int dt_per_filter, int M, const int32_t* ids, const void* pos,
                              const void* quat, int mode, const uint8_t* skip)
{
    DeviceGuard guard_(h);
    if (!h || nframes < 0 || nframes > FBUS_MAX_WINDOW_FRAMES || M < 0 || M > FBUS_MAX_VISIBLE) return FBUS_ERR_INVALID;
    if (nframes > 0 && !kcount) return FBUS_ERR_INVALID;
    if (mode != FBUS_MODE_NEAREST && mode != FBUS_MODE_STACKED) return FBUS_ERR_UNSUPPORTED;
    unsigned char kc[FBUS_MAX_WINDOW_FRAMES];
    size_t total = 0;
    for (int f = 0; f < nframes; ++f) {
        if (kcount[f] < 0 || kcount[f] > 255) return FBUS_ERR_INVALID;
        kc[f] = (unsigned char)kcount[f];
        total += (size_t)kcount[f];
    }
    if (total > 0 && (!accel || !gyro || !dt)) return FBUS_ERR_INVALID;
    if (M > 0 && nframes > 0 && (!ids || !pos || !quat)) return FBUS_ERR_INVALID;
    if (nframes == 0) return FBUS_OK;
    // no resident-record kernel for fp64 and for (Joseph, nearest) -- see launch_frame_t: those windows run frame by frame,
    // the same arithmetic
    const bool resident = h->dtype == 32 && !(h->prm.cov_form == FBUS_COV_JOSEPH && mode != FBUS_MODE_STACKED);
    if (resident) return launch_frames(h, nframes, kc, accel, gyro, dt, dt_per_filter, M, ids, pos, quat, mode, skip);
    const size_t es = esize(h), B = (size_t)h->B;
    size_t k0 = 0;
    for (int f = 0; f < nframes; ++f) {
        const int rc = launch_frame(h, kc[f], (const char*)accel + k0 * B * 3 * es, (const char*)gyro + k0 * B * 3 * es,
                                    (const char*)dt + k0 * (dt_per_filter ? B : 1) * es, dt_per_filter, M,
                                    ids ? ids + (size_t)f * B * M : nullptr, pos ? (const char*)pos + (size_t)f * B * M * 3 * es : nullptr,
                                    quat ? (const char*)quat + (size_t)f * B * M * 4 * es : nullptr, mode,
                                    skip ? skip + (size_t)f * B : nullptr);
        if (rc != FBUS_OK) return rc;
        k0 += kc[f];
    }
    return FBUS_OK;
}

int fbus_ekf_frame_dev(fbus_ekf_t h, int K, const void* accel, const void* gyro, const void* dt, int dt_per_filter,
                       int M, const int32_t* ids, const void* pos, const void* quat, int mode, const uint8_t* skip)
{
    DeviceGuard guard_(h);
    // everything is validated before the first launch (the same checks as frame_fused_dev): a rejected call must not
    // leave the state advanced by the K predicts
    if (!h || K < 0 || M < 0 || M > FBUS_MAX_VISIBLE) return FBUS_ERR_INVALID;
    if (K > 0 && (!accel || !gyro || !dt)) return FBUS_ERR_INVALID;
    if (M > 0 && (!ids || !pos || !quat)) return FBUS_ERR_INVALID;
    if (mode != FBUS_MODE_NEAREST && mode != FBUS_MODE_STACKED) return FBUS_ERR_UNSUPPORTED;
    const size_t es = esize(h), B = (size_t)h->B;
    // one event pair around the whole run of K back-to-back predict launches: a pair per launch
    // would cost ~8 us of stream time each and read ~3 us long; duration / K is the per-launch time
    const bool sampled = (h->frame_count++ % h->timing_stride) == 0;
    const int ev = (K > 0 && sampled) ? timing_begin(h, FBUS_KERNEL_PREDICT, K) : -1;
    h->timing_suspended = true;
    int rc = FBUS_OK;
    for (int k = 0; k < K && rc == FBUS_OK; ++k) {
        const char* a = (const char*)accel + (size_t)k * B * 3 * es;
        const char* g = (const char*)gyro + (size_t)k * B * 3 * es;
        const char* d = (const char*)dt + (size_t)k * (dt_per_filter ? B : 1) * es;
        rc = launch_predict(h, 1, a, g, d, dt_per_filter);
    }
    timing_end(h, ev);
    if (rc != FBUS_OK) { h->timing_suspended = false; return rc; }
    h->timing_suspended = !sampled;
    if (M > 0) rc = fbus_ekf_correct_dev(h, M, ids, pos, quat, mode, skip);
    h->timing_suspended = false;
    return rc;
}

int fbus_ekf_marker_pose_dev(fbus_ekf_t h, int n, int geometry, const void* left, const void* right, void* pos,
                             void* quat, void* corners3d)
{
    DeviceGuard guard_(h);
    if (!h || n < 1 || !left || !pos || !quat) return FBUS_ERR_INVALID;
    if (geometry != FBUS_VIS_REFRACTIVE && geometry != FBUS_VIS_PINHOLE && geometry != FBUS_VIS_CORNERS3D)
        return FBUS_ERR_UNSUPPORTED;
    if (geometry != FBUS_VIS_CORNERS3D && !right) return FBUS_ERR_INVALID;
    if (h->dtype == 32) return launch_marker_pose_t<float>(h, n, geometry, left, right, pos, quat, corners3d);
    return launch_marker_pose_t<double>(h, n, geometry, left, right, pos, quat, corners3d);
}

int fbus_ekf_marker_pose(fbus_ekf_t h, int n, int geometry, const void* left, const void* right, void* pos,
                         void* quat, void* corners3d)
{
    DeviceGuard guard_(h);
    if (!h || n < 1 || !left || !pos || !quat) return FBUS_ERR_INVALID;
    const size_t es = esize(h), nn = (size_t)n;
    const size_t in_w = geometry == FBUS_VIS_CORNERS3D ? 12 : 8;
    const void *dl, *dr;
    int rc;
    if ((rc = stage_in(h, 0, left, nn * in_w * es, &dl)) != FBUS_OK) return rc;
    if ((rc = stage_in(h, 1, geometry == FBUS_VIS_CORNERS3D ? nullptr : right, nn * 8 * es, &dr)) != FBUS_OK) return rc;
    if ((rc = ensure_stage(h, 2, nn * 3 * es)) != FBUS_OK) return rc;
    if ((rc = ensure_stage(h, 3, nn * 4 * es)) != FBUS_OK) return rc;
    if (corners3d && (rc = ensure_stage(h, 4, nn * 12 * es)) != FBUS_OK) return rc;
    if ((rc = fbus_ekf_marker_pose_dev(h, n, geometry, dl, dr, h->stage[2], h->stage[3],
                                       corners3d ? h->stage[4] : nullptr)) != FBUS_OK) return rc;
    HIP_TRY(h, hipMemcpyAsync(pos, h->stage[2], nn * 3 * es, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipMemcpyAsync(quat, h->stage[3], nn * 4 * es, hipMemcpyDeviceToHost, h->stream));
    if (corners3d) HIP_TRY(h, hipMemcpyAsync(corners3d, h->stage[4], nn * 12 * es, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return FBUS_OK;
}

int fbus_ekf_init_gravity_bias_dev(fbus_ekf_t h, int T, const void* accel, const void* gyro)
{
    DeviceGuard guard_(h);
    if (!h || T < 1 || !accel || !gyro) return FBUS_ERR_INVALID;
    return do_init_gb(h, T, accel, gyro);
}

int fbus_ekf_init_gravity_bias(fbus_ekf_t h, int T, const void* accel, const void* gyro)
{
    DeviceGuard guard_(h);
    if (!h || T < 1 || !accel || !gyro) return FBUS_ERR_INVALID;
    const size_t bytes = (size_t)T * h->B * 3 * esize(h);
    const void *da, *dg;
    int rc;
    if ((rc = stage_in(h, 0, accel, bytes, &da)) != FBUS_OK) return rc;
    if ((rc = stage_in(h, 1, gyro, bytes, &dg)) != FBUS_OK) return rc;
    if ((rc = do_init_gb(h, T, da, dg)) != FBUS_OK) return rc;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return FBUS_OK;
}

int fbus_ekf_pose_init_dev(fbus_ekf_t h, int M, const int32_t* ids, const void* pos, const void* quat, int what,
                           const uint8_t* mask)
{
    DeviceGuard guard_(h);
    if (!h || !ids || !pos || !quat || M < 1 || M > FBUS_MAX_VISIBLE) return FBUS_ERR_INVALID;
    if (what != FBUS_POSE_INIT && what != FBUS_POSE_RESET) return FBUS_ERR_INVALID;
    return do_pose_init(h, M, ids, pos, quat, what, mask, nullptr);
}

int fbus_ekf_vision_only_pose_dev(fbus_ekf_t h, int M, const int32_t* ids, const void* pos, const void* quat,
                                  void* out_pose)
{
    DeviceGuard guard_(h);
    if (!h || !ids || !pos || !quat || !out_pose || M < 1 || M > FBUS_MAX_VISIBLE) return FBUS_ERR_INVALID;
    return do_pose_init(h, M, ids, pos, quat, 2, nullptr, out_pose);
}

static int pose_host(fbus_ekf_t h, int M, const int32_t* ids, const void* pos, const void* quat, int what,
                     const uint8_t* mask, void* out_pose)
{
    const size_t es = esize(h), B = (size_t)h->B;
    const void *di, *dp, *dq, *dm;
    int rc;
    if ((rc = stage_in(h, 0, ids, B * M * 4, &di)) != FBUS_OK) return rc;
    if ((rc = stage_in(h, 1, pos, B * M * 3 * es, &dp)) != FBUS_OK) return rc;
    if ((rc = stage_in(h, 2, quat, B * M * 4 * es, &dq)) != FBUS_OK) return rc;
    if ((rc = stage_in(h, 3, mask, B, &dm)) != FBUS_OK) return rc;
    if (out_pose && (rc = ensure_stage(h, 4, B * 7 * es)) != FBUS_OK) return rc;
    if (out_pose) HIP_TRY(h, hipMemsetAsync(h->stage[4], 0, B * 7 * es, h->stream));
    if ((rc = do_pose_init(h, M, (const int32_t*)di, dp, dq, what, (const uint8_t*)dm, out_pose ? h->stage[4] : nullptr)) != FBUS_OK)
        return rc;
    if (out_pose) HIP_TRY(h, hipMemcpyAsync(out_pose, h->stage[4], B * 7 * es, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return FBUS_OK;
}

int fbus_ekf_pose_init(fbus_ekf_t h, int M, const int32_t* ids, const void* pos, const void* quat, int what,
                       const uint8_t* mask)
{
    DeviceGuard guard_(h);
    if (!h || !ids || !pos || !quat || M < 1 || M > FBUS_MAX_VISIBLE) return FBUS_ERR_INVALID;
    if (what != FBUS_POSE_INIT && what != FBUS_POSE_RESET) return FBUS_ERR_INVALID;
    return pose_host(h, M, ids, pos, quat, what, mask, nullptr);
}

int fbus_ekf_vision_only_pose(fbus_ekf_t h, int M, const int32_t* ids, const void* pos, const void* quat, void* out_pose)
{
    DeviceGuard guard_(h);
    if (!h || !ids || !pos || !quat || !out_pose || M < 1 || M > FBUS_MAX_VISIBLE) return FBUS_ERR_INVALID;
    return pose_host(h, M, ids, pos, quat, 2, nullptr, out_pose);
}

int fbus_ekf_imu_ema_dev(fbus_ekf_t h, int T, void* accel, void* gyro, int restart)
{
    DeviceGuard guard_(h);
    if (!h || T < 0 || (T > 0 && (!accel || !gyro))) return FBUS_ERR_INVALID;
    if (!h->d_ema_carry) HIP_TRY(h, hipMalloc(&h->d_ema_carry, (size_t)h->B * 6 * esize(h)));
    if (restart) h->ema_has_carry = false;
    if (T == 0) return FBUS_OK;
    const int grid = (h->B + 255) / 256;
    if (h->dtype == 32)
        hipLaunchKernelGGL((imu_ema_kernel<float>), dim3(grid), dim3(256), 0, h->stream, h->B, T, (float*)accel,
                           (float*)gyro, (float*)h->d_ema_carry, h->ema_has_carry ? 1 : 0);
    else
        hipLaunchKernelGGL((imu_ema_kernel<double>), dim3(grid), dim3(256), 0, h->stream, h->B, T, (double*)accel,
                           (double*)gyro, (double*)h->d_ema_carry, h->ema_has_carry ? 1 : 0);
    HIP_TRY(h, hipGetLastError());
    h->ema_has_carry = true;
    return FBUS_OK;
}

int fbus_ekf_imu_ema(fbus_ekf_t h, int T, void* accel, void* gyro, int restart)
{
    DeviceGuard guard_(h);
    if (!h || T < 0 || (T > 0 && (!accel || !gyro))) return FBUS_ERR_INVALID;
    if (T == 0) return fbus_ekf_imu_ema_dev(h, 0, nullptr, nullptr, restart);
    const size_t bytes = (size_t)T * h->B * 3 * esize(h);
    const void *da, *dg;
    int rc;
    if ((rc = stage_in(h, 0, accel, bytes, &da)) != FBUS_OK) return rc;
    if ((rc = stage_in(h, 1, gyro, bytes, &dg)) != FBUS_OK) return rc;
    if ((rc = fbus_ekf_imu_ema_dev(h, T, h->stage[0], h->stage[1], restart)) != FBUS_OK) return rc;
    HIP_TRY(h, hipMemcpyAsync(accel, h->stage[0], bytes, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipMemcpyAsync(gyro, h->stage[1], bytes, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return FBUS_OK;
}

int fbus_ekf_graph_begin(fbus_ekf_t h)
{
    DeviceGuard guard_(h);
    if (!h) return FBUS_ERR_INVALID;
    if (h->capturing) return fail(h, FBUS_ERR_INVALID, "graph_begin: already capturing");
    const int rc = flush_events(h);
    if (rc != FBUS_OK) return rc;
    HIP_TRY(h, hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
    h->capturing = true;
    return FBUS_OK;
}

int fbus_ekf_graph_end(fbus_ekf_t h, int* graph_id)
{
    DeviceGuard guard_(h);
    if (!h || !graph_id) return FBUS_ERR_INVALID;
    if (!h->capturing) return fail(h, FBUS_ERR_INVALID, "graph_end: not capturing");
    h->capturing = false;
    hipGraph_t g = nullptr;
    HIP_TRY(h, hipStreamEndCapture(h->stream, &g));
    hipGraphExec_t ex = nullptr;
    const hipError_t e = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    if (e != hipSuccess) return fail(h, FBUS_ERR_HIP, std::string("hipGraphInstantiate: ") + hipGetErrorString(e));
    h->graphs.push_back(ex);
    *graph_id = (int)h->graphs.size() - 1;
    return FBUS_OK;
}

int fbus_ekf_graph_launch(fbus_ekf_t h, int graph_id)
{
    DeviceGuard guard_(h);
    if (!h || graph_id < 0 || graph_id >= (int)h->graphs.size() || !h->graphs[graph_id]) return FBUS_ERR_INVALID;
    HIP_TRY(h, hipGraphLaunch(h->graphs[graph_id], h->stream));
    return FBUS_OK;
}

int fbus_ekf_graph_destroy(fbus_ekf_t h, int graph_id)
{
    DeviceGuard guard_(h);
    if (!h || graph_id < 0 || graph_id >= (int)h->graphs.size() || !h->graphs[graph_id]) return FBUS_ERR_INVALID;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    HIP_TRY(h, hipGraphExecDestroy(h->graphs[graph_id]));
    h->graphs[graph_id] = nullptr;
    return FBUS_OK;
}

int fbus_ekf_timing_enable(fbus_ekf_t h, int on)
{
    DeviceGuard guard_(h);
    if (!h) return FBUS_ERR_INVALID;
    if (!on) { const int rc = flush_events(h); if (rc != FBUS_OK) return rc; }
    h->timing = on != 0;
    h->timing_stride = on > 1 ? on : 1;
    return FBUS_OK;
}

int fbus_ekf_timing_reset(fbus_ekf_t h)
{
    DeviceGuard guard_(h);
    if (!h) return FBUS_ERR_INVALID;
    const int rc = flush_events(h);
    if (rc != FBUS_OK) return rc;
    for (int i = 0; i < FBUS_KERNEL_COUNT; ++i) { h->t_ms[i] = 0; h->t_n[i] = 0; }
    h->frame_count = 0;                 // the first frame after a reset is a sampled one, whatever the stride
    return FBUS_OK;
}

int fbus_ekf_timing_read(fbus_ekf_t h, int kernel, double* total_ms, int64_t* launches)
{
    DeviceGuard guard_(h);
    if (!h || kernel < 0 || kernel >= FBUS_KERNEL_COUNT) return FBUS_ERR_INVALID;
    const int rc = flush_events(h);
    if (rc != FBUS_OK) return rc;
    if (total_ms) *total_ms = h->t_ms[kernel];
    if (launches) *launches = h->t_n[kernel];
    return FBUS_OK;
}

int fbus_ekf_l0_eval(fbus_ekf_t h, int op, int n, const void* a, const void* b, void* out)
{
    DeviceGuard guard_(h);
    static const int wa[] = { 4, 4, 4, 4, 3, 3, 1 }, wb[] = { 4, 0, 0, 0, 1, 0, 0 }, wo[] = { 4, 9, 9, 4, 9, 4, 4 };
    if (!h || op < 0 || op > FBUS_L0_SINCOS_HALF || n < 1 || !a || !out || (wb[op] && !b)) return FBUS_ERR_INVALID;
    const size_t es = esize(h);
    const void *da, *db = nullptr;
    int rc;
    if ((rc = stage_in(h, 0, a, (size_t)n * wa[op] * es, &da)) != FBUS_OK) return rc;
    if (wb[op] && (rc = stage_in(h, 1, b, (size_t)n * wb[op] * es, &db)) != FBUS_OK) return rc;
    if ((rc = ensure_stage(h, 2, (size_t)n * wo[op] * es)) != FBUS_OK) return rc;
    const int grid = (n + 255) / 256;
    if (h->dtype == 32)
        hipLaunchKernelGGL((l0_eval_kernel<float>), dim3(grid), dim3(256), 0, h->stream, op, n, (const float*)da,
                           (const float*)db, (float*)h->stage[2]);
    else
        hipLaunchKernelGGL((l0_eval_kernel<double>), dim3(grid), dim3(256), 0, h->stream, op, n, (const double*)da,
                           (const double*)db, (double*)h->stage[2]);
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipMemcpyAsync(out, h->stage[2], (size_t)n * wo[op] * es, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return FBUS_OK;
}

}  // extern "C"
