// stoprule.hip -- the trainer's episode stop rule without its host sync, single- and multi-GPU.
//
// Reference: scripts/Poker/trainGPU.py:27-33,99 -- every 5th step `terminated.float().mean() > 0.8` is read back
// (a blocking device->host copy in the middle of the loop) and ends the episode.  Here the done tables of a chunk
// are counted on the device (per-wavefront counts stored by the chunk's last step launch, or a counting kernel over
// the caller's flags), summed and -- with one process per GPU -- all-reduced over the ranks on a SIDE stream (RCCL
// over xGMI: 8 bytes, latency-bound, never on the stream the steps run on), copied to pinned host memory, and the
// verdict of chunk c is taken after chunk c + lag has been enqueued.  The lag is FIXED (not "whatever has arrived"):
// every run, and every rank, ends every episode at the same step, so all ranks issue identical collective sequences.
#include <hip/hip_runtime.h>

#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>

#include <cerrno>
#include <chrono>
#include <cstdio>
#include <cstring>

#include "poker_device.h"

using pulse_dev::kBlock;

// ---------------------------------------------------------------- RCCL, bound at run time
// The library must load (and every white-box test must run) on a box without RCCL; the collective is resolved from
// the RCCL copy already in the process (PyTorch-ROCm's, so that one RCCL serves both) or the system one.
namespace {
struct NcclUniqueId { char internal[128]; };
typedef void* NcclComm;
struct Rccl {
    void* lib = nullptr;
    int (*GetUniqueId)(NcclUniqueId*) = nullptr;
    int (*CommInitRank)(NcclComm*, int, NcclUniqueId, int) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, NcclComm, hipStream_t) = nullptr;
    int (*CommDestroy)(NcclComm) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};
Rccl g_rccl;
constexpr int kNcclSum = 0, kNcclInt64 = 4;          // rccl.h: ncclSum, ncclInt64

int rccl_load() {
    if (g_rccl.lib) return 0;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    for (const char* n : names) if ((h = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;     // the copy torch loaded
    if (!h) for (const char* n : names) if ((h = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
    if (!h) return pulse::fail(PULSE_ENODEVICE, "pulse_comm: librccl.so not found");
    Rccl r; r.lib = h;
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
    r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(dlsym(h, "ncclAllReduce"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
    if (!r.GetUniqueId || !r.CommInitRank || !r.AllReduce || !r.CommDestroy || !r.GetErrorString)
        return pulse::fail(PULSE_ENODEVICE, "pulse_comm: librccl.so lacks an nccl* symbol");
    g_rccl = r;
    return 0;
}
int fail_nccl(int rc, const char* what) {
    char buf[256];
    std::snprintf(buf, sizeof buf, "%s: RCCL error %d (%s)", what, rc, g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "?");
    return pulse::fail(PULSE_ELAUNCH, buf);
}
}  // namespace

struct PulseComm { NcclComm comm; int rank, world; };

extern "C" {

int pulse_comm_unique_id(uint8_t* out128) {
    if (!out128) return pulse::fail(PULSE_EINVAL, "pulse_comm_unique_id: null argument");
    if (int rc = rccl_load()) return rc;
    NcclUniqueId id;
    if (int rc = g_rccl.GetUniqueId(&id)) return fail_nccl(rc, "ncclGetUniqueId");
    std::memcpy(out128, id.internal, sizeof id.internal);
    return 0;
}

int pulse_comm_create(const uint8_t* id128, int32_t rank, int32_t world, void** out) {
    if (!id128 || !out || world < 1 || rank < 0 || rank >= world) return pulse::fail(PULSE_EINVAL, "pulse_comm_create: bad argument");
    if (int rc = rccl_load()) return rc;
    NcclUniqueId id;
    std::memcpy(id.internal, id128, sizeof id.internal);
    PulseComm* c = new PulseComm{nullptr, rank, world};
    if (int rc = g_rccl.CommInitRank(&c->comm, world, id, rank)) { delete c; return fail_nccl(rc, "ncclCommInitRank"); }
    *out = c;
    return 0;
}

int pulse_comm_all_reduce_i64(void* comm, const int64_t* send, int64_t* recv, int32_t count, void* stream) {
    PulseComm* c = static_cast<PulseComm*>(comm);
    if (!c || !send || !recv || count < 0) return pulse::fail(PULSE_EINVAL, "pulse_comm_all_reduce_i64: bad argument");
    if (int rc = g_rccl.AllReduce(send, recv, (size_t)count, kNcclInt64, kNcclSum, c->comm, (hipStream_t)stream))
        return fail_nccl(rc, "ncclAllReduce");
    return 0;
}

int pulse_comm_destroy(void* comm) {
    PulseComm* c = static_cast<PulseComm*>(comm);
    if (!c) return 0;
    int rc = g_rccl.CommDestroy ? g_rccl.CommDestroy(c->comm) : 0;
    delete c;
    return rc ? fail_nccl(rc, "ncclCommDestroy") : 0;
}

}  // extern "C"

// ---------------------------------------------------------------- the rule
namespace {
constexpr int kSlots = 4;                 // check points in flight: lag < kSlots
constexpr int kMaxPartials = 1024;        // workgroups of the flag-counting kernel
constexpr int kVerdictCopies = 64, kVerdictStride = 16;     // a relayed verdict word: 64 copies, one 128-byte line each (poker_step.hip)

// What the host polls, in coherent pinned memory: the counts of one check point, then its sequence number (written
// last, behind a system-scope fence) -- the host learns of a result a microsecond after the kernel wrote it, without
// an event wait's wake-up (tens of microseconds).
struct Published { long long local, global, seq, pad; };

// sum + publish as a kernel of its own: the flush of a check point no later launch will carry (lag 0, the trainer's
// flag counts), and the first half of the RCCL path (host == nullptr: the device pair only)
__global__ __launch_bounds__(kBlock) void stoprule_sum_kernel(const uint32_t* __restrict__ partials, int n, long long* __restrict__ pair,
                                                              Published* host, long long seq) {
    pulse_dev::sum_and_publish(partials, n, pair, reinterpret_cast<long long*>(host), seq);
}
__global__ void stoprule_publish_kernel(const long long* __restrict__ pair, Published* host, long long seq) {
    host->local = pair[0]; host->global = pair[1];
    __threadfence_system();
    *reinterpret_cast<volatile long long*>(&host->seq) = seq;
    __threadfence_system();
}
// set flags of a caller-owned bool tensor (the trainer's `terminated`), one partial per workgroup
__global__ __launch_bounds__(kBlock) void stoprule_flags_kernel(const uint8_t* __restrict__ flags, int n, uint32_t* __restrict__ partials) {
    int c = 0;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) c += flags[i] != 0;
    for (int m = 32; m >= 1; m >>= 1) c += __shfl_xor(c, m);
    __shared__ int w[kBlock / 64];
    if ((threadIdx.x & 63) == 0) w[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) { int t = 0; for (int k = 0; k < kBlock / 64; ++k) t += w[k]; partials[blockIdx.x] = (uint32_t)t; }
}

// One cache line per (rank, slot) in a POSIX shared-memory segment: how the ranks of one node tell each other their
// counts.  8 bytes per rank every 5 steps: a store and a few loads of host memory, no device involved.
struct ShmRecord { long long count, seq, pad[6]; };
}  // namespace

// The exchange as an object of its own (host code only: usable, and tested, without a GPU).  all_sum(index, value):
// every rank calls it with the same sequence of indices; returns the sum of the ranks' values for that index.
// ... followed by one 64-byte record per rank naming the GPU the rank computes on (PCI bus id; empty = not told yet)
struct ShmDevice { char bus_id[56]; long long set; };
struct PulseShm { ShmRecord* mem; size_t bytes; int rank, world; ShmDevice* devices; };

extern "C" {

int pulse_shm_create(const char* name, int32_t rank, int32_t world, void** out) {
    if (!name || !out || world < 1 || rank < 0 || rank >= world) return pulse::fail(PULSE_EINVAL, "pulse_shm_create: bad argument");
    PulseShm* h = new PulseShm{nullptr, (size_t)world * kSlots * sizeof(ShmRecord) + (size_t)world * sizeof(ShmDevice), rank, world, nullptr};
    // Whoever comes first creates the segment (O_EXCL: a fresh, zero-filled one, seq 0 = nothing published); the others
    // open it.  A stale segment of that name (a crashed job whose rank-0 pid was reused) would carry old sequence
    // numbers: the host code removes the name before it tells the ranks (stoprule.py: _shared_name), and the name
    // carries a time stamp besides the pid.
    int fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd < 0 && errno == EEXIST) fd = shm_open(name, O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, (off_t)h->bytes) != 0) { if (fd >= 0) close(fd); delete h; return pulse::fail(PULSE_EINTERNAL, "pulse_shm_create: shm_open / ftruncate failed"); }
    void* m = mmap(nullptr, h->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (m == MAP_FAILED) { delete h; return pulse::fail(PULSE_EINTERNAL, "pulse_shm_create: mmap failed"); }
    h->mem = static_cast<ShmRecord*>(m);             // a fresh segment is zero-filled: seq 0 = nothing published
    h->devices = reinterpret_cast<ShmDevice*>(h->mem + (size_t)world * kSlots);
    *out = h;
    return 0;
}

/* Names the GPU this rank computes on (any string that is equal exactly for ranks on the same device: the PCI bus id). */
int pulse_shm_set_device(void* handle, const char* bus_id) {
    PulseShm* h = static_cast<PulseShm*>(handle);
    if (!h || !bus_id) return pulse::fail(PULSE_EINVAL, "pulse_shm_set_device: null argument");
    ShmDevice* d = h->devices + h->rank;
    std::memset(d->bus_id, 0, sizeof d->bus_id);
    std::strncpy(d->bus_id, bus_id, sizeof d->bus_id - 1);
    __atomic_store_n(&d->set, 1ll, __ATOMIC_RELEASE);
    return 0;
}

/* 1: every rank has named its GPU and no other rank shares this rank's; 0: another rank computes on the same GPU (or a rank
 * has not named its own within wait_ms) -- launches that wait for each other's hosts must then not be used. */
int pulse_shm_device_is_private(void* handle, int32_t wait_ms) {
    PulseShm* h = static_cast<PulseShm*>(handle);
    if (!h) return pulse::fail(PULSE_EINVAL, "pulse_shm_device_is_private: null argument");
    const ShmDevice* mine = h->devices + h->rank;
    if (!__atomic_load_n(&mine->set, __ATOMIC_ACQUIRE)) return 0;
    const auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < h->world; ++r) {
        if (r == h->rank) continue;
        const ShmDevice* d = h->devices + r;
        while (!__atomic_load_n(&d->set, __ATOMIC_ACQUIRE)) {
            if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(wait_ms)) return 0;
            usleep(200);
        }
        if (std::strncmp(d->bus_id, mine->bus_id, sizeof d->bus_id) == 0) return 0;
    }
    return 1;
}

int pulse_shm_all_sum(void* handle, int64_t index, int64_t value, int64_t* total) {
    PulseShm* h = static_cast<PulseShm*>(handle);
    if (!h || !total || index < 0) return pulse::fail(PULSE_EINVAL, "pulse_shm_all_sum: bad argument");
    const int slot = (int)(index % kSlots);
    ShmRecord* mine = h->mem + (size_t)h->rank * kSlots + slot;
    // the slot's previous use (index - kSlots) has been read by every rank: each of them has since published index - kSlots + 1 .. index - 1
    // only after reading it, and this rank has read those -- kSlots >= 2 suffices for that argument
    mine->count = value;
    __atomic_store_n(&mine->seq, (long long)index + 1, __ATOMIC_RELEASE);
    long long sum = 0;
    const auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < h->world; ++r) {
        ShmRecord* rec = h->mem + (size_t)r * kSlots + slot;
        for (long long spins = 1; __atomic_load_n(&rec->seq, __ATOMIC_ACQUIRE) != (long long)index + 1; ++spins) {
            __builtin_ia32_pause();
            if ((spins & 0xFFFFF) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120))
                return pulse::fail(PULSE_EINTERNAL, "pulse_shm_all_sum: a rank did not publish its value within 120 s");
        }
        sum += rec->count;
    }
    *total = sum;
    return 0;
}

int pulse_shm_destroy(void* handle) {
    PulseShm* h = static_cast<PulseShm*>(handle);
    if (!h) return 0;
    munmap(h->mem, h->bytes);
    delete h;
    return 0;
}

}  // extern "C"

enum { kModeLocal = 0, kModeRccl = 1, kModeShm = 2 };

struct PulseStopRule {
    int mode;
    uint32_t* partials_dev;               // [kSlots][max_partials]
    int n_partials[kSlots];
    Published* host;                      // [kSlots], coherent pinned memory the publishing kernels write into
    long long* pair_dev;                  // [kSlots][2] = {local count, global count} (RCCL path / flush scratch)
    int max_partials;
    long long submitted, epoch_first;     // check points submitted so far / first one of the current episode
    long long scheduled;                  // check points whose publication has been enqueued (a launch carries it, or a flush did)
    hipStream_t last_stream;              // where the newest check point's counts were written (a flush goes there)
    int n_local; long long n_global; double threshold; int lag;
    // RCCL exchange: side stream + events
    PulseComm* comm; hipStream_t side; hipEvent_t ready[kSlots], copied[kSlots];
    // shared-memory exchange
    PulseShm* shm; int rank, world;
    long long side_launches;              // check points that went through the side stream (sum -> all-reduce -> publish)
    // paired launches (pulse_internal.h: StopRulePair)
    long long* verdict_host;              // [kSlots + 1] pinned: {launch id << 8 | flags}, written by the host; the last word: a launch's "gave up" mark
    long long* verdict_dev;               // [kSlots] device: the same word, relayed by thread 0 of the launch
    long long launches;                   // paired launches issued so far (ids start at 1)
    long long verdicts_known;             // check points below this index have been read and did not end the episode ...
    bool over_known;                      // ... unless this is set: check point verdicts_known - 1 did
    long long global_known[kSlots], global_known_idx[kSlots];   // job-wide counts already summed over the ranks (never summed twice)
    // a paired launch waits for its host's verdict at most verdict_wait_ticks (100 MHz); a host that finds itself later than
    // half of that lets the launch give up (it runs nothing), counts a time-out and goes on with one check interval per launch
    long long verdict_wait_ticks;
    long long verdict_timeouts;
    bool pairs_off;                       // set by a time-out, for the life of the handle
    int device_private;                   // shm exchange: -1 = not asked yet, 0 = another rank shares this GPU (no pairs), 1 = no
    bool allow_shared_device_pairs;
    int debug_late;                       // test hook: that many coming verdicts are treated as "host too late"
    std::chrono::steady_clock::time_point pair_enqueued;
};

namespace {
int wait_published(PulseStopRule* h, long long c) {
    const int slot = (int)(c % kSlots);
    volatile long long* seq = &h->host[slot].seq;
    if (*seq == c + 1) { __atomic_thread_fence(__ATOMIC_ACQUIRE); return 0; }
    const auto t0 = std::chrono::steady_clock::now();
    for (long long spins = 1; *seq != c + 1; ++spins) {
        __builtin_ia32_pause();
        if ((spins & 0xFFFFF) == 0) {                       // now and then: is the device still alive?
            const hipError_t q = hipStreamQuery(h->mode == kModeRccl ? h->side : h->last_stream);
            if (q != hipSuccess && q != hipErrorNotReady) return pulse::fail_hip((int)q, "stop rule: the stream that publishes the count");
            if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(60))
                return pulse::fail(PULSE_EINTERNAL, "stop rule: no count after 60 s");
        }
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    return 0;
}

// check point c has no later launch to carry its publication (lag 0, trainer flags): sum + publish it with a kernel of its own
int flush(PulseStopRule* h, long long c) {
    const int slot = (int)(c % kSlots);
    hipLaunchKernelGGL(stoprule_sum_kernel, dim3(1), dim3(kBlock), 0, h->last_stream, h->partials_dev + (size_t)slot * h->max_partials,
                       h->n_partials[slot], h->pair_dev + 2 * slot, h->host + slot, c + 1);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return pulse::fail_hip((int)e, "stop rule: flush");
    h->scheduled = c + 1;
    return 0;
}

}  // namespace

namespace pulse {

// Slot the check point about to be submitted will use, and -- if the previous check point has not been scheduled for
// publication yet -- what the launch about to be enqueued shall sum and publish on its way (workgroup 0 of it, first thing).
int stoprule_claim(PulseStopRule* h, int n_partials, uint32_t** partials_out, StopRuleCarry* carry) {
    if (n_partials <= 0 || n_partials > h->max_partials) return fail(PULSE_EINVAL, "stop rule: too many partial counts for this handle");
    const int slot = (int)(h->submitted % kSlots);
    // the slot's previous occupant (kSlots check points ago) must have been consumed: its verdict was due lag + 1 <= kSlots submissions ago
    if (!partials_out || !carry) return fail(PULSE_EINVAL, "stop rule: null argument");
    *partials_out = h->partials_dev + (size_t)slot * h->max_partials;
    *carry = StopRuleCarry{nullptr, 0, nullptr, 0};
    const long long prev = h->submitted - 1;
    if (h->mode != kModeRccl && prev >= 0 && h->scheduled < prev + 1) {
        const int ps = (int)(prev % kSlots);
        *carry = StopRuleCarry{h->partials_dev + (size_t)ps * h->max_partials, h->n_partials[ps], reinterpret_cast<long long*>(h->host + ps), prev + 1};
        h->scheduled = prev + 1;
    }
    return 0;
}

// The partial counts of the claimed slot are (being) written in `st` order.
int stoprule_commit(PulseStopRule* h, int n_partials, hipStream_t st) {
    const int slot = (int)(h->submitted % kSlots);
    h->n_partials[slot] = n_partials;
    h->last_stream = st;
    if (h->mode == kModeRccl) {            // sum -> all-reduce -> publish on the side stream (an event hands the chunk over)
        hipError_t e = hipEventRecord(h->ready[slot], st);
        if (e == hipSuccess) e = hipStreamWaitEvent(h->side, h->ready[slot], 0);
        if (e != hipSuccess) return fail_hip((int)e, "stop rule: event hand-off");
        long long* pair = h->pair_dev + 2 * slot;
        hipLaunchKernelGGL(stoprule_sum_kernel, dim3(1), dim3(kBlock), 0, h->side, h->partials_dev + (size_t)slot * h->max_partials, n_partials, pair,
                           (Published*)nullptr, 0ll);
        if (int rc = pulse_comm_all_reduce_i64(h->comm, reinterpret_cast<const int64_t*>(pair), reinterpret_cast<int64_t*>(pair + 1), 1, h->side)) return rc;
        hipLaunchKernelGGL(stoprule_publish_kernel, dim3(1), dim3(1), 0, h->side, pair, h->host + slot, h->submitted + 1);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipEventRecord(h->copied[slot], h->side);
        if (e != hipSuccess) return fail_hip((int)e, "stop rule: publish to host");
        h->scheduled = h->submitted + 1;
        ++h->side_launches;
    }
    ++h->submitted;
    return 0;
}


// ---- paired launches (pulse_internal.h: StopRulePair)
// Ranks that share ONE device must not pair (shm exchange): a paired launch waits for its host, the host for every rank's
// launch to have STARTED (their first workgroups publish the counts it needs), and the grids of several processes need not
// fit the device together -- the resident wavefronts of one rank would spin while the other rank's first workgroup cannot be
// scheduled.  One check interval per launch never waits inside a kernel.
bool stoprule_pairs_supported(const PulseStopRule* h_, int n_partials) {
    PulseStopRule* h = const_cast<PulseStopRule*>(h_);
    if (!(h && h->lag == 1 && h->mode != kModeRccl && !h->pairs_off && n_partials > 0 && n_partials <= h->max_partials)) return false;
    if (h->mode == kModeShm && !h->allow_shared_device_pairs) {
        if (h->device_private < 0) h->device_private = pulse_shm_device_is_private(h->shm, 20000) == 1 ? 1 : 0;
        if (!h->device_private) return false;
    }
    return true;
}

namespace {
// the count of check point c (this episode's), job-wide; reads it once per check point, in index order
int pair_count(PulseStopRule* h, long long c, bool* over) {
    if (c < h->verdicts_known) { *over = false; return 0; }            // read before: it did not end the episode (or we would not be here)
    if (int rc = wait_published(h, c)) return rc;
    const int slot = (int)(c % kSlots);
    long long glob = h->host[slot].global;
    if (h->mode == kModeShm) { int64_t total = 0; if (int rc = pulse_shm_all_sum(h->shm, c, h->host[slot].local, &total)) return rc; glob = total; }
    h->global_known[slot] = glob; h->global_known_idx[slot] = c;
    *over = (double)glob > h->threshold * (double)h->n_global;
    h->verdicts_known = c + 1; h->over_known = *over;
    return 0;
}
}  // namespace

int stoprule_pair_claim(PulseStopRule* h, int n_partials, int n_chunks, StopRulePair* plan) {
    if (!plan || n_chunks < 1 || n_chunks > 2 || !stoprule_pairs_supported(h, n_partials)) return fail(PULSE_EINVAL, "stop rule: paired launch not possible with this handle");
    if (const long long gave_up = __atomic_load_n(h->verdict_host + kSlots, __ATOMIC_ACQUIRE)) {
        char msg[160];
        std::snprintf(msg, sizeof msg, "stop rule: paired launch %lld waited 20 s for its verdict and ran nothing (a rank of the job is gone or stalled)", gave_up);
        return fail(PULSE_EINTERNAL, msg);
    }
    if (h->over_known) return 1;                                       // a count read earlier already ended the episode
    const long long a = h->submitted;
    *plan = StopRulePair{};
    plan->first_check_point = a; plan->n_chunks = n_chunks;
    plan->launch_id = ++h->launches;
    plan->verdict_host = h->verdict_host + (plan->launch_id % kSlots);
    plan->verdict_dev = h->verdict_dev + (plan->launch_id % kSlots) * kVerdictCopies * kVerdictStride;
    plan->verdict_err = h->verdict_host + kSlots;
    plan->wait_ticks = h->verdict_wait_ticks;
    uint32_t* first = h->partials_dev + (size_t)(a % kSlots) * h->max_partials;
    uint32_t* second = h->partials_dev + (size_t)((a + 1) % kSlots) * h->max_partials;
    plan->wave_done_mid = n_chunks == 2 ? first : nullptr;
    plan->wave_done_fin = n_chunks == 2 ? second : first;
    // carries: this episode's check points a - 2, a - 1 whose publication nobody has enqueued yet
    int k = 0;
    for (long long c = a - 2; c < a; ++c) {
        if (c < h->epoch_first || c < h->scheduled) continue;
        const int ps = (int)(c % kSlots);
        plan->carry[k++] = StopRuleCarry{h->partials_dev + (size_t)ps * h->max_partials, h->n_partials[ps], reinterpret_cast<long long*>(h->host + ps), c + 1};
    }
    if (h->scheduled < a) h->scheduled = a;
    return 0;
}

int stoprule_pair_commit(PulseStopRule* h, const StopRulePair* plan, int n_partials, hipStream_t st) {
    for (int i = 0; i < plan->n_chunks; ++i) h->n_partials[(plan->first_check_point + i) % kSlots] = n_partials;
    h->last_stream = st;
    h->submitted = plan->first_check_point + plan->n_chunks;
    h->pair_enqueued = std::chrono::steady_clock::now();
    return 0;
}

// The verdict of one paired launch.  *gave_up = 1: the host was too late for the launch (see below), which therefore ran
// NOTHING and left no trace; the handle has been set back to before the launch and no longer pairs -- the caller runs the
// same steps with one check interval per launch (`over` still tells whether the episode had ended before them).
int stoprule_pair_verdict(PulseStopRule* h, const StopRulePair* plan, int* chunks_run, int* over, int* gave_up) {
    const long long a = plan->first_check_point;
    bool skip = false, stop = false;
    *gave_up = 0;
    if (a - 2 >= h->epoch_first) if (int rc = pair_count(h, a - 2, &skip)) return rc;
    if (!skip && a - 1 >= h->epoch_first) if (int rc = pair_count(h, a - 1, &stop)) return rc;
    // The launch waits wait_ticks (of its 100 MHz clock) from its start for this word, then gives up.  Who decides whether
    // it got it?  The host, by its own clock, BEFORE writing: the launch cannot have started before it was enqueued, so up
    // to half the wait after the enqueue the word certainly arrives in time and is written; later than that (a rank of the
    // job stalled, this process was stopped) the word is NOT written and the host waits for the launch's give-up mark.
    // Either way both sides agree on what ran -- no word and mark crossing each other.
    const double wait_s = (double)plan->wait_ticks * 1e-8;
    const double late_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - h->pair_enqueued).count();
    bool late = late_s > 0.5 * wait_s;
    if (h->debug_late > 0) { --h->debug_late; late = true; }
    if (late) {
        const auto t0 = std::chrono::steady_clock::now();
        while (__atomic_load_n(plan->verdict_err, __ATOMIC_ACQUIRE) != plan->launch_id) {
            usleep(100);
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 2.0 * wait_s + 5.0)
                return fail(PULSE_EINTERNAL, "stop rule: a paired launch neither took its verdict nor gave up (is the device gone?)");
        }
        __atomic_store_n(plan->verdict_err, 0ll, __ATOMIC_RELEASE);
        ++h->verdict_timeouts; h->pairs_off = true;
        h->submitted = a;                                   // the launch's check points were never counted
        *chunks_run = 0; *over = skip ? 1 : 0; *gave_up = 1;
        return 0;
    }
    // (stop_mid only means something to a launch of two chunks; a one-chunk launch whose chunk is the episode's last runs it)
    const long long word = (plan->launch_id << 8) | (skip ? 1 : 0) | (stop && plan->n_chunks == 2 ? 2 : 0);
    __atomic_store_n(const_cast<long long*>(plan->verdict_host), word, __ATOMIC_RELEASE);
    *chunks_run = skip ? 0 : (stop ? 1 : plan->n_chunks);
    *over = (skip || stop) ? 1 : 0;
    return 0;
}

}  // namespace pulse

extern "C" {

int pulse_stoprule_create(int32_t n_local, int64_t n_global, double threshold, int32_t lag, void* comm, const char* shm_name,
                          int32_t rank, int32_t world, void** out) {
    if (!out || n_local < 0 || n_global < n_local || lag < 0 || lag >= kSlots - 1 || world < 1 || rank < 0 || rank >= world || (comm && shm_name))
        return pulse::fail(PULSE_EINVAL, "pulse_stoprule_create: need 0 <= n_local <= n_global, 0 <= lag < 3, 0 <= rank < world, at most one of comm / shm_name");
    PulseStopRule* h = new PulseStopRule();
    std::memset(h, 0, sizeof *h);
    h->n_local = n_local; h->n_global = n_global; h->threshold = threshold; h->lag = lag;
    h->verdict_wait_ticks = 2000000000ll;                   // 20 s
    h->device_private = -1;
    for (int i = 0; i < kSlots; ++i) h->global_known_idx[i] = -1;
    h->comm = static_cast<PulseComm*>(comm); h->rank = rank; h->world = world;
    // a communicator is honoured whatever its size: a world of one is a valid all-reduce, and it is what a one-GPU box
    // can run of the side-stream path (event hand-over, sum, ncclAllReduce, publish)
    h->mode = h->comm ? kModeRccl : (shm_name && world > 1) ? kModeShm : kModeLocal;
    h->max_partials = (int)(((long long)n_local * 4 + 63) / 64) + 4;          // one per wavefront of a step launch (4 lanes per table at most)
    if (h->max_partials < kMaxPartials) h->max_partials = kMaxPartials;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&h->partials_dev), (size_t)kSlots * h->max_partials * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&h->pair_dev), kSlots * 2 * sizeof(long long));
    if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void**>(&h->host), kSlots * sizeof(Published), hipHostMallocCoherent | hipHostMallocMapped);
    if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void**>(&h->verdict_host), (kSlots + 1) * sizeof(long long), hipHostMallocCoherent | hipHostMallocMapped);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&h->verdict_dev), kSlots * kVerdictCopies * kVerdictStride * sizeof(long long));
    if (e == hipSuccess) e = hipMemset(h->verdict_dev, 0, kSlots * kVerdictCopies * kVerdictStride * sizeof(long long));
    if (e == hipSuccess && h->mode == kModeRccl) {
        e = hipStreamCreateWithFlags(&h->side, hipStreamNonBlocking);
        for (int i = 0; i < kSlots && e == hipSuccess; ++i) {
            e = hipEventCreateWithFlags(&h->ready[i], hipEventDisableTiming);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&h->copied[i], hipEventDisableTiming);
        }
    }
    if (e != hipSuccess) {                      // release whatever was created (destroy tolerates the missing pieces)
        const int code = pulse::fail_hip((int)e, "pulse_stoprule_create");
        (void)hipGetLastError();
        h->submitted = 0; h->last_stream = nullptr;
        (void)pulse_stoprule_destroy(h);
        return code;
    }
    std::memset(h->host, 0, kSlots * sizeof(Published));
    std::memset(h->verdict_host, 0, (kSlots + 1) * sizeof(long long));
    if (h->mode == kModeShm) {
        void* shm = nullptr;
        if (int rc = pulse_shm_create(shm_name, rank, world, &shm)) { (void)pulse_stoprule_destroy(h); return rc; }
        h->shm = static_cast<PulseShm*>(shm);
        int dev = 0; char bus[64] = {0};
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetPCIBusId(bus, (int)sizeof bus, dev) != hipSuccess) {
            (void)hipGetLastError();
            std::snprintf(bus, sizeof bus, "unknown-device");     // equal for every rank that could not tell: treated as shared
        }
        (void)pulse_shm_set_device(shm, bus);
    }
    *out = h;
    return 0;
}

int pulse_stoprule_destroy(void* handle) {
    PulseStopRule* h = static_cast<PulseStopRule*>(handle);
    if (!h) return 0;
    if (h->mode == kModeRccl) {
        if (h->side) (void)hipStreamSynchronize(h->side);
        for (int i = 0; i < kSlots; ++i) { if (h->ready[i]) (void)hipEventDestroy(h->ready[i]); if (h->copied[i]) (void)hipEventDestroy(h->copied[i]); }
        if (h->side) (void)hipStreamDestroy(h->side);
    } else if (h->last_stream || h->submitted) {
        (void)hipStreamSynchronize(h->last_stream);       // a launch may still be about to write into the pinned block
    }
    if (h->shm) (void)pulse_shm_destroy(h->shm);
    if (h->partials_dev) (void)hipFree(h->partials_dev);
    if (h->pair_dev) (void)hipFree(h->pair_dev);
    if (h->host) (void)hipHostFree(h->host);
    if (h->verdict_host) (void)hipHostFree(h->verdict_host);
    if (h->verdict_dev) (void)hipFree(h->verdict_dev);
    delete h;
    return 0;
}

int pulse_stoprule_mode(void* handle) {
    PulseStopRule* h = static_cast<PulseStopRule*>(handle);
    if (!h) return pulse::fail(PULSE_EINVAL, "pulse_stoprule_mode: null argument");
    return h->mode;
}

int pulse_stoprule_set_option(void* handle, int32_t option, int64_t value) {
    PulseStopRule* h = static_cast<PulseStopRule*>(handle);
    if (!h) return pulse::fail(PULSE_EINVAL, "pulse_stoprule_set_option: null handle");
    switch (option) {
    case PULSE_STOPRULE_OPT_VERDICT_WAIT_TICKS:
        if (value < 100000) return pulse::fail(PULSE_EINVAL, "pulse_stoprule_set_option: the verdict wait must be at least 1 ms (100,000 ticks)");
        h->verdict_wait_ticks = value; return 0;
    case PULSE_STOPRULE_OPT_ALLOW_SHARED_DEVICE_PAIRS: h->allow_shared_device_pairs = value != 0; return 0;
    case PULSE_STOPRULE_OPT_DEBUG_LATE_VERDICTS: h->debug_late = (int)value; return 0;
    default: return pulse::fail(PULSE_EINVAL, "pulse_stoprule_set_option: unknown option");
    }
}

int pulse_stoprule_stats(void* handle, int64_t* out4) {
    PulseStopRule* h = static_cast<PulseStopRule*>(handle);
    if (!h || !out4) return pulse::fail(PULSE_EINVAL, "pulse_stoprule_stats: null argument");
    out4[0] = h->launches; out4[1] = h->verdict_timeouts;
    out4[2] = pulse::stoprule_pairs_supported(h, 1) ? 1 : 0;
    out4[3] = h->side_launches;
    return 0;
}

int64_t pulse_stoprule_side_launches(void* handle) {
    PulseStopRule* h = static_cast<PulseStopRule*>(handle);
    return h ? h->side_launches : -1;
}

int pulse_stoprule_submit(void* handle, const uint8_t* flags, int32_t n, void* stream) {
    PulseStopRule* h = static_cast<PulseStopRule*>(handle);
    if (!h || !flags || n < 0) return pulse::fail(PULSE_EINVAL, "pulse_stoprule_submit: bad argument");
    const int grid = n == 0 ? 1 : min(kMaxPartials, (n + kBlock - 1) / kBlock);
    uint32_t* partials = nullptr;
    pulse::StopRuleCarry none;
    if (h->mode != kModeRccl && h->submitted > 0 && h->scheduled < h->submitted)         // an unpublished roll-out check point: flush it first
        if (int rc = flush(h, h->submitted - 1)) return rc;
    if (int rc = pulse::stoprule_claim(h, grid, &partials, &none)) return rc;
    hipLaunchKernelGGL(stoprule_flags_kernel, dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, flags, n, partials);
    if (int rc = pulse::stoprule_commit(h, grid, (hipStream_t)stream)) return rc;
    if (h->mode != kModeRccl) return flush(h, h->submitted - 1);      // no later launch of ours will carry it
    return 0;
}

int pulse_stoprule_counts(void* handle, int64_t* local, int64_t* global, int32_t* have) {
    PulseStopRule* h = static_cast<PulseStopRule*>(handle);
    if (!h || !local || !global || !have) return pulse::fail(PULSE_EINVAL, "pulse_stoprule_counts: null argument");
    const long long c = h->submitted - 1 - h->lag;          // the check point whose verdict is due now
    *have = 0; *local = 0; *global = 0;
    if (c < h->epoch_first) return 0;
    if (h->scheduled < c + 1) if (int rc = flush(h, c)) return rc;       // lag 0: nothing else will publish it
    if (int rc = wait_published(h, c)) return rc;
    const int slot = (int)(c % kSlots);
    long long loc = h->host[slot].local, glob = h->host[slot].global;
    if (h->global_known_idx[slot] == c) glob = h->global_known[slot];       // summed before (a paired launch's verdict)
    else if (h->mode == kModeShm) { int64_t total = 0; if (int rc = pulse_shm_all_sum(h->shm, c, loc, &total)) return rc; glob = total; }
    h->global_known[slot] = glob; h->global_known_idx[slot] = c;
    *local = loc; *global = glob; *have = 1;
    return 0;
}

int pulse_stoprule_publish(void* handle) {
    PulseStopRule* h = static_cast<PulseStopRule*>(handle);
    if (!h) return pulse::fail(PULSE_EINVAL, "pulse_stoprule_publish: null argument");
    if (h->mode == kModeRccl || h->submitted == 0 || h->scheduled >= h->submitted) return 0;     // nothing unpublished (RCCL: the side stream does it)
    return flush(h, h->submitted - 1);
}

int pulse_stoprule_decide(void* handle, int32_t* over) {
    PulseStopRule* h = static_cast<PulseStopRule*>(handle);
    if (!over) return pulse::fail(PULSE_EINVAL, "pulse_stoprule_decide: null argument");
    int64_t local = 0, global = 0; int32_t have = 0;
    if (int rc = pulse_stoprule_counts(handle, &local, &global, &have)) return rc;
    *over = (have && (double)global > h->threshold * (double)h->n_global) ? 1 : 0;
    return 0;
}

int pulse_stoprule_drain(void* handle) {
    PulseStopRule* h = static_cast<PulseStopRule*>(handle);
    if (!h) return pulse::fail(PULSE_EINVAL, "pulse_stoprule_drain: null argument");
    h->epoch_first = h->submitted;           // check points of the finished episode never decide anything again
    h->verdicts_known = h->submitted; h->over_known = false;
    if (h->mode != kModeRccl && h->scheduled < h->submitted) h->scheduled = h->submitted;     // ... and need no publication
    return 0;
}

}  // extern "C"
