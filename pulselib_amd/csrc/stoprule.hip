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
constexpr int kSlots = 4;                 // chunks in flight: lag < kSlots
constexpr int kMaxPartials = 1024;        // workgroups of the flag-counting kernel

// partial counts of one chunk -> {local, local} (pair[1] is overwritten by the all-reduce when there is one)
__global__ __launch_bounds__(kBlock) void stoprule_sum_kernel(const uint32_t* __restrict__ partials, int n, long long* __restrict__ pair) {
    long long s = 0;
    for (int i = threadIdx.x; i < n; i += kBlock) s += partials[i];
    for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m);
    __shared__ long long w[kBlock / 64];
    if ((threadIdx.x & 63) == 0) w[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) { long long t = 0; for (int k = 0; k < kBlock / 64; ++k) t += w[k]; pair[0] = t; pair[1] = t; }
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
}  // namespace

struct PulseStopRule {
    hipStream_t side;
    hipEvent_t ready[kSlots], copied[kSlots];
    uint32_t* partials_dev;               // [kSlots][max_partials]
    long long* pair_dev;                  // [kSlots][2] = {local count, global count}
    long long* pair_host;                 // pinned, same shape
    int max_partials;
    long long submitted, epoch_first;     // chunks submitted so far / first chunk of the current episode
    int n_local; long long n_global; double threshold; int lag;
    PulseComm* comm;
};

namespace pulse {

// Slot the chunk about to be submitted will use; its previous occupant (kSlots chunks ago) must have landed.
int stoprule_claim(PulseStopRule* h, int n_partials, uint32_t** partials_out) {
    if (n_partials <= 0 || n_partials > h->max_partials) return fail(PULSE_EINVAL, "stop rule: too many partial counts for this handle");
    const int slot = (int)(h->submitted % kSlots);
    if (h->submitted >= kSlots) {
        const hipError_t e = hipEventSynchronize(h->copied[slot]);
        if (e != hipSuccess) return fail_hip((int)e, "stop rule: hipEventSynchronize");
    }
    *partials_out = h->partials_dev + (size_t)slot * h->max_partials;
    return 0;
}

// The partial counts of the claimed slot are (being) written in `st` order: sum, all-reduce and copy them out on the side stream.
int stoprule_commit(PulseStopRule* h, int n_partials, hipStream_t st) {
    const int slot = (int)(h->submitted % kSlots);
    hipError_t e = hipEventRecord(h->ready[slot], st);
    if (e == hipSuccess) e = hipStreamWaitEvent(h->side, h->ready[slot], 0);
    if (e != hipSuccess) return fail_hip((int)e, "stop rule: event hand-off");
    long long* pair = h->pair_dev + 2 * slot;
    hipLaunchKernelGGL(stoprule_sum_kernel, dim3(1), dim3(kBlock), 0, h->side, h->partials_dev + (size_t)slot * h->max_partials, n_partials, pair);
    if (h->comm && h->comm->world > 1)
        if (int rc = pulse_comm_all_reduce_i64(h->comm, reinterpret_cast<const int64_t*>(pair), reinterpret_cast<int64_t*>(pair + 1), 1, h->side)) return rc;
    e = hipMemcpyAsync(h->pair_host + 2 * slot, pair, 2 * sizeof(long long), hipMemcpyDeviceToHost, h->side);
    if (e == hipSuccess) e = hipEventRecord(h->copied[slot], h->side);
    if (e != hipSuccess) return fail_hip((int)e, "stop rule: copy to host");
    ++h->submitted;
    return 0;
}

}  // namespace pulse

extern "C" {

int pulse_stoprule_create(int32_t n_local, int64_t n_global, double threshold, int32_t lag, void* comm, void** out) {
    if (!out || n_local < 0 || n_global < n_local || lag < 0 || lag >= kSlots)
        return pulse::fail(PULSE_EINVAL, "pulse_stoprule_create: need 0 <= n_local <= n_global and 0 <= lag < 4");
    PulseStopRule* h = new PulseStopRule();
    h->n_local = n_local; h->n_global = n_global; h->threshold = threshold; h->lag = lag;
    h->comm = static_cast<PulseComm*>(comm);
    h->max_partials = (int)(((long long)n_local * 4 + 63) / 64) + 4;          // one per wavefront of a step launch (4 lanes per table)
    if (h->max_partials < kMaxPartials) h->max_partials = kMaxPartials;
    hipError_t e = hipStreamCreateWithFlags(&h->side, hipStreamNonBlocking);
    for (int i = 0; i < kSlots && e == hipSuccess; ++i) {
        e = hipEventCreateWithFlags(&h->ready[i], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&h->copied[i], hipEventDisableTiming);
    }
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&h->partials_dev), (size_t)kSlots * h->max_partials * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&h->pair_dev), kSlots * 2 * sizeof(long long));
    if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void**>(&h->pair_host), kSlots * 2 * sizeof(long long), hipHostMallocDefault);
    if (e != hipSuccess) { delete h; return pulse::fail_hip((int)e, "pulse_stoprule_create"); }
    std::memset(h->pair_host, 0, kSlots * 2 * sizeof(long long));
    *out = h;
    return 0;
}

int pulse_stoprule_destroy(void* handle) {
    PulseStopRule* h = static_cast<PulseStopRule*>(handle);
    if (!h) return 0;
    (void)hipStreamSynchronize(h->side);
    for (int i = 0; i < kSlots; ++i) { (void)hipEventDestroy(h->ready[i]); (void)hipEventDestroy(h->copied[i]); }
    (void)hipFree(h->partials_dev); (void)hipFree(h->pair_dev); (void)hipHostFree(h->pair_host);
    (void)hipStreamDestroy(h->side);
    delete h;
    return 0;
}

int pulse_stoprule_submit(void* handle, const uint8_t* flags, int32_t n, void* stream) {
    PulseStopRule* h = static_cast<PulseStopRule*>(handle);
    if (!h || !flags || n < 0) return pulse::fail(PULSE_EINVAL, "pulse_stoprule_submit: bad argument");
    const int grid = n == 0 ? 1 : min(kMaxPartials, (n + kBlock - 1) / kBlock);
    uint32_t* partials = nullptr;
    if (int rc = pulse::stoprule_claim(h, grid, &partials)) return rc;
    hipLaunchKernelGGL(stoprule_flags_kernel, dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, flags, n, partials);
    return pulse::stoprule_commit(h, grid, (hipStream_t)stream);
}

int pulse_stoprule_counts(void* handle, int64_t* local, int64_t* global, int32_t* have) {
    PulseStopRule* h = static_cast<PulseStopRule*>(handle);
    if (!h || !local || !global || !have) return pulse::fail(PULSE_EINVAL, "pulse_stoprule_counts: null argument");
    const long long c = h->submitted - 1 - h->lag;          // the chunk whose verdict is due now
    *have = 0; *local = 0; *global = 0;
    if (c < h->epoch_first) return 0;
    const int slot = (int)(c % kSlots);
    const hipError_t e = hipEventSynchronize(h->copied[slot]);       // normally long complete: `lag` chunks are queued behind it
    if (e != hipSuccess) return pulse::fail_hip((int)e, "pulse_stoprule_counts: hipEventSynchronize");
    *local = h->pair_host[2 * slot]; *global = h->pair_host[2 * slot + 1]; *have = 1;
    return 0;
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
    h->epoch_first = h->submitted;           // chunks of the finished episode never decide anything again
    return 0;
}

}  // extern "C"
