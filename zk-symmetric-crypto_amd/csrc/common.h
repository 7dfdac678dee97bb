// Shared host-side helpers for the HIP prover (error checks, device buffers).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <atomic>
#include <stdexcept>
#include <string>

// A/B switches of past scheduling / tiling experiments (ZKPROVE_SORT_AHEAD, ZKPROVE_NTT_STREAM, ZKPROVE_COPY_PRIO, ZKPROVE_SOLVER_PACK,
// ZKPROVE_SOLVER_OCC, ZKPROVE_SEG_THREADS, ZKPROVE_G2_SIDE, ZKPROVE_NTT_PIPE, ZKPROVE_COMB_Q, ZKPROVE_COMB_PAT_REG, ZKPROVE_COMB_FOLD_TAIL,
// ZKPROVE_MSM_PIECE, ZKPROVE_SOLVER_RUNS, ZKPROVE_SOLVER_RUN_WAVES) are read by the experiments build only; the product library runs the measured winner and keeps the operational
// knobs (ZKPROVE_COMB, ZKPROVE_COMB_K, ZKPROVE_COMB_MIN_BATCH, ZKPROVE_WCOMB_K, ZKPROVE_PIPE_DEPTH, ZKPROVE_DEVICES, ZKPROVE_VK_DIR,
// ZKPROVE_MEMLOG, ZKPROVE_TIMELINE, ZKPROVE_ALLOW_UNPINNED_PK).
#ifdef ZK_EXPERIMENTS
#define ZK_AB_ENV(name) getenv(name)
#else
#define ZK_AB_ENV(name) ((const char *)nullptr)
#endif

namespace zk {

struct ZkError : public std::runtime_error {
    explicit ZkError(const std::string &m) : std::runtime_error(m) {}
};

#define ZK_HIP(expr)                                                                                   \
    do {                                                                                               \
        hipError_t _e = (expr);                                                                        \
        if (_e != hipSuccess)                                                                          \
            throw zk::ZkError(std::string("HIP error: ") + hipGetErrorString(_e) + " at " + __FILE__ + \
                              ":" + std::to_string(__LINE__) + " (" #expr ")");                        \
    } while (0)

// RAII device buffer
// A large allocation must leave DEVICE_HEADROOM free: a device run down to its last megabyte cannot give the next kernel its scratch
// memory, and the HIP runtime aborts the process for that (HSA_STATUS_ERROR_OUT_OF_RESOURCES) instead of returning an error.  Throws
// ZkError like a failed hipMalloc, so the memory-pressure valves (prover.hip, relieve_pressure) see both the same way.
void require_headroom(size_t bytes);
void memlog_large_alloc(size_t bytes);      // hostutil.cpp: ZKPROVE_MEMLOG=1 prints allocations >= 256 MiB with their call sites
template <class T>
struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    DevBuf() = default;
    explicit DevBuf(size_t count) { alloc(count); }
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    DevBuf(DevBuf &&o) noexcept : p(o.p), n(o.n) { o.p = nullptr; o.n = 0; }
    DevBuf &operator=(DevBuf &&o) noexcept {
        if (this != &o) { release(); p = o.p; n = o.n; o.p = nullptr; o.n = 0; }
        return *this;
    }
    ~DevBuf() { release(); }
    void alloc(size_t count) {
        release();
        if (count * sizeof(T) >= ((size_t)256 << 20)) require_headroom(count * sizeof(T));
        if (count) ZK_HIP(hipMalloc((void **)&p, count * sizeof(T)));      // throws on failure with the buffer left empty (n = 0)
        n = count;
        if (count * sizeof(T) >= ((size_t)256 << 20)) memlog_large_alloc(count * sizeof(T));
    }
    void ensure(size_t count) { if (count > n) alloc(count); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr; n = 0;
    }
    void upload(const T *src, size_t count, hipStream_t s = 0) {
        ensure(count);
        if (count) ZK_HIP(hipMemcpyAsync(p, src, count * sizeof(T), hipMemcpyHostToDevice, s));
    }
    void download(T *dst, size_t count, hipStream_t s = 0) const {
        if (count) ZK_HIP(hipMemcpyAsync(dst, p, count * sizeof(T), hipMemcpyDeviceToHost, s));
    }
    void zero(hipStream_t s = 0) { if (n) ZK_HIP(hipMemsetAsync(p, 0, n * sizeof(T), s)); }
    size_t bytes() const { return n * sizeof(T); }
};

static inline unsigned ceil_div(size_t a, size_t b) { return (unsigned)((a + b - 1) / b); }

// ---- device binding ------------------------------------------------------------------------------------------
// hipSetDevice is per THREAD.  Every exported entry point and every worker thread binds the device it is about to
// use before its first HIP call (cabi.cpp); nothing relies on a binding made earlier by another thread.
// default_device(): the process-wide choice -- ZkpSetDevice(k), else LOCAL_RANK (one process per GPU under
// torchrun), else 0.
int default_device();
void set_default_device(int dev);
void bind_device(int dev);                       // throws ZkError
inline void bind_default_device() { bind_device(default_device()); }
// Scope guard of the prover's entry points: binds the replica's device on the calling thread and puts the thread's previous device
// back on exit -- the current device is state of the HOST application's thread (torch, another HIP user), not of this library.
struct DeviceScope {
    int prev = -1;
    explicit DeviceScope(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        bind_device(dev);
    }
    ~DeviceScope() { if (prev >= 0) (void)hipSetDevice(prev); }
    DeviceScope(const DeviceScope &) = delete;
    DeviceScope &operator=(const DeviceScope &) = delete;
};

void verifier_reset();                           // verifier.cpp: drop the device images of the verifying keys (ZkpReset)
void set_last_error(const std::string &msg);     // cabi.cpp: what zkp_last_error() returns on this thread

// hipFuncSetAttribute is per device: run fn the first time each device passes through
struct PerDeviceOnce {
    std::atomic<uint64_t> done{0};
    template <class Fn>
    void run(Fn fn) {
        int d = 0;
        ZK_HIP(hipGetDevice(&d));
        const uint64_t bit = 1ull << (d & 63);
        if (done.load(std::memory_order_acquire) & bit) return;
        fn();                                    // idempotent: a racing second call is harmless
        done.fetch_or(bit, std::memory_order_release);
    }
};

}  // namespace zk
