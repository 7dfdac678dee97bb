// Shared host-side helpers for the HIP prover (error checks, device buffers).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdexcept>
#include <string>

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
        n = count;
        if (count) ZK_HIP(hipMalloc((void **)&p, count * sizeof(T)));
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

}  // namespace zk
