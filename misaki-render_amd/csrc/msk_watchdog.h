// msk_watchdog.h — the progress watchdog of the wavefront loop (round 5).
//
// SamplingIntegrator::render returns, and main.cpp:55-57 catches what it throws; a caller of msk_gpu_render must get the same
// guarantee from a back end whose loop waits on a device.  Two ways the loop of msk_gpu.hip (run_wavefront) could wait for ever:
//   * a sync group's kernels never finish (a wave that spins, a device that stopped answering): hipStreamSynchronize has no
//     timeout, so the loop waits with hipStreamQuery polls instead and gives up after MSK_WATCHDOG_S seconds (default 120 — a
//     sync group is 8 iterations of a few milliseconds; 0 = plain hipStreamSynchronize, no wall limit);
//   * the kernels finish but nothing moves: the counters the host reads after every group (samples finished, segments traced,
//     samples not yet started, live paths) are the same for MSK_WATCHDOG_GROUPS consecutive groups (default 64).
// Either ends the render with MSK_ERR_HIP "no progress ...": the context is marked lost — every later call on it fails at
// once, msk_gpu_shutdown releases host memory only (destroying a stream that still holds a hung kernel would wait for it) —
// and the caller is expected to exit or to start over in a fresh child process.  Nothing here restarts or replaces a process
// that has touched the GPU.
//
// The decision logic is plain C++ (no HIP): tests/native/watchdog_check.cpp exercises it on the CPU.
#pragma once
#include <chrono>
#include <cstdint>
#include <cstdlib>

namespace mskwd {

struct Limits {
    double wall_s = 120.0;          // per sync group; <= 0: no wall limit
    uint32_t stalled_groups = 64;   // consecutive groups without any change of the counters; 0: not checked
};

// MSK_WATCHDOG_S / MSK_WATCHDOG_GROUPS; anything that is not a number keeps the default
inline Limits limits_from_env() {
    Limits l;
    if (const char *s = std::getenv("MSK_WATCHDOG_S")) { char *e = nullptr; const double v = std::strtod(s, &e); if (e != s && *e == '\0' && v >= 0.0 && v < 1e9) l.wall_s = v; }
    if (const char *s = std::getenv("MSK_WATCHDOG_GROUPS")) { char *e = nullptr; const unsigned long v = std::strtoul(s, &e, 10); if (e != s && *e == '\0' && v < (1ul << 31)) l.stalled_groups = (uint32_t) v; }
    return l;
}

// What the host knows after a sync group: everything k_reduce_ctl sums.  Any change is progress (segments alone grow with every
// sweep that holds a live path; a loop whose kernels run but change nothing leaves all four where they were).
struct Counters {
    unsigned long long samples_done = 0, segments = 0, remaining = 0, live = 0;
    bool operator==(const Counters &o) const { return samples_done == o.samples_done && segments == o.segments && remaining == o.remaining && live == o.live; }
};

enum Verdict { OK = 0, STALLED = 1, TIMED_OUT = 2 };

class Progress {
public:
    explicit Progress(const Limits &l) : m_limits(l) {}
    // after every sync group that did not end the loop; STALLED once the counters have stood still for `stalled_groups` groups
    Verdict group_done(const Counters &c) {
        if (m_have && c == m_last) { ++m_stalled; }
        else { m_stalled = 0; m_last = c; m_have = true; }
        return (m_limits.stalled_groups && m_stalled >= m_limits.stalled_groups) ? STALLED : OK;
    }
    uint32_t stalled() const { return m_stalled; }
    // the wall-clock side: has a wait that began `waited_s` seconds ago run out?
    Verdict waited(double waited_s) const { return (m_limits.wall_s > 0.0 && waited_s > m_limits.wall_s) ? TIMED_OUT : OK; }
    const Limits &limits() const { return m_limits; }
private:
    Limits m_limits;
    Counters m_last;
    bool m_have = false;
    uint32_t m_stalled = 0;
};

// How long the next poll of a wait sleeps: nothing (a yield) for the first 20 ms — a sync group normally takes a few milliseconds
// and the loop's next launches wait for this thread —, then 0.2 ms steps, 5 ms once a second has passed.
inline unsigned poll_sleep_us(double waited_s) { return waited_s < 0.020 ? 0u : waited_s < 1.0 ? 200u : 5000u; }

}  // namespace mskwd

// the waiting side needs the HIP runtime: msk_gpu.hip defines MSK_WATCHDOG_SYNC before it includes this file, the CPU test does not
#ifdef MSK_WATCHDOG_SYNC
#include <hip/hip_runtime.h>
#include <thread>
namespace mskwd {
// hipStreamSynchronize with a wall limit: hipSuccess, the stream's error, or hipErrorNotReady when `p` says the wait ran out
inline hipError_t sync(hipStream_t stream, const Progress &p) {
    if (p.limits().wall_s <= 0.0) return hipStreamSynchronize(stream);
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        const hipError_t e = hipStreamQuery(stream);
        if (e != hipErrorNotReady) return e;
        const double waited = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (p.waited(waited) == TIMED_OUT) { (void) hipGetLastError(); return hipErrorNotReady; }
        const unsigned us = poll_sleep_us(waited);
        if (us) std::this_thread::sleep_for(std::chrono::microseconds(us)); else std::this_thread::yield();
    }
}
}  // namespace mskwd
#endif
