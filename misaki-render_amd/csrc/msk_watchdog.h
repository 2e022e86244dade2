// msk_watchdog.h — how a render waits for its device, and when it stops waiting (rounds 5-6).
//
// SamplingIntegrator::render returns, and main.cpp:55-57 catches what it throws; a caller of msk_gpu_render must get the same
// guarantee from a back end whose loop waits on a device.  Two ways a render could wait for ever:
//   * queued work never finishes (a wave that spins, a device that stopped answering): hipStreamSynchronize has no timeout, so
//     EVERY wait of a render — a sync group of the wavefront loop (wait_any below), the film replay, Film::put, the film's
//     copy-back, the one long kernel of MSK_RNG_PCG_BLOCK (msk_gpu.hip: ctx_sync) — is a timed wait and gives up after
//     MSK_WATCHDOG_S seconds (default 120 per sync group — eight iterations of a few milliseconds —, 30 x that for the
//     one-kernel render; 0 = no wall limit);
//   * the kernels finish but nothing moves: the counters the host reads after every group (samples finished, segments traced,
//     samples not yet started, live paths) are the same for MSK_WATCHDOG_GROUPS consecutive groups (default 64).
// Either ends the render with MSK_ERR_HIP "no progress ...": the context is marked lost — every later call on it fails at
// once, teardown releases host memory only (hipFree / hipStreamDestroy would wait for the kernel that never finished; local
// device buffers of the failed call are dropped, not freed) — and the caller is expected to exit or to start over in a fresh
// child process.  Nothing here restarts or replaces a process that has touched the GPU.
//
// HOW a thread waits is MSK_WAIT (below): by default it sleeps on a condition variable that a host function queued behind the
// awaited work signals — rounds 4-5 polled hipStreamQuery with yields, one spinning thread per wavefront loop.
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
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <thread>
namespace mskwd {

// HOW a host thread waits for the end of a sync group (MSK_WAIT; round 6, DESIGN.md section 7 "host-side waits"): eight ranks of
// one node share its CPU quota (16 CPUs on the bench boxes), and a wait that spins takes a CPU for the whole render.
//   callback  hipLaunchHostFunc behind the group's last command signals a condition variable the waiting thread sleeps on
//             (timed: the wall limit holds) — no CPU while the device works
//   sleep     hipStreamQuery every 50 us from the start (no yield phase): a few per cent of one CPU per waiting thread
//   poll      rounds 4-5: hipStreamQuery + yield for the first 20 ms of a wait (spins through every ordinary group), then sleeps
//   event     a hipEventBlockingSync event + hipEventSynchronize: blocks in the driver; NO wall limit (the stall check stays)
//   sync      hipStreamSynchronize: the runtime's own wait; NO wall limit
enum WaitMode { WAIT_POLL = 0, WAIT_SLEEP = 1, WAIT_CALLBACK = 2, WAIT_EVENT = 3, WAIT_SYNC = 4 };
#ifndef MSK_WAIT_DEFAULT
#define MSK_WAIT_DEFAULT WAIT_CALLBACK
#endif
inline const char *wait_mode_name(WaitMode m) { static const char *n[] = {"poll", "sleep", "callback", "event", "sync"}; return n[(int) m]; }
inline WaitMode wait_mode_from_env() {
    const char *s = std::getenv("MSK_WAIT");
    if (s) for (int m = 0; m <= (int) WAIT_SYNC; ++m) if (!std::strcmp(s, wait_mode_name((WaitMode) m))) return (WaitMode) m;
    return (WaitMode) MSK_WAIT_DEFAULT;
}

#define MSK_WAIT_SLOTS 8
// What the host functions of WAIT_CALLBACK signal.  Allocated on its own and LEAKED with a lost context: a host function that
// was queued behind a kernel which never finished may still fire after the context is gone.
struct Hub {
    std::mutex m;
    std::condition_variable cv;
    uint64_t fired[MSK_WAIT_SLOTS] = {};
    uint64_t armed[MSK_WAIT_SLOTS] = {};                // (written by the arming thread only)
    struct Slot { Hub *hub; int k; } slots[MSK_WAIT_SLOTS];
    Hub() { for (int k = 0; k < MSK_WAIT_SLOTS; ++k) slots[k] = Slot{this, k}; }
    static void fire(void *arg) {
        Slot *s = (Slot *) arg;
        { std::lock_guard<std::mutex> lk(s->hub->m); ++s->hub->fired[s->k]; }
        s->hub->cv.notify_all();
    }
};

// one thing a thread waits for: everything queued on `stream` up to arm()
struct Ticket {
    hipStream_t stream = nullptr;
    int slot = 0;                   // of the hub (one per part of the pool; waits outside the loop use slot 0)
    hipEvent_t event = nullptr;     // WAIT_EVENT: a hipEventBlockingSync event owned by the context
    uint64_t seq = 0;
};

struct Waiter {
    WaitMode mode = (WaitMode) MSK_WAIT_DEFAULT;
    Hub *hub = nullptr;
};

// after the last command of what is to be waited for has been queued
inline hipError_t arm(const Waiter &w, Ticket &t) {
    if (w.mode == WAIT_CALLBACK) { t.seq = ++w.hub->armed[t.slot]; return hipLaunchHostFunc(t.stream, Hub::fire, &w.hub->slots[t.slot]); }
    if (w.mode == WAIT_EVENT) return hipEventRecord(t.event, t.stream);
    return hipSuccess;
}

// Waits until ONE of the n armed tickets is complete: its index, with *err = what its stream reported; or -1 when the wall limit
// of `p` ran out first (poll / sleep / callback).  event / sync wait for tickets[0] — the caller lists the oldest first.
inline int wait_any(const Waiter &w, Ticket *const *tickets, int n, const Progress &p, hipError_t *err) {
    *err = hipSuccess;
    const double limit = p.limits().wall_s;
    if (w.mode == WAIT_SYNC || (w.mode == WAIT_EVENT && !tickets[0]->event)) { *err = hipStreamSynchronize(tickets[0]->stream); return 0; }
    if (w.mode == WAIT_EVENT) { *err = hipEventSynchronize(tickets[0]->event); return 0; }
    if (w.mode == WAIT_CALLBACK) {
        std::unique_lock<std::mutex> lk(w.hub->m);
        int hit = -1;
        auto ready = [&]() { for (int i = 0; i < n; ++i) if (w.hub->fired[tickets[i]->slot] >= tickets[i]->seq) { hit = i; return true; } return false; };
        if (limit > 0.0) { if (!w.hub->cv.wait_for(lk, std::chrono::duration<double>(limit), ready)) return -1; }
        else w.hub->cv.wait(lk, ready);
        return hit;
    }
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        for (int i = 0; i < n; ++i) {
            const hipError_t e = hipStreamQuery(tickets[i]->stream);
            if (e != hipErrorNotReady) { *err = e; return i; }
        }
        (void) hipGetLastError();
        const double waited = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (p.waited(waited) == TIMED_OUT) return -1;
        const unsigned us = w.mode == WAIT_SLEEP ? (waited < 1.0 ? 50u : 5000u) : poll_sleep_us(waited);
        if (us) std::this_thread::sleep_for(std::chrono::microseconds(us)); else std::this_thread::yield();
    }
}

}  // namespace mskwd
#endif
