// The decision logic of the wavefront loop's watchdog (misaki-render_amd/csrc/msk_watchdog.h), on the CPU: compiled and run by
// tests/test_watchdog.py.  Prints "ok" or "FAIL ...".
#include <cstdio>
#include <cstdlib>
#include "../../misaki-render_amd/csrc/msk_watchdog.h"

#define CHECK(c) do { if (!(c)) { std::printf("FAIL line %d: %s\n", __LINE__, #c); return 1; } } while (0)

int main() {
    using namespace mskwd;
    {   // a render that moves: never stalled, whatever changes
        Progress p(Limits{120.0, 4});
        Counters c;
        for (int g = 0; g < 1000; ++g) { c.segments += 5; if (g % 3 == 0) c.samples_done += 1; CHECK(p.group_done(c) == OK); }
        // the thin end: only `live` shrinks
        for (int g = 0; g < 50; ++g) { c.live = 1000 - g; c.segments += 1; CHECK(p.group_done(c) == OK); }
    }
    {   // the counters stand still: STALLED at the limit, not before; any change starts the count again
        Progress p(Limits{120.0, 4});
        Counters c{10, 20, 30, 40};
        CHECK(p.group_done(c) == OK);                   // first sight
        CHECK(p.group_done(c) == OK && p.stalled() == 1);
        CHECK(p.group_done(c) == OK && p.group_done(c) == OK && p.stalled() == 3);
        c.remaining -= 1;
        CHECK(p.group_done(c) == OK && p.stalled() == 0);
        for (int g = 0; g < 3; ++g) CHECK(p.group_done(c) == OK);
        CHECK(p.group_done(c) == STALLED && p.stalled() == 4);
        CHECK(p.group_done(c) == STALLED);
    }
    {   // stalled_groups = 0 switches the check off
        Progress p(Limits{120.0, 0});
        Counters c{1, 2, 3, 4};
        for (int g = 0; g < 500; ++g) CHECK(p.group_done(c) == OK);
    }
    {   // the wall side
        Progress p(Limits{2.5, 64});
        CHECK(p.waited(0.0) == OK && p.waited(2.5) == OK && p.waited(2.5001) == TIMED_OUT);
        Progress off(Limits{0.0, 64});
        CHECK(off.waited(1e12) == OK);
        CHECK(poll_sleep_us(0.001) == 0u && poll_sleep_us(0.5) == 200u && poll_sleep_us(3.0) == 5000u);
    }
    {   // the environment: numbers are taken, anything else keeps the defaults
        unsetenv("MSK_WATCHDOG_S"); unsetenv("MSK_WATCHDOG_GROUPS");
        Limits d = limits_from_env();
        CHECK(d.wall_s == 120.0 && d.stalled_groups == 64);
        setenv("MSK_WATCHDOG_S", "7.5", 1); setenv("MSK_WATCHDOG_GROUPS", "9", 1);
        Limits a = limits_from_env();
        CHECK(a.wall_s == 7.5 && a.stalled_groups == 9);
        setenv("MSK_WATCHDOG_S", "soon", 1); setenv("MSK_WATCHDOG_GROUPS", "-3x", 1);
        Limits b = limits_from_env();
        CHECK(b.wall_s == 120.0 && b.stalled_groups == 64);
        setenv("MSK_WATCHDOG_S", "0", 1);
        CHECK(limits_from_env().wall_s == 0.0);
    }
    std::printf("ok\n");
    return 0;
}
