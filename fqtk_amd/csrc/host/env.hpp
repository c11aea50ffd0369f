// env.hpp -- developer / test switches in the environment.
#pragma once
#include <sched.h>

#include <cstdio>
#include <algorithm>
#include <cstdlib>

namespace fqtk_host {

// Set, not empty and not "0" (an empty variable left behind by a shell script must not count).
inline bool env_on(const char *name) {
    const char *v = std::getenv(name);
    return v && *v && !(v[0] == '0' && v[1] == 0);
}

// A number from the environment (A/B runs), or the default.
inline long env_num(const char *name, long dflt) {
    const char *v = std::getenv(name);
    return v && *v ? std::atol(v) : dflt;
}

// CPUs this process can really use: the affinity mask, capped by the cgroup CPU quota (containers often show every
// logical CPU of the host under a quota of a few: threads beyond the quota only take turns).
inline unsigned usable_cpus() {
    unsigned n = 0;
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) n = (unsigned)CPU_COUNT(&set);
    if (n == 0) n = 1;
    if (FILE *f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {   // cgroup v2: "<quota|max> <period>"
        char q[32] = {0};
        long long period = 0;
        if (std::fscanf(f, "%31s %lld", q, &period) == 2 && period > 0 && q[0] != 'm') {
            const long long quota = std::atoll(q);
            if (quota > 0) n = std::min<unsigned>(n, (unsigned)std::max<long long>(1, quota / period));
        }
        std::fclose(f);
    } else {
        long long quota = -1, period = 0;
        if (FILE *g = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (std::fscanf(g, "%lld", &quota) != 1) quota = -1; std::fclose(g); }
        if (FILE *g = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (std::fscanf(g, "%lld", &period) != 1) period = 0; std::fclose(g); }
        if (quota > 0 && period > 0) n = std::min<unsigned>(n, (unsigned)std::max<long long>(1, quota / period));
    }
    return n;
}

}  // namespace fqtk_host
