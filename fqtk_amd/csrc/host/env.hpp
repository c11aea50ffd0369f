// env.hpp -- developer / test switches in the environment.
#pragma once
#include <cstdlib>

namespace fqtk_host {

// Set, not empty and not "0" (an empty variable left behind by a shell script must not count).
inline bool env_on(const char *name) {
    const char *v = std::getenv(name);
    return v && *v && !(v[0] == '0' && v[1] == 0);
}

}  // namespace fqtk_host
