// knobs.cpp -- see knobs.h
#include <stdlib.h>
#include <algorithm>
#include "knobs.h"

namespace crp {

static int env_int(const char *name, int dflt)
{
    const char *v = getenv(name);
    return (v != NULL && *v != 0) ? atoi(v) : dflt;
}

const Knobs &knobs()
{
    static const Knobs k = [] {
        Knobs x;
        x.t2_chain = std::max(0, std::min(64, env_int("CRPSPMM_T2_CHAIN", 0)));
        x.t2_latorder = env_int("CRPSPMM_T2_LATORDER", 1) != 0;
        return x;
    }();
    return k;
}

}  // namespace crp
