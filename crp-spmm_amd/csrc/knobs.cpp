// knobs.cpp -- see knobs.h
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include "knobs.h"

namespace crp {

static int env_int(const char *name, int dflt)
{
    const char *v = getenv(name);
    return (v != NULL && *v != 0) ? atoi(v) : dflt;
}

static bool env_is(const char *name, const char *what)
{
    const char *v = getenv(name);
    return v != NULL && strcmp(v, what) == 0;
}

static Knobs read_knobs()
{
    Knobs k;
    k.timing = env_int("CRPSPMM_TIMING", 0) != 0;
    k.num_threads = std::max(0, env_int("CRPSPMM_NUM_THREADS", env_int("OMP_NUM_THREADS", 0)));
    k.sync_release = getenv("CRPSPMM_SYNC_RELEASE") != NULL;
    k.spmm_variant = env_int("CRPSPMM_SPMM_VARIANT", 0);
    k.reorder = env_int("CRPSPMM_REORDER", -1);
    k.panel_order = env_int("CRPSPMM_PANEL_ORDER", -1);
    k.narrow_max = env_int("CRPSPMM_NARROW_MAX", 0);
    k.team2_compact = env_int("CRPSPMM_TEAM2_COMPACT", -1);
    k.team2r = env_int("CRPSPMM_TEAM2R", -1);
    k.t2_latorder = env_int("CRPSPMM_T2_LATORDER", 1) != 0;
    k.overlap = env_int("CRPSPMM_OVERLAP", 1);
    k.exchange_host = env_is("CRPSPMM_EXCHANGE", "host");
    k.replicate_host = env_is("CRPSPMM_REPLICATE", "host");
    {
        const char *v = getenv("CRPSPMM_RCCL_TIMEOUT");
        const double t = v ? atof(v) : 120.0;
        k.rccl_timeout = t > 0.0 ? t : 120.0;
    }
    k.rccl_blocking = env_int("CRPSPMM_RCCL_BLOCKING", 0) != 0;
    k.engine_a_static = env_int("CRPSPMM_ENGINE_A_STATIC", -1);
    return k;
}

const Knobs &knobs()
{
    static const bool live = env_int("CRPSPMM_KNOBS_LIVE", 0) != 0;
    static Knobs k = read_knobs();
    if (live) k = read_knobs();         // (tests: a knob may change between two calls; not thread-safe, and not meant to be)
    return k;
}

}  // namespace crp
