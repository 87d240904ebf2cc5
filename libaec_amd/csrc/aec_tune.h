// aec_tune.h -- measurement knobs.  The library as shipped reads ONE environment variable (AEC_ABI_TRACE,
// aec_abi.cpp).  Everything else -- forced kernel variants, geometry overrides, diagnostics -- exists only in a
// build with -DAEC_TUNING (make tuning: libaec_amd/lib/tuning/libaec.so.0, loaded by the tests and benches that
// compare variants via AEC_AMD_LIB); without it the functions below return their defaults.
#pragma once
#include <stdint.h>
#include <stdlib.h>

namespace aec {

#ifdef AEC_TUNING
inline uint32_t tune(const char *name, uint32_t dflt)
{
    const char *e = getenv(name);
    return e ? (uint32_t)atoi(e) : dflt;
}
inline bool tune_set(const char *name) { return getenv(name) != nullptr; }
#else
inline uint32_t tune(const char *, uint32_t dflt) { return dflt; }
inline bool tune_set(const char *) { return false; }
#endif

}  // namespace aec
