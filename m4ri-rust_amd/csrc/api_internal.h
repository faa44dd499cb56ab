// api_internal.h -- shared between mzd_host.cpp and m4ri_hip_api.cpp
#pragma once
[[noreturn]] void gf2_die(const char *msg);
#include "../../include/m4ri_hip.h"
mzd_t *gf2_mzd_init_uncleared(rci_t r, rci_t c);  // mzd_init without the memset (callers overwrite every word)
