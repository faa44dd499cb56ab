// api_internal.h -- shared between mzd_host.cpp and m4ri_hip_api.cpp
#pragma once
[[noreturn]] void gf2_die(const char *msg);
